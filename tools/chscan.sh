#!/bin/bash
# usage: tools/chscan.sh  — number of z-chunks of smoother kernels A / B at 512³ (WL_PAIR_CH_A / WL_PAIR_CH_B), ms per step over both solves
run() { python bench.py --phases --steps 8 --warmup 3 --size 512 --no-cpu-baseline > gpurun_out/ch.json 2> gpurun_out/ch.err; python - <<PY
import json
j=json.loads(open("gpurun_out/ch.json").read().strip().splitlines()[-1]); p=j["phases_ms_per_step"]
print("$1", "step", round(j["ms_per_step"],3), "A", round(p.get("gsrb_A",0),3), "B", round(p.get("gsrb_B",0),3))
PY
}
run base
for c in 4 7 10 13 16 20 24 32; do WL_PAIR_CH_A=$c WL_PAIR_CH_B=$c run ch$c; done
run base
