#!/bin/bash
# usage: tools/rj_chunks.sh [SIZE] — number of z-chunks of the fused projection head (WL_RJ_CHUNKS), ms per step of the head (both solves)
SIZE=${1:-512}
run() { python bench.py --phases --steps 8 --warmup 3 --size $SIZE --no-cpu-baseline > gpurun_out/cg.json 2> gpurun_out/cg.err; python - <<PY
import json
j=json.loads(open("gpurun_out/cg.json").read().strip().splitlines()[-1]); p=j["phases_ms_per_step"]
print("$SIZE", "$1", "step", round(j["ms_per_step"],3), "head", round(p.get("residual",0),3))
PY
}
run model
for c in 5 8 11 14 17 18 24; do WL_RJ_CHUNKS=$c run ch$c; done
run model
