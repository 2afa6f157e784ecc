#!/bin/bash
# usage: tools/rj_rows.sh — tile height of the fused projection head (WL_RJ_ROWS=16|32) at several sizes: ms per step, head (both solves)
run() { name=$1; shift; env "$@" python bench.py --phases --steps ${STEPS} --warmup 3 --size $SIZE --no-cpu-baseline > gpurun_out/cg.json 2> gpurun_out/cg.err; python - <<PY
import json
j=json.loads(open("gpurun_out/cg.json").read().strip().splitlines()[-1]); p=j["phases_ms_per_step"]
print("$SIZE", "$name", "step", round(j["ms_per_step"],3), "head", round(p.get("residual",0),3))
PY
}
for SIZE in 256 384 512; do
STEPS=$([ $SIZE = 512 ] && echo 8 || echo 30)
run auto A=1; run rows16 WL_RJ_ROWS=16; run rows32 WL_RJ_ROWS=32; run auto A=1
done
