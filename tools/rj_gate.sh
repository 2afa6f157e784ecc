#!/bin/bash
# usage: tools/rj_gate.sh — size gate of the fused projection head: default (8 Mi cells) vs forced (WL_OPT_resjac_min=0) at 128³…224³
run() { name=$1; shift; env "$@" python bench.py --steps 150 --warmup 20 --size ${SIZE} --no-cpu-baseline --no-phases > gpurun_out/cg.json 2> gpurun_out/cg.err; python - <<PY
import json
j=json.loads(open("gpurun_out/cg.json").read().strip().splitlines()[-1])
print("${SIZE}", "$name", round(j["ms_per_step"],4))
PY
}
for SIZE in 128 160 192 224; do
run base A=1; run head WL_OPT_resjac_min=0; run base A=1; run head WL_OPT_resjac_min=0
done
