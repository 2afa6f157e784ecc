#!/bin/bash
# usage: tools/minnx_gate.sh — pair kernels on the 34-wide level too (WL_PAIR_MIN_NX=34) vs the one-cell blocked kernels there
run() { name=$1; shift; env "$@" python bench.py --steps 150 --warmup 20 --size ${SIZE} --no-cpu-baseline --no-phases > gpurun_out/cg.json 2> gpurun_out/cg.err; python - <<PY
import json
j=json.loads(open("gpurun_out/cg.json").read().strip().splitlines()[-1])
print("${SIZE}", "$name", round(j["ms_per_step"],4), j["config"]["smoother_kinds"])
PY
}
for SIZE in 128 256; do
run base A=1
run nx34 WL_PAIR_MIN_NX=34
run base A=1
run nx34 WL_PAIR_MIN_NX=34
done
