#!/bin/bash
# usage: tools/small_gates.sh  — 128³ / 256³ steps with the size-gated kernels forced on / the LDS tail off (no phase events)
set -e
run() { name=$1; shift; env "$@" python bench.py --steps 200 --warmup 20 --size ${SIZE} --no-cpu-baseline --no-phases > gpurun_out/sm_${SIZE}_$name.json 2> gpurun_out/sm_${SIZE}_$name.err; python - <<PY
import json
j=json.loads(open("gpurun_out/sm_${SIZE}_$name.json").read().strip().splitlines()[-1])
print("${SIZE}", "$name", round(j["ms_per_step"],4), j["config"].get("mean_pois_n"))
PY
}
for SIZE in 128 256; do
run base A=1
run gtail WL_OPT_tail_lds=0
run convt WL_OPT_convt_min=0
run resjac WL_OPT_resjac_min=0
run base2 A=1
done
