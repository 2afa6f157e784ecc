#!/bin/bash
# usage: tools/abv.sh variant...   — bench each libwlhip_<variant>.so (and "default") with --phases; prints ms/step and the conv_diff phase
for v in "$@"; do
  if [ "$v" = default ]; then unset WLHIP_LIB; else export WLHIP_LIB=$PWD/waterlily.jl_amd/libwlhip_$v.so; fi
  python bench.py --phases --steps ${STEPS:-10} --warmup 3 --size ${SIZE:-512} --no-cpu-baseline > gpurun_out/abv_$v.json 2> gpurun_out/abv_$v.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/abv_$v.json").read().strip().splitlines()[-1])
p=j["phases_ms_per_step"]
print("$v", "step", round(j["ms_per_step"],3), "conv", round(p["conv_diff"],3), "smooth", round(p.get("smooth",0),3), "A", round(p.get("gsrb_A",0),3), "B", round(p.get("gsrb_B",0),3), "jac", round(p.get("jacobi",0),3), "resid", round(p.get("residual",0),3), "pois_n", j["config"]["mean_pois_n"])
PY
done
