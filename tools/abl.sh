#!/bin/bash
# usage: tools/abl.sh variant...  — per-launch times of the finest-level smoother kernels A / B (bench.py's HIP-event roofline section) for each
# libwlhip_<variant>.so ("default" = the product library).  Made for timing ablations whose results are wrong: only avg_ms is meaningful.
for v in "$@"; do
  if [ "$v" = default ]; then unset WLHIP_LIB; else export WLHIP_LIB=$PWD/waterlily.jl_amd/libwlhip_$v.so; fi
  timeout -k 10 240 python bench.py --steps ${STEPS:-4} --warmup 1 --size ${SIZE:-512} --no-cpu-baseline > gpurun_out/abl_$v.json 2> gpurun_out/abl_$v.err
  python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/abl_$v.json").read().strip().splitlines()[-1])
    k=j["roofline"]["kernels"]
    print("$v", "A", round(k["A"]["avg_ms"],4), "B", round(k["B"]["avg_ms"],4), "launches", j["roofline"]["launches"], "step", round(j["ms_per_step"],2), "n", j["config"]["mean_pois_n"])
except Exception as e:
    print("$v", "failed", e)
PY
done
