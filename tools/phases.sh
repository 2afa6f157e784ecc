#!/bin/bash
# usage: tools/phases.sh SIZE [STEPS]  — phase table (ms per step) of bench.py --phases
python bench.py --phases --steps ${2:-10} --warmup 3 --size $1 --no-cpu-baseline > gpurun_out/ph_$1.json 2> gpurun_out/ph_$1.err
python - <<PY
import json
j=json.loads(open("gpurun_out/ph_$1.json").read().strip().splitlines()[-1])
print("size", j["config"]["size"], "ms/step", round(j["ms_per_step"],4), "pois_n", j["config"]["mean_pois_n"])
p=j["phases_ms_per_step"]
for k,v in sorted(p.items(), key=lambda kv:-kv[1]): print(f"  {k:24s} {v:8.3f}")
PY
