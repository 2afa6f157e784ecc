#!/bin/bash
# usage: tools/capscan.sh cap...  — bench.py --phases with WL_MARCH_CHUNK_CAP=cap (z-chunk length of the streaming z-march kernels: projection tails, CFL, div/residual); prints the phase table's projection/cfl rows
for c in "$@"; do
  WL_MARCH_CHUNK_CAP=$c python bench.py --phases --steps ${STEPS:-8} --warmup 2 --size ${SIZE:-512} --no-cpu-baseline > gpurun_out/cap_$c.json 2> gpurun_out/cap_$c.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/cap_$c.json").read().strip().splitlines()[-1])
p=j["phases_ms_per_step"]
print("cap $c step", round(j["ms_per_step"],3), {k: round(v,3) for k,v in sorted(p.items(), key=lambda kv:-kv[1])})
PY
done
