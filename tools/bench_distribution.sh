#!/bin/bash
# usage: tools/bench_distribution.sh N  — N default bench runs alternating with N runs without placement trials (each its own process: its own allocations);
# one line per run: ms/step, G cells*steps/s, smoother pair ms, roofline frac, placement scores
for i in $(seq 1 ${1:-6}); do
  for t in default 1; do
    if [ "$t" = default ]; then unset WL_PLACEMENT_TRIALS; else export WL_PLACEMENT_TRIALS=1; fi
    python bench.py --no-cpu-baseline > gpurun_out/bd.json 2> gpurun_out/bd.err
    python3 - "$t" <<'PY'
import json,sys
j=json.loads(open("gpurun_out/bd.json").read().strip().splitlines()[-1]); r=j["roofline"]
print(f"trials={sys.argv[1]:7s} {j['ms_per_step']:7.3f} ms/step  {j['value']/1e9:6.2f} G  pair {r['avg_launch_ms']:.4f} ms  frac {r['frac']:.4f}  A {r['kernels']['A']['avg_ms']:.4f} B {r['kernels']['B']['avg_ms']:.4f}  scores {j['config']['placement_trial_ms']}", flush=True)
PY
  done
done
