#!/bin/bash
# quick per-kernel time table: rocprofv3 --kernel-trace --stats of a short bench run (run through gpurun from the repo root)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/qs
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --size ${1:-512} --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/1e6/13)
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  /step {float(r['TotalDurationNs'])/1e6/13:7.3f} ms")
PY
