#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/qs_sphere
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/sphere_probe.py > $OUT/out.txt 2> $OUT/err.log
tail -1 $OUT/out.txt
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/1e6/13)
for r in rows[:18]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  /step {float(r['TotalDurationNs'])/1e6/13:6.3f} ms")
PY
