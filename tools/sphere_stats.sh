export TMPDIR=/tmp
OUT=$PWD/gpurun_out/qs_sphere
rm -rf $OUT; mkdir -p $OUT
python3 tools/sphere_bench.py 256 20 > $OUT/plain.txt 2>&1; cat $OUT/plain.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/sphere_bench.py 256 20 > $OUT/out.txt 2> $OUT/err.log
python3 - <<PY
import csv,glob,re
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/1e6/25)
for r in rows[:24]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n)
    print(f"{n[:60]:60s} calls/step {int(r['Calls'])/25:6.2f} avg {float(r['AverageNs'])/1e3:9.1f} us  /step {float(r['TotalDurationNs'])/1e6/25:7.3f} ms")
PY
