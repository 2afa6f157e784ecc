#!/bin/bash
# usage: tools/kstat.sh TAG PATTERN [SIZE]  — rocprofv3 --kernel-trace --stats of a short bench run (environment as given); prints the kernels matching PATTERN (regex)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ks_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --size ${3:-512} --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
python3 - "$1" "$2" <<PY
import csv,glob,sys,re,json
tag,pat=sys.argv[1],sys.argv[2]
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
try: ms=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])["ms_per_step"]
except Exception: ms=-1
print(tag,'step',round(ms,3),'kernel ms/step', round(tot/1e6/10,3))
for r in rows:
    if re.search(pat, r['Name']): print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
