#!/bin/bash
# usage: tools/pmc.sh "<counter list>" [kernel-substring]   — one rocprofv3 --pmc pass over a short 512^3 run, per-kernel averages
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --size ${3:-512} --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
tail -3 $OUT/err.log
python3 - "$2" <<'PY'
import csv,glob,sys,collections
pat=sys.argv[1]
fs=glob.glob(sys.argv[0] and "gpurun_out/pmc/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"]
    if pat and pat not in k: continue
    key=k[:60]+" g="+r["Grid_Size"]
    a=acc[key][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,v in sorted(acc.items(), key=lambda kv:-max(x[1] for x in kv[1].values()))[:14]:
    print(k, {c:round(a[0]/a[1],2) for c,a in v.items()}, "n=",max(a[1] for a in v.values()))
PY
