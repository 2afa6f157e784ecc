#!/bin/bash
# usage: tools/slab_rank_trace.sh "P:r" [size]  — rocprofv3 kernel trace of tools/slab_rank_bench.py for one (ranks, rank) case; per-kernel ms per step of the
# slab rank (the part of the trace after the first RCCL kernel) beside the single-domain reference of the same process (the part before it)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/srt
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/slab_rank_bench.py ${2:-512} "$1" > $OUT/out.json 2> $OUT/err.log
python3 - "$1" <<PY
import csv,glob,re,collections,json,sys
f=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
first=next(i for i,r in enumerate(rows) if 'nccl' in r['Kernel_Name'].lower())
def short(n):
    n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'^void ','',n); return re.sub(r'\(.*','',n)[:44]
def agg(rs):
    d=collections.defaultdict(lambda:[0,0.0])
    for r in rs:
        k=short(r['Kernel_Name']); d[k][0]+=1; d[k][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    return d
ref=agg(rows[:first]); sl=agg(rows[first:])
t=open("$OUT/out.json").read(); j=json.loads(t[t.index("{\n"):])
print('single domain ms/step', round(j['single_domain']['ms_per_step'],3), '| slab case', [(c['ranks'],c['rank'],round(c['ms_per_step'],3)) for c in j['cases']])
# reference: 13 steps (3 warm + 10) + setup; slab: steps = warm 3 + timed
nref=13.0; nsl=None
for c in j['cases']: nsl=3+c.get('steps',10)
print(f"{'kernel':44s} {'ref ms/step':>11s} {'slab ms/step':>12s} {'slab calls/step':>15s}")
for k,v in sorted(sl.items(), key=lambda kv:-kv[1][1])[:22]:
    print(f"{k:44s} {ref.get(k,[0,0])[1]/nref:11.3f} {v[1]/nsl:12.3f} {v[0]/nsl:15.1f}")
print('sum slab kernels ms/step', sum(v[1] for v in sl.values())/nsl, ' ref', sum(v[1] for v in ref.values())/nref)
PY
