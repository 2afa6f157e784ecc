#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1) --kernel-trace --stats of the default bench command  -> per-kernel time
#   2) separate --pmc passes (FETCH_SIZE ; WRITE_SIZE) of a short run -> HBM bytes per launch of the dominant kernel
# Outputs land in gpurun_out/prof_<tag>/ ; summarise with profiles/summarise.py and commit the summaries.
set -e
TAG=${1:-r01}
SIZE=${2:-512}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --size $SIZE --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --size $SIZE --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --size $SIZE --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
python3 profiles/summarise.py $OUT $TAG $SIZE
