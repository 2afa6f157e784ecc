run() { name=$1; shift; env "$@" python bench.py --steps 150 --warmup 20 --size ${SIZE} --no-cpu-baseline --no-phases > gpurun_out/cg.json 2> gpurun_out/cg.err; python - <<PY
import json
j=json.loads(open("gpurun_out/cg.json").read().strip().splitlines()[-1])
print("${SIZE}", "$name", round(j["ms_per_step"],4))
PY
}
for SIZE in 128 256; do
run base A=1
run rows16 WL_PAIR_ROWS=16
run rows32 WL_PAIR_ROWS=32
run base A=1
done
