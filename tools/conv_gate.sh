run() { name=$1; shift; env "$@" python bench.py --steps 150 --warmup 20 --size ${SIZE} --no-cpu-baseline --no-phases > gpurun_out/cg.json 2> gpurun_out/cg.err; python - <<PY
import json
j=json.loads(open("gpurun_out/cg.json").read().strip().splitlines()[-1])
print("${SIZE}", "$name", round(j["ms_per_step"],4))
PY
}
for SIZE in 128 160 192; do
run base A=1
run tile WL_OPT_convt_min=0
run base A=1
run tile WL_OPT_convt_min=0
done
