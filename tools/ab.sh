for v in 0 1; do
WL_OPT_store_f=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$v.log").read().strip().splitlines()[-1])
p=j["phases_ms_per_step"]
print("store_f=$v", round(j["ms_per_step"],2), "conv", round(p["conv_diff"],3), "resid", round(p["residual"],3))
PY
done
