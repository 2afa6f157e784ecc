for v in 1 0 1 0; do
WL_OPT_fuse_cfl=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$v.log").read().strip().splitlines()[-1])
p=j["phases_ms_per_step"]
print("fuse_cfl=$v", round(j["ms_per_step"],2), "conv", round(p["conv_diff"],3), "smooth", round(p["smooth"],3))
PY
done
