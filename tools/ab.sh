for v in default w6 w7 default w6 w7; do
if [ $v != default ]; then export WLHIP_LIB=$PWD/tools/var/libwlhip_$v.so; else unset WLHIP_LIB; fi
WL_OPT_convm=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$v.log").read().strip().splitlines()[-1])
p=j["phases_ms_per_step"]
print("$v", round(j["ms_per_step"],2), "conv", round(p["conv_diff"],3), "smooth", round(p["smooth"],3))
PY
done
