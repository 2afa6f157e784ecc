for t in 1536 -512 800 3072 6144; do
WL_PAIR_WGS=$t python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$t.log 2>&1
python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$t.log").read().strip().splitlines()[-1])
p=j["phases_ms_per_step"]
print("$t", round(j["ms_per_step"],2), "A", round(p["gsrb_A"],3), "B", round(p["gsrb_B"],3), "smooth", round(p["smooth"],3))
PY
done
