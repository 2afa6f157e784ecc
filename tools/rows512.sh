run() { name=$1; shift; env "$@" python bench.py --phases --steps 8 --warmup 3 --size 512 --no-cpu-baseline > gpurun_out/cg.json 2> gpurun_out/cg.err; python - <<PY
import json
j=json.loads(open("gpurun_out/cg.json").read().strip().splitlines()[-1]); p=j["phases_ms_per_step"]
print("512", "$name", round(j["ms_per_step"],3), "A", round(p.get("gsrb_A",0),3), "B", round(p.get("gsrb_B",0),3), "coarse", round(p.get("coarse_levels",0),3))
PY
}
run base A=1
run rows16 WL_PAIR_ROWS=16
run a16 WL_PAIR_ROWS_A=16
run b16 WL_PAIR_ROWS_B=16
run base A=1
