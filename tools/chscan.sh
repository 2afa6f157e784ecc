run() { python bench.py --phases --steps 8 --warmup 3 --size 512 --no-cpu-baseline > gpurun_out/ch.json 2> gpurun_out/ch.err; python - <<PY
import json
j=json.loads(open("gpurun_out/ch.json").read().strip().splitlines()[-1]); p=j["phases_ms_per_step"]
print("$1", "step", round(j["ms_per_step"],3), "A", round(p.get("gsrb_A",0),3), "B", round(p.get("gsrb_B",0),3))
PY
}
run base
WL_PAIR_CH_A=4 WL_PAIR_CH_B=5 run a4_b5
WL_PAIR_CH_A=10 WL_PAIR_CH_B=10 run a10_b10
WL_PAIR_CH_A=13 WL_PAIR_CH_B=14 run a13_b14
WL_PAIR_CH_A=16 WL_PAIR_CH_B=6 run a16_b6
WL_PAIR_CH_A=7 WL_PAIR_CH_B=7 run a7_b7
WL_PAIR_CH_A=22 WL_PAIR_CH_B=21 run a22_b21
run base
