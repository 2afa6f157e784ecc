#!/usr/bin/env python3
"""print ms/step and the phase table of gpurun_out/bench_*.json side by side (A/B runs of bench.py with WL_OPT_<option>=…)"""
import glob
import json
import sys

files = sys.argv[1:] or sorted(glob.glob("gpurun_out/bench_[a-z].json"))
for f in files:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ph = d.get("phases_ms_per_step", {})
    print(f, f"{d['ms_per_step']:.3f} ms/step  n={d['config'].get('mean_pois_n')}", " ".join(f"{k}={v:.3f}" for k, v in ph.items() if v))
