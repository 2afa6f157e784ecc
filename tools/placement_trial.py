#!/usr/bin/env python3
"""Does the physical placement of the simulation's allocations change the step time?  One process, several simulations of the same 512^3 TGV
created one after the other (the earlier ones are kept alive, so each lands in new memory): per simulation the ms/step and the per-launch
times of the finest-level smoother kernels A / B (HIP events).  usage (GPU box): python tools/placement_trial.py [n] [size]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import waterlily_jl_amd as w
from waterlily_jl_amd._lib import check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = w.lib()
check(lib.wl_init(0))
hold = []
for i in range(n):
    sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
    import ctypes as C
    sc = (C.c_double * 8)(); nsc = lib.wl_placement_scores(sc, 8)
    scores = [round(float(sc[k]), 3) for k in range(min(nsc, 8))]
    for _ in range(3):
        sim.mom_step_()
    sim.sync()
    check(lib.wl_prof_enable(1))
    t0 = time.perf_counter()
    for _ in range(8):
        sim.mom_step_()
    sim.sync()
    ms = (time.perf_counter() - t0) / 8 * 1e3
    prof = bench.read_prof(lib)
    check(lib.wl_prof_enable(0))
    row = {"sim": i, "placement_scores": scores, "ms_per_step": round(ms, 3), "A_ms": round(prof["gsrb_A"]["avg_ms"], 4), "B_ms": round(prof["gsrb_B"]["avg_ms"], 4),
           "conv_ms_per_step": round(prof["conv_diff"]["total_ms"] / 8, 3), "resid_ms_per_step": round(prof["residual"]["total_ms"] / 8, 3),
           "coarse_ms_per_step": round(prof["coarse_levels"]["total_ms"] / 8, 3)}
    print(json.dumps(row), flush=True)
    hold.append(sim)
