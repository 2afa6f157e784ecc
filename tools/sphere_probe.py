#!/usr/bin/env python3
"""sphere 256^3 Re=3700 (BASELINE configs[3]) — a few steps, for rocprofv3 --kernel-trace --stats"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import waterlily_jl_amd as w

N = int(os.environ.get("WL_N", "256")); R = N / 8.0
c = (N / 2 - 1,) * 3
sim = w.FusedSimulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=2 * R / 3700, has_body=True)
for k, v in os.environ.items():          # WL_OPT_<option>=<int>: A/B switches, as in bench.py
    if k.startswith("WL_OPT_"):
        sim.set_option(k[7:], int(v))
sim.measure_sphere_(c, R, 1.0)
print("smoother kinds", sim.smoother_kinds())
for _ in range(3):
    sim.mom_step_()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    sim.mom_step_()
torch.cuda.synchronize()
print(f"sphere {N}^3 ms/step", (time.perf_counter() - t0) / 10 * 1e3, "pois.n", sim.pois_n[-6:])
