#!/usr/bin/env python3
"""sphere (closed-form sdf) N³ Re=3700: ms per mom_step! (BASELINE configs[3] at N=256) — for rocprofv3 kernel stats of the body path"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import waterlily_jl_amd as w
from waterlily_jl_amd._lib import check

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = w.lib()
check(lib.wl_init(0))
R, c = N / 8, (N / 2 - 1,) * 3
sim = w.FusedSimulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=2 * R / 3700, has_body=True)
for key, val in os.environ.items():      # A/B switches: WL_OPT_<option of wl_sim_set_option>=value (before measure!: some act at update! time)
    if key.startswith("WL_OPT_"):
        sim.set_option(key[7:], int(val))
sim.measure_sphere_(c, R, 1.0)
for _ in range(5):
    sim.mom_step_()
sim.sync()
t0 = time.perf_counter()
for _ in range(steps):
    sim.mom_step_()
sim.sync()
print(f"sphere {N}^3: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step, pois.n mean {sum(sim.pois_n) / len(sim.pois_n):.2f}")
