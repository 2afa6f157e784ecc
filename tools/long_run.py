#!/usr/bin/env python3
"""Long-horizon sanity run of the default (fast) path: 256^3 wall-bounded TGV for many steps — solver iteration counts as the
flow develops, kinetic energy decay, finiteness.  usage: tools/long_run.py [N] [steps]"""
import collections
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waterlily_jl_amd as w

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
t0 = time.perf_counter()
ke = []
for k in range(steps):
    sim.mom_step_()
    if k % 50 == 49 or k == steps - 1:
        u = sim.field("u")
        assert np.isfinite(u).all()
        ke.append(float(np.sum(u[1:-1, 1:-1, 1:-1].astype(np.float64) ** 2)) / N**3)
        print(f"step {k + 1}: tU/L={sim.sim_time():.3f} dt={float(sim.dt[-1]):.4f} KE={ke[-1]:.6f} pois.n last 10={sim.pois_n[-10:]}", flush=True)
el = time.perf_counter() - t0
hist = collections.Counter(sim.pois_n)
print("pois.n histogram:", dict(sorted(hist.items())), f"wall {el:.1f}s ({el / steps * 1e3:.2f} ms/step incl. host reads)")
assert all(b <= a * (1 + 1e-6) for a, b in zip(ke, ke[1:])), ke
