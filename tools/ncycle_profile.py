#!/usr/bin/env python3
"""What one more V-cycle costs (VERDICT r02 "next" #7): the benchmark's TGV takes the single mandatory V-cycle per solve (pois.n = 1), so the
bench line shows the solver's floor.  Here the same 512^3 box starts from the TGV plus a random solenoidal-free perturbation: the first solves
need several V-cycles, fewer as the step proceeds.  Every step is timed on its own (device-synchronised) next to the number of V-cycles of its
two solves; a least-squares line ms = a + b·(n1+n2) gives the cost b of one V-cycle (Vcycle! + smooth! + norms, src/MultiLevelPoisson.jl:108-128).
usage (GPU box): python tools/ncycle_profile.py [size] [steps] > profiles/r03_vcycle_cost_512.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waterlily_jl_amd as w
from waterlily_jl_amd._lib import check

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
AMP = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
lib = w.lib()
check(lib.wl_init(0))
sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
u = sim.field("u")
rng = np.random.default_rng(3)
u[1:-1, 1:-1, 1:-1, :] += rng.uniform(-AMP, AMP, size=(N, N, N, 3)).astype(np.float32)
sim.set_field("u", u); sim.set_field("u0", u)
del u
rows = []
for s in range(steps):
    n0 = len(sim.pois_n)
    sim.sync(); t0 = time.perf_counter()
    sim.mom_step_()
    sim.sync(); ms = (time.perf_counter() - t0) * 1e3
    n = sim.pois_n[n0:]
    rows.append({"step": s, "ms": ms, "n": n, "dt": float(sim.dt[-1])})
    print(rows[-1], file=sys.stderr, flush=True)
x = np.array([sum(r["n"]) for r in rows[1:]], dtype=float)     # (the first step carries first-launch costs)
y = np.array([r["ms"] for r in rows[1:]])
A = np.vstack([np.ones_like(x), x]).T
(a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "size": N, "steps": rows,
       "fit": {"ms_per_step_at_n_total_0": float(a), "ms_per_vcycle_iteration": float(b), "ms_per_step_at_n1_n1": float(a + 2 * b),
               "max_abs_residual_ms": float(np.abs(A @ np.array([a, b]) - y).max())}}
print(json.dumps(out, indent=1))
