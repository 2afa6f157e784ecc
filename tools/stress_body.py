#!/usr/bin/env python3
"""Randomised differential test for body flows (not part of the test-suite): a sphere of random size and position in random boxes, stepped through the
default fast kernels (z-split smoother with pair kernels on the far planes, tiled conv_diff! on the body-free plane ranges, far/near BDIM! masks) and
through the general kernels — u, p, pois.n, Δt must be identical bit for bit; the body moves between steps.
usage: tools/stress_body.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waterlily_jl_amd as w

PLAIN = {"zsplit": 0, "farmask": 0, "hybrid": 0, "constl": 0, "body_tile": 0, "fused_smoother": 0, "tail": 0}
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    dims = tuple(int(rng.choice([48, 64, 96, 128, 160])) for _ in range(3))
    if np.prod(dims) > 3e6:
        dims = (dims[0], 64, 64)
    R = float(rng.uniform(3.0, min(dims) / 6))
    def centre():
        return tuple(float(rng.uniform(R + 3, n - R - 3)) for n in dims)
    c0, c1 = centre(), centre()
    exitBC = bool(rng.random() < 0.3)
    res = {}
    for tag, opts in (("fast", {"zsplit": 2, "convt_min": 0}), ("plain", PLAIN)):
        s = w.FusedSimulation(dims, (1, 0, 0), 2 * R, U=1, nu=2 * R / 250, has_body=True, exitBC=exitBC)
        for k, v in opts.items():
            s.set_option(k, v)
        s.measure_sphere_(c0, R, 1.0)
        for _ in range(2):
            s.mom_step_()
        s.measure_sphere_(c1, R, 1.0)
        s.mom_step_()
        res[tag] = (s.field("u"), s.field("p"), s.pois_n, [float(d) for d in s.dt], s.smoother_kinds() if tag == "fast" else None)
        del s
    bit = res["fast"][2] == res["plain"][2] and res["fast"][3] == res["plain"][3] and np.array_equal(res["fast"][0], res["plain"][0]) and np.array_equal(res["fast"][1], res["plain"][1])
    du = float(np.abs(res["fast"][0] - res["plain"][0]).max()); dp = float(np.abs(res["fast"][1] - res["plain"][1]).max())
    ok = bit or (du < 2e-4 and dp < 2e-3 * max(1.0, float(np.abs(res["plain"][1]).max())) and all(abs(a - b) <= 1 for a, b in zip(res["fast"][2], res["plain"][2])))
    print(f"case {c}: dims={dims} R={R:.1f} exit={exitBC} kinds={res['fast'][4]} n={res['fast'][2]}/{res['plain'][2]} du={du:.1e} dp={dp:.1e} {'bitwise' if bit else ('ok' if ok else 'MISMATCH')}", flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
