#!/usr/bin/env python3
"""Randomised HIP-vs-oracle campaign on small boxes (2-D and 3-D; periodic directions, convective exit, immersed sphere/circle,
QUICK/vanLeer/CDS): three mom_step! each, compared with the tolerances of the test-suite.  usage: tools/stress_oracle.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import waterlily_jl_amd as w
from oracle import oracle as orc

orc.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for c in range(cases):
    D = int(rng.choice([2, 3, 3]))
    dims = tuple(int(rng.choice([3, 5, 4, 6, 7])) * 2 ** int(rng.integers(2, 4 if D == 3 else 6)) for _ in range(D))
    if np.prod(dims) > 300000:
        continue
    body = bool(rng.integers(0, 2))
    exitbc = bool(rng.integers(0, 2)) and True
    perdir = tuple(int(d) for d in range(2, D + 1) if rng.random() < 0.3) if not body else ()
    scheme = int(rng.choice([w.core.QUICK, w.core.VANLEER, w.core.CDS]))
    U = (float(rng.uniform(0.3, 1.0)),) + (0.0,) * (D - 1)
    nu = float(rng.choice([0.0, 0.01, 0.05]))
    Ng = tuple(n + 2 for n in dims)
    u0 = np.asfortranarray((np.float32(U[0]) * np.eye(1, D, 0, dtype=np.float32)[0] + rng.uniform(-0.2, 0.2, size=Ng + (D,))).astype(np.float32))
    R = min(dims) / 6.0
    ctr = tuple(n / 2 - 1 + 0.37 * k for k, n in enumerate(dims))
    try:
        so = orc.Simulation(dims, U, dims[0], U=1, nu=nu, perdir=perdir, exitBC=exitbc, scheme=scheme, body=("sphere", ctr, R) if body else None, T=np.float32)
        sg = w.FusedSimulation(dims, U, dims[0], U=1, nu=nu, perdir=perdir, exitBC=exitbc, lam=scheme, has_body=body, u0=u0)
    except AssertionError as e:
        print(f"case {c}: dims={dims} skipped ({str(e)[:50]})", flush=True)
        continue
    orc.BC(u0, U, exitbc, perdir)
    so.field("u")[...] = u0; so.field("u0")[...] = u0
    sg.set_field("u", u0); sg.set_field("u0", u0)
    if body:
        sg.measure_sphere_(ctr, R, 1.0)
        sg.set_field("mu0", so.field("mu0")); sg.set_field("mu1", so.field("mu1")); sg.update_()
    ok, worst = True, 0.0
    for st in range(3):
        so.step(remeasure=False); sg.mom_step_()
        du = float(np.abs(sg.field("u") - so.u).max()); worst = max(worst, du)
        ok = ok and sg.pois_n[-2:] == so.pois_n[-2:] and du < 1e-4 and abs(float(sg.dt[-1]) - float(so.dt[-1])) <= 1e-5 * float(so.dt[-1])
    print(f"case {c}: D={D} dims={dims} body={body} exit={exitbc} per={perdir} scheme={scheme} nu={nu} n={sg.pois_n[-2:]} max|du|={worst:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
