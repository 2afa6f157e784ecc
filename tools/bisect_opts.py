#!/usr/bin/env python3
"""which implementation switch makes a shape deviate from the plain path?  usage: tools/bisect_opts.py nx ny nz [nu] [Ux]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waterlily_jl_amd as w
dims = tuple(int(v) for v in sys.argv[1:4])
nu = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
U = (float(sys.argv[5]) if len(sys.argv) > 5 else 0.3, 0.0, 0.0)
rng = np.random.default_rng(5)
u0 = np.asfortranarray(rng.uniform(-0.5, 0.5, size=tuple(n + 2 for n in dims) + (3,)).astype(np.float32))
PLAIN = {"fused_smoother": 0, "fuse_p": 0, "constl": 0, "fuse_cfl": 0, "store_f": 1, "tail": 0, "jacobi_march": 0}
def run(opts, steps=2):
    s = w.FusedSimulation(dims, U, dims[0], U=1, nu=nu, u0=u0)
    for k, v in opts.items(): s.set_option(k, v)
    out = []
    for _ in range(steps):
        s.mom_step_(); out.append((s.field("u"), s.field("p"), list(s.pois_n), float(s.dt[-1])))
    return out
ref = run(PLAIN)
for name, val in (("fused_smoother", 1), ("fuse_p", 1), ("constl", 1), ("fuse_cfl", 1), ("store_f", 0), ("tail", 1), ("jacobi_march", 1)):
    o = dict(PLAIN); o[name] = val
    if name == "fuse_cfl": o["fuse_p"] = 1
    if name == "jacobi_march": o["constl"] = 1
    r = run(o)
    for st in range(len(ref)):
        du = float(np.abs(r[st][0] - ref[st][0]).max()); dp = float(np.abs(r[st][1] - ref[st][1]).max())
        print(f"{name}={val} step {st}: du={du:.2e} dp={dp:.2e} n={r[st][2]} vs {ref[st][2]} dt={r[st][3]:.6f}/{ref[st][3]:.6f}")
r = run({})
for st in range(len(ref)):
    du = float(np.abs(r[st][0] - ref[st][0]).max()); dp = float(np.abs(r[st][1] - ref[st][1]).max())
    print(f"ALL FAST step {st}: du={du:.2e} dp={dp:.2e} n={r[st][2]} vs {ref[st][2]} dt={r[st][3]:.6f}/{ref[st][3]:.6f}")
# pairs of options
import itertools
names = [("fused_smoother", 1), ("fuse_p", 1), ("constl", 1), ("store_f", 0), ("tail", 1)]
for (a, va), (b, vb) in itertools.combinations(names, 2):
    o = dict(PLAIN); o[a] = va; o[b] = vb
    r = run(o, 1)
    du = float(np.abs(r[0][0] - ref[0][0]).max())
    if du != 0.0: print(f"{a}+{b}: du={du:.2e} n={r[0][2]} vs {ref[0][2]}")
