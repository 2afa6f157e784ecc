#!/usr/bin/env python3
"""Randomised differential test of the multigrid fast paths (minutes of GPU time, not part of the test-suite): random box shapes with a
NoBody coefficient field; Vcycle! + smooth! through the default kernels (pair smoother with fast steps, x increment deferred to kernel B,
chunk model, LDS-resident coarse tail) and through one launch per pass — r, x, ϵ of every level must be identical bit for bit.
usage: tools/stress_vcycle.py [cases] [seed]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waterlily_jl_amd as w

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = w.lib()
bad = 0


def F(shape):
    return np.zeros(shape, dtype=np.float32, order="F")


for c in range(cases):
    def dim():
        a = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 17]))
        k = int(rng.integers(2, 8))
        return a * 2**k
    dims = tuple(min(dim(), 320) for _ in range(3))
    if np.prod(dims) > 24e6 or min(dims) < 8:
        continue
    N = tuple(n + 2 for n in dims)
    cc = [float(v) for v in rng.choice([1.0, 0.5, 2.0, 0.75], size=3)] if rng.random() < 0.3 else [1.0, 1.0, 1.0]
    L = F(N + (3,))
    for a in range(3):
        L[..., a] = cc[a]
        sl = [slice(None)] * 3
        sl[a] = slice(0, 2); L[tuple(sl) + (a,)] = 0      # BC!(L,0): faces at Julia index 1, 2 and N are wall faces
        sl[a] = slice(N[a] - 1, N[a]); L[tuple(sl) + (a,)] = 0
    x0, z, r0 = F(N), F(N), F(N)
    r0[1:-1, 1:-1, 1:-1] = rng.uniform(-1, 1, size=dims).astype(np.float32)
    x0[1:-1, 1:-1, 1:-1] = rng.uniform(-1, 1, size=dims).astype(np.float32)
    om = float(rng.choice([1.0, 0.9, 0.73]))
    res = {}
    try:
        for tag, fused in (("fast", True), ("passes", False)):
            xg, Lg, zg = w.to_device(x0), w.to_device(L), w.to_device(z)
            pg = w.MultiLevelPoisson(xg, Lg, zg)
            pg.set_fused(fused, fused)
            w._lib.check(lib.wl_h2d(lib.wl_mg_level_field(pg._h, 0, b"r"), r0.ctypes.data_as(C.c_void_p), r0.nbytes, w.core.stream()))
            pg.Vcycle_(0, om); pg.smooth_(0, 4, om)
            nl = pg.nlevels if hasattr(pg, "nlevels") else len(pg.levels)
            res[tag] = [(pg.levels[l].r, pg.levels[l].x) for l in range(nl)] + [w.to_host(xg)]
            kinds = [int(lib.wl_mg_smoother_kind(pg._h, l)) for l in range(nl)]
            pg.set_fused(True, True)
            del pg
    except AssertionError as e:
        print(f"case {c}: dims={dims} skipped ({e})", flush=True)
        continue
    ok = True
    for l in range(len(res["fast"]) - 1):
        for k in range(2):
            ok = ok and np.array_equal(res["fast"][l][k], res["passes"][l][k])
    ok = ok and np.array_equal(res["fast"][-1], res["passes"][-1])
    print(f"case {c}: dims={dims} c={cc} ω={om} levels={len(res['fast']) - 1} {'bitwise' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
