#!/usr/bin/env python3
"""conv_diff! leaf on an emulated z-periodic slab (one process, no transport): the slab arrays are cut from the single-domain periodic field with the
wrapped neighbours' planes as ghost planes; the owned planes of r must equal the single-domain result."""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waterlily_jl_amd as w
from waterlily_jl_amd._lib import check, wl_grid
from waterlily_jl_amd.core import perdir_mask, ptr, stream
lib = w.lib()
dims = (24, 16, 32); N = tuple(n + 2 for n in dims); P = 2; gh = 5
rng = np.random.default_rng(3)
u = np.asfortranarray(rng.uniform(-1, 1, size=N + (3,)).astype(np.float32))
per = (1, 2, 3)
ug = w.to_device(u)
w.BC_(ug, (0, 0, 0), perdir=per)                 # periodic ghost cells
u = w.to_host(ug)
r = w.to_device(np.zeros(N + (3,), np.float32, order="F")); Phi = w.to_device(np.zeros(N, np.float32, order="F"))
w.conv_diff_(r, ug, Phi, nu=0.05, perdir=per)
rref = w.to_host(r)
nloc = dims[2] // P
bad = 0
for rank in range(P):
    # local planes: global 0-based plane index of local plane l is gk + l, wrapped periodically over the interior planes 1..N-2
    gk = 1 + rank * nloc - gh
    nzl = nloc + 2 * gh
    idx = [((gk + l - 1) % dims[2]) + 1 for l in range(nzl)]
    ul = np.asfortranarray(u[:, :, idx, :])
    g = wl_grid(3, N[0], N[1], nzl, gh, gh + nloc, gk, N[2])
    uld = w.to_device(ul); rl = w.to_device(np.zeros_like(ul)); pl = w.to_device(np.zeros(ul.shape[:3], np.float32, order="F"))
    check(lib.wl_conv_diff(ptr(rl), ptr(uld), ptr(pl), C.byref(g), 0.05, perdir_mask(per), 0, stream()))
    rl = w.to_host(rl)
    own = rl[1:-1, 1:-1, gh:gh + nloc, :]
    ref = rref[1:-1, 1:-1, 1 + rank * nloc:1 + (rank + 1) * nloc, :]
    d = np.abs(own - ref).max()
    print("rank", rank, "max|dr| on the owned interior:", d, "equal" if np.array_equal(own, ref) else "DIFFERENT")
    if d > 0:
        kk = np.unravel_index(np.argmax(np.abs(own - ref)), own.shape); print("  worst at", kk)
    bad += d > 0
sys.exit(1 if bad else 0)
