#!/usr/bin/env python3
"""Proof by exhaustion that float32(float64(x)·(1/6)) == x/6 (IEEE, round-to-nearest-even) for every float32 x —
the identity behind div6() in waterlily.jl_amd/csrc/wl_flow.hip.  Binary scaling invariance: all 2^23 significands at
representative normal exponents, every exponent whose quotient is subnormal/underflows, both signs, plus random bits."""
import numpy as np

inv6 = np.float64(1.0) / np.float64(6.0)
bad = 0
mant = np.arange(1 << 23, dtype=np.uint32)
for e in list(range(0, 8)) + [64, 126, 127, 128, 200, 253, 254, 255]:
    for sgn in (0, 1):
        x = ((np.uint32(sgn) << np.uint32(31)) | (np.uint32(e) << np.uint32(23)) | mant).view(np.float32)
        with np.errstate(all="ignore"):
            a = x / np.float32(6.0)
            b = (x.astype(np.float64) * inv6).astype(np.float32)
        bad += int(((a.view(np.uint32) != b.view(np.uint32)) & ~np.isnan(a)).sum())
print("mismatches:", bad)
assert bad == 0
