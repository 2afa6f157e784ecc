#!/usr/bin/env python3
"""periodic directions on z-slabs against the single domain (P gloo ranks on the one GPU): tests/slab_worker.py gpu_per"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_slab_cpu import run_ranks
cases = [(2, "64x32x64", "12"), (2, "64x64x64", "123"), (2, "64x32x64", "3"), (4, "64x64x128", "123"), (3, "48x32x96", "13"), (2, "128x64x64", "3")]
bad = 0
for n, dims, per in cases:
    try:
        out = run_ranks(n, "gpu_per", dims, "3", per, timeout=600)
        print(n, dims, per, "OK", [l for l in out.splitlines() if l.startswith("step")][-1], flush=True)
    except AssertionError as e:
        bad += 1
        msg = str(e)
        print(n, dims, per, "FAILED", "\n".join([l for l in msg.splitlines() if "step" in l or "Error" in l or "error" in l or "assert" in l][-12:]), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
