#!/usr/bin/env python3
"""extra slab shapes: P gloo ranks on the one GPU against the single-domain run (tests/slab_worker.py gpu_sim / gpu_exit)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_slab_cpu import run_ranks
cases = [(2, "gpu_sim", "72x40x96"), (4, "gpu_sim", "80x64x128"), (3, "gpu_sim", "66x34x144"), (2, "gpu_sim", "128x64x160"), (4, "gpu_sim", "130x66x64"),
         (2, "gpu_exit", "96x48x64"), (3, "gpu_exit", "64x64x96")]
bad = 0
for n, mode, dims in cases:
    try:
        out = run_ranks(n, mode, dims, "3", timeout=600)
        ok = all(f"rank {r}: {mode} ok" in out for r in range(n))
        last = [l for l in out.splitlines() if l.startswith("step 2")]
        print(n, mode, dims, "ok" if ok else "FAILED", last[-1] if last else "", flush=True)
    except Exception as e:
        ok = False
        print(n, mode, dims, "EXCEPTION", repr(e)[:300], flush=True)
    bad += not ok
print("failures:", bad)
sys.exit(1 if bad else 0)
