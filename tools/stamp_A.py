#!/usr/bin/env python3
"""Diagnostic (WL_STAMP build of wl_fused2.hip): where a fast step of smoother kernel A spends its cycles — per wave, s_memtime stamps around
[rotate + loads issued + prolongation stage + wait for operands] | [barrier] | [LDS reads + two sweeps + LDS writes] | [stores issued].
usage (GPU box): WLHIP_LIB=$PWD/waterlily.jl_amd/libwlhip_stamp.so python tools/stamp_A.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waterlily_jl_amd as w
from waterlily_jl_amd._lib import check
lib = w.lib()
check(lib.wl_init(0))
raw = C.CDLL(os.environ["WLHIP_LIB"])
N = 512
sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
for _ in range(3):
    sim.mom_step_()
sim.sync()
out = (C.c_ulonglong * 5)()
raw.wl_debug_stamps(out, 1)
for _ in range(6):
    sim.mom_step_()
sim.sync()
raw.wl_debug_stamps(out, 0)
pre, bar, swp, sto, n = [int(v) for v in out]
tot = pre + bar + swp + sto
print(f"steps (wave-steps) {n}; cycles per wave-step: total {tot / n:.0f} = pre-barrier {pre / n:.0f} ({100 * pre / tot:.0f} %) + barrier wait {bar / n:.0f} ({100 * bar / tot:.0f} %) "
      f"+ sweeps {swp / n:.0f} ({100 * swp / tot:.0f} %) + stores {sto / n:.0f} ({100 * sto / tot:.0f} %)   [s_memtime ticks; each stamp costs ≈40]")
