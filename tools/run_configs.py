#!/usr/bin/env python3
"""Run the BASELINE.json configs that fit one GPU and print one JSON line each:
 [1] TGV 128³, [2] TGV 256³ (smoother GB/s), [3] sphere 256³ Re=3700 (BDIM! + pressure_force)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import waterlily_jl_amd as w
from waterlily_jl_amd._lib import check

lib = w.lib()
check(lib.wl_init(0))


def prof(names):
    out = {}
    for slot, nm in names.items():
        cnt, tot = C.c_int(), C.c_double()
        check(lib.wl_prof_read(slot, C.byref(cnt), C.byref(tot)))
        out[nm] = (tot.value / cnt.value) if cnt.value else None
    return out


def tgv(N, steps=50, warm=10):
    sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
    for _ in range(warm):
        sim.mom_step_()
    sim.sync()
    check(lib.wl_prof_enable(1))
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.mom_step_()
    sim.sync()
    el = time.perf_counter() - t0
    pr = prof({1: "smooth_ms", 9: "gsrb_A_ms", 10: "gsrb_B_ms", 3: "conv_ms"})
    check(lib.wl_prof_enable(0))
    sm = pr["smooth_ms"]
    print(json.dumps({"config": f"TGV {N}^3", "ms_per_step": el / steps * 1e3, "cells_steps_per_s": N**3 * steps / el, "mean_pois_n": float(np.mean(sim.pois_n[2 * warm:])),
                      "smooth_ms": sm, "smooth_op_GBs": 40.0 * N**3 / (sm * 1e-3) / 1e9 if sm else None, **pr}), flush=True)


def sphere(N=256, steps=40):
    R = N / 8
    c = (N / 2 - 1,) * 3
    sim = w.FusedSimulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=2 * R / 3700, has_body=True)
    sim.measure_sphere_(c, R, 1.0)
    t0 = time.perf_counter()
    cd = []
    for k in range(steps):
        sim.mom_step_()
        if k % 10 == 9:
            f = sim.pressure_force_sphere(c, R)
            cd.append(float(f[0] / (0.5 * np.pi * R**2)))
    sim.sync()
    el = time.perf_counter() - t0
    u = sim.field("u")
    print(json.dumps({"config": f"sphere {N}^3 Re=3700", "ms_per_step": el / steps * 1e3, "cells_steps_per_s": N**3 * steps / el, "pois_n_max": max(sim.pois_n), "pois_n_mean": float(np.mean(sim.pois_n)),
                      "pressure_drag_coefficient_history": cd, "tU_over_L": sim.sim_time(), "finite": bool(np.isfinite(u).all()), "umax": float(np.abs(u).max())}), flush=True)


if __name__ == "__main__":
    tgv(128)
    tgv(256)
    sphere(256)
