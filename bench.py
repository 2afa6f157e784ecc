#!/usr/bin/env python3
"""bench.py — cells·steps/s of WaterLily's time step (mom_step!, src/Flow.jl:156-167) on MI355X.

One "step" = one mom_step! (predictor + corrector, two multigrid pressure solves, CFL) of the 3-D
wall-bounded Taylor–Green vortex of SURVEY §8d (Float32, NoBody, remeasure=false), every field resident in
HBM when the timed region starts.  N=1 workload: 512³ (the size BASELINE.json's metric and the ≥60 %
smoother target are quoted on); with --gpus N the same 512³ domain is cut into N z-slabs (strong scaling).

Prints ONE JSON line (rank 0): metric/value + `roofline` (dominant kernel, HIP-event timed inside the timed
region on the launch stream) + `cpu_baseline` (the oracle — a CPU restatement of the reference algorithm,
kind "port" — timed on this box's host cores on a bounded 128³ sample of the same workload).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ≈6.3 achievable)
# Algorithmic bytes per interior cell, f32, 3-D.
# (1) per OPERATION as the reference API defines it (SURVEY §8d / BASELINE.md §4): distinct elements read + written
BYTES_OP_GSRB = 40.0          # GaussSeidelRB!(it=4) as ONE operation: R r,iD,L₁₋₃,D,x  W ϵ,r,x
BYTES_OP_PROLONG_INC = 36.5   # prolongate!+increment! (src/MultiLevelPoisson.jl:99-100) with the ϵ round trip fused away
# (2) per KERNEL as built here (what each launch must move; D, iD recomputed in registers, final ϵ not stored by the composite)
BYTES_KERNEL = {
    # general coefficients (L loaded)                       constant-coefficient levels (NoBody: L, D, iD evaluated, never loaded)
    ("A", False): 20.0,   # R r,L₁₋₃  W ϵ_mid                   ("A", True): R r  W ϵ_mid
    ("A", True): 8.0,
    ("A_pro", False): 32.5,   # R r,x,L₁₋₃,x_c/8  W r',x,ϵ_mid
    ("A_pro", True): 20.5,    # R r,x,x_c/8  W r',x,ϵ_mid
    ("B", False): 32.0,   # R ϵ_mid,r,L₁₋₃,x  W r',x
    ("B", True): 20.0,    # R ϵ_mid,r,x  W r',x
}


def cpu_baseline(n=128, warm=2, steps=None, budget_s=12.0):
    """Time the oracle (CPU restatement, OpenMP = analogue of the reference's KA CPU threads) on the same
    TGV workload at 128³.  Bounded: stops after `budget_s` seconds of timed work."""
    import math
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    L = orc.lib(omp=True)
    # threads = the CPUs this process may actually use (cgroup quota / affinity), not the host's logical count
    cores = min(L.wlo_max_threads(), len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per))))
    except Exception:
        pass
    cores = int(os.environ.get("WL_CPU_THREADS", cores))
    L.wlo_set_threads(cores)
    sim = orc.Simulation((n, n, n), (0, 0, 0), n, U=1, nu=n / 1600.0, T=np.float32, omp=True)
    kap = math.pi / n
    u = sim.u
    ax = np.arange(n + 2, dtype=np.float32) + np.float32(1) - np.float32(1.5)     # loc(0,I) per axis
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    h = np.float32(0.5)
    u[..., 0] = (-np.sin(kap * (X - h).astype(np.float64)) * np.cos(kap * Y.astype(np.float64)) * np.cos(kap * Z.astype(np.float64))).astype(np.float32)
    u[..., 1] = (np.cos(kap * X.astype(np.float64)) * np.sin(kap * (Y - h).astype(np.float64)) * np.cos(kap * Z.astype(np.float64))).astype(np.float32)
    u[..., 2] = 0
    orc.BC(u, (0, 0, 0))
    sim.field("u0")[...] = u
    for _ in range(warm):
        sim.step(remeasure=False)
    t0 = time.perf_counter()
    k = 0
    while True:
        sim.step(remeasure=False)
        k += 1
        el = time.perf_counter() - t0
        if (steps is not None and k >= steps) or (steps is None and el >= budget_s) or k >= 400:
            break
    el = time.perf_counter() - t0
    nmean = float(np.mean(sim.pois_n[2 * warm:]))
    return {"value": n**3 * k / el, "unit": "cells*steps/s", "cores": int(cores), "kind": "port",
            "sample": f"3D TGV {n}^3 f32, {k} mom_step! after {warm} warm-up, OpenMP over {cores} threads, mean pois.n={nmean:.2f}",
            "seconds": el}


def read_prof(lib):
    """HIP-event timings recorded by the library on its launch stream (wl_prof_*), per named slot"""
    from waterlily_jl_amd._lib import check
    prof = {}
    names = {0: "gs_sweep", 1: "smooth", 2: "jacobi", 3: "conv_diff", 4: "residual", 5: "bdim", 6: "prolong_increment", 7: "coarse_levels", 8: "mom_step",
             9: "gsrb_A", 10: "gsrb_B"}
    for slot, nm in names.items():
        cnt, tot = C.c_int(), C.c_double()
        check(lib.wl_prof_read(slot, C.byref(cnt), C.byref(tot)))
        prof[nm] = {"launches": cnt.value, "avg_ms": (tot.value / cnt.value) if cnt.value else None, "total_ms": tot.value}
    return prof


def build_roofline(prof, ncell, const0, kind0, N, use_traffic):
    """`roofline` object for the finest-level smooth! of THIS rank (ncell = the cells its launches process)"""
    # The roofline kernel: the finest-level smooth! (GaussSeidelRB!, it=4) as executed — the temporally blocked kernel pair A+B.
    if not prof["gsrb_B"]["launches"]:   # experiments with the blocked smoother switched off: report the plain colour sweep instead
        prof["gsrb_B"], prof["gsrb_A"] = prof["gs_sweep"], prof["gs_sweep"]
    kb_ms, ka_ms = prof["gsrb_B"]["avg_ms"], prof["gsrb_A"]["avg_ms"]
    pro_fused = prof["prolong_increment"]["launches"] == 0       # the V-cycle's prolongate!+increment! is folded into kernel A
    pair_ms = ka_ms + kb_ms
    bytes_op = BYTES_OP_GSRB + (BYTES_OP_PROLONG_INC if pro_fused else 0.0)
    bytes_a = BYTES_KERNEL[("A_pro" if pro_fused else "A", const0)]
    bytes_b = BYTES_KERNEL[("B", const0)]
    gbs = lambda by, ms: by * ncell / (ms * 1e-3) / 1e9
    traffic, tsrc, tper = None, None, {}
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if use_traffic and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("size") == N and {"A", "B"} <= set(tj.get("kernels", {})):
                tper = {k: v["hbm_bytes_per_launch"] for k, v in tj["kernels"].items()}
                traffic, tsrc = tper["A"] + tper["B"], tj.get("source")
        except Exception:
            pass
    kname = {2: "k_gsrb2_A + k_gsrb2_B (wl_fused2.hip, pair kernels)", 1: "k_gsrb_A + k_gsrb_B (wl_fused.hip)", 0: "k_gs_sweep passes"}[kind0]
    roof = {
        "bound": "hbm",
        "kernel": f"finest-level smooth! = GaussSeidelRB!(it=4) (src/Poisson.jl:141-148){' + the V-cycle prolongate!+increment! (src/MultiLevelPoisson.jl:99-100)' if pro_fused else ''}, executed as {kname}",
        # SURVEY §8d's per-operation figure × cells ÷ HIP-event duration of the kernel pair (measured live, launch stream)
        "achieved": gbs(bytes_op, pair_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(bytes_op, pair_ms) / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": tsrc,
        "bytes_per_cell": bytes_op, "bytes_definition": "reference-defined operation bytes (SURVEY §8d): GaussSeidelRB! 40"
                          + (" + prolongate!/increment! 36.5" if pro_fused else "") + " B/cell; the kernels move fewer (see kernels.*)",
        "avg_launch_ms": pair_ms, "launches": prof["gsrb_B"]["launches"],
        "constant_coefficient_kernels": const0,
        "own_bytes": {"bytes_per_cell": bytes_a + bytes_b, "achieved": gbs(bytes_a + bytes_b, pair_ms), "frac": gbs(bytes_a + bytes_b, pair_ms) / HBM_PEAK_GBS,
                      "what": "bytes the kernel pair itself must move (its own algorithmic minimum) ÷ the same time"},
        "kernels": {
            "A": {"what": ("prolongate!+increment! + eps=r*iD + colour sweeps 1,2" if pro_fused else "eps=r*iD + colour sweeps 1,2"),
                  "bytes_per_cell": bytes_a, "avg_ms": ka_ms, "achieved": gbs(bytes_a, ka_ms), "frac": gbs(bytes_a, ka_ms) / HBM_PEAK_GBS,
                  "traffic": tper.get("A")},
            "B": {"what": "colour sweeps 3,4 + increment! (+ L1/Linf of the new residual)",
                  "bytes_per_cell": bytes_b, "avg_ms": kb_ms, "achieved": gbs(bytes_b, kb_ms), "frac": gbs(bytes_b, kb_ms) / HBM_PEAK_GBS,
                  "traffic": tper.get("B")},
        },
    }
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=512, help="interior cells per side of the TGV box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="HIP-event pairs around every phase of the step (phases_ms_per_step); the default run "
                    "only brackets the roofline kernels — each pair costs a few µs of stream time, ≈1 %% of a 512³ step with all of them on")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("WL_BENCH_ONE_GPU"):      # rehearsal of the multi-rank path on a one-GPU box (with WL_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import waterlily_jl_amd as w
    from waterlily_jl_amd._lib import check
    lib = w.lib()
    check(lib.wl_init(local_rank))

    N = args.size
    if world > 1:
        from waterlily_jl_amd import slab
        return slab.bench_main(args, world, rank, local_rank, read_prof=read_prof, build_roofline=build_roofline)

    sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
    for key, val in os.environ.items():      # A/B switches for experiments: WL_OPT_<option of wl_sim_set_option>=0/1 (defaults: fast paths on)
        if key.startswith("WL_OPT_"):
            sim.set_option(key[7:], int(val))
    for _ in range(args.warmup):
        sim.mom_step_()
    sim.sync()
    n_warm = len(sim.pois_n)
    all_phases = args.phases or any(k.startswith("WL_OPT_") for k in os.environ)
    check(lib.wl_prof_enable(1 if all_phases else 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.mom_step_()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = read_prof(lib)
    check(lib.wl_prof_enable(0))
    ncell = float(N) ** 3
    pn = sim.pois_n[n_warm:]
    roof = build_roofline(prof, ncell, bool(sim.const_levels()[0]), sim.smoother_kinds()[0], N, use_traffic=True)
    out = {
        "metric": "cells*steps/sec (3D TGV) ; smoother HBM GB/s vs peak", "value": ncell * args.steps / el, "unit": "cells*steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"3D Taylor-Green vortex {N}^3 Float32, wall-bounded, Re=1600, NoBody, remeasure=false (BASELINE configs[4] domain on 1 GPU)",
                   "size": N, "mean_pois_n": float(sum(pn)) / max(1, len(pn)), "dt_last": float(sim.dt[-1]),
                   "constant_coefficient_levels": sim.const_levels(), "smoother_kinds": sim.smoother_kinds()},
        "roofline": roof,
        "phases_ms_per_step": {k: (v["total_ms"] / args.steps) for k, v in prof.items() if v["launches"]},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(budget_s=args.cpu_budget)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
