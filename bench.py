#!/usr/bin/env python3
"""bench.py — cells·steps/s of WaterLily's time step (mom_step!, src/Flow.jl:156-167) on MI355X.

One "step" = one mom_step! (predictor + corrector, two multigrid pressure solves, CFL) of the 3-D
wall-bounded Taylor–Green vortex of SURVEY §8d (Float32, NoBody, remeasure=false), every field resident in
HBM when the timed region starts.  N=1 workload: 512³ (the size BASELINE.json's metric and the ≥60 %
smoother target are quoted on); with --gpus N the same 512³ domain is cut into N z-slabs (strong scaling).

`python bench.py --gpus N` needs no launcher: it starts its N ranks itself (torch.distributed.run children, one per GPU).

Prints ONE JSON line (rank 0): metric/value + `roofline` (the finest-level smoother kernel pair, HIP-event timed inside the
timed region on the launch stream; `frac` = the kernels' OWN algorithmic bytes ÷ time ÷ 8 TB/s, `traffic_frac` = PMC bytes,
`op_equivalent` = the reference formulation's operation bytes ÷ the same time) + `cpu_baseline` (the oracle — a CPU restatement
of the reference algorithm, kind "port" — timed on this box's host cores on bounded samples of the same workload: all cores at
128³ and 256³, one thread at 128³).
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
    # pair kernels with the V-cycle's `x += ω·x_c↓` handed from A to B (wl::XDefer, the default): x makes one round trip per smooth!
    ("A_pro_xd", True): 12.5,   # R r,x_c/8  W r',ϵ_mid
    ("B_xd", True): 20.5,       # R ϵ_mid,r,x,x_c/8  W r',x
}


def _oracle_tgv(orc, n, omp):
    """the bench workload (wall-bounded TGV) on the oracle at n³"""
    import math
    import numpy as np
    sim = orc.Simulation((n, n, n), (0, 0, 0), n, U=1, nu=n / 1600.0, T=np.float32, omp=omp)
    kap = math.pi / n
    u = sim.u
    ax = (np.arange(n + 2, dtype=np.float64) - 0.5)                                # loc(0,I) per axis (0-based array index − 0.5)
    cx, cy, cz = np.cos(kap * ax), np.cos(kap * ax), np.cos(kap * ax)
    sxh, syh = np.sin(kap * (ax - 0.5)), np.sin(kap * (ax - 0.5))
    u[..., 0] = (-sxh[:, None, None] * cy[None, :, None] * cz[None, None, :]).astype(np.float32)
    u[..., 1] = (cx[:, None, None] * syh[None, :, None] * cz[None, None, :]).astype(np.float32)
    u[..., 2] = 0
    orc.BC(u, (0, 0, 0))
    sim.field("u0")[...] = u
    return sim


def _time_oracle(orc, L, n, threads, warm, budget_s, max_steps=400):
    import numpy as np
    L.wlo_set_threads(threads)
    sim = _oracle_tgv(orc, n, omp=True)
    for _ in range(warm):
        sim.step(remeasure=False)
    t0 = time.perf_counter()
    k = 0
    while True:
        sim.step(remeasure=False)
        k += 1
        el = time.perf_counter() - t0
        if el >= budget_s or k >= max_steps:
            break
    nmean = float(np.mean(sim.pois_n[2 * warm:]))
    return {"value": n**3 * k / el, "unit": "cells*steps/s", "cores": int(threads), "kind": "port",
            "sample": f"3D TGV {n}^3 f32, {k} mom_step! after {warm} warm-up, {'serial (1 thread)' if threads == 1 else f'OpenMP over {threads} threads'}, mean pois.n={nmean:.2f}",
            "seconds": el}


def cpu_baseline(budget_s=12.0):
    """Time the oracle (CPU restatement of the reference algorithm, kind "port") on the bench workload on this box's host
    cores, SURVEY §8d: (i) serial — the analogue of the reference's backend="SIMD" loops (src/core.jl:146-155) — at 128³,
    (ii) all cores (one OpenMP parallel-for per @loop, the analogue of the KA CPU backend) at 128³ and, budget permitting, 256³.
    `value` is the all-core figure at the largest size that ran; the others sit beside it.  Bounded to ≈ 2·budget_s."""
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
    serial = _time_oracle(orc, L, 128, 1, 1, budget_s * 0.4, max_steps=12)
    omp128 = _time_oracle(orc, L, 128, cores, 2, budget_s * 0.5)
    out = dict(omp128)
    # 256³ fits the budget when a step takes < budget/6 at the 128³ rate (8× the cells)
    if cores > 1 and 256**3 / omp128["value"] * 6 < budget_s * 1.1:
        try:
            out = _time_oracle(orc, L, 256, cores, 1, budget_s * 0.6, max_steps=12)
            out["omp_128"] = omp128
        except MemoryError:
            pass
    out["serial"] = serial
    L.wlo_set_threads(cores)
    return out


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


KERNEL_CONFIG_FILES = ("wl_fused2_body.inc", "wl_fused2.hip", "wl_fused.hip", "wl_common.hpp")


def kernel_config_sha():
    """identity of the smoother kernels' sources: a PMC traffic figure is only quoted beside a timing of the SAME kernels"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_CONFIG_FILES:
        with open(os.path.join(ROOT, "waterlily.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_roofline(prof, ncell, const0, kind0, N, use_traffic, xdefer=None):
    """`roofline` object for the finest-level smooth! of THIS rank (ncell = the cells its launches process).

    achieved / frac   : the kernels' OWN algorithmic bytes (what each launch must move, BYTES_KERNEL) ÷ HIP-event time ÷ 8 TB/s
    traffic(_frac)    : HBM bytes from the PMC counters (2·FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes) of the same kernel sources, else null
    op_equivalent     : SURVEY §8d's per-OPERATION bytes (GaussSeidelRB! 40 + prolongate!/increment! 36.5 B/cell — what the reference's
                        own formulation would have to move for the work these kernels do) ÷ the same time: a speed-up figure, not a bandwidth."""
    # The roofline kernel: the finest-level smooth! (GaussSeidelRB!, it=4) as executed — the temporally blocked kernel pair A+B.
    if not prof["gsrb_B"]["launches"]:   # experiments with the blocked smoother switched off: report the plain colour sweep instead
        prof["gsrb_B"], prof["gsrb_A"] = prof["gs_sweep"], prof["gs_sweep"]
    kb_ms, ka_ms = prof["gsrb_B"]["avg_ms"], prof["gsrb_A"]["avg_ms"]
    pro_fused = prof["prolong_increment"]["launches"] == 0       # the V-cycle's prolongate!+increment! is folded into kernel A
    pair_ms = ka_ms + kb_ms
    bytes_op = BYTES_OP_GSRB + (BYTES_OP_PROLONG_INC if pro_fused else 0.0)
    # which kernel applied the V-cycle's x += ω·x_c↓: the library's own decision (wl_sim_counter "xdefer"), not an assumption
    x_deferred = bool(pro_fused and const0 and kind0 == 2 and (xdefer == 1 if xdefer is not None else True))
    bytes_a = BYTES_KERNEL[(("A_pro_xd" if x_deferred else "A_pro") if pro_fused else "A", const0)]
    bytes_b = BYTES_KERNEL[("B_xd" if x_deferred else "B", const0)]
    gbs = lambda by, ms: by * ncell / (ms * 1e-3) / 1e9
    traffic, tsrc, tper = None, None, {}
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if use_traffic and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("size") == N and {"A", "B"} <= set(tj.get("kernels", {})) and tj.get("kernel_config_sha") == kernel_config_sha():
                tper = {k: v["hbm_bytes_per_launch"] for k, v in tj["kernels"].items()}
                traffic, tsrc = tper["A"] + tper["B"], tj.get("source")
        except Exception:
            pass
    kname = {2: "k_gsrb2_A + k_gsrb2_B (wl_fused2.hip, pair kernels)", 1: "k_gsrb_A + k_gsrb_B (wl_fused.hip)", 0: "k_gs_sweep passes"}[kind0]
    own = bytes_a + bytes_b
    roof = {
        "bound": "hbm",
        "kernel": f"finest-level smooth! = GaussSeidelRB!(it=4) (src/Poisson.jl:141-148){' + the V-cycle prolongate!+increment! (src/MultiLevelPoisson.jl:99-100)' if pro_fused else ''}, executed as {kname}",
        "achieved": gbs(own, pair_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(own, pair_ms) / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": tsrc,
        "traffic_frac": (traffic / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "bytes_per_cell": own, "bytes_definition": "the kernels' own algorithmic bytes per interior cell and launch pair (distinct elements each launch must read + write; "
                          "L, D, iD are evaluated from the cell position on constant-coefficient levels) — see kernels.*",
        "avg_launch_ms": pair_ms, "launches": prof["gsrb_B"]["launches"],
        "constant_coefficient_kernels": const0, "x_increment_deferred_to_B": x_deferred,
        "op_equivalent": {"bytes_per_cell": bytes_op, "achieved": gbs(bytes_op, pair_ms), "frac": gbs(bytes_op, pair_ms) / HBM_PEAK_GBS,
                          "what": "reference-defined operation bytes (SURVEY §8d: GaussSeidelRB! 40" + (" + prolongate!/increment! 36.5" if pro_fused else "")
                                  + " B/cell) ÷ the same time — the rate the reference's formulation would need for this work; NOT bytes moved"},
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


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU) through torch.distributed.run and relay
    their output; rank 0 prints the JSON line.  Called before this process imports torch or touches the GPU; the children are fresh
    processes (no exec of a GPU-initialised process)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=512, help="interior cells per side of the TGV box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="HIP-event pairs around every phase of the step (phases_ms_per_step); the default run "
                    "only brackets the roofline kernels — each pair costs a few µs of stream time, ≈1 %% of a 512³ step with all of them on")
    ap.add_argument("--no-phases", action="store_true", help="with WL_OPT_* switches set: keep the default (roofline kernels only) event pairs")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or run without a launcher)")
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("WL_BENCH_DRY"):
        # launcher rehearsal without a GPU (CPU test of the --gpus N contract): rendezvous over gloo, count the ranks, rank 0 prints a line
        import torch
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group(backend="gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            nr = int(t.item())
            dist.destroy_process_group()
        else:
            nr = 1
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": nr, "steps": args.steps, "warmup": args.warmup, "size": args.size}), flush=True)
        return
    import torch
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("WL_BENCH_ONE_GPU"):      # rehearsal of the multi-rank path on a one-GPU box (with WL_DIST_BACKEND=gloo)
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} devices, this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    import waterlily_jl_amd as w
    from waterlily_jl_amd._lib import check
    lib = w.lib()
    check(lib.wl_init(local_rank))

    N = args.size
    if world > 1:
        from waterlily_jl_amd import slab
        return slab.bench_main(args, world, rank, local_rank, read_prof=read_prof, build_roofline=build_roofline,
                               cpu_baseline=None if args.no_cpu_baseline else (lambda: cpu_baseline(budget_s=args.cpu_budget)))

    sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
    import ctypes as C
    sc = (C.c_double * 8)()
    nsc = lib.wl_placement_scores(sc, 8)      # wl_sim_create's placement trials (large grids): ms of a timed mom_project! pair per candidate allocation; the fastest was kept
    placement = [round(float(sc[i]), 3) for i in range(min(nsc, 8))]
    for key, val in os.environ.items():      # A/B switches for experiments: WL_OPT_<option of wl_sim_set_option>=0/1 (defaults: fast paths on)
        if key.startswith("WL_OPT_"):
            sim.set_option(key[7:], int(val))
    sim.mom_steps_(args.warmup)      # wl_sim_mom_steps: the K steps of the timed region are ONE library call too (same results as K calls of wl_sim_mom_step)
    sim.sync()
    n_warm = len(sim.pois_n)
    all_phases = args.phases or (any(k.startswith("WL_OPT_") for k in os.environ) and not args.no_phases)
    check(lib.wl_prof_enable(1 if all_phases else 2))
    def counter(name):
        v = C.c_long()
        check(lib.wl_sim_counter(sim._h, name.encode(), C.byref(v)))
        return int(v.value)
    l0, rj0, rd0 = int(lib.wl_launch_count()), counter("resjac"), counter("resjac_redo")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.mom_steps_(args.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    launches_per_step = (int(lib.wl_launch_count()) - l0) / args.steps
    prof = read_prof(lib)
    check(lib.wl_prof_enable(0))
    ncell = float(N) ** 3
    pn = sim.pois_n[n_warm:]
    roof = build_roofline(prof, ncell, bool(sim.const_levels()[0]), sim.smoother_kinds()[0], N, use_traffic=True, xdefer=counter("xdefer"))
    out = {
        "metric": "cells*steps/sec (3D TGV) ; smoother HBM GB/s vs peak", "value": ncell * args.steps / el, "unit": "cells*steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"3D Taylor-Green vortex {N}^3 Float32, wall-bounded, Re=1600, NoBody, remeasure=false (BASELINE configs[4] domain on 1 GPU)",
                   "size": N, "mean_pois_n": float(sum(pn)) / max(1, len(pn)), "dt_last": float(sim.dt[-1]),
                   "constant_coefficient_levels": sim.const_levels(), "smoother_kinds": sim.smoother_kinds(),
                   # kernel launches per mom_step! in the timed region; solves whose fused projection head stood / had to be redone (residual!'s mean shift due)
                   "placement_trial_ms": placement, "launches_per_step": launches_per_step, "resjac": counter("resjac") - rj0, "resjac_redo": counter("resjac_redo") - rd0},
        "roofline": roof,
        "phases_ms_per_step": {k: (v["total_ms"] / args.steps) for k, v in prof.items() if v["launches"]},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(budget_s=args.cpu_budget)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
