#!/usr/bin/env python3
"""What ONE GPU can measure of the N-GPU z-slab run (VERDICT r02 "next" #4): rank r's exact slab of the 512^3 TGV — 512/P owned planes,
5-deep ghost planes, the distributed/replicated level split of a P-rank run — stepped alone on one MI355X with the RCCL communicator in
rehearsal mode (wl_comm_set_virtual: ncclSend/ncclRecv to itself, in-place ncclAllGather).  Per (P, rank): compute ms per step, exchange
rounds / bytes / scalar combines / plane all-gathers per step, launches per step, pois.n; plus the latency of one exchange round as issued
through RCCL on this box (loopback: an upper bound on the software cost, not the xGMI transfer time).
usage (GPU box): python tools/slab_rank_bench.py [size] > gpurun_out/slab_rank.json"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", device_id=dev)
import waterlily_jl_amd as w
from waterlily_jl_amd import slab
from waterlily_jl_amd._lib import check, lib
L = lib()
check(L.wl_init(0))
out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "size": N, "cases": [], "exchange_latency": []}

# single-domain reference on the same box
ref = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
for _ in range(3):
    ref.mom_step_()
ref.sync(); l0 = L.wl_launch_count(); t0 = time.perf_counter()
for _ in range(10):
    ref.mom_step_()
ref.sync(); el = time.perf_counter() - t0
out["single_domain"] = {"ms_per_step": el / 10 * 1e3, "launches_per_step": (L.wl_launch_count() - l0) / 10, "mean_pois_n": sum(ref.pois_n[6:]) / 20}
del ref
torch.cuda.empty_cache()

CASES = ((8, (0, 3)), (4, (0, 1)), (2, (0,)))
if len(sys.argv) > 2:      # e.g. "8:0,3" — only these ranks of an 8-rank run
    CASES = tuple((int(c.split(":")[0]), tuple(int(r) for r in c.split(":")[1].split(","))) for c in sys.argv[2:])
for P, ranks in CASES:
    for r in ranks:
      for transport in (1,):          # 1: exchanges through RCCL to itself.  (0 = no transfers at all exists — wl_comm_set_virtual_transport — but stale ghost planes make residual!'s mean shift due and the step takes the redo path: not the same work)
          comm = slab.RcclComm(dist, dev)
          check(L.wl_comm_set_virtual(comm.handle, r, P))
          check(L.wl_comm_set_virtual_transport(comm.handle, transport))
          sim = slab.SlabSimulation(comm, (N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
          g = sim.grid
          # the slab sees ITSELF as its neighbours: that field is not a solution of the P-rank problem and the solver would iterate to its cap.  The real
          # run takes the single mandatory V-cycle per solve (single_domain.mean_pois_n = 1): the rehearsal is capped at that, so that it does the same work.
          check(L.wl_sim_set_option(sim._h, b"itmx", 1))
          for _ in range(3):
              sim.mom_step_()
          sim.sync(); torch.cuda.synchronize()
          nw = len(sim.pois_n)
          cs0 = slab.comm_stats(comm); l0 = L.wl_launch_count()
          t0 = time.perf_counter()
          steps = 10
          for _ in range(steps):
              sim.mom_step_()
          sim.sync(); torch.cuda.synchronize()
          el = time.perf_counter() - t0
          cs1 = slab.comm_stats(comm)
          pn = sim.pois_n[nw:]
          case = {"ranks": P, "rank": r, "transport": "rccl-to-self" if transport else "none (pure compute)", "owned_planes": g.k1 - g.k0, "local_planes_with_ghosts": g.nz, "ms_per_step": el / steps * 1e3,
                  "launches_per_step": (L.wl_launch_count() - l0) / steps, "mean_pois_n": sum(pn) / max(1, len(pn)),
                  "comm_per_step": {k: (cs1[k] - cs0[k]) / steps for k in cs1},
                  "ideal_share_of_single_domain_ms": out["single_domain"]["ms_per_step"] / P}
          case["compute_efficiency_vs_ideal_share"] = case["ideal_share_of_single_domain_ms"] / case["ms_per_step"]
          out["cases"].append(case)
          print(json.dumps(case), file=sys.stderr, flush=True)
          # latency of exchange rounds issued through RCCL on this box (self send/recv): the software + launch cost per round
          if (P, r) == (8, 3) and transport == 1:
              from waterlily_jl_amd._lib import wl_grid
              a = torch.zeros((3, g.nz, g.ny, g.nx), dtype=torch.float32, device=dev)
              for ncomp, depth, label in ((1, 5, "smooth!: r, 5 planes"), (3, 2, "BC!: u, 3 components x 2 planes"), (1, 1, "x, 1 plane")):
                  for fn, tag in ((L.wl_halo_exchange, "compute stream"), (L.wl_comm_halo_async, "communicator's stream")):
                      for _ in range(5):
                          check(fn(comm.handle, C.c_void_p(a.data_ptr()), C.byref(g), ncomp, depth, None))
                      torch.cuda.synchronize(); t0 = time.perf_counter()
                      reps = 100
                      for _ in range(reps):
                          check(fn(comm.handle, C.c_void_p(a.data_ptr()), C.byref(g), ncomp, depth, None))
                      torch.cuda.synchronize(); us = (time.perf_counter() - t0) / reps * 1e6
                      by = 2 * ncomp * depth * g.nx * g.ny * 4
                      out["exchange_latency"].append({"exchange": label, "stream": tag, "bytes_sent_per_round": by, "us_per_round": us, "loopback_GBps": by / us / 1e3})
          del sim
          comm.destroy()
          torch.cuda.empty_cache()
dist.destroy_process_group()
print(json.dumps(out, indent=1))
