"""z-slab multi-GPU path (one process per GPU): communicator construction and the slab Simulation.

The reference has no multi-device path (README.md:153-155) — this is new design: every rank owns Nz/P interior
x-y planes (+2 ghost planes per side), stencil kernels index cells globally (colour parity, wall faces), halos
are single contiguous planes, coarse multigrid levels are replicated on every rank below 32 planes, and the
scalars that steer the solver are combined on device from an all-gather so all ranks branch identically.
Transport: RCCL (ncclSend/ncclRecv/ncclAllGather on the compute stream, over xGMI) in production; for tests a
callback communicator moves the same buffers through torch.distributed (gloo) and host memory.
"""
import ctypes as C
import json
import os
import time

import numpy as np

from ._lib import ALLGATHER_FN, SENDRECV_FN, check, lib, wl_grid, wl_sim_desc
from .core import perdir_mask


def slab_grid(global_dims_with_ghosts, rank, size, halo=2):
    """wl_grid of `rank` (host helper of the library — no GPU needed)."""
    g = wl_grid()
    arr = (C.c_int32 * 3)(*global_dims_with_ghosts)
    check(lib().wl_grid_slab(C.byref(g), 3, arr, rank, size, halo))
    return g


class CallbackComm:
    """wl_comm whose transport is torch.distributed point-to-point / all_gather on HOST tensors (gloo).

    stage_out(dst_numpy_uint8, src_ptr, nbytes) / stage_in(dst_ptr, src_numpy_uint8, nbytes) move bytes between the
    library's buffers and host memory: wl_d2h / wl_h2d on a GPU, plain memmove when the 'device' buffers are host
    arrays (CPU tests of the halo plumbing)."""

    def __init__(self, dist, rank=None, size=None, host_buffers=False, group=None):
        import torch
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank() if rank is None else rank
        self.size = dist.get_world_size() if size is None else size
        L = lib()
        if host_buffers:
            self._out = lambda dst, src, n: C.memmove(dst.ctypes.data, src, n)
            self._in = lambda dst, src, n: C.memmove(dst, src.ctypes.data, n)
            self._sync = lambda st: None
        else:
            self._out = lambda dst, src, n: check(L.wl_d2h(dst.ctypes.data_as(C.c_void_p), src, n, None))
            self._in = lambda dst, src, n: (check(L.wl_h2d(dst, src.ctypes.data_as(C.c_void_p), n, None)), check(L.wl_stream_sync(None)))
            self._sync = lambda st: check(L.wl_stream_sync(st))
        self._sr = SENDRECV_FN(self._sendrecv)
        self._ag = ALLGATHER_FN(self._allgather)
        h = C.c_void_p()
        check(L.wl_comm_callbacks_create(C.byref(h), self.rank, self.size, None, C.cast(self._sr, C.c_void_p), C.cast(self._ag, C.c_void_p)))
        self.handle = h

    def _sendrecv(self, ctx, slo, rlo, shi, rhi, nbytes, stream):
        try:
            torch, dist = self.torch, self.dist
            self._sync(stream)
            reqs, keep = [], []
            # peers modulo the size (a z-periodic domain wraps around); tag 0: planes travelling down (my lower planes -> the lower neighbour's upper
            # ghosts), tag 1: planes travelling up — with two ranks both neighbours are the same process and the tags keep the two apart
            lo, hi = (self.rank - 1) % self.size, (self.rank + 1) % self.size
            if slo:
                b = np.empty(nbytes, dtype=np.uint8); self._out(b, slo, nbytes); t = torch.from_numpy(b); keep.append(t)
                reqs.append(dist.isend(t, lo, group=self.group, tag=0))
            if shi:
                b = np.empty(nbytes, dtype=np.uint8); self._out(b, shi, nbytes); t = torch.from_numpy(b); keep.append(t)
                reqs.append(dist.isend(t, hi, group=self.group, tag=1))
            rl = rh = None
            if rlo:
                rl = torch.empty(nbytes, dtype=torch.uint8); reqs.append(dist.irecv(rl, lo, group=self.group, tag=1))
            if rhi:
                rh = torch.empty(nbytes, dtype=torch.uint8); reqs.append(dist.irecv(rh, hi, group=self.group, tag=0))
            for r in reqs:
                r.wait()
            if rl is not None:
                self._in(rlo, rl.numpy(), nbytes)
            if rh is not None:
                self._in(rhi, rh.numpy(), nbytes)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("halo callback error:", repr(e), flush=True)
            return 1

    def _allgather(self, ctx, send, recv, nbytes, stream):
        try:
            torch, dist = self.torch, self.dist
            self._sync(stream)
            b = np.empty(nbytes, dtype=np.uint8); self._out(b, send, nbytes)
            outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.size)]
            dist.all_gather(outs, torch.from_numpy(b), group=self.group)
            for r, t in enumerate(outs):
                self._in(recv + r * nbytes, t.numpy(), nbytes)
            return 0
        except Exception as e:
            print("allgather callback error:", repr(e), flush=True)
            return 1

    def destroy(self):
        if self.handle:
            lib().wl_comm_destroy(self.handle)
            self.handle = None


class RcclComm:
    """wl_comm over RCCL.  Two communicators (compute-stream traffic / overlapped halo stream), whose unique ids are created on
    rank 0 and broadcast through torch.distributed.  Ranks first AGREE (all-reduce) that RCCL is loadable everywhere, so that
    no rank enters ncclCommInitRank while another has already given up; any failure raises on every rank — there is no fallback."""

    def __init__(self, dist, device, dual=True):
        import torch
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        L = lib()
        on_dev = dist.get_backend() == "nccl"
        ok = torch.tensor([int(L.wl_comm_rccl_available())], dtype=torch.int32, device=device if on_dev else "cpu")
        if self.size > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            raise WlCommError("RCCL (librccl.so.1) is not loadable on every rank")
        nid = 2 if dual else 1
        raw = bytearray(128 * nid + 1)
        if self.rank == 0:
            good = 1
            for q in range(nid):
                uid = C.create_string_buffer(128)
                if L.wl_comm_rccl_unique_id(uid) != 0:
                    good = 0
                raw[128 * q:128 * (q + 1)] = uid.raw
            raw[-1] = good
        t = torch.frombuffer(raw, dtype=torch.uint8).clone()
        if self.size > 1:
            if on_dev:
                t = t.to(device)
            dist.broadcast(t, 0)
        raw = bytes(t.cpu().numpy().tobytes())
        if raw[-1] != 1:
            raise WlCommError("ncclGetUniqueId failed on rank 0")
        h = C.c_void_p()
        check(L.wl_comm_rccl_create(C.byref(h), self.rank, self.size, raw[:128]))
        self.handle = h
        if dual:
            check(L.wl_comm_rccl_add_async(h, raw[128:256]))

    def destroy(self):
        if self.handle:
            lib().wl_comm_destroy(self.handle)
            self.handle = None


class WlCommError(RuntimeError):
    pass


class SlabSimulation:
    """Simulation on a z-slab (wl_sim_create_slab): same step as FusedSimulation, fields distributed along z."""

    def __init__(self, comm, dims, uBC, L, U=None, dt=0.25, nu=0.0, perdir=(), lam=0, has_body=False, ic="uBC", exitBC=False):
        D = 3
        assert len(dims) == 3
        if U is None:
            U = float(np.sqrt(sum(float(v) ** 2 for v in uBC)))
        self.comm, self.U, self.L, self.D = comm, float(U), float(L), D
        self.dims = tuple(int(n) for n in dims)
        d = wl_sim_desc()
        d.D = D
        for k in range(3):
            d.dims[k] = self.dims[k]
            d.uBC[k] = float(uBC[k])
        d.nu, d.dt0 = float(nu), float(dt)
        d.perdir_mask, d.exitBC, d.scheme, d.has_body = perdir_mask(perdir), int(bool(exitBC)), int(lam), int(bool(has_body))
        h = C.c_void_p()
        check(lib().wl_sim_create_slab(C.byref(h), C.byref(d), comm.handle))
        self._h = h
        g = wl_grid()
        check(lib().wl_sim_grid(h, C.byref(g)))
        self.grid = g
        check(lib().wl_sim_apply_ic(h, {"uBC": 0, "tgv": 1, "tgv_periodic": 2}[ic], None))
        check(lib().wl_sim_init_flow(h, None))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().wl_sim_destroy(h)
            except Exception:
                pass
            self._h = None

    def mom_step_(self):
        check(lib().wl_sim_mom_step(self._h, None))

    def measure_sphere_(self, center, R, eps=1.0):
        """measure!(sim) for AutoBody(|x-c|-R) in GLOBAL coordinates: closed form on device + update!(pois)"""
        c = (C.c_float * 3)(*[float(v) for v in center])
        check(lib().wl_sim_measure_sphere(self._h, c, float(R), float(eps), None))

    def measure_body_(self, body, eps=1.0):
        from ._lib import make_body
        b = make_body(body, self.D)
        check(lib().wl_sim_measure_body(self._h, C.byref(b), float(eps), None))

    def pressure_force_body(self, body):
        """pressure_force(sim) over all slabs — collective: every rank calls it and gets the global value"""
        from ._lib import make_body
        b = make_body(body, self.D)
        out = (C.c_double * 3)()
        check(lib().wl_sim_pressure_force_body(self._h, C.byref(b), out, None))
        return np.array(out[:3])

    def viscous_force_body(self, body):
        """viscous_force(sim) over all slabs — collective"""
        from ._lib import make_body
        b = make_body(body, self.D)
        out = (C.c_double * 3)()
        check(lib().wl_sim_viscous_force_body(self._h, C.byref(b), out, None))
        return np.array(out[:3])

    def total_force_body(self, body):
        return self.pressure_force_body(body) + self.viscous_force_body(body)

    def sync(self):
        check(lib().wl_stream_sync(None))

    @property
    def dt(self):
        out = (C.c_float * 100000)()
        k = lib().wl_sim_dt(self._h, out, 100000)
        return [np.float32(v) for v in out[:k]]

    @property
    def pois_n(self):
        out = (C.c_int16 * 65536)()
        k = lib().wl_mg_history(lib().wl_sim_pois(self._h), out, 65536)
        return [int(v) for v in out[:k]]

    def local_field(self, name):
        """this rank's slab (all local planes incl. ghosts), Fortran order"""
        g = self.grid
        nc = {"p": (), "sigma": ()}.get(name, (3,))
        out = np.empty((g.nx, g.ny, g.nz) + nc, dtype=np.float32, order="F")
        check(lib().wl_d2h(out.ctypes.data_as(C.c_void_p), lib().wl_sim_field(self._h, name.encode()), out.nbytes, None))
        return out

    def gather_field(self, name, dist):
        """assemble the global ghosted array on every rank from the owned planes (+ the physical z-ghost planes)"""
        import torch
        g = self.grid
        loc = self.local_field(name)
        lo = g.k0 - (1 if g.gk + g.k0 == 1 else 0)
        hi = g.k1 + (1 if g.gk + g.k1 == g.gnz - 1 else 0)
        mine = np.ascontiguousarray(loc[:, :, lo:hi])
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, (g.gk + lo, mine))
        nc = mine.shape[3:]
        full = np.zeros((g.nx, g.ny, g.gnz) + nc, dtype=np.float32, order="F")
        for z0, a in parts:
            full[:, :, z0:z0 + a.shape[2]] = a
        return full


def comm_stats(comm):
    """{halo exchanges, bytes sent in them by this rank, scalar combines, plane all-gathers} since the communicator was created"""
    out = (C.c_int64 * 4)()
    check(lib().wl_comm_stats(comm.handle, out))
    return {"halo_exchanges": int(out[0]), "halo_bytes_sent": int(out[1]), "scalar_combines": int(out[2]), "plane_allgathers": int(out[3])}


def make_comm(dist, device):
    """The transport follows the process group: nccl(=RCCL) -> RcclComm (ncclSend/Recv over xGMI), and a failure to create it is an
    error on every rank (no silent fallback: a scaling number on host staging would be meaningless); gloo -> the host-staged
    callback transport, which exists for tests and one-GPU rehearsals and is labelled as such in the bench line."""
    if dist.get_backend() == "nccl":
        return RcclComm(dist, device)
    return CallbackComm(dist)


def bench_main(args, world, rank, local_rank, read_prof=None, build_roofline=None, cpu_baseline=None):
    """bench.py --gpus N (N>1): the same 512³ TGV cut into N z-slabs (strong scaling), one process per GPU."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("WL_DIST_BACKEND", "nccl")
    dist.init_process_group(backend=backend, device_id=dev if backend == "nccl" else None)
    comm = make_comm(dist, dev)
    N = args.size
    sim = SlabSimulation(comm, (N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
    for _ in range(args.warmup):
        sim.mom_step_()
    sim.sync()
    n_warm = len(sim.pois_n)
    if rank == 0 and read_prof is not None:
        check(lib().wl_prof_enable(2))          # HIP-event pairs around rank 0's finest-level smoother kernels only
    dist.barrier()
    torch.cuda.synchronize()
    cs0 = comm_stats(comm)
    l0 = int(lib().wl_launch_count())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.mom_step_()
    torch.cuda.synchronize()
    dist.barrier()
    el = time.perf_counter() - t0
    cs1 = comm_stats(comm)
    launches_per_step = (int(lib().wl_launch_count()) - l0) / args.steps
    t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    roof = None
    if rank == 0 and read_prof is not None:
        try:
            prof = read_prof(lib())
            check(lib().wl_prof_enable(0))
            mg = lib().wl_sim_pois(sim._h)
            g = sim.grid
            ncell_local = float(N) * float(N) * float(g.k1 - g.k0)
            import ctypes as _C
            xd = _C.c_long(-1)
            check(lib().wl_sim_counter(sim._h, b"xdefer", _C.byref(xd)))
            roof = build_roofline(prof, ncell_local, bool(lib().wl_mg_level_is_const(mg, 0)), int(lib().wl_mg_smoother_kind(mg, 0)), N, use_traffic=False, xdefer=int(xd.value))
            roof["scope"] = f"rank 0 of {world}: its {g.k1 - g.k0} planes of the finest level"
        except Exception as e:   # noqa: BLE001 — the throughput line must not depend on the optional roofline block
            roof = {"error": repr(e)}
    if rank == 0:
        pn = sim.pois_n[n_warm:]
        out = {"metric": "cells*steps/sec (3D TGV) ; smoother HBM GB/s vs peak", "value": float(N) ** 3 * args.steps / el, "unit": "cells*steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"3D Taylor-Green vortex {N}^3 Float32, wall-bounded, Re=1600, NoBody, remeasure=false, {world} z-slabs",
                          "size": N, "parallelism": f"zslab{world}", "transport": type(comm).__name__, "mean_pois_n": float(sum(pn)) / max(1, len(pn)),
                          "dt_last": float(sim.dt[-1]), "launches_per_step_rank0": launches_per_step,
                          "comm_per_step_rank0": {k: (cs1[k] - cs0[k]) / args.steps for k in cs1}},
               "roofline": roof, "cpu_baseline": None}
        if cpu_baseline is not None:      # outside the timed region; the other ranks wait at the barrier below
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:   # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    dist.barrier()
    del sim
    comm.destroy()
    dist.destroy_process_group()
