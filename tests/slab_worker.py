"""Worker for the multi-process slab tests (launched by test_slab_cpu.py / test_gpu_slab.py, one process per rank).
mode cpu_halo : gloo + HOST buffers — exercises the library's halo/all-gather pointer arithmetic and the transport (no GPU).
mode gpu_sim  : gloo + the one GPU of the box — P slab ranks against the single-domain FusedSimulation.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1]
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="nccl" if mode == "gpu_rccl1" else "gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    import waterlily_jl_amd as w
    from waterlily_jl_amd import slab
    from waterlily_jl_amd._lib import check, lib
    L = lib()
    if mode == "cpu_halo":
        comm = slab.CallbackComm(dist, host_buffers=True)
        gd = (10, 7, 2 + 6 * size)
        g = slab.slab_grid(gd, rank, size, halo=2)
        assert (g.nz, g.k0, g.k1, g.gnz) == (6 + 4, 2, 8, gd[2]) and g.gk == 1 + rank * 6 - 2
        ncomp = 3
        # global field f(x,y,K,c) = unique number; local slab filled on owned planes only, halos = -1
        a = np.full((g.nx, g.ny, g.nz, ncomp), -1.0, dtype=np.float32, order="F")
        val = lambda K, c: (np.arange(g.nx)[:, None] + 100 * np.arange(g.ny)[None, :] + 10000 * K + 1e6 * c).astype(np.float32)
        for k in range(g.k0, g.k1):
            for c in range(ncomp):
                a[:, :, k, c] = val(g.gk + k, c)
        for depth in (1, 2):
            b = a.copy(order="F")
            check(L.wl_halo_exchange(comm.handle, b.ctypes.data_as(C.c_void_p), C.byref(g), ncomp, depth, None))
            for c in range(ncomp):
                for d in range(1, depth + 1):
                    lo, hi = g.k0 - d, g.k1 + d - 1
                    if rank > 0:
                        assert np.array_equal(b[:, :, lo, c], val(g.gk + lo, c)), ("lo", depth, d)
                    else:
                        assert np.all(b[:, :, lo, c] == -1)
                    if rank < size - 1:
                        assert np.array_equal(b[:, :, hi, c], val(g.gk + hi, c)), ("hi", depth, d)
                    else:
                        assert np.all(b[:, :, hi, c] == -1)
                if depth == 1:   # the outer ghost plane must be untouched by a depth-1 exchange
                    assert np.all(b[:, :, g.k0 - 2, c] == -1) and np.all(b[:, :, g.k1 + 1, c] == -1)
            assert np.array_equal(b[:, :, g.k0:g.k1], a[:, :, g.k0:g.k1])
        # all-gather of the planes each rank computed of a replicated array
        nc = 3
        from waterlily_jl_amd._lib import wl_grid
        view = wl_grid(); view.D = 3; view.nx, view.ny = 6, 5; view.gnz = view.nz = 2 + nc * size; view.gk = 0
        view.k0, view.k1 = 1 + rank * nc, 1 + (rank + 1) * nc
        full = np.zeros((6, 5, view.nz, 2), dtype=np.float32, order="F")
        for k in range(view.k0, view.k1):
            full[:, :, k, :] = 1000 * k + rank + 1
        check(L.wl_allgather_planes(comm.handle, full.ctypes.data_as(C.c_void_p), C.byref(view), 2, None))
        for r in range(size):
            for k in range(1 + r * nc, 1 + (r + 1) * nc):
                assert np.all(full[:, :, k, :] == 1000 * k + r + 1)
        assert np.all(full[:, :, 0, :] == 0) and np.all(full[:, :, -1, :] == 0)
        # the communicator's counters (wl_comm_stats): two exchanges (depth 1 and 2) of ncomp planes to each existing neighbour, two gathers
        st = slab.comm_stats(comm)
        nb = (1 if rank > 0 else 0) + (1 if rank < size - 1 else 0)
        plane = g.nx * g.ny * 4
        assert st["halo_exchanges"] == 2 and st["plane_allgathers"] == 2 and st["scalar_combines"] == 0, st
        assert st["halo_bytes_sent"] == (1 + 2) * plane * ncomp * nb, st
        comm.destroy()
        print(f"rank {rank}: cpu_halo ok", flush=True)
    elif mode == "cpu_halo_periodic":
        # z-periodic domain: the exchange wraps around (rank 0's lower ghost planes are the last rank's top planes and vice versa)
        comm = slab.CallbackComm(dist, host_buffers=True)
        check(L.wl_comm_set_periodic(comm.handle, 1))
        nloc, ncomp = 6, 2
        gd = (10, 7, 2 + nloc * size)
        g = slab.slab_grid(gd, rank, size, halo=3)
        a = np.full((g.nx, g.ny, g.nz, ncomp), -1.0, dtype=np.float32, order="F")
        val = lambda K, c: (np.arange(g.nx)[:, None] + 100 * np.arange(g.ny)[None, :] + 10000 * K + 1e6 * c).astype(np.float32)
        wrap = lambda K: (K - 1) % (nloc * size) + 1            # global interior plane a (ghost) plane index is the periodic image of
        for k in range(g.k0, g.k1):
            for c in range(ncomp):
                a[:, :, k, c] = val(g.gk + k, c)
        for depth in (1, 2, 3):
            b = a.copy(order="F")
            check(L.wl_halo_exchange(comm.handle, b.ctypes.data_as(C.c_void_p), C.byref(g), ncomp, depth, None))
            for c in range(ncomp):
                for d in range(1, depth + 1):
                    lo, hi = g.k0 - d, g.k1 + d - 1
                    assert np.array_equal(b[:, :, lo, c], val(wrap(g.gk + lo), c)), ("lo", rank, depth, d)
                    assert np.array_equal(b[:, :, hi, c], val(wrap(g.gk + hi), c)), ("hi", rank, depth, d)
            assert np.array_equal(b[:, :, g.k0:g.k1], a[:, :, g.k0:g.k1])
        comm.destroy()
        print(f"rank {rank}: cpu_halo_periodic ok", flush=True)
    elif mode == "gpu_sim":
        torch.cuda.set_device(0)
        dims = tuple(int(v) for v in sys.argv[2].split("x"))
        steps = int(sys.argv[3])
        comm = slab.CallbackComm(dist)
        nu = dims[0] / 1600.0
        sim = slab.SlabSimulation(comm, dims, (0, 0, 0), dims[0], U=1, nu=nu, ic="tgv")
        ref = w.FusedSimulation(dims, (0, 0, 0), dims[0], U=1, nu=nu, ic="tgv") if rank == 0 else None
        for kv in filter(None, os.environ.get("WL_SLAB_OPTS", "").split(",")):      # e.g. "resjac_min=0": size gates lowered for small test boxes (process-wide switches)
            k, v = kv.split("=")
            check(L.wl_sim_set_option(sim._h, k.encode(), int(v)))
        mg = L.wl_sim_pois(sim._h)
        kinds = [L.wl_mg_smoother_kind(mg, l) for l in range(L.wl_mg_nlevels(mg))]
        if rank == 0:
            print("smoother kinds per level (slab run):", kinds, flush=True)
        if dims[0] >= 64 and dims[1] >= 32 and dims[2] // size >= 8:
            assert kinds[0] == 2, kinds      # the blocked pair kernels run on the distributed finest level
        for s in range(steps):
            sim.mom_step_()
            u = sim.gather_field("u", dist)
            p = sim.gather_field("p", dist)
            if rank == 0:
                ref.mom_step_()
                ur, pr = ref.field("u"), ref.field("p")
                du, dp = np.abs(u - ur).max(), np.abs(p - pr).max()
                print(f"step {s}: max|du|={du:.3e} max|dp|={dp:.3e} n_slab={sim.pois_n[-2:]} n_ref={ref.pois_n[-2:]} dt={sim.dt[-1]:.6f}/{ref.dt[-1]:.6f}", flush=True)
                assert sim.pois_n == ref.pois_n
                assert abs(float(sim.dt[-1]) - float(ref.dt[-1])) <= 1e-6 * float(ref.dt[-1])
                assert du < 2e-5 and dp < 2e-4, (du, dp)   # only the reductions' association order differs
        cnt = C.c_long()
        check(L.wl_sim_counter(sim._h, b"resjac", C.byref(cnt)))
        print(f"rank {rank}: fused projection heads on the slab: {cnt.value}", flush=True)
        dist.barrier()
        del sim
        comm.destroy()
        print(f"rank {rank}: gpu_sim ok", flush=True)
    elif mode == "gpu_per":
        # periodic directions on z-slabs: the periodic TGV (κ = 2π/N) with perdir given as digits ("12": x and y; "123": all three)
        torch.cuda.set_device(0)
        dims = tuple(int(v) for v in sys.argv[2].split("x"))
        steps = int(sys.argv[3])
        perdir = tuple(int(c) for c in sys.argv[4])
        comm = slab.CallbackComm(dist)
        nu = dims[0] / 1600.0
        sim = slab.SlabSimulation(comm, dims, (0, 0, 0), dims[0], U=1, nu=nu, ic="tgv_periodic", perdir=perdir)
        ref = w.FusedSimulation(dims, (0, 0, 0), dims[0], U=1, nu=nu, ic="tgv_periodic", perdir=perdir) if rank == 0 else None
        for s in range(steps):
            sim.mom_step_()
            u = sim.gather_field("u", dist)
            p = sim.gather_field("p", dist)
            if os.environ.get("WL_PER_DEBUG") and s == 0:
                m0 = sim.gather_field("mu0", dist)
                if rank == 0:
                    print("  mu0_z slab planes 0,1,2,N-2,N-1:", [float(m0[3, 3, k, 2]) for k in (0, 1, 2, -2, -1)], flush=True)
            if rank == 0:
                ref.mom_step_()
                ur, pr = ref.field("u"), ref.field("p")
                ins = (slice(1, -1),) * 3
                du, dp = np.abs(u[ins] - ur[ins]).max(), np.abs(p[ins] - pr[ins]).max()
                print(f"step {s}: max|du|={du:.3e} max|dp|={dp:.3e} n_slab={sim.pois_n[-2:]} n_ref={ref.pois_n[-2:]} dt={sim.dt[-1]:.6f}/{ref.dt[-1]:.6f}", flush=True)
                if os.environ.get("WL_PER_DEBUG") and s == 0:
                    m0r = ref.field("mu0")
                    print("  mu0_z ref planes 0,1,2,N-2,N-1:", [float(m0r[3, 3, k, 2]) for k in (0, 1, 2, -2, -1)], flush=True)
                    for c in range(3):
                        prof = np.abs(u[ins][..., c] - ur[ins][..., c]).max(axis=(0, 1))
                        print(f"  |du_{c}| per interior plane:", " ".join(f"{v:.0e}" for v in prof), flush=True)
                    print("  |dp| per interior plane:", " ".join(f"{v:.0e}" for v in np.abs(p[ins] - pr[ins]).max(axis=(0, 1))), flush=True)
                assert sim.pois_n == ref.pois_n
                assert abs(float(sim.dt[-1]) - float(ref.dt[-1])) <= 1e-6 * float(ref.dt[-1])
                assert du < 2e-5 and dp < 2e-4, (du, dp)
        dist.barrier()
        del sim
        comm.destroy()
        print(f"rank {rank}: gpu_per ok", flush=True)
    elif mode == "gpu_per_body":
        # immersed sphere in a stream along x, periodic in y and z, on z-slabs (the sphere sits on a slab boundary: μ₁, V and f cross it; the
        # z-periodic wrap carries the wake's images)
        torch.cuda.set_device(0)
        dims = tuple(int(v) for v in sys.argv[2].split("x"))
        steps = int(sys.argv[3])
        comm = slab.CallbackComm(dist)
        R, c = dims[1] / 8.0, (dims[0] / 4.0, dims[1] / 2.0 - 1, dims[2] / 2.0 - 1)
        nu = 2 * R / 250.0
        sim = slab.SlabSimulation(comm, dims, (1.0, 0.0, 0.0), 2 * R, U=1, nu=nu, has_body=True, perdir=(2, 3))
        sim.measure_sphere_(c, R)
        ref = None
        if rank == 0:
            ref = w.FusedSimulation(dims, (1.0, 0.0, 0.0), 2 * R, U=1, nu=nu, has_body=True, perdir=(2, 3))
            ref.measure_sphere_(c, R)
        for s in range(steps):
            sim.mom_step_()
            u = sim.gather_field("u", dist)
            if rank == 0:
                ref.mom_step_()
                ins = (slice(1, -1),) * 3
                du = np.abs(u[ins] - ref.field("u")[ins]).max()
                print(f"step {s}: max|du|={du:.3e} n_slab={sim.pois_n[-2:]} n_ref={ref.pois_n[-2:]} dt={sim.dt[-1]:.6f}/{ref.dt[-1]:.6f}", flush=True)
                assert sim.pois_n == ref.pois_n
                assert du < 5e-5, du
        dist.barrier()
        del sim
        comm.destroy()
        print(f"rank {rank}: gpu_per_body ok", flush=True)
    elif mode == "gpu_exit":
        # convective exit (exitBC!) + immersed sphere on slabs: the x-exit face is shared by all ranks, its means are global
        torch.cuda.set_device(0)
        dims = tuple(int(v) for v in sys.argv[2].split("x"))
        steps = int(sys.argv[3])
        comm = slab.CallbackComm(dist)
        R, c = dims[1] / 8.0, (dims[0] / 4.0, dims[1] / 2.0 - 1, dims[2] / 2.0 - 1)
        nu = 2 * R / 250.0
        sim = slab.SlabSimulation(comm, dims, (1.0, 0.0, 0.0), 2 * R, U=1, nu=nu, has_body=True, exitBC=True)
        sim.measure_sphere_(c, R)
        ref = None
        if rank == 0:
            ref = w.FusedSimulation(dims, (1.0, 0.0, 0.0), 2 * R, U=1, nu=nu, has_body=True, exitBC=True)
            ref.measure_sphere_(c, R)
        for s in range(steps):
            sim.mom_step_()
            u = sim.gather_field("u", dist)
            if rank == 0:
                ref.mom_step_()
                du = np.abs(u - ref.field("u")).max()
                print(f"step {s}: max|du|={du:.3e} n_slab={sim.pois_n[-2:]} n_ref={ref.pois_n[-2:]}", flush=True)
                assert sim.pois_n == ref.pois_n
                assert du < 5e-5, du
        # force read-outs are collective on slabs: every rank sums its planes, the sums are added on device
        body = ("sphere", c, R)
        fp, fv = sim.pressure_force_body(body), sim.viscous_force_body(body)
        if rank == 0:
            rp, rv = ref.pressure_force_body(body), ref.viscous_force_body(body)
            print(f"forces: slab p={fp} v={fv}  single p={rp} v={rv}", flush=True)
            assert np.abs(rp).max() > 0 and np.allclose(fp, rp, rtol=1e-3, atol=1e-3 * np.abs(rp).max())
            assert np.abs(rv).max() > 0 and np.allclose(fv, rv, rtol=1e-3, atol=1e-3 * np.abs(rv).max())
        dist.barrier()
        del sim
        comm.destroy()
        print(f"rank {rank}: gpu_exit ok", flush=True)
    elif mode == "gpu_rccl1":
        # RCCL transport on the ranks one box offers (1).  Multi-rank RCCL cannot run here (RCCL refuses two ranks on one device),
        # so this covers what one rank can: dlopen of librccl.so.1 shared with torch, the availability agreement, unique-id
        # broadcast, ncclCommInitRank of BOTH communicators, and — through the one-rank loopback test mode (both neighbours are
        # this rank) — the ncclSend/ncclRecv groups of the plane exchange on the compute stream and on the communicator's own
        # stream, the in-place ncclAllGather and the device-side scalar combine.  A slab step with >1 ranks over RCCL stays unverified.
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        comm = slab.RcclComm(dist, dev)
        from waterlily_jl_amd._lib import wl_grid
        view = wl_grid(); view.D = 3; view.nx, view.ny = 6, 5; view.gnz = view.nz = 5; view.gk = 0; view.k0, view.k1 = 1, 4
        t = torch.arange(6 * 5 * 5, dtype=torch.float32, device=dev)
        before = t.clone()
        check(L.wl_allgather_planes(comm.handle, C.c_void_p(t.data_ptr()), C.byref(view), 1, None))     # size 1, no loopback: early return
        torch.cuda.synchronize()
        assert torch.equal(t, before)
        sim = slab.SlabSimulation(comm, (32, 32, 32), (0, 0, 0), 32, U=1, nu=0.02, ic="tgv")
        ref = w.FusedSimulation((32, 32, 32), (0, 0, 0), 32, U=1, nu=0.02, ic="tgv")
        sim.mom_step_(); ref.mom_step_()
        assert np.array_equal(sim.local_field("u"), ref.field("u")) and sim.pois_n == ref.pois_n
        del sim
        # ---- loopback: the NCCL calls themselves execute
        check(L.wl_comm_set_loopback(comm.handle, 1))
        g = slab.slab_grid((18, 10, 2 + 12), 0, 2, halo=2)              # a slab that is NOT the whole domain (nz < gnz)
        ncomp = 3
        for use_async in (0, 1):
            for depth in (1, 2):
                a = torch.full((ncomp, g.nz, g.ny, g.nx), -1.0, dtype=torch.float32, device=dev)    # memory order: x fastest, component slowest
                for c in range(ncomp):
                    for k in range(g.k0, g.k1):
                        a[c, k] = 1000.0 * c + k
                b = a.clone()
                fn = L.wl_comm_halo_async if use_async else L.wl_halo_exchange
                check(fn(comm.handle, C.c_void_p(b.data_ptr()), C.byref(g), ncomp, depth, None))
                check(L.wl_stream_sync(None)); torch.cuda.synchronize()
                for c in range(ncomp):
                    for d in range(1, depth + 1):
                        # periodic wrap onto itself: lower ghost planes = own top planes, upper ghost planes = own bottom planes
                        assert torch.all(b[c, g.k0 - d] == 1000.0 * c + (g.k1 - d)), (use_async, depth, c, d)
                        assert torch.all(b[c, g.k1 + d - 1] == 1000.0 * c + (g.k0 + d - 1)), (use_async, depth, c, d)
                    if depth == 1:
                        assert torch.all(b[c, g.k0 - 2] == -1) and torch.all(b[c, g.k1 + 1] == -1)
                assert torch.equal(b[:, g.k0:g.k1], a[:, g.k0:g.k1])
        t = torch.arange(6 * 5 * 5, dtype=torch.float32, device=dev)
        check(L.wl_allgather_planes(comm.handle, C.c_void_p(t.data_ptr()), C.byref(view), 1, None))     # in-place ncclAllGather, 1 rank
        torch.cuda.synchronize()
        assert torch.equal(t, before)
        rec = torch.zeros(32, dtype=torch.float32, device=dev)           # 128-byte record: 8 doubles then 8 floats
        recd = rec[:16].view(torch.float64); recf = rec[16:24]
        recd.copy_(torch.arange(1, 9, dtype=torch.float64)); recf.copy_(torch.arange(-3, 5, dtype=torch.float32))
        check(L.wl_comm_combine_test(comm.handle, C.c_void_p(rec.data_ptr()), C.c_void_p(rec.data_ptr() + 64), None))
        check(L.wl_stream_sync(None)); torch.cuda.synchronize()
        assert torch.equal(recd.cpu(), torch.arange(1, 9, dtype=torch.float64)) and torch.equal(recf.cpu(), torch.arange(-3, 5, dtype=torch.float32))
        st = slab.comm_stats(comm)
        assert st["halo_exchanges"] >= 4 and st["scalar_combines"] >= 1 and st["plane_allgathers"] >= 1, st
        comm.destroy()
        print(f"rank {rank}: gpu_rccl1 ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
