"""BASELINE.json's configurations by name.
configs[0] (CPU, here): the reference README's 2-D circle (README.md:41-57) — 3·2^5 × 2^6, Re=100, Float64, serial, 10 sim_step! —
  on the oracle: the plumbing case; its 6-level semi-coarsened multigrid hierarchy is SURVEY §8's (98,66)…(5,4).
configs[0] (GPU): the same workload in Float32 through the 2-D HIP kernels against the oracle in Float32 (VERDICT r02 #6a).
configs[3] (GPU): sphere (AutoBody sdf) 256³ Re=3700 Float32 — default kernels vs the one-kernel-per-pass general kernels bitwise,
  pressure_force of both paths equal; and the same flow at 64³ against the oracle with the z-split smoother forced on.
The drag history of configs[3] has no reference-held value (no Julia here, none in the reference's tests): PARITY UNPINNED for
that number; pressure_force itself is pinned by test/test_metrics.jl:35-40 and test/test_flow.jl:161-173 (tests/test_oracle_kat.py)."""
import numpy as np
import pytest


def test_config0_circle_2d_float64_on_the_oracle(oracle):
    n, m = 3 * 2**5, 2**6
    radius, center = m / 8, m / 2 - 1
    Re, U = 100, 1
    sim = oracle.Simulation((n, m), (U, 0), 2 * radius, U=U, nu=U * 2 * radius / Re, body=("sphere", (center, center), radius), T=np.float64)
    # MultiLevelPoisson: levels are halved while every dimension is divisible (semi-coarsening at the end), src/MultiLevelPoisson.jl:20-48
    assert sim.nlevels == 6
    assert [sim.level_dims(l) for l in range(6)] == [(98, 66), (50, 34), (26, 18), (14, 10), (8, 6), (5, 4)]
    for _ in range(10):                      # sim_step!(circ) ten times (remeasure=true: a static body is re-measured to the same μ)
        sim.step(remeasure=True)
    pn = sim.pois_n
    assert len(pn) == 20 and len(sim.dt) == 11
    assert all(1 <= v <= 32 for v in pn) and max(pn[2:]) <= 6, pn         # impulsive start may take more V-cycles than the later solves
    u = sim.u
    assert np.isfinite(u).all() and np.isfinite(sim.p).all()
    # the flow the README describes after the first steps: ≈0 inside the circle, ≈U far upstream, accelerated beside the circle
    ic, jc = int(center) + 1, int(center) + 1
    assert abs(u[ic, jc, 0]) < 0.05
    assert abs(u[3, jc, 0] - 1.0) < 0.05
    assert u[ic, jc + int(radius) + 3, 0] > 1.1
    assert all(0 < d <= 10 for d in sim.dt) and sim.dt[-1] < 0.5
    # the projection leaves a divergence-free field (to the solver tolerance) on the interior
    div = (u[2:, 1:-1, 0] - u[1:-1, 1:-1, 0]) + (u[1:-1, 2:, 1] - u[1:-1, 1:-1, 1])      # div(I) = Σ u[I+δ,i] − u[I,i], src/Flow.jl:13-19
    assert np.abs(div).max() < 1e-2
    f = sim.pressure_force()
    # pressure_force = Σ p·n·kern (src/Metrics.jl:124-133): n points out of the body, so a drag in +x shows as a NEGATIVE x-component
    # (the reference's own accelerating-circle test expects −1, test/test_flow.jl:167); ≈ symmetric in y
    assert np.isfinite(f).all() and f[0] < 0 and abs(f[1]) < 0.2 * abs(f[0])


@pytest.fixture(scope="module")
def w():
    import waterlily_jl_amd as w
    w.core.device()
    return w


PLAIN = {"zsplit": 0, "farmask": 0, "hybrid": 0, "constl": 0}


@pytest.mark.gpu
def test_config3_sphere_256_default_equals_general_kernels(w):
    N = 256
    R, c = N / 8, (N / 2 - 1,) * 3
    res = {}
    for tag, opts in (("default", {}), ("plain", PLAIN)):
        sim = w.FusedSimulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=2 * R / 3700, has_body=True)
        for k, v in opts.items():
            sim.set_option(k, v)
        sim.measure_sphere_(c, R, 1.0)
        for _ in range(3):
            sim.mom_step_()
        res[tag] = (sim.field("u"), sim.field("p"), sim.pois_n, sim.dt, sim.pressure_force_sphere(c, R), sim.smoother_kinds()[0])
        del sim
    d, p = res["default"], res["plain"]
    assert d[5] == 3 and p[5] == 1                     # the z-split smoother is the default at this size; the plain run uses the general blocked kernels
    assert d[2] == p[2] and d[3] == p[3] and max(d[2]) <= 5, d[2]
    assert np.isfinite(d[0]).all() and np.isfinite(d[1]).all()
    assert np.array_equal(d[0], p[0]) and np.array_equal(d[1], p[1])
    assert np.array_equal(d[4], p[4]) and d[4][0] != 0 and np.isfinite(d[4]).all()      # pressure_force of the two paths


@pytest.mark.gpu
def test_config3_sphere_64_matches_oracle_with_zsplit(w, oracle):
    N = 64
    R, c = N / 8, (N / 2 - 1,) * 3
    nu = 2 * R / 3700
    so = oracle.Simulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=nu, body=("sphere", c, R), T=np.float32)
    sg = w.FusedSimulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=nu, has_body=True)
    sg.set_option("zsplit", 2)                         # force the far/near plane split on a level far below its size gate
    sg.measure_sphere_(c, R, 1.0)
    sg.set_field("mu0", so.field("mu0")); sg.set_field("mu1", so.field("mu1")); sg.update_()     # identical coefficients: the comparison is about the step
    assert sg.smoother_kinds()[0] == 3
    for step in range(3):
        so.step(remeasure=False); sg.mom_step_()
        assert sg.pois_n == so.pois_n and max(sg.pois_n) <= 5
        assert np.abs(sg.field("u") - so.u).max() < 5e-5           # reductions (mean shift, norms, CFL) differ in summation order only
        assert np.abs(sg.field("p") - so.p).max() < 5e-4
    fo, fg = so.pressure_force(), sg.pressure_force_sphere(c, R)
    assert np.allclose(fg, fo, rtol=2e-3, atol=2e-3 * np.abs(fo).max())


@pytest.mark.gpu
def test_config0_circle_2d_on_the_hip_path_matches_oracle(w, oracle):
    """configs[0]'s workload — circle 3·2^5 × 2^6, Re=100, ten sim_step! with remeasure on (README.md:41-57) — through the 2-D HIP kernels
    (wl_sim, D = 2) in Float32 against the oracle in Float32: same multigrid hierarchy, same pois.n, |Δu| ≤ 1e-5 (reductions differ in
    summation order only), Δt to rounding."""
    n, m = 3 * 2**5, 2**6
    radius, center = m / 8, m / 2 - 1
    Re, U = 100, 1
    nu = U * 2 * radius / Re
    so = oracle.Simulation((n, m), (U, 0), 2 * radius, U=U, nu=nu, body=("sphere", (center, center), radius), T=np.float32)
    sg = w.FusedSimulation((n, m), (U, 0), 2 * radius, U=U, nu=nu, has_body=True)
    sg.measure_sphere_((center, center), radius, 1.0)
    # MultiLevelPoisson's 6-level hierarchy, semi-coarsened at the end (src/MultiLevelPoisson.jl:20-48,70), read back from the device handle
    want = [(98, 66), (50, 34), (26, 18), (14, 10), (8, 6), (5, 4)]
    assert sg.nlevels() == so.nlevels == 6
    from waterlily_jl_amd._lib import lib, wl_grid
    import ctypes as C
    got = []
    for l in range(6):
        g = wl_grid()
        assert lib().wl_mg_level_grid(lib().wl_sim_pois(sg._h), l, C.byref(g)) == 0
        assert g.D == 2
        got.append((g.nx, g.ny))
    assert got == want == [so.level_dims(l) for l in range(6)]
    # the measured coefficients agree to rounding (device measure! evaluates the same closed form in Float32)
    assert np.abs(sg.field("mu0") - so.field("mu0")).max() < 2e-6
    for step in range(10):
        so.step(remeasure=True)
        sg.measure_sphere_((center, center), radius, 1.0)       # sim_step!(remeasure=true): measure!(sim) + update!(pois), src/WaterLily.jl:136-149
        sg.mom_step_()
        assert sg.pois_n == so.pois_n, (step, sg.pois_n, so.pois_n)
        du = np.abs(sg.field("u") - so.u).max()
        assert du <= 1e-5, (step, du)
        assert np.abs(sg.field("p") - so.p).max() <= 2e-4 * max(1.0, np.abs(so.p).max())
    assert len(sg.pois_n) == 20 and max(sg.pois_n[2:]) <= 6
    dg, do = np.array(sg.dt, dtype=np.float64), np.array(so.dt, dtype=np.float64)
    assert dg.shape == do.shape == (11,) and np.abs(dg / do - 1).max() < 1e-5
    fo, fg = so.pressure_force(), sg.pressure_force_sphere((center, center), radius)
    assert fg[0] < 0 and np.allclose(fg, fo, rtol=2e-3, atol=2e-3 * np.abs(fo).max())


@pytest.mark.gpu
def test_config3_sphere_256_matches_the_oracle_at_full_size(w, oracle):
    """configs[3] at its own size against the oracle (OpenMP build on the box's host cores; ≈0.5 s per step): sphere 256³ Re=3700, identical measured
    coefficients, two mom_step! through the DEFAULT body path (z-split smoother, tiled conv_diff! on the body-free planes, split head/tails): same pois.n,
    |Δu| ≤ 5e-5, pressure_force to 2e-3 — the drag number itself has no reference-held value (parity unpinned, see the module docstring)."""
    N = 256
    R, c = N / 8, (N / 2 - 1,) * 3
    nu = 2 * R / 3700
    so = oracle.Simulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=nu, body=("sphere", c, R), T=np.float32, omp=True)
    sg = w.FusedSimulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=nu, has_body=True)
    sg.measure_sphere_(c, R, 1.0)
    assert np.abs(sg.field("mu0") - so.field("mu0")).max() < 2e-6       # device measure! = the oracle's to rounding; then made identical
    sg.set_field("mu0", so.field("mu0")); sg.set_field("mu1", so.field("mu1")); sg.update_()
    assert sg.smoother_kinds()[0] == 3
    for step in range(2):
        so.step(remeasure=False); sg.mom_step_()
        assert sg.pois_n == so.pois_n and max(sg.pois_n) <= 6, (step, sg.pois_n, so.pois_n)
        du = float(np.abs(sg.field("u") - so.u).max())
        assert du <= 5e-5, (step, du)
    fo, fg = so.pressure_force(), sg.pressure_force_sphere(c, R)
    assert np.allclose(fg, fo, rtol=2e-3, atol=2e-3 * np.abs(fo).max())
