"""Solver- and step-level parity of the HIP path (through the C ABI) against the oracle, including the
reference's own known-answer tests re-run on the GPU path (file:line given per test)."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def F(shape, fill=0.0):
    return np.full(shape, fill, dtype=np.float32, order="F")


@pytest.fixture(scope="module")
def w():
    import waterlily_jl_amd as w
    w.core.device()
    return w


def poisson_setup_gpu(w, oracle, N):
    """Poisson_setup(MultiLevelPoisson,N)   test/test_poisson.jl:1-12 on the HIP path"""
    D = len(N)
    c = F(N + (D,), 1.0)
    cg = w.to_device(c)
    w.BC_(cg, (0,) * D)
    xg, zg = w.jl_zeros(N), w.jl_zeros(N)
    pois = w.MultiLevelPoisson(xg, cg, zg)
    soln = np.asfortranarray(np.broadcast_to((np.arange(N[0], dtype=np.float32) + 1).reshape((N[0],) + (1,) * (D - 1)), N).copy(order="F"))
    I = (1,) * D
    soln -= soln[I]
    # z = mult!(pois, soln)
    lvl0 = w.Poisson.__new__(w.Poisson)
    import ctypes as C
    g = w.core.sgrid(xg)
    D0 = pois.levels[0]
    lib = w.lib()
    sd = w.to_device(soln)
    w._lib.check(lib.wl_mult(w.core.ptr(zg), w.core.ptr(cg), lib.wl_mg_level_field(pois._h, 0, b"D"), w.core.ptr(sd), C.byref(g), w.core.stream()))
    n = pois.solver_()
    x = w.to_host(xg)
    x -= x[I]
    err = oracle.L2(x - soln) / oracle.L2(soln)
    return err, pois, n


# test/test_poisson.jl:54-60
def test_multilevel_coarse_diagonal_and_update(w, oracle):
    err, pois, _ = poisson_setup_gpu(w, oracle, (10, 10))
    D3 = np.array([[0, 0, 0, 0], [0, -2, -2, 0], [0, -2, -2, 0], [0, 0, 0, 0]], dtype=np.float32)
    assert np.array_equal(pois.levels[2].D, D3)
    assert err < 1e-5
    pois.L[4:6, :, 0] = 0
    pois.update_()
    assert np.array_equal(pois.levels[2].D, D3 / 2)


# test/test_poisson.jl:42  (the @assert of src/MultiLevelPoisson.jl:73-74)
def test_too_few_levels_raises(w, oracle):
    with pytest.raises(AssertionError, match="MultiLevelPoisson requires size=a2ⁿ, where n>2"):
        poisson_setup_gpu(w, oracle, (15 + 2, 3**4 + 2))


# test/test_poisson.jl:62-70
def test_multigrid_convergence(w, oracle):
    err, pois, n = poisson_setup_gpu(w, oracle, (2**6 + 2, 2**6 + 2))
    assert err < 1e-6 and n <= 4 and pois.n[-1] == n
    assert pois.level_norms(0)[1] < 2e-3
    err, pois, n = poisson_setup_gpu(w, oracle, (2**4 + 2,) * 3)
    assert err < 1e-6 and n <= 3


@pytest.mark.parametrize("N", [(18, 18), (34, 34, 34), (66, 18, 10), (130, 18)])
def test_solver_matches_oracle_per_iteration(w, oracle, N):
    """same random problem on both paths: identical level hierarchy, per-iteration L₁/L∞/ω log within
    f32 reduction tolerance, solution within 1e-5."""
    rng = np.random.default_rng(31)
    D = len(N)
    L = np.asfortranarray(rng.uniform(0.2, 1.0, size=N + (D,)).astype(np.float32))
    oracle.BC(L, (0,) * D)
    x0 = np.asfortranarray(rng.uniform(-1, 1, size=N).astype(np.float32))
    sl = tuple(slice(1, -1) for _ in range(D))
    z = F(N)
    zz = rng.uniform(-1, 1, size=tuple(n - 2 for n in N)).astype(np.float32)
    z[sl] = zz - zz.mean()
    xo, Lo, zo = x0.copy(order="F"), L.copy(order="F"), z.copy(order="F")
    po = oracle.MultiLevelPoisson(xo, Lo, zo)
    xg, Lg, zg = w.to_device(x0), w.to_device(L), w.to_device(z)
    pg = w.MultiLevelPoisson(xg, Lg, zg)
    assert len(pg.levels) == po.nlevels
    for l in range(po.nlevels):
        assert pg.levels[l].dims == po.level_dims(l)
        assert np.array_equal(pg.levels[l].L, po.field("L", l)), l
        assert np.array_equal(pg.levels[l].D, po.field("D", l)), l
        assert np.array_equal(pg.levels[l].iD, po.field("iD", l)), l
    no, ng = po.solve(), pg.solver_()
    r1o, rio, wo = po.log()
    r1g, rig, wg = pg.log()
    assert ng == no
    assert np.allclose(r1g, r1o, rtol=2e-4) and np.allclose(rig, rio, rtol=2e-3, atol=1e-6) and np.array_equal(wg, wo)
    assert np.allclose(w.to_host(xg), xo, rtol=0, atol=1e-5 * max(1.0, np.abs(xo).max()))


def test_vcycle_matches_oracle_on_every_level(w, oracle):
    rng = np.random.default_rng(37)
    N = (34, 18, 18)
    D = 3
    L = np.asfortranarray(rng.uniform(0.2, 1.0, size=N + (D,)).astype(np.float32))
    oracle.BC(L, (0,) * D)
    x0 = F(N)
    z = F(N)
    r0 = F(N)
    r0[1:-1, 1:-1, 1:-1] = rng.uniform(-1, 1, size=(32, 16, 16)).astype(np.float32)
    po = oracle.MultiLevelPoisson(x0.copy(order="F"), L.copy(order="F"), z.copy(order="F"))
    xg, Lg, zg = w.to_device(x0), w.to_device(L), w.to_device(z)
    pg = w.MultiLevelPoisson(xg, Lg, zg)
    po.field("r", 0)[...] = r0
    import ctypes as C
    lib = w.lib()
    w._lib.check(lib.wl_h2d(lib.wl_mg_level_field(pg._h, 0, b"r"), r0.ctypes.data_as(C.c_void_p), r0.nbytes, w.core.stream()))
    po.Vcycle(0, 0.9); pg.Vcycle_(0, 0.9)
    for l in range(po.nlevels):
        for name in ("r", "x"):
            assert np.array_equal(getattr(pg.levels[l], name), po.field(name, l)), (l, name)


def tgv3d(N):
    kap = math.pi / N
    return lambda i, x: (-math.sin(kap * x[0]) * math.cos(kap * x[1]) * math.cos(kap * x[2]) if i == 1 else
                         (math.cos(kap * x[0]) * math.sin(kap * x[1]) * math.cos(kap * x[2]) if i == 2 else 0.0))


@pytest.mark.parametrize("fused", [True, False])
def test_tgv_steps_match_oracle(w, oracle, fused):
    """3-D wall-bounded TGV at 32³: fields after 1..3 mom_step! against the oracle.
    Tolerance: 2e-5 absolute on u (|u|≤1) and 2e-4 on p — f32 reductions decide the mean shift of r and
    nothing else differs; Δt and pois.n must agree exactly/within 1 ulp."""
    N = 32
    nu = N / 1600.0
    so = oracle.Simulation((N, N, N), (0, 0, 0), N, U=1, nu=nu, u0=tgv3d(N), T=np.float32)
    u_init = so.u.copy(order="F")
    if fused:
        sg = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=nu, u0=u_init)
    else:
        sg = w.Simulation((N, N, N), (0, 0, 0), N, U=1, nu=nu, u0=u_init)
    for step in range(3):
        so.step(remeasure=False)
        if fused:
            sg.mom_step_()
            ug, pgp, dtg, ng = sg.field("u"), sg.field("p"), sg.dt, sg.pois_n
        else:
            sg.sim_step_(remeasure=False)
            ug, pgp, dtg, ng = w.to_host(sg.flow.u), w.to_host(sg.flow.p), sg.flow.dt, sg.pois.n
        assert ng == so.pois_n
        assert np.allclose(np.array(dtg, dtype=np.float64), np.array(so.dt), rtol=1e-6)
        assert np.abs(ug - so.u).max() < 2e-5, step
        assert np.abs(pgp - so.p).max() < 2e-4, step


def test_fused_phases_match_oracle(w, oracle):
    N = 16
    so = oracle.Simulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, u0=tgv3d(N), T=np.float32)
    sg = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, u0=so.u.copy(order="F"))
    sg.set_option("store_f", 1)      # materialise the intermediates f and z (the time-step path itself never reads them again)
    for ph in range(6):
        so.phase(ph); sg.phase_(ph)
        tol = 0 if ph in (0, 1, 3) else 2e-5
        for name in ("u", "f"):
            d = np.abs(sg.field(name) - so.field(name)).max()
            assert d <= tol, (ph, name, d)
    assert np.allclose(sg.dt[-1], so.dt[-1], rtol=1e-6)


def test_periodic_tgv_2d_runs_and_matches(w, oracle):
    """periodic 2-D TGV as in test/helper.jl:4-15 but with the initial field as u0 and tuple BCs (all directions
    periodic, so uBC is never evaluated; Re=1e8 makes dU/dt negligible): 64², tU/L = π/100."""
    Lg = 64
    kap = 2 * math.pi / Lg
    nu = 1 / (kap * 1e8)
    ic = lambda i, x: -math.sin(kap * x[0]) * math.cos(kap * x[1]) if i == 1 else math.cos(kap * x[0]) * math.sin(kap * x[1])
    so = oracle.Simulation((Lg, Lg), (0, 0), Lg, U=1, nu=nu, u0=ic, perdir=(1, 2), T=np.float32)
    sg = w.FusedSimulation((Lg, Lg), (0, 0), Lg, U=1, nu=nu, perdir=(1, 2), u0=so.u.copy(order="F"))
    so.step_until(math.pi / 100, remeasure=False)
    sg.sim_step_(math.pi / 100)
    assert len(sg.dt) == len(so.dt) and sg.pois_n == so.pois_n
    ug = sg.field("u")
    assert np.abs(ug - so.u).max() < 2e-5
    ue = np.zeros_like(ug)
    t = sg.time()
    for i in (1, 2):
        for a in range(ug.shape[0]):
            for b in range(ug.shape[1]):
                ue[a, b, i - 1] = ic(i, oracle.loc(i, (a + 1, b + 1))) * math.exp(-2 * kap**2 * nu * t)
    assert oracle.L2(ug[:, :, 0] - ue[:, :, 0]) < 1e-4 and oracle.L2(ug[:, :, 1] - ue[:, :, 1]) < 1e-4   # test/test_flow.jl:107-108


# test/test_flow.jl:76-84 on the HIP path
def test_impulsive_flow_in_box(w, oracle):
    U = (2 / 3, -1 / 3)
    a = w.Flow((16, 16), U)
    b = w.MultiLevelPoisson(a.p, a.mu0, a.sigma)
    w.mom_step_(a, b)
    u = w.to_host(a.u)
    assert oracle.L2(u[:, :, 0] - np.float32(U[0])) < 2e-5
    assert oracle.L2(u[:, :, 1] - np.float32(U[1])) < 1e-5


def test_sphere_measure_steps_and_force(w, oracle):
    """configs[3] in miniature: sphere R=4 in 32³ (closed-form measure!, BDIM with μ₁, pressure_force)."""
    N, R = 32, 4.0
    c = (N / 2 - 1,) * 3
    nu = 2 * R / 3700
    so = oracle.Simulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=nu, body=("sphere", c, R), T=np.float32)
    sg = w.FusedSimulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=nu, has_body=True)
    sg.measure_sphere_(c, R, 1.0)
    for name, tol in (("mu0", 2e-6), ("mu1", 2e-6), ("V", 0)):
        assert np.abs(sg.field(name) - so.field(name)).max() <= tol, name
    # identical coefficients from here on so that the step comparison is about the step
    sg.set_field("mu0", so.field("mu0")); sg.set_field("mu1", so.field("mu1")); sg.update_()
    for step in range(3):
        so.step(remeasure=False); sg.mom_step_()
        assert sg.pois_n == so.pois_n
        assert np.abs(sg.field("u") - so.u).max() < 5e-5
    fo, fg = so.pressure_force(), sg.pressure_force_sphere(c, R)
    assert np.allclose(fg, fo, rtol=2e-3, atol=2e-3 * np.abs(fo).max())
    vo, vg = so.viscous_force(), sg.viscous_force_sphere(c, R)              # src/Metrics.jl:140-154 (row f2)
    assert np.abs(vo).max() > 0 and np.allclose(vg, vo, rtol=2e-3, atol=2e-3 * np.abs(vo).max())
    assert np.allclose(sg.total_force_sphere(c, R), so.total_force(), rtol=2e-3, atol=2e-3 * np.abs(fo).max())


@pytest.mark.parametrize("N", [(66, 34, 18), (130, 130, 34), (34, 18, 130), (70, 45, 35), (514, 66, 20)])
@pytest.mark.parametrize("omega", [1.0, 0.73])
def test_fused_smoother_is_bit_identical(w, oracle, N, omega):
    """the temporally blocked GaussSeidelRB! (two z-marching kernels, overlapped tiles, several z-chunks) against the
    oracle AND against the one-kernel-per-pass path: ϵ, r, x bit for bit — incl. zero coefficients (iD==0), odd last
    dimension (quirk Q4) and tiles that end exactly on the domain boundary."""
    import ctypes as C
    rng = np.random.default_rng(41)
    D = 3
    L = np.asfortranarray(rng.uniform(0.0, 1.0, size=N + (D,)).astype(np.float32))
    L[L < 0.1] = 0
    L[5:8, 4:7, 3:6] = 0
    oracle.BC(L, (0,) * D)
    x0 = np.asfortranarray(rng.uniform(-1, 1, size=N).astype(np.float32))
    z = F(N)
    r0 = F(N)
    r0[1:-1, 1:-1, 1:-1] = rng.uniform(-1, 1, size=tuple(n - 2 for n in N)).astype(np.float32)
    try:
        po = oracle.MultiLevelPoisson(x0.copy(order="F"), L.copy(order="F"), z.copy(order="F"))
    except AssertionError:
        po = oracle.Poisson(x0.copy(order="F"), L.copy(order="F"), z.copy(order="F"))
    po.field("r", 0)[...] = r0
    po.GaussSeidelRB(0, 4, omega)
    lib = w.lib()
    res = {}
    for fused in (True, False):
        xg, Lg, zg = w.to_device(x0), w.to_device(L), w.to_device(z)
        pg = w.MultiLevelPoisson(xg, Lg, zg, maxlevels=2 if not hasattr(po, "multilevel") or not po.multilevel else 10) if False else None
        try:
            pg = w.MultiLevelPoisson(xg, Lg, zg)
        except AssertionError:
            pytest.skip("shape has fewer than 3 levels")
        pg.set_fused(fused)
        w._lib.check(lib.wl_h2d(lib.wl_mg_level_field(pg._h, 0, b"r"), r0.ctypes.data_as(C.c_void_p), r0.nbytes, w.core.stream()))
        pg.smooth_(0, 4, omega)
        res[fused] = (pg.levels[0].eps, pg.levels[0].r, w.to_host(xg))
    for name, k in (("eps", 0), ("r", 1), ("x", 2)):
        assert np.array_equal(res[True][k], res[False][k]), ("fused vs passes", name)
        assert np.array_equal(res[True][k], po.field(name, 0)), ("fused vs oracle", name)


@pytest.mark.parametrize("dims", [(32, 32, 32), (72, 24, 40), (130, 34, 16)])
@pytest.mark.parametrize("body", [False, True])
def test_zmarching_conv_diff_is_bit_identical(w, oracle, dims, body):
    """predictor and corrector (conv_diff! [+BDIM!]) through the z-marching flux-once kernel vs the gather kernel vs the
    oracle: f and u bit for bit (tiles ending on the boundary, several z-chunks, body and NoBody paths)."""
    rng = np.random.default_rng(43)
    nu = 0.03
    R, c = 4.0, tuple(n / 2 - 1 for n in dims)
    so = oracle.Simulation(dims, (1.0, 0.0, 0.0), dims[0], U=1, nu=nu, body=("sphere", c, R) if body else None, T=np.float32)
    Ng = tuple(n + 2 for n in dims)
    u_init = np.asfortranarray(rng.uniform(-1, 1, size=Ng + (3,)).astype(np.float32))
    oracle.BC(u_init, (1.0, 0.0, 0.0))
    so.field("u")[...] = u_init
    so.field("u0")[...] = u_init
    res = {}
    for convz in (1, 0):
        sg = w.FusedSimulation(dims, (1.0, 0.0, 0.0), dims[0], U=1, nu=nu, has_body=body, u0=u_init)
        if body:
            sg.set_field("mu0", so.field("mu0")); sg.set_field("mu1", so.field("mu1")); sg.update_()
        sg.set_option("convz", convz)
        sg.set_option("store_f", 1)
        out = []
        for ph in (0, 1, 2, 3):
            sg.phase_(ph)
            if ph in (1, 3):
                out.append((sg.field("f"), sg.field("u"), sg.field("sigma")))
        res[convz] = out
    outo = []
    for ph in (0, 1, 2, 3):
        so.phase(ph)
        if ph in (1, 3):
            outo.append((so.field("f").copy(), so.field("u").copy(), so.field("sigma").copy()))
    ghost = np.ones(Ng, dtype=bool)
    ghost[1:-1, 1:-1, 1:-1] = False
    for k in range(2):
        if k == 0:   # predictor: inputs identical bit for bit on all three paths
            assert np.array_equal(res[1][k][0], outo[k][0]) and np.array_equal(res[1][k][1], outo[k][1]), "z-march vs oracle"
            assert np.array_equal(res[0][k][0], outo[k][0]) and np.array_equal(res[0][k][1], outo[k][1]), "gather vs oracle"
            assert np.array_equal(res[1][k][2][ghost], outo[k][2][ghost]), "Q1 ghost fluxes"
        # corrector input went through a pressure solve (reductions): the two HIP paths must still agree exactly
        assert np.array_equal(res[1][k][0], res[0][k][0]) and np.array_equal(res[1][k][1], res[0][k][1]), ("z-march vs gather", k)
        assert np.abs(res[1][k][1] - outo[k][1]).max() < 2e-5


@pytest.mark.parametrize("dims", [(64, 32, 48), (32, 32, 32), (128, 64), (20, 36, 24), (128, 64, 32), (192, 32, 16)])
@pytest.mark.parametrize("fused", [1, 0])
def test_constant_coefficient_levels_are_bit_identical(w, dims, fused):
    """NoBody: L, D, iD evaluated from the cell position (wl::ConstL, verified at update!) vs loaded from memory — whole
    time steps bit for bit, with and without the blocked smoother (semi-coarsened and odd-sized hierarchies included)."""
    out = {}
    for constl in (1, 0):
        U = (1.0,) + (0.0,) * (len(dims) - 1)
        rng = np.random.default_rng(3)
        u = np.asfortranarray(rng.uniform(-1, 1, size=tuple(n + 2 for n in dims) + (len(dims),)).astype(np.float32))
        s = w.FusedSimulation(dims, U, dims[0], U=1, nu=0.01, u0=u)
        s.set_option("constl", constl); s.set_option("fused_smoother", fused)
        if constl:
            assert s.const_levels()[0]
        else:
            assert not any(s.const_levels())
        for _ in range(3):
            s.mom_step_()
        out[constl] = (s.field("u"), s.field("p"), list(s.pois_n), list(s.dt))
    assert out[0][2] == out[1][2] and out[0][3] == out[1][3]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def _const_L(N, c):
    L = np.zeros(N + (3,), dtype=np.float32, order="F")
    for a in range(3):
        L[..., a] = np.float32(c[a])
    return L


@pytest.mark.parametrize("N,c", [((66, 34, 18), (1, 1, 1)), ((130, 130, 34), (1, 1, 1)), ((70, 46, 35), (0.5, 1.0, 0.25)), ((514, 66, 20), (1, 2, 4)),
                                 ((66, 34, 130), (1, 1, 1)), ((194, 66, 12), (2, 1, 1))])
@pytest.mark.parametrize("omega", [1.0, 0.73])
def test_pair_smoother_is_bit_identical(w, oracle, N, c, omega):
    """constant-coefficient level: two-cells-per-thread blocked GaussSeidelRB! (wl_fused2.hip) vs the one-cell blocked kernels
    vs one kernel per pass vs the oracle — ϵ, r, x bit for bit (tiles ending on the boundary, several z-chunks, quirk Q4,
    anisotropic coefficients as on semi-coarsened levels)."""
    import ctypes as C
    rng = np.random.default_rng(47)
    L = _const_L(N, c)
    oracle.BC(L, (0, 0, 0))
    x0 = np.asfortranarray(rng.uniform(-1, 1, size=N).astype(np.float32))
    z, r0 = F(N), F(N)
    r0[1:-1, 1:-1, 1:-1] = rng.uniform(-1, 1, size=tuple(n - 2 for n in N)).astype(np.float32)
    try:
        po = oracle.MultiLevelPoisson(x0.copy(order="F"), L.copy(order="F"), z.copy(order="F"))
    except AssertionError:
        po = oracle.Poisson(x0.copy(order="F"), L.copy(order="F"), z.copy(order="F"))
    po.field("r", 0)[...] = r0
    po.GaussSeidelRB(0, 4, omega)
    lib = w.lib()
    res = {}
    for tag, fused, pair in (("pair", True, True), ("one", True, False), ("passes", False, False)):
        xg, Lg, zg = w.to_device(x0), w.to_device(L), w.to_device(z)
        try:
            pg = w.MultiLevelPoisson(xg, Lg, zg)
        except AssertionError:
            pytest.skip("shape has fewer than 3 levels")
        assert pg.level_is_const(0)
        pg.set_fused(fused, pair)
        w._lib.check(lib.wl_h2d(lib.wl_mg_level_field(pg._h, 0, b"r"), r0.ctypes.data_as(C.c_void_p), r0.nbytes, w.core.stream()))
        pg.smooth_(0, 4, omega)
        res[tag] = (pg.levels[0].eps, pg.levels[0].r, w.to_host(xg))
        pg.set_fused(True, True)
    for name, k in (("eps", 0), ("r", 1), ("x", 2)):
        assert np.array_equal(res["pair"][k], po.field(name, 0)), ("pair vs oracle", name)
        assert np.array_equal(res["pair"][k], res["one"][k]), ("pair vs one-cell blocked", name)
        assert np.array_equal(res["pair"][k], res["passes"][k]), ("pair vs passes", name)


@pytest.mark.parametrize("N", [(130, 66, 34), (66, 66, 66), (98, 50, 26), (258, 34, 18)])
def test_pair_vcycle_with_fused_prolongation_is_bit_identical(w, oracle, N):
    """Vcycle! + the solver's post-smoothing on a NoBody hierarchy: prolongate!+increment! folded into pair kernel A,
    L₁/L∞ from pair kernel B; r and x of every level bit for bit against the oracle and the unfused path."""
    import ctypes as C
    rng = np.random.default_rng(53)
    L = _const_L(N, (1, 1, 1))
    oracle.BC(L, (0, 0, 0))
    x0, z, r0 = F(N), F(N), F(N)
    r0[1:-1, 1:-1, 1:-1] = rng.uniform(-1, 1, size=tuple(n - 2 for n in N)).astype(np.float32)
    po = oracle.MultiLevelPoisson(x0.copy(order="F"), L.copy(order="F"), z.copy(order="F"))
    po.field("r", 0)[...] = r0
    po.Vcycle(0, 0.9); po.GaussSeidelRB(0, 4, 0.9)
    lib = w.lib()
    res = {}
    for tag, fused, pair in (("pair", True, True), ("passes", False, False)):
        xg, Lg, zg = w.to_device(x0), w.to_device(L), w.to_device(z)
        pg = w.MultiLevelPoisson(xg, Lg, zg)
        pg.set_fused(fused, pair)
        w._lib.check(lib.wl_h2d(lib.wl_mg_level_field(pg._h, 0, b"r"), r0.ctypes.data_as(C.c_void_p), r0.nbytes, w.core.stream()))
        pg.Vcycle_(0, 0.9); pg.smooth_(0, 4, 0.9)
        res[tag] = [(pg.levels[l].r, pg.levels[l].x) for l in range(po.nlevels)]
        pg.set_fused(True, True)
    for l in range(po.nlevels):
        for k, name in enumerate(("r", "x")):
            assert np.array_equal(res["pair"][l][k], po.field(name, l)), ("pair vs oracle", l, name)
            assert np.array_equal(res["pair"][l][k], res["passes"][l][k]), ("pair vs passes", l, name)


@pytest.mark.parametrize("N", [(66, 66, 66), (130, 66, 34), (34, 34, 34), (66, 18, 34), (258, 34, 18)])
@pytest.mark.parametrize("const", [True, False])
def test_lds_resident_coarse_tail_is_bit_identical(w, oracle, N, const):
    """Vcycle! whose smallest levels run as ONE launch with r, x, ϵ held in LDS (k_vcycle_tail_lds; coefficients evaluated on
    constant-coefficient hierarchies, loaded otherwise) vs the global-memory tail vs one launch per operation vs the oracle:
    r, x and ϵ of every level bit for bit."""
    import ctypes as C
    rng = np.random.default_rng(71)
    if const:
        L = _const_L(N, (1, 1, 1))
    else:
        L = np.asfortranarray(rng.uniform(0.2, 1, size=N + (3,)).astype(np.float32))
        L[5:9, 4:8, 3:7] = 0
    oracle.BC(L, (0, 0, 0))
    x0, z, r0 = F(N), F(N), F(N)
    r0[1:-1, 1:-1, 1:-1] = rng.uniform(-1, 1, size=tuple(n - 2 for n in N)).astype(np.float32)
    po = oracle.MultiLevelPoisson(x0.copy(order="F"), L.copy(order="F"), z.copy(order="F"))
    po.field("r", 0)[...] = r0
    po.Vcycle(0, 0.9); po.GaussSeidelRB(0, 4, 0.9)
    lib = w.lib()
    res = {}
    for tag, tail, lds in (("lds", True, True), ("global", True, False), ("launches", False, True)):
        xg, Lg, zg = w.to_device(x0), w.to_device(L), w.to_device(z)
        pg = w.MultiLevelPoisson(xg, Lg, zg)
        assert pg.level_is_const(0) == const
        pg.set_fused(True, True, tail=tail, tail_lds=lds)
        w._lib.check(lib.wl_h2d(lib.wl_mg_level_field(pg._h, 0, b"r"), r0.ctypes.data_as(C.c_void_p), r0.nbytes, w.core.stream()))
        pg.Vcycle_(0, 0.9); pg.smooth_(0, 4, 0.9)
        res[tag] = [(pg.levels[l].r, pg.levels[l].x, pg.levels[l].eps) for l in range(po.nlevels)]
        pg.set_fused(True, True)
    for l in range(po.nlevels):
        for k, name in enumerate(("r", "x", "eps")):
            assert np.array_equal(res["lds"][l][k], res["global"][l][k]), ("LDS tail vs global-memory tail", l, name)
            if name != "eps" or l > 0:   # (level 0's ϵ is scratch the pair kernels may skip)
                assert np.array_equal(res["lds"][l][k], res["launches"][l][k]), ("LDS tail vs launches", l, name)
            if name != "eps":
                assert np.array_equal(res["lds"][l][k], po.field(name, l)), ("LDS tail vs oracle", l, name)


# ---------------------------------------------------------------- single-level Poisson (pcg!, solver!)
def poisson_setup_single_gpu(w, oracle, N):
    """Poisson_setup(Poisson,N)   test/test_poisson.jl:1-12 on the HIP path"""
    import ctypes as C
    D = len(N)
    cg = w.to_device(F(N + (D,), 1.0))
    w.BC_(cg, (0,) * D)
    xg, zg = w.jl_zeros(N), w.jl_zeros(N)
    pois = w.Poisson(xg, cg, zg)
    soln = np.asfortranarray(np.broadcast_to((np.arange(N[0], dtype=np.float32) + 1).reshape((N[0],) + (1,) * (D - 1)), N).copy(order="F"))
    I = (1,) * D
    soln -= soln[I]
    w.mult_(pois, w.to_device(soln))          # z = mult!(pois, soln)
    n = w.poisson.solver_(pois)
    x = w.to_host(xg)
    x -= x[I]
    return oracle.L2(x - soln) / oracle.L2(soln), pois, n


# test/test_poisson.jl:14-27 on the HIP path
def test_single_level_poisson_solver(w, oracle):
    err, pois, _ = poisson_setup_single_gpu(w, oracle, (5, 5))
    Dexp = np.array([[0, 0, 0, 0, 0], [0, -2, -3, -2, 0], [0, -3, -4, -3, 0], [0, -2, -3, -2, 0], [0, 0, 0, 0, 0]], dtype=np.float32)
    assert np.array_equal(w.to_host(pois.D), Dexp)
    assert err < 1e-5
    err, pois, n = poisson_setup_single_gpu(w, oracle, (2**6 + 2, 2**6 + 2))
    assert err < 5e-6
    assert n < 340 and pois.n[-1] == n
    assert w.Linf(pois) < 2e-3
    err, pois, n = poisson_setup_single_gpu(w, oracle, (2**4 + 2,) * 3)
    assert err < 1e-6
    assert n < 40


@pytest.mark.parametrize("N", [(34, 18), (18, 18, 18), (20, 11, 14)])
def test_pcg_matches_oracle(w, oracle, N):
    """pcg!(p;it=6) from the same state: x, r, ϵ against the oracle.  The element-wise stages are the reference's statements;
    α, β come from dot products whose summation order differs ⇒ tolerance 2e-5 relative to the field scale; and the whole
    solver! needs the same number of pcg! calls as the oracle."""
    rng = np.random.default_rng(61)
    D = len(N)
    L = np.asfortranarray(rng.uniform(0.3, 1.0, size=N + (D,)).astype(np.float32))
    oracle.BC(L, (0,) * D)
    x0 = np.asfortranarray(rng.uniform(-1, 1, size=N).astype(np.float32))
    z0 = F(N)
    inner = (slice(1, -1),) * D
    z0[inner] = rng.uniform(-1, 1, size=tuple(n - 2 for n in N)).astype(np.float32)
    z0[inner] -= z0[inner].mean(dtype=np.float64).astype(np.float32)
    po = oracle.Poisson(x0.copy(order="F"), L.copy(order="F"), z0.copy(order="F"))
    pg = w.Poisson(w.to_device(x0), w.to_device(L), w.to_device(z0))
    po.residual(); w.residual_(pg)
    po.pcg(0, 6); w.pcg_(pg, 6)
    for name, a, b in (("x", w.to_host(pg.x), po.field("x")), ("r", w.to_host(pg.r), po.field("r")), ("eps", w.to_host(pg.eps), po.field("eps"))):
        scale = max(1.0, float(np.abs(b).max()))
        assert np.abs(a - b).max() <= 2e-5 * scale, (name, np.abs(a - b).max())
    # full solver! from a fresh start
    po = oracle.Poisson(x0.copy(order="F"), L.copy(order="F"), z0.copy(order="F"))
    pg = w.Poisson(w.to_device(x0), w.to_device(L), w.to_device(z0))
    no = po.solve()
    ng = w.poisson.solver_(pg)
    assert abs(ng - no) <= 1 and w.Linf(pg) < 2e-3
    assert np.abs(w.to_host(pg.x) - po.field("x")).max() < 5e-3 * max(1.0, float(np.abs(po.field("x")).max()))


# ---------------------------------------------------------------- time-dependent uniform uBC / body force
# test/test_flow.jl:111-121 on the HIP path (Float32 here; the reference test runs Float64)
def test_increasing_body_force(w, oracle):
    N, jerk = 8, 4
    Us = math.sqrt(N)
    sim = w.FusedSimulation((N, N), (Us, 0.0), N, nu=0.001, dt=0.001, perdir=(1,), g=lambda i, t: t * jerk if i == 1 else 0.0)
    sim.sim_step_(1.0)
    u = sim.field("u")
    uFinal = np.float32(Us + 0.5 * jerk * sim.time() ** 2)
    assert oracle.L2(u[:, :, 0] - uFinal) < 1e-4 and oracle.L2(u[:, :, 1]) < 1e-4


def test_forced_steps_match_oracle(w, oracle):
    """accelerate! + time-dependent boundary velocity: uBC(i,t) = (1+t/2, 0, 0), g(i,t) = (0, 0.3t, -0.1): ten steps against the oracle."""
    dims = (24, 16, 16)
    ufn = lambda i, t: 1.0 + 0.5 * t if i == 1 else 0.0
    dufn = lambda i, t: 0.5 if i == 1 else 0.0
    gfn = lambda i, t: (0.0, 0.3 * t, -0.1)[i - 1]
    so = oracle.Simulation(dims, lambda i, x, t: ufn(i, t), 16, U=1, nu=0.01, T=np.float32, g=lambda i, x, t: gfn(i, t), duBC_dt=lambda i, x, t: dufn(i, t))
    sg = w.FusedSimulation(dims, ufn, 16, U=1, nu=0.01, g=gfn, duBC_dt=dufn)
    for step in range(10):
        so.step(remeasure=False); sg.mom_step_()
        assert sg.pois_n[-2:] == so.pois_n[-2:]
        assert np.abs(sg.field("u") - so.u).max() < 3e-5 * np.abs(so.u).max(), step       # the flow accelerates to |u|≈3
        assert abs(float(sg.dt[-1]) - float(so.dt[-1])) <= 1e-5 * float(so.dt[-1])


# test/test_flow.jl:161-173 on the HIP path: circle in an accelerating flow uBC(i,x,t) = i==1 ? t : 0 — added mass
def test_circle_in_accelerating_flow(w):
    radius, H = 32, 16
    n = radius * 2 * H
    c = (H * radius, H * radius)
    sim = w.FusedSimulation((n, n), lambda i, t: t if i == 1 else 0.0, radius, U=1, has_body=True, duBC_dt=lambda i, t: 1.0 if i == 1 else 0.0)
    sim.measure_sphere_(c, radius, 1.0)
    sim.mom_step_()
    force = sim.pressure_force_sphere(c, radius) / (math.pi * radius**2)
    assert np.allclose(force, [-1, 0], atol=0.04)            # added-mass force of a circle: -π R² dU/dt
    u = sim.field("u")
    assert u.max() / u[1, 1, 0] > 1.91                       # potential flow: maximum speed 2U at the shoulder
    for _ in range(3):
        sim.mom_step_()
    assert all(k <= 2 for k in sim.pois_n)


# test/test_flow.jl:134-140 on the HIP path: a parabolic inflow profile uBC(i,x,t) is maintained by the BCs
def test_boundary_layer_profile(w):
    L = 32
    prof = lambda i, x, t: float(np.float32(4.0 * (((x[1] + 0.5) / (2 * L)) - ((x[1] + 0.5) / (2 * L)) ** 2))) if i == 1 else 0.0
    sim = w.Simulation((L, L), prof, L, nu=0.001, U=1, duBC_dt=lambda i, x, t: 0.0)
    sim.sim_step_(10)
    u = w.to_host(sim.flow.u)
    assert np.allclose(u[0, :, 0], u[-1, :, 0], rtol=float(np.sqrt(np.finfo(np.float32).eps)))


# test/test_flow.jl:142-158 on the HIP path (Float32 here): solid-body rotation seen from the rotating frame — Coriolis and
# centrifugal forces through g(i,x,t), velocity BC through uBC(i,x,t): the pressure stays (nearly) zero
def test_rotating_reference_frame(w, oracle):
    import math
    Lh = 4
    N, om = 2 * Lh, 1.0 / Lh
    x0 = (float(Lh), float(Lh))

    def velocity(i, x, t):
        s, c = math.sin(om * t), math.cos(om * t)
        y = (om * (x[0] - x0[0]), om * (x[1] - x0[1]))
        return s * y[0] + c * y[1] if i == 1 else -c * y[0] + s * y[1]

    def dvel(i, x, t):
        s, c = math.sin(om * t), math.cos(om * t)
        y = (om * (x[0] - x0[0]), om * (x[1] - x0[1]))
        return om * (c * y[0] - s * y[1]) if i == 1 else om * (s * y[0] + c * y[1])

    coriolis = lambda i, x, t: 2 * om * velocity(2, x, t) if i == 1 else -2 * om * velocity(1, x, t)
    centrifugal = lambda i, x, t: om**2 * (x[i - 1] - x0[i - 1])
    g = lambda i, x, t: coriolis(i, x, t) + centrifugal(i, x, t)
    sim = w.Simulation((N, N), velocity, N, U=1, g=g, duBC_dt=dvel)
    sim.sim_step_()
    assert oracle.L2(w.to_host(sim.flow.p)) < 3e-3


# test/test_metrics.jl:67-90 on the HIP path: temporal averages of a steady boundary-layer flow equal the instantaneous fields
def test_meanflow_temporal_averages(w, tmp_path):
    L = 32
    prof = lambda i, x, t: float(np.float32(4.0 * (((x[1] + 0.5) / (2 * L)) - ((x[1] + 0.5) / (2 * L)) ** 2))) if i == 1 else 0.0
    sim = w.Simulation((L, L), prof, L, nu=0.001, U=1, duBC_dt=lambda i, x, t: 0.0)
    mean = w.MeanFlow(sim.flow, uu_stats=True)
    for t in np.arange(0.0, 4.0 + 1e-9, 0.2):
        sim.sim_step_(float(t))
        mean.update_(sim.flow)
    tol = float(np.sqrt(np.finfo(np.float32).eps))
    u, p = w.to_host(sim.flow.u), w.to_host(sim.flow.p)
    U, P, UU = w.to_host(mean.U), w.to_host(mean.P), w.to_host(mean.UU)
    assert np.allclose(u, U, atol=tol) and np.allclose(p, P, atol=tol)
    for i in range(2):
        for j in range(2):
            assert np.allclose(u[:, :, i] * u[:, :, j], UU[:, :, i, j], atol=tol)
    tau = w.to_host(mean.uu())
    for i in range(2):
        for j in range(2):
            assert np.allclose(UU[:, :, i, j] - U[:, :, i] * U[:, :, j], tau[:, :, i, j], atol=tol)
    assert sim.flow.time() == mean.time()
    # one update step against the formula in float32 (bit for bit)
    m2 = w.MeanFlow(sim.flow, t_init=0.0, uu_stats=True)
    m2.P.fill_(0.25); m2.U.fill_(-0.5); m2.UU.fill_(2.0)
    m2.t = [np.float32(0.0), np.float32(1.0)]
    dt = np.float32(sim.flow.time() - np.float32(1.0))
    eps = np.float32(dt / np.float32(dt + np.float32(1.0) + np.finfo(np.float32).eps))
    m2.update_(sim.flow)
    one_m = np.float32(1) - eps
    assert np.array_equal(w.to_host(m2.P), eps * p + one_m * np.float32(0.25))
    assert np.array_equal(w.to_host(m2.U), eps * u + one_m * np.float32(-0.5))
    assert np.array_equal(w.to_host(m2.UU)[:, :, 0, 1], eps * (u[:, :, 0] * u[:, :, 1]) + one_m * np.float32(2.0))
    # checkpoint round trip (the reference's JLD2 extension stores u, p, Δt)
    path = str(tmp_path / "flow.npz")
    w.save_checkpoint(path, sim.flow)
    dt_hist = list(sim.flow.dt)
    sim.flow.u.zero_(); sim.flow.p.zero_(); sim.flow.dt[:] = [np.float32(0.25)]
    w.load_checkpoint(path, sim.flow)
    assert np.array_equal(w.to_host(sim.flow.u), u) and np.array_equal(w.to_host(sim.flow.p), p) and list(sim.flow.dt) == dt_hist
    mean.reset_()
    assert float(w.to_host(mean.U).max()) == 0.0 and mean.t == [np.float32(0.0)]


@pytest.mark.parametrize("perdir", [(1, 2, 3), (1, 3), (2,)])
def test_periodic_3d_steps_match_oracle(w, oracle, perdir):
    """3-D boxes with periodic directions (perBC!, ϕuP, periodic μ₀ — SURVEY §8d's secondary TGV case when all three are):
    the fused projection head/tail run here too; three steps against the oracle."""
    N = 32
    rng = np.random.default_rng(83)
    Ng = (N + 2,) * 3
    u_init = np.asfortranarray(rng.uniform(-0.5, 0.5, size=Ng + (3,)).astype(np.float32))
    U = (0.4, 0.0, 0.0)
    so = oracle.Simulation((N, N, N), U, N, U=1, nu=0.01, perdir=perdir, T=np.float32)
    oracle.BC(u_init, U, False, perdir)
    so.field("u")[...] = u_init
    so.field("u0")[...] = u_init
    sg = w.FusedSimulation((N, N, N), U, N, U=1, nu=0.01, perdir=perdir, u0=u_init)
    for step in range(3):
        so.step(remeasure=False); sg.mom_step_()
        assert sg.pois_n[-2:] == so.pois_n[-2:]
        assert np.abs(sg.field("u") - so.u).max() < 3e-5, step
        assert np.abs(sg.field("p") - so.p).max() < 3e-4, step


@pytest.mark.parametrize("exitBC", [True, False])
def test_body_mask_fast_path_is_bit_identical(w, exitBC):
    """Far from the body BDIM! degenerates to the NoBody form: (i) its u pass skips μ₁, V and the f neighbours in workgroups where
    μ₁ ≡ 0 and V ≡ 0; (ii) without the convective exit conv_diff! applies it directly (k_conv_diff<…,FUSE=2>) and the two-pass kernels
    run on the near workgroups only.  Masks are refreshed by measure!/update!: same bits as the general path, also after the body moved."""
    N, R = 64, 8.0
    res = {}
    for fm in (1, 0):
        sim = w.FusedSimulation((N, N, N), (1, 0, 0), 2 * R, U=1, nu=2 * R / 250, has_body=True, exitBC=exitBC)
        sim.set_option("farmask", fm); sim.set_option("hybrid", fm)
        sim.measure_sphere_((N / 4, N / 2 - 1, N / 2 - 1), R, 1.0)
        for _ in range(2):
            sim.mom_step_()
        sim.measure_sphere_((N / 4 + 3.5, N / 2 + 2, N / 2 - 1), R, 1.0)      # the mask follows the body
        for _ in range(2):
            sim.mom_step_()
        res[fm] = (sim.field("u"), sim.field("p"), sim.pois_n)
    assert res[0][2] == res[1][2]
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("exitBC", [True, False])
def test_zsplit_smoother_on_body_levels_is_bit_identical(w, exitBC):
    """smooth! on a level with a body: the planes at least four away from every cell whose coefficients leave the NoBody pattern
    run the constant-coefficient pair kernels, the planes around the body the general blocked kernels (plane sub-ranges of the same
    launchers).  Same bits as the general kernels over the whole level, also after the body moved along z."""
    n = (128, 64, 96)
    R = 8.0
    res = {}
    for zs in (1, 0):
        sim = w.FusedSimulation(n, (1, 0, 0), 2 * R, U=1, nu=2 * R / 250, has_body=True, exitBC=exitBC)
        sim.set_option("zsplit", 2 if zs else 0)         # 2: also on levels below the size where the split pays (set before measure!)
        sim.measure_sphere_((n[0] / 4, n[1] / 2 - 1, n[2] / 2 - 1), R, 1.0)
        assert sim.smoother_kinds()[0] == (3 if zs else 1)
        for _ in range(2):
            sim.mom_step_()
        sim.measure_sphere_((n[0] / 4 + 1.5, n[1] / 2, n[2] / 2 + 20.5), R, 1.0)
        assert sim.smoother_kinds()[0] == (3 if zs else 1)
        for _ in range(2):
            sim.mom_step_()
        res[zs] = (sim.field("u"), sim.field("p"), sim.pois_n)
    assert res[0][2] == res[1][2]
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("scheme", [0, 1])
def test_tiled_conv_diff_on_the_body_free_planes_is_bit_identical(w, scheme):
    """conv_diff!+BDIM! with a body: the planes on which no workgroup is near the body, keeps f or loads μ₀ are NoBody planes bit for bit and
    run the LDS-tiled kernel (two plane ranges, below and above the body), the planes in between the gather kernel with the body masks.
    Same u, p and pois.n as the gather kernel over the whole domain; the split follows the body when it moves along z."""
    n = (96, 48, 112)
    R = 6.0
    res = {}
    for bt in (1, 0):
        sim = w.FusedSimulation(n, (1, 0, 0), 2 * R, U=1, nu=2 * R / 250, has_body=True, lam=scheme)
        sim.set_option("body_tile", bt)
        sim.set_option("convt_min", 0)          # no size gate: the tiled kernel on this small box, several z-chunks per range
        sim.measure_sphere_((n[0] / 4, n[1] / 2 - 1, 40.0), R, 1.0)
        for _ in range(2):
            sim.mom_step_()
        sim.measure_sphere_((n[0] / 4 + 1.5, n[1] / 2, 70.5), R, 1.0)
        for _ in range(2):
            sim.mom_step_()
        sim.measure_sphere_((n[0] / 4 + 1.5, n[1] / 2, 10.5), R, 1.0)     # close to the lower wall: only the upper range is long enough
        sim.mom_step_()
        res[bt] = (sim.field("u"), sim.field("p"), sim.pois_n)
        sim.set_option("body_tile", 1)
    assert res[0][2] == res[1][2]
    assert np.isfinite(res[1][0]).all()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_ghost_shell_of_p_is_scaled_when_it_is_not_zero(w):
    """mom_project!'s `x .*= dt` scales ALL cells.  The fused head skips its ghost-shell pass while p's ghost cells are +0 (the state the library itself
    maintains) — after ghost values were written from outside (set_field) the pass must run: same u, p as the two-kernel path, which scales every cell."""
    N = 96
    res = {}
    rng = np.random.default_rng(5)
    for tag, rj in (("fused", 1), ("plain", 0)):
        sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
        sim.set_option("resjac", rj); sim.set_option("resjac_min", 0)
        sim.mom_step_()
        p = sim.field("p")
        g = np.random.default_rng(7).uniform(-1, 1, size=p.shape).astype(np.float32)
        p[0, :, :] = g[0, :, :]; p[-1, :, :] = g[-1, :, :]; p[:, 0, :] = g[:, 0, :]; p[:, -1, :] = g[:, -1, :]; p[:, :, 0] = g[:, :, 0]; p[:, :, -1] = g[:, :, -1]
        sim.set_field("p", p)
        for _ in range(2):
            sim.mom_step_()
        res[tag] = (sim.field("u"), sim.field("p"), sim.pois_n)
    assert res["fused"][2] == res["plain"][2]
    assert np.array_equal(res["fused"][0], res["plain"][0]) and np.array_equal(res["fused"][1], res["plain"][1])
    assert np.abs(res["fused"][1][0, 5, 5]) > 0          # the ghost values survived (scaled by dt and back)


BODIES_3D = [
    ("cylinder", (11.0, 13.5, 0.0), 4.0, 2),                                   # along z: the circle of the reference's 2-D cases, extruded
    ("cylinder", (0.0, 14.0, 12.5), 3.5, 0),                                   # along x
    ("plane", (0.0, 5.0, 0.0), (0.0, 1.0, 0.0)),                                # a floor: solid below y = 5
    ("plane", (16.0, 8.0, 0.0), (0.3, 1.0, 0.0)),                               # an inclined wall, normal not unit
    ("sphere", (12.0, 15.0, 15.5), 4.0, (0.25, -0.125, 0.0)),                   # a translating sphere: V ≠ 0
]


@pytest.mark.parametrize("body", BODIES_3D, ids=lambda b: b[0] + str(len(b)))
def test_closed_form_bodies_measure_steps_and_forces(w, oracle, body):
    """row f1: measure! for the closed-form shapes beyond the sphere — cylinder (axis mask), plane, and a translating body whose
    velocity lands in flow.V — then mom_step! with BDIM! on them and the force read-outs, against the oracle."""
    N = 32
    nu = 0.02
    so = oracle.Simulation((N, N, N), (1, 0, 0), 8.0, U=1, nu=nu, body=body, T=np.float32)
    sg = w.FusedSimulation((N, N, N), (1, 0, 0), 8.0, U=1, nu=nu, has_body=True)
    sg.measure_body_(body, 1.0)
    for name, tol in (("mu0", 2e-6), ("mu1", 2e-6), ("V", 0)):
        assert np.abs(sg.field(name) - so.field(name)).max() <= tol, name
    if len(body) > 3 and body[0] == "sphere":
        assert np.abs(so.field("V")).max() == 0.25                             # the band around the body carries the body velocity
    sg.set_field("mu0", so.field("mu0")); sg.set_field("mu1", so.field("mu1")); sg.update_()
    for step in range(3):
        so.step(remeasure=False); sg.mom_step_()
        assert sg.pois_n == so.pois_n
        assert np.abs(sg.field("u") - so.u).max() < 5e-5, step
    fo, fg = so.pressure_force(), sg.pressure_force_body(body)
    assert np.allclose(fg, fo, rtol=2e-3, atol=2e-3 * np.abs(fo).max())
    vo, vg = so.viscous_force(), sg.viscous_force_body(body)
    assert np.allclose(vg, vo, rtol=2e-3, atol=2e-3 * max(np.abs(vo).max(), 1e-6))
    x0 = (14.0, 12.5, 17.0)                                                     # pressure_moment / viscous_moment about a point (src/Metrics.jl:169-188)
    mo, mg = so.pressure_moment(x0), sg.pressure_moment_body(x0, body)
    assert np.abs(mo).max() > 0 and np.allclose(mg, mo, rtol=2e-3, atol=2e-3 * np.abs(mo).max())
    wo, wg = so.viscous_moment(x0), sg.viscous_moment_body(x0, body)
    assert np.allclose(wg, wo, rtol=2e-3, atol=2e-3 * max(np.abs(wo).max(), 1e-6))


def test_moving_cylinder_remeasure_every_step(w, oracle):
    """sim_step!(remeasure=true) with a body that moves: every step the host passes the new centre and the velocity, measure! and
    update!(pois) run on device (src/WaterLily.jl:136-149) — positions, V, the step and pois.n follow the oracle."""
    n = (48, 32)
    R, Ub = 4.0, (0.5, 0.0)
    c0 = np.array([12.0, 15.0])
    so = oracle.Simulation(n, (0, 0), 2 * R, U=1, nu=0.05, body=("sphere", tuple(c0), R, Ub), T=np.float32)
    sg = w.FusedSimulation(n, (0, 0), 2 * R, U=1, nu=0.05, has_body=True)
    for step in range(5):
        t = float(np.sum(so.dt[:-1]))
        c = tuple(c0 + np.array(Ub) * t)
        so.set_body(("sphere", c, R, Ub)); so.step(remeasure=True)
        sg.measure_body_(("sphere", c, R, Ub), 1.0); sg.mom_step_()
        assert np.abs(sg.field("V") - so.field("V")).max() == 0
        assert np.abs(sg.field("mu0") - so.field("mu0")).max() < 2e-6
        assert sg.pois_n[-2:] == so.pois_n[-2:]
        assert np.abs(sg.field("u") - so.u).max() < 1e-4, step
    assert np.abs(so.u[:, :, 0]).max() > 0.3                                    # the fluid was set in motion by the body


@pytest.mark.parametrize("body", [("sphere", (11.0, 15.0, 15.5), 4.0), ("cylinder", (11.0, 13.5, 0.0), 4.0, 2), ("plane", (0.0, 5.0, 0.0), (0.0, 1.0, 0.0))],
                         ids=["sphere", "cylinder", "plane"])
def test_reference_orchestration_with_a_body(w, oracle, body):
    """Simulation(…; body) over the leaf operations — measure!(flow,body) (wl_measure_body), update!(pois), mom_step! with the general
    BDIM!, pressure_force / viscous_force (wl_*_force_body) on the caller's arrays: the calls a `measure!(::Flow{HipArray}, body)` method
    makes.  Against the oracle, and the fused composite gives the same fields."""
    N = 32
    so = oracle.Simulation((N, N, N), (1, 0, 0), 8.0, U=1, nu=0.02, body=body, T=np.float32)
    sl = w.Simulation((N, N, N), (1, 0, 0), 8.0, U=1, nu=0.02, body=body)
    sf = w.FusedSimulation((N, N, N), (1, 0, 0), 8.0, U=1, nu=0.02, has_body=True)
    sf.measure_body_(body, 1.0)
    for name, tol in (("mu0", 2e-6), ("mu1", 2e-6), ("V", 0)):
        a = w.to_host(getattr(sl.flow, name))
        assert np.abs(a - so.field(name)).max() <= tol, name
        assert np.array_equal(a, sf.field(name)), name
    for step in range(3):
        so.step(remeasure=True); sl.sim_step_(remeasure=True)
        assert list(sl.pois.n) == so.pois_n
        assert np.abs(w.to_host(sl.flow.u) - so.u).max() < 5e-5, step
    fo = so.pressure_force()
    assert np.allclose(sl.pressure_force(), fo, rtol=2e-3, atol=2e-3 * np.abs(fo).max())
    vo = so.viscous_force()
    assert np.allclose(sl.viscous_force(), vo, rtol=2e-3, atol=2e-3 * max(np.abs(vo).max(), 1e-6))
    assert np.allclose(sl.total_force(), so.total_force(), rtol=2e-3, atol=2e-3 * np.abs(fo).max())
    x0 = (10.0, 16.0, 13.5)
    mo = so.total_moment(x0)
    assert np.abs(mo).max() > 0 and np.allclose(sl.total_moment(x0), mo, rtol=2e-3, atol=2e-3 * np.abs(mo).max())


@pytest.mark.parametrize("case", ["tgv", "sphere", "tgv_odd"])
def test_residual_shift_and_norms_without_a_pass_over_r(w, case):
    """residual!'s mean shift (src/Poisson.jl:95-97) and solver!'s first L₁/L∞ (src/MultiLevelPoisson.jl:111) without a pass of their own:
    the finest level's z-marching Jacobi! (the V-cycle's first operation) shifts r as it loads it and accumulates the norms.  Same
    iteration counts, Δt and fields as with the plain pass (a level with a body keeps the pass)."""
    N = 64
    res = {}
    for mode in ("plain", "defer"):
        if case == "tgv":
            sim = w.FusedSimulation((N, N, N), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
        elif case == "tgv_odd":
            sim = w.FusedSimulation((N + 6, N - 4, N + 2), (0, 0, 0), N, U=1, nu=N / 1600.0, ic="tgv")
        else:
            sim = w.FusedSimulation((N, N, N), (1, 0, 0), 16.0, U=1, nu=16.0 / 250, has_body=True)
        sim.set_option("defer_shift", int(mode == "defer"))
        if case == "sphere":
            sim.measure_sphere_((N / 4, N / 2 - 1, N / 2 - 1), 8.0, 1.0)
        for _ in range(6):
            sim.mom_step_()
        res[mode] = (sim.field("u"), sim.field("p"), sim.pois_n, sim.dt)
    for mode in ("defer",):
        assert res[mode][2] == res["plain"][2] and res[mode][3] == res["plain"][3], mode
        assert np.array_equal(res[mode][0], res["plain"][0]) and np.array_equal(res[mode][1], res["plain"][1]), mode


def test_moments_2d_hydrostatic(w, oracle):
    """the reference's moment tests on the device (test/test_metrics.jl:58-66): a fluid at rest has no viscous moment, p = y no
    pressure moment about the circle's centre; about a point shifted by a in x the moment is a·F_y (both slots hold the 2-D scalar)."""
    N = 34                                                                      # array extent (32 interior cells: a·2ⁿ for the multigrid)
    R = 8
    body = ("sphere", (N / 2, N / 2), R)
    sim = w.Simulation((N - 2, N - 2), (0.0, 0.0), N, U=1, nu=1.0, body=body)
    p = np.zeros((N, N), dtype=np.float32, order="F")
    for b in range(1, N - 1):
        p[1:-1, b] = b - 0.5
    sim.flow.p.copy_(w.to_device(p))
    assert np.all(sim.viscous_moment((N / 2, N / 2)) == 0)
    Fy = sim.pressure_force()[1]
    assert abs(Fy / (np.pi * R**2) - 1) < 2e-3                                  # test_metrics.jl:40
    m0 = sim.pressure_moment((N / 2, N / 2))
    assert abs(m0[0]) < 1e-4 * abs(Fy)
    ms = sim.pressure_moment((N / 2 - 3.0, N / 2))
    assert np.isclose(ms[0], 3.0 * Fy, rtol=1e-4) and ms[0] == ms[1]
    df = np.zeros((N, N, 2), dtype=np.float32, order="F")
    assert np.allclose(ms, oracle.pressure_moment_body((N / 2 - 3.0, N / 2), p, df, body), rtol=1e-5)


@pytest.mark.parametrize("dims", [(64, 32, 24), (128, 48, 11), (72, 24, 40), (52, 36, 20), (130, 34, 16), (36, 20, 12)])
@pytest.mark.parametrize("lam", [0, 1, 2])
def test_tiled_conv_diff_is_bit_identical(w, oracle, dims, lam):
    """predictor and corrector (conv_diff!+BDIM!, NoBody) through the LDS-tiled z-marching kernels — wl_convf.hip (default: every face
    flux evaluated once; upper faces come from the next lane, from the row above through LDS, and from one generic flux per thread
    on the tile's upper edges) and wl_convt.hip (convf=0: two cells per thread, upper faces re-evaluated) — vs the oracle and vs the
    plane kernel: u bit for bit.  Shapes: whole tiles (64·a × 16·b) and ragged ones (tiles cut by the boundary in x and y, pairs
    straddling the last column), several z-chunks (the test threshold makes chunks of 5 planes), QUICK / vanLeer / CDS."""
    if lam == 1 and dims not in ((64, 32, 24), (52, 36, 20)):
        pytest.skip("vanLeer: two shapes are enough")
    rng = np.random.default_rng(47)
    nu = 0.03
    so = oracle.Simulation(dims, (1.0, 0.0, 0.0), dims[0], U=1, nu=nu, T=np.float32, scheme=lam)
    Ng = tuple(n + 2 for n in dims)
    u_init = np.asfortranarray(rng.uniform(-1, 1, size=Ng + (3,)).astype(np.float32))
    oracle.BC(u_init, (1.0, 0.0, 0.0))
    so.field("u")[...] = u_init
    so.field("u0")[...] = u_init
    res = {}
    for mode, (convt, convf) in {"flux": (1, 1), "tile": (1, 0), "plane": (0, 1)}.items():
        sg = w.FusedSimulation(dims, (1.0, 0.0, 0.0), dims[0], U=1, nu=nu, u0=u_init, lam=lam)
        sg.set_option("convt", convt)
        sg.set_option("convf", convf)
        sg.set_option("convt_min", 0)
        out = []
        for ph in (0, 1, 2, 3):
            sg.phase_(ph)
            if ph in (1, 3):
                out.append(sg.field("u"))
        res[mode] = out
        sg.set_option("convt_min", 8192)
        sg.set_option("convt", 1)
        sg.set_option("convf", 1)
    outo = []
    for ph in (0, 1, 2, 3):
        so.phase(ph)
        if ph in (1, 3):
            outo.append(so.field("u").copy())
    assert np.array_equal(res["plane"][0], outo[0]), "plane kernel vs oracle (predictor)"
    assert np.array_equal(res["flux"][0], outo[0]), "flux-once kernel vs oracle (predictor)"
    assert np.array_equal(res["tile"][0], outo[0]), "tiled kernel vs oracle (predictor)"
    # the corrector's input went through a pressure solve (reductions): the HIP paths must still agree exactly
    assert np.array_equal(res["flux"][1], res["plane"][1]), "flux-once vs plane kernel (corrector)"
    assert np.array_equal(res["tile"][1], res["plane"][1]), "tiled vs plane kernel (corrector)"
    assert np.abs(res["flux"][1] - outo[1]).max() < 2e-5


@pytest.mark.parametrize("dims", [(64, 32, 24), (130, 34, 16), (128, 48, 12)])
def test_mom_steps_equals_repeated_mom_step(w, oracle, dims):
    """wl_sim_mom_steps(n) — between its steps Δt stays on the device until the next predictor has been queued (option lazydt: the predictor reads it through a
    pointer, the host copies the CFL maximum while it runs) — against n calls of wl_sim_mom_step, with the option on and off: u, u⁰, p on every cell, pois.n and the
    whole Δt history bit for bit, a second batch of steps included (the history must be complete whenever a call returns)."""
    rng = np.random.default_rng(73)
    Ng = tuple(n + 2 for n in dims)
    uBC = (0.3, -0.2, 0.1)
    u_init = np.asfortranarray(rng.uniform(-0.4, 0.4, size=Ng + (3,)).astype(np.float32))
    res = {}
    for mode in ("single", "batch", "batch_nolazy"):
        sg = w.FusedSimulation(dims, uBC, dims[0], U=1, nu=0.02, u0=u_init)
        sg.set_option("convt_min", 0)
        sg.set_option("resjac_min", 0)
        sg.set_option("lazydt", 0 if mode == "batch_nolazy" else 1)
        if mode == "single":
            for _ in range(5):
                sg.mom_step_()
            mid = sg.dt
            for _ in range(2):
                sg.mom_step_()
        else:
            sg.mom_steps_(5)
            mid = sg.dt
            sg.mom_steps_(2)
        res[mode] = (sg.field("u"), sg.field("u0"), sg.field("p"), sg.pois_n, sg.dt, mid)
        sg.set_option("convt_min", 8192)
        sg.set_option("resjac_min", 8 << 20)
    assert len(res["single"][4]) == 8 and len(res["single"][5]) == 6
    for mode in ("batch", "batch_nolazy"):
        assert res[mode][3] == res["single"][3] and res[mode][4] == res["single"][4] and res[mode][5] == res["single"][5], mode
        for q in range(3):
            assert np.array_equal(res[mode][q], res["single"][q]), (mode, ("u", "u0", "p")[q])


@pytest.mark.parametrize("dims", [(64, 32, 24), (72, 40, 16), (128, 48, 12)])
def test_tail_queued_ahead_of_the_convergence_read_is_bit_identical(w, oracle, dims):
    """The projection tail queued behind the smoother BEFORE the host has read that iteration's norms, gated on the device by solver!'s break test (and, on the
    first iteration, the fused head's mean-shift test) — option tailspec — against the tail launched after the read: u, u⁰, p on every cell, pois.n, Δt after four
    steps from a random field (the first solves need several V-cycles: the gated tail must have done nothing on all but the last iteration), and the counter."""
    rng = np.random.default_rng(71)
    Ng = tuple(n + 2 for n in dims)
    uBC = (0.3, -0.2, 0.1)
    u_init = np.asfortranarray(rng.uniform(-0.4, 0.4, size=Ng + (3,)).astype(np.float32))
    so = oracle.Simulation(dims, uBC, dims[0], U=1, nu=0.02, T=np.float32)
    oracle.BC(u_init, uBC)
    so.field("u")[...] = u_init
    so.field("u0")[...] = u_init
    res = {}
    for spec in (1, 0):
        sg = w.FusedSimulation(dims, uBC, dims[0], U=1, nu=0.02, u0=u_init)
        sg.set_option("tailspec", spec)
        sg.set_option("convt_min", 0)
        sg.set_option("resjac_min", 0)
        for _ in range(4):
            sg.mom_step_()
        res[spec] = (sg.field("u"), sg.field("u0"), sg.field("p"), sg.pois_n, sg.dt)
        assert (sg.counter("tailspec") >= 4) if spec else (sg.counter("tailspec") == 0)     # (a solve whose head had to be redone, or that hit the iteration cap, launches its tail after the read)
        sg.set_option("convt_min", 8192)
        sg.set_option("resjac_min", 8 << 20)
    for _ in range(4):
        so.step(remeasure=False)
    assert max(res[1][3]) > 1, "the case must contain solves that iterate"
    assert res[1][3] == res[0][3] and res[1][4] == res[0][4]
    for q in range(3):
        assert np.array_equal(res[1][q], res[0][q]), ("u", "u0", "p")[q]
    assert res[1][3] == so.pois_n
    assert np.abs(res[1][0] - so.u).max() < 5e-5


@pytest.mark.parametrize("dims", [(64, 32, 24), (72, 40, 16), (130, 34, 16), (128, 48, 12)])
@pytest.mark.parametrize("uBC", [(1.0, 0.0, 0.0), (0.3, -0.2, 0.1)])
def test_deferred_bc_is_bit_identical(w, oracle, dims, uBC):
    """mom_step! with BC!(u,U) after the fused conv_diff!+BDIM! left to the projection (option bcdefer: the fused head and the pair tail read U on the
    wall-normal boundary faces, the tails' folded stores rewrite every boundary location; two k_bc_vec launches fewer per step) against the step that
    applies BC! where the reference does — u, u⁰, p on EVERY cell (ghosts, edges, corners), pois.n, Δt after three steps; with the fused head standing
    (deferral live: 2 per step) and with the head forced onto its redo path (the deferred BC! is applied before the two-kernel head reads u)."""
    rng = np.random.default_rng(67)
    Ng = tuple(n + 2 for n in dims)
    u_init = np.asfortranarray(rng.uniform(-0.4, 0.4, size=Ng + (3,)).astype(np.float32))
    so = oracle.Simulation(dims, uBC, dims[0], U=1, nu=0.02, T=np.float32)
    oracle.BC(u_init, uBC)
    so.field("u")[...] = u_init
    so.field("u0")[...] = u_init
    res = {}
    for mode, (defer, rj) in {"defer": (1, 1), "plain": (0, 1), "defer_redo": (1, 3)}.items():
        sg = w.FusedSimulation(dims, uBC, dims[0], U=1, nu=0.02, u0=u_init)
        sg.set_option("bcdefer", defer)
        sg.set_option("convt_min", 0)
        sg.set_option("resjac_min", 0)
        sg.set_option("resjac", rj)
        for _ in range(3):
            sg.mom_step_()
        res[mode] = (sg.field("u"), sg.field("u0"), sg.field("p"), sg.pois_n, sg.dt)
        assert sg.counter("bcdefer") == {"defer": 6, "plain": 0, "defer_redo": 6}[mode], mode
        sg.set_option("convt_min", 8192)
        sg.set_option("resjac_min", 8 << 20)
    for _ in range(3):
        so.step(remeasure=False)
    for mode in ("defer", "defer_redo"):
        assert res[mode][3] == res["plain"][3] and res[mode][4] == res["plain"][4], mode
        for q in range(3):
            assert np.array_equal(res[mode][q], res["plain"][q]), (mode, ("u", "u0", "p")[q])
    assert res["defer"][3] == so.pois_n
    assert np.abs(res["defer"][0] - so.u).max() < 5e-5 and np.abs(res["defer"][2] - so.p).max() < 5e-4


@pytest.mark.parametrize("dims", [(64, 32, 24), (128, 48, 16), (64, 16, 12), (192, 32, 9)])
@pytest.mark.parametrize("uBC,lam", [((1.0, 0.0, 0.0), 0), ((0.3, -0.2, 0.1), 0), ((0.3, -0.2, 0.1), 2), ((1.0, 0.0, 0.0), 1)])
def test_deferred_projection_is_bit_identical(w, oracle, dims, uBC, lam):
    """mom_step! with the first projection's tail (u −= L∇x, then BC!) evaluated by the corrector's conv_diff! loader (wl_convf.hip, PROJ: the
    projected predictor velocity is never written) against the step with the separate tail launch — u, u⁰, p, pois.n, Δt bit for bit on every
    cell after three steps, wall tiles and interior tiles, several z-chunks, tuple U with all components nonzero — and against the oracle."""
    if lam == 1 and dims != (64, 32, 24):
        pytest.skip("vanLeer: one shape is enough")
    rng = np.random.default_rng(61)
    Ng = tuple(n + 2 for n in dims)
    u_init = np.asfortranarray(rng.uniform(-0.4, 0.4, size=Ng + (3,)).astype(np.float32))
    so = oracle.Simulation(dims, uBC, dims[0], U=1, nu=0.02, T=np.float32, scheme=lam)
    oracle.BC(u_init, uBC)
    so.field("u")[...] = u_init
    so.field("u0")[...] = u_init
    res = {}
    for fuse in (1, 0):
        sg = w.FusedSimulation(dims, uBC, dims[0], U=1, nu=0.02, u0=u_init, lam=lam)
        sg.set_option("tailfuse", fuse)
        sg.set_option("convt_min", 0)
        sg.set_option("resjac_min", 0)
        for _ in range(3):
            sg.mom_step_()
        res[fuse] = (sg.field("u"), sg.field("u0"), sg.field("p"), sg.pois_n, sg.dt)
        assert sg.counter("tailfuse") == (3 if fuse else 0)
        sg.set_option("convt_min", 8192)
        sg.set_option("resjac_min", 8 << 20)
    for _ in range(3):
        so.step(remeasure=False)
    assert res[1][3] == res[0][3] and res[1][4] == res[0][4]
    for q in range(3):
        assert np.array_equal(res[1][q], res[0][q]), ("u", "u0", "p")[q]
        assert np.array_equal(np.signbit(res[1][q]), np.signbit(res[0][q])), ("sign of zero", ("u", "u0", "p")[q])
    assert res[1][3] == so.pois_n
    assert np.abs(res[1][0] - so.u).max() < 5e-5 and np.abs(res[1][2] - so.p).max() < 5e-4


@pytest.mark.parametrize("dims", [(64, 32, 24), (96, 64, 40), (70, 44, 18), (128, 36, 12)])
def test_fused_projection_head_is_bit_identical(w, oracle, dims):
    """mom_project!'s head (z=∇·u, x·=dt, residual!) and the V-cycle's first Jacobi! in ONE z-marching kernel (wl_resjac.hip), assuming
    the mean shift of residual! is not due — against the two-kernel path bit for bit and against the oracle (u, p, pois.n, Δt), on whole
    and ragged tiles and several z-chunks; and the redo path (shift declared due: results discarded, two-kernel path taken)."""
    rng = np.random.default_rng(53)
    Ng = tuple(n + 2 for n in dims)
    u_init = np.asfortranarray(rng.uniform(-0.4, 0.4, size=Ng + (3,)).astype(np.float32))
    u_init[..., 0] += 1.0
    so = oracle.Simulation(dims, (1.0, 0.0, 0.0), dims[0], U=1, nu=0.02, T=np.float32)
    oracle.BC(u_init, (1.0, 0.0, 0.0))
    so.field("u")[...] = u_init
    so.field("u0")[...] = u_init
    res = {}
    for mode in (1, 0, 2, 11, 12):      # 1: fused head (first V-cycle queued before Σr is read: headspec), 0: two kernels, 2: forced redo; 11 / 12: as 1 / 2 with headspec off
        sg = w.FusedSimulation(dims, (1.0, 0.0, 0.0), dims[0], U=1, nu=0.02, u0=u_init)
        sg.set_option("resjac_min", 0)
        sg.set_option("resjac", mode % 10)
        sg.set_option("headspec", 0 if mode > 10 else 1)
        os.environ.pop("WL_RJ_CHUNK", None)
        for _ in range(3):
            sg.mom_step_()
        res[mode] = (sg.field("u"), sg.field("p"), sg.pois_n, sg.dt)
        sg.set_option("resjac_min", 8 << 20)
    for _ in range(3):
        so.step(remeasure=False)
    for mode in (1, 2, 11, 12):
        assert res[mode][2] == res[0][2] and res[mode][3] == res[0][3], mode
        assert np.array_equal(res[mode][0], res[0][0]) and np.array_equal(res[mode][1], res[0][1]), mode
    assert res[1][2] == so.pois_n
    assert np.abs(res[1][0] - so.u).max() < 5e-5 and np.abs(res[1][1] - so.p).max() < 5e-4


@pytest.mark.parametrize("dims", [(64, 32, 24), (72, 40, 16), (130, 34, 16), (16, 16, 16)])
@pytest.mark.parametrize("uBC", [(1.0, 0.0, 0.0), (0.3, -0.2, 0.1)])
def test_bc_folded_into_producers_is_bit_identical(w, oracle, dims, uBC):
    """BC!(u,U) for a tuple U folded into the stores of the kernels that produce u (tiled conv_diff!+BDIM!: x/y in the wall tiles + one
    z-plane launch; projection tails: every boundary-adjacent cell also writes the ghost locations it owns) — against the separate
    k_bc_vec launches bit for bit on EVERY cell (ghosts, edges, corners), and against the oracle."""
    rng = np.random.default_rng(59)
    Ng = tuple(n + 2 for n in dims)
    u_init = np.asfortranarray(rng.uniform(-0.4, 0.4, size=Ng + (3,)).astype(np.float32))
    so = oracle.Simulation(dims, uBC, dims[0], U=1, nu=0.02, T=np.float32)
    oracle.BC(u_init, uBC)
    so.field("u")[...] = u_init
    so.field("u0")[...] = u_init
    res = {}
    for fold in (3, 1, 0):
        sg = w.FusedSimulation(dims, uBC, dims[0], U=1, nu=0.02, u0=u_init)
        sg.set_option("bcfold", fold)      # bit 0: projection tails, bit 1: tiled conv_diff!+BDIM!
        sg.set_option("convt_min", 0)
        sg.set_option("resjac_min", 0)
        for _ in range(3):
            sg.mom_step_()
        res[fold] = (sg.field("u"), sg.field("u0"), sg.field("p"), sg.pois_n, sg.dt)
        sg.set_option("convt_min", 8192)
        sg.set_option("resjac_min", 8 << 20)
    for _ in range(3):
        so.step(remeasure=False)
    for fold in (3, 1):
        assert res[fold][3] == res[0][3] and res[fold][4] == res[0][4]
        for q in range(3):
            assert np.array_equal(res[fold][q], res[0][q]), (fold, ("u", "u0", "p")[q])
    assert res[3][3] == so.pois_n
    assert np.abs(res[3][0] - so.u).max() < 5e-5
