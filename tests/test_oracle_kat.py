"""Pin the oracle (CPU restatement) with the reference's OWN known-answer tests.

The reference (pure Julia) cannot run in this pipeline, so each test below re-states one of its
analytical tests against oracle/ — file:line of the original is given per test.  These run on CPU.
"""
import math

import numpy as np
import pytest


def F(shape, T=np.float32, fill=0.0):
    return np.full(shape, fill, dtype=T, order="F")


# ------------------------------------------------------------------ test/test_core.jl:2-10
def test_loc(oracle):
    assert np.allclose(oracle.loc(3, (3, 4, 5)), np.array([3, 4, 4.5]) - 1.5)
    rng = np.random.default_rng(0)
    I = tuple(int(v) for v in rng.integers(2, 11, 3))
    assert np.allclose(oracle.loc(0, I), np.array(I) - 1.5)


# ------------------------------------------------------------------ test/test_core.jl:19-56
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_BC_tuple_saveexit_periodic(oracle, T):
    rng = np.random.default_rng(1)
    Ng, D, U = (6, 6), 2, (1.0, 0.5)
    u = np.asfortranarray(rng.random(Ng + (D,)).astype(T))
    s = np.asfortranarray(rng.random(Ng).astype(T))
    oracle.BC(u, U)
    assert np.all(u[0, :, 0] == U[0]) and np.all(u[1, :, 0] == U[0]) and np.all(u[-1, :, 0] == U[0])
    assert np.all(u[2:-1, 0, 0] == u[2:-1, 1, 0]) and np.all(u[2:-1, -1, 0] == u[2:-1, -2, 0])
    assert np.all(u[:, 0, 1] == U[1]) and np.all(u[:, 1, 1] == U[1]) and np.all(u[:, -1, 1] == U[1])
    assert np.all(u[0, 2:-1, 1] == u[1, 2:-1, 1]) and np.all(u[-1, 2:-1, 1] == u[-2, 2:-1, 1])

    u[-1, :, 0] = 3
    oracle.BC(u, U, True)  # save exit values
    assert np.all(u[-1, :, 0] == 3)

    oracle.exitBC(u, u, 0.0)  # conservative exit check
    assert np.all(u[-1, 1:-1, 0] == U[0])

    # BC with a Function
    Ubc = lambda i, x, t: 1.0 if i == 1 else 0.5
    v = np.asfortranarray(rng.random(Ng + (D,)).astype(T))
    oracle.BC(v, Ubc, False)
    oracle.BC(u, U, False)
    assert np.all(v[0, :, 0] == u[0, :, 0]) and np.all(v[1, :, 0] == u[1, :, 0]) and np.all(v[-1, :, 0] == u[-1, :, 0])
    assert np.all(v[:, 0, 1] == u[:, 0, 1]) and np.all(v[:, 1, 1] == u[:, 1, 1]) and np.all(v[:, -1, 1] == u[:, -1, 1])
    v[-1, :, 0] = 3
    oracle.BC(v, Ubc, True)
    assert np.all(v[-1, :, 0] == 3)

    oracle.BC(u, U, True, (2,))  # periodic in y and save exit values
    assert np.all(u[:, 0:2, 0] == u[:, -2:, 0])
    oracle.perBC(s, (1, 2))
    assert np.all(s[0, 1:-1] == s[-2, 1:-1]) and np.all(s[1:-1, 0] == s[1:-1, -2])

    u = np.asfortranarray(rng.random(Ng + (D,)).astype(T))
    oracle.BC(u, U, True, (1,))  # saveexit has no effect here as x-periodic
    assert np.all(u[0:2, :, 0] == u[-2:, :, 0]) and np.all(u[0:2, :, 1] == u[-2:, :, 1])
    assert np.all(u[:, 0, 1] == U[1]) and np.all(u[:, 1, 1] == U[1]) and np.all(u[:, -1, 1] == U[1])


# ------------------------------------------------------------------ test/test_core.jl:57-70
def test_BC_function_nonuniform(oracle):
    Ng, D = (6, 6), 2
    v = F(Ng + (D,), np.float64)
    oracle.BC(v, lambda i, x, t: x[1] if i == 1 else x[0])
    assert np.allclose(v[0, 1:-1, 0], v[-1, 1:-1, 0])
    assert np.allclose(v[1:-1, 0, 1], v[1:-1, -1, 1])
    Ng, D = (8, 8, 8), 3
    u = F(Ng + (D,), np.float64)
    Ubc2 = lambda i, x, t: math.cos(2 * math.pi * x[0] / 8) if i == 1 else (math.sin(2 * math.pi * x[1] / 8) if i == 2 else math.tan(math.pi * x[2] / 16))
    oracle.BC(u, Ubc2)
    pi = math.pi
    assert np.allclose(u[0, :, :, 0], math.cos(-pi / 4)) and np.allclose(u[1, :, :, 0], 1.0) and np.allclose(u[-1, :, :, 0], math.cos(6 * pi / 4), atol=1e-6)
    assert np.allclose(u[:, 0, :, 1], math.sin(-pi / 4)) and np.allclose(u[:, 1, :, 1], 0.0, atol=1e-7) and np.allclose(u[:, -1, :, 1], math.sin(6 * pi / 4))
    assert np.allclose(u[:, :, 0, 2], math.tan(-pi / 16)) and np.allclose(u[:, :, 1, 2], 0.0, atol=1e-7) and np.all(u[:, :, -1, 2] - math.tan(6 * pi / 16) < 1e-6)


# ------------------------------------------------------------------ test/test_flow.jl:2-41
def test_limiters_and_boundary_fluxes(oracle):
    o = oracle
    assert o.vanLeer(1, 0, 1) == 0 and o.vanLeer(1, 2, 1) == 2
    assert o.vanLeer(1, 2, 3) == 2.5 and o.vanLeer(3, 2, 1) == 1.5
    assert o.cds(1, 0, 1) == 0.5 and o.cds(1, 2, -1) == 0.5
    f = [0.0, 0.5, 2.0]
    assert o.flux1d("ϕuL", f, 2, 1.0) == o.flux1d("ϕ", f, 2, 0.0)
    assert o.flux1d("ϕuL", f, 2, -1.0) == -o.quick(2.0, 0.5, 0.0)
    assert o.flux1d("ϕuR", f, 3, 1.0) == o.quick(0.0, 0.5, 2.0)
    assert o.flux1d("ϕuR", f, 3, -1.0) == -o.flux1d("ϕ", f, 3, 0.0)
    assert o.flux1d("ϕu", f, 3, 1.0) == o.flux1d("ϕuP", f, 3, 1.0, Ip=1)
    assert o.flux1d("ϕu", f, 2, -1.0) == o.flux1d("ϕuP", f, 2, -1.0, Ip=0 + 1)  # Ip unused for u<0
    f = [1.0, 1.25, 1.5, 1.75, 2.0]
    assert o.flux1d("ϕuP", f, 3, 1.0, Ip=1) == o.quick(f[0], f[1], f[2])
    assert o.flux1d("ϕuP", f, 3, 1.0, Ip=len(f) - 2) == o.quick(f[len(f) - 3], f[1], f[2])


# ------------------------------------------------------------------ test/test_flow.jl:47-51 (L₂(p)==187) + src/Poisson.jl:188
def test_L2_is_sum_of_squares(oracle):
    p = F((4, 5), np.float64)
    for i in range(4):
        for j in range(5):
            x = oracle.loc(0, (i + 1, j + 1))
            p[i, j] = x[0] + x[1] + 3
    assert oracle.L2(p) == 187


# ------------------------------------------------------------------ test/test_bodies.jl:2-5
def test_kernel_moments(oracle):
    assert oracle.mu0(3.0, 6) == oracle.mu0(0.5, 1)
    assert oracle.mu0(0.0, 1) == 0.5
    assert oracle.mu0(np.finfo(np.float64).eps - 1, 1) == 0
    assert oracle.mu1(0.0, 2) == 2 * (1 / 4 - 1 / math.pi**2)


# ------------------------------------------------------------------ test/test_poisson.jl:1-12  Poisson_setup
def poisson_setup(oracle, N, multilevel, T=np.float32):
    D = len(N)
    c = F(N + (D,), T, 1.0)
    oracle.BC(c, (0,) * D)
    x = F(N, T)
    z = F(N, T)
    pois = oracle.Poisson(x, c, z, multilevel=multilevel)
    soln = np.asfortranarray(np.broadcast_to((np.arange(N[0], dtype=T) + 1).reshape((N[0],) + (1,) * (D - 1)), N).copy(order="F"))
    I = (1,) * D  # first(inside(x)), 0-based
    soln -= soln[I]
    pois.mult(soln)  # z = A·soln
    pois.solve()
    x -= x[I]
    err = oracle.L2(x - soln) / oracle.L2(soln)
    return err, pois


# ------------------------------------------------------------------ test/test_poisson.jl:16-27
def test_single_level_poisson(oracle):
    err, pois = poisson_setup(oracle, (5, 5), False)
    Dexp = np.array([[0, 0, 0, 0, 0], [0, -2, -3, -2, 0], [0, -3, -4, -3, 0], [0, -2, -3, -2, 0], [0, 0, 0, 0, 0]], dtype=np.float32)
    assert np.array_equal(pois.field("D"), Dexp)
    iDexp = np.array([[0, 0, 0, 0, 0], [0, -1 / 2, -1 / 3, -1 / 2, 0], [0, -1 / 3, -1 / 4, -1 / 3, 0], [0, -1 / 2, -1 / 3, -1 / 2, 0], [0, 0, 0, 0, 0]], dtype=np.float32)
    assert np.allclose(pois.field("iD"), iDexp)
    assert err < 1e-5
    err, pois = poisson_setup(oracle, (2**6 + 2, 2**6 + 2), False)
    assert err < 5e-6
    assert pois.n[-1] < 340
    assert pois.Linf() < 2e-3
    err, pois = poisson_setup(oracle, (2**4 + 2,) * 3, False)
    assert err < 1e-6
    assert pois.n[-1] < 40


# ------------------------------------------------------------------ test/test_poisson.jl:39-52
def test_multigrid_index_maps(oracle):
    I = (4, 3, 2)
    full = (True, True, True)
    assert all(oracle.down(J, full) == I for J in oracle.up(I, full))
    with pytest.raises(AssertionError, match="MultiLevelPoisson requires size=a2ⁿ, where n>2"):
        poisson_setup(oracle, (15 + 2, 3**4 + 2), True)
    assert oracle.coarsen_mask((18, 18, 6)) == (True, True, True)
    assert oracle.coarsen_mask((18, 18, 4)) == (True, True, False)
    assert oracle.coarsen_mask((18, 17, 6)) == (True, False, True)
    c, I = (True, True, False), (4, 3, 5)
    assert all(oracle.down(J, c) == I for J in oracle.up(I, c))
    assert all(J[2] == I[2] for J in oracle.up(I, c))
    assert len(oracle.up(I, full)) == 8 and oracle.up(I, full)[0] == (6, 4, 8) and oracle.up(I, full)[-1] == (7, 5, 9)


# ------------------------------------------------------------------ test/test_poisson.jl:54-60
def test_multilevel_coarse_diagonal_and_update(oracle):
    err, pois = poisson_setup(oracle, (10, 10), True)
    D3 = np.array([[0, 0, 0, 0], [0, -2, -2, 0], [0, -2, -2, 0], [0, 0, 0, 0]], dtype=np.float32)
    assert np.array_equal(pois.field("D", 2), D3)
    assert err < 1e-5
    pois.field("L", 0)[4:6, :, 0] = 0
    pois.update()
    assert np.array_equal(pois.field("D", 2), D3 / 2)


# ------------------------------------------------------------------ test/test_poisson.jl:62-70
def test_multigrid_convergence(oracle):
    err, pois = poisson_setup(oracle, (2**6 + 2, 2**6 + 2), True)
    assert err < 1e-6
    assert pois.n[-1] <= 4
    assert pois.Linf() < 2e-3
    err, pois = poisson_setup(oracle, (2**4 + 2,) * 3, True)
    assert err < 1e-6
    assert pois.n[-1] <= 3


# ------------------------------------------------------------------ test/test_poisson.jl:72-82 semi-coarsening
def test_semicoarsening_channel_and_duct(oracle):
    H = 2**4
    R = H // 4
    sim = oracle.Simulation((8 * H, H), (1, 0), R, nu=R / 100, body=("sphere", (4 * H, H // 2), R), T=np.float32)
    for _ in range(4):
        sim.step(remeasure=False)
    assert all(n <= 10 for n in sim.pois_n) and len(sim.pois_n) == 8
    assert sim.level_dims(sim.nlevels - 1)[1] == 4  # y stopped coarsening before x => semi-coarsened tail
    H = 2**3
    R = H // 4
    sim = oracle.Simulation((8 * H, H, H), (1, 0, 0), R, nu=R / 100, body=("sphere", (4 * H, H // 2, H // 2), R), T=np.float32)
    for _ in range(4):
        sim.step(remeasure=False)
    assert all(n <= 12 for n in sim.pois_n)
    assert np.all(np.isfinite(sim.u))


# ------------------------------------------------------------------ test/test_flow.jl:76-84 impulsive box
def test_impulsive_flow_in_box(oracle):
    U = (2 / 3, -1 / 3)
    sim = oracle.Simulation((16, 16), U, 16, T=np.float32)
    sim.phase(0); sim.phase(1); sim.phase(2); sim.phase(3); sim.phase(4); sim.phase(5)   # == mom_step!
    u = sim.u
    assert oracle.L2(u[:, :, 0] - np.float32(U[0])) < 2e-5
    assert oracle.L2(u[:, :, 1] - np.float32(U[1])) < 1e-5


# ------------------------------------------------------------------ test/test_flow.jl:87-98 scheme selection
def test_convection_scheme_selection(oracle):
    nonuniform = lambda i, x: math.sin(math.pi * x[0] / 8) if i == 1 else 0.0
    mk = lambda s: oracle.Simulation((16, 16), (1.0, 0.0), 16, U=1, T=np.float64, perdir=(1, 2), u0=nonuniform, scheme=s)
    sq, sc = mk(oracle.QUICK), mk(oracle.CDS)
    sq.step(); sc.step()
    assert np.max(np.abs(sq.u - sc.u)) > 1e-6


# ------------------------------------------------------------------ test/test_flow.jl:100-109 + test/helper.jl:4-15 periodic TGV
def test_periodic_TGV_2d(oracle):
    L = 64
    T = np.float32
    kap = T(2 * math.pi / L)
    Re = T(1e8)
    nu = T(1 / (kap * Re))

    def TGV(i, x, t):
        xs, ys = x[0] * float(kap), x[1] * float(kap)
        dec = math.exp(-2 * float(kap) ** 2 * float(nu) * t)
        return -math.sin(xs) * math.cos(ys) * dec if i == 1 else math.cos(xs) * math.sin(ys) * dec

    dTGV = lambda i, x, t: -2 * float(kap) ** 2 * float(nu) * TGV(i, x, t)
    sim = oracle.Simulation((L, L), TGV, L, U=1, nu=float(nu), T=T, perdir=(1, 2), duBC_dt=dTGV)
    sim.step_until(math.pi / 100)
    t = sim.time()
    u = sim.u
    ue = np.zeros_like(u)
    for i in (1, 2):
        for a in range(u.shape[0]):
            for b in range(u.shape[1]):
                x = oracle.loc(i, (a + 1, b + 1))
                ue[a, b, i - 1] = TGV(i, x, t)
    assert oracle.L2(u[:, :, 0] - ue[:, :, 0]) < 1e-4
    assert oracle.L2(u[:, :, 1] - ue[:, :, 1]) < 1e-4


# ------------------------------------------------------------------ test/test_flow.jl:111-132 constant-jerk body force
def test_increasing_body_force(oracle):
    N = 8
    jerk = 4
    Us = math.sqrt(N)
    g = lambda i, x, t: t * jerk if i == 1 else 0.0
    sim = oracle.Simulation((N, N), (Us, 0.0), N, nu=0.001, g=g, dt=0.001, perdir=(1,), T=np.float64)
    sim.step_until(1.0)
    u = sim.u
    uFinal = Us + 0.5 * jerk * sim.time() ** 2
    assert oracle.L2(u[:, :, 0] - uFinal) < 1e-4 and oracle.L2(u[:, :, 1]) < 1e-4


# ------------------------------------------------------------------ test/test_flow.jl:134-140 boundary layer profile kept
def test_boundary_layer_profile(oracle):
    L = 32
    prof = lambda i, x, t: float(np.float32(4.0 * (((x[1] + 0.5) / (2 * L)) - ((x[1] + 0.5) / (2 * L)) ** 2))) if i == 1 else 0.0
    sim = oracle.Simulation((L, L), prof, L, nu=0.001, U=1, T=np.float32, duBC_dt=lambda i, x, t: 0.0)
    sim.step_until(10)
    u = sim.u
    assert np.allclose(u[0, :, 0], u[-1, :, 0], rtol=np.sqrt(np.finfo(np.float32).eps))


# ------------------------------------------------------------------ test/test_flow.jl:161-173 circle in accelerating flow
def test_circle_in_accelerating_flow(oracle):
    radius, H = 32, 16
    n = radius * 2 * H
    # uBC(i,x,t) = i==1 ? t : 0 via the oracle's native functor (a Python callback per cell is too slow at 1024²)
    sim = oracle.Simulation((n, n), "accel_x", radius, U=1, T=np.float32, body=("sphere", (H * radius, H * radius), radius), omp=True)
    sim.step()
    force = sim.pressure_force() / (math.pi * sim.L**2)
    assert np.allclose(force, [-1, 0], atol=0.04)
    u = sim.u
    assert u.max() / u[1, 1, 0] > 1.91
    for _ in range(3):
        sim.step()
    assert all(k <= 2 for k in sim.pois_n)


# ------------------------------------------------------------------ test/test_metrics.jl:35-40 pressure force of p=y on a circle
def test_pressure_force_linear_pressure(oracle):
    N = 32
    p = F((N, N), np.float64)
    for a in range(N):
        for b in range(N):
            p[a, b] = oracle.loc(0, (a + 1, b + 1))[1]
    p[0, :] = p[-1, :] = 0
    p[:, 0] = p[:, -1] = 0  # @inside only
    df = F((N, N, 2), np.float64)
    force = oracle.pressure_force(p, df, (N / 2, N / 2), N // 4)
    assert np.sum(np.abs(force / (math.pi * (N / 4) ** 2) - np.array([0, 1]))) < 2e-3


# ------------------------------------------------------------------ test/test_simulation.jl:15-22 sim_time stop rule
def test_sim_time_stop_rule(oracle):
    sim = oracle.Simulation((16, 16), (1, 0), 8, nu=0.1, body=("sphere", (8, 8), 3), T=np.float32)
    sim.step_until(1.0)
    assert sim.sim_time() >= 1.0
    dts = sim.dt
    assert sum(dts[:-2]) * sim.U / sim.L < 1.0 <= sum(dts[:-1]) * sim.U / sim.L
    assert len(sim.pois_n) == 2 * (len(dts) - 1)


# ------------------------------------------------------------------ test/test_metrics.jl:54-57 viscous force of a fluid at rest is zero
def test_viscous_force_zero_and_symmetry(oracle):
    N = 32
    for D in (2, 3):
        shape = (N,) * D
        c = (N / 2,) * D
        u = F(shape + (D,), np.float32)
        df = F(shape + (D,), np.float32)
        assert np.all(oracle.viscous_force(u, 1.0, df, c, N / 4) == 0)
        # pure shear u_x = y: S = [[0,1/2],[1/2,0]] — the force on a closed surface vanishes to the discrete symmetry of nds
        ax = np.arange(N, dtype=np.float32)
        u[..., 0] = ax.reshape((1, N) + (1,) * (D - 2))
        f = oracle.viscous_force(u, 1.0, df, c, N / 4)
        assert np.abs(f).max() < 1e-3 * (N / 4) ** (D - 1)


# ------------------------------------------------------------------ test/test_bodies.jl:6-7,15-18,42-43,51-56
def test_autobody_measure_known_answers(oracle):
    r2 = math.sqrt(2.0)
    # NoBody: measure == (Inf, 0, 0)   (:6-7) — the Flow-level measure! is a no-op for it (src/Body.jl:83)
    # circ(x,t) = |x| − 2; body1 = circ − t: at t the radius is 2+t
    d, n, V = oracle.body_measure(("sphere", (0, 0), 2.0), (r2, r2))
    assert abs(d) < 1e-12 and np.allclose(n, [math.sqrt(.5)] * 2) and np.allclose(V, 0)                   # :15
    d, n, V = oracle.body_measure(("sphere", (0, 0, 0), 3.0), (2.0, 0.0, 0.0))
    assert np.isclose(d, -1.0) and np.allclose(n, [1, 0, 0]) and np.allclose(V, 0)                         # :16  (t=1)
    # body2 = AutoBody(circ, (x,t)->x.+t²): ξ = x − c with c = −t², body velocity −2t
    d, n, V = oracle.body_measure(("sphere", (0, 0), 2.0, (0.0, 0.0)), (r2, r2))
    assert abs(d) < 1e-12 and np.allclose(n, [math.sqrt(.5)] * 2) and np.allclose(V, 0)                   # :17  (t=0)
    d, n, V = oracle.body_measure(("sphere", (-1, -1, -1), 2.0, (-2.0, -2.0, -2.0)), (1.0, -1.0, -1.0))
    assert abs(d) < 1e-12 and np.allclose(n, [1, 0, 0]) and np.allclose(V, [-2, -2, -2])                   # :18  (t=1)
    # fast version: outside fastd² only the distance is returned   (:42-43)
    full = oracle.body_measure(("sphere", (0, 0), 2.0), (3.0, 4.0))
    fast = oracle.body_measure(("sphere", (0, 0), 2.0), (3.0, 4.0), fastd2=9)
    assert np.isclose(full[0], 3.0) and np.isclose(fast[0], 3.0) and np.allclose(full[1], fast[1]) and np.allclose(full[1], [0.6, 0.8])
    cut = oracle.body_measure(("sphere", (0, 0), 2.0), (3.0, 4.0), fastd2=8)
    assert np.isclose(cut[0], 3.0) and np.allclose(cut[1], 0) and np.allclose(cut[2], 0)
    # RigidMap with a linear velocity (a rotation does not change the circle): (1/2, [1,0], [1,0])   (:51-56)
    for T in (np.float32, np.float64):
        d, n, V = oracle.body_measure(("sphere", (0, 0), 1.0, (1.0, 0.0)), (1.5, 0.0), T=T)
        assert np.isclose(d, 0.5) and np.allclose(n, [1, 0]) and np.allclose(V, [1, 0])


def test_cylinder_and_plane_bodies(oracle):
    """closed-form shapes beyond the sphere: a cylinder is the circle sdf with one axis masked out, a plane is n·(x−c) with the
    pseudo-sdf correction d/|∇f| (src/AutoBody.jl:34-35) — checked against their analytical distance / normal"""
    d, n, V = oracle.body_measure(("cylinder", (1.0, 2.0, 7.0), 2.0, 2), (4.0, 6.0, -3.0))
    assert np.isclose(d, 3.0) and np.allclose(n, [0.6, 0.8, 0.0])
    d, n, V = oracle.body_measure(("cylinder", (1.0, 2.0, 7.0), 2.0, 0), (40.0, 2.0, 10.0))
    assert np.isclose(d, 1.0) and np.allclose(n, [0, 0, 1])
    d, n, V = oracle.body_measure(("plane", (0.0, 1.0, 0.0), (0.0, 2.0, 0.0), (0.0, 0.5, 0.0)), (3.0, 4.0, 5.0))
    assert np.isclose(d, 3.0) and np.allclose(n, [0, 1, 0]) and np.allclose(V, [0, 0.5, 0])
    d, n, V = oracle.body_measure(("plane", (0.0, 0.0), (3.0, 4.0)), (1.0, 1.0))
    assert np.isclose(d, 7.0 / 5.0) and np.allclose(n, [0.6, 0.8])
    # measure! of a wall y < 4 in a 2-D box: μ₀ of the y-faces ramps 0 → 1 across the plane, nothing varies along x, V = 0
    sim = oracle.Simulation((16, 16), (1.0, 0.0), 16, nu=0.1, body=("plane", (0.0, 4.0), (0.0, 1.0)), T=np.float64)
    mu0 = sim.field("mu0")
    assert np.allclose(mu0[2:-2, :, 1], mu0[3:4, :, 1])          # (BC!(μ₀,0) only touches the boundary faces)
    col = mu0[5, 1:-1, 1]
    assert col[0] == 0 and col[-1] == 1 and np.all(np.diff(col) >= 0)
    assert np.isclose(mu0[5, 5, 1], 0.5)          # 0-based index 5 = I 6: its lower y-face sits at y = 6 − 1.5 − ½ = 4 → d = 0 → μ₀ = ½
    assert np.all(sim.field("V") == 0)


# ------------------------------------------------------------------ test/test_metrics.jl:58-66
def test_moments_known_answers(oracle):
    """viscous moment of a fluid at rest is zero; a hydrostatic pressure p = y exerts no moment about the body's centre (2-D: the
    scalar moment, 3-D: all three components); and, beyond the reference's tests, p = y about a point shifted by a in x gives the
    moment −a·F_y of the buoyancy force F = (0, πR², 0) applied at the centre (r×F with r = centre − x₀)."""
    N = 32
    body2, body3 = ("sphere", (N / 2, N / 2), N // 4), ("sphere", (N / 2, N / 2, N / 2), N // 4)
    u2, u3 = F((N, N, 2), np.float64), F((N, N, N, 3), np.float64)
    df2, df3 = F((N, N, 2), np.float64), F((N, N, N, 3), np.float64)
    assert np.all(oracle.viscous_moment_body((N / 2, N / 2), u2, 1.0, df2, body2) == 0)                     # :60
    assert np.all(oracle.viscous_moment_body((N / 2,) * 3, u3, 1.0, df3, body3) == 0)                       # :61
    p2, p3 = F((N, N), np.float64), F((N, N, N), np.float64)
    for b in range(1, N - 1):
        p2[1:-1, b] = oracle.loc(0, (2, b + 1))[1]
        p3[1:-1, b, 1:-1] = oracle.loc(0, (2, b + 1, 2))[1]
    m2 = oracle.pressure_moment_body((N / 2, N / 2), p2, df2, body2)
    assert abs(m2[0]) < 1e-9 * N**3                                                                         # :65
    m3 = oracle.pressure_moment_body((N / 2,) * 3, p3, df3, body3)
    assert np.all(np.abs(m3) < 1e-9 * N**4)                                                                 # :66
    a = 3.0
    Fy = oracle.pressure_force_body(p2, df2, body2)[1]
    m2s = oracle.pressure_moment_body((N / 2 - a, N / 2), p2, df2, body2)
    assert np.isclose(m2s[0], a * Fy, rtol=1e-9) and m2s[0] == m2s[1]          # r = (a,0), F = (0,Fy): r×F = a·Fy; 2-D: scalar in both slots
