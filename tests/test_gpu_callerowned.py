"""wl_sim_create on CALLER-OWNED arrays — the call sequence of the Julia binding (waterlily.jl_amd/julia/WaterLilyHIPExt.jl:
`Simulation(...; mem=HipArray)` builds the fields, `pois_ctor` hands them to wl_sim_create, `mom_step!(::HFlow, ::HipMultiLevel)` calls
wl_sim_mom_step) — against the handle-owned FusedSimulation, bit for bit.  Both ownership modes of include/wlhip.h (wl_sim_desc.us):
spare array given ⇒ the roles of {u,u0,us} rotate and are read back after every step; no spare ⇒ pointers never move."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def w():
    import waterlily_jl_amd as w
    w.core.device()
    return w


class CallerOwnedSim:
    """What the Julia side does, in Python: allocate the Flow fields as device arrays (Flow ctor, src/Flow.jl:133-147), create the
    composite on them, step, follow the array roles."""

    def __init__(self, w, dims, uBC, nu, u_init, with_spare, body=None, exitBC=False):
        from waterlily_jl_amd._lib import check, lib, wl_sim_desc
        self.w, self.check, self.lib = w, check, lib()
        D = len(dims)
        Ng = tuple(n + 2 for n in dims)
        self.D, self.Ng = D, Ng
        self.arr = {"u": w.to_device(u_init), "f": w.jl_zeros(Ng + (D,)), "p": w.jl_zeros(Ng), "sigma": w.jl_zeros(Ng),
                    "V": w.jl_zeros(Ng + (D,)), "mu0": w.jl_zeros(Ng + (D,), 1.0), "mu1": w.jl_zeros(Ng + (D, D))}
        w.BC_(self.arr["u"], uBC, exitBC)                                   # BC!(u,uBC,exitBC,perdir)      :141
        if True:
            w.exitBC_(self.arr["u"], self.arr["u"], 0.0)                    # exitBC!(u,u,zero(T))          :141
        self.arr["u0"] = self.arr["u"].clone()                              # u⁰ = copy(u)                  :142
        w.BC_(self.arr["mu0"], (0,) * D)                                    # BC!(μ₀,0)                     :145
        if with_spare:
            self.arr["us"] = w.jl_zeros(Ng + (D,))
        self.role = {k: k for k in ("u", "u0", "us") if k in self.arr}      # role -> buffer name
        if body is not None:                                                # measure!(flow,body)           src/WaterLily.jl:104
            import waterlily_jl_amd.simulation as S
            from waterlily_jl_amd._lib import make_body
            from waterlily_jl_amd.core import ptr, sgrid, stream
            b = make_body(body, D)
            g = sgrid(self.arr["sigma"])
            check(self.lib.wl_measure_body(ptr(self.arr["sigma"]), ptr(self.arr["mu0"]), ptr(self.arr["mu1"]), ptr(self.arr["V"]), C.byref(g), C.byref(b), 1.0, int(exitBC), 0, stream()))
        d = wl_sim_desc()
        d.D = D
        for k in range(3):
            d.dims[k] = dims[k] if k < D else 1
            d.uBC[k] = float(uBC[k]) if k < D else 0.0
        d.nu, d.dt0, d.perdir_mask, d.exitBC, d.scheme, d.has_body = float(nu), 0.25, 0, int(exitBC), 0, int(body is not None)
        for name in ("u", "u0", "f", "p", "sigma", "V", "mu0", "mu1") + (("us",) if with_spare else ()):
            setattr(d, name, w.core.ptr(self.arr[name]).value)
        # pois_ctor(flow) = MultiLevelPoisson(flow.p, flow.μ₀, flow.σ) -> wl_mg_create; the composite adopts it at the first mom_step!
        from waterlily_jl_amd.core import sgrid
        g0 = sgrid(self.arr["p"])
        mg = C.c_void_p()
        check(self.lib.wl_mg_create(C.byref(mg), w.core.ptr(self.arr["p"]), w.core.ptr(self.arr["mu0"]), w.core.ptr(self.arr["sigma"]), C.byref(g0), 0, 10))
        self.mg = mg
        h = C.c_void_p()
        check(self.lib.wl_sim_create_on(C.byref(h), C.byref(d), mg))
        assert self.lib.wl_sim_pois(h) == mg.value
        self.h = h
        self.dt = [np.float32(0.25)]
        self._ptr2name = {self.arr[k].data_ptr(): k for k in self.role}

    def mom_step(self):
        from waterlily_jl_amd.core import stream
        self.check(self.lib.wl_sim_set_dt_last(self.h, float(self.dt[-1])))  # the host owns flow.Δt
        self.check(self.lib.wl_sim_mom_step(self.h, stream()))
        self.dt.append(np.float32(self.lib.wl_sim_dt_last(self.h)))          # push!(a.Δt, CFL(a))
        for role in self.role:                                               # re-point u / u⁰ / spare (HipArray.ptr on the Julia side)
            self.role[role] = self._ptr2name[self.lib.wl_sim_field(self.h, role.encode())]

    def field(self, name):
        buf = self.role.get(name, name)
        return self.w.to_host(self.arr[buf])

    def pois_n(self):
        out = (C.c_int16 * 4096)()
        k = self.lib.wl_mg_history(self.lib.wl_sim_pois(self.h), out, 4096)
        return [int(v) for v in out[:k]]

    def close(self):
        self.check(self.lib.wl_sim_destroy(self.h))          # the wl_sim first: it uses the wl_mg
        self.check(self.lib.wl_mg_destroy(self.mg))


@pytest.mark.parametrize("dims", [(64, 64, 64), (96, 48, 40), (48, 40)])
@pytest.mark.parametrize("with_spare", [True, False])
def test_caller_owned_arrays_equal_handle_owned(w, dims, with_spare):
    rng = np.random.default_rng(5)
    D = len(dims)
    Ng = tuple(n + 2 for n in dims)
    uBC = (1.0,) + (0.0,) * (D - 1)
    u_init = np.asfortranarray(rng.uniform(-0.3, 0.3, size=Ng + (D,)).astype(np.float32))
    u_init[..., 0] += 1.0
    ref = w.FusedSimulation(dims, uBC, dims[0], U=1, nu=0.02, u0=u_init)
    sim = CallerOwnedSim(w, dims, uBC, 0.02, u_init, with_spare)
    moved = False
    for step in range(3):
        ref.mom_step_(); sim.mom_step()
        assert np.array_equal(sim.field("u"), ref.field("u")), step
        assert np.array_equal(sim.field("u0"), ref.field("u0")), step
        assert np.array_equal(sim.field("p"), ref.field("p")), step
        assert [float(v) for v in sim.dt] == [float(v) for v in ref.dt]
        moved = moved or sim.role["u"] != "u"
    assert sim.pois_n() == ref.pois_n
    if with_spare and D == 3:
        assert moved                       # the fused out-of-place kernels ran: the roles of the three buffers rotated
    if not with_spare:
        assert sim.role == {"u": "u", "u0": "u0"}      # pointer-stable mode
    sim.close()


def test_caller_owned_arrays_with_a_body(w):
    dims, R = (64, 48, 48), 6.0
    c = (dims[0] / 4, dims[1] / 2 - 1, dims[2] / 2 - 1)
    Ng = tuple(n + 2 for n in dims)
    u_init = np.zeros(Ng + (3,), dtype=np.float32, order="F")
    u_init[..., 0] = 1.0
    ref = w.FusedSimulation(dims, (1.0, 0, 0), 2 * R, U=1, nu=2 * R / 250, has_body=True, u0=u_init)
    ref.measure_sphere_(c, R, 1.0)
    for with_spare in (True, False):
        sim = CallerOwnedSim(w, dims, (1.0, 0, 0), 2 * R / 250, u_init, with_spare, body=("sphere", c, R))
        from waterlily_jl_amd.core import stream
        sim.check(sim.lib.wl_sim_update(sim.h, stream()))       # update!(pois) after measure!
        r2 = w.FusedSimulation(dims, (1.0, 0, 0), 2 * R, U=1, nu=2 * R / 250, has_body=True, u0=u_init)
        r2.measure_sphere_(c, R, 1.0)
        for step in range(2):
            r2.mom_step_(); sim.mom_step()
            assert np.array_equal(sim.field("u"), r2.field("u")) and np.array_equal(sim.field("p"), r2.field("p")), (with_spare, step)
        assert sim.pois_n() == r2.pois_n
        sim.close()


@pytest.mark.parametrize("dims", [(64, 48, 40), (64, 48)])
def test_convective_exit_rotates_the_buffers_and_matches_oracle(w, oracle, dims):
    """exitBC=true (src/core.jl:226-233; BC! leaves the exit face alone, :207): the handle-owned flow and a caller-owned flow WITH a spare
    array rotate their velocity buffers (the exit face travels with the role); a caller-owned flow WITHOUT the spare copies `u⁰ .= u`.
    All three give the same bits, and they match the oracle to the summation order of the two face means."""
    rng = np.random.default_rng(11)
    D = len(dims)
    Ng = tuple(n + 2 for n in dims)
    uBC = (1.0,) + (0.0,) * (D - 1)
    u_init = np.asfortranarray(rng.uniform(-0.2, 0.2, size=Ng + (D,)).astype(np.float32))
    u_init[..., 0] += 1.0
    ui = u_init.copy(order="F")
    oracle.BC(ui, uBC, True, ())                           # Flow(): BC!(u,uBC,exitBC,perdir); exitBC!(u,u,zero(T))   src/Flow.jl:141
    oracle.exitBC(ui, ui.copy(order="F"), 0.0)
    so = oracle.Simulation(dims, uBC, dims[0], U=1, nu=0.02, exitBC=True, T=np.float32)
    so.field("u")[...] = ui; so.field("u0")[...] = ui
    ref = w.FusedSimulation(dims, uBC, dims[0], U=1, nu=0.02, exitBC=True, u0=u_init)
    ref.set_field("u", ui); ref.set_field("u0", ui)       # every flow starts from the same bits (the face means of exitBC! are reductions)
    assert w.lib().wl_sim_field(ref._h, b"us")            # the handle owns a spare array although the exit is convective
    sims = {sp: CallerOwnedSim(w, dims, uBC, 0.02, u_init, sp, exitBC=True) for sp in (True, False)}
    for sim in sims.values():
        sim.arr["u"].copy_(w.to_device(ui)); sim.arr["u0"].copy_(w.to_device(ui))
    moved = False
    for step in range(4):
        so.step(remeasure=False); ref.mom_step_()
        for sp, sim in sims.items():
            sim.mom_step()
            assert np.array_equal(sim.field("u"), ref.field("u")), (sp, step)
            assert np.array_equal(sim.field("u0"), ref.field("u0")), (sp, step)
            assert np.array_equal(sim.field("p"), ref.field("p")), (sp, step)
        moved = moved or sims[True].role["u"] != "u"
        assert ref.pois_n == so.pois_n
        assert np.abs(ref.field("u") - so.u).max() < 2e-5, step
    assert moved and sims[False].role == {"u": "u", "u0": "u0"}
    for sim in sims.values():
        assert sim.pois_n() == ref.pois_n
        sim.close()
