"""Simulation — host-side mirror of /root/reference/src/WaterLily.jl:86-149, plus the fused composite
handle (`wl_sim`) that bench.py times."""
import ctypes as C

import numpy as np

from . import core
from ._lib import check, lib, wl_grid, wl_sim_desc
from .core import perdir_mask, ptr, stream
from .flow import Flow, mom_step_
from .poisson import MultiLevelPoisson


def measure_(a, body, eps=1.0):
    """measure!(a::Flow, body; ϵ) for a closed-form AutoBody   src/Body.jl:28-51"""
    from ._lib import make_body
    from .core import ptr, sgrid, perdir_mask
    b = make_body(body, a.D)
    g = sgrid(a.sigma)
    check(lib().wl_measure_body(ptr(a.sigma), ptr(a.mu0), ptr(a.mu1), ptr(a.V), C.byref(g), C.byref(b), float(eps), int(a.exitBC), perdir_mask(a.perdir), stream()))
    a.has_body = True


def pressure_force(a, body):
    """pressure_force(p,df,body)   src/Metrics.jl:116-133 (Float64 sums; flow.f is not used as scratch)"""
    from ._lib import make_body
    from .core import ptr, sgrid
    b = make_body(body, a.D)
    g = sgrid(a.p)
    out = (C.c_double * 3)()
    check(lib().wl_pressure_force_body(ptr(a.p), C.byref(g), C.byref(b), out, stream()))
    return np.array(out[: a.D])


def viscous_force(a, body):
    """viscous_force(u,ν,df,body)   src/Metrics.jl:140-154"""
    from ._lib import make_body
    from .core import ptr, sgrid
    b = make_body(body, a.D)
    g = sgrid(a.p)
    out = (C.c_double * 3)()
    check(lib().wl_viscous_force_body(ptr(a.u), C.byref(g), float(a.nu), C.byref(b), out, stream()))
    return np.array(out[: a.D])


def pressure_moment(x0, a, body):
    """pressure_moment(x₀,flow,body)   src/Metrics.jl:168-174"""
    from ._lib import make_body
    from .core import ptr, sgrid
    b = make_body(body, a.D)
    g = sgrid(a.p)
    out = (C.c_double * 3)()
    xx = (C.c_float * 3)(*([float(v) for v in x0] + [0.0] * (3 - a.D)))
    check(lib().wl_pressure_moment_body(xx, ptr(a.p), C.byref(g), C.byref(b), out, stream()))
    return np.array(out[: a.D])


def viscous_moment(x0, a, body):
    """viscous_moment(x₀,flow,body)   src/Metrics.jl:182-188"""
    from ._lib import make_body
    from .core import ptr, sgrid
    b = make_body(body, a.D)
    g = sgrid(a.p)
    out = (C.c_double * 3)()
    xx = (C.c_float * 3)(*([float(v) for v in x0] + [0.0] * (3 - a.D)))
    check(lib().wl_viscous_moment_body(xx, ptr(a.u), C.byref(g), float(a.nu), C.byref(b), out, stream()))
    return np.array(out[: a.D])


class Simulation:
    """Simulation(dims,uBC,L;U,Δt,ν,ϵ,perdir,exitBC,λ,body,T) over leaf operations (reference orchestration)."""

    def __init__(self, dims, uBC, L, U=None, dt=0.25, nu=0.0, eps=1, perdir=(), u0=None, exitBC=False, lam=core.QUICK,
                 body=None, T=np.float32, g=None, duBC_dt=None):
        if U is None:
            assert not callable(uBC), "`U` (velocity scale) must be specified if boundary conditions `uBC` is a `Function`"   # :99
            U = float(np.sqrt(sum(float(v) ** 2 for v in uBC)))                  # :100
        self.U, self.L, self.eps = float(U), float(L), eps
        self.flow = Flow(dims, uBC, dt=dt, nu=nu, g=g, u0=u0, perdir=perdir, exitBC=exitBC, lam=lam, T=T, duBC_dt=duBC_dt)   # :103
        self.body = body          # None (NoBody) or a closed-form AutoBody: ("sphere", c, R) | ("cylinder", c, R, axis) | ("plane", point, normal) [+ velocity]
        if body is not None:
            measure_(self.flow, body, eps=self.eps)                                                 # :104
        self.pois = MultiLevelPoisson(self.flow.p, self.flow.mu0, self.flow.sigma, perdir=perdir)   # :97,105

    def sim_time(self):
        return float(self.flow.time()) * self.U / self.L                          # :117

    def sim_step_(self, t_end=None, remeasure=True, max_steps=2**31 - 1):
        """sim_step!(sim[,t_end];remeasure,max_steps)   :128-139"""
        if t_end is None:
            if remeasure:
                self.measure_()
            mom_step_(self.flow, self.pois)
            return
        steps0 = len(self.flow.dt)
        while self.sim_time() < t_end and len(self.flow.dt) - steps0 < max_steps:
            self.sim_step_(remeasure=remeasure)

    def measure_(self, body=None):
        """measure!(sim): measure!(flow,body) + update!(pois)   :146-149 (quirk Q3: runs every step when remeasure=true; NoBody => only
        update!).  `body`: the body's new description (position, velocity) — the stand-in for the reference's map(x,t)."""
        if body is not None:
            self.body = body
        if self.body is not None:
            measure_(self.flow, self.body, eps=self.eps)
        self.pois.update_()

    def pressure_force(self):
        return pressure_force(self.flow, self.body)

    def viscous_force(self):
        return viscous_force(self.flow, self.body)

    def total_force(self):
        """total_force(sim) = pressure_force + viscous_force   src/Metrics.jl:156-161"""
        return self.pressure_force() + self.viscous_force()

    def pressure_moment(self, x0):
        return pressure_moment(x0, self.flow, self.body)

    def viscous_moment(self, x0):
        return viscous_moment(x0, self.flow, self.body)

    def total_moment(self, x0):
        """total_moment(x₀,sim)   src/Metrics.jl:195"""
        return self.pressure_moment(x0) + self.viscous_moment(x0)


class FusedSimulation:
    """The composite path: one `wl_sim` handle holds every field in HBM and runs mom_step! (src/Flow.jl:156-167)
    as a fixed sequence of fused HIP kernels on one stream.  This is what bench.py times."""

    def __init__(self, dims, uBC, L, U=None, dt=0.25, nu=0.0, perdir=(), exitBC=False, lam=core.QUICK, has_body=False, ic="uBC", u0=None,
                 g=None, duBC_dt=None):
        """uBC: tuple, or a function uBC(i,t) that is uniform in space (i = 1..D as in the reference) together with its time
        derivative duBC_dt(i,t) (the reference differentiates it with ForwardDiff, src/Flow.jl:72-73); g: body force g(i,t),
        uniform in space.  Position-dependent uBC/g are not a device path."""
        core.device()
        D = len(dims)
        self._ufn = uBC if callable(uBC) else None
        self._dufn, self._gfn = duBC_dt, g
        if self._ufn is not None:
            assert U is not None, "`U` (velocity scale) must be specified if boundary conditions `uBC` is a `Function`"   # src/WaterLily.jl:99
            uBC = tuple(float(self._ufn(i + 1, 0.0)) for i in range(D))
        if U is None:
            U = float(np.sqrt(sum(float(v) ** 2 for v in uBC)))
        self.U, self.L, self.D = float(U), float(L), D
        self.dims = tuple(int(n) for n in dims)
        self.Ng = tuple(n + 2 for n in self.dims)
        d = wl_sim_desc()
        d.D = D
        for k in range(3):
            d.dims[k] = self.dims[k] if k < D else 1
            d.uBC[k] = float(uBC[k]) if k < D else 0.0
        d.nu, d.dt0 = float(nu), float(dt)
        d.perdir_mask, d.exitBC, d.scheme, d.has_body = perdir_mask(perdir), int(bool(exitBC)), int(lam), int(bool(has_body))
        h = C.c_void_p()
        check(lib().wl_sim_create(C.byref(h), C.byref(d)))
        self._h = h
        self.nu = float(nu)
        self.has_body = bool(has_body)
        if u0 is not None:
            self.set_field("u", np.asfortranarray(u0, dtype=np.float32))
        else:
            check(lib().wl_sim_apply_ic(h, {"uBC": 0, "tgv": 1, "tgv_periodic": 2}[ic], stream()))
        check(lib().wl_sim_init_flow(h, stream()))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().wl_sim_destroy(h)
            except Exception:
                pass
            self._h = None

    def _shape(self, name):
        D = self.D
        return {"p": self.Ng, "sigma": self.Ng, "mu1": self.Ng + (D, D)}.get(name, self.Ng + (D,))

    def field(self, name):
        """`Array(flow.<name>)`"""
        out = np.empty(self._shape(name), dtype=np.float32, order="F")
        p = lib().wl_sim_field(self._h, name.encode())
        if not p:
            raise KeyError(name)
        check(lib().wl_d2h(out.ctypes.data_as(C.c_void_p), p, out.nbytes, stream()))
        return out

    def set_field(self, name, a):
        a = np.asfortranarray(a, dtype=np.float32)
        assert a.shape == self._shape(name)
        check(lib().wl_h2d(lib().wl_sim_field(self._h, name.encode()), a.ctypes.data_as(C.c_void_p), a.nbytes, stream()))
        check(lib().wl_stream_sync(stream()))

    def _forcing(self):
        """uBC(i,t₁) and g(i,t)+dU(i,t)/dt at t₀, t₁ for the coming step (mom_step!, src/Flow.jl:157; accelerate! :69-73)"""
        if self._ufn is None and self._gfn is None:
            return
        D = self.D
        dtl = np.float32(lib().wl_sim_dt_last(self._h))
        t1 = np.float32(np.float32(lib().wl_sim_time(self._h)) + dtl)      # t₁ = sum(Δt) ; t₀ = t₁ - Δt[end]   src/Flow.jl:157
        t0 = np.float32(t1 - dtl)
        arr = lambda v: (C.c_float * 3)(*([float(x) for x in v] + [0.0] * (3 - D)))
        U1 = arr([self._ufn(i + 1, float(t1)) for i in range(D)]) if self._ufn is not None else None
        acc = lambda t: arr([(self._gfn(i + 1, float(t)) if self._gfn else 0.0) + (self._dufn(i + 1, float(t)) if (self._ufn and self._dufn) else 0.0) for i in range(D)])
        check(lib().wl_sim_set_forcing(self._h, U1, acc(t0), acc(t1)))

    def mom_step_(self):
        self._forcing()
        check(lib().wl_sim_mom_step(self._h, stream()))

    def mom_steps_(self, n):
        """n × mom_step! in one library call (no time-dependent uBC / g between the steps: those are evaluated by the host per step — use mom_step_)"""
        if self._ufn is not None or self._gfn is not None:
            for _ in range(int(n)):
                self.mom_step_()
            return
        check(lib().wl_sim_mom_steps(self._h, int(n), stream()))

    def phase_(self, k):
        check(lib().wl_sim_phase(self._h, int(k), stream()))

    def sim_step_(self, t_end=None, remeasure=False, max_steps=2**31 - 1):
        if t_end is None:
            if remeasure:
                check(lib().wl_sim_update(self._h, stream()))
            self.mom_step_()
            return
        n = 0
        while self.sim_time() < t_end and n < max_steps:
            self.sim_step_(remeasure=remeasure)
            n += 1

    def update_(self):
        check(lib().wl_sim_update(self._h, stream()))

    def set_option(self, name, value):
        """implementation switches: "convz", "fused_smoother" (1 = default fast path, 0 = one kernel per pass)"""
        check(lib().wl_sim_set_option(self._h, name.encode(), int(value)))

    def counter(self, name):
        """path counters of the handle (include/wlhip_bench.h wl_sim_counter): "resjac", "resjac_redo", "resjac_backoff", "xdefer", "tailfuse", "bcdefer", "tailspec" """
        v = C.c_long(0)
        check(lib().wl_sim_counter(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    @property
    def dt(self):
        out = (C.c_float * 1000000)()
        k = lib().wl_sim_dt(self._h, out, 1000000)
        return [np.float32(v) for v in out[:k]]

    def time(self):
        return lib().wl_sim_time(self._h)

    def sim_time(self):
        return self.time() * self.U / self.L

    @property
    def pois_n(self):
        mg = lib().wl_sim_pois(self._h)
        out = (C.c_int16 * 65536)()
        k = lib().wl_mg_history(mg, out, 65536)
        return [int(v) for v in out[:k]]

    def pois_level(self, name, l=0):
        mg = lib().wl_sim_pois(self._h)
        g = wl_grid()
        check(lib().wl_mg_level_grid(mg, l, C.byref(g)))
        dims = (g.nx, g.ny) if g.D == 2 else (g.nx, g.ny, g.nz)
        shape = dims + (g.D,) if name == "L" else dims
        out = np.empty(shape, dtype=np.float32, order="F")
        check(lib().wl_d2h(out.ctypes.data_as(C.c_void_p), lib().wl_mg_level_field(mg, l, name.encode()), out.nbytes, stream()))
        return out

    def nlevels(self):
        return lib().wl_mg_nlevels(lib().wl_sim_pois(self._h))

    def const_levels(self):
        """per level: was L verified to be 'constant inside, zero on wall faces' (constant-coefficient kernels in use)"""
        mg = lib().wl_sim_pois(self._h)
        return [bool(lib().wl_mg_level_is_const(mg, l)) for l in range(self.nlevels())]

    def smoother_kinds(self):
        """per level: 0 one kernel per pass, 1 temporally blocked smoother, 2 blocked pair kernels (constant coefficients),
        3 z-split (pair kernels on the planes away from the body, general blocked kernels around it)"""
        mg = lib().wl_sim_pois(self._h)
        return [int(lib().wl_mg_smoother_kind(mg, l)) for l in range(self.nlevels())]

    def measure_sphere_(self, center, R, eps=1.0):
        """measure!(sim) for AutoBody(|x-c|-R): closed form on device + update!(pois)"""
        c = (C.c_float * 3)(*([float(v) for v in center] + [0.0] * (3 - self.D)))
        check(lib().wl_sim_measure_sphere(self._h, c, float(R), float(eps), stream()))

    def measure_body_(self, body, eps=1.0):
        """measure!(sim) for a closed-form AutoBody — ("sphere", c, R) | ("cylinder", c, R, axis) | ("plane", point, normal), optionally
        followed by the body's translation velocity (stored in flow.V, src/AutoBody.jl:36-37) — on device + update!(pois)"""
        from ._lib import make_body
        b = make_body(body, self.D)
        check(lib().wl_sim_measure_body(self._h, C.byref(b), float(eps), stream()))

    def pressure_force_body(self, body):
        from ._lib import make_body
        b = make_body(body, self.D)
        out = (C.c_double * 3)()
        check(lib().wl_sim_pressure_force_body(self._h, C.byref(b), out, stream()))
        return np.array(out[: self.D])

    def viscous_force_body(self, body):
        from ._lib import make_body
        b = make_body(body, self.D)
        out = (C.c_double * 3)()
        check(lib().wl_sim_viscous_force_body(self._h, C.byref(b), out, stream()))
        return np.array(out[: self.D])

    def total_force_body(self, body):
        return self.pressure_force_body(body) + self.viscous_force_body(body)

    def _moment(self, fn, x0, body):
        from ._lib import make_body
        b = make_body(body, self.D)
        out = (C.c_double * 3)()
        xx = (C.c_float * 3)(*([float(v) for v in x0] + [0.0] * (3 - self.D)))
        check(fn(self._h, xx, C.byref(b), out, stream()))
        return np.array(out[: self.D])

    def pressure_moment_body(self, x0, body):
        """pressure_moment(x₀,sim)   src/Metrics.jl:167-174"""
        return self._moment(lib().wl_sim_pressure_moment_body, x0, body)

    def viscous_moment_body(self, x0, body):
        """viscous_moment(x₀,sim)   src/Metrics.jl:181-188"""
        return self._moment(lib().wl_sim_viscous_moment_body, x0, body)

    def total_moment_body(self, x0, body):
        return self.pressure_moment_body(x0, body) + self.viscous_moment_body(x0, body)

    def pressure_force_sphere(self, center, R):
        c = (C.c_float * 3)(*([float(v) for v in center] + [0.0] * (3 - self.D)))
        out = (C.c_double * 3)()
        check(lib().wl_sim_pressure_force_sphere(self._h, c, float(R), out, stream()))
        return np.array(out[: self.D])

    def viscous_force_sphere(self, center, R):
        """viscous_force(sim) for the sphere/circle   src/Metrics.jl:148-154"""
        c = (C.c_float * 3)(*([float(v) for v in center] + [0.0] * (3 - self.D)))
        out = (C.c_double * 3)()
        check(lib().wl_sim_viscous_force_sphere(self._h, c, float(R), out, stream()))
        return np.array(out[: self.D])

    def total_force_sphere(self, center, R):
        """total_force(sim) = pressure_force + viscous_force   src/Metrics.jl:156-161"""
        return self.pressure_force_sphere(center, R) + self.viscous_force_sphere(center, R)

    def sync(self):
        check(lib().wl_stream_sync(stream()))
