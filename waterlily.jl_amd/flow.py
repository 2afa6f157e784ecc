"""Flow — host-side mirror of /root/reference/src/Flow.jl over the HIP C ABI."""
import ctypes as C

import numpy as np

from . import core
from ._lib import check, lib
from .core import BC_, exitBC_, jl_zeros, perdir_mask, ptr, sgrid, stream, vgrid


def conv_diff_(r, u, Phi, lam=core.QUICK, nu=0.1, perdir=()):
    """conv_diff!(r,u,Φ,λ;ν,perdir)   src/Flow.jl:38-62"""
    g = vgrid(u)
    check(lib().wl_conv_diff(ptr(r), ptr(u), ptr(Phi), C.byref(g), float(nu), perdir_mask(perdir), int(lam), stream()))


def scale_u_(a, scale):
    """scale_u!(a,scale)   src/Flow.jl:211-214"""
    g = vgrid(a.u)
    check(lib().wl_scale_u(ptr(a.u), C.byref(g), float(scale), stream()))


def BDIM_(a, pre=1.0, post=1.0):
    """BDIM!(a)   src/Flow.jl:176-180 (pre/post fold the neighbouring scale_u! calls; defaults = plain BDIM!)"""
    g = vgrid(a.u)
    V = ptr(a.V) if a.has_body else None
    mu1 = ptr(a.mu1) if a.has_body else None
    check(lib().wl_bdim(ptr(a.u), ptr(a.u0), ptr(a.f), V, ptr(a.mu0), mu1, C.byref(g), float(a.dt[-1]), float(pre), float(post), stream()))


def CFL(a, dt_max=10):
    """CFL(a)   src/Flow.jl:234-237"""
    out = C.c_float()
    g = sgrid(a.sigma)
    check(lib().wl_cfl(ptr(a.u), ptr(a.sigma), C.byref(g), float(a.nu), float(dt_max), C.byref(out), stream()))
    return np.float32(out.value)


class Flow:
    """Flow(N,uBC;Δt,ν,u0,perdir,exitBC,λ)   src/Flow.jl:114-148 — fields live in HBM (float32)."""

    def __init__(self, N, uBC, dt=0.25, nu=0.0, g=None, u0=None, perdir=(), exitBC=False, lam=core.QUICK, T=np.float32, duBC_dt=None):
        """uBC: tuple or function uBC(i,x,t); g: None or function g(i,x,t); duBC_dt(i,x,t): the time derivative of a function uBC
        (the reference obtains it with ForwardDiff, src/Flow.jl:72-73).  Functions are tabulated on the host each time they are
        needed (wl_bc_vec_fn / wl_accelerate_field) — correct for any closure, meant for small problems."""
        if np.dtype(T) != np.float32:
            raise NotImplementedError("the HIP path computes in Float32")
        D = len(N)
        self.D, self.N = D, tuple(int(n) for n in N)
        Ng = tuple(n + 2 for n in self.N)
        self.Ng = Ng
        self.uBC = uBC if callable(uBC) else tuple(float(v) for v in uBC)
        self.duBC_dt = duBC_dt
        self.dt = [np.float32(dt)]
        self.nu = np.float32(nu)
        self.g = g
        self.exitBC = bool(exitBC)
        self.perdir = tuple(perdir)
        self.lam = lam
        # u = Array{T}(undef, Nd...) |> mem; apply!(u0,u)                 :139-140
        if u0 is None and callable(self.uBC):
            u0 = lambda i, x: self.uBC(i, x, 0.0)        # Simulation: u0 defaults to uBC at t=0   src/WaterLily.jl:100
        if u0 is None:
            u_host = np.empty(Ng + (D,), dtype=np.float32, order="F")
            for i in range(D):
                u_host[..., i] = self.uBC[i]
        elif callable(u0):
            u_host = np.empty(Ng + (D,), dtype=np.float32, order="F")
            for i in range(1, D + 1):
                for I in np.ndindex(*Ng):
                    u_host[I + (i - 1,)] = u0(i, core.loc(i, tuple(k + 1 for k in I)))
        else:
            u_host = np.asfortranarray(u0, dtype=np.float32)   # pre-evaluated initial field
            assert u_host.shape == Ng + (D,)
        self.u = core.to_device(u_host)
        BC_(self.u, self.uBC, self.exitBC, self.perdir)               # :141
        if self.exitBC:
            exitBC_(self.u, self.u, 0.0)
        self.u0 = jl_zeros(Ng + (D,))
        check(lib().wl_d2d(ptr(self.u0), ptr(self.u), 4 * self.u.numel(), stream()))   # u⁰ = copy(u)  :142
        self.f, self.p, self.sigma = jl_zeros(Ng + (D,)), jl_zeros(Ng), jl_zeros(Ng)   # :143
        self.V, self.mu0, self.mu1 = jl_zeros(Ng + (D,)), jl_zeros(Ng + (D,), 1.0), jl_zeros(Ng + (D, D))   # :144
        BC_(self.mu0, (0.0,) * D, False, self.perdir)                 # :145
        self.has_body = False

    def time(self):
        """time(a) = sum(Δt[1:end-1])   :174"""
        s = np.float32(0)
        for d in self.dt[:-1]:
            s = np.float32(s + d)
        return s


def _times(a):
    """t₁ = sum(Δt), t₀ = t₁ - Δt[end]   src/Flow.jl:157"""
    t1 = np.float32(0)
    for d in a.dt:
        t1 = np.float32(t1 + d)
    return np.float32(t1 - a.dt[-1]), t1


def mom_predict_(a, t0=0.0, t1=0.0):
    """mom_predict!(a,t₀,t₁)   src/Flow.jl:190-196"""
    conv_diff_(a.f, a.u0, a.sigma, a.lam, nu=a.nu, perdir=a.perdir)
    core.accelerate_(a.f, float(t0), a.g, a.uBC, a.duBC_dt)
    BDIM_(a)
    BC_(a.u, a.uBC, a.exitBC, a.perdir, float(t1))
    if a.exitBC:
        exitBC_(a.u, a.u0, a.dt[-1])


def mom_correct_(a, t=0.0):
    """mom_correct!(a,t)   src/Flow.jl:205-210"""
    conv_diff_(a.f, a.u, a.sigma, a.lam, nu=a.nu, perdir=a.perdir)
    core.accelerate_(a.f, float(t), a.g, a.uBC, a.duBC_dt)
    BDIM_(a)
    scale_u_(a, 0.5)
    BC_(a.u, a.uBC, a.exitBC, a.perdir, float(t))


def mom_project_(a, b, w, t=0.0):
    """mom_project!(a,b,w,t)   src/Flow.jl:223-232"""
    dt = np.float32(np.float32(w) * a.dt[-1])
    g = sgrid(b.x)
    n = int(np.prod(b.x.shape))
    check(lib().wl_div(ptr(b.z), ptr(a.u), C.byref(g), stream()))          # @inside b.z[I] = div(I,a.u)
    check(lib().wl_scale(ptr(b.x), float(dt), n, stream()))                 # b.x .*= dt
    b.solver_()
    check(lib().wl_project(ptr(a.u), ptr(b.L), ptr(b.x), C.byref(g), stream()))
    check(lib().wl_div_scalar(ptr(b.x), float(dt), n, stream()))            # b.x ./= dt
    BC_(a.u, a.uBC, a.exitBC, a.perdir, float(t))


def mom_step_(a, b):
    """mom_step!(a::Flow,b::AbstractPoisson)   src/Flow.jl:156-167 — written on the leaf operations, line for line."""
    check(lib().wl_d2d(ptr(a.u0), ptr(a.u), 4 * a.u.numel(), stream()))        # a.u⁰ .= a.u
    scale_u_(a, 0)
    t0, t1 = _times(a)
    mom_predict_(a, t0, t1)
    mom_project_(a, b, 1, t1)
    mom_correct_(a, t1)
    mom_project_(a, b, 0.5, t1)
    a.dt.append(CFL(a))
