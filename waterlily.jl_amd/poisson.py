"""Poisson / MultiLevelPoisson — host-side mirror of /root/reference/src/Poisson.jl and
src/MultiLevelPoisson.jl over the HIP C ABI."""
import ctypes as C

import numpy as np

from . import core
from ._lib import WlError, check, lib, wl_grid
from .core import jl_zeros, perBC_, perdir_mask, ptr, sgrid, stream


class Poisson:
    """Poisson(x,L,z;perdir)   src/Poisson.jl:22-39 — x,L,z alias the caller's arrays."""

    def __init__(self, x, L, z, perdir=()):
        assert tuple(x.shape) == tuple(z.shape) and tuple(L.shape) == tuple(x.shape) + (x.dim(),)   # :33
        self.x, self.L, self.z = x, L, z
        self.perdir = tuple(perdir)
        self.r, self.eps, self.D, self.iD = (jl_zeros(tuple(x.shape)) for _ in range(4))
        self.n = []
        self.g = sgrid(x)
        set_diag_(self.D, self.iD, self.L)

    def n_inside(self):
        return int(np.prod([s - 2 for s in self.x.shape]))


def set_diag_(D, iD, L):
    """set_diag!   src/Poisson.jl:43-46"""
    g = sgrid(D)
    check(lib().wl_set_diag(ptr(D), ptr(iD), ptr(L), C.byref(g), stream()))


def update_(p):
    """update!(p::Poisson)   :47"""
    set_diag_(p.D, p.iD, p.L)


def mult_(p, x):
    """mult!(p,x): p.z = A x   :63-69"""
    perBC_(x, p.perdir)
    check(lib().wl_mult(ptr(p.z), ptr(p.L), ptr(p.D), ptr(x), C.byref(p.g), stream()))
    return p.z


def residual_(p):
    """residual!   :92-98"""
    perBC_(p.x, p.perdir)
    check(lib().wl_residual(ptr(p.r), ptr(p.x), ptr(p.z), ptr(p.L), ptr(p.D), ptr(p.iD), C.byref(p.g), None, stream()))


def increment_(p, w=1.0):
    """increment!(p;ω)   :100-104"""
    perBC_(p.eps, p.perdir)
    check(lib().wl_increment(ptr(p.r), ptr(p.x), ptr(p.eps), ptr(p.L), ptr(p.D), C.byref(p.g), float(w), stream()))


def Jacobi_(p, it=1, w=1.0):
    """Jacobi!(p;it,ω)   :111-114"""
    check(lib().wl_jacobi(ptr(p.eps), ptr(p.r), ptr(p.x), ptr(p.L), ptr(p.D), ptr(p.iD), C.byref(p.g), int(it), float(w), perdir_mask(p.perdir), stream()))


def GaussSeidelRB_(p, it=4, w=1.0):
    """GaussSeidelRB!(p;it,ω)   :141-148"""
    check(lib().wl_gsrb(ptr(p.eps), ptr(p.r), ptr(p.x), ptr(p.L), ptr(p.D), ptr(p.iD), C.byref(p.g), int(it), float(w), perdir_mask(p.perdir), stream()))


smooth_ = GaussSeidelRB_   # src/MultiLevelPoisson.jl:106


def pcg_(p, it=6):
    """pcg!(p;it)   src/Poisson.jl:166-186"""
    check(lib().wl_pcg(ptr(p.eps), ptr(p.r), ptr(p.x), ptr(p.z), ptr(p.L), ptr(p.D), ptr(p.iD), C.byref(p.g), int(it), perdir_mask(p.perdir), stream()))


def poisson_solver_(p, tol=2e-3, itmx=1e3):
    """solver!(p::Poisson;tol,itmx)   src/Poisson.jl:212-223 — residual!, pcg! until L₁ < tol/10·N and L∞ < tol"""
    n, r1, rinf = C.c_int(), C.c_double(), C.c_float()
    check(lib().wl_poisson_solve(ptr(p.eps), ptr(p.r), ptr(p.x), ptr(p.z), ptr(p.L), ptr(p.D), ptr(p.iD), C.byref(p.g), float(tol), int(itmx),
                                 perdir_mask(p.perdir), C.byref(n), C.byref(r1), C.byref(rinf), stream()))
    p.n.append(n.value)
    return n.value


def norms(p):
    """(L₁(p), L∞(p))   :190-191 — one fused pass"""
    l1, linf = C.c_double(), C.c_float()
    check(lib().wl_norms(ptr(p.r), C.byref(p.g), C.byref(l1), C.byref(linf), None, stream()))
    return np.float32(l1.value), np.float32(linf.value)


def L1(p):
    return norms(p)[0]


def Linf(p):
    return norms(p)[1]


def restrict_(a, b):
    """restrict!(a,b,c): a coarse, b fine   src/MultiLevelPoisson.jl:49"""
    ga, gb = sgrid(a), sgrid(b)
    check(lib().wl_restrict(ptr(a), C.byref(ga), ptr(b), C.byref(gb), stream()))


def prolongate_(a, b):
    """prolongate!(a,b,c): a fine, b coarse   :50"""
    ga, gb = sgrid(a), sgrid(b)
    check(lib().wl_prolongate(ptr(a), C.byref(ga), ptr(b), C.byref(gb), stream()))


def restrictL_(a, b, perdir=()):
    """restrictL!(a,b,c;perdir)   :42-48"""
    ga, gb = core.vgrid(a), core.vgrid(b)
    check(lib().wl_restrictL(ptr(a), C.byref(ga), ptr(b), C.byref(gb), perdir_mask(perdir), stream()))


class _LevelView:
    """Read access to pois.levels[k].{L,D,iD,x,eps,r,z} (device -> host copy, like `Array(a)`)."""

    def __init__(self, ml, l):
        self._ml, self._l = ml, l
        g = wl_grid()
        check(lib().wl_mg_level_grid(ml._h, l, C.byref(g)))
        self.grid = g
        self.dims = (g.nx, g.ny) if g.D == 2 else (g.nx, g.ny, g.nz)

    def _get(self, name):
        D = len(self.dims)
        shape = self.dims + (D,) if name == "L" else self.dims
        out = np.empty(shape, dtype=np.float32, order="F")
        p = lib().wl_mg_level_field(self._ml._h, self._l, name.encode())
        check(lib().wl_d2h(out.ctypes.data_as(C.c_void_p), p, out.nbytes, stream()))
        return out

    L = property(lambda s: s._get("L"))
    D = property(lambda s: s._get("D"))
    iD = property(lambda s: s._get("iD"))
    x = property(lambda s: s._get("x"))
    eps = property(lambda s: s._get("eps"))
    r = property(lambda s: s._get("r"))
    z = property(lambda s: s._get("z"))


class MultiLevelPoisson:
    """MultiLevelPoisson(x,L,z;maxlevels,perdir)   src/MultiLevelPoisson.jl:61-77 — wraps the `wl_mg` handle."""

    def __init__(self, x, L, z, maxlevels=10, perdir=()):
        assert tuple(x.shape) == tuple(z.shape) and tuple(L.shape) == tuple(x.shape) + (x.dim(),)
        self.x, self.L, self.z = x, L, z
        self.perdir = tuple(perdir)
        h = C.c_void_p()
        g = sgrid(x)
        rc = lib().wl_mg_create(C.byref(h), ptr(x), ptr(L), ptr(z), C.byref(g), perdir_mask(perdir), int(maxlevels))
        if rc == -3:
            raise AssertionError("MultiLevelPoisson requires size=a2ⁿ, where n>2")   # :73-74
        check(rc)
        self._h = h
        self.levels = [_LevelView(self, l) for l in range(lib().wl_mg_nlevels(h))]

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().wl_mg_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def n(self):
        out = (C.c_int16 * 65536)()
        k = lib().wl_mg_history(self._h, out, 65536)
        return [int(v) for v in out[:k]]

    def update_(self):
        """update!(ml)   :79-86"""
        check(lib().wl_mg_update(self._h, stream()))

    def Vcycle_(self, l=0, w=1.0):
        """Vcycle!(ml;l,ω)   :88-101 (l is 0-based here)"""
        check(lib().wl_mg_vcycle(self._h, int(l), float(w), stream()))

    def smooth_(self, l=0, it=4, w=1.0):
        """smooth!(levels[l];ω) = GaussSeidelRB!   src/MultiLevelPoisson.jl:106"""
        check(lib().wl_mg_smooth(self._h, int(l), int(it), float(w), stream()))

    def set_fused(self, on, pair=True, tail=None, tail_lds=True):
        """on: temporally blocked smoother; pair: its two-cells-per-thread variant on constant-coefficient levels;
        tail: the smallest levels of the V-cycle in one launch (default: same as `on`); tail_lds: that launch keeps its levels in LDS"""
        tail = bool(on) if tail is None else bool(tail)
        check(lib().wl_mg_set_fused(self._h, int(bool(on)) | (0 if pair else 4) | (0 if tail else 8) | (0 if tail_lds else 32)))

    def level_is_const(self, l):
        return bool(lib().wl_mg_level_is_const(self._h, l))

    def solver_(self, tol=2e-3, itmx=32):
        """solver!(ml;tol,itmx)   :108-128"""
        n, r1, rinf = C.c_int(), C.c_double(), C.c_float()
        check(lib().wl_mg_solve(self._h, float(tol), int(itmx), C.byref(n), C.byref(r1), C.byref(rinf), stream()))
        return n.value

    def log(self):
        cap = 80
        a, b, c = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
        k = lib().wl_mg_last_log(self._h, a, b, c, cap)
        return np.array(a[:k]), np.array(b[:k]), np.array(c[:k])

    def level_norms(self, l=0):
        lv = self.levels[l]
        l1, linf = C.c_double(), C.c_float()
        p = lib().wl_mg_level_field(self._h, l, b"r")
        check(lib().wl_norms(p, C.byref(lv.grid), C.byref(l1), C.byref(linf), None, stream()))
        return np.float32(l1.value), np.float32(linf.value)


def update_ml_(ml):
    ml.update_()


def solver_(b, **kw):
    """solver!(b) — dispatches on the type like the reference (Poisson: pcg!, MultiLevelPoisson: V-cycles)"""
    return poisson_solver_(b, **kw) if isinstance(b, Poisson) else b.solver_(**kw)
