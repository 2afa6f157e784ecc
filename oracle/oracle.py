"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/libwloracle*.so.

CPU restatement of the WaterLily.jl hot path (see oracle/wl_oracle.hpp for the file:line map).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (waterlily.jl_amd/) never does.

Arrays are numpy arrays in Fortran (column-major) order with the Julia shapes
(Ng...,), (Ng...,D), (Ng...,D,D); indices in this API are Julia 1-based where they appear.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BC_FN = C.CFUNCTYPE(C.c_double, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_void_p)
IC_FN = C.CFUNCTYPE(C.c_double, C.c_int, C.POINTER(C.c_double), C.c_void_p)
QUICK, VANLEER, CDS = 0, 1, 2


def build(force=False):
    """Compile the restatement (gcc).  Building the checker is not using it."""
    need = force or not all(os.path.exists(os.path.join(_HERE, n)) for n in ("libwloracle.so", "libwloracle_omp.so"))
    if not need:
        src_m = max(os.path.getmtime(os.path.join(_HERE, n)) for n in ("wl_oracle.hpp", "wl_oracle_capi.cpp"))
        need = any(os.path.getmtime(os.path.join(_HERE, n)) < src_m for n in ("libwloracle.so", "libwloracle_omp.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-j2"], stdout=subprocess.DEVNULL)


_libs = {}


def lib(omp=False):
    key = bool(omp)
    if key in _libs:
        return _libs[key]
    path = os.path.join(_HERE, "libwloracle_omp.so" if omp else "libwloracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    dbl = C.c_double
    for name, res, args in [
        ("wlo_last_error", C.c_char_p, []),
        ("wlo_max_threads", C.c_int, []),
        ("wlo_set_threads", None, [C.c_int]),
        ("wlo_builtin_fn", C.c_void_p, [C.c_int]),
        ("wlo_quick", dbl, [C.c_int, dbl, dbl, dbl]),
        ("wlo_vanLeer", dbl, [C.c_int, dbl, dbl, dbl]),
        ("wlo_cds", dbl, [C.c_int, dbl, dbl, dbl]),
        ("wlo_median", dbl, [dbl, dbl, dbl]),
        ("wlo_mu0", dbl, [C.c_int, dbl, dbl]),
        ("wlo_mu1", dbl, [C.c_int, dbl, dbl]),
        ("wlo_kern", dbl, [C.c_int, dbl]),
        ("wlo_flux1d", dbl, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, dbl, C.c_int]),
        ("wlo_loc", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
        ("wlo_divisible", C.c_int, [C.c_int]),
        ("wlo_down", None, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
        ("wlo_up", None, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        ("wlo_BC", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, BC_FN, C.c_void_p, C.c_int, C.c_uint, dbl]),
        ("wlo_perBC", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint]),
        ("wlo_exitBC", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, dbl]),
        ("wlo_conv_diff", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, dbl, C.c_uint]),
        ("wlo_BDIM", None, [C.c_int, C.c_int] + [C.c_void_p] * 7 + [dbl]),
        ("wlo_scale_u", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, dbl]),
        ("wlo_div", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
        ("wlo_project", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        ("wlo_CFL", dbl, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, dbl]),
        ("wlo_restrict", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        ("wlo_prolongate", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        ("wlo_restrictL", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]),
        ("wlo_L2_inside", dbl, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
        ("wlo_pressure_force", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, dbl, C.c_void_p]),
        ("wlo_pois_create", C.c_void_p, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int]),
        ("wlo_pois_destroy", None, [C.c_void_p]),
        ("wlo_pois_nlevels", C.c_int, [C.c_void_p]),
        ("wlo_pois_level_dims", None, [C.c_void_p, C.c_int, C.c_void_p]),
        ("wlo_pois_level_field", C.c_void_p, [C.c_void_p, C.c_int, C.c_char_p]),
        ("wlo_pois_solve", C.c_int, [C.c_void_p, dbl, C.c_int]),
        ("wlo_pois_log", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
        ("wlo_pois_update", None, [C.c_void_p]),
        ("wlo_pois_op", None, [C.c_void_p, C.c_int, C.c_int, C.c_int, dbl, C.c_void_p]),
        ("wlo_pois_vcycle", None, [C.c_void_p, C.c_int, dbl]),
        ("wlo_pois_norm", dbl, [C.c_void_p, C.c_int, C.c_int]),
        ("wlo_pois_nhist", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
        ("wlo_sim_create", C.c_void_p, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, BC_FN, BC_FN, dbl, dbl, dbl, dbl, dbl, C.c_uint, C.c_int,
                                        C.c_int, IC_FN, C.c_int, C.c_void_p, dbl, BC_FN, C.c_void_p]),
        ("wlo_sim_destroy", None, [C.c_void_p]),
        ("wlo_sim_field", C.c_void_p, [C.c_void_p, C.c_char_p]),
        ("wlo_sim_step", None, [C.c_void_p, C.c_int]),
        ("wlo_sim_step_until", C.c_int, [C.c_void_p, dbl, C.c_int, C.c_int]),
        ("wlo_sim_measure", None, [C.c_void_p]),
        ("wlo_sim_set_body", None, [C.c_void_p, C.c_int, C.c_void_p, dbl, C.c_void_p, C.c_void_p]),
        ("wlo_body_measure", None, [C.c_int, C.c_int, C.c_int, C.c_void_p, dbl, C.c_void_p, C.c_void_p, C.c_void_p, dbl, C.c_void_p]),
        ("wlo_pressure_force_body", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, dbl, C.c_void_p, C.c_void_p]),
        ("wlo_viscous_force_body", None, [C.c_int, C.c_int, C.c_void_p, dbl, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, dbl, C.c_void_p, C.c_void_p]),
        ("wlo_pressure_moment_body", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, dbl, C.c_void_p, C.c_void_p]),
        ("wlo_viscous_moment_body", None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, dbl, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, dbl, C.c_void_p, C.c_void_p]),
        ("wlo_sim_pressure_moment", None, [C.c_void_p, C.c_void_p, C.c_void_p]),
        ("wlo_sim_viscous_moment", None, [C.c_void_p, C.c_void_p, C.c_void_p]),
        ("wlo_sim_time", dbl, [C.c_void_p]),
        ("wlo_sim_flow_time", dbl, [C.c_void_p]),
        ("wlo_sim_dt", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
        ("wlo_sim_nhist", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
        ("wlo_sim_nlevels", C.c_int, [C.c_void_p]),
        ("wlo_sim_level_dims", None, [C.c_void_p, C.c_int, C.c_void_p]),
        ("wlo_sim_level_field", C.c_void_p, [C.c_void_p, C.c_int, C.c_char_p]),
        ("wlo_sim_pressure_force", None, [C.c_void_p, C.c_void_p]),
        ("wlo_sim_viscous_force", None, [C.c_void_p, C.c_void_p]),
        ("wlo_viscous_force", None, [C.c_int, C.c_int, C.c_void_p, dbl, C.c_void_p, C.c_void_p, C.c_void_p, dbl, C.c_void_p]),
        ("wlo_sim_pois_norm", dbl, [C.c_void_p, C.c_int]),
        ("wlo_sim_phase", None, [C.c_void_p, C.c_int]),
    ]:
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _libs[key] = L
    return L


def _dt(a):
    if a.dtype == np.float32:
        return 0
    if a.dtype == np.float64:
        return 1
    raise TypeError("oracle supports float32/float64 only")


def _tag(T):
    return 0 if np.dtype(T) == np.float32 else 1


def _ptr(a):
    assert a.flags.f_contiguous, "oracle arrays must be Fortran (column-major) contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _ints(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def _dbls(v):
    return (C.c_double * len(v))(*[float(x) for x in v])


def perdir_mask(perdir):
    m = 0
    for j in perdir:
        m |= 1 << (int(j) - 1)
    return m


def _wrap_bc(fn):
    """Python uBC(i,x,t) -> C callback (keeps a reference alive on the returned object)."""
    if fn is None:
        return C.cast(None, BC_FN)
    return BC_FN(lambda i, x, t, user, _fn=fn: float(_fn(i, x, t)))


class _XView:
    """SVector-like view of the C double* handed to callbacks (0-based indexing)."""
    __slots__ = ("p", "n")

    def __init__(self, p, n):
        self.p, self.n = p, n

    def __getitem__(self, k):
        return self.p[k]

    def __len__(self):
        return self.n


# ---------------------------------------------------------------- scalar helpers
def quick(u, c, d, T=np.float64):
    return lib().wlo_quick(_tag(T), u, c, d)


def vanLeer(u, c, d, T=np.float64):
    return lib().wlo_vanLeer(_tag(T), u, c, d)


def cds(u, c, d, T=np.float64):
    return lib().wlo_cds(_tag(T), u, c, d)


def mu0(d, e, T=np.float64):
    return lib().wlo_mu0(_tag(T), d, e)


def mu1(d, e, T=np.float64):
    return lib().wlo_mu1(_tag(T), d, e)


def kern(d, T=np.float64):
    return lib().wlo_kern(_tag(T), d)


def flux1d(which, f, I, u, scheme=QUICK, Ip=0):
    """which ∈ {'ϕu','ϕuL','ϕuR','ϕuP','ϕ'} on a 1-D float64 array f, 1-based I."""
    w = {"ϕu": 0, "ϕuL": 1, "ϕuR": 2, "ϕuP": 3, "ϕ": 4}[which]
    f = np.asarray(f, dtype=np.float64)
    return lib().wlo_flux1d(w, f.ctypes.data_as(C.c_void_p), len(f), I, Ip, u, scheme)


def loc(i, I):
    x = (C.c_double * len(I))()
    lib().wlo_loc(len(I), i, _ints(I), x)
    return np.array(list(x))


def divisible(n):
    return bool(lib().wlo_divisible(n))


def coarsen_mask(N):
    return tuple(divisible(n) for n in N)


def down(I, c):
    out = (C.c_int * len(I))()
    lib().wlo_down(len(I), _ints(I), _ints([1 if x else 0 for x in c]), out)
    return tuple(out)


def up(I, c):
    lo = (C.c_int * len(I))()
    hi = (C.c_int * len(I))()
    lib().wlo_up(len(I), _ints(I), _ints([1 if x else 0 for x in c]), lo, hi)
    import itertools
    rng = [range(lo[d], hi[d] + 1) for d in range(len(I))]
    return [tuple(reversed(t)) for t in itertools.product(*reversed(rng))]


# ---------------------------------------------------------------- leaf array ops
def BC(a, U, saveexit=False, perdir=(), t=0.0, omp=False):
    """BC!(a,U,saveexit,perdir,t); U is a tuple or a callable uBC(i,x,t)."""
    D = a.ndim - 1
    if callable(U):
        cb = BC_FN(lambda i, x, tt, user: float(U(i, _XView(x, D), tt)))
        lib(omp).wlo_BC(_dt(a), D, _ptr(a), _ints(a.shape[:D]), None, cb, None, int(saveexit), perdir_mask(perdir), float(t))
    else:
        lib(omp).wlo_BC(_dt(a), D, _ptr(a), _ints(a.shape[:D]), _dbls(U), C.cast(None, BC_FN), None, int(saveexit), perdir_mask(perdir), float(t))


def perBC(a, perdir, omp=False):
    lib(omp).wlo_perBC(_dt(a), a.ndim, _ptr(a), _ints(a.shape), perdir_mask(perdir))


def exitBC(u, u0, dt, omp=False):
    D = u.ndim - 1
    lib(omp).wlo_exitBC(_dt(u), D, _ptr(u), _ptr(u0), _ints(u.shape[:D]), float(dt))


def conv_diff(r, u, Phi, nu=0.1, perdir=(), scheme=QUICK, omp=False):
    D = u.ndim - 1
    lib(omp).wlo_conv_diff(_dt(u), D, _ptr(r), _ptr(u), _ptr(Phi), _ints(u.shape[:D]), scheme, float(nu), perdir_mask(perdir))


def BDIM(u, u0, f, V, mu0_, mu1_, dt, omp=False):
    D = u.ndim - 1
    lib(omp).wlo_BDIM(_dt(u), D, _ptr(u), _ptr(u0), _ptr(f), _ptr(V), _ptr(mu0_), _ptr(mu1_), _ints(u.shape[:D]), float(dt))


def scale_u(u, s, omp=False):
    D = u.ndim - 1
    lib(omp).wlo_scale_u(_dt(u), D, _ptr(u), _ints(u.shape[:D]), float(s))


def div(z, u, omp=False):
    lib(omp).wlo_div(_dt(u), z.ndim, _ptr(z), _ptr(u), _ints(z.shape))


def project(u, L, x, omp=False):
    lib(omp).wlo_project(_dt(u), x.ndim, _ptr(u), _ptr(L), _ptr(x), _ints(x.shape))


def CFL(u, sigma, nu, omp=False):
    return lib(omp).wlo_CFL(_dt(u), sigma.ndim, _ptr(u), _ptr(sigma), _ints(sigma.shape), float(nu))


def restrict(a, b, omp=False):
    """restrict!(a,b,c): a coarse, b fine; c recovered from the sizes."""
    lib(omp).wlo_restrict(_dt(a), a.ndim, _ptr(a), _ints(a.shape), _ptr(b), _ints(b.shape))


def prolongate(a, b, omp=False):
    """prolongate!(a,b,c): a fine, b coarse."""
    lib(omp).wlo_prolongate(_dt(a), a.ndim, _ptr(a), _ints(a.shape), _ptr(b), _ints(b.shape))


def restrictL(a, b, perdir=(), omp=False):
    D = a.ndim - 1
    lib(omp).wlo_restrictL(_dt(a), D, _ptr(a), _ints(a.shape[:D]), _ptr(b), _ints(b.shape[:D]), perdir_mask(perdir))


def L2(a):
    """L₂(a) = Σ_inside a² (src/Poisson.jl:188)."""
    a = np.asfortranarray(a)
    return lib().wlo_L2_inside(_dt(a), a.ndim, _ptr(a), _ints(a.shape))


def pressure_force(p, df, center, R):
    out = (C.c_double * p.ndim)()
    lib().wlo_pressure_force(_dt(p), p.ndim, _ptr(p), _ptr(df), _ints(p.shape), _dbls(center), float(R), out)
    return np.array(list(out))


def viscous_force(u, nu, df, center, R):
    """viscous_force(u,ν,df,body) for a sphere/circle   src/Metrics.jl:148-154"""
    D = u.ndim - 1
    out = (C.c_double * D)()
    lib().wlo_viscous_force(_dt(u), D, _ptr(u), float(nu), _ptr(df), _ints(u.shape[:D]), _dbls(center), float(R), out)
    return np.array(list(out))


def _body(body, D):
    """("sphere", c, R) | ("cylinder", c, R, axis) — axis (0-based) is the direction the cylinder extends along | ("plane", point, normal)
    [+ optional trailing translation velocity] -> (kind, c, R, m, vel) for the C API"""
    name = body[0]
    if name == "sphere":
        kind, c, R, m, rest = 1, body[1], float(body[2]), [1.0] * D, body[3:]
    elif name == "cylinder":
        kind, c, R, m, rest = 1, body[1], float(body[2]), [0.0 if k == int(body[3]) else 1.0 for k in range(D)], body[4:]
    elif name == "plane":
        kind, c, R, m, rest = 2, body[1], 0.0, [float(v) for v in body[2]], body[3:]
    else:
        raise ValueError(name)
    vel = [float(v) for v in rest[0]] if rest else [0.0] * D
    return kind, _dbls(c), R, _dbls(m), _dbls(vel)


def body_measure(body, x, fastd2=float("inf"), T=np.float64):
    """measure(body,x,t;fastd²) -> (d, n, V)   src/AutoBody.jl:29-37"""
    D = len(x)
    kind, c, R, m, vel = _body(body, D)
    out = (C.c_double * (1 + 2 * D))()
    lib().wlo_body_measure(_tag(T), D, kind, c, R, m, vel, _dbls(x), float(fastd2), out)
    return out[0], np.array(out[1:1 + D]), np.array(out[1 + D:1 + 2 * D])


def pressure_force_body(p, df, body):
    kind, c, R, m, _ = _body(body, p.ndim)
    out = (C.c_double * p.ndim)()
    lib().wlo_pressure_force_body(_dt(p), p.ndim, _ptr(p), _ptr(df), _ints(p.shape), kind, c, R, m, out)
    return np.array(list(out))


def viscous_force_body(u, nu, df, body):
    D = u.ndim - 1
    kind, c, R, m, _ = _body(body, D)
    out = (C.c_double * D)()
    lib().wlo_viscous_force_body(_dt(u), D, _ptr(u), float(nu), _ptr(df), _ints(u.shape[:D]), kind, c, R, m, out)
    return np.array(list(out))


def pressure_moment_body(x0, p, df, body):
    """pressure_moment(x₀,p,df,body)   src/Metrics.jl:169-174"""
    kind, c, R, m, _ = _body(body, p.ndim)
    out = (C.c_double * p.ndim)()
    lib().wlo_pressure_moment_body(_dt(p), p.ndim, _dbls(x0), _ptr(p), _ptr(df), _ints(p.shape), kind, c, R, m, out)
    return np.array(list(out))


def viscous_moment_body(x0, u, nu, df, body):
    """viscous_moment(x₀,u,ν,df,body)   src/Metrics.jl:183-188"""
    D = u.ndim - 1
    kind, c, R, m, _ = _body(body, D)
    out = (C.c_double * D)()
    lib().wlo_viscous_moment_body(_dt(u), D, _dbls(x0), _ptr(u), float(nu), _ptr(df), _ints(u.shape[:D]), kind, c, R, m, out)
    return np.array(list(out))


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape, order="F")


class Poisson:
    """Poisson(x,L,z;perdir) / MultiLevelPoisson(x,L,z;perdir): x, L, z are ALIASED (as in the reference)."""

    def __init__(self, x, L, z, perdir=(), multilevel=False, omp=False):
        self._lib = lib(omp)
        self.x, self.L, self.z = x, L, z
        self.D = x.ndim
        self.dtype = x.dtype
        self.h = self._lib.wlo_pois_create(_dt(x), self.D, _ptr(x), _ptr(L), _ptr(z), _ints(x.shape), perdir_mask(perdir), int(multilevel))
        if not self.h:
            raise AssertionError(self._lib.wlo_last_error().decode())
        self.multilevel = multilevel

    def __del__(self):
        if getattr(self, "h", None):
            self._lib.wlo_pois_destroy(self.h)
            self.h = None

    @property
    def nlevels(self):
        return self._lib.wlo_pois_nlevels(self.h)

    def level_dims(self, l):
        d = (C.c_int * self.D)()
        self._lib.wlo_pois_level_dims(self.h, l, d)
        return tuple(d)

    def field(self, name, l=0):
        dims = self.level_dims(l)
        shape = dims + (self.D,) if name == "L" else dims
        return _view(self._lib.wlo_pois_level_field(self.h, l, name.encode()), shape, self.dtype)

    def solve(self, tol=2e-3, itmx=-1):
        return self._lib.wlo_pois_solve(self.h, tol, itmx)

    def log(self):
        cap = 80
        a, b, c = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
        n = self._lib.wlo_pois_log(self.h, a, b, c, cap)
        return np.array(a[:n]), np.array(b[:n]), np.array(c[:n])

    def update(self):
        self._lib.wlo_pois_update(self.h)

    def residual(self, l=0):
        self._lib.wlo_pois_op(self.h, l, 0, 0, 1.0, None)

    def Jacobi(self, l=0, it=1, w=1.0):
        self._lib.wlo_pois_op(self.h, l, 1, it, w, None)

    def GaussSeidelRB(self, l=0, it=4, w=1.0):
        self._lib.wlo_pois_op(self.h, l, 2, it, w, None)

    def increment(self, l=0, w=1.0):
        self._lib.wlo_pois_op(self.h, l, 3, 0, w, None)

    def pcg(self, l=0, it=6):
        self._lib.wlo_pois_op(self.h, l, 4, it, 1.0, None)

    def mult(self, x, l=0):
        """mult!(p,x): fills level-l z with A x and returns it."""
        self._lib.wlo_pois_op(self.h, l, 5, 0, 1.0, _ptr(x))
        return self.field("z", l)

    def set_diag(self, l=0):
        self._lib.wlo_pois_op(self.h, l, 6, 0, 1.0, None)

    def Vcycle(self, l=0, w=1.0):
        self._lib.wlo_pois_vcycle(self.h, l, w)

    def L1(self, l=0):
        return self._lib.wlo_pois_norm(self.h, l, 0)

    def Linf(self, l=0):
        return self._lib.wlo_pois_norm(self.h, l, 1)

    def L2(self, l=0):
        return self._lib.wlo_pois_norm(self.h, l, 2)

    @property
    def n(self):
        out = (C.c_int * 4096)()
        k = self._lib.wlo_pois_nhist(self.h, out, 4096)
        return list(out[:k])


def MultiLevelPoisson(x, L, z, perdir=(), omp=False):
    return Poisson(x, L, z, perdir=perdir, multilevel=True, omp=omp)


class Simulation:
    """Simulation(dims,uBC,L;U,Δt,ν,ϵ,g,u0,perdir,exitBC,λ,body,T) — src/WaterLily.jl:93-106.

    uBC: tuple or callable (i,x,t) [then duBC_dt(i,x,t) must supply the time derivative AD gives the reference];
    body: None | ("sphere", centre, radius);  u0: None | callable (i,x).
    """

    def __init__(self, dims, uBC, L, U=None, dt=0.25, nu=0.0, eps=1.0, g=None, u0=None, perdir=(), exitBC=False,
                 scheme=QUICK, body=None, T=np.float32, duBC_dt=None, omp=False):
        self._lib = lib(omp)
        self.D = len(dims)
        self.dims = tuple(int(n) for n in dims)
        self.Ng = tuple(n + 2 for n in self.dims)
        self.T = np.dtype(T)
        D = self.D
        self._keep = []
        null_bc = C.cast(None, BC_FN)
        if isinstance(uBC, str):
            assert uBC == "accel_x" and U is not None  # native functor: uBC(i,x,t) = i==1 ? t : 0
            ufn = C.cast(self._lib.wlo_builtin_fn(0), BC_FN)
            dufn = C.cast(self._lib.wlo_builtin_fn(1), BC_FN)
            Uarr = None
        elif callable(uBC):
            assert U is not None, "`U` (velocity scale) must be specified if boundary conditions `uBC` is a `Function`"
            ufn = BC_FN(lambda i, x, t, user: float(uBC(i, _XView(x, D), t)))
            dufn = BC_FN(lambda i, x, t, user: float(duBC_dt(i, _XView(x, D), t))) if duBC_dt else BC_FN(lambda i, x, t, user: 0.0)
            Uarr = None
        else:
            ufn, dufn = null_bc, null_bc
            Uarr = _dbls(uBC)
            if U is None:
                U = float(np.sqrt(sum(float(v) ** 2 for v in uBC)))
        gfn = BC_FN(lambda i, x, t, user: float(g(i, _XView(x, D), t))) if g else null_bc
        icfn = IC_FN(lambda i, x, user: float(u0(i, _XView(x, D)))) if u0 else C.cast(None, IC_FN)
        self._keep += [ufn, dufn, gfn, icfn]
        kind, c, R = 0, None, 0.0
        if body is not None:
            kind, c, R, bm, bvel = _body(body, D)
        self.U, self.L, self.nu = float(U), float(L), float(nu)
        self.h = self._lib.wlo_sim_create(_tag(T), D, _ints(dims), Uarr, ufn, dufn, float(L), float(U), float(dt), float(nu), float(eps),
                                          perdir_mask(perdir), int(exitBC), scheme, icfn, kind, c, R, gfn, None)
        if not self.h:
            raise AssertionError(self._lib.wlo_last_error().decode())
        if body is not None and (body[0] != "sphere" or len(body) > 3):
            self.set_body(body)      # the constructor's measure! only knows the sphere: redo it with the full description
            self.measure()

    def __del__(self):
        if getattr(self, "h", None):
            self._lib.wlo_sim_destroy(self.h)
            self.h = None

    def field(self, name):
        D = self.D
        shape = {"p": self.Ng, "sigma": self.Ng, "mu1": self.Ng + (D, D)}.get(name, self.Ng + (D,))
        return _view(self._lib.wlo_sim_field(self.h, name.encode()), shape, self.T)

    u = property(lambda s: s.field("u"))
    p = property(lambda s: s.field("p"))

    def step(self, remeasure=True):
        """sim_step!(sim;remeasure)"""
        self._lib.wlo_sim_step(self.h, int(remeasure))

    def step_until(self, t_end, remeasure=True, max_steps=2**31 - 1):
        """sim_step!(sim,t_end;remeasure,max_steps)"""
        return self._lib.wlo_sim_step_until(self.h, float(t_end), int(remeasure), int(max_steps))

    def phase(self, k):
        self._lib.wlo_sim_phase(self.h, k)

    def set_body(self, body):
        """replace the body (e.g. its new position and velocity); follow with measure() or step(remeasure=True)"""
        kind, c, R, m, vel = _body(body, self.D)
        self._lib.wlo_sim_set_body(self.h, kind, c, R, m, vel)

    def measure(self):
        self._lib.wlo_sim_measure(self.h)

    def sim_time(self):
        return self._lib.wlo_sim_time(self.h)

    def time(self):
        return self._lib.wlo_sim_flow_time(self.h)

    @property
    def dt(self):
        out = (C.c_double * 100000)()
        k = self._lib.wlo_sim_dt(self.h, out, 100000)
        return list(out[:k])

    @property
    def pois_n(self):
        out = (C.c_int * 200000)()
        k = self._lib.wlo_sim_nhist(self.h, out, 200000)
        return list(out[:k])

    @property
    def nlevels(self):
        return self._lib.wlo_sim_nlevels(self.h)

    def level_dims(self, l):
        d = (C.c_int * self.D)()
        self._lib.wlo_sim_level_dims(self.h, l, d)
        return tuple(d)

    def level_field(self, name, l=0):
        dims = self.level_dims(l)
        shape = dims + (self.D,) if name == "L" else dims
        return _view(self._lib.wlo_sim_level_field(self.h, l, name.encode()), shape, self.T)

    def pressure_force(self):
        out = (C.c_double * self.D)()
        self._lib.wlo_sim_pressure_force(self.h, out)
        return np.array(list(out))

    def viscous_force(self):
        out = (C.c_double * self.D)()
        self._lib.wlo_sim_viscous_force(self.h, out)
        return np.array(list(out))

    def total_force(self):
        """total_force(sim) = pressure_force + viscous_force   src/Metrics.jl:156-161"""
        return self.pressure_force() + self.viscous_force()

    def pressure_moment(self, x0):
        out = (C.c_double * self.D)()
        self._lib.wlo_sim_pressure_moment(self.h, _dbls(x0), out)
        return np.array(list(out))

    def viscous_moment(self, x0):
        out = (C.c_double * self.D)()
        self._lib.wlo_sim_viscous_moment(self.h, _dbls(x0), out)
        return np.array(list(out))

    def total_moment(self, x0):
        """total_moment(x₀,sim)   src/Metrics.jl:195"""
        return self.pressure_moment(x0) + self.viscous_moment(x0)

    def pois_norm(self, which):
        return self._lib.wlo_sim_pois_norm(self.h, {"L1": 0, "Linf": 1, "L2": 2}[which])
