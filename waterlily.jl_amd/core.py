"""Array/layout layer and boundary conditions — host-side mirror of /root/reference/src/core.jl.

Device arrays are torch tensors on the MI355X laid out exactly like the Julia arrays of the reference
(column-major: x fastest, vector component slowest), so the same device pointer could be handed to the
Julia side unchanged.  PyTorch is plumbing here (device memory + streams); every operation is a
hand-written HIP kernel reached through the C ABI (include/wlhip.h).
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib, wl_grid

QUICK, VANLEER, CDS = 0, 1, 2


def perdir_mask(perdir):
    m = 0
    for j in perdir:
        m |= 1 << (int(j) - 1)
    return m


def device():
    if not torch.cuda.is_available():
        raise RuntimeError("waterlily_jl_amd needs an MI355X (no HIP device visible); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def jl_zeros(shape, fill=0.0):
    """zeros(T, shape) |> mem  (src/Flow.jl:143-144): column-major float32 device array."""
    t = torch.full(tuple(reversed(shape)), float(fill), dtype=torch.float32, device=device())
    return t.permute(*reversed(range(len(shape))))


def to_device(a):
    """`mem(::Array)` constructor: H2D of a numpy array, keeping the Julia (column-major) layout."""
    a = np.asfortranarray(np.asarray(a, dtype=np.float32))
    t = torch.from_numpy(np.ascontiguousarray(a.transpose())).to(device())
    return t.permute(*reversed(range(a.ndim)))


def to_host(t):
    """`Array(a)`: D2H into a Fortran-ordered numpy array."""
    nd = t.dim()
    base = t.permute(*reversed(range(nd)))
    assert base.is_contiguous(), "not a Julia-layout array"
    return np.asfortranarray(base.cpu().numpy().transpose())


def ptr(t):
    if t is None:
        return None
    nd = t.dim()
    assert t.permute(*reversed(range(nd))).is_contiguous(), "array must be dense column-major (Julia layout)"
    assert t.dtype == torch.float32 and t.is_cuda
    return C.c_void_p(t.data_ptr())


def grid_of(dims_with_ghosts):
    D = len(dims_with_ghosts)
    arr = (C.c_int32 * 3)(*(list(dims_with_ghosts) + [1] * (3 - D)))
    return lib().wl_grid_single(D, arr)


def sgrid(a):
    """grid descriptor of a scalar array"""
    return grid_of(tuple(a.shape))


def vgrid(a):
    """grid descriptor of a vector array (Ng...,D)"""
    return grid_of(tuple(a.shape[:-1]))


def inside(a, buff=1):
    """inside(a;buff) as a tuple of slices (0-based)   src/core.jl:47"""
    return tuple(slice(buff, n - buff) for n in a.shape)


def loc(i, I, T=np.float32):
    """loc(i,I) = I - 1.5 - δ(i)/2 with Julia 1-based I   src/core.jl:177"""
    return np.array([T(I[d]) - T(1.5) - T(1 if d == i - 1 else 0) / T(2) for d in range(len(I))], dtype=T)


def tabulate_shell(fn, shape, D, t, perdir=()):
    """uBC(i,loc(i,I),t) on the two outermost layers of every non-periodic direction of a (Ng...,D) array (host side of
    wl_bc_vec_fn: the closure cannot cross the C ABI, its boundary values can)."""
    Ng = tuple(shape[:D])
    tab = np.zeros(Ng + (D,), dtype=np.float32, order="F")
    done = np.zeros(Ng, dtype=bool)
    for j in range(D):
        if (j + 1) in perdir:
            continue
        for layer in (0, 1, Ng[j] - 2, Ng[j] - 1):
            rng = [range(n) for n in Ng]
            rng[j] = (layer,)
            for I in np.ndindex(*[len(r) for r in rng]):
                J = tuple(rng[d][I[d]] for d in range(D))
                if done[J]:
                    continue
                done[J] = True
                for i in range(1, D + 1):
                    tab[J + (i - 1,)] = fn(i, loc(i, tuple(k + 1 for k in J)), t)
    return tab


def BC_(a, U, saveexit=False, perdir=(), t=0):
    """BC!(a,U,saveexit,perdir,t)   src/core.jl:200-219 — U: tuple, or a function uBC(i,x,t) (tabulated on the host)"""
    if callable(U):
        D = a.dim() - 1
        Ub = to_device(tabulate_shell(U, tuple(a.shape), D, t, perdir))
        g = vgrid(a)
        check(lib().wl_bc_vec_fn(ptr(a), ptr(Ub), C.byref(g), int(bool(saveexit)), perdir_mask(perdir), stream()))
        check(lib().wl_stream_sync(stream()))     # Ub is a temporary
        return
    D = a.dim() - 1
    Uc = (C.c_float * 3)(*([float(v) for v in U] + [0.0] * (3 - D)))
    g = vgrid(a)
    check(lib().wl_bc_vec(ptr(a), C.byref(g), Uc, int(bool(saveexit)), perdir_mask(perdir), stream()))


def accelerate_(r, t, g=None, uBC=None, duBC_dt=None):
    """accelerate!(r,t,g,U)   src/Flow.jl:69-73: r[I,i] += g(i,x,t) + ∂ₜuBC(i,x,t), tabulated on the host (ForwardDiff is the
    reference's way to get ∂ₜuBC; here the caller supplies it)"""
    if g is None and not (callable(uBC) and duBC_dt is not None):
        return
    D = r.dim() - 1
    Ng = tuple(r.shape[:D])
    G = np.zeros(Ng + (D,), dtype=np.float32, order="F")
    for I in np.ndindex(*Ng):
        for i in range(1, D + 1):
            x = loc(i, tuple(k + 1 for k in I))
            G[I + (i - 1,)] = (g(i, x, t) if g is not None else 0.0) + (duBC_dt(i, x, t) if (callable(uBC) and duBC_dt is not None) else 0.0)
    Gd = to_device(G)
    gr = vgrid(r)
    check(lib().wl_accelerate_field(ptr(r), ptr(Gd), C.byref(gr), stream()))
    check(lib().wl_stream_sync(stream()))


def perBC_(a, perdir):
    """perBC!(a,perdir)   src/core.jl:239-243"""
    if not perdir:
        return
    g = sgrid(a)
    check(lib().wl_bc_per_scalar(ptr(a), C.byref(g), perdir_mask(perdir), stream()))


def exitBC_(u, u0, dt):
    """exitBC!(u,u⁰,Δt)   src/core.jl:226-233"""
    g = vgrid(u)
    check(lib().wl_exit_bc(ptr(u), ptr(u0), C.byref(g), float(dt), stream()))


def L2(a):
    """L₂(a) = Σ_inside a²  (src/Poisson.jl:188; device method like ext/WaterLilyAMDGPUExt.jl:18)"""
    out = C.c_double()
    g = sgrid(a)
    check(lib().wl_L2_inside(ptr(a), C.byref(g), C.byref(out), stream()))
    return out.value
