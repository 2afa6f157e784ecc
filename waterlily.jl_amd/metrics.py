"""MeanFlow — temporal averages of pressure, velocity and u⊗u on device, mirror of /root/reference/src/Metrics.jl:200-255,
and a plain checkpoint of (u, p, Δt) (the reference's JLD2 extension, ext/WaterLilyJLD2Ext.jl, stores the same three)."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .core import jl_zeros, ptr, sgrid, stream, to_device, to_host


class MeanFlow:
    """MeanFlow(flow; t_init=time(flow), uu_stats=false)   src/Metrics.jl:205-226"""

    def __init__(self, flow, t_init=None, uu_stats=False):
        D = flow.D
        self.D = D
        self.P = jl_zeros(tuple(flow.p.shape))
        self.U = jl_zeros(tuple(flow.u.shape))
        self.UU = jl_zeros(tuple(flow.p.shape) + (D, D)) if uu_stats else None
        self.t = [np.float32(flow.time() if t_init is None else t_init)]
        self.uu_stats = bool(uu_stats)

    def time(self):
        """time(meanflow) = t[end] - t[1]   :228"""
        return np.float32(self.t[-1] - self.t[0])

    def reset_(self, t_init=0.0):
        """reset!(meanflow; t_init)   :230-235"""
        for a in (self.P, self.U) + ((self.UU,) if self.UU is not None else ()):
            a.zero_()
        self.t = [np.float32(t_init)]

    def update_(self, flow):
        """update!(meanflow, flow)   :236-248"""
        dt = np.float32(flow.time() - self.t[-1])
        eps = np.float32(dt / np.float32(dt + self.time() + np.finfo(np.float32).eps))
        if len(self.t) == 1:
            eps = np.float32(1)          # the first update takes the instantaneous field
        g = sgrid(flow.p)
        check(lib().wl_meanflow_update(ptr(self.P), ptr(self.U), ptr(self.UU) if self.UU is not None else None, ptr(flow.p), ptr(flow.u),
                                       C.byref(g), float(eps), stream()))
        self.t.append(np.float32(self.t[-1] + dt))

    def uu(self):
        """uu(a): Reynolds stresses τ = UU - U⊗U   :250-258"""
        assert self.UU is not None
        tau = jl_zeros(tuple(self.UU.shape))
        g = sgrid(self.P)
        check(lib().wl_meanflow_uu(ptr(tau), ptr(self.UU), ptr(self.U), C.byref(g), stream()))
        return tau


def save_checkpoint(path, flow):
    """u, p and the Δt history of a Flow (what ext/WaterLilyJLD2Ext.jl's save! writes), as an .npz"""
    np.savez(path, u=to_host(flow.u), p=to_host(flow.p), dt=np.asarray(flow.dt, dtype=np.float32))


def load_checkpoint(path, flow):
    """load!(flow): restore u, p, Δt (allow_pickle stays off)"""
    with np.load(path) as z:
        assert tuple(z["u"].shape) == tuple(flow.u.shape) and tuple(z["p"].shape) == tuple(flow.p.shape)
        flow.u.copy_(to_device(np.asfortranarray(z["u"])))
        flow.p.copy_(to_device(np.asfortranarray(z["p"])))
        flow.dt[:] = [np.float32(v) for v in z["dt"]]
