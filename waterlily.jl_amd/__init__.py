"""waterlily.jl_amd — MI355X (gfx950) backend for WaterLily's time-step hot path.

Host-side mirror (Python) of the reference's Flow / AbstractPoisson / Simulation surface for that path, over
hand-written HIP kernels behind a flat C ABI (include/wlhip.h, libwlhip.so).  No CPU fallback exists.
"""
from ._lib import LIB_PATH, SIGNATURES, WlError, lib  # noqa: F401
from .core import (BC_, CDS, QUICK, VANLEER, L2, exitBC_, inside, jl_zeros, loc, perBC_, to_device, to_host)  # noqa: F401
from .flow import BDIM_, CFL, Flow, conv_diff_, mom_correct_, mom_predict_, mom_project_, mom_step_, scale_u_  # noqa: F401
from .poisson import (pcg_, poisson_solver_, GaussSeidelRB_, Jacobi_, L1, Linf, MultiLevelPoisson, Poisson, increment_, mult_, norms, prolongate_,  # noqa: F401
                      residual_, restrict_, restrictL_, set_diag_, smooth_, update_)
from .metrics import MeanFlow, load_checkpoint, save_checkpoint  # noqa: F401
from .simulation import (FusedSimulation, Simulation, measure_, pressure_force, pressure_moment, viscous_force,  # noqa: F401
                         viscous_moment)
