"""ctypes binding of libwlhip.so (include/wlhip.h).  There is NO CPU fallback: if the library is
missing or no MI355X is visible, every compute entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WLHIP_LIB") or os.path.join(_HERE, "libwlhip.so")   # WLHIP_LIB: explicit library (kernel A/B experiments)


class wl_grid(C.Structure):
    _fields_ = [("D", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("k0", C.c_int32), ("k1", C.c_int32), ("gk", C.c_int32), ("gnz", C.c_int32)]


class wl_body(C.Structure):
    """include/wlhip.h wl_body: closed-form AutoBody shapes"""
    _fields_ = [("kind", C.c_int32), ("c", C.c_float * 3), ("R", C.c_float), ("m", C.c_float * 3), ("V", C.c_float * 3)]


def make_body(body, D):
    """("sphere", c, R) | ("cylinder", c, R, axis) — axis (0-based) is the direction the cylinder extends along |
    ("plane", point, normal), each optionally followed by the body's translation velocity -> wl_body"""
    name = body[0]
    pad = lambda v: [float(x) for x in v] + [0.0] * (3 - len(v))   # noqa: E731
    b = wl_body()
    if name == "sphere":
        b.kind, b.R, m, rest = 1, float(body[2]), [1.0] * D, body[3:]
    elif name == "cylinder":
        b.kind, b.R, m, rest = 1, float(body[2]), [0.0 if k == int(body[3]) else 1.0 for k in range(D)], body[4:]
    elif name == "plane":
        b.kind, b.R, m, rest = 2, 0.0, list(body[2]), body[3:]
    else:
        raise ValueError(f"unknown body {name!r}")
    b.c = (C.c_float * 3)(*pad(list(body[1])))
    b.m = (C.c_float * 3)(*pad(m))
    b.V = (C.c_float * 3)(*pad(list(rest[0]) if rest else [0.0] * D))
    return b


class wl_sim_desc(C.Structure):
    _fields_ = [("D", C.c_int32), ("dims", C.c_int32 * 3), ("uBC", C.c_float * 3), ("nu", C.c_float), ("dt0", C.c_float),
                ("perdir_mask", C.c_uint32), ("exitBC", C.c_int32), ("scheme", C.c_int32), ("has_body", C.c_int32),
                ("u", C.c_void_p), ("u0", C.c_void_p), ("f", C.c_void_p), ("p", C.c_void_p), ("sigma", C.c_void_p),
                ("V", C.c_void_p), ("mu0", C.c_void_p), ("mu1", C.c_void_p), ("us", C.c_void_p)]


SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
P, G = C.c_void_p, C.POINTER(wl_grid)
f32, f64, i32, u32, sz = C.c_float, C.c_double, C.c_int, C.c_uint, C.c_size_t
# name -> (restype, argtypes); every symbol include/wlhip.h declares
SIGNATURES = {
    "wl_init": (i32, [i32]),
    "wl_last_error_string": (C.c_char_p, []),
    "wl_version": (i32, []),
    "wl_malloc": (i32, [C.POINTER(P), sz]),
    "wl_free": (i32, [P]),
    "wl_h2d": (i32, [P, P, sz, P]),
    "wl_d2h": (i32, [P, P, sz, P]),
    "wl_d2d": (i32, [P, P, sz, P]),
    "wl_stream_sync": (i32, [P]),
    "wl_grid_single": (wl_grid, [i32, C.POINTER(C.c_int32)]),
    "wl_fill": (i32, [P, f32, sz, P]),
    "wl_scale": (i32, [P, f32, sz, P]),
    "wl_div_scalar": (i32, [P, f32, sz, P]),
    "wl_sum": (i32, [P, sz, C.POINTER(f64), P]),
    "wl_sum_abs_max_abs": (i32, [P, sz, C.POINTER(f64), C.POINTER(f32), P]),
    "wl_max": (i32, [P, sz, C.POINTER(f32), P]),
    "wl_dot": (i32, [P, P, sz, C.POINTER(f64), P]),
    "wl_L2_inside": (i32, [P, G, C.POINTER(f64), P]),
    "wl_bc_vec": (i32, [P, G, C.POINTER(f32), i32, u32, P]),
    "wl_bc_per_scalar": (i32, [P, G, u32, P]),
    "wl_bc_vec_fn": (i32, [P, P, G, i32, u32, P]),
    "wl_accelerate_field": (i32, [P, P, G, P]),
    "wl_meanflow_update": (i32, [P, P, P, P, P, G, f32, P]),
    "wl_meanflow_uu": (i32, [P, P, P, G, P]),
    "wl_exit_bc": (i32, [P, P, G, f32, P]),
    "wl_conv_diff": (i32, [P, P, P, G, f32, u32, i32, P]),
    "wl_bdim": (i32, [P, P, P, P, P, P, G, f32, f32, f32, P]),
    "wl_scale_u": (i32, [P, G, f32, P]),
    "wl_div": (i32, [P, P, G, P]),
    "wl_project": (i32, [P, P, P, G, P]),
    "wl_cfl": (i32, [P, P, G, f32, f32, C.POINTER(f32), P]),
    "wl_set_diag": (i32, [P, P, P, G, P]),
    "wl_mult": (i32, [P, P, P, P, G, P]),
    "wl_residual": (i32, [P, P, P, P, P, P, G, P, P]),
    "wl_increment": (i32, [P, P, P, P, P, G, f32, P]),
    "wl_jacobi": (i32, [P, P, P, P, P, P, G, i32, f32, u32, P]),
    "wl_gsrb": (i32, [P, P, P, P, P, P, G, i32, f32, u32, P]),
    "wl_norms": (i32, [P, G, C.POINTER(f64), C.POINTER(f32), P, P]),
    "wl_pcg": (i32, [P, P, P, P, P, P, P, G, i32, u32, P]),
    "wl_poisson_solve": (i32, [P, P, P, P, P, P, P, G, f64, i32, u32, P, P, P, P]),
    "wl_reduce_workspace_bytes": (sz, []),
    "wl_restrict": (i32, [P, G, P, G, P]),
    "wl_prolongate": (i32, [P, G, P, G, P]),
    "wl_restrictL": (i32, [P, G, P, G, u32, P]),
    "wl_coarsen_dims": (i32, [i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "wl_mg_create": (i32, [C.POINTER(P), P, P, P, G, u32, i32]),
    "wl_mg_destroy": (i32, [P]),
    "wl_mg_update": (i32, [P, P]),
    "wl_mg_nlevels": (i32, [P]),
    "wl_mg_level_grid": (i32, [P, i32, G]),
    "wl_mg_level_field": (P, [P, i32, C.c_char_p]),
    "wl_mg_vcycle": (i32, [P, i32, f32, P]),
    "wl_mg_smooth": (i32, [P, i32, i32, f32, P]),
    "wl_mg_set_fused": (i32, [P, i32]),
    "wl_mg_level_is_const": (i32, [P, i32]),
    "wl_mg_smoother_kind": (i32, [P, i32]),
    "wl_mg_solve": (i32, [P, f64, i32, C.POINTER(i32), C.POINTER(f64), C.POINTER(f32), P]),
    "wl_mg_history": (i32, [P, C.POINTER(C.c_int16), i32]),
    "wl_mg_last_log": (i32, [P, C.POINTER(f64), C.POINTER(f64), C.POINTER(f64), i32]),
    "wl_sim_create": (i32, [C.POINTER(P), C.POINTER(wl_sim_desc)]),
    "wl_sim_create_on": (i32, [C.POINTER(P), C.POINTER(wl_sim_desc), P]),
    "wl_sim_destroy": (i32, [P]),
    "wl_sim_field": (P, [P, C.c_char_p]),
    "wl_sim_pois": (P, [P]),
    "wl_sim_grid": (i32, [P, G]),
    "wl_sim_init_flow": (i32, [P, P]),
    "wl_sim_set_option": (i32, [P, C.c_char_p, i32]),
    "wl_sim_update": (i32, [P, P]),
    "wl_sim_set_forcing": (i32, [P, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]),
    "wl_accelerate": (i32, [P, G, C.POINTER(f32), P]),
    "wl_sim_mom_step": (i32, [P, P]),
    "wl_sim_mom_steps": (i32, [P, i32, P]),
    "wl_sim_dt": (i32, [P, C.POINTER(f32), i32]),
    "wl_sim_time": (f64, [P]),
    "wl_sim_dt_last": (f32, [P]),
    "wl_sim_set_dt_last": (i32, [P, f32]),
    "wl_sim_phase": (i32, [P, i32, P]),
    "wl_sim_apply_ic": (i32, [P, i32, P]),
    "wl_sim_measure_sphere": (i32, [P, C.POINTER(f32), f32, f32, P]),
    "wl_comm_rccl_unique_id": (i32, [C.c_char_p]),
    "wl_comm_rccl_create": (i32, [C.POINTER(P), i32, i32, C.c_char_p]),
    "wl_comm_callbacks_create": (i32, [C.POINTER(P), i32, i32, P, P, P]),
    "wl_comm_rccl_available": (i32, []),
    "wl_comm_rccl_add_async": (i32, [P, C.c_char_p]),
    "wl_comm_set_loopback": (i32, [P, i32]),
    "wl_comm_set_virtual": (i32, [P, i32, i32]),
    "wl_comm_set_virtual_transport": (i32, [P, i32]),
    "wl_comm_set_periodic": (i32, [P, i32]),
    "wl_comm_halo_async": (i32, [P, P, G, i32, i32, P]),
    "wl_comm_combine_test": (i32, [P, P, P, P]),
    "wl_comm_destroy": (i32, [P]),
    "wl_comm_rank": (i32, [P]),
    "wl_comm_size": (i32, [P]),
    "wl_comm_stats": (i32, [P, C.POINTER(C.c_int64)]),
    "wl_halo_exchange": (i32, [P, P, G, i32, i32, P]),
    "wl_allgather_planes": (i32, [P, P, G, i32, P]),
    "wl_grid_slab": (i32, [G, i32, C.POINTER(C.c_int32), i32, i32, i32]),
    "wl_sim_create_slab": (i32, [C.POINTER(P), C.POINTER(wl_sim_desc), P]),
    "wl_launch_count": (C.c_long, []),
    "wl_reset_process_options": (i32, []),
    "wl_placement_scores": (i32, [C.POINTER(C.c_double), i32]),
    "wl_sim_counter": (i32, [P, C.c_char_p, C.POINTER(C.c_long)]),
    "wl_prof_enable": (i32, [i32]),
    "wl_prof_read": (i32, [i32, C.POINTER(i32), C.POINTER(f64)]),
    "wl_sim_pressure_force_sphere": (i32, [P, C.POINTER(f32), f32, C.POINTER(f64), P]),
    "wl_measure_body": (i32, [P, P, P, P, G, C.POINTER(wl_body), f32, i32, C.c_uint32, P]),
    "wl_pressure_force_body": (i32, [P, G, C.POINTER(wl_body), C.POINTER(f64), P]),
    "wl_viscous_force_body": (i32, [P, G, f32, C.POINTER(wl_body), C.POINTER(f64), P]),
    "wl_pressure_moment_body": (i32, [C.POINTER(f32), P, G, C.POINTER(wl_body), C.POINTER(f64), P]),
    "wl_viscous_moment_body": (i32, [C.POINTER(f32), P, G, f32, C.POINTER(wl_body), C.POINTER(f64), P]),
    "wl_sim_pressure_moment_body": (i32, [P, C.POINTER(f32), C.POINTER(wl_body), C.POINTER(f64), P]),
    "wl_sim_viscous_moment_body": (i32, [P, C.POINTER(f32), C.POINTER(wl_body), C.POINTER(f64), P]),
    "wl_sim_measure_body": (i32, [P, C.POINTER(wl_body), f32, P]),
    "wl_sim_pressure_force_body": (i32, [P, C.POINTER(wl_body), C.POINTER(f64), P]),
    "wl_sim_viscous_force_body": (i32, [P, C.POINTER(wl_body), C.POINTER(f64), P]),

    "wl_sim_viscous_force_sphere": (i32, [P, C.POINTER(f32), f32, C.POINTER(f64), P]),
}

_lib = None


class WlError(RuntimeError):
    pass


def lib():
    """Load libwlhip.so (built by waterlily.jl_amd/csrc/Makefile).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WlError(f"{LIB_PATH} not found: build it with `make -C waterlily.jl_amd/csrc` (no CPU fallback exists)")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise WlError(f"libwlhip error {rc}: {lib().wl_last_error_string().decode(errors='replace')}")
