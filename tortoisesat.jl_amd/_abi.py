"""ctypes view of include/tortoise_hip.h (struct layouts + function prototypes).

The structs here are the single Python definition of ``tsat_options`` / ``tsat_stats``; the test-side checker
binds the same classes so that checker and product agree on the layout.
"""
import ctypes as C
import os
import sys

import numpy as np

TSAT_NX, TSAT_NU, TSAT_NC = 7, 3, 6
TSAT_MAX_LINESEARCH = 32
TSAT_COMM_ID_BYTES = 128
TSAT_CONVERGED, TSAT_MAX_OUTER, TSAT_REG_FAIL, TSAT_DIVERGED = 0, 1, 2, 3


class Options(C.Structure):
    """``tsat_options`` — replaces AugmentedLagrangianSolverOptions (src/TortoiseSat.jl:194-196)."""

    _fields_ = [
        ("n_knots", C.c_int32), ("n_tab", C.c_int32), ("integrator", C.c_int32), ("precision", C.c_int32),
        ("max_outer", C.c_int32), ("max_inner", C.c_int32), ("max_linesearch", C.c_int32),
        ("dj_counter_limit", C.c_int32),
        ("cost_tol", C.c_double), ("grad_tol", C.c_double), ("constraint_tol", C.c_double),
        ("penalty_init", C.c_double), ("penalty_scale", C.c_double), ("penalty_max", C.c_double),
        ("dual_max", C.c_double),
        ("reg_init", C.c_double), ("reg_scale", C.c_double), ("reg_min", C.c_double), ("reg_max", C.c_double),
        ("reg_fp", C.c_double),
        ("ls_lower", C.c_double), ("ls_upper", C.c_double), ("max_state", C.c_double), ("u_scale", C.c_double),
        ("terminal_mask", C.c_int32), ("error_state", C.c_int32),
    ]

    def copy(self):
        o = Options()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(Options))
        return o


class Stats(C.Structure):
    """``tsat_stats`` — per-trajectory outcome record."""

    _fields_ = [
        ("status", C.c_int32), ("outer_iters", C.c_int32), ("inner_iters", C.c_int32), ("ls_trials", C.c_int32),
        ("n_backward", C.c_int32), ("n_forward", C.c_int32), ("bp_restarts", C.c_int32), ("fp_fails", C.c_int32),
        ("cost", C.c_double), ("cost_al", C.c_double), ("c_max", C.c_double), ("grad", C.c_double),
    ]


STATS_DTYPE = np.dtype(
    [("status", "<i4"), ("outer_iters", "<i4"), ("inner_iters", "<i4"), ("ls_trials", "<i4"),
     ("n_backward", "<i4"), ("n_forward", "<i4"), ("bp_restarts", "<i4"), ("fp_fails", "<i4"),
     ("cost", "<f8"), ("cost_al", "<f8"), ("c_max", "<f8"), ("grad", "<f8")]
)
assert STATS_DTYPE.itemsize == C.sizeof(Stats) == 64

class TvlqrOptions(C.Structure):
    """``tsat_tvlqr_options`` — closed-loop tracking of a solved slew (src/attitude_controller.jl:1-119)."""

    _fields_ = [("n_knots", C.c_int32), ("n_tab", C.c_int32), ("linearize_dt_sq", C.c_int32), ("min_steps", C.c_int32),
                ("u_scale", C.c_double), ("w_tol", C.c_double), ("angle_tol", C.c_double),
                ("noise_mode", C.c_int32), ("rate_as_written", C.c_int32), ("noise_seed", C.c_uint64),
                ("sigma_gyro", C.c_double), ("sigma_att", C.c_double), ("field_amp", C.c_double)]


class BtableOptions(C.Structure):
    """``tsat_btable_options`` — field-table generation (src/magnetic_toolbox.jl:33-106)."""

    _fields_ = [("n_half", C.c_int32), ("reserved", C.c_int32), ("mjd", C.c_double), ("gm", C.c_double),
                ("r_igrf_km", C.c_double), ("date", C.c_double)]


TVLQR_STATS_DTYPE = np.dtype([("slew_index", "<i4"), ("failed", "<i4"), ("slew_time", "<f8"), ("final_w_norm", "<f8"),
                              ("final_angle", "<f8")])
assert TVLQR_STATS_DTYPE.itemsize == 32

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

#: every symbol include/tortoise_hip.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "tsat_version": (C.c_int, []),
    "tsat_default_options": (None, [C.POINTER(Options)]),
    "tsat_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "tsat_destroy": (C.c_int, [C.c_void_p]),
    "tsat_last_error": (C.c_char_p, [C.c_void_p]),
    "tsat_solve_batch": (C.c_int, [C.c_void_p, C.POINTER(Options), C.c_int64, C.c_int64,
                                   _dp, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                   _dp, _dp, _dp, C.c_void_p]),
    "tsat_batch_reserve": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    "tsat_batch_upload": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "tsat_batch_knots": (C.c_int, [C.c_void_p, _ip]),
    "tsat_batch_run": (C.c_int, [C.c_void_p, C.POINTER(Options), C.POINTER(C.c_float)]),
    "tsat_batch_download": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_void_p]),
    "tsat_batch_export_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsat_batch_bytes": (C.c_int64, [C.c_void_p]),
    "tsat_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "tsat_workspace_trim": (C.c_int, [C.c_void_p, C.c_int32]),
    "tsat_btable_generation": (C.c_int64, [C.c_void_p]),
    "tsat_comm_available": (C.c_int, []),
    "tsat_batch_trace": (C.c_int, [C.c_void_p, C.c_int32]),
    "tsat_batch_trace_download": (C.c_int, [C.c_void_p, _dp]),
    "tsat_set_kernel_variant": (C.c_int, [C.c_void_p, C.c_int32]),
    "tsat_set_endgame": (C.c_int, [C.c_void_p, C.c_int32]),
    "tsat_selected_build": (C.c_int, [C.c_void_p, C.POINTER(Options), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tsat_tvlqr_resident": (C.c_int, [C.c_void_p, C.POINTER(TvlqrOptions), _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_void_p,
                                      C.POINTER(C.c_int64)]),
    "tsat_mpc_run": (C.c_int, [C.c_void_p, C.POINTER(Options), C.c_int32, C.c_int32, _dp, _dp, C.c_void_p, C.POINTER(C.c_float)]),
    "tsat_mpc_tally": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "tsat_comm_unique_id": (C.c_int, [C.c_void_p]),
    "tsat_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "tsat_sweep_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "tsat_comm_destroy": (C.c_int, [C.c_void_p]),
    "tsat_btable_default_options": (None, [C.POINTER(BtableOptions)]),
    "tsat_bryson_eigen_axis_batch": (C.c_int, [C.c_int64, _ip, C.c_double, C.c_double, C.c_double, _dp, _dp, C.c_double, C.c_double,
                                               _dp, _dp, _dp]),
    "tsat_btable_batch": (C.c_int, [C.c_void_p, C.POINTER(BtableOptions), C.c_int64, _dp, _dp, _dp, _dp, _dp]),
    "tsat_horizon_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, _dp, _dp, _dp, _ip, _dp]),
    "tsat_tvlqr_default_options": (None, [C.POINTER(TvlqrOptions)]),
    "tsat_tvlqr_batch": (C.c_int, [C.c_void_p, C.POINTER(TvlqrOptions), C.c_int64, C.c_int64, _dp, _dp, _dp, _dp, _ip,
                                   _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_void_p, _ip, C.POINTER(C.c_int64)]),
}

LIB_NAME = "libtortoise_hip.so"


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", LIB_NAME)


_lib = None


def load():
    """dlopen the HIP solver library and bind every prototype. Raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64, and a process keeps whichever HIP runtime it loaded first: if
    # this library came first, a later `import torch` finds no GPU. bench.py and sweep.py hand torch device buffers to
    # this library, so torch — when installed — is imported before the dlopen (TSAT_NO_TORCH_PRELOAD=1 skips it).
    if "torch" not in sys.modules and os.environ.get("TSAT_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback for the solve path."
        )
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def as_dp(a):
    """double* of a C-contiguous float64 array (or NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def as_ip(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_ip)
