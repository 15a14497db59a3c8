"""Helpers shared by the CPU and GPU test tiers: golden-fixture IO and option plumbing."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BATCH_FIELDS = ("x0", "xf", "Btab", "btab_idx", "tau0", "dtau", "dt", "Jmat", "Qd", "Qfd", "Rd", "ulo", "uhi", "U0")
OPT_FIELDS = ("integrator", "max_outer", "max_inner", "max_linesearch", "dj_counter_limit", "cost_tol", "grad_tol",
              "constraint_tol", "penalty_init", "penalty_scale", "penalty_max", "dual_max", "reg_init", "reg_scale",
              "reg_min", "reg_max", "reg_fp", "ls_lower", "ls_upper", "max_state", "u_scale", "terminal_mask", "error_state")


def save_case(path, batch, opts, res):
    d = {f"in_{k}": getattr(batch, k) for k in BATCH_FIELDS}
    d["in_N"] = np.int64(batch.N)
    for k in OPT_FIELDS:
        d[f"opt_{k}"] = np.float64(getattr(opts, k))
    d["out_X"], d["out_U"], d["out_K"], d["out_stats"] = res["X"], res["U"], res["K"], res["stats"]
    np.savez_compressed(path, **d)


def load_case(name, pkg, ol):
    z = np.load(os.path.join(GOLDEN_DIR, name))
    ss = pkg.slew_setup
    b = ss.SlewBatch(N=int(z["in_N"]), n_tab=int(z["in_Btab"].shape[1]),
                     **{k: np.ascontiguousarray(z[f"in_{k}"]) for k in BATCH_FIELDS})
    o = ol.default_options()
    for k in OPT_FIELDS:
        if f"opt_{k}" not in z.files:     # fixtures written before the field existed
            continue
        v = float(z[f"opt_{k}"])
        setattr(o, k, int(v) if isinstance(getattr(o, k), int) else v)
    ref = dict(X=z["out_X"], U=z["out_U"], K=z["out_K"], stats=z["out_stats"])
    return b, o, ref


def golden_cases():
    return sorted(f for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith(("stages_", "igrf12syn_")))


def stage_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def abi_options_like(o, pkg, N, n_tab):
    """copy an oracle-side Options (same ctypes struct) and fill the batch dimensions"""
    a = o.copy()
    a.n_knots, a.n_tab, a.precision = N, n_tab, 64
    return a
