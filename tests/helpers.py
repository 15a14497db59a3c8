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


def rk3_numpy(x, u, b):
    """vectorised rk3 of the slew dynamics for (T, N-1, .) arrays; rows k, k, k+1 (dtau = 1, tau0 = 0)"""
    Bt = b.Btab[b.btab_idx]                       # (T, n_tab, 3)
    Jd = np.array([b.Jmat[0, 0], b.Jmat[0, 4], b.Jmat[0, 8]])   # diagonal inertia in this workload
    h = 0.2

    def f(x, bb):
        w, q = x[..., :3], x[..., 3:]
        q = q / np.linalg.norm(q, axis=-1, keepdims=True)
        s, v = q[..., :1], q[..., 1:]
        qd = 0.5 * np.concatenate([-np.sum(v * w, -1, keepdims=True), s * w + np.cross(v, w)], -1)
        BB = bb + 2 * np.cross(v, np.cross(v, bb) + s * bb)
        tau = np.cross(u * 1e-2, BB)
        wd = (tau - np.cross(w, Jd * w)) / Jd
        return np.concatenate([wd, qd], -1)

    n = x.shape[1]
    b0 = Bt[:, :n]
    b2 = Bt[:, 1:n + 1]
    k1 = f(x, b0) * h
    k2 = f(x + k1 / 2, b0) * h
    k3 = f(x - k1 + 2 * k2, b2) * h
    return x + (k1 + 4 * k2 + k3) / 6
