"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see tsat_oracle.cpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from tsat_loader import load_package  # noqa: E402

_abi = load_package()._abi
Options, Stats, STATS_DTYPE = _abi.Options, _abi.Stats, _abi.STATS_DTYPE

_dp = C.POINTER(C.c_double)
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("TSAT_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")      # (another build of the oracle: tools/oracle_self_perturbation.py)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.orc_rk_scalar.restype = C.c_double
    lib.orc_rk_scalar.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
    lib.orc_solve_batch.restype = C.c_int
    lib.orc_num_procs.restype = C.c_int
    _lib = lib
    return lib


def _p(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def default_options():
    o = Options()
    load().orc_default_options(C.byref(o))
    return o


def _call_vec(name, out_n, *args):
    lib = load()
    keep, cargs = [], []
    for a in args:
        if isinstance(a, (int, np.integer)):
            cargs.append(C.c_int(int(a)))
        elif isinstance(a, float):
            cargs.append(C.c_double(a))
        else:
            arr, ptr = _p(a)
            keep.append(arr)
            cargs.append(ptr)
    outs = []
    for n in (out_n if isinstance(out_n, (tuple, list)) else (out_n,)):
        o = np.zeros(n)
        outs.append(o)
        cargs.append(o.ctypes.data_as(_dp))
    getattr(lib, name)(*cargs)
    return outs[0] if len(outs) == 1 else tuple(outs)


def qmult(q1, q2): return _call_vec("orc_qmult", 4, q1, q2)
def qrot(q, r): return _call_vec("orc_qrot", 3, q, r)
def qinv(q): return _call_vec("orc_qinv", 4, q)
def hat(x): return _call_vec("orc_hat", 9, x).reshape(3, 3).T          # column-major -> [r,c]
def gmat(q): return _call_vec("orc_gmat", 12, q).reshape(3, 4).T       # 4x3
def emat(q): return _call_vec("orc_emat", 42, q).reshape(7, 6)
def inv3(M): return _call_vec("orc_inv3", 9, np.asarray(M).T.reshape(9)).reshape(3, 3).T


def _jcm(J):
    return np.ascontiguousarray(np.asarray(J, dtype=np.float64).T.reshape(9))


def deriv8(x8, u, Btab, Nglob, J, tspan, u_scale=1e-2):
    Btab = np.ascontiguousarray(Btab, dtype=np.float64)
    return _call_vec("orc_deriv8", 8, x8, u, Btab, int(Btab.shape[0]), int(Nglob), _jcm(J), float(tspan), float(u_scale))


def attitude_dynamics(x7, u, BB, J): return _call_vec("orc_attitude_dynamics", 7, x7, u, BB, _jcm(J))
def dyn7(x, u, b, J, u_scale=1e-2): return _call_vec("orc_dyn7", 7, x, u, b, _jcm(J), float(u_scale))


def rk_step(integ, x, u, b0, b1, b2, h, J, u_scale=1e-2):
    return _call_vec("orc_rk_step", 7, int(integ), x, u, b0, b1, b2, float(h), _jcm(J), float(u_scale))


def rk_scalar(integ, lam, x, h):
    return load().orc_rk_scalar(int(integ), float(lam), float(x), float(h))


def discrete_jacobian(integ, x, u, b0, b1, b2, h, J, u_scale=1e-2):
    A, B = _call_vec("orc_discrete_jacobian", (49, 21), int(integ), x, u, b0, b1, b2, float(h), _jcm(J), float(u_scale))
    return A.reshape(7, 7), B.reshape(7, 3)


def quaternion_error(X1, X2): return _call_vec("orc_quaternion_error", 7, X1, X2)


def reduce_error_state(A, B, qk, qn):
    Ah, Bh = _call_vec("orc_reduce_error_state", (36, 18), A, B, qk, qn)
    return Ah.reshape(6, 6), Bh.reshape(6, 3)


def quaternion_expansion(Q, qlin, x):
    Qxx, Qx = _call_vec("orc_quaternion_expansion", (36, 6), Q, qlin, x)
    return Qxx.reshape(6, 6), Qx


def tvlqr_riccati(Ah, Bh, Q, R, Qf):
    Ah = np.ascontiguousarray(Ah, dtype=np.float64)
    N = Ah.shape[0] + 1
    K = _call_vec("orc_tvlqr_riccati", 18 * (N - 1), int(N), Ah, Bh, Q, R, Qf)
    return K.reshape(N - 1, 3, 6)


def solve_batch(batch, opts, nthreads=1, want_K=True, trace_rows=0):
    """Oracle solve of a SlewBatch. Returns dict(X (T,N,7), U (T,N-1,3), K (T,N-1,7,3) ABI order, stats, trace)."""
    lib = load()
    T, N = batch.T, batch.N
    o = opts.copy()
    o.n_knots, o.n_tab = N, batch.n_tab
    X = np.zeros((T, N, 7)); U = np.zeros((T, N - 1, 3))
    K = np.zeros((T, N - 1, 7, 3)) if want_K else None
    stats = np.zeros(T, dtype=STATS_DTYPE)
    trace = np.zeros((T, trace_rows, 8)) if trace_rows else None
    d = lambda a: a.ctypes.data_as(_dp)
    rc = lib.orc_solve_batch(
        C.byref(o), C.c_int64(T), C.c_int64(batch.Btab.shape[0]), d(batch.x0), d(batch.xf), d(batch.Btab),
        batch.btab_idx.ctypes.data_as(C.POINTER(C.c_int32)), d(batch.tau0), d(batch.dtau), d(batch.dt),
        d(batch.Jmat), d(batch.Qd), d(batch.Qfd), d(batch.Rd), d(batch.ulo), d(batch.uhi), d(batch.U0),
        d(X), d(U), d(K) if want_K else None, stats.ctypes.data_as(C.c_void_p), C.c_int(nthreads),
        d(trace) if trace_rows else None, C.c_int(trace_rows),
        None if batch.n_knots is None else np.ascontiguousarray(batch.n_knots, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise RuntimeError(f"orc_solve_batch failed rc={rc}")
    return dict(X=X, U=U, K=K, stats=stats, trace=trace)


def mpc_batch(batch, opts, n_steps, plant_integrator=4, nthreads=1):
    """Oracle receding-horizon loop (see tsat_mpc_run). Returns dict(X_hist (T,n_steps+1,7), U_hist (T,n_steps,3),
    stats (last solve), X, U (last plan))."""
    lib = load()
    T, N = batch.T, batch.N
    o = opts.copy()
    o.n_knots, o.n_tab = N, batch.n_tab
    Xh = np.zeros((T, n_steps + 1, 7)); Uh = np.zeros((T, n_steps, 3))
    X = np.zeros((T, N, 7)); U = np.zeros((T, N - 1, 3))
    stats = np.zeros(T, dtype=STATS_DTYPE)
    d = lambda a: a.ctypes.data_as(_dp)
    rc = lib.orc_mpc_batch(
        C.byref(o), C.c_int64(T), C.c_int64(batch.Btab.shape[0]), d(batch.x0), d(batch.xf), d(batch.Btab),
        batch.btab_idx.ctypes.data_as(C.POINTER(C.c_int32)), d(batch.tau0), d(batch.dtau), d(batch.dt),
        d(batch.Jmat), d(batch.Qd), d(batch.Qfd), d(batch.Rd), d(batch.ulo), d(batch.uhi), d(batch.U0),
        C.c_int32(n_steps), C.c_int32(plant_integrator), d(Xh), d(Uh), stats.ctypes.data_as(C.c_void_p), d(X), d(U),
        C.c_int(nthreads),
        None if batch.n_knots is None else np.ascontiguousarray(batch.n_knots, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise RuntimeError(f"orc_mpc_batch failed rc={rc}")
    return dict(X_hist=Xh, U_hist=Uh, stats=stats, X=X, U=U)


TvlqrOptions, TVLQR_STATS_DTYPE = _abi.TvlqrOptions, _abi.TVLQR_STATS_DTYPE


def tvlqr_default_options():
    o = TvlqrOptions()
    load().orc_tvlqr_default_options(C.byref(o))
    return o


def philox4x32_10(key, ctr):
    out = (C.c_uint32 * 4)()
    load().orc_philox4x32_10((C.c_uint32 * 2)(*key), (C.c_uint32 * 4)(*ctr), out)
    return [int(v) for v in out]


def plant_noise(seed, tid, knot, stage, sg, sa, fa):
    nz = np.zeros(9)
    load().orc_plant_noise(C.c_uint64(seed), C.c_int64(tid), C.c_int32(knot), C.c_int32(stage), C.c_double(sg), C.c_double(sa),
                           C.c_double(fa), nz.ctypes.data_as(_dp))
    return nz


def tvlqr_batch(batch, X, U, Qd, Qfd, Rd, x0_sim, noise=None, opts=None, nthreads=1, noise_ids=None):
    """Oracle closed-loop tracking of solved trajectories (see tsat_tvlqr_batch). Returns X_sim, U_sim, K_lqr, stats."""
    lib = load()
    T, N = batch.T, batch.N
    o = opts or tvlqr_default_options()
    o.n_knots, o.n_tab = N, batch.n_tab
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    X, U, Qd, Qfd, Rd, x0_sim = c(X), c(U), c(Qd), c(Qfd), c(Rd), c(x0_sim)
    noise = None if noise is None else c(noise)
    Xs = np.zeros((T, N, 7)); Us = np.zeros((T, N - 1, 3)); K = np.zeros((T, N - 1, 6, 3))
    st = np.zeros(T, dtype=TVLQR_STATS_DTYPE)
    d = lambda a: a.ctypes.data_as(_dp)
    rc = lib.orc_tvlqr_batch(C.byref(o), C.c_int64(T), C.c_int64(batch.Btab.shape[0]), d(X), d(U), d(batch.xf), d(batch.Btab),
                             batch.btab_idx.ctypes.data_as(C.POINTER(C.c_int32)), d(batch.tau0), d(batch.dtau), d(batch.dt),
                             d(batch.Jmat), d(Qd), d(Qfd), d(Rd), d(x0_sim), d(noise) if noise is not None else None,
                             d(Xs), d(Us), d(K), st.ctypes.data_as(C.c_void_p), C.c_int(nthreads),
                             None if batch.n_knots is None else np.ascontiguousarray(batch.n_knots, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32)),
                             None if noise_ids is None else np.ascontiguousarray(noise_ids, dtype=np.int64).ctypes.data_as(C.POINTER(C.c_int64)))
    if rc != 0:
        raise RuntimeError(f"orc_tvlqr_batch failed rc={rc}")
    return dict(X_sim=Xs, U_sim=Us, K=K, stats=st)


class BtableOptions(C.Structure):
    _fields_ = [("n_half", C.c_int32), ("reserved", C.c_int32), ("mjd", C.c_double), ("gm", C.c_double),
                ("r_igrf_km", C.c_double), ("date", C.c_double)]


def igrf12(date, r_m, lat, lon):
    return _call_vec("orc_igrf12", 3, float(date), float(r_m), float(lat), float(lon))


def kep_eci(kep, t0, GM):
    return _call_vec("orc_kep_eci", (3, 3), np.asarray(kep, dtype=np.float64), float(t0), float(GM))


def btable_batch(kep, t0, tf, n_half, mjd=58155.0, gm=3.986004418e5, r_igrf_km=6771.0, date=2019.0, want_pos=True):
    """Oracle magnetic_simulation for T orbits. Returns (Btab (T, 2N, 3), pos (T, 2N+1, 3))."""
    lib = load()
    kep = np.ascontiguousarray(kep, dtype=np.float64)
    T = kep.shape[0]
    t0 = np.ascontiguousarray(np.broadcast_to(t0, (T,)), dtype=np.float64)
    tf = np.ascontiguousarray(np.broadcast_to(tf, (T,)), dtype=np.float64)
    o = BtableOptions(int(n_half), 0, mjd, gm, r_igrf_km, date)
    B = np.zeros((T, 2 * n_half, 3)); pos = np.zeros((T, 2 * n_half + 1, 3))
    d = lambda a: a.ctypes.data_as(_dp)
    rc = lib.orc_btable_batch(C.byref(o), C.c_int64(T), d(kep), d(t0), d(tf), d(B), d(pos) if want_pos else None)
    if rc != 0:
        raise RuntimeError("orc_btable_batch failed")
    return B, pos


def horizon_batch(Btab, dt_row, cutoff, want_all=False):
    """Oracle magnetic_gramian + condition_based_time. Btab (T, n_rows, 3). Returns (tf_index, cond_at[, cond_all])."""
    lib = load()
    Btab = np.ascontiguousarray(Btab, dtype=np.float64)
    T, n = Btab.shape[0], Btab.shape[1]
    dt_row = np.ascontiguousarray(np.broadcast_to(dt_row, (T,)), dtype=np.float64)
    cutoff = np.ascontiguousarray(np.broadcast_to(cutoff, (T,)), dtype=np.float64)
    idx = np.zeros(T, dtype=np.int32); cat = np.zeros(T)
    call = np.zeros((T, n)) if want_all else None
    d = lambda a: a.ctypes.data_as(_dp)
    rc = lib.orc_horizon_batch(C.c_int64(T), C.c_int32(n), d(Btab), d(dt_row), d(cutoff), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                               d(cat), d(call) if want_all else None)
    if rc != 0:
        raise RuntimeError("orc_horizon_batch failed")
    return (idx, cat, call) if want_all else (idx, cat)


def num_procs():
    return load().orc_num_procs()
