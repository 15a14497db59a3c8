"""Closed-loop tracking of solved slews on the GPU — the caller right after ``solve!`` in the reference.

``attitude_simulation(f!, f_gains!, integration, X, U, dt, x0_lqr, t0, tf, Q_lqr, R_lqr, Qf_lqr)``
(src/attitude_controller.jl:1-48) computes TVLQR gains around the optimised trajectory (``attitude_lqr``, :50-119),
simulates the noisy plant (src/simulator.jl) under ``U_sim = U - K dX`` and returns ``X_sim, U_sim, dX, K``;
src/monte_carlo.jl:242-262 then turns ``X_sim`` into a slew time / failure flag. Here the same call runs for a whole
batch through ``tsat_tvlqr_batch``; randomness is an explicit input so that runs are reproducible and checkable.
"""
import ctypes as C

import numpy as np

from . import _abi
from .slew_setup import SlewBatch, qmult


def tvlqr_weights(T, alpha=10.0, beta=10.0, r=7.5e3):
    """Q_lqr = diag(alpha 1_3, beta 1_3), Qf_lqr = 100 Q_lqr, R_lqr = r I (src/TortoiseSat.jl:251-260; the
    Monte-Carlo uses r = 0.5e3, src/monte_carlo.jl:228)."""
    Qd = np.tile(np.r_[np.full(3, alpha), np.full(3, beta)], (T, 1))
    return Qd, 100.0 * Qd, np.full((T, 3), float(r))


def perturbed_initial_state(x0, rng, sigma=(np.pi / 180.0) ** 2):
    """x0_lqr of src/TortoiseSat.jl:227-234: rates kept, attitude rotated by a random small rotation vector."""
    x0 = np.asarray(x0, dtype=np.float64)
    out = x0.copy()
    for t in range(x0.shape[0]):
        nq = rng.standard_normal(3) * sigma
        th = np.linalg.norm(nq)
        out[t, 3:7] = qmult(x0[t, 3:7], np.r_[np.cos(th / 2.0), nq / th * np.sin(th / 2.0)])
    return out


def simulator_noise(T, N, rng):
    """The draws ``simulator`` makes per dynamics evaluation (src/simulator.jl:5,10,22), laid out (T, N-1, 4, 9):
    gyro noise randn(3)(0.38 deg)^2, attitude-noise rotation vector randn(3)(1 deg)^2, field noise rand(3) 1e-10."""
    g = rng.standard_normal((T, N - 1, 4, 3)) * (0.38 * np.pi / 180.0) ** 2
    a = rng.standard_normal((T, N - 1, 4, 3)) * (1.0 * np.pi / 180.0) ** 2
    b = rng.random((T, N - 1, 4, 3)) * (1.0e-5) ** 2
    return np.ascontiguousarray(np.concatenate([g, a, b], axis=3))


_M0, _M1, _W0, _W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(key, ctr):
    """Philox4x32-10 on arrays: key (2,) or (..., 2), ctr (..., 4) of 32-bit words held in uint64. Returns (..., 4)."""
    k0, k1 = np.uint64(key[..., 0]) + np.zeros(ctr.shape[:-1], np.uint64), np.uint64(key[..., 1]) + np.zeros(ctr.shape[:-1], np.uint64)
    c0, c1, c2, c3 = (ctr[..., i].astype(np.uint64) for i in range(4))
    for r in range(10):
        if r:
            k0, k1 = (k0 + _W0) & _LO, (k1 + _W1) & _LO
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & _LO, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & _LO
    return np.stack([c0, c1, c2, c3], axis=-1)


def generated_noise(seed, ids, N, sigma_gyro=(0.38 * np.pi / 180.0) ** 2, sigma_att=(np.pi / 180.0) ** 2, field_amp=1e-10):
    """What ``noise_mode = 1`` draws inside the kernel (include/tortoise_hip.h), as an array (T, N-1, 4, 9) — the same
    run can then be repeated in array mode, or inspected."""
    ids = np.asarray(ids, dtype=np.int64).astype(np.uint64)
    T = ids.shape[0]
    key = np.array([np.uint64(seed) & _LO, np.uint64(seed) >> np.uint64(32)], dtype=np.uint64)
    ctr = np.zeros((T, N - 1, 4, 3, 4), dtype=np.uint64)
    ctr[..., 0] = (ids & _LO)[:, None, None, None]
    ctr[..., 1] = (ids >> np.uint64(32))[:, None, None, None]
    ctr[..., 2] = np.arange(N - 1, dtype=np.uint64)[None, :, None, None]
    ctr[..., 3] = (4 * np.arange(4, dtype=np.uint64))[None, None, :, None] + np.arange(3, dtype=np.uint64)[None, None, None, :]
    w = philox4x32_10(key, ctr).astype(np.float64)                       # (T, N-1, 4, 3, 4)
    u = (w + 0.5) / 4294967296.0

    def bm(a, b):
        r, th = np.sqrt(-2.0 * np.log(a)), 2.0 * np.pi * b
        return r * np.cos(th), r * np.sin(th)

    z0, z1 = bm(u[..., 0, 0], u[..., 0, 1]); z2, z3 = bm(u[..., 0, 2], u[..., 0, 3]); z4, z5 = bm(u[..., 1, 0], u[..., 1, 1])
    out = np.stack([sigma_gyro * z0, sigma_gyro * z1, sigma_gyro * z2, sigma_att * z3, sigma_att * z4, sigma_att * z5,
                    field_amp * u[..., 1, 2], field_amp * u[..., 1, 3], field_amp * u[..., 2, 0]], axis=-1)
    return np.ascontiguousarray(out)


def attitude_simulation(solver, batch: SlewBatch, X, U, x0_sim, Qd, Qfd, Rd, noise=None, linearize_dt_sq=True,
                        u_scale=1e-2, min_steps=10, w_tol=0.05, angle_tol=0.08727, noise_seed=None, noise_ids=None,
                        want_K=True, want_trajectories=True, rate_as_written=False, trial_ids=None):
    """Batched ``attitude_simulation`` + slew-time statistic. ``solver`` is an AugmentedLagrangianSolver (owns the GPU
    handle); X (T,N,7), U (T,N-1,3) are the solved trajectories — or both ``None`` to track the batch that is resident on
    the device right after ``solve_`` (no re-upload of trajectories and tables; ``batch`` must be the one just solved).
    Plant noise: ``noise`` array (T,N-1,4,9), or ``noise_seed`` (+ optional per-trajectory ``noise_ids``) to have the
    kernel draw it, or neither for the noise-free plant. Returns dict(X_sim, U_sim, K (T,N-1,6,3), stats);
    ``want_trajectories=False`` brings only the slew-time statistic back (X_sim = U_sim = None).
    ``rate_as_written=True`` evaluates the statistic as the reference's line reads, ``norm(sim_states[i][1:3,i])``
    (src/monte_carlo.jl:247): for every sample j the rate of sample i, the 1-based number of the trial — ``trial_ids`` (0-based,
    default: ``noise_ids``, else the position in the batch); the default takes the rate of sample j, which the line means."""
    lib = _abi.load()
    T, N = batch.T, batch.N
    o = _abi.TvlqrOptions()
    lib.tsat_tvlqr_default_options(C.byref(o))
    ids = None
    if noise_seed is not None:
        if noise is not None:
            raise ValueError("give either a noise array or a noise seed")
        o.noise_mode, o.noise_seed = 1, int(noise_seed)
        ids = None if noise_ids is None else np.ascontiguousarray(noise_ids, dtype=np.int64)
    elif noise_ids is not None and not rate_as_written:
        raise ValueError("noise_ids key the noise the kernel draws: they need noise_seed (or rate_as_written, whose trial numbers they are)")
    if rate_as_written:
        o.rate_as_written = 1
        if ids is None and noise_ids is not None:      # the trial numbers, whatever the noise comes from (an array, or none)
            ids = np.ascontiguousarray(noise_ids, dtype=np.int64)
        if trial_ids is not None:
            if ids is not None and not np.array_equal(ids, np.asarray(trial_ids, dtype=np.int64)):
                raise ValueError("trial_ids and noise_ids are the same index of the reference's loop: give one, or equal arrays")
            ids = np.ascontiguousarray(trial_ids, dtype=np.int64)
    o.n_knots, o.n_tab, o.linearize_dt_sq, o.min_steps = N, batch.n_tab, int(bool(linearize_dt_sq)), int(min_steps)
    o.u_scale, o.w_tol, o.angle_tol = float(u_scale), float(w_tol), float(angle_tol)
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    resident = X is None and U is None
    x0_sim, Qd, Qfd, Rd = c(x0_sim), c(Qd), c(Qfd), c(Rd)
    if not resident:
        X, U = c(X), c(U)
        if X.shape != (T, N, 7) or U.shape != (T, N - 1, 3):
            raise ValueError("array shapes do not match the batch")
    if x0_sim.shape != (T, 7) or Qd.shape != (T, 6) or Qfd.shape != (T, 6) or Rd.shape != (T, 3):
        raise ValueError("array shapes do not match the batch")
    if noise is not None:
        noise = c(noise)
        if noise.shape != (T, N - 1, 4, 9):
            raise ValueError("noise must be (T, N-1, 4, 9)")
    Xs = np.empty((T, N, 7)) if want_trajectories else None
    Us = np.empty((T, N - 1, 3)) if want_trajectories else None
    K = np.empty((T, N - 1, 6, 3)) if want_K else None
    nk = None if batch.n_knots is None else np.ascontiguousarray(batch.n_knots, dtype=np.int32)
    st = np.zeros(T, dtype=_abi.TVLQR_STATS_DTYPE)
    d = _abi.as_dp
    idp = None if ids is None else ids.ctypes.data_as(C.POINTER(C.c_int64))
    if resident:
        rc = lib.tsat_tvlqr_resident(solver._h, C.byref(o), d(Qd), d(Qfd), d(Rd), d(x0_sim), d(noise), d(Xs), d(Us), d(K),
                                     st.ctypes.data_as(C.c_void_p), idp)
        solver._check(rc, "tsat_tvlqr_resident")
        return dict(X_sim=Xs, U_sim=Us, K=K, stats=st)
    rc = lib.tsat_tvlqr_batch(solver._h, C.byref(o), T, batch.Btab.shape[0], d(X), d(U), d(batch.xf), d(batch.Btab),
                              _abi.as_ip(batch.btab_idx), d(batch.tau0), d(batch.dtau), d(batch.dt), d(batch.Jmat),
                              d(Qd), d(Qfd), d(Rd), d(x0_sim), d(noise), d(Xs), d(Us), d(K), st.ctypes.data_as(C.c_void_p),
                              _abi.as_ip(nk), idp)
    solver._check(rc, "tsat_tvlqr_batch")
    return dict(X_sim=Xs, U_sim=Us, K=K, stats=st)
