"""Horizon selection on the GPU — the caller right before the solve in the reference.

``B_gram = magnetic_gramian(B, dt)`` + ``condition_based_time(B_gram, cutoff)`` (src/magnetic_toolbox.jl:1-31) pick, per
orbit, the first table row at which the magnetic control Gramian is well enough conditioned; the scripts turn it into
the slew horizon ``t_final = index*(tf-t0)/N`` and the knot count ``floor(t_final/dt)`` (src/TortoiseSat.jl:73-86,
src/monte_carlo.jl:137-145). ``condition_based_time`` runs that for a batch of coarse tables through
``tsat_horizon_batch``; ``knots_from_index`` is the script arithmetic that follows.
"""
import ctypes as C

import numpy as np

from . import _abi


def condition_based_time(solver, Btab, dt_row, cutoff, resident_shape=None):
    """Btab (T, n_rows, 3) coarse field tables, dt_row / cutoff scalars or (T,). Returns (tf_index (T,) 1-based with 0 =
    never below the cutoff, cond_at (T,)). ``Btab = None`` with ``resident_shape = (T, n_rows)``: the tables the last
    ``magnetic_simulation(..., host=False)`` left on the device."""
    lib = _abi.load()
    if Btab is None:
        T, n = int(resident_shape[0]), int(resident_shape[1])
    else:
        Btab = np.ascontiguousarray(Btab, dtype=np.float64)
        if Btab.ndim != 3 or Btab.shape[2] != 3:
            raise ValueError("Btab must be (T, n_rows, 3)")
        T, n = Btab.shape[0], Btab.shape[1]
    dt_row = np.ascontiguousarray(np.broadcast_to(np.asarray(dt_row, dtype=np.float64), (T,)))
    cutoff = np.ascontiguousarray(np.broadcast_to(np.asarray(cutoff, dtype=np.float64), (T,)))
    idx = np.zeros(T, dtype=np.int32)
    cond = np.zeros(T)
    rc = lib.tsat_horizon_batch(solver._h, T, n, _abi.as_dp(Btab), _abi.as_dp(dt_row), _abi.as_dp(cutoff), _abi.as_ip(idx),
                                _abi.as_dp(cond))
    solver._check(rc, "tsat_horizon_batch")
    return idx, cond


def knots_from_index(tf_index, span, n_rows, dt=0.2, t0=0.0):
    """t_final = index*(tf-t0)/N and N_knots = floor((t_final-t0)/dt) (src/TortoiseSat.jl:82-85). This is the single-slew
    script's `convert(Int64,floor((t_final-t0)/dt))`, a plain fp64 floor (2.4/0.2 -> 11), reproduced as written; the
    Monte-Carlo script's `range(t0,step=.2,stop=t_final)` counts in twice precision — `monte_carlo.knot_counts`."""
    t_final = np.asarray(tf_index, dtype=np.float64) * float(span) / float(n_rows)
    return t_final, np.floor((t_final - t0) / dt).astype(np.int32)
