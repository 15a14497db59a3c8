"""Receding-horizon re-solve on the GPU (BASELINE.json configs[4]).

Not in the reference — it follows its optimised plan with TVLQR (src/attitude_controller.jl:1-48). Defined here as the
obvious closed loop around the same solve: at every control step re-solve the horizon from the
current state, warm-started with the previous plan shifted by one knot, apply the first control to the noise-free plant,
move on. The whole loop runs on the resident batch through ``tsat_mpc_run``; only the closed-loop history comes back.
"""
import ctypes as C

import numpy as np

from . import _abi
from .trajopt import BatchProblem


def receding_horizon(prob, solver, n_steps, plant_integrator=4, max_outer=1, max_inner=3):
    """``prob``: BatchProblem (or SlewBatch via BatchProblem.from_arrays) — its N is the re-solve horizon, its field
    tables must cover ``n_steps`` further rows. Returns dict(X_hist (T, n_steps+1, 7), U_hist (T, n_steps, 3), stats
    (last solve), ms (device time of the loop)); the last plan stays resident (``solver.download()``)."""
    lib = _abi.load()
    b = prob.arrays
    opts = solver.opts
    o = opts.to_abi(b.N, b.n_tab, prob.integrator, prob.terminal_mask, error_state=prob.error_state)
    o.max_outer, o.max_inner = int(max_outer), int(max_inner)
    solver.upload(b, o.max_linesearch)
    T = b.T
    Xh = np.empty((T, n_steps + 1, 7)); Uh = np.empty((T, n_steps, 3))
    st = np.zeros(T, dtype=_abi.STATS_DTYPE)
    ms = C.c_float(0.0)
    rc = lib.tsat_mpc_run(solver._h, C.byref(o), int(n_steps), int(plant_integrator), _abi.as_dp(Xh), _abi.as_dp(Uh),
                          st.ctypes.data_as(C.c_void_p), C.byref(ms))
    solver._check(rc, "tsat_mpc_run")
    return dict(X_hist=Xh, U_hist=Uh, stats=st, ms=float(ms.value))
