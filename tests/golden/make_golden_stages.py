#!/usr/bin/env python3
"""Generates tests/golden/stages_*.npz: fixed inputs and oracle outputs of the stages around the solve — field tables
(tsat_btable_batch), horizon selection (tsat_horizon_batch), closed-loop tracking with kernel-drawn noise
(tsat_tvlqr_batch, noise_mode = 1) and the receding-horizon loop (tsat_mpc_run). Produced by this repository's CPU oracle
("parity unpinned", see make_golden.py); the CPU tier checks the oracle and the emulated kernels against them, the GPU
tier the real kernels.

    python tests/golden/make_golden_stages.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(HERE)]
import numpy as np  # noqa: E402
import oracle_lib as ol  # noqa: E402
from tsat_loader import load_package  # noqa: E402

pkg = load_package()
ss = pkg.slew_setup

KEP = np.array([[0.0, 6771.0, 96.6, 30.0, 0.0, 40.0], [0.0, 6578.0, 96.0, 0.0, 0.0, 90.0], [0.02, 7000.0, 51.6, 120.0, 33.0, 250.0]])


def solve_opts(**kw):
    o = ol.default_options()
    o.max_outer, o.max_inner, o.dj_counter_limit = 3, 6, 1
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def tracking_case():
    b = ss.workload_monte_carlo(T=3, N=60, seed=77)
    b.n_knots = np.array([60, 37, 12], dtype=np.int32)
    r = ol.solve_batch(b, solve_opts())
    Qd, Qfd, Rd = pkg.tracking.tvlqr_weights(b.T, r=0.5e3)
    x0s = pkg.tracking.perturbed_initial_state(b.x0, np.random.default_rng(5))
    o = ol.tvlqr_default_options()
    o.noise_mode, o.noise_seed = 1, 20190530
    ids = np.array([4, 2 ** 33 + 1, 0], dtype=np.int64)
    tv = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, opts=o, noise_ids=ids)
    return dict(seed=77, n_knots=b.n_knots, X=r["X"], U=r["U"], Qd=Qd, Qfd=Qfd, Rd=Rd, x0_sim=x0s, noise_seed=20190530, noise_ids=ids,
                X_sim=tv["X_sim"], U_sim=tv["U_sim"], K=tv["K"], slew_index=tv["stats"]["slew_index"], slew_time=tv["stats"]["slew_time"],
                failed=tv["stats"]["failed"])


def mpc_case():
    b = ss.workload_monte_carlo(T=2, N=30, seed=41)
    B = ss.dipole_btable(120, 0.2, 6771.0, 96.6)
    b.Btab, b.n_tab = np.ascontiguousarray(B[None]), 120
    b.dtau[:] = 1.0
    o = solve_opts(max_outer=1, max_inner=3)
    r = ol.mpc_batch(b, o, 8, plant_integrator=4)
    return dict(seed=41, rows=120, n_steps=8, X_hist=r["X_hist"], U_hist=r["U_hist"], X=r["X"], U=r["U"], inner_iters=r["stats"]["inner_iters"])


if __name__ == "__main__":
    B, pos = ol.btable_batch(KEP, [0.0, 5.0, 0.0], [300.0, 400.0, 350.0], 40)
    idx, cond = ol.horizon_batch(B, 7.5, [40.0, 80.0, 1e9])
    np.savez_compressed(os.path.join(HERE, "stages_btable_horizon.npz"), kep=KEP, t0=[0.0, 5.0, 0.0], tf=[300.0, 400.0, 350.0], n_half=40,
                        B=B, pos=pos, dt_row=7.5, cutoff=[40.0, 80.0, 1e9], tf_index=idx, cond_at=cond)
    np.savez_compressed(os.path.join(HERE, "stages_tracking.npz"), **tracking_case())
    np.savez_compressed(os.path.join(HERE, "stages_mpc.npz"), **mpc_case())
    for f in ("stages_btable_horizon.npz", "stages_tracking.npz", "stages_mpc.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
    print("horizon indices", idx, cond)
