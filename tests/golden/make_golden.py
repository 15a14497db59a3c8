#!/usr/bin/env python3
"""Generates the committed golden fixtures tests/golden/*.npz.

The reference (Julia; its solver is an un-vendored dependency) cannot run in this environment and has no test
vectors of its own, so these vectors are produced by this repository's CPU oracle
(oracle/liboracle.so) — "parity unpinned" — and serve as a regression pin for the oracle and as fixed
input/output pairs for the emulated and the real HIP kernel. Each file holds the complete inputs
(ABI arrays + options) and the oracle's outputs (X, U, K, stats).

    python tests/golden/make_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(HERE)]
import numpy as np  # noqa: E402
import oracle_lib as ol  # noqa: E402
from tsat_loader import load_package  # noqa: E402
import helpers  # noqa: E402

pkg = load_package()
ss = pkg.slew_setup


def opts(**kw):
    o = ol.default_options()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def neg_r(b):
    b.Rd[:] = -1.0e-4
    return b


CASES = {
    # Monte-Carlo flavour (src/monte_carlo.jl:107-198), short horizons
    "mc_n60_t4_rk3.npz": (lambda: ss.workload_monte_carlo(T=4, N=60), opts(max_outer=3, max_inner=6, dj_counter_limit=1)),
    "mc_n200_t2_rk3.npz": (lambda: ss.workload_monte_carlo(T=2, N=200, seed=7), opts(max_outer=5, max_inner=10, dj_counter_limit=1)),
    "mc_n80_t2_rk4.npz": (lambda: ss.workload_monte_carlo(T=2, N=80, seed=11), opts(max_outer=2, max_inner=5, integrator=4)),
    # random orbit per trajectory (one table each) and a ragged chunk size (N-1 = 96 + 1)
    "mc_orbit_n98_t3_rk3.npz": (lambda: ss.workload_monte_carlo(T=3, N=98, seed=3, random_orbit=True),
                                opts(max_outer=3, max_inner=5, dj_counter_limit=1, max_linesearch=12)),
    # single-slew flavour (src/TortoiseSat.jl:117-199): 1P inertia, |u| <= 1 active bounds, U0 = 0
    "single_n150_rk3.npz": (lambda: ss.workload_single_slew(N=150), opts(max_outer=4, max_inner=12)),
    # quaternion hooks of the old-API Monte-Carlo (src/monte_carlo.jl:158): error_state = 1
    "mc_es_n120_t3_rk3.npz": (lambda: ss.workload_monte_carlo(T=3, N=120, seed=13), opts(max_outer=4, max_inner=8, dj_counter_limit=1, error_state=1)),
    "mc_es_n70_t2_rk4.npz": (lambda: ss.workload_monte_carlo(T=2, N=70, seed=17, random_orbit=True), opts(max_outer=2, max_inner=5, integrator=4, error_state=1)),
    # negative control weight: Quu is indefinite, so the backward sweep has to restart with growing regularisation
    "negR_n70_t2_rk3.npz": (lambda: neg_r(ss.workload_monte_carlo(T=2, N=70, seed=5)), opts(max_outer=2, max_inner=4)),
}

if __name__ == "__main__":
    for name, (mk, o) in CASES.items():
        b = mk()
        res = ol.solve_batch(b, o)
        helpers.save_case(os.path.join(HERE, name), b, o, res)
        st = res["stats"]
        print(name, os.path.getsize(os.path.join(HERE, name)) // 1024, "KiB", "status", st["status"], "inner", st["inner_iters"],
              "ls", st["ls_trials"], "restarts", st["bp_restarts"], "fp_fails", st["fp_fails"], "cmax", st["c_max"])
