"""Receding-horizon re-solve (BASELINE.json configs[4]; SURVEY §8d config 5 — defined here, not in the reference):
the oracle loop against its own single solves, the emulated kernels and (gpu tier) the resident GPU loop."""
import numpy as np
import pytest

import refmath as rm
from conftest import oracle_options


def _batch(pkg, T=2, N=20, seed=41, rows=200):
    ss = pkg.slew_setup
    b = ss.workload_monte_carlo(T=T, N=N, seed=seed)
    # a table long enough for the simulation: one row per knot, the clock advances one row per control step
    B = ss.dipole_btable(rows, 0.2, 6771.0, 96.6)
    b.Btab, b.n_tab = np.ascontiguousarray(B[None]), rows
    b.dtau[:] = 1.0
    return b


def _opts(ol, **kw):
    o = oracle_options(ol, max_outer=1, max_inner=3, dj_counter_limit=1)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def test_loop_is_solve_plant_shift(pkg, ol):
    """three steps of the oracle loop rebuilt by hand from single oracle solves"""
    b = _batch(pkg)
    o = _opts(ol)
    got = ol.mpc_batch(b, o, 3, plant_integrator=4)
    J = np.diag([0.00125] * 3)
    x, U0, tau = b.x0.copy(), b.U0.copy(), b.tau0.copy()
    for s in range(3):
        b2 = b.slice(0, b.T)
        b2.x0, b2.U0, b2.tau0 = np.ascontiguousarray(x), np.ascontiguousarray(U0), np.ascontiguousarray(tau)
        r = ol.solve_batch(b2, o, want_K=False)
        np.testing.assert_array_equal(got["X_hist"][:, s], x)
        np.testing.assert_array_equal(got["U_hist"][:, s], r["U"][:, 0])
        for t in range(b.T):
            row = lambda c: b.Btab[0][min(int(np.floor(tau[t] + c * b.dtau[t])), b.n_tab - 1)]
            x[t] = ol.rk_step(4, x[t], r["U"][t, 0], row(0.0), row(0.5), row(1.0), 0.2, J)
        U0 = np.concatenate([r["U"][:, 1:], r["U"][:, -1:]], axis=1)
        tau = tau + b.dtau
    np.testing.assert_array_equal(got["X_hist"][:, 3], x)
    np.testing.assert_array_equal(got["U"], r["U"])
    # the plant really is the reference's rk4 of the reference's dynamics (src/attitude_controller.jl:122-132)
    f = lambda xx, row: rm.attitude_dynamics(xx, got["U_hist"][0, 0] / 100.0, rm.qrot(xx[3:7] / np.linalg.norm(xx[3:7]), row), J)
    x0, r0, r1 = b.x0[0], b.Btab[0][0], b.Btab[0][1]
    k1 = f(x0, r0) * 0.2; k2 = f(x0 + k1 / 2, r0) * 0.2; k3 = f(x0 + k2 / 2, r0) * 0.2; k4 = f(x0 + k3, r1) * 0.2
    np.testing.assert_allclose(got["X_hist"][0, 1], x0 + (k1 + 2 * k2 + 2 * k3 + k4) / 6, atol=1e-14)


def test_closed_loop_turns_towards_the_goal(pkg, ol):
    """30 s of closed loop with a 20 s horizon: the distance the objective penalises, |q - qf| (component-wise, no
    double-cover handling — SURVEY quirk 6), shrinks for every trajectory"""
    b = _batch(pkg, T=3, N=100, seed=7, rows=400)
    got = ol.mpc_batch(b, _opts(ol, terminal_mask=0), 150, plant_integrator=4, nthreads=3)
    for t in range(3):
        d = np.linalg.norm(got["X_hist"][t, :, 3:7] - b.xf[t, 3:7], axis=1)
        assert d[-1] < d[75] < d[0], (d[0], d[75], d[-1])
    assert np.all(np.isfinite(got["U_hist"]))


@pytest.mark.parametrize("plant,ragged", [(4, False), (3, True)])
def test_emulated_loop_matches_oracle(pkg, ol, emu, plant, ragged):
    b = _batch(pkg, T=2, N=20)
    if ragged:
        b.n_knots = np.array([20, 13], dtype=np.int32)
    o = _opts(ol)
    ref = ol.mpc_batch(b, o, 4, plant_integrator=plant)
    got = emu.mpc(b, o, 4, plant_integrator=plant)
    _same(ref, got)


def _same(ref, got, plan=True):
    assert np.max(np.abs(ref["X_hist"] - got["X_hist"])) < 1e-9
    assert np.max(np.abs(ref["U_hist"] - got["U_hist"])) < 1e-8
    for k in ("inner_iters", "ls_trials", "status"):
        assert np.array_equal(ref["stats"][k], got["stats"][k]), k
    if plan:
        assert np.max(np.abs(ref["X"] - got["X"])) < 1e-9 and np.max(np.abs(ref["U"] - got["U"])) < 1e-8


@pytest.mark.gpu
def test_gpu_receding_horizon_matches_oracle(pkg, ol):
    """configs[4] shape: 200-knot horizon re-solved every control step (1 x 3 budget), rk4 plant"""
    to, mpc = pkg.trajopt, pkg.mpc
    b = _batch(pkg, T=8, N=200, seed=3, rows=400)
    s = to.AugmentedLagrangianSolver(None, to.AugmentedLagrangianSolverOptions())
    s.opts.opts_uncon.dJ_counter_limit = 1
    prob = to.BatchProblem.from_arrays(b)
    got = mpc.receding_horizon(prob, s, 30, plant_integrator=4)
    got.update(s.download(want_K=False))
    ref = ol.mpc_batch(b, _opts(ol), 30, plant_integrator=4, nthreads=8)
    _same(ref, got)
    # continuing the resident simulation = one longer call
    lib, abi = pkg._abi.load(), pkg._abi
    import ctypes as C
    o = s.opts.to_abi(b.N, b.n_tab, 3)
    o.max_outer, o.max_inner = 1, 3
    Xh = np.empty((8, 11, 7)); Uh = np.empty((8, 10, 3))
    assert lib.tsat_mpc_run(s._h, C.byref(o), 10, 4, abi.as_dp(Xh), abi.as_dp(Uh), None, None) == 0
    ref2 = ol.mpc_batch(b, _opts(ol), 40, plant_integrator=4, nthreads=8)
    assert np.max(np.abs(Xh - ref2["X_hist"][:, 30:])) < 1e-9 and np.max(np.abs(Uh - ref2["U_hist"][:, 30:])) < 1e-8
    np.testing.assert_array_equal(Xh[:, 0], got["X_hist"][:, -1])
    # ragged horizons + quaternion hooks + rk3 plant
    b.n_knots = np.array([200, 150, 64, 65, 200, 2, 31, 199], dtype=np.int32)
    got = mpc.receding_horizon(to.BatchProblem.from_arrays(b, error_state=1), s, 12, plant_integrator=3)
    got.update(s.download(want_K=False))
    _same(ol.mpc_batch(b, _opts(ol, error_state=1), 12, plant_integrator=3, nthreads=8), got)
    # tracking the last plan of the resident loop: tsat_mpc_run advanced x0 / tau0 on the device, and the handle's host
    # mirrors follow (they feed tsat_tvlqr_resident), so resident tracking equals the array entry point given the plan
    # and the ADVANCED table clock
    b = _batch(pkg, T=4, N=60, seed=9, rows=300)
    got = mpc.receding_horizon(to.BatchProblem.from_arrays(b), s, 25, plant_integrator=4)
    plan = s.download(want_K=False)
    Qd, Qfd, Rd = pkg.tracking.tvlqr_weights(b.T)
    x0s = np.ascontiguousarray(got["X_hist"][:, -1])
    res = pkg.tracking.attitude_simulation(s, b, None, None, x0s, Qd, Qfd, Rd)
    adv = b.slice(0, b.T)
    adv.tau0 = np.ascontiguousarray(b.tau0 + 25 * b.dtau)
    arr = pkg.tracking.attitude_simulation(s, adv, plan["X"], plan["U"], x0s, Qd, Qfd, Rd)
    assert np.array_equal(res["X_sim"], arr["X_sim"]) and np.array_equal(res["K"], arr["K"])
    stale = pkg.tracking.attitude_simulation(s, b, plan["X"], plan["U"], x0s, Qd, Qfd, Rd)      # the un-advanced clock differs
    assert np.max(np.abs(stale["K"] - arr["K"])) > 0
    # bad arguments are codes
    assert lib.tsat_mpc_run(s._h, C.byref(o), 0, 4, abi.as_dp(Xh), abi.as_dp(Uh), None, None) < 0
    assert lib.tsat_mpc_run(s._h, C.byref(o), 10, 5, abi.as_dp(Xh), abi.as_dp(Uh), None, None) < 0
    s.close()
