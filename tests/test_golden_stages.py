"""The committed fixtures of the stages around the solve (tests/golden/stages_*.npz, made by make_golden_stages.py):
oracle and emulated kernels on the CPU tier, the real kernels through the C ABI on the GPU tier."""
import numpy as np
import pytest

import helpers
from conftest import oracle_options


def _tracking_inputs(pkg, g):
    b = pkg.slew_setup.workload_monte_carlo(T=3, N=60, seed=int(g["seed"]))
    b.n_knots = g["n_knots"].astype(np.int32)
    return b


def _mpc_inputs(pkg, g):
    ss = pkg.slew_setup
    b = ss.workload_monte_carlo(T=2, N=30, seed=int(g["seed"]))
    B = ss.dipole_btable(int(g["rows"]), 0.2, 6771.0, 96.6)
    b.Btab, b.n_tab = np.ascontiguousarray(B[None]), int(g["rows"])
    b.dtau[:] = 1.0
    return b


def _check_tables(g, B, pos, idx, cond):
    assert np.max(np.abs(pos - g["pos"])) < 1e-8
    assert np.max(np.abs(B - g["B"])) < 1e-9 * np.max(np.abs(g["B"]))
    assert np.array_equal(idx, g["tf_index"])
    np.testing.assert_allclose(cond, g["cond_at"], rtol=1e-8)


def _check_tracking(g, tv):
    assert np.max(np.abs(tv["X_sim"] - g["X_sim"])) < 1e-9 and np.max(np.abs(tv["U_sim"] - g["U_sim"])) < 1e-8
    assert np.max(np.abs(tv["K"] - g["K"])) < 1e-8 * max(1.0, float(np.max(np.abs(g["K"]))))
    assert np.array_equal(tv["stats"]["slew_index"], g["slew_index"]) and np.array_equal(tv["stats"]["failed"], g["failed"])
    np.testing.assert_allclose(tv["stats"]["slew_time"], g["slew_time"], rtol=1e-14)


def _check_mpc(g, r):
    assert np.max(np.abs(r["X_hist"] - g["X_hist"])) < 1e-9 and np.max(np.abs(r["U_hist"] - g["U_hist"])) < 1e-8
    assert np.max(np.abs(r["X"] - g["X"])) < 1e-9 and np.max(np.abs(r["U"] - g["U"])) < 1e-8
    assert np.array_equal(r["stats"]["inner_iters"], g["inner_iters"])


@pytest.mark.parametrize("who", ["oracle", "emu"])
def test_stage_fixtures_on_cpu(pkg, ol, emu, who):
    g = helpers.stage_golden("stages_btable_horizon.npz")
    if who == "oracle":
        B, pos = ol.btable_batch(g["kep"], g["t0"], g["tf"], int(g["n_half"]))
        idx, cond = ol.horizon_batch(B, float(g["dt_row"]), g["cutoff"])
    else:
        B, pos = emu.btable(g["kep"], g["t0"], g["tf"], int(g["n_half"]))
        idx, cond = emu.horizon(B, float(g["dt_row"]), g["cutoff"])
    _check_tables(g, B, pos, idx, cond)

    g = helpers.stage_golden("stages_tracking.npz")
    b = _tracking_inputs(pkg, g)
    if who == "oracle":
        o = ol.tvlqr_default_options(); o.noise_mode, o.noise_seed = 1, int(g["noise_seed"])
        tv = ol.tvlqr_batch(b, g["X"], g["U"], g["Qd"], g["Qfd"], g["Rd"], g["x0_sim"], opts=o, noise_ids=g["noise_ids"])
    else:
        o = pkg._abi.TvlqrOptions.from_buffer_copy(ol.tvlqr_default_options()); o.noise_mode, o.noise_seed = 1, int(g["noise_seed"])
        tv = emu.tvlqr(b, g["X"], g["U"], g["Qd"], g["Qfd"], g["Rd"], g["x0_sim"], opts=o, noise_ids=g["noise_ids"])
    _check_tracking(g, tv)

    g = helpers.stage_golden("stages_mpc.npz")
    b = _mpc_inputs(pkg, g)
    o = oracle_options(ol, max_outer=1, max_inner=3, dj_counter_limit=1)
    _check_mpc(g, (ol.mpc_batch if who == "oracle" else emu.mpc)(b, o, int(g["n_steps"]), plant_integrator=4))


@pytest.mark.gpu
def test_stage_fixtures_on_gpu(pkg):
    """no oracle in the loop: the C ABI against the committed vectors"""
    to, mg, hz, tr, mpc = pkg.trajopt, pkg.magnetic, pkg.horizon, pkg.tracking, pkg.mpc
    s = to.AugmentedLagrangianSolver(None, to.AugmentedLagrangianSolverOptions())
    g = helpers.stage_golden("stages_btable_horizon.npz")
    B, pos = mg.magnetic_simulation(s, g["kep"], g["t0"], g["tf"], int(g["n_half"]))
    idx, cond = hz.condition_based_time(s, B, float(g["dt_row"]), g["cutoff"])
    _check_tables(g, B, pos, idx, cond)
    g = helpers.stage_golden("stages_tracking.npz")
    tv = tr.attitude_simulation(s, _tracking_inputs(pkg, g), g["X"], g["U"], g["x0_sim"], g["Qd"], g["Qfd"], g["Rd"],
                                noise_seed=int(g["noise_seed"]), noise_ids=g["noise_ids"])
    _check_tracking(g, tv)
    g = helpers.stage_golden("stages_mpc.npz")
    s.opts.opts_uncon.dJ_counter_limit = 1
    r = mpc.receding_horizon(to.BatchProblem.from_arrays(_mpc_inputs(pkg, g)), s, int(g["n_steps"]), plant_integrator=4)
    r.update(s.download(want_K=False))
    _check_mpc(g, r)
    s.close()
