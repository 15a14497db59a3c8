"""Horizon selection: magnetic_gramian + condition_based_time (src/magnetic_toolbox.jl:1-31)."""
import numpy as np
import pytest

import refmath as rm


def _tables(pkg, T=5, n=700, seed=3):
    ss = pkg.slew_setup
    rng = np.random.default_rng(seed)
    return np.stack([ss.dipole_btable(n, 2400.0 / n, 6771.0, 96.6 - 15 * t, rng.random() * 360, rng.random() * 360) for t in range(T)])


def _reference(B, dt, cutoff):
    # literal NumPy transcription of src/magnetic_toolbox.jl:1-31 (np.linalg.cond = 2-norm condition number, as Julia's cond)
    G = rm.hat(B[0]) @ rm.hat(B[0]).T
    conds = [np.linalg.cond(G)]
    for i in range(1, B.shape[0]):
        G = G + rm.hat(B[i]) @ rm.hat(B[i]).T * dt
        conds.append(np.linalg.cond(G))
    conds = np.array(conds)
    hit = np.nonzero(conds < cutoff)[0]
    return (int(hit[0]) + 1 if len(hit) else 0), conds


def test_oracle_matches_reference_text(pkg, ol):
    B = _tables(pkg, T=3, n=500)
    for cutoff in (30.0, 50.0, 1.5):
        idx, cat, call = ol.horizon_batch(B, 4.8, cutoff, want_all=True)
        for t in range(3):
            ridx, conds = _reference(B[t], 4.8, cutoff)
            assert idx[t] == ridx
            fin = np.isfinite(conds) & (conds < 1e12)
            np.testing.assert_allclose(call[t][fin], conds[fin], rtol=1e-8)
    assert np.any(ol.horizon_batch(B, 4.8, 1.5)[0] == 0)        # a cutoff that is never reached


def test_emulated_kernel_matches_oracle(pkg, ol, emu):
    B = _tables(pkg, T=4, n=300)
    cut = np.array([30.0, 50.0, 100.0, 1.01])
    ri, rc = ol.horizon_batch(B, 8.0, cut)
    gi, gc = emu.horizon(B, 8.0, cut)
    assert np.array_equal(ri, gi)
    np.testing.assert_allclose(gc[ri > 0], rc[ri > 0], rtol=1e-9)


def test_knots_from_index(pkg):
    t_final, n = pkg.horizon.knots_from_index(np.array([1250, 0]), 5400.0, 5000)
    assert t_final[0] == pytest.approx(1350.0) and n[0] == 6750 and n[1] == 0


@pytest.mark.gpu
def test_gpu_horizon_then_ragged_solve(pkg, ol):
    """the reference's sequence: coarse table -> Gramian horizon -> per-run knot count -> solve (src/monte_carlo.jl:134-196)"""
    to, hz, ss = pkg.trajopt, pkg.horizon, pkg.slew_setup
    T, n_rows = 6, 5000
    B = _tables(pkg, T=T, n=n_rows, seed=9)
    s = to.AugmentedLagrangianSolver(None, None)
    idx, cond = hz.condition_based_time(s, B, 2400.0 / n_rows, 30.0)
    ridx, rcond = ol.horizon_batch(B, 2400.0 / n_rows, 30.0)
    assert np.array_equal(idx, ridx) and np.all(idx > 0)
    np.testing.assert_allclose(cond, rcond, rtol=1e-9)
    t_final, nk = hz.knots_from_index(idx, 2400.0, n_rows)
    nk = np.clip(nk, 2, 400).astype(np.int32)                       # keep the test small
    N = int(nk.max())
    b = ss.workload_monte_carlo(T=T, N=N, seed=19, random_orbit=True)
    b.n_knots = nk
    opts = to.AugmentedLagrangianSolverOptions()
    opts.iterations, opts.opts_uncon.iterations, opts.opts_uncon.dJ_counter_limit = 2, 4, 1
    s.opts = opts
    res = to.solve_(to.BatchProblem.from_arrays(b), s)
    from conftest import assert_same_solution, oracle_options
    assert_same_solution(ol.solve_batch(b, oracle_options(ol, max_outer=2, max_inner=4, dj_counter_limit=1), nthreads=4), res)
    s.close()
