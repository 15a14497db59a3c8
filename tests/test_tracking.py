"""Closed-loop TVLQR tracking: oracle pinned against a NumPy transcription of the reference text,
emulated kernel and (gpu tier) the real kernel against the oracle."""
import numpy as np
import pytest

import refmath as rm
from conftest import oracle_options


def _solved(pkg, ol, T=2, N=60, seed=61, budget=(3, 6)):
    b = pkg.slew_setup.workload_monte_carlo(T=T, N=N, seed=seed)
    r = ol.solve_batch(b, oracle_options(ol, max_outer=budget[0], max_inner=budget[1], dj_counter_limit=1))
    return b, r


def _setup(pkg, b, seed=5, noisy=True):
    tr = pkg.tracking
    rng = np.random.default_rng(seed)
    Qd, Qfd, Rd = tr.tvlqr_weights(b.T, r=0.5e3)
    x0s = tr.perturbed_initial_state(b.x0, rng)
    nz = tr.simulator_noise(b.T, b.N, rng) if noisy else None
    return Qd, Qfd, Rd, x0s, nz


def test_oracle_tracking_matches_reference_text(pkg, ol):
    """gains: NumPy restatement of attitude_lqr (src/attitude_controller.jl:50-119) with central-difference Jacobians of
    the rk4 map over dt^2; closed loop: src/attitude_controller.jl:39-45 with the noise-free plant"""
    b, r = _solved(pkg, ol, T=1, N=40)
    Qd, Qfd, Rd, x0s, _ = _setup(pkg, b, noisy=False)
    tv = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s)
    X, U, N, dt = r["X"][0], r["U"][0], b.N, 0.2
    J = np.diag([0.00125] * 3)
    Bt = b.Btab[0]

    def f(x, u, row):      # gain_simulator on the 7-state with the row already looked up
        return rm.attitude_dynamics(x, u / 100.0, rm.qrot(x[3:7] / np.linalg.norm(x[3:7]), Bt[row]), J)

    def rk4_lin(x, u, k):  # augmented rk4 with dt = S[end]^2 (src/attitude_controller.jl:134-145); rows stay at knot k
        h = dt * dt
        k1 = f(x, u, k) * h; k2 = f(x + k1 / 2, u, k) * h; k3 = f(x + k2 / 2, u, k) * h; k4 = f(x + k3, u, k) * h
        return x + (k1 + 2 * k2 + 2 * k3 + k4) / 6

    A = np.zeros((N - 1, 6, 6)); B = np.zeros((N - 1, 6, 3))
    for k in range(N - 1):
        Aq = np.zeros((7, 7)); Bq = np.zeros((7, 3))
        for j in range(7):
            e = np.zeros(7); e[j] = 1e-6
            Aq[:, j] = (rk4_lin(X[k] + e, U[k], k) - rk4_lin(X[k] - e, U[k], k)) / 2e-6
        for j in range(3):
            e = np.zeros(3); e[j] = 1e-3
            Bq[:, j] = (rk4_lin(X[k], U[k] + e, k) - rk4_lin(X[k], U[k] - e, k)) / 2e-3
        A[k], B[k] = rm.reduce_error_state(Aq, Bq, X[k, 3:7], X[k + 1, 3:7])
    K = rm.tvlqr_riccati(A, B, np.diag(Qd[0]), np.diag(Rd[0]), np.diag(Qfd[0]))
    np.testing.assert_allclose(tv["K"][0].transpose(0, 2, 1), K, rtol=2e-5, atol=1e-8)

    xs = x0s[0].copy()
    for k in range(N - 1):
        dX = np.r_[xs[:3] - X[k, :3], rm.qmult(rm.q_inv(X[k, 3:7]), xs[3:7])[1:]]
        us = U[k] - tv["K"][0, k].T @ dX
        np.testing.assert_allclose(tv["U_sim"][0, k], us, atol=1e-12)
        g = lambda x, kk: f(x, us, kk)
        k1 = g(xs, k) * dt; k2 = g(xs + k1 / 2, k) * dt; k3 = g(xs + k2 / 2, k) * dt; k4 = g(xs + k3, min(k + 1, N - 1)) * dt
        xs = xs + (k1 + 2 * k2 + 2 * k3 + k4) / 6
        np.testing.assert_allclose(tv["X_sim"][0, k + 1], xs, atol=1e-12)


def test_oracle_tracking_statistic(pkg, ol):
    """src/monte_carlo.jl:242-262 recomputed from X_sim"""
    b, r = _solved(pkg, ol, T=3, N=400, budget=(5, 10))
    Qd, Qfd, Rd, x0s, nz = _setup(pkg, b)
    tv = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz)
    for t in range(b.T):
        idx = 0
        for j in range(1, b.N + 1):
            xs = tv["X_sim"][t, j - 1]
            ang = 2 * np.arccos(min(rm.qmult(rm.q_inv(b.xf[t, 3:]), xs[3:7])[0], 1.0))
            if j > 10 and np.linalg.norm(xs[:3]) < 0.05 and ang < 0.08727:
                idx = j
                break
        st = tv["stats"][t]
        assert st["slew_index"] == idx and st["failed"] == (idx == 0)
        assert st["slew_time"] == pytest.approx(0.2 * (idx if idx else b.N))
    assert np.max(np.abs(tv["X_sim"] - r["X"])) < 0.05          # the loop tracks the plan


def test_statistic_as_the_reference_line_reads(pkg, ol, emu):
    """rate_as_written = 1: `omega_norm_vec[j] = norm(sim_states[i][1:3,i])` (src/monte_carlo.jl:247) — for every sample j the
    rate of sample i, the trial's 1-based number; recomputed from X_sim as the Julia loop does it, for trial numbers from the
    position in the batch and from explicit ids (one past the end of the trajectory: clamped where Julia raises). Limits chosen
    so that the slews of this short batch do arrive under the default reading."""
    b, r = _solved(pkg, ol, T=3, N=400, budget=(5, 10))
    Qd, Qfd, Rd, x0s, nz = _setup(pkg, b)
    base = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz)
    rates = np.linalg.norm(base["X_sim"][:, :, :3], axis=2)
    w_tol, ang_tol = 1.5 * float(rates[:, 200:].min(axis=1).max()), 3.2
    o0 = ol.tvlqr_default_options(); o0.w_tol, o0.angle_tol = w_tol, ang_tol
    plain = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz, opts=o0)
    assert np.all(plain["stats"]["slew_index"] > 10)
    o = ol.tvlqr_default_options(); o.w_tol, o.angle_tol, o.rate_as_written = w_tol, ang_tol, 1
    differs = 0
    for ids in (None, np.array([350, 5, 4000], dtype=np.int64)):
        tv = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz, opts=o, noise_ids=ids)
        assert np.array_equal(tv["X_sim"], plain["X_sim"])
        for t in range(b.T):
            i = (t if ids is None else int(ids[t])) + 1                       # Julia's i
            w_i = np.linalg.norm(tv["X_sim"][t, min(i, b.N) - 1, :3])         # sim_states[i][1:3,i]
            idx = 0
            for j in range(1, b.N + 1):
                xs = tv["X_sim"][t, j - 1]
                ang = 2 * np.arccos(min(rm.qmult(rm.q_inv(b.xf[t, 3:]), xs[3:7])[0], 1.0))
                if j > 10 and w_i < w_tol and ang < ang_tol:
                    idx = j
                    break
            st = tv["stats"][t]
            assert st["slew_index"] == idx and st["failed"] == (idx == 0), (t, ids)
            assert st["final_w_norm"] == plain["stats"]["final_w_norm"][t]    # (the last sample's own rate either way)
            differs += int(idx != plain["stats"]["slew_index"][t])
        got = emu.tvlqr(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz, opts=o, noise_ids=ids)
        _same_tracking(tv, got)
    assert differs > 0          # the two readings are not the same statistic


@pytest.mark.parametrize("noisy", [False, True])
def test_emulated_tracking_kernel_matches_oracle(pkg, ol, emu, noisy):
    b, r = _solved(pkg, ol, T=2, N=60)
    Qd, Qfd, Rd, x0s, nz = _setup(pkg, b, noisy=noisy)
    ref = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz)
    got = emu.tvlqr(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz)
    _same_tracking(ref, got)


def _ragged(pkg, ol, T=3, N=60, nk=(60, 37, 12)):
    b = pkg.slew_setup.workload_monte_carlo(T=T, N=N, seed=77)
    b.n_knots = np.array(nk, dtype=np.int32)
    r = ol.solve_batch(b, oracle_options(ol, max_outer=3, max_inner=6, dj_counter_limit=1))
    return b, r


def test_ragged_tracking_equals_separate_runs(pkg, ol, emu):
    """variable horizons (t_total[i], src/monte_carlo.jl:145): each trajectory is tracked over its own knots — same
    numbers as tracking it alone in a batch of its own length — and its slabs are zero beyond"""
    b, r = _ragged(pkg, ol)
    Qd, Qfd, Rd, x0s, nz = _setup(pkg, b)
    ref = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz)
    got = emu.tvlqr(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz)
    _same_tracking(ref, got)
    for t, n in enumerate(b.n_knots):
        assert np.all(ref["X_sim"][t, n:] == 0) and np.all(ref["U_sim"][t, n - 1:] == 0) and np.all(got["X_sim"][t, n:] == 0)
        assert np.all(got["K"][t, n - 1:] == 0)
        one = b.slice(t, t + 1)
        one.N, one.n_knots = int(n), None
        one.U0 = np.ascontiguousarray(one.U0[:, :n - 1])
        alone = ol.tvlqr_batch(one, r["X"][t:t + 1, :n], r["U"][t:t + 1, :n - 1], Qd[t:t + 1], Qfd[t:t + 1], Rd[t:t + 1],
                               x0s[t:t + 1], noise=nz[t:t + 1, :n - 1])
        assert np.array_equal(alone["X_sim"][0], ref["X_sim"][t, :n])
        assert alone["stats"]["slew_index"][0] == ref["stats"]["slew_index"][t]
        assert alone["stats"]["slew_time"][0] == ref["stats"]["slew_time"][t]


def test_philox_known_answers_and_draw_layout(pkg, ol):
    """Philox4x32-10 known-answer vectors of the Random123 distribution (kat_vectors), on the oracle's C++ and on the
    NumPy generator of the host module; then the draw layout of include/tortoise_hip.h on both"""
    tr = pkg.tracking
    kats = [((0, 0), (0, 0, 0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff, 0xffffffff), (0xffffffff,) * 4, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0xa4093822, 0x299f31d0), (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for key, ctr, out in kats:
        assert tuple(ol.philox4x32_10(key, ctr)) == out
        got = tr.philox4x32_10(np.array(key, dtype=np.uint64), np.array([ctr], dtype=np.uint64))[0]
        assert tuple(int(v) for v in got) == out
    seed, ids = 0x1234567890abcdef, np.array([0, 7, 2 ** 40 + 3])
    nz = tr.generated_noise(seed, ids, 6)
    assert nz.shape == (3, 5, 4, 9)
    for t, k, st in ((0, 0, 0), (1, 4, 3), (2, 2, 1)):
        np.testing.assert_allclose(nz[t, k, st], ol.plant_noise(seed, int(ids[t]), k, st, (0.38 * np.pi / 180) ** 2, (np.pi / 180) ** 2, 1e-10),
                                   rtol=1e-13, atol=0)
    big = tr.generated_noise(99, np.arange(64), 200)                       # 460k normals, 154k uniforms
    g = big[..., :6] / np.r_[[(0.38 * np.pi / 180) ** 2] * 3, [(np.pi / 180) ** 2] * 3]
    assert abs(g.mean()) < 5e-3 and abs(g.std() - 1) < 5e-3 and abs(np.mean(g ** 4) - 3) < 0.05
    f = big[..., 6:] / 1e-10
    assert f.min() > 0 and f.max() < 1 and abs(f.mean() - 0.5) < 3e-3
    assert abs(np.corrcoef(g[..., 0].ravel(), g[..., 1].ravel())[0, 1]) < 5e-3


def test_seeded_noise_equals_the_same_noise_as_an_array(pkg, ol, emu):
    """noise_mode = 1 (drawn where it is used) against array mode fed with the host-side generator: same run"""
    b, r = _ragged(pkg, ol)
    Qd, Qfd, Rd, x0s, _ = _setup(pkg, b)
    ids = np.array([5, 900, 2 ** 33], dtype=np.int64)
    nz = pkg.tracking.generated_noise(2019, ids, b.N)
    arr = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz)
    o = ol.tvlqr_default_options(); o.noise_mode, o.noise_seed = 1, 2019
    ref = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, opts=o, noise_ids=ids)
    assert np.max(np.abs(arr["X_sim"] - ref["X_sim"])) < 1e-13
    assert np.array_equal(arr["stats"]["slew_index"], ref["stats"]["slew_index"])
    o2 = pkg._abi.TvlqrOptions.from_buffer_copy(o)
    got = emu.tvlqr(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, opts=o2, noise_ids=ids)
    _same_tracking(ref, got)
    # ids default to the trajectory index; another seed is another run
    o.noise_seed = 2020
    other = ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, opts=o)
    assert np.max(np.abs(other["X_sim"] - ref["X_sim"])) > 1e-9


def _same_tracking(ref, got):
    kscale = max(float(np.max(np.abs(ref["K"]))), 1.0)
    assert np.max(np.abs(ref["K"] - got["K"])) < 1e-8 * kscale
    assert np.max(np.abs(ref["X_sim"] - got["X_sim"])) < 1e-9
    assert np.max(np.abs(ref["U_sim"] - got["U_sim"])) < 1e-8
    assert np.array_equal(ref["stats"]["slew_index"], got["stats"]["slew_index"])
    assert np.array_equal(ref["stats"]["failed"], got["stats"]["failed"])
    np.testing.assert_allclose(got["stats"]["final_angle"], ref["stats"]["final_angle"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(got["stats"]["final_w_norm"], ref["stats"]["final_w_norm"], rtol=1e-6, atol=1e-12)


@pytest.mark.gpu
def test_gpu_tracking_matches_oracle(pkg, ol):
    """solve on the GPU, track on the GPU (1000 knots, noise injected), compare both stages with the oracle"""
    to, tr = pkg.trajopt, pkg.tracking
    b = pkg.slew_setup.workload_monte_carlo(T=12, N=1000)
    opts = to.AugmentedLagrangianSolverOptions()
    opts.iterations, opts.opts_uncon.iterations, opts.opts_uncon.dJ_counter_limit = 5, 10, 1
    s = to.AugmentedLagrangianSolver(None, opts)
    res = to.solve_(to.BatchProblem.from_arrays(b), s)
    for noisy in (False, True):
        Qd, Qfd, Rd, x0s, nz = _setup(pkg, b, noisy=noisy)
        got = tr.attitude_simulation(s, b, res["X"], res["U"], x0s, Qd, Qfd, Rd, noise=nz)
        ref = ol.tvlqr_batch(b, res["X"], res["U"], Qd, Qfd, Rd, x0s, noise=nz, nthreads=min(12, ol.num_procs()))
        _same_tracking(ref, got)
    res_dev = tr.attitude_simulation(s, b, None, None, x0s, Qd, Qfd, Rd, noise=nz)      # the resident batch, nothing re-uploaded
    for k in ("X_sim", "U_sim", "K"):
        assert np.array_equal(res_dev[k], got[k]), k
    assert np.array_equal(res_dev["stats"], got["stats"])
    got2 = tr.attitude_simulation(s, b, res["X"], res["U"], x0s, Qd, Qfd, Rd, noise=nz, linearize_dt_sq=False)
    o = ol.tvlqr_default_options(); o.linearize_dt_sq = 0
    _same_tracking(ol.tvlqr_batch(b, res["X"], res["U"], Qd, Qfd, Rd, x0s, noise=nz, opts=o, nthreads=8), got2)
    assert np.max(np.abs(got2["K"] - got["K"])) > 1e-3          # the dt^2 quirk changes the gains
    # ragged batch through the C ABI
    b, r = _ragged(pkg, ol, T=5, N=300, nk=(300, 211, 64, 65, 2))
    Qd, Qfd, Rd, x0s, nz = _setup(pkg, b)
    got = tr.attitude_simulation(s, b, r["X"], r["U"], x0s, Qd, Qfd, Rd, noise=nz)
    _same_tracking(ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz, nthreads=5), got)
    for t, n in enumerate(b.n_knots):
        assert np.all(got["X_sim"][t, n:] == 0) and np.all(got["U_sim"][t, n - 1:] == 0)
    # noise drawn inside the kernel (Philox) = the same noise handed over as an array = the oracle's
    ids = np.array([3, 1, 2 ** 35, 0, 77], dtype=np.int64)
    gen = tr.attitude_simulation(s, b, r["X"], r["U"], x0s, Qd, Qfd, Rd, noise_seed=424242, noise_ids=ids)
    arr = tr.attitude_simulation(s, b, r["X"], r["U"], x0s, Qd, Qfd, Rd, noise=tr.generated_noise(424242, ids, b.N))
    assert np.max(np.abs(gen["X_sim"] - arr["X_sim"])) < 1e-12 and np.array_equal(gen["stats"]["slew_index"], arr["stats"]["slew_index"])
    o = ol.tvlqr_default_options(); o.noise_mode, o.noise_seed = 1, 424242
    _same_tracking(ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, opts=o, noise_ids=ids, nthreads=5), gen)
    # the statistic as the reference's line reads (rate of sample i = trial number, src/monte_carlo.jl:247)
    o = ol.tvlqr_default_options(); o.rate_as_written = 1
    for tid in (None, np.array([250, 1, 2, 9000, 0], dtype=np.int64)):
        got = tr.attitude_simulation(s, b, r["X"], r["U"], x0s, Qd, Qfd, Rd, noise=nz, rate_as_written=True, trial_ids=tid)
        _same_tracking(ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz, opts=o, noise_ids=tid, nthreads=5), got)
    # ... with the trial numbers given as noise_ids next to a noise ARRAY (a sharded Monte-Carlo that brings its own noise): they are
    # the trial numbers all the same — not silently dropped for the position in the batch
    tid = np.array([250, 1, 2, 9000, 0], dtype=np.int64)
    got = tr.attitude_simulation(s, b, r["X"], r["U"], x0s, Qd, Qfd, Rd, noise=nz, rate_as_written=True, noise_ids=tid)
    _same_tracking(ol.tvlqr_batch(b, r["X"], r["U"], Qd, Qfd, Rd, x0s, noise=nz, opts=o, noise_ids=tid, nthreads=5), got)
    with pytest.raises(ValueError):      # ids that nothing would use
        tr.attitude_simulation(s, b, r["X"], r["U"], x0s, Qd, Qfd, Rd, noise=nz, noise_ids=tid)
    with pytest.raises(RuntimeError):
        tr.attitude_simulation(s, b, r["X"], r["U"], x0s, Qd, Qfd, Rd, noise=nz, noise_seed=None if False else None, min_steps=-1)
    # the switch was a reserved word of the options struct before version 300: garbage in it is rejected, not read as "on"
    import ctypes as C
    ob = pkg._abi.TvlqrOptions()
    pkg._abi.load().tsat_tvlqr_default_options(C.byref(ob))
    ob.n_knots, ob.n_tab, ob.rate_as_written = b.N, b.n_tab, 7
    st = np.zeros(b.T, dtype=pkg._abi.TVLQR_STATS_DTYPE)
    d = pkg._abi.as_dp
    rc = pkg._abi.load().tsat_tvlqr_batch(s._h, C.byref(ob), b.T, b.Btab.shape[0], d(r["X"]), d(r["U"]), d(b.xf), d(b.Btab), pkg._abi.as_ip(b.btab_idx),
                                         d(b.tau0), d(b.dtau), d(b.dt), d(b.Jmat), d(Qd), d(Qfd), d(Rd), d(x0s), None, None, None, None,
                                         st.ctypes.data_as(C.c_void_p), pkg._abi.as_ip(np.ascontiguousarray(b.n_knots, dtype=np.int32)), None)
    assert rc != 0 and b"rate_as_written" in pkg._abi.load().tsat_last_error(s._h)
    s.close()
