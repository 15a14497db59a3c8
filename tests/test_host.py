"""CPU tier: host-side logic above the C ABI — problem setup mirrored from the reference scripts and the
TrajectoryOptimization-style surface."""
import numpy as np
import pytest

import refmath as rm


def test_eigen_axis_slew_matches_reference_text(pkg):
    ss = pkg.slew_setup
    rng = np.random.default_rng(3)
    for _ in range(5):
        q0 = ss.random_unit_quats(rng, 1)[0]
        x0 = np.r_[0, 0, 0, q0]
        xf = np.r_[0, 0, 0, np.sqrt(2) / 2, np.sqrt(2) / 2, 0, 0]
        t = 0.2 * np.arange(101)
        w, q = ss.eigen_axis_slew(x0, xf, t)
        wr, qr = rm.eigen_axis_slew(x0, xf, t)
        np.testing.assert_allclose(w, wr, atol=1e-14)
        np.testing.assert_allclose(q, qr, atol=1e-14)
        np.testing.assert_allclose(q[0], q0, atol=1e-14)
        assert np.allclose(np.linalg.norm(q, axis=1), 1.0)


def test_bryson_weights_follow_tortoisesat_lines_157_168(pkg):
    ss = pkg.slew_setup
    b = ss.workload_single_slew(N=100)
    t = 0.2 * np.arange(101)
    w, _ = rm.eigen_axis_slew(b.x0[0], b.xf[0], t)
    J = ss.INERTIA["1P"]
    w_max = np.max(np.abs(w))
    tau_max = np.max(J @ np.diff(w.T, axis=1) / 0.2)
    m_max = tau_max / 1e-5 * 1e2
    np.testing.assert_allclose(b.Qd[0], np.r_[np.full(3, 10 / w_max**2), np.full(4, 1e4)], rtol=1e-12)
    np.testing.assert_allclose(b.Qfd[0], 10 * b.Qd[0])
    np.testing.assert_allclose(b.Rd[0], np.full(3, 1 / m_max**2), rtol=1e-12)
    # q0 = 90 deg about [1,0,1]/sqrt2 (src/TortoiseSat.jl:121-123)
    np.testing.assert_allclose(b.x0[0, 3:], [np.cos(np.pi / 4), 0.5, 0, 0.5], atol=1e-15)
    assert np.all(b.uhi == 1) and np.all(b.ulo == -1) and np.all(b.U0 == 0)


def test_workload_monte_carlo_is_deterministic_and_config2_shaped(pkg):
    ss = pkg.slew_setup
    a, b = ss.workload_monte_carlo(T=16, N=50), ss.workload_monte_carlo(T=16, N=50)
    for f in ("x0", "U0", "Qd", "Rd", "Btab"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
    assert a.Btab.shape == (1, 50, 3) and np.all(a.btab_idx == 0)           # one orbit
    assert np.allclose(np.linalg.norm(a.x0[:, 3:], axis=1), 1.0)
    assert np.all(a.uhi == 19) and a.meta["max_outer"] == 5 and a.meta["max_inner"] == 10
    assert 1e-5 < np.linalg.norm(a.Btab[0], axis=1).min() and np.linalg.norm(a.Btab[0], axis=1).max() < 7e-5
    c = ss.workload_monte_carlo(T=4, N=50, random_orbit=True)
    assert c.Btab.shape == (4, 50, 3) and np.array_equal(c.btab_idx, np.arange(4))
    s = a.slice(4, 9)
    assert s.T == 5 and np.array_equal(s.x0, a.x0[4:9])


def test_trajopt_surface_builds_the_same_batch_as_arrays(pkg):
    """Problem/Model/LQRObjective/Constraints used exactly as at src/TortoiseSat.jl:145-191."""
    ss, to = pkg.slew_setup, pkg.trajopt
    N, dt = 60, 0.2
    ref = ss.workload_single_slew(N=N)
    B = ref.Btab[0]
    J = ss.INERTIA["1P"]
    n, m = 8, 3
    x0 = np.r_[ref.x0[0], 0.0]
    xf = np.r_[ref.xf[0], 1.0]
    model = to.Model(to.DerivFunction(J, B), n, m)
    model_d = to.rk3(model)
    Q = np.zeros((n, n)); Qf = np.zeros((n, n))
    Q[:7, :7] = np.diag(ref.Qd[0]); Qf[:7, :7] = np.diag(ref.Qfd[0])
    R = np.diag(ref.Rd[0])
    obj = to.LQRObjective(Q, R, Qf, xf, N)
    bnd = to.BoundConstraint(n, m, u_max=1, u_min=-1)
    goal = to.goal_constraint(xf)
    constraints = to.Constraints(N)
    for k in range(1, N):
        constraints[k] += bnd
    constraints[N] += goal
    sat = to.Problem(model_d, obj, constraints=constraints, x0=x0, xf=xf, N=N, dt=dt)
    to.initial_controls_(sat, np.zeros((3, N + 1)))
    bp = to.BatchProblem([sat])
    for f in ("x0", "xf", "Qd", "Qfd", "Rd", "ulo", "uhi", "U0", "Jmat", "dt", "tau0", "dtau"):
        np.testing.assert_allclose(getattr(bp.arrays, f), getattr(ref, f), rtol=0, atol=0, err_msg=f)
    assert bp.integrator == 3 and bp.terminal_mask == 0x7F
    opts = to.AugmentedLagrangianSolverOptions()
    opts.opts_uncon.iterations = 50
    opts.iterations = 20
    o = opts.to_abi(N, B.shape[0], 3)
    assert (o.max_outer, o.max_inner, o.n_knots, o.n_tab) == (20, 50, N, N)


def test_trajopt_surface_rejects_what_the_kernel_cannot_do(pkg):
    to = pkg.trajopt
    with pytest.raises(ValueError):
        to.Model(to.DerivFunction(np.eye(3), np.zeros((4, 3))), 6, 3)
    obj = to.LQRObjective(np.ones((8, 8)), np.eye(3), np.eye(8), np.zeros(8), 10)
    with pytest.raises(ValueError, match="diagonal"):
        obj.diagonals()
    p = to.Problem(to.Model(to.DerivFunction(np.eye(3), np.zeros((10, 3))), 8, 3), None, N=10)
    with pytest.raises(ValueError, match="rk3"):
        to.BatchProblem([p])


def test_shard_range_partitions_exactly(pkg):
    sr = pkg.sweep.shard_range
    for T in (1, 7, 8, 1024, 65536, 1000):
        for W in (1, 2, 3, 8):
            if W > T:
                continue
            blocks = [sr(T, r, W) for r in range(W)]
            assert blocks[0][0] == 0 and blocks[-1][1] == T
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_inclination_sweep_dealt_out_to_ranks_covers_the_sweep(pkg):
    """configs[3] shards: contiguous blocks (stride 1) or the sweep dealt out to the ranks (j0 = rank, stride = world) — either
    way the ranks together hold every inclination i = 90 (j + 1/2) / T_total of src/paper_images/heatmap.jl:120 exactly once"""
    ss = pkg.slew_setup
    W, T_total = 4, 64
    want = 90.0 * (np.arange(T_total) + 0.5) / T_total
    for shards in ([ss.workload_inclination_sweep(T=16, N=20, j0=16 * r, T_total=T_total, tables=False) for r in range(W)],
                   [ss.workload_inclination_sweep(T=16, N=20, j0=r, stride=W, T_total=T_total, tables=False) for r in range(W)]):
        inc = np.sort(np.concatenate([b.meta["kep"][:, 2] for b in shards]))
        np.testing.assert_allclose(inc, want, rtol=0, atol=1e-12)
    dealt = ss.workload_inclination_sweep(T=16, N=20, j0=1, stride=W, T_total=T_total, tables=False)
    assert dealt.meta["kep"][0, 2] == want[1] and dealt.meta["kep"][1, 2] == want[1 + W]
    assert dealt.meta["error_state"] == 1 and dealt.meta["max_outer"] == 3 and dealt.meta["max_inner"] == 50


def test_model_hooks_select_error_state_mode(pkg):
    """Model(f!, n, m, quaternion_error, quaternion_expansion) as at src/monte_carlo.jl:158"""
    to = pkg.trajopt
    f = to.DerivFunction(np.eye(3) * 1e-3, np.zeros((10, 3)))
    assert to.Model(f, 8, 3).error_state == 0
    m = to.rk3(to.Model(f, 8, 3, to.quaternion_error, to.quaternion_expansion))
    assert m.error_state == 1 and m.integrator == 3
    with pytest.raises(ValueError):
        to.Model(f, 8, 3, to.quaternion_error, None).error_state
    with pytest.raises(ValueError):
        to.Model(f, 8, 3, lambda a, b: a - b, lambda *a: None).error_state
    o = to.AugmentedLagrangianSolverOptions().to_abi(10, 10, 3, error_state=1)
    assert o.error_state == 1


def test_batch_of_problems_with_different_horizons_is_ragged(pkg):
    """the Monte-Carlo gives every run its own horizon (src/monte_carlo.jl:140-145): BatchProblem pads to the longest"""
    ss, to = pkg.slew_setup, pkg.trajopt
    J = ss.INERTIA["1U"]
    B = ss.dipole_btable(80, 0.2, 6771.0, 96.6)
    probs = []
    for N in (80, 50):
        model = to.rk3(to.Model(to.DerivFunction(J, B, rows_per_knot=1.0), 8, 3))
        obj = to.LQRObjective(np.diag([1.0] * 7 + [0.0]), np.eye(3), np.diag([10.0] * 7 + [0.0]), np.r_[0, 0, 0, 1, 0, 0, 0, 1.0], N)
        cons = to.Constraints(N)
        for k in range(1, N):
            cons[k] += to.BoundConstraint(8, 3, u_max=19, u_min=-19)
        cons[N] += to.goal_constraint(obj.xf)
        p = to.Problem(model, obj, constraints=cons, x0=np.r_[0, 0, 0, 0, 1.0, 0, 0, 0], xf=obj.xf, N=N, dt=0.2)
        to.initial_controls_(p, np.full((3, N - 1), 1e-3))
        probs.append(p)
    bp = to.BatchProblem(probs)
    assert bp.arrays.N == 80 and list(bp.arrays.n_knots) == [80, 50]
    assert bp.arrays.U0.shape == (2, 79, 3) and np.all(bp.arrays.U0[1, 49:] == 0) and np.all(bp.arrays.U0[1, :49] == 1e-3)
    assert to.BatchProblem(probs[:1]).arrays.n_knots is None


def test_batched_bryson_weights_equal_the_per_trial_functions(pkg):
    """tsat_bryson_eigen_axis_batch (the Monte-Carlo's per-trial guess + weights, src/monte_carlo.jl:161-176, as one host call)
    against eigen_axis_slew + bryson_weights trial by trial: the same formulas, equal to rounding"""
    ss = pkg.slew_setup
    x0 = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    xf = np.array([0.0, 0.0, 0.0, np.sqrt(2.0) / 2.0, np.sqrt(2.0) / 2.0, 0.0, 0.0])
    n = np.r_[5, 6, 7, 13, 797, 3130, np.random.default_rng(3).integers(20, 3131, size=90)]
    for name, J in ss.INERTIA.items():
        Qd, Qfd, Rd = ss.bryson_weights_ragged(x0, xf, n, 0.0, 0.2, J, 0.1, 1e3)
        for j, k in enumerate(n):
            q, qf, r = ss.bryson_weights(ss.eigen_axis_slew(x0, xf, 0.2 * np.arange(k), rates_only=True)[0], J, 0.2, 0.1, 1e3)
            np.testing.assert_allclose(Qd[j], q, rtol=1e-13); np.testing.assert_allclose(Qfd[j], qf, rtol=1e-13)
            np.testing.assert_allclose(Rd[j], r, rtol=1e-12)
    with pytest.raises(ValueError):
        ss.bryson_weights_ragged(x0, xf, [2, 40], 0.0, 0.2, ss.INERTIA["1U"], 0.1, 1e3)
    with pytest.raises(ValueError):
        ss.bryson_weights_ragged(x0, x0, [40], 0.0, 0.2, ss.INERTIA["1U"], 0.1, 1e3)


def test_sharded_workloads_concatenate_to_the_whole(pkg):
    """configs[3] / configs[4] instances do not depend on the world size: RAAN, anomaly, q0 and U0 of GLOBAL trajectory j come from
    a counter-based stream keyed by (seed, j) (slew_setup.trajectory_stream), so shards built with j0 = r T / W for W in {1, 2, 8}
    concatenate to the same arrays, bit for bit (sharded sweep = concatenation of the single-GPU results;
    src/paper_images/heatmap.jl:114-127 draws once per run index)."""
    import os, sys
    ss = pkg.slew_setup
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    T, N = 64, 40
    whole = ss.workload_inclination_sweep(T=T, N=N, j0=0, T_total=T)
    fields = ("x0", "xf", "U0", "Qd", "Qfd", "Rd", "ulo", "uhi", "dt", "tau0", "dtau", "Jmat")
    for W in (2, 8):
        parts = [ss.workload_inclination_sweep(T=T // W, N=N, j0=r * T // W, T_total=T) for r in range(W)]
        for f in fields:
            assert np.array_equal(np.concatenate([getattr(q, f) for q in parts]), getattr(whole, f)), (W, f)
        assert np.array_equal(np.concatenate([q.Btab[q.btab_idx] for q in parts]), whole.Btab[whole.btab_idx]), W
        assert np.array_equal(np.concatenate([q.meta["kep"] for q in parts]), whole.meta["kep"]), W
    # a shard does not depend on its neighbours either: another shard size starting at the same global index gives the same head
    a, b = ss.workload_inclination_sweep(T=8, N=N, j0=24, T_total=T), ss.workload_inclination_sweep(T=16, N=N, j0=24, T_total=T)
    assert np.array_equal(a.U0, b.U0[:8]) and np.array_equal(a.meta["kep"], b.meta["kep"][:8])
    # the receding-horizon workload of bench.py --config 4 (4096 trajectories sharded): the same property through its own builder
    class _S:      # the two attributes mpc_workload touches on a solver
        class opts:
            class opts_uncon:
                dJ_counter_limit = 1
            @staticmethod
            def to_abi(N, n_tab, m, error_state=0):
                return pkg.trajopt.AugmentedLagrangianSolverOptions().to_abi(N, n_tab, m, error_state=error_state)
    w, _ = bench.mpc_workload(ss, _S, 32, 20, 8, 0, 1)
    for W in (2, 8):
        parts = [bench.mpc_workload(ss, _S, 32 // W, 20, 8, r * 32 // W, 1)[0] for r in range(W)]
        for f in ("x0", "U0", "Qd", "Rd"):
            assert np.array_equal(np.concatenate([getattr(q, f) for q in parts]), getattr(w, f)), (W, f)
