"""The Monte-Carlo experiment end to end (src/monte_carlo.jl:107-262, 334-343): host arithmetic between the stages,
chunk/shard invariance, result files; (gpu tier) the chained GPU stages against the chained oracle stages."""
import numpy as np
import pytest

from mc_stages import OracleStages


def small_setup(pkg, **kw):
    # same script, coarser clocks so that the CPU oracle finishes in seconds: 1 s knots, 600-row tables
    return pkg.monte_carlo.MonteCarloSetup(N=600, dt=1.0, outer=2, inner=4, **kw)


def test_script_arithmetic(pkg, ol):
    mc = pkg.monte_carlo
    s = small_setup(pkg)
    out = mc.run_trials(OracleStages(ol), seed=11, lo=0, hi=3, setup=s)
    assert np.all(out["A"][:, 0] == 0) and np.all(out["A"][:, 1] == 6771.0) and np.all(out["A"][:, 2] == 96.6)
    assert np.all((out["A"][:, [3, 5]] >= 0) & (out["A"][:, [3, 5]] < 360))
    assert np.all(out["found"]) and np.all(out["tf_index"] > 0)
    np.testing.assert_allclose(out["t_final"], out["tf_index"] * 2400.0 / 600)                 # (:140)
    for j in range(3):
        n = out["n_knots"][j]
        assert n == len(np.arange(0.0, out["t_final"][j] + 1e-9, s.dt))                          # length(t0:dt:t_final) (:145)
        assert out["states"][j].shape == (7, n) and out["control_inputs"][j].shape == (3, n - 1)
        assert out["sim_states"][j].shape == (7, n) and out["sim_control_inputs"][j].shape == (3, n - 1)
        assert out["B_ECI_total"][j].shape == (2 * s.N, 3) and np.all(out["B_ECI_total"][j][-1] == 0)
        np.testing.assert_array_equal(out["states"][j][:, 0], [0, 0, 0, 1, 0, 0, 0])
        assert out["slew_time"][j] == (s.dt * out["tracking_stats"]["slew_index"][j] if not out["fails"][j] else s.dt * n)
    sm = mc.summarize(out)
    assert sm["number_sims"] == 3 and sm["slew_time_mean"] == pytest.approx(np.mean(out["slew_time"]))
    assert list(sm["fails"]) == list(np.nonzero(out["fails"])[0])


def test_knot_counts_follow_julia_range_length(pkg):
    """length(range(0, step=.2, stop=idx*0.48)) (src/monte_carlo.jl:140,145): Julia counts float ranges in twice precision,
    so horizon indices that are multiples of 5 (2.4/0.2 = 11.999999999999998 in plain fp64) must not lose their last knot"""
    mc = pkg.monte_carlo
    idx = np.arange(1, 5001)
    n = mc.knot_counts(idx * 2400.0 / 5000, 0.0, 0.2)
    assert np.array_equal(n, (idx * 12) // 5 + 1)                      # exact rational count for the script's constants
    assert list(mc.knot_counts(np.array([5, 10, 20]) * 0.48, 0.0, 0.2)) == [13, 25, 49]
    assert list(mc.knot_counts(np.array([2.4, 2.39999, 2.5, 0.2]), 0.0, 0.2)) == [13, 12, 13, 2]


def test_field_replay_rate_option(pkg, ol):
    """the script replays the resampled table at 1/(tf - t0) of the COARSE span; "physical" at its own"""
    mc = pkg.monte_carlo
    A = mc.draw_orbits(5, 0, 2, small_setup(pkg))
    tfin = np.array([300.0, 480.0])
    B = np.zeros((2, 1200, 3))
    phys, _ = mc.build_batch(np.arange(2), tfin, B, 5, small_setup(pkg))
    ref, _ = mc.build_batch(np.arange(2), tfin, B, 5, small_setup(pkg, field_rate="reference"))
    np.testing.assert_allclose(phys.dtau, 600 / tfin)                 # one table row per t_final/N seconds
    np.testing.assert_allclose(ref.dtau, [600 / 2400.0] * 2)
    assert np.array_equal(phys.U0, ref.U0) and A.shape == (2, 6)


def test_trials_do_not_depend_on_chunking(pkg, ol):
    mc = pkg.monte_carlo
    s, st = small_setup(pkg), OracleStages(ol)
    whole = mc.run_trials(st, 3, 0, 4, s)
    parts = [mc.run_trials(st, 3, 0, 1, s), mc.run_trials(st, 3, 1, 4, s)]
    for k in ("A", "t_final", "slew_time", "fails", "n_knots"):
        assert np.array_equal(whole[k], np.concatenate([p[k] for p in parts])), k
    assert np.array_equal(whole["sim_states"][2], parts[1]["sim_states"][1])
    one = mc.monte_carlo(st, number_sims=4, seed=3, setup=s, chunk=3)
    assert np.array_equal(one["slew_time"], whole["slew_time"]) and len(one["parts"]) == 2


@pytest.mark.parametrize("fmt", ["h5", "npz"])
def test_result_files_round_trip(pkg, ol, tmp_path, fmt):
    """the script's file set (src/monte_carlo.jl:334-343): {n}_A.h5 and per trial states / control / B_N / t_total files, the
    datasets named as there ("one_state" and "states" both hold sim_states[i]); HDF5 through the system's libhdf5"""
    import shutil
    import subprocess
    mc, rs, h5 = pkg.monte_carlo, pkg.results, pkg.hdf5io
    if fmt == "h5" and not h5.available():
        pytest.skip("no libhdf5 in this image")
    out = mc.run_trials(OracleStages(ol), 21, 0, 2, small_setup(pkg))
    files = rs.write_monte_carlo(str(tmp_path), out, fmt=fmt)
    assert sorted(f.split("/")[-1] for f in files) == sorted(
        [f"2_A.{fmt}"] + [f"2_{k}_{i}.{fmt}" for k in ("states", "control", "B_N", "t_total") for i in (1, 2)])   # (:334-343)
    back = rs.read_monte_carlo(str(tmp_path), 2)
    assert np.array_equal(back["A"], out["A"])
    for i in (1, 2):
        assert np.array_equal(back["states"][i], out["sim_states"][i - 1])
        assert np.array_equal(back["control"][i], out["sim_control_inputs"][i - 1])
        assert np.array_equal(back["B_ECI"][i], out["B_ECI_total"][i - 1])
        assert np.array_equal(back["t_total"][i], out["t_total"][i - 1])
    if fmt == "h5":
        n1 = out["sim_states"][0].shape[1]
        st = str(tmp_path / "2_states_1.h5")
        # what HDF5.jl stores for the script's 7 x n array: an (n, 7) float64 dataset, under both names
        assert h5.h5read(st, "states").shape == (n1, 7) and np.array_equal(h5.h5read(st, "one_state"), h5.h5read(st, "states"))
        assert h5.h5read(str(tmp_path / "2_A.h5"), "A").shape == (6, 2)
        dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if shutil.os.path.exists("/opt/conda/bin/h5dump") else None)
        if dump:    # an independent reader of the format agrees
            hdr = subprocess.run([dump, "-H", st], capture_output=True, text=True, check=True).stdout
            assert 'DATASET "states"' in hdr and 'DATASET "one_state"' in hdr and "H5T_IEEE_F64LE" in hdr and f"( {n1}, 7 )" in hdr


@pytest.mark.gpu
def test_gpu_experiment_matches_oracle_experiment(pkg, ol):
    """six trials at the script's own clocks (N = 5000, 0.2 s knots, 5 x 10 budget): every stage on the GPU, chained,
    against every stage on the CPU oracle, chained"""
    mc, to = pkg.monte_carlo, pkg.trajopt
    s = mc.MonteCarloSetup()
    solver = to.AugmentedLagrangianSolver(None, None)
    got = mc.run_trials(mc.GpuStages(solver), 2019, 0, 6, s)
    ref = mc.run_trials(OracleStages(ol, nthreads=min(6, ol.num_procs())), 2019, 0, 6, s)
    assert np.array_equal(got["tf_index"], ref["tf_index"]) and np.array_equal(got["n_knots"], ref["n_knots"])
    assert np.all(got["found"]) and got["n_knots"].min() >= 2
    for j in range(6):
        assert np.max(np.abs(got["B_ECI_total"][j] - ref["B_ECI_total"][j])) < 1e-9 * 5e-5
        assert np.array_equal(got["solve_stats"]["inner_iters"][j], ref["solve_stats"]["inner_iters"][j])
        assert np.max(np.abs(got["states"][j] - ref["states"][j])) < 1e-8
        assert np.max(np.abs(got["control_inputs"][j] - ref["control_inputs"][j])) < 1e-7
        assert np.max(np.abs(got["sim_states"][j] - ref["sim_states"][j])) < 1e-7
    assert np.array_equal(got["fails"], ref["fails"]) and np.array_equal(got["slew_time"], ref["slew_time"])
    solver.close()


@pytest.mark.gpu
def test_gpu_resident_tables_are_the_downloaded_ones(pkg, ol):
    """field tables left on the device between the stages (Btab = NULL through the ABI) give the run the host round trip gives"""
    mc, to = pkg.monte_carlo, pkg.trajopt
    s = small_setup(pkg)
    solver = to.AugmentedLagrangianSolver(None, None)

    class HostTables(mc.GpuStages):
        resident_tables = False

    a = mc.run_trials(mc.GpuStages(solver), 5, 0, 5, s)
    b = mc.run_trials(HostTables(solver), 5, 0, 5, s)
    c = mc.run_trials(mc.GpuStages(solver), 5, 0, 5, s, keep_trajectories=False)
    for k in ("tf_index", "n_knots", "slew_time", "fails"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), k
    for j in range(len(a["selected"])):
        assert np.array_equal(a["states"][j], b["states"][j]) and np.array_equal(a["sim_states"][j], b["sim_states"][j])
        assert np.array_equal(a["B_ECI_total"][j], b["B_ECI_total"][j])
    solver.close()
