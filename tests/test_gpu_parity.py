"""GPU tier (`pytest -m gpu`): the HIP path, called through the C ABI, against the CPU oracle on identical
inputs (bar: 1e-9 absolute on X and U at equal iteration counts, fp64 — BASELINE.md §3), against the committed
golden fixtures, and — at BASELINE.json's full size — through size-independent properties."""
import ctypes as C

import numpy as np
import pytest

import helpers
from conftest import assert_same_solution, oracle_options

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(pkg):
    import torch

    assert torch.cuda.is_available(), "the gpu tier needs an MI355X"
    s = pkg.trajopt.AugmentedLagrangianSolver(None, None, device=0)   # raises if the HIP library cannot open a gfx950 GPU
    yield s
    s.close()


def gpu_solve(pkg, solver, b, o, trace_rows=0):
    a = helpers.abi_options_like(o, pkg, b.N, b.n_tab)
    solver.upload(b, a.max_linesearch)
    solver.trace(trace_rows)
    solver.run(a)
    res = solver.download()
    res["trace"] = solver.trace_download() if trace_rows else None
    return res


def k_close(ref, got, rel=1e-9):
    scale = max(float(np.max(np.abs(ref["K"]))), 1.0)
    assert np.max(np.abs(ref["K"] - got["K"])) < rel * scale


# ----------------------------------------------------------------------------------------------- fixtures
@pytest.mark.parametrize("name", helpers.golden_cases())
def test_gpu_reproduces_golden(pkg, ol, solver, name):
    b, o, ref = helpers.load_case(name, pkg, ol)
    got = gpu_solve(pkg, solver, b, o)
    assert_same_solution(ref, got, tol=1e-9)
    k_close(ref, got)


# ----------------------------------------------------------------------------------------------- oracle parity
@pytest.mark.parametrize("N", [2, 3, 33, 34, 49, 50, 97])
def test_gpu_ragged_knot_counts(pkg, ol, solver, N):
    b = pkg.slew_setup.workload_monte_carlo(T=5, N=N, seed=100 + N, degenerate_rd=0.03)   # N = 2: no acceleration sample in the guess -> Bryson R undefined
    o = oracle_options(ol, max_outer=2, max_inner=3, dj_counter_limit=1)
    assert_same_solution(ol.solve_batch(b, o), gpu_solve(pkg, solver, b, o))


@pytest.mark.parametrize("integ", [3, 4])
def test_gpu_monte_carlo_1000_knots(pkg, ol, solver, integ):
    """configs[1] at full horizon, a slice the oracle finishes in seconds"""
    b = pkg.slew_setup.workload_monte_carlo(T=24, N=1000)
    o = oracle_options(ol, max_outer=5, max_inner=10, dj_counter_limit=1, integrator=integ)
    ref = ol.solve_batch(b, o, nthreads=min(16, ol.num_procs()), trace_rows=64)
    got = gpu_solve(pkg, solver, b, o, trace_rows=64)
    assert_same_solution(ref, got, tol=1e-9)
    k_close(ref, got)
    # identical line-search decisions, iteration by iteration
    assert np.array_equal(ref["trace"][:, :, 4], got["trace"][:, :, 4])
    np.testing.assert_allclose(got["trace"][:, :, 3], ref["trace"][:, :, 3], rtol=1e-9)


def test_gpu_error_state_mode_1000_knots(pkg, ol, solver):
    """the quaternion hooks of the old-API Monte-Carlo (src/monte_carlo.jl:158), full horizon"""
    b = pkg.slew_setup.workload_monte_carlo(T=16, N=1000)
    o = oracle_options(ol, max_outer=5, max_inner=10, dj_counter_limit=1, error_state=1)
    ref = ol.solve_batch(b, o, nthreads=min(16, ol.num_procs()))
    got = gpu_solve(pkg, solver, b, o)
    assert_same_solution(ref, got, tol=1e-9)
    k_close(ref, got)
    assert np.all(got["K"][:, :, 6, :] == 0)


@pytest.mark.parametrize("N", [2, 53, 54, 105])
def test_gpu_error_state_ragged(pkg, ol, solver, N):
    b = pkg.slew_setup.workload_monte_carlo(T=3, N=N, seed=300 + N, degenerate_rd=0.03)   # N = 2: no acceleration sample in the guess -> Bryson R undefined
    for integ in (3, 4):
        o = oracle_options(ol, max_outer=2, max_inner=3, dj_counter_limit=1, error_state=1, integrator=integ)
        assert_same_solution(ol.solve_batch(b, o), gpu_solve(pkg, solver, b, o))


def test_gpu_single_slew_config0(pkg, ol, solver):
    """configs[0]: the reference's own case (src/TortoiseSat.jl:117-199) — 500 knots, |u| <= 1, 20 x 50 budget"""
    b = pkg.slew_setup.workload_single_slew(N=500)
    o = oracle_options(ol, max_outer=20, max_inner=50)
    ref = ol.solve_batch(b, o)
    got = gpu_solve(pkg, solver, b, o)
    assert_same_solution(ref, got, tol=1e-9)


def test_gpu_random_orbit_tables(pkg, ol, solver):
    b = pkg.slew_setup.workload_monte_carlo(T=12, N=200, seed=20190531, random_orbit=True)   # configs[2] inputs (fp64)
    o = oracle_options(ol, max_outer=3, max_inner=6, dj_counter_limit=1)
    assert_same_solution(ol.solve_batch(b, o, nthreads=4), gpu_solve(pkg, solver, b, o))


def test_gpu_inclination_sweep_inputs(pkg, ol, solver):
    b = pkg.slew_setup.workload_inclination_sweep(T=8, N=150, j0=4096, T_total=65536)         # configs[3] inputs
    o = oracle_options(ol, max_outer=3, max_inner=8, dj_counter_limit=1)
    assert_same_solution(ol.solve_batch(b, o, nthreads=4), gpu_solve(pkg, solver, b, o))


def test_gpu_edge_options(pkg, ol, solver):
    ss = pkg.slew_setup
    b = ss.workload_monte_carlo(T=3, N=45, seed=21)
    for kw in (dict(max_linesearch=1), dict(max_linesearch=32), dict(terminal_mask=0), dict(terminal_mask=0b1111000),
               dict(penalty_init=10.0, penalty_scale=3.0), dict(reg_init=1e-3)):
        o = oracle_options(ol, max_outer=3, max_inner=4, **kw)
        assert_same_solution(ol.solve_batch(b, o), gpu_solve(pkg, solver, b, o))
    b.dtau[:] = [0.37, 1.9, 1.0]
    b.tau0[:] = [3.2, 0.0, 7.0]
    o = oracle_options(ol, max_outer=2, max_inner=3)
    assert_same_solution(ol.solve_batch(b, o), gpu_solve(pkg, solver, b, o))


def test_gpu_full_inertia_tensor(pkg, ol, solver):
    ss = pkg.slew_setup
    b = ss.workload_monte_carlo(T=2, N=60, seed=8)
    J = np.array([[2.0e-3, 1.0e-4, -2.0e-4], [1.0e-4, 1.5e-3, 3.0e-4], [-2.0e-4, 3.0e-4, 2.5e-3]])
    b.Jmat[:] = ss.jmat_cm(J)
    o = oracle_options(ol, max_outer=2, max_inner=4)
    assert_same_solution(ol.solve_batch(b, o), gpu_solve(pkg, solver, b, o))


def test_gpu_diagonal_3u_inertia(pkg, ol, solver):
    ss = pkg.slew_setup
    b = ss.workload_monte_carlo(T=4, N=300, seed=41)
    b.Jmat[:] = ss.jmat_cm(ss.INERTIA["3U"])          # diagonal, not isotropic (src/input_parameters.jl:45-51)
    for es in (0, 1):
        o = oracle_options(ol, max_outer=3, max_inner=6, error_state=es)
        assert_same_solution(ol.solve_batch(b, o, nthreads=4), gpu_solve(pkg, solver, b, o))


def test_gpu_ragged_batch(pkg, ol, solver):
    """per-trajectory horizons (tsat_batch_knots): empty-ish (2 knots), chunk-boundary and full-length trajectories
    in one launch, both modes; then back to a uniform batch on the same handle"""
    lib = pkg._abi.load()
    b = pkg.slew_setup.workload_monte_carlo(T=6, N=260, seed=51, random_orbit=True)
    b.n_knots = np.array([260, 2, 57, 58, 113, 200], dtype=np.int32)
    for es in (0, 1):
        o = oracle_options(ol, max_outer=3, max_inner=5, dj_counter_limit=1, error_state=es)
        r, g = ol.solve_batch(b, o, nthreads=4), gpu_solve(pkg, solver, b, o)
        assert_same_solution(r, g)
        for t, n in enumerate(b.n_knots):
            assert np.all(g["X"][t, n:] == 0) and np.all(g["U"][t, n - 1:] == 0) and np.all(g["K"][t, n - 1:] == 0)
    bad = np.array([260, 1, 57, 58, 113, 200], dtype=np.int32)
    assert lib.tsat_batch_knots(solver._h, pkg._abi.as_ip(bad)) < 0        # knot counts are validated
    b.n_knots = None
    o = oracle_options(ol, max_outer=2, max_inner=3)
    assert_same_solution(ol.solve_batch(b, o, nthreads=4), gpu_solve(pkg, solver, b, o))


def test_gpu_failure_statuses(pkg, ol, solver):
    ss, abi = pkg.slew_setup, pkg._abi
    b = ss.workload_monte_carlo(T=2, N=40, seed=5)
    b.Rd[:] = -1e-4
    o = oracle_options(ol, max_outer=2, max_inner=3, reg_max=1e-6)
    r, g = ol.solve_batch(b, o), gpu_solve(pkg, solver, b, o)
    assert np.all(g["stats"]["status"] == abi.TSAT_REG_FAIL)
    assert_same_solution(r, g)
    b2 = ss.workload_monte_carlo(T=2, N=40, seed=5)
    b2.U0[0] = 1e12
    o2 = oracle_options(ol, max_outer=2, max_inner=3)
    r2, g2 = ol.solve_batch(b2, o2), gpu_solve(pkg, solver, b2, o2)
    assert g2["stats"]["status"][0] == abi.TSAT_DIVERGED and np.array_equal(r2["stats"]["status"], g2["stats"]["status"])
    np.testing.assert_allclose(g2["X"][1], r2["X"][1], atol=1e-9)   # the healthy neighbour is unaffected


def test_gpu_api_errors_are_codes_not_crashes(pkg, ol, solver):
    lib, abi = pkg._abi.load(), pkg._abi
    b = pkg.slew_setup.workload_monte_carlo(T=2, N=30)
    o = helpers.abi_options_like(oracle_options(ol, max_outer=1, max_inner=1), pkg, b.N, b.n_tab)
    solver.upload(b, o.max_linesearch)
    for field, bad in (("integrator", 5), ("precision", 16), ("error_state", 2), ("max_linesearch", 33), ("n_knots", 31)):
        o2 = o.copy()
        setattr(o2, field, bad)
        rc = lib.tsat_batch_run(solver._h, C.byref(o2), None)
        assert rc < 0 and lib.tsat_last_error(solver._h)
    assert lib.tsat_batch_reserve(solver._h, 0, 30, 30, 1, 20) < 0
    assert lib.tsat_batch_run(solver._h, C.byref(o), None) == 0      # the handle is still usable
    # widened entry points: bad arguments come back as codes with a message, never as a launch
    d, ip = abi.as_dp, abi.as_ip
    bo = abi.BtableOptions()
    lib.tsat_btable_default_options(C.byref(bo))
    kep = np.array([[0.0, 6771.0, 96.6, 10.0, 0.0, 20.0]]); t0 = np.zeros(1); tf = np.full(1, 60.0)
    B = np.zeros((1, 2 * 8, 3))
    for field, bad in (("date", 2020.0), ("date", 2014.9), ("n_half", 0)):
        b2 = abi.BtableOptions.from_buffer_copy(bo)
        b2.n_half = 8
        setattr(b2, field, bad)
        assert lib.tsat_btable_batch(solver._h, C.byref(b2), 1, d(kep), d(t0), d(tf), d(B), None) < 0
        assert lib.tsat_last_error(solver._h)
    bo.n_half = 8
    for bad_kep, bad_tf in ((np.array([[1.0, 6771.0, 96.6, 0, 0, 0]]), tf), (np.array([[0.0, -1.0, 96.6, 0, 0, 0]]), tf), (kep, np.zeros(1))):
        assert lib.tsat_btable_batch(solver._h, C.byref(bo), 1, d(np.ascontiguousarray(bad_kep, dtype=np.float64)), d(t0), d(bad_tf), d(B), None) < 0
    assert lib.tsat_btable_batch(solver._h, C.byref(bo), 1, d(kep), d(t0), d(tf), d(B), None) == 0
    assert lib.tsat_horizon_batch(solver._h, 1, 0, d(B), d(t0), d(tf), ip(np.zeros(1, np.int32)), None) < 0
    res = solver.download(want_K=False)
    Qd, Qfd, Rd = pkg.tracking.tvlqr_weights(b.T)
    b.n_knots = np.array([30, 31], dtype=np.int32)                    # beyond the batch stride
    with pytest.raises(RuntimeError):
        pkg.tracking.attitude_simulation(solver, b, res["X"], res["U"], b.x0, Qd, Qfd, Rd)
    b.n_knots = np.array([30, 1], dtype=np.int32)
    with pytest.raises(RuntimeError):
        pkg.tracking.attitude_simulation(solver, b, res["X"], res["U"], b.x0, Qd, Qfd, Rd)


def test_gpu_upload_rejects_degenerate_inputs(pkg, ol, solver):
    """non-finite weights / bounds and singular inertias are API errors at upload, not DIVERGED statuses later"""
    ss = pkg.slew_setup
    with pytest.raises(ValueError):                            # Bryson R of a guess without acceleration is undefined (1/0)
        ss.bryson_weights(np.tile([[0.01, 0.0, 0.0]], (5, 1)), ss.INERTIA["1U"], 0.2, 0.1, 1e3)
    assert np.all(ss.bryson_weights(np.tile([[0.01, 0.0, 0.0]], (5, 1)), ss.INERTIA["1U"], 0.2, 0.1, 1e3, degenerate_rd=0.03)[2] == 0.03)
    with pytest.raises(ValueError):
        ss.eigen_axis_slew(np.r_[0, 0, 0, 1, 0, 0, 0.], np.r_[0, 0, 0, 1, 0, 0, 0.], 0.2 * np.arange(5))
    for field, val in (("Rd", np.inf), ("Qd", np.nan), ("uhi", np.inf), ("Jmat", 0.0), ("dt", 0.0)):
        b = ss.workload_monte_carlo(T=3, N=20, seed=2)
        getattr(b, field)[1] = val
        with pytest.raises(RuntimeError):
            solver.upload(b, 20)
    b = ss.workload_monte_carlo(T=3, N=20, seed=2)
    solver.upload(b, 20)                                       # the handle is still usable


def test_gpu_native_sweep_allgather_single_rank(pkg, ol, solver):
    """tsat_comm_* / tsat_sweep_allgather (RCCL behind the C ABI) with a one-rank communicator: the gathered arrays are
    the downloaded ones, through host buffers and through caller-owned device buffers"""
    import torch
    b = pkg.slew_setup.workload_monte_carlo(T=5, N=64, seed=31)
    o = oracle_options(ol, max_outer=2, max_inner=3, dj_counter_limit=1)
    ref = gpu_solve(pkg, solver, b, o)
    solver.comm_init(solver.comm_unique_id(), 0, 1)
    got = solver.sweep_allgather()
    assert np.array_equal(got["X"], ref["X"]) and np.array_equal(got["U"], ref["U"]) and np.array_equal(got["stats"], ref["stats"])
    X = torch.empty((5, 64, 7), dtype=torch.float64, device="cuda")
    st = torch.empty((5, 64), dtype=torch.uint8, device="cuda")
    solver.sweep_allgather(X.data_ptr(), None, st.data_ptr(), on_device=True)
    assert np.array_equal(X.cpu().numpy(), ref["X"]) and np.array_equal(st.cpu().numpy().view(pkg._abi.STATS_DTYPE).reshape(-1), ref["stats"])
    solver.comm_destroy()
    with pytest.raises(RuntimeError):
        solver.sweep_allgather()


def test_gpu_igrf_tables_attached_to_a_workload(pkg, ol, solver):
    """the bench workload's field: IGRF-12 along the orbit from tsat_btable_batch (one orbit, and one per trajectory)"""
    ss, mg = pkg.slew_setup, pkg.magnetic
    o = oracle_options(ol, max_outer=3, max_inner=6, dj_counter_limit=1, error_state=1)
    for kw in (dict(), dict(random_orbit=True, tables=False)):
        b = mg.attach_igrf_tables(solver, ss.workload_monte_carlo(T=6, N=300, seed=17, **kw))
        assert b.n_tab == 308 and b.Btab.shape == ((6 if kw else 1), 308, 3) and np.all(b.dtau == 1.0)
        Bo = ol.btable_batch(b.meta["kep"], 0.0, 60.0, 300, want_pos=False)[0][:, :308]
        assert np.max(np.abs(b.Btab - Bo)) < 1e-9 * np.max(np.abs(Bo))
        assert_same_solution(ol.solve_batch(b, o, nthreads=6), gpu_solve(pkg, solver, b, o))


def test_gpu_dense_and_wide_builds_agree(pkg, ol, solver):
    """tsat_set_kernel_variant: the seven fp64 builds of the solve kernel (wide, dense, packed with 4 and with 8 trajectories per
    wavefront at two wavefronts per SIMD, with 4, 8 and 16 at one) give bit-identical results; batches above 1024 trajectories
    take the dense one automatically, from 2048 / 4097 / 8193 / 16384 the packed4w, packed8w, packed8 and packed16w ones"""
    b = pkg.slew_setup.workload_monte_carlo(T=37, N=300, seed=12, random_orbit=True)      # (partial last wavefronts of 4, 8 and 16)
    o = oracle_options(ol, max_outer=3, max_inner=6, dj_counter_limit=1)
    out = {}
    for name, v in (("wide", 1), ("dense", 2), ("packed", 3), ("packed8", 4), ("packed8w", 5), ("packed16w", 6), ("packed4w", 7)):
        solver.set_kernel_variant(v)
        out[name] = gpu_solve(pkg, solver, b, o)
    solver.set_kernel_variant(0)
    for other in ("dense", "packed", "packed8", "packed8w", "packed16w", "packed4w"):
        for k in ("X", "U", "K"):
            assert np.array_equal(out["wide"][k], out[other][k]), (other, k)
        for f in out["wide"]["stats"].dtype.names:      # n_forward counts executed sweeps (a packed sweep carries fewer candidates)
            assert f == "n_forward" or np.array_equal(out["wide"]["stats"][f], out[other]["stats"][f]), (other, f)
    assert_same_solution(ol.solve_batch(b, o, nthreads=8), out["dense"])
    big = pkg.slew_setup.workload_monte_carlo(T=1030, N=40, seed=13)                     # automatic: dense
    o2 = oracle_options(ol, max_outer=2, max_inner=3, dj_counter_limit=1)
    assert_same_solution(ol.solve_batch(big, o2, nthreads=8), gpu_solve(pkg, solver, big, o2))
    with pytest.raises(RuntimeError):
        solver.set_kernel_variant(8)


def test_gpu_automatic_build_by_batch_size(pkg, ol, solver):
    """tsat_selected_build: which build a batch takes by itself (include/tortoise_hip.h, tsat_set_kernel_variant) — by size, and from
    16384 trajectories on by the iteration budget —, and that the launch it names is the one that runs: short solves of every size
    class in the automatic build, bit for bit those of the dense build, the last wavefront of each partial"""
    ss = pkg.slew_setup
    base = ss.workload_monte_carlo(T=64, N=24, seed=77)

    def batch(T):
        rep = lambda a: np.ascontiguousarray(np.concatenate([a] * ((T + 63) // 64))[:T])
        return ss.SlewBatch(base.N, base.n_tab, rep(base.x0), rep(base.xf), base.Btab, rep(base.btab_idx), rep(base.tau0), rep(base.dtau), rep(base.dt),
                            rep(base.Jmat), rep(base.Qd), rep(base.Qfd), rep(base.Rd), rep(base.ulo), rep(base.uhi), rep(base.U0))

    o = oracle_options(ol, max_outer=2, max_inner=3, dj_counter_limit=1, error_state=1)
    long_budget = oracle_options(ol, max_outer=3, max_inner=50, dj_counter_limit=1, error_state=1)
    solver.set_kernel_variant(0)
    for T, want in ((1024, 1), (1025, 2), (2047, 2), (2048, 7), (4096, 7), (4097, 5), (8192, 5), (8193, 4), (16383, 4), (16387, 6)):
        b = batch(T)
        a = helpers.abi_options_like(o, pkg, b.N, b.n_tab)
        solver.upload(b, a.max_linesearch)
        assert solver.selected_build(a)[0] == want, (T, solver.selected_build(a))
        a32 = helpers.abi_options_like(o, pkg, b.N, b.n_tab)
        a32.precision = 32
        assert solver.selected_build(a32)[0] == (2 if want == 1 else want), (T, "precision 32")
        if T >= 16384:
            al = helpers.abi_options_like(long_budget, pkg, b.N, b.n_tab)
            assert solver.selected_build(al) == (4, 2048), solver.selected_build(al)       # long budget: eight per wavefront, endgame at an eighth
        solver.run(a)
        auto = solver.download()
        if want != 2:
            solver.set_kernel_variant(2)
            solver.run(a)
            dense = solver.download()
            solver.set_kernel_variant(0)
            for k in ("X", "U", "K"):
                assert np.array_equal(auto[k], dense[k]), (T, k)
        for k in ("X", "U"):                               # the batch is 64 solves repeated: so are the results, into the last partial wavefront
            assert np.array_equal(auto[k][:64][: T % 64 or 64], auto[k][T - (T % 64 or 64):]), (T, k)


@pytest.mark.parametrize("variant", [3, 4, 5, 6, 7])
@pytest.mark.parametrize("es", [0, 1])
def test_gpu_packed_builds_ragged_and_tiny_horizons(pkg, ol, solver, variant, es):
    """the packed builds forced onto small odd batches: horizons of 2 ... 97 knots inside one batch (partial 16-knot passes,
    record-ring tails), a trajectory count that fills neither the last wavefront nor its backward passes"""
    b = pkg.slew_setup.workload_monte_carlo(T=13, N=97, seed=41 + es, degenerate_rd=0.03)
    b.n_knots = np.array([97, 2, 3, 16, 17, 18, 33, 5, 96, 49, 64, 65, 4], dtype=np.int32)
    o = oracle_options(ol, max_outer=2, max_inner=4, dj_counter_limit=1, error_state=es)
    ref = ol.solve_batch(b, o, nthreads=8)
    solver.set_kernel_variant(variant)
    got = gpu_solve(pkg, solver, b, o)
    solver.set_kernel_variant(0)
    assert_same_solution(ref, got)
    for t, n in enumerate(b.n_knots):
        assert np.all(got["X"][t, n:] == 0) and np.all(got["U"][t, n - 1:] == 0)


@pytest.mark.parametrize("variant,prec", [(3, 64), (4, 64), (3, 32)])
def test_gpu_packed_endgame_is_the_same_solve(pkg, ol, solver, variant, prec):
    """tsat_set_endgame: the packed launch parks its last live trajectories and a second kernel finishes each on a wavefront
    of its own. Which ones are parked depends on the order the hardware ran the waves in; the results must not: bit-identical
    to the launch without an endgame for a spread of iteration counts (ragged horizons, a diverging rollout, a regularisation
    failure), at several thresholds including "everything at once", and equal to the oracle."""
    T, N = 203, 64
    b = pkg.slew_setup.workload_monte_carlo(T=T, N=N, seed=77, random_orbit=True, degenerate_rd=0.03)
    rng = np.random.default_rng(5)
    b.n_knots = rng.integers(2, N + 1, size=T).astype(np.int32)
    b.n_knots[::3] = N
    b.U0[7] = 1e12
    b.Rd[11] = -1e-4
    o = oracle_options(ol, max_outer=3, max_inner=8, dj_counter_limit=1, error_state=1, reg_max=1e-2)
    a = helpers.abi_options_like(o, pkg, b.N, b.n_tab)
    a.precision = prec
    solver.upload(b, a.max_linesearch)

    def run():
        solver.run(a)
        return solver.download()

    solver.set_kernel_variant(variant)
    try:
        solver.set_endgame(0)
        plain = run()
        for at in (60, 150, 1000):
            solver.set_endgame(at)
            got = run()
            for k in ("X", "U", "K"):
                # (equal_nan: the diverged trajectory never has a backward sweep; its K is whatever the buffer held before)
                bad = [t for t in range(T) if not np.array_equal(plain[k][t], got[k][t], equal_nan=True)]
                assert not bad, (at, k, bad, plain["stats"]["status"][bad].tolist())
            for f in plain["stats"].dtype.names:
                assert f == "n_forward" or np.array_equal(plain["stats"][f], got["stats"][f]), (at, f)
        with pytest.raises(RuntimeError):
            solver.set_endgame(-2)
    finally:
        solver.set_endgame(-1)
        solver.set_kernel_variant(0)
    assert len(np.unique(plain["stats"]["inner_iters"])) > 5      # the iteration counts do spread
    if prec == 64:
        ref = ol.solve_batch(b, o, nthreads=8)
        assert np.array_equal(ref["stats"]["status"], plain["stats"]["status"]) and ref["stats"]["status"][7] == pkg._abi.TSAT_DIVERGED
        ok = np.flatnonzero(ref["stats"]["status"] != pkg._abi.TSAT_DIVERGED)      # (a diverged rollout holds whatever overflowed)
        pick = lambda r: dict(X=r["X"][ok], U=r["U"][ok], K=r["K"][ok], stats=r["stats"][ok])
        assert_same_solution(pick(ref), pick(plain))


def test_gpu_one_call_abi_entry(pkg, ol, solver):
    """tsat_solve_batch: the single call a Julia `ccall` would make in place of solve!(prob, solver)"""
    lib, abi = pkg._abi.load(), pkg._abi
    b = pkg.slew_setup.workload_monte_carlo(T=3, N=50, seed=4)
    oo = oracle_options(ol, max_outer=2, max_inner=4)
    o = helpers.abi_options_like(oo, pkg, b.N, b.n_tab)
    X = np.zeros((3, 50, 7)); U = np.zeros((3, 49, 3)); K = np.zeros((3, 49, 7, 3))
    st = np.zeros(3, dtype=abi.STATS_DTYPE)
    d = abi.as_dp
    rc = lib.tsat_solve_batch(solver._h, C.byref(o), 3, 1, d(b.x0), d(b.xf), d(b.Btab), abi.as_ip(b.btab_idx), d(b.tau0),
                              d(b.dtau), d(b.dt), d(b.Jmat), d(b.Qd), d(b.Qfd), d(b.Rd), d(b.ulo), d(b.uhi), d(b.U0),
                              d(X), d(U), d(K), st.ctypes.data_as(C.c_void_p))
    assert rc == 0
    ref = ol.solve_batch(b, oo)
    assert_same_solution(ref, dict(X=X, U=U, K=K, stats=st))
    k_close(ref, dict(K=K))


def test_gpu_trajopt_surface_end_to_end(pkg, ol):
    """the reference script's call sequence (src/TortoiseSat.jl:145-203) on the GPU"""
    ss, to = pkg.slew_setup, pkg.trajopt
    N = 120
    ref_b = ss.workload_single_slew(N=N)
    n, m = 8, 3
    x0, xf = np.r_[ref_b.x0[0], 0.0], np.r_[ref_b.xf[0], 1.0]
    model_d = to.rk3(to.Model(to.DerivFunction(ss.INERTIA["1P"], ref_b.Btab[0]), n, m))
    Q = np.zeros((n, n)); Qf = np.zeros((n, n))
    Q[:7, :7] = np.diag(ref_b.Qd[0]); Qf[:7, :7] = np.diag(ref_b.Qfd[0])
    obj = to.LQRObjective(Q, np.diag(ref_b.Rd[0]), Qf, xf, N)
    constraints = to.Constraints(N)
    for k in range(1, N):
        constraints[k] += to.BoundConstraint(n, m, u_max=1, u_min=-1)
    constraints[N] += to.goal_constraint(xf)
    sat = to.Problem(model_d, obj, constraints=constraints, x0=x0, xf=xf, N=N, dt=0.2)
    to.initial_controls_(sat, np.zeros((3, N + 1)))
    opts_al = to.AugmentedLagrangianSolverOptions()
    opts_al.opts_uncon.iterations = 12
    opts_al.iterations = 4
    s = to.AugmentedLagrangianSolver(sat, opts_al)
    to.solve_(sat, s)
    s.close()
    ref = ol.solve_batch(ref_b, oracle_options(ol, max_outer=4, max_inner=12))
    assert sat.X.shape == (7, N) and sat.U.shape == (3, N - 1) and sat.K.shape == (3, 7, N - 1)
    np.testing.assert_allclose(sat.X.T, ref["X"][0], atol=1e-9)
    np.testing.assert_allclose(sat.U.T, ref["U"][0], atol=1e-9)
    np.testing.assert_allclose(sat.K.transpose(2, 1, 0), ref["K"][0], rtol=1e-8, atol=1e-7)


# ----------------------------------------------------------------------------------------------- full size
def test_gpu_full_size_properties(pkg, ol, solver):
    """BASELINE.json configs[1] in full (1024 x 1000 knots, 5 x 10): properties that need no oracle run —
    determinism, shard-concatenation, dynamic consistency of the returned rollout, cost/violation recomputation."""
    ss = pkg.slew_setup
    T, N = 1024, 1000
    b = ss.workload_monte_carlo(T=T, N=N)
    o = oracle_options(ol, max_outer=5, max_inner=10, dj_counter_limit=1)
    g = gpu_solve(pkg, solver, b, o)
    st = g["stats"]
    assert np.all(np.isin(st["status"], (0, 1))) and np.all(st["inner_iters"] >= 5) and np.all(st["inner_iters"] <= 50)
    assert np.all(np.isfinite(g["X"])) and np.all(np.isfinite(g["U"])) and np.all(np.isfinite(g["K"]))
    # (1) bitwise deterministic re-run
    g2 = gpu_solve(pkg, solver, b, o)
    assert np.array_equal(g["X"], g2["X"]) and np.array_equal(g["U"], g2["U"]) and np.array_equal(st, g2["stats"])
    # (2) a shard solved alone equals the same rows of the full batch, bit for bit (what the multi-GPU sweep relies on)
    gs = gpu_solve(pkg, solver, b.slice(512, 640), o)
    assert np.array_equal(gs["X"], g["X"][512:640]) and np.array_equal(gs["U"], g["U"][512:640])
    # (3) x_{k+1} = rk3(x_k, u_k) along every trajectory (vectorised NumPy restatement of the step)
    X, U = g["X"], g["U"]
    assert np.array_equal(X[:, 0], b.x0)
    xn = helpers.rk3_numpy(X[:, :-1], U, b)
    assert np.max(np.abs(xn - X[:, 1:])) < 1e-12
    # (4) reported cost / violation match a recomputation from the returned arrays
    e = X - b.xf[:, None, :]
    cost = 0.5 * np.sum(b.Qd[:, None] * e[:, :-1] ** 2, axis=(1, 2)) + 0.5 * np.sum(b.Rd[:, None] * U ** 2, axis=(1, 2)) \
        + 0.5 * np.sum(b.Qfd * e[:, -1] ** 2, axis=1)
    np.testing.assert_allclose(st["cost"], cost, rtol=1e-11)
    cmax = np.maximum(np.maximum(np.max(U - b.uhi[:, None], axis=(1, 2)), np.max(b.ulo[:, None] - U, axis=(1, 2))),
                      np.max(np.abs(e[:, -1]), axis=1))
    np.testing.assert_allclose(st["c_max"], np.maximum(cmax, 0.0), rtol=1e-11, atol=1e-14)
    assert np.array_equal(st["status"] == 0, st["c_max"] < 1e-3)
    # (5) the solve did its job: terminal attitude error far below the initial one
    assert np.median(np.linalg.norm(e[:, -1, 3:], axis=1)) < 0.05 * np.median(np.linalg.norm(e[:, 0, 3:], axis=1))
    # (6) spot-check 8 random trajectories of the full batch against the oracle
    idx = np.random.default_rng(0).choice(T, 8, replace=False)
    for i in idx:
        r = ol.solve_batch(b.slice(int(i), int(i) + 1), o)
        assert np.max(np.abs(r["X"][0] - X[i])) < 1e-9 and np.max(np.abs(r["U"][0] - U[i])) < 1e-9
        assert r["stats"]["inner_iters"][0] == st["inner_iters"][i]


def test_gpu_example_script_runs(pkg):
    """examples/single_slew.py: the reference's single-slew script end to end (field table, horizon, solve, tracking)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("single_slew", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "single_slew.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(verbose=False)
    assert out["N"] > 100 and out["stats"]["status"] in (0, 1) and np.all(np.isfinite(out["X"])) and np.all(np.abs(out["U"]) <= 1 + 1e-3)
