"""CPU tier: the packed and fp32 builds of the solve kernel on the lane emulator (tests/emu, the very source hipcc compiles).

packed (tsat_packed.hpp, -DTSAT_PACKED): PK_G trajectories per wavefront — bit-identical to the one-trajectory builds, whatever
the group is made of (partial last group, ragged horizons, trajectories that finish early, fail or diverge next to healthy ones).
mixed precision (-DTSAT_JAC32, options.precision = 32): float linearisation (Jacobian lanes and knot records), everything else
double — against the fp64 oracle at the bar of SURVEY.md §8(d) for fp32 (1e-3 + status agreement); in fact far inside it, with
the oracle's iteration counts.
"""
import numpy as np
import pytest

from conftest import assert_same_solution, oracle_options


def _same_bits(a, b):
    for k in ("X", "U", "K"):
        assert np.array_equal(a[k], b[k]), k
    for f in a["stats"].dtype.names:      # n_forward counts executed sweeps: a packed sweep carries PK_C candidates, not 64
        if f != "n_forward":
            assert np.array_equal(a["stats"][f], b["stats"][f]), f


@pytest.mark.parametrize("T,N,es,integ", [(1, 30, 0, 3), (3, 41, 1, 3), (4, 18, 0, 4), (5, 35, 1, 3), (9, 17, 1, 4)])
def test_packed_build_is_the_same_solve(pkg, ol, emu, emu_packed, T, N, es, integ):
    """full and partial groups around the 16-knot pass boundaries: equal to the oracle, bit-identical to the wide build"""
    b = pkg.slew_setup.workload_monte_carlo(T=T, N=N, seed=500 + 7 * T + N, random_orbit=(T == 5))
    o = oracle_options(ol, max_outer=2, max_inner=4, dj_counter_limit=1, error_state=es, integrator=integ)
    wide, packed = emu.solve(b, o), emu_packed.solve(b, o)
    assert_same_solution(ol.solve_batch(b, o), packed)
    _same_bits(wide, packed)


@pytest.mark.parametrize("T,N,es,integ", [(9, 18, 0, 4), (11, 34, 1, 3)])
def test_packed8_build_is_the_same_solve(pkg, ol, emu, emu_packed8, T, N, es, integ):
    """eight trajectories per wavefront (two backward passes of four, one single-buffered forward chunk), ragged horizons, partial groups"""
    b = pkg.slew_setup.workload_monte_carlo(T=T, N=N, seed=500 + 7 * T + N, random_orbit=(T == 11))
    b.n_knots = np.array([N, 2, max(3, N // 2), N - 1, 3, N, N - 2, 7, N, N, 5, N, 4, N, N, 9, N][:T], dtype=np.int32)
    o = oracle_options(ol, max_outer=2, max_inner=4, dj_counter_limit=1, error_state=es, integrator=integ)
    wide, packed = emu.solve(b, o), emu_packed8.solve(b, o)
    assert_same_solution(ol.solve_batch(b, o), packed)
    _same_bits(wide, packed)


@pytest.mark.parametrize("which,T,N,es,integ", [("emu_packed4w", 11, 34, 1, 3), ("emu_packed8w", 11, 34, 1, 3), ("emu_packed8w", 19, 40, 0, 4), ("emu_packed16w", 19, 40, 1, 3),
                                                ("emu_packed16w", 35, 21, 0, 3)])
def test_one_wavefront_per_simd_builds_are_the_same_solve(pkg, ol, request, emu, which, T, N, es, integ):
    """four, eight and sixteen trajectories per wavefront at one wavefront per SIMD (40 KB of LDS: a twelve-knot record ring whose slots
    are taken modulo twelve, double-buffered forward chunks; sixteen: four line-search candidates per trajectory and sweep, so
    deeper searches go on in further sweeps): ragged horizons, partial last wavefronts, a diverging roll-out and an indefinite
    Quu next to healthy trajectories — bit-identical to the wide build, equal to the oracle"""
    e = request.getfixturevalue(which)
    assert 20480 < e.lib.emu_lds_bytes() <= 40960            # one wavefront per SIMD
    b = pkg.slew_setup.workload_monte_carlo(T=T, N=N, seed=500 + 7 * T + N, random_orbit=(T == 11))
    nk = [N, 2, max(3, N // 2), N - 1, 3, N, N - 2, 7, N, N, 5, N, 4, N, N, 9, N, N, 6, N, N - 1, 3, N, N, N // 2, N, 8, N, N, 2, N, N, N, 5, N]
    b.n_knots = np.array(nk[:T], dtype=np.int32)
    kw = {}
    if T == 19:
        b.U0[1] = 1e12
        b.Rd[3] = -1e-4
        kw = dict(reg_max=1e-2)
    o = oracle_options(ol, max_outer=2, max_inner=4, dj_counter_limit=1, error_state=es, integrator=integ, **kw)
    wide, packed = emu.solve(b, o), e.solve(b, o)
    _same_bits(wide, packed)
    if T != 19:
        assert_same_solution(ol.solve_batch(b, o), packed)
    else:
        assert np.array_equal(ol.solve_batch(b, o)["stats"]["status"], packed["stats"]["status"])
    if which == "emu_packed16w":          # a search deeper than the four candidates of a sweep
        b2 = pkg.slew_setup.workload_monte_carlo(T=5, N=36, seed=9)
        o3 = oracle_options(ol, max_outer=2, max_inner=3, ls_lower=0.999999, ls_upper=1.000001, max_linesearch=20)
        _same_bits(emu.solve(b2, o3), e.solve(b2, o3))


@pytest.mark.parametrize("es", [0, 1])
def test_packed_build_ragged_groups(pkg, ol, emu, emu_packed, es):
    """every trajectory of a group has its own horizon (t_total[i] = 0:0.2:t_final[i], src/monte_carlo.jl:140-145)"""
    ss = pkg.slew_setup
    b = ss.workload_monte_carlo(T=6, N=47, seed=77)
    b.n_knots = np.array([47, 2, 19, 33, 3, 46], dtype=np.int32)
    o = oracle_options(ol, max_outer=2, max_inner=4, dj_counter_limit=1, error_state=es)
    wide, packed = emu.solve(b, o), emu_packed.solve(b, o)
    assert_same_solution(ol.solve_batch(b, o), packed)
    _same_bits(wide, packed)
    for t, n in enumerate(b.n_knots):
        assert np.all(packed["X"][t, n:] == 0) and np.all(packed["U"][t, n - 1:] == 0)


def test_packed_build_mixed_fates_in_one_group(pkg, ol, emu, emu_packed):
    """a diverging rollout, a regularisation failure (with restarts on the way), a long line search and healthy neighbours
    inside the same wavefront: nobody's arithmetic depends on the others"""
    ss, abi = pkg.slew_setup, pkg._abi
    b = ss.workload_monte_carlo(T=4, N=36, seed=9)
    b.U0[1] = 1e12                   # DIVERGED
    b.Rd[2] = -1e-4                  # Quu indefinite: restarts, then REG_FAIL (reg_max small)
    o = oracle_options(ol, max_outer=3, max_inner=4, reg_max=1e-6)
    wide, packed, ref = emu.solve(b, o), emu_packed.solve(b, o), ol.solve_batch(b, o)
    assert ref["stats"]["status"][1] == abi.TSAT_DIVERGED and ref["stats"]["status"][2] == abi.TSAT_REG_FAIL
    assert np.array_equal(ref["stats"]["status"], packed["stats"]["status"])
    _same_bits(wide, packed)
    for t in (0, 3):
        assert np.max(np.abs(ref["X"][t] - packed["X"][t])) < 1e-9
    # restarts that succeed: negative R, large reg_max
    b2 = ss.workload_monte_carlo(T=3, N=36, seed=10)
    b2.Rd[1] = -1e-4
    o2 = oracle_options(ol, max_outer=2, max_inner=3)
    w2, p2, r2 = emu.solve(b2, o2), emu_packed.solve(b2, o2), ol.solve_batch(b2, o2)
    assert r2["stats"]["bp_restarts"][1] > 0
    assert_same_solution(r2, p2)
    _same_bits(w2, p2)
    # deep line search (forced by a tiny acceptance window) and a failed one
    o3 = oracle_options(ol, max_outer=2, max_inner=3, ls_lower=0.999999, ls_upper=1.000001, max_linesearch=20)
    w3, p3 = emu.solve(b, o3), emu_packed.solve(b, o3)
    _same_bits(w3, p3)


@pytest.mark.parametrize("which,T,N,at", [("emu_packed", 11, 33, 6), ("emu_packed", 7, 20, 100), ("emu_packed8", 19, 21, 9),
                                          ("emu_packed_mixed", 10, 30, 5), ("emu_packed4w", 9, 21, 4), ("emu_packed8w", 19, 21, 9), ("emu_packed16w", 37, 21, 12),
                                          ("emu_packed16w_mixed", 21, 30, 6)])
def test_packed_endgame_is_the_same_solve(pkg, ol, request, monkeypatch, which, T, N, at):
    """tsat_set_endgame: once `at` trajectories are left, the wavefronts park theirs and a second launch finishes each on a
    wavefront of its own (tsat_resume_kernel_packed). Which ones get parked depends on how the waves were scheduled — here: on
    the emulator's threads —, the results do not: bit-identical to the launch without it, for a spread of iteration counts
    (ragged horizons, a diverging rollout, a regularisation failure), also when everything is parked at once (at >= T)."""
    e = request.getfixturevalue(which)
    b = pkg.slew_setup.workload_monte_carlo(T=T, N=N, seed=900 + T, random_orbit=True)
    b.n_knots = np.array(([N, 4, N - 3, N, 9, N, N // 2, N, 2, N, N - 1, N, N, 6, N, N, 3, N, N] * 2)[:T], dtype=np.int32)
    b.U0[1] = 1e12
    b.Rd[3] = -1e-4
    o = oracle_options(ol, max_outer=3, max_inner=5, dj_counter_limit=1, error_state=T % 2, reg_max=1e-2)
    if which.endswith("_mixed"):
        o.precision = 32
    monkeypatch.delenv("TSAT_EMU_SUSPEND_AT", raising=False)
    plain = e.solve(b, o)
    monkeypatch.setenv("TSAT_EMU_SUSPEND_AT", str(at))
    parked = e.solve(b, o)
    n_parked = e.lib.emu_parked()
    assert 0 <= n_parked <= min(at, T)          # (none, if the last `at` were all inside their final iteration when the count fell)
    if at >= T:
        assert n_parked == int(np.sum(plain["stats"]["status"] != pkg._abi.TSAT_DIVERGED))   # all that survive their rollout
    _same_bits(plain, parked)


def test_mixed_build_against_the_fp64_oracle(pkg, ol, emu_mixed):
    """precision = 32 (float linearisation): same statuses, iteration and line-search counts as the fp64 oracle; |dX| < 1e-3,
    |dU| < 1e-3 of the control scale (fp32 bar 1e-3 + status agreement) — measured here: |dX| 1e-7, |dU| 1e-4 of the scale"""
    assert emu_mixed.lib.emu_lds_bytes() <= 20480
    for es in (0, 1):
        b = pkg.slew_setup.workload_monte_carlo(T=3, N=60, seed=40 + es)
        o = oracle_options(ol, max_outer=2, max_inner=3, dj_counter_limit=1, error_state=es)
        o.precision = 32
        ref, got = ol.solve_batch(b, o), emu_mixed.solve(b, o)
        for f in ("status", "inner_iters", "ls_trials"):
            assert np.array_equal(ref["stats"][f], got["stats"][f]), f
        scale = np.maximum(1.0, np.max(np.abs(ref["U"]), axis=(1, 2)))
        dX = np.max(np.abs(ref["X"] - got["X"])); dU = np.max(np.max(np.abs(ref["U"] - got["U"]), axis=(1, 2)) / scale)
        print(f"[mixed emu es={es}] max|dX| {dX:.2e}  max|dU|/scale {dU:.2e}")
        assert dX < 1e-5 and dU < 1e-3
        np.testing.assert_allclose(got["stats"]["cost"], ref["stats"]["cost"], rtol=1e-6)


@pytest.mark.parametrize("which,T,N,es", [("emu_packed_mixed", 5, 37, 1), ("emu_packed_mixed", 4, 26, 0), ("emu_packed16w_mixed", 5, 37, 1), ("emu_packed16w_mixed", 18, 26, 0)])
def test_mixed_packed_build_is_the_mixed_solve(pkg, ol, request, emu_mixed, which, T, N, es):
    """precision = 32 on large batches takes the packed mixed builds: bit-identical to the one-trajectory mixed build (sixteen per
    wavefront at one wavefront per SIMD: all sixteen knots of a pass in the record ring, nothing through the workspace)"""
    emu_packed_mixed = request.getfixturevalue(which)
    assert emu_packed_mixed.lib.emu_lds_bytes() <= (20480 if which == "emu_packed_mixed" else 40960)
    b = pkg.slew_setup.workload_monte_carlo(T=T, N=N, seed=900 + N)
    b.n_knots = np.array(([N, max(2, N // 3), N - 1, N, 5] * 4)[:T], dtype=np.int32)
    o = oracle_options(ol, max_outer=3, max_inner=4, dj_counter_limit=1, error_state=es)
    o.precision = 32
    _same_bits(emu_mixed.solve(b, o), emu_packed_mixed.solve(b, o))
