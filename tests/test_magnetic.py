"""Field-table generation: orbit + IGRF-12 + frames (src/magnetic_toolbox.jl:33-106)."""
import numpy as np
import pytest

import refmath_igrf as ri
import refmath_igrf_syn as rsyn

KEP = np.array([[0.0, 6771.0, 96.6, 30.0, 0.0, 40.0],       # SSO of the Monte-Carlo (src/monte_carlo.jl:122-127)
                [0.0, 6578.0, 96.0, 0.0, 0.0, 90.0],        # the single slew (src/TortoiseSat.jl:35-42)
                [0.02, 7000.0, 51.6, 120.0, 33.0, 250.0]])  # eccentric


def test_igrf_transcription_is_minus_grad_of_the_potential():
    """pins the transcription itself: B = -grad V by central differences of the Schmidt-normalised potential"""
    def V(r, th, ph):
        P = ri.legendre_schmidt(th, 13)
        a, kg, kh, s = 6371.2, 0, 0, 0.0
        for n in range(1, 14):
            for m in range(0, n + 1):
                g = ri.G2015[kg] + ri.GSV[kg] * 4; kg += 1
                h = 0.0
                if m > 0:
                    h = ri.H2015[kh] + ri.HSV[kh] * 4; kh += 1
                s += a * (a / r) ** (n + 1) * (g * np.cos(m * ph) + h * np.sin(m * ph)) * P[n, m]
        return s
    for lat, lon in ((0.7, -1.1), (-1.2, 2.9), (0.05, 0.4)):
        r, th, ph, e = 6771.0, np.pi / 2 - lat, lon % (2 * np.pi), 1e-5
        Br = -(V(r + e, th, ph) - V(r - e, th, ph)) / (2 * e)
        Bt = -(V(r, th + e, ph) - V(r, th - e, ph)) / (2 * e) / r
        Bp = -(V(r, th, ph + e) - V(r, th, ph - e)) / (2 * e) / (r * np.sin(th))
        np.testing.assert_allclose(ri.igrf12(2019, r * 1000, lat, lon), [-Bt, Bp, -Br], rtol=2e-6, atol=5e-3)
    # leading-dipole sanity (src/monte_carlo.jl:96)
    b = np.linalg.norm(ri.igrf12(2019, 6771e3, 0.6, 0.3)) / 1e9
    assert abs(b / (3.12e-5 * (6371 / 6771) ** 3 * np.sqrt(1 + 3 * np.sin(0.6) ** 2)) - 1) < 0.1


def test_oracle_matches_reference_text(ol):
    for lat, lon in ((0.7, -1.1), (-1.2, 2.9), (0.0, 0.0), (np.pi / 2, 0.3)):
        np.testing.assert_allclose(ol.igrf12(2019, 6771e3, lat, lon), ri.igrf12(2019, 6771e3, lat, lon), rtol=1e-11, atol=1e-8)
    for k in KEP:
        r, v = ol.kep_eci(k, 12.0, 3.986004418e5)
        rr, vr = ri.kep_ECI(k, 12.0, 3.986004418e5)
        np.testing.assert_allclose(r, rr, rtol=1e-13, atol=1e-9)
        np.testing.assert_allclose(v, vr, rtol=1e-13, atol=1e-12)
    B, pos = ol.btable_batch(KEP, 0.0, 900.0, 60)
    for t in range(len(KEP)):
        Br, pr = ri.magnetic_simulation(KEP[t], 0.0, 900.0, 60, 58155.0, 3.986004418e5, 6771.0)
        np.testing.assert_allclose(pos[t], pr, rtol=0, atol=1e-9)
        np.testing.assert_allclose(B[t], Br, rtol=0, atol=1e-14)   # 3e-10 relative; the two sum the 195 harmonics in different orders
        assert np.all(B[t, -1] == 0) and 1.5e-5 < np.linalg.norm(B[t, 0]) < 6e-5


def _syn_table(kep, t0, tf, N):
    """magnetic_simulation (src/magnetic_toolbox.jl:33-106) with the field evaluated by the reference's second algorithm"""
    orig = ri.igrf12
    ri.igrf12 = rsyn.igrf12_geocentric
    try:
        return ri.magnetic_simulation(kep, t0, tf, N, 58155.0, 3.986004418e5, 6771.0)[0]
    finally:
        ri.igrf12 = orig


def test_reference_held_anchor_igrf12_vs_igrf12syn(ol):
    """The one numerical statement the reference makes about itself: igrf12 agrees with igrf12syn to about 0.01 nT
    (src/igrf.jl:283-287). igrf12syn (src/igrf.jl:335-534) and its coefficient array gh_igrf12
    (src/igrf12syn_coefs.jl:38-474, data fixture tests/golden/igrf12syn_gh.npz) share nothing with the oracle's igrf12 or
    with the G/H table the kernels read, so this pins the field model to something this repository did not write."""
    rng = np.random.default_rng(5)
    worst = 0.0
    for _ in range(200):
        lat, lon = np.arcsin(rng.uniform(-1, 1)), rng.uniform(-np.pi, np.pi)
        r, date = rng.uniform(6600e3, 7400e3), rng.uniform(2015.0, 2019.99)
        worst = max(worst, float(np.max(np.abs(ol.igrf12(date, r, lat, lon) - rsyn.igrf12_geocentric(date, r, lat, lon)))))
    assert worst < 1e-2, worst                       # the reference's own bar, nT (field ~ 2-5e4 nT)
    assert worst < 1e-6, worst                       # what is actually observed: rounding only
    # the transcription of the first algorithm agrees with the second one as well
    assert np.max(np.abs(ri.igrf12(2019, 6771e3, 0.7, -1.1) - rsyn.igrf12_geocentric(2019, 6771e3, 0.7, -1.1))) < 1e-6
    # secular-variation block of the flat table = the SV column the kernels use (epoch 2015 + 4 years = 2019.0)
    b15, b19 = rsyn.igrf12_geocentric(2015, 6771e3, 0.3, 1.0), rsyn.igrf12_geocentric(2019, 6771e3, 0.3, 1.0)
    sv = np.array(rsyn.igrf12syn(1, 2017, 2, 6771.0, 90 - np.degrees(0.3), np.degrees(1.0))[:3])
    np.testing.assert_allclose(b19 - b15, 4 * sv, rtol=1e-12, atol=1e-9)
    # a whole field table through the second algorithm; the reference's bar is 0.01 nT = 1e-11 T, asserted: 1e-15 T (rounding)
    B, _ = ol.btable_batch(KEP[:1], 0.0, 600.0, 40)
    assert np.max(np.abs(B[0] - _syn_table(KEP[0], 0.0, 600.0, 40))) < 1e-15


def test_emulated_kernel_matches_oracle(ol, emu):
    B, pos = ol.btable_batch(KEP, [0.0, 5.0, 0.0], [300.0, 400.0, 350.0], 40)
    Bg, pg = emu.btable(KEP, [0.0, 5.0, 0.0], [300.0, 400.0, 350.0], 40)
    assert np.max(np.abs(pg - pos)) < 1e-8
    assert np.max(np.abs(Bg - B)) < 1e-9 * np.max(np.abs(B))
    assert np.all(Bg[:, -1] == 0)


@pytest.mark.gpu
def test_gpu_tables_horizon_solve(pkg, ol):
    """the whole reference sequence on the GPU: elements -> field table -> Gramian horizon -> solve on that table"""
    to, mg, hz, ss = pkg.trajopt, pkg.magnetic, pkg.horizon, pkg.slew_setup
    rng = np.random.default_rng(2)
    T, Nc = 8, 5000
    kep = np.tile(KEP[0], (T, 1))
    kep[:, 3] = rng.random(T) * 360
    kep[:, 5] = rng.random(T) * 360
    s = to.AugmentedLagrangianSolver(None, None)
    B, pos = mg.magnetic_simulation(s, kep, 0.0, 2400.0, Nc)              # coarse table (src/monte_carlo.jl:134)
    Bo, po = ol.btable_batch(kep, 0.0, 2400.0, Nc)
    assert np.max(np.abs(pos - po)) < 1e-7 and np.max(np.abs(B - Bo)) < 1e-9 * np.max(np.abs(Bo))
    # the reference-held anchor on the GPU table: first 400 rows of one orbit against the reference's second algorithm,
    # igrf12syn, the reference's bar is 0.01 nT = 1e-11 T (src/igrf.jl:283-287); asserted: 1e-15 T (rounding of the frame chain)
    Bs, _ = mg.magnetic_simulation(s, kep[:1], 0.0, 600.0, 200)
    assert np.max(np.abs(Bs[0] - _syn_table(kep[0], 0.0, 600.0, 200))) < 1e-15
    idx, _ = hz.condition_based_time(s, B[:, :Nc], 2400.0 / Nc, 30.0)     # (src/monte_carlo.jl:137-140)
    assert np.array_equal(idx, ol.horizon_batch(Bo[:, :Nc], 2400.0 / Nc, 30.0)[0]) and np.all(idx > 0)
    N = 300                                                               # fine table over the slew horizon, one row per knot
    Bf, _ = mg.magnetic_simulation(s, kep, 0.0, N * 0.2, N, want_pos=False)
    b = ss.workload_monte_carlo(T=T, N=N, seed=23)
    b.Btab, b.btab_idx, b.n_tab = np.ascontiguousarray(Bf), np.arange(T, dtype=np.int32), 2 * N
    b.dtau[:] = 1.0
    opts = to.AugmentedLagrangianSolverOptions()
    opts.iterations, opts.opts_uncon.iterations, opts.opts_uncon.dJ_counter_limit = 3, 6, 1
    s.opts = opts
    res = to.solve_(to.BatchProblem.from_arrays(b), s)
    from conftest import assert_same_solution, oracle_options
    assert_same_solution(ol.solve_batch(b, oracle_options(ol, max_outer=3, max_inner=6, dj_counter_limit=1), nthreads=8), res)
    s.close()
