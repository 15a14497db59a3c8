"""Field-table generation: orbit + IGRF-12 + frames (src/magnetic_toolbox.jl:33-106)."""
import numpy as np
import pytest

import refmath_igrf as ri

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
