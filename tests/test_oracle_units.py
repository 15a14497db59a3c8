"""Pins the CPU oracle's building blocks against an independent NumPy transcription of
the reference's own Julia text (tests/refmath.py), analytic invariants, and SciPy's DARE. The reference has no
tests or golden vectors of its own, so this is the strongest pin available: "parity unpinned"."""
import numpy as np
import pytest
import scipy.linalg

import refmath as rm

RNG = np.random.default_rng(12345)
J_FULL = np.array([[2.0e-3, 1.0e-4, -2.0e-4], [1.0e-4, 1.5e-3, 3.0e-4], [-2.0e-4, 3.0e-4, 2.5e-3]])
J_1U = np.diag([0.00125] * 3)


def rq(unit=True):
    q = RNG.standard_normal(4)
    return q / np.linalg.norm(q) if unit else q * 1.3


# ----------------------------------------------------------------------------- A1 quaternion algebra
def test_qmult_qrot_qinv_hat_gmat_match_reference_text(ol):
    for _ in range(20):
        q1, q2, r = rq(False), rq(False), RNG.standard_normal(3)
        np.testing.assert_allclose(ol.qmult(q1, q2), rm.qmult(q1, q2), rtol=0, atol=1e-15)
        np.testing.assert_allclose(ol.qrot(q1, r), rm.qrot(q1, r), rtol=0, atol=1e-14)
        np.testing.assert_array_equal(ol.qinv(q1), rm.q_inv(q1))
        np.testing.assert_array_equal(ol.hat(r), rm.hat(r))
        np.testing.assert_array_equal(ol.gmat(q1), rm.gmat(q1))


def test_quaternion_identities(ol):
    for _ in range(20):
        q, p, r, s = rq(False), rq(False), RNG.standard_normal(3), RNG.standard_normal(3)
        # norm is multiplicative
        assert abs(np.linalg.norm(ol.qmult(q, p)) - np.linalg.norm(q) * np.linalg.norm(p)) < 1e-13
        # hat(a) b = a x b
        np.testing.assert_allclose(ol.hat(r) @ s, np.cross(r, s), atol=1e-15)
        qu = q / np.linalg.norm(q)
        # qrot(q, r) = vec(q (x) [0;r] (x) q^-1) for unit q
        full = ol.qmult(ol.qmult(qu, np.r_[0.0, r]), ol.qinv(qu))
        np.testing.assert_allclose(ol.qrot(qu, r), full[1:], atol=1e-14)
        assert abs(full[0]) < 1e-14
        # rotations preserve length; G(q)' q = 0; G'G = |q|^2 I
        assert abs(np.linalg.norm(ol.qrot(qu, r)) - np.linalg.norm(r)) < 1e-14
        G = ol.gmat(q)
        np.testing.assert_allclose(G.T @ q, 0, atol=1e-15)
        np.testing.assert_allclose(G.T @ G, (q @ q) * np.eye(3), atol=1e-14)


def test_inv3(ol):
    np.testing.assert_allclose(ol.inv3(J_FULL), np.linalg.inv(J_FULL), rtol=1e-13)


# ----------------------------------------------------------------------------- A2/A3 dynamics
def test_deriv_function_matches_reference_text(ol):
    N = 50
    B = RNG.standard_normal((2 * N, 3)) * 3e-5
    B[-1] = 0.0  # the reference leaves the last row zero (src/magnetic_toolbox.jl:73,76)
    for J in (J_1U, J_FULL):
        for _ in range(10):
            x = np.r_[RNG.standard_normal(3) * 0.1, rq(False), RNG.random() * 0.9]
            u = RNG.standard_normal(3) * 5
            got = ol.deriv8(x, u, B, N, J, 5400.0)
            ref = rm.deriv_function(x, u, B, N, J, 5400.0, 0.0)
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-16)


def test_dyn7_equals_deriv8_on_looked_up_row(ol):
    N = 40
    B = RNG.standard_normal((N, 3)) * 3e-5
    x = np.r_[RNG.standard_normal(3) * 0.1, rq(False), 0.37]
    u = RNG.standard_normal(3)
    row = int(np.floor(0.37 * N))
    np.testing.assert_array_equal(ol.dyn7(x[:7], u, B[row], J_FULL), ol.deriv8(x, u, B, N, J_FULL, 100.0)[:7])


def test_attitude_dynamics_matches_reference_text(ol):
    for _ in range(10):
        x = np.r_[RNG.standard_normal(3) * 0.1, rq(False)]
        u, BB = RNG.standard_normal(3), RNG.standard_normal(3) * 3e-5
        np.testing.assert_allclose(ol.attitude_dynamics(x, u, BB, J_FULL), rm.attitude_dynamics(x, u, BB, J_FULL),
                                   rtol=1e-12, atol=1e-16)


def test_dynamics_invariants(ol):
    x = np.r_[RNG.standard_normal(3) * 0.2, rq(False)]
    xd = ol.dyn7(x, np.zeros(3), RNG.standard_normal(3) * 3e-5, J_FULL)
    # torque free: d/dt |J w|^2 = 0 and kinetic energy constant
    w, wd = x[:3], xd[:3]
    assert abs((J_FULL @ w) @ (J_FULL @ wd)) < 1e-18
    assert abs(w @ (J_FULL @ wd)) < 1e-18
    # qdot is orthogonal to the normalised quaternion
    assert abs(xd[3:] @ (x[3:7] / np.linalg.norm(x[3:7]))) < 1e-16
    # q is normalised inside f (src/DerivFunction.jl:5): f is invariant to the quaternion's length
    x2 = x.copy(); x2[3:7] *= 1.7
    np.testing.assert_allclose(ol.dyn7(x2, np.ones(3), np.ones(3) * 1e-5, J_FULL), ol.dyn7(x, np.ones(3), np.ones(3) * 1e-5, J_FULL), rtol=1e-13, atol=1e-18)
    # u scaling: controls enter in 0.01 A m^2 (src/DerivFunction.jl:37)
    b = RNG.standard_normal(3) * 3e-5
    u = RNG.standard_normal(3)
    np.testing.assert_allclose(ol.dyn7(x, u, b, J_1U, 1e-2)[:3], ol.dyn7(x, u * 1e-2, b, J_1U, 1.0)[:3], rtol=1e-13)


# ----------------------------------------------------------------------------- A4 integrators
@pytest.mark.parametrize("integ,order", [(3, 3), (4, 4)])
def test_rk_order_on_scalar_test_equation(ol, integ, order):
    lam, T = -1.3, 1.0
    errs = []
    for n in (20, 40, 80):
        x, h = 1.0, T / n
        for _ in range(n):
            x = ol.rk_scalar(integ, lam, x, h)
        errs.append(abs(x - np.exp(lam * T)))
    rates = np.log2(np.array(errs[:-1]) / np.array(errs[1:]))
    assert np.all(np.abs(rates - order) < 0.15), rates


@pytest.mark.parametrize("integ", [3, 4])
def test_rk_step_matches_reference_tableau_with_stage_rows(ol, integ):
    # stage rows at tau, tau + dtau/2, tau + dtau: drive the NumPy rk with a time-dependent field
    dt = 0.2
    b0, b1, b2 = (RNG.standard_normal(3) * 3e-5 for _ in range(3))
    x = np.r_[RNG.standard_normal(3) * 0.1, rq(False)]
    u = RNG.standard_normal(3) * 3
    rows = {0.0: b0, 0.5: b1, 1.0: b2}

    def f8(x8, u):  # 8-state with stage time -> row, like DerivFunction's floor lookup
        c = min(rows, key=lambda k: abs(k - x8[7]))
        return np.r_[rm.attitude_dynamics(x8[:7], u * 1e-2, rm.qrot(x8[3:7] / np.linalg.norm(x8[3:7]), rows[c]), J_FULL), 1.0 / dt]

    ref = (rm.rk3 if integ == 3 else rm.rk4)(f8, dt)(np.r_[x, 0.0], u)
    got = ol.rk_step(integ, x, u, b0, b1, b2, dt, J_FULL)
    np.testing.assert_allclose(got, ref[:7], rtol=1e-12, atol=1e-15)
    assert abs(ref[7] - 1.0) < 1e-14


# ----------------------------------------------------------------------------- A5 Jacobians
@pytest.mark.parametrize("integ", [3, 4])
def test_discrete_jacobian_vs_central_differences(ol, integ):
    dt = 0.2
    b0, b1, b2 = (RNG.standard_normal(3) * 4e-5 for _ in range(3))
    x = np.r_[RNG.standard_normal(3) * 0.1, rq(False)]
    u = RNG.standard_normal(3) * 5
    A, B = ol.discrete_jacobian(integ, x, u, b0, b1, b2, dt, J_FULL)
    f = lambda x_, u_: ol.rk_step(integ, x_, u_, b0, b1, b2, dt, J_FULL)
    An, Bn = np.zeros((7, 7)), np.zeros((7, 3))
    for j in range(7):
        e = np.zeros(7); e[j] = 1e-6
        An[:, j] = (f(x + e, u) - f(x - e, u)) / 2e-6
    for j in range(3):
        e = np.zeros(3); e[j] = 1e-3
        Bn[:, j] = (f(x, u + e) - f(x, u - e)) / 2e-3
    np.testing.assert_allclose(A, An, atol=2e-9)
    np.testing.assert_allclose(B, Bn, atol=2e-9)


# ----------------------------------------------------------------------------- A6 error-state reduction
def test_error_state_reduction_and_hooks_match_reference_text(ol):
    for _ in range(5):
        A, B = RNG.standard_normal((7, 7)), RNG.standard_normal((7, 3))
        qk, qn = rq(False), rq(False)
        Ah, Bh = ol.reduce_error_state(A, B, qk, qn)
        Ar, Br = rm.reduce_error_state(A, B, qk, qn)
        np.testing.assert_allclose(Ah, Ar, atol=1e-13)
        np.testing.assert_allclose(Bh, Br, atol=1e-13)
        X1, X2 = np.r_[RNG.standard_normal(3), rq()], np.r_[RNG.standard_normal(3), rq()]
        np.testing.assert_allclose(ol.quaternion_error(X1, X2), rm.quaternion_error(X1, X2), atol=1e-14)
        Q = np.diag(RNG.random(7) + 0.1)
        ql = RNG.standard_normal(7)
        Qxx, Qx = ol.quaternion_expansion(Q, ql, X1)
        Rxx, Rx = rm.quaternion_expansion(Q, ql, X1)
        np.testing.assert_allclose(Qxx, Rxx, atol=1e-13)
        np.testing.assert_allclose(Qx, Rx, atol=1e-13)
    # identical attitudes -> zero MRP
    np.testing.assert_allclose(ol.quaternion_error(X1, X1)[3:6], 0, atol=1e-16)


# ----------------------------------------------------------------------------- A7 Riccati
def test_tvlqr_riccati_matches_reference_text(ol):
    N = 30
    A = np.eye(6)[None] + 0.05 * RNG.standard_normal((N - 1, 6, 6))
    B = 0.1 * RNG.standard_normal((N - 1, 6, 3))
    Q, R, Qf = np.diag([10.0] * 6), 7.5e3 * np.eye(3) * 1e-3, np.diag([1000.0] * 6)
    np.testing.assert_allclose(ol.tvlqr_riccati(A, B, Q, R, Qf), rm.tvlqr_riccati(A, B, Q, R, Qf), rtol=1e-9, atol=1e-11)


def test_tvlqr_riccati_converges_to_dare_on_lti(ol):
    A1 = np.eye(6) + 0.02 * RNG.standard_normal((6, 6))
    B1 = 0.2 * RNG.standard_normal((6, 3))
    Q, R = np.eye(6), 0.5 * np.eye(3)
    N = 600
    K = ol.tvlqr_riccati(np.repeat(A1[None], N - 1, 0), np.repeat(B1[None], N - 1, 0), Q, R, Q)
    S = scipy.linalg.solve_discrete_are(A1, B1, Q, R)
    Kinf = np.linalg.solve(R + B1.T @ S @ B1, B1.T @ S @ A1)
    np.testing.assert_allclose(K[0], Kinf, rtol=1e-8, atol=1e-10)


def test_error_state_solver_blocks_use_the_reference_projection(ol):
    """the E(q) the error-state solve projects with is blkdiag(I3, G(q)) of src/attitude_controller.jl:69-78, and
    projecting with it reproduces the reference's reduction and cost expansion"""
    for _ in range(5):
        qk, qn = rq(False), rq(False)
        Ek, En = ol.emat(qk), ol.emat(qn)
        ref = np.zeros((7, 6)); ref[:3, :3] = np.eye(3); ref[3:, 3:] = rm.gmat(qk)
        np.testing.assert_array_equal(Ek, ref)
        A, B = RNG.standard_normal((7, 7)), RNG.standard_normal((7, 3))
        Ar, Br = rm.reduce_error_state(A, B, qk, qn)
        np.testing.assert_allclose(En.T @ A @ Ek, Ar, atol=1e-13)
        np.testing.assert_allclose(En.T @ B, Br, atol=1e-13)
        Q = np.diag(RNG.random(7) + 0.1)
        x = np.r_[RNG.standard_normal(3), qk]
        Rxx, Rx = rm.quaternion_expansion(Q, -Q @ np.ones(7), x)
        np.testing.assert_allclose(Ek.T @ Q @ Ek, Rxx, atol=1e-13)
        np.testing.assert_allclose(Ek.T @ (Q @ (x - np.ones(7))), Rx, atol=1e-13)
