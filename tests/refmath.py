"""Independent NumPy transcription of the reference's Julia text for the hot-path pieces that ARE in the
reference. Written from the .jl files line by line — deliberately naive
(allocating, matrix-form) and sharing no code with oracle/ or the package — so it can pin the oracle.
Each function cites the lines it transcribes."""
import numpy as np


def qmult(q1, q2):
    # src/DerivFunction.jl:54-56
    q1 = np.asarray(q1, float); q2 = np.asarray(q2, float)
    return np.concatenate([[q1[0] * q2[0] - q1[1:4] @ q2[1:4]],
                           q1[0] * q2[1:4] + q2[0] * q1[1:4] + np.cross(q1[1:4], q2[1:4])])


def qrot(q, r):
    # src/DerivFunction.jl:50-52
    q = np.asarray(q, float); r = np.asarray(r, float)
    return r + 2 * np.cross(q[1:4], np.cross(q[1:4], r) + q[0] * r)


def q_inv(q):
    # src/attitude_controller.jl:164-166
    return np.concatenate([[q[0]], -np.asarray(q[1:4], float)])


def hat(x):
    # src/attitude_controller.jl:172-176
    return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]], float)


def gmat(q):
    # src/attitude_controller.jl:69  Gk = [-vk'; sk*I + hat(vk)]
    return np.vstack([-np.asarray(q[1:4], float)[None, :], q[0] * np.eye(3) + hat(q[1:4])])


def deriv_function(x, u, B_ECI, N, J, tf, t0):
    # src/DerivFunction.jl:1-48 with the globals (B_ECI, N, p.J, tf, t0) as arguments
    omega = x[0:3]
    q = x[3:7] / np.linalg.norm(x[3:7])
    t = x[7]
    q_dot = 0.5 * qmult(q, np.concatenate([[0.0], omega]))
    B_B = qrot(q, B_ECI[int(np.floor(t * N + 1)) - 1, :])      # Julia 1-based row
    tau_c = np.cross(u[0:3] * 1.0e-2, B_B)
    omega_dot = np.linalg.inv(J) @ (tau_c - np.cross(omega, J @ omega))
    return np.concatenate([omega_dot, q_dot, [1.0 / (tf - t0)]])


def attitude_dynamics(x, u, B_B, J):
    # src/attitude_dynamics.jl:2-24
    omega = x[0:3]
    q = x[3:7] / np.linalg.norm(x[3:7])
    q_dot = 0.5 * qmult(q, np.concatenate([[0.0], omega]))
    tau_c = np.cross(u[0:3], B_B)
    omega_dot = np.linalg.inv(J) @ (tau_c - np.cross(omega, J @ omega))
    return np.concatenate([omega_dot, q_dot])


def rk3(f, dt):
    # src/attitude_controller.jl:178-187
    def fd(x, u):
        k1 = f(x, u) * dt
        k2 = f(x + k1 / 2, u) * dt
        k3 = f(x - k1 + 2 * k2, u) * dt
        return x + (k1 + 4 * k2 + k3) / 6
    return fd


def rk4(f, dt):
    # src/attitude_controller.jl:122-132
    def fd(x, u):
        k1 = f(x, u) * dt
        k2 = f(x + k1 / 2, u) * dt
        k3 = f(x + k2 / 2, u) * dt
        k4 = f(x + k3, u) * dt
        return x + (k1 + 2 * k2 + 2 * k3 + k4) / 6
    return fd


def reduce_error_state(Aq, Bq, qk, qn):
    # src/attitude_controller.jl:59-81
    Gk, Gn = gmat(qk), gmat(qn)
    perm_Gn = np.zeros((6, 7)); perm_Gk = np.zeros((7, 6))
    perm_Gn[0:3, 0:3] = np.eye(3); perm_Gn[3:6, 3:7] = Gn.T
    perm_Gk[0:3, 0:3] = np.eye(3); perm_Gk[3:7, 3:6] = Gk
    return perm_Gn @ Aq @ perm_Gk, perm_Gn @ Bq


def tvlqr_riccati(A, B, Q, R, Qf):
    # src/attitude_controller.jl:83-92 ; A (N-1,6,6), B (N-1,6,3)
    N = A.shape[0] + 1
    S = Qf.copy()
    K = np.zeros((N - 1, 3, 6))
    for k in range(N - 2, -1, -1):
        K[k] = np.linalg.inv(R + B[k].T @ S @ B[k]) @ (B[k].T @ S @ A[k])
        Acl = A[k] - B[k] @ K[k]
        S = Q + K[k].T @ R @ K[k] + Acl.T @ S @ Acl
    return K


def quaternion_error(X1, X2):
    # src/quaternion_toolbox.jl:58-75 (7-vector; slot 7 stays zero)
    dx = np.zeros(7)
    dx[0:3] = X1[0:3] - X2[0:3]
    q_e = qmult(q_inv(X2[3:7]), X1[3:7])
    dx[3:6] = q_e[1:4] / (1 + q_e[0])
    return dx


def quaternion_expansion(Q, qlin, x):
    # src/quaternion_toolbox.jl:15-36 on the 7-state (time row/col dropped)
    G = gmat(x[3:7])
    perm_Gn = np.zeros((6, 7)); perm_Gk = np.zeros((7, 6))
    perm_Gn[0:3, 0:3] = np.eye(3); perm_Gn[3:6, 3:7] = G.T
    perm_Gk[0:3, 0:3] = np.eye(3); perm_Gk[3:7, 3:6] = G
    return perm_Gn @ Q @ perm_Gk, perm_Gn @ (Q @ x + qlin)


def eigen_axis_slew(x0, xf, t):
    # src/eigen_axis_slew.jl:1-38 (line 16 reads only the first 4 entries of [q2; -q2[2:4]])
    q1 = x0[3:7]; q2 = xf[3:7]
    q_e = qmult(np.concatenate([q2, -q2[1:4]])[0:4], q1)
    theta_f = 2 * np.arccos(q_e[0])
    axis = -q_e[1:4] / np.sin(theta_f / 2)
    alpha = np.pi / t[-1]
    theta = theta_f * 0.5 * (np.ones(len(t)) - np.cos(alpha * t))
    d_theta = list(np.diff(theta) / (t[1] - t[0]))
    d_theta.append(d_theta[-1])
    w = np.zeros((len(t), 3)); qg = np.zeros((len(t), 4))
    for i in range(len(t)):
        w[i] = d_theta[i] * axis
        qg[i] = qmult(q1, np.concatenate([[np.cos(theta[i] / 2)], axis * np.sin(theta[i] / 2)]))
    return w, qg
