"""Host-side problem setup for the slew solve — the part the reference keeps in its driver scripts.

Everything here is cheap O(N) NumPy that runs once per batch before the HIP solve; it mirrors, with
file:line citations, what src/TortoiseSat.jl and src/monte_carlo.jl do between "pick an orbit" and
"call solve!": inertia presets, the eigen-axis guess used only for Bryson weights, the weights themselves,
and the synthetic B tables / workloads of BASELINE.json's configs.

Array convention: the C ABI is column-major with the component fastest (reshape(X,7,N,T)); in NumPy that is
the C-ordered shape (T, N, 7). All builders return C-contiguous float64 arrays in that convention.
"""
from dataclasses import dataclass, field

import numpy as np

GM_EARTH = 3.986004418e14          # m^3/s^2 (src/input_parameters.jl:26, converted to SI)
R_EARTH_KM = 6371.0                # src/input_parameters.jl:27
MU_DIPOLE = 7.9e15                 # T*m^3   (src/comparison/psiaki2001_Period_LQR.jl:112)

#: inertia presets, kg*m^2 (src/input_parameters.jl:29-51)
INERTIA = {
    "1U": np.diag([0.00125, 0.00125, 0.00125]),
    "1P": np.diag([0.0001041667, 0.0001041667, 0.0001041667]),
    "3U": np.diag([0.020833, 0.020833, 0.0041666]),
}


# --------------------------------------------------------------------------------------------------
# quaternion helpers (scalar first), vectorised over leading axes
# --------------------------------------------------------------------------------------------------
def qmult(q1, q2):
    """Hamilton product, src/DerivFunction.jl:54-56."""
    q1 = np.asarray(q1, dtype=np.float64)
    q2 = np.asarray(q2, dtype=np.float64)
    s1, v1 = q1[..., :1], q1[..., 1:4]
    s2, v2 = q2[..., :1], q2[..., 1:4]
    s = s1 * s2 - np.sum(v1 * v2, axis=-1, keepdims=True)
    v = s1 * v2 + s2 * v1 + np.cross(v1, v2)
    return np.concatenate([s, v], axis=-1)


def qrot(q, r):
    """r + 2 v x (v x r + s r), src/DerivFunction.jl:50-52."""
    q = np.asarray(q, dtype=np.float64)
    r = np.asarray(r, dtype=np.float64)
    s, v = q[..., :1], q[..., 1:4]
    return r + 2.0 * np.cross(v, np.cross(v, r) + s * r)


def q_inv(q):
    """src/attitude_controller.jl:164-166."""
    q = np.asarray(q, dtype=np.float64)
    return np.concatenate([q[..., :1], -q[..., 1:4]], axis=-1)


def axis_angle_quat(axis, angle_rad):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(angle_rad / 2.0)], axis * np.sin(angle_rad / 2.0)])


# --------------------------------------------------------------------------------------------------
# eigen-axis guess + Bryson weights
# --------------------------------------------------------------------------------------------------
def eigen_axis_slew(x0, xf, t, rates_only=False):
    """Versine eigen-axis guess (omega_guess (n,3), q_guess (n,4)), src/eigen_axis_slew.jl:1-38.
    ``rates_only`` skips the quaternion history (the Bryson weights need the rates alone) and returns (omega_guess, None).

    Reproduces the reference's line 16 literally: ``qmult([q2;-q2[2:4]],q1)`` only reads the first four
    entries of its first argument, i.e. it multiplies by q2 itself, not by its conjugate.
    """
    x0 = np.asarray(x0, dtype=np.float64)
    xf = np.asarray(xf, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    q1, q2 = x0[3:7], xf[3:7]
    q_e = qmult(q2, q1)
    theta_f = 2.0 * np.arccos(np.clip(q_e[0], -1.0, 1.0))
    if not np.sin(theta_f / 2.0) > 0.0:
        raise ValueError("eigen_axis_slew: zero-angle slew (q0 and qf coincide): the guess has no rotation axis "
                         "(src/eigen_axis_slew.jl:19 divides by sin(theta/2))")
    axis = -q_e[1:4] / np.sin(theta_f / 2.0)
    alpha = np.pi / t[-1]
    theta = theta_f * 0.5 * (1.0 - np.cos(alpha * t))
    d_theta = np.diff(theta) / (t[1] - t[0])
    d_theta = np.append(d_theta, d_theta[-1])
    w_guess = d_theta[:, None] * axis[None, :]
    if rates_only:
        return w_guess, None
    dq = np.concatenate([np.cos(theta / 2.0)[:, None], axis[None, :] * np.sin(theta / 2.0)[:, None]], axis=1)
    q_guess = qmult(q1[None, :], dq)
    return w_guess, q_guess


def bryson_weights(w_guess, J, dt, alpha, beta, degenerate_rd=None, r_scale=1.0, signed_w_max=False):
    """Diagonal (Qd(7), Qfd(7), Rd(3)) per src/TortoiseSat.jl:157-168 / src/monte_carlo.jl:165-176.

    ``r_scale`` multiplies R and ``signed_w_max`` takes ``maximum(X[1:3,:])`` without the ``abs.``: the inclination-sweep
    script's variants (``R = (1/m_max^2)*.1``, src/paper_images/heatmap.jl:172; ``omega_max``, :164).

    ``tau_max`` is the *signed* maximum, as written in the reference (``maximum(J*diff(w)/dt)``). A guess without an
    acceleration sample (two knots, or a constant rate) has ``tau_max <= 0`` and the reference's ``R = 1/m_max^2`` is
    infinite: that raises here, unless ``degenerate_rd`` names the control weight to use instead.
    """
    w_max = np.max(w_guess) if signed_w_max else np.max(np.abs(w_guess))
    if not w_max > 0.0:
        raise ValueError("bryson_weights: the rate guess has no positive maximum (w_max <= 0): Q = alpha/w_max^2 is undefined"
                         if signed_w_max else
                         "bryson_weights: the rate guess is identically zero (w_max = 0): Q = alpha/w_max^2 is undefined")
    dw = np.diff(w_guess, axis=0)
    tau_max = np.max((J @ dw.T) / dt) if dw.shape[0] > 0 else 0.0
    if not tau_max > 0.0:
        if degenerate_rd is None:
            raise ValueError("bryson_weights: the guess has no positive torque sample (tau_max <= 0): R = 1/m_max^2 is "
                             "undefined; pass degenerate_rd to choose a control weight")
        Qd = np.concatenate([np.full(3, alpha / w_max**2), np.full(4, alpha * beta)])
        return Qd, 10.0 * Qd, np.full(3, float(degenerate_rd))
    m_max = tau_max / 1.0e-5 * 1.0e2
    Qd = np.concatenate([np.full(3, alpha / w_max**2), np.full(4, alpha * beta)])
    Qfd = 10.0 * Qd
    Rd = np.full(3, 1.0 / m_max**2 * r_scale)
    return Qd, Qfd, Rd


def bryson_weights_ragged(x0, xf, n_knots, t0, dt, J, alpha, beta):
    """(Qd (T,7), Qfd (T,7), Rd (T,3)) of T slews between the same attitudes whose eigen-axis guesses differ only in the
    horizon: what ``bryson_weights(eigen_axis_slew(x0, xf, t0 + dt*arange(n), rates_only=True)[0], J, dt, alpha, beta)`` gives
    for every n of ``n_knots`` (src/monte_carlo.jl:161-176 in the loop body), computed by the library's batched host routine
    ``tsat_bryson_eigen_axis_batch`` instead of one Python iteration per trial (same formulas; equal to the per-trial functions
    to rounding — tests/test_host.py)."""
    import ctypes as C

    from . import _abi
    x0 = np.asarray(x0, dtype=np.float64); xf = np.asarray(xf, dtype=np.float64)
    q_e = qmult(xf[3:7], x0[3:7])                                  # as eigen_axis_slew: the script's line 16, literally
    theta_f = 2.0 * np.arccos(np.clip(q_e[0], -1.0, 1.0))
    if not np.sin(theta_f / 2.0) > 0.0:
        raise ValueError("eigen_axis_slew: zero-angle slew (q0 and qf coincide): the guess has no rotation axis")
    axis = np.ascontiguousarray(-q_e[1:4] / np.sin(theta_f / 2.0))
    nk = np.ascontiguousarray(n_knots, dtype=np.int32)
    T = nk.shape[0]
    Qd = np.empty((T, 7)); Qfd = np.empty((T, 7)); Rd = np.empty((T, 3))
    Jrm = np.ascontiguousarray(np.asarray(J, dtype=np.float64).reshape(9))
    rc = _abi.load().tsat_bryson_eigen_axis_batch(T, _abi.as_ip(nk), float(t0), float(dt), float(theta_f), _abi.as_dp(axis),
                                                  _abi.as_dp(Jrm), float(alpha), float(beta), _abi.as_dp(Qd), _abi.as_dp(Qfd), _abi.as_dp(Rd))
    if rc == -2:
        raise ValueError("bryson_weights: a guess has no positive torque sample (tau_max <= 0): R = 1/m_max^2 is undefined")
    if rc:
        raise ValueError("tsat_bryson_eigen_axis_batch: bad arguments (every horizon needs at least three knots)")
    return Qd, Qfd, Rd


# --------------------------------------------------------------------------------------------------
# synthetic B tables (tilted-dipole surrogate until the IGRF row §8f-1 exists)
# --------------------------------------------------------------------------------------------------
def dipole_btable(n_tab, dt_row, a_km, inc_deg, raan_deg=0.0, nu_deg=0.0):
    """(n_tab, 3) field table in Tesla.

    B(t) = Rz(RAAN) * mu_f/a^3 * [cos(w0 t + nu) sin i; -cos i; 2 sin(w0 t + nu) sin i]
    — the reference's simplified dipole (src/comparison/psiaki2001_Period_LQR.jl:109-117) phased by the
    true anomaly and rotated by the ascending node. Synthetic stand-in for magnetic_simulation
    (src/magnetic_toolbox.jl:33-106); magnitudes 2-5e-5 T as with IGRF.
    """
    a = a_km * 1000.0
    w0 = np.sqrt(GM_EARTH / a**3)
    inc = np.deg2rad(inc_deg)
    ph = w0 * dt_row * np.arange(n_tab) + np.deg2rad(nu_deg)
    B = MU_DIPOLE / a**3 * np.stack([np.cos(ph) * np.sin(inc), -np.cos(inc) * np.ones_like(ph), 2.0 * np.sin(ph) * np.sin(inc)], axis=1)
    c, s = np.cos(np.deg2rad(raan_deg)), np.sin(np.deg2rad(raan_deg))
    Rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    return np.ascontiguousarray(B @ Rz.T)


# --------------------------------------------------------------------------------------------------
# batch container
# --------------------------------------------------------------------------------------------------
@dataclass
class SlewBatch:
    """Plain arrays of one batch, in ABI order (see include/tortoise_hip.h: tsat_solve_batch)."""

    N: int
    n_tab: int
    x0: np.ndarray      # (T,7)
    xf: np.ndarray      # (T,7)
    Btab: np.ndarray    # (n_btab, n_tab, 3)
    btab_idx: np.ndarray  # (T,) int32
    tau0: np.ndarray    # (T,)
    dtau: np.ndarray    # (T,)
    dt: np.ndarray      # (T,)
    Jmat: np.ndarray    # (T,9) column-major 3x3
    Qd: np.ndarray      # (T,7)
    Qfd: np.ndarray     # (T,7)
    Rd: np.ndarray      # (T,3)
    ulo: np.ndarray     # (T,3)
    uhi: np.ndarray     # (T,3)
    U0: np.ndarray      # (T,N-1,3)
    meta: dict = field(default_factory=dict)
    n_knots: np.ndarray = None   # (T,) int32 per-trajectory knot counts (ragged batch) or None: all N

    @property
    def T(self):
        return self.x0.shape[0]

    def slice(self, lo, hi):
        """Contiguous shard [lo, hi) of the batch (tables are kept whole, indices re-used)."""
        s = slice(lo, hi)
        c = np.ascontiguousarray
        return SlewBatch(self.N, self.n_tab, c(self.x0[s]), c(self.xf[s]), self.Btab, c(self.btab_idx[s]),
                         c(self.tau0[s]), c(self.dtau[s]), c(self.dt[s]), c(self.Jmat[s]), c(self.Qd[s]),
                         c(self.Qfd[s]), c(self.Rd[s]), c(self.ulo[s]), c(self.uhi[s]), c(self.U0[s]),
                         dict(self.meta), None if self.n_knots is None else c(self.n_knots[s]))


def jmat_cm(J):
    """3x3 (or (T,3,3)) inertia -> column-major 9-vector(s) as the ABI wants."""
    J = np.asarray(J, dtype=np.float64)
    if J.ndim == 2:
        return np.ascontiguousarray(J.T.reshape(9))
    return np.ascontiguousarray(np.transpose(J, (0, 2, 1)).reshape(-1, 9))


def make_batch(N, dt, q0, qf, J, Btab, btab_idx, alpha, beta, u_bnd, U0, w0=None, wf=None, degenerate_rd=None, r_scale=1.0,
               signed_w_max=False):
    """Assemble a SlewBatch from per-trajectory initial/goal quaternions (T,4); weights per trajectory from
    each trajectory's own eigen-axis guess (src/monte_carlo.jl:161-176)."""
    q0 = np.atleast_2d(np.asarray(q0, dtype=np.float64))
    qf = np.atleast_2d(np.asarray(qf, dtype=np.float64))
    T = q0.shape[0]
    if qf.shape[0] == 1 and T > 1:
        qf = np.repeat(qf, T, axis=0)
    w0 = np.zeros((T, 3)) if w0 is None else np.asarray(w0, dtype=np.float64)
    wf = np.zeros((T, 3)) if wf is None else np.asarray(wf, dtype=np.float64)
    x0 = np.ascontiguousarray(np.concatenate([w0, q0], axis=1))
    xf = np.ascontiguousarray(np.concatenate([wf, qf], axis=1))
    t = dt * np.arange(N + 1)  # t0:dt:t_final has N+1 points for N knots (src/TortoiseSat.jl:85-86)
    Qd = np.empty((T, 7)); Qfd = np.empty((T, 7)); Rd = np.empty((T, 3))
    same = T > 1 and np.all(x0 == x0[0]) and np.all(xf == xf[0])      # one guess serves a sweep from a fixed attitude
    for i in range(T):
        if same and i > 0:
            Qd[i], Qfd[i], Rd[i] = Qd[0], Qfd[0], Rd[0]
            continue
        wg, _ = eigen_axis_slew(x0[i], xf[i], t, rates_only=True)
        Qd[i], Qfd[i], Rd[i] = bryson_weights(wg, J, dt, alpha, beta, degenerate_rd, r_scale, signed_w_max)
    Btab = np.ascontiguousarray(np.asarray(Btab, dtype=np.float64))
    if Btab.ndim == 2:
        Btab = Btab[None]
    n_tab = Btab.shape[1]
    u_bnd = np.broadcast_to(np.asarray(u_bnd, dtype=np.float64), (3,))
    return SlewBatch(
        N=N, n_tab=n_tab, x0=x0, xf=xf, Btab=Btab,
        btab_idx=np.ascontiguousarray(np.asarray(btab_idx, dtype=np.int32)),
        tau0=np.zeros(T), dtau=np.full(T, float(n_tab) / float(N)), dt=np.full(T, float(dt)),
        Jmat=np.ascontiguousarray(np.repeat(jmat_cm(J)[None], T, axis=0)),
        Qd=Qd, Qfd=Qfd, Rd=Rd,
        ulo=np.ascontiguousarray(np.repeat(-u_bnd[None], T, axis=0)),
        uhi=np.ascontiguousarray(np.repeat(u_bnd[None], T, axis=0)),
        U0=np.ascontiguousarray(np.asarray(U0, dtype=np.float64).reshape(T, N - 1, 3)),
    )


def random_unit_quats(rng, T):
    q = rng.standard_normal((T, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def trajectory_stream(seed, j):
    """The random stream of GLOBAL trajectory ``j`` of a sweep: a counter-based generator (Philox4x64) keyed by (seed, j), so that
    what is drawn for a trajectory depends on its index alone — not on the shard it falls into, nor on how many ranks share the
    sweep (a sharded sweep is the concatenation of the single-GPU results; src/paper_images/heatmap.jl:114-127 draws
    once per run index i)."""
    return np.random.Generator(np.random.Philox(key=np.array([int(seed) & 0xFFFFFFFFFFFFFFFF, int(j)], dtype=np.uint64)))


# --------------------------------------------------------------------------------------------------
# BASELINE.json workloads
# --------------------------------------------------------------------------------------------------
def workload_single_slew(N=500):
    """configs[0]: the reference's own single slew (src/TortoiseSat.jl:117-199): ISS-like orbit, 1P inertia,
    90 deg about [1,0,1]/sqrt2 to identity, dt 0.2, U0 = 0, |u| <= 1, alpha = 10, beta = 1e3, rk3, 20 x 50."""
    dt = 0.2
    a_km = R_EARTH_KM + 400.0
    B = dipole_btable(N, dt, a_km, 51.6, 0.0, 90.0)
    q0 = axis_angle_quat([1.0, 0.0, 1.0], np.deg2rad(90.0))
    qf = np.array([1.0, 0.0, 0.0, 0.0])
    b = make_batch(N, dt, q0[None], qf[None], INERTIA["1P"], B, np.zeros(1, np.int32), 10.0, 1.0e3, 1.0,
                   np.zeros((1, N - 1, 3)))
    b.meta = dict(name="single_slew", max_outer=20, max_inner=50, dj_counter_limit=10)
    return b


def workload_monte_carlo(T=1024, N=1000, seed=20190530, random_orbit=False, degenerate_rd=None, tables=True, j0=None):
    """configs[1] (and [2] with random_orbit=True): Monte-Carlo of src/monte_carlo.jl:107-198 with the initial
    attitude randomised — q0 uniform on S^3, qf = [sqrt2/2, sqrt2/2, 0, 0] (:114), 1U inertia (:31-33),
    dt 0.2, U0 ~ U(0,1e-3) (:193), Bryson weights alpha = 0.1, beta = 1e3 (:170-176), |u| <= 19 (:179),
    budget 5 x 10 with dJ_counter_limit 1 (:189-191), SSO i = 96.6 deg (:124).
    ``j0`` = None: one stream for the batch (the instance depends on ``seed`` and ``T``). ``j0`` = the global index of the batch's
    first trajectory: every trajectory draws from its own stream keyed by (seed, j0 + i) (``trajectory_stream``) — shards of a
    sweep built with different ``j0`` / ``T`` concatenate to the same arrays, bit for bit, whatever the number of ranks."""
    dt = 0.2
    a_km = R_EARTH_KM + 400.0
    qf = np.array([np.sqrt(2.0) / 2.0, np.sqrt(2.0) / 2.0, 0.0, 0.0])
    if j0 is None:
        rng = np.random.Generator(np.random.PCG64(seed))
        q0 = random_unit_quats(rng, T)
        U0 = rng.random((T, N - 1, 3)) / 1000.0
        if random_orbit:
            raan = rng.random(T) * 360.0
            nu = rng.random(T) * 360.0
    else:
        q0 = np.empty((T, 4)); U0 = np.empty((T, N - 1, 3)); raan = np.empty(T); nu = np.empty(T)
        for i in range(T):
            g = trajectory_stream(seed, j0 + i)
            q0[i] = random_unit_quats(g, 1)[0]
            raan[i], nu[i] = g.random(2) * 360.0
            U0[i] = g.random((N - 1, 3)) / 1000.0
    if random_orbit:
        # tables = False: placeholders, for callers that attach IGRF tables afterwards (magnetic.attach_igrf_tables)
        B = np.stack([dipole_btable(N, dt, a_km, 96.6, raan[i], nu[i]) for i in range(T)]) if tables else np.zeros((T, 1, 3))
        idx = np.arange(T, dtype=np.int32)
    else:
        B = dipole_btable(N, dt, a_km, 96.6, 0.0, 0.0)
        idx = np.zeros(T, np.int32)
    b = make_batch(N, dt, q0, qf[None], INERTIA["1U"], B, idx, 0.1, 1.0e3, 19.0, U0, degenerate_rd=degenerate_rd)
    # A[i,:] = [0, alt + R_E, 96.6, RAAN, 0, anomaly] (src/monte_carlo.jl:118-127): what magnetic.attach_igrf_tables reads
    kep = np.zeros((T if random_orbit else 1, 6))
    kep[:, 1], kep[:, 2] = a_km, 96.6
    if random_orbit:
        kep[:, 3], kep[:, 5] = raan, nu
    b.meta = dict(name="monte_carlo_random_orbit" if random_orbit else "monte_carlo", max_outer=5, max_inner=10,
                  dj_counter_limit=1, seed=seed, kep=kep, field="tilted dipole (surrogate)")
    return b


def workload_inclination_sweep(T=8192, N=1000, seed=20190601, j0=0, T_total=65536, tables=True, stride=1):
    """configs[3] shard: deterministic inclination sweep i = 90 (j + 1/2)/T_total deg (the reference draws
    rand*90, src/paper_images/heatmap.jl:120), random RAAN / true anomaly, q0 = [0,0,1,0] (heatmap.jl:106),
    Bryson weights with that script's R scaled by 0.1 (:172) — its ``omega_max = maximum(X[1:3,:])`` without ``abs.`` (:164) is
    NOT followed: for this fixed q0 / qf pair the eigen-axis guess has no positive rate component (axis (0, -c, -c)), so the
    line as written gives omega_max = 0 and an infinite Q; the ``abs.`` of src/monte_carlo.jl:167 is used instead —
    budget 3 x 50 (heatmap.jl:197-198), the quaternion hooks of ``Model(DerivFunction,n,m,quaternion_error,
    quaternion_expansion)`` (:154; ``meta["error_state"] = 1``). ``j0`` is the first global index of this shard, ``stride``
    the step between its indices: 1 for a contiguous block of the sweep; ``world`` (with ``j0 = rank``) deals the sweep out
    to the ranks like cards, so that every rank sees every inclination band — the iteration count of a slew depends on the
    inclination, and contiguous blocks give the ranks unequal work (tools/shard_balance.py).
    RAAN, true anomaly and U0 of global trajectory j come from its own stream keyed by (seed, j) (``trajectory_stream``): the
    sweep is ONE set of 65536 slews however it is sharded — shards concatenate to the whole, bit for bit."""
    dt = 0.2
    a_km = R_EARTH_KM + 400.0
    jg = j0 + stride * np.arange(T)
    inc = 90.0 * (jg + 0.5) / float(T_total)
    raan = np.empty(T); nu = np.empty(T); U0 = np.empty((T, N - 1, 3))
    for i in range(T):
        g = trajectory_stream(seed, jg[i])
        raan[i], nu[i] = g.random(2) * 360.0
        U0[i] = g.random((N - 1, 3)) / 1000.0
    B = np.stack([dipole_btable(N, dt, a_km, inc[i], raan[i], nu[i]) for i in range(T)]) if tables else np.zeros((T, 1, 3))
    q0 = np.repeat(np.array([[0.0, 0.0, 1.0, 0.0]]), T, axis=0)
    qf = np.array([np.sqrt(2.0) / 2.0, np.sqrt(2.0) / 2.0, 0.0, 0.0])
    b = make_batch(N, dt, q0, qf[None], INERTIA["1U"], B, np.arange(T, dtype=np.int32), 0.1, 1.0e3, 19.0, U0, r_scale=0.1)
    kep = np.zeros((T, 6))
    kep[:, 1], kep[:, 2], kep[:, 3], kep[:, 5] = a_km, inc, raan, nu
    b.meta = dict(name="inclination_sweep", max_outer=3, max_inner=50, dj_counter_limit=1, error_state=1, seed=seed, kep=kep,
                  field="tilted dipole (surrogate)")
    return b
