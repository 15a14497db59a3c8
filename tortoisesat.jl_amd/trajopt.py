"""Host-side mirror of the TrajectoryOptimization.jl surface the reference drives.

The reference builds ONE problem and calls ``solve!(prob, solver)`` (src/TortoiseSat.jl:145-199); its Monte-Carlo
does that in a serial loop (src/monte_carlo.jl:118-235). Here the same vocabulary builds a *batch* and hands it
to the HIP library in one call:

    model   = Model(DerivFunction(J, B_ECI), 8, 3);  model_d = rk3(model)            # :145-146
    obj     = LQRObjective(Q, R, Qf, xf, N)                                           # :169
    cons    = Constraints(N); cons[k] += BoundConstraint(8,3,u_max=1,u_min=-1); cons[N] += goal_constraint(xf)
    prob    = Problem(model_d, obj, constraints=cons, x0=x0, xf=xf, N=N, dt=dt)       # :190
    initial_controls_(prob, U0)                                                       # :191  (Julia: initial_controls!)
    opts    = AugmentedLagrangianSolverOptions(); opts.opts_uncon.iterations = 50; opts.iterations = 20
    solver  = AugmentedLagrangianSolver(prob, opts)                                   # :197
    solve_(prob, solver)                                                              # :199  (Julia: solve!)
    X, U    = prob.X, prob.U                                                          # :202-203

``BatchProblem([prob, ...])`` / ``BatchProblem.from_arrays(SlewBatch)`` + ``solve_`` do the same for T slews.
Every solve goes through libtortoise_hip.so (ctypes); there is no CPU path in this module.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _abi
from .slew_setup import SlewBatch, jmat_cm

STATUS_NAMES = {0: "converged", 1: "max_outer", 2: "reg_fail", 3: "diverged"}


# ------------------------------------------------------------------------------------------------------
# model / objective / constraints (data carriers; the dynamics themselves live in the HIP kernel)
# ------------------------------------------------------------------------------------------------------
@dataclass
class DerivFunction:
    """The reference's dynamics callback (src/DerivFunction.jl:1-48) as data: the globals it reads
    (``p.J``, ``B_ECI``, ``N``, ``tf``, ``t0``) become explicit fields."""

    J: np.ndarray                 # 3x3 inertia
    B_ECI: np.ndarray             # (n_tab, 3) table, Tesla
    tau0: float = 0.0             # table row of knot 0
    rows_per_knot: float = None   # d(row)/d(knot); None -> n_tab / N (physically consistent playback)


def quaternion_error(X1, X2):
    """Marker for the reference's state-difference hook (src/quaternion_toolbox.jl:58-75); evaluated on the GPU."""
    raise NotImplementedError("hook marker: pass it to Model(...); the HIP kernel evaluates it")


def quaternion_expansion(cost, x, u=None):
    """Marker for the reference's cost-expansion hook (src/quaternion_toolbox.jl:15-50); evaluated on the GPU."""
    raise NotImplementedError("hook marker: pass it to Model(...); the HIP kernel evaluates it")


@dataclass
class Model:
    """Model(f!, n, m) (src/TortoiseSat.jl:145) or Model(f!, n, m, quaternion_error, quaternion_expansion)
    (src/monte_carlo.jl:158): passing both hooks selects the kernel's error-state mode."""

    f: DerivFunction
    n: int = 8
    m: int = 3
    state_diff: object = None
    cost_expansion: object = None
    integrator: int = 0           # 0 = continuous; 3 / 4 after rk3 / rk4

    @property
    def error_state(self):
        hooks = (self.state_diff is quaternion_error, self.cost_expansion is quaternion_expansion)
        if any(hooks) and not all(hooks):
            raise ValueError("pass both quaternion_error and quaternion_expansion, or neither (src/monte_carlo.jl:158)")
        if (self.state_diff is not None or self.cost_expansion is not None) and not all(hooks):
            raise ValueError("only the reference's quaternion hooks are implemented on the GPU")
        return 1 if all(hooks) else 0

    def __post_init__(self):
        if self.n not in (7, 8) or self.m != 3:
            raise ValueError("the HIP path implements the TortoiseSat model only: n = 8 (or 7), m = 3")


def rk3(model):
    """TrajectoryOptimization.rk3(model) (src/TortoiseSat.jl:146)."""
    return Model(model.f, model.n, model.m, model.state_diff, model.cost_expansion, 3)


def rk4(model):
    return Model(model.f, model.n, model.m, model.state_diff, model.cost_expansion, 4)


@dataclass
class LQRObjective:
    """LQRObjective(Q, R, Qf, xf, N) (src/TortoiseSat.jl:169). Q/Qf/R must be diagonal (they are, :157-168)."""

    Q: np.ndarray
    R: np.ndarray
    Qf: np.ndarray
    xf: np.ndarray
    N: int

    def diagonals(self):
        out = []
        for M, n in ((self.Q, 7), (self.Qf, 7), (self.R, 3)):
            M = np.asarray(M, dtype=np.float64)
            if M.ndim == 2:
                if np.any(M - np.diag(np.diag(M)) != 0.0):
                    raise ValueError("only diagonal LQR weights are supported (as in src/TortoiseSat.jl:157-168)")
                M = np.diag(M)
            if M.shape[0] == 8:
                if M[7] != 0.0:
                    raise ValueError("the time state (8th) must carry zero weight (src/TortoiseSat.jl:157-167)")
                M = M[:7]
            if M.shape[0] != n:
                raise ValueError("weight dimension mismatch")
            out.append(np.ascontiguousarray(M))
        return out


@dataclass
class BoundConstraint:
    """BoundConstraint(n, m, u_max=, u_min=) (src/TortoiseSat.jl:178) — control box only."""

    n: int
    m: int
    u_max: object = np.inf
    u_min: object = -np.inf


@dataclass
class GoalConstraint:
    xf: np.ndarray


def goal_constraint(xf):
    """goal_constraint(xf) (src/TortoiseSat.jl:182)."""
    return GoalConstraint(np.asarray(xf, dtype=np.float64))


class _KnotConstraints(list):
    def __iadd__(self, c):
        self.append(c)
        return self


class Constraints:
    """Constraints(N); ``constraints[k] += c`` with Julia's 1-based k (src/TortoiseSat.jl:184-188)."""

    def __init__(self, N):
        self.N = N
        self._k = [_KnotConstraints() for _ in range(N)]

    def __getitem__(self, k):
        return self._k[k - 1]

    def __setitem__(self, k, v):
        self._k[k - 1] = v


@dataclass
class Problem:
    """Problem(model_d, obj; constraints, x0, xf, N, dt) (src/TortoiseSat.jl:190)."""

    model: Model
    obj: LQRObjective
    constraints: Constraints = None
    x0: np.ndarray = None
    xf: np.ndarray = None
    N: int = 0
    dt: float = 0.2
    U0: np.ndarray = None
    X: np.ndarray = None   # (7, N) after solve_  (hcat(sat.X...), src/TortoiseSat.jl:202)
    U: np.ndarray = None   # (3, N-1)
    K: np.ndarray = None   # (3, 7, N-1)
    stats: object = None


def initial_controls_(prob, U0):
    """initial_controls!(prob, U0) (src/TortoiseSat.jl:191). Accepts (3, N-1) or the reference's (3, N+1)."""
    U0 = np.asarray(U0, dtype=np.float64)
    if U0.shape[0] != 3 or U0.shape[1] < prob.N - 1:
        raise ValueError("U0 must be 3 x (>= N-1)")
    prob.U0 = np.ascontiguousarray(U0[:, : prob.N - 1])


@dataclass
class _UnconOptions:
    iterations: int = 50
    cost_tolerance: float = 1e-4
    gradient_norm_tolerance: float = 1e-5
    iterations_linesearch: int = 20
    line_search_lower_bound: float = 1e-8
    line_search_upper_bound: float = 10.0
    bp_reg_initial: float = 0.0
    bp_reg_increase_factor: float = 1.6
    bp_reg_min: float = 1e-8
    bp_reg_max: float = 1e8
    bp_reg_fp: float = 10.0
    max_state_value: float = 1e8
    dJ_counter_limit: int = 10


@dataclass
class AugmentedLagrangianSolverOptions:
    """AugmentedLagrangianSolverOptions{Float64}() (src/TortoiseSat.jl:194); field names follow TrajOpt."""

    iterations: int = 20
    constraint_tolerance: float = 1e-3
    penalty_initial: float = 1.0
    penalty_scaling: float = 10.0
    penalty_max: float = 1e8
    dual_max: float = 1e8
    opts_uncon: _UnconOptions = field(default_factory=_UnconOptions)

    def to_abi(self, N, n_tab, integrator, terminal_mask=0x7F, u_scale=1e-2, error_state=0):
        o = _abi.Options()
        _abi.load().tsat_default_options(C.byref(o))
        u = self.opts_uncon
        o.n_knots, o.n_tab, o.integrator, o.precision = N, n_tab, integrator, 64
        o.max_outer, o.max_inner = self.iterations, u.iterations
        o.max_linesearch, o.dj_counter_limit = u.iterations_linesearch, u.dJ_counter_limit
        o.cost_tol, o.grad_tol, o.constraint_tol = u.cost_tolerance, u.gradient_norm_tolerance, self.constraint_tolerance
        o.penalty_init, o.penalty_scale, o.penalty_max, o.dual_max = (
            self.penalty_initial, self.penalty_scaling, self.penalty_max, self.dual_max)
        o.reg_init, o.reg_scale, o.reg_min, o.reg_max, o.reg_fp = (
            u.bp_reg_initial, u.bp_reg_increase_factor, u.bp_reg_min, u.bp_reg_max, u.bp_reg_fp)
        o.ls_lower, o.ls_upper, o.max_state = u.line_search_lower_bound, u.line_search_upper_bound, u.max_state_value
        o.u_scale, o.terminal_mask, o.error_state = u_scale, terminal_mask, error_state
        return o


# ------------------------------------------------------------------------------------------------------
# the solver object: owns a tsat_handle (one per GPU)
# ------------------------------------------------------------------------------------------------------
class AugmentedLagrangianSolver:
    """AugmentedLagrangianSolver(prob, opts) (src/TortoiseSat.jl:197): owns the GPU workspace."""

    def __init__(self, prob=None, opts=None, device=0):
        self.opts = opts if opts is not None else AugmentedLagrangianSolverOptions()
        self._lib = _abi.load()
        h = C.c_void_p()
        rc = self._lib.tsat_create(C.byref(h), int(device))
        if rc != 0:
            raise RuntimeError(
                f"tsat_create(device={device}) failed with code {rc}: a gfx950 GPU is required "
                "(-2: no HIP device, -5: not gfx950). There is no CPU fallback.")
        self._h = h
        self.device = device
        self.last_kernel_ms = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tsat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.tsat_last_error(self._h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def set_kernel_variant(self, variant):
        """0 automatic (dense above 1024 trajectories), 1 wide (one wavefront per SIMD), 2 dense (two): see
        tsat_set_kernel_variant. Results do not depend on it."""
        self._check(self._lib.tsat_set_kernel_variant(self._h, int(variant)), "tsat_set_kernel_variant")

    def set_endgame(self, suspend_at):
        """packed builds: live count at which the wavefronts park their trajectories for a one-per-wavefront launch
        (-1 automatic, 0 never): see tsat_set_endgame. Results do not depend on it."""
        self._check(self._lib.tsat_set_endgame(self._h, int(suspend_at)), "tsat_set_endgame")

    def selected_build(self, abi_opts):
        """(build, endgame_at) the next run with these options launches on the reserved batch: build 1 wide, 2 dense, 3 packed,
        4 packed8, 5 packed8w, 6 packed16w, 7 packed4w; endgame_at = live count at which a packed launch parks its trajectories (0: none). tsat_selected_build."""
        b, e = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.tsat_selected_build(self._h, C.byref(abi_opts), C.byref(b), C.byref(e)), "tsat_selected_build")
        return int(b.value), int(e.value)

    # ---- resident-batch API -------------------------------------------------------------------
    def upload(self, batch: SlewBatch, max_linesearch):
        # batch.Btab is None: one table per trajectory, left on the device by the last tsat_btable_batch (magnetic.py, host=False)
        n_btab = batch.T if batch.Btab is None else batch.Btab.shape[0]
        self._check(self._lib.tsat_batch_reserve(self._h, batch.T, batch.N, batch.n_tab, n_btab,
                                                 max_linesearch), "tsat_batch_reserve")
        d = _abi.as_dp
        self._check(self._lib.tsat_batch_upload(
            self._h, d(batch.x0), d(batch.xf), d(batch.Btab), _abi.as_ip(batch.btab_idx), d(batch.tau0),
            d(batch.dtau), d(batch.dt), d(batch.Jmat), d(batch.Qd), d(batch.Qfd), d(batch.Rd), d(batch.ulo),
            d(batch.uhi), d(batch.U0)), "tsat_batch_upload")
        if batch.n_knots is not None:   # ragged batch: per-trajectory horizons (src/monte_carlo.jl:140-145)
            nk = np.ascontiguousarray(batch.n_knots, dtype=np.int32)
            self._check(self._lib.tsat_batch_knots(self._h, _abi.as_ip(nk)), "tsat_batch_knots")
        self._shape = (batch.T, batch.N)

    def run(self, abi_opts):
        ms = C.c_float(0.0)
        self._check(self._lib.tsat_batch_run(self._h, C.byref(abi_opts), C.byref(ms)), "tsat_batch_run")
        self.last_kernel_ms = float(ms.value)
        return self.last_kernel_ms

    def download(self, want_K=True, want_trajectories=True):
        T, N = self._shape
        X = np.empty((T, N, 7)) if want_trajectories else None      # None: the statistics only, the batch stays on the device
        U = np.empty((T, N - 1, 3)) if want_trajectories else None
        K = np.empty((T, N - 1, 7, 3)) if want_K else None
        stats = np.zeros(T, dtype=_abi.STATS_DTYPE)
        self._check(self._lib.tsat_batch_download(self._h, _abi.as_dp(X), _abi.as_dp(U), _abi.as_dp(K),
                                                  stats.ctypes.data_as(C.c_void_p)), "tsat_batch_download")
        return dict(X=X, U=U, K=K, stats=stats)

    def export_device(self, X_ptr=None, U_ptr=None, K_ptr=None, stats_ptr=None):
        """Unpack results into caller-owned device buffers (raw pointers, e.g. ``tensor.data_ptr()``)."""
        self._check(self._lib.tsat_batch_export_device(self._h, X_ptr, U_ptr, K_ptr, stats_ptr),
                    "tsat_batch_export_device")

    # ---- sweep exchange (RCCL all-gather behind the C ABI) ------------------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte communicator id (rank 0 creates it, every rank passes the same bytes to ``comm_init``)."""
        buf = C.create_string_buffer(_abi.TSAT_COMM_ID_BYTES)
        rc = _abi.load().tsat_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"tsat_comm_unique_id failed ({rc}): RCCL not available")
        return buf.raw

    @staticmethod
    def comm_available():
        """local probe, no communication: can this process load RCCL behind the C ABI?"""
        return _abi.load().tsat_comm_available() == 0

    def comm_init(self, id_bytes, rank, world):
        self._check(self._lib.tsat_comm_init(self._h, bytes(id_bytes), int(rank), int(world)), "tsat_comm_init")
        self._comm = (int(rank), int(world))

    def comm_destroy(self):
        self._check(self._lib.tsat_comm_destroy(self._h), "tsat_comm_destroy")

    def sweep_allgather(self, X_all=None, U_all=None, stats_all=None, on_device=False):
        """All ranks, after ``run``: results of every rank's shard in rank order. Arguments are raw pointers (device
        pointers with ``on_device``), or — host mode — None to have NumPy arrays allocated and returned."""
        if on_device:
            self._check(self._lib.tsat_sweep_allgather(self._h, X_all, U_all, stats_all, 1), "tsat_sweep_allgather")
            return None
        T, N = self._shape
        W = self._comm[1]
        X = np.empty((W * T, N, 7)); U = np.empty((W * T, N - 1, 3)); st = np.zeros(W * T, dtype=_abi.STATS_DTYPE)
        self._check(self._lib.tsat_sweep_allgather(self._h, X.ctypes.data_as(C.c_void_p), U.ctypes.data_as(C.c_void_p),
                                                   st.ctypes.data_as(C.c_void_p), 0), "tsat_sweep_allgather")
        return dict(X=X, U=U, stats=st)

    def trace(self, rows):
        self._check(self._lib.tsat_batch_trace(self._h, rows), "tsat_batch_trace")
        self._trace_rows = rows

    def trace_download(self):
        T, _ = self._shape
        tr = np.zeros((T, self._trace_rows, 8))
        self._check(self._lib.tsat_batch_trace_download(self._h, _abi.as_dp(tr)), "tsat_batch_trace_download")
        return tr

    def reserved_bytes(self):
        return int(self._lib.tsat_batch_bytes(self._h))

    def workspace_bytes(self):
        """HBM held by the grow-only workspaces of the stages around the solve (downloads, gathers, tracking, tables ...)"""
        return int(self._lib.tsat_workspace_bytes(self._h))

    def workspace_trim(self, everything=False):
        """free the staging buffers of downloads / gathers (``everything``: all workspaces, resident field tables included)"""
        self._check(self._lib.tsat_workspace_trim(self._h, 1 if everything else 0), "tsat_workspace_trim")

    def btable_generation(self):
        """counter bumped by every ``tsat_btable_batch`` on this handle: identifies the resident field tables"""
        return int(self._lib.tsat_btable_generation(self._h))


# ------------------------------------------------------------------------------------------------------
# batches of Problems
# ------------------------------------------------------------------------------------------------------
def _terminal_mask(prob):
    """bit mask of state components under the terminal goal constraint (the time state is dropped)."""
    if prob.constraints is None:
        return 0
    for c in prob.constraints[prob.N]:
        if isinstance(c, GoalConstraint):
            return 0x7F
    return 0


def _control_box(prob):
    lo, hi = np.full(3, -np.inf), np.full(3, np.inf)
    if prob.constraints is not None:
        for k in range(1, prob.N):
            for c in prob.constraints[k]:
                if isinstance(c, BoundConstraint):
                    hi = np.minimum(hi, np.broadcast_to(np.asarray(c.u_max, dtype=np.float64), (3,)))
                    lo = np.maximum(lo, np.broadcast_to(np.asarray(c.u_min, dtype=np.float64), (3,)))
                    return lo, hi
    return lo, hi


class BatchProblem:
    """T independent slews solved in one HIP launch — the loop body of src/monte_carlo.jl:118-235 as data."""

    def __init__(self, problems):
        problems = list(problems)
        if not problems:
            raise ValueError("empty batch")
        p0 = problems[0]
        integ = p0.model.integrator
        N = max(p.N for p in problems)      # common stride; shorter horizons become a ragged batch (tsat_batch_knots)
        if integ not in (3, 4):
            raise ValueError("discretise the model with rk3(model) or rk4(model) first (src/TortoiseSat.jl:146)")
        self.problems = problems
        self.integrator = integ
        self.error_state = p0.model.error_state
        self.terminal_mask = _terminal_mask(p0)
        T = len(problems)
        tabs, idx, seen = [], np.zeros(T, np.int32), {}
        x0 = np.zeros((T, 7)); xf = np.zeros((T, 7)); Jm = np.zeros((T, 9))
        Qd = np.zeros((T, 7)); Qfd = np.zeros((T, 7)); Rd = np.zeros((T, 3))
        ulo = np.zeros((T, 3)); uhi = np.zeros((T, 3)); U0 = np.zeros((T, N - 1, 3))
        tau0 = np.zeros(T); dtau = np.zeros(T); dt = np.zeros(T)
        n_knots = np.array([p.N for p in problems], dtype=np.int32)
        for t, p in enumerate(problems):
            if (p.N < 2 or p.model.integrator != integ or _terminal_mask(p) != self.terminal_mask
                    or p.model.error_state != self.error_state):
                raise ValueError("all problems of a batch must share integrator, hooks and constraint structure")
            f = p.model.f
            key = id(f.B_ECI)
            if key not in seen:
                seen[key] = len(tabs)
                tabs.append(np.asarray(f.B_ECI, dtype=np.float64))
            idx[t] = seen[key]
            x0[t] = np.asarray(p.x0, dtype=np.float64)[:7]
            xf[t] = np.asarray(p.xf if p.xf is not None else p.obj.xf, dtype=np.float64)[:7]
            Jm[t] = jmat_cm(f.J)
            Qd[t], Qfd[t], Rd[t] = p.obj.diagonals()
            ulo[t], uhi[t] = _control_box(p)
            if p.U0 is None:
                raise ValueError("call initial_controls_(prob, U0) first (src/TortoiseSat.jl:191)")
            U0[t, : p.N - 1] = p.U0.T
            n_tab = tabs[idx[t]].shape[0]
            tau0[t] = f.tau0
            dtau[t] = f.rows_per_knot if f.rows_per_knot is not None else n_tab / float(p.N)
            dt[t] = p.dt
        n_tab = tabs[0].shape[0]
        if any(tb.shape != (n_tab, 3) for tb in tabs):
            raise ValueError("all B tables of a batch must have the same number of rows")
        if not np.all(np.isfinite(ulo)) or not np.all(np.isfinite(uhi)):
            raise ValueError("a finite control box is required (BoundConstraint, src/TortoiseSat.jl:178)")
        self.arrays = SlewBatch(N, n_tab, x0, xf, np.ascontiguousarray(np.stack(tabs)), idx, tau0, dtau, dt, Jm,
                                Qd, Qfd, Rd, ulo, uhi, U0)
        if np.any(n_knots != N):
            self.arrays.n_knots = n_knots

    @classmethod
    def from_arrays(cls, batch: SlewBatch, integrator=3, terminal_mask=0x7F, error_state=0):
        self = cls.__new__(cls)
        self.problems = None
        self.integrator = integrator
        self.error_state = error_state
        self.terminal_mask = terminal_mask
        self.arrays = batch
        return self


def solve_(prob, solver, want_K=True, want_trajectories=True):
    """solve!(prob, solver) (src/TortoiseSat.jl:199). ``prob`` may be a Problem or a BatchProblem.

    Mutates the problem(s): ``.X`` (7,N), ``.U`` (3,N-1), ``.K`` (3,7,N-1), ``.stats``; returns the raw result
    dict (X (T,N,7), U (T,N-1,3), K (T,N-1,7,3), stats)."""
    batch = prob if isinstance(prob, BatchProblem) else BatchProblem([prob])
    b = batch.arrays
    o = solver.opts.to_abi(b.N, b.n_tab, batch.integrator, batch.terminal_mask, error_state=batch.error_state)
    solver.upload(b, o.max_linesearch)
    solver.run(o)
    res = solver.download(want_K=want_K, want_trajectories=want_trajectories)
    if batch.problems is not None and want_trajectories:
        for t, p in enumerate(batch.problems):
            p.X = np.ascontiguousarray(res["X"][t, : p.N].T)
            p.U = np.ascontiguousarray(res["U"][t, : p.N - 1].T)
            p.K = np.ascontiguousarray(res["K"][t, : p.N - 1].transpose(2, 1, 0)) if want_K else None
            p.stats = res["stats"][t]
    return res
