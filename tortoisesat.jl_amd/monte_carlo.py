"""The reference's Monte-Carlo experiment end to end, one batch per call (src/monte_carlo.jl:107-262).

Per trial the script (a serial loop there, a batch here) draws a sun-synchronous orbit (:118-127), samples a coarse
field table over 40 min (:134), picks the horizon ``t_final`` where the magnetic Gramian is conditioned below the
cutoff (:137-140), resamples the field over that horizon (:149), builds the eigen-axis guess and Bryson weights
(:161-176), solves the slew with a 5 x 10 AL-iLQR budget (:179-196), tracks it with TVLQR on the noisy plant (:203-230)
and reduces the closed-loop run to a slew time and a failure flag (:237-262). Every numeric stage is a call into the
C-ABI library (field tables, horizon, solve, tracking); this module is the script arithmetic in between.

Deliberate differences, all visible as arguments:
  * ``field_rate``: "physical" replays the resampled table at its own rate, "reference" at the script's
    ``1/(tf - t0)`` with tf = 2400 s (src/DerivFunction.jl:44 with the script globals);
  * the state box ``x_bnd = 10`` (:180) never binds (|w| << 10, |q| <= 1) and is not modelled;
  * randomness comes from generators keyed per trial index (``numpy`` PCG64 for the script's draws, Philox4x32-10
    inside the tracking kernel for the plant noise), so a sweep gives the same trials whatever the number of ranks or
    the chunking.
"""
from dataclasses import dataclass

import numpy as np

from . import horizon as hz
from . import magnetic as mg
from . import tracking as tr
from . import trajopt as to
from .slew_setup import INERTIA, SlewBatch, bryson_weights, bryson_weights_ragged, eigen_axis_slew, jmat_cm  # noqa: F401


@dataclass
class MonteCarloSetup:
    """The script's constants (src/monte_carlo.jl:24-80, 107-116, 163-191, 217-228)."""

    N: int = 5000                 # table resolution (:38)
    t0: float = 0.0
    tf: float = 2400.0            # coarse span, 40 min (:77)
    cutoff: float = 30.0          # Gramian condition number (:78)
    alt: float = 400.0
    R_E: float = 6371.0
    GM: float = 3.986004418e5     # km^3/s^2 (:25)
    inclination: float = 96.6     # (:124)
    mjd: float = 58155.0          # (:75)
    igrf_date: float = 2019.0     # (:82)
    dt: float = 0.2               # t_total step (:145)
    alpha: float = 0.1            # Bryson (:170-176)
    beta: float = 1.0e3
    u_bnd: float = 19.0           # (:179)
    outer: int = 5                # (:189-191)
    inner: int = 10
    dJ_counter_limit: int = 1
    lqr_alpha: float = 10.0       # (:217-227)
    lqr_beta: float = 10.0
    lqr_r: float = 0.5e3          # (:228)
    w_tol: float = 0.05           # slew_limits (:70-71)
    angle_tol: float = 0.08727
    rate_as_written: bool = False  # statistic: True takes `norm(sim_states[i][1:3,i])` literally (:247: the rate of sample i = trial number)
    field_rate: str = "physical"
    inertia: str = "1U"           # (:31-33)


class GpuStages:
    """The four numeric stages through the C ABI; tests substitute the oracle's by passing another object."""

    resident_tables = True    # field tables stay on the device between the stages (run_trials asks with host=False)

    def __init__(self, solver):
        self.solver = solver
        self._resident_shape = None
        self._resident_gen = None

    def magnetic_simulation(self, kep, t0, tf, N, s, host=True):
        self._resident_shape = (len(kep), 2 * N)
        B = mg.magnetic_simulation(self.solver, kep, t0, tf, N, mjd=s.mjd, gm=s.GM, alt=s.alt, R_E=s.R_E, date=s.igrf_date,
                                   want_pos=False, host=host)[0]
        self._resident_gen = self.solver.btable_generation()       # identifies THESE tables to the consumers below
        return B

    def _resident_tables_are_mine(self, what):
        # the library identifies resident tables by shape only; any other tsat_btable_batch on the handle in between (a second
        # experiment, attach_igrf_tables) would be picked up silently — the generation counter tells (include/tortoise_hip.h)
        if self.solver.btable_generation() != self._resident_gen:
            raise RuntimeError(f"{what}: the field tables resident on the device are not the ones this experiment generated "
                               "(another tsat_btable_batch ran on the handle in between)")

    def condition_based_time(self, B, dt_row, cutoff):
        if B is None:
            self._resident_tables_are_mine("condition_based_time")
        return hz.condition_based_time(self.solver, B, dt_row, cutoff, resident_shape=self._resident_shape)[0]

    def solve(self, batch, s, want_trajectories=True):
        if batch.Btab is None:
            self._resident_tables_are_mine("solve")
        opts = to.AugmentedLagrangianSolverOptions()
        opts.iterations, opts.opts_uncon.iterations, opts.opts_uncon.dJ_counter_limit = s.outer, s.inner, s.dJ_counter_limit
        self.solver.opts = opts
        return to.solve_(to.BatchProblem.from_arrays(batch, error_state=1), self.solver, want_K=False, want_trajectories=want_trajectories)

    def attitude_simulation(self, batch, X, U, x0_sim, Qd, Qfd, Rd, noise_seed, noise_ids, s, want_trajectories=True):
        # the batch just solved is still resident: its trajectories and tables do not travel again
        return tr.attitude_simulation(self.solver, batch, None, None, x0_sim, Qd, Qfd, Rd, noise_seed=noise_seed, noise_ids=noise_ids,
                                      w_tol=s.w_tol, angle_tol=s.angle_tol, want_K=False, want_trajectories=want_trajectories,
                                      rate_as_written=bool(getattr(s, "rate_as_written", False)))


def trial_rng(seed, i):
    return np.random.Generator(np.random.PCG64([int(seed), int(i)]))


def draw_orbits(seed, lo, hi, s):
    """A[i,:] = [0, alt + R_E, 96.6, rand 360, 0, rand 360] (src/monte_carlo.jl:118-127)."""
    A = np.zeros((hi - lo, 6))
    A[:, 1], A[:, 2] = s.alt + s.R_E, s.inclination
    for j, i in enumerate(range(lo, hi)):
        r = trial_rng(seed, i)
        A[j, 3], A[j, 5] = r.random() * 360.0, r.random() * 360.0
    return A


def knot_counts(t_final, t0, dt):
    """length(range(t0, step=dt, stop=t_final)) (src/monte_carlo.jl:145). Julia builds a float range in twice precision, so
    0:0.2:2.4 has 13 elements although 2.4/0.2 evaluates to 11.999999999999998 in plain fp64; with the script's constants
    (t_final = idx*0.48) that hits every idx that is a multiple of 5. The quotient is therefore floored with a relative
    guard of a few ulps instead of bare ``floor``."""
    q = (np.asarray(t_final, dtype=np.float64) - t0) / dt
    return (np.floor(q * (1.0 + 8.0 * np.finfo(np.float64).eps) + 1e-12).astype(np.int64) + 1).astype(np.int32)


def build_batch(ids, t_final, B_fine, seed, s, table_rows=None):
    """Guess, weights and the ragged batch of trials `ids` (global indices) whose horizon was found
    (src/monte_carlo.jl:145-193)."""
    T = len(ids)
    n_knots = knot_counts(t_final, s.t0, s.dt)                                               # length(t0:0.2:t_final)
    N = int(n_knots.max())
    J = INERTIA[s.inertia]
    x0 = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])                                       # (:107-110)
    xf = np.array([0.0, 0.0, 0.0, np.sqrt(2.0) / 2.0, np.sqrt(2.0) / 2.0, 0.0, 0.0])         # (:113-115)
    # eigen-axis guess + Bryson weights of every trial (:161-176): the library's batched host routine (per trial in Python: the
    # same numbers from eigen_axis_slew + bryson_weights, 0.15 ms each)
    Qd, Qfd, Rd = bryson_weights_ragged(x0, xf, n_knots, s.t0, s.dt, J, s.alpha, s.beta)
    U0 = np.zeros((T, N - 1, 3))
    for j, i in enumerate(ids):
        n = int(n_knots[j])
        r = trial_rng(seed, i)
        r.random(2)                                                                          # the orbit draws
        U0[j, : n - 1] = r.random((n - 1, 3)) / 1000.0                                       # (:193)
    rows_per_s = s.N / (t_final - s.t0) if s.field_rate == "physical" else np.full(T, s.N / (s.tf - s.t0))
    c = np.ascontiguousarray
    # the table has 2N rows over 2 t_final (src/magnetic_toolbox.jl:73); the solve reads rows 0 .. N (+ the stage rows of
    # the last step), so only those travel to the solver
    # (B_fine None: the tables are on the device, `table_rows` rows each, and the upload packs them there)
    n_rows = min(B_fine.shape[1] if B_fine is not None else int(table_rows), s.N + 8)
    assert np.all((n_knots.astype(np.float64) - 1.0) * s.dt * rows_per_s < n_rows - 1)     # last row any RK stage reads
    b = SlewBatch(N=N, n_tab=n_rows, x0=c(np.tile(x0, (T, 1))), xf=c(np.tile(xf, (T, 1))),
                  Btab=c(B_fine[:, :n_rows]) if B_fine is not None else None,
                  btab_idx=np.arange(T, dtype=np.int32), tau0=np.zeros(T), dtau=c(s.dt * rows_per_s), dt=np.full(T, s.dt),
                  Jmat=c(np.tile(jmat_cm(J), (T, 1))), Qd=Qd, Qfd=Qfd, Rd=Rd, ulo=np.full((T, 3), -s.u_bnd),
                  uhi=np.full((T, 3), s.u_bnd), U0=U0)
    b.n_knots = n_knots if np.any(n_knots != N) else None
    b.meta = dict(name="monte_carlo_full", seed=seed)
    return b, n_knots


def draw_tracking_inputs(batch, ids, seed):
    """x0_lqr (src/monte_carlo.jl:203-211) per trial. The draws `simulator` makes inside the dynamics
    (src/simulator.jl:5,10,22) are generated by the tracking kernel itself, keyed by (seed, trial index)."""
    x0s = np.empty((batch.T, 7))
    for j, i in enumerate(ids):
        r = np.random.Generator(np.random.PCG64([int(seed), int(i), 1]))
        x0s[j] = tr.perturbed_initial_state(batch.x0[j:j + 1], r)[0]
    return x0s


def run_trials(stages, seed, lo, hi, setup=None, keep_trajectories=True):
    """Trials [lo, hi) as one batch. Returns a dict: A, t_final, n_knots, found (horizon exists), slew_time, fails,
    solve/tracking stats and — if asked — the per-trial arrays the script keeps (states, control_inputs, sim_states,
    sim_control_inputs, B_ECI_total, t_total; src/monte_carlo.jl:52-58, 199-235)."""
    s = setup or MonteCarloSetup()
    A = draw_orbits(seed, lo, hi, s)
    T = hi - lo
    resident = bool(getattr(stages, "resident_tables", False))      # GPU stages: tables stay on the device between the calls
    B_init = stages.magnetic_simulation(A, s.t0, s.tf, s.N, s, host=False) if resident else stages.magnetic_simulation(A, s.t0, s.tf, s.N, s)   # (:134)
    idx = stages.condition_based_time(B_init, (s.tf - s.t0) / s.N, s.cutoff)                  # (:137-140)
    t_final = idx.astype(np.float64) * (s.tf - s.t0) / s.N
    found = (idx > 0) & (t_final - s.t0 >= 2 * s.dt)
    out = dict(A=A, t_final=t_final, tf_index=idx, found=found, slew_time=np.array(t_final), fails=np.ones(T, dtype=np.int32),
               n_knots=np.zeros(T, dtype=np.int32), selected=np.nonzero(found)[0])
    sel = out["selected"]
    if len(sel) == 0:
        return out
    ids = lo + sel
    if resident:     # (:149) the script keeps B_ECI_total: downloaded once if the caller keeps trajectories, never uploaded again
        B_fine = stages.magnetic_simulation(A[sel], s.t0, t_final[sel], s.N, s, host=keep_trajectories)
        batch, n_knots = build_batch(ids, t_final[sel], None, seed, s, table_rows=2 * s.N)
    else:
        B_fine = stages.magnetic_simulation(A[sel], s.t0, t_final[sel], s.N, s)
        batch, n_knots = build_batch(ids, t_final[sel], B_fine, seed, s)
    # summaries only (keep_trajectories = False) on resident stages: neither the solved nor the tracked trajectories come back
    kw = {} if (keep_trajectories or not resident) else dict(want_trajectories=False)
    res = stages.solve(batch, s, **kw)
    Qd, Qfd, Rd = tr.tvlqr_weights(batch.T, s.lqr_alpha, s.lqr_beta, s.lqr_r)
    x0s = draw_tracking_inputs(batch, ids, seed)
    tv = stages.attitude_simulation(batch, res["X"], res["U"], x0s, Qd, Qfd, Rd, int(seed), ids.astype(np.int64), s, **kw)
    out["n_knots"][sel] = n_knots
    out["slew_time"][sel] = tv["stats"]["slew_time"]
    out["fails"][sel] = tv["stats"]["failed"]
    out["solve_stats"], out["tracking_stats"] = res["stats"], tv["stats"]
    if keep_trajectories:
        k = lambda a, j, n: a[j, :n].T      # Julia-shaped views (7 x n_i, 3 x (n_i - 1)) of the batch arrays, no copies
        out["t_total"] = [s.t0 + s.dt * np.arange(n) for n in n_knots]
        out["states"] = [k(res["X"], j, n) for j, n in enumerate(n_knots)]
        out["control_inputs"] = [k(res["U"], j, n - 1) for j, n in enumerate(n_knots)]
        out["sim_states"] = [k(tv["X_sim"], j, n) for j, n in enumerate(n_knots)]
        out["sim_control_inputs"] = [k(tv["U_sim"], j, n - 1) for j, n in enumerate(n_knots)]
        out["B_ECI_total"] = [B_fine[j] for j in range(len(sel))]
    return out


def summarize(out):
    """slew_time_mean over the trials with slew_time > 0, and the failed trials (src/monte_carlo.jl:310-330)."""
    st = np.asarray(out["slew_time"], dtype=np.float64)
    pos = st > 0.0
    return dict(slew_time_mean=float(st[pos].mean()) if np.any(pos) else 0.0, fails=np.nonzero(np.asarray(out["fails"]) == 1)[0],
                number_sims=int(st.shape[0]))


def monte_carlo(stages, number_sims=100, seed=0, setup=None, rank=0, world=1, group=None, chunk=1024, device=None):
    """The whole experiment over `world` ranks: each rank runs a contiguous block of trials in chunks (1024 = one
    wavefront per SIMD of an MI355X), then the
    per-trial summaries (A, t_final, slew_time, fails) are all-gathered in trial order (the result lists the script
    appends to, src/monte_carlo.jl:60-66). Trajectories stay on the rank that produced them."""
    from .sweep import shard_range

    lo, hi = shard_range(number_sims, rank, world)
    parts = [run_trials(stages, seed, a, min(a + chunk, hi), setup) for a in range(lo, hi, chunk)]
    loc = {k: np.concatenate([p[k] for p in parts]) for k in ("A", "t_final", "slew_time", "fails", "n_knots")} if parts else None
    if world == 1:
        return dict(loc, parts=parts)
    import torch
    import torch.distributed as dist

    per = -(-number_sims // world)
    pack = np.zeros((per, 10))
    n = hi - lo
    pack[:n, :6], pack[:n, 6], pack[:n, 7], pack[:n, 8], pack[:n, 9] = loc["A"], loc["t_final"], loc["slew_time"], loc["fails"], loc["n_knots"]
    t = torch.from_numpy(pack)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    rows = np.concatenate([o.cpu().numpy()[: shard_range(number_sims, r, world)[1] - shard_range(number_sims, r, world)[0]]
                           for r, o in enumerate(outs)])
    return dict(A=rows[:, :6], t_final=rows[:, 6], slew_time=rows[:, 7], fails=rows[:, 8].astype(np.int32),
                n_knots=rows[:, 9].astype(np.int32), parts=parts)
