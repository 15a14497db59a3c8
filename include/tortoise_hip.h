/*
 * tortoise_hip.h — C ABI of the MI355X-native batched AL-iLQR attitude-slew solver.
 *
 * This is the drop-in boundary for the ONE hot path of RoboticExplorationLab/TortoiseSat.jl:
 * the `TrajectoryOptimization.solve!(prob, solver)` call of src/TortoiseSat.jl:199 (and the old-API
 * `solve(solver,U)` of src/monte_carlo.jl:196) together with the dynamics callback it drives
 * (src/DerivFunction.jl:1-48).  A host (Julia `ccall`, Python `ctypes`, C++) builds the problem exactly
 * as the reference scripts do and then hands a whole *batch* of independent slews to `tsat_solve_batch`.
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no callbacks, no torch/HIP types in signatures;
 *   - every function returns 0 on success, <0 on API error (text via tsat_last_error);
 *     per-trajectory outcomes are reported in tsat_stats.status, never as an error code;
 *   - arrays are Julia/Fortran column-major with the component fastest, then knot, then trajectory:
 *     X is reshape(X,7,N,T); U is reshape(U,3,N-1,T); K is reshape(K,3,7,N-1,T);
 *   - state x = [omega(3); q(4, scalar first)] — the reference's 8th "time" state
 *     (src/DerivFunction.jl:6,44) is folded into (tau0, dtau): the B-table row used at knot k, RK stage
 *     offset c in {0, 1/2, 1} is floor(fma(k + c, dtau, tau0)), clamped to [0, n_tab-1]
 *     (reference: B_ECI[floor(Int, t*N + 1), :], src/DerivFunction.jl:28);
 *   - controls are in units of u_scale A*m^2 (u_scale = 1e-2, src/DerivFunction.jl:37);
 *   - the library is synchronous and keeps no caller pointers after a call returns;
 *   - one handle per (host thread, GPU). Handles are not thread-safe.
 *
 * There is NO CPU fallback behind this ABI: if no gfx950 device is usable, tsat_create fails.
 */
#ifndef TORTOISE_HIP_H
#define TORTOISE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSAT_NX 7 /* omega(3) + quaternion(4)            (src/DerivFunction.jl:4-5)   */
#define TSAT_NU 3 /* magnetic dipole command (3)         (src/DerivFunction.jl:37)    */
#define TSAT_NC 6 /* control box rows per knot: u<=uhi, u>=ulo (src/TortoiseSat.jl:178) */
#define TSAT_MAX_LINESEARCH 32

/* per-trajectory outcome codes (tsat_stats.status) */
enum {
  TSAT_CONVERGED = 0,   /* c_max < constraint_tol after an inner solve                       */
  TSAT_MAX_OUTER = 1,   /* outer budget exhausted (src/TortoiseSat.jl:196 `iterations`)      */
  TSAT_REG_FAIL  = 2,   /* backward-pass regularisation exceeded reg_max                     */
  TSAT_DIVERGED  = 3    /* non-finite cost in the initial rollout (no backward sweep ran: K of
                           such a trajectory is undefined, X / U hold the overflowed rollout) */
};

/*
 * Solver options. Replaces AugmentedLagrangianSolverOptions{Float64}() + .opts_uncon
 * (src/TortoiseSat.jl:194-196) and solver.opts.* of the old API (src/monte_carlo.jl:187-192).
 * Field defaults (tsat_default_options) follow the published AL-iLQR/ALTRO algorithm, SURVEY.md App. A.
 */
typedef struct tsat_options {
  int32_t n_knots;          /* N, knot points incl. terminal (src/TortoiseSat.jl:85)                */
  int32_t n_tab;            /* rows in each B table                                                 */
  int32_t integrator;       /* 3 = rk3 (src/TortoiseSat.jl:146), 4 = rk4 (src/attitude_controller.jl:122) */
  int32_t precision;        /* 64 (fp64) or 32 (mixed precision, BASELINE.json configs[2]): float LINEARISATION — the   */
                            /* Jacobian lanes and the knot records they leave for the Riccati recursion — while the      */
                            /* roll-out, the costs, the recursion, the gains and every array in HBM stay double, so the   */
                            /* line-search decisions follow the fp64 path; tsat_mpc_run takes 64 only                     */
  int32_t max_outer;        /* opts_al.iterations            (src/TortoiseSat.jl:196; 20)           */
  int32_t max_inner;        /* opts_al.opts_uncon.iterations (src/TortoiseSat.jl:195; 50)           */
  int32_t max_linesearch;   /* backtracking trials alpha = 2^-j, j < max_linesearch (<= 32)         */
  int32_t dj_counter_limit; /* solver.opts.dJ_counter_limit  (src/monte_carlo.jl:191)               */
  double  cost_tol;         /* inner: stop when 0 < dJ < cost_tol                                   */
  double  grad_tol;         /* inner: stop when mean_k max_i |d|/(|u|+1) < grad_tol                 */
  double  constraint_tol;   /* outer: stop when c_max < constraint_tol                              */
  double  penalty_init, penalty_scale, penalty_max, dual_max;
  double  reg_init, reg_scale, reg_min, reg_max, reg_fp;
  double  ls_lower, ls_upper;
  double  max_state;        /* rollout rejected when |x_i| or |u_i| exceeds this                     */
  double  u_scale;          /* 1e-2 (src/DerivFunction.jl:37)                                        */
  int32_t terminal_mask;    /* bit i set: state component i has a terminal equality x_N[i]=xf[i]
                               (goal_constraint, src/TortoiseSat.jl:182,188)                         */
  int32_t error_state;      /* 0: plain 7-state differences (Model(DerivFunction,8,3), src/TortoiseSat.jl:145)
                               1: the quaternion hooks Model(f!,n,m,quaternion_error,quaternion_expansion)
                                  (src/monte_carlo.jl:158): state difference [dw; MRP(q^-1 (x) q_new)]
                                  (src/quaternion_toolbox.jl:58-75), dynamics blocks E(q_{k+1})'AE(q_k), E(q_{k+1})'B
                                  (src/attitude_controller.jl:59-81), cost expansion projected through E(q)
                                  (src/quaternion_toolbox.jl:15-50); gains K act on the 6 error coordinates and
                                  are returned with a zero 7th column */
} tsat_options;

/* per-trajectory result record */
typedef struct tsat_stats {
  int32_t status;       /* TSAT_* code                                                    */
  int32_t outer_iters;  /* AL outer iterations executed                                   */
  int32_t inner_iters;  /* total iLQR iterations over all outer iterations                */
  int32_t ls_trials;    /* line-search candidates a sequential backtracking search would
                           have rolled out (index of accepted alpha + 1, summed)          */
  int32_t n_backward;   /* backward sweeps executed (incl. regularisation restarts)       */
  int32_t n_forward;    /* forward sweeps executed on this backend                        */
  int32_t bp_restarts;  /* backward sweeps abandoned on a non-PD Quu                      */
  int32_t fp_fails;     /* line searches in which no alpha was accepted                   */
  double  cost;         /* LQR objective of the returned trajectory (no AL terms)         */
  double  cost_al;      /* augmented-Lagrangian cost at exit                              */
  double  c_max;        /* max constraint violation of the returned trajectory            */
  double  grad;         /* last Todorov gradient                                          */
} tsat_stats;

typedef struct tsat_handle tsat_handle;

/* library/ABI version: major*100 + minor */
int  tsat_version(void);

/* fill *o with defaults (rk3, 20 x 50 budget as src/TortoiseSat.jl:195-196, ALTRO constants) */
void tsat_default_options(tsat_options* o);

/* open GPU `device_id` (must be a gfx950 device); replaces AugmentedLagrangianSolver(prob,opts)
 * (src/TortoiseSat.jl:197) as the object that owns solver workspace. */
int  tsat_create(tsat_handle** h, int device_id);
int  tsat_destroy(tsat_handle* h);
const char* tsat_last_error(const tsat_handle* h);

/*
 * One-call drop-in for solve!(prob, solver) over a batch of T independent slews
 * (src/TortoiseSat.jl:199; loop body of src/monte_carlo.jl:118-235). Host pointers in, host pointers out.
 *   x0, xf  7 x T            initial / goal state              (src/TortoiseSat.jl:124,130)
 *   Btab    3 x n_tab x n_btab  ECI field tables [T]           (global B_ECI, src/TortoiseSat.jl:89)
 *   btab_idx T (or NULL)     table used by trajectory t (NULL: t, requires n_btab == T)
 *   tau0, dtau  T            table row of knot 0 and rows per knot (see header comment)
 *   dt      T                step [s]                          (src/TortoiseSat.jl:84)
 *   Jmat    9 x T            inertia, column-major 3x3         (src/input_parameters.jl:29-51)
 *   Qd,Qfd  7 x T, Rd 3 x T  diagonal LQR weights              (src/TortoiseSat.jl:157-169)
 *   ulo,uhi 3 x T            control box                       (src/TortoiseSat.jl:178)
 *   U0      3 x (N-1) x T    initial controls                  (initial_controls!, src/TortoiseSat.jl:191)
 *   X       7 x N x T  out;  U  3 x (N-1) x T out;  K 3 x 7 x (N-1) x T out (may be NULL);  stats T out
 */
int  tsat_solve_batch(tsat_handle* h, const tsat_options* o, int64_t T, int64_t n_btab,
                      const double* x0, const double* xf, const double* Btab, const int32_t* btab_idx,
                      const double* tau0, const double* dtau, const double* dt, const double* Jmat,
                      const double* Qd, const double* Qfd, const double* Rd,
                      const double* ulo, const double* uhi, const double* U0,
                      double* X, double* U, double* K, tsat_stats* stats);

/*
 * Resident-batch API (what tsat_solve_batch is made of). Lets a caller keep a batch in HBM, re-run it,
 * time only the solve, and export results straight into device buffers it owns (e.g. for an RCCL all-gather).
 */
int  tsat_batch_reserve(tsat_handle* h, int64_t T, int32_t n_knots, int32_t n_tab, int64_t n_btab,
                        int32_t max_linesearch);
int  tsat_batch_upload(tsat_handle* h,
                       const double* x0, const double* xf, const double* Btab, const int32_t* btab_idx,
                       const double* tau0, const double* dtau, const double* dt, const double* Jmat,
                       const double* Qd, const double* Qfd, const double* Rd,
                       const double* ulo, const double* uhi, const double* U0);
/* Ragged batches: give trajectory t its own knot count n_knots[t] in [2, N] (NULL: all N again). Replaces the
 * per-run horizon of the reference's Monte-Carlo, t_total[i] = 0:0.2:t_final[i] (src/monte_carlo.jl:140-145).
 * Arrays keep the common stride N; U0 entries beyond a trajectory's horizon are ignored and its X/U/K slabs are
 * returned zero-filled beyond it. Call after tsat_batch_upload. */
int  tsat_batch_knots(tsat_handle* h, const int32_t* n_knots /* T */);
/* run the solve on the resident batch; blocks until done. It starts from the RESIDENT x0 / tau0 / U0: those are the uploaded
 * ones unless tsat_mpc_run has advanced them since (see there) — upload again to start over.
 * *kernel_ms (may be NULL) receives the HIP-event time of the solve kernel on the handle's stream. */
int  tsat_batch_run(tsat_handle* h, const tsat_options* o, float* kernel_ms);
/* unpack results to HOST buffers (any may be NULL) */
int  tsat_batch_download(tsat_handle* h, double* X, double* U, double* K, tsat_stats* stats);
/* unpack results to DEVICE buffers owned by the caller on the same GPU (any may be NULL); blocks */
int  tsat_batch_export_device(tsat_handle* h, void* X_dev, void* U_dev, void* K_dev, void* stats_dev);
/* bytes of HBM currently reserved by the handle for the resident batch */
int64_t tsat_batch_bytes(const tsat_handle* h);
/* The stages around the solve (downloads, gathers, tracking, horizon, field tables, receding-horizon history) keep grow-only
 * device workspaces in the handle so that repeated calls pay no allocation; after ONE large download or gather they hold
 * multi-GB staging buffers for the life of the handle. tsat_workspace_bytes reports them; tsat_workspace_trim frees them:
 *   what = 0  the staging buffers of tsat_batch_download and tsat_sweep_allgather only
 *   what = 1  every workspace — including the resident field tables of tsat_btable_batch(Btab = NULL), which are then gone
 * They are allocated again on the next call that needs them. */
int64_t tsat_workspace_bytes(const tsat_handle* h);
int  tsat_workspace_trim(tsat_handle* h, int32_t what);
/* optional per-iteration trace for debugging parity: rows of 8 doubles
 * [outer, inner, J_prev, J_new, alpha_index(-1 = none), rho, dV1, dV2], `rows` per trajectory.
 * Pass rows = 0 to disable. Call before tsat_batch_run; read back with tsat_batch_trace_download. */
int  tsat_batch_trace(tsat_handle* h, int32_t rows);
int  tsat_batch_trace_download(tsat_handle* h, double* trace /* 8 x rows x T */);

/* ------------------------------------------------------------------------------------------------------------
 * Closed-loop tracking of a solved slew (the caller right after solve! in the reference): TVLQR gains around the
 * optimised trajectory + simulation + the slew-time statistic.
 *   attitude_simulation(f!, f_gains!, :rk4, X, U, dt, x0_lqr, t0, tf, Q_lqr, R_lqr, Qf_lqr)   src/attitude_controller.jl:1-48
 *   attitude_lqr (Jacobians + G(q) reduction + Riccati)                                         src/attitude_controller.jl:50-119
 *   simulator / gain_simulator (plant with / without injected noise)                            src/simulator.jl, src/gain_simulator.jl
 *   slew-time / failure statistic                                                               src/monte_carlo.jl:242-262
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct tsat_tvlqr_options {
  int32_t n_knots;          /* N                                                                              */
  int32_t n_tab;            /* rows in each B table                                                           */
  int32_t linearize_dt_sq;  /* 1: gains linearised over a step of dt^2 (the reference's `dt = S[end]^2`,
                               src/attitude_controller.jl:137); 0: over dt                    */
  int32_t min_steps;        /* statistic ignores the first min_steps samples (10, src/monte_carlo.jl:251)     */
  double  u_scale;          /* 1e-2: `u/100` (src/gain_simulator.jl:42)                                       */
  double  w_tol;            /* 0.05 rad/s   (src/monte_carlo.jl:70)                                           */
  double  angle_tol;        /* 0.08727 rad  (src/monte_carlo.jl:71)                                           */
  int32_t noise_mode;       /* 0: the `noise` array (NULL = noise-free plant); 1: drawn inside the kernel, below */
  int32_t rate_as_written;  /* (a reserved word before version 300: values other than 0 / 1 are rejected — fill the struct with
                               tsat_tvlqr_default_options first.) statistic: 0 (default) |w| of sample j; 1 the line as the reference has it,
                               `omega_norm_vec[j] = norm(sim_states[i][1:3,i])` (src/monte_carlo.jl:247): for every j the
                               rate of sample i, the 1-based number of the trial (`noise_id` + 1, or the position in the
                               batch + 1) — past the end of a shorter trajectory Julia raises a BoundsError, here the
                               last sample is taken                                                           */
  uint64_t noise_seed;      /* key of the counter-based generator                                             */
  double  sigma_gyro;       /* (0.38 deg)^2: `randn(3,1)*(.38*pi/180)^2`, src/simulator.jl:5                  */
  double  sigma_att;        /* (1 deg)^2:    `randn(3,1)*(1*pi/180)^2`,   src/simulator.jl:10                 */
  double  field_amp;        /* (1e-5)^2:     `rand(3)*1e-5^2`,            src/simulator.jl:22                 */
} tsat_tvlqr_options;

/* noise_mode = 1: the nine draws of plant evaluation (trajectory id, knot k, RK4 stage s) come from Philox4x32-10 with
 * key (noise_seed lo, hi) and counters (id lo, id hi, k, 4 s + j), j = 0, 1, 2 -> words w[j][0..3]:
 *   U(w) = (w + 0.5) 2^-32;  (z, z') = sqrt(-2 ln U(a)) (cos, sin)(2 pi U(b))          (Box-Muller)
 *   gyro   = sigma_gyro (z(w00,w01), z'(w00,w01), z(w02,w03))
 *   att    = sigma_att  (z'(w02,w03), z(w10,w11), z'(w10,w11))
 *   field  = field_amp  (U(w12), U(w13), U(w20))
 * id = noise_id[t] (NULL: t), so a sweep draws the same noise however it is sharded or chunked. The reference draws
 * from Julia's global generator inside `simulator`; any generator is as faithful, this one is reproducible. */

typedef struct tsat_tvlqr_stats {
  int32_t slew_index;       /* first 1-based sample j > min_steps with |w| < w_tol (rate_as_written: see the options) and
                               error angle < angle_tol; 0 = none                                              */
  int32_t failed;           /* 1 when no such sample exists (`fails[i]`, src/monte_carlo.jl:258-261)           */
  double  slew_time;        /* dt * slew_index, or dt * N when failed (`slew_time[i]`, :237,252)               */
  double  final_w_norm;     /* |w| of the last simulated sample                                                */
  double  final_angle;      /* error angle of the last simulated sample                                        */
} tsat_tvlqr_stats;

void tsat_tvlqr_default_options(tsat_tvlqr_options* o);

/*
 * Host pointers in and out, column-major as everywhere in this ABI.
 *   X 7xNxT, U 3x(N-1)xT        the optimised trajectories (outputs of tsat_solve_batch)
 *   xf 7xT                       goal state: its quaternion is `q_final` of the statistic
 *   Btab, btab_idx, tau0, dtau, dt, Jmat   as for tsat_solve_batch
 *   Qd, Qfd 6xT, Rd 3xT          diagonals of Q_lqr, Qf_lqr, R_lqr (src/TortoiseSat.jl:251-260)
 *   x0_sim 7xT                   perturbed initial state x0_lqr (src/TortoiseSat.jl:227-234)
 *   noise 9x4x(N-1)xT or NULL    per step and RK4 stage: gyro noise(3), attitude-noise rotation vector(3), field
 *                                noise(3) — the values the reference draws inside `simulator` (src/simulator.jl:5,10,22);
 *                                NULL = noise-free plant (`gain_simulator`)
 *   X_sim 7xNxT, U_sim 3x(N-1)xT, K_lqr 3x6x(N-1)xT (each may be NULL: only what is asked for travels back), stats T   outputs
 *   n_knots T or NULL            per-trajectory horizons as for tsat_batch_knots (`t_total[i]`, src/monte_carlo.jl:145):
 *                                trajectory t is tracked over its first n_knots[t] samples, the rest of its slabs is
 *                                zero and its statistic counts n_knots[t] samples; NULL = all N
 *   noise_id T or NULL           generator ids of the trajectories when options.noise_mode = 1
 */
int  tsat_tvlqr_batch(tsat_handle* h, const tsat_tvlqr_options* o, int64_t T, int64_t n_btab,
                      const double* X, const double* U, const double* xf,
                      const double* Btab, const int32_t* btab_idx, const double* tau0, const double* dtau,
                      const double* dt, const double* Jmat, const double* Qd, const double* Qfd, const double* Rd,
                      const double* x0_sim, const double* noise,
                      double* X_sim, double* U_sim, double* K_lqr, tsat_tvlqr_stats* stats,
                      const int32_t* n_knots, const int64_t* noise_id);

/* Build of the solve kernel used by tsat_batch_run / tsat_solve_batch / tsat_mpc_run. 0 = automatic (default), by batch size:
 *   1 wide    one trajectory per wavefront, one wavefront per SIMD (40 KB of LDS, full register file): up to 1024 trajectories;
 *   2 dense   one trajectory per wavefront, two wavefronts per SIMD (20 KB, 256 registers): 1025 .. 2047;
 *   3 packed  four trajectories per wavefront share every forward sweep (16 line-search candidates each) and run their
 *             backward sweeps together (Jacobian lanes = trajectory x knot x column quarter, Riccati recursion on 16 lanes per
 *             trajectory), two wavefronts per SIMD: never automatically;
 *   4 packed8 the same with eight trajectories per wavefront (eight candidates each, two backward passes): 8193 .. 16383;
 *   5 packed8w  eight trajectories per wavefront at ONE wavefront per SIMD (40 KB of LDS: twelve of a backward pass's sixteen knot
 *             records stay on the chip instead of four, all sixteen float ones; the whole register file: no spills): 4097 .. 8192;
 *   6 packed16w sixteen trajectories per wavefront (four candidates each) at one wavefront per SIMD: from 16384, unless the
 *             iteration budget max_outer * max_inner is 100 or more (then packed8: a wavefront lasts as long as its slowest trajectory).
 *   7 packed4w  four trajectories per wavefront at one wavefront per SIMD: 2048 .. 4096 (the receding-horizon loop of 4096).
 * precision = 32 has the dense layout and the packed ones (below 2048 trajectories the dense one; `variant` 1 means 2 there).
 * The builds of one precision give bit-identical results (X, U, K, iteration counts; `n_forward` counts the sweeps a build
 * executed and differs). The switch exists for tuning and for the tests. */
int  tsat_set_kernel_variant(tsat_handle* h, int32_t variant);

/* Endgame of the packed builds (variants 3 to 7). The trajectories of one launch need different numbers of iterations (21 .. 150
 * on the inclination sweep of src/paper_images/heatmap.jl:139-185 with its 3 x 50 budget), and a wavefront that holds four or
 * eight of them lasts as long as its slowest: towards the end of a launch a few wavefronts with several long trajectories keep
 * running while the rest of the machine is idle. Once at most `suspend_at` trajectories of the batch are still iterating, every
 * wavefront therefore parks its live ones (state in HBM) and ends, and a second kernel, queued behind the first on the same
 * stream, continues each of them on a wavefront of its own.
 *   -1  automatic (default): an eighth of the batch, at most 2048, when max_outer * max_inner >= 20; never otherwise;
 *    0  never;  n > 0: at n live trajectories.
 * Results do not depend on it (X, U, K, costs, iteration counts: bit-identical); `n_forward`, the sweeps that were executed,
 * does. For tuning and for the tests. */
int  tsat_set_endgame(tsat_handle* h, int32_t suspend_at);

/* What the next tsat_batch_run / tsat_mpc_run with options `o` will launch on the reserved batch: *build = 1 wide, 2 dense,
 * 3 packed, 4 packed8, 5 packed8w, 6 packed16w, 7 packed4w (of the precision `o` names), *endgame_at = the live count at which a packed launch parks its trajectories
 * (0: no endgame). For reports (bench.py labels its lines with it); either pointer may be NULL. */
int  tsat_selected_build(tsat_handle* h, const tsat_options* o, int32_t* build, int32_t* endgame_at);

/* The same tracking for the RESIDENT batch right after tsat_batch_run: reference trajectories, field tables, table
 * clocks, inertias, goal states and per-trajectory horizons are the ones already on the device — nothing of the solve
 * travels back and forth between the two calls (src/monte_carlo.jl:196 -> :230). n_knots / n_tab of `o` are ignored. */
int  tsat_tvlqr_resident(tsat_handle* h, const tsat_tvlqr_options* o, const double* Qd, const double* Qfd,
                         const double* Rd, const double* x0_sim, const double* noise,
                         double* X_sim, double* U_sim, double* K_lqr, tsat_tvlqr_stats* stats, const int64_t* noise_id);

/* ------------------------------------------------------------------------------------------------------------
 * Receding-horizon re-solve on the RESIDENT batch (BASELINE.json configs[4]). NOT in the reference —
 * it tracks its plan with TVLQR (src/attitude_controller.jl:1-48); defined here as: n_steps times
 *   1. solve the horizon from the current x0 with the current initial controls, budget and options of `o`
 *      (exactly tsat_batch_run);
 *   2. record x_t = x0 and u_t = U[0];
 *   3. advance the noise-free plant one step with u_t: rk3 or rk4 (plant_integrator 3 | 4) of the model dynamics
 *      (src/DerivFunction.jl:1-48) over dt, table rows at the current table time;
 *   4. warm start of the next solve = the plan shifted by one knot, last control repeated; tau0 += dtau.
 * Nothing returns to the host between steps. tsat_mpc_run MUTATES the resident batch: afterwards it holds the advanced
 * x0 / tau0, the shifted warm start in place of the uploaded U0, and the last plan (tsat_batch_download); a further call
 * continues the simulation, tsat_batch_run re-solves from the advanced state, and tsat_tvlqr_resident tracks the last plan
 * from the advanced table clock (the handle's host copies of x0 / tau0 are refreshed from the device after the loop).
 *   X_hist 7 x (n_steps+1) x T   states x_0 .. x_{n_steps}
 *   U_hist 3 x n_steps x T       applied controls
 *   stats_last T (may be NULL)   statistics of the last solve
 *   solve_ms (may be NULL)       device time of the whole loop
 * ------------------------------------------------------------------------------------------------------------ */
int  tsat_mpc_run(tsat_handle* h, const tsat_options* o, int32_t n_steps, int32_t plant_integrator,
                  double* X_hist, double* U_hist, tsat_stats* stats_last, float* solve_ms);
/* executed counts of the last tsat_mpc_run, summed over its control steps on the device: per trajectory
 * [backward sweeps, forward sweeps, AL dual updates (outer_iters - 1), inner iterations]  (4 x T, int64) — what a measurement
 * needs to price the loop's memory traffic without bringing per-step statistics to the host */
int  tsat_mpc_tally(tsat_handle* h, int64_t* tally /* 4 x T */);

/* ------------------------------------------------------------------------------------------------------------
 * Sweep exchange across GPUs. The reference's Monte-Carlo is a serial loop whose iterations share nothing but the result
 * lists they append to (src/monte_carlo.jl:52-66, 199-235; src/paper_images/heatmap.jl:114-243). Here each rank — one
 * process, one handle, one GPU — solves a contiguous shard of equal size T, and ONE collective at the end appends
 * everybody's results in rank order: an RCCL all-gather over xGMI, issued on the handle's stream straight from the
 * resident batch (export kernel -> ncclAllGather, no host staging between them).
 *   tsat_comm_unique_id   rank 0 creates the 128-byte communicator id; the host distributes it to the other ranks by
 *                         whatever it has (MPI.bcast, a shared file, Distributed.jl, torch.distributed)
 *   tsat_comm_init        every rank, collectively: binds a communicator of `world` ranks to the handle's GPU
 *   tsat_sweep_allgather  every rank, collectively, after tsat_batch_run: X_all 7 x N x (world T), U_all 3 x (N-1) x (world T),
 *                         stats_all (world T) in rank order; any of the three may be NULL. on_device = 0: host buffers;
 *                         on_device = 1: device buffers of the caller on the handle's GPU. Blocks until the data is there.
 *   tsat_comm_destroy     releases the communicator (tsat_destroy does it too)
 *   tsat_comm_available   0 when RCCL can be loaded in this process, -11 otherwise: a LOCAL probe without any communication, so that
 *                         the ranks can agree (by their own means) to use this exchange before any of them enters the collective
 *                         tsat_comm_init — a rank that fails on its own would leave the others waiting inside ncclCommInitRank
 * RCCL is loaded at the first tsat_comm_* call (librccl.so.1, or $TSAT_RCCL_LIB); the library has no link-time dependency
 * on it, error code -11 reports that it is missing or that a collective failed. Every rank must hold a shard of the same
 * shape (T, N) — ragged totals: pad the last shard; tsat_sweep_allgather verifies it with a 16-byte all-gather whenever the shape
 * is new to the communicator and fails with -1 on EVERY rank when the shapes differ.
 * ------------------------------------------------------------------------------------------------------------ */
#define TSAT_COMM_ID_BYTES 128
int  tsat_comm_available(void);
int  tsat_comm_unique_id(void* id_out /* TSAT_COMM_ID_BYTES */);
int  tsat_comm_init(tsat_handle* h, const void* id /* TSAT_COMM_ID_BYTES */, int32_t rank, int32_t world);
int  tsat_sweep_allgather(tsat_handle* h, void* X_all, void* U_all, void* stats_all, int32_t on_device);
int  tsat_comm_destroy(tsat_handle* h);

/* ------------------------------------------------------------------------------------------------------------
 * Horizon selection (the caller right before the solve): cumulative magnetic Gramian of a field table and the first
 * row at which its condition number drops below `cutoff`.
 *   magnetic_gramian(B_N, dt)          src/magnetic_toolbox.jl:1-12   G_1 = hat(B_1)hat(B_1)', G_i = G_{i-1} + hat(B_i)hat(B_i)' dt
 *   condition_based_time(B_gram, cut)  src/magnetic_toolbox.jl:14-31  first 1-based i with cond(G_i) < cut, 0 if none
 * Call sites src/TortoiseSat.jl:73-82, src/monte_carlo.jl:137-140 (t_final = index * (tf - t0) / N).
 *   Btab 3 x n_rows x T (one coarse table per trajectory), dt_row T, cutoff T
 *   tf_index T out (1-based, 0 = never), cond_at T out (condition number at that row; may be NULL)
 * ------------------------------------------------------------------------------------------------------------ */
int  tsat_horizon_batch(tsat_handle* h, int64_t T, int32_t n_rows, const double* Btab, const double* dt_row,
                        const double* cutoff, int32_t* tf_index, double* cond_at);

/* ------------------------------------------------------------------------------------------------------------
 * Field-table generation (the first step of the reference's per-run setup): orbit from Keplerian elements, fixed-step
 * Euler propagation, GMST rotation, geocentric latitude/longitude, IGRF-12 (degree 13, epoch 2015 + secular variation),
 * NED -> ENU -> ECEF -> ECI.
 *   magnetic_simulation(p, t0, tf, N, mag_field)   src/magnetic_toolbox.jl:33-106   (2N rows, the last one left zero)
 *   kep_ECI(kep, t0, GM)                           src/kep_ECI.jl:1-35
 *   OrbitPlotter(x, p, t)                          src/OrbitPlotter.jl:1-52
 *   igrf12(date, r, lat, lon) (geocentric)         src/igrf.jl:70-274, src/legendre.jl:254-292, src/dlegendre.jl:221-309
 * Reference quirks are reproduced: the fixed IGRF radius (alt + R_E), the GMST expression as written, the J2 term as
 * written, element 6 used as mean anomaly.
 *   kep 6 x T   [e, a (km), i, RAAN, argp, anomaly] in degrees     t0, tf T   seconds
 *   Btab 3 x 2N x T out [Tesla, ECI] — directly usable as the `Btab` of tsat_solve_batch with n_tab = 2N
 *   pos  3 x (2N+1) x T out [km, ECI] (may be NULL)
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct tsat_btable_options {
  int32_t n_half;       /* N of magnetic_simulation: step (tf - t0)/N, 2N table rows                              */
  int32_t reserved;
  double  mjd;          /* p.MJD  (58155.0, src/TortoiseSat.jl:44)                                                */
  double  gm;           /* p.GM   km^3/s^2 (3.986004418e5, src/input_parameters.jl:26)                            */
  double  r_igrf_km;    /* alt + R_E = 400 + 6371 (src/magnetic_toolbox.jl:44,81)                                 */
  double  date;         /* decimal year handed to igrf12 (2019); 2015 <= date < 2020 is supported                 */
} tsat_btable_options;

void tsat_btable_default_options(tsat_btable_options* o);
/* ------------------------------------------------------------------------------------------------------------
 * Script arithmetic of the Monte-Carlo loop body, batched on the host (no GPU work): Bryson weights of the versine eigen-axis
 * guess (src/eigen_axis_slew.jl:1-38; Q = alpha / w_max^2, Qf = 10 Q, R = 1 / m_max^2 with the SIGNED maximum torque of the
 * guess as written, src/monte_carlo.jl:165-176) for T slews between the same two attitudes that differ only in their horizon
 * t = t0 : dt : t0 + (n_knots[t] - 1) dt. theta_f = rotation angle and axis(3) = rotation axis of the slew (computed once by
 * the caller), Jrm = inertia, row-major 3 x 3. Outputs Qd 7 x T, Qfd 7 x T, Rd 3 x T. Returns -2 when a guess is degenerate
 * (no positive torque sample), -1 on bad arguments (n_knots >= 3).
 * ------------------------------------------------------------------------------------------------------------ */
int  tsat_bryson_eigen_axis_batch(int64_t T, const int32_t* n_knots, double t0, double dt, double theta_f, const double* axis,
                                  const double* Jrm, double alpha, double beta, double* Qd, double* Qfd, double* Rd);

/* Resident field tables: with Btab = NULL, tsat_btable_batch leaves its tables on the device ([T][2 n_half][3], in the handle's
 * workspace) instead of downloading them; tsat_horizon_batch with Btab = NULL (same T, n_rows = 2 n_half) reads them there,
 * and tsat_batch_upload with Btab = NULL (n_btab = T, n_tab <= 2 n_half, btab_idx = NULL or identity) packs their first n_tab
 * rows into the solver's tables device to device — the Monte-Carlo's chain magnetic_simulation -> condition_based_time ->
 * magnetic_simulation -> solve! (src/monte_carlo.jl:134-196) without a table crossing PCIe.
 * The resident tables are the LAST call's, identified to the NULL-Btab consumers by shape only: any later tsat_btable_batch on
 * the handle replaces them (a failed or rejected call leaves none). tsat_btable_generation returns a counter that every
 * tsat_btable_batch call (and a trim of the workspace) bumps: a host that interleaves table generations records it after its
 * call and compares before the consuming call (tortoisesat.jl_amd/monte_carlo.py does). */
int64_t tsat_btable_generation(const tsat_handle* h);
int  tsat_btable_batch(tsat_handle* h, const tsat_btable_options* o, int64_t T, const double* kep, const double* t0,
                       const double* tf, double* Btab, double* pos);

#ifdef __cplusplus
}
#endif
#endif /* TORTOISE_HIP_H */
