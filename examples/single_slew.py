#!/usr/bin/env python3
"""The reference's single-slew script (src/TortoiseSat.jl) through this repository's GPU path, call for call.

    python examples/single_slew.py            (needs an MI355X; ~1 s)

src/TortoiseSat.jl lines                       here
  34-44   orbit, epoch                          kep, MJD_0
  59-66   coarse field table over 1.5 orbits    magnetic.magnetic_simulation        (tsat_btable_batch)
  73-86   Gramian horizon, t_final, N           horizon.condition_based_time        (tsat_horizon_batch)
  89      field table over the horizon          magnetic.magnetic_simulation
  117-130 x0, xf                                90 deg about [1,0,1]/sqrt2 to identity
  145-146 Model(DerivFunction, n, m), rk3       trajopt.Model, trajopt.rk3
  157-169 Bryson weights, LQRObjective          slew_setup.bryson_weights, trajopt.LQRObjective
  178-188 BoundConstraint, goal_constraint      trajopt.BoundConstraint, trajopt.goal_constraint
  190-199 Problem, initial_controls!, solve!    trajopt.Problem, initial_controls_, solve_    (tsat_solve_batch)
  227-265 TVLQR tracking of the plan            tracking.attitude_simulation         (tsat_tvlqr_resident)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from tsat_loader import load_package  # noqa: E402

load_package()
from tortoisesat_jl_amd import horizon, magnetic, slew_setup as ss, tracking, trajopt as to  # noqa: E402


def main(verbose=True):
    say = print if verbose else (lambda *a, **k: None)
    solver = to.AugmentedLagrangianSolver(None, None)
    # orbit and field ----------------------------------------------------------------------------------------
    kep = np.array([[0.0, 400.0 + 6371.0, 51.6, 0.0, 0.0, 90.0]])              # ISS-like (:34-42)
    t0, tf, N_tab, cutoff, dt = 0.0, 5400.0, 5000, 20.0, 0.2
    B_coarse, _ = magnetic.magnetic_simulation(solver, kep, t0, tf, N_tab)
    idx, cond = horizon.condition_based_time(solver, B_coarse, (tf - t0) / N_tab, cutoff)
    t_final, N = horizon.knots_from_index(idx, tf - t0, N_tab, dt=dt)
    t_final, N = float(t_final[0]), int(N[0])
    say(f"Gramian condition number {cond[0]:.1f} < {cutoff} after {t_final:.0f} s -> {N} knots of {dt} s")
    B_ECI, _ = magnetic.magnetic_simulation(solver, kep, t0, t_final, N)         # 2N rows over 2 t_final (:89)
    # problem ------------------------------------------------------------------------------------------------
    n, m = 8, 3
    J = ss.INERTIA["1P"]
    q0 = ss.axis_angle_quat([1.0, 0.0, 1.0], np.deg2rad(90.0))
    x0 = np.r_[0.0, 0.0, 0.0, q0, 0.0]
    xf = np.r_[0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]
    model_d = to.rk3(to.Model(to.DerivFunction(J, B_ECI[0]), n, m))
    w_guess, _ = ss.eigen_axis_slew(x0[:7], xf[:7], dt * np.arange(N + 1))
    Qd, Qfd, Rd = ss.bryson_weights(w_guess, J, dt, 10.0, 1.0e3)
    Q = np.zeros((n, n)); Qf = np.zeros((n, n))
    Q[:7, :7], Qf[:7, :7] = np.diag(Qd), np.diag(Qfd)
    obj = to.LQRObjective(Q, np.diag(Rd), Qf, xf, N)
    constraints = to.Constraints(N)
    for k in range(1, N):
        constraints[k] += to.BoundConstraint(n, m, u_max=1, u_min=-1)
    constraints[N] += to.goal_constraint(xf)
    sat = to.Problem(model_d, obj, constraints=constraints, x0=x0, xf=xf, N=N, dt=dt)
    to.initial_controls_(sat, np.zeros((m, N + 1)))
    opts_al = to.AugmentedLagrangianSolverOptions()
    opts_al.opts_uncon.iterations, opts_al.iterations = 50, 20
    solver.opts = opts_al
    batch = to.BatchProblem([sat])
    to.solve_(batch, solver)
    st = sat.stats
    ang = lambda q: 2 * np.degrees(np.arccos(min(abs(float(q[0])) / np.linalg.norm(q), 1.0)))
    say(f"solve: status {st['status']}, {st['outer_iters']} outer / {st['inner_iters']} inner iterations, cost {st['cost']:.3f}, "
        f"max violation {st['c_max']:.2e}, {solver.last_kernel_ms:.1f} ms on the GPU; attitude error {ang(sat.X[3:7, 0]):.1f} -> {ang(sat.X[3:7, -1]):.2f} deg")
    # closed loop -------------------------------------------------------------------------------------------
    rng = np.random.default_rng(0)
    Ql, Qfl, Rl = tracking.tvlqr_weights(1)
    x0_lqr = tracking.perturbed_initial_state(batch.arrays.x0, rng)
    tv = tracking.attitude_simulation(solver, batch.arrays, None, None, x0_lqr, Ql, Qfl, Rl, noise_seed=1)
    s = tv["stats"][0]
    say(f"tracking with plant noise: slew {'not ' if s['failed'] else ''}completed"
        + ("" if s["failed"] else f" after {s['slew_time']:.1f} s") + f"; final rate {s['final_w_norm']:.2e} rad/s, final angle {np.degrees(s['final_angle']):.2f} deg")
    solver.close()
    return dict(N=N, t_final=t_final, stats=st, tracking=s, X=sat.X, U=sat.U)


if __name__ == "__main__":
    main()
