"""Developer tool: parity of the GPU solve with the CPU oracle on the WHOLE bench workload (BASELINE.json configs[1]:
1024 trajectories x 1000 knots, 5 x 10 budget) and on the same workload with the quaternion hooks — every trajectory,
not the 96-trajectory sample bench.py times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
from tsat_loader import load_package
load_package()
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss
import oracle_lib as ol

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
what = sys.argv[2] if len(sys.argv) > 2 else "configs1"
if what == "configs3":      # a slice of the inclination sweep (BASELINE.json configs[3]): own table per trajectory, 3 x 50 budget
    b = ss.workload_inclination_sweep(T=T, N=1000, j0=20000)
    budget = (3, 50)
else:
    b = ss.workload_monte_carlo(T=T, N=1000)
    budget = (5, 10)
print(f"workload {what}, budget {budget[0]} x {budget[1]}; batches above 1024 trajectories run the dense build of the kernel", flush=True)
for es in (0, 1):
    opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = budget[0]
    opts.opts_uncon.iterations = budget[1]; opts.opts_uncon.dJ_counter_limit = 1
    s = to.AugmentedLagrangianSolver(None, opts)
    got = to.solve_(to.BatchProblem.from_arrays(b, error_state=es), s, want_K=False)
    s.close()
    o = ol.default_options(); o.max_outer, o.max_inner, o.dj_counter_limit, o.error_state = budget[0], budget[1], 1, es
    t0 = time.time()
    ref = ol.solve_batch(b, o, nthreads=ol.num_procs(), want_K=False)
    dt = time.time() - t0
    same = np.ones(T, dtype=bool)
    # n_forward is not compared: the kernel rolls all line-search candidates of an iteration out in ONE sweep, the oracle
    # one rollout per trial (ls_trials is the common count)
    for k in ("status", "outer_iters", "inner_iters", "ls_trials", "n_backward", "bp_restarts", "fp_fails"):
        eq = ref["stats"][k] == got["stats"][k]
        if not eq.all():
            print(f"  {k}: differs on {int((~eq).sum())} trajectories")
        same &= eq
    dX = np.max(np.abs(ref["X"] - got["X"]), axis=(1, 2)); dU = np.max(np.abs(ref["U"] - got["U"]), axis=(1, 2))
    print(f"error_state={es}: {T} trajectories; identical iteration/line-search/restart counts on {int(same.sum())}/{T}; "
          f"max|dX| {dX[same].max():.2e}, max|dU| {dU[same].max():.2e} on those (tolerance 1e-9); "
          f"rel. cost difference max {np.max(np.abs(ref['stats']['cost'][same] / got['stats']['cost'][same] - 1)):.1e}; "
          f"oracle {dt:.1f} s on {ol.num_procs()} threads", flush=True)
    if not same.all():
        bad = np.nonzero(~same)[0]
        print("  trajectories with different counts:", bad[:20], "inner (oracle, gpu):",
              list(zip(ref["stats"]["inner_iters"][bad][:10], got["stats"]["inner_iters"][bad][:10])))
