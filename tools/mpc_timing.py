"""Developer tool: BASELINE.json configs[4] shape on ONE GPU — a shard of trajectories x 200-knot horizon re-solved
every control step (1 x 3 budget, warm start = shifted plan, rk4 plant), resident loop of tsat_mpc_run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
if os.environ.get("TSAT_LIB"):            # developer switch: another build of the same library (experiments)
    pkg._abi.LIB_NAME = os.environ["TSAT_LIB"]
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss, mpc

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512        # 4096 over 8 GPUs
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
N = 200
b = ss.workload_monte_carlo(T=T, N=N, seed=20190602)
B = ss.dipole_btable(steps + N + 8, 0.2, 6771.0, 96.6)
b.Btab, b.n_tab = np.ascontiguousarray(B[None]), B.shape[0]
b.dtau[:] = 1.0
s = to.AugmentedLagrangianSolver(None, to.AugmentedLagrangianSolverOptions())
s.opts.opts_uncon.dJ_counter_limit = 1
VARIANT = int(os.environ.get("TSAT_VARIANT", "0"))
prob = to.BatchProblem.from_arrays(b)
mpc.receding_horizon(prob, s, 5)
if VARIANT:
    s.set_kernel_variant(VARIANT)
t0 = time.time()
r = mpc.receding_horizon(prob, s, steps, plant_integrator=4)
wall = time.time() - t0
d0 = np.linalg.norm(r["X_hist"][:, 0, 3:7] - b.xf[:, 3:7], axis=1).mean()
d1 = np.linalg.norm(r["X_hist"][:, -1, 3:7] - b.xf[:, 3:7], axis=1).mean()
print(f"T={T} horizon={N} steps={steps}: device loop {r['ms']:.1f} ms = {r['ms']/steps:.3f} ms/step "
      f"-> {T*steps/(r['ms']*1e-3):.0f} re-solves/s ({steps/(r['ms']*1e-3):.0f} control steps/s for the shard); wall {wall:.2f} s; "
      f"mean |q-qf| {d0:.3f} -> {d1:.3f}; last inner its {r['stats']['inner_iters'].mean():.2f}")
s.close()
