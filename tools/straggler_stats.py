"""Developer tool (GPU): iteration spread of a configs[3] shard — how the live trajectories of an 8192-trajectory shard thin out
over the joint iterations, and what the launch would cost if the live ones were kept packed (an upper bound for any compaction)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
from tortoisesat_jl_amd import magnetic as mg, slew_setup as ss, trajopt as to

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
J0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3 * 8192      # first global index of the shard (0 with T = 65536: the whole sweep)
ONLY_VARIANTS = len(sys.argv) > 3
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 3
opts.opts_uncon.iterations = 50; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
b = mg.attach_igrf_tables(s, ss.workload_inclination_sweep(T=T, N=1000, j0=J0, tables=False))
o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
s.upload(b, o.max_linesearch)
for v in ((0, 3, 4, 5, 6) if T > 8192 else (0, 2, 3, 4, 5, 6)):
    s.set_kernel_variant(v)
    ms = s.run(o); ms = s.run(o)
    print(f"variant {v}: {ms:.1f} ms -> {T / ms * 1e3:.0f} solves/s")
if ONLY_VARIANTS:
    s.close(); sys.exit(0)
s.set_kernel_variant(3)
for at in (0, 512, 1024, 1536, 2048, 3072, 4096):     # tsat_set_endgame: park the last `at` live trajectories for a one-per-wavefront launch
    s.set_endgame(at)
    ms = s.run(o); ms = s.run(o)
    print(f"packed, endgame at {at}: {ms:.1f} ms -> {T / ms * 1e3:.0f} solves/s")
s.set_endgame(-1); s.set_kernel_variant(0)
s.run(o)
st = s.download(want_K=False)["stats"]
it = st["inner_iters"].astype(int) + st["bp_restarts"]
print("status counts", np.bincount(st["status"], minlength=4), "; inner iterations: min / q10 / median / mean / q90 / max",
      it.min(), int(np.quantile(it, .1)), int(np.median(it)), f"{it.mean():.1f}", int(np.quantile(it, .9)), it.max())
print("ls trials per iteration", st["ls_trials"].sum() / st["inner_iters"].sum(), "; fp_fails", st["fp_fails"].sum(), "; n_forward per iteration", st["n_forward"].sum() / st["inner_iters"].sum())
live = np.array([(it > k).sum() for k in range(it.max() + 1)])
print("live trajectories after k iterations:", {k: int(live[k]) for k in range(0, it.max() + 1, 10)})
w4 = it.reshape(-1, 4).max(1)
print(f"packed as launched: mean of per-wave maxima {w4.mean():.1f}, slowest wave {w4.max()} joint iterations; sum of iterations {it.sum()}")
s.close()
