"""Quick GPU-vs-oracle check + timing (developer tool; the real tests are tests/test_gpu_*.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from tsat_loader import load_package
pkg = load_package()
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss
import oracle_lib as ol

def run(T, N, outer, inner, djl, integ=3, cmp_T=None, trace=0):
    b = ss.workload_monte_carlo(T=T, N=N)
    opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = outer
    opts.opts_uncon.iterations = inner; opts.opts_uncon.dJ_counter_limit = djl
    solver = to.AugmentedLagrangianSolver(None, opts)
    bp = to.BatchProblem.from_arrays(b, integrator=integ)
    o = opts.to_abi(b.N, b.n_tab, integ)
    solver.upload(b, o.max_linesearch)
    if trace: solver.trace(trace)
    t = time.time(); ms = solver.run(o); wall = time.time() - t
    res = solver.download()
    st = res["stats"]
    print(f"T={T} N={N} {outer}x{inner} rk{integ}: kernel {ms:.2f} ms wall {wall*1e3:.1f} ms -> {T/(ms*1e-3):.1f} solves/s; "
          f"HBM reserved {solver.reserved_bytes()/2**20:.0f} MiB")
    print("  status hist", np.bincount(st["status"], minlength=4), "inner mean", st["inner_iters"].mean(),
          "ls mean", st["ls_trials"].mean(), "bp_restarts", st["bp_restarts"].sum(), "fp_fails", st["fp_fails"].sum())
    ms2 = solver.run(o); print(f"  second run kernel {ms2:.2f} ms")
    if cmp_T:
        bs = b.slice(0, cmp_T)
        oo = ol.default_options(); oo.max_outer, oo.max_inner, oo.dj_counter_limit, oo.integrator = outer, inner, djl, integ
        t = time.time(); ref = ol.solve_batch(bs, oo, nthreads=ol.num_procs(), trace_rows=trace); tc = time.time() - t
        print(f"  oracle {cmp_T} solves in {tc:.2f} s on {ol.num_procs()} threads")
        for k in ("X", "U", "K"):
            dmax = np.max(np.abs(ref[k] - res[k][:cmp_T])); print(f"  max|d{k}| = {dmax:.3e} (scale {np.max(np.abs(ref[k])):.3e})")
        for f in ("status", "outer_iters", "inner_iters", "ls_trials", "n_backward", "bp_restarts", "fp_fails"):
            neq = int(np.sum(ref["stats"][f] != st[f][:cmp_T])); print(f"  stats.{f}: {neq} mismatches")
        print("  cost rel diff", np.max(np.abs(ref["stats"]["cost"] - st["cost"][:cmp_T]) / np.abs(ref["stats"]["cost"])))
        if trace:
            tr = solver.trace_download()
            bad = np.argwhere(np.abs(tr[:cmp_T, :, 4] - ref["trace"][:, :, 4]) > 0)
            print("  trace alpha mismatches:", bad[:5])
    solver.close()

if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "small"
    if mode == "small":
        run(8, 60, 3, 6, 1, cmp_T=8, trace=32)
        run(4, 100, 2, 4, 1, integ=4, cmp_T=4)
        run(64, 1000, 5, 10, 1, cmp_T=16, trace=64)
    elif mode == "bench":
        run(1024, 1000, 5, 10, 1, cmp_T=16)
