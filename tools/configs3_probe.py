"""Developer tool (GPU): worst trajectories of the configs[3] parity run, build by build, against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from tsat_loader import load_package
pkg = load_package()
import oracle_lib as ol
from tortoisesat_jl_amd import magnetic as mg, slew_setup as ss, trajopt as to

lo, hi = int(sys.argv[1]), int(sys.argv[2])
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 3
opts.opts_uncon.iterations = 50; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
full = mg.attach_igrf_tables(s, ss.workload_inclination_sweep(T=3072, N=1000, j0=20000, tables=False))     # the test's batch
b = full.slice(lo, hi)
b.Btab, b.btab_idx = np.ascontiguousarray(full.Btab[lo:hi]), np.arange(hi - lo, dtype=np.int32)
oo = ol.default_options(); oo.max_outer, oo.max_inner, oo.dj_counter_limit, oo.error_state = 3, 50, 1, 1
ROWS = 160
ref = ol.solve_batch(b, oo, nthreads=ol.num_procs(), want_K=False, trace_rows=ROWS)
o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
s.upload(b, o.max_linesearch); s.trace(ROWS)
res = {}
for v in (1, 2, 3):
    s.set_kernel_variant(v); s.run(o); g = s.download(want_K=False); g["trace"] = s.trace_download(); res[v] = g
    dX = np.max(np.abs(ref["X"] - g["X"]), axis=(1, 2))
    sc = np.maximum(1, np.max(np.abs(ref["U"]), axis=(1, 2)))
    dU = np.max(np.abs(ref["U"] - g["U"]), axis=(1, 2)) / sc
    w = int(np.argmax(dX))
    print(f"variant {v}: max|dX| {dX.max():.2e} at {lo + w} (iters {g['stats']['inner_iters'][w]}, outer {g['stats']['outer_iters'][w]}, status {g['stats']['status'][w]}), "
          f"max|dU|/scale {dU.max():.2e}; counts equal {np.array_equal(ref['stats']['inner_iters'], g['stats']['inner_iters'])}; "
          f"over 1e-9: {int(np.sum(dX >= 1e-9))} of {len(dX)}; q50/q99 {np.median(dX):.1e}/{np.quantile(dX, .99):.1e}")
    if v == 1:
        tr, tg = ref["trace"][w], g["trace"][w]
        n = int(g["stats"]["inner_iters"][w])
        rel = np.abs(tr[:n, 3] - tg[:n, 3]) / np.abs(tr[:n, 3])
        print("   rel |dJ| per iteration:", " ".join(f"{x:.0e}" for x in rel))
        print("   accepted idx:", tr[:n, 4].astype(int).tolist())
        print("   rho:", " ".join(f"{x:.1e}" for x in tr[:n, 5]))
print("builds bit-identical:", all(np.array_equal(res[1]["X"], res[v]["X"]) and np.array_equal(res[1]["U"], res[v]["U"]) for v in (2, 3)))
s.close()
