"""Developer tool: the mixed-precision builds (options.precision = 32: float linearisation, everything else double) on
BASELINE.json configs[2] — 16384 trajectories x 1000 knots, random q0 + orbit, quaternion hooks, IGRF tables — timing next to the
fp64 builds, and accuracy against the fp64 CPU oracle on the first trajectories (incl. the fraction that follows the oracle's
iteration path: the same accepted line-search index in every iteration).   python tools/fp32_eval.py [T] [n_oracle]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
from tsat_loader import load_package
load_package()
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss, magnetic as mg
import oracle_lib as ol

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n_or = int(sys.argv[2]) if len(sys.argv) > 2 else 512
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 5
opts.opts_uncon.iterations = 10; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
t0 = time.time()
b = mg.attach_igrf_tables(s, ss.workload_monte_carlo(T=T, N=1000, seed=20190531, random_orbit=True, tables=False))
print(f"workload {T} x 1000 knots built in {time.time()-t0:.1f} s", flush=True)
o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
s.upload(b, o.max_linesearch)
res = {}
s.trace(64)
for name, prec, var in (("fp64 auto (packed16w)", 64, 0), ("fp64 packed, 4 per wave", 64, 3), ("mixed auto (packed16w)", 32, 0), ("mixed packed, 4 per wave", 32, 3),
                        ("mixed one trajectory per wave, 2 waves/SIMD", 32, 2)):
    o.precision = prec
    s.set_kernel_variant(var)
    ms = [s.run(o) for _ in range(3)][1:]
    st = s.download(want_K=False)
    st["trace"] = s.trace_download()[:n_or]
    res[name] = st
    print(f"{name}: kernel {np.mean(ms):.1f} ms -> {T/(np.mean(ms)*1e-3):.0f} solves/s; status {np.bincount(st['stats']['status'], minlength=4)}; "
          f"mean inner {st['stats']['inner_iters'].mean():.2f}; mean ls {st['stats']['ls_trials'].mean():.2f}", flush=True)
s.set_kernel_variant(0)
oo = ol.default_options(); oo.max_outer, oo.max_inner, oo.dj_counter_limit, oo.error_state = 5, 10, 1, 1
t0 = time.time()
ref = ol.solve_batch(b.slice(0, n_or), oo, nthreads=ol.num_procs(), want_K=False, trace_rows=64)
paths = lambda tr: np.where(tr[:, :, 1] > 0, tr[:, :, 4], -9).astype(np.int64)
print(f"oracle (fp64) on the first {n_or}: {time.time()-t0:.1f} s", flush=True)
for name, g in res.items():
    dX = np.max(np.abs(ref["X"] - g["X"][:n_or]), axis=(1, 2))
    dU = np.max(np.abs(ref["U"] - g["U"][:n_or]), axis=(1, 2)) / np.maximum(1.0, np.max(np.abs(ref["U"]), axis=(1, 2)))
    same = np.all(paths(ref["trace"]) == paths(g["trace"]), axis=1)
    stat = np.mean(ref["stats"]["status"] == g["stats"]["status"][:n_or])
    rc = np.abs(g["stats"]["cost"][:n_or] / ref["stats"]["cost"] - 1)
    q = lambda a: "/".join(f"{np.quantile(a, p):.1e}" for p in (0.5, 0.9, 0.99, 1.0))
    print(f"{name}: same iteration path {same.mean()*100:.1f} %; status agreement {stat*100:.1f} %; |dX| q50/90/99/max {q(dX)}; |dU|/scale {q(dU)}; "
          f"|dU| on same-path {q(dU[same]) if same.any() else '-'}; rel cost diff {q(rc)}; frac |dX|<1e-3 {np.mean(dX<1e-3):.3f}, |dU|/scale<1e-3 {np.mean(dU<1e-3):.3f}", flush=True)
s.close()
