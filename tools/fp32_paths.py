"""Developer tool (GPU): iteration PATHS of the fp32 builds against the fp64 oracle on configs[2] inputs.

"Same path" in tests/test_gpu_fp32.py used to mean equal iteration and line-search COUNTS; two solves can have equal counts
and still accept different steps on the way (index 1 at iteration 3 and 0 at iteration 7 instead of the reverse). This tool
compares the per-iteration traces (accepted line-search index of every iteration) and prints, per build, how many
trajectories share the oracle's whole path, the worst |dX| among them, and for the worst equal-count trajectory the two
index sequences side by side — where its path leaves the oracle's.

    python tools/fp32_paths.py [T]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

from tsat_loader import load_package  # noqa: E402

pkg = load_package()
import oracle_lib as ol  # noqa: E402
from tortoisesat_jl_amd import magnetic as mg, slew_setup as ss, trajopt as to  # noqa: E402

ROWS = 64


def path_of(trace):
    """accepted index per iteration, (T, ROWS); rows beyond the last iteration are zero in every column"""
    used = trace[:, :, 1] > 0
    return np.where(used, trace[:, :, 4], -9).astype(np.int64)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 5
    opts.opts_uncon.iterations = 10; opts.opts_uncon.dJ_counter_limit = 1
    s = to.AugmentedLagrangianSolver(None, opts)
    b = mg.attach_igrf_tables(s, ss.workload_monte_carlo(T=T, N=1000, seed=20190531, random_orbit=True, tables=False))
    oo = ol.default_options()
    oo.max_outer, oo.max_inner, oo.dj_counter_limit, oo.error_state = 5, 10, 1, 1
    ref = ol.solve_batch(b, oo, nthreads=ol.num_procs(), want_K=False, trace_rows=ROWS)
    pr = path_of(ref["trace"])
    o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
    s.upload(b, o.max_linesearch)
    s.trace(ROWS)
    res = {}
    for prec, variant, name in ((64, 1, "fp64 wide"), (32, 12, "fp32 one trajectory per wave"), (32, 3, "fp32 packed"), (32, 4, "fp32 packed8")):
        o.precision = prec
        s.set_kernel_variant(variant)
        s.run(o)
        g = s.download(want_K=False)
        g["trace"] = s.trace_download()
        res[name] = g
        pg = path_of(g["trace"])
        same_path = np.all(pg == pr, axis=1)
        rs, gs = ref["stats"], g["stats"]
        same_counts = (rs["inner_iters"] == gs["inner_iters"]) & (rs["ls_trials"] == gs["ls_trials"])
        dX = np.max(np.abs(ref["X"] - g["X"]), axis=(1, 2))
        print(f"[{name}] same path {same_path.mean():.3f} (max |dX| on them {dX[same_path].max() if same_path.any() else 0:.2e}, "
              f"q90 {np.quantile(dX[same_path], 0.9) if same_path.any() else 0:.2e}); equal counts {same_counts.mean():.3f} "
              f"(max |dX| {dX[same_counts].max() if same_counts.any() else 0:.2e}); equal counts but another path: "
              f"{int(np.sum(same_counts & ~same_path))}")
        odd = same_counts & ~same_path
        if odd.any():
            t = int(np.flatnonzero(odd)[np.argmax(dX[odd])])
            n = int(rs["inner_iters"][t])
            print(f"    worst of those: trajectory {t}, |dX| {dX[t]:.2e}; accepted indices oracle {pr[t, :n].tolist()}")
            print(f"    {'':>44}this build {pg[t, :n].tolist()}")
            k = int(np.argmax(pg[t] != pr[t]))
            print(f"    first difference at iteration {k + 1}: J_prev oracle {ref['trace'][t, k, 2]:.10g} / build {g['trace'][t, k, 2]:.10g}; "
                  f"J oracle {ref['trace'][t, k, 3]:.10g} / build {g['trace'][t, k, 3]:.10g}")
    a, c = res["fp32 one trajectory per wave"], res["fp32 packed"]
    d = np.max(np.abs(a["X"] - c["X"]), axis=(1, 2))
    print(f"fp32 packed vs fp32 one-trajectory build: bit-identical on {np.mean(d == 0):.3f} of the trajectories; max |dX| {d.max():.2e}; "
          f"same path {np.mean(np.all(path_of(a['trace']) == path_of(c['trace']), axis=1)):.3f}")
    c8 = res["fp32 packed8"]
    print(f"fp32 packed8 vs packed: bit-identical {np.array_equal(c8['X'], c['X']) and np.array_equal(c8['U'], c['U'])}")
    s.close()


if __name__ == "__main__":
    main()
