"""Developer tool: the builds of the solve kernel by batch size (where should the automatic choice switch?)
    python tools/threshold_timing.py [T ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
if os.environ.get("TSAT_LIB"):            # developer switch: another build of the same library (experiments)
    pkg._abi.LIB_NAME = os.environ["TSAT_LIB"]
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss
base = ss.workload_monte_carlo(T=1024, N=1000, seed=20190531, random_orbit=True)
SIZES = tuple(int(a) for a in sys.argv[1:]) or (1024, 2048, 3072, 4096, 6144, 8192)
for T in SIZES:
    k = (T + 1023) // 1024
    rep = lambda a: np.ascontiguousarray(np.concatenate([a] * k)[:T])      # (tiling the 1024 draws; a ragged last tile is cut)
    b = ss.SlewBatch(base.N, base.n_tab, rep(base.x0), rep(base.xf), base.Btab, rep(base.btab_idx), rep(base.tau0), rep(base.dtau),
                     rep(base.dt), rep(base.Jmat), rep(base.Qd), rep(base.Qfd), rep(base.Rd), rep(base.ulo), rep(base.uhi), rep(base.U0))
    opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 5
    opts.opts_uncon.iterations = 10; opts.opts_uncon.dJ_counter_limit = 1
    s = to.AugmentedLagrangianSolver(None, opts)
    o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
    s.upload(b, o.max_linesearch)
    out = []
    builds = [("wide", 64, 1), ("dense", 64, 2), ("packed", 64, 3), ("packed8", 64, 4), ("packed8w", 64, 5), ("packed16w", 64, 6), ("packed4w", 64, 7), ("mixed dense", 32, 2),
              ("mixed packed", 32, 3), ("mixed packed8", 32, 4), ("mixed packed8w", 32, 5), ("mixed packed16w", 32, 6), ("mixed packed4w", 32, 7)]
    if T > 4096: builds = [b for b in builds if b[2] > 2]
    if os.environ.get("TSAT_EXP_VARIANT"): builds = [("packed", 64, 3), ("experimental variant", 64, int(os.environ["TSAT_EXP_VARIANT"]))]
    ref = {}
    for name, prec, var in builds:
        o.precision = prec
        s.set_kernel_variant(var)
        ms = [s.run(o) for _ in range(2)][1:]
        res = s.download(want_K=False)
        same = ""
        if prec in ref: same = "" if (np.array_equal(ref[prec]["X"], res["X"]) and np.array_equal(ref[prec]["U"], res["U"])) else " DIFFERENT BITS"
        else: ref[prec] = res
        out.append(f"{name} {np.mean(ms):.1f} ms ({T/(np.mean(ms)*1e-3):.0f}/s){same}")
    print(f"T={T}: " + "; ".join(out), flush=True)
    s.close()
