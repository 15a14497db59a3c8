"""Developer tool: are the builds of the solve kernel bit-identical on the GPU?  python tools/variant_bits.py [T] [N] [es]
Prints, per pair of builds, the first knot (from the end) at which the gains differ after ONE backward sweep, and the full-solve
differences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
if os.environ.get("TSAT_LIB"):
    pkg._abi.LIB_NAME = os.environ["TSAT_LIB"]
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
es = int(sys.argv[3]) if len(sys.argv) > 3 else 1
b = ss.workload_monte_carlo(T=T, N=N, seed=3, random_orbit=True)
for outer, inner in ((1, 1), (3, 6)):
    opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = outer
    opts.opts_uncon.iterations = inner; opts.opts_uncon.dJ_counter_limit = 1
    s = to.AugmentedLagrangianSolver(None, opts)
    o = opts.to_abi(b.N, b.n_tab, 3, error_state=es)
    s.upload(b, o.max_linesearch)
    res = {}
    for v, name in ((1, "wide"), (2, "dense"), (3, "packed")):
        s.set_kernel_variant(v); s.trace(4); s.run(o); res[name] = s.download(want_K=True); res[name]["trace"] = s.trace_download()
    s.close()
    for other in ("dense", "packed"):
        dK = np.abs(res["wide"]["K"] - res[other]["K"]).max(axis=(0, 2, 3))
        nz = np.nonzero(dK)[0]
        print(f"{outer}x{inner} wide vs {other}: max|dX| {np.abs(res['wide']['X']-res[other]['X']).max():.2e} max|dU| {np.abs(res['wide']['U']-res[other]['U']).max():.2e} "
              f"max|dK| {dK.max():.2e} (|K| {np.abs(res['wide']['K']).max():.1e}); knots with differing gains: {len(nz)} of {N-1}, last {nz.max() if len(nz) else '-'}")
        if outer == 1:
            tw, tp = res["wide"]["trace"][:, 0], res[other]["trace"][:, 0]
            print("   trace row 0 [outer, it, Jprev, J, jw, rho, dV1, dV2] max abs diff:", np.abs(tw - tp).max(axis=0))
        if len(nz) and outer == 1:
            k = nz.max(); d = np.abs(res["wide"]["K"][:, k] - res[other]["K"][:, k]).max(axis=0)
            print("   per-entry max over trajectories at that knot (7 state columns x 3 controls):\n", d)
            print("   knots:", nz.tolist()[-12:])
