"""Developer tool (GPU): wide vs dense build on the batch of test_gpu_dense_and_wide_builds_agree, field by field."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from tsat_loader import load_package
pkg = load_package()
import oracle_lib as ol, helpers
from tortoisesat_jl_amd import slew_setup as ss, trajopt as to
b = ss.workload_monte_carlo(T=24, N=300, seed=12, random_orbit=True)
o = ol.default_options(); o.max_outer, o.max_inner, o.dj_counter_limit = 3, 6, 1
s = to.AugmentedLagrangianSolver(None, None)
a = helpers.abi_options_like(o, pkg, b.N, b.n_tab)
out = {}
for rep in range(2):
    for v in (1, 2):
        s.set_kernel_variant(v); s.upload(b, a.max_linesearch); s.run(a); out[v] = s.download()
    w, d = out[1], out[2]
    bad = [t for t in range(b.T) if any(w["stats"][f][t] != d["stats"][f][t] for f in w["stats"].dtype.names if f != "n_forward")]
    print(f"rep {rep}: trajectories with differing stats: {bad}")
    for t in bad[:4]:
        print("  wide ", w["stats"][t]); print("  dense", d["stats"][t])
        print("  X equal", np.array_equal(w["X"][t], d["X"][t]), "U equal", np.array_equal(w["U"][t], d["U"][t]), "max|dX|", np.max(np.abs(w["X"][t] - d["X"][t])))
    print("  all X equal:", np.array_equal(w["X"], d["X"]), "all U equal:", np.array_equal(w["U"], d["U"]))
s.close()

# second part: the same after the handle's memory and the queue's scratch have been used by other launches (recycled, dirty)
s = to.AugmentedLagrangianSolver(None, None)
big = ss.workload_monte_carlo(T=512, N=1000, seed=3)
o2 = ol.default_options(); o2.max_outer, o2.max_inner, o2.dj_counter_limit, o2.error_state = 5, 10, 1, 1
a2 = helpers.abi_options_like(o2, pkg, big.N, big.n_tab)
ref = None
for v in (1, 3, 2, 2, 1):
    s.set_kernel_variant(v); s.upload(big, a2.max_linesearch); s.run(a2); r = s.download(want_K=False)
    st = r["stats"]
    ref = ref or r
    badt = np.flatnonzero(st["inner_iters"] != ref["stats"]["inner_iters"])
    if len(badt):
        print("   first bad trajectories:", badt[:20].tolist(), "... count", len(badt))
        raw = st.view(np.float64).reshape(len(st), 8)
        for t in badt[:3]:
            print("   stats[%d] as doubles:" % t, raw[t].tolist())
    print(f"dirtying run, variant {v}: status values {np.unique(st['status'])[:6]}, inner range {st['inner_iters'].min()}..{st['inner_iters'].max()}; "
          f"X equal to the first run {np.array_equal(r['X'], ref['X'])}, finite {np.all(np.isfinite(r['X']))}, stats equal on {np.mean(st['inner_iters'] == ref['stats']['inner_iters']):.3f}")
for rep in range(3):
    for v in (1, 2):
        s.set_kernel_variant(v); s.upload(b, a.max_linesearch); s.trace(0); s.run(a); out[v] = s.download()
    w, d = out[1], out[2]
    bad = [t for t in range(b.T) if any(w["stats"][f][t] != d["stats"][f][t] for f in w["stats"].dtype.names if f != "n_forward")]
    print(f"dirty rep {rep}: trajectories with differing stats: {bad}")
    for t in bad[:4]:
        print("  wide ", w["stats"][t]); print("  dense", d["stats"][t])
        print("  X equal", np.array_equal(w["X"][t], d["X"][t]), "U equal", np.array_equal(w["U"][t], d["U"][t]), "max|dX|", np.max(np.abs(w["X"][t] - d["X"][t])))
    print("  all X equal:", np.array_equal(w["X"], d["X"]), "all U equal:", np.array_equal(w["U"], d["U"]))
s.close()
