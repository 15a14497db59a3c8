"""Developer tool (GPU): stored line-search candidates per sweep of the packed builds against launch time (every stored candidate
is 80 B per knot and sweep of HBM writes; a winner that was not stored costs its trajectory one more sweep).

    python tools/store_probe.py [T=8192]        (runs the diagnostic build libtortoise_hip_prof.so: its launches read TSAT_PK_STORE / TSAT_PK_FEW)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
pkg._abi.LIB_NAME = os.environ.get("TSAT_PROF_LIB", "libtortoise_hip_prof.so")     # the TSAT_PK_* knobs exist in the diagnostic build only
from tortoisesat_jl_amd import magnetic as mg, slew_setup as ss, trajopt as to

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 3
opts.opts_uncon.iterations = 50; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
b = mg.attach_igrf_tables(s, ss.workload_inclination_sweep(T=T, N=1000, j0=3 * 8192, tables=False))
o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
s.set_endgame(0)
ref = None
for variant in (3, 4):
    for slots in (6, 4, 3, 2, 1):
        os.environ["TSAT_PK_STORE"] = str(slots)
        s.upload(b, o.max_linesearch)
        s.set_kernel_variant(variant)
        ms = s.run(o); ms = s.run(o)
        st = s.download(want_K=False)["stats"]
        same = ref is None or all(np.array_equal(st[f], ref[f]) for f in st.dtype.names if f != "n_forward")
        ref = st if ref is None else ref
        print(f"variant {variant}, {slots:2d} stored slots: {ms:.1f} ms; sweeps per iteration {st['n_forward'].sum() / st['inner_iters'].sum():.3f}; same results {same}", flush=True)
os.environ.pop("TSAT_PK_STORE", None)
# the rule that is built in: keep `few` roll-outs while the trajectory's line searches end early, all six after a deep one
for variant in (3, 4):
    for few in (6, 4, 3, 2):
        os.environ["TSAT_PK_FEW"] = str(few)
        s.upload(b, o.max_linesearch)
        s.set_kernel_variant(variant)
        ms = s.run(o); ms = s.run(o)
        st = s.download(want_K=False)["stats"]
        same = all(np.array_equal(st[f], ref[f]) for f in st.dtype.names if f != "n_forward")
        print(f"variant {variant}, {few} roll-outs kept while searches end early (6 after a deep one): {ms:.1f} ms; sweeps per iteration {st['n_forward'].sum() / st['inner_iters'].sum():.3f}; same results {same}", flush=True)
os.environ.pop("TSAT_PK_FEW", None)
w = st["ls_trials"].sum() / st["inner_iters"].sum()
print("mean accepted index + 1:", w)
s.close()
