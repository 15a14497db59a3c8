"""Developer tool: configs[2] (16384 x 1000, fp32) — the fp32 builds next to the fp64 packed build.  python tools/fp32_packed_timing.py [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
if os.environ.get("TSAT_LIB"):
    pkg._abi.LIB_NAME = os.environ["TSAT_LIB"]
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
k = T // 1024
base = ss.workload_monte_carlo(T=1024, N=1000, seed=20190531, random_orbit=True)
rep = lambda a: np.ascontiguousarray(np.concatenate([a] * k))
b = ss.SlewBatch(base.N, base.n_tab, rep(base.x0), rep(base.xf), base.Btab, rep(base.btab_idx), rep(base.tau0), rep(base.dtau),
                 rep(base.dt), rep(base.Jmat), rep(base.Qd), rep(base.Qfd), rep(base.Rd), rep(base.ulo), rep(base.uhi), rep(base.U0))
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 5
opts.opts_uncon.iterations = 10; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
s.upload(b, o.max_linesearch)
ref = None
for name, prec, var in (("fp64 packed", 64, 3), ("fp32 one trajectory per wave (4 waves/SIMD layout)", 32, 14), ("fp32 packed", 32, 3)):
    o.precision = prec
    s.set_kernel_variant(var)
    ms = [s.run(o) for _ in range(3)][1:]
    r = s.download(want_K=False)
    st = r["stats"]
    extra = ""
    if prec == 32:
        if ref is None:
            ref = r
        else:
            extra = f"; bit-identical to the one-trajectory fp32 build: {np.array_equal(ref['X'], r['X']) and np.array_equal(ref['U'], r['U'])}"
    print(f"{name}: T={T} kernel {np.mean(ms):.1f} ms -> {T/(np.mean(ms)*1e-3):.0f} solves/s; status {np.bincount(st['status'], minlength=4)}; "
          f"mean inner {st['inner_iters'].mean():.2f}; mean ls {st['ls_trials'].mean():.2f}{extra}", flush=True)
s.close()
