"""Developer tool: the same 16384 x 1000 batch four times per build — do repeated runs give identical statuses and iteration
totals? (This is how the store-counted s_waitcnt of the packed forward sweep was caught: under full memory load a store can
retire before an older LDS copy.)   python tools/repeat_runs.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
if os.environ.get("TSAT_LIB"): pkg._abi.LIB_NAME = os.environ["TSAT_LIB"]
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss
T=16384; k=16
base = ss.workload_monte_carlo(T=1024, N=1000, seed=20190531, random_orbit=True)
rep = lambda a: np.ascontiguousarray(np.concatenate([a] * k))
b = ss.SlewBatch(base.N, base.n_tab, rep(base.x0), rep(base.xf), base.Btab, rep(base.btab_idx), rep(base.tau0), rep(base.dtau),
                 rep(base.dt), rep(base.Jmat), rep(base.Qd), rep(base.Qfd), rep(base.Rd), rep(base.ulo), rep(base.uhi), rep(base.U0))
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 5
opts.opts_uncon.iterations = 10; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
s.upload(b, o.max_linesearch)
for prec, var in ((32, 6), (32, 5), (32, 7), (32, 4), (32, 3), (64, 6), (64, 5), (64, 7), (64, 4), (64, 3)):
    o.precision = prec; s.set_kernel_variant(var)
    out=[]
    for r in range(int(os.environ.get("REPS", "4"))):
        ms = s.run(o); st = s.download(want_K=False)["stats"]
        out.append((round(ms,1), np.bincount(st["status"], minlength=4).tolist(), int(st["inner_iters"].sum()), float(st["cost"].sum())))
    print(prec, var, out, flush=True)
s.close()
