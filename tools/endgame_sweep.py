"""Developer tool (GPU): launch time against the endgame threshold (tsat_set_endgame) on BASELINE configs[2] (16384 x 1000, 5 x 10, both
precisions) and on a configs[3] shard (8192 x 1000, 3 x 50): what the automatic rule should be.   python tools/endgame_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
from tortoisesat_jl_amd import magnetic as mg, slew_setup as ss, trajopt as to

for name, T, prec in (("configs[3] shard", 8192, 64), ("configs[2]", 16384, 64), ("configs[2]", 16384, 32)):
    opts = to.AugmentedLagrangianSolverOptions()
    s = to.AugmentedLagrangianSolver(None, opts)
    if name.startswith("configs[3]"):
        b = mg.attach_igrf_tables(s, ss.workload_inclination_sweep(T=T, N=1000, j0=3 * 8192, tables=False))
    else:
        b = mg.attach_igrf_tables(s, ss.workload_monte_carlo(T=T, N=1000, seed=20190531, random_orbit=True, tables=False))
    opts.iterations, opts.opts_uncon.iterations, opts.opts_uncon.dJ_counter_limit = b.meta["max_outer"], b.meta["max_inner"], 1
    o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
    o.precision = prec
    s.upload(b, o.max_linesearch)
    row = []
    for at in (-1, 0, T // 32, T // 16, T // 8, 3 * T // 16, T // 4):
        s.set_endgame(at)
        ms = [s.run(o) for _ in range(3)][1:]
        row.append(f"{'auto' if at < 0 else at}: {np.mean(ms):.1f}")
    print(f"{name}, {T} x 1000, precision {prec}, build {s.selected_build(o)}: ms per launch by endgame threshold   " + "   ".join(row), flush=True)
    s.close()
