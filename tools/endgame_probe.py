"""Developer tool (GPU): the two kernels of a packed launch with an endgame (tsat_set_endgame) under `rocprofv3 --kernel-trace
--stats`: how long the packed kernel and the resume kernel take at each threshold on a configs[3] shard.

    rocprofv3 --kernel-trace --stats -d gpurun_out/x -- python3 tools/endgame_probe.py 8192 0 1024 2048
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
from tortoisesat_jl_amd import magnetic as mg, slew_setup as ss, trajopt as to

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ats = [int(x) for x in sys.argv[2:]] or [0, 2048]
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 3
opts.opts_uncon.iterations = 50; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
b = mg.attach_igrf_tables(s, ss.workload_inclination_sweep(T=T, N=1000, j0=3 * 8192, tables=False))
o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
s.upload(b, o.max_linesearch)
s.set_kernel_variant(3)
for at in ats:
    s.set_endgame(at)
    ms = s.run(o)
    st = s.download(want_K=False)["stats"]
    print(f"endgame at {at}: {ms:.1f} ms; n_forward sum {int(st['n_forward'].sum())}, iterations {int(st['inner_iters'].sum())}", flush=True)
s.close()
