"""Developer tool (GPU): how evenly the 8 shards of configs[3] (65536 trajectories, 8192 per GPU) are loaded — launch time and
iteration counts of every shard, for contiguous blocks of the inclination sweep and for the sweep dealt out with stride 8.
With one process per GPU the step time is that of the SLOWEST shard.

    python tools/shard_balance.py [world=8]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
from tortoisesat_jl_amd import magnetic as mg, slew_setup as ss, trajopt as to

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = 65536 // W
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 3
opts.opts_uncon.iterations = 50; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
for name, kw in (("contiguous blocks", lambda r: dict(j0=r * T, stride=1)), (f"dealt out with stride {W}", lambda r: dict(j0=r, stride=W))):
    times, means = [], []
    for r in range(W):
        b = mg.attach_igrf_tables(s, ss.workload_inclination_sweep(T=T, N=1000, tables=False, **kw(r)))
        o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
        s.upload(b, o.max_linesearch)
        ms = s.run(o); ms = s.run(o)
        st = s.download(want_K=False)["stats"]
        it = st["inner_iters"].astype(int)
        times.append(ms); means.append(it.mean())
        print(f"  {name}, shard {r}: {ms:7.1f} ms, iterations mean {it.mean():5.1f}, q90 {int(np.quantile(it, .9))}, at the cap (150) {np.mean(it >= 150):.3f}", flush=True)
    print(f"{name}: slowest shard {max(times):.1f} ms, mean {np.mean(times):.1f} ms -> {65536 / max(times) * 1e3:.0f} solves/s on {W} GPUs "
          f"(balance {np.mean(times) / max(times):.2f})", flush=True)
s.close()
