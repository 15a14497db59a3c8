"""Worker of tests/test_dist_gloo.py: one rank of a world_size-2 `gloo` sweep on the CPU.

The sharding / padding / all-gather logic is the product's (tortoisesat.jl_amd/sweep.py); only the per-shard
solve is swapped for the CPU oracle, because there is no GPU in this tier."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), HERE]

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import oracle_lib as ol  # noqa: E402
from tsat_loader import load_package  # noqa: E402


def main():
    out, T_total = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_package()
    ss, sweep = pkg.slew_setup, pkg.sweep
    if len(sys.argv) > 3 and sys.argv[3] == "mc":   # the whole experiment, summaries all-gathered in trial order
        from mc_stages import OracleStages
        mc = pkg.monte_carlo
        r = mc.monte_carlo(OracleStages(ol, nthreads=2), number_sims=T_total, seed=8, rank=rank, world=world, chunk=2,
                           setup=mc.MonteCarloSetup(N=600, dt=1.0, outer=2, inner=4))
        np.savez(f"{out}.rank{rank}.npz", **{k: r[k] for k in ("A", "t_final", "slew_time", "fails", "n_knots")})
        dist.barrier()
        dist.destroy_process_group()
        return
    full = ss.workload_monte_carlo(T=T_total, N=40, seed=99, random_orbit=True)
    o = ol.default_options()
    o.max_outer, o.max_inner = 2, 3
    res = sweep.monte_carlo_sweep(lambda lo, hi: full.slice(lo, hi), lambda b: ol.solve_batch(b, o, want_K=False),
                                  T_total, rank, world)
    np.savez(f"{out}.rank{rank}.npz", X=res["X"], U=res["U"], stats=res["stats"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
