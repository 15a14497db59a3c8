"""CPU tier: the N > 1 path — shard, solve, all-gather — with world_size 2 over `gloo`."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("T_total", [6, 5])   # even and ragged split
def test_two_rank_sweep_equals_single_process(pkg, ol, tmp_path, T_total):
    port = free_port()
    out = str(tmp_path / "sweep")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), out, str(T_total)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    full = pkg.slew_setup.workload_monte_carlo(T=T_total, N=40, seed=99, random_orbit=True)
    o = ol.default_options()
    o.max_outer, o.max_inner = 2, 3
    ref = ol.solve_batch(full, o, want_K=False)
    for r in range(2):
        z = np.load(f"{out}.rank{r}.npz")
        assert z["X"].shape == ref["X"].shape
        # independent trajectories: the sharded sweep is the concatenation of the shards, bit for bit
        assert np.array_equal(z["X"], ref["X"]) and np.array_equal(z["U"], ref["U"])
        assert np.array_equal(z["stats"], ref["stats"])


def test_two_rank_monte_carlo_equals_single_process(pkg, ol, tmp_path):
    """the full experiment (tables -> horizon -> solve -> tracking -> statistic) sharded over two ranks"""
    from mc_stages import OracleStages
    port = free_port()
    out = str(tmp_path / "mc")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), out, "5", "mc"], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    mc = pkg.monte_carlo
    ref = mc.monte_carlo(OracleStages(ol), number_sims=5, seed=8, setup=mc.MonteCarloSetup(N=600, dt=1.0, outer=2, inner=4))
    for r in range(2):
        z = np.load(f"{out}.rank{r}.npz")
        for k in ("A", "t_final", "slew_time", "fails", "n_knots"):
            assert np.array_equal(z[k], ref[k]), k


@pytest.mark.parametrize("T_sweep", [13, 16])      # a sweep the two ranks do not divide (one trajectory is solved twice), and one they do
def test_bench_config3_two_ranks_rehearsal(pkg, tmp_path, T_sweep):
    """`bench.py --gpus 2 --config 3` itself — its shard ranges, the padding of shards to equal length, T_global, the gather and
    the order of what it gathers, the barrier / max-over-ranks clock — on the CPU over gloo with the solver stubbed
    (TSAT_BENCH_REHEARSAL, tests/bench_stub.py): the first real multi-GPU run then has RCCL and the kernels left to prove, not the
    bookkeeping. The stub's "results" carry the global index of their trajectory."""
    import json
    root = os.path.dirname(HERE)
    port = free_port()
    out = str(tmp_path / "bench")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TSAT_BENCH_REHEARSAL=os.path.join(HERE, "bench_stub.py"), TSAT_BENCH_REHEARSAL_T=str(T_sweep), TSAT_BENCH_REHEARSAL_N="12",
                   TSAT_BENCH_REHEARSAL_OUT=out, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "3", "--steps", "2", "--warmup", "1",
                                       "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    line = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][-1])
    assert not any(l.startswith("{") for l in outs[1].splitlines())     # ONE line, from rank 0 (gloo prints its own banner)
    assert line["rehearsal"] is True and line["value"] is None     # not a measurement
    per = -(-T_sweep // 2)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["traj_total"] == T_sweep and line["config"]["traj_per_gpu"] == per
    assert line["config"]["gather_impl"] == "torch" and line["steps"] == 2
    # every rank holds the same gathered arrays: rank 0's shard [0, per), then rank 1's — shifted back so that it ends with the sweep
    expect = np.concatenate([np.arange(0, per), np.arange(T_sweep - per, T_sweep)])
    whole = pkg.slew_setup.workload_inclination_sweep(T=T_sweep, N=12, j0=0, T_total=T_sweep, tables=False)
    for r in range(2):
        z = np.load(f"{out}.rank{r}.npz")
        st = np.ascontiguousarray(z["stats"]).view(pkg._abi.STATS_DTYPE).reshape(-1)
        assert int(z["T_shard"]) == per and z["X"].shape == (2 * per, 12, 7)
        assert np.array_equal(st["inner_iters"], expect) and np.array_equal(z["X"][:, 0, 0], expect.astype(float))
        # ... and the shards were built from the global index: the gathered initial controls are rows of the one-piece sweep
        assert np.array_equal(z["U"][:, :, 0], whole.U0[expect][:, :, 0])
