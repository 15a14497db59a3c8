"""CPU tier: the set-up of the library's own sweep exchange (sweep.NativeGather / make_gatherer) with world_size 2 over gloo
and the collective stubbed — the communicator-id hand-off from rank 0, the buffer arithmetic of the gathered arrays, and the
all-ranks-together decision when the native path cannot be set up (a rank that decides alone leaves the
others inside a collective)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, fail):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / f"ng_{fail}")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "native_gather_worker.py"), out, fail], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0          # a mismatched collective would hang here, not fail
    return [np.load(f"{out}.rank{r}.npz") for r in range(2)]


def _expected(abi, T=5, N=9):
    X = np.concatenate([np.random.default_rng(100 + r).standard_normal((T, N, 7)) for r in range(2)])
    it = np.concatenate([1000 * r + np.arange(T) for r in range(2)])
    return X, it


def test_native_path_hands_over_the_id_and_gathers_in_rank_order(pkg, tmp_path):
    z = _run(tmp_path, "ok")
    X, it = _expected(pkg._abi)
    want_id = np.array([(7 * i + 3) % 256 for i in range(pkg._abi.TSAT_COMM_ID_BYTES)], dtype=np.uint8)
    for r in range(2):
        assert str(z[r]["impl"]) == "native"
        assert np.array_equal(z[r]["comm"], want_id)                    # every rank initialised with rank 0's id
        assert z[r]["X"].shape == (10, 9, 7) and z[r]["U"].shape == (10, 8, 3)
        assert np.array_equal(z[r]["X"], X) and np.array_equal(z[r]["stats"]["inner_iters"], it)
    assert "unique_id" in str(z[0]["calls"]) and "unique_id" not in str(z[1]["calls"])


@pytest.mark.parametrize("fail", ["load", "id", "init"])
def test_every_rank_falls_back_together(pkg, tmp_path, fail):
    """RCCL not loadable on rank 1 / rank 0 cannot make the id / comm_init fails on rank 1: BOTH ranks end up on torch.distributed's
    communicator with the right data, nobody is left inside a collective"""
    z = _run(tmp_path, fail)
    X, it = _expected(pkg._abi)
    for r in range(2):
        assert str(z[r]["impl"]) == "torch", (fail, r, str(z[r]["impl"]))
        assert np.array_equal(z[r]["X"], X) and np.array_equal(z[r]["stats"]["inner_iters"], it)
    calls = [str(z[r]["calls"]).split(",") for r in range(2)]
    if fail == "load":          # agreed before anybody touched the library's collective set-up
        assert all("init" not in c and "unique_id" not in c for c in calls)
    if fail == "id":
        assert all("init" not in c for c in calls)
    if fail == "init":          # the rank whose comm_init succeeded gave its communicator back
        assert "destroy" in calls[0]


def test_unequal_shards_are_refused_on_every_rank(pkg, tmp_path):
    z = _run(tmp_path, "shape")
    for r in range(2):
        assert "different shapes" in str(z[r]["error"]) and str(z[r]["impl"]) == ""
