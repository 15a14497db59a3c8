"""Worker of tests/test_native_gather.py: one rank of a world_size-2 `gloo` run of the PRODUCT's gatherer set-up
(tortoisesat.jl_amd/sweep.py: NativeGather / make_gatherer) around a stub of the solver whose collective is emulated with
gloo — so the id hand-off, the buffer arithmetic and the all-ranks-together fallback are exercised without a GPU, and the first
real multi-GPU run only has RCCL itself left to prove."""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tsat_loader import load_package  # noqa: E402


def view(ptr, shape, dtype):
    n = int(np.prod(shape))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class StubSolver:
    """what sweep.py needs of AugmentedLagrangianSolver; `fail` seeds a failure on ONE rank (or in the id creation on rank 0)"""

    def __init__(self, abi, rank, world, T, N, fail):
        self.abi, self.rank, self.world, self.T, self.N, self.fail = abi, rank, world, T, N, fail
        self.comm = None
        self.calls = []
        rng = np.random.default_rng(100 + rank)
        self.X = rng.standard_normal((T, N, 7)); self.U = rng.standard_normal((T, N - 1, 3))
        self.stats = np.zeros(T, dtype=abi.STATS_DTYPE)
        self.stats["inner_iters"] = 1000 * rank + np.arange(T)

    def comm_available(self):
        self.calls.append("available")
        return not (self.fail == "load" and self.rank == 1)

    def comm_unique_id(self):
        self.calls.append("unique_id")
        if self.fail == "id":
            raise RuntimeError("tsat_comm_unique_id failed (-11)")
        return bytes((7 * i + 3) % 256 for i in range(self.abi.TSAT_COMM_ID_BYTES))

    def comm_init(self, cid, rank, world):
        self.calls.append("init")
        assert (rank, world) == (self.rank, self.world) and len(cid) == self.abi.TSAT_COMM_ID_BYTES
        if self.fail == "init" and self.rank == 1:
            raise RuntimeError("tsat_comm_init failed (-11): ncclCommInitRank")
        self.comm = bytes(cid)

    def comm_destroy(self):
        self.calls.append("destroy")
        self.comm = None

    def export_device(self, X_ptr=None, U_ptr=None, K_ptr=None, stats_ptr=None):
        if X_ptr:
            view(X_ptr, self.X.shape, np.float64)[...] = self.X
        if U_ptr:
            view(U_ptr, self.U.shape, np.float64)[...] = self.U
        if stats_ptr:
            view(stats_ptr, (self.T, self.abi.STATS_DTYPE.itemsize), np.uint8)[...] = self.stats.view(np.uint8).reshape(self.T, -1)

    def sweep_allgather(self, X_all=None, U_all=None, stats_all=None, on_device=False):
        assert self.comm is not None and on_device
        W = self.world
        for ptr, loc in ((X_all, self.X), (U_all, self.U), (stats_all, self.stats.view(np.uint8).reshape(self.T, -1))):
            if not ptr:
                continue
            parts = [torch.empty(loc.shape, dtype=torch.from_numpy(loc).dtype) for _ in range(W)]
            dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(loc)))
            view(ptr, (W * loc.shape[0],) + loc.shape[1:], loc.dtype)[...] = torch.cat(parts).numpy()


def main():
    out, fail = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_package()
    T = 5 if not (fail == "shape" and rank == 1) else 4
    N = 9
    s = StubSolver(pkg._abi, rank, world, T, N, fail)
    dev = torch.device("cpu")
    res = dict(impl="", error="", calls="")
    try:
        gat, impl = pkg.sweep.make_gatherer(s, T, N, world, rank, dev, mode="full", impl="native")
        g = gat.gather()
        res.update(impl=impl, X=g["X"].numpy(), U=g["U"].numpy(), stats=np.ascontiguousarray(g["stats"].numpy()).view(pkg._abi.STATS_DTYPE).reshape(-1),
                   comm=np.frombuffer(s.comm or b"", dtype=np.uint8))
    except ValueError as e:
        res["error"] = str(e)
    res["calls"] = ",".join(s.calls)
    np.savez(f"{out}.rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
