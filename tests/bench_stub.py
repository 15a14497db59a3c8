"""Stub of AugmentedLagrangianSolver for the rehearsal of bench.py's multi-rank plumbing on the CPU (tests/test_dist_gloo.py,
bench.py: TSAT_BENCH_REHEARSAL). Nothing is solved: a "result" carries the GLOBAL index of its trajectory — recovered from the
inclination the workload builder gave it — so that the test can read shard ranges, padding and gather order off the gathered arrays."""
import ctypes as C
import os

import numpy as np


def view(ptr, shape, dtype):
    n = int(np.prod(shape))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class StubSolver:
    def __init__(self, opts):
        from tortoisesat_jl_amd import _abi
        self.abi, self.opts = _abi, opts
        self.batch = None

    def upload(self, batch, max_linesearch):
        self.batch = batch
        T, N = batch.T, batch.N
        T_total = int(os.environ.get("TSAT_BENCH_REHEARSAL_T", "65536"))
        kep = batch.meta.get("kep")
        j = np.rint(kep[:, 2] * T_total / 90.0 - 0.5).astype(np.int64) if kep is not None and kep.shape[0] == T else np.arange(T)
        self.X = np.zeros((T, N, 7)); self.U = np.zeros((T, N - 1, 3))
        self.X[:, :, 0] = j[:, None]
        self.U[:, :, 0] = batch.U0[:, :, 0]
        self.stats = np.zeros(T, dtype=self.abi.STATS_DTYPE)
        self.stats["inner_iters"] = j
        self.stats["n_backward"] = 1; self.stats["n_forward"] = 1; self.stats["outer_iters"] = 1

    def run(self, abi):
        return 1.0

    def selected_build(self, abi):
        return 3, 0

    def download(self, want_K=False):
        return dict(X=self.X, U=self.U, stats=self.stats)

    def export_device(self, X_ptr=None, U_ptr=None, K_ptr=None, stats_ptr=None):
        if X_ptr:
            view(X_ptr, self.X.shape, np.float64)[...] = self.X
        if U_ptr:
            view(U_ptr, self.U.shape, np.float64)[...] = self.U
        if stats_ptr:
            view(stats_ptr, (len(self.stats), self.abi.STATS_DTYPE.itemsize), np.uint8)[...] = self.stats.view(np.uint8).reshape(len(self.stats), -1)

    def close(self):
        pass
