"""Monte-Carlo sweep over GPUs: shard the independent slews, solve, all-gather the results.

The reference runs its Monte-Carlo as a serial loop whose iterations share nothing but the result lists they
append to (src/monte_carlo.jl:118-235; src/paper_images/heatmap.jl:114-243). Here each rank (one process per
GPU) owns a contiguous block of trajectories; there is no communication during the solve and exactly one
exchange at the end — an all-gather of the per-trajectory results (RCCL over xGMI on GPUs, gloo in CPU tests).
"""
import numpy as np

from . import _abi


def shard_range(T_total, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one trajectory."""
    base, rem = divmod(int(T_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather_rows(t, world, group=None):
    """all-gather along dim 0 of equally-shaped tensors (CUDA: all_gather_into_tensor; CPU/gloo: all_gather)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return t
    if t.is_cuda:
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
        return out
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.cat(parts, dim=0)


def gather_results(res, world, group=None, device=None):
    """All-gather a result dict of equal-size shards (X (T,N,7), U (T,N-1,3), stats) -> dict over world*T
    trajectories in rank order. NumPy in, NumPy out (tensors are staged on `device` if given)."""
    import torch

    out = {}
    for k in ("X", "U"):
        t = torch.from_numpy(np.ascontiguousarray(res[k]))
        if device is not None:
            t = t.to(device)
        out[k] = _all_gather_rows(t, world, group).cpu().numpy()
    sb = torch.from_numpy(np.ascontiguousarray(res["stats"]).view(np.uint8).reshape(len(res["stats"]), -1).copy())
    if device is not None:
        sb = sb.to(device)
    g = _all_gather_rows(sb, world, group).cpu().numpy()
    out["stats"] = np.ascontiguousarray(g).view(_abi.STATS_DTYPE).reshape(-1)
    return out


def monte_carlo_sweep(make_shard, solve_shard, T_total, rank, world, group=None, device=None):
    """Sharded sweep: ``make_shard(lo, hi)`` builds this rank's SlewBatch, ``solve_shard(batch)`` returns its
    result dict; results of all ranks come back concatenated in global trajectory order.
    Ragged totals are padded to equal shard sizes for the collective and trimmed afterwards."""
    lo, hi = shard_range(T_total, rank, world)
    per = -(-int(T_total) // int(world))
    res = solve_shard(make_shard(lo, hi)) if hi > lo else None
    n_loc = hi - lo
    if res is None:
        raise ValueError("more ranks than trajectories")
    if n_loc < per:  # pad with copies of the last row so every rank contributes `per` rows
        pad = per - n_loc
        res = dict(X=np.concatenate([res["X"], np.repeat(res["X"][-1:], pad, 0)]),
                   U=np.concatenate([res["U"], np.repeat(res["U"][-1:], pad, 0)]),
                   stats=np.concatenate([res["stats"], np.repeat(res["stats"][-1:], pad, 0)]))
    g = gather_results(res, world, group, device)
    keep = np.concatenate([np.arange(r * per, r * per + (shard_range(T_total, r, world)[1] - shard_range(T_total, r, world)[0]))
                           for r in range(world)])
    return dict(X=g["X"][keep], U=g["U"][keep], stats=g["stats"][keep])


class ResultGather:
    """Per-step exchange used by bench.py: unpack the resident batch straight into torch device tensors
    (tsat_batch_export_device) and all-gather them across ranks without touching the host."""

    def __init__(self, solver, T, N, world, device, mode="full", group=None, force_collective=False):
        import torch

        self.solver, self.world, self.mode, self.group = solver, world, mode, group
        self.collective = world > 1 or force_collective
        self.stats = torch.empty((T, _abi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=device)
        self.X = self.U = None
        if mode == "full":
            self.X = torch.empty((T, N, 7), dtype=torch.float64, device=device)
            self.U = torch.empty((T, N - 1, 3), dtype=torch.float64, device=device)
        self.gathered = {}
        if self.collective and mode != "none":
            self.gathered["stats"] = torch.empty((world * T, _abi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=device)
            if mode == "full":
                self.gathered["X"] = torch.empty((world * T, N, 7), dtype=torch.float64, device=device)
                self.gathered["U"] = torch.empty((world * T, N - 1, 3), dtype=torch.float64, device=device)

    def gather(self):
        if self.mode == "none":
            return None
        import torch
        import torch.distributed as dist

        if self.collective and self.stats.is_cuda:
            # the previous step's all-gather (asynchronous, it overlaps the solve that ran since) must have drained
            # before its source buffers are overwritten by this export
            torch.cuda.current_stream().synchronize()
        self.solver.export_device(self.X.data_ptr() if self.X is not None else None,
                                  self.U.data_ptr() if self.U is not None else None,
                                  None, self.stats.data_ptr())
        if not self.collective:
            return dict(X=self.X, U=self.U, stats=self.stats)
        for k, src in (("stats", self.stats), ("X", self.X), ("U", self.U)):
            if k in self.gathered:
                if src.is_cuda:
                    dist.all_gather_into_tensor(self.gathered[k], src, group=self.group)
                else:                    # gloo (CPU tests)
                    self.gathered[k].copy_(_all_gather_rows(src, self.world, self.group))
        return self.gathered


def _agree(flag, world, device, group=None):
    """True iff `flag` holds on EVERY rank (all-reduce MIN of a 0/1 flag): the ranks take the same branch afterwards"""
    if world == 1:
        return bool(flag)
    import torch
    import torch.distributed as dist

    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class NativeGather:
    """The same per-step exchange through the library's own collective (``tsat_sweep_allgather``: export kernel +
    ncclAllGather on the solver's stream) — what a Julia host without torch calls. torch.distributed is used here only to
    hand the 128-byte communicator id from rank 0 to the other ranks and to agree on the outcome of every step of the set-up;
    the gathered arrays land in torch device tensors.

    The set-up is COLLECTIVE and so is its failure, as far as that can be had: every step that can fail on one rank alone BEFORE
    the communicator is initialised (RCCL not loadable, the id cannot be made, the shards have different shapes) is followed by
    an agreement (all-reduce MIN of a success flag), and either every rank goes on or every rank raises
    ``NativeGatherUnavailable`` having left no collective half-entered — a caller's fallback (``make_gatherer``) then takes the
    same branch everywhere; a rank deciding on its own would leave the others inside ``dist.broadcast``. Step 4,
    ``ncclCommInitRank``, is itself the collective: when it RETURNS an error on some rank the same agreement follows, but a rank
    that dies or raises before entering it leaves its peers blocked inside the call until the NCCL timeout — that failure is
    not recoverable here (the process group ends with a non-zero exit), and nobody falls back to torch after a partial init."""

    def __init__(self, solver, T, N, world, rank, device, mode="full", group=None):
        import torch
        import torch.distributed as dist

        self.solver, self.world, self.mode = solver, world, mode
        # 1. can every rank load RCCL behind the C ABI? (local probe, then agreement: nobody has entered a collective of the library yet)
        try:
            ok = bool(solver.comm_available())
        except Exception:
            ok = False
        if not _agree(ok, world, device, group):
            raise NativeGatherUnavailable("RCCL is not loadable behind the C ABI on at least one rank")
        # 2. equal shard shapes on every rank (the receive buffers are world x the local shard)
        if world > 1:
            shp = torch.tensor([T, N, -T, -N], dtype=torch.int64, device=device)
            dist.all_reduce(shp, op=dist.ReduceOp.MAX, group=group)
            if int(shp[0]) != -int(shp[2]) or int(shp[1]) != -int(shp[3]):
                raise ValueError(f"NativeGather: the ranks hold shards of different shapes (this rank {T} x {N}; largest "
                                 f"{int(shp[0])} x {int(shp[1])}, smallest {-int(shp[2])} x {-int(shp[3])}): pad the shards to equal size")
        # 3. rank 0 makes the id and broadcasts it together with a status byte (zeros = it could not)
        buf = torch.zeros(1 + _abi.TSAT_COMM_ID_BYTES, dtype=torch.uint8, device=device)
        if rank == 0:
            try:
                cid = bytes(solver.comm_unique_id())
                buf[1:] = torch.frombuffer(bytearray(cid), dtype=torch.uint8).to(device)
                buf[0] = 1
            except Exception:
                pass
        if world > 1:
            dist.broadcast(buf, src=0, group=group)
        host = buf.cpu().numpy()
        if not host[0]:
            raise NativeGatherUnavailable("rank 0 could not create the communicator id")
        self.comm_id = host[1:].tobytes()
        # 4. the collective comm_init, then agreement on its outcome
        try:
            solver.comm_init(self.comm_id, rank, world)
            ok, why = True, ""
        except Exception as e:
            ok, why = False, str(e)
        if not _agree(ok, world, device, group):
            try:
                solver.comm_destroy()
            except Exception:
                pass
            raise NativeGatherUnavailable(f"tsat_comm_init failed on at least one rank{': ' + why if why else ''}")
        self.gathered = {}
        if mode != "none":
            self.gathered["stats"] = torch.empty((world * T, _abi.STATS_DTYPE.itemsize), dtype=torch.uint8, device=device)
            if mode == "full":
                self.gathered["X"] = torch.empty((world * T, N, 7), dtype=torch.float64, device=device)
                self.gathered["U"] = torch.empty((world * T, N - 1, 3), dtype=torch.float64, device=device)

    def gather(self):
        if self.mode == "none":
            return None
        g = self.gathered
        self.solver.sweep_allgather(g["X"].data_ptr() if "X" in g else None, g["U"].data_ptr() if "U" in g else None,
                                    g["stats"].data_ptr(), on_device=True)
        return g


class NativeGatherUnavailable(RuntimeError):
    """raised by NativeGather on EVERY rank alike when the library's own collective cannot be set up"""


def make_gatherer(solver, T, N, world, rank, device, mode="full", impl="native", group=None, force_collective=False):
    """(gatherer, impl actually used). ``impl = "native"``: the library's RCCL all-gather, falling back — on every rank together —
    to torch.distributed's communicator when it cannot be set up; ``"torch"``: torch.distributed straight away."""
    collective = world > 1 or force_collective
    if collective and impl == "native":
        try:
            return NativeGather(solver, T, N, world, rank, device, mode=mode, group=group), "native"
        except NativeGatherUnavailable as e:
            import sys
            print(f"[sweep] native gather unavailable on every rank ({e}); using torch.distributed", file=sys.stderr)
    g = ResultGather(solver, T, N, world, device, mode=mode, group=group, force_collective=collective)
    return g, ("torch" if collective else "export-only")
