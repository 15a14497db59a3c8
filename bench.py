#!/usr/bin/env python3
"""bench.py — AL-iLQR trajectory solves/sec (1000 knots) on N MI355X (BASELINE.json metric).

A "step" = one full AL-iLQR solve of the per-GPU batch: BASELINE.json configs[1] — 1024-trajectory Monte-Carlo,
1000 knots, random q0 in one orbit, fp64, budget 5 x 10 (src/monte_carlo.jl:107-198) — with inputs already
resident in HBM, followed (N > 1) by the RCCL all-gather of every rank's trajectories and stats.
Weak scaling: every GPU solves its own 1024 trajectories (seed offset by rank); value = all trajectories solved
by all ranks / max-over-ranks wall time.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); 6290 measured copy
T_PER_GPU = 1024
N_KNOTS = 1000
SEED = 20190530


def algorithmic_bytes(stats, N, w=8):
    """SURVEY.md §8(d) per-knot figures x executed counts: backward sweep 49 scalars/knot,
    forward sweep 47 scalars/knot (ONE trial per sweep: all backtracking candidates share the reads and only the
    accepted one has to be written), AL outer update 34 scalars/knot."""
    nb = stats["n_backward"].astype(np.float64)
    nf = stats["n_forward"].astype(np.float64)
    no = np.maximum(stats["outer_iters"].astype(np.float64) - 1.0, 0.0)
    return float(np.sum(N * w * (49.0 * nb + 47.0 * nf + 34.0 * no)))


def cpu_baseline(batch, abi_opts, res, sample, threads):
    """Time the CPU oracle (oracle/liboracle.so, the C++ port of the reference algorithm) on the first `sample`
    trajectories of the same workload, `threads` OpenMP threads; also reports parity of the GPU result on them."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as ol

    sub = batch.slice(0, sample)
    oo = ol.default_options()
    for f in ("integrator", "max_outer", "max_inner", "max_linesearch", "dj_counter_limit", "cost_tol", "grad_tol",
              "constraint_tol", "penalty_init", "penalty_scale", "penalty_max", "dual_max", "reg_init", "reg_scale",
              "reg_min", "reg_max", "reg_fp", "ls_lower", "ls_upper", "max_state", "u_scale", "terminal_mask", "error_state"):
        setattr(oo, f, getattr(abi_opts, f))
    t0 = time.perf_counter()
    ref = ol.solve_batch(sub, oo, nthreads=threads, want_K=False)
    dt = time.perf_counter() - t0
    dX = float(np.max(np.abs(ref["X"] - res["X"][:sample])))
    dU = float(np.max(np.abs(ref["U"] - res["U"][:sample])))
    same = bool(np.all(ref["stats"]["inner_iters"] == res["stats"]["inner_iters"][:sample])
                and np.all(ref["stats"]["ls_trials"] == res["stats"]["ls_trials"][:sample]))
    return dict(value=sample / dt, unit="solves/s", cores=threads, kind="port",
                sample=f"first {sample} trajectories of the same workload, {dt:.2f} s wall on {threads} OpenMP threads "
                       f"(oracle/liboracle.so, fp64)",
                parity=dict(max_abs_dX=dX, max_abs_dU=dU, iteration_counts_equal=same, tol=1e-9))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--traj-per-gpu", type=int, default=T_PER_GPU)
    ap.add_argument("--knots", type=int, default=N_KNOTS)
    ap.add_argument("--gather", choices=["full", "stats", "none"], default="full")
    ap.add_argument("--cpu-sample", type=int, default=96)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the solve path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # TSAT_BENCH_FORCE_DIST=1 rehearses the RCCL path (init, all-gather, barrier, all-reduce) with a single rank
    use_dist = world > 1 or os.environ.get("TSAT_BENCH_FORCE_DIST") == "1"
    # RCCL prints a version banner on stdout when the communicator comes up; the contract is ONE JSON line on stdout,
    # so stdout is pointed at stderr until the result line is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from tsat_loader import load_package
    load_package()
    from tortoisesat_jl_amd import slew_setup as ss, trajopt as to, sweep

    T, N = args.traj_per_gpu, args.knots
    batch = ss.workload_monte_carlo(T=T, N=N, seed=SEED + rank)  # configs[1]; every rank its own draw
    opts = to.AugmentedLagrangianSolverOptions()
    opts.iterations = batch.meta["max_outer"]
    opts.opts_uncon.iterations = batch.meta["max_inner"]
    opts.opts_uncon.dJ_counter_limit = batch.meta["dj_counter_limit"]
    solver = to.AugmentedLagrangianSolver(None, opts, device=local_rank)
    abi = opts.to_abi(N, batch.n_tab, 3)
    solver.upload(batch, abi.max_linesearch)          # inputs resident in HBM before the timed region
    gat = sweep.ResultGather(solver, T, N, world, dev, mode=args.gather, force_collective=use_dist)

    def step():
        ms = solver.run(abi)                          # blocks until the solve kernel has finished
        gat.gather()                                  # export + RCCL all-gather (no-op collective at world 1)
        return ms

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = []
    for _ in range(args.steps):
        kms.append(step())
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    res = solver.download(want_K=False)
    st = res["stats"]
    if rank == 0:
        kernel_ms = float(np.mean(kms))
        bytes_launch = algorithmic_bytes(st, N)
        achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
        traffic, issue = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    pm = json.load(f)
                traffic = pm.get("hbm_bytes_per_launch")
                # what actually bounds this kernel: VALU instructions issued per launch (SQ_INSTS_VALU, a
                # separate --pmc pass) against what 1024 SIMDs can issue in fp64 — one wave64 instruction per 4 cycles
                # at 2.4 GHz — during the measured launch
                n_valu = pm.get("sq", {}).get("SQ_INSTS_VALU")
                if n_valu:
                    cap = 1024 * 2.4e9 / 4.0 * kernel_ms * 1e-3
                    issue = {"valu_insts_per_launch": n_valu, "fp64_issue_capacity": cap, "frac": n_valu / cap,
                             "note": "one wavefront per SIMD can use about 45 % of it (profiles/r01/lane_mask_ubench.txt)"}
            except Exception:
                traffic, issue = None, None
        out = {
            "metric": "AL-iLQR trajectory solves/sec (1000 knots)",
            "value": world * T * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{T}-trajectory Monte-Carlo per GPU, {N} knots, random q0 in one orbit, fp64, "
                                   f"AL-iLQR budget {abi.max_outer}x{abi.max_inner} (BASELINE.json configs[1])",
                       "traj_per_gpu": T, "knots": N, "integrator": "rk3", "gather": args.gather,
                       "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "tsat_solve_kernel<double,3>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": bytes_launch, "fp64_issue": issue},
            "solve_stats": {"converged": int(np.sum(st["status"] == 0)), "max_outer": int(np.sum(st["status"] == 1)),
                            "reg_fail": int(np.sum(st["status"] == 2)), "diverged": int(np.sum(st["status"] == 3)),
                            "mean_inner_iters": float(st["inner_iters"].mean()),
                            "mean_ls_trials": float(st["ls_trials"].mean())},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = min(args.cpu_sample, T)
            threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            out["cpu_baseline"] = cpu_baseline(batch, abi, res, sample, threads)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    solver.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
