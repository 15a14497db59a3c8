"""Developer tool: timing of larger resident batches (configs[2]/[3] shapes, fp64) on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
if os.environ.get("TSAT_LIB"):            # developer switch: another build of the same library (experiments)
    pkg._abi.LIB_NAME = os.environ["TSAT_LIB"]
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss

VARIANTS = {0: "auto", 1: "wide", 2: "dense", 3: "packed", 4: "packed8", 5: "packed8w", 6: "packed16w", 7: "packed4w"}


def run(name, b, outer, inner, es=0, variants=(0,)):
    opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = outer
    opts.opts_uncon.iterations = inner; opts.opts_uncon.dJ_counter_limit = 1
    s = to.AugmentedLagrangianSolver(None, opts)
    o = opts.to_abi(b.N, b.n_tab, 3, error_state=es)
    t = time.time(); s.upload(b, o.max_linesearch); up = time.time() - t
    ref = None
    for v in variants:
        s.set_kernel_variant(v)
        ms = s.run(o); ms = s.run(o)
        res = s.download(want_K=False)
        st = res["stats"]
        same = "" if ref is None else f"; bit-identical to {VARIANTS[variants[0]]}: {np.array_equal(ref['X'], res['X']) and np.array_equal(ref['U'], res['U'])}"
        ref = ref or res
        print(f"{name} [{VARIANTS[v]}]: T={b.T} N={b.N} {outer}x{inner} es={es}: kernel {ms:.1f} ms -> {b.T/(ms*1e-3):.0f} solves/s; upload {up:.2f} s; "
              f"HBM {s.reserved_bytes()/2**30:.2f} GiB; status {np.bincount(st['status'], minlength=4)}; mean inner {st['inner_iters'].mean():.1f}{same}", flush=True)
    s.close()

if __name__ == "__main__":
    base = ss.workload_monte_carlo(T=1024, N=1000)
    run("configs[1]", base, 5, 10)
    run("configs[1] + quaternion hooks", base, 5, 10, es=1)
    big_variants = tuple(int(v) for v in os.environ.get("TSAT_VARIANTS", "2,3").split(","))
    # larger batches re-use the 1024 draws (tiling) so that host-side setup stays cheap; tables are per trajectory
    rep = lambda a, k: np.ascontiguousarray(np.concatenate([a] * k))
    for k, nm in ((8, "configs[3] shard (8192 / GPU), budget 3x50"), (16, "configs[2] shape (16384), fp64")):
        b = ss.workload_monte_carlo(T=1024, N=1000, seed=20190531, random_orbit=True)
        big = ss.SlewBatch(b.N, b.n_tab, rep(b.x0, k), rep(b.xf, k), b.Btab, rep(b.btab_idx, k), rep(b.tau0, k), rep(b.dtau, k),
                           rep(b.dt, k), rep(b.Jmat, k), rep(b.Qd, k), rep(b.Qfd, k), rep(b.Rd, k), rep(b.ulo, k), rep(b.uhi, k), rep(b.U0, k))
        run(nm, big, 3 if k == 8 else 5, 50 if k == 8 else 10, es=1, variants=big_variants)
