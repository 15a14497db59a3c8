"""CPU experiment behind the stated bar of the 3 x 50 budget (tests/test_gpu_full_configs.py::_assert_long_budget_parity).

The kernel and the oracle take the same decisions on every trajectory of configs[3] and still end up to 4e-8 apart on the slews
that run 130 - 150 iterations; the explanation given was that 150 iterations of this aggressive weighting amplify a rounding-level
difference (analytic tangents against dual numbers) by up to 1e8. Shown here instead of argued: the ORACLE AGAINST ITSELF under an
equally small perturbation of its own arithmetic — the shipped build (g++ -O2 -march=x86-64-v3: a*b + c contracted to FMAs where
the compiler likes) against the same source built with -ffp-contract=off (no FMA contraction: every product rounded) — on 64 of
the 130 - 150-iteration trajectories of the sweep. If the oracle moves by 1e-8 against itself, no kernel can be held to 1e-9 there.

    python tools/oracle_self_perturbation.py [n_pool] > profiles/r04/oracle_self_perturbation.txt      (CPU only, ~2 minutes)
"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np

N, J0 = 1000, 20000


def workload(ol, ss, n):
    b = ss.workload_inclination_sweep(T=n, N=N, j0=J0, tables=False)
    rows = N + 8
    B, _ = ol.btable_batch(b.meta["kep"], 0.0, N * 0.2, N, want_pos=False)          # IGRF-12 tables on the CPU (oracle: orc_btable_batch)
    b.Btab, b.n_tab = np.ascontiguousarray(B[:, :rows]), rows
    b.btab_idx = np.arange(n, dtype=np.int32); b.tau0[:] = 0.0; b.dtau[:] = 1.0
    return b


def child(out, n, pick):
    from tsat_loader import load_package
    pkg = load_package()
    import oracle_lib as ol
    b = workload(ol, pkg.slew_setup, n)
    if pick:
        idx = np.load(pick)
        sb = b.slice(0, 1)
        for f in ("x0", "xf", "tau0", "dtau", "dt", "Jmat", "Qd", "Qfd", "Rd", "ulo", "uhi", "U0"):
            setattr(sb, f, np.ascontiguousarray(getattr(b, f)[idx]))
        sb.Btab, sb.btab_idx = np.ascontiguousarray(b.Btab[idx]), np.arange(len(idx), dtype=np.int32)
        b = sb
    o = ol.default_options()
    o.max_outer, o.max_inner, o.dj_counter_limit, o.error_state = 3, 50, 1, 1
    r = ol.solve_batch(b, o, nthreads=ol.num_procs(), want_K=False)
    np.savez(out, X=r["X"], U=r["U"], stats=r["stats"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
        sys.exit(0)
    n_pool = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    tmp = tempfile.mkdtemp()
    flags = "-O2 -std=c++17 -fPIC -fopenmp -march=x86-64-v3 -shared"
    src = os.path.join(ROOT, "oracle", "tsat_oracle.cpp")
    nofma = os.path.join(tmp, "liboracle_nofma.so")
    subprocess.check_call(f"g++ {flags} -ffp-contract=off -o {nofma} {src}", shell=True)
    run = lambda lib, out, pick=None: subprocess.check_call([sys.executable, __file__, "--child", out, str(n_pool)] + ([pick] if pick else []),
                                                            env=dict(os.environ, **({"TSAT_ORACLE_LIB": lib} if lib else {})))
    a_all = os.path.join(tmp, "a_all.npz")
    run(None, a_all)
    za = np.load(a_all)
    it = za["stats"]["inner_iters"].astype(int) + za["stats"]["bp_restarts"]
    long_ = np.flatnonzero(it >= 130)[:64]
    pick = os.path.join(tmp, "pick.npy"); np.save(pick, long_)
    b_sel = os.path.join(tmp, "b_sel.npz")
    run(nofma, b_sel, pick)
    zb = np.load(b_sel)
    Xa, Ua, sa = za["X"][long_], za["U"][long_], za["stats"][long_]
    dX = np.max(np.abs(Xa - zb["X"]), axis=(1, 2))
    dU = np.max(np.abs(Ua - zb["U"]), axis=(1, 2)) / np.maximum(1.0, np.max(np.abs(Ua), axis=(1, 2)))
    keys = ("status", "outer_iters", "inner_iters", "ls_trials", "n_backward", "bp_restarts", "fp_fails")
    same = np.all([sa[k] == zb["stats"][k] for k in keys], axis=0)
    print(f"configs[3] inputs (inclination sweep from global index {J0}, IGRF-12 tables, R * 0.1, hooks, 3 x 50), pool of {n_pool}: iterations "
          f"min / median / max {it.min()} / {int(np.median(it))} / {it.max()}; {len(long_)} trajectories with >= 130 iterations taken")
    print("oracle (shipped build: g++ -O2 -march=x86-64-v3, FMA contraction on) against the same source built with -ffp-contract=off:")
    print(f"  identical iteration / line-search / restart counts and statuses: {int(same.sum())} of {len(long_)}")
    q = lambda a: " / ".join(f"{np.quantile(a, p):.1e}" for p in (0.5, 0.9, 1.0))
    print(f"  max|dX| per trajectory, median / q90 / max (all {len(long_)}):              {q(dX)}")
    print(f"  max|dU| / control scale, median / q90 / max (all {len(long_)}):             {q(dU)}")
    if same.any():
        print(f"  ... on the {int(same.sum())} with identical counts:  |dX| {q(dX[same])};  |dU|/scale {q(dU[same])}")
        print(f"  inside the 1e-9 bar of the 5 x 10 configs: {np.mean((dX[same] < 1e-9) & (dU[same] < 1e-9)):.3f} of them; "
              f"inside the long-budget bar (1e-6 / 1e-5): {np.mean((dX[same] < 1e-6) & (dU[same] < 1e-5)):.3f}")
    print("(the GPU kernel against the shipped oracle on the same kind of trajectories: counts identical on all, worst |dX| 4.2e-8, "
          "tests/test_gpu_full_configs.py / profiles/r04/parity_full_configs.txt)")
