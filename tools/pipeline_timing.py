"""Developer tool: the reference's whole per-batch sequence on one GPU at the bench size —
field tables -> horizon selection -> AL-iLQR solve -> TVLQR tracking — for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
load_package()
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss, tracking as tr, horizon as hz, magnetic as mg

T, N = 1024, 1000
b = ss.workload_monte_carlo(T=T, N=N)
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 5
opts.opts_uncon.iterations = 10; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
rng = np.random.default_rng(1)
kep = np.tile([0.0, 6771.0, 96.6, 0.0, 0.0, 0.0], (T, 1)); kep[:, 3] = rng.random(T) * 360; kep[:, 5] = rng.random(T) * 360
for rep in range(2):
    tb = time.time(); coarse, _ = mg.magnetic_simulation(s, kep, 0.0, 2400.0, 5000, want_pos=False); coarse = coarse[:, :5000]
    print(f"rep {rep}: field tables {1e3*(time.time()-tb):.1f} ms (host-inclusive, {T} orbits x 10000 rows)", flush=True)
    coarse = np.ascontiguousarray(coarse)
    t0 = time.time(); idx, cond = hz.condition_based_time(s, coarse, 2400.0 / 5000, 30.0); t1 = time.time()
    res = to.solve_(to.BatchProblem.from_arrays(b), s, want_K=False); t2 = time.time()
    Qd, Qfd, Rd = tr.tvlqr_weights(T, r=0.5e3)
    x0s = tr.perturbed_initial_state(b.x0, rng)
    nz = tr.simulator_noise(T, N, rng)
    t3 = time.time(); tv = tr.attitude_simulation(s, b, res["X"], res["U"], x0s, Qd, Qfd, Rd, noise=nz); t4 = time.time()
    print(f"rep {rep}: horizon {1e3*(t1-t0):.1f} ms (host-inclusive), solve {1e3*(t2-t1):.1f} ms (incl. upload/download), "
          f"tracking {1e3*(t4-t3):.1f} ms (host-inclusive); solve kernel {s.last_kernel_ms:.1f} ms; "
          f"slews found {int(np.sum(tv['stats']['failed'] == 0))}/{T}, median slew time {np.median(tv['stats']['slew_time']):.1f} s", flush=True)
s.close()
