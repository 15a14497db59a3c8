"""Developer tool: run the diagnostic build (libtortoise_hip_prof.so, -DTSAT_PROFILE) on the bench workload and
print where a wavefront spends its shader-clock cycles (forward sweep / Jacobian lanes / Riccati / parallel passes).
The stamped build is slower than the product; read SHARES, not absolute time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
pkg._abi.LIB_NAME = os.environ.get("TSAT_PROF_LIB", "libtortoise_hip_prof.so")
from tortoisesat_jl_amd import trajopt as to, slew_setup as ss

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # 0 auto, 1 wide, 2 dense, 3 packed (PK_G trajectories per wave)
es = int(sys.argv[4]) if len(sys.argv) > 4 else 0
prec = int(sys.argv[5]) if len(sys.argv) > 5 else 64      # 32: the mixed-precision builds
base = ss.workload_monte_carlo(T=min(T, 1024), N=N)
if T > 1024:       # larger batches re-use the 1024 draws (tiling) so that host-side setup stays cheap
    k = T // 1024
    rep = lambda a: np.ascontiguousarray(np.concatenate([a] * k))
    b = ss.SlewBatch(base.N, base.n_tab, rep(base.x0), rep(base.xf), base.Btab, rep(base.btab_idx), rep(base.tau0), rep(base.dtau),
                     rep(base.dt), rep(base.Jmat), rep(base.Qd), rep(base.Qfd), rep(base.Rd), rep(base.ulo), rep(base.uhi), rep(base.U0))
    T = b.T
else:
    b = base
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 5
opts.opts_uncon.iterations = 10; opts.opts_uncon.dJ_counter_limit = 1
solver = to.AugmentedLagrangianSolver(None, opts)
o = opts.to_abi(b.N, b.n_tab, 3, error_state=es)
o.precision = prec
solver.set_kernel_variant(variant)
solver.set_endgame(0)     # one kernel: the stamps of a wavefront cover its trajectories from start to end
solver.upload(b, o.max_linesearch); solver.trace(1)
ms = solver.run(o); ms = solver.run(o)
tr = solver.trace_download()[:, 0, :]
nfw = solver.download(want_K=False)["stats"]["n_forward"].astype(float)
if variant >= 3 or (variant == 0 and T >= 3072):   # the packed builds stamp one row per wavefront (its first trajectory): sums over
    tr = tr[::int(os.environ.get("TSAT_PK_G", "4"))]   # its PK_G trajectories
print(f"T = {T}, variant {variant}, error_state {es}, precision {prec}: {len(tr)} stamped wavefronts")
it = tr[:, 4]; nb = tr[:, 5]
tot = tr[:, :4].sum(1)
print(f"kernel {ms:.2f} ms (stamped build); mean inner its {it.mean():.1f}")
for i, name in enumerate(("forward sweep", "jacobian lanes", "riccati", "parallel passes")):
    print(f"  {name:16s}: {tr[:, i].mean()/1e6:8.2f} Mcycles/wave  ({100*tr[:, i].sum()/tot.sum():5.1f} %)  "
          f"per iteration {np.mean(tr[:, i]/np.maximum(it,1))/1e3:8.1f} kcycles; per knot-iteration {np.mean(tr[:, i]/np.maximum(it,1))/N:7.1f} cycles")
if variant >= 3 or (variant == 0 and T >= 3072):
    for i, name in ((6, "of the passes: copy of the accepted roll-out + gradient"), (7, "of the passes: end of an inner loop (duals, penalty, next outer)")):
        print(f"  {name:66s}: per knot-iteration {np.mean(tr[:, i]/np.maximum(it,1))/N:7.1f} cycles")
if variant >= 3 or (variant == 0 and T >= 3072):   # a sweep serves the whole wavefront: cycles per sweep and knot of the WAVE
    G = int(os.environ.get("TSAT_PK_G", "4"))
    sw = nfw.reshape(-1, G).max(1)
    print(f"  forward sweep, per executed sweep of a wavefront and knot: {np.mean(tr[:, 0] / np.maximum(sw, 1)) / N:7.1f} cycles ({sw.mean():.1f} sweeps per wavefront)")
print(f"  slowest wave: {tot.max()/1e6:.1f} Mcycles, {int(it[np.argmax(tot)])} iterations (the launch ends with it); mean wave {tot.mean()/1e6:.1f}")
print(f"  sum of stamped phases {tot.mean()/1e6:.1f} Mcycles/wave = {tot.mean()/ (ms*1e-3)/1e9:.2f} GHz-equivalent of the kernel time")
solver.close()
