"""Developer tool: the reference's whole Monte-Carlo (src/monte_carlo.jl:107-262) at its own clocks on one GPU, with
wall time per stage (GPU calls are host-inclusive: upload, kernel, download)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
load_package()
from tortoisesat_jl_amd import trajopt as to, monte_carlo as mc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
solver = to.AugmentedLagrangianSolver(None, None)


class Timed(mc.GpuStages):
    def __init__(self, s):
        super().__init__(s)
        self.t = {}

    def _wrap(self, name, f, *a):
        t0 = time.time(); r = f(*a); self.t[name] = self.t.get(name, 0.0) + time.time() - t0
        return r

    def magnetic_simulation(self, *a, **kw): return self._wrap("field tables", lambda *x: mc.GpuStages.magnetic_simulation(self, *x, **kw), *a)
    def condition_based_time(self, *a): return self._wrap("horizon", super().condition_based_time, *a)
    def solve(self, *a, **kw):
        r = self._wrap("solve", lambda *x: mc.GpuStages.solve(self, *x, **kw), *a)
        self.t["solve kernel"] = self.t.get("solve kernel", 0.0) + self.solver.last_kernel_ms / 1e3
        return r
    def attitude_simulation(self, *a, **kw): return self._wrap("tracking", lambda *x: mc.GpuStages.attitude_simulation(self, *x, **kw), *a)


st = Timed(solver)
mc.run_trials(st, 1, 0, 8)          # warm-up
for keep, label in ((True, "every trajectory and table kept on the host, as the script does"), (False, "summaries only (keep_trajectories=False): trajectories and tables stay on the device")):
    st.t.clear()
    t0 = time.time()
    parts = [mc.run_trials(st, 2019, a, min(a + chunk, n), keep_trajectories=keep) for a in range(0, n, chunk)]
    wall = time.time() - t0
    out = {k: np.concatenate([p[k] for p in parts]) for k in ("slew_time", "fails", "n_knots")}
    sm = mc.summarize(out)
    print(f"{label}:\n  {n} trials in {wall:.2f} s ({n / wall:.1f} trials/s), chunk {chunk}; knots min/median/max "
          f"{out['n_knots'].min()}/{int(np.median(out['n_knots']))}/{out['n_knots'].max()}; "
          f"failed {len(sm['fails'])}, mean slew time {sm['slew_time_mean']:.1f} s")
    for k, v in st.t.items():
        print(f"  {k:14s} {v:8.3f} s")
    print(f"  {'host glue':14s} {wall - sum(v for k, v in st.t.items() if k != 'solve kernel'):8.3f} s")
solver.close()
