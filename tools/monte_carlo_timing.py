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

    def magnetic_simulation(self, *a): return self._wrap("field tables", super().magnetic_simulation, *a)
    def condition_based_time(self, *a): return self._wrap("horizon", super().condition_based_time, *a)
    def solve(self, *a):
        r = self._wrap("solve", super().solve, *a)
        self.t["solve kernel"] = self.t.get("solve kernel", 0.0) + self.solver.last_kernel_ms / 1e3
        return r
    def attitude_simulation(self, *a): return self._wrap("tracking", super().attitude_simulation, *a)


st = Timed(solver)
mc.run_trials(st, 1, 0, 8)          # warm-up
st.t.clear()
t0 = time.time()
out = mc.monte_carlo(st, number_sims=n, seed=2019, chunk=chunk)
wall = time.time() - t0
sm = mc.summarize(out)
print(f"{n} trials in {wall:.2f} s ({n / wall:.1f} trials/s), chunk {chunk}; knots min/median/max "
      f"{out['n_knots'].min()}/{int(np.median(out['n_knots']))}/{out['n_knots'].max()}; "
      f"failed {len(sm['fails'])}, mean slew time {sm['slew_time_mean']:.1f} s")
for k, v in st.t.items():
    print(f"  {k:14s} {v:8.3f} s")
print(f"  {'host glue':14s} {wall - sum(v for k, v in st.t.items() if k != 'solve kernel'):8.3f} s")
solver.close()
