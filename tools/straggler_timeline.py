"""Developer tool (GPU, diagnostic build libtortoise_hip_prof.so): WHEN each iteration of each trajectory of a configs[3] shard
ends (the -DTSAT_PROFILE build stamps the 100 MHz wall clock into column 7 of the iteration trace) — live trajectories against
time, and how fast the trajectories that end the launch advance while the machine empties around them.

    make -C tortoisesat.jl_amd/csrc libtortoise_hip_prof.so && python tools/straggler_timeline.py [T=8192] [endgame thresholds ...]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tsat_loader import load_package
pkg = load_package()
pkg._abi.LIB_NAME = os.environ.get("TSAT_PROF_LIB", "libtortoise_hip_prof.so")
from tortoisesat_jl_amd import magnetic as mg, slew_setup as ss, trajopt as to

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ats = [int(x) for x in sys.argv[2:]] or [0]
opts = to.AugmentedLagrangianSolverOptions(); opts.iterations = 3
opts.opts_uncon.iterations = 50; opts.opts_uncon.dJ_counter_limit = 1
s = to.AugmentedLagrangianSolver(None, opts)
b = mg.attach_igrf_tables(s, ss.workload_inclination_sweep(T=T, N=1000, j0=3 * 8192, tables=False))
o = opts.to_abi(b.N, b.n_tab, 3, error_state=1)
s.upload(b, o.max_linesearch)
s.trace(152)
s.set_kernel_variant(3)
for at in ats:
    s.set_endgame(at)
    ms = s.run(o); ms = s.run(o)
    st = s.download(want_K=False)["stats"]
    tr = s.trace_download()                       # (T, rows, 8)
    n = st["inner_iters"].astype(int)
    clk = tr[:, :, 7] / 100.0 / 1e3               # ms; row 0 of a wavefront's first trajectory holds the phase clocks instead (diagnostic build)
    t0 = min(clk[t, 1] for t in range(T) if n[t] > 1)     # about two iterations after the launch began
    end = np.array([clk[t, n[t] - 1] - t0 if n[t] > 1 else 0.0 for t in range(T)])
    clk[:, 0] = t0
    print(f"== endgame at {at}: launch {ms:.1f} ms (stamped build); last iteration ends {end.max():.1f} ms after the first one")
    assert 0 < end.max() < 1e5, "no wall-clock stamps in the trace: is this the diagnostic build?"
    grid = np.arange(0, end.max() + 50, 50.0)
    print("   live trajectories at t [ms]:", {int(g): int((end > g).sum()) for g in grid})
    # the trajectories that end the launch: their iteration index against time
    last = np.argsort(end)[-8:]
    for t in last[-3:]:
        ks = np.searchsorted(clk[t, :n[t]] - t0, grid)
        print(f"   trajectory {t} ({n[t]} iterations, wave {t // 4}, wave-mates' iterations {n[t // 4 * 4: t // 4 * 4 + 4].tolist()}): iterations done at those times {ks.tolist()}")
    # mean duration of an iteration of those trajectories per 100-ms window
    d = np.diff(clk[last[-8:], :], axis=1)
    for w0 in np.arange(0, end.max(), 100.0):
        sel = [(d[i, k]) for i, t in enumerate(last[-8:]) for k in range(n[t] - 1) if w0 <= clk[t, k + 1] - t0 < w0 + 100]
        if sel:
            print(f"   window {int(w0):4d}..{int(w0) + 100:4d} ms: an iteration of the last 8 trajectories takes {np.mean(sel):.2f} ms (n = {len(sel)})")
s.close()
