"""tortoisesat.jl_amd — MI355X-native batched AL-iLQR for TortoiseSat.jl's magnetorquer-only attitude slew.

Only the hot path lives here: ``csrc/`` holds the hand-written HIP kernels and the C ABI
(include/tortoise_hip.h); the Python modules are the thin host side above that ABI:

  _abi        ctypes view of the C ABI (no fallback: raises if libtortoise_hip.so is not built)
  slew_setup  problem setup the reference keeps in its scripts (weights, guesses, workloads)
  trajopt     mirror of the TrajectoryOptimization.jl surface used at src/TortoiseSat.jl:145-199
  sweep       Monte-Carlo sharding over GPUs + RCCL all-gather (src/monte_carlo.jl:118-235)
  magnetic    orbit + IGRF-12 field tables (src/magnetic_toolbox.jl:33-106)
  horizon     Gramian-based horizon selection (src/magnetic_toolbox.jl:1-31)
  monte_carlo the whole experiment of src/monte_carlo.jl:107-262 as a batch; results: its HDF5 file set (:334-343);
              hdf5io: h5write / h5read over the system's libhdf5
  mpc         receding-horizon re-solve on the resident batch (BASELINE.json configs[4]; not in the reference)
  tracking    closed-loop TVLQR tracking + slew-time statistic (src/attitude_controller.jl:1-119)
"""
from . import _abi, hdf5io, horizon, magnetic, monte_carlo, mpc, results, slew_setup, sweep, tracking, trajopt  # noqa: F401

__all__ = ["_abi", "hdf5io", "horizon", "magnetic", "monte_carlo", "mpc", "results", "slew_setup", "sweep", "tracking", "trajopt"]
