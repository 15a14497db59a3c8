// tsat_kernels_dense.hip — the "dense" build of the solve kernel: the same source (tsat_device.hpp) compiled with
// TSAT_DENSE — 25-knot Jacobian chunks (19 992 B of LDS) and at most 256 registers — so that TWO wavefronts fit a SIMD.
// One wave per SIMD uses 45 % of the fp64 issue rate, two use 65 % (profiles/r01/lane_mask_ubench.txt): for batches that
// do not fit the GPU at one wave per SIMD (> 1024 trajectories) this build is 16-28 % faster (profiles/r01/large_batch.txt);
// at <= 1024 the wide build in tsat_kernels.hip is 11 % faster. Results are bit-identical (the chunking does not touch
// the arithmetic). Separate translation unit because LDS size and register budget are per-kernel compile-time facts.
// Compiled a second time through tsat_kernels_dense_mixed.hip (-DTSAT_JAC32: float linearisation, options.precision = 32).
#define TSAT_DENSE 1
#ifndef TSAT_DENSE_NAME
#define TSAT_DENSE_NAME(base) base
#endif
#include <hip/hip_runtime.h>
#include "tsat_device.hpp"

using namespace tsat;

template <typename real, int INTEG, int DIAGJ, int ES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void TSAT_DENSE_NAME(tsat_solve_kernel_dense)(KArgs<real> a) {
  const int traj = blockIdx.x;
  if (traj >= a.T) return;
  solve_trajectory<real, INTEG, DIAGJ, ES>(a, traj);
}

// called by tsat_kernels.hip; same variant axes as the wide build: integrator x inertia class x error-state mode
hipError_t TSAT_DENSE_NAME(tsat_launch_solve_dense)(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream) {
  using kern_t = void (*)(KArgs<double>);
  static const kern_t variants[2][3][2] = {
      {{TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 3, 0, 0>, TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 3, 0, 1>},
       {TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 3, 1, 0>, TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 3, 1, 1>},
       {TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 3, 2, 0>, TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 3, 2, 1>}},
      {{TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 4, 0, 0>, TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 4, 0, 1>},
       {TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 4, 1, 0>, TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 4, 1, 1>},
       {TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 4, 2, 0>, TSAT_DENSE_NAME(tsat_solve_kernel_dense)<double, 4, 2, 1>}}};
  hipLaunchKernelGGL(variants[rk4 ? 1 : 0][inertia_class][error_state ? 1 : 0], dim3((unsigned)a.T), dim3(64), 0, stream, a);
  return hipGetLastError();
}
