// tsat_kernels_f32.hip — the fp32 build of the solve kernel (options.precision = 32; BASELINE.json configs[2]): the same
// source (tsat_device.hpp) compiled with TSAT_F32. Every HBM array (x,u records, gains, multipliers, stored line-search
// candidates, field tables, parameter records) and every LDS record is float — half the traffic, half the LDS (20.4 KB
// per wavefront with the wide chunking, so eight wavefronts share a CU) and one register per value — and the dynamics,
// Jacobian and Riccati arithmetic run at the fp32 VALU rate. What stays in double: the cost of every rollout, the expected
// reductions dV1 / dV2, the line-search ratio and the convergence tests (J ~ 1e4 must resolve dJ ~ 1e-4; the fp32 build
// appends a 64-double reduction scratch to its LDS block for that), and the table clock (row = floor(fma(k + c, dtau, tau0))
// from (hi, lo) float pairs). Separate translation unit because LDS size and register budget are per-kernel facts.
//
// The file is compiled three times — itself (TSAT_OCC = 2) and through tsat_kernels_f32_o3.hip / _o4.hip — for LDS budgets
// of two, three and four wavefronts per SIMD (tsat_device.hpp: TSAT_OCC); tsat_kernels.hip picks the build by batch size.
#define TSAT_F32 1
#ifndef TSAT_OCC
#define TSAT_OCC 2
#endif
#define TSAT_CAT2(a, b) a##b
#define TSAT_CAT(a, b) TSAT_CAT2(a, b)
#define TSAT_F32_NAME(base) TSAT_CAT(base, TSAT_OCC)
#include <hip/hip_runtime.h>
#include "tsat_device.hpp"

using namespace tsat;

template <int INTEG, int DIAGJ, int ES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TSAT_OCC, TSAT_OCC))) void TSAT_F32_NAME(tsat_solve_kernel_f32_o)(KArgs<float> a) {
  const int traj = blockIdx.x;
  if (traj >= a.T) return;
  solve_trajectory<float, INTEG, DIAGJ, ES>(a, traj);
}

// called by tsat_kernels.hip; same variant axes as the fp64 builds: integrator x inertia class x error-state mode
hipError_t TSAT_F32_NAME(tsat_launch_solve_f32_o)(const KArgs<float>& a, int rk4, int inertia_class, int error_state, hipStream_t stream) {
  using kern_t = void (*)(KArgs<float>);
  static const kern_t variants[2][3][2] = {
      {{TSAT_F32_NAME(tsat_solve_kernel_f32_o)<3, 0, 0>, TSAT_F32_NAME(tsat_solve_kernel_f32_o)<3, 0, 1>},
       {TSAT_F32_NAME(tsat_solve_kernel_f32_o)<3, 1, 0>, TSAT_F32_NAME(tsat_solve_kernel_f32_o)<3, 1, 1>},
       {TSAT_F32_NAME(tsat_solve_kernel_f32_o)<3, 2, 0>, TSAT_F32_NAME(tsat_solve_kernel_f32_o)<3, 2, 1>}},
      {{TSAT_F32_NAME(tsat_solve_kernel_f32_o)<4, 0, 0>, TSAT_F32_NAME(tsat_solve_kernel_f32_o)<4, 0, 1>},
       {TSAT_F32_NAME(tsat_solve_kernel_f32_o)<4, 1, 0>, TSAT_F32_NAME(tsat_solve_kernel_f32_o)<4, 1, 1>},
       {TSAT_F32_NAME(tsat_solve_kernel_f32_o)<4, 2, 0>, TSAT_F32_NAME(tsat_solve_kernel_f32_o)<4, 2, 1>}}};
  hipLaunchKernelGGL(variants[rk4 ? 1 : 0][inertia_class][error_state ? 1 : 0], dim3((unsigned)a.T), dim3(64), 0, stream, a);
  return hipGetLastError();
}
