// tsat_kernels_packed_f32.hip — the packed build (tsat_packed.hpp: PK_G trajectories per wavefront share the forward sweeps and the
// Riccati recursion) in float storage / arithmetic with double costs: options.precision = 32 on batches several times larger
// than the machine (BASELINE.json configs[2]: 16384 trajectories, "fp32"). Two wavefronts per SIMD like the fp64 packed build (the
// float carve-up would fit three, the registers do not). Same semantics as the one-trajectory fp32 build (tsat_kernels_f32.hip); separate translation unit
// because LDS size and register budget are per-kernel compile-time facts. Compiled twice, like tsat_kernels_packed.hip: itself
// (four trajectories per wavefront) and through tsat_kernels_packed8_f32.hip (eight).
#define TSAT_F32 1
#define TSAT_OCC 2          /* chunk constants of the one-trajectory code paths this build does not use */
#define TSAT_DENSE 1
#define TSAT_PACKED 1
#ifndef TSAT_PKF_WAVES
#define TSAT_PKF_WAVES 2    /* three fit the LDS (-DTSAT_PK_LDS_BYTES=13568) but not the registers: 168 spill into the hot loops (measured 1.75x slower) */
#endif
#ifndef TSAT_PK_NAME
#define TSAT_PK_NAME(base) base
#endif
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "tsat_packed.hpp"

using namespace tsat;

template <int INTEG, int DIAGJ, int ES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TSAT_PKF_WAVES, TSAT_PKF_WAVES))) void TSAT_PK_NAME(tsat_solve_kernel_packed_f32)(KArgs<float> a) {
  const int wave = blockIdx.x;
  if (wave * PK_G >= a.T) return;
  solve_group<float, INTEG, DIAGJ, ES>(a, wave);
}

// endgame of the launch: the parked trajectories, one per wavefront (see tsat_kernels_packed.hip)
template <int INTEG, int DIAGJ, int ES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TSAT_PKF_WAVES, TSAT_PKF_WAVES))) void TSAT_PK_NAME(tsat_resume_kernel_packed_f32)(KArgs<float> a) {
  const int w = blockIdx.x;
  if (w >= *a.susp_n) return;
  (void)continue_trajectory<float, INTEG, DIAGJ, ES>(a, a.susp_ids[w], reinterpret_cast<const Resume<float>*>(a.susp_state)[w]);
}
__global__ void TSAT_PK_NAME(tsat_endgame_init_kernel_f32)(int* live, int* susp_n, int T) { *live = T; *susp_n = 0; }

hipError_t TSAT_PK_NAME(tsat_launch_solve_packed_f32)(const KArgs<float>& a, int rk4, int inertia_class, int error_state, hipStream_t stream) {
  using kern_t = void (*)(KArgs<float>);
  static const kern_t variants[2][3][2] = {
      {{TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<3, 0, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<3, 0, 1>},
       {TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<3, 1, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<3, 1, 1>},
       {TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<3, 2, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<3, 2, 1>}},
      {{TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<4, 0, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<4, 0, 1>},
       {TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<4, 1, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<4, 1, 1>},
       {TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<4, 2, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed_f32)<4, 2, 1>}}};
  const unsigned waves = (unsigned)((a.T + PK_G - 1) / PK_G);
  KArgs<float> b = a;
  b.max_ls = a.max_ls < PK_STORE ? a.max_ls : PK_STORE;      // stored candidates per sweep, as in tsat_kernels_packed.hip
  if (const char* e = getenv("TSAT_PK_FEW")) { const int v = atoi(e); if (v >= 1) b.pk_few = v; }   // tuning (tools/store_probe.py)
  static const kern_t resume[2][3][2] = {
      {{TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<3, 0, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<3, 0, 1>},
       {TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<3, 1, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<3, 1, 1>},
       {TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<3, 2, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<3, 2, 1>}},
      {{TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<4, 0, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<4, 0, 1>},
       {TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<4, 1, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<4, 1, 1>},
       {TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<4, 2, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed_f32)<4, 2, 1>}}};
  const bool endgame = b.suspend_at > 0 && b.live && b.susp_n && b.susp_ids && b.susp_state;
  if (!endgame) b.suspend_at = 0;
  else hipLaunchKernelGGL(TSAT_PK_NAME(tsat_endgame_init_kernel_f32), dim3(1), dim3(1), 0, stream, b.live, b.susp_n, b.T);
  hipLaunchKernelGGL(variants[rk4 ? 1 : 0][inertia_class][error_state ? 1 : 0], dim3(waves), dim3(64), 0, stream, b);
  if (endgame) {
    KArgs<float> c = b;
    c.live = nullptr;          // nothing parks in the second launch
    hipLaunchKernelGGL(resume[rk4 ? 1 : 0][inertia_class][error_state ? 1 : 0], dim3((unsigned)b.suspend_at), dim3(64), 0, stream, c);
  }
  return hipGetLastError();
}
