// tsat_kernels_packed.hip — the "packed" builds of the solve kernel for batches larger than the machine: PK_G trajectories per
// wavefront share every forward sweep and run their backward sweeps together (tsat_packed.hpp). Bit-identical results to the wide
// and dense builds. Separate translation units because LDS size, register budget and wavefronts per SIMD are per-kernel
// compile-time facts; this file is the body of all of them:
//   itself                      PK_G = 4, two wavefronts per SIMD (the dense build's budget: 20 KB of LDS, 256 registers)
//   tsat_kernels_packed8.hip    PK_G = 8, two per SIMD, one single-buffered four-knot forward chunk
//   tsat_kernels_packed4w.hip, ...8w.hip, ...16w.hip    PK_G = 4 / 8 / 16 at ONE wavefront per SIMD (40 KB, 512 registers): what
//                               the automatic choice takes from 2048 / 4097 / 16384 trajectories on (tsat_kernels.hip)
//   ..._mixed.hip               each of them with float linearisation (TSAT_JAC32, options.precision = 32)
#define TSAT_DENSE 1
#define TSAT_PACKED 1
#ifndef TSAT_PK_WAVES
#define TSAT_PK_WAVES 2      /* wavefronts per SIMD the kernels are compiled for */
#endif
#ifndef TSAT_PK_NAME
#define TSAT_PK_NAME(base) base
#endif
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "tsat_packed.hpp"

using namespace tsat;

template <typename real, int INTEG, int DIAGJ, int ES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TSAT_PK_WAVES, TSAT_PK_WAVES))) void TSAT_PK_NAME(tsat_solve_kernel_packed)(KArgs<real> a) {
  const int wave = blockIdx.x;
  if (wave * PK_G >= a.T) return;
  solve_group<real, INTEG, DIAGJ, ES>(a, wave);
}

// Endgame (tsat_packed.hpp, suspend_if_endgame): the trajectories the solve kernel parked, one per wavefront from here on.
// Launched with a.suspend_at blocks; the parked count is known on the device only. The kernel runs ONE trajectory per wavefront in
// the dense layout, which two wavefronts per SIMD serve best: the one-wavefront-per-SIMD units (TSAT_PK_WAVES == 1) do not compile
// one of their own but launch that of the two-wavefront unit of their precision (a Resume record is the same everywhere).
#if TSAT_PK_WAVES == 1
#ifdef TSAT_JAC32
#define TSAT_PK_RESUME tsat_launch_resume_packed_mixed
#else
#define TSAT_PK_RESUME tsat_launch_resume_packed
#endif
hipError_t TSAT_PK_RESUME(const KArgs<double>& c, int rk4, int inertia_class, int error_state, hipStream_t stream);
#else
#define TSAT_PK_RESUME TSAT_PK_NAME(tsat_launch_resume_packed)
template <typename real, int INTEG, int DIAGJ, int ES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TSAT_PK_WAVES, TSAT_PK_WAVES))) void TSAT_PK_NAME(tsat_resume_kernel_packed)(KArgs<real> a) {
  const int w = blockIdx.x;
  if (w >= *a.susp_n) return;
  (void)continue_trajectory<real, INTEG, DIAGJ, ES>(a, a.susp_ids[w], reinterpret_cast<const Resume<real>*>(a.susp_state)[w]);
}
// c: the arguments of the solve launch that parked the trajectories (c.max_ls included: the slab numbering of Resume::cur)
hipError_t TSAT_PK_RESUME(const KArgs<double>& c, int rk4, int inertia_class, int error_state, hipStream_t stream) {
  using kern_t = void (*)(KArgs<double>);
  static const kern_t resume[2][3][2] = {
      {{TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 3, 0, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 3, 0, 1>},
       {TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 3, 1, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 3, 1, 1>},
       {TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 3, 2, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 3, 2, 1>}},
      {{TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 4, 0, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 4, 0, 1>},
       {TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 4, 1, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 4, 1, 1>},
       {TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 4, 2, 0>, TSAT_PK_NAME(tsat_resume_kernel_packed)<double, 4, 2, 1>}}};
  KArgs<double> r = c;
  r.live = nullptr;          // nothing parks in the second launch
  hipLaunchKernelGGL(resume[rk4 ? 1 : 0][inertia_class][error_state ? 1 : 0], dim3((unsigned)c.suspend_at), dim3(64), 0, stream, r);
  return hipGetLastError();
}
#endif
__global__ void TSAT_PK_NAME(tsat_endgame_init_kernel)(int* live, int* susp_n, int T) { *live = T; *susp_n = 0; }

// called by tsat_kernels.hip; same variant axes as the other builds: integrator x inertia class x error-state mode
hipError_t TSAT_PK_NAME(tsat_launch_solve_packed)(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream) {
  using kern_t = void (*)(KArgs<double>);
  static const kern_t variants[2][3][2] = {
      {{TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 3, 0, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 3, 0, 1>},
       {TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 3, 1, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 3, 1, 1>},
       {TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 3, 2, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 3, 2, 1>}},
      {{TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 4, 0, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 4, 0, 1>},
       {TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 4, 1, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 4, 1, 1>},
       {TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 4, 2, 0>, TSAT_PK_NAME(tsat_solve_kernel_packed)<double, 4, 2, 1>}}};
  const unsigned waves = (unsigned)((a.T + PK_G - 1) / PK_G);
  // Stored line-search candidates per sweep: PK_STORE instead of NSTORE = 12. On the Monte-Carlo workloads the accepted step is
  // alpha = 2^-j with j <= 5 in 99.8 % of the iterations; a deeper winner costs the wave one more sweep (shift += n_store),
  // while every stored candidate costs 80 B per knot and sweep of HBM writes whether it wins or not.
  KArgs<double> b = a;
  b.max_ls = a.max_ls < PK_STORE ? a.max_ls : PK_STORE;
#ifdef TSAT_PROFILE   // tuning knobs of the diagnostic build only (tools/store_probe.py): the product's launches do not read the environment
  if (const char* e = getenv("TSAT_PK_STORE")) { const int v = atoi(e); if (v >= 1 && v < b.max_ls) b.max_ls = v; }
  if (const char* e = getenv("TSAT_PK_FEW")) { const int v = atoi(e); if (v >= 1) b.pk_few = v; }
#endif
  const bool endgame = b.suspend_at > 0 && b.live && b.susp_n && b.susp_ids && b.susp_state;
  if (!endgame) b.suspend_at = 0;
  else hipLaunchKernelGGL(TSAT_PK_NAME(tsat_endgame_init_kernel), dim3(1), dim3(1), 0, stream, b.live, b.susp_n, b.T);
  hipLaunchKernelGGL(variants[rk4 ? 1 : 0][inertia_class][error_state ? 1 : 0], dim3(waves), dim3(64), 0, stream, b);
  if (endgame) return TSAT_PK_RESUME(b, rk4, inertia_class, error_state, stream);
  return hipGetLastError();
}
int TSAT_PK_NAME(tsat_packed_group)(void) { return PK_G; }
