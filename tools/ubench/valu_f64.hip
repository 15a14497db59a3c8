// Issue cost and dependent latency of the fp64 VALU forms the sequential sweeps are made of, one wavefront per SIMD (the
// situation of the one-trajectory mapping at T <= 1024), and of the cross-lane forms that could replace LDS exchanges:
//   v_fma_f64 independent / one dependent chain / 2, 3, 4 interleaved chains
//   v_fmac_f64_dpp row_newbcast (DP-ALU DPP: the broadcast of one lane of each 16-lane row as src0 of the FMA itself)
//   v_mov_b64_dpp row_newbcast, v_mul_f64_dpp, v_add_f64_dpp
//   v_readlane_b32 x2 -> SGPR pair as an FMA operand
//   v_mov_b32_dpp quad_perm x2 (a 64-bit lane permutation)
//   v_rsq_f64, v_rcp_f64
//   ds_write_b64 -> ds_read_b64 round trip (what a Riccati exchange costs today)
// Cycles are s_memtime ticks per instruction. Build: hipcc --offload-arch=gfx950 -O3 valu_f64.hip -o valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define STR2(x) #x
#define STR(x) STR2(x)
#define REPT 64

__device__ inline unsigned long long now() {
  unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return t;
}

// every test: out[0] = cycles of `iters` x REPT instruction groups, out[1] = a value (keeps the work alive / correctness)
#define TEST_BEGIN(name)                                                       \
  __global__ __launch_bounds__(64) void name(double* out, unsigned long long* cyc, int iters) { \
    const int l = threadIdx.x;                                                 \
    double a0 = 1.0 + 1e-9 * l, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7; \
    double x = 1.0 + 1e-12 * l, y = 1e-13 * (l + 1), z = 0.5;                  \
    (void)a1; (void)a2; (void)a3; (void)a4; (void)a5; (void)a6; (void)a7; (void)z; \
    const unsigned long long t0 = now();                                       \
    for (int it = 0; it < iters; ++it) {
#define TEST_END                                                               \
    }                                                                          \
    const unsigned long long t1 = now();                                       \
    if (l == 0) cyc[blockIdx.x] = t1 - t0;                                     \
    out[blockIdx.x * 64 + l] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + x;      \
  }

TEST_BEGIN(fma_indep8)
asm volatile(".rept " STR(REPT) "\n v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
             " v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n.endr"
             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
TEST_END
TEST_BEGIN(fma_dep1)
asm volatile(".rept " STR(REPT) "\n .rept 8\n v_fma_f64 %0, %0, %1, %2\n .endr\n.endr" : "+v"(a0) : "v"(x), "v"(y));
TEST_END
TEST_BEGIN(fma_dep2)
asm volatile(".rept " STR(REPT) "\n .rept 4\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n .endr\n.endr" : "+v"(a0), "+v"(a1) : "v"(x), "v"(y));
TEST_END
TEST_BEGIN(fma_dep4)
asm volatile(".rept " STR(REPT) "\n .rept 2\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n .endr\n.endr"
             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
TEST_END
TEST_BEGIN(mul_dep1)
asm volatile(".rept " STR(REPT) "\n .rept 8\n v_mul_f64 %0, %0, %1\n .endr\n.endr" : "+v"(a0) : "v"(x));
TEST_END
TEST_BEGIN(add_dep1)
asm volatile(".rept " STR(REPT) "\n .rept 8\n v_add_f64 %0, %0, %1\n .endr\n.endr" : "+v"(a0) : "v"(y));
TEST_END
// DP-ALU DPP: src0 = lane 3 of the row, independent accumulators
TEST_BEGIN(fmac_dpp_indep8)
asm volatile(".rept " STR(REPT) "\n v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
             " v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
             " v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
             " v_fmac_f64_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n.endr"
             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "v"(x));
TEST_END
// one accumulator chain through the DPP form (the accumulator is not the DPP operand: no DPP hazard)
TEST_BEGIN(fmac_dpp_dep1)
asm volatile(".rept " STR(REPT) "\n .rept 8\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n .endr\n.endr" : "+v"(a0) : "v"(y), "v"(x));
TEST_END
// the DPP source is the value just written (VALU write -> DPP read: 2 wait states, inserted by hand)
TEST_BEGIN(fmac_dpp_src_chain)
asm volatile(".rept " STR(REPT) "\n .rept 4\n v_mul_f64 %1, %0, %2\n s_nop 1\n v_fmac_f64_dpp %0, %1, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n .endr\n.endr"
             : "+v"(a0), "+v"(a1) : "v"(x), "v"(y));
TEST_END
TEST_BEGIN(mov_dpp64_indep)
asm volatile(".rept " STR(REPT) "\n v_mov_b64_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
             " v_mov_b64_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
             " v_mov_b64_dpp %4, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
             " v_mov_b64_dpp %6, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n.endr"
             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
TEST_END
TEST_BEGIN(mul_dpp_indep8)
asm volatile(".rept " STR(REPT) "\n .rept 8\n v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n .endr\n.endr" : "+v"(a0) : "v"(y), "v"(x));
TEST_END
// 64-bit lane permutation inside quads as two 32-bit DPP moves (8 permutations = 16 instructions per group)
TEST_BEGIN(mov_dpp32_quadperm_x2)
unsigned pl = 0, ph = 0; const unsigned xl = (unsigned)l, xh = (unsigned)l * 3u;
asm volatile(".rept " STR(REPT) "\n .rept 8\n v_mov_b32_dpp %0, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n .endr\n.endr"
             : "+v"(pl), "+v"(ph) : "v"(xl), "v"(xh));
a1 += (double)(pl + ph);
TEST_END
// v_readlane x2 -> SGPR pair -> FMA operand (8 per group: 16 readlanes + 8 FMAs = 24 instructions)
TEST_BEGIN(readlane_fma)
const unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x);
asm volatile(".rept " STR(REPT) "\n .rept 8\n v_readlane_b32 s20, %1, 7\n v_readlane_b32 s21, %2, 7\n v_fma_f64 %0, s[20:21], %3, %0\n .endr\n.endr"
             : "+v"(a0) : "v"(xl), "v"(xh), "v"(y) : "s20", "s21");
TEST_END
// readlanes batched first, the FMAs afterwards on other accumulators (no immediate SGPR dependency)
TEST_BEGIN(readlane_only)
const unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x);
asm volatile(".rept " STR(REPT) "\n .rept 4\n v_readlane_b32 s20, %0, 7\n v_readlane_b32 s21, %1, 7\n v_readlane_b32 s22, %0, 9\n v_readlane_b32 s23, %1, 9\n .endr\n.endr"
             : : "v"(xl), "v"(xh) : "s20", "s21", "s22", "s23");
TEST_END
TEST_BEGIN(rsq_indep8)
asm volatile(".rept " STR(REPT) "\n v_rsq_f64 %0, %8\n v_rsq_f64 %1, %8\n v_rsq_f64 %2, %8\n v_rsq_f64 %3, %8\n v_rcp_f64 %4, %8\n v_rcp_f64 %5, %8\n v_rcp_f64 %6, %8\n v_rcp_f64 %7, %8\n.endr"
             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
TEST_END
TEST_BEGIN(rsq_dep1)
asm volatile(".rept " STR(REPT) "\n .rept 8\n v_rsq_f64 %0, %0\n .endr\n.endr" : "+v"(a0));
TEST_END
// dependency distance: N interleaved chains for N = 5, 6, 7 (where does the one-cycle penalty of a dependent FMA end?)
TEST_BEGIN(fma_dep5)
asm volatile(".rept " STR(REPT) "\n v_fma_f64 %0, %0, %5, %6\n v_fma_f64 %1, %1, %5, %6\n v_fma_f64 %2, %2, %5, %6\n v_fma_f64 %3, %3, %5, %6\n v_fma_f64 %4, %4, %5, %6\n.endr"
             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(x), "v"(y));
TEST_END
TEST_BEGIN(fma_dep6)
asm volatile(".rept " STR(REPT) "\n v_fma_f64 %0, %0, %6, %7\n v_fma_f64 %1, %1, %6, %7\n v_fma_f64 %2, %2, %6, %7\n v_fma_f64 %3, %3, %6, %7\n v_fma_f64 %4, %4, %6, %7\n v_fma_f64 %5, %5, %6, %7\n.endr"
             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(x), "v"(y));
TEST_END
TEST_BEGIN(fma_dep8)
asm volatile(".rept " STR(REPT) "\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
             " v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n.endr"
             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
TEST_END
// the dependent operand as src2 (the accumulator of an fmac) instead of src0
TEST_BEGIN(fmac_acc_dep1)
asm volatile(".rept " STR(REPT) "\n .rept 8\n v_fmac_f64 %0, %1, %2\n .endr\n.endr" : "+v"(a0) : "v"(x), "v"(y));
TEST_END
// a dependent FMA followed by an independent one, alternating (half of the instructions pay the penalty?)
TEST_BEGIN(fma_dep_alt)
asm volatile(".rept " STR(REPT) "\n .rept 4\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %2, %3, %2\n .endr\n.endr" : "+v"(a0), "+v"(a1) : "v"(x), "v"(y));
TEST_END
// fp32 FMA for comparison
__global__ __launch_bounds__(64) void fma32_indep8(double* out, unsigned long long* cyc, int iters) {
  const int l = threadIdx.x;
  float a0 = 1.f + l, a1 = 1.1f, a2 = 1.2f, a3 = 1.3f, a4 = 1.4f, a5 = 1.5f, a6 = 1.6f, a7 = 1.7f, x = 1.0f, y = 1e-7f;
  const unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it)
    asm volatile(".rept " STR(REPT) "\n v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                 " v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n.endr"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
  const unsigned long long t1 = now();
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + l] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// LDS exchange as the Riccati steps do it: every lane writes a double, barrier-free (one wave), every lane reads another lane's
__global__ __launch_bounds__(64) void lds_roundtrip(double* out, unsigned long long* cyc, int iters) {
  __shared__ double buf[128];
  const int l = threadIdx.x;
  double a0 = 1.0 + l;
  const unsigned long long t0 = now();
  for (int it = 0; it < iters * REPT; ++it) {
    buf[l] = a0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    a0 = buf[(l * 7 + 3) & 63] + 1.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const unsigned long long t1 = now();
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + l] = a0;
}
// correctness of the DPP forms: lane l of row r (16 lanes) gets x[16 r + 5] * w[l] + acc
__global__ __launch_bounds__(64) void dpp_check(double* out) {
  const int l = threadIdx.x;
  double xv = 100.0 + l, w = 2.0 + 0.001 * l, acc = 0.5, m = 0, mu = 0, ad = 0;
  asm volatile("s_nop 4\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(xv), "v"(w));
  asm volatile("s_nop 4\n v_mov_b64_dpp %0, %1 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(xv));
  asm volatile("s_nop 4\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(mu) : "v"(xv), "v"(w));
  ad = w;
  asm volatile("s_nop 4\n v_fmac_f64_dpp %0, -%1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(ad) : "v"(xv), "v"(w));
  out[l * 4 + 0] = acc; out[l * 4 + 1] = m; out[l * 4 + 2] = mu; out[l * 4 + 3] = ad;
}

typedef void (*kern_t)(double*, unsigned long long*, int);
struct T { const char* name; kern_t k; int per_group; };

int main() {
  double* d; unsigned long long* c;
  hipMalloc(&d, 2048 * 64 * 8); hipMalloc(&c, 2048 * 8);
  std::vector<unsigned long long> hc(2048);
  {
    dpp_check<<<1, 64>>>(d);
    std::vector<double> h(256);
    hipMemcpy(h.data(), d, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      const int r = l / 16;
      const double w = 2.0 + 0.001 * l;
      const double e0 = __builtin_fma(100.0 + 16 * r + 5, w, 0.5), e1 = 100.0 + 16 * r + 9, e2 = (100.0 + 16 * r + 2) * w, e3 = __builtin_fma(-(100.0 + 16 * r + 15), w, w);
      if (h[l * 4] != e0 || h[l * 4 + 1] != e1 || h[l * 4 + 2] != e2 || h[l * 4 + 3] != e3) {
        if (bad < 4) printf("  lane %d: got %.6f %.6f %.6f %.6f, expected %.6f %.6f %.6f %.6f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], e0, e1, e2, e3);
        ++bad;
      }
    }
    printf("DP-ALU DPP row_newbcast forms (fmac, mov_b64, fmac with -src0): %s\n", bad ? "MISMATCH" : "correct on all 64 lanes");
  }
  const T tests[] = {
    {"v_fma_f64, 8 independent accumulators", fma_indep8, 8}, {"v_fma_f64, one dependent chain", fma_dep1, 8},
    {"v_fma_f64, 2 interleaved chains", fma_dep2, 8}, {"v_fma_f64, 4 interleaved chains", fma_dep4, 8},
    {"v_fma_f64, 5 interleaved chains", fma_dep5, 5}, {"v_fma_f64, 6 interleaved chains", fma_dep6, 6}, {"v_fma_f64, 8 interleaved chains (dependent through src0)", fma_dep8, 8},
    {"v_fmac_f64, one chain through the accumulator (src2)", fmac_acc_dep1, 8}, {"v_fma_f64, dependent / independent alternating", fma_dep_alt, 8},
    {"v_mul_f64, one dependent chain", mul_dep1, 8}, {"v_add_f64, one dependent chain", add_dep1, 8},
    {"v_fmac_f64_dpp row_newbcast, 8 independent", fmac_dpp_indep8, 8}, {"v_fmac_f64_dpp row_newbcast, one accumulator chain", fmac_dpp_dep1, 8},
    {"v_mul_f64 -> s_nop 1 -> v_fmac_f64_dpp on the product (chain; per pair+nop)", fmac_dpp_src_chain, 4},
    {"v_mov_b64_dpp row_newbcast, independent", mov_dpp64_indep, 8}, {"v_fmac_f64_dpp with -src0, one accumulator chain", mul_dpp_indep8, 8},
    {"v_mov_b32_dpp quad_perm x2 (one 64-bit permutation; per pair)", mov_dpp32_quadperm_x2, 8},
    {"v_readlane_b32 x2 + v_fma_f64 with the SGPR pair (per triple)", readlane_fma, 8}, {"v_readlane_b32 (per instruction)", readlane_only, 16},
    {"v_rsq_f64 / v_rcp_f64, independent", rsq_indep8, 8}, {"v_rsq_f64, dependent chain", rsq_dep1, 8},
    {"v_fma_f32, 8 independent accumulators", fma32_indep8, 8}, {"LDS write -> other lane reads (per round trip)", lds_roundtrip, 1},
  };
  const int iters = 200;
  for (int blocks : {1, 1024, 2048}) {
    printf("--- %d workgroups of one wavefront (%s)\n", blocks, blocks == 1 ? "alone on the chip" : blocks == 1024 ? "one per SIMD" : "two per SIMD");
    for (const T& t : tests) {
      t.k<<<blocks, 64>>>(d, c, 4);
      hipDeviceSynchronize();
      t.k<<<blocks, 64>>>(d, c, iters);
      hipDeviceSynchronize();
      hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; for (int b = 0; b < blocks; ++b) s += (double)hc[b];
      s /= blocks;
      printf("%-78s %7.2f cycles\n", t.name, s / ((double)iters * REPT * t.per_group));
    }
  }
  return 0;
}
