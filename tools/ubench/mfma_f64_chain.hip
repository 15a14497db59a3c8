// Microbenchmark + layout check for a Riccati step on v_mfma_f64_16x16x4_f64 (candidate for the next round).
//   1. layouts: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], C/D[row = (l>>4) + 4 reg][col = l&15]. Consequences checked
//      here with exact integer data: (a) a product D = X Y can be fed as the B operand of the next product summing over D's
//      rows, register s = k-step s, no lane movement; (b) the register that held F as B operand IS F^T as A operand;
//      (c) a SYMMETRIC D can be fed as the A operand the same way.
//   2. timing: one wave per SIMD, a dependent chain W = S F (3 k-steps), M = F^T W (3 k-steps), S <- scaled M — the
//      two big products of the backward recursion with [u; x; 1] ordering (11 x 11 padded to 16 x 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ inline d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// S, F: 16 x 16 row-major (zero padded). out: W (S F), M (F^T W), P (M F)
__global__ __launch_bounds__(64) void check(const double* S, const double* F, double* W, double* M, double* P) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  double sa[4], fb[4];
  for (int s = 0; s < 4; ++s) { sa[s] = S[i * 16 + (g + 4 * s)]; fb[s] = F[(g + 4 * s) * 16 + i]; }   // A: S[i][k]; B: F[k][j]
  d4 w = {0, 0, 0, 0}, m = {0, 0, 0, 0}, p = {0, 0, 0, 0};
  for (int s = 0; s < 4; ++s) w = mfma(sa[s], fb[s], w);
  for (int s = 0; s < 4; ++s) m = mfma(fb[s], w[s], m);      // (b): fb as A = F^T;  (a): w[s] as B = rows 4s..4s+3 of W
  for (int s = 0; s < 4; ++s) p = mfma(m[s], fb[s], p);      // (c): symmetric M as A
  for (int r = 0; r < 4; ++r) {
    W[(g + 4 * r) * 16 + i] = w[r]; M[(g + 4 * r) * 16 + i] = m[r]; P[(g + 4 * r) * 16 + i] = p[r];
  }
}

template <int NM>
__global__ __launch_bounds__(64) void chain(double* out, int iters) {
  const int l = threadIdx.x;
  double sa[3], fb[3];
  for (int s = 0; s < 3; ++s) { sa[s] = 1e-3 * ((l * 7 + s) % 13); fb[s] = 1e-2 * ((l * 5 + s) % 11); }
  d4 m = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    d4 w = {0, 0, 0, 0};
    m = (d4){0, 0, 0, 0};
    for (int s = 0; s < 3; ++s) w = mfma(sa[s], fb[s], w);
    for (int s = 0; s < 3; ++s) m = mfma(fb[s], w[s], m);
    if (NM > 6) {   // three more dependent products (gain, Schur complement, regularisation term)
      d4 k = {0, 0, 0, 0};
      k = mfma(sa[0], m[0], k);
      m = mfma(m[0], k[0], m);
      m = mfma(k[0], k[0], m);
    }
    for (int s = 0; s < 3; ++s) sa[s] = 1e-3 * m[s] + 1e-3;
  }
  out[blockIdx.x * 64 + l] = m[0] + m[1] + m[2] + m[3];
}

int main() {
  std::vector<double> S(256, 0), F(256, 0), W(256), M(256), P(256);
  for (int i = 0; i < 11; ++i)
    for (int j = 0; j < 11; ++j) { S[i * 16 + j] = S[j * 16 + i] = (double)((i * 3 + j * 5) % 7 - 3 + (i == j ? 9 : 0)) * (i <= j ? 1 : 1); F[i * 16 + j] = (double)((i * 7 + j * 11 + 3) % 9 - 4); }
  for (int i = 0; i < 11; ++i) for (int j = 0; j < i; ++j) S[i * 16 + j] = S[j * 16 + i];
  double *dS, *dF, *dW, *dM, *dP;
  hipMalloc(&dS, 2048); hipMalloc(&dF, 2048); hipMalloc(&dW, 2048); hipMalloc(&dM, 2048); hipMalloc(&dP, 2048);
  hipMemcpy(dS, S.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dF, F.data(), 2048, hipMemcpyHostToDevice);
  check<<<1, 64>>>(dS, dF, dW, dM, dP);
  hipMemcpy(W.data(), dW, 2048, hipMemcpyDeviceToHost); hipMemcpy(M.data(), dM, 2048, hipMemcpyDeviceToHost); hipMemcpy(P.data(), dP, 2048, hipMemcpyDeviceToHost);
  double eW = 0, eM = 0, eP = 0;
  std::vector<double> Wh(256, 0), Mh(256, 0), Ph(256, 0);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double v = 0; for (int k = 0; k < 16; ++k) v += S[i * 16 + k] * F[k * 16 + j]; Wh[i * 16 + j] = v; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double v = 0; for (int k = 0; k < 16; ++k) v += F[k * 16 + i] * Wh[k * 16 + j]; Mh[i * 16 + j] = v; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double v = 0; for (int k = 0; k < 16; ++k) v += Mh[i * 16 + k] * F[k * 16 + j]; Ph[i * 16 + j] = v; }
  for (int e = 0; e < 256; ++e) { eW = fmax(eW, fabs(W[e] - Wh[e])); eM = fmax(eM, fabs(M[e] - Mh[e])); eP = fmax(eP, fabs(P[e] - Ph[e])); }
  printf("layout check (exact integers): max|W - S F| = %g, max|M - F^T W| = %g, max|P - M F| = %g  (|M| up to %g)\n", eW, eM, eP, Mh[5 * 16 + 5]);
  double* d;
  hipMalloc(&d, 1024 * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 1 << 15;
  for (int nm : {6, 9}) {
    if (nm == 6) chain<6><<<1024, 64>>>(d, 64); else chain<9><<<1024, 64>>>(d, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    if (nm == 6) chain<6><<<1024, 64>>>(d, iters); else chain<9><<<1024, 64>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;
    printf("chain of %d dependent f64 MFMAs + 6 VALU per step, 1 wave/SIMD: %.1f ns per step (~%.0f cycles @2.4 GHz), %.1f ns per MFMA\n", nm, ns, ns * 2.4, ns / nm);
  }
  return 0;
}
