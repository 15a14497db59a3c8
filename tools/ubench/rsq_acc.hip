// accuracy of v_rsq_f64 / v_rcp_f64 seeds with 0/1/2 Newton steps (decides how many steps rsqrt_/rcp_ need)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* s, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = s[i];
  double y0 = __builtin_amdgcn_rsq(x);
  double e = x * y0 * y0; double y1 = y0 * __builtin_fma(-0.5, e, 1.5);
  e = x * y1 * y1; double y2 = y1 * __builtin_fma(-0.5, e, 1.5);
  double r0 = __builtin_amdgcn_rcp(x);
  double r1 = r0 * __builtin_fma(-x, r0, 2.0);
  double r2 = r1 * __builtin_fma(-x, r1, 2.0);
  { double e3 = __builtin_fma(-(x * y0), y0, 1.0); double p3 = __builtin_fma(0.375, e3, 0.5) * e3; y2 = __builtin_fma(y0, p3, y0); }
  { double e3 = __builtin_fma(-x, r0, 1.0); double p3 = __builtin_fma(e3, e3, e3); r2 = __builtin_fma(r0, p3, r0); }
  o[6 * i] = y0; o[6 * i + 1] = y1; o[6 * i + 2] = y2; o[6 * i + 3] = r0; o[6 * i + 4] = r1; o[6 * i + 5] = r2;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> s(n), o(6 * n);
  for (int i = 0; i < n; ++i) s[i] = 0.25 + 3.75 * (double)rand() / RAND_MAX * ((i & 1) ? 1.0 : 1e-3 + 1.0);
  double *ds, *dout; hipMalloc(&ds, n * 8); hipMalloc(&dout, 6 * n * 8);
  hipMemcpy(ds, s.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(ds, dout, n); hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
  double m[6] = {0};
  for (int i = 0; i < n; ++i) {
    long double ry = 1.0L / sqrtl((long double)s[i]), rr = 1.0L / (long double)s[i];
    for (int j = 0; j < 3; ++j) { double e = fabsl((o[6 * i + j] - ry) / ry); if (e > m[j]) m[j] = e; }
    for (int j = 3; j < 6; ++j) { double e = fabsl((o[6 * i + j] - rr) / rr); if (e > m[j]) m[j] = e; }
  }
  printf("rsq: seed %.3e  1 Newton %.3e  1 third-order step %.3e\nrcp: seed %.3e  1 Newton %.3e  1 third-order step %.3e  (eps = 1.1e-16)\n", m[0], m[1], m[2], m[3], m[4], m[5]);
}
