// Microbenchmark: does a wave64 fp64 VALU instruction get cheaper when only the low lanes are active (EXEC mask)?
// Decides whether the forward sweep should pack its line-search candidates into lanes 0..15.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T>
__global__ __launch_bounds__(64) void chain(T* out, int iters, int active, int offset) {
  const int lane = threadIdx.x;
  T a0 = lane * (T)1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const T m = (T)0.999, c = (T)1e-4;
  if (lane >= offset && lane < offset + active) {
    for (int i = 0; i < iters; ++i) {
      a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
      a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
    }
  }
  out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename T>
void run(const char* name, int waves_per_simd) {
  const int blocks = 1024 * waves_per_simd, iters = 1 << 16;
  T* d;
  hipMalloc(&d, blocks * 64 * sizeof(T));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int cfg[][2] = {{64, 0}, {32, 0}, {32, 32}, {16, 0}, {16, 16}, {16, 48}, {8, 0}, {20, 0}, {1, 0}, {48, 0}};
  for (auto& c : cfg) {
    chain<T><<<blocks, 64>>>(d, 1024, c[0], c[1]);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<T><<<blocks, 64>>>(d, iters, c[0], c[1]);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double per = ms * 1e-3 / (double(iters) * 8) * 1e9;  // ns per FMA instruction per wave
    printf("%s waves/SIMD=%d active=%2d offset=%2d : %8.3f ms  %6.3f ns/instr (~%.2f cyc @2.4GHz)\n", name, waves_per_simd,
           c[0], c[1], ms, per, per * 2.4);
  }
  hipFree(d);
}

int main() {
  run<double>("f64", 1);
  run<double>("f64", 2);
  run<float>("f32", 1);
  return 0;
}
