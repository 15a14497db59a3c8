// Developer probe: register footprint of the IGRF evaluation alone (hipcc -S -Rpass-analysis=kernel-resource-usage).
#include <hip/hip_runtime.h>
#include "../../tortoisesat.jl_amd/csrc/tsat_device.hpp"
using namespace tsat;
__global__ __launch_bounds__(64) void probe(const double* gh, const double* in, double* out) {
  double b[3];
  igrf12_eval<double>((const TSAT_CONSTMEM double*)gh, 6771.0, in[2 * threadIdx.x], in[2 * threadIdx.x + 1], b);
  out[3 * threadIdx.x] = b[0]; out[3 * threadIdx.x + 1] = b[1]; out[3 * threadIdx.x + 2] = b[2];
}
