// tsat_kernels_f32_o3.hip — the fp32 build of the solve kernel laid out for 3 wavefronts per SIMD (see tsat_kernels_f32.hip)
#define TSAT_OCC 3
#include "tsat_kernels_f32.hip"
