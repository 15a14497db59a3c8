// tsat_kernels_f32_o4.hip — the fp32 build of the solve kernel laid out for 4 wavefronts per SIMD (see tsat_kernels_f32.hip)
#define TSAT_OCC 4
#include "tsat_kernels_f32.hip"
