// tsat_kernels_packed8_f32.hip — the fp32 packed build with EIGHT trajectories per wavefront (see tsat_kernels_packed_f32.hip)
#define TSAT_PK_G 8
#define TSAT_PK_CK 2
#define TSAT_PK_NAME(base) base##8
#include "tsat_kernels_packed_f32.hip"
