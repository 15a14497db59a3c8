// tsat_kernels_packed8_mixed.hip — the mixed-precision packed build with EIGHT trajectories per wavefront (see
// tsat_kernels_packed_mixed.hip, tsat_kernels_packed8.hip)
#define TSAT_PK_G 8
#define TSAT_PK_CK 4
#define TSAT_PK_NBUF 1
#define TSAT_PK_NAME(base) base##_mixed8
#include "tsat_kernels_packed_mixed.hip"
