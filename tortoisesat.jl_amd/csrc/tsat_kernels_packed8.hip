// tsat_kernels_packed8.hip — the packed build with EIGHT trajectories per wavefront (see tsat_kernels_packed.hip)
#define TSAT_PK_G 8
#define TSAT_PK_CK 2
#define TSAT_PK_NAME(base) base##8
#include "tsat_kernels_packed.hip"
