// tsat_kernels_packed8.hip — the packed build with EIGHT trajectories per wavefront (see tsat_kernels_packed.hip)
#define TSAT_PK_G 8
#define TSAT_PK_CK 4      /* with eight trajectories the LDS holds ONE four-knot forward chunk: the copy of the next one waits for it */
#define TSAT_PK_NBUF 1    /* (measured: 4 % faster than two double-buffered two-knot chunks, whose wait comes twice as often)          */
#define TSAT_PK_NAME(base) base##8
#include "tsat_kernels_packed.hip"
