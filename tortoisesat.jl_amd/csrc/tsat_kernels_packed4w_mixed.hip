// tsat_kernels_packed4w_mixed.hip — the mixed-precision build of tsat_kernels_packed4w.hip (all sixteen knots of a pass in the LDS ring)
#define TSAT_JAC32 1
#define TSAT_PK_G 4
#define TSAT_PK_CK 4
#define TSAT_PK_NBUF 2
#define TSAT_PK_WAVES 1
#define TSAT_PK_LDS_BYTES 40960
#define TSAT_PK_RING 16
#define TSAT_PK_NAME(base) base##_mixed4w
#include "tsat_kernels_packed.hip"
