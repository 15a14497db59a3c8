// tsat_kernels_packed16w_mixed.hip — the mixed-precision build of tsat_kernels_packed16w.hip (all sixteen knots of a pass in the LDS ring)
#define TSAT_JAC32 1
#define TSAT_PK_G 16
#define TSAT_PK_CK 4
#define TSAT_PK_NBUF 1
#define TSAT_PK_STORE 4
#define TSAT_PK_WAVES 1
#define TSAT_PK_LDS_BYTES 40960
#define TSAT_PK_RING 16
#define TSAT_PK_NAME(base) base##_mixed16w
#include "tsat_kernels_packed.hip"
