// tsat_kernels_packed8w_mixed.hip — the mixed-precision build (float linearisation, tsat_kernels_packed_mixed.hip) of
// tsat_kernels_packed8w.hip: with float records ALL sixteen knots of a pass stay in the LDS ring — nothing goes through the workspace
#define TSAT_JAC32 1
#define TSAT_PK_G 8
#define TSAT_PK_CK 4
#define TSAT_PK_NBUF 2
#define TSAT_PK_WAVES 1
#define TSAT_PK_LDS_BYTES 40960
#define TSAT_PK_RING 16
#define TSAT_PK_NAME(base) base##_mixed8w
#include "tsat_kernels_packed.hip"
