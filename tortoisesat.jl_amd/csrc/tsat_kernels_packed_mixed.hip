// tsat_kernels_packed_mixed.hip — the mixed-precision packed build (options.precision = 32): tsat_kernels_packed.hip compiled
// with TSAT_JAC32 (float linearisation: Jacobian lanes and knot records; everything else as in the fp64 build). Eight instead of
// four knots of every trajectory stay in the LDS record ring, so half as many records travel through the wavefront's workspace,
// at half the size.
#define TSAT_JAC32 1
#ifndef TSAT_PK_NAME
#define TSAT_PK_NAME(base) base##_mixed
#endif
#include "tsat_kernels_packed.hip"
