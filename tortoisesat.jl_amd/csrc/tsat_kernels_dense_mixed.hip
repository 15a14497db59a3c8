// tsat_kernels_dense_mixed.hip — the mixed-precision build (options.precision = 32; BASELINE.json configs[2]) of the one-trajectory
// mapping at two wavefronts per SIMD: tsat_kernels_dense.hip compiled with TSAT_JAC32 (tsat_device.hpp: jac_t = float). The
// Jacobian lanes — nine tangent passes per knot, the bulk of the flops — run in float and leave float knot records (59-knot
// passes in the 20 KB that hold 29 double records); the roll-out, the costs, the Riccati recursion, the gains, the multipliers,
// the field tables and every array in HBM are those of the fp64 builds, so the accept / reject decisions of the line search
// follow the fp64 path. Serves batches below 3072 trajectories and, inside the packed mixed builds, the last live trajectory
// of a wavefront.
#define TSAT_JAC32 1
#define TSAT_DENSE_NAME(base) base##_mixed
#include "tsat_kernels_dense.hip"
