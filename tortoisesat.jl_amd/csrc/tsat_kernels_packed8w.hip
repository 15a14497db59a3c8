// tsat_kernels_packed8w.hip — eight trajectories per wavefront at ONE wavefront per SIMD (tsat_kernels_packed.hip with 40 KB of LDS
// and the whole register file): twelve of a pass's sixteen knot records stay in the LDS ring and four go through the HBM workspace
// (two wavefronts per SIMD: four stay, twelve go), the forward chunks are double-buffered. For batches of 8192 .. 16383
// trajectories, which are one round of the machine's 1024 SIMDs this way. Same bits as every other build.
#define TSAT_PK_G 8
#define TSAT_PK_CK 4
#define TSAT_PK_NBUF 2
#define TSAT_PK_WAVES 1
#define TSAT_PK_LDS_BYTES 40960
#define TSAT_PK_RING 12
#define TSAT_PK_NAME(base) base##8w
#include "tsat_kernels_packed.hip"
