// tsat_kernels_packed16w.hip — SIXTEEN trajectories per wavefront (four line-search candidates each) at one wavefront per SIMD
// (see tsat_kernels_packed8w.hip): 16384 trajectories are one round of the machine's 1024 SIMDs, larger batches go through in
// rounds that start as wavefronts finish. One single-buffered four-knot forward chunk (31 KB for sixteen trajectories).
#define TSAT_PK_G 16
#define TSAT_PK_CK 4
#define TSAT_PK_NBUF 1
#define TSAT_PK_STORE 4
#define TSAT_PK_WAVES 1
#define TSAT_PK_LDS_BYTES 40960
#define TSAT_PK_RING 12
#define TSAT_PK_NAME(base) base##16w
#include "tsat_kernels_packed.hip"
