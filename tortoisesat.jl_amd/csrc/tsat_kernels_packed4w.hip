// tsat_kernels_packed4w.hip — four trajectories per wavefront at ONE wavefront per SIMD (see tsat_kernels_packed8w.hip): sixteen
// line-search candidates per trajectory, one backward pass per wavefront, twelve-knot record ring, double-buffered forward chunks.
// For 3072 .. 4096 trajectories — one round of the machine's 1024 SIMDs — e.g. the receding-horizon loop of BASELINE.json
// configs[4], whose short solves (200 knots, 1 x 3 budget) gain most: 3.95 -> 2.84 ms per control step.
#define TSAT_PK_G 4
#define TSAT_PK_CK 4
#define TSAT_PK_NBUF 2
#define TSAT_PK_WAVES 1
#define TSAT_PK_LDS_BYTES 40960
#define TSAT_PK_RING 12
#define TSAT_PK_NAME(base) base##4w
#include "tsat_kernels_packed.hip"
