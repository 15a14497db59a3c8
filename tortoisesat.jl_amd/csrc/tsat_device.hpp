// tsat_device.hpp — device code of the batched AL-iLQR slew solver (one trajectory per 64-lane wavefront).
//
// Replaces, for a whole batch, TrajectoryOptimization.solve!(prob, solver) (src/TortoiseSat.jl:199) and the
// dynamics callback it drives (src/DerivFunction.jl:1-48). Algorithm semantics: SURVEY.md Appendix A.
//
// Mapping of the solve onto a CDNA4 wavefront:
//   forward sweep  (sequential in k): lane j rolls out line-search candidate alpha = 2^-j — all backtracking
//                   trials of one iteration run in ONE sweep; knot records (x̄,ū,K,d,λ,B rows) are staged
//                   through LDS in CK-knot chunks by coalesced 16-byte loads and read back as broadcasts.
//   Jacobians      (parallel in k):   lane l linearises knot k0+l (analytic forward-mode through the RK stages)
//                   and leaves [A|B] + cost gradients in LDS for the Riccati step; nothing is stored to HBM.
//   Riccati sweep  (sequential in k): the 64 lanes are the elements of the 8x10 / 7x7 / 3x7 blocks; operands are
//                   broadcast through LDS; 4 dependent stages per knot.
//   AL/cost/copy   (parallel in k):   lane-strided over knots + butterfly reductions.
//
// The file is plain C++ over a tiny lane abstraction so that tests/emu can run it on the CPU with 64 host
// threads per wavefront (TSAT_EMU); the GPU build (tsat_kernels.hip) maps it onto threadIdx/__syncthreads.
#pragma once
#include <stdint.h>
#include "../../include/tortoise_hip.h"

#ifdef TSAT_EMU
#include <cmath>
namespace tsat_emu { int lane(); void sync(); void* lds(); double* xch(); }
#define TSAT_DEV inline
#define TSAT_PHASE inline
#define TSAT_FWD inline
#define TSAT_GLOBAL
#define TSAT_CONSTMEM
#define TSAT_LANE() (tsat_emu::lane())
#define TSAT_SYNC() (tsat_emu::sync())
#define TSAT_SYNC_LDS() (tsat_emu::sync())
#define TSAT_SCHED_FENCE() ((void)0)
#define TSAT_WAIT_LDS() ((void)0)
#define TSAT_NO_UNROLL
// device-wide counters (emulated wavefronts run on several host threads)
#define TSAT_ATOMIC_ADD(ptr, v) __atomic_fetch_add((ptr), (v), __ATOMIC_RELAXED)
#define TSAT_ATOMIC_LOAD(ptr) __atomic_load_n((ptr), __ATOMIC_RELAXED)
#define TSAT_UNIFORM_INT(x) (x)
#else
#define TSAT_DEV __device__ __forceinline__
// Each sweep is its own (non-inlined) function: the register allocator then works on one hot loop at a time
// instead of on the whole solve, which otherwise spills loop invariants of every phase into every other phase.
#define TSAT_PHASE __device__ __noinline__
// the forward sweep is inlined into the kernel body: there the register allocator can park values in AGPRs
// (one-instruction reload), while a called function has to spill to scratch memory. The dense build has no AGPRs to park
// anything in (256 registers in all, two wavefronts per SIMD): there — and in the packed builds, which hand a wavefront's last live
// trajectory over to this mapping — the sweep is a function of its own like the other phases, with its own register allocation,
// and the code around it keeps its counters and pointers out of scratch.
#if defined(TSAT_DENSE)
#define TSAT_FWD __device__ __noinline__
#else
#define TSAT_FWD __device__ __forceinline__
#endif
// HBM pointers that cross a (non-inlined) function boundary must carry their address space, otherwise every
// access through them is a flat_* instruction (both memory pipes, both wait counters) instead of global_*.
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
#define TSAT_GLOBAL __attribute__((address_space(1)))
#define TSAT_CONSTMEM __attribute__((address_space(4)))   /* read-only for the kernel's lifetime: uniform reads become s_load */
#else
#define TSAT_GLOBAL   /* host pass of the same translation unit: only parses the device functions */
#define TSAT_CONSTMEM
#endif
#define TSAT_LANE() ((int)threadIdx.x)
// full fence: orders global AND LDS traffic between the lanes of the wave (s_waitcnt vmcnt(0) lgkmcnt(0))
#define TSAT_SYNC() __syncthreads()
// LDS-only ordering between the lanes of ONE wavefront. A wave's DS instructions execute in issue order, so a
// ds_write followed in program order by another lane's ds_read of the same address needs no s_waitcnt; all that
// is required is that the compiler keeps the order. Crucially this does NOT drain vmcnt, so global stores issued
// inside a sequential sweep (candidates) stay in flight instead of stalling every knot on their write-ack.
#define TSAT_SYNC_LDS()                                       \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)
#define TSAT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// s_waitcnt lgkmcnt(0) as an instruction the compiler's own wait-count pass sees (vmcnt / expcnt fields at their maxima: no wait).
// Placed BEFORE a batch of LDS reads that is to stay in flight across the code that follows: lgkmcnt has four bits, so a use of
// OLDER data behind more than 15 newer reads can only be expressed as lgkmcnt(0), which would wait for the new batch as well.
#define TSAT_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)
#define TSAT_NO_UNROLL _Pragma("unroll 1")
#define TSAT_ATOMIC_ADD(ptr, v) atomicAdd((ptr), (v))
#define TSAT_ATOMIC_LOAD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// an int that is the same in every lane, said so to the compiler (lane 0's copy in an SGPR): a loop or branch on it is scalar
#define TSAT_UNIFORM_INT(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif

// Every build contracts a*b + c ONLY where the source writes it in one expression (the language's own rule, decided by the front end
// per expression): all builds of the solve kernel — one trajectory per wavefront in its LDS layouts, packed, packed8, both
// precisions — then round the same source expression the same way and agree bit for bit on the GPU as they do on the emulator.
// hipcc's default (fast: the back end fuses across statements, and picks which product of a*b + c*d to fuse differently in
// different code shapes) made builds differ by rounding whenever a loop was restructured — the packed float builds from the
// one-trajectory float build in round 2, the double builds from each other when the forward sweep got its two-knot read-ahead
// in round 3 — and a budget-limited solve amplifies a rounding into another iteration path. Costs seven more fp64 instructions
// per knot of the forward sweep (2.7 %) and nine per Jacobian knot. Where one expression holds two products, the fused one is
// still spelled out (dmm_, dot3_, fma_ chains): the emulator, compiled without any contraction, keeps exactly those.
#if !defined(TSAT_EMU)
#pragma clang fp contract(on)
#endif

namespace tsat {

// Storage / arithmetic type of the translation unit's solve kernel: double (tsat_kernels.hip, tsat_kernels_dense.hip) or
// float (tsat_kernels_f32.hip, -DTSAT_F32: options.precision = 32). The LDS carve-up below is counted in elements of this
// type, so the fp32 build needs half the LDS per wavefront. Whatever the storage type, costs, expected reductions and the
// line-search test are carried in double (acc_t): J ~ 1e4 has to resolve dJ ~ 1e-4.
#ifdef TSAT_F32
typedef float cfg_real;
#else
typedef double cfg_real;
#endif
typedef double acc_t;
constexpr int RPU = 16 / (int)sizeof(cfg_real);   // reals per 16-byte copy unit: 2 (double) or 4 (float)
// Storage / arithmetic type of the LINEARISATION: the Jacobian lanes and the knot records they leave for the Riccati recursion.
// -DTSAT_JAC32 (options.precision = 32, the mixed-precision builds): float — the discrete Jacobians are the bulk of the flops
// (nine tangent passes per knot) and of the on-chip / workspace bytes (72 | 84 values per knot and iteration), and an error of 1e-7
// in [A|B] only perturbs the search direction; everything a line-search decision rests on — the roll-out state, the feedback,
// the costs, the cost-to-go recursion, the gains, the multipliers, the field tables — stays double (J ~ 1e4 has to resolve
// dJ ~ 1e-4, and a budget-limited solve follows the fp64 iteration path only if its accept / reject decisions do).
#ifdef TSAT_JAC32
typedef float jac_t;
#else
typedef cfg_real jac_t;
#endif
constexpr int RPUJ = 16 / (int)sizeof(jac_t);     // record values per 16-byte copy unit

// The fp32 build comes in three LDS budgets, TSAT_OCC = wavefronts per SIMD it is laid out for (the compiler derives the
// register budget from the LDS-limited occupancy): 2 -> <= 20 480 B, 3 -> <= 13 653 B, 4 -> <= 10 240 B per wavefront.
#if defined(TSAT_F32) && !defined(TSAT_OCC)
#define TSAT_OCC 2
#endif

constexpr int WAVE = 64;
#if defined(TSAT_F32) && TSAT_OCC == 4
constexpr int CK = 16;    // knots per forward-sweep LDS chunk
#else
constexpr int CK = 32;
#endif
constexpr int PSTRIDE = 64;
// line-search candidates whose rollouts are kept in HBM. A deeper winner is re-rolled on its own (a second sweep)
// instead of every sweep streaming all max_linesearch candidate trajectories to memory. The launch ends with its
// slowest wavefront, so re-rolls must stay rare: on the reference Monte-Carlo workload the accepted step is
// alpha = 2^-j with j <= 5 in 99.8 % and j <= 11 in 99.96 % of the iterations.
constexpr int NSTORE = 12;
constexpr int N_FEW = 4;    // roll-outs a sweep keeps while the trajectory's line searches end early (solve_trajectory)
// per-trajectory parameter record (reals)
enum { P_X0 = 0, P_XF = 7, P_QD = 14, P_QFD = 21, P_RD = 28, P_ULO = 31, P_UHI = 34, P_J = 37, P_JI = 46,
       P_TAU0 = 55, P_DTAU = 56, P_DT = 57,
       // slots 58..60: P_QATT (tracking kernel). 61, 62: low-order parts of the table clock — the field row of knot k is
       // floor(fma(k + c, dtau, tau0)) in double whatever the storage type, so a float record carries tau0 and dtau as
       // (hi, lo) pairs; zero in a double record
       P_TAU0L = 61, P_DTAUL = 62 };
// Jacobian record of one knot (reals), left in LDS by the Jacobian lanes for the Riccati lanes (one-trajectory builds) or sent
// through the wavefront's workspace and the record ring (packed builds, tsat_packed.hpp) — one layout for both:
//   full state        F = [A|B] 7 x 10 column-major, column stride 7 (70), lx (7), lu (3), luu (3), pad (1)            = 84 reals
//   error coordinates F^ = [A^|B^] 6 x 9, column stride 6 (54; 48-byte columns: 16-byte LDS reads), G'QG upper triangle (6),
//                     lx^ (6), lu (3), luu (3)                                                                        = 72 reals
// Both are whole numbers of 16-byte units in either precision.
constexpr int FS = 7;
template <int ES> struct PkRec {
  static constexpr int FSR = ES ? 6 : FS;
  static constexpr int QQ = 54;
  static constexpr int LX = ES ? 60 : 70, LU = ES ? 66 : 77, LUU = ES ? 69 : 80;
  static constexpr int RECS = ES ? 72 : 84;
  // packed builds: one ring slot = the records of ONE knot of a pass's four trajectories + one 16-byte pad unit: the Jacobian lanes
  // (lane = trajectory, knot) write their records side by side, and four whole records are a multiple of 256 bytes — every knot's
  // lane would hit the same LDS banks. With the pad the slots start 16 bytes (mod 256) apart.
  static constexpr int SLOT = (64 / 16) * RECS + RPUJ;
};
// the tracking kernel (tvlqr_trajectory) reduces a full-state record in place: error-state blocks with the full-state strides
struct TvRec {
  static constexpr int FSR = FS, LX = 70, LU = 77, LUU = 80, QQ = 83, RECS = 89;
};
constexpr int TV_CHB = 52;                     // knots per chunk of the tracking kernel's gain recursion (wide build only)
template <int ES> struct BwdCfg {
  static constexpr int NH = ES ? 6 : 7;        // dimension of the state difference the gains act on
  static constexpr int RECS = PkRec<ES>::RECS; // reals per knot record
  // knots per backward chunk = Jacobian lanes per pass: as many as the LDS budget of the build holds, 64 at most
#if defined(TSAT_F32)
  // fp32 build: 4-byte records plus the 512-byte double reduction scratch inside the budget of TSAT_OCC waves per SIMD
  static constexpr int CHB = (TSAT_OCC == 2) ? (ES ? 63 : 54) : (TSAT_OCC == 3) ? (ES ? 39 : 34) : (ES ? 27 : 23);
#elif defined(TSAT_JAC32) && defined(TSAT_DENSE)
  // mixed-precision builds at two wavefronts per SIMD: float records, twice the knots of the double build in the same 20 KB
  static constexpr int CHB = ES ? 59 : 50;
#elif defined(TSAT_JAC32)
  static constexpr int CHB = 64;
#elif defined(TSAT_DENSE)
  // "dense" build of the solve kernel (tsat_kernels_dense.hip) for batches of more than one wave per SIMD: 8 waves x
  // <= 20 KB per CU and <= 256 registers, so that two trajectories share a SIMD (1.45x fp64 issue); the price is
  // 25 / 29-knot Jacobian chunks and 88 spilled registers, 11 % per wave
  static constexpr int CHB = ES ? 29 : 25;
#else
  static constexpr int CHB = ES ? 64 : 55;     // 4 waves x <= 40.6 KB fit one CU
#endif
};
// forward-sweep chunk arrays
constexpr int KDW = 24, XUW = 10, LMW = 6, BSW = 9;
// Per-trajectory strides of the HBM arrays, in reals, rounded up to whole 16-byte units so that every trajectory's slab
// (and with it every LDS copy unit) starts 16-byte aligned. For double the rounding is the identity (80 / 192 / 48-byte
// records); a float slab of an odd number of 40- or 24-byte records gets up to 12 bytes of padding.
template <typename real> constexpr size_t pad16(size_t n) { return (n + (16 / sizeof(real)) - 1) / (16 / sizeof(real)) * (16 / sizeof(real)); }
template <typename real> constexpr size_t xu_stride(int NS) { return pad16<real>((size_t)NS * XUW); }
template <typename real> constexpr size_t kd_stride(int NS) { return (size_t)(NS - 1) * KDW; }
template <typename real> constexpr size_t lam_stride(int NS) { return pad16<real>((size_t)(NS - 1) * LMW); }
template <typename real> constexpr size_t u0_stride(int NS) { return (size_t)(NS - 1) * 3; }

template <typename real>
struct KArgs {
  int T, N, n_tab, max_ls;   // max_ls: candidate slots reserved per trajectory (<= NSTORE)
  tsat_options opt;
  const real* P;      // [T][PSTRIDE]
  const real* BT;     // [n_btab][n_tab][4]
  const int* bidx;    // [T]
  const int* nk;      // [T] knots of each trajectory (2 <= nk[t] <= N) or null: all N  (ragged batches)
  const real* U0;     // [T][N-1][3]
  real* XU;           // [T][N][10]      nominal knot records x(7),u(3)
  real* KD;           // [T][N-1][24]    K (3x7 row-major), d(3)
  real* LAM;          // [T][N-1][6]     control-box multipliers [upper(3), lower(3)]
  real* CAND;         // [T][max_ls][N][10] stored line-search candidates
  tsat_stats* stats;  // [T]
  double* trace;      // [T][trace_rows][8] or null
  int trace_rows;
  real* JW;           // packed builds: Jacobian records of the backward chunk in flight, [wavefront][4][16][84] (tsat_packed.hpp)
  // Endgame of a packed launch (tsat_packed.hpp, "suspension"): once at most `suspend_at` trajectories of the batch are still
  // iterating, every wavefront parks its live ones — id and Resume record appended to the lists below — and leaves; a second
  // launch gives each parked trajectory a wavefront of its own in the one-trajectory mapping (tsat_resume_kernel_*).
  int pk_few = 0;               // packed builds: roll-outs kept per sweep while line searches end early (0: PK_FEW; tuning)
  int suspend_at = 0;           // 0: never
  int* live = nullptr;          // [1] trajectories that have not finished (set to T before the launch)
  int* susp_n = nullptr;        // [1] parked so far
  int* susp_ids = nullptr;      // [T]
  void* susp_state = nullptr;   // [T] Resume<real> records
};

// reals of a.JW per group of four trajectories (4 trajectories x 16 knots x 84-real records): host allocation and kernels agree on it
constexpr int TSAT_JW_REALS_PER_4 = 16 * (4 * 84 + 4);      // (16 ring slots: four records + a pad unit each)

// per-trajectory pointers handed (by value) to the phase functions
template <typename real>
struct TPtrs {
  TSAT_GLOBAL real* XU;          // [N][10]
  TSAT_GLOBAL real* KD;          // [N-1][24]
  TSAT_GLOBAL real* LAM;         // [N-1][6]
  TSAT_GLOBAL real* CAND;        // [max_ls][N][10]
  const TSAT_GLOBAL real* bt;    // [n_tab][4]
  // One-trajectory builds: the nominal trajectory is ADOPTED BY POINTER, not copied. The trajectory owns 1 + max_ls slabs of
  // [N][10] records — slab 0 is its slot of a.XU (XU0), slabs 1 .. max_ls its candidate slots (CAND) — of which `cur` holds the
  // nominal one (XU points at it) and the others, in order, take the stored candidates of a sweep (cand_slab). solve_trajectory
  // copies the final nominal back into slab 0 once, at the end. The packed builds copy per iteration and leave cur = 0.
  TSAT_GLOBAL real* XU0;
  int cur;
};
// slab that holds stored candidate c (c-th slab other than the nominal one), and its address
TSAT_DEV int cand_slab(int cur, int c) { return (c < cur) ? c : c + 1; }
template <typename real>
TSAT_DEV TSAT_GLOBAL real* slab_ptr(const TPtrs<real>& p, int N, int id) {
  return (id == 0) ? p.XU0 : p.CAND + (size_t)(id - 1) * (size_t)N * XUW;
}

// LDS carve-up (in reals)
constexpr int L_RED = 0;                 // 64 reduction scratch
constexpr int L_TR = L_RED + 64;         // trajectory constants: parameter record (64) + hJi(9) + hh + us (+pad) = 76
constexpr int TR_HJI = PSTRIDE, TR_HH = PSTRIDE + 9, TR_US = PSTRIDE + 10;
constexpr int L_NU = L_TR + 76;          // terminal multipliers nu(7) (+1)
constexpr int L_PC = L_NU + 8;           // diagnostic phase clocks (4)
constexpr int L_ST = L_PC + 4;           // S~ : 8 rows x 9
constexpr int L_WT = L_ST + 72;          // W~ : 10 cols x 9
constexpr int L_HXX = L_WT + 90;         // 7 x 7 (+1)
constexpr int L_HUX = L_HXX + 50;        // 3 x 8  (col 7 = Qu)
constexpr int L_HUU = L_HUX + 24;        // 3 x 3 (+1)
constexpr int L_KD = L_HUU + 10;         // 3 x 8  (col 7 = d)
constexpr int L_ZERO = L_KD + 24;        // a constant 0 (branch-free "no initial value" source)
constexpr int L_SINK = L_ZERO + 1;       // write target of lanes without a role in a step
constexpr int L_UNION = L_SINK + 1;      // 424 (16-byte aligned)
static_assert(L_UNION % 2 == 0, "phase buffers must stay 16-byte aligned");
constexpr int L_REC = L_UNION;           // BwdCfg::CHB x BwdCfg::RECS
// Forward-sweep chunk buffer (reals). The arrays are filled by global_load_lds_dwordx4 — 64 lanes x 16 bytes
// = 128 reals per instruction, lane-linear — so each is padded to a whole number of instructions. (The control-box multipliers
// are not staged: the roll-out does not evaluate costs, candidate_costs does, with lanes = knots.)
constexpr int GLDS = RPU * WAVE;                                       // reals per copy instruction
constexpr int BROW_UNITS = 4 / RPU;                                    // 16-byte units per 4-real field-table row: 2 | 1
constexpr int FB_KD = 0;                                               // CK x 24 = 6 instructions
constexpr int FB_XU = FB_KD + ((CK * KDW + GLDS - 1) / GLDS) * GLDS;   // CK x 10 -> 3 instructions
constexpr int FB_BA = FB_XU + ((CK * XUW + GLDS - 1) / GLDS) * GLDS;   // 3 CK stage rows: (b0, b1) [double] or the whole row (b0, b1, b2, pad) [float]
constexpr int FB_BB = FB_BA + ((CK * 3 * RPU + GLDS - 1) / GLDS) * GLDS; // 3 CK stage rows, (b2, pad) [double only]
constexpr int FB_SIZE = FB_BB + (BROW_UNITS == 2 ? ((CK * 3 * RPU + GLDS - 1) / GLDS) * GLDS : 0);
// candidate_costs: per-knot costs of CG candidates x KB knots (+2: the CG summing lanes read different banks)
constexpr int CG = 2, KB = 4 * WAVE, KBS = KB + 2;
#if defined(TSAT_DENSE) || (defined(TSAT_F32) && TSAT_OCC >= 3)
constexpr int FWD_NBUF = 1;   // 20 KB budget: one buffer; the second wavefront on the SIMD covers the copy latency
#else
constexpr int FWD_NBUF = 2;   // the next chunk is copied while this one is rolled out
#endif
constexpr int L_FWD = L_UNION;
constexpr int L_FWD_END = L_FWD + (FWD_NBUF * FB_SIZE > CG * KBS ? FWD_NBUF * FB_SIZE : CG * KBS);
constexpr int L_BWD_RECV = (BwdCfg<0>::CHB * BwdCfg<0>::RECS > BwdCfg<1>::CHB * BwdCfg<1>::RECS
                               ? BwdCfg<0>::CHB * BwdCfg<0>::RECS : BwdCfg<1>::CHB * BwdCfg<1>::RECS);     // record values of a chunk
constexpr int L_BWD_SOLVE = (L_BWD_RECV * (int)sizeof(jac_t) + (int)sizeof(cfg_real) - 1) / (int)sizeof(cfg_real);   // ... in reals
#if !defined(TSAT_DENSE) && !defined(TSAT_F32)
constexpr int L_BWD_END = L_REC + (L_BWD_SOLVE > TV_CHB * TvRec::RECS ? L_BWD_SOLVE : TV_CHB * TvRec::RECS);   // + the tracking kernel
#else
constexpr int L_BWD_END = L_REC + L_BWD_SOLVE;
#endif
#if defined(TSAT_PACKED)
// packed builds (tsat_packed.hpp): their own carve-up behind L_UNION, sized to the 20 480 B of two wavefronts per SIMD or to
// TSAT_PK_LDS_BYTES (40 960 B: the one-wavefront-per-SIMD units, tsat_kernels_packed4w / 8w / 16w.hip)
#ifndef TSAT_PK_LDS_BYTES
#define TSAT_PK_LDS_BYTES 20480
#endif
constexpr int LDS_REALS = (TSAT_PK_LDS_BYTES - ((sizeof(cfg_real) == 8) ? 0 : 64 * 8)) / (int)sizeof(cfg_real);
#else
constexpr int LDS_REALS = (L_FWD_END > L_BWD_END ? L_FWD_END : L_BWD_END);
#endif

// The wavefront's LDS block: a STATIC module-level __shared__ array. Declared at namespace scope so that every phase
// function addresses it as LDS (address space 3) at a link-time constant address — a generic pointer argument would
// turn ds_* into flat_* accesses, and a dynamic (extern) array makes every non-kernel function fetch its base address
// with an s_load inside the hot loops.
// the fp32 build appends 64 doubles: scratch of the wave reductions that are carried in double (acc_t). In the fp64 builds
// those reductions use the ordinary reduction scratch L_RED (same type), so their LDS footprint is unchanged.
constexpr int RED64_BYTES = (sizeof(cfg_real) == 8) ? 0 : 64 * 8;
constexpr int LDS_BYTES = LDS_REALS * (int)sizeof(cfg_real) + RED64_BYTES;
#ifndef TSAT_EMU
__shared__ __align__(16) unsigned char tsat_smem[LDS_BYTES];
#endif
template <typename real>
TSAT_DEV real* lds_base() {
  static_assert(sizeof(real) == sizeof(cfg_real), "the LDS carve-up is counted in the translation unit's storage type");
#ifdef TSAT_EMU
  return reinterpret_cast<real*>(tsat_emu::lds());
#else
  return reinterpret_cast<real*>(tsat_smem);
#endif
}
TSAT_DEV acc_t* red64() {
#ifdef TSAT_EMU
  unsigned char* b = reinterpret_cast<unsigned char*>(tsat_emu::lds());
#else
  unsigned char* b = tsat_smem;
#endif
  return reinterpret_cast<acc_t*>(b + (RED64_BYTES ? LDS_REALS * (int)sizeof(cfg_real) : L_RED * (int)sizeof(cfg_real)));
}

// S~ = [S; s'] of the row-oriented Riccati recursion between two chunks, in double whatever the storage type: row stride 9 from
// L_ST on — in the double builds exactly where terminal_cost_to_go leaves it; the float builds widen it in place over the scratch
// of the element-oriented recursion that follows L_ST (unused by the row-oriented one)
TSAT_DEV double* st64() {
#ifdef TSAT_EMU
  unsigned char* b = reinterpret_cast<unsigned char*>(tsat_emu::lds());
#else
  unsigned char* b = tsat_smem;
#endif
  return reinterpret_cast<double*>(b + L_ST * (int)sizeof(cfg_real));
}

// --------------------------------------------------------------------------------------------------
// small math helpers
// --------------------------------------------------------------------------------------------------
template <typename real> TSAT_DEV real rsqrt_(real s);
#ifdef TSAT_EMU
template <> TSAT_DEV double rsqrt_<double>(double s) { return 1.0 / std::sqrt(s); }
template <> TSAT_DEV float rsqrt_<float>(float s) { return 1.0f / std::sqrt(s); }
TSAT_DEV double fabs_(double a) { return std::fabs(a); }
TSAT_DEV double fmax_(double a, double b) { return a > b ? a : b; }   // NaN in b ignored, like v_max
TSAT_DEV double fmaxabs_(double a, double b) { const double c = std::fabs(b); return a > c ? a : c; }
TSAT_DEV double floor_(double a) { return std::floor(a); }
TSAT_DEV double fma_(double a, double b, double c) { return std::fma(a, b, c); }
TSAT_DEV double rcp_(double a) { return 1.0 / a; }
TSAT_DEV double sqrt_(double a) { return std::sqrt(a); }
TSAT_DEV double sin_(double a) { return std::sin(a); }
TSAT_DEV double cos_(double a) { return std::cos(a); }
TSAT_DEV double acos_(double a) { return std::acos(a); }
TSAT_DEV double asin_(double a) { return std::asin(a); }
TSAT_DEV double atan2_(double a, double b) { return std::atan2(a, b); }
TSAT_DEV double fmod_(double a, double b) { return std::fmod(a, b); }
TSAT_DEV double log_(double a) { return std::log(a); }
TSAT_DEV float fabs_(float a) { return std::fabs(a); }
TSAT_DEV float fmax_(float a, float b) { return a > b ? a : b; }
TSAT_DEV float fmaxabs_(float a, float b) { const float c = std::fabs(b); return a > c ? a : c; }
TSAT_DEV float rcp_(float a) { return 1.0f / a; }
TSAT_DEV float fma_(float a, float b, float c) { return std::fma(a, b, c); }
#else
// v_rsq_f64 seed (rel. error <= 5.3e-8, profiles/r01/rsq_rcp_accuracy.txt) + ONE third-order step:
// y (1 + e/2 + 3e^2/8), e = 1 - s y^2  ->  error O(e^3) ~ 1e-22, i.e. rounding only; 5 instructions instead of the 8
// of two Newton steps
template <> TSAT_DEV double rsqrt_<double>(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  const double e = __builtin_fma(-(s * y), y, 1.0);
  const double p = __builtin_fma(0.375, e, 0.5) * e;
  return __builtin_fma(y, p, y);
}
template <> TSAT_DEV float rsqrt_<float>(float s) { return __builtin_amdgcn_rsqf(s); }
TSAT_DEV double fabs_(double a) { return __builtin_fabs(a); }
// one v_max_f64 each. (__builtin_fmax costs two: the compiler first canonicalises the operands with an extra
// v_max_f64 v,v,v; a NaN operand is ignored by the instruction either way, which is what the callers rely on.)
TSAT_DEV double fmax_(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
TSAT_DEV double fmaxabs_(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
TSAT_DEV double floor_(double a) { return __builtin_floor(a); }
TSAT_DEV double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// v_rcp_f64 seed (rel. error <= 4.7e-8) + one third-order step y (1 + e + e^2), e = 1 - a y
TSAT_DEV double rcp_(double a) {
  const double y = __builtin_amdgcn_rcp(a);
  const double e = __builtin_fma(-a, y, 1.0);
  const double p = __builtin_fma(e, e, e);
  return __builtin_fma(y, p, y);
}
TSAT_DEV double sqrt_(double a) { return __builtin_sqrt(a); }
TSAT_DEV double sin_(double a) { return ::sin(a); }     // HIP device math (ocml); used by the tracking kernel only
TSAT_DEV double cos_(double a) { return ::cos(a); }
TSAT_DEV double acos_(double a) { return ::acos(a); }
TSAT_DEV double asin_(double a) { return ::asin(a); }
TSAT_DEV double atan2_(double a, double b) { return ::atan2(a, b); }
TSAT_DEV double fmod_(double a, double b) { return ::fmod(a, b); }
TSAT_DEV double log_(double a) { return ::log(a); }
TSAT_DEV float fabs_(float a) { return __builtin_fabsf(a); }
TSAT_DEV float fmax_(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
TSAT_DEV float fmaxabs_(float a, float b) { float r; asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
TSAT_DEV float rcp_(float a) { return __builtin_amdgcn_rcpf(a); }   // 1 ulp
TSAT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#endif
// a b - c d with the rounding spelled out (one product rounded, then fused): a compiler is free to contract either product of
// the plain expression, and the builds of the solve kernel must agree to the last bit
template <typename real> TSAT_DEV real dmm_(real a, real b, real c, real d) { return fma_(a, b, -(c * d)); }
// a0 b0 + a1 b1 + a2 b2, left to right, likewise
template <typename real> TSAT_DEV real dot3_(real a0, real b0, real a1, real b1, real a2, real b2) { return fma_(a2, b2, fma_(a1, b1, a0 * b0)); }

// phase timing for the diagnostic build (-DTSAT_PROFILE): shader-clock stamps accumulated per phase and written to
// the trace buffer's row 0 of each trajectory; the production build compiles all of it away.
#if defined(TSAT_PROFILE) && !defined(TSAT_EMU)
TSAT_DEV unsigned long long tick_() { return __builtin_amdgcn_s_memtime(); }
#else
TSAT_DEV unsigned long long tick_() { return 0ull; }
#endif

// a[i] for a register-resident array and a run-time i (select chain: keeps `a` out of scratch memory)
template <typename real>
TSAT_DEV real sel7(const real a[7], int i) {
  real v = a[0];
  for (int m = 1; m < 7; ++m) v = (i == m) ? a[m] : v;
  return v;
}

template <typename real> TSAT_DEV real inf_() { return (real)__builtin_huge_val(); }

// Store of a branch-free lane-role step: lanes without a role in the step aim at the sink word L_SINK (never read). On the
// GPU that is one ds_write for the whole wave — lanes hitting the same address in one instruction are well defined — while
// host threads storing to one word concurrently are a data race, so the emulator drops the sink stores instead.
template <typename real>
TSAT_DEV void role_store(real* lds, int off, int sink, real v) {
#ifdef TSAT_EMU
  if (off == sink) return;
#else
  (void)sink;
#endif
  lds[off] = v;
}

// wave collectives through LDS scratch; identical butterfly order in the GPU and emulated builds
template <typename real>
TSAT_DEV real wave_sum(real v, real* red) {
  const int lane = TSAT_LANE();
  for (int s = 1; s < WAVE; s <<= 1) {
    red[lane] = v;
    TSAT_SYNC_LDS();
    v = v + red[lane ^ s];
    TSAT_SYNC_LDS();
  }
  return v;
}
template <typename real>
TSAT_DEV real wave_max(real v, real* red) {
  const int lane = TSAT_LANE();
  for (int s = 1; s < WAVE; s <<= 1) {
    red[lane] = v;
    TSAT_SYNC_LDS();
    real o = red[lane ^ s];
    v = (o > v) ? o : v;
    TSAT_SYNC_LDS();
  }
  return v;
}
template <typename real>
TSAT_DEV real wave_bcast(real v, int src, real* red) {
  const int lane = TSAT_LANE();
  if (lane == src) red[0] = v;
  TSAT_SYNC_LDS();
  real r = red[0];
  TSAT_SYNC_LDS();
  return r;
}
// smallest lane index whose flag is set, or WAVE
template <typename real>
TSAT_DEV int wave_first(bool flag, real* red) {
  real v = flag ? (real)TSAT_LANE() : (real)WAVE;
  const int lane = TSAT_LANE();
  for (int s = 1; s < WAVE; s <<= 1) {
    red[lane] = v;
    TSAT_SYNC_LDS();
    real o = red[lane ^ s];
    v = (o < v) ? o : v;
    TSAT_SYNC_LDS();
  }
  return (int)v;
}

// --------------------------------------------------------------------------------------------------
// per-trajectory constants (wave-uniform)
// --------------------------------------------------------------------------------------------------
template <typename real>
struct Traj {
  real xf[7], Qd[7], Qfd[7], Rd[3], ulo[3], uhi[3];
  real J[9];    // row-major J(r,c) = J[3r+c]
  real hJi[9];  // h * inv(J), row-major
  real h, hh, us;
  real usj;     // us * hJi[0]: control scale of the isotropic-inertia variants (DIAGJ == 2), where h inv(J) = (h / j) I is folded into it
  double tau0, dtau;
  int N, n_tab;
  const TSAT_GLOBAL real* bt;  // [n_tab][4]
};

// The constants live in the wave's LDS block (written once by stage_traj); every phase function re-reads the
// ones it uses, so they sit in VGPRs local to that phase instead of in SGPRs spilled across the whole kernel.
template <typename real>
TSAT_DEV Traj<real> load_traj_at(const real* t, int N, int n_tab, const TSAT_GLOBAL real* bt) {
  Traj<real> tr;
  for (int i = 0; i < 7; ++i) { tr.xf[i] = t[P_XF + i]; tr.Qd[i] = t[P_QD + i]; tr.Qfd[i] = t[P_QFD + i]; }
  for (int i = 0; i < 3; ++i) { tr.Rd[i] = t[P_RD + i]; tr.ulo[i] = t[P_ULO + i]; tr.uhi[i] = t[P_UHI + i]; }
  for (int i = 0; i < 9; ++i) { tr.J[i] = t[P_J + i]; tr.hJi[i] = t[TR_HJI + i]; }
  tr.h = t[P_DT]; tr.hh = t[TR_HH]; tr.us = t[TR_US]; tr.usj = tr.us * tr.hJi[0];
  tr.tau0 = (double)t[P_TAU0] + (double)t[P_TAU0L]; tr.dtau = (double)t[P_DTAU] + (double)t[P_DTAUL];
  tr.N = N; tr.n_tab = n_tab; tr.bt = bt;
  return tr;
}
template <typename real>
TSAT_DEV Traj<real> load_traj(int N, int n_tab, const TSAT_GLOBAL real* bt) {
  return load_traj_at<real>(lds_base<real>() + L_TR, N, n_tab, bt);
}
// copy the parameter record into LDS and append the derived constants (all lanes; wave-uniform values)
template <typename real>
TSAT_DEV void stage_traj(const TSAT_GLOBAL real* P, real u_scale) {
  real* t = lds_base<real>() + L_TR;
  const int lane = TSAT_LANE();
  t[lane] = P[lane];                       // PSTRIDE == WAVE
  const real h = P[P_DT];
  if (lane < 9) t[TR_HJI + lane] = h * P[P_JI + lane];
  if (lane == 9) t[TR_HH] = (real)0.5 * h;
  if (lane == 10) t[TR_US] = u_scale;
  if (lane < 8) lds_base<real>()[L_NU + lane] = 0;
  if (lane < 4) lds_base<real>()[L_PC + lane] = 0;
}

template <typename real>
TSAT_DEV int brow_index(const Traj<real>& tr, int k, double c) {
  // floor(fma(k + c, dtau, tau0)) clamped — the reference's B_ECI[floor(Int, t*N + 1), :] (src/DerivFunction.jl:28)
  double r = floor_(fma_((double)k + c, tr.dtau, tr.tau0));
  int i = (r >= 0.0) ? (r > (double)(tr.n_tab - 1) ? tr.n_tab - 1 : (int)r) : 0;
  return i;
}

// h * f(x,u;b): src/DerivFunction.jl:4-44 on the 7-state. `us` = u * u_scale (:37) — times h / j in the isotropic-inertia variant
// (DIAGJ == 2: inv(J) is a multiple of the identity and w x Jw vanishes, so wdot = (h / j) (us x B_B) and the factor is folded
// into the control once per knot, control_scale(), instead of into every stage's three components). Intermediates needed by the
// tangent pass are returned in `sb`.
template <typename real, int DIAGJ> TSAT_DEV real control_scale(const Traj<real>& tr) { return (DIAGJ == 2) ? tr.usj : tr.us; }
template <typename real>
struct StageBase {
  real w[3], qh[4], rn, c[3], BB[3], Jw[3];
};

template <typename real, int DIAGJ>
TSAT_DEV void dyn_h(const Traj<real>& tr, const real x[7], const real us[3], const real b[3], real k[7],
                    StageBase<real>& sb) {
  const real w0 = x[0], w1 = x[1], w2 = x[2];
  const real s = x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6];
  const real rn = rsqrt_<real>(s);
  const real q0 = x[3] * rn, q1 = x[4] * rn, q2 = x[5] * rn, q3 = x[6] * rn;
  // qdot = 0.5 qmult(q,[0;w])  (:24)
  k[3] = -tr.hh * (q1 * w0 + q2 * w1 + q3 * w2);
  k[4] = tr.hh * (q0 * w0 + (q2 * w2 - q3 * w1));
  k[5] = tr.hh * (q0 * w1 + (q3 * w0 - q1 * w2));
  k[6] = tr.hh * (q0 * w2 + (q1 * w1 - q2 * w0));
  // B_B = qrot(q,b) = b + 2 v x (v x b + s b)  (:28,:50-52)
  const real c0 = (q2 * b[2] - q3 * b[1]) + q0 * b[0];
  const real c1 = (q3 * b[0] - q1 * b[2]) + q0 * b[1];
  const real c2 = (q1 * b[1] - q2 * b[0]) + q0 * b[2];
  const real B0 = b[0] + 2 * (q2 * c2 - q3 * c1);
  const real B1 = b[1] + 2 * (q3 * c0 - q1 * c2);
  const real B2 = b[2] + 2 * (q1 * c1 - q2 * c0);
  // tau_c = cross(u*1e-2, B_B)  (:37)
  const real t0 = us[1] * B2 - us[2] * B1;
  const real t1 = us[2] * B0 - us[0] * B2;
  const real t2 = us[0] * B1 - us[1] * B0;
  // wdot = inv(J) (tau_c - w x Jw)  (:41)
  // every inertia preset of the reference is diagonal (src/input_parameters.jl:29-51): DIAGJ = 1 drops the zero
  // terms; the 1U and 1P presets are isotropic (J = j I), where w x Jw vanishes identically: DIAGJ = 2 drops it too
  const real Jw0 = DIAGJ ? tr.J[0] * w0 : tr.J[0] * w0 + tr.J[1] * w1 + tr.J[2] * w2;
  const real Jw1 = DIAGJ ? tr.J[4] * w1 : tr.J[3] * w0 + tr.J[4] * w1 + tr.J[5] * w2;
  const real Jw2 = DIAGJ ? tr.J[8] * w2 : tr.J[6] * w0 + tr.J[7] * w1 + tr.J[8] * w2;
  const real r0 = (DIAGJ == 2) ? t0 : t0 - (w1 * Jw2 - w2 * Jw1);
  const real r1 = (DIAGJ == 2) ? t1 : t1 - (w2 * Jw0 - w0 * Jw2);
  const real r2 = (DIAGJ == 2) ? t2 : t2 - (w0 * Jw1 - w1 * Jw0);
  k[0] = (DIAGJ == 2) ? r0 : DIAGJ ? tr.hJi[0] * r0 : tr.hJi[0] * r0 + tr.hJi[1] * r1 + tr.hJi[2] * r2;
  k[1] = (DIAGJ == 2) ? r1 : DIAGJ ? tr.hJi[4] * r1 : tr.hJi[3] * r0 + tr.hJi[4] * r1 + tr.hJi[5] * r2;
  k[2] = (DIAGJ == 2) ? r2 : DIAGJ ? tr.hJi[8] * r2 : tr.hJi[6] * r0 + tr.hJi[7] * r1 + tr.hJi[8] * r2;
  sb.w[0] = w0; sb.w[1] = w1; sb.w[2] = w2;
  sb.qh[0] = q0; sb.qh[1] = q1; sb.qh[2] = q2; sb.qh[3] = q3;
  sb.rn = rn;
  sb.c[0] = c0; sb.c[1] = c1; sb.c[2] = c2;
  sb.BB[0] = B0; sb.BB[1] = B1; sb.BB[2] = B2;
  sb.Jw[0] = Jw0; sb.Jw[1] = Jw1; sb.Jw[2] = Jw2;
}

// directional derivative of h*f at a stage (applied to a tangent instead of forming F):
// dx = [dw; dq] tangent of the stage state, dus = (control scale) * du.
// ZW / ZQ / ZU: the rate part, the quaternion part, the control part of the tangent is identically zero (the FIRST stage of a
// seeded column: a unit rate has no attitude or control component, and so on) — the terms it would feed are left out instead of
// being multiplied by zeros; what remains is evaluated exactly as in the general case, so a column has the same value either way
// (up to the sign of a zero).
template <typename real, int DIAGJ, bool ZW = false, bool ZQ = false, bool ZU = false>
TSAT_DEV void dyn_h_jvp(const Traj<real>& tr, const StageBase<real>& sb, const real us[3], const real b[3],
                        const real dx[7], const real dus[3], real dk[7]) {
  const real q0 = sb.qh[0], q1 = sb.qh[1], q2 = sb.qh[2], q3 = sb.qh[3];
  const real w0 = sb.w[0], w1 = sb.w[1], w2 = sb.w[2];
  real dw0 = 0, dw1 = 0, dw2 = 0;
  if (!ZW) { dw0 = dx[0]; dw1 = dx[1]; dw2 = dx[2]; }
  real e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  if (!ZQ) {
    // d qhat = rn (dq - qhat (qhat . dq))   (normalisation inside f, src/DerivFunction.jl:5)
    const real t = q0 * dx[3] + q1 * dx[4] + q2 * dx[5] + q3 * dx[6];
    e0 = sb.rn * (dx[3] - q0 * t);
    e1 = sb.rn * (dx[4] - q1 * t);
    e2 = sb.rn * (dx[5] - q2 * t);
    e3 = sb.rn * (dx[6] - q3 * t);
  }
  if (!ZQ && !ZW) {
    dk[3] = -tr.hh * ((e1 * w0 + e2 * w1 + e3 * w2) + (q1 * dw0 + q2 * dw1 + q3 * dw2));
    dk[4] = tr.hh * ((e0 * w0 + (e2 * w2 - e3 * w1)) + (q0 * dw0 + (q2 * dw2 - q3 * dw1)));
    dk[5] = tr.hh * ((e0 * w1 + (e3 * w0 - e1 * w2)) + (q0 * dw1 + (q3 * dw0 - q1 * dw2)));
    dk[6] = tr.hh * ((e0 * w2 + (e1 * w1 - e2 * w0)) + (q0 * dw2 + (q1 * dw1 - q2 * dw0)));
  } else if (!ZQ) {
    dk[3] = -tr.hh * (e1 * w0 + e2 * w1 + e3 * w2);
    dk[4] = tr.hh * (e0 * w0 + (e2 * w2 - e3 * w1));
    dk[5] = tr.hh * (e0 * w1 + (e3 * w0 - e1 * w2));
    dk[6] = tr.hh * (e0 * w2 + (e1 * w1 - e2 * w0));
  } else if (!ZW) {
    dk[3] = -tr.hh * (q1 * dw0 + q2 * dw1 + q3 * dw2);
    dk[4] = tr.hh * (q0 * dw0 + (q2 * dw2 - q3 * dw1));
    dk[5] = tr.hh * (q0 * dw1 + (q3 * dw0 - q1 * dw2));
    dk[6] = tr.hh * (q0 * dw2 + (q1 * dw1 - q2 * dw0));
  } else {
    dk[3] = 0; dk[4] = 0; dk[5] = 0; dk[6] = 0;
  }
  // dtau = dus x BB + us x dBB, dBB = 2 (dv x c + v x dc), dc = dv x b + ds b
  real t0 = 0, t1 = 0, t2 = 0;
  if (!ZQ) {
    const real dc0 = (e2 * b[2] - e3 * b[1]) + e0 * b[0];
    const real dc1 = (e3 * b[0] - e1 * b[2]) + e0 * b[1];
    const real dc2 = (e1 * b[1] - e2 * b[0]) + e0 * b[2];
    const real dB0 = 2 * ((e2 * sb.c[2] - e3 * sb.c[1]) + (q2 * dc2 - q3 * dc1));
    const real dB1 = 2 * ((e3 * sb.c[0] - e1 * sb.c[2]) + (q3 * dc0 - q1 * dc2));
    const real dB2 = 2 * ((e1 * sb.c[1] - e2 * sb.c[0]) + (q1 * dc1 - q2 * dc0));
    if (!ZU) {
      t0 = (dus[1] * sb.BB[2] - dus[2] * sb.BB[1]) + (us[1] * dB2 - us[2] * dB1);
      t1 = (dus[2] * sb.BB[0] - dus[0] * sb.BB[2]) + (us[2] * dB0 - us[0] * dB2);
      t2 = (dus[0] * sb.BB[1] - dus[1] * sb.BB[0]) + (us[0] * dB1 - us[1] * dB0);
    } else {
      t0 = us[1] * dB2 - us[2] * dB1;
      t1 = us[2] * dB0 - us[0] * dB2;
      t2 = us[0] * dB1 - us[1] * dB0;
    }
  } else if (!ZU) {
    t0 = dus[1] * sb.BB[2] - dus[2] * sb.BB[1];
    t1 = dus[2] * sb.BB[0] - dus[0] * sb.BB[2];
    t2 = dus[0] * sb.BB[1] - dus[1] * sb.BB[0];
  }
  // d(w x Jw) = dw x Jw + w x J dw
  real r0 = t0, r1 = t1, r2 = t2;
  if (DIAGJ != 2 && !ZW) {
    const real dJ0 = DIAGJ ? tr.J[0] * dw0 : tr.J[0] * dw0 + tr.J[1] * dw1 + tr.J[2] * dw2;
    const real dJ1 = DIAGJ ? tr.J[4] * dw1 : tr.J[3] * dw0 + tr.J[4] * dw1 + tr.J[5] * dw2;
    const real dJ2 = DIAGJ ? tr.J[8] * dw2 : tr.J[6] * dw0 + tr.J[7] * dw1 + tr.J[8] * dw2;
    r0 = t0 - ((dw1 * sb.Jw[2] - dw2 * sb.Jw[1]) + (w1 * dJ2 - w2 * dJ1));
    r1 = t1 - ((dw2 * sb.Jw[0] - dw0 * sb.Jw[2]) + (w2 * dJ0 - w0 * dJ2));
    r2 = t2 - ((dw0 * sb.Jw[1] - dw1 * sb.Jw[0]) + (w0 * dJ1 - w1 * dJ0));
  }
  dk[0] = (DIAGJ == 2) ? r0 : DIAGJ ? tr.hJi[0] * r0 : tr.hJi[0] * r0 + tr.hJi[1] * r1 + tr.hJi[2] * r2;
  dk[1] = (DIAGJ == 2) ? r1 : DIAGJ ? tr.hJi[4] * r1 : tr.hJi[3] * r0 + tr.hJi[4] * r1 + tr.hJi[5] * r2;
  dk[2] = (DIAGJ == 2) ? r2 : DIAGJ ? tr.hJi[8] * r2 : tr.hJi[6] * r0 + tr.hJi[7] * r1 + tr.hJi[8] * r2;
}

// one RK step (rk3: src/attitude_controller.jl:178-187; rk4: :122-132); b0/b1/b2 = rows at tau, tau+dtau/2, tau+dtau
template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_DEV void rk_step(const Traj<real>& tr, const real x[7], const real u[3], const real b0[3], const real b1[3],
                      const real b2[3], real xn[7]) {
  const real cs = control_scale<real, DIAGJ>(tr);
  const real us[3] = {u[0] * cs, u[1] * cs, u[2] * cs};
  real k1[7], k2[7], k3[7], t[7];
  StageBase<real> sb;
  dyn_h<real, DIAGJ>(tr, x, us, b0, k1, sb);
  for (int i = 0; i < 7; ++i) t[i] = x[i] + (real)0.5 * k1[i];
  dyn_h<real, DIAGJ>(tr, t, us, b1, k2, sb);
  if (INTEG == 3) {
    for (int i = 0; i < 7; ++i) t[i] = x[i] - k1[i] + 2 * k2[i];
    dyn_h<real, DIAGJ>(tr, t, us, b2, k3, sb);
    for (int i = 0; i < 7; ++i) xn[i] = x[i] + (k1[i] + 4 * k2[i] + k3[i]) * (real)(1.0 / 6.0);
  } else {
    real k4[7];
    for (int i = 0; i < 7; ++i) t[i] = x[i] + (real)0.5 * k2[i];
    dyn_h<real, DIAGJ>(tr, t, us, b1, k3, sb);
    for (int i = 0; i < 7; ++i) t[i] = x[i] + k3[i];
    dyn_h<real, DIAGJ>(tr, t, us, b2, k4, sb);
    for (int i = 0; i < 7; ++i) xn[i] = x[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) * (real)(1.0 / 6.0);
  }
}

// discrete Jacobians of one RK step, column by column, written to LDS record `F` (column stride FS).
// Equivalent of ForwardDiff.jacobian! over the discretised dynamics (src/attitude_controller.jl:95-119): the primal stages
// once (rk_primal), then one tangent pass through the stages per column (rk_tangent) — each column is its own pass, so
// splitting the columns over lanes (the packed build's Jacobian lanes, tsat_packed.hpp) leaves every column's arithmetic
// untouched.
template <typename real> struct RkStages { StageBase<real> s1, s2, s3, s4; real us[3]; };
template <int S> struct SeedTag { static constexpr int value = S; };
template <typename real, int INTEG, int DIAGJ>
TSAT_DEV void rk_primal(const Traj<real>& tr, const real x[7], const real u[3], const real b0[3], const real b1[3],
                        const real b2[3], RkStages<real>& st) {
  for (int a = 0; a < 3; ++a) st.us[a] = u[a] * control_scale<real, DIAGJ>(tr);
  real k1[7], k2[7], k3[7], t[7];
  dyn_h<real, DIAGJ>(tr, x, st.us, b0, k1, st.s1);
  for (int i = 0; i < 7; ++i) t[i] = x[i] + (real)0.5 * k1[i];
  dyn_h<real, DIAGJ>(tr, t, st.us, b1, k2, st.s2);
  if (INTEG == 3) {
    for (int i = 0; i < 7; ++i) t[i] = x[i] - k1[i] + 2 * k2[i];
    dyn_h<real, DIAGJ>(tr, t, st.us, b2, k3, st.s3);
  } else {
    for (int i = 0; i < 7; ++i) t[i] = x[i] + (real)0.5 * k2[i];
    dyn_h<real, DIAGJ>(tr, t, st.us, b1, k3, st.s3);
    for (int i = 0; i < 7; ++i) t[i] = x[i] + k3[i];
    real k4[7];
    dyn_h<real, DIAGJ>(tr, t, st.us, b2, k4, st.s4);
  }
}
// directional derivative of the RK step along (ex, du): out = ex + (weighted sum of the stage tangents).
// SEED: what the seed (ex, du) is known to be — 0 anything; 1 a rate direction (no attitude, no control part); 2 an attitude
// direction (no rate, no control part); 3 a control direction (ex = 0). The first stage sees the seed itself and leaves the terms
// of its zero parts out; a control seed reaches the second stage still without an attitude part.
template <typename real, int INTEG, int DIAGJ, int SEED = 0>
TSAT_DEV void rk_tangent(const Traj<real>& tr, const RkStages<real>& st, const real b0[3], const real b1[3], const real b2[3],
                         const real ex[7], const real du[3], real out[7]) {
  real v1[7], v2[7], v3[7], arg[7];
  if (SEED == 1) dyn_h_jvp<real, DIAGJ, false, true, true>(tr, st.s1, st.us, b0, ex, du, v1);
  else if (SEED == 2) dyn_h_jvp<real, DIAGJ, true, false, true>(tr, st.s1, st.us, b0, ex, du, v1);
  else if (SEED == 3) dyn_h_jvp<real, DIAGJ, true, true, false>(tr, st.s1, st.us, b0, ex, du, v1);
  else dyn_h_jvp<real, DIAGJ>(tr, st.s1, st.us, b0, ex, du, v1);
  for (int i = 0; i < 7; ++i) arg[i] = ex[i] + (real)0.5 * v1[i];
  if (SEED == 3) dyn_h_jvp<real, DIAGJ, false, true, false>(tr, st.s2, st.us, b1, arg, du, v2);
  else dyn_h_jvp<real, DIAGJ>(tr, st.s2, st.us, b1, arg, du, v2);
  if (INTEG == 3) {
    for (int i = 0; i < 7; ++i) arg[i] = ex[i] - v1[i] + 2 * v2[i];
    dyn_h_jvp<real, DIAGJ>(tr, st.s3, st.us, b2, arg, du, v3);
    for (int i = 0; i < 7; ++i) out[i] = ex[i] + (v1[i] + 4 * v2[i] + v3[i]) * (real)(1.0 / 6.0);
  } else {
    real v4[7];
    for (int i = 0; i < 7; ++i) arg[i] = ex[i] + (real)0.5 * v2[i];
    dyn_h_jvp<real, DIAGJ>(tr, st.s3, st.us, b1, arg, du, v3);
    for (int i = 0; i < 7; ++i) arg[i] = ex[i] + v3[i];
    dyn_h_jvp<real, DIAGJ>(tr, st.s4, st.us, b2, arg, du, v4);
    for (int i = 0; i < 7; ++i) out[i] = ex[i] + (v1[i] + 2 * v2[i] + 2 * v3[i] + v4[i]) * (real)(1.0 / 6.0);
  }
}
// columns c_lo .. c_hi - 1 of [A|B] (7 x 10) in the full state: unit seeds
template <typename real, int INTEG, int DIAGJ, int ES, typename FP, int FSS = FS>
TSAT_DEV void rk_jacobian_cols(const Traj<real>& tr, const real x[7], const real u[3], const real b0[3],
                               const real b1[3], const real b2[3], FP F, int c_lo, int c_hi) {
  RkStages<real> st;
  rk_primal<real, INTEG, DIAGJ>(tr, x, u, b0, b1, b2, st);
  auto column = [&](auto seed, int c) {
    real ex[7], du[3], out[7];
    for (int i = 0; i < 7; ++i) ex[i] = (i == c) ? (real)1 : (real)0;
    for (int a = 0; a < 3; ++a) du[a] = (c == 7 + a) ? control_scale<real, DIAGJ>(tr) : (real)0;
    rk_tangent<real, INTEG, DIAGJ, decltype(seed)::value>(tr, st, b0, b1, b2, ex, du, out);
    for (int i = 0; i < 7; ++i) F[c * FSS + i] = out[i];
  };
  // one loop per kind of seed (rate, quaternion, control): each gets the tangent pass without the terms of its zero parts
  TSAT_NO_UNROLL
  for (int c = c_lo; c < (c_hi < 3 ? c_hi : 3); ++c) column(SeedTag<1>{}, c);
  TSAT_NO_UNROLL
  for (int c = (c_lo > 3 ? c_lo : 3); c < (c_hi < 7 ? c_hi : 7); ++c) column(SeedTag<2>{}, c);
  TSAT_NO_UNROLL
  for (int c = (c_lo > 7 ? c_lo : 7); c < c_hi; ++c) column(SeedTag<3>{}, c);
}
template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_DEV void rk_jacobian(const Traj<real>& tr, const real x[7], const real u[3], const real b0[3],
                          const real b1[3], const real b2[3], real* F) {
  rk_jacobian_cols<real, INTEG, DIAGJ, ES>(tr, x, u, b0, b1, b2, F, 0, 10);
}

// o = G(q)' r for a 4-vector r, G(q) = [-v'; s I + hat(v)] with the raw state quaternion (src/attitude_controller.jl:69)
template <typename real>
TSAT_DEV void gt_apply(const real q[4], real r0, real r1, real r2, real r3, real o[3]) {
  const real s = q[0], v0 = q[1], v1 = q[2], v2 = q[3];
  // (two-product terms with the rounding spelled out, see dmm_: every build of the kernel rounds them the same way)
  o[0] = dmm_(s, r1, v0, r0) + dmm_(v2, r2, v1, r3);
  o[1] = fma_(s, r2, v0 * r3) - fma_(v2, r1, v1 * r0);
  o[2] = dmm_(v1, r1, v2, r0) + dmm_(s, r3, v0, r2);
}

// The same in ERROR coordinates (error_state = 1), columns c_lo .. c_hi - 1 of [A^|B^] (6 x 9):
//   A^ = E(q_{k+1})' A E(q_k), B^ = E(q_{k+1})' B, E(q) = blkdiag(I3, G(q))   (src/attitude_controller.jl:59-81)
// A E(q_k) is taken by linearity: the tangent pass of column c is seeded with column c of E(q_k) — a unit rate, or a column of
// G(q_k) in the quaternion slots — instead of forming the four quaternion columns of A and mixing them afterwards (nine
// passes instead of ten, and no pass over the finished record); E(q_{k+1})' acts on each finished column by itself. qn is the
// NOMINAL next quaternion, as in the reference. Column c lands at F[c * FSS + 0..5] (FSS: column stride of the record).
template <typename real, int INTEG, int DIAGJ, typename FP, int FSS = FS>
TSAT_DEV void rk_jacobian_es_cols(const Traj<real>& tr, const real x[7], const real u[3], const real b0[3], const real b1[3],
                                  const real b2[3], const real qn[4], FP F, int c_lo, int c_hi) {
  RkStages<real> st;
  rk_primal<real, INTEG, DIAGJ>(tr, x, u, b0, b1, b2, st);
  const real sq = x[3], v0 = x[4], v1 = x[5], v2 = x[6];
  auto column = [&](auto seed, int c) {
    real ex[7], du[3], out[7], o[3];
    const int t = c - 3;      // column of G(q_k) = [-v'; s I + hat(v)] on the attitude columns
    for (int i = 0; i < 3; ++i) ex[i] = (i == c) ? (real)1 : (real)0;
    ex[3] = (t == 0) ? -v0 : (t == 1 ? -v1 : (t == 2 ? -v2 : (real)0));
    ex[4] = (t == 0) ? sq : (t == 1 ? -v2 : (t == 2 ? v1 : (real)0));
    ex[5] = (t == 0) ? v2 : (t == 1 ? sq : (t == 2 ? -v0 : (real)0));
    ex[6] = (t == 0) ? -v1 : (t == 1 ? v0 : (t == 2 ? sq : (real)0));
    for (int a = 0; a < 3; ++a) du[a] = (c == 6 + a) ? control_scale<real, DIAGJ>(tr) : (real)0;
    rk_tangent<real, INTEG, DIAGJ, decltype(seed)::value>(tr, st, b0, b1, b2, ex, du, out);
    gt_apply(qn, out[3], out[4], out[5], out[6], o);
    for (int i = 0; i < 3; ++i) { F[c * FSS + i] = out[i]; F[c * FSS + 3 + i] = o[i]; }
  };
  // one loop per kind of seed (rate, attitude, control): each gets the tangent pass without the terms of its zero parts
  TSAT_NO_UNROLL
  for (int c = c_lo; c < (c_hi < 3 ? c_hi : 3); ++c) column(SeedTag<1>{}, c);
  TSAT_NO_UNROLL
  for (int c = (c_lo > 3 ? c_lo : 3); c < (c_hi < 6 ? c_hi : 6); ++c) column(SeedTag<2>{}, c);
  TSAT_NO_UNROLL
  for (int c = (c_lo > 6 ? c_lo : 6); c < c_hi; ++c) column(SeedTag<3>{}, c);
}

// --------------------------------------------------------------------------------------------------
// cost pieces (A8/A9): LQR objective (src/TortoiseSat.jl:169) + AL terms of the control box (:178) and the
// terminal goal constraint (:182)
// --------------------------------------------------------------------------------------------------
template <typename real>
TSAT_DEV real stage_cost(const Traj<real>& tr, const real x[7], const real u[3], const real lam[6], real mu,
                         bool with_al) {
  real l = 0;
  for (int i = 0; i < 7; ++i) { real e = x[i] - tr.xf[i]; l += (real)0.5 * tr.Qd[i] * e * e; }
  for (int a = 0; a < 3; ++a) l += (real)0.5 * tr.Rd[a] * u[a] * u[a];
  if (with_al) {
    for (int a = 0; a < 3; ++a) {
      real c = u[a] - tr.uhi[a], lm = lam[a];
      l += lm * c;
      real act = (c > 0 || lm > 0) ? mu : (real)0;
      l += (real)0.5 * act * c * c;
    }
    for (int a = 0; a < 3; ++a) {
      real c = tr.ulo[a] - u[a], lm = lam[3 + a];
      l += lm * c;
      real act = (c > 0 || lm > 0) ? mu : (real)0;
      l += (real)0.5 * act * c * c;
    }
  }
  return l;
}
template <typename real>
TSAT_DEV real term_cost(const Traj<real>& tr, const real x[7], const real nu[7], real mu, int mask, bool with_al) {
  real l = 0;
  for (int i = 0; i < 7; ++i) { real e = x[i] - tr.xf[i]; l += (real)0.5 * tr.Qfd[i] * e * e; }
  if (with_al)
    for (int i = 0; i < 7; ++i)
      if ((mask >> i) & 1) { real e = x[i] - tr.xf[i]; l += nu[i] * e + (real)0.5 * mu * e * e; }
  return l;
}

// The same AL stage cost with the per-knot work of the sequential sweep trimmed: half-weights are precomputed and
// the activity test `c > 0 || lambda > 0` becomes one max against a per-row gate staged with the multipliers:
// gate = -inf when lambda > 0 (row always active: t = c) and 0 otherwise (t = max(c, 0)), so I_mu c^2 = mu t^2.
template <typename real>
struct HalfWeights { real hQd[7], hRd[3], hmu; };
template <typename real>
TSAT_DEV real stage_cost_gated(const Traj<real>& tr, const HalfWeights<real>& hw, const real x[7], const real u[3],
                               const real* lam, const real* gate) {
  real l = 0;
  for (int i = 0; i < 7; ++i) { const real e = x[i] - tr.xf[i]; l += hw.hQd[i] * (e * e); }
  for (int a = 0; a < 3; ++a) l += hw.hRd[a] * (u[a] * u[a]);
  for (int a = 0; a < 3; ++a) {
    const real c = u[a] - tr.uhi[a];
    const real t = fmax_(c, gate[a]);
    l += lam[a] * c;
    l += hw.hmu * (t * t);
  }
  for (int a = 0; a < 3; ++a) {
    const real c = tr.ulo[a] - u[a];
    const real t = fmax_(c, gate[3 + a]);
    l += lam[3 + a] * c;
    l += hw.hmu * (t * t);
  }
  return l;
}

// One 16-byte element per lane, global -> LDS, lane-linear: `dst` is THIS lane's destination (base + 2 lane reals; the
// instruction takes the wave-uniform base and adds lane x 16 bytes itself). No VGPR destination, no wait: the data is
// in LDS after the next vmcnt(0). The emulator copies synchronously.
template <typename real>
TSAT_DEV void glds_put(real* dst, const TSAT_GLOBAL real* src) {
  static_assert(sizeof(real) == sizeof(cfg_real), "16-byte copy units are RPU reals");
#ifdef TSAT_EMU
  for (int i = 0; i < RPU; ++i) dst[i] = src[i];
#else
  typedef __attribute__((address_space(1))) const void* gp_t;
  typedef __attribute__((address_space(3))) void* lp_t;
  __builtin_amdgcn_global_load_lds((gp_t)src, (lp_t)(dst - RPU * TSAT_LANE()), 16, 0, 0);
#endif
}
// One (x,u) knot record to HBM as exactly FIVE store instructions (16-byte pieces of the 80-byte double record, 8-byte
// pieces of the 40-byte float record; both are aligned that way by construction).
constexpr int REC_STORES = 5;
template <typename real>
TSAT_DEV void store_record5(TSAT_GLOBAL real* cr, const real x[7], const real u[3]) {
#ifdef TSAT_EMU
  for (int i = 0; i < 7; ++i) cr[i] = x[i];
  for (int c = 0; c < 3; ++c) cr[7 + c] = u[c];
#else
  typedef real v2 __attribute__((ext_vector_type(2)));
  typedef TSAT_GLOBAL v2* gv2;
  v2 a = {x[0], x[1]}, b = {x[2], x[3]}, c = {x[4], x[5]}, d = {x[6], u[0]}, e = {u[1], u[2]};
  ((gv2)cr)[0] = a; ((gv2)cr)[1] = b; ((gv2)cr)[2] = c; ((gv2)cr)[3] = d; ((gv2)cr)[4] = e;
#endif
}
// wait until at most `n` vector-memory operations of the wave are outstanding, then order LDS traffic between the lanes. Loads
// (and LDS copies) retire in issue order among themselves, stores among themselves, but not relative to each other: the only
// safe use is to let n count YOUNGER LOADS — then at most n loads are outstanding and an older load has landed whatever the
// stores do (the record ring of tsat_packed.hpp). The emulator synchronises fully.
#ifdef TSAT_EMU
#define TSAT_SYNC_OLDER_THAN(n) (tsat_emu::sync())
#else
#define TSAT_SYNC_OLDER_THAN(n)                               \
  do {                                                        \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); \
    TSAT_SYNC_LDS();                                          \
  } while (0)
#endif
// the same with the instruction's wave-uniform LDS base given directly: this lane's unit lands at base + RPU * lane
template <typename real>
TSAT_DEV void glds_put_at(real* base, const TSAT_GLOBAL real* src) {
  constexpr int U = 16 / (int)sizeof(real);      // values per 16-byte copy unit
#ifdef TSAT_EMU
  for (int i = 0; i < U; ++i) base[U * TSAT_LANE() + i] = src[i];
#else
  typedef __attribute__((address_space(1))) const void* gp_t;
  typedef __attribute__((address_space(3))) void* lp_t;
  __builtin_amdgcn_global_load_lds((gp_t)src, (lp_t)base, 16, 0, 0);
#endif
}
// BYTES (4 or 16) per lane, global -> LDS, lane-linear from the wave-uniform LDS address `base`: lane l's piece lands at
// base + BYTES * l. Counted in vmcnt like any load; the emulator copies synchronously.
template <int BYTES>
TSAT_DEV void glds_copy(void* base, const TSAT_GLOBAL void* src) {
  static_assert(BYTES == 4 || BYTES == 16, "LDS-DMA piece sizes used here");
#ifdef TSAT_EMU
  unsigned char* d = reinterpret_cast<unsigned char*>(base) + BYTES * TSAT_LANE();
  const unsigned char* q = reinterpret_cast<const unsigned char*>(src);
  for (int i = 0; i < BYTES; ++i) d[i] = q[i];
#else
  typedef __attribute__((address_space(1))) const void* gp_t;
  typedef __attribute__((address_space(3))) void* lp_t;
  if constexpr (BYTES == 4) __builtin_amdgcn_global_load_lds((gp_t)src, (lp_t)base, 4, 0, 0);      // (the size must be a literal)
  else __builtin_amdgcn_global_load_lds((gp_t)src, (lp_t)base, 16, 0, 0);
#endif
}
// the three field rows of step kk of a staged chunk (rows at tau, tau + dtau/2, tau + dtau)
template <typename real>
TSAT_DEV void fwd_brows(const real* fb, int kk, real b0[3], real b1[3], real b2[3]) {
  if (BROW_UNITS == 2) {   // (b0, b1) in FB_BA, (b2, pad) in FB_BB
    const real* ba = fb + FB_BA + kk * 6;
    const real* bb = fb + FB_BB + kk * 6;
    b0[0] = ba[0]; b0[1] = ba[1]; b0[2] = bb[0];
    b1[0] = ba[2]; b1[1] = ba[3]; b1[2] = bb[2];
    b2[0] = ba[4]; b2[1] = ba[5]; b2[2] = bb[4];
  } else {                 // whole 16-byte rows in FB_BA
    const real* r = fb + FB_BA + kk * 12;
    b0[0] = r[0]; b0[1] = r[1]; b0[2] = r[2];
    b1[0] = r[4]; b1[1] = r[5]; b1[2] = r[6];
    b2[0] = r[8]; b2[1] = r[9]; b2[2] = r[10];
  }
}

// Issue the copy of one forward chunk (knots k0 .. k0 + nk - 1) into the chunk buffer `fb`: gains K,d (closed-loop sweeps),
// nominal (x,u) records and the three field rows of every step. On the GPU these are
// global_load_lds_dwordx4 instructions: no VGPR destination (the sweep is at the register cap) and no wait — the data is
// guaranteed in LDS after the next vmcnt(0) (TSAT_SYNC). A lane past the end of a short last chunk copies the last element
// again into the padding. The emulator copies synchronously.
template <typename real>
TSAT_DEV void fwd_chunk_issue(real* fb, const TSAT_GLOBAL real* KDg, const TSAT_GLOBAL real* XUg,
                              const Traj<real>& tr, int k0, int nk, int closed) {
  static_assert(sizeof(real) == sizeof(cfg_real), "16-byte copy units are RPU reals");
  const int lane = TSAT_LANE();
  // in 16-byte units, rounded up: a float chunk of an odd number of 40- / 24-byte records ends inside a unit whose tail is
  // the next record (x,u: the terminal knot's record exists) or the slab's alignment padding (multipliers)
  const int n2k = (nk * KDW + RPU - 1) / RPU, n2x = (nk * XUW + RPU - 1) / RPU, nb = nk * 3;
  const TSAT_GLOBAL real* kd = KDg + (size_t)k0 * KDW;
  const TSAT_GLOBAL real* xu = XUg + (size_t)k0 * XUW;
  auto put = [&](real* dst, const TSAT_GLOBAL real* src) { glds_put<real>(dst, src); };
  if (closed)
    for (int j = 0; j < (CK * KDW + GLDS - 1) / GLDS; ++j) {
      const int i = lane + WAVE * j, ic = (i < n2k) ? i : n2k - 1;
      put(fb + FB_KD + RPU * i, kd + RPU * ic);
    }
  for (int j = 0; j < (CK * XUW + GLDS - 1) / GLDS; ++j) {
    const int i = lane + WAVE * j, ic = (i < n2x) ? i : n2x - 1;
    put(fb + FB_XU + RPU * i, xu + RPU * ic);
  }
  for (int j = 0; j < (CK * 3 * RPU + GLDS - 1) / GLDS; ++j) {
    const int e = lane + WAVE * j, ec = (e < nb) ? e : nb - 1;
    const int kk = ec / 3, st = ec - 3 * kk;
    const TSAT_GLOBAL real* br = tr.bt + (size_t)brow_index(tr, k0 + kk, 0.5 * (double)st) * 4;
    put(fb + FB_BA + RPU * e, br);
    if (BROW_UNITS == 2) put(fb + FB_BB + RPU * e, br + 2);
  }
}
template <typename real> struct FwdOut { acc_t J; int ok; };
template <typename real> struct BwdOut { acc_t dV1, dV2; int pd_ok; };

// what the roll-out of one knot reads from the staged chunk: nominal (x,u) record, gains K (3 x 7 row-major) and d, three field rows
template <typename real> struct KnotIn { real xu[XUW], kd[KDW], b0[3], b1[3], b2[3]; };
template <typename real>
TSAT_DEV KnotIn<real> fwd_knot_load(const real* fb, int kk) {
  KnotIn<real> in;
  const real* xu = fb + FB_XU + kk * XUW;
  const real* kd = fb + FB_KD + kk * KDW;
  for (int i = 0; i < XUW; ++i) in.xu[i] = xu[i];
  for (int i = 0; i < KDW; ++i) in.kd[i] = kd[i];
  fwd_brows<real>(fb, kk, in.b0, in.b1, in.b2);
  return in;
}

// --------------------------------------------------------------------------------------------------
// forward sweep: all line-search candidates at once (lane j: alpha = 2^-(j + alpha_shift)). The first n_cand lanes keep their
// roll-outs in HBM (CAND[lane]); NOTHING else is computed in the sequential loop — the AL cost and the validity of a candidate are
// evaluated afterwards from its stored records with lanes = knots (candidate_costs), 64 knots per instruction instead of one: the
// per-knot cost terms were a fifth of the loop's instructions, on the critical path of the launch, and only the first candidates
// of a sweep ever need them.
// --------------------------------------------------------------------------------------------------
template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_FWD void forward_sweep(TPtrs<real> p, int N, int n_tab, int closed, int n_cand, int alpha_shift) {
  real* lds = lds_base<real>();
  const Traj<real> tr = load_traj<real>(N, n_tab, p.bt);
  const int lane = TSAT_LANE();
  real alpha = 1;   // lane j rolls out alpha = 2^-(j + alpha_shift); the first n_cand lanes keep their rollout in HBM
  for (int j = 0; j < lane + alpha_shift && j < TSAT_MAX_LINESEARCH; ++j) alpha *= (real)0.5;
  real x[7];
  for (int i = 0; i < 7; ++i) x[i] = lds[L_TR + P_X0 + i];
  const TSAT_GLOBAL real* XUg = p.XU;
  const TSAT_GLOBAL real* KDg = p.KD;
  TSAT_GLOBAL real* Cg = slab_ptr<real>(p, N, cand_slab(p.cur, lane < n_cand ? lane : 0));
  // chunk pipeline: buffer `cur` holds the chunk being rolled out; with two buffers the copy of the next chunk is issued
  // before the roll-out and has landed long before it ends (32 knots ~ 30 us against ~1 us of memory latency)
  int cur = 0;
  fwd_chunk_issue<real>(lds + L_FWD, KDg, XUg, tr, 0, (N - 1 < CK) ? N - 1 : CK, closed);
  TSAT_SYNC();
  for (int k0 = 0; k0 < N - 1; k0 += CK) {
    const int nk = (N - 1 - k0 < CK) ? (N - 1 - k0) : CK;
    const int kn = k0 + CK, nkn = (N - 1 - kn < CK) ? (N - 1 - kn) : CK;     // next chunk (nkn <= 0: none)
    real* fb = lds + L_FWD + cur * FB_SIZE;
    real* fbn = lds + L_FWD + ((FWD_NBUF == 2) ? (1 - cur) : 0) * FB_SIZE;
    if (FWD_NBUF == 2 && nkn > 0) fwd_chunk_issue<real>(fbn, KDg, XUg, tr, kn, nkn, closed);
    // The staged records are read one knot AHEAD: knot kk + 1's nominal record, gains and field rows are requested from LDS
    // before knot kk is rolled out and have long arrived when its turn comes. Left to the compiler the reads sit right in front
    // of their first use (register pressure), nine exposed LDS latencies per knot on the critical path of the launch.
    auto knot = [&](const KnotIn<real>& kin, int kk) {
      const real* xu = kin.xu;
      real u[3] = {xu[7], xu[8], xu[9]};
      if (closed) {
        const real* kd = kin.kd;
        real dx[7];
        if (ES) {
          // quaternion_error(new, nominal) = [dw; MRP(q_nom^-1 (x) q_new)]  (src/quaternion_toolbox.jl:58-75)
          for (int i = 0; i < 3; ++i) dx[i] = x[i] - xu[i];
          const real s1 = xu[3], a1 = -xu[4], a2 = -xu[5], a3 = -xu[6];   // conjugate of the nominal quaternion
          const real s2 = x[3], b1 = x[4], b2 = x[5], b3 = x[6];
          const real e0 = s1 * s2 - (a1 * b1 + a2 * b2 + a3 * b3);
          const real e1 = s1 * b1 + s2 * a1 + (a2 * b3 - a3 * b2);
          const real e2 = s1 * b2 + s2 * a2 + (a3 * b1 - a1 * b3);
          const real e3 = s1 * b3 + s2 * a3 + (a1 * b2 - a2 * b1);
          const real ir = rcp_((real)1 + e0);
          dx[3] = e1 * ir; dx[4] = e2 * ir; dx[5] = e3 * ir; dx[6] = 0;
        } else {
          for (int i = 0; i < 7; ++i) dx[i] = x[i] - xu[i];
        }
        for (int c = 0; c < 3; ++c) {
          real v = u[c];
          for (int j = 0; j < BwdCfg<ES>::NH; ++j) v += kd[c * 7 + j] * dx[j];
          u[c] = v + alpha * kd[21 + c];
        }
      }
      if (lane < n_cand) {
        TSAT_GLOBAL real* cr = Cg + (size_t)(k0 + kk) * XUW;
        for (int i = 0; i < 7; ++i) cr[i] = x[i];
        for (int c = 0; c < 3; ++c) cr[7 + c] = u[c];
      }
      real xn[7];
      rk_step<real, INTEG, DIAGJ, ES>(tr, x, u, kin.b0, kin.b1, kin.b2, xn);    // stage rows tau, tau + dtau/2, tau + dtau
      for (int i = 0; i < 7; ++i) x[i] = xn[i];
    };
    // two knots per turn, two register sets A / B: while one is rolled out the other is on its way from LDS (a single set would
    // have to be copied, forty register moves per knot)
#if defined(TSAT_DENSE) && !defined(TSAT_F32)
    // the dense double build (256 registers in all, no AGPRs) cannot hold two read-ahead sets of 43 doubles without spilling
    // into this loop; its second wavefront per SIMD covers part of the LDS latency instead
    for (int kk = 0; kk < nk; ++kk) {
      const KnotIn<real> A = fwd_knot_load<real>(fb, kk);
      knot(A, kk);
    }
#else
    KnotIn<real> A = fwd_knot_load<real>(fb, 0);
    for (int kk = 0; kk < nk; kk += 2) {
      TSAT_WAIT_LDS();               // A has arrived (requested a whole knot ago); only then the next batch is put in flight
      const KnotIn<real> B = fwd_knot_load<real>(fb, (kk + 1 < nk) ? kk + 1 : kk);
      TSAT_SCHED_FENCE();
      knot(A, kk);
      if (kk + 1 < nk) {
        TSAT_WAIT_LDS();
        A = fwd_knot_load<real>(fb, (kk + 2 < nk) ? kk + 2 : kk + 1);
        TSAT_SCHED_FENCE();
        knot(B, kk + 1);
      }
    }
#endif
    if (nkn > 0) {
      if (FWD_NBUF == 1) {          // single buffer: copy the next chunk now that this one has been consumed
        TSAT_SYNC_LDS();
        fwd_chunk_issue<real>(fbn, KDg, XUg, tr, kn, nkn, closed);
      }
      TSAT_SYNC();                  // vmcnt(0): the copy issued before (or just after) the roll-out has landed
      if (FWD_NBUF == 2) cur = 1 - cur;
    }
  }
  if (lane < n_cand) {
    TSAT_GLOBAL real* cr = Cg + (size_t)(N - 1) * XUW;
    for (int i = 0; i < 7; ++i) cr[i] = x[i];
    for (int c = 0; c < 3; ++c) cr[7 + c] = 0;
  }
}

// AL cost and validity of the stored candidates c0 .. c0 + nc - 1 (nc <= CG) of the sweep just finished, for lane c < nc the
// candidate c0 + c. Lanes = knots: every lane evaluates the stage cost of its knots of each candidate (the very arithmetic the
// packed builds run inside their sequential sweeps: stage_cost_gated on the stored record) and leaves it in LDS, KB knots at a
// time; lane c then adds candidate c's costs IN KNOT ORDER, so J is, bit for bit, the sum a sequential roll-out accumulates. A
// roll-out is valid when no |x_i|, |u_i| exceeded max_state and J is a number (a non-finite roll-out leaves amax = inf or J = NaN).
// The records come from HBM: a step's loads (64 knots: multipliers + the candidates' records) are issued one step ahead of
// their use, two register sets taking turns.
template <typename real> struct CostIn { real lam[6], r[CG][XUW]; };
template <typename real>
TSAT_PHASE FwdOut<real> candidate_costs(TPtrs<real> p, int N, int c0, int nc, real mu, int term_mask, real max_state) {
  real* lds = lds_base<real>();
  const Traj<real> tr = load_traj<real>(N, 1, p.bt);
  const int lane = TSAT_LANE();
  HalfWeights<real> hw;
  for (int i = 0; i < 7; ++i) hw.hQd[i] = (real)0.5 * tr.Qd[i];
  for (int i = 0; i < 3; ++i) hw.hRd[i] = (real)0.5 * tr.Rd[i];
  hw.hmu = (real)0.5 * mu;
  real* L = lds + L_FWD;                       // [CG][KBS]
  real am[CG];
  for (int c = 0; c < CG; ++c) am[c] = 0;
  acc_t J = 0;
  const int cme = (lane < nc) ? lane : 0;
  const TSAT_GLOBAL real* Cc[CG];
  for (int c = 0; c < CG; ++c) Cc[c] = slab_ptr<real>(p, N, cand_slab(p.cur, c0 + (c < nc ? c : 0)));
  const int S = (N + WAVE - 1) / WAVE;         // steps of 64 knots
  constexpr int SPB = KB / WAVE;               // steps per summed block
  auto load = [&](int st) {
    CostIn<real> in;
    const int k = st * WAVE + lane, kx = (k < N) ? k : N - 1, kl = (k < N - 1) ? k : 0;
    for (int e = 0; e < 6; ++e) in.lam[e] = p.LAM[(size_t)kl * LMW + e];
    for (int c = 0; c < CG; ++c)
      for (int e = 0; e < XUW; ++e) in.r[c][e] = (c < nc) ? Cc[c][(size_t)kx * XUW + e] : (real)0;
    return in;
  };
  auto step = [&](const CostIn<real>& in, int st) {
    const int k = st * WAVE + lane;
    real gate[6];
    for (int e = 0; e < 6; ++e) gate[e] = (in.lam[e] > 0) ? -inf_<real>() : (real)0;
    for (int c = 0; c < CG; ++c) {
      if (c >= nc) break;
      real a = am[c];
      if (k < N)
        for (int e = 0; e < 7; ++e) a = fmaxabs_(a, in.r[c][e]);
      if (k < N - 1) {
        for (int e = 7; e < 10; ++e) a = fmaxabs_(a, in.r[c][e]);
        L[c * KBS + (st % SPB) * WAVE + lane] = stage_cost_gated(tr, hw, in.r[c], in.r[c] + 7, in.lam, gate);
      }
      am[c] = a;
    }
    if ((st % SPB) == SPB - 1 || st == S - 1) {          // a block is complete: candidate `lane` adds its costs in knot order
      TSAT_SYNC_LDS();
      if (lane < nc) {
        const int k0 = (st / SPB) * KB;
        const int n = (N - 1 - k0 < KB) ? (N - 1 - k0) : KB;
        const real* Lc = L + cme * KBS;
        int kk = 0;
        for (; kk + 16 <= n; kk += 16) {                 // the reads of a batch are issued together, the adds stay one chain
          real v[16];
          for (int e = 0; e < 16; ++e) v[e] = Lc[kk + e];
          for (int e = 0; e < 16; ++e) J += (acc_t)v[e];
        }
        for (; kk < n; ++kk) J += (acc_t)Lc[kk];
      }
      TSAT_SYNC_LDS();
    }
  };
  CostIn<real> A = load(0);
  for (int st = 0; st < S; st += 2) {
    const CostIn<real> B = load((st + 1 < S) ? st + 1 : st);
    step(A, st);
    if (st + 1 < S) {
      A = load((st + 2 < S) ? st + 2 : st + 1);
      step(B, st + 1);
    }
  }
  real amax = 0;
  for (int c = 0; c < CG; ++c) {
    if (c >= nc) break;
    const real m = wave_max(am[c], lds + L_RED);
    amax = (lane == c) ? m : amax;
  }
  real nu[7], xN[7];
  const TSAT_GLOBAL real* cN = slab_ptr<real>(p, N, cand_slab(p.cur, c0 + cme)) + (size_t)(N - 1) * XUW;
  for (int i = 0; i < 7; ++i) { nu[i] = lds[L_NU + i]; xN[i] = cN[i]; }
  J += (acc_t)term_cost(tr, xN, nu, mu, term_mask, true);
  FwdOut<real> out;
  out.J = J;
  out.ok = (lane < nc && (amax <= max_state) && (J == J)) ? 1 : 0;
  return out;
}


// control gradient and Hessian diagonal of one knot with the AL terms of the control box: lu = R u + (lambda+ + I mu c+) -
// (lambda- + I mu c-), luu = R + I+ mu + I- mu, row active iff c > 0 or lambda > 0. The fused operations are spelled out
// (see dmm_): the Jacobian lanes of every build round them the same way.
template <typename real>
TSAT_DEV void al_control_terms(const Traj<real>& tr, const real u[3], const real lam[6], real mu, real* lu_out, real* luu_out) {
  for (int c = 0; c < 3; ++c) {
    const real chi = u[c] - tr.uhi[c], clo = tr.ulo[c] - u[c];
    const real mhi = (chi > 0 || lam[c] > 0) ? mu : (real)0;
    const real mlo = (clo > 0 || lam[3 + c] > 0) ? mu : (real)0;
    real lu = fma_(tr.Rd[c], u[c], fma_(mhi, chi, lam[c]));
    lu = lu - fma_(mlo, clo, lam[3 + c]);
    lu_out[c] = lu;
    luu_out[c] = (tr.Rd[c] + mhi) + mlo;
  }
}

// --------------------------------------------------------------------------------------------------
// The knot record of one knot (layout PkRec<ES>, values of type jac_t) through the pointer `rc` (LDS, or generic in the packed
// builds): [A|B] — in error coordinates A^ = E(q_{k+1})' A E(q_k), B^ = E(q_{k+1})' B column by column (rk_jacobian_es_cols) —
// from the tangent passes in jac_t arithmetic (float in the mixed-precision builds: the linearisation point and the trajectory
// constants are rounded once, the passes run at the float rate with half the registers), and the cost terms lx (error
// coordinates: lx^ = E(q_k)' lx), G'QG = E(q_k)' Q E(q_k) (src/quaternion_toolbox.jl:15-36), lu, luu with the AL terms of the
// control box in `real` arithmetic, rounded to jac_t when stored. qn: the NOMINAL next quaternion (error coordinates).
// --------------------------------------------------------------------------------------------------
template <typename T, typename S>
TSAT_DEV Traj<T> traj_cast(const Traj<S>& a) {
  Traj<T> t;
  for (int i = 0; i < 7; ++i) { t.xf[i] = (T)a.xf[i]; t.Qd[i] = (T)a.Qd[i]; t.Qfd[i] = (T)a.Qfd[i]; }
  for (int i = 0; i < 3; ++i) { t.Rd[i] = (T)a.Rd[i]; t.ulo[i] = (T)a.ulo[i]; t.uhi[i] = (T)a.uhi[i]; }
  for (int i = 0; i < 9; ++i) { t.J[i] = (T)a.J[i]; t.hJi[i] = (T)a.hJi[i]; }
  t.h = (T)a.h; t.hh = (T)a.hh; t.us = (T)a.us; t.usj = (T)a.usj;
  t.tau0 = a.tau0; t.dtau = a.dtau; t.N = a.N; t.n_tab = a.n_tab; t.bt = nullptr;
  return t;
}
template <typename real, int INTEG, int DIAGJ, int ES, typename JP>
TSAT_DEV void knot_record(const Traj<real>& tr, const real x[7], const real u[3], const real lam[6], const real b0[3], const real b1[3],
                          const real b2[3], const real qn[4], real mu, JP rc) {
  using R = PkRec<ES>;
  if constexpr (sizeof(jac_t) == sizeof(real)) {
    if (!ES) rk_jacobian_cols<real, INTEG, DIAGJ, ES, JP, R::FSR>(tr, x, u, b0, b1, b2, rc, 0, 10);
    else rk_jacobian_es_cols<real, INTEG, DIAGJ, JP, R::FSR>(tr, x, u, b0, b1, b2, qn, rc, 0, 9);
  } else {
    const Traj<jac_t> tj = traj_cast<jac_t, real>(tr);
    jac_t xj[7], uj[3], c0[3], c1[3], c2[3], qj[4];
    for (int i = 0; i < 7; ++i) xj[i] = (jac_t)x[i];
    for (int i = 0; i < 3; ++i) { uj[i] = (jac_t)u[i]; c0[i] = (jac_t)b0[i]; c1[i] = (jac_t)b1[i]; c2[i] = (jac_t)b2[i]; }
    for (int i = 0; i < 4; ++i) qj[i] = (jac_t)qn[i];
    if (!ES) rk_jacobian_cols<jac_t, INTEG, DIAGJ, ES, JP, R::FSR>(tj, xj, uj, c0, c1, c2, rc, 0, 10);
    else rk_jacobian_es_cols<jac_t, INTEG, DIAGJ, JP, R::FSR>(tj, xj, uj, c0, c1, c2, qj, rc, 0, 9);
  }
  real lx[7];
  for (int i = 0; i < 7; ++i) lx[i] = tr.Qd[i] * (x[i] - tr.xf[i]);
  if (ES) {
    const real qk[4] = {x[3], x[4], x[5], x[6]};
    real o[3];
    gt_apply(qk, lx[3], lx[4], lx[5], lx[6], o);
    lx[3] = o[0]; lx[4] = o[1]; lx[5] = o[2];
    // G' diag(Qd[3:7]) G, upper triangle (0,0)(0,1)(0,2)(1,1)(1,2)(2,2); G rows: [-v'; s I + hat(v)]
    const real sq = qk[0], v0 = qk[1], v1 = qk[2], v2 = qk[3];
    const real G[4][3] = {{-v0, -v1, -v2}, {sq, -v2, v1}, {v2, sq, -v0}, {-v1, v0, sq}};
    int idx = 0;
    for (int j = 0; j < 3; ++j)
      for (int l = j; l < 3; ++l) {
        real acc = 0;
        for (int r = 0; r < 4; ++r) acc += G[r][j] * tr.Qd[3 + r] * G[r][l];
        rc[R::QQ + idx++] = (jac_t)acc;
      }
  }
  for (int i = 0; i < BwdCfg<ES>::NH; ++i) rc[R::LX + i] = (jac_t)lx[i];
  real lu[3], luu[3];
  al_control_terms(tr, u, lam, mu, lu, luu);
  for (int c = 0; c < 3; ++c) { rc[R::LU + c] = (jac_t)lu[c]; rc[R::LUU + c] = (jac_t)luu[c]; }
}

// the chunk's records in LDS (values of type jac_t, from L_REC on)
TSAT_DEV jac_t* rec_base() { return reinterpret_cast<jac_t*>(lds_base<cfg_real>() + L_REC); }

// --------------------------------------------------------------------------------------------------
// Jacobian lanes of one backward chunk: lane l linearises knot k0 + l and leaves its record in LDS
// --------------------------------------------------------------------------------------------------
template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_PHASE void jacobian_chunk(TPtrs<real> p, int N, int n_tab, int k0, int nk, real mu) {
  const int lane = TSAT_LANE();
  if (lane < nk) {
    const Traj<real> tr = load_traj<real>(N, n_tab, p.bt);
    const int k = k0 + lane;
    const TSAT_GLOBAL real* xu = p.XU + (size_t)k * XUW;
    real x[7], u[3], lam[6], b0[3], b1[3], b2[3], qn[4];
    for (int i = 0; i < 7; ++i) x[i] = xu[i];
    for (int c = 0; c < 3; ++c) u[c] = xu[7 + c];
    for (int c = 0; c < 6; ++c) lam[c] = p.LAM[(size_t)k * LMW + c];
    for (int i = 0; i < 4; ++i) qn[i] = xu[XUW + 3 + i];
    const TSAT_GLOBAL real* p0 = tr.bt + (size_t)brow_index(tr, k, 0.0) * 4;
    const TSAT_GLOBAL real* p1 = tr.bt + (size_t)brow_index(tr, k, 0.5) * 4;
    const TSAT_GLOBAL real* p2 = tr.bt + (size_t)brow_index(tr, k, 1.0) * 4;
    for (int c = 0; c < 3; ++c) { b0[c] = p0[c]; b1[c] = p1[c]; b2[c] = p2[c]; }
    knot_record<real, INTEG, DIAGJ, ES, jac_t*>(tr, x, u, lam, b0, b1, b2, qn, mu, rec_base() + lane * PkRec<ES>::RECS);
  }
}

// --------------------------------------------------------------------------------------------------
// backward sweep = Jacobian lanes + Riccati recursion. pd_ok = 0 (wave-uniform) when some Quu_reg is not PD.
// --------------------------------------------------------------------------------------------------
TSAT_DEV void pair_ut(int L, int n, int& i, int& j) {  // L in [0, n(n+1)/2) -> (i<=j) of an n x n upper triangle, row by row
  int r = 0, base = 0;
  while (L >= base + (n - r)) { base += n - r; ++r; }
  i = r;
  j = r + (L - base);
}

// Riccati recursion over one chunk whose Jacobian records are in LDS (last knot first). Its own function, so that
// the lane-role tables live in registers for exactly this loop (nothing survives the call to jacobian_chunk).
// NH = 7: plain state differences; NH = 6: error coordinates (records reduced by jacobian_chunk). R: layout of the knot record.
template <typename real, int NH, typename R>
TSAT_PHASE BwdOut<real> riccati_chunk(TSAT_GLOBAL real* KDg, int k0, int nk, real rho, acc_t dV1, acc_t dV2) {
  constexpr int ES = (NH == 6) ? 1 : 0;
  constexpr int RECS = R::RECS, FSR = R::FSR, R_LX = R::LX, R_LU = R::LU, R_LUU = R::LUU, R_QQ = R::QQ;
  constexpr int NC = NH + 3;               // columns of [A|B]
  constexpr int NP = NH * (NH + 1) / 2;    // unique entries of a symmetric NH x NH block
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  // ---- lane roles. Every step is branch-free: each lane owns LDS offsets for its operands and outputs; lanes
  // without a role in a step compute on harmless operands and write to L_SINK. -----------------------------
  // step 1: W~[r1][c1] = sum_m S~[r1][m] F[m][c1] (row NH of S~ is s'), plus columns >= 8 on the lanes with small c1
  // (NH = 6: the 7 x 9 elements of W~ are exactly 63 lanes, one dot product each; NH = 7: 8 x 10 = 80 elements, lane (r, c) of
  // an 8 x 8 grid takes column c and, for small c, column 8 + c as well)
  constexpr bool ONE_DOT = (NH + 1) * NC <= WAVE;
  const int r1 = ONE_DOT ? lane % (NH + 1) : (lane & 7), c1 = ONE_DOT ? lane / (NH + 1) : (lane >> 3);
  const int r1c = (r1 <= NH) ? r1 : NH;
  const int s1_st = L_ST + r1c * 9;
  const int s1_fa = ((c1 < NC) ? c1 : 0) * FSR, s1_fb = ((8 + (c1 & 1) < NC) ? (8 + (c1 & 1)) : 0) * FSR;   // relative to the knot record
  const int s1_oa = (r1 <= NH && c1 < NC) ? (L_WT + c1 * 9 + r1) : L_SINK;
  const int s1_ob = (!ONE_DOT && r1 <= NH && 8 + c1 < NC) ? (L_WT + (8 + c1) * 9 + r1) : L_SINK;
  // step 2: acc = diag + init + dot(F[:,colA], opB[0..NH-1]) -> lds[o1], lds[o2]
  int s2_fa = 0, s2_b = L_WT, s2_init = -1, s2_o1 = L_SINK, s2_o2 = L_SINK;
  real s2_diag = 0;
  if (lane < NP) {                                   // Qxx(i,j), i <= j  = lxx + A'SA
    int i, j; pair_ut(lane, NH, i, j);
    s2_fa = i * FSR; s2_b = L_WT + j * 9; s2_o1 = L_HXX + i * 7 + j; s2_o2 = L_HXX + j * 7 + i;
    if (ES) {   // projected Hessian: diag(Qd[0:3]) on the rate block, G'QG (per-knot record) on the attitude block
      if (i == j && i < 3) s2_diag = lds[L_TR + P_QD + i];
      if (i >= 3) { const int a = i - 3, b = j - 3; s2_init = R_QQ + (a == 0 ? b : (a == 1 ? 2 + b : 5)); }
    } else if (i == j) {
      s2_diag = lds[L_TR + P_QD + i];
    }
  } else if (lane < NP + 3 * NH) {                   // Qux(a,j) = B'SA
    const int aa = (lane - NP) / NH, j = (lane - NP) % NH;
    s2_fa = (NH + aa) * FSR; s2_b = L_WT + j * 9; s2_o1 = s2_o2 = L_HUX + aa * 8 + j;
  } else if (lane < NP + 3 * NH + 6) {               // Quu(a,b), a <= b = luu + B'SB
    const int L = lane - (NP + 3 * NH);              // (0,0)(0,1)(0,2)(1,1)(1,2)(2,2)
    const int aa = (L < 3) ? 0 : (L < 5 ? 1 : 2);
    const int bb = (L < 3) ? L : (L < 5 ? L - 2 : 2);
    s2_fa = (NH + aa) * FSR; s2_b = L_WT + (NH + bb) * 9; s2_o1 = L_HUU + aa * 3 + bb; s2_o2 = L_HUU + bb * 3 + aa;
    if (aa == bb) s2_init = R_LUU + aa;
  } else if (lane < NP + 3 * NH + 9) {               // Qu(a) = lu + B's'  -> Hux[a][7]
    const int aa = lane - (NP + 3 * NH + 6);
    s2_fa = (NH + aa) * FSR; s2_b = L_ST + NH * 9; s2_init = R_LU + aa; s2_o1 = s2_o2 = L_HUX + aa * 8 + 7;
  }
  // step 3: K[a3][j3] (j3 == 7: d[a3]) on the lanes with a3 < 3; with NH = 6 the unused gain column 6 is written as 0
  const int a3 = lane >> 3, j3 = lane & 7;
  const int a3c = (a3 < 3) ? a3 : 2;
  const bool s3_live = (j3 < NH) || (j3 == 7);
  const int s3_o = (a3 < 3) ? (L_KD + a3 * 8 + j3) : L_SINK;
  const int s3_slot = (a3 < 3) ? ((j3 < 7) ? (a3 * 7 + j3) : (21 + a3)) : -1;   // position inside the K,d record
  // step 4: S(i,j), i <= j on lanes 0..NP-1; s(i) on the next NH lanes written as the "column 7" of the same formula
  int i4 = 0, j4 = 0, s4_b1 = L_ZERO, s4_b1rel = -1, s4_b2 = L_ZERO, s4_o1 = L_SINK, s4_o2 = L_SINK;
  if (lane < NP) {
    pair_ut(lane, NH, i4, j4);
    s4_b1 = L_HXX + i4 * 7 + j4; s4_o1 = L_ST + i4 * 9 + j4; s4_o2 = L_ST + j4 * 9 + i4;
  } else if (lane < NP + NH) {
    i4 = lane - NP; j4 = 7;
    s4_b1rel = R_LX + i4; s4_b2 = L_WT + i4 * 9 + NH; s4_o1 = s4_o2 = L_ST + NH * 9 + i4;
  }
  const int s4_hi = L_HUX + i4, s4_hj = L_HUX + j4, s4_ki = L_KD + i4, s4_kj = L_KD + j4;

  // The recursion is software-pipelined by hand: in every step the reads that depend on the previous step are issued
  // first, then the reads that do not (operands of later steps, the next knot's Jacobian columns), then a scheduling
  // fence, then the arithmetic. LDS returns data in issue order, so the arithmetic waits only for the dependent
  // reads while the prefetches ride along under it.
  bool pd_ok = true;
  real fa1[NH], fb1[NH], fa2[NH], ini2;
  {
    const int rcb = L_REC + (nk - 1) * RECS;
    for (int m = 0; m < NH; ++m) { fa1[m] = lds[rcb + s1_fa + m]; fb1[m] = ONE_DOT ? (real)0 : lds[rcb + s1_fb + m]; fa2[m] = lds[rcb + s2_fa + m]; }
    ini2 = lds[(s2_init >= 0) ? (rcb + s2_init) : L_ZERO];
  }
  for (int l = nk - 1; l >= 0; --l) {
    const int rcb = L_REC + l * RECS;
    // step 1: W~ = [S; s'] [A|B]   ((NH+1) x (NH+3))
    {
      real sv[NH];
      for (int m = 0; m < NH; ++m) sv[m] = lds[s1_st + m];
      TSAT_SCHED_FENCE();
      real acc = 0, acc2 = 0;
      for (int m = 0; m < NH; ++m) {
        acc = fma_(sv[m], fa1[m], acc);
        if (!ONE_DOT) acc2 = fma_(sv[m], fb1[m], acc2);
      }
      role_store(lds, s1_oa, L_SINK, acc);
      if (!ONE_DOT) role_store(lds, s1_ob, L_SINK, acc2);
    }
    TSAT_SYNC_LDS();
    // step 2: Qxx = lxx + A'SA, Qux = B'SA, Quu = luu + B'SB, Qu = lu + B's
    {
      real wb[NH];
      for (int m = 0; m < NH; ++m) wb[m] = lds[s2_b + m];
      TSAT_SCHED_FENCE();
      real acc = s2_diag + ini2;
      for (int m = 0; m < NH; ++m) acc = fma_(fa2[m], wb[m], acc);
      role_store(lds, s2_o1, L_SINK, acc);
      role_store(lds, s2_o2, L_SINK, acc);
    }
    TSAT_SYNC_LDS();
    // step 3: regularise, PD test (Sylvester), adjugate inverse, K = -Quu_reg^-1 Qux, d = -Quu_reg^-1 Qu
    real qu0, qu1, qu2, hi[3], hj[3], b1, b2;
    {
      const real* Huu = lds + L_HUU;
      const real* Hux = lds + L_HUX;
      real hu[9];
      hu[0] = Huu[0]; hu[1] = Huu[1]; hu[2] = Huu[2]; hu[4] = Huu[4]; hu[5] = Huu[5]; hu[8] = Huu[8];
      const real h0 = Hux[0 * 8 + j3], h1 = Hux[1 * 8 + j3], h2 = Hux[2 * 8 + j3];
      // prefetch for step 4 (all produced by step 2): Qu, the Qux columns i and j, the base terms
      qu0 = lds[L_HUX + 7]; qu1 = lds[L_HUX + 15]; qu2 = lds[L_HUX + 23];
      for (int c = 0; c < 3; ++c) { hi[c] = lds[s4_hi + c * 8]; hj[c] = lds[s4_hj + c * 8]; }
      b1 = lds[(s4_b1rel >= 0) ? (rcb + s4_b1rel) : s4_b1];
      b2 = lds[s4_b2];
      TSAT_SCHED_FENCE();
      const real q00 = hu[0] + rho, q11 = hu[4] + rho, q22 = hu[8] + rho;
      const real q10 = hu[1], q20 = hu[2], q21 = hu[5];
      const real c00 = dmm_(q11, q22, q21, q21);
      const real c01 = dmm_(q20, q21, q10, q22);
      const real c02 = dmm_(q10, q21, q20, q11);
      const real c11 = dmm_(q00, q22, q20, q20);
      const real c12 = dmm_(q10, q20, q00, q21);
      const real c22 = dmm_(q00, q11, q10, q10);
      const real det = dot3_(q00, c00, q10, c01, q20, c02);
      if (!(q00 > 0 && c22 > 0 && det > 0)) pd_ok = false;
      const real nid = -rcp_(det);
      // row a3 of the inverse (negated)
      const real Qi0 = ((a3c == 0) ? c00 : (a3c == 1 ? c01 : c02)) * nid;
      const real Qi1 = ((a3c == 0) ? c01 : (a3c == 1 ? c11 : c12)) * nid;
      const real Qi2 = ((a3c == 0) ? c02 : (a3c == 1 ? c12 : c22)) * nid;
      const real vv = dot3_(Qi0, h0, Qi1, h1, Qi2, h2);
      const real v = s3_live ? vv : (real)0;
      role_store(lds, s3_o, L_SINK, v);
      if (s3_slot >= 0) KDg[(size_t)(k0 + l) * KDW + s3_slot] = v;   // stays in flight: no vmcnt wait in this loop
    }
    if (!pd_ok) break;  // wave-uniform: every lane read the same Huu
    TSAT_SYNC_LDS();
    // step 4: cost-to-go. With K = -Quu_reg^-1 Qux: Quu K + Qux = -rho K, so
    //   Sxx = Qxx + sym(Qux'K) - rho K'K ;  Sx = Qx + sym(Qux'd, Qu'K) - rho K'd   (Appendix A, compacted;
    //   Qux'd = Qu'K in exact arithmetic, so the s-lanes run the very same formula with j = "column 7")
    {
      real ki[3], kj[3];
      const real d0 = lds[L_KD + 7], d1 = lds[L_KD + 15], d2 = lds[L_KD + 23];
      for (int c = 0; c < 3; ++c) { ki[c] = lds[s4_ki + c * 8]; kj[c] = lds[s4_kj + c * 8]; }
      if (l > 0) {   // prefetch the next knot's Jacobian columns (written by jacobian_chunk, never by this loop)
        const int rcn = rcb - RECS;
        for (int m = 0; m < NH; ++m) { fa1[m] = lds[rcn + s1_fa + m]; if (!ONE_DOT) fb1[m] = lds[rcn + s1_fb + m]; fa2[m] = lds[rcn + s2_fa + m]; }
        ini2 = lds[(s2_init >= 0) ? (rcn + s2_init) : L_ZERO];
      }
      TSAT_SCHED_FENCE();
      // dV1 += d'Qu ; dV2 += 0.5 d'Quu d. With d = -Quu_reg^-1 Qu: Quu d = -Qu - rho d, so d'Quu d = -(d'Qu + rho d'd)
      const real dqu = dot3_(d0, qu0, d1, qu1, d2, qu2);
      dV1 += (acc_t)dqu;
      dV2 -= (acc_t)((real)0.5 * fma_(rho, dot3_(d0, d0, d1, d1, d2, d2), dqu));
      real acc = b1 + b2;
      real sy = 0, kk = 0;
      for (int c = 0; c < 3; ++c) {
        sy += fma_(hi[c], kj[c], hj[c] * ki[c]);
        kk = fma_(ki[c], kj[c], kk);
      }
      acc += dmm_((real)0.5, sy, rho, kk);
      role_store(lds, s4_o1, L_SINK, acc);
      role_store(lds, s4_o2, L_SINK, acc);
    }
    TSAT_SYNC_LDS();
  }
  BwdOut<real> out;
  out.dV1 = dV1;
  out.dV2 = dV2;
  out.pd_ok = pd_ok ? 1 : 0;
  return out;
}

// --------------------------------------------------------------------------------------------------
// Row-oriented Riccati recursion (every build of the SOLVE kernel; the tracking kernel keeps riccati_chunk).
// One trajectory per 16-lane ROW of the wavefront: lane j of the row owns COLUMN j of F = [A|B] (state columns j < NH, control
// columns NH .. NH + 2) and column j of the cost-to-go S~ = [S; s'] (S~[r][j], r <= NH, in registers). Nothing goes through LDS
// between the steps of a knot: a value another lane holds is read as the DPP operand `row_newbcast:n` of the FMA that consumes
// it (v_fmac_f64_dpp: src0 = lane n of the reader's row; issue cost and latency of a plain v_fma_f64,
// profiles/r04/valu_f64_ubench.txt), against ~90 cycles for an LDS write -> read exchange, of which riccati_chunk pays four per
// knot and the packed builds' former column recursion three. The one-trajectory builds run the same instructions with the
// four rows holding the same trajectory; the packed builds run four trajectories at once (tsat_packed.hpp).
//   step 1  W[r]  = sum_m S~[r][m] F[m][j]                r <= NH; row NH starts from l_j = [lx; lu]_j: W[NH] = Qx_j | Qu_b
//   step 2  Q[i]  = lxx(i,j) | luu + sum_m F[m][i] W[m]   i < NH + 3: Qxx(:, j), Qux(:, j) on a state lane, Quu(:, b) on a control lane
//   step 3  Quu, Qu to every lane (v_mov_b64_dpp); regularise, Sylvester test, adjugate inverse, K(:, j), d (every lane)
//   step 4  S(i,j) = Qxx(i,j) + sum_c T[c][i] K[c][j],  s(j) = Qx_j + sum_c T[c][j] d[c],  T = Qux - rho K
//           (= Qxx + Qux'K - rho K'K: Quu K = -Qux - rho K by the definition of K)
// S is formed column by column (both triangles) and READ through its upper triangle only — S~[r][m] with r > m is taken as row m
// of lane r's column, the same DPP read with the roles of register and lane exchanged — so the cost-to-go the recursion works
// with is symmetric to the last bit. The arithmetic is double whatever the storage type of the records.
// --------------------------------------------------------------------------------------------------
#ifndef TSAT_EMU
#include "tsat_riccati_dpp.inc"
#endif
template <int NH> struct RowState { double Sc[NH + 1]; };
#ifdef TSAT_EMU
// emulator: the DPP reads of a knot become three exchanges through scratch blocks — every lane publishes the registers the
// following blocks read across lanes, ONE barrier, then the blocks' arithmetic in the GPU's operation order. Three blocks take
// turns (b = 0, 1, 2): a lane can publish into block b again only after the barriers of the two other exchanges, which every
// lane reaches only when it has finished reading block b — no second barrier per exchange.
constexpr int XCH_W = 16, XCH_BLOCK = 64 * XCH_W;
TSAT_DEV double* xch_mine(int b) { return tsat_emu::xch() + b * XCH_BLOCK + TSAT_LANE() * XCH_W; }
TSAT_DEV const double* xch_of(int b, int row_lane) { return tsat_emu::xch() + b * XCH_BLOCK + ((TSAT_LANE() & ~15) + row_lane) * XCH_W; }
#endif

// per-lane constants of riccati_row_step for lane j of a row; Qd: the trajectory's stage weights.
//   cj: the column of F this lane reads (j clamped to the record's columns: lanes beyond them repeat column 0 and are never read);
//   qc[i]: lxx(i, j) for the rows whose stage Hessian is a constant of the trajectory (error coordinates: the three rate rows);
//   mq, mu_[a]: 1 on the lanes that own an entry of G'QG / luu(a), else 0; qq_off[a]: where G'QG(a, j - 3) sits in the record
template <int NH, typename R>
struct RowRoles {
  int cj, qq_off[3];
  double qc[NH], mq, mu_[3];
  template <typename real>
  TSAT_DEV void set(int j, const real* Qd) {
    constexpr int ES = (NH == 6) ? 1 : 0, NC = NH + 3;
    cj = (j < NC) ? j : 0;
    for (int i = 0; i < NH; ++i) qc[i] = (j == i && (!ES || i < 3)) ? (double)Qd[i] : 0.0;
    const bool att = ES && j >= 3 && j < NH;
    mq = att ? 1.0 : 0.0;
    for (int a = 0; a < 3; ++a) {
      // G'QG upper triangle (0,0)(0,1)(0,2)(1,1)(1,2)(2,2): entry (min, max) of (a, j - 3)
      const int b = att ? j - 3 : 0, lo = (a < b) ? a : b, hi = (a < b) ? b : a;
      qq_off[a] = R::QQ + (lo == 0 ? hi : (lo == 1 ? 2 + hi : 5));
      mu_[a] = (j == NH + a) ? 1.0 : 0.0;
    }
  }
};
// what a lane reads from its row's knot record (LDS; layout R, storage type real): its column of F, l_j = [lx; lu]_j (contiguous in
// the record), the entries of the stage Hessian its column starts from (before the masks)
template <int NH> struct RowIn { double f[NH], l, qq[3], luu[3]; };
template <typename real, int NH, typename R>
TSAT_DEV RowIn<NH> row_load(const real* rc, const RowRoles<NH, R>& ro) {
  constexpr int ES = (NH == 6) ? 1 : 0;
  RowIn<NH> in;
  if constexpr (ES && sizeof(real) == 8) {          // 48-byte columns: three 16-byte reads
    typedef real v2 __attribute__((vector_size(16)));
    const v2* q = reinterpret_cast<const v2*>(rc + ro.cj * R::FSR);
    for (int t = 0; t < NH / 2; ++t) { const v2 v = q[t]; in.f[2 * t] = (double)v[0]; in.f[2 * t + 1] = (double)v[1]; }
  } else if constexpr (ES && sizeof(real) == 4) {   // 24-byte columns: three 8-byte reads
    typedef real v2 __attribute__((vector_size(8)));
    const v2* q = reinterpret_cast<const v2*>(rc + ro.cj * R::FSR);
    for (int t = 0; t < NH / 2; ++t) { const v2 v = q[t]; in.f[2 * t] = (double)v[0]; in.f[2 * t + 1] = (double)v[1]; }
  } else {
    for (int m = 0; m < NH; ++m) in.f[m] = (double)rc[ro.cj * R::FSR + m];
  }
  in.l = (double)rc[R::LX + ro.cj];
  for (int a = 0; a < 3; ++a) { in.qq[a] = ES ? (double)rc[ro.qq_off[a]] : 0.0; in.luu[a] = (double)rc[R::LUU + a]; }
  return in;
}

// One knot of one row. Returns the Sylvester test (row-uniform); Kc: column j of K (state lanes), d: the feed-forward (every lane).
template <int NH, typename R>
TSAT_DEV bool riccati_row_step(RowState<NH>& st, const RowIn<NH>& in, const RowRoles<NH, R>& ro, double rho, double Kc[3], double d[3],
                               acc_t& dV1, acc_t& dV2) {
  constexpr int ES = (NH == 6) ? 1 : 0, NC = NH + 3;
  double f[NH], W[NH + 1], Q[NC], qu[3], h00, h01, h02, h11, h12, h22;
  for (int m = 0; m < NH; ++m) f[m] = in.f[m];
  W[NH] = in.l;
  // stage Hessian entries this lane's column starts from
  for (int i = 0; i < NH; ++i) {
    if (ES && i >= 3) Q[i] = in.qq[i - 3] * ro.mq;   // G'QG(i - 3, j - 3) on the attitude lanes
    else Q[i] = ro.qc[i];
  }
  for (int a = 0; a < 3; ++a) Q[NH + a] = in.luu[a] * ro.mu_[a];
#ifdef TSAT_EMU
  {
    double* x = xch_mine(0);
    for (int r = 0; r <= NH; ++r) x[r] = st.Sc[r];
    for (int m = 0; m < NH; ++m) x[NH + 1 + m] = f[m];
    tsat_emu::sync();
    for (int r = 0; r < NH; ++r) W[r] = 0;
    for (int m = 0; m < NH; ++m)
      for (int r = 0; r <= NH; ++r) W[r] = fma_((r < NH && r > m) ? xch_of(0, r)[m] : xch_of(0, m)[r], f[m], W[r]);
    for (int m = 0; m < NH; ++m)
      for (int i = 0; i < NC; ++i) Q[i] = fma_(xch_of(0, i)[NH + 1 + m], W[m], Q[i]);
    double* y = xch_mine(1);
    for (int a = 0; a < 3; ++a) y[a] = Q[NH + a];
    y[3] = W[NH];
    tsat_emu::sync();
    h00 = xch_of(1, NH)[0]; h01 = xch_of(1, NH + 1)[0]; h02 = xch_of(1, NH + 2)[0];
    h11 = xch_of(1, NH + 1)[1]; h12 = xch_of(1, NH + 2)[1]; h22 = xch_of(1, NH + 2)[2];
    for (int a = 0; a < 3; ++a) qu[a] = xch_of(1, NH + a)[3];
  }
#else
  if constexpr (NH == 6) { TSAT_RB1_6(); TSAT_RB2_6(); TSAT_RB3_6(); }
  else { TSAT_RB1_7(); TSAT_RB2_7(); TSAT_RB3_7(); }
#endif
  // step 3 (every lane; the operations of riccati_chunk's step 3)
  const double q00 = h00 + rho, q11 = h11 + rho, q22 = h22 + rho;
  const double q10 = h01, q20 = h02, q21 = h12;
  const double c00 = dmm_(q11, q22, q21, q21);
  const double c01 = dmm_(q20, q21, q10, q22);
  const double c02 = dmm_(q10, q21, q20, q11);
  const double c11 = dmm_(q00, q22, q20, q20);
  const double c12 = dmm_(q10, q20, q00, q21);
  const double c22 = dmm_(q00, q11, q10, q10);
  const double det = dot3_(q00, c00, q10, c01, q20, c02);
  const bool pd = (q00 > 0 && c22 > 0 && det > 0);
  const double nid = -rcp_(det);
  const double i00 = c00 * nid, i01 = c01 * nid, i02 = c02 * nid, i11 = c11 * nid, i12 = c12 * nid, i22 = c22 * nid;
  Kc[0] = dot3_(i00, Q[NH], i01, Q[NH + 1], i02, Q[NH + 2]);
  Kc[1] = dot3_(i01, Q[NH], i11, Q[NH + 1], i12, Q[NH + 2]);
  Kc[2] = dot3_(i02, Q[NH], i12, Q[NH + 1], i22, Q[NH + 2]);
  d[0] = dot3_(i00, qu[0], i01, qu[1], i02, qu[2]);
  d[1] = dot3_(i01, qu[0], i11, qu[1], i12, qu[2]);
  d[2] = dot3_(i02, qu[0], i12, qu[1], i22, qu[2]);
  // step 4. With K = -Quu_reg^-1 Qux: Quu K = -Qux - rho K, so the new cost-to-go is Qxx + Qux'K - rho K'K = Qxx + T'K and
  // Qx + T'd with T = Qux - rho K (column j of T from the lane's own Qux and K columns)
  double Tc[3];
  for (int c = 0; c < 3; ++c) Tc[c] = fma_(-rho, Kc[c], Q[NH + c]);
#ifdef TSAT_EMU
  {
    double* x = xch_mine(2);
    for (int c = 0; c < 3; ++c) x[c] = Tc[c];
    tsat_emu::sync();
    for (int c = 0; c < 3; ++c)
      for (int i = 0; i < NH; ++i) Q[i] = fma_(xch_of(2, i)[c], Kc[c], Q[i]);
  }
#else
  if constexpr (NH == 6) { TSAT_RB4_6(); } else { TSAT_RB4_7(); }
#endif
  double sn = W[NH];
  for (int c = 0; c < 3; ++c) sn = fma_(Tc[c], d[c], sn);
  // dV1 += d'Qu ; dV2 += 0.5 d'Quu d = -0.5 (d'Qu + rho d'd)   (riccati_chunk, step 4)
  const double dqu = dot3_(d[0], qu[0], d[1], qu[1], d[2], qu[2]);
  dV1 += (acc_t)dqu;
  dV2 -= (acc_t)(0.5 * fma_(rho, dot3_(d[0], d[0], d[1], d[1], d[2], d[2]), dqu));
  for (int r = 0; r < NH; ++r) st.Sc[r] = Q[r];
  st.Sc[NH] = sn;
  return pd;
}

// The recursion over one chunk whose records are in LDS at L_REC (one-trajectory builds; last knot first). S~ comes from and
// goes back to L_ST (row stride 9) between chunks: jacobian_chunk, a function of its own, runs in between.
template <typename real, int NH, typename R>
TSAT_PHASE BwdOut<real> riccati_rows(TSAT_GLOBAL real* KDg, int k0_, int nk_, real rho_, acc_t dV1, acc_t dV2) {
  // (arguments of a non-inlined function arrive in VGPRs: the chunk bounds are said to be wave-uniform, the loop below is scalar)
  const int k0 = TSAT_UNIFORM_INT(k0_), nk = TSAT_UNIFORM_INT(nk_);
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE(), j = lane & 15;
  RowRoles<NH, R> ro;
  ro.template set<real>(j, lds + L_TR + P_QD);
  const double rho = (double)rho_;
  double* S64 = st64();
  RowState<NH> st;
  const int js = (j < NH) ? j : 0;
  for (int r = 0; r <= NH; ++r) st.Sc[r] = S64[r * 9 + js];
  // K,d record of a knot: lanes 0 .. 6 store their gain column (zero beyond NH), lane 7 the feed-forward — as Kc km + d dm with
  // per-lane 0 / 1 factors instead of selects; the stores stay in flight (no vmcnt wait in this loop)
  const double km = (lane < NH) ? 1.0 : 0.0, dm = (lane == 7) ? 1.0 : 0.0;
  int slot[3];
  for (int c = 0; c < 3; ++c) slot[c] = (lane < 7) ? (c * 7 + lane) : (21 + c);
  bool pd_ok = true;
  // One knot: the recursion step on the record read a knot ago, then its K,d record to HBM. Returns the Sylvester test.
  auto knot = [&](const RowIn<NH>& in, int l) {
    double Kc[3], d[3];
    const bool pd = riccati_row_step<NH, R>(st, in, ro, rho, Kc, d, dV1, dV2);
    if (lane < 8) {
      TSAT_GLOBAL real* kd = KDg + (size_t)(k0 + l) * KDW;
      for (int c = 0; c < 3; ++c) kd[slot[c]] = (real)fma_(Kc[c], km, d[c] * dm);
    }
    return TSAT_UNIFORM_INT(pd) != 0;      // wave-uniform: the four rows hold the same trajectory
  };
  // The record of knot l - 1 is read while knot l is worked on (one wavefront per SIMD has nobody to hide an LDS latency behind);
  // two knots per turn, two register sets A / B taking turns (a single set would have to be copied every knot).
  auto rec = [&](int l) { return rec_base() + ((l > 0) ? l : 0) * R::RECS; };
  RowIn<NH> A = row_load<jac_t, NH, R>(rec(nk - 1), ro);
  for (int l = nk - 1; l >= 0; l -= 2) {
    const RowIn<NH> B = row_load<jac_t, NH, R>(rec(l - 1), ro);
    TSAT_SCHED_FENCE();
    if (!knot(A, l)) { pd_ok = false; break; }
    if (l - 1 >= 0) {
      A = row_load<jac_t, NH, R>(rec(l - 2), ro);
      TSAT_SCHED_FENCE();
      if (!knot(B, l - 1)) { pd_ok = false; break; }
    }
  }
  TSAT_SYNC_LDS();
  if (lane < NH)
    for (int r = 0; r <= NH; ++r) S64[r * 9 + lane] = st.Sc[r];
  BwdOut<real> out;
  out.dV1 = dV1; out.dV2 = dV2; out.pd_ok = pd_ok ? 1 : 0;
  return out;
}

// terminal cost-to-go into the Riccati scratch S~ (L_ST, row stride 9; row NH = s'): reads the staged trajectory constants
// (L_TR) and terminal multipliers (L_NU)
template <typename real, int ES>
TSAT_DEV void terminal_cost_to_go(const TSAT_GLOBAL real* XUg, int N, real mu, int term_mask) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  struct { const TSAT_GLOBAL real* XU; } p = {XUg};
  // terminal cost-to-go: Sxx = Qf + mu*mask, Sx = Qf e + mask (nu + mu e)   (Appendix A backward); in error
  // coordinates both are projected through E(q_N): E'SxxE, E'Sx (src/quaternion_toolbox.jl:38-50)
  {
    constexpr int NH = BwdCfg<ES>::NH;
    const int r1 = lane & 7, c1 = lane >> 3;
    const TSAT_GLOBAL real* xN = p.XU + (size_t)(N - 1) * XUW;
    if (!ES) {
      if (c1 < 7) {
        real v = 0;
        const real qf = lds[L_TR + P_QFD + c1];
        const real e = xN[c1] - lds[L_TR + P_XF + c1];
        const bool m = (term_mask >> c1) & 1;
        if (r1 < 7) {
          if (r1 == c1) v = qf + (m ? mu : (real)0);
        } else {
          v = fma_(qf, e, m ? fma_(mu, e, lds[L_NU + c1]) : (real)0);   // fused operations spelled out, see dmm_
        }
        lds[L_ST + r1 * 9 + c1] = v;
      }
    } else if (c1 < NH && r1 <= NH) {
      // E(q_N) column c: identity on the rate block, G(q_N) on the attitude block
      const real sq = xN[3], v0 = xN[4], v1 = xN[5], v2 = xN[6];
      const real G[4][3] = {{-v0, -v1, -v2}, {sq, -v2, v1}, {v2, sq, -v0}, {-v1, v0, sq}};
      real acc = 0;
      for (int m = 0; m < 7; ++m) {
        const real qf = lds[L_TR + P_QFD + m];
        const real e = xN[m] - lds[L_TR + P_XF + m];
        const bool msk = (term_mask >> m) & 1;
        const real sd = qf + (msk ? mu : (real)0);
        const real sf = fma_(qf, e, msk ? fma_(mu, e, lds[L_NU + m]) : (real)0);
        // (r, c) taken in the order (min, max) on the matrix rows: the projected S is symmetric to the last bit, whichever
        // triangle a reader takes (the packed build keeps only the upper one)
        const int ra = (r1 < NH && r1 > c1) ? c1 : r1, ca = (r1 < NH && r1 > c1) ? r1 : c1;
        real ec = 0, er = 0;   // E[m][ca], E[m][ra]
        for (int t = 0; t < 3; ++t) {
          const real em = (m < 3) ? ((m == t) ? (real)1 : (real)0) : (real)0;
          const real gm = (m >= 3) ? G[m >= 3 ? m - 3 : 0][t] : (real)0;
          if (ca == t) ec = em;
          if (ca == 3 + t) ec = gm;
          if (ra == t) er = em;
          if (ra == 3 + t) er = gm;
        }
        acc = fma_((r1 < NH) ? er * sd : sf, ec, acc);
      }
      lds[L_ST + r1 * 9 + c1] = acc;
    }
    if (lane == 0) { lds[L_ZERO] = 0; lds[L_SINK] = 0; }
  }
}

template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_PHASE BwdOut<real> backward_sweep(TPtrs<real> p, int N, int n_tab, real mu, real rho, int term_mask) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  terminal_cost_to_go<real, ES>(p.XU, N, mu, term_mask);
  BwdOut<real> acc;
  acc.dV1 = 0; acc.dV2 = 0; acc.pd_ok = 1;
  TSAT_SYNC();
  constexpr int CHB = BwdCfg<ES>::CHB;
  const int nchunks = (N - 1 + CHB - 1) / CHB;
  for (int ch = nchunks - 1; ch >= 0 && acc.pd_ok; --ch) {
    const int k0 = ch * CHB;
    const int nk = (N - 1 - k0 < CHB) ? (N - 1 - k0) : CHB;
    const unsigned long long t_j0 = tick_();
    jacobian_chunk<real, INTEG, DIAGJ, ES>(p, N, n_tab, k0, nk, mu);
    TSAT_SYNC();
    const unsigned long long t_j1 = tick_();
    acc = riccati_rows<real, BwdCfg<ES>::NH, PkRec<ES>>(p.KD, k0, nk, rho, acc.dV1, acc.dV2);
    TSAT_SYNC();
#ifdef TSAT_PROFILE
    if (lane == 0) {
      lds[L_PC + 1] += (real)(t_j1 - t_j0);
      lds[L_PC + 2] += (real)(tick_() - t_j1);
    }
#else
    (void)t_j0; (void)t_j1;
#endif
  }
  return acc;
}

// --------------------------------------------------------------------------------------------------
// parallel-in-k passes
// --------------------------------------------------------------------------------------------------
// AL (or plain LQR) cost of the nominal trajectory
template <typename real>
TSAT_PHASE acc_t nominal_cost(TPtrs<real> p, int N, real mu, int term_mask, int with_al) {
  real* lds = lds_base<real>();
  const Traj<real> tr = load_traj<real>(N, 1, p.bt);
  const int lane = TSAT_LANE();
  real nu[7];
  for (int i = 0; i < 7; ++i) nu[i] = lds[L_NU + i];
  acc_t J = 0;
  for (int k = lane; k < N; k += WAVE) {
    real x[7], u[3], lam[6];
    for (int i = 0; i < 7; ++i) x[i] = p.XU[(size_t)k * XUW + i];
    if (k < N - 1) {
      for (int c = 0; c < 3; ++c) u[c] = p.XU[(size_t)k * XUW + 7 + c];
      for (int c = 0; c < 6; ++c) lam[c] = p.LAM[(size_t)k * LMW + c];
      J += (acc_t)stage_cost(tr, x, u, lam, mu, with_al != 0);
    } else {
      J += (acc_t)term_cost(tr, x, nu, mu, term_mask, with_al != 0);
    }
  }
  (void)lds;
  return wave_sum(J, red64());
}

// max constraint violation of the nominal trajectory; optionally applies the dual update of the control-box
// multipliers in the same pass (Appendix A "solve_AL")
template <typename real>
TSAT_PHASE real violation_and_duals(TPtrs<real> p, int N, real mu, int term_mask, int update, real dmax) {
  real* lds = lds_base<real>();
  const Traj<real> tr = load_traj<real>(N, 1, p.bt);
  const int lane = TSAT_LANE();
  real c = 0;
  for (int k = lane; k < N; k += WAVE) {
    if (k < N - 1) {
      for (int m = 0; m < 3; ++m) {
        const real u = p.XU[(size_t)k * XUW + 7 + m];
        const real chi = u - tr.uhi[m], clo = tr.ulo[m] - u;
        c = fmax_(c, chi);
        c = fmax_(c, clo);
        if (update) {
          real l = p.LAM[(size_t)k * LMW + m] + mu * chi;
          l = l > 0 ? l : (real)0; l = l < dmax ? l : dmax;
          p.LAM[(size_t)k * LMW + m] = l;
          l = p.LAM[(size_t)k * LMW + 3 + m] + mu * clo;
          l = l > 0 ? l : (real)0; l = l < dmax ? l : dmax;
          p.LAM[(size_t)k * LMW + 3 + m] = l;
        }
      }
    } else {
      for (int i = 0; i < 7; ++i)
        if ((term_mask >> i) & 1) c = fmax_(c, fabs_(p.XU[(size_t)k * XUW + i] - tr.xf[i]));
    }
  }
  return wave_max(c, lds + L_RED);
}

// adopt candidate `jw` as the nominal trajectory (or keep it when jw < 0) and return the Todorov gradient
// mean_k max_i |d_k,i| / (|u_k,i| + 1) evaluated with the (new) nominal controls.
template <typename real>
TSAT_PHASE real adopt_and_gradient(TPtrs<real> p, int N, int jw) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  const TSAT_GLOBAL real* Cg = p.CAND + (size_t)(jw >= 0 ? jw : 0) * (size_t)N * XUW;
  real g = 0;
  for (int k = lane; k < N; k += WAVE) {
    real r[10];
    const TSAT_GLOBAL real* src = (jw >= 0) ? (Cg + (size_t)k * XUW) : (p.XU + (size_t)k * XUW);
    for (int i = 0; i < 10; ++i) r[i] = src[i];
    if (jw >= 0)
      for (int i = 0; i < 10; ++i) p.XU[(size_t)k * XUW + i] = r[i];
    if (k < N - 1) {
      real m = 0;
      for (int c = 0; c < 3; ++c) m = fmax_(m, fabs_(p.KD[(size_t)k * KDW + 21 + c]) / (fabs_(r[7 + c]) + (real)1));
      g += m;
    }
  }
  g = wave_sum(g, lds + L_RED);
  return g / (real)(N - 1);
}

// Todorov gradient mean_k max_i |d_k,i| / (|u_k,i| + 1) with the controls of the record slab `XUs` (the nominal trajectory, or
// the candidate about to become it): what adopt_and_gradient returns, without the copy
template <typename real>
TSAT_PHASE real todorov_gradient(const TSAT_GLOBAL real* XUs, const TSAT_GLOBAL real* KDg, int N) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  real g = 0;
  for (int k = lane; k < N - 1; k += WAVE) {
    real m = 0;
    for (int c = 0; c < 3; ++c) m = fmax_(m, fabs_(KDg[(size_t)k * KDW + 21 + c]) / (fabs_(XUs[(size_t)k * XUW + 7 + c]) + (real)1));
    g += m;
  }
  g = wave_sum(g, lds + L_RED);
  return g / (real)(N - 1);
}

// --------------------------------------------------------------------------------------------------
// the whole AL-iLQR solve of one trajectory by one wavefront
// --------------------------------------------------------------------------------------------------
// Where a trajectory stands at the top of an inner iteration (its backward sweep is next): what solve_trajectory needs to carry
// on a solve that another mapping began. The packed builds hand the LAST live trajectory of a wavefront over to this
// one-trajectory mapping (tsat_packed.hpp): every build gives the same bits, so the hand-over changes the time only.
template <typename real>
struct Resume {
  acc_t Jprev;
  real mu, rho, drho, grad, nu[7];
  int outer, it, djz, inner_iters, ls_trials, n_backward, n_forward, bp_restarts, fp_fails, trow;
  int cur, pad;     // slab that holds the nominal trajectory (cand_slab numbering; both mappings use the launch's a.max_ls slots)
};

template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_DEV int solve_trajectory(const KArgs<real>& a, int traj, const Resume<real>* rs = nullptr) {   // returns 1: parked (packed builds' endgame)
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  const tsat_options& o = a.opt;
  // the batch is laid out with the common stride a.N; a trajectory may use fewer knots (variable-horizon sweeps:
  // t_total[i] = 0:0.2:t_final[i], src/monte_carlo.jl:140-145)
  const int NS = a.N, n_tab = a.n_tab;
  const int N = a.nk ? a.nk[traj] : a.N;
  TPtrs<real> p;
  p.XU = (TSAT_GLOBAL real*)(a.XU + (size_t)traj * xu_stride<real>(NS));
  p.KD = (TSAT_GLOBAL real*)(a.KD + (size_t)traj * kd_stride<real>(NS));
  p.LAM = (TSAT_GLOBAL real*)(a.LAM + (size_t)traj * lam_stride<real>(NS));
  p.CAND = (TSAT_GLOBAL real*)(a.CAND + (size_t)traj * a.max_ls * xu_stride<real>(NS));
  p.bt = (const TSAT_GLOBAL real*)(a.BT + (size_t)a.bidx[traj] * n_tab * 4);
  p.XU0 = p.XU; p.cur = 0;
// the stored candidate `slot` of the sweep just evaluated becomes the nominal trajectory: a pointer swap
#define TSAT_ADOPT(slot) do { p.cur = cand_slab(p.cur, (slot)); p.XU = slab_ptr<real>(p, N, p.cur); } while (0)
  stage_traj<real>((const TSAT_GLOBAL real*)(a.P + (size_t)traj * PSTRIDE), (real)o.u_scale);

  const real* U0g = a.U0 + (size_t)traj * u0_stride<real>(NS);
  double* trace = a.trace ? a.trace + (size_t)traj * a.trace_rows * 8 : nullptr;
  int trow = 0;
  real mu = (real)o.penalty_init;
  const real max_state = (real)o.max_state;
  const int tmask = o.terminal_mask;
  int status = TSAT_MAX_OUTER, outer_iters = 0, inner_iters = 0, ls_trials = 0, n_backward = 0, n_forward = 0,
      bp_restarts = 0, fp_fails = 0;
  real grad = 0;
  int last_jw = 0;                  // accepted line-search index of the previous iteration (first iteration: a shallow search is assumed)
  unsigned long long pc_fwd = 0, pc_par = 0;
  bool start_ok = true;

  if (rs) {
    // carry on: the nominal trajectory, gains' inputs and multipliers are in HBM; counters and terminal multipliers come along
    trow = rs->trow; mu = rs->mu; grad = rs->grad;
    p.cur = rs->cur; p.XU = slab_ptr<real>(p, N, p.cur);
    inner_iters = rs->inner_iters; ls_trials = rs->ls_trials; n_backward = rs->n_backward; n_forward = rs->n_forward;
    bp_restarts = rs->bp_restarts; fp_fails = rs->fp_fails;
    if (lane < 7) lds[L_NU + lane] = rs->nu[lane];
    TSAT_SYNC();
  } else {
    // initial_controls!(prob, U0) (src/TortoiseSat.jl:191) + zero multipliers
    for (int k = lane; k < N; k += WAVE) {
      for (int i = 0; i < 7; ++i) p.XU[(size_t)k * XUW + i] = 0;
      for (int c = 0; c < 3; ++c) p.XU[(size_t)k * XUW + 7 + c] = (k < N - 1) ? U0g[(size_t)k * 3 + c] : (real)0;
      if (k < N - 1)
        for (int c = 0; c < 6; ++c) p.LAM[(size_t)k * LMW + c] = 0;
    }
    TSAT_SYNC();
    // open-loop rollout of U0
    forward_sweep<real, INTEG, DIAGJ, ES>(p, N, n_tab, 0, 1, 0);
    n_forward++;
    TSAT_SYNC();
    const FwdOut<real> f0 = candidate_costs<real>(p, N, 0, 1, mu, tmask, max_state);
    const acc_t J0 = wave_bcast(f0.J, 0, red64());
    const int ok0 = wave_first<real>(!f0.ok, lds + L_RED) > 0;  // lane 0 ok?
    TSAT_SYNC();
    TSAT_ADOPT(0);
    start_ok = ok0 && (J0 - J0 == 0);
  }
  if (!start_ok) {
    status = TSAT_DIVERGED;
  } else {
    bool resumed = rs != nullptr;
    for (int outer = rs ? rs->outer : 1; outer <= o.max_outer; ++outer) {
      acc_t Jprev;
      real rho = (real)o.reg_init, drho = 0;
      int djz = 0, it0 = 1;
      if (resumed) {      // the outer iteration was begun by the other mapping: its cost, regularisation and counters stand
        Jprev = rs->Jprev; rho = rs->rho; drho = rs->drho; djz = rs->djz; it0 = rs->it;
        resumed = false;
      } else {
        Jprev = nominal_cost<real>(p, N, mu, tmask, 1);
      }
      bool regfail = false;
      for (int it = it0; it <= o.max_inner; ++it) {
#ifdef TSAT_PACKED
        // A continuation inside a packed launch (hand_over_last) parks for the endgame like the wavefront it came from
        // (tsat_packed.hpp, suspend_if_endgame): the launch that follows cannot begin before this one has drained.
        if (rs && a.suspend_at && a.live) {
          int* seen = reinterpret_cast<int*>(lds + L_RED);
          TSAT_SYNC_LDS();
          if (lane == 0) seen[0] = TSAT_ATOMIC_LOAD(a.live);
          TSAT_SYNC_LDS();
          const int live_now = seen[0];
          TSAT_SYNC_LDS();
          if (live_now <= a.suspend_at) {
            if (lane == 0) {
              const int pos = TSAT_ATOMIC_ADD(a.susp_n, 1);
              a.susp_ids[pos] = traj;
              Resume<real>& r = reinterpret_cast<Resume<real>*>(a.susp_state)[pos];
              r.Jprev = Jprev; r.mu = mu; r.rho = rho; r.drho = drho; r.grad = grad;
              for (int i = 0; i < 7; ++i) r.nu[i] = lds[L_NU + i];
              r.outer = outer; r.it = it; r.djz = djz; r.inner_iters = inner_iters; r.ls_trials = ls_trials;
              r.n_backward = n_backward; r.n_forward = n_forward; r.bp_restarts = bp_restarts; r.fp_fails = fp_fails; r.trow = trow;
              r.cur = p.cur; r.pad = 0;
            }
            TSAT_SYNC();
            return 1;
          }
        }
#endif
        BwdOut<real> bw;
        for (;;) {
          n_backward++;
          bw = backward_sweep<real, INTEG, DIAGJ, ES>(p, N, n_tab, mu, rho, tmask);
          if (bw.pd_ok) break;
          bp_restarts++;
          drho = (drho * (real)o.reg_scale > (real)o.reg_scale) ? drho * (real)o.reg_scale : (real)o.reg_scale;
          rho = (rho * drho > (real)o.reg_min) ? rho * drho : (real)o.reg_min;
          if (rho > (real)o.reg_max) { regfail = true; break; }
        }
        if (regfail) break;
        const acc_t dV1 = bw.dV1, dV2 = bw.dV2;
        const real rho_used = rho;
        {  // regularisation decrease
          const real inv = (real)1 / (real)o.reg_scale;
          drho = (drho / (real)o.reg_scale < inv) ? drho / (real)o.reg_scale : inv;
          const real r = rho * drho;
          rho = (r > (real)o.reg_min) ? r : (real)0;
        }
        TSAT_SYNC();
        // Line search. One sweep rolls out the candidates alpha = 2^-(shift + lane) and keeps the first n_store roll-outs; their
        // costs are then evaluated CG candidates at a time (lanes = knots) until one is accepted — the first accepted index is
        // exactly what sequential backtracking picks, and the usual winner is among the first CG. Candidates beyond the stored
        // ones (max_linesearch > n_store) take another sweep with the next n_store.
        const unsigned long long t_f0 = tick_();
        // How many roll-outs a sweep keeps costs nothing in time (the stores are never waited on) but is most of the launch's HBM
        // writes, so a trajectory whose last accepted step was among the first few keeps only N_FEW at first and takes the rest in
        // a further sweep if it has to; one that has been searching deep keeps all the slots at once. The accepted candidate — and
        // with it every result — is the same either way.
        const int n_slots = (o.max_linesearch < a.max_ls) ? o.max_linesearch : a.max_ls;
        int jw = WAVE, slot = 0;
        acc_t Jw = 0;
        unsigned long long t_cost = 0;
        int n_store = (last_jw >= N_FEW - 1 || n_slots < N_FEW) ? n_slots : N_FEW;
        for (int shift = 0; shift < o.max_linesearch && jw == WAVE; shift += n_store, n_store = n_slots) {
          const int n_here = (o.max_linesearch - shift < n_store) ? o.max_linesearch - shift : n_store;
          forward_sweep<real, INTEG, DIAGJ, ES>(p, N, n_tab, 1, n_here, shift);
          n_forward++;
          TSAT_SYNC();                   // the candidates' records are in HBM before the cost lanes read them
          const unsigned long long t_c0 = tick_();
          for (int c0 = 0; c0 < n_here && jw == WAVE; c0 += CG) {
            const int nc = (n_here - c0 < CG) ? n_here - c0 : CG;
            const FwdOut<real> fw = candidate_costs<real>(p, N, c0, nc, mu, tmask, max_state);
            const int ci = shift + c0 + lane;            // this lane's candidate (lanes < nc)
            acc_t alpha = 1;
            for (int j = 0; j < ci && j < TSAT_MAX_LINESEARCH; ++j) alpha *= (acc_t)0.5;
            const acc_t Jc = fw.J;
            const acc_t expected = -alpha * (dV1 + alpha * dV2);
            const acc_t z = (expected > 0) ? (Jprev - Jc) / expected : (acc_t)-1;
            const bool acc = (lane < nc) && fw.ok && ((z > (acc_t)o.ls_lower && z <= (acc_t)o.ls_upper) || Jc < Jprev);
            const int jl = wave_first<real>(acc, lds + L_RED);
            if (jl < WAVE) {
              jw = shift + c0 + jl;
              slot = c0 + jl;
              Jw = wave_bcast(Jc, jl, red64());
            }
          }
          t_cost += tick_() - t_c0;
        }
        const unsigned long long t_f1 = tick_();
        pc_fwd += (t_f1 - t_f0) - t_cost;
        pc_par += t_cost;
        acc_t J;
        TSAT_SYNC();
        last_jw = (jw < WAVE) ? jw : o.max_linesearch;
        if (jw < WAVE) {
          J = Jw;
          ls_trials += jw + 1;
          TSAT_ADOPT(slot);
          grad = todorov_gradient<real>(p.XU, p.KD, N);
        } else {
          J = Jprev;
          ls_trials += o.max_linesearch;
          fp_fails++;
          drho = (drho * (real)o.reg_scale > (real)o.reg_scale) ? drho * (real)o.reg_scale : (real)o.reg_scale;
          rho = (rho * drho > (real)o.reg_min) ? rho * drho : (real)o.reg_min;
          rho += (real)o.reg_fp;
          grad = todorov_gradient<real>(p.XU, p.KD, N);
        }
        TSAT_SYNC();
        pc_par += tick_() - t_f1;
        acc_t dJ = J - Jprev;
        dJ = dJ < 0 ? -dJ : dJ;
        if (trace && lane == 0 && trow < a.trace_rows) {
          double* r = trace + 8 * trow;
          r[0] = outer; r[1] = it; r[2] = (double)Jprev; r[3] = (double)J; r[4] = (jw < WAVE) ? jw : -1;
          r[5] = (double)rho_used; r[6] = (double)dV1; r[7] = (double)dV2;
#if defined(TSAT_PROFILE) && !defined(TSAT_EMU)
          r[7] = (double)wall_clock64();     // diagnostic build: when (100 MHz counter) the iteration ended (tools/straggler_timeline.py)
#endif
        }
        trow++;
        Jprev = J;
        djz = (dJ == 0) ? djz + 1 : 0;
        inner_iters++;
        if (0 < dJ && dJ < (acc_t)o.cost_tol) break;
        if (grad < (real)o.grad_tol) break;
        if (djz > o.dj_counter_limit) break;
      }
      outer_iters = outer;
      const real cmax = violation_and_duals<real>(p, N, mu, tmask, 0, (real)o.dual_max);
      if (regfail) { status = TSAT_REG_FAIL; break; }
      if (cmax < (real)o.constraint_tol) { status = TSAT_CONVERGED; break; }
      if (outer == o.max_outer) break;
      (void)violation_and_duals<real>(p, N, mu, tmask, 1, (real)o.dual_max);   // dual update of the control box
      if (lane < 7 && ((tmask >> lane) & 1)) {
        const real dmax = (real)o.dual_max;
        real v = lds[L_NU + lane] + mu * (p.XU[(size_t)(N - 1) * XUW + lane] - lds[L_TR + P_XF + lane]);
        v = v > -dmax ? v : -dmax; v = v < dmax ? v : dmax;
        lds[L_NU + lane] = v;
      }
      mu = (mu * (real)o.penalty_scale < (real)o.penalty_max) ? mu * (real)o.penalty_scale : (real)o.penalty_max;
      TSAT_SYNC();
    }
  }
  TSAT_SYNC();
  if (p.cur != 0) {     // the nominal trajectory lives in one of the candidate slabs: bring it home, once per solve
    for (int k = lane; k < N; k += WAVE)
      for (int i = 0; i < XUW; ++i) p.XU0[(size_t)k * XUW + i] = p.XU[(size_t)k * XUW + i];
    p.XU = p.XU0; p.cur = 0;
    TSAT_SYNC();
  }
  const real cmax = violation_and_duals<real>(p, N, mu, tmask, 0, (real)o.dual_max);
  const acc_t cost = nominal_cost<real>(p, N, mu, tmask, 0);
  const acc_t cost_al = nominal_cost<real>(p, N, mu, tmask, 1);
  if (lane == 0) {
    tsat_stats& st = a.stats[traj];
    st.status = status; st.outer_iters = outer_iters; st.inner_iters = inner_iters; st.ls_trials = ls_trials;
    st.n_backward = n_backward; st.n_forward = n_forward; st.bp_restarts = bp_restarts; st.fp_fails = fp_fails;
    st.cost = (double)cost; st.cost_al = (double)cost_al; st.c_max = (double)cmax; st.grad = (double)grad;
#ifdef TSAT_PROFILE
    if (trace && a.trace_rows > 0) {  // diagnostic build only: row 0 carries the phase clocks instead of iteration 1
      trace[0] = (double)pc_fwd; trace[1] = (double)lds[L_PC + 1]; trace[2] = (double)lds[L_PC + 2];
      trace[3] = (double)pc_par; trace[4] = (double)inner_iters; trace[5] = (double)n_backward;
    }
#else
    (void)pc_fwd; (void)pc_par;
#endif
  }
  return 0;
}

// ==================================================================================================
// Closed-loop TVLQR tracking of a solved slew — the reference's attitude_simulation
// (src/attitude_controller.jl:1-119) for a batch: one trajectory per wavefront, reusing the solver's machinery:
//   gains     Jacobian lanes (rk4 over the linearisation step) + in-place G(q) reduction + riccati_chunk<real,6>.
//             The iLQR recursion with zero gradients, Q0 = Q_lqr, luu = R_lqr, S_N = Qf_lqr is the TVLQR Riccati
//             equation in Schur form; its gain is the negative of the reference's K (u = U - K dX).
//   tracking  one sequential rk4 rollout of the (optionally noise-driven) plant, src/simulator.jl
//   statistic first sample with |w| < w_tol and error angle < angle_tol (src/monte_carlo.jl:242-262), lanes = samples
// ==================================================================================================
constexpr int P_QATT = 58;   // spare slots of the parameter record: attitude weights of Q_lqr (3)

template <typename real>
struct TvArgs {
  int T, N, n_tab, lin_sq, min_steps;
  real us, w_tol, ang_tol;
  const real* P;      // [T][PSTRIDE]
  const real* BT;     // [n_btab][n_tab][4]
  const int* bidx;    // [T]
  const int* nk;      // [T] per-trajectory knot counts or null (all N)
  const real* XUR;    // [T][N][10]   optimised (X,U) records
  const real* NZ;     // [T][N-1][4][9] noise or null
  int noise_mode;     // 1: draw the noise in the kernel (Philox4x32-10, see include/tortoise_hip.h)
  unsigned k0, k1;    // generator key
  const long long* nid;   // [T] generator ids or null (id = trajectory index)
  int rate_as_written;    // statistic: the rate of sample `id` for every j (src/monte_carlo.jl:247 as written)
  real sg, sa, fa;    // sigma_gyro, sigma_att, field_amp
  real* KD;           // [T][N-1][24] gains (solver sign) + unused d
  real* XS;           // [T][N][10]   simulated (x,u) records
  tsat_tvlqr_stats* stats;
};

// Philox4x32-10 (Salmon et al., SC'11): ten rounds of two 32x32->64 multiplies on the counter, key bumped by the Weyl
// constants between rounds
TSAT_DEV void philox4x32_10(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned out[4]) {
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
template <typename real> TSAT_DEV real unit_open(unsigned w) { return ((real)w + (real)0.5) * (real)(1.0 / 4294967296.0); }
template <typename real> TSAT_DEV void box_muller(unsigned a, unsigned b, real& z0, real& z1) {
  const real r = sqrt_(-2 * log_(unit_open<real>(a))), th = (real)6.283185307179586476925 * unit_open<real>(b);
  z0 = r * cos_(th); z1 = r * sin_(th);
}
// the nine draws of one plant evaluation (include/tortoise_hip.h, noise_mode = 1)
template <typename real>
TSAT_DEV void plant_noise(unsigned k0, unsigned k1, long long id, int knot, int stage, real sg, real sa, real fa, real nz[9]) {
  unsigned w0[4], w1[4], w2[4];
  const unsigned ilo = (unsigned)((unsigned long long)id & 0xFFFFFFFFull), ihi = (unsigned)((unsigned long long)id >> 32);
  philox4x32_10(k0, k1, ilo, ihi, (unsigned)knot, (unsigned)(4 * stage + 0), w0);
  philox4x32_10(k0, k1, ilo, ihi, (unsigned)knot, (unsigned)(4 * stage + 1), w1);
  philox4x32_10(k0, k1, ilo, ihi, (unsigned)knot, (unsigned)(4 * stage + 2), w2);
  real z[6];
  box_muller<real>(w0[0], w0[1], z[0], z[1]);
  box_muller<real>(w0[2], w0[3], z[2], z[3]);
  box_muller<real>(w1[0], w1[1], z[4], z[5]);
  for (int i = 0; i < 3; ++i) { nz[i] = sg * z[i]; nz[3 + i] = sa * z[3 + i]; }
  nz[6] = fa * unit_open<real>(w1[2]); nz[7] = fa * unit_open<real>(w1[3]); nz[8] = fa * unit_open<real>(w2[0]);
}

// plant with the reference's three noise sources injected (src/simulator.jl:5-23); noisy == false: src/gain_simulator.jl
template <typename real, int DIAGJ>
TSAT_DEV void dyn_sim_h(const Traj<real>& tr, const real x[7], const real us[3], const real b[3],
                        bool noisy, const real nz[9], real k[7]) {
  real xx[7], bb[3];
  for (int i = 0; i < 7; ++i) xx[i] = x[i];
  for (int i = 0; i < 3; ++i) bb[i] = b[i];
  if (noisy) {
    const real rn = rsqrt_<real>(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
    const real q0 = x[3] * rn, q1 = x[4] * rn, q2 = x[5] * rn, q3 = x[6] * rn;
    const real n0 = nz[3], n1 = nz[4], n2 = nz[5];
    const real th = sqrt_(n0 * n0 + n1 * n1 + n2 * n2);
    const real sh = sin_((real)0.5 * th) / th, ch = cos_((real)0.5 * th);
    const real d1 = n0 * sh, d2 = n1 * sh, d3 = n2 * sh;
    for (int i = 0; i < 3; ++i) { xx[i] = x[i] + nz[i]; bb[i] = b[i] + nz[6 + i]; }
    xx[3] = q0 * ch - (q1 * d1 + q2 * d2 + q3 * d3);          // qmult(q, [cos; r sin])
    xx[4] = q0 * d1 + ch * q1 + (q2 * d3 - q3 * d2);
    xx[5] = q0 * d2 + ch * q2 + (q3 * d1 - q1 * d3);
    xx[6] = q0 * d3 + ch * q3 + (q1 * d2 - q2 * d1);
  }
  StageBase<real> sb;
  dyn_h<real, DIAGJ>(tr, xx, us, bb, k, sb);   // normalises again inside: exact for the noise-free plant, 1 ulp otherwise
}

template <typename real, int DIAGJ>
TSAT_PHASE void tv_jacobian_chunk(TPtrs<real> p, int N, int n_tab, int k0, int nk, real hl, real frac) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  if (lane < nk) {
    Traj<real> tr = load_traj<real>(N, n_tab, p.bt);
    tr.h = hl; tr.hh = (real)0.5 * hl;
    for (int i = 0; i < 9; ++i) tr.hJi[i] = hl * lds[L_TR + P_JI + i];
    tr.usj = tr.us * tr.hJi[0];
    const int k = k0 + lane;
    const TSAT_GLOBAL real* xu = p.XU + (size_t)k * XUW;
    real x[7], u[3], b0[3], b1[3], b2[3];
    for (int i = 0; i < 7; ++i) x[i] = xu[i];
    for (int c = 0; c < 3; ++c) u[c] = xu[7 + c];
    const TSAT_GLOBAL real* p0 = tr.bt + (size_t)brow_index(tr, k, 0.0) * 4;
    const TSAT_GLOBAL real* p1 = tr.bt + (size_t)brow_index(tr, k, 0.5 * (double)frac) * 4;
    const TSAT_GLOBAL real* p2 = tr.bt + (size_t)brow_index(tr, k, (double)frac) * 4;
    for (int c = 0; c < 3; ++c) { b0[c] = p0[c]; b1[c] = p1[c]; b2[c] = p2[c]; }
    constexpr int R_LX = TvRec::LX, R_LU = TvRec::LU, R_LUU = TvRec::LUU, R_QQ = TvRec::QQ;
    real* rc = lds + L_REC + lane * TvRec::RECS;
    real* F = rc;
    rk_jacobian<real, 4, DIAGJ, 1>(tr, x, u, b0, b1, b2, F);
    real qk[4], qn[4];
    for (int i = 0; i < 4; ++i) { qk[i] = x[3 + i]; qn[i] = xu[XUW + 3 + i]; }
    for (int i = 0; i < 7; ++i) {
      real o[3];
      gt_apply(qk, F[3 * FS + i], F[4 * FS + i], F[5 * FS + i], F[6 * FS + i], o);
      F[3 * FS + i] = o[0]; F[4 * FS + i] = o[1]; F[5 * FS + i] = o[2];
    }
    for (int a = 0; a < 3; ++a)
      for (int i = 0; i < 7; ++i) F[(6 + a) * FS + i] = F[(7 + a) * FS + i];
    for (int c = 0; c < 9; ++c) {
      real o[3];
      gt_apply(qn, F[c * FS + 3], F[c * FS + 4], F[c * FS + 5], F[c * FS + 6], o);
      F[c * FS + 3] = o[0]; F[c * FS + 4] = o[1]; F[c * FS + 5] = o[2];
    }
    for (int i = 0; i < 7; ++i) rc[R_LX + i] = 0;
    for (int c = 0; c < 3; ++c) { rc[R_LU + c] = 0; rc[R_LUU + c] = lds[L_TR + P_RD + c]; }
    rc[R_QQ + 0] = lds[L_TR + P_QATT + 0]; rc[R_QQ + 1] = 0; rc[R_QQ + 2] = 0;
    rc[R_QQ + 3] = lds[L_TR + P_QATT + 1]; rc[R_QQ + 4] = 0; rc[R_QQ + 5] = lds[L_TR + P_QATT + 2];
  }
}

template <typename real>
TSAT_DEV real wave_min(real v, real* red) {
  const int lane = TSAT_LANE();
  for (int s = 1; s < WAVE; s <<= 1) {
    red[lane] = v;
    TSAT_SYNC_LDS();
    real o = red[lane ^ s];
    v = (o < v) ? o : v;
    TSAT_SYNC_LDS();
  }
  return v;
}

template <typename real, int DIAGJ>
TSAT_DEV void tvlqr_trajectory(const TvArgs<real>& a, int traj) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  const int NS = a.N, n_tab = a.n_tab;            // slab stride; own horizon below (as solve_trajectory)
  const int N = a.nk ? a.nk[traj] : a.N;
  TPtrs<real> p;
  p.XU = (TSAT_GLOBAL real*)(a.XUR + (size_t)traj * NS * XUW);
  p.KD = (TSAT_GLOBAL real*)(a.KD + (size_t)traj * (NS - 1) * KDW);
  p.LAM = nullptr; p.CAND = nullptr; p.XU0 = p.XU; p.cur = 0;
  p.bt = (const TSAT_GLOBAL real*)(a.BT + (size_t)a.bidx[traj] * n_tab * 4);
  stage_traj<real>((const TSAT_GLOBAL real*)(a.P + (size_t)traj * PSTRIDE), a.us);   // whole record, P_QATT slots included
  TSAT_SYNC();
  const real h = lds[L_TR + P_DT];
  const real hl = a.lin_sq ? h * h : h;
  const real frac = hl / h;
  // ---- gains: terminal S = Qf_lqr, then chunks of Jacobian lanes + Riccati ----------------------------
  {
    const int r1 = lane & 7, c1 = lane >> 3;
    if (c1 < 6 && r1 <= 6) lds[L_ST + r1 * 9 + c1] = (r1 == c1) ? lds[L_TR + P_QFD + c1] : (real)0;
    if (lane == 0) { lds[L_ZERO] = 0; lds[L_SINK] = 0; }
  }
  TSAT_SYNC();
  constexpr int CHB = TV_CHB;
  BwdOut<real> acc;
  acc.dV1 = 0; acc.dV2 = 0; acc.pd_ok = 1;
  for (int ch = (N - 1 + CHB - 1) / CHB - 1; ch >= 0 && acc.pd_ok; --ch) {
    const int k0 = ch * CHB;
    const int nk = (N - 1 - k0 < CHB) ? (N - 1 - k0) : CHB;
    tv_jacobian_chunk<real, DIAGJ>(p, N, n_tab, k0, nk, hl, frac);
    TSAT_SYNC();
    acc = riccati_chunk<real, 6, TvRec>(p.KD, k0, nk, (real)0, acc.dV1, acc.dV2);
    TSAT_SYNC();
  }
  // ---- tracking: x_sim(k+1) = rk4(plant)(x_sim(k), U(k) - K(k) dX(k))   (src/attitude_controller.jl:39-45) -----
  const Traj<real> tr = load_traj<real>(N, n_tab, p.bt);
  TSAT_GLOBAL real* XSg = (TSAT_GLOBAL real*)(a.XS + (size_t)traj * NS * XUW);
  const TSAT_GLOBAL real* NZg = a.NZ ? (const TSAT_GLOBAL real*)(a.NZ + (size_t)traj * (NS - 1) * 36) : nullptr;
  const bool gen = a.noise_mode == 1, noisy = gen || NZg != nullptr;
  const long long gid = a.nid ? a.nid[traj] : (long long)traj;
  constexpr int NZK = WAVE / 4;            // knots per generated chunk: lanes = (knot, RK4 stage) pairs
  real* nzl = lds + L_UNION;               // [NZK][4][9]; the Jacobian records that lived here are spent
  // reference records, gains and field rows of CK knots at a time through the forward sweep's chunk buffer
  // (global_load_lds, one exposed memory latency per chunk instead of one per knot); array-mode noise rides along
  real* fb = nzl + NZK * 36;
  real* nzc = fb + FB_SIZE;                // [CK][4][9] noise of the chunk (array mode)
  static_assert(L_UNION + NZK * 36 + FB_SIZE + CK * 36 <= LDS_REALS + 0 * (int)sizeof(real),
                "tracking buffers fit the wave's LDS block (wide build only; the dense build has the solve kernel alone)");
  real x[7];
  for (int i = 0; i < 7; ++i) x[i] = lds[L_TR + P_X0 + i];
  for (int k = 0; k < N - 1; ++k) {
    if ((k % CK) == 0) {
      const int nk = (N - 1 - k < CK) ? (N - 1 - k) : CK;
      TSAT_SYNC_LDS();
      fwd_chunk_issue<real>(fb, p.KD, p.XU, tr, k, nk, 1);
      if (NZg) {
        const int n2 = (nk * 36) >> 1;
        for (int j = 0; j < (CK * 36 + GLDS - 1) / GLDS; ++j) {
          const int i = lane + WAVE * j, ic = (i < n2) ? i : n2 - 1;
          glds_put<real>(nzc + 2 * i, NZg + (size_t)k * 36 + 2 * ic);
        }
      }
      TSAT_SYNC();
    }
    if (gen && (k % NZK) == 0) {
      TSAT_SYNC();
      const int kk = k + (lane >> 2);
      if (kk < N - 1) plant_noise<real>(a.k0, a.k1, gid, kk, lane & 3, a.sg, a.sa, a.fa, nzl + lane * 9);
      TSAT_SYNC();
    }
    const int kc = k % CK;
    const real* xr = fb + FB_XU + kc * XUW;
    const real* kd = fb + FB_KD + kc * KDW;
    real dX[6];
    for (int i = 0; i < 3; ++i) dX[i] = x[i] - xr[i];
    {  // vector part of q_ref^-1 (x) q_sim  (:42)
      const real s1 = xr[3], a1 = -xr[4], a2 = -xr[5], a3 = -xr[6];
      dX[3] = s1 * x[4] + x[3] * a1 + (a2 * x[6] - a3 * x[5]);
      dX[4] = s1 * x[5] + x[3] * a2 + (a3 * x[4] - a1 * x[6]);
      dX[5] = s1 * x[6] + x[3] * a3 + (a1 * x[5] - a2 * x[4]);
    }
    real u[3];
    for (int c = 0; c < 3; ++c) {
      real v = xr[7 + c];
      for (int j = 0; j < 6; ++j) v += kd[c * 7 + j] * dX[j];   // kd = -K_lqr
      u[c] = v;
    }
    if (lane == 0) {
      for (int i = 0; i < 7; ++i) XSg[(size_t)k * XUW + i] = x[i];
      for (int c = 0; c < 3; ++c) XSg[(size_t)k * XUW + 7 + c] = u[c];
    }
    const real* ba = fb + FB_BA + kc * 6;
    const real* bb = fb + FB_BB + kc * 6;
    const real b0[3] = {ba[0], ba[1], bb[0]}, b1[3] = {ba[2], ba[3], bb[2]}, b2[3] = {ba[4], ba[5], bb[4]};
    real nz[36];
    if (gen) {
      for (int i = 0; i < 36; ++i) nz[i] = nzl[(k % NZK) * 36 + i];
    } else if (NZg) {
      for (int i = 0; i < 36; ++i) nz[i] = nzc[kc * 36 + i];
    } else {
      for (int i = 0; i < 36; ++i) nz[i] = 0;
    }
    const real cs = control_scale<real, DIAGJ>(tr);
    const real us[3] = {u[0] * cs, u[1] * cs, u[2] * cs};
    real k1[7], k2[7], k3[7], k4[7], t[7];
    dyn_sim_h<real, DIAGJ>(tr, x, us, b0, noisy, nz, k1);
    for (int i = 0; i < 7; ++i) t[i] = x[i] + (real)0.5 * k1[i];
    dyn_sim_h<real, DIAGJ>(tr, t, us, b1, noisy, nz + 9, k2);
    for (int i = 0; i < 7; ++i) t[i] = x[i] + (real)0.5 * k2[i];
    dyn_sim_h<real, DIAGJ>(tr, t, us, b1, noisy, nz + 18, k3);
    for (int i = 0; i < 7; ++i) t[i] = x[i] + k3[i];
    dyn_sim_h<real, DIAGJ>(tr, t, us, b2, noisy, nz + 27, k4);
    for (int i = 0; i < 7; ++i) x[i] = x[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) * (real)(1.0 / 6.0);
  }
  if (lane == 0) {
    for (int i = 0; i < 7; ++i) XSg[(size_t)(N - 1) * XUW + i] = x[i];
    for (int c = 0; c < 3; ++c) XSg[(size_t)(N - 1) * XUW + 7 + c] = 0;
  }
  TSAT_SYNC();
  // ---- statistic (src/monte_carlo.jl:242-262): lanes = samples ------------------------------------------
  real first = (real)(N + 1), wN = 0, angN = 0;
  const real qf0 = tr.xf[3], qf1 = -tr.xf[4], qf2 = -tr.xf[5], qf3 = -tr.xf[6];
  // `omega_norm_vec[j] = norm(sim_states[i][1:3,i])` (src/monte_carlo.jl:247) indexes the TRIAL where the sample is meant:
  // rate_as_written reproduces it (sample i = id + 1, clamped to the trajectory), the default takes sample j
  real w_trial = 0;
  if (a.rate_as_written) {
    const long long id = a.nid ? a.nid[traj] : (long long)traj;
    const int ji = (id < 0) ? 0 : (id > (long long)(N - 1) ? N - 1 : (int)id);
    const TSAT_GLOBAL real* xi = XSg + (size_t)ji * XUW;
    w_trial = sqrt_(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
  }
  for (int j = 1 + lane; j <= N; j += WAVE) {
    const TSAT_GLOBAL real* xs = XSg + (size_t)(j - 1) * XUW;
    const real wj = sqrt_(xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2]);
    const real wn = a.rate_as_written ? w_trial : wj;
    const real e0 = qf0 * xs[3] - (qf1 * xs[4] + qf2 * xs[5] + qf3 * xs[6]);
    const real ang = 2 * acos_(e0 < 1 ? e0 : (real)1);
    if (j > a.min_steps && wn < a.w_tol && ang < a.ang_tol && (real)j < first) first = (real)j;
    if (j == N) { wN = wj; angN = ang; }
  }
  first = wave_min(first, lds + L_RED);
  const int srcN = (N - 1) % WAVE;
  wN = wave_bcast(wN, srcN, lds + L_RED);
  angN = wave_bcast(angN, srcN, lds + L_RED);
  if (lane == 0) {
    tsat_tvlqr_stats& st = a.stats[traj];
    const bool ok = first <= (real)N;
    st.slew_index = ok ? (int)first : 0;
    st.failed = ok ? 0 : 1;
    st.slew_time = (double)h * (ok ? (double)first : (double)N);
    st.final_w_norm = (double)wN;
    st.final_angle = (double)angN;
  }
}

// ==================================================================================================
// Receding-horizon step (BASELINE.json configs[4]; NOT in the reference, which tracks its plan with
// TVLQR): after a solve of the resident batch, record (x_t, u_t = U[0]), advance the noise-free plant one step with
// the applied control (rk3 / rk4 of the model dynamics), make the plan shifted by one knot (last control repeated)
// the next warm start and move the table clock on by one knot. Lanes = knots for the shift; lane 0 steps the plant.
// ==================================================================================================
template <typename real>
struct MpcArgs {
  int T, N, n_tab, plant_integ, step, n_steps;
  real us;
  real* P;            // [T][PSTRIDE]   x0 and tau0 are advanced in place
  const real* BT;     // [n_btab][n_tab][4]
  const int* bidx;    // [T]
  const int* nk;      // [T] or null
  const real* XU;     // [T][N][10]     the plan just solved
  real* U0;           // [T][N-1][3]    next warm start
  real* HX;           // [T][n_steps+1][7]
  real* HU;           // [T][n_steps][3]
  const tsat_stats* stats;   // [T] statistics of the solve that has just finished
  long long* tally;   // [T][4] running sums over the control steps: n_backward, n_forward, outer_iters - 1, inner_iters
};

template <typename real, int DIAGJ>
TSAT_DEV void mpc_advance_trajectory(const MpcArgs<real>& a, int traj) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  const int NS = a.N;
  const int N = a.nk ? a.nk[traj] : a.N;
  TSAT_GLOBAL real* Pg = (TSAT_GLOBAL real*)(a.P + (size_t)traj * PSTRIDE);
  stage_traj<real>(Pg, a.us);
  TSAT_SYNC();
  const Traj<real> tr = load_traj<real>(N, a.n_tab, (const TSAT_GLOBAL real*)(a.BT + (size_t)a.bidx[traj] * a.n_tab * 4));
  const TSAT_GLOBAL real* XUg = (const TSAT_GLOBAL real*)(a.XU + (size_t)traj * NS * XUW);
  TSAT_GLOBAL real* U0g = (TSAT_GLOBAL real*)(a.U0 + (size_t)traj * (NS - 1) * 3);
  for (int k = lane; k < N - 1; k += WAVE) {
    const int src = (k + 1 < N - 1) ? k + 1 : N - 2;
    for (int c = 0; c < 3; ++c) U0g[(size_t)k * 3 + c] = XUg[(size_t)src * XUW + 7 + c];
  }
  if (lane == 0) {
    real x[7], u[3], xn[7], b0[3], b1[3], b2[3];
    for (int i = 0; i < 7; ++i) x[i] = lds[L_TR + P_X0 + i];
    for (int c = 0; c < 3; ++c) u[c] = XUg[7 + c];
    const TSAT_GLOBAL real* p0 = tr.bt + (size_t)brow_index(tr, 0, 0.0) * 4;
    const TSAT_GLOBAL real* p1 = tr.bt + (size_t)brow_index(tr, 0, 0.5) * 4;
    const TSAT_GLOBAL real* p2 = tr.bt + (size_t)brow_index(tr, 0, 1.0) * 4;
    for (int c = 0; c < 3; ++c) { b0[c] = p0[c]; b1[c] = p1[c]; b2[c] = p2[c]; }
    if (a.plant_integ == 4) rk_step<real, 4, DIAGJ, 0>(tr, x, u, b0, b1, b2, xn);
    else rk_step<real, 3, DIAGJ, 0>(tr, x, u, b0, b1, b2, xn);
    real* hx = a.HX + ((size_t)traj * (a.n_steps + 1) + a.step) * 7;
    real* hu = a.HU + ((size_t)traj * a.n_steps + a.step) * 3;
    for (int i = 0; i < 7; ++i) { hx[i] = x[i]; Pg[P_X0 + i] = xn[i]; }
    for (int c = 0; c < 3; ++c) hu[c] = u[c];
    if (a.step == a.n_steps - 1)
      for (int i = 0; i < 7; ++i) hx[7 + i] = xn[i];
    Pg[P_TAU0] = (real)(tr.tau0 + tr.dtau);
    if (a.tally) {       // executed counts of this control step's solve, for the measurement (bench.py --config 4)
      const tsat_stats& st = a.stats[traj];
      long long* t = a.tally + (size_t)traj * 4;
      t[0] += st.n_backward; t[1] += st.n_forward; t[2] += (st.outer_iters > 1 ? st.outer_iters - 1 : 0); t[3] += st.inner_iters;
    }
  }
}

// ==================================================================================================
// Horizon selection: magnetic_gramian + condition_based_time (src/magnetic_toolbox.jl:1-31) for a
// batch. One trajectory per wavefront, lanes = table rows: each 64-row block is an inclusive wave scan of the six
// unique Gramian entries plus the carry of the previous blocks; every lane then has ITS prefix Gramian and evaluates
// its condition number with the closed-form (trigonometric) eigenvalues of a symmetric 3x3.
// ==================================================================================================
template <typename real>
struct HzArgs {
  int T, n_rows;
  const real* BT;       // [T][n_rows][3]
  const real* dt_row;   // [T]
  const real* cutoff;   // [T]
  int* tf_index;        // [T]
  real* cond_at;        // [T] or null
};

template <typename real>
TSAT_DEV real cond_sym3(real a00, real a01, real a02, real a11, real a12, real a22) {
  const real p1 = a01 * a01 + a02 * a02 + a12 * a12;
  const real q = (a00 + a11 + a22) * (real)(1.0 / 3.0);
  const real d0 = a00 - q, d1 = a11 - q, d2 = a22 - q;
  const real p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2 * p1;
  if (!(p2 > 0)) return (q > 0) ? (real)1 : inf_<real>();     // multiple of the identity
  const real p = sqrt_(p2 * (real)(1.0 / 6.0));
  const real ip = (real)1 / p;
  const real b00 = d0 * ip, b11 = d1 * ip, b22 = d2 * ip, b01 = a01 * ip, b02 = a02 * ip, b12 = a12 * ip;
  real r = (real)0.5 * (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02));
  r = r < -1 ? (real)-1 : (r > 1 ? (real)1 : r);
  const real phi = acos_(r) * (real)(1.0 / 3.0);
  const real lmax = q + 2 * p * cos_(phi);
  const real lmin = q + 2 * p * cos_(phi + (real)2.0943951023931954923);   // + 2 pi / 3
  return (lmin > 0) ? lmax / lmin : inf_<real>();
}

template <typename real>
TSAT_DEV void horizon_trajectory(const HzArgs<real>& a, int traj) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  const int n = a.n_rows;
  const real* B = a.BT + (size_t)traj * n * 3;
  const real dt = a.dt_row[traj], cut = a.cutoff[traj];
  real carry[6] = {0, 0, 0, 0, 0, 0};
  int found = 0;
  real cfound = inf_<real>();
  for (int i0 = 0; i0 < n && !found; i0 += WAVE) {
    const int i = i0 + lane;
    real g[6] = {0, 0, 0, 0, 0, 0};
    if (i < n) {
      const real b0 = B[3 * i], b1 = B[3 * i + 1], b2 = B[3 * i + 2];
      const real w = (i == 0) ? (real)1 : dt;                  // the first slice is not scaled by dt (:6)
      const real n2 = b0 * b0 + b1 * b1 + b2 * b2;
      g[0] = (n2 - b0 * b0) * w; g[1] = (-b0 * b1) * w; g[2] = (-b0 * b2) * w;
      g[3] = (n2 - b1 * b1) * w; g[4] = (-b1 * b2) * w; g[5] = (n2 - b2 * b2) * w;
    }
    // inclusive scan over the lanes (Hillis-Steele through LDS), six entries
    for (int c = 0; c < 6; ++c) {
      real v = g[c];
      for (int s = 1; s < WAVE; s <<= 1) {
        lds[L_RED + lane] = v;
        TSAT_SYNC_LDS();
        if (lane >= s) v += lds[L_RED + lane - s];
        TSAT_SYNC_LDS();
      }
      g[c] = v + carry[c];
    }
    const real cnd = (i < n) ? cond_sym3(g[0], g[1], g[2], g[3], g[4], g[5]) : inf_<real>();
    const int first = wave_first<real>(cnd < cut, lds + L_RED);
    if (first < WAVE) {
      found = i0 + first + 1;
      cfound = wave_bcast(cnd, first, lds + L_RED);
    }
    for (int c = 0; c < 6; ++c) carry[c] = wave_bcast(g[c], WAVE - 1, lds + L_RED);
  }
  if (lane == 0) {
    a.tf_index[traj] = found;
    if (a.cond_at) a.cond_at[traj] = cfound;
  }
}

// ==================================================================================================
// Field-table generation: magnetic_simulation (src/magnetic_toolbox.jl:33-106) for a batch.
// One orbit per wavefront: the Euler orbit is a short sequential loop (every lane runs it, lane 0 stores the
// positions); then lanes = table rows, each evaluating the degree-13 IGRF-12 sum for its sample with the Legendre
// recurrences fully unrolled (three rolling rows of P in registers, coefficients broadcast from LDS).
// ==================================================================================================
constexpr int IGRF_NMAX = 13, IGRF_NREC = 104, IGRF_RECW = 8;
// One record per (n, m), m = 0..n, n = 1..13, built on the host for the call's date and radius (tsat_host_pack.hpp,
// igrf_records): {g, h, d1, d2, a', b', -(n+1)/r, (a/r)^(n+1)}, where (a', b') are the Schmidt recurrence factors of the
// NEXT function the sweep needs (P[n][m+1], or P[n+1][0] on the diagonal). Every lane reads the same record, so the
// table is read with scalar loads (one s_load_dwordx16 per harmonic) and the loop carries no per-lane constants.
template <typename real>
struct BtArgs {
  int T, n_half;
  real mjd, gm, r_igrf_km;
  const real* tab;                  // [104][8] records
  const real* kep;                  // [T][6]
  const real* t0;                   // [T]
  const real* tf;                   // [T]
  real* pos;                        // [T][2N+1][3]
  real* B;                          // [T][2N][3]
};

// igrf12(date, r, lat, lon) geocentric, nT (src/igrf.jl:70-274; Legendre: src/legendre.jl:254-292, src/dlegendre.jl:221-309)
template <typename real>
TSAT_DEV void igrf12_eval(const TSAT_CONSTMEM real* tab, real r_km, real lat, real lon, real out[3]) {
  const real PI = (real)3.14159265358979323846;
  const real theta = PI / 2 - lat;
  const real phi = (lon >= 0) ? lon : 2 * PI + lon;
  const real c = cos_(theta), s = sqrt_(1 - c * c);
  const real a = (real)6371.2, r = r_km;
  const real sin_p = sin_(phi), cos_p = cos_(phi);
  const real dfact = (fmod_(theta, 2 * PI) > PI) ? (real)-1 : (real)1;
  const bool pole = (theta == 0);
  // sin(m phi), cos(m phi) by the reference's two-term recurrences (src/igrf.jl:208-246): the same for every degree
  real sm[IGRF_NMAX + 1], cm[IGRF_NMAX + 1];
  {
    real s1 = 0, s2 = -sin_p, c1 = 1, c2 = cos_p;
#ifndef TSAT_EMU
#pragma unroll
#endif
    for (int m = 1; m <= IGRF_NMAX; ++m) {
      sm[m] = 2 * cos_p * s1 - s2;
      cm[m] = 2 * cos_p * c1 - c2;
      s2 = s1; s1 = sm[m]; c2 = c1; c1 = cm[m];
    }
  }
  real dVr = 0, dVt = 0, dVp = 0;
  real Pa[IGRF_NMAX + 3], Pb[IGRF_NMAX + 3], Pc[IGRF_NMAX + 3];   // rows n-2, n-1, n (zero beyond the diagonal)
  for (int i = 0; i < IGRF_NMAX + 3; ++i) { Pa[i] = 0; Pb[i] = 0; Pc[i] = 0; }
  Pb[0] = 1; Pc[0] = c; Pc[1] = s;   // P[0][0]; P[1][0], P[1][1]
  const TSAT_CONSTMEM real* nxt = tab;
  real ar = 0, at = 0, ap = 0;
#ifndef TSAT_EMU
#pragma unroll
#endif
  for (int n = 1; n <= IGRF_NMAX; ++n) {
#ifndef TSAT_EMU
#pragma unroll
#endif
    for (int m = 0; m <= n; ++m) {
      const TSAT_CONSTMEM real* rec = nxt;   // record (n, m)
      nxt = rec + IGRF_RECW;
#ifndef TSAT_EMU
      // The records are invariant loads: left alone, all 104 x 8 values are hoisted to the top and spill 800 SGPRs.
      // Tying the NEXT record's address to this harmonic's running sum keeps exactly one record in flight ahead.
      asm volatile("" : "+s"(nxt), "+v"(ar), "+v"(at), "+v"(ap));
#endif
      if (n >= 2 && m + 1 <= n)     // P[n][m+1]: recurrence below the diagonal, closed form on it
        Pc[m + 1] = (m + 1 < n) ? rec[4] * c * Pb[m + 1] - rec[5] * Pa[m + 1] : s * rec[4] * Pb[n - 1];
      const real q = rec[6];
      if (m == 0) {
        const real dP0 = (rec[2] * Pc[1] + rec[3] * Pc[1]) * dfact;
        ar = q * rec[0] * Pc[0];
        at = rec[0] * dP0;
        ap = 0;
      } else {
        const real Gnm = rec[0], Hnm = rec[1];
        const real dPm = (rec[2] * Pc[m - 1] + rec[3] * Pc[m + 1]) * dfact;
        const real GcHs = Gnm * cm[m] + Hnm * sm[m], GsHc = Gnm * sm[m] - Hnm * cm[m];
        ar += q * GcHs * Pc[m];
        at += GcHs * dPm;
        ap += -(real)m * GsHc * (pole ? dPm : Pc[m]);
      }
      if (m == n) {
        const real fact = rec[7];
        dVr += ar * fact; dVp += ap * fact; dVt += at * fact;
        for (int i = 0; i < IGRF_NMAX + 3; ++i) { Pa[i] = Pb[i]; Pb[i] = Pc[i]; }
        Pc[n + 1] = 0; Pc[n + 2] = 0;
        if (n < IGRF_NMAX) Pc[0] = rec[4] * c * Pb[0] - rec[5] * Pa[0];   // P[n+1][0]
      }
    }
  }
  dVr *= a; dVp *= a; dVt *= a;
  out[0] = 1 / r * dVt;
  out[1] = pole ? -1 / r * dVp : -1 / (r * sin_(theta)) * dVp;
  out[2] = dVr;
}

template <typename real>
TSAT_DEV void btable_trajectory(const BtArgs<real>& a, int traj) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE();
  const int N = a.n_half;
  const real PI = (real)3.14159265358979323846, D2R = PI / 180;
  // ---- kep_ECI (src/kep_ECI.jl:1-35) ---------------------------------------------------------------
  const real* kp = a.kep + (size_t)traj * 6;
  const real t0 = a.t0[traj], tf = a.tf[traj];
  const real e = kp[0], sma = kp[1];
  const real Ma = fmod_(kp[5] + t0 * sqrt_(a.gm / (sma * sma * sma)), (real)360);
  real E = Ma / 180 * PI;
  for (int i = 0; i < 100; ++i) E = E - (E - e * sin_(E) - Ma / 180 * PI) / (1 - e * cos_(E));
  const real nu = 2 * atan2_(sqrt_(1 + e) * sin_(E / 2), sqrt_(1 - e) * cos_(E / 2)) * 180 / PI;
  const real r_c = sma * (1 - e * cos_(E));
  const real o0 = r_c * cos_(nu * D2R), o1 = r_c * sin_(nu * D2R);
  const real kk = sqrt_(a.gm * sma) / r_c;
  const real od0 = kk * -sin_(E), od1 = kk * sqrt_(1 - e * e) * cos_(E);
  real M[9];
  {  // R_z(-RAAN) R_x(-i) R_z(-argp), degree rotations [c s 0; -s c 0; 0 0 1] / [1 0 0; 0 c s; 0 -s c]
    const real ca = cos_(-kp[3] * D2R), sa = sin_(-kp[3] * D2R);
    const real cb = cos_(-kp[2] * D2R), sb = sin_(-kp[2] * D2R);
    const real cc = cos_(-kp[4] * D2R), sc = sin_(-kp[4] * D2R);
    const real A[9] = {ca, sa, 0, -sa, ca, 0, 0, 0, 1}, Bm[9] = {1, 0, 0, 0, cb, sb, 0, -sb, cb}, Cm[9] = {cc, sc, 0, -sc, cc, 0, 0, 0, 1};
    real T1[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { real v = 0; for (int k = 0; k < 3; ++k) v += A[3 * i + k] * Bm[3 * k + j]; T1[3 * i + j] = v; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { real v = 0; for (int k = 0; k < 3; ++k) v += T1[3 * i + k] * Cm[3 * k + j]; M[3 * i + j] = v; }
  }
  real u[6];
  for (int i = 0; i < 3; ++i) { u[i] = M[3 * i] * o0 + M[3 * i + 1] * o1; u[3 + i] = M[3 * i] * od0 + M[3 * i + 1] * od1; }
  // ---- Euler orbit, 2N+1 samples (src/magnetic_toolbox.jl:51-54, src/OrbitPlotter.jl:1-48) ---------------
  const real dt = (tf - t0) / (real)N;
  real* Pg = a.pos + (size_t)traj * 3 * (2 * N + 1);
  const real GMo = (real)(3.986004418E14 * 1e-9), J2 = (real)0.0010826359;
  for (int i = 0; i <= 2 * N; ++i) {
    if (lane == 0) { Pg[3 * i] = u[0]; Pg[3 * i + 1] = u[1]; Pg[3 * i + 2] = u[2]; }
    const real nr = sqrt_(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const real nr2 = nr * nr, nr7 = nr2 * nr2 * nr2 * nr;
    const real rho2 = u[0] * u[0] + u[1] * u[1];
    const real g = GMo / nr2 / nr;
    const real ax = g * -u[0] + J2 * u[0] / nr7 * (6 * u[2] - (real)1.5 * rho2);
    const real ay = g * -u[1] + J2 * u[1] / nr7 * (6 * u[2] - (real)1.5 * rho2);
    const real az = g * -u[2] + J2 * u[2] / nr7 * (3 * u[2] - (real)4.5 * rho2);
    const real v0 = u[3], v1 = u[4], v2 = u[5];
    u[0] += dt * v0; u[1] += dt * v1; u[2] += dt * v2;
    u[3] += dt * ax; u[4] += dt * ay; u[5] += dt * az;
  }
  TSAT_SYNC();
  // ---- rows: GMST, lat/long, IGRF, NED -> ENU -> ECEF -> ECI (src/magnetic_toolbox.jl:56-98) ------------------
  real* Bg = a.B + (size_t)traj * 3 * 2 * N;
  for (int i = lane; i < 2 * N; i += WAVE) {
    real b0 = 0, b1 = 0, b2 = 0;
    if (i < 2 * N - 1) {                        // the last row is left zero (:76)
      const real ti = t0 + dt * (real)i;
      const real gmst = ((real)280.4606 + (real)360.9856473 * (ti / 24 / 60 / 60 + a.mjd) - (real)51544.5) / 180 * PI;   // as written (:60)
      const real cg = cos_(gmst), sg = sin_(gmst);
      const real p0 = Pg[3 * i], p1 = Pg[3 * i + 1], p2 = Pg[3 * i + 2];
      const real e0 = cg * p0 + sg * p1, e1 = -sg * p0 + cg * p1, e2 = p2;
      const real lat = asin_(e2 / sqrt_(e0 * e0 + e1 * e1 + e2 * e2));
      const real lon = atan2_(e1, e0);
      real bn[3];
      igrf12_eval<real>((const TSAT_CONSTMEM real*)a.tab, a.r_igrf_km, lat, lon, bn);
      const real en0 = bn[1] * (real)1e-9, en1 = bn[0] * (real)1e-9, en2 = -bn[2] * (real)1e-9;    // /1e9, NED_to_ENU
      const real sl = sin_(lon), cl = cos_(lon), sa = sin_(lat), ca = cos_(lat);
      const real x = -sl * en0 - sa * cl * en1 + ca * cl * en2;
      const real y = cl * en0 - sa * sl * en1 + ca * sl * en2;
      const real z = ca * en1 + sa * en2;
      b0 = cg * x - sg * y; b1 = sg * x + cg * y; b2 = z;
    }
    Bg[3 * i] = b0; Bg[3 * i + 1] = b1; Bg[3 * i + 2] = b2;
  }
}

}  // namespace tsat
