// tsat_packed.hpp — packed solve for large batches: PK_G trajectories per wavefront (included after tsat_device.hpp).
//
// With one trajectory per wavefront the forward sweep spends a whole wave instruction on 64 line-search candidates of which
// about two are needed (mean accepted index 0.9), and it is half of all instructions of a solve. Once a batch is several
// times larger than the machine (T >> 1024 wavefronts) the lanes are better spent on MORE TRAJECTORIES: a wavefront owns
// PK_G trajectories, lane = (trajectory g, candidate c) with PK_C = 64 / PK_G candidates alpha = 2^-(c + shift) per sweep
// (a further sweep with shift += PK_C serves the trajectories whose search went deeper — rare), so ONE forward sweep advances
// PK_G trajectories. The backward sweeps run jointly too, four trajectories at a time (below: 64-lane Jacobian passes into an
// LDS record ring and, for the knots it does not hold, an HBM workspace; the row-oriented Riccati recursion on 16 lanes per
// trajectory); the lane-strided
// passes (adopt, costs, duals) stay per trajectory and run one after the other through the very same phase functions as the
// one-trajectory kernel. Each trajectory keeps its own position in the AL-iLQR iteration (outer / inner counters, penalty,
// regularisation, multipliers): the driver below is the loop body of solve_trajectory (tsat_device.hpp) turned into a
// per-trajectory state machine that is advanced between two forward sweeps. The arithmetic of a trajectory is, operation
// for operation, that of solve_trajectory: results are bit-identical to the other builds (tests: emulator and GPU).
// Compiled for two wavefronts per SIMD (20 KB of LDS, 256 registers: tsat_kernels_packed.hip, ...packed8.hip) and for ONE (40 KB,
// 512 registers: ...packed4w / 8w / 16w.hip, TSAT_PK_WAVES = 1), which is what the automatic choice takes (tsat_kernels.hip).
//
// Replaces, like solve_trajectory, TrajectoryOptimization.solve!(prob, solver) (src/TortoiseSat.jl:199; loop body of
// src/monte_carlo.jl:118-235) for a batch.
#pragma once
#include "tsat_device.hpp"

namespace tsat {

#ifndef TSAT_PK_G
#define TSAT_PK_G 4
#endif
#ifndef TSAT_PK_CK
#define TSAT_PK_CK 4
#endif
#ifndef TSAT_PK_NBUF
#define TSAT_PK_NBUF 2
#endif
#ifndef TSAT_PK_STORE
#define TSAT_PK_STORE 6
#endif
#ifndef TSAT_PK_WAVES
#define TSAT_PK_WAVES 2
#endif
// One wavefront per SIMD (the `w` builds): nobody hides this wavefront's waits, so the Riccati lanes read the record of knot l - 1
// from the ring while they work on knot l (riccati_rows does the same in the one-trajectory builds) and pad their copy stream so
// that a wait never has to drain the gain stores (riccati_group).
constexpr bool PK_ALONE = (TSAT_PK_WAVES == 1);
constexpr int PK_STORE = TSAT_PK_STORE;     // line-search candidates per trajectory whose rollouts a sweep keeps in HBM at most
constexpr int PK_FEW = 3;                   // ... and while the trajectory's line searches end early (see solve_group)
constexpr int PK_G = TSAT_PK_G;             // trajectories per wavefront
constexpr int PK_C = WAVE / PK_G;           // lanes (line-search candidates) per trajectory
constexpr int PK_CK = TSAT_PK_CK;           // knots per forward chunk and trajectory
constexpr int PK_NBUF = TSAT_PK_NBUF;       // 2: the next chunk is copied while this one is rolled out
static_assert(PK_G * PK_C == WAVE && (PK_C & (PK_C - 1)) == 0, "PK_G must be a power of two");
// Forward chunk buffer (reals): one region per array, each [trajectory][units of the chunk + ONE 16-byte pad unit] (field rows:
// + one pad row). global_load_lds fills LDS lane-linearly, so a region is linear in (trajectory, unit) and a copy
// instruction simply covers 64 consecutive units of it; the pad makes the trajectories' sub-blocks start 16 bytes (field rows:
// 160 bytes) past a multiple of 256 bytes... in effect at different LDS banks, so that the PK_G-address broadcast reads of
// the roll-out (same offset, different trajectory) do not conflict.
constexpr int PK_UKD = PK_CK * KDW / RPU, PK_ULM = PK_CK * LMW / RPU, PK_UXU = PK_CK * XUW / RPU;   // units per trajectory
static_assert((PK_CK * KDW) % RPU == 0 && (PK_CK * LMW) % RPU == 0 && (PK_CK * XUW) % RPU == 0, "chunks are whole 16-byte units");
constexpr int PK_SKD = (PK_UKD + 1) * RPU, PK_SLM = (PK_ULM + 1) * RPU, PK_SXU = (PK_UXU + 1) * RPU;   // strides in reals
constexpr int PK_BROWS = PK_CK * 3 + 1;                          // field rows per trajectory incl. the pad row
constexpr int PK_SB = PK_BROWS * 4, PK_SGT = PK_SLM;
constexpr int PK_R_KD = 0, PK_R_LM = PK_R_KD + PK_G * PK_SKD, PK_R_XU = PK_R_LM + PK_G * PK_SLM, PK_R_B = PK_R_XU + PK_G * PK_SXU,
              PK_R_GT = PK_R_B + PK_G * PK_SB, PK_FB = PK_R_GT + PK_G * PK_SGT;
// copy instructions per region (64 units each)
constexpr int PK_NI_KD = (PK_G * (PK_UKD + 1) + WAVE - 1) / WAVE, PK_NI_LM = (PK_G * (PK_ULM + 1) + WAVE - 1) / WAVE,
              PK_NI_XU = (PK_G * (PK_UXU + 1) + WAVE - 1) / WAVE, PK_NI_B = (PK_G * PK_BROWS * BROW_UNITS + WAVE - 1) / WAVE;
// (the lanes of a region's last copy instruction that lie past the region never copy — their `lim` is NEVER — so a buffer
// needs no slack behind it)
static_assert(L_FWD + PK_NBUF * PK_FB <= LDS_REALS, "packed forward buffers fit the wave's LDS block");

// ---- joint backward sweep: carve-up behind L_UNION (the forward chunk buffers overlay all of it between backward sweeps) -----
// The PK_G trajectories of the wave run their backward sweeps TOGETHER, four at a time, 16 knots of each per chunk:
//   Jacobian lanes   lane = (trajectory, knot): all 64 lanes linearise a knot each, all columns, and leave the finished record
//                    (84 values of type jac_t) in the LDS record ring — the last PK_RING knots of the pass, the first the recursion
//                    consumes — or in the wavefront's HBM workspace a.JW: the LDS of two wavefronts per SIMD holds four knots per
//                    trajectory (eight with float records), that of one wavefront per SIMD twelve (all sixteen float ones: no
//                    workspace traffic); a narrower Jacobian pass would repeat the primal stages on every lane of a knot
//                    (measured 2.3x the instructions per knot). The workspace is written and read back by the same wavefront
//                    within microseconds (42 KB per wavefront);
//   record ring      the Riccati lanes stream the workspace records back: the record PK_RING - 1 knots ahead is copied by
//                    global_load_lds into the slot of the knot just consumed (slots modulo PK_RING) while the recursion works; the
//                    wait counts younger LOADS only: safe whatever the stores do. A slot carries a 16-byte pad (PkRec::SLOT);
//   Riccati lanes    PK_C = 16 lanes per trajectory = one DPP row: lane j < NH + 3 owns COLUMN j of [A|B] and of the cost-to-go; the
//                    very step function of the one-trajectory builds (riccati_row_step, tsat_device.hpp: every cross-lane operand
//                    is the `row_newbcast` source of the FMA that consumes it — no exchange through LDS inside a knot), four
//                    trajectories side by side: the results are bit-identical to the one-trajectory builds.
constexpr int PK_JCH = 16;                        // knots per trajectory and Jacobian pass (= lanes per trajectory)
#ifndef TSAT_PK_RING
#define TSAT_PK_RING ((sizeof(jac_t) == 8) ? 4 : 8)
#endif
constexpr int PK_RING = TSAT_PK_RING;      // knots of every trajectory resident in LDS during the recursion (20 KB: 4 double records, 8 float ones; 40 KB: 12 / 16)
constexpr int PK_CPB = 16;                        // bytes per lane of a ring copy instruction (riccati_group)
constexpr int PK_BC = 16;                         // lanes per trajectory in a backward pass
constexpr int PK_BG = WAVE / PK_BC;               // trajectories per backward pass (4); a wave of PK_G trajectories takes PK_G / PK_BG passes
static_assert(PK_G % PK_BG == 0 && PK_JCH == PK_BC, "Jacobian lanes: the 16 lanes of a trajectory linearise 16 knots");
// Knot record: PkRec<ES> (tsat_device.hpp) — 84 reals in the full state, 72 in error coordinates.
constexpr int PK_RECS = 84;                       // the larger of the two: LDS ring and workspace are sized for it
constexpr int PK_GTRW = 88;                       // per-trajectory constants: staged parameter record (76), nu (8), pad
constexpr int PK_GT_NU = 76;
constexpr int PK_GXW = 64;                        // per-trajectory block of the Riccati lanes: S~ = [S; s'] between two passes, doubles [r][8]
static_assert(PkRec<0>::RECS % RPUJ == 0 && PkRec<1>::RECS % RPUJ == 0 && PkRec<0>::RECS <= PK_RECS, "a record is a whole number of 16-byte units");
constexpr int L_GTR = L_UNION;
constexpr int L_GX = L_GTR + PK_BG * PK_GTRW;
constexpr int L_GREC = L_GX + PK_BG * PK_GXW;
constexpr int PK_SLOT = PK_BG * PK_RECS + RPUJ;         // record values (jac_t) per ring slot (PkRec::SLOT of the larger record)
static_assert(L_GREC * (int)sizeof(cfg_real) + PK_RING * PK_SLOT * (int)sizeof(jac_t) <= LDS_REALS * (int)sizeof(cfg_real), "joint backward sweep fits the wave's LDS block");
constexpr int PK_JW_WAVE = PK_JCH * PK_SLOT;             // workspace values (jac_t) per wavefront (its passes of four trajectories share it)
// the record ring (values of type jac_t, from L_GREC on)
TSAT_DEV jac_t* ring_base() { return reinterpret_cast<jac_t*>(lds_base<cfg_real>() + L_GREC); }
static_assert(PK_JW_WAVE <= TSAT_JW_REALS_PER_4, "host allocation of a.JW");
// packed index of (i <= j) in an n x n upper triangle, row by row
constexpr int sym_ut(int i, int j, int n) { return i * n - (i * (i - 1)) / 2 + (j - i); }
// hand-over between the trajectories' state lanes (PK_C per trajectory) and the lanes of a backward pass (PK_BC per trajectory),
// in the part of the one-trajectory Riccati scratch the packed build does not use
template <typename real> struct BwdIn { int N, need; real mu, rho; int cur, pad; };
template <typename real> struct BwdRes { acc_t dV1, dV2; int ok, pad; };
constexpr int L_BWT = L_WT;      // (the scratch of the element-oriented recursion: unused by the solve kernels)
static_assert(L_BWT % 2 == 0, "hand-over tables are 8-byte aligned");

// per-trajectory position in the AL-iLQR iteration; lives in the registers of the trajectory's PK_C lanes and is made
// wave-uniform (through LDS) for the phases that work on one trajectory at a time
template <typename real>
struct GState {
  acc_t Jprev, dV1, dV2, Jw;
  real mu, rho, drho, grad, rho_used, nu[7];
  int N, active, status, outer, it, inner_iters, ls_trials, n_backward, n_forward, bp_restarts, fp_fails, djz, trow, regfail,
      found, jw, slot, need_bwd, last_jw, cur;      // last_jw: accepted line-search index of the previous iteration; cur: the slab that holds the nominal trajectory (0 = XU, 1 .. = candidate slabs; cand_slab)
};

template <typename real>
TSAT_DEV GState<real> gstate_bcast(const GState<real>& s, int src) {
  static_assert(sizeof(GState<real>) <= (L_UNION - L_ST) * sizeof(real), "state record fits the Riccati scratch");
  GState<real>* buf = reinterpret_cast<GState<real>*>(lds_base<real>() + L_ST);   // free outside the Riccati recursion
  if (TSAT_LANE() == src) *buf = s;
  TSAT_SYNC_LDS();
  const GState<real> r = *buf;
  TSAT_SYNC_LDS();
  return r;
}

template <typename real>
TSAT_DEV TPtrs<real> group_ptrs(const KArgs<real>& a, int traj, int cur = 0, int N = 0) {      // cur != 0: N = the trajectory's own knot count
  const int NS = a.N;
  TPtrs<real> p;
  p.XU = (TSAT_GLOBAL real*)(a.XU + (size_t)traj * xu_stride<real>(NS));
  p.KD = (TSAT_GLOBAL real*)(a.KD + (size_t)traj * kd_stride<real>(NS));
  p.LAM = (TSAT_GLOBAL real*)(a.LAM + (size_t)traj * lam_stride<real>(NS));
  p.CAND = (TSAT_GLOBAL real*)(a.CAND + (size_t)traj * a.max_ls * xu_stride<real>(NS));
  p.bt = (const TSAT_GLOBAL real*)(a.BT + (size_t)a.bidx[traj] * a.n_tab * 4);
  p.XU0 = p.XU; p.cur = cur;
  if (cur) p.XU = slab_ptr<real>(p, N, cur);      // the accepted roll-out is adopted by pointer, as in solve_trajectory
  return p;
}

// One forward sweep for the PK_G trajectories of the wave: lane (g, c) rolls out alpha = 2^-(c + shift) for trajectory g.
// `live`: this lane's trajectory takes part; N, mu, nu: its horizon, penalty and terminal multipliers. Same per-knot
// arithmetic as forward_sweep (tsat_device.hpp); the knot records of all PK_G trajectories are staged through LDS in
// PK_CK-knot chunks (global_load_lds, lane = 16-byte unit) and read back as PK_G-address broadcasts.
// its own (non-inlined) function: the sweep gets a register allocation of its own — no scratch traffic in its loop — instead of
// sharing one with the driver's per-trajectory state (-DTSAT_PK_FWD_INLINE: measured 4 % slower)
#ifdef TSAT_PK_FWD_INLINE
#define TSAT_PK_FWD TSAT_FWD
#else
#define TSAT_PK_FWD TSAT_PHASE
#endif
template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_PK_FWD FwdOut<real> forward_sweep_packed(const KArgs<real>& a, int traj0, int closed, int shift, int n_store, bool live, int N,
                                           real mu, const real nu[7], int term_mask, real max_state, int nom) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE(), myg = lane / PK_C, myc = lane % PK_C;
  const int n_tab = a.n_tab;
  const int tmax = a.T - 1;
  const int traj = (traj0 + myg <= tmax) ? traj0 + myg : tmax;
  // this lane's trajectory constants, straight from its parameter record
  Traj<real> tr;
  real x[7];
  {
    const TSAT_GLOBAL real* P = (const TSAT_GLOBAL real*)(a.P + (size_t)traj * PSTRIDE);
    for (int i = 0; i < 7; ++i) { tr.xf[i] = P[P_XF + i]; tr.Qd[i] = P[P_QD + i]; tr.Qfd[i] = P[P_QFD + i]; x[i] = P[P_X0 + i]; }
    for (int i = 0; i < 3; ++i) { tr.Rd[i] = P[P_RD + i]; tr.ulo[i] = P[P_ULO + i]; tr.uhi[i] = P[P_UHI + i]; }
    tr.h = P[P_DT];
    for (int i = 0; i < 9; ++i) { tr.J[i] = P[P_J + i]; tr.hJi[i] = tr.h * P[P_JI + i]; }
    tr.hh = (real)0.5 * tr.h; tr.us = (real)a.opt.u_scale; tr.usj = tr.us * tr.hJi[0];
    tr.tau0 = (double)P[P_TAU0] + (double)P[P_TAU0L]; tr.dtau = (double)P[P_DTAU] + (double)P[P_DTAUL];
    tr.N = N; tr.n_tab = n_tab; tr.bt = nullptr;
  }
  // table of the copy lanes (they serve every trajectory of the wave): horizon and table clock per trajectory, in the Riccati
  // scratch, which is free during a forward sweep
  double* pk_tau0 = reinterpret_cast<double*>(lds + L_ST);
  double* pk_dtau = pk_tau0 + PK_G;
  int* pk_n = reinterpret_cast<int*>(pk_dtau + PK_G);
  int* pk_cur = pk_n + PK_G;                 // slab of each trajectory's nominal records (`nom`: this lane's trajectory)
  static_assert((2 * PK_G * sizeof(double) + 2 * PK_G * sizeof(int)) <= (L_UNION - L_ST) * sizeof(real), "copy-lane tables fit the Riccati scratch");
  TSAT_SYNC_LDS();
  if (myc == 0) { pk_tau0[myg] = tr.tau0; pk_dtau[myg] = tr.dtau; pk_n[myg] = live ? N : 0; pk_cur[myg] = nom; }
  TSAT_SYNC_LDS();
  int nmax = 0;
  for (int g = 0; g < PK_G; ++g) nmax = (pk_n[g] > nmax) ? pk_n[g] : nmax;
  real alpha = 1;
  for (int j = 0; j < myc + shift && j < TSAT_MAX_LINESEARCH; ++j) alpha *= (real)0.5;
  const TPtrs<real> pm = group_ptrs<real>(a, traj);
  TSAT_GLOBAL real* Cg = slab_ptr<real>(pm, N, cand_slab(nom, myc < n_store ? myc : 0));     // the slabs other than the nominal one, in order
  const bool store = live && myc < n_store;
  HalfWeights<real> hw;
  for (int i = 0; i < 7; ++i) hw.hQd[i] = (real)0.5 * tr.Qd[i];
  for (int i = 0; i < 3; ++i) hw.hRd[i] = (real)0.5 * tr.Rd[i];
  hw.hmu = (real)0.5 * mu;
  acc_t J = 0;
  real amax = 0;

  // ---- static roles of this lane in the copy instructions (fixed for the whole sweep): source pointer of chunk 0 and the
  // last chunk start for which the lane has something to copy (k0 < lim); a lane of a pad unit, of a trajectory that does not
  // take part, or past the region's end never copies
  constexpr int NEVER = -0x40000000;
  const TSAT_GLOBAL real* kd_src[PK_NI_KD]; int kd_lim[PK_NI_KD];
  const TSAT_GLOBAL real* lm_src[PK_NI_LM]; int lm_lim[PK_NI_LM];
  const TSAT_GLOBAL real* xu_src[PK_NI_XU]; int xu_lim[PK_NI_XU];
  const TSAT_GLOBAL real* b_src[PK_NI_B]; int b_lim[PK_NI_B]; int b_kk[PK_NI_B]; double b_t0[PK_NI_B], b_dt[PK_NI_B], b_st[PK_NI_B];
  auto role = [&](int i, int units, int W, const real* base, size_t stride, const TSAT_GLOBAL real*& src, int& lim) {
    const int g = i / (units + 1), e = i - g * (units + 1);
    const int tg = (traj0 + g <= tmax) ? traj0 + g : tmax;
    const int Ng = (g < PK_G) ? pk_n[g] : 0;
    src = (const TSAT_GLOBAL real*)(base + (size_t)tg * stride + (size_t)e * RPU);
    lim = (g < PK_G && e < units && Ng > 0) ? Ng - 1 - (e * RPU) / W : NEVER;
  };
#ifndef TSAT_EMU
#pragma unroll
#endif
  for (int j = 0; j < PK_NI_KD; ++j) {
    role(lane + WAVE * j, PK_UKD, KDW, a.KD, kd_stride<real>(a.N), kd_src[j], kd_lim[j]);
    if (!closed) kd_lim[j] = NEVER;
  }
#ifndef TSAT_EMU
#pragma unroll
#endif
  for (int j = 0; j < PK_NI_LM; ++j) role(lane + WAVE * j, PK_ULM, LMW, a.LAM, lam_stride<real>(a.N), lm_src[j], lm_lim[j]);
#ifndef TSAT_EMU
#pragma unroll
#endif
  for (int j = 0; j < PK_NI_XU; ++j) {
    role(lane + WAVE * j, PK_UXU, XUW, a.XU, xu_stride<real>(a.N), xu_src[j], xu_lim[j]);
    const int i = lane + WAVE * j, g = i / (PK_UXU + 1), e = i - g * (PK_UXU + 1);
    if (g < PK_G && pk_cur[g] != 0) {        // this trajectory's nominal records live in a candidate slab
      const int tg = (traj0 + g <= tmax) ? traj0 + g : tmax;
      xu_src[j] = (const TSAT_GLOBAL real*)(a.CAND + (size_t)tg * a.max_ls * xu_stride<real>(a.N) + (size_t)(pk_cur[g] - 1) * (size_t)pk_n[g] * XUW + (size_t)e * RPU);
    }
  }
#ifndef TSAT_EMU
#pragma unroll
#endif
  for (int j = 0; j < PK_NI_B; ++j) {
    const int i = lane + WAVE * j;
    const int g = i / (PK_BROWS * BROW_UNITS), rem = i - g * (PK_BROWS * BROW_UNITS), r = rem / BROW_UNITS, part = rem - r * BROW_UNITS;
    const int tg = (traj0 + g <= tmax) ? traj0 + g : tmax;
    const int Ng = (g < PK_G) ? pk_n[g] : 0;
    const int kk = r / 3;
    b_src[j] = (const TSAT_GLOBAL real*)(a.BT + (size_t)a.bidx[tg] * n_tab * 4 + (size_t)part * RPU);
    b_lim[j] = (g < PK_G && r < PK_CK * 3 && Ng > 0) ? Ng - 1 - kk : NEVER;
    b_kk[j] = kk; b_st[j] = 0.5 * (double)(r - 3 * kk);
    b_t0[j] = pk_tau0[g < PK_G ? g : 0]; b_dt[j] = pk_dtau[g < PK_G ? g : 0];
  }
  // copy of chunk k0 into buffer `fb`
  auto issue = [&](real* fb, int k0) {
#ifndef TSAT_EMU
#pragma unroll
#endif
    for (int j = 0; j < PK_NI_KD; ++j)
      if (k0 < kd_lim[j]) glds_put_at<real>(fb + PK_R_KD + GLDS * j, kd_src[j] + (size_t)k0 * KDW);
#ifndef TSAT_EMU
#pragma unroll
#endif
    for (int j = 0; j < PK_NI_LM; ++j)
      if (k0 < lm_lim[j]) glds_put_at<real>(fb + PK_R_LM + GLDS * j, lm_src[j] + (size_t)k0 * LMW);
#ifndef TSAT_EMU
#pragma unroll
#endif
    for (int j = 0; j < PK_NI_XU; ++j)
      if (k0 < xu_lim[j]) glds_put_at<real>(fb + PK_R_XU + GLDS * j, xu_src[j] + (size_t)k0 * XUW);
#ifndef TSAT_EMU
#pragma unroll
#endif
    for (int j = 0; j < PK_NI_B; ++j)
      if (k0 < b_lim[j]) {
        // floor(fma(k + c, dtau, tau0)) clamped — the reference's B_ECI[floor(Int, t*N + 1), :] (src/DerivFunction.jl:28)
        const double rr = floor_(fma_((double)(k0 + b_kk[j]) + b_st[j], b_dt[j], b_t0[j]));
        const int row = (rr >= 0.0) ? (rr > (double)(n_tab - 1) ? n_tab - 1 : (int)rr) : 0;
        glds_put_at<real>(fb + PK_R_B + GLDS * j, b_src[j] + (size_t)row * 4);
      }
  };
  auto gates = [&](real* fb) {
    for (int e = lane; e < PK_G * PK_CK * LMW; e += WAVE) {
      const int g = e / (PK_CK * LMW), r = e - g * (PK_CK * LMW);
      fb[PK_R_GT + g * PK_SGT + r] = (fb[PK_R_LM + g * PK_SLM + r] > 0) ? -inf_<real>() : (real)0;
    }
  };
  int cur = 0;
  issue(lds + L_FWD, 0);
  TSAT_SYNC();
  gates(lds + L_FWD);
  TSAT_SYNC_LDS();
  for (int k0 = 0; k0 < nmax - 1; k0 += PK_CK) {
    const int kn = k0 + PK_CK;
    const bool more = kn < nmax - 1;
    real* fb = lds + L_FWD + cur * PK_FB;
    real* fbn = lds + L_FWD + ((PK_NBUF == 2) ? (1 - cur) : 0) * PK_FB;
    if (PK_NBUF == 2 && more) issue(fbn, kn);
    for (int kk = 0; kk < PK_CK; ++kk) {
      const int k = k0 + kk;
      if (live && k < N - 1) {
        const real* xu = fb + PK_R_XU + myg * PK_SXU + kk * XUW;
        real u[3] = {xu[7], xu[8], xu[9]};
        if (closed) {
          const real* kd = fb + PK_R_KD + myg * PK_SKD + kk * KDW;
          real dx[7];
          if (ES) {
            // quaternion_error(new, nominal) = [dw; MRP(q_nom^-1 (x) q_new)]  (src/quaternion_toolbox.jl:58-75)
            for (int i = 0; i < 3; ++i) dx[i] = x[i] - xu[i];
            const real s1 = xu[3], a1 = -xu[4], a2 = -xu[5], a3 = -xu[6];
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
        for (int i = 0; i < 7; ++i) amax = fmaxabs_(amax, x[i]);
        for (int c = 0; c < 3; ++c) amax = fmaxabs_(amax, u[c]);
        J += (acc_t)stage_cost_gated(tr, hw, x, u, fb + PK_R_LM + myg * PK_SLM + kk * LMW, fb + PK_R_GT + myg * PK_SGT + kk * LMW);
#ifdef TSAT_PK_EXPERIMENT_NOSTORE   /* timing experiment only: results are wrong without the candidate records */
        if (store && k == 0) store_record5<real>(Cg + (size_t)k * XUW, x, u);
#else
        if (store) store_record5<real>(Cg + (size_t)k * XUW, x, u);   // REC_STORES store instructions (counted below)
#endif
        const real* br = fb + PK_R_B + myg * PK_SB + kk * 12;
        const real b0[3] = {br[0], br[1], br[2]}, b1[3] = {br[4], br[5], br[6]}, b2[3] = {br[8], br[9], br[10]};
        real xn[7];
        rk_step<real, INTEG, DIAGJ, ES>(tr, x, u, b0, b1, b2, xn);
        for (int i = 0; i < 7; ++i) x[i] = xn[i];
      }
    }
    if (more) {
      if (PK_NBUF == 1) {          // single buffer: copy the next chunk now that this one has been consumed
        TSAT_SYNC_LDS();
        issue(fbn, kn);
      }
      // The copy of the next chunk has to have landed: a full vmcnt(0). (Waiting for "all but the chunk's candidate stores" —
      // s_waitcnt vmcnt(PK_CK x REC_STORES), the copy being older than those stores — is NOT safe: loads and stores retire in
      // order among themselves but not relative to each other, so under memory load a store can retire before the older copy
      // and the count is met with the copy still in flight. It showed as run-to-run differences of the float eight-per-wave
      // build at 16384 trajectories (tools/repeat_runs.py); the drain costs 0 - 3 %.)
      TSAT_SYNC();
      gates(fbn);
      TSAT_SYNC_LDS();
      if (PK_NBUF == 2) cur = 1 - cur;
    }
  }
  for (int i = 0; i < 7; ++i) amax = fmaxabs_(amax, x[i]);
  J += (acc_t)term_cost(tr, x, nu, mu, term_mask, true);
  if (store) {
    TSAT_GLOBAL real* cr = Cg + (size_t)(N - 1) * XUW;
    for (int i = 0; i < 7; ++i) cr[i] = x[i];
    for (int c = 0; c < 3; ++c) cr[7 + c] = 0;
  }
  FwdOut<real> out;
  out.J = J;
  out.ok = ((amax <= max_state) && (J == J)) ? 1 : 0;
  return out;
}

// --------------------------------------------------------------------------------------------------
// joint backward sweep, Jacobian lanes: knots kb0 .. kb0 + 15 of the pass's four trajectories (`need`, group-uniform; N, mu:
// the lane's own trajectory), lane = (trajectory, knot). Each lane leaves the finished record of its knot — [A|B], lx, lu, luu;
// error-state mode: in error coordinates, G'QG in the tenth column's place — in the wavefront's workspace `jw`
// ([4][16][PK_RECS]). Same per-knot arithmetic as jacobian_chunk (tsat_device.hpp).
// --------------------------------------------------------------------------------------------------
template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_PHASE void jacobian16(const KArgs<real>& a, int traj0, TSAT_GLOBAL jac_t* jw, int kb0, bool need, int N, real mu, int cur) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE(), g = lane / PK_BC, kk = lane % PK_BC;
  const int k = kb0 + kk;
  if (!(need && k < N - 1)) return;
  const int tmax = a.T - 1;
  const int traj = (traj0 + g <= tmax) ? traj0 + g : tmax;
  const TPtrs<real> p = group_ptrs<real>(a, traj, cur, N);
  const Traj<real> tr = load_traj_at<real>(lds + L_GTR + g * PK_GTRW, N, a.n_tab, p.bt);
  const TSAT_GLOBAL real* xu = p.XU + (size_t)k * XUW;
  real x[7], u[3], lam[6], b0[3], b1[3], b2[3];
  for (int i = 0; i < 7; ++i) x[i] = xu[i];
  for (int c = 0; c < 3; ++c) u[c] = xu[7 + c];
  for (int c = 0; c < 6; ++c) lam[c] = p.LAM[(size_t)k * LMW + c];
  const TSAT_GLOBAL real* p0 = tr.bt + (size_t)brow_index(tr, k, 0.0) * 4;
  const TSAT_GLOBAL real* p1 = tr.bt + (size_t)brow_index(tr, k, 0.5) * 4;
  const TSAT_GLOBAL real* p2 = tr.bt + (size_t)brow_index(tr, k, 1.0) * 4;
  for (int c = 0; c < 3; ++c) { b0[c] = p0[c]; b1[c] = p1[c]; b2[c] = p2[c]; }
  // the last PK_RING knots of the chunk are the first the recursion consumes: their lanes put the record straight into its
  // ring slot in LDS; the others go through the workspace (one code path for both: a generic pointer, flat stores)
  using R = PkRec<ES>;
  jac_t* rc = (kk >= PK_JCH - PK_RING) ? ring_base() + (kk % PK_RING) * R::SLOT + g * R::RECS
                                       : (jac_t*)(jw + (size_t)kk * R::SLOT + g * R::RECS);      // workspace: [knot][trajectory][record], a ring slot contiguous
  real qn[4];
  for (int i = 0; i < 4; ++i) qn[i] = xu[XUW + 3 + i];
  knot_record<real, INTEG, DIAGJ, ES, jac_t*>(tr, x, u, lam, b0, b1, b2, qn, mu, rc);
}

// --------------------------------------------------------------------------------------------------
// joint backward sweep, Riccati lanes: the recursion over one chunk (last knot first) for the four trajectories of a pass at once
// --------------------------------------------------------------------------------------------------
// wait until at most min(l, PK_RING - 1) copy sets of NCI instructions are outstanding (the immediate of s_waitcnt is a
// compile-time constant: one case per possible count)
template <int NCI, int M = PK_RING - 1>
TSAT_DEV void ring_wait(int l) {
  if (l >= M) { TSAT_SYNC_OLDER_THAN(M * NCI); }
  else if constexpr (M > 0) ring_wait<NCI, M - 1>(l);
}
// one wavefront per SIMD (riccati_group): copy instructions issued at knots l + PK_RING - 3 ... l — NCI per knot that copies a record,
// `dummy` per knot that does not
constexpr int ring_younger(int nci, int dummy, int l) {
  int c = 0;
  for (int i = l; i <= l + PK_RING - 3; ++i) c += (i >= PK_RING - 1 && i < PK_JCH - 1) ? nci : dummy;
  return c;
}
template <typename real> struct GBwd { acc_t dV1, dV2; int ok; };

// ROW-oriented (riccati_row_step, tsat_device.hpp): the trajectory's 16 lanes are a DPP row, lane j owns column j of F and of S~,
// and every value another lane holds is read as the `row_newbcast` operand of the FMA that consumes it — no exchange block, no
// barrier and no re-read of S~ between the steps of a knot; the four trajectories of the pass run the same instructions side by
// side (a row whose trajectory does not take part in a knot is switched off as a whole). The one-trajectory builds run the very
// same step function: bit-identical results. S~ lives in the registers of the row across the 16 knots of a pass and in the
// trajectory's exchange block (as doubles, [r][8]) between passes.
// Record ring: the records of 12 (8 with float records) of a pass's 16 knots come back from the wavefront's workspace: a slot (one
// knot of the four trajectories, contiguous) is copied by global_load_lds PK_RING - 1 knots ahead of its use, every copy lane
// with a fixed role; the wait counts the copy sets issued SINCE — younger loads only, safe whatever the gain stores do.
template <typename real, int NH>
TSAT_PHASE GBwd<real> riccati_group(const KArgs<real>& a, int traj0, const TSAT_GLOBAL jac_t* jw, int kb0, bool need, int N, real rho_,
                                         acc_t dV1, acc_t dV2, int ok_in) {
  constexpr int ES = (NH == 6) ? 1 : 0;
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE(), g = lane / PK_BC, j = lane % PK_BC;
  using R = PkRec<ES>;
  static_assert(sizeof(real) == 8, "the exchange block holds S~ as doubles");
  double* gs = reinterpret_cast<double*>(lds + L_GX + g * PK_GXW);
  static_assert((NH + 1) * 8 <= PK_GXW, "S~ fits the exchange block");
  RowRoles<NH, R> ro;
  ro.template set<real>(j, lds + L_GTR + g * PK_GTRW + P_QD);
  const jac_t* recs = ring_base() + g * R::RECS;                   // this trajectory's record inside a ring slot
  // Copy lanes of the record ring. A slot (one knot of the pass's four trajectories) is contiguous in the workspace as it is in
  // LDS ([knot][trajectory][record]): NCI copy instructions of PK_CPB bytes per lane, lane-linear. The wait below may only
  // count younger LOADS (M x NCI of them) while vmcnt counts the gain stores of the knots in between as well: with three 16-byte
  // copies per knot the window is 9 operations in the double builds (14 with float records). A perturbation experiment
  // (profiles/r04/kstore_probe.txt — a scratch diagnostic build that issued the gain stores of a knot n more times: three MORE stores per knot add 1000 cycles per trajectory and knot) shows where that window
  // saturates; as built it does not: widening it to 27 with nine 4-byte copies per knot (PK_CPB = 4) changed nothing in the packed
  // build and cost the packed8 build 250 cycles per trajectory and knot in copy instructions.
  constexpr int SLOT_BYTES = R::SLOT * (int)sizeof(jac_t), NCI = (SLOT_BYTES + WAVE * PK_CPB - 1) / (WAVE * PK_CPB);
  static_assert((PK_RING - 1) * NCI <= 63, "the wait count fits vmcnt");
  const TSAT_GLOBAL unsigned char* cp_src = reinterpret_cast<const TSAT_GLOBAL unsigned char*>(jw) + PK_CPB * lane;
  auto ring_copy = [&](int q) {
    unsigned char* slot = reinterpret_cast<unsigned char*>(ring_base() + (q % PK_RING) * R::SLOT);
    const TSAT_GLOBAL unsigned char* src = cp_src + (size_t)q * SLOT_BYTES;
    for (int i = 0; i < NCI; ++i)
      if (PK_CPB * (lane + WAVE * i) < SLOT_BYTES) glds_copy<PK_CPB>(slot + PK_CPB * WAVE * i, src + PK_CPB * WAVE * i);
  };
  const int tmax = a.T - 1;
  const int traj = (traj0 + g <= tmax) ? traj0 + g : tmax;
  TSAT_GLOBAL real* KDg = (TSAT_GLOBAL real*)(a.KD + (size_t)traj * kd_stride<real>(a.N));
  const double rho = (double)rho_;
  const double km = (j < NH) ? 1.0 : 0.0, dm = (j == 7) ? 1.0 : 0.0;
  int slot_[3];
  for (int c = 0; c < 3; ++c) slot_[c] = (j < 7) ? (c * 7 + j) : (21 + c);
  bool ok = ok_in != 0;
  RowState<NH> st;
  const int js = (j < NH) ? j : 0;
  for (int r = 0; r <= NH; ++r) st.Sc[r] = gs[r * 8 + js];
#ifndef TSAT_EMU
  if constexpr (PK_ALONE) {
    // One wavefront per SIMD. The record of knot l - 1 is read while knot l is worked on, two register sets taking turns. fetch(l),
    // at the top of knot l: this knot's copy instructions, then — if it came through the workspace — the wait for the copy of
    // record l - 1, then its read.
    // The wait may only count YOUNGER LOADS (TSAT_SYNC_OLDER_THAN), and vmcnt counts the gain stores in between as well: with
    // nothing else in flight the last copies of a pass would be waited for with vmcnt(0), i.e. together with every gain store
    // of the pass — a store takes microseconds to retire under the launch's write load, and this wavefront has no neighbour on
    // its SIMD to fill the time. So every knot that has no record to copy issues PK_DUMMY four-byte copies into a sink behind
    // the ring: loads like the others, retiring in order behind the real copies. The copy of record q, issued at knot
    // q + PK_RING - 1, is then always followed by the copy instructions of PK_RING - 2 more knots before it is waited for, and
    // the wait lets that many operations — gain stores among them — stay in flight.
    static_assert(PK_JCH % 2 == 0 && PK_RING >= 3, "two knots per turn");
    constexpr int PK_DUMMY = 2, NVIA = PK_JCH - PK_RING;       // NVIA: records per trajectory and pass that go through the workspace
    unsigned char* sink = reinterpret_cast<unsigned char*>(lds + LDS_REALS) - 4 * WAVE;      // (one sink for all of them)
    static_assert(NVIA == 0 || (L_GREC * (int)sizeof(cfg_real) + PK_RING * PK_SLOT * (int)sizeof(jac_t) + 4 * WAVE <= LDS_REALS * (int)sizeof(cfg_real)),
                  "the sink of the padding copies lies behind the ring");
    const TSAT_GLOBAL unsigned char* dsrc = reinterpret_cast<const TSAT_GLOBAL unsigned char*>(jw) + 4 * lane;
    auto recp = [&](int l) { return recs + (((l > 0) ? l : 0) % PK_RING) * R::SLOT; };
    auto fetch = [&](int l) {
      if constexpr (NVIA > 0) {
        if (l >= PK_RING - 1 && l < PK_JCH - 1) ring_copy(l - (PK_RING - 1));
        else if (l >= 1 && l < PK_RING - 1)
          for (int i = 0; i < PK_DUMMY; ++i) glds_copy<4>(sink, dsrc + 4 * WAVE * i);
        // record l - 1 (< NVIA) was copied at knot l + PK_RING - 2; younger copy instructions: those of knots l + PK_RING - 3 ... l
        static_assert(NVIA <= 4, "one case per record that goes through the workspace");
        if (l == 1) { constexpr int c = ring_younger(NCI, PK_DUMMY, 1); static_assert(c <= 63); TSAT_SYNC_OLDER_THAN(c); }
        else if (l == 2 && NVIA >= 2) { constexpr int c = ring_younger(NCI, PK_DUMMY, 2); static_assert(c <= 63); TSAT_SYNC_OLDER_THAN(c); }
        else if (l == 3 && NVIA >= 3) { constexpr int c = ring_younger(NCI, PK_DUMMY, 3); static_assert(c <= 63); TSAT_SYNC_OLDER_THAN(c); }
        else if (l == 4 && NVIA >= 4) { constexpr int c = ring_younger(NCI, PK_DUMMY, 4); static_assert(c <= 63); TSAT_SYNC_OLDER_THAN(c); }
      }
      return row_load<jac_t, NH, R>(recp(l - 1), ro);
    };
    auto knot = [&](const RowIn<NH>& in, int l) {
      const int k = kb0 + l;
      const bool act = need && ok && k < N - 1;
      double Kc[3], d[3];
      bool pd = true;
      if (act) pd = riccati_row_step<NH, R>(st, in, ro, rho, Kc, d, dV1, dV2);
      if (act && j < 8) {
        TSAT_GLOBAL real* kd = KDg + (size_t)k * KDW;
        for (int c = 0; c < 3; ++c) kd[slot_[c]] = (real)fma_(Kc[c], km, d[c] * dm);
      }
      ok = ok && (!act || pd);
    };
    RowIn<NH> A = row_load<jac_t, NH, R>(recp(PK_JCH - 1), ro);
    // (unrolled: which knots copy, pad or wait, and the ring slot of every knot, are then compile-time facts — 19 % off the pass)
#pragma unroll
    for (int l = PK_JCH - 1; l >= 0; l -= 2) {
      const RowIn<NH> B = fetch(l);
      TSAT_SCHED_FENCE();
      knot(A, l);
      A = fetch(l - 1);
      TSAT_SCHED_FENCE();
      knot(B, l - 1);
    }
  } else
#pragma unroll
#endif
  for (int l = PK_JCH - 1; l >= 0; --l) {
    const int k = kb0 + l;
    const bool act = need && ok && k < N - 1;
    if (l >= PK_RING - 1 && l < PK_JCH - 1) ring_copy(l - (PK_RING - 1));
    ring_wait<NCI>(l);
    const jac_t* rc = recs + (l % PK_RING) * R::SLOT;
    double Kc[3], d[3];
#ifdef TSAT_EMU
    // every emulated lane takes part in the exchanges of the step; a row that does not take part keeps its state
    RowState<NH> st2 = st;
    acc_t v1 = dV1, v2 = dV2;
    const bool pd = riccati_row_step<NH, R>(st2, row_load<jac_t, NH, R>(rc, ro), ro, rho, Kc, d, v1, v2);
    if (act) { st = st2; dV1 = v1; dV2 = v2; }
#else
    bool pd = true;
    if (act) pd = riccati_row_step<NH, R>(st, row_load<jac_t, NH, R>(rc, ro), ro, rho, Kc, d, dV1, dV2);
#endif
    if (act && j < 8) {      // K,d record of the knot: lanes 0..6 their gain column (zero beyond NH), lane 7 the feed-forward; stays in flight
      TSAT_GLOBAL real* kd = KDg + (size_t)k * KDW;
      for (int c = 0; c < 3; ++c) kd[slot_[c]] = (real)fma_(Kc[c], km, d[c] * dm);
    }
    ok = ok && (!act || pd);
  }
  TSAT_SYNC_LDS();
  if (j < NH)
    for (int r = 0; r <= NH; ++r) gs[r * 8 + j] = st.Sc[r];
  GBwd<real> out;
  out.dV1 = dV1; out.dV2 = dV2; out.ok = ok ? 1 : 0;
  return out;
}

// The rest of ONE trajectory's solve in the one-trajectory mapping (solve_trajectory, tsat_device.hpp, in the layout of the dense
// build: this translation unit's LDS block holds it), from the top of an inner iteration on. A wavefront whose other trajectories
// have finished runs its last one this way: a joint iteration costs the same whether four trajectories of the wave are live or
// one (the forward sweep and the Riccati lanes are per wavefront), while the one-trajectory mapping spends all 64 lanes on
// that trajectory — Jacobian passes of 29 knots, 64 candidates per sweep, the 64-lane Riccati step — and iterates 2 - 3 times faster. With an
// iteration budget that spreads the trajectories' iteration counts (3 x 50: 21 ... 150), most wavefronts are down to one live
// trajectory for the last third of the launch. Same bits either way (every build is the same solve).
template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_PHASE int continue_trajectory(const KArgs<real>& a, int traj, Resume<real> r) {      // 1: parked for the endgame, not finished
  return solve_trajectory<real, INTEG, DIAGJ, ES>(a, traj, &r);
}

// the whole AL-iLQR solve of trajectories traj0 .. traj0 + PK_G - 1 (traj0 = wave * PK_G) by one wavefront
template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_DEV void solve_group(const KArgs<real>& a, int wave) {
  real* lds = lds_base<real>();
  const int lane = TSAT_LANE(), myg = lane / PK_C, myc = lane % PK_C;
  const tsat_options& o = a.opt;
  const int NS = a.N, n_tab = a.n_tab;
  const int traj0 = wave * PK_G;
  const int ntr = (a.T - traj0 < PK_G) ? (a.T - traj0) : PK_G;
  const real max_state = (real)o.max_state;
  const int tmask = o.terminal_mask;
  const int max_ls = o.max_linesearch;
  const int ns0 = (max_ls < a.max_ls) ? max_ls : a.max_ls;
  const int n_store = (ns0 < PK_C) ? ns0 : PK_C;     // candidate slots written per sweep

  GState<real> mine;
  mine.Jprev = 0; mine.dV1 = 0; mine.dV2 = 0; mine.Jw = 0;
  mine.mu = (real)o.penalty_init; mine.rho = 0; mine.drho = 0; mine.grad = 0; mine.rho_used = 0;
  for (int i = 0; i < 7; ++i) mine.nu[i] = 0;
  mine.active = (myg < ntr) ? 1 : 0;
  mine.N = mine.active ? (a.nk ? a.nk[traj0 + myg] : NS) : 2;
  mine.status = TSAT_MAX_OUTER; mine.outer = 0; mine.it = 0; mine.inner_iters = 0; mine.ls_trials = 0; mine.n_backward = 0;
  mine.n_forward = 0; mine.bp_restarts = 0; mine.fp_fails = 0; mine.djz = 0; mine.trow = 0; mine.regfail = 0;
  mine.found = 0; mine.jw = 0; mine.slot = 0; mine.need_bwd = 0; mine.cur = 0; mine.last_jw = 0;

  // initial_controls!(prob, U0) (src/TortoiseSat.jl:191) + zero multipliers, trajectory by trajectory (lanes = knots)
  for (int g = 0; g < ntr; ++g) {
    const int traj = traj0 + g;
    const int N = a.nk ? a.nk[traj] : NS;
    const TPtrs<real> p = group_ptrs<real>(a, traj);
    const real* U0g = a.U0 + (size_t)traj * u0_stride<real>(NS);
    for (int k = lane; k < N; k += WAVE) {
      for (int i = 0; i < 7; ++i) p.XU[(size_t)k * XUW + i] = 0;
      for (int c = 0; c < 3; ++c) p.XU[(size_t)k * XUW + 7 + c] = (k < N - 1) ? U0g[(size_t)k * 3 + c] : (real)0;
      if (k < N - 1)
        for (int c = 0; c < 6; ++c) p.LAM[(size_t)k * LMW + c] = 0;
    }
  }
  if (lane < 4) lds[L_PC + lane] = 0;
  TSAT_SYNC();

  // ---- per-trajectory pieces of the driver; `u` is the wave-uniform copy of the state of the trajectory being advanced ---
  // trajectory constants + terminal multipliers into the wave's LDS block: what the per-trajectory phase functions read
  auto stage_for = [&](int g, const GState<real>& u) {
    TSAT_SYNC();
    stage_traj<real>((const TSAT_GLOBAL real*)(a.P + (size_t)(traj0 + g) * PSTRIDE), (real)o.u_scale);
    if (lane < 7) lds[L_NU + lane] = u.nu[lane];
    TSAT_SYNC();
  };
  auto finish = [&](int g, const TPtrs<real>& p_in, GState<real>& u) {   // final statistics of trajectory g
    TSAT_SYNC();
    TPtrs<real> p = p_in;
    if (u.cur != 0) {     // the nominal trajectory lives in one of the candidate slabs: bring it home, once per solve
      const TSAT_GLOBAL real* src = slab_ptr<real>(p, u.N, u.cur);
      for (int k = lane; k < u.N; k += WAVE)
        for (int i = 0; i < XUW; ++i) p.XU0[(size_t)k * XUW + i] = src[(size_t)k * XUW + i];
      p.XU = p.XU0; p.cur = 0; u.cur = 0;
      TSAT_SYNC();
    }
    const real cmax = violation_and_duals<real>(p, u.N, u.mu, tmask, 0, (real)o.dual_max);
    const acc_t cost = nominal_cost<real>(p, u.N, u.mu, tmask, 0);
    const acc_t cost_al = nominal_cost<real>(p, u.N, u.mu, tmask, 1);
    if (lane == 0) {
      tsat_stats& st = a.stats[traj0 + g];
      st.status = u.status; st.outer_iters = u.outer; st.inner_iters = u.inner_iters; st.ls_trials = u.ls_trials;
      st.n_backward = u.n_backward; st.n_forward = u.n_forward; st.bp_restarts = u.bp_restarts; st.fp_fails = u.fp_fails;
      st.cost = (double)cost; st.cost_al = (double)cost_al; st.c_max = (double)cmax; st.grad = (double)u.grad;
      if (a.live) (void)TSAT_ATOMIC_ADD(a.live, -1);
    }
    u.active = 0; u.need_bwd = 0;
  };
  // first step of an outer iteration: AL cost of the nominal, fresh regularisation; the backward sweep follows (joint_backward)
  auto start_outer = [&](const TPtrs<real>& p, GState<real>& u, int outer) {
    u.outer = outer;
    u.Jprev = nominal_cost<real>(p, u.N, u.mu, tmask, 1);
    u.rho = (real)o.reg_init; u.drho = 0; u.djz = 0; u.regfail = 0; u.it = 1;
    u.need_bwd = 1;
  };
  // end of an inner loop (budget, tolerance or a regularisation failure): outer-loop bookkeeping; the trajectory is finished or
  // starts its next outer iteration
  auto end_inner = [&](int g, const TPtrs<real>& p, GState<real>& u) {
    TSAT_SYNC();
    const real cmax = violation_and_duals<real>(p, u.N, u.mu, tmask, 0, (real)o.dual_max);
    if (u.regfail) { u.status = TSAT_REG_FAIL; finish(g, p, u); return; }
    if (cmax < (real)o.constraint_tol) { u.status = TSAT_CONVERGED; finish(g, p, u); return; }
    if (u.outer == o.max_outer) { finish(g, p, u); return; }
    (void)violation_and_duals<real>(p, u.N, u.mu, tmask, 1, (real)o.dual_max);   // dual update of the control box
    if (lane < 7 && ((tmask >> lane) & 1)) {
      const real dmax = (real)o.dual_max;
      real v = lds[L_NU + lane] + u.mu * (p.XU[(size_t)(u.N - 1) * XUW + lane] - lds[L_TR + P_XF + lane]);
      v = v > -dmax ? v : -dmax; v = v < dmax ? v : dmax;
      lds[L_NU + lane] = v;
    }
    u.mu = (u.mu * (real)o.penalty_scale < (real)o.penalty_max) ? u.mu * (real)o.penalty_scale : (real)o.penalty_max;
    TSAT_SYNC();
    for (int i = 0; i < 7; ++i) u.nu[i] = lds[L_NU + i];
    start_outer(p, u, u.outer + 1);
  };
  // Advance trajectory g from "a forward sweep has just been evaluated" (after_forward) or from the very start to the point
  // where it needs its next backward sweep (u.need_bwd) or is finished.
  unsigned long long pc_adopt = 0, pc_end = 0;   // diagnostic build: the copy of the accepted roll-out + gradient; outer-loop bookkeeping
  auto advance = [&](int g, GState<real>& u, bool after_forward) {
    const int traj = traj0 + g;
    double* trace = a.trace ? a.trace + (size_t)traj * a.trace_rows * 8 : nullptr;
    stage_for(g, u);
    if (!after_forward) {
      u.cur = cand_slab(u.cur, 0);                      // the open-loop rollout (slot 0) becomes the nominal trajectory: a pointer swap
      TSAT_SYNC();
      start_outer(group_ptrs<real>(a, traj, u.cur, u.N), u, 1);
      return;
    }
    acc_t J;
    const unsigned long long t_ad0 = tick_();
    u.last_jw = u.found ? u.jw : max_ls;
    if (u.found) {
      J = u.Jw;
      u.ls_trials += u.jw + 1;
      u.cur = cand_slab(u.cur, u.slot);                 // adoption of the accepted roll-out by pointer
    } else {
      J = u.Jprev;
      u.ls_trials += max_ls;
      u.fp_fails++;
      u.drho = (u.drho * (real)o.reg_scale > (real)o.reg_scale) ? u.drho * (real)o.reg_scale : (real)o.reg_scale;
      u.rho = (u.rho * u.drho > (real)o.reg_min) ? u.rho * u.drho : (real)o.reg_min;
      u.rho += (real)o.reg_fp;
    }
    const TPtrs<real> p = group_ptrs<real>(a, traj, u.cur, u.N);
    u.grad = todorov_gradient<real>(p.XU, p.KD, u.N);
    TSAT_SYNC();
    pc_adopt += tick_() - t_ad0;
    acc_t dJ = J - u.Jprev;
    dJ = dJ < 0 ? -dJ : dJ;
    if (trace && lane == 0 && u.trow < a.trace_rows) {
      double* r = trace + 8 * u.trow;
      r[0] = u.outer; r[1] = u.it; r[2] = (double)u.Jprev; r[3] = (double)J; r[4] = u.found ? u.jw : -1;
      r[5] = (double)u.rho_used; r[6] = (double)u.dV1; r[7] = (double)u.dV2;
#if defined(TSAT_PROFILE) && !defined(TSAT_EMU)
      r[7] = (double)wall_clock64();     // diagnostic build: when (100 MHz counter) the iteration ended (tools/straggler_timeline.py)
#endif
    }
    u.trow++;
    u.Jprev = J;
    u.djz = (dJ == 0) ? u.djz + 1 : 0;
    u.inner_iters++;
    const bool inner_over = (0 < dJ && dJ < (acc_t)o.cost_tol) || (u.grad < (real)o.grad_tol) || (u.djz > o.dj_counter_limit) ||
                            (u.it >= o.max_inner);
    if (!inner_over) {
      u.it++;
      u.need_bwd = 1;
      return;
    }
    const unsigned long long t_e0 = tick_();
    end_inner(g, p, u);
    pc_end += tick_() - t_e0;
  };
  unsigned long long pc_jac = 0, pc_ric = 0;   // diagnostic build: shader clocks in the Jacobian lanes / the Riccati lanes
  // Backward sweeps of all trajectories that need one (mine.need_bwd), together, with their regularisation restarts:
  // a trajectory whose Quu_reg was not positive definite raises its rho and takes part in the next round.
  auto joint_backward = [&]() {
    constexpr int NH = BwdCfg<ES>::NH, NP = NH * (NH + 1) / 2;
    BwdIn<real>* tin = reinterpret_cast<BwdIn<real>*>(lds + L_BWT);
    BwdRes<real>* tres = reinterpret_cast<BwdRes<real>*>(tin + PK_G);
    static_assert(L_BWT * sizeof(real) + PK_G * (sizeof(BwdIn<real>) + sizeof(BwdRes<real>)) <= L_UNION * sizeof(real), "hand-over tables fit");
    const int bg = lane / PK_BC;                 // this lane's trajectory within a backward pass
    for (;;) {
      const bool need = mine.need_bwd != 0;
      TSAT_SYNC_LDS();
      if (myc == 0) { tin[myg].N = mine.N; tin[myg].need = need ? 1 : 0; tin[myg].mu = mine.mu; tin[myg].rho = mine.rho; tin[myg].cur = mine.cur; }
      TSAT_SYNC_LDS();
      int any = 0;
      for (int g = 0; g < PK_G; ++g) any |= tin[g].need;
      if (!any) break;
      for (int t0 = 0; t0 < ntr; t0 += PK_BG) {      // a pass: trajectories t0 .. t0 + PK_BG - 1 of the wave
        int nmax = 0;
        for (int g = 0; g < PK_BG; ++g)
          if (tin[t0 + g].need && tin[t0 + g].N > nmax) nmax = tin[t0 + g].N;
        const BwdIn<real> in = tin[t0 + bg];
        GBwd<real> bw;
        bw.dV1 = 0; bw.dV2 = 0; bw.ok = 1;
        if (nmax > 0) {
          // constants and terminal cost-to-go of every trajectory that takes part, into its blocks
          for (int g = 0; g < PK_BG && t0 + g < ntr; ++g) {
            if (!tin[t0 + g].need) continue;
            const GState<real> u = gstate_bcast(mine, (t0 + g) * PK_C);
            stage_for(t0 + g, u);
            real* gt = lds + L_GTR + g * PK_GTRW;
            for (int e = lane; e < PK_GT_NU + 8; e += WAVE) gt[e] = lds[L_TR + e];     // L_NU follows L_TR
            const TSAT_GLOBAL real* XUg = group_ptrs<real>(a, traj0 + t0 + g, u.cur, u.N).XU;
            terminal_cost_to_go<real, ES>(XUg, u.N, u.mu, tmask);
            TSAT_SYNC_LDS();
            real* gx = lds + L_GX + g * PK_GXW;
            {          // row-oriented recursion: S~ as doubles, [r][8]
              double* gs = reinterpret_cast<double*>(gx);
              for (int e = lane; e < (NH + 1) * 8; e += WAVE)
                if ((e & 7) < NH) gs[e] = (double)lds[L_ST + (e >> 3) * 9 + (e & 7)];
            }
            TSAT_SYNC_LDS();
          }
          const bool bneed = in.need != 0;
          TSAT_GLOBAL jac_t* jw = (TSAT_GLOBAL jac_t*)a.JW + (size_t)wave * PK_JW_WAVE;
          const int nch = (nmax - 1 + PK_JCH - 1) / PK_JCH;
          for (int ch = nch - 1; ch >= 0; --ch) {
            const int kb0 = ch * PK_JCH;
            const unsigned long long c0 = tick_();
            jacobian16<real, INTEG, DIAGJ, ES>(a, traj0 + t0, jw, kb0, bneed && bw.ok, in.N, in.mu, in.cur);
            TSAT_SYNC();         // the records are in the workspace (vmcnt(0)) before the ring copies read them
            const unsigned long long c1 = tick_();
            bw = riccati_group<real, NH>(a, traj0 + t0, jw, kb0, bneed, in.N, in.rho, bw.dV1, bw.dV2, bw.ok);
            // every copy has landed and every record has been consumed before the next pass overwrites them. One wavefront per
            // SIMD: every real copy has been waited for inside the pass and the padding copies write a sink, so the gain stores
            // need not be drained here (the Jacobian lanes' own drain, a pass of arithmetic later, finds them retired)
            if (PK_ALONE) { TSAT_SYNC_LDS(); } else { TSAT_SYNC(); }
            pc_jac += c1 - c0; pc_ric += tick_() - c1;
          }
          if (PK_ALONE) TSAT_SYNC();     // the gains are in memory before anything reads them
        }
        if (lane % PK_BC == 0) { tres[t0 + bg].dV1 = bw.dV1; tres[t0 + bg].dV2 = bw.dV2; tres[t0 + bg].ok = bw.ok; }
        TSAT_SYNC_LDS();
      }
      if (need) {
        const BwdRes<real> bw = tres[myg];
        mine.n_backward++;
        if (bw.ok) {
          mine.dV1 = bw.dV1; mine.dV2 = bw.dV2;
          mine.rho_used = mine.rho;
          const real inv = (real)1 / (real)o.reg_scale;    // regularisation decrease
          mine.drho = (mine.drho / (real)o.reg_scale < inv) ? mine.drho / (real)o.reg_scale : inv;
          const real r = mine.rho * mine.drho;
          mine.rho = (r > (real)o.reg_min) ? r : (real)0;
          mine.need_bwd = 0;
        } else {
          mine.bp_restarts++;
          mine.drho = (mine.drho * (real)o.reg_scale > (real)o.reg_scale) ? mine.drho * (real)o.reg_scale : (real)o.reg_scale;
          mine.rho = (mine.rho * mine.drho > (real)o.reg_min) ? mine.rho * mine.drho : (real)o.reg_min;
          if (mine.rho > (real)o.reg_max) { mine.regfail = 1; mine.need_bwd = 0; }
        }
      }
    }
    TSAT_SYNC();      // the gains are in HBM before a forward sweep reads them
    // a trajectory whose regularisation ran out ends here (status REG_FAIL)
    for (int g = 0; g < ntr; ++g) {
      GState<real> u = gstate_bcast(mine, g * PK_C);
      if (u.active && u.regfail) {
        stage_for(g, u);
        end_inner(g, group_ptrs<real>(a, traj0 + g, u.cur, u.N), u);
      }
      if (myg == g) mine = u;
    }
  };
  auto any_lane = [&](bool f) { return wave_first<real>(f, lds + L_RED) < WAVE; };
  // If exactly one trajectory of the wave is still iterating (it stands at the top of an inner iteration: its backward sweep is
  // next), the one-trajectory mapping takes it to the end (continue_trajectory) and the wave is done.
  auto hand_over_last = [&]() {
    static_assert(L_BWD_END <= LDS_REALS && L_FWD_END <= LDS_REALS, "the one-trajectory phases (dense layout) fit this build's LDS block");
    int* flags = reinterpret_cast<int*>(lds + L_ST);
    TSAT_SYNC_LDS();
    if (myc == 0) flags[myg] = (mine.active && mine.need_bwd && !mine.regfail) ? 1 : (mine.active ? 2 : 0);
    TSAT_SYNC_LDS();
    int n_live = 0, gl = 0;
    for (int g = 0; g < PK_G; ++g) {
      const int f = flags[g];
      if (f) { n_live += (f == 1) ? 1 : 2; gl = g; }        // a live trajectory in any other state keeps the wave in the joint loop
    }
    TSAT_SYNC_LDS();
    if (n_live != 1) return;
    const GState<real> u = gstate_bcast(mine, gl * PK_C);
    Resume<real> r;
    r.Jprev = u.Jprev; r.mu = u.mu; r.rho = u.rho; r.drho = u.drho; r.grad = u.grad;
    for (int i = 0; i < 7; ++i) r.nu[i] = u.nu[i];
    r.outer = u.outer; r.it = u.it; r.djz = u.djz; r.inner_iters = u.inner_iters; r.ls_trials = u.ls_trials;
    r.n_backward = u.n_backward; r.n_forward = u.n_forward; r.bp_restarts = u.bp_restarts; r.fp_fails = u.fp_fails; r.trow = u.trow;
        r.cur = u.cur;
    TSAT_SYNC();
    const int parked = continue_trajectory<real, INTEG, DIAGJ, ES>(a, traj0 + gl, r);
    TSAT_SYNC();
    if (a.live && !parked && lane == 0) (void)TSAT_ATOMIC_ADD(a.live, -1);
    if (myg == gl) { mine.active = 0; mine.need_bwd = 0; }
  };
  // Suspension: the batch is down to a.suspend_at live trajectories — fewer than the machine has wavefront slots for, so each can
  // have a wavefront (and the faster one-trajectory mapping) of its own. The wave parks its live trajectories, each at the top of
  // an inner iteration, and leaves; tsat_resume_kernel_packed (the launch that follows on the stream) carries them on with
  // continue_trajectory. No wave ever waits for another: the counter is only read. Which trajectories end up parked depends on
  // the order the hardware ran the waves in; their results do not (every mapping is the same solve, bit for bit) — only the
  // diagnostic count n_forward, which already differs between the builds, does.
  auto suspend_if_endgame = [&]() {
    if (!a.suspend_at || !a.live) return;
    int* seen = reinterpret_cast<int*>(lds + L_RED);     // one read for the whole wave: the decision is wave-uniform
    TSAT_SYNC_LDS();
    if (lane == 0) seen[0] = TSAT_ATOMIC_LOAD(a.live);
    TSAT_SYNC_LDS();
    const int live_now = seen[0];
    TSAT_SYNC_LDS();
    if (live_now > a.suspend_at) return;
    for (int g = 0; g < ntr; ++g) {
      const GState<real> u = gstate_bcast(mine, g * PK_C);
      if (!(u.active && u.need_bwd && !u.regfail)) continue;       // (a trajectory in any other state finishes in the joint loop)
      if (lane == 0) {
        const int pos = TSAT_ATOMIC_ADD(a.susp_n, 1);
        a.susp_ids[pos] = traj0 + g;
        Resume<real>& r = reinterpret_cast<Resume<real>*>(a.susp_state)[pos];
        r.Jprev = u.Jprev; r.mu = u.mu; r.rho = u.rho; r.drho = u.drho; r.grad = u.grad;
        for (int i = 0; i < 7; ++i) r.nu[i] = u.nu[i];
        r.outer = u.outer; r.it = u.it; r.djz = u.djz; r.inner_iters = u.inner_iters; r.ls_trials = u.ls_trials;
        r.n_backward = u.n_backward; r.n_forward = u.n_forward; r.bp_restarts = u.bp_restarts; r.fp_fails = u.fp_fails; r.trow = u.trow;
        r.cur = u.cur;
      }
      if (myg == g) { mine.active = 0; mine.need_bwd = 0; }
    }
    TSAT_SYNC();
  };

  unsigned long long pc_fwd = 0, pc_adv = 0;   // diagnostic build (-DTSAT_PROFILE): shader clocks in forward sweeps / everything else
  // ---- open-loop rollout of U0 for every trajectory of the wave, then the first backward sweeps ------------------------
  {
    const FwdOut<real> f0 = forward_sweep_packed<real, INTEG, DIAGJ, ES>(a, traj0, 0, 0, 1, mine.active != 0, mine.N, mine.mu, mine.nu,
                                                                         tmask, max_state, mine.cur);
    if (mine.active) mine.n_forward++;
    mine.Jw = f0.J;            // lane (g, 0) holds the rollout's cost (all PK_C lanes of a group computed the same rollout)
    mine.found = f0.ok;
    TSAT_SYNC();
    for (int g = 0; g < ntr; ++g) {
      GState<real> u = gstate_bcast(mine, g * PK_C);
      if (!u.found || !(u.Jw - u.Jw == 0)) {            // non-finite initial rollout
        u.status = TSAT_DIVERGED;
        const TPtrs<real> p = group_ptrs<real>(a, traj0 + g);
        TSAT_SYNC();
        stage_traj<real>((const TSAT_GLOBAL real*)(a.P + (size_t)(traj0 + g) * PSTRIDE), (real)o.u_scale);
        if (lane < 7) lds[L_NU + lane] = 0;
        TSAT_SYNC();
        u.cur = cand_slab(u.cur, 0);               // (the overflowed rollout is what the caller gets back)
        finish(g, p, u);
      } else {
        advance(g, u, false);
      }
      if (myg == g) mine = u;
    }
    hand_over_last();
    suspend_if_endgame();
    joint_backward();
  }
  // ---- main loop: one forward sweep for all trajectories that are still iterating, then each of them moves on ------------
  while (any_lane(mine.active != 0)) {
    // Line search of every iterating trajectory: a sweep rolls out its PK_C candidates alpha = 2^-(shift + c) and keeps the
    // rollouts of the first n_store in HBM. The first accepted candidate wins (exactly what sequential backtracking picks); if
    // its rollout was not kept (index >= n_store: 0.2 % of the iterations on the Monte-Carlo workloads) the trajectory rolls
    // out that one alpha again in the next sweep of the wave, into slot 0; if none of the PK_C was accepted and candidates are
    // left (max_linesearch > shift + PK_C), the next sweep takes the next PK_C.
    mine.found = 0;
    int g_shift = 0, g_mode = 0;              // per trajectory: 0 searching, 1 re-roll of the winner pending, 2 search over
    // How many roll-outs a trajectory keeps is a fifth of the launch's HBM writes: while its line searches end early it keeps
    // PK_FEW of them (a deeper winner is rolled out once more, below), after a deep search all n_store — the rule of
    // solve_trajectory. The accepted candidate, and with it every result, is the same either way. Measured on one MI355X
    // (tools/store_probe.py, TSAT_PK_FEW): 16384 x 1000 fp64 573 / 547 / 538 / 539 ms with 6 / 4 / 3 / 2 kept — the eight-per-wave
    // launch is short of HBM write bandwidth —, nothing either way on the configs[3] shard, whose searches are deep.
    const int few = a.pk_few > 0 ? a.pk_few : PK_FEW;
    const int my_store = (mine.last_jw >= few - 1 || n_store < few) ? n_store : few;
    for (;;) {
      const bool live = mine.active && !mine.found && g_mode != 2;
      if (!any_lane(live)) break;
      TSAT_SYNC();
      const unsigned long long t_f0 = tick_();
      const FwdOut<real> fw = forward_sweep_packed<real, INTEG, DIAGJ, ES>(a, traj0, 1, g_shift, (g_mode == 1) ? 1 : my_store, live, mine.N,
                                                                           mine.mu, mine.nu, tmask, max_state, mine.cur);
      pc_fwd += tick_() - t_f0;
      if (live) mine.n_forward++;
      // first accepted candidate of each searching trajectory
      const int ci = myc + g_shift;
      acc_t alpha = 1;
      for (int j = 0; j < ci && j < TSAT_MAX_LINESEARCH; ++j) alpha *= (acc_t)0.5;
      const acc_t Jc = fw.J;
      const acc_t expected = -alpha * (mine.dV1 + alpha * mine.dV2);
      const acc_t z = (expected > 0) ? (mine.Jprev - Jc) / expected : (acc_t)-1;
      const bool acc = live && g_mode == 0 && (ci < max_ls) && fw.ok &&
                       ((z > (acc_t)o.ls_lower && z <= (acc_t)o.ls_upper) || Jc < mine.Jprev);
      real* red = lds + L_RED;
      acc_t* r64 = red64();
      real v = acc ? (real)myc : (real)PK_C;
      for (int s = 1; s < PK_C; s <<= 1) {   // minimum over the PK_C lanes of the group (lane ^ s stays inside it)
        red[lane] = v;
        TSAT_SYNC_LDS();
        const real w = red[lane ^ s];
        v = (w < v) ? w : v;
        TSAT_SYNC_LDS();
      }
      const int jl = (int)v;
      r64[lane] = Jc;
      TSAT_SYNC_LDS();
      const acc_t Jwin = r64[myg * PK_C + (jl < PK_C ? jl : 0)];
      TSAT_SYNC_LDS();
      if (live) {
        if (g_mode == 1) {                       // the winner's rollout is in slot 0 now
          mine.found = 1; mine.slot = 0;
        } else if (jl < PK_C) {
          mine.jw = g_shift + jl; mine.Jw = Jwin;
          if (jl < my_store) { mine.found = 1; mine.slot = jl; }
          else { g_mode = 1; g_shift = mine.jw; }
        } else {
          g_shift += PK_C;
          if (g_shift >= max_ls) g_mode = 2;     // no candidate accepted: the line search has failed
        }
      }
    }
    TSAT_SYNC();
    const unsigned long long t_a0 = tick_();
    for (int g = 0; g < ntr; ++g) {
      GState<real> u = gstate_bcast(mine, g * PK_C);
      if (u.active) advance(g, u, true);
      if (myg == g) mine = u;
    }
    hand_over_last();
    suspend_if_endgame();
    joint_backward();
    pc_adv += tick_() - t_a0;
  }
#ifdef TSAT_PROFILE
  {   // row 0 of the wave's first trajectory carries the phase clocks of the whole wave (PK_G trajectories)
    int its = 0, nb = 0;
    for (int g = 0; g < ntr; ++g) {
      const GState<real> u = gstate_bcast(mine, g * PK_C);
      its += u.inner_iters; nb += u.n_backward;
    }
    if (a.trace && a.trace_rows > 0 && lane == 0) {
      double* trace = a.trace + (size_t)traj0 * a.trace_rows * 8;
      const double jac = (double)pc_jac, ric = (double)pc_ric;
      trace[0] = (double)pc_fwd; trace[1] = jac; trace[2] = ric; trace[3] = (double)pc_adv - jac - ric;
      trace[4] = (double)its; trace[5] = (double)nb; trace[6] = (double)pc_adopt; trace[7] = (double)pc_end;

    }
  }
#else
  (void)pc_fwd; (void)pc_adv; (void)pc_jac; (void)pc_ric; (void)pc_adopt; (void)pc_end;
#endif
}

}  // namespace tsat
