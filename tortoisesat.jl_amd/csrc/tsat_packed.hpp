// tsat_packed.hpp — packed solve for large batches: PK_G trajectories per wavefront (included after tsat_device.hpp).
//
// With one trajectory per wavefront the forward sweep spends a whole wave instruction on 64 line-search candidates of which
// about two are needed (mean accepted index 0.9), and it is half of all instructions of a solve. Once a batch is several
// times larger than the machine (T >> 1024 wavefronts) the lanes are better spent on MORE TRAJECTORIES: a wavefront owns
// PK_G trajectories, lane = (trajectory g, candidate c) with PK_C = 64 / PK_G candidates alpha = 2^-(c + shift) per sweep
// (a further sweep with shift += PK_C serves the trajectories whose search went deeper — rare), so ONE forward sweep advances
// PK_G trajectories. The backward sweeps (Jacobian lanes + Riccati recursion, lanes = knots / matrix elements) and the
// lane-strided passes stay per trajectory and run one after the other through the very same phase functions as the
// one-trajectory kernel. Each trajectory keeps its own position in the AL-iLQR iteration (outer / inner counters, penalty,
// regularisation, multipliers): the driver below is the loop body of solve_trajectory (tsat_device.hpp) turned into a
// per-trajectory state machine that is advanced between two forward sweeps. The arithmetic of a trajectory is, operation
// for operation, that of solve_trajectory: results are bit-identical to the other builds (tests: emulator and GPU).
//
// Replaces, like solve_trajectory, TrajectoryOptimization.solve!(prob, solver) (src/TortoiseSat.jl:199; loop body of
// src/monte_carlo.jl:118-235) for a batch.
#pragma once
#include "tsat_device.hpp"

namespace tsat {

#ifndef TSAT_PK_G
#define TSAT_PK_G 8
#endif
#ifndef TSAT_PK_CK
#define TSAT_PK_CK 4
#endif
#ifndef TSAT_PK_NBUF
#define TSAT_PK_NBUF 1
#endif
constexpr int PK_G = TSAT_PK_G;             // trajectories per wavefront
constexpr int PK_C = WAVE / PK_G;           // line-search candidates per trajectory and sweep
constexpr int PK_CK = TSAT_PK_CK;           // knots per forward chunk and trajectory
constexpr int PK_NBUF = TSAT_PK_NBUF;
static_assert(PK_G * PK_C == WAVE && (PK_C & (PK_C - 1)) == 0, "PK_G must be a power of two");
// chunk record of one trajectory (reals): gains, multipliers, (x,u) records, 3 field rows per knot (4 reals each), gates
constexpr int PK_KD = 0, PK_LM = PK_KD + PK_CK * KDW, PK_XU = PK_LM + PK_CK * LMW, PK_B = PK_XU + PK_CK * XUW,
              PK_GT = PK_B + PK_CK * 12, PK_END = PK_GT + PK_CK * LMW;
static_assert(PK_LM % RPU == 0 && PK_XU % RPU == 0 && PK_B % RPU == 0 && PK_GT % RPU == 0, "segments start on 16-byte units");
// stride between the trajectories' records: 16 bytes past a multiple of 256 bytes, so that the PK_G addresses of one broadcast
// read (same offset, different trajectory) fall into different LDS banks
constexpr int PK_STRIDE = ((PK_END - RPU + 16 * RPU - 1) / (16 * RPU)) * (16 * RPU) + RPU;
constexpr int PK_U_KD = PK_LM / RPU, PK_U_LM = PK_XU / RPU, PK_U_XU = PK_B / RPU;   // cumulative unit boundaries
constexpr int PK_UNITS = PK_U_XU + PK_CK * 3 * BROW_UNITS;                          // 16-byte units copied per trajectory and chunk
static_assert(L_FWD + PK_NBUF * PK_G * PK_STRIDE <= LDS_REALS, "packed forward buffers fit the wave's LDS block");

// per-trajectory position in the AL-iLQR iteration; lives in the registers of the trajectory's PK_C lanes and is made
// wave-uniform (through LDS) for the phases that work on one trajectory at a time
template <typename real>
struct GState {
  acc_t Jprev, dV1, dV2, Jw;
  real mu, rho, drho, grad, rho_used, nu[7];
  int N, active, status, outer, it, inner_iters, ls_trials, n_backward, n_forward, bp_restarts, fp_fails, djz, trow, regfail,
      found, jw, slot;
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
TSAT_DEV TPtrs<real> group_ptrs(const KArgs<real>& a, int traj) {
  const int NS = a.N;
  TPtrs<real> p;
  p.XU = (TSAT_GLOBAL real*)(a.XU + (size_t)traj * xu_stride<real>(NS));
  p.KD = (TSAT_GLOBAL real*)(a.KD + (size_t)traj * kd_stride<real>(NS));
  p.LAM = (TSAT_GLOBAL real*)(a.LAM + (size_t)traj * lam_stride<real>(NS));
  p.CAND = (TSAT_GLOBAL real*)(a.CAND + (size_t)traj * a.max_ls * xu_stride<real>(NS));
  p.bt = (const TSAT_GLOBAL real*)(a.BT + (size_t)a.bidx[traj] * a.n_tab * 4);
  return p;
}

// One forward sweep for the PK_G trajectories of the wave: lane (g, c) rolls out alpha = 2^-(c + shift) for trajectory g.
// `live`: this lane's trajectory takes part; N, mu, nu: its horizon, penalty and terminal multipliers. Same per-knot
// arithmetic as forward_sweep (tsat_device.hpp); the knot records of all PK_G trajectories are staged through LDS in
// PK_CK-knot chunks (global_load_lds, lane = 16-byte unit) and read back as PK_G-address broadcasts.
#ifdef TSAT_PK_FWD_NOINLINE
#define TSAT_PK_FWD TSAT_PHASE
#else
#define TSAT_PK_FWD TSAT_FWD
#endif
template <typename real, int INTEG, int DIAGJ, int ES>
TSAT_PK_FWD FwdOut<real> forward_sweep_packed(const KArgs<real>& a, int traj0, int closed, int shift, int n_store, bool live, int N,
                                           real mu, const real nu[7], int term_mask, real max_state) {
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
    tr.hh = (real)0.5 * tr.h; tr.us = (real)a.opt.u_scale;
    tr.tau0 = (double)P[P_TAU0] + (double)P[P_TAU0L]; tr.dtau = (double)P[P_DTAU] + (double)P[P_DTAUL];
    tr.N = N; tr.n_tab = n_tab; tr.bt = nullptr;
  }
  // table of the copy lanes (they serve every trajectory of the wave): horizon and table clock per trajectory, in the Riccati
  // scratch, which is free during a forward sweep
  double* pk_tau0 = reinterpret_cast<double*>(lds + L_ST);
  double* pk_dtau = pk_tau0 + PK_G;
  int* pk_n = reinterpret_cast<int*>(pk_dtau + PK_G);
  TSAT_SYNC_LDS();
  if (myc == 0) { pk_tau0[myg] = tr.tau0; pk_dtau[myg] = tr.dtau; pk_n[myg] = live ? N : 0; }
  TSAT_SYNC_LDS();
  int nmax = 0;
  for (int g = 0; g < PK_G; ++g) nmax = (pk_n[g] > nmax) ? pk_n[g] : nmax;
  real alpha = 1;
  for (int j = 0; j < myc + shift && j < TSAT_MAX_LINESEARCH; ++j) alpha *= (real)0.5;
  const TPtrs<real> pm = group_ptrs<real>(a, traj);
  TSAT_GLOBAL real* Cg = pm.CAND + (size_t)(myc < n_store ? myc : 0) * (size_t)N * XUW;
  const bool store = live && myc < n_store;
  HalfWeights<real> hw;
  for (int i = 0; i < 7; ++i) hw.hQd[i] = (real)0.5 * tr.Qd[i];
  for (int i = 0; i < 3; ++i) hw.hRd[i] = (real)0.5 * tr.Rd[i];
  hw.hmu = (real)0.5 * mu;
  acc_t J = 0;
  real amax = 0;

  // copy of chunk k0 into buffer `fb`: per trajectory, lane = 16-byte unit of its [gains | multipliers | records | field rows]
  auto issue = [&](real* fb, int k0) {
    for (int g = 0; g < PK_G; ++g) {
      const int Ng = pk_n[g];
      const int nk = (Ng - 1 - k0 < PK_CK) ? (Ng - 1 - k0) : PK_CK;     // knots of this trajectory in the chunk (<= 0: none)
      if (nk <= 0) continue;
      const int tg = (traj0 + g <= tmax) ? traj0 + g : tmax;
      const TPtrs<real> pg = group_ptrs<real>(a, tg);
      const double tau0 = pk_tau0[g], dtau = pk_dtau[g];
      real* fbg = fb + g * PK_STRIDE;
      for (int j = 0; j < (PK_UNITS + WAVE - 1) / WAVE; ++j) {
        const int i = lane + WAVE * j;
        const TSAT_GLOBAL real* src = nullptr;
        if (i < PK_U_KD) {
          if (closed && i * RPU < nk * KDW) src = pg.KD + (size_t)k0 * KDW + (size_t)i * RPU;
        } else if (i < PK_U_LM) {
          const int e = i - PK_U_KD;
          if (e * RPU < nk * LMW) src = pg.LAM + (size_t)k0 * LMW + (size_t)e * RPU;
        } else if (i < PK_U_XU) {
          const int e = i - PK_U_LM;
          if (e * RPU < nk * XUW) src = pg.XU + (size_t)k0 * XUW + (size_t)e * RPU;
        } else if (i < PK_UNITS) {
          const int e = i - PK_U_XU, r = e / BROW_UNITS, part = e - r * BROW_UNITS, kk = r / 3, st = r - 3 * kk;
          if (kk < nk) {
            const double rr = floor_(fma_((double)(k0 + kk) + 0.5 * (double)st, dtau, tau0));
            const int row = (rr >= 0.0) ? (rr > (double)(n_tab - 1) ? n_tab - 1 : (int)rr) : 0;
            src = pg.bt + (size_t)row * 4 + (size_t)part * RPU;
          }
        }
        if (src) glds_put<real>(fbg + (size_t)i * RPU, src);
      }
    }
  };
  auto gates = [&](real* fb) {
    for (int e = lane; e < PK_G * PK_CK * LMW; e += WAVE) {
      const int g = e / (PK_CK * LMW), r = e - g * (PK_CK * LMW);
      real* fbg = fb + g * PK_STRIDE;
      fbg[PK_GT + r] = (fbg[PK_LM + r] > 0) ? -inf_<real>() : (real)0;
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
    real* fb = lds + L_FWD + cur * PK_G * PK_STRIDE;
    real* fbn = lds + L_FWD + ((PK_NBUF == 2) ? (1 - cur) : 0) * PK_G * PK_STRIDE;
    if (PK_NBUF == 2 && more) issue(fbn, kn);
    const real* fbg = fb + myg * PK_STRIDE;
    for (int kk = 0; kk < PK_CK; ++kk) {
      const int k = k0 + kk;
      if (live && k < N - 1) {
        const real* xu = fbg + PK_XU + kk * XUW;
        real u[3] = {xu[7], xu[8], xu[9]};
        if (closed) {
          const real* kd = fbg + PK_KD + kk * KDW;
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
        J += (acc_t)stage_cost_gated(tr, hw, x, u, fbg + PK_LM + kk * LMW, fbg + PK_GT + kk * LMW);
        if (store) {
          TSAT_GLOBAL real* cr = Cg + (size_t)k * XUW;
          for (int i = 0; i < 7; ++i) cr[i] = x[i];
          for (int c = 0; c < 3; ++c) cr[7 + c] = u[c];
        }
        const real* br = fbg + PK_B + kk * 12;
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
      TSAT_SYNC();                 // vmcnt(0): the copy has landed
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
  mine.found = 0; mine.jw = 0; mine.slot = 0;

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
  auto backward_for = [&](const TPtrs<real>& p, GState<real>& u) {   // backward sweep with regularisation restarts
    BwdOut<real> bw;
    bw.dV1 = 0; bw.dV2 = 0; bw.pd_ok = 0;
    for (;;) {
      u.n_backward++;
      bw = backward_sweep<real, INTEG, DIAGJ, ES>(p, u.N, n_tab, u.mu, u.rho, tmask);
      if (bw.pd_ok) break;
      u.bp_restarts++;
      u.drho = (u.drho * (real)o.reg_scale > (real)o.reg_scale) ? u.drho * (real)o.reg_scale : (real)o.reg_scale;
      u.rho = (u.rho * u.drho > (real)o.reg_min) ? u.rho * u.drho : (real)o.reg_min;
      if (u.rho > (real)o.reg_max) { u.regfail = 1; break; }
    }
    if (u.regfail) return;
    u.dV1 = bw.dV1; u.dV2 = bw.dV2;
    u.rho_used = u.rho;
    const real inv = (real)1 / (real)o.reg_scale;    // regularisation decrease
    u.drho = (u.drho / (real)o.reg_scale < inv) ? u.drho / (real)o.reg_scale : inv;
    const real r = u.rho * u.drho;
    u.rho = (r > (real)o.reg_min) ? r : (real)0;
    TSAT_SYNC();
  };
  auto start_outer = [&](const TPtrs<real>& p, GState<real>& u, int outer) {   // AL cost of the nominal, fresh regularisation, backward
    u.outer = outer;
    u.Jprev = nominal_cost<real>(p, u.N, u.mu, tmask, 1);
    u.rho = (real)o.reg_init; u.drho = 0; u.djz = 0; u.regfail = 0; u.it = 1;
    backward_for(p, u);
  };
  auto finish = [&](int g, const TPtrs<real>& p, GState<real>& u) {   // final statistics of trajectory g
    TSAT_SYNC();
    const real cmax = violation_and_duals<real>(p, u.N, u.mu, tmask, 0, (real)o.dual_max);
    const acc_t cost = nominal_cost<real>(p, u.N, u.mu, tmask, 0);
    const acc_t cost_al = nominal_cost<real>(p, u.N, u.mu, tmask, 1);
    if (lane == 0) {
      tsat_stats& st = a.stats[traj0 + g];
      st.status = u.status; st.outer_iters = u.outer; st.inner_iters = u.inner_iters; st.ls_trials = u.ls_trials;
      st.n_backward = u.n_backward; st.n_forward = u.n_forward; st.bp_restarts = u.bp_restarts; st.fp_fails = u.fp_fails;
      st.cost = (double)cost; st.cost_al = (double)cost_al; st.c_max = (double)cmax; st.grad = (double)u.grad;
    }
    u.active = 0;
  };
  // Advance trajectory g from "a forward sweep has just been evaluated" (after_forward) or from the very start to the point
  // where it needs the next forward sweep (u.active stays 1, gains of a fresh backward sweep in HBM) or is finished.
  auto advance = [&](int g, GState<real>& u, bool after_forward) {
    const int traj = traj0 + g;
    const TPtrs<real> p = group_ptrs<real>(a, traj);
    double* trace = a.trace ? a.trace + (size_t)traj * a.trace_rows * 8 : nullptr;
    // trajectory constants + terminal multipliers into the wave's LDS block: what the per-trajectory phase functions read
    TSAT_SYNC();
    stage_traj<real>((const TSAT_GLOBAL real*)(a.P + (size_t)traj * PSTRIDE), (real)o.u_scale);
    if (lane < 7) lds[L_NU + lane] = u.nu[lane];
    TSAT_SYNC();
    bool inner_over = false;
    if (!after_forward) {
      (void)adopt_and_gradient<real>(p, u.N, 0);        // the open-loop rollout becomes the nominal trajectory
      TSAT_SYNC();
      start_outer(p, u, 1);
      inner_over = u.regfail != 0;
    } else {
      acc_t J;
      if (u.found) {
        J = u.Jw;
        u.ls_trials += u.jw + 1;
        u.grad = adopt_and_gradient<real>(p, u.N, u.slot);
      } else {
        J = u.Jprev;
        u.ls_trials += max_ls;
        u.fp_fails++;
        u.drho = (u.drho * (real)o.reg_scale > (real)o.reg_scale) ? u.drho * (real)o.reg_scale : (real)o.reg_scale;
        u.rho = (u.rho * u.drho > (real)o.reg_min) ? u.rho * u.drho : (real)o.reg_min;
        u.rho += (real)o.reg_fp;
        u.grad = adopt_and_gradient<real>(p, u.N, -1);
      }
      TSAT_SYNC();
      acc_t dJ = J - u.Jprev;
      dJ = dJ < 0 ? -dJ : dJ;
      if (trace && lane == 0 && u.trow < a.trace_rows) {
        double* r = trace + 8 * u.trow;
        r[0] = u.outer; r[1] = u.it; r[2] = (double)u.Jprev; r[3] = (double)J; r[4] = u.found ? u.jw : -1;
        r[5] = (double)u.rho_used; r[6] = (double)u.dV1; r[7] = (double)u.dV2;
      }
      u.trow++;
      u.Jprev = J;
      u.djz = (dJ == 0) ? u.djz + 1 : 0;
      u.inner_iters++;
      inner_over = (0 < dJ && dJ < (acc_t)o.cost_tol) || (u.grad < (real)o.grad_tol) || (u.djz > o.dj_counter_limit) ||
                   (u.it >= o.max_inner);
      if (!inner_over) {
        u.it++;
        backward_for(p, u);
        inner_over = u.regfail != 0;
      }
    }
    while (inner_over) {      // end of an inner loop: outer-loop bookkeeping, possibly straight into the next outer iteration
      inner_over = false;
      TSAT_SYNC();
      const real cmax = violation_and_duals<real>(p, u.N, u.mu, tmask, 0, (real)o.dual_max);
      if (u.regfail) { u.status = TSAT_REG_FAIL; finish(g, p, u); break; }
      if (cmax < (real)o.constraint_tol) { u.status = TSAT_CONVERGED; finish(g, p, u); break; }
      if (u.outer == o.max_outer) { finish(g, p, u); break; }
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
      inner_over = u.regfail != 0;
    }
  };
  auto any_lane = [&](bool f) { return wave_first<real>(f, lds + L_RED) < WAVE; };

  // ---- open-loop rollout of U0 for every trajectory of the wave, then the first backward sweeps ------------------------
  {
    const FwdOut<real> f0 = forward_sweep_packed<real, INTEG, DIAGJ, ES>(a, traj0, 0, 0, 1, mine.active != 0, mine.N, mine.mu, mine.nu,
                                                                         tmask, max_state);
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
        (void)adopt_and_gradient<real>(p, u.N, 0);
        finish(g, p, u);
      } else {
        advance(g, u, false);
      }
      if (myg == g) mine = u;
    }
  }
  // ---- main loop: one forward sweep for all trajectories that are still iterating, then each of them moves on ------------
  while (any_lane(mine.active != 0)) {
    mine.found = 0;
    for (int shift = 0; shift < max_ls; shift += n_store) {   // n_store candidates alpha = 2^-(shift + c) per sweep
      const bool live = mine.active && !mine.found;
      if (!any_lane(live)) break;
      TSAT_SYNC();
      const FwdOut<real> fw = forward_sweep_packed<real, INTEG, DIAGJ, ES>(a, traj0, 1, shift, n_store, live, mine.N, mine.mu, mine.nu,
                                                                           tmask, max_state);
      if (live) mine.n_forward++;
      // first accepted candidate of each trajectory (sequential backtracking picks exactly this one)
      const int ci = myc + shift;
      acc_t alpha = 1;
      for (int j = 0; j < ci && j < TSAT_MAX_LINESEARCH; ++j) alpha *= (acc_t)0.5;
      const acc_t Jc = fw.J;
      const acc_t expected = -alpha * (mine.dV1 + alpha * mine.dV2);
      const acc_t z = (expected > 0) ? (mine.Jprev - Jc) / expected : (acc_t)-1;
      const bool acc = live && (ci < max_ls) && (myc < n_store) && fw.ok &&
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
      if (live && jl < PK_C) { mine.found = 1; mine.jw = shift + jl; mine.slot = jl; mine.Jw = Jwin; }
    }
    TSAT_SYNC();
    for (int g = 0; g < ntr; ++g) {
      GState<real> u = gstate_bcast(mine, g * PK_C);
      if (u.active) advance(g, u, true);
      if (myg == g) mine = u;
    }
  }
}

}  // namespace tsat
