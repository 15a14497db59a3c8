/*
 * tsat_oracle.cpp — CPU restatement (fp64, scalar C++) of the TortoiseSat.jl attitude-slew hot path.
 *
 * >>> TEST INFRASTRUCTURE ONLY. <<<  Nothing under oracle/ is imported, linked or executed by the product
 * (tortoisesat.jl_amd/, libtortoise_hip.so). Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load liboracle.so — as the checker / the CPU baseline, never as the thing shipped.
 *
 * >>> PARITY UNPINNED. <<<  The reference repository has no tests, no golden vectors and no fixtures, and no
 * Julia interpreter exists in the authoring container or on the GPU box, so this restatement cannot be pinned
 * against outputs of the reference itself.  What IS pinned (tests/test_oracle_*.py): every function whose text
 * is physically in the reference (A1..A7 below) is checked against an independent NumPy transcription of the
 * same reference lines, against analytic invariants, and against SciPy's DARE.  The AL-iLQR driver (A10) lives
 * in the un-vendored dependency TrajectoryOptimization.jl v0.1.2 (Manifest.toml:994-998, git-tree-sha1
 * b933cc4febea19e643ea6fc4a17dfb53d1f926f3); it is restated here from the published algorithm
 * (Howell, Jackson, Manchester, "ALTRO", IROS 2019) and anchored on the reference's
 * call sites (src/TortoiseSat.jl:145-199, src/monte_carlo.jl:158-198).
 *
 * Map of functions to reference text:
 *   A1  orc_qmult/orc_qrot/orc_qinv/orc_hat/orc_gmat   src/DerivFunction.jl:50-56, src/attitude_controller.jl:69,164-176
 *   A2  orc_deriv8                                      src/DerivFunction.jl:1-48 (8-state, table lookup)
 *   A3  orc_attitude_dynamics                           src/attitude_dynamics.jl:2-24
 *   A4  orc_rk_step, orc_rk_generic                     src/attitude_controller.jl:122-132 (rk4), :178-187 (rk3)
 *   A5  orc_discrete_jacobian                           src/attitude_controller.jl:95-119 (forward-mode duals, as ForwardDiff)
 *   A6  orc_reduce_error_state, orc_quaternion_error,
 *       orc_quaternion_expansion                        src/attitude_controller.jl:50-81, src/quaternion_toolbox.jl:15-75
 *   A7  orc_tvlqr_riccati                               src/attitude_controller.jl:83-92
 *   A8/A9/A10 orc_solve_batch                           src/TortoiseSat.jl:157-199 + SURVEY.md Appendix A
 *             (its RESULT is pinned against SciPy's SLSQP on an independent NumPy rollout, tests/test_golden_and_emu.py)
 *   §8f-1 orc_kep_eci, orc_igrf12, orc_btable_batch     src/kep_ECI.jl:1-35, src/OrbitPlotter.jl:1-52, src/igrf.jl:70-274,
 *                                                       src/legendre.jl:254-292, src/dlegendre.jl:221-309, src/magnetic_toolbox.jl:33-106
 *   §8f-2 orc_horizon_batch                             src/magnetic_toolbox.jl:1-31
 *   §8f-3 orc_tvlqr_batch (+ orc_philox4x32_10,
 *         orc_plant_noise for noise_mode = 1)           src/attitude_controller.jl:1-119, src/simulator.jl, src/gain_simulator.jl,
 *                                                       src/monte_carlo.jl:242-262
 *   orc_mpc_batch                                       no reference text (BASELINE.json configs[4]; see tsat_mpc_run in the header)
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <limits>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/tortoise_hip.h"
#include "../include/igrf12_2015_coeffs.h"   // IGRF-12 model constants (data)

namespace {

// ------------------------------------------------------------------------------------------------
// forward-mode dual numbers (what ForwardDiff.jacobian! does at src/attitude_controller.jl:103)
// ------------------------------------------------------------------------------------------------
template <int NP>
struct Dual {
  double v;
  double p[NP];
  Dual() : v(0) { for (int i = 0; i < NP; ++i) p[i] = 0; }
  Dual(double a) : v(a) { for (int i = 0; i < NP; ++i) p[i] = 0; }
};
template <int NP> inline Dual<NP> operator+(const Dual<NP>& a, const Dual<NP>& b) { Dual<NP> r; r.v = a.v + b.v; for (int i = 0; i < NP; ++i) r.p[i] = a.p[i] + b.p[i]; return r; }
template <int NP> inline Dual<NP> operator-(const Dual<NP>& a, const Dual<NP>& b) { Dual<NP> r; r.v = a.v - b.v; for (int i = 0; i < NP; ++i) r.p[i] = a.p[i] - b.p[i]; return r; }
template <int NP> inline Dual<NP> operator-(const Dual<NP>& a) { Dual<NP> r; r.v = -a.v; for (int i = 0; i < NP; ++i) r.p[i] = -a.p[i]; return r; }
template <int NP> inline Dual<NP> operator*(const Dual<NP>& a, const Dual<NP>& b) { Dual<NP> r; r.v = a.v * b.v; for (int i = 0; i < NP; ++i) r.p[i] = a.p[i] * b.v + a.v * b.p[i]; return r; }
template <int NP> inline Dual<NP> operator/(const Dual<NP>& a, const Dual<NP>& b) { Dual<NP> r; r.v = a.v / b.v; for (int i = 0; i < NP; ++i) r.p[i] = (a.p[i] - r.v * b.p[i]) / b.v; return r; }
template <int NP> inline Dual<NP> operator*(double a, const Dual<NP>& b) { return Dual<NP>(a) * b; }
template <int NP> inline Dual<NP> operator*(const Dual<NP>& a, double b) { return a * Dual<NP>(b); }
template <int NP> inline Dual<NP> operator/(const Dual<NP>& a, double b) { return a / Dual<NP>(b); }
template <int NP> inline Dual<NP> dsqrt(const Dual<NP>& a) { Dual<NP> r; r.v = std::sqrt(a.v); for (int i = 0; i < NP; ++i) r.p[i] = a.p[i] / (2.0 * r.v); return r; }
inline double dsqrt(double a) { return std::sqrt(a); }

// ------------------------------------------------------------------------------------------------
// A1: quaternion algebra, scalar-first
// ------------------------------------------------------------------------------------------------
template <class S> inline void cross3(const S a[3], const S b[3], S o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// src/DerivFunction.jl:54-56  qmult(q1,q2) = [s1 s2 - v1'v2 ; s1 v2 + s2 v1 + v1 x v2]
template <class S> inline void qmult(const S q1[4], const S q2[4], S o[4]) {
  S c[3];
  cross3(q1 + 1, q2 + 1, c);
  S s = q1[0] * q2[0] - (q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3]);
  for (int i = 0; i < 3; ++i) o[i + 1] = q1[0] * q2[i + 1] + q2[0] * q1[i + 1] + c[i];
  o[0] = s;
}
// src/DerivFunction.jl:50-52  qrot(q,r) = r + 2 v x (v x r + s r)
template <class S> inline void qrot(const S q[4], const S r[3], S o[3]) {
  S t[3], c[3];
  cross3(q + 1, r, t);
  for (int i = 0; i < 3; ++i) t[i] = t[i] + q[0] * r[i];
  cross3(q + 1, t, c);
  for (int i = 0; i < 3; ++i) o[i] = r[i] + 2.0 * c[i];
}

struct Phys {
  double J[9];     // J(r,c) = J[r + 3c] (column-major, as handed over the ABI)
  double Jinv[9];  // same layout
  double u_scale;
};

// closed-form 3x3 inverse (adjugate); replaces inv(p.J) of src/DerivFunction.jl:41
void inv3(const double* M, double* Mi) {
  auto m = [&](int r, int c) { return M[r + 3 * c]; };
  double c00 = m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1);
  double c01 = m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2);
  double c02 = m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0);
  double det = m(0, 0) * c00 + m(0, 1) * c01 + m(0, 2) * c02;
  double id = 1.0 / det;
  Mi[0 + 3 * 0] = c00 * id;
  Mi[1 + 3 * 0] = c01 * id;
  Mi[2 + 3 * 0] = c02 * id;
  Mi[0 + 3 * 1] = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * id;
  Mi[1 + 3 * 1] = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * id;
  Mi[2 + 3 * 1] = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * id;
  Mi[0 + 3 * 2] = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * id;
  Mi[1 + 3 * 2] = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * id;
  Mi[2 + 3 * 2] = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * id;
}

// ------------------------------------------------------------------------------------------------
// A2/A3: continuous dynamics on the 7-state with the table row `b` already looked up.
// src/DerivFunction.jl:4-44: q normalised inside f (:5), qdot = 0.5 qmult(q,[0;w]) (:24),
// B_B = qrot(q, b) (:28), tau_c = cross(u*1e-2, B_B) (:37), wdot = inv(J)(tau_c - w x Jw) (:41).
// ------------------------------------------------------------------------------------------------
template <class S>
void dyn7(const S x[7], const S u[3], const double b[3], const Phys& ph, S xd[7]) {
  S w[3] = {x[0], x[1], x[2]};
  S nq = dsqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  S q[4] = {x[3] / nq, x[4] / nq, x[5] / nq, x[6] / nq};
  S p[4] = {S(0.0), w[0], w[1], w[2]};
  S qd[4];
  qmult(q, p, qd);
  S bs[3] = {S(b[0]), S(b[1]), S(b[2])};
  S BB[3];
  qrot(q, bs, BB);
  S us[3] = {u[0] * ph.u_scale, u[1] * ph.u_scale, u[2] * ph.u_scale};
  S tau[3];
  cross3(us, BB, tau);
  S Jw[3], wJw[3];
  for (int r = 0; r < 3; ++r) Jw[r] = ph.J[r + 0] * w[0] + ph.J[r + 3] * w[1] + ph.J[r + 6] * w[2];
  cross3(w, Jw, wJw);
  S rhs[3] = {tau[0] - wJw[0], tau[1] - wJw[1], tau[2] - wJw[2]};
  for (int r = 0; r < 3; ++r) xd[r] = ph.Jinv[r + 0] * rhs[0] + ph.Jinv[r + 3] * rhs[1] + ph.Jinv[r + 6] * rhs[2];
  for (int i = 0; i < 4; ++i) xd[3 + i] = 0.5 * qd[i];
}

// ------------------------------------------------------------------------------------------------
// A4: discretisers. b0/b1/b2 = table rows at stage times tau, tau + dtau/2, tau + dtau.
// rk3: src/attitude_controller.jl:178-187; rk4: :122-132.
// ------------------------------------------------------------------------------------------------
template <class S>
void rk_step(int integ, const S x[7], const S u[3], const double* b0, const double* b1, const double* b2,
             double h, const Phys& ph, S xn[7]) {
  S k1[7], k2[7], k3[7], k4[7], t[7];
  dyn7(x, u, b0, ph, k1);
  for (int i = 0; i < 7; ++i) { k1[i] = k1[i] * h; t[i] = x[i] + k1[i] / 2.0; }
  dyn7(t, u, b1, ph, k2);
  if (integ == 3) {
    for (int i = 0; i < 7; ++i) { k2[i] = k2[i] * h; t[i] = x[i] - k1[i] + 2.0 * k2[i]; }
    dyn7(t, u, b2, ph, k3);
    for (int i = 0; i < 7; ++i) { k3[i] = k3[i] * h; xn[i] = x[i] + (k1[i] + 4.0 * k2[i] + k3[i]) / 6.0; }
  } else {
    for (int i = 0; i < 7; ++i) { k2[i] = k2[i] * h; t[i] = x[i] + k2[i] / 2.0; }
    dyn7(t, u, b1, ph, k3);
    for (int i = 0; i < 7; ++i) { k3[i] = k3[i] * h; t[i] = x[i] + k3[i]; }
    dyn7(t, u, b2, ph, k4);
    for (int i = 0; i < 7; ++i) { k4[i] = k4[i] * h; xn[i] = x[i] + (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) / 6.0; }
  }
}

// A5: A = d x+/dx (7x7, row-major A[i*7+j]), B = d x+/du (7x3, row-major B[i*3+a])
void discrete_jacobian(int integ, const double x[7], const double u[3], const double* b0, const double* b1,
                       const double* b2, double h, const Phys& ph, double* A, double* B) {
  typedef Dual<10> D;
  D xd[7], ud[3], xn[7];
  for (int i = 0; i < 7; ++i) { xd[i] = D(x[i]); xd[i].p[i] = 1.0; }
  for (int a = 0; a < 3; ++a) { ud[a] = D(u[a]); ud[a].p[7 + a] = 1.0; }
  rk_step<D>(integ, xd, ud, b0, b1, b2, h, ph, xn);
  for (int i = 0; i < 7; ++i) {
    for (int j = 0; j < 7; ++j) A[i * 7 + j] = xn[i].p[j];
    for (int a = 0; a < 3; ++a) B[i * 3 + a] = xn[i].p[7 + a];
  }
}

// ------------------------------------------------------------------------------------------------
// problem description of one trajectory
// ------------------------------------------------------------------------------------------------
struct Traj {
  int N, n_tab, integ;
  const double *x0, *xf, *Bt;  // Bt: [n_tab][3]
  double tau0, dtau, dt;
  Phys ph;
  const double *Qd, *Qfd, *Rd, *ulo, *uhi;
};
inline const double* brow(const Traj& t, int k, double c) {
  double r = std::floor(std::fma((double)k + c, t.dtau, t.tau0));
  long i = (long)r;
  if (!(r >= 0.0)) i = 0;  // also catches NaN
  if (i > t.n_tab - 1) i = t.n_tab - 1;
  return t.Bt + 3 * i;
}
inline void step(const Traj& t, int k, const double* x, const double* u, double* xn) {
  rk_step<double>(t.integ, x, u, brow(t, k, 0.0), brow(t, k, 0.5), brow(t, k, 1.0), t.dt, t.ph, xn);
}

struct AL {
  std::vector<double> lam;  // 6 per knot k < N-1 : [upper(3), lower(3)]
  double nu[7];
  double mu;
};

// A8 + A9: stage cost 0.5 (x-xf)'Q(x-xf) + 0.5 u'Ru (src/TortoiseSat.jl:169) + AL terms of the control box
// (src/TortoiseSat.jl:178,185-187): lam'c + 0.5 c' I_mu c, I_mu,ii = mu if c_i > 0 or lam_i > 0.
double stage_cost(const Traj& t, const double* x, const double* u, const double* lam, double mu, bool with_al) {
  double l = 0.0;
  for (int i = 0; i < 7; ++i) { double e = x[i] - t.xf[i]; l += 0.5 * t.Qd[i] * e * e; }
  for (int a = 0; a < 3; ++a) l += 0.5 * t.Rd[a] * u[a] * u[a];
  if (with_al) {
    for (int a = 0; a < 3; ++a) {
      double c = u[a] - t.uhi[a], lm = lam[a];
      l += lm * c;
      if (c > 0.0 || lm > 0.0) l += 0.5 * mu * c * c;
    }
    for (int a = 0; a < 3; ++a) {
      double c = t.ulo[a] - u[a], lm = lam[3 + a];
      l += lm * c;
      if (c > 0.0 || lm > 0.0) l += 0.5 * mu * c * c;
    }
  }
  return l;
}
// terminal cost 0.5 (x-xf)'Qf(x-xf) + goal-constraint AL terms (src/TortoiseSat.jl:182,188)
double term_cost(const Traj& t, const double* x, const double* nu, double mu, int mask, bool with_al) {
  double l = 0.0;
  for (int i = 0; i < 7; ++i) { double e = x[i] - t.xf[i]; l += 0.5 * t.Qfd[i] * e * e; }
  if (with_al)
    for (int i = 0; i < 7; ++i)
      if (mask >> i & 1) { double e = x[i] - t.xf[i]; l += nu[i] * e + 0.5 * mu * e * e; }
  return l;
}
double total_cost(const Traj& t, const tsat_options& o, const double* X, const double* U, const AL& al, bool with_al) {
  double J = 0.0;
  for (int k = 0; k < t.N - 1; ++k) J += stage_cost(t, X + 7 * k, U + 3 * k, al.lam.data() + 6 * k, al.mu, with_al);
  J += term_cost(t, X + 7 * (t.N - 1), al.nu, al.mu, o.terminal_mask, with_al);
  return J;
}
double max_violation(const Traj& t, const tsat_options& o, const double* X, const double* U) {
  double c = 0.0;
  for (int k = 0; k < t.N - 1; ++k)
    for (int a = 0; a < 3; ++a) {
      c = std::max(c, U[3 * k + a] - t.uhi[a]);
      c = std::max(c, t.ulo[a] - U[3 * k + a]);
    }
  for (int i = 0; i < 7; ++i)
    if (o.terminal_mask >> i & 1) c = std::max(c, std::fabs(X[7 * (t.N - 1) + i] - t.xf[i]));
  return c;
}

// state difference fed to the gains: plain x - xbar, or with error_state = 1 the reference's hook
// quaternion_error(X1, X2) = [dw; MRP(q2^-1 (x) q1)] (src/quaternion_toolbox.jl:58-75), a 6-vector
void state_diff(int es, const double* xnew, const double* xnom, double* dx) {
  if (!es) { for (int j = 0; j < 7; ++j) dx[j] = xnew[j] - xnom[j]; return; }
  for (int j = 0; j < 3; ++j) dx[j] = xnew[j] - xnom[j];
  double qi[4] = {xnom[3], -xnom[4], -xnom[5], -xnom[6]}, qe[4];
  qmult<double>(qi, xnew + 3, qe);
  for (int j = 0; j < 3; ++j) dx[3 + j] = qe[1 + j] / (1.0 + qe[0]);
  dx[6] = 0.0;
}

// forward rollout with the iLQR policy u = ubar + K dx + alpha d   (Appendix A "forward"). K rows are stored with
// stride 7 in both modes (column 6 is zero in error-state mode).
bool rollout(const Traj& t, const tsat_options& o, const double* X, const double* U, const double* K,
             const double* d, double alpha, bool closed, double* Xc, double* Uc) {
  for (int i = 0; i < 7; ++i) Xc[i] = t.x0[i];
  bool ok = true;
  const int nh = o.error_state ? 6 : 7;
  for (int k = 0; k < t.N - 1; ++k) {
    double* xc = Xc + 7 * k;
    double* uc = Uc + 3 * k;
    double dx[7];
    if (closed) state_diff(o.error_state, xc, X + 7 * k, dx);
    for (int a = 0; a < 3; ++a) {
      double v = U[3 * k + a];
      if (closed) {
        for (int j = 0; j < nh; ++j) v += K[(k * 3 + a) * 7 + j] * dx[j];
        v += alpha * d[3 * k + a];
      }
      uc[a] = v;
    }
    for (int i = 0; i < 7; ++i) if (!(std::fabs(xc[i]) <= o.max_state)) ok = false;
    for (int a = 0; a < 3; ++a) if (!(std::fabs(uc[a]) <= o.max_state)) ok = false;
    step(t, k, xc, uc, xc + 7);
  }
  for (int i = 0; i < 7; ++i) if (!(std::fabs(Xc[7 * (t.N - 1) + i]) <= o.max_state)) ok = false;
  return ok;
}

// E(q) = blkdiag(I3, G(q)) (7x6 row-major) with the raw state quaternion (src/attitude_controller.jl:61-78)
void emat(const double* q, double* E) {
  for (int i = 0; i < 42; ++i) E[i] = 0.0;
  for (int i = 0; i < 3; ++i) E[i * 6 + i] = 1.0;
  const double s = q[0], v0 = q[1], v1 = q[2], v2 = q[3];
  const double G[12] = {-v0, -v1, -v2, s, -v2, v1, v2, s, -v0, -v1, v0, s};  // rows: [-v'; s I + hat(v)]
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 3; ++c) E[(3 + r) * 6 + 3 + c] = G[r * 3 + c];
}

// backward Riccati sweep of iLQR on the AL cost (Appendix A "backward"); false if some Quu_reg is not PD.
// error_state = 0: 7-state differences. error_state = 1: the reference's quaternion hooks — dynamics blocks reduced to
// E(q_{k+1})' A E(q_k), E(q_{k+1})' B (src/attitude_controller.jl:59-81), cost expansion projected through E(q_k)
// (src/quaternion_toolbox.jl:15-50); everything runs on nh = 6 error coordinates. K is written with row stride 7.
bool backward(const Traj& t, const tsat_options& o, const double* X, const double* U, const AL& al, double rho,
              double* K, double* d, double dV[2]) {
  const int N = t.N;
  const int es = o.error_state, nh = es ? 6 : 7;
  double S[49], s[7];
  for (int i = 0; i < 49; ++i) S[i] = 0.0;
  const double* xN = X + 7 * (N - 1);
  {
    double Sd[7], sf[7];
    for (int i = 0; i < 7; ++i) {
      double e = xN[i] - t.xf[i];
      Sd[i] = t.Qfd[i];
      sf[i] = t.Qfd[i] * e;
      if (o.terminal_mask >> i & 1) { Sd[i] += al.mu; sf[i] += al.nu[i] + al.mu * e; }
    }
    if (!es) {
      for (int i = 0; i < 7; ++i) { S[i * 7 + i] = Sd[i]; s[i] = sf[i]; }
    } else {
      double E[42];
      emat(xN + 3, E);
      for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 7; ++m) a += E[m * 6 + i] * Sd[m] * E[m * 6 + j]; S[i * 7 + j] = a; }
        double a = 0; for (int m = 0; m < 7; ++m) a += E[m * 6 + i] * sf[m]; s[i] = a;
      }
    }
  }
  dV[0] = dV[1] = 0.0;
  double A7[49], B7[21], A[49], B[21];
  for (int k = N - 2; k >= 0; --k) {
    const double* x = X + 7 * k;
    const double* u = U + 3 * k;
    discrete_jacobian(t.integ, x, u, brow(t, k, 0.0), brow(t, k, 0.5), brow(t, k, 1.0), t.dt, t.ph, A7, B7);
    double lx7[7], lx[7], lu[3], luu[3], Q0[49];
    for (int i = 0; i < 49; ++i) Q0[i] = 0.0;
    for (int i = 0; i < 7; ++i) lx7[i] = t.Qd[i] * (x[i] - t.xf[i]);
    if (!es) {
      for (int i = 0; i < 49; ++i) A[i] = A7[i];
      for (int i = 0; i < 21; ++i) B[i] = B7[i];
      for (int i = 0; i < 7; ++i) { lx[i] = lx7[i]; Q0[i * 7 + i] = t.Qd[i]; }
    } else {
      double Ek[42], En[42], T[42];
      emat(x + 3, Ek);
      emat(x + 7 + 3, En);
      for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 7; ++m) a += A7[i * 7 + m] * Ek[m * 6 + j]; T[i * 6 + j] = a; }
      for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 7; ++m) a += En[m * 6 + i] * T[m * 6 + j]; A[i * 7 + j] = a; }
        for (int c = 0; c < 3; ++c) { double a = 0; for (int m = 0; m < 7; ++m) a += En[m * 6 + i] * B7[m * 3 + c]; B[i * 3 + c] = a; }
        double a = 0; for (int m = 0; m < 7; ++m) a += Ek[m * 6 + i] * lx7[m]; lx[i] = a;
        for (int j = 0; j < 6; ++j) { double b = 0; for (int m = 0; m < 7; ++m) b += Ek[m * 6 + i] * t.Qd[m] * Ek[m * 6 + j]; Q0[i * 7 + j] = b; }
      }
    }
    const double* lam = al.lam.data() + 6 * k;
    for (int a = 0; a < 3; ++a) {
      lu[a] = t.Rd[a] * u[a];
      luu[a] = t.Rd[a];
      double c = u[a] - t.uhi[a], lm = lam[a];
      bool act = (c > 0.0 || lm > 0.0);
      lu[a] += lm + (act ? al.mu * c : 0.0);
      if (act) luu[a] += al.mu;
      c = t.ulo[a] - u[a]; lm = lam[3 + a];
      act = (c > 0.0 || lm > 0.0);
      lu[a] -= lm + (act ? al.mu * c : 0.0);
      if (act) luu[a] += al.mu;
    }
    // SA = S A (nh x nh), SB = S B (nh x 3); all blocks use row stride 7 / 3
    double SA[49], SB[21];
    for (int i = 0; i < nh; ++i) {
      for (int j = 0; j < nh; ++j) { double a = 0; for (int m = 0; m < nh; ++m) a += S[i * 7 + m] * A[m * 7 + j]; SA[i * 7 + j] = a; }
      for (int c = 0; c < 3; ++c) { double a = 0; for (int m = 0; m < nh; ++m) a += S[i * 7 + m] * B[m * 3 + c]; SB[i * 3 + c] = a; }
    }
    double Qx[7], Qu[3], Qxx[49], Quu[9], Qux[21];
    for (int j = 0; j < nh; ++j) { double a = lx[j]; for (int m = 0; m < nh; ++m) a += A[m * 7 + j] * s[m]; Qx[j] = a; }
    for (int c = 0; c < 3; ++c) { double a = lu[c]; for (int m = 0; m < nh; ++m) a += B[m * 3 + c] * s[m]; Qu[c] = a; }
    for (int i = 0; i < nh; ++i)
      for (int j = 0; j < nh; ++j) { double a = Q0[i * 7 + j]; for (int m = 0; m < nh; ++m) a += A[m * 7 + i] * SA[m * 7 + j]; Qxx[i * 7 + j] = a; }
    for (int c = 0; c < 3; ++c)
      for (int e = 0; e < 3; ++e) { double a = (c == e) ? luu[c] : 0.0; for (int m = 0; m < nh; ++m) a += B[m * 3 + c] * SB[m * 3 + e]; Quu[c * 3 + e] = a; }
    for (int c = 0; c < 3; ++c)
      for (int j = 0; j < nh; ++j) { double a = 0; for (int m = 0; m < nh; ++m) a += B[m * 3 + c] * SA[m * 7 + j]; Qux[c * 7 + j] = a; }
    // control regularisation + Sylvester PD test + adjugate inverse of the symmetric 3x3
    double q00 = Quu[0] + rho, q11 = Quu[4] + rho, q22 = Quu[8] + rho;
    double q10 = 0.5 * (Quu[3] + Quu[1]), q20 = 0.5 * (Quu[6] + Quu[2]), q21 = 0.5 * (Quu[7] + Quu[5]);
    double c00 = q11 * q22 - q21 * q21;
    double c01 = q20 * q21 - q10 * q22;
    double c02 = q10 * q21 - q20 * q11;
    double c11 = q00 * q22 - q20 * q20;
    double c12 = q10 * q20 - q00 * q21;
    double c22 = q00 * q11 - q10 * q10;
    double det = q00 * c00 + q10 * c01 + q20 * c02;
    if (!(q00 > 0.0 && c22 > 0.0 && det > 0.0)) return false;
    double id = 1.0 / det;
    double Qi[9] = {c00 * id, c01 * id, c02 * id, c01 * id, c11 * id, c12 * id, c02 * id, c12 * id, c22 * id};
    double* Kk = K + (size_t)k * 21;
    double* dk = d + (size_t)k * 3;
    for (int c = 0; c < 3; ++c) {
      for (int j = 0; j < 7; ++j) Kk[c * 7 + j] = 0.0;
      for (int j = 0; j < nh; ++j) Kk[c * 7 + j] = -(Qi[c * 3 + 0] * Qux[0 * 7 + j] + Qi[c * 3 + 1] * Qux[1 * 7 + j] + Qi[c * 3 + 2] * Qux[2 * 7 + j]);
      dk[c] = -(Qi[c * 3 + 0] * Qu[0] + Qi[c * 3 + 1] * Qu[1] + Qi[c * 3 + 2] * Qu[2]);
    }
    // cost-to-go update (literal Appendix A form, un-regularised Quu)
    double QuuK[21], Quud[3];
    for (int c = 0; c < 3; ++c) {
      for (int j = 0; j < nh; ++j) QuuK[c * 7 + j] = Quu[c * 3 + 0] * Kk[0 * 7 + j] + Quu[c * 3 + 1] * Kk[1 * 7 + j] + Quu[c * 3 + 2] * Kk[2 * 7 + j];
      Quud[c] = Quu[c * 3 + 0] * dk[0] + Quu[c * 3 + 1] * dk[1] + Quu[c * 3 + 2] * dk[2];
    }
    double Sn[49], sn[7];
    for (int i = 0; i < nh; ++i) {
      double a = Qx[i];
      for (int c = 0; c < 3; ++c) a += Kk[c * 7 + i] * Quud[c] + Kk[c * 7 + i] * Qu[c] + Qux[c * 7 + i] * dk[c];
      sn[i] = a;
      for (int j = 0; j < nh; ++j) {
        double b = Qxx[i * 7 + j];
        for (int c = 0; c < 3; ++c) b += Kk[c * 7 + i] * QuuK[c * 7 + j] + Kk[c * 7 + i] * Qux[c * 7 + j] + Qux[c * 7 + i] * Kk[c * 7 + j];
        Sn[i * 7 + j] = b;
      }
    }
    for (int i = 0; i < nh; ++i) {
      s[i] = sn[i];
      for (int j = 0; j < nh; ++j) S[i * 7 + j] = 0.5 * (Sn[i * 7 + j] + Sn[j * 7 + i]);
    }
    dV[0] += dk[0] * Qu[0] + dk[1] * Qu[1] + dk[2] * Qu[2];
    dV[1] += 0.5 * (dk[0] * Quud[0] + dk[1] * Quud[1] + dk[2] * Quud[2]);
  }
  return true;
}

inline void reg_increase(const tsat_options& o, double& rho, double& drho) {
  drho = std::max(drho * o.reg_scale, o.reg_scale);
  rho = std::max(rho * drho, o.reg_min);
}
inline void reg_decrease(const tsat_options& o, double& rho, double& drho) {
  drho = std::min(drho / o.reg_scale, 1.0 / o.reg_scale);
  double r = rho * drho;
  rho = (r > o.reg_min) ? r : 0.0;
}

struct Work {
  std::vector<double> X, U, Xc, Uc, K, d;
};

void solve_one(const Traj& t, const tsat_options& o, const double* U0, double* Xout, double* Uout, double* Kout,
               tsat_stats* st, double* trace, int trace_rows) {
  const int N = t.N;
  Work w;
  w.X.assign((size_t)7 * N, 0.0); w.U.assign((size_t)3 * (N - 1), 0.0);
  w.Xc.assign((size_t)7 * N, 0.0); w.Uc.assign((size_t)3 * (N - 1), 0.0);
  w.K.assign((size_t)21 * (N - 1), 0.0); w.d.assign((size_t)3 * (N - 1), 0.0);
  AL al;
  al.lam.assign((size_t)6 * (N - 1), 0.0);
  for (int i = 0; i < 7; ++i) al.nu[i] = 0.0;
  al.mu = o.penalty_init;
  std::memset(st, 0, sizeof(*st));
  int trow = 0;

  for (size_t i = 0; i < w.U.size(); ++i) w.U[i] = U0[i];
  bool ok0 = rollout(t, o, w.X.data(), w.U.data(), nullptr, nullptr, 0.0, false, w.Xc.data(), w.Uc.data());
  w.X = w.Xc;  // (Uc == U for an open-loop rollout)
  st->n_forward = 1;
  double J0 = total_cost(t, o, w.X.data(), w.U.data(), al, true);
  if (!ok0 || !std::isfinite(J0)) {
    st->status = TSAT_DIVERGED;
  } else {
    st->status = TSAT_MAX_OUTER;
    for (int outer = 1; outer <= o.max_outer; ++outer) {
      // ---------------- inner iLQR on the AL cost ----------------
      double Jprev = total_cost(t, o, w.X.data(), w.U.data(), al, true);
      double rho = o.reg_init, drho = 0.0;
      int djz = 0;
      bool regfail = false;
      for (int it = 1; it <= o.max_inner; ++it) {
        double dV[2];
        for (;;) {
          st->n_backward++;
          if (backward(t, o, w.X.data(), w.U.data(), al, rho, w.K.data(), w.d.data(), dV)) break;
          st->bp_restarts++;
          reg_increase(o, rho, drho);
          if (rho > o.reg_max) { regfail = true; break; }
        }
        if (regfail) break;
        double rho_used = rho;
        reg_decrease(o, rho, drho);
        // forward pass with backtracking line search
        double alpha = 1.0, J = Jprev;
        int jacc = -1;
        for (int j = 0; j < o.max_linesearch; ++j) {
          st->n_forward++;
          st->ls_trials++;
          bool ok = rollout(t, o, w.X.data(), w.U.data(), w.K.data(), w.d.data(), alpha, true, w.Xc.data(), w.Uc.data());
          if (ok) {
            double Jc = total_cost(t, o, w.Xc.data(), w.Uc.data(), al, true);
            double expected = -alpha * (dV[0] + alpha * dV[1]);
            double z = (expected > 0.0) ? (Jprev - Jc) / expected : -1.0;
            if ((z > o.ls_lower && z <= o.ls_upper) || Jc < Jprev) { jacc = j; J = Jc; break; }
          }
          alpha *= 0.5;
        }
        if (jacc >= 0) {
          w.X.swap(w.Xc);
          w.U.swap(w.Uc);
        } else {
          st->fp_fails++;
          J = Jprev;
          reg_increase(o, rho, drho);
          rho += o.reg_fp;
        }
        double dJ = std::fabs(J - Jprev);
        if (trace && trow < trace_rows) {
          double* r = trace + 8 * (trow++);
          r[0] = outer; r[1] = it; r[2] = Jprev; r[3] = J; r[4] = jacc; r[5] = rho_used; r[6] = dV[0]; r[7] = dV[1];
        }
        Jprev = J;
        djz = (dJ == 0.0) ? djz + 1 : 0;
        double g = 0.0;
        for (int k = 0; k < N - 1; ++k) {
          double m = 0.0;
          for (int a = 0; a < 3; ++a) m = std::max(m, std::fabs(w.d[3 * k + a]) / (std::fabs(w.U[3 * k + a]) + 1.0));
          g += m;
        }
        st->grad = g / (double)(N - 1);
        st->inner_iters++;
        if (0.0 < dJ && dJ < o.cost_tol) break;
        if (st->grad < o.grad_tol) break;
        if (djz > o.dj_counter_limit) break;
      }
      st->outer_iters = outer;
      st->c_max = max_violation(t, o, w.X.data(), w.U.data());
      if (regfail) { st->status = TSAT_REG_FAIL; break; }
      if (st->c_max < o.constraint_tol) { st->status = TSAT_CONVERGED; break; }
      if (outer == o.max_outer) break;
      // dual + penalty update (Appendix A "solve_AL")
      for (int k = 0; k < N - 1; ++k) {
        const double* u = w.U.data() + 3 * k;
        double* lam = al.lam.data() + 6 * k;
        for (int a = 0; a < 3; ++a) {
          double c = u[a] - t.uhi[a];
          lam[a] = std::min(std::max(lam[a] + al.mu * c, 0.0), o.dual_max);
          c = t.ulo[a] - u[a];
          lam[3 + a] = std::min(std::max(lam[3 + a] + al.mu * c, 0.0), o.dual_max);
        }
      }
      for (int i = 0; i < 7; ++i)
        if (o.terminal_mask >> i & 1) {
          double e = w.X[7 * (N - 1) + i] - t.xf[i];
          al.nu[i] = std::min(std::max(al.nu[i] + al.mu * e, -o.dual_max), o.dual_max);
        }
      al.mu = std::min(al.mu * o.penalty_scale, o.penalty_max);
    }
  }
  st->c_max = max_violation(t, o, w.X.data(), w.U.data());
  st->cost = total_cost(t, o, w.X.data(), w.U.data(), al, false);
  st->cost_al = total_cost(t, o, w.X.data(), w.U.data(), al, true);
  std::memcpy(Xout, w.X.data(), sizeof(double) * 7 * N);
  std::memcpy(Uout, w.U.data(), sizeof(double) * 3 * (N - 1));
  if (Kout)  // ABI layout: K(a, j, k) column-major 3 x 7 x (N-1)
    for (int k = 0; k < N - 1; ++k)
      for (int a = 0; a < 3; ++a)
        for (int j = 0; j < 7; ++j) Kout[(size_t)k * 21 + j * 3 + a] = w.K[(size_t)(k * 3 + a) * 7 + j];
}

}  // namespace

// =================================================================================================
// C entry points (ctypes)
// =================================================================================================
extern "C" {

void orc_default_options(tsat_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->n_knots = 0; o->n_tab = 0; o->integrator = 3; o->precision = 64;
  o->max_outer = 20; o->max_inner = 50; o->max_linesearch = 20; o->dj_counter_limit = 10;
  o->cost_tol = 1e-4; o->grad_tol = 1e-5; o->constraint_tol = 1e-3;
  o->penalty_init = 1.0; o->penalty_scale = 10.0; o->penalty_max = 1e8; o->dual_max = 1e8;
  o->reg_init = 0.0; o->reg_scale = 1.6; o->reg_min = 1e-8; o->reg_max = 1e8; o->reg_fp = 10.0;
  o->ls_lower = 1e-8; o->ls_upper = 10.0; o->max_state = 1e8; o->u_scale = 1e-2;
  o->terminal_mask = 0x7f; o->error_state = 0;
}

void orc_qmult(const double* q1, const double* q2, double* o) { qmult<double>(q1, q2, o); }
void orc_qrot(const double* q, const double* r, double* o) { qrot<double>(q, r, o); }
// src/attitude_controller.jl:164-166
void orc_qinv(const double* q, double* o) { o[0] = q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = -q[3]; }
// src/attitude_controller.jl:172-176, column-major 3x3
void orc_hat(const double* x, double* H) {
  H[0] = 0;     H[3] = -x[2]; H[6] = x[1];
  H[1] = x[2];  H[4] = 0;     H[7] = -x[0];
  H[2] = -x[1]; H[5] = x[0];  H[8] = 0;
}
// G(q) = [-v'; s I + hat(v)]  (4x3, column-major)   src/attitude_controller.jl:69
void orc_gmat(const double* q, double* G) {
  double H[9];
  orc_hat(q + 1, H);
  for (int c = 0; c < 3; ++c) {
    G[0 + 4 * c] = -q[1 + c];
    for (int r = 0; r < 3; ++r) G[(1 + r) + 4 * c] = (r == c ? q[0] : 0.0) + H[r + 3 * c];
  }
}
void orc_inv3(const double* M, double* Mi) { inv3(M, Mi); }
// E(q) = blkdiag(I3, G(q)) used by the error-state solve (7x6 row-major)
void orc_emat(const double* q, double* E) { emat(q, E); }

// A2 literal: DerivFunction(dx,x,u) with the globals passed explicitly.
// Btab is [rows][3]; Nglob = the script-global N (src/DerivFunction.jl:28), tspan = tf - t0 (:44).
void orc_deriv8(const double* x8, const double* u, const double* Btab, int rows, int Nglob, const double* Jcm,
                double tspan, double u_scale, double* dx8) {
  Phys ph;
  std::memcpy(ph.J, Jcm, sizeof(ph.J));
  inv3(ph.J, ph.Jinv);
  ph.u_scale = u_scale;
  long idx = (long)std::floor(x8[7] * (double)Nglob + 1.0) - 1;  // Julia is 1-based
  if (idx < 0) idx = 0;
  if (idx > rows - 1) idx = rows - 1;
  dyn7<double>(x8, u, Btab + 3 * idx, ph, dx8);
  dx8[7] = 1.0 / tspan;
}
// A3 literal: attitude_dynamics(x,u,B_B,J) — body-frame field passed in, no unit scaling.
void orc_attitude_dynamics(const double* x7, const double* u, const double* BB, const double* Jcm, double* xd) {
  double Jinv[9];
  inv3(Jcm, Jinv);
  double w[3] = {x7[0], x7[1], x7[2]};
  double nq = std::sqrt(x7[3] * x7[3] + x7[4] * x7[4] + x7[5] * x7[5] + x7[6] * x7[6]);
  double q[4] = {x7[3] / nq, x7[4] / nq, x7[5] / nq, x7[6] / nq};
  double p[4] = {0, w[0], w[1], w[2]}, qd[4], tau[3], Jw[3], wJw[3];
  qmult<double>(q, p, qd);
  cross3<double>(u, BB, tau);
  for (int r = 0; r < 3; ++r) Jw[r] = Jcm[r] * w[0] + Jcm[r + 3] * w[1] + Jcm[r + 6] * w[2];
  cross3<double>(w, Jw, wJw);
  for (int r = 0; r < 3; ++r)
    xd[r] = Jinv[r] * (tau[0] - wJw[0]) + Jinv[r + 3] * (tau[1] - wJw[1]) + Jinv[r + 6] * (tau[2] - wJw[2]);
  for (int i = 0; i < 4; ++i) xd[3 + i] = 0.5 * qd[i];
}
void orc_dyn7(const double* x, const double* u, const double* b, const double* Jcm, double u_scale, double* xd) {
  Phys ph;
  std::memcpy(ph.J, Jcm, sizeof(ph.J));
  inv3(ph.J, ph.Jinv);
  ph.u_scale = u_scale;
  dyn7<double>(x, u, b, ph, xd);
}
void orc_rk_step(int integ, const double* x, const double* u, const double* b0, const double* b1, const double* b2,
                 double h, const double* Jcm, double u_scale, double* xn) {
  Phys ph;
  std::memcpy(ph.J, Jcm, sizeof(ph.J));
  inv3(ph.J, ph.Jinv);
  ph.u_scale = u_scale;
  rk_step<double>(integ, x, u, b0, b1, b2, h, ph, xn);
}
// tableau check on the scalar test equation x' = lam x (order-of-accuracy test)
double orc_rk_scalar(int integ, double lam, double x, double h) {
  double k1 = lam * x * h, k2 = lam * (x + k1 / 2) * h;
  if (integ == 3) { double k3 = lam * (x - k1 + 2 * k2) * h; return x + (k1 + 4 * k2 + k3) / 6; }
  double k3 = lam * (x + k2 / 2) * h, k4 = lam * (x + k3) * h;
  return x + (k1 + 2 * k2 + 2 * k3 + k4) / 6;
}
// A5: row-major A(7x7), B(7x3)
void orc_discrete_jacobian(int integ, const double* x, const double* u, const double* b0, const double* b1,
                           const double* b2, double h, const double* Jcm, double u_scale, double* A, double* B) {
  Phys ph;
  std::memcpy(ph.J, Jcm, sizeof(ph.J));
  inv3(ph.J, ph.Jinv);
  ph.u_scale = u_scale;
  discrete_jacobian(integ, x, u, b0, b1, b2, h, ph, A, B);
}

// A6: src/quaternion_toolbox.jl:58-75 — 7-vector [dw; MRP(q2^-1 (x) q1); 0]
void orc_quaternion_error(const double* X1, const double* X2, double* dx7) {
  double qi[4], qe[4];
  orc_qinv(X2 + 3, qi);
  qmult<double>(qi, X1 + 3, qe);
  for (int i = 0; i < 3; ++i) dx7[i] = X1[i] - X2[i];
  for (int i = 0; i < 3; ++i) dx7[3 + i] = qe[1 + i] / (1.0 + qe[0]);
  dx7[6] = 0.0;
}
// A6: src/attitude_controller.jl:59-81 — Ahat = E(q_{k+1})' A E(q_k) (6x6), Bhat = E(q_{k+1})' B (6x3);
// A (7x7) and B (7x3) row-major in, row-major out.
void orc_reduce_error_state(const double* A, const double* B, const double* qk, const double* qn, double* Ah, double* Bh) {
  double Gk[12], Gn[12];
  orc_gmat(qk, Gk);
  orc_gmat(qn, Gn);
  double Ek[7 * 6] = {0}, En[7 * 6] = {0};  // row-major 7x6
  for (int i = 0; i < 3; ++i) { Ek[i * 6 + i] = 1.0; En[i * 6 + i] = 1.0; }
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 3; ++c) { Ek[(3 + r) * 6 + 3 + c] = Gk[r + 4 * c]; En[(3 + r) * 6 + 3 + c] = Gn[r + 4 * c]; }
  double T[7 * 6];
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 7; ++m) a += A[i * 7 + m] * Ek[m * 6 + j]; T[i * 6 + j] = a; }
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 7; ++m) a += En[m * 6 + i] * T[m * 6 + j]; Ah[i * 6 + j] = a; }
    for (int c = 0; c < 3; ++c) { double a = 0; for (int m = 0; m < 7; ++m) a += En[m * 6 + i] * B[m * 3 + c]; Bh[i * 3 + c] = a; }
  }
}
// A6: src/quaternion_toolbox.jl:15-36 on the 7-state (time row/column dropped): returns
// Qxx = E'QE (6x6 row-major), Qx = E'(Q x + q) (6), for a full symmetric Q (7x7 row-major) and linear term q.
void orc_quaternion_expansion(const double* Q, const double* qlin, const double* x, double* Qxx, double* Qx) {
  double G[12];
  orc_gmat(x + 3, G);
  double E[7 * 6] = {0};
  for (int i = 0; i < 3; ++i) E[i * 6 + i] = 1.0;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 3; ++c) E[(3 + r) * 6 + 3 + c] = G[r + 4 * c];
  double g[7], T[7 * 6];
  for (int i = 0; i < 7; ++i) { double a = qlin[i]; for (int m = 0; m < 7; ++m) a += Q[i * 7 + m] * x[m]; g[i] = a; }
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 7; ++m) a += Q[i * 7 + m] * E[m * 6 + j]; T[i * 6 + j] = a; }
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 7; ++m) a += E[m * 6 + i] * T[m * 6 + j]; Qxx[i * 6 + j] = a; }
    double a = 0; for (int m = 0; m < 7; ++m) a += E[m * 6 + i] * g[m]; Qx[i] = a;
  }
}

// A7: src/attitude_controller.jl:83-92 — TVLQR Riccati on reduced blocks.
// Ah: [N-1][6x6 row-major], Bh: [N-1][6x3 row-major], Q,Qf 6x6 row-major, R 3x3 row-major. K out: [N-1][3x6 row-major].
void orc_tvlqr_riccati(int N, const double* Ah, const double* Bh, const double* Q, const double* R, const double* Qf, double* K) {
  double S[36];
  std::memcpy(S, Qf, sizeof(S));
  for (int k = N - 2; k >= 0; --k) {
    const double* A = Ah + (size_t)k * 36;
    const double* B = Bh + (size_t)k * 18;
    double SA[36], SB[18], Mi[9], BSA[18];
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 6; ++m) a += S[i * 6 + m] * A[m * 6 + j]; SA[i * 6 + j] = a; }
      for (int c = 0; c < 3; ++c) { double a = 0; for (int m = 0; m < 6; ++m) a += S[i * 6 + m] * B[m * 3 + c]; SB[i * 3 + c] = a; }
    }
    double Mcm[9];
    for (int c = 0; c < 3; ++c) {
      for (int e = 0; e < 3; ++e) { double a = R[c * 3 + e]; for (int m = 0; m < 6; ++m) a += B[m * 3 + c] * SB[m * 3 + e]; Mcm[c + 3 * e] = a; }
      for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 6; ++m) a += B[m * 3 + c] * SA[m * 6 + j]; BSA[c * 6 + j] = a; }
    }
    double Micm[9];
    inv3(Mcm, Micm);
    for (int c = 0; c < 3; ++c) for (int e = 0; e < 3; ++e) Mi[c * 3 + e] = Micm[c + 3 * e];
    double* Kk = K + (size_t)k * 18;
    for (int c = 0; c < 3; ++c)
      for (int j = 0; j < 6; ++j) Kk[c * 6 + j] = Mi[c * 3 + 0] * BSA[0 * 6 + j] + Mi[c * 3 + 1] * BSA[1 * 6 + j] + Mi[c * 3 + 2] * BSA[2 * 6 + j];
    // S = Q + K'RK + (A-BK)' S (A-BK)
    double Acl[36], SAcl[36], RK[18], Sn[36];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) { double a = A[i * 6 + j]; for (int c = 0; c < 3; ++c) a -= B[i * 3 + c] * Kk[c * 6 + j]; Acl[i * 6 + j] = a; }
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) { double a = 0; for (int m = 0; m < 6; ++m) a += S[i * 6 + m] * Acl[m * 6 + j]; SAcl[i * 6 + j] = a; }
    for (int c = 0; c < 3; ++c)
      for (int j = 0; j < 6; ++j) RK[c * 6 + j] = R[c * 3 + 0] * Kk[0 * 6 + j] + R[c * 3 + 1] * Kk[1 * 6 + j] + R[c * 3 + 2] * Kk[2 * 6 + j];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        double a = Q[i * 6 + j];
        for (int c = 0; c < 3; ++c) a += Kk[c * 6 + i] * RK[c * 6 + j];
        for (int m = 0; m < 6; ++m) a += Acl[m * 6 + i] * SAcl[m * 6 + j];
        Sn[i * 6 + j] = a;
      }
    std::memcpy(S, Sn, sizeof(S));
  }
}

// pieces of the solver exposed for unit tests ------------------------------------------------------
static Traj make_traj(const tsat_options* o, int64_t t, const double* x0, const double* xf, const double* Btab,
                      const int32_t* btab_idx, const double* tau0, const double* dtau, const double* dt,
                      const double* Jmat, const double* Qd, const double* Qfd, const double* Rd, const double* ulo,
                      const double* uhi) {
  Traj tr;
  tr.N = o->n_knots; tr.n_tab = o->n_tab; tr.integ = o->integrator;
  tr.x0 = x0 + 7 * t; tr.xf = xf + 7 * t;
  int64_t bi = btab_idx ? btab_idx[t] : t;
  tr.Bt = Btab + (size_t)bi * 3 * o->n_tab;
  tr.tau0 = tau0[t]; tr.dtau = dtau[t]; tr.dt = dt[t];
  std::memcpy(tr.ph.J, Jmat + 9 * t, sizeof(tr.ph.J));
  inv3(tr.ph.J, tr.ph.Jinv);
  tr.ph.u_scale = o->u_scale;
  tr.Qd = Qd + 7 * t; tr.Qfd = Qfd + 7 * t; tr.Rd = Rd + 3 * t; tr.ulo = ulo + 3 * t; tr.uhi = uhi + 3 * t;
  return tr;
}

/* Same argument list as tsat_solve_batch (include/tortoise_hip.h) minus the handle, plus:
 *   nthreads    OpenMP threads over trajectories (1 = serial);
 *   trace       optional 8 x trace_rows x T per-iteration log (see tsat_batch_trace);
 *   n_knots     optional per-trajectory knot counts (see tsat_batch_knots).                       */
int orc_solve_batch(const tsat_options* o, int64_t T, int64_t n_btab, const double* x0, const double* xf,
                    const double* Btab, const int32_t* btab_idx, const double* tau0, const double* dtau,
                    const double* dt, const double* Jmat, const double* Qd, const double* Qfd, const double* Rd,
                    const double* ulo, const double* uhi, const double* U0, double* X, double* U, double* K,
                    tsat_stats* stats, int nthreads, double* trace, int trace_rows, const int32_t* n_knots) {
  if (!o || o->n_knots < 2 || o->n_tab < 1 || (o->integrator != 3 && o->integrator != 4)) return -1;
  if (o->max_linesearch < 1 || o->max_linesearch > TSAT_MAX_LINESEARCH) return -1;
  if (o->error_state != 0 && o->error_state != 1) return -2;
  if (!btab_idx && n_btab != T) return -1;
  const int N = o->n_knots;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t t = 0; t < T; ++t) {
    Traj tr = make_traj(o, t, x0, xf, Btab, btab_idx, tau0, dtau, dt, Jmat, Qd, Qfd, Rd, ulo, uhi);
    double* Xt = X + (size_t)7 * N * t;
    double* Ut = U + (size_t)3 * (N - 1) * t;
    double* Kt = K ? K + (size_t)21 * (N - 1) * t : nullptr;
    if (n_knots) {   // ragged batch: own horizon, slabs zero-filled beyond it (as tsat_batch_knots)
      tr.N = n_knots[t];
      std::memset(Xt, 0, sizeof(double) * 7 * N);
      std::memset(Ut, 0, sizeof(double) * 3 * (N - 1));
      if (Kt) std::memset(Kt, 0, sizeof(double) * 21 * (N - 1));
    }
    solve_one(tr, *o, U0 + (size_t)3 * (N - 1) * t, Xt, Ut, Kt, stats + t,
              trace ? trace + (size_t)8 * trace_rows * t : nullptr, trace_rows);
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Receding-horizon re-solve (BASELINE.json configs[4]; not in the reference — see tsat_mpc_run in the header): per
// trajectory, n_steps times { solve from the current state with the current warm start; record x_t, u_t = U[0];
// plant step with rk3/rk4 of the model dynamics; shift the plan by one knot (last control repeated); tau0 += dtau }.
// X_last / U_last (may be NULL) receive the plan of the last solve, laid out as orc_solve_batch's X / U.
// ------------------------------------------------------------------------------------------------------------
int orc_mpc_batch(const tsat_options* o, int64_t T, int64_t n_btab, const double* x0, const double* xf,
                  const double* Btab, const int32_t* btab_idx, const double* tau0, const double* dtau,
                  const double* dt, const double* Jmat, const double* Qd, const double* Qfd, const double* Rd,
                  const double* ulo, const double* uhi, const double* U0, int32_t n_steps, int32_t plant_integrator,
                  double* X_hist, double* U_hist, tsat_stats* stats_last, double* X_last, double* U_last, int nthreads,
                  const int32_t* n_knots) {
  if (!o || o->n_knots < 2 || o->n_tab < 1 || (o->integrator != 3 && o->integrator != 4)) return -1;
  if (n_steps < 1 || (plant_integrator != 3 && plant_integrator != 4)) return -1;
  if (!btab_idx && n_btab != T) return -1;
  const int NS = o->n_knots;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t t = 0; t < T; ++t) {
    Traj tr = make_traj(o, t, x0, xf, Btab, btab_idx, tau0, dtau, dt, Jmat, Qd, Qfd, Rd, ulo, uhi);
    if (n_knots) tr.N = n_knots[t];
    const int N = tr.N;
    double xc[7];
    std::vector<double> Uw(U0 + (size_t)3 * (NS - 1) * t, U0 + (size_t)3 * (NS - 1) * (t + 1)), X((size_t)7 * NS, 0.0),
        U((size_t)3 * (NS - 1), 0.0);
    for (int i = 0; i < 7; ++i) xc[i] = x0[7 * t + i];
    tsat_stats st;
    for (int s = 0; s < n_steps; ++s) {
      tr.x0 = xc;
      solve_one(tr, *o, Uw.data(), X.data(), U.data(), nullptr, &st, nullptr, 0);
      double* hx = X_hist + ((size_t)t * (n_steps + 1) + s) * 7;
      double* hu = U_hist + ((size_t)t * n_steps + s) * 3;
      for (int i = 0; i < 7; ++i) hx[i] = xc[i];
      for (int c = 0; c < 3; ++c) hu[c] = U[c];
      double xn[7];
      rk_step<double>(plant_integrator, xc, U.data(), brow(tr, 0, 0.0), brow(tr, 0, 0.5), brow(tr, 0, 1.0), tr.dt, tr.ph, xn);
      for (int k = 0; k < N - 1; ++k) {
        const int src = (k + 1 < N - 1) ? k + 1 : N - 2;
        for (int c = 0; c < 3; ++c) Uw[(size_t)3 * k + c] = U[(size_t)3 * src + c];
      }
      for (int i = 0; i < 7; ++i) xc[i] = xn[i];
      if (s == n_steps - 1)
        for (int i = 0; i < 7; ++i) hx[7 + i] = xn[i];
      tr.tau0 = tr.tau0 + tr.dtau;
    }
    if (stats_last) stats_last[t] = st;
    if (X_last) std::memcpy(X_last + (size_t)7 * NS * t, X.data(), sizeof(double) * 7 * NS);
    if (U_last) std::memcpy(U_last + (size_t)3 * (NS - 1) * t, U.data(), sizeof(double) * 3 * (NS - 1));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Closed-loop TVLQR tracking: literal restatement of attitude_simulation / attitude_lqr
// (src/attitude_controller.jl:1-119), the plants (src/simulator.jl, src/gain_simulator.jl) and the slew-time statistic
// (src/monte_carlo.jl:242-262). Randomness is an input: `noise` carries the values `simulator` would draw.
// ------------------------------------------------------------------------------------------------------------
void orc_tvlqr_default_options(tsat_tvlqr_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->linearize_dt_sq = 1; o->min_steps = 10; o->u_scale = 1e-2; o->w_tol = 0.05; o->angle_tol = 0.08727;
  const double deg = M_PI / 180.0;
  o->noise_mode = 0; o->noise_seed = 0;
  o->sigma_gyro = (0.38 * deg) * (0.38 * deg);   // src/simulator.jl:5
  o->sigma_att = deg * deg;                       // src/simulator.jl:10
  o->field_amp = 1e-5 * 1e-5;                     // src/simulator.jl:22
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) and the draws of
// one plant evaluation as include/tortoise_hip.h defines them for noise_mode = 1
void orc_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
  uint32_t k0 = key[0], k1 = key[1], c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  for (int r = 0; r < 10; ++r) {
    if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline double unit_open(uint32_t w) { return ((double)w + 0.5) / 4294967296.0; }
void orc_plant_noise(uint64_t seed, int64_t id, int32_t knot, int32_t stage, double sg, double sa, double fa, double* nz) {
  const uint32_t key[2] = {(uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32)};
  uint32_t w[3][4];
  for (uint32_t j = 0; j < 3; ++j) {
    const uint32_t ctr[4] = {(uint32_t)((uint64_t)id & 0xFFFFFFFFull), (uint32_t)((uint64_t)id >> 32), (uint32_t)knot, (uint32_t)(4 * stage) + j};
    orc_philox4x32_10(key, ctr, w[j]);
  }
  auto bm = [](uint32_t a, uint32_t b, double& z0, double& z1) {
    const double r = std::sqrt(-2.0 * std::log(unit_open(a))), th = 2.0 * M_PI * unit_open(b);
    z0 = r * std::cos(th); z1 = r * std::sin(th);
  };
  double z[6];
  bm(w[0][0], w[0][1], z[0], z[1]); bm(w[0][2], w[0][3], z[2], z[3]); bm(w[1][0], w[1][1], z[4], z[5]);
  for (int i = 0; i < 3; ++i) { nz[i] = sg * z[i]; nz[3 + i] = sa * z[3 + i]; }
  nz[6] = fa * unit_open(w[1][2]); nz[7] = fa * unit_open(w[1][3]); nz[8] = fa * unit_open(w[2][0]);
}

// src/simulator.jl:1-42 with the three random draws passed in (nz: gyro(3), attitude rotation vector(3), field(3));
// nz == nullptr is src/gain_simulator.jl:1-53
static void sim_dyn(const double x[7], const double u[3], const double b[3], const double* nz, const Phys& ph, double xd[7]) {
  double w[3] = {x[0], x[1], x[2]};
  double nq = std::sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  double q[4] = {x[3] / nq, x[4] / nq, x[5] / nq, x[6] / nq};
  double bb[3] = {b[0], b[1], b[2]};
  if (nz) {
    for (int i = 0; i < 3; ++i) w[i] += nz[i];
    const double th = std::sqrt(nz[3] * nz[3] + nz[4] * nz[4] + nz[5] * nz[5]);
    const double sh = std::sin(th / 2) / th;
    double dq[4] = {std::cos(th / 2), nz[3] * sh, nz[4] * sh, nz[5] * sh}, qn[4];
    qmult<double>(q, dq, qn);
    for (int i = 0; i < 4; ++i) q[i] = qn[i];
    for (int i = 0; i < 3; ++i) bb[i] += nz[6 + i];
  }
  double p[4] = {0, w[0], w[1], w[2]}, qd[4], BB[3], us[3] = {u[0] * ph.u_scale, u[1] * ph.u_scale, u[2] * ph.u_scale}, tau[3], Jw[3], wJw[3];
  qmult<double>(q, p, qd);
  qrot<double>(q, bb, BB);
  cross3<double>(us, BB, tau);
  for (int r = 0; r < 3; ++r) Jw[r] = ph.J[r] * w[0] + ph.J[r + 3] * w[1] + ph.J[r + 6] * w[2];
  cross3<double>(w, Jw, wJw);
  for (int r = 0; r < 3; ++r)
    xd[r] = ph.Jinv[r] * (tau[0] - wJw[0]) + ph.Jinv[r + 3] * (tau[1] - wJw[1]) + ph.Jinv[r + 6] * (tau[2] - wJw[2]);
  for (int i = 0; i < 4; ++i) xd[3 + i] = 0.5 * qd[i];
}

void orc_reduce_error_state(const double*, const double*, const double*, const double*, double*, double*);
void orc_tvlqr_riccati(int, const double*, const double*, const double*, const double*, const double*, double*);

int orc_tvlqr_batch(const tsat_tvlqr_options* o, int64_t T, int64_t n_btab, const double* X, const double* U,
                    const double* xf, const double* Btab, const int32_t* btab_idx, const double* tau0,
                    const double* dtau, const double* dt, const double* Jmat, const double* Qd, const double* Qfd,
                    const double* Rd, const double* x0_sim, const double* noise, double* X_sim, double* U_sim,
                    double* K_lqr, tsat_tvlqr_stats* stats, int nthreads, const int32_t* n_knots, const int64_t* noise_id) {
  if (!o || o->n_knots < 2 || o->n_tab < 1) return -1;
  if (o->noise_mode == 1 && noise) return -1;
  if (!btab_idx && n_btab != T) return -1;
  const int NS = o->n_knots;   // slab stride; a ragged trajectory uses its first n_knots[t] samples (zero beyond)
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t t = 0; t < T; ++t) {
    Traj tr;
    const int N = n_knots ? n_knots[t] : NS;
    tr.N = N; tr.n_tab = o->n_tab; tr.integ = 4;
    tr.Bt = Btab + (size_t)(btab_idx ? btab_idx[t] : t) * 3 * o->n_tab;
    tr.tau0 = tau0[t]; tr.dtau = dtau[t]; tr.dt = dt[t];
    std::memcpy(tr.ph.J, Jmat + 9 * t, sizeof(tr.ph.J));
    inv3(tr.ph.J, tr.ph.Jinv);
    tr.ph.u_scale = o->u_scale;
    const double* Xt = X + (size_t)7 * NS * t;
    const double* Ut = U + (size_t)3 * (NS - 1) * t;
    const double h = tr.dt, hl = o->linearize_dt_sq ? h * h : h, frac = hl / h;
    // gains: Jacobians of the rk4-discretised gain_simulator at (X_k, U_k) (src/attitude_controller.jl:95-119) ...
    std::vector<double> Ah((size_t)36 * (N - 1)), Bh((size_t)18 * (N - 1)), Kt((size_t)18 * (N - 1));
    for (int k = 0; k < N - 1; ++k) {
      double A7[49], B7[21];
      discrete_jacobian(4, Xt + 7 * k, Ut + 3 * k, brow(tr, k, 0.0), brow(tr, k, 0.5 * frac), brow(tr, k, frac), hl, tr.ph, A7, B7);
      // ... reduced with G(q_k), G(q_{k+1}) (:59-81)
      orc_reduce_error_state(A7, B7, Xt + 7 * k + 3, Xt + 7 * (k + 1) + 3, Ah.data() + (size_t)36 * k, Bh.data() + (size_t)18 * k);
    }
    double Q[36] = {0}, Qf[36] = {0}, R[9] = {0};
    for (int i = 0; i < 6; ++i) { Q[i * 6 + i] = Qd[6 * t + i]; Qf[i * 6 + i] = Qfd[6 * t + i]; }
    for (int i = 0; i < 3; ++i) R[i * 3 + i] = Rd[3 * t + i];
    orc_tvlqr_riccati(N, Ah.data(), Bh.data(), Q, R, Qf, Kt.data());   // (:83-92)
    // closed loop (:39-45)
    double* Xs = X_sim + (size_t)7 * NS * t;
    double* Us = U_sim + (size_t)3 * (NS - 1) * t;
    std::memset(Xs, 0, sizeof(double) * 7 * NS);
    std::memset(Us, 0, sizeof(double) * 3 * (NS - 1));
    if (K_lqr) std::memset(K_lqr + (size_t)t * (NS - 1) * 18, 0, sizeof(double) * 18 * (NS - 1));
    for (int i = 0; i < 7; ++i) Xs[i] = x0_sim[7 * t + i];
    for (int k = 0; k < N - 1; ++k) {
      const double* xs = Xs + 7 * k;
      const double* xr = Xt + 7 * k;
      double dX[6], qi[4] = {xr[3], -xr[4], -xr[5], -xr[6]}, qe[4];
      for (int i = 0; i < 3; ++i) dX[i] = xs[i] - xr[i];
      qmult<double>(qi, xs + 3, qe);
      for (int i = 0; i < 3; ++i) dX[3 + i] = qe[1 + i];
      double us[3];
      for (int a = 0; a < 3; ++a) {
        double v = Ut[3 * k + a];
        for (int j = 0; j < 6; ++j) v -= Kt[(size_t)18 * k + a * 6 + j] * dX[j];
        us[a] = v;
        Us[3 * k + a] = v;
      }
      double drawn[36];
      if (o->noise_mode == 1)
        for (int st = 0; st < 4; ++st)
          orc_plant_noise(o->noise_seed, noise_id ? noise_id[t] : t, k, st, o->sigma_gyro, o->sigma_att, o->field_amp, drawn + 9 * st);
      const double* nz = o->noise_mode == 1 ? drawn : (noise ? noise + ((size_t)t * (NS - 1) + k) * 36 : nullptr);
      const double *b0 = brow(tr, k, 0.0), *b1 = brow(tr, k, 0.5), *b2 = brow(tr, k, 1.0);
      double k1[7], k2[7], k3[7], k4[7], tmp[7];
      sim_dyn(xs, us, b0, nz, tr.ph, k1);
      for (int i = 0; i < 7; ++i) { k1[i] *= h; tmp[i] = xs[i] + k1[i] / 2; }
      sim_dyn(tmp, us, b1, nz ? nz + 9 : nullptr, tr.ph, k2);
      for (int i = 0; i < 7; ++i) { k2[i] *= h; tmp[i] = xs[i] + k2[i] / 2; }
      sim_dyn(tmp, us, b1, nz ? nz + 18 : nullptr, tr.ph, k3);
      for (int i = 0; i < 7; ++i) { k3[i] *= h; tmp[i] = xs[i] + k3[i]; }
      sim_dyn(tmp, us, b2, nz ? nz + 27 : nullptr, tr.ph, k4);
      for (int i = 0; i < 7; ++i) { k4[i] *= h; Xs[7 * (k + 1) + i] = xs[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6; }
    }
    if (K_lqr)
      for (int k = 0; k < N - 1; ++k)
        for (int a = 0; a < 3; ++a)
          for (int j = 0; j < 6; ++j) K_lqr[((size_t)t * (NS - 1) + k) * 18 + j * 3 + a] = Kt[(size_t)18 * k + a * 6 + j];
    // slew-time statistic (src/monte_carlo.jl:242-262; the norm is taken at sample j — the reference indexes the
    // run number there, `norm(sim_states[i][1:3,i])` (:247), an evident slip that rate_as_written = 1 reproduces: the
    // rate of sample i = trial number, for every j; clamped to the trajectory where Julia would raise a BoundsError)
    tsat_tvlqr_stats& st = stats[t];
    st.slew_index = 0; st.failed = 1; st.slew_time = h * N;
    const double* qf = xf + 7 * t + 3;
    double qfi[4] = {qf[0], -qf[1], -qf[2], -qf[3]};
    double w_trial = 0;
    if (o->rate_as_written) {
      const int64_t id = noise_id ? noise_id[t] : t;
      const double* xi = Xs + 7 * (size_t)std::min<int64_t>(std::max<int64_t>(id, 0), N - 1);
      w_trial = std::sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
    }
    for (int j = 1; j <= N; ++j) {
      const double* xs = Xs + 7 * (j - 1);
      const double wj = std::sqrt(xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2]);
      double wn = o->rate_as_written ? w_trial : wj, qe[4];
      qmult<double>(qfi, xs + 3, qe);
      double ang = 2.0 * std::acos(std::min(qe[0], 1.0));
      if (j > o->min_steps && wn < o->w_tol && ang < o->angle_tol && st.slew_index == 0) { st.slew_index = j; st.failed = 0; st.slew_time = h * j; }
      if (j == N) { st.final_w_norm = wj; st.final_angle = ang; }
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Horizon selection: src/magnetic_toolbox.jl:1-31 restated. cond() of the symmetric PSD 3x3 Gramian by
// cyclic Jacobi rotations (2-norm condition number = lambda_max / lambda_min).
// ------------------------------------------------------------------------------------------------------------
static double cond_sym3(const double G[6]) {  // G = [g00 g01 g02 g11 g12 g22]
  double a[3][3] = {{G[0], G[1], G[2]}, {G[1], G[3], G[4]}, {G[2], G[4], G[5]}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        double th = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 3; ++k) { double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - sn * akq; a[k][q] = sn * akp + c * akq; }
        for (int k = 0; k < 3; ++k) { double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - sn * aqk; a[q][k] = sn * apk + c * aqk; }
      }
  }
  double l0 = std::fabs(a[0][0]), l1 = std::fabs(a[1][1]), l2 = std::fabs(a[2][2]);
  double mx = std::max(l0, std::max(l1, l2)), mn = std::min(l0, std::min(l1, l2));
  return mn > 0.0 ? mx / mn : std::numeric_limits<double>::infinity();
}

int orc_horizon_batch(int64_t T, int32_t n_rows, const double* Btab, const double* dt_row, const double* cutoff,
                      int32_t* tf_index, double* cond_at, double* cond_all /* T x n_rows or NULL */) {
  if (T < 1 || n_rows < 1) return -1;
  for (int64_t t = 0; t < T; ++t) {
    const double* B = Btab + (size_t)t * n_rows * 3;
    double G[6] = {0, 0, 0, 0, 0, 0};
    tf_index[t] = 0;
    if (cond_at) cond_at[t] = std::numeric_limits<double>::infinity();
    for (int i = 0; i < n_rows; ++i) {
      const double b0 = B[3 * i], b1 = B[3 * i + 1], b2 = B[3 * i + 2];
      const double w = (i == 0) ? 1.0 : dt_row[t];          // the first slice is not scaled by dt (:6)
      // hat(b) hat(b)' = |b|^2 I - b b'
      const double n2 = b0 * b0 + b1 * b1 + b2 * b2;
      G[0] += (n2 - b0 * b0) * w; G[1] += (-b0 * b1) * w; G[2] += (-b0 * b2) * w;
      G[3] += (n2 - b1 * b1) * w; G[4] += (-b1 * b2) * w; G[5] += (n2 - b2 * b2) * w;
      const double c = cond_sym3(G);
      if (cond_all) cond_all[(size_t)t * n_rows + i] = c;
      if (tf_index[t] == 0 && c < cutoff[t]) { tf_index[t] = i + 1; if (cond_at) cond_at[t] = c; if (!cond_all) break; }
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Field-table generation: src/magnetic_toolbox.jl:33-106 with src/kep_ECI.jl, src/OrbitPlotter.jl,
// src/igrf.jl:70-274, src/legendre.jl:254-292, src/dlegendre.jl:221-309 restated.
// ------------------------------------------------------------------------------------------------------------
void orc_btable_default_options(tsat_btable_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->n_half = 5000; o->mjd = 58155.0; o->gm = 3.986004418e14 * 1e-9; o->r_igrf_km = 400.0 + 6371.0; o->date = 2019.0;
}

// igrf12(date, r [m], lat, lon) geocentric, 2015 <= date < 2020: returns [x; y; z] in nT (src/igrf.jl:70-274)
void orc_igrf12(double date, double r, double lam, double Om, double* out) {
  const int NM = IGRF12_NMAX;
  const double theta = M_PI / 2 - lam;
  const double phi = (Om >= 0) ? Om : 2 * M_PI + Om;
  r /= 1000;
  const double dt = date - 2015.0;
  double P[NM + 1][NM + 3], dP[NM + 1][NM + 3];
  for (int i = 0; i <= NM; ++i) for (int j = 0; j < NM + 3; ++j) { P[i][j] = 0; dP[i][j] = 0; }
  {  // legendre_schmidt_quasi_normalized!, ph_term = false (src/legendre.jl:254-292)
    const double c = std::cos(theta), s = std::sqrt(1 - c * c);
    P[0][0] = 1; P[1][0] = c; P[1][1] = s;
    for (int n = 2; n <= NM; ++n) {
      for (int m = 0; m < n; ++m) {
        const double aux = (double)((n - m) * (n + m));
        const double a_nm = std::sqrt(((2.0 * n - 1) * (2.0 * n - 1)) / aux);
        const double b_nm = std::sqrt(((double)(n + m - 1) * (n - m - 1)) / aux);
        P[n][m] = a_nm * c * P[n - 1][m] - b_nm * P[n - 2][m];
      }
      P[n][n] = s * std::sqrt((2.0 * n - 1) / (2.0 * n)) * P[n - 1][n - 1];
    }
  }
  {  // dlegendre_fully_normalized! (src/dlegendre.jl:221-309), ph_term = false
    const double fact = (std::fmod(theta, 2 * M_PI) > M_PI) ? -1.0 : 1.0;
    for (int n = 1; n <= NM; ++n)
      for (int m = 0; m <= n; ++m) {
        double v;
        if (m == 0) {
          const double aux = std::sqrt(n * (n + 1) / 2.0);
          v = -(0.5 * aux) * P[n][1] + (-0.5 * aux) * P[n][1];
        } else if (m == 1) {
          v = 0.5 * std::sqrt(2.0 * n * (n + 1)) * P[n][0] - 0.5 * std::sqrt((double)(n + 2) * (n - 1)) * P[n][2];
        } else if (n != m) {
          v = 0.5 * std::sqrt((double)(n + m) * (n - m + 1)) * P[n][m - 1] - 0.5 * std::sqrt((double)(n + m + 1) * (n - m)) * P[n][m + 1];
        } else {
          v = 0.5 * std::sqrt((double)(n + m) * (n - m + 1)) * P[n][m - 1];
        }
        dP[n][m] = v * fact;
      }
  }
  const double a = 6371.2;
  const double sin_p = std::sin(phi), cos_p = std::cos(phi);
  const double ratio = a / r;
  double fact = ratio, dVr = 0, dVt = 0, dVp = 0;
  int kg = 0, kh = 0;
  for (int n = 1; n <= NM; ++n) {
    double ar = 0, at = 0, ap = 0;
    double Gnm = IGRF12_G2015[kg] + IGRF12_GSV[kg] * dt;
    ++kg;
    ar += -(n + 1) / r * Gnm * P[n][0];
    at += Gnm * dP[n][0];
    double sin_m1 = 0.0, sin_m2 = -sin_p, cos_m1 = 1.0, cos_m2 = cos_p;
    for (int m = 1; m <= n; ++m) {
      const double sin_m = 2 * cos_p * sin_m1 - sin_m2;
      const double cos_m = 2 * cos_p * cos_m1 - cos_m2;
      Gnm = IGRF12_G2015[kg] + IGRF12_GSV[kg] * dt;
      const double Hnm = IGRF12_H2015[kh] + IGRF12_HSV[kh] * dt;
      ++kg; ++kh;
      const double GcHs = Gnm * cos_m + Hnm * sin_m, GsHc = Gnm * sin_m - Hnm * cos_m;
      ar += -(n + 1) / r * GcHs * P[n][m];
      at += GcHs * dP[n][m];
      ap += (theta == 0) ? -m * GsHc * dP[n][m] : -m * GsHc * P[n][m];
      sin_m2 = sin_m1; sin_m1 = sin_m; cos_m2 = cos_m1; cos_m1 = cos_m;
    }
    fact *= ratio;
    dVr += ar * fact; dVp += ap * fact; dVt += at * fact;
  }
  dVr *= a; dVp *= a; dVt *= a;
  out[0] = 1 / r * dVt;
  out[1] = (theta == 0) ? -1 / r * dVp : -1 / (r * std::sin(theta)) * dVp;
  out[2] = dVr;
}

static void rz_deg(double ang, double M[9]) {   // R_z of src/kep_ECI.jl:37-42 (row-major)
  const double c = std::cos(ang * M_PI / 180), s = std::sin(ang * M_PI / 180);
  double R[9] = {c, s, 0, -s, c, 0, 0, 0, 1};
  std::memcpy(M, R, sizeof(R));
}
static void rx_deg(double ang, double M[9]) {   // R_x of src/kep_ECI.jl:44-49
  const double c = std::cos(ang * M_PI / 180), s = std::sin(ang * M_PI / 180);
  double R[9] = {1, 0, 0, 0, c, s, 0, -s, c};
  std::memcpy(M, R, sizeof(R));
}
static void mm3(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double a = 0; for (int k = 0; k < 3; ++k) a += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = a; }
}
// kep_ECI (src/kep_ECI.jl:1-35)
void orc_kep_eci(const double* kep, double t0, double GM, double* r, double* v) {
  double A[6];
  for (int i = 0; i < 6; ++i) A[i] = kep[i];
  A[5] = std::fmod(kep[5] + t0 * std::sqrt(GM / (kep[1] * kep[1] * kep[1])), 360.0);
  double E = A[5] / 180 * M_PI;
  for (int i = 0; i < 100; ++i) E = E - (E - A[0] * std::sin(E) - A[5] / 180 * M_PI) / (1 - A[0] * std::cos(E));
  const double nu = 2 * std::atan2(std::sqrt(1 + A[0]) * std::sin(E / 2), std::sqrt(1 - A[0]) * std::cos(E / 2)) * 180 / M_PI;
  const double r_c = A[1] * (1 - A[0] * std::cos(E));
  const double o[3] = {r_c * std::cos(nu * M_PI / 180), r_c * std::sin(nu * M_PI / 180), 0};
  const double k = std::sqrt(GM * A[1]) / r_c;
  const double od[3] = {k * -std::sin(E), k * std::sqrt(1 - A[0] * A[0]) * std::cos(E), 0};
  double Ra[9], Rb[9], Rc[9], T1[9], M[9];
  rz_deg(-A[3], Ra); rx_deg(-A[2], Rb); rz_deg(-A[4], Rc);
  mm3(Ra, Rb, T1); mm3(T1, Rc, M);
  for (int i = 0; i < 3; ++i) {
    r[i] = M[3 * i] * o[0] + M[3 * i + 1] * o[1] + M[3 * i + 2] * o[2];
    v[i] = M[3 * i] * od[0] + M[3 * i + 1] * od[1] + M[3 * i + 2] * od[2];
  }
}
// OrbitPlotter (src/OrbitPlotter.jl:1-48): two-body + the J2 term as written
static void orbit_rhs(const double* x, double* xd) {
  const double GM = 3.986004418E14 * (1.0 / 1000) * (1.0 / 1000) * (1.0 / 1000);
  const double nr = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  const double J2 = 0.0010826359, nr7 = std::pow(nr, 7);
  const double rho2 = x[0] * x[0] + x[1] * x[1];
  const double fj[3] = {J2 * x[0] / nr7 * (6 * x[2] - 1.5 * rho2), J2 * x[1] / nr7 * (6 * x[2] - 1.5 * rho2), J2 * x[2] / nr7 * (3 * x[2] - 4.5 * rho2)};
  for (int i = 0; i < 3; ++i) { xd[i] = x[3 + i]; xd[3 + i] = GM / (nr * nr) * -x[i] / nr + fj[i]; }
}

int orc_btable_batch(const tsat_btable_options* o, int64_t T, const double* kep, const double* t0, const double* tf,
                     double* Btab, double* pos) {
  if (!o || o->n_half < 1 || !(o->date >= 2015.0 && o->date < 2020.0)) return -1;
  const int N = o->n_half;
  for (int64_t t = 0; t < T; ++t) {
    double u[6], ud[6];
    orc_kep_eci(kep + 6 * t, t0[t], o->gm, u, u + 3);
    const double dt = (tf[t] - t0[t]) / N;
    std::vector<double> P((size_t)3 * (2 * N + 1));
    for (int i = 0; i <= 2 * N; ++i) {          // Euler(), adaptive = false, tspan = (t0, 2 tf) (src/magnetic_toolbox.jl:51-54)
      for (int c = 0; c < 3; ++c) P[3 * i + c] = u[c];
      orbit_rhs(u, ud);
      for (int c = 0; c < 6; ++c) u[c] += dt * ud[c];
    }
    if (pos) std::memcpy(pos + (size_t)t * 3 * (2 * N + 1), P.data(), sizeof(double) * P.size());
    double* B = Btab + (size_t)t * 3 * 2 * N;
    for (int i = 0; i < 2 * N; ++i) {
      double* b = B + 3 * i;
      b[0] = b[1] = b[2] = 0.0;
      if (i == 2 * N - 1) break;                // the last row is left zero (:76)
      const double ti = t0[t] + dt * i;
      const double gmst = (280.4606 + 360.9856473 * (ti / 24 / 60 / 60 + o->mjd) - 51544.5) / 180 * M_PI;   // as written (:60)
      const double cg = std::cos(gmst), sg = std::sin(gmst);
      const double* p = P.data() + 3 * i;
      const double pe[3] = {cg * p[0] + sg * p[1], -sg * p[0] + cg * p[1], p[2]};
      const double lat = std::asin(pe[2] / std::sqrt(pe[0] * pe[0] + pe[1] * pe[1] + pe[2] * pe[2]));
      const double lon = std::atan2(pe[1], pe[0]);
      double bn[3];
      orc_igrf12(o->date, o->r_igrf_km * 1000, lat, lon, bn);
      for (int c = 0; c < 3; ++c) bn[c] /= 1.0e9;
      const double e[3] = {bn[1], bn[0], -bn[2]};                       // NED_to_ENU (:74)
      const double sl = std::sin(lon), cl = std::cos(lon), sa = std::sin(lat), ca = std::cos(lat);
      const double xyz[3] = {-sl * e[0] - sa * cl * e[1] + ca * cl * e[2], cl * e[0] - sa * sl * e[1] + ca * sl * e[2], ca * e[1] + sa * e[2]};   // (:92-95)
      b[0] = cg * xyz[0] - sg * xyz[1];                                 // Rz(GMST)' (:96)
      b[1] = sg * xyz[0] + cg * xyz[1];
      b[2] = xyz[2];
    }
  }
  return 0;
}

int orc_num_procs(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

}  // extern "C"
