// tsat_host_pack.hpp — host-side marshalling between the C ABI arrays (include/tortoise_hip.h) and the
// device-resident layouts of tsat_device.hpp. Pure C++ (no HIP) so the CPU lane-emulator in tests/emu
// exercises exactly the same packing code as libtortoise_hip.so.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "tsat_device.hpp"
#include "../../include/igrf12_2015_coeffs.h"   // IGRF-12 model constants (data)

namespace tsat {

// closed-form inverse of the 3x3 inertia (column-major in, row-major out); replaces inv(p.J) evaluated on every
// dynamics call by the reference (src/DerivFunction.jl:41)
inline void inertia_inverse_rm(const double* Jcm, double* Jrm, double* Jirm) {
  auto m = [&](int r, int c) { return Jcm[r + 3 * c]; };
  const double c00 = m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1);
  const double c01 = m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2);
  const double c02 = m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0);
  const double det = m(0, 0) * c00 + m(0, 1) * c01 + m(0, 2) * c02;
  const double id = 1.0 / det;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Jrm[3 * r + c] = m(r, c);
  Jirm[0] = c00 * id;
  Jirm[3] = c01 * id;
  Jirm[6] = c02 * id;
  Jirm[1] = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * id;
  Jirm[4] = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * id;
  Jirm[7] = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * id;
  Jirm[2] = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * id;
  Jirm[5] = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * id;
  Jirm[8] = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * id;
}

// inertia class of a batch: 2 = every tensor isotropic (J = j I: the 1U / 1P presets of src/input_parameters.jl:29-43),
// 1 = every tensor diagonal (3U preset), 0 = general. Selects the dynamics specialisation of the kernel.
inline int inertia_class(int64_t T, const double* Jmat) {
  int cls = 2;
  for (int64_t t = 0; t < T; ++t) {
    const double* J = Jmat + 9 * t;
    for (int i = 0; i < 9; ++i)
      if (i % 4 != 0 && J[i] != 0.0) return 0;
    if (!(J[0] == J[4] && J[4] == J[8])) cls = 1;
  }
  return cls;
}

// validate the per-trajectory inputs of an upload; returns "" or an error text. A non-finite weight or bound (e.g. the
// reference's R = 1/m_max^2 with m_max = 0 for a two-knot guess) or a singular inertia would otherwise only show up as a
// DIVERGED status or a nonsense cost.
inline std::string check_inputs(int64_t T, const double* x0, const double* xf, const double* tau0, const double* dtau,
                                const double* dt, const double* Jmat, const double* Qd, const double* Qfd,
                                const double* Rd, const double* ulo, const double* uhi) {
  auto finite = [](const double* a, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
      if (!std::isfinite(a[i])) return false;
    return true;
  };
  if (!finite(x0, 7 * T) || !finite(xf, 7 * T)) return "x0 / xf must be finite";
  if (!finite(tau0, T) || !finite(dtau, T)) return "tau0 / dtau must be finite";
  if (!finite(Qd, 7 * T) || !finite(Qfd, 7 * T) || !finite(Rd, 3 * T)) return "Qd / Qfd / Rd must be finite (Bryson weights of a degenerate guess?)";
  if (!finite(ulo, 3 * T) || !finite(uhi, 3 * T)) return "ulo / uhi must be finite";
  for (int64_t t = 0; t < T; ++t) {
    if (!(dt[t] > 0.0) || !std::isfinite(dt[t])) return "dt must be positive";
    for (int a = 0; a < 3; ++a)
      if (!(ulo[3 * t + a] <= uhi[3 * t + a])) return "ulo must not exceed uhi";
    const double* J = Jmat + 9 * t;
    if (!finite(J, 9)) return "Jmat must be finite";
    const double det = J[0] * (J[4] * J[8] - J[7] * J[5]) - J[3] * (J[1] * J[8] - J[7] * J[2]) + J[6] * (J[1] * J[5] - J[4] * J[2]);
    double nrm = 0;
    for (int i = 0; i < 9; ++i) nrm = std::fmax(nrm, std::fabs(J[i]));
    if (!(std::fabs(det) > 1e-12 * nrm * nrm * nrm)) return "Jmat is singular";
  }
  return "";
}

// validate an options block against a reserved batch; returns "" or an error text
inline std::string check_options(const tsat_options& o, int N, int n_tab, int max_ls_reserved) {
  if (o.n_knots != N) return "options.n_knots does not match the reserved batch";
  if (o.n_tab != n_tab) return "options.n_tab does not match the reserved batch";
  if (N < 2) return "n_knots must be >= 2";
  if (n_tab < 1) return "n_tab must be >= 1";
  if (o.integrator != 3 && o.integrator != 4) return "integrator must be 3 (rk3) or 4 (rk4)";
  if (o.precision != 64 && o.precision != 32) return "precision must be 64 or 32";
  if (o.error_state != 0 && o.error_state != 1) return "error_state must be 0 or 1";
  if (o.max_linesearch < 1 || o.max_linesearch > TSAT_MAX_LINESEARCH) return "max_linesearch must be in [1,32]";
  if (o.max_linesearch > max_ls_reserved) return "max_linesearch exceeds the reserved candidate slots";
  if (o.max_outer < 1 || o.max_inner < 1) return "iteration budgets must be >= 1";
  if (!(o.reg_scale > 1.0)) return "reg_scale must be > 1";
  if (!(o.penalty_scale >= 1.0) || !(o.penalty_init > 0.0)) return "penalty_init > 0 and penalty_scale >= 1 required";
  return "";
}

// pack per-trajectory parameters into [T][PSTRIDE] records
template <typename real>
void pack_params(int64_t T, const double* x0, const double* xf, const double* tau0, const double* dtau,
                 const double* dt, const double* Jmat, const double* Qd, const double* Qfd, const double* Rd,
                 const double* ulo, const double* uhi, real* P) {
  for (int64_t t = 0; t < T; ++t) {
    real* p = P + (size_t)t * PSTRIDE;
    for (int i = 0; i < PSTRIDE; ++i) p[i] = 0;
    for (int i = 0; i < 7; ++i) {
      p[P_X0 + i] = (real)x0[7 * t + i];
      p[P_XF + i] = (real)xf[7 * t + i];
      p[P_QD + i] = (real)Qd[7 * t + i];
      p[P_QFD + i] = (real)Qfd[7 * t + i];
    }
    for (int i = 0; i < 3; ++i) {
      p[P_RD + i] = (real)Rd[3 * t + i];
      p[P_ULO + i] = (real)ulo[3 * t + i];
      p[P_UHI + i] = (real)uhi[3 * t + i];
    }
    double Jr[9], Ji[9];
    inertia_inverse_rm(Jmat + 9 * t, Jr, Ji);
    for (int i = 0; i < 9; ++i) { p[P_J + i] = (real)Jr[i]; p[P_JI + i] = (real)Ji[i]; }
    p[P_TAU0] = (real)tau0[t];
    p[P_DTAU] = (real)dtau[t];
    p[P_DT] = (real)dt[t];
    // a float record carries the table clock as (hi, lo) pairs so that the field row of every knot is the double-precision
    // floor(fma(k + c, dtau, tau0)) of the fp64 builds; exact zeros in a double record
    p[P_TAU0L] = (real)(tau0[t] - (double)p[P_TAU0]);
    p[P_DTAUL] = (real)(dtau[t] - (double)p[P_DTAU]);
  }
}

// B tables [n_btab][n_tab][3] -> [n_btab][n_tab][4] (rows padded to 32 bytes for aligned 16-byte loads)
template <typename real>
void pack_btab(int64_t n_btab, int n_tab, const double* Btab, real* BT) {
  for (int64_t i = 0; i < n_btab * (int64_t)n_tab; ++i) {
    BT[4 * i + 0] = (real)Btab[3 * i + 0];
    BT[4 * i + 1] = (real)Btab[3 * i + 1];
    BT[4 * i + 2] = (real)Btab[3 * i + 2];
    BT[4 * i + 3] = 0;
  }
}

// element-wise export of the resident records into the ABI result arrays; `e` indexes knot records.
//   X (T,N,7) <- XU[.,.,0:7];  U (T,N-1,3) <- XU[.,.,7:10];  K (T,N-1,7,3)[t][k][j][a] <- KD[t][k][a*7+j]
template <typename real>
TSAT_DEV void export_record(int64_t e, int N, const int* nk, const real* XU, const real* KD, double* X, double* U,
                            double* K) {
  const int64_t t = e / N;
  const int k = (int)(e - t * N);
  const int n = nk ? nk[t] : N;           // knots this trajectory actually has; the rest of its slab is zero-filled
  const real* r = XU + (size_t)t * xu_stride<real>(N) + (size_t)k * XUW;
  if (X)
    for (int i = 0; i < 7; ++i) X[(size_t)e * 7 + i] = (k < n) ? (double)r[i] : 0.0;
  if (k < N - 1) {
    const size_t ek = (size_t)t * (N - 1) + k;
    const bool live = k < n - 1;
    if (U)
      for (int c = 0; c < 3; ++c) U[ek * 3 + c] = live ? (double)r[7 + c] : 0.0;
    if (K) {
      const real* kd = KD + (size_t)t * kd_stride<real>(N) + (size_t)k * KDW;
      for (int j = 0; j < 7; ++j)
        for (int c = 0; c < 3; ++c) K[ek * 21 + j * 3 + c] = live ? (double)kd[c * 7 + j] : 0.0;
    }
  }
}

// ---- closed-loop tracking (tsat_tvlqr_batch): marshalling --------------------------------------------------
// parameter record for the tracking kernel: x0 <- perturbed initial state, Qd[0:3]/P_QATT <- Q_lqr rate / attitude
// weights, Qfd[0:6] <- Qf_lqr, Rd <- R_lqr (src/TortoiseSat.jl:251-260)
template <typename real>
void pack_tv_params(int64_t T, const double* x0_sim, const double* xf, const double* tau0, const double* dtau,
                    const double* dt, const double* Jmat, const double* Qd6, const double* Qfd6, const double* Rd3,
                    real* P) {
  for (int64_t t = 0; t < T; ++t) {
    real* p = P + (size_t)t * PSTRIDE;
    for (int i = 0; i < PSTRIDE; ++i) p[i] = 0;
    for (int i = 0; i < 7; ++i) { p[P_X0 + i] = (real)x0_sim[7 * t + i]; p[P_XF + i] = (real)xf[7 * t + i]; }
    for (int i = 0; i < 3; ++i) {
      p[P_QD + i] = (real)Qd6[6 * t + i];
      p[P_QATT + i] = (real)Qd6[6 * t + 3 + i];
      p[P_RD + i] = (real)Rd3[3 * t + i];
    }
    for (int i = 0; i < 6; ++i) p[P_QFD + i] = (real)Qfd6[6 * t + i];
    double Jr[9], Ji[9];
    inertia_inverse_rm(Jmat + 9 * t, Jr, Ji);
    for (int i = 0; i < 9; ++i) { p[P_J + i] = (real)Jr[i]; p[P_JI + i] = (real)Ji[i]; }
    p[P_TAU0] = (real)tau0[t];
    p[P_DTAU] = (real)dtau[t];
    p[P_DT] = (real)dt[t];
  }
}
// X (T,N,7) + U (T,N-1,3) -> knot records [T][N][10]
template <typename real>
void pack_xu_records(int64_t T, int N, const double* X, const double* U, real* XU) {
  for (int64_t t = 0; t < T; ++t)
    for (int k = 0; k < N; ++k) {
      real* r = XU + ((size_t)t * N + k) * XUW;
      for (int i = 0; i < 7; ++i) r[i] = (real)X[((size_t)t * N + k) * 7 + i];
      for (int c = 0; c < 3; ++c) r[7 + c] = (k < N - 1) ? (real)U[((size_t)t * (N - 1) + k) * 3 + c] : (real)0;
    }
}
// simulated records + solver-sign gains -> X_sim, U_sim, K_lqr (3x6x(N-1)xT column-major, K_lqr = -K_solver)
template <typename real>
void unpack_tv(int64_t T, int N, const real* XS, const real* KD, double* X_sim, double* U_sim, double* K_lqr) {
  for (int64_t t = 0; t < T; ++t)
    for (int k = 0; k < N; ++k) {
      const real* r = XS + ((size_t)t * N + k) * XUW;
      if (X_sim)
        for (int i = 0; i < 7; ++i) X_sim[((size_t)t * N + k) * 7 + i] = (double)r[i];
      if (k < N - 1) {
        const size_t ek = (size_t)t * (N - 1) + k;
        if (U_sim)
          for (int c = 0; c < 3; ++c) U_sim[ek * 3 + c] = (double)r[7 + c];
        if (K_lqr)
          for (int j = 0; j < 6; ++j)
            for (int c = 0; c < 3; ++c) K_lqr[ek * 18 + j * 3 + c] = -(double)KD[ek * KDW + c * 7 + j];
      }
    }
}
inline std::string check_tv_options(const tsat_tvlqr_options& o) {
  if (o.n_knots < 2) return "n_knots must be >= 2";
  if (o.n_tab < 1) return "n_tab must be >= 1";
  if (o.min_steps < 0) return "min_steps must be >= 0";
  if (o.noise_mode != 0 && o.noise_mode != 1) return "noise_mode must be 0 (array / none) or 1 (generated)";
  if (o.rate_as_written != 0 && o.rate_as_written != 1)
    return "rate_as_written must be 0 or 1 (the field was a reserved word before version 300: initialise the struct with tsat_tvlqr_default_options)";
  return "";
}

// scales of the three draws of `simulator` (src/simulator.jl:5,10,22)
inline void tv_noise_defaults(tsat_tvlqr_options& o) {
  const double deg = 3.14159265358979323846 / 180.0;
  o.noise_mode = 0; o.rate_as_written = 0; o.noise_seed = 0;
  o.sigma_gyro = (0.38 * deg) * (0.38 * deg);
  o.sigma_att = deg * deg;
  o.field_amp = 1e-5 * 1e-5;
}

template <typename real>
inline void fill_tv_noise(const tsat_tvlqr_options& o, const long long* ids, TvArgs<real>& a) {
  a.noise_mode = o.noise_mode;
  a.k0 = (unsigned)(o.noise_seed & 0xFFFFFFFFull); a.k1 = (unsigned)(o.noise_seed >> 32);
  a.nid = ids;
  a.rate_as_written = o.rate_as_written;
  a.sg = (real)o.sigma_gyro; a.sa = (real)o.sigma_att; a.fa = (real)o.field_amp;
}

// Records of the field-table kernel (tsat_device.hpp, igrf12_eval): per (n, m) the Gauss coefficients advanced to
// `date` (src/igrf.jl:172-177), the derivative factors (src/dlegendre.jl:221-309), the Schmidt recurrence factors
// (src/legendre.jl:254-292) of the next function the sweep computes, and the two radial factors of degree n
// (src/igrf.jl:196-252).
inline void igrf_schmidt_factors(int n, int m, double& a_nm, double& b_nm) {   // P[n][m], n >= 2
  if (m < n) {
    const double aux = (double)((n - m) * (n + m));
    a_nm = std::sqrt(((2.0 * n - 1) * (2.0 * n - 1)) / aux);
    b_nm = std::sqrt(((double)(n + m - 1) * (n - m - 1)) / aux);
  } else {
    a_nm = std::sqrt((2.0 * n - 1) / (2.0 * n));
    b_nm = 0.0;
  }
}

inline void igrf_records(double date, double r_km, std::vector<double>& tab) {
  tab.assign((size_t)IGRF_NREC * IGRF_RECW, 0.0);
  const double dt = date - 2015.0, ratio = 6371.2 / r_km;
  double fact = ratio;
  int kg = 0, kh = 0;
  for (int n = 1; n <= IGRF_NMAX; ++n) {
    fact *= ratio;
    for (int m = 0; m <= n; ++m, ++kg) {
      double* rec = tab.data() + (size_t)kg * IGRF_RECW;
      rec[0] = IGRF12_G2015[kg] + IGRF12_GSV[kg] * dt;
      if (m > 0) { rec[1] = IGRF12_H2015[kh] + IGRF12_HSV[kh] * dt; ++kh; }
      if (m == 0) {
        const double aux = std::sqrt(n * (n + 1) / 2.0);
        rec[2] = -(0.5 * aux); rec[3] = -0.5 * aux;
      } else if (m == 1) {
        rec[2] = 0.5 * std::sqrt(2.0 * n * (n + 1)); rec[3] = -(0.5 * std::sqrt((double)(n + 2) * (n - 1)));
      } else {
        rec[2] = 0.5 * std::sqrt((double)(n + m) * (n - m + 1));
        rec[3] = (n != m) ? -(0.5 * std::sqrt((double)(n + m + 1) * (n - m))) : 0.0;
      }
      if (m < n) { if (n >= 2) igrf_schmidt_factors(n, m + 1, rec[4], rec[5]); }
      else if (n < IGRF_NMAX) igrf_schmidt_factors(n + 1, 0, rec[4], rec[5]);
      rec[6] = -(n + 1) / r_km;
      rec[7] = fact;
    }
  }
}

}  // namespace tsat
