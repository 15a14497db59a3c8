// synth.hpp — one small deterministic workload for the sanitizer drivers (TEST INFRASTRUCTURE): two ragged slews whose
// field tables come from the field-table stage itself, then horizon, solve (both state-difference modes), tracking with
// in-kernel noise, and a two-step receding-horizon loop. The stage functions are passed in, so the same sequence drives
// the CPU oracle (AddressSanitizer + UBSan build) and the lane emulator of the HIP kernel source (ThreadSanitizer build).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../include/tortoise_hip.h"

namespace synth {

struct Api {
  int (*btable)(const tsat_btable_options*, int64_t, const double*, const double*, const double*, double*, double*);
  int (*horizon)(int64_t, int32_t, const double*, const double*, const double*, int32_t*, double*);
  int (*solve)(const tsat_options*, int64_t, int64_t, const double*, const double*, const double*, const int32_t*, const double*,
               const double*, const double*, const double*, const double*, const double*, const double*, const double*,
               const double*, const double*, double*, double*, double*, tsat_stats*, const int32_t*);
  int (*tvlqr)(const tsat_tvlqr_options*, int64_t, int64_t, const double*, const double*, const double*, const double*,
               const int32_t*, const double*, const double*, const double*, const double*, const double*, const double*,
               const double*, const double*, const double*, double*, double*, double*, tsat_tvlqr_stats*, const int32_t*,
               const int64_t*);
  int (*mpc)(const tsat_options*, int64_t, int64_t, const double*, const double*, const double*, const int32_t*, const double*,
             const double*, const double*, const double*, const double*, const double*, const double*, const double*,
             const double*, const double*, int32_t, int32_t, double*, double*, tsat_stats*, double*, double*, const int32_t*);
};

inline double checksum(const std::vector<double>& v) {
  double s = 0;
  for (size_t i = 0; i < v.size(); ++i) s += v[i] * (1.0 + 1e-3 * (double)(i % 7));
  return s;
}

// SYNTH_T > 2 (solve-only drivers of the packed builds: full and partial groups of four / eight trajectories per wavefront):
// trajectory t repeats the inputs of trajectory t % 2 with its own horizon and a perturbed initial attitude
#ifndef SYNTH_T
#define SYNTH_T 2
#endif
template <typename V> inline void tile2(V& v, int per, int T) {
  v.resize((size_t)per * T);
  for (int t = 2; t < T; ++t)
    for (int i = 0; i < per; ++i) v[(size_t)t * per + i] = v[(size_t)(t % 2) * per + i];
}

inline int run(const Api& api, const tsat_options& defaults, const tsat_tvlqr_options& tv_defaults) {
  const int T = SYNTH_T, N = 40, NH = 24;                 // NH: half-length of the field tables (2 NH = 48 rows)
  // ragged: chunk boundaries of the forward (32) sweeps on both sides; shortest possible, one past a 16-knot pass, ...
  const int32_t nk_all[12] = {40, 33, 2, 17, 40, 5, 29, 40, 16, 40, 3, 34};
  static_assert(SYNTH_T >= 2 && SYNTH_T <= 12, "SYNTH_T");
#ifndef SYNTH_SOLVE_ONLY
  static_assert(SYNTH_T == 2, "the whole pipeline is laid out for two orbits; more trajectories only in the solve-only drivers");
#endif
  const int32_t* nk = nk_all;
  std::vector<double> kep = {0.0, 6771.0, 96.6, 30.0, 0.0, 40.0, 0.01, 6900.0, 51.6, 200.0, 10.0, 300.0};
  std::vector<double> t0 = {0.0, 5.0}, tf = {NH * 0.2, 5.0 + NH * 0.2};
  std::vector<double> B((size_t)T * 2 * NH * 3), pos((size_t)T * (2 * NH + 1) * 3);
  tsat_btable_options bo{};
  bo.n_half = NH; bo.mjd = 58155.0; bo.gm = 3.986004418e5; bo.r_igrf_km = 6771.0; bo.date = 2019.0;
#ifdef SYNTH_SOLVE_ONLY   // builds that carry the solve kernel only (the dense build): an analytic dipole-like table instead
  for (int t = 0; t < T; ++t)
    for (int r = 0; r < 2 * NH; ++r) {
      const double ph = 1.1e-3 * 0.2 * r + 0.7 * t;
      double* b = &B[((size_t)t * 2 * NH + r) * 3];
      b[0] = 2.5e-5 * std::cos(ph); b[1] = -3e-6; b[2] = 5e-5 * std::sin(ph);
    }
  (void)bo;
#else
  if (api.btable(&bo, T, kep.data(), t0.data(), tf.data(), B.data(), pos.data())) return 10;
  std::vector<double> dtr = {0.2, 0.2}, cut = {1e6, 1e6}, cond(T);
  std::vector<int32_t> idx(T);
  if (api.horizon(T, 2 * NH - 1, B.data(), dtr.data(), cut.data(), idx.data(), cond.data())) return 11;
  std::printf("btable %.12e horizon %d %d\n", checksum(B), idx[0], idx[1]);
#endif

  std::vector<double> x0 = {0, 0, 0, 0.5, 0.5, -0.5, 0.5, 0.01, -0.02, 0.0, 0.8, 0.0, 0.6, 0.0};
  std::vector<double> xf = {0, 0, 0, M_SQRT1_2, M_SQRT1_2, 0, 0, 0, 0, 0, M_SQRT1_2, M_SQRT1_2, 0, 0};
  std::vector<double> tau0 = {0, 1.5}, dtau = {1.0, 0.9}, dt = {0.2, 0.2};
  std::vector<double> J = {1.25e-3, 0, 0, 0, 1.25e-3, 0, 0, 0, 1.25e-3, 2.0e-3, 1.0e-4, -2.0e-4, 1.0e-4, 1.5e-3, 3.0e-4, -2.0e-4, 3.0e-4, 2.5e-3};
  if (T > 2) {
    tile2(x0, 7, T); tile2(xf, 7, T); tile2(tau0, 1, T); tile2(dtau, 1, T); tile2(dt, 1, T); tile2(J, 9, T);
    for (int t = 2; t < T; ++t) {        // another initial attitude per trajectory (normalised inside the dynamics only)
      x0[7 * t + 3] += 0.05 * t; x0[7 * t + 5] -= 0.03 * t; x0[7 * t + 0] = 1e-3 * t;
    }
  }
  std::vector<double> Qd(7 * T), Qfd(7 * T), Rd(3 * T, 0.03), ulo(3 * T, -19.0), uhi(3 * T, 19.0), U0((size_t)T * (N - 1) * 3);
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < 7; ++i) { Qd[7 * t + i] = (i < 3) ? 40.0 : 100.0; Qfd[7 * t + i] = 10 * Qd[7 * t + i]; }
  for (size_t i = 0; i < U0.size(); ++i) U0[i] = 1e-3 * std::fabs(std::sin(0.37 * (double)i));
  std::vector<double> X((size_t)T * N * 7), U((size_t)T * (N - 1) * 3), K((size_t)T * (N - 1) * 21);
  std::vector<tsat_stats> st(T);
  for (int es = 0; es < 2; ++es)
    for (int integ = 3; integ <= 4; ++integ) {
      tsat_options o = defaults;
      o.n_knots = N; o.n_tab = 2 * NH; o.integrator = integ; o.precision = 64; o.max_outer = 2; o.max_inner = 3;
      o.dj_counter_limit = 1; o.error_state = es;
      if (api.solve(&o, T, T, x0.data(), xf.data(), B.data(), nullptr, tau0.data(), dtau.data(), dt.data(), J.data(), Qd.data(),
                    Qfd.data(), Rd.data(), ulo.data(), uhi.data(), U0.data(), X.data(), U.data(), K.data(), st.data(), nk))
        return 12;
      std::printf("solve es=%d rk%d X %.12e U %.12e K %.9e iters %d %d\n", es, integ, checksum(X), checksum(U), checksum(K),
                  st[0].inner_iters, st[1].inner_iters);
    }
#ifdef SYNTH_SOLVE_ONLY
  (void)tv_defaults;
  return 0;
#else
  tsat_tvlqr_options to = tv_defaults;
  to.n_knots = N; to.n_tab = 2 * NH; to.noise_mode = 1; to.noise_seed = 7;
  std::vector<double> Ql(6 * T, 10.0), Qfl(6 * T, 1000.0), Rl(3 * T, 500.0), Xs((size_t)T * N * 7), Us((size_t)T * (N - 1) * 3),
      Kl((size_t)T * (N - 1) * 18);
  std::vector<tsat_tvlqr_stats> ts(T);
  const int64_t ids[2] = {5, 9};
  if (api.tvlqr(&to, T, T, X.data(), U.data(), xf.data(), B.data(), nullptr, tau0.data(), dtau.data(), dt.data(), J.data(), Ql.data(),
                Qfl.data(), Rl.data(), x0.data(), nullptr, Xs.data(), Us.data(), Kl.data(), ts.data(), nk, ids))
    return 13;
  std::printf("tvlqr X %.12e U %.12e K %.9e slew %d %d\n", checksum(Xs), checksum(Us), checksum(Kl), ts[0].slew_index, ts[1].slew_index);
  {
    const int NM = 12, steps = 2;                        // receding horizon: 12-knot plan re-solved twice
    tsat_options o = defaults;
    o.n_knots = NM; o.n_tab = 2 * NH; o.integrator = 3; o.precision = 64; o.max_outer = 1; o.max_inner = 2; o.dj_counter_limit = 1;
    std::vector<double> U0m((size_t)T * (NM - 1) * 3, 1e-4), Xh((size_t)T * (steps + 1) * 7), Uh((size_t)T * steps * 3),
        Xl((size_t)T * NM * 7), Ul((size_t)T * (NM - 1) * 3);
    if (api.mpc(&o, T, T, x0.data(), xf.data(), B.data(), nullptr, tau0.data(), dtau.data(), dt.data(), J.data(), Qd.data(), Qfd.data(),
                Rd.data(), ulo.data(), uhi.data(), U0m.data(), steps, 4, Xh.data(), Uh.data(), st.data(), Xl.data(), Ul.data(), nullptr))
      return 14;
    std::printf("mpc X %.12e U %.12e\n", checksum(Xh), checksum(Uh));
  }
  return 0;
#endif
}

}  // namespace synth
