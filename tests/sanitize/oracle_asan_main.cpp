// AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU oracle (TEST INFRASTRUCTURE): the oracle's translation unit
// is compiled into this driver, which takes it through every batch entry point on the workload of synth.hpp.
#include "../../oracle/tsat_oracle.cpp"
#include "synth.hpp"

int main() {
  synth::Api api;
  api.btable = orc_btable_batch;
  api.horizon = [](int64_t T, int32_t n, const double* B, const double* d, const double* c, int32_t* i, double* ca) {
    return orc_horizon_batch(T, n, B, d, c, i, ca, nullptr);
  };
  api.solve = [](const tsat_options* o, int64_t T, int64_t nb, const double* x0, const double* xf, const double* B, const int32_t* bi,
                 const double* tau0, const double* dtau, const double* dt, const double* J, const double* Qd, const double* Qfd,
                 const double* Rd, const double* ulo, const double* uhi, const double* U0, double* X, double* U, double* K,
                 tsat_stats* st, const int32_t* nk) {
    return orc_solve_batch(o, T, nb, x0, xf, B, bi, tau0, dtau, dt, J, Qd, Qfd, Rd, ulo, uhi, U0, X, U, K, st, 2, nullptr, 0, nk);
  };
  api.tvlqr = [](const tsat_tvlqr_options* o, int64_t T, int64_t nb, const double* X, const double* U, const double* xf, const double* B,
                 const int32_t* bi, const double* tau0, const double* dtau, const double* dt, const double* J, const double* Qd,
                 const double* Qfd, const double* Rd, const double* x0s, const double* nz, double* Xs, double* Us, double* Kl,
                 tsat_tvlqr_stats* st, const int32_t* nk, const int64_t* id) {
    return orc_tvlqr_batch(o, T, nb, X, U, xf, B, bi, tau0, dtau, dt, J, Qd, Qfd, Rd, x0s, nz, Xs, Us, Kl, st, 2, nk, id);
  };
  api.mpc = [](const tsat_options* o, int64_t T, int64_t nb, const double* x0, const double* xf, const double* B, const int32_t* bi,
               const double* tau0, const double* dtau, const double* dt, const double* J, const double* Qd, const double* Qfd,
               const double* Rd, const double* ulo, const double* uhi, const double* U0, int32_t ns, int32_t pi, double* Xh, double* Uh,
               tsat_stats* st, double* Xl, double* Ul, const int32_t* nk) {
    return orc_mpc_batch(o, T, nb, x0, xf, B, bi, tau0, dtau, dt, J, Qd, Qfd, Rd, ulo, uhi, U0, ns, pi, Xh, Uh, st, Xl, Ul, 2, nk);
  };
  tsat_options o;
  orc_default_options(&o);
  tsat_tvlqr_options tv;
  orc_tvlqr_default_options(&tv);
  return synth::run(api, o, tv);
}
