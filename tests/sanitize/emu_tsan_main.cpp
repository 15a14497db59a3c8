// ThreadSanitizer run of the lane emulator (TEST INFRASTRUCTURE): the HIP kernel source, compiled with TSAT_EMU so that a
// wavefront is 64 host threads and both TSAT_SYNC() and TSAT_SYNC_LDS() are std::barrier waits, runs the workload of
// synth.hpp. Every LDS word and every HBM record that one lane writes and another lane reads must be separated by one of
// the two macros, or TSan reports the pair: this is the independent check of the kernels' intra-wave ordering protocol
//  — on the GPU TSAT_SYNC_LDS() orders LDS traffic only, so cross-lane traffic through HBM additionally has to
// sit behind TSAT_SYNC(), which is checked by review and by the GPU parity tests, not by this run.
#include "../emu/tsat_emu.cpp"
#include "synth.hpp"

int main() {
  synth::Api api;
#ifndef SYNTH_SOLVE_ONLY
  api.btable = emu_btable_batch;
  api.horizon = emu_horizon_batch;
  api.tvlqr = emu_tvlqr_batch;
  api.mpc = emu_mpc_batch;
#else
  api.btable = nullptr; api.horizon = nullptr; api.tvlqr = nullptr; api.mpc = nullptr;
#endif
  api.solve = [](const tsat_options* o, int64_t T, int64_t nb, const double* x0, const double* xf, const double* B, const int32_t* bi,
                 const double* tau0, const double* dtau, const double* dt, const double* J, const double* Qd, const double* Qfd,
                 const double* Rd, const double* ulo, const double* uhi, const double* U0, double* X, double* U, double* K,
                 tsat_stats* st, const int32_t* nk) {
    const int rc = emu_solve_batch(o, T, nb, x0, xf, B, bi, tau0, dtau, dt, J, Qd, Qfd, Rd, ulo, uhi, U0, X, U, K, st, nullptr, 0, nk);
    if (std::getenv("TSAT_EMU_SUSPEND_AT")) std::fprintf(stderr, "endgame parked %d\n", emu_parked());   // packed builds (tests/test_sanitizers.py)
    return rc;
  };
  tsat_options o;
  std::memset(&o, 0, sizeof(o));
  o.integrator = 3; o.precision = 64; o.max_outer = 20; o.max_inner = 50; o.max_linesearch = 20; o.dj_counter_limit = 10;
  o.cost_tol = 1e-4; o.grad_tol = 1e-5; o.constraint_tol = 1e-3; o.penalty_init = 1.0; o.penalty_scale = 10.0; o.penalty_max = 1e8;
  o.dual_max = 1e8; o.reg_init = 0.0; o.reg_scale = 1.6; o.reg_min = 1e-8; o.reg_max = 1e8; o.reg_fp = 10.0; o.ls_lower = 1e-8;
  o.ls_upper = 10.0; o.max_state = 1e8; o.u_scale = 1e-2; o.terminal_mask = 0x7f;
  tsat_tvlqr_options tv;
  std::memset(&tv, 0, sizeof(tv));
  tv.linearize_dt_sq = 1; tv.min_steps = 10; tv.u_scale = 1e-2; tv.w_tol = 0.05; tv.angle_tol = 0.08727;
#ifndef SYNTH_SOLVE_ONLY
  tv_noise_defaults(tv);
#endif
  return synth::run(api, o, tv);
}
