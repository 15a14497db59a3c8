// tsat_emu.cpp — CPU lane-emulator of the HIP solve kernel (TEST INFRASTRUCTURE).
//
// Compiles tortoisesat.jl_amd/csrc/tsat_device.hpp — the very source hipcc compiles for gfx950 — with
// TSAT_EMU: one wavefront = 64 host threads, __syncthreads() = std::barrier, LDS = a heap block.
// Lets the CPU-only test tier check the kernel's indexing / lane roles / control flow against the oracle;
// it is never used by the product and says nothing about performance.
#define TSAT_EMU
#include <atomic>
#include <barrier>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>
#include <memory>

// Two ways of running the 64 lanes of an emulated wavefront:
//  * TSAT_EMU_THREADS (the ThreadSanitizer drivers, tests/sanitize): 64 host threads, a barrier is std::barrier — real
//    concurrency, so that a missing barrier shows as a data race; one wavefront at a time;
//  * default (the functional tests): 64 FIBERS on one host thread, switched round-robin at the barriers — a barrier costs 64
//    user-level context switches instead of 64 futex round trips on an oversubscribed machine (5 - 10x faster), and
//    wavefronts run side by side on the host's cores. Same lane code, same barrier semantics, deterministic.
namespace tsat_emu {
#ifdef TSAT_EMU_THREADS
thread_local int g_lane = 0;
std::barrier<>* g_bar = nullptr;
void* g_lds = nullptr;     // one emulated wavefront at a time
double g_xch[3 * 64 * 16]; // scratch of the emulated cross-lane (DPP) reads: three blocks of 16 doubles per lane
int lane() { return g_lane; }
void sync() { g_bar->arrive_and_wait(); }
void* lds() { return g_lds; }
double* xch() { return g_xch; }
template <typename F>
void run_wave(size_t lds_bytes, F&& body) {
  std::vector<double> lds((lds_bytes + 7) / 8, 0.0);
  std::barrier<> bar(64);
  g_bar = &bar;
  g_lds = lds.data();
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l)
    th.emplace_back([&, l]() { g_lane = l; body(); });
  for (auto& t : th) t.join();
}
template <typename F>
void for_each_wave(int n, F&& fn) { for (int w = 0; w < n; ++w) fn(w); }
#else
extern "C" void tsat_ctx_switch(void** save_sp, void* to_sp);
// x86-64 SysV: callee-saved registers on the outgoing stack, swap stack pointers, restore, return into the other fiber
asm(R"(
.text
.globl tsat_ctx_switch
.type tsat_ctx_switch,@function
tsat_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size tsat_ctx_switch,.-tsat_ctx_switch
)");
struct Wave {
  static constexpr int W = 64;
  static constexpr size_t STACK = 512 * 1024;
  void* sp[W + 1] = {};                 // saved stack pointers of the lanes; [W] = the host thread's own
  bool done[W] = {};
  int cur = 0, arrived = 0, n_done = 0;
  unsigned gen = 0;
  void* lds = nullptr;
  double xch[3 * 64 * 16] = {};         // scratch of the emulated cross-lane (DPP) reads: three blocks of 16 doubles per lane
  std::function<void()> body;
  std::unique_ptr<char[]> stacks;
};
thread_local Wave* g_w = nullptr;
int lane() { return g_w->cur; }
void* lds() { return g_w->lds; }
double* xch() { return g_w->xch; }
static void yield_from(int from) {      // to the next lane that has not finished (round-robin); back here when it is our turn
  Wave* w = g_w;
  int to = from;
  do { to = (to + 1) % Wave::W; } while (w->done[to] && to != from);
  if (to == from) return;
  w->cur = to;
  tsat_ctx_switch(&w->sp[from], w->sp[to]);
}
void sync() {
  Wave* w = g_w;
  const unsigned g = w->gen;
  if (++w->arrived >= Wave::W - w->n_done) { w->arrived = 0; ++w->gen; }
  while (w->gen == g) yield_from(w->cur);
}
static void lane_entry() {
  Wave* w = g_w;
  w->body();
  const int me = w->cur;
  w->done[me] = true;
  ++w->n_done;
  if (w->arrived > 0 && w->arrived >= Wave::W - w->n_done) { w->arrived = 0; ++w->gen; }   // the others were waiting for a lane that has left
  void* dummy;
  if (w->n_done == Wave::W) tsat_ctx_switch(&dummy, w->sp[Wave::W]);      // the last lane returns to the host thread
  for (;;) {                                                              // a finished lane never runs again
    int to = me;
    do { to = (to + 1) % Wave::W; } while (w->done[to]);
    w->cur = to;
    tsat_ctx_switch(&dummy, w->sp[to]);
  }
}
template <typename F>
void run_wave(size_t lds_bytes, F&& body) {
  Wave w;
  std::vector<double> lds((lds_bytes + 7) / 8, 0.0);
  w.lds = lds.data();
  w.body = body;
  w.stacks.reset(new char[Wave::W * Wave::STACK]);
  for (int l = 0; l < Wave::W; ++l) {
    uintptr_t top = (reinterpret_cast<uintptr_t>(w.stacks.get()) + (size_t)(l + 1) * Wave::STACK) & ~(uintptr_t)15;
    void** s = reinterpret_cast<void**>(top);
    *--s = nullptr;                                  // return address of lane_entry (it never returns)
    *--s = reinterpret_cast<void*>(&lane_entry);     // `ret` of the first switch lands here, stack aligned as after a call
    for (int r = 0; r < 6; ++r) *--s = nullptr;      // rbp rbx r12 r13 r14 r15
    w.sp[l] = s;
  }
  Wave* prev = g_w;
  g_w = &w;
  w.cur = 0;
  tsat_ctx_switch(&w.sp[Wave::W], w.sp[0]);
  g_w = prev;
}
// wavefronts side by side on the host's cores (each host thread runs whole wavefronts, one after the other)
template <typename F>
void for_each_wave(int n, F&& fn) {
  const int nt = (int)std::min<unsigned>((unsigned)n, std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
  if (nt <= 1) { for (int w = 0; w < n; ++w) fn(w); return; }
  std::atomic<int> next{0};
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&]() { for (int w = next++; w < n; w = next++) fn(w); });
  for (auto& t : th) t.join();
}
#endif
}  // namespace tsat_emu

#include "../../tortoisesat.jl_amd/csrc/tsat_host_pack.hpp"
#ifdef TSAT_PACKED
#include "../../tortoisesat.jl_amd/csrc/tsat_packed.hpp"   // PK_G trajectories per emulated wavefront
#endif

using namespace tsat;

using R = tsat::cfg_real;   // storage type of the solve kernel's arrays (double in every build)

template <int INTEG, int DIAGJ, int ES>
static void run_block(const KArgs<R>& a, int traj) {
  tsat_emu::run_wave(LDS_BYTES, [&]() {
#ifdef TSAT_PACKED
    solve_group<R, INTEG, DIAGJ, ES>(a, traj);      // `traj` = wave index
#else
    solve_trajectory<R, INTEG, DIAGJ, ES>(a, traj);
#endif
  });
}

#ifdef TSAT_PACKED
template <int INTEG, int DIAGJ, int ES>
static void resume_block(const KArgs<R>& a, int w) {      // tsat_resume_kernel_packed
  tsat_emu::run_wave(LDS_BYTES, [&]() {
    (void)continue_trajectory<R, INTEG, DIAGJ, ES>(a, a.susp_ids[w], reinterpret_cast<const Resume<R>*>(a.susp_state)[w]);
  });
}
#endif
static int emu_parked_last = 0;
extern "C" int emu_parked(void) { return emu_parked_last; }     // trajectories the last packed emu_solve_batch parked

extern "C" int emu_lds_bytes(void) { return LDS_BYTES; }

extern "C" int emu_solve_batch(const tsat_options* o, int64_t T, int64_t n_btab, const double* x0, const double* xf,
                               const double* Btab, const int32_t* btab_idx, const double* tau0, const double* dtau,
                               const double* dt, const double* Jmat, const double* Qd, const double* Qfd,
                               const double* Rd, const double* ulo, const double* uhi, const double* U0, double* X,
                               double* U, double* K, tsat_stats* stats, double* trace, int trace_rows,
                               const int32_t* n_knots) {
  const int N = o->n_knots, n_tab = o->n_tab;
  if (!check_options(*o, N, n_tab, o->max_linesearch).empty()) return -1;
  const int max_ls = o->max_linesearch < NSTORE ? o->max_linesearch : NSTORE;   // stored candidate slots
  std::vector<R> P((size_t)T * PSTRIDE), BT((size_t)n_btab * n_tab * 4), U0r(U0, U0 + (size_t)T * (N - 1) * 3);
  std::vector<int> bidx(T);
  pack_params<R>(T, x0, xf, tau0, dtau, dt, Jmat, Qd, Qfd, Rd, ulo, uhi, P.data());
  pack_btab<R>(n_btab, n_tab, Btab, BT.data());
  for (int64_t t = 0; t < T; ++t) bidx[t] = btab_idx ? btab_idx[t] : (int)t;
  std::vector<R> XU((size_t)T * xu_stride<R>(N), (R)0), KD((size_t)T * kd_stride<R>(N), (R)0),
      LAM((size_t)T * lam_stride<R>(N), (R)0), CAND((size_t)T * max_ls * xu_stride<R>(N), (R)0);
  KArgs<R> a;
  a.T = (int)T; a.N = N; a.n_tab = n_tab; a.max_ls = max_ls; a.opt = *o;
#ifdef TSAT_PACKED
  a.max_ls = max_ls < PK_STORE ? max_ls : PK_STORE;      // as tsat_launch_solve_packed does
#endif
  a.P = P.data(); a.BT = BT.data(); a.bidx = bidx.data(); a.nk = n_knots; a.U0 = U0r.data();
  a.XU = XU.data(); a.KD = KD.data(); a.LAM = LAM.data(); a.CAND = CAND.data();
  a.stats = stats; a.trace = trace; a.trace_rows = trace_rows;
  std::vector<R> JW((size_t)((T + 3) / 4) * TSAT_JW_REALS_PER_4, (R)0);     // packed builds: Jacobian records in flight
  a.JW = JW.data();
  const int cls = inertia_class(T, Jmat);   // same variant selection as tsat_batch_upload
  using blk_t = void (*)(const KArgs<R>&, int);
  static const blk_t variants[2][3][2] = {
      {{run_block<3, 0, 0>, run_block<3, 0, 1>}, {run_block<3, 1, 0>, run_block<3, 1, 1>}, {run_block<3, 2, 0>, run_block<3, 2, 1>}},
      {{run_block<4, 0, 0>, run_block<4, 0, 1>}, {run_block<4, 1, 0>, run_block<4, 1, 1>}, {run_block<4, 2, 0>, run_block<4, 2, 1>}}};
  const blk_t blk = variants[o->integrator == 4 ? 1 : 0][cls][o->error_state ? 1 : 0];
#ifdef TSAT_PACKED
  // endgame of the launch as tsat_launch_solve_packed queues it (TSAT_EMU_SUSPEND_AT=n: park at n live trajectories)
  const char* sa = std::getenv("TSAT_EMU_SUSPEND_AT");
  int counters[2] = {(int)T, 0};
  std::vector<int> ids((size_t)T);
  std::vector<Resume<R>> parked((size_t)T);
  if (sa && std::atoi(sa) > 0) {
    a.suspend_at = std::atoi(sa); a.live = &counters[0]; a.susp_n = &counters[1]; a.susp_ids = ids.data(); a.susp_state = parked.data();
  }
  tsat_emu::for_each_wave(((int)T + PK_G - 1) / PK_G, [&](int w) { blk(a, w); });
  if (a.suspend_at) {
    using res_t = void (*)(const KArgs<R>&, int);
    static const res_t resume[2][3][2] = {
        {{resume_block<3, 0, 0>, resume_block<3, 0, 1>}, {resume_block<3, 1, 0>, resume_block<3, 1, 1>}, {resume_block<3, 2, 0>, resume_block<3, 2, 1>}},
        {{resume_block<4, 0, 0>, resume_block<4, 0, 1>}, {resume_block<4, 1, 0>, resume_block<4, 1, 1>}, {resume_block<4, 2, 0>, resume_block<4, 2, 1>}}};
    const res_t res = resume[o->integrator == 4 ? 1 : 0][cls][o->error_state ? 1 : 0];
    emu_parked_last = counters[1];
    a.live = nullptr;          // nothing parks in the second launch
    tsat_emu::for_each_wave(counters[1], [&](int w) { res(a, w); });
  }
#else
  tsat_emu::for_each_wave((int)T, [&](int t) { blk(a, t); });
#endif
  for (int64_t e = 0; e < T * (int64_t)N; ++e) export_record<R>(e, N, n_knots, XU.data(), KD.data(), X, U, K);
  return 0;
}

#if !defined(TSAT_DENSE)   // the dense build exists for the solve kernel only (tortoisesat.jl_amd/csrc/tsat_kernels_dense.hip)
template <int DIAGJ>
static void run_mpc_block(const MpcArgs<double>& a, int traj) {
  tsat_emu::run_wave((size_t)LDS_REALS * 8, [&]() { mpc_advance_trajectory<double, DIAGJ>(a, traj); });
}

// the loop of tsat_mpc_run: solve blocks, then advance blocks, n_steps times, on emulated device buffers
extern "C" int emu_mpc_batch(const tsat_options* o, int64_t T, int64_t n_btab, const double* x0, const double* xf,
                             const double* Btab, const int32_t* btab_idx, const double* tau0, const double* dtau,
                             const double* dt, const double* Jmat, const double* Qd, const double* Qfd,
                             const double* Rd, const double* ulo, const double* uhi, const double* U0, int32_t n_steps,
                             int32_t plant_integrator, double* X_hist, double* U_hist, tsat_stats* stats_last, double* X_last,
                             double* U_last, const int32_t* n_knots) {
  const int N = o->n_knots, n_tab = o->n_tab;
  if (!check_options(*o, N, n_tab, o->max_linesearch).empty()) return -1;
  const int max_ls = o->max_linesearch < NSTORE ? o->max_linesearch : NSTORE;
  std::vector<double> P((size_t)T * PSTRIDE), BT((size_t)n_btab * n_tab * 4), U0w(U0, U0 + (size_t)T * (N - 1) * 3);
  std::vector<int> bidx(T);
  pack_params<double>(T, x0, xf, tau0, dtau, dt, Jmat, Qd, Qfd, Rd, ulo, uhi, P.data());
  pack_btab<double>(n_btab, n_tab, Btab, BT.data());
  for (int64_t t = 0; t < T; ++t) bidx[t] = btab_idx ? btab_idx[t] : (int)t;
  std::vector<double> XU((size_t)T * N * XUW, 0.0), KD((size_t)T * (N - 1) * KDW, 0.0),
      LAM((size_t)T * (N - 1) * LMW, 0.0), CAND((size_t)T * max_ls * N * XUW, 0.0);
  KArgs<double> a;
  a.T = (int)T; a.N = N; a.n_tab = n_tab; a.max_ls = max_ls; a.opt = *o;
  a.P = P.data(); a.BT = BT.data(); a.bidx = bidx.data(); a.nk = n_knots; a.U0 = U0w.data();
  a.XU = XU.data(); a.KD = KD.data(); a.LAM = LAM.data(); a.CAND = CAND.data();
  a.stats = stats_last; a.trace = nullptr; a.trace_rows = 0;
  MpcArgs<double> m;
  m.T = (int)T; m.N = N; m.n_tab = n_tab; m.plant_integ = plant_integrator; m.n_steps = n_steps; m.us = o->u_scale;
  m.P = P.data(); m.BT = BT.data(); m.bidx = bidx.data(); m.nk = n_knots; m.XU = XU.data(); m.U0 = U0w.data();
  m.HX = X_hist; m.HU = U_hist; m.stats = nullptr; m.tally = nullptr;
  const int cls = inertia_class(T, Jmat);
  using blk_t = void (*)(const KArgs<double>&, int);
  static const blk_t variants[2][3][2] = {
      {{run_block<3, 0, 0>, run_block<3, 0, 1>}, {run_block<3, 1, 0>, run_block<3, 1, 1>}, {run_block<3, 2, 0>, run_block<3, 2, 1>}},
      {{run_block<4, 0, 0>, run_block<4, 0, 1>}, {run_block<4, 1, 0>, run_block<4, 1, 1>}, {run_block<4, 2, 0>, run_block<4, 2, 1>}}};
  const blk_t blk = variants[o->integrator == 4 ? 1 : 0][cls][o->error_state ? 1 : 0];
  for (int s = 0; s < n_steps; ++s) {
    tsat_emu::for_each_wave((int)T, [&](int t) { blk(a, t); });
    m.step = s;
    for (int t = 0; t < (int)T; ++t) {
      { if (cls == 2) run_mpc_block<2>(m, t); else if (cls == 1) run_mpc_block<1>(m, t); else run_mpc_block<0>(m, t); }
    }
  }
  for (int64_t e = 0; e < T * (int64_t)N; ++e) export_record<double>(e, N, n_knots, XU.data(), KD.data(), X_last, U_last, nullptr);
  return 0;
}

template <int DIAGJ>
static void run_tv_block(const TvArgs<double>& a, int traj) {
  tsat_emu::run_wave((size_t)LDS_REALS * 8, [&]() { tvlqr_trajectory<double, DIAGJ>(a, traj); });
}

extern "C" int emu_tvlqr_batch(const tsat_tvlqr_options* o, int64_t T, int64_t n_btab, const double* X, const double* U,
                               const double* xf, const double* Btab, const int32_t* btab_idx, const double* tau0,
                               const double* dtau, const double* dt, const double* Jmat, const double* Qd,
                               const double* Qfd, const double* Rd, const double* x0_sim, const double* noise,
                               double* X_sim, double* U_sim, double* K_lqr, tsat_tvlqr_stats* stats,
                               const int32_t* n_knots, const int64_t* noise_id) {
  if (!check_tv_options(*o).empty()) return -1;
  const int N = o->n_knots, n_tab = o->n_tab;
  std::vector<double> P((size_t)T * PSTRIDE), BT((size_t)n_btab * n_tab * 4), XUR((size_t)T * N * XUW),
      KD((size_t)T * (N - 1) * KDW, 0.0), XS((size_t)T * N * XUW, 0.0);
  std::vector<int> bidx(T);
  pack_tv_params<double>(T, x0_sim, xf, tau0, dtau, dt, Jmat, Qd, Qfd, Rd, P.data());
  pack_btab<double>(n_btab, n_tab, Btab, BT.data());
  pack_xu_records<double>(T, N, X, U, XUR.data());
  for (int64_t t = 0; t < T; ++t) bidx[t] = btab_idx ? btab_idx[t] : (int)t;
  TvArgs<double> a;
  a.T = (int)T; a.N = N; a.n_tab = n_tab; a.lin_sq = o->linearize_dt_sq; a.min_steps = o->min_steps;
  a.us = o->u_scale; a.w_tol = o->w_tol; a.ang_tol = o->angle_tol;
  fill_tv_noise<double>(*o, (const long long*)noise_id, a);
  a.P = P.data(); a.BT = BT.data(); a.bidx = bidx.data(); a.nk = n_knots; a.XUR = XUR.data(); a.NZ = noise; a.KD = KD.data(); a.XS = XS.data();
  a.stats = stats;
  const int cls = inertia_class(T, Jmat);
  tsat_emu::for_each_wave((int)T, [&](int t) {
    if (cls == 2) run_tv_block<2>(a, t); else if (cls == 1) run_tv_block<1>(a, t); else run_tv_block<0>(a, t);
  });
  unpack_tv<double>(T, N, XS.data(), KD.data(), X_sim, U_sim, K_lqr);
  return 0;
}

extern "C" int emu_horizon_batch(int64_t T, int32_t n_rows, const double* Btab, const double* dt_row, const double* cutoff,
                                 int32_t* tf_index, double* cond_at) {
  HzArgs<double> a;
  a.T = (int)T; a.n_rows = n_rows; a.BT = Btab; a.dt_row = dt_row; a.cutoff = cutoff; a.tf_index = tf_index; a.cond_at = cond_at;
  tsat_emu::for_each_wave((int)T, [&](int t) { tsat_emu::run_wave((size_t)LDS_REALS * 8, [&]() { horizon_trajectory<double>(a, t); }); });
  return 0;
}

extern "C" int emu_btable_batch(const tsat_btable_options* o, int64_t T, const double* kep, const double* t0, const double* tf,
                                double* Btab, double* pos) {
  const int N = o->n_half;
  std::vector<double> coef, P((size_t)T * 3 * (2 * N + 1));
  igrf_records(o->date, o->r_igrf_km, coef);
  BtArgs<double> a;
  a.T = (int)T; a.n_half = N; a.mjd = o->mjd; a.gm = o->gm; a.r_igrf_km = o->r_igrf_km;
  a.tab = coef.data(); a.kep = kep; a.t0 = t0; a.tf = tf; a.pos = P.data(); a.B = Btab;
  tsat_emu::for_each_wave((int)T, [&](int t) { tsat_emu::run_wave((size_t)LDS_REALS * 8, [&]() { btable_trajectory<double>(a, t); }); });
  if (pos) std::memcpy(pos, P.data(), P.size() * sizeof(double));
  return 0;
}
#endif  // TSAT_DENSE
