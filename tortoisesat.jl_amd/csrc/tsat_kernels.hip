// tsat_kernels.hip — gfx950 kernels + the C ABI of include/tortoise_hip.h (libtortoise_hip.so).
//
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared tsat_kernels.hip -o libtortoise_hip.so
// One wavefront (a 64-thread workgroup) owns one trajectory for the whole AL-iLQR solve; see tsat_device.hpp.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen'ed on first use (tsat_comm_*), never linked
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "tsat_host_pack.hpp"

using namespace tsat;

// the dense build of the solve kernel lives in its own translation unit (tsat_kernels_dense.hip)
hipError_t tsat_launch_solve_dense(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
// the packed build — 8 trajectories per wavefront share the forward sweeps (tsat_kernels_packed.hip, tsat_packed.hpp)
hipError_t tsat_launch_solve_packed(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
hipError_t tsat_launch_solve_packed8(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
hipError_t tsat_launch_solve_packed8w(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
hipError_t tsat_launch_solve_packed16w(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
hipError_t tsat_launch_solve_packed4w(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
// the mixed-precision builds (options.precision = 32: float linearisation, everything else double; -DTSAT_JAC32 in tsat_device.hpp)
// of the dense, packed and packed8 layouts, on the very arrays of the fp64 builds (tsat_kernels_*_mixed.hip)
hipError_t tsat_launch_solve_dense_mixed(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
hipError_t tsat_launch_solve_packed_mixed(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
hipError_t tsat_launch_solve_packed_mixed8(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
hipError_t tsat_launch_solve_packed_mixed8w(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
hipError_t tsat_launch_solve_packed_mixed16w(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);
hipError_t tsat_launch_solve_packed_mixed4w(const KArgs<double>& a, int rk4, int inertia_class, int error_state, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
template <typename real, int INTEG, int DIAGJ, int ES>
__global__ __launch_bounds__(64) void tsat_solve_kernel(KArgs<real> a) {
  const int traj = blockIdx.x;
  if (traj >= a.T) return;
  solve_trajectory<real, INTEG, DIAGJ, ES>(a, traj);
}

template <typename real, int DIAGJ>
__global__ __launch_bounds__(64) void tsat_tvlqr_kernel(TvArgs<real> a) {
  const int traj = blockIdx.x;
  if (traj >= a.T) return;
  tvlqr_trajectory<real, DIAGJ>(a, traj);
}

template <typename real, int DIAGJ>
__global__ __launch_bounds__(64) void tsat_mpc_advance_kernel(MpcArgs<real> a) {
  const int traj = blockIdx.x;
  if (traj >= a.T) return;
  mpc_advance_trajectory<real, DIAGJ>(a, traj);
}

template <typename real>
__global__ __launch_bounds__(64) void tsat_btable_kernel(BtArgs<real> a) {
  const int traj = blockIdx.x;
  if (traj >= a.T) return;
  btable_trajectory<real>(a, traj);
}

template <typename real>
__global__ __launch_bounds__(64) void tsat_horizon_kernel(HzArgs<real> a) {
  const int traj = blockIdx.x;
  if (traj >= a.T) return;
  horizon_trajectory<real>(a, traj);
}

// resident field tables [T][rows][3] (tsat_btable_batch) -> the solver's padded tables [T][n_tab][4], first n_tab rows
__global__ __launch_bounds__(256) void tsat_pack_tables_kernel(int64_t n, int rows, int n_tab, const double* B, double* BT) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int64_t t = e / n_tab;
  const int r = (int)(e - t * n_tab);
  const double* src = B + ((size_t)t * rows + r) * 3;
  double* dst = BT + (size_t)e * 4;
  dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = 0.0;
}
template <typename real>
__global__ __launch_bounds__(256) void tsat_export_kernel(int64_t n_rec, int N, const int* nk, const real* XU,
                                                          const real* KD, double* X, double* U, double* K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_rec) export_record<real>(e, N, nk, XU, KD, X, U, K);
}

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
struct tsat_handle {
  int dev = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  // reserved batch
  int64_t T = 0, n_btab = 0;
  int N = 0, n_tab = 0, max_ls = 0, trace_rows = 0;
  bool uploaded = false, solved = false;
  int inertia_class = 0;      // 0 full, 1 every uploaded inertia tensor diagonal, 2 every one isotropic -> DIAGJ variant
  double *P = nullptr, *BT = nullptr, *U0 = nullptr, *XU = nullptr, *KD = nullptr, *LAM = nullptr, *CAND = nullptr;
  int* bidx = nullptr;
  int* nk = nullptr;          // per-trajectory knot counts (ragged batch) or null
  bool ragged = false;
  tsat_stats* stats = nullptr;
  double* trace = nullptr;
  int64_t bytes = 0;
  int endgame = -1;           // packed builds: park the last trajectories for a one-per-wavefront launch; -1 automatic, 0 never, n: at n live
  int variant = 0;            // solve-kernel build: 0 automatic (dense above 1024 trajectories), 1 wide, 2 dense
  // host copies of the small per-trajectory inputs of the last upload (26 doubles each), for tsat_tvlqr_resident
  std::vector<double> hx0, hxf, htau0, hdtau, hdt, hJ;
  int64_t bt_T = 0;      // field tables left on the device by the last tsat_btable_batch: [bt_T][bt_rows][3] in slot WS_BT_B
  int bt_rows = 0;
  int64_t bt_gen = 0;    // bumped by every tsat_btable_batch call that touches the resident tables (tsat_btable_generation)
  // grow-only device workspaces of the stages around the solve (tracking, horizon, field tables, MPC history, export):
  // allocated on first use and kept for the life of the handle, so repeated calls pay no hipMalloc / hipFree
  enum { WS_TV_NZ, WS_TV_KD, WS_TV_XS, WS_TV_NID, WS_TV_ST, WS_TV_P, WS_TVB_P, WS_TVB_BT, WS_TVB_XUR, WS_TVB_BI, WS_TVB_NK,
         WS_HZ_B, WS_HZ_DT, WS_HZ_CUT, WS_HZ_C, WS_HZ_I, WS_BT_COEF, WS_BT_KEP, WS_BT_T0, WS_BT_TF, WS_BT_POS, WS_BT_B,
         WS_JW, WS_MPC_HX, WS_MPC_HU,
         WS_DL_X, WS_DL_U, WS_DL_K, WS_AG_X, WS_AG_U, WS_AG_ST, WS_AG_XA, WS_AG_UA, WS_AG_STA,   // staging: WS_DL_X .. WS_AG_STA (tsat_workspace_trim)
         WS_AG_CHK, WS_MPC_TALLY, WS_ENDGAME, WS_COUNT };
  void* ws[WS_COUNT] = {};
  size_t ws_bytes[WS_COUNT] = {};
  // RCCL communicator of the sweep (tsat_comm_init): one rank per handle / GPU
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  int64_t comm_T = -1; int comm_N = -1;
  int64_t mpc_tally_T = 0;   // trajectories covered by the tally of the last tsat_mpc_run (slot WS_MPC_TALLY)   // shard shape every rank of the communicator was last found to agree on
};

namespace {

int fail(tsat_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}
#define TSAT_HIP(h, call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail((h), -10, std::string(#call) + ": " + hipGetErrorString(e_));                \
  } while (0)

void release(tsat_handle* h) {
  void* ptrs[] = {h->P, h->BT, h->U0, h->XU, h->KD, h->LAM, h->CAND, h->bidx, h->nk, h->stats, h->trace};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  h->P = h->BT = h->U0 = h->XU = h->KD = h->LAM = h->CAND = nullptr;
  h->bidx = nullptr; h->nk = nullptr; h->ragged = false; h->stats = nullptr; h->trace = nullptr;
  h->T = 0; h->bytes = 0; h->uploaded = h->solved = false;
}

// workspace `slot` with room for `bytes` (grow-only; contents are not preserved across a growth); nullptr on failure
void* ws_get(tsat_handle* h, int slot, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (h->ws[slot] && h->ws_bytes[slot] >= bytes) return h->ws[slot];
  if (h->ws[slot]) { (void)hipFree(h->ws[slot]); h->ws[slot] = nullptr; h->ws_bytes[slot] = 0; }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  h->ws[slot] = p;
  h->ws_bytes[slot] = bytes;
  return p;
}
void ws_release(tsat_handle* h) {
  for (int i = 0; i < tsat_handle::WS_COUNT; ++i)
    if (h->ws[i]) { (void)hipFree(h->ws[i]); h->ws[i] = nullptr; h->ws_bytes[i] = 0; }
}

template <typename Tp>
int dev_alloc(tsat_handle* h, Tp** p, size_t n) {
  const size_t b = n * sizeof(Tp);
  TSAT_HIP(h, hipMalloc(reinterpret_cast<void**>(p), b ? b : 16));
  h->bytes += (int64_t)b;
  return 0;
}

}  // namespace

extern "C" {

int tsat_version(void) { return 300; }

void tsat_default_options(tsat_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->integrator = 3; o->precision = 64;
  o->max_outer = 20; o->max_inner = 50;            // src/TortoiseSat.jl:195-196
  o->max_linesearch = 20; o->dj_counter_limit = 10;
  o->cost_tol = 1e-4; o->grad_tol = 1e-5; o->constraint_tol = 1e-3;
  o->penalty_init = 1.0; o->penalty_scale = 10.0; o->penalty_max = 1e8; o->dual_max = 1e8;
  o->reg_init = 0.0; o->reg_scale = 1.6; o->reg_min = 1e-8; o->reg_max = 1e8; o->reg_fp = 10.0;
  o->ls_lower = 1e-8; o->ls_upper = 10.0; o->max_state = 1e8;
  o->u_scale = 1e-2;                               // src/DerivFunction.jl:37
  o->terminal_mask = 0x7f; o->error_state = 0;
}

int tsat_create(tsat_handle** out, int device_id) {
  if (!out) return -1;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return -2;  // no GPU: there is no CPU fallback
  if (device_id < 0 || device_id >= n) return -3;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return -4;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return -5;  // code object is gfx950-only
  if (hipSetDevice(device_id) != hipSuccess) return -6;
  tsat_handle* h = new tsat_handle();
  h->dev = device_id;
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    delete h;
    return -7;
  }
  *out = h;
  return 0;
}

int tsat_destroy(tsat_handle* h) {
  if (!h) return -1;
  (void)hipSetDevice(h->dev);
  (void)tsat_comm_destroy(h);
  release(h);
  ws_release(h);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

const char* tsat_last_error(const tsat_handle* h) { return h ? h->err.c_str() : "null handle"; }

int64_t tsat_batch_bytes(const tsat_handle* h) { return h ? h->bytes : 0; }

int64_t tsat_workspace_bytes(const tsat_handle* h) {
  int64_t b = 0;
  if (h)
    for (int i = 0; i < tsat_handle::WS_COUNT; ++i) b += (int64_t)h->ws_bytes[i];
  return b;
}

int tsat_workspace_trim(tsat_handle* h, int32_t what) {
  if (!h) return -1;
  if (what != 0 && what != 1) return fail(h, -1, "what must be 0 (staging buffers of downloads and gathers) or 1 (every workspace)");
  TSAT_HIP(h, hipSetDevice(h->dev));
  TSAT_HIP(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < tsat_handle::WS_COUNT; ++i) {
    const bool staging = i >= tsat_handle::WS_DL_X && i <= tsat_handle::WS_AG_STA;
    if (!h->ws[i] || !(what == 1 || staging)) continue;
    (void)hipFree(h->ws[i]);
    h->ws[i] = nullptr; h->ws_bytes[i] = 0;
    if (i == tsat_handle::WS_BT_B) { h->bt_T = 0; h->bt_rows = 0; h->bt_gen++; }     // the resident field tables are gone
  }
  return 0;
}

int64_t tsat_btable_generation(const tsat_handle* h) { return h ? h->bt_gen : 0; }

int tsat_batch_reserve(tsat_handle* h, int64_t T, int32_t n_knots, int32_t n_tab, int64_t n_btab,
                       int32_t max_linesearch) {
  if (!h) return -1;
  if (T < 1 || n_knots < 2 || n_tab < 1 || n_btab < 1) return fail(h, -1, "bad batch dimensions");
  if (max_linesearch < 1 || max_linesearch > TSAT_MAX_LINESEARCH) return fail(h, -1, "max_linesearch must be in [1,32]");
  TSAT_HIP(h, hipSetDevice(h->dev));
  if (h->T == T && h->N == n_knots && h->n_tab == n_tab && h->n_btab == n_btab && h->max_ls == max_linesearch)
    return 0;
  const int trows = h->trace_rows;
  release(h);
  const size_t N = (size_t)n_knots, Tn = (size_t)T;
  int rc = 0;
  rc |= dev_alloc(h, &h->P, Tn * PSTRIDE);
  rc |= dev_alloc(h, &h->BT, (size_t)n_btab * n_tab * 4);
  rc |= dev_alloc(h, &h->bidx, Tn);
  rc |= dev_alloc(h, &h->nk, Tn);
  rc |= dev_alloc(h, &h->U0, Tn * (N - 1) * 3);
  rc |= dev_alloc(h, &h->XU, Tn * N * XUW);
  rc |= dev_alloc(h, &h->KD, Tn * (N - 1) * KDW);
  rc |= dev_alloc(h, &h->LAM, Tn * (N - 1) * LMW);
  const int slots = max_linesearch < NSTORE ? max_linesearch : NSTORE;   // stored candidates (tsat_device.hpp)
  rc |= dev_alloc(h, &h->CAND, Tn * (size_t)slots * N * XUW);
  rc |= dev_alloc(h, &h->stats, Tn);
  if (trows > 0) rc |= dev_alloc(h, &h->trace, Tn * (size_t)trows * 8);
  if (rc) { release(h); return -10; }
  h->T = T; h->N = n_knots; h->n_tab = n_tab; h->n_btab = n_btab; h->max_ls = max_linesearch;
  h->trace_rows = trows;
  return 0;
}

int tsat_batch_trace(tsat_handle* h, int32_t rows) {
  if (!h || rows < 0) return -1;
  TSAT_HIP(h, hipSetDevice(h->dev));
  if (h->trace) { (void)hipFree(h->trace); h->trace = nullptr; }
  h->trace_rows = rows;
  if (rows > 0 && h->T > 0) {
    if (dev_alloc(h, &h->trace, (size_t)h->T * rows * 8)) return -10;
  }
  return 0;
}

int tsat_batch_upload(tsat_handle* h, const double* x0, const double* xf, const double* Btab,
                      const int32_t* btab_idx, const double* tau0, const double* dtau, const double* dt,
                      const double* Jmat, const double* Qd, const double* Qfd, const double* Rd,
                      const double* ulo, const double* uhi, const double* U0) {
  if (!h) return -1;
  if (h->T < 1) return fail(h, -1, "tsat_batch_reserve has not been called");
  if (!x0 || !xf || !tau0 || !dtau || !dt || !Jmat || !Qd || !Qfd || !Rd || !ulo || !uhi || !U0)
    return fail(h, -1, "null input array");
  if (!Btab && (h->bt_T != h->T || h->n_btab != h->T || h->n_tab > h->bt_rows))
    return fail(h, -1, "Btab is NULL but the field tables left by the last tsat_btable_batch do not cover this batch "
                       "(need one table per trajectory, n_btab = T, and n_tab <= its 2 n_half rows)");
  if (!btab_idx && h->n_btab != h->T) return fail(h, -1, "btab_idx is NULL but n_btab != T");
  TSAT_HIP(h, hipSetDevice(h->dev));
  const int64_t T = h->T;
  std::vector<int> bi((size_t)T);
  for (int64_t t = 0; t < T; ++t) {
    const int64_t v = btab_idx ? btab_idx[t] : t;
    if (v < 0 || v >= h->n_btab) return fail(h, -1, "btab_idx out of range");
    bi[(size_t)t] = (int)v;
  }
  const std::string bad = check_inputs(T, x0, xf, tau0, dtau, dt, Jmat, Qd, Qfd, Rd, ulo, uhi);
  if (!bad.empty()) return fail(h, -1, bad);
  h->inertia_class = inertia_class(T, Jmat);
  std::vector<double> P((size_t)T * PSTRIDE);
  pack_params<double>(T, x0, xf, tau0, dtau, dt, Jmat, Qd, Qfd, Rd, ulo, uhi, P.data());
  TSAT_HIP(h, hipMemcpy(h->P, P.data(), P.size() * sizeof(double), hipMemcpyHostToDevice));
  if (Btab) {
    std::vector<double> BT((size_t)h->n_btab * h->n_tab * 4);
    pack_btab<double>(h->n_btab, h->n_tab, Btab, BT.data());
    TSAT_HIP(h, hipMemcpy(h->BT, BT.data(), BT.size() * sizeof(double), hipMemcpyHostToDevice));
  } else {   // device to device: the tables tsat_btable_batch left in its workspace, padded to the solver's row layout
    const int64_t n = T * (int64_t)h->n_tab;
    const double* B = (const double*)ws_get(h, tsat_handle::WS_BT_B, (size_t)h->bt_T * h->bt_rows * 3 * 8);
    if (!B) return fail(h, -10, "resident field tables are gone");
    hipLaunchKernelGGL(tsat_pack_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, n, h->bt_rows, h->n_tab, B, h->BT);
    TSAT_HIP(h, hipGetLastError());
    TSAT_HIP(h, hipStreamSynchronize(h->stream));
  }
  TSAT_HIP(h, hipMemcpy(h->bidx, bi.data(), bi.size() * sizeof(int), hipMemcpyHostToDevice));
  TSAT_HIP(h, hipMemcpy(h->U0, U0, (size_t)T * (h->N - 1) * 3 * sizeof(double), hipMemcpyHostToDevice));
  h->hx0.assign(x0, x0 + 7 * T); h->hxf.assign(xf, xf + 7 * T); h->htau0.assign(tau0, tau0 + T);
  h->hdtau.assign(dtau, dtau + T); h->hdt.assign(dt, dt + T); h->hJ.assign(Jmat, Jmat + 9 * T);
  h->uploaded = true;
  h->solved = false;
  h->ragged = false;   // a fresh upload is a uniform batch until tsat_batch_knots says otherwise
  return 0;
}

int tsat_batch_knots(tsat_handle* h, const int32_t* n_knots) {
  if (!h) return -1;
  if (!h->uploaded) return fail(h, -1, "tsat_batch_upload has not been called");
  if (!n_knots) { h->ragged = false; return 0; }
  for (int64_t t = 0; t < h->T; ++t)
    if (n_knots[t] < 2 || n_knots[t] > h->N) return fail(h, -1, "n_knots[t] must be in [2, N]");
  TSAT_HIP(h, hipSetDevice(h->dev));
  TSAT_HIP(h, hipMemcpy(h->nk, n_knots, (size_t)h->T * sizeof(int), hipMemcpyHostToDevice));
  h->ragged = true;
  h->solved = false;
  return 0;
}

namespace {
using solve_kern_t = void (*)(KArgs<double>);
// kernel variant: integrator x inertia class (full / diagonal / isotropic) x error-state mode. LDS is a static
// module-level array (tsat_device.hpp): nothing dynamic to request at launch.
solve_kern_t solve_variant(const tsat_handle* h, const tsat_options* o) {
  static const solve_kern_t variants[2][3][2] = {
      {{tsat_solve_kernel<double, 3, 0, 0>, tsat_solve_kernel<double, 3, 0, 1>},
       {tsat_solve_kernel<double, 3, 1, 0>, tsat_solve_kernel<double, 3, 1, 1>},
       {tsat_solve_kernel<double, 3, 2, 0>, tsat_solve_kernel<double, 3, 2, 1>}},
      {{tsat_solve_kernel<double, 4, 0, 0>, tsat_solve_kernel<double, 4, 0, 1>},
       {tsat_solve_kernel<double, 4, 1, 0>, tsat_solve_kernel<double, 4, 1, 1>},
       {tsat_solve_kernel<double, 4, 2, 0>, tsat_solve_kernel<double, 4, 2, 1>}}};
  return variants[o->integrator == 4 ? 1 : 0][h->inertia_class][o->error_state ? 1 : 0];
}

// Build by batch size (measured on one MI355X, profiles/r04/build_by_batch_size.txt): one wavefront per SIMD and trajectory (wide
// build) while the batch fits the GPU that way — 256 CUs x 4 SIMDs; two wavefronts per SIMD (dense build) up to twice that; from
// there the packed builds at ONE wavefront per SIMD, whose wavefronts own several trajectories (40 KB of LDS keep twelve of a
// backward pass's sixteen knot records on the chip, all sixteen float ones): four per wavefront up to one round of the 1024 SIMDs
// (packed4w, 2048 .. 4096), eight per wavefront up to one round (packed8w, .. 8192), sixteen from one full round on (packed16w);
// between 8192 and 16384, and from there on with a long iteration budget, eight per wavefront at two wavefronts per SIMD
// (packed8: selected_build). packed (four per wavefront at two per SIMD) is no longer chosen automatically.
constexpr int64_t TSAT_WIDE_MAX_T = 1024;
constexpr int64_t TSAT_PACKED_MIN_T = 2048, TSAT_PACKED4W_MAX_T = 4096;
constexpr int64_t TSAT_PACKED8W_MAX_T = 8192;
constexpr int64_t TSAT_PACKED16W_MIN_T = 16384, TSAT_LONG_BUDGET = 100;     // (budget = max_outer x max_inner)
// the build (1 wide, 2 dense, 3 packed, 4 packed8, 5 packed8w, 6 packed16w, 7 packed4w) that (h->variant, batch size, precision) selects.
// precision = 32 — the mixed-precision builds — has no wide layout: below 2048 trajectories its dense build runs (59-knot Jacobian
// passes in the 20 KB of two wavefronts per SIMD, which the double records do not allow)
int selected_build(const tsat_handle* h, int precision, int64_t budget) {
  if (h->variant >= 3) return h->variant;
  // from one full round on: sixteen per wavefront — unless the iteration budget is long enough to spread the trajectories' iteration
  // counts widely (3 x 50 on the inclination sweep: 30 ... 150), where a wavefront lasts as long as the slowest of its trajectories
  // and eight per wavefront on 2048 wavefront slots lose less (65536 trajectories: 3.91 s against 4.27 s; 5 x 10: 0.47 against 0.43 s)
  if (h->variant == 0 && h->T >= TSAT_PACKED16W_MIN_T) return budget >= TSAT_LONG_BUDGET ? 4 : 6;
  if (h->variant == 0 && h->T > TSAT_PACKED8W_MAX_T) return 4;
  if (h->variant == 0 && h->T > TSAT_PACKED4W_MAX_T) return 5;
  if (h->variant == 0 && h->T >= TSAT_PACKED_MIN_T) return 7;
  if (precision == 32) return 2;
  return (h->variant == 2 || (h->variant != 1 && h->T > TSAT_WIDE_MAX_T)) ? 2 : 1;
}
hipError_t launch_solve(const tsat_handle* h, const tsat_options* o, const KArgs<double>& a) {
  const int build = selected_build(h, o->precision, (int64_t)o->max_outer * o->max_inner), rk4 = o->integrator == 4;
  if (o->precision == 32) {
    if (build == 7) return tsat_launch_solve_packed_mixed4w(a, rk4, h->inertia_class, o->error_state, h->stream);
    if (build == 6) return tsat_launch_solve_packed_mixed16w(a, rk4, h->inertia_class, o->error_state, h->stream);
    if (build == 5) return tsat_launch_solve_packed_mixed8w(a, rk4, h->inertia_class, o->error_state, h->stream);
    if (build == 4) return tsat_launch_solve_packed_mixed8(a, rk4, h->inertia_class, o->error_state, h->stream);
    if (build == 3) return tsat_launch_solve_packed_mixed(a, rk4, h->inertia_class, o->error_state, h->stream);
    return tsat_launch_solve_dense_mixed(a, rk4, h->inertia_class, o->error_state, h->stream);
  }
  if (build == 7) return tsat_launch_solve_packed4w(a, rk4, h->inertia_class, o->error_state, h->stream);
  if (build == 6) return tsat_launch_solve_packed16w(a, rk4, h->inertia_class, o->error_state, h->stream);
  if (build == 5) return tsat_launch_solve_packed8w(a, rk4, h->inertia_class, o->error_state, h->stream);
  if (build == 4) return tsat_launch_solve_packed8(a, rk4, h->inertia_class, o->error_state, h->stream);
  if (build == 3) return tsat_launch_solve_packed(a, rk4, h->inertia_class, o->error_state, h->stream);
  if (build == 2) return tsat_launch_solve_dense(a, rk4, h->inertia_class, o->error_state, h->stream);
  hipLaunchKernelGGL(solve_variant(h, o), dim3((unsigned)h->T), dim3(64), 0, h->stream, a);
  return hipGetLastError();
}

// Jacobian-record workspace of the packed builds (one block per wavefront, i.e. per four trajectories at most), persistent in
// the handle; sized for double records, the mixed-precision builds (float records) use half of it
void* packed_workspace(tsat_handle* h) {
  return ws_get(h, tsat_handle::WS_JW, (size_t)((h->T + 3) / 4) * TSAT_JW_REALS_PER_4 * sizeof(double));
}

// does the build that (h->variant, batch size, precision) selects need the packed builds' Jacobian workspace a.JW?
bool uses_packed_build(const tsat_handle* h, int precision) { return selected_build(h, precision, 0) >= 3; }     // (whatever the budget)

// Endgame of a packed launch (tsat_packed.hpp, suspend_if_endgame). Automatic: once an eighth of the batch (profiles/r04/endgame_sweep.txt), at most the 2048
// wavefront slots of the machine (256 CUs x 4 SIMDs x 2), is all that still iterates — and only for iteration budgets long
// enough to spread the trajectories (the 1 x 3 budget of the receding-horizon loop ends all of them together).
// One block for both precisions: two counters, then T ids, then T Resume records (sized for doubles).
struct EndgameArgs { int suspend_at = 0; int *live = nullptr, *susp_n = nullptr, *susp_ids = nullptr; void* susp_state = nullptr; };
int endgame_threshold(const tsat_handle* h, const tsat_options* o) {
  int at = h->endgame;
  if (at < 0) at = ((int64_t)o->max_outer * o->max_inner >= 20) ? (int)std::min<int64_t>(2048, h->T / 8) : 0;
  return at <= 0 ? 0 : (int)std::min<int64_t>(at, h->T);
}
EndgameArgs endgame_args(tsat_handle* h, const tsat_options* o) {
  EndgameArgs a;
  const int at = endgame_threshold(h, o);
  if (at <= 0) return a;
  const size_t T = (size_t)h->T, ids_at = 16, st_at = (ids_at + T * sizeof(int) + 15) / 16 * 16;
  char* w = (char*)ws_get(h, tsat_handle::WS_ENDGAME, st_at + T * sizeof(Resume<double>));
  if (!w) return a;     // no endgame then: the packed kernel runs every trajectory to its end, as before
  a.suspend_at = (int)std::min<int64_t>(at, h->T);
  a.live = (int*)w; a.susp_n = (int*)w + 1; a.susp_ids = (int*)(w + ids_at); a.susp_state = w + st_at;
  return a;
}

KArgs<double> solve_args(tsat_handle* h, const tsat_options* o) {
  KArgs<double> a;
  a.T = (int)h->T; a.N = h->N; a.n_tab = h->n_tab; a.max_ls = h->max_ls < NSTORE ? h->max_ls : NSTORE; a.opt = *o;
  a.P = h->P; a.BT = h->BT; a.bidx = h->bidx; a.nk = h->ragged ? h->nk : nullptr; a.U0 = h->U0;
  a.XU = h->XU; a.KD = h->KD; a.LAM = h->LAM; a.CAND = h->CAND;
  a.stats = h->stats; a.trace = h->trace; a.trace_rows = h->trace ? h->trace_rows : 0;
  a.JW = uses_packed_build(h, 64) ? (double*)packed_workspace(h) : nullptr;
  if (a.JW) {
    const EndgameArgs e = endgame_args(h, o);
    a.suspend_at = e.suspend_at; a.live = e.live; a.susp_n = e.susp_n; a.susp_ids = e.susp_ids; a.susp_state = e.susp_state;
  }
  return a;
}

}  // namespace

int tsat_batch_run(tsat_handle* h, const tsat_options* o, float* kernel_ms) {
  if (!h || !o) return -1;
  if (!h->uploaded) return fail(h, -1, "tsat_batch_upload has not been called");
  const std::string why = check_options(*o, h->N, h->n_tab, h->max_ls);
  if (!why.empty()) return fail(h, -1, why);
  TSAT_HIP(h, hipSetDevice(h->dev));
  const KArgs<double> a = solve_args(h, o);
  if (!a.JW && uses_packed_build(h, 64))
    return fail(h, -10, "device allocation of the packed builds' Jacobian workspace failed");
  if (h->trace) TSAT_HIP(h, hipMemsetAsync(h->trace, 0, (size_t)h->T * h->trace_rows * 8 * sizeof(double), h->stream));
  TSAT_HIP(h, hipEventRecord(h->ev0, h->stream));
  TSAT_HIP(h, launch_solve(h, o, a));
  TSAT_HIP(h, hipEventRecord(h->ev1, h->stream));
  TSAT_HIP(h, hipStreamSynchronize(h->stream));
  if (kernel_ms) TSAT_HIP(h, hipEventElapsedTime(kernel_ms, h->ev0, h->ev1));
  h->solved = true;
  return 0;
}

int tsat_set_kernel_variant(tsat_handle* h, int32_t variant) {
  if (!h) return -1;
  if (!(variant >= 0 && variant <= 7))
    return fail(h, -1, "variant must be 0 (automatic), 1 (wide), 2 (dense), 3 (packed: 4 trajectories per wavefront), 4 (packed8), 5 (packed8w: 8, one wavefront per SIMD) or 6 (packed16w) or 7 (packed4w: 4, one wavefront per SIMD)");
  h->variant = variant;
  return 0;
}

int tsat_selected_build(tsat_handle* h, const tsat_options* o, int32_t* build, int32_t* endgame_at) {
  if (!h || !o) return -1;
  if (h->T <= 0) return fail(h, -1, "tsat_batch_reserve has not been called");
  const int b = selected_build(h, o->precision, (int64_t)o->max_outer * o->max_inner);
  if (build) *build = b;
  if (endgame_at) *endgame_at = b >= 3 ? endgame_threshold(h, o) : 0;
  return 0;
}

int tsat_set_endgame(tsat_handle* h, int32_t suspend_at) {
  if (!h) return -1;
  if (suspend_at < -1) return fail(h, -1, "suspend_at must be -1 (automatic), 0 (never) or the live count at which the packed builds park their trajectories");
  h->endgame = suspend_at;
  return 0;
}

int tsat_mpc_run(tsat_handle* h, const tsat_options* o, int32_t n_steps, int32_t plant_integrator, double* X_hist,
                 double* U_hist, tsat_stats* stats_last, float* solve_ms) {
  if (!h || !o) return -1;
  if (!h->uploaded) return fail(h, -1, "tsat_batch_upload has not been called");
  const std::string why = check_options(*o, h->N, h->n_tab, h->max_ls);
  if (!why.empty()) return fail(h, -1, why);
  if (n_steps < 1) return fail(h, -1, "n_steps must be >= 1");
  if (o->precision != 64) return fail(h, -1, "tsat_mpc_run runs the fp64 build only (precision must be 64)");
  if (plant_integrator != 3 && plant_integrator != 4) return fail(h, -1, "plant_integrator must be 3 (rk3) or 4 (rk4)");
  if (!X_hist || !U_hist) return fail(h, -1, "null array");
  TSAT_HIP(h, hipSetDevice(h->dev));
  const size_t T = (size_t)h->T, nX = T * ((size_t)n_steps + 1) * 7, nU = T * (size_t)n_steps * 3;
  double* dHX = (double*)ws_get(h, tsat_handle::WS_MPC_HX, nX * 8);
  double* dHU = (double*)ws_get(h, tsat_handle::WS_MPC_HU, nU * 8);
  long long* dTally = (long long*)ws_get(h, tsat_handle::WS_MPC_TALLY, T * 4 * sizeof(long long));
  if (!dHX || !dHU || !dTally) return fail(h, -10, "device allocation failed in tsat_mpc_run");
  TSAT_HIP(h, hipMemsetAsync(dTally, 0, T * 4 * sizeof(long long), h->stream));
  h->mpc_tally_T = (int64_t)T;
  const KArgs<double> a = solve_args(h, o);
  if (!a.JW && uses_packed_build(h, 64))     // a packed kernel launched with JW = nullptr would fault on the device
    return fail(h, -10, "device allocation of the packed builds' Jacobian workspace failed");
  MpcArgs<double> m;
  m.T = (int)h->T; m.N = h->N; m.n_tab = h->n_tab; m.plant_integ = plant_integrator; m.n_steps = n_steps; m.us = o->u_scale;
  m.P = h->P; m.BT = h->BT; m.bidx = h->bidx; m.nk = h->ragged ? h->nk : nullptr; m.XU = h->XU; m.U0 = h->U0;
  m.HX = dHX; m.HU = dHU; m.stats = h->stats; m.tally = dTally;
  auto adv = h->inertia_class == 2 ? tsat_mpc_advance_kernel<double, 2>
                                   : (h->inertia_class == 1 ? tsat_mpc_advance_kernel<double, 1> : tsat_mpc_advance_kernel<double, 0>);
  int rc = 0;
  if (hipEventRecord(h->ev0, h->stream) != hipSuccess) rc = -10;
  for (int s = 0; s < n_steps && !rc; ++s) {   // 2 n_steps launches queued back to back; the stream orders them
    if (launch_solve(h, o, a) != hipSuccess) rc = -10;
    m.step = s;
    hipLaunchKernelGGL(adv, dim3((unsigned)h->T), dim3(64), 0, h->stream, m);
    if (hipGetLastError() != hipSuccess) rc = -10;
  }
  if (!rc && (hipEventRecord(h->ev1, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)) rc = -10;
  if (!rc && solve_ms && hipEventElapsedTime(solve_ms, h->ev0, h->ev1) != hipSuccess) rc = -10;
  if (!rc && hipMemcpy(X_hist, dHX, nX * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && hipMemcpy(U_hist, dHU, nU * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && stats_last && hipMemcpy(stats_last, h->stats, T * sizeof(tsat_stats), hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  // the loop advanced x0 and tau0 inside the device parameter records: refresh the host mirrors tsat_tvlqr_resident packs
  // its own records from, so that tracking after an MPC run linearises against the field rows the last plan was solved on
  std::vector<double> Pb(T * PSTRIDE);
  if (!rc && hipMemcpy(Pb.data(), h->P, Pb.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (rc) { h->solved = false; return fail(h, rc, "launch or copy failed in tsat_mpc_run"); }
  for (size_t t = 0; t < T; ++t) {
    for (int i = 0; i < 7; ++i) h->hx0[7 * t + i] = Pb[t * PSTRIDE + P_X0 + i];
    h->htau0[t] = Pb[t * PSTRIDE + P_TAU0];
  }
  h->solved = true;
  return 0;
}

int tsat_mpc_tally(tsat_handle* h, int64_t* tally) {
  if (!h || !tally) return -1;
  if (h->mpc_tally_T != h->T || h->T < 1 || !h->ws[tsat_handle::WS_MPC_TALLY]) return fail(h, -1, "tsat_mpc_run has not been called on this batch");
  TSAT_HIP(h, hipSetDevice(h->dev));
  static_assert(sizeof(long long) == sizeof(int64_t), "tally element");
  TSAT_HIP(h, hipMemcpy(tally, h->ws[tsat_handle::WS_MPC_TALLY], (size_t)h->T * 4 * sizeof(int64_t), hipMemcpyDeviceToHost));
  return 0;
}

int tsat_batch_export_device(tsat_handle* h, void* X_dev, void* U_dev, void* K_dev, void* stats_dev) {
  if (!h) return -1;
  if (!h->solved) return fail(h, -1, "tsat_batch_run has not been called");
  TSAT_HIP(h, hipSetDevice(h->dev));
  const int64_t n_rec = h->T * (int64_t)h->N;
  if (X_dev || U_dev || K_dev) {
    const unsigned blocks = (unsigned)((n_rec + 255) / 256);
    hipLaunchKernelGGL(tsat_export_kernel<double>, dim3(blocks), dim3(256), 0, h->stream, n_rec, h->N,
                       h->ragged ? h->nk : nullptr, h->XU, h->KD,
                       (double*)X_dev, (double*)U_dev, (double*)K_dev);
    TSAT_HIP(h, hipGetLastError());
  }
  if (stats_dev)
    TSAT_HIP(h, hipMemcpyAsync(stats_dev, h->stats, (size_t)h->T * sizeof(tsat_stats), hipMemcpyDeviceToDevice, h->stream));
  TSAT_HIP(h, hipStreamSynchronize(h->stream));
  return 0;
}

int tsat_batch_download(tsat_handle* h, double* X, double* U, double* K, tsat_stats* stats) {
  if (!h) return -1;
  if (!h->solved) return fail(h, -1, "tsat_batch_run has not been called");
  TSAT_HIP(h, hipSetDevice(h->dev));
  const size_t T = (size_t)h->T, N = (size_t)h->N;
  double *dX = nullptr, *dU = nullptr, *dK = nullptr;
  int rc = 0;
  if (X && !(dX = (double*)ws_get(h, tsat_handle::WS_DL_X, T * N * 7 * sizeof(double)))) rc = -10;
  if (!rc && U && !(dU = (double*)ws_get(h, tsat_handle::WS_DL_U, T * (N - 1) * 3 * sizeof(double)))) rc = -10;
  if (!rc && K && !(dK = (double*)ws_get(h, tsat_handle::WS_DL_K, T * (N - 1) * 21 * sizeof(double)))) rc = -10;
  if (!rc) rc = tsat_batch_export_device(h, dX, dU, dK, nullptr);
  if (!rc && X && hipMemcpy(X, dX, T * N * 7 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && U && hipMemcpy(U, dU, T * (N - 1) * 3 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && K && hipMemcpy(K, dK, T * (N - 1) * 21 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && stats && hipMemcpy(stats, h->stats, T * sizeof(tsat_stats), hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (rc == -10) h->err = "device allocation or copy failed in tsat_batch_download";
  return rc;
}

int tsat_batch_trace_download(tsat_handle* h, double* trace) {
  if (!h || !trace) return -1;
  if (!h->trace) return fail(h, -1, "tracing is disabled");
  TSAT_HIP(h, hipSetDevice(h->dev));
  TSAT_HIP(h, hipMemcpy(trace, h->trace, (size_t)h->T * h->trace_rows * 8 * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int tsat_solve_batch(tsat_handle* h, const tsat_options* o, int64_t T, int64_t n_btab, const double* x0,
                     const double* xf, const double* Btab, const int32_t* btab_idx, const double* tau0,
                     const double* dtau, const double* dt, const double* Jmat, const double* Qd,
                     const double* Qfd, const double* Rd, const double* ulo, const double* uhi, const double* U0,
                     double* X, double* U, double* K, tsat_stats* stats) {
  if (!h || !o) return -1;
  int rc = tsat_batch_reserve(h, T, o->n_knots, o->n_tab, n_btab, o->max_linesearch);
  if (rc) return rc;
  rc = tsat_batch_upload(h, x0, xf, Btab, btab_idx, tau0, dtau, dt, Jmat, Qd, Qfd, Rd, ulo, uhi, U0);
  if (rc) return rc;
  rc = tsat_batch_run(h, o, nullptr);
  if (rc) return rc;
  return tsat_batch_download(h, X, U, K, stats);
}

void tsat_tvlqr_default_options(tsat_tvlqr_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->linearize_dt_sq = 1;   // `dt = S[end]^2`, src/attitude_controller.jl:137
  o->min_steps = 10; o->u_scale = 1e-2; o->w_tol = 0.05; o->angle_tol = 0.08727;   // src/monte_carlo.jl:70-71,251
  tv_noise_defaults(*o);
}

namespace {
// launch + read-back shared by tsat_tvlqr_batch (everything uploaded for the call) and tsat_tvlqr_resident (reference
// records, tables and knot counts are the resident solved batch)
int run_tvlqr(tsat_handle* h, const tsat_tvlqr_options* o, int64_t T, int N, int n_tab, int cls, const double* dP,
              const double* dBT, const int* dbi, const int* dnk, const double* dXUR, const double* noise,
              const int64_t* noise_id, double* X_sim, double* U_sim, double* K_lqr, tsat_tvlqr_stats* stats) {
  const size_t Tn = (size_t)T;
  const size_t nNZ = Tn * (size_t)(N - 1) * 36, nKD = Tn * (size_t)(N - 1) * KDW, nXS = Tn * N * XUW;
  double *dNZ = nullptr, *dKD = nullptr, *dXS = nullptr;
  long long* dnid = nullptr;
  tsat_tvlqr_stats* dst = nullptr;
  int rc = 0;
  auto A = [&](void** p, int slot, size_t bytes) { if (!rc && !(*p = ws_get(h, slot, bytes))) rc = -10; };
  auto C = [&](void* d, const void* s, size_t bytes) { if (!rc && hipMemcpy(d, s, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = -10; };
  if (noise) { A((void**)&dNZ, tsat_handle::WS_TV_NZ, nNZ * 8); C(dNZ, noise, nNZ * 8); }
  A((void**)&dKD, tsat_handle::WS_TV_KD, nKD * 8); A((void**)&dXS, tsat_handle::WS_TV_XS, nXS * 8);
  A((void**)&dst, tsat_handle::WS_TV_ST, Tn * sizeof(tsat_tvlqr_stats));
  if ((o->noise_mode == 1 || o->rate_as_written) && noise_id) { A((void**)&dnid, tsat_handle::WS_TV_NID, Tn * sizeof(long long)); C(dnid, noise_id, Tn * sizeof(long long)); }
  // ragged batch: the slabs beyond a trajectory's own horizon stay zero
  if (dnk && !rc && (hipMemsetAsync(dKD, 0, nKD * 8, h->stream) != hipSuccess || hipMemsetAsync(dXS, 0, nXS * 8, h->stream) != hipSuccess)) rc = -10;
  if (!rc) {
    TvArgs<double> a;
    a.T = (int)T; a.N = N; a.n_tab = n_tab; a.lin_sq = o->linearize_dt_sq; a.min_steps = o->min_steps;
    a.us = o->u_scale; a.w_tol = o->w_tol; a.ang_tol = o->angle_tol;
    fill_tv_noise<double>(*o, dnid, a);
    a.P = dP; a.BT = dBT; a.bidx = dbi; a.nk = dnk; a.XUR = dXUR; a.NZ = dNZ; a.KD = dKD; a.XS = dXS; a.stats = dst;
    auto kern = cls == 2 ? tsat_tvlqr_kernel<double, 2> : (cls == 1 ? tsat_tvlqr_kernel<double, 1> : tsat_tvlqr_kernel<double, 0>);
    hipLaunchKernelGGL(kern, dim3((unsigned)T), dim3(64), 0, h->stream, a);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) rc = -10;
  }
  const bool want_traj = X_sim || U_sim || K_lqr;       // the statistic alone travels when no trajectory is asked for
  std::vector<double> XS(want_traj ? nXS : 0), KD(K_lqr ? nKD : 0);     // the gains travel only when asked for
  if (!rc && want_traj && hipMemcpy(XS.data(), dXS, nXS * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && K_lqr && hipMemcpy(KD.data(), dKD, nKD * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && hipMemcpy(stats, dst, Tn * sizeof(tsat_tvlqr_stats), hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && want_traj) unpack_tv<double>(T, N, XS.data(), KD.data(), X_sim, U_sim, K_lqr);
  return rc;
}
}  // namespace

int tsat_tvlqr_batch(tsat_handle* h, const tsat_tvlqr_options* o, int64_t T, int64_t n_btab, const double* X,
                     const double* U, const double* xf, const double* Btab, const int32_t* btab_idx,
                     const double* tau0, const double* dtau, const double* dt, const double* Jmat, const double* Qd,
                     const double* Qfd, const double* Rd, const double* x0_sim, const double* noise, double* X_sim,
                     double* U_sim, double* K_lqr, tsat_tvlqr_stats* stats, const int32_t* n_knots, const int64_t* noise_id) {
  if (!h || !o) return -1;
  const std::string why = check_tv_options(*o);
  if (!why.empty()) return fail(h, -1, why);
  if (T < 1 || n_btab < 1) return fail(h, -1, "bad batch dimensions");
  if (!X || !U || !xf || !Btab || !tau0 || !dtau || !dt || !Jmat || !Qd || !Qfd || !Rd || !x0_sim || !stats)
    return fail(h, -1, "null array");
  if (!btab_idx && n_btab != T) return fail(h, -1, "btab_idx is NULL but n_btab != T");
  if (o->noise_mode == 1 && noise) return fail(h, -1, "noise_mode = 1 draws the noise in the kernel: pass noise = NULL");
  TSAT_HIP(h, hipSetDevice(h->dev));
  const int N = o->n_knots, n_tab = o->n_tab;
  const size_t Tn = (size_t)T;
  std::vector<int> bi(Tn);
  for (int64_t t = 0; t < T; ++t) {
    const int64_t v = btab_idx ? btab_idx[t] : t;
    if (v < 0 || v >= n_btab) return fail(h, -1, "btab_idx out of range");
    if (!(dt[t] > 0.0)) return fail(h, -1, "dt must be positive");
    if (n_knots && (n_knots[t] < 2 || n_knots[t] > N)) return fail(h, -1, "n_knots[t] must be in [2, N]");
    bi[(size_t)t] = (int)v;
  }
  std::vector<double> P(Tn * PSTRIDE), BT((size_t)n_btab * n_tab * 4), XUR(Tn * N * XUW);
  pack_tv_params<double>(T, x0_sim, xf, tau0, dtau, dt, Jmat, Qd, Qfd, Rd, P.data());
  pack_btab<double>(n_btab, n_tab, Btab, BT.data());
  pack_xu_records<double>(T, N, X, U, XUR.data());
  double *dP = nullptr, *dBT = nullptr, *dXUR = nullptr;
  int *dbi = nullptr, *dnk = nullptr;
  int rc = 0;
  auto A = [&](void** p, int slot, size_t bytes) { if (!rc && !(*p = ws_get(h, slot, bytes))) rc = -10; };
  auto C = [&](void* d, const void* s, size_t bytes) { if (!rc && hipMemcpy(d, s, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = -10; };
  A((void**)&dP, tsat_handle::WS_TVB_P, P.size() * 8); A((void**)&dBT, tsat_handle::WS_TVB_BT, BT.size() * 8);
  A((void**)&dXUR, tsat_handle::WS_TVB_XUR, XUR.size() * 8); A((void**)&dbi, tsat_handle::WS_TVB_BI, Tn * sizeof(int));
  C(dP, P.data(), P.size() * 8); C(dBT, BT.data(), BT.size() * 8); C(dXUR, XUR.data(), XUR.size() * 8); C(dbi, bi.data(), Tn * sizeof(int));
  if (n_knots) { A((void**)&dnk, tsat_handle::WS_TVB_NK, Tn * sizeof(int)); C(dnk, n_knots, Tn * sizeof(int)); }
  if (!rc) rc = run_tvlqr(h, o, T, N, n_tab, inertia_class(T, Jmat), dP, dBT, dbi, dnk, dXUR, noise, noise_id, X_sim, U_sim, K_lqr, stats);
  if (rc) h->err = "device allocation, copy or launch failed in tsat_tvlqr_batch";
  return rc;
}

int tsat_tvlqr_resident(tsat_handle* h, const tsat_tvlqr_options* o, const double* Qd, const double* Qfd, const double* Rd,
                        const double* x0_sim, const double* noise, double* X_sim, double* U_sim, double* K_lqr,
                        tsat_tvlqr_stats* stats, const int64_t* noise_id) {
  if (!h || !o) return -1;
  if (!h->solved) return fail(h, -1, "tsat_batch_run has not been called");
  tsat_tvlqr_options oo = *o;
  oo.n_knots = h->N; oo.n_tab = h->n_tab;
  const std::string why = check_tv_options(oo);
  if (!why.empty()) return fail(h, -1, why);
  if (!Qd || !Qfd || !Rd || !x0_sim || !stats) return fail(h, -1, "null array");
  if (oo.noise_mode == 1 && noise) return fail(h, -1, "noise_mode = 1 draws the noise in the kernel: pass noise = NULL");
  TSAT_HIP(h, hipSetDevice(h->dev));
  const int64_t T = h->T;
  std::vector<double> P((size_t)T * PSTRIDE);
  pack_tv_params<double>(T, x0_sim, h->hxf.data(), h->htau0.data(), h->hdtau.data(), h->hdt.data(), h->hJ.data(), Qd, Qfd, Rd, P.data());
  double* dP = (double*)ws_get(h, tsat_handle::WS_TV_P, P.size() * 8);
  if (!dP) return fail(h, -10, "device allocation failed in tsat_tvlqr_resident");
  int rc = hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice) == hipSuccess ? 0 : -10;
  const double* dXUR = h->XU;
  if (!rc) rc = run_tvlqr(h, &oo, T, h->N, h->n_tab, h->inertia_class, dP, h->BT, h->bidx, h->ragged ? h->nk : nullptr, dXUR,
                          noise, noise_id, X_sim, U_sim, K_lqr, stats);
  if (rc) h->err = "device allocation, copy or launch failed in tsat_tvlqr_resident";
  return rc;
}

int tsat_horizon_batch(tsat_handle* h, int64_t T, int32_t n_rows, const double* Btab, const double* dt_row,
                       const double* cutoff, int32_t* tf_index, double* cond_at) {
  if (!h) return -1;
  if (T < 1 || n_rows < 1) return fail(h, -1, "bad dimensions");
  if (!dt_row || !cutoff || !tf_index) return fail(h, -1, "null array");
  if (!Btab && (h->bt_T != T || h->bt_rows != n_rows))
    return fail(h, -1, "Btab is NULL but the field tables left by the last tsat_btable_batch are not T x n_rows");
  TSAT_HIP(h, hipSetDevice(h->dev));
  const size_t Tn = (size_t)T, nB = Tn * (size_t)n_rows * 3;
  double *dB = nullptr, *ddt = nullptr, *dcut = nullptr, *dc = nullptr;
  int* di = nullptr;
  int rc = 0;
  auto A = [&](void** p, int slot, size_t bytes) { if (!rc && !(*p = ws_get(h, slot, bytes))) rc = -10; };
  if (Btab) A((void**)&dB, tsat_handle::WS_HZ_B, nB * 8);
  else dB = (double*)ws_get(h, tsat_handle::WS_BT_B, nB * 8);       // same size as allocated: the resident tables themselves
  A((void**)&ddt, tsat_handle::WS_HZ_DT, Tn * 8); A((void**)&dcut, tsat_handle::WS_HZ_CUT, Tn * 8);
  A((void**)&dc, tsat_handle::WS_HZ_C, Tn * 8); A((void**)&di, tsat_handle::WS_HZ_I, Tn * sizeof(int));
  auto Cp = [&](void* d, const void* s, size_t bytes) { if (!rc && hipMemcpy(d, s, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = -10; };
  if (Btab) Cp(dB, Btab, nB * 8);
  Cp(ddt, dt_row, Tn * 8); Cp(dcut, cutoff, Tn * 8);
  if (!rc) {
    HzArgs<double> a;
    a.T = (int)T; a.n_rows = n_rows; a.BT = dB; a.dt_row = ddt; a.cutoff = dcut; a.tf_index = di; a.cond_at = dc;
    hipLaunchKernelGGL(tsat_horizon_kernel<double>, dim3((unsigned)T), dim3(64), 0, h->stream, a);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) rc = -10;
  }
  if (!rc && hipMemcpy(tf_index, di, Tn * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && cond_at && hipMemcpy(cond_at, dc, Tn * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (rc) h->err = "device allocation, copy or launch failed in tsat_horizon_batch";
  return rc;
}

// ------------------------------------------------------------------------------------------------
// Sweep exchange: the RCCL all-gather of every rank's converged trajectories (north_star). RCCL is loaded on
// first use — a single-GPU host never needs it — and called on the handle's own stream right behind the export kernel.
// ------------------------------------------------------------------------------------------------
}  // extern "C"
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
RcclApi& rccl() {
  static RcclApi api;
  if (api.lib || !api.why.empty()) return api;
  // a host that already carries an RCCL (PyTorch-ROCm bundles one) keeps using that copy; otherwise the system library
  const char* env = std::getenv("TSAT_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)
    if (n && *n && (api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!api.lib)
    for (const char* n : names)
      if (n && *n && (api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!api.lib) { api.why = "librccl.so.1 not found (set TSAT_RCCL_LIB)"; return api; }
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
  api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) { api.why = "RCCL symbols missing"; api.lib = nullptr; }
  return api;
}
int rccl_fail(tsat_handle* h, const char* what, ncclResult_t r) {
  RcclApi& a = rccl();
  return fail(h, -11, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error"));
}
}  // namespace
extern "C" {

int tsat_comm_available(void) { return rccl().lib ? 0 : -11; }

int tsat_comm_unique_id(void* id_out) {
  if (!id_out) return -1;
  RcclApi& a = rccl();
  if (!a.lib) return -11;
  static_assert(sizeof(ncclUniqueId) == TSAT_COMM_ID_BYTES, "unique-id size");
  ncclUniqueId id;
  if (a.GetUniqueId(&id) != ncclSuccess) return -11;
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

int tsat_comm_init(tsat_handle* h, const void* id_in, int32_t rank, int32_t world) {
  if (!h || !id_in) return -1;
  if (world < 1 || rank < 0 || rank >= world) return fail(h, -1, "need 0 <= rank < world");
  RcclApi& a = rccl();
  if (!a.lib) return fail(h, -11, "RCCL unavailable: " + a.why);
  TSAT_HIP(h, hipSetDevice(h->dev));
  (void)tsat_comm_destroy(h);
  ncclUniqueId id;
  std::memcpy(&id, id_in, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = a.CommInitRank(&c, world, id, rank);
  if (r != ncclSuccess) return rccl_fail(h, "ncclCommInitRank", r);
  h->comm = c; h->comm_rank = rank; h->comm_world = world;
  h->comm_T = -1; h->comm_N = -1;
  return 0;
}

int tsat_comm_destroy(tsat_handle* h) {
  if (!h) return -1;
  if (h->comm) {
    (void)hipSetDevice(h->dev);
    (void)hipStreamSynchronize(h->stream);
    (void)rccl().CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr; h->comm_rank = 0; h->comm_world = 1;
    h->comm_T = -1; h->comm_N = -1;
  }
  return 0;
}

int tsat_sweep_allgather(tsat_handle* h, void* X_all, void* U_all, void* stats_all, int32_t on_device) {
  if (!h) return -1;
  if (!h->solved) return fail(h, -1, "tsat_batch_run has not been called");
  if (!h->comm) return fail(h, -1, "tsat_comm_init has not been called");
  TSAT_HIP(h, hipSetDevice(h->dev));
  RcclApi& a = rccl();
  const size_t T = (size_t)h->T, N = (size_t)h->N, W = (size_t)h->comm_world;
  const size_t nX = T * N * 7, nU = T * (N - 1) * 3, nS = T * sizeof(tsat_stats);
  // Every rank must hold a shard of the same shape (T, N): the receive buffers are sized world x this rank's shard and the
  // collective sends this rank's count. Checked by a 16-byte all-gather of (T, N) whenever the shape is new to the communicator;
  // every rank sees the same gathered list, so a mismatch fails on all of them alike instead of corrupting memory on some.
  if (h->comm_T != h->T || h->comm_N != h->N) {
    int64_t* chk = (int64_t*)ws_get(h, tsat_handle::WS_AG_CHK, (W + 1) * 2 * sizeof(int64_t));
    if (!chk) return fail(h, -10, "device allocation failed in tsat_sweep_allgather");
    const int64_t mine[2] = {h->T, (int64_t)h->N};
    std::vector<int64_t> all(2 * W);
    TSAT_HIP(h, hipMemcpyAsync(chk + 2 * W, mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
    const ncclResult_t rc0 = a.AllGather(chk + 2 * W, chk, 2, ncclInt64, (ncclComm_t)h->comm, h->stream);
    if (rc0 != ncclSuccess) return rccl_fail(h, "ncclAllGather (shard shapes)", rc0);
    TSAT_HIP(h, hipMemcpyAsync(all.data(), chk, 2 * W * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    TSAT_HIP(h, hipStreamSynchronize(h->stream));
    for (size_t r = 0; r < W; ++r)
      if (all[2 * r] != h->T || all[2 * r + 1] != h->N)
        return fail(h, -1, "tsat_sweep_allgather: rank " + std::to_string(r) + " holds a " + std::to_string(all[2 * r]) + " x " +
                               std::to_string(all[2 * r + 1]) + " shard, this rank " + std::to_string(h->T) + " x " + std::to_string(h->N) +
                               ": shards must have equal shapes (pad the last one)");
    h->comm_T = h->T; h->comm_N = h->N;
  }
  // this rank's shard in the ABI layout (export kernel, device to device), then one ncclAllGather per array on the same
  // stream; with host outputs the gathered arrays land in library workspaces and are copied down afterwards
  double* dX = X_all ? (double*)ws_get(h, tsat_handle::WS_AG_X, nX * 8) : nullptr;
  double* dU = U_all ? (double*)ws_get(h, tsat_handle::WS_AG_U, nU * 8) : nullptr;
  void* dS = stats_all ? ws_get(h, tsat_handle::WS_AG_ST, nS) : nullptr;
  if ((X_all && !dX) || (U_all && !dU) || (stats_all && !dS)) return fail(h, -10, "device allocation failed in tsat_sweep_allgather");
  void *gX = X_all, *gU = U_all, *gS = stats_all;
  if (!on_device) {
    gX = X_all ? ws_get(h, tsat_handle::WS_AG_XA, W * nX * 8) : nullptr;
    gU = U_all ? ws_get(h, tsat_handle::WS_AG_UA, W * nU * 8) : nullptr;
    gS = stats_all ? ws_get(h, tsat_handle::WS_AG_STA, W * nS) : nullptr;
    if ((X_all && !gX) || (U_all && !gU) || (stats_all && !gS)) return fail(h, -10, "device allocation failed in tsat_sweep_allgather");
  }
  const int64_t n_rec = h->T * (int64_t)h->N;
  if (dX || dU) {
    hipLaunchKernelGGL(tsat_export_kernel<double>, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, h->stream, n_rec, h->N,
                       h->ragged ? h->nk : nullptr, h->XU, h->KD, dX, dU, (double*)nullptr);
    TSAT_HIP(h, hipGetLastError());
  }
  if (dS) TSAT_HIP(h, hipMemcpyAsync(dS, h->stats, nS, hipMemcpyDeviceToDevice, h->stream));
  ncclResult_t r = ncclSuccess;
  if (dS && r == ncclSuccess) r = a.AllGather(dS, gS, nS, ncclUint8, (ncclComm_t)h->comm, h->stream);
  if (dX && r == ncclSuccess) r = a.AllGather(dX, gX, nX, ncclFloat64, (ncclComm_t)h->comm, h->stream);
  if (dU && r == ncclSuccess) r = a.AllGather(dU, gU, nU, ncclFloat64, (ncclComm_t)h->comm, h->stream);
  if (r != ncclSuccess) return rccl_fail(h, "ncclAllGather", r);
  if (!on_device) {
    if (gX) TSAT_HIP(h, hipMemcpyAsync(X_all, gX, W * nX * 8, hipMemcpyDeviceToHost, h->stream));
    if (gU) TSAT_HIP(h, hipMemcpyAsync(U_all, gU, W * nU * 8, hipMemcpyDeviceToHost, h->stream));
    if (gS) TSAT_HIP(h, hipMemcpyAsync(stats_all, gS, W * nS, hipMemcpyDeviceToHost, h->stream));
  }
  TSAT_HIP(h, hipStreamSynchronize(h->stream));
  return 0;
}

void tsat_btable_default_options(tsat_btable_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->n_half = 5000;                          // src/TortoiseSat.jl:61
  o->mjd = 58155.0;                          // src/TortoiseSat.jl:44
  o->gm = 3.986004418e14 * 1e-9;             // km^3/s^2, src/input_parameters.jl:26
  o->r_igrf_km = 400.0 + 6371.0;             // alt + R_E, src/magnetic_toolbox.jl:44,81
  o->date = 2019.0;                          // src/magnetic_toolbox.jl:81
}

// Host-side script arithmetic, batched: Bryson weights of the versine eigen-axis guess for T slews that differ only in their
// horizon (src/eigen_axis_slew.jl:1-38 + src/monte_carlo.jl:161-176 in the loop body of the Monte-Carlo script). No GPU work.
int tsat_bryson_eigen_axis_batch(int64_t T, const int32_t* n_knots, double t0, double dt, double theta_f, const double* axis,
                                 const double* Jrm, double alpha, double beta, double* Qd, double* Qfd, double* Rd) {
  if (T < 1 || !n_knots || !axis || !Jrm || !Qd || !Qfd || !Rd || !(dt > 0)) return -1;
  for (int64_t t = 0; t < T; ++t)
    if (n_knots[t] < 3) return -1;
  std::atomic<int> bad{0};
  auto work = [&](int64_t lo, int64_t hi) {
    std::vector<double> d;
    for (int64_t j = lo; j < hi; ++j) {
      const int n = n_knots[j];
      const double a = 3.14159265358979323846 / (t0 + dt * (double)(n - 1));        // alpha = pi / t[end]
      const double h = (t0 + dt * 1.0) - (t0 + dt * 0.0);                             // t[2] - t[1]
      d.resize((size_t)n - 1);
      double th0 = theta_f * 0.5 * (1.0 - std::cos(a * (t0 + dt * 0.0)));
      for (int k = 0; k < n - 1; ++k) {                                              // d_theta = diff(theta) / (t[2] - t[1])
        const double th1 = theta_f * 0.5 * (1.0 - std::cos(a * (t0 + dt * (double)(k + 1))));
        d[(size_t)k] = (th1 - th0) / h;
        th0 = th1;
      }
      double wmax = 0.0, taumax = -1.0 / 0.0;
      for (int k = 0; k < n - 1; ++k)
        for (int c = 0; c < 3; ++c) wmax = std::max(wmax, std::fabs(d[(size_t)k] * axis[c]));
      // w has n rows (the last rate repeated): diff(w) has n - 1 rows, the last one zero; tau = J dw / dt, signed maximum
      for (int k = 0; k < n - 1; ++k) {
        double dw[3];
        for (int c = 0; c < 3; ++c) dw[c] = (k < n - 2) ? d[(size_t)k + 1] * axis[c] - d[(size_t)k] * axis[c] : 0.0;
        for (int r = 0; r < 3; ++r) taumax = std::max(taumax, ((Jrm[3 * r] * dw[0] + Jrm[3 * r + 1] * dw[1]) + Jrm[3 * r + 2] * dw[2]) / dt);
      }
      if (!(wmax > 0.0) || !(taumax > 0.0)) { bad = 1; continue; }
      const double mmax = taumax / 1.0e-5 * 1.0e2;
      for (int i = 0; i < 7; ++i) { Qd[7 * j + i] = (i < 3) ? alpha / (wmax * wmax) : alpha * beta; Qfd[7 * j + i] = 10.0 * Qd[7 * j + i]; }
      for (int c = 0; c < 3; ++c) Rd[3 * j + c] = 1.0 / (mmax * mmax);
    }
  };
  const int nth = (int)std::min<int64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())), (T + 63) / 64);
  std::vector<std::thread> th;
  for (int i = 0; i < nth; ++i) th.emplace_back(work, T * i / nth, T * (i + 1) / nth);
  for (auto& x : th) x.join();
  return bad ? -2 : 0;
}

int tsat_btable_batch(tsat_handle* h, const tsat_btable_options* o, int64_t T, const double* kep, const double* t0,
                      const double* tf, double* Btab, double* pos) {
  if (!h || !o) return -1;
  // whatever happens below, the tables of an earlier call are no longer "the last call's": a consumer with Btab = NULL must
  // never pick up stale ones by shape alone (a failed call leaves none)
  h->bt_T = 0; h->bt_rows = 0; h->bt_gen++;
  if (T < 1 || o->n_half < 1) return fail(h, -1, "bad dimensions");
  if (!(o->date >= 2015.0 && o->date < 2020.0)) return fail(h, -1, "date must be in [2015, 2020): IGRF-12 epoch 2015 + secular variation");
  if (!kep || !t0 || !tf) return fail(h, -1, "null array");
  for (int64_t t = 0; t < T; ++t)
    if (!(kep[6 * t] >= 0.0 && kep[6 * t] < 1.0) || !(kep[6 * t + 1] > 0.0) || !(tf[t] > t0[t]))
      return fail(h, -1, "need 0 <= e < 1, a > 0 and tf > t0");
  TSAT_HIP(h, hipSetDevice(h->dev));
  const int N = o->n_half;
  const size_t Tn = (size_t)T, nP = Tn * 3 * (2 * (size_t)N + 1), nB = Tn * 3 * 2 * (size_t)N;
  std::vector<double> coef;
  igrf_records(o->date, o->r_igrf_km, coef);
  double *dc = nullptr, *dk = nullptr, *d0 = nullptr, *d1 = nullptr, *dP = nullptr, *dB = nullptr;
  int rc = 0;
  auto A = [&](void** p, int slot, size_t bytes) { if (!rc && !(*p = ws_get(h, slot, bytes))) rc = -10; };
  A((void**)&dc, tsat_handle::WS_BT_COEF, coef.size() * 8); A((void**)&dk, tsat_handle::WS_BT_KEP, Tn * 6 * 8);
  A((void**)&d0, tsat_handle::WS_BT_T0, Tn * 8); A((void**)&d1, tsat_handle::WS_BT_TF, Tn * 8);
  A((void**)&dP, tsat_handle::WS_BT_POS, nP * 8); A((void**)&dB, tsat_handle::WS_BT_B, nB * 8);
  auto Cp = [&](void* d, const void* s, size_t bytes) { if (!rc && hipMemcpy(d, s, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = -10; };
  Cp(dc, coef.data(), coef.size() * 8); Cp(dk, kep, Tn * 6 * 8); Cp(d0, t0, Tn * 8); Cp(d1, tf, Tn * 8);
  if (!rc) {
    BtArgs<double> a;
    a.T = (int)T; a.n_half = N; a.mjd = o->mjd; a.gm = o->gm; a.r_igrf_km = o->r_igrf_km;
    a.tab = dc; a.kep = dk; a.t0 = d0; a.tf = d1; a.pos = dP; a.B = dB;
    hipLaunchKernelGGL(tsat_btable_kernel<double>, dim3((unsigned)T), dim3(64), 0, h->stream, a);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) rc = -10;
  }
  if (!rc && Btab && hipMemcpy(Btab, dB, nB * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (!rc && pos && hipMemcpy(pos, dP, nP * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -10;
  if (rc) h->err = "device allocation, copy or launch failed in tsat_btable_batch";
  h->bt_T = rc ? 0 : T; h->bt_rows = rc ? 0 : 2 * N;     // the tables stay resident for tsat_horizon_batch / tsat_batch_upload
  return rc;
}

}  // extern "C"
