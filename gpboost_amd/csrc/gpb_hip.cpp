// gpboost_amd/csrc/gpb_hip.cpp -- host side of the C ABI declared in include/gpb_hip.h.
//
// Owns device memory, streams and launch sequencing for the gfx950 kernels; contains no
// numerical code of its own except the libstdc++-dependent argsort the reference performs on
// the host as well (include/GPBoost/utils.h:230-238).  There is deliberately no CPU fallback:
// without a gfx950 device every entry point fails with a message.
#include "../../include/gpb_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cerrno>
#include <vector>

#include "dense_kernels.h"
#include "hist_kernels.h"
#include "leaf_kernels.h"
#include "nn_kernels.h"
#include "vecchia_kernels.h"
#include "vif_kernels.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

thread_local char g_err[512] = "everything is fine";

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// GPB_HIP_API_TIMING=1: every entry point of this ABI accumulates its wall time and call count (inclusive: an entry point that calls another one
// counts both); the table goes to stderr at process exit and on gpb_hip_api_timing_report.  The integration's counterpart of the reference's TIMETAG
// build (include/LightGBM/utils/common.h:989-1068): it separates the time the reference's host code spends INSIDE this library from the time it spends
// in its own code around the seams.  Off (the default): one predictable branch per call.
// GPB_HIP_API_TIMING=2 (round 6): additionally the TIMELINE of the calls since the last report -- name, duration and the gap since the previous call returned, i.e. the
// time the caller's own code ran between two seams (route B: the reference's O(n) host passes of a boosting iteration; nested calls are listed where they return).
struct ApiTiming {
  struct Row { const char* name; double seconds; long calls; };
  struct Ev { const char* name; double t0, t1; };
  std::mutex mu;
  std::vector<Row> rows;
  std::vector<Ev> timeline;
  bool on, line;
  std::chrono::steady_clock::time_point epoch = std::chrono::steady_clock::now();
  ApiTiming() { const char* e = std::getenv("GPB_HIP_API_TIMING"); on = e && e[0] && e[0] != '0'; line = on && e[0] == '2'; }
  ~ApiTiming() { if (on) report(stderr, true); }
  void add(const char* name, double s, std::chrono::steady_clock::time_point t0) {
    std::lock_guard<std::mutex> lk(mu);
    if (line && timeline.size() < 200000) { const double a = std::chrono::duration<double>(t0 - epoch).count(); timeline.push_back(Ev{ name, a, a + s }); }
    for (auto& r : rows) if (r.name == name || std::strcmp(r.name, name) == 0) { r.seconds += s; ++r.calls; return; }
    rows.push_back(Row{ name, s, 1 });
  }
  void report(FILE* f, bool reset) {
    std::lock_guard<std::mutex> lk(mu);
    std::vector<Row> v = rows;
    std::sort(v.begin(), v.end(), [](const Row& a, const Row& b) { return a.seconds > b.seconds; });
    std::fprintf(f, "[gpb_hip api timing] %-52s %10s %12s %12s\n", "entry point (inclusive)", "calls", "total ms", "ms / call");
    for (const auto& r : v) std::fprintf(f, "[gpb_hip api timing] %-52s %10ld %12.3f %12.4f\n", r.name, r.calls, 1e3 * r.seconds, 1e3 * r.seconds / r.calls);
    if (line && !timeline.empty()) {
      std::fprintf(f, "[gpb_hip api timeline] %-48s %12s %12s %14s\n", "call (in order of return)", "start ms", "inside ms", "caller ms before");
      double prev_end = timeline.front().t0;
      const size_t first = timeline.size() > 400 ? timeline.size() - 400 : 0;     // the last 400 calls
      for (size_t i = 0; i < timeline.size(); ++i) {
        const Ev& e = timeline[i];
        const double gap = e.t0 > prev_end ? e.t0 - prev_end : 0.0;               // (a nested call starts before its parent returns: no gap)
        if (i >= first) std::fprintf(f, "[gpb_hip api timeline] %-48s %12.3f %12.3f %14.3f\n", e.name, 1e3 * (e.t0 - timeline.front().t0), 1e3 * (e.t1 - e.t0), 1e3 * gap);
        if (e.t1 > prev_end) prev_end = e.t1;
      }
    }
    if (reset) { rows.clear(); timeline.clear(); }
  }
};
ApiTiming g_api_timing;
struct ApiTimer {
  const char* name;
  std::chrono::steady_clock::time_point t0;
  explicit ApiTimer(const char* n) : name(g_api_timing.on ? n : nullptr) { if (name) t0 = std::chrono::steady_clock::now(); }
  ~ApiTimer() { if (name) g_api_timing.add(name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), t0); }
};

#define API_BEGIN() ApiTimer api_timer_(__func__); try {
#define API_END()                                                        \
  }                                                                      \
  catch (const std::exception& ex) { return fail("%s", ex.what()); }     \
  catch (...) { return fail("unknown exception"); }                      \
  return 0;

// runs f on every way out of a scope (error returns included): temporaries of a call, a half-built handle
template <class F>
struct ScopeExit { F f; ~ScopeExit() { f(); } };
template <class F>
ScopeExit<F> scope_exit(F f) { return ScopeExit<F>{ f }; }

template <class T>
void dev_free(T*& p) {
  if (p) { (void)hipFree(p); p = nullptr; }
}

int check_device() {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0)
    return fail("no HIP device available (%s): the gfx950 hot path has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  return 0;
}

std::vector<double> exp_table() {
  std::vector<double> t(GPB_EXP_TAB_SIZE);
  for (int j = 0; j < GPB_EXP_TAB_SIZE; ++j) t[j] = std::exp2((double)j / (double)GPB_EXP_TAB_SIZE);
  return t;
}

}  // namespace


// ------------------------------------------------------------------------------------------
// Communicator of a sharded handle.  Two transports behind one all-reduce:
//   * RCCL (one process per GPU, ncclAllReduce on the handle's stream over xGMI): the production form;
//   * an IN-PROCESS group: the ranks are threads of one process and their handles may share one device (gpb_hip_local_group_create).
//     Every rank publishes its buffer, a barrier, every rank reduces all published buffers IN RANK ORDER into its own scratch, a
//     barrier, scratch -> buffer.  Same results on every rank, bit for bit, for every type.  This is how the sharded code paths
//     (neighbour search parts, likelihood terms, y_aux, histograms, the data-parallel tree grower) run with SEVERAL ranks on the one
//     MI355X a test box has: same host code, same kernels, only the transport differs.
#define NCCL_OK(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) return fail("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

enum GpbType { GPB_T_I32 = 0, GPB_T_I64 = 1, GPB_T_U64 = 2, GPB_T_F64 = 3 };
enum GpbOp { GPB_OP_SUM = 0, GPB_OP_MAX = 1 };
constexpr int kLocalGroupMaxWorld = 16;

struct gpb_hip_local_group {
  int world = 1;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long generation = 0;
  const void* bufs[kLocalGroupMaxWorld] = {nullptr};
  bool broken = false;                 // a rank gave up (error elsewhere): nobody waits for ever
  // false when the barrier was abandoned (gpb_hip_local_group_abort or 120 s without the peers)
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const unsigned long long g = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
    const bool ok = cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != g || broken; });
    if (!ok) { broken = true; cv.notify_all(); }
    return ok && !broken;
  }
};

struct LocalBufs { const void* p[kLocalGroupMaxWorld]; };

template <class T, int OP>
__global__ __launch_bounds__(256) void local_allreduce_kernel(LocalBufs b, int world, T* __restrict__ out, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    T acc = static_cast<const T*>(b.p[0])[i];
    for (int r = 1; r < world; ++r) {
      const T v = static_cast<const T*>(b.p[r])[i];
      if (OP == GPB_OP_SUM) acc = acc + v; else acc = v > acc ? v : acc;
    }
    out[i] = acc;
  }
}

struct GpbComm {
  ncclComm_t nccl = nullptr;
  gpb_hip_local_group* lg = nullptr;
  int rank = 0, world = 1;
  void* tmp = nullptr; size_t tmp_cap = 0;
  bool active() const { return nccl != nullptr || lg != nullptr; }
  void release() {
    if (nccl) { (void)ncclCommDestroy(nccl); nccl = nullptr; }
    lg = nullptr;
    if (tmp) { (void)hipFree(tmp); tmp = nullptr; tmp_cap = 0; }
  }
};

static int comm_init_rccl(GpbComm& c, const unsigned char* id128, int rank, int world) {
  c.release();
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  NCCL_OK(ncclCommInitRank(&c.nccl, world, id, rank));
  c.rank = rank; c.world = world;
  return 0;
}

static int comm_init_local(GpbComm& c, gpb_hip_local_group* g, int rank) {
  if (!g) return fail("null group");
  if (rank < 0 || rank >= g->world) return fail("in-process group: rank %d of %d", rank, g->world);
  c.release();
  c.lg = g; c.rank = rank; c.world = g->world;
  return 0;
}

// in place, on `st`; the caller's later work on `st` sees the result
static int comm_allreduce(GpbComm& c, void* buf, size_t count, GpbType type, GpbOp op, hipStream_t st) {
  if (!c.active()) return fail("no communicator on this handle");
  if (c.nccl) {
    static const ncclDataType_t t[4] = {ncclInt32, ncclInt64, ncclUint64, ncclDouble};
    NCCL_OK(ncclAllReduce(buf, buf, count, t[type], op == GPB_OP_SUM ? ncclSum : ncclMax, c.nccl, st));
    return 0;
  }
  gpb_hip_local_group* g = c.lg;
  const size_t bytes = count * (type == GPB_T_I32 ? 4 : 8);
  if (c.tmp_cap < bytes) { if (c.tmp) (void)hipFree(c.tmp); c.tmp = nullptr; c.tmp_cap = 0; HIP_OK(hipMalloc(&c.tmp, bytes)); c.tmp_cap = bytes; }
  HIP_OK(hipStreamSynchronize(st));                           // this rank's contribution is complete
  { std::lock_guard<std::mutex> lk(g->mu); g->bufs[c.rank] = buf; }
  if (!g->barrier()) return fail("in-process group: a peer rank did not arrive at the all-reduce");
  LocalBufs b;
  { std::lock_guard<std::mutex> lk(g->mu); for (int r = 0; r < g->world; ++r) b.p[r] = g->bufs[r]; }
  const dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((count + 255) / 256, 4096))), block(256);
#define GPB_LOCAL_AR(T)                                                                                                       \
  do {                                                                                                                        \
    if (op == GPB_OP_SUM) hipLaunchKernelGGL((local_allreduce_kernel<T, GPB_OP_SUM>), grid, block, 0, st, b, g->world, (T*)c.tmp, count); \
    else hipLaunchKernelGGL((local_allreduce_kernel<T, GPB_OP_MAX>), grid, block, 0, st, b, g->world, (T*)c.tmp, count);      \
  } while (0)
  switch (type) {
    case GPB_T_I32: GPB_LOCAL_AR(int); break;
    case GPB_T_I64: GPB_LOCAL_AR(long long); break;
    case GPB_T_U64: GPB_LOCAL_AR(unsigned long long); break;
    default: GPB_LOCAL_AR(double); break;
  }
#undef GPB_LOCAL_AR
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(st));                           // every peer's buffer has been read by this rank ...
  if (!g->barrier()) return fail("in-process group: a peer rank did not finish the all-reduce");   // ... and this rank's by every peer
  HIP_OK(hipMemcpyAsync(buf, c.tmp, bytes, hipMemcpyDeviceToDevice, st));
  return 0;
}

// reduce-scatter(sum, int64): rank r receives the sum over the ranks of send[r * count .. (r + 1) * count) in recv (count values); on `st`
static int comm_reducescatter_i64(GpbComm& c, const long long* send, long long* recv, size_t count, hipStream_t st) {
  if (!c.active()) return fail("no communicator on this handle");
  if (c.nccl) {
    NCCL_OK(ncclReduceScatter(send, recv, count, ncclInt64, ncclSum, c.nccl, st));
    return 0;
  }
  gpb_hip_local_group* g = c.lg;
  HIP_OK(hipStreamSynchronize(st));                           // this rank's contribution is complete
  { std::lock_guard<std::mutex> lk(g->mu); g->bufs[c.rank] = send; }
  if (!g->barrier()) return fail("in-process group: a peer rank did not arrive at the reduce-scatter");
  LocalBufs b;
  { std::lock_guard<std::mutex> lk(g->mu); for (int r = 0; r < g->world; ++r) b.p[r] = static_cast<const long long*>(g->bufs[r]) + (size_t)c.rank * count; }
  const dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((count + 255) / 256, 4096))), block(256);
  hipLaunchKernelGGL((local_allreduce_kernel<long long, GPB_OP_SUM>), grid, block, 0, st, b, g->world, recv, count);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(st));                           // every peer's buffer has been read by this rank ...
  if (!g->barrier()) return fail("in-process group: a peer rank did not finish the reduce-scatter");   // ... and this rank's by every peer
  return 0;
}

// ------------------------------------------------------------------------------------------
// Node-local MAILBOX for the 3 / 7 sums of a sharded likelihood evaluation (round 4).  ncclAllReduce of 24 / 56 bytes
// costs ~24 us with ONE rank in the loop (launch of the collective kernel + the publish kernel behind it) -- the largest term outside the shard
// kernel at N = 8.  The sums do not need a device collective at all: every rank's finisher workgroup already stores them
// straight into pinned host memory with system-scope stores ("the data is the flag", vecchia_kernels.hip: vecchia_finish) and its host polls
// for them.  The mailbox makes that pinned buffer a POSIX shared-memory segment mapped by every rank of the node: rank r's GPU writes slot r,
// every host polls all `world` slots and adds them IN RANK ORDER -- the same bits on every rank, no collective launch, no second kernel, no
// peer access; the cross-rank step is the hosts' cache coherence.  Layout: header {magic, world, ready[world]}, then [3 generations][world][8]
// doubles.  Generation g = evaluation count % 3; at the start of evaluation e a rank re-arms ITS OWN slot of generation (e + 1) % 3 with the
// sentinel: no rank can still be reading that generation (it was read in evaluation e - 2, and a rank only starts evaluation e after every
// rank's result of e - 1 has arrived, i.e. after every rank finished e - 2).  RCCL stays for the big messages (y_aux, histograms, tables).
struct GpbMailbox {
  int world = 0, rank = -1;
  void* base = nullptr; size_t bytes = 0;
  double* dev_base = nullptr;                 // device address of the mapping (hipHostGetDevicePointer)
  unsigned long long evals = 0;
  bool dead = false;                          // an evaluation failed or timed out on this rank: the generation counters of the ranks may be out of step, nothing more goes through it
  char name[64] = "";
  // poll limits (seconds): GPB_MAILBOX_TIMEOUT_S for an evaluation's sums (default 30), GPB_MAILBOX_ATTACH_TIMEOUT_S for the attach rendezvous (default 120)
  static double env_seconds(const char* var, double dflt) {
    const char* v = std::getenv(var);
    if (!v || !v[0]) return dflt;
    const double x = std::atof(v);
    return x > 0.0 ? x : dflt;
  }
  static constexpr unsigned long long kMagic = 0x4750424d41494c42ull;   // "GPBMAILB"
  static size_t header_bytes(int w) { return ((16 + 8 * (size_t)w) + 63) / 64 * 64; }
  static size_t total_bytes(int w) { return header_bytes(w) + sizeof(double) * 3 * (size_t)w * 8; }
  bool active() const { return base != nullptr; }
  volatile unsigned long long* slot(int gen, int r) const {
    return reinterpret_cast<volatile unsigned long long*>(static_cast<char*>(base) + header_bytes(world)) + ((size_t)gen * world + r) * 8;
  }
  double* dev_slot(int gen, int r) const {
    return reinterpret_cast<double*>(reinterpret_cast<char*>(dev_base) + header_bytes(world)) + ((size_t)gen * world + r) * 8;
  }
  void release() {
    if (base) { (void)hipHostUnregister(base); (void)munmap(base, bytes); }
    base = nullptr; dev_base = nullptr; world = 0; rank = -1; bytes = 0; evals = 0; dead = false;
  }
};

struct LaplaceState;                                   // gpb_laplace.inc (Vecchia-Laplace workspace, row a13)
static void laplace_state_free(LaplaceState* s);

struct gpb_hip_vecchia {
  double* d_batch = nullptr; size_t batch_cap = 0;   // gpb_hip_vecchia_nll_terms_batch: 3 K shard sums
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = true;
  int n = 0, d = 0, m = 0;
  int i_begin = 0, i_end = 0;
  double4* d_pts = nullptr;
  double* d_coords_nd = nullptr;   // d > 3: [n][d] coordinates (Vecchia order); the records then only carry the response
  int* d_nn = nullptr;
  double* d_exp_tab = nullptr;
  double* d_partials = nullptr;
  double* d_out = nullptr;   // GPB_NUM_PARTIALS + 1
  double* h_out = nullptr;   // pinned, coherent: written by reduce_partials_kernel, polled by vecchia_fetch
  double* h_red = nullptr;   // pinned, coherent: the all-reduced terms, published by a one-wavefront kernel after ncclAllReduce and polled by the host
  int launches_unfetched = 0; // reductions enqueued since the last vecchia_fetch (polling is only unambiguous for exactly one)
  double* d_A = nullptr; double* d_D = nullptr; double* d_u = nullptr; double* d_v = nullptr; double* d_w = nullptr;
  double* d_ystage = nullptr;
  double* d_X = nullptr; double* d_U = nullptr; double* d_G = nullptr; double* d_beta = nullptr; int p_cov = 0;   // linear-regression covariates (Vecchia order)
  int* d_tptr = nullptr; int* d_tpos = nullptr;
  int* d_flag = nullptr;
  bool has_nn = false, has_y = false, has_factor = false, has_transpose = false, has_levels = false, has_yaux = false, nn_partial = false;
  bool u_stale = false;           // A, D belong to the current parameters but u = B y to an earlier response (a new y arrived: refresh_u renews it, no refactorisation)
  // spatially sorted gather (round 5): rank of every point in Morton order, the neighbour table rewritten into the sorted copy d_pts[n ..], built at the third
  // launch on a neighbour table (a prediction's temporary handle launches once), refreshed when the response in the records changed
  int* d_rank = nullptr; int* d_nn2 = nullptr; bool sorted_ready = false, gpts_dirty = false; int launches_on_table = 0;
  int sorted_mode = -1;           // gpb_hip_vecchia_set_sorted_gather: -1 as described, 0 never, 1 from the first launch on whatever n
  // a MODE_FACTOR launch into other buffers with observation-specific diagonal additions (the "vecchia_response" preconditioner's factor of W^-1 + Sigma,
  // gpb_laplace.inc pc_refresh): when set, vecchia_launch writes A / D / u there and takes the per-point diagonal entries var + nug[.] from it
  struct FactorOverride { double* A; double* D; double* u; const double* nug; };
  const FactorOverride* factor_override = nullptr;
  double* d_nug = nullptr;        // sample weights (Gaussian likelihood): observation-specific nugget 1 / w_i, Vecchia order (gpb_hip_vecchia_set_nugget_diag)
  // full-scale Vecchia (VIF): k inducing points [k][3]; row-major [n][kq] matrices (vif_kernels.hip): cross-covariances C (column k: the response),
  // whitened V, Q = B C; k x k matrices of the products [6][kq][kq]; Gram tiles; per-point partial sums [12][n]; the gradient's matrices are
  // allocated at its first use
  int vif_k = 0, vif_kp = 0, vif_kq = 0; double* d_ip = nullptr; double* d_V = nullptr; double* d_vif_part = nullptr;
  double* d_vC = nullptr; double* d_vQ = nullptr; double* d_vM = nullptr; double* d_vG = nullptr; double* d_vgpart = nullptr; double* d_vout = nullptr;
  double* d_vdC = nullptr; double* d_vQdC = nullptr; double* d_vX1 = nullptr; double* d_vV1 = nullptr; double* d_vX2 = nullptr; double* d_vHm = nullptr;
  double* d_vw = nullptr; double* d_vv = nullptr; double* d_vz = nullptr; double* d_vdA = nullptr; double* d_vdD = nullptr;
  bool vif_has_grad_inputs = false, vif_has_grad_factor = false;
  std::vector<double> vif_ip_host;      // the inducing points [k][3] on the host (the k x k matrices of the VIF-Laplace path are built there, gpb_laplace.inc)
  int* d_leaf = nullptr; double* d_leaf_part = nullptr; double* d_leaf_out = nullptr; size_t leaf_part_cap = 0;
  LaplaceState* lap = nullptr;
  GpbComm comm;                   // optional: in-library all-reduce of the partial terms (gpb_hip_vecchia_comm_init / _comm_init_local)
  // optional in-loop timing of the dominant kernel (gpb_hip_vecchia_timing): a ring of HIP event pairs recorded around every point-kernel launch on
  // the handle's stream, so that the kernel time reported next to a timed loop is measured INSIDE that loop
  bool timing_on = false; unsigned long long timing_count = 0; std::vector<hipEvent_t> timing_ev;
  GpbMailbox mbox;                // optional: node-local shared-memory mailbox for the 3 / 7 sums of a sharded evaluation (gpb_hip_vecchia_mailbox_attach)
  double* d_red = nullptr;        // 8 doubles: all-reduce buffer in the caller-facing term order
  std::vector<double> coords;   // host copy, column-major n x d (for the neighbour search set-up)
  std::vector<int> nn_host;
};

static void vif_free(gpb_hip_vecchia* h) {
  dev_free(h->d_ip); dev_free(h->d_V); dev_free(h->d_vif_part); dev_free(h->d_vC); dev_free(h->d_vQ); dev_free(h->d_vM); dev_free(h->d_vG); dev_free(h->d_vgpart);
  dev_free(h->d_vout); dev_free(h->d_vdC); dev_free(h->d_vQdC); dev_free(h->d_vX1); dev_free(h->d_vV1); dev_free(h->d_vX2); dev_free(h->d_vHm); dev_free(h->d_vw);
  dev_free(h->d_vv); dev_free(h->d_vz); dev_free(h->d_vdA); dev_free(h->d_vdD);
  h->vif_has_grad_inputs = h->vif_has_grad_factor = false;
}

struct gpb_hip_exact {
  int device = 0;
  hipStream_t stream = nullptr;
  int n = 0, d = 0, np = 0;
  double4* d_pts = nullptr;
  double* d_P = nullptr; double* d_y = nullptr; double* d_z = nullptr; double* d_x = nullptr; double* d_out = nullptr;
  double* d_work = nullptr;        // np doubles: the running right-hand side of the blocked triangular solves
  double* d_exp_tab = nullptr;
  int* d_info = nullptr;
  bool has_y = false;
  hipStream_t stream2 = nullptr; hipEvent_t ev_panels = nullptr, ev_rest = nullptr;   // look-ahead of the blocked Cholesky
  double* d_P2 = nullptr;       // gradient: augmented matrix [[Psi, .], [I, 0]], (2 np)^2, allocated on first use
  double* d_gpart = nullptr;    // gradient: [4][tiles] partial sums
  double* d_g4 = nullptr;       // gradient: the four sums
};

static const unsigned long long kFetchSentinel = 0x7ff8dead0000beefull;   // a NaN payload: "not written yet" in the pinned result buffer

struct gpb_hip_hist {
  int device = 0;
  hipStream_t stream = nullptr;
  int n = 0, F = 0, fpad = 0, total_bins = 0;
  int num_cu = 0;                                          // compute units of the device (chunking of the build kernel)
  // regularisation of the split search beyond lambda_l2 (gpb_hip_hist_set_regularisation); parent_output: of the next single-leaf searches
  double reg_l1 = 0., reg_max_delta_step = 0., reg_path_smooth = 0., reg_parent_output = 0.;
  int max_depth = 0;                                       // depth limit of gpb_hip_hist_grow_tree (<= 0: none)
  std::vector<signed char> feature_mask;                   // columns the next trees may split on (empty: all)
  int* d_root_rows = nullptr; int root_cnt = 0;            // bagging: the rows of the root of the next trees (root_cnt = 0: all rows)
  uint8_t* d_bins_rm = nullptr;
  uint8_t* d_bins_cm = nullptr; int rstride = 0;   // compact copy [n][rstride] for the streaming root pass (hist_kernels.h: HistKernelArgs::bins_cm); absent when F % 16 == 0
  int* d_bin_offsets = nullptr;
  double* d_grad = nullptr; double* d_hess = nullptr;
  bool has_hess = false, has_grad = false;
  int* d_idx = nullptr; int idx_cap = 0;
  long long* d_part_grad = nullptr; long long* d_part_hess = nullptr; uint32_t* d_part_cnt = nullptr; int part_chunks = 0;
  unsigned long long* d_absmax = nullptr;                  // bits of max |grad|, max |hess|: the scale of the fixed-point histogram sums
  double* d_hist = nullptr; unsigned long long* d_cnt = nullptr;
  GpbComm comm;                                            // optional: data-parallel histogram all-reduce (rows sharded per rank)
  long long* d_limbs = nullptr;                            // sharded handles: integer totals [5][total_bins] {grad hi, grad lo, count, hess hi, hess lo} between reduce, all-reduce and conversion
  double* d_pool = nullptr; int nslots = 0;                 // resident leaf histograms (HistogramPool), 2 * total_bins doubles each
  int* d_fix = nullptr; bool has_fix = false;              // view_offset[F], num_bin[F], most_freq_bin[F]
  int* d_meta3 = nullptr; bool has_split_info = false;     // per feature: FeatureMetainfo::offset, default_bin, missing_type
  std::vector<int> h_fix, h_meta3, h_bin_offsets;          // host copies for the scalar arguments of the partition kernel
  int* d_part = nullptr; int part_cap = 0;                 // partition workspace: block counts / offsets, lte, gt
  double* d_split = nullptr; int* d_split_i = nullptr; signed char* d_used = nullptr;   // split search outputs: F x 10, F + 1 ints
  // gpb_hip_hist_grow_tree: resident row lists of the leaves, two sets of search outputs (device + pinned host)
  unsigned long long* d_ptags = nullptr; unsigned part_epoch = 0; int tree_seq = 0;   // one-pass partition granules / epochs; sequence numbers of the polled splits
  int* d_rows2 = nullptr; int* d_counts = nullptr; int* h_counts = nullptr;     // second (ping-pong) row buffer; {left rows of this rank, of all ranks} of the current split
  int* d_rows = nullptr; double* d_split2 = nullptr; int* d_split2_i = nullptr; signed char* d_used2 = nullptr;
  double* h_split2 = nullptr; int* h_split2_i = nullptr;
  double* d_tree_red = nullptr;                            // 4 doubles: root sums / left count of the data-parallel tree grower
  // caller buffers seen by set_gradients / grow_tree (the Booster hands over the SAME gradient / hessian / leaf-index arrays every
  // iteration): page-locked once with hipHostRegister, so that the per-tree copies run at the PCIe rate instead of through the
  // runtime's staging buffer (8 MB at n = 1e6: ~2.5 ms pageable, ~0.35 ms registered); unregistered when the pointer changes / at free
  struct Pinned { const void* p = nullptr; size_t bytes = 0; };
  Pinned pin[3];
  std::vector<double> tree_node_info;                      // last tree of gpb_hip_hist_grow_tree: per node {left / right output, count, sum of hessians}
  // categorical features (round 5; gpb_hip_hist_set_categorical): flags per feature, the search's configuration, bitsets over bins of the last searches
  std::vector<signed char> h_is_cat; signed char* d_is_cat = nullptr; bool any_cat = false;
  gpb::CatCfg cat;
  unsigned* d_cat_bits = nullptr;                          // [F][8]: gpb_hip_hist_find_best_split
  unsigned* d_cat_bits2 = nullptr; unsigned* h_cat_bits2 = nullptr;     // [2][F][8]: the tree grower's two candidate sets (device, pinned host)
  // feature-block exchange of the data-parallel tree grower (round 5): contiguous blocks of features, one per rank; ONE reduce-scatter of the integer totals per
  // leaf, every rank converts and searches its own block, the ranks' best candidates are exchanged (gpb_tree.inc)
  int block_exchange = -1;                                 // gpb_hip_hist_set_feature_block_exchange: 1 on, 0 off (every rank all-reduces and searches everything), -1 by message size
  std::vector<int> blk_f0, blk_bin0, blk_bins; int blk_max = 0;   // per rank: first feature (world + 1 entries), first bin, number of bins; the common padded length
  long long* d_rs_send = nullptr; long long* d_rs_recv = nullptr; double* d_xchg = nullptr; double* h_xchg = nullptr;
  std::vector<int> tree_node_is_cat; std::vector<unsigned> tree_node_cat_bits;   // last tree: per node, is the split categorical / the 8 words of its set of bins
};

extern "C" {

const char* gpb_hip_get_last_error(void) { return g_err; }

int gpb_hip_api_timing_report(int reset) {
  if (!g_api_timing.on) return fail("gpb_hip_api_timing_report: GPB_HIP_API_TIMING is not set");
  g_api_timing.report(stderr, reset != 0);
  return 0;
}

int gpb_hip_api_mark(const char* name) {
  if (g_api_timing.line && name) g_api_timing.add(name, 0.0, std::chrono::steady_clock::now());
  return 0;
}

int gpb_hip_device_count(int* count) {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess) { cnt = 0; (void)hipGetLastError(); }
  *count = cnt;
  return 0;
}

int gpb_hip_set_device(int device) {
  API_BEGIN();
  if (check_device()) return -1;
  HIP_OK(hipSetDevice(device));
  API_END();
}

/* Route B seam switch: while set, the reference's find_nearest_neighbors_Vecchia_fast (src/GPBoost/Vecchia_utils.cpp, patched)
   hands the ordered neighbour search to the device.  Thread-local: REModelTemplate's constructor sets it around CreateREComponentsVecchia. */
static thread_local int g_route_b_device_search = 0;
int gpb_hip_route_b_set_device_search(int on) { g_route_b_device_search = on ? 1 : 0; return 0; }
int gpb_hip_route_b_get_device_search(void) { return g_route_b_device_search; }

int gpb_hip_device_is_gfx950(int* yes) {
  API_BEGIN();
  if (check_device()) return -1;
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, dev));
  *yes = (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : 0;
  API_END();
}

int gpb_hip_dev_alloc(uint64_t bytes, void** out) {
  API_BEGIN();
  if (!out) return fail("null argument");
  if (check_device()) return -1;
  HIP_OK(hipMalloc(out, bytes ? bytes : 8));
  API_END();
}
int gpb_hip_dev_free(void* p) {
  API_BEGIN();
  if (p) HIP_OK(hipFree(p));
  API_END();
}
int gpb_hip_dev_to_host(void* dst_host, const void* src_dev, uint64_t bytes) {
  API_BEGIN();
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  API_END();
}

int gpb_hip_selftest(void) {
  API_BEGIN();
  if (check_device()) return -1;
  std::vector<double> in(192), out(256);
  for (int t = 0; t < 192; ++t) in[t] = std::sin(0.37 * t) + 1.5;
  double *d_in = nullptr, *d_out = nullptr;
  const auto free_tmp = scope_exit([&] { (void)hipFree(d_in); (void)hipFree(d_out); });
  HIP_OK(hipMalloc(&d_in, 192 * sizeof(double)));
  HIP_OK(hipMalloc(&d_out, 256 * sizeof(double)));
  HIP_OK(hipMemcpy(d_in, in.data(), 192 * sizeof(double), hipMemcpyHostToDevice));
  HIP_OK(gpb::launch_dpp_selftest(d_in, d_out, nullptr));
  HIP_OK(hipMemcpy(out.data(), d_out, 256 * sizeof(double), hipMemcpyDeviceToHost));
  for (int t = 0; t < 64; ++t) {
    const int row = t & ~15;
    const double eb = in[row + 5], ef = in[128 + t] - in[row + 11] * in[64 + t];
    if (out[4 * t] != eb || out[4 * t + 1] != eb)
      return fail("fp64 DPP row_newbcast self-test failed at lane %d: asm %.17g builtin %.17g expected %.17g", t, out[4 * t], out[4 * t + 1], eb);
    // asm and compiler-scheduled forms must agree bitwise; the host value is not fused, so it only bounds the result
    if (out[4 * t + 2] != out[4 * t + 3] || std::fabs(out[4 * t + 2] - ef) > 4e-16 * (std::fabs(in[128 + t]) + std::fabs(in[row + 11] * in[64 + t])))
      return fail("fp64 DPP fmac self-test failed at lane %d: asm %.17g builtin %.17g expected %.17g", t, out[4 * t + 2], out[4 * t + 3], ef);
  }
  API_END();
}

// ------------------------------------------------------------------------------------------
int gpb_hip_vecchia_create(int32_t n, int32_t d, int32_t num_neighbors, const double* coords_colmajor,
                           gpb_hip_vecchia_t** out) {
  API_BEGIN();
  if (!out) return fail("gpb_hip_vecchia_create: out is NULL");
  *out = nullptr;
  if (check_device()) return -1;
  if (n < 1) return fail("gpb_hip_vecchia_create: n = %d", n);
  if (d < 1 || d > GPB_MAX_DIM) return fail("gpb_hip_vecchia_create: coordinate dimension %d not supported by the HIP hot path (1..%d)", d, GPB_MAX_DIM);
  if (!coords_colmajor) return fail("gpb_hip_vecchia_create: coords is NULL");
  int m = num_neighbors;
  if (m > n - 1) m = n - 1;   // Vecchia_utils.cpp:755-758
  if (m < 0) m = 0;
  if (num_neighbors < 1 && n > 1) return fail("gpb_hip_vecchia_create: num_neighbors = %d", num_neighbors);
  if (m > GPB_MAX_NEIGHBORS_BIG) return fail("gpb_hip_vecchia_create: num_neighbors = %d exceeds the supported maximum %d", m, GPB_MAX_NEIGHBORS_BIG);
  auto* h = new gpb_hip_vecchia();
  bool built = false;
  const auto drop_half_built = scope_exit([&] { if (!built) gpb_hip_vecchia_free(h); });
  h->n = n; h->d = d; h->m = m < 1 ? 1 : m;   // n == 1: one (empty, -1) column keeps indexing uniform
  h->i_begin = 0; h->i_end = n;
  h->coords.assign(coords_colmajor, coords_colmajor + (size_t)n * d);
  HIP_OK(hipGetDevice(&h->device));
  HIP_OK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  std::vector<double4> pts(n);
  for (int i = 0; i < n; ++i) {
    pts[i].x = coords_colmajor[i];
    pts[i].y = d > 1 ? coords_colmajor[(size_t)n + i] : 0.0;
    pts[i].z = d > 2 ? coords_colmajor[(size_t)2 * n + i] : 0.0;
    pts[i].w = 0.0;
  }
  HIP_OK(hipMalloc(&h->d_pts, sizeof(double4) * 2 * (size_t)n));      // (second half: the spatially sorted copy the neighbour gathers read, sorted_gather_prepare)
  HIP_OK(hipMemcpy(h->d_pts, pts.data(), sizeof(double4) * (size_t)n, hipMemcpyHostToDevice));
  if (d > 3) {       // generality path: row-major coordinates for the LDS-resident point kernel and the d-dimensional neighbour search
    std::vector<double> rm((size_t)n * d);
    for (int i = 0; i < n; ++i) for (int c = 0; c < d; ++c) rm[(size_t)i * d + c] = coords_colmajor[(size_t)c * n + i];
    HIP_OK(hipMalloc(&h->d_coords_nd, sizeof(double) * rm.size()));
    HIP_OK(hipMemcpy(h->d_coords_nd, rm.data(), sizeof(double) * rm.size(), hipMemcpyHostToDevice));
  }
  HIP_OK(hipMalloc(&h->d_nn, sizeof(int) * (size_t)n * h->m));
  const std::vector<double> tab = exp_table();
  HIP_OK(hipMalloc(&h->d_exp_tab, GPB_EXP_TAB_SIZE * sizeof(double)));
  HIP_OK(hipMemcpy(h->d_exp_tab, tab.data(), GPB_EXP_TAB_SIZE * sizeof(double), hipMemcpyHostToDevice));
  const int nblocks = (h->m > GPB_MAX_NEIGHBORS || d > 3) ? n : (n + 15) / 16;      // m > 62 or d > 3: the LDS-resident kernel, one workgroup per point
  HIP_OK(hipMalloc(&h->d_partials, sizeof(double) * (size_t)nblocks * GPB_NUM_PARTIALS));
  HIP_OK(hipMemset(h->d_partials, 0xFF, sizeof(double) * (size_t)nblocks * GPB_NUM_PARTIALS));   // every slot "empty" (vecchia_kernels.hip: the finisher workgroup)
  HIP_OK(hipMalloc(&h->d_out, sizeof(double) * 8));
  HIP_OK(hipHostMalloc(&h->h_out, sizeof(double) * 8, hipHostMallocCoherent));
  HIP_OK(hipHostMalloc(&h->h_red, sizeof(double) * 8, hipHostMallocCoherent));
  HIP_OK(hipMalloc(&h->d_flag, sizeof(int)));
  built = true;
  *out = h;
  API_END();
}

int gpb_hip_vecchia_free(gpb_hip_vecchia_t* h) {
  API_BEGIN();
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream || !h->owns_stream) { (void)hipStreamSynchronize(h->stream); if (h->owns_stream) (void)hipStreamDestroy(h->stream); }
  dev_free(h->d_pts); dev_free(h->d_coords_nd); dev_free(h->d_nn); dev_free(h->d_exp_tab); dev_free(h->d_partials); dev_free(h->d_out); dev_free(h->d_batch);
  dev_free(h->d_A); dev_free(h->d_D); dev_free(h->d_u); dev_free(h->d_v); dev_free(h->d_w); dev_free(h->d_ystage); dev_free(h->d_X); dev_free(h->d_U); dev_free(h->d_G); dev_free(h->d_beta);
  dev_free(h->d_tptr); dev_free(h->d_tpos); dev_free(h->d_flag); dev_free(h->d_red);
  dev_free(h->d_leaf); dev_free(h->d_leaf_part); dev_free(h->d_leaf_out);
  vif_free(h); dev_free(h->d_nug); dev_free(h->d_rank); dev_free(h->d_nn2);
  h->comm.release();
  h->mbox.release();
  for (hipEvent_t ev : h->timing_ev) (void)hipEventDestroy(ev);
  h->timing_ev.clear();
  laplace_state_free(h->lap); h->lap = nullptr;
  if (h->h_out) (void)hipHostFree(h->h_out);
  if (h->h_red) (void)hipHostFree(h->h_red);
  delete h;
  API_END();
}

int gpb_hip_vecchia_set_stream(gpb_hip_vecchia_t* h, void* hip_stream) {
  API_BEGIN();
  if (!h) return fail("null handle");
  HIP_OK(hipStreamSynchronize(h->stream));
  if (h->owns_stream && h->stream) (void)hipStreamDestroy(h->stream);
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  h->owns_stream = false;
  API_END();
}

int gpb_hip_vecchia_sync(gpb_hip_vecchia_t* h) {
  API_BEGIN();
  if (!h) return fail("null handle");
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

// Neighbour search for the positions (coordinate-sum order) [pos0, pos1); rows of other queries are left at INT_MIN when the
// range is not the whole table (multi-GPU: one block of positions per rank, merged by a max-all-reduce -- a valid entry is >= -1).
static int find_neighbors_impl(gpb_hip_vecchia_t* h, int part, int nparts, int* has_duplicates) {
  HIP_OK(hipSetDevice(h->device));
  const int n = h->n, d = h->d, m = h->m;
  // coordinate sums and their argsort, on the host exactly as Vecchia_utils.cpp:774-786 does
  std::vector<double> csum(n);
  for (int i = 0; i < n; ++i) {
    double s = h->coords[i];
    for (int c = 1; c < d; ++c) s += h->coords[(size_t)c * n + i];
    csum[i] = s;
  }
  std::vector<int> sort_sum(n);
  std::iota(sort_sum.begin(), sort_sum.end(), 0);
  const double* v = csum.data();
  std::sort(sort_sum.begin(), sort_sum.end(), [v](int i1, int i2) { return v[i1] < v[i2]; });
  std::vector<double4> rec(n);
  for (int k = 0; k < n; ++k) {
    const int i = sort_sum[k];
    rec[k].x = h->coords[i];
    rec[k].y = d > 1 ? h->coords[(size_t)n + i] : 0.0;
    rec[k].z = d > 2 ? h->coords[(size_t)2 * n + i] : 0.0;
    rec[k].w = csum[i];
  }
  double4* d_rec = nullptr; int* d_idx = nullptr; double* d_rec_nd = nullptr; int* d_qorder = nullptr;
  const auto free_tmp = scope_exit([&] { (void)hipFree(d_rec); (void)hipFree(d_idx); (void)hipFree(d_qorder); (void)hipFree(d_rec_nd); });
  HIP_OK(hipMalloc(&d_rec, sizeof(double4) * (size_t)n));
  HIP_OK(hipMalloc(&d_idx, sizeof(int) * (size_t)n));
  HIP_OK(hipMemcpyAsync(d_rec, rec.data(), sizeof(double4) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  std::vector<double> rec_nd;
  if (d > 3) {
    rec_nd.resize((size_t)n * (d + 1));
    for (int k = 0; k < n; ++k) {
      const int i = sort_sum[k];
      for (int c = 0; c < d; ++c) rec_nd[(size_t)k * (d + 1) + c] = h->coords[(size_t)c * n + i];
      rec_nd[(size_t)k * (d + 1) + d] = csum[i];
    }
    HIP_OK(hipMalloc(&d_rec_nd, sizeof(double) * rec_nd.size()));
    HIP_OK(hipMemcpyAsync(d_rec_nd, rec_nd.data(), sizeof(double) * rec_nd.size(), hipMemcpyHostToDevice, h->stream));
  }
  HIP_OK(hipMemcpyAsync(d_idx, sort_sum.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
  gpb::NNKernelArgs a;
  a.sorted_rec = d_rec; a.sorted_idx = d_idx; a.pts = h->d_pts; a.nn = h->d_nn; a.has_duplicates = h->d_flag; a.n = n; a.m = m;
  a.sorted_nd = d_rec_nd; a.coords_nd = h->d_coords_nd;
  // positions are in coordinate-sum order, i.e. random with respect to the index that decides a query's cost: equal blocks of
  // positions are balanced (SURVEY.md 8e suggested cyclic assignment for index-ordered blocks; not needed here)
  a.start_at = 0; a.end_search_at = n - 2;
  a.pos0 = (int)((long long)n * part / nparts); a.pos1 = (int)((long long)n * (part + 1) / nparts);
  // the order in which the lanes take this block's queries: grouped by index octave quarter (nn_kernels.h)
  std::vector<int> qorder((size_t)std::max(a.pos1 - a.pos0, 1));
  int nq = 0;
  gpb::nn_query_order(sort_sum.data(), a.pos0, a.pos1, m, 0, qorder.data(), &nq);
  HIP_OK(hipMalloc(&d_qorder, sizeof(int) * qorder.size()));
  HIP_OK(hipMemcpyAsync(d_qorder, qorder.data(), sizeof(int) * (size_t)std::max(nq, 1), hipMemcpyHostToDevice, h->stream));
  a.qorder = d_qorder; a.nq = nq;
  if (nparts > 1) HIP_OK(hipMemsetAsync(h->d_nn, 0x80, sizeof(int) * (size_t)n * m, h->stream));     // 0x80808080 < -1: "not mine"
  if (n == 1) {
    HIP_OK(hipMemsetAsync(h->d_nn, 0xff, sizeof(int) * (size_t)n * m, h->stream));
  } else {
    HIP_OK(gpb::launch_vecchia_nn(d, a, h->stream));
  }
  int flag = 0;
  HIP_OK(hipMemcpyAsync(&flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (has_duplicates) *has_duplicates = flag;
  h->has_nn = nparts == 1; h->nn_partial = nparts > 1;
  h->has_transpose = false; h->has_levels = false; h->has_factor = false; h->nn_host.clear();
  return 0;
}

int gpb_hip_vecchia_find_neighbors(gpb_hip_vecchia_t* h, int* has_duplicates) {
  API_BEGIN();
  if (!h) return fail("null handle");
  if (find_neighbors_impl(h, 0, 1, has_duplicates)) return -1;
  API_END();
}

int gpb_hip_vecchia_find_neighbors_part(gpb_hip_vecchia_t* h, int32_t part, int32_t nparts, int* has_duplicates) {
  API_BEGIN();
  if (!h) return fail("null handle");
  if (nparts < 1 || part < 0 || part >= nparts) return fail("gpb_hip_vecchia_find_neighbors_part: part %d of %d", part, nparts);
  if (find_neighbors_impl(h, part, nparts, has_duplicates)) return -1;
  API_END();
}

int gpb_hip_vecchia_set_neighbors(gpb_hip_vecchia_t* h, const int32_t* nn) {
  API_BEGIN();
  if (!h || !nn) return fail("null argument");
  HIP_OK(hipSetDevice(h->device));
  const size_t cnt = (size_t)h->n * h->m;
  for (size_t t = 0; t < cnt; ++t) {
    const int i = (int)(t / h->m);
    if (nn[t] >= i || nn[t] < -1) return fail("gpb_hip_vecchia_set_neighbors: row %d has neighbour %d (must be < row index)", i, nn[t]);
  }
  HIP_OK(hipMemcpyAsync(h->d_nn, nn, sizeof(int) * cnt, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->nn_host.assign(nn, nn + cnt);
  h->has_nn = true; h->has_transpose = false; h->has_levels = false; h->has_factor = false;
  h->sorted_ready = false; h->launches_on_table = 0;      // a new neighbour table: the sorted gather is rebuilt for it
  API_END();
}

/* The neighbour gathers of the point kernel read a spatially sorted copy of the records (sorted_gather_prepare): mode -1 (default) = for n >= 32768 from the third
 * launch on a neighbour table, 0 = never, 1 = always.  The results do not depend on it by a bit (tests/test_vecchia_gpu.py). */
int gpb_hip_vecchia_set_sorted_gather(gpb_hip_vecchia_t* h, int mode) {
  API_BEGIN();
  if (!h) return fail("null handle");
  h->sorted_mode = mode < 0 ? -1 : (mode != 0 ? 1 : 0);
  if (h->sorted_mode == 0) h->sorted_ready = false;
  h->launches_on_table = 0;
  API_END();
}

int gpb_hip_vecchia_get_neighbors(gpb_hip_vecchia_t* h, int32_t* nn) {
  API_BEGIN();
  if (!h || !nn) return fail("null argument");
  if (!h->has_nn && !h->nn_partial) return fail("neighbours have not been determined");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipMemcpyAsync(nn, h->d_nn, sizeof(int) * (size_t)h->n * h->m, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

int gpb_hip_vecchia_set_shard(gpb_hip_vecchia_t* h, int32_t i_begin, int32_t i_end) {
  API_BEGIN();
  if (!h) return fail("null handle");
  if (i_begin < 0 || i_end > h->n || i_begin >= i_end) return fail("gpb_hip_vecchia_set_shard: invalid range [%d, %d) for n = %d", i_begin, i_end, h->n);
  h->i_begin = i_begin; h->i_end = i_end;
  h->has_factor = false; h->u_stale = false;
  API_END();
}

// A new response: the Vecchia factor (A_i, D_i) does not depend on it, only u = B y does -- the factor stays, u is renewed at its next use
// (refresh_u: one pass over A, no Cholesky).  The GPBoost algorithm sets a new response (F - y) every boosting iteration at unchanged
// parameters (CalcGradientF, re_model_template.h:3313-3316).  The full-scale factor carries the response inside Q, v, z: that one goes.
static void response_changed(gpb_hip_vecchia_t* h) {
  h->has_yaux = false;
  if (h->vif_k > 0) h->has_factor = false;
  else if (h->has_factor) h->u_stale = true;
}
static int refresh_u(gpb_hip_vecchia_t* h) {
  if (!h->u_stale) return 0;
  HIP_OK(gpb::launch_By_pts(h->d_A, h->d_nn, h->d_pts, h->m, h->i_begin, h->i_end, h->d_u, h->stream));
  h->u_stale = false;
  return 0;
}

int gpb_hip_vecchia_set_y_dev(gpb_hip_vecchia_t* h, const double* y_dev) {
  API_BEGIN();
  if (!h || !y_dev) return fail("null argument");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(gpb::launch_pack_y(h->d_pts, y_dev, h->n, h->stream));
  h->gpts_dirty = true;
  h->has_y = true; response_changed(h);
  API_END();
}

int gpb_hip_pinned_alloc(size_t bytes, void** out) {
  API_BEGIN();
  if (!out) return fail("null argument");
  *out = nullptr;
  HIP_OK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  API_END();
}

int gpb_hip_pinned_free(void* p) {
  API_BEGIN();
  if (p) HIP_OK(hipHostFree(p));
  API_END();
}

int gpb_hip_vecchia_set_y(gpb_hip_vecchia_t* h, const double* y_host) {
  API_BEGIN();
  if (!h || !y_host) return fail("null argument");
  HIP_OK(hipSetDevice(h->device));
  if (!h->d_ystage) HIP_OK(hipMalloc(&h->d_ystage, sizeof(double) * (size_t)h->n));
  HIP_OK(hipMemcpyAsync(h->d_ystage, y_host, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  HIP_OK(gpb::launch_pack_y(h->d_pts, h->d_ystage, h->n, h->stream));
  h->gpts_dirty = true;
  HIP_OK(hipStreamSynchronize(h->stream));   // y_host is borrowed for the call only
  h->has_y = true; response_changed(h);
  API_END();
}

// Spatially sorted gather (round 5; VERDICT r04 #9: FETCH + WRITE was 1.72x the algorithmic bytes).  The point kernel gathers m records of 32 bytes per point from
// positions that are random in Vecchia order -- one 64-byte sector fetched per record.  The neighbours of a point are near it in SPACE, so in a copy of the
// records sorted along a Morton curve they share sectors.  The copy lives behind the records themselves (d_pts[n .. 2 n)), the neighbour table is rewritten once
// to point into it; own records, outputs and everything else keep the Vecchia order, the arithmetic does not change by a bit.  Built at the third launch on a
// neighbour table (host: Morton keys + sort, ~0.1 s per million points), the copy is renewed by one scatter pass when the response in the records changed.
static int sorted_gather_prepare(gpb_hip_vecchia_t* h) {
  static const bool enabled = [] { const char* e = std::getenv("GPB_VECCHIA_SORTED_GATHER"); return !(e && e[0] == '0'); }();
  if (h->sorted_mode == 0 || (h->sorted_mode < 0 && (!enabled || h->n < 32768))) return 0;
  const int n = h->n;
  if (!h->sorted_ready) {
    if (++h->launches_on_table < 3 && h->sorted_mode < 0) return 0;
    std::vector<double4> pts((size_t)n);
    HIP_OK(hipMemcpyAsync(pts.data(), h->d_pts, sizeof(double4) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int k = 0; k < n; ++k) {
      const double c[3] = {pts[k].x, pts[k].y, pts[k].z};
      for (int q = 0; q < h->d; ++q) { lo[q] = std::min(lo[q], c[q]); hi[q] = std::max(hi[q], c[q]); }
    }
    const int bits = h->d == 3 ? 21 : (h->d == 2 ? 31 : 62);
    std::vector<std::pair<unsigned long long, int>> keys((size_t)n);
    for (int k = 0; k < n; ++k) {
      const double c[3] = {pts[k].x, pts[k].y, pts[k].z};
      unsigned long long q[3] = {0, 0, 0};
      for (int t = 0; t < h->d; ++t) {
        const double span = hi[t] - lo[t];
        const double u = span > 0. ? (c[t] - lo[t]) / span : 0.;
        q[t] = (unsigned long long)(std::min(std::max(u, 0.), 1. - 1e-16) * (double)(1ull << bits));
      }
      unsigned long long key = 0;
      for (int b = bits - 1; b >= 0; --b) for (int t = 0; t < h->d; ++t) key = (key << 1) | ((q[t] >> b) & 1ull);
      keys[k] = std::make_pair(key, k);
    }
    std::sort(keys.begin(), keys.end());
    std::vector<int> rank((size_t)n);
    for (int p = 0; p < n; ++p) rank[keys[p].second] = p;
    if (!h->d_rank) HIP_OK(hipMalloc(&h->d_rank, sizeof(int) * (size_t)n));
    if (!h->d_nn2) HIP_OK(hipMalloc(&h->d_nn2, sizeof(int) * (size_t)n * h->m));
    HIP_OK(hipMemcpyAsync(h->d_rank, rank.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    HIP_OK(gpb::launch_remap_nn(h->d_nn, h->d_rank, (size_t)n * h->m, n, h->d_nn2, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));      // (rank is a host temporary)
    h->sorted_ready = true; h->gpts_dirty = true;
  }
  if (h->gpts_dirty) {
    HIP_OK(gpb::launch_scatter_pts(h->d_pts, h->d_rank, n, h->stream));
    h->gpts_dirty = false;
  }
  return 0;
}

static int vecchia_launch(gpb_hip_vecchia_t* h, int mode, int cov_type, double var, double a, int gauss, double* out_dev,
                          int nout, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, double* host_slot_dev = nullptr) {
  if (!h) return fail("null handle");
  if (!h->has_nn) return fail("neighbours have not been determined (call gpb_hip_vecchia_find_neighbors / _set_neighbors)");
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  if (cov_type < 0 || cov_type > 2) return fail("covariance type %d is not on the HIP hot path (Matern 0.5/1.5/2.5 only)", cov_type);
  if (!(var > 0.) || !(a > 0.)) return fail("covariance parameters must be positive (var = %g, range = %g)", var, a);
  if (!(a > 1e-20) || !(a < 1e20) || !(var < 1e100)) return fail("covariance parameters out of the supported range (var = %g, transformed range = %g)", var, a);
  HIP_OK(hipSetDevice(h->device));
  gpb::VecchiaKernelArgs k;
  k.pts = h->d_pts; k.nn = h->d_nn; k.exp_tab = h->d_exp_tab; k.partials = h->d_partials;
  k.A = h->d_A; k.D = h->d_D; k.u = h->d_u;
  k.m = h->m; k.i_begin = h->i_begin; k.i_end = h->i_end;
  k.var = var; k.a = a;
  k.diag_nn = gauss ? var + 1.0 : var * (1.0 + 1e-10);   // Vecchia_utils.cpp:1599-1609
  k.diag_i = gauss ? var + 1.0 : var;                    // :1410-1417 + :1555-1563
  k.nugget = gauss ? 1.0 : 0.0;
  if (h->d_nug) {
    if (!gauss) return fail("observation-specific nuggets (sample weights) are for the Gaussian likelihood only");
    k.nug = h->d_nug;
  }
  if (h->factor_override) {
    if (mode != gpb::MODE_FACTOR) return fail("internal: a factor override is for MODE_FACTOR launches only");
    k.A = h->factor_override->A; k.D = h->factor_override->D; k.u = h->factor_override->u; k.nug = h->factor_override->nug;
  }
  const bool big = h->m > GPB_MAX_NEIGHBORS || h->d > 3; 
  if (!big && !k.nug && sorted_gather_prepare(h)) return -1;
  if (!big && !k.nug && h->sorted_ready) k.nn = h->d_nn2;             // (k.pts stays d_pts: own records in [0, n), the gathers' sorted copy in [n, 2 n))      // 62 < m <= 126 or 3 < d <= 10: LDS-resident generality kernel (vecchia_big_kernels.hip)
  k.coords_nd = h->d_coords_nd; k.dim = h->d;
  // the launch's sums go to d_out, to the caller's device buffer (documented order) and straight to the pinned host buffer: vecchia_fetch
  // needs no copy on the stream and can poll for them
  (void)nout;
  for (int t = 0; t < GPB_NUM_PARTIALS; ++t) reinterpret_cast<volatile unsigned long long*>(h->h_out)[t] = kFetchSentinel;
  ++h->launches_unfetched;
  if (!ev0 && h->timing_on) {
    const size_t slot = (size_t)(h->timing_count++ % (h->timing_ev.size() / 2));
    ev0 = h->timing_ev[2 * slot]; ev1 = h->timing_ev[2 * slot + 1];
  }
  if (ev0) HIP_OK(hipEventRecord(ev0, h->stream));
  if (big) {
    HIP_OK(gpb::launch_vecchia_point_big(mode, cov_type, h->d > 3 ? 0 : (h->d == 3 ? 3 : 2), k, h->stream));
    if (ev1) HIP_OK(hipEventRecord(ev1, h->stream));
    HIP_OK(gpb::launch_reduce_partials(h->d_partials, h->i_end - h->i_begin, mode == gpb::MODE_GRAD ? GPB_NUM_PARTIALS : 3, h->d_out, out_dev, h->stream,
                                       host_slot_dev ? host_slot_dev : h->h_out));
  } else {
    // ONE launch per evaluation: persistent worker workgroups + a finisher workgroup that adds up their sums (vecchia_kernels.hip)
    k.ngroups = (h->i_end - h->i_begin + 15) / 16;
    k.out = h->d_out; k.out_user = out_dev; k.out_host = host_slot_dev ? host_slot_dev : h->h_out;
    HIP_OK(gpb::launch_vecchia_point_kernel(mode, cov_type, h->d == 3, k, h->stream));
    if (ev1) HIP_OK(hipEventRecord(ev1, h->stream));
  }
  // every MODE_FACTOR launch rewrites u = B y from the response resident NOW (Gaussian factor, the Laplace seams' factor launches alike): the one
  // place where "u belongs to an earlier response" ends.  (refresh_u's u is the fma chain of vecchia_By_pts_kernel over the stored A -- equal to the
  // factor kernel's u to rounding, ~1e-16 relative, not bit for bit: stated in include/gpb_hip.h at gpb_hip_vecchia_set_y.)
  if (mode == gpb::MODE_FACTOR && !h->factor_override) h->u_stale = false;
  return 0;
}

// The sums of the LAST reduction: when it is the only one enqueued since the previous fetch, the host polls the pinned buffer for the
// kernel's own stores (every term starts as a NaN payload no arithmetic produces) -- the wake-up latency of hipStreamSynchronize
// (about 20 us of a 0.9 ms evaluation) is not paid; otherwise, or after 20 ms of polling, it synchronises the stream.
static int vecchia_fetch(gpb_hip_vecchia_t* h, double* out_host, int nout) {
  bool polled = false;
  if (h->launches_unfetched == 1) {
    const volatile unsigned long long* v = reinterpret_cast<const volatile unsigned long long*>(h->h_out);
    const int nterms = nout > 3 ? GPB_NUM_PARTIALS : 3;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
      bool all = true;
      for (int t = 0; t < nterms; ++t) all = all && v[t] != kFetchSentinel;
      if (all) { polled = true; break; }
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  h->launches_unfetched = 0;
  if (!polled) HIP_OK(hipStreamSynchronize(h->stream));          // reduce_partials_kernel has written h_out itself
  out_host[0] = h->h_out[gpb::GPB_P_QUAD];
  out_host[1] = h->h_out[gpb::GPB_P_LOGDET];
  for (int t = 2; t < nout; ++t) out_host[t] = h->h_out[t];
  return 0;
}

int gpb_hip_vecchia_nll_terms(gpb_hip_vecchia_t* h, int cov_type, double var, double a, int gauss_likelihood,
                              double* out3_host) {
  API_BEGIN();
  if (!out3_host) return fail("null output");
  if (vecchia_launch(h, gpb::MODE_NLL, cov_type, var, a, gauss_likelihood, nullptr, 0)) return -1;
  if (vecchia_fetch(h, out3_host, 3)) return -1;
  API_END();
}

int gpb_hip_vecchia_nll_terms_dev(gpb_hip_vecchia_t* h, int cov_type, double var, double a, int gauss_likelihood,
                                  double* out3_dev) {
  API_BEGIN();
  if (!out3_dev) return fail("null output");
  if (vecchia_launch(h, gpb::MODE_NLL, cov_type, var, a, gauss_likelihood, out3_dev, 3)) return -1;
  API_END();
}

// ---- in-library RCCL reduction of the per-shard terms (one process per GPU) ---------------------------------------
int gpb_hip_comm_get_unique_id(unsigned char* id128) {
  API_BEGIN();
  if (!id128) return fail("null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCL_OK(ncclGetUniqueId(&id));
  std::memcpy(id128, &id, 128);
  API_END();
}

int gpb_hip_vecchia_comm_init(gpb_hip_vecchia_t* h, const unsigned char* id128, int rank, int world) {
  API_BEGIN();
  if (!h || !id128) return fail("null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail("gpb_hip_vecchia_comm_init: rank %d / world %d", rank, world);
  HIP_OK(hipSetDevice(h->device));
  if (comm_init_rccl(h->comm, id128, rank, world)) return -1;
  if (!h->d_red) HIP_OK(hipMalloc(&h->d_red, sizeof(double) * 8));
  API_END();
}

/* In-process group (ranks = threads of one process, handles may share a device): see GpbComm above. */
int gpb_hip_local_group_create(int world, gpb_hip_local_group_t** out) {
  API_BEGIN();
  if (!out) return fail("null argument");
  if (world < 1 || world > kLocalGroupMaxWorld) return fail("gpb_hip_local_group_create: world = %d (1..%d)", world, kLocalGroupMaxWorld);
  auto* g = new gpb_hip_local_group();
  g->world = world;
  *out = g;
  API_END();
}
int gpb_hip_local_group_abort(gpb_hip_local_group_t* g) {
  API_BEGIN();
  if (!g) return 0;
  { std::lock_guard<std::mutex> lk(g->mu); g->broken = true; }
  g->cv.notify_all();
  API_END();
}
int gpb_hip_local_group_free(gpb_hip_local_group_t* g) {
  API_BEGIN();
  delete g;
  API_END();
}
int gpb_hip_vecchia_comm_init_local(gpb_hip_vecchia_t* h, gpb_hip_local_group_t* g, int rank) {
  API_BEGIN();
  if (!h) return fail("null argument");
  HIP_OK(hipSetDevice(h->device));
  if (comm_init_local(h->comm, g, rank)) return -1;
  if (!h->d_red) HIP_OK(hipMalloc(&h->d_red, sizeof(double) * 8));
  API_END();
}

int gpb_hip_vecchia_comm_info(gpb_hip_vecchia_t* h, int* rank, int* world) {
  API_BEGIN();
  if (!h) return fail("null argument");
  if (rank) *rank = 0;
  if (world) *world = 0;
  if (h->comm.nccl) {     // what RCCL itself says about the communicator, not what was passed to comm_init
    int r = 0, w = 0;
    NCCL_OK(ncclCommUserRank(h->comm.nccl, &r));
    NCCL_OK(ncclCommCount(h->comm.nccl, &w));
    if (rank) *rank = r;
    if (world) *world = w;
  } else if (h->comm.lg) {
    if (rank) *rank = h->comm.rank;
    if (world) *world = h->comm.world;
  } else if (h->mbox.active()) {     // a handle whose only transport is the node-local mailbox (the 3 / 7 sums of a sharded evaluation)
    if (rank) *rank = h->mbox.rank;
    if (world) *world = h->mbox.world;
  }
  API_END();
}

// ---- node-local mailbox (GpbMailbox above) ----------------------------------------------------------------------------------------
// rank 0: create the segment for `world` ranks; name_out: 64 bytes, the name every rank attaches to (hand it over like the ncclUniqueId)
int gpb_hip_mailbox_create(int world, char* name_out64) {
  API_BEGIN();
  if (!name_out64 || world < 1 || world > 64) return fail("gpb_hip_mailbox_create: invalid argument");
  char name[64];
  std::snprintf(name, sizeof(name), "/gpb_mbox_%d_%llx", (int)getpid(),
                (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return fail("gpb_hip_mailbox_create: shm_open(%s) failed: %s", name, std::strerror(errno));
  const size_t bytes = GpbMailbox::total_bytes(world);
  if (ftruncate(fd, (off_t)bytes) != 0) { (void)close(fd); (void)shm_unlink(name); return fail("gpb_hip_mailbox_create: ftruncate failed: %s", std::strerror(errno)); }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  (void)close(fd);
  if (p == MAP_FAILED) { (void)shm_unlink(name); return fail("gpb_hip_mailbox_create: mmap failed: %s", std::strerror(errno)); }
  auto* hdr = static_cast<volatile unsigned long long*>(p);      // fresh segments are zero-filled: ready[] = 0
  hdr[1] = (unsigned long long)world;
  std::atomic_thread_fence(std::memory_order_release);
  hdr[0] = GpbMailbox::kMagic;
  (void)munmap(p, bytes);
  std::memset(name_out64, 0, 64);
  std::strncpy(name_out64, name, 63);
  API_END();
}
// every rank (rank 0 included): map the segment, page-lock it for this rank's device, arm the own slots, wait until all ranks have done so;
// the name is unlinked once everybody is attached (the mappings keep the segment alive)
int gpb_hip_vecchia_mailbox_attach(gpb_hip_vecchia_t* h, const char* name, int rank, int world) {
  API_BEGIN();
  if (!h || !name || !name[0]) return fail("null argument");
  if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail("gpb_hip_vecchia_mailbox_attach: rank %d / world %d", rank, world);
  HIP_OK(hipSetDevice(h->device));
  h->mbox.release();
  const int fd = shm_open(name, O_RDWR, 0600);
  if (fd < 0) return fail("gpb_hip_vecchia_mailbox_attach: shm_open(%s) failed: %s", name, std::strerror(errno));
  const size_t bytes = GpbMailbox::total_bytes(world);
  struct stat st;
  // a failed attach ends the mailbox for every rank (the callers agree on that by an all-reduce of a success flag): whoever fails removes the name,
  // so that no /dev/shm/gpb_mbox_* outlives the job (the mappings of ranks already attached keep the memory alive until they release it)
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) { (void)close(fd); (void)shm_unlink(name); return fail("gpb_hip_vecchia_mailbox_attach: segment %s is smaller than %zu bytes (world mismatch?)", name, bytes); }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  (void)close(fd);
  if (p == MAP_FAILED) { (void)shm_unlink(name); return fail("gpb_hip_vecchia_mailbox_attach: mmap failed: %s", std::strerror(errno)); }
  auto* hdr = static_cast<volatile unsigned long long*>(p);
  if (hdr[0] != GpbMailbox::kMagic || hdr[1] != (unsigned long long)world) { (void)munmap(p, bytes); (void)shm_unlink(name); return fail("gpb_hip_vecchia_mailbox_attach: %s is not a mailbox for %d ranks", name, world); }
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped);
  if (e != hipSuccess) { (void)munmap(p, bytes); (void)shm_unlink(name); return fail("hipHostRegister of the mailbox failed: %s", hipGetErrorString(e)); }
  void* dp = nullptr;
  e = hipHostGetDevicePointer(&dp, p, 0);
  if (e != hipSuccess) { (void)hipHostUnregister(p); (void)munmap(p, bytes); (void)shm_unlink(name); return fail("hipHostGetDevicePointer of the mailbox failed: %s", hipGetErrorString(e)); }
  GpbMailbox& mb = h->mbox;
  mb.base = p; mb.bytes = bytes; mb.dev_base = static_cast<double*>(dp); mb.world = world; mb.rank = rank; mb.evals = 0;
  std::strncpy(mb.name, name, sizeof(mb.name) - 1);
  for (int g = 0; g < 3; ++g) for (int t = 0; t < 8; ++t) mb.slot(g, rank)[t] = kFetchSentinel;
  std::atomic_thread_fence(std::memory_order_release);
  hdr[2 + rank] = 1ull;
  const double attach_limit_s = GpbMailbox::env_seconds("GPB_MAILBOX_ATTACH_TIMEOUT_S", 120.0);
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    bool all = true;
    for (int r = 0; r < world; ++r) all = all && hdr[2 + r] != 0ull;
    if (all) break;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > attach_limit_s) {
      mb.release(); (void)shm_unlink(name);
      return fail("gpb_hip_vecchia_mailbox_attach: not all %d ranks attached within %.0f s (GPB_MAILBOX_ATTACH_TIMEOUT_S)", world, attach_limit_s);
    }
    std::this_thread::yield();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  if (rank == 0) (void)shm_unlink(name);
  API_END();
}
int gpb_hip_vecchia_mailbox_info(gpb_hip_vecchia_t* h, int* rank, int* world) {
  API_BEGIN();
  if (!h) return fail("null argument");
  if (rank) *rank = h->mbox.active() ? h->mbox.rank : 0;
  if (world) *world = h->mbox.active() ? h->mbox.world : 0;
  API_END();
}
int gpb_hip_vecchia_mailbox_detach(gpb_hip_vecchia_t* h) {
  API_BEGIN();
  if (!h) return 0;
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->mbox.release();
  API_END();
}

// one sharded evaluation through the mailbox: ONE launch (the point kernel, whose finisher writes this rank's slot), then the host polls all slots
static int vecchia_mailbox_terms(gpb_hip_vecchia_t* h, int mode, int cov_type, double var, double a, int gauss, double* out_host, int nout) {
  GpbMailbox& mb = h->mbox;
  if (mb.dead) return fail("mailbox: an earlier evaluation failed or timed out on this rank; the ranks' slot generations may be out of step -- detach and attach a new mailbox");
  // the evaluation counter advances only once this rank's launch is enqueued: a launch that fails leaves the rank in step with nothing written, and the
  // mailbox is marked dead (the peers run into their poll limit for this evaluation and mark theirs dead too -- no later evaluation reads stale slots)
  const unsigned long long e = mb.evals + 1;
  const int gen = (int)(e % 3), next = (int)((e + 1) % 3);
  for (int t = 0; t < 8; ++t) mb.slot(next, mb.rank)[t] = kFetchSentinel;       // re-arm the own slot of the generation after this one (see GpbMailbox)
  std::atomic_thread_fence(std::memory_order_release);
  if (vecchia_launch(h, mode, cov_type, var, a, gauss, nullptr, nout, nullptr, nullptr, mb.dev_slot(gen, mb.rank))) { mb.dead = true; return -1; }
  mb.evals = e;
  h->launches_unfetched = 0;
  const int nterms = nout > 3 ? GPB_NUM_PARTIALS : 3;
  const double limit_s = GpbMailbox::env_seconds("GPB_MAILBOX_TIMEOUT_S", 30.0);
  const auto t0 = std::chrono::steady_clock::now();
  bool own_synced = false;
  for (unsigned spin = 0;; ++spin) {
    bool all = true;
    for (int r = 0; r < mb.world && all; ++r) {
      const volatile unsigned long long* v = mb.slot(gen, r);
      for (int t = 0; t < nterms; ++t) all = all && v[t] != kFetchSentinel;
    }
    if (all) break;
    if ((spin & 1023u) == 1023u) {
      const auto dt = std::chrono::steady_clock::now() - t0;
      if (!own_synced && dt > std::chrono::milliseconds(50)) {      // (surfaces a launch error of this rank)
        const hipError_t se = hipStreamSynchronize(h->stream);
        if (se != hipSuccess) { mb.dead = true; return fail("mailbox: this rank's evaluation failed on the device: %s", hipGetErrorString(se)); }
        own_synced = true;
      }
      if (std::chrono::duration<double>(dt).count() > limit_s) {
        mb.dead = true;
        return fail("mailbox: the sums of all %d ranks did not arrive within %.0f s (evaluation %llu; GPB_MAILBOX_TIMEOUT_S)", mb.world, limit_s, e);
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  double acc[GPB_NUM_PARTIALS];
  for (int t = 0; t < nterms; ++t) acc[t] = 0.0;
  for (int r = 0; r < mb.world; ++r) {                                             // rank order: the same bits on every rank
    const volatile unsigned long long* v = mb.slot(gen, r);
    for (int t = 0; t < nterms; ++t) { const unsigned long long raw = v[t]; double d; std::memcpy(&d, &raw, 8); acc[t] += d; }
  }
  out_host[0] = acc[gpb::GPB_P_QUAD];
  out_host[1] = acc[gpb::GPB_P_LOGDET];
  for (int t = 2; t < nout; ++t) out_host[t] = acc[t];
  return 0;
}

// point kernel + fixed-order reduction + ncclAllReduce(sum) of the 3 / 7 terms on the handle's stream, result to the host
static int vecchia_allreduce_terms(gpb_hip_vecchia_t* h, int mode, int cov_type, double var, double a, int gauss,
                                   double* out_host, int nout) {
  if (!h || !out_host) return fail("null argument");
  if (h->mbox.active()) return vecchia_mailbox_terms(h, mode, cov_type, var, a, gauss, out_host, nout);
  if (!h->comm.active()) return fail("no communicator: call gpb_hip_vecchia_comm_init (or gpb_hip_vecchia_mailbox_attach) first");
  if (vecchia_launch(h, mode, cov_type, var, a, gauss, h->d_red, nout)) return -1;
  if (comm_allreduce(h->comm, h->d_red, (size_t)nout, GPB_T_F64, GPB_OP_SUM, h->stream)) return -1;
  // the job's sums reach the host without a copy engine and without the wake-up of a stream synchronisation: a one-wavefront kernel
  // stores them into pinned memory behind the all-reduce, the host polls (as vecchia_fetch does for the single-GPU evaluation)
  volatile unsigned long long* v = reinterpret_cast<volatile unsigned long long*>(h->h_red);
  for (int t = 0; t < nout; ++t) v[t] = kFetchSentinel;
  HIP_OK(gpb::launch_publish(h->d_red, h->h_red, nout, h->stream));
  h->launches_unfetched = 0;
  bool polled = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    bool all = true;
    for (int t = 0; t < nout; ++t) all = all && v[t] != kFetchSentinel;
    if (all) { polled = true; break; }
    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  if (!polled) HIP_OK(hipStreamSynchronize(h->stream));
  for (int t = 0; t < nout; ++t) out_host[t] = h->h_red[t];
  return 0;
}

static int yaux_enqueue(gpb_hip_vecchia_t* h);
// multi-GPU neighbour search: every rank searched its block of positions (gpb_hip_vecchia_find_neighbors_part(rank, world));
// the table is completed by one max-all-reduce (rows a rank did not search hold a value < -1), the duplicates flag by another
int gpb_hip_vecchia_neighbors_allreduce(gpb_hip_vecchia_t* h, int* has_duplicates) {
  API_BEGIN();
  if (!h) return fail("null handle");
  if (!h->comm.active()) return fail("no communicator: call gpb_hip_vecchia_comm_init first");
  if (!h->nn_partial && !h->has_nn) return fail("no neighbour search has run on this handle");
  HIP_OK(hipSetDevice(h->device));
  if (comm_allreduce(h->comm, h->d_nn, (size_t)h->n * h->m, GPB_T_I32, GPB_OP_MAX, h->stream)) return -1;
  if (comm_allreduce(h->comm, h->d_flag, 1, GPB_T_I32, GPB_OP_MAX, h->stream)) return -1;
  int flag = 0;
  HIP_OK(hipMemcpyAsync(&flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (has_duplicates) *has_duplicates = flag;
  h->has_nn = true; h->nn_partial = false;
  h->sorted_ready = false; h->launches_on_table = 0;      // a new neighbour table: the sorted gather is rebuilt for it
  API_END();
}

// multi-GPU y_aux: this shard's contribution, summed over the ranks in place (one all-reduce of n doubles), to the host
int gpb_hip_vecchia_yaux_allreduce(gpb_hip_vecchia_t* h, double* yaux_host) {
  API_BEGIN();
  if (!h || !yaux_host) return fail("null argument");
  if (!h->comm.active()) return fail("no communicator: call gpb_hip_vecchia_comm_init first");
  if (yaux_enqueue(h)) return -1;
  if (comm_allreduce(h->comm, h->d_w, (size_t)h->n, GPB_T_F64, GPB_OP_SUM, h->stream)) return -1;
  HIP_OK(hipMemcpyAsync(yaux_host, h->d_w, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

int gpb_hip_vecchia_nll_terms_allreduce(gpb_hip_vecchia_t* h, int cov_type, double var, double a, int gauss_likelihood,
                                        double* out3_host) {
  API_BEGIN();
  if (vecchia_allreduce_terms(h, gpb::MODE_NLL, cov_type, var, a, gauss_likelihood, out3_host, 3)) return -1;
  API_END();
}

int gpb_hip_vecchia_grad_terms_allreduce(gpb_hip_vecchia_t* h, int cov_type, double var, double a, double* out7_host) {
  API_BEGIN();
  if (vecchia_allreduce_terms(h, gpb::MODE_GRAD, cov_type, var, a, 1, out7_host, 7)) return -1;
  API_END();
}

/* K likelihood evaluations (K parameter sets: the trial points of a line search, a grid, a batch of proposals) with ONE synchronisation:
   the K point-kernel + reduction launches are enqueued back to back, the 3 K shard sums are combined by ONE ncclAllReduce when the handle
   has a communicator, and copied to the host once.  At N = 8 GPUs a single evaluation is latency-bound (kernel 115 us + launch / collective
   / sync ~50 us); batched, the per-evaluation cost is the kernel's.  out: K x {y' Psi^-1 y, log|Psi|, #(D <= 0)}. */
int gpb_hip_vecchia_nll_terms_batch(gpb_hip_vecchia_t* h, int cov_type, int32_t K, const double* var, const double* a, int gauss_likelihood,
                                    double* out3K_host) {
  API_BEGIN();
  if (!h || !var || !a || !out3K_host) return fail("null argument");
  if (K < 1 || K > 4096) return fail("gpb_hip_vecchia_nll_terms_batch: K = %d (1..4096)", K);
  HIP_OK(hipSetDevice(h->device));
  if (h->mbox.active() && !h->comm.active()) {      // the mailbox is the handle's only transport (no RCCL: e.g. several ranks on one device): K evaluations through it
    for (int k = 0; k < K; ++k)
      if (vecchia_mailbox_terms(h, gpb::MODE_NLL, cov_type, var[k], a[k], gauss_likelihood, out3K_host + (size_t)3 * k, 3)) return -1;
    return 0;
  }
  if (h->batch_cap < (size_t)3 * K) {
    dev_free(h->d_batch);
    HIP_OK(hipMalloc(&h->d_batch, sizeof(double) * 3 * (size_t)K));
    h->batch_cap = (size_t)3 * K;
  }
  for (int k = 0; k < K; ++k)
    if (vecchia_launch(h, gpb::MODE_NLL, cov_type, var[k], a[k], gauss_likelihood, h->d_batch + (size_t)3 * k, 3)) return -1;
  if (h->comm.active() && comm_allreduce(h->comm, h->d_batch, (size_t)3 * K, GPB_T_F64, GPB_OP_SUM, h->stream)) return -1;
  HIP_OK(hipMemcpyAsync(out3K_host, h->d_batch, sizeof(double) * 3 * (size_t)K, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

int gpb_hip_vecchia_grad_terms(gpb_hip_vecchia_t* h, int cov_type, double var, double a, double* out7_host) {
  API_BEGIN();
  if (!out7_host) return fail("null output");
  if (vecchia_launch(h, gpb::MODE_GRAD, cov_type, var, a, 1, nullptr, 0)) return -1;
  if (vecchia_fetch(h, out7_host, 7)) return -1;
  API_END();
}

int gpb_hip_vecchia_grad_terms_dev(gpb_hip_vecchia_t* h, int cov_type, double var, double a, double* out7_dev) {
  API_BEGIN();
  if (!out7_dev) return fail("null output");
  if (vecchia_launch(h, gpb::MODE_GRAD, cov_type, var, a, 1, out7_dev, 7)) return -1;
  API_END();
}

int gpb_hip_vecchia_bench(gpb_hip_vecchia_t* h, int mode, int cov_type, double var, double a, int warmup, int steps,
                          double* ms_total, double* ms_point_kernel_avg, double* out7_host) {
  API_BEGIN();
  if (!h || !ms_total || !ms_point_kernel_avg) return fail("null argument");
  if (mode != gpb::MODE_NLL && mode != gpb::MODE_GRAD) return fail("gpb_hip_vecchia_bench: mode must be 0 (nll) or 2 (grad)");
  if (steps < 1 || steps > 100000) return fail("gpb_hip_vecchia_bench: steps = %d", steps);
  HIP_OK(hipSetDevice(h->device));
  for (int w = 0; w < warmup; ++w)
    if (vecchia_launch(h, mode, cov_type, var * (1. + 1e-3 * (w + 1)), a, 1, nullptr, 0)) return -1;
  HIP_OK(hipStreamSynchronize(h->stream));
  std::vector<hipEvent_t> e0(steps), e1(steps);
  hipEvent_t t0, t1;
  HIP_OK(hipEventCreate(&t0)); HIP_OK(hipEventCreate(&t1));
  for (int s = 0; s < steps; ++s) { HIP_OK(hipEventCreate(&e0[s])); HIP_OK(hipEventCreate(&e1[s])); }
  HIP_OK(hipEventRecord(t0, h->stream));
  for (int s = 0; s < steps; ++s)   // parameters change every step: nothing can be cached between evaluations
    if (vecchia_launch(h, mode, cov_type, var * (1. + 1e-3 * (s + 1)), a * (1. - 1e-3 * (s % 7)), 1, nullptr, 0, e0[s], e1[s])) return -1;
  HIP_OK(hipEventRecord(t1, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, t0, t1));
  *ms_total = ms;
  double acc = 0.;
  for (int s = 0; s < steps; ++s) { HIP_OK(hipEventElapsedTime(&ms, e0[s], e1[s])); acc += ms; }
  *ms_point_kernel_avg = acc / steps;
  for (int s = 0; s < steps; ++s) { (void)hipEventDestroy(e0[s]); (void)hipEventDestroy(e1[s]); }
  (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
  if (out7_host && vecchia_fetch(h, out7_host, 7)) return -1;
  API_END();
}

/* In-loop kernel timing: enable = 1 starts recording an event pair around every point-kernel launch of this handle (ring of 256 pairs, count reset);
   enable = 0 stops, synchronises and returns the number of launches seen and the mean kernel time of the last min(count, 256) of them. */
int gpb_hip_vecchia_timing(gpb_hip_vecchia_t* h, int enable, int64_t* launches, double* mean_kernel_ms) {
  API_BEGIN();
  if (!h) return fail("null handle");
  HIP_OK(hipSetDevice(h->device));
  constexpr int kPairs = 256;
  if (enable) {
    if (h->timing_ev.empty()) {
      h->timing_ev.resize(2 * kPairs);
      for (auto& ev : h->timing_ev) HIP_OK(hipEventCreate(&ev));
    }
    h->timing_count = 0; h->timing_on = true;
    return 0;
  }
  h->timing_on = false;
  HIP_OK(hipStreamSynchronize(h->stream));
  const unsigned long long cnt = h->timing_count;
  double acc = 0.;
  const int used = (int)std::min<unsigned long long>(cnt, kPairs);
  for (int q = 0; q < used; ++q) {
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, h->timing_ev[2 * q], h->timing_ev[2 * q + 1]));
    acc += ms;
  }
  if (launches) *launches = (int64_t)cnt;
  if (mean_kernel_ms) *mean_kernel_ms = used ? acc / used : 0.;
  API_END();
}

/* ---- linear-regression covariates (Gaussian likelihood; GPB_OptimLinRegrCoefCovPar with optimizer_coef "wls": the coefficients are profiled out
   by generalised least squares at every evaluation, optim_utils.h:296-302 -> ProfileOutCoef, re_model_template.h:2665-2683) ----
   X: p columns of n values in Vecchia order (column-major [p][n]); the response last uploaded with gpb_hip_vecchia_set_y is y0. */
int gpb_hip_vecchia_set_covariates(gpb_hip_vecchia_t* h, int32_t p, const double* X_colmajor) {
  API_BEGIN();
  if (!h) return fail("null handle");
  if (p < 0 || p > 256 || (p > 0 && !X_colmajor)) return fail("gpb_hip_vecchia_set_covariates: p = %d", p);
  HIP_OK(hipSetDevice(h->device));
  dev_free(h->d_X); dev_free(h->d_U); dev_free(h->d_G); dev_free(h->d_beta);
  h->p_cov = p;
  if (p == 0) return 0;
  HIP_OK(hipMalloc(&h->d_X, sizeof(double) * (size_t)p * h->n));
  HIP_OK(hipMalloc(&h->d_U, sizeof(double) * (size_t)(p + 1) * h->n));
  HIP_OK(hipMalloc(&h->d_G, sizeof(double) * (size_t)(p + 1) * (p + 1)));
  HIP_OK(hipMalloc(&h->d_beta, sizeof(double) * (size_t)p));
  HIP_OK(hipMemcpy(h->d_X, X_colmajor, sizeof(double) * (size_t)p * h->n, hipMemcpyHostToDevice));
  API_END();
}

/* Gram matrix of [X, y0] in the Psi^-1 inner product from the factor on the device (gpb_hip_vecchia_factor must have run at the parameters
   of interest): G = (B [X, y0])' D^-1 (B [X, y0]), (p+1) x (p+1) row-major -- X' Psi^-1 X (CalcXTPsiInvX, re_model_template.h:6624-6628),
   X' Psi^-1 y0 and y0' Psi^-1 y0 in one pass. */
int gpb_hip_vecchia_gram(gpb_hip_vecchia_t* h, double* G_host) {
  API_BEGIN();
  if (!h || !G_host) return fail("null argument");
  if (h->p_cov < 1) return fail("gpb_hip_vecchia_gram: no covariates have been set");
  if (!h->has_factor) return fail("gpb_hip_vecchia_gram needs the factor (call gpb_hip_vecchia_factor first)");
  if (!h->d_ystage || !h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  if (h->i_begin != 0 || h->i_end != h->n) return fail("gpb_hip_vecchia_gram needs the whole factor on this device");
  HIP_OK(hipSetDevice(h->device));
  const int n = h->n, q = h->p_cov + 1;
  for (int j = 0; j < h->p_cov; ++j) HIP_OK(gpb::launch_By(h->d_A, h->d_nn, n, h->m, h->d_X + (size_t)j * n, h->d_U + (size_t)j * n, h->stream));
  HIP_OK(gpb::launch_By(h->d_A, h->d_nn, n, h->m, h->d_ystage, h->d_U + (size_t)h->p_cov * n, h->stream));
  HIP_OK(gpb::launch_gram(h->d_U, h->d_D, n, q, h->d_G, h->stream));
  HIP_OK(hipMemcpyAsync(G_host, h->d_G, sizeof(double) * (size_t)q * q, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

/* response := y0 - X beta (UpdateFixedEffects, re_model_template.h:2859-2871); beta = NULL restores y0 */
int gpb_hip_vecchia_set_resid(gpb_hip_vecchia_t* h, const double* beta_host) {
  API_BEGIN();
  if (!h) return fail("null handle");
  if (!h->d_ystage || !h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  HIP_OK(hipSetDevice(h->device));
  h->gpts_dirty = true;
  if (!beta_host || h->p_cov < 1) { HIP_OK(gpb::launch_pack_y(h->d_pts, h->d_ystage, h->n, h->stream)); }
  else {
    HIP_OK(hipMemcpyAsync(h->d_beta, beta_host, sizeof(double) * (size_t)h->p_cov, hipMemcpyHostToDevice, h->stream));
    HIP_OK(gpb::launch_resid(h->d_pts, h->d_ystage, h->d_X, h->d_beta, h->n, h->p_cov, h->stream));
  }
  HIP_OK(hipStreamSynchronize(h->stream));     // beta_host is borrowed for the call only
  response_changed(h);
  API_END();
}

int gpb_hip_vecchia_factor(gpb_hip_vecchia_t* h, int cov_type, double var, double a, int gauss_likelihood) {
  API_BEGIN();
  if (!h) return fail("null handle");
  HIP_OK(hipSetDevice(h->device));
  if (!h->d_A) {
    HIP_OK(hipMalloc(&h->d_A, sizeof(double) * (size_t)h->n * h->m));
    HIP_OK(hipMalloc(&h->d_D, sizeof(double) * (size_t)h->n));
    HIP_OK(hipMalloc(&h->d_u, sizeof(double) * (size_t)h->n));
  }
  h->has_yaux = false;
  if (vecchia_launch(h, gpb::MODE_FACTOR, cov_type, var, a, gauss_likelihood, nullptr, 0)) return -1;
  HIP_OK(hipStreamSynchronize(h->stream));
  h->has_factor = true; h->u_stale = false;
  API_END();
}

/* Lloyd iterations of the inducing-point selection (kmeans_plusplus, src/GPBoost/GP_utils.cpp:282-308, after its random_plusplus start on the host):
   the assignment step -- n x k distances per iteration, all of the 10 s the host spent at n = 1e5, k = 200 -- runs on the device with the reference's
   arithmetic (nn_kernels.hip: kmeans_assign_kernel); the mean update stays on the host in the reference's order (per mean: its rows ascending, then one
   division), so the means are the reference's bit for bit; iterations until the means repeat (the previous or the one before) or max_it.
   x: column-major n x d (host); means: ROW-major k x d, in: the start, out: the result. */
int gpb_hip_kmeans_lloyd(int32_t n, int32_t d, const double* x_colmajor, int32_t k, double* means_rowmajor, int32_t max_it, int32_t* iterations) {
  API_BEGIN();
  if (!x_colmajor || !means_rowmajor || n < 1 || d < 1 || d > 3 || k < 1 || k > 256) return fail("gpb_hip_kmeans_lloyd: invalid argument (d <= 3, k <= 256)");
  if (check_device()) return -1;                        // (the current device, as gpb_hip_vecchia_create)
  double* d_x = nullptr; double* d_m = nullptr; int* d_cl = nullptr;
  auto guard = scope_exit([&] { dev_free(d_x); dev_free(d_m); dev_free(d_cl); });
  HIP_OK(hipMalloc(&d_x, sizeof(double) * (size_t)n * d));
  HIP_OK(hipMalloc(&d_m, sizeof(double) * (size_t)k * d));
  HIP_OK(hipMalloc(&d_cl, sizeof(int) * (size_t)n));
  HIP_OK(hipMemcpy(d_x, x_colmajor, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice));
  std::vector<double> means(means_rowmajor, means_rowmajor + (size_t)k * d), old(means.size(), 0.), oldold(means.size(), 0.), mnew(means.size());
  std::vector<int> cl(n), cnt(k);
  int count = 0;
  do {
    oldold = old; old = means;
    HIP_OK(hipMemcpy(d_m, means.data(), sizeof(double) * means.size(), hipMemcpyHostToDevice));
    HIP_OK(gpb::launch_kmeans_assign(d_x, d_m, n, d, k, d_cl, nullptr));
    HIP_OK(hipMemcpy(cl.data(), d_cl, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    std::fill(mnew.begin(), mnew.end(), 0.); std::fill(cnt.begin(), cnt.end(), 0);
    for (int r = 0; r < n; ++r) { for (int c = 0; c < d; ++c) mnew[(size_t)cl[r] * d + c] += x_colmajor[(size_t)c * n + r]; cnt[cl[r]]++; }   // per mean: its rows in ascending order
    for (int j = 0; j < k; ++j) if (cnt[j] > 0) for (int c = 0; c < d; ++c) means[(size_t)j * d + c] = mnew[(size_t)j * d + c] / cnt[j];
    ++count;
  } while (means != old && means != oldold && count != max_it);
  std::copy(means.begin(), means.end(), means_rowmajor);
  if (iterations) *iterations = count;
  API_END();
}

/* ---- full-scale Vecchia ("VIF") approximation, Gaussian likelihood (vif_kernels.hip) ----
   set_inducing_points: k <= 256 inducing points (column-major k x d, from the host's kmeans++); allocates the row-major n x kq matrices.
   vif_factor: C_nm (+ its range derivative), V = C L_m^-T (Linv: k x k row-major inverse of the host's chol(Sigma_m)), the residual-process factor
   A, D, u, Q = B [C, y] and the Gram matrix Q' D^-1 Q; out3 = {sum u_i^2 / D_i, sum log D_i, #(D_i <= 0)}, G: (k + 1) x (k + 1).
   vif_grad_sums: the four n x k x k products, the derivative factor kernel and its twelve sums. */
int gpb_hip_vecchia_vif_set_inducing_points(gpb_hip_vecchia_t* h, int32_t k, const double* ip_colmajor) {
  API_BEGIN();
  if (!h || !ip_colmajor) return fail("null argument");
  if (k < 1 || k > 256 || k >= h->n) return fail("gpb_hip_vecchia_vif_set_inducing_points: %d inducing points (1..256, fewer than data points) are supported", k);
  if (h->d > 3) return fail("full-scale Vecchia: coordinate dimensions 1..3 are on the HIP hot path (got %d)", h->d);
  HIP_OK(hipSetDevice(h->device));
  const int kp = (k | 1), kq = gpb::vif_kq(k);
  if (gpb::vif_resid_lds_bytes(h->m, kq) > 150 * 1024) return fail("full-scale Vecchia: %d neighbours x %d inducing points exceed the LDS of a CU", h->m, k);
  HIP_OK(hipStreamSynchronize(h->stream));
  vif_free(h);
  std::vector<double> ip3((size_t)k * 3, 0.0);
  for (int j = 0; j < k; ++j) for (int c = 0; c < h->d; ++c) ip3[(size_t)j * 3 + c] = ip_colmajor[(size_t)c * k + j];
  const size_t nk = sizeof(double) * (size_t)h->n * kq;
  HIP_OK(hipMalloc(&h->d_ip, sizeof(double) * ip3.size()));
  HIP_OK(hipMemcpy(h->d_ip, ip3.data(), sizeof(double) * ip3.size(), hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&h->d_vC, nk)); HIP_OK(hipMalloc(&h->d_V, nk)); HIP_OK(hipMalloc(&h->d_vQ, nk));
  HIP_OK(hipMalloc(&h->d_vM, sizeof(double) * 6 * (size_t)kq * kq));
  HIP_OK(hipMalloc(&h->d_vG, sizeof(double) * (size_t)kq * kq));
  HIP_OK(hipMalloc(&h->d_vgpart, sizeof(double) * gpb::vif_gram_part_doubles(h->n, kq)));
  HIP_OK(hipMalloc(&h->d_vif_part, sizeof(double) * GPB_VIF_GRAD_TERMS * (size_t)h->n));
  HIP_OK(hipMalloc(&h->d_vout, sizeof(double) * 16));
  h->vif_k = k; h->vif_kp = kp; h->vif_kq = kq; h->vif_ip_host = ip3;
  API_END();
}

// M [kq][kq] row-major (zero outside the k x k block) <- src (k x k row-major), optionally transposed, into slot `slot` of d_vM
static int vif_upload_matrix(gpb_hip_vecchia_t* h, int slot, const double* src, bool transpose) {
  const int k = h->vif_k, kq = h->vif_kq;
  std::vector<double> M((size_t)kq * kq, 0.0);
  for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) M[(size_t)i * kq + j] = transpose ? src[(size_t)j * k + i] : src[(size_t)i * k + j];
  HIP_OK(hipMemcpyAsync(h->d_vM + (size_t)slot * kq * kq, M.data(), sizeof(double) * M.size(), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));     // (M is a local buffer)
  return 0;
}

int gpb_hip_vecchia_vif_factor(gpb_hip_vecchia_t* h, int cov_type, double var, double a, const double* Linv_rowmajor, int with_grad, double* out3_host,
                               double* G_host) {
  API_BEGIN();
  if (!h || !Linv_rowmajor || !out3_host || !G_host) return fail("null argument");
  if (h->vif_k < 1) return fail("no inducing points have been set (call gpb_hip_vecchia_vif_set_inducing_points)");
  if (!h->has_nn) return fail("neighbours have not been determined (call gpb_hip_vecchia_find_neighbors / _set_neighbors)");
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  if (cov_type < 0 || cov_type > 2) return fail("covariance type %d is not on the HIP hot path (Matern 0.5/1.5/2.5 only)", cov_type);
  if (!(var > 0.) || !(a > 0.)) return fail("covariance parameters must be positive (var = %g, range = %g)", var, a);
  if (h->i_begin != 0 || h->i_end != h->n) return fail("full-scale Vecchia needs the whole factor on this device");
  HIP_OK(hipSetDevice(h->device));
  const int n = h->n, k = h->vif_k, kp = h->vif_kp, kq = h->vif_kq;
  if (!h->d_A) {
    HIP_OK(hipMalloc(&h->d_A, sizeof(double) * (size_t)n * h->m));
    HIP_OK(hipMalloc(&h->d_D, sizeof(double) * (size_t)n));
    HIP_OK(hipMalloc(&h->d_u, sizeof(double) * (size_t)n));
  }
  const size_t nk = sizeof(double) * (size_t)n * kq;
  if (with_grad && !h->d_vdC) { HIP_OK(hipMalloc(&h->d_vdC, nk)); HIP_OK(hipMalloc(&h->d_vQdC, nk)); }
  h->has_yaux = false; h->vif_has_grad_inputs = false; h->vif_has_grad_factor = false;
  if (vif_upload_matrix(h, 0, Linv_rowmajor, true)) return -1;                   // V = C Linv'
  HIP_OK(gpb::launch_vif_crosscov(cov_type, h->d_pts, h->d_ip, 0, n, k, kq, h->d, var, a, h->d_vC, with_grad ? h->d_vdC : nullptr, h->stream));
  HIP_OK(gpb::launch_vif_gemm(h->d_vC, h->d_vM, n, kq, h->d_V, false, h->stream));
  gpb::VecchiaKernelArgs ka;
  ka.pts = h->d_pts; ka.nn = h->d_nn; ka.exp_tab = h->d_exp_tab; ka.partials = h->d_vif_part;
  ka.A = h->d_A; ka.D = h->d_D; ka.u = h->d_u;
  ka.m = h->m; ka.i_begin = 0; ka.i_end = n;
  ka.var = var; ka.a = a; ka.diag_nn = var + 1.0; ka.diag_i = var + 1.0; ka.nugget = 1.0;
  HIP_OK(gpb::launch_vif_resid_factor(cov_type, ka, h->d_V, k, kq, h->stream));
  HIP_OK(gpb::launch_reduce_partials(h->d_vif_part, n, 3, h->d_vout, nullptr, h->stream, nullptr));
  HIP_OK(gpb::launch_vif_spmm(h->d_A, h->d_nn, 0, n, h->m, kq, h->d_vC, h->d_vQ, with_grad ? h->d_vdC : nullptr, with_grad ? h->d_vQdC : nullptr, h->stream));
  HIP_OK(gpb::launch_vif_gram(h->d_vQ, h->d_D, n, kq, h->d_vgpart, h->d_vG, h->stream));
  double o[3];
  HIP_OK(hipMemcpyAsync(o, h->d_vout, sizeof(double) * 3, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpy2DAsync(G_host, sizeof(double) * (size_t)(k + 1), h->d_vG, sizeof(double) * (size_t)kq, sizeof(double) * (size_t)(k + 1), (size_t)(k + 1),
                          hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  out3_host[0] = o[gpb::GPB_P_QUAD]; out3_host[1] = o[gpb::GPB_P_LOGDET]; out3_host[2] = o[gpb::GPB_P_BAD];
  h->has_factor = true; h->vif_has_grad_inputs = with_grad != 0;
  API_END();
}

/* Gradient sums of the full-scale Vecchia likelihood (after gpb_hip_vecchia_vif_factor(with_grad = 1) at the same parameters).
   Winv = W^-1 (Woodbury matrix), Si = Sigma_m^-1 (jittered), N0 = 2 Si - Si dSigma_m^var Si, negMp1 = -Si dSigma_m^range Si: k x k row-major (symmetric);
   w = W^-1 (B C)' D^-1 B y.  sums12 = {S1, S2, S3, S4, S5, S6} x {variance, range} in the order [2 * S + p] (vif_kernels.hip).  keep_factor: also keep
   dA / dD on the device for gpb_hip_vecchia_vif_get_grad_factor (tests). */
static int vif_grad_sums_impl(gpb_hip_vecchia_t* h, int cov_type, double var, double a, const double* Winv, const double* Si, const double* N0,
                              const double* negMp1, const double* w_host, int keep_factor, double* sums12_host, bool latent);
int gpb_hip_vecchia_vif_grad_sums(gpb_hip_vecchia_t* h, int cov_type, double var, double a, const double* Winv, const double* Si, const double* N0,
                                  const double* negMp1, const double* w_host, int keep_factor, double* sums12_host) {
  API_BEGIN();
  if (vif_grad_sums_impl(h, cov_type, var, a, Winv, Si, N0, negMp1, w_host, keep_factor, sums12_host, false)) return -1;
  API_END();
}
// latent: the residual process of a NON-GAUSSIAN model (no nugget, the neighbours' diagonal x (1 + 1e-10); gpb_laplace.inc uses the derivative factors dA / dD only)
static int vif_grad_sums_impl(gpb_hip_vecchia_t* h, int cov_type, double var, double a, const double* Winv, const double* Si, const double* N0,
                              const double* negMp1, const double* w_host, int keep_factor, double* sums12_host, bool latent) {
  {
  if (!h || !Winv || !Si || !N0 || !negMp1 || !w_host || !sums12_host) return fail("null argument");
  if (h->vif_k < 1 || !h->has_factor || !h->vif_has_grad_inputs) return fail("gpb_hip_vecchia_vif_grad_sums needs gpb_hip_vecchia_vif_factor(with_grad = 1) first");
  if (cov_type < 0 || cov_type > 2) return fail("covariance type %d is not on the HIP hot path (Matern 0.5/1.5/2.5 only)", cov_type);
  HIP_OK(hipSetDevice(h->device));
  const int n = h->n, k = h->vif_k, kp = h->vif_kp, kq = h->vif_kq;
  const size_t nk = sizeof(double) * (size_t)n * kq;
  if (!h->d_vX1) {
    HIP_OK(hipMalloc(&h->d_vX1, nk)); HIP_OK(hipMalloc(&h->d_vV1, nk)); HIP_OK(hipMalloc(&h->d_vX2, nk)); HIP_OK(hipMalloc(&h->d_vHm, nk));
    HIP_OK(hipMalloc(&h->d_vw, sizeof(double) * (size_t)kq)); HIP_OK(hipMalloc(&h->d_vv, sizeof(double) * (size_t)n)); HIP_OK(hipMalloc(&h->d_vz, sizeof(double) * (size_t)n));
  }
  if (keep_factor && !h->d_vdA) {
    HIP_OK(hipMalloc(&h->d_vdA, sizeof(double) * 2 * (size_t)n * h->m)); HIP_OK(hipMalloc(&h->d_vdD, sizeof(double) * 2 * (size_t)n));
  }
  if (vif_upload_matrix(h, 1, Winv, false) || vif_upload_matrix(h, 2, Si, false) || vif_upload_matrix(h, 3, N0, false) || vif_upload_matrix(h, 4, negMp1, false)) return -1;
  std::vector<double> wq((size_t)kq, 0.0);
  for (int j = 0; j < k; ++j) wq[j] = w_host[j];
  HIP_OK(hipMemcpyAsync(h->d_vw, wq.data(), sizeof(double) * (size_t)kq, hipMemcpyHostToDevice, h->stream));
  const size_t mm = (size_t)kq * kq;
  HIP_OK(gpb::launch_vif_gemm(h->d_vQ, h->d_vM + 1 * mm, n, kq, h->d_vHm, false, h->stream));      // Hm  = Q W^-1
  HIP_OK(gpb::launch_vif_gemm(h->d_vQ, h->d_vM + 2 * mm, n, kq, h->d_vX1, false, h->stream));      // X1  = Q Si
  HIP_OK(gpb::launch_vif_gemm(h->d_vQ, h->d_vM + 3 * mm, n, kq, h->d_vV1, false, h->stream));      // V1  = Q (2 Si - Si dSm0 Si) = X1 + X2^0
  HIP_OK(gpb::launch_vif_gemm(h->d_vQdC, h->d_vM + 2 * mm, n, kq, h->d_vX2, false, h->stream));    // X2r = (B dC) Si ...
  HIP_OK(gpb::launch_vif_gemm(h->d_vQ, h->d_vM + 4 * mm, n, kq, h->d_vX2, true, h->stream));       //       ... - Q Si dSm1 Si
  HIP_OK(gpb::launch_vif_vec(h->d_vQ, h->d_vC, h->d_D, h->d_vw, n, k, kq, h->d_vv, h->d_vz, h->stream));
  gpb::VecchiaKernelArgs ka;
  ka.pts = h->d_pts; ka.nn = h->d_nn; ka.exp_tab = h->d_exp_tab; ka.partials = nullptr;
  ka.A = h->d_A; ka.D = h->d_D; ka.u = h->d_u;
  ka.m = h->m; ka.i_begin = 0; ka.i_end = n;
  ka.var = var; ka.a = a; ka.diag_nn = var + 1.0; ka.diag_i = var + 1.0; ka.nugget = 1.0;
  if (latent) { ka.diag_nn = var; ka.diag_i = var; ka.nugget = 0.0; ka.diag_mult = 1.0 + 1e-10; }
  gpb::VifGradLaunch L;
  L.V = h->d_V; L.C = h->d_vC; L.dC = h->d_vdC; L.Q = h->d_vQ; L.QdC = h->d_vQdC; L.X1 = h->d_vX1; L.V1 = h->d_vV1; L.X2r = h->d_vX2; L.Hm = h->d_vHm;
  L.w = h->d_vw; L.v = h->d_vv; L.z = h->d_vz;
  L.dA0 = keep_factor ? h->d_vdA : nullptr; L.dA1 = keep_factor ? h->d_vdA + (size_t)n * h->m : nullptr;
  L.dD0 = keep_factor ? h->d_vdD : nullptr; L.dD1 = keep_factor ? h->d_vdD + n : nullptr;
  L.partials = h->d_vif_part;
  HIP_OK(gpb::launch_vif_resid_grad(cov_type, ka, L, k, kq, h->stream));
  HIP_OK(gpb::launch_reduce_partials(h->d_vif_part, n, GPB_VIF_GRAD_TERMS, h->d_vout, nullptr, h->stream, nullptr));
  HIP_OK(hipMemcpyAsync(sums12_host, h->d_vout, sizeof(double) * GPB_VIF_GRAD_TERMS, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->vif_has_grad_factor = keep_factor != 0;
  }
  return 0;
}

/* dA (n x m) and dD (n) of parameter p (0: variance, 1: range; log scale) of the last gpb_hip_vecchia_vif_grad_sums(keep_factor = 1), Vecchia order */
int gpb_hip_vecchia_vif_get_grad_factor(gpb_hip_vecchia_t* h, int p, double* dA_host, double* dD_host) {
  API_BEGIN();
  if (!h || !dA_host || !dD_host || p < 0 || p > 1) return fail("bad argument");
  if (!h->vif_has_grad_factor) return fail("gpb_hip_vecchia_vif_get_grad_factor needs gpb_hip_vecchia_vif_grad_sums(keep_factor = 1) first");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipMemcpy(dA_host, h->d_vdA + (size_t)p * h->n * h->m, sizeof(double) * (size_t)h->n * h->m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(dD_host, h->d_vdD + (size_t)p * h->n, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost));
  API_END();
}

/* Sample weights of a Gaussian model (re_model_template.h:403-431): observation i carries the error variance sigma^2 / w_i, i.e. on the
   transformed scale the nugget 1 / w_i instead of 1 on every diagonal entry that belongs to it (GetGaussianNuggetDiagFromWeights, :6393-6417;
   Vecchia_utils.cpp:1418-1422, 1610-1614).  nug: n values 1 / w_i in Vecchia order; NULL restores the uniform nugget. */
int gpb_hip_vecchia_set_nugget_diag(gpb_hip_vecchia_t* h, const double* nug_host) {
  API_BEGIN();
  if (!h) return fail("null handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (!nug_host) { dev_free(h->d_nug); }
  else {
    for (int i = 0; i < h->n; ++i) if (!(nug_host[i] > 0.) || !std::isfinite(nug_host[i])) return fail("gpb_hip_vecchia_set_nugget_diag: entry %d = %g (must be positive and finite)", i, nug_host[i]);
    if (h->vif_k > 0) return fail("sample weights with the full-scale Vecchia approximation are not on the HIP hot path");
    if (!h->d_nug) HIP_OK(hipMalloc(&h->d_nug, sizeof(double) * (size_t)h->n));
    HIP_OK(hipMemcpy(h->d_nug, nug_host, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice));
  }
  h->has_factor = false; h->u_stale = false; h->has_yaux = false;
  API_END();
}

int gpb_hip_vecchia_get_factor(gpb_hip_vecchia_t* h, double* A_host, double* D_host, double* u_host) {
  API_BEGIN();
  if (!h) return fail("null handle");
  if (!h->has_factor) return fail("the factor has not been computed (call gpb_hip_vecchia_factor)");
  HIP_OK(hipSetDevice(h->device));
  if (A_host) HIP_OK(hipMemcpyAsync(A_host, h->d_A, sizeof(double) * (size_t)h->n * h->m, hipMemcpyDeviceToHost, h->stream));
  if (D_host) HIP_OK(hipMemcpyAsync(D_host, h->d_D, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  if (u_host && refresh_u(h)) return -1;
  if (u_host) HIP_OK(hipMemcpyAsync(u_host, h->d_u, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

// Transposed neighbour index (columns of B): built once per neighbour table, on the host (counting sort).
static int build_transpose(gpb_hip_vecchia_t* h) {
  const int n = h->n, m = h->m;
  if (h->nn_host.empty()) {
    h->nn_host.resize((size_t)n * m);
    HIP_OK(hipMemcpyAsync(h->nn_host.data(), h->d_nn, sizeof(int) * (size_t)n * m, hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
  }
  std::vector<int> ptr(n + 1, 0);
  for (size_t e = 0; e < (size_t)n * m; ++e) if (h->nn_host[e] >= 0) ptr[h->nn_host[e] + 1]++;
  for (int j = 0; j < n; ++j) ptr[j + 1] += ptr[j];
  std::vector<int> pos(ptr[n] > 0 ? ptr[n] : 1), fill(ptr.begin(), ptr.end() - 1);
  for (size_t e = 0; e < (size_t)n * m; ++e) if (h->nn_host[e] >= 0) pos[fill[h->nn_host[e]]++] = (int)e;
  dev_free(h->d_tptr); dev_free(h->d_tpos);
  HIP_OK(hipMalloc(&h->d_tptr, sizeof(int) * (size_t)(n + 1)));
  HIP_OK(hipMalloc(&h->d_tpos, sizeof(int) * pos.size()));
  HIP_OK(hipMemcpy(h->d_tptr, ptr.data(), sizeof(int) * (size_t)(n + 1), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->d_tpos, pos.data(), sizeof(int) * pos.size(), hipMemcpyHostToDevice));
  h->has_transpose = true;
  return 0;
}

static int yaux_enqueue(gpb_hip_vecchia_t* h) {
  if (!h->has_factor) return fail("the factor has not been computed (call gpb_hip_vecchia_factor)");
  HIP_OK(hipSetDevice(h->device));
  if (!h->has_transpose && build_transpose(h)) return -1;
  if (!h->d_v) { HIP_OK(hipMalloc(&h->d_v, sizeof(double) * (size_t)h->n)); HIP_OK(hipMalloc(&h->d_w, sizeof(double) * (size_t)h->n)); }
  if (refresh_u(h)) return -1;
  HIP_OK(gpb::launch_scale_by_Dinv(h->d_u, h->d_D, h->n, h->i_begin, h->i_end, h->d_v, h->stream));
  HIP_OK(gpb::launch_Bt(h->d_A, h->d_tptr, h->d_tpos, h->n, h->m, h->i_begin, h->i_end, h->d_v, h->d_w, h->stream));
  h->has_yaux = true;
  return 0;
}

int gpb_hip_vecchia_yaux(gpb_hip_vecchia_t* h, double* yaux_host) {
  API_BEGIN();
  if (!h || !yaux_host) return fail("null argument");
  if (h->i_begin != 0 || h->i_end != h->n) return fail("gpb_hip_vecchia_yaux needs the full factor on this device (shard is [%d,%d)); use gpb_hip_vecchia_yaux_partial_dev + an all-reduce", h->i_begin, h->i_end);
  if (yaux_enqueue(h)) return -1;
  HIP_OK(hipMemcpyAsync(yaux_host, h->d_w, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

// diag(Psi^-1) = diag(B^T D^-1 B) on the transformed scale from the stored factor (full factor on this device): the predictive variances of
// the training-data random effects are sigma2 (1 - diag) (PredictTrainingDataRandomEffects, re_model_template.h:4508-4514)
int gpb_hip_vecchia_psi_inv_diag(gpb_hip_vecchia_t* h, double* diag_host) {
  API_BEGIN();
  if (!h || !diag_host) return fail("null argument");
  if (!h->has_factor) return fail("the factor has not been computed (call gpb_hip_vecchia_factor)");
  if (h->i_begin != 0 || h->i_end != h->n) return fail("gpb_hip_vecchia_psi_inv_diag needs the full factor on this device (shard is [%d,%d))", h->i_begin, h->i_end);
  HIP_OK(hipSetDevice(h->device));
  if (!h->has_transpose && build_transpose(h)) return -1;
  if (!h->d_v) { HIP_OK(hipMalloc(&h->d_v, sizeof(double) * (size_t)h->n)); HIP_OK(hipMalloc(&h->d_w, sizeof(double) * (size_t)h->n)); }
  HIP_OK(gpb::launch_BtDinvB_diag(h->d_A, h->d_D, h->d_tptr, h->d_tpos, h->n, h->m, h->d_v, h->stream));
  h->has_yaux = false;                                  // d_v is scratch of the y_aux pass too
  HIP_OK(hipMemcpyAsync(diag_host, h->d_v, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

int gpb_hip_vecchia_yaux_partial_dev(gpb_hip_vecchia_t* h, double* w_dev) {
  API_BEGIN();
  if (!h || !w_dev) return fail("null argument");
  if (yaux_enqueue(h)) return -1;
  HIP_OK(hipMemcpyAsync(w_dev, h->d_w, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
  API_END();
}

// Prediction at new locations, every prediction point conditioning on its nearest OBSERVED points only:
// CalcPredVecchiaObservedFirstOrder(CondObsOnly = true), src/GPBoost/Vecchia_utils.cpp:1701-2060, Gaussian likelihood.
// The reference appends the prediction coordinates to the observed ones, searches neighbours with start_at = n_obs and
// end_search_at = n_obs - 1 (:1792-1799), and runs the same per-point local factorisation as the likelihood (:1883-1975):
// pred_mean = A_p y_nn, Dp = 1 + sigma1^2/sigma^2 - A_p c.  Here that is one launch of the neighbour-search kernel and one of
// vecchia_point_kernel<MODE_FACTOR> over the appended rows (their own response slot is 0, so u = -A_p y_nn).
// Shared by the two prediction types: temporary state [observed (Vecchia order); prediction], neighbour search for the appended rows
// only (find_nearest_neighbors_Vecchia_fast with start_at = n_obs and end_search_at = n_obs - 1 [cond_obs_only] or -1 [cond_all],
// Vecchia_utils.cpp:1792-1822), MODE_FACTOR over the appended rows.  *out_t owns the state (caller frees), *out_m = neighbours used.
// pred_first / all_rows (the joint orderings of 'order_pred_first' and the 'latent_*' types, Vecchia_utils.cpp:2228-2255, 2517-2561): the
// prediction points come first / every row of the joint ordering is searched (start_at = 0) and factored, not only the appended ones.
static int predict_factor_appended(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred,
                                   bool cond_all, int cov_type, double var, double a, gpb_hip_vecchia_t** out_t, int* out_m, int* has_duplicates,
                                   int gauss_likelihood = 1, bool pred_first = false, bool all_rows = false, bool do_factor = true) {
  *out_t = nullptr;
  if (!h || !coords_pred_colmajor) return fail("null argument");
  if (n_pred < 1) return fail("Vecchia prediction: n_pred = %d", n_pred);
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_vecchia_set_y)");
  const int n_obs = h->n, d = h->d, n_all = n_obs + n_pred;
  int m = num_neighbors_pred;
  const int m_cap = cond_all ? n_all - 1 : n_obs;               // :755-758 with end_search_at = num_data - 2 / n_obs - 1
  if (m > m_cap) m = m_cap;
  if (m < 1 || m > GPB_MAX_NEIGHBORS_BIG) return fail("Vecchia prediction: num_neighbors_pred = %d (1..%d supported)", num_neighbors_pred, GPB_MAX_NEIGHBORS_BIG);
  HIP_OK(hipSetDevice(h->device));
  // [observed (Vecchia order); prediction] as one temporary state; the observed records (coordinates + y) are copied on the device
  if (pred_first && !(all_rows && cond_all)) return fail("Vecchia prediction: the prediction-first ordering searches and factors every row");
  const int obs0 = pred_first ? n_pred : 0, pred0 = pred_first ? 0 : n_obs;          // where the two groups start in the joint ordering
  std::vector<double> call((size_t)n_all * d);
  for (int c = 0; c < d; ++c) {
    std::copy(h->coords.begin() + (size_t)c * n_obs, h->coords.begin() + (size_t)(c + 1) * n_obs, call.begin() + (size_t)c * n_all + obs0);
    std::copy(coords_pred_colmajor + (size_t)c * n_pred, coords_pred_colmajor + (size_t)(c + 1) * n_pred, call.begin() + (size_t)c * n_all + pred0);
  }
  gpb_hip_vecchia_t* t = nullptr;
  if (gpb_hip_vecchia_create(n_all, d, m, call.data(), &t)) return -1;
  *out_t = t;
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipMemcpy(t->d_pts + obs0, h->d_pts, sizeof(double4) * (size_t)n_obs, hipMemcpyDeviceToDevice));   // pred rows keep y = 0
  t->has_y = true;
  if (h->d_nug && gauss_likelihood) {      // sample weights: observed neighbours carry 1 / w, prediction points the plain nugget (Vecchia_utils.cpp:1952-1958, 2386-2393); the latent factor has no nugget at all
    std::vector<double> ones((size_t)n_pred, 1.0);
    HIP_OK(hipMalloc(&t->d_nug, sizeof(double) * (size_t)n_all));
    HIP_OK(hipMemcpy(t->d_nug + obs0, h->d_nug, sizeof(double) * (size_t)n_obs, hipMemcpyDeviceToDevice));
    HIP_OK(hipMemcpy(t->d_nug + pred0, ones.data(), sizeof(double) * (size_t)n_pred, hipMemcpyHostToDevice));
  }
  // neighbour search for the appended rows only
  {
    std::vector<double> csum(n_all);
    for (int i = 0; i < n_all; ++i) { double s = call[i]; for (int c = 1; c < d; ++c) s += call[(size_t)c * n_all + i]; csum[i] = s; }
    std::vector<int> sort_sum(n_all);
    std::iota(sort_sum.begin(), sort_sum.end(), 0);
    const double* v = csum.data();
    std::sort(sort_sum.begin(), sort_sum.end(), [v](int i1, int i2) { return v[i1] < v[i2]; });
    std::vector<double4> rec(n_all);
    for (int k = 0; k < n_all; ++k) {
      const int i = sort_sum[k];
      rec[k].x = call[i]; rec[k].y = d > 1 ? call[(size_t)n_all + i] : 0.0; rec[k].z = d > 2 ? call[(size_t)2 * n_all + i] : 0.0; rec[k].w = csum[i];
    }
    double4* d_rec = nullptr; int* d_idx = nullptr; double* d_rec_nd = nullptr; int* d_qorder = nullptr;
    const auto free_tmp = scope_exit([&] { (void)hipFree(d_rec); (void)hipFree(d_idx); (void)hipFree(d_qorder); (void)hipFree(d_rec_nd); });
    HIP_OK(hipMalloc(&d_rec, sizeof(double4) * (size_t)n_all));
    HIP_OK(hipMalloc(&d_idx, sizeof(int) * (size_t)n_all));
    HIP_OK(hipMemcpyAsync(d_rec, rec.data(), sizeof(double4) * (size_t)n_all, hipMemcpyHostToDevice, t->stream));
    std::vector<double> rec_nd;
    if (d > 3) {
      rec_nd.resize((size_t)n_all * (d + 1));
      for (int k = 0; k < n_all; ++k) {
        const int i = sort_sum[k];
        for (int c = 0; c < d; ++c) rec_nd[(size_t)k * (d + 1) + c] = call[(size_t)c * n_all + i];
        rec_nd[(size_t)k * (d + 1) + d] = csum[i];
      }
      HIP_OK(hipMalloc(&d_rec_nd, sizeof(double) * rec_nd.size()));
      HIP_OK(hipMemcpyAsync(d_rec_nd, rec_nd.data(), sizeof(double) * rec_nd.size(), hipMemcpyHostToDevice, t->stream));
    }
    HIP_OK(hipMemcpyAsync(d_idx, sort_sum.data(), sizeof(int) * (size_t)n_all, hipMemcpyHostToDevice, t->stream));
    HIP_OK(hipMemsetAsync(t->d_flag, 0, sizeof(int), t->stream));
    HIP_OK(hipMemsetAsync(t->d_nn, 0xff, sizeof(int) * (size_t)n_all * t->m, t->stream));
    gpb::NNKernelArgs na;
    na.sorted_rec = d_rec; na.sorted_idx = d_idx; na.pts = t->d_pts; na.nn = t->d_nn; na.has_duplicates = t->d_flag; na.n = n_all; na.m = t->m;
    na.sorted_nd = d_rec_nd; na.coords_nd = t->d_coords_nd;
    const int start_at = all_rows ? 0 : n_obs;
    na.start_at = start_at; na.end_search_at = cond_all ? n_all - 2 : n_obs - 1; na.pos0 = 0; na.pos1 = n_all;
    // only the appended rows are searched (all_rows: every row): their positions, grouped like the training-time search
    std::vector<int> qorder((size_t)std::max(all_rows ? n_all : n_pred, 1));
    int nq = 0;
    gpb::nn_query_order(sort_sum.data(), 0, n_all, t->m, start_at, qorder.data(), &nq);
    HIP_OK(hipMalloc(&d_qorder, sizeof(int) * qorder.size()));
    HIP_OK(hipMemcpyAsync(d_qorder, qorder.data(), sizeof(int) * (size_t)std::max(nq, 1), hipMemcpyHostToDevice, t->stream));
    na.qorder = d_qorder; na.nq = nq;
    HIP_OK(gpb::launch_vecchia_nn(d, na, t->stream));
    int flag = 0;
    HIP_OK(hipMemcpyAsync(&flag, t->d_flag, sizeof(int), hipMemcpyDeviceToHost, t->stream));
    HIP_OK(hipStreamSynchronize(t->stream));
    if (has_duplicates) *has_duplicates = flag;
    t->has_nn = true;
  }
  t->i_begin = all_rows ? 0 : n_obs; t->i_end = n_all;
  if (!do_factor) { *out_m = m; return 0; }              // (full-scale Vecchia: the caller runs the residual-process factor)
  if (gpb_hip_vecchia_factor(t, cov_type, var, a, gauss_likelihood)) return -1;     // non-Gaussian: no nugget, diagonal x (1 + 1e-10) (Vecchia_utils.cpp:1963-1965)
  *out_m = m;
  return 0;
}

int gpb_hip_vecchia_predict_obs_only(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred,
                                     int cov_type, double var, double a, double* pred_mean, double* pred_D, int* has_duplicates) {
  API_BEGIN();
  if (!pred_mean || !pred_D) return fail("null argument");
  gpb_hip_vecchia_t* t = nullptr;
  int m = 0;
  const int rc = predict_factor_appended(h, n_pred, coords_pred_colmajor, num_neighbors_pred, false, cov_type, var, a, &t, &m, has_duplicates);
  struct Guard { gpb_hip_vecchia_t* p; ~Guard() { if (p) gpb_hip_vecchia_free(p); } } guard{t};
  if (rc) return -1;
  const int n_obs = h->n;
  std::vector<double> u(n_pred);
  HIP_OK(hipMemcpy(u.data(), t->d_u + n_obs, sizeof(double) * (size_t)n_pred, hipMemcpyDeviceToHost));
  for (int k = 0; k < n_pred; ++k) pred_mean[k] = -u[k];
  HIP_OK(hipMemcpy(pred_D, t->d_D + n_obs, sizeof(double) * (size_t)n_pred, hipMemcpyDeviceToHost));
  API_END();
}

// Latent predictive mean of a non-Gaussian (Vecchia-Laplace) model, 'latent_order_obs_first_cond_obs_only': pred_mean = -Bpo mode
// (PredictLaplaceApproxVecchia, likelihoods.h:8600-8602) with the factor rows of the prediction points computed without a nugget.
// The "response" on the handle must be the mode (gpb_hip_vecchia_set_y(mode), Vecchia order).
int gpb_hip_vecchia_predict_latent_obs_only(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred,
                                            int cov_type, double var, double a, double* pred_mean, int* has_duplicates) {
  API_BEGIN();
  if (!pred_mean) return fail("null argument");
  gpb_hip_vecchia_t* t = nullptr;
  int m = 0;
  const int rc = predict_factor_appended(h, n_pred, coords_pred_colmajor, num_neighbors_pred, false, cov_type, var, a, &t, &m, has_duplicates, 0);
  struct Guard { gpb_hip_vecchia_t* p; ~Guard() { if (p) gpb_hip_vecchia_free(p); } } guard{t};
  if (rc) return -1;
  std::vector<double> u(n_pred);
  HIP_OK(hipMemcpy(u.data(), t->d_u + h->n, sizeof(double) * (size_t)n_pred, hipMemcpyDeviceToHost));
  for (int k = 0; k < n_pred; ++k) pred_mean[k] = -u[k];
  API_END();
}

// 'order_obs_first_cond_all' (CondObsOnly = false, Vecchia_utils.cpp:1806-1822, 1883-1975): the factor rows of the appended prediction
// points, which condition on observed AND preceding prediction points.  Outputs, row-major [n_pred][*m_used]: neighbour indices into
// (observed, prediction) (-1 padded), A_p = C_nn^-1 c, and D_p (nugget included, transformed scale).  The forward substitution with
// Bp = I - A_pp and the rows of Bp^-1 are the host half (GPB_HIP_PredictCondAllHost).
int gpb_hip_vecchia_predict_cond_all(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred,
                                     int cov_type, double var, double a, int32_t* m_used, int32_t* nn_pred, double* A_pred, double* D_pred,
                                     int* has_duplicates) {
  API_BEGIN();
  if (!m_used || !nn_pred || !A_pred || !D_pred) return fail("null argument");
  gpb_hip_vecchia_t* t = nullptr;
  int m = 0;
  const int rc = predict_factor_appended(h, n_pred, coords_pred_colmajor, num_neighbors_pred, true, cov_type, var, a, &t, &m, has_duplicates);
  struct Guard { gpb_hip_vecchia_t* p; ~Guard() { if (p) gpb_hip_vecchia_free(p); } } guard{t};
  if (rc) return -1;
  const int n_obs = h->n;
  *m_used = m;
  HIP_OK(hipMemcpy(nn_pred, t->d_nn + (size_t)n_obs * m, sizeof(int) * (size_t)n_pred * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(A_pred, t->d_A + (size_t)n_obs * m, sizeof(double) * (size_t)n_pred * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(D_pred, t->d_D + n_obs, sizeof(double) * (size_t)n_pred, hipMemcpyDeviceToHost));
  API_END();
}

// The same rows for the LATENT process (no nugget, diagonal x (1 + 1e-10)): what PredictLaplaceApproxVecchia gets from
// CalcPredVecchiaObservedFirstOrder(CondObsOnly = false) for non-Gaussian likelihoods ('latent_order_obs_first_cond_all').
int gpb_hip_vecchia_predict_cond_all_latent(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred,
                                            int cov_type, double var, double a, int32_t* m_used, int32_t* nn_pred, double* A_pred, double* D_pred,
                                            int* has_duplicates) {
  API_BEGIN();
  if (!m_used || !nn_pred || !A_pred || !D_pred) return fail("null argument");
  gpb_hip_vecchia_t* t = nullptr;
  int m = 0;
  const int rc = predict_factor_appended(h, n_pred, coords_pred_colmajor, num_neighbors_pred, true, cov_type, var, a, &t, &m, has_duplicates, 0);
  struct Guard { gpb_hip_vecchia_t* p; ~Guard() { if (p) gpb_hip_vecchia_free(p); } } guard{t};
  if (rc) return -1;
  const int n_obs = h->n;
  *m_used = m;
  HIP_OK(hipMemcpy(nn_pred, t->d_nn + (size_t)n_obs * m, sizeof(int) * (size_t)n_pred * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(A_pred, t->d_A + (size_t)n_obs * m, sizeof(double) * (size_t)n_pred * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(D_pred, t->d_D + n_obs, sizeof(double) * (size_t)n_pred, hipMemcpyDeviceToHost));
  API_END();
}

// Full-scale Vecchia (VIF), prediction 'order_obs_first_cond_obs_only' (CalcPredVecchiaObservedFirstOrder with the full_scale_vecchia arguments,
// Vecchia_utils.cpp:1701-2060; re_model_template.h:4041-4056): the device half.  The prediction points are appended to the observed ones, their
// neighbours searched among the observed points, cross-covariances with the inducing points and their whitened form computed for all rows, and the
// RESIDUAL-process factor rows (A_p, D_p, u_p = -A_p y_nn) and (B C)_p = C_p - A_p C_nn for the appended rows.  The caller combines them with the
// Woodbury quantities of the observed points (GPB_PredictREModel): mean = -u_p + (B C)_p W^-1 (B C)' D^-1 B y, var = D_p + (B C)_p W^-1 (B C)_p'.
// ip_colmajor: k x d inducing points; Linv_rowmajor: inverse Cholesky factor of Sigma_m (as gpb_hip_vecchia_vif_factor takes it).
// Outputs: u_pred, D_pred (n_pred), BC_pred (n_pred x k, row-major).
static int vif_predict_appended(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred, bool cond_all,
                                const double* ip_colmajor, int cov_type, double var, double a, const double* Linv_rowmajor,
                                double* u_pred, double* D_pred, double* BC_pred, int* has_duplicates, int32_t* m_used, int32_t* nn_pred, double* A_pred);

int gpb_hip_vecchia_vif_predict_obs_only(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred,
                                         const double* ip_colmajor, int cov_type, double var, double a, const double* Linv_rowmajor,
                                         double* u_pred, double* D_pred, double* BC_pred, int* has_duplicates) {
  API_BEGIN();
  if (vif_predict_appended(h, n_pred, coords_pred_colmajor, num_neighbors_pred, false, ip_colmajor, cov_type, var, a, Linv_rowmajor, u_pred, D_pred, BC_pred,
                           has_duplicates, nullptr, nullptr, nullptr)) return -1;
  API_END();
}

// 'order_obs_first_cond_all' of a full-scale Vecchia model (round 5; CalcPredVecchiaObservedFirstOrder with CondObsOnly = false and the full_scale_vecchia
// arguments, Vecchia_utils.cpp:1803-1826, 1889-1925): the same device half with the neighbours of the appended points searched among the observed AND the
// preceding prediction points (the low-rank correction of a residual covariance reads the whitened cross-covariances of prediction points as well);
// returns in addition the rows of [Bpo Bp]: neighbour indices into (observed, prediction) points (-1 padded) and the coefficients A, m_used per row.
int gpb_hip_vecchia_vif_predict_cond_all(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred,
                                         const double* ip_colmajor, int cov_type, double var, double a, const double* Linv_rowmajor,
                                         int32_t* m_used, int32_t* nn_pred, double* A_pred, double* u_pred, double* D_pred, double* BC_pred, int* has_duplicates) {
  API_BEGIN();
  if (!m_used || !nn_pred || !A_pred) return fail("null argument");
  if (vif_predict_appended(h, n_pred, coords_pred_colmajor, num_neighbors_pred, true, ip_colmajor, cov_type, var, a, Linv_rowmajor, u_pred, D_pred, BC_pred,
                           has_duplicates, m_used, nn_pred, A_pred)) return -1;
  API_END();
}

static int vif_predict_appended(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred, bool cond_all,
                                const double* ip_colmajor, int cov_type, double var, double a, const double* Linv_rowmajor,
                                double* u_pred, double* D_pred, double* BC_pred, int* has_duplicates, int32_t* m_used, int32_t* nn_pred, double* A_pred) {
  {
  if (!h || !ip_colmajor || !Linv_rowmajor || !u_pred || !D_pred || !BC_pred) return fail("null argument");
  if (h->vif_k < 1) return fail("no inducing points have been set (call gpb_hip_vecchia_vif_set_inducing_points)");
  gpb_hip_vecchia_t* t = nullptr;
  int m = 0;
  const int rc = predict_factor_appended(h, n_pred, coords_pred_colmajor, num_neighbors_pred, cond_all, cov_type, var, a, &t, &m, has_duplicates, 1, false, false, false);
  struct Guard { gpb_hip_vecchia_t* p; ~Guard() { if (p) gpb_hip_vecchia_free(p); } } guard{t};
  if (rc) return -1;
  const int n_obs = h->n, n_all = n_obs + n_pred, k = h->vif_k;
  if (gpb_hip_vecchia_vif_set_inducing_points(t, k, ip_colmajor)) return -1;
  const int kp = t->vif_kp, kq = t->vif_kq;
  if (!t->d_A) {
    HIP_OK(hipMalloc(&t->d_A, sizeof(double) * (size_t)n_all * t->m));
    HIP_OK(hipMalloc(&t->d_D, sizeof(double) * (size_t)n_all));
    HIP_OK(hipMalloc(&t->d_u, sizeof(double) * (size_t)n_all));
  }
  if (vif_upload_matrix(t, 0, Linv_rowmajor, true)) return -1;
  HIP_OK(gpb::launch_vif_crosscov(cov_type, t->d_pts, t->d_ip, 0, n_all, k, kq, t->d, var, a, t->d_vC, nullptr, t->stream));
  HIP_OK(gpb::launch_vif_gemm(t->d_vC, t->d_vM, n_all, kq, t->d_V, false, t->stream));
  gpb::VecchiaKernelArgs ka;
  ka.pts = t->d_pts; ka.nn = t->d_nn; ka.exp_tab = t->d_exp_tab; ka.partials = t->d_vif_part;
  ka.A = t->d_A; ka.D = t->d_D; ka.u = t->d_u;
  ka.m = t->m; ka.i_begin = n_obs; ka.i_end = n_all;
  ka.var = var; ka.a = a; ka.diag_nn = var + 1.0; ka.diag_i = var + 1.0; ka.nugget = 1.0;
  HIP_OK(gpb::launch_vif_resid_factor(cov_type, ka, t->d_V, k, kq, t->stream));
  // (B C) of the appended rows
  HIP_OK(gpb::launch_vif_spmm(t->d_A, t->d_nn, n_obs, n_all, t->m, kq, t->d_vC, t->d_vQ, nullptr, nullptr, t->stream));
  HIP_OK(hipMemcpyAsync(u_pred, t->d_u + n_obs, sizeof(double) * (size_t)n_pred, hipMemcpyDeviceToHost, t->stream));
  HIP_OK(hipMemcpyAsync(D_pred, t->d_D + n_obs, sizeof(double) * (size_t)n_pred, hipMemcpyDeviceToHost, t->stream));
  HIP_OK(hipMemcpy2DAsync(BC_pred, sizeof(double) * (size_t)k, t->d_vQ + (size_t)n_obs * kq, sizeof(double) * (size_t)kq, sizeof(double) * (size_t)k, (size_t)n_pred,
                          hipMemcpyDeviceToHost, t->stream));
  if (nn_pred) {
    *m_used = m;
    HIP_OK(hipMemcpyAsync(nn_pred, t->d_nn + (size_t)n_obs * m, sizeof(int) * (size_t)n_pred * m, hipMemcpyDeviceToHost, t->stream));
    HIP_OK(hipMemcpyAsync(A_pred, t->d_A + (size_t)n_obs * m, sizeof(double) * (size_t)n_pred * m, hipMemcpyDeviceToHost, t->stream));
  }
  HIP_OK(hipStreamSynchronize(t->stream));
  }
  return 0;
}

// Factor rows of EVERY point of a joint (observed, prediction) ordering -- the device half of the prediction types that re-factor the
// observed points too:
//   layout 1, 'order_pred_first' (CalcPredVecchiaPredictedFirstOrder, Vecchia_utils.cpp:2228-2255, 2328-2416): prediction points first,
//     then the observed ones in Vecchia order; neighbours among all preceding points; nugget (1 / w_i for observed points) on every diagonal
//   layout 0, 'latent_order_obs_first_cond_obs_only' / '..._cond_all' (CalcPredVecchiaLatentObservedFirstOrder, :2517-2596): observed points
//     first; candidates restricted to the observed points unless cond_all; gauss_likelihood = 0: the LATENT process, no nugget, diagonal x (1 + 1e-10)
// Outputs, row-major over the n_obs + n_pred rows of the joint ordering: neighbour indices (-1 padded), A_i, D_i, u_i = (B y)_i with y = 0 at
// the prediction points.  *has_duplicates: a zero distance was met in the search (the latent types refuse that, :2563-2566).
int gpb_hip_vecchia_predict_joint_factor(gpb_hip_vecchia_t* h, int32_t n_pred, const double* coords_pred_colmajor, int32_t num_neighbors_pred,
                                         int layout_pred_first, int cond_all, int gauss_likelihood, int cov_type, double var, double a,
                                         int32_t* m_used, int32_t* nn_all, double* A_all, double* D_all, double* u_all, int* has_duplicates) {
  API_BEGIN();
  if (!m_used || !nn_all || !A_all || !D_all) return fail("null argument");
  gpb_hip_vecchia_t* t = nullptr;
  int m = 0;
  const int rc = predict_factor_appended(h, n_pred, coords_pred_colmajor, num_neighbors_pred, cond_all != 0, cov_type, var, a, &t, &m, has_duplicates,
                                         gauss_likelihood, layout_pred_first != 0, true);
  struct Guard { gpb_hip_vecchia_t* p; ~Guard() { if (p) gpb_hip_vecchia_free(p); } } guard{t};
  if (rc) return -1;
  const size_t n_all = (size_t)h->n + n_pred;
  *m_used = m;
  HIP_OK(hipMemcpy(nn_all, t->d_nn, sizeof(int) * n_all * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(A_all, t->d_A, sizeof(double) * n_all * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(D_all, t->d_D, sizeof(double) * n_all, hipMemcpyDeviceToHost));
  if (u_all) HIP_OK(hipMemcpy(u_all, t->d_u, sizeof(double) * n_all, hipMemcpyDeviceToHost));
  API_END();
}

// Dense symmetric positive definite service for the prediction types above (the conditional precision matrices the reference hands to
// its sparse Cholesky, Vecchia_utils.cpp:2419-2441, 2601-2650): x = M^-1 rhs and / or rows and columns [sub0, n) of M^-1, by the exact-GP
// machinery -- the blocked MFMA Cholesky (dense_kernels.hip); the inverse as the Schur complement of ONE partial factorisation of
// [[M, .], [I, 0]] (-M^-1 in the bottom-right block, as gpb_hip_exact_grad_terms).  M_host: row-major n x n, lower triangle significant.
// inv_sub_host: (n - sub0)^2 row-major, symmetric.  n <= 24000 with the inverse, <= 60000 without.
int gpb_hip_dense_spd_solve(int32_t n, const double* M_host, const double* rhs_host, double* x_host, int32_t sub0, double* inv_sub_host) {
  API_BEGIN();
  if (!M_host || n < 1 || (!x_host && !inv_sub_host) || (x_host && !rhs_host)) return fail("gpb_hip_dense_spd_solve: invalid argument");
  if (check_device()) return -1;
  const bool inv = inv_sub_host != nullptr;
  if (inv && (sub0 < 0 || sub0 >= n)) return fail("gpb_hip_dense_spd_solve: sub0 = %d", sub0);
  if (n > (inv ? 24000 : 60000)) return fail("gpb_hip_dense_spd_solve: n = %d is too large for the dense path (%s)", n, inv ? "inverse: 24000" : "solve: 60000");
  const int np = ((n + 63) / 64) * 64, ld = inv ? 2 * np : np;
  struct Bufs {
    double *P = nullptr, *y = nullptr, *z = nullptr, *x = nullptr, *work = nullptr, *out = nullptr; int* info = nullptr; hipStream_t st = nullptr;
    ~Bufs() { dev_free(P); dev_free(y); dev_free(z); dev_free(x); dev_free(work); dev_free(out); dev_free(info); if (st) (void)hipStreamDestroy(st); }
  } b;
  HIP_OK(hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking));
  HIP_OK(hipMalloc(&b.P, sizeof(double) * (size_t)ld * ld));
  HIP_OK(hipMalloc(&b.y, sizeof(double) * (size_t)np)); HIP_OK(hipMalloc(&b.z, sizeof(double) * (size_t)np));
  HIP_OK(hipMalloc(&b.x, sizeof(double) * (size_t)np)); HIP_OK(hipMalloc(&b.work, sizeof(double) * (size_t)np));
  HIP_OK(hipMalloc(&b.out, sizeof(double) * 2)); HIP_OK(hipMalloc(&b.info, sizeof(int)));
  HIP_OK(hipMemsetAsync(b.P, 0, sizeof(double) * (size_t)ld * ld, b.st));
  HIP_OK(hipMemsetAsync(b.info, 0, sizeof(int), b.st));
  HIP_OK(hipMemsetAsync(b.y, 0, sizeof(double) * (size_t)np, b.st));
  HIP_OK(hipMemcpy2DAsync(b.P, sizeof(double) * (size_t)ld, M_host, sizeof(double) * (size_t)n, sizeof(double) * (size_t)n, (size_t)n, hipMemcpyHostToDevice, b.st));
  if (np > n) {                                    // identity on the padding: the factor of the padded matrix is that of M
    std::vector<double> ones((size_t)(np - n), 1.0);
    HIP_OK(hipMemcpy2DAsync(b.P + (size_t)n * ld + n, sizeof(double) * ((size_t)ld + 1), ones.data(), sizeof(double), sizeof(double), (size_t)(np - n),
                            hipMemcpyHostToDevice, b.st));
    HIP_OK(hipStreamSynchronize(b.st));
  }
  if (rhs_host) HIP_OK(hipMemcpyAsync(b.y, rhs_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, b.st));
  if (inv) HIP_OK(gpb::launch_dense_aug_identity(b.P, np, ld, b.st));
  HIP_OK(gpb::launch_dense_cholesky(b.P, ld, b.info, b.st, nullptr, nullptr, nullptr, np));
  if (x_host) HIP_OK(gpb::launch_dense_solve(b.P, n, np, ld, b.y, b.z, b.out, b.x, b.st, b.work));
  int info = 0;
  HIP_OK(hipMemcpyAsync(&info, b.info, sizeof(int), hipMemcpyDeviceToHost, b.st));
  if (x_host) HIP_OK(hipMemcpyAsync(x_host, b.x, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, b.st));
  const int ns = n - sub0;
  if (inv) HIP_OK(hipMemcpy2DAsync(inv_sub_host, sizeof(double) * (size_t)ns, b.P + (size_t)(np + sub0) * ld + np + sub0, sizeof(double) * (size_t)ld,
                                   sizeof(double) * (size_t)ns, (size_t)ns, hipMemcpyDeviceToHost, b.st));
  HIP_OK(hipStreamSynchronize(b.st));
  if (info != 0) return fail("the conditional precision matrix is not positive definite (dense Cholesky failed)");
  if (inv) for (int i = 0; i < ns; ++i) {          // the Schur complement holds -M^-1 in its lower triangle
    for (int j = 0; j <= i; ++j) { const double v = -inv_sub_host[(size_t)i * ns + j]; inv_sub_host[(size_t)i * ns + j] = v; inv_sub_host[(size_t)j * ns + i] = v; }
  }
  API_END();
}

int gpb_hip_vecchia_newton_leaf_values(gpb_hip_vecchia_t* h, const int32_t* leaf_index, int32_t num_leaves, double* leaf_values) {
  API_BEGIN();
  if (!h || !leaf_index || !leaf_values) return fail("null argument");
  if (!h->has_factor || !h->has_yaux) return fail("gpb_hip_vecchia_newton_leaf_values needs the factor and y_aux of F - y (call gpb_hip_vecchia_factor and gpb_hip_vecchia_yaux first)");
  if (h->i_begin != 0 || h->i_end != h->n) return fail("gpb_hip_vecchia_newton_leaf_values needs the whole factor on this device (shard is [%d,%d))", h->i_begin, h->i_end);
  if (num_leaves < 1 || num_leaves > 64) return fail("gpb_hip_vecchia_newton_leaf_values: num_leaves = %d is outside the supported range 1..64", num_leaves);
  const int n = h->n, L = num_leaves;
  for (int i = 0; i < n; ++i)
    if (leaf_index[i] < 0 || leaf_index[i] >= L) return fail("leaf index %d at position %d is outside [0, %d)", leaf_index[i], i, L);
  HIP_OK(hipSetDevice(h->device));
  const int LP = L <= 16 ? 16 : (L <= 32 ? 32 : 64), len = LP * LP + LP;
  const size_t need = (size_t)gpb::leaf_num_workgroups(n) * len;
  if (!h->d_leaf) HIP_OK(hipMalloc(&h->d_leaf, sizeof(int) * (size_t)n));
  if (!h->d_leaf_out) HIP_OK(hipMalloc(&h->d_leaf_out, sizeof(double) * (64 * 64 + 64)));
  if (h->leaf_part_cap < need) { dev_free(h->d_leaf_part); HIP_OK(hipMalloc(&h->d_leaf_part, sizeof(double) * need)); h->leaf_part_cap = need; }
  HIP_OK(hipMemcpyAsync(h->d_leaf, leaf_index, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  HIP_OK(gpb::launch_leaf_gram(LP, h->d_A, h->d_D, h->d_nn, h->d_w, h->d_leaf, n, h->m, h->d_leaf_part, h->d_leaf_out, h->stream));
  std::vector<double> mr(len);
  HIP_OK(hipMemcpyAsync(mr.data(), h->d_leaf_out, sizeof(double) * len, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  // L x L Cholesky solve on the host (HTPsiInvH.llt().solve(HTYAux), re_model_template.h:5057)
  std::vector<double> G((size_t)L * L);
  for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) G[(size_t)i * L + j] = mr[(size_t)i * LP + j];
  const double* rhs = mr.data() + (size_t)LP * LP;
  for (int j = 0; j < L; ++j) {
    double s = G[(size_t)j * L + j];
    for (int k = 0; k < j; ++k) s -= G[(size_t)j * L + k] * G[(size_t)j * L + k];
    if (!(s > 0.)) return fail("H^T Psi^-1 H is not positive definite (leaf %d holds no data point?)", j);
    const double g = std::sqrt(s);
    G[(size_t)j * L + j] = g;
    for (int i = j + 1; i < L; ++i) {
      double t = G[(size_t)i * L + j];
      for (int k = 0; k < j; ++k) t -= G[(size_t)i * L + k] * G[(size_t)j * L + k];
      G[(size_t)i * L + j] = t / g;
    }
  }
  for (int i = 0; i < L; ++i) { double t = rhs[i]; for (int k = 0; k < i; ++k) t -= G[(size_t)i * L + k] * leaf_values[k]; leaf_values[i] = t / G[(size_t)i * L + i]; }
  for (int i = L - 1; i >= 0; --i) { double t = leaf_values[i]; for (int k = i + 1; k < L; ++k) t -= G[(size_t)k * L + i] * leaf_values[k]; leaf_values[i] = t / G[(size_t)i * L + i]; }
  API_END();
}

// The stream of the look-ahead ("REST") updates of the blocked Cholesky (dense_kernels.hip: launch_dense_cholesky).
// Round 6, measured and NOT adopted (profiles/r06_f_dense_cholesky_cu_mask_not_adopted.log): creating it with a CU mask (hipExtStreamCreateWithCUMask) that leaves 16 compute
// units to the panel chain on the main stream -- a REST update is one 128 x 128 tile per workgroup with 139 KB of LDS, ~55 us each, so its grid occupies every CU and the ~24 small
// launches of a block column's panel chain queue behind it -- made the n = 16 384 factorisation SLOWER, 65.0 against 43.9 ms (masks of 32 / 64 CUs were not honoured: 43.9 ms).
static hipError_t create_lookahead_stream(hipStream_t* out) { return hipStreamCreateWithFlags(out, hipStreamNonBlocking); }

// ------------------------------------------------------------------------------------------
int gpb_hip_exact_create(int32_t n, int32_t d, const double* coords_colmajor, gpb_hip_exact_t** out) {
  API_BEGIN();
  if (!out) return fail("gpb_hip_exact_create: out is NULL");
  *out = nullptr;
  if (check_device()) return -1;
  if (n < 1 || d < 1 || d > 3 || !coords_colmajor) return fail("gpb_hip_exact_create: invalid arguments (n = %d, d = %d; d in 1..3)", n, d);
  if (n > 100000) return fail("gpb_hip_exact_create: n = %d is too large for the dense path (use gp_approx = 'vecchia')", n);
  auto* h = new gpb_hip_exact();
  bool built = false;
  const auto drop_half_built = scope_exit([&] { if (!built) gpb_hip_exact_free(h); });
  h->n = n; h->d = d; h->np = ((n + 63) / 64) * 64;
  HIP_OK(hipGetDevice(&h->device));
  {   // the main stream carries the panel chain of the factorisation (latency-bound, the critical path); the look-ahead updates run on
      // stream2 at normal priority
    int lo = 0, hi = 0;
    HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_OK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, hi));
  }
  std::vector<double4> pts(n);
  for (int i = 0; i < n; ++i) {
    pts[i].x = coords_colmajor[i];
    pts[i].y = d > 1 ? coords_colmajor[(size_t)n + i] : 0.0;
    pts[i].z = d > 2 ? coords_colmajor[(size_t)2 * n + i] : 0.0;
    pts[i].w = 0.0;
  }
  HIP_OK(hipMalloc(&h->d_pts, sizeof(double4) * (size_t)n));
  HIP_OK(hipMemcpy(h->d_pts, pts.data(), sizeof(double4) * (size_t)n, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&h->d_P, sizeof(double) * (size_t)(h->np + 64) * (h->np + 64)));     // + the 64-row block that carries y (gpb_hip_exact_nll_terms)
  HIP_OK(hipMalloc(&h->d_y, sizeof(double) * (size_t)h->np));
  HIP_OK(hipMalloc(&h->d_z, sizeof(double) * (size_t)h->np));
  HIP_OK(hipMalloc(&h->d_x, sizeof(double) * (size_t)h->np));
  HIP_OK(hipMalloc(&h->d_work, sizeof(double) * (size_t)h->np));
  HIP_OK(hipMalloc(&h->d_out, sizeof(double) * 2));
  HIP_OK(hipMalloc(&h->d_info, sizeof(int)));
  const std::vector<double> tab = exp_table();
  HIP_OK(hipMalloc(&h->d_exp_tab, GPB_EXP_TAB_SIZE * sizeof(double)));
  HIP_OK(hipMemcpy(h->d_exp_tab, tab.data(), GPB_EXP_TAB_SIZE * sizeof(double), hipMemcpyHostToDevice));
  built = true;
  *out = h;
  API_END();
}

int gpb_hip_exact_free(gpb_hip_exact_t* h) {
  API_BEGIN();
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
  if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
  if (h->ev_panels) (void)hipEventDestroy(h->ev_panels);
  if (h->ev_rest) (void)hipEventDestroy(h->ev_rest);
  dev_free(h->d_pts); dev_free(h->d_P); dev_free(h->d_y); dev_free(h->d_z); dev_free(h->d_x); dev_free(h->d_out); dev_free(h->d_work);
  dev_free(h->d_exp_tab); dev_free(h->d_info); dev_free(h->d_P2); dev_free(h->d_gpart); dev_free(h->d_g4);
  delete h;
  API_END();
}

int gpb_hip_exact_set_y(gpb_hip_exact_t* h, const double* y_host) {
  API_BEGIN();
  if (!h || !y_host) return fail("null argument");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipMemcpyAsync(h->d_y, y_host, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->has_y = true;
  API_END();
}

int gpb_hip_exact_nll_terms(gpb_hip_exact_t* h, int cov_type, double var, double a, double* out2_host, double* yaux_host,
                            double* ms3) {
  API_BEGIN();
  if (!h || !out2_host) return fail("null argument");
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_exact_set_y)");
  if (cov_type < 0 || cov_type > 2) return fail("covariance type %d is not on the HIP hot path (Matern 0.5/1.5/2.5 only)", cov_type);
  if (!(var > 0.) || !(a > 0.)) return fail("covariance parameters must be positive (var = %g, range = %g)", var, a);
  HIP_OK(hipSetDevice(h->device));
  hipEvent_t e[4];
  for (auto& ev : e) HIP_OK(hipEventCreate(&ev));
  HIP_OK(hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  HIP_OK(hipEventRecord(e[0], h->stream));
  // Two forms.  yrow (default): y rides along as row np of the (np + 64)-row matrix -- the panel solves of the factorisation leave
  // z = L^-1 y in that row and the trailing update leaves -z'z at [np][np]: no forward substitution (np / 64 launches less; the backward
  // one only when y_aux is asked for).  (Round 2's form -- the np x np matrix and both substitutions -- was removed in round 4.)
  const int ld = h->np + 64;
  HIP_OK(hipMemsetAsync(h->d_P + (size_t)h->np * ld, 0, sizeof(double) * (size_t)64 * ld, h->stream));
  HIP_OK(gpb::launch_dense_cov(cov_type, h->d == 3, h->d_pts, h->n, h->np, ld, var, a, 1.0, h->d_exp_tab, h->d_P, h->stream));   // Psi = Sigma + I (:9273-9287)
  HIP_OK(gpb::launch_dense_set_yrow(h->d_P, h->n, h->np, ld, h->d_y, h->stream));
  HIP_OK(hipEventRecord(e[1], h->stream));
  if (!h->stream2) {
    HIP_OK(create_lookahead_stream(&h->stream2));
    HIP_OK(hipEventCreateWithFlags(&h->ev_panels, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&h->ev_rest, hipEventDisableTiming));
  }
  HIP_OK(gpb::launch_dense_cholesky(h->d_P, ld, h->d_info, h->stream, h->stream2, h->ev_panels, h->ev_rest, h->np));
  HIP_OK(hipEventRecord(e[2], h->stream));
  HIP_OK(gpb::launch_dense_yrow_sums(h->d_P, h->n, h->np, ld, h->d_out, h->stream));
  if (yaux_host) {
    HIP_OK(hipMemcpyAsync(h->d_work, h->d_P + (size_t)h->np * ld, sizeof(double) * (size_t)h->np, hipMemcpyDeviceToDevice, h->stream));
    HIP_OK(gpb::launch_dense_solve_backward(h->d_P, h->np, ld, h->d_work, h->d_x, h->stream));
  }
  HIP_OK(hipEventRecord(e[3], h->stream));
  int info = 0;
  HIP_OK(hipMemcpyAsync(out2_host, h->d_out, sizeof(double) * 2, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (yaux_host) HIP_OK(hipMemcpyAsync(yaux_host, h->d_x, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (ms3) for (int t = 0; t < 3; ++t) { float ms = 0.f; HIP_OK(hipEventElapsedTime(&ms, e[t], e[t + 1])); ms3[t] = ms; }
  for (auto& ev : e) (void)hipEventDestroy(ev);
  if (info != 0) return fail("the covariance matrix is not positive definite (dense Cholesky failed)");
  API_END();
}

/* Likelihood terms AND the covariance-parameter gradient sums of the exact (dense) GP: CalcGradPars, dense branch
   (re_model_template.h:2016-2040) with CalcPsiInv (:6586-6614).  Psi^-1 comes out of ONE partial factorisation: the blocked Cholesky run
   on the first np columns of [[Psi, .], [I, 0]] leaves L in the top-left block, L^-T below it and -L^-T L^-1 = -Psi^-1 as the Schur
   complement in the bottom-right block (the same MFMA trailing updates as the factorisation itself).  out7 has the layout of
   gpb_hip_vecchia_grad_terms: {y' Psi^-1 y, log|Psi|, 0, g1_var, g2_var, g1_range, g2_range}, gradient_k = g1_k / sigma2 + g2_k. */
int gpb_hip_exact_grad_terms(gpb_hip_exact_t* h, int cov_type, double var, double a, double* out7_host) {
  API_BEGIN();
  if (!h || !out7_host) return fail("null argument");
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_exact_set_y)");
  if (cov_type < 0 || cov_type > 2) return fail("covariance type %d is not on the HIP hot path (Matern 0.5/1.5/2.5 only)", cov_type);
  if (!(var > 0.) || !(a > 0.)) return fail("covariance parameters must be positive (var = %g, range = %g)", var, a);
  if (h->n > 24000) return fail("gpb_hip_exact_grad_terms: n = %d is too large for the dense gradient (the augmented matrix has (2 n)^2 entries); use gp_approx = 'vecchia'", h->n);
  HIP_OK(hipSetDevice(h->device));
  const int np = h->np, ld = 2 * np, ntiles = gpb::dense_grad_num_tiles(np);
  if (!h->d_P2) {
    HIP_OK(hipMalloc(&h->d_P2, sizeof(double) * (size_t)ld * ld));
    HIP_OK(hipMalloc(&h->d_gpart, sizeof(double) * 4 * (size_t)ntiles));
    HIP_OK(hipMalloc(&h->d_g4, sizeof(double) * 8));
  }
  if (!h->stream2) {
    HIP_OK(create_lookahead_stream(&h->stream2));
    HIP_OK(hipEventCreateWithFlags(&h->ev_panels, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&h->ev_rest, hipEventDisableTiming));
  }
  HIP_OK(hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  HIP_OK(hipMemsetAsync(h->d_P2, 0, sizeof(double) * (size_t)ld * ld, h->stream));
  HIP_OK(gpb::launch_dense_cov(cov_type, h->d == 3, h->d_pts, h->n, np, ld, var, a, 1.0, h->d_exp_tab, h->d_P2, h->stream));
  HIP_OK(gpb::launch_dense_aug_identity(h->d_P2, np, ld, h->stream));
  HIP_OK(gpb::launch_dense_cholesky(h->d_P2, ld, h->d_info, h->stream, h->stream2, h->ev_panels, h->ev_rest, np));
  HIP_OK(gpb::launch_dense_solve(h->d_P2, h->n, np, ld, h->d_y, h->d_z, h->d_out, h->d_x, h->stream, h->d_work));           // y' Psi^-1 y, log|Psi|, ya = Psi^-1 y
  HIP_OK(gpb::launch_dense_grad(cov_type, h->d == 3, h->d_pts, h->n, np, ld, var, a, h->d_exp_tab, h->d_P2, h->d_x, h->d_gpart, h->stream));
  HIP_OK(gpb::launch_reduce_partials(h->d_gpart, ntiles, 4, h->d_g4, nullptr, h->stream));
  double o2[2], g4[4];
  int info = 0;
  HIP_OK(hipMemcpyAsync(o2, h->d_out, sizeof(double) * 2, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(g4, h->d_g4, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (info != 0) return fail("the covariance matrix is not positive definite (dense Cholesky failed)");
  // g4 = {S.yy, S.W, dS.yy, dS.W} with W = -Psi^-1:  -1/2 ya' dPsi ya  and  1/2 tr(Psi^-1 dPsi) = -1/2 sum dPsi o W
  out7_host[0] = o2[0]; out7_host[1] = o2[1]; out7_host[2] = 0.;
  out7_host[3] = -0.5 * g4[0]; out7_host[4] = -0.5 * g4[1];
  out7_host[5] = -0.5 * g4[2]; out7_host[6] = -0.5 * g4[3];
  API_END();
}

/* diag(Psi^-1) of the exact GP on the transformed scale: the predictive variances of the training-data random effects are sigma2 (1 - diag)
   (PredictTrainingDataRandomEffects, dense branch: Cov[b | y] = Sigma - Sigma Psi^-1 Sigma = sigma2 (I - Psi_t^-1) with Sigma_t = Psi_t - I).  -Psi^-1 is the
   Schur complement of the partial factorisation of [[Psi, .], [I, 0]] (as gpb_hip_exact_grad_terms); its diagonal is copied out.  n <= 24000. */
int gpb_hip_exact_psi_inv_diag(gpb_hip_exact_t* h, int cov_type, double var, double a, double* diag_host) {
  API_BEGIN();
  if (!h || !diag_host) return fail("null argument");
  if (cov_type < 0 || cov_type > 2) return fail("covariance type %d is not on the HIP hot path (Matern 0.5/1.5/2.5 only)", cov_type);
  if (!(var > 0.) || !(a > 0.)) return fail("covariance parameters must be positive (var = %g, range = %g)", var, a);
  if (h->n > 24000) return fail("gpb_hip_exact_psi_inv_diag: n = %d is too large for the dense inverse (the augmented matrix has (2 n)^2 entries)", h->n);
  HIP_OK(hipSetDevice(h->device));
  const int np = h->np, ld = 2 * np;
  if (!h->d_P2) {
    const int ntiles = gpb::dense_grad_num_tiles(np);
    HIP_OK(hipMalloc(&h->d_P2, sizeof(double) * (size_t)ld * ld));
    HIP_OK(hipMalloc(&h->d_gpart, sizeof(double) * 4 * (size_t)ntiles));
    HIP_OK(hipMalloc(&h->d_g4, sizeof(double) * 8));
  }
  if (!h->stream2) {
    HIP_OK(create_lookahead_stream(&h->stream2));
    HIP_OK(hipEventCreateWithFlags(&h->ev_panels, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&h->ev_rest, hipEventDisableTiming));
  }
  HIP_OK(hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  HIP_OK(hipMemsetAsync(h->d_P2, 0, sizeof(double) * (size_t)ld * ld, h->stream));
  HIP_OK(gpb::launch_dense_cov(cov_type, h->d == 3, h->d_pts, h->n, np, ld, var, a, 1.0, h->d_exp_tab, h->d_P2, h->stream));
  HIP_OK(gpb::launch_dense_aug_identity(h->d_P2, np, ld, h->stream));
  HIP_OK(gpb::launch_dense_cholesky(h->d_P2, ld, h->d_info, h->stream, h->stream2, h->ev_panels, h->ev_rest, np));
  int info = 0;
  HIP_OK(hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpy2DAsync(diag_host, sizeof(double), h->d_P2 + (size_t)np * ld + np, sizeof(double) * ((size_t)ld + 1), sizeof(double), (size_t)h->n,
                          hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (info != 0) return fail("the covariance matrix is not positive definite (dense Cholesky failed)");
  for (int i = 0; i < h->n; ++i) diag_host[i] = -diag_host[i];
  API_END();
}

/* Prediction of the exact GP at new locations (REModelTemplate::Predict, dense Gaussian branch: mean = Sigma_po Psi^-1 y, covariance =
   Sigma_pp - Sigma_po Psi^-1 Sigma_op; re_model_template.h:4239-4330 with CalcPred).  Transformed scale (Psi = Sigma / sigma2 + I): ONE
   partial factorisation (first np columns) of [[Psi, ., .], [C, 0, .], [y', 0, 0]], C = Sigma_po / sigma2, leaves
     mean_out = C Psi^-1 y  (= -Schur[y row][C columns])      q_out = C Psi^-1 C'  (= -Schur[C rows][C columns], n_pred^2 row-major, may be NULL)
   The caller adds the prior covariance of the prediction points and the scale sigma2.  n_pred <= 20000. */
int gpb_hip_exact_predict(gpb_hip_exact_t* h, int cov_type, double var, double a, int32_t n_pred, const double* coords_pred_colmajor,
                          double* mean_out, double* q_out) {
  API_BEGIN();
  if (!h || !coords_pred_colmajor || !mean_out) return fail("null argument");
  if (!h->has_y) return fail("response data has not been set (call gpb_hip_exact_set_y)");
  if (cov_type < 0 || cov_type > 2) return fail("covariance type %d is not on the HIP hot path (Matern 0.5/1.5/2.5 only)", cov_type);
  if (!(var > 0.) || !(a > 0.)) return fail("covariance parameters must be positive (var = %g, range = %g)", var, a);
  if (n_pred < 1 || n_pred > 20000) return fail("gpb_hip_exact_predict: n_pred = %d (1..20000)", n_pred);
  HIP_OK(hipSetDevice(h->device));
  const int n = h->n, np = h->np, npp = ((n_pred + 63) / 64) * 64, ld = np + npp + 64, yrow = np + npp;
  struct Bufs { double* P = nullptr; double4* pred = nullptr; ~Bufs() { dev_free(P); dev_free(pred); } } b;
  HIP_OK(hipMalloc(&b.P, sizeof(double) * (size_t)ld * ld));
  HIP_OK(hipMalloc(&b.pred, sizeof(double4) * (size_t)n_pred));
  std::vector<double4> pp(n_pred);
  for (int i = 0; i < n_pred; ++i) {
    pp[i].x = coords_pred_colmajor[i];
    pp[i].y = h->d > 1 ? coords_pred_colmajor[(size_t)n_pred + i] : 0.0;
    pp[i].z = h->d > 2 ? coords_pred_colmajor[(size_t)2 * n_pred + i] : 0.0;
    pp[i].w = 0.0;
  }
  HIP_OK(hipMemcpyAsync(b.pred, pp.data(), sizeof(double4) * (size_t)n_pred, hipMemcpyHostToDevice, h->stream));
  if (!h->stream2) {
    HIP_OK(create_lookahead_stream(&h->stream2));
    HIP_OK(hipEventCreateWithFlags(&h->ev_panels, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&h->ev_rest, hipEventDisableTiming));
  }
  HIP_OK(hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  HIP_OK(hipMemsetAsync(b.P, 0, sizeof(double) * (size_t)ld * ld, h->stream));
  HIP_OK(gpb::launch_dense_cov(cov_type, h->d == 3, h->d_pts, n, np, ld, var, a, 1.0, h->d_exp_tab, b.P, h->stream));
  HIP_OK(gpb::launch_dense_cross_cov(cov_type, h->d == 3, h->d_pts, n, b.pred, n_pred, ld, var, a, h->d_exp_tab, b.P, np, h->stream));
  HIP_OK(gpb::launch_dense_set_row(b.P, n, ld, yrow, h->d_y, h->stream));
  HIP_OK(gpb::launch_dense_cholesky(b.P, ld, h->d_info, h->stream, h->stream2, h->ev_panels, h->ev_rest, np));
  int info = 0;
  HIP_OK(hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(mean_out, b.P + (size_t)yrow * ld + np, sizeof(double) * (size_t)n_pred, hipMemcpyDeviceToHost, h->stream));
  if (q_out) HIP_OK(hipMemcpy2DAsync(q_out, sizeof(double) * (size_t)n_pred, b.P + (size_t)np * ld + np, sizeof(double) * (size_t)ld,
                                     sizeof(double) * (size_t)n_pred, (size_t)n_pred, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (info != 0) return fail("the covariance matrix is not positive definite (dense Cholesky failed)");
  for (int i = 0; i < n_pred; ++i) mean_out[i] = -mean_out[i];
  if (q_out) for (int i = 0; i < n_pred; ++i)            // the Schur complement holds -C Psi^-1 C' in its lower triangle
    for (int j = 0; j <= i; ++j) { const double v = -q_out[(size_t)i * n_pred + j]; q_out[(size_t)i * n_pred + j] = v; q_out[(size_t)j * n_pred + i] = v; }
  API_END();
}

/* Standard errors of (sigma2, sigma1_2, rho) of the exact GP from the Fisher information on the ORIGINAL scale: CalcStdDevCovPar ->
   CalcFisherInformation, dense branch with include_error_var, !transf_scale (re_model_template.h:10788-10815, 10066-10127):
   FI_ab = 1/2 tr(P dPsi_a P dPsi_b), P = (sigma2 Psi)^-1, dPsi = {I, Sigma / sigma1_2, d(Sigma)/d rho}.  All six traces come out of ONE partial
   factorisation of a 4 np x 4 np augmented matrix (dense_kernels.hip); with E1 = Sigma_t = ratio k, E2 = dSigma_t / dlog(a) on the transformed
   scale (ratio = sigma1_2 / sigma2, d/d rho = -(1 / rho) d/dlog a) and T_ab the traces over Psi_t = Sigma_t + I:
     FI_00 = T_00 / (2 s^4)          FI_01 = T_10 / (2 s^4 ratio)         FI_02 = -T_20 / (2 s^2 rho)      (s^2 = sigma2)
     FI_11 = T_11 / (2 s^4 ratio^2)  FI_12 = -T_21 / (2 s^2 rho ratio)    FI_22 = T_22 / (2 rho^2)
   se = sqrt(diag(FI^-1)), NaN where FI is not positive definite (as the reference).  n <= 24000 ((4 n)^2 doubles of device memory). */
int gpb_hip_exact_fisher_std_errors(gpb_hip_exact_t* h, int cov_type, double sigma2, double ratio, double a, double rho, double* se3_host) {
  API_BEGIN();
  if (!h || !se3_host) return fail("null argument");
  if (cov_type < 0 || cov_type > 2) return fail("covariance type %d is not on the HIP hot path (Matern 0.5/1.5/2.5 only)", cov_type);
  if (!(sigma2 > 0.) || !(ratio > 0.) || !(a > 0.) || !(rho > 0.)) return fail("covariance parameters must be positive");
  if (h->n > 24000) return fail("gpb_hip_exact_fisher_std_errors: n = %d is too large for the dense Fisher information (the augmented matrix has (4 n)^2 entries); use gp_approx = 'vecchia'", h->n);
  HIP_OK(hipSetDevice(h->device));
  const int np = h->np, ld = 4 * np, ntiles = gpb::dense_grad_num_tiles(np);
  struct Bufs { double *P = nullptr, *part = nullptr, *t6 = nullptr; ~Bufs() { dev_free(P); dev_free(part); dev_free(t6); } } b;
  HIP_OK(hipMalloc(&b.P, sizeof(double) * (size_t)ld * ld));
  HIP_OK(hipMalloc(&b.part, sizeof(double) * 6 * (size_t)ntiles));
  HIP_OK(hipMalloc(&b.t6, sizeof(double) * 8));
  if (!h->stream2) {
    HIP_OK(create_lookahead_stream(&h->stream2));
    HIP_OK(hipEventCreateWithFlags(&h->ev_panels, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&h->ev_rest, hipEventDisableTiming));
  }
  HIP_OK(hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  HIP_OK(hipMemsetAsync(b.P, 0, sizeof(double) * (size_t)ld * ld, h->stream));
  HIP_OK(gpb::launch_dense_cov(cov_type, h->d == 3, h->d_pts, h->n, np, ld, ratio, a, 1.0, h->d_exp_tab, b.P, h->stream));
  HIP_OK(gpb::launch_dense_aug_identity(b.P, np, ld, h->stream));
  HIP_OK(gpb::launch_dense_deriv_blocks(cov_type, h->d == 3, h->d_pts, h->n, ld, ratio, a, h->d_exp_tab, b.P, 2 * np, 3 * np, h->stream));
  HIP_OK(gpb::launch_dense_cholesky(b.P, ld, h->d_info, h->stream, h->stream2, h->ev_panels, h->ev_rest, np));
  HIP_OK(gpb::launch_dense_fisher_sums(b.P, h->n, np, ld, b.part, h->stream));
  HIP_OK(gpb::launch_reduce_partials(b.part, ntiles, 6, b.t6, nullptr, h->stream));
  double T[6];
  int info = 0;
  HIP_OK(hipMemcpyAsync(T, b.t6, sizeof(double) * 6, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (info != 0) return fail("the covariance matrix is not positive definite (dense Cholesky failed)");
  const double s2 = sigma2, s4 = sigma2 * sigma2;
  double FI[3][3];
  FI[0][0] = T[0] / (2. * s4);
  FI[0][1] = FI[1][0] = T[1] / (2. * s4 * ratio);
  FI[0][2] = FI[2][0] = -T[2] / (2. * s2 * rho);
  FI[1][1] = T[3] / (2. * s4 * ratio * ratio);
  FI[1][2] = FI[2][1] = -T[4] / (2. * s2 * rho * ratio);
  FI[2][2] = T[5] / (2. * rho * rho);
  // sqrt(diag(FI^-1)) through the Cholesky factor (Eigen::LLT in the reference); NaN if it fails
  const double nan = std::numeric_limits<double>::quiet_NaN();
  for (int j = 0; j < 3; ++j) se3_host[j] = nan;
  double L[3][3] = {{0}};
  bool ok = true;
  for (int i = 0; i < 3 && ok; ++i) for (int j = 0; j <= i; ++j) {
    double v = FI[i][j];
    for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
    if (i == j) { if (!(v > 0.)) { ok = false; break; } L[i][i] = std::sqrt(v); } else L[i][j] = v / L[j][j];
  }
  if (ok) for (int c = 0; c < 3; ++c) {          // (FI^-1)_cc = || L^-1 e_c ||^2
    double z[3] = {0., 0., 0.}, ss = 0.;
    for (int i = c; i < 3; ++i) { double v = (i == c) ? 1. : 0.; for (int k = c; k < i; ++k) v -= L[i][k] * z[k]; z[i] = v / L[i][i]; ss += z[i] * z[i]; }
    if (std::isfinite(ss) && ss >= 0.) se3_host[c] = std::sqrt(ss);
  }
  API_END();
}

// ------------------------------------------------------------------------------------------
int gpb_hip_hist_create(int32_t n, int32_t num_features, const uint8_t* bins, const int32_t* bin_offsets,
                        gpb_hip_hist_t** out) {
  API_BEGIN();
  if (!out) return fail("gpb_hip_hist_create: out is NULL");
  *out = nullptr;
  if (check_device()) return -1;
  if (n < 1 || num_features < 1 || !bins || !bin_offsets) return fail("gpb_hip_hist_create: invalid arguments");
  for (int f = 0; f < num_features; ++f) {
    const int nb = bin_offsets[f + 1] - bin_offsets[f];
    if (nb < 1 || nb > GPB_HIST_MAX_BIN) return fail("gpb_hip_hist_create: feature %d has %d bins (1..256 supported)", f, nb);
  }
  auto* h = new gpb_hip_hist();
  bool built = false;
  const auto drop_half_built = scope_exit([&] { if (!built) gpb_hip_hist_free(h); });
  h->n = n; h->F = num_features; h->fpad = ((num_features + 15) / 16) * 16; h->total_bins = bin_offsets[num_features];
  HIP_OK(hipGetDevice(&h->device));
  HIP_OK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  uint8_t* d_fm = nullptr;
  const auto free_tmp = scope_exit([&] { (void)hipFree(d_fm); });
  HIP_OK(hipMalloc(&d_fm, (size_t)n * num_features));
  HIP_OK(hipMemcpy(d_fm, bins, (size_t)n * num_features, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&h->d_bins_rm, (size_t)n * h->fpad));
  HIP_OK(gpb::launch_bins_transpose(d_fm, h->d_bins_rm, n, num_features, h->fpad, h->stream));
  {
    // the compact copy the whole-row kernel streams when it visits EVERY row (root pass of a tree / gpb_hip_hist_build without an index list): F rounded up to
    // 4 bytes per row instead of fpad -- at F = 50: 52 instead of 64 (the padded rows were 1.24x the algorithmic bytes in the counters)
    const int rs = ((num_features + 3) / 4) * 4;
    const char* off = getenv("GPB_HIST_NO_COMPACT_ROWS");
    if (rs < h->fpad && h->fpad / GPB_HIST_FG >= 4 && !(off && off[0] == '1')) {
      HIP_OK(hipMalloc(&h->d_bins_cm, (size_t)n * rs + 16));
      HIP_OK(hipMemsetAsync(h->d_bins_cm + (size_t)n * rs, 0, 16, h->stream));
      HIP_OK(gpb::launch_bins_transpose(d_fm, h->d_bins_cm, n, num_features, h->fpad, h->stream, rs));
      h->rstride = rs;
    }
  }
  HIP_OK(hipStreamSynchronize(h->stream));
  (void)hipFree(d_fm); d_fm = nullptr;
  HIP_OK(hipMalloc(&h->d_bin_offsets, sizeof(int) * (size_t)(num_features + 1)));
  HIP_OK(hipMemcpy(h->d_bin_offsets, bin_offsets, sizeof(int) * (size_t)(num_features + 1), hipMemcpyHostToDevice));
  h->h_bin_offsets.assign(bin_offsets, bin_offsets + num_features + 1);
  HIP_OK(hipMalloc(&h->d_grad, sizeof(double) * (size_t)n));
  HIP_OK(hipMalloc(&h->d_hess, sizeof(double) * (size_t)n));
  HIP_OK(hipMalloc(&h->d_hist, sizeof(double) * 2 * (size_t)h->total_bins));
  HIP_OK(hipMalloc(&h->d_cnt, sizeof(unsigned long long) * (size_t)h->total_bins));
  built = true;
  *out = h;
  API_END();
}

// Page-locking of caller buffers is OPT-IN (gpb_hip_hist_register_host_buffers; ADVICE r03: the library used to register whatever pointer
// set_gradients / grow_tree saw and keep the registration after the call returned -- undefined once the caller frees a temporary).  A caller
// that hands over the SAME arrays every iteration (the reference's Booster through route B) registers them once and owns the lifetime contract:
// the arrays stay allocated until gpb_hip_hist_unregister_host_buffers / gpb_hip_hist_free.  slot: 0 gradients, 1 hessians, 2 leaf index.
static void hist_pin(gpb_hip_hist_t* h, int slot, const void* p, size_t bytes) {
  gpb_hip_hist::Pinned& q = h->pin[slot];
  if (q.p == p && q.bytes >= bytes) return;
  if (q.p) { (void)hipHostUnregister(const_cast<void*>(q.p)); q.p = nullptr; q.bytes = 0; }
  if (!p || bytes < (1u << 20)) return;                    // small arrays: not worth a registration
  if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess) { q.p = p; q.bytes = bytes; }
  else (void)hipGetLastError();                            // harmless: the copies then take the pageable path
}
int gpb_hip_hist_register_host_buffers(gpb_hip_hist_t* h, const double* grad, const double* hess, const int32_t* data_leaf_index) {
  API_BEGIN();
  if (!h) return fail("null handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  hist_pin(h, 0, grad, sizeof(double) * (size_t)h->n);
  hist_pin(h, 1, hess, sizeof(double) * (size_t)h->n);
  hist_pin(h, 2, data_leaf_index, sizeof(int) * (size_t)h->n);
  API_END();
}
int gpb_hip_hist_unregister_host_buffers(gpb_hip_hist_t* h) {
  API_BEGIN();
  if (!h) return 0;
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  for (auto& q : h->pin) if (q.p) { (void)hipHostUnregister(const_cast<void*>(q.p)); q.p = nullptr; q.bytes = 0; }
  API_END();
}

int gpb_hip_hist_free(gpb_hip_hist_t* h) {
  API_BEGIN();
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  // copies from the caller-registered staging buffers may still be in flight: drain the stream BEFORE they lose their registration (same order as
  // gpb_hip_hist_unregister_host_buffers; the caller may destroy the buffers right after this call)
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& q : h->pin) if (q.p) { (void)hipHostUnregister(const_cast<void*>(q.p)); q.p = nullptr; }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  dev_free(h->d_bins_rm); dev_free(h->d_bins_cm); dev_free(h->d_bin_offsets); dev_free(h->d_grad); dev_free(h->d_hess); dev_free(h->d_idx);
  dev_free(h->d_part_grad); dev_free(h->d_part_hess); dev_free(h->d_part_cnt); dev_free(h->d_hist); dev_free(h->d_cnt); dev_free(h->d_absmax);
  dev_free(h->d_pool); dev_free(h->d_fix); dev_free(h->d_meta3); dev_free(h->d_part); dev_free(h->d_split); dev_free(h->d_split_i); dev_free(h->d_used);
  dev_free(h->d_tree_red); dev_free(h->d_rows); dev_free(h->d_rows2); dev_free(h->d_counts); dev_free(h->d_root_rows);
  if (h->h_counts) (void)hipHostFree(h->h_counts); dev_free(h->d_ptags); dev_free(h->d_split2); dev_free(h->d_split2_i); dev_free(h->d_used2);
  if (h->h_split2) (void)hipHostFree(h->h_split2);
  if (h->h_split2_i) (void)hipHostFree(h->h_split2_i);
  dev_free(h->d_is_cat); dev_free(h->d_cat_bits); dev_free(h->d_cat_bits2); dev_free(h->d_rs_send); dev_free(h->d_rs_recv); dev_free(h->d_xchg);
  if (h->h_xchg) (void)hipHostFree(h->h_xchg);
  if (h->h_cat_bits2) (void)hipHostFree(h->h_cat_bits2);
  h->comm.release(); dev_free(h->d_limbs);
  delete h;
  API_END();
}

/* Categorical features (round 5): is_categorical[f] != 0 -> feature f is searched by FindBestThresholdCategoricalInner (feature_histogram.hpp:278-519)
 * instead of the threshold scans, and its splits are sets of bins (DenseBin::SplitCategorical).  NULL flags: every feature numerical again. */
int gpb_hip_hist_set_categorical(gpb_hip_hist_t* h, const int8_t* is_categorical, int32_t max_cat_to_onehot, int32_t max_cat_threshold, double cat_smooth,
                                 double cat_l2, int32_t min_data_per_group) {
  API_BEGIN();
  if (!h) return fail("null argument");
  HIP_OK(hipSetDevice(h->device));
  h->any_cat = false;
  h->h_is_cat.assign((size_t)h->F, 0);
  if (is_categorical) for (int f = 0; f < h->F; ++f) { h->h_is_cat[f] = is_categorical[f] ? 1 : 0; h->any_cat = h->any_cat || is_categorical[f]; }
  if (!h->any_cat) return 0;
  if (max_cat_to_onehot < 0 || max_cat_threshold < 1 || !(cat_smooth >= 0.0) || !(cat_l2 >= 0.0) || min_data_per_group < 0)
    return fail("gpb_hip_hist_set_categorical: max_cat_to_onehot %d, max_cat_threshold %d, cat_smooth %g, cat_l2 %g, min_data_per_group %d", max_cat_to_onehot,
                max_cat_threshold, cat_smooth, cat_l2, min_data_per_group);
  h->cat.max_cat_to_onehot = max_cat_to_onehot; h->cat.max_cat_threshold = max_cat_threshold; h->cat.cat_smooth = cat_smooth; h->cat.cat_l2 = cat_l2;
  h->cat.min_data_per_group = min_data_per_group;
  if (!h->d_is_cat) HIP_OK(hipMalloc(&h->d_is_cat, (size_t)h->F));
  HIP_OK(hipMemcpy(h->d_is_cat, h->h_is_cat.data(), (size_t)h->F, hipMemcpyHostToDevice));
  if (!h->d_cat_bits) {
    HIP_OK(hipMalloc(&h->d_cat_bits, sizeof(unsigned) * (size_t)h->F * 8)); HIP_OK(hipMemset(h->d_cat_bits, 0, sizeof(unsigned) * (size_t)h->F * 8));
    HIP_OK(hipMalloc(&h->d_cat_bits2, sizeof(unsigned) * (size_t)h->F * 16)); HIP_OK(hipMemset(h->d_cat_bits2, 0, sizeof(unsigned) * (size_t)h->F * 16));
    HIP_OK(hipHostMalloc(&h->h_cat_bits2, sizeof(unsigned) * (size_t)h->F * 16)); std::memset(h->h_cat_bits2, 0, sizeof(unsigned) * (size_t)h->F * 16);
  }
  API_END();
}

int gpb_hip_hist_set_gradients(gpb_hip_hist_t* h, const double* grad, const double* hess) {
  API_BEGIN();
  if (!h || !grad) return fail("null argument");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipMemcpyAsync(h->d_grad, grad, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  if (hess) HIP_OK(hipMemcpyAsync(h->d_hess, hess, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  if (!h->d_absmax) HIP_OK(hipMalloc(&h->d_absmax, 2 * sizeof(unsigned long long)));
  HIP_OK(gpb::launch_hist_absmax(h->d_grad, h->n, h->d_absmax, h->stream));
  if (hess) HIP_OK(gpb::launch_hist_absmax(h->d_hess, h->n, h->d_absmax + 1, h->stream));
  else HIP_OK(hipMemsetAsync(h->d_absmax + 1, 0, sizeof(unsigned long long), h->stream));
  // sharded rows: ONE scale for all ranks -- the max over the ranks of the IEEE bit patterns (monotone for non-negative doubles) -- or every
  // rank would round its gradients to a different q and the sums would depend on how the rows were dealt to ranks
  if (h->comm.active() && comm_allreduce(h->comm, h->d_absmax, 2, GPB_T_U64, GPB_OP_MAX, h->stream)) return -1;
  HIP_OK(hipStreamSynchronize(h->stream));
  h->has_hess = hess != nullptr; h->has_grad = true;
  API_END();
}

static int hist_build_impl(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t num_data, double const_hess,
                           double* hist_out, uint64_t* cnt_out, int reps, double* ms_avg, double* d_target = nullptr,
                           const int* dev_indices = nullptr, bool dev_all_rows = false);

// Sharded handle (DataParallelTreeLearner's scheme, data_parallel_tree_learner.cpp:155-173, with integers on the wire): the reduce kernel has
// left this rank's INTEGER totals in d_limbs (word-major); one all-reduce(sum, int64) of 3 words per bin -- {grad hi, grad lo, count}: 24 bytes, the
// constant-hessian case of the GPBoost algorithm with a Gaussian likelihood -- or 5 with per-row hessians, then the same conversion as on one GPU.
// (Round 4 sent 5 words per bin in every case: 510 KB per leaf at config 3; now 306 KB.  The reference's reduce-scatter moves 16 bytes per bin of
// DOUBLES, data_parallel_tree_learner.cpp:155-173 -- its sums then depend on the rank layout, these do not.)
// Counts are exact, and so are the sums of the once-rounded gradients: the job's histogram is bit-identical for every rank layout.
static int hist_finish_sharded(gpb_hip_hist_t* h, double const_hess, double* d_hist_out, unsigned long long* d_cnt_out) {
  if (comm_allreduce(h->comm, h->d_limbs, (h->has_hess ? 5 : 3) * (size_t)h->total_bins, GPB_T_I64, GPB_OP_SUM, h->stream)) return -1;
  HIP_OK(gpb::launch_hist_convert(h->d_limbs, h->total_bins, h->d_absmax, h->d_absmax + 1, const_hess, h->has_hess ? 1 : 0, d_hist_out, d_cnt_out, h->stream));
  return 0;
}

// Feature-block form (the tree grower's smaller-child builds): d_limbs -> blocks -> ONE reduce-scatter -> this rank's block of the histogram.  The blocks are
// contiguous runs of features with about total_bins / world bins each (whole features: a feature's bins are searched by one rank).
static int hist_blocks_setup(gpb_hip_hist_t* h) {
  const int W = h->comm.world, F = h->F;
  if ((int)h->blk_f0.size() == W + 1) return 0;
  if (W > 16) return fail("feature-block exchange: %d ranks (at most 16)", W);
  h->blk_f0.assign(W + 1, F); h->blk_bin0.assign(W, 0); h->blk_bins.assign(W, 0);
  int f = 0;
  for (int r = 0; r < W; ++r) {
    h->blk_f0[r] = f;
    if (r == W - 1) { f = F; break; }
    const long long target = (long long)h->total_bins * (r + 1) / W;          // whole features whose bins end within the rank's share (a block may be empty)
    while (f < F && h->h_bin_offsets[f + 1] <= target) ++f;
  }
  h->blk_f0[W] = F;
  h->blk_max = 1;
  for (int r = 0; r < W; ++r) {
    h->blk_bin0[r] = h->h_bin_offsets[h->blk_f0[r]];
    h->blk_bins[r] = h->h_bin_offsets[h->blk_f0[r + 1]] - h->blk_bin0[r];
    h->blk_max = std::max(h->blk_max, h->blk_bins[r]);
  }
  dev_free(h->d_rs_send); dev_free(h->d_rs_recv);
  HIP_OK(hipMalloc(&h->d_rs_send, sizeof(long long) * 5 * (size_t)h->blk_max * W));
  HIP_OK(hipMalloc(&h->d_rs_recv, sizeof(long long) * 5 * (size_t)h->blk_max));
  if (!h->d_xchg) { HIP_OK(hipMalloc(&h->d_xchg, sizeof(double) * 2 * 16 * 24)); HIP_OK(hipHostMalloc(&h->h_xchg, sizeof(double) * 2 * 16 * 24)); }
  return 0;
}
static int hist_finish_sharded_block(gpb_hip_hist_t* h, double const_hess, double* d_hist_out) {
  const int nw = h->has_hess ? 5 : 3, W = h->comm.world, r = h->comm.rank;
  HIP_OK(gpb::launch_hist_limbs_pack(h->d_limbs, h->total_bins, nw, h->blk_bin0.data(), h->blk_bins.data(), W, h->blk_max, h->d_rs_send, h->stream));
  if (comm_reducescatter_i64(h->comm, h->d_rs_send, h->d_rs_recv, (size_t)nw * h->blk_max, h->stream)) return -1;
  HIP_OK(gpb::launch_hist_convert_block(h->d_rs_recv, h->blk_max, h->blk_bins[r], h->blk_bin0[r], h->d_absmax, h->d_absmax + 1, const_hess, h->has_hess ? 1 : 0, d_hist_out,
                                        h->stream));
  return 0;
}

/* Data-parallel tree grower: on = reduce-scatter of the integer totals by feature block + exchange of the ranks' best splits (DataParallelTreeLearner,
 * data_parallel_tree_learner.cpp:131, :155-173, :244); off = every rank all-reduces every histogram and searches every feature; < 0 (the default) = by the size
 * of a histogram message: feature blocks from 2 MB on.  Below that a message is latency-bound on any fabric and the feature-block form pays one more collective and
 * one more synchronisation per split: with 2 / 4 / 8 ranks time-sharing one MI355X, config 3's 306 KB histograms, 5.8 / 7.7 / 13.5 ms per tree against
 * 3.5 / 5.4 / 10.4 ms for the all-reduce (profiles/r05_p_tree_multirank.log).  The trees are identical either way (tests/test_multirank_gpu.py). */
int gpb_hip_hist_set_feature_block_exchange(gpb_hip_hist_t* h, int on) {
  API_BEGIN();
  if (!h) return fail("null argument");
  h->block_exchange = on < 0 ? -1 : (on != 0 ? 1 : 0);
  API_END();
}

int gpb_hip_hist_build(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t num_data, double const_hess,
                       double* hist_out, uint64_t* cnt_out) {
  API_BEGIN();
  if (!h || !hist_out) return fail("null argument");
  if (hist_build_impl(h, data_indices, num_data, const_hess, hist_out, cnt_out, 1, nullptr)) return -1;
  API_END();
}

int gpb_hip_hist_bench(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t num_data, double const_hess, int reps,
                       double* ms_avg) {
  API_BEGIN();
  if (!h || !ms_avg || reps < 1) return fail("invalid argument");
  if (hist_build_impl(h, data_indices, num_data, const_hess, nullptr, nullptr, reps, ms_avg)) return -1;
  API_END();
}

static int hist_build_impl(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t num_data, double const_hess,
                           double* hist_out, uint64_t* cnt_out, int reps, double* ms_avg, double* d_target, const int* dev_indices,
                           bool dev_all_rows) {
  {
  if (!h->has_grad) return fail("gradients have not been set (call gpb_hip_hist_set_gradients)");
  if (!data_indices && !dev_indices) num_data = h->n;      // dev_all_rows: the caller's resident list is still the identity
  (void)dev_all_rows;
  if (num_data < 0 || num_data > h->n) return fail("gpb_hip_hist_build: num_data = %d", num_data);
  HIP_OK(hipSetDevice(h->device));
  if (data_indices) {
    if (h->idx_cap < num_data) { dev_free(h->d_idx); HIP_OK(hipMalloc(&h->d_idx, sizeof(int) * (size_t)h->n)); h->idx_cap = h->n; }
    HIP_OK(hipMemcpyAsync(h->d_idx, data_indices, sizeof(int) * (size_t)num_data, hipMemcpyHostToDevice, h->stream));
  }
  // chunking: four workgroups (32 KB of LDS each; five do not fit next to the runtime's own LDS use) are resident per CU: ONE full round
  // of them (1024 workgroups on 256 CUs: 4, 8, 12, 16 per CU measured -> 0.232 / 0.246 / 0.26 / 0.27 ms at n = 1e7), at least 1024 rows per chunk
  const int groups = h->fpad / GPB_HIST_FG;
  if (h->num_cu <= 0) { HIP_OK(hipDeviceGetAttribute(&h->num_cu, hipDeviceAttributeMultiprocessorCount, h->device)); if (h->num_cu <= 0) h->num_cu = 256; }
  const int chunk_mult = h->has_hess ? 2 : 4;          // with hessians the workgroup holds two 32 KB arrays: two per CU
  int nchunks = std::max(1, std::min((num_data + 1023) / 1024, std::max(1, chunk_mult * h->num_cu / groups)));
  if (nchunks >= 16) nchunks &= ~7;                 // multiples of 8: the XCD-aware workgroup order of hist_build_kernel
  // whole rows per lane (hist_build_rows_kernel: 128 KB of LDS, one workgroup of 512 lanes per CU and quad of feature groups) when the
  // hessian is constant, there are at least four feature groups and every CU gets a workgroup with >= 2048 rows
  // Per-row hessians: hist_build_kernel (a workgroup per 16 features, four workgroups per CU).  (A whole-row form with two words per (bin, feature)
  // was measured in round 3 -- same bits, 0.395 vs 0.335 ms at n = 1e7, F = 50, profiles/r03_f_hist_per_row_hessians.log: 64 LDS atomics per row
  // and lane on ONE workgroup per CU lose more to the atomic rate than the shared row prologue saves -- and removed in round 4.)
  const bool rows_kernel = !h->has_hess && groups >= 4 && (long long)num_data >= 2048LL * h->num_cu;
  if (rows_kernel) nchunks = std::max(1, h->num_cu);      // one workgroup per CU and quad of feature groups (launches of whole quads, then the partial one)
  const int rows_per_chunk = (num_data + nchunks - 1) / std::max(nchunks, 1);
  if (nchunks < 16 && rows_per_chunk > 0) nchunks = (num_data + rows_per_chunk - 1) / rows_per_chunk;
  if (nchunks < 1) nchunks = 1;
  if (h->part_chunks < nchunks) {
    dev_free(h->d_part_grad); dev_free(h->d_part_hess); dev_free(h->d_part_cnt);
    const size_t cnt = (size_t)nchunks * h->fpad * GPB_HIST_MAX_BIN;
    HIP_OK(hipMalloc(&h->d_part_grad, sizeof(long long) * cnt));
    HIP_OK(hipMalloc(&h->d_part_hess, sizeof(long long) * cnt));
    HIP_OK(hipMalloc(&h->d_part_cnt, sizeof(uint32_t) * cnt));
    h->part_chunks = nchunks;
  }
  gpb::HistKernelArgs a;
  a.bins_rm = h->d_bins_rm; a.data_indices = dev_indices ? dev_indices : (data_indices ? h->d_idx : nullptr); a.grad = h->d_grad;
  a.hess = h->has_hess ? h->d_hess : nullptr;
  a.part_grad = h->d_part_grad; a.part_hess = h->d_part_hess; a.part_cnt = h->d_part_cnt;
  a.grad_max_bits = h->d_absmax; a.hess_max_bits = h->d_absmax + 1;
  a.fpad = h->fpad; a.num_data = num_data; a.rows_per_chunk = std::max(rows_per_chunk, 1); a.nchunks = nchunks; a.num_features = h->F;
  a.use_rows_kernel = rows_kernel ? 1 : 0;
  a.bins_cm = h->d_bins_cm; a.rstride = h->rstride;
  gpb::HistReduceArgs r;
  r.part_grad = h->d_part_grad; r.part_hess = h->d_part_hess; r.part_cnt = h->d_part_cnt; r.bin_offsets = h->d_bin_offsets;
  r.grad_max_bits = h->d_absmax; r.hess_max_bits = h->d_absmax + 1;
  r.hist_out = d_target ? d_target : h->d_hist; r.cnt_out = h->d_cnt; r.fpad = h->fpad; r.nchunks = nchunks; r.num_features = h->F;
  r.const_hess = const_hess; r.has_hess = h->has_hess ? 1 : 0;
  const bool sharded = h->comm.active() && !ms_avg;          // (gpb_hip_hist_bench times the local build)
  if (sharded) {
    if (!h->d_limbs) HIP_OK(hipMalloc(&h->d_limbs, sizeof(long long) * 5 * (size_t)h->total_bins));
    r.limbs_out = h->d_limbs; r.limb_stride = h->total_bins;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ms_avg) { HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventRecord(e0, h->stream)); }
  for (int rep = 0; rep < reps; ++rep) {
    HIP_OK(gpb::launch_hist_build(a, h->stream));
    HIP_OK(gpb::launch_hist_reduce(r, h->stream));
  }
  if (sharded && hist_finish_sharded(h, const_hess, r.hist_out, h->d_cnt)) return -1;
  if (ms_avg) HIP_OK(hipEventRecord(e1, h->stream));
  if (hist_out) HIP_OK(hipMemcpyAsync(hist_out, d_target ? d_target : h->d_hist, sizeof(double) * 2 * (size_t)h->total_bins, hipMemcpyDeviceToHost, h->stream));
  if (cnt_out) HIP_OK(hipMemcpyAsync(cnt_out, h->d_cnt, sizeof(unsigned long long) * (size_t)h->total_bins, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (ms_avg) { float ms = 0.f; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); *ms_avg = ms / reps; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
  }
  return 0;
}

// Tree grower: histogram of the SMALLER child of the split whose left counts the partition kernels have left in h->d_counts -- enqueued
// without a host round trip; ub_rows = upper bound of the child's rows on this rank (sizes the launch), result in d_target.
static int hist_build_planned(gpb_hip_hist_t* h, const int* rows_base, int seg_begin, int seg_cnt, int seg_gcnt, int min_data_in_leaf, int ub_rows,
                              double const_hess, double* d_target, int* nchunks_without_reduce = nullptr, bool block_form = false) {
  const int groups = h->fpad / GPB_HIST_FG;
  if (h->num_cu <= 0) { HIP_OK(hipDeviceGetAttribute(&h->num_cu, hipDeviceAttributeMultiprocessorCount, h->device)); if (h->num_cu <= 0) h->num_cu = 256; }
  const int chunk_mult = h->has_hess ? 2 : 4;
  int nchunks = std::max(1, std::min((ub_rows + 1023) / 1024, std::max(1, chunk_mult * h->num_cu / groups)));
  if (nchunks >= 16) nchunks &= ~7;
  if (h->part_chunks < nchunks) {
    dev_free(h->d_part_grad); dev_free(h->d_part_hess); dev_free(h->d_part_cnt);
    const size_t cnt = (size_t)nchunks * h->fpad * GPB_HIST_MAX_BIN;
    HIP_OK(hipMalloc(&h->d_part_grad, sizeof(long long) * cnt));
    HIP_OK(hipMalloc(&h->d_part_hess, sizeof(long long) * cnt));
    HIP_OK(hipMalloc(&h->d_part_cnt, sizeof(uint32_t) * cnt));
    h->part_chunks = nchunks;
  }
  gpb::HistKernelArgs a;
  a.bins_rm = h->d_bins_rm; a.data_indices = rows_base; a.grad = h->d_grad; a.hess = h->has_hess ? h->d_hess : nullptr;
  a.part_grad = h->d_part_grad; a.part_hess = h->d_part_hess; a.part_cnt = h->d_part_cnt;
  a.grad_max_bits = h->d_absmax; a.hess_max_bits = h->d_absmax + 1;
  a.fpad = h->fpad; a.num_data = 0; a.rows_per_chunk = 1; a.nchunks = nchunks; a.num_features = h->F;
  a.seg_counts = h->d_counts; a.seg_begin = seg_begin; a.seg_cnt = seg_cnt; a.seg_gcnt = seg_gcnt; a.seg_min_data_in_leaf = min_data_in_leaf;
  gpb::HistReduceArgs r;
  r.part_grad = h->d_part_grad; r.part_hess = h->d_part_hess; r.part_cnt = h->d_part_cnt; r.bin_offsets = h->d_bin_offsets;
  r.grad_max_bits = h->d_absmax; r.hess_max_bits = h->d_absmax + 1;
  r.hist_out = d_target; r.cnt_out = nullptr; r.fpad = h->fpad; r.nchunks = nchunks; r.num_features = h->F;
  r.const_hess = const_hess; r.has_hess = h->has_hess ? 1 : 0;
  if (h->comm.active()) {
    if (!h->d_limbs) HIP_OK(hipMalloc(&h->d_limbs, sizeof(long long) * 5 * (size_t)h->total_bins));
    r.limbs_out = h->d_limbs; r.limb_stride = h->total_bins;
  }
  HIP_OK(gpb::launch_hist_build(a, h->stream));
  // few chunks (leaves below ~8000 rows: most splits of a tree): the children's search sums them itself, one launch and one kernel boundary less per
  // split; with many chunks the dedicated reduction (16 slices per word) is faster than the search workgroups' serial sums (measured: 36 us for 49 chunks)
  if (nchunks_without_reduce && !h->comm.active() && nchunks <= 8) { *nchunks_without_reduce = nchunks; return 0; }
  HIP_OK(gpb::launch_hist_reduce(r, h->stream));
  if (h->comm.active()) {
    if (block_form) { if (hist_finish_sharded_block(h, const_hess, d_target)) return -1; }
    else if (hist_finish_sharded(h, const_hess, d_target, nullptr)) return -1;
  }
  return 0;
}

// ---- data-parallel histograms (mirrors DataParallelTreeLearner, data_parallel_tree_learner.cpp:155-173): every
// rank holds a shard of the rows and builds the leaf histogram of ITS rows; one all-reduce of 5 INTEGER words per bin (fixed-point
// gradient / hessian totals in two limbs each + the count; the fixed-point scale is agreed by a max-all-reduce in
// gpb_hip_hist_set_gradients) completes it, converted once afterwards: counts exact, sums bit-identical for every rank layout.
// Once a communicator is set EVERY build on the handle (gpb_hip_hist_build, _build_slot, the tree grower) is the job-wide histogram.
int gpb_hip_hist_comm_init(gpb_hip_hist_t* h, const unsigned char* id128, int rank, int world) {
  API_BEGIN();
  if (!h || !id128) return fail("null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail("gpb_hip_hist_comm_init: rank %d / world %d", rank, world);
  HIP_OK(hipSetDevice(h->device));
  if (comm_init_rccl(h->comm, id128, rank, world)) return -1;
  h->blk_f0.clear();
  h->has_grad = false;          // the scale of the fixed-point sums must be agreed by all ranks: set the gradients again
  API_END();
}

int gpb_hip_hist_comm_init_local(gpb_hip_hist_t* h, gpb_hip_local_group_t* g, int rank) {
  API_BEGIN();
  if (!h) return fail("null argument");
  HIP_OK(hipSetDevice(h->device));
  if (comm_init_local(h->comm, g, rank)) return -1;
  h->blk_f0.clear();
  h->has_grad = false;
  API_END();
}

int gpb_hip_hist_build_allreduce(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t num_data, double const_hess, double* hist_out,
                                 uint64_t* cnt_out) {
  API_BEGIN();
  if (!h) return fail("null handle");
  if (!h->comm.active()) return fail("no communicator: call gpb_hip_hist_comm_init first");
  if (hist_build_impl(h, data_indices, num_data, const_hess, nullptr, nullptr, 1, nullptr)) return -1;     // sharded: integer totals all-reduced inside
  if (hist_out) HIP_OK(hipMemcpyAsync(hist_out, h->d_hist, sizeof(double) * 2 * (size_t)h->total_bins, hipMemcpyDeviceToHost, h->stream));
  if (cnt_out) HIP_OK(hipMemcpyAsync(cnt_out, h->d_cnt, sizeof(unsigned long long) * (size_t)h->total_bins, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

// ---- resident leaf histograms: build into a slot, FixHistogram, parent - smaller (row a12) ------------------
int gpb_hip_hist_pool_resize(gpb_hip_hist_t* h, int32_t num_slots) {
  API_BEGIN();
  if (!h || num_slots < 1) return fail("gpb_hip_hist_pool_resize: invalid arguments");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  dev_free(h->d_pool);
  HIP_OK(hipMalloc(&h->d_pool, sizeof(double) * 2 * (size_t)h->total_bins * num_slots));
  HIP_OK(hipMemset(h->d_pool, 0, sizeof(double) * 2 * (size_t)h->total_bins * num_slots));
  h->nslots = num_slots;
  API_END();
}

int gpb_hip_hist_set_fix_info(gpb_hip_hist_t* h, const int32_t* view_offset, const int32_t* num_bin, const int32_t* most_freq_bin) {
  API_BEGIN();
  if (!h || !view_offset || !num_bin || !most_freq_bin) return fail("null argument");
  for (int f = 0; f < h->F; ++f) {
    if (most_freq_bin[f] <= 0) continue;
    if (num_bin[f] < 1 || view_offset[f] < 0 || view_offset[f] + num_bin[f] > h->total_bins || most_freq_bin[f] >= num_bin[f])
      return fail("gpb_hip_hist_set_fix_info: feature %d: view [%d, %d) / most_freq_bin %d outside the histogram of %d bins", f,
                  view_offset[f], view_offset[f] + num_bin[f], most_freq_bin[f], h->total_bins);
  }
  HIP_OK(hipSetDevice(h->device));
  if (!h->d_fix) HIP_OK(hipMalloc(&h->d_fix, sizeof(int) * 3 * (size_t)h->F));
  HIP_OK(hipMemcpy(h->d_fix, view_offset, sizeof(int) * (size_t)h->F, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->d_fix + h->F, num_bin, sizeof(int) * (size_t)h->F, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->d_fix + 2 * (size_t)h->F, most_freq_bin, sizeof(int) * (size_t)h->F, hipMemcpyHostToDevice));
  h->h_fix.assign(view_offset, view_offset + h->F); h->h_fix.insert(h->h_fix.end(), num_bin, num_bin + h->F);
  h->h_fix.insert(h->h_fix.end(), most_freq_bin, most_freq_bin + h->F);
  h->has_fix = true;
  API_END();
}

static double* hist_slot(gpb_hip_hist_t* h, int slot) {
  return (h->d_pool && slot >= 0 && slot < h->nslots) ? h->d_pool + 2 * (size_t)h->total_bins * slot : nullptr;
}

int gpb_hip_hist_build_slot(gpb_hip_hist_t* h, int32_t slot, const int32_t* data_indices, int32_t num_data, double const_hess) {
  API_BEGIN();
  if (!h) return fail("null handle");
  double* dst = hist_slot(h, slot);
  if (!dst) return fail("gpb_hip_hist_build_slot: slot %d outside the pool of %d (call gpb_hip_hist_pool_resize)", slot, h->nslots);
  if (hist_build_impl(h, data_indices, num_data, const_hess, nullptr, nullptr, 1, nullptr, dst)) return -1;
  API_END();
}

int gpb_hip_hist_fix_slot(gpb_hip_hist_t* h, int32_t slot, double sum_gradient, double sum_hessian) {
  API_BEGIN();
  if (!h) return fail("null handle");
  double* dst = hist_slot(h, slot);
  if (!dst) return fail("gpb_hip_hist_fix_slot: slot %d outside the pool of %d", slot, h->nslots);
  if (!h->has_fix) return fail("gpb_hip_hist_fix_slot: the feature views have not been set (call gpb_hip_hist_set_fix_info)");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(gpb::launch_hist_fix(dst, h->F, h->d_fix, h->d_fix + h->F, h->d_fix + 2 * (size_t)h->F, sum_gradient, sum_hessian, h->stream));
  API_END();
}

int gpb_hip_hist_subtract_slots(gpb_hip_hist_t* h, int32_t parent_slot, int32_t smaller_slot, int32_t out_slot) {
  API_BEGIN();
  if (!h) return fail("null handle");
  double *p = hist_slot(h, parent_slot), *sm = hist_slot(h, smaller_slot), *o = hist_slot(h, out_slot);
  if (!p || !sm || !o) return fail("gpb_hip_hist_subtract_slots: slot outside the pool of %d", h->nslots);
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(gpb::launch_hist_subtract(p, sm, o, 2 * h->total_bins, h->stream));
  API_END();
}

int gpb_hip_hist_set_split_info(gpb_hip_hist_t* h, const int32_t* offset, const int32_t* default_bin, const int32_t* missing_type) {
  API_BEGIN();
  if (!h || !offset || !default_bin || !missing_type) return fail("null argument");
  std::vector<int> m3((size_t)h->F * 3);
  for (int f = 0; f < h->F; ++f) {
    if (offset[f] < 0 || offset[f] > 1 || missing_type[f] < 0 || missing_type[f] > 2)
      return fail("gpb_hip_hist_set_split_info: feature %d: offset %d / missing type %d", f, offset[f], missing_type[f]);
    m3[3 * f] = offset[f]; m3[3 * f + 1] = default_bin[f]; m3[3 * f + 2] = missing_type[f];
  }
  HIP_OK(hipSetDevice(h->device));
  if (!h->d_meta3) HIP_OK(hipMalloc(&h->d_meta3, sizeof(int) * m3.size()));
  HIP_OK(hipMemcpy(h->d_meta3, m3.data(), sizeof(int) * m3.size(), hipMemcpyHostToDevice));
  h->h_meta3 = m3;
  h->has_split_info = true;
  API_END();
}

int gpb_hip_hist_set_regularisation(gpb_hip_hist_t* h, double lambda_l1, double max_delta_step, double path_smooth, double parent_output) {
  API_BEGIN();
  if (!h) return fail("null argument");
  if (!(lambda_l1 >= 0.0) || !(path_smooth >= 0.0) || !std::isfinite(max_delta_step) || !std::isfinite(parent_output))
    return fail("gpb_hip_hist_set_regularisation: lambda_l1 and path_smooth must be >= 0 (got %g, %g), max_delta_step and parent_output finite", lambda_l1, path_smooth);
  h->reg_l1 = lambda_l1; h->reg_max_delta_step = max_delta_step; h->reg_path_smooth = path_smooth; h->reg_parent_output = parent_output;
  API_END();
}

int gpb_hip_hist_set_max_depth(gpb_hip_hist_t* h, int32_t max_depth) {
  API_BEGIN();
  if (!h) return fail("null argument");
  h->max_depth = max_depth;
  API_END();
}

int gpb_hip_hist_set_root_rows(gpb_hip_hist_t* h, const int32_t* rows, int32_t cnt) {
  API_BEGIN();
  if (!h) return fail("null argument");
  if (!rows || cnt <= 0) { h->root_cnt = 0; return 0; }
  if (cnt > h->n) return fail("gpb_hip_hist_set_root_rows: %d rows, the handle holds %d", cnt, h->n);
  for (int i = 0; i < cnt; ++i)
    if (rows[i] < 0 || rows[i] >= h->n || (i > 0 && rows[i] <= rows[i - 1])) return fail("gpb_hip_hist_set_root_rows: rows must be ascending row indices in [0, %d)", h->n);
  HIP_OK(hipSetDevice(h->device));
  if (!h->d_root_rows) HIP_OK(hipMalloc(&h->d_root_rows, sizeof(int) * (size_t)h->n));
  HIP_OK(hipMemcpyAsync(h->d_root_rows, rows, sizeof(int) * (size_t)cnt, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  h->root_cnt = cnt;
  API_END();
}

int gpb_hip_hist_set_feature_mask(gpb_hip_hist_t* h, const int8_t* is_feature_used) {
  API_BEGIN();
  if (!h) return fail("null argument");
  if (!is_feature_used) h->feature_mask.clear();
  else h->feature_mask.assign(is_feature_used, is_feature_used + h->F);
  API_END();
}

int gpb_hip_hist_find_best_split(gpb_hip_hist_t* h, int32_t slot, double sum_gradient, double sum_hessian, int32_t num_data, double lambda_l2,
                                 int32_t min_data_in_leaf, double min_sum_hessian_in_leaf, double min_gain_to_split,
                                 const int8_t* is_feature_used, int32_t* best_feature, double* per_feature_out10,
                                 int32_t* per_feature_default_left, int32_t* per_feature_splittable) {
  API_BEGIN();
  if (!h || !best_feature) return fail("null argument");
  double* src = hist_slot(h, slot);
  if (!src) return fail("gpb_hip_hist_find_best_split: slot %d outside the pool of %d", slot, h->nslots);
  if (!h->has_fix || !h->has_split_info) return fail("gpb_hip_hist_find_best_split: feature views / metas have not been set (gpb_hip_hist_set_fix_info, gpb_hip_hist_set_split_info)");
  HIP_OK(hipSetDevice(h->device));
  const int F = h->F;
  if (!h->d_split) {
    HIP_OK(hipMalloc(&h->d_split, sizeof(double) * (size_t)F * 10));
    HIP_OK(hipMalloc(&h->d_split_i, sizeof(int) * (size_t)(F + 1)));
    HIP_OK(hipMalloc(&h->d_used, (size_t)F));
  }
  if (is_feature_used) HIP_OK(hipMemcpyAsync(h->d_used, is_feature_used, (size_t)F, hipMemcpyHostToDevice, h->stream));
  HIP_OK(gpb::launch_hist_best_split(src, F, h->d_fix, h->d_fix + F, h->d_meta3, sum_gradient, sum_hessian, num_data, lambda_l2,
                                     min_data_in_leaf, min_sum_hessian_in_leaf, min_gain_to_split,
                                     gpb::SplitReg{ h->reg_l1, h->reg_max_delta_step, h->reg_path_smooth, h->reg_parent_output },
                                     is_feature_used ? h->d_used : nullptr,
                                     h->d_split, h->d_split_i, h->d_split_i + F, h->stream, h->any_cat ? h->d_is_cat : nullptr, h->cat, h->d_cat_bits));
  std::vector<int> ints(F + 1);
  HIP_OK(hipMemcpyAsync(ints.data(), h->d_split_i, sizeof(int) * (size_t)(F + 1), hipMemcpyDeviceToHost, h->stream));
  if (per_feature_out10) HIP_OK(hipMemcpyAsync(per_feature_out10, h->d_split, sizeof(double) * (size_t)F * 10, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  *best_feature = ints[F];
  for (int f = 0; f < F; ++f) {
    if (per_feature_default_left) per_feature_default_left[f] = ints[f] & 1;
    if (per_feature_splittable) per_feature_splittable[f] = (ints[f] >> 1) & 1;
  }
  API_END();
}

/* the sets of bins (8 words per feature) the categorical features' candidates of the LAST gpb_hip_hist_find_best_split send left */
int gpb_hip_hist_last_split_cat_bits(gpb_hip_hist_t* h, uint32_t* bits_out) {
  API_BEGIN();
  if (!h || !bits_out) return fail("null argument");
  std::fill(bits_out, bits_out + (size_t)h->F * 8, 0u);
  if (!h->any_cat) return 0;
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipMemcpyAsync(bits_out, h->d_cat_bits, sizeof(unsigned) * (size_t)h->F * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

static int hist_split_leaf_impl(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t cnt, int32_t feature, uint32_t threshold,
                                int default_left, const uint32_t* cat_bits8, int32_t* lte_out, int32_t* gt_out, int32_t* lte_count);

int gpb_hip_hist_split_leaf(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t cnt, int32_t feature, uint32_t threshold,
                            int default_left, int32_t* lte_out, int32_t* gt_out, int32_t* lte_count) {
  API_BEGIN();
  if (hist_split_leaf_impl(h, data_indices, cnt, feature, threshold, default_left, nullptr, lte_out, gt_out, lte_count)) return -1;
  API_END();
}

/* the same for a categorical feature: the rows whose bin is in the set (8 words, bit b = the feature's bin b) go left (Dataset::Split with a bitset,
 * DenseBin::SplitCategorical, dense_bin.hpp:305-362) */
int gpb_hip_hist_split_leaf_categorical(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t cnt, int32_t feature, const uint32_t* cat_bits8,
                                        int32_t* lte_out, int32_t* gt_out, int32_t* lte_count) {
  API_BEGIN();
  if (!cat_bits8) return fail("null argument");
  if (hist_split_leaf_impl(h, data_indices, cnt, feature, 0u, 0, cat_bits8, lte_out, gt_out, lte_count)) return -1;
  API_END();
}

static int hist_split_leaf_impl(gpb_hip_hist_t* h, const int32_t* data_indices, int32_t cnt, int32_t feature, uint32_t threshold,
                                int default_left, const uint32_t* cat_bits8, int32_t* lte_out, int32_t* gt_out, int32_t* lte_count) {
  {
  if (!h || !lte_out || !gt_out || !lte_count) return fail("null argument");
  if (!h->has_fix || !h->has_split_info) return fail("gpb_hip_hist_split_leaf: feature metas have not been set (gpb_hip_hist_set_fix_info, gpb_hip_hist_set_split_info)");
  if (feature < 0 || feature >= h->F) return fail("gpb_hip_hist_split_leaf: feature %d of %d", feature, h->F);
  if (!data_indices) cnt = h->n;
  if (cnt < 0 || cnt > h->n) return fail("gpb_hip_hist_split_leaf: cnt = %d", cnt);
  *lte_count = 0;
  if (cnt == 0) return 0;
  HIP_OK(hipSetDevice(h->device));
  const int nblk = (cnt + 1023) / 1024;
  const int need = 2 * (nblk + 1) + 3 * h->n;
  if (h->part_cap < need) { dev_free(h->d_part); HIP_OK(hipMalloc(&h->d_part, sizeof(int) * (size_t)need)); h->part_cap = need; }
  int *blk_cnt = h->d_part, *blk_off = blk_cnt + nblk + 1, *d_idx = blk_off + nblk + 1, *d_lte = d_idx + h->n, *d_gt = d_lte + h->n;
  if (data_indices) HIP_OK(hipMemcpyAsync(d_idx, data_indices, sizeof(int) * (size_t)cnt, hipMemcpyHostToDevice, h->stream));
  const int F = h->F;
  const int max_bin = h->h_bin_offsets[feature + 1] - h->h_bin_offsets[feature] - 1;      // stored bins of the (single-feature) group - 1
  HIP_OK(gpb::launch_hist_partition(h->d_bins_rm, h->fpad, feature, max_bin, h->h_meta3[3 * feature + 1], h->h_fix[2 * F + feature],
                                    h->h_meta3[3 * feature + 2], default_left ? 1 : 0, threshold, data_indices ? d_idx : nullptr, cnt,
                                    blk_cnt, blk_off, d_lte, d_gt, h->stream, cat_bits8));
  int nl = 0;
  HIP_OK(hipMemcpyAsync(&nl, blk_off + nblk, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  if (nl > 0) HIP_OK(hipMemcpyAsync(lte_out, d_lte, sizeof(int) * (size_t)nl, hipMemcpyDeviceToHost, h->stream));
  if (cnt - nl > 0) HIP_OK(hipMemcpyAsync(gt_out, d_gt, sizeof(int) * (size_t)(cnt - nl), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  *lte_count = nl;
  }
  return 0;
}

int gpb_hip_hist_get_slot(gpb_hip_hist_t* h, int32_t slot, double* hist_out) {
  API_BEGIN();
  if (!h || !hist_out) return fail("null argument");
  double* src = hist_slot(h, slot);
  if (!src) return fail("gpb_hip_hist_get_slot: slot %d outside the pool of %d", slot, h->nslots);
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipMemcpyAsync(hist_out, src, sizeof(double) * 2 * (size_t)h->total_bins, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  API_END();
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
#include "gpb_laplace.inc"
#include "gpb_tree.inc"
