// tfhe_hip.hip -- C ABI (include/tfhe_hip.h) over the gfx950 kernels.
//
// Host side of the engine: context (device, stream, converted cloud key,
// scratch), launch geometry, and the batched entry points that compose
//   gate prep + blind rotate (one persistent kernel)  ->  key switch (one kernel)
// exactly as gates::batch_* composes them in the reference
// (src/gates.rs:357-383: prepare, batch_blind_rotate, extract + key switch).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/tfhe_hip.h"
#include <errno.h>
#include <sys/random.h>

#include "blind_rotate.hpp"
#include "blind_rotate_wide.hpp"
#if defined(TFHE_EXPERIMENT) && defined(TFHE_EXP_L1)
#include "../../profiles/exp/superseded/blind_rotate_l1.hpp"
#endif
#if defined(TFHE_EXPERIMENT) && defined(TFHE_EXP_WIDE1)
#include "../../profiles/exp/superseded/blind_rotate_wide1.hpp"
#endif
#include "key_switch.hpp"
#include "key_switch_mfma.hpp"
#include "keygen.hpp"
#include "twiddles_host.hpp"

using namespace tfhe;

namespace {

thread_local std::string g_create_error = "";

// Error text is kept PER (calling thread, handle): the header promises Send + Sync use of a handle
// (src/bootstrap/mod.rs:23), so two threads may fail on one context at the same time, and each must read its own
// message -- not a string another thread is assigning or has freed.  Every context and pool carries an id that is
// never reused; the text lives in a thread-local table keyed by it, and tfhe_hip_last_error() returns the calling
// thread's entry (valid until that thread's next failing call on the same handle).
// A handle's entry is erased by the thread that destroys it; entries other threads made for it would stay for those
// threads' lifetimes (a long-lived worker that fails once on each of many short-lived contexts), so the table is bounded:
// when a thread's table has grown past kErrTableSoftCap entries, the next failure on that thread drops the entries of
// handles that no longer exist (g_live_handles).  Reading never inserts.
std::atomic<uint64_t> g_next_handle_id{1};
thread_local std::unordered_map<uint64_t, std::string> t_errors;
std::mutex g_live_mu;
std::unordered_set<uint64_t> g_live_handles;
constexpr size_t kErrTableSoftCap = 64;
inline uint64_t new_handle_id() {  // never reused
  const uint64_t id = g_next_handle_id.fetch_add(1);
  std::lock_guard<std::mutex> lk(g_live_mu);
  g_live_handles.insert(id);
  return id;
}
inline void handle_gone(uint64_t id) {
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_handles.erase(id);
  }
  t_errors.erase(id);
}
// the slot a FAILING call writes its text into (references survive rehashing)
inline std::string &err_slot(uint64_t id) {
  if (t_errors.size() >= kErrTableSoftCap && !t_errors.count(id)) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    for (auto it = t_errors.begin(); it != t_errors.end();) it = g_live_handles.count(it->first) ? std::next(it) : t_errors.erase(it);
  }
  return t_errors[id];
}
// the calling thread's last failure text on handle `id`, "" if it never failed there (no entry is made)
inline const char *err_text(uint64_t id) {
  static const std::string none;
  const auto it = t_errors.find(id);
  return it == t_errors.end() ? none.c_str() : it->second.c_str();
}

constexpr int kKsG = 32;  // ciphertexts per key-switch workgroup

// Calls on one handle are serialised IN ARRIVAL ORDER (a ticket lock).  std::mutex makes no such promise: a thread that
// issues calls back to back re-acquires it before a waiting thread wakes, and under `Send + Sync` use (a Rayon team
// calling Bootstrap::bootstrap on one strategy, src/bootstrap/mod.rs:23) one worker could starve the rest for as long
// as it had work -- seen in tests/cpp/test_mirror.cpp, where 400 failing calls took minutes beside a thread that kept
// bootstrapping.
class FairMutex {
 public:
  void lock() {
    std::unique_lock<std::mutex> lk(m_);
    const uint64_t ticket = next_++;
    cv_.wait(lk, [&] { return serving_ == ticket; });
  }
  void unlock() {
    {
      std::lock_guard<std::mutex> lk(m_);
      ++serving_;
    }
    cv_.notify_all();
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  uint64_t next_ = 0, serving_ = 0;
};

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};
struct PinBuf {  // pinned host arena (hipHostMalloc)
  void *p = nullptr;
  size_t cap = 0;
  bool heap = false;  // front-end lanes only: pinned memory was not to be had, this is ordinary memory (copies still work)
};

}  // namespace

// One resident cloud key in the engine layouts (src/key.rs:51-56).  A context owns one (`own`); every key view
// created with tfhe_hip_key_create owns another and runs on its parent's device, stream, scratch and mutex.
struct KeyState {
  double2 *d_bsk = nullptr;
  uint32_t *d_ksk = nullptr;
  unsigned char *d_ksk8 = nullptr;  // base-4 sets: the key as signed byte planes in MFMA fragment order (k_ksk_planes)
  uint32_t *d_testvec = nullptr;
  uint32_t offset = 0;
  bool key_loaded = false;
  bool reenc_loaded = false;  // d_ksk (+ d_ksk8) hold a proxy re-encryption key (proxy_reenc.rs:224-233) instead of a cloud key's
};

struct Combiner;  // combine.hpp: the combining front end of the small host-pointer calls

struct tfhe_hip_ctx {
  tfhe_hip_params P{};
  int device = 0;
  hipStream_t stream = nullptr;
  KeyState own;
  KeyState *K = &own;            // the key of the call in progress (bound by ENTER under the mutex)
  tfhe_hip_ctx *parent = nullptr;  // non-null: this handle is a key view of `parent` (only P, own, parent are used)
  int views = 0;                 // live key views of this context
  bool dying = false;            // destroyed while views were alive: the last view to go frees the context
  double2 *d_tw = nullptr;
  DevBuf lv1, u1, u2, h_a, h_b, h_c, h_out, h_tv, h_idx, ks_out, ks_dig;  // scratch / host-API staging (ks_dig: key-switch digit bytes)
  PinBuf p_a, p_b, p_c, p_out;  // pinned staging arenas behind h_a / h_b / h_c / h_out (pool members, combiner lanes)
  PinBuf p_tv, p_idx;           // ... behind h_tv / h_idx (combiner lanes only)
  Combiner *comb = nullptr;      // base contexts: concurrent small host-pointer calls are merged into shared launches (combine.hpp)
  bool is_lane = false;          // this context is a combiner lane of another one (never handed to a caller)
  bool stage_pinned = false;     // set by a pool with several members: stage pageable operands through the arenas
  FairMutex mu;  // one call at a time per context, first come first served
  uint64_t id = new_handle_id();  // key of this context's per-thread error text (err_slot)
  bool profiling = false;
  int num_cus = 0;
  bool fast_round = false;  // |pre-rounding value| < 2^51 guaranteed (see round_to_torus<FAST>)
  // ---- dispatch (plan_blind_rotate / plan_key_switch below; tfhe_hip_describe_dispatch prints a plan) ----
  // The two selectors are the supported controls (TFHE_HIP_BR_KERNEL / TFHE_HIP_KS_KERNEL, include/tfhe_hip.h):
  // AUTO picks per batch size, anything else runs that kernel at every batch size.
  int br_force = 0;  // BrKind + 1, 0 = auto
  int ks_force = 0;  // KsKind + 1, 0 = auto
  // crossovers of the automatic choice, in ciphertexts; set from the CU count at creation.  Numeric overrides exist
  // only in -DTFHE_EXPERIMENT builds (profiles/exp/).
  size_t wide_max = 256;      // blind rotate: one eight-wave workgroup per ciphertext up to this batch size
  size_t pair_lo = 0, pair_max = 0;  // ... two ciphertexts per workgroup (k_blind_rotate_pair) for pair_lo < count <= pair_max
  size_t ks_split_max = 256;  // key switch: coefficient walk split over 32 workgroups up to this batch size
  size_t ks_mfma_min = 64;    // smallest batch the matrix-core kernel takes (below: the split kernel)
  size_t ks_sl_chunk_min = 384;  // wider bases: smallest batch the column-sliced kernel takes (with K chunks; below: the split kernel)
  int ks_sliced_sets = 0;     // 0: accumulator sets per lane picked per launch (ks_sl2_pick_sets); else forced (24..36; experiment builds)
  int ks_mfma_ksplit = 0;     // 0: K chunks per row block picked per launch; else forced (1, 2, 4, 8, 16)
  int ks_sl_kchunks = 0;      // 0: K chunks of the column-sliced kernel picked per launch; else forced (1 ... 64, a power of two)
  long br_chunk = 0;  // blind-rotate workgroups per launch: 0 = whole batch (default), -1 = resident set, N = N
  bool exp_wide1 = false;  // (experiment builds with -DTFHE_EXP_WIDE1: SINGLE runs the superseded one-wave-per-row kernel)
  // experiment builds (profiles/exp/midsize.py): the two parts of a blind-rotation plan on TWO streams, and a forced cut
  bool exp_overlap = false;
  size_t exp_split_at = 0;
  int exp_split_kind[2] = {0, 0};
  hipStream_t exp_stream2 = nullptr;
  hipEvent_t exp_ev[2] = {nullptr, nullptr};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_br, ev_ks;
  uint64_t bootstraps = 0;
  hipStream_t scratch_owner = nullptr;  // stream whose queued work may still use lv1/u1/u2
  bool scratch_owned = false;
  // device diagnostics: [0] shader cycles, [1] constant-rate ticks (both summed over blind-rotate
  // workgroups while profiling is on), [2] error flag raised by kernels (bad gate code), [4] / [5] the same two
  // sums for the matrix-core key switch
  unsigned long long *d_diag = nullptr;
  int rtc_khz = 100000;  // rate of s_memrealtime (hipDeviceAttributeWallClockRate)
  ~tfhe_hip_ctx() { handle_gone(id); }  // (registration and erasure cannot drift apart: every `delete` passes here)
};

namespace {
// The HIP current device is per host thread and shared with every other library in the process
// (torch included): set ours for the duration of a call and put the caller's back.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) err = hipSetDevice(dev);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};
}  // namespace

// Every entry point: a key view runs on its parent -- `ctx` is re-pointed at the parent, whose mutex serialises the
// call, whose device is made current (the caller's is put back on return) and whose key pointer K is bound to the
// view's key for the duration of the call (RAII; a plain context binds its own).
namespace {
struct KeyBind {
  tfhe_hip_ctx *c;
  KeyBind(tfhe_hip_ctx *base, KeyState *k) : c(base) { c->K = k; }
  ~KeyBind() { c->K = &c->own; }
};
}  // namespace
#define ENTER(ctx)                                                                     \
  tfhe_hip_ctx *self_ = (ctx);                                                         \
  if (self_->parent) (ctx) = self_->parent;                                            \
  std::lock_guard<FairMutex> lk_((ctx)->mu);                                           \
  KeyBind kb_((ctx), &self_->own);                                                     \
  DeviceGuard dg_((ctx)->device);                                                      \
  if (dg_.err != hipSuccess) {                                                         \
    err_slot((ctx)->id) = std::string("hipSetDevice: ") + hipGetErrorString(dg_.err);  \
    return TFHE_HIP_EHIP;                                                              \
  }

#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      err_slot((ctx)->id) = std::string(#call) + ": " + hipGetErrorString(e_);              \
      return TFHE_HIP_EHIP;                                                                 \
    }                                                                                       \
  } while (0)

#define CHK(expr)                 \
  do {                            \
    int rc_ = (expr);             \
    if (rc_ != TFHE_HIP_OK) return rc_; \
  } while (0)

namespace {

int fail(tfhe_hip_ctx *ctx, int code, const std::string &msg) {
  err_slot(ctx->id) = msg;
  return code;
}

int ensure(tfhe_hip_ctx *ctx, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return TFHE_HIP_OK;
  if (b.p) HIPCHK(ctx, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 4;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    err_slot(ctx->id) = std::string("hipMalloc scratch: ") + hipGetErrorString(e);
    return TFHE_HIP_ENOMEM;
  }
  b.cap = want;
  return TFHE_HIP_OK;
}

struct GatePrep {
  uint32_t ca, cb, cconst;
};

// src/gates.rs:54-150; constants are utils::f64_to_torus(+-0.125 / +-0.25) (utils.rs:9-12)
bool gate_prep(int gate, GatePrep &g) {
  const uint32_t P8 = 0x20000000u, M8 = 0xE0000000u, P4 = 0x40000000u, M4 = 0xC0000000u;
  const uint32_t ONE = 1u, NEG = 0xFFFFFFFFu, TWO = 2u, NEG2 = 0xFFFFFFFEu;
  switch (gate) {
    case TFHE_HIP_NAND: g = {NEG, NEG, P8}; return true;
    case TFHE_HIP_OR: g = {ONE, ONE, P8}; return true;
    case TFHE_HIP_AND: g = {ONE, ONE, M8}; return true;
    case TFHE_HIP_XOR: g = {ONE, TWO, P4}; return true;
    case TFHE_HIP_XNOR: g = {ONE, NEG2, M4}; return true;
    case TFHE_HIP_NOR: g = {NEG, NEG, M8}; return true;
    case TFHE_HIP_ANDNY: g = {NEG, ONE, M8}; return true;
    case TFHE_HIP_ANDYN: g = {ONE, NEG, M8}; return true;
    case TFHE_HIP_ORNY: g = {NEG, ONE, P8}; return true;
    case TFHE_HIP_ORYN: g = {ONE, NEG, P8}; return true;
    case TFHE_HIP_COPY: g = {ONE, 0u, 0u}; return true;
    default: return false;
  }
}

int record_begin(tfhe_hip_ctx *ctx, hipStream_t s, std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) {
  if (!ctx->profiling) return TFHE_HIP_OK;
  hipEvent_t a, b;
  HIPCHK(ctx, hipEventCreate(&a));
  HIPCHK(ctx, hipEventCreate(&b));
  HIPCHK(ctx, hipEventRecord(a, s));
  v.emplace_back(a, b);
  return TFHE_HIP_OK;
}

int record_end(tfhe_hip_ctx *ctx, hipStream_t s, std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) {
  if (!ctx->profiling) return TFHE_HIP_OK;
  HIPCHK(ctx, hipEventRecord(v.back().second, s));
  return TFHE_HIP_OK;
}

// Small calls are arriving at this context's front end (combine.hpp; defined after it): bulk launches then go out in
// chunks, so that a merged launch waits for a chunk boundary and not for the whole batch.
bool comb_interactive(const tfhe_hip_ctx *ctx);
// A one-ciphertext call made while a 65,536-ciphertext launch holds every CU waits for all of it (median 13 ms, p90 308 ms:
// the latency kernel's workgroup needs a whole CU and the batch kernel refills every slot the moment it frees; stream
// priority changes nothing).  Cut into launches of 8,192 ciphertexts the batch costs 1.7 % more (190.6 vs 193.9 k
// bootstraps/s) and the small call waits 39 ms in the median, 50 ms at p90 (16,384: 0.6 %, 70 / 90 ms; 4,096: 5 %, 19 / 23 ms
// -- profiles/exp/logs/r6r_bulk_chunks.log).  Done only while small calls have arrived within the last 250 ms.
constexpr size_t kYieldChunk = 8192;
constexpr int64_t kYieldWindowNs = 250 * 1000 * 1000;

typedef void (*br_kernel_t)(BlindRotateArgs);
#if defined(TFHE_EXPERIMENT) && defined(TFHE_EXP_L1)  // the three-waves-per-SIMD l = 1 experiment (profiles/exp/superseded/blind_rotate_l1.hpp)
bool br_is_l1(const tfhe_hip_ctx *ctx) { return ctx->P.l == 1; }
int br_waves(const tfhe_hip_ctx *ctx) { return br_is_l1(ctx) ? kL1Waves : kBrWaves; }
#else
constexpr bool br_is_l1(const tfhe_hip_ctx *) { return false; }
constexpr int br_waves(const tfhe_hip_ctx *) { return kBrWaves; }
#endif
br_kernel_t br_kernel(const tfhe_hip_ctx *ctx) {
  const bool f = ctx->fast_round;
#if defined(TFHE_EXPERIMENT) && defined(TFHE_EXP_L1)
  if (br_is_l1(ctx)) return f ? k_blind_rotate_l1<true> : k_blind_rotate_l1<false>;
#endif
  switch (ctx->P.l) {
    case 1: return f ? k_blind_rotate<1, true> : k_blind_rotate<1, false>;
    case 2: return f ? k_blind_rotate<2, true> : k_blind_rotate<2, false>;
    default: return f ? k_blind_rotate<3, true> : k_blind_rotate<3, false>;
  }
}

br_kernel_t br_pair_kernel(const tfhe_hip_ctx *ctx) {
  const bool f = ctx->fast_round;
  switch (ctx->P.l) {
    case 1: return f ? k_blind_rotate_pair<1, true> : k_blind_rotate_pair<1, false>;
    case 2: return f ? k_blind_rotate_pair<2, true> : k_blind_rotate_pair<2, false>;
    default: return f ? k_blind_rotate_pair<3, true> : k_blind_rotate_pair<3, false>;
  }
}

br_kernel_t br_single_kernel(const tfhe_hip_ctx *ctx) {
  const bool f = ctx->fast_round;
#if defined(TFHE_EXPERIMENT) && defined(TFHE_EXP_WIDE1)  // the superseded one-wave-per-row latency kernel (profiles/exp/superseded/blind_rotate_wide1.hpp)
  if (ctx->exp_wide1) switch (ctx->P.l) {
      case 1: return f ? k_blind_rotate_wide<1, true> : k_blind_rotate_wide<1, false>;
      case 2: return f ? k_blind_rotate_wide<2, true> : k_blind_rotate_wide<2, false>;
      default: return f ? k_blind_rotate_wide<3, true> : k_blind_rotate_wide<3, false>;
    }
#endif
  switch (ctx->P.l) {
    case 1: return f ? k_blind_rotate_wide2<1, true> : k_blind_rotate_wide2<1, false>;
    case 2: return f ? k_blind_rotate_wide2<2, true> : k_blind_rotate_wide2<2, false>;
    default: return f ? k_blind_rotate_wide2<3, true> : k_blind_rotate_wide2<3, false>;
  }
}

typedef void (*ep_kernel_t)(const uint32_t *, const int32_t *, const double2 *, uint32_t, const double2 *, int,
                            uint32_t, uint32_t *);
ep_kernel_t ep_kernel(const tfhe_hip_ctx *ctx) {
  const bool f = ctx->fast_round;
  switch (ctx->P.l) {
    case 1: return f ? k_external_product<1, true> : k_external_product<1, false>;
    case 2: return f ? k_external_product<2, true> : k_external_product<2, false>;
    default: return f ? k_external_product<3, true> : k_external_product<3, false>;
  }
}

size_t br_lds_bytes(const tfhe_hip_ctx *ctx) {
#if defined(TFHE_EXPERIMENT) && defined(TFHE_EXP_L1)
  if (br_is_l1(ctx)) return blind_rotate_l1_lds_bytes();
#endif
  return blind_rotate_lds_bytes(ctx->P.n);
}

// ---- which blind-rotation kernels a batch of `count` runs on ----------------------------------------
// Three kernels, one arithmetic (the same per-element operations in the same order: results are the same bits
// whichever runs, tests/test_gpu_parity.py::test_dispatch_crossovers_bit_exact):
//   BATCH   k_blind_rotate        one wave per ciphertext, four per workgroup: the throughput kernel
//   SINGLE  k_blind_rotate_wide2  one eight-wave workgroup per ciphertext: the latency kernel
//   PAIR    k_blind_rotate_pair   two ciphertexts per eight-wave workgroup, half a step apart
// A plan is at most two launches over contiguous parts of the batch.
enum BrKind { BR_BATCH = 0, BR_SINGLE = 1, BR_PAIR = 2 };
const char *const kBrKindName[3] = {"batch", "single", "pair"};
struct BrPlan {
  int nparts = 0;
  BrKind kind[2] = {BR_BATCH, BR_BATCH};
  size_t begin[2] = {0, 0}, count[2] = {0, 0};
  void add(BrKind k, size_t b, size_t c) {
    kind[nparts] = k;
    begin[nparts] = b;
    count[nparts] = c;
    ++nparts;
  }
};

BrPlan plan_blind_rotate(const tfhe_hip_ctx *ctx, size_t count) {
  BrPlan pl;
  if (count == 0) return pl;
  if (ctx->br_force) {  // TFHE_HIP_BR_KERNEL: that kernel at every batch size
    pl.add((BrKind)(ctx->br_force - 1), 0, count);
    return pl;
  }
  // Small batches, N = #CUs (measured at 128 bit, profiles/exp/logs/r3x_pair_kernel.log, r3o_crossover.log):
  // up to N ciphertexts one workgroup each (2.2 ms); N < count <= 2N two ciphertexts per workgroup, half a step apart
  // (3.6 ms; two rounds of singles take 4.5); up to 3N the first 2N as pairs and the rest as singles (5.8; three
  // rounds of singles 6.4, pairs alone 7.3, the batch kernel 7.0 for anything up to 4N).
  const size_t N1 = ctx->pair_lo, N2 = ctx->pair_max;
  const bool pairs_on = N2 > N1 && ctx->wide_max >= N1;
  if (pairs_on && count > N1 && count <= N2) {
    pl.add(BR_PAIR, 0, count);
    return pl;
  }
  // 2N < count <= 3N: the first 2N as pairs, the rest one per workgroup (not at l = 1, where the batch kernel's first
  // step is cheaper than a pair launch plus a single one: SECURITY_UINT4 4.3 vs 4.9 ms; l = 2: 5.1 vs 5.4, l = 3: 5.8 vs 6.9)
  if (pairs_on && ctx->P.l >= 2 && N2 == 2 * N1 && count > N2 && count <= N2 + N1) {
    pl.add(BR_PAIR, 0, N2);
    pl.add(BR_SINGLE, N2, count - N2);
    return pl;
  }
  if (count <= ctx->wide_max) {
    pl.add(BR_SINGLE, 0, count);
    return pl;
  }
  // The batch kernel's time is a staircase with a step every 4N (one more four-wave workgroup per CU: 6.9 / 11.5 /
  // 17.1 / 21.8 ms at 1,024 / 2,048 / 3,072 / 4,096).  A tail of up to 2N ciphertexts above a step is cheaper as a
  // latency-kernel launch of its own (2.2 ms up to N, 3.6 up to 2N) than as a whole further step (1,100: 9.1 vs 10.8
  // ms, 2,200: 14.4 vs 17.9, 3,300: 19.7 vs 22.9); done up to 32N, beyond which the step is a few per cent of the launch.
  // (at l = 1 a pair launch costs as much as the step it would save: tails of up to N only)
  size_t tail = 0;
  if (pairs_on && ctx->br_chunk == 0 && count <= 32 * N1) {
    const size_t r = count % (4 * N1);
    if (r > 0 && r <= (ctx->P.l == 1 ? N1 : N2) && count > r) tail = r;
  }
  pl.add(BR_BATCH, 0, count - tail);
  if (tail) pl.add(tail <= N1 ? BR_SINGLE : BR_PAIR, count - tail, tail);
  return pl;
}

int launch_blind_rotate(tfhe_hip_ctx *ctx, hipStream_t s, const uint32_t *in_a, const uint32_t *in_b,
                        GatePrep gp, const uint32_t *testvec, int per_ct, size_t count,
                        uint32_t *out_trlwe, uint32_t *out_lv1, uint32_t *out_ext2,
                        const uint8_t *gate_codes = nullptr) {
  if (count == 0) return TFHE_HIP_OK;
  if (count > 0x7FFFFFFFull) return fail(ctx, TFHE_HIP_EINVAL, "count too large");
  BlindRotateArgs A;
  A.in_a = in_a;
  A.in_b = gp.cb ? in_b : nullptr;
  A.ca = gp.ca;
  A.cb = gp.cb;
  A.cconst = gp.cconst;
  A.gate_codes = gate_codes;
  A.testvec = testvec ? testvec : ctx->K->d_testvec;
  A.per_ct_stride = (testvec && per_ct) ? (size_t)2 * kN : 0;
  A.bsk = ctx->K->d_bsk;
  A.tw = ctx->d_tw;
  A.n = ctx->P.n;
  A.bgbit = ctx->P.bgbit;
  A.offset = ctx->K->offset;
  A.out_trlwe = out_trlwe;
  A.out_lv1 = out_lv1;
  A.out_ext2 = out_ext2;
  A.count = count;
  A.clk = ctx->profiling ? ctx->d_diag : nullptr;
  A.err_flag = reinterpret_cast<uint32_t *>(ctx->d_diag + 2);
  // sample_extract_index_2 reads a[n - i] of an N-coefficient polynomial (trlwe.rs:122-136): the reference
  // indexes out of bounds (panics) for n > N; refuse instead of reading the b half of the accumulator
  if (out_ext2 && ctx->P.n > kN)
    return fail(ctx, TFHE_HIP_EINVAL, "bootstrap without key switch needs n <= N (sample_extract_index_2)");
  if (gp.cb && !in_b) return fail(ctx, TFHE_HIP_EINVAL, "second gate operand is NULL");
  // ciphertexts [done, done + m) of this call as a launch of their own
  auto part = [&](size_t done, size_t m) {
    BlindRotateArgs S = A;
    S.in_a = A.in_a + done * (size_t)(ctx->P.n + 1);
    if (A.in_b) S.in_b = A.in_b + done * (size_t)(ctx->P.n + 1);
    if (A.gate_codes) S.gate_codes = A.gate_codes + done;
    S.testvec = A.testvec + done * A.per_ct_stride;
    if (A.out_trlwe) S.out_trlwe = A.out_trlwe + done * (size_t)(2 * kN);
    if (A.out_lv1) S.out_lv1 = A.out_lv1 + done * (size_t)(kN + 1);
    if (A.out_ext2) S.out_ext2 = A.out_ext2 + done * (size_t)(ctx->P.n + 1);
    S.count = m;
    return S;
  };
  auto launch = [&](br_kernel_t kern, unsigned grid, unsigned threads, size_t lds, const BlindRotateArgs &S) -> int {
    CHK(record_begin(ctx, s, ctx->ev_br));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, S);
    HIPCHK(ctx, hipGetLastError());
    return record_end(ctx, s, ctx->ev_br);
  };
  BrPlan pl = plan_blind_rotate(ctx, count);
#ifdef TFHE_EXPERIMENT
  if (ctx->exp_split_at && count > ctx->exp_split_at) {  // a forced cut: kind[0] over [0, at), kind[1] over the rest
    pl = BrPlan();
    pl.add((BrKind)ctx->exp_split_kind[0], 0, ctx->exp_split_at);
    pl.add((BrKind)ctx->exp_split_kind[1], ctx->exp_split_at, count - ctx->exp_split_at);
  }
  const bool overlap = ctx->exp_overlap && pl.nparts == 2 && ctx->exp_stream2;
  hipStream_t s_home = s;
  // (event and wait calls take the runtime's own name for the default stream, NULL: they fault on the hipStreamLegacy handle)
  hipStream_t s_rt = s == hipStreamLegacy ? (hipStream_t) nullptr : s;
  if (overlap) {  // the second part may start as soon as what precedes this call on `s` is done
    HIPCHK(ctx, hipEventRecord(ctx->exp_ev[0], s_rt));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->exp_stream2, ctx->exp_ev[0], 0));
  }
#endif
  for (int q = 0; q < pl.nparts; ++q) {
#ifdef TFHE_EXPERIMENT
    if (overlap) s = q == 1 ? ctx->exp_stream2 : s_home;
#endif
    const size_t begin = pl.begin[q], m_all = pl.count[q];
    if (pl.kind[q] == BR_PAIR) {
      CHK(launch(br_pair_kernel(ctx), (unsigned)((m_all + 1) / 2), 64u * kPairWaves, blind_rotate_pair_lds_bytes(ctx->P.n),
                 part(begin, m_all)));
    } else if (pl.kind[q] == BR_SINGLE) {
#if defined(TFHE_EXPERIMENT) && defined(TFHE_EXP_WIDE1)
      if (ctx->exp_wide1) {
        CHK(launch(br_single_kernel(ctx), (unsigned)m_all, 128u * (unsigned)ctx->P.l,
                   blind_rotate_wide_lds_bytes(ctx->P.n, ctx->P.l), part(begin, m_all)));
        continue;
      }
#endif
      CHK(launch(br_single_kernel(ctx), (unsigned)m_all, 64u * kWide2Waves, blind_rotate_wide2_lds_bytes(ctx->P.n, ctx->P.l),
                 part(begin, m_all)));
    } else {
      // Default: ONE launch of the whole part.  The four waves of a workgroup meet at a barrier every CMUX step
      // (blind_rotate.hpp), which keeps the resident workgroups streaming the key in near lock-step on its own:
      // L1 86 % / L2 97 % hits.  (Experiment builds: TFHE_HIP_BR_CHUNK splits the batch into launches of N ciphertexts,
      // -1: the resident set -- the round-1 remedy for free-running one-wave workgroups.)
      const size_t lds = br_lds_bytes(ctx);
      size_t chunk = m_all;
      if (ctx->br_chunk > 0) chunk = (size_t)ctx->br_chunk;
      if (ctx->br_chunk < 0) {
        int per_cu = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, br_kernel(ctx), 64 * br_waves(ctx), lds);
        if (e == hipSuccess && per_cu > 0 && ctx->num_cus > 0) chunk = (size_t)per_cu * ctx->num_cus * br_waves(ctx);
      }
      if (ctx->br_chunk == 0 && m_all >= 2 * kYieldChunk && comb_interactive(ctx)) chunk = kYieldChunk;
      if (chunk == 0 || chunk > m_all) chunk = m_all;
      for (size_t done = 0; done < m_all; done += chunk) {
        const size_t m = (m_all - done < chunk) ? m_all - done : chunk;
        const unsigned bw = (unsigned)br_waves(ctx);
        CHK(launch(br_kernel(ctx), (unsigned)((m + bw - 1) / bw), 64u * bw, lds, part(begin + done, m)));
      }
    }
  }
#ifdef TFHE_EXPERIMENT
  if (overlap) {  // whatever follows on the caller's stream waits for the second part
    HIPCHK(ctx, hipEventRecord(ctx->exp_ev[1], ctx->exp_stream2));
    HIPCHK(ctx, hipStreamWaitEvent(s_rt, ctx->exp_ev[1], 0));
  }
#endif
  ctx->bootstraps += count;
  return TFHE_HIP_OK;
}

// ---- base-4 key switch on the matrix cores (key_switch_mfma.hpp) -----------------------------------
typedef void (*km_kernel_t)(const uint32_t *, const unsigned char *, int, int, uint32_t *, size_t, unsigned long long *, int);
// instantiated tile counts of the widest column block (32 columns each): 1 .. kKmMaxTiles
int ks_mfma_nt(int n) {
  const int need = ks_mfma_tiles(n);
  return (need >= 1 && need <= kKmMaxTiles) ? need : 0;
}
template <int NT>
km_kernel_t km_kernel_from(int nt) {
  if constexpr (NT > kKmMaxTiles) return nullptr;
  else return nt == NT ? (km_kernel_t)k_key_switch_mfma<NT> : km_kernel_from<NT + 1>(nt);
}
km_kernel_t km_kernel(int nt) { return km_kernel_from<1>(nt); }
bool ks_mfma_possible(const tfhe_hip_params &P) {
  return P.basebit == 2 && P.t >= 6 && P.t <= 13 && ks_mfma_nt(P.n) != 0;
}
// rows of a level-1 buffer the matrix-core kernel may read: whole 256-row workgroups
size_t lv1_rows(size_t count) { return (count + kKmRows - 1) / kKmRows * kKmRows; }

// ---- which key-switch kernel a batch of `count` runs on ----------------------------------------------
// Integer arithmetic: every kernel returns the same bits (trgsw.rs:332-360).
//   MFMA     k_key_switch_mfma    base 4: one-hot byte-plane contraction on the int8 matrix cores
//   SLICED   k_key_switch_sliced  wider bases: 64-column slices of all `base` candidate rows through an LDS ring
//   B4       k_key_switch_b4      base 4 without the matrix cores: candidate rows through a wave-private LDS ring
//   GENERIC  k_key_switch         any set: buffer loads, 32 ciphertexts per workgroup
//   SPLIT    k_key_switch_split   small batches: one ciphertext's walk cut over 32 workgroups
// MFMA, SPLIT and SLICED with K chunks merge partial sums with integer atomics into a zeroed output: `atomics`.
enum KsKind { KS_MFMA = 0, KS_SLICED = 1, KS_B4 = 2, KS_GENERIC = 3, KS_SPLIT = 4 };
const char *const kKsKindName[5] = {"mfma", "sliced", "b4", "generic", "split"};
struct KsPlan {
  KsKind kind = KS_GENERIC;
  int kparts = 1;  // MFMA: K chunks per row block; SLICED: K chunks (grid.z)
  int sets = 0;    // SLICED: accumulator sets per lane
  bool atomics = false;
};

// column-sliced kernel (second form): bases 16 .. 128, its ring + stage within a CU's LDS
bool ks_sliced_fits(const tfhe_hip_params &P) {
  return P.basebit >= 4 && P.basebit <= 7 && ks_sl2_lds_bytes(P.basebit, 36, ks_sl2_rp(P.basebit)) <= 160 * 1024;
}
constexpr size_t kKsSl2Slab = 131072;  // ciphertexts per launch of the column-sliced kernel (bounds its digit scratch: 0.4 GB at t = 3)
typedef void (*sl2_kernel_t)(const uint32_t *, size_t, const uint32_t *, const unsigned char *, int, int, uint32_t *, size_t);
template <int BB>
sl2_kernel_t sl2_kernel_sets(int sets) {
  constexpr int RP = ks_sl2_rp(BB);
  switch (sets) {
    case 24: return k_key_switch_sliced<BB, 24, RP>;
    case 28: return k_key_switch_sliced<BB, 28, RP>;
    case 36: return k_key_switch_sliced<BB, 36, RP>;
    default: return k_key_switch_sliced<BB, 32, RP>;
  }
}
sl2_kernel_t sl2_kernel(int basebit, int sets) {
  switch (basebit) {
    case 4: return sl2_kernel_sets<4>(sets);
    case 5: return sl2_kernel_sets<5>(sets);
    case 6: return sl2_kernel_sets<6>(sets);
    default: return sl2_kernel_sets<7>(sets);
  }
}
// Accumulator sets per lane, chosen per launch so that the grid fills whole rounds of the machine: a workgroup's time is
// proportional to S, the grid is ceil(count / 16S) x slices workgroups, `slots` of them run at once, so the launch
// costs ceil(grid / slots) x S (SECURITY_UINT4, 65,536 ciphertexts, 13 slices, 512 slots: S = 32 is 3.25 rounds = 4 x 32,
// S = 36 is 2.9 rounds = 3 x 36; base 64 / 128 run ONE eight-wave workgroup per CU, 256 slots: SECURITY_UINT7, 19
// slices: S = 32 is 4.75 rounds = 5 x 32 -- 15.8 ms measured, 16.8 / 17.8 at 36 / 28) -- profiles/exp/logs/r4_ks_sl_ablation.log.
bool ks_sl2_sets_allowed(int basebit, int sets) {
  (void)basebit;
  return sets == 24 || sets == 28 || sets == 32 || sets == 36;
}
int ks_sl2_pick_sets(int basebit, size_t count, int slices, int slots) {
  int best = basebit == 7 ? 36 : basebit == 6 ? 28 : 32;
  size_t best_cost = ~(size_t)0;
  for (int sets : {24, 28, 32, 36}) {
    if (!ks_sl2_sets_allowed(basebit, sets)) continue;
    const size_t grid = ((count + (size_t)ks_sl2_cts(basebit, sets) - 1) / (size_t)ks_sl2_cts(basebit, sets)) * (size_t)slices;
    const size_t cost = ((grid + (size_t)slots - 1) / (size_t)slots) * (size_t)sets;
    if (cost < best_cost || (cost == best_cost && sets == 32)) {
      best = sets;
      best_cost = cost;
    }
  }
  return best;
}
bool ks_b4_fits(const tfhe_hip_params &P) {
  const int bd = ((ksk_row_words(P.n) >> 2) + 63) & ~63;
  return P.basebit == 2 && ks_b4_lds_bytes(bd >> 6, kKsG) <= 64 * 1024;
}
bool ks_kind_possible(const tfhe_hip_params &P, KsKind k) {
  switch (k) {
    case KS_MFMA: return ks_mfma_possible(P);
    case KS_SLICED: return ks_sliced_fits(P);
    case KS_B4: return ks_b4_fits(P);
    default: return true;
  }
}

KsPlan plan_key_switch(const tfhe_hip_ctx *ctx, size_t count) {
  const tfhe_hip_params &P = ctx->P;
  const int n = P.n;
  KsPlan pl;
  const bool forced = ctx->ks_force != 0;
  // base-4 sets from 64 ciphertexts up: the matrix-core kernel, its walk over K cut into chunks while the batch is
  // too small to fill the chip with row blocks (0.10 / 0.11 / 0.15 / 0.18 / 0.25 / 0.40 ms at 64 / 256 / 512 / 1,024 /
  // 2,048 / 4,096 ciphertexts; the split kernel takes 0.12 / 0.33 / 0.55 / 0.97 / 1.8 / 3.6, the matrix-core kernel
  // without chunks 0.42-0.49 throughout: profiles/exp/logs/r3_ks_splitk.log).
  // Wider bases from 384 ciphertexts up: the column-sliced kernel with its walk over the coefficients cut into chunks
  // (SECURITY_UINT4: 0.37 / 0.37 / 0.57 / 1.05 ms at 512 / 1,024 / 2,048 / 4,096 ciphertexts where the split kernel
  // takes 0.46 / 0.85 / 1.62 / 3.42 and wins below: 0.27 vs 0.29 at 256 -- profiles/exp/logs/r3_ks_sl_chunks.log).
  // Smaller batches, and whatever neither LDS kernel covers, up to ks_split_max: the split kernel.
  if (forced) pl.kind = (KsKind)(ctx->ks_force - 1);
  else if (ctx->K->d_ksk8 && count >= ctx->ks_mfma_min) pl.kind = KS_MFMA;
  else {
    const bool sliced_ok = ks_sliced_fits(P);
    if (count <= ctx->ks_split_max && !(sliced_ok && count >= ctx->ks_sl_chunk_min)) pl.kind = KS_SPLIT;
    else if (sliced_ok) pl.kind = KS_SLICED;
    else if (ks_b4_fits(P)) pl.kind = KS_B4;
    else pl.kind = KS_GENERIC;
  }
  if (pl.kind == KS_MFMA) {
    // one workgroup per (128 rows, column block, byte plane, K chunk); small batches have few row blocks: the walk over
    // K is cut into up to 16 chunks so that there are about two workgroups per CU to run
    const size_t rb = (count + kKmRows - 1) / kKmRows;
    const int tiles = ks_mfma_total_tiles(n);
    const size_t ncb = (size_t)(tiles < kKmColBlocks ? tiles : kKmColBlocks);
    int ksplit = 1;
    while (ksplit < 16 && rb * ncb * 4 * (size_t)ksplit < 2 * (size_t)ctx->num_cus) ksplit *= 2;
    if (ctx->ks_mfma_ksplit) ksplit = ctx->ks_mfma_ksplit;
    pl.kparts = ksplit;
    pl.atomics = true;
  } else if (pl.kind == KS_SLICED) {
    // accumulator sets per lane: whichever fills whole rounds of the machine (workgroups resident at once: two per CU
    // while two rings fit a CU's LDS, else one)
    const int slices = (n + 1 + 63) / 64, rp = ks_sl2_rp(P.basebit);
    const size_t in_launch = count < kKsSl2Slab ? count : kKsSl2Slab;
    // workgroups resident per CU: two waves per SIMD by registers (8 wave slots), and the rings must fit the LDS
    const int by_waves = 8 / ks_sl2_waves(P.basebit), by_lds = 2 * ks_sl2_lds_bytes(P.basebit, 36, rp) <= 160 * 1024 ? 2 : 1;
    const int per_cu = by_waves < by_lds ? by_waves : by_lds;
    int sets = ks_sl2_pick_sets(P.basebit, in_launch, slices, per_cu * ctx->num_cus);
    if (ctx->ks_sliced_sets) sets = ctx->ks_sliced_sets;
    // small batches have few ciphertext groups: the walk over the N * t groups is cut into up to 64 chunks (grid.z)
    // so that about two workgroups per CU exist; the chunks meet in the zeroed output through integer atomics.
    // A chunk is whole rings (2 * rp groups).
    const size_t groups = (in_launch + (size_t)ks_sl2_cts(P.basebit, sets) - 1) / (size_t)ks_sl2_cts(P.basebit, sets);
    int kchunks = 1;
    while (kchunks < 64 && groups * (size_t)slices * (size_t)kchunks < 2 * (size_t)ctx->num_cus) kchunks *= 2;
    if (ctx->ks_sl_kchunks) kchunks = ctx->ks_sl_kchunks;
    while (kchunks > 1 && (kN * P.t / kchunks) % (2 * rp) != 0) kchunks /= 2;
    pl.sets = sets;
    pl.kparts = kchunks;
    pl.atomics = kchunks > 1;
  } else if (pl.kind == KS_SPLIT) {
    pl.kparts = 32;
    pl.atomics = true;
  }
  return pl;
}

// (re)build the byte planes from the u32 engine key; called wherever a key becomes current
int build_ksk_planes(tfhe_hip_ctx *ctx) {
  if (!ks_mfma_possible(ctx->P) || (ctx->ks_force && ctx->ks_force - 1 != KS_MFMA)) return TFHE_HIP_OK;
  const tfhe_hip_params &P = ctx->P;
  const size_t bytes = ks_mfma_key_bytes(P.n, P.t);
  if (!ctx->K->d_ksk8) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_ksk8, bytes + kKmKeyTailPad));
  const size_t chunks = bytes / 16;
  hipLaunchKernelGGL(k_ksk_planes, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, ctx->stream, ctx->K->d_ksk,
                     ctx->K->d_ksk8, P.n, P.t, chunks);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_HIP_OK;
}

int launch_key_switch(tfhe_hip_ctx *ctx, hipStream_t s, const uint32_t *lv1, uint32_t *out, size_t count) {
  if (count == 0) return TFHE_HIP_OK;
  const tfhe_hip_params &P = ctx->P;
  const int n = P.n;
  const KsPlan pl = plan_key_switch(ctx, count);
  if (pl.kind == KS_MFMA && !ctx->K->d_ksk8) return fail(ctx, TFHE_HIP_EINVAL, "matrix-core key switch: byte planes not built");
  const size_t obytes = count * (size_t)(n + 1) * 4;
  // Kernels that merge partial sums with integer atomics need a zeroed DEVICE destination: a host (pinned, zero-copy)
  // output would take the atomics over PCIe, which not every root complex completes (AtomicOps are optional), and
  // silently returns wrong words where it does not -- such outputs go through a device buffer and leave in one copy.
  // Classified and allocated before the event pair opens, so that a failure here leaves no half-recorded pair behind.
  uint32_t *dst = out;
  bool host_out = false;
  if (pl.atomics) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, (const void *)out) != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, TFHE_HIP_EINVAL, "key switch output is not GPU-addressable memory");
    }
    if (at.type == hipMemoryTypeHost) {
      host_out = true;
      CHK(ensure(ctx, ctx->ks_out, obytes));
      dst = (uint32_t *)ctx->ks_out.p;
    }
  }
  if (pl.kind == KS_SLICED) {
    const size_t in_launch = count < kKsSl2Slab ? count : kKsSl2Slab;
    CHK(ensure(ctx, ctx->ks_dig, (size_t)(kN * P.t / 4) * ks_sl2_ct_stride(in_launch, P.basebit, pl.sets) * 4));
  }
  const int rw4 = ksk_row_words(n) >> 2;
  const int bd = (rw4 + 63) & ~63;  // <= 320 for n <= 1279
  const size_t ksk_bytes = (size_t)kN * P.t * (1u << P.basebit) * ksk_row_words(n) * 4;
  CHK(record_begin(ctx, s, ctx->ev_ks));
  auto body = [&]() -> int {
    if (pl.atomics) HIPCHK(ctx, hipMemsetAsync(dst, 0, obytes, s));
    switch (pl.kind) {
      case KS_MFMA: {
        const int nt = ks_mfma_nt(n);
        const size_t rb = (count + kKmRows - 1) / kKmRows;
        const int tiles = ks_mfma_total_tiles(n);
        const unsigned ncb = (unsigned)(tiles < kKmColBlocks ? tiles : kKmColBlocks);
        hipLaunchKernelGGL(km_kernel(nt), dim3((unsigned)rb * (unsigned)pl.kparts, ncb, 4), dim3(64 * kKmWaves),
                           ks_mfma_lds_bytes(nt), s, lv1, (const unsigned char *)ctx->K->d_ksk8, n, P.t, dst, count,
                           ctx->profiling ? ctx->d_diag + 4 : nullptr, pl.kparts);
        break;
      }
      case KS_SPLIT:
        // small batch: split each ciphertext's walk over 32 workgroups, merge with integer atomics
        hipLaunchKernelGGL(k_key_switch_split, dim3((unsigned)count, (unsigned)pl.kparts), dim3(bd), 0, s, lv1,
                           (const uint4 *)ctx->K->d_ksk, (uint32_t)ksk_bytes, n, P.basebit, P.t, dst);
        break;
      case KS_SLICED: {
        // digits first (k_ks_digits: one byte per (ciphertext, group), [quad of groups][ciphertext]), then the walk;
        // slabs of kKsSl2Slab ciphertexts bound the digit scratch
        const int slices = (n + 1 + 63) / 64, rp = ks_sl2_rp(P.basebit);
        const size_t lds = ks_sl2_lds_bytes(P.basebit, pl.sets, rp);
        const sl2_kernel_t kern = sl2_kernel(P.basebit, pl.sets);
        const unsigned quads = (unsigned)(kN * P.t / 4);
        for (size_t done = 0; done < count; done += kKsSl2Slab) {
          const size_t m = count - done < kKsSl2Slab ? count - done : kKsSl2Slab;
          const size_t ct_stride = ks_sl2_ct_stride(m, P.basebit, pl.sets);
          const uint32_t *src = lv1 + done * (size_t)(kN + 1);
          hipLaunchKernelGGL(k_ks_digits, dim3((unsigned)((ct_stride + 63) / 64), quads / 32), dim3(256), 0, s, src,
                             (uint32_t *)ctx->ks_dig.p, ct_stride, m, P.basebit, P.t);
          const size_t groups = (m + (size_t)ks_sl2_cts(P.basebit, pl.sets) - 1) / (size_t)ks_sl2_cts(P.basebit, pl.sets);
          hipLaunchKernelGGL(kern, dim3((unsigned)groups, (unsigned)slices, (unsigned)pl.kparts), dim3(64u * (unsigned)ks_sl2_waves(P.basebit)), lds, s,
                             (const uint32_t *)ctx->ks_dig.p, ct_stride, src, (const unsigned char *)ctx->K->d_ksk, n, P.t,
                             dst + done * (size_t)(n + 1), m);
        }
        break;
      }
      case KS_B4:
        hipLaunchKernelGGL((k_key_switch_b4<kKsG>), dim3((unsigned)((count + kKsG - 1) / kKsG)), dim3(bd),
                           ks_b4_lds_bytes(bd >> 6, kKsG), s, lv1, (const unsigned char *)ctx->K->d_ksk, n, P.t, dst, count);
        break;
      default:
        hipLaunchKernelGGL((k_key_switch<kKsG>), dim3((unsigned)((count + kKsG - 1) / kKsG)), dim3(bd), 0, s, lv1,
                           (const uint4 *)ctx->K->d_ksk, (uint32_t)ksk_bytes, n, P.basebit, P.t, dst, count);
        break;
    }
    HIPCHK(ctx, hipGetLastError());
    if (host_out) HIPCHK(ctx, hipMemcpyAsync(out, dst, obytes, hipMemcpyDefault, s));
    return TFHE_HIP_OK;
  };
  const int rc = body();
  const int rc_end = record_end(ctx, s, ctx->ev_ks);  // the pair is closed whichever way the body ended
  return rc != TFHE_HIP_OK ? rc : rc_end;
}

int need_key(tfhe_hip_ctx *ctx) {
  if (!ctx->K->key_loaded) return fail(ctx, TFHE_HIP_ENOKEY, "cloud key not loaded");
  return TFHE_HIP_OK;
}

int need_reenc_key(tfhe_hip_ctx *ctx) {
  if (!ctx->K->reenc_loaded) return fail(ctx, TFHE_HIP_ENOKEY, "re-encryption key not loaded");
  return TFHE_HIP_OK;
}

hipStream_t pick(tfhe_hip_ctx *ctx, void *stream) { return stream ? (hipStream_t)stream : ctx->stream; }

// The intermediate buffers (lv1, u1, u2) belong to the context, not to a call.  Work queued on one
// stream may still be using them when the next call arrives on another stream: drain the previous
// owner first (same-stream calls are ordered by the stream itself and pay nothing).
int claim_scratch(tfhe_hip_ctx *ctx, hipStream_t s) {
  if (ctx->scratch_owned && ctx->scratch_owner != s) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  ctx->scratch_owner = s;
  ctx->scratch_owned = true;
  return TFHE_HIP_OK;
}

// ---- device-pointer implementations (mutex held by caller) -------------------

int gate_dev(tfhe_hip_ctx *ctx, int gate, const uint32_t *a, const uint32_t *b, uint32_t *out,
             size_t count, hipStream_t s) {
  GatePrep gp;
  if (!gate_prep(gate, gp)) return fail(ctx, TFHE_HIP_EINVAL, "unknown gate");
  CHK(claim_scratch(ctx, s));
  CHK(ensure(ctx, ctx->lv1, lv1_rows(count) * (size_t)(kN + 1) * 4));
  CHK(launch_blind_rotate(ctx, s, a, b, gp, nullptr, 0, count, nullptr, (uint32_t *)ctx->lv1.p, nullptr));
  return launch_key_switch(ctx, s, (const uint32_t *)ctx->lv1.p, out, count);
}

int gates_mixed_dev(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a, const uint32_t *b, uint32_t *out,
                    size_t count, hipStream_t s) {
  GatePrep gp{1u, 1u, 0u};  // placeholders; cb != 0 keeps in_b attached, the kernel reads the codes
  CHK(claim_scratch(ctx, s));
  CHK(ensure(ctx, ctx->lv1, lv1_rows(count) * (size_t)(kN + 1) * 4));
  CHK(launch_blind_rotate(ctx, s, a, b, gp, nullptr, 0, count, nullptr, (uint32_t *)ctx->lv1.p, nullptr, gates));
  return launch_key_switch(ctx, s, (const uint32_t *)ctx->lv1.p, out, count);
}

// proxy_reenc::reencrypt_tlwe_lv0 (proxy_reenc.rs:468-510): pad to the key switch's source shape, then the key switch
int reencrypt_dev(tfhe_hip_ctx *ctx, const uint32_t *in, uint32_t *out, size_t count, hipStream_t s) {
  if (count == 0) return TFHE_HIP_OK;
  CHK(claim_scratch(ctx, s));
  CHK(ensure(ctx, ctx->lv1, lv1_rows(count) * (size_t)(kN + 1) * 4));
  hipLaunchKernelGGL(k_reenc_pad, dim3((unsigned)count), dim3(256), 0, s, in, (uint32_t *)ctx->lv1.p, ctx->P.n);
  HIPCHK(ctx, hipGetLastError());
  return launch_key_switch(ctx, s, (const uint32_t *)ctx->lv1.p, out, count);
}

// per-ciphertext gates, bootstrap_without_key_switch outputs (the first level of Gates::mux, gates.rs:165-177)
int gates_mixed_nks_dev(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a, const uint32_t *b, uint32_t *out,
                        size_t count, hipStream_t s) {
  GatePrep gp{1u, 1u, 0u};
  return launch_blind_rotate(ctx, s, a, b, gp, nullptr, 0, count, nullptr, nullptr, out, gates);
}

int bootstrap_dev(tfhe_hip_ctx *ctx, const uint32_t *in, const uint32_t *testvec, int per_ct,
                  int keyswitch, uint32_t *out, size_t count, hipStream_t s) {
  GatePrep gp;
  gate_prep(TFHE_HIP_COPY, gp);
  if (keyswitch) {
    CHK(claim_scratch(ctx, s));
    CHK(ensure(ctx, ctx->lv1, lv1_rows(count) * (size_t)(kN + 1) * 4));
    CHK(launch_blind_rotate(ctx, s, in, nullptr, gp, testvec, per_ct, count, nullptr,
                            (uint32_t *)ctx->lv1.p, nullptr));
    return launch_key_switch(ctx, s, (const uint32_t *)ctx->lv1.p, out, count);
  }
  return launch_blind_rotate(ctx, s, in, nullptr, gp, testvec, per_ct, count, nullptr, nullptr, out);
}

int lincomb_bootstrap_dev(tfhe_hip_ctx *ctx, GatePrep gp, const uint32_t *a, const uint32_t *b,
                          const uint32_t *testvec, int per_ct, int keyswitch, uint32_t *out, size_t count,
                          hipStream_t s) {
  if (keyswitch) {
    CHK(claim_scratch(ctx, s));
    CHK(ensure(ctx, ctx->lv1, lv1_rows(count) * (size_t)(kN + 1) * 4));
    CHK(launch_blind_rotate(ctx, s, a, b, gp, testvec, per_ct, count, nullptr, (uint32_t *)ctx->lv1.p, nullptr));
    return launch_key_switch(ctx, s, (const uint32_t *)ctx->lv1.p, out, count);
  }
  return launch_blind_rotate(ctx, s, a, b, gp, testvec, per_ct, count, nullptr, nullptr, out);
}

int lincomb_dev(tfhe_hip_ctx *ctx, GatePrep gp, const uint32_t *a, const uint32_t *b, uint32_t *out, size_t count,
                hipStream_t s) {
  if (count == 0) return TFHE_HIP_OK;
  const size_t total = count * (size_t)(ctx->P.n + 1);
  hipLaunchKernelGGL(k_tlwe_lincomb, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gp.ca, a,
                     gp.cb, gp.cb ? b : nullptr, gp.cconst, out, (uint32_t)(ctx->P.n + 1), total);
  HIPCHK(ctx, hipGetLastError());
  return TFHE_HIP_OK;
}

int mux_dev(tfhe_hip_ctx *ctx, int naive, const uint32_t *a, const uint32_t *b, const uint32_t *c,
            uint32_t *out, size_t count, hipStream_t s) {
  const size_t ctb = count * (size_t)(ctx->P.n + 1) * 4;
  CHK(claim_scratch(ctx, s));
  CHK(ensure(ctx, ctx->u1, ctb));
  CHK(ensure(ctx, ctx->u2, ctb));
  uint32_t *u1 = (uint32_t *)ctx->u1.p, *u2 = (uint32_t *)ctx->u2.p;
  GatePrep g_and, g_andny, g_or;
  gate_prep(TFHE_HIP_AND, g_and);
  gate_prep(TFHE_HIP_ANDNY, g_andny);  // and(not(a), c) = -a + c - 1/8  (gates.rs:172-175, 196-197)
  gate_prep(TFHE_HIP_OR, g_or);
  if (naive) {  // gates.rs:189-199
    CHK(gate_dev(ctx, TFHE_HIP_AND, a, b, u1, count, s));
    CHK(gate_dev(ctx, TFHE_HIP_ANDNY, a, c, u2, count, s));
    return gate_dev(ctx, TFHE_HIP_OR, u1, u2, out, count, s);
  }
  // gates.rs:157-183: two bootstrap_without_key_switch, add, one full bootstrap
  CHK(launch_blind_rotate(ctx, s, a, b, g_and, nullptr, 0, count, nullptr, nullptr, u1));
  CHK(launch_blind_rotate(ctx, s, a, c, g_andny, nullptr, 0, count, nullptr, nullptr, u2));
  return gate_dev(ctx, TFHE_HIP_OR, u1, u2, out, count, s);
}

// host staging helpers
// Pool members (several contexts fed from one host by one thread each) stage pageable operands through a pinned
// arena of their own: the member's thread copies its slice with memcpy, the DMA engine takes it from there, and
// no two members meet in the runtime's single pageable-copy staging path.  A lone context keeps the runtime's
// pipelined pageable copy (one thread cannot memcpy 550 MB faster than that).
int ensure_pinned(tfhe_hip_ctx *ctx, PinBuf &b, size_t bytes) {
  if (bytes <= b.cap) return TFHE_HIP_OK;
  if (b.p && b.heap) free(b.p);
  else if (b.p) HIPCHK(ctx, hipHostFree(b.p));
  b.heap = false;
  b.p = nullptr;
  b.cap = 0;
  const size_t want = bytes + bytes / 4;
  if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    b.p = nullptr;
    return TFHE_HIP_ENOMEM;  // caller falls back to the pageable copy
  }
  b.cap = want;
  return TFHE_HIP_OK;
}

int to_dev(tfhe_hip_ctx *ctx, DevBuf &b, const void *src, size_t bytes) {
  CHK(ensure(ctx, b, bytes));
  if (ctx->stage_pinned && bytes >= (1u << 20)) {
    PinBuf *pin = &b == &ctx->h_a ? &ctx->p_a : &b == &ctx->h_b ? &ctx->p_b : &b == &ctx->h_c ? &ctx->p_c : nullptr;
    if (pin && ensure_pinned(ctx, *pin, bytes) == TFHE_HIP_OK) {
      memcpy(pin->p, src, bytes);
      HIPCHK(ctx, hipMemcpyAsync(b.p, pin->p, bytes, hipMemcpyHostToDevice, ctx->stream));
      return TFHE_HIP_OK;
    }
  }
  HIPCHK(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return TFHE_HIP_OK;
}

int to_host(tfhe_hip_ctx *ctx, void *dst, const DevBuf &b, size_t bytes) {
  if (ctx->stage_pinned && bytes >= (1u << 20) && &b == &ctx->h_out && ensure_pinned(ctx, ctx->p_out, bytes) == TFHE_HIP_OK) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->p_out.p, b.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->p_out.p, bytes);
    return TFHE_HIP_OK;
  }
  HIPCHK(ctx, hipMemcpyAsync(dst, b.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_HIP_OK;
}

// Zero-copy for pinned host buffers: memory from tfhe_hip_host_alloc (hipHostMalloc) or registered with
// hipHostRegister is addressable by the GPU, so the host entry points hand such buffers to the kernels as they
// are -- each ciphertext is read once in the blind rotation's prologue and written once by the key switch, and
// those PCIe transactions spread over the whole launch instead of three staging copies around it.
// Returns the device view of `p`, or nullptr when `p` is ordinary pageable memory.
template <class T>
T *pinned_view(T *p, size_t bytes) {
  if (!p) return nullptr;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, (const void *)p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory is "invalid value" to the runtime: not an error here
    return nullptr;
  }
  if (at.type != hipMemoryTypeHost) return nullptr;
  // the LAST byte must belong to the same pinned allocation / registration as the first: a range registered
  // shorter than the operand (or an interior pointer near the end of one) would fault in the kernel, where the
  // staged path works -- so such operands are staged
  if (bytes > 1) {
    hipPointerAttribute_t last;
    const void *q = (const void *)((const unsigned char *)p + bytes - 1);
    if (hipPointerGetAttributes(&last, q) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    if (last.type != hipMemoryTypeHost) return nullptr;
    void *b0 = nullptr, *b1 = nullptr;
    size_t s0 = 0, s1 = 0;
    if (hipMemGetAddressRange((hipDeviceptr_t *)&b0, &s0, (hipDeviceptr_t)p) == hipSuccess &&
        hipMemGetAddressRange((hipDeviceptr_t *)&b1, &s1, (hipDeviceptr_t)q) == hipSuccess) {
      if (b0 != b1) return nullptr;
    } else {
      (void)hipGetLastError();  // registered (not allocated) memory may have no address range: both ends are host-pinned
    }
  }
  void *d = nullptr;
  if (hipHostGetDevicePointer(&d, (void *)p, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return (T *)d;
}

}  // namespace

#include "combine.hpp"

namespace {
bool comb_interactive(const tfhe_hip_ctx *ctx) {
  const Combiner *C = ctx->comb;
  if (!C || C->max_count.load(std::memory_order_relaxed) == 0) return false;
  const int64_t t = C->last_arrival_ns.load(std::memory_order_relaxed);
  return t != 0 && combq::now_ns() - t < kYieldWindowNs;
}
}  // namespace

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

const char *tfhe_hip_name(void) { return TFHE_ABLATED ? "hip-gfx950-EXPERIMENT" : "hip-gfx950"; }

int tfhe_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char *tfhe_hip_last_error(const tfhe_hip_ctx *ctx) {
  if (ctx && ctx->parent) ctx = ctx->parent;
  return ctx ? err_text(ctx->id) : g_create_error.c_str();
}

int tfhe_hip_ctx_create(const tfhe_hip_params *p, int device, tfhe_hip_ctx **out) {
  if (!p || !out) {
    g_create_error = "null argument";
    return TFHE_HIP_EINVAL;
  }
  *out = nullptr;
  if (p->n < 1 || p->n > 1279 || p->l < 1 || p->l > 3 || p->bgbit < 1 || p->l * p->bgbit > 32 ||
      p->basebit < 1 || p->basebit > 10 || p->t < 1 || p->basebit * p->t > 31 ||
      (double)kN * p->t * (double)(1u << p->basebit) * ksk_row_words(p->n) * 4.0 >= 4294967296.0) {
    g_create_error = "unsupported parameter set";
    return TFHE_HIP_EINVAL;
  }
  // The supported controls (include/tfhe_hip.h): force ONE kernel at every batch size.  The parity suite uses them to
  // hold every shipped kernel to the CPU checker; they never change result bits.  (Validated before anything is allocated.)
  int br_force = 0, ks_force = 0;
  if (const char *env = getenv("TFHE_HIP_BR_KERNEL")) {
    const std::string v(env);
    if (v == "batch") br_force = BR_BATCH + 1;
    else if (v == "single") br_force = BR_SINGLE + 1;
    else if (v == "pair") br_force = BR_PAIR + 1;
    else if (v != "auto" && !v.empty()) {
      g_create_error = "TFHE_HIP_BR_KERNEL must be auto, batch, single or pair";
      return TFHE_HIP_EINVAL;
    }
    if (br_force == BR_PAIR + 1 && blind_rotate_pair_lds_bytes(p->n) > 160 * 1024) {
      g_create_error = "TFHE_HIP_BR_KERNEL=pair: not available for this parameter set";
      return TFHE_HIP_EINVAL;
    }
  }
  if (const char *env = getenv("TFHE_HIP_KS_KERNEL")) {
    const std::string v(env);
    int k = -1;
    for (int i = 0; i < 5; ++i)
      if (v == kKsKindName[i]) k = i;
    if (k < 0 && v != "auto" && !v.empty()) {
      g_create_error = "TFHE_HIP_KS_KERNEL must be auto, mfma, sliced, b4, generic or split";
      return TFHE_HIP_EINVAL;
    }
    if (k >= 0 && !ks_kind_possible(*p, (KsKind)k)) {
      g_create_error = std::string("TFHE_HIP_KS_KERNEL=") + v + ": not available for this parameter set";
      return TFHE_HIP_EINVAL;
    }
    ks_force = k + 1;
  }
  long combine_env = -1;  // -1: the default (the device's CU count)
  if (const char *env = getenv("TFHE_HIP_COMBINE")) {
    char *end = nullptr;
    const long v = strtol(env, &end, 10);
    if (*env && end && *end == 0 && v >= 0 && v <= (long)Combiner::kBatchCap) combine_env = v;
    else if (*env) {
      g_create_error = "TFHE_HIP_COMBINE must be 0 (off) or the largest call to merge, at most 4096 ciphertexts";
      return TFHE_HIP_EINVAL;
    }
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = std::string("no HIP device: ") + hipGetErrorString(e);
    return TFHE_HIP_EHIP;
  }
  if (device < 0 || device >= ndev) {
    g_create_error = "device ordinal out of range";
    return TFHE_HIP_EINVAL;
  }
  tfhe_hip_ctx *ctx = new tfhe_hip_ctx();
  ctx->P = *p;
  ctx->device = device;
  auto bail = [&](const char *what, hipError_t err) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(err);
    delete ctx;
    return TFHE_HIP_EHIP;
  };
  DeviceGuard dg(device);
  if (dg.err != hipSuccess) return bail("hipSetDevice", dg.err);
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return bail("hipStreamCreate", e);
  if ((e = hipMalloc((void **)&ctx->d_diag, 1024)) != hipSuccess) return bail("hipMalloc diagnostics", e);
  if ((e = hipMemset(ctx->d_diag, 0, 1024)) != hipSuccess) return bail("hipMemset diagnostics", e);
  if (hipDeviceGetAttribute(&ctx->rtc_khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || ctx->rtc_khz <= 0)
    ctx->rtc_khz = 100000;
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
  ctx->num_cus = prop.multiProcessorCount;
  const size_t combine_max = combine_env >= 0 ? (size_t)combine_env : 2 * (size_t)ctx->num_cus;
  // pre-rounding magnitude bound: 2l polynomials x N terms x (Bg/2) digit x 2^31 key coefficient
  ctx->fast_round = std::log2(2.0 * p->l) + 10.0 + (p->bgbit - 1) + 31.0 < 51.0;
  // crossovers of the automatic dispatch, measured on the 256-CU part and kept as multiples of the CU count:
  // key switch vs the group kernels ~7.8k ciphertexts (base 4, LDS ring), ~4.1k (column-sliced); blind rotation vs the
  // batch kernel (7.0 ms for anything up to 1,024 ciphertexts at 128 bit): the eight-wave form takes 2.2 / 4.5 / 6.6 /
  // 8.5 ms for 1 / 2 / 3 / 4 rounds of one workgroup per CU (profiles/exp/logs/r3o_crossover.log) -- three rounds only
  // at l = 3: at l = 1, 2 the batch kernel's first step (4.3 / 5.4 ms) is cheaper than three rounds of singles (6.2 ms)
  ctx->ks_split_max = (p->basebit == 2 ? 28 : 16) * (size_t)ctx->num_cus;
  ctx->wide_max = (p->l >= 3 ? 3 : 2) * (size_t)ctx->num_cus;
  ctx->pair_lo = (size_t)ctx->num_cus;
  ctx->pair_max = 2 * (size_t)ctx->num_cus;
  if (blind_rotate_pair_lds_bytes(p->n) > 160 * 1024) ctx->pair_max = 0;  // (n > 1,900: no parameter set)
  ctx->br_force = br_force;
  ctx->ks_force = ks_force;
#ifdef TFHE_EXPERIMENT
  // Numeric overrides of the crossovers and of the per-launch choices: experiment builds only (profiles/exp/).
  if (const char *env = getenv("TFHE_HIP_FAST_ROUND")) ctx->fast_round = ctx->fast_round && atoi(env) != 0;
  if (const char *env = getenv("TFHE_HIP_KS_SLICED_SETS")) {
    const int v = atoi(env);
    ctx->ks_sliced_sets = (v == 24 || v == 28 || v == 32 || v == 36) ? v : 0;
  }
  if (const char *env = getenv("TFHE_HIP_KS_MFMA_MIN")) ctx->ks_mfma_min = (size_t)atol(env);
  if (const char *env = getenv("TFHE_HIP_KS_SL_CHUNK_MIN")) ctx->ks_sl_chunk_min = (size_t)atol(env);
  if (const char *env = getenv("TFHE_HIP_KS_SL_KCHUNKS")) {
    const int v = atoi(env);
    ctx->ks_sl_kchunks = (v >= 1 && v <= 64 && (v & (v - 1)) == 0) ? v : 0;
  }
  if (const char *env = getenv("TFHE_HIP_KS_MFMA_KSPLIT")) {
    const int v = atoi(env);
    ctx->ks_mfma_ksplit = (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ? v : 0;
  }
  if (const char *env = getenv("TFHE_HIP_WIDE_MAX")) ctx->wide_max = (size_t)atol(env);
  if (const char *env = getenv("TFHE_HIP_PAIR_LO")) ctx->pair_lo = (size_t)atol(env);
  if (const char *env = getenv("TFHE_HIP_PAIR_MAX")) ctx->pair_max = (size_t)atol(env);
  if (const char *env = getenv("TFHE_HIP_KS_SPLIT_MAX")) ctx->ks_split_max = (size_t)atol(env);
  if (const char *env = getenv("TFHE_HIP_BR_CHUNK")) ctx->br_chunk = atol(env);
  if (const char *env = getenv("TFHE_HIP_BR_OVERLAP")) ctx->exp_overlap = atoi(env) != 0;
  if (const char *env = getenv("TFHE_HIP_BR_SPLIT")) {  // "<at>:<kind0>:<kind1>", kinds 0 batch / 1 single / 2 pair
    unsigned long at = 0;
    int k0 = 0, k1 = 0;
    if (sscanf(env, "%lu:%d:%d", &at, &k0, &k1) == 3 && k0 >= 0 && k0 <= 2 && k1 >= 0 && k1 <= 2) {
      ctx->exp_split_at = at;
      ctx->exp_split_kind[0] = k0;
      ctx->exp_split_kind[1] = k1;
    }
  }
  if (ctx->exp_overlap) {
    (void)hipStreamCreateWithFlags(&ctx->exp_stream2, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&ctx->exp_ev[0], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&ctx->exp_ev[1], hipEventDisableTiming);
  }
#ifdef TFHE_EXP_WIDE1
  if (const char *env = getenv("TFHE_HIP_BR_WIDE2")) ctx->exp_wide1 = atoi(env) == 0;
#endif
#endif
  // Dynamic LDS above the 64 KiB default.  The attribute belongs to the (kernel, device), not to the context, and
  // contexts of different n share an instantiation (SECURITY_80/110/128_BIT all run k_blind_rotate<3, true>): it is
  // set to the size for the LARGEST supported n, so the order in which contexts are created cannot lower it below
  // what an earlier context launches with.  (A launch still requests only what its own n needs.)
  {
    constexpr int kMaxN = 1279;  // tfhe_hip_ctx_create's bound
    auto set_lds = [&](const void *kern, size_t bytes) {
      const size_t cap = 160 * 1024;
      return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes < cap ? bytes : cap));
    };
    if ((e = set_lds((const void *)br_kernel(ctx), br_is_l1(ctx) ? br_lds_bytes(ctx) : blind_rotate_lds_bytes(kMaxN))) != hipSuccess)
      return bail("hipFuncSetAttribute(k_blind_rotate)", e);
    if ((e = set_lds((const void *)br_single_kernel(ctx), blind_rotate_wide2_lds_bytes(kMaxN, p->l))) != hipSuccess)
      return bail("hipFuncSetAttribute(k_blind_rotate_wide2)", e);
    if (ctx->pair_max && (e = set_lds((const void *)br_pair_kernel(ctx), blind_rotate_pair_lds_bytes(kMaxN))) != hipSuccess)
      return bail("hipFuncSetAttribute(k_blind_rotate_pair)", e);
    if (ks_sliced_fits(*p))
      for (int sets : {24, 28, 32, 36})
        if ((e = set_lds((const void *)sl2_kernel(p->basebit, sets), ks_sl2_lds_bytes(p->basebit, sets, ks_sl2_rp(p->basebit)))) != hipSuccess)
          return bail("hipFuncSetAttribute(k_key_switch_sliced)", e);
    if (ks_mfma_possible(*p)) {  // one instantiation per tile count: its LDS size does not depend on n beyond that
      const int nt = ks_mfma_nt(p->n);
      if ((e = set_lds((const void *)km_kernel(nt), ks_mfma_lds_bytes(nt))) != hipSuccess)
        return bail("hipFuncSetAttribute(k_key_switch_mfma)", e);
    }
#if defined(TFHE_EXPERIMENT) && defined(TFHE_EXP_WIDE1)
    {
      const bool keep = ctx->exp_wide1;
      ctx->exp_wide1 = true;
      (void)set_lds((const void *)br_single_kernel(ctx), blind_rotate_wide_lds_bytes(kMaxN, p->l));
      ctx->exp_wide1 = keep;
    }
#endif
  }
  std::vector<double2> tw;
  make_twiddles(tw);
  if ((e = hipMalloc((void **)&ctx->d_tw, tw.size() * sizeof(double2))) != hipSuccess)
    return bail("hipMalloc twiddles", e);
  if ((e = hipMemcpy(ctx->d_tw, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice)) != hipSuccess)
    return bail("hipMemcpy twiddles", e);
  // The combining front end (combine.hpp): host-pointer calls of up to `max_count` ciphertexts from concurrent threads
  // share launches.  Default bound: 2 x #CUs (what the pair kernel runs in one go).  Measured with calls of 512 / 1,024 /
  // 4,096 gates (profiles/exp/logs/r7_combining_bound.log): merging them costs a LONE caller nothing at 512 (4.08 vs 4.16 ms),
  // 2.6 % at 1,024 and 13 % at 4,096 (one thread packs 23 MB), and gives 8 concurrent callers +23 % at 512, +16 % at
  // 1,024 and nothing at 4,096.  TFHE_HIP_COMBINE=0 switches the front end off, any other number is the bound
  // (tfhe_hip_set_combining changes it at run time).
  ctx->comb = new Combiner();
  ctx->comb->max_count = combine_max;
#ifdef TFHE_EXPERIMENT
  if (const char *env = getenv("TFHE_HIP_COMBINE_LANES")) ctx->comb->nlanes = std::max(1, std::min((int)Combiner::kLanes, atoi(env)));
  if (const char *env = getenv("TFHE_HIP_COMBINE_ZEROCOPY")) ctx->comb->zero_copy_in = atoi(env) != 0;
  if (const char *env = getenv("TFHE_HIP_COMBINE_HEAP")) g_comb_force_heap = atoi(env) != 0;
  if (const char *env = getenv("TFHE_HIP_LANE_PRIORITY")) ctx->comb->lane_high_priority = atoi(env) != 0;
  if (const char *env = getenv("TFHE_HIP_LINGER_WINDOW_US")) ctx->comb->linger_window_us = atol(env);
  if (const char *env = getenv("TFHE_HIP_LINGER_QUIET_US")) ctx->comb->linger_quiet_us = atol(env);
  if (const char *env = getenv("TFHE_HIP_LINGER_MAX_US")) ctx->comb->linger_max_us = atol(env);
#endif
  *out = ctx;
  return TFHE_HIP_OK;
}

namespace {
void free_key(KeyState &k) {
  if (k.d_bsk) (void)hipFree(k.d_bsk);
  if (k.d_ksk) (void)hipFree(k.d_ksk);
  if (k.d_ksk8) (void)hipFree(k.d_ksk8);
  if (k.d_testvec) (void)hipFree(k.d_testvec);
  k = KeyState();
}
}  // namespace

void tfhe_hip_ctx_destroy(tfhe_hip_ctx *ctx) {
  if (!ctx) return;
  if (ctx->parent) {  // a key view: drain the work that may still read its key, free the key, leave the parent alone
    tfhe_hip_ctx *base = ctx->parent;
    bool last_of_dying = false;
    {
      std::lock_guard<FairMutex> lk(base->mu);
      DeviceGuard dg(base->device);
      if (base->scratch_owned && base->scratch_owner != base->stream) (void)hipStreamSynchronize(base->scratch_owner);
      if (base->stream) (void)hipStreamSynchronize(base->stream);
      comb_quiesce(base);  // (merged launches run on the lanes' streams)
      free_key(ctx->own);
      last_of_dying = --base->views == 0 && base->dying;
      delete ctx;  // (a view's id keys no text -- its errors are its parent's -- but it is registered as live until here)
    }
    if (last_of_dying) tfhe_hip_ctx_destroy(base);  // the parent was destroyed first: it has waited for its views
    return;
  }
  {
    // Destroyed before its views (the header asks for the opposite order): the views still run on this context's
    // stream, scratch and mutex, so keep it alive until the last of them goes.
    std::lock_guard<FairMutex> lk(ctx->mu);
    if (ctx->views > 0) {
      ctx->dying = true;
      return;
    }
  }
  DeviceGuard dg(ctx->device);
  comb_destroy(ctx);  // the front end's lanes (private sibling contexts) go first
  if (ctx->scratch_owned && ctx->scratch_owner != ctx->stream) (void)hipStreamSynchronize(ctx->scratch_owner);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_diag) (void)hipFree(ctx->d_diag);
  for (auto &p : ctx->ev_br) {
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  for (auto &p : ctx->ev_ks) {
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  DevBuf *bufs[] = {&ctx->lv1, &ctx->u1, &ctx->u2, &ctx->h_a, &ctx->h_b, &ctx->h_c, &ctx->h_out, &ctx->h_tv, &ctx->h_idx, &ctx->ks_out, &ctx->ks_dig};
  for (DevBuf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  free_key(ctx->own);
  for (PinBuf *b : {&ctx->p_a, &ctx->p_b, &ctx->p_c, &ctx->p_out, &ctx->p_tv, &ctx->p_idx})
    if (b->p) b->heap ? free(b->p) : (void)hipHostFree(b->p);
  if (ctx->d_tw) (void)hipFree(ctx->d_tw);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// ---- key views: several resident cloud keys on ONE context ------------------------------------------------
int tfhe_hip_key_create(tfhe_hip_ctx *ctx, tfhe_hip_ctx **out) {
  if (!out) return TFHE_HIP_EINVAL;
  *out = nullptr;
  if (!ctx) return TFHE_HIP_EINVAL;
  tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;  // a view of a view is a view of the same context
  std::lock_guard<FairMutex> lk(base->mu);
  tfhe_hip_ctx *v = new tfhe_hip_ctx();
  v->P = base->P;
  v->device = base->device;
  v->parent = base;
  ++base->views;
  *out = v;
  return TFHE_HIP_OK;
}

tfhe_hip_ctx *tfhe_hip_key_parent(tfhe_hip_ctx *key) { return key ? (key->parent ? key->parent : key) : nullptr; }

int tfhe_hip_key_is_loaded(tfhe_hip_ctx *ctx) {
  if (!ctx) return 0;
  // No lock: the host mirrors ask this before EVERY call (is the view's key resident yet?), and the context's mutex may
  // be held for the length of a 65,536-ciphertext host call -- a one-gate call on another thread would wait 330 ms just to
  // learn what it already knows.  The flag is set last by the calls that load a key and cleared first by those that
  // change one; a caller that races its own key load is the caller's to order (as for any call under that key).
  return __atomic_load_n(&ctx->own.key_loaded, __ATOMIC_ACQUIRE) ? 1 : 0;
}

int tfhe_hip_load_cloud_key(tfhe_hip_ctx *ctx, const double *bsk, const uint32_t *ksk,
                            uint32_t decomp_offset, const uint32_t *testvec) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (!bsk || !ksk || !testvec) return fail(ctx, TFHE_HIP_EINVAL, "null key pointer");
  // Work queued earlier on the caller's streams (*_dev entry points) may still be reading the key this
  // call is about to overwrite: drain it first.
  if (ctx->scratch_owned) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scratch_owned = false;
  comb_quiesce(ctx);  // ... and merged launches on the front end's lanes
  const tfhe_hip_params &P = ctx->P;
  const size_t polys = (size_t)P.n * 2 * P.l * 2;
  const size_t bsk_bytes = polys * kN * sizeof(double);
  const int base = 1 << P.basebit;
  const size_t ksk_words = (size_t)kN * P.t * base * (size_t)(P.n + 1);
  ctx->K->key_loaded = ctx->K->reenc_loaded = false;
  if (!ctx->K->d_bsk) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_bsk, bsk_bytes));
  if (!ctx->K->d_ksk)
    HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_ksk, (size_t)kN * P.t * base * ksk_row_words(P.n) * 4 + 4096));
  if (!ctx->K->d_testvec) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_testvec, 2 * kN * 4));
  // bootstrapping key: upload the reference layout, permute + scale on the device
  double *d_ref = nullptr;
  HIPCHK(ctx, hipMalloc((void **)&d_ref, bsk_bytes));
  hipError_t e = hipMemcpyAsync(d_ref, bsk, bsk_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_bsk_convert, dim3((unsigned)polys), dim3(512), 0, ctx->stream, d_ref, ctx->K->d_bsk, polys, key_scale(ctx->fast_round));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_ref);
  if (e != hipSuccess) return fail(ctx, TFHE_HIP_EHIP, std::string("bsk upload: ") + hipGetErrorString(e));
  // key-switching key: upload the reference layout, pad rows to 16 B and zero the k == 0 rows
  {
    const size_t rows = (size_t)kN * P.t * base;
    uint32_t *d_kref = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d_kref, ksk_words * 4));
    hipError_t e2 = hipMemcpyAsync(d_kref, ksk, ksk_words * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e2 == hipSuccess) {
      hipLaunchKernelGGL(k_ksk_convert, dim3((unsigned)rows), dim3(256), 0, ctx->stream, d_kref, ctx->K->d_ksk, P.n, base, rows);
      e2 = hipGetLastError();
    }
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_kref);
    if (e2 != hipSuccess) return fail(ctx, TFHE_HIP_EHIP, std::string("ksk upload: ") + hipGetErrorString(e2));
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->K->d_testvec, testvec, 2 * kN * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  CHK(build_ksk_planes(ctx));
  ctx->K->offset = decomp_offset;
  ctx->K->key_loaded = true;
  comb_prepare(ctx);
  return TFHE_HIP_OK;
}

namespace {
// ctx->mu held, ctx's device current
int gen_cloud_key_locked(tfhe_hip_ctx *ctx, const uint32_t *key_lv0, const uint32_t *key_lv1, double alpha_ksk,
                         double alpha_bsk, const ChaChaKey &rk) {
  if (!key_lv0 || !key_lv1) return fail(ctx, TFHE_HIP_EINVAL, "null key pointer");
  if (!(alpha_ksk >= 0.0) || !(alpha_bsk >= 0.0)) return fail(ctx, TFHE_HIP_EINVAL, "negative noise parameter");
  // Work queued earlier on the caller's streams (*_dev entry points) may still be reading the key this
  // call is about to overwrite: drain it first.
  if (ctx->scratch_owned) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scratch_owned = false;
  comb_quiesce(ctx);  // ... and merged launches on the front end's lanes
  const tfhe_hip_params &P = ctx->P;
  const int base = 1 << P.basebit;
  const size_t polys = (size_t)P.n * 2 * P.l * 2;
  ctx->K->key_loaded = ctx->K->reenc_loaded = false;
  if (!ctx->K->d_bsk) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_bsk, polys * kN * sizeof(double)));
  if (!ctx->K->d_ksk)
    HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_ksk, (size_t)kN * P.t * base * ksk_row_words(P.n) * 4 + 4096));
  if (!ctx->K->d_testvec) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_testvec, 2 * kN * 4));
  // The secret keys, the spectrum of the ring key and the generator key do not outlive the call on the device,
  // whichever way it ends: the guard zeroes the four staging buffers and drains the stream on every exit path
  // (an early return would otherwise leave them in buffers that later batches reuse as plain staging space, and
  // could return while an asynchronous copy still reads this frame).
  struct Wipe {
    tfhe_hip_ctx *c;
    ~Wipe() {
      for (DevBuf *b : {&c->h_a, &c->h_b, &c->h_c, &c->h_idx})
        if (b->p) (void)hipMemsetAsync(b->p, 0, b->cap < 65536 ? b->cap : 65536, c->stream);
      (void)hipStreamSynchronize(c->stream);
    }
  } wipe{ctx};
  CHK(to_dev(ctx, ctx->h_a, key_lv0, (size_t)P.n * 4));
  CHK(to_dev(ctx, ctx->h_b, key_lv1, (size_t)kN * 4));
  CHK(ensure(ctx, ctx->h_c, (size_t)kN2 * sizeof(double2)));
  const uint32_t *d_k0 = (const uint32_t *)ctx->h_a.p, *d_k1 = (const uint32_t *)ctx->h_b.p;
  double2 *d_spec = (double2 *)ctx->h_c.p;
  hipLaunchKernelGGL(k_key_spectrum, dim3(1), dim3(64), kStageLdsBytes, ctx->stream, d_k1, ctx->d_tw, d_spec);
  HIPCHK(ctx, hipGetLastError());
  // the generator key travels in a device buffer (not in kernel-argument memory) and is wiped with the other secrets
  CHK(ensure(ctx, ctx->h_idx, sizeof(ChaChaKey)));
  HIPCHK(ctx, hipMemcpy(ctx->h_idx.p, &rk, sizeof(ChaChaKey), hipMemcpyHostToDevice));  // synchronous: rk is the caller's stack
  const ChaChaKey *d_rk = (const ChaChaKey *)ctx->h_idx.p;
  const dim3 bgrid((unsigned)(P.n * 2 * P.l));
  switch (P.l) {
    case 1: hipLaunchKernelGGL(k_gen_bsk<1>, bgrid, dim3(64), kStageLdsBytes, ctx->stream, d_k0, d_spec, ctx->d_tw, ctx->K->d_bsk, P.bgbit, alpha_bsk, d_rk, key_scale(ctx->fast_round)); break;
    case 2: hipLaunchKernelGGL(k_gen_bsk<2>, bgrid, dim3(64), kStageLdsBytes, ctx->stream, d_k0, d_spec, ctx->d_tw, ctx->K->d_bsk, P.bgbit, alpha_bsk, d_rk, key_scale(ctx->fast_round)); break;
    default: hipLaunchKernelGGL(k_gen_bsk<3>, bgrid, dim3(64), kStageLdsBytes, ctx->stream, d_k0, d_spec, ctx->d_tw, ctx->K->d_bsk, P.bgbit, alpha_bsk, d_rk, key_scale(ctx->fast_round)); break;
  }
  HIPCHK(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_gen_ksk, dim3((unsigned)((size_t)kN * P.t * base)), dim3(256), 0, ctx->stream, d_k0, d_k1,
                     ctx->K->d_ksk, P.n, P.basebit, P.t, alpha_ksk, d_rk);
  HIPCHK(ctx, hipGetLastError());
  // decomposition offset (key.rs:78-89) and test vector (key.rs:91-100)
  uint32_t off = 0;
  for (int i = 0; i < P.l; ++i) off += ((1u << P.bgbit) / 2) * (1u << (32 - (i + 1) * P.bgbit));
  std::vector<uint32_t> tv(2 * kN, 0u);
  for (int i = 0; i < kN; ++i) tv[kN + i] = 0x20000000u;  // f64_to_torus(0.125)
  HIPCHK(ctx, hipMemcpyAsync(ctx->K->d_testvec, tv.data(), 2 * kN * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (the local test vector above is read by an asynchronous copy)
  CHK(build_ksk_planes(ctx));
  ctx->K->offset = off;
  ctx->K->key_loaded = true;
  comb_prepare(ctx);
  return TFHE_HIP_OK;
}

// 64-bit seed -> 256-bit generator key (SplitMix64): reproducible, and only as strong as the seed
ChaChaKey key_from_seed(uint64_t seed) {
  ChaChaKey k;
  uint64_t x = seed;
  for (int i = 0; i < 4; ++i) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k.k[2 * i] = (uint32_t)z;
    k.k[2 * i + 1] = (uint32_t)(z >> 32);
  }
  return k;
}
}  // namespace

int tfhe_hip_gen_cloud_key(tfhe_hip_ctx *ctx, const uint32_t *key_lv0, const uint32_t *key_lv1, double alpha_ksk,
                           double alpha_bsk, uint64_t seed) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  return gen_cloud_key_locked(ctx, key_lv0, key_lv1, alpha_ksk, alpha_bsk, key_from_seed(seed));
}

int tfhe_hip_gen_cloud_key_with_key(tfhe_hip_ctx *ctx, const uint32_t *key_lv0, const uint32_t *key_lv1,
                                    double alpha_ksk, double alpha_bsk, const uint8_t rng_key[32]) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (!rng_key) return fail(ctx, TFHE_HIP_EINVAL, "null generator key");
  ChaChaKey k;
  memcpy(k.k, rng_key, 32);
  const int rc = gen_cloud_key_locked(ctx, key_lv0, key_lv1, alpha_ksk, alpha_bsk, k);
  volatile uint32_t *wipe = k.k;
  for (int i = 0; i < 8; ++i) wipe[i] = 0;
  return rc;
}

int tfhe_hip_gen_cloud_key_secure(tfhe_hip_ctx *ctx, const uint32_t *key_lv0, const uint32_t *key_lv1,
                                  double alpha_ksk, double alpha_bsk) {
  if (!ctx) return TFHE_HIP_EINVAL;
  uint8_t buf[32];
  size_t got = 0;
  while (got < sizeof(buf)) {  // the kernel's CSPRNG, as the reference's thread_rng is seeded (OsRng)
    const ssize_t r = getrandom(buf + got, sizeof(buf) - got, 0);
    if (r < 0) {
      if (errno == EINTR) continue;
      tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;
      std::lock_guard<FairMutex> lk(base->mu);
      return fail(base, TFHE_HIP_EHIP, std::string("getrandom: ") + strerror(errno));
    }
    got += (size_t)r;
  }
  const int rc = tfhe_hip_gen_cloud_key_with_key(ctx, key_lv0, key_lv1, alpha_ksk, alpha_bsk, buf);
  volatile uint8_t *wipe = buf;
  for (size_t i = 0; i < sizeof(buf); ++i) wipe[i] = 0;
  return rc;
}


int tfhe_hip_export_cloud_key(tfhe_hip_ctx *ctx, double *bsk, uint32_t *ksk, uint32_t *decomp_offset,
                              uint32_t *testvec) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  const tfhe_hip_params &P = ctx->P;
  const int base = 1 << P.basebit;
  if (bsk) {
    const size_t polys = (size_t)P.n * 2 * P.l * 2;
    CHK(ensure(ctx, ctx->h_out, polys * kN * sizeof(double)));
    hipLaunchKernelGGL(k_bsk_export, dim3((unsigned)polys), dim3(512), 0, ctx->stream, ctx->K->d_bsk, (double *)ctx->h_out.p, polys, 1.0 / key_scale(ctx->fast_round));
    HIPCHK(ctx, hipGetLastError());
    CHK(to_host(ctx, bsk, ctx->h_out, polys * kN * sizeof(double)));
  }
  if (ksk) {
    const size_t rows = (size_t)kN * P.t * base;
    CHK(ensure(ctx, ctx->h_out, rows * (size_t)(P.n + 1) * 4));
    hipLaunchKernelGGL(k_ksk_export, dim3((unsigned)rows), dim3(256), 0, ctx->stream, ctx->K->d_ksk, (uint32_t *)ctx->h_out.p, P.n, rows);
    HIPCHK(ctx, hipGetLastError());
    CHK(to_host(ctx, ksk, ctx->h_out, rows * (size_t)(P.n + 1) * 4));
  }
  if (decomp_offset) *decomp_offset = ctx->K->offset;
  if (testvec) {
    HIPCHK(ctx, hipMemcpyAsync(testvec, ctx->K->d_testvec, 2 * kN * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return TFHE_HIP_OK;
}

int tfhe_hip_cloud_key_buffers(tfhe_hip_ctx *ctx, void **bsk, size_t *bsk_bytes, void **ksk, size_t *ksk_bytes,
                               void **testvec, size_t *testvec_bytes, uint32_t *decomp_offset) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  const tfhe_hip_params &P = ctx->P;
  const size_t bb = (size_t)P.n * 2 * P.l * 2 * kN * sizeof(double);
  const size_t kb = (size_t)kN * P.t * (1u << P.basebit) * ksk_row_words(P.n) * 4;
  if (!ctx->K->d_bsk) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_bsk, bb));
  if (!ctx->K->d_ksk) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_ksk, kb + 4096));
  if (!ctx->K->d_testvec) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_testvec, 2 * kN * 4));
  if (bsk) *bsk = ctx->K->d_bsk;
  if (bsk_bytes) *bsk_bytes = bb;
  if (ksk) *ksk = ctx->K->d_ksk;
  if (ksk_bytes) *ksk_bytes = kb;
  if (testvec) *testvec = ctx->K->d_testvec;
  if (testvec_bytes) *testvec_bytes = 2 * kN * 4;
  if (decomp_offset) *decomp_offset = ctx->K->offset;
  return TFHE_HIP_OK;
}

int tfhe_hip_adopt_cloud_key(tfhe_hip_ctx *ctx, uint32_t decomp_offset) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (!ctx->K->d_bsk || !ctx->K->d_ksk || !ctx->K->d_testvec)
    return fail(ctx, TFHE_HIP_EINVAL, "tfhe_hip_adopt_cloud_key before tfhe_hip_cloud_key_buffers");
  // whatever filled the buffers (a peer copy, an RCCL broadcast on another stream) must have finished
  comb_quiesce(ctx);
  HIPCHK(ctx, hipDeviceSynchronize());
  CHK(build_ksk_planes(ctx));
  ctx->K->offset = decomp_offset;
  ctx->K->key_loaded = true;
  comb_prepare(ctx);
  ctx->K->reenc_loaded = false;
  return TFHE_HIP_OK;
}

// ---- device-pointer entry points ---------------------------------------------

int tfhe_hip_batch_gate_dev(tfhe_hip_ctx *ctx, int gate, const uint32_t *a, const uint32_t *b,
                            uint32_t *out, size_t count, void *stream) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count && (!a || !out)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  return gate_dev(ctx, gate, a, b, out, count, pick(ctx, stream));
}

int tfhe_hip_batch_gates_mixed_dev(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                   uint32_t *out, size_t count, void *stream) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count && (!gates || !a || !b || !out)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  return gates_mixed_dev(ctx, gates, a, b, out, count, pick(ctx, stream));
}

int tfhe_hip_batch_gates_mixed_nks_dev(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                       uint32_t *out, size_t count, void *stream) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count && (!gates || !a || !b || !out)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  return gates_mixed_nks_dev(ctx, gates, a, b, out, count, pick(ctx, stream));
}

int tfhe_hip_batch_bootstrap_dev(tfhe_hip_ctx *ctx, const uint32_t *in, const uint32_t *testvec,
                                 int per_ct, int keyswitch, uint32_t *out, size_t count,
                                 void *stream) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count && (!in || !out)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  return bootstrap_dev(ctx, in, testvec, per_ct, keyswitch, out, count, pick(ctx, stream));
}

int tfhe_hip_batch_tlwe_lincomb_dev(tfhe_hip_ctx *ctx, uint32_t ca, const uint32_t *a, uint32_t cb,
                                    const uint32_t *b, uint32_t cconst, uint32_t *out, size_t count,
                                    void *stream) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (count && (!a || !out || (cb && !b))) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  return lincomb_dev(ctx, GatePrep{ca, cb, cconst}, a, b, out, count, pick(ctx, stream));
}

int tfhe_hip_batch_lincomb_bootstrap_dev(tfhe_hip_ctx *ctx, uint32_t ca, const uint32_t *a, uint32_t cb,
                                         const uint32_t *b, uint32_t cconst, const uint32_t *testvec,
                                         int per_ct, int keyswitch, uint32_t *out, size_t count,
                                         void *stream) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count && (!a || !out || (cb && !b))) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  return lincomb_bootstrap_dev(ctx, GatePrep{ca, cb, cconst}, a, cb ? b : nullptr, testvec, per_ct, keyswitch, out,
                               count, pick(ctx, stream));
}

int tfhe_hip_batch_blind_rotate_dev(tfhe_hip_ctx *ctx, const uint32_t *in, const uint32_t *testvec,
                                    uint32_t *out_trlwe, size_t count, void *stream) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count && (!in || !out_trlwe)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  GatePrep gp;
  gate_prep(TFHE_HIP_COPY, gp);
  return launch_blind_rotate(ctx, pick(ctx, stream), in, nullptr, gp, testvec, 0, count, out_trlwe,
                             nullptr, nullptr);
}

int tfhe_hip_batch_mux_dev(tfhe_hip_ctx *ctx, int naive, const uint32_t *a, const uint32_t *b,
                           const uint32_t *c, uint32_t *out, size_t count, void *stream) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count && (!a || !b || !c || !out)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  return mux_dev(ctx, naive, a, b, c, out, count, pick(ctx, stream));
}

// ---- host-pointer entry points -----------------------------------------------

// Small calls (comb_takes) go through the combining front end: the caller checks its own arguments, exactly as the
// direct path below does and in the same order, then queues (combine.hpp).
namespace {
#define COMB_FAIL(base, code, msg)      \
  do {                                  \
    err_slot((base)->id) = (msg);       \
    return (code);                      \
  } while (0)
int comb_gate_call(tfhe_hip_ctx *ctx, int gate, const uint8_t *codes, int keyswitch, const uint32_t *a, const uint32_t *b,
                   const uint32_t *testvec, int per_ct, uint32_t *out, size_t count, const GatePrep *lin = nullptr) {
  tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;
  if (!ctx->own.key_loaded) COMB_FAIL(base, TFHE_HIP_ENOKEY, "cloud key not loaded");
  if (codes) {
    if (!a || !b || !out) COMB_FAIL(base, TFHE_HIP_EINVAL, "null pointer");
    for (size_t i = 0; i < count; ++i)
      if (codes[i] > TFHE_HIP_COPY) COMB_FAIL(base, TFHE_HIP_EINVAL, "unknown gate");
  } else {
    GatePrep gp;
    if (!gate_prep(gate, gp)) COMB_FAIL(base, TFHE_HIP_EINVAL, "unknown gate");
    if (!a || !out || (gp.cb && !b)) COMB_FAIL(base, TFHE_HIP_EINVAL, "null pointer");
  }
  if (!keyswitch && base->P.n > kN)
    COMB_FAIL(base, TFHE_HIP_EINVAL, "bootstrap without key switch needs n <= N (sample_extract_index_2)");
  CombReq r;
  r.key = &ctx->own;
  r.cls = CB_GATES;
  r.gate = gate;
  r.codes = codes;
  r.keyswitch = keyswitch;
  r.a = a;
  r.b = b;
  r.testvec = testvec;
  r.per_ct = per_ct;
  r.out = out;
  r.count = count;
  if (lin) {
    r.lin = true;
    r.lin_ca = lin->ca;
    r.lin_cb = lin->cb;
    r.lin_cconst = lin->cconst;
  }
  return comb_submit(base, r);
}
int comb_mux_call(tfhe_hip_ctx *ctx, int naive, const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *out,
                  size_t count) {
  tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;
  if (!ctx->own.key_loaded) COMB_FAIL(base, TFHE_HIP_ENOKEY, "cloud key not loaded");
  if (!a || !b || !c || !out) COMB_FAIL(base, TFHE_HIP_EINVAL, "null pointer");
  if (!naive && base->P.n > kN)
    COMB_FAIL(base, TFHE_HIP_EINVAL, "bootstrap without key switch needs n <= N (sample_extract_index_2)");
  CombReq r;
  r.key = &ctx->own;
  r.cls = naive ? CB_MUX_NAIVE : CB_MUX;
  r.a = a;
  r.b = b;
  r.c = c;
  r.out = out;
  r.count = count;
  return comb_submit(base, r);
}
#undef COMB_FAIL
}  // namespace

int tfhe_hip_batch_gate(tfhe_hip_ctx *ctx, int gate, const uint32_t *a, const uint32_t *b,
                        uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  if (comb_takes(ctx, count)) return comb_gate_call(ctx, gate, nullptr, 1, a, b, nullptr, 0, out, count);
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  GatePrep gp;
  if (!gate_prep(gate, gp)) return fail(ctx, TFHE_HIP_EINVAL, "unknown gate");
  if (!a || !out || (gp.cb && !b)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  const size_t bytes = count * (size_t)(ctx->P.n + 1) * 4;
  {  // all operands pinned: no staging (see pinned_view)
    const uint32_t *da = pinned_view(a, bytes), *db = gp.cb ? pinned_view(b, bytes) : nullptr;
    uint32_t *dout = pinned_view(out, bytes);
    if (da && dout && (!gp.cb || db)) {
      CHK(gate_dev(ctx, gate, da, db, dout, count, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      return TFHE_HIP_OK;
    }
  }
  CHK(to_dev(ctx, ctx->h_a, a, bytes));
  if (gp.cb) CHK(to_dev(ctx, ctx->h_b, b, bytes));
  CHK(ensure(ctx, ctx->h_out, bytes));
  CHK(gate_dev(ctx, gate, (uint32_t *)ctx->h_a.p, (uint32_t *)ctx->h_b.p, (uint32_t *)ctx->h_out.p, count, ctx->stream));
  return to_host(ctx, out, ctx->h_out, bytes);
}

int tfhe_hip_batch_gates_mixed(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                               uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  if (comb_takes(ctx, count) && gates) return comb_gate_call(ctx, 0, gates, 1, a, b, nullptr, 0, out, count);
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  if (!gates || !a || !b || !out) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  for (size_t i = 0; i < count; ++i)
    if (gates[i] > TFHE_HIP_COPY) return fail(ctx, TFHE_HIP_EINVAL, "unknown gate");
  const size_t bytes = count * (size_t)(ctx->P.n + 1) * 4;
  CHK(to_dev(ctx, ctx->h_idx, gates, count));  // one byte per ciphertext: always staged
  {
    const uint32_t *da = pinned_view(a, bytes), *db = pinned_view(b, bytes);
    uint32_t *dout = pinned_view(out, bytes);
    if (da && db && dout) {
      CHK(gates_mixed_dev(ctx, (const uint8_t *)ctx->h_idx.p, da, db, dout, count, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      return TFHE_HIP_OK;
    }
  }
  CHK(to_dev(ctx, ctx->h_a, a, bytes));
  CHK(to_dev(ctx, ctx->h_b, b, bytes));
  CHK(ensure(ctx, ctx->h_out, bytes));
  CHK(gates_mixed_dev(ctx, (const uint8_t *)ctx->h_idx.p, (uint32_t *)ctx->h_a.p, (uint32_t *)ctx->h_b.p,
                      (uint32_t *)ctx->h_out.p, count, ctx->stream));
  return to_host(ctx, out, ctx->h_out, bytes);
}

int tfhe_hip_batch_gates_mixed_nks(tfhe_hip_ctx *ctx, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                   uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  if (comb_takes(ctx, count) && gates) return comb_gate_call(ctx, 0, gates, 0, a, b, nullptr, 0, out, count);
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  if (!gates || !a || !b || !out) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  for (size_t i = 0; i < count; ++i)
    if (gates[i] > TFHE_HIP_COPY) return fail(ctx, TFHE_HIP_EINVAL, "unknown gate");
  const size_t bytes = count * (size_t)(ctx->P.n + 1) * 4;
  CHK(to_dev(ctx, ctx->h_a, a, bytes));
  CHK(to_dev(ctx, ctx->h_b, b, bytes));
  CHK(to_dev(ctx, ctx->h_idx, gates, count));
  CHK(ensure(ctx, ctx->h_out, bytes));
  CHK(gates_mixed_nks_dev(ctx, (const uint8_t *)ctx->h_idx.p, (const uint32_t *)ctx->h_a.p, (const uint32_t *)ctx->h_b.p,
                          (uint32_t *)ctx->h_out.p, count, ctx->stream));
  return to_host(ctx, out, ctx->h_out, bytes);
}

int tfhe_hip_batch_bootstrap(tfhe_hip_ctx *ctx, const uint32_t *in, const uint32_t *testvec,
                             int per_ct, int keyswitch, uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  if (comb_takes(ctx, count))
    return comb_gate_call(ctx, TFHE_HIP_COPY, nullptr, keyswitch != 0, in, nullptr, testvec, per_ct, out, count);
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  if (!in || !out) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  const size_t bytes = count * (size_t)(ctx->P.n + 1) * 4;
  const uint32_t *d_tv = nullptr;
  if (testvec) {
    d_tv = pinned_view(testvec, (per_ct ? count : 1) * (size_t)2 * kN * 4);
    if (!d_tv) {
      CHK(to_dev(ctx, ctx->h_tv, testvec, (per_ct ? count : 1) * (size_t)2 * kN * 4));
      d_tv = (const uint32_t *)ctx->h_tv.p;
    }
  }
  {
    const uint32_t *din = pinned_view(in, bytes);
    uint32_t *dout = pinned_view(out, bytes);
    if (din && dout) {
      CHK(bootstrap_dev(ctx, din, d_tv, per_ct, keyswitch, dout, count, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      return TFHE_HIP_OK;
    }
  }
  CHK(to_dev(ctx, ctx->h_a, in, bytes));
  CHK(ensure(ctx, ctx->h_out, bytes));
  CHK(bootstrap_dev(ctx, (uint32_t *)ctx->h_a.p, d_tv, per_ct, keyswitch, (uint32_t *)ctx->h_out.p, count, ctx->stream));
  return to_host(ctx, out, ctx->h_out, bytes);
}

int tfhe_hip_batch_tlwe_lincomb(tfhe_hip_ctx *ctx, uint32_t ca, const uint32_t *a, uint32_t cb,
                                const uint32_t *b, uint32_t cconst, uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (count == 0) return TFHE_HIP_OK;
  if (!a || !out || (cb && !b)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  const size_t bytes = count * (size_t)(ctx->P.n + 1) * 4;
  CHK(to_dev(ctx, ctx->h_a, a, bytes));
  if (cb) CHK(to_dev(ctx, ctx->h_b, b, bytes));
  CHK(ensure(ctx, ctx->h_out, bytes));
  CHK(lincomb_dev(ctx, GatePrep{ca, cb, cconst}, (const uint32_t *)ctx->h_a.p, (const uint32_t *)ctx->h_b.p,
                  (uint32_t *)ctx->h_out.p, count, ctx->stream));
  return to_host(ctx, out, ctx->h_out, bytes);
}

int tfhe_hip_batch_lincomb_bootstrap(tfhe_hip_ctx *ctx, uint32_t ca, const uint32_t *a, uint32_t cb,
                                     const uint32_t *b, uint32_t cconst, const uint32_t *testvec,
                                     int per_ct, int keyswitch, uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  if (comb_takes(ctx, count) && a && out && (!cb || b)) {
    // small call: it joins the merged launches as a COPY bootstrap whose rows are formed on the device first
    // (k_tlwe_lincomb over the packed rows, one launch per run of requests with the same coefficients)
    const GatePrep lin{ca, cb, cconst};
    return comb_gate_call(ctx, TFHE_HIP_COPY, nullptr, keyswitch != 0, a, cb ? b : nullptr, testvec, per_ct, out, count, &lin);
  }
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  if (!a || !out || (cb && !b)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  const size_t bytes = count * (size_t)(ctx->P.n + 1) * 4;
  CHK(to_dev(ctx, ctx->h_a, a, bytes));
  if (cb) CHK(to_dev(ctx, ctx->h_b, b, bytes));
  const uint32_t *d_tv = nullptr;
  if (testvec) {
    CHK(to_dev(ctx, ctx->h_tv, testvec, (per_ct ? count : 1) * (size_t)2 * kN * 4));
    d_tv = (const uint32_t *)ctx->h_tv.p;
  }
  CHK(ensure(ctx, ctx->h_out, bytes));
  CHK(lincomb_bootstrap_dev(ctx, GatePrep{ca, cb, cconst}, (const uint32_t *)ctx->h_a.p,
                            cb ? (const uint32_t *)ctx->h_b.p : nullptr, d_tv, per_ct, keyswitch,
                            (uint32_t *)ctx->h_out.p, count, ctx->stream));
  return to_host(ctx, out, ctx->h_out, bytes);
}

int tfhe_hip_batch_blind_rotate(tfhe_hip_ctx *ctx, const uint32_t *in, const uint32_t *testvec,
                                uint32_t *out_trlwe, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  if (comb_takes(ctx, count) && in && out_trlwe) {  // small call: merged with the other threads' (combine.hpp)
    tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;
    if (!ctx->own.key_loaded) {
      err_slot(base->id) = "cloud key not loaded";
      return TFHE_HIP_ENOKEY;
    }
    CombReq r;
    r.key = &ctx->own;
    r.cls = CB_ROTATE;
    r.a = in;
    r.testvec = testvec;
    r.out = out_trlwe;
    r.count = count;
    return comb_submit(base, r);
  }
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  if (!in || !out_trlwe) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  CHK(to_dev(ctx, ctx->h_a, in, count * (size_t)(ctx->P.n + 1) * 4));
  const uint32_t *d_tv = nullptr;
  if (testvec) {
    CHK(to_dev(ctx, ctx->h_tv, testvec, (size_t)2 * kN * 4));
    d_tv = (const uint32_t *)ctx->h_tv.p;
  }
  const size_t obytes = count * (size_t)2 * kN * 4;
  CHK(ensure(ctx, ctx->h_out, obytes));
  GatePrep gp;
  gate_prep(TFHE_HIP_COPY, gp);
  CHK(launch_blind_rotate(ctx, ctx->stream, (uint32_t *)ctx->h_a.p, nullptr, gp, d_tv, 0, count,
                          (uint32_t *)ctx->h_out.p, nullptr, nullptr));
  return to_host(ctx, out_trlwe, ctx->h_out, obytes);
}

int tfhe_hip_batch_mux(tfhe_hip_ctx *ctx, int naive, const uint32_t *a, const uint32_t *b,
                       const uint32_t *c, uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  if (comb_takes(ctx, count)) return comb_mux_call(ctx, naive, a, b, c, out, count);
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  if (!a || !b || !c || !out) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  const size_t bytes = count * (size_t)(ctx->P.n + 1) * 4;
  {
    const uint32_t *da = pinned_view(a, bytes), *db = pinned_view(b, bytes), *dc = pinned_view(c, bytes);
    uint32_t *dout = pinned_view(out, bytes);
    if (da && db && dc && dout) {
      CHK(mux_dev(ctx, naive, da, db, dc, dout, count, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      return TFHE_HIP_OK;
    }
  }
  CHK(to_dev(ctx, ctx->h_a, a, bytes));
  CHK(to_dev(ctx, ctx->h_b, b, bytes));
  CHK(to_dev(ctx, ctx->h_c, c, bytes));
  CHK(ensure(ctx, ctx->h_out, bytes));
  CHK(mux_dev(ctx, naive, (uint32_t *)ctx->h_a.p, (uint32_t *)ctx->h_b.p, (uint32_t *)ctx->h_c.p,
              (uint32_t *)ctx->h_out.p, count, ctx->stream));
  return to_host(ctx, out, ctx->h_out, bytes);
}

// ---- single stages --------------------------------------------------------------

int tfhe_hip_batch_external_product(tfhe_hip_ctx *ctx, const uint32_t *trlwe_in,
                                    const int32_t *bsk_index, uint32_t *trlwe_out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  if (!trlwe_in || !bsk_index || !trlwe_out) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  for (size_t i = 0; i < count; ++i)
    if (bsk_index[i] < 0 || bsk_index[i] >= ctx->P.n) return fail(ctx, TFHE_HIP_EINVAL, "bsk_index out of range");
  const size_t bytes = count * (size_t)2 * kN * 4;
  CHK(to_dev(ctx, ctx->h_a, trlwe_in, bytes));
  CHK(to_dev(ctx, ctx->h_idx, bsk_index, count * 4));
  CHK(ensure(ctx, ctx->h_out, bytes));
  dim3 grid((unsigned)count), block(64);
  const uint32_t *in = (const uint32_t *)ctx->h_a.p;
  const int32_t *idx = (const int32_t *)ctx->h_idx.p;
  uint32_t *o = (uint32_t *)ctx->h_out.p;
  const uint32_t bsk_bytes = (uint32_t)((size_t)ctx->P.n * 2 * ctx->P.l * 2 * kN2 * 16);
  hipLaunchKernelGGL(ep_kernel(ctx), grid, block, kStageLdsBytes, ctx->stream, in, idx, ctx->K->d_bsk, bsk_bytes, ctx->d_tw,
                     ctx->P.bgbit, ctx->K->offset, o);
  HIPCHK(ctx, hipGetLastError());
  return to_host(ctx, trlwe_out, ctx->h_out, bytes);
}

int tfhe_hip_batch_sample_extract(tfhe_hip_ctx *ctx, const uint32_t *trlwe, int k, uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (k < 0 || k >= kN) return fail(ctx, TFHE_HIP_EINVAL, "extraction index out of range");
  if (count == 0) return TFHE_HIP_OK;
  if (!trlwe || !out) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  CHK(to_dev(ctx, ctx->h_a, trlwe, count * (size_t)2 * kN * 4));
  const size_t obytes = count * (size_t)(kN + 1) * 4;
  CHK(ensure(ctx, ctx->h_out, obytes));
  hipLaunchKernelGGL(k_sample_extract, dim3((unsigned)count), dim3(256), 0, ctx->stream,
                     (const uint32_t *)ctx->h_a.p, k, (uint32_t *)ctx->h_out.p, count);
  HIPCHK(ctx, hipGetLastError());
  return to_host(ctx, out, ctx->h_out, obytes);
}

int tfhe_hip_batch_identity_key_switch(tfhe_hip_ctx *ctx, const uint32_t *tlwe_lv1, uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  if (!tlwe_lv1 || !out) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  CHK(ensure(ctx, ctx->h_a, lv1_rows(count) * (size_t)(kN + 1) * 4));  // whole 256-row groups readable (k_key_switch_mfma)
  CHK(to_dev(ctx, ctx->h_a, tlwe_lv1, count * (size_t)(kN + 1) * 4));
  const size_t obytes = count * (size_t)(ctx->P.n + 1) * 4;
  CHK(ensure(ctx, ctx->h_out, obytes));
  CHK(launch_key_switch(ctx, ctx->stream, (const uint32_t *)ctx->h_a.p, (uint32_t *)ctx->h_out.p, count));
  return to_host(ctx, out, ctx->h_out, obytes);
}

// ---- proxy re-encryption (src/proxy_reenc.rs; feature `proxy-reenc` of the reference) -------------------------------
// key [n][t][base][n+1]: ProxyReencryptionKey::key_encryptions (proxy_reenc.rs:224-233, index base*t*i + base*j + k).
// It is stored as a key-switching key whose coefficients n .. N-1 have all-zero rows (never selected: k_reenc_pad).
int tfhe_hip_load_reenc_key(tfhe_hip_ctx *ctx, const uint32_t *key) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (!key) return fail(ctx, TFHE_HIP_EINVAL, "null key pointer");
  const tfhe_hip_params &P = ctx->P;
  // the source rides in the key switch's N-coefficient rows (k_reenc_pad): SECURITY_UINT5 .. 8 (n = 1071 / 1160) do not fit
  if (P.n > kN) return fail(ctx, TFHE_HIP_EINVAL, "proxy re-encryption needs n <= N = 1024 (this parameter set's n is larger)");
  if (ctx->scratch_owned) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scratch_owned = false;
  comb_quiesce(ctx);  // ... and merged launches on the front end's lanes
  const int base = 1 << P.basebit;
  const size_t eng_bytes = (size_t)kN * P.t * base * ksk_row_words(P.n) * 4;
  const size_t rows = (size_t)P.n * P.t * base, words = rows * (size_t)(P.n + 1);
  ctx->K->key_loaded = ctx->K->reenc_loaded = false;  // the buffer is shared with a cloud key's key-switching key
  if (!ctx->K->d_ksk) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_ksk, eng_bytes + 4096));
  uint32_t *d_ref = nullptr;
  HIPCHK(ctx, hipMalloc((void **)&d_ref, words * 4));
  hipError_t e = hipMemsetAsync(ctx->K->d_ksk, 0, eng_bytes, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_ref, key, words * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_ksk_convert, dim3((unsigned)rows), dim3(256), 0, ctx->stream, d_ref, ctx->K->d_ksk, P.n, base, rows);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_ref);
  if (e != hipSuccess) return fail(ctx, TFHE_HIP_EHIP, std::string("re-encryption key upload: ") + hipGetErrorString(e));
  CHK(build_ksk_planes(ctx));
  ctx->K->reenc_loaded = true;
  return TFHE_HIP_OK;
}

int tfhe_hip_reenc_key_is_loaded(tfhe_hip_ctx *ctx) {  // 0 / 1, never an error code (no device call is made)
  if (!ctx) return 0;
  return __atomic_load_n(&ctx->own.reenc_loaded, __ATOMIC_ACQUIRE) ? 1 : 0;  // (no lock: see tfhe_hip_key_is_loaded)
}

int tfhe_hip_batch_reencrypt_dev(tfhe_hip_ctx *ctx, const uint32_t *in, uint32_t *out, size_t count, void *stream) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_reenc_key(ctx));
  if (count && (!in || !out)) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  return reencrypt_dev(ctx, in, out, count, pick(ctx, stream));
}

int tfhe_hip_batch_reencrypt(tfhe_hip_ctx *ctx, const uint32_t *in, uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  CHK(need_reenc_key(ctx));
  if (count == 0) return TFHE_HIP_OK;
  if (!in || !out) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  const size_t bytes = count * (size_t)(ctx->P.n + 1) * 4;
  CHK(to_dev(ctx, ctx->h_a, in, bytes));
  CHK(ensure(ctx, ctx->h_out, bytes));
  CHK(reencrypt_dev(ctx, (const uint32_t *)ctx->h_a.p, (uint32_t *)ctx->h_out.p, count, ctx->stream));
  return to_host(ctx, out, ctx->h_out, bytes);
}

int tfhe_hip_batch_ifft(tfhe_hip_ctx *ctx, double *res, const uint32_t *src, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (count == 0) return TFHE_HIP_OK;
  if (!res || !src) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  CHK(to_dev(ctx, ctx->h_a, src, count * (size_t)kN * 4));
  CHK(ensure(ctx, ctx->h_out, count * (size_t)kN * 8));
  hipLaunchKernelGGL(k_ifft, dim3((unsigned)count), dim3(64), kStageLdsBytes, ctx->stream,
                     (const uint32_t *)ctx->h_a.p, ctx->d_tw, (double *)ctx->h_out.p);
  HIPCHK(ctx, hipGetLastError());
  return to_host(ctx, res, ctx->h_out, count * (size_t)kN * 8);
}

int tfhe_hip_batch_fft(tfhe_hip_ctx *ctx, uint32_t *res, const double *src, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (count == 0) return TFHE_HIP_OK;
  if (!res || !src) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  CHK(to_dev(ctx, ctx->h_a, src, count * (size_t)kN * 8));
  CHK(ensure(ctx, ctx->h_out, count * (size_t)kN * 4));
  hipLaunchKernelGGL(k_fft, dim3((unsigned)count), dim3(64), kStageLdsBytes, ctx->stream,
                     (const double *)ctx->h_a.p, ctx->d_tw, (uint32_t *)ctx->h_out.p);
  HIPCHK(ctx, hipGetLastError());
  return to_host(ctx, res, ctx->h_out, count * (size_t)kN * 4);
}

int tfhe_hip_batch_poly_mul(tfhe_hip_ctx *ctx, uint32_t *res, const uint32_t *a, const uint32_t *b, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (count == 0) return TFHE_HIP_OK;
  if (!res || !a || !b) return fail(ctx, TFHE_HIP_EINVAL, "null pointer");
  const size_t bytes = count * (size_t)kN * 4;
  CHK(to_dev(ctx, ctx->h_a, a, bytes));
  CHK(to_dev(ctx, ctx->h_b, b, bytes));
  CHK(ensure(ctx, ctx->h_out, bytes));
  hipLaunchKernelGGL(k_poly_mul, dim3((unsigned)count), dim3(64), kStageLdsBytes, ctx->stream,
                     (const uint32_t *)ctx->h_a.p, (const uint32_t *)ctx->h_b.p, ctx->d_tw,
                     (uint32_t *)ctx->h_out.p);
  HIPCHK(ctx, hipGetLastError());
  return to_host(ctx, res, ctx->h_out, bytes);
}

// ---- measurement ---------------------------------------------------------------

int tfhe_hip_set_profiling(tfhe_hip_ctx *ctx, int enabled) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (enabled && !ctx->profiling) {
    HIPCHK(ctx, hipMemsetAsync(ctx->d_diag, 0, 16, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_diag + 4, 0, 16, ctx->stream));
  }
  ctx->profiling = enabled != 0;
  comb_with_idle_lanes(ctx, [&](Combiner &C) {  // the front end's lanes record their launches too
    C.profiling = enabled != 0;
    for (tfhe_hip_ctx *x : C.lane_ctx)
      if (x) x->profiling = enabled != 0;
  });
  return TFHE_HIP_OK;
}

int tfhe_hip_get_kernel_times(tfhe_hip_ctx *ctx, tfhe_hip_kernel_times *out) {
  if (!ctx || !out) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  memset(out, 0, sizeof(*out));
  auto drain = [&](std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, double &ms, uint64_t &cnt) -> int {
    for (auto &p : v) {
      HIPCHK(ctx, hipEventSynchronize(p.second));
      float t = 0.f;
      HIPCHK(ctx, hipEventElapsedTime(&t, p.first, p.second));
      ms += (double)t;
      ++cnt;
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
    v.clear();
    return TFHE_HIP_OK;
  };
  CHK(drain(ctx->ev_br, out->blind_rotate_ms, out->blind_rotate_launches));
  CHK(drain(ctx->ev_ks, out->key_switch_ms, out->key_switch_launches));
  out->bootstraps = ctx->bootstraps;
  ctx->bootstraps = 0;
  int lane_rc = TFHE_HIP_OK;  // merged launches of small calls ran on the front end's lanes
  comb_with_idle_lanes(ctx, [&](Combiner &C) {
    for (tfhe_hip_ctx *x : C.lane_ctx) {
      if (!x) continue;
      if (lane_rc == TFHE_HIP_OK) lane_rc = drain(x->ev_br, out->blind_rotate_ms, out->blind_rotate_launches);
      if (lane_rc == TFHE_HIP_OK) lane_rc = drain(x->ev_ks, out->key_switch_ms, out->key_switch_launches);
      out->bootstraps += x->bootstraps;
      x->bootstraps = 0;
    }
  });
  return lane_rc;
}

int tfhe_hip_get_clock_sample(tfhe_hip_ctx *ctx, tfhe_hip_clock_sample *out) {
  if (!ctx || !out) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (ctx->scratch_owned) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long h[2] = {0, 0};
  HIPCHK(ctx, hipMemcpy(h, ctx->d_diag, 16, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemset(ctx->d_diag, 0, 16));
  out->shader_cycles = h[0];
  out->rtc_ticks = h[1];
  out->rtc_mhz = ctx->rtc_khz / 1000.0;
  out->shader_mhz = h[1] ? (double)h[0] / (double)h[1] * out->rtc_mhz : 0.0;
  return TFHE_HIP_OK;
}

int tfhe_hip_get_key_switch_clock_sample(tfhe_hip_ctx *ctx, tfhe_hip_clock_sample *out) {
  if (!ctx || !out) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (ctx->scratch_owned) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long h[2] = {0, 0};
  HIPCHK(ctx, hipMemcpy(h, ctx->d_diag + 4, 16, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemset(ctx->d_diag + 4, 0, 16));
  out->shader_cycles = h[0];
  out->rtc_ticks = h[1];
  out->rtc_mhz = ctx->rtc_khz / 1000.0;
  out->shader_mhz = h[1] ? (double)h[0] / (double)h[1] * out->rtc_mhz : 0.0;
  return TFHE_HIP_OK;
}

int tfhe_hip_describe_dispatch(tfhe_hip_ctx *ctx, size_t count, char *buf, size_t buflen) {
  if (!ctx || !buf || buflen == 0) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  std::string d = "blind_rotate=";
  const BrPlan bp = plan_blind_rotate(ctx, count);
  for (int q = 0; q < bp.nparts; ++q) {
    if (q) d += "+";
    d += std::string(kBrKindName[bp.kind[q]]) + "[" + std::to_string(bp.begin[q]) + "," + std::to_string(bp.begin[q] + bp.count[q]) + ")";
  }
  if (bp.nparts == 0) d += "none";
  d += " key_switch=";
  if (count == 0) d += "none";
  else {
    const KsPlan kp = plan_key_switch(ctx, count);
    d += kKsKindName[kp.kind];
    if (kp.kind == KS_MFMA || kp.kind == KS_SLICED || kp.kind == KS_SPLIT) d += "(k=" + std::to_string(kp.kparts);
    if (kp.kind == KS_SLICED) d += ",sets=" + std::to_string(kp.sets);
    if (kp.kind == KS_MFMA || kp.kind == KS_SLICED || kp.kind == KS_SPLIT) d += ")";
  }
  if (d.size() + 1 > buflen) return fail(ctx, TFHE_HIP_EINVAL, "tfhe_hip_describe_dispatch: buffer too small");
  memcpy(buf, d.c_str(), d.size() + 1);
  return TFHE_HIP_OK;
}

const char *tfhe_hip_rounding_mode(const tfhe_hip_ctx *ctx) {
  if (!ctx) return "none";
  const tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;
  return base->fast_round ? "fast" : "general";
}

#ifdef TFHE_EXPERIMENT
// experiment builds only (profiles/exp/): the raw diagnostics words, e.g. the per-phase stamps of TFHE_LAT_STAMPS
extern "C" int tfhe_hip_experiment_diag(tfhe_hip_ctx *ctx, unsigned long long *out, size_t words) {
  if (!ctx || !out || words > 128) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(out, ctx->d_diag, words * 8, hipMemcpyDeviceToHost));
  return TFHE_HIP_OK;
}
#endif

int tfhe_hip_set_combining(tfhe_hip_ctx *ctx, size_t max_count) {
  if (!ctx) return TFHE_HIP_EINVAL;
  tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;
  if (max_count > Combiner::kBatchCap) {
    err_slot(base->id) = "tfhe_hip_set_combining: at most 4096 ciphertexts per merged call";
    return TFHE_HIP_EINVAL;
  }
  if (base->comb) base->comb->max_count.store(max_count, std::memory_order_relaxed);
  return TFHE_HIP_OK;
}

int tfhe_hip_get_combine_stats(tfhe_hip_ctx *ctx, tfhe_hip_combine_stats *out) {
  if (!ctx || !out) return TFHE_HIP_EINVAL;
  tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;
  memset(out, 0, sizeof(*out));
  Combiner *C = base->comb;
  if (!C) return TFHE_HIP_OK;
  std::lock_guard<std::mutex> lk(C->st_mu);
  out->max_count = C->max_count.load(std::memory_order_relaxed);
  out->launches = C->st_launches;
  out->requests = C->st_requests;
  out->ciphertexts = C->st_units;
  out->max_requests_per_launch = C->st_max_requests;
  out->lingers = C->st_lingers;
  out->linger_us = C->st_linger_us;
  out->pack_us = C->st_pack_us;
  out->gpu_us = C->st_gpu_us;
  out->unpack_us = C->st_unpack_us;
  C->st_launches = C->st_requests = C->st_units = C->st_max_requests = C->st_lingers = 0;
  C->st_linger_us = C->st_pack_us = C->st_gpu_us = C->st_unpack_us = 0;
  return TFHE_HIP_OK;
}

int tfhe_hip_host_alloc(size_t bytes, void **out) {
  if (!out) return TFHE_HIP_EINVAL;
  *out = nullptr;
  if (bytes == 0) return TFHE_HIP_OK;
  const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    g_create_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
    *out = nullptr;
    return e == hipErrorOutOfMemory ? TFHE_HIP_ENOMEM : TFHE_HIP_EHIP;
  }
  return TFHE_HIP_OK;
}

void tfhe_hip_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int tfhe_hip_synchronize(tfhe_hip_ctx *ctx) {
  if (!ctx) return TFHE_HIP_EINVAL;
  ENTER(ctx);
  if (ctx->scratch_owned && ctx->scratch_owner != ctx->stream) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scratch_owned = false;
  uint32_t flag = 0;
  HIPCHK(ctx, hipMemcpy(&flag, ctx->d_diag + 2, 4, hipMemcpyDeviceToHost));
  if (flag) {
    HIPCHK(ctx, hipMemset(ctx->d_diag + 2, 0, 8));
    return fail(ctx, TFHE_HIP_EINVAL, "a *_mixed_dev launch saw a gate code outside tfhe_hip_gate (treated as COPY)");
  }
  return TFHE_HIP_OK;
}

}  // extern "C"
#include "pool.hpp"
