// tfhe_hip.hip -- C ABI (include/tfhe_hip.h) over the gfx950 kernels.
//
// Host side of the engine: context (device, stream, converted cloud key,
// scratch), launch geometry, and the batched entry points that compose
//   gate prep + blind rotate (one persistent kernel)  ->  key switch (one kernel)
// exactly as gates::batch_* composes them in the reference
// (src/gates.rs:357-383: prepare, batch_blind_rotate, extract + key switch).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/tfhe_hip.h"
#include <errno.h>
#include <sys/random.h>

#include "blind_rotate.hpp"
#include "blind_rotate_wide.hpp"
#include "key_switch.hpp"
#include "key_switch_mfma.hpp"
#include "keygen.hpp"
#include "twiddles_host.hpp"

using namespace tfhe;

namespace {

thread_local std::string g_create_error = "";

constexpr int kKsG = 32;  // ciphertexts per key-switch workgroup

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};
struct PinBuf {  // pinned host arena (hipHostMalloc)
  void *p = nullptr;
  size_t cap = 0;
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
};

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
  DevBuf lv1, u1, u2, h_a, h_b, h_c, h_out, h_tv, h_idx, ks_out;  // scratch / host-API staging
  PinBuf p_a, p_b, p_c, p_out;  // pinned staging arenas behind h_a / h_b / h_c / h_out (pool members only)
  bool stage_pinned = false;     // set by a pool with several members: stage pageable operands through the arenas
  std::mutex mu;
  std::string err = "";
  bool profiling = false;
  int num_cus = 0;
  bool fast_round = false;  // |pre-rounding value| < 2^51 guaranteed (see round_to_torus<FAST>)
  bool ks_b4 = true;  // base-4 key switch streams candidate rows through an LDS ring (k_key_switch_b4)
  int ks_sliced = 1;  // wider bases: column-sliced LDS kernel (k_key_switch_sliced); 2 = also at base 4
  int ks_sliced_sets = 0;  // 0: accumulator sets per lane picked per launch (ks_sliced_pick_sets); else forced (24..40)
  int ks_mfma = 1;    // base 4: int8 matrix-core key switch (k_key_switch_mfma); 2 = at every batch size
  size_t ks_mfma_min = 64;   // smallest batch the matrix-core kernel takes (below: the split kernel)
  int ks_mfma_ksplit = 0;    // 0: K chunks per row block picked per launch; else forced (1, 2, 4, 8, 16)
  size_t ks_sl_chunk_min = 384;  // wider bases: smallest batch the column-sliced kernel takes (with K chunks; below: the split kernel)
  int ks_sl_kchunks = 0;     // 0: K chunks of the column-sliced kernel picked per launch; else forced (1 ... 64, a power of two)
  bool br_wide = true;      // small batches use the latency kernels
  bool br_wide2 = true;     // ... in their eight-wave form (blind_rotate_wide.hpp); false: one wave per row (round 1-2)
  size_t wide_max = 256;    // blind rotate: 2l waves per ciphertext up to this batch size (set from #CUs)
  size_t pair_lo = 0, pair_max = 0;  // ... two ciphertexts per eight-wave workgroup (k_blind_rotate_pair) for pair_lo < count <= pair_max
  size_t ks_split_max = 256;  // key switch: coefficient walk split over 32 workgroups up to this batch size
  long br_chunk = 0;  // blind-rotate workgroups per launch: 0 = whole batch (default), -1 = resident set, N = N
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_br, ev_ks;
  uint64_t bootstraps = 0;
  hipStream_t scratch_owner = nullptr;  // stream whose queued work may still use lv1/u1/u2
  bool scratch_owned = false;
  // device diagnostics: [0] shader cycles, [1] constant-rate ticks (both summed over blind-rotate
  // workgroups while profiling is on), [2] error flag raised by kernels (bad gate code), [4] / [5] the same two
  // sums for the matrix-core key switch
  unsigned long long *d_diag = nullptr;
  int rtc_khz = 100000;  // rate of s_memrealtime (hipDeviceAttributeWallClockRate)
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
  std::lock_guard<std::mutex> lk_((ctx)->mu);                                          \
  KeyBind kb_((ctx), &self_->own);                                                     \
  DeviceGuard dg_((ctx)->device);                                                      \
  if (dg_.err != hipSuccess) {                                                         \
    (ctx)->err = std::string("hipSetDevice: ") + hipGetErrorString(dg_.err);           \
    return TFHE_HIP_EHIP;                                                              \
  }

#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
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
  ctx->err = msg;
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
    ctx->err = std::string("hipMalloc scratch: ") + hipGetErrorString(e);
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

typedef void (*br_kernel_t)(BlindRotateArgs);
br_kernel_t br_kernel(const tfhe_hip_ctx *ctx) {
  const bool f = ctx->fast_round;
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

br_kernel_t br_wide_kernel(const tfhe_hip_ctx *ctx) {
  const bool f = ctx->fast_round;
  if (ctx->br_wide2) switch (ctx->P.l) {
      case 1: return f ? k_blind_rotate_wide2<1, true> : k_blind_rotate_wide2<1, false>;
      case 2: return f ? k_blind_rotate_wide2<2, true> : k_blind_rotate_wide2<2, false>;
      default: return f ? k_blind_rotate_wide2<3, true> : k_blind_rotate_wide2<3, false>;
    }
  switch (ctx->P.l) {
    case 1: return f ? k_blind_rotate_wide<1, true> : k_blind_rotate_wide<1, false>;
    case 2: return f ? k_blind_rotate_wide<2, true> : k_blind_rotate_wide<2, false>;
    default: return f ? k_blind_rotate_wide<3, true> : k_blind_rotate_wide<3, false>;
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

size_t br_lds_bytes(const tfhe_hip_ctx *ctx) { return blind_rotate_lds_bytes(ctx->P.n); }

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
  // Small batches (latency kernels).  Eight-wave form, N = #CUs: up to N ciphertexts one workgroup each (2.2 ms at
  // 128 bit); N < count <= 2N two ciphertexts per workgroup, half a step apart (3.6 ms; two rounds of the former take
  // 4.5); up to 3N the first 2N as pairs and the rest as singles (5.8; three rounds of singles 6.4, pairs alone 7.3,
  // the batch kernel 7.0 for anything up to 4N) -- profiles/exp/logs/r3x_pair_kernel.log.  wide_max / pair_lo /
  // pair_max hold those bounds.
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
  auto launch_pair = [&](const BlindRotateArgs &S) -> int {
    CHK(record_begin(ctx, s, ctx->ev_br));
    hipLaunchKernelGGL(br_pair_kernel(ctx), dim3((unsigned)((S.count + 1) / 2)), dim3(64u * kPairWaves),
                       blind_rotate_pair_lds_bytes(ctx->P.n), s, S);
    HIPCHK(ctx, hipGetLastError());
    return record_end(ctx, s, ctx->ev_br);
  };
  auto launch_wide = [&](const BlindRotateArgs &S) -> int {
    const size_t wlds = ctx->br_wide2 ? blind_rotate_wide2_lds_bytes(ctx->P.n, ctx->P.l) : blind_rotate_wide_lds_bytes(ctx->P.n, ctx->P.l);
    const unsigned wthreads = ctx->br_wide2 ? 64u * kWide2Waves : 128u * (unsigned)ctx->P.l;
    CHK(record_begin(ctx, s, ctx->ev_br));
    hipLaunchKernelGGL(br_wide_kernel(ctx), dim3((unsigned)S.count), dim3(wthreads), wlds, s, S);
    HIPCHK(ctx, hipGetLastError());
    return record_end(ctx, s, ctx->ev_br);
  };
  const bool pairs_on = ctx->br_wide && ctx->br_wide2 && ctx->pair_max > ctx->pair_lo;
  if (pairs_on && count > ctx->pair_lo && count <= ctx->pair_max) {
    CHK(launch_pair(A));
    ctx->bootstraps += count;
    return TFHE_HIP_OK;
  }
  // 2N < count <= 3N: the first 2N as pairs, the rest one per workgroup (3.6 + 2.2 ms; three rounds of singles: 6.4)
  // (not at l = 1, where the batch kernel's first step is cheaper than a pair launch plus a single one: SECURITY_UINT4
  // 4.3 vs 4.9 ms; l = 2: 5.1 vs 5.4, l = 3: 5.8 vs 6.9)
  if (pairs_on && ctx->P.l >= 2 && ctx->pair_max == 2 * ctx->pair_lo && count > ctx->pair_max &&
      count <= ctx->pair_max + ctx->pair_lo) {
    CHK(launch_pair(part(0, ctx->pair_max)));
    CHK(launch_wide(part(ctx->pair_max, count - ctx->pair_max)));
    ctx->bootstraps += count;
    return TFHE_HIP_OK;
  }
  if (ctx->br_wide && count <= ctx->wide_max) {
    CHK(launch_wide(A));
    ctx->bootstraps += count;
    return TFHE_HIP_OK;
  }
  // The batch kernel's time is a staircase with a step every 4N (one more four-wave workgroup
  // per CU: 6.9 / 11.5 / 17.1 / 21.8 ms at 1,024 / 2,048 / 3,072 / 4,096).  A tail of up to 2N ciphertexts above a
  // step is cheaper as a latency-kernel launch of its own (2.2 ms up to N, 3.6 up to 2N) than as a whole further step
  // (1,100: 9.1 vs 10.8 ms, 2,200: 14.4 vs 17.9, 3,300: 19.7 vs 22.9 -- r3x_pair_kernel.log); done up to 32N, beyond
  // which the step is a few per cent of the launch.
  size_t tail = 0;
  if (pairs_on && ctx->br_chunk == 0 && count > ctx->wide_max && count <= 32 * ctx->pair_lo) {
    const size_t r = count % (4 * ctx->pair_lo);
    // (at l = 1 a pair launch costs as much as the step it would save: tails of up to N only)
    if (r > 0 && r <= (ctx->P.l == 1 ? ctx->pair_lo : ctx->pair_max) && count > r) tail = r;
  }
  if (tail) {
    const size_t head = count - tail;
    const BlindRotateArgs H = part(0, head);
    CHK(record_begin(ctx, s, ctx->ev_br));
    hipLaunchKernelGGL(br_kernel(ctx), dim3((unsigned)((head + kBrWaves - 1) / kBrWaves)), dim3(64 * kBrWaves), br_lds_bytes(ctx), s, H);
    HIPCHK(ctx, hipGetLastError());
    CHK(record_end(ctx, s, ctx->ev_br));
    if (tail <= ctx->pair_lo) CHK(launch_wide(part(head, tail)));
    else CHK(launch_pair(part(head, tail)));
    ctx->bootstraps += count;
    return TFHE_HIP_OK;
  }
  dim3 block(64 * kBrWaves);
  size_t lds = br_lds_bytes(ctx);
  // Default: ONE launch of the whole batch.  The four waves of a workgroup meet at a barrier every CMUX step
  // (blind_rotate.hpp), which keeps the resident workgroups streaming the key in near lock-step on its own:
  // L1 86 % / L2 97 % hits, 36 GB of HBM-side traffic per 65,536 bootstraps.  TFHE_HIP_BR_CHUNK splits the
  // batch into launches of N ciphertexts (-1: the resident set) -- the round-1 remedy for free-running
  // one-wave workgroups, whose rounds drifted apart until the 4 MiB L2s thrashed; kept for experiments.
  size_t chunk = count;
  if (ctx->br_chunk > 0) chunk = (size_t)ctx->br_chunk;
  if (ctx->br_chunk < 0) {
    int per_cu = 0;
    hipError_t e = hipErrorUnknown;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, br_kernel(ctx), 64 * kBrWaves, lds);
    if (e == hipSuccess && per_cu > 0 && ctx->num_cus > 0) chunk = (size_t)per_cu * ctx->num_cus * kBrWaves;
  }
  if (chunk == 0 || chunk > count) chunk = count;
  for (size_t done = 0; done < count; done += chunk) {
    const size_t m = (count - done < chunk) ? count - done : chunk;
    const BlindRotateArgs S = part(done, m);
    dim3 grid((unsigned)((m + kBrWaves - 1) / kBrWaves));
    CHK(record_begin(ctx, s, ctx->ev_br));
    hipLaunchKernelGGL(br_kernel(ctx), grid, block, lds, s, S);
    HIPCHK(ctx, hipGetLastError());
    CHK(record_end(ctx, s, ctx->ev_br));
  }
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
bool ks_mfma_possible(const tfhe_hip_ctx *ctx) {
  return ctx->P.basebit == 2 && ctx->P.t >= 6 && ctx->P.t <= 13 && ks_mfma_nt(ctx->P.n) != 0;
}
bool ks_mfma_wanted(const tfhe_hip_ctx *ctx, size_t count) {
  if (!ctx->K->d_ksk8 || !ctx->ks_mfma) return false;
  return ctx->ks_mfma > 1 || count >= ctx->ks_mfma_min;
}
// rows of a level-1 buffer the matrix-core kernel may read: whole 256-row workgroups
size_t lv1_rows(size_t count) { return (count + kKmRows - 1) / kKmRows * kKmRows; }

// (re)build the byte planes from the u32 engine key; called wherever a key becomes current
int build_ksk_planes(tfhe_hip_ctx *ctx) {
  if (!ks_mfma_possible(ctx) || !ctx->ks_mfma) return TFHE_HIP_OK;
  const tfhe_hip_params &P = ctx->P;
  const int nt = ks_mfma_nt(P.n);
  const size_t bytes = ks_mfma_key_bytes(P.n, P.t);
  if (!ctx->K->d_ksk8) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_ksk8, bytes + kKmKeyTailPad));
  HIPCHK(ctx, hipFuncSetAttribute((const void *)km_kernel(nt), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)ks_mfma_lds_bytes(nt)));
  const size_t chunks = bytes / 16;
  hipLaunchKernelGGL(k_ksk_planes, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, ctx->stream, ctx->K->d_ksk,
                     ctx->K->d_ksk8, P.n, P.t, chunks);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_HIP_OK;
}

int launch_key_switch_mfma(tfhe_hip_ctx *ctx, hipStream_t s, const uint32_t *lv1, uint32_t *out, size_t count) {
  const int n = ctx->P.n, nt = ks_mfma_nt(n);
  km_kernel_t kern = km_kernel(nt);
  // the planes are merged with integer atomics: host (pinned, zero-copy) outputs go through a device buffer
  uint32_t *dst = out;
  hipPointerAttribute_t at;
  const size_t obytes = count * (size_t)(n + 1) * 4;
  bool host_out = false;
  if (hipPointerGetAttributes(&at, (const void *)out) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ctx, TFHE_HIP_EINVAL, "key switch output is not GPU-addressable memory");
  }
  if (at.type == hipMemoryTypeHost) {
    host_out = true;
    CHK(ensure(ctx, ctx->ks_out, obytes));
    dst = (uint32_t *)ctx->ks_out.p;
  }
  HIPCHK(ctx, hipMemsetAsync(dst, 0, obytes, s));
  // one workgroup per (128 rows, column block, byte plane, K chunk); consecutive workgroups walk the same key plane.
  // Small batches have few row blocks: the walk over K is cut into up to 16 chunks so that there are about two
  // workgroups per CU to run (a 256-ciphertext batch is 2 row blocks x 8 streams = 16 workgroups otherwise).
  const size_t rb = (count + kKmRows - 1) / kKmRows;
  const size_t lds = ks_mfma_lds_bytes(nt);
  const int tiles = ks_mfma_total_tiles(n);
  const unsigned ncb = (unsigned)(tiles < kKmColBlocks ? tiles : kKmColBlocks);
  int ksplit = 1;
  while (ksplit < 16 && rb * ncb * 4 * (size_t)ksplit < 2 * (size_t)ctx->num_cus) ksplit *= 2;
  if (ctx->ks_mfma_ksplit) ksplit = ctx->ks_mfma_ksplit;
  hipLaunchKernelGGL(kern, dim3((unsigned)rb * (unsigned)ksplit, ncb, 4), dim3(64 * kKmWaves), lds, s, lv1,
                     (const unsigned char *)ctx->K->d_ksk8, n, ctx->P.t, dst, count,
                     ctx->profiling ? ctx->d_diag + 4 : nullptr, ksplit);
  HIPCHK(ctx, hipGetLastError());
  if (host_out) HIPCHK(ctx, hipMemcpyAsync(out, dst, obytes, hipMemcpyDefault, s));
  return TFHE_HIP_OK;
}

int launch_key_switch(tfhe_hip_ctx *ctx, hipStream_t s, const uint32_t *lv1, uint32_t *out, size_t count) {
  if (count == 0) return TFHE_HIP_OK;
  const int n = ctx->P.n;
  const int rw4 = ksk_row_words(n) >> 2;
  const int bd = (rw4 + 63) & ~63;  // <= 320 for n <= 1279
  dim3 grid((unsigned)((count + kKsG - 1) / kKsG)), block(bd);
  CHK(record_begin(ctx, s, ctx->ev_ks));
  // base-4 sets from 64 ciphertexts up: the matrix-core kernel, its walk over K cut into chunks while the batch is
  // too small to fill the chip with row blocks (0.10 / 0.11 / 0.15 / 0.18 / 0.25 / 0.40 ms at 64 / 256 / 512 / 1,024 /
  // 2,048 / 4,096 ciphertexts; the split kernel takes 0.12 / 0.33 / 0.55 / 0.97 / 1.8 / 3.6, the matrix-core kernel
  // without chunks 0.42-0.49 throughout: profiles/exp/logs/r3_ks_splitk.log)
  if (ks_mfma_wanted(ctx, count)) {
    CHK(launch_key_switch_mfma(ctx, s, lv1, out, count));
    CHK(record_end(ctx, s, ctx->ev_ks));
    return TFHE_HIP_OK;
  }
  const size_t sl_lds = ks_sliced_lds_bytes(1 << ctx->P.basebit);  // at the default S; the launch re-derives it for the S it picks
  const bool sliced_ok = (ctx->P.basebit != 2 || ctx->ks_sliced > 1) && ctx->ks_sliced && sl_lds <= 64 * 1024;
  // wider bases from 384 ciphertexts up: the column-sliced kernel with its walk over the coefficients cut into chunks
  // (below; SECURITY_UINT4: 0.37 / 0.37 / 0.57 / 1.05 ms at 512 / 1,024 / 2,048 / 4,096 ciphertexts where the split
  // kernel takes 0.46 / 0.85 / 1.62 / 3.42 and wins below: 0.27 vs 0.29 at 256 -- profiles/exp/logs/r3_ks_sl_chunks.log).
  // Smaller batches, and whatever neither LDS kernel covers:
  if (ctx->br_wide && count <= ctx->ks_split_max && !(sliced_ok && count >= ctx->ks_sl_chunk_min)) {
    // small batch: split each ciphertext's walk over 32 workgroups, merge with integer atomics
    const size_t kb = (size_t)kN * ctx->P.t * (1u << ctx->P.basebit) * ksk_row_words(n) * 4;
    HIPCHK(ctx, hipMemsetAsync(out, 0, count * (size_t)(n + 1) * 4, s));
    hipLaunchKernelGGL(k_key_switch_split, dim3((unsigned)count, 32), block, 0, s, lv1, (const uint4 *)ctx->K->d_ksk,
                       (uint32_t)kb, n, ctx->P.basebit, ctx->P.t, out);
    HIPCHK(ctx, hipGetLastError());
    CHK(record_end(ctx, s, ctx->ev_ks));
    return TFHE_HIP_OK;
  }
  const size_t ksk_bytes = (size_t)kN * ctx->P.t * (1u << ctx->P.basebit) * ksk_row_words(n) * 4;
  const size_t b4_lds = ks_b4_lds_bytes(bd >> 6, kKsG);
  const bool b4_fits = b4_lds <= 64 * 1024;
  if (sliced_ok) {
    // accumulator sets per lane: whichever fills whole rounds of the machine (two workgroups per CU)
    const int slices = (n + 1 + 63) / 64, base = 1 << ctx->P.basebit;
    int sets = ks_sliced_pick_sets(count, slices, 2 * ctx->num_cus);
    if (ctx->ks_sliced_sets) sets = ctx->ks_sliced_sets;
    if (ks_sliced_lds_bytes(base, sets) > 64 * 1024) sets = kKsSlSets;
    const size_t lds = ks_sliced_lds_bytes(base, sets);
    // small batches have few ciphertext groups: the walk over the N coefficients is cut into up to 64 chunks (grid.z)
    // so that about two workgroups per CU exist; the chunks meet in the zeroed output through integer atomics
    const size_t groups = (count + (size_t)ks_sliced_cts(sets) - 1) / (size_t)ks_sliced_cts(sets);
    int kchunks = 1;
    if (ctx->br_wide)
      while (kchunks < 64 && groups * (size_t)slices * (size_t)kchunks < 2 * (size_t)ctx->num_cus) kchunks *= 2;
    if (ctx->ks_sl_kchunks) kchunks = ctx->ks_sl_kchunks;
    uint32_t *dst = out;
    bool host_out = false;
    const size_t obytes = count * (size_t)(n + 1) * 4;
    if (kchunks > 1) {  // atomics: host (pinned, zero-copy) outputs go through a device buffer, as for the matrix-core kernel
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
      HIPCHK(ctx, hipMemsetAsync(dst, 0, obytes, s));
    }
    dim3 sgrid((unsigned)groups, (unsigned)slices, (unsigned)kchunks);
    typedef void (*sl_kernel_t)(const uint32_t *, const unsigned char *, int, int, int, uint32_t *, size_t);
    sl_kernel_t kern = nullptr;
    const bool ic8 = ks_sliced_stage(base) == 8;
    switch (sets) {
      case 24: kern = ic8 ? k_key_switch_sliced<8, 24> : k_key_switch_sliced<16, 24>; break;
      case 28: kern = ic8 ? k_key_switch_sliced<8, 28> : k_key_switch_sliced<16, 28>; break;
      case 36: kern = ic8 ? k_key_switch_sliced<8, 36> : k_key_switch_sliced<16, 36>; break;
      case 40: kern = ic8 ? k_key_switch_sliced<8, 40> : k_key_switch_sliced<16, 40>; break;
      default: kern = ic8 ? k_key_switch_sliced<8, 32> : k_key_switch_sliced<16, 32>; break;
    }
    hipLaunchKernelGGL(kern, sgrid, dim3(256), lds, s, lv1, (const unsigned char *)ctx->K->d_ksk, n, ctx->P.basebit, ctx->P.t,
                       dst, count);
    if (host_out) {
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipMemcpyAsync(out, dst, obytes, hipMemcpyDefault, s));
    }
  } else if (ctx->P.basebit == 2 && ctx->ks_b4 && b4_fits)
    hipLaunchKernelGGL((k_key_switch_b4<kKsG>), grid, block, b4_lds, s, lv1, (const unsigned char *)ctx->K->d_ksk, n,
                       ctx->P.t, out, count);
  else
    hipLaunchKernelGGL((k_key_switch<kKsG>), grid, block, 0, s, lv1, (const uint4 *)ctx->K->d_ksk, (uint32_t)ksk_bytes,
                       n, ctx->P.basebit, ctx->P.t, out, count);
  HIPCHK(ctx, hipGetLastError());
  CHK(record_end(ctx, s, ctx->ev_ks));
  return TFHE_HIP_OK;
}

int need_key(tfhe_hip_ctx *ctx) {
  if (!ctx->K->key_loaded) return fail(ctx, TFHE_HIP_ENOKEY, "cloud key not loaded");
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
  if (b.p) HIPCHK(ctx, hipHostFree(b.p));
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

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

const char *tfhe_hip_name(void) { return TFHE_ABLATED ? "hip-gfx950-EXPERIMENT" : "hip-gfx950"; }

const char *tfhe_hip_last_error(const tfhe_hip_ctx *ctx) {
  if (ctx && ctx->parent) ctx = ctx->parent;
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
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
  // pre-rounding magnitude bound: 2l polynomials x N terms x (Bg/2) digit x 2^31 key coefficient
  ctx->fast_round = std::log2(2.0 * p->l) + 10.0 + (p->bgbit - 1) + 31.0 < 51.0;
  if (const char *env = getenv("TFHE_HIP_FAST_ROUND")) ctx->fast_round = ctx->fast_round && atoi(env) != 0;
  if (const char *env = getenv("TFHE_HIP_KS_B4")) ctx->ks_b4 = atoi(env) != 0;
  if (const char *env = getenv("TFHE_HIP_KS_SLICED")) ctx->ks_sliced = atoi(env);
  if (const char *env = getenv("TFHE_HIP_KS_SLICED_SETS")) {
    const int v = atoi(env);
    ctx->ks_sliced_sets = (v == 24 || v == 28 || v == 32 || v == 36 || v == 40) ? v : 0;
  }
  if (const char *env = getenv("TFHE_HIP_KS_MFMA")) ctx->ks_mfma = atoi(env);
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
  // measured crossovers vs the group kernels: ~7.8k ciphertexts (base 4, LDS ring), ~4.1k (column-sliced)
  ctx->ks_split_max = (p->basebit == 2 ? 28 : 16) * (size_t)ctx->num_cus;
  if (const char *env = getenv("TFHE_HIP_BR_WIDE")) ctx->br_wide = atoi(env) != 0;
  if (const char *env = getenv("TFHE_HIP_BR_WIDE2")) ctx->br_wide2 = atoi(env) != 0;
  // crossover vs the batch kernel (7.0 ms for anything up to 1,024 ciphertexts at 128 bit): the eight-wave form takes
  // 2.2 / 4.5 / 6.6 / 8.5 ms for 1 / 2 / 3 / 4 rounds of one workgroup per CU, the six-wave form 3.0 / 6.1 / 9.1
  // (profiles/exp/logs/r3o_crossover.log, r2q_crossover_latency_vs_batch.log)
  // (three rounds only at l = 3: at l = 1, 2 the batch kernel's first step -- 4.3 / 5.4 ms -- is cheaper than three
  // rounds of singles -- 6.2 ms)
  ctx->wide_max = ((ctx->br_wide2 && p->l >= 3) ? 3 : 2) * (size_t)ctx->num_cus;
  if (const char *env = getenv("TFHE_HIP_WIDE_MAX")) ctx->wide_max = (size_t)atol(env);
  ctx->pair_lo = (size_t)ctx->num_cus;
  ctx->pair_max = 2 * (size_t)ctx->num_cus;
  if (const char *env = getenv("TFHE_HIP_PAIR_LO")) ctx->pair_lo = (size_t)atol(env);
  if (const char *env = getenv("TFHE_HIP_PAIR_MAX")) ctx->pair_max = (size_t)atol(env);
  if (const char *env = getenv("TFHE_HIP_KS_SPLIT_MAX")) ctx->ks_split_max = (size_t)atol(env);
  if (const char *env = getenv("TFHE_HIP_BR_CHUNK")) ctx->br_chunk = atol(env);
  // dynamic LDS above the 64 KiB default, declared once per context for the kernels of this parameter set
  {
    const size_t lds = blind_rotate_lds_bytes(p->n);
    const size_t wlds = ctx->br_wide2 ? blind_rotate_wide2_lds_bytes(p->n, p->l) : blind_rotate_wide_lds_bytes(p->n, p->l);
    if ((e = hipFuncSetAttribute((const void *)br_kernel(ctx), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
      return bail("hipFuncSetAttribute(k_blind_rotate)", e);
    if ((e = hipFuncSetAttribute((const void *)br_wide_kernel(ctx), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds)) != hipSuccess)
      return bail("hipFuncSetAttribute(k_blind_rotate_wide)", e);
    if (blind_rotate_pair_lds_bytes(p->n) > 160 * 1024) ctx->pair_max = 0;  // (n > 1,900: no parameter set)
    else if ((e = hipFuncSetAttribute((const void *)br_pair_kernel(ctx), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)blind_rotate_pair_lds_bytes(p->n))) != hipSuccess)
      return bail("hipFuncSetAttribute(k_blind_rotate_pair)", e);
  }
  std::vector<double2> tw;
  make_twiddles(tw);
  if ((e = hipMalloc((void **)&ctx->d_tw, tw.size() * sizeof(double2))) != hipSuccess)
    return bail("hipMalloc twiddles", e);
  if ((e = hipMemcpy(ctx->d_tw, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice)) != hipSuccess)
    return bail("hipMemcpy twiddles", e);
  // the persistent kernel may want > 64 KiB of dynamic LDS only for absurd n; the
  // default limit (64 KiB) is ample: 9216 + 2n bytes.
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
      std::lock_guard<std::mutex> lk(base->mu);
      DeviceGuard dg(base->device);
      if (base->scratch_owned && base->scratch_owner != base->stream) (void)hipStreamSynchronize(base->scratch_owner);
      if (base->stream) (void)hipStreamSynchronize(base->stream);
      free_key(ctx->own);
      last_of_dying = --base->views == 0 && base->dying;
      delete ctx;
    }
    if (last_of_dying) tfhe_hip_ctx_destroy(base);  // the parent was destroyed first: it has waited for its views
    return;
  }
  {
    // Destroyed before its views (the header asks for the opposite order): the views still run on this context's
    // stream, scratch and mutex, so keep it alive until the last of them goes.
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->views > 0) {
      ctx->dying = true;
      return;
    }
  }
  DeviceGuard dg(ctx->device);
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
  DevBuf *bufs[] = {&ctx->lv1, &ctx->u1, &ctx->u2, &ctx->h_a, &ctx->h_b, &ctx->h_c, &ctx->h_out, &ctx->h_tv, &ctx->h_idx, &ctx->ks_out};
  for (DevBuf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  free_key(ctx->own);
  for (PinBuf *b : {&ctx->p_a, &ctx->p_b, &ctx->p_c, &ctx->p_out})
    if (b->p) (void)hipHostFree(b->p);
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
  std::lock_guard<std::mutex> lk(base->mu);
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
  tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;
  std::lock_guard<std::mutex> lk(base->mu);
  return ctx->own.key_loaded ? 1 : 0;
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
  const tfhe_hip_params &P = ctx->P;
  const size_t polys = (size_t)P.n * 2 * P.l * 2;
  const size_t bsk_bytes = polys * kN * sizeof(double);
  const int base = 1 << P.basebit;
  const size_t ksk_words = (size_t)kN * P.t * base * (size_t)(P.n + 1);
  ctx->K->key_loaded = false;
  if (!ctx->K->d_bsk) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_bsk, bsk_bytes));
  if (!ctx->K->d_ksk)
    HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_ksk, (size_t)kN * P.t * base * ksk_row_words(P.n) * 4 + 4096));
  if (!ctx->K->d_testvec) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_testvec, 2 * kN * 4));
  // bootstrapping key: upload the reference layout, permute + scale on the device
  double *d_ref = nullptr;
  HIPCHK(ctx, hipMalloc((void **)&d_ref, bsk_bytes));
  hipError_t e = hipMemcpyAsync(d_ref, bsk, bsk_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_bsk_convert, dim3((unsigned)polys), dim3(512), 0, ctx->stream, d_ref, ctx->K->d_bsk, polys);
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
  const tfhe_hip_params &P = ctx->P;
  const int base = 1 << P.basebit;
  const size_t polys = (size_t)P.n * 2 * P.l * 2;
  ctx->K->key_loaded = false;
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
    case 1: hipLaunchKernelGGL(k_gen_bsk<1>, bgrid, dim3(64), kStageLdsBytes, ctx->stream, d_k0, d_spec, ctx->d_tw, ctx->K->d_bsk, P.bgbit, alpha_bsk, d_rk); break;
    case 2: hipLaunchKernelGGL(k_gen_bsk<2>, bgrid, dim3(64), kStageLdsBytes, ctx->stream, d_k0, d_spec, ctx->d_tw, ctx->K->d_bsk, P.bgbit, alpha_bsk, d_rk); break;
    default: hipLaunchKernelGGL(k_gen_bsk<3>, bgrid, dim3(64), kStageLdsBytes, ctx->stream, d_k0, d_spec, ctx->d_tw, ctx->K->d_bsk, P.bgbit, alpha_bsk, d_rk); break;
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
      std::lock_guard<std::mutex> lk(base->mu);
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
    hipLaunchKernelGGL(k_bsk_export, dim3((unsigned)polys), dim3(512), 0, ctx->stream, ctx->K->d_bsk, (double *)ctx->h_out.p, polys);
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
  HIPCHK(ctx, hipDeviceSynchronize());
  CHK(build_ksk_planes(ctx));
  ctx->K->offset = decomp_offset;
  ctx->K->key_loaded = true;
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

int tfhe_hip_batch_gate(tfhe_hip_ctx *ctx, int gate, const uint32_t *a, const uint32_t *b,
                        uint32_t *out, size_t count) {
  if (!ctx) return TFHE_HIP_EINVAL;
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
  return TFHE_HIP_OK;
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
