// pool.hpp -- several GPUs behind ONE handle of the C ABI (included by tfhe_hip.hip; needs tfhe_hip_ctx).
//
// The reference's batch entry points are `par_iter().map().collect()` over the ciphertexts of a slice
// (src/parallel/rayon_impl.rs:40-47, called from src/gates.rs:357-383): order-preserving, embarrassingly
// parallel, one shared read-only &CloudKey.  The pool is that map over devices: one context per device, the
// cloud key generated / uploaded ONCE and replicated device-to-device in the engine layouts (no second
// conversion, no host round trip), a contiguous order-preserving split of the batch, one host thread per
// shard (each thread makes its shard's device current; the HIP current device is per thread), results
// written in place into the caller's output slice.  No collective and no exchange on the data path.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is opened at run time (rccl_api)

#include <system_error>
#include <thread>

struct tfhe_hip_pool {
  std::vector<tfhe_hip_ctx *> ctxs;
  std::mutex mu;  // one batch at a time per pool (the contexts' host staging buffers are per context)
  std::string err = "";
  bool replicated_by_rccl = false;  // how the last key reached the members (tfhe_hip_pool_key_transport)
};

namespace {

// [lo, hi) of shard r of `world` over `count` items: contiguous, order-preserving, sizes differ by at most 1
inline void pool_shard(size_t count, int r, int world, size_t &lo, size_t &hi) {
  const size_t base = count / (size_t)world, rem = count % (size_t)world;
  lo = (size_t)r * base + ((size_t)r < rem ? (size_t)r : rem);
  hi = lo + base + ((size_t)r < rem ? 1 : 0);
}

int pool_fail(tfhe_hip_pool *p, int code, const std::string &msg) {
  p->err = msg;
  return code;
}

// dst takes src's key (engine layouts), device to device.  Either may be a key view; both idle on entry.
int clone_key(tfhe_hip_ctx *dst, tfhe_hip_ctx *src) {
  const KeyState *from = &src->own;  // the key src OWNS (K is only bound during a call)
  const int src_device = src->device;
  tfhe_hip_ctx *ctx = dst;
  ENTER(ctx);
  if (ctx->scratch_owned) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scratch_owned = false;
  const tfhe_hip_params &P = ctx->P;
  const size_t bsk_bytes = (size_t)P.n * 2 * P.l * 2 * kN * sizeof(double);
  const size_t ksk_bytes = (size_t)kN * P.t * (1u << P.basebit) * ksk_row_words(P.n) * 4;
  ctx->K->key_loaded = false;
  if (!ctx->K->d_bsk) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_bsk, bsk_bytes));
  if (!ctx->K->d_ksk) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_ksk, ksk_bytes + 4096));
  if (!ctx->K->d_testvec) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_testvec, 2 * kN * 4));
  HIPCHK(ctx, hipMemcpyPeer(ctx->K->d_bsk, ctx->device, from->d_bsk, src_device, bsk_bytes));
  HIPCHK(ctx, hipMemcpyPeer(ctx->K->d_ksk, ctx->device, from->d_ksk, src_device, ksk_bytes));
  HIPCHK(ctx, hipMemcpyPeer(ctx->K->d_testvec, ctx->device, from->d_testvec, src_device, 2 * kN * 4));
  HIPCHK(ctx, hipDeviceSynchronize());
  CHK(build_ksk_planes(ctx));
  ctx->K->offset = from->offset;
  ctx->K->key_loaded = true;
  return TFHE_HIP_OK;
}

// ---- key replication by RCCL broadcast (xGMI), when the pool's devices are distinct -------------------------
// north_star: "RCCL over xGMI used only for the trivial scatter/gather": the one exchange this path has is the
// replication of the cloud key (172 MB, once per key).  librccl is opened at run time (no link-time dependency: a
// process that already carries torch's RCCL reuses it by SONAME); one communicator per pool, one grouped
// ncclBroadcast per key buffer, in place in the engine layouts.  Anything that fails -- library absent, duplicate
// devices (ncclCommInitAll refuses them), a transport error -- falls back to the serial hipMemcpyPeer path below.
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};
inline RcclApi &rccl_api() {
  static RcclApi a = [] {
    RcclApi r;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return r;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
    r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
    r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Broadcast;
    return r;
  }();
  return a;
}

// member i >= 1: drained, its key buffers allocated, no valid key until finish_replica
int prepare_replica(tfhe_hip_ctx *member) {
  tfhe_hip_ctx *ctx = member;
  ENTER(ctx);
  if (ctx->scratch_owned) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scratch_owned = false;
  const tfhe_hip_params &P = ctx->P;
  ctx->K->key_loaded = false;
  if (!ctx->K->d_bsk) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_bsk, (size_t)P.n * 2 * P.l * 2 * kN * sizeof(double)));
  if (!ctx->K->d_ksk)
    HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_ksk, (size_t)kN * P.t * (1u << P.basebit) * ksk_row_words(P.n) * 4 + 4096));
  if (!ctx->K->d_testvec) HIPCHK(ctx, hipMalloc((void **)&ctx->K->d_testvec, 2 * kN * 4));
  return TFHE_HIP_OK;
}
int finish_replica(tfhe_hip_ctx *member, uint32_t offset) {
  tfhe_hip_ctx *ctx = member;
  ENTER(ctx);
  HIPCHK(ctx, hipDeviceSynchronize());
  CHK(build_ksk_planes(ctx));
  ctx->K->offset = offset;
  ctx->K->key_loaded = true;
  return TFHE_HIP_OK;
}

// true: every member holds member 0's key.  false: nothing usable happened (members >= 1 may hold garbage and are
// marked unloaded): the caller takes the peer-copy path.
bool replicate_key_rccl(tfhe_hip_pool *p) {
  const int n = (int)p->ctxs.size();
  const char *env = getenv("TFHE_HIP_POOL_RCCL");
  const int mode = env ? atoi(env) : 1;  // 0 = never, 1 = pools of >= 2 members, 2 = also a pool of one (plumbing test)
  if (mode == 0 || n < (mode >= 2 ? 1 : 2)) return false;
  RcclApi &R = rccl_api();
  if (!R.ok) return false;
  std::vector<int> devs;
  for (auto *c : p->ctxs) devs.push_back(c->device);
  for (int i = 0; i < n; ++i)  // one rank per GPU: a pool that repeats a device takes the peer-copy path
    for (int j = i + 1; j < n; ++j)
      if (devs[(size_t)i] == devs[(size_t)j]) return false;
  for (int i = 1; i < n; ++i)
    if (prepare_replica(p->ctxs[(size_t)i]) != TFHE_HIP_OK) return false;
  int prev = -1;
  (void)hipGetDevice(&prev);
  std::vector<ncclComm_t> comms((size_t)n, nullptr);
  if (R.CommInitAll(comms.data(), n, devs.data()) != ncclSuccess) {
    (void)hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    return false;
  }
  const tfhe_hip_params &P = p->ctxs[0]->P;
  const size_t bytes[3] = {(size_t)P.n * 2 * P.l * 2 * kN * sizeof(double),
                           (size_t)kN * P.t * (1u << P.basebit) * ksk_row_words(P.n) * 4, (size_t)2 * kN * 4};
  bool good = true;
  for (int b = 0; b < 3 && good; ++b) {
    good = R.GroupStart() == ncclSuccess;
    for (int i = 0; i < n && good; ++i) {
      tfhe_hip_ctx *c = p->ctxs[(size_t)i];
      tfhe_hip_ctx *base = c->parent ? c->parent : c;
      KeyState &k = c->own;
      void *buf = b == 0 ? (void *)k.d_bsk : b == 1 ? (void *)k.d_ksk : (void *)k.d_testvec;
      good = hipSetDevice(c->device) == hipSuccess &&
             R.Broadcast(buf, buf, bytes[b], ncclUint8, 0, comms[(size_t)i], base->stream) == ncclSuccess;
    }
    good = (R.GroupEnd() == ncclSuccess) && good;
  }
  for (int i = 0; i < n; ++i) {
    tfhe_hip_ctx *c = p->ctxs[(size_t)i];
    tfhe_hip_ctx *base = c->parent ? c->parent : c;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(base->stream) != hipSuccess) good = false;
    (void)R.CommDestroy(comms[(size_t)i]);
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  if (!good) {
    (void)hipGetLastError();
    return false;
  }
  for (int i = 1; i < n; ++i)
    if (finish_replica(p->ctxs[(size_t)i], p->ctxs[0]->own.offset) != TFHE_HIP_OK) return false;
  p->replicated_by_rccl = true;
  return true;
}

int replicate_key(tfhe_hip_pool *p) {
  p->replicated_by_rccl = false;
  if (replicate_key_rccl(p)) return TFHE_HIP_OK;
  for (size_t i = 1; i < p->ctxs.size(); ++i) {
    const int rc = clone_key(p->ctxs[i], p->ctxs[0]);
    if (rc != TFHE_HIP_OK) return pool_fail(p, rc, "device " + std::to_string(p->ctxs[i]->device) + ": " + tfhe_hip_last_error(p->ctxs[i]));
  }
  return TFHE_HIP_OK;
}

// run(ctx, lo, hi) on every non-empty shard, shard 0 on the calling thread; first failure wins
// Members a batch of `count` is spread over: one device runs up to 256 ciphertexts in the time of one (the
// latency kernels give every ciphertext its own workgroup), so smaller batches are not cut thinner than that.
inline int pool_world_for(const tfhe_hip_pool *p, size_t count) {
  const size_t want = (count + 255) / 256;
  const size_t have = p->ctxs.size();
  return (int)(want < 1 ? 1 : (want < have ? want : have));
}

template <class F>
int pool_map(tfhe_hip_pool *p, size_t count, F &&run) {
  const int world = pool_world_for(p, count);
  std::vector<int> rc((size_t)world, TFHE_HIP_OK);
  std::vector<std::thread> th;
  for (int r = 1; r < world; ++r) {
    size_t lo, hi;
    pool_shard(count, r, world, lo, hi);
    if (hi <= lo) continue;
    try {
      th.emplace_back([&, r, lo, hi] { rc[(size_t)r] = run(p->ctxs[(size_t)r], lo, hi); });
    } catch (const std::system_error &) {  // no thread to be had: this shard runs on the calling thread
      rc[(size_t)r] = run(p->ctxs[(size_t)r], lo, hi);
    }
  }
  {
    size_t lo, hi;
    pool_shard(count, 0, world, lo, hi);
    if (hi > lo) rc[0] = run(p->ctxs[0], lo, hi);
  }
  for (auto &t : th) t.join();
  for (int r = 0; r < world; ++r)
    if (rc[(size_t)r] != TFHE_HIP_OK)
      return pool_fail(p, rc[(size_t)r], "device " + std::to_string(p->ctxs[(size_t)r]->device) + ": " + tfhe_hip_last_error(p->ctxs[(size_t)r]));
  return TFHE_HIP_OK;
}

}  // namespace

int tfhe_hip_pool_create(const tfhe_hip_params *params, const int *devices, int ndev, tfhe_hip_pool **out) {
  if (!out) return TFHE_HIP_EINVAL;
  *out = nullptr;
  if (!params || !devices || ndev <= 0 || ndev > 64) {
    g_create_error = "pool: need 1..64 devices";
    return TFHE_HIP_EINVAL;
  }
  tfhe_hip_pool *p = new tfhe_hip_pool();
  for (int i = 0; i < ndev; ++i) {
    tfhe_hip_ctx *c = nullptr;
    const int rc = tfhe_hip_ctx_create(params, devices[i], &c);
    if (rc != TFHE_HIP_OK) {  // g_create_error holds the text
      for (auto *x : p->ctxs) tfhe_hip_ctx_destroy(x);
      delete p;
      return rc;
    }
    p->ctxs.push_back(c);
  }
  // several members fed by one host: each stages its pageable shard through a pinned arena of its own (to_dev)
  const char *env = getenv("TFHE_HIP_POOL_PINNED_STAGING");
  const bool stage = env ? atoi(env) != 0 : ndev > 1;
  for (auto *c : p->ctxs) c->stage_pinned = stage;
  *out = p;
  return TFHE_HIP_OK;
}

void tfhe_hip_pool_destroy(tfhe_hip_pool *p) {
  if (!p) return;
  for (auto *c : p->ctxs) tfhe_hip_ctx_destroy(c);
  delete p;
}

// A key view of a pool: one key view per member context (same devices, same streams and scratch, its own cloud key).
int tfhe_hip_pool_key_create(tfhe_hip_pool *pool, tfhe_hip_pool **out) {
  if (!out) return TFHE_HIP_EINVAL;
  *out = nullptr;
  if (!pool) return TFHE_HIP_EINVAL;
  tfhe_hip_pool *v = new tfhe_hip_pool();
  for (auto *c : pool->ctxs) {
    tfhe_hip_ctx *kv = nullptr;
    const int rc = tfhe_hip_key_create(c, &kv);
    if (rc != TFHE_HIP_OK) {
      for (auto *x : v->ctxs) tfhe_hip_ctx_destroy(x);
      delete v;
      return rc;
    }
    v->ctxs.push_back(kv);
  }
  *out = v;
  return TFHE_HIP_OK;
}

int tfhe_hip_pool_size(const tfhe_hip_pool *p) { return p ? (int)p->ctxs.size() : 0; }

tfhe_hip_ctx *tfhe_hip_pool_ctx(tfhe_hip_pool *p, int i) {
  return (p && i >= 0 && (size_t)i < p->ctxs.size()) ? p->ctxs[(size_t)i] : nullptr;
}

const char *tfhe_hip_pool_last_error(const tfhe_hip_pool *p) { return p ? p->err.c_str() : g_create_error.c_str(); }

// "rccl" when the pool's last cloud key reached its members by ncclBroadcast, "peer-copy" when by hipMemcpyPeer
// (or when there was nothing to replicate).
const char *tfhe_hip_pool_key_transport(const tfhe_hip_pool *p) { return (p && p->replicated_by_rccl) ? "rccl" : "peer-copy"; }

int tfhe_hip_pool_members_for(const tfhe_hip_pool *p, size_t count) { return p ? pool_world_for(p, count) : 0; }

void tfhe_hip_pool_shard(size_t count, int shard, int nshards, size_t *lo, size_t *hi) {
  size_t a = 0, b = 0;
  if (nshards > 0 && shard >= 0 && shard < nshards) pool_shard(count, shard, nshards, a, b);
  if (lo) *lo = a;
  if (hi) *hi = b;
}

#define POOL_ENTER(p)               \
  if (!(p)) return TFHE_HIP_EINVAL; \
  std::lock_guard<std::mutex> plk_((p)->mu)
#define POOL_FIRST(p, call)                                                                                        \
  do {                                                                                                             \
    const int rc_ = (call);                                                                                        \
    if (rc_ != TFHE_HIP_OK) return pool_fail(p, rc_, "device " + std::to_string((p)->ctxs[0]->device) + ": " + tfhe_hip_last_error((p)->ctxs[0])); \
  } while (0)

int tfhe_hip_pool_load_cloud_key(tfhe_hip_pool *p, const double *bsk, const uint32_t *ksk, uint32_t decomp_offset,
                                 const uint32_t *testvec) {
  POOL_ENTER(p);
  POOL_FIRST(p, tfhe_hip_load_cloud_key(p->ctxs[0], bsk, ksk, decomp_offset, testvec));
  return replicate_key(p);
}

int tfhe_hip_pool_gen_cloud_key_secure(tfhe_hip_pool *p, const uint32_t *key_lv0, const uint32_t *key_lv1,
                                       double alpha_ksk, double alpha_bsk) {
  POOL_ENTER(p);
  POOL_FIRST(p, tfhe_hip_gen_cloud_key_secure(p->ctxs[0], key_lv0, key_lv1, alpha_ksk, alpha_bsk));
  return replicate_key(p);
}

int tfhe_hip_pool_gen_cloud_key_with_key(tfhe_hip_pool *p, const uint32_t *key_lv0, const uint32_t *key_lv1,
                                         double alpha_ksk, double alpha_bsk, const uint8_t rng_key[32]) {
  POOL_ENTER(p);
  POOL_FIRST(p, tfhe_hip_gen_cloud_key_with_key(p->ctxs[0], key_lv0, key_lv1, alpha_ksk, alpha_bsk, rng_key));
  return replicate_key(p);
}

int tfhe_hip_pool_gen_cloud_key(tfhe_hip_pool *p, const uint32_t *key_lv0, const uint32_t *key_lv1, double alpha_ksk,
                                double alpha_bsk, uint64_t seed) {
  POOL_ENTER(p);
  POOL_FIRST(p, tfhe_hip_gen_cloud_key(p->ctxs[0], key_lv0, key_lv1, alpha_ksk, alpha_bsk, seed));
  return replicate_key(p);
}

int tfhe_hip_pool_export_cloud_key(tfhe_hip_pool *p, int member, double *bsk, uint32_t *ksk, uint32_t *decomp_offset,
                                   uint32_t *testvec) {
  POOL_ENTER(p);
  if (member < 0 || (size_t)member >= p->ctxs.size()) return pool_fail(p, TFHE_HIP_EINVAL, "no such pool member");
  tfhe_hip_ctx *c = p->ctxs[(size_t)member];
  const int rc = tfhe_hip_export_cloud_key(c, bsk, ksk, decomp_offset, testvec);
  if (rc != TFHE_HIP_OK) return pool_fail(p, rc, "device " + std::to_string(c->device) + ": " + tfhe_hip_last_error(c));
  return TFHE_HIP_OK;
}

int tfhe_hip_pool_batch_gate(tfhe_hip_pool *p, int gate, const uint32_t *a, const uint32_t *b, uint32_t *out,
                             size_t count) {
  POOL_ENTER(p);
  if (count && (!a || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_gate(c, gate, a + lo * w, b ? b + lo * w : nullptr, out + lo * w, hi - lo);
  });
}

int tfhe_hip_pool_batch_gates_mixed(tfhe_hip_pool *p, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                    uint32_t *out, size_t count) {
  POOL_ENTER(p);
  if (count && (!gates || !a || !b || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_gates_mixed(c, gates + lo, a + lo * w, b + lo * w, out + lo * w, hi - lo);
  });
}

int tfhe_hip_pool_batch_bootstrap(tfhe_hip_pool *p, const uint32_t *in, const uint32_t *testvec, int per_ct,
                                  int keyswitch, uint32_t *out, size_t count) {
  POOL_ENTER(p);
  if (count && (!in || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  const size_t tvs = (testvec && per_ct) ? (size_t)2 * kN : 0;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_bootstrap(c, in + lo * w, testvec ? testvec + lo * tvs : nullptr, per_ct, keyswitch, out + lo * w,
                                    hi - lo);
  });
}

int tfhe_hip_pool_batch_mux(tfhe_hip_pool *p, int naive, const uint32_t *a, const uint32_t *b, const uint32_t *c3,
                            uint32_t *out, size_t count) {
  POOL_ENTER(p);
  if (count && (!a || !b || !c3 || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_mux(c, naive, a + lo * w, b + lo * w, c3 + lo * w, out + lo * w, hi - lo);
  });
}

int tfhe_hip_pool_batch_blind_rotate(tfhe_hip_pool *p, const uint32_t *in, const uint32_t *testvec, uint32_t *out_trlwe,
                                     size_t count) {
  POOL_ENTER(p);
  if (count && (!in || !out_trlwe)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_blind_rotate(c, in + lo * w, testvec, out_trlwe + lo * (size_t)2 * kN, hi - lo);
  });
}
#undef POOL_ENTER
#undef POOL_FIRST
