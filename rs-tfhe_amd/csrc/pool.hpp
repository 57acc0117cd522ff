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

#include <chrono>
#include <system_error>
#include <thread>

struct RcclApi;
struct tfhe_hip_pool {
  std::vector<tfhe_hip_ctx *> ctxs;
  tfhe_hip_pool *parent = nullptr;  // non-null: a key view of `parent` (same devices, streams, staging, communicator and mutex)
  int views = 0;                    // live key views (root only)
  bool dying = false;               // destroyed while views were alive: the last view to go frees the pool
  FairMutex own_mu;  // one call at a time per ROOT pool, first come first served: a view's calls serialise with its parent's (they share streams and staging)
  uint64_t id = new_handle_id();  // key of the per-thread error text (err_slot)
  bool replicated_by_rccl = false;  // how the last key reached the members (tfhe_hip_pool_key_transport)
  // ---- root only: what the device-resident (_dev) calls and the key replication share ----
  std::vector<ncclComm_t> comms;    // ONE persistent communicator per pool (created on first use, destroyed with the pool)
  int comm_state = 0;               // 0 = not tried yet, 1 = ready, -1 = unavailable (duplicate devices, no librccl, init failed)
  bool comms_dropped = false;       // a group failed half-way (pool_drop_comms): streams it touched may never drain
  struct Stage {                    // per member: staging for shards that arrive from / leave for another member's GPU
    DevBuf in[5], out;
    hipEvent_t done = nullptr;      // recorded on the member's stream when its shard's result has left
  };
  std::vector<Stage> stage;
  hipEvent_t ready = nullptr;       // recorded on the home stream when the call's operands are ready (peer-copy path)
  int ready_device = -1;
  const char *last_transport = "none";  // transport of the last _dev call: "rccl" / "peer-copy" / "none" (nothing moved)
  // transfer timing (tfhe_hip_pool_get_transfer_times): per moved shard one event pair on the MEMBER's stream (the
  // receiver of a scatter, the sender of a gather); on the RCCL path also one pair per call and direction on the HOME
  // stream around the group (`home_ix` into ev_home).  A transfer can only start when BOTH ends have reached it, so
  // the pair of the side that arrived last brackets the transfer alone and the other one also brackets its wait for
  // the peer (e.g. the home stream's receive group waits for the slowest member's compute): a shard's transfer time
  // is the SHORTER of its two brackets.
  struct Timed {
    hipEvent_t a = nullptr, b = nullptr;
    int member = 0;    // whose device the events live on
    long home_ix = -1;  // RCCL: the call's home-stream pair
    bool gather = false;  // (ev_home) which direction's group this pair brackets
  };
  bool timing = false;
  std::vector<Timed> ev_scatter, ev_gather, ev_home;
  uint64_t scatter_bytes = 0, gather_bytes = 0, dev_calls = 0;
  // set-up costs, host wall clock (not reset by reading): creating the persistent communicator (ncclCommInitAll, once
  // per pool) and the last replication of a cloud key to the members (ncclBroadcast or peer copies + the members'
  // byte-plane builds), so that neither hides inside a "key generation" figure
  double comm_create_ms = 0.0, key_replication_ms = 0.0;

  ~tfhe_hip_pool() { handle_gone(id); }
  tfhe_hip_pool *root() { return parent ? parent : this; }
  const tfhe_hip_pool *root() const { return parent ? parent : this; }
};

namespace {

// [lo, hi) of shard r of `world` over `count` items: contiguous, order-preserving, sizes differ by at most 1
inline void pool_shard(size_t count, int r, int world, size_t &lo, size_t &hi) {
  const size_t base = count / (size_t)world, rem = count % (size_t)world;
  lo = (size_t)r * base + ((size_t)r < rem ? (size_t)r : rem);
  hi = lo + base + ((size_t)r < rem ? 1 : 0);
}

int pool_fail(tfhe_hip_pool *p, int code, const std::string &msg) {
  err_slot(p->id) = msg;
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
  comb_quiesce(ctx);
  const tfhe_hip_params &P = ctx->P;
  const size_t bsk_bytes = (size_t)P.n * 2 * P.l * 2 * kN * sizeof(double);
  const size_t ksk_bytes = (size_t)kN * P.t * (1u << P.basebit) * ksk_row_words(P.n) * 4;
  ctx->K->key_loaded = ctx->K->reenc_loaded = false;
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
  comb_prepare(ctx);
  return TFHE_HIP_OK;
}

// ---- RCCL over xGMI: one persistent communicator per pool -------------------------------------------------------
// north_star: "RCCL over xGMI used only for the trivial scatter/gather".  The exchanges this path has are the
// replication of the cloud key (172 MB, once per key: one grouped ncclBroadcast per key buffer, in place in the
// engine layouts) and, for a batch that is resident on ONE member's GPU (the *_dev pool calls below), the scatter of
// its shards and the gather of their results (grouped ncclSend / ncclRecv, one pair per peer, so all xGMI links run
// concurrently).  librccl is opened at run time (no link-time dependency: a process that already carries torch's
// RCCL reuses it by SONAME).  The communicator is created on first use and lives as long as the pool.  Anything that
// fails -- library absent, duplicate devices (ncclCommInitAll refuses them), a transport error -- falls back to
// hipMemcpyPeer[Async].
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
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
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
    r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
    r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
    r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Broadcast && r.Send && r.Recv;
    return r;
  }();
  return a;
}

inline int pool_rccl_mode() {  // 0 = never, 1 = pools of >= 2 distinct devices, 2 = also a pool of one (plumbing test)
  const char *env = getenv("TFHE_HIP_POOL_RCCL");
  return env ? atoi(env) : 1;
}

// The pool's communicator (root pool's mutex held): created once, reused by every key replication and every
// device-resident batch call, destroyed by tfhe_hip_pool_destroy.  nullptr: take the peer-copy path.
std::vector<ncclComm_t> *pool_comms(tfhe_hip_pool *p) {
  tfhe_hip_pool *root = p->root();
  if (root->comm_state == 1) return &root->comms;
  if (root->comm_state < 0) return nullptr;
  root->comm_state = -1;
  const int n = (int)root->ctxs.size();
  const int mode = pool_rccl_mode();
  if (mode == 0 || n < (mode >= 2 ? 1 : 2)) return nullptr;
  RcclApi &R = rccl_api();
  if (!R.ok) return nullptr;
  std::vector<int> devs;
  for (auto *c : root->ctxs) devs.push_back(c->device);
  for (int i = 0; i < n; ++i)  // one rank per GPU: a pool that repeats a device takes the peer-copy path
    for (int j = i + 1; j < n; ++j)
      if (devs[(size_t)i] == devs[(size_t)j]) return nullptr;
  int prev = -1;
  (void)hipGetDevice(&prev);
  std::vector<ncclComm_t> comms((size_t)n, nullptr);
  const auto t0 = std::chrono::steady_clock::now();
  const bool good = R.CommInitAll(comms.data(), n, devs.data()) == ncclSuccess;
  root->comm_create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (prev >= 0) (void)hipSetDevice(prev);
  if (!good) {
    (void)hipGetLastError();
    return nullptr;
  }
  root->comms = std::move(comms);
  root->comm_state = 1;
  return &root->comms;
}

// A call failed INSIDE an open RCCL group (or its GroupEnd did): sends / receives of the group may have been launched
// without their partners, so the communicator's state is undefined and the streams it was used on may never drain.
// The pool stops using it: the communicators are aborted (ncclCommAbort where the library has it, which also releases
// kernels that wait for a partner) and every later call takes the peer-copy path.
void pool_drop_comms(tfhe_hip_pool *p) {
  tfhe_hip_pool *root = p->root();
  if (root->comm_state != 1) return;
  RcclApi &R = rccl_api();
  // (no ncclCommAbort in this librccl: the communicators are LEAKED -- ncclCommDestroy waits for outstanding work, and a
  // send whose receive was never launched would hang the caller here, with the pool's mutex held)
  for (ncclComm_t c : root->comms)
    if (c && R.CommAbort) (void)R.CommAbort(c);
  root->comms.clear();
  root->comm_state = -1;
  root->comms_dropped = true;
  (void)hipGetLastError();
}

// member i >= 1: drained, its key buffers allocated, no valid key until finish_replica
int prepare_replica(tfhe_hip_ctx *member) {
  tfhe_hip_ctx *ctx = member;
  ENTER(ctx);
  if (ctx->scratch_owned) HIPCHK(ctx, hipStreamSynchronize(ctx->scratch_owner));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scratch_owned = false;
  comb_quiesce(ctx);
  const tfhe_hip_params &P = ctx->P;
  ctx->K->key_loaded = ctx->K->reenc_loaded = false;
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
  comb_prepare(ctx);
  return TFHE_HIP_OK;
}

// true: every member holds member 0's key.  false: nothing usable happened (members >= 1 may hold garbage and are
// marked unloaded): the caller takes the peer-copy path.  The root pool's mutex is held by the caller, which keeps
// every other pool call (the parent's and its views': they share that mutex) off the members' streams meanwhile.
bool replicate_key_rccl(tfhe_hip_pool *p) {
  const int n = (int)p->ctxs.size();
  std::vector<ncclComm_t> *comms = pool_comms(p);
  if (!comms) return false;
  RcclApi &R = rccl_api();
  for (int i = 1; i < n; ++i)
    if (prepare_replica(p->ctxs[(size_t)i]) != TFHE_HIP_OK) return false;
  int prev = -1;
  (void)hipGetDevice(&prev);
  const tfhe_hip_params &P = p->ctxs[0]->P;
  const size_t bytes[3] = {(size_t)P.n * 2 * P.l * 2 * kN * sizeof(double),
                           (size_t)kN * P.t * (1u << P.basebit) * ksk_row_words(P.n) * 4, (size_t)2 * kN * 4};
  // the members' own mutexes: a member borrowed with tfhe_hip_pool_ctx() and used from another thread waits
  std::vector<std::unique_lock<FairMutex>> held;
  for (int i = 0; i < n; ++i) {
    tfhe_hip_ctx *c = p->ctxs[(size_t)i];
    held.emplace_back((c->parent ? c->parent : c)->mu);
  }
  bool good = true;
  for (int b = 0; b < 3 && good; ++b) {
    good = R.GroupStart() == ncclSuccess;
    for (int i = 0; i < n && good; ++i) {
      tfhe_hip_ctx *c = p->ctxs[(size_t)i];
      tfhe_hip_ctx *base = c->parent ? c->parent : c;
      KeyState &k = c->own;
      void *buf = b == 0 ? (void *)k.d_bsk : b == 1 ? (void *)k.d_ksk : (void *)k.d_testvec;
      good = hipSetDevice(c->device) == hipSuccess &&
             R.Broadcast(buf, buf, bytes[b], ncclUint8, 0, (*comms)[(size_t)i], base->stream) == ncclSuccess;
    }
    good = (R.GroupEnd() == ncclSuccess) && good;
  }
  for (int i = 0; i < n; ++i) {
    tfhe_hip_ctx *c = p->ctxs[(size_t)i];
    tfhe_hip_ctx *base = c->parent ? c->parent : c;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(base->stream) != hipSuccess) good = false;
  }
  held.clear();
  if (prev >= 0) (void)hipSetDevice(prev);
  if (!good) {
    pool_drop_comms(p);  // a broadcast group that failed half-way: the communicator is not used again
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
  tfhe_hip_pool *root = p->root();
  const bool had_comm = root->comm_state == 1;
  auto t0 = std::chrono::steady_clock::now();
  struct Stamp {  // (the communicator's creation, when this call caused it, is reported on its own)
    tfhe_hip_pool *root;
    std::chrono::steady_clock::time_point t0;
    bool had_comm;
    ~Stamp() {
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (!had_comm && root->comm_state == 1) ms -= root->comm_create_ms;
      root->key_replication_ms = ms > 0 ? ms : 0.0;
    }
  } stamp{root, t0, had_comm};
  if (replicate_key_rccl(p)) return TFHE_HIP_OK;
  for (size_t i = 1; i < p->ctxs.size(); ++i) {
    const int rc = clone_key(p->ctxs[i], p->ctxs[0]);
    if (rc != TFHE_HIP_OK) return pool_fail(p, rc, "device " + std::to_string(p->ctxs[i]->device) + ": " + tfhe_hip_last_error(p->ctxs[i]));
  }
  return TFHE_HIP_OK;
}

// Members a batch of `count` is spread over: one device runs up to 256 ciphertexts in the time of one (the
// latency kernels give every ciphertext its own workgroup), so smaller batches are not cut thinner than that.
inline int pool_world_for(const tfhe_hip_pool *p, size_t count) {
  const size_t want = (count + 255) / 256;
  const size_t have = p->ctxs.size();
  return (int)(want < 1 ? 1 : (want < have ? want : have));
}

// run(ctx, lo, hi) on every non-empty shard, shard 0 on the calling thread; first failure wins.  A shard's error text
// is read on the thread that ran it (the text is per thread, err_slot).
template <class F>
int pool_map(tfhe_hip_pool *p, size_t count, F &&run) {
  const int world = pool_world_for(p, count);
  std::vector<int> rc((size_t)world, TFHE_HIP_OK);
  std::vector<std::string> text((size_t)world);
  auto shard = [&](int r, size_t lo, size_t hi) {
    tfhe_hip_ctx *c = p->ctxs[(size_t)r];
    rc[(size_t)r] = run(c, lo, hi);
    if (rc[(size_t)r] != TFHE_HIP_OK) text[(size_t)r] = tfhe_hip_last_error(c);
  };
  std::vector<std::thread> th;
  for (int r = 1; r < world; ++r) {
    size_t lo, hi;
    pool_shard(count, r, world, lo, hi);
    if (hi <= lo) continue;
    try {
      th.emplace_back([&, r, lo, hi] { shard(r, lo, hi); });
    } catch (const std::system_error &) {  // no thread to be had: this shard runs on the calling thread
      shard(r, lo, hi);
    }
  }
  {
    size_t lo, hi;
    pool_shard(count, 0, world, lo, hi);
    if (hi > lo) shard(0, lo, hi);
  }
  for (auto &t : th) t.join();
  for (int r = 0; r < world; ++r)
    if (rc[(size_t)r] != TFHE_HIP_OK)
      return pool_fail(p, rc[(size_t)r], "device " + std::to_string(p->ctxs[(size_t)r]->device) + ": " + text[(size_t)r]);
  return TFHE_HIP_OK;
}

// ---- a batch that is RESIDENT on one member's GPU (the *_dev pool calls) ---------------------------------------
// "scatter inputs / gather outputs as grouped ncclSend / ncclRecv when the batch is resident on one
// GPU".  The caller's operands and result are device pointers on member `home`'s GPU and the call is only
// ENQUEUED (like the single-context *_dev calls): on `stream` (home) and on the other members' own streams, tied
// together by events / by the RCCL kernels themselves.  Shard r of the order-preserving split runs on member
// (home + r) mod size, so shard 0 -- and a batch too small to be cut -- never leaves home.  Per call:
//   home stream   [ sends of shards 1.. ]  [ shard 0 computed in place ]  [ receives of results 1.. ]
//   peer stream   [ receive shard ]        [ compute from / into staging ] [ send result ]
// One grouped ncclSend/ncclRecv per (peer, operand): every xGMI link carries its own peer's shard concurrently
// (65,536 NAND over 8 GPUs: 2 x 23 MB in, 23 MB out per peer = 0.45 ms at 153 GB/s, against 42 ms of compute).
// Without a communicator (a repeated device, no librccl, TFHE_HIP_POOL_RCCL=0): hipMemcpyPeerAsync on the peer's
// stream behind an event of the home stream, and an event per peer ahead of whatever follows on the home stream.
struct PoolIn {
  const void *ptr = nullptr;  // on home's GPU; nullptr = operand absent
  size_t row_bytes = 0;       // per ciphertext when sharded, the whole operand otherwise
  bool sharded = true;        // false: every member needs all of it (a shared test vector)
};

// op(ctx, ins[5], out, m, stream): enqueue the batched operation for m ciphertexts on member `ctx`
template <class Op>
int pool_dev_map(tfhe_hip_pool *p, int home, size_t count, void *stream_v, const PoolIn (&ins)[5], void *out,
                 size_t out_row_bytes, Op &&op) {
  tfhe_hip_pool *root = p->root();
  const int size = (int)p->ctxs.size();
  if (home < 0 || home >= size) return pool_fail(p, TFHE_HIP_EINVAL, "no such pool member (home)");
  root->last_transport = "none";
  if (count == 0) return TFHE_HIP_OK;
  tfhe_hip_ctx *hctx = p->ctxs[(size_t)home];
  tfhe_hip_ctx *hbase = hctx->parent ? hctx->parent : hctx;
  hipStream_t hs = stream_v ? (hipStream_t)stream_v : hbase->stream;
  // `hs` goes to the member's *_dev entry point as it is (there NULL means "the context's own stream", so the default
  // stream is named by hipStreamLegacy); the event, wait and RCCL calls below take the runtime's own name for the
  // default stream, NULL (hipStreamWaitEvent on the hipStreamLegacy handle faults in this HIP runtime)
  hipStream_t hrt = hs == hipStreamLegacy ? (hipStream_t) nullptr : hs;
  const int world = pool_world_for(p, count);
  std::vector<ncclComm_t> *comms = world > 1 || pool_rccl_mode() >= 2 ? pool_comms(p) : nullptr;
  // TFHE_HIP_POOL_RCCL=2 on a pool of ONE member (plumbing test on a one-GPU box): home's own shard takes the remote
  // path through a self send / receive, so the same symbols, group structure and stream handling run as among peers
  const bool loopback = comms && size == 1 && pool_rccl_mode() >= 2;
  const bool any_remote = world > 1 || loopback;
  RcclApi &R = rccl_api();
  if (root->stage.size() < (size_t)size) root->stage.resize((size_t)size);
  int prev = -1;
  (void)hipGetDevice(&prev);
  struct Restore {
    int d;
    ~Restore() {
      if (d >= 0) (void)hipSetDevice(d);
    }
  } restore{prev};
  auto hipfail = [&](const char *what, hipError_t e, int member) {
    return pool_fail(p, TFHE_HIP_EHIP, "device " + std::to_string(p->ctxs[(size_t)member]->device) + ": " + what + ": " + hipGetErrorString(e));
  };
  auto ensure_on = [&](int member, DevBuf &b, size_t bytes) -> int {
    if (bytes <= b.cap) return TFHE_HIP_OK;
    hipError_t e = hipSetDevice(p->ctxs[(size_t)member]->device);
    if (e != hipSuccess) return hipfail("hipSetDevice", e, member);
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t want = bytes + bytes / 4;
    e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return pool_fail(p, TFHE_HIP_ENOMEM, "device " + std::to_string(p->ctxs[(size_t)member]->device) + ": hipMalloc pool staging");
    }
    b.cap = want;
    return TFHE_HIP_OK;
  };
  // transfer timing: begin records the first event of a new pair on `s` and returns the pair's index (-1: off), end
  // records the second event of pair `idx`
  using TimedVec = std::vector<tfhe_hip_pool::Timed>;
  auto timed_begin = [&](TimedVec &v, int member, hipStream_t s, long home_ix = -1) -> long {
    if (!root->timing || v.size() >= 4096) return -1;  // (a caller that never collects: stop at 4,096 pending pairs)
    (void)hipSetDevice(p->ctxs[(size_t)member]->device);
    tfhe_hip_pool::Timed t;
    if (hipEventCreate(&t.a) != hipSuccess) return -1;
    if (hipEventCreate(&t.b) != hipSuccess) {
      (void)hipEventDestroy(t.a);
      return -1;
    }
    (void)hipEventRecord(t.a, s);
    t.member = member;
    t.home_ix = home_ix;
    v.push_back(t);
    return (long)v.size() - 1;
  };
  auto timed_end = [&](TimedVec &v, long idx, hipStream_t s) {
    if (idx < 0) return;
    (void)hipSetDevice(p->ctxs[(size_t)v[(size_t)idx].member]->device);
    (void)hipEventRecord(v[(size_t)idx].b, s);
  };
  // shard table: shard r -> member, [lo, hi), remote?
  struct Sh {
    int member;
    size_t lo, hi;
    bool remote;
  };
  std::vector<Sh> shards;
  for (int r = 0; r < world; ++r) {
    size_t lo, hi;
    pool_shard(count, r, world, lo, hi);
    if (hi <= lo) continue;
    const int member = (home + r) % size;
    shards.push_back({member, lo, hi, member != home || loopback});
  }
  // 1. staging on the receiving members
  for (const Sh &sh : shards) {
    if (!sh.remote) continue;
    tfhe_hip_pool::Stage &st = root->stage[(size_t)sh.member];
    for (int k = 0; k < 5; ++k)
      if (ins[k].ptr) CHK(ensure_on(sh.member, st.in[k], ins[k].sharded ? (sh.hi - sh.lo) * ins[k].row_bytes : ins[k].row_bytes));
    CHK(ensure_on(sh.member, st.out, (sh.hi - sh.lo) * out_row_bytes));
    if (!st.done) {
      (void)hipSetDevice(p->ctxs[(size_t)sh.member]->device);
      const hipError_t e = hipEventCreateWithFlags(&st.done, hipEventDisableTiming);
      if (e != hipSuccess) return hipfail("hipEventCreate", e, sh.member);
    }
  }
  auto member_stream = [&](int member) {
    tfhe_hip_ctx *c = p->ctxs[(size_t)member];
    return (c->parent ? c->parent : c)->stream;
  };
  // 2. scatter
  if (any_remote) {
    root->last_transport = comms ? "rccl" : "peer-copy";
    if (comms) {
      // everything that can fail for a reason of ours is checked BEFORE the group opens (every device can be made
      // current, every staging buffer exists): inside it only the RCCL calls themselves can fail, and if one does the
      // communicator is dropped (pool_drop_comms) -- a send that was launched without its receive may never complete,
      // so after such an error the home stream may be poisoned and the caller should not wait on it
      for (const Sh &sh : shards) {
        if (!sh.remote) continue;
        const hipError_t e = hipSetDevice(p->ctxs[(size_t)sh.member]->device);
        if (e != hipSuccess) return hipfail("hipSetDevice", e, sh.member);
        for (int k = 0; k < 5; ++k)
          if (ins[k].ptr && !root->stage[(size_t)sh.member].in[k].p) return pool_fail(p, TFHE_HIP_EHIP, "pool staging buffer missing");
      }
      {
        const hipError_t e = hipSetDevice(hctx->device);
        if (e != hipSuccess) return hipfail("hipSetDevice", e, home);
      }
      // (events go in before the group opens: inside it nothing is enqueued yet)
      const long hix = timed_begin(root->ev_home, home, hrt);
      std::vector<long> tix(shards.size(), -1);
      for (size_t q = 0; q < shards.size(); ++q)
        if (shards[q].remote) tix[q] = timed_begin(root->ev_scatter, shards[q].member, member_stream(shards[q].member), hix);
      if (R.GroupStart() != ncclSuccess) return pool_fail(p, TFHE_HIP_EHIP, "ncclGroupStart (scatter)");
      bool good = true;  // (a group that was opened is always closed, whatever fails inside it)
      for (const Sh &sh : shards) {
        if (!sh.remote) continue;
        tfhe_hip_pool::Stage &st = root->stage[(size_t)sh.member];
        for (int k = 0; k < 5 && good; ++k) {
          if (!ins[k].ptr) continue;
          const size_t bytes = ins[k].sharded ? (sh.hi - sh.lo) * ins[k].row_bytes : ins[k].row_bytes;
          const unsigned char *src = (const unsigned char *)ins[k].ptr + (ins[k].sharded ? sh.lo * ins[k].row_bytes : 0);
          good = hipSetDevice(hctx->device) == hipSuccess &&
                 R.Send(src, bytes, ncclUint8, sh.member, (*comms)[(size_t)home], hrt) == ncclSuccess &&
                 hipSetDevice(p->ctxs[(size_t)sh.member]->device) == hipSuccess &&
                 R.Recv(st.in[k].p, bytes, ncclUint8, home, (*comms)[(size_t)sh.member], member_stream(sh.member)) == ncclSuccess;
          root->scatter_bytes += bytes;
        }
      }
      good = (R.GroupEnd() == ncclSuccess) && good;
      for (size_t q = 0; q < shards.size(); ++q)
        if (shards[q].remote) timed_end(root->ev_scatter, tix[q], member_stream(shards[q].member));
      timed_end(root->ev_home, hix, hrt);
      if (!good) {
        pool_drop_comms(p);
        return pool_fail(p, TFHE_HIP_EHIP, "RCCL scatter (ncclSend / ncclRecv) failed inside the group: communicator dropped, later calls use peer copies; the home stream may not drain");
      }
    } else {
      hipError_t e = hipSetDevice(hctx->device);
      if (e != hipSuccess) return hipfail("hipSetDevice", e, home);
      if (!root->ready || root->ready_device != hctx->device) {
        if (root->ready) (void)hipEventDestroy(root->ready);
        root->ready = nullptr;
        if ((e = hipEventCreateWithFlags(&root->ready, hipEventDisableTiming)) != hipSuccess) return hipfail("hipEventCreate", e, home);
        root->ready_device = hctx->device;
      }
      if ((e = hipEventRecord(root->ready, hrt)) != hipSuccess) return hipfail("hipEventRecord", e, home);
      for (const Sh &sh : shards) {
        if (!sh.remote) continue;
        tfhe_hip_pool::Stage &st = root->stage[(size_t)sh.member];
        const int ddev = p->ctxs[(size_t)sh.member]->device;
        hipStream_t ms = member_stream(sh.member);
        if ((e = hipSetDevice(ddev)) != hipSuccess) return hipfail("hipSetDevice", e, sh.member);
        if ((e = hipStreamWaitEvent(ms, root->ready, 0)) != hipSuccess) return hipfail("hipStreamWaitEvent", e, sh.member);
        const long tix = timed_begin(root->ev_scatter, sh.member, ms);
        for (int k = 0; k < 5; ++k) {
          if (!ins[k].ptr) continue;
          const size_t bytes = ins[k].sharded ? (sh.hi - sh.lo) * ins[k].row_bytes : ins[k].row_bytes;
          const unsigned char *src = (const unsigned char *)ins[k].ptr + (ins[k].sharded ? sh.lo * ins[k].row_bytes : 0);
          if ((e = hipMemcpyPeerAsync(st.in[k].p, ddev, src, hctx->device, bytes, ms)) != hipSuccess)
            return hipfail("hipMemcpyPeerAsync (scatter)", e, sh.member);
          root->scatter_bytes += bytes;
        }
        timed_end(root->ev_scatter, tix, ms);
      }
    }
  }
  // 3. compute: remote shards first (their members are idle until then), home's own shard last
  for (int pass = 0; pass < 2; ++pass)
    for (const Sh &sh : shards) {
      if ((pass == 0) != sh.remote) continue;
      const void *ptrs[5];
      void *o;
      if (sh.remote) {
        tfhe_hip_pool::Stage &st = root->stage[(size_t)sh.member];
        for (int k = 0; k < 5; ++k) ptrs[k] = ins[k].ptr ? st.in[k].p : nullptr;
        o = st.out.p;
      } else {
        for (int k = 0; k < 5; ++k)
          ptrs[k] = ins[k].ptr ? (const unsigned char *)ins[k].ptr + (ins[k].sharded ? sh.lo * ins[k].row_bytes : 0) : nullptr;
        o = (unsigned char *)out + sh.lo * out_row_bytes;
      }
      tfhe_hip_ctx *c = p->ctxs[(size_t)sh.member];
      const int rc = op(c, ptrs, o, sh.hi - sh.lo, sh.remote ? (void *)member_stream(sh.member) : (void *)hs);
      if (rc != TFHE_HIP_OK)
        return pool_fail(p, rc, "device " + std::to_string(c->device) + ": " + tfhe_hip_last_error(c));
    }
  // 4. gather
  if (any_remote) {
    if (comms) {
      // per shard a pair on the SENDING member's stream (its first event follows the member's compute, so the pair
      // brackets the transfer and whatever it waits for on the home side); one pair on the home stream around the
      // group's receives (which also waits for the slowest member's result)
      const long hix = timed_begin(root->ev_home, home, hrt);
      if (hix >= 0) root->ev_home[(size_t)hix].gather = true;
      std::vector<long> gix(shards.size(), -1);
      for (size_t q = 0; q < shards.size(); ++q)
        if (shards[q].remote) gix[q] = timed_begin(root->ev_gather, shards[q].member, member_stream(shards[q].member), hix);
      if (R.GroupStart() != ncclSuccess) return pool_fail(p, TFHE_HIP_EHIP, "ncclGroupStart (gather)");
      bool good = true;
      for (const Sh &sh : shards) {
        if (!sh.remote || !good) continue;
        tfhe_hip_pool::Stage &st = root->stage[(size_t)sh.member];
        const size_t bytes = (sh.hi - sh.lo) * out_row_bytes;
        good = hipSetDevice(p->ctxs[(size_t)sh.member]->device) == hipSuccess &&
               R.Send(st.out.p, bytes, ncclUint8, home, (*comms)[(size_t)sh.member], member_stream(sh.member)) == ncclSuccess &&
               hipSetDevice(hctx->device) == hipSuccess &&
               R.Recv((unsigned char *)out + sh.lo * out_row_bytes, bytes, ncclUint8, sh.member, (*comms)[(size_t)home], hrt) == ncclSuccess;
        root->gather_bytes += bytes;
      }
      good = (R.GroupEnd() == ncclSuccess) && good;
      for (size_t q = 0; q < shards.size(); ++q)
        if (shards[q].remote) timed_end(root->ev_gather, gix[q], member_stream(shards[q].member));
      timed_end(root->ev_home, hix, hrt);
      if (!good) {
        pool_drop_comms(p);
        return pool_fail(p, TFHE_HIP_EHIP, "RCCL gather (ncclSend / ncclRecv) failed inside the group: communicator dropped, later calls use peer copies; the home stream may not drain");
      }
    } else {
      for (const Sh &sh : shards) {
        if (!sh.remote) continue;
        tfhe_hip_pool::Stage &st = root->stage[(size_t)sh.member];
        const int sdev = p->ctxs[(size_t)sh.member]->device;
        hipStream_t ms = member_stream(sh.member);
        const size_t bytes = (sh.hi - sh.lo) * out_row_bytes;
        hipError_t e = hipSetDevice(sdev);
        if (e != hipSuccess) return hipfail("hipSetDevice", e, sh.member);
        const long gix = timed_begin(root->ev_gather, sh.member, ms);
        if ((e = hipMemcpyPeerAsync((unsigned char *)out + sh.lo * out_row_bytes, hctx->device, st.out.p, sdev, bytes, ms)) != hipSuccess)
          return hipfail("hipMemcpyPeerAsync (gather)", e, sh.member);
        timed_end(root->ev_gather, gix, ms);
        if ((e = hipEventRecord(st.done, ms)) != hipSuccess) return hipfail("hipEventRecord", e, sh.member);
        root->gather_bytes += bytes;
      }
      hipError_t e = hipSetDevice(hctx->device);
      if (e != hipSuccess) return hipfail("hipSetDevice", e, home);
      for (const Sh &sh : shards)
        if (sh.remote && (e = hipStreamWaitEvent(hrt, root->stage[(size_t)sh.member].done, 0)) != hipSuccess)
          return hipfail("hipStreamWaitEvent", e, home);
    }
  }
  ++root->dev_calls;
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
  // direct xGMI for the peer-copy transport (key clone, shards of the device-resident calls when no communicator is to be
  // had): peer access between every pair of distinct devices, where the topology allows it (errors are not fatal: the
  // runtime stages such copies through the host)
  {
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int i = 0; i < ndev; ++i)
      for (int j = 0; j < ndev; ++j) {
        if (devices[i] == devices[j]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) continue;
        if (hipSetDevice(devices[i]) == hipSuccess) (void)hipDeviceEnablePeerAccess(devices[j], 0);
      }
    (void)hipGetLastError();  // (hipErrorPeerAccessAlreadyEnabled and the like)
    if (prev >= 0) (void)hipSetDevice(prev);
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
  if (p->parent) {  // a key view: its keys go, everything else is the parent's
    tfhe_hip_pool *root = p->parent;
    bool last_of_dying = false;
    {
      std::lock_guard<FairMutex> lk(root->own_mu);
      for (auto *c : p->ctxs) tfhe_hip_ctx_destroy(c);  // (drains the member's queued work first)
      last_of_dying = --root->views == 0 && root->dying;
    }
    delete p;
    if (last_of_dying) tfhe_hip_pool_destroy(root);  // the parent was destroyed first: it has waited for its views
    return;
  }
  {
    // destroyed before its views (the header asks for the opposite order): they share this pool's mutex, staging and
    // communicator, so it stays alive until the last of them goes
    std::lock_guard<FairMutex> lk(p->own_mu);
    if (p->views > 0) {
      p->dying = true;
      return;
    }
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  // queued device-resident calls may still use the staging buffers and the communicator: drain the members first
  // (not after a group failed half-way: a stream that carries a send without its receive never drains)
  for (auto *c : p->ctxs) {
    if (!p->comms_dropped && hipSetDevice(c->device) == hipSuccess) (void)hipStreamSynchronize(c->stream);
  }
  if (p->comm_state == 1)
    for (ncclComm_t c : p->comms)
      if (c) (void)rccl_api().CommDestroy(c);
  for (size_t i = 0; i < p->stage.size() && i < p->ctxs.size(); ++i) {
    (void)hipSetDevice(p->ctxs[i]->device);
    for (DevBuf &b : p->stage[i].in)
      if (b.p) (void)hipFree(b.p);
    if (p->stage[i].out.p) (void)hipFree(p->stage[i].out.p);
    if (p->stage[i].done) (void)hipEventDestroy(p->stage[i].done);
  }
  if (p->ready) (void)hipEventDestroy(p->ready);
  for (auto *v : {&p->ev_scatter, &p->ev_gather, &p->ev_home})
    for (auto &e : *v) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
  if (prev >= 0) (void)hipSetDevice(prev);
  for (auto *c : p->ctxs) tfhe_hip_ctx_destroy(c);
  delete p;
}

// A key view of a pool: one key view per member context (same devices, same streams and scratch, its own cloud key);
// it shares the parent's mutex, staging buffers and communicator.
int tfhe_hip_pool_key_create(tfhe_hip_pool *pool, tfhe_hip_pool **out) {
  if (!out) return TFHE_HIP_EINVAL;
  *out = nullptr;
  if (!pool) return TFHE_HIP_EINVAL;
  tfhe_hip_pool *root = pool->root();
  std::lock_guard<FairMutex> lk(root->own_mu);
  tfhe_hip_pool *v = new tfhe_hip_pool();
  v->parent = root;
  for (auto *c : root->ctxs) {
    tfhe_hip_ctx *kv = nullptr;
    const int rc = tfhe_hip_key_create(c, &kv);
    if (rc != TFHE_HIP_OK) {
      for (auto *x : v->ctxs) tfhe_hip_ctx_destroy(x);
      delete v;
      return rc;
    }
    v->ctxs.push_back(kv);
  }
  ++root->views;
  *out = v;
  return TFHE_HIP_OK;
}

int tfhe_hip_pool_size(const tfhe_hip_pool *p) { return p ? (int)p->ctxs.size() : 0; }

tfhe_hip_ctx *tfhe_hip_pool_ctx(tfhe_hip_pool *p, int i) {
  return (p && i >= 0 && (size_t)i < p->ctxs.size()) ? p->ctxs[(size_t)i] : nullptr;
}

const char *tfhe_hip_pool_last_error(const tfhe_hip_pool *p) { return p ? err_text(p->id) : g_create_error.c_str(); }

// "rccl" when the pool's last cloud key reached its members by ncclBroadcast, "peer-copy" when by hipMemcpyPeer
// (or when there was nothing to replicate).
const char *tfhe_hip_pool_key_transport(const tfhe_hip_pool *p) { return (p && p->replicated_by_rccl) ? "rccl" : "peer-copy"; }

// how the last device-resident (_dev) call moved its shards: "rccl", "peer-copy", or "none" (nothing left home)
const char *tfhe_hip_pool_data_transport(const tfhe_hip_pool *p) { return p ? p->root()->last_transport : "none"; }

int tfhe_hip_pool_members_for(const tfhe_hip_pool *p, size_t count) { return p ? pool_world_for(p, count) : 0; }

void tfhe_hip_pool_shard(size_t count, int shard, int nshards, size_t *lo, size_t *hi) {
  size_t a = 0, b = 0;
  if (nshards > 0 && shard >= 0 && shard < nshards) pool_shard(count, shard, nshards, a, b);
  if (lo) *lo = a;
  if (hi) *hi = b;
}

#define POOL_ENTER(p)               \
  if (!(p)) return TFHE_HIP_EINVAL; \
  std::lock_guard<FairMutex> plk_((p)->root()->own_mu)

// Small host-pointer calls from concurrent threads (`Send + Sync`, src/bootstrap/mod.rs:23): they do not queue on the
// pool's mutex; each goes to the member whose combining front end (combine.hpp) has the least work queued and in
// flight, where it shares launches with the other threads' calls.  (Key loads and device-resident calls keep the
// pool's mutex; as for a context, changing a key while calls under it are in flight is the caller's to avoid.)
namespace {
// (up to 256 ciphertexts: beyond that a pool call is cut over several members, pool_world_for -- two devices run 512 in the
// time one runs 256)
inline bool pool_small(const tfhe_hip_pool *p, size_t count) { return p && !p->ctxs.empty() && count <= 256 && comb_takes(p->ctxs[0], count); }
inline tfhe_hip_ctx *pool_least_loaded(tfhe_hip_pool *p) {
  // members that share a device count once (the first of them): two front ends on one GPU would launch on two streams that
  // need not overlap, and every call would wait for the other member's launch (measured on a pool of {0, 0}: 1.5 k gates/s
  // from 8 threads against 3.1 k through one member, profiles/exp/logs/r6c_front_end.log)
  tfhe_hip_ctx *best = p->ctxs[0];
  size_t best_load = ~(size_t)0;
  for (size_t i = 0; i < p->ctxs.size(); ++i) {
    tfhe_hip_ctx *c = p->ctxs[i];
    bool repeat = false;
    for (size_t j = 0; j < i && !repeat; ++j) repeat = p->ctxs[j]->device == c->device;
    if (repeat) continue;
    const tfhe_hip_ctx *base = c->parent ? c->parent : c;
    const size_t load = base->comb ? base->comb->pending.load(std::memory_order_relaxed) : 0;
    if (load < best_load) {
      best = c;
      best_load = load;
    }
  }
  return best;
}
template <class F>
int pool_small_call(tfhe_hip_pool *p, F &&call) {
  tfhe_hip_ctx *c = pool_least_loaded(p);
  const int rc = call(c);
  if (rc != TFHE_HIP_OK) return pool_fail(p, rc, "device " + std::to_string(c->device) + ": " + tfhe_hip_last_error(c));
  return TFHE_HIP_OK;
}
}  // namespace
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

// ---- host-pointer batch calls: one host thread per shard ---------------------------------------------------------
int tfhe_hip_pool_batch_gate(tfhe_hip_pool *p, int gate, const uint32_t *a, const uint32_t *b, uint32_t *out,
                             size_t count) {
  if (pool_small(p, count)) {
    if (!a || !out) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
    return pool_small_call(p, [&](tfhe_hip_ctx *c) { return tfhe_hip_batch_gate(c, gate, a, b, out, count); });
  }
  POOL_ENTER(p);
  if (count && (!a || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_gate(c, gate, a + lo * w, b ? b + lo * w : nullptr, out + lo * w, hi - lo);
  });
}

int tfhe_hip_pool_batch_gates_mixed(tfhe_hip_pool *p, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                    uint32_t *out, size_t count) {
  if (pool_small(p, count)) {
    if (!gates || !a || !b || !out) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
    return pool_small_call(p, [&](tfhe_hip_ctx *c) { return tfhe_hip_batch_gates_mixed(c, gates, a, b, out, count); });
  }
  POOL_ENTER(p);
  if (count && (!gates || !a || !b || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_gates_mixed(c, gates + lo, a + lo * w, b + lo * w, out + lo * w, hi - lo);
  });
}

int tfhe_hip_pool_batch_gates_mixed_nks(tfhe_hip_pool *p, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                                        uint32_t *out, size_t count) {
  if (pool_small(p, count)) {
    if (!gates || !a || !b || !out) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
    return pool_small_call(p, [&](tfhe_hip_ctx *c) { return tfhe_hip_batch_gates_mixed_nks(c, gates, a, b, out, count); });
  }
  POOL_ENTER(p);
  if (count && (!gates || !a || !b || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_gates_mixed_nks(c, gates + lo, a + lo * w, b + lo * w, out + lo * w, hi - lo);
  });
}

int tfhe_hip_pool_batch_bootstrap(tfhe_hip_pool *p, const uint32_t *in, const uint32_t *testvec, int per_ct,
                                  int keyswitch, uint32_t *out, size_t count) {
  if (pool_small(p, count)) {
    if (!in || !out) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
    return pool_small_call(p, [&](tfhe_hip_ctx *c) { return tfhe_hip_batch_bootstrap(c, in, testvec, per_ct, keyswitch, out, count); });
  }
  POOL_ENTER(p);
  if (count && (!in || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  const size_t tvs = (testvec && per_ct) ? (size_t)2 * kN : 0;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_bootstrap(c, in + lo * w, testvec ? testvec + lo * tvs : nullptr, per_ct, keyswitch, out + lo * w,
                                    hi - lo);
  });
}

int tfhe_hip_pool_batch_tlwe_lincomb(tfhe_hip_pool *p, uint32_t ca, const uint32_t *a, uint32_t cb, const uint32_t *b,
                                     uint32_t cconst, uint32_t *out, size_t count) {
  POOL_ENTER(p);
  if (count && (!a || !out || (cb && !b))) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_tlwe_lincomb(c, ca, a + lo * w, cb, b ? b + lo * w : nullptr, cconst, out + lo * w, hi - lo);
  });
}

int tfhe_hip_pool_batch_lincomb_bootstrap(tfhe_hip_pool *p, uint32_t ca, const uint32_t *a, uint32_t cb,
                                          const uint32_t *b, uint32_t cconst, const uint32_t *testvec, int per_ct,
                                          int keyswitch, uint32_t *out, size_t count) {
  if (pool_small(p, count)) {
    if (!a || !out || (cb && !b)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
    return pool_small_call(p, [&](tfhe_hip_ctx *c) {
      return tfhe_hip_batch_lincomb_bootstrap(c, ca, a, cb, b, cconst, testvec, per_ct, keyswitch, out, count);
    });
  }
  POOL_ENTER(p);
  if (count && (!a || !out || (cb && !b))) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  const size_t tvs = (testvec && per_ct) ? (size_t)2 * kN : 0;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_lincomb_bootstrap(c, ca, a + lo * w, cb, b ? b + lo * w : nullptr, cconst,
                                            testvec ? testvec + lo * tvs : nullptr, per_ct, keyswitch, out + lo * w, hi - lo);
  });
}

int tfhe_hip_pool_batch_mux(tfhe_hip_pool *p, int naive, const uint32_t *a, const uint32_t *b, const uint32_t *c3,
                            uint32_t *out, size_t count) {
  if (pool_small(p, count)) {
    if (!a || !b || !c3 || !out) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
    return pool_small_call(p, [&](tfhe_hip_ctx *c) { return tfhe_hip_batch_mux(c, naive, a, b, c3, out, count); });
  }
  POOL_ENTER(p);
  if (count && (!a || !b || !c3 || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_mux(c, naive, a + lo * w, b + lo * w, c3 + lo * w, out + lo * w, hi - lo);
  });
}

int tfhe_hip_pool_batch_blind_rotate(tfhe_hip_pool *p, const uint32_t *in, const uint32_t *testvec, uint32_t *out_trlwe,
                                     size_t count) {
  if (pool_small(p, count)) {
    if (!in || !out_trlwe) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
    return pool_small_call(p, [&](tfhe_hip_ctx *c) { return tfhe_hip_batch_blind_rotate(c, in, testvec, out_trlwe, count); });
  }
  POOL_ENTER(p);
  if (count && (!in || !out_trlwe)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = (size_t)p->ctxs[0]->P.n + 1;
  return pool_map(p, count, [&](tfhe_hip_ctx *c, size_t lo, size_t hi) {
    return tfhe_hip_batch_blind_rotate(c, in + lo * w, testvec, out_trlwe + lo * (size_t)2 * kN, hi - lo);
  });
}

// ---- device-resident batch calls: the batch lives on member `home`'s GPU (pool_dev_map) -------------------------
int tfhe_hip_pool_batch_gate_dev(tfhe_hip_pool *p, int home, int gate, const uint32_t *a, const uint32_t *b,
                                 uint32_t *out, size_t count, void *stream) {
  POOL_ENTER(p);
  GatePrep gp;
  if (!gate_prep(gate, gp)) return pool_fail(p, TFHE_HIP_EINVAL, "unknown gate");
  if (count && (!a || !out || (gp.cb && !b))) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = ((size_t)p->ctxs[0]->P.n + 1) * 4;
  const PoolIn ins[5] = {{a, w, true}, {gp.cb ? b : nullptr, w, true}, {}, {}, {}};
  return pool_dev_map(p, home, count, stream, ins, out, w, [&](tfhe_hip_ctx *c, const void *const *q, void *o, size_t m, void *s) {
    return tfhe_hip_batch_gate_dev(c, gate, (const uint32_t *)q[0], (const uint32_t *)q[1], (uint32_t *)o, m, s);
  });
}

int tfhe_hip_pool_batch_gates_mixed_dev(tfhe_hip_pool *p, int home, const uint8_t *gates, const uint32_t *a,
                                        const uint32_t *b, uint32_t *out, size_t count, void *stream) {
  POOL_ENTER(p);
  if (count && (!gates || !a || !b || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = ((size_t)p->ctxs[0]->P.n + 1) * 4;
  const PoolIn ins[5] = {{a, w, true}, {b, w, true}, {gates, 1, true}, {}, {}};
  return pool_dev_map(p, home, count, stream, ins, out, w, [&](tfhe_hip_ctx *c, const void *const *q, void *o, size_t m, void *s) {
    return tfhe_hip_batch_gates_mixed_dev(c, (const uint8_t *)q[2], (const uint32_t *)q[0], (const uint32_t *)q[1], (uint32_t *)o, m, s);
  });
}

int tfhe_hip_pool_batch_gates_mixed_nks_dev(tfhe_hip_pool *p, int home, const uint8_t *gates, const uint32_t *a,
                                            const uint32_t *b, uint32_t *out, size_t count, void *stream) {
  POOL_ENTER(p);
  if (count && (!gates || !a || !b || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = ((size_t)p->ctxs[0]->P.n + 1) * 4;
  const PoolIn ins[5] = {{a, w, true}, {b, w, true}, {gates, 1, true}, {}, {}};
  return pool_dev_map(p, home, count, stream, ins, out, w, [&](tfhe_hip_ctx *c, const void *const *q, void *o, size_t m, void *s) {
    return tfhe_hip_batch_gates_mixed_nks_dev(c, (const uint8_t *)q[2], (const uint32_t *)q[0], (const uint32_t *)q[1], (uint32_t *)o, m, s);
  });
}

int tfhe_hip_pool_batch_bootstrap_dev(tfhe_hip_pool *p, int home, const uint32_t *in, const uint32_t *testvec, int per_ct,
                                      int keyswitch, uint32_t *out, size_t count, void *stream) {
  POOL_ENTER(p);
  if (count && (!in || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = ((size_t)p->ctxs[0]->P.n + 1) * 4;
  const PoolIn ins[5] = {{in, w, true}, {testvec, (size_t)2 * kN * 4, testvec && per_ct}, {}, {}, {}};
  return pool_dev_map(p, home, count, stream, ins, out, w, [&](tfhe_hip_ctx *c, const void *const *q, void *o, size_t m, void *s) {
    return tfhe_hip_batch_bootstrap_dev(c, (const uint32_t *)q[0], (const uint32_t *)q[1], per_ct, keyswitch, (uint32_t *)o, m, s);
  });
}

int tfhe_hip_pool_batch_tlwe_lincomb_dev(tfhe_hip_pool *p, int home, uint32_t ca, const uint32_t *a, uint32_t cb,
                                         const uint32_t *b, uint32_t cconst, uint32_t *out, size_t count, void *stream) {
  POOL_ENTER(p);
  if (count && (!a || !out || (cb && !b))) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = ((size_t)p->ctxs[0]->P.n + 1) * 4;
  const PoolIn ins[5] = {{a, w, true}, {cb ? b : nullptr, w, true}, {}, {}, {}};
  return pool_dev_map(p, home, count, stream, ins, out, w, [&](tfhe_hip_ctx *c, const void *const *q, void *o, size_t m, void *s) {
    return tfhe_hip_batch_tlwe_lincomb_dev(c, ca, (const uint32_t *)q[0], cb, (const uint32_t *)q[1], cconst, (uint32_t *)o, m, s);
  });
}

int tfhe_hip_pool_batch_lincomb_bootstrap_dev(tfhe_hip_pool *p, int home, uint32_t ca, const uint32_t *a, uint32_t cb,
                                              const uint32_t *b, uint32_t cconst, const uint32_t *testvec, int per_ct,
                                              int keyswitch, uint32_t *out, size_t count, void *stream) {
  POOL_ENTER(p);
  if (count && (!a || !out || (cb && !b))) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = ((size_t)p->ctxs[0]->P.n + 1) * 4;
  const PoolIn ins[5] = {{a, w, true}, {cb ? b : nullptr, w, true}, {testvec, (size_t)2 * kN * 4, testvec && per_ct}, {}, {}};
  return pool_dev_map(p, home, count, stream, ins, out, w, [&](tfhe_hip_ctx *c, const void *const *q, void *o, size_t m, void *s) {
    return tfhe_hip_batch_lincomb_bootstrap_dev(c, ca, (const uint32_t *)q[0], cb, (const uint32_t *)q[1], cconst,
                                                (const uint32_t *)q[2], per_ct, keyswitch, (uint32_t *)o, m, s);
  });
}

int tfhe_hip_pool_batch_mux_dev(tfhe_hip_pool *p, int home, int naive, const uint32_t *a, const uint32_t *b,
                                const uint32_t *c3, uint32_t *out, size_t count, void *stream) {
  POOL_ENTER(p);
  if (count && (!a || !b || !c3 || !out)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = ((size_t)p->ctxs[0]->P.n + 1) * 4;
  const PoolIn ins[5] = {{a, w, true}, {b, w, true}, {c3, w, true}, {}, {}};
  return pool_dev_map(p, home, count, stream, ins, out, w, [&](tfhe_hip_ctx *c, const void *const *q, void *o, size_t m, void *s) {
    return tfhe_hip_batch_mux_dev(c, naive, (const uint32_t *)q[0], (const uint32_t *)q[1], (const uint32_t *)q[2], (uint32_t *)o, m, s);
  });
}

int tfhe_hip_pool_batch_blind_rotate_dev(tfhe_hip_pool *p, int home, const uint32_t *in, const uint32_t *testvec,
                                         uint32_t *out_trlwe, size_t count, void *stream) {
  POOL_ENTER(p);
  if (count && (!in || !out_trlwe)) return pool_fail(p, TFHE_HIP_EINVAL, "null pointer");
  const size_t w = ((size_t)p->ctxs[0]->P.n + 1) * 4;
  const PoolIn ins[5] = {{in, w, true}, {testvec, (size_t)2 * kN * 4, false}, {}, {}, {}};
  return pool_dev_map(p, home, count, stream, ins, out_trlwe, (size_t)2 * kN * 4, [&](tfhe_hip_ctx *c, const void *const *q, void *o, size_t m, void *s) {
    return tfhe_hip_batch_blind_rotate_dev(c, (const uint32_t *)q[0], (const uint32_t *)q[1], (uint32_t *)o, m, s);
  });
}

// Block until every member has finished what the pool's *_dev calls enqueued (home streams passed by the caller are
// the caller's to synchronise; the members' own streams are drained here).
int tfhe_hip_pool_synchronize(tfhe_hip_pool *p) {
  POOL_ENTER(p);
  for (auto *c : p->ctxs) {
    const int rc = tfhe_hip_synchronize(c);
    if (rc != TFHE_HIP_OK) return pool_fail(p, rc, "device " + std::to_string(c->device) + ": " + tfhe_hip_last_error(c));
  }
  return TFHE_HIP_OK;
}

int tfhe_hip_pool_set_profiling(tfhe_hip_pool *p, int enabled) {
  POOL_ENTER(p);
  p->root()->timing = enabled != 0;
  for (auto *c : p->ctxs) {
    const int rc = tfhe_hip_set_profiling(c, enabled);
    if (rc != TFHE_HIP_OK) return pool_fail(p, rc, "device " + std::to_string(c->device) + ": " + tfhe_hip_last_error(c));
  }
  return TFHE_HIP_OK;
}

int tfhe_hip_pool_get_transfer_times(tfhe_hip_pool *p, tfhe_hip_pool_transfer_times *out) {
  if (!out) return TFHE_HIP_EINVAL;
  POOL_ENTER(p);
  tfhe_hip_pool *root = p->root();
  memset(out, 0, sizeof(*out));
  int prev = -1;
  (void)hipGetDevice(&prev);
  auto elapsed = [&](tfhe_hip_pool::Timed &t) -> double {  // < 0: not measurable
    (void)hipSetDevice(root->ctxs[(size_t)t.member]->device);
    float ms = 0.f;
    if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) return (double)ms;
    (void)hipGetLastError();
    return -1.0;
  };
  std::vector<double> home_ms(root->ev_home.size(), -1.0);
  for (size_t i = 0; i < root->ev_home.size(); ++i) home_ms[i] = elapsed(root->ev_home[i]);
  // between(x, y): ms from event x to event y on ONE device (negative when y fired first), -1e30 if not measurable
  auto between = [&](hipEvent_t x, hipEvent_t y) -> double {
    float ms = 0.f;
    if (hipEventSynchronize(x) == hipSuccess && hipEventSynchronize(y) == hipSuccess && hipEventElapsedTime(&ms, x, y) == hipSuccess)
      return (double)ms;
    (void)hipGetLastError();
    return -1e30;
  };
  auto drain = [&](std::vector<tfhe_hip_pool::Timed> &v, double &sum, double &mx) {
    for (auto &t : v) {
      double ms = elapsed(t);
      if (t.home_ix >= 0 && (size_t)t.home_ix < home_ms.size()) {
        tfhe_hip_pool::Timed &h = root->ev_home[(size_t)t.home_ix];
        const bool same_device = root->ctxs[(size_t)h.member]->device == root->ctxs[(size_t)t.member]->device;
        const double ab = same_device ? between(t.a, h.a) : -1e30, bb = same_device ? between(t.b, h.b) : -1e30;
        if (ab > -1e29 && bb > -1e29) {
          // both brackets on ONE device (a pool that repeats a device, the self send / receive of the plumbing test): the
          // events are comparable, so the transfer is measured exactly -- from the moment the LATER end arrived to the
          // moment the later end finished (a self copy runs inside only one of the two RCCL kernels)
          const double span = between(ab >= 0 ? h.a : t.a, bb >= 0 ? h.b : t.b);
          if (span > -1e29) ms = span;
        } else if (home_ms[(size_t)t.home_ix] >= 0 && (ms < 0 || home_ms[(size_t)t.home_ix] < ms)) {
          // different devices: event times are not comparable; the shorter bracket (see tfhe_hip_pool::Timed)
          ms = home_ms[(size_t)t.home_ix];
        }
      }
      if (ms >= 0) {
        sum += ms;
        if (ms > mx) mx = ms;
      }
    }
    for (auto &t : v) {
      (void)hipEventDestroy(t.a);
      (void)hipEventDestroy(t.b);
    }
    v.clear();
  };
  drain(root->ev_scatter, out->scatter_ms_sum, out->scatter_ms_max);
  drain(root->ev_gather, out->gather_ms_sum, out->gather_ms_max);
  // the home-stream brackets on their own: one per call and direction, around the whole RCCL group (all peers' transfers,
  // and for a gather the wait for the slowest member's compute) -- an upper bound of any one transfer in it
  for (size_t i = 0; i < root->ev_home.size(); ++i)
    if (home_ms[i] >= 0) (root->ev_home[i].gather ? out->gather_group_ms_sum : out->scatter_group_ms_sum) += home_ms[i];
  for (auto &t : root->ev_home) {
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  root->ev_home.clear();
  if (prev >= 0) (void)hipSetDevice(prev);
  out->scatter_bytes = root->scatter_bytes;
  out->gather_bytes = root->gather_bytes;
  out->calls = root->dev_calls;
  out->comm_create_ms = root->comm_create_ms;
  out->key_replication_ms = root->key_replication_ms;
  root->scatter_bytes = root->gather_bytes = root->dev_calls = 0;
  return TFHE_HIP_OK;
}
#undef POOL_ENTER
#undef POOL_FIRST
