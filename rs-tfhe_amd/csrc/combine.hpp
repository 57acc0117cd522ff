// combine.hpp -- the combining front end of the host-pointer entry points (included by tfhe_hip.hip).
//
// The reference's strategy trait is `Send + Sync` (src/bootstrap/mod.rs:23): a Rayon team may call
// Bootstrap::bootstrap / Gates::nand on ONE strategy from every worker at once (src/parallel/rayon_impl.rs:40-47 is the
// same shape inside the crate).  On the CPU each of those calls owns a core.  Here a one-ciphertext call occupies one
// workgroup on one of 256 CUs for 2.2 ms, so calls that ran one after the other would leave the chip 99.6 % idle
// however many threads were waiting.  The front end merges them:
//
//   * a small call (count <= max_count) does not take the context's mutex; it joins a queue;
//   * the call at the head of the queue, when a LANE is free, becomes the leader of that lane: it takes everything
//     that queued up while the previous launches ran (natural batching), packs the operands into the lane's pinned
//     arena, issues ONE launch per (key view, operation class) -- per-ciphertext gate codes and per-ciphertext test
//     vectors already exist in the kernels (tfhe_hip_batch_gates_mixed, per_ct test vectors) --, hands every caller
//     its slice of the result and wakes them;
//   * a lane is a private sibling context (its own stream, scratch and staging; the caller's key is bound to it per
//     launch), so two merged launches can be in flight beside each other and beside a large call on the context's
//     own stream;
//   * a leader that follows a merged launch closely waits a bounded moment for the callers of that launch to come back
//     (they were all released at the same instant; without this the first one back would launch alone and the rest
//     would wait a whole launch behind it).  A lone caller never waits: it leads at once and sees the latency of a
//     plain one-ciphertext call.
//
// Same kernels, same per-element operations in the same order as the unmerged call: the results are the same bits
// (tests/test_gpu_combine.py holds every word to the CPU checker).  Errors stay per calling thread: argument errors
// are found by the caller before it queues, and a failure of the merged launch is copied into every request it carried
// and filed under the calling thread's own error text.
#pragma once
#include <chrono>
#include <deque>
#include <thread>

struct CombReq {
  KeyState *key = nullptr;
  int cls = 0;                    // CombClass
  int gate = TFHE_HIP_COPY;       // CB_GATES: the call's gate when `codes` is NULL
  const uint8_t *codes = nullptr;  // CB_GATES: per-ciphertext gates
  int keyswitch = 1;              // CB_GATES: 0 = bootstrap_without_key_switch
  const uint32_t *a = nullptr, *b = nullptr, *c = nullptr;
  const uint32_t *testvec = nullptr;  // CB_GATES: NULL = the key's own
  int per_ct = 0;
  uint32_t *out = nullptr;
  size_t count = 0;
  int rc = TFHE_HIP_OK;
  std::string err;
  bool done = false;
};

struct Combiner {
  static constexpr int kLanes = 4;  // most lanes a front end can have
  int nlanes = 2;                   // lanes in use
  static constexpr size_t kBatchCap = 4096;  // ciphertexts per merged launch (bounds the pinned arenas)
  struct Lane {
    tfhe_hip_ctx *x = nullptr;  // created by its first leader
    bool busy = false;
    uint64_t gen = 0;  // merged launches completed on this lane
  };
  std::mutex mu;
  std::condition_variable cv;
  std::deque<CombReq *> q;
  Lane lane[kLanes];
  std::atomic<size_t> max_count{0};  // calls of up to this many ciphertexts are merged; 0 = front end off
  std::atomic<size_t> pending{0};    // ciphertexts queued or in flight (a pool picks its least loaded member by it)
  std::atomic<uint64_t> arrivals{0};  // requests ever queued (the lingering leader watches it grow)
  bool profiling = false;             // what lanes created later start with
  // the last merged launch that completed: how many requests it carried, and when
  size_t last_reqs = 0;
  std::chrono::steady_clock::time_point last_done{};
  // lingering (see the header comment): only within kLingerWindow of a merged launch that carried several requests;
  // ends when as many requests are queued as that launch carried, when nobody has arrived for kLingerQuiet, or after
  // kLingerMax
  long linger_window_us = 1000, linger_quiet_us = 25, linger_max_us = 250;
  // statistics (tfhe_hip_get_combine_stats)
  uint64_t st_launches = 0, st_requests = 0, st_cts = 0, st_max_requests = 0, st_lingers = 0;
  double st_linger_us = 0;
};

namespace {

enum CombClass { CB_GATES = 0, CB_MUX = 1, CB_MUX_NAIVE = 2 };

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

// a lane's staging pair for one operand: pinned arena (packed by the leader) -> device buffer
int comb_arena(tfhe_hip_ctx *x, PinBuf &pin, DevBuf &dev, size_t bytes) {
  CHK(ensure(x, dev, bytes));
  if (ensure_pinned(x, pin, bytes) != TFHE_HIP_OK) return fail(x, TFHE_HIP_ENOMEM, "hipHostMalloc: merged-call arena");
  return TFHE_HIP_OK;
}

// One merged launch: requests of one key view and one operation class, in queue order.  x's device is current.
int comb_run_group(tfhe_hip_ctx *x, KeyState *key, const std::vector<CombReq *> &g) {
  KeyBind kb(x, key);
  const CombReq &r0 = *g[0];
  const size_t w = (size_t)x->P.n + 1, wb = w * 4;
  size_t m = 0;
  for (const CombReq *r : g) m += r->count;
  hipStream_t s = x->stream;
  const bool mux = r0.cls != CB_GATES;
  // which operands the launch reads; one gate throughout (`uniform`) needs no per-ciphertext codes
  const int gate0 = r0.codes ? (int)r0.codes[0] : r0.gate;
  bool need_b = mux, need_c = mux, uniform = !mux;
  const bool has_tv = !mux && r0.testvec != nullptr;
  if (!mux)
    for (const CombReq *r : g)
      for (size_t i = 0; i < (r->codes ? r->count : 1); ++i) {
        const int code = r->codes ? (int)r->codes[i] : r->gate;
        if (code != gate0) uniform = false;
        GatePrep q;
        if (gate_prep(code, q) && q.cb) need_b = true;
      }
  // pack: every request's rows behind one another
  CHK(comb_arena(x, x->p_a, x->h_a, m * wb));
  if (need_b) CHK(comb_arena(x, x->p_b, x->h_b, m * wb));
  if (need_c) CHK(comb_arena(x, x->p_c, x->h_c, m * wb));
  if (has_tv) CHK(comb_arena(x, x->p_tv, x->h_tv, m * (size_t)2 * kN * 4));
  if (!mux && !uniform) CHK(comb_arena(x, x->p_idx, x->h_idx, m));
  CHK(comb_arena(x, x->p_out, x->h_out, m * wb));
  {
    size_t at = 0;
    for (const CombReq *r : g) {
      memcpy((uint32_t *)x->p_a.p + at * w, r->a, r->count * wb);
      if (need_b && r->b) memcpy((uint32_t *)x->p_b.p + at * w, r->b, r->count * wb);
      if (need_c) memcpy((uint32_t *)x->p_c.p + at * w, r->c, r->count * wb);
      if (has_tv)
        for (size_t i = 0; i < r->count; ++i)
          memcpy((uint32_t *)x->p_tv.p + (at + i) * (size_t)2 * kN, r->testvec + (r->per_ct ? i * (size_t)2 * kN : 0), (size_t)2 * kN * 4);
      if (!mux && !uniform) {
        if (r->codes) memcpy((uint8_t *)x->p_idx.p + at, r->codes, r->count);
        else memset((uint8_t *)x->p_idx.p + at, r->gate, r->count);
      }
      at += r->count;
    }
  }
  HIPCHK(x, hipMemcpyAsync(x->h_a.p, x->p_a.p, m * wb, hipMemcpyHostToDevice, s));
  if (need_b) HIPCHK(x, hipMemcpyAsync(x->h_b.p, x->p_b.p, m * wb, hipMemcpyHostToDevice, s));
  if (need_c) HIPCHK(x, hipMemcpyAsync(x->h_c.p, x->p_c.p, m * wb, hipMemcpyHostToDevice, s));
  if (has_tv) HIPCHK(x, hipMemcpyAsync(x->h_tv.p, x->p_tv.p, m * (size_t)2 * kN * 4, hipMemcpyHostToDevice, s));
  if (!mux && !uniform) HIPCHK(x, hipMemcpyAsync(x->h_idx.p, x->p_idx.p, m, hipMemcpyHostToDevice, s));
  const uint32_t *da = (const uint32_t *)x->h_a.p, *db = need_b ? (const uint32_t *)x->h_b.p : nullptr;
  uint32_t *dout = (uint32_t *)x->h_out.p;
  if (mux) {
    CHK(mux_dev(x, r0.cls == CB_MUX_NAIVE, da, db, (const uint32_t *)x->h_c.p, dout, m, s));
  } else {
    GatePrep gp{1u, need_b ? 1u : 0u, 0u};  // mixed: placeholders, the kernel reads the codes (cb != 0 keeps in_b attached)
    if (uniform) gate_prep(gate0, gp);
    const uint8_t *dcodes = uniform ? nullptr : (const uint8_t *)x->h_idx.p;
    const uint32_t *dtv = has_tv ? (const uint32_t *)x->h_tv.p : nullptr;
    if (r0.keyswitch) {
      CHK(claim_scratch(x, s));
      CHK(ensure(x, x->lv1, lv1_rows(m) * (size_t)(kN + 1) * 4));
      CHK(launch_blind_rotate(x, s, da, db, gp, dtv, 1, m, nullptr, (uint32_t *)x->lv1.p, nullptr, dcodes));
      CHK(launch_key_switch(x, s, (const uint32_t *)x->lv1.p, dout, m));
    } else {
      CHK(launch_blind_rotate(x, s, da, db, gp, dtv, 1, m, nullptr, nullptr, dout, dcodes));
    }
  }
  HIPCHK(x, hipMemcpyAsync(x->p_out.p, x->h_out.p, m * wb, hipMemcpyDeviceToHost, s));
  HIPCHK(x, hipStreamSynchronize(s));
  {
    size_t at = 0;
    for (CombReq *r : g) {
      memcpy(r->out, (const uint32_t *)x->p_out.p + at * w, r->count * wb);
      at += r->count;
    }
  }
  return TFHE_HIP_OK;
}

// the lane's private context: the base's parameters and dispatch, its own stream / scratch / staging
int comb_make_lane(tfhe_hip_ctx *base, Combiner &C, Combiner::Lane &L, std::string &why) {
  tfhe_hip_ctx *x = nullptr;
  const int rc = tfhe_hip_ctx_create(&base->P, base->device, &x);
  if (rc != TFHE_HIP_OK) {
    why = std::string("merged-call lane: ") + g_create_error;
    return rc;
  }
  x->is_lane = true;
  delete x->comb;  // (a lane has no front end of its own)
  x->comb = nullptr;
  x->br_force = base->br_force;
  x->ks_force = base->ks_force;
  x->wide_max = base->wide_max;
  x->pair_lo = base->pair_lo;
  x->pair_max = base->pair_max;
  x->ks_split_max = base->ks_split_max;
  x->ks_mfma_min = base->ks_mfma_min;
  x->ks_sl_chunk_min = base->ks_sl_chunk_min;
  x->ks_sliced_sets = base->ks_sliced_sets;
  x->ks_mfma_ksplit = base->ks_mfma_ksplit;
  x->ks_sl_kchunks = base->ks_sl_kchunks;
  x->br_chunk = base->br_chunk;
  x->exp_wide1 = base->exp_wide1;
  x->fast_round = base->fast_round;
  x->profiling = C.profiling;
  L.x = x;
  return TFHE_HIP_OK;
}

// The calling thread leads lane `li`: C.mu held on entry and on return, released while the launch runs.  `me` is at
// the head of the queue, so it is part of what is taken.
void comb_lead(tfhe_hip_ctx *base, Combiner &C, int li, std::unique_lock<std::mutex> &lk) {
  using clock = std::chrono::steady_clock;
  Combiner::Lane &L = C.lane[li];
  L.busy = true;
  // the callers of the merged launch that has just completed are on their way back: give them a bounded moment
  if (C.last_reqs > 1 && C.q.size() < C.last_reqs && clock::now() - C.last_done < std::chrono::microseconds(C.linger_window_us)) {
    const size_t want = C.last_reqs;
    const uint64_t arrived0 = C.arrivals.load(std::memory_order_relaxed);
    const size_t queued0 = C.q.size();
    lk.unlock();
    const auto t0 = clock::now();
    auto last_growth = t0;
    uint64_t seen = arrived0;
    for (;;) {
      cpu_relax();
      const auto now = clock::now();
      const uint64_t cur = C.arrivals.load(std::memory_order_relaxed);
      if (cur != seen) {
        seen = cur;
        last_growth = now;
      }
      if (queued0 + (size_t)(cur - arrived0) >= want) break;
      if (now - last_growth > std::chrono::microseconds(C.linger_quiet_us)) break;
      if (now - t0 > std::chrono::microseconds(C.linger_max_us)) break;
    }
    lk.lock();
    ++C.st_lingers;
    C.st_linger_us += std::chrono::duration<double, std::micro>(clock::now() - t0).count();
  }
  std::vector<CombReq *> batch;
  size_t total = 0;
  while (!C.q.empty() && (batch.empty() || total + C.q.front()->count <= Combiner::kBatchCap)) {
    batch.push_back(C.q.front());
    total += C.q.front()->count;
    C.q.pop_front();
  }
  lk.unlock();
  size_t launches = 0;
  {
    DeviceGuard dg(base->device);
    std::string why;
    int rc = dg.err == hipSuccess ? TFHE_HIP_OK : TFHE_HIP_EHIP;
    if (rc != TFHE_HIP_OK) why = std::string("hipSetDevice: ") + hipGetErrorString(dg.err);
    if (rc == TFHE_HIP_OK && !L.x) rc = comb_make_lane(base, C, L, why);
    if (rc != TFHE_HIP_OK) {
      for (CombReq *r : batch) {
        r->rc = rc;
        r->err = why;
      }
    } else {
      // groups: (key view, class, key switch or not, own test vector or not), each in queue order
      std::vector<bool> taken(batch.size(), false);
      for (size_t i = 0; i < batch.size(); ++i) {
        if (taken[i]) continue;
        const CombReq &h = *batch[i];
        std::vector<CombReq *> g;
        for (size_t j = i; j < batch.size(); ++j) {
          const CombReq &r = *batch[j];
          if (taken[j] || r.key != h.key || r.cls != h.cls || r.keyswitch != h.keyswitch || (r.testvec != nullptr) != (h.testvec != nullptr)) continue;
          taken[j] = true;
          g.push_back(batch[j]);
        }
        const int grc = comb_run_group(L.x, h.key, g);
        ++launches;
        if (grc != TFHE_HIP_OK) {
          const std::string text = err_text(L.x->id);
          for (CombReq *r : g) {
            r->rc = grc;
            r->err = text;
          }
        }
      }
    }
  }
  lk.lock();
  for (CombReq *r : batch) r->done = true;
  C.pending.fetch_sub(total, std::memory_order_relaxed);
  L.busy = false;
  ++L.gen;
  C.last_reqs = batch.size();
  C.last_done = clock::now();
  C.st_launches += launches;
  C.st_requests += batch.size();
  C.st_cts += total;
  if (batch.size() > C.st_max_requests) C.st_max_requests = batch.size();
  C.cv.notify_all();
}

// queue the request, lead if it is this thread's turn, return when the request has been served
int comb_submit(tfhe_hip_ctx *base, CombReq &r) {
  Combiner &C = *base->comb;
  {
    std::unique_lock<std::mutex> lk(C.mu);
    C.q.push_back(&r);
    C.pending.fetch_add(r.count, std::memory_order_relaxed);
    C.arrivals.fetch_add(1, std::memory_order_relaxed);
    while (!r.done) {
      int free_lane = -1;
      if (C.q.front() == &r)
        for (int i = 0; i < C.nlanes && free_lane < 0; ++i)
          if (!C.lane[i].busy) free_lane = i;
      if (free_lane >= 0) comb_lead(base, C, free_lane, lk);
      else C.cv.wait(lk);
    }
  }
  if (r.rc != TFHE_HIP_OK) err_slot(base->id) = r.err;
  return r.rc;
}

// does the front end take a call of `count` ciphertexts on this handle?
inline bool comb_takes(const tfhe_hip_ctx *ctx, size_t count) {
  const tfhe_hip_ctx *base = ctx->parent ? ctx->parent : ctx;
  return base->comb && count > 0 && count <= base->comb->max_count.load(std::memory_order_relaxed);
}

// Whatever the lanes had in flight when this is called has completed when it returns (a key is about to change or go:
// nothing may still read it).  Calls under the key that is changing are the caller's to keep away, as for any call.
void comb_quiesce(tfhe_hip_ctx *base) {
  Combiner *C = base->comb;
  if (!C) return;
  std::unique_lock<std::mutex> lk(C->mu);
  for (int i = 0; i < Combiner::kLanes; ++i) {
    const uint64_t g = C->lane[i].gen;
    C->cv.wait(lk, [&] { return !C->lane[i].busy || C->lane[i].gen != g; });
  }
}

// every lane idle, and held idle while f runs (f must not submit)
template <class F>
void comb_with_idle_lanes(tfhe_hip_ctx *base, F &&f) {
  Combiner *C = base->comb;
  if (!C) return;
  std::unique_lock<std::mutex> lk(C->mu);
  C->cv.wait(lk, [&] {
    for (int i = 0; i < Combiner::kLanes; ++i)
      if (C->lane[i].busy) return false;
    return true;
  });
  f(*C);
}

void comb_destroy(tfhe_hip_ctx *base) {
  Combiner *C = base->comb;
  if (!C) return;
  comb_quiesce(base);
  for (int i = 0; i < Combiner::kLanes; ++i)
    if (C->lane[i].x) tfhe_hip_ctx_destroy(C->lane[i].x);
  base->comb = nullptr;
  delete C;
}

}  // namespace
