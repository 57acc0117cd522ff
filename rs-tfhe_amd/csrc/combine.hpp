// combine.hpp -- the combining front end of the host-pointer entry points (included by tfhe_hip.hip).
//
// The reference's strategy trait is `Send + Sync` (src/bootstrap/mod.rs:23): a Rayon team may call
// Bootstrap::bootstrap / Gates::nand on ONE strategy from every worker at once (src/parallel/rayon_impl.rs:40-47 is the
// same shape inside the crate).  On the CPU each of those calls owns a core.  Here a one-ciphertext call occupies one
// workgroup on one of 256 CUs for 2.2 ms, so calls that ran one after the other would leave the chip 99.6 % idle
// however many threads were waiting.  The front end merges them:
//
//   * a small call (count <= max_count) does not take the context's mutex; it pushes itself onto a lock-free arrival
//     list and, if a LANE is free, becomes that lane's leader: it takes everything that has arrived -- in particular
//     everything that arrived while the previous launch ran (natural batching) --, packs the operands into the lane's
//     pinned arena, issues ONE launch per (key view, operation class) -- per-ciphertext gate codes and per-ciphertext
//     test vectors already exist in the kernels (tfhe_hip_batch_gates_mixed, per_ct test vectors) --, hands every
//     caller its rows of the result, marks the requests done and wakes the sleepers; everybody else sleeps on one
//     futex word until its request is done or a lane is free;
//   * a lane is a private sibling context (its own stream, scratch and staging; the caller's key is bound to it per
//     launch), so a merged launch runs beside a large call on the context's own stream;
//   * a leader that follows a merged launch closely waits a bounded moment for the callers of that launch to come back
//     (they were all released at the same instant; without this the first one back would launch alone and the rest
//     would wait a whole launch behind it).  A lone caller never waits: it leads at once and sees the latency of a
//     plain one-ciphertext call.
//
// Lanes.  The machinery takes up to four lanes; ONE is used.  Two lanes let a second merged launch start while the
// first is in flight, which only pays while both are small, and only if their streams land on different hardware
// queues: measured on MI355X (profiles/exp/logs/r6b_front_end.log), 64 threads get 25-26 k gates/s on one lane and the
// same on two lanes when the launches overlap -- but when the runtime maps the two streams to one hardware queue (it
// hands out four per process and shares beyond that; seen whenever the second lane was created later than the first)
// the launches run one after the other and every call takes two launch times: 14-15 k gates/s.  One lane cannot lose
// that lottery.  (-DTFHE_EXPERIMENT builds read TFHE_HIP_COMBINE_LANES.)
//
// Same kernels, same per-element operations in the same order as the unmerged call: the results are the same bits
// (tests/test_gpu_combine.py holds every word to the CPU checker).  Errors stay per calling thread: argument errors
// are found by the caller before it queues, and a failure of the merged launch is copied into every request it carried
// and filed under the calling thread's own error text.
#pragma once
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <chrono>
#include <climits>
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
  // QUEUED until a leader takes it, DONE when its rows are in `out` (or rc / err say why not).  The owner's frame may
  // go away the moment it reads DONE: a leader does not touch a request after storing that.
  enum : uint32_t { QUEUED = 0, TAKEN = 1, DONE = 2 };
  std::atomic<uint32_t> state{QUEUED};
  CombReq *next = nullptr;  // the arrival list (newest first)
};

// No mutex on the callers' path: with hundreds of threads released at the same instant by one merged launch, a
// condition variable's mutex is re-acquired by every one of them in turn (measured: a 256-thread team spent as long in
// that queue as in the launch).  Arrivals push themselves onto a lock-free list, lanes are bits of one word, and
// everybody who has to wait sleeps on ONE futex word (`epoch`) that is bumped whenever something completes or a lane
// becomes free; woken threads look at their own request's state and go back to sleep if it is not their turn.
struct Combiner {
  static constexpr int kLanes = 4;  // most lanes a front end can have
  static constexpr size_t kBatchCap = 4096;  // ciphertexts per merged launch (bounds the pinned arenas)
  int nlanes = 1;                   // lanes in use (see the note on lanes below)
  struct Lane {
    tfhe_hip_ctx *x = nullptr;  // created by its first leader
    std::atomic<uint64_t> gen{0};  // leader rounds completed on this lane
  };
  Lane lane[kLanes];
  std::atomic<CombReq *> arrivals_head{nullptr};
  std::atomic<uint32_t> lanes_busy{0};  // bit i: lane i has a leader
  std::atomic<uint32_t> epoch{0};       // the futex word
  std::atomic<uint32_t> sleepers{0};    // threads in (or about to enter) futex_wait: nobody to wake, no system call
  std::atomic<uint32_t> collecting{0};  // leaders between taking a lane and taking the arrival list: arrivals do not lead, they will be taken
  std::atomic<size_t> max_count{0};  // calls of up to this many ciphertexts are merged; 0 = front end off
  std::atomic<size_t> pending{0};    // ciphertexts queued or in flight (a pool picks its least loaded member by it)
  std::atomic<uint64_t> arrivals{0};  // requests ever queued (the lingering leader watches it grow)
  std::atomic<uint64_t> taken{0};     // requests ever taken by a leader
  std::atomic<bool> profiling{false};  // what lanes created later start with
  // the last leader round that completed: how many requests it carried, and when (steady_clock ns)
  std::atomic<size_t> last_reqs{0};
  std::atomic<int64_t> last_done_ns{0};
  // lingering (see the header comment): only within linger_window of a round that carried several requests; ends when
  // as many requests are waiting as that round carried (`want`), when nobody has arrived for linger_quiet + want / 4
  // microseconds, or after linger_max + want microseconds (a team of hundreds of threads takes that long to come back
  // through the scheduler; a launch that leaves without most of them makes them wait a whole launch: measured at 256
  // threads, 41 k gates/s with a 10 us quiet gap, 65 k with 100 us -- profiles/exp/logs/r6b_front_end.log)
  long linger_window_us = 1000, linger_quiet_us = 25, linger_max_us = 250;
  // statistics (tfhe_hip_get_combine_stats); leaders only
  std::mutex st_mu;
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
  x->profiling = C.profiling.load(std::memory_order_relaxed);
  L.x = x;
  return TFHE_HIP_OK;
}

// ---- waiting and waking: one futex word ---------------------------------------------------------------------------
inline void comb_wait(Combiner &C, uint32_t seen) {
  C.sleepers.fetch_add(1, std::memory_order_seq_cst);
  if (C.epoch.load(std::memory_order_seq_cst) == seen)
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&C.epoch), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
  C.sleepers.fetch_sub(1, std::memory_order_seq_cst);
}
inline void comb_wake_all(Combiner &C) {
  C.epoch.fetch_add(1, std::memory_order_seq_cst);
  if (C.sleepers.load(std::memory_order_seq_cst))
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&C.epoch), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
inline int comb_try_lane(Combiner &C) {  // a free lane, now this thread's; -1: none
  uint32_t busy = C.lanes_busy.load(std::memory_order_relaxed);
  for (;;) {
    int li = -1;
    for (int i = 0; i < C.nlanes && li < 0; ++i)
      if (!(busy & (1u << i))) li = i;
    if (li < 0) return -1;
    if (C.lanes_busy.compare_exchange_weak(busy, busy | (1u << li), std::memory_order_acquire, std::memory_order_relaxed)) return li;
  }
}
inline void comb_release_lane(Combiner &C, int li) {
  C.lane[li].gen.fetch_add(1, std::memory_order_release);
  C.lanes_busy.fetch_and(~(1u << li), std::memory_order_release);
  comb_wake_all(C);
}
inline int64_t comb_now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The calling thread holds lane `li`: it takes whatever has arrived, runs it, marks it done and releases the lane.
void comb_lead(tfhe_hip_ctx *base, Combiner &C, int li) {
  Combiner::Lane &L = C.lane[li];
  C.collecting.fetch_add(1, std::memory_order_seq_cst);
  // the callers of the round that has just completed are on their way back: give them a bounded moment
  {
    const size_t want = C.last_reqs.load(std::memory_order_relaxed);
    const uint64_t taken0 = C.taken.load(std::memory_order_relaxed);
    auto waiting = [&] { return (size_t)(C.arrivals.load(std::memory_order_relaxed) - taken0); };
    const int64_t t0 = comb_now_ns();
    if (want > 1 && waiting() < want && t0 - C.last_done_ns.load(std::memory_order_relaxed) < C.linger_window_us * 1000) {
      int64_t last_growth = t0, now = t0;
      size_t seen = waiting();
      for (;;) {
        cpu_relax();
        now = comb_now_ns();
        const size_t cur = waiting();
        if (cur != seen) {
          seen = cur;
          last_growth = now;
        }
        if (cur >= want || now - last_growth > (C.linger_quiet_us * 4 + (long)want) * 250 || now - t0 > (C.linger_max_us + (long)want) * 1000) break;
      }
      std::lock_guard<std::mutex> lk(C.st_mu);
      ++C.st_lingers;
      C.st_linger_us += (double)(now - t0) * 1e-3;
    }
  }
  // take the arrival list (newest first) and put it in arrival order
  std::vector<CombReq *> all;
  for (CombReq *r = C.arrivals_head.exchange(nullptr, std::memory_order_acquire); r;) {
    CombReq *nx = r->next;  // (read before anything can complete the request)
    all.push_back(r);
    r = nx;
  }
  std::reverse(all.begin(), all.end());
  C.taken.fetch_add(all.size(), std::memory_order_relaxed);
  for (CombReq *r : all) r->state.store(CombReq::TAKEN, std::memory_order_seq_cst);
  C.collecting.fetch_sub(1, std::memory_order_seq_cst);
  // a request that arrived after the list was taken may have seen `collecting` and gone to sleep expecting to be taken
  if (C.arrivals_head.load(std::memory_order_seq_cst)) comb_wake_all(C);
  size_t launches = 0, total = 0;
  if (!all.empty()) {
    DeviceGuard dg(base->device);
    std::string why;
    int rc = dg.err == hipSuccess ? TFHE_HIP_OK : TFHE_HIP_EHIP;
    if (rc != TFHE_HIP_OK) why = std::string("hipSetDevice: ") + hipGetErrorString(dg.err);
    if (rc == TFHE_HIP_OK && !L.x) rc = comb_make_lane(base, C, L, why);
    if (rc != TFHE_HIP_OK) {
      for (CombReq *r : all) {
        r->rc = rc;
        r->err = why;
      }
    } else {
      // groups: (key view, class, key switch or not, own test vector or not), each in arrival order and cut at
      // kBatchCap ciphertexts
      std::vector<bool> placed(all.size(), false);
      for (size_t i = 0; i < all.size(); ++i) {
        if (placed[i]) continue;
        const CombReq &h = *all[i];
        std::vector<CombReq *> g;
        size_t m = 0;
        for (size_t j = i; j < all.size(); ++j) {
          const CombReq &r = *all[j];
          if (placed[j] || r.key != h.key || r.cls != h.cls || r.keyswitch != h.keyswitch || (r.testvec != nullptr) != (h.testvec != nullptr)) continue;
          if (!g.empty() && m + r.count > Combiner::kBatchCap) break;  // the rest of this group: a launch of its own
          placed[j] = true;
          g.push_back(all[j]);
          m += r.count;
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
    for (CombReq *r : all) total += r->count;
    C.pending.fetch_sub(total, std::memory_order_relaxed);
    C.last_reqs.store(all.size(), std::memory_order_relaxed);
    C.last_done_ns.store(comb_now_ns(), std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(C.st_mu);
      C.st_launches += launches;
      C.st_requests += all.size();
      C.st_cts += total;
      if (all.size() > C.st_max_requests) C.st_max_requests = all.size();
    }
    for (CombReq *r : all) r->state.store(CombReq::DONE, std::memory_order_release);  // (r may be gone after this)
  }
  comb_release_lane(C, li);
}

// queue the request, lead when a lane is free, return when the request has been served
int comb_submit(tfhe_hip_ctx *base, CombReq &r) {
  Combiner &C = *base->comb;
  C.pending.fetch_add(r.count, std::memory_order_relaxed);
  {
    CombReq *h = C.arrivals_head.load(std::memory_order_relaxed);
    do r.next = h;
    while (!C.arrivals_head.compare_exchange_weak(h, &r, std::memory_order_release, std::memory_order_relaxed));
  }
  C.arrivals.fetch_add(1, std::memory_order_relaxed);
  for (;;) {
    const uint32_t e = C.epoch.load(std::memory_order_seq_cst);  // (before the checks: a wake-up in between is not lost)
    const uint32_t st = r.state.load(std::memory_order_acquire);
    if (st == CombReq::DONE) break;
    if (st == CombReq::QUEUED && C.collecting.load(std::memory_order_seq_cst) == 0) {
      const int li = comb_try_lane(C);
      if (li >= 0) {
        if (r.state.load(std::memory_order_seq_cst) == CombReq::QUEUED) comb_lead(base, C, li);  // takes this request too
        else comb_release_lane(C, li);  // another leader took it meanwhile
        continue;
      }
    }
    comb_wait(C, e);
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
  for (int i = 0; i < Combiner::kLanes; ++i) {
    const uint64_t g = C->lane[i].gen.load(std::memory_order_acquire);
    for (;;) {
      const uint32_t e = C->epoch.load(std::memory_order_seq_cst);
      if (!(C->lanes_busy.load(std::memory_order_acquire) & (1u << i)) || C->lane[i].gen.load(std::memory_order_acquire) != g) break;
      comb_wait(*C, e);
    }
  }
}

// every lane idle, and held idle while f runs (f must not submit)
template <class F>
void comb_with_idle_lanes(tfhe_hip_ctx *base, F &&f) {
  Combiner *C = base->comb;
  if (!C) return;
  const uint32_t all = (1u << Combiner::kLanes) - 1;
  for (;;) {  // take every lane bit at once
    const uint32_t e = C->epoch.load(std::memory_order_seq_cst);
    uint32_t none = 0;
    if (C->lanes_busy.compare_exchange_strong(none, all, std::memory_order_acquire, std::memory_order_relaxed)) break;
    comb_wait(*C, e);
  }
  f(*C);
  C->lanes_busy.store(0, std::memory_order_release);
  comb_wake_all(*C);
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
