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
#include "combine_queue.hpp"

struct CombReq : combq::Node {
  KeyState *key = nullptr;
  int cls = 0;                    // CombClass
  int gate = TFHE_HIP_COPY;       // CB_GATES: the call's gate when `codes` is NULL
  const uint8_t *codes = nullptr;  // CB_GATES: per-ciphertext gates
  int keyswitch = 1;              // CB_GATES: 0 = bootstrap_without_key_switch
  const uint32_t *a = nullptr, *b = nullptr, *c = nullptr;
  const uint32_t *testvec = nullptr;  // CB_GATES: NULL = the key's own
  int per_ct = 0;
  uint32_t *out = nullptr;
  // CB_GATES with gate == COPY: bootstrap ca * a + cb * b, [n] += cconst (tfhe_hip_batch_lincomb_bootstrap); the
  // combination is formed on the device, in the packed rows, ahead of the merged launch
  bool lin = false;
  uint32_t lin_ca = 1, lin_cb = 0, lin_cconst = 0;
  int rc = TFHE_HIP_OK;
  std::string err;
};

// The queue (combine_queue.hpp: arrival list, lane bits, the futex word, lingering, statistics) plus what is HIP's:
// the lanes' private contexts and the bound on the calls that are merged.
struct Combiner : combq::Queue {
  static constexpr size_t kBatchCap = 4096;  // ciphertexts per merged launch (bounds the pinned arenas)
  tfhe_hip_ctx *lane_ctx[kLanes] = {};  // created by the lane's first leader
  std::atomic<size_t> max_count{0};     // calls of up to this many ciphertexts are merged; 0 = front end off
  std::atomic<bool> profiling{false};   // what lanes created later start with
  bool zero_copy_in = false;            // gate groups: the kernels read the pinned arena in place instead of a copy of it
  bool lane_high_priority = false;      // the lanes' streams are created at the device's highest stream priority
};

namespace {

bool g_comb_force_heap = false;  // (experiment builds, TFHE_HIP_COMBINE_HEAP: take the no-pinned-memory path of the lanes' arenas)

enum CombClass { CB_GATES = 0, CB_MUX = 1, CB_MUX_NAIVE = 2, CB_ROTATE = 3 };  // CB_ROTATE: trgsw::blind_rotate, the output is the TRLWE

// a lane's staging pair for one operand: host arena (packed by the leader; pinned, or ordinary memory where pinned memory
// is not to be had -- the copies work from either, only slower) -> device buffer
int comb_arena(tfhe_hip_ctx *x, PinBuf &pin, DevBuf &dev, size_t bytes) {
  CHK(ensure(x, dev, bytes));
  if (bytes <= pin.cap) return TFHE_HIP_OK;
  if (!pin.heap && !g_comb_force_heap && ensure_pinned(x, pin, bytes) == TFHE_HIP_OK) return TFHE_HIP_OK;
  if (pin.p && pin.heap) free(pin.p);
  pin.p = nullptr;
  pin.cap = 0;
  const size_t want = bytes + bytes / 4;
  pin.p = aligned_alloc(4096, (want + 4095) & ~(size_t)4095);
  if (!pin.p) return fail(x, TFHE_HIP_ENOMEM, "merged-call arena: out of host memory");
  pin.cap = want;
  pin.heap = true;
  return TFHE_HIP_OK;
}

// One merged launch: requests of one key view and one operation class, in queue order.  x's device is current.
int comb_run_group(tfhe_hip_ctx *x, KeyState *key, const std::vector<CombReq *> &g, double (&phase_us)[3], bool zero_copy_in) {
  KeyBind kb(x, key);
  const auto t_begin = std::chrono::steady_clock::now();
  const CombReq &r0 = *g[0];
  const size_t w = (size_t)x->P.n + 1, wb = w * 4;
  size_t m = 0;
  for (const CombReq *r : g) m += r->count;
  hipStream_t s = x->stream;
  const bool mux = r0.cls == CB_MUX || r0.cls == CB_MUX_NAIVE;
  const bool rotate = r0.cls == CB_ROTATE;  // (rows in: [n+1]; rows out: [2][N])
  const size_t ow = rotate ? (size_t)2 * kN : w, owb = ow * 4;
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
  bool any_lin = false;  // requests whose rows are a linear combination still to be formed (k_tlwe_lincomb, on the device)
  if (!mux)
    for (const CombReq *r : g)
      if (r->lin) {
        any_lin = true;
        if (r->lin_cb) need_b = true;
      }
  // pack: every request's rows behind one another
  CHK(comb_arena(x, x->p_a, x->h_a, m * wb));
  if (need_b) CHK(comb_arena(x, x->p_b, x->h_b, m * wb));
  if (need_c) CHK(comb_arena(x, x->p_c, x->h_c, m * wb));
  if (has_tv) CHK(comb_arena(x, x->p_tv, x->h_tv, m * (size_t)2 * kN * 4));
  if (!mux && !uniform) CHK(comb_arena(x, x->p_idx, x->h_idx, m));
  CHK(comb_arena(x, x->p_out, x->h_out, m * owb));
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
  const auto t_packed = std::chrono::steady_clock::now();
  // gate groups read each operand row once, in the blind rotation's prologue: with zero_copy_in the kernel takes the
  // pinned arena as it is (no copy to wait for ahead of the launch); mux reads its operands in three launches: copied
  const bool zc = zero_copy_in && !mux && !any_lin && !x->p_a.heap && !x->p_b.heap && !x->p_tv.heap;
  const uint32_t *da = (const uint32_t *)x->h_a.p, *db = need_b ? (const uint32_t *)x->h_b.p : nullptr;
  const uint32_t *dtv0 = has_tv ? (const uint32_t *)x->h_tv.p : nullptr;
  if (zc) {
    da = pinned_view((const uint32_t *)x->p_a.p, m * wb);
    if (need_b) db = pinned_view((const uint32_t *)x->p_b.p, m * wb);
    if (has_tv) dtv0 = pinned_view((const uint32_t *)x->p_tv.p, m * (size_t)2 * kN * 4);
    if (!da || (need_b && !db) || (has_tv && !dtv0)) return fail(x, TFHE_HIP_EHIP, "merged-call arena is not device-addressable");
  } else {
    HIPCHK(x, hipMemcpyAsync(x->h_a.p, x->p_a.p, m * wb, hipMemcpyHostToDevice, s));
    if (need_b) HIPCHK(x, hipMemcpyAsync(x->h_b.p, x->p_b.p, m * wb, hipMemcpyHostToDevice, s));
    if (need_c) HIPCHK(x, hipMemcpyAsync(x->h_c.p, x->p_c.p, m * wb, hipMemcpyHostToDevice, s));
    if (has_tv) HIPCHK(x, hipMemcpyAsync(x->h_tv.p, x->p_tv.p, m * (size_t)2 * kN * 4, hipMemcpyHostToDevice, s));
  }
  if (!mux && !uniform) HIPCHK(x, hipMemcpyAsync(x->h_idx.p, x->p_idx.p, m, hipMemcpyHostToDevice, s));
  if (any_lin) {  // one streaming launch per run of requests with the same coefficients, in place in the packed rows
    size_t at = 0;
    for (size_t i = 0; i < g.size();) {
      const CombReq *r = g[i];
      size_t rows = r->count, j = i + 1;
      if (r->lin) {
        while (j < g.size() && g[j]->lin && g[j]->lin_ca == r->lin_ca && g[j]->lin_cb == r->lin_cb && g[j]->lin_cconst == r->lin_cconst) rows += g[j++]->count;
        const size_t total = rows * w;
        uint32_t *pa = (uint32_t *)x->h_a.p + at * w;
        hipLaunchKernelGGL(k_tlwe_lincomb, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, r->lin_ca, pa, r->lin_cb,
                           r->lin_cb ? (const uint32_t *)x->h_b.p + at * w : nullptr, r->lin_cconst, pa, (uint32_t)w, total);
        HIPCHK(x, hipGetLastError());
      }
      at += rows;
      i = j;
    }
  }
  uint32_t *dout = (uint32_t *)x->h_out.p;
  if (mux) {
    CHK(mux_dev(x, r0.cls == CB_MUX_NAIVE, da, db, (const uint32_t *)x->h_c.p, dout, m, s));
  } else {
    GatePrep gp{1u, need_b ? 1u : 0u, 0u};  // mixed: placeholders, the kernel reads the codes (cb != 0 keeps in_b attached)
    if (uniform) gate_prep(gate0, gp);
    const uint8_t *dcodes = uniform ? nullptr : (const uint8_t *)x->h_idx.p;
    const uint32_t *dtv = dtv0;
    if (rotate) {
      CHK(launch_blind_rotate(x, s, da, nullptr, gp, dtv, 1, m, dout, nullptr, nullptr, nullptr));
    } else if (r0.keyswitch) {
      CHK(claim_scratch(x, s));
      CHK(ensure(x, x->lv1, lv1_rows(m) * (size_t)(kN + 1) * 4));
      CHK(launch_blind_rotate(x, s, da, db, gp, dtv, 1, m, nullptr, (uint32_t *)x->lv1.p, nullptr, dcodes));
      CHK(launch_key_switch(x, s, (const uint32_t *)x->lv1.p, dout, m));
    } else {
      CHK(launch_blind_rotate(x, s, da, db, gp, dtv, 1, m, nullptr, nullptr, dout, dcodes));
    }
  }
  HIPCHK(x, hipMemcpyAsync(x->p_out.p, x->h_out.p, m * owb, hipMemcpyDeviceToHost, s));
  HIPCHK(x, hipStreamSynchronize(s));
  const auto t_synced = std::chrono::steady_clock::now();
  {
    size_t at = 0;
    for (CombReq *r : g) {
      memcpy(r->out, (const uint32_t *)x->p_out.p + at * ow, r->count * owb);
      at += r->count;
    }
  }
  const auto t_end = std::chrono::steady_clock::now();
  phase_us[0] += std::chrono::duration<double, std::micro>(t_packed - t_begin).count();
  phase_us[1] += std::chrono::duration<double, std::micro>(t_synced - t_packed).count();
  phase_us[2] += std::chrono::duration<double, std::micro>(t_end - t_synced).count();
  return TFHE_HIP_OK;
}

// the lane's private context: the base's parameters and dispatch, its own stream / scratch / staging
int comb_make_lane(tfhe_hip_ctx *base, Combiner &C, int li, std::string &why) {
  tfhe_hip_ctx *x = nullptr;
  const int rc = tfhe_hip_ctx_create(&base->P, base->device, &x);
  if (rc != TFHE_HIP_OK) {
    why = std::string("merged-call lane: ") + g_create_error;
    return rc;
  }
  x->is_lane = true;
  delete x->comb;  // (a lane has no front end of its own)
  x->comb = nullptr;
  if (C.lane_high_priority) {
    // Small calls are the latency-sensitive ones: their launches go to a stream of the highest priority the device has,
    // so that the command processor dispatches their few workgroups ahead of a bulk launch's thousands when both are
    // pending (a 65,536-ciphertext batch on the context's own stream holds every CU for 330 ms).
    int least = 0, greatest = 0;
    hipStream_t hs = nullptr;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
        hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, greatest) == hipSuccess) {
      (void)hipStreamDestroy(x->stream);
      x->stream = hs;
    } else {
      (void)hipGetLastError();
    }
  }
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
  // arenas for a full round of the default bound up front (a, b, out: < 1 MB each): the first calls of a team do not
  // pay for pinned allocations, and a growing team does not re-allocate them
  {
    const size_t rows = C.max_count.load(std::memory_order_relaxed) ? C.max_count.load(std::memory_order_relaxed) : 1;
    const size_t bytes = rows * ((size_t)x->P.n + 1) * 4;
    (void)comb_arena(x, x->p_a, x->h_a, bytes);
    (void)comb_arena(x, x->p_b, x->h_b, bytes);
    (void)comb_arena(x, x->p_out, x->h_out, bytes);
    (void)ensure(x, x->lv1, lv1_rows(rows) * (size_t)(kN + 1) * 4);
  }
  C.lane_ctx[li] = x;
  return TFHE_HIP_OK;
}

// What the leader of lane `li` does with the requests it took: groups of (key view, class, key switch or not, own test
// vector or not), each in arrival order and cut at kBatchCap ciphertexts, one merged launch per group.
combq::Round comb_run_round(tfhe_hip_ctx *base, Combiner &C, int li, std::vector<combq::Node *> &nodes) {
  combq::Round rd;
  std::vector<CombReq *> all;
  all.reserve(nodes.size());
  for (combq::Node *n : nodes) all.push_back(static_cast<CombReq *>(n));
  DeviceGuard dg(base->device);
  std::string why;
  int rc = dg.err == hipSuccess ? TFHE_HIP_OK : TFHE_HIP_EHIP;
  if (rc != TFHE_HIP_OK) why = std::string("hipSetDevice: ") + hipGetErrorString(dg.err);
  if (rc == TFHE_HIP_OK && !C.lane_ctx[li]) rc = comb_make_lane(base, C, li, why);
  if (rc != TFHE_HIP_OK) {
    for (CombReq *r : all) {
      r->rc = rc;
      r->err = why;
    }
    return rd;
  }
  tfhe_hip_ctx *x = C.lane_ctx[li];
  double phase_us[3] = {0, 0, 0};
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
    const int grc = comb_run_group(x, h.key, g, phase_us, C.zero_copy_in);
    ++rd.launches;
    if (grc != TFHE_HIP_OK) {
      const std::string text = err_text(x->id);
      for (CombReq *r : g) {
        r->rc = grc;
        r->err = text;
      }
    }
  }
  rd.pack_us = phase_us[0];
  rd.gpu_us = phase_us[1];
  rd.unpack_us = phase_us[2];
  return rd;
}

// queue the request, lead when the lane is free, return when the request has been served
int comb_submit(tfhe_hip_ctx *base, CombReq &r) {
  Combiner &C = *base->comb;
  C.submit(r, [&](std::vector<combq::Node *> &nodes, int li) { return comb_run_round(base, C, li, nodes); });
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
  if (base->comb) base->comb->quiesce();
}

// every lane idle, and held idle while f runs (f must not submit)
template <class F>
void comb_with_idle_lanes(tfhe_hip_ctx *base, F &&f) {
  Combiner *C = base->comb;
  if (!C) return;
  C->with_idle_lanes([&] { f(*C); });
}

// A key has just become current on `base` (its device is current): have the first lane ready, so that a caller's first
// small call does not pay for a context and its arenas (6 ms measured; a failure here is not one -- the first leader
// tries again and reports it).
void comb_prepare(tfhe_hip_ctx *base) {
  Combiner *C = base->comb;
  if (!C || base->is_lane || C->max_count.load(std::memory_order_relaxed) == 0) return;
  C->with_idle_lanes([&] {
    std::string why;
    if (!C->lane_ctx[0]) (void)comb_make_lane(base, *C, 0, why);
  });
}

void comb_destroy(tfhe_hip_ctx *base) {
  Combiner *C = base->comb;
  if (!C) return;
  C->quiesce();
  for (int i = 0; i < Combiner::kLanes; ++i)
    if (C->lane_ctx[i]) tfhe_hip_ctx_destroy(C->lane_ctx[i]);
  base->comb = nullptr;
  delete C;
}

}  // namespace
