// callers.cpp -- a team of host threads that call the C ABI the way a Rayon team calls one `Send + Sync` strategy
// (src/bootstrap/mod.rs:23-38; `pairs.par_iter().map(|(a, b)| gates.nand(a, b, ck))` in user code,
// src/parallel/rayon_impl.rs:40-47 inside the crate): every thread makes its own sequence of SMALL blocking
// host-pointer calls on one shared handle.  Load generator for tests/test_gpu_combine.py and bench.py's
// `gpu_concurrent_single_gate` figure (Python threads would measure the interpreter lock, not the library); it only
// uses what include/tfhe_hip.h exports and is not part of the product path.  The entry points arrive as a table of
// function pointers (the caller has the library open already: no second copy of it, no link-time dependency).
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tfhe_hip.h"

extern "C" {

typedef struct tfhe_callers_api {
  decltype(&tfhe_hip_batch_gate) batch_gate;
  decltype(&tfhe_hip_batch_bootstrap) batch_bootstrap;
  decltype(&tfhe_hip_batch_mux) batch_mux;
  decltype(&tfhe_hip_last_error) last_error;
  decltype(&tfhe_hip_pool_batch_gate) pool_batch_gate;
  decltype(&tfhe_hip_pool_batch_bootstrap) pool_batch_bootstrap;
  decltype(&tfhe_hip_pool_batch_mux) pool_batch_mux;
  decltype(&tfhe_hip_pool_last_error) pool_last_error;
} tfhe_callers_api;

// op: 0 = tfhe_hip_batch_gate (gates[i] = the tfhe_hip_gate of call i), 1 = tfhe_hip_batch_bootstrap with call i's own
// test vector testvecs[i] ([2][N]) and key switch, 2 = tfhe_hip_batch_mux (naive = gates[i] & 1).
// Call i of thread t handles ciphertexts [ (t * calls + i) * per_call, + per_call ) of a / b / c / out ([..][width]).
// handle: a tfhe_hip_ctx (or key view) when is_pool == 0, a tfhe_hip_pool otherwise.  All threads start together;
// *seconds is the wall time from the common start to the last thread's last return.  Returns 0, or the first failing
// call's code with its text (as read by the failing thread) in err[errlen].
int tfhe_callers_run(const tfhe_callers_api *api, void *handle, int is_pool, int op, const uint8_t *gates, const uint32_t *a, const uint32_t *b,
                     const uint32_t *c, const uint32_t *testvecs, uint32_t *out, size_t width, size_t per_call,
                     int threads, int calls, double *seconds, double *call_ms /* [threads * calls] or NULL */, char *err,
                     size_t errlen) {
  if (!api || !handle || !a || !out || threads < 1 || calls < 1 || per_call < 1) return TFHE_HIP_EINVAL;
  if ((op == 0 && !gates) || (op == 2 && (!b || !c))) return TFHE_HIP_EINVAL;
  std::atomic<int> ready{0}, first_rc{0};
  std::atomic<bool> go{false};
  std::string first_text;
  std::atomic<bool> text_taken{false};
  auto body = [&](int t) {
    ready.fetch_add(1);
    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
    for (int i = 0; i < calls; ++i) {
      const size_t k = (size_t)t * (size_t)calls + (size_t)i, row = k * per_call * width;
      const auto t0 = std::chrono::steady_clock::now();
      int rc;
      if (op == 0) {
        rc = is_pool ? api->pool_batch_gate((tfhe_hip_pool *)handle, gates[k], a + row, b ? b + row : nullptr, out + row, per_call)
                     : api->batch_gate((tfhe_hip_ctx *)handle, gates[k], a + row, b ? b + row : nullptr, out + row, per_call);
      } else if (op == 1) {
        const uint32_t *tv = testvecs ? testvecs + k * (size_t)2 * TFHE_HIP_N : nullptr;
        rc = is_pool ? api->pool_batch_bootstrap((tfhe_hip_pool *)handle, a + row, tv, 0, 1, out + row, per_call)
                     : api->batch_bootstrap((tfhe_hip_ctx *)handle, a + row, tv, 0, 1, out + row, per_call);
      } else {
        rc = is_pool ? api->pool_batch_mux((tfhe_hip_pool *)handle, gates ? gates[k] & 1 : 0, a + row, b + row, c + row, out + row, per_call)
                     : api->batch_mux((tfhe_hip_ctx *)handle, gates ? gates[k] & 1 : 0, a + row, b + row, c + row, out + row, per_call);
      }
      if (call_ms) call_ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rc != TFHE_HIP_OK) {
        int expected = 0;
        if (first_rc.compare_exchange_strong(expected, rc)) {
          first_text = is_pool ? api->pool_last_error((tfhe_hip_pool *)handle) : api->last_error((tfhe_hip_ctx *)handle);
          text_taken.store(true, std::memory_order_release);
        }
        return;
      }
    }
  };
  std::vector<std::thread> team;
  team.reserve((size_t)threads);
  for (int t = 0; t < threads; ++t) team.emplace_back(body, t);
  while (ready.load() < threads) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto &th : team) th.join();
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const int rc = first_rc.load();
  if (rc != 0 && err && errlen) {
    const std::string text = text_taken.load(std::memory_order_acquire) ? first_text : std::string();
    const size_t n = text.size() < errlen - 1 ? text.size() : errlen - 1;
    memcpy(err, text.data(), n);
    err[n] = 0;
  }
  return rc;
}

}  // extern "C"
