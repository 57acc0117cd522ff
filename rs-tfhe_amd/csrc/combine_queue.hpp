// combine_queue.hpp -- the queue of the combining front end (combine.hpp), host code only: no HIP in this file, so that
// tests/cpp/test_combine_queue.cpp can run it under ThreadSanitizer on a CPU with a stand-in for the launch.
//
// No mutex on the callers' path: with hundreds of threads released at the same instant by one merged launch, a
// condition variable's mutex is re-acquired by every one of them in turn (measured: a 256-thread team spent as long in
// that queue as in the launch).  Arrivals push themselves onto a lock-free list, lanes are bits of one word, and
// everybody who has to wait sleeps on ONE futex word (`epoch`) that is bumped whenever a round completes or a lane
// becomes free; woken threads look at their own request's state and go back to sleep if it is not their turn.
#pragma once
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

namespace combq {

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}
inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// What a caller queues (the front end's request derives from it).  QUEUED until a leader takes it, DONE when it has
// been served.  The owner's frame may go away the moment it reads DONE: a leader does not touch a node after storing that.
struct Node {
  enum : uint32_t { QUEUED = 0, TAKEN = 1, DONE = 2 };
  std::atomic<uint32_t> state{QUEUED};
  Node *next = nullptr;  // the arrival list (newest first)
  size_t count = 0;      // units (ciphertexts) the request carries
};

// what a leader's run reports back
struct Round {
  size_t launches = 0;
  double pack_us = 0, gpu_us = 0, unpack_us = 0;
};

struct Queue {
  static constexpr int kLanes = 4;  // most lanes a queue can have
  int nlanes = 1;                   // lanes in use
  std::atomic<uint64_t> lane_gen[kLanes] = {};  // leader rounds completed per lane
  std::atomic<Node *> arrivals_head{nullptr};
  std::atomic<uint32_t> lanes_busy{0};  // bit i: lane i has a leader
  std::atomic<uint32_t> epoch{0};       // the futex word
  std::atomic<uint32_t> sleepers{0};    // threads in (or about to enter) futex_wait: nobody to wake, no system call
  std::atomic<uint32_t> collecting{0};  // leaders between taking a lane and taking the arrival list: arrivals do not lead, they will be taken
  std::atomic<size_t> pending{0};       // units queued or in flight
  std::atomic<uint64_t> arrivals{0};    // requests ever queued (the lingering leader watches it grow)
  std::atomic<uint64_t> taken{0};       // requests ever taken by a leader
  std::atomic<int64_t> last_arrival_ns{0};  // when the last request was queued (bulk launches yield while small calls are arriving)
  // The last leader round that completed: how many callers it knew of -- the requests it carried PLUS those already
  // waiting behind it when it finished (in a closed loop that is the whole team: a leader that waited only for as many
  // as the last round carried settles at a part of the team and leaves the rest a launch behind, measured: 165 of 256
  // per round) -- and when (steady_clock ns).
  std::atomic<size_t> last_reqs{0};
  std::atomic<int64_t> last_done_ns{0};
  // Lingering: only within linger_window of a round that knew of several callers; ends when as many requests are waiting
  // as that (`want`), when nobody has arrived for linger_quiet + want / 4 microseconds, or after linger_max + want
  // microseconds (a team of hundreds of threads takes that long to come back through the scheduler; a launch that leaves
  // without most of them makes them wait a whole launch).
  long linger_window_us = 1000, linger_quiet_us = 50, linger_max_us = 250;
  // statistics; leaders only
  std::mutex st_mu;
  uint64_t st_launches = 0, st_requests = 0, st_units = 0, st_max_requests = 0, st_lingers = 0;
  double st_linger_us = 0, st_pack_us = 0, st_gpu_us = 0, st_unpack_us = 0;  // where the leaders' time went

  // ---- waiting and waking: one futex word ------------------------------------------------------------------------
  void wait(uint32_t seen) {
    sleepers.fetch_add(1, std::memory_order_seq_cst);
    if (epoch.load(std::memory_order_seq_cst) == seen)
      (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&epoch), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
    sleepers.fetch_sub(1, std::memory_order_seq_cst);
  }
  void wake_all() {
    epoch.fetch_add(1, std::memory_order_seq_cst);
    if (sleepers.load(std::memory_order_seq_cst))
      (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&epoch), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
  }
  int try_lane() {  // a free lane, now this thread's; -1: none
    uint32_t busy = lanes_busy.load(std::memory_order_relaxed);
    for (;;) {
      int li = -1;
      for (int i = 0; i < nlanes && li < 0; ++i)
        if (!(busy & (1u << i))) li = i;
      if (li < 0) return -1;
      if (lanes_busy.compare_exchange_weak(busy, busy | (1u << li), std::memory_order_acquire, std::memory_order_relaxed)) return li;
    }
  }
  void release_lane(int li) {
    lane_gen[li].fetch_add(1, std::memory_order_release);
    lanes_busy.fetch_and(~(1u << li), std::memory_order_release);
    wake_all();
  }

  // The calling thread holds lane `li`: it takes whatever has arrived, has `run(all, li)` serve it (the runner fills in
  // each request's result; it must not touch `state`), marks it done and releases the lane.
  template <class Run>
  void lead(int li, Run &&run) {
    collecting.fetch_add(1, std::memory_order_seq_cst);
    // the callers the last round knew of are on their way back: give them a bounded moment
    {
      const size_t want = last_reqs.load(std::memory_order_relaxed);
      const uint64_t taken0 = taken.load(std::memory_order_relaxed);
      auto waiting = [&] { return (size_t)(arrivals.load(std::memory_order_relaxed) - taken0); };
      const int64_t t0 = now_ns();
      if (want > 1 && waiting() < want && t0 - last_done_ns.load(std::memory_order_relaxed) < linger_window_us * 1000) {
        int64_t last_growth = t0, now = t0;
        size_t seen = waiting();
        for (;;) {
          cpu_relax();
          now = now_ns();
          const size_t cur = waiting();
          if (cur != seen) {
            seen = cur;
            last_growth = now;
          }
          if (cur >= want || now - last_growth > (linger_quiet_us * 4 + (long)want) * 250 || now - t0 > (linger_max_us + (long)want) * 1000) break;
        }
        std::lock_guard<std::mutex> lk(st_mu);
        ++st_lingers;
        st_linger_us += (double)(now - t0) * 1e-3;
      }
    }
    // take the arrival list (newest first) and put it in arrival order
    std::vector<Node *> all;
    for (Node *r = arrivals_head.exchange(nullptr, std::memory_order_acquire); r;) {
      Node *nx = r->next;  // (read before anything can complete the request)
      all.push_back(r);
      r = nx;
    }
    std::reverse(all.begin(), all.end());
    taken.fetch_add(all.size(), std::memory_order_relaxed);
    for (Node *r : all) r->state.store(Node::TAKEN, std::memory_order_seq_cst);
    collecting.fetch_sub(1, std::memory_order_seq_cst);
    // a request that arrived after the list was taken may have seen `collecting` and gone to sleep expecting to be taken
    if (arrivals_head.load(std::memory_order_seq_cst)) wake_all();
    if (!all.empty()) {
      size_t total = 0;
      for (Node *r : all) total += r->count;
      const Round rd = run(all, li);
      pending.fetch_sub(total, std::memory_order_relaxed);
      last_reqs.store(all.size() + (size_t)(arrivals.load(std::memory_order_relaxed) - taken.load(std::memory_order_relaxed)), std::memory_order_relaxed);
      last_done_ns.store(now_ns(), std::memory_order_relaxed);
      {
        std::lock_guard<std::mutex> lk(st_mu);
        st_launches += rd.launches;
        st_requests += all.size();
        st_units += total;
        if (all.size() > st_max_requests) st_max_requests = all.size();
        st_pack_us += rd.pack_us;
        st_gpu_us += rd.gpu_us;
        st_unpack_us += rd.unpack_us;
      }
      for (Node *r : all) r->state.store(Node::DONE, std::memory_order_release);  // (r may be gone after this)
    }
    release_lane(li);
  }

  // queue the request, lead when a lane is free, return when the request has been served
  template <class Run>
  void submit(Node &r, Run &&run) {
    pending.fetch_add(r.count, std::memory_order_relaxed);
    {
      Node *h = arrivals_head.load(std::memory_order_relaxed);
      do r.next = h;
      while (!arrivals_head.compare_exchange_weak(h, &r, std::memory_order_release, std::memory_order_relaxed));
    }
    arrivals.fetch_add(1, std::memory_order_relaxed);
    last_arrival_ns.store(now_ns(), std::memory_order_relaxed);
    for (;;) {
      const uint32_t e = epoch.load(std::memory_order_seq_cst);  // (before the checks: a wake-up in between is not lost)
      const uint32_t st = r.state.load(std::memory_order_acquire);
      if (st == Node::DONE) break;
      if (st == Node::QUEUED && collecting.load(std::memory_order_seq_cst) == 0) {
        const int li = try_lane();
        if (li >= 0) {
          if (r.state.load(std::memory_order_seq_cst) == Node::QUEUED) lead(li, run);  // takes this request too
          else release_lane(li);  // another leader took it meanwhile
          continue;
        }
      }
      wait(e);
    }
  }

  // Whatever the lanes had in flight when this is called has completed when it returns.
  void quiesce() {
    for (int i = 0; i < kLanes; ++i) {
      const uint64_t g = lane_gen[i].load(std::memory_order_acquire);
      for (;;) {
        const uint32_t e = epoch.load(std::memory_order_seq_cst);
        if (!(lanes_busy.load(std::memory_order_acquire) & (1u << i)) || lane_gen[i].load(std::memory_order_acquire) != g) break;
        wait(e);
      }
    }
  }

  // every lane idle, and held idle while f runs (f must not submit)
  template <class F>
  void with_idle_lanes(F &&f) {
    const uint32_t all = (1u << kLanes) - 1;
    for (;;) {  // take every lane bit at once
      const uint32_t e = epoch.load(std::memory_order_seq_cst);
      uint32_t none = 0;
      if (lanes_busy.compare_exchange_strong(none, all, std::memory_order_acquire, std::memory_order_relaxed)) break;
      wait(e);
    }
    f();
    lanes_busy.store(0, std::memory_order_release);
    wake_all();
  }
};

}  // namespace combq
