// The queue of the combining front end (rs-tfhe_amd/csrc/combine_queue.hpp) under a stand-in for the launch, built with
// -fsanitize=thread (tests/test_cpp_mirror.py): many threads, many small requests, several lane counts.  Every request
// must be served exactly once, by exactly one leader, with its own result; a request's frame goes away the moment it
// returns (so a leader that touched a request after marking it done would be a use-after-return here); quiesce() and
// with_idle_lanes() run beside the traffic.  ThreadSanitizer reports any data race on the requests' plain fields
// (next, count, the payload), which are handed between threads by the atomics alone.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../rs-tfhe_amd/csrc/combine_queue.hpp"

struct Req : combq::Node {
  uint64_t in = 0, out = 0;
  int served = 0;   // plain on purpose: written by the leader, read by the owner after DONE
  int leader = -1;  // lane that served it
};

static int run_case(int lanes, int threads, int calls, int sleep_us) {
  combq::Queue Q;
  Q.nlanes = lanes;
  std::atomic<int> in_flight[combq::Queue::kLanes] = {};
  std::atomic<long> bad{0}, rounds{0}, served{0};
  auto runner = [&](std::vector<combq::Node *> &nodes, int li) {
    if (in_flight[li].fetch_add(1) != 0) ++bad;  // two leaders on one lane
    uint64_t units = 0;
    for (combq::Node *n : nodes) {
      Req *r = static_cast<Req *>(n);
      if (r->state.load() != combq::Node::TAKEN) ++bad;
      r->out = r->in * 3 + 1;
      ++r->served;
      r->leader = li;
      units += r->count;
    }
    if (sleep_us) std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));  // the "launch"
    in_flight[li].fetch_sub(1);
    ++rounds;
    served += (long)nodes.size();
    combq::Round rd;
    rd.launches = 1;
    (void)units;
    return rd;
  };
  std::atomic<bool> stop{false};
  std::thread side([&] {  // what a key change / a profiling switch does, beside the traffic
    while (!stop.load()) {
      Q.quiesce();
      Q.with_idle_lanes([&] {
        for (int i = 0; i < combq::Queue::kLanes; ++i)
          if (in_flight[i].load() != 0) ++bad;  // held idle means idle
      });
      std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
  });
  std::vector<std::thread> team;
  for (int t = 0; t < threads; ++t)
    team.emplace_back([&, t] {
      for (int i = 0; i < calls; ++i) {
        Req r;  // on this frame: gone right after submit() returns
        r.in = (uint64_t)t * 1000003u + (uint64_t)i;
        r.count = 1 + (size_t)((t + i) % 3);
        Q.submit(r, runner);
        if (r.served != 1 || r.out != r.in * 3 + 1 || r.leader < 0 || r.leader >= lanes) ++bad;
      }
    });
  for (auto &th : team) th.join();
  stop.store(true);
  side.join();
  if (Q.pending.load() != 0) ++bad;
  if (served.load() != (long)threads * calls || (long)Q.st_requests != (long)threads * calls) ++bad;
  std::printf("lanes %d, %d threads x %d calls, launch %d us: %ld rounds (%.1f requests each), %llu lingers, bad = %ld\n", lanes, threads, calls,
              sleep_us, rounds.load(), (double)served.load() / (double)(rounds.load() ? rounds.load() : 1), (unsigned long long)Q.st_lingers, bad.load());
  return bad.load() ? 1 : 0;
}

int main() {
  int failures = 0;
  failures += run_case(1, 1, 200, 0);     // a lone caller: every call its own round
  failures += run_case(1, 8, 300, 100);
  failures += run_case(1, 48, 120, 200);  // one lane (the shipped setting)
  failures += run_case(2, 48, 120, 200);
  failures += run_case(4, 64, 60, 50);
  failures += run_case(1, 64, 100, 0);    // no launch time at all: leaders hand the lane over as fast as they can
  std::printf(failures ? "%d FAILURES\n" : "all queue checks passed (%d failures)\n", failures);
  return failures ? 1 : 0;
}
