// threads.h -- how this library starts host threads (header only; used by the index builder in libnimble_hip.so and by
// every thread pool of libnimble_host.so).
//
// A std::vector<std::thread> filled in a loop is a process abort waiting for a bad day: when the k-th std::thread
// constructor throws (EAGAIN: a per-user thread limit, a cgroup pids limit), unwinding destroys k joinable threads and
// ~thread calls std::terminate -- SIGABRT with one line on stderr.  Group never lets that happen: spawn() reports a failed
// start as `false`, whatever was started is joined by join() or by the destructor (also while an exception unwinds), and
// the callers either make do with the threads they got (work-sharing loops) or turn the failure into an ordinary error.
// The reference hands `num_cores` to its thread pools (src/bin/main.rs:121-128, src/process/bam.rs:152-154); a pool that
// cannot be built there is a panic with a message, never an abort without one.
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace nimble {
namespace threads {

// CPUs this process may use: hardware threads, cut by the affinity mask and by the cgroup quota (v2 cpu.max, v1 cfs quota);
// NIMBLE_CPUS overrides.  A container that was given 16 of a host's 256 CPUs sees 256 from hardware_concurrency().
inline unsigned usable_cpus() {
  static const unsigned cached = [] {
    unsigned n = std::thread::hardware_concurrency();
    if (!n) n = 4;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
      const int c = CPU_COUNT(&set);
      if (c > 0) n = std::min<unsigned>(n, (unsigned)c);
    }
    auto quota = [](const char *path, const char *period_path) -> double {
      FILE *f = fopen(path, "r");
      if (!f) return 0;
      char a[64] = {0}, b[64] = {0};
      const int got = fscanf(f, "%63s %63s", a, b);
      fclose(f);
      if (got < 1 || !strcmp(a, "max") || atof(a) <= 0) return 0;
      double period = got >= 2 ? atof(b) : 0;
      if (period_path) {
        FILE *g = fopen(period_path, "r");
        if (g) {
          if (fscanf(g, "%63s", b) == 1) period = atof(b);
          fclose(g);
        }
      }
      return period > 0 ? atof(a) / period : 0;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", nullptr);                                                   // cgroup v2
    if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");  // v1
    if (q > 0) n = std::min<unsigned>(n, std::max(1u, (unsigned)(q + 0.5)));
    if (const char *e = getenv("NIMBLE_CPUS")) n = (unsigned)std::max(1, atoi(e));
    return std::max(1u, n);
  }();
  return cached;
}

// Test hook: NIMBLE_FAIL_SPAWN_AT=k makes the k-th thread start of the process (counted from 0 over every Group) and all
// later ones fail the way a thread limit would, so that every pool's degraded path can be exercised without one.
inline bool spawn_refused() {
  static const long at = [] {
    const char *e = getenv("NIMBLE_FAIL_SPAWN_AT");
    return e ? atol(e) : -1L;
  }();
  if (at < 0) return false;
  static std::atomic<long> started{0};
  return started.fetch_add(1) >= at;
}

class Group {
 public:
  Group() = default;
  Group(const Group &) = delete;
  Group &operator=(const Group &) = delete;
  ~Group() { join(); }
  // start a thread running f(); false (and nothing started) when the system refuses it
  template <class F>
  bool spawn(F &&f) noexcept {
    if (spawn_refused()) return false;
    try {
      th_.emplace_back(std::forward<F>(f));
      return true;
    } catch (...) {  // std::system_error (EAGAIN) from the constructor, std::bad_alloc from the vector
      return false;
    }
  }
  size_t size() const { return th_.size(); }
  void join() noexcept {
    for (auto &t : th_)
      if (t.joinable()) t.join();
    th_.clear();
  }

 private:
  std::vector<std::thread> th_;
};

// work() on up to `helpers` threads beside the calling one (work-sharing loops over an atomic cursor: fewer threads only
// take longer).  Returns the number of threads that ran it, the caller included.
template <class F>
unsigned run_beside(unsigned helpers, F work) {
  Group g;
  unsigned n = 1;
  for (unsigned t = 0; t < helpers; ++t) {
    if (!g.spawn(work)) break;
    ++n;
  }
  work();
  g.join();
  return n;
}

// fn(t) for every t in [0, tasks), each on a thread of its own where the system allows; the tasks whose thread could not
// be started run on the calling thread afterwards.  The first exception of any task is rethrown after all have ended.
template <class F>
void run_indexed(unsigned tasks, F fn) {
  if (tasks <= 1) {
    if (tasks) fn(0u);
    return;
  }
  std::vector<std::exception_ptr> err(tasks);
  auto guarded = [&](unsigned t) {
    try {
      fn(t);
    } catch (...) {
      err[t] = std::current_exception();
    }
  };
  {
    Group g;
    unsigned started = 0;
    for (; started + 1 < tasks; ++started)
      if (!g.spawn([&, started] { guarded(started); })) break;
    for (unsigned t = started; t < tasks; ++t) guarded(t);  // the last task, and whatever found no thread
    g.join();
  }
  for (auto &e : err)
    if (e) std::rethrow_exception(e);
}

// fn(t) for every t in [0, tasks) AT THE SAME TIME, each on a thread of its own -- for tasks that meet each other at
// barriers or collectives (one rank per device), where running some of them without the others would wait forever.  The
// threads are held at a gate until all of them exist; when the system refuses one, none runs and the call returns false.
// The first exception of any task is rethrown after all have ended.
template <class F>
bool run_all_or_none(unsigned tasks, F fn) {
  if (tasks == 0) return true;
  std::vector<std::exception_ptr> err(tasks);
  std::mutex mu;
  std::condition_variable cv;
  int gate = 0;  // 0 = wait, 1 = go, -1 = cancelled
  bool all = true;
  {
    Group g;
    for (unsigned t = 0; t < tasks; ++t) {
      const bool ok = g.spawn([&, t] {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return gate != 0; });
          if (gate < 0) return;
        }
        try {
          fn(t);
        } catch (...) {
          err[t] = std::current_exception();
        }
      });
      if (!ok) {
        all = false;
        break;
      }
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      gate = all ? 1 : -1;
    }
    cv.notify_all();
    g.join();
  }
  if (!all) return false;
  for (auto &e : err)
    if (e) std::rethrow_exception(e);
  return true;
}

}  // namespace threads
}  // namespace nimble
