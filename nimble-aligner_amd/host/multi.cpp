// multi.cpp -- successive score::calls over the GPUs of one node, driven natively: one host thread per rank, the
// pipelined step of the C ABI (include/nimble_hip.h nimble_steps_*), RCCL inside libnimble_hip.so, no torch.
//
// What it stands for in the reference: a host that runs one score::call after the other on a pool of workers
// (src/process/bam.rs:183-226 is the reference's only parallel consumer); here the workers are GPUs and one call spans
// all of them, because the dedup scope of a call is the call (src/align.rs:496-505): equal read keys have to meet on one
// rank, so every batch is packed where it lies, routed by key hash and exchanged before the ranks run their share.
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "../../include/nimble_hip.h"
#include "nimble_host.hpp"

namespace nimble {
namespace process {
namespace multi {

namespace {

// a rank that ended with an exception never arrives: abort() lets everybody out, and wait() says so by throwing
struct Aborted {};
struct Barrier {
  explicit Barrier(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> lock(mu_);
    if (aborted_) throw Aborted();
    const uint64_t gen = gen_;
    if (++arrived_ == n_) {
      arrived_ = 0;
      ++gen_;
      cv_.notify_all();
    } else {
      cv_.wait(lock, [&] { return gen_ != gen || aborted_; });
      if (gen_ == gen) throw Aborted();
    }
  }
  void abort() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      aborted_ = true;
    }
    cv_.notify_all();
  }
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, arrived_ = 0;
  uint64_t gen_ = 0;
  bool aborted_ = false;
};

void check_dev(int rc, const char *what) {
  if (rc != 0) throw Panic(std::string(what) + ": " + nimble_last_error());
}

}  // namespace

StepsResult run_steps(std::vector<std::unique_ptr<align::PseudoAligner>> &indices,
                      const reference_library::Reference &reference, const align::AlignFilterConfig &config,
                      const std::vector<int> &devices, const std::vector<std::vector<const uint8_t *>> &reads,
                      const std::vector<std::vector<const uint8_t *>> &mates, uint64_t n, uint32_t fixed_len, int warmup,
                      int steps, int align_grid_pct) {
  const int W = (int)devices.size();
  if (W < 1 || (int)indices.size() != W || (int)reads.size() != W) throw Panic("run_steps: one index and one read set list per rank");
  const bool paired = !mates.empty();
  if (paired && (int)mates.size() != W) throw Panic("run_steps: mates for every rank or for none");
  const size_t n_sets = reads[0].size();
  if (n_sets == 0 || steps < 1 || warmup < 0) throw Panic("run_steps: bad argument");
  nimble_comm *comm = nullptr;
  check_dev(nimble_comm_create(devices.data(), W, &comm), "nimble_comm_create");
  struct CommGuard {
    nimble_comm *c;
    ~CommGuard() { nimble_comm_free(c); }
  } guard{comm};
  nimble_align_params prm;
  align::device_params(config, &prm);
  const uint32_t max_len = std::max<uint32_t>(32, (fixed_len + 31u) / 32u * 32u);

  // the callsets of the job: agreed by content (this is one process: a shared dictionary); ids in order of first sight
  std::mutex dict_mu;
  std::map<std::vector<std::string>, size_t> dict;
  std::vector<const std::vector<std::string> *> by_id;
  Barrier bar(W);
  std::vector<std::exception_ptr> err((size_t)W);
  std::vector<int64_t> final_vec;
  std::chrono::steady_clock::time_point t0, t1;
  std::atomic<bool> failed{false};

  auto rank_main = [&](int r) {
    align::PseudoAligner &pa = *indices[(size_t)r];
    nimble_ctx *call[2] = {pa.ctx(0), pa.ctx(1)};
    nimble_ctx *util = pa.ctx(2);
    for (int k = 0; k < 2; ++k) {
      // room beside the persistent align grid for RCCL's kernels; the tail of a call stays on the launch stream (one
      // stream fewer next to the exchange)
      check_dev(nimble_ctx_set_option(call[k], NIMBLE_OPT_ALIGN_GRID_PCT, align_grid_pct), "nimble_ctx_set_option");
      check_dev(nimble_ctx_set_option(call[k], NIMBLE_OPT_TAIL_ASIDE, 0), "nimble_ctx_set_option");
    }
    check_dev(nimble_steps_begin(comm, r, call[0], call[1], util, &prm, paired ? 1 : 0, max_len), "nimble_steps_begin");
    std::vector<size_t> l2g;  // this rank's memo ids -> job ids
    std::vector<int64_t> vec;
    // F: the rows of the call in `slot`, into the job's table
    auto finish = [&](int slot, bool keep) {
      align::CallOutput out = align::end_calls(0, pa, reference, config, slot);
      const align::RowRefs &rr = out.refs;
      bool fresh = false;
      for (int32_t id : rr.ids)
        if ((size_t)id >= l2g.size() || l2g[(size_t)id] == (size_t)-1) fresh = true;
      if (fresh) {
        std::lock_guard<std::mutex> lk(dict_mu);
        for (size_t i = 0; i < rr.size(); ++i) {
          const size_t id = (size_t)rr.ids[i];
          if (id >= l2g.size()) l2g.resize(id + 1, (size_t)-1);
          if (l2g[id] != (size_t)-1) continue;
          auto ins = dict.emplace(rr.features(i), dict.size());
          if (ins.second) by_id.push_back(&ins.first->first);
          l2g[id] = ins.first->second;
        }
      }
      bar.wait();  // the dictionary holds every rank's callsets of this step
      size_t len;
      {
        std::lock_guard<std::mutex> lk(dict_mu);
        len = dict.size();
      }
      vec.assign(len, 0);
      for (size_t i = 0; i < rr.size(); ++i) vec[l2g[(size_t)rr.ids[i]]] += rr.counts[i];
      check_dev(nimble_counts_allreduce_host(comm, r, vec.data(), vec.size()), "nimble_counts_allreduce_host");
      if (keep && r == 0) final_vec = vec;
    };
    const int total = warmup + steps;
    int prev_slot = -1;      // the call launched by the previous submit
    for (int b = 0; b < total; ++b) {
      if (b == warmup) {
        bar.wait();
        if (r == 0) t0 = std::chrono::steady_clock::now();
      }
      const size_t set = (size_t)b % n_sets;
      nimble_ctx *launched = nullptr;
      check_dev(nimble_steps_submit(comm, r, reads[(size_t)r][set], nullptr, paired ? mates[(size_t)r][set] : nullptr, nullptr, n,
                                    fixed_len, NIMBLE_MEM_DEVICE, &launched),
                "nimble_steps_submit");
      // F(b - 2) while C(b - 1) runs and X(b) moves
      if (prev_slot >= 0) finish(prev_slot, false);
      prev_slot = launched ? (launched == call[0] ? 0 : 1) : -1;
    }
    nimble_ctx *launched = nullptr;
    check_dev(nimble_steps_flush(comm, r, &launched), "nimble_steps_flush");
    if (prev_slot >= 0) finish(prev_slot, launched == nullptr);
    if (launched) finish(launched == call[0] ? 0 : 1, true);
    bar.wait();
    if (r == 0) t1 = std::chrono::steady_clock::now();
    check_dev(nimble_steps_end(comm, r), "nimble_steps_end");
  };
  // One thread per rank, all or none (csrc/threads.h).  A rank that fails takes the others out with it instead of taking
  // the process down: a failure the collectives have agreed on (a bad argument, a full buffer, an RCCL error) is every
  // rank's own return value already; a host-side fault of ONE rank aborts the barrier here and the communicator
  // (nimble_comm_abort), so that the ranks waiting for it return with an error too.  The first failure is the one reported.
  std::atomic<int> first_failed{-1};
  const bool ran = threads::run_all_or_none((unsigned)W, [&](unsigned ur) {
    const int r = (int)ur;
    try {
      rank_main(r);
    } catch (const Aborted &) {
      // (another rank failed first; its error is the job's)
    } catch (...) {
      int none = -1;
      if (first_failed.compare_exchange_strong(none, r)) err[(size_t)r] = std::current_exception();
      failed = true;
      bar.abort();
      nimble_comm_abort(comm);
    }
  });
  if (!ran) throw Panic("run_steps: could not start one thread per rank (thread limit reached)");
  if (failed) {
    // (the communicator is freed below without draining the exchange streams: an exchange whose peer never posted its
    // half would never end)
    const int f = first_failed.load();
    if (f >= 0 && err[(size_t)f]) std::rethrow_exception(err[(size_t)f]);
    throw Panic("run_steps: a rank failed");
  }
  StepsResult res;
  res.ms_per_step = std::chrono::duration<double, std::milli>(t1 - t0).count() / (double)steps;
  res.rccl = nimble_comm_uses_rccl(comm) != 0;
  std::map<std::vector<std::string>, int64_t> table;
  for (size_t id = 0; id < final_vec.size() && id < by_id.size(); ++id)
    if (final_vec[id]) table[*by_id[id]] = final_vec[id];
  res.last = align::rows_from_table(*indices[0], reference, config, table);
  return res;
}

}  // namespace multi
}  // namespace process
}  // namespace nimble
