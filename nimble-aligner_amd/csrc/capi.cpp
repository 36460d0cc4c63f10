// capi.cpp -- implementation of the C ABI declared in include/nimble_hip.h.
//
// Owns device memory, streams and the launch sequence of one `score::call` (src/score.rs:14-46):
//   pack -> align -> intern (claim / verify rounds) -> dedup -> count
// There is no CPU path here: every entry point that computes needs a HIP device.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "flat_index.h"
#include "kernels.h"

using namespace nimble;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return fail(NIMBLE_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                    \
  } while (0)

// HIP keeps the last error of ANY earlier call of the process (torch, RCCL, another library) until somebody reads it:
// every entry point that launches reads it away first, so that the check behind its own launches reports its own errors
// only (round 2: a use-after-free in one test surfaced as "invalid device ordinal" in the next test's first call).
#define DRAIN_STALE_HIP_ERROR() (void)hipGetLastError()

uint64_t env_u64(const char *name, uint64_t dflt) {
  const char *v = getenv(name);
  if (!v || !*v) return dflt;
  return strtoull(v, nullptr, 10);
}

uint64_t pow2_at_least(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

// growable device buffer
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes, uint64_t *total) {
    if (bytes <= cap) return NIMBLE_OK;
    if (p) {
      (void)hipFree(p);
      if (total) *total -= cap;
      p = nullptr;
      cap = 0;
    }
    size_t want = bytes + 64;  // tail padding: 16-byte vector loads may touch the last partial chunk
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return fail(NIMBLE_E_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    cap = bytes;
    if (total) *total += cap;
    return NIMBLE_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace

struct nimble_index {
  int device = 0;
  // One result stream for all contexts of the index.  HIP multiplexes streams onto a few hardware queues
  // (GPU_MAX_HW_QUEUES, 4 by default); two streams that share a queue run in order, so every extra stream makes it
  // likelier that somebody else's stream (RCCL's) lands behind the launch stream's persistent kernels.
  hipStream_t copy_stream = nullptr;
  int n_ctx = 0;  // contexts alive on this index
  // The class table (intern slots, descriptors, id pool) is the one part of the index that calls write.  Claim and
  // verify kernels of different contexts -- possibly on different streams, enqueued by different host threads -- are
  // chained through this event so that they never run beside each other: a kernel boundary lies between one context's
  // claims and the next one's, which is what keeps class ids canonical (one id per content).
  std::mutex intern_mu;
  hipEvent_t ev_intern = nullptr;
  bool intern_chained = false;
  hipStream_t intern_last = nullptr;  // the stream ev_intern was last recorded on
  bool released = false;  // nimble_index_free was called while contexts were alive: the last context frees the index
  DevIndex dev{};
  DevBuf b_ht, b_bitmap, b_l1, b_mleft, b_rec, b_ledge, b_unitig, b_cls_desc, b_cls_off, b_cls_ids, b_cls_bits, b_intern, b_dyn_state,
      b_srec, b_srec_first, b_srec_base, b_srec_many;
  uint64_t device_bytes = 0;
  uint64_t n_kmers = 0, n_nodes = 0, n_static = 0, unitig_bases = 0, static_entries = 0, ht_slots = 0;
  std::vector<uint32_t> h_col_off, h_col_ids;  // host mirror of the static classes
  ~nimble_index() {
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (ev_intern) (void)hipEventDestroy(ev_intern);
    for (DevBuf *b : {&b_ht, &b_bitmap, &b_l1, &b_mleft, &b_rec, &b_ledge, &b_unitig, &b_cls_desc, &b_cls_off, &b_cls_ids, &b_cls_bits, &b_intern,
                      &b_dyn_state, &b_srec, &b_srec_first, &b_srec_base, &b_srec_many})
      b->release();
  }
};

struct nimble_ctx {
  nimble_index *ix = nullptr;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // Results come back on a side stream that waits for the call's last event: when two contexts share one
  // launch stream (calls in flight back to back), reading one call's results does not wait for the next call.
  hipStream_t copy_stream = nullptr;
  uint64_t *p_state = nullptr;  // pinned mirrors of b_state / the index's dyn_state at call start
  uint32_t *p_dyn = nullptr;
  CallBuffers cb{};
  uint64_t bytes = 0;
  // streamed call (nimble_stream_*): reads arrive in batches, arrays are laid out for `stream_cap` reads
  bool streaming = false;
  uint64_t stream_n = 0, stream_cap = 0;
  uint32_t stream_max_len = 0;
  hipStream_t h2d_stream = nullptr;
  DevBuf b_stage[2][2], b_stage_off[2][2];  // [slot][mate]: double-buffered device staging of host batches
  hipEvent_t ev_h2d[2] = {}, ev_used[2] = {};
  bool stage_busy[2] = {false, false};
  int stage_k = 0;
  uint64_t align_from = 0;  // streamed call: first appended read the align kernel has not been launched for yet
  bool h2d_pending[2] = {false, false};  // staging slots whose copy the host has not waited for yet (NIMBLE_MEM_HOST_PINNED)
  DevBuf b_keys, b_len[2], b_hash, b_pre[2], b_reason[2], b_score[2], b_mism[2], b_cls[2], b_dyn_off[2], b_dyn_len[2],
      b_dyn_hash[2], b_dyn_pos[2], b_slot, b_counted, b_scratch, b_ws, b_dedup, b_hist_keys, b_hist_cnt, b_state;
  DevBuf b_tile_ctr;          // tile counters of the align launches (kernels.h CallBuffers::tile_ctr)
  DevBuf b_in[2], b_in_off[2];  // staging of host inputs
  DevBuf b_plog;
  uint32_t plog_max_len = 0;
  DevBuf b_min_cov;
  double min_cov_percent = -1.0;
  uint32_t min_cov_len = 0;
  DevBuf b_out_c1, b_out_c2, b_out_cnt, b_out_seg, b_out_rep;
  // The results the host reads after every call (state words, compacted histogram) are written by the kernels
  // straight into page-locked host memory: a device-to-host copy of a few kB is a blit KERNEL, and behind the
  // persistent align grid of the next call in flight it waited ~1.3 ms for a CU (tools/end_probe.py).
  struct HostOut {
    void *base = nullptr;
    uint64_t cap = 0;  // entries
    uint32_t *c1 = nullptr, *c2 = nullptr, *seg = nullptr, *rep = nullptr;
    uint64_t *cnt = nullptr;
  } hout;
  bool out_pinned = false;  // the last compaction went to hout
  bool tail_aside = false;  // the last call's dedup / count / compaction ran on the side stream
  uint32_t tail_aside_grid = 0;  // workgroups of k_dedup there (NIMBLE_OPT_TAIL_ASIDE; 0 = stay on the launch stream)
  // BAM-mode extras (nimble_call_ex)
  DevBuf b_route;  // scratch of nimble_route_records
  // align-where-the-reads-are form (nimble_ctx_defer_dedup): the next call routes its keys and stops before dedup
  struct Defer {
    uint32_t world = 0;            // 0 = not armed
    uint32_t world_of_call = 0;
    uint32_t counts_world = 0;     // destinations of the last routing (nimble_route_counts)
    uint64_t *records = nullptr;   // caller's device buffers
    uint32_t *perm = nullptr;
    const uint8_t *verdict = nullptr;
    bool active = false;           // the call in flight is a deferred one
    bool routed = false;           // its keys were routed (never again: the verdicts follow that record order)
    bool counted = false;          // nimble_count_verdicts was enqueued
    hipEvent_t ev_route = nullptr;
    uint64_t *p_counts = nullptr;  // pinned, 256 entries
  } defer;
  DevBuf b_seg, b_alen[2], b_skip[2], b_qual[2], b_trim_ls, b_trim_qp, b_hist_rep, b_hot;
  double trim_strictness = -1.0;
  uint64_t trim_target = ~0ULL;
  std::vector<uint32_t> h_seg, h_rep;
  uint64_t scratch_cap = 0;
  uint64_t hist_slots = 0;
  hipEvent_t ev[7] = {};
  bool have_events = false;
  bool called = false;
  int want_counters = 0;
  int align_grid_pct = 100;
  int stream_cus = 0;  // CUs the launch stream is confined to (0 = all): the persistent align grid is sized to them
  uint32_t dyn_before = 0, dyn_after = 0;
  // arguments of the call in flight (kept so that finish_call can re-enqueue after growing a pool)
  nimble_align_params prm{};
  const uint8_t *in_r[2] = {nullptr, nullptr};
  const uint64_t *in_off[2] = {nullptr, nullptr};
  uint32_t in_fixed_len = 0, in_max_len = 0;
  // nimble_call_words: the reads came packed (device pointers once staged); in_words[0] != nullptr selects k_pack_words
  const uint64_t *in_words[2] = {nullptr, nullptr};
  const uint32_t *in_len32[2] = {nullptr, nullptr};
  uint32_t in_stride[2] = {0, 0};
  uint64_t dslots = 0;
  uint64_t dedup_clean_slots = 0;  // slots [0, this) of b_dedup are known to be zero (cleared at the tail of the last call)
  bool finished = true;
  bool skip_pack = false;  // the call started from packed keys (nimble_call_packed)
  bool counted_marked = true;  // the `counted` flags of the last call are valid (k_count ran)
  int attempt = 0;
  std::vector<uint64_t> h_state = std::vector<uint64_t>(16, 0);
  std::vector<uint32_t> h_c1, h_c2;  // histogram of the last finished call, sorted by (c1, c2)
  std::vector<uint64_t> h_cnt;
  ~nimble_ctx() {
    for (DevBuf *b : {&b_keys, &b_len[0], &b_len[1], &b_hash, &b_pre[0], &b_pre[1], &b_reason[0], &b_reason[1],
                      &b_score[0], &b_score[1], &b_mism[0], &b_mism[1], &b_cls[0], &b_cls[1], &b_dyn_off[0],
                      &b_dyn_off[1], &b_dyn_len[0], &b_dyn_len[1], &b_dyn_hash[0], &b_dyn_hash[1], &b_dyn_pos[0],
                      &b_dyn_pos[1], &b_slot, &b_counted, &b_scratch, &b_ws, &b_dedup, &b_hist_keys, &b_hist_cnt,
                      &b_state, &b_tile_ctr, &b_in[0], &b_in[1], &b_in_off[0], &b_in_off[1], &b_plog, &b_min_cov, &b_out_c1, &b_out_c2,
                      &b_out_cnt, &b_out_seg, &b_out_rep, &b_seg, &b_alen[0], &b_alen[1], &b_skip[0], &b_skip[1], &b_qual[0],
                      &b_qual[1], &b_route, &b_trim_ls, &b_trim_qp, &b_hist_rep, &b_hot, &b_stage[0][0], &b_stage[0][1], &b_stage[1][0], &b_stage[1][1], &b_stage_off[0][0],
                      &b_stage_off[0][1], &b_stage_off[1][0], &b_stage_off[1][1]})
      b->release();
    if (h2d_stream) {
      (void)hipStreamDestroy(h2d_stream);
      for (int k = 0; k < 2; ++k) {
        (void)hipEventDestroy(ev_h2d[k]);
        (void)hipEventDestroy(ev_used[k]);
      }
    }
    if (have_events)
      for (auto &e : ev) (void)hipEventDestroy(e);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
    if (p_state) (void)hipHostFree(p_state);
    if (hout.base) (void)hipHostFree(hout.base);
    if (p_dyn) (void)hipHostFree(p_dyn);
    if (defer.ev_route) (void)hipEventDestroy(defer.ev_route);
    if (defer.p_counts) (void)hipHostFree(defer.p_counts);
  }
};

namespace {

template <class T>
int upload(DevBuf &b, const std::vector<T> &v, uint64_t *total, size_t min_elems = 0) {
  size_t n = std::max(v.size(), min_elems);
  int rc = b.ensure(std::max<size_t>(n * sizeof(T), 16), total);
  if (rc) return rc;
  if (!v.empty()) HIPCHK(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return NIMBLE_OK;
}

int ensure_plog(nimble_ctx *c, uint32_t max_len) {
  if (c->b_plog.p && c->plog_max_len >= max_len) return NIMBLE_OK;
  // row L (L+1 entries, at offset L(L+1)/2): (k/L) * log2(k/L) in IEEE double, the terms of
  // utils::shannon_entropy (src/utils.rs:111-116); computed by the host libm so that the device sum
  // reproduces the CPU value bit for bit
  uint32_t m = std::max<uint32_t>(max_len, 256);
  std::vector<double> t(((size_t)m + 1) * (m + 2) / 2, 0.0);
  for (uint32_t L = 1; L <= m; ++L) {
    double *row = t.data() + ((size_t)L * (L + 1)) / 2;
    for (uint32_t k = 1; k <= L; ++k) {
      double f = (double)k / (double)L;
      row[k] = f * std::log2(f);
    }
  }
  int rc = upload(c->b_plog, t, &c->bytes);
  if (rc) return rc;
  c->plog_max_len = m;
  return NIMBLE_OK;
}

// maxinfo's two integer tables (align.rs:873-897): length scores [1000] and quality scores [61], normalised to
// i64 exactly as the reference does (f64 arithmetic of the host libm, saturating cast); the device only adds them
int ensure_trim_tables(nimble_ctx *c, uint64_t target_length, double strictness) {
  if (c->b_trim_ls.p && c->trim_target == target_length && c->trim_strictness == strictness) return NIMBLE_OK;
  const size_t LONGEST_READ = 1000, MAXQUAL = 60;
  std::vector<double> ls(LONGEST_READ), qp(MAXQUAL + 1);
  for (size_t i = 0; i < LONGEST_READ; ++i) {
    const double pow1 = std::exp((double)target_length - (double)i - 1.0);
    ls[i] = std::log(1.0 / (1.0 + pow1)) + std::log((double)(i + 1)) * (1.0 - strictness);
  }
  for (size_t i = 0; i <= MAXQUAL; ++i)
    qp[i] = std::log(1.0 - std::pow(10.0, -((0.5 + (double)i) / 10.0))) * strictness;
  auto norm_ratio = [](const std::vector<double> &a, size_t margin) {
    double mx = std::fabs(a[0]);
    for (size_t i = 1; i < a.size(); ++i) mx = std::max(mx, std::fabs(a[i]));
    return (double)INT64_MAX / (mx * (double)margin);
  };
  auto as_i64 = [](double v) -> int64_t {  // Rust `as i64`: saturating, NaN -> 0
    if (std::isnan(v)) return 0;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
  };
  const double ratio = std::max(norm_ratio(ls, LONGEST_READ * 2), norm_ratio(qp, LONGEST_READ * 2));
  std::vector<int64_t> lsi(LONGEST_READ), qpi(MAXQUAL + 1);
  for (size_t i = 0; i < LONGEST_READ; ++i) lsi[i] = as_i64(ls[i] * ratio);
  for (size_t i = 0; i <= MAXQUAL; ++i) qpi[i] = as_i64(qp[i] * ratio);
  int rc = upload(c->b_trim_ls, lsi, &c->bytes);
  if (!rc) rc = upload(c->b_trim_qp, qpi, &c->bytes);
  if (rc) return rc;
  c->trim_target = target_length;
  c->trim_strictness = strictness;
  return NIMBLE_OK;
}

// min_cov[L] = smallest integer score s with (double)s / (double)L >= score_percent, evaluated with the
// host's IEEE division -- the device then decides `normalized_score >= score_percent` (filter/align.rs:16)
// with one integer compare, bit-exact
int ensure_min_cov(nimble_ctx *c, double percent, uint32_t max_len) {
  if (c->b_min_cov.p && c->min_cov_percent == percent && c->min_cov_len >= max_len) return NIMBLE_OK;
  uint32_t m = std::max<uint32_t>(max_len, 256);
  std::vector<uint32_t> t(m + 1, 0xFFFFFFFFu);
  for (uint32_t L = 1; L <= m; ++L) {
    // the quotient is monotone in s: binary search over 0 .. 4L (scores never exceed the read length)
    uint32_t lo = 0, hi = 4 * L + 1;
    while (lo < hi) {
      uint32_t mid = lo + (hi - lo) / 2;
      if ((double)mid / (double)L >= percent) hi = mid;
      else lo = mid + 1;
    }
    t[L] = lo <= 4 * L ? lo : 0xFFFFFFFFu;
  }
  t[0] = percent <= 0.0 ? 0u : 0xFFFFFFFFu;  // 0/0 is NaN: never >= percent unless the compare is vacuous
  int rc = upload(c->b_min_cov, t, &c->bytes);
  if (rc) return rc;
  c->min_cov_percent = percent;
  c->min_cov_len = m;
  return NIMBLE_OK;
}

// ev[6] marks the end of everything enqueued for the call so far; the state words are published to the host first
int mark_done(nimble_ctx *c, hipStream_t on = nullptr) {
  hipStream_t s = on ? on : c->stream;
  launch_publish_state(s, c->b_state.as<uint64_t>(), c->p_state);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->ev[6], s));
  return NIMBLE_OK;
}

int fetch_state(nimble_ctx *c) {
  HIPCHK(hipEventSynchronize(c->ev[6]));
  std::copy(c->p_state, c->p_state + 16, c->h_state.begin());
  return NIMBLE_OK;
}

int ensure_host_out(nimble_ctx *c, uint64_t cap) {
  if (c->hout.cap >= cap) return NIMBLE_OK;
  // a kernel of an earlier call may still be writing the old block: wait before it goes away
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->copy_stream) HIPCHK(hipStreamSynchronize(c->copy_stream));
  if (c->hout.base) (void)hipHostFree(c->hout.base);
  c->hout = nimble_ctx::HostOut();
  void *p = nullptr;
  if (hipHostMalloc(&p, cap * 24, hipHostMallocDefault) != hipSuccess)
    return fail(NIMBLE_E_NOMEM, "page-locked result buffer");
  c->hout.base = p;
  c->hout.cap = cap;
  c->hout.cnt = static_cast<uint64_t *>(p);
  c->hout.c1 = reinterpret_cast<uint32_t *>(c->hout.cnt + cap);
  c->hout.c2 = c->hout.c1 + cap;
  c->hout.seg = c->hout.c2 + cap;
  c->hout.rep = c->hout.seg + cap;
  return NIMBLE_OK;
}

// histogram compaction into the context's output arrays (entry count lands in state[11])
int enqueue_compact(nimble_ctx *c, hipStream_t on = nullptr) {
  hipStream_t st = on ? on : c->stream;
  const uint64_t slots = c->hist_slots;
  c->out_pinned = slots <= (1ULL << 22);  // 24 B per slot page-locked; larger tables go through device arrays
  HIPCHK(hipMemsetAsync((uint64_t *)c->b_state.p + 11, 0, 8, st));
  if (c->out_pinned) {
    int rc = ensure_host_out(c, slots);
    if (rc) return rc;
    launch_hist_compact(st, c->cb, c->hout.c1, c->hout.c2, c->hout.cnt, slots, c->hout.seg, c->hout.rep);
    return NIMBLE_OK;
  }
  int rc = c->b_out_c1.ensure(slots * 4, &c->bytes);
  if (!rc) rc = c->b_out_c2.ensure(slots * 4, &c->bytes);
  if (!rc) rc = c->b_out_cnt.ensure(slots * 8, &c->bytes);
  if (!rc) rc = c->b_out_seg.ensure(slots * 4, &c->bytes);
  if (!rc) rc = c->b_out_rep.ensure(slots * 4, &c->bytes);
  if (rc) return rc;
  launch_hist_compact(st, c->cb, c->b_out_c1.as<uint32_t>(), c->b_out_c2.as<uint32_t>(),
                      c->b_out_cnt.as<uint64_t>(), slots, c->b_out_seg.as<uint32_t>(), c->b_out_rep.as<uint32_t>());
  return NIMBLE_OK;
}

// count stage of a deferred call: the owners' verdicts pick the copies that enter the histogram
int enqueue_count_verdicts(nimble_ctx *c) {
  hipStream_t s = c->stream;
  launch_count_verdicts(s, c->prm, c->cb, c->defer.perm, c->defer.verdict);
  HIPCHK(hipEventRecord(c->ev[4], s));
  HIPCHK(hipEventRecord(c->ev[5], s));
  c->counted_marked = true;
  int rc = enqueue_compact(c);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  return mark_done(c);
}

// routing of a deferred call's keys, between pack and align: the host waits for ev_route only
int enqueue_route(nimble_ctx *c) {
  const uint32_t world = c->defer.world_of_call;
  const uint64_t cells = (uint64_t)route_grid() * world;
  uint64_t *block_first = c->b_route.as<uint64_t>();
  uint64_t *totals = block_first + cells;
  uint32_t *block_counts = reinterpret_cast<uint32_t *>(totals + 256);
  launch_route(c->stream, c->cb, world, block_counts, block_first, totals, c->defer.records, c->defer.perm);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->defer.p_counts, totals, world * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->defer.p_counts + 256, (uint64_t *)c->b_state.p + 14, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipEventRecord(c->defer.ev_route, c->stream));
  c->defer.counts_world = world;
  c->defer.routed = true;
  return NIMBLE_OK;
}

// One round of class interning (claim, then verify behind the kernel boundary), chained behind the interning of
// every other context of the index (see nimble_index::ev_intern).
int enqueue_intern_round(nimble_ctx *c, int round, hipStream_t on = nullptr) {
  nimble_index *ix = c->ix;
  hipStream_t s = on ? on : c->stream;
  std::lock_guard<std::mutex> lock(ix->intern_mu);
  if (!ix->ev_intern) HIPCHK(hipEventCreateWithFlags(&ix->ev_intern, hipEventDisableTiming));
  if (ix->intern_chained) HIPCHK(hipStreamWaitEvent(s, ix->ev_intern, 0));
  launch_intern_claim(s, ix->dev, c->cb, round);
  launch_intern_verify(s, ix->dev, c->cb);
  HIPCHK(hipEventRecord(ix->ev_intern, s));
  ix->intern_chained = true;
  ix->intern_last = s;
  return NIMBLE_OK;
}

// Head of a call: clear the per-call tables and counters.
int enqueue_head(nimble_ctx *c) {
  hipStream_t s = c->stream;
  CallBuffers &cb = c->cb;
  if (c->tail_aside) {  // the previous call of this context finished on the side stream: its buffers are re-used now
    HIPCHK(hipStreamWaitEvent(s, c->ev[6], 0));
    c->tail_aside = false;
  }
  if (c->dedup_clean_slots < c->dslots) HIPCHK(hipMemsetAsync(c->b_dedup.p, 0, c->dslots * 8, s));
  c->dedup_clean_slots = c->defer.active ? c->dslots : 0;  // a deferred call never touches its own table
  // histogram table, state words, hot-key set -- and the input-error latch of k_pack, unless the keys of this call were
  // packed before it (nimble_pack ... nimble_call_packed / nimble_call_records): a latch left by an earlier call whose
  // results nobody fetched must not fail this one
  launch_clear_call(s, cb, !c->skip_pack);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->ev[0], s));
  return NIMBLE_OK;
}

// Tail of a call, over all cb.n reads: class interning, dedup, count, histogram compaction.
int enqueue_tail(nimble_ctx *c) {
  hipStream_t s = c->stream;
  CallBuffers &cb = c->cb;
  HIPCHK(hipEventRecord(c->ev[2], s));
  // class interning, round 0: claim, then verify behind the kernel boundary.  Tag collisions (practically
  // never) leave reads unresolved; finish_call() then runs further rounds and redoes dedup + count.
  {
    int rc = enqueue_intern_round(c, 0);
    if (rc) return rc;
  }
  HIPCHK(hipEventRecord(c->ev[3], s));
  if (c->defer.active) {
    // the dedup verdicts come from the keys' owners: the call stays open until nimble_count_verdicts
    // (a re-enqueue after pool growth already has them)
    HIPCHK(hipGetLastError());
    return c->defer.counted ? enqueue_count_verdicts(c) : mark_done(c);
  }
  // Dedup, count and compaction may run on the index's side stream with a small grid: k_dedup is bound by the
  // chip's atomic rate, which 64 workgroups reach (tools/probes/atomic_rate_cus.hip), so the next call's pack and
  // align (another context, this launch stream) get the rest of the machine meanwhile.
  // Only when the index has a second context (calls in flight): a lone call just pays the hop to the other stream.
  const uint32_t aside_grid = c->tail_aside_grid;
  hipStream_t t = s;
  c->tail_aside = aside_grid != 0 && c->ix->n_ctx > 1 && !c->streaming && c->copy_stream != nullptr;
  if (c->tail_aside) {
    t = c->copy_stream;
    HIPCHK(hipStreamWaitEvent(t, c->ev[3], 0));
  }
  launch_dedup(t, c->prm, cb, c->tail_aside ? aside_grid : 0u);
  HIPCHK(hipEventRecord(c->ev[4], t));
  if (!cb.fuse_count) launch_count(t, cb);
  c->counted_marked = !cb.fuse_count;
  HIPCHK(hipEventRecord(c->ev[5], t));
  int rc = enqueue_compact(c, t);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  return mark_done(c, t);
}

// Enqueue the whole launch sequence of one call on the context's stream; nothing here waits for the GPU.
int enqueue_call(nimble_ctx *c) {
  hipStream_t s = c->stream;
  CallBuffers &cb = c->cb;
  int rc = enqueue_head(c);
  if (rc) return rc;
  if (!c->skip_pack) {
    if (c->in_words[0])
      launch_pack_words(s, c->in_words[0], c->in_len32[0], c->in_stride[0], c->in_words[1], c->in_len32[1], c->in_stride[1],
                        c->in_max_len, c->prm.min_read_length, c->b_plog.as<double>(), cb);
    else
      launch_pack(s, c->in_r[0], c->in_off[0], c->in_r[1], c->in_off[1], c->in_fixed_len, c->in_max_len,
                  c->prm.min_read_length, c->b_plog.as<double>(), c->plog_max_len, cb);
  }
  HIPCHK(hipEventRecord(c->ev[1], s));
  if (c->defer.active && !c->defer.routed) {
    rc = enqueue_route(c);
    if (rc) return rc;
  }
  launch_align(s, c->ix->dev, c->prm, cb, c->want_counters, c->align_grid_pct, c->stream_cus);
  return enqueue_tail(c);
}

int redo_dedup_count(nimble_ctx *c) {
  hipStream_t s = c->stream;
  if (c->cb.hot) HIPCHK(hipMemsetAsync(c->cb.hot, 0, (size_t)HOT_KEYS * 8, s));
  HIPCHK(hipMemsetAsync(c->b_dedup.p, 0, c->dslots * 8, s));
  HIPCHK(hipMemsetAsync(c->b_hist_keys.p, 0xFF, c->hist_slots * 8, s));
  HIPCHK(hipMemsetAsync(c->b_hist_cnt.p, 0, c->hist_slots * 8, s));
  if (c->cb.hist_rep) HIPCHK(hipMemsetAsync(c->cb.hist_rep, 0, c->hist_slots * 4, s));
  HIPCHK(hipMemsetAsync((uint64_t *)c->b_state.p + 10, 0, 8, s));
  HIPCHK(hipEventRecord(c->ev[3], s));
  if (c->defer.active) return enqueue_count_verdicts(c);
  launch_dedup(s, c->prm, c->cb);
  HIPCHK(hipEventRecord(c->ev[4], s));
  if (!c->cb.fuse_count) launch_count(s, c->cb);
  c->counted_marked = !c->cb.fuse_count;
  HIPCHK(hipEventRecord(c->ev[5], s));
  int rc = enqueue_compact(c);
  return rc ? rc : mark_done(c);
}

// Wait for the call and resolve the rare conditions that need host intervention (pool growth, intern tag
// collisions, histogram growth).  Idempotent; every getter goes through here.
int finish_call(nimble_ctx *c) {
  if (c->finished) return NIMBLE_OK;
  if (c->defer.active && !c->defer.counted)
    return fail(NIMBLE_E_INVALID, "the call defers its dedup: nimble_count_verdicts has not been called yet");
  for (;;) {
    int rc = fetch_state(c);
    if (rc) return rc;
    const uint64_t err = c->h_state[10];
    if (c->h_state[14] != 0) {
      // k_pack met offsets it could not trust (device-resident inputs are validated where they are read)
      HIPCHK(hipMemsetAsync((uint64_t *)c->b_state.p + 14, 0, 8, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      c->finished = true;
      c->h_c1.clear();
      c->h_c2.clear();
      c->h_cnt.clear();
      c->h_seg.clear();
      c->h_rep.clear();
      return fail(NIMBLE_E_INVALID, "offsets not monotone or a read longer than max_len (found on the device)");
    }
    if (err & ERR_SCRATCH) {
      if (c->attempt >= 3 || c->scratch_cap >= 0xFFFFFF00ULL)
        return fail(NIMBLE_E_OVERFLOW, "class scratch pool overflow (raise NIMBLE_SCRATCH_PER_READ)");
      c->attempt++;
      c->scratch_cap = std::min<uint64_t>(c->scratch_cap * 4, 0xFFFFFF00ULL);
      rc = c->b_scratch.ensure(c->scratch_cap * 4, &c->bytes);
      if (rc) return rc;
      c->cb.scratch = c->b_scratch.as<uint32_t>();
      c->cb.scratch_cap = (uint32_t)c->scratch_cap;
      rc = enqueue_call(c);
      if (rc) return rc;
      continue;
    }
    if (err & ERR_CLASS_CAP) return fail(NIMBLE_E_OVERFLOW, "dynamic class table full (raise NIMBLE_DYN_CLASSES)");
    if (err & ERR_IDS_CAP) return fail(NIMBLE_E_OVERFLOW, "dynamic class id pool full (raise NIMBLE_DYN_IDS)");
    if (c->h_state[9] != 0) {  // unresolved interns: probe further, then redo the stages that used class ids
      for (int round = 1; c->h_state[9] != 0; ++round) {
        if (round > 64) return fail(NIMBLE_E_INTERNAL, "class interning did not converge");
        HIPCHK(hipMemsetAsync((uint64_t *)c->b_state.p + 9, 0, 8, c->stream));
        rc = enqueue_intern_round(c, round);
        if (!rc) rc = mark_done(c);
        if (!rc) rc = fetch_state(c);
        if (rc) return rc;
        if (c->h_state[10] & (ERR_CLASS_CAP | ERR_IDS_CAP))
          return fail(NIMBLE_E_OVERFLOW, "dynamic class table full (raise NIMBLE_DYN_CLASSES / NIMBLE_DYN_IDS)");
      }
      rc = redo_dedup_count(c);
      if (rc) return rc;
      continue;
    }
    if (err & ERR_HIST) {
      if (c->hist_slots >= (1ULL << 30)) return fail(NIMBLE_E_OVERFLOW, "histogram table overflow");
      c->hist_slots *= 16;
      rc = c->b_hist_keys.ensure(c->hist_slots * 8, &c->bytes);
      if (!rc) rc = c->b_hist_cnt.ensure(c->hist_slots * 8, &c->bytes);
      if (rc) return rc;
      c->cb.hist_keys = c->b_hist_keys.as<uint64_t>();
      c->cb.hist_cnt = c->b_hist_cnt.as<uint64_t>();
      c->cb.hist_mask = c->hist_slots - 1;
      if (c->cb.hist_rep) {
        rc = c->b_hist_rep.ensure(c->hist_slots * 4, &c->bytes);
        if (rc) return rc;
        c->cb.hist_rep = c->b_hist_rep.as<uint32_t>();
      }
      rc = redo_dedup_count(c);
      if (rc) return rc;
      continue;
    }
    break;
  }
  // histogram entries to the host (sorted by class pair).  The dedup table stays as it is until the next call
  // clears it: nimble_read_records may still need it to mark the representatives.
  {
    const uint64_t ne = c->h_state[11];
    c->h_c1.resize(ne);
    c->h_c2.resize(ne);
    c->h_cnt.resize(ne);
    c->h_seg.resize(ne);
    c->h_rep.resize(ne);
    if (ne) {
      std::vector<uint32_t> a(ne), b(ne), sg(ne), rp(ne);
      std::vector<uint64_t> k(ne);
      if (c->out_pinned) {  // written by k_hist_compact into page-locked memory; fetch_state waited for it
        std::copy(c->hout.c1, c->hout.c1 + ne, a.begin());
        std::copy(c->hout.c2, c->hout.c2 + ne, b.begin());
        std::copy(c->hout.cnt, c->hout.cnt + ne, k.begin());
        std::copy(c->hout.seg, c->hout.seg + ne, sg.begin());
        std::copy(c->hout.rep, c->hout.rep + ne, rp.begin());
      } else {
        HIPCHK(hipMemcpyAsync(a.data(), c->b_out_c1.p, ne * 4, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(hipMemcpyAsync(b.data(), c->b_out_c2.p, ne * 4, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(hipMemcpyAsync(k.data(), c->b_out_cnt.p, ne * 8, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(hipMemcpyAsync(sg.data(), c->b_out_seg.p, ne * 4, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(hipMemcpyAsync(rp.data(), c->b_out_rep.p, ne * 4, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(hipStreamSynchronize(c->copy_stream));
      }
      // order by (segment, c1, c2): LSD radix sort of the entry indices, 16 bits per pass, skipping constant digits
      std::vector<uint32_t> idx(ne), tmp(ne);
      for (uint64_t i = 0; i < ne; ++i) idx[i] = (uint32_t)i;
      const std::vector<uint32_t> *cols[3] = {&b, &a, &sg};  // least significant first
      std::vector<uint32_t> hist(65536);
      for (int col = 0; col < 3; ++col) {
        const std::vector<uint32_t> &v = *cols[col];
        for (int shift = 0; shift < 32; shift += 16) {
          std::fill(hist.begin(), hist.end(), 0u);
          for (uint64_t i = 0; i < ne; ++i) hist[(v[i] >> shift) & 0xFFFFu]++;
          if (hist[(v[0] >> shift) & 0xFFFFu] == ne) continue;  // every entry has the same digit
          uint32_t run = 0;
          for (uint32_t d = 0; d < 65536; ++d) {
            const uint32_t c0 = hist[d];
            hist[d] = run;
            run += c0;
          }
          for (uint64_t i = 0; i < ne; ++i) tmp[hist[(v[idx[i]] >> shift) & 0xFFFFu]++] = idx[i];
          idx.swap(tmp);
        }
      }
      for (uint64_t i = 0; i < ne; ++i) {
        c->h_c1[i] = a[idx[i]];
        c->h_c2[i] = b[idx[i]];
        c->h_cnt[i] = k[idx[i]];
        c->h_seg[i] = sg[idx[i]];
        c->h_rep[i] = rp[idx[i]];
      }
    }
  }
  c->dyn_before = c->p_dyn[0];  // copied at the head of the call, ahead of everything fetch_state waited for
  c->finished = true;
  return NIMBLE_OK;
}

}  // namespace

extern "C" {

int nimble_abi_version(void) { return NIMBLE_ABI_VERSION; }
const char *nimble_last_error(void) { return g_err.c_str(); }

int nimble_device_count(int *count) {
  if (!count) return fail(NIMBLE_E_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(NIMBLE_E_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return NIMBLE_OK;
}

static int index_build_impl(const uint8_t *seqs, const uint64_t *seq_off, uint32_t n_seqs, int device, nimble_index **out) {
  if (!out || (!seqs && n_seqs) || !seq_off) return fail(NIMBLE_E_INVALID, "nimble_index_build: NULL argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NIMBLE_E_NO_DEVICE, "nimble_index_build: no HIP device (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(NIMBLE_E_INVALID, "nimble_index_build: bad device ordinal");
  HIPCHK(hipSetDevice(device));
  FlatIndex fi;
  build_flat_index(seqs, seq_off, n_seqs, fi);
  nimble_index *ix = new (std::nothrow) nimble_index();
  if (!ix) return fail(NIMBLE_E_NOMEM, "out of memory");
  struct Guard {  // (whatever leaves this function early, by return or by exception, takes the half-built index with it)
    nimble_index *p;
    ~Guard() { delete p; }
  } guard{ix};
  ix->device = device;
  ix->n_kmers = fi.n_kmers;
  ix->n_nodes = fi.n_nodes;
  ix->n_static = fi.n_colours;
  ix->unitig_bases = fi.unitig_bases;
  ix->static_entries = fi.col_ids.size();
  ix->ht_slots = fi.ht_slots;
  const uint64_t dyn_classes = env_u64("NIMBLE_DYN_CLASSES", 1ULL << 20);
  const uint64_t dyn_ids = env_u64("NIMBLE_DYN_IDS", 1ULL << 26);
  const uint64_t cls_cap = fi.n_colours + dyn_classes;
  const uint64_t ids_cap = fi.col_ids.size() + dyn_ids;
  if (cls_cap >= 0xFFFFFFF0ULL || ids_cap >= 0xFFFFFFF0ULL) return fail(NIMBLE_E_INVALID, "class table capacity exceeds 32 bits");
  int rc = NIMBLE_OK;
  auto up = [&](auto &buf, const auto &vec, size_t min_elems = 0) {
    if (rc == NIMBLE_OK) rc = upload(buf, vec, &ix->device_bytes, min_elems);
  };
  // An index with stretch records names a k-mer's unitig in the dictionary by the unitig's FIRST RECORD (the fast walk then
  // starts without asking srec_first -- one gather per read less); the general walk, which wants the unitig, asks
  // srec_node (kernels.hip seed_node).  Decided here, where the dictionary is uploaded.
  static const bool use_mleft_ = env_u64("NIMBLE_LOCAL_RESEED", 1) != 0;
  static const bool use_srec_ = env_u64("NIMBLE_FAST_ALIGN", 1) != 0;
  const bool have_srec = use_srec_ && !fi.srec.empty() && use_mleft_ && !fi.mleft.empty();
  std::vector<uint32_t> srec_node;
  if (have_srec) {
    srec_node.assign(fi.srec.size() / 8, 0);
    for (size_t nd = 0; nd < fi.n_nodes; ++nd) {
      const uint32_t first = fi.srec_first[nd];
      const uint32_t inf = (fi.node_rec[nd * 16] & 0xFFFFFFu) - KMER;
      const uint32_t n_rec = std::max<uint32_t>(1u, (inf + 31u) / 32u);
      for (uint32_t j = 0; j < n_rec; ++j) srec_node[first + j] = (uint32_t)nd;
    }
    for (uint64_t sl = 0; sl < fi.ht_slots; ++sl)
      if (fi.ht[2 * sl] != HT_EMPTY) {
        const uint64_t v = fi.ht[2 * sl + 1];
        fi.ht[2 * sl + 1] = ((uint64_t)fi.srec_first[(uint32_t)(v >> 32)] << 32) | (uint32_t)v;
      }
  }
  up(ix->b_ht, fi.ht);
  up(ix->b_bitmap, fi.bitmap);
  static const bool use_l1 = env_u64("NIMBLE_FILTER_L1", 1) != 0;
  if (use_l1 && !fi.l1.empty()) up(ix->b_l1, fi.l1);
  static const bool use_mleft = env_u64("NIMBLE_LOCAL_RESEED", 1) != 0;
  if (use_mleft && !fi.mleft.empty()) up(ix->b_mleft, fi.mleft);
  up(ix->b_rec, fi.node_rec);
  if (have_srec) {
    up(ix->b_srec, fi.srec);
    up(ix->b_srec_first, srec_node);  // (record -> unitig: the dictionary holds records)
    up(ix->b_srec_base, fi.srec_base);
    up(ix->b_srec_many, fi.srec_many);
  }
  up(ix->b_ledge, fi.node_ledge);
  up(ix->b_unitig, fi.unitig);
  std::vector<uint32_t> len(fi.n_colours);
  for (size_t c = 0; c < fi.n_colours; ++c) len[c] = fi.col_off[c + 1] - fi.col_off[c];
  up(ix->b_cls_desc, fi.cls_desc, cls_cap * 4);
  std::vector<uint32_t> coff(fi.col_off.begin(), fi.col_off.end() - 1);
  up(ix->b_cls_off, coff, cls_cap);
  up(ix->b_cls_ids, fi.col_ids, ids_cap);
  up(ix->b_cls_bits, fi.cls_bits, 2);
  // intern table seeded with the static classes, so that an intersection equal to a k-mer colour
  // resolves to that colour's id (class ids are canonical by content)
  const uint64_t islots = pow2_at_least(4 * cls_cap);
  std::vector<uint64_t> intern(islots, 0);
  for (size_t c = 0; c < fi.n_colours; ++c) {
    const uint64_t h = class_hash_of_ids(fi.col_ids.data() + fi.col_off[c], len[c]);
    uint64_t pos = h & (islots - 1);
    while (intern[pos] != 0) pos = (pos + 1) & (islots - 1);
    intern[pos] = ((uint64_t)intern_tag(h) << 32) | (uint32_t)c;
  }
  up(ix->b_intern, intern);
  std::vector<uint32_t> dyn_state = {(uint32_t)fi.n_colours, (uint32_t)fi.col_ids.size(), 0, 0};
  up(ix->b_dyn_state, dyn_state);
  if (rc != NIMBLE_OK) return rc;
  ix->h_col_off = std::move(fi.col_off);
  ix->h_col_ids = std::move(fi.col_ids);
  DevIndex &d = ix->dev;
  d.ht = ix->b_ht.as<uint4>();
  d.ht_mask = fi.ht_slots - 1;
  d.ht_log2 = fi.ht_log2;
  d.bitmap = ix->b_bitmap.as<uint4>();
  d.bm_lines_log2 = fi.bm_lines_log2;
  d.l1 = ix->b_l1.p ? ix->b_l1.as<uint32_t>() : nullptr;
  d.mleft = ix->b_mleft.p ? ix->b_mleft.as<uint64_t>() : nullptr;
  d.mleft_log2 = fi.mleft_log2;
  d.node_rec = ix->b_rec.as<uint4>();
  d.node_ledge = ix->b_ledge.as<uint4>();
  d.srec = have_srec ? ix->b_srec.as<uint4>() : nullptr;
  d.srec_node = have_srec ? ix->b_srec_first.as<uint32_t>() : nullptr;
  d.srec_base = have_srec ? ix->b_srec_base.as<uint32_t>() : nullptr;
  d.srec_many = have_srec ? ix->b_srec_many.as<uint4>() : nullptr;
  d.unitig = ix->b_unitig.as<uint64_t>();
  d.all_local = fi.all_classes_local ? 1u : 0u;
  d.uniform_windows = (fi.uniform_windows && env_u64("NIMBLE_UNIFORM_WINDOWS", 1) != 0) ? 1u : 0u;
  d.all_bitmaps = (fi.all_wide_have_bitmaps && env_u64("NIMBLE_WIDE_WINDOW", 1) != 0) ? 1u : 0u;
  // bitmaps longer than the 256-row register window (allele families of several hundred rows): the window moves to LDS, as
  // many words per lane as the longest bitmap has (at most NIMBLE_LDS_WINDOW_WORDS, default 32 = 2048 rows = 64 KiB per
  // block; classes beyond that keep the colour list); NIMBLE_LDS_WINDOW=0 switches it off
  d.window_words = 0;
  if (d.all_bitmaps && fi.max_bitmap_words > 4 && env_u64("NIMBLE_LDS_WINDOW", 1) != 0)
    d.window_words = (uint32_t)std::min<uint64_t>(fi.max_bitmap_words, std::min<uint64_t>(env_u64("NIMBLE_LDS_WINDOW_WORDS", 32), 64));  // (64 words = 128 KiB per block: the most that leaves room for any key)
  d.cls_desc = ix->b_cls_desc.as<uint4>();
  d.cls_off = ix->b_cls_off.as<uint32_t>();
  d.cls_ids = ix->b_cls_ids.as<uint32_t>();
  d.cls_bits = ix->b_cls_bits.as<uint64_t>();
  d.n_static = (uint32_t)fi.n_colours;
  d.cls_cap = (uint32_t)cls_cap;
  d.ids_cap = (uint32_t)ids_cap;
  d.intern = ix->b_intern.as<uint64_t>();
  d.intern_mask = islots - 1;
  d.dyn_state = ix->b_dyn_state.as<uint32_t>();
  guard.p = nullptr;
  *out = ix;
  return NIMBLE_OK;
}

// (the whole build sits inside the try: the host vectors of the upload and of the intern table allocate too, and nothing
// may be thrown across the C ABI)
int nimble_index_build(const uint8_t *seqs, const uint64_t *seq_off, uint32_t n_seqs, int device, nimble_index **out) {
  try {
    return index_build_impl(seqs, seq_off, n_seqs, device, out);
  } catch (const std::bad_alloc &) {
    return fail(NIMBLE_E_NOMEM, "nimble_index_build: out of host memory");
  } catch (const std::exception &e) {
    return fail(NIMBLE_E_INTERNAL, e.what());
  } catch (...) {
    return fail(NIMBLE_E_INTERNAL, "nimble_index_build: unknown failure");
  }
}

int nimble_flat_index_selfcheck(const uint8_t *seqs, const uint64_t *seq_off, uint32_t n_seqs, uint64_t *n_records) {
  if ((!seqs && n_seqs) || !seq_off) return fail(NIMBLE_E_INVALID, "nimble_flat_index_selfcheck: NULL argument");
  try {
    FlatIndex fi;
    build_flat_index(seqs, seq_off, n_seqs, fi);
    if (n_records) *n_records = fi.srec.size() / 8;
    const std::string what = check_stretch_records(fi);
    if (!what.empty()) return fail(NIMBLE_E_INTERNAL, "stretch records: " + what);
  } catch (const std::exception &e) {
    return fail(NIMBLE_E_INTERNAL, e.what());
  } catch (...) {
    return fail(NIMBLE_E_INTERNAL, "nimble_flat_index_selfcheck: unknown failure");
  }
  return NIMBLE_OK;
}

int nimble_flat_index_stats(const uint8_t *seqs, const uint64_t *seq_off, uint32_t n_seqs, uint64_t s[5]) {
  if ((!seqs && n_seqs) || !seq_off || !s) return fail(NIMBLE_E_INVALID, "nimble_flat_index_stats: NULL argument");
  try {
    FlatIndex fi;
    build_flat_index(seqs, seq_off, n_seqs, fi);
    s[0] = fi.n_kmers;
    s[1] = fi.n_nodes;
    s[2] = fi.n_colours;
    s[3] = fi.unitig_bases;
    s[4] = fi.col_ids.size();
  } catch (const std::exception &e) {
    return fail(NIMBLE_E_INTERNAL, e.what());
  }
  return NIMBLE_OK;
}

void nimble_index_free(nimble_index *ix) {
  if (!ix) return;
  {
    // contexts keep a pointer to their index: an index freed first (a garbage collector picks its own order) stays
    // until its last context has gone
    std::lock_guard<std::mutex> lock(ix->intern_mu);
    if (ix->n_ctx > 0) {
      ix->released = true;
      return;
    }
  }
  (void)hipSetDevice(ix->device);
  delete ix;
}

int nimble_index_stats(const nimble_index *ix, uint64_t s[8]) {
  if (!ix || !s) return fail(NIMBLE_E_INVALID, "nimble_index_stats: NULL argument");
  s[0] = ix->n_kmers;
  s[1] = ix->n_nodes;
  s[2] = ix->n_static;
  s[3] = ix->unitig_bases;
  s[4] = ix->static_entries;
  s[5] = ix->ht_slots;
  s[6] = ix->device_bytes;
  uint32_t st[4] = {0, 0, 0, 0};
  HIPCHK(hipSetDevice(ix->device));
  HIPCHK(hipMemcpy(st, ix->b_dyn_state.p, sizeof(st), hipMemcpyDeviceToHost));
  s[7] = st[0] - ix->n_static;
  return NIMBLE_OK;
}

int nimble_class_get(const nimble_index *ix, uint32_t id, uint32_t *ids, uint32_t cap, uint32_t *len) {
  if (!ix || !len) return fail(NIMBLE_E_INVALID, "nimble_class_get: NULL argument");
  if (id < ix->n_static) {
    uint32_t o = ix->h_col_off[id], l = ix->h_col_off[id + 1] - o;
    *len = l;
    if (ids) memcpy(ids, ix->h_col_ids.data() + o, sizeof(uint32_t) * std::min(l, cap));
    return NIMBLE_OK;
  }
  HIPCHK(hipSetDevice(ix->device));
  uint32_t st[4];
  HIPCHK(hipMemcpy(st, ix->b_dyn_state.p, sizeof(st), hipMemcpyDeviceToHost));
  if (id >= st[0] || id >= ix->dev.cls_cap) return fail(NIMBLE_E_INVALID, "nimble_class_get: unknown class id");
  uint32_t desc0 = 0, o = 0;
  HIPCHK(hipMemcpy(&desc0, ix->dev.cls_desc + (size_t)id, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&o, ix->dev.cls_off + id, 4, hipMemcpyDeviceToHost));
  const uint32_t l = desc0 & ~CLS_MASK_FLAG;
  *len = l;
  if (ids && l && cap)
    HIPCHK(hipMemcpy(ids, ix->dev.cls_ids + o, sizeof(uint32_t) * std::min(l, cap), hipMemcpyDeviceToHost));
  return NIMBLE_OK;
}

int nimble_class_table_read(const nimble_index *ix, uint32_t first, uint32_t count, uint32_t *len, uint32_t *pool_off) {
  if (!ix || (count && (!len || !pool_off))) return fail(NIMBLE_E_INVALID, "nimble_class_table_read: NULL argument");
  if ((uint64_t)first + count > ix->dev.cls_cap) return fail(NIMBLE_E_INVALID, "nimble_class_table_read: beyond the class table");
  if (count == 0) return NIMBLE_OK;
  HIPCHK(hipSetDevice(ix->device));
  std::vector<uint32_t> desc((size_t)count * 4);
  HIPCHK(hipMemcpy(desc.data(), ix->dev.cls_desc + first, (size_t)count * 16, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(pool_off, ix->dev.cls_off + first, (size_t)count * 4, hipMemcpyDeviceToHost));
  for (uint32_t k = 0; k < count; ++k) len[k] = desc[(size_t)k * 4] & ~CLS_MASK_FLAG;
  return NIMBLE_OK;
}

int nimble_class_pool_read(const nimble_index *ix, uint32_t pool_off, uint32_t count, uint32_t *ids) {
  if (!ix || (count && !ids)) return fail(NIMBLE_E_INVALID, "nimble_class_pool_read: NULL argument");
  if ((uint64_t)pool_off + count > ix->dev.ids_cap) return fail(NIMBLE_E_INVALID, "nimble_class_pool_read: beyond the id pool");
  if (count == 0) return NIMBLE_OK;
  HIPCHK(hipSetDevice(ix->device));
  HIPCHK(hipMemcpy(ids, ix->dev.cls_ids + pool_off, (size_t)count * 4, hipMemcpyDeviceToHost));
  return NIMBLE_OK;
}

int nimble_ctx_create(nimble_index *ix, void *stream, nimble_ctx **out) {
  if (!ix || !out) return fail(NIMBLE_E_INVALID, "nimble_ctx_create: NULL argument");
  *out = nullptr;
  HIPCHK(hipSetDevice(ix->device));
  nimble_ctx *c = new (std::nothrow) nimble_ctx();
  if (!c) return fail(NIMBLE_E_NOMEM, "out of memory");
  c->ix = ix;
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return fail(NIMBLE_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    c->own_stream = true;
  }
  for (auto &e : c->ev) {
    if (hipEventCreate(&e) != hipSuccess) {
      delete c;
      return fail(NIMBLE_E_HIP, "hipEventCreate failed");
    }
  }
  c->have_events = true;
  if (!ix->copy_stream && hipStreamCreateWithFlags(&ix->copy_stream, hipStreamNonBlocking) != hipSuccess)
    ix->copy_stream = nullptr;
  c->copy_stream = ix->copy_stream;
  if (!c->copy_stream ||
      hipHostMalloc((void **)&c->p_state, 16 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void **)&c->p_dyn, 4 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
    delete c;
    return fail(NIMBLE_E_HIP, "nimble_ctx_create: side stream / pinned buffers");
  }
  c->want_counters = (int)env_u64("NIMBLE_COUNTERS", 0);
  c->tail_aside_grid = (uint32_t)std::min<uint64_t>(env_u64("NIMBLE_DEDUP_ASIDE", 1280), 1u << 20);
  {
    std::lock_guard<std::mutex> lock(ix->intern_mu);
    ix->n_ctx++;
  }
  *out = c;
  return NIMBLE_OK;
}

void nimble_ctx_free(nimble_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->ix->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  {
    // the interning this context enqueued has finished; nobody may wait on an event of a stream that is about to go
    std::lock_guard<std::mutex> lock(c->ix->intern_mu);
    if (c->ix->intern_last == c->stream) {
      c->ix->intern_chained = false;
      c->ix->intern_last = nullptr;
    }
  }
  nimble_index *ix = c->ix;
  delete c;
  bool last = false;
  {
    std::lock_guard<std::mutex> lock(ix->intern_mu);
    last = --ix->n_ctx == 0 && ix->released;
  }
  if (last) delete ix;
}

void *nimble_ctx_stream(nimble_ctx *c) { return c ? (void *)c->stream : nullptr; }

int nimble_ctx_set_option(nimble_ctx *c, int option, int64_t value) {
  if (!c) return fail(NIMBLE_E_INVALID, "NULL context");
  if (option == NIMBLE_OPT_COUNTERS) {
    c->want_counters = value != 0;
    return NIMBLE_OK;
  }
  if (option == NIMBLE_OPT_ALIGN_GRID_PCT) {
    if (value < 10 || value > 100) return fail(NIMBLE_E_INVALID, "NIMBLE_OPT_ALIGN_GRID_PCT: 10..100");
    c->align_grid_pct = (int)value;
    return NIMBLE_OK;
  }
  if (option == NIMBLE_OPT_TAIL_ASIDE) {
    if (value < 0 || value > (1 << 20)) return fail(NIMBLE_E_INVALID, "NIMBLE_OPT_TAIL_ASIDE: 0..1048576 workgroups");
    c->tail_aside_grid = (uint32_t)value;
    return NIMBLE_OK;
  }
  return fail(NIMBLE_E_INVALID, "nimble_ctx_set_option: unknown option");
}

int nimble_ctx_synchronize(nimble_ctx *c) {
  DRAIN_STALE_HIP_ERROR();
  if (!c) return fail(NIMBLE_E_INVALID, "NULL context");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NIMBLE_OK;
}

// stage the read buffers on the device when they are handed over as host memory
static int stage_inputs(nimble_ctx *c, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                        const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem) {
  if (mem == NIMBLE_MEM_HOST_PINNED) mem = NIMBLE_MEM_HOST;  // outside a stream the copy is simply waited for
  c->in_r[0] = r1;
  c->in_r[1] = r2;
  c->in_off[0] = r1_off;
  c->in_off[1] = r2_off;
  c->in_fixed_len = fixed_len;
  c->in_max_len = max_len;
  c->in_words[0] = c->in_words[1] = nullptr;
  if (mem != NIMBLE_MEM_HOST) return NIMBLE_OK;
  const int nm = r2 ? 2 : 1;
  for (int m = 0; m < nm; ++m) {
    const uint8_t *src = m ? r2 : r1;
    const uint64_t *off = m ? r2_off : r1_off;
    uint64_t bytes = off ? off[n] : n * (uint64_t)fixed_len;
    if (off) {
      for (uint64_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i] || off[i + 1] - off[i] > max_len)
          return fail(NIMBLE_E_INVALID, "nimble_call: offsets not monotone or a read longer than max_len");
    }
    int rc = c->b_in[m].ensure(std::max<uint64_t>(bytes, 16), &c->bytes);
    if (rc) return rc;
    // (offsets are absolute: a slice of a larger batch is the same buffer with a later offsets pointer, and only the
    // bytes its reads cover travel)
    const uint64_t lo = off && n ? off[0] : 0;
    if (bytes > lo) HIPCHK(hipMemcpyAsync(c->b_in[m].as<uint8_t>() + lo, src + lo, bytes - lo, hipMemcpyHostToDevice, c->stream));
    c->in_r[m] = c->b_in[m].as<uint8_t>();
    if (off) {
      rc = c->b_in_off[m].ensure((n + 1) * 8, &c->bytes);
      if (rc) return rc;
      HIPCHK(hipMemcpyAsync(c->b_in_off[m].p, off, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
      c->in_off[m] = c->b_in_off[m].as<uint64_t>();
    }
  }
  return NIMBLE_OK;
}

static int check_read_args(const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2, const uint64_t *r2_off,
                           uint64_t n, uint32_t fixed_len, uint32_t &max_len, int mem) {
  if (n && !r1) return fail(NIMBLE_E_INVALID, "nimble_call: r1 is NULL");
  if (!r1_off && fixed_len == 0 && n) return fail(NIMBLE_E_INVALID, "nimble_call: neither offsets nor fixed_len given");
  if (r2 && ((r1_off == nullptr) != (r2_off == nullptr)))
    return fail(NIMBLE_E_INVALID, "nimble_call: R1 and R2 must both use offsets or both be fixed length");
  if (n >= 0xFFFFFFF0ULL) return fail(NIMBLE_E_INVALID, "nimble_call: more than 2^32 reads in one call");
  if (!r1_off && fixed_len > max_len) max_len = fixed_len;
  if (max_len == 0) max_len = 1;
  if (max_len > 65535) return fail(NIMBLE_E_INVALID, "nimble_call: max_len above 65535 is not supported");
  if (mem != NIMBLE_MEM_HOST && mem != NIMBLE_MEM_DEVICE && mem != NIMBLE_MEM_HOST_PINNED)
    return fail(NIMBLE_E_INVALID, "nimble_call: bad mem");
  return NIMBLE_OK;
}

// Buffers and tables of one call over n reads.  `ext` (may be NULL) supplies caller-owned device arrays for the
// packed form (keys, lengths, key hash, prefilter verdicts) instead of the context's own.
static int setup_call(nimble_ctx *c, const nimble_align_params *p, uint64_t n, bool paired, uint32_t max_len,
                      const nimble_packed *ext, const uint64_t *records = nullptr) {
  const int nm = paired ? 2 : 1;
  const uint32_t kw = (max_len * (uint32_t)nm + 31u) / 32u;
  // the walk keeps every key of a tile in LDS; one workgroup may take up to 160 KiB on gfx950
  // (an index with wide allele families adds its per-lane row window: DevIndex.window_words words of every lane)
  const size_t window_bytes = c->ix->dev.all_local ? 0 : (size_t)c->ix->dev.window_words * 256 * 8;
  if ((size_t)(kw + 1) * 256 * 8 + (size_t)align_lds_cols() * 256 * 4 + 4096 + window_bytes > 160 * 1024)
    return fail(NIMBLE_E_INVALID, window_bytes ? "nimble_call: reads too long for the LDS-resident walk beside this index's row "
                                                 "window (set NIMBLE_LDS_WINDOW_WORDS lower, or NIMBLE_LDS_WINDOW=0)"
                                               : "nimble_call: reads too long for the LDS-resident walk (max_len * mates > ~2400)");
  if (ext && ext->key_words != kw) return fail(NIMBLE_E_INVALID, "packed buffers: key_words does not match max_len");
  CallBuffers &cb = c->cb;
  cb.n = n;
  cb.key_stride = n;
  cb.key_words = kw;
  cb.paired = paired ? 1 : 0;
  cb.fuse_count = paired ? 0u : 1u;  // single-end: classes are a function of the dedup key (see k_dedup)
  const size_t nn = std::max<uint64_t>(n, 1);
  int rc = NIMBLE_OK;
  auto need = [&](DevBuf &b, size_t bytes) {
    if (rc == NIMBLE_OK) rc = b.ensure(bytes, &c->bytes);
  };
  if (!ext && !records) {
    need(c->b_keys, nn * kw * 8);
    need(c->b_hash, nn * 8);
  }
  need(c->b_slot, nn * 4);
  need(c->b_counted, nn);
  need(c->b_tile_ctr, (size_t)TILE_COUNTERS * TILE_COUNTER_STRIDE);
  for (int m = 0; m < 2; ++m) {
    if (!records && (!ext || (m == 1 && !ext->len[1]))) {
      need(c->b_len[m], nn * 4);
      need(c->b_pre[m], nn);
    }
    need(c->b_reason[m], nn);
    need(c->b_score[m], nn * 4);
    need(c->b_mism[m], nn * 4);
    need(c->b_cls[m], nn * 4);
    need(c->b_dyn_off[m], nn * 4);
    need(c->b_dyn_len[m], nn * 4);
    need(c->b_dyn_hash[m], nn * 8);
    need(c->b_dyn_pos[m], nn * 4);
  }
  if (c->scratch_cap == 0) c->scratch_cap = std::max<uint64_t>(1ULL << 20, env_u64("NIMBLE_SCRATCH_PER_READ", 4) * n);
  c->scratch_cap = std::min<uint64_t>(std::max<uint64_t>(c->scratch_cap, env_u64("NIMBLE_SCRATCH_PER_READ", 4) * n),
                                      0xFFFFFF00ULL);
  need(c->b_scratch, c->scratch_cap * 4);
  // spill area of the visited-colour lists: only an index with classes outside the 64-row mask form keeps lists
  const uint32_t ws_rows = (!c->ix->dev.all_local && max_len > align_lds_cols()) ? max_len - align_lds_cols() : 0;
  need(c->b_ws, std::max<size_t>((size_t)ws_rows * align_ws_lanes() * 4, 16));
  // load factor <= 2/3 even if every read is kept (NIMBLE_DEDUP_SLOTS_PCT: slots per 100 reads)
  static const uint64_t slots_pct = std::min<uint64_t>(std::max<uint64_t>(env_u64("NIMBLE_DEDUP_SLOTS_PCT", 150), 110), 400);
  const uint64_t dslots = std::min<uint64_t>(std::max<uint64_t>(n * slots_pct / 100, 1024), 0xFFFFFFF0ULL);
  {
    const void *before = c->b_dedup.p;
    need(c->b_dedup, dslots * 8);
    if (c->b_dedup.p != before) c->dedup_clean_slots = 0;
  }
  // (class pair) histogram: 256 k slots to start with (a 10 M-read call has 5-40 k entries); grows 16x on overflow
  if (c->hist_slots == 0) c->hist_slots = pow2_at_least(env_u64("NIMBLE_HIST_SLOTS", 1ULL << 18));
  need(c->b_hist_keys, c->hist_slots * 8);
  need(c->b_hist_cnt, c->hist_slots * 8);
  {
    const void *before = c->b_state.p;
    need(c->b_state, 16 * 8);
    if (rc == NIMBLE_OK && c->b_state.p != before) HIPCHK(hipMemsetAsync(c->b_state.p, 0, 16 * 8, c->stream));
  }
  static const bool hot_keys = env_u64("NIMBLE_HOT_KEYS", 1) != 0;
  if (hot_keys) need(c->b_hot, (size_t)HOT_KEYS * 8);
  if (rc != NIMBLE_OK) return rc;
  rc = ensure_plog(c, max_len);
  if (rc != NIMBLE_OK) return rc;
  rc = ensure_min_cov(c, p->score_percent, max_len);
  if (rc != NIMBLE_OK) return rc;

  cb.rec = records;  // the call reads keys, lengths, prefilter verdicts and hashes from exchange records
  cb.rec_words = kw + 2;
  cb.keys = records ? nullptr : ext ? ext->keys : c->b_keys.as<uint64_t>();
  cb.key_hash = records ? nullptr : ext ? ext->hash : c->b_hash.as<uint64_t>();
  cb.slot = c->b_slot.as<uint32_t>();
  cb.counted = c->b_counted.as<uint8_t>();
  for (int m = 0; m < 2; ++m) {
    const bool use_ext = ext && ext->len[m];
    cb.len[m] = records ? nullptr : use_ext ? ext->len[m] : c->b_len[m].as<uint32_t>();
    cb.pre[m] = records ? nullptr : use_ext ? ext->pre[m] : c->b_pre[m].as<uint8_t>();
    cb.reason[m] = c->b_reason[m].as<uint8_t>();
    cb.score[m] = c->b_score[m].as<uint32_t>();
    cb.mism[m] = c->b_mism[m].as<uint32_t>();
    cb.cls[m] = c->b_cls[m].as<uint32_t>();
    cb.dyn_off[m] = c->b_dyn_off[m].as<uint32_t>();
    cb.dyn_len[m] = c->b_dyn_len[m].as<uint32_t>();
    cb.dyn_hash[m] = c->b_dyn_hash[m].as<uint64_t>();
    cb.dyn_pos[m] = c->b_dyn_pos[m].as<uint32_t>();
  }
  for (int m = 0; m < 2; ++m) {
    cb.alen[m] = cb.len[m];  // aligned length == read length unless nimble_call_ex trims for quality
    cb.skip[m] = nullptr;
  }
  cb.seg = nullptr;
  cb.cls_bits = 0;
  cb.hist_rep = nullptr;
  cb.min_cov = c->b_min_cov.as<uint32_t>();
  cb.scratch = c->b_scratch.as<uint32_t>();
  cb.scratch_cap = (uint32_t)c->scratch_cap;
  cb.ws_cols = c->b_ws.as<uint32_t>();
  cb.ws_rows = ws_rows;
  cb.ws_lanes = align_ws_lanes();
  cb.dedup = c->b_dedup.as<uint64_t>();
  cb.hot = c->b_hot.p ? c->b_hot.as<uint64_t>() : nullptr;
  cb.dedup_slots = (uint32_t)dslots;
  cb.hist_keys = c->b_hist_keys.as<uint64_t>();
  cb.hist_cnt = c->b_hist_cnt.as<uint64_t>();
  cb.hist_mask = c->hist_slots - 1;
  cb.state = c->b_state.as<uint64_t>();
  cb.tile_ctr = c->b_tile_ctr.as<uint64_t>();
  c->prm = *p;
  if (c->prm.min_read_length == 0) c->prm.min_read_length = 40;
  c->dslots = dslots;
  return NIMBLE_OK;
}

static int start_call(nimble_ctx *c) {
  // number of interned classes before this call (for the counters); read back without waiting
  HIPCHK(hipMemcpyAsync(c->p_dyn, c->ix->b_dyn_state.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  c->finished = false;
  c->attempt = 0;
  c->defer.active = c->defer.world != 0;
  c->defer.routed = c->defer.counted = false;
  c->defer.verdict = nullptr;
  if (c->defer.active) {
    if (c->streaming || c->cb.seg || !c->cb.fuse_count) {
      c->defer.active = false;
      c->defer.world = 0;
      return fail(NIMBLE_E_INVALID, "deferred dedup needs a plain call whose classes follow from the key "
                                    "(single-end or fixed-length mates)");
    }
    c->defer.world_of_call = c->defer.world;
    c->defer.world = 0;  // one call per arming
  }
  int rc = enqueue_call(c);
  if (rc) return rc;
  c->called = true;
  return NIMBLE_OK;
}

int nimble_call(nimble_ctx *c, const nimble_align_params *p, const uint8_t *r1, const uint64_t *r1_off,
                const uint8_t *r2, const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !p) return fail(NIMBLE_E_INVALID, "nimble_call: NULL argument");
  int rc = check_read_args(r1, r1_off, r2, r2_off, n, fixed_len, max_len, mem);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->ix->device));
  rc = stage_inputs(c, r1, r1_off, r2, r2_off, n, fixed_len, max_len, mem);
  if (rc) return rc;
  rc = setup_call(c, p, n, r2 != nullptr, max_len, nullptr);
  if (rc) return rc;
  // classes are a function of the dedup key when the key fixes where R1 ends: single-end, or fixed-length mates
  c->cb.fuse_count = (!r2 || !r1_off) ? 1u : 0u;
  c->skip_pack = false;
  return start_call(c);
}

int nimble_call_words(nimble_ctx *c, const nimble_align_params *p, const uint64_t *r1_words, const uint32_t *r1_len,
                      uint32_t r1_stride, const uint64_t *r2_words, const uint32_t *r2_len, uint32_t r2_stride, uint64_t n,
                      uint32_t max_len, int mem) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !p) return fail(NIMBLE_E_INVALID, "nimble_call_words: NULL argument");
  if (mem == NIMBLE_MEM_HOST_PINNED) mem = NIMBLE_MEM_HOST;
  if (mem != NIMBLE_MEM_HOST && mem != NIMBLE_MEM_DEVICE) return fail(NIMBLE_E_INVALID, "nimble_call_words: bad mem");
  const bool paired = r2_words != nullptr;
  if ((r2_len != nullptr) != paired) return fail(NIMBLE_E_INVALID, "nimble_call_words: mate words and mate lengths go together");
  if (n && (!r1_words || !r1_len || r1_stride == 0 || (paired && r2_stride == 0)))
    return fail(NIMBLE_E_INVALID, "nimble_call_words: NULL buffer or zero stride");
  if (n >= 0xFFFFFFF0ULL) return fail(NIMBLE_E_INVALID, "nimble_call_words: more than 2^32 reads in one call");
  if (max_len == 0) max_len = 1;
  if (max_len > 65535) return fail(NIMBLE_E_INVALID, "nimble_call_words: max_len above 65535 is not supported");
  HIPCHK(hipSetDevice(c->ix->device));
  c->in_r[0] = c->in_r[1] = nullptr;
  c->in_off[0] = c->in_off[1] = nullptr;
  c->in_fixed_len = 0;
  c->in_max_len = max_len;
  const uint64_t *w[2] = {r1_words, r2_words};
  const uint32_t *l[2] = {r1_len, r2_len};
  const uint32_t st[2] = {r1_stride, r2_stride};
  for (int m = 0; m < (paired ? 2 : 1); ++m) {
    if (mem == NIMBLE_MEM_HOST) {
      const uint64_t cap = std::min<uint64_t>(max_len, 32ULL * st[m]);
      for (uint64_t i = 0; i < n; ++i)
        if (l[m][i] > cap) return fail(NIMBLE_E_INVALID, "nimble_call_words: a read longer than max_len or than its words");
      int rc = c->b_in[m].ensure(std::max<uint64_t>(n * st[m] * 8, 16), &c->bytes);
      if (rc) return rc;
      rc = c->b_in_off[m].ensure(std::max<uint64_t>(n * 4, 16), &c->bytes);
      if (rc) return rc;
      if (n) {
        HIPCHK(hipMemcpyAsync(c->b_in[m].p, w[m], n * st[m] * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->b_in_off[m].p, l[m], n * 4, hipMemcpyHostToDevice, c->stream));
      }
      c->in_words[m] = c->b_in[m].as<uint64_t>();
      c->in_len32[m] = c->b_in_off[m].as<uint32_t>();
    } else {
      c->in_words[m] = w[m];
      c->in_len32[m] = l[m];
    }
    c->in_stride[m] = st[m];
  }
  if (!paired) {
    c->in_words[1] = nullptr;
    c->in_len32[1] = nullptr;
    c->in_stride[1] = 0;
  }
  int rc = setup_call(c, p, n, paired, max_len, nullptr);
  if (rc) return rc;
  c->cb.fuse_count = paired ? 0u : 1u;  // (mates of varying length: where R1 ends is not fixed by the key)
  c->skip_pack = false;
  return start_call(c);
}

int nimble_call_ex(nimble_ctx *c, const nimble_align_params *p, const uint8_t *r1, const uint64_t *r1_off,
                   const uint8_t *r2, const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                   const nimble_call_extra *ex) {
  DRAIN_STALE_HIP_ERROR();
  if (!ex) return nimble_call(c, p, r1, r1_off, r2, r2_off, n, fixed_len, max_len, mem);
  if (!c || !p) return fail(NIMBLE_E_INVALID, "nimble_call_ex: NULL argument");
  int rc = check_read_args(r1, r1_off, r2, r2_off, n, fixed_len, max_len, mem);
  if (rc) return rc;
  if (!r2 && (ex->qual[1] || ex->skip[1])) return fail(NIMBLE_E_INVALID, "nimble_call_ex: mate extras without mates");
  HIPCHK(hipSetDevice(c->ix->device));
  rc = stage_inputs(c, r1, r1_off, r2, r2_off, n, fixed_len, max_len, mem);
  if (rc) return rc;
  rc = setup_call(c, p, n, r2 != nullptr, max_len, nullptr);
  if (rc) return rc;
  CallBuffers &cb = c->cb;
  const size_t nn = std::max<uint64_t>(n, 1);
  const bool host = mem == NIMBLE_MEM_HOST;
  auto stage = [&](DevBuf &b, const void *src, size_t bytes, const void *&dev) -> int {
    if (!host) {
      dev = src;
      return NIMBLE_OK;
    }
    int r = b.ensure(std::max<size_t>(bytes, 16), &c->bytes);
    if (r) return r;
    if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    dev = b.p;
    return NIMBLE_OK;
  };
  // dedup / count scope per read
  if (ex->segment) {
    uint64_t nseg = ex->n_segments;
    if (nseg == 0) {
      if (!host) return fail(NIMBLE_E_INVALID, "nimble_call_ex: n_segments is required for device-resident segment ids");
      for (uint64_t i = 0; i < n; ++i) nseg = std::max<uint64_t>(nseg, (uint64_t)ex->segment[i] + 1);
    }
    uint32_t b = 1;
    while (b < 32 && (1ULL << b) <= (uint64_t)c->ix->dev.cls_cap) ++b;  // ids + 1 fit in b bits
    if (2 * b >= 64 || nseg >= (1ULL << (64 - 2 * b)))
      return fail(NIMBLE_E_OVERFLOW, "nimble_call_ex: too many segments for one call (split the batch)");
    const void *dev = nullptr;
    rc = stage(c->b_seg, ex->segment, n * 4, dev);
    if (rc) return rc;
    cb.seg = (const uint32_t *)dev;
    cb.cls_bits = b;
  }
  rc = c->b_hist_rep.ensure(c->hist_slots * 4, &c->bytes);
  if (rc) return rc;
  cb.hist_rep = c->b_hist_rep.as<uint32_t>();
  // SKIP_ALIGN dummies
  for (int m = 0; m < (r2 ? 2 : 1); ++m)
    if (ex->skip[m]) {
      const void *dev = nullptr;
      rc = stage(c->b_skip[m], ex->skip[m], n, dev);
      if (rc) return rc;
      cb.skip[m] = (const uint8_t *)dev;
    }
  // quality trim: aligned length per mate from the quality strings (same layout as the bases)
  bool trimmed = false;
  for (int m = 0; m < (r2 ? 2 : 1); ++m)
    if (ex->qual[m]) {
      rc = ensure_trim_tables(c, ex->trim_target_length, ex->trim_strictness);
      if (rc) return rc;
      const uint64_t *off_h = m ? r2_off : r1_off;
      const uint64_t bytes = !host ? 0 : (off_h ? off_h[n] : n * (uint64_t)fixed_len);
      const void *dev = nullptr;
      rc = stage(c->b_qual[m], ex->qual[m], bytes, dev);
      if (rc) return rc;
      rc = c->b_alen[m].ensure(nn * 4, &c->bytes);
      if (rc) return rc;
      launch_maxinfo(c->stream, (const uint8_t *)dev, c->in_off[m], fixed_len, n, c->b_trim_ls.as<int64_t>(),
                     c->b_trim_qp.as<int64_t>(), c->b_alen[m].as<uint32_t>());
      cb.alen[m] = c->b_alen[m].as<uint32_t>();
      trimmed = true;
    }
  // classes are a function of the dedup key only when nothing but the key decides what is aligned
  cb.fuse_count = (!trimmed && !ex->skip[0] && !ex->skip[1] && (!r2 || !r1_off)) ? 1u : 0u;
  c->skip_pack = false;
  return start_call(c);
}

uint32_t nimble_key_words(uint32_t max_len, int paired) { return (max_len * (paired ? 2u : 1u) + 31u) / 32u; }

int nimble_pack(nimble_ctx *c, const nimble_align_params *p, const uint8_t *r1, const uint64_t *r1_off,
                const uint8_t *r2, const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                const nimble_packed *out) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !p || !out) return fail(NIMBLE_E_INVALID, "nimble_pack: NULL argument");
  if (!out->keys || !out->hash || !out->len[0] || !out->pre[0] || (r2 && (!out->len[1] || !out->pre[1])))
    return fail(NIMBLE_E_INVALID, "nimble_pack: output arrays missing");
  int rc = check_read_args(r1, r1_off, r2, r2_off, n, fixed_len, max_len, mem);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->ix->device));
  rc = stage_inputs(c, r1, r1_off, r2, r2_off, n, fixed_len, max_len, mem);
  if (rc) return rc;
  rc = setup_call(c, p, n, r2 != nullptr, max_len, out);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync((uint64_t *)c->b_state.p + 14, 0, 8, c->stream));  // (the latch is this pack's from here on)
  launch_pack(c->stream, c->in_r[0], c->in_off[0], c->in_r[1], c->in_off[1], c->in_fixed_len, c->in_max_len,
              c->prm.min_read_length, c->b_plog.as<double>(), c->plog_max_len, c->cb);
  HIPCHK(hipGetLastError());
  return NIMBLE_OK;
}

// view of caller-owned packed arrays as CallBuffers (only the fields the exchange kernels touch)
static void packed_view(CallBuffers &v, const nimble_packed *pk, uint64_t n) {
  memset(&v, 0, sizeof v);
  v.n = n;
  v.key_stride = n;
  v.key_words = pk->key_words;
  v.paired = pk->paired ? 1 : 0;
  v.keys = pk->keys;
  v.key_hash = pk->hash;
  for (int m = 0; m < 2; ++m) {
    v.len[m] = pk->len[m];
    v.pre[m] = pk->pre[m];
  }
}

int nimble_route_counts(nimble_ctx *c, uint64_t *counts);
int nimble_route_records(nimble_ctx *c, const nimble_packed *in, uint64_t n, uint32_t world, uint64_t *records,
                         uint64_t *counts) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !in || (n && !records)) return fail(NIMBLE_E_INVALID, "nimble_route_records: NULL argument");
  if (world == 0 || world > 256) return fail(NIMBLE_E_INVALID, "nimble_route_records: world must be 1..256");
  if (n && (!in->keys || !in->hash || !in->len[0] || !in->pre[0] || (in->paired && (!in->len[1] || !in->pre[1]))))
    return fail(NIMBLE_E_INVALID, "nimble_route_records: packed arrays missing");
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = NIMBLE_OK;
  {
    const void *before = c->b_state.p;
    rc = c->b_state.ensure(16 * 8, &c->bytes);
    if (rc) return rc;
    if (c->b_state.p != before) HIPCHK(hipMemsetAsync(c->b_state.p, 0, 16 * 8, c->stream));
  }
  const uint64_t cells = (uint64_t)route_grid() * world;
  rc = c->b_route.ensure(cells * 4 + cells * 8 + 256 * 8, &c->bytes);
  if (rc) return rc;
  uint64_t *block_first = c->b_route.as<uint64_t>();
  uint64_t *totals = block_first + cells;
  uint32_t *block_counts = reinterpret_cast<uint32_t *>(totals + 256);
  CallBuffers v{};
  packed_view(v, in, n);
  launch_route(c->stream, v, world, block_counts, block_first, totals, records);
  HIPCHK(hipGetLastError());
  if (!c->defer.ev_route) HIPCHK(hipEventCreateWithFlags(&c->defer.ev_route, hipEventDisableTiming));
  if (!c->defer.p_counts)
    HIPCHK(hipHostMalloc((void **)&c->defer.p_counts, 257 * sizeof(uint64_t), hipHostMallocDefault));
  HIPCHK(hipMemcpyAsync(c->defer.p_counts, totals, world * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->defer.p_counts + 256, (uint64_t *)c->b_state.p + 14, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipEventRecord(c->defer.ev_route, c->stream));
  c->defer.counts_world = world;
  // counts == NULL: asynchronous; nimble_route_counts waits for exactly this routing and hands the counts over
  return counts ? nimble_route_counts(c, counts) : NIMBLE_OK;
}

int nimble_unpack_records(nimble_ctx *c, const uint64_t *records, uint64_t n, const nimble_packed *out) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !out || (n && !records)) return fail(NIMBLE_E_INVALID, "nimble_unpack_records: NULL argument");
  if (n && (!out->keys || !out->hash || !out->len[0] || !out->pre[0] || (out->paired && (!out->len[1] || !out->pre[1]))))
    return fail(NIMBLE_E_INVALID, "nimble_unpack_records: packed arrays missing");
  HIPCHK(hipSetDevice(c->ix->device));
  CallBuffers v{};
  packed_view(v, out, n);
  launch_records_unpack(c->stream, records, v);
  HIPCHK(hipGetLastError());
  return NIMBLE_OK;
}

int nimble_call_records(nimble_ctx *c, const nimble_align_params *p, const uint64_t *records, uint64_t n,
                        uint32_t max_len, int paired) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !p || (n && !records)) return fail(NIMBLE_E_INVALID, "nimble_call_records: NULL argument");
  if (n >= 0xFFFFFFF0ULL) return fail(NIMBLE_E_INVALID, "nimble_call_records: more than 2^32 reads in one call");
  if (max_len == 0 || max_len > 65535) return fail(NIMBLE_E_INVALID, "nimble_call_records: bad max_len");
  HIPCHK(hipSetDevice(c->ix->device));
  static const uint64_t none[1] = {0};
  int rc = setup_call(c, p, n, paired != 0, max_len, nullptr, n ? records : none);
  if (rc) return rc;
  c->cb.fuse_count = paired ? 0u : 1u;
  c->skip_pack = true;
  return start_call(c);
}

// ---- align-where-the-reads-are form of the multi-GPU step -----------------------------------------------------
int nimble_ctx_defer_dedup(nimble_ctx *c, uint32_t world, uint64_t *records, uint32_t *perm) {
  if (!c) return fail(NIMBLE_E_INVALID, "nimble_ctx_defer_dedup: NULL context");
  if (world == 0) {
    c->defer.world = 0;
    return NIMBLE_OK;
  }
  if (world > 256) return fail(NIMBLE_E_INVALID, "nimble_ctx_defer_dedup: world must be 1..256");
  if (!records || !perm) return fail(NIMBLE_E_INVALID, "nimble_ctx_defer_dedup: NULL buffer");
  HIPCHK(hipSetDevice(c->ix->device));
  const uint64_t cells = (uint64_t)route_grid() * world;
  int rc = c->b_route.ensure(cells * 4 + cells * 8 + 256 * 8, &c->bytes);
  if (rc) return rc;
  if (!c->defer.ev_route) HIPCHK(hipEventCreateWithFlags(&c->defer.ev_route, hipEventDisableTiming));
  if (!c->defer.p_counts) HIPCHK(hipHostMalloc((void **)&c->defer.p_counts, 257 * sizeof(uint64_t), hipHostMallocDefault));
  c->defer.world = world;
  c->defer.records = records;
  c->defer.perm = perm;
  return NIMBLE_OK;
}

int nimble_route_counts(nimble_ctx *c, uint64_t *counts) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !counts) return fail(NIMBLE_E_INVALID, "nimble_route_counts: NULL argument");
  if (c->defer.counts_world == 0) return fail(NIMBLE_E_INVALID, "nimble_route_counts: nothing was routed");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipEventSynchronize(c->defer.ev_route));  // the routing only; whatever was enqueued behind it keeps running
  std::copy(c->defer.p_counts, c->defer.p_counts + c->defer.counts_world, counts);
  if (c->defer.p_counts[256] != 0) {  // the pack that produced these keys met offsets it could not trust
    c->defer.p_counts[256] = 0;
    HIPCHK(hipMemsetAsync((uint64_t *)c->b_state.p + 14, 0, 8, c->stream));
    return fail(NIMBLE_E_INVALID, "offsets not monotone or a read longer than max_len (found on the device)");
  }
  return NIMBLE_OK;
}

int nimble_dedup_records(nimble_ctx *c, const uint64_t *records, uint64_t n, uint32_t key_words, uint8_t *verdict) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || (n && (!records || !verdict))) return fail(NIMBLE_E_INVALID, "nimble_dedup_records: NULL argument");
  if (n >= 0xFFFFFFF0ULL) return fail(NIMBLE_E_INVALID, "nimble_dedup_records: more than 2^32 records");
  if (key_words == 0) return fail(NIMBLE_E_INVALID, "nimble_dedup_records: key_words is 0");
  if (c->called && !c->finished)
    return fail(NIMBLE_E_INVALID, "nimble_dedup_records: the context holds a call in flight (use another context)");
  HIPCHK(hipSetDevice(c->ix->device));
  const uint64_t slots = std::max<uint64_t>(n + n / 2, 1024);
  int rc = c->b_dedup.ensure(slots * 8, &c->bytes);
  if (rc) return rc;
  c->dedup_clean_slots = 0;
  HIPCHK(hipMemsetAsync(c->b_dedup.p, 0, slots * 8, c->stream));
  launch_dedup_records(c->stream, records, n, key_words, c->b_dedup.as<uint64_t>(), (uint32_t)slots, verdict);
  HIPCHK(hipGetLastError());
  return NIMBLE_OK;
}

int nimble_count_verdicts(nimble_ctx *c, const uint8_t *verdict) {
  DRAIN_STALE_HIP_ERROR();
  if (!c) return fail(NIMBLE_E_INVALID, "nimble_count_verdicts: NULL context");
  if (!c->defer.active || c->finished)
    return fail(NIMBLE_E_INVALID, "nimble_count_verdicts: no call with deferred dedup in flight");
  if (c->defer.counted) return fail(NIMBLE_E_INVALID, "nimble_count_verdicts: already called for this call");
  if (c->cb.n && !verdict) return fail(NIMBLE_E_INVALID, "nimble_count_verdicts: NULL verdicts");
  HIPCHK(hipSetDevice(c->ix->device));
  c->defer.verdict = verdict;
  c->defer.counted = true;
  return enqueue_count_verdicts(c);
}

int nimble_call_packed(nimble_ctx *c, const nimble_align_params *p, const nimble_packed *in, uint64_t n,
                       uint32_t max_len) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !p || !in) return fail(NIMBLE_E_INVALID, "nimble_call_packed: NULL argument");
  if (n && (!in->keys || !in->hash || !in->len[0] || !in->pre[0]))
    return fail(NIMBLE_E_INVALID, "nimble_call_packed: packed arrays missing");
  if (n >= 0xFFFFFFF0ULL) return fail(NIMBLE_E_INVALID, "nimble_call_packed: more than 2^32 reads in one call");
  if (max_len == 0 || max_len > 65535) return fail(NIMBLE_E_INVALID, "nimble_call_packed: bad max_len");
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = setup_call(c, p, n, in->paired != 0, max_len, in);
  if (rc) return rc;
  c->cb.fuse_count = in->paired ? 0u : 1u;
  c->skip_pack = true;
  return start_call(c);
}

// ---- streamed call: one score::call whose reads arrive in batches ---------------------------------
// Arrays are laid out for `stream_cap` reads; every batch packs and aligns its own slice [stream_n,
// stream_n + m) while the next batch is parsed and copied; dedup and count run once, over everything, at
// nimble_stream_end -- the dedup scope stays the whole call (src/align.rs:496-505).

static int stream_grow(nimble_ctx *c, uint64_t newcap) {
  HIPCHK(hipStreamSynchronize(c->stream));
  const uint64_t oldcap = c->stream_cap, used = c->stream_n;
  const uint32_t kw = c->cb.key_words;
  auto fresh = [&](DevBuf &nb, size_t bytes) -> int {
    int rc = nb.ensure(std::max<size_t>(bytes, 16), &c->bytes);
    if (rc) return rc;
    HIPCHK(hipMemset(nb.p, 0, bytes));
    return NIMBLE_OK;
  };
  auto swap_in = [&](DevBuf &b, DevBuf &nb) {
    if (c->bytes >= b.cap) c->bytes -= b.cap;
    b.release();
    b = nb;
    nb.p = nullptr;
    nb.cap = 0;
  };
  {
    DevBuf nk;
    int rc = fresh(nk, newcap * kw * 8);
    if (rc) return rc;
    if (used)
      HIPCHK(hipMemcpy2D(nk.p, newcap * 8, c->b_keys.p, oldcap * 8, used * 8, kw, hipMemcpyDeviceToDevice));
    swap_in(c->b_keys, nk);
  }
  auto regrow = [&](DevBuf &b, size_t elem) -> int {
    DevBuf nb;
    int rc = fresh(nb, newcap * elem);
    if (rc) return rc;
    if (used) HIPCHK(hipMemcpy(nb.p, b.p, used * elem, hipMemcpyDeviceToDevice));
    swap_in(b, nb);
    return NIMBLE_OK;
  };
  int rc = regrow(c->b_hash, 8);
  for (int m = 0; m < 2 && !rc; ++m) {
    rc = regrow(c->b_len[m], 4);
    if (!rc) rc = regrow(c->b_pre[m], 1);
    if (!rc) rc = regrow(c->b_reason[m], 1);
    if (!rc) rc = regrow(c->b_score[m], 4);
    if (!rc) rc = regrow(c->b_mism[m], 4);
    if (!rc) rc = regrow(c->b_cls[m], 4);
    if (!rc) rc = regrow(c->b_dyn_off[m], 4);
    if (!rc) rc = regrow(c->b_dyn_len[m], 4);
    if (!rc) rc = regrow(c->b_dyn_hash[m], 8);
    if (!rc) rc = regrow(c->b_dyn_pos[m], 4);
  }
  if (rc) return rc;
  {  // the pending-class scratch keeps its content; setup_call would otherwise re-allocate it for the new size
    const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(c->scratch_cap, env_u64("NIMBLE_SCRATCH_PER_READ", 4) * newcap),
                                             0xFFFFFF00ULL);
    if (want > c->scratch_cap) {
      DevBuf nb;
      rc = fresh(nb, want * 4);
      if (rc) return rc;
      HIPCHK(hipMemcpy(nb.p, c->b_scratch.p, c->scratch_cap * 4, hipMemcpyDeviceToDevice));
      swap_in(c->b_scratch, nb);
      c->scratch_cap = want;
    }
  }
  rc = setup_call(c, &c->prm, newcap, c->cb.paired != 0, c->stream_max_len, nullptr);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(c->b_dedup.p, 0, c->dslots * 8, c->stream));
  c->dedup_clean_slots = 0;
  c->stream_cap = newcap;
  return NIMBLE_OK;
}

int nimble_stream_begin(nimble_ctx *c, const nimble_align_params *p, int paired, uint32_t max_len,
                        uint64_t capacity_hint) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !p) return fail(NIMBLE_E_INVALID, "nimble_stream_begin: NULL argument");
  if (c->streaming) return fail(NIMBLE_E_INVALID, "nimble_stream_begin: a streamed call is already open");
  if (max_len == 0 || max_len > 65535) return fail(NIMBLE_E_INVALID, "nimble_stream_begin: bad max_len");
  HIPCHK(hipSetDevice(c->ix->device));
  if (!c->h2d_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
      HIPCHK(hipEventCreateWithFlags(&c->ev_h2d[k], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->ev_used[k], hipEventDisableTiming));
    }
  }
  const uint64_t cap = std::min<uint64_t>(std::max<uint64_t>(capacity_hint, 1ULL << 16), 0xFFFFFFEFULL);
  int rc = setup_call(c, p, cap, paired != 0, max_len, nullptr);
  if (rc) return rc;
  c->prm = *p;
  c->stream_cap = cap;
  c->stream_n = 0;
  c->stream_max_len = max_len;
  c->stage_busy[0] = c->stage_busy[1] = false;
  c->stage_k = 0;
  c->h2d_pending[0] = c->h2d_pending[1] = false;
  c->align_from = 0;
  HIPCHK(hipMemcpyAsync(c->p_dyn, c->ix->b_dyn_state.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  c->defer.active = false;
  c->defer.world = 0;
  c->skip_pack = false;  // (the appends pack: the head clears the input-error latch)
  rc = enqueue_head(c);
  if (rc) return rc;
  c->streaming = true;
  c->finished = true;  // nothing to fetch until nimble_stream_end
  c->called = false;
  return NIMBLE_OK;
}

// the call arrays as one batch of a streamed call sees them: reads [from, from + count)
static CallBuffers stream_view(const nimble_ctx *c, uint64_t from, uint64_t count) {
  CallBuffers v = c->cb;
  v.n = count;
  v.key_stride = c->stream_cap;
  v.keys += from;
  v.key_hash += from;
  v.slot += from;
  v.counted += from;
  for (int mt = 0; mt < 2; ++mt) {
    v.len[mt] += from;
    v.alen[mt] += from;  // aliases len (a streamed call does not trim)
    v.pre[mt] += from;
    v.reason[mt] += from;
    v.score[mt] += from;
    v.mism[mt] += from;
    v.cls[mt] += from;
    v.dyn_off[mt] += from;
    v.dyn_len[mt] += from;
    v.dyn_hash[mt] += from;
    v.dyn_pos[mt] += from;
  }
  return v;
}

// The align kernel of a streamed call runs over everything packed since its last launch, once that is worth a launch
// (a persistent grid over a 25 k-read batch is mostly launch and drain: 40 us for 10 us of work, and with the batches
// arriving packed that was what the stream waited for) or when the stream ends.
constexpr uint64_t STREAM_ALIGN_READS = 1u << 18;
static int stream_flush_align(nimble_ctx *c, bool force) {
  const uint64_t pending = c->stream_n - c->align_from;
  if (pending == 0 || (!force && pending < STREAM_ALIGN_READS)) return NIMBLE_OK;
  launch_align(c->stream, c->ix->dev, c->prm, stream_view(c, c->align_from, pending), c->want_counters, c->align_grid_pct, c->stream_cus);
  HIPCHK(hipGetLastError());
  c->align_from = c->stream_n;
  return NIMBLE_OK;
}

int nimble_stream_append(nimble_ctx *c, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                         const uint64_t *r2_off, uint64_t m, uint32_t fixed_len, int mem) {
  DRAIN_STALE_HIP_ERROR();
  if (!c) return fail(NIMBLE_E_INVALID, "nimble_stream_append: NULL context");
  if (!c->streaming) return fail(NIMBLE_E_INVALID, "nimble_stream_append: no streamed call is open");
  if ((r2 != nullptr) != (c->cb.paired != 0))
    return fail(NIMBLE_E_INVALID, "nimble_stream_append: mates given for a single-end stream or missing for a paired one");
  uint32_t max_len = c->stream_max_len;
  if (!r1_off && fixed_len > max_len)
    return fail(NIMBLE_E_INVALID, "nimble_stream_append: a read longer than the stream's max_len");
  int rc = check_read_args(r1, r1_off, r2, r2_off, m, fixed_len, max_len, mem);
  if (rc) return rc;
  if (m == 0) return NIMBLE_OK;
  if (c->stream_n + m >= 0xFFFFFFF0ULL) return fail(NIMBLE_E_INVALID, "nimble_stream_append: more than 2^32 reads");
  HIPCHK(hipSetDevice(c->ix->device));
  if (c->stream_n + m > c->stream_cap) {
    rc = stream_grow(c, std::min<uint64_t>(std::max<uint64_t>(2 * c->stream_cap, c->stream_n + m), 0xFFFFFFEFULL));
    if (rc) return rc;
  }
  const uint8_t *in_r[2] = {r1, r2};
  const uint64_t *in_off[2] = {r1_off, r2_off};
  const int k = c->stage_k;
  const bool lazy = mem == NIMBLE_MEM_HOST_PINNED;
  // Page-locked batches: the copy of the batch before this one may still run -- the link never idles while the host
  // queues the next batch's work -- and the one before that (the last user of staging slot k and of its event) is waited
  // for here.  A batch in any other memory ends every copy still pending first.
  for (int q = 0; q < 2; ++q)
    if (c->h2d_pending[q] && (q == k || !lazy)) {
      HIPCHK(hipEventSynchronize(c->ev_h2d[q]));
      c->h2d_pending[q] = false;
    }
  if (lazy) mem = NIMBLE_MEM_HOST;
  if (mem == NIMBLE_MEM_HOST) {
    // device staging slot k: wait until the pack kernel that read it last has run, copy on the side stream
    // (overlaps the kernels of the previous batch), let the launch stream wait for the copy
    if (c->stage_busy[k]) HIPCHK(hipEventSynchronize(c->ev_used[k]));
    for (int mt = 0; mt < (r2 ? 2 : 1); ++mt) {
      const uint64_t *off = in_off[mt];
      const uint64_t bytes = off ? off[m] - off[0] : m * (uint64_t)fixed_len;
      if (off)
        for (uint64_t i = 0; i < m; ++i)
          if (off[i + 1] < off[i] || off[i + 1] - off[i] > max_len)
            return fail(NIMBLE_E_INVALID, "nimble_stream_append: offsets not monotone or a read longer than max_len");
      rc = c->b_stage[k][mt].ensure(std::max<uint64_t>((off ? off[m] : bytes), 16), &c->bytes);
      if (rc) return rc;
      const uint64_t lo = off ? off[0] : 0;
      if (bytes)
        HIPCHK(hipMemcpyAsync(c->b_stage[k][mt].as<uint8_t>() + lo, in_r[mt] + lo, bytes, hipMemcpyHostToDevice,
                              c->h2d_stream));
      in_r[mt] = c->b_stage[k][mt].as<uint8_t>();
      if (off) {
        rc = c->b_stage_off[k][mt].ensure((m + 1) * 8, &c->bytes);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->b_stage_off[k][mt].p, off, (m + 1) * 8, hipMemcpyHostToDevice, c->h2d_stream));
        in_off[mt] = c->b_stage_off[k][mt].as<uint64_t>();
      }
    }
    HIPCHK(hipEventRecord(c->ev_h2d[k], c->h2d_stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_h2d[k], 0));
  }
  const CallBuffers v = stream_view(c, c->stream_n, m);
  launch_pack(c->stream, in_r[0], in_off[0], in_r[1], in_off[1], fixed_len, max_len, c->prm.min_read_length,
              c->b_plog.as<double>(), c->plog_max_len, v);
  HIPCHK(hipGetLastError());
  if (mem == NIMBLE_MEM_HOST) {
    HIPCHK(hipEventRecord(c->ev_used[k], c->stream));
    c->stage_busy[k] = true;
    c->stage_k ^= 1;
    if (lazy) c->h2d_pending[k] = true;                // waited for by the next append but one / the end of the stream
    else HIPCHK(hipEventSynchronize(c->ev_h2d[k]));  // the host buffers are free again
  }
  c->stream_n += m;
  return stream_flush_align(c, false);
}

int nimble_stream_append_packed(nimble_ctx *c, const uint64_t *r1_words, const uint32_t *r1_len, uint32_t r1_stride,
                                const uint64_t *r2_words, const uint32_t *r2_len, uint32_t r2_stride, uint64_t m) {
  DRAIN_STALE_HIP_ERROR();
  if (!c) return fail(NIMBLE_E_INVALID, "nimble_stream_append_packed: NULL context");
  if (!c->streaming) return fail(NIMBLE_E_INVALID, "nimble_stream_append_packed: no streamed call is open");
  const bool paired = c->cb.paired != 0;
  if ((r2_words != nullptr) != paired || (r2_len != nullptr) != paired)
    return fail(NIMBLE_E_INVALID, "nimble_stream_append_packed: mates given for a single-end stream or missing for a paired one");
  if (m == 0) return NIMBLE_OK;
  if (!r1_words || !r1_len || r1_stride == 0 || (paired && r2_stride == 0))
    return fail(NIMBLE_E_INVALID, "nimble_stream_append_packed: NULL buffer or zero stride");
  const uint32_t max_len = c->stream_max_len;
  for (int mt = 0; mt < (paired ? 2 : 1); ++mt) {
    const uint32_t *len = mt ? r2_len : r1_len;
    const uint64_t cap = std::min<uint64_t>(max_len, 32ULL * (mt ? r2_stride : r1_stride));
    for (uint64_t i = 0; i < m; ++i)
      if (len[i] > cap)
        return fail(NIMBLE_E_INVALID, "nimble_stream_append_packed: a read longer than the stream's max_len or than its words");
  }
  if (c->stream_n + m >= 0xFFFFFFF0ULL) return fail(NIMBLE_E_INVALID, "nimble_stream_append_packed: more than 2^32 reads");
  HIPCHK(hipSetDevice(c->ix->device));
  int rc;
  if (c->stream_n + m > c->stream_cap) {
    rc = stream_grow(c, std::min<uint64_t>(std::max<uint64_t>(2 * c->stream_cap, c->stream_n + m), 0xFFFFFFEFULL));
    if (rc) return rc;
  }
  const int k = c->stage_k;
  // as NIMBLE_MEM_HOST_PINNED: the copy of the batch before this one may still run; the one before that is waited for
  if (c->h2d_pending[k]) {
    HIPCHK(hipEventSynchronize(c->ev_h2d[k]));
    c->h2d_pending[k] = false;
  }
  if (c->stage_busy[k]) HIPCHK(hipEventSynchronize(c->ev_used[k]));
  const uint64_t *d_words[2] = {nullptr, nullptr};
  const uint32_t *d_len[2] = {nullptr, nullptr};
  for (int mt = 0; mt < (paired ? 2 : 1); ++mt) {
    const uint64_t wbytes = m * (uint64_t)(mt ? r2_stride : r1_stride) * 8;
    rc = c->b_stage[k][mt].ensure(std::max<uint64_t>(wbytes, 16), &c->bytes);
    if (rc) return rc;
    rc = c->b_stage_off[k][mt].ensure(std::max<uint64_t>(m * 4, 16), &c->bytes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->b_stage[k][mt].p, mt ? r2_words : r1_words, wbytes, hipMemcpyHostToDevice, c->h2d_stream));
    HIPCHK(hipMemcpyAsync(c->b_stage_off[k][mt].p, mt ? r2_len : r1_len, m * 4, hipMemcpyHostToDevice, c->h2d_stream));
    d_words[mt] = c->b_stage[k][mt].as<uint64_t>();
    d_len[mt] = c->b_stage_off[k][mt].as<uint32_t>();
  }
  HIPCHK(hipEventRecord(c->ev_h2d[k], c->h2d_stream));
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_h2d[k], 0));
  launch_pack_words(c->stream, d_words[0], d_len[0], r1_stride, d_words[1], d_len[1], r2_stride, max_len,
                    c->prm.min_read_length, c->b_plog.as<double>(), stream_view(c, c->stream_n, m));
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->ev_used[k], c->stream));
  c->stage_busy[k] = true;
  c->stage_k ^= 1;
  c->h2d_pending[k] = true;
  c->stream_n += m;
  return stream_flush_align(c, false);
}

int nimble_stream_end(nimble_ctx *c) {
  DRAIN_STALE_HIP_ERROR();
  if (!c) return fail(NIMBLE_E_INVALID, "nimble_stream_end: NULL context");
  if (!c->streaming) return fail(NIMBLE_E_INVALID, "nimble_stream_end: no streamed call is open");
  HIPCHK(hipSetDevice(c->ix->device));
  for (int q = 0; q < 2; ++q)
    if (c->h2d_pending[q]) {
      HIPCHK(hipEventSynchronize(c->ev_h2d[q]));
      c->h2d_pending[q] = false;
    }
  {
    const int frc = stream_flush_align(c, true);
    if (frc) return frc;
  }
  c->streaming = false;
  c->cb.n = c->stream_n;
  c->cb.key_stride = c->stream_cap;
  c->skip_pack = true;  // a pool regrow in finish_call re-aligns everything from the packed keys
  c->attempt = 0;
  HIPCHK(hipEventRecord(c->ev[1], c->stream));
  int rc = enqueue_tail(c);
  if (rc) return rc;
  c->finished = false;
  c->called = true;
  return NIMBLE_OK;
}

int nimble_pinned_alloc(uint64_t bytes, void **out) {
  if (!out) return fail(NIMBLE_E_INVALID, "nimble_pinned_alloc: NULL argument");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, std::max<uint64_t>(bytes, 16), hipHostMallocDefault);
  if (e != hipSuccess) return fail(NIMBLE_E_NOMEM, std::string("hipHostMalloc failed: ") + hipGetErrorString(e));
  return NIMBLE_OK;
}

void nimble_pinned_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int nimble_pinned_register(void *p, uint64_t bytes) {
  if (!p || !bytes) return fail(NIMBLE_E_INVALID, "nimble_pinned_register: NULL argument");
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(NIMBLE_E_HIP, std::string("hipHostRegister failed: ") + hipGetErrorString(e));
  }
  return NIMBLE_OK;
}

void nimble_pinned_unregister(void *p) {
  if (p && hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
}

static int finish_count_stage(nimble_ctx *c) { return finish_call(c); }

int nimble_histogram(nimble_ctx *c, uint32_t *class_r1, uint32_t *class_r2, uint64_t *count, uint64_t cap,
                     uint64_t *n_entries) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !n_entries) return fail(NIMBLE_E_INVALID, "nimble_histogram: NULL argument");
  if (!c->called) return fail(NIMBLE_E_INVALID, "nimble_histogram: no call has been made on this context");
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = finish_count_stage(c);
  if (rc) return rc;
  const uint64_t ne = c->h_cnt.size();
  *n_entries = ne;
  if (cap == 0 || ne == 0) return NIMBLE_OK;
  if (!class_r1 || !class_r2 || !count) return fail(NIMBLE_E_INVALID, "nimble_histogram: NULL output");
  const uint64_t m = std::min(ne, cap);
  memcpy(class_r1, c->h_c1.data(), m * 4);
  memcpy(class_r2, c->h_c2.data(), m * 4);
  memcpy(count, c->h_cnt.data(), m * 8);
  return NIMBLE_OK;
}

int nimble_histogram_seg(nimble_ctx *c, uint32_t *segment, uint32_t *class_r1, uint32_t *class_r2, uint64_t *count,
                         uint32_t *representative, uint64_t cap, uint64_t *n_entries) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !n_entries) return fail(NIMBLE_E_INVALID, "nimble_histogram_seg: NULL argument");
  if (!c->called) return fail(NIMBLE_E_INVALID, "nimble_histogram_seg: no call has been made on this context");
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = finish_count_stage(c);
  if (rc) return rc;
  const uint64_t ne = c->h_cnt.size();
  *n_entries = ne;
  if (cap == 0 || ne == 0) return NIMBLE_OK;
  const uint64_t m = std::min(ne, cap);
  if (segment) memcpy(segment, c->h_seg.data(), m * 4);
  if (class_r1) memcpy(class_r1, c->h_c1.data(), m * 4);
  if (class_r2) memcpy(class_r2, c->h_c2.data(), m * 4);
  if (count) memcpy(count, c->h_cnt.data(), m * 8);
  if (representative) memcpy(representative, c->h_rep.data(), m * 4);
  return NIMBLE_OK;
}

int nimble_read_align_len(nimble_ctx *c, int mate, uint32_t *align_len, uint64_t n) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !align_len) return fail(NIMBLE_E_INVALID, "nimble_read_align_len: NULL argument");
  if (!c->called || n != c->cb.n) return fail(NIMBLE_E_INVALID, "nimble_read_align_len: n does not match the last call");
  if (mate < 0 || mate > 1) return fail(NIMBLE_E_INVALID, "nimble_read_align_len: mate must be 0 or 1");
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = finish_count_stage(c);
  if (rc) return rc;
  if (n == 0) return NIMBLE_OK;
  if (mate == 1 && !c->cb.paired) {
    memset(align_len, 0, n * 4);
    return NIMBLE_OK;
  }
  if (c->cb.rec) return fail(NIMBLE_E_INVALID, "nimble_read_align_len: the call read exchange records (lengths are in them)");
  HIPCHK(hipMemcpy(align_len, c->cb.alen[mate], n * 4, hipMemcpyDeviceToHost));
  return NIMBLE_OK;
}

int nimble_histogram_dense_se(nimble_ctx *c, int64_t *counts_dev, uint32_t n_classes) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !counts_dev) return fail(NIMBLE_E_INVALID, "nimble_histogram_dense_se: NULL argument");
  if (!c->called) return fail(NIMBLE_E_INVALID, "nimble_histogram_dense_se: no call has been made");
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = finish_count_stage(c);
  if (rc) return rc;
  if (c->cb.cls_bits) return fail(NIMBLE_E_INVALID, "nimble_histogram_dense_se: the last call was segmented");
  HIPCHK(hipMemsetAsync(counts_dev, 0, (size_t)n_classes * 8, c->stream));
  launch_hist_dense_se(c->stream, c->cb, counts_dev, n_classes);
  HIPCHK(hipGetLastError());
  return NIMBLE_OK;
}

int nimble_read_records(nimble_ctx *c, int mate, int32_t *reason, int32_t *score, int32_t *mism, uint32_t *cls,
                        uint8_t *counted, uint64_t n) {
  DRAIN_STALE_HIP_ERROR();
  if (!c) return fail(NIMBLE_E_INVALID, "NULL context");
  if (!c->called || n != c->cb.n) return fail(NIMBLE_E_INVALID, "nimble_read_records: n does not match the last call");
  if (mate < 0 || mate > 1) return fail(NIMBLE_E_INVALID, "nimble_read_records: mate must be 0 or 1");
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = finish_count_stage(c);
  if (rc) return rc;
  if (n == 0) return NIMBLE_OK;
  if (counted && !c->counted_marked) {
    // the call counted inside k_dedup; the flags of the representatives (last copy of each key) are made now
    launch_count(c->stream, c->cb);
    HIPCHK(hipStreamSynchronize(c->stream));
    c->counted_marked = true;
  }
  if (mate == 1 && !c->cb.paired) {
    for (uint64_t i = 0; i < n; ++i) {
      if (reason) reason[i] = NIMBLE_R_SUCCESSFUL_MATCH;  // src/align.rs:596-599: no mate filter -> SuccessfulMatch
      if (score) score[i] = 0;
      if (mism) mism[i] = 0;
      if (cls) cls[i] = NIMBLE_CLASS_NONE;
    }
    if (counted) HIPCHK(hipMemcpy(counted, c->cb.counted, n, hipMemcpyDeviceToHost));
    return NIMBLE_OK;
  }
  if (reason) {
    std::vector<uint8_t> t(n);
    HIPCHK(hipMemcpy(t.data(), c->cb.reason[mate], n, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) reason[i] = t[i];
  }
  if (score) HIPCHK(hipMemcpy(score, c->cb.score[mate], n * 4, hipMemcpyDeviceToHost));
  if (mism) HIPCHK(hipMemcpy(mism, c->cb.mism[mate], n * 4, hipMemcpyDeviceToHost));
  if (cls) HIPCHK(hipMemcpy(cls, c->cb.cls[mate], n * 4, hipMemcpyDeviceToHost));
  if (counted) HIPCHK(hipMemcpy(counted, c->cb.counted, n, hipMemcpyDeviceToHost));
  return NIMBLE_OK;
}

// development aid, not part of the ABI in include/nimble_hip.h: section clocks of k_align in a profiling build
extern "C" int nimble_debug_sections(uint64_t out[16], int reset) { return debug_sections(out, reset); }
// the 16 state words of the context's last finished call (kernels.h CallBuffers::state)
// (test hook, not part of the interface) leave a stale error in THIS process's HIP runtime -- the one this library is linked
// against -- the way a foreign library's failed call would: returns the hipError_t of hipSetDevice(999).  (A test that
// loads "libamdhip64.so" by name may get a second copy of the runtime beside torch's: two runtimes in one process.)
extern "C" int nimble_debug_stale_error(void) { return (int)hipSetDevice(999); }

extern "C" int nimble_debug_state(nimble_ctx *c, uint64_t out[16]) {
  if (!c || !out || !c->called) return NIMBLE_E_INVALID;
  int rc = finish_count_stage(c);
  if (rc) return rc;
  for (int i = 0; i < 16; ++i) out[i] = c->h_state[i];
  return NIMBLE_OK;
}

int nimble_call_counters(nimble_ctx *c, uint64_t out[8]) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || !out) return fail(NIMBLE_E_INVALID, "NULL argument");
  if (!c->called) return fail(NIMBLE_E_INVALID, "nimble_call_counters: no call has been made");
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = finish_count_stage(c);
  if (rc) return rc;
  uint64_t ne = 0;
  rc = nimble_histogram(c, nullptr, nullptr, nullptr, 0, &ne);
  if (rc) return rc;
  uint64_t uniq = 0;
  for (uint64_t v : c->h_cnt) uniq += v;
  uint32_t dst[4];
  HIPCHK(hipMemcpy(dst, c->ix->b_dyn_state.p, sizeof(dst), hipMemcpyDeviceToHost));
  out[0] = c->cb.n;
  out[1] = uniq;
  for (int i = 2; i <= 6; ++i) out[i] = c->h_state[i];
  out[7] = dst[0] - c->dyn_before;
  return NIMBLE_OK;
}

int nimble_call_timing(nimble_ctx *c, float ms[6]) {
  if (!c || !ms) return fail(NIMBLE_E_INVALID, "NULL argument");
  if (!c->called) return fail(NIMBLE_E_INVALID, "nimble_call_timing: no call has been made");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipEventSynchronize(c->ev[5]));
  for (int i = 0; i < 5; ++i) HIPCHK(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
  HIPCHK(hipEventElapsedTime(&ms[5], c->ev[0], c->ev[5]));
  return NIMBLE_OK;
}

}  // extern "C"

// =================================================================================================================
// Single-node multi-GPU: one process, one rank (host thread) per device.  The ranks of a nimble_comm exchange routed
// records all-to-all and sum count vectors with RCCL over xGMI; no torch, no MPI.  (SURVEY 8(b) `counts_allreduce`,
// 8(e): reads shard by record, equal keys must meet on one rank because the dedup scope is the whole call.)
// =================================================================================================================
struct nimble_comm {
  int n = 0;
  std::vector<int> devices;
  bool rccl = false;                 // distinct devices: RCCL communicators; repeated devices: device copies (one GPU)
  std::vector<ncclComm_t> comms;
  // what the ranks tell each other between the barriers of a collective call (all in this process's memory)
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  std::vector<uint64_t> counts;            // [src][dst] records of the running all-to-all
  std::vector<const uint64_t *> send_ptr;  // [src]
  std::vector<void *> vec_ptr;             // [rank] host vectors of the running all-reduce
  std::vector<int> status;                 // [rank] a rank's failure is everybody's
  // per rank: the sharded call (nimble_sharded_*)
  struct Shard {
    nimble_ctx *ctx = nullptr;
    nimble_align_params prm{};
    int paired = 0;
    uint32_t max_len = 0, rec_words = 0;
    DevBuf send, acc, vec;
    uint64_t n_acc = 0, acc_cap = 0;
    bool open = false;
    // successive calls, software-pipelined (nimble_steps_*)
    struct Steps {
      bool open = false;
      nimble_ctx *call[2] = {nullptr, nullptr}, *util = nullptr;
      hipStream_t xstream = nullptr;           // the exchange runs here, beside the call of the batch before
      hipEvent_t ev_routed = nullptr, ev_x[2] = {nullptr, nullptr};
      DevBuf send[2], recv[2];
      uint64_t recv_cap[2] = {0, 0}, n_recv[2] = {0, 0};
      uint64_t *p_counts = nullptr;            // page-locked: per-destination counts [0, W) and the input-error latch [W]
      uint64_t b = 0;                          // batches submitted
      bool pending = false;                    // batch b - 1 is exchanged, its call not yet launched
    } st;
  };
  std::vector<Shard> shard;
  // nimble_comm_abort: a rank that failed OUTSIDE the collectives (a host-side fault of its driver thread) will never
  // arrive; the barriers stop holding anybody and every agreement ends in an error, so the other ranks come home
  bool aborted = false;
  void barrier() {
    std::unique_lock<std::mutex> lock(mu);
    if (aborted) return;
    const uint64_t gen = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return generation != gen || aborted; });
    }
  }
  // every rank reports its status; all of them leave with the worst one
  int agree(int rank, int rc) {
    status[rank] = rc;
    barrier();
    int worst = NIMBLE_OK;
    for (int r = 0; r < n; ++r)
      if (status[r] != NIMBLE_OK) worst = status[r];
    barrier();
    {
      std::lock_guard<std::mutex> lock(mu);
      if (aborted && worst == NIMBLE_OK) worst = NIMBLE_E_INTERNAL;
    }
    return worst;
  }
};

#define NCCLCHK(expr)                                                                                  \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess) return fail(NIMBLE_E_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
  } while (0)

extern "C" {

int nimble_comm_create(const int *devices, int n, nimble_comm **out) {
  if (!devices || !out || n < 1 || n > 256) return fail(NIMBLE_E_INVALID, "nimble_comm_create: bad argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NIMBLE_E_NO_DEVICE, "nimble_comm_create: no HIP device (this library has no CPU path)");
  std::vector<int> sorted(devices, devices + n);
  for (int d : sorted)
    if (d < 0 || d >= ndev) return fail(NIMBLE_E_INVALID, "nimble_comm_create: bad device ordinal");
  std::sort(sorted.begin(), sorted.end());
  const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
  if (!distinct && sorted.front() != sorted.back())
    return fail(NIMBLE_E_INVALID, "nimble_comm_create: ranks must be on distinct devices, or all on one");
  nimble_comm *c = new (std::nothrow) nimble_comm();
  if (!c) return fail(NIMBLE_E_NOMEM, "out of memory");
  c->n = n;
  c->devices.assign(devices, devices + n);
  c->rccl = distinct;
  c->counts.assign((size_t)n * n, 0);
  c->send_ptr.assign(n, nullptr);
  c->vec_ptr.assign(n, nullptr);
  c->status.assign(n, NIMBLE_OK);
  c->shard.resize(n);
  if (c->rccl) {
    c->comms.resize(n);
    ncclResult_t r = ncclCommInitAll(c->comms.data(), n, c->devices.data());
    if (r != ncclSuccess) {
      delete c;
      return fail(NIMBLE_E_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
    }
  }
  *out = c;
  return NIMBLE_OK;
}

void nimble_comm_free(nimble_comm *c) {
  if (!c) return;
  for (int r = 0; r < c->n; ++r) {
    (void)hipSetDevice(c->devices[r]);
    c->shard[r].send.release();
    c->shard[r].acc.release();
    c->shard[r].vec.release();
    nimble_comm::Shard::Steps &st = c->shard[r].st;
    for (int k = 0; k < 2; ++k) {
      st.send[k].release();
      st.recv[k].release();
      if (st.ev_x[k]) (void)hipEventDestroy(st.ev_x[k]);
    }
    if (st.ev_routed) (void)hipEventDestroy(st.ev_routed);
    if (st.xstream) (void)hipStreamDestroy(st.xstream);
    if (st.p_counts) (void)hipHostFree(st.p_counts);
  }
  for (ncclComm_t m : c->comms) (void)ncclCommDestroy(m);
  delete c;
}

int nimble_comm_size(const nimble_comm *c) { return c ? c->n : 0; }
void nimble_comm_abort(nimble_comm *c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lock(c->mu);
    c->aborted = true;
  }
  c->cv.notify_all();
}
int nimble_comm_uses_rccl(const nimble_comm *c) { return c && c->rccl ? 1 : 0; }

int nimble_counts_allreduce(nimble_comm *c, int rank, int64_t *counts_dev, uint64_t len, void *stream) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || rank < 0 || rank >= c->n || (len && !counts_dev)) return fail(NIMBLE_E_INVALID, "nimble_counts_allreduce: bad argument");
  HIPCHK(hipSetDevice(c->devices[rank]));
  hipStream_t s = (hipStream_t)stream;
  if (c->rccl) {
    if (len) NCCLCHK(ncclAllReduce(counts_dev, counts_dev, len, ncclInt64, ncclSum, c->comms[rank], s));
    return NIMBLE_OK;
  }
  // ranks that share one device: every rank's vector through the host, summed by each rank for itself
  std::vector<int64_t> mine(len), sum(len, 0);
  int rc = NIMBLE_OK;
  if (len && hipMemcpyAsync(mine.data(), counts_dev, len * 8, hipMemcpyDeviceToHost, s) != hipSuccess) rc = NIMBLE_E_HIP;
  if (rc == NIMBLE_OK && hipStreamSynchronize(s) != hipSuccess) rc = NIMBLE_E_HIP;
  c->vec_ptr[rank] = mine.data();
  rc = c->agree(rank, rc);
  if (rc == NIMBLE_OK)
    for (int r = 0; r < c->n; ++r) {
      const int64_t *v = static_cast<const int64_t *>(c->vec_ptr[r]);
      for (uint64_t i = 0; i < len; ++i) sum[i] += v[i];
    }
  c->barrier();  // nobody's vector goes away before everybody has read it
  if (rc != NIMBLE_OK) return fail(rc, "nimble_counts_allreduce: a rank failed");
  if (len) HIPCHK(hipMemcpyAsync(counts_dev, sum.data(), len * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return NIMBLE_OK;
}

int nimble_counts_allreduce_host(nimble_comm *c, int rank, int64_t *counts, uint64_t len) {
  if (!c || rank < 0 || rank >= c->n || (len && !counts)) return fail(NIMBLE_E_INVALID, "nimble_counts_allreduce_host: bad argument");
  HIPCHK(hipSetDevice(c->devices[rank]));
  nimble_comm::Shard &sh = c->shard[rank];
  int rc = sh.vec.ensure(std::max<uint64_t>(len * 8, 16), nullptr);
  if (rc) return rc;
  hipStream_t s = sh.ctx ? sh.ctx->stream : nullptr;
  if (len) HIPCHK(hipMemcpyAsync(sh.vec.p, counts, len * 8, hipMemcpyHostToDevice, s));
  rc = nimble_counts_allreduce(c, rank, sh.vec.as<int64_t>(), len, s);
  if (rc) return rc;
  if (len) HIPCHK(hipMemcpyAsync(counts, sh.vec.p, len * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return NIMBLE_OK;
}

int nimble_records_alltoall(nimble_comm *c, int rank, const uint64_t *send, const uint64_t *send_counts,
                            uint32_t rec_words, uint64_t *recv, uint64_t recv_cap, uint64_t *n_recv, void *stream) {
  DRAIN_STALE_HIP_ERROR();
  // Whatever goes wrong on one rank -- a bad argument included -- is carried into agree(), never returned past it: the
  // other rank threads stand in that barrier, and a rank that left early would leave them there for good.  (Only a call
  // that cannot name its communicator or rank has nobody to tell.)
  if (!c || rank < 0 || rank >= c->n) return fail(NIMBLE_E_INVALID, "nimble_records_alltoall: bad communicator or rank");
  hipStream_t s = (hipStream_t)stream;
  const int W = c->n;
  int rc = NIMBLE_OK;
  std::string why;
  if (!send_counts || !n_recv || rec_words == 0) {
    rc = NIMBLE_E_INVALID;
    why = "nimble_records_alltoall: bad argument";
  } else if (hipSetDevice(c->devices[rank]) != hipSuccess) {
    rc = NIMBLE_E_HIP;
    why = "nimble_records_alltoall: hipSetDevice failed";
  } else if (!c->rccl && hipStreamSynchronize(s) != hipSuccess) {  // peers copy straight out of `send`
    rc = NIMBLE_E_HIP;
    why = "nimble_records_alltoall: the send stream failed";
  }
  for (int d = 0; d < W; ++d) c->counts[(size_t)rank * W + d] = rc == NIMBLE_OK ? send_counts[d] : 0;
  c->send_ptr[rank] = send;
  {
    const int mine = rc;
    rc = c->agree(rank, rc);
    if (rc != NIMBLE_OK) return fail(rc, mine != NIMBLE_OK ? why : "nimble_records_alltoall: a rank failed");
  }
  uint64_t total = 0;
  for (int src = 0; src < W; ++src) total += c->counts[(size_t)src * W + rank];
  *n_recv = total;
  rc = total > recv_cap ? NIMBLE_E_OVERFLOW : NIMBLE_OK;
  rc = c->agree(rank, rc);  // nobody starts moving records unless everybody has room
  if (rc != NIMBLE_OK) return fail(rc, "nimble_records_alltoall: a receive buffer is too small");
  if (c->rccl) {
    uint64_t so = 0, ro = 0;
    ncclResult_t nr_ = ncclGroupStart();
    for (int peer = 0; peer < W && nr_ == ncclSuccess; ++peer) {
      const uint64_t ns = send_counts[peer], nr = c->counts[(size_t)peer * W + rank];
      if (ns) nr_ = ncclSend(send + so * rec_words, ns * rec_words, ncclUint64, peer, c->comms[rank], s);
      if (nr && nr_ == ncclSuccess) nr_ = ncclRecv(recv + ro * rec_words, nr * rec_words, ncclUint64, peer, c->comms[rank], s);
      so += ns;
      ro += nr;
    }
    {
      const ncclResult_t end_ = ncclGroupEnd();  // (always closed: an open group would swallow the next collective)
      if (nr_ == ncclSuccess) nr_ = end_;
    }
    // the count table is free for the next call -- and every rank learns whether the exchange was issued everywhere
    rc = c->agree(rank, nr_ == ncclSuccess ? NIMBLE_OK : NIMBLE_E_HIP);
    if (rc != NIMBLE_OK)
      return fail(rc, nr_ != ncclSuccess ? std::string("nimble_records_alltoall: ") + ncclGetErrorString(nr_)
                                          : std::string("nimble_records_alltoall: the exchange failed on another rank"));
    return NIMBLE_OK;
  }
  uint64_t ro = 0;
  hipError_t e = hipSuccess;
  for (int src = 0; src < W && e == hipSuccess; ++src) {
    uint64_t so = 0;
    for (int d = 0; d < rank; ++d) so += c->counts[(size_t)src * W + d];
    const uint64_t nr = c->counts[(size_t)src * W + rank];
    if (nr) e = hipMemcpyAsync(recv + ro * rec_words, c->send_ptr[src] + so * rec_words, nr * rec_words * 8,
                               hipMemcpyDeviceToDevice, s);
    ro += nr;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  rc = c->agree(rank, e == hipSuccess ? NIMBLE_OK : NIMBLE_E_HIP);  // send buffers may be refilled after this
  if (rc != NIMBLE_OK) return fail(rc, "nimble_records_alltoall: device copy failed");
  return NIMBLE_OK;
}

// ---- one score::call whose reads are spread over the ranks: every rank appends its share batch by batch (pack where the
//      reads are, route by key hash, all-to-all, keep what it owns) and finally runs the call over the records it owns.
int nimble_sharded_begin(nimble_comm *c, int rank, nimble_ctx *ctx, const nimble_align_params *p, int paired,
                         uint32_t max_len) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || rank < 0 || rank >= c->n || !ctx || !p) return fail(NIMBLE_E_INVALID, "nimble_sharded_begin: bad argument");
  if (ctx->ix->device != c->devices[rank]) return fail(NIMBLE_E_INVALID, "nimble_sharded_begin: the context's index is on another device");
  if (max_len == 0 || max_len > 65535) return fail(NIMBLE_E_INVALID, "nimble_sharded_begin: bad max_len");
  nimble_comm::Shard &sh = c->shard[rank];
  sh.ctx = ctx;
  sh.prm = *p;
  sh.paired = paired ? 1 : 0;
  sh.max_len = max_len;
  sh.rec_words = nimble_key_words(max_len, paired) + 2;
  sh.n_acc = 0;
  sh.open = true;
  return NIMBLE_OK;
}

// what an append is given: ASCII reads (host or device memory) or reads the host has packed to 2 bits (host memory)
struct ShardedIn {
  const uint8_t *r1 = nullptr, *r2 = nullptr;
  const uint64_t *r1_off = nullptr, *r2_off = nullptr;
  uint32_t fixed_len = 0;
  int mem = NIMBLE_MEM_HOST;
  const uint64_t *w1 = nullptr, *w2 = nullptr;  // packed form: words, lengths, words per read
  const uint32_t *l1 = nullptr, *l2 = nullptr;
  uint32_t s1 = 0, s2 = 0;
  bool packed = false;
  bool has_mates() const { return packed ? w2 != nullptr : r2 != nullptr; }
};

static int sharded_append_impl(nimble_comm *c, int rank, const ShardedIn &in, uint64_t n) {
  if (!c || rank < 0 || rank >= c->n) return fail(NIMBLE_E_INVALID, "nimble_sharded_append: bad argument");
  nimble_comm::Shard &sh = c->shard[rank];
  nimble_ctx *ctx = sh.ctx;
  const int W = c->n;
  // (a failure of one rank must not leave the others waiting at a barrier: local errors are carried to the collectives)
  int rc = sh.open ? NIMBLE_OK : fail(NIMBLE_E_INVALID, "nimble_sharded_append: no sharded call is open on this rank");
  uint32_t max_len = sh.max_len;
  if (rc == NIMBLE_OK && in.has_mates() != (sh.paired != 0)) rc = fail(NIMBLE_E_INVALID, "nimble_sharded_append: mates given for a single-end call or missing for a paired one");
  if (rc == NIMBLE_OK && !in.packed) {
    if (!in.r1_off && in.fixed_len > max_len) rc = fail(NIMBLE_E_INVALID, "nimble_sharded_append: a read longer than max_len");
    if (rc == NIMBLE_OK) rc = check_read_args(in.r1, in.r1_off, in.r2, in.r2_off, n, in.fixed_len, max_len, in.mem);
  }
  if (rc == NIMBLE_OK && in.packed && n) {
    if (!in.w1 || !in.l1 || in.s1 == 0 || (in.w2 && (!in.l2 || in.s2 == 0)))
      rc = fail(NIMBLE_E_INVALID, "nimble_sharded_append_packed: NULL buffer or zero stride");
    for (int mt = 0; rc == NIMBLE_OK && mt < (in.w2 ? 2 : 1); ++mt) {
      const uint32_t *len = mt ? in.l2 : in.l1;
      const uint64_t cap = std::min<uint64_t>(max_len, 32ULL * (mt ? in.s2 : in.s1));
      for (uint64_t i = 0; i < n; ++i)
        if (len[i] > cap) {
          rc = fail(NIMBLE_E_INVALID, "nimble_sharded_append_packed: a read longer than max_len or than its words");
          break;
        }
    }
  }
  std::vector<uint64_t> counts(W, 0);
  const uint32_t rw = sh.rec_words;
  if (rc == NIMBLE_OK && hipSetDevice(c->devices[rank]) != hipSuccess) rc = fail(NIMBLE_E_HIP, "hipSetDevice");
  if (rc == NIMBLE_OK && n) {
    if (ctx->called && !ctx->finished) rc = fail(NIMBLE_E_INVALID, "nimble_sharded_append: the context holds a call in flight");
    if (rc == NIMBLE_OK && !in.packed) rc = stage_inputs(ctx, in.r1, in.r1_off, in.r2, in.r2_off, n, in.fixed_len, sh.max_len, in.mem);
    if (rc == NIMBLE_OK) rc = setup_call(ctx, &sh.prm, n, in.has_mates(), sh.max_len, nullptr);
    if (rc == NIMBLE_OK && hipMemsetAsync((uint64_t *)ctx->b_state.p + 14, 0, 8, ctx->stream) != hipSuccess)
      rc = fail(NIMBLE_E_HIP, "hipMemsetAsync");
    if (rc == NIMBLE_OK && in.packed) {
      // the packed batch goes over in two copies per mate on the rank's stream, k_pack_words reads it there
      const uint64_t *d_words[2] = {nullptr, nullptr};
      const uint32_t *d_len[2] = {nullptr, nullptr};
      for (int mt = 0; rc == NIMBLE_OK && mt < (in.w2 ? 2 : 1); ++mt) {
        const uint64_t wbytes = n * (uint64_t)(mt ? in.s2 : in.s1) * 8;
        rc = ctx->b_stage[0][mt].ensure(std::max<uint64_t>(wbytes, 16), &ctx->bytes);
        if (rc == NIMBLE_OK) rc = ctx->b_stage_off[0][mt].ensure(std::max<uint64_t>(n * 4, 16), &ctx->bytes);
        if (rc == NIMBLE_OK &&
            (hipMemcpyAsync(ctx->b_stage[0][mt].p, mt ? in.w2 : in.w1, wbytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
             hipMemcpyAsync(ctx->b_stage_off[0][mt].p, mt ? in.l2 : in.l1, n * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess))
          rc = fail(NIMBLE_E_HIP, "nimble_sharded_append_packed: copy to the device failed");
        d_words[mt] = ctx->b_stage[0][mt].as<uint64_t>();
        d_len[mt] = ctx->b_stage_off[0][mt].as<uint32_t>();
      }
      if (rc == NIMBLE_OK)
        launch_pack_words(ctx->stream, d_words[0], d_len[0], in.s1, d_words[1], d_len[1], in.s2, sh.max_len,
                          ctx->prm.min_read_length, ctx->b_plog.as<double>(), ctx->cb);
    } else if (rc == NIMBLE_OK) {
      launch_pack(ctx->stream, ctx->in_r[0], ctx->in_off[0], ctx->in_r[1], ctx->in_off[1], ctx->in_fixed_len, ctx->in_max_len,
                  ctx->prm.min_read_length, ctx->b_plog.as<double>(), ctx->plog_max_len, ctx->cb);
    }
    if (rc == NIMBLE_OK) {
      const uint64_t cells = (uint64_t)route_grid() * W;
      rc = ctx->b_route.ensure(cells * 4 + cells * 8 + 256 * 8, &ctx->bytes);
      if (rc == NIMBLE_OK) rc = sh.send.ensure(std::max<uint64_t>(n * rw * 8, 16), nullptr);
    }
    if (rc == NIMBLE_OK) {
      const uint64_t cells = (uint64_t)route_grid() * W;
      uint64_t *block_first = ctx->b_route.as<uint64_t>();
      uint64_t *totals = block_first + cells;
      uint32_t *block_counts = reinterpret_cast<uint32_t *>(totals + 256);
      launch_route(ctx->stream, ctx->cb, (uint32_t)W, block_counts, block_first, totals, sh.send.as<uint64_t>());
      uint64_t latch = 0;
      if (hipMemcpyAsync(counts.data(), totals, (size_t)W * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(&latch, (uint64_t *)ctx->b_state.p + 14, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = fail(NIMBLE_E_HIP, "nimble_sharded_append: routing failed");
      else if (latch != 0) {
        (void)hipMemsetAsync((uint64_t *)ctx->b_state.p + 14, 0, 8, ctx->stream);
        rc = fail(NIMBLE_E_INVALID, "offsets not monotone or a read longer than max_len (found on the device)");
      }
    }
  }
  const std::string my_error = rc == NIMBLE_OK ? std::string() : g_err;
  if (rc != NIMBLE_OK) std::fill(counts.begin(), counts.end(), 0);
  // room for what the others send: the incoming total is known only after everybody has published its counts, so the
  // exchange runs in two steps -- counts first (host memory of this process), then the records
  for (int d = 0; d < W; ++d) c->counts[(size_t)rank * W + d] = counts[d];
  int all = c->agree(rank, rc);
  uint64_t incoming = 0;
  for (int src = 0; src < W; ++src) incoming += c->counts[(size_t)src * W + rank];
  c->barrier();
  if (all != NIMBLE_OK) return rc != NIMBLE_OK ? fail(rc, my_error) : fail(all, "nimble_sharded_append: another rank failed");
  if (sh.n_acc + incoming > sh.acc_cap) {  // grow the rank's store of owned records (contents kept)
    const uint64_t cap = std::max<uint64_t>(std::max<uint64_t>(2 * sh.acc_cap, sh.n_acc + incoming), 1u << 16);
    DevBuf nb;
    int r2c = nb.ensure(cap * rw * 8, nullptr);
    if (r2c == NIMBLE_OK && sh.n_acc &&
        (hipMemcpyAsync(nb.p, sh.acc.p, sh.n_acc * rw * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
         hipStreamSynchronize(ctx->stream) != hipSuccess))
      r2c = NIMBLE_E_HIP;
    if (r2c == NIMBLE_OK) {
      sh.acc.release();
      sh.acc = nb;
      sh.acc_cap = cap;
    } else {
      nb.release();
      rc = fail(r2c, "nimble_sharded_append: out of device memory for the owned records");
    }
  }
  all = c->agree(rank, rc);
  if (all != NIMBLE_OK) return rc != NIMBLE_OK ? rc : fail(all, "nimble_sharded_append: another rank failed");
  uint64_t got = 0;
  rc = nimble_records_alltoall(c, rank, sh.send.as<uint64_t>(), counts.data(), rw, sh.acc.as<uint64_t>() + sh.n_acc * rw,
                               sh.acc_cap - sh.n_acc, &got, ctx->stream);
  if (rc) return rc;
  // the send buffer is refilled by the next append: the exchange must have left it (RCCL runs on the stream)
  if (c->rccl) HIPCHK(hipStreamSynchronize(ctx->stream));
  sh.n_acc += got;
  return NIMBLE_OK;
}

int nimble_sharded_append(nimble_comm *c, int rank, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                          const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, int mem) {
  DRAIN_STALE_HIP_ERROR();
  ShardedIn in;
  in.r1 = r1;
  in.r1_off = r1_off;
  in.r2 = r2;
  in.r2_off = r2_off;
  in.fixed_len = fixed_len;
  in.mem = mem;
  return sharded_append_impl(c, rank, in, n);
}

int nimble_sharded_append_packed(nimble_comm *c, int rank, const uint64_t *r1_words, const uint32_t *r1_len, uint32_t r1_stride,
                                 const uint64_t *r2_words, const uint32_t *r2_len, uint32_t r2_stride, uint64_t n) {
  DRAIN_STALE_HIP_ERROR();
  ShardedIn in;
  in.packed = true;
  in.w1 = r1_words;
  in.l1 = r1_len;
  in.s1 = r1_stride;
  in.w2 = r2_words;
  in.l2 = r2_len;
  in.s2 = r2_stride;
  return sharded_append_impl(c, rank, in, n);
}

int nimble_sharded_grow(nimble_comm *c, int rank, uint32_t max_len) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || rank < 0 || rank >= c->n) return fail(NIMBLE_E_INVALID, "nimble_sharded_grow: bad argument");
  nimble_comm::Shard &sh = c->shard[rank];
  if (!sh.open) return fail(NIMBLE_E_INVALID, "nimble_sharded_grow: no sharded call is open on this rank");
  if (max_len < sh.max_len || max_len > 65535) return fail(NIMBLE_E_INVALID, "nimble_sharded_grow: bad max_len");
  const uint32_t rw_old = sh.rec_words, rw_new = nimble_key_words(max_len, sh.paired) + 2;
  sh.max_len = max_len;
  if (rw_new == rw_old) return NIMBLE_OK;
  HIPCHK(hipSetDevice(c->devices[rank]));
  hipStream_t s = sh.ctx->stream;
  if (sh.n_acc) {
    // rows of rw_old words -> rows of rw_new: the key words keep their place, zero words follow them (the words behind a
    // key are zero in every record), hash and lengths move to the end of the row
    DevBuf nb;
    const uint64_t cap = std::max<uint64_t>(sh.acc_cap, sh.n_acc);
    int rc = nb.ensure(cap * rw_new * 8, nullptr);
    if (rc) return rc;
    const size_t kw_old = (size_t)(rw_old - 2) * 8;
    hipError_t e = hipMemsetAsync(nb.p, 0, sh.n_acc * rw_new * 8, s);
    if (e == hipSuccess && kw_old)
      e = hipMemcpy2DAsync(nb.p, (size_t)rw_new * 8, sh.acc.p, (size_t)rw_old * 8, kw_old, sh.n_acc, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess)
      e = hipMemcpy2DAsync((uint8_t *)nb.p + (size_t)(rw_new - 2) * 8, (size_t)rw_new * 8, (const uint8_t *)sh.acc.p + kw_old,
                           (size_t)rw_old * 8, 16, sh.n_acc, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      nb.release();
      return fail(NIMBLE_E_HIP, std::string("nimble_sharded_grow: ") + hipGetErrorString(e));
    }
    sh.acc.release();
    sh.acc = nb;
    sh.acc_cap = cap;
  } else {
    sh.acc_cap = sh.acc_cap * rw_old / rw_new;  // (the store is empty: only its capacity in records changes)
  }
  sh.rec_words = rw_new;
  return NIMBLE_OK;
}

int nimble_sharded_abort(nimble_comm *c, int rank) {
  if (!c || rank < 0 || rank >= c->n) return fail(NIMBLE_E_INVALID, "nimble_sharded_abort: bad argument");
  nimble_comm::Shard &sh = c->shard[rank];
  sh.open = false;
  sh.n_acc = 0;
  return NIMBLE_OK;
}

// ---- successive score::calls over reads spread across the ranks, software-pipelined.  For batch b of a rank:
//        P(b)  pack + route by key hash          launch stream (utility context)
//        X(b)  all-to-all of the routed records  the rank's exchange stream, beside C(b-1)
//        C(b)  the call over the records the rank received     launch stream, call context b % 2
//      submit(b) enqueues P(b) and, right behind it, C(b-1) (the launch stream waits for X(b-1)'s event, not the host); the host
//      waits for the routing of b alone (an event behind P(b)) and issues X(b).  Nothing here waits for a call: the caller reads
//      the results of the context `launched` names (nimble_histogram ...) whenever it likes -- before it submits twice more.
int nimble_steps_begin(nimble_comm *c, int rank, nimble_ctx *call0, nimble_ctx *call1, nimble_ctx *util,
                       const nimble_align_params *p, int paired, uint32_t max_len) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || rank < 0 || rank >= c->n || !call0 || !call1 || !util || !p) return fail(NIMBLE_E_INVALID, "nimble_steps_begin: bad argument");
  if (call0->stream != call1->stream || call0->stream != util->stream)
    return fail(NIMBLE_E_INVALID, "nimble_steps_begin: the three contexts must launch on one stream");
  if (call0->ix->device != c->devices[rank]) return fail(NIMBLE_E_INVALID, "nimble_steps_begin: the contexts' index is on another device");
  if (max_len == 0 || max_len > 65535) return fail(NIMBLE_E_INVALID, "nimble_steps_begin: bad max_len");
  HIPCHK(hipSetDevice(c->devices[rank]));
  nimble_comm::Shard &sh = c->shard[rank];
  nimble_comm::Shard::Steps &st = sh.st;
  if (st.open) return fail(NIMBLE_E_INVALID, "nimble_steps_begin: already open on this rank");
  if (!st.xstream) HIPCHK(hipStreamCreateWithFlags(&st.xstream, hipStreamNonBlocking));
  if (!st.ev_routed) HIPCHK(hipEventCreateWithFlags(&st.ev_routed, hipEventDisableTiming));
  for (int k = 0; k < 2; ++k)
    if (!st.ev_x[k]) HIPCHK(hipEventCreateWithFlags(&st.ev_x[k], hipEventDisableTiming));
  if (!st.p_counts) HIPCHK(hipHostMalloc((void **)&st.p_counts, (size_t)(c->n + 1) * 8, hipHostMallocDefault));
  st.call[0] = call0;
  st.call[1] = call1;
  st.util = util;
  sh.prm = *p;
  sh.paired = paired ? 1 : 0;
  sh.max_len = max_len;
  sh.rec_words = nimble_key_words(max_len, paired) + 2;
  st.b = 0;
  st.pending = false;
  st.open = true;
  return NIMBLE_OK;
}

// launch C(b - 1): the call over what the exchange of batch b - 1 delivered
static int steps_launch_pending(nimble_comm *c, int rank, nimble_ctx **launched) {
  nimble_comm::Shard &sh = c->shard[rank];
  nimble_comm::Shard::Steps &st = sh.st;
  if (launched) *launched = nullptr;
  if (!st.pending) return NIMBLE_OK;
  const int k = (int)((st.b - 1) & 1);
  nimble_ctx *ctx = st.call[k];
  if (ctx->called && !ctx->finished)
    return fail(NIMBLE_E_INVALID, "nimble_steps: the results of the call before last were not fetched (its context is still busy)");
  HIPCHK(hipStreamWaitEvent(ctx->stream, st.ev_x[k], 0));
  int rc = nimble_call_records(ctx, &sh.prm, st.recv[k].as<uint64_t>(), st.n_recv[k], sh.max_len, sh.paired);
  if (rc) return rc;
  st.pending = false;
  if (launched) *launched = ctx;
  return NIMBLE_OK;
}

int nimble_steps_submit(nimble_comm *c, int rank, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                        const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, int mem, nimble_ctx **launched) {
  DRAIN_STALE_HIP_ERROR();
  if (launched) *launched = nullptr;
  if (!c || rank < 0 || rank >= c->n) return fail(NIMBLE_E_INVALID, "nimble_steps_submit: bad argument");
  nimble_comm::Shard &sh = c->shard[rank];
  nimble_comm::Shard::Steps &st = sh.st;
  const int W = c->n;
  const int k = (int)(st.b & 1);
  nimble_ctx *u = st.util;
  // (a failure of one rank must not leave the others waiting at a barrier: local errors are carried to the collectives)
  int rc = st.open ? NIMBLE_OK : fail(NIMBLE_E_INVALID, "nimble_steps_submit: not open on this rank");
  if (rc == NIMBLE_OK && (r2 != nullptr) != (sh.paired != 0)) rc = fail(NIMBLE_E_INVALID, "nimble_steps_submit: mates given for a single-end call or missing for a paired one");
  if (rc == NIMBLE_OK && !r1_off && fixed_len > sh.max_len) rc = fail(NIMBLE_E_INVALID, "nimble_steps_submit: a read longer than max_len");
  if (rc == NIMBLE_OK) rc = check_read_args(r1, r1_off, r2, r2_off, n, fixed_len, sh.max_len, mem);
  if (rc == NIMBLE_OK && hipSetDevice(c->devices[rank]) != hipSuccess) rc = fail(NIMBLE_E_HIP, "hipSetDevice");
  const uint32_t rw = sh.rec_words;
  std::vector<uint64_t> counts(W, 0);
  // ---- P(b)
  if (rc == NIMBLE_OK && n) {
    rc = stage_inputs(u, r1, r1_off, r2, r2_off, n, fixed_len, sh.max_len, mem);
    if (rc == NIMBLE_OK) rc = setup_call(u, &sh.prm, n, r2 != nullptr, sh.max_len, nullptr);
    const uint64_t cells = (uint64_t)route_grid() * W;
    if (rc == NIMBLE_OK) rc = u->b_route.ensure(cells * 4 + cells * 8 + 256 * 8, &u->bytes);
    if (rc == NIMBLE_OK) rc = st.send[k].ensure(std::max<uint64_t>(n * rw * 8, 16), nullptr);
    if (rc == NIMBLE_OK) {
      if (hipMemsetAsync((uint64_t *)u->b_state.p + 14, 0, 8, u->stream) != hipSuccess) rc = fail(NIMBLE_E_HIP, "hipMemsetAsync");
    }
    if (rc == NIMBLE_OK) {
      launch_pack(u->stream, u->in_r[0], u->in_off[0], u->in_r[1], u->in_off[1], u->in_fixed_len, u->in_max_len,
                  u->prm.min_read_length, u->b_plog.as<double>(), u->plog_max_len, u->cb);
      uint64_t *block_first = u->b_route.as<uint64_t>();
      uint64_t *totals = block_first + cells;
      uint32_t *block_counts = reinterpret_cast<uint32_t *>(totals + 256);
      launch_route(u->stream, u->cb, (uint32_t)W, block_counts, block_first, totals, st.send[k].as<uint64_t>());
      if (hipMemcpyAsync(st.p_counts, totals, (size_t)W * 8, hipMemcpyDeviceToHost, u->stream) != hipSuccess ||
          hipMemcpyAsync(st.p_counts + W, (uint64_t *)u->b_state.p + 14, 8, hipMemcpyDeviceToHost, u->stream) != hipSuccess ||
          hipGetLastError() != hipSuccess)
        rc = fail(NIMBLE_E_HIP, "nimble_steps_submit: pack / route launch failed");
    }
  }
  if (hipEventRecord(st.ev_routed, u ? u->stream : nullptr) != hipSuccess && rc == NIMBLE_OK) rc = fail(NIMBLE_E_HIP, "hipEventRecord");
  // ---- C(b - 1) right behind P(b) on the launch stream
  if (rc == NIMBLE_OK) rc = steps_launch_pending(c, rank, launched);
  // ---- the host waits for the routing of b alone
  if (hipEventSynchronize(st.ev_routed) != hipSuccess && rc == NIMBLE_OK) rc = fail(NIMBLE_E_HIP, "nimble_steps_submit: routing failed");
  if (rc == NIMBLE_OK && n) {
    if (st.p_counts[W] != 0) rc = fail(NIMBLE_E_INVALID, "offsets not monotone or a read longer than max_len (found on the device)");
    else std::copy(st.p_counts, st.p_counts + W, counts.begin());
  }
  const std::string my_error = rc == NIMBLE_OK ? std::string() : g_err;
  // ---- room for what the others send (the incoming total is known once everybody has published its counts)
  for (int d = 0; d < W; ++d) c->counts[(size_t)rank * W + d] = counts[d];
  int all = c->agree(rank, rc);
  uint64_t incoming = 0;
  for (int src = 0; src < W; ++src) incoming += c->counts[(size_t)src * W + rank];
  c->barrier();
  if (all != NIMBLE_OK) return rc != NIMBLE_OK ? fail(rc, my_error) : fail(all, "nimble_steps_submit: another rank failed");
  if (incoming > st.recv_cap[k]) {  // (C(b - 2), the last reader of this buffer, has finished: P(b) stood behind it)
    const uint64_t cap = std::max<uint64_t>(incoming + incoming / 8, 1u << 16);
    rc = st.recv[k].ensure(cap * rw * 8, nullptr);
    if (rc == NIMBLE_OK) st.recv_cap[k] = cap;
  }
  all = c->agree(rank, rc);
  if (all != NIMBLE_OK) return rc != NIMBLE_OK ? rc : fail(all, "nimble_steps_submit: another rank failed");
  // ---- X(b) beside C(b - 1)
  uint64_t got = 0;
  rc = nimble_records_alltoall(c, rank, st.send[k].as<uint64_t>(), counts.data(), rw, st.recv[k].as<uint64_t>(), st.recv_cap[k],
                               &got, st.xstream);
  if (rc) return rc;
  HIPCHK(hipEventRecord(st.ev_x[k], st.xstream));
  st.n_recv[k] = got;
  st.pending = true;
  st.b += 1;
  return NIMBLE_OK;
}

int nimble_steps_flush(nimble_comm *c, int rank, nimble_ctx **launched) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || rank < 0 || rank >= c->n) return fail(NIMBLE_E_INVALID, "nimble_steps_flush: bad argument");
  if (!c->shard[rank].st.open) return fail(NIMBLE_E_INVALID, "nimble_steps_flush: not open on this rank");
  HIPCHK(hipSetDevice(c->devices[rank]));
  return steps_launch_pending(c, rank, launched);
}

int nimble_steps_end(nimble_comm *c, int rank) {
  if (!c || rank < 0 || rank >= c->n) return fail(NIMBLE_E_INVALID, "nimble_steps_end: bad argument");
  nimble_comm::Shard::Steps &st = c->shard[rank].st;
  if (st.xstream) {
    (void)hipSetDevice(c->devices[rank]);
    (void)hipStreamSynchronize(st.xstream);
  }
  st.open = false;
  st.pending = false;
  return NIMBLE_OK;
}

int nimble_sharded_end(nimble_comm *c, int rank, uint64_t *n_owned) {
  DRAIN_STALE_HIP_ERROR();
  if (!c || rank < 0 || rank >= c->n) return fail(NIMBLE_E_INVALID, "nimble_sharded_end: bad argument");
  nimble_comm::Shard &sh = c->shard[rank];
  if (!sh.open) return fail(NIMBLE_E_INVALID, "nimble_sharded_end: no sharded call is open on this rank");
  sh.open = false;
  if (n_owned) *n_owned = sh.n_acc;
  return nimble_call_records(sh.ctx, &sh.prm, sh.acc.as<uint64_t>(), sh.n_acc, sh.max_len, sh.paired);
}

}  // extern "C"
