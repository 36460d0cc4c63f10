// host_capi.cpp -- C ABI of the host mirror (include/nimble_host.h).
#include <algorithm>
#include <cstring>

#include "../../include/nimble_hip.h"
#include "../../include/nimble_host.h"
#include "nimble_host.hpp"

using namespace nimble;

namespace {
thread_local std::string g_err;

template <class F>
int guarded(F &&f) {
  try {
    f();
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  } catch (...) {  // (nothing is thrown across the C ABI, whatever it is)
    g_err = "unknown failure";
    return -1;
  }
}

std::string join_tab(const std::vector<std::string> &v) {
  std::string s;
  for (size_t i = 0; i < v.size(); ++i) {
    if (i) s.push_back('\t');
    s += v[i];
  }
  return s;
}

std::vector<std::string> split_on(const char *s, char sep, bool keep_single_empty) {
  std::vector<std::string> v;
  if (!s || (!*s && !keep_single_empty)) return v;
  std::string cur;
  for (const char *p = s; *p; ++p) {
    if (*p == sep) {
      v.push_back(cur);
      cur.clear();
    } else {
      cur.push_back(*p);
    }
  }
  v.push_back(cur);
  return v;
}
int copy_text(const std::string &j, char *out, int cap) {
  if ((int)j.size() + 1 > cap) throw Panic("output buffer too small");
  memcpy(out, j.c_str(), j.size() + 1);
  return (int)j.size();
}
std::string join_nl(const std::vector<std::string> &v) {
  std::string s;
  for (size_t i = 0; i < v.size(); ++i) {
    if (i) s.push_back('\n');
    s += v[i];
  }
  return s;
}
}  // namespace

struct nimble_library {
  align::AlignFilterConfig cfg;
  reference_library::Reference ref;
  std::unique_ptr<align::PseudoAligner> index;
  std::unique_ptr<align::CallStream> stream;  // nimble_score_stream_begin .. _end
  bool pending[4] = {false, false, false, false};  // nimble_score_call_begin without its _end yet
  uint64_t pending_n[4] = {0, 0, 0, 0};
};
struct nimble_umi_rows {
  align::UmiOutput out;
  std::vector<std::string> joined;
};
struct nimble_rows {
  align::CallOutput out;  // refs only (the index runs in light mode): no per-call string copies
};

extern "C" {

const char *nimble_host_last_error(void) { return g_err.c_str(); }

int nimble_library_load(const char *path, int strand_filter, nimble_library **out) {
  *out = nullptr;
  return guarded([&] {
    auto pr = reference_library::get_reference_library(path, (align::LibraryChemistry)strand_filter);
    nimble_library *l = new nimble_library();
    l->cfg = pr.first;
    l->ref = std::move(pr.second);
    *out = l;
  });
}
int nimble_library_parse(const char *text, int strand_filter, nimble_library **out) {
  *out = nullptr;
  return guarded([&] {
    auto pr = reference_library::parse_reference_library(text, (align::LibraryChemistry)strand_filter);
    nimble_library *l = new nimble_library();
    l->cfg = pr.first;
    l->ref = std::move(pr.second);
    *out = l;
  });
}
int nimble_library_from_table(int n_cols, const char *const *headers, const int *rows_of, const char *const *cells,
                              int group_on, int sequence_name_idx, int sequence_idx, nimble_library **out) {
  *out = nullptr;
  return guarded([&] {
    if (n_cols < 0 || group_on < 0 || sequence_name_idx < 0 || sequence_idx < 0) throw Panic("nimble_library_from_table: bad argument");
    std::unique_ptr<nimble_library> l(new nimble_library());
    size_t at = 0;
    for (int c = 0; c < n_cols; ++c) {
      l->ref.headers.push_back(headers[c]);
      l->ref.columns.emplace_back();
      for (int r = 0; r < rows_of[c]; ++r) l->ref.columns.back().push_back(cells[at++]);
    }
    l->ref.group_on = (size_t)group_on;
    l->ref.sequence_name_idx = (size_t)sequence_name_idx;
    l->ref.sequence_idx = (size_t)sequence_idx;
    l->cfg.max_hits_to_report = 10;  // (a default config: the table is what the caller is after)
    *out = l.release();
  });
}
void nimble_library_free(nimble_library *l) { delete l; }

int nimble_library_get_config(const nimble_library *l, nimble_host_config *o) {
  const auto &c = l->cfg;
  o->reference_genome_size = c.reference_genome_size;
  o->score_percent = c.score_percent;
  o->score_threshold = c.score_threshold;
  o->num_mismatches = c.num_mismatches;
  o->discard_nonzero_mismatch = c.discard_nonzero_mismatch;
  o->discard_multiple_matches = c.discard_multiple_matches;
  o->score_filter = c.score_filter;
  o->intersect_level = (int)c.intersect_level;
  o->require_valid_pair = c.require_valid_pair;
  o->strand_filter = (int)c.strand_filter;
  o->discard_multi_hits = c.discard_multi_hits;
  o->max_hits_to_report = c.max_hits_to_report;
  o->trim_strictness = c.trim_strictness;
  o->trim_target_length = c.trim_target_length;
  return 0;
}
int nimble_library_set_config(nimble_library *l, const nimble_host_config *i) {
  align::AlignFilterConfig c = l->cfg;
  c.reference_genome_size = i->reference_genome_size;
  c.score_percent = i->score_percent;
  c.score_threshold = i->score_threshold;
  c.num_mismatches = i->num_mismatches;
  c.discard_nonzero_mismatch = i->discard_nonzero_mismatch != 0;
  c.discard_multiple_matches = i->discard_multiple_matches != 0;
  c.score_filter = i->score_filter;
  c.intersect_level = (align::IntersectLevel)i->intersect_level;
  c.require_valid_pair = i->require_valid_pair != 0;
  c.strand_filter = (align::LibraryChemistry)i->strand_filter;
  c.discard_multi_hits = i->discard_multi_hits;
  c.max_hits_to_report = i->max_hits_to_report;
  c.trim_strictness = i->trim_strictness;
  c.trim_target_length = i->trim_target_length;
  return guarded([&] {
    reference_library::sanity_check_align_config(c);
    l->cfg = c;
  });
}
int nimble_library_n_rows(const nimble_library *l) { return l->ref.columns.empty() ? 0 : (int)l->ref.columns[0].size(); }
int nimble_library_n_cols(const nimble_library *l) { return (int)l->ref.columns.size(); }
int nimble_library_group_on(const nimble_library *l) { return (int)l->ref.group_on; }
void nimble_library_set_group_on(nimble_library *l, int col) { l->ref.group_on = (size_t)col; }
int nimble_library_sequence_name_idx(const nimble_library *l) { return (int)l->ref.sequence_name_idx; }
int nimble_library_sequence_idx(const nimble_library *l) { return (int)l->ref.sequence_idx; }
const char *nimble_library_header(const nimble_library *l, int c) { return l->ref.headers.at((size_t)c).c_str(); }
const char *nimble_library_cell(const nimble_library *l, int c, int r) {
  return l->ref.columns.at((size_t)c).at((size_t)r).c_str();
}
int nimble_library_push_column(nimble_library *l, const char *header, const char *const *values, int n) {
  l->ref.headers.push_back(header);
  std::vector<std::string> col;
  for (int i = 0; i < n; ++i) col.push_back(values[i]);
  l->ref.columns.push_back(col);
  return (int)l->ref.columns.size() - 1;
}
int nimble_library_build_index(nimble_library *l, int device) {
  return guarded([&] {
    auto data = utils::get_reference_sequence_data(l->ref);
    l->index = align::PseudoAligner::build_index(data.first, data.second, device);
    l->index->set_light_rows(true);
  });
}
void *nimble_library_index(nimble_library *l) { return l->index ? (void *)l->index->index() : nullptr; }
void *nimble_library_ctx(nimble_library *l) { return l->index ? (void *)l->index->ctx() : nullptr; }
void *nimble_library_ctx_slot(nimble_library *l, int slot) {
  void *p = nullptr;
  if (l->index && slot >= 0 && slot <= 3) guarded([&] { p = (void *)l->index->ctx(slot); });
  return p;
}

// slots that may hold a call in flight (2 is the utility context)
static bool call_slot(int slot) { return slot == 0 || slot == 1 || slot == 3; }

static nimble_rows *make_rows(align::CallOutput &&o) {
  nimble_rows *r = new nimble_rows();
  r->out = std::move(o);
  return r;
}

int nimble_score_call(nimble_library *l, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                      const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                      nimble_rows **out) {
  *out = nullptr;
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_call: the library has no index (call nimble_library_build_index)");
    align::ReadBatch b1, b2;
    b1.bases = r1;
    b1.offsets = r1_off;
    b1.n = n;
    b1.fixed_len = fixed_len;
    b1.max_len = max_len;
    b1.device = mem == NIMBLE_MEM_DEVICE;
    b2 = b1;
    b2.bases = r2;
    b2.offsets = r2_off;
    // rows come back as references in Vec<String> order (the order of utils::sort_score_vector)
    *out = make_rows(align::get_calls(b1, r2 ? &b2 : nullptr, *l->index, l->ref, l->cfg));
  });
}

int nimble_score_call_begin(nimble_library *l, int slot, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                            const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem) {
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_call_begin: the library has no index");
    if (!call_slot(slot)) throw Panic("nimble_score_call_begin: slot must be 0, 1 or 3");
    if (l->pending[slot]) throw Panic("nimble_score_call_begin: the slot already holds a call (end it first)");
    align::ReadBatch b1, b2;
    b1.bases = r1;
    b1.offsets = r1_off;
    b1.n = n;
    b1.fixed_len = fixed_len;
    b1.max_len = max_len;
    b1.device = mem == NIMBLE_MEM_DEVICE;
    b2 = b1;
    b2.bases = r2;
    b2.offsets = r2_off;
    align::begin_calls(b1, r2 ? &b2 : nullptr, *l->index, l->cfg, slot);
    l->pending[slot] = true;
    l->pending_n[slot] = n;
  });
}

int nimble_score_call_begin_words(nimble_library *l, int slot, const uint64_t *r1_words, const uint32_t *r1_len,
                                  uint32_t r1_stride, const uint64_t *r2_words, const uint32_t *r2_len, uint32_t r2_stride,
                                  uint64_t n, uint32_t max_len, int mem) {
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_call_begin_words: the library has no index");
    if (!call_slot(slot)) throw Panic("nimble_score_call_begin_words: slot must be 0, 1 or 3");
    if (l->pending[slot]) throw Panic("nimble_score_call_begin_words: the slot already holds a call (end it first)");
    if (!r1_words || !r1_len || r1_stride == 0 || ((r2_words != nullptr) != (r2_len != nullptr)) || (r2_words && r2_stride == 0))
      throw Panic("nimble_score_call_begin_words: bad argument");
    align::ReadBatch b1, b2;
    b1.n = n;
    b1.max_len = max_len;
    b1.device = mem == NIMBLE_MEM_DEVICE;
    b2 = b1;
    b1.words = r1_words;
    b1.lens = r1_len;
    b1.stride = r1_stride;
    b2.words = r2_words;
    b2.lens = r2_len;
    b2.stride = r2_stride;
    align::begin_calls(b1, r2_words ? &b2 : nullptr, *l->index, l->cfg, slot);
    l->pending[slot] = true;
    l->pending_n[slot] = n;
  });
}

int nimble_score_call_end(nimble_library *l, int slot, nimble_rows **out) {
  *out = nullptr;
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_call_end: the library has no index");
    if (!call_slot(slot) || !l->pending[slot]) throw Panic("nimble_score_call_end: no call was begun in this slot");
    // (a refused end -- e.g. a deferred-dedup call whose verdicts are not in yet -- leaves the slot open)
    *out = make_rows(align::end_calls(l->pending_n[slot], *l->index, l->ref, l->cfg, slot));
    l->pending[slot] = false;
  });
}

int nimble_score_call_umis(nimble_library *l, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                           const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                           const nimble_umi_extra *extra, int want_per_read, nimble_umi_rows **out) {
  *out = nullptr;
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_call_umis: the library has no index");
    align::ReadBatch b1, b2;
    b1.bases = r1;
    b1.offsets = r1_off;
    b1.n = n;
    b1.fixed_len = fixed_len;
    b1.max_len = max_len;
    b1.device = mem == NIMBLE_MEM_DEVICE;
    b2 = b1;
    b2.bases = r2;
    b2.offsets = r2_off;
    align::UmiExtras ex;
    if (extra) {
      ex.segment = extra->segment;
      ex.n_segments = extra->n_segments;
      for (int m = 0; m < 2; ++m) {
        ex.qual[m] = extra->qual[m];
        ex.skip[m] = extra->skip[m];
      }
    }
    std::unique_ptr<nimble_umi_rows> r(new nimble_umi_rows());
    r->out = align::get_calls_umis(b1, r2 ? &b2 : nullptr, ex, *l->index, l->ref, l->cfg, want_per_read != 0);
    for (auto &row : r->out.rows) r->joined.push_back(join_tab(r->out.features(row)));
    *out = r.release();
  });
}
void nimble_umi_rows_free(nimble_umi_rows *r) { delete r; }
uint64_t nimble_umi_rows_count(const nimble_umi_rows *r) { return r->out.rows.size(); }
const char *nimble_umi_rows_get(const nimble_umi_rows *r, uint64_t i, uint32_t *segment, int32_t *count,
                                uint32_t *representative) {
  const align::UmiRow &row = r->out.rows.at(i);
  if (segment) *segment = row.segment;
  if (count) *count = row.count;
  if (representative) *representative = row.representative;
  return r->joined.at(i).c_str();
}
uint64_t nimble_umi_rows_reads(const nimble_umi_rows *r) { return r->out.per_read.size(); }
int nimble_umi_rows_filter(const nimble_umi_rows *r, uint64_t read, int32_t out[5]) {
  if (read >= r->out.per_read.size()) return -1;
  const align::FilterRecord &f = r->out.per_read[read];
  out[0] = (int32_t)f.r1;
  out[1] = (int32_t)f.score1;
  out[2] = (int32_t)f.r2;
  out[3] = (int32_t)f.score2;
  out[4] = (int32_t)f.triage;
  return 0;
}

int nimble_score_stream_begin(nimble_library *l, int paired, uint32_t max_len, uint64_t capacity_hint) {
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_stream_begin: the library has no index");
    if (l->stream) throw Panic("nimble_score_stream_begin: a stream is already open");
    l->stream.reset(new align::CallStream(*l->index, l->cfg, paired != 0, max_len, capacity_hint));
  });
}

int nimble_score_stream_append(nimble_library *l, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                               const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, int mem) {
  return guarded([&] {
    if (!l->stream) throw Panic("nimble_score_stream_append: no stream is open");
    align::ReadBatch b1, b2;
    b1.bases = r1;
    b1.offsets = r1_off;
    b1.n = n;
    b1.fixed_len = fixed_len;
    b1.device = mem == NIMBLE_MEM_DEVICE;
    b2 = b1;
    b2.bases = r2;
    b2.offsets = r2_off;
    l->stream->append(b1, r2 ? &b2 : nullptr);
  });
}

int nimble_score_stream_end(nimble_library *l, nimble_rows **out) {
  *out = nullptr;
  return guarded([&] {
    if (!l->stream) throw Panic("nimble_score_stream_end: no stream is open");
    std::unique_ptr<align::CallStream> st = std::move(l->stream);
    *out = make_rows(st->finish(l->ref));
  });
}

int nimble_library_pack(nimble_library *l, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                        const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                        const nimble_packed *out) {
  return guarded([&] {
    if (!l->index) throw Panic("nimble_library_pack: the library has no index");
    align::ReadBatch b1, b2;
    b1.bases = r1;
    b1.offsets = r1_off;
    b1.n = n;
    b1.fixed_len = fixed_len;
    b1.max_len = max_len;
    b1.device = mem == NIMBLE_MEM_DEVICE;
    b2 = b1;
    b2.bases = r2;
    b2.offsets = r2_off;
    align::pack_reads(b1, r2 ? &b2 : nullptr, *l->index, l->cfg, *out);
  });
}

int nimble_library_pack_slot(nimble_library *l, int slot, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                             const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                             const nimble_packed *out) {
  return guarded([&] {
    if (!l->index) throw Panic("nimble_library_pack_slot: the library has no index");
    align::ReadBatch b1, b2;
    b1.bases = r1;
    b1.offsets = r1_off;
    b1.n = n;
    b1.fixed_len = fixed_len;
    b1.max_len = max_len;
    b1.device = mem == NIMBLE_MEM_DEVICE;
    b2 = b1;
    b2.bases = r2;
    b2.offsets = r2_off;
    align::pack_reads(b1, r2 ? &b2 : nullptr, *l->index, l->cfg, *out, slot);
  });
}

int nimble_score_call_packed_begin(nimble_library *l, int slot, const nimble_packed *in, uint64_t n, uint32_t max_len) {
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_call_packed_begin: the library has no index");
    if (!call_slot(slot)) throw Panic("nimble_score_call_packed_begin: slot must be 0, 1 or 3");
    if (l->pending[slot]) throw Panic("nimble_score_call_packed_begin: the slot already holds a call (end it first)");
    align::begin_calls_packed(*in, n, max_len, *l->index, l->cfg, slot);
    l->pending[slot] = true;
    l->pending_n[slot] = n;
  });
}

int nimble_score_call_records_begin(nimble_library *l, int slot, const uint64_t *records, uint64_t n, uint32_t max_len,
                                    int paired) {
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_call_records_begin: the library has no index");
    if (!call_slot(slot)) throw Panic("nimble_score_call_records_begin: slot must be 0, 1 or 3");
    if (l->pending[slot]) throw Panic("nimble_score_call_records_begin: the slot already holds a call (end it first)");
    align::begin_calls_records(records, n, max_len, paired != 0, *l->index, l->cfg, slot);
    l->pending[slot] = true;
    l->pending_n[slot] = n;
  });
}

int nimble_score_call_packed(nimble_library *l, const nimble_packed *in, uint64_t n, uint32_t max_len,
                             nimble_rows **out) {
  *out = nullptr;
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_call_packed: the library has no index");
    *out = make_rows(align::get_calls_packed(*in, n, max_len, *l->index, l->ref, l->cfg));
  });
}

int nimble_score_call_fastq(nimble_library *l, const char *p1, const char *p2, nimble_rows **out) {
  *out = nullptr;
  return guarded([&] {
    if (!l->index) throw Panic("nimble_score_call_fastq: the library has no index");
    parse::fastq::FastqData d1 = parse::fastq::read_fastq(p1, false), d2;
    align::ReadBatch b1, b2;
    b1.bases = d1.bases.data();
    b1.offsets = d1.offsets.data();
    b1.n = d1.n();
    b1.max_len = d1.max_len;
    if (p2) {
      d2 = parse::fastq::read_fastq(p2, true);
      if (d2.n() < d1.n()) throw Panic("Error -- read and reverse read files do not have matching lengths: ");
      b2.bases = d2.bases.data();
      b2.offsets = d2.offsets.data();
      b2.n = d1.n();
      b2.max_len = d2.max_len;
    }
    *out = make_rows(align::get_calls(b1, p2 ? &b2 : nullptr, *l->index, l->ref, l->cfg));
  });
}

void nimble_rows_free(nimble_rows *r) { delete r; }
uint64_t nimble_rows_signature(const nimble_rows *r) {
  // the callset ids index the index's memo, which only ever appends: equal id lists = equal keys in equal order
  uint64_t h = 1469598103934665603ULL ^ (uint64_t)r->out.refs.size();
  for (int32_t id : r->out.refs.ids) h = (h ^ (uint64_t)(uint32_t)id) * 1099511628211ULL;
  return h ^ (uint64_t)(uintptr_t)r->out.refs.memo.get();
}
void nimble_rows_counts(const nimble_rows *r, int64_t *out) {
  for (size_t i = 0; i < r->out.refs.size(); ++i) out[i] = r->out.refs.counts[i];
}
uint64_t nimble_rows_count(const nimble_rows *r) { return r->out.refs.size(); }
const char *nimble_rows_get(const nimble_rows *r, uint64_t i, int32_t *count) {
  *count = r->out.refs.counts.at(i);
  return r->out.refs.joined(i).c_str();
}

int nimble_fastq_process(int n_inputs, const char *const *inputs, int n_libs, nimble_library *const *libs,
                         const char *const *outputs) {
  return guarded([&] {
    std::vector<std::string> in, out;
    for (int i = 0; i < n_inputs; ++i) in.push_back(inputs[i]);
    std::vector<std::unique_ptr<align::PseudoAligner>> idx;
    std::vector<reference_library::Reference> refs;
    std::vector<align::AlignFilterConfig> cfgs;
    for (int i = 0; i < n_libs; ++i) {
      if (!libs[i]->index) throw Panic("nimble_fastq_process: a library has no index");
      idx.push_back(std::move(libs[i]->index));  // process() consumes the indices, as the reference does
      refs.push_back(libs[i]->ref);
      cfgs.push_back(libs[i]->cfg);
      out.push_back(outputs[i]);
    }
    try {
      process::fastq::process(in, idx, refs, cfgs, out);
    } catch (...) {
      for (int i = 0; i < n_libs; ++i) libs[i]->index = std::move(idx[(size_t)i]);
      throw;
    }
    for (int i = 0; i < n_libs; ++i) libs[i]->index = std::move(idx[(size_t)i]);
  });
}

int nimble_bam_process(const char *input, int n_libs, nimble_library *const *libs, const char *const *outputs, int cores,
                       int force_bam_paired) {
  return guarded([&] {
    std::vector<std::unique_ptr<align::PseudoAligner>> idx;
    std::vector<reference_library::Reference> refs;
    std::vector<align::AlignFilterConfig> cfgs;
    std::vector<std::string> out;
    for (int i = 0; i < n_libs; ++i) {
      if (!libs[i]->index) throw Panic("nimble_bam_process: a library has no index");
      idx.push_back(std::move(libs[i]->index));
      refs.push_back(libs[i]->ref);
      cfgs.push_back(libs[i]->cfg);
      out.push_back(outputs[i]);
    }
    try {
      process::bam::process({std::string(input)}, idx, refs, cfgs, out, (size_t)std::max(cores, 1), force_bam_paired != 0);
    } catch (...) {
      for (int i = 0; i < n_libs; ++i) libs[i]->index = std::move(idx[(size_t)i]);
      throw;
    }
    for (int i = 0; i < n_libs; ++i) libs[i]->index = std::move(idx[(size_t)i]);
  });
}

// The UMI groups of a BAM file as the reference's UMIReader hands them to the aligner, one text line per item (no GPU
// needed): "G <umi> <cell barcode>" opens a group, "R <sequence> <38 fields>" is a record of it (tab-separated; the QUAL
// field, raw Phred bytes, in hex).  The last group is marked "G* ..." : the reference reads it and never sends it.
int nimble_host_bam_dump(const char *input, int force_bam_paired, const char *out_path) {
  return guarded([&] {
    FILE *f = fopen(out_path, "w");
    if (!f) throw Panic("nimble_host_bam_dump: cannot write the output file");
    parse::bam::UMIReader reader(input, false, force_bam_paired != 0);
    bool has_aligned = false;
    for (;;) {
      const bool final_umi = reader.next();
      const bool dropped = final_umi && has_aligned;
      if (dropped && reader.current_umi_group.empty()) break;  // (a file of one group: nothing is left over)
      fprintf(f, "%s\t%s\t%s\n", dropped ? "G*" : "G", reader.current_umi.c_str(), reader.current_cell_barcode.c_str());
      for (size_t i = 0; i < reader.current_umi_group.size(); ++i) {
        fprintf(f, "R\t%s", reader.current_umi_group[i].c_str());
        const auto &md = reader.current_metadata_group[i];
        for (size_t k = 0; k < md.size(); ++k) {
          fputc('\t', f);
          if (k == 1) for (unsigned char c : md[k]) fprintf(f, "%02x", c);
          else fputs(md[k].c_str(), f);
        }
        fputc('\n', f);
      }
      if (dropped) break;
      has_aligned = true;
    }
    fclose(f);
  });
}

// The parallel inflate on its own (host/pgzip.cpp): the decompressed stream of a gzip file written to out_path, pieces
// resolved in order.  Diagnostic and test surface; no CRC check here (the FASTQ reader makes it).
int nimble_host_pgzip_decompress(const char *path, int threads, const char *out_path, uint64_t *n_pieces) {
  return guarded([&] {
    parse::pgzip::Reader rd(path, (unsigned)std::max(threads, 1));
    FILE *f = fopen(out_path, "wb");
    if (!f) throw Panic("nimble_host_pgzip_decompress: cannot write the output file");
    parse::pgzip::Piece p;
    std::vector<uint8_t> buf;
    uint64_t k = 0;
    try {
      while (rd.next(p)) {
        buf.resize(p.size());
        parse::pgzip::resolve(p, buf.data());
        if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), f);
        ++k;
      }
    } catch (...) {
      fclose(f);
      throw;
    }
    fclose(f);
    if (n_pieces) *n_pieces = k;
  });
}

int nimble_host_pack_reads_2bit(const uint8_t *bases, const uint64_t *off, uint64_t n, uint32_t stride, uint64_t *words,
                                uint32_t *lens) {
  return guarded([&] {
    if (n && (!bases || !off || !words || !lens || stride == 0)) throw Panic("nimble_host_pack_reads_2bit: bad argument");
    for (uint64_t i = 0; i < n; ++i)
      if (off[i + 1] < off[i] || off[i + 1] - off[i] > 32ULL * stride)
        throw Panic("nimble_host_pack_reads_2bit: a read does not fit its words");
    parse::fastq::pack_reads_2bit(bases, off, n, stride, words, lens);
  });
}

int nimble_host_reverse_comp_if_needed(const char *seq, int reverse_comp, char *out, uint64_t cap) {
  return guarded([&] {
    const std::string r = process::bam::reverse_comp_if_needed(seq, reverse_comp != 0);
    if (r.size() + 1 > cap) throw Panic("nimble_host_reverse_comp_if_needed: buffer too small");
    memcpy(out, r.c_str(), r.size() + 1);
  });
}

int nimble_host_parse_str_as_bool(const char *v, int *out) {
  return guarded([&] { *out = process::bam::parse_str_as_bool(v) ? 1 : 0; });
}

int nimble_fastq_process_sharded(int n_inputs, const char *const *inputs, nimble_library *lib, const int *devices,
                                 int n_devices, const char *output) {
  return guarded([&] {
    if (!lib || !devices || n_devices < 1) throw Panic("nimble_fastq_process_sharded: bad argument");
    std::vector<std::string> in;
    for (int i = 0; i < n_inputs; ++i) in.push_back(inputs[i]);
    const std::vector<int> dev(devices, devices + n_devices);
    auto data = utils::get_reference_sequence_data(lib->ref);
    std::vector<std::vector<std::unique_ptr<align::PseudoAligner>>> idx(1);
    for (int d : dev) idx[0].push_back(align::PseudoAligner::build_index(data.first, data.second, d));
    process::fastq::process_sharded(in, idx, {lib->ref}, {lib->cfg}, {std::string(output)}, dev);
  });
}

int nimble_multi_steps(nimble_library *const *libs, const int *devices, int world, const uint8_t *const *r1,
                       const uint8_t *const *r2, int n_sets, uint64_t n, uint32_t fixed_len, int warmup, int steps,
                       int align_grid_pct, double *ms_per_step, int *used_rccl, nimble_rows **last) {
  if (last) *last = nullptr;
  return guarded([&] {
    if (!libs || !devices || world < 1 || !r1 || n_sets < 1 || !ms_per_step || !last) throw Panic("nimble_multi_steps: bad argument");
    std::vector<std::unique_ptr<align::PseudoAligner>> idx;
    // the libraries lend their indices for the run (one per rank, each on its rank's device)
    for (int r = 0; r < world; ++r) {
      if (!libs[r] || !libs[r]->index) throw Panic("nimble_multi_steps: a library has no index");
      idx.push_back(std::move(libs[r]->index));
    }
    struct Return {
      nimble_library *const *libs;
      std::vector<std::unique_ptr<align::PseudoAligner>> &idx;
      ~Return() {
        for (size_t r = 0; r < idx.size(); ++r) libs[r]->index = std::move(idx[r]);
      }
    } give_back{libs, idx};
    std::vector<std::vector<const uint8_t *>> reads((size_t)world), mates;
    if (r2) mates.resize((size_t)world);
    for (int r = 0; r < world; ++r)
      for (int s = 0; s < n_sets; ++s) {
        reads[(size_t)r].push_back(r1[(size_t)r * n_sets + s]);
        if (r2) mates[(size_t)r].push_back(r2[(size_t)r * n_sets + s]);
      }
    const std::vector<int> dev(devices, devices + world);
    process::multi::StepsResult res = process::multi::run_steps(idx, libs[0]->ref, libs[0]->cfg, dev, reads, mates, n, fixed_len,
                                                                warmup, steps, align_grid_pct);
    *ms_per_step = res.ms_per_step;
    if (used_rccl) *used_rccl = res.rccl ? 1 : 0;
    *last = make_rows(std::move(res.last));
  });
}

int nimble_write_to_tsv(const nimble_rows *r, const char *path) {
  return guarded([&] {
    align::CallOutput copy;
    copy.refs = r->out.refs;
    copy.materialize();
    utils::write_to_tsv(copy.rows, path);
  });
}

int nimble_host_coerce(const nimble_library *l, int has1, const uint32_t *c1, int n1, int has2, const uint32_t *c2,
                       int n2, char *out, int cap) {
  int triage = -1;
  int rc = guarded([&] {
    align::Coercer co(l->ref, l->cfg);
    align::FilterReason t;
    std::vector<std::string> cs = co.coerce(has1 != 0, std::vector<uint32_t>(c1, c1 + n1), has2 != 0,
                                            std::vector<uint32_t>(c2, c2 + n2), t);
    std::string j = join_tab(cs);
    if ((int)j.size() + 1 > cap) throw Panic("output buffer too small");
    memcpy(out, j.c_str(), j.size() + 1);
    triage = (int)t;
  });
  return rc ? -1 : triage;
}
int nimble_host_parse_calls(const nimble_library *l, const char *calls, char *out, int cap) {
  return guarded([&] {
    align::Coercer co(l->ref, l->cfg);
    std::vector<std::string> lines;
    for (const auto &c : co.parse_calls(split_on(calls, '\n', false))) lines.push_back(c.first + "\t" + (c.second ? "1" : "0"));
    copy_text(join_nl(lines), out, cap);
  });
}
int nimble_host_unmap(const nimble_library *l, const char *features, uint32_t *out, int cap) {
  int n = -1;
  int rc = guarded([&] {
    align::Coercer co(l->ref, l->cfg);
    std::vector<uint32_t> ids = co.unmap(split_on(features, '\n', false));
    if ((int)ids.size() > cap) throw Panic("output buffer too small");
    for (size_t i = 0; i < ids.size(); ++i) out[i] = ids[i];
    n = (int)ids.size();
  });
  return rc ? -1 : n;
}
int nimble_host_feature_list(const nimble_library *l, const uint32_t *cls, int n, int ignore_group_rollup, char *out, int cap) {
  return guarded([&] {
    align::Coercer co(l->ref, l->cfg);
    copy_text(join_nl(co.feature_list(std::vector<uint32_t>(cls, cls + n), ignore_group_rollup != 0)), out, cap);
  });
}
int nimble_host_reference_sequence_data(const nimble_library *l, char *out, int cap) {
  return guarded([&] {
    auto data = utils::get_reference_sequence_data(l->ref);
    std::vector<std::string> lines;
    for (size_t i = 0; i < data.first.size(); ++i) lines.push_back(data.second[i] + "\t" + data.first[i]);
    copy_text(join_nl(lines), out, cap);
  });
}
int nimble_host_sort_score_vector(const char *keys, int n_rows, int32_t *order) {
  return guarded([&] {
    std::vector<std::pair<std::vector<std::string>, int32_t>> rows;
    std::vector<std::string> lines = split_on(keys, '\n', n_rows == 1);
    if ((int)lines.size() != n_rows) throw Panic("row count does not match");
    for (size_t i = 0; i < lines.size(); ++i) rows.emplace_back(split_on(lines[i].c_str(), '\t', true), (int32_t)i);
    utils::sort_score_vector(rows);
    for (size_t i = 0; i < rows.size(); ++i) order[i] = rows[i].second;
  });
}
int nimble_host_natural_lexical_cmp(const char *a, const char *b) { return utils::natural_lexical_cmp(a, b); }
double nimble_host_shannon_entropy(const char *dna) { return utils::shannon_entropy(dna); }
int nimble_host_revcomp(const char *seq, char *out) {
  return guarded([&] {
    std::string r = utils::revcomp(seq);
    memcpy(out, r.c_str(), r.size() + 1);
  });
}
uint64_t nimble_host_maxinfo(const char *q, int qlen, uint64_t target, double strictness) {
  return align::maxinfo(std::string(q, (size_t)qlen), (size_t)target, strictness);
}
int nimble_host_read_fastq(const char *path, uint64_t *n, uint64_t *bases, uint32_t *max_len) {
  return guarded([&] {
    parse::fastq::FastqData d = parse::fastq::read_fastq(path, false);
    *n = d.n();
    *bases = d.bases.size();
    *max_len = d.max_len;
  });
}
int nimble_host_read_fastq_batched(const char *path, uint64_t batch_reads, uint64_t *n, uint64_t *bases,
                                   uint32_t *max_len, uint64_t *n_batches, uint64_t *checksum) {
  return guarded([&] {
    parse::fastq::BatchReader rd(path, false, (size_t)batch_reads);
    *n = *bases = *n_batches = 0;
    *max_len = 0;
    uint64_t h = 1469598103934665603ULL;
    for (;;) {
      std::unique_ptr<parse::fastq::BatchReader::Batch> b = rd.next();
      *n += b->data.n();
      *bases += b->data.bases.size();
      *max_len = std::max(*max_len, b->data.max_len);
      ++*n_batches;
      for (uint64_t i = 0; checksum && i < b->data.n(); ++i) {  // FNV over (length, bases) of every record, in order
        h = (h ^ (b->data.offsets[i + 1] - b->data.offsets[i])) * 1099511628211ULL;
        for (uint64_t k = b->data.offsets[i]; k < b->data.offsets[i + 1]; ++k)
          h = (h ^ b->data.bases[k]) * 1099511628211ULL;
      }
      if (!b->error.empty()) {
        if (checksum) *checksum = h;
        throw Panic(b->error);
      }
      if (b->last) break;
      rd.recycle(std::move(b));
    }
    if (checksum) *checksum = h;
  });
}
int nimble_host_read_fastq_packed(const char *path, uint64_t batch_reads, uint64_t *n, uint64_t *bases, uint32_t *max_len,
                                  uint64_t *n_batches, uint64_t *checksum) {
  return guarded([&] {
    parse::fastq::BatchReader rd(path, false, (size_t)batch_reads, true);
    *n = *bases = *n_batches = 0;
    *max_len = 0;
    uint64_t h = 1469598103934665603ULL;
    for (;;) {
      std::unique_ptr<parse::fastq::BatchReader::Batch> b = rd.next();
      const uint64_t m = b->data.n();
      if (m && b->stride == 0) throw Panic("nimble_host_read_fastq_packed: a batch came without its packed form");
      *n += m;
      *max_len = std::max(*max_len, b->data.max_len);
      ++*n_batches;
      for (uint64_t i = 0; i < m; ++i) {  // FNV over (length, bases) of every record, bases as the words spell them
        const uint32_t len = b->lens[i];
        if (len != b->data.offsets[i + 1] - b->data.offsets[i]) throw Panic("nimble_host_read_fastq_packed: lengths disagree");
        *bases += len;
        h = (h ^ len) * 1099511628211ULL;
        const uint64_t *w = b->words.data() + i * (uint64_t)b->stride;
        for (uint32_t k = 0; k < len; ++k) h = (h ^ (uint64_t)"ACGT"[(w[k >> 5] >> (62 - 2 * (k & 31))) & 3]) * 1099511628211ULL;
        for (uint32_t k = len; k < 32u * b->stride; ++k)
          if ((w[k >> 5] >> (62 - 2 * (k & 31))) & 3) throw Panic("nimble_host_read_fastq_packed: bits behind the last base");
      }
      if (!b->error.empty()) {
        if (checksum) *checksum = h;
        throw Panic(b->error);
      }
      if (b->last) break;
      rd.recycle(std::move(b));
    }
    if (checksum) *checksum = h;
  });
}
const char *nimble_host_filter_reason_text(int r) { return align::to_string((align::FilterReason)r); }

}  // extern "C"
