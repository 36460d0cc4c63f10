// align.cpp -- host side of align::get_calls / score::call over the device C ABI.
//
// The per-read work (pseudoalign, filters, dedup, counting) runs on the GPU behind
// include/nimble_hip.h.  What stays here is what the reference does once per *unique read* with
// strings (filter_and_coerce_sequence_call_orientations, src/align.rs:178-252): it depends only on the
// (class R1, class R2) pair, so it is evaluated once per distinct pair of the device histogram.
//
// The coercion is written over interned strings: every string comparison the reference makes on this
// path is an equality test or the natural_lexical sort, so names, parsed feature names and group values
// are interned once per library (equal strings <-> equal ids) and ranked once with natural_lexical_cmp.
// All of the reference's string quirks are evaluated on the real strings when the tables are built:
// `ends_with("rev")` / trim_end_matches parsing (align.rs:276-285), strip_suffix("§rev")
// (align.rs:149,164), first-match `unmap` (align.rs:851-864), the discarded `unique()` (align.rs:794).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <atomic>
#include <thread>
#include <unordered_set>

#include "../../include/nimble_hip.h"
#include "nimble_host.hpp"

namespace nimble {
namespace align {

namespace {

bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}
std::string trim_end_matches(std::string s, const std::string &pat) {
  while (!pat.empty() && ends_with(s, pat)) s.resize(s.size() - pat.size());
  return s;
}

void check_rc(int rc, const char *what) {
  if (rc != 0) throw Panic(std::string(what) + ": " + nimble_last_error());
}

}  // namespace

// ------------------------------------------------------------------------------------------------
struct Coercer::Impl {
  AlignFilterConfig cfg;
  bool no_rollup = false;  // headers[group_on] == "nt_sequence" (align.rs:810)
  size_t n_rows = 0;
  // string pool
  std::vector<std::string> strs;
  std::unordered_map<std::string, uint32_t> ids;
  std::vector<uint32_t> rank;  // natural_lexical rank of each pooled string
  // per row
  std::vector<uint32_t> row_name;   // sid of the row's sequence_name
  std::vector<uint32_t> row_group;  // sid of the roll-up value (group value, or name when it is empty)
  // per pooled string that is a row name
  std::vector<uint32_t> strip_sid;   // sid of strip_suffix("§rev") (itself when there is no suffix)
  std::vector<uint8_t> has_suffix;
  std::vector<uint32_t> parse_sid;   // parse_calls: feature sid
  std::vector<uint8_t> parse_rev;    // parse_calls: rev flag
  std::vector<int64_t> first_row;    // unmap: first row whose name is this string, -1 = panic

  uint32_t intern(const std::string &s) {
    auto it = ids.find(s);
    if (it != ids.end()) return it->second;
    uint32_t id = (uint32_t)strs.size();
    strs.push_back(s);
    ids.emplace(s, id);
    return id;
  }

  typedef std::pair<uint32_t, bool> Call;  // (feature sid, rev)

  void sort_by_rank(std::vector<uint32_t> &v) const {
    std::sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return rank[a] < rank[b]; });
  }

  // process_equivalence_class_to_feature_list(cls, .., ignore_group_rollup = true): names, sorted
  std::vector<uint32_t> names_of(const std::vector<uint32_t> &cls) const {
    std::vector<uint32_t> out;
    out.reserve(cls.size());
    for (uint32_t r : cls) {
      if (r >= n_rows) throw Panic("index out of bounds: equivalence class row beyond the reference");
      out.push_back(row_name[r]);
    }
    sort_by_rank(out);
    return out;
  }

  // align.rs:144-171
  std::vector<uint32_t> filter_read_calls_with_orientation(const std::vector<uint32_t> &cls) const {
    std::unordered_set<uint32_t> seen, to_remove;
    for (uint32_t f : cls) {
      uint32_t base = strip_sid[f];
      if (seen.count(base)) to_remove.insert(base);
      else seen.insert(base);
    }
    std::vector<uint32_t> out;
    for (uint32_t f : cls) {
      uint32_t key = has_suffix[f] ? strip_sid[f] : f;
      if (!to_remove.count(key)) out.push_back(f);
    }
    return out;
  }

  std::vector<Call> parse_calls(const std::vector<uint32_t> &calls) const {
    std::vector<Call> out;
    out.reserve(calls.size());
    for (uint32_t f : calls) out.emplace_back(parse_sid[f], parse_rev[f] != 0);
    return out;
  }

  // align.rs:287-309
  static void filter_unstranded(const std::vector<Call> &seq, const std::vector<Call> &mate, std::vector<Call> &so,
                                std::vector<Call> &mo) {
    auto has = [](const std::vector<Call> &v, const Call &c) { return std::find(v.begin(), v.end(), c) != v.end(); };
    for (const auto &c : seq)
      if (!has(mate, c)) so.push_back(c);
    for (const auto &c : mate)
      if (!has(seq, c)) mo.push_back(c);
  }

  // align.rs:311-342 (five prime keeps forward calls of the first mate) / :344-375 (three prime keeps reverse)
  static void filter_prime(const std::vector<Call> &seq, const std::vector<Call> &mate, bool keep_rev,
                           std::vector<uint32_t> &so, std::vector<uint32_t> &mo) {
    std::vector<Call> su, mu;
    filter_unstranded(seq, mate, su, mu);
    std::vector<Call> kept_seq;
    std::vector<Call> m = mu;
    for (const auto &call : su) {
      if (call.second != keep_rev) {
        for (size_t p = 0; p < m.size(); ++p)
          if (m[p].first == call.first) { m.erase(m.begin() + (long)p); break; }
      } else {
        kept_seq.push_back(call);
      }
    }
    for (const auto &mc : m) {
      bool constrained = keep_rev ? mc.second : !mc.second;
      bool keep = true;
      if (constrained) {
        keep = false;
        for (const auto &s : kept_seq)
          if (s.first == mc.first) { keep = true; break; }
      }
      if (keep) mo.push_back(mc.first);
    }
    for (const auto &s : kept_seq) so.push_back(s.first);
  }

  // process_equivalence_class_to_feature_list(rows, .., ignore_group_rollup = false), align.rs:802-849
  std::vector<uint32_t> rollup(const std::vector<uint32_t> &rows) const {
    std::vector<uint32_t> out;
    if (no_rollup) {
      for (uint32_t r : rows) out.push_back(row_name[r]);
    } else {
      for (uint32_t r : rows) {
        uint32_t g = row_group[r];
        if (std::find(out.begin(), out.end(), g) == out.end()) out.push_back(g);
      }
    }
    if (cfg.discard_multi_hits > 0 && out.size() > cfg.discard_multi_hits) return {};
    sort_by_rank(out);
    return out;
  }
};

Coercer::Coercer(const reference_library::Reference &ref, const AlignFilterConfig &config) : impl_(new Impl()) {
  Impl &I = *impl_;
  I.cfg = config;
  const std::string sep = reference_library::SPECIAL_REVCOMP_FEATURE_NAME_SEPARATOR;
  const std::string rev_suffix = sep + "rev";
  const auto &names = ref.columns.at(ref.sequence_name_idx);
  I.n_rows = names.size();
  I.no_rollup = ref.headers.at(ref.group_on) == "nt_sequence";
  const auto &groups = ref.columns.at(ref.group_on);
  for (size_t r = 0; r < names.size(); ++r) {
    I.row_name.push_back(I.intern(names[r]));
    const std::string &g = r < groups.size() ? groups[r] : names[r];
    I.row_group.push_back(I.intern(g.empty() ? names[r] : g));
  }
  // derived strings of every row name (parse / strip); may add new strings to the pool
  const size_t n_named = I.strs.size();
  std::vector<std::string> stripped(n_named), parsed(n_named);
  std::vector<uint8_t> suf(n_named, 0), rev(n_named, 0);
  for (size_t s = 0; s < n_named; ++s) {
    const std::string str = I.strs[s];
    suf[s] = ends_with(str, rev_suffix);
    stripped[s] = suf[s] ? str.substr(0, str.size() - rev_suffix.size()) : str;
    if (ends_with(str, "rev")) {
      parsed[s] = trim_end_matches(trim_end_matches(str, "rev"), sep);
      rev[s] = 1;
    } else {
      parsed[s] = str;
    }
  }
  I.strip_sid.resize(n_named);
  I.parse_sid.resize(n_named);
  for (size_t s = 0; s < n_named; ++s) {
    I.strip_sid[s] = I.intern(stripped[s]);
    I.parse_sid[s] = I.intern(parsed[s]);
  }
  const size_t n_all = I.strs.size();
  I.strip_sid.resize(n_all);
  I.parse_sid.resize(n_all);
  I.has_suffix.assign(n_all, 0);
  I.parse_rev.assign(n_all, 0);
  for (size_t s = 0; s < n_named; ++s) { I.has_suffix[s] = suf[s]; I.parse_rev[s] = rev[s]; }
  for (size_t s = n_named; s < n_all; ++s) { I.strip_sid[s] = (uint32_t)s; I.parse_sid[s] = (uint32_t)s; }
  // unmap: first row carrying the name
  I.first_row.assign(n_all, -1);
  for (size_t r = names.size(); r-- > 0;) I.first_row[I.row_name[r]] = (int64_t)r;
  // natural-lexical rank of every pooled string
  std::vector<uint32_t> order(n_all);
  for (size_t s = 0; s < n_all; ++s) order[s] = (uint32_t)s;
  std::sort(order.begin(), order.end(),
            [&](uint32_t a, uint32_t b) { return utils::natural_lexical_cmp(I.strs[a], I.strs[b]) < 0; });
  I.rank.assign(n_all, 0);
  for (size_t k = 0; k < n_all; ++k) I.rank[order[k]] = (uint32_t)k;
}

std::vector<std::pair<std::string, bool>> Coercer::parse_calls(const std::vector<std::string> &calls) const {
  const Impl &I = *impl_;
  const std::string sep = reference_library::SPECIAL_REVCOMP_FEATURE_NAME_SEPARATOR;
  std::vector<std::pair<std::string, bool>> out;
  for (const auto &c : calls) {
    auto it = I.ids.find(c);
    if (it != I.ids.end() && it->second < I.parse_rev.size() && (I.parse_rev[it->second] || I.parse_sid[it->second] == it->second)) {
      out.emplace_back(I.strs[I.parse_sid[it->second]], I.parse_rev[it->second] != 0);  // (the table coerce() reads)
    } else if (ends_with(c, "rev")) {
      out.emplace_back(trim_end_matches(trim_end_matches(c, "rev"), sep), true);
    } else {
      out.emplace_back(c, false);
    }
  }
  return out;
}

std::vector<uint32_t> Coercer::unmap(const std::vector<std::string> &feature_list) const {
  const Impl &I = *impl_;
  std::vector<uint32_t> out;
  for (const auto &f : feature_list) {
    auto it = I.ids.find(f);
    if (it == I.ids.end() || I.first_row[it->second] < 0) throw Panic("Feature not found in reference columns");
    out.push_back((uint32_t)I.first_row[it->second]);
  }
  return out;
}

std::vector<std::string> Coercer::feature_list(const std::vector<uint32_t> &cls, bool ignore_group_rollup) const {
  const Impl &I = *impl_;
  for (uint32_t r : cls)
    if (r >= I.n_rows) throw Panic("index out of bounds: equivalence class row beyond the reference");
  std::vector<uint32_t> sids = ignore_group_rollup ? I.names_of(cls) : I.rollup(cls);
  std::vector<std::string> out;
  for (uint32_t sid : sids) out.push_back(I.strs[sid]);
  return out;
}

std::vector<std::string> Coercer::coerce(bool has1, const std::vector<uint32_t> &c1, bool has2,
                                         const std::vector<uint32_t> &c2, FilterReason &triage) const {
  const Impl &I = *impl_;
  triage = FilterReason::None;
  std::vector<uint32_t> sf, mf;
  if (has1) sf = I.names_of(c1);
  if (has2) mf = I.names_of(c2);
  sf = I.filter_read_calls_with_orientation(sf);
  mf = I.filter_read_calls_with_orientation(mf);
  std::vector<Impl::Call> ps = I.parse_calls(sf), pm = I.parse_calls(mf);
  std::vector<uint32_t> s2, m2;
  switch (I.cfg.strand_filter) {
    case LibraryChemistry::None:
      for (auto &c : ps) s2.push_back(c.first);
      for (auto &c : pm) m2.push_back(c.first);
      break;
    case LibraryChemistry::Unstranded: {
      std::vector<Impl::Call> a, b;
      Impl::filter_unstranded(ps, pm, a, b);
      for (auto &c : a) s2.push_back(c.first);
      for (auto &c : b) m2.push_back(c.first);
      break;
    }
    case LibraryChemistry::FivePrime: Impl::filter_prime(ps, pm, false, s2, m2); break;
    case LibraryChemistry::ThreePrime: Impl::filter_prime(ps, pm, true, s2, m2); break;
  }
  // align.rs:219-227
  std::vector<uint32_t> final_callset;
  auto all_calls = [&]() {  // get_all_calls: append; the reference discards the result of unique()
    std::vector<uint32_t> v = s2;
    v.insert(v.end(), m2.begin(), m2.end());
    return v;
  };
  if (I.cfg.intersect_level == IntersectLevel::NoIntersect) {
    final_callset = all_calls();
  } else {
    // array_tool Intersect: unique elements of self (first occurrences) that occur in other
    std::vector<uint32_t> cls;
    for (uint32_t x : s2) {
      if (std::find(cls.begin(), cls.end(), x) != cls.end()) continue;
      if (std::find(m2.begin(), m2.end(), x) != m2.end()) cls.push_back(x);
    }
    if (!cls.empty()) final_callset = cls;
    else if (I.cfg.intersect_level == IntersectLevel::IntersectWithFallback) final_callset = all_calls();
    else triage = FilterReason::ForceIntersectFailure;
  }
  std::vector<uint32_t> rows;
  for (uint32_t sid : final_callset) {
    int64_t r = I.first_row[sid];
    if (r < 0) throw Panic("Feature not found in reference columns");
    rows.push_back((uint32_t)r);
  }
  std::vector<uint32_t> feature_callset = I.rollup(rows);
  if (feature_callset.size() > I.cfg.max_hits_to_report) { triage = FilterReason::MaxHitsExceeded; return {}; }
  if (feature_callset.empty()) { triage = FilterReason::TriageEmptyEquivalenceClass; return {}; }
  std::vector<std::string> out;
  for (uint32_t sid : feature_callset) out.push_back(I.strs[sid]);
  return out;
}

// ------------------------------------------------------------------------------------------------
std::unique_ptr<PseudoAligner> PseudoAligner::build_index(const std::vector<std::string> &sequences,
                                                          const std::vector<std::string> &names, int device) {
  (void)names;  // names are only carried through by the reference's index; they stay on the host
  std::vector<uint8_t> buf;
  std::vector<uint64_t> off(1, 0);
  for (const auto &s : sequences) {
    buf.insert(buf.end(), s.begin(), s.end());
    off.push_back(buf.size());
  }
  std::unique_ptr<PseudoAligner> pa(new PseudoAligner());
  check_rc(nimble_index_build(buf.data(), off.data(), (uint32_t)sequences.size(), device, &pa->index_),
           "Error -- could not create pseudoaligner index of the reference library");
  check_rc(nimble_ctx_create(pa->index_, nullptr, &pa->ctx_), "nimble_ctx_create");
  return pa;
}

// open-addressing map u64 -> i32 (class pair -> callset id); ~0 never occurs as a key (a pair has at least one class)
struct FlatMap64 {
  std::vector<uint64_t> keys;
  std::vector<int32_t> vals;
  size_t used = 0;
  FlatMap64() : keys(1u << 12, ~0ULL), vals(1u << 12, 0) {}
  static uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    return x;
  }
  const int32_t *find(uint64_t k) const {
    const size_t mask = keys.size() - 1;
    for (size_t p = (size_t)mix(k) & mask;; p = (p + 1) & mask) {
      if (keys[p] == k) return &vals[p];
      if (keys[p] == ~0ULL) return nullptr;
    }
  }
  void insert(uint64_t k, int32_t v) {
    if ((used + 1) * 2 > keys.size()) {
      FlatMap64 big;
      big.keys.assign(keys.size() * 2, ~0ULL);
      big.vals.assign(keys.size() * 2, 0);
      for (size_t i = 0; i < keys.size(); ++i)
        if (keys[i] != ~0ULL) big.insert(keys[i], vals[i]);
      *this = std::move(big);
    }
    const size_t mask = keys.size() - 1;
    size_t p = (size_t)mix(k) & mask;
    while (keys[p] != ~0ULL && keys[p] != k) p = (p + 1) & mask;
    if (keys[p] == ~0ULL) ++used;
    keys[p] = k;
    vals[p] = v;
  }
};

struct PseudoAligner::CoercionMemo {
  // exact copies of everything Coercer reads
  size_t group_on = 0, name_idx = 0;
  std::string group_header;
  std::vector<std::string> names, groups;
  LibraryChemistry strand = LibraryChemistry::None;
  IntersectLevel level = IntersectLevel::NoIntersect;
  size_t discard_multi_hits = 0, max_hits = 0;
  std::unique_ptr<Coercer> coercer;
  // (class R1 << 32 | class R2) -> index into `callsets` (or -1 when the pair is triaged away)
  FlatMap64 pairs;
  std::deque<std::string> joined;  // callsets joined by '\t', by callset id (stable addresses)
  std::unordered_map<uint64_t, FilterReason> pair_triage;  // why a pair with id -1 was dropped (align.rs:228-239,776)
  std::vector<std::vector<std::string>> callsets;
  std::map<std::vector<std::string>, int32_t> callset_ids;
  std::vector<int32_t> sorted;  // callset ids in Vec<String> order; rebuilt when callsets grew
  std::vector<int64_t> counts;  // scratch, one slot per callset

  bool matches(const reference_library::Reference &r, const AlignFilterConfig &c) const {
    return group_on == r.group_on && name_idx == r.sequence_name_idx && strand == c.strand_filter &&
           level == c.intersect_level && discard_multi_hits == c.discard_multi_hits &&
           max_hits == c.max_hits_to_report && group_header == r.headers.at(r.group_on) &&
           names == r.columns.at(r.sequence_name_idx) && groups == r.columns.at(r.group_on);
  }
};

PseudoAligner::CoercionMemo &PseudoAligner::memo_for(const reference_library::Reference &r,
                                                     const AlignFilterConfig &c) {
  if (!memo_ || !memo_->matches(r, c)) {
    memo_.reset(new CoercionMemo());
    memo_->group_on = r.group_on;
    memo_->name_idx = r.sequence_name_idx;
    memo_->group_header = r.headers.at(r.group_on);
    memo_->names = r.columns.at(r.sequence_name_idx);
    memo_->groups = r.columns.at(r.group_on);
    memo_->strand = c.strand_filter;
    memo_->level = c.intersect_level;
    memo_->discard_multi_hits = c.discard_multi_hits;
    memo_->max_hits = c.max_hits_to_report;
    memo_->coercer.reset(new Coercer(r, c));
  }
  return *memo_;
}

nimble_ctx *PseudoAligner::ctx(int slot) {
  if (slot == 0) return ctx_;
  if (slot < 1 || slot > 3) throw Panic("PseudoAligner::ctx: slot must be 0..3");
  nimble_ctx *&c = extra_[slot - 1];
  // slot 1 launches on slot 0's stream (calls in flight stay in order); the utility context gets a stream of its
  // own when NIMBLE_UTIL_STREAM is set (experiment: pack / route beside the align kernel of the call in flight)
  static const bool util_own = getenv("NIMBLE_UTIL_STREAM") != nullptr;
  if (!c)
    check_rc(nimble_ctx_create(index_, (slot == 2 && util_own) ? nullptr : nimble_ctx_stream(ctx_), &c),
             "nimble_ctx_create");
  return c;
}

PseudoAligner::~PseudoAligner() {
  for (nimble_ctx *c : extra_)
    if (c) nimble_ctx_free(c);
  if (ctx_) nimble_ctx_free(ctx_);
  if (index_) nimble_index_free(index_);
}

const std::vector<uint32_t> &PseudoAligner::eq_class(uint32_t id) {
  auto it = class_cache_.find(id);
  if (it != class_cache_.end()) return it->second;
  uint32_t len = 0;
  check_rc(nimble_class_get(index_, id, nullptr, 0, &len), "nimble_class_get");
  std::vector<uint32_t> v(len);
  if (len) check_rc(nimble_class_get(index_, id, v.data(), len, &len), "nimble_class_get");
  return class_cache_.emplace(id, std::move(v)).first->second;
}

void PseudoAligner::prefetch_classes(std::vector<uint32_t> ids) {
  std::sort(ids.begin(), ids.end());
  ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
  ids.erase(std::remove_if(ids.begin(), ids.end(), [&](uint32_t c) { return class_cache_.count(c) != 0; }), ids.end());
  if (ids.size() < 16) return;  // (a handful: eq_class fetches them one by one)
  // the table rows of [lo, hi] in two copies, then the slice of the id pool they point into in one
  const uint32_t lo = ids.front(), hi = ids.back();
  std::vector<uint32_t> len(hi - lo + 1), off(hi - lo + 1);
  check_rc(nimble_class_table_read(index_, lo, hi - lo + 1, len.data(), off.data()), "nimble_class_table_read");
  uint64_t plo = ~0ULL, phi = 0;
  for (uint32_t c : ids) {
    const uint32_t l = len[c - lo], o = off[c - lo];
    if (l == 0) continue;
    plo = std::min<uint64_t>(plo, o);
    phi = std::max<uint64_t>(phi, (uint64_t)o + l);
  }
  if (plo >= phi) return;
  if (phi - plo > (1ULL << 28)) return;  // (classes scattered over more than 1 GiB of the pool: leave it to eq_class)
  std::vector<uint32_t> pool(phi - plo);
  check_rc(nimble_class_pool_read(index_, (uint32_t)plo, (uint32_t)(phi - plo), pool.data()), "nimble_class_pool_read");
  for (uint32_t c : ids) {
    const uint32_t l = len[c - lo], o = off[c - lo];
    if (l == 0) continue;
    class_cache_.emplace(c, std::vector<uint32_t>(pool.begin() + (long)(o - plo), pool.begin() + (long)(o - plo + l)));
  }
}

const std::vector<std::string> &RowRefs::features(size_t i) const { return memo->callsets.at((size_t)ids.at(i)); }

const std::string &RowRefs::joined(size_t i) const {
  PseudoAligner::CoercionMemo &m = *memo;
  while (m.joined.size() < m.callsets.size()) {
    const std::vector<std::string> &f = m.callsets[m.joined.size()];
    std::string j;
    for (size_t k = 0; k < f.size(); ++k) {
      if (k) j.push_back('\t');
      j += f[k];
    }
    m.joined.push_back(std::move(j));
  }
  return m.joined[(size_t)ids.at(i)];
}

void CallOutput::materialize() {
  if (rows.size() == refs.size()) return;
  rows.clear();
  rows.reserve(refs.size());
  for (size_t i = 0; i < refs.size(); ++i) rows.emplace_back(refs.features(i), refs.counts[i]);
}

static size_t kept_score(FilterReason r, uint32_t cls, int32_t score) {
  const bool passed = r == FilterReason::SuccessfulMatch || (r == FilterReason::NotMatchingPair && cls != NIMBLE_CLASS_NONE);
  return passed ? (size_t)score : 0;
}

static nimble_align_params make_params(const AlignFilterConfig &config) {
  nimble_align_params p;
  memset(&p, 0, sizeof p);
  p.score_percent = config.score_percent;
  p.score_threshold = config.score_threshold;
  p.num_mismatches = (uint32_t)config.num_mismatches;
  p.discard_nonzero_mismatch = config.discard_nonzero_mismatch;
  p.discard_multiple_matches = config.discard_multiple_matches;
  p.require_valid_pair = config.require_valid_pair;
  p.min_read_length = (uint32_t)MIN_READ_LENGTH;
  return p;
}

void device_params(const AlignFilterConfig &config, nimble_align_params *out) { *out = make_params(config); }

static CallOutput finish_calls(uint64_t n_reads, PseudoAligner &index, const reference_library::Reference &reference,
                               const AlignFilterConfig &config, bool want_per_read,
                               std::chrono::steady_clock::time_point t0, int slot = 0);

void begin_calls_packed(const nimble_packed &in, uint64_t n, uint32_t max_len, PseudoAligner &index,
                        const AlignFilterConfig &config, int slot) {
  nimble_align_params p = make_params(config);
  check_rc(nimble_call_packed(index.ctx(slot), &p, &in, n, max_len), "nimble_call_packed");
}

void begin_calls_records(const uint64_t *records, uint64_t n, uint32_t max_len, bool paired, PseudoAligner &index,
                         const AlignFilterConfig &config, int slot) {
  nimble_align_params p = make_params(config);
  check_rc(nimble_call_records(index.ctx(slot), &p, records, n, max_len, paired ? 1 : 0), "nimble_call_records");
}

void pack_reads(const ReadBatch &seqs, const ReadBatch *mates, PseudoAligner &index, const AlignFilterConfig &config,
                const nimble_packed &out, int slot) {
  if (mates && mates->n != seqs.n)
    throw Panic("Error -- read and reverse read files do not have matching lengths: ");
  nimble_align_params p = make_params(config);
  uint32_t max_len = std::max(seqs.max_len, mates ? mates->max_len : 0u);
  if (max_len == 0) max_len = std::max(seqs.fixed_len, mates ? mates->fixed_len : 0u);
  check_rc(nimble_pack(index.ctx(slot), &p, seqs.bases, seqs.offsets, mates ? mates->bases : nullptr,
                       mates ? mates->offsets : nullptr, seqs.n, seqs.fixed_len, max_len,
                       seqs.device ? NIMBLE_MEM_DEVICE : NIMBLE_MEM_HOST, &out),
           "nimble_pack");
}

CallOutput get_calls_packed(const nimble_packed &in, uint64_t n, uint32_t max_len, PseudoAligner &index,
                            const reference_library::Reference &reference, const AlignFilterConfig &config) {
  auto t0 = std::chrono::steady_clock::now();
  nimble_align_params p = make_params(config);
  check_rc(nimble_call_packed(index.ctx(), &p, &in, n, max_len), "nimble_call_packed");
  return finish_calls(n, index, reference, config, false, t0);
}

void begin_calls(const ReadBatch &seqs, const ReadBatch *mates, PseudoAligner &index, const AlignFilterConfig &config,
                 int slot) {
  if (mates && mates->n != seqs.n)
    throw Panic("Error -- read and reverse read files do not have matching lengths: ");
  nimble_align_params p = make_params(config);
  uint32_t max_len = std::max(seqs.max_len, mates ? mates->max_len : 0u);
  if (max_len == 0) max_len = std::max(seqs.fixed_len, mates ? mates->fixed_len : 0u);
  if (seqs.stride && (!mates || mates->stride)) {  // reads that are already packed (what score::call gets: DnaStrings)
    check_rc(nimble_call_words(index.ctx(slot), &p, seqs.words, seqs.lens, seqs.stride, mates ? mates->words : nullptr,
                               mates ? mates->lens : nullptr, mates ? mates->stride : 0u, seqs.n, max_len,
                               seqs.device ? NIMBLE_MEM_DEVICE : NIMBLE_MEM_HOST),
             "nimble_call_words");
    return;
  }
  // the device call is asynchronous: the coercion tables are (re)built while the GPU works
  check_rc(nimble_call(index.ctx(slot), &p, seqs.bases, seqs.offsets, mates ? mates->bases : nullptr,
                       mates ? mates->offsets : nullptr, seqs.n, seqs.fixed_len, max_len,
                       seqs.device ? NIMBLE_MEM_DEVICE : NIMBLE_MEM_HOST),
           "nimble_call");
}

// A table of (callset, count) as the rows of `index` (light rows: ids into its coercion memo, which learns the callsets it
// has not produced itself -- the counts of several ranks summed, handed out through rank 0's index).
CallOutput rows_from_table(PseudoAligner &index, const reference_library::Reference &reference,
                           const AlignFilterConfig &config, const std::map<std::vector<std::string>, int64_t> &table) {
  PseudoAligner::CoercionMemo &memo = index.memo_for(reference, config);
  CallOutput out;
  out.refs.memo = index.memo_ptr();
  for (const auto &kv : table) {  // std::map order == Vec<String> order
    if (kv.second == 0) continue;
    auto ins = memo.callset_ids.emplace(kv.first, (int32_t)memo.callsets.size());
    if (ins.second) {
      memo.callsets.push_back(kv.first);
      memo.counts.push_back(0);
      memo.sorted.clear();
    }
    out.refs.ids.push_back(ins.first->second);
    out.refs.counts.push_back((int32_t)kv.second);
  }
  if (!index.light_rows()) out.materialize();
  return out;
}

CallOutput end_calls(uint64_t n_reads, PseudoAligner &index, const reference_library::Reference &reference,
                     const AlignFilterConfig &config, int slot, bool want_per_read) {
  return finish_calls(n_reads, index, reference, config, want_per_read, std::chrono::steady_clock::now(), slot);
}

CallStream::CallStream(PseudoAligner &index, const AlignFilterConfig &config, bool paired, uint32_t max_len,
                       uint64_t capacity_hint)
    : index_(index), config_(config), paired_(paired), max_len_(max_len), t0_(std::chrono::steady_clock::now()) {
  nimble_align_params p = make_params(config);
  check_rc(nimble_stream_begin(index.ctx(), &p, paired ? 1 : 0, max_len, capacity_hint), "nimble_stream_begin");
  open_ = true;
}

CallStream::~CallStream() {
  if (open_) (void)nimble_stream_end(index_.ctx());  // abandoned (an exception is in flight): just close it
}

void CallStream::append(const ReadBatch &seqs, const ReadBatch *mates) {
  if ((mates != nullptr) != paired_ || (mates && mates->n != seqs.n))
    throw Panic("Error -- read and reverse read files do not have matching lengths: ");
  if (seqs.stride && (!mates || mates->stride)) {  // packed by the host's parser threads: a quarter of the bytes on the link
    check_rc(nimble_stream_append_packed(index_.ctx(), seqs.words, seqs.lens, seqs.stride, mates ? mates->words : nullptr,
                                         mates ? mates->lens : nullptr, mates ? mates->stride : 0u, seqs.n),
             "nimble_stream_append_packed");
    n_ += seqs.n;
    return;
  }
  check_rc(nimble_stream_append(index_.ctx(), seqs.bases, seqs.offsets, mates ? mates->bases : nullptr,
                                mates ? mates->offsets : nullptr, seqs.n, seqs.fixed_len,
                                seqs.device ? NIMBLE_MEM_DEVICE
                                            : (seqs.pinned && (!mates || mates->pinned) ? NIMBLE_MEM_HOST_PINNED : NIMBLE_MEM_HOST)),
           "nimble_stream_append");
  n_ += seqs.n;
}

CallOutput CallStream::finish(const reference_library::Reference &reference, bool want_per_read) {
  open_ = false;
  check_rc(nimble_stream_end(index_.ctx()), "nimble_stream_end");
  return finish_calls(n_, index_, reference, config_, want_per_read, t0_);
}

CallOutput get_calls(const ReadBatch &seqs, const ReadBatch *mates, PseudoAligner &index,
                     const reference_library::Reference &reference, const AlignFilterConfig &config,
                     bool want_per_read) {
  auto t0 = std::chrono::steady_clock::now();
  begin_calls(seqs, mates, index, config, 0);
  return finish_calls(seqs.n, index, reference, config, want_per_read, t0);
}

static CallOutput finish_calls(uint64_t n_reads, PseudoAligner &index, const reference_library::Reference &reference,
                               const AlignFilterConfig &config, bool want_per_read,
                               std::chrono::steady_clock::time_point t0, int slot) {
  nimble_ctx *const ctx = index.ctx(slot);
  static const bool timing = getenv("NIMBLE_HOST_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  PseudoAligner::CoercionMemo &memo = index.memo_for(reference, config);
  auto t1 = now();
  uint64_t ne = 0;
  check_rc(nimble_histogram(ctx, nullptr, nullptr, nullptr, 0, &ne), "nimble_histogram");
  std::vector<uint32_t> c1(ne), c2(ne);
  std::vector<uint64_t> cnt(ne);
  if (ne) check_rc(nimble_histogram(ctx, c1.data(), c2.data(), cnt.data(), ne, &ne), "nimble_histogram");

  auto t2 = now();
  // the `results` HashMap of align.rs:434, as dense counts over the memoised callsets
  static const std::vector<uint32_t> empty;
  memo.counts.assign(memo.callsets.size(), 0);
  // Class pairs this memo has not met: their classes come over in bulk and they are coerced by all host threads (the
  // coercion is a pure function of the pair); a library of allele families leaves 10^5 new pairs in its first histograms
  // -- one by one, with a device round trip per class, that was 10 s for 4 M reads.
  {
    std::vector<uint64_t> fresh;
    for (uint64_t e = 0; e < ne; ++e) {
      const uint64_t key = ((uint64_t)c1[e] << 32) | c2[e];
      if (!memo.pairs.find(key)) fresh.push_back(e);
    }
    if (fresh.size() >= 64) {
      std::vector<uint32_t> need;
      for (uint64_t e : fresh) {
        if (c1[e] != NIMBLE_CLASS_NONE) need.push_back(c1[e]);
        if (c2[e] != NIMBLE_CLASS_NONE) need.push_back(c2[e]);
      }
      index.prefetch_classes(std::move(need));
      static const std::vector<uint32_t> none;
      std::vector<const std::vector<uint32_t> *> k1(fresh.size()), k2(fresh.size());
      for (size_t i = 0; i < fresh.size(); ++i) {  // (eq_class fills the cache: not from the worker threads)
        const uint64_t e = fresh[i];
        k1[i] = c1[e] != NIMBLE_CLASS_NONE ? &index.eq_class(c1[e]) : &none;
        k2[i] = c2[e] != NIMBLE_CLASS_NONE ? &index.eq_class(c2[e]) : &none;
      }
      std::vector<std::vector<std::string>> sets(fresh.size());
      std::vector<FilterReason> tri(fresh.size());
      const unsigned n_threads = std::max(1u, std::min(parse::usable_cpus(), 32u));
      std::atomic<size_t> next{0};
      // (work-sharing over an atomic cursor: every task runs the same loop, so tasks whose thread the system refused simply
      // run on this thread -- csrc/threads.h)
      threads::run_indexed(n_threads, [&](unsigned) {
        for (size_t i = next.fetch_add(256); i < fresh.size(); i = next.fetch_add(256))
          for (size_t j = i; j < std::min(fresh.size(), i + 256); ++j) {
            const uint64_t e = fresh[j];
            sets[j] = memo.coercer->coerce(c1[e] != NIMBLE_CLASS_NONE, *k1[j], c2[e] != NIMBLE_CLASS_NONE, *k2[j], tri[j]);
          }
      });
      for (size_t i = 0; i < fresh.size(); ++i) {  // into the memo in histogram order, as the serial loop would
        const uint64_t e = fresh[i];
        const uint64_t key = ((uint64_t)c1[e] << 32) | c2[e];
        if (memo.pairs.find(key)) continue;
        int32_t id = -1;
        if (!sets[i].empty()) {
          auto ins = memo.callset_ids.emplace(sets[i], (int32_t)memo.callsets.size());
          if (ins.second) {
            memo.callsets.push_back(std::move(sets[i]));
            memo.counts.push_back(0);
            memo.sorted.clear();
          }
          id = ins.first->second;
        }
        if (id < 0) memo.pair_triage[key] = tri[i];
        memo.pairs.insert(key, id);
      }
    }
  }
  for (uint64_t e = 0; e < ne; ++e) {
    const uint64_t key = ((uint64_t)c1[e] << 32) | c2[e];
    const int32_t *hit = memo.pairs.find(key);
    int32_t id;
    if (hit) {
      id = *hit;
    } else {
      bool has1 = c1[e] != NIMBLE_CLASS_NONE, has2 = c2[e] != NIMBLE_CLASS_NONE;
      FilterReason triage;
      std::vector<std::string> callset = memo.coercer->coerce(has1, has1 ? index.eq_class(c1[e]) : empty, has2,
                                                              has2 ? index.eq_class(c2[e]) : empty, triage);
      id = -1;
      if (!callset.empty()) {
        auto ins = memo.callset_ids.emplace(callset, (int32_t)memo.callsets.size());
        if (ins.second) {
          memo.callsets.push_back(std::move(callset));
          memo.counts.push_back(0);
          memo.sorted.clear();
        }
        id = ins.first->second;
      }
      if (id < 0) memo.pair_triage[key] = triage;
      memo.pairs.insert(key, id);
    }
    if (id >= 0) memo.counts[(size_t)id] += (int64_t)cnt[e];
  }
  auto t3 = now();
  if (memo.sorted.size() != memo.callsets.size()) {
    memo.sorted.clear();
    for (auto &kv : memo.callset_ids) memo.sorted.push_back(kv.second);  // std::map order == Vec<String> order
  }
  CallOutput out;
  out.refs.memo = index.memo_ptr();
  for (int32_t id : memo.sorted)
    if (memo.counts[(size_t)id]) {
      out.refs.ids.push_back(id);
      out.refs.counts.push_back((int32_t)memo.counts[(size_t)id]);
    }
  if (!index.light_rows()) out.materialize();
  if (timing) {
    auto t4 = now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[nimble host] enqueue + coercion tables %.3f ms, wait + histogram %.3f ms, coerce %llu pairs "
                    "%.3f ms, rows %.3f ms\n",
            ms(t0, t1), ms(t1, t2), (unsigned long long)ne, ms(t2, t3), ms(t3, t4));
  }
  if (want_per_read) {
    out.per_read.resize(n_reads);
    std::vector<int32_t> r[2], s[2];
    std::vector<uint32_t> cl[2];
    for (int m = 0; m < 2; ++m) {
      r[m].resize(n_reads);
      s[m].resize(n_reads);
      cl[m].resize(n_reads);
      check_rc(nimble_read_records(ctx, m, r[m].data(), s[m].data(), nullptr, cl[m].data(), nullptr, n_reads),
               "nimble_read_records");
    }
    for (uint64_t i = 0; i < n_reads; ++i) {
      FilterRecord &fr = out.per_read[i];
      fr.r1 = (FilterReason)r[0][i];
      fr.r2 = (FilterReason)r[1][i];
      // the score slot of filter_reasons holds the score only for alignments that passed pseudoalign (align.rs:561-572);
      // a pair dropped by require_valid_pair keeps the scores of the mates that had (align.rs:586: a passing mate still
      // carries its class here)
      fr.score1 = kept_score(fr.r1, cl[0][i], s[0][i]);
      fr.score2 = kept_score(fr.r2, cl[1][i], s[1][i]);
      fr.triage = FilterReason::None;
    }
  }
  return out;
}

// The BAM pipeline's calls (process/bam.rs:183-226,229-290): one score::call per UMI group, here all groups
// of a batch in ONE device call (segment ids scope the dedup and the counts), reads trimmed for quality on the
// device, SKIP_ALIGN dummies skipped.  Rows come back per segment, sorted by (segment, callset).
UmiOutput get_calls_umis(const ReadBatch &seqs, const ReadBatch *mates, const UmiExtras &ex, PseudoAligner &index,
                         const reference_library::Reference &reference, const AlignFilterConfig &config,
                         bool want_per_read) {
  if (mates && mates->n != seqs.n)
    throw Panic("Error -- read and reverse read files do not have matching lengths: ");
  nimble_align_params p = make_params(config);
  uint32_t max_len = std::max(seqs.max_len, mates ? mates->max_len : 0u);
  if (max_len == 0) max_len = std::max(seqs.fixed_len, mates ? mates->fixed_len : 0u);
  nimble_call_extra x;
  memset(&x, 0, sizeof x);
  x.segment = ex.segment;
  x.n_segments = ex.n_segments;
  x.qual[0] = ex.qual[0];
  x.qual[1] = ex.qual[1];
  x.skip[0] = ex.skip[0];
  x.skip[1] = ex.skip[1];
  x.trim_strictness = config.trim_strictness;
  x.trim_target_length = config.trim_target_length;
  nimble_ctx *ctx = index.ctx();
  check_rc(nimble_call_ex(ctx, &p, seqs.bases, seqs.offsets, mates ? mates->bases : nullptr,
                          mates ? mates->offsets : nullptr, seqs.n, seqs.fixed_len, max_len,
                          seqs.device ? NIMBLE_MEM_DEVICE : NIMBLE_MEM_HOST, &x),
           "nimble_call_ex");
  PseudoAligner::CoercionMemo &memo = index.memo_for(reference, config);
  uint64_t ne = 0;
  check_rc(nimble_histogram_seg(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &ne), "nimble_histogram_seg");
  std::vector<uint32_t> sg(ne), c1(ne), c2(ne), rep(ne);
  std::vector<uint64_t> cnt(ne);
  if (ne)
    check_rc(nimble_histogram_seg(ctx, sg.data(), c1.data(), c2.data(), cnt.data(), rep.data(), ne, &ne),
             "nimble_histogram_seg");
  static const std::vector<uint32_t> empty;
  auto callset_of = [&](uint32_t a, uint32_t b) -> int32_t {
    const uint64_t key = ((uint64_t)a << 32) | b;
    if (const int32_t *hit = memo.pairs.find(key)) return *hit;
    const bool has1 = a != NIMBLE_CLASS_NONE, has2 = b != NIMBLE_CLASS_NONE;
    FilterReason triage;
    std::vector<std::string> callset =
        memo.coercer->coerce(has1, has1 ? index.eq_class(a) : empty, has2, has2 ? index.eq_class(b) : empty, triage);
    int32_t id = -1;
    if (!callset.empty()) {
      auto ins = memo.callset_ids.emplace(callset, (int32_t)memo.callsets.size());
      if (ins.second) {
        memo.callsets.push_back(std::move(callset));
        memo.counts.push_back(0);
        memo.sorted.clear();
      }
      id = ins.first->second;
    } else {
      memo.pair_triage[key] = triage;
    }
    memo.pairs.insert(key, id);
    return id;
  };
  UmiOutput out;
  out.callsets = &memo.callsets;
  // entries arrive sorted by (segment, c1, c2): fold each segment's class pairs into callsets (a handful per segment: a
  // small array, no map, and no copy of the feature names per row -- 131 k segments a call made this the call's longest part)
  std::vector<std::pair<int32_t, std::pair<int64_t, uint32_t>>> acc;  // callset id -> (count, representative)
  for (uint64_t e = 0; e < ne;) {
    const uint32_t seg = sg[e];
    acc.clear();
    for (; e < ne && sg[e] == seg; ++e) {
      const int32_t id = callset_of(c1[e], c2[e]);
      if (id < 0) continue;
      size_t k = 0;
      while (k < acc.size() && acc[k].first != id) ++k;
      if (k == acc.size()) acc.push_back({id, {0, 0u}});
      acc[k].second.first += (int64_t)cnt[e];
      acc[k].second.second = std::max(acc[k].second.second, rep[e]);
    }
    const size_t first = out.rows.size();
    for (auto &kv : acc) {
      UmiRow r;
      r.segment = seg;
      r.callset = kv.first;
      r.count = (int32_t)kv.second.first;
      r.representative = kv.second.second;
      out.rows.push_back(r);
    }
    if (out.rows.size() - first > 1)
      std::sort(out.rows.begin() + (long)first, out.rows.end(), [&](const UmiRow &a, const UmiRow &b) {
        return memo.callsets[(size_t)a.callset] < memo.callsets[(size_t)b.callset];
      });
  }
  if (want_per_read) {
    const uint64_t n = seqs.n;
    out.per_read.resize(n);
    std::vector<int32_t> r[2], sc[2];
    std::vector<uint32_t> cl[2];
    for (int m = 0; m < 2; ++m) {
      r[m].resize(n);
      sc[m].resize(n);
      cl[m].resize(n);
      check_rc(nimble_read_records(ctx, m, r[m].data(), sc[m].data(), nullptr, cl[m].data(), nullptr, n),
               "nimble_read_records");
    }
    for (uint64_t i = 0; i < n; ++i) {
      FilterRecord &fr = out.per_read[i];
      fr.r1 = (FilterReason)r[0][i];
      fr.r2 = (FilterReason)r[1][i];
      fr.score1 = kept_score(fr.r1, cl[0][i], sc[0][i]);
      fr.score2 = kept_score(fr.r2, cl[1][i], sc[1][i]);
      fr.triage = FilterReason::None;
      // post_triaged_keys (align.rs:228-239): reads that reached the coercion and were dropped there
      const uint32_t a = cl[0][i], b = cl[1][i];
      if (fr.r1 != FilterReason::NotMatchingPair && (a != NIMBLE_CLASS_NONE || b != NIMBLE_CLASS_NONE)) {
        auto t = memo.pair_triage.find(((uint64_t)a << 32) | b);
        if (t != memo.pair_triage.end()) fr.triage = t->second;
      }
    }
  }
  return out;
}

// ------------------------------------------------------------------------------------------------
// BAM-only trimming helper (align.rs:873-942)
namespace {
double norm_ratio(const std::vector<double> &a, size_t margin) {
  double mx = std::fabs(a[0]);
  for (size_t i = 1; i < a.size(); ++i) mx = std::max(mx, std::fabs(a[i]));
  return (double)INT64_MAX / (mx * (double)margin);
}
int64_t as_i64(double v) {  // Rust `as i64`
  if (std::isnan(v)) return 0;
  if (v >= 9223372036854775807.0) return INT64_MAX;
  if (v <= -9223372036854775808.0) return INT64_MIN;
  return (int64_t)v;
}
}  // namespace

size_t maxinfo(const std::string &quality, size_t target_length, double strictness) {
  const size_t LONGEST_READ = 1000, MAXQUAL = 60;
  std::vector<double> ls(LONGEST_READ), qp(MAXQUAL + 1);
  for (size_t i = 0; i < LONGEST_READ; ++i) {
    double pow1 = std::exp((double)target_length - (double)i - 1.0);
    ls[i] = std::log(1.0 / (1.0 + pow1)) + std::log((double)(i + 1)) * (1.0 - strictness);
  }
  for (size_t i = 0; i <= MAXQUAL; ++i)
    qp[i] = std::log(1.0 - std::pow(10.0, -((0.5 + (double)i) / 10.0))) * strictness;
  double ratio = std::max(norm_ratio(ls, LONGEST_READ * 2), norm_ratio(qp, LONGEST_READ * 2));
  std::vector<int64_t> lsi(LONGEST_READ), qpi(MAXQUAL + 1);
  for (size_t i = 0; i < LONGEST_READ; ++i) lsi[i] = as_i64(ls[i] * ratio);
  for (size_t i = 0; i <= MAXQUAL; ++i) qpi[i] = as_i64(qp[i] * ratio);
  uint64_t accum = 0;
  double max_score = -1.7976931348623157e308;
  size_t pos = 0;
  for (size_t i = 0; i < quality.size(); ++i) {
    size_t q = (unsigned char)quality[i];
    if (q > MAXQUAL) q = MAXQUAL;
    accum += (uint64_t)qpi[q];
    int64_t score = (int64_t)((i < LONGEST_READ ? (uint64_t)lsi[i] : 0ULL) + accum);
    if ((double)score >= max_score) { max_score = (double)score; pos = i + 1; }
  }
  if (pos < 1 || max_score == 0.0) return 0;
  return pos < quality.size() ? pos : quality.size();
}

}  // namespace align

namespace score {
align::CallOutput call(const align::ReadBatch &sequences, const align::ReadBatch *mate_sequences,
                       align::PseudoAligner &reference_index, const reference_library::Reference &reference,
                       const align::AlignFilterConfig &aligner_config, bool want_per_read) {
  align::CallOutput out = align::get_calls(sequences, mate_sequences, reference_index, reference, aligner_config,
                                           want_per_read);
  out.materialize();
  utils::sort_score_vector(out.rows);  // (utils.rs:54-59, called at score.rs:42)
  return out;
}

align::CallOutput call_packed(const nimble_packed &in, uint64_t n, uint32_t max_len,
                              align::PseudoAligner &reference_index, const reference_library::Reference &reference,
                              const align::AlignFilterConfig &aligner_config) {
  align::CallOutput out = align::get_calls_packed(in, n, max_len, reference_index, reference, aligner_config);
  out.materialize();
  utils::sort_score_vector(out.rows);
  return out;
}
}  // namespace score

}  // namespace nimble
