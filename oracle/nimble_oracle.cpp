// nimble_oracle.cpp -- CPU oracle for the nimble-aligner hot path.
//
// TEST INFRASTRUCTURE ONLY (see nimble_oracle.h).  A literal, single-file CPU restatement of
//   src/score.rs:14-46, src/align.rs:18-989, src/filter/align.rs:4-45, src/utils.rs:7-24,54-119,
//   src/reference_library.rs:8-226 (row expansion)
// of BimberLab/nimble-aligner, plus the behaviour of the two external crates the reference calls
// at src/align.rs:21,965 and src/bin/main.rs:121-128 (hextraza/rust-pseudoaligner `build_index`,
// `map_read_with_mismatch`; 10XGenomics/rust-debruijn `DnaString`, `Kmer30`) -- neither crate is
// vendored or version-pinned (Cargo.toml:22-23), so those parts restate the published algorithm
// and are anchored on the reference's own known-answer tests (tests/test_oracle_golden.py).
//
// Style: deliberately literal (strings, hash maps keyed by read strings, base-by-base compares),
// mirroring the reference's data flow so that each function can be read next to the cited lines.
#include "nimble_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

thread_local std::string g_err;

constexpr size_t K = 30;                         // Kmer30 (src/align.rs:21)
constexpr uint64_t KMASK = (1ULL << (2 * K)) - 1;
constexpr size_t MIN_READ_LENGTH = 40;           // src/align.rs:18
constexpr double MIN_ENTROPY_SCORE = 1.75;       // src/align.rs:19
const std::string SEP = "\xC2\xA7";              // "§", reference_library.rs:8
const std::string REV_SUFFIX = SEP + "rev";

// ------------------------------------------------------------------------------------------
// a12: debruijn::dna_string::DnaString -- encoding semantics only.
// from_acgt_bytes: A/a->0 C/c->1 G/g->2 T/t->3, anything else -> 0 ('A').  to_string renders ACGT.
// ------------------------------------------------------------------------------------------
inline uint8_t base_to_bits(uint8_t c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 0;
  }
}
const char BITS_TO_BASE[4] = {'A', 'C', 'G', 'T'};

struct Dna {
  std::vector<uint8_t> b;  // one 2-bit code per element (packing is not observable)
  static Dna from_acgt_bytes(const uint8_t *p, size_t n) {
    Dna d;
    d.b.resize(n);
    for (size_t i = 0; i < n; ++i) d.b[i] = base_to_bits(p[i]);
    return d;
  }
  size_t len() const { return b.size(); }
  uint8_t get(size_t i) const { return b[i]; }
  std::string to_string() const {
    std::string s(b.size(), 'A');
    for (size_t i = 0; i < b.size(); ++i) s[i] = BITS_TO_BASE[b[i]];
    return s;
  }
  // Kmer30: first base in the most significant position, so integer order == lexicographic order
  uint64_t get_kmer(size_t pos) const {
    uint64_t v = 0;
    for (size_t i = 0; i < K; ++i) v = (v << 2) | b[pos + i];
    return v;
  }
};

// ------------------------------------------------------------------------------------------
// reference_library::Reference (reference_library.rs:10-17)
// ------------------------------------------------------------------------------------------
struct Ref {
  size_t group_on = 0;
  std::vector<std::string> headers;
  std::vector<std::vector<std::string>> columns;
  size_t sequence_name_idx = 0;
  size_t sequence_idx = 0;
};

// utils::revcomp (utils.rs:61-94)
bool is_valid_base_pair(char bp) {
  switch (bp) {
    case 'A': case 'a': case 'C': case 'c': case 'G': case 'g':
    case 'T': case 't': case 'U': case 'u': case 'N': case 'n':
      return true;
    default:
      return false;
  }
}
char revcomp_base_pair(char bp) {
  switch (bp) {
    case 'a': return 't';
    case 'c': return 'g';
    case 't': return 'a';
    case 'g': return 'c';
    case 'u': return 'a';
    case 'A': return 'T';
    case 'C': return 'G';
    case 'T': return 'A';
    case 'G': return 'C';
    case 'U': return 'A';
    default: return 'N';
  }
}
std::string revcomp(const std::string &s) {
  std::string out;
  out.reserve(s.size());
  for (size_t i = s.size(); i-- > 0;) {
    char bp = s[i];
    if (!is_valid_base_pair(bp))
      throw std::runtime_error(std::string("Input sequence base is not DNA: ") + bp);
    out.push_back(revcomp_base_pair(bp));
  }
  return out;
}

int get_column_index(const std::vector<std::string> &headers, const std::string &h) {
  for (size_t i = 0; i < headers.size(); ++i)
    if (headers[i] == h) return (int)i;
  return -1;
}

// ------------------------------------------------------------------------------------------
// External crate: coloured compacted de Bruijn graph index (a5') -- see SURVEY 8(c).
//   * all 30-mers of all sequences, stranded (no canonicalisation)
//   * colour of a k-mer = sorted set of sequence ids containing it, interned into eq_classes
//   * k-mer extensions = union of the neighbouring bases seen in the sequences
//   * maximal unitigs: extend x -> y iff x has exactly one right extension, y exactly one left
//     extension, colour(x) == colour(y), and y is not already on a path (cuts cycles)
//   * exact dictionary k-mer -> (node, offset)  (== MPHF + verification of the hit)
// ------------------------------------------------------------------------------------------
struct Node {
  std::vector<uint8_t> seq;
  uint8_t lext = 0, rext = 0;  // bit b set = extension by base b exists
  uint32_t colour = 0;
  uint32_t l_edge[4] = {0, 0, 0, 0};
  uint32_t r_edge[4] = {0, 0, 0, 0};
};

struct KmerTable {
  std::vector<uint64_t> keys;
  std::vector<uint64_t> vals;
  uint64_t mask = 0;
  static uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
  }
  void init(size_t n) {
    size_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    keys.assign(cap, ~0ULL);
    vals.assign(cap, 0);
    mask = cap - 1;
  }
  void put(uint64_t k, uint64_t v) {
    uint64_t h = mix(k) & mask;
    while (keys[h] != ~0ULL && keys[h] != k) h = (h + 1) & mask;
    keys[h] = k;
    vals[h] = v;
  }
  bool get(uint64_t k, uint64_t &v) const {
    if (keys.empty()) return false;
    uint64_t h = mix(k) & mask;
    while (keys[h] != ~0ULL) {
      if (keys[h] == k) { v = vals[h]; return true; }
      h = (h + 1) & mask;
    }
    return false;
  }
};

struct Index {
  std::vector<Node> nodes;
  std::vector<std::vector<uint32_t>> eq_classes;
  KmerTable table;  // kmer -> (node << 32 | offset)
  uint64_t n_kmers = 0;
};

inline int popcount4(uint8_t m) { return __builtin_popcount(m & 0xF); }
inline int single_bit(uint8_t m) { return __builtin_ctz(m); }

Index *build_index(const std::vector<Dna> &seqs) {
  struct Occ { uint64_t kmer; uint32_t seq; uint8_t lext, rext; };
  std::vector<Occ> occ;
  size_t total = 0;
  for (auto &s : seqs) if (s.len() >= K) total += s.len() - K + 1;
  occ.reserve(total);
  for (size_t si = 0; si < seqs.size(); ++si) {
    const Dna &s = seqs[si];
    if (s.len() < K) continue;
    uint64_t km = s.get_kmer(0);
    for (size_t p = 0; p + K <= s.len(); ++p) {
      if (p > 0) km = ((km << 2) | s.get(p + K - 1)) & KMASK;
      Occ o;
      o.kmer = km;
      o.seq = (uint32_t)si;
      o.lext = p > 0 ? (uint8_t)(1u << s.get(p - 1)) : 0;
      o.rext = p + K < s.len() ? (uint8_t)(1u << s.get(p + K)) : 0;
      occ.push_back(o);
    }
  }
  std::sort(occ.begin(), occ.end(), [](const Occ &a, const Occ &b) {
    return a.kmer != b.kmer ? a.kmer < b.kmer : a.seq < b.seq;
  });

  Index *ix = new Index();
  // distinct k-mers with their summaries (CountFilterEqClass::summarize: exts union, sorted dedup ids)
  std::vector<uint64_t> kmers;
  std::vector<uint8_t> lext, rext;
  std::vector<uint32_t> colour;
  std::map<std::vector<uint32_t>, uint32_t> intern;
  for (size_t i = 0; i < occ.size();) {
    size_t j = i;
    uint8_t l = 0, r = 0;
    std::vector<uint32_t> ids;
    while (j < occ.size() && occ[j].kmer == occ[i].kmer) {
      l |= occ[j].lext;
      r |= occ[j].rext;
      if (ids.empty() || ids.back() != occ[j].seq) ids.push_back(occ[j].seq);
      ++j;
    }
    auto it = intern.find(ids);
    uint32_t cid;
    if (it == intern.end()) {
      cid = (uint32_t)ix->eq_classes.size();
      intern.emplace(ids, cid);
      ix->eq_classes.push_back(ids);
    } else {
      cid = it->second;
    }
    kmers.push_back(occ[i].kmer);
    lext.push_back(l);
    rext.push_back(r);
    colour.push_back(cid);
    i = j;
  }
  occ.clear();
  occ.shrink_to_fit();
  const size_t n = kmers.size();
  ix->n_kmers = n;
  auto find = [&](uint64_t km) -> size_t {
    size_t p = std::lower_bound(kmers.begin(), kmers.end(), km) - kmers.begin();
    if (p >= n || kmers[p] != km) throw std::runtime_error("missing link");
    return p;
  };

  // compaction into maximal unitigs; iteration in ascending k-mer order makes cycle cuts canonical
  std::vector<uint8_t> used(n, 0);
  std::vector<uint32_t> kmer_node(n, 0), kmer_off(n, 0);
  for (size_t i = 0; i < n; ++i) {
    if (used[i]) continue;
    used[i] = 1;
    std::vector<size_t> right, left;
    size_t cur = i;
    while (popcount4(rext[cur]) == 1) {
      uint64_t nk = ((kmers[cur] << 2) | (uint64_t)single_bit(rext[cur])) & KMASK;
      size_t j = find(nk);
      if (used[j] || popcount4(lext[j]) != 1 || colour[j] != colour[cur]) break;
      used[j] = 1;
      right.push_back(j);
      cur = j;
    }
    cur = i;
    while (popcount4(lext[cur]) == 1) {
      uint64_t pk = ((uint64_t)single_bit(lext[cur]) << (2 * (K - 1))) | (kmers[cur] >> 2);
      size_t j = find(pk);
      if (used[j] || popcount4(rext[j]) != 1 || colour[j] != colour[cur]) break;
      used[j] = 1;
      left.push_back(j);
      cur = j;
    }
    std::vector<size_t> path(left.rbegin(), left.rend());
    path.push_back(i);
    path.insert(path.end(), right.begin(), right.end());
    Node nd;
    uint32_t nid = (uint32_t)ix->nodes.size();
    uint64_t first = kmers[path[0]];
    for (size_t b = 0; b < K; ++b) nd.seq.push_back((uint8_t)((first >> (2 * (K - 1 - b))) & 3));
    for (size_t t = 1; t < path.size(); ++t) nd.seq.push_back((uint8_t)(kmers[path[t]] & 3));
    nd.lext = lext[path.front()];
    nd.rext = rext[path.back()];
    nd.colour = colour[i];
    for (size_t t = 0; t < path.size(); ++t) {
      kmer_node[path[t]] = nid;
      kmer_off[path[t]] = (uint32_t)t;
    }
    ix->nodes.push_back(std::move(nd));
  }
  // edges (debruijn::graph::find_edges: the neighbour k-mer of the terminal k-mer)
  for (auto &nd : ix->nodes) {
    uint64_t firstk = 0, lastk = 0;
    for (size_t b = 0; b < K; ++b) firstk = (firstk << 2) | nd.seq[b];
    for (size_t b = nd.seq.size() - K; b < nd.seq.size(); ++b) lastk = (lastk << 2) | nd.seq[b];
    for (int b = 0; b < 4; ++b) {
      if (nd.rext & (1 << b)) {
        size_t j = find(((lastk << 2) | (uint64_t)b) & KMASK);
        if (kmer_off[j] != 0) throw std::runtime_error("right edge does not land on a node start");
        nd.r_edge[b] = kmer_node[j];
      }
      if (nd.lext & (1 << b)) {
        size_t j = find(((uint64_t)b << (2 * (K - 1))) | (firstk >> 2));
        nd.l_edge[b] = kmer_node[j];
      }
    }
  }
  ix->table.init(n);
  for (size_t i = 0; i < n; ++i) ix->table.put(kmers[i], ((uint64_t)kmer_node[i] << 32) | kmer_off[i]);
  return ix;
}

// ------------------------------------------------------------------------------------------
// a5: Pseudoaligner::map_read_to_nodes_with_mismatch / nodes_to_eq_class / map_read_with_mismatch
// ------------------------------------------------------------------------------------------
struct WalkCounters { uint64_t probes = 0, nodes = 0, class_entries = 0; };

bool find_kmer_match(const Index &ix, const Dna &read, size_t &kmer_pos, size_t last_kmer_pos,
                     uint32_t &nid, uint32_t &off, WalkCounters &wc) {
  while (kmer_pos <= last_kmer_pos) {
    uint64_t km = read.get_kmer(kmer_pos);
    uint64_t v;
    wc.probes++;
    if (ix.table.get(km, v)) {  // exact dictionary == MPHF lookup + verification of the k-mer
      nid = (uint32_t)(v >> 32);
      off = (uint32_t)v;
      return true;
    }
    kmer_pos += 3;
  }
  return false;
}

bool map_read_to_nodes_with_mismatch(const Index &ix, const Dna &read, std::vector<uint32_t> &nodes,
                                     size_t allowed_mismatches, size_t &coverage_out,
                                     size_t &mismatch_out, WalkCounters &wc) {
  const size_t read_length = read.len();
  size_t read_coverage = 0, mismatch_count = 0;
  if (read_length < K) return false;
  const size_t left_extend_threshold = (size_t)(0.2 * (double)read_length);
  size_t kmer_pos = 0;
  const size_t kmer_length = K;
  const size_t last_kmer_pos = read_length - kmer_length;

  uint32_t node_id = 0, kmer_offset = 0;
  bool have = find_kmer_match(ix, read, kmer_pos, last_kmer_pos, node_id, kmer_offset, wc);

  // left extension when the first seed was found late in the read
  if (kmer_pos >= left_extend_threshold && have) {
    size_t last_pos = kmer_pos - 1;
    uint32_t prev_node_id = node_id;
    size_t prev_kmer_offset = kmer_offset > 0 ? kmer_offset - 1 : 0;
    for (;;) {
      const Node &node = ix.nodes[prev_node_id];
      size_t skipped_read = last_pos + 1;
      size_t skipped_ref = prev_kmer_offset + 1;
      size_t max_matchable_pos = std::min(skipped_read, skipped_ref);
      bool premature_break = false;
      size_t matched_bases = 0, seen_snp = 0;
      for (size_t idx = 0; idx < max_matchable_pos; ++idx) {
        size_t ref_pos = prev_kmer_offset - idx;
        size_t read_offset = last_pos - idx;
        if (node.seq[ref_pos] != read.get(read_offset)) {
          mismatch_count += 1;
          seen_snp += 1;
          if (seen_snp > allowed_mismatches) { premature_break = true; break; }
        }
        matched_bases += 1;
        read_coverage += 1;
      }
      if (last_pos + 1 - matched_bases == 0 || premature_break) break;
      last_pos -= matched_bases;
      uint8_t next_base = read.get(last_pos);
      if (node.lext & (1 << next_base)) {
        prev_node_id = node.l_edge[next_base];
        prev_kmer_offset = ix.nodes[prev_node_id].seq.size() - kmer_length;
        nodes.push_back(prev_node_id);
      } else {
        break;
      }
    }
  }

  // forward search
  if (kmer_pos <= last_kmer_pos) {
    for (;;) {
      const Node &node = ix.nodes[node_id];
      kmer_pos += kmer_length;
      read_coverage += kmer_length;
      nodes.push_back(node_id);
      size_t remaining_read = read_length - kmer_pos;
      size_t ref_length = node.seq.size();
      size_t ref_offset = kmer_offset + kmer_length;
      size_t informative_ref = ref_length - ref_offset;
      size_t max_matchable_pos = std::min(remaining_read, informative_ref);
      bool premature_break = false;
      size_t matched_bases = 0, seen_snp = 0;
      for (size_t idx = 0; idx < max_matchable_pos; ++idx) {
        size_t ref_pos = ref_offset + idx;
        size_t read_offset = kmer_pos + idx;
        if (node.seq[ref_pos] != read.get(read_offset)) {
          mismatch_count += 1;
          seen_snp += 1;
          if (seen_snp > allowed_mismatches) { premature_break = true; break; }
        }
        matched_bases += 1;
        read_coverage += 1;
      }
      kmer_pos += matched_bases;
      if (kmer_pos >= read_length) break;
      uint8_t next_base = read.get(kmer_pos);
      if (!premature_break && (node.rext & (1 << next_base))) {
        node_id = node.r_edge[next_base];
        kmer_offset = 0;
        kmer_pos -= kmer_length - 1;
        read_coverage -= kmer_length - 1;
      } else {
        if (kmer_pos > last_kmer_pos) break;
        if (!find_kmer_match(ix, read, kmer_pos, last_kmer_pos, node_id, kmer_offset, wc)) break;
      }
    }
  }
  if (nodes.empty()) return false;
  wc.nodes += nodes.size();
  coverage_out = read_coverage;
  mismatch_out = mismatch_count;
  return true;
}

// sorted-list intersection, result stays in `a`
void intersect_sorted(std::vector<uint32_t> &a, const std::vector<uint32_t> &b) {
  size_t i = 0, j = 0, o = 0;
  while (i < a.size() && j < b.size()) {
    if (a[i] < b[j]) ++i;
    else if (a[i] > b[j]) ++j;
    else { a[o++] = a[i]; ++i; ++j; }
  }
  a.resize(o);
}

void nodes_to_eq_class(const Index &ix, std::vector<uint32_t> &nodes, std::vector<uint32_t> &eq_class,
                       WalkCounters &wc) {
  eq_class.clear();
  if (nodes.empty()) return;
  std::stable_sort(nodes.begin(), nodes.end(), [&](uint32_t a, uint32_t b) {
    return ix.eq_classes[ix.nodes[a].colour].size() < ix.eq_classes[ix.nodes[b].colour].size();
  });
  const auto &first = ix.eq_classes[ix.nodes[nodes[0]].colour];
  eq_class.assign(first.begin(), first.end());
  wc.class_entries += first.size();
  for (size_t i = 1; i < nodes.size(); ++i) {
    const auto &c = ix.eq_classes[ix.nodes[nodes[i]].colour];
    wc.class_entries += c.size();
    intersect_sorted(eq_class, c);
  }
}

bool map_read_with_mismatch(const Index &ix, const Dna &read, size_t allowed,
                            std::vector<uint32_t> &eq_class, size_t &coverage, size_t &mismatches,
                            WalkCounters &wc) {
  std::vector<uint32_t> nodes;
  if (!map_read_to_nodes_with_mismatch(ix, read, nodes, allowed, coverage, mismatches, wc)) return false;
  nodes_to_eq_class(ix, nodes, eq_class, wc);
  return true;
}

// ------------------------------------------------------------------------------------------
// a9: utils::shannon_entropy (utils.rs:96-119) -- counts in the order A, T, C, G
// ------------------------------------------------------------------------------------------
double shannon_entropy(const std::string &dna) {
  double total_length = (double)dna.size();
  double f[4] = {0.0, 0.0, 0.0, 0.0};
  for (char c : dna) {
    switch (c) {
      case 'A': f[0] += 1.0; break;
      case 'T': f[1] += 1.0; break;
      case 'C': f[2] += 1.0; break;
      case 'G': f[3] += 1.0; break;
      default: break;
    }
  }
  for (double &x : f) x /= total_length;
  double entropy = 0.0;
  for (double x : f)
    if (x > 0.0) entropy += x * std::log2(x);
  return -entropy;
}

// ------------------------------------------------------------------------------------------
// a6: filter::align::filter_alignment_by_metrics (filter/align.rs:4-45)
// returns ORA_SUCCESSFUL_MATCH when the alignment is kept, else the FilterReason
// ------------------------------------------------------------------------------------------
int filter_alignment_by_metrics(size_t cls_len, size_t score, double normalized, size_t score_threshold,
                                double score_percent, bool discard_multiple_matches,
                                size_t mismatch_threshold, size_t mismatches) {
  if (score >= score_threshold && normalized >= score_percent && cls_len != 0) {
    if (discard_multiple_matches && cls_len > 1) return ORA_DISCARDED_MULTIPLE_MATCH;
    if (mismatches > mismatch_threshold) return ORA_ABOVE_MISMATCH_THRESHOLD;
    return ORA_SUCCESSFUL_MATCH;
  }
  return ORA_SCORE_BELOW_THRESHOLD;
}

// ------------------------------------------------------------------------------------------
// a4: align::pseudoalign (align.rs:945-989)
// ------------------------------------------------------------------------------------------
struct Alignment {  // (AlignmentScore, Filter)
  bool some = false;             // AlignmentScore is Some
  std::vector<uint32_t> cls;     // equivalence class (also kept on the filtered path for records)
  double normalized = 0.0;
  size_t score = 0;
  int reason = ORA_SUCCESSFUL_MATCH;  // Filter reason when !some
  double f_normalized = 0.0;          // Filter tuple values
  size_t f_score = 0;
  size_t walk_score = 0, walk_mm = 0;  // raw a5 outputs for parity records
  bool walk_some = false;
};

Alignment pseudoalign(const Dna &sequence, const Index &ix, const ora_config &cfg, size_t min_read_length,
                      WalkCounters &wc) {
  Alignment a;
  if (sequence.len() < min_read_length) { a.reason = ORA_SHORT_READ; return a; }
  if (shannon_entropy(sequence.to_string()) < MIN_ENTROPY_SCORE) { a.reason = ORA_HIGH_ENTROPY; return a; }
  size_t score = 0, mismatches = 0;
  std::vector<uint32_t> cls;
  if (!map_read_with_mismatch(ix, sequence, (size_t)cfg.num_mismatches, cls, score, mismatches, wc)) {
    a.reason = ORA_NO_MATCH;
    return a;
  }
  a.walk_some = true;
  a.walk_score = score;
  a.walk_mm = mismatches;
  double normalized = (double)score / (double)sequence.len();
  if (cfg.discard_nonzero_mismatch && mismatches != 0) {
    a.reason = ORA_DISCARDED_NONZERO_MISMATCH;
    a.cls = std::move(cls);
    return a;
  }
  int r = filter_alignment_by_metrics(cls.size(), score, normalized, (size_t)cfg.score_threshold,
                                      cfg.score_percent, cfg.discard_multiple_matches != 0,
                                      (size_t)cfg.num_mismatches, mismatches);
  a.cls = std::move(cls);
  if (r == ORA_SUCCESSFUL_MATCH) {
    a.some = true;
    a.normalized = normalized;
    a.score = score;
  } else {
    a.reason = r;
    a.f_normalized = normalized;
    a.f_score = score;
  }
  return a;
}

// ------------------------------------------------------------------------------------------
// a7: filter_pair (align.rs:732-760)
// ------------------------------------------------------------------------------------------
bool filter_pair(std::vector<uint32_t> a, std::vector<uint32_t> b) {
  if (!a.empty() && !b.empty()) {
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    size_t matching = 0;
    for (size_t i = 0; i < a.size() && i < b.size(); ++i)
      if (a[i] == b[i]) ++matching;
    if (matching != a.size() || matching != b.size()) return true;
  } else {
    return true;
  }
  return false;
}

// ------------------------------------------------------------------------------------------
// lexical_sort::natural_lexical_cmp (align.rs:15,846) -- third-party crate, restated from its
// documentation: transliterate to ASCII (any_ascii), ignore case, digit runs by numeric value;
// ties broken by the plain string order.  Only U+00A7 needs transliteration on this path ("SS").
// ------------------------------------------------------------------------------------------
std::string lexical_form(const std::string &s) {
  std::string o;
  for (size_t i = 0; i < s.size(); ++i) {
    unsigned char c = (unsigned char)s[i];
    if (c == 0xC2 && i + 1 < s.size() && (unsigned char)s[i + 1] == 0xA7) { o += "ss"; ++i; continue; }
    if (c >= 'A' && c <= 'Z') c = (unsigned char)(c - 'A' + 'a');
    o.push_back((char)c);
  }
  return o;
}
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
int natural_lexical_cmp(const std::string &s1, const std::string &s2) {
  std::string a = lexical_form(s1), b = lexical_form(s2);
  size_t i = 0, j = 0;
  for (;;) {
    if (i == a.size() && j == b.size()) break;
    if (i == a.size()) return -1;
    if (j == b.size()) return 1;
    if (is_digit(a[i]) && is_digit(b[j])) {
      size_t i2 = i, j2 = j;
      while (i2 < a.size() && is_digit(a[i2])) ++i2;
      while (j2 < b.size() && is_digit(b[j2])) ++j2;
      size_t ia = i, jb = j;
      while (ia + 1 < i2 && a[ia] == '0') ++ia;
      while (jb + 1 < j2 && b[jb] == '0') ++jb;
      size_t la = i2 - ia, lb = j2 - jb;
      if (la != lb) return la < lb ? -1 : 1;
      int c = a.compare(ia, la, b, jb, lb);
      if (c != 0) return c < 0 ? -1 : 1;
      i = i2;
      j = j2;
    } else {
      if (a[i] != b[j]) return (unsigned char)a[i] < (unsigned char)b[j] ? -1 : 1;
      ++i;
      ++j;
    }
  }
  int c = s1.compare(s2);
  return c < 0 ? -1 : (c > 0 ? 1 : 0);
}

// ------------------------------------------------------------------------------------------
// a8 helpers (align.rs:143-376, 763-864)
// ------------------------------------------------------------------------------------------
typedef std::vector<std::string> Strs;
typedef std::pair<std::string, bool> Call;

bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}
std::string trim_end_matches(std::string s, const std::string &pat) {
  while (!pat.empty() && ends_with(s, pat)) s.resize(s.size() - pat.size());
  return s;
}

// align.rs:802-849
Strs process_equivalence_class_to_feature_list(const std::vector<uint32_t> &cls, const Ref &ref,
                                               const ora_config &cfg, bool ignore_group_rollup) {
  Strs results;
  if (ignore_group_rollup || ref.headers[ref.group_on] == "nt_sequence") {
    for (uint32_t idx : cls) results.push_back(ref.columns[ref.sequence_name_idx].at(idx));
  } else {
    for (uint32_t idx : cls) {
      const std::string *group = &ref.columns[ref.group_on].at(idx);
      if (group->empty()) group = &ref.columns[ref.sequence_name_idx].at(idx);
      if (std::find(results.begin(), results.end(), *group) == results.end()) results.push_back(*group);
    }
  }
  if (!ignore_group_rollup && cfg.discard_multi_hits > 0 && results.size() > cfg.discard_multi_hits)
    return Strs();
  std::sort(results.begin(), results.end(),
            [](const std::string &x, const std::string &y) { return natural_lexical_cmp(x, y) < 0; });
  return results;
}

// align.rs:144-171
Strs filter_read_calls_with_orientation(const Strs &cls) {
  std::unordered_set<std::string> seen, to_remove;
  for (const auto &feature : cls) {
    std::string base = ends_with(feature, REV_SUFFIX) ? feature.substr(0, feature.size() - REV_SUFFIX.size())
                                                      : feature;
    if (seen.count(base)) to_remove.insert(base);
    else seen.insert(base);
  }
  Strs out;
  for (const auto &call : cls) {
    if (ends_with(call, REV_SUFFIX)) {
      if (!to_remove.count(call.substr(0, call.size() - REV_SUFFIX.size()))) out.push_back(call);
    } else if (!to_remove.count(call)) {
      out.push_back(call);
    }
  }
  return out;
}

// align.rs:276-285
std::vector<Call> parse_calls(const Strs &calls) {
  std::vector<Call> out;
  for (const auto &call : calls) {
    if (ends_with(call, "rev")) out.emplace_back(trim_end_matches(trim_end_matches(call, "rev"), SEP), true);
    else out.emplace_back(call, false);
  }
  return out;
}

struct CallHash {
  size_t operator()(const Call &c) const { return std::hash<std::string>()(c.first) * 2 + (c.second ? 1 : 0); }
};

// align.rs:287-309
void filter_unstranded(const std::vector<Call> &seq, const std::vector<Call> &mate, std::vector<Call> &seq_out,
                       std::vector<Call> &mate_out) {
  std::unordered_set<Call, CallHash> sset(seq.begin(), seq.end()), mset(mate.begin(), mate.end());
  for (const auto &c : seq)
    if (!mset.count(c)) seq_out.push_back(c);
  for (const auto &c : mate)
    if (!sset.count(c)) mate_out.push_back(c);
}

// align.rs:311-342 (keep_rev == false) and :344-375 (keep_rev == true)
void filter_stranded(const std::vector<Call> &seq, const std::vector<Call> &mate, bool keep_rev, Strs &seq_out,
                     Strs &mate_out) {
  std::vector<Call> su, mu;
  filter_unstranded(seq, mate, su, mu);
  std::vector<Call> seq_filtered;
  std::vector<Call> mate_filtered = mu;
  for (const auto &call : su) {
    bool rev = call.second;
    if (rev != keep_rev) {
      for (size_t p = 0; p < mate_filtered.size(); ++p)
        if (mate_filtered[p].first == call.first) { mate_filtered.erase(mate_filtered.begin() + p); break; }
    } else {
      seq_filtered.push_back(call);
    }
  }
  std::vector<Call> kept;
  for (const auto &m : mate_filtered) {
    bool check = keep_rev ? m.second : !m.second;
    if (check) {
      bool any = false;
      for (const auto &s : seq_filtered)
        if (s.first == m.first) { any = true; break; }
      if (any) kept.push_back(m);
    } else {
      kept.push_back(m);
    }
  }
  for (auto &c : seq_filtered) seq_out.push_back(c.first);
  for (auto &c : kept) mate_out.push_back(c.first);
}

// align.rs:255-274
void filter_orientation_on_library_chemistry(const Strs &seq_calls, const Strs &mate_calls, int chem, Strs &seq_out,
                                             Strs &mate_out) {
  std::vector<Call> ps = parse_calls(seq_calls), pm = parse_calls(mate_calls);
  switch (chem) {
    case ORA_CHEM_NONE:
      for (auto &c : ps) seq_out.push_back(c.first);
      for (auto &c : pm) mate_out.push_back(c.first);
      break;
    case ORA_UNSTRANDED: {
      std::vector<Call> a, b;
      filter_unstranded(ps, pm, a, b);
      for (auto &c : a) seq_out.push_back(c.first);
      for (auto &c : b) mate_out.push_back(c.first);
      break;
    }
    case ORA_FIVE_PRIME: filter_stranded(ps, pm, false, seq_out, mate_out); break;
    case ORA_THREE_PRIME: filter_stranded(ps, pm, true, seq_out, mate_out); break;
    default: throw std::runtime_error("bad strand_filter");
  }
}

// array_tool::vec::Uniq::unique / Intersect::intersect (align.rs:12-13,771,794)
Strs unique_strs(const Strs &v) {
  Strs out;
  for (const auto &x : v)
    if (std::find(out.begin(), out.end(), x) == out.end()) out.push_back(x);
  return out;
}
Strs intersect_strs(const Strs &self, const Strs &other) {
  Strs out;
  for (const auto &x : unique_strs(self))
    if (std::find(other.begin(), other.end(), x) != other.end()) out.push_back(x);
  return out;
}

// align.rs:788-796 -- note: the result of unique() is discarded by the reference
Strs get_all_calls(Strs a, const Strs &b) {
  a.insert(a.end(), b.begin(), b.end());
  (void)unique_strs(a);
  return a;
}

// align.rs:763-785; triage receives ForceIntersectFailure when the read is dropped
Strs get_intersecting_reads(const Strs &a, const Strs &b, bool fallback, int &triage) {
  Strs cls = intersect_strs(a, b);
  if (cls.empty() && fallback) return get_all_calls(a, b);
  if (!cls.empty()) return cls;
  triage = ORA_FORCE_INTERSECT_FAILURE;
  return Strs();
}

// align.rs:851-864
std::vector<uint32_t> unmap(const Strs &features, const Ref &ref) {
  std::vector<uint32_t> out;
  const auto &names = ref.columns[ref.sequence_name_idx];
  for (const auto &f : features) {
    auto it = std::find(names.begin(), names.end(), f);
    if (it == names.end()) throw std::runtime_error("Feature not found in reference columns");
    out.push_back((uint32_t)(it - names.begin()));
  }
  return out;
}

// align.rs:178-252.  Returns the callset (empty when triaged) and the triage reason.
Strs filter_and_coerce(bool has1, const std::vector<uint32_t> &c1, bool has2, const std::vector<uint32_t> &c2,
                       const Ref &ref, const ora_config &cfg, int &triage) {
  triage = ORA_NONE;
  Strs sf, mf;
  if (has1) sf = process_equivalence_class_to_feature_list(c1, ref, cfg, true);
  if (has2) mf = process_equivalence_class_to_feature_list(c2, ref, cfg, true);
  sf = filter_read_calls_with_orientation(sf);
  mf = filter_read_calls_with_orientation(mf);
  Strs sf2, mf2;
  filter_orientation_on_library_chemistry(sf, mf, cfg.strand_filter, sf2, mf2);
  Strs final_callset;
  switch (cfg.intersect_level) {
    case 0: final_callset = get_all_calls(sf2, mf2); break;
    case 1: final_callset = get_intersecting_reads(sf2, mf2, true, triage); break;
    case 2: final_callset = get_intersecting_reads(sf2, mf2, false, triage); break;
    default: throw std::runtime_error("bad intersect_level");
  }
  std::vector<uint32_t> ids = unmap(final_callset, ref);
  Strs feature_callset = process_equivalence_class_to_feature_list(ids, ref, cfg, false);
  if (feature_callset.size() > cfg.max_hits_to_report) { triage = ORA_MAX_HITS_EXCEEDED; return Strs(); }
  if (feature_callset.empty()) { triage = ORA_TRIAGE_EMPTY_EQUIVALENCE_CLASS; return Strs(); }
  return feature_callset;
}

// ------------------------------------------------------------------------------------------
// BAM-only trimming (align.rs:866-942), restated for the unit-test literals
// ------------------------------------------------------------------------------------------
double compute_norm_ratio(const std::vector<double> &arr, size_t margin) {
  double max_val = std::fabs(arr[0]);
  for (size_t i = 1; i < arr.size(); ++i) max_val = std::max(max_val, std::fabs(arr[i]));
  return (double)INT64_MAX / (max_val * (double)margin);
}
int64_t sat_cast_i64(double v) {  // Rust `as i64`: saturating, NaN -> 0
  if (std::isnan(v)) return 0;
  if (v >= 9223372036854775807.0) return INT64_MAX;
  if (v <= -9223372036854775808.0) return INT64_MIN;
  return (int64_t)v;
}
size_t maxinfo(const std::string &quality, size_t target_length, double strictness) {
  const size_t LONGEST_READ = 1000, MAXQUAL = 60;
  std::vector<double> length_scores(LONGEST_READ), qual_probs(MAXQUAL + 1);
  for (size_t i = 0; i < LONGEST_READ; ++i) {
    double pow1 = std::exp((double)target_length - (double)i - 1.0);
    double unique = std::log(1.0 / (1.0 + pow1));
    double coverage = std::log((double)(i + 1)) * (1.0 - strictness);
    length_scores[i] = unique + coverage;
  }
  for (size_t i = 0; i <= MAXQUAL; ++i) {
    double prob_correct = 1.0 - std::pow(10.0, -((0.5 + (double)i) / 10.0));
    qual_probs[i] = std::log(prob_correct) * strictness;
  }
  double norm_ratio = std::max(compute_norm_ratio(length_scores, LONGEST_READ * 2),
                               compute_norm_ratio(qual_probs, LONGEST_READ * 2));
  std::vector<int64_t> ls(LONGEST_READ), qp(MAXQUAL + 1);
  for (size_t i = 0; i < LONGEST_READ; ++i) ls[i] = sat_cast_i64(length_scores[i] * norm_ratio);
  for (size_t i = 0; i <= MAXQUAL; ++i) qp[i] = sat_cast_i64(qual_probs[i] * norm_ratio);
  int64_t accum_quality = 0;
  double max_score = -1.7976931348623157e308;  // f64::MIN
  size_t max_score_position = 0;
  for (size_t i = 0; i < quality.size(); ++i) {
    size_t q = (unsigned char)quality[i];
    if (q > MAXQUAL) q = MAXQUAL;
    accum_quality = (int64_t)((uint64_t)accum_quality + (uint64_t)qp[q]);
    int64_t l = i < LONGEST_READ ? ls[i] : 0;
    int64_t score = (int64_t)((uint64_t)l + (uint64_t)accum_quality);
    if ((double)score >= max_score) {
      max_score = (double)score;
      max_score_position = i + 1;
    }
  }
  if (max_score_position < 1 || max_score == 0.0) return 0;
  if (max_score_position < quality.size()) return max_score_position;
  return quality.size();
}

// ------------------------------------------------------------------------------------------
// a1-a3: score::call -> align::get_calls -> score_sequences, over in-memory reads
// ------------------------------------------------------------------------------------------
struct ScoreEntry {  // value of score_map (align.rs:496-505); metadata is empty on the FASTQ path
  bool has1 = false, has2 = false;
  std::vector<uint32_t> c1, c2;
  uint64_t rep = 0;  // index of the read whose insert is the live one (last writer)
};

uint64_t fnv_class(const std::vector<uint32_t> &c) {
  if (c.empty()) return 0;
  uint64_t h = 0xcbf29ce484222325ULL ^ (uint64_t)c.size();
  for (uint32_t v : c)
    for (int k = 0; k < 4; ++k) { h ^= (v >> (8 * k)) & 0xFF; h *= 0x100000001b3ULL; }
  return h ? h : 1;
}

struct Result {
  std::vector<std::pair<std::string, int32_t>> rows;  // features joined by '\t', count -- sorted
  std::vector<int32_t> reason[2], score[2], mism[2];
  std::vector<uint64_t> class_hash[2];
  std::vector<uint8_t> kept[2];  // the mate's alignment passed pseudoalign (before the pair filter)
  std::vector<uint8_t> counted;
  uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // BAM-mode call (one score::call per UMI): segment id of every row, aligned (trimmed) length per read
  std::vector<uint32_t> row_segment;
  std::vector<int32_t> align_len[2];
};

// what the BAM pipeline adds per read (process/bam.rs:229-290, align.rs:516-552): quality strings for the 3' trim
// and the SKIP_ALIGN flag of unpaired dummies; all optional
struct UmiExtras {
  const uint8_t *q1 = nullptr, *q2 = nullptr;   // qualities, same offsets as the bases
  const uint8_t *skip1 = nullptr, *skip2 = nullptr;
};

struct Partial {
  std::unordered_map<std::string, ScoreEntry> score_map;
  std::unordered_set<std::string> filter_reason_keys;
  WalkCounters wc;
  uint64_t seeded = 0, prefiltered = 0;
  std::map<Strs, int32_t> results;
  std::string err;
};

// one partition of score_sequences (align.rs:475-729); `mine(i)` selects the reads of this partition
size_t maxinfo(const std::string &quality, size_t target_length, double strictness);

// align.rs:866-871 trim_sequence: the first maxinfo(quality) bases
Dna trim_sequence(const Dna &sequence, const uint8_t *qual, size_t qlen, const ora_config &cfg, size_t &trimmed_length) {
  trimmed_length = maxinfo(std::string((const char *)qual, qlen), (size_t)cfg.trim_target_length, cfg.trim_strictness);
  std::string s = sequence.to_string();
  if (trimmed_length > s.size()) throw std::runtime_error("byte index out of range in trim_sequence");
  return Dna::from_acgt_bytes((const uint8_t *)s.data(), trimmed_length);
}

void score_sequences(const Index &ix, const ora_config &cfg, const uint8_t *r1, const uint64_t *o1,
                     const uint8_t *r2, const uint64_t *o2, uint64_t n, const std::function<bool(uint64_t)> &mine,
                     Partial &P, Result *rec, const std::vector<uint64_t> *order = nullptr,
                     const UmiExtras *ex = nullptr) {
  const uint64_t count = order ? order->size() : n;
  for (uint64_t it = 0; it < count; ++it) {
    const uint64_t i = order ? (*order)[it] : it;
    if (!order && !mine(i)) continue;
    Dna read = Dna::from_acgt_bytes(r1 + o1[i], (size_t)(o1[i + 1] - o1[i]));
    WalkCounters before = P.wc;
    // align.rs:519-529: trim for quality when there is metadata, skip the alignment of an unpaired dummy
    size_t t1 = read.len(), t2 = 0;
    Alignment a1;
    if (ex && ex->skip1 && ex->skip1[i]) {
      a1.reason = ORA_SKIPPED_ALIGN_DUE_TO_UNPAIRED_DUMMY;
    } else if (ex && ex->q1) {
      Dna trimmed = trim_sequence(read, ex->q1 + o1[i], (size_t)(o1[i + 1] - o1[i]), cfg, t1);
      a1 = pseudoalign(trimmed, ix, cfg, MIN_READ_LENGTH, P.wc);
    } else {
      a1 = pseudoalign(read, ix, cfg, MIN_READ_LENGTH, P.wc);
    }
    bool have_mate = r2 != nullptr;
    Alignment a2;
    Dna mate;
    if (have_mate) {
      mate = Dna::from_acgt_bytes(r2 + o2[i], (size_t)(o2[i + 1] - o2[i]));
      t2 = mate.len();
      if (ex && ex->skip2 && ex->skip2[i]) {
        a2.reason = ORA_SKIPPED_ALIGN_DUE_TO_UNPAIRED_DUMMY;
      } else if (ex && ex->q2) {
        Dna trimmed = trim_sequence(mate, ex->q2 + o2[i], (size_t)(o2[i + 1] - o2[i]), cfg, t2);
        a2 = pseudoalign(trimmed, ix, cfg, MIN_READ_LENGTH, P.wc);
      } else {
        a2 = pseudoalign(mate, ix, cfg, MIN_READ_LENGTH, P.wc);
      }
    }
    if (rec && !rec->align_len[0].empty()) {
      rec->align_len[0][i] = (int32_t)t1;
      rec->align_len[1][i] = (int32_t)t2;
    }
    if (a1.walk_some || (have_mate && a2.walk_some)) P.seeded++;
    if (a1.reason == ORA_SHORT_READ || a1.reason == ORA_HIGH_ENTROPY) P.prefiltered++;
    (void)before;

    // align.rs:561-572
    std::vector<uint32_t> cls1 = a1.some ? a1.cls : std::vector<uint32_t>();
    std::vector<uint32_t> cls2 = (have_mate && a2.some) ? a2.cls : std::vector<uint32_t>();
    // align.rs:576-579
    std::string read_key = have_mate ? read.to_string() + mate.to_string() : read.to_string();

    if (rec) {
      rec->reason[0][i] = a1.some ? ORA_SUCCESSFUL_MATCH : a1.reason;
      rec->score[0][i] = (int32_t)a1.walk_score;
      rec->mism[0][i] = (int32_t)a1.walk_mm;
      rec->class_hash[0][i] = a1.walk_some ? fnv_class(a1.cls) : 0;
      rec->kept[0][i] = a1.some ? 1 : 0;
      if (have_mate) {
        rec->reason[1][i] = a2.some ? ORA_SUCCESSFUL_MATCH : a2.reason;
        rec->score[1][i] = (int32_t)a2.walk_score;
        rec->mism[1][i] = (int32_t)a2.walk_mm;
        rec->class_hash[1][i] = a2.walk_some ? fnv_class(a2.cls) : 0;
        rec->kept[1][i] = a2.some ? 1 : 0;
      } else {
        rec->reason[1][i] = ORA_SUCCESSFUL_MATCH;  // align.rs:596-599: None -> SuccessfulMatch
      }
    }

    // align.rs:582-601
    if (have_mate && cfg.require_valid_pair && filter_pair(cls1, cls2)) {
      P.filter_reason_keys.insert(read_key);
      if (rec) { rec->reason[0][i] = ORA_NOT_MATCHING_PAIR; rec->reason[1][i] = ORA_NOT_MATCHING_PAIR; }
      continue;
    }
    P.filter_reason_keys.insert(read_key);

    // align.rs:604-685
    if (!cls1.empty() || !cls2.empty()) {
      ScoreEntry e;
      e.has1 = !cls1.empty();
      e.has2 = !cls2.empty();
      e.c1 = std::move(cls1);
      e.c2 = std::move(cls2);
      e.rep = i;
      P.score_map[read_key] = std::move(e);  // insert(): last writer wins
    }
    // (read_matches, align.rs:665-724, is built by the reference and discarded by both pipelines)
  }
}

void coerce_partition(const Ref &ref, const ora_config &cfg, Partial &P) {
  for (auto &kv : P.score_map) {
    int triage;
    Strs callset = filter_and_coerce(kv.second.has1, kv.second.c1, kv.second.has2, kv.second.c2, ref, cfg, triage);
    if (!callset.empty()) P.results[callset] += 1;
  }
}

uint64_t key_hash(const uint8_t *p, uint64_t n, uint64_t h) {
  for (uint64_t i = 0; i < n; ++i) { h ^= BITS_TO_BASE[base_to_bits(p[i])]; h *= 0x100000001b3ULL; }
  return h;
}

Result *call(const Index &ix, const Ref &ref, const ora_config &cfg, const uint8_t *r1, const uint64_t *o1,
             const uint8_t *r2, const uint64_t *o2, uint64_t n, int n_threads, bool keep) {
  Result *res = new Result();
  Result *rec = nullptr;
  if (keep) {
    for (int m = 0; m < 2; ++m) {
      res->reason[m].assign(n, ORA_NONE);
      res->score[m].assign(n, 0);
      res->mism[m].assign(n, 0);
      res->class_hash[m].assign(n, 0);
      res->kept[m].assign(n, 0);
    }
    res->counted.assign(n, 0);
    rec = res;
  }
  if (n_threads < 1) n_threads = 1;
  std::vector<Partial> parts((size_t)n_threads);
  if (n_threads == 1) {
    try {
      score_sequences(ix, cfg, r1, o1, r2, o2, n, [](uint64_t) { return true; }, parts[0], rec);
      coerce_partition(ref, cfg, parts[0]);
    } catch (const std::exception &e) { parts[0].err = e.what(); }
  } else {
    // CPU-S baseline: partition by hash(read_key) so that duplicates meet in one partition
    std::vector<uint8_t> part(n);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
      th.emplace_back([&, t]() {
        uint64_t lo = n * (uint64_t)t / (uint64_t)n_threads, hi = n * (uint64_t)(t + 1) / (uint64_t)n_threads;
        for (uint64_t i = lo; i < hi; ++i) {
          uint64_t h = key_hash(r1 + o1[i], o1[i + 1] - o1[i], 0xcbf29ce484222325ULL);
          if (r2) h = key_hash(r2 + o2[i], o2[i + 1] - o2[i], h);
          part[i] = (uint8_t)((h >> 17) % (uint64_t)n_threads);
        }
      });
    for (auto &t : th) t.join();
    th.clear();
    for (int t = 0; t < n_threads; ++t)
      th.emplace_back([&, t]() {
        try {
          score_sequences(ix, cfg, r1, o1, r2, o2, n, [&part, t](uint64_t i) { return part[i] == t; }, parts[t], rec);
          coerce_partition(ref, cfg, parts[t]);
        } catch (const std::exception &e) { parts[t].err = e.what(); }
      });
    for (auto &t : th) t.join();
  }
  std::map<Strs, int32_t> merged;
  for (auto &P : parts) {
    if (!P.err.empty()) { g_err = P.err; delete res; return nullptr; }
    for (auto &kv : P.results) merged[kv.first] += kv.second;
    res->counters[1] += P.score_map.size();
    res->counters[2] += P.wc.probes;
    res->counters[3] += P.wc.nodes;
    res->counters[4] += P.wc.class_entries;
    res->counters[5] += P.seeded;
    res->counters[6] += P.prefiltered;
    res->counters[7] += P.filter_reason_keys.size();
    if (keep)
      for (auto &kv : P.score_map) res->counted[kv.second.rep] = 1;
  }
  res->counters[0] = n;
  // score.rs:42 / utils.rs:54-59: rows sorted by Vec<String> Ord == std::map order over Strs
  for (auto &kv : merged) {
    std::string joined;
    for (size_t i = 0; i < kv.first.size(); ++i) {
      if (i) joined.push_back('\t');
      joined += kv.first[i];
    }
    res->rows.emplace_back(joined, kv.second);
  }
  return res;
}

// The BAM pipeline's use of score::call (process/bam.rs:183-226,229-290): one call per UMI group.  `segment`
// groups the reads (all reads of one id form one call, in input order); rows come back sorted by (segment, callset).
Result *call_umi(const Index &ix, const Ref &ref, const ora_config &cfg, const uint8_t *r1, const uint64_t *o1,
                 const uint8_t *r2, const uint64_t *o2, const UmiExtras &ex, const uint32_t *segment, uint64_t n,
                 bool keep) {
  Result *res = new Result();
  Result *rec = nullptr;
  if (keep) {
    for (int m = 0; m < 2; ++m) {
      res->reason[m].assign(n, ORA_NONE);
      res->score[m].assign(n, 0);
      res->mism[m].assign(n, 0);
      res->class_hash[m].assign(n, 0);
      res->kept[m].assign(n, 0);
      res->align_len[m].assign(n, 0);
    }
    res->counted.assign(n, 0);
    rec = res;
  }
  std::map<uint32_t, std::vector<uint64_t>> groups;
  for (uint64_t i = 0; i < n; ++i) groups[segment ? segment[i] : 0u].push_back(i);
  try {
    for (auto &g : groups) {
      Partial P;
      score_sequences(ix, cfg, r1, o1, r2, o2, n, [](uint64_t) { return true; }, P, rec, &g.second, &ex);
      coerce_partition(ref, cfg, P);
      res->counters[1] += P.score_map.size();
      res->counters[2] += P.wc.probes;
      res->counters[3] += P.wc.nodes;
      res->counters[4] += P.wc.class_entries;
      res->counters[5] += P.seeded;
      res->counters[6] += P.prefiltered;
      res->counters[7] += P.filter_reason_keys.size();
      if (keep)
        for (auto &kv : P.score_map) res->counted[kv.second.rep] = 1;
      for (auto &kv : P.results) {
        std::string joined;
        for (size_t i = 0; i < kv.first.size(); ++i) {
          if (i) joined.push_back('\t');
          joined += kv.first[i];
        }
        res->rows.emplace_back(joined, kv.second);
        res->row_segment.push_back(g.first);
      }
    }
  } catch (const std::exception &e) {
    g_err = e.what();
    delete res;
    return nullptr;
  }
  res->counters[0] = n;
  return res;
}

std::string join_lines(const Strs &v) {
  std::string o;
  for (size_t i = 0; i < v.size(); ++i) { if (i) o.push_back('\n'); o += v[i]; }
  return o;
}
Strs split_lines(const char *s) {
  Strs v;
  if (!s || !*s) return v;
  std::string cur;
  for (const char *p = s; *p; ++p) {
    if (*p == '\n') { v.push_back(cur); cur.clear(); }
    else cur.push_back(*p);
  }
  v.push_back(cur);
  return v;
}
int copy_out(const std::string &s, char *out, int cap) {
  if ((int)s.size() + 1 > cap) { g_err = "output buffer too small"; return -1; }
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}


// utils::get_reference_sequence_data (utils.rs:7-24): the sequence column as DnaStrings, the name column beside it; a name
// column shorter than the sequence column is the reference's panic
std::pair<std::vector<Dna>, Strs> get_reference_sequence_data(const Ref &ref) {
  const auto &seqs = ref.columns[ref.sequence_idx];
  const auto &names = ref.columns[ref.sequence_name_idx];
  std::vector<Dna> dna;
  Strs out_names;
  size_t next_name = 0;
  for (const auto &s : seqs) {
    dna.push_back(Dna::from_acgt_bytes((const uint8_t *)s.data(), s.size()));
    if (next_name >= names.size())
      throw std::runtime_error("Error -- could not read library name after JSON parse, corrupted internal state.");
    out_names.push_back(names[next_name++]);
  }
  return {dna, out_names};
}

// utils::sort_score_vector (utils.rs:54-59): `sort_by(|a, b| a.0.cmp(&b.0))` -- a STABLE sort on the Vec<String> key
// (element-wise, each element byte-wise); returns the order as indices into the input
std::vector<int32_t> sort_score_vector_order(const std::vector<Strs> &keys) {
  std::vector<int32_t> order(keys.size());
  for (size_t i = 0; i < keys.size(); ++i) order[i] = (int32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return keys[(size_t)a] < keys[(size_t)b]; });
  return order;
}

Strs split_tabs(const std::string &s) {
  Strs v;
  std::string cur;
  for (char c : s) {
    if (c == '\t') { v.push_back(cur); cur.clear(); }
    else cur.push_back(c);
  }
  v.push_back(cur);
  return v;
}
}  // namespace

struct ora_ref { Ref r; };
struct ora_index { Index *ix; };
struct ora_result { Result *r; };

extern "C" {

const char *ora_last_error(void) { return g_err.c_str(); }

ora_ref *ora_ref_create(int n_cols, const char *const *headers, int n_rows, const char *const *cells,
                        const char *group_on) {
  try {
    std::vector<std::string> hdr;
    for (int c = 0; c < n_cols; ++c) hdr.push_back(headers[c]);
    int name_idx = get_column_index(hdr, "sequence_name");
    if (name_idx < 0) throw std::runtime_error("Could not find header sequence_name");
    int group_idx;
    std::string g = group_on ? group_on : "";
    if (g.empty()) group_idx = name_idx;
    else {
      group_idx = get_column_index(hdr, g);
      if (group_idx < 0) throw std::runtime_error("Error -- could not find column for group_on " + g);
    }
    int seq_idx = get_column_index(hdr, "sequence");
    if (seq_idx < 0) throw std::runtime_error("Error -- could not find sequences column");
    // reference_library.rs:128-161: per input row, the row itself then its reverse complement
    std::vector<std::vector<std::string>> cols((size_t)n_cols);
    for (int r = 0; r < n_rows; ++r) {
      std::vector<std::string> row, rev;
      for (int c = 0; c < n_cols; ++c) {
        std::string v = cells[(size_t)c * n_rows + r];
        if (c == seq_idx)
          for (char &ch : v) { if (ch == 'U') ch = 'T'; else if (ch == 'u') ch = 't'; }
        row.push_back(v);
        rev.push_back(v);
      }
      rev[name_idx] = rev[name_idx] + REV_SUFFIX;
      rev[seq_idx] = revcomp(rev[seq_idx]);
      for (int c = 0; c < n_cols; ++c) { cols[c].push_back(row[c]); cols[c].push_back(rev[c]); }
    }
    ora_ref *o = new ora_ref();
    o->r.group_on = (size_t)group_idx;
    o->r.headers = hdr;
    o->r.columns = std::move(cols);
    o->r.sequence_name_idx = (size_t)name_idx;
    o->r.sequence_idx = (size_t)seq_idx;
    return o;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

ora_ref *ora_ref_create_raw(int n_cols, const char *const *headers, int n_rows, const char *const *cells,
                            int group_on, int sequence_name_idx, int sequence_idx) {
  ora_ref *o = new ora_ref();
  for (int c = 0; c < n_cols; ++c) {
    o->r.headers.push_back(headers[c]);
    std::vector<std::string> col;
    for (int r = 0; r < n_rows; ++r) col.push_back(cells[(size_t)c * n_rows + r]);
    o->r.columns.push_back(col);
  }
  o->r.group_on = (size_t)group_on;
  o->r.sequence_name_idx = (size_t)sequence_name_idx;
  o->r.sequence_idx = (size_t)sequence_idx;
  return o;
}

void ora_ref_free(ora_ref *r) { delete r; }
int ora_ref_n_rows(const ora_ref *r) { return r->r.columns.empty() ? 0 : (int)r->r.columns[0].size(); }
int ora_ref_n_cols(const ora_ref *r) { return (int)r->r.columns.size(); }
int ora_ref_group_on(const ora_ref *r) { return (int)r->r.group_on; }
int ora_ref_sequence_name_idx(const ora_ref *r) { return (int)r->r.sequence_name_idx; }
int ora_ref_sequence_idx(const ora_ref *r) { return (int)r->r.sequence_idx; }
const char *ora_ref_header(const ora_ref *r, int c) { return r->r.headers[c].c_str(); }
const char *ora_ref_cell(const ora_ref *r, int c, int row) { return r->r.columns[c][row].c_str(); }
int ora_ref_push_column(ora_ref *r, const char *header, const char *const *values, int n) {
  r->r.headers.push_back(header);
  std::vector<std::string> col;
  for (int i = 0; i < n; ++i) col.push_back(values[i]);
  r->r.columns.push_back(col);
  return (int)r->r.columns.size() - 1;
}
void ora_ref_set_group_on(ora_ref *r, int col) { r->r.group_on = (size_t)col; }

int ora_sanity_check_config(const ora_config *c) {
  if (!(c->score_percent >= 0.0 && c->score_percent <= 1.0)) { g_err = "Error -- score_percent must be between 0 and 1"; return -1; }
  if (c->score_filter < 0) { g_err = "Error -- score_filter must be positive"; return -1; }
  if (!(c->trim_strictness >= 0.0 && c->trim_strictness <= 1.0)) { g_err = "Error -- trim_strictness must be between 0 and 1"; return -1; }
  return 0;
}

ora_index *ora_index_build_from_ref(const ora_ref *r) {
  try {
    std::vector<Dna> seqs = get_reference_sequence_data(r->r).first;  // (utils.rs:7-24, as src/bin/main.rs:118 does)
    ora_index *o = new ora_index();
    o->ix = build_index(seqs);
    return o;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

ora_index *ora_index_build(int n_seqs, const char *const *seqs) {
  try {
    std::vector<Dna> v;
    for (int i = 0; i < n_seqs; ++i) v.push_back(Dna::from_acgt_bytes((const uint8_t *)seqs[i], strlen(seqs[i])));
    ora_index *o = new ora_index();
    o->ix = build_index(v);
    return o;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

void ora_index_free(ora_index *i) { if (i) { delete i->ix; delete i; } }

void ora_index_stats(const ora_index *i, uint64_t *s) {
  s[0] = i->ix->n_kmers;
  s[1] = i->ix->nodes.size();
  s[2] = i->ix->eq_classes.size();
  uint64_t bases = 0, ce = 0;
  for (auto &n : i->ix->nodes) bases += n.seq.size();
  for (auto &c : i->ix->eq_classes) ce += c.size();
  s[3] = bases;
  s[4] = ce;
}

int ora_index_node(const ora_index *i, uint32_t node, char *seq, int cap, uint32_t *colour, uint32_t *lext,
                   uint32_t *rext) {
  if (node >= i->ix->nodes.size()) return -1;
  const Node &n = i->ix->nodes[node];
  int len = (int)n.seq.size();
  for (int k = 0; k < len && k < cap - 1; ++k) seq[k] = BITS_TO_BASE[n.seq[k]];
  if (cap > 0) seq[std::min(len, cap - 1)] = 0;
  if (colour) *colour = n.colour;
  if (lext) *lext = n.lext;
  if (rext) *rext = n.rext;
  return len;
}

int ora_index_class(const ora_index *i, uint32_t colour, uint32_t *ids, int cap) {
  if (colour >= i->ix->eq_classes.size()) return -1;
  const auto &c = i->ix->eq_classes[colour];
  for (size_t k = 0; k < c.size() && (int)k < cap; ++k) ids[k] = c[k];
  return (int)c.size();
}

int ora_map_read(const ora_index *i, const char *read, int len, int allowed, uint32_t *cls, int cls_cap,
                 int *cls_len, int *score, int *mismatches) {
  Dna d = Dna::from_acgt_bytes((const uint8_t *)read, (size_t)len);
  std::vector<uint32_t> c;
  size_t cov = 0, mm = 0;
  WalkCounters wc;
  if (!map_read_with_mismatch(*i->ix, d, (size_t)allowed, c, cov, mm, wc)) return 0;
  for (size_t k = 0; k < c.size() && (int)k < cls_cap; ++k) cls[k] = c[k];
  *cls_len = (int)c.size();
  *score = (int)cov;
  *mismatches = (int)mm;
  return 1;
}

int ora_pseudoalign(const ora_index *i, const ora_config *cfg, const char *read, int len, int min_read_length,
                    uint32_t *cls, int cls_cap, int *cls_len, int *reason, double *normalized, int *score) {
  Dna d = Dna::from_acgt_bytes((const uint8_t *)read, (size_t)len);
  WalkCounters wc;
  Alignment a = pseudoalign(d, *i->ix, *cfg, (size_t)min_read_length, wc);
  *cls_len = 0;
  if (a.some) {
    for (size_t k = 0; k < a.cls.size() && (int)k < cls_cap; ++k) cls[k] = a.cls[k];
    *cls_len = (int)a.cls.size();
    *reason = ORA_SUCCESSFUL_MATCH;
    *normalized = a.normalized;
    *score = (int)a.score;
    return 1;
  }
  *reason = a.reason;
  *normalized = a.f_normalized;
  *score = (int)a.f_score;
  return 0;
}

int ora_filter_alignment_by_metrics(int cls_len, uint64_t score, double normalized, uint64_t score_threshold,
                                    double score_percent, int discard_multiple_matches,
                                    uint64_t mismatch_threshold, uint64_t mismatches) {
  return filter_alignment_by_metrics((size_t)cls_len, (size_t)score, normalized, (size_t)score_threshold,
                                     score_percent, discard_multiple_matches != 0, (size_t)mismatch_threshold,
                                     (size_t)mismatches);
}

int ora_filter_pair(const uint32_t *a, int na, const uint32_t *b, int nb) {
  return filter_pair(std::vector<uint32_t>(a, a + na), std::vector<uint32_t>(b, b + nb)) ? 1 : 0;
}

double ora_shannon_entropy(const char *dna) { return shannon_entropy(dna); }
int ora_natural_lexical_cmp(const char *a, const char *b) { return natural_lexical_cmp(a, b); }
uint64_t ora_maxinfo(const char *q, int qlen, uint64_t target_length, double strictness) {
  return maxinfo(std::string(q, (size_t)qlen), (size_t)target_length, strictness);
}
int ora_revcomp(const char *seq, char *out) {
  try {
    std::string r = revcomp(seq);
    memcpy(out, r.c_str(), r.size() + 1);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

int ora_coerce(const ora_ref *r, const ora_config *cfg, int has_r1, const uint32_t *c1, int n1, int has_r2,
               const uint32_t *c2, int n2, char *out, int cap) {
  try {
    int triage;
    Strs cs = filter_and_coerce(has_r1 != 0, std::vector<uint32_t>(c1, c1 + n1), has_r2 != 0,
                                std::vector<uint32_t>(c2, c2 + n2), r->r, *cfg, triage);
    std::string joined;
    for (size_t i = 0; i < cs.size(); ++i) { if (i) joined.push_back('\t'); joined += cs[i]; }
    if (copy_out(joined, out, cap) < 0) return -1;
    return triage;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

int ora_filter_read_calls_with_orientation(const char *in, char *out, int cap) {
  return copy_out(join_lines(filter_read_calls_with_orientation(split_lines(in))), out, cap);
}

int ora_filter_orientation_on_library_chemistry(const char *seq, const char *mate, int chem, char *out_seq,
                                                char *out_mate, int cap) {
  try {
    Strs a, b;
    filter_orientation_on_library_chemistry(split_lines(seq), split_lines(mate), chem, a, b);
    if (copy_out(join_lines(a), out_seq, cap) < 0) return -1;
    if (copy_out(join_lines(b), out_mate, cap) < 0) return -1;
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

int ora_parse_calls(const char *in, char *out, int cap) {
  Strs lines;
  for (const auto &c : parse_calls(split_lines(in))) lines.push_back(c.first + "\t" + (c.second ? "1" : "0"));
  return copy_out(join_lines(lines), out, cap);
}

int ora_unmap(const ora_ref *r, const char *features, uint32_t *out, int cap) {
  try {
    std::vector<uint32_t> ids = unmap(split_lines(features), r->r);
    if ((int)ids.size() > cap) { g_err = "output buffer too small"; return -1; }
    for (size_t i = 0; i < ids.size(); ++i) out[i] = ids[i];
    return (int)ids.size();
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

int ora_reference_sequence_data(const ora_ref *r, char *out, int cap) {
  try {
    auto data = get_reference_sequence_data(r->r);
    Strs lines;
    for (size_t i = 0; i < data.first.size(); ++i) lines.push_back(data.second[i] + "\t" + data.first[i].to_string());
    return copy_out(join_lines(lines), out, cap);
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

int ora_sort_score_vector(const char *keys, int n_rows, int32_t *order) {
  std::vector<Strs> k;
  Strs rows = split_lines(keys);
  if ((int)rows.size() != n_rows && !(n_rows == 0 && rows.empty())) { g_err = "row count does not match"; return -1; }
  for (const auto &row : rows) k.push_back(split_tabs(row));
  std::vector<int32_t> o = sort_score_vector_order(k);
  for (size_t i = 0; i < o.size(); ++i) order[i] = o[i];
  return 0;
}

int ora_process_class_to_features(const ora_ref *r, const ora_config *cfg, const uint32_t *cls, int n,
                                  int ignore_rollup, char *out, int cap) {
  try {
    return copy_out(join_lines(process_equivalence_class_to_feature_list(std::vector<uint32_t>(cls, cls + n), r->r,
                                                                         *cfg, ignore_rollup != 0)),
                    out, cap);
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

ora_result *ora_call(const ora_index *ix, const ora_ref *ref, const ora_config *cfg, const uint8_t *r1,
                     const uint64_t *r1_off, const uint8_t *r2, const uint64_t *r2_off, uint64_t n, int n_threads,
                     int keep_per_read) {
  Result *r = call(*ix->ix, ref->r, *cfg, r1, r1_off, r2, r2_off, n, n_threads, keep_per_read != 0);
  if (!r) return nullptr;
  ora_result *o = new ora_result();
  o->r = r;
  return o;
}
ora_result *ora_call_umi(const ora_index *ix, const ora_ref *ref, const ora_config *cfg, const uint8_t *r1,
                         const uint64_t *r1_off, const uint8_t *r2, const uint64_t *r2_off, const uint8_t *q1,
                         const uint8_t *q2, const uint8_t *skip1, const uint8_t *skip2, const uint32_t *segment,
                         uint64_t n, int keep_per_read) {
  UmiExtras ex;
  ex.q1 = q1;
  ex.q2 = q2;
  ex.skip1 = skip1;
  ex.skip2 = skip2;
  Result *r = call_umi(*ix->ix, ref->r, *cfg, r1, r1_off, r2, r2_off, ex, segment, n, keep_per_read != 0);
  if (!r) return nullptr;
  ora_result *o = new ora_result();
  o->r = r;
  return o;
}
uint32_t ora_result_row_segment(const ora_result *r, uint64_t i) {
  return i < r->r->row_segment.size() ? r->r->row_segment[i] : 0u;
}
const int32_t *ora_result_align_len(const ora_result *r, int m) {
  return r->r->align_len[m].empty() ? nullptr : r->r->align_len[m].data();
}
void ora_result_free(ora_result *r) { if (r) { delete r->r; delete r; } }
uint64_t ora_result_n_rows(const ora_result *r) { return r->r->rows.size(); }
const char *ora_result_row(const ora_result *r, uint64_t i, int32_t *count) {
  *count = r->r->rows[i].second;
  return r->r->rows[i].first.c_str();
}
const int32_t *ora_result_reason(const ora_result *r, int m) { return r->r->reason[m].data(); }
const int32_t *ora_result_score(const ora_result *r, int m) { return r->r->score[m].data(); }
const int32_t *ora_result_mismatch(const ora_result *r, int m) { return r->r->mism[m].data(); }
const uint64_t *ora_result_class_hash(const ora_result *r, int m) { return r->r->class_hash[m].data(); }
const uint8_t *ora_result_kept(const ora_result *r, int m) { return r->r->kept[m].data(); }
const uint8_t *ora_result_counted(const ora_result *r) { return r->r->counted.data(); }
void ora_result_counters(const ora_result *r, uint64_t *c8) { memcpy(c8, r->r->counters, sizeof(uint64_t) * 8); }

}  // extern "C"
