// flat_index.cpp -- sort-based construction of the flat device index (see flat_index.h).
//
// Semantics (external crate behaviour at src/bin/main.rs:121-128):
//   * every 30-mer of every row, stranded; colour = ascending set of row ids containing it
//   * extensions of a k-mer = union of the neighbouring bases observed in the rows
//   * k-mers x -> y are joined iff x has one right extension, y one left extension and equal colour;
//     chains of joins are the unitigs; a pure cycle is cut in front of its smallest k-mer
//   * dictionary k-mer -> (unitig, offset); edges of a unitig = unitigs starting/ending with the
//     neighbour k-mer of its terminal k-mers
#include "flat_index.h"
#include "threads.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <thread>
#include <unordered_map>

namespace nimble {
namespace {

struct Occ {
  uint64_t kmer;
  uint32_t seq;
  uint32_t exts;  // lext | rext << 4
};

inline int popc4(uint32_t m) { return __builtin_popcount(m & 0xF); }

// NIMBLE_INDEX_TIMING=1: phase times of the build on stderr
struct PhaseTimer {
  bool on = getenv("NIMBLE_INDEX_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[nimble index] %-28s %.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

// threads of the build: the CPUs this process may really use (affinity mask, cgroup quota -- a container granted 16 of a
// host's 256 CPUs reads 256 from hardware_concurrency()), at most 32; NIMBLE_INDEX_THREADS overrides.  The reference
// takes the number from its caller (`num_cores`, src/bin/main.rs:121-128).
unsigned build_threads() {
  if (const char *e = getenv("NIMBLE_INDEX_THREADS")) return (unsigned)std::max(1, atoi(e));
  return std::min(threads::usable_cpus(), 32u);
}

// fn(t) for t in [0, threads): on threads of their own where the system starts them, on the calling thread where it does
// not (csrc/threads.h -- a refused thread start must not take the process down)
template <class F>
void parallel_threads(unsigned n_threads, F fn) {
  threads::run_indexed(n_threads, fn);
}

// fn(t, lo, hi) over [0, n) in `threads` contiguous slices (slice t = [n t / threads, n (t+1) / threads))
template <class F>
void parallel_slices(size_t n, unsigned threads, F fn) {
  if (threads <= 1 || n < 4096) {
    fn(0u, (size_t)0, n);
    return;
  }
  parallel_threads(threads, [&](unsigned t) { fn(t, n * t / threads, n * (t + 1) / threads); });
}

}  // namespace

void build_flat_index(const uint8_t *seqs, const uint64_t *off, uint32_t n_seqs, FlatIndex &out) {
  PhaseTimer timer;
  const unsigned threads = build_threads();
  // 1. all k-mer occurrences
  uint64_t total = 0;
  for (uint32_t s = 0; s < n_seqs; ++s) {
    uint64_t len = off[s + 1] - off[s];
    if (len >= KMER) total += len - KMER + 1;
  }
  std::vector<Occ> occ;
  occ.reserve(total);
  for (uint32_t s = 0; s < n_seqs; ++s) {
    const uint8_t *p = seqs + off[s];
    uint64_t len = off[s + 1] - off[s];
    if (len < KMER) continue;
    uint64_t km = 0;
    for (uint32_t i = 0; i < KMER - 1; ++i) km = (km << 2) | encode_base(p[i]);
    for (uint64_t pos = 0; pos + KMER <= len; ++pos) {
      km = ((km << 2) | encode_base(p[pos + KMER - 1])) & KMER_MASK;
      uint32_t l = pos > 0 ? 1u << encode_base(p[pos - 1]) : 0u;
      uint32_t r = pos + KMER < len ? 1u << encode_base(p[pos + KMER]) : 0u;
      occ.push_back(Occ{km, s, l | (r << 4)});
    }
  }
  timer.lap("k-mer occurrences");
  {
    // sort by (k-mer, row): 256 buckets on the top k-mer bits, filled and then sorted by the worker threads
    auto less = [](const Occ &a, const Occ &b) { return a.kmer != b.kmer ? a.kmer < b.kmer : a.seq < b.seq; };
    if (threads <= 1 || occ.size() < (1u << 16)) {
      std::sort(occ.begin(), occ.end(), less);
    } else {
      constexpr unsigned NB = 256, SH = 2 * KMER - 8;
      std::vector<std::vector<uint64_t>> cnt(threads, std::vector<uint64_t>(NB, 0));
      parallel_slices(occ.size(), threads, [&](unsigned t, size_t lo, size_t hi) {
        std::vector<uint64_t> &c = cnt[t];
        for (size_t i = lo; i < hi; ++i) c[occ[i].kmer >> SH]++;
      });
      // bucket starts, and inside a bucket one range per slice (slices in order: the fill is deterministic)
      std::vector<uint64_t> start(NB + 1, 0);
      for (unsigned b = 0; b < NB; ++b) {
        uint64_t run = start[b];
        for (unsigned t = 0; t < threads; ++t) {
          const uint64_t c = cnt[t][b];
          cnt[t][b] = run;
          run += c;
        }
        start[b + 1] = run;
      }
      std::vector<Occ> tmp(occ.size());
      parallel_slices(occ.size(), threads, [&](unsigned t, size_t lo, size_t hi) {
        std::vector<uint64_t> &c = cnt[t];
        for (size_t i = lo; i < hi; ++i) tmp[c[occ[i].kmer >> SH]++] = occ[i];
      });
      occ.swap(tmp);
      std::vector<Occ>().swap(tmp);
      std::atomic<unsigned> nextb{0};
      parallel_threads(threads, [&](unsigned) {
        for (unsigned b = nextb++; b < NB; b = nextb++)
          std::sort(occ.begin() + (long)start[b], occ.begin() + (long)start[b + 1], less);
      });
    }
  }
  timer.lap("sort");

  // 2. distinct k-mers, their extension masks and colour classes (interned by content)
  std::vector<uint64_t> kmers;
  std::vector<uint8_t> exts;
  std::vector<uint32_t> colour;
  out.col_off.assign(1, 0);
  out.col_ids.clear();
  std::unordered_multimap<uint64_t, uint32_t> by_hash;
  std::vector<uint32_t> ids;
  for (size_t i = 0; i < occ.size();) {
    size_t j = i;
    uint32_t e = 0;
    ids.clear();
    while (j < occ.size() && occ[j].kmer == occ[i].kmer) {
      e |= occ[j].exts;
      if (ids.empty() || ids.back() != occ[j].seq) ids.push_back(occ[j].seq);
      ++j;
    }
    uint64_t h = class_hash_init();
    for (uint32_t v : ids) h = class_hash_step(h, v);
    h = class_hash_final(h, (uint32_t)ids.size());
    uint32_t cid = UINT32_MAX;
    auto range = by_hash.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
      uint32_t c = it->second;
      uint32_t len = out.col_off[c + 1] - out.col_off[c];
      if (len == ids.size() && std::equal(ids.begin(), ids.end(), out.col_ids.begin() + out.col_off[c])) {
        cid = c;
        break;
      }
    }
    if (cid == UINT32_MAX) {
      cid = (uint32_t)out.col_off.size() - 1;
      out.col_ids.insert(out.col_ids.end(), ids.begin(), ids.end());
      out.col_off.push_back((uint32_t)out.col_ids.size());
      by_hash.emplace(h, cid);
    }
    kmers.push_back(occ[i].kmer);
    exts.push_back((uint8_t)e);
    colour.push_back(cid);
    i = j;
  }
  std::vector<Occ>().swap(occ);
  timer.lap("distinct k-mers + colours");
  const size_t n = kmers.size();
  out.n_kmers = n;
  out.n_colours = out.col_off.size() - 1;

  auto find = [&](uint64_t km) -> size_t {
    size_t p = std::lower_bound(kmers.begin(), kmers.end(), km) - kmers.begin();
    if (p >= n || kmers[p] != km) throw std::runtime_error("index build: neighbour k-mer missing");
    return p;
  };

  // 3. join relation next[]/prev[] (UINT32_MAX = none)
  if (n >= UINT32_MAX) throw std::runtime_error("index build: too many k-mers");
  std::vector<uint32_t> next(n, UINT32_MAX), prev(n, UINT32_MAX);
  // (a k-mer with one left extension has one predecessor: no two i write the same prev[j])
  parallel_slices(n, threads, [&](unsigned, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      uint32_t r = exts[i] >> 4;
      if (popc4(r) != 1) continue;
      uint64_t nk = ((kmers[i] << 2) | (uint64_t)__builtin_ctz(r)) & KMER_MASK;
      size_t j = find(nk);
      if (popc4(exts[j] & 0xF) != 1 || colour[j] != colour[i]) continue;
      next[i] = (uint32_t)j;
      prev[j] = (uint32_t)i;
    }
  });
  timer.lap("join relation");
  // cut pure cycles in front of their smallest k-mer: walk chains from heads first, then leftovers
  std::vector<uint8_t> seen(n, 0);
  std::vector<uint32_t> heads;
  for (size_t i = 0; i < n; ++i)
    if (prev[i] == UINT32_MAX) {
      heads.push_back((uint32_t)i);
      for (uint32_t c = (uint32_t)i; c != UINT32_MAX; c = next[c]) seen[c] = 1;
    }
  for (size_t i = 0; i < n; ++i)
    if (!seen[i]) {  // smallest member of a cycle (ascending scan)
      uint32_t p = prev[i];
      next[p] = UINT32_MAX;
      prev[i] = UINT32_MAX;
      heads.push_back((uint32_t)i);
      for (uint32_t c = (uint32_t)i; c != UINT32_MAX; c = next[c]) seen[c] = 1;
    }
  // Node ids follow the library, not the k-mer order: unitigs are numbered by the first row of their colour class (rows of
  // one gene family are adjacent, forward and reverse rows interleaved), then by head k-mer.  A read's walk stays inside
  // its family, so reads grouped by seed node (sorted inputs, BAM groups) keep to one stretch of the node records -- the
  // stretch an XCD works on stays in its L2 however large the library is.
  std::sort(heads.begin(), heads.end(), [&](uint32_t a, uint32_t b) {
    const uint32_t ra = out.col_ids[out.col_off[colour[a]]], rb = out.col_ids[out.col_off[colour[b]]];
    return ra != rb ? ra < rb : a < b;
  });

  timer.lap("chains");
  // 4. unitigs
  const size_t n_nodes = heads.size();
  out.n_nodes = n_nodes;
  out.node_rec.assign(n_nodes * 16, 0);
  out.node_ledge.assign(n_nodes * 4, 0);
  std::vector<uint32_t> kmer_node(n), kmer_off(n);
  std::vector<uint32_t> tail(n_nodes), node_len(n_nodes);
  parallel_slices(n_nodes, threads, [&](unsigned, size_t lo, size_t hi) {
    for (size_t nd = lo; nd < hi; ++nd) {
      uint32_t o = 0, c = heads[nd], last = c;
      for (; c != UINT32_MAX; c = next[c]) {
        kmer_node[c] = (uint32_t)nd;
        kmer_off[c] = o++;
        last = c;
      }
      tail[nd] = last;
      node_len[nd] = o;
    }
  });
  uint64_t bases = 0;
  for (size_t nd = 0; nd < n_nodes; ++nd) {
    const uint32_t last = tail[nd];
    uint64_t len = (uint64_t)node_len[nd] + KMER - 1;
    if (bases + len >= (1ULL << 32)) throw std::runtime_error("index build: unitig buffer exceeds 2^32 bases");
    if (len >= (1u << 24)) throw std::runtime_error("index build: unitig longer than 2^24 bases");
    const uint32_t e = (uint32_t)(exts[heads[nd]] & 0xF) | ((uint32_t)(exts[last] >> 4) << 4);
    out.node_rec[nd * 16 + 0] = (uint32_t)len | (e << 24);
    out.node_rec[nd * 16 + 1] = colour[heads[nd]];
    out.node_rec[nd * 16 + 2] = (uint32_t)bases;
    bases += len;
  }
  out.unitig_bases = bases;
  out.unitig.assign((bases + 31) / 32 + 4, 0);  // +4 words so a 3-word window never reads past the end
  // neighbouring unitigs share a word of the packed buffer: OR the bases in atomically
  auto put_base = [&](uint64_t pos, uint64_t b) {
    if (b) __atomic_fetch_or(&out.unitig[pos >> 5], b << (62 - 2 * (pos & 31)), __ATOMIC_RELAXED);
  };
  parallel_slices(n_nodes, threads, [&](unsigned, size_t lo, size_t hi) {
    for (size_t nd = lo; nd < hi; ++nd) {
      uint64_t pos = out.node_rec[nd * 16 + 2];
      uint64_t inl[4] = {0, 0, 0, 0};
      uint32_t k = 0;
      auto put = [&](uint64_t b) {
        put_base(pos++, b);
        if (k >= NODE_INLINE_FIRST && k < NODE_INLINE_FIRST + NODE_INLINE_BASES) {
          const uint32_t q = k - NODE_INLINE_FIRST;
          inl[q >> 5] |= b << (62 - 2 * (q & 31));
        }
        ++k;
      };
      uint64_t first = kmers[heads[nd]];
      for (uint32_t b = 0; b < KMER; ++b) put((first >> (2 * (KMER - 1 - b))) & 3);
      for (uint32_t c = next[heads[nd]]; c != UINT32_MAX; c = next[c]) put(kmers[c] & 3);
      for (int w = 0; w < 2; ++w) {
        out.node_rec[nd * 16 + 12 + 2 * w] = (uint32_t)inl[w];
        out.node_rec[nd * 16 + 13 + 2 * w] = (uint32_t)(inl[w] >> 32);
      }
    }
  });
  timer.lap("unitigs");
  // 5. edges
  parallel_slices(n_nodes, threads, [&](unsigned, size_t lo_, size_t hi_) {
  for (size_t nd = lo_; nd < hi_; ++nd) {
    uint32_t e = out.node_rec[nd * 16 + 0] >> 24;
    uint64_t firstk = kmers[heads[nd]], lastk = kmers[tail[nd]];
    for (uint32_t b = 0; b < 4; ++b) {
      if ((e >> 4) & (1u << b)) {
        size_t j = find(((lastk << 2) | b) & KMER_MASK);
        if (kmer_off[j] != 0) throw std::runtime_error("index build: right edge does not land on a unitig start");
        out.node_rec[nd * 16 + 8 + b] = kmer_node[j];
      }
      if (e & (1u << b)) {
        size_t j = find(((uint64_t)b << (2 * (KMER - 1))) | (firstk >> 2));
        if (j != tail[kmer_node[j]]) throw std::runtime_error("index build: left edge does not land on a unitig end");
        out.node_ledge[nd * 4 + b] = kmer_node[j];
      }
    }
  }
  });
  timer.lap("edges");
  // 6. dictionary, load factor <= 0.25 (before rounding the table up to a power of two).  A wave probes for 64 reads at once and
  // waits for its worst lane: at load 0.31 (the bench library in a table sized for 0.5) the probes of a wave of bench reads went
  // 6.5 times to a second, third ... slot, each a round trip beyond L2 for one or two lanes; at 0.15: k_align -3.4 % on the bench
  // reads, -5.5 % on foreign reads; twice that table again (256 MiB, past the Infinity Cache) gives half of it back
  // (profiles/r04_experiments.txt 26)
  uint64_t slots = 64;
  uint32_t log2_slots = 6;
  static const uint64_t per_kmer = [] {
    const char *e = getenv("NIMBLE_HT_SLOTS_PER_KMER");  // (experiments: dictionary slots per k-mer, before rounding up to 2^k)
    const long v = e ? atol(e) : 4;
    return (uint64_t)(v >= 2 && v <= 64 ? v : 4);
  }();
  while (slots < per_kmer * (uint64_t)n + 2) { slots <<= 1; ++log2_slots; }
  if (log2_slots + 2 > 32) throw std::runtime_error("index build: too many k-mers for the 32-bit slot hash");
  out.ht_slots = slots;
  out.ht_log2 = log2_slots;
  out.ht.assign(slots * 2, 0);
  for (uint64_t s = 0; s < slots; ++s) out.ht[2 * s] = HT_EMPTY;
  for (size_t i = 0; i < n; ++i) {
    uint64_t h = kmer_slot(kmers[i], log2_slots);
    while (out.ht[2 * h] != HT_EMPTY) h = (h + 1) & (slots - 1);
    out.ht[2 * h] = kmers[i];
    out.ht[2 * h + 1] = ((uint64_t)kmer_node[i] << 32) | kmer_off[i];
  }
  timer.lap("dictionary");
  // presence filter: 2 bits in each of 7 lines per k-mer, sized for <= ~8 % bit density (the 24 shared bits allow up to 2^24
  // lines).  A false positive costs its wave a dictionary probe for ONE lane (a hash, a fetch from beyond L2): at 16 % density a
  // wave of foreign reads ran 1.5 such probes per filter round; 8 %: k_align -1.5 % on the bench reads, -2.6 % on foreign reads;
  // 4 %: another half per cent (profiles/r04_experiments.txt 25)
  static const double density = [] {
    const char *e = getenv("NIMBLE_FILTER_DENSITY");  // (experiments: bit density the filter is sized for)
    const double d = e ? atof(e) : 0.08;
    return d > 0.001 && d < 0.9 ? d : 0.08;
  }();
  out.bm_lines_log2 = 8;
  while (out.bm_lines_log2 < 24 && (double)n * SCAN_ROUND * 2.0 > density * 128.0 * (double)(1ull << out.bm_lines_log2))
    ++out.bm_lines_log2;
  out.bitmap.assign((size_t)4 << out.bm_lines_log2, 0);
  // first level in front of it: one bit per possible value of the 12 shared bases (4^12 bits = 2 MiB, small enough to
  // live in an XCD's L2): set when ANY library k-mer can be asked under that value.  A round whose bit is clear has no
  // candidate at all, and its filter line (a fetch from beyond L2) is never loaded.
  out.l1.assign((size_t)1 << (2 * SCAN_SHARED - 5), 0);
  parallel_slices(n, threads, [&](unsigned, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      const uint32_t bits = round_bits(kmers[i]);
      const uint32_t half = (bits >> 12) & 1u;
      const uint32_t b1 = (bits & 63u) + 64u * half, b2 = ((bits >> 6) & 63u) + 64u * half;
      for (uint32_t j = 0; j < SCAN_ROUND; ++j) {
        const uint64_t shared = round_shared_of_kmer(kmers[i], j);
        const uint64_t line = round_line(shared, out.bm_lines_log2);
        __atomic_fetch_or(&out.bitmap[line * 4 + (b1 >> 5)], 1u << (b1 & 31), __ATOMIC_RELAXED);
        __atomic_fetch_or(&out.bitmap[line * 4 + (b2 >> 5)], 1u << (b2 & 31), __ATOMIC_RELAXED);
        __atomic_fetch_or(&out.l1[shared >> 5], 1u << (shared & 31), __ATOMIC_RELAXED);
      }
    }
  });
  {
    // worth its L2 space only while it rejects most rounds of a read that is not from the library
    uint64_t set = 0;
    for (uint32_t w : out.l1) set += (uint64_t)__builtin_popcount(w);
    out.l1_density = (double)set / (double)((uint64_t)out.l1.size() * 32);
    if (out.l1_density > 0.5) std::vector<uint32_t>().swap(out.l1);
  }
  timer.lap("presence filter");
  {
    // 29-mers with more than one left flank: the sorted k-mers fall into four runs by their first base, each run sorted
    // by the 29 bases behind it -- a four-way merge finds the 29-mers that head more than one run
    const uint64_t SUF = (1ULL << (2 * (KMER - 1))) - 1ULL;
    size_t at[5];
    for (uint64_t b = 0; b <= 4; ++b)
      at[b] = b == 4 ? n : (size_t)(std::lower_bound(kmers.begin(), kmers.end(), b << (2 * (KMER - 1))) - kmers.begin());
    size_t cur[4] = {at[0], at[1], at[2], at[3]};
    std::vector<uint64_t> multi;
    for (;;) {
      uint64_t lo = ~0ULL;
      for (int b = 0; b < 4; ++b)
        if (cur[b] < at[b + 1] && (kmers[cur[b]] & SUF) < lo) lo = kmers[cur[b]] & SUF;
      if (lo == ~0ULL) break;
      int heads = 0;
      for (int b = 0; b < 4; ++b)
        if (cur[b] < at[b + 1] && (kmers[cur[b]] & SUF) == lo) {
          ++heads;
          ++cur[b];
        }
      if (heads > 1) multi.push_back(lo);
    }
    out.n_mleft = multi.size();
    out.mleft_log2 = 9;  // 32 table bits per entry, at least 4 KiB
    while (out.mleft_log2 < 26 && (64ULL << out.mleft_log2) < 32ULL * multi.size()) ++out.mleft_log2;
    out.mleft.assign((size_t)1 << out.mleft_log2, 0ULL);
    for (uint64_t s29 : multi) {
      const uint32_t h = mleft_hash(s29);
      out.mleft[h >> (32u - out.mleft_log2)] |= (1ULL << (h & 63u)) | (1ULL << ((h >> 6) & 63u));
    }
  }
  timer.lap("left-flank set");
  // 7. class descriptors
  out.cls_desc.assign(out.n_colours * 4, 0);
  for (size_t c = 0; c < out.n_colours; ++c) {
    uint32_t o = out.col_off[c], l = out.col_off[c + 1] - o;
    make_class_desc(out.col_ids.data() + o, l, &out.cls_desc[c * 4]);
    if (!(out.cls_desc[c * 4] & CLS_MASK_FLAG)) {
      out.all_classes_local = false;
      // a class wider than the 64-row mask form gets a bitmap over the rows it spans (allele families: the rows of
      // a gene are neighbours, so a class of 100 alleles spans a few hundred rows): the device intersects such
      // classes word by word instead of searching id lists
      // (the bitmap starts at a multiple of 64 rows: a 64-row window at a multiple of 64 is then ONE word of it)
      const uint32_t first = out.col_ids[o] & ~63u, span = out.col_ids[o + l - 1] - first + 1u;
      if (l && span <= CLS_BITMAP_MAX_ROWS && out.cls_bits.size() + (span + 63u) / 64u < 0xFFFFFFFEull) {
        const size_t at = out.cls_bits.size();
        out.cls_bits.resize(at + (span + 63u) / 64u, 0ULL);
        for (uint32_t t = 0; t < l; ++t) {
          const uint32_t r = out.col_ids[o + t] - first;
          out.cls_bits[at + (r >> 6)] |= 1ULL << (r & 63u);
        }
        out.cls_desc[c * 4 + 1] = first;
        out.cls_desc[c * 4 + 2] = span;
        out.cls_desc[c * 4 + 3] = (uint32_t)at + 1u;
        out.max_bitmap_words = std::max(out.max_bitmap_words, (span + 63u) / 64u);
      } else {
        out.all_wide_have_bitmaps = false;
      }
    }
  }
  for (size_t nd = 0; nd < n_nodes; ++nd) {
    const uint32_t c = out.node_rec[nd * 16 + 1];
    for (int k = 0; k < 4; ++k) out.node_rec[nd * 16 + 3 + k] = out.cls_desc[(size_t)c * 4 + k];
  }
  // Component windows.  The rows of a library row's k-mers hang together through the graph's edges, so a connected
  // component of unitigs is a set of whole rows and the classes of two components share no row.  When the classes of
  // every component span fewer than 64 rows, each node record gets its class mask relative to the component's first row.
  if (out.all_classes_local && n_nodes) {
    std::vector<uint32_t> parent(n_nodes);
    for (size_t i = 0; i < n_nodes; ++i) parent[i] = (uint32_t)i;
    auto root = [&](uint32_t x) {
      while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
      }
      return x;
    };
    for (size_t nd = 0; nd < n_nodes; ++nd) {
      const uint32_t e = out.node_rec[nd * 16] >> 24;
      for (uint32_t b = 0; b < 4; ++b)
        if ((e >> 4) & (1u << b)) {
          const uint32_t ra = root((uint32_t)nd), rb = root(out.node_rec[nd * 16 + 8 + b]);
          if (ra != rb) parent[ra < rb ? rb : ra] = ra < rb ? ra : rb;
        }
    }
    std::vector<uint32_t> lo(n_nodes, 0xFFFFFFFFu), hi(n_nodes, 0);
    for (size_t nd = 0; nd < n_nodes; ++nd) {
      const uint32_t r = root((uint32_t)nd);
      const uint32_t base = out.node_rec[nd * 16 + 4];
      const uint64_t m = (uint64_t)out.node_rec[nd * 16 + 5] | ((uint64_t)out.node_rec[nd * 16 + 6] << 32);
      const uint32_t last = base + (m ? 63u - (uint32_t)__builtin_clzll(m) : 0u);
      lo[r] = std::min(lo[r], base);
      hi[r] = std::max(hi[r], last);
    }
    bool fits = true;
    for (size_t nd = 0; nd < n_nodes && fits; ++nd)
      if (parent[nd] == nd && hi[nd] - lo[nd] >= CLS_WINDOW) fits = false;
    if (fits) {
      for (size_t nd = 0; nd < n_nodes; ++nd) {
        const uint32_t r = root((uint32_t)nd);
        const uint32_t base = out.node_rec[nd * 16 + 4];
        const uint64_t m = ((uint64_t)out.node_rec[nd * 16 + 5] | ((uint64_t)out.node_rec[nd * 16 + 6] << 32)) << (base - lo[r]);
        out.node_rec[nd * 16 + 4] = lo[r];
        out.node_rec[nd * 16 + 5] = (uint32_t)m;
        out.node_rec[nd * 16 + 6] = (uint32_t)(m >> 32);
      }
      out.uniform_windows = true;
    }
  }
  timer.lap("class descriptors");
  // Stretch records for the fast walk (flat_index.h): every unitig's bases behind its first k-mer in stretches of 32.
  if (out.uniform_windows && n_nodes) {
    // Order of the records = order in which a walk meets them, as far as one array can have it: what bounds the fast walk
    // is the number of 128-byte lines a CU fetches (a line per ~4 cycles from L2, whatever part of it is used), and a read
    // crosses five or six unitigs.  Unitigs are laid out component by component (a component = the rows of a gene family in
    // one orientation) in topological order of the graph's edges, first come first served (Kahn's algorithm with a queue):
    // the alternatives behind a fork lie side by side, the unitig they merge into right behind them, so a walk reads its
    // records front to back with small gaps -- two or three lines instead of one per unitig.  (Unitigs on cycles, which no
    // topological order reaches, follow at the end of their component.)
    std::vector<uint32_t> order;
    order.reserve(n_nodes);
    {
      std::vector<uint32_t> comp(n_nodes), indeg(n_nodes, 0);
      for (size_t i = 0; i < n_nodes; ++i) comp[i] = (uint32_t)i;
      auto root = [&](uint32_t x) {
        while (comp[x] != x) {
          comp[x] = comp[comp[x]];
          x = comp[x];
        }
        return x;
      };
      for (size_t nd = 0; nd < n_nodes; ++nd) {
        const uint32_t rext = (out.node_rec[nd * 16] >> 28) & 0xFu;
        for (uint32_t b = 0; b < 4; ++b)
          if (rext & (1u << b)) {
            const uint32_t t = out.node_rec[nd * 16 + 8 + b];
            ++indeg[t];
            const uint32_t ra = root((uint32_t)nd), rb = root(t);
            if (ra != rb) comp[ra < rb ? rb : ra] = ra < rb ? ra : rb;  // (the smallest node id names the component)
          }
      }
      // members of every component, in node order
      std::vector<uint32_t> comp_size(n_nodes, 0), comp_at(n_nodes + 1, 0), members(n_nodes);
      for (size_t nd = 0; nd < n_nodes; ++nd) ++comp_size[root((uint32_t)nd)];
      for (size_t c = 0; c < n_nodes; ++c) comp_at[c + 1] = comp_at[c] + comp_size[c];
      {
        std::vector<uint32_t> fill(comp_at.begin(), comp_at.end() - 1);
        for (size_t nd = 0; nd < n_nodes; ++nd) members[fill[root((uint32_t)nd)]++] = (uint32_t)nd;
      }
      std::vector<uint8_t> placed(n_nodes, 0);
      std::vector<uint32_t> queue;
      for (size_t c = 0; c < n_nodes; ++c) {
        if (!comp_size[c]) continue;
        const uint32_t *m = &members[comp_at[c]];
        queue.clear();
        for (uint32_t i = 0; i < comp_size[c]; ++i)
          if (indeg[m[i]] == 0) queue.push_back(m[i]);
        for (size_t q = 0; q < queue.size(); ++q) {
          const uint32_t nd = queue[q];
          placed[nd] = 1;
          order.push_back(nd);
          const uint32_t rext = (out.node_rec[(size_t)nd * 16] >> 28) & 0xFu;
          for (uint32_t b = 0; b < 4; ++b)
            if (rext & (1u << b)) {
              const uint32_t t = out.node_rec[(size_t)nd * 16 + 8 + b];
              if (--indeg[t] == 0) queue.push_back(t);
            }
        }
        for (uint32_t i = 0; i < comp_size[c]; ++i)
          if (!placed[m[i]]) order.push_back(m[i]);
      }
    }
    out.srec_first.resize(n_nodes);
    uint64_t total = 0;
    for (uint32_t nd : order) {
      const uint32_t inf = (out.node_rec[(size_t)nd * 16] & 0xFFFFFFu) - KMER;
      out.srec_first[nd] = (uint32_t)total;
      total += std::max<uint32_t>(1u, (inf + 31u) / 32u);
    }
    if (total < (1ULL << 32)) {
      out.srec.assign(total * 8, 0);
      out.srec_base.assign(total, 0);
      std::vector<uint32_t> many_at(n_nodes, 0);
      uint32_t n_many = 0;
      for (size_t nd = 0; nd < n_nodes; ++nd)
        if (__builtin_popcount((out.node_rec[nd * 16] >> 28) & 0xFu) > 2) many_at[nd] = n_many++;
      out.srec_many.assign((size_t)std::max<uint32_t>(n_many, 1u) * 4, 0);
      parallel_slices(n_nodes, threads, [&](unsigned, size_t lo_, size_t hi_) {
        for (size_t nd = lo_; nd < hi_; ++nd) {
          const uint32_t *nr = &out.node_rec[nd * 16];
          const uint32_t inf = (nr[0] & 0xFFFFFFu) - KMER, rext = (nr[0] >> 28) & 0xFu;
          const uint32_t n_rec = std::max<uint32_t>(1u, (inf + 31u) / 32u);
          const uint64_t seq = (uint64_t)nr[2] + KMER;  // the unitig's base 30 in the packed buffer
          for (uint32_t j = 0; j < n_rec; ++j) {
            uint32_t *r = &out.srec[((size_t)out.srec_first[nd] + j) * 8];
            const uint32_t nb = std::min<uint32_t>(32u, inf - 32u * j);
            uint64_t bases = 0;
            for (uint32_t b = 0; b < nb; ++b) {
              const uint64_t pos = seq + 32u * j + b;
              bases |= ((out.unitig[pos >> 5] >> (62 - 2 * (pos & 31))) & 3ULL) << (62 - 2 * b);
            }
            r[2] = (uint32_t)bases;
            r[3] = (uint32_t)(bases >> 32);
            r[4] = nr[5];
            r[5] = nr[6];
            uint32_t hdr = nb | (rext << 8) | ((nr[3] & 0x7Fu) << 16);
            if (j + 1 == n_rec) {
              hdr |= SREC_LAST;
              if (__builtin_popcount(rext) > 2) {  // a fork with three or four ways out: the neighbours stand in srec_many
                hdr |= SREC_MANY;
                r[6] = many_at[nd];
                for (uint32_t b = 0; b < 4; ++b)
                  if (rext & (1u << b)) out.srec_many[(size_t)many_at[nd] * 4 + b] = out.srec_first[nr[8 + b]];
              } else {
                uint32_t k = 0;
                for (uint32_t b = 0; b < 4 && k < 2; ++b)
                  if (rext & (1u << b)) r[6 + k++] = out.srec_first[nr[8 + b]];
              }
            }
            r[0] = hdr;
            r[1] = nr[1];
            out.srec_base[(size_t)out.srec_first[nd] + j] = nr[4];
          }
        }
      });
    } else {
      out.srec_first.clear();
    }
    timer.lap("stretch records");
  }
}

std::string check_stretch_records(const FlatIndex &fi) {
  if (fi.srec.empty()) return fi.uniform_windows && fi.n_nodes ? "an index with uniform windows has no stretch records" : "";
  const size_t n_rec = fi.srec.size() / 8;
  if (fi.srec_first.size() != fi.n_nodes || fi.srec_base.size() != n_rec) return "srec_first / srec_base have the wrong size";
  auto base_at = [&](uint64_t pos) { return (uint32_t)((fi.unitig[pos >> 5] >> (62 - 2 * (pos & 31))) & 3ULL); };
  std::vector<uint8_t> owned(n_rec, 0);
  for (size_t nd = 0; nd < fi.n_nodes; ++nd) {
    const uint32_t *nr = &fi.node_rec[nd * 16];
    const uint32_t len = nr[0] & 0xFFFFFFu, rext = (nr[0] >> 28) & 0xFu, inf = len - KMER;
    const uint32_t want_rec = std::max<uint32_t>(1u, (inf + 31u) / 32u), first = fi.srec_first[nd];
    if ((size_t)first + want_rec > n_rec) return "a unitig's records lie beyond the array";
    uint32_t covered = 0;
    for (uint32_t j = 0; j < want_rec; ++j) {
      const uint32_t *r = &fi.srec[((size_t)first + j) * 8];
      if (owned[first + j]++) return "two unitigs share a record";
      const uint32_t nb = r[0] & 63u;
      if (nb > 32u || (j + 1 < want_rec && nb != 32u)) return "a stretch that is not the unitig's last holds fewer than 32 bases";
      if (((r[0] >> 8) & 0xFu) != rext) return "a record's extension bits differ from its unitig's";
      if (((r[0] & SREC_LAST) != 0) != (j + 1 == want_rec)) return "SREC_LAST is not on the unitig's last stretch only";
      if (((r[0] >> 16) & 0x7Fu) != (nr[3] & 0x7Fu) || !(nr[3] & CLS_MASK_FLAG)) return "class length / form differs";
      if (r[1] != nr[1]) return "colour differs";
      if (r[4] != nr[5] || r[5] != nr[6]) return "class mask differs";
      if (fi.srec_base[first + j] != nr[4]) return "srec_base is not the component's first row";
      const uint64_t bases = (uint64_t)r[2] | ((uint64_t)r[3] << 32);
      for (uint32_t b = 0; b < nb; ++b)
        if (((bases >> (62 - 2 * b)) & 3ULL) != base_at((uint64_t)nr[2] + KMER + covered + b)) return "a stretch's bases differ from the unitig's";
      if (nb < 32u && (bases << (2 * nb)) != 0 && nb != 0) return "bits behind a stretch's last base";
      covered += nb;
      if (j + 1 == want_rec) {
        const uint32_t ways = (uint32_t)__builtin_popcount(rext);
        if (((r[0] & SREC_MANY) != 0) != (ways > 2)) return "SREC_MANY does not mark the forks with more than two ways out";
        uint32_t k = 0;
        for (uint32_t b = 0; b < 4; ++b)
          if (rext & (1u << b)) {
            const uint32_t target = fi.srec_first[nr[8 + b]];
            const uint32_t have = ways > 2 ? fi.srec_many[(size_t)r[6] * 4 + b] : r[6 + k];
            if (have != target) return "a right neighbour's record is wrong";
            ++k;
          }
      }
    }
    if (covered != inf) return "a unitig's stretches do not add up to its bases behind the first k-mer";
  }
  for (size_t i = 0; i < n_rec; ++i)
    if (!owned[i]) return "a record belongs to no unitig";
  return "";
}

}  // namespace nimble
