// flat_index.h -- host-side builder of the device-resident index (flat arrays, no pointers).
//
// Replaces debruijn_mapping::build_index::<Kmer30> (call sites src/bin/main.rs:121-128,
// tests/utils.rs:48-51, src/align.rs:1007-1012 of the reference): a stranded, coloured, compacted
// de Bruijn graph over all 30-mers of the library rows plus an exact k-mer dictionary.  The layout is
// designed for the gfx950 walk kernel: 16-byte hash slots, 16-byte node records, 2-bit packed
// unitigs, CSR colour classes.  Built once per library on the host (the reference does the same on
// the CPU with `num_cores` threads); it is not on the per-read path.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace nimble {

constexpr uint32_t KMER = 30;
constexpr uint64_t KMER_MASK = (1ULL << (2 * KMER)) - 1;
constexpr uint64_t HT_EMPTY = ~0ULL;
constexpr uint32_t NODE_INLINE_BASES = 64;   // bases a node record holds inline ...
constexpr uint32_t NODE_INLINE_FIRST = KMER;  // ... starting at this base: the walk never compares a unitig's first k-mer
                                              // (it is the seed, or the overlap with the unitig it came from)
constexpr uint32_t SREC_LAST = 1u << 12, SREC_MANY = 1u << 13;  // stretch-record header bits (FlatIndex::srec)
constexpr uint32_t CLS_WINDOW = 64;          // a class whose rows span < 64 is stored as base + 64-bit mask
constexpr uint32_t CLS_MASK_FLAG = 0x80000000u;
constexpr uint32_t CLS_BITMAP_MAX_ROWS = 1u << 16;  // widest span a static class gets a row bitmap for (8 KiB)  // set in the descriptor's len word when the mask form is valid

// shared by host build and device kernels ----------------------------------------------------
#if defined(__HIPCC__)
#define NIMBLE_HD __host__ __device__ inline
#else
#define NIMBLE_HD inline
#endif

NIMBLE_HD uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
// slot of a k-mer in the dictionary / presence bitmap: the 60-bit key folded to 32 bits, one 32-bit
// multiply (Fibonacci hashing), top log2_slots bits (log2_slots <= 32)
NIMBLE_HD uint64_t kmer_slot(uint64_t km, uint32_t log2_slots) {
  const uint32_t f = (uint32_t)km ^ (uint32_t)(km >> 29);
  return (uint64_t)((f * 0x9E3779B1u) >> (32u - log2_slots));
}
// Round-anchored presence filter.  A seed scan looks at SCAN_ROUND = 7 positions p, p+3, .., p+18 per
// round; the 7 k-mers of a round share the 12 bases [p+18, p+30).  Those 24 bits select a 128-bit filter
// line, the k-mer itself selects the bit inside the line, so ONE 16-byte load answers a whole round.
// At build time every indexed k-mer sets its two bits in the 7 lines it can be asked under (slot j of a round
// <-> shared bases at offset 3(6-j) of the k-mer).
constexpr uint32_t SCAN_ROUND = 7;
constexpr uint32_t SCAN_SHARED = 30 - 3 * (SCAN_ROUND - 1);  // 12 bases
NIMBLE_HD uint64_t round_line(uint64_t shared, uint32_t lines_log2) {
  return (uint64_t)(((uint32_t)shared * 0x85EBCA6Bu) >> (32u - lines_log2));
}
// two filter bits per k-mer, both in one 64-bit half of the 128-bit line (Bloom k = 2 inside the half): bit
// positions in [0,6) and [6,12) of the result, the half in bit 12 -- the test is two 64-bit shifts of one word
NIMBLE_HD uint32_t round_bits(uint64_t km) {
  const uint32_t f = (uint32_t)km ^ (uint32_t)(km >> 31);
  return (f * 0xC2B2AE35u) >> 19;  // 13 bits
}
// the shared bases of k-mer km when it sits in slot j of a round
NIMBLE_HD uint64_t round_shared_of_kmer(uint64_t km, uint32_t j) {
  const uint32_t start = 3u * (SCAN_ROUND - 1u - j);  // offset of the shared bases inside the k-mer
  return (km >> (2u * (30u - start - SCAN_SHARED))) & ((1ULL << (2u * SCAN_SHARED)) - 1ULL);
}

// "Several left flanks" set (local re-seed of the walk, kernels.hip): the 29-mers S for which MORE THAN ONE base b makes
// b.S a library k-mer, as a small Bloom table (two bits in one 64-bit word).  A walk that broke on read base p inside a
// unitig knows that a.S is in the library (a = the unitig's base there, S = the 29 bases behind it, equal in read and
// graph); when S is not in this set, a is the ONLY such base, so the read's k-mer at p (its own base b != a, then S) is
// absent -- without asking the dictionary or the presence filter.
NIMBLE_HD uint32_t mleft_hash(uint64_t s29) {
  uint32_t f = (uint32_t)s29 ^ (uint32_t)(s29 >> 27);
  f *= 0x85EBCA6Bu;
  f ^= f >> 15;
  f *= 0xC2B2AE35u;
  return f ^ (f >> 13);
}
NIMBLE_HD bool mleft_maybe(const uint64_t *table, uint32_t log2_words, uint64_t s29) {
  const uint32_t h = mleft_hash(s29);
  const uint64_t w = table[h >> (32u - log2_words)];
  return ((w >> (h & 63u)) & (w >> ((h >> 6) & 63u)) & 1ULL) != 0;
}

// content hash of an equivalence class (ascending ids); streaming form
NIMBLE_HD uint64_t class_hash_init() { return 0x9E3779B97F4A7C15ULL; }
NIMBLE_HD uint64_t class_hash_step(uint64_t h, uint32_t id) {
  h = (h ^ (uint64_t)id) * 0xff51afd7ed558ccdULL;
  return h ^ (h >> 29);
}
NIMBLE_HD uint64_t class_hash_final(uint64_t h, uint32_t len) { return mix64(h ^ ((uint64_t)len * 0x9FB21C651E98DF25ULL)); }
// Hash of a class for interning.  A class that fits the mask form (rows within a 64-row window) is hashed
// from (len, base, mask); any other class from its ids.  Both sides (host seeding, device lookup) apply the
// same rule, which depends on the content only.
NIMBLE_HD uint64_t class_hash_mask(uint32_t len, uint32_t base, uint64_t mask) {
  return mix64(mask ^ (((uint64_t)base << 32) | (uint64_t)len) * 0x9FB21C651E98DF25ULL);
}
NIMBLE_HD uint64_t class_hash_of_ids(const uint32_t *ids, uint32_t len) {
  if (len && ids[len - 1] - ids[0] < 64u) {
    uint64_t m = 0;
    for (uint32_t t = 0; t < len; ++t) m |= 1ULL << (ids[t] - ids[0]);
    return class_hash_mask(len, ids[0], m);
  }
  uint64_t h = class_hash_init();
  for (uint32_t t = 0; t < len; ++t) h = class_hash_step(h, ids[t]);
  return class_hash_final(h, len);
}

// DnaString::from_acgt_bytes: A/a C/c G/g T/t -> 0..3, everything else -> 0
NIMBLE_HD uint32_t encode_base(uint32_t c) {
  uint32_t l = c | 0x20u;
  uint32_t v = (l >> 1) & 3u;  // a->0 c->1 g->3 t->2
  v ^= v >> 1;                 // a->0 c->1 g->2 t->3
  bool ok = (l == 'a') | (l == 'c') | (l == 'g') | (l == 't');
  return ok ? v : 0u;
}

// descriptor of a class given its ascending ids (host build and device interning share this)
NIMBLE_HD void make_class_desc(const uint32_t *ids, uint32_t len, uint32_t desc[4]) {
  desc[0] = len;
  desc[1] = 0;
  desc[2] = 0;
  desc[3] = 0;
  if (len == 0) return;
  const uint32_t base = ids[0];
  if (ids[len - 1] - base >= 64u) return;
  uint64_t m = 0;
  for (uint32_t t = 0; t < len; ++t) m |= 1ULL << (ids[t] - base);
  desc[0] = len | 0x80000000u;
  desc[1] = base;
  desc[2] = (uint32_t)m;
  desc[3] = (uint32_t)(m >> 32);
}

// intern table slot: high 32 bits = tag (never 0), low 32 bits = class id or INTERN_PENDING
constexpr uint32_t INTERN_PENDING = 0xFFFFFFFFu;
NIMBLE_HD uint32_t intern_tag(uint64_t h) { return (uint32_t)(h >> 32) | 1u; }

struct FlatIndex {
  // exact dictionary k-mer -> (node, offset): slot = {key, node<<32 | offset}; linear probing
  std::vector<uint64_t> ht;  // 2 x u64 per slot
  uint64_t ht_slots = 0;     // power of two
  uint32_t ht_log2 = 0;
  // round-anchored presence filter (see round_line / round_bit): lines of 128 bits
  std::vector<uint32_t> bitmap;      // 4 x u32 per line
  uint32_t bm_lines_log2 = 0;
  // first level of the filter: one bit per value of the 12 shared bases (empty when more than half the bits are set)
  std::vector<uint32_t> l1;
  double l1_density = 0.0;
  // 29-mers with several left flanks among the library k-mers (mleft_maybe): 2^mleft_log2 words
  std::vector<uint64_t> mleft;
  uint32_t mleft_log2 = 0;
  uint64_t n_mleft = 0;
  // node record, one 64-byte line per unitig so that a hop costs one dependent memory level and the
  // class intersection needs no load at all:
  //   u32[0]      = len (bases, low 24 bits) | exts (lext | rext<<4) << 24
  //   u32[1]      = colour (class id)
  //   u32[2]      = seq_start (base offset into unitig)
  //   u32[3]      = class descriptor word 0: class length | CLS_MASK_FLAG
  //   u32[4]      = class base row
  //   u32[5..6]   = class mask (lo, hi)
  //   u32[7]      = 0
  //   u32[8..11]  = right-edge target node per base
  //   u32[12..15] = bases 30..93 (2 x u64, low word first), base 30 + i at word i>>5, bits 62-2*(i&31): what the forward
  //                 walk compares in a unitig starts behind its first k-mer (NODE_INLINE_FIRST)
  // bases from 94 on (and the first 30, for the left extension) come from the packed unitig buffer.
  std::vector<uint32_t> node_rec;    // 16 x u32 per node
  std::vector<uint32_t> node_ledge;  // 4 x u32 per node (left extension only)
  std::vector<uint64_t> unitig;      // 2-bit packed, base i at word i>>5, bits 62-2*(i&31)
  std::vector<uint32_t> col_off;     // CSR over static classes, n_colours+1
  std::vector<uint32_t> col_ids;
  // class descriptor, 16 bytes: {len | CLS_MASK_FLAG, base row, mask lo, mask hi}: bit i of the mask = row
  // base + i.  Library rows of one gene family are adjacent, so almost every class is local and the
  // intersection of visited colours is one 16-byte load per colour plus shift/AND.  Classes that span 64
  // rows or more keep len without the flag and are intersected through the CSR ids.
  std::vector<uint32_t> cls_desc;    // 4 x u32 per class
  std::vector<uint64_t> cls_bits;    // row bitmaps of the classes wider than the mask form (cls_desc words 1..3)
  bool all_classes_local = true;     // every static class has the mask form
  // ... and the classes of every connected component of the graph fit ONE 64-row window: the class descriptor inside a node
  // record is then written relative to its component's first row (same base word for every unitig a walk can hop to), and
  // the walk intersects without shifting (kernels.hip push_col)
  bool uniform_windows = false;
  // Stretch records (round 4, the fast walk of kernels.hip): what a walk compares in a unitig -- its bases from base 30
  // on -- cut into stretches of at most 32 bases, one 32-byte record per stretch, so that one step of the fast walk is
  // one record = two 16-byte gathers = one compare, whatever the unitig's length.  Built only for an index with
  // uniform_windows (class masks relative to the component's first row: a walk that follows edges needs no base row).
  //   u32[0]    = bases in this stretch (bits 0..5) | right-extension bits of the unitig (8..11) | SREC_LAST (12): the
  //               unitig ends with this stretch | SREC_MANY (13): more than two right edges (u32[6] is then the fork's place
  //               in srec_many, which holds its four neighbours) | class length (16..22)
  //   u32[1]    = colour (class id) of the unitig
  //   u32[2..3] = the stretch's bases, first base in the highest bit pair (bases 30 + 32 j .. of the unitig, j = record)
  //   u32[4..5] = class mask of the unitig (relative to its component's first row)
  //   u32[6..7] = (last stretch) record index of the right neighbour behind the lowest / the other extension base
  // The first 16 bytes are what a compare needs, the second 16 what the way out of the stretch needs.
  // srec_first[node] = the unitig's first record; srec_base[record] = first row of the unitig's component.
  std::vector<uint32_t> srec;        // 8 x u32 per record
  std::vector<uint32_t> srec_first;  // per node
  std::vector<uint32_t> srec_base;   // per record
  std::vector<uint32_t> srec_many;   // 4 x u32 per fork with more than two ways out: the neighbour's record per extension base
  bool all_wide_have_bitmaps = true; // every static class outside the mask form has a row bitmap in cls_bits
  uint32_t max_bitmap_words = 0;     // ... and the longest of those bitmaps, in 64-row words (sizes the device's row window)
  uint64_t n_kmers = 0, n_nodes = 0, n_colours = 0, unitig_bases = 0;
};

// seqs: concatenated ASCII, off[n+1].  Throws std::runtime_error on inconsistency.
void build_flat_index(const uint8_t *seqs, const uint64_t *off, uint32_t n_seqs, FlatIndex &out);
// The stretch records against the unitig records they were cut from: every unitig's records spell its bases from base 30 on
// (against the packed unitig buffer), carry its class mask, length and colour, end in its right neighbours' first records
// (forks with more than two in srec_many), and srec_first / srec_base name what they should.  Returns an empty string, or what
// is wrong first.  (CPU test of the builder: tests/test_host_cpu.py.)
std::string check_stretch_records(const FlatIndex &fi);

}  // namespace nimble
