// kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the nimble hot path.
//
// Integer / bit work throughout; no MFMA.  What is mirrored from the reference:
//   k_pack          DnaString::from_acgt_bytes (src/parse/fastq.rs:36), the read_key of
//                   src/align.rs:576-579 (R1 bases ++ R2 bases), the length and entropy prefilters of
//                   pseudoalign (src/align.rs:955-962) with utils::shannon_entropy (src/utils.rs:96-119)
//   k_align         Pseudoaligner::map_read_with_mismatch (external crate, call site src/align.rs:965),
//                   the rest of pseudoalign (src/align.rs:966-988) and filter_alignment_by_metrics
//                   (src/filter/align.rs:4-45)
//   k_intern_*      gives every intersected class a canonical id (content-addressed), the device form of
//                   "the equivalence class" the reference carries around as Vec<u32>
//   k_dedup/k_count score_map keyed by the read string (src/align.rs:496-505,685), require_valid_pair /
//                   filter_pair (src/align.rs:582-588,732-760) and the per-key `+= 1` of src/align.rs:245-249,
//                   grouped by (class R1, class R2)
//
// Layout rules used here: one lane per read(-pair); 64-wide waves; packed keys are stored word-major
// ([word][read]) so every per-read load/store of a wave is one contiguous 512-byte segment; ASCII input is
// staged through LDS with 16-byte coalesced loads; the walk keeps the lane's packed read in an LDS column
// (dynamic indexing without scratch) and only gathers 16-byte hash slots / node records / 8-byte unitig
// words from the (L2 / Infinity-Cache resident) index.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

#include "flat_index.h"
#include "kernels.h"

namespace nimble {

namespace {

#ifndef NIMBLE_TENT
#define NIMBLE_TENT 1  // 0 (experiments): the local re-seed of walk() compiled out
#endif
constexpr int PACK_BLOCK = 256;
#ifndef NIMBLE_ALIGN_BLOCK
#define NIMBLE_ALIGN_BLOCK 256
#endif
constexpr int ALIGN_BLOCK = NIMBLE_ALIGN_BLOCK;  // reads per tile (64 .. 512: the wave counters below hold 8 waves)
constexpr int ALIGN_GRID = 2048 * 256 / ALIGN_BLOCK;  // 256 CUs x 8 resident blocks; grid-stride over tiles of 256 reads
constexpr int LDS_COLS = 4;
constexpr int ALIGN_LDS_EXTRA = 16 + 64 + ALIGN_BLOCK * 2 + ALIGN_BLOCK * 8;  // tile slot, wave counts, perm, seeds
constexpr int ALIGN_LDS_META = ALIGN_BLOCK * 8;  // (fast walk only, behind them: lengths and verdicts per read of the tile)
constexpr double MIN_ENTROPY_SCORE = 1.75;  // src/align.rs:19

// NIMBLE_PROFILE_SECTIONS (debug builds only, tools/build_variant.sh prof -DNIMBLE_PROFILE_SECTIONS=1): wave clock cycles
// per section of k_align, summed over all waves into g_prof (read with nimble_debug_sections).
#ifndef NIMBLE_PROFILE_SECTIONS
#define NIMBLE_PROFILE_SECTIONS 0
#endif
// NIMBLE_PROFILE_SECTIONS=2: no clocks; WCOUNT(i) counts how often a WAVE executes the block it stands in (whatever the number
// of active lanes): wave-level trip counts of the divergent paths, to be multiplied with the block's static instruction count.
#if NIMBLE_PROFILE_SECTIONS == 2
__device__ unsigned long long g_prof[16];
#define WCOUNT(i) { if (__lane_id() == (unsigned)__ffsll((unsigned long long)__ballot(1)) - 1u) atomicAdd(&g_prof[i], 1ULL); }
#define PROF_DECL
#define PROF(i)
#define PROF_ARGS
#define PROF_PASS
#elif NIMBLE_PROFILE_SECTIONS
#define WCOUNT(i)
__device__ unsigned long long g_prof[16];
#define PROF_DECL unsigned long long prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_t = clock64();
#define PROF(i) { const unsigned long long prof_n = clock64(); prof_acc[i] += prof_n - prof_t; prof_t = prof_n; }
#define PROF_ARGS , unsigned long long *prof_acc, unsigned long long &prof_t
#define PROF_PASS , prof_acc, prof_t
#else
#define WCOUNT(i)
#define PROF_DECL
#define PROF(i)
#define PROF_ARGS
#define PROF_PASS
#endif

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if (lane >= (uint32_t)d) x += y;
  }
  total = __shfl(x, 63, 64);
  return x - v;
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// k_pack: ASCII -> 2-bit packed key, lengths, key hash, prefilter verdicts
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PACK_BLOCK) void k_pack(const uint8_t *__restrict__ r1, const uint64_t *__restrict__ off1,
                                                     const uint8_t *__restrict__ r2, const uint64_t *__restrict__ off2,
                                                     uint32_t fixed_len, uint32_t max_len, uint32_t rpb,
                                                     uint32_t tile_bytes, uint32_t min_len,
                                                     const double *__restrict__ plog, CallBuffers cb) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t tid = threadIdx.x;
  const uint64_t n = cb.n;
  const uint64_t r0 = (uint64_t)blockIdx.x * rpb;
  if (r0 >= n) return;
  const uint32_t cnt = (uint32_t)((n - r0 < rpb) ? (n - r0) : rpb);
  const int nm = cb.paired ? 2 : 1;
  uint64_t seg_start[2] = {0, 0};
  uint32_t shift[2] = {0, 0};
  for (int m = 0; m < nm; ++m) {
    const uint8_t *src = m ? r2 : r1;
    const uint64_t *off = m ? off2 : off1;
    uint64_t s = off ? off[r0] : r0 * fixed_len;
    uint64_t e = off ? off[r0 + cnt] : (r0 + cnt) * fixed_len;
    // offsets handed over in device memory were never seen by the host: a span that does not fit the tile (non-monotone
    // offsets, a read longer than max_len) is cut here and reported through the call's error word
    bool bad_span = false;
    if (e < s || e - s > (uint64_t)cnt * max_len) {
      e = s;
      bad_span = true;
      if (tid == 0) atomicOr((unsigned long long *)&cb.state[14], 1ULL);
    }
    uintptr_t a = (uintptr_t)(src + s);
    uintptr_t a0 = a & ~(uintptr_t)15;
    shift[m] = (uint32_t)(a - a0);
    seg_start[m] = s;
    // (a span that made no sense is not touched at all: its start is as untrustworthy as its length)
    uint32_t nvec = bad_span ? 0u : (uint32_t)((e - s + shift[m] + 15) >> 4);
    uint4 *dst = reinterpret_cast<uint4 *>(lds + (size_t)m * tile_bytes);
    const uint4 *g = reinterpret_cast<const uint4 *>(a0);
    // 16 B / lane, coalesced, written to LDS by the load itself (global_load_lds_dwordx4: wave-uniform LDS base +
    // lane * 16): every load of the tile is in flight before the single wait in front of the barrier
    for (uint32_t base = 0; base < nvec; base += PACK_BLOCK) {
      const uint32_t i = base + tid;
      if (i < nvec)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + i),
                                         (__attribute__((address_space(3))) void *)(dst + (base + (tid & ~63u))), 16, 0, 0);
    }
  }
  __syncthreads();
  if (tid >= cnt) return;
  const uint64_t r = r0 + tid;
  uint32_t L[2] = {0, 0};
  const uint8_t *p[2] = {nullptr, nullptr};
  for (int m = 0; m < nm; ++m) {
    const uint64_t *off = m ? off2 : off1;
    uint64_t s = off ? off[r] : r * fixed_len;
    uint64_t e = off ? off[r + 1] : s + fixed_len;
    // the read must lie inside what the block staged: [seg_start, seg_start + cnt * max_len)
    if (e < s || e - s > max_len || s < seg_start[m] || e - seg_start[m] > (uint64_t)cnt * max_len) {
      atomicOr((unsigned long long *)&cb.state[14], 1ULL);
      s = seg_start[m];
      e = s;  // an empty read: ShortRead, nothing is read beyond the tile
    }
    L[m] = (uint32_t)(e - s);
    p[m] = lds + (size_t)m * tile_bytes + shift[m] + (uint32_t)(s - seg_start[m]);
  }
  const uint32_t total = L[0] + L[1];
  uint64_t acc = 0, h = 0x8F1BBCDCCA62C1D6ULL ^ (uint64_t)total;
  uint32_t nb = 0, w = 0;            // bases pending in acc (< 32), words emitted
  uint32_t nT = 0, nG = 0, nC = 0;   // running base counts over everything packed so far
  uint32_t pT = 0, pG = 0, pC = 0;   // the same at the end of the previous mate
  auto count_word = [&](uint64_t word) {
    const uint64_t hi = (word >> 1) & 0x5555555555555555ULL, lo = word & 0x5555555555555555ULL;
    nT += (uint32_t)__popcll(hi & lo);
    nG += (uint32_t)__popcll(hi & ~lo);
    nC += (uint32_t)__popcll(~hi & lo);
  };
  auto emit = [&](uint64_t word) {
    cb.keys[(uint64_t)w * cb.key_stride + r] = word;
    h = (h ^ word) * 0xff51afd7ed558ccdULL;
    h ^= h >> 32;
    ++w;
    count_word(word);
  };
  // append k (1..4) bases given as 2k right-aligned bits
  auto append = [&](uint32_t bits, uint32_t k) {
    if (nb + k < 32u) {
      acc = (acc << (2u * k)) | bits;
      nb += k;
    } else {
      const uint32_t first = 32u - nb, rest = k - first;  // first >= 1
      emit((acc << (2u * first)) | (bits >> (2u * rest)));
      acc = bits & ((1u << (2u * rest)) - 1u);
      nb = rest;
    }
  };
  for (int m = 0; m < nm; ++m) {
    // 4 bases per step: aligned LDS dwords funnel-shifted to the read's byte alignment, SWAR conversion
    const uint32_t a = (uint32_t)(p[m] - lds);
    const uint32_t mis = a & 3u;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(lds) + (a >> 2);
    const uint32_t len = L[m];
    uint32_t prev = q[0];
    uint32_t i = 0;
    // 4 ASCII bytes (first base in the low byte) -> 8 bits of 2-bit codes, first base highest
    auto conv4 = [&](uint32_t raw) -> uint32_t {
      const uint32_t x = raw | 0x20202020u;                 // lower-case
      uint32_t code = (x >> 1) & 0x03030303u;               // a->0 c->1 g->3 t->2
      code ^= (code >> 1) & 0x01010101u;                    // a->0 c->1 g->2 t->3
      const uint32_t letter = __builtin_amdgcn_perm(0u, 0x74676361u, code);  // code -> 'a','c','g','t'
      const uint32_t d = letter ^ x;                        // zero byte <=> a valid base
      if (d) {                                              // rare: something that is not A/C/G/T encodes as 'A' (0)
        const uint32_t zb = ~(((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d | 0x7F7F7F7Fu);  // 0x80 where valid
        code &= (zb >> 7) * 0xFFu;
      }
      return (code * 0x40100401u) >> 24;
    };
    auto fetch4 = [&]() -> uint32_t {                       // next 4 bytes of the read
      const uint32_t next = q[++i];
      const uint32_t raw = __builtin_amdgcn_alignbyte(next, prev, mis);
      prev = next;
      return raw;
    };
    uint32_t done = 0;
    // whole 32-base groups: the word is assembled with constant shifts and meets the pending bases once
    for (; done + 32u <= len; done += 32u) {
      uint32_t hi = 0, lo = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) hi = (hi << 8) | conv4(fetch4());
#pragma unroll
      for (int j = 0; j < 4; ++j) lo = (lo << 8) | conv4(fetch4());
      const uint64_t w32 = ((uint64_t)hi << 32) | lo;
      if (nb == 0) {
        emit(w32);
      } else {
        emit((acc << (64u - 2u * nb)) | (w32 >> (2u * nb)));
        acc = w32 & ((1ULL << (2u * nb)) - 1ULL);
      }
    }
    for (; done < len; done += 4) {
      const uint32_t packed = conv4(fetch4());
      const uint32_t k = len - done < 4u ? len - done : 4u;
      append(packed >> (2u * (4u - k)), k);
    }
    cb.len[m][r] = len;
    // base counts of this mate: totals so far (emitted words + the pending bases) minus the previous mate's
    const uint64_t phi = (acc >> 1) & 0x5555555555555555ULL, plo = acc & 0x5555555555555555ULL;
    const uint32_t tT = nT + (uint32_t)__popcll(phi & plo), tG = nG + (uint32_t)__popcll(phi & ~plo);
    const uint32_t tC = nC + (uint32_t)__popcll(~phi & plo & ((nb ? (1ULL << (2u * nb)) : 1ULL) - 1ULL));
    uint32_t cT = tT - pT, cG = tG - pG, cC = tC - pC;
    pT = tT;
    pG = tG;
    pC = tC;
    uint32_t alen = len;
    if (cb.alen[m] != cb.len[m]) {
      // quality-trimmed call: the prefilters see the first alen bases only (pseudoalign gets the trimmed read,
      // align.rs:519-523); recount over that prefix
      alen = cb.alen[m][r];
      alen = alen < len ? alen : len;
      cb.alen[m][r] = alen;
      cT = cG = cC = 0;
      uint32_t pv = q[0], ii = 0;
      for (uint32_t d0 = 0; d0 < alen; d0 += 4) {
        const uint32_t nx = q[++ii];
        const uint32_t raw = __builtin_amdgcn_alignbyte(nx, pv, mis);
        pv = nx;
        uint32_t c8 = conv4(raw);                       // first base in the highest pair
        const uint32_t k = alen - d0 < 4u ? alen - d0 : 4u;
        c8 >>= 2u * (4u - k);                           // keep the first k bases
        const uint32_t hi = (c8 >> 1) & 0x55u, lo = c8 & 0x55u;
        cT += (uint32_t)__popc(hi & lo);
        cG += (uint32_t)__popc(hi & ~lo);
        cC += (uint32_t)__popc(~hi & lo & ((1u << (2u * k)) - 1u));
      }
    }
    const uint32_t cA = alen - cT - cG - cC;
    uint8_t verdict = (uint8_t)R_TODO;
    if (cb.skip[m] && cb.skip[m][r]) {
      verdict = NIMBLE_R_SKIPPED_ALIGN_DUE_TO_UNPAIRED_DUMMY;
    } else if (alen < min_len) {
      verdict = NIMBLE_R_SHORT_READ;
    } else {
      // shannon_entropy: sum f*log2(f) over A, T, C, G in that order, terms from the host-built table
      const double *row = plog + ((uint64_t)alen * (alen + 1)) / 2;
      double e = 0.0;
      if (cA) e += row[cA];
      if (cT) e += row[cT];
      if (cC) e += row[cC];
      if (cG) e += row[cG];
      if (-e < MIN_ENTROPY_SCORE) verdict = NIMBLE_R_HIGH_ENTROPY;
    }
    cb.pre[m][r] = verdict;
  }
  if (nb) {
    // the last word is left-aligned; its padding is zero = 'A' codes, excluded from the counts above
    const uint64_t word = acc << (64u - 2u * nb);
    cb.keys[(uint64_t)w * cb.key_stride + r] = word;
    h = (h ^ word) * 0xff51afd7ed558ccdULL;
    h ^= h >> 32;
  }
  // words past the end of the key stay defined (the exchange copies whole records)
  for (uint32_t z = w + (nb ? 1u : 0u); z < cb.key_words; ++z) cb.keys[(uint64_t)z * cb.key_stride + r] = 0ULL;
  cb.key_hash[r] = mix64(h);
}

// ---------------------------------------------------------------------------------------------
// k_pack_words: the same outputs as k_pack from reads the host has already packed (nimble_stream_append_packed): mate m of
// read r is `len[m][r]` bases in words w[m] + r * stride[m], 32 bases a word, first base in the highest bit pair, zero
// behind the last base (a byte that is no A/C/G/T was packed as A, as k_pack does).  The key is mate 0's bases followed
// at once by mate 1's; key words, key hash, lengths and prefilter verdicts come out exactly as k_pack writes them.
// A thread a read: 40 bytes in for a 150-base read, the reads of a wave are neighbours in memory.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_words(const uint64_t *__restrict__ w1, const uint32_t *__restrict__ len1,
                                                    uint32_t stride1, const uint64_t *__restrict__ w2,
                                                    const uint32_t *__restrict__ len2, uint32_t stride2, uint32_t max_len,
                                                    uint32_t min_len, const double *__restrict__ plog, CallBuffers cb) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= cb.n) return;
  const int nm = cb.paired ? 2 : 1;
  uint32_t L[2] = {0, 0};
  for (int m = 0; m < nm; ++m) {
    uint32_t l = (m ? len2 : len1)[r];
    if (l > max_len || l > 32u * (m ? stride2 : stride1)) {  // lengths handed over in device memory: cut, and say so
      atomicOr((unsigned long long *)&cb.state[14], 1ULL);
      l = 0;
    }
    L[m] = l;
  }
  const uint32_t total = L[0] + L[1];
  uint64_t acc = 0, h = 0x8F1BBCDCCA62C1D6ULL ^ (uint64_t)total;
  uint32_t nb = 0, w = 0;  // bases pending in acc (right-aligned, < 32), key words written
  auto emit = [&](uint64_t word) {
    cb.keys[(uint64_t)w * cb.key_stride + r] = word;
    h = (h ^ word) * 0xff51afd7ed558ccdULL;
    h ^= h >> 32;
    ++w;
  };
  for (int m = 0; m < nm; ++m) {
    const uint64_t *src = (m ? w2 : w1) + r * (uint64_t)(m ? stride2 : stride1);
    const uint32_t len = L[m];
    uint32_t cT = 0, cG = 0, cC = 0;
    for (uint32_t done = 0; done < len; done += 32u) {
      const uint32_t k = len - done < 32u ? len - done : 32u;  // bases in this word
      uint64_t word = src[done >> 5];
      if (k < 32u) word &= ~0ULL << (64u - 2u * k);             // (whatever the host left behind the last base)
      const uint64_t hi = (word >> 1) & 0x5555555555555555ULL, lo = word & 0x5555555555555555ULL;
      cT += (uint32_t)__popcll(hi & lo);
      cG += (uint32_t)__popcll(hi & ~lo);
      cC += (uint32_t)__popcll(~hi & lo);                       // (padding is 00: no C there)
      // append k bases (left-aligned in `word`) behind the nb pending ones
      if (nb == 0) {
        if (k == 32u) emit(word);
        else { acc = word >> (64u - 2u * k); nb = k; }
      } else if (nb + k < 32u) {
        acc = (acc << (2u * k)) | (word >> (64u - 2u * k));
        nb += k;
      } else {
        const uint32_t first = 32u - nb, rest = k - first;      // first >= 1 bases complete the pending word
        emit((acc << (2u * first)) | (word >> (64u - 2u * first)));
        acc = rest ? (word << (2u * first)) >> (64u - 2u * rest) : 0ULL;
        nb = rest;
      }
    }
    cb.len[m][r] = len;
    const uint32_t cA = len - cT - cG - cC;
    uint8_t verdict = (uint8_t)R_TODO;
    if (cb.skip[m] && cb.skip[m][r]) {
      verdict = NIMBLE_R_SKIPPED_ALIGN_DUE_TO_UNPAIRED_DUMMY;
    } else if (len < min_len) {
      verdict = NIMBLE_R_SHORT_READ;
    } else {
      const double *row = plog + ((uint64_t)len * (len + 1)) / 2;  // shannon_entropy, terms in the order A, T, C, G
      double e = 0.0;
      if (cA) e += row[cA];
      if (cT) e += row[cT];
      if (cC) e += row[cC];
      if (cG) e += row[cG];
      if (-e < MIN_ENTROPY_SCORE) verdict = NIMBLE_R_HIGH_ENTROPY;
    }
    cb.pre[m][r] = verdict;
  }
  if (nb) {
    const uint64_t word = acc << (64u - 2u * nb);
    cb.keys[(uint64_t)w * cb.key_stride + r] = word;
    h = (h ^ word) * 0xff51afd7ed558ccdULL;
    h ^= h >> 32;
  }
  for (uint32_t z = w + (nb ? 1u : 0u); z < cb.key_words; ++z) cb.keys[(uint64_t)z * cb.key_stride + r] = 0ULL;
  cb.key_hash[r] = mix64(h);
}

// ---------------------------------------------------------------------------------------------
// k_align helpers
// ---------------------------------------------------------------------------------------------

// A call may take its reads straight from exchange records (cb.rec: rows of rec_words = key_words + 2 u64 =
// [key words..., key hash, len0 | len1 << 16 | pre0 << 32 | pre1 << 40]) instead of the packed arrays.
__device__ __forceinline__ uint64_t rd_key(const CallBuffers &cb, uint32_t w, uint64_t r) {
  return cb.rec ? cb.rec[r * cb.rec_words + w] : cb.keys[(uint64_t)w * cb.key_stride + r];
}
__device__ __forceinline__ uint64_t rd_meta(const CallBuffers &cb, uint64_t r) {
  return cb.rec[r * cb.rec_words + cb.key_words + 1];
}
__device__ __forceinline__ uint32_t rd_len(const CallBuffers &cb, int m, uint64_t r) {
  return cb.rec ? (uint32_t)((rd_meta(cb, r) >> (16 * m)) & 0xFFFF) : cb.len[m][r];
}
__device__ __forceinline__ uint32_t rd_alen(const CallBuffers &cb, int m, uint64_t r) {
  return cb.rec ? (uint32_t)((rd_meta(cb, r) >> (16 * m)) & 0xFFFF) : cb.alen[m][r];
}
__device__ __forceinline__ uint32_t rd_pre(const CallBuffers &cb, int m, uint64_t r) {
  return cb.rec ? (uint32_t)((rd_meta(cb, r) >> (32 + 8 * m)) & 0xFF) : cb.pre[m][r];
}
__device__ __forceinline__ uint64_t rd_hash(const CallBuffers &cb, uint64_t r) {
  return cb.rec ? cb.rec[r * cb.rec_words + cb.key_words] : cb.key_hash[r];
}

struct Lane {
  const uint64_t *rd;  // LDS column holding the packed key (stride ALIGN_BLOCK words)
  uint32_t *lc;        // LDS column of visited colours (stride ALIGN_BLOCK)
  uint32_t *ws;        // global spill column (stride ws_lanes)
  uint32_t ws_lanes, ws_rows;
  uint32_t n_cols, last_col, walk_nodes;
  // running intersection of the visited colours (mask form): window base, surviving rows, smallest class
  uint64_t acc;
  uint32_t fbase, min_len, min_col;
  uint32_t last_rec;  // (fast walk) a stretch record of the walk: srec_base[last_rec] is the walk's fbase, fetched only if needed
  bool all_mask;
  bool keep_list;  // some class of the index is not local: keep the visited colours for the general path
  // ... unless the first visited class spans at most WIDE_ROWS rows: then the intersection is folded during the walk into
  // a register window of that many rows starting at the class's first row (allele families of up to ~100 alleles: every
  // class of a gene lies inside), and neither a colour list nor a pass over it is needed
  bool window_ok;
  bool use_window;  // the index allows it (every wide class has a row bitmap)
  bool uniform;     // the record masks are relative to their component's first row: equal bases, or nothing in common
  uint32_t wbase;
  uint64_t wacc[4];
  // an index whose class bitmaps are longer than that (allele families of several hundred rows) keeps the window in an LDS
  // column of the lane instead: wcap words (stride ALIGN_BLOCK), wn of them in use for the read at hand
  uint64_t *wl;
  uint32_t wn, wcap;
  const uint64_t *cls_bits;
  uint32_t probes, nodes;
  uint64_t entries;
  int want_counters;
  uint32_t overflow;
};

// one unitig, all from one 64-byte line: header, the descriptor of its colour class, right edges and the
// first 64 bases (layout: flat_index.h)
struct NodeRec {
  uint4 q0;  // {len | exts << 24, colour, seq_start, class len | flag}
  uint4 q1;  // {class base, mask lo, mask hi, 0}
  uint4 re;  // right-edge targets
  uint4 sq;  // bases 0..63
};
__device__ __forceinline__ NodeRec load_node(const DevIndex &ix, uint32_t node) {
  const uint4 *p = ix.node_rec + (size_t)node * 4;
  NodeRec r;
  r.q0 = p[0];
  r.q1 = p[1];
  r.re = p[2];
  r.sq = p[3];
  return r;
}
__device__ __forceinline__ uint32_t nr_len(const NodeRec &r) { return r.q0.x & 0xFFFFFFu; }
__device__ __forceinline__ uint32_t nr_exts(const NodeRec &r) { return r.q0.x >> 24; }
__device__ __forceinline__ uint4 nr_desc(const NodeRec &r) { return make_uint4(r.q0.w, r.q1.x, r.q1.y, r.q1.z); }
__device__ __forceinline__ uint64_t u64of(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

// One-touch streams (packed keys in, per-read results out) use the non-temporal hint so that they do not push
// the re-used node records out of the XCD's L2.  (Measured: the hint on dictionary / filter gathers is a
// loss -- those live in the Infinity Cache and want the default policy.)
#ifdef NIMBLE_NO_NT
__device__ __forceinline__ uint64_t ld_stream(const uint64_t *p) { return *p; }
template <class T> __device__ __forceinline__ void st_stream(T *p, T v) { *p = v; }
#else
__device__ __forceinline__ uint64_t ld_stream(const uint64_t *p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ void st_stream(T *p, T v) { __builtin_nontemporal_store(v, p); }
#endif

// 16-byte gathers from the dictionary and the presence filter (tables beyond L2).  NIMBLE_GATHER_POLICY (experiments
// only) issues them with explicit cache-policy bits: 1 = nt, 2 = sc1, 3 = sc0 sc1, 4 = sc0 sc1 nt, 5 = sc1 nt.
#ifndef NIMBLE_GATHER_POLICY
#define NIMBLE_GATHER_POLICY 0
#endif
// a 16-byte load the compiler keeps whole (it takes loads of HIP's uint4 apart into dwords and puts them together again as it
// likes: a stretch record's two halves became three requests -- 8 + 16 + 8 bytes -- on one line, and the rate of a walk is
// requests per line, tools/probes/chain.hip)
__device__ __forceinline__ uint4 ld16(const uint4 *p) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = *reinterpret_cast<const u32x4_t *>(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 ld_gather(const uint4 *p) {
#if defined(NIMBLE_GATHER_NT)
  // (experiments, round 4: the non-temporal hint without a wait behind it -- the compiler's own load, scheduled as any other)
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#elif NIMBLE_GATHER_POLICY == 0
  return *p;
#else
  uint4 v;
#if NIMBLE_GATHER_POLICY == 1
  asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
#elif NIMBLE_GATHER_POLICY == 2
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
#elif NIMBLE_GATHER_POLICY == 3
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
#elif NIMBLE_GATHER_POLICY == 4
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
#else
  asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
#endif
  return v;
#endif
}

// nb (1..32) bases of the lane's key starting at base `pos`, right-aligned
// (branch-free: a shift by 64 - s is split into 1 + (63 - s) so that s = 0 needs no special case -- the branch the
// compiler made of `s ? ... : hi` sat in the innermost loop of the walk; the column has a zero word behind the key)
__device__ __forceinline__ uint64_t funnel(uint64_t hi, uint64_t lo, uint32_t s) {  // s = 0, 2, .. 62
  return (hi << s) | ((lo >> 1) >> (63u - s));
}
__device__ __forceinline__ uint64_t lds_bits(const uint64_t *rd, uint32_t pos, uint32_t nb) {
  const uint32_t w = pos >> 5, s = (pos & 31u) * 2u;
  return funnel(rd[w * ALIGN_BLOCK], rd[(w + 1) * ALIGN_BLOCK], s) >> (64u - 2u * nb);
}
// the 32 bases from `pos` on, left-aligned
__device__ __forceinline__ uint64_t lds_window(const uint64_t *rd, uint32_t pos) {
  const uint32_t w = pos >> 5, s = (pos & 31u) * 2u;
  return funnel(rd[w * ALIGN_BLOCK], rd[(w + 1) * ALIGN_BLOCK], s);
}
__device__ __forceinline__ uint32_t lds_base(const uint64_t *rd, uint32_t pos) {
  return (uint32_t)(rd[(pos >> 5) * ALIGN_BLOCK] >> (62u - 2u * (pos & 31u))) & 3u;
}
__device__ __forceinline__ uint64_t g_bits(const uint64_t *__restrict__ u, uint64_t pos, uint32_t nb) {
  uint64_t w = pos >> 5;
  uint32_t s = (uint32_t)(pos & 31u) * 2u;
  return funnel(u[w], u[w + 1], s) >> (64u - 2u * nb);  // (the unitig buffer ends in spare words)
}
// nb bases of a unitig at node-relative position pos: from the record while inside its 64 inline bases, which are the
// bases [30, 94) -- the forward walk never looks at a unitig's first k-mer
__device__ __forceinline__ uint64_t node_bits(const NodeRec &r, const uint64_t *__restrict__ unitig, uint32_t pos,
                                              uint32_t nb) {
  const uint32_t q = pos - NODE_INLINE_FIRST;  // (wraps for pos < 30: the test below fails)
  if (q + nb <= NODE_INLINE_BASES && pos >= NODE_INLINE_FIRST) {
    const uint64_t w0 = u64of(r.sq.x, r.sq.y), w1 = u64of(r.sq.z, r.sq.w);
    const uint32_t s = (q & 31u) * 2u;
    const uint64_t hi = q < 32u ? w0 : w1;
    const uint64_t lo = q < 32u ? w1 : 0ULL;
    return funnel(hi, lo, s) >> (64u - 2u * nb);
  }
  return g_bits(unitig, (uint64_t)r.q0.z + pos, nb);
}
__device__ __forceinline__ uint32_t sel4(const uint4 &v, uint32_t b) {
  const uint32_t lo = (b & 1u) ? v.y : v.x, hi = (b & 1u) ? v.w : v.z;
  return (b & 2u) ? hi : lo;
}

// finish a probe whose first slot neither matched nor was empty (linear probing, rare)
// returns node << 32 | offset, or HT_EMPTY when the k-mer is absent (values only: no scratch traffic)
__device__ __noinline__ uint64_t ht_resolve_slow(const uint4 *__restrict__ ht, uint64_t mask, uint64_t km,
                                                 uint64_t h) {
  for (;;) {
    WCOUNT(15)
    h = (h + 1) & mask;
    uint4 s = ht[h];
    uint64_t key = u64of(s.x, s.y);
    if (key == km) return u64of(s.z, s.w);
    if (key == HT_EMPTY) return HT_EMPTY;
  }
}

// one direct dictionary probe at mate position pos
__device__ __forceinline__ bool probe_direct(const DevIndex &ix, Lane &ln, uint32_t base0, uint32_t pos, uint32_t &node,
                                             uint32_t &off, bool count = true) {
  const uint64_t km = lds_bits(ln.rd, base0 + pos, KMER);
  const uint64_t h = kmer_slot(km, ix.ht_log2);
  const uint4 s = ld_gather(ix.ht + h);
  const uint64_t key = u64of(s.x, s.y);
  if (count) ln.probes++;
  if (key == km) { off = s.z; node = s.w; return true; }
  if (key != HT_EMPTY) {
    const uint64_t v = ht_resolve_slow(ix.ht, ix.ht_mask, km, h);
    if (v != HT_EMPTY) { off = (uint32_t)v; node = (uint32_t)(v >> 32); return true; }
  }
  return false;
}

// The filter's answer for one round of the seed scan at kmer_pos (<= last_kmer_pos): bit i = the k-mer at
// kmer_pos + 3i may be in the dictionary.  SCAN_ROUND positions at stride 3 are answered by ONE 16-byte line of the
// presence filter, selected by the 12 bases all 7 k-mers share; positions beyond the last k-mer are masked off.
// FIRST: the scan for the first seed of a mate (most of those mates are not from the library at all): the L2-resident
// first level is asked before the filter line is fetched; a re-seed inside a walk goes to the line at once.
__device__ __forceinline__ uint32_t round_maybe(const DevIndex &ix, const uint64_t *rd, uint32_t base0, uint32_t kmer_pos,
                                                uint32_t last_kmer_pos, bool FIRST = false) {
  uint32_t maybe = 0;
  const uint32_t span = KMER + 3u * (SCAN_ROUND - 1);            // 48 bases
  const uint32_t avail = (last_kmer_pos + KMER) - kmer_pos;      // bases of the mate from kmer_pos on
  uint64_t km = lds_bits(rd, base0 + kmer_pos, KMER);
  if (FIRST && ix.l1) {
    const uint32_t sh = (uint32_t)km & ((1u << (2u * SCAN_SHARED)) - 1u);
    if (!((ix.l1[sh >> 5] >> (sh & 31u)) & 1u)) return 0u;
  }
  const uint32_t extra = avail > KMER ? (avail - KMER < span - KMER ? avail - KMER : span - KMER) : 0u;
  WCOUNT(9)
  const uint64_t tail = extra ? (lds_bits(rd, base0 + kmer_pos + KMER, extra) << (2u * (span - KMER - extra))) : 0ULL;
  const uint4 line = ld_gather(ix.bitmap + round_line(km & ((1ULL << (2u * SCAN_SHARED)) - 1ULL), ix.bm_lines_log2));
  const uint64_t half0 = u64of(line.x, line.y), half1 = u64of(line.z, line.w);
#pragma unroll
  for (int i = 0; i < (int)SCAN_ROUND; ++i) {
    const uint32_t bb = round_bits(km);
    const uint64_t hw = (bb >> 12) & 1u ? half1 : half0;
    maybe |= (uint32_t)((hw >> (bb & 63u)) & (hw >> ((bb >> 6) & 63u)) & 1ULL) << i;
    if (i + 1 < (int)SCAN_ROUND) {
      const uint32_t sh = 2u * (span - KMER) - 6u * (uint32_t)(i + 1);
      km = ((km << 6) | ((tail >> sh) & 63ULL)) & KMER_MASK;
    }
  }
  const uint32_t valid = last_kmer_pos - kmer_pos;  // positions kmer_pos + 3i <= last  <=>  3i <= valid
  const uint32_t nvalid = valid / 3u + 1u < SCAN_ROUND ? valid / 3u + 1u : SCAN_ROUND;
  return maybe & ((1u << nvalid) - 1u);
}

// One round of the seed scan at kmer_pos (<= last_kmer_pos): candidates of round_maybe are verified in the
// dictionary in read order.  Found: kmer_pos / node / off are set.  Not found: kmer_pos moves past the round.
__device__ __forceinline__ bool scan_round(const DevIndex &ix, Lane &ln, uint32_t base0, uint32_t &kmer_pos,
                                           uint32_t last_kmer_pos, uint32_t &node, uint32_t &off, bool first = false) {
  uint32_t maybe = round_maybe(ix, ln.rd, base0, kmer_pos, last_kmer_pos, first);
  WCOUNT(8)
  const uint32_t valid = last_kmer_pos - kmer_pos;
  const uint32_t nvalid = valid / 3u + 1u < SCAN_ROUND ? valid / 3u + 1u : SCAN_ROUND;
  uint32_t examined = nvalid;
  bool found = false;
  while (maybe) {  // candidates in read order (filter false positives or a real seed); usually none
    WCOUNT(10)
    const uint32_t i = (uint32_t)__ffs((int)maybe) - 1u;
    maybe &= maybe - 1u;
    const uint32_t p = kmer_pos + 3u * i;
    const uint64_t km = lds_bits(ln.rd, base0 + p, KMER);
    const uint64_t h = kmer_slot(km, ix.ht_log2);
    const uint4 sl = ld_gather(ix.ht + h);
    const uint64_t key = u64of(sl.x, sl.y);
    uint64_t v = u64of(sl.z, sl.w);
    bool hit = key == km;
    if (!hit && key != HT_EMPTY) {
      v = ht_resolve_slow(ix.ht, ix.ht_mask, km, h);
      hit = v != HT_EMPTY;
    }
    if (hit) {
      found = true;
      examined = i + 1u;
      kmer_pos = p;
      off = (uint32_t)v;
      node = (uint32_t)(v >> 32);
      maybe = 0;
    }
  }
  ln.probes += examined;  // the reference examines positions one by one up to the first hit
  if (found) return true;
  kmer_pos += 3u * SCAN_ROUND;
  return false;
}

// seed search with stride 3 from kmer_pos (positions relative to the mate).  The first probe goes alone
// (it hits for most on-target reads); after a miss PROBE_BATCH independent probes are kept in flight.
__device__ __forceinline__ bool find_match(const DevIndex &ix, Lane &ln, uint32_t base0, uint32_t &kmer_pos,
                                           uint32_t last_kmer_pos, uint32_t &node, uint32_t &off,
                                           bool skip_direct = false, bool first = false) {
  if (kmer_pos > last_kmer_pos) return false;
  WCOUNT(7)
  if (!skip_direct) {
    const uint64_t km = lds_bits(ln.rd, base0 + kmer_pos, KMER);
    const uint64_t h = kmer_slot(km, ix.ht_log2);
    const uint4 s = ld_gather(ix.ht + h);
    const uint64_t key = u64of(s.x, s.y);
    ln.probes++;
    if (key == km) { off = s.z; node = s.w; return true; }
    if (key != HT_EMPTY) {
      const uint64_t v = ht_resolve_slow(ix.ht, ix.ht_mask, km, h);
      if (v != HT_EMPTY) { off = (uint32_t)v; node = (uint32_t)(v >> 32); return true; }
    }
    kmer_pos += 3;
  }
  if (first && ix.l1) {
    // The first seed of a mate (most of those mates are not from the library at all).  The L2-resident first level clears
    // 93 % of a foreign read's rounds, but a wave of 64 such reads has a lane that passes it in practically every round, and
    // the wave then runs the filter-line block (and a candidate probe behind it) round after round for four or five lanes:
    // six times per wave on the bench reads (profiles/r04_experiments.txt 13).  So: the first level for ALL rounds first
    // (independent loads, every lane busy), then each lane goes through the rounds that passed -- the wave runs the heavy
    // block as often as its worst lane has rounds left (two or three times), not once per round.
    // The reference examines positions one by one up to the first hit: the count follows from the hit's position.
    const uint32_t start = kmer_pos;
    bool found = false;
    for (uint32_t chunk = start; chunk <= last_kmer_pos && !found; chunk += 3u * SCAN_ROUND * 32u) {
      uint32_t pend = 0, nr = 0;
      for (uint32_t kp = chunk; kp <= last_kmer_pos && nr < 32u; kp += 3u * SCAN_ROUND, ++nr) {
        const uint32_t sh = (uint32_t)lds_bits(ln.rd, base0 + kp + (KMER - SCAN_SHARED), SCAN_SHARED);  // (the k-mer's last 12 bases)
        pend |= ((ix.l1[sh >> 5] >> (sh & 31u)) & 1u) << nr;
      }
      const uint32_t counted = ln.probes;
      while (pend) {
        const uint32_t r = (uint32_t)__ffs((int)pend) - 1u;
        pend &= pend - 1u;
        uint32_t kp = chunk + 3u * SCAN_ROUND * r;
        if (scan_round(ix, ln, base0, kp, last_kmer_pos, node, off, false)) {
          found = true;
          kmer_pos = kp;
          pend = 0;
        }
      }
      ln.probes = counted;  // (scan_round counted per round; in closed form below)
    }
    ln.probes += ((found ? kmer_pos : last_kmer_pos) - start) / 3u + 1u;
    return found;
  }
  while (kmer_pos <= last_kmer_pos)
    if (scan_round(ix, ln, base0, kmer_pos, last_kmer_pos, node, off, first)) return true;
  return false;
}

__device__ __forceinline__ uint32_t desc_len(const uint4 &d) { return d.x & ~CLS_MASK_FLAG; }
__device__ __forceinline__ bool desc_is_mask(const uint4 &d) { return (d.x & CLS_MASK_FLAG) != 0; }
__device__ __forceinline__ uint64_t desc_mask(const uint4 &d) { return u64of(d.z, d.w); }

// mask of class d expressed in the window starting at row `base`
__device__ __forceinline__ uint64_t mask_in_window(const uint4 &d, uint32_t base) {
  const uint64_t m = desc_mask(d);
  const int32_t delta = (int32_t)(d.y - base);
  const uint32_t ad = (uint32_t)(delta < 0 ? -delta : delta);
  const uint64_t v = delta < 0 ? (m >> (ad & 63u)) : (m << (ad & 63u));  // selects, no branch: this sits in the walk loop
  return ad < 64u ? v : 0ULL;
}

constexpr uint32_t WIDE_ROWS = 256;
// 64 rows of class d starting at row `row0`, as a bit mask.  Mask-form classes carry their bits in the descriptor;
// a wider static class has a span bitmap in ix.cls_bits (descriptor: y = first row of the bitmap, a multiple of 64 at or
// below the class's first row; z = rows spanned from there; w = word offset + 1), built with the index.
__device__ __forceinline__ uint64_t class_rows64(const uint4 &d, const uint64_t *__restrict__ cls_bits, uint32_t row0) {
  if (desc_is_mask(d)) return mask_in_window(d, row0);
  if (row0 + 64u <= d.y || row0 >= d.y + d.z) return 0ULL;
  const uint64_t *__restrict__ w = cls_bits + (d.w - 1u);
  const uint32_t n_words = (d.z + 63u) >> 6;
  if (row0 < d.y) return w[0] << (d.y - row0);  // the window starts before the class does (delta < 64 here)
  const uint32_t off = row0 - d.y, q = off >> 6, sh = off & 63u;
  uint64_t v = w[q] >> sh;
  if (sh && q + 1u < n_words) v |= w[q + 1u] << (64u - sh);
  return v;
}

// a visited node: counters, running mask intersection and, only when the index has non-local classes, the
// colour list for the general path
__device__ __forceinline__ void push_col(Lane &ln, uint32_t colour, const uint4 &desc) {
  ln.nodes++;
  if (ln.want_counters) ln.entries += desc_len(desc);
  const uint32_t l = desc_len(desc);
  if (ln.walk_nodes == 0) {
    ln.fbase = desc.y;
    ln.acc = ~0ULL;
    ln.all_mask = true;
    ln.min_len = l;
    ln.min_col = colour;
  } else if (l < ln.min_len) {
    ln.min_len = l;
    ln.min_col = colour;
  }
  const bool first_node = ln.walk_nodes == 0;
  ln.walk_nodes++;
  // idempotent: repeated colours cost nothing.  Component-relative masks need no shift: a unitig of another component (a
  // re-seed that landed in another gene family) shares no row with what was visited
  ln.acc &= ln.uniform ? (desc.y == ln.fbase ? desc_mask(desc) : 0ULL) : mask_in_window(desc, ln.fbase);
  if (ln.keep_list) {
    ln.all_mask = ln.all_mask && desc_is_mask(desc);
    if (first_node) {
      // the window starts at a multiple of 64 rows, like the class bitmaps: one window word = one bitmap word
      ln.wbase = desc.y & ~63u;
      const uint32_t end = desc_is_mask(desc) ? desc.y + 64u : desc.y + desc.z;  // one past the last row it can hold
      if (ln.wl) {
        ln.wn = (end - ln.wbase + 63u) >> 6;
        ln.window_ok = ln.use_window && ln.wn <= ln.wcap;
        if (ln.window_ok)
          for (uint32_t q = 0; q < ln.wn; ++q) ln.wl[q * ALIGN_BLOCK] = ~0ULL;
      } else {
        ln.window_ok = ln.use_window && end - ln.wbase <= WIDE_ROWS;
        ln.wacc[0] = ln.wacc[1] = ln.wacc[2] = ln.wacc[3] = ~0ULL;
      }
    }
    if (ln.n_cols && colour == ln.last_col) return;
    if (ln.window_ok) {
      ln.last_col = colour;
      ln.n_cols = 1;
      if (ln.wl) {  // (uniform) the window in LDS: same folding, word by word
        if (desc_is_mask(desc)) {
          for (uint32_t q = 0; q < ln.wn; ++q) ln.wl[q * ALIGN_BLOCK] &= mask_in_window(desc, ln.wbase + 64u * q);
        } else {
          const uint64_t *__restrict__ w = ln.cls_bits + (desc.w - 1u);
          const uint32_t n_words = (desc.z + 63u) >> 6;
          const int32_t k0 = ((int32_t)ln.wbase - (int32_t)desc.y) >> 6;
          // WCH words at a time: their bitmap loads go out together (one latency per chunk, not one per word) and fold into
          // the window with ds_and -- no read of the window, nothing for the next chunk's loads to wait for.  (Chunks of 4 / 8 /
          // 12 words: the same time; pairs of words per 16-byte load: slower, profiles/r03_experiments.txt 13.)
          constexpr uint32_t WCH = 8;
          for (uint32_t q0 = 0; q0 < ln.wn; q0 += WCH) {
            uint64_t g[WCH];
#pragma unroll
            for (uint32_t j = 0; j < WCH; ++j) {
              const int32_t k = k0 + (int32_t)(q0 + j);
              g[j] = (q0 + j < ln.wn && k >= 0 && (uint32_t)k < n_words) ? w[k] : 0ULL;
            }
#pragma unroll
            for (uint32_t j = 0; j < WCH; ++j)
              if (q0 + j < ln.wn) atomicAnd((unsigned long long *)&ln.wl[(q0 + j) * ALIGN_BLOCK], (unsigned long long)g[j]);
          }
        }
        return;
      }
      if (desc_is_mask(desc)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ln.wacc[q] &= mask_in_window(desc, ln.wbase + 64u * (uint32_t)q);
      } else {
        // word k of the class bitmap = rows desc.y + 64 k (desc.y is a multiple of 64): no shifts, and a window word
        // that is already empty is not loaded
        const uint64_t *__restrict__ w = ln.cls_bits + (desc.w - 1u);
        const uint32_t n_words = (desc.z + 63u) >> 6;
        const int32_t k0 = ((int32_t)ln.wbase - (int32_t)desc.y) >> 6;  // bitmap word under window word 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int32_t k = k0 + q;
          if (ln.wacc[q]) ln.wacc[q] &= (k >= 0 && (uint32_t)k < n_words) ? w[k] : 0ULL;
        }
      }
      return;
    }
    uint32_t j = ln.n_cols;
    ln.last_col = colour;
    if (j < LDS_COLS) {
      ln.lc[j * ALIGN_BLOCK] = colour;
    } else if (j - LDS_COLS < ln.ws_rows) {
      ln.ws[(uint64_t)(j - LDS_COLS) * ln.ws_lanes] = colour;
    } else {
      ln.overflow = 1;
      return;
    }
    ln.n_cols = j + 1;
  }
}

// compare n bases backward: key[rlast - i] vs unitig[ulast - i]
__device__ __forceinline__ uint32_t cmp_bwd(const Lane &ln, const NodeRec &nr, const uint64_t *__restrict__ unitig,
                                            uint32_t rlast, uint32_t ulast, uint32_t n, uint32_t allowed,
                                            uint32_t &mism, bool &premature) {
  uint32_t matched = 0, seen = 0;
  premature = false;
  while (matched < n) {
    uint32_t c = n - matched < 32u ? n - matched : 32u;
    uint64_t x = lds_bits(ln.rd, rlast - matched - c + 1u, c) ^ node_bits(nr, unitig, ulast - matched - c + 1u, c);
    uint64_t m = (x | (x >> 1)) & 0x5555555555555555ULL;
    uint32_t cnt = (uint32_t)__popcll(m);
    if (seen + cnt <= allowed) {
      seen += cnt;
      mism += cnt;
      matched += c;
    } else {
      uint32_t k = allowed - seen;
      for (uint32_t t = 0; t < k; ++t) m &= m - 1;
      uint32_t bit = (uint32_t)__ffsll((long long)m) - 1u;
      matched += bit >> 1;
      mism += k + 1;
      premature = true;
      break;
    }
  }
  return matched;
}

// map_read_to_nodes_with_mismatch for the mate occupying key bases [base0, base0 + L).
//
// Same walk as the reference's (seed search, optional left extension, forward loop with re-seeding), laid out
// as alternating phases so that a wave does not pay for scan rounds on every walk iteration: lanes that need
// a seed scan together (SEED phase), then every seeded lane walks until it ends or needs the next seed
// (WALK phase); a typical read goes through 1-3 cycles.
// pre: result of a direct probe of position 0 already made for this mate (by the tile partition step):
// 0 = none made, 1 = miss, 2 = hit with pre_seed = node << 32 | offset
// the unitig a dictionary entry names: itself, or -- in an index with stretch records, whose dictionary names a unitig by its
// first record (walk_fast) -- the record's unitig
__device__ __forceinline__ uint32_t seed_node(const DevIndex &ix, uint32_t w) { return ix.srec_node ? ix.srec_node[w] : w; }

__device__ bool walk(const DevIndex &ix, Lane &ln, uint32_t base0, uint32_t L, uint32_t allowed, uint32_t &coverage,
                     uint32_t &mismatches, uint32_t pre, uint64_t pre_seed PROF_ARGS) {
  ln.n_cols = 0;
  ln.walk_nodes = 0;
  if (L < KMER) return false;
  uint32_t cov = 0, mm = 0;
  const uint32_t left_thr = (uint32_t)(0.2 * (double)L);
  uint32_t kmer_pos = 0;
  const uint32_t last_kmer_pos = L - KMER;
  uint32_t node = 0, koff = 0;
  bool first = true, done = false, need_seed = true;
  do {
    PROF(3)
    if (need_seed) {  // SEED phase
      need_seed = false;
      bool have;
      if (first && pre == 2u) {
        have = true;
        node = (uint32_t)(pre_seed >> 32);
        koff = (uint32_t)pre_seed;
      } else if (first && pre == 1u) {
        kmer_pos = 3;
        have = find_match(ix, ln, base0, kmer_pos, last_kmer_pos, node, koff, true, true);
      } else {
        // (a re-seed: the scan rounds start at kmer_pos itself -- the k-mer there holds the base that just failed, a
        // direct probe of it is a wasted fetch; the first seed of mate 1 still goes through the direct probe)
        have = find_match(ix, ln, base0, kmer_pos, last_kmer_pos, node, koff, !first, first);
      }
      if (have) node = seed_node(ix, node);
      if (!have) {
        done = true;
      } else if (first && kmer_pos >= left_thr) {  // left extension, only behind a late first seed
        uint32_t last_pos = kmer_pos - 1;
        uint32_t pnode = node;
        uint32_t poff = koff > 0 ? koff - 1 : 0;
        for (;;) {
          NodeRec nr = load_node(ix, pnode);
          uint32_t n = last_pos + 1 < poff + 1 ? last_pos + 1 : poff + 1;
          bool prem;
          uint32_t matched = cmp_bwd(ln, nr, ix.unitig, base0 + last_pos, poff, n, allowed, mm, prem);
          cov += matched;
          if (last_pos + 1 - matched == 0 || prem) break;
          last_pos -= matched;
          uint32_t nbase = lds_base(ln.rd, base0 + last_pos);
          if (nr_exts(nr) & (1u << nbase)) {
            uint4 le = ix.node_ledge[pnode];
            pnode = sel4(le, nbase);
            const uint4 h0 = ix.node_rec[(size_t)pnode * 4], h1 = ix.node_rec[(size_t)pnode * 4 + 1];
            poff = (h0.x & 0xFFFFFFu) - KMER;
            push_col(ln, h0.y, make_uint4(h0.w, h1.x, h1.y, h1.z));
          } else {
            break;
          }
        }
      }
      first = false;
    }
    // WALK phase, flattened: one iteration = (enter the next unitig, if the lane stands at one) + (compare ONE stretch of at
    // most 32 bases) + (leave the unitig, if that stretch was its last).  With a loop per unitig around a loop per stretch
    // the wave ran max-over-lanes stretches for every unitig (5 x 8 iterations for reads that need 8 each); here a lane
    // takes as many iterations as it has stretches, whatever unitigs they fall in.
    //
    // LOCAL RE-SEED (tent / tneed).  A compare that breaks on read base p leaves the reference searching its next seed at
    // p, p + 3, ...  For a read with a substitution that seed is the k-mer at p + 3, three bases further down the very
    // path the walk is on -- so instead of a filter line and a dictionary probe (two fetches from beyond L2, one behind
    // the other) the walk goes on TENTATIVELY: it skips base p and compares the next 32 bases against the graph, hopping
    // unitigs by the read's bases as usual, without counting anything.  When all 32 agree,
    //   * the k-mer at p + 3 is the graph's k-mer at that place (every k-mer has one place in the graph): the seed the
    //     reference finds, in the unitig the last compared base lies in, ending at that base;
    //   * the k-mer at p is absent: the 29 bases behind p are the graph's, the graph's k-mer there starts with another
    //     base, and the "several left flanks" set (flat_index.h) says no second base makes a library k-mer with them.
    // The walk then stands exactly where the reference's stands after that seed (probes + 2: p missed, p + 3 hit).  Any
    // disagreement, a missing edge, a 29-mer that may have several flanks: back to p and to the general seed search.
    // State of the lane inside this phase: ST_IN = inside a unitig, ST_ENTER = stands at the head of `node` (offset koff),
    // ST_SEED = needs the general seed search, ST_DONE.  (One integer instead of a handful of flags: every flag that lives
    // across divergent branches is a lane mask the compiler has to patch at each of them.)
    // tneed: bases the tentative walk has still to compare; NO_TENT (huge) outside it, so that one three-way minimum
    // bounds a stretch in both modes, and counting it down everywhere is harmless.
    PROF(4)
    enum : uint32_t { ST_IN = 0, ST_ENTER = 1, ST_ENTER_T = 2, ST_SEED = 3, ST_DONE = 4 };
    constexpr uint32_t NO_TENT = 0x40000000u;
    uint32_t st = done ? ST_DONE : ST_ENTER, tneed = NO_TENT;
    NodeRec nr;
    uint32_t upos = 0, n_left = 0, seen = 0;
    // (everything the tentative walk needs sits behind tests that fail for a lane outside it: on reads without a
    // difference the loop costs what it cost before -- the first form, with the tests spread over the common path, took
    // 10 % more time on exact reads)
    while (st < ST_SEED) {
      if (st == ST_ENTER) {
        st = ST_IN;
        nr = load_node(ix, node);
        kmer_pos += KMER;
        cov += KMER;
        upos = koff + KMER;
        push_col(ln, nr.q0.y, nr_desc(nr));
        const uint32_t remaining = L - kmer_pos, informative = nr_len(nr) - upos;
        n_left = remaining < informative ? remaining : informative;
        seen = 0;
      } else if (st == ST_ENTER_T) {  // a hop of the tentative walk: nothing is pushed, tneed bounds the stretch
        st = ST_IN;
        nr = load_node(ix, node);
        kmer_pos += KMER;
        cov += KMER;
        upos = KMER;
        const uint32_t informative = nr_len(nr) - upos;
        n_left = tneed < informative ? tneed : informative;
      }
      bool prem = false;
      uint32_t junction = 4u;  // 0..3: the read base at kmer_pos, known from the compare of this iteration; 4: not known
      if (n_left) {
        const uint32_t c = n_left < 32u ? n_left : 32u;
        // (the 32 read bases from kmer_pos on; the compare takes the first c, and the base behind them -- the one that
        // picks the way out of the unitig when the stretch ends it -- comes out of the same window)
        const uint64_t win = lds_window(ln.rd, base0 + kmer_pos);
        junction = c < 32u ? (uint32_t)(win >> (62u - 2u * c)) & 3u : 4u;
        const uint64_t x = (win >> (64u - 2u * c)) ^ node_bits(nr, ix.unitig, upos, c);
        uint64_t m = (x | (x >> 1)) & 0x5555555555555555ULL;
        const uint32_t cnt = (uint32_t)__popcll(m);
        uint32_t adv = c;
        // the usual setting, and the tentative walk under any setting: the first mismatch ends the compare
        bool strict = allowed == 0;  // (uniform)
        if (!strict) strict = tneed < NO_TENT / 2;
        if (strict) {
          prem = cnt != 0;
          const uint32_t bit = 63u - (uint32_t)__clzll((long long)(m | 1ULL));
          const uint32_t k = c - 1u - (bit >> 1);  // bases in front of the first differing one
          adv = prem ? k : c;
          bool on = false;
          // A differing base p inside the stretch, the only one in it, and the walk not tentative yet: go on tentatively
          // with the rest of this very stretch (it agrees) instead of ending the iteration at p -- the lane would otherwise
          // spend an iteration on the split, and the wave runs as many iterations as its slowest lane.
          if (NIMBLE_TENT && prem && cnt == 1u && tneed >= NO_TENT / 2 && ix.mleft && kmer_pos + k + 3u <= last_kmer_pos)
            on = !mleft_maybe(ix.mleft, ix.mleft_log2, lds_bits(ln.rd, base0 + kmer_pos + k + 1u, KMER - 1u));
          if (on) {
            const uint32_t r = c - 1u - k;  // bases of the stretch behind p
            mm += 1;
            prem = false;
            adv = c;
            cov -= 1;                       // (p itself is not counted; the r bases behind it are, tentatively)
            tneed = 32u - r + c;            // (c comes off again below)
            const uint32_t rest = n_left - c;
            n_left = rest < 32u - r ? rest : 32u - r;
          } else {
            mm += prem ? 1u : 0u;
            n_left = prem ? 0u : n_left - c;
          }
        } else if (seen + cnt <= allowed) {
          seen += cnt;
          mm += cnt;
          n_left -= c;
        } else {  // the base that takes this unitig's mismatches above `allowed` ends the compare, unaccepted
          const uint32_t k = allowed - seen;
          for (uint32_t t = 0; t < k; ++t) m &= ~(1ULL << (63 - __clzll((long long)m)));
          const uint32_t bit = 63u - (uint32_t)__clzll((long long)m);
          adv = c - 1u - (bit >> 1);
          mm += k + 1;
          prem = true;
          n_left = 0;
        }
        cov += adv;
        kmer_pos += adv;
        upos += adv;
        tneed -= adv;
      }
      if (n_left == 0) {  // the unitig, or the tentative stretch, is done: a successor, a new seed, or the end
        // the read base at kmer_pos (the column ends in a zero word: safe at kmer_pos == L); a compare that ran its whole
        // stretch in this iteration has it already
        const uint32_t nbase = (junction < 4u && !prem) ? junction : lds_base(ln.rd, base0 + kmer_pos);
        const bool edge = ((nr_exts(nr) >> 4) >> nbase) & 1u;
        if (tneed < NO_TENT / 2) {  // inside the tentative walk
          if (!prem && tneed == 0) {
            // all 32 bases behind p agree: the seed is the k-mer that ends with the last of them, in this unitig.  Enter it
            // the way a seed is entered (the record comes from L1 this time): kmer_pos = p + 3, nothing counted so far.
            // (Pushing the unitig right here instead costs a second copy of push_col in the loop: measured 25 % slower
            // on every kind of read -- the loop's register allocation does not survive it.)
            koff = upos - KMER;
            kmer_pos -= KMER;
            cov -= 32u;
            ln.probes += 2;
            tneed = NO_TENT;
            st = ST_ENTER;
          } else if (!prem && edge) {  // on along the read's base; the junction base agrees by the edge's label
            node = sel4(nr.re, nbase);
            kmer_pos -= KMER - 1;
            cov -= KMER - 1;
            tneed -= 1;
            st = ST_ENTER_T;
          } else {  // a differing base, or the read leaves the graph: back to p, general search
            kmer_pos -= 33u - tneed;
            cov -= 32u - tneed;
            mm -= prem ? 1u : 0u;
            tneed = NO_TENT;
            st = ST_SEED;
          }
        } else if (!prem && kmer_pos < L && edge) {
          node = sel4(nr.re, nbase);
          koff = 0;
          kmer_pos -= KMER - 1;
          cov -= KMER - 1;
          st = ST_ENTER;
        } else if (kmer_pos >= L || kmer_pos > last_kmer_pos) {
          st = ST_DONE;
        } else {
          st = ST_SEED;  // dead end or mismatch budget exceeded: search the next seed from kmer_pos
        }
      }
    }
    PROF(5)
    done = st == ST_DONE;
    need_seed = st == ST_SEED;
  } while (!done);
  if (ln.walk_nodes == 0) return false;
  coverage = cov;
  mismatches = mm;
  return true;
}

// ---------------------------------------------------------------------------------------------
// walk_fast (round 4): map_read_with_mismatch for ONE mate in its common shapes -- or nothing at all.
//
// walk() above is every case of the reference's walk in one loop over 64-byte unitig records, and a wave pays for every
// case in every iteration.  This one walks the index's STRETCH RECORDS (flat_index.h: a unitig's bases behind its first
// k-mer cut into stretches of at most 32, a 32-byte record each): one step = one record (two 16-byte gathers of one line)
// = one compare = one way out -- the next stretch of the same unitig, or the unitig behind the read's next base.  What
// bounds both walks is VALU issue at a quarter of the lanes -- how often a wave runs a block for a few of its lanes -- and the
// length of a tile's chain of round trips (profiles/r04_experiments.txt 13; the chip's rate of dependent gathers, item 1 there,
// is three times what the launch asks of it).  The seed search is walk()'s (find_match); a substitution is
// stepped over the way walk() does it (the local re-seed: the 32 bases behind it must agree along the graph, then the
// reference's next seed is the k-mer that ends there), a read that leaves the graph searches its next seed and walks on.
// Everything else is walk()'s, step by step: the left extension behind a late first seed (over the general walk's unitig
// records: once per read, for one bench read in seventy), the flank test of a short cut (asked when it commits), the mismatch
// budget per unitig.  (A first form handed those reads to a second launch of the general walk through a redo list: 2.4 % of the
// bench reads, 0.14 ms per 10 M -- the latency of that launch's hardest tile; profiles/r04_experiments.txt 3, 10.)
// Same counters as walk(): a unitig entered or re-entered by a seed counts as a visit, a committed short cut as two probes.
// STRICT: num_mismatches == 0 (the usual setting); otherwise a unitig tolerates `allowed` differing bases.
// pre / pre_seed: the tile's direct probe of position 0, as for walk().  Returns false = no unitig visited (NoMatch), true =
// walked (the lane's running intersection -- ln.acc / min_* / last_rec -- describes the visited classes).
template <bool STRICT>
__device__ __forceinline__ bool walk_fast(const DevIndex &ix, Lane &ln, uint32_t base0, uint32_t L, uint32_t allowed,
                                          uint32_t pre, uint64_t pre_seed, uint32_t &coverage, uint32_t &mismatches PROF_ARGS) {
  constexpr uint32_t NO_TENT = 0x40000000u;
  ln.walk_nodes = 0;
  WCOUNT(1)
  ln.n_cols = 0;
  if (L < KMER) return false;
  const uint64_t *rd = ln.rd;
  const uint32_t last_kmer_pos = L - KMER;
  // ---- the first seed: the tile's direct probe, or the scan rounds from position 3 (0 for a mate nobody has probed)
  uint32_t node = (uint32_t)(pre_seed >> 32), koff = (uint32_t)pre_seed, kmer_pos = 0;
  bool have = pre == 2u;
  if (pre != 0u) ln.probes++;  // (the tile's direct probe of this mate)
  if (!have) {
    kmer_pos = pre == 1u ? 3u : 0u;
    have = find_match(ix, ln, base0, kmer_pos, last_kmer_pos, node, koff, pre == 1u, true);
  }
  PROF(4)
  if (!have) return false;
  uint32_t cov = 0, mm = 0;
  uint32_t nodes = 0, commits = 0;
  uint64_t entries = 0;
  uint64_t acc = ~0ULL;
  uint32_t min_len = 0xFFFFFFFFu, min_col = 0, rec = 0;
  int status;  // -1 walking, 0 done, 2 the next seed has to be searched from kpos
  if (kmer_pos >= (uint32_t)(0.2 * (double)L)) {
    // LEFT EXTENSION behind a late first seed, as in walk(): the read's bases in front of the seed against the unitig, and
    // on into the left neighbours picked by the read's base, every entered neighbour a visit.  Over the general walk's
    // unitig records (this happens once per read, for one read in seventy of the bench recipe).
    uint32_t last_pos = kmer_pos - 1;
    uint32_t pnode = ix.srec_node[node];  // (the dictionary names a unitig by its first record)
    uint32_t poff = koff > 0 ? koff - 1 : 0;
    for (;;) {
      WCOUNT(12)
      const NodeRec nr = load_node(ix, pnode);
      const uint32_t n = last_pos + 1 < poff + 1 ? last_pos + 1 : poff + 1;
      bool prem;
      const uint32_t matched = cmp_bwd(ln, nr, ix.unitig, base0 + last_pos, poff, n, allowed, mm, prem);
      cov += matched;
      if (last_pos + 1 - matched == 0 || prem) break;
      last_pos -= matched;
      const uint32_t nbase = lds_base(rd, base0 + last_pos);
      if (!(nr_exts(nr) & (1u << nbase))) break;
      pnode = sel4(ix.node_ledge[pnode], nbase);
      const uint4 h0 = ix.node_rec[(size_t)pnode * 4], h1 = ix.node_rec[(size_t)pnode * 4 + 1];
      poff = (h0.x & 0xFFFFFFu) - KMER;
      const uint32_t l = h0.w & ~CLS_MASK_FLAG;
      ++nodes;
      if (ln.want_counters) entries += l;
      const bool smaller = l < min_len;
      min_len = smaller ? l : min_len;
      min_col = smaller ? h0.y : min_col;
      acc &= u64of(h1.y, h1.z);  // (class masks are relative to the component's first row, and a neighbour is of the component)
    }
  }
  bool first_seed = true;
  for (;;) {  // one seed, one forward walk
    // the seed's k-mer is the unitig's k-mer at offset koff: what is compared next is the unitig's base 30 + koff, i.e.
    // base o of its stretch j (o = 32 <=> that stretch is used up; koff = 0: the head of the unitig)
    const uint32_t j = koff ? (koff - 1u) >> 5 : 0u;
    const uint32_t seed_rec = node + j;  // (the dictionary of an index with stretch records names records)
    // (a seed in another component than the walk so far -- `rec` still is a record of that -- has no row in common with it)
    if (!first_seed && ix.srec_base[seed_rec] != ix.srec_base[rec]) acc = 0;
    first_seed = false;
    rec = seed_rec;
    uint32_t o = koff - 32u * j;
    uint32_t kpos = kmer_pos + KMER;
    cov += KMER;
    uint32_t seen = 0, tneed = NO_TENT, t_pos = 0;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = make_uint4(0, 0, 0, 0);
    bool load = true, push = true;
    status = -1;
    while (status < 0) {
      WCOUNT(2)
      if (load) {
        WCOUNT(3)
        // the record's first half (header, colour, bases: what the compare needs) ahead of the second (class mask,
        // neighbours: needed behind the compare)
        r0 = ld16(ix.srec + (size_t)rec * 2);
        r1 = ld16(ix.srec + (size_t)rec * 2 + 1);
        load = false;
      }
      const uint32_t hdr = r0.x, nb = hdr & 63u;
      if (tneed == 0) {
        WCOUNT(5)
        tneed = NO_TENT;
        // the flank test of the short cut (walk(): "on"), asked only now that the 32 bases behind the substitution have agreed
        // -- a stretch that breaks off earlier never needed it, and a positive answer falls back exactly as a broken stretch
        // does (same position, same coverage, same counters)
        if (mleft_maybe(ix.mleft, ix.mleft_log2, lds_bits(rd, base0 + t_pos + 1u, KMER - 1u))) {
          cov = t_pos - (kpos - cov - 1u);
          kpos = t_pos;
          status = 2;
          continue;
        }
        // the reference's seed is the k-mer that ends with the last of the 32 bases, in this unitig -- entered the way a seed
        // is entered (two probes: p missed, p + 3 hit; the 30 bases of the seed count, the 32 compared did not)
        cov -= 2;
        ++commits;
        push = true;
      }
      bool entered = false;
      if (push) {  // a unitig entered by a seed or along an edge (never inside a tentative stretch)
        push = false;
        entered = true;
        const uint32_t l = (hdr >> 16) & 0x7Fu;
        ++nodes;
        if (ln.want_counters) entries += l;
        const bool smaller = l < min_len;
        min_len = smaller ? l : min_len;
        min_col = smaller ? r0.y : min_col;
        seen = 0;
      }
      bool prem = false;
      uint32_t junction = 4u;
      const uint32_t rem = nb - o, ahead = L - kpos;
      const uint32_t lim = ahead < tneed ? ahead : tneed;
      const uint32_t c = rem < lim ? rem : lim;
      if (c) {
        const uint64_t win = lds_window(rd, base0 + kpos);
        junction = c < 32u ? (uint32_t)(win >> (62u - 2u * c)) & 3u : 4u;
        const uint64_t x = (win ^ (u64of(r0.z, r0.w) << (2u * o))) >> (64u - 2u * c);
        uint64_t m = (x | (x >> 1)) & 0x5555555555555555ULL;
        uint32_t adv = c;
        if (m) {
          WCOUNT(4)
          const uint32_t cnt = (uint32_t)__popcll(m);
          const bool tent = tneed < NO_TENT / 2;
          if (STRICT || tent) {
            const uint32_t bit = 63u - (uint32_t)__clzll((long long)m);
            const uint32_t k = c - 1u - (bit >> 1);  // bases in front of the first differing one
            if (tent) {
              // a second difference inside the 32 bases: back to the first one, general search (walk(): the same)
              // (coverage at the differing base: read position and coverage move together, the tentative stretch is one behind)
              cov = t_pos - (kpos - cov - 1u);
              kpos = t_pos;
              adv = 0;
              status = 2;
            } else if (cnt == 1u && kpos + k + 3u <= last_kmer_pos) {
              // the rest of this very stretch agrees: go on tentatively from here (walk(): "on"; its flank test: at the commit)
              t_pos = kpos + k;  // (where the walk stands if the short cut does not work out: at the differing base)
              mm += 1;
              cov -= 1;  // (the differing base itself is not covered)
              tneed = 32u - (c - 1u - k) + c;
            } else {
              prem = true;
              mm += 1;
              adv = k;
            }
          } else if (seen + cnt <= allowed) {
            seen += cnt;
            mm += cnt;
          } else {  // the base that takes this unitig above `allowed` ends the compare, unaccepted
            const uint32_t k = allowed - seen;
            for (uint32_t t = 0; t < k; ++t) m &= ~(1ULL << (63 - __clzll((long long)m)));
            const uint32_t bit = 63u - (uint32_t)__clzll((long long)m);
            adv = c - 1u - (bit >> 1);
            mm += k + 1;
            prem = true;
          }
        }
        cov += adv;
        kpos += adv;
        o += adv;
        tneed -= adv;
      }
      acc &= entered ? u64of(r1.x, r1.y) : ~0ULL;  // (the entered unitig's class: the record's second half, here at the latest)
      if (status < 0 && tneed != 0) {  // (a finished tentative stretch commits first, at the head of the next step)
        const bool tent = tneed < NO_TENT / 2;
        if (prem) {
          status = kpos > last_kmer_pos ? 0 : 2;  // the end, or a new seed is needed
        } else if (o < nb) {
          status = 0;  // the read ends inside this stretch (kpos == L)
        } else if (!(hdr & SREC_LAST)) {
          ++rec;  // the unitig goes on
          o = 0;
          load = true;
        } else if (kpos >= L && !tent) {
          status = 0;
        } else {  // the unitig is done: on along the read's next base (the junction base agrees by the edge's label)
          WCOUNT(6)
          const uint32_t nbase = junction < 4u ? junction : lds_base(rd, base0 + kpos);
          const uint32_t rext = (hdr >> 8) & 0xFu;
          if ((rext >> nbase) & 1u) {
            if (hdr & SREC_MANY) rec = sel4(ix.srec_many[r1.z], nbase);  // (a fork with three or four ways out: rare)
            else rec = (rext & ((1u << nbase) - 1u)) ? r1.w : r1.z;      // (the lowest extension base's neighbour first)
            o = 0;
            load = true;
            kpos += 1;
            cov += 1;
            tneed -= tent ? 1u : 0u;
            push = !tent;
          } else if (tent) {
            // the read leaves the graph inside the tentative stretch: back to the differing base, general search
            cov = t_pos - (kpos - cov - 1u);
            kpos = t_pos;
            status = 2;
          } else {
            status = kpos > last_kmer_pos ? 0 : 2;  // no such edge: the end, or a new seed is needed
          }
        }
      }
    }
    PROF(5)
    if (status != 2) break;
    // the next seed, searched the way walk() does behind a walk (the scan rounds start at kpos itself)
    kmer_pos = kpos;
    WCOUNT(11)
    if (!find_match(ix, ln, base0, kmer_pos, last_kmer_pos, node, koff, true, false)) {
      status = 0;
      PROF(8)
      break;
    }
    PROF(8)
  }
  ln.nodes += nodes;
  ln.probes += 2u * commits;
  if (ln.want_counters) ln.entries += entries;
  ln.walk_nodes = nodes;
  ln.acc = acc;
  ln.last_rec = rec;  // (srec_base of any record of the walk is its component's first row: the walk's fbase, if anybody asks)
  ln.fbase = 0;
  ln.min_len = min_len;
  ln.min_col = min_col;
  ln.all_mask = true;
  coverage = cov;
  mismatches = mm;
  return true;
}

// general form of nodes_to_eq_class (some visited class spans 64 rows or more): smallest class first,
// membership test of each of its rows in every other visited class.  Everything is passed and returned by
// value so that the (rare) out-of-line call does not force the caller's state into scratch.
struct IRes {
  uint32_t count;
  uint64_t hash;
};
__device__ __noinline__ IRes intersect_general(const uint4 *cls_desc, const uint32_t *cls_off, const uint32_t *cls_ids,
                                               const uint64_t *cls_bits, const uint32_t *lc, const uint32_t *ws,
                                               uint32_t ws_lanes, uint32_t n_cols, uint32_t best, uint32_t bl,
                                               uint32_t *out) {
  uint32_t count = 0, first_id = 0, last_id = 0;
  uint64_t gmask = 0;
  uint64_t h = class_hash_init();
  // Bitmap form: when every visited class is a 64-row mask or has a span bitmap, the intersection is an AND of
  // 64-row words over the rows all of them cover -- a few loads per class and word instead of a binary search per
  // row of the smallest class and visited class (allele families of 100 rows: 384 -> see DESIGN.md).
  bool bitmaps = true;
  uint32_t lo = 0, hi = 0xFFFFFFFFu;  // rows every class could contain: [lo, hi]
  for (uint32_t j = 0; j < n_cols; ++j) {
    const uint32_t c = j < LDS_COLS ? lc[j * ALIGN_BLOCK] : ws[(uint64_t)(j - LDS_COLS) * ws_lanes];
    const uint4 d = cls_desc[c];
    uint32_t f, l;
    if (desc_is_mask(d)) {
      const uint64_t m = desc_mask(d);
      f = d.y;
      l = d.y + 63u - (uint32_t)__clzll((long long)m);
    } else if (d.w != 0u) {
      f = d.y;
      l = d.y + d.z - 1u;
    } else {
      bitmaps = false;
      break;
    }
    lo = f > lo ? f : lo;
    hi = l < hi ? l : hi;
  }
  if (bitmaps) {
    for (uint32_t row0 = lo; lo <= hi && row0 <= hi; row0 += 64u) {
      uint64_t acc = hi - row0 >= 63u ? ~0ULL : ((1ULL << (hi - row0 + 1u)) - 1ULL);
      for (uint32_t j = 0; j < n_cols && acc; ++j) {
        const uint32_t c = j < LDS_COLS ? lc[j * ALIGN_BLOCK] : ws[(uint64_t)(j - LDS_COLS) * ws_lanes];
        acc &= class_rows64(cls_desc[c], cls_bits, row0);
      }
      for (; acc; acc &= acc - 1) {
        const uint32_t id = row0 + (uint32_t)__ffsll((long long)acc) - 1u;
        if (out) out[count] = id;
        h = class_hash_step(h, id);
        if (count == 0) first_id = id;
        last_id = id;
        if (id - first_id < 64u) gmask |= 1ULL << (id - first_id);
        ++count;
      }
      if (row0 > 0xFFFFFFFFu - 64u) break;
    }
  } else {
    const uint32_t *bids = cls_ids + cls_off[best];
    for (uint32_t t = 0; t < bl; ++t) {
      const uint32_t id = bids[t];
      bool ok = true;
      for (uint32_t j = 0; j < n_cols && ok; ++j) {
        const uint32_t c = j < LDS_COLS ? lc[j * ALIGN_BLOCK] : ws[(uint64_t)(j - LDS_COLS) * ws_lanes];
        if (c == best) continue;
        const uint4 d = cls_desc[c];
        if (desc_is_mask(d)) {
          const uint32_t off = id - d.y;
          ok = id >= d.y && off < 64u && ((desc_mask(d) >> off) & 1ULL);
        } else {
          const uint32_t len = desc_len(d);
          const uint32_t *__restrict__ ids = cls_ids + cls_off[c];
          uint32_t blo = 0, bhi = len;
          while (blo < bhi) {
            uint32_t mid = (blo + bhi) >> 1;
            if (ids[mid] < id) blo = mid + 1;
            else bhi = mid;
          }
          ok = blo < len && ids[blo] == id;
        }
      }
      if (ok) {
        if (out) out[count] = id;
        h = class_hash_step(h, id);
        if (count == 0) first_id = id;
        last_id = id;
        if (id - first_id < 64u) gmask |= 1ULL << (id - first_id);
        ++count;
      }
    }
  }
  IRes r;
  r.count = count;
  r.hash = class_hash_final(h, count);
  if (count && last_id - first_id < 64u) r.hash = class_hash_mask(count, first_id, gmask);  // mask-form rule
  return r;
}

// nodes_to_eq_class.  The mask-form intersection was folded during the walk (push_col); this turns it into
// the class length, the content hash and, when `out` is given, the ascending row ids.  best_col / best_len
// = the smallest visited class: a result of that length *is* that class.
struct MaskRes {  // the result as base + mask (valid when is_mask), normalised so that bit 0 is set
  bool is_mask;
  uint32_t base;
  uint64_t mask;
};
template <bool WIDE>
__device__ uint32_t finish_class(const DevIndex &ix, const Lane &ln, uint64_t &hash, uint32_t *out, MaskRes &mr) {
  mr.is_mask = !WIDE || ln.all_mask;
  mr.base = 0;
  mr.mask = 0;
  if (WIDE && !ln.all_mask) {
    if (ln.window_ok) {
      // the intersection was folded into the register window during the walk: rows wbase + 64 q + bit
      uint32_t count = 0, first_id = 0, last_id = 0;
      uint64_t gmask = 0;
      uint64_t h = class_hash_init();
      auto take = [&](uint32_t q, uint64_t word) {
        for (uint64_t m = word; m; m &= m - 1) {
          const uint32_t id = ln.wbase + 64u * q + (uint32_t)__ffsll((long long)m) - 1u;
          if (out) out[count] = id;
          h = class_hash_step(h, id);
          if (count == 0) first_id = id;
          last_id = id;
          if (id - first_id < 64u) gmask |= 1ULL << (id - first_id);
          ++count;
        }
      };
      if (ln.wl) {
        for (uint32_t q = 0; q < ln.wn; ++q) take(q, ln.wl[q * ALIGN_BLOCK]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) take((uint32_t)q, ln.wacc[q]);
      }
      hash = class_hash_final(h, count);
      if (count && last_id - first_id < 64u) {  // the result fits the mask form: same rule as everywhere
        hash = class_hash_mask(count, first_id, gmask);
        mr.is_mask = true;
        mr.base = first_id;
        mr.mask = gmask;
      }
      return count;
    }
    const IRes r = intersect_general(ix.cls_desc, ix.cls_off, ix.cls_ids, ix.cls_bits, ln.lc, ln.ws, ln.ws_lanes, ln.n_cols,
                                     ln.min_col, ln.min_len, out);
    hash = r.hash;
    return r.count;
  }
  const uint64_t acc = ln.acc;
  const uint32_t count = (uint32_t)__popcll(acc);
  if (acc) {
    const uint32_t tz = (uint32_t)__ffsll((long long)acc) - 1u;
    mr.base = ln.fbase + tz;
    mr.mask = acc >> tz;
  }
  hash = class_hash_mask(count, mr.base, mr.mask);
  if (out) {
    uint32_t k = 0;
    for (uint64_t m = acc; m; m &= m - 1) out[k++] = ln.fbase + (uint32_t)__ffsll((long long)m) - 1u;
  }
  return count;
}

// number of rows in the register window (cheap: the class itself is only spelled out when somebody needs it)
__device__ __forceinline__ uint32_t window_count(const Lane &ln) {
  if (ln.wl) {
    uint32_t c = 0;
    for (uint32_t q = 0; q < ln.wn; ++q) c += (uint32_t)__popcll(ln.wl[q * ALIGN_BLOCK]);
    return c;
  }
  return (uint32_t)(__popcll(ln.wacc[0]) + __popcll(ln.wacc[1]) + __popcll(ln.wacc[2]) + __popcll(ln.wacc[3]));
}
// is interned class `id` (ascending ids in cls_ids, `count` of them) exactly the content of the register window?
__device__ __forceinline__ bool window_equals_class(const DevIndex &ix, const Lane &ln, uint32_t id, uint32_t count) {
  const uint32_t *__restrict__ ids = ix.cls_ids + ix.cls_off[id];
  if (ln.wl) {
    for (uint32_t t = 0; t < count; ++t) {
      const uint32_t off = ids[t] - ln.wbase, q = off >> 6;  // (wraps far above the window for a row below it)
      if (q >= ln.wn || !((ln.wl[q * ALIGN_BLOCK] >> (off & 63u)) & 1ULL)) return false;
    }
    return true;
  }
  for (uint32_t t = 0; t < count; ++t) {
    const uint32_t off = ids[t] - ln.wbase;  // wraps far above WIDE_ROWS for a row below the window
    const uint32_t q = off >> 6;
    const uint64_t w = q == 0 ? ln.wacc[0] : (q == 1 ? ln.wacc[1] : (q == 2 ? ln.wacc[2] : ln.wacc[3]));  // (no scratch)
    if (off >= WIDE_ROWS || !((w >> (off & 63u)) & 1ULL)) return false;
  }
  return true;  // `count` distinct rows, all in the window, and the window holds exactly `count`
}

// ---------------------------------------------------------------------------------------------
// k_align: walk + class + thresholds, one lane per read(-pair), persistent grid-stride blocks
// ---------------------------------------------------------------------------------------------
#ifndef NIMBLE_ALIGN_WAVES
#define NIMBLE_ALIGN_WAVES 8
#endif
#ifndef NIMBLE_RESULT_STAGED
#define NIMBLE_RESULT_STAGED 1
#endif
// WIDE: the index has classes wider than the 64-row mask form (ix.all_local == 0): the visited colours are kept and
// the intersection may go through intersect_general.  The other instantiation carries none of that code -- the
// out-of-line call alone costs the walk registers around it.
// (the instantiation for indexes with wide classes carries the register window and the general intersection: it gets
// 80 registers, i.e. 6 waves per SIMD, instead of spilling at 64)
// MODE (round 4): 0 = every mate through walk() over the 64-byte unitig records; 1 = through walk_fast over the index's
// stretch records (indexes without wide classes that carry them).
template <bool PAIRED, bool COUNTERS, bool WIDE, int MODE>
// (residency: 8 waves per SIMD for the general walk -- 7 and 6 measured slower, round 2 --, 6 for indexes with wide classes (the
// register window), 7 for the fast walk: at 64 registers it spills inside the tile loop, 72 hold it: bench recipe 1.149 -> 1.115 ms,
// exact reads 0.891 -> 0.771, 6 waves 1.207 / 0.761; paired calls carry two mates' state: 5 waves (92 registers, 48 B of scratch:
// configs[3] 2.37 ms at 7 waves, 2.26 at 6, 2.19 at 5 and at 4);
// the counters variant of the fast walk, never timed, gets half of that:
// profiles/r04_experiments.txt 8)
#ifndef NIMBLE_FAST_WAVES
#define NIMBLE_FAST_WAVES 7
#endif
#ifndef NIMBLE_WIDE_WAVES
#define NIMBLE_WIDE_WAVES ((NIMBLE_ALIGN_WAVES * 3) / 4)
#endif
#ifndef NIMBLE_FAST_WAVES_PAIRED
#define NIMBLE_FAST_WAVES_PAIRED (NIMBLE_FAST_WAVES - 2)
#endif
__global__ __launch_bounds__(ALIGN_BLOCK, MODE == 1 ? (COUNTERS ? NIMBLE_FAST_WAVES / 2 : (PAIRED ? NIMBLE_FAST_WAVES_PAIRED : NIMBLE_FAST_WAVES))
                                                    : (WIDE ? NIMBLE_WIDE_WAVES : NIMBLE_ALIGN_WAVES)) void k_align(DevIndex ix, nimble_align_params p,
                                                                           CallBuffers cb) {
  static_assert(!(WIDE && MODE == 1), "the fast walk is for indexes whose classes all have the mask form");
  constexpr int want_counters = COUNTERS ? 1 : 0;
  extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
  const uint32_t tid = threadIdx.x;
  const uint32_t kw = cb.key_words;
  uint64_t *col = lds64 + tid;
  constexpr int nm = PAIRED ? 2 : 1;
  // rows of the LDS columns: the key words, a zero word behind them -- and at least nm + 1, because a finished tile
  // leaves its results in the columns (below)
  const uint32_t krows = kw + 1u > (uint32_t)nm + 1u ? kw + 1u : (uint32_t)nm + 1u;
  // (the LDS columns of the visited-colour lists: the fast walk keeps no lists -- 4 KiB less per block, an eighth block per CU)
  constexpr int LIST_COLS = MODE == 1 ? 0 : LDS_COLS;
  Lane ln;
  ln.rd = col;
  ln.lc = reinterpret_cast<uint32_t *>(lds64 + (size_t)krows * ALIGN_BLOCK) + tid;
  ln.ws_lanes = cb.ws_lanes;
  ln.ws_rows = cb.ws_rows;
  ln.ws = cb.ws_cols + ((uint64_t)blockIdx.x * ALIGN_BLOCK + tid);
  ln.want_counters = want_counters;
  ln.probes = ln.nodes = 0;
  ln.entries = 0;
  ln.overflow = 0;
  ln.n_cols = ln.last_col = ln.walk_nodes = 0;
  ln.acc = 0;
  ln.fbase = ln.min_len = ln.min_col = 0;
  ln.last_rec = 0;
  ln.all_mask = true;
  ln.keep_list = WIDE;
  ln.window_ok = false;
  ln.use_window = WIDE && ix.all_bitmaps != 0;
  ln.uniform = !WIDE && ix.uniform_windows != 0;
  ln.wbase = 0;
  ln.wacc[0] = ln.wacc[1] = ln.wacc[2] = ln.wacc[3] = 0;
  ln.wcap = WIDE ? ix.window_words : 0u;
  ln.wn = 0;
  ln.wl = ln.wcap ? reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(reinterpret_cast<uint32_t *>(lds64 + (size_t)krows * ALIGN_BLOCK) + LIST_COLS * ALIGN_BLOCK) + ALIGN_LDS_EXTRA) + tid
                  : nullptr;
  ln.cls_bits = ix.cls_bits;
  uint32_t c_seeded = 0, c_pre = 0;
  const uint64_t n = cb.n;
  const uint64_t n_tiles = (n + ALIGN_BLOCK - 1) / ALIGN_BLOCK;
  unsigned long long *const tile_counter = (unsigned long long *)cb.tile_ctr;

  // small block-shared arrays behind the columns in the dynamic region (extern base stays 16-byte aligned)
  uint8_t *extra = reinterpret_cast<uint8_t *>(reinterpret_cast<uint32_t *>(lds64 + (size_t)krows * ALIGN_BLOCK) +
                                               LIST_COLS * ALIGN_BLOCK);
  unsigned long long &s_tile = *reinterpret_cast<unsigned long long *>(extra);
  uint32_t *s_cnt = reinterpret_cast<uint32_t *>(extra + 16);
  uint64_t *s_seed = reinterpret_cast<uint64_t *>(extra + 16 + 64);
  uint16_t *s_perm = reinterpret_cast<uint16_t *>(extra + 16 + 64 + ALIGN_BLOCK * 8);
  // per read of the tile: aligned length of mate 0 | of mate 1 << 16 | length of mate 0 << 32 | prefilter verdicts << 48, 56 --
  // loaded with the key in read order (whole lines) and picked up from here behind the partition: the lanes used to gather
  // them again from the call's arrays for their permuted reads, two more round trips in every tile's chain
  uint64_t *s_meta = reinterpret_cast<uint64_t *>(extra + 16 + 64 + ALIGN_BLOCK * 8 + ALIGN_BLOCK * 2);
  uint64_t prev_tile = ~0ULL;
  // Tiles are handed out dynamically (they differ a lot in cost), but NOT through one counter: atomics on one address run
  // at ~80 M/s on this chip whoever issues them, so an increment per 256-read tile is 39 k increments per 10 M reads =
  // 0.5 ms of the atomic unit's time -- which a launch over reads that do nothing showed to be the whole launch
  // (tools/probes/skeleton.hip: 0.497 ms with the counter, 0.006 ms without).  TILE_COUNTERS counters, a 128-byte line
  // each, own a contiguous range of the tiles; a block (its thread 0) draws from the counter of its number and, when that
  // range is used up, from the next one, and it asks for its next tile while it works on the current one (the atomic's
  // round trip used to stand between two barriers of every tile).
  const unsigned long long tiles_per_counter = (n_tiles + TILE_COUNTERS - 1) / TILE_COUNTERS;
  // (thread 0's state of this lives in LDS beside the tile slot, not in registers every lane would carry through the walk:
  // the tile fetched ahead in the low 48 bits -- all ones: none is left --, the counter drawn from and how many were found
  // used up above them)
  unsigned long long &s_next = *reinterpret_cast<unsigned long long *>(extra + 8);
  constexpr unsigned long long NO_TILE = (1ULL << 48) - 1ULL;
  auto fetch_tile = [&](bool first_call) {
    uint32_t my_counter = first_call ? blockIdx.x % TILE_COUNTERS : (uint32_t)(s_next >> 48) & 0xFFu;
    uint32_t counters_tried = first_call ? 0u : (uint32_t)(s_next >> 56);
    unsigned long long got = NO_TILE;
    while (counters_tried < TILE_COUNTERS) {
      const unsigned long long first = (unsigned long long)my_counter * tiles_per_counter;
      if (first < n_tiles) {  // (a small launch leaves most ranges empty: nothing to ask them)
        const unsigned long long t = atomicAdd(tile_counter + (size_t)my_counter * (TILE_COUNTER_STRIDE / 8), 1ULL);
        if (t < tiles_per_counter && first + t < n_tiles) {
          got = first + t;
          break;
        }
      }
      my_counter = my_counter + 1 == TILE_COUNTERS ? 0u : my_counter + 1;  // used up: on to the neighbour's range
      ++counters_tried;
    }
    s_next = got | ((unsigned long long)my_counter << 48) | ((unsigned long long)counters_tried << 56);
  };
  static_assert(TILE_COUNTERS <= 128, "the counter's number and the count of used-up ones share 16 bits of a word");
#ifndef NIMBLE_STATIC_TILES
  if (tid == 0) fetch_tile(true);
#endif
  PROF_DECL
  for (;;) {
    // dynamic tile scheduling: tiles differ a lot in cost (off-target reads probe 41 times)
    PROF(7)
    __syncthreads();
    PROF(0)
#if NIMBLE_RESULT_STAGED
    // The results of the tile just finished sit in the LDS columns, one column per read in read order (the lanes worked
    // on a permutation of the tile): written out here, every wave stores 64 consecutive reads per instruction -- whole
    // lines instead of 64 scattered words per store that leave L2 as partial lines.
    if (prev_tile != ~0ULL) {
      const uint64_t rp = prev_tile * ALIGN_BLOCK + tid;
#ifndef NIMBLE_SKEL_NOSTORE  // (experiments: the skeleton without its result stores)
      if (rp < n) {
        const uint64_t rs = col[nm * ALIGN_BLOCK];
#pragma unroll
        for (int m = 0; m < nm; ++m) {
          const uint64_t v = col[m * ALIGN_BLOCK];
          st_stream(&cb.reason[m][rp], (uint8_t)(rs >> (8 * m)));
          st_stream(&cb.score[m][rp], (uint32_t)(v & 0xFFFFu));
          st_stream(&cb.mism[m][rp], (uint32_t)((v >> 16) & 0xFFFFu));
          st_stream(&cb.cls[m][rp], (uint32_t)(v >> 32));
        }
      }
#endif
    }
#endif
    if (tid == 0) {
#ifdef NIMBLE_STATIC_TILES
      s_tile = prev_tile == ~0ULL ? (unsigned long long)blockIdx.x : prev_tile + gridDim.x;
      (void)tile_counter;
#else
      s_tile = (s_next & NO_TILE) == NO_TILE ? n_tiles : (s_next & NO_TILE);
#endif
    }
    __syncthreads();
    const uint64_t tile = s_tile;
    PROF(0)
    if (tile >= n_tiles) break;
    WCOUNT(0)
#ifndef NIMBLE_STATIC_TILES
    if (tid == 0) fetch_tile(false);  // (consumed at the head of the next round)
#endif
    prev_tile = tile;
    // ---- own slot: key into LDS column tid, first direct probe of mate 0
    const uint64_t r_own = tile * ALIGN_BLOCK + tid;
    uint32_t kind = 2;  // 0 = needs a seed scan, 1 = seed known, 2 = nothing to walk for mate 0
    uint64_t seedv = ~0ULL;
    uint64_t metav = 0;
    if (r_own < n) {
      // Everything the tile needs from memory is asked for at once: the key words do not wait for the lengths (what lies
      // behind a key is masked when the lengths are there).  A tile is a chain of dependent round trips -- tile index,
      // lengths, key words, first probe, the walk's records -- and a block does nothing else meanwhile: a launch over reads
      // that stop at the prefilter, which fetches and stores 70 bytes a read, took 0.5 of k_align's 1.2 ms.
      const uint32_t pre0 = rd_pre(cb, 0, r_own);
      const uint32_t k0 = rd_len(cb, 0, r_own);
      const uint32_t l1 = nm == 2 ? rd_len(cb, 1, r_own) : 0u;
      const uint32_t l0 = rd_alen(cb, 0, r_own);  // bases of mate 0 that are aligned
      metav = (uint64_t)l0 | ((uint64_t)k0 << 32) | ((uint64_t)pre0 << 48);
      if (nm == 2) metav |= ((uint64_t)rd_alen(cb, 1, r_own) << 16) | ((uint64_t)rd_pre(cb, 1, r_own) << 56);
      constexpr uint32_t KCH = 8;
      const uint64_t *row = cb.rec ? cb.rec + r_own * cb.rec_words : nullptr;
      for (uint32_t w0 = 0; w0 < kw; w0 += KCH) {
        uint64_t v[KCH];
#pragma unroll
        for (uint32_t j = 0; j < KCH; ++j)
#ifdef NIMBLE_SKEL_NOKEYS  // (experiments: the skeleton without its key loads)
          v[j] = 0ULL;
#else
          v[j] = w0 + j < kw ? (row ? row[w0 + j] : ld_stream(cb.keys + (uint64_t)(w0 + j) * cb.key_stride + r_own)) : 0ULL;
#endif
        const uint32_t nw = (k0 + l1 + 31u) >> 5;
#pragma unroll
        for (uint32_t j = 0; j < KCH; ++j)
          if (w0 + j < kw) col[(w0 + j) * ALIGN_BLOCK] = w0 + j < nw ? v[j] : 0ULL;
      }
      col[kw * ALIGN_BLOCK] = 0ULL;
      if (pre0 == R_TODO && l0 >= KMER) {
        uint32_t nd = 0, of = 0;
        ln.rd = col;
        if (probe_direct(ix, ln, 0u, 0u, nd, of, MODE != 1)) {  // (walk_fast counts the tile's probe of a mate itself)
          kind = 1;
          seedv = u64of(of, nd);
        } else {
          kind = 0;
        }
      }
    }
    // ---- partition the tile: reads that still need a scan first, then the seeded ones, then the rest, so
    // that the waves of the block are (nearly) homogeneous and most of them skip the scan rounds entirely
    PROF(1)
    s_seed[tid] = seedv;
    if (MODE == 1) s_meta[tid] = metav;
    {
      const uint64_t b0 = __ballot(kind == 0), b1 = __ballot(kind == 1);
      const uint32_t wv = tid >> 6, lane = tid & 63u;
      if (lane == 0) {
        s_cnt[wv * 2] = (uint32_t)__popcll(b0);
        s_cnt[wv * 2 + 1] = (uint32_t)__popcll(b1);
      }
      __syncthreads();
      uint32_t tot0 = 0, tot1 = 0, pre0 = 0, pre1 = 0;
#pragma unroll
      for (uint32_t w = 0; w < ALIGN_BLOCK / 64; ++w) {
        const uint32_t c0 = s_cnt[w * 2], c1 = s_cnt[w * 2 + 1];
        if (w < wv) { pre0 += c0; pre1 += c1; }
        tot0 += c0;
        tot1 += c1;
      }
      const uint64_t below = (1ULL << lane) - 1ULL;
      const uint32_t r0 = (uint32_t)__popcll(b0 & below), r1 = (uint32_t)__popcll(b1 & below);
      const uint32_t r2 = lane - r0 - r1;                       // rank among kind 2 in this wave
      const uint32_t pre2 = wv * 64u - pre0 - pre1;             // kind-2 lanes in earlier waves
      const uint32_t pos = kind == 0 ? pre0 + r0 : (kind == 1 ? tot0 + pre1 + r1 : tot0 + tot1 + pre2 + r2);
      s_perm[pos] = (uint16_t)tid;
      __syncthreads();
    }
    PROF(2)
    const uint32_t slot = s_perm[tid];
    const uint64_t r = tile * ALIGN_BLOCK + slot;
    const bool active = r < n;
    ln.rd = lds64 + slot;
    const uint64_t pre_seed = s_seed[slot];
    uint32_t L[2] = {0, 0};   // aligned bases per mate
    uint32_t mate1_at = 0;    // where mate 1 starts inside the key (the untrimmed length of mate 0)
    // (the general walk's launch has no room for this array in LDS at the longest reads it takes: it asks the call's arrays)
    uint64_t meta = 0;
    if (MODE == 1) {
      meta = s_meta[slot];
    } else if (active) {
      meta = (uint64_t)rd_alen(cb, 0, r) | ((uint64_t)rd_pre(cb, 0, r) << 48);
      if (nm == 2)
        meta |= ((uint64_t)rd_alen(cb, 1, r) << 16) | ((uint64_t)rd_len(cb, 0, r) << 32) | ((uint64_t)rd_pre(cb, 1, r) << 56);
    }
    if (active) {
      L[0] = (uint32_t)meta & 0xFFFFu;
      if (nm == 2) {
        L[1] = (uint32_t)(meta >> 16) & 0xFFFFu;
        mate1_at = (uint32_t)(meta >> 32) & 0xFFFFu;
      }
    }
    bool any_walk = false;
    uint64_t res_v[2] = {0, 0}, res_r = 0;  // (staged results: score | mismatches << 16 | class << 32 per mate; reasons)
    for (int m = 0; m < nm; ++m) {
      uint32_t reason = NIMBLE_R_NONE, score = 0, mm = 0, cls = CLS_NONE;
      uint32_t need = 0, best_col = 0, best_len = 0;
      uint64_t dhash = 0;
      MaskRes mres;
      mres.is_mask = false;
      mres.base = 0;
      mres.mask = 0;
      if (active) {
        const uint32_t pre = (uint32_t)(meta >> (48 + 8 * m)) & 0xFFu;
        if (pre != R_TODO) {
          reason = pre;
          if (m == 0) c_pre++;
        } else {
          uint32_t cov = 0, mis = 0;
          const uint32_t pre_state = m == 0 ? (pre_seed != ~0ULL ? 2u : 1u) : 0u;
          PROF(3)
          bool some;
          if (MODE == 1) {
            const uint32_t base0 = m ? mate1_at : 0u;
            some = p.num_mismatches == 0 ? walk_fast<true>(ix, ln, base0, L[m], 0u, pre_state, pre_seed, cov, mis PROF_PASS)
                                         : walk_fast<false>(ix, ln, base0, L[m], p.num_mismatches, pre_state, pre_seed, cov, mis PROF_PASS);
          } else {
            some = walk(ix, ln, m ? mate1_at : 0u, L[m], p.num_mismatches, cov, mis, pre_state, pre_seed PROF_PASS);
          }
          PROF(3)
          if (!some) {
            reason = NIMBLE_R_NO_MATCH;
          } else {
            any_walk = true;
            score = cov;
            mm = mis;
            uint32_t count;
            best_col = ln.min_col;
            best_len = ln.min_len;
            // the class is spelled out (hash, mask form) only when it is needed: a walk in the register window knows its
            // size from four popcounts, and a result of the size of the smallest visited class IS that class
            const bool windowed = WIDE && !ln.all_mask && ln.window_ok;
            // (fast launch: the class's size is a popcount; its first row -- one more line from the index -- is fetched only
            // when the class is not simply the smallest visited one, below)
            count = MODE == 1 ? (uint32_t)__popcll(ln.acc)
                              : (windowed ? window_count(ln) : finish_class<WIDE>(ix, ln, dhash, nullptr, mres));
            // `score as f64 / len as f64 >= score_percent` (align.rs:968, filter/align.rs:16) as an exact
            // integer test: min_cov[len] is the smallest score whose IEEE quotient reaches score_percent
            if (p.discard_nonzero_mismatch && mis != 0) {
              reason = NIMBLE_R_DISCARDED_NONZERO_MISMATCH;
            } else if ((uint64_t)cov >= p.score_threshold && cov >= cb.min_cov[L[m]] && count != 0) {
              if (p.discard_multiple_matches && count > 1) reason = NIMBLE_R_DISCARDED_MULTIPLE_MATCH;
              else if (mis > p.num_mismatches) reason = NIMBLE_R_ABOVE_MISMATCH_THRESHOLD;
              else {
                reason = NIMBLE_R_SUCCESSFUL_MATCH;
                if (count == best_len) {
                  cls = best_col;  // the intersection is the smallest colour itself
                } else {
                  // an intersection that is not one of the visited colours: find its canonical id in the
                  // content-addressed class table (exact compare: the mask form against the descriptor, a wider class
                  // in the register window against the stored ids); only a class never seen before goes through the
                  // claim / verify kernels
                  cls = CLS_PENDING;
                  if (MODE == 1) {
                    WCOUNT(13)
                    ln.fbase = ix.srec_base[ln.last_rec];
                    finish_class<WIDE>(ix, ln, dhash, nullptr, mres);
                  }
                  if (windowed) finish_class<WIDE>(ix, ln, dhash, nullptr, mres);
                  if (mres.is_mask || windowed) {
                    const uint32_t tag = intern_tag(dhash);
                    uint64_t pos = dhash & ix.intern_mask;
                    for (;;) {
                      const uint64_t slot = ix.intern[pos];
                      WCOUNT(14)
                      if (slot == 0) break;
                      const uint32_t id = (uint32_t)slot;
                      if ((uint32_t)(slot >> 32) == tag && id != INTERN_PENDING && id < ix.cls_cap) {
                        const uint4 d = ix.cls_desc[id];
                        if (mres.is_mask) {
                          if (d.x == (count | CLS_MASK_FLAG) && d.y == mres.base && desc_mask(d) == mres.mask) {
                            cls = id;
                            break;
                          }
                        } else if (d.x == count && window_equals_class(ix, ln, id, count)) {
                          cls = id;
                          break;
                        }
                      }
                      pos = (pos + 1) & ix.intern_mask;
                    }
                  }
                  if (cls == CLS_PENDING) need = count;
                }
              }
            } else {
              reason = NIMBLE_R_SCORE_BELOW_THRESHOLD;
            }
          }
        }
      }
      // wave-aggregated allocation of scratch space for classes that need interning (convergent point)
      PROF(6)
      uint32_t total;
      uint32_t ofs = wave_excl_scan(need, total);
      unsigned long long base = 0;
      if (total) {
        if ((tid & 63u) == 0) base = atomicAdd((unsigned long long *)&cb.state[8], (unsigned long long)total);
        base = __shfl(base, 0, 64);
      }
      if (need) {
        if (base + ofs + need <= (unsigned long long)cb.scratch_cap) {
          uint64_t hh;
          MaskRes m2;
          finish_class<WIDE>(ix, ln, hh, cb.scratch + base + ofs, m2);
          cb.dyn_off[m][r] = (uint32_t)base + ofs;
          cb.dyn_len[m][r] = need;
          cb.dyn_hash[m][r] = dhash;
        } else {
          atomicOr((unsigned long long *)&cb.state[10], (unsigned long long)ERR_SCRATCH);
          cls = CLS_NONE;
        }
      }
#if NIMBLE_RESULT_STAGED
      res_v[m] = (uint64_t)score | ((uint64_t)mm << 16) | ((uint64_t)cls << 32);  // (a mate has at most 65 535 bases)
      res_r |= (uint64_t)reason << (8 * m);
#else
      if (active) {
        st_stream(&cb.reason[m][r], (uint8_t)reason);
        st_stream(&cb.score[m][r], score);
        st_stream(&cb.mism[m][r], mm);
        st_stream(&cb.cls[m][r], cls);
      }
#endif
    }
#if NIMBLE_RESULT_STAGED
    if (active) {  // the lane's own column is free now: both mates are done with the key
      uint64_t *c = lds64 + slot;
#pragma unroll
      for (int m = 0; m < nm; ++m) c[m * ALIGN_BLOCK] = res_v[m];
      c[nm * ALIGN_BLOCK] = res_r;
    }
#endif
    if (any_walk) c_seeded++;
  }
#if NIMBLE_PROFILE_SECTIONS == 1
  if ((tid & 63u) == 0)
    for (int i = 0; i < 10; ++i) atomicAdd(&g_prof[i], prof_acc[i]);
#endif
  if (ln.overflow) atomicOr((unsigned long long *)&cb.state[10], (unsigned long long)ERR_SCRATCH);
  if (want_counters) {
    uint64_t v2 = wave_sum64(ln.probes), v3 = wave_sum64(ln.nodes), v4 = wave_sum64(ln.entries);
    uint64_t v5 = wave_sum64(c_seeded), v6 = wave_sum64(c_pre);
    if ((tid & 63u) == 0) {
      atomicAdd((unsigned long long *)&cb.state[2], (unsigned long long)v2);
      atomicAdd((unsigned long long *)&cb.state[3], (unsigned long long)v3);
      atomicAdd((unsigned long long *)&cb.state[4], (unsigned long long)v4);
      atomicAdd((unsigned long long *)&cb.state[5], (unsigned long long)v5);
      atomicAdd((unsigned long long *)&cb.state[6], (unsigned long long)v6);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// class interning: content-addressed table {tag | class id}; claims and verification are split by a
// kernel boundary so that no lane ever waits on another lane.
// ---------------------------------------------------------------------------------------------
__device__ void intern_claim_one(const DevIndex &ix, const CallBuffers &cb, int round, uint64_t i, int m) {
  if (cb.cls[m][i] != CLS_PENDING) return;
  const uint64_t h = cb.dyn_hash[m][i];
  const uint32_t tag = intern_tag(h);
  uint64_t pos = round == 0 ? (h & ix.intern_mask) : (((uint64_t)cb.dyn_pos[m][i] + 1) & ix.intern_mask);
  for (;;) {
    uint64_t cur = atomicCAS((unsigned long long *)&ix.intern[pos], 0ULL,
                             ((unsigned long long)tag << 32) | INTERN_PENDING);
    if (cur == 0) {
      // this lane owns the slot: append the class to the table and publish its id
      const uint32_t len = cb.dyn_len[m][i];
      uint32_t id = atomicAdd(&ix.dyn_state[0], 1u);
      uint32_t off = atomicAdd(&ix.dyn_state[1], len);
      if (id >= ix.cls_cap || (uint64_t)off + len > ix.ids_cap) {
        atomicOr((unsigned long long *)&cb.state[10],
                 (unsigned long long)(id >= ix.cls_cap ? ERR_CLASS_CAP : ERR_IDS_CAP));
        ix.intern[pos] = ((uint64_t)tag << 32) | 0u;  // keep the table consistent; the call reports the error
        cb.dyn_pos[m][i] = (uint32_t)pos;
        return;
      }
      const uint32_t *src = cb.scratch + cb.dyn_off[m][i];
      for (uint32_t t = 0; t < len; ++t) ix.cls_ids[off + t] = src[t];
      uint32_t desc[4];
      make_class_desc(src, len, desc);
      ix.cls_off[id] = off;
      ix.cls_desc[id] = make_uint4(desc[0], desc[1], desc[2], desc[3]);
      // the content is in memory before the id is: a lookup that sees the id (k_align, any stream) finds the class
      // behind it.  (Claim / verify kernels of different contexts never overlap, capi.cpp serialises them per index.)
      __threadfence();
      __hip_atomic_store(&ix.intern[pos], ((uint64_t)tag << 32) | id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cb.dyn_pos[m][i] = (uint32_t)pos;
      return;
    }
    if ((uint32_t)(cur >> 32) == tag) {  // candidate; content is checked after the kernel boundary
      cb.dyn_pos[m][i] = (uint32_t)pos;
      return;
    }
    pos = (pos + 1) & ix.intern_mask;
  }
}

// (a capped grid striding over the reads: in the usual case -- nothing pending -- a thousand workgroups leave at once
// instead of forty thousand)
__global__ void k_intern_claim(DevIndex ix, CallBuffers cb, int round) {
  if (cb.state[8] == 0) return;  // no class was parked in the scratch pool: nothing is pending (the usual case)
  const int m = (int)blockIdx.y;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cb.n; i += (uint64_t)gridDim.x * blockDim.x)
    intern_claim_one(ix, cb, round, i, m);
}

__device__ void intern_verify_one(const DevIndex &ix, const CallBuffers &cb, uint64_t i, int m) {
  if (cb.cls[m][i] != CLS_PENDING) return;
  const uint32_t id = (uint32_t)__hip_atomic_load(&ix.intern[cb.dyn_pos[m][i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (id == INTERN_PENDING) {
    // claimed but not yet published (only a claim running beside this kernel could leave that): not a collision --
    // the next round looks at the SAME slot again (it starts one past dyn_pos)
    cb.dyn_pos[m][i] = (uint32_t)(((uint64_t)cb.dyn_pos[m][i] + ix.intern_mask) & ix.intern_mask);
    atomicAdd((unsigned long long *)&cb.state[9], 1ULL);
    return;
  }
  const uint32_t len = cb.dyn_len[m][i];
  uint4 d0 = make_uint4(0, 0, 0, 0);
  if (id < ix.cls_cap) d0 = ix.cls_desc[id];
  bool same = id < ix.cls_cap && desc_len(d0) == len;
  if (same) {
    const uint32_t *a = cb.scratch + cb.dyn_off[m][i];
    const uint32_t *b = ix.cls_ids + ix.cls_off[id];
    for (uint32_t t = 0; t < len; ++t)
      if (a[t] != b[t]) { same = false; break; }
  }
  if (same) cb.cls[m][i] = id;
  else atomicAdd((unsigned long long *)&cb.state[9], 1ULL);  // tag collision: next round probes further
}

__global__ void k_intern_verify(DevIndex ix, CallBuffers cb) {
  if (cb.state[8] == 0) return;
  const int m = (int)blockIdx.y;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cb.n; i += (uint64_t)gridDim.x * blockDim.x)
    intern_verify_one(ix, cb, i, m);
}

// one unique read key adds 1 to the (segment, class R1, class R2) histogram (open addressing over u64 keys)
__device__ __forceinline__ uint64_t hist_key(const CallBuffers &cb, uint32_t seg, uint32_t c1, uint32_t c2) {
  if (cb.cls_bits == 0) return ((uint64_t)c1 << 32) | c2;
  const uint32_t b = cb.cls_bits;  // CLS_NONE + 1 wraps to 0
  return ((uint64_t)seg << (2u * b)) | ((uint64_t)(uint32_t)(c1 + 1u) << b) | (uint64_t)(uint32_t)(c2 + 1u);
}

__device__ __forceinline__ void hist_add_n(const CallBuffers &cb, uint64_t key, uint64_t count, uint32_t read) {
  uint64_t pos = mix64(key) & cb.hist_mask;
  for (uint64_t probes = 0; probes <= cb.hist_mask; ++probes) {
    uint64_t cur = cb.hist_keys[pos];  // almost always already present: skip the CAS
    if (cur != key)
      cur = atomicCAS((unsigned long long *)&cb.hist_keys[pos], (unsigned long long)HIST_EMPTY, (unsigned long long)key);
    if (cur == HIST_EMPTY || cur == key) {
      atomicAdd((unsigned long long *)&cb.hist_cnt[pos], (unsigned long long)count);
      if (cb.hist_rep) atomicMax(&cb.hist_rep[pos], read);
      return;
    }
    pos = (pos + 1) & cb.hist_mask;
  }
  atomicOr((unsigned long long *)&cb.state[10], (unsigned long long)ERR_HIST);
}

__device__ __forceinline__ void hist_add(const CallBuffers &cb, uint32_t seg, uint32_t c1, uint32_t c2, uint32_t read) {
  hist_add_n(cb, hist_key(cb, seg, c1, c2), 1ULL, read);
}

// Per-block counters of the callsets a block meets first (k_dedup): a feature that draws a large share of the reads
// would otherwise take one global atomic per read on ONE counter (30 % of the reads on one feature: dedup 1.0 instead
// of 0.55 ms).  A block adds in LDS and hands each counter over once, at its end.  Entries are claimed once.
constexpr uint32_t HCLS_ENTRIES = 64;
struct HotCls {
  uint32_t state, pad;
  uint64_t key;
  unsigned long long count;
};
__device__ __forceinline__ void hist_add_block(const CallBuffers &cb, HotCls *hcls, uint32_t seg, uint32_t c1,
                                               uint32_t c2, uint32_t read) {
  const uint64_t key = hist_key(cb, seg, c1, c2);
  if (hcls) {
    volatile HotCls *e = hcls + (mix64(key) & (uint64_t)(HCLS_ENTRIES - 1));
    uint32_t st = e->state;
    if (st == 0u && atomicCAS((uint32_t *)&e->state, 0u, 1u) == 0u) {
      e->key = key;
      e->count = 1ULL;
      __threadfence_block();
      e->state = 2u;
      return;
    }
    if (st == 2u && e->key == key) {
      atomicAdd((unsigned long long *)&e->count, 1ULL);
      return;
    }
  }
  hist_add_n(cb, key, 1ULL, read);
}

// ---------------------------------------------------------------------------------------------
// k_dedup: pair filter + insert of the read key into the call's dedup table (last writer wins)
// ---------------------------------------------------------------------------------------------
// Per-block LDS cache of dominant keys (see dedup_one): hash, key words, the slot of the key and an index known to
// represent it.  Entries are claimed once (state 0 -> 1 -> 2) and never replaced, so a reader never sees a mix of two.
constexpr uint32_t HC_ENTRIES = 64, HC_WORDS = 10;
struct HotEntry {
  uint32_t state, j, pos, total;
  uint64_t h;
  uint32_t seg, pad;
  uint64_t key[HC_WORDS];
};

// mode: 0 = plain, 1 = the sample launch (plain, but duplicates are counted and leave their hash in the set),
//       2 = the input has dominant keys (the sample found duplicates): look before the atomic, LDS cache
__device__ __forceinline__ void dedup_one(const nimble_align_params &p, const CallBuffers &cb, uint64_t i,
                                          HotEntry *hc, HotCls *hcls, int mode) {
  uint32_t c1 = cb.cls[0][i];
  uint32_t c2 = cb.paired ? cb.cls[1][i] : CLS_NONE;
  cb.counted[i] = 0;
  if (cb.paired && p.require_valid_pair) {
    // filter_pair: keep only when both classes are non-empty and identical (ids are canonical)
    if (!(c1 != CLS_NONE && c2 != CLS_NONE && c1 == c2)) {
      cb.reason[0][i] = NIMBLE_R_NOT_MATCHING_PAIR;
      cb.reason[1][i] = NIMBLE_R_NOT_MATCHING_PAIR;
      cb.slot[i] = SLOT_NONE;
      return;
    }
  }
  if (c1 == CLS_NONE && c2 == CLS_NONE) {
    cb.slot[i] = SLOT_NONE;
    return;
  }
  // the dedup scope is the segment (one UMI = one score::call): it is part of the key
  const uint32_t seg = cb.seg ? cb.seg[i] : 0u;
  const uint64_t h0 = rd_hash(cb, i);
  const uint64_t h = cb.seg ? mix64(h0 ^ ((uint64_t)seg * 0x9E3779B97F4A7C15ULL)) : h0;
  const uint32_t tag = (uint32_t)(h >> 32) | 1u;
  const uint64_t mine = ((uint64_t)tag << 32) | (uint32_t)i;
  const uint32_t total = rd_len(cb, 0, i) + (cb.paired ? rd_len(cb, 1, i) : 0u);
  const uint32_t nw = (total + 31u) >> 5;
  uint32_t pos = __umulhi((uint32_t)h, cb.dedup_slots);  // low hash half picks the slot, high half is the tag
  // A key with very many copies (a dominant transcript, an adapter dimer) would put one atomic per copy on ONE
  // address, and same-address atomics run at ~90 M/s: 5 % copies of one read took 11 ms here.  Every key found a
  // second time leaves its hash in a small set; a read whose hash is in the set LOOKS at a slot (plain load) before
  // it touches it with an atomic, and when it finds its key already represented by a later read (reads are taken
  // from the end of the input backwards, so that is the usual case) it is done without any atomic.  Both kinds of
  // read share one probe loop, so that a wave with a few such lanes does not run two loops one after the other.
  const uint64_t hot_slot = h & (uint64_t)(HOT_KEYS - 1);
  // ... and what such a read needs to know sits in LDS once a lane of this block has been through memory for it: the
  // loads of half a million copies otherwise queue at the ONE L2 channel that holds the set entry, the slot and the
  // representative (~4 ns each: 5 % copies of one read still cost 2 ms with the atomics out of the way).  Every read
  // looks there first (one LDS read), before the set in global memory.
  volatile HotEntry *he = hc ? hc + (h & (uint64_t)(HC_ENTRIES - 1)) : nullptr;
  if (he && nw <= HC_WORDS && he->state == 2u && he->h == h && he->total == total && he->seg == seg && he->j > i) {
    uint64_t diff = 0;
    for (uint32_t w = 0; w < nw; ++w) diff |= rd_key(cb, w, i) ^ he->key[w];
    if (diff == 0) {
      cb.slot[i] = he->pos;
      return;
    }
  }
  bool look = mode == 2 && cb.hot[hot_slot] == h;
  for (;;) {
    uint64_t cur;
    if (look) cur = cb.dedup[pos];   // an ordinary cached load; a stale value only costs an atomic further down
    else cur = atomicCAS((unsigned long long *)&cb.dedup[pos], 0ULL, (unsigned long long)mine);
    if (cur == 0) {
      if (look) {  // empty as far as this CU can see: claim it atomically
        look = false;
        continue;
      }
      // first copy of this key.  When the classes are a function of the key alone (single-end, or mates of one
      // fixed length) any copy may stand for the key in the histogram, so count it here and skip k_count.
      if (cb.fuse_count) hist_add_block(cb, hcls, seg, c1, c2, (uint32_t)i);
      break;
    }
    if ((uint32_t)(cur >> 32) == tag) {
      const uint64_t j = (uint32_t)cur;
      // all words at once (no early exit): the loads go out back to back, one latency instead of one per word
      uint64_t diff = (uint64_t)((rd_len(cb, 0, j) + (cb.paired ? rd_len(cb, 1, j) : 0u)) ^ total);
      if (cb.seg) diff |= (uint64_t)(cb.seg[j] ^ seg);
      for (uint32_t w = 0; w < nw; ++w) diff |= rd_key(cb, w, i) ^ rd_key(cb, w, j);
      if (diff == 0) {
        if (mode == 1) {
          // the sample: a second copy marks the key in the set; a copy that finds it marked already is a third or later
          // one, and state[13] counts those -- what tells a key with MANY copies, which the main launch has to be
          // careful with, from an input that merely holds pairs of equal reads (5 % of the bench reads are such pairs:
          // taken for dominant keys they cost the main launch 5 % for nothing)
          const unsigned long long was = atomicExch((unsigned long long *)&cb.hot[hot_slot], (unsigned long long)h);
          if (was == h) atomicAdd((unsigned long long *)&cb.state[13], 1ULL);
          atomicAdd((unsigned long long *)&cb.state[15], 1ULL);
        } else if (mode != 0 && !look) {
          cb.hot[hot_slot] = h;
        }
        if (j < i) atomicMax((unsigned long long *)&cb.dedup[pos], (unsigned long long)mine);
        else if (look && he && nw <= HC_WORDS && atomicCAS((uint32_t *)&he->state, 0u, 1u) == 0u) {
          he->h = h;
          he->total = total;
          he->seg = seg;
          he->j = (uint32_t)j;
          he->pos = pos;
          for (uint32_t w = 0; w < nw; ++w) he->key[w] = rd_key(cb, w, i);
          __threadfence_block();
          he->state = 2u;
        }
        break;
      }
    }
    pos = pos + 1 == cb.dedup_slots ? 0u : pos + 1;
  }
  cb.slot[i] = (uint32_t)pos;
}

__global__ __launch_bounds__(256) void k_dedup(nimble_align_params p, CallBuffers cb, uint64_t g_begin,
                                               uint64_t g_end, int block_counters) {
  __shared__ HotEntry s_hot[HC_ENTRIES];
  for (uint32_t e = threadIdx.x; e < HC_ENTRIES; e += blockDim.x) s_hot[e].state = 0u;
  __syncthreads();
  const uint64_t n = cb.n;
  // The sample launch (the last reads, g_begin == 0 of a split launch) counts the duplicates it meets (state[15]) and,
  // apart, the third and later copies of a key (state[13]); only when it met some of THOSE does the main launch pay for
  // the look-before-atomic machinery.
  const bool is_sample = cb.hot && g_begin == 0 && g_end < n;
  const int mode = !cb.hot ? 0 : is_sample ? 1 : (g_begin != 0 && cb.state[13] >= 4 ? 2 : 0);
  HotEntry *hc = mode == 2 ? s_hot : nullptr;
  // per-block callset counters: plain calls only (a representative read per entry, BAM mode, needs the global max)
  __shared__ HotCls s_cls[HCLS_ENTRIES];
  HotCls *hcls = (block_counters && cb.fuse_count && !cb.hist_rep && !is_sample) ? s_cls : nullptr;
  for (uint32_t e = threadIdx.x; e < HCLS_ENTRIES; e += blockDim.x) s_cls[e].state = 0u;
  __syncthreads();
  // grid-stride: the launch may use a small grid (the kernel is bound by the chip's atomic rate, which 64 workgroups
  // already reach, and then leaves the other CUs to the next call's kernels)
  // from the last read backwards: the representative of a key is its LAST copy (score_map.insert overwrites,
  // src/align.rs:685), so the copies that come later in this order find a slot that already holds a larger index
  for (uint64_t g = g_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < g_end;
       g += (uint64_t)gridDim.x * blockDim.x)
    dedup_one(p, cb, n - 1 - g, hc, hcls, mode);
  if (hcls) {
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < HCLS_ENTRIES; e += blockDim.x)
      if (s_cls[e].state == 2u) hist_add_n(cb, s_cls[e].key, s_cls[e].count, 0u);
  }
}

// k_count: the representative of each key adds one to the (class R1, class R2) histogram.  When k_dedup has
// already counted (fuse_count), only the `counted` flags are set here, and only when somebody asks for them.
__global__ void k_count(CallBuffers cb) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cb.n) return;
  const uint32_t s = cb.slot[i];
  if (s == SLOT_NONE) return;
  if ((uint32_t)cb.dedup[s] != (uint32_t)i) return;
  cb.counted[i] = 1;
  if (cb.fuse_count) return;
  hist_add(cb, cb.seg ? cb.seg[i] : 0u, cb.cls[0][i], cb.paired ? cb.cls[1][i] : CLS_NONE, (uint32_t)i);
}

__global__ void k_hist_compact(CallBuffers cb, uint32_t *c1, uint32_t *c2, uint64_t *cnt, uint64_t cap, uint32_t *seg,
                               uint32_t *rep) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > cb.hist_mask) return;
  const uint64_t key = cb.hist_keys[i];
  if (key == HIST_EMPTY) return;
  uint64_t o = atomicAdd((unsigned long long *)&cb.state[11], 1ULL);
  if (o < cap) {
    if (cb.cls_bits == 0) {
      c1[o] = (uint32_t)(key >> 32);
      c2[o] = (uint32_t)key;
      if (seg) seg[o] = 0;
    } else {
      const uint32_t b = cb.cls_bits;
      const uint64_t m = (1ULL << b) - 1ULL;
      c1[o] = (uint32_t)((key >> b) & m) - 1u;  // 0 wraps back to CLS_NONE
      c2[o] = (uint32_t)(key & m) - 1u;
      if (seg) seg[o] = (uint32_t)(key >> (2u * b));
    }
    cnt[o] = cb.hist_cnt[i];
    if (rep) rep[o] = cb.hist_rep ? cb.hist_rep[i] : 0u;
  }
}

// trim_sequence / maxinfo (align.rs:866-942): the number of leading bases that are aligned, from the quality
// string.  Integer scores from two host-built tables; the running best is compared as f64, as the reference does.
__global__ void k_maxinfo(const uint8_t *__restrict__ qual, const uint64_t *__restrict__ off, uint32_t fixed_len,
                          uint64_t n, const int64_t *__restrict__ length_scores, const int64_t *__restrict__ qual_probs,
                          uint32_t *__restrict__ out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t s = off ? off[r] : r * fixed_len;
  const uint32_t len = off ? (uint32_t)(off[r + 1] - s) : fixed_len;
  const uint8_t *q = qual + s;
  int64_t accum = 0;
  double max_score = -1.7976931348623157e308;  // f64::MIN
  uint32_t pos = 0;
  for (uint32_t i = 0; i < len; ++i) {
    uint32_t c = q[i];
    c = c > 60u ? 60u : c;                     // MAXQUAL; the raw byte, not Phred (align.rs:905-906)
    accum += qual_probs[c];
    const int64_t score = (i < 1000u ? length_scores[i] : 0) + accum;
    const double sf = (double)score;
    if (sf >= max_score) {
      max_score = sf;
      pos = i + 1;
    }
  }
  uint32_t t;
  if (pos < 1 || max_score == 0.0) t = 0;
  else t = pos < len ? pos : len;
  out[r] = t;
}

__global__ void k_hist_dense_se(CallBuffers cb, int64_t *counts, uint32_t n_classes) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > cb.hist_mask) return;
  const uint64_t key = cb.hist_keys[i];
  if (key == HIST_EMPTY) return;
  const uint32_t c1 = (uint32_t)(key >> 32), c2 = (uint32_t)key;
  if (c2 == CLS_NONE && c1 < n_classes) counts[c1] = (int64_t)cb.hist_cnt[i];
}

// ---- exchange helpers (multi-GPU): reads routed by key hash as fixed-width records ---------------------
// record = [key words ..., key hash, len0 | len1 << 16 | pre0 << 32 | pre1 << 40]  (key_words + 2 u64)
__device__ __forceinline__ uint32_t route_of(uint64_t h, uint32_t world) {
  return (uint32_t)((h & 0x7FFFFFFFFFFFFFFFULL) % world);
}

// Two passes over a fixed block -> reads mapping (block b owns the tiles b, b + G, b + 2G, ... of 256 reads), no
// global atomics: pass 1 leaves per-block counts, a one-block scan turns them into the first record slot of every
// (block, destination), pass 2 hands out slots from LDS cursors.  The layout is deterministic.
constexpr uint32_t ROUTE_GRID = 2048;

__global__ void k_route_count(const uint64_t *__restrict__ hash, uint64_t n, uint32_t world, uint32_t *block_counts) {
  __shared__ uint32_t s_c[256];
  if (threadIdx.x < world) s_c[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  for (uint64_t tile = blockIdx.x; tile * 256 < n; tile += gridDim.x) {
    const uint64_t i = tile * 256 + threadIdx.x;
    const uint32_t d = i < n ? route_of(hash[i], world) : 0xFFFFFFFFu;
    for (uint32_t t = 0; t < world; ++t) {
      const uint64_t m = __ballot(d == t);
      if (lane == 0 && m) atomicAdd(&s_c[t], (uint32_t)__popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x < world) block_counts[(uint64_t)blockIdx.x * world + threadIdx.x] = s_c[threadIdx.x];
}

// one block of 256 threads: totals[t] and block_first[b][t] = first record slot of (block b, destination t).
// Thread j owns the n_blocks / 256 consecutive blocks starting at j * per; a block-wide scan joins the chunks.
__global__ void k_route_scan(const uint32_t *__restrict__ block_counts, uint32_t n_blocks, uint32_t world,
                             uint64_t *totals, uint64_t *block_first) {
  __shared__ uint64_t s_part[256];
  __shared__ uint64_t s_start;
  const uint32_t j = threadIdx.x;
  const uint32_t per = (n_blocks + 255u) / 256u;
  if (j == 0) s_start = 0;
  __syncthreads();
  for (uint32_t t = 0; t < world; ++t) {
    uint64_t mine = 0;
    for (uint32_t k = 0; k < per; ++k) {
      const uint32_t b = j * per + k;
      if (b < n_blocks) mine += block_counts[(uint64_t)b * world + t];
    }
    s_part[j] = mine;
    __syncthreads();
    // inclusive scan over the 256 partial sums (Hillis-Steele; 8 steps)
    for (uint32_t d = 1; d < 256; d <<= 1) {
      const uint64_t v = j >= d ? s_part[j - d] : 0;
      __syncthreads();
      s_part[j] += v;
      __syncthreads();
    }
    uint64_t run = s_start + s_part[j] - mine;  // first slot of this thread's chunk
    for (uint32_t k = 0; k < per; ++k) {
      const uint32_t b = j * per + k;
      if (b < n_blocks) {
        block_first[(uint64_t)b * world + t] = run;
        run += block_counts[(uint64_t)b * world + t];
      }
    }
    __syncthreads();
    if (j == 255) {
      totals[t] = s_part[255];
      s_start += s_part[255];
    }
    __syncthreads();
  }
}

__global__ void k_route_scatter(CallBuffers cb, uint32_t world, const uint64_t *__restrict__ block_first,
                                uint64_t *__restrict__ rec, uint32_t *__restrict__ perm) {
  __shared__ unsigned long long s_cur[256];
  if (threadIdx.x < world) s_cur[threadIdx.x] = block_first[(uint64_t)blockIdx.x * world + threadIdx.x];
  __syncthreads();
  const uint32_t kw = cb.key_words;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint64_t tile = blockIdx.x; tile * 256 < cb.n; tile += gridDim.x) {
    const uint64_t i = tile * 256 + threadIdx.x;
    const bool live = i < cb.n;
    const uint64_t h = live ? cb.key_hash[i] : 0;
    const uint32_t d = live ? route_of(h, world) : 0xFFFFFFFFu;
    uint64_t slot = 0;
    for (uint32_t t = 0; t < world; ++t) {
      const uint64_t m = __ballot(d == t);
      if (m == 0) continue;
      const uint32_t leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(&s_cur[t], (unsigned long long)__popcll(m));
      base = __shfl(base, (int)leader, 64);
      if (d == t) slot = base + (uint64_t)__popcll(m & ((1ULL << lane) - 1ULL));
    }
    if (live) {
      uint64_t *o = rec + slot * (kw + 2);
      for (uint32_t w = 0; w < kw; ++w) o[w] = cb.keys[(uint64_t)w * cb.key_stride + i];
      o[kw] = h;
      o[kw + 1] = (uint64_t)cb.len[0][i] | ((uint64_t)(cb.paired ? cb.len[1][i] : 0u) << 16) |
                  ((uint64_t)cb.pre[0][i] << 32) | ((uint64_t)(cb.paired ? cb.pre[1][i] : 0u) << 40);
      if (perm) perm[i] = (uint32_t)slot;
    }
  }
}

// The same scatter with the records of a wave gathered in LDS first, ordered by destination: every (wave,
// destination) run is contiguous in LDS and in `rec`, so the write-out goes in runs of whole records instead of one
// 8-byte word per lane per instruction into 64 different records.  Wave-synchronous: no block barrier in the loop.
__global__ __launch_bounds__(256) void k_route_scatter_lds(CallBuffers cb, uint32_t world,
                                                           const uint64_t *__restrict__ block_first,
                                                           uint64_t *__restrict__ rec, uint32_t *__restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_words[];  // [4 waves][64 records][rw]
  __shared__ unsigned long long s_cur[256];
  __shared__ uint32_t s_slot[256];  // [4 waves][64]: global slot of the record at each local position
  if (threadIdx.x < world) s_cur[threadIdx.x] = block_first[(uint64_t)blockIdx.x * world + threadIdx.x];
  __syncthreads();
  const uint32_t kw = cb.key_words, rw = kw + 2;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint64_t *sw = s_words + (size_t)wave * 64u * rw;
  uint32_t *ss = s_slot + wave * 64u;
  const uint32_t q64 = 64u / rw, r64 = 64u % rw;
  for (uint64_t tile = blockIdx.x; tile * 256 < cb.n; tile += gridDim.x) {
    const uint64_t i = tile * 256 + threadIdx.x;
    const bool live = i < cb.n;
    const uint64_t h = live ? cb.key_hash[i] : 0;
    const uint32_t d = live ? route_of(h, world) : 0xFFFFFFFFu;
    uint32_t pos = 0, before = 0, n_live = 0;
    uint64_t slot = 0;
    for (uint32_t t = 0; t < world; ++t) {
      const uint64_t m = __ballot(d == t);
      if (m == 0) continue;
      const uint32_t cnt = (uint32_t)__popcll(m);
      const uint32_t leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(&s_cur[t], (unsigned long long)cnt);
      base = __shfl(base, (int)leader, 64);
      if (d == t) {
        const uint32_t rank = (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
        pos = before + rank;
        slot = base + rank;
      }
      before += cnt;
    }
    n_live = before;
    if (live) {
      uint64_t *o = sw + (size_t)pos * rw;
      for (uint32_t w = 0; w < kw; ++w) o[w] = cb.keys[(uint64_t)w * cb.key_stride + i];
      o[kw] = h;
      o[kw + 1] = (uint64_t)cb.len[0][i] | ((uint64_t)(cb.paired ? cb.len[1][i] : 0u) << 16) |
                  ((uint64_t)cb.pre[0][i] << 32) | ((uint64_t)(cb.paired ? cb.pre[1][i] : 0u) << 40);
      ss[pos] = (uint32_t)slot;
      if (perm) perm[i] = (uint32_t)slot;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes have landed
    // write-out: linear index x over the wave's n_live * rw words; (p, w) = (x / rw, x % rw) kept incrementally
    uint32_t pp = lane / rw, ww = lane % rw;
    const uint32_t total = n_live * rw;
    for (uint32_t x = lane; x < total; x += 64u) {
      rec[(uint64_t)ss[pp] * rw + ww] = sw[x];
      pp += q64;
      ww += r64;
      if (ww >= rw) {
        ww -= rw;
        ++pp;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the LDS reads are done before the next tile overwrites the staging
  }
}

__global__ void k_records_unpack(const uint64_t *__restrict__ rec, CallBuffers cb) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cb.n) return;
  const uint32_t kw = cb.key_words;
  const uint64_t *o = rec + i * (kw + 2);
  for (uint32_t w = 0; w < kw; ++w) cb.keys[(uint64_t)w * cb.key_stride + i] = o[w];
  cb.key_hash[i] = o[kw];
  const uint64_t meta = o[kw + 1];
  cb.len[0][i] = (uint32_t)(meta & 0xFFFF);
  cb.pre[0][i] = (uint8_t)((meta >> 32) & 0xFF);
  if (cb.paired) {
    cb.len[1][i] = (uint32_t)((meta >> 16) & 0xFFFF);
    cb.pre[1][i] = (uint8_t)((meta >> 40) & 0xFF);
  }
}


// ---------------------------------------------------------------------------------------------
// Multi-GPU, align-where-the-reads-are form.  The owner of a key (hash mod world) sees every copy of it as an
// exchange record and answers one byte per record: 1 = this copy stands for the key.  The classes stay on the
// rank that aligned the read; k_count_verdicts adds the chosen copies to that rank's histogram.
// ---------------------------------------------------------------------------------------------
template <bool STAGED>
__global__ void k_dedup_records(const uint64_t *__restrict__ rec, uint64_t n, uint32_t kw, uint64_t *table,
                                uint32_t slots, uint8_t *__restrict__ verdict) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_rec[];
  const uint32_t rw = kw + 2;
  const uint64_t first_rec = (uint64_t)blockIdx.x * blockDim.x;
  const uint64_t j = first_rec + threadIdx.x;
  const uint64_t *o = rec + j * rw;
  if (STAGED) {
    // the block's records are one contiguous run: load it with full lines, then every lane reads its own row from LDS
    const uint64_t live = n - first_rec < blockDim.x ? n - first_rec : blockDim.x;
    const uint64_t *src = rec + first_rec * rw;
    for (uint64_t w = threadIdx.x; w < live * rw; w += blockDim.x) s_rec[w] = src[w];
    __syncthreads();
    o = s_rec + (uint64_t)threadIdx.x * rw;
  }
  if (j >= n) return;
  const uint64_t h = o[kw];
  const uint64_t meta = o[kw + 1];
  const uint32_t total = (uint32_t)(meta & 0xFFFF) + (uint32_t)((meta >> 16) & 0xFFFF);
  const uint32_t nw = (total + 31u) >> 5;
  const uint32_t tag = (uint32_t)(h >> 32) | 1u;
  const uint64_t mine = ((uint64_t)tag << 32) | (uint32_t)j;
  uint32_t pos = __umulhi((uint32_t)h, slots);
  uint8_t first = 0;
  bool look = true;  // look before the atomic: the copies of a dominant key must not all CAS its one slot (k_dedup)
  for (;;) {
    uint64_t cur;
    if (look) cur = table[pos];
    else cur = atomicCAS((unsigned long long *)&table[pos], 0ULL, (unsigned long long)mine);
    if (cur == 0) {
      if (look) {
        look = false;
        continue;
      }
      first = 1;
      break;
    }
    look = true;
    if ((uint32_t)(cur >> 32) == tag) {
      const uint64_t *q = rec + (uint64_t)(uint32_t)cur * rw;
      const uint64_t m2 = q[kw + 1];
      bool same = ((uint32_t)(m2 & 0xFFFF) + (uint32_t)((m2 >> 16) & 0xFFFF)) == total;
      for (uint32_t w = 0; same && w < nw; ++w) same = q[w] == o[w];
      if (same) break;
    }
    pos = pos + 1 == slots ? 0u : pos + 1;
  }
  verdict[j] = first;
}

// The verdict of read i sits at its record slot perm[i] (the order k_route_scatter put the records in).  Same pair
// filter as k_dedup; classes are a function of the key here (single-end or fixed-length mates), so whichever copy
// the owner picked may stand for the key.
__global__ void k_count_verdicts(nimble_align_params p, CallBuffers cb, const uint32_t *__restrict__ perm,
                                 const uint8_t *__restrict__ verdict) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cb.n) return;
  const uint32_t c1 = cb.cls[0][i];
  const uint32_t c2 = cb.paired ? cb.cls[1][i] : CLS_NONE;
  uint8_t counted = 0;
  bool live = true;
  if (cb.paired && p.require_valid_pair) {
    if (!(c1 != CLS_NONE && c2 != CLS_NONE && c1 == c2)) {
      cb.reason[0][i] = NIMBLE_R_NOT_MATCHING_PAIR;
      cb.reason[1][i] = NIMBLE_R_NOT_MATCHING_PAIR;
      live = false;
    }
  }
  if (live && !(c1 == CLS_NONE && c2 == CLS_NONE) && verdict[perm[i]]) {
    counted = 1;
    hist_add(cb, 0u, c1, c2, (uint32_t)i);
  }
  cb.counted[i] = counted;
}

// head of a call: the histogram table, the state words and the hot-key set cleared by one launch (five separate
// fills cost a launch each)
__global__ void k_clear_call(CallBuffers cb, int clear_latch) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= cb.hist_mask) {
    cb.hist_keys[i] = HIST_EMPTY;
    cb.hist_cnt[i] = 0;
    if (cb.hist_rep) cb.hist_rep[i] = 0;
  }
  // [14] latches input errors found by k_pack until the host has reported them: it goes with the call whose pack set it
  if (i < 16 && (i != 14 || clear_latch)) cb.state[i] = 0;
  if (cb.hot && i < HOT_KEYS) cb.hot[i] = 0;
}

// the 16 state words of a call, to page-locked host memory (read by the host after the call's last event)
__global__ void k_publish_state(const uint64_t *__restrict__ state, uint64_t *__restrict__ host) {
  if (threadIdx.x < 16) host[threadIdx.x] = state[threadIdx.x];
}

__global__ void k_fill_u64(uint64_t *p, uint64_t v, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

inline uint32_t blocks_for(uint64_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

}  // namespace

int debug_sections(uint64_t out[16], int reset) {
#if NIMBLE_PROFILE_SECTIONS
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 16; ++i) out[i] = h[i];
  if (reset) {
    for (int i = 0; i < 16; ++i) h[i] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), h, sizeof(h)) != hipSuccess) return -1;
  }
  return 1;
#else
  (void)out;
  (void)reset;
  return 0;
#endif
}

uint32_t align_ws_lanes() { return (uint32_t)ALIGN_GRID * ALIGN_BLOCK; }
uint32_t align_lds_cols() { return LDS_COLS; }

void launch_pack(hipStream_t s, const uint8_t *r1, const uint64_t *off1, const uint8_t *r2, const uint64_t *off2,
                 uint32_t fixed_len, uint32_t max_len, uint32_t min_len, const double *plog, uint32_t plog_max_len,
                 const CallBuffers &cb) {
  (void)plog_max_len;
  if (cb.n == 0) return;
  // reads per block: as many as fit a 64 KiB LDS budget (both mates), at most one per lane
  const uint32_t budget = 64 * 1024;
  const uint32_t nm = cb.paired ? 2 : 1;
  uint32_t rpb = PACK_BLOCK;
  while (rpb > 1 && (uint64_t)nm * ((uint64_t)rpb * max_len + 48) > budget) rpb >>= 1;
  uint32_t tile_bytes = (uint32_t)((((uint64_t)rpb * max_len + 32) + 15) & ~15ULL);
  uint32_t grid = blocks_for(cb.n, rpb);
  hipLaunchKernelGGL(k_pack, dim3(grid), dim3(PACK_BLOCK), nm * tile_bytes, s, r1, off1, r2, off2, fixed_len, max_len,
                     rpb, tile_bytes, min_len, plog, cb);
}

void launch_pack_words(hipStream_t s, const uint64_t *w1, const uint32_t *len1, uint32_t stride1, const uint64_t *w2,
                       const uint32_t *len2, uint32_t stride2, uint32_t max_len, uint32_t min_len, const double *plog,
                       const CallBuffers &cb) {
  if (cb.n == 0) return;
  hipLaunchKernelGGL(k_pack_words, dim3(blocks_for(cb.n, 256)), dim3(256), 0, s, w1, len1, stride1, w2, len2, stride2, max_len,
                     min_len, plog, cb);
}

// resident blocks per CU of an align kernel, and the device's CU count
static int resident_blocks(const void *fn, size_t lds, int *cus_out) {
  int per_cu = 0, dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, ALIGN_BLOCK, lds) != hipSuccess || per_cu < 1) per_cu = 4;
  if (cus < 1) cus = 256;
  *cus_out = cus;
  return per_cu;
}

// mode: 0 = the general walk, 1 = the fast walk (indexes with stretch records)
static const void *align_kernel(bool paired, bool counters, bool wide, int mode) {
#define NIMBLE_PICK(P, C)                                    \
  (mode == 1 ? (const void *)k_align<P, C, false, 1>          \
             : (wide ? (const void *)k_align<P, C, true, 0> : (const void *)k_align<P, C, false, 0>))
  return paired ? (counters ? NIMBLE_PICK(true, true) : NIMBLE_PICK(true, false))
                : (counters ? NIMBLE_PICK(false, true) : NIMBLE_PICK(false, false));
#undef NIMBLE_PICK
}

// one launch of an align kernel: a persistent grid of exactly as many blocks as are resident at once (a larger grid
// would run a second, nearly empty round), tiles handed out through a counter; `want_blocks` caps the grid (0 = no cap)
static void launch_align_kernel(hipStream_t s, const void *fn, size_t lds, const DevIndex &ix, const nimble_align_params &p,
                                const CallBuffers &cb, uint64_t tiles, int grid_pct, int n_cus, uint64_t want_blocks) {
  // Residency and the opt-in to more dynamic LDS than the default limit (long reads) are per device and per kernel:
  // rank threads of a sharded call launch concurrently.
  int dev = 0;
  (void)hipGetDevice(&dev);
  static std::mutex mu;
  static std::map<std::tuple<int, const void *, size_t>, std::pair<int, int>> resident_cache;  // -> blocks per CU, CUs
  int resident_blocks_now;
  {
    std::lock_guard<std::mutex> g(mu);
    auto key = std::make_tuple(dev, fn, lds);
    auto it = resident_cache.find(key);
    if (it == resident_cache.end()) {
      if (lds > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      int cus = 0;
      const int per_cu = resident_blocks(fn, lds, &cus);
      it = resident_cache.emplace(key, std::make_pair(per_cu, cus)).first;
    }
    const int cus = n_cus > 0 && n_cus < it->second.second ? n_cus : it->second.second;
    resident_blocks_now = it->second.first * cus;
    if (resident_blocks_now > ALIGN_GRID) resident_blocks_now = ALIGN_GRID;  // (ws_cols holds that many lanes)
  }
  // grid_pct < 100 leaves block slots free on every CU: a persistent grid that fills the chip would keep the
  // kernels of another stream (RCCL's exchange) waiting until it ends
  uint64_t resident = (uint64_t)resident_blocks_now * (uint64_t)(grid_pct < 10 ? 10 : (grid_pct > 100 ? 100 : grid_pct)) / 100;
  if (resident < 1) resident = 1;
  uint64_t grid = tiles < resident ? tiles : resident;
  if (want_blocks && grid > want_blocks) grid = want_blocks;
  void *args[] = {(void *)&ix, (void *)&p, (void *)&cb};
  (void)hipLaunchKernel(fn, dim3((uint32_t)grid), dim3(ALIGN_BLOCK), args, lds, s);
}

static size_t align_lds(const DevIndex &ix, const CallBuffers &cb, bool fast) {
  const bool wide = ix.all_local == 0;
  const uint32_t nm = cb.paired ? 2u : 1u;
  const uint32_t min_rows = nm + 1u;  // a finished tile leaves its results in the columns
  const uint32_t krows = cb.key_words + 1 > min_rows ? cb.key_words + 1 : min_rows;
  return (size_t)krows * ALIGN_BLOCK * 8 + (size_t)(fast ? 0 : LDS_COLS) * ALIGN_BLOCK * 4 + ALIGN_LDS_EXTRA + (fast ? ALIGN_LDS_META : 0) +
         (wide ? (size_t)ix.window_words * ALIGN_BLOCK * 8 : 0);  // (the LDS row window of wide indexes, push_col)
}

// Indexes with stretch records (no wide classes, components that fit a 64-row window, a "several left flanks" set) take the
// fast walk (walk_fast); NIMBLE_FAST_ALIGN=0 or any other index: the general one.
void launch_align(hipStream_t s, const DevIndex &ix, const nimble_align_params &p, const CallBuffers &cb,
                  int want_counters, int grid_pct, int n_cus) {
  if (cb.n == 0) return;
  const uint64_t tiles = (cb.n + ALIGN_BLOCK - 1) / ALIGN_BLOCK;
  const bool wide = ix.all_local == 0;
  static const bool fast_on = !(getenv("NIMBLE_FAST_ALIGN") && atoi(getenv("NIMBLE_FAST_ALIGN")) == 0);
  const bool fast = fast_on && !wide && ix.srec && ix.mleft;
  (void)hipMemsetAsync(cb.tile_ctr, 0, (size_t)TILE_COUNTERS * TILE_COUNTER_STRIDE, s);  // (the launch's tile counters)
  launch_align_kernel(s, align_kernel(cb.paired != 0, want_counters != 0, wide, fast ? 1 : 0), align_lds(ix, cb, fast), ix, p, cb, tiles,
                      grid_pct, n_cus, 0);
}

void launch_intern_claim(hipStream_t s, const DevIndex &ix, const CallBuffers &cb, int round) {
  if (cb.n == 0) return;
  const uint32_t full = blocks_for(cb.n, 256);
  hipLaunchKernelGGL(k_intern_claim, dim3(full < 1024u ? full : 1024u, cb.paired ? 2 : 1), dim3(256), 0, s, ix, cb, round);
}
void launch_intern_verify(hipStream_t s, const DevIndex &ix, const CallBuffers &cb) {
  if (cb.n == 0) return;
  const uint32_t full = blocks_for(cb.n, 256);
  hipLaunchKernelGGL(k_intern_verify, dim3(full < 1024u ? full : 1024u, cb.paired ? 2 : 1), dim3(256), 0, s, ix, cb);
}
void launch_dedup(hipStream_t s, const nimble_align_params &p, const CallBuffers &cb, uint32_t grid) {
  if (cb.n == 0) return;
  // A small first launch over the last reads fills the hot-key set (dedup_one) before the half million threads of
  // the main launch start together: otherwise every copy of a dominant key in that first wave of threads still
  // goes to its slot with atomics (5 % copies of one read: 2.5 ms; with the sample ahead: see DESIGN.md).
  static const uint64_t min_reads = getenv("NIMBLE_HOT_MIN_READS") ? strtoull(getenv("NIMBLE_HOT_MIN_READS"), nullptr, 10)
                                                                     : (1ull << 18);
  const uint64_t sample = cb.hot && cb.n > min_reads && cb.n >= 8 ? (cb.n / 4 < (1u << 14) ? cb.n / 4 : (1u << 14)) : 0;
  if (sample)
    hipLaunchKernelGGL(k_dedup, dim3(blocks_for(sample, 256)), dim3(256), 0, s, p, cb, (uint64_t)0, sample, 0);
  // at most 4096 workgroups striding over the reads: what a block learns about a dominant key serves all its later
  // tiles (2048: dominant keys cheaper still, but 0.63 instead of 0.53 ms without any; the full grid: 0.51 ms)
  const uint32_t full = blocks_for(cb.n - sample, 256);
  static const uint32_t persistent = (uint32_t)(getenv("NIMBLE_DEDUP_GRID") ? atoi(getenv("NIMBLE_DEDUP_GRID")) : 4096);
  const uint32_t cap = grid ? grid : persistent;
  // per-block callset counters pay once a block sees a few tiles (NIMBLE_HCLS_MIN_TILES, default 4)
  static const uint32_t min_tiles = (uint32_t)(getenv("NIMBLE_HCLS_MIN_TILES") ? atoi(getenv("NIMBLE_HCLS_MIN_TILES")) : 4);
  const uint32_t blocks = cap < full ? cap : full;
  hipLaunchKernelGGL(k_dedup, dim3(blocks), dim3(256), 0, s, p, cb, sample, cb.n,
                     (uint64_t)blocks * min_tiles <= (uint64_t)full ? 1 : 0);
}
void launch_count(hipStream_t s, const CallBuffers &cb) {
  if (cb.n == 0) return;
  hipLaunchKernelGGL(k_count, dim3(blocks_for(cb.n, 256)), dim3(256), 0, s, cb);
}
void launch_hist_compact(hipStream_t s, const CallBuffers &cb, uint32_t *c1, uint32_t *c2, uint64_t *cnt,
                         uint64_t cap, uint32_t *seg, uint32_t *rep) {
  hipLaunchKernelGGL(k_hist_compact, dim3(blocks_for(cb.hist_mask + 1, 256)), dim3(256), 0, s, cb, c1, c2, cnt, cap,
                     seg, rep);
}
void launch_maxinfo(hipStream_t s, const uint8_t *qual, const uint64_t *off, uint32_t fixed_len, uint64_t n,
                    const int64_t *length_scores, const int64_t *qual_probs, uint32_t *out) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_maxinfo, dim3(blocks_for(n, 256)), dim3(256), 0, s, qual, off, fixed_len, n, length_scores,
                     qual_probs, out);
}
void launch_hist_dense_se(hipStream_t s, const CallBuffers &cb, int64_t *counts, uint32_t n_classes) {
  hipLaunchKernelGGL(k_hist_dense_se, dim3(blocks_for(cb.hist_mask + 1, 256)), dim3(256), 0, s, cb, counts,
                     n_classes);
}
uint32_t route_grid() { return ROUTE_GRID; }
void launch_route(hipStream_t s, const CallBuffers &cb, uint32_t world, uint32_t *block_counts, uint64_t *block_first,
                  uint64_t *totals, uint64_t *rec, uint32_t *perm) {
  hipLaunchKernelGGL(k_route_count, dim3(ROUTE_GRID), dim3(256), 0, s, cb.key_hash, cb.n, world, block_counts);
  hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(256), 0, s, block_counts, ROUTE_GRID, world, totals, block_first);
  if (cb.n == 0) return;
  const size_t lds = (size_t)256 * (cb.key_words + 2) * 8;
  if (lds <= 56 * 1024)
    hipLaunchKernelGGL(k_route_scatter_lds, dim3(ROUTE_GRID), dim3(256), lds, s, cb, world, block_first, rec, perm);
  else
    hipLaunchKernelGGL(k_route_scatter, dim3(ROUTE_GRID), dim3(256), 0, s, cb, world, block_first, rec, perm);
}
void launch_dedup_records(hipStream_t s, const uint64_t *rec, uint64_t n, uint32_t key_words, uint64_t *table,
                          uint32_t slots, uint8_t *verdict) {
  if (n == 0) return;
  const size_t lds = (size_t)256 * (key_words + 2) * 8;
  if (lds <= 48 * 1024)
    hipLaunchKernelGGL(k_dedup_records<true>, dim3(blocks_for(n, 256)), dim3(256), lds, s, rec, n, key_words, table,
                       slots, verdict);
  else
    hipLaunchKernelGGL(k_dedup_records<false>, dim3(blocks_for(n, 256)), dim3(256), 0, s, rec, n, key_words, table,
                       slots, verdict);
}
void launch_count_verdicts(hipStream_t s, const nimble_align_params &p, const CallBuffers &cb, const uint32_t *perm,
                           const uint8_t *verdict) {
  if (cb.n == 0) return;
  hipLaunchKernelGGL(k_count_verdicts, dim3(blocks_for(cb.n, 256)), dim3(256), 0, s, p, cb, perm, verdict);
}
void launch_records_unpack(hipStream_t s, const uint64_t *rec, const CallBuffers &cb) {
  if (cb.n == 0) return;
  hipLaunchKernelGGL(k_records_unpack, dim3(blocks_for(cb.n, 256)), dim3(256), 0, s, rec, cb);
}
void launch_clear_call(hipStream_t s, const CallBuffers &cb, bool clear_latch) {
  const uint64_t items = cb.hist_mask + 1 > HOT_KEYS ? cb.hist_mask + 1 : HOT_KEYS;
  hipLaunchKernelGGL(k_clear_call, dim3(blocks_for(items, 256)), dim3(256), 0, s, cb, clear_latch ? 1 : 0);
}
void launch_publish_state(hipStream_t s, const uint64_t *state, uint64_t *host) {
  hipLaunchKernelGGL(k_publish_state, dim3(1), dim3(64), 0, s, state, host);
}
void launch_fill_u64(hipStream_t s, uint64_t *p, uint64_t v, uint64_t n) {
  if (n == 0) return;
  uint32_t grid = blocks_for(n, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_fill_u64, dim3(grid), dim3(256), 0, s, p, v, n);
}

}  // namespace nimble
