// kernels.h -- launch interface between the C ABI (capi.cpp) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "../../include/nimble_hip.h"

namespace nimble {

constexpr uint32_t CLS_NONE = 0xFFFFFFFFu;     // NIMBLE_CLASS_NONE
constexpr uint32_t CLS_PENDING = 0xFFFFFFFEu;  // passing alignment whose class awaits interning
constexpr uint32_t SLOT_NONE = 0xFFFFFFFFu;
constexpr uint32_t HOT_KEYS = 4096;  // direct-mapped set of duplicated key hashes, per call
constexpr uint64_t HIST_EMPTY = ~0ULL;
constexpr uint32_t R_TODO = 255;  // prefilter verdict "go on to the walk"

// device view of the index (all pointers are device memory)
struct DevIndex {
  const uint4 *ht;          // {key lo, key hi, offset, node}
  uint64_t ht_mask;
  uint32_t ht_log2;
  const uint4 *bitmap;      // round-anchored presence filter, one 128-bit line per uint4 (flat_index.h)
  uint32_t bm_lines_log2;
  const uint32_t *l1;       // may be NULL: first level of the filter, one bit per value of the 12 shared bases (2 MiB)
  const uint64_t *mleft;    // may be NULL: 29-mers with several left flanks (flat_index.h mleft_maybe); NULL = no local re-seed
  uint32_t mleft_log2;
  const uint4 *node_rec;    // 4 x uint4 per node: {len, colour, exts, seq_start} {redge[4]} {bases 0..63} {64..127}
  const uint4 *node_ledge;
  // stretch records of the fast walk (flat_index.h FlatIndex::srec; all NULL when the index has none: classic launch only)
  const uint4 *srec;          // 2 x uint4 per record
  const uint32_t *srec_node;  // per record: its unitig.  With stretch records the dictionary names a k-mer's unitig by its first
                              // record; the general walk turns that into the unitig here (kernels.hip seed_node)
  const uint32_t *srec_base;  // per record
  const uint4 *srec_many;     // neighbours of the forks with more than two ways out
  const uint64_t *unitig;
  // class table: static colour classes first, device-interned intersections appended
  uint4 *cls_desc;          // {len | CLS_MASK_FLAG, base, mask lo, mask hi}; a wider static class: {len, first row,
                            //  rows spanned, word offset + 1 into cls_bits (0 = no bitmap)}
  uint32_t *cls_off;        // offset of the class in cls_ids (CSR form, every class has it)
  uint32_t *cls_ids;
  const uint64_t *cls_bits; // row bitmaps of the static classes wider than the mask form (see cls_desc)
  uint32_t all_local;       // every static class is in mask form: no colour list is kept during the walk
  uint32_t uniform_windows; // ... and the masks in the node records are relative to their component's first row (flat_index.h)
  uint32_t all_bitmaps;     // every static class outside the mask form has a row bitmap (the walk may intersect in a
                            // 256-row register window instead of keeping the visited colours)
  uint32_t window_words;    // ... or, when some bitmap is longer than 256 rows, in an LDS window of this many 64-row words per lane
                            // (0 = the register window)
  uint32_t n_static;
  uint32_t cls_cap;       // capacity in classes
  uint32_t ids_cap;       // capacity of cls_ids
  uint64_t *intern;       // {tag<<32 | class id}, 0 = empty
  uint64_t intern_mask;
  uint32_t *dyn_state;    // [0]=next class id [1]=next free id slot in cls_ids [2]=overflow flag
};

// per-call device arrays (SoA, stride = n; the packed keys have their own stride so that a streamed call can
// fill a slice [base, base + n) of arrays laid out for the call's capacity)
struct CallBuffers {
  uint64_t n;
  uint64_t key_stride;  // reads per key word row (>= n)
  uint32_t key_words;   // words per packed key (R1 ++ R2)
  uint32_t paired;
  uint64_t *keys;       // [key_words][n]
  const uint64_t *rec;  // may be non-NULL instead of keys / len / pre / key_hash: exchange records, one row per read
  uint32_t rec_words;   // key_words + 2
  uint32_t *len[2];     // bases per mate (the dedup key)
  uint32_t *alen[2];    // bases per mate that are aligned: == len unless the call trims for quality (BAM mode)
  const uint8_t *skip[2];   // per mate, may be NULL: SKIP_ALIGN dummies (align.rs:527-528)
  const uint32_t *seg;      // may be NULL: dedup / count scope per read (one UMI = one score::call)
  uint32_t cls_bits;        // segmented histogram key = seg << 2b | (c1+1) << b | (c2+1); 0 = c1 << 32 | c2
  uint32_t *hist_rep;       // may be NULL: largest read index counted into each histogram entry
  uint64_t *key_hash;   // hash of (total length, packed words)
  uint8_t *pre[2];      // prefilter verdict per mate: ShortRead / HighEntropy / R_TODO
  const uint32_t *min_cov;  // [len] -> smallest score with score/len >= score_percent in IEEE double
  uint8_t *reason[2];
  uint32_t *score[2];
  uint32_t *mism[2];
  uint32_t *cls[2];
  uint32_t *dyn_off[2];   // offset of the pending class in scratch
  uint32_t *dyn_len[2];
  uint64_t *dyn_hash[2];
  uint32_t *dyn_pos[2];   // intern slot claimed / matched in the last round
  uint32_t *slot;         // dedup slot per read or SLOT_NONE
  uint8_t *counted;
  uint32_t fuse_count;    // k_dedup counts the first copy of every key itself (k_count then only marks `counted`)
  uint32_t *scratch;      // pending class ids of this call
  uint32_t scratch_cap;
  uint32_t *ws_cols;      // overflow of the per-lane visited-colour list: [rows][lanes]
  uint32_t ws_rows;
  uint32_t ws_lanes;
  uint64_t *dedup;        // {tag<<32 | read index}, 0 = empty
  uint64_t *hot;          // may be NULL: HOT_KEYS hashes of keys seen more than once (see dedup_one)
  uint32_t dedup_slots;   // not a power of two: slot = mulhi32(hash >> 32, dedup_slots)
  uint64_t *hist_keys;    // (cls1 << 32 | cls2), HIST_EMPTY = empty
  uint64_t *hist_cnt;
  uint64_t hist_mask;
  // [0..7] = counters of nimble_call_counters, [8]=scratch used [9]=unresolved interns
  // [10]=error flags [11]=histogram entries (compaction) [12]=(unused since round 4: the align tile counters are tile_ctr)
  // [13]=third and later copies of a key met by the dedup sample [14]=input-error latch of k_pack (device-resident offsets that do not fit max_len;
  // cleared by the host) [15]=duplicates met by the dedup sample
  uint64_t *state;
  // tile counters of the align launch (kernels.hip k_align): TILE_COUNTERS counters, TILE_COUNTER_STRIDE bytes apart, zeroed by
  // launch_align
  uint64_t *tile_ctr;
};
constexpr uint32_t TILE_COUNTERS = 64, TILE_COUNTER_STRIDE = 128;

enum { ERR_SCRATCH = 1, ERR_CLASS_CAP = 2, ERR_IDS_CAP = 4, ERR_HIST = 8 };

void launch_pack(hipStream_t s, const uint8_t *r1, const uint64_t *off1, const uint8_t *r2, const uint64_t *off2,
                 uint32_t fixed_len, uint32_t max_len, uint32_t min_len, const double *plog, uint32_t plog_max_len,
                 const CallBuffers &cb);
// the same from reads packed by the host: 32 bases a word, `stride` words a read (kernels.hip: k_pack_words)
void launch_pack_words(hipStream_t s, const uint64_t *w1, const uint32_t *len1, uint32_t stride1, const uint64_t *w2,
                       const uint32_t *len2, uint32_t stride2, uint32_t max_len, uint32_t min_len, const double *plog,
                       const CallBuffers &cb);
// n_cus: CUs the stream may use (a CU-masked stream; 0 = the whole device): the persistent grid is sized to them
void launch_align(hipStream_t s, const DevIndex &ix, const nimble_align_params &p, const CallBuffers &cb,
                  int want_counters, int grid_pct = 100, int n_cus = 0);
void launch_intern_claim(hipStream_t s, const DevIndex &ix, const CallBuffers &cb, int round);
void launch_intern_verify(hipStream_t s, const DevIndex &ix, const CallBuffers &cb);
void launch_dedup(hipStream_t s, const nimble_align_params &p, const CallBuffers &cb, uint32_t grid = 0);  // 0 = one thread per read
void launch_count(hipStream_t s, const CallBuffers &cb);
void launch_hist_compact(hipStream_t s, const CallBuffers &cb, uint32_t *c1, uint32_t *c2, uint64_t *cnt,
                         uint64_t cap, uint32_t *seg = nullptr, uint32_t *rep = nullptr);
// trim_sequence / maxinfo (align.rs:866-942): aligned length per read from the quality string and the two
// host-built integer tables (length scores [1000], quality scores [61])
void launch_maxinfo(hipStream_t s, const uint8_t *qual, const uint64_t *off, uint32_t fixed_len, uint64_t n,
                    const int64_t *length_scores, const int64_t *qual_probs, uint32_t *out);
void launch_hist_dense_se(hipStream_t s, const CallBuffers &cb, int64_t *counts, uint32_t n_classes);
// multi-GPU exchange: route packed reads by key hash into fixed-width records, and back into the packed arrays
uint32_t route_grid();  // blocks of the routing kernels (sizes block_counts / block_first: grid * world entries)
void launch_route(hipStream_t s, const CallBuffers &cb, uint32_t world, uint32_t *block_counts, uint64_t *block_first,
                  uint64_t *totals, uint64_t *rec, uint32_t *perm = nullptr);
// align-where-the-reads-are form of the multi-GPU step: the key's owner picks one copy per key (verdict bytes in
// record order); the aligning rank counts the picked copies (perm = read index -> record slot, from launch_route)
void launch_dedup_records(hipStream_t s, const uint64_t *rec, uint64_t n, uint32_t key_words, uint64_t *table,
                          uint32_t slots, uint8_t *verdict);
void launch_count_verdicts(hipStream_t s, const nimble_align_params &p, const CallBuffers &cb, const uint32_t *perm,
                           const uint8_t *verdict);
void launch_records_unpack(hipStream_t s, const uint64_t *rec, const CallBuffers &cb);
void launch_clear_call(hipStream_t s, const CallBuffers &cb, bool clear_latch);  // histogram table, state words, hot-key set
void launch_publish_state(hipStream_t s, const uint64_t *state, uint64_t *host);  // host: page-locked, 16 words
void launch_fill_u64(hipStream_t s, uint64_t *p, uint64_t v, uint64_t n);

// debug builds (-DNIMBLE_PROFILE_SECTIONS=1): clock cycles per section of k_align; returns 0 when not compiled in
int debug_sections(uint64_t out[16], int reset);
uint32_t align_ws_lanes();   // lanes of the align grid (sizes ws_cols)
uint32_t align_lds_cols();   // visited colours kept in LDS before spilling to ws_cols

}  // namespace nimble
