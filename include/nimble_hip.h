/* nimble_hip.h -- C ABI of the MI355X (gfx950) device path of nimble-aligner's hot path.
 *
 * This is the boundary a maintainer of BimberLab/nimble-aligner binds from Rust (see
 * INTEGRATION.md for the `extern "C"` block).  The reference has no FFI of its own: the
 * hot path sits behind the in-process seam `score::call` (src/score.rs:14-31) plus index
 * construction `debruijn_mapping::build_index::<Kmer30>` (src/bin/main.rs:121-128).  Each
 * entry point below names the reference interface it replaces.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative NIMBLE_E_* code, with text from nimble_last_error(); nothing throws across
 * the ABI.  There is no CPU fallback: without a HIP device every device entry point fails
 * with NIMBLE_E_NO_DEVICE.
 */
#ifndef NIMBLE_HIP_H
#define NIMBLE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIMBLE_ABI_VERSION 1

enum {
  NIMBLE_OK = 0,
  NIMBLE_E_INVALID = -1,   /* bad argument */
  NIMBLE_E_NO_DEVICE = -2, /* no HIP device / HIP runtime error at init */
  NIMBLE_E_HIP = -3,       /* HIP runtime error */
  NIMBLE_E_NOMEM = -4,
  NIMBLE_E_OVERFLOW = -5,  /* a device pool overflowed even after growth */
  NIMBLE_E_INTERNAL = -6
};

/* FilterReason codes, same order as `enum FilterReason` (src/align.rs:33-51) */
enum {
  NIMBLE_R_SCORE_BELOW_THRESHOLD = 0,
  NIMBLE_R_DISCARDED_MULTIPLE_MATCH = 1,
  NIMBLE_R_DISCARDED_NONZERO_MISMATCH = 2,
  NIMBLE_R_NO_MATCH = 3,
  NIMBLE_R_NOT_MATCHING_PAIR = 6,
  NIMBLE_R_SHORT_READ = 8,
  NIMBLE_R_HIGH_ENTROPY = 10,
  NIMBLE_R_SUCCESSFUL_MATCH = 11,
  NIMBLE_R_ABOVE_MISMATCH_THRESHOLD = 14,
  NIMBLE_R_SKIPPED_ALIGN_DUE_TO_UNPAIRED_DUMMY = 15,
  NIMBLE_R_NONE = 16
};

#define NIMBLE_CLASS_NONE 0xFFFFFFFFu /* "Option::None" for an equivalence class */

/* The part of AlignFilterConfig (src/align.rs:79-95) the per-read path reads
 * (pseudoalign src/align.rs:945-989, filter_alignment_by_metrics src/filter/align.rs:4-45,
 * require_valid_pair src/align.rs:582-588). */
typedef struct nimble_align_params {
  double score_percent;
  uint64_t score_threshold;
  uint32_t num_mismatches;
  uint32_t discard_nonzero_mismatch;
  uint32_t discard_multiple_matches;
  uint32_t require_valid_pair;
  uint32_t min_read_length; /* MIN_READ_LENGTH = 40 (src/align.rs:18) */
  uint32_t reserved;
} nimble_align_params;

typedef struct nimble_index nimble_index; /* replaces align::PseudoAligner (src/align.rs:21) */
typedef struct nimble_ctx nimble_ctx;     /* per-caller workspace: one `score::call` in flight */

/* ---- library / device ---- */
int nimble_abi_version(void);
const char *nimble_last_error(void);
int nimble_device_count(int *count);

/* ---- index: replaces build_index::<Kmer30>(&seqs, &names, &HashMap::new(), threads)
 *      (src/bin/main.rs:121-128, tests/utils.rs:48-51); input is what
 *      utils::get_reference_sequence_data (src/utils.rs:7-24) produces: one ASCII sequence per
 *      library row (fwd and §rev rows interleaved), converted as DnaString::from_acgt_bytes does.
 *      Names stay on the host.
 *
 *      Concurrency contract.  The graph, the dictionary and the static colour classes are immutable after build.
 *      The class table is not: a call whose reads produce an intersection that is no k-mer colour appends that class
 *      to the index (nimble_index_stats [7]) so that every class has ONE id per content for the life of the index.
 *      An index may be shared by any number of contexts, on one stream or several, driven from one host thread or
 *      several (one thread per context; a context itself is not thread-safe): the kernels that append classes are
 *      chained per index and never overlap, whichever streams they are on, and lookups beside them are safe (a class
 *      being appended is simply resolved one step later).  Calls on different streams therefore serialise at their
 *      interning step -- behind the other call's alignment -- and run concurrently everywhere else. */
int nimble_index_build(const uint8_t *seqs, const uint64_t *seq_off, uint32_t n_seqs, int device,
                       nimble_index **out);
/* An index freed while contexts on it are alive is released when the last of them is freed. */
void nimble_index_free(nimble_index *);
/* stats[0]=distinct k-mers [1]=unitigs [2]=static classes [3]=unitig bases [4]=static class entries
 * [5]=hash slots [6]=device bytes [7]=dynamic classes interned so far */
int nimble_index_stats(const nimble_index *, uint64_t stats[8]);
/* Equivalence class content (ascending library row ids).  Class ids below stats[2] are the k-mer
 * colour classes; higher ids are intersections interned on the device.  Returns the class length in
 * *len; copies at most cap ids. */
int nimble_class_get(const nimble_index *, uint32_t class_id, uint32_t *ids, uint32_t cap, uint32_t *len);
/* The same for many classes at once (a host that meets thousands of new classes in one histogram -- allele families of a
 * hundred rows -- must not pay a device round trip per class): lengths and pool offsets of the classes [first, first +
 * count), then any slice of the id pool.  Classes that a call still in flight is interning may read as length 0. */
int nimble_class_table_read(const nimble_index *, uint32_t first, uint32_t count, uint32_t *len, uint32_t *pool_off);
int nimble_class_pool_read(const nimble_index *, uint32_t pool_off, uint32_t count, uint32_t *ids);

/* Host-only diagnostic: builds the flat index exactly as nimble_index_build does but uploads nothing.
 * stats[0..4] as nimble_index_stats.  Lets CPU-only tests check the index builder. */
int nimble_flat_index_stats(const uint8_t *seqs, const uint64_t *seq_off, uint32_t n_seqs, uint64_t stats[5]);
/* Host-only as well: builds the flat index and checks its stretch records (what the fast walk of k_align reads) against the
 * unitig records they were cut from; *n_records = how many there are (0: this index takes the general walk).  NIMBLE_OK, or
 * NIMBLE_E_INTERNAL with what is wrong first. */
int nimble_flat_index_selfcheck(const uint8_t *seqs, const uint64_t *seq_off, uint32_t n_seqs, uint64_t *n_records);

/* ---- context ---- */
/* stream: a hipStream_t (as void*) to launch on, or NULL for the context's own stream. */
int nimble_ctx_create(nimble_index *, void *stream, nimble_ctx **out);
void nimble_ctx_free(nimble_ctx *);
/* The hipStream_t the context launches on.  A second context created on the same stream keeps two calls in
 * flight back to back: call i+1 runs on the GPU while the host reads the histogram of call i (results are
 * fetched on a side stream that waits only for their own call). */
void *nimble_ctx_stream(nimble_ctx *);

/* Options: NIMBLE_OPT_COUNTERS (default 0) -- collect the work counters of nimble_call_counters (probes, nodes,
 * class entries, seeded, prefiltered) inside the align kernel; costs about a third of that kernel's time, so it is
 * off unless asked for (reads and unique_keys are always available). */
enum {
  NIMBLE_OPT_COUNTERS = 1,
  /* percent (10..100, default 100) of the resident block slots the persistent align grid takes.  Below 100 every
   * CU keeps room for the kernels of another stream -- a multi-GPU pipeline sets 87 so that RCCL's exchange of the
   * next batch runs beside the align kernel instead of behind it */
  NIMBLE_OPT_ALIGN_GRID_PCT = 2,
  /* workgroups (0..1048576, default 1280 = NIMBLE_DEDUP_ASIDE) of the dedup kernel when the tail of a call (dedup,
   * count, compaction) runs on the index's side stream instead of the launch stream.  Used only while the index
   * has more than one context, i.e. calls in flight: the dedup is bound by the chip's atomic rate, not by CUs, so
   * the next call's pack and align start beside it.  0 keeps the tail on the launch stream. */
  NIMBLE_OPT_TAIL_ASIDE = 3
};
int nimble_ctx_set_option(nimble_ctx *, int option, int64_t value);

/* Memory space of the read buffers handed to nimble_call */
/* NIMBLE_MEM_HOST_PINNED (nimble_stream_append only): page-locked host buffers (nimble_pinned_alloc / _register) that the
 * caller leaves untouched until the SECOND following nimble_stream_append, or nimble_stream_end, on the context has returned
 * -- the copy of batch i then runs while the host prepares batch i + 1 and queues its copy behind it, so the link does
 * not idle between batches (three batches' buffers are in use at a time). */
enum { NIMBLE_MEM_HOST = 0, NIMBLE_MEM_DEVICE = 1, NIMBLE_MEM_HOST_PINNED = 2 };

/* ---- the hot path: replaces score::call (src/score.rs:14-46) up to, and excluding, the
 *      string-level coercion of distinct class pairs (which stays with the host caller).
 *
 * One call = one dedup scope, exactly like one `score::call`: reads are keyed by their base string
 * (R1 string + R2 string, src/align.rs:576-579); identical keys count once (last writer).
 *
 *   r1 / r1_off : concatenated ASCII bases of the n reads and n+1 byte offsets.
 *                 r1_off == NULL means fixed length: read i is r1[i*fixed_len .. (i+1)*fixed_len).
 *   r2 / r2_off : mates, or r2 == NULL for single-end.
 *   max_len     : upper bound on any single read length (sizes the packed key and the LDS tiles).
 *   mem         : NIMBLE_MEM_HOST (buffers are copied) or NIMBLE_MEM_DEVICE (used in place).
 *
 * Runs asynchronously on the context's stream; results are read with the getters below, which
 * synchronise. */
int nimble_call(nimble_ctx *, const nimble_align_params *, const uint8_t *r1, const uint64_t *r1_off,
                const uint8_t *r2, const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len,
                int mem);

/* The call with the reads already packed -- the form score::call itself receives them in: its `sequences` are DnaStrings
 * (src/score.rs:14-31; DnaString::from_acgt_bytes runs in the reader, src/parse/fastq.rs:32), 32 bases a u64.  Mate m of
 * read i is r<m>_len[i] bases in the r<m>_stride words at r<m>_words + i * r<m>_stride, first base in the highest bit
 * pair, A=0 C=1 G=2 T=3, zero bits behind the last base (layout and rules as nimble_stream_append_packed).  mem: host or
 * device memory; as with nimble_call the buffers are borrowed until the first getter returns.  Results are those of
 * nimble_call on the same reads. */
int nimble_call_words(nimble_ctx *, const nimble_align_params *, const uint64_t *r1_words, const uint32_t *r1_len,
                      uint32_t r1_stride, const uint64_t *r2_words, const uint32_t *r2_len, uint32_t r2_stride, uint64_t n,
                      uint32_t max_len, int mem);

/* ---- the call as the BAM pipeline makes it (src/process/bam.rs:183-226,229-290; src/align.rs:516-552): many
 *      UMI groups in one launch, reads trimmed for quality before they are aligned, unpaired dummies skipped.
 *      Every field is optional (NULL / 0 = absent); the arrays live in the same memory space as the reads.
 *
 *  segment   [n] scope id per read(-pair): the reference calls score::call once per UMI, so the dedup of
 *            src/align.rs:685 and the per-callset counts are per segment.  Ids need not be sorted or dense;
 *            n_segments must exceed the largest id (0 = take it from the array, host memory only).
 *  qual      quality strings laid out exactly like the bases (same offsets / fixed_len).  Each mate is aligned on
 *            its first maxinfo(quality, trim_target_length, trim_strictness) bases (trim_sequence,
 *            src/align.rs:866-942) while the dedup key stays the untrimmed read (src/align.rs:576-579).
 *  skip      [n] per mate: non-zero = SKIP_ALIGN dummy (src/align.rs:527-528,549-550): not aligned, reason
 *            NIMBLE_R_SKIPPED_ALIGN_DUE_TO_UNPAIRED_DUMMY. */
typedef struct nimble_call_extra {
  const uint32_t *segment;
  uint32_t n_segments;
  uint32_t reserved;
  const uint8_t *qual[2];
  double trim_strictness;
  uint64_t trim_target_length;
  const uint8_t *skip[2];
} nimble_call_extra;
int nimble_call_ex(nimble_ctx *, const nimble_align_params *, const uint8_t *r1, const uint64_t *r1_off,
                   const uint8_t *r2, const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len,
                   int mem, const nimble_call_extra *extra);
/* Histogram of a segmented call: one entry per distinct (segment, class R1, class R2), sorted that way, with the
 * number of unique read keys and one representative read index (the reference keeps the metadata of one read per
 * callset, src/align.rs:245-251).  After a call without segments every entry has segment 0. */
int nimble_histogram_seg(nimble_ctx *, uint32_t *segment, uint32_t *class_r1, uint32_t *class_r2, uint64_t *count,
                         uint32_t *representative, uint64_t cap, uint64_t *n_entries);
/* bases of each mate that were aligned (the read length unless the call trimmed for quality) */
int nimble_read_align_len(nimble_ctx *, int mate, uint32_t *align_len, uint64_t n);

/* ---- split form of the call, for multi-GPU runs: reads are packed where they are, exchanged between
 *      ranks in packed form (40 B instead of 150 B per read; the key hash that routes them is a function of
 *      the converted bases, so equal keys meet on one rank), and the rest of the call runs on the receiver.
 *
 * All arrays are caller-owned DEVICE memory with n entries (keys: key_words planes of n u64, word-major).
 * nimble_pack fills them; nimble_call_packed consumes them and must see them alive until the first getter. */
typedef struct nimble_packed {
  uint64_t *keys;     /* [key_words][n]: R1 bases ++ R2 bases, 2 bits each, first base in the high bits */
  uint32_t *len[2];   /* bases per mate (len[1] may be NULL for single-end) */
  uint64_t *hash;     /* hash of (total length, packed words): dedup and routing key */
  uint8_t *pre[2];    /* prefilter verdict per mate: NIMBLE_R_SHORT_READ / NIMBLE_R_HIGH_ENTROPY / 255 = align */
  uint32_t key_words; /* nimble_key_words(max_len, paired) */
  uint32_t paired;
} nimble_packed;
uint32_t nimble_key_words(uint32_t max_len, int paired);
int nimble_pack(nimble_ctx *, const nimble_align_params *, const uint8_t *r1, const uint64_t *r1_off,
                const uint8_t *r2, const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                const nimble_packed *out);
int nimble_call_packed(nimble_ctx *, const nimble_align_params *, const nimble_packed *in, uint64_t n,
                       uint32_t max_len);
/* The exchange itself moves fixed-width records, one per read: key_words + 2 u64 = [key words ..., key hash,
 * len0 | len1 << 16 | pre0 << 32 | pre1 << 40].  nimble_route_records groups the n packed reads by destination
 * rank (key hash mod world; order inside a destination is arbitrary) into `records` (device, n * (key_words + 2)
 * u64) and returns the number of records per destination in counts[world] (host) -- the split sizes of the
 * all-to-all; complete on return.  With counts == NULL it only enqueues the routing on the context's stream;
 * nimble_route_counts(ctx, counts) then waits for that routing alone -- work enqueued behind it in the meantime
 * keeps running -- and returns the counts.  nimble_unpack_records turns received records back into packed arrays on the
 * context's stream (nimble_call_packed on the same context may follow without a wait). */
int nimble_route_records(nimble_ctx *, const nimble_packed *in, uint64_t n, uint32_t world, uint64_t *records,
                         uint64_t *counts);
int nimble_unpack_records(nimble_ctx *, const uint64_t *records, uint64_t n, const nimble_packed *out);
/* ... or run the call straight off the received records (no unpack; the records must stay alive and untouched until
 * the first getter): nimble_call_packed semantics, reads in record order. */
int nimble_call_records(nimble_ctx *, const nimble_align_params *, const uint64_t *records, uint64_t n,
                        uint32_t max_len, int paired);

/* ---- multi-GPU, second form: align where the reads are.  Only the keys travel: the rank that owns a key (hash
 *      mod world) sees every copy of it and answers one byte per copy -- 1 = this copy stands for the key in
 *      `score_map` (src/align.rs:496-505,685; any copy may, since the classes follow from the key) -- and the rank
 *      that aligned the read counts it.  Per-read records and classes never leave the rank that read the input.
 *
 *      nimble_ctx_defer_dedup arms the NEXT nimble_call on the context (plain call, single-end or fixed-length
 *      mates): after packing it routes the keys into `records` (device, n * (key_words + 2) u64, grouped by
 *      destination as nimble_route_records does) and `perm` (device, n u32: read index -> record slot), then
 *      aligns, and stops before the dedup.  nimble_route_counts waits for the routing only (the alignment keeps
 *      running) and returns the split sizes of the all-to-all (counts: host, world entries).  The owner runs nimble_dedup_records over the
 *      records it received (any context without a call in flight; asynchronous on its stream; verdict = device,
 *      one byte per record, in record order).  The verdict bytes go back by the inverse all-to-all, and
 *      nimble_count_verdicts (verdict in this rank's record order; must stay alive until the first getter) closes
 *      the call: pair filter, count, compaction.  The getters then behave as after nimble_call, the histogram
 *      holding this rank's share of the counts (sum over ranks = the single-GPU table). */
int nimble_ctx_defer_dedup(nimble_ctx *, uint32_t world, uint64_t *records, uint32_t *perm);
int nimble_route_counts(nimble_ctx *, uint64_t *counts);
int nimble_dedup_records(nimble_ctx *, const uint64_t *records, uint64_t n, uint32_t key_words, uint8_t *verdict);
int nimble_count_verdicts(nimble_ctx *, const uint8_t *verdict);

/* ---- streamed form of the call: ONE score::call whose reads arrive in batches (a FASTQ file larger than one
 *      buffer; src/process/fastq.rs:15-29 feeds the whole file to a single score::call, so the dedup scope is
 *      the whole stream).  begin lays the call arrays out for capacity_hint reads (they grow by doubling);
 *      every append copies its batch to the device on a side stream (overlapping the kernels of the previous
 *      batch), packs and aligns it; end runs interning, dedup and count over everything.  The getters below
 *      then behave as after nimble_call over all appended reads, in append order.
 *      append returns once the batch has left the host buffers (they may be refilled at once); with
 *      NIMBLE_MEM_DEVICE the buffers must stay untouched until the next append or end returns.
 *      max_len bounds every read of the stream (it fixes the key width). */
int nimble_stream_begin(nimble_ctx *, const nimble_align_params *, int paired, uint32_t max_len,
                        uint64_t capacity_hint);
int nimble_stream_append(nimble_ctx *, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                         const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, int mem);
/* The same batch handed over already packed, for a host whose link to the device is the bottleneck (a 150-base read is
 * 40 + 4 bytes this way instead of 150 + 8): mate m of read i is r<m>_len[i] bases in the r<m>_stride 64-bit words at
 * r<m>_words + i * r<m>_stride, 32 bases a word, the first base in the highest bit pair, A=0 C=1 G=2 T=3 (either case),
 * any other byte packed as A -- what DnaString::from_acgt_bytes makes of it (src/align.rs:576-579 builds the read key from
 * those) -- and zero bits behind the last base.  Results are those of nimble_stream_append on the same reads.  The
 * buffers are page-locked host memory (nimble_pinned_alloc / _register) and stay untouched until the SECOND following
 * append, or nimble_stream_end, has returned (as NIMBLE_MEM_HOST_PINNED).  Batches of both forms may be mixed in one
 * stream. */
int nimble_stream_append_packed(nimble_ctx *, const uint64_t *r1_words, const uint32_t *r1_len, uint32_t r1_stride,
                                const uint64_t *r2_words, const uint32_t *r2_len, uint32_t r2_stride, uint64_t m);
int nimble_stream_end(nimble_ctx *);
/* page-locked host memory for the batches (full-rate asynchronous H2D) */
int nimble_pinned_alloc(uint64_t bytes, void **out);
void nimble_pinned_free(void *);
/* ... or page-lock a buffer the caller already owns (until nimble_pinned_unregister) */
int nimble_pinned_register(void *p, uint64_t bytes);
void nimble_pinned_unregister(void *p);

/* Histogram of the call: one entry per distinct (class of R1, class of R2) over the unique read keys
 * that survived the per-read filters (the `score_map` of src/align.rs:496-505, grouped).
 * class == NIMBLE_CLASS_NONE where that mate has no passing alignment (PairState First/Second).
 * Call with cap == 0 to query *n_entries. */
int nimble_histogram(nimble_ctx *, uint32_t *class_r1, uint32_t *class_r2, uint64_t *count, uint64_t cap,
                     uint64_t *n_entries);

/* Per-read records of the last call (any pointer may be NULL).  mate = 0 or 1.
 *   reason : FilterReason code (SUCCESSFUL_MATCH when the mate's alignment was kept)
 *   score  : bases covered by the walk (0 when there was none)
 *   mism   : mismatches seen by the walk
 *   cls    : class id of a kept alignment, else NIMBLE_CLASS_NONE
 *   counted: 1 for the read that represents its key in the dedup scope */
int nimble_read_records(nimble_ctx *, int mate, int32_t *reason, int32_t *score, int32_t *mism, uint32_t *cls,
                        uint8_t *counted, uint64_t n);

/* Counters of the last call: [0]=reads [1]=unique keys kept [2]=seed probes [3]=nodes visited
 * [4]=class entries read [5]=reads with a seed hit [6]=reads prefiltered [7]=dynamic classes added */
int nimble_call_counters(nimble_ctx *, uint64_t c[8]);

/* Device time of the stages of the last call, measured with HIP events on the context's stream:
 * ms[0]=pack ms[1]=align ms[2]=intern ms[3]=dedup ms[4]=count ms[5]=total (first launch to last). */
int nimble_call_timing(nimble_ctx *, float ms[6]);
int nimble_ctx_synchronize(nimble_ctx *);

/* Dense count vector over the first n_classes class ids, single-end convenience for multi-GPU
 * reduction: counts[c] = number of unique keys whose R1 class is c and whose R2 class is NONE.
 * `counts` is DEVICE memory (int64), written on the context's stream, ready for an RCCL all-reduce. */
int nimble_histogram_dense_se(nimble_ctx *, int64_t *counts_dev, uint32_t n_classes);

/* ---- single-node multi-GPU without torch or MPI: one process, one rank (= one host thread) per device.
 *      Replaces nothing in the reference (it has no multi-device path); it is what BASELINE.json's north star asks
 *      of this build: reads shard by record, the ranks exchange routed records all-to-all and sum per-callset count
 *      vectors with RCCL over xGMI (SURVEY.md 8(b) `counts_allreduce`, 8(e)).
 *
 *      devices[n]: the HIP ordinal of each rank.  Distinct ordinals give RCCL communicators (ncclCommInitAll) and
 *      every collective below is an RCCL call on the rank's stream.  The SAME ordinal n times puts n ranks on one
 *      GPU -- the rehearsal and test configuration of a one-GPU box -- and the same entry points move the data with
 *      device copies instead (nimble_comm_uses_rccl tells which).
 *
 *      Every entry that takes (comm, rank) is COLLECTIVE: each rank calls it once, from its own thread, in the same
 *      order; a rank's failure is reported to all of them (nobody is left waiting). */
typedef struct nimble_comm nimble_comm;
int nimble_comm_create(const int *devices, int n, nimble_comm **out);
void nimble_comm_free(nimble_comm *);
int nimble_comm_size(const nimble_comm *);
int nimble_comm_uses_rccl(const nimble_comm *);
/* A rank's driver thread that fails OUTSIDE the collectives (a host-side fault: it will never make its next collective
 * call) calls this on its way out: the ranks waiting for it inside a collective return NIMBLE_E_INTERNAL instead of
 * waiting forever, and so does every later collective call on this communicator.  (The reference's analogue: a worker
 * thread that panics poisons the channel its peers wait on, src/process/bam.rs:183-226.)  Any thread may call it. */
void nimble_comm_abort(nimble_comm *);
/* In-place sum over the ranks of an int64 vector (counts are i32 in the reference, src/align.rs:186; int64 on the
 * wire): device memory on the rank's stream, or host memory (staged through the device, complete on return). */
int nimble_counts_allreduce(nimble_comm *, int rank, int64_t *counts_dev, uint64_t len, void *stream);
int nimble_counts_allreduce_host(nimble_comm *, int rank, int64_t *counts, uint64_t len);
/* All-to-all of routed records (nimble_route_records): `send` holds this rank's records grouped by destination,
 * send_counts[size] records for each; `recv` (device, room for recv_cap records) receives every rank's share for this
 * rank, in source order; *n_recv = how many.  Fails with NIMBLE_E_OVERFLOW on every rank, nothing moved, if somebody's
 * recv buffer is too small. */
int nimble_records_alltoall(nimble_comm *, int rank, const uint64_t *send, const uint64_t *send_counts,
                            uint32_t rec_words, uint64_t *recv, uint64_t recv_cap, uint64_t *n_recv, void *stream);
/* One score::call whose reads are spread over the ranks (the dedup scope is the whole call, so equal read keys must
 * meet on one rank): begin names the rank's context (its index lives on the rank's device); every append is one
 * round -- each rank packs its batch where it is, routes it by key hash, the records travel all-to-all and each rank
 * keeps what it owns (a rank without reads in a round appends n = 0); end runs the rest of the call over the records the
 * rank owns.  The context's getters then behave as after nimble_call_records, and the ranks' histograms add up to the
 * single-GPU one. */
int nimble_sharded_begin(nimble_comm *, int rank, nimble_ctx *, const nimble_align_params *, int paired,
                         uint32_t max_len);
int nimble_sharded_append(nimble_comm *, int rank, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                          const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, int mem);
/* the same round with reads the host has packed to 2 bits (host memory; layout of nimble_stream_append_packed): a quarter
 * of the bytes cross the link */
int nimble_sharded_append_packed(nimble_comm *, int rank, const uint64_t *r1_words, const uint32_t *r1_len, uint32_t r1_stride,
                                 const uint64_t *r2_words, const uint32_t *r2_len, uint32_t r2_stride, uint64_t n);
int nimble_sharded_end(nimble_comm *, int rank, uint64_t *n_owned);
/* A later batch holds a read longer than the call was opened for: every rank widens the records it has kept to the new
 * max_len (a local device copy, nothing is exchanged and nothing re-read; the key hash covers the bases, not the record
 * width, so routing stands) -- between two appends, on every rank with the same value, never smaller than before. */
int nimble_sharded_grow(nimble_comm *, int rank, uint32_t max_len);
/* Gives the open sharded call up on this rank: the records kept so far are dropped, no kernel is launched, the context is
 * free for the next begin.  (Local: the other ranks abort or end for themselves.) */
int nimble_sharded_abort(nimble_comm *, int rank);
/* Successive score::calls over reads spread across the ranks, software-pipelined -- a host that runs one call after the
 * other (the multi-GPU bench step; the consumer pool of src/process/bam.rs:183-226 seen from the device).  The three
 * contexts launch on one stream (two call contexts, one for packing).  submit(b) packs and routes batch b, launches the
 * call of batch b-1 right behind it (the launch stream waits for that batch's exchange, the host does not), waits for the
 * routing of b alone and starts its exchange on the rank's exchange stream, beside the running call.  *launched names
 * the context whose call has just been enqueued (NULL for the first submit): read its results (nimble_histogram, ...)
 * before submitting twice more.  flush launches the call of the last batch; end closes.  Every rank submits in step
 * (n = 0: no reads this round); an error of one rank is reported to all. */
int nimble_steps_begin(nimble_comm *, int rank, nimble_ctx *call0, nimble_ctx *call1, nimble_ctx *util,
                       const nimble_align_params *, int paired, uint32_t max_len);
int nimble_steps_submit(nimble_comm *, int rank, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                        const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, int mem, nimble_ctx **launched);
int nimble_steps_flush(nimble_comm *, int rank, nimble_ctx **launched);
int nimble_steps_end(nimble_comm *, int rank);

#ifdef __cplusplus
}
#endif
#endif
