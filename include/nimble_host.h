/* nimble_host.h -- C ABI over the C++ host mirror (nimble-aligner_amd/host), for bindings and tests.
 *
 * The functions follow the reference's own module surface:
 *   nimble_library_*      reference_library::get_reference_library (src/reference_library.rs:20-174)
 *                         + utils::get_reference_sequence_data (src/utils.rs:7-24)
 *                         + build_index::<Kmer30> (src/bin/main.rs:121-128)
 *   nimble_score_call     score::call (src/score.rs:14-46) over in-memory reads
 *   nimble_fastq_process  process::fastq::process (src/process/fastq.rs:7-30)
 *   nimble_write_to_tsv   utils::write_to_tsv (src/utils.rs:27-51)
 * Rust panics surface as a non-zero return with the reference's message in nimble_host_last_error().
 */
#ifndef NIMBLE_HOST_H
#define NIMBLE_HOST_H
#include <stdint.h>

#include "nimble_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* AlignFilterConfig (src/align.rs:79-95) */
typedef struct nimble_host_config {
  uint64_t reference_genome_size;
  double score_percent;
  uint64_t score_threshold;
  uint64_t num_mismatches;
  int32_t discard_nonzero_mismatch;
  int32_t discard_multiple_matches;
  int32_t score_filter;
  int32_t intersect_level; /* 0 NoIntersect, 1 IntersectWithFallback, 2 ForceIntersect */
  int32_t require_valid_pair;
  int32_t strand_filter; /* 0 unstranded, 1 fiveprime, 2 threeprime, 3 none (src/align.rs:97-103) */
  uint64_t discard_multi_hits;
  uint64_t max_hits_to_report;
  double trim_strictness;
  uint64_t trim_target_length;
} nimble_host_config;

typedef struct nimble_library nimble_library; /* (AlignFilterConfig, Reference [, PseudoAligner]) */
typedef struct nimble_rows nimble_rows;       /* Vec<(Vec<String>, i32)> */

const char *nimble_host_last_error(void);

/* get_reference_library(path, strand_filter): host only, no device needed */
int nimble_library_load(const char *json_path, int strand_filter, nimble_library **out);
int nimble_library_parse(const char *json_text, int strand_filter, nimble_library **out);
void nimble_library_free(nimble_library *);
int nimble_library_get_config(const nimble_library *, nimble_host_config *out);
int nimble_library_set_config(nimble_library *, const nimble_host_config *in); /* tests mutate the config */
int nimble_library_n_rows(const nimble_library *);
int nimble_library_n_cols(const nimble_library *);
int nimble_library_group_on(const nimble_library *);
void nimble_library_set_group_on(nimble_library *, int col);
int nimble_library_sequence_name_idx(const nimble_library *);
int nimble_library_sequence_idx(const nimble_library *);
const char *nimble_library_header(const nimble_library *, int col);
const char *nimble_library_cell(const nimble_library *, int col, int row);
int nimble_library_push_column(nimble_library *, const char *header, const char *const *values, int n);
/* A `Reference` written out by hand, the way the reference's unit tests build theirs (src/align.rs:1533-1546,
 * src/utils.rs:129-139,151-161): column c has rows_of[c] cells (columns may differ in length -- that is what
 * utils.rs:149-164 tests), cells are given column after column; nothing is added (no reverse-complement rows), the
 * config is a default one. */
int nimble_library_from_table(int n_cols, const char *const *headers, const int *rows_of, const char *const *cells,
                              int group_on, int sequence_name_idx, int sequence_idx, nimble_library **out);
/* get_reference_sequence_data + build_index on `device`; needs a GPU */
int nimble_library_build_index(nimble_library *, int device);
/* the device handles behind the library's PseudoAligner (NULL before build_index); borrowed */
void *nimble_library_index(nimble_library *);
void *nimble_library_ctx(nimble_library *);
void *nimble_library_ctx_slot(nimble_library *, int slot); /* context of slot 0 / 1 / 3 (calls) or 2 (utility); NULL on error */

/* score::call.  r2 == NULL for single-end; *_off == NULL means fixed_len; mem as in nimble_hip.h */
int nimble_score_call(nimble_library *, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                      const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                      nimble_rows **out);
/* score::call in two halves, for callers that stream batches (the BAM pipeline calls score::call once per UMI
 * batch, src/process/bam.rs:183-226): begin enqueues the device work of one batch and returns; end waits for that
 * slot and returns its sorted rows.  The call slots (0, 1 and 3) share one launch stream, so batch i+1 runs on the GPU while
 * the host turns batch i's histogram into rows.  The read buffers of a slot are borrowed until its end. */
int nimble_score_call_begin(nimble_library *, int slot, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                            const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem);
/* the same with the reads already packed, as score::call receives them (DnaStrings): layout of nimble_call_words */
int nimble_score_call_begin_words(nimble_library *, int slot, const uint64_t *r1_words, const uint32_t *r1_len,
                                  uint32_t r1_stride, const uint64_t *r2_words, const uint32_t *r2_len, uint32_t r2_stride,
                                  uint64_t n, uint32_t max_len, int mem);
int nimble_score_call_end(nimble_library *, int slot, nimble_rows **out);
/* The BAM pipeline's calls (src/process/bam.rs:183-226,229-290): the reference runs score::call once per UMI
 * group with the reads' BAM metadata; here a whole batch of groups is ONE device call.  `extra` (all fields
 * optional) carries the group id per read(-pair), the quality strings for trim_sequence (src/align.rs:866-871,
 * trim settings from the library's config) and the SKIP_ALIGN flags (src/align.rs:527-528).  Rows: per group, the
 * callsets of get_calls with their counts and one representative read (whose BAM fields the caller reports,
 * src/align.rs:245-251), sorted by (group, callset).  With want_per_read, the filter_reasons entry of every read:
 * {r1 reason, r1 score, r2 reason, r2 score, triage reason} (src/align.rs:453-458); orientation is always None. */
typedef struct nimble_umi_extra {
  const uint32_t *segment;
  uint32_t n_segments;
  uint32_t reserved;
  const uint8_t *qual[2];
  const uint8_t *skip[2];
} nimble_umi_extra;
typedef struct nimble_umi_rows nimble_umi_rows;
int nimble_score_call_umis(nimble_library *, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                           const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                           const nimble_umi_extra *extra, int want_per_read, nimble_umi_rows **out);
void nimble_umi_rows_free(nimble_umi_rows *);
uint64_t nimble_umi_rows_count(const nimble_umi_rows *);
const char *nimble_umi_rows_get(const nimble_umi_rows *, uint64_t i, uint32_t *segment, int32_t *count,
                                uint32_t *representative);
uint64_t nimble_umi_rows_reads(const nimble_umi_rows *);
int nimble_umi_rows_filter(const nimble_umi_rows *, uint64_t read, int32_t out[5]);

/* score::call over reads that arrive in batches: ONE call (dedup over everything appended, as process/fastq.rs
 * feeds a whole file to one score::call); batch i is packed and aligned on the GPU while the caller parses batch
 * i+1 (nimble_stream_* in nimble_hip.h).  max_len bounds every read of the stream. */
int nimble_score_stream_begin(nimble_library *, int paired, uint32_t max_len, uint64_t capacity_hint);
int nimble_score_stream_append(nimble_library *, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                               const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, int mem);
int nimble_score_stream_end(nimble_library *, nimble_rows **out);
/* split form of score::call for the multi-GPU driver (see nimble_pack / nimble_call_packed in nimble_hip.h):
 * pack on the rank that holds the reads, exchange the packed arrays, finish on the receiving rank */
int nimble_library_pack(nimble_library *, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                        const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                        const nimble_packed *out);
int nimble_score_call_packed(nimble_library *, const nimble_packed *in, uint64_t n, uint32_t max_len,
                             nimble_rows **out);
/* the same, for a pipeline that keeps calls in flight while the next batch is packed and exchanged: pack on a
 * given context slot (2 = the utility context, free of calls), begin the packed call on slot 0 / 1 and collect it
 * with nimble_score_call_end */
int nimble_library_pack_slot(nimble_library *, int slot, const uint8_t *r1, const uint64_t *r1_off, const uint8_t *r2,
                             const uint64_t *r2_off, uint64_t n, uint32_t fixed_len, uint32_t max_len, int mem,
                             const nimble_packed *out);
int nimble_score_call_packed_begin(nimble_library *, int slot, const nimble_packed *in, uint64_t n, uint32_t max_len);
/* the same straight off received exchange records (device, n rows of key_words + 2 u64; nimble_call_records) */
int nimble_score_call_records_begin(nimble_library *, int slot, const uint64_t *records, uint64_t n, uint32_t max_len,
                                    int paired);
/* get_error_checked_fastq_readers + score::call */
int nimble_score_call_fastq(nimble_library *, const char *r1_path, const char *r2_path, nimble_rows **out);
void nimble_rows_free(nimble_rows *);
uint64_t nimble_rows_count(const nimble_rows *);
/* digest of the row keys in order, and the counts as an array: lets a multi-GPU driver re-use its key table from
 * one call to the next and move only the counts */
uint64_t nimble_rows_signature(const nimble_rows *);
void nimble_rows_counts(const nimble_rows *, int64_t *out);
/* features joined by '\t' (the TSV cell layout) and the count */
const char *nimble_rows_get(const nimble_rows *, uint64_t i, int32_t *count);

/* fastq::process: libraries must have their index built */
int nimble_fastq_process(int n_inputs, const char *const *inputs, int n_libs, nimble_library *const *libs,
                         const char *const *outputs);
/* The same pipeline over several GPUs of one node, one rank (host thread) per entry of devices[] (an ordinal may repeat:
 * ranks sharing a GPU), reads exchanged by key and counts summed over RCCL -- include/nimble_hip.h nimble_comm_*.  The
 * library's index is built on every rank's device for the run. */
int nimble_fastq_process_sharded(int n_inputs, const char *const *inputs, nimble_library *lib, const int *devices,
                                 int n_devices, const char *output);
/* Successive score::calls over device-resident read sets spread across the GPUs of one node, natively: one host thread per
 * rank, the pipelined step of include/nimble_hip.h (nimble_steps_*), RCCL inside the device library.  libs[rank]: the
 * library with its index built on devices[rank] (an ordinal may repeat: ranks sharing a GPU, rehearsal and tests);
 * r1[rank * n_sets + set] (r2 likewise, or NULL): device pointers to n reads of fixed_len bases; step b works on set
 * b % n_sets; every step ends with the whole job's rows on rank 0.  *ms_per_step: wall time per timed step (pipeline fill
 * and drain inside); *last: the table of the last step (free with nimble_rows_free). */
int nimble_multi_steps(nimble_library *const *libs, const int *devices, int world, const uint8_t *const *r1,
                       const uint8_t *const *r2, int n_sets, uint64_t n, uint32_t fixed_len, int warmup, int steps,
                       int align_grid_pct, double *ms_per_step, int *used_rccl, nimble_rows **last);
/* process::bam::process (src/process/bam.rs:45-243): a single BAM file, one gzip-compressed TSV per library.  The reader is
 * this build's own (BGZF + BAM records on zlib, no htslib); the UMI grouping follows src/parse/sorted_bam_reader.rs and
 * src/parse/bam.rs. */
int nimble_bam_process(const char *input, int n_libs, nimble_library *const *libs, const char *const *outputs, int cores,
                       int force_bam_paired);
/* The UMI groups of a BAM file as the reference's UMIReader yields them, as text (needs no GPU; see host_capi.cpp). */
int nimble_host_bam_dump(const char *input, int force_bam_paired, const char *out_path);
/* The parallel gzip decoder of the FASTQ reader on its own (diagnostic): the decompressed stream written to out_path. */
int nimble_host_pgzip_decompress(const char *path, int threads, const char *out_path, uint64_t *n_pieces);
/* process/bam.rs:407-423 */
/* The FASTQ reader's packer, for tests and for callers that build batches themselves: n reads (ASCII, off[n + 1]) into the
 * form nimble_stream_append_packed takes (include/nimble_hip.h): read i in words[i * stride ..], lens[i] its length. */
int nimble_host_pack_reads_2bit(const uint8_t *bases, const uint64_t *off, uint64_t n, uint32_t stride, uint64_t *words,
                                uint32_t *lens);
int nimble_host_reverse_comp_if_needed(const char *seq, int reverse_comp, char *out, uint64_t cap);
int nimble_host_parse_str_as_bool(const char *v, int *out);
int nimble_write_to_tsv(const nimble_rows *, const char *output_path);

/* host-only pieces, exposed for CPU tests of the host logic */
/* filter_and_coerce_sequence_call_orientations on explicit classes; callset joined by '\t' into out;
 * returns the triage FilterReason (16 = None) or -1 on panic */
int nimble_host_coerce(const nimble_library *, int has_r1, const uint32_t *c1, int n1, int has_r2, const uint32_t *c2,
                       int n2, char *out, int cap);
/* the pieces of the coercion the reference tests on their own (lists are '\n'-joined):
 * AlignmentOrientation::parse_calls (src/align.rs:276-285; test :1234-1252) -> one "feature\t0|1" line per call;
 * unmap (src/align.rs:851-864; tests :1533-1608) -> row of each feature, returns how many or -1 on its panic;
 * process_equivalence_class_to_feature_list (src/align.rs:802-849);
 * utils::get_reference_sequence_data (src/utils.rs:7-24; tests :127-160) -> one "name\tsequence" line per row;
 * utils::sort_score_vector (src/utils.rs:54-59; tests :283-360) on n_rows keys ('\t'-joined strings): order[i] = input
 * index of output row i */
int nimble_host_parse_calls(const nimble_library *, const char *calls, char *out, int cap);
int nimble_host_unmap(const nimble_library *, const char *features, uint32_t *out, int cap);
int nimble_host_feature_list(const nimble_library *, const uint32_t *cls, int n, int ignore_group_rollup, char *out, int cap);
int nimble_host_reference_sequence_data(const nimble_library *, char *out, int cap);
int nimble_host_sort_score_vector(const char *keys, int n_rows, int32_t *order);
int nimble_host_natural_lexical_cmp(const char *a, const char *b);
double nimble_host_shannon_entropy(const char *dna);
int nimble_host_revcomp(const char *seq, char *out); /* out holds strlen(seq)+1 bytes */
uint64_t nimble_host_maxinfo(const char *quality, int qlen, uint64_t target_length, double strictness);
/* read_fastq: n reads, total bases, max length; -1 on the reference's panic */
int nimble_host_read_fastq(const char *path, uint64_t *n, uint64_t *bases, uint32_t *max_len);
/* the threaded batch reader of the FASTQ pipeline run to the end: totals, number of batches and an FNV-1a
 * checksum over (length, bases) of every record in file order; on a malformed record the totals cover the
 * records before it and the return is -1 with the reference's panic text */
int nimble_host_read_fastq_batched(const char *path, uint64_t batch_reads, uint64_t *n, uint64_t *bases,
                                   uint32_t *max_len, uint64_t *n_batches, uint64_t *checksum);
/* the same reader in the mode the pipeline runs it in (batches packed for nimble_stream_append_packed, no ASCII copy where
 * the file allows it): the same totals, and the checksum over (length, bases) with the bases as the packed words spell them
 * (upper case, anything that is no A/C/G/T as A); also checks that nothing but zero bits follows a read's last base */
int nimble_host_read_fastq_packed(const char *path, uint64_t batch_reads, uint64_t *n, uint64_t *bases, uint32_t *max_len,
                                  uint64_t *n_batches, uint64_t *checksum);
const char *nimble_host_filter_reason_text(int reason); /* Display for FilterReason */

#ifdef __cplusplus
}
#endif
#endif
