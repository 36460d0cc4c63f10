/* nimble_oracle.h -- C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the reference's
 * read-vs-library pseudoalignment + scoring path (BimberLab/nimble-aligner
 * src/align.rs + src/score.rs and the external k=30 de Bruijn walk they call).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (nimble-aligner_amd/) never links, imports or calls anything here.
 *
 * PARITY STATUS: pinned by every known-answer test the reference holds for this path
 * (tests/basic-cases.rs x8, tests/mismatch.rs x2, src/align.rs:1062-1107 x5 and the
 * unit-test literals of src/align.rs, src/filter/align.rs, src/utils.rs).  The
 * traversal details those tests do not reach (seed stride, left extension, cycle
 * cuts) follow the published algorithm of the un-vendored, un-pinned crates
 * hextraza/rust-pseudoaligner + 10XGenomics/rust-debruijn (Cargo.toml:22-23) and
 * are "parity unpinned" -- see DESIGN.md.
 */
#ifndef NIMBLE_ORACLE_H
#define NIMBLE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FilterReason, same order as src/align.rs:33-51 */
enum {
  ORA_SCORE_BELOW_THRESHOLD = 0,
  ORA_DISCARDED_MULTIPLE_MATCH = 1,
  ORA_DISCARDED_NONZERO_MISMATCH = 2,
  ORA_NO_MATCH = 3,
  ORA_NO_MATCH_AND_SCORE_BELOW_THRESHOLD = 4,
  ORA_DIFFERENT_FILTER_REASONS = 5,
  ORA_NOT_MATCHING_PAIR = 6,
  ORA_FORCE_INTERSECT_FAILURE = 7,
  ORA_SHORT_READ = 8,
  ORA_MAX_HITS_EXCEEDED = 9,
  ORA_HIGH_ENTROPY = 10,
  ORA_SUCCESSFUL_MATCH = 11,
  ORA_STRAND_WAS_WRONG = 12,
  ORA_TRIAGE_EMPTY_EQUIVALENCE_CLASS = 13,
  ORA_ABOVE_MISMATCH_THRESHOLD = 14,
  ORA_SKIPPED_ALIGN_DUE_TO_UNPAIRED_DUMMY = 15,
  ORA_NONE = 16
};

/* LibraryChemistry, src/align.rs:97-103 */
enum { ORA_UNSTRANDED = 0, ORA_FIVE_PRIME = 1, ORA_THREE_PRIME = 2, ORA_CHEM_NONE = 3 };

/* AlignFilterConfig, src/align.rs:79-95 */
typedef struct ora_config {
  uint64_t reference_genome_size;
  double score_percent;
  uint64_t score_threshold;
  uint64_t num_mismatches;
  int32_t discard_nonzero_mismatch;
  int32_t discard_multiple_matches;
  int32_t score_filter;
  int32_t intersect_level; /* 0 NoIntersect, 1 IntersectWithFallback, 2 ForceIntersect */
  int32_t require_valid_pair;
  int32_t strand_filter; /* LibraryChemistry */
  uint64_t discard_multi_hits;
  uint64_t max_hits_to_report;
  double trim_strictness;
  uint64_t trim_target_length;
} ora_config;

typedef struct ora_ref ora_ref;       /* reference_library::Reference */
typedef struct ora_index ora_index;   /* align::PseudoAligner */
typedef struct ora_result ora_result; /* return value of score::call */

/* ---- reference_library.rs:20-174 (row expansion part; JSON is parsed by the caller) ---- */
/* cells are column-major: cells[c * n_rows + r].  Returns NULL and sets ora_last_error on the
 * reference's panics (missing sequence_name / sequence / group_on column, non-DNA base). */
ora_ref *ora_ref_create(int n_cols, const char *const *headers, int n_rows, const char *const *cells,
                        const char *group_on);
/* build a Reference verbatim (no rev rows added), for restating the reference's unit tests */
ora_ref *ora_ref_create_raw(int n_cols, const char *const *headers, int n_rows,
                            const char *const *cells, int group_on, int sequence_name_idx,
                            int sequence_idx);
void ora_ref_free(ora_ref *);
int ora_ref_n_rows(const ora_ref *);
int ora_ref_n_cols(const ora_ref *);
int ora_ref_group_on(const ora_ref *);
int ora_ref_sequence_name_idx(const ora_ref *);
int ora_ref_sequence_idx(const ora_ref *);
const char *ora_ref_header(const ora_ref *, int col);
const char *ora_ref_cell(const ora_ref *, int col, int row);
/* tests/basic-cases.rs:30-37 mutates the Reference: push a column and point group_on at it */
int ora_ref_push_column(ora_ref *, const char *header, const char *const *values, int n);
void ora_ref_set_group_on(ora_ref *, int col);
/* reference_library.rs:209-226; 0 = ok, else error text in ora_last_error */
int ora_sanity_check_config(const ora_config *);

/* ---- index: utils.rs:7-24 + external build_index::<Kmer30> ---- */
ora_index *ora_index_build_from_ref(const ora_ref *);
ora_index *ora_index_build(int n_seqs, const char *const *seqs); /* from_acgt_bytes on each */
void ora_index_free(ora_index *);
/* stats[0]=distinct kmers, [1]=nodes(unitigs), [2]=eq classes, [3]=total unitig bases, [4]=sum of class lens */
void ora_index_stats(const ora_index *, uint64_t *stats5);
/* node dump for structure tests: returns length; seq written as ACGT (cap bytes) */
int ora_index_node(const ora_index *, uint32_t node, char *seq, int cap, uint32_t *colour,
                   uint32_t *lext, uint32_t *rext);
int ora_index_class(const ora_index *, uint32_t colour, uint32_t *ids, int cap);

/* ---- a5: Pseudoaligner::map_read_with_mismatch (call site src/align.rs:965) ---- */
/* returns 1 = Some, 0 = None.  cls_len may exceed cls_cap (then only cls_cap ids are written). */
int ora_map_read(const ora_index *, const char *read, int len, int allowed_mismatches,
                 uint32_t *cls, int cls_cap, int *cls_len, int *score, int *mismatches);

/* ---- a4: align::pseudoalign (src/align.rs:945-989) ----
 * returns 1 when an AlignmentScore is produced (reason = ORA_SUCCESSFUL_MATCH), 0 when filtered.
 * normalized/score carry the Filter tuple values on the filtered path as the reference does. */
int ora_pseudoalign(const ora_index *, const ora_config *, const char *read, int len,
                    int min_read_length, uint32_t *cls, int cls_cap, int *cls_len, int *reason,
                    double *normalized, int *score);

/* ---- a6 / a7 / a9 and helpers restated for unit tests ---- */
int ora_filter_alignment_by_metrics(int cls_len, uint64_t score, double normalized,
                                    uint64_t score_threshold, double score_percent,
                                    int discard_multiple_matches, uint64_t mismatch_threshold,
                                    uint64_t mismatches); /* returns FilterReason or SUCCESSFUL_MATCH */
int ora_filter_pair(const uint32_t *a, int na, const uint32_t *b, int nb); /* 1 = filter */
double ora_shannon_entropy(const char *dna);
int ora_natural_lexical_cmp(const char *a, const char *b);
uint64_t ora_maxinfo(const char *quality, int qlen, uint64_t target_length, double strictness);
/* utils::revcomp; returns 0 ok / -1 on the reference's panic.  out must hold len+1 bytes */
int ora_revcomp(const char *seq, char *out);
/* coercion (a8) on explicit classes: has_r1/has_r2 = Option is Some.  Writes the callset joined by
 * '\t' into out (cap bytes).  Returns FilterReason recorded by the triage (ORA_NONE when counted). */
int ora_coerce(const ora_ref *, const ora_config *, int has_r1, const uint32_t *c1, int n1, int has_r2,
               const uint32_t *c2, int n2, char *out, int cap);
/* string-list helpers of src/align.rs restated 1:1 for the unit-test literals; lists are '\n'-joined */
int ora_filter_read_calls_with_orientation(const char *in, char *out, int cap);
int ora_filter_orientation_on_library_chemistry(const char *seq, const char *mate, int chem,
                                                char *out_seq, char *out_mate, int cap);
int ora_process_class_to_features(const ora_ref *, const ora_config *, const uint32_t *cls, int n,
                                  int ignore_rollup, char *out, int cap);
/* AlignmentOrientation::parse_calls (align.rs:276-285): one "feature\t0|1" line per call */
int ora_parse_calls(const char *in, char *out, int cap);
/* unmap (align.rs:851-864): row index of every feature name (first match); -1 = the reference's panic */
int ora_unmap(const ora_ref *, const char *features, uint32_t *out, int cap);
/* utils::get_reference_sequence_data (utils.rs:7-24): one "name\tDnaString::to_string()" line per row; -1 = its panic */
int ora_reference_sequence_data(const ora_ref *, char *out, int cap);
/* utils::sort_score_vector (utils.rs:54-59) on n_rows keys ('\n'-joined rows of '\t'-joined strings): order[i] = input
 * index of output row i */
int ora_sort_score_vector(const char *keys, int n_rows, int32_t *order);

/* ---- a1/a2/a3: score::call (src/score.rs:14-46) over in-memory reads ----
 * r1/r2: concatenated ASCII bases, off[n+1] byte offsets.  r2 may be NULL (single-end).
 * n_threads <= 1: one sequential call, exactly the reference's FASTQ pipeline.
 * n_threads  > 1: reads hash-partitioned by read_key over threads, merged counts (CPU-S baseline). */
ora_result *ora_call(const ora_index *, const ora_ref *, const ora_config *, const uint8_t *r1,
                     const uint64_t *r1_off, const uint8_t *r2, const uint64_t *r2_off, uint64_t n,
                     int n_threads, int keep_per_read);
/* ---- the BAM pipeline's use of score::call: one call per UMI group (src/process/bam.rs:183-226,229-290) ----
 * segment[n] (NULL = one group): reads with the same id form one score::call, in input order.
 * q1/q2 (NULL = no metadata): quality strings laid out like the bases (same offsets); the read is aligned after
 *   trim_sequence (align.rs:866-871) while the dedup key stays the untrimmed read (align.rs:576-579).
 * skip1/skip2 (NULL = none): SKIP_ALIGN dummies (align.rs:527-528,549-550).
 * Rows are sorted by (segment, callset); ora_result_row_segment gives the segment of a row. */
ora_result *ora_call_umi(const ora_index *, const ora_ref *, const ora_config *, const uint8_t *r1,
                         const uint64_t *r1_off, const uint8_t *r2, const uint64_t *r2_off, const uint8_t *q1,
                         const uint8_t *q2, const uint8_t *skip1, const uint8_t *skip2, const uint32_t *segment,
                         uint64_t n, int keep_per_read);
uint32_t ora_result_row_segment(const ora_result *, uint64_t i);
const int32_t *ora_result_align_len(const ora_result *, int mate); /* bases aligned after the trim (keep_per_read) */
void ora_result_free(ora_result *);
uint64_t ora_result_n_rows(const ora_result *);
/* features joined by '\t' (the TSV cell layout of utils::write_to_tsv), count */
const char *ora_result_row(const ora_result *, uint64_t i, int32_t *count);
/* per-read records (only when keep_per_read != 0); arrays of length n */
const int32_t *ora_result_reason(const ora_result *, int mate);   /* FilterReason per read */
const int32_t *ora_result_score(const ora_result *, int mate);    /* raw score (bases covered) */
const int32_t *ora_result_mismatch(const ora_result *, int mate); /* mismatches seen by the walk */
const uint64_t *ora_result_class_hash(const ora_result *, int mate); /* FNV-1a of the walk's class, 0 if there was no walk */
const uint8_t *ora_result_kept(const ora_result *, int mate); /* 1 = the mate's alignment passed pseudoalign (before filter_pair) */
const uint8_t *ora_result_counted(const ora_result *);  /* 1 for the read that represents its key in score_map */
/* counters for the roofline formula of SURVEY 8(d):
 * [0]=reads [1]=unique keys in score_map [2]=sum P (seed probes) [3]=sum U (nodes visited)
 * [4]=sum E (class entries read) [5]=reads with a seed hit [6]=reads prefiltered [7]=filter_reasons size */
void ora_result_counters(const ora_result *, uint64_t *c8);

const char *ora_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
