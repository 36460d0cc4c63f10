// nimble_host.hpp -- C++ host mirror of the reference's interfaces around the hot path.
//
// The reference is a Rust crate; its toolchain is absent from this image, so the host side above the
// device C ABI (include/nimble_hip.h) is written in C++ with the same module / function names,
// argument meaning and error behaviour:
//   nimble::reference_library  <-  src/reference_library.rs
//   nimble::utils              <-  src/utils.rs
//   nimble::align              <-  src/align.rs   (types, get_calls, coercion of class pairs)
//   nimble::score              <-  src/score.rs   (call)
//   nimble::parse::fastq       <-  src/parse/fastq.rs
//   nimble::process::fastq     <-  src/process/fastq.rs
// Rust panics become nimble::Panic exceptions carrying the reference's message text.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <map>
#include <chrono>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../csrc/threads.h"

struct nimble_index;
struct nimble_ctx;
struct nimble_packed;
struct nimble_align_params;

namespace nimble {

struct Panic : std::runtime_error {
  explicit Panic(const std::string &m) : std::runtime_error(m) {}
};

namespace align {

// src/align.rs:25-30
enum class IntersectLevel { NoIntersect = 0, IntersectWithFallback = 1, ForceIntersect = 2 };
// src/align.rs:97-103
enum class LibraryChemistry { Unstranded = 0, FivePrime = 1, ThreePrime = 2, None = 3 };
// src/align.rs:33-51
enum class FilterReason {
  ScoreBelowThreshold = 0, DiscardedMultipleMatch, DiscardedNonzeroMismatch, NoMatch, NoMatchAndScoreBelowThreshold,
  DifferentFilterReasons, NotMatchingPair, ForceIntersectFailure, ShortRead, MaxHitsExceeded, HighEntropy,
  SuccessfulMatch, StrandWasWrong, TriageEmptyEquivalenceClass, AboveMismatchThreshold,
  SkippedAlignDueToUnpairedDummy, None
};
const char *to_string(FilterReason r);  // Display impl, src/align.rs:53-77

// src/align.rs:79-95
struct AlignFilterConfig {
  size_t reference_genome_size = 0;
  double score_percent = 0.0;
  size_t score_threshold = 0;
  size_t num_mismatches = 0;
  bool discard_nonzero_mismatch = false;
  bool discard_multiple_matches = false;
  int32_t score_filter = 0;
  IntersectLevel intersect_level = IntersectLevel::NoIntersect;
  bool require_valid_pair = false;
  size_t discard_multi_hits = 0;
  size_t max_hits_to_report = 0;
  LibraryChemistry strand_filter = LibraryChemistry::Unstranded;
  double trim_strictness = 0.0;
  size_t trim_target_length = 0;
};

constexpr size_t MIN_READ_LENGTH = 40;        // src/align.rs:18
constexpr double MIN_ENTROPY_SCORE = 1.75;    // src/align.rs:19

}  // namespace align

namespace reference_library {

extern const char *const SPECIAL_REVCOMP_FEATURE_NAME_SEPARATOR;  // "§", reference_library.rs:8

// reference_library.rs:10-17
struct Reference {
  size_t group_on = 0;
  std::vector<std::string> headers;
  std::vector<std::vector<std::string>> columns;
  size_t sequence_name_idx = 0;
  size_t sequence_idx = 0;
};

// reference_library.rs:20-174
std::pair<align::AlignFilterConfig, Reference> get_reference_library(const std::string &path,
                                                                     align::LibraryChemistry strand_filter);
// same, from JSON text already in memory
std::pair<align::AlignFilterConfig, Reference> parse_reference_library(const std::string &json_text,
                                                                       align::LibraryChemistry strand_filter);
// reference_library.rs:209-226
void sanity_check_align_config(const align::AlignFilterConfig &cfg);

}  // namespace reference_library

namespace utils {

// utils.rs:7-24: (sequences as ASCII, names); the device index applies from_acgt_bytes itself
std::pair<std::vector<std::string>, std::vector<std::string>> get_reference_sequence_data(
    const reference_library::Reference &reference);
// utils.rs:27-51
void write_to_tsv(const std::vector<std::pair<std::vector<std::string>, int32_t>> &results,
                  const std::string &output_path);
// utils.rs:61-94
std::string revcomp(const std::string &sequence);
// utils.rs:96-119
double shannon_entropy(const std::string &dna);
// lexical_sort::natural_lexical_cmp as used at align.rs:846
int natural_lexical_cmp(const std::string &a, const std::string &b);
// utils.rs:54-59: `scores.sort_by(|a, b| a.0.cmp(&b.0))` -- a STABLE sort on the Vec<String> key (element-wise, each
// element byte-wise).  Rows are anything whose `.first` is the key.
template <class Row>
void sort_score_vector(std::vector<Row> &scores) {
  std::stable_sort(scores.begin(), scores.end(), [](const Row &a, const Row &b) { return a.first < b.first; });
}

}  // namespace utils

namespace align {

class Coercer;

// Replaces `PseudoAligner` (src/align.rs:21): the device-resident index plus a calling context.
class PseudoAligner {
 public:
  // build_index::<Kmer30>(&seqs, &names, &HashMap::new(), threads) -- src/bin/main.rs:121-128
  static std::unique_ptr<PseudoAligner> build_index(const std::vector<std::string> &sequences,
                                                    const std::vector<std::string> &names, int device = 0);
  ~PseudoAligner();
  nimble_index *index() const { return index_; }
  // slot 0 is the context every call uses; slot 1 (created on first use, same launch stream) lets a second
  // call be enqueued while the first one's results are read (begin_calls / end_calls); slot 2 is a utility
  // context for pack / route / unpack while calls are in flight on the other two (multi-GPU pipeline); slot 3
  // is a third call slot (the align-where-the-reads-are pipeline keeps three calls open)
  nimble_ctx *ctx(int slot = 0);
  const std::vector<uint32_t> &eq_class(uint32_t class_id);  // cached nimble_class_get
  // fetch the classes of `ids` that are not cached yet with a few bulk copies (nimble_class_table_read / _pool_read)
  void prefetch_classes(std::vector<uint32_t> ids);

  // Coercion memo: class ids are stable for the life of the index, and the coercion of a class pair depends
  // only on (Reference names / groups, the coercion part of the config).  The memo is keyed by an exact
  // copy of those inputs and is dropped as soon as a call arrives with different ones.
  struct CoercionMemo;
  CoercionMemo &memo_for(const reference_library::Reference &reference, const AlignFilterConfig &config);
  std::shared_ptr<CoercionMemo> memo_ptr() const { return memo_; }
  // light mode: calls return RowRefs only (no per-call copies of the callset strings); CallOutput::materialize()
  // builds the owning rows on demand.  The C ABI runs the index in light mode.
  void set_light_rows(bool on) { light_rows_ = on; }
  bool light_rows() const { return light_rows_; }

 private:
  PseudoAligner() = default;
  nimble_index *index_ = nullptr;
  nimble_ctx *ctx_ = nullptr;
  nimble_ctx *extra_[3] = {nullptr, nullptr, nullptr};  // slots 1, 2 and 3
  std::unordered_map<uint32_t, std::vector<uint32_t>> class_cache_;
  std::shared_ptr<CoercionMemo> memo_;
  bool light_rows_ = false;
};

// A batch of reads in memory: concatenated ASCII bases + n+1 offsets (what the reference's boxed
// iterators of Result<DnaString> yield, materialised).  `device` marks buffers already in HBM.
struct ReadBatch {
  const uint8_t *bases = nullptr;
  const uint64_t *offsets = nullptr;  // nullptr => fixed_len
  uint64_t n = 0;
  uint32_t fixed_len = 0;
  uint32_t max_len = 0;
  bool device = false;
  bool pinned = false;  // page-locked host buffers the caller keeps untouched until the next append but one has returned
  // the same reads packed by the host (parse::fastq::pack_reads_2bit); when set, a stream sends these instead of `bases`
  const uint64_t *words = nullptr;
  const uint32_t *lens = nullptr;
  uint32_t stride = 0;
};

// One row of the result: (callset, (count, metadata, metadata)); metadata is empty on the FASTQ path
typedef std::pair<std::vector<std::string>, int32_t> ScoreRow;

struct FilterRecord {  // value of the filter_reasons map, src/align.rs:408
  FilterReason r1, r2;
  size_t score1, score2;
  FilterReason triage;
};

// The rows of one call without a string copy: callset ids into the index's coercion memo (which keeps every
// callset it has produced, and its '\t'-joined form) in Vec<String> order, and the counts.
struct RowRefs {
  std::shared_ptr<PseudoAligner::CoercionMemo> memo;
  std::vector<int32_t> ids, counts;
  size_t size() const { return ids.size(); }
  const std::vector<std::string> &features(size_t i) const;
  const std::string &joined(size_t i) const;
};

struct CallOutput {
  std::vector<ScoreRow> rows;  // in Vec<String> order; empty until materialize() when the index is in light mode
  RowRefs refs;                // always filled
  // filled only when requested: read index -> reasons (the reference keys this by read string)
  std::vector<FilterRecord> per_read;
  void materialize();          // rows from refs (idempotent)
};

// align::get_calls (src/align.rs:392-467).  mates == nullptr for single-end.
CallOutput get_calls(const ReadBatch &sequences, const ReadBatch *mate_sequences, PseudoAligner &index,
                     const reference_library::Reference &reference, const AlignFilterConfig &config,
                     bool want_per_read = false);

// get_calls in two halves, for callers that keep two calls in flight (batch i+1 on the GPU while the host
// turns batch i's histogram into rows).  begin enqueues the device work and returns; end waits for that slot.
void begin_calls(const ReadBatch &sequences, const ReadBatch *mate_sequences, PseudoAligner &index,
                 const AlignFilterConfig &config, int slot);
CallOutput end_calls(uint64_t n_reads, PseudoAligner &index, const reference_library::Reference &reference,
                     const AlignFilterConfig &config, int slot, bool want_per_read = false);

// a summed table (several ranks) as rows of `index`
CallOutput rows_from_table(PseudoAligner &index, const reference_library::Reference &reference,
                           const AlignFilterConfig &config, const std::map<std::vector<std::string>, int64_t> &table);

// What the BAM pipeline adds to a call (process/bam.rs:229-290, align.rs:516-552); every field optional.
struct UmiExtras {
  const uint32_t *segment = nullptr;  // [n] UMI group per read(-pair): one score::call each
  uint32_t n_segments = 0;            // > largest id (0 = scan, host memory only)
  const uint8_t *qual[2] = {nullptr, nullptr};  // qualities, laid out like the bases: 3' quality trim
  const uint8_t *skip[2] = {nullptr, nullptr};  // SKIP_ALIGN dummies
};
struct UmiRow {
  uint32_t segment;
  int32_t callset;  // index into UmiOutput::callsets (a row does not carry a copy of its feature names)
  int32_t count;
  uint32_t representative;  // a read of this callset (the reference keeps one read's BAM fields per callset)
};
struct UmiOutput {
  std::vector<UmiRow> rows;           // sorted by (segment, callset)
  // the feature lists the rows refer to: the index's memo for (reference, config), which outlives the output and only grows
  const std::vector<std::vector<std::string>> *callsets = nullptr;
  const std::vector<std::string> &features(const UmiRow &r) const { return (*callsets)[(size_t)r.callset]; }
  std::vector<FilterRecord> per_read;  // when requested
};
UmiOutput get_calls_umis(const ReadBatch &sequences, const ReadBatch *mate_sequences, const UmiExtras &extras,
                         PseudoAligner &index, const reference_library::Reference &reference,
                         const AlignFilterConfig &config, bool want_per_read = false);

// get_calls over reads that arrive in batches (one call, dedup over everything appended): the FASTQ pipeline
// parses batch i+1 while the GPU packs and aligns batch i (include/nimble_hip.h: nimble_stream_*).
class CallStream {
 public:
  CallStream(PseudoAligner &index, const AlignFilterConfig &config, bool paired, uint32_t max_len,
             uint64_t capacity_hint);
  ~CallStream();
  void append(const ReadBatch &sequences, const ReadBatch *mate_sequences);
  uint64_t reads() const { return n_; }
  uint32_t max_len() const { return max_len_; }
  // closes the stream and returns the rows of get_calls over all appended reads (unsorted)
  CallOutput finish(const reference_library::Reference &reference, bool want_per_read = false);

 private:
  PseudoAligner &index_;
  AlignFilterConfig config_;
  bool paired_, open_ = false;
  uint32_t max_len_;
  uint64_t n_ = 0;
  std::chrono::steady_clock::time_point t0_;
};

// Split form used by the multi-GPU driver: pack where the reads are, exchange the packed form, run the rest
// of get_calls on the receiving rank (include/nimble_hip.h: nimble_pack / nimble_call_packed).
void pack_reads(const ReadBatch &sequences, const ReadBatch *mate_sequences, PseudoAligner &index,
                const AlignFilterConfig &config, const nimble_packed &out, int slot = 0);
// first half of get_calls_packed (the second half is end_calls on the same slot)
void begin_calls_records(const uint64_t *records, uint64_t n, uint32_t max_len, bool paired, PseudoAligner &index,
                         const AlignFilterConfig &config, int slot = 0);  // the same, straight off exchange records
void begin_calls_packed(const nimble_packed &in, uint64_t n, uint32_t max_len, PseudoAligner &index,
                        const AlignFilterConfig &config, int slot);
CallOutput get_calls_packed(const nimble_packed &in, uint64_t n, uint32_t max_len, PseudoAligner &index,
                            const reference_library::Reference &reference, const AlignFilterConfig &config);

// The coercion of one (class R1, class R2) pair into a callset: filter_and_coerce_sequence_call_orientations
// (src/align.rs:178-252).  Pure host code; evaluated once per distinct pair of a call.
class Coercer {
 public:
  Coercer(const reference_library::Reference &reference, const AlignFilterConfig &config);
  // returns the callset (empty when triaged) and sets `triage`
  std::vector<std::string> coerce(bool has1, const std::vector<uint32_t> &c1, bool has2,
                                  const std::vector<uint32_t> &c2, FilterReason &triage) const;
  // the pieces the reference tests on their own, through the same interned tables coerce() uses (names the reference
  // does not hold fall back to the string rule, as in the reference):
  // AlignmentOrientation::parse_calls (align.rs:276-285)
  std::vector<std::pair<std::string, bool>> parse_calls(const std::vector<std::string> &calls) const;
  // unmap (align.rs:851-864): row of the first sequence_name equal to each feature; panics on an unknown one
  std::vector<uint32_t> unmap(const std::vector<std::string> &feature_list) const;
  // process_equivalence_class_to_feature_list (align.rs:802-849)
  std::vector<std::string> feature_list(const std::vector<uint32_t> &equivalence_class, bool ignore_group_rollup) const;

 private:
  struct Impl;
  std::shared_ptr<Impl> impl_;
};

// the part of the config the device path reads (include/nimble_hip.h nimble_align_params)
void device_params(const AlignFilterConfig &config, nimble_align_params *out);

// BAM-only quality trimming (src/align.rs:866-942), kept for parity of the unit-level surface
size_t maxinfo(const std::string &quality, size_t target_length, double strictness);

}  // namespace align

namespace score {
// score::call (src/score.rs:14-46): get_calls + rows sorted by callset
align::CallOutput call(const align::ReadBatch &sequences, const align::ReadBatch *mate_sequences,
                       align::PseudoAligner &reference_index, const reference_library::Reference &reference,
                       const align::AlignFilterConfig &aligner_config, bool want_per_read = false);
align::CallOutput call_packed(const nimble_packed &in, uint64_t n, uint32_t max_len,
                              align::PseudoAligner &reference_index, const reference_library::Reference &reference,
                              const align::AlignFilterConfig &aligner_config);
}  // namespace score

namespace parse {
namespace fastq {
// An in-memory FASTQ file: what get_error_checked_fastq_readers (src/parse/fastq.rs:8-43) iterates.
// Reads plain or gzip-compressed files (niffler auto-detection by magic bytes).
struct FastqData {
  std::vector<uint8_t> bases;
  std::vector<uint64_t> offsets;  // n+1
  uint32_t max_len = 0;
  uint64_t n() const { return offsets.empty() ? 0 : offsets.size() - 1; }
};
// Panics with "Error -- could not parse read. Input R1 data malformed." (src/align.rs:517) /
// "... reverse read. Input R2 data malformed." (src/align.rs:541) on a malformed record.
FastqData read_fastq(const std::string &path, bool is_mate);
// n reads (ASCII at bases + off[i] .. off[i + 1]) as 2-bit words: read i in words[i * stride ..], first base in the highest
// bit pair of word 0, A=0 C=1 G=2 T=3 in either case, any other byte as A (DnaString::from_acgt_bytes), zero bits behind the
// last base; lens[i] = its length.  stride * 32 must hold the longest read.
void pack_reads_2bit(const uint8_t *bases, const uint64_t *off, uint64_t n, uint32_t stride, uint64_t *words, uint32_t *lens);
// The same, the way the reference's lazy iterators behave: at most `max_records` records are read, and a malformed record
// ends the read with its panic text in *error (empty when none) instead of panicking here.
FastqData read_fastq_lazy(const std::string &path, bool is_mate, uint64_t max_records, std::string *error);

// The same reader, incremental: batches in file order.  Compressed files: one thread inflates and parses ahead,
// `batch_reads` records per batch.  Plain files: the mapped file is parsed in 8 MiB chunks by a pool of up to 8 threads
// (one batch per chunk, whatever it holds).  A malformed record ends the file: the batch before it carries the records that parsed and the
// reference's panic text in `error` (the consumer decides when that panic fires, as the reference's lazy
// iterators would).
class BatchReader {
 public:
  struct Batch {
    FastqData data;
    bool last = false;        // no further batch follows (end of file, or `error`)
    std::string error;        // panic text of the malformed record that follows data, if any
    uint64_t raw_offset = 0;  // (compressed) file bytes consumed when the batch was closed
    uint64_t start = 0, end = 0;  // byte range of the records (plain files parsed in parallel chunks)
    // The reads once more, packed for the link to the device (include/nimble_hip.h: nimble_stream_append_packed): read i is
    // lens[i] bases in words[i * stride .. (i + 1) * stride), 32 bases a word.  stride == 0: not packed (the one-thread
    // readers, NIMBLE_FASTQ_PACK=0); the consumer then sends data.bases.
    std::vector<uint64_t> words;
    std::vector<uint32_t> lens;
    uint32_t stride = 0;
    // the two buffers that travel stay page-locked while the batch object is re-used (full-rate asynchronous H2D)
    void *pinned[2] = {nullptr, nullptr};
    void pin();
    void unpin();
    ~Batch() { unpin(); }
  };
  // packed: the batches are for nimble_stream_append_packed -- words and lens are filled (stride > 0), and where the file
  // allows it the ASCII copy is not made at all (data.bases empty, data.offsets the running lengths)
  BatchReader(const std::string &path, bool is_mate, size_t batch_reads, bool packed = false);
  ~BatchReader();
  std::unique_ptr<Batch> next();            // blocks until the next batch is parsed
  void recycle(std::unique_ptr<Batch> b);  // hand the buffers back for reuse

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};
}  // namespace fastq
// CPUs this process may really use: the smaller of the affinity mask and the cgroup's CPU quota (a container on a 256-thread
// host with a quota of 16 CPUs reports 256 from hardware_concurrency(); thread pools sized by that spend the quota on
// contention and run slower).  At least 1.
unsigned usable_cpus();

namespace pgzip {
// One gzip stream inflated by many threads (host/pgzip.cpp): pieces of the decompressed stream in order, each as 16-bit
// symbols -- a value below 256 is a byte, 256 + k is byte k of the 32 KiB `window` in front of the piece -- so that the
// expensive part of turning them into bytes runs in parallel too (resolve), with the member trailers for the CRC check.
// A large array that is written once and read once: anonymous memory on 2 MiB pages where the kernel grants them, never
// zero-filled by us, never copied on growth below its reservation.  (std::vector costs a page fault per 4 KiB of fresh
// memory and a memset on resize(); with a dozen threads filling such arrays at once the faults serialise in the kernel
// and the first pass over new memory ran ten times slower than the decoding it was for.)
template <class T>
class HugeBuf {
 public:
  HugeBuf() = default;
  ~HugeBuf() { release(); }
  HugeBuf(HugeBuf &&o) noexcept : map_(o.map_), map_bytes_(o.map_bytes_), p_(o.p_), n_(o.n_), cap_(o.cap_) {
    o.map_ = nullptr;
    o.p_ = nullptr;
    o.map_bytes_ = o.n_ = o.cap_ = 0;
  }
  HugeBuf &operator=(HugeBuf &&o) noexcept {
    if (this != &o) {
      release();
      new (this) HugeBuf(std::move(o));
    }
    return *this;
  }
  HugeBuf(const HugeBuf &) = delete;
  HugeBuf &operator=(const HugeBuf &) = delete;
  T *data() { return p_; }
  const T *data() const { return p_; }
  size_t size() const { return n_; }
  size_t capacity() const { return cap_; }
  bool empty() const { return n_ == 0; }
  T &operator[](size_t i) { return p_[i]; }
  const T &operator[](size_t i) const { return p_[i]; }
  void reserve(size_t cap);           // virtual address space only: untouched pages cost nothing
  void resize(size_t n) {             // new entries are unspecified (fresh pages read as zero)
    if (n > cap_) reserve(std::max(n, cap_ * 2));
    n_ = n;
  }
  void release();

 private:
  void *map_ = nullptr;
  size_t map_bytes_ = 0;
  T *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};
void *huge_map(size_t bytes, void **map, size_t *map_bytes);  // 2 MiB aligned, MADV_HUGEPAGE; panics when refused
void huge_unmap(void *map, size_t map_bytes);
template <class T>
void HugeBuf<T>::reserve(size_t cap) {
  if (cap <= cap_) return;
  void *m = nullptr;
  size_t mb = 0;
  T *q = (T *)huge_map(cap * sizeof(T), &m, &mb);
  if (n_) memcpy(q, p_, n_ * sizeof(T));
  if (map_) huge_unmap(map_, map_bytes_);
  map_ = m;
  map_bytes_ = mb;
  p_ = q;
  cap_ = cap;
}
template <class T>
void HugeBuf<T>::release() {
  if (map_) huge_unmap(map_, map_bytes_);
  map_ = nullptr;
  p_ = nullptr;
  map_bytes_ = n_ = cap_ = 0;
}

struct Piece {
  HugeBuf<uint16_t> sym;         // 32768 window placeholders, then the data
  std::vector<uint8_t> window;   // the 32768 bytes of the stream in front of the piece
  std::vector<uint64_t> member_ends;              // data positions where a gzip member ended ...
  std::vector<uint32_t> member_crc, member_isize;  // ... and that member's trailer
  size_t size() const { return sym.size() - 32768; }
};
class Reader {
 public:
  Reader(const std::string &path, unsigned threads);
  ~Reader();
  bool next(Piece &out);  // false at the end of the stream; panics on corrupt input
  void recycle(HugeBuf<uint16_t> &&sym);  // a piece's symbol array, done with: the next chunk decodes into it
  struct Impl;
 private:
  std::unique_ptr<Impl> impl_;
};
void resolve(const Piece &p, uint8_t *out);  // p.size() bytes
// bytes [from, to) of the piece into out[from .. to), and zlib's crc32 of them continued from `crc`
uint32_t resolve_crc(const Piece &p, size_t from, size_t to, uint8_t *out, uint32_t crc);
uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len);  // zlib's crc32, by carry-less multiplication
}  // namespace pgzip

namespace bam {
// What the reference reads of a BAM record (rust_htslib::bam::Record), decoded from BGZF + BAM with zlib alone.
struct Record {
  int32_t tid = -1, pos = -1, mtid = -1, mpos = -1, tlen = 0;
  uint32_t flag = 0, mapq = 0;
  std::string qname, seq /* ASCII, "=ACMGRSVTWYHKDBN" */, qual /* raw Phred bytes */;
  std::vector<uint8_t> aux;
  std::string skip_align;  // the SKIP_ALIGN tag SortedBamReader pushes ("TRUE" / "FALSE"; empty = not pushed)
  std::string cb;          // the CB tag, kept by SortedBamReader for its sort (not a BAM field)
  bool aux_string(const char *tag, std::string &out) const;  // Aux::String (type Z) only
};
extern const char *const BAM_FIELDS_TO_REPORT[38];  // src/parse/bam.rs:9-49
// A record by reference (the pipeline's own form): the body bytes -- everything behind block_size -- stay where the BGZF
// reader inflated them (kept alive by a Hold, below), and what the grouping needs of the aux data was found in ONE walk over it
// when the record came in.  Every reported field is formatted straight from these bytes when a row is written; no string is
// built per field and record, and the bytes are not copied on their way through the readers (two copies per record that
// missed the cache were most of the decoder thread's 250 ns).
struct Raw {
  const uint8_t *body = nullptr;     // the body (valid while the Hold that came with the record is)
  uint32_t len = 0;
  uint32_t aux = 0;                  // where the aux data starts inside the body
  uint32_t cb = 0, cb_len = 0;       // value of CB:Z inside the body; cb_len = NONE: no such string tag
  uint32_t umi = 0, umi_len = 0;     // value of UB:Z, else of UR:Z
  uint32_t l_seq = 0;                // bases of the record (kept here: the call's inputs are sized without touching the bodies)
  uint16_t flag = 0;
  uint8_t skip = 0;                  // SKIP_ALIGN as SortedBamReader pushes it: 0 = not pushed, 1 = "FALSE", 2 = "TRUE"
  uint8_t qual_bad = 0;              // the quality bytes are no ASCII (0xFF = absent): reported once, the field reads empty
  static constexpr uint32_t NONE = 0xFFFFFFFFu;
};
using Hold = std::shared_ptr<const void>;  // keeps a piece of inflated BAM data alive
class Reader {
 public:
  explicit Reader(const std::string &path);
  ~Reader();
  bool next(Record &r);  // false at end of file; panics on a truncated record
  // the same record by reference, described (lengths, aux offset, CB / UMI tags); r.body stays valid while hold() -- as it
  // reads right after the call -- is kept
  bool next_raw(Raw &r);
  const Hold &hold() const;
 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};
// src/parse/sorted_bam_reader.rs:6-186: records of one UMI at a time, ordered by cell barcode, dummies for unpaired reads
class SortedBamReader {
 public:
  SortedBamReader(const std::string &path, bool force_bam_paired);
  bool next(Raw &out);  // out.body stays valid while the holds() of that moment are kept
  const std::vector<Hold> &holds() const { return holds_; }
  uint64_t generation() const { return generation_; }  // changes whenever holds() does
 private:
  void fill_buffer();
  void add_dummy_paired_reads();
  void filter_paired_reads();
  Reader reader_;
  bool force_bam_paired_;
  std::string current_umi_, next_umi_;
  std::vector<Hold> holds_;   // what the records of buffer_ and next_records_ lie in
  Hold next_hold_;
  uint64_t generation_ = 0;
  std::vector<Raw> buffer_, next_records_;
  size_t cursor_ = 0;
};
// (UMI, cell barcode) groups of src/parse/bam.rs:51-288 by reference, many of them in ONE record array (a group of its own
// vectors cost two allocations made by the reader thread and freed by the consumer: half a million groups of a 4 M-pair file
// took the consumer a second to free).  Records 2k / 2k + 1 of a group are a pair.
struct UmiBatch {
  std::vector<Hold> holds;  // the inflated data the records' bodies lie in
  std::vector<Raw> recs;
  struct Group {
    uint32_t first = 0, count = 0;  // in recs
  };
  std::vector<Group> groups;
  void clear() {
    holds.clear();
    recs.clear();
    groups.clear();
  }
  // the reference's current_umi / current_cell_barcode of group g: the UMI of its first record, the cell barcode (CB without
  // its last two characters) of its last
  std::string umi_of(size_t g) const;
  std::string cell_of(size_t g) const;
};
// what the reference's UMIReader hands on per record, made from a Raw on demand (src/parse/bam.rs:139-254)
inline size_t record_seq_len(const uint8_t *body) { int32_t v; memcpy(&v, body + 16, 4); return (size_t)(uint32_t)v; }
inline bool record_is_reverse(const uint8_t *body) { return (body[14] & 0x10) != 0; }
void raw_sequence(const uint8_t *body, const Raw &r, std::string &out);   // bases with the non-biological ones clipped, A/C/G/T
void raw_quality(const uint8_t *body, const Raw &r, std::string &out);    // qualities in read direction, clipped alike
void raw_fields(const uint8_t *body, const Raw &r, std::vector<std::string> &out);  // the 38 BAM_FIELDS_TO_REPORT values
void raw_row_fields(const uint8_t *body, const Raw &r, std::string &out);
void raw_call_input(const uint8_t *body, const Raw &r, uint8_t *bases, uint8_t *quals);  // the read as the call takes it
bool raw_quality_is_text(const uint8_t *body, const Raw &r);  // no byte above 0x7F (the reference needs the qualities as UTF-8)
void append_int(std::string &out, long long v);  // the 36 written ones (no QUAL, no SEQ), tab-separated, appended
// src/parse/bam.rs:51-288: one (UMI, cell barcode) group at a time: sequences with the non-biological bases clipped, and
// the 38 reported fields per record
class UMIReader {
 public:
  UMIReader(const std::string &path, bool terminate_on_error, bool force_bam_paired);
  bool next();  // true = the final UMI has been read (the reference's `final_umi`)
  std::vector<std::string> current_umi_group;
  std::vector<std::vector<std::string>> current_metadata_group;
  std::string current_umi, current_cell_barcode;
  // the same step without the strings: the group appended to `out` by reference (what process::bam::process takes); false =
  // the input ended with this group
  bool next_group(UmiBatch &out);
 private:
  SortedBamReader reader_;
  [[maybe_unused]] bool terminate_on_error_;  // the reference panics on every record error it meets, whatever this says
  size_t read_counter_ = 0;
  // the record that ended the group before: the first of the next group
  Raw pend_;
  std::vector<Hold> pend_holds_;
  bool have_pend_ = false;
  uint64_t merged_generation_ = ~0ULL;  // SortedBamReader::generation() whose holds the batch in hand has already
  std::string current_iteration_key_, next_iteration_key_;
};
}  // namespace bam
}  // namespace parse

namespace process {
namespace bam {
// src/process/bam.rs:45-243: one gzip-compressed TSV per library, a row per callset of every UMI group plus a row per pair
// that stands for none
void process(const std::vector<std::string> &input_files,
             std::vector<std::unique_ptr<align::PseudoAligner>> &reference_indices,
             const std::vector<reference_library::Reference> &references,
             const std::vector<align::AlignFilterConfig> &aligner_configs, const std::vector<std::string> &output_paths,
             size_t num_cores, bool force_bam_paired);
std::string reverse_comp_if_needed(const std::string &seq, bool reverse_comp);  // process/bam.rs:407-415
bool parse_str_as_bool(const std::string &v);                                   // process/bam.rs:417-423
}  // namespace bam
namespace fastq {
// src/process/fastq.rs:7-30
void process(const std::vector<std::string> &input_files,
             std::vector<std::unique_ptr<align::PseudoAligner>> &reference_indices,
             const std::vector<reference_library::Reference> &references,
             const std::vector<align::AlignFilterConfig> &aligner_configs,
             const std::vector<std::string> &output_paths);
// The same pipeline over several GPUs of one node: one rank (host thread) per device, no torch (include/nimble_hip.h
// nimble_comm_* / nimble_sharded_*).  reference_indices[library][rank]: the library's index on each rank's device.  Every
// batch of the input is split over the ranks; each packs its share, the packed reads travel to the rank that owns their
// key (RCCL all-to-all), each rank runs the call over what it owns, the callsets are agreed by content and the count
// vectors summed with an RCCL all-reduce; rank 0 writes the TSV.  `devices` may name one device several times (ranks
// sharing a GPU: rehearsal and tests on a one-GPU box).
void process_sharded(const std::vector<std::string> &input_files,
                     std::vector<std::vector<std::unique_ptr<align::PseudoAligner>>> &reference_indices,
                     const std::vector<reference_library::Reference> &references,
                     const std::vector<align::AlignFilterConfig> &aligner_configs,
                     const std::vector<std::string> &output_paths, const std::vector<int> &devices);
}  // namespace fastq

namespace multi {
// Successive score::calls over device-resident read sets spread across the GPUs of one node, through the pipelined native
// step of the C ABI (include/nimble_hip.h nimble_steps_*): one rank = one host thread per device, RCCL inside the
// library, no torch -- what a host with a consumer pool (src/process/bam.rs:183-226) does call after call, and the
// multi-GPU step bench.py times (`--form native`).  indices[rank]: the library's index on that rank's device.
// reads[rank][set] (mates[rank][set], or empty): device pointers to n reads of fixed_len bases each; step b works on set
// b % n_sets.  Every step ends with the rows of the whole job on rank 0 (callsets agreed by content, counts summed by
// nimble_counts_allreduce).  Returns the wall milliseconds per timed step (fill and drain of the pipeline inside) and
// leaves the last step's table in *last.
struct StepsResult {
  double ms_per_step = 0.0;
  align::CallOutput last;
  bool rccl = false;
};
StepsResult run_steps(std::vector<std::unique_ptr<align::PseudoAligner>> &indices,
                      const reference_library::Reference &reference, const align::AlignFilterConfig &config,
                      const std::vector<int> &devices, const std::vector<std::vector<const uint8_t *>> &reads,
                      const std::vector<std::vector<const uint8_t *>> &mates, uint64_t n, uint32_t fixed_len, int warmup,
                      int steps, int align_grid_pct);
}  // namespace multi
}  // namespace process

}  // namespace nimble
