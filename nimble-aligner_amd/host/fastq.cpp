// fastq.cpp -- FASTQ ingest and the FASTQ pipeline driver.
// Mirrors src/parse/fastq.rs:8-43 (niffler + bio::io::fastq::Reader -> DnaString per record) and
// src/process/fastq.rs:7-30 (one score::call per library over the whole file, then write_to_tsv).
//
// Ingest is overlapped with the device work: one reader thread per input file inflates / reads and parses
// records into batches; the pipeline thread hands batch i to the streamed call of every library
// (align::CallStream: copy + pack + align of that batch on the GPU) while batch i+1 is being parsed.  It is
// still ONE score::call per library: dedup and counting run once over everything at the end.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <immintrin.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <atomic>
#include <exception>
#include <map>
#include <algorithm>
#include <mutex>
#include <thread>

#include "../../include/nimble_hip.h"
#include "nimble_host.hpp"

namespace nimble {
namespace parse {
namespace fastq {

namespace {

// Line source over a plain or gzip file (magic-byte detection is gzread's: plain files pass through unchanged,
// what niffler::from_path does).  Lines are returned as [b, b + len) inside an internal window.
class LineSource {
 public:
  explicit LineSource(const std::string &path) : path_(path) {
    f_ = gzopen(path.c_str(), "rb");
    if (!f_) throw Panic("Error -- could not determine compression format for " + path);
    gzbuffer(f_, 1 << 20);
    buf_.resize(1 << 22);
  }
  ~LineSource() {
    if (f_) gzclose(f_);
  }
  LineSource(const LineSource &) = delete;
  LineSource &operator=(const LineSource &) = delete;

  // read_line: false at EOF; the line excludes its terminator
  bool next(const uint8_t *&b, size_t &len) {
    for (;;) {
      const uint8_t *nl = p_ < end_ ? (const uint8_t *)memchr(buf_.data() + p_, '\n', end_ - p_) : nullptr;
      if (nl) {
        b = buf_.data() + p_;
        len = (size_t)(nl - b);
        p_ += len + 1;
        return true;
      }
      if (eof_) {
        if (p_ >= end_) return false;
        b = buf_.data() + p_;
        len = end_ - p_;
        p_ = end_;
        return true;
      }
      refill();
    }
  }
  // compressed (or plain) bytes consumed so far, for size estimates
  uint64_t raw_offset() const { return (uint64_t)std::max<z_off_t>(gzoffset(f_), 0); }

 private:
  void refill() {
    if (p_ > 0) {
      memmove(buf_.data(), buf_.data() + p_, end_ - p_);
      end_ -= p_;
      p_ = 0;
    }
    if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);  // a line longer than the window
    int got = gzread(f_, buf_.data() + end_, (unsigned)std::min<size_t>(buf_.size() - end_, 1u << 30));
    if (got < 0) throw Panic("Error -- could not determine compression format for " + path_);
    if (got == 0) eof_ = true;
    end_ += (size_t)got;
  }
  std::string path_;
  gzFile f_ = nullptr;
  std::vector<uint8_t> buf_;
  size_t p_ = 0, end_ = 0;
  bool eof_ = false;
};

// the same interface over a memory range (a mapped plain file)
class MemLines {
 public:
  MemLines(const uint8_t *d, size_t pos, size_t end) : d_(d), p_(pos), end_(end) {}
  bool next(const uint8_t *&b, size_t &len) {
    if (p_ >= end_) return false;
    b = d_ + p_;
    const uint8_t *nl = (const uint8_t *)memchr(b, '\n', end_ - p_);
    len = nl ? (size_t)(nl - b) : end_ - p_;
    p_ += len + (nl ? 1 : 0);
    return true;
  }
  size_t pos() const { return p_; }

 private:
  const uint8_t *d_;
  size_t p_, end_;
};

inline size_t trimmed(const uint8_t *b, size_t len) {
  while (len > 0 && (b[len - 1] == ' ' || b[len - 1] == '\t' || b[len - 1] == '\r' || b[len - 1] == '\n' ||
                     b[len - 1] == '\f' || b[len - 1] == '\v'))
    --len;
  return len;
}

// One record of bio::io::fastq::Reader::read appended to `out`: '@' header, sequence lines up to the '+' line,
// then as many quality lines as there were sequence lines; sequence and quality lengths are not compared.
// Returns false at a clean EOF; throws on a malformed record.
template <class Lines>
bool read_record(Lines &ln, FastqData &out, const char *malformed) {
  const uint8_t *b;
  size_t len;
  if (!ln.next(b, len)) return false;
  if (len == 0 || b[0] != '@') throw Panic(std::string(malformed) + ": Unable to read sequence");
  size_t seq_lines = 0;
  bool more = ln.next(b, len);
  while (more && !(len > 0 && b[0] == '+')) {
    const size_t t = trimmed(b, len);
    out.bases.insert(out.bases.end(), b, b + t);
    ++seq_lines;
    more = ln.next(b, len);
  }
  // rust-bio reads the quality lines raw, terminators included, and calls the record incomplete only when that text is
  // empty, i.e. when not one quality line could be read; a BLANK quality line is a line ("\n") and the record stands
  size_t qual_lines = 0;
  for (size_t k = 0; k < seq_lines; ++k) {
    if (!ln.next(b, len)) break;
    ++qual_lines;
  }
  if (qual_lines == 0) throw Panic(std::string(malformed) + ": Unable to read sequence");  // IncompleteRecord
  const uint64_t rl = out.bases.size() - out.offsets.back();
  if (rl > out.max_len) out.max_len = (uint32_t)rl;
  out.offsets.push_back(out.bases.size());
  return true;
}

const char *malformed_text(bool is_mate) {
  return is_mate ? "Error -- could not parse reverse read. Input R2 data malformed."
                 : "Error -- could not parse read. Input R1 data malformed.";
}

}  // namespace

FastqData read_fastq(const std::string &path, bool is_mate) {
  LineSource ln(path);
  FastqData out;
  out.offsets.push_back(0);
  while (read_record(ln, out, malformed_text(is_mate))) {
  }
  return out;
}

FastqData read_fastq_lazy(const std::string &path, bool is_mate, uint64_t max_records, std::string *error) {
  LineSource ln(path);
  FastqData out;
  out.offsets.push_back(0);
  error->clear();
  try {
    while (out.n() < max_records && read_record(ln, out, malformed_text(is_mate))) {
    }
  } catch (const Panic &e) {
    // (the record that failed may have left bases behind the last offset: cut them)
    out.bases.resize((size_t)out.offsets.back());
    *error = e.what();
  }
  return out;
}

// ---- 2-bit packing on the host: a 150-base read crosses the link as 44 bytes instead of 158 ---------------------------
namespace {
// 32 ASCII bases -> one word, first base in the highest bit pair
__attribute__((target("avx2"))) inline uint64_t pack32_avx2(const uint8_t *p) {
  const __m256i x = _mm256_or_si256(_mm256_loadu_si256((const __m256i *)p), _mm256_set1_epi8(0x20));  // lower case
  __m256i code = _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(3));                      // a0 c1 g3 t2
  code = _mm256_xor_si256(code, _mm256_and_si256(_mm256_srli_epi16(code, 1), _mm256_set1_epi8(1)));  // a0 c1 g2 t3
  const __m256i letters = _mm256_setr_epi8('a', 'c', 'g', 't', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'a', 'c', 'g', 't', 0, 0, 0,
                                           0, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i valid = _mm256_cmpeq_epi8(_mm256_shuffle_epi8(letters, code), x);  // anything else packs as A
  code = _mm256_and_si256(code, valid);
  // pairs: c0 * 4 + c1 (16-bit lanes), then quads: * 16 + next pair (32-bit lanes, one byte of four bases each)
  const __m256i pairs = _mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0104));
  const __m256i quads = _mm256_madd_epi16(pairs, _mm256_set1_epi32(0x00010010));
  // the low byte of each 32-bit lane, last quad first, so that a little-endian load has the first base on top
  const __m256i sh = _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 12, 8, 4, 0, -1, -1, -1, -1, -1,
                                      -1, -1, -1, -1, -1, -1, -1);
  const __m256i by = _mm256_shuffle_epi8(quads, sh);
  const uint32_t hi = (uint32_t)_mm256_extract_epi32(by, 0), lo = (uint32_t)_mm256_extract_epi32(by, 4);
  return ((uint64_t)hi << 32) | lo;
}
inline uint64_t pack_scalar(const uint8_t *p, uint32_t k) {  // k <= 32 bases, left-aligned
  uint64_t w = 0;
  for (uint32_t i = 0; i < k; ++i) {
    const uint8_t x = p[i] | 0x20;
    const uint64_t c = x == 'c' ? 1 : x == 'g' ? 2 : x == 't' ? 3 : 0;
    w |= c << (62 - 2 * i);
  }
  return w;
}
__attribute__((target("avx2"))) void pack_reads_avx2(const uint8_t *bases, const uint64_t *off, uint64_t n, uint32_t stride,
                                                     uint64_t *words, uint32_t *lens) {
  for (uint64_t i = 0; i < n; ++i) {
    const uint8_t *p = bases + off[i];
    const uint32_t len = (uint32_t)(off[i + 1] - off[i]);
    uint64_t *w = words + i * (uint64_t)stride;
    uint32_t k = 0, d = 0;
    for (; d + 32 <= len; d += 32) w[k++] = pack32_avx2(p + d);
    if (d < len) {  // the last, partial word the same way: from a copy padded with 'A' (code 0 = the zero padding)
      alignas(32) uint8_t tail[32];
      memset(tail, 'A', 32);
      memcpy(tail, p + d, len - d);
      w[k++] = pack32_avx2(tail);
    }
    for (; k < stride; ++k) w[k] = 0;
    lens[i] = len;
  }
}
}  // namespace

void pack_reads_2bit(const uint8_t *bases, const uint64_t *off, uint64_t n, uint32_t stride, uint64_t *words, uint32_t *lens) {
  static const bool avx2 = __builtin_cpu_supports("avx2") && getenv("NIMBLE_FASTQ_NO_AVX2") == nullptr;
  if (avx2) return pack_reads_avx2(bases, off, n, stride, words, lens);
  for (uint64_t i = 0; i < n; ++i) {
    const uint8_t *p = bases + off[i];
    const uint32_t len = (uint32_t)(off[i + 1] - off[i]);
    uint64_t *w = words + i * (uint64_t)stride;
    uint32_t k = 0;
    for (uint32_t d = 0; d < len; d += 32) w[k++] = pack_scalar(p + d, std::min<uint32_t>(32, len - d));
    for (; k < stride; ++k) w[k] = 0;
    lens[i] = len;
  }
}

namespace {
__attribute__((target("avx2"), always_inline)) inline size_t find_nl_avx2(const uint8_t *d, size_t from, size_t hard_end) {
  // position of the next line terminator, or hard_end
  const __m256i nlv = _mm256_set1_epi8('\n');
  size_t q = from;
  while (q + 32 <= hard_end) {
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(d + q)), nlv));
    if (m) return q + (size_t)__builtin_ctz(m);
    q += 32;
  }
  for (; q < hard_end; ++q)
    if (d[q] == '\n') return q;
  return hard_end;
}

// Parse and pack in one pass, for a reader whose batches only travel packed: the ordinary record -- '@' line, ONE sequence
// line, '+' line, one quality line, all four terminators inside the data -- is found with in-line 32-byte compares (memchr
// costs more to call on a 13-byte header than it spends scanning it) and its bases go from the file text straight into
// the batch's words; no ASCII copy is kept (data.bases stays empty, data.offsets holds the running lengths).  Records
// starting in [p, limit) are taken while they have that shape and fit `stride` words; the position of the first that does
// not (or >= limit) is returned, and the caller then does the whole range the general way.  A record must END before
// hard_end - 1 (what follows it has to be visible, as in parse_records for a window that is not the end of the input).
__attribute__((target("avx2"))) size_t parse_pack_avx2(const uint8_t *d, size_t p, size_t limit, size_t hard_end,
                                                       uint32_t stride, BatchReader::Batch &b) {
  FastqData &out = b.data;
  uint64_t total = 0;
  uint64_t n = 0;
  while (p < limit) {
    if (d[p] != '@') return p;
    const size_t n1 = find_nl_avx2(d, p, hard_end);
    if (n1 >= hard_end) return p;
    const size_t s = n1 + 1;
    if (s >= hard_end || d[s] == '+') return p;
    const size_t n2 = find_nl_avx2(d, s, hard_end);
    if (n2 + 1 >= hard_end || d[n2 + 1] != '+') return p;  // (a second sequence line, or the data ends)
    const size_t n3 = find_nl_avx2(d, n2 + 1, hard_end);
    if (n3 >= hard_end) return p;
    const size_t n4 = find_nl_avx2(d, n3 + 1, hard_end);
    if (n4 + 1 >= hard_end) return p;
    size_t len = n2 - s;
    if (len && d[s + len - 1] <= ' ') len = trimmed(d + s, len);  // (trailing blanks, a '\r': rare)
    if (len > 32u * (size_t)stride) return p;
    const uint64_t at = n * stride;
    if (at + stride > b.words.capacity() || n + 1 > b.lens.capacity()) {
      b.unpin();  // (the vectors move: see pack_batch)
      b.words.reserve(std::max<uint64_t>(2 * b.words.capacity(), at + stride + 1024));
      b.lens.reserve(std::max<uint64_t>(2 * b.lens.capacity(), n + 1024));
    }
    b.words.resize(at + stride);
    uint64_t *w = b.words.data() + at;
    uint32_t k = 0;
    size_t dn = 0;
    for (; dn + 32 <= len; dn += 32) w[k++] = pack32_avx2(d + s + dn);
    if (dn < len) {
      alignas(32) uint8_t tail[32];
      memset(tail, 'A', 32);
      memcpy(tail, d + s + dn, len - dn);
      w[k++] = pack32_avx2(tail);
    }
    for (; k < stride; ++k) w[k] = 0;
    b.lens.push_back((uint32_t)len);
    total += len;
    out.offsets.push_back(total);
    if (len > out.max_len) out.max_len = (uint32_t)len;
    ++n;
    p = n4 + 1;
  }
  return p;
}

bool packing_enabled() {
  static const bool on = [] {
    const char *e = getenv("NIMBLE_FASTQ_PACK");
    return !(e && atoi(e) == 0);
  }();
  return on;
}

// the packed form of a parsed batch (worker threads call this on their own chunk)
void pack_batch(BatchReader::Batch &b) {
  const bool on = packing_enabled();
  b.stride = 0;
  const uint64_t n = b.data.n();
  if (!on || n == 0 || b.data.max_len == 0) return;
  b.stride = (b.data.max_len + 31u) / 32u;
  const uint64_t nw = n * (uint64_t)b.stride;
  if (nw > b.words.capacity() || n > b.lens.capacity()) {
    // the vectors are about to move: their page-lock registration must not outlive the memory it names (the allocator
    // hands a freed block to another batch, whose own registration -- or a copy from it -- then collides with the stale one)
    b.unpin();
    b.words.reserve(nw + nw / 4);
    b.lens.reserve(n + n / 4);
  }
  b.words.resize(nw);
  b.lens.resize(n);
  pack_reads_2bit(b.data.bases.data(), b.data.offsets.data(), n, b.stride, b.words.data(), b.lens.data());
}
}  // namespace

// Page-lock the batch's buffers for the device copy.  A recycled batch keeps its registration as long as its vectors
// have not been re-allocated; pinning is best effort (pageable memory still works, only slower).
void BatchReader::Batch::pin() {
  static const bool off = getenv("NIMBLE_FASTQ_NO_PIN") != nullptr;
  if (off) return;
  // A packed batch is a megabyte: copied from pageable memory it is on the device before the append returns, at the cost of
  // the copy's own 20 us, and that measured 5 % FASTER end to end than page-locking the pool's buffers one by one at the
  // start of a run (0.3 ms each) for copies that then run behind the append.  ASCII batches are four times the size and
  // stay page-locked.
  static const bool pin_packed = getenv("NIMBLE_FASTQ_PIN_PACKED") != nullptr;
  if (stride && !pin_packed) {
    unpin();
    return;
  }
  // what travels: the packed words and lengths when the batch has them, else the ASCII bases and their offsets
  void *want[2] = {data.bases.capacity() ? (void *)data.bases.data() : nullptr,
                   data.offsets.capacity() ? (void *)data.offsets.data() : nullptr};
  uint64_t bytes[2] = {data.bases.capacity(), data.offsets.capacity() * sizeof(uint64_t)};
  if (stride) {
    want[0] = words.capacity() ? (void *)words.data() : nullptr;
    want[1] = lens.capacity() ? (void *)lens.data() : nullptr;
    bytes[0] = words.capacity() * sizeof(uint64_t);
    bytes[1] = lens.capacity() * sizeof(uint32_t);
  }
  for (int k = 0; k < 2; ++k) {
    if (pinned[k] == want[k]) continue;
    if (pinned[k]) nimble_pinned_unregister(pinned[k]);
    pinned[k] = nullptr;
    if (want[k] && bytes[k] >= (64u << 10) && nimble_pinned_register(want[k], bytes[k]) == 0) pinned[k] = want[k];
  }
}
void BatchReader::Batch::unpin() {
  for (int k = 0; k < 2; ++k) {
    if (pinned[k]) nimble_pinned_unregister(pinned[k]);
    pinned[k] = nullptr;
  }
}

// ---- plain (uncompressed) files: chunks of the mapped file are parsed by a pool of threads ------------------
// Chunk c covers the records that START inside [c * chunk, (c + 1) * chunk).  A worker does not know where the
// first of them starts, so it guesses (the first line that begins with '@' and whose next-but-one line begins with
// '+': exact for four-line records) and parses from there; the consumer, who knows where chunk c - 1 really ended,
// accepts the result when the guess was right and parses the chunk again itself when it was not (multi-line
// records can fool the guess; the outcome is the same either way, only slower).
class ParallelPlain {
 public:
  typedef BatchReader::Batch Batch;
  // batches shared by the parsers of successive memory windows (a compressed file comes in windows)
  struct Pool {
    std::mutex mu;
    std::vector<std::unique_ptr<Batch>> free;
  };
  // the same parser over a window of memory [d + from, d + n): `final` = the input ends with it; otherwise the record that
  // runs into the end of the window is left alone (carry_from() says where it starts) and no batch is the last
  ParallelPlain(const uint8_t *d, size_t from, size_t n, bool is_mate, bool final, std::shared_ptr<Pool> pool, bool packed,
                uint32_t stride_hint = 0)
      : is_mate_(is_mate), packed_(packed), final_(final), shared_(std::move(pool)) {
    stride_hint_ = stride_hint;
    data_ = d;
    size_ = n;
    base_ = from;
    true_start_ = from;
    start_workers();
  }
  ParallelPlain(const std::string &path, bool is_mate, bool packed) : is_mate_(is_mate), packed_(packed) {
    fd_ = open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw Panic("Error -- could not determine compression format for " + path);
    struct stat st;
    if (fstat(fd_, &st) != 0) {
      close(fd_);
      throw Panic("Error -- could not determine compression format for " + path);
    }
    size_ = (size_t)st.st_size;
    if (size_) {
      void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
      if (m == MAP_FAILED) {
        close(fd_);
        throw Panic("Error -- could not determine compression format for " + path);
      }
      data_ = (const uint8_t *)m;
      (void)madvise(m, size_, MADV_SEQUENTIAL);
    }
    mapped_ = data_ != nullptr;
    start_workers();
  }
  void start_workers() {
    chunk_ = 8u << 20;
    if (const char *e = getenv("NIMBLE_FASTQ_CHUNK")) chunk_ = std::max<size_t>((size_t)strtoull(e, nullptr, 10), 64);
    const size_t span = size_ - base_;
    n_chunks_ = span ? (span + chunk_ - 1) / chunk_ : 1;
    unsigned t = std::min(parse::usable_cpus(), 16u);
    if (const char *e = getenv("NIMBLE_FASTQ_THREADS")) t = (unsigned)std::max(1, atoi(e));
    t = (unsigned)std::min<size_t>(t, n_chunks_);
    window_ = 2 * (size_t)t + 1;
    results_.resize(n_chunks_);
    for (unsigned i = 0; i < t; ++i)
      if (!workers_.spawn([this] { work(); })) break;  // (fewer parser threads than asked for: slower, not wrong)
    if (workers_.size() == 0) {  // (the destructor does not run for a constructor that throws)
      if (mapped_) munmap((void *)data_, size_);
      if (fd_ >= 0) close(fd_);
      throw Panic("could not start a FASTQ parser thread (thread limit reached)");
    }
  }
  ~ParallelPlain() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    workers_.join();
    if (mapped_) munmap((void *)data_, size_);
    if (fd_ >= 0) close(fd_);
  }
  uint32_t stride_hint() const { return stride_hint_.load(std::memory_order_relaxed); }
  size_t carry_from() const { return true_start_; }  // (after the last batch) where the unparsed tail of the window starts
  static bool is_plain(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;  // the gzip path reports the failure
    unsigned char m[2] = {0, 0};
    size_t got = fread(m, 1, 2, f);
    fclose(f);
    return !(got == 2 && m[0] == 0x1f && m[1] == 0x8b);
  }

  // the next chunk's records, in file order
  std::unique_ptr<Batch> next() {
    const size_t c = delivered_;
    std::unique_ptr<Batch> b;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return results_[c] != nullptr; });
      b = std::move(results_[c]);
    }
    if (b->start != true_start_) {
      // the guess was wrong (or an earlier chunk ran into this one): parse the chunk from where it really starts
      parse_range(true_start_, chunk_end(c), *b);
    }
    b->pin();
    true_start_ = b->end;
    b->raw_offset = b->end;
    b->last = !b->error.empty() || (final_ && c + 1 == n_chunks_);
    {
      std::lock_guard<std::mutex> lk(mu_);
      ++delivered_;
    }
    cv_.notify_all();
    return b;
  }
  bool done() const { return delivered_ >= n_chunks_; }
  void recycle(std::unique_ptr<Batch> b) {
    if (shared_) {
      std::lock_guard<std::mutex> lk(shared_->mu);
      shared_->free.push_back(std::move(b));
      return;
    }
    std::lock_guard<std::mutex> lk(mu_);
    pool_.push_back(std::move(b));
  }

 private:
  std::unique_ptr<Batch> take() {
    if (shared_) {
      std::lock_guard<std::mutex> lk(shared_->mu);
      if (!shared_->free.empty()) {
        std::unique_ptr<Batch> b = std::move(shared_->free.back());
        shared_->free.pop_back();
        return b;
      }
    } else {
      std::lock_guard<std::mutex> lk(mu_);
      if (!pool_.empty()) {
        std::unique_ptr<Batch> b = std::move(pool_.back());
        pool_.pop_back();
        return b;
      }
    }
    std::unique_ptr<Batch> b(new Batch());
    b->data.bases.reserve(chunk_ / 2 + (chunk_ >> 4));  // one allocation (and one page-lock) per pooled batch
    b->data.offsets.reserve(chunk_ / 64 + 16);
    return b;
  }

  size_t chunk_end(size_t c) const { return std::min(size_, base_ + (c + 1) * chunk_); }

  // first record start at or after `from`: a line starting with '@' whose next-but-one line starts with '+'
  size_t guess_start(size_t from, size_t limit) const {
    size_t p = from;
    if (p > 0 && data_[p - 1] != '\n') {
      const uint8_t *nl = (const uint8_t *)memchr(data_ + p, '\n', size_ - p);
      if (!nl) return size_;
      p = (size_t)(nl - data_) + 1;
    }
    while (p < limit) {
      const uint8_t *n1 = (const uint8_t *)memchr(data_ + p, '\n', size_ - p);
      if (data_[p] == '@' && n1) {
        const size_t l2 = (size_t)(n1 - data_) + 1;
        const uint8_t *n2 = l2 < size_ ? (const uint8_t *)memchr(data_ + l2, '\n', size_ - l2) : nullptr;
        if (n2) {
          const size_t l3 = (size_t)(n2 - data_) + 1;
          if (l3 < size_ && data_[l3] == '+') return p;
        }
      }
      if (!n1) return size_;
      p = (size_t)(n1 - data_) + 1;
    }
    return p;  // no record starts inside the chunk
  }

  // records starting in [start, limit)
  void parse_range(size_t start, size_t limit, Batch &b) const {
    b.stride = 0;
    if (!packed_) return parse_records(start, limit, b);
    static const bool avx2 = __builtin_cpu_supports("avx2") && getenv("NIMBLE_FASTQ_NO_AVX2") == nullptr;
    if (avx2 && start < limit) {
      // one pass from the file text to the words; the stride is that of the reads seen so far (the first record when none)
      uint32_t stride = stride_hint_.load(std::memory_order_relaxed);
      if (stride == 0) {
        MemLines probe(data_, start, size_);
        const uint8_t *l;
        size_t len = 0;
        if (probe.next(l, len) && probe.next(l, len)) stride = (uint32_t)std::min<size_t>((trimmed(l, len) + 31) / 32, 1u << 20);
      }
      if (stride) {
        b.data.bases.clear();
        b.data.offsets.assign(1, 0);
        b.data.max_len = 0;
        b.error.clear();
        b.words.clear();
        b.lens.clear();
        b.start = start;
        const size_t q = parse_pack_avx2(data_, start, limit, size_, stride, b);
        if (q >= limit) {
          b.end = std::max(q, start);
          b.stride = b.data.n() ? stride : 0;
          uint32_t seen = stride_hint_.load(std::memory_order_relaxed);
          while (b.stride > seen && !stride_hint_.compare_exchange_weak(seen, b.stride, std::memory_order_relaxed)) {
          }
          return;
        }
      }
    }
    // something in the range is no ordinary record (or is longer than the stride, or ends the data): the general way
    parse_records(start, limit, b);
    pack_batch(b);
    uint32_t seen = stride_hint_.load(std::memory_order_relaxed);
    while (b.stride > seen && !stride_hint_.compare_exchange_weak(seen, b.stride, std::memory_order_relaxed)) {
    }
  }
  void parse_records(size_t start, size_t limit, Batch &b) const {
    b.data.bases.clear();
    b.data.offsets.assign(1, 0);
    b.data.max_len = 0;
    b.error.clear();
    b.start = start;
    MemLines ln(data_, start, size_);
    size_t rec = start;  // where the record being read starts
    try {
      while (ln.pos() < limit) {
        rec = ln.pos();
        if (!read_record(ln, b.data, malformed_text(is_mate_))) break;
        if (!final_ && ln.pos() >= size_) {
          // the record reaches the end of a window that is not the end of the input: it may go on in the next one
          b.data.offsets.pop_back();
          b.data.bases.resize(b.data.offsets.back());
          b.end = rec;
          return;
        }
      }
    } catch (const Panic &e) {
      b.data.bases.resize(b.data.offsets.back());
      if (!final_ && ln.pos() >= size_) {  // the lines ran out inside the record: the next window has the rest
        b.end = rec;
        return;
      }
      b.error = e.what();
    }
    b.end = std::max(ln.pos(), start);
  }

  void work() {
    for (;;) {
      size_t c;
      {
        std::unique_lock<std::mutex> lk(mu_);
        // run ahead of the consumer by at most `window_` chunks
        cv_.wait(lk, [&] { return stop_ || claimed_ >= n_chunks_ || claimed_ < delivered_ + window_; });
        if (stop_ || claimed_ >= n_chunks_) return;
        c = claimed_++;
      }
      std::unique_ptr<Batch> b = take();
      const size_t lo = base_ + c * chunk_, hi = chunk_end(c);
      parse_range(c == 0 ? base_ : guess_start(lo, hi), hi, *b);
      {
        std::lock_guard<std::mutex> lk(mu_);
        results_[c] = std::move(b);
      }
      cv_.notify_all();
    }
  }

  bool is_mate_;
  bool packed_ = false;  // batches travel packed (BatchReader's `packed`): words and lengths, no ASCII copy where avoidable
  mutable std::atomic<uint32_t> stride_hint_{0};
  bool final_ = true, mapped_ = false;
  std::shared_ptr<Pool> shared_;
  int fd_ = -1;
  const uint8_t *data_ = nullptr;
  size_t size_ = 0, base_ = 0, chunk_ = 0, n_chunks_ = 0, window_ = 0;
  std::vector<std::unique_ptr<Batch>> results_, pool_;
  threads::Group workers_;
  std::mutex mu_;
  std::condition_variable cv_;
  size_t claimed_ = 0, delivered_ = 0, true_start_ = 0;
  bool stop_ = false;
};

// ---- gzip files: the stream is inflated by many threads (pgzip.cpp) into windows of memory, each window parsed like a
// mapped plain file; the record that straddles two windows is carried over -------------------------------------------
class ParallelGz {
 public:
  typedef BatchReader::Batch Batch;
  ParallelGz(const std::string &path, bool is_mate, bool packed)
      : is_mate_(is_mate), packed_(packed), pool_(new ParallelPlain::Pool()) {
    unsigned t = std::min(parse::usable_cpus(), 32u);
    if (const char *e = getenv("NIMBLE_GZIP_THREADS")) t = (unsigned)std::max(1, atoi(e));
    threads_ = t;
    target_ = 256u << 20;
    if (const char *e = getenv("NIMBLE_GZIP_WINDOW")) target_ = std::max<size_t>((size_t)strtoull(e, nullptr, 10), 1u << 16);
    gz_.reset(new pgzip::Reader(path, t));
    producer_ = std::thread([this] { produce(); });
  }
  ~ParallelGz() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (producer_.joinable()) producer_.join();
  }
  std::unique_ptr<Batch> next() {
    for (;;) {
      if (!cur_) {
        std::unique_ptr<Window> w;
        {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] { return !windows_.empty(); });
          w = std::move(windows_.front());
          windows_.pop_front();
        }
        cv_.notify_all();
        if (!w->error.empty()) throw Panic(w->error);
        // the unparsed tail of the window before goes in front of this one's data (there is headroom for it)
        if (carry_.size() > w->head) {  // (a record longer than the headroom: move the window's data up)
          pgzip::HugeBuf<uint8_t> nb;
          const size_t body = w->buf.size() - w->head;
          nb.resize(carry_.size() + body);
          memcpy(nb.data() + carry_.size(), w->buf.data() + w->head, body);
          w->buf = std::move(nb);
          w->head = carry_.size();
        }
        if (!carry_.empty()) memcpy(w->buf.data() + w->head - carry_.size(), carry_.data(), carry_.size());
        const size_t from = w->head - carry_.size();
        win_ = std::move(w);
        cur_.reset(new ParallelPlain(win_->buf.data(), from, win_->buf.size(), is_mate_, win_->final, pool_, packed_, stride_hint_));
      }
      if (!cur_->done()) {
        std::unique_ptr<Batch> b = cur_->next();
        raw_ += 1;
        return b;
      }
      const size_t cf = cur_->carry_from();
      stride_hint_ = std::max(stride_hint_, cur_->stride_hint());
      carry_.assign(win_->buf.data() + std::min(cf, win_->buf.size()), win_->buf.data() + win_->buf.size());
      const bool was_final = win_->final;
      cur_.reset();
      {
        std::lock_guard<std::mutex> lk(mu_);  // the window's memory goes round
        if (spare_.size() < 3) spare_.push_back(std::move(win_));
      }
      win_.reset();
      if (was_final) return nullptr;  // (not reached: the final window's last batch has `last` set)
    }
  }
  void recycle(std::unique_ptr<Batch> b) {
    std::lock_guard<std::mutex> lk(pool_->mu);
    pool_->free.push_back(std::move(b));
  }

 private:
  struct Window {
    pgzip::HugeBuf<uint8_t> buf;
    size_t head = 0;  // the data starts here; the space in front takes the carried tail of the previous window
    bool final = false;
    std::string error;
  };
  // pieces -> windows: the pieces of a window are resolved to bytes and CRC-checked by a thread each
  void produce() {
    uint32_t crc = 0;
    uint64_t member_len = 0;
    bool more = true;
    try {
      // the first windows are small, so that parsing (and the device) start while most of the file is still being inflated
      size_t target = std::min<size_t>(target_, 32u << 20);
      while (more) {
        std::vector<pgzip::Piece> pieces;
        size_t total = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (total < target) {
          pgzip::Piece p;
          if (!gz_->next(p)) {
            more = false;
            break;
          }
          total += p.size();
          pieces.push_back(std::move(p));
        }
        target = std::min(target_, target * 2);
        const auto t1 = std::chrono::steady_clock::now();
        std::unique_ptr<Window> w;
        {
          std::lock_guard<std::mutex> lk(mu_);
          if (!spare_.empty()) {
            w = std::move(spare_.back());
            spare_.pop_back();
          }
        }
        if (!w) w.reset(new Window());
        w->head = 1u << 16;
        w->buf.resize(w->head + total);
        w->final = !more;
        std::vector<size_t> at(pieces.size() + 1, w->head);
        for (size_t k = 0; k < pieces.size(); ++k) at[k + 1] = at[k] + pieces[k].size();
        // the pieces cut into stretches of at most 2 MiB that end where a gzip member ends; every thread takes stretches
        // (a piece per thread left most threads idle: a window holds about as many pieces as the decoder has threads, of
        // very different sizes), their CRCs are combined in order below
        struct Task {
          size_t piece, from, to;
          bool ends_member;
          uint32_t member;  // index into the piece's member list when ends_member
          uint32_t crc;
        };
        std::vector<Task> tasks;
        const size_t STRETCH = 2u << 20;
        for (size_t k = 0; k < pieces.size(); ++k) {
          const pgzip::Piece &p = pieces[k];
          size_t from = 0;
          for (size_t m = 0; m <= p.member_ends.size(); ++m) {
            const size_t to = m < p.member_ends.size() ? (size_t)p.member_ends[m] : p.size();
            for (size_t q = from; q < to || q == from; q += STRETCH) {
              const size_t e = std::min(to, q + STRETCH);
              tasks.push_back(Task{k, q, e, e == to && m < p.member_ends.size(), (uint32_t)m, 0});
              if (e == to) break;
            }
            from = to;
          }
        }
        std::atomic<size_t> nextt{0};
        auto body = [&] {
          for (size_t i = nextt++; i < tasks.size(); i = nextt++) {
            Task &t = tasks[i];
            t.crc = pgzip::resolve_crc(pieces[t.piece], t.from, t.to, w->buf.data() + at[t.piece], 0);
          }
        };
        const unsigned nt = (unsigned)std::min<size_t>(threads_, std::max<size_t>(tasks.size(), 1));
        threads::run_beside(nt - 1, body);
        for (const Task &t : tasks) {
          crc = (uint32_t)crc32_combine(crc, t.crc, (z_off_t)(t.to - t.from));
          member_len += t.to - t.from;
          if (t.ends_member) {
            const pgzip::Piece &p = pieces[t.piece];
            if (crc != p.member_crc[t.member] || (uint32_t)member_len != p.member_isize[t.member])
              throw Panic("Error -- could not determine compression format (gzip member fails its CRC-32 / length check)");
            crc = 0;
            member_len = 0;
          }
        }
        for (auto &p : pieces) gz_->recycle(std::move(p.sym));
        const auto t2 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || windows_.size() < 2; });
        if (stop_) return;
        windows_.push_back(std::move(w));
        lk.unlock();
        cv_.notify_all();
        const auto t3 = std::chrono::steady_clock::now();
        t_gather_ += std::chrono::duration<double>(t1 - t0).count();
        t_resolve_ += std::chrono::duration<double>(t2 - t1).count();
        t_blocked_ += std::chrono::duration<double>(t3 - t2).count();
      }
      if (getenv("NIMBLE_GZIP_DEBUG"))
        fprintf(stderr, "[pgzip] window builder: %.3f s waiting for inflated pieces, %.3f s resolving + CRC, %.3f s blocked on the parser\n",
                t_gather_, t_resolve_, t_blocked_);
    } catch (const std::exception &e) {
      std::unique_ptr<Window> w(new Window());
      w->error = e.what();
      w->final = true;
      std::lock_guard<std::mutex> lk(mu_);
      windows_.push_back(std::move(w));
      cv_.notify_all();
    }
  }

  bool is_mate_;
  bool packed_ = false;
  uint32_t stride_hint_ = 0;  // words a read of the windows parsed so far (the next window's parser starts from it)
  std::shared_ptr<ParallelPlain::Pool> pool_;
  unsigned threads_ = 1;
  size_t target_ = 0;
  std::unique_ptr<pgzip::Reader> gz_;
  std::thread producer_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::unique_ptr<Window>> windows_;
  double t_gather_ = 0, t_resolve_ = 0, t_blocked_ = 0;
  std::vector<std::unique_ptr<Window>> spare_;  // windows parsed to the end: their memory takes the next one
  bool stop_ = false;
  std::unique_ptr<Window> win_;
  std::unique_ptr<ParallelPlain> cur_;
  std::vector<uint8_t> carry_;
  uint64_t raw_ = 0;
};

// ---- batch reader: a thread parses ahead, batches are handed over in file order ----------------------
struct BatchReader::Impl {
  std::unique_ptr<ParallelPlain> plain;  // plain files: chunk-parallel parse (then no reader thread of our own)
  std::unique_ptr<ParallelGz> gz;        // gzip files: parallel inflate into windows, each parsed the same way
  std::string path;
  bool is_mate;
  size_t batch_reads;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::unique_ptr<Batch>> ready, spare;
  bool stop = false;
  bool packed = false;  // the one-thread reader packs its batches too (they keep their ASCII copy)

  void run() {
    std::unique_ptr<Batch> cur;
    auto take_spare = [&]() -> std::unique_ptr<Batch> {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return stop || !spare.empty(); });
      if (stop) return nullptr;
      std::unique_ptr<Batch> b = std::move(spare.front());
      spare.pop_front();
      return b;
    };
    auto publish = [&](std::unique_ptr<Batch> b) {
      if (packed) pack_batch(*b);
      std::lock_guard<std::mutex> lk(mu);
      ready.push_back(std::move(b));
      cv.notify_all();
    };
    auto reset = [](Batch &b) {
      b.data.bases.clear();
      b.data.offsets.clear();
      b.data.offsets.push_back(0);
      b.data.max_len = 0;
      b.last = false;
      b.error.clear();
      b.raw_offset = 0;
    };
    cur = take_spare();
    if (!cur) return;
    reset(*cur);
    try {
      LineSource ln(path);
      for (;;) {
        const bool got = read_record(ln, cur->data, malformed_text(is_mate));
        if (!got) {
          cur->last = true;
          cur->raw_offset = ln.raw_offset();
          publish(std::move(cur));
          return;
        }
        if (cur->data.n() >= batch_reads) {
          cur->raw_offset = ln.raw_offset();
          publish(std::move(cur));
          cur = take_spare();
          if (!cur) return;
          reset(*cur);
        }
      }
    } catch (const Panic &e) {
      // records parsed before the malformed one stay in the batch; the consumer decides which panic wins
      cur->data.bases.resize(cur->data.offsets.back());
      cur->last = true;
      cur->error = e.what();
      publish(std::move(cur));
    }
  }
};

BatchReader::BatchReader(const std::string &path, bool is_mate, size_t batch_reads, bool packed) : impl_(new Impl()) {
  impl_->path = path;
  impl_->is_mate = is_mate;
  impl_->batch_reads = std::max<size_t>(batch_reads, 1);
  impl_->packed = packed && packing_enabled();
  static const bool serial = getenv("NIMBLE_FASTQ_SERIAL") != nullptr;
  if (!serial && ParallelPlain::is_plain(path)) {
    impl_->plain.reset(new ParallelPlain(path, is_mate, packed && packing_enabled()));
    return;
  }
  const bool serial_gz = getenv("NIMBLE_GZIP_SERIAL") != nullptr;
  if (!serial && !serial_gz) {
    impl_->gz.reset(new ParallelGz(path, is_mate, packed && packing_enabled()));
    return;
  }
  for (int i = 0; i < 3; ++i) impl_->spare.emplace_back(new Batch());
  impl_->th = std::thread([this] { impl_->run(); });
}

BatchReader::~BatchReader() {
  {
    std::lock_guard<std::mutex> lk(impl_->mu);
    impl_->stop = true;
    impl_->cv.notify_all();
  }
  if (impl_->th.joinable()) impl_->th.join();
}

std::unique_ptr<BatchReader::Batch> BatchReader::next() {
  if (impl_->plain) return impl_->plain->next();
  if (impl_->gz) return impl_->gz->next();
  std::unique_lock<std::mutex> lk(impl_->mu);
  impl_->cv.wait(lk, [&] { return !impl_->ready.empty(); });
  std::unique_ptr<Batch> b = std::move(impl_->ready.front());
  impl_->ready.pop_front();
  lk.unlock();
  b->pin();
  return b;
}

void BatchReader::recycle(std::unique_ptr<Batch> b) {
  if (impl_->plain) {
    impl_->plain->recycle(std::move(b));
    return;
  }
  if (impl_->gz) {
    impl_->gz->recycle(std::move(b));
    return;
  }
  std::lock_guard<std::mutex> lk(impl_->mu);
  impl_->spare.push_back(std::move(b));
  impl_->cv.notify_all();
}

}  // namespace fastq
}  // namespace parse

namespace process {
namespace fastq {

namespace {

struct NeedWholeFile {};  // a later batch holds a read longer than the stream was opened for

uint64_t file_size(const std::string &p) {
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return 0;
  fseek(f, 0, SEEK_END);
  long s = ftell(f);
  fclose(f);
  return s > 0 ? (uint64_t)s : 0;
}

// Length of the first record's sequence line and the bytes that record takes in the (inflated) file: what the call can be
// opened with before the first parsed batch is there.  zlib's gz layer reads plain files as they are.  false = nothing usable.
bool peek_first_record(const std::string &path, uint32_t &seq_len, uint32_t &record_bytes, bool &gz) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[1 << 16];
  const int n = gzread(f, buf, sizeof buf);
  gz = gzdirect(f) == 0;
  gzclose(f);
  if (n <= 0) return false;
  int nl[4], k = 0;
  for (int i = 0; i < n && k < 4; ++i)
    if (buf[i] == '\n') nl[k++] = i;
  if (k < 4 || buf[0] != '@') return false;
  int len = nl[1] - nl[0] - 1;
  if (len > 0 && buf[nl[1] - 1] == '\r') --len;
  if (len <= 0) return false;
  seq_len = (uint32_t)len;
  record_bytes = (uint32_t)(nl[3] + 1);
  return true;
}

// One input file as a stream of records: the current batch and how much of it has been consumed.
struct Cursor {
  parse::fastq::BatchReader rd;
  std::unique_ptr<parse::fastq::BatchReader::Batch> b;
  // the two batches before the current one: their copies to the device may still run (NIMBLE_MEM_HOST_PINNED: buffers stay
  // untouched until the second following append has returned), so a batch goes back to the reader two steps late
  std::unique_ptr<parse::fastq::BatchReader::Batch> held, held2;
  uint64_t used = 0;
  Cursor(const std::string &path, bool is_mate, size_t batch_reads, bool packed) : rd(path, is_mate, batch_reads, packed) {}
  uint64_t avail() const { return b ? b->data.n() - used : 0; }
  bool at_end() const { return b && b->last && used == b->data.n(); }  // the file's event (EOF or bad record) is next
  void fill() {  // make records available unless the file is at its event
    while (!b || (used == b->data.n() && !b->last)) {
      if (held2) rd.recycle(std::move(held2));
      held2 = std::move(held);
      held = std::move(b);
      b = rd.next();
      used = 0;
    }
  }
};

// The reference's loop pulls R1 record i, then R2 record i (align.rs:513-541): the first file to hit an event
// (end of file or a malformed record) at the smaller record index decides how the run ends.
void streamed(const std::vector<std::string> &input_files,
              std::vector<std::unique_ptr<align::PseudoAligner>> &reference_indices,
              const std::vector<reference_library::Reference> &references,
              const std::vector<align::AlignFilterConfig> &aligner_configs,
              const std::vector<std::string> &output_paths, size_t batch_reads) {
  const bool paired = input_files.size() > 1;
  Cursor c1(input_files.at(0), false, batch_reads, true);  // (batches travel packed: nimble_stream_append_packed)
  std::unique_ptr<Cursor> c2;
  if (paired) c2.reset(new Cursor(input_files[1], true, batch_reads, true));

  std::vector<std::unique_ptr<align::CallStream>> streams;
  uint32_t stream_max_len = 0;
  const std::string lengths = "Error -- read and reverse read files do not have matching lengths: ";
  // where the consumer's time goes (NIMBLE_HOST_TIMING): waiting for parsed batches, opening the streams, appending
  double t_fill = 0, t_open = 0, t_append = 0, t_finish = 0;
  uint64_t n_appends = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  // The call is opened (the context's arrays allocated for the whole file: ~13 ms) while the readers parse their first
  // batches, from a look at the first record of each file: reads of that length, as many as the file's size suggests.  A
  // first batch that holds longer reads takes the whole-file path as before; a capacity that proves too small grows.
  if (!getenv("NIMBLE_FASTQ_LATE_OPEN")) {
    uint32_t l1 = 0, rb1 = 0, l2 = 0, rb2 = 0;
    bool gz1 = false, gz2 = false;
    if (peek_first_record(input_files[0], l1, rb1, gz1) && (!paired || peek_first_record(input_files[1], l2, rb2, gz2))) {
      const auto to = now();
      stream_max_len = std::max<uint32_t>(32, (std::max(l1, l2) + 31u) / 32u * 32u);
      // (a .gz holds about four times its size in text)
      const uint64_t cap = (uint64_t)((double)file_size(input_files[0]) * (gz1 ? 4.0 : 1.0) / (double)rb1 * 1.05) + 1024;
      for (size_t i = 0; i < reference_indices.size(); ++i)
        streams.emplace_back(new align::CallStream(*reference_indices[i], aligner_configs.at(i), paired, stream_max_len, cap));
      t_open += secs(to, now());
    }
  }
  for (;;) {
    const auto tf = now();
    c1.fill();
    if (paired) c2->fill();
    t_fill += secs(tf, now());
    const uint64_t n = paired ? std::min(c1.avail(), c2->avail()) : c1.avail();
    if (streams.empty()) {
      const auto to = now();
      struct Lap {
        double &acc;
        std::chrono::steady_clock::time_point t0;
        ~Lap() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
      } lap{t_open, to};
      const uint32_t ml = std::max<uint32_t>(c1.b->data.max_len, paired ? c2->b->data.max_len : 0u);
      stream_max_len = std::max<uint32_t>(32, (ml + 31u) / 32u * 32u);
      uint64_t cap = n;  // capacity from the bytes the first batch took on disk
      if (!c1.b->last && c1.b->raw_offset > 0 && c1.b->data.n() > 0)
        cap = (uint64_t)((double)file_size(input_files[0]) / (double)c1.b->raw_offset * (double)c1.b->data.n() * 1.05) +
              1024;
      for (size_t i = 0; i < reference_indices.size(); ++i)
        streams.emplace_back(new align::CallStream(*reference_indices[i], aligner_configs.at(i), paired, stream_max_len,
                                                   std::max<uint64_t>(cap, n)));
    }
    if (c1.b->data.max_len > stream_max_len || (paired && c2->b->data.max_len > stream_max_len)) throw NeedWholeFile();
    if (n) {
      // offsets are absolute inside each batch's buffer: a slice is the same buffer with a later offsets pointer
      align::ReadBatch a, m;
      a.bases = c1.b->data.bases.data();
      a.offsets = c1.b->data.offsets.data() + c1.used;
      a.n = n;
      a.max_len = c1.b->data.max_len;
      a.pinned = c1.b->pinned[0] != nullptr;  // (the small offsets array may be pageable: HIP stages such a copy before it returns)
      if (c1.b->stride) {  // (page-locked or not: a pageable buffer is copied before the append returns)
        a.words = c1.b->words.data() + c1.used * (uint64_t)c1.b->stride;
        a.lens = c1.b->lens.data() + c1.used;
        a.stride = c1.b->stride;
      }
      if (paired) {
        m.bases = c2->b->data.bases.data();
        m.offsets = c2->b->data.offsets.data() + c2->used;
        m.n = n;
        m.max_len = c2->b->data.max_len;
        m.pinned = c2->b->pinned[0] != nullptr;
        if (c2->b->stride) {
          m.words = c2->b->words.data() + c2->used * (uint64_t)c2->b->stride;
          m.lens = c2->b->lens.data() + c2->used;
          m.stride = c2->b->stride;
        }
      }
      const auto ta = now();
      for (auto &st : streams) st->append(a, paired ? &m : nullptr);
      t_append += secs(ta, now());
      ++n_appends;
      c1.used += n;
      if (paired) c2->used += n;
      continue;  // look again: one of the files may simply need its next batch
    }
    // no record pair is available: a file is at its event.  R1 is pulled first, so its event decides a tie.
    if (c1.at_end()) {
      if (!c1.b->error.empty()) throw Panic(c1.b->error);
      break;  // R1 ended: R2's remaining records are never pulled
    }
    // R1 still has a record, so R2 is at its event: a missing mate or a malformed one
    throw Panic(c2->b->error.empty() ? lengths : c2->b->error);
  }
  const auto tfin = now();
  for (size_t i = 0; i < streams.size(); ++i) {
    align::CallOutput res = streams[i]->finish(references.at(i));
    res.materialize();
    std::sort(res.rows.begin(), res.rows.end(),
              [](const align::ScoreRow &a, const align::ScoreRow &b) { return a.first < b.first; });
    utils::write_to_tsv(res.rows, output_paths.at(i));
  }
  t_finish = secs(tfin, now());
  if (getenv("NIMBLE_HOST_TIMING"))
    fprintf(stderr, "[nimble host] consumer: %.3f s waiting for parsed batches, %.3f s opening the call, %.3f s in %llu appends, "
            "%.3f s finishing (tail of the call, coercion, TSV)\n", t_fill, t_open, t_append, (unsigned long long)n_appends, t_finish);
}

void whole_file(const std::vector<std::string> &input_files,
                std::vector<std::unique_ptr<align::PseudoAligner>> &reference_indices,
                const std::vector<reference_library::Reference> &references,
                const std::vector<align::AlignFilterConfig> &aligner_configs,
                const std::vector<std::string> &output_paths) {
  // The reference re-opens the file(s) for every library (process/fastq.rs:15-23); parsing once and
  // re-using the in-memory reads is equivalent.
  // The reference pulls one R1 record, then one R2 record, and panics at the first that fails (align.rs:511-541): R2 is
  // never read beyond R1's last good record, and of two faults the one with the smaller record index fires (R1 first at
  // equal index).  So: R1 up to its first malformed record, R2 for at most that many records, then the faults in order.
  std::string err1, err2;
  parse::fastq::FastqData r1 = parse::fastq::read_fastq_lazy(input_files.at(0), false, ~0ULL, &err1);
  parse::fastq::FastqData r2;
  const bool paired = input_files.size() > 1;
  if (paired) {
    r2 = parse::fastq::read_fastq_lazy(input_files[1], true, r1.n(), &err2);
    if (r2.n() < r1.n()) {
      if (!err2.empty()) throw Panic(err2);
      throw Panic("Error -- read and reverse read files do not have matching lengths: ");
    }
  }
  if (!err1.empty()) throw Panic(err1);
  for (size_t i = 0; i < reference_indices.size(); ++i) {
    align::ReadBatch b1;
    b1.bases = r1.bases.data();
    b1.offsets = r1.offsets.data();
    b1.n = r1.n();
    b1.max_len = r1.max_len;
    align::ReadBatch b2;
    if (paired) {
      // the reference walks R1 and pulls one R2 record per R1 record (align.rs:537-541): extra R2
      // records are ignored, missing ones panic
      if (r2.n() < r1.n()) throw Panic("Error -- read and reverse read files do not have matching lengths: ");
      b2.bases = r2.bases.data();
      b2.offsets = r2.offsets.data();
      b2.n = r1.n();
      b2.max_len = r2.max_len;
    }
    align::CallOutput res =
        score::call(b1, paired ? &b2 : nullptr, *reference_indices[i], references.at(i), aligner_configs.at(i));
    utils::write_to_tsv(res.rows, output_paths.at(i));
  }
}

}  // namespace

void process(const std::vector<std::string> &input_files,
             std::vector<std::unique_ptr<align::PseudoAligner>> &reference_indices,
             const std::vector<reference_library::Reference> &references,
             const std::vector<align::AlignFilterConfig> &aligner_configs,
             const std::vector<std::string> &output_paths) {
  // NIMBLE_FASTQ_BATCH: records per ingest batch (0 = read the whole file first, no overlap)
  size_t batch = 1u << 19;
  if (const char *e = getenv("NIMBLE_FASTQ_BATCH")) batch = (size_t)strtoull(e, nullptr, 10);
  if (batch == 0 || reference_indices.empty()) {
    const auto t0 = std::chrono::steady_clock::now();
    whole_file(input_files, reference_indices, references, aligner_configs, output_paths);
    if (getenv("NIMBLE_HOST_TIMING"))
      fprintf(stderr, "[nimble host] fastq pipeline (parse + device + tsv) %.3f s, whole file first\n",
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  try {
    streamed(input_files, reference_indices, references, aligner_configs, output_paths, batch);
  } catch (const NeedWholeFile &) {
    whole_file(input_files, reference_indices, references, aligner_configs, output_paths);
  }
  if (getenv("NIMBLE_HOST_TIMING"))
    fprintf(stderr, "[nimble host] fastq pipeline (parse + device + tsv) %.3f s, ingest batch %zu records\n",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), batch);
}

namespace {

// run f(rank) on one thread per rank and wait for all of them; the first exception is re-thrown here
template <class F>
void on_every_rank(int world, F f) {
  // (the ranks meet in collectives: all of them run, or none does -- csrc/threads.h)
  if (!threads::run_all_or_none((unsigned)world, [&](unsigned r) { f((int)r); }))
    throw Panic("could not start one thread per rank (thread limit reached)");
}

void check_dev(int rc, const char *what) {
  if (rc != 0) throw Panic(std::string(what) + ": " + nimble_last_error());
}

}  // namespace

void process_sharded(const std::vector<std::string> &input_files,
                     std::vector<std::vector<std::unique_ptr<align::PseudoAligner>>> &reference_indices,
                     const std::vector<reference_library::Reference> &references,
                     const std::vector<align::AlignFilterConfig> &aligner_configs,
                     const std::vector<std::string> &output_paths, const std::vector<int> &devices) {
  const int W = (int)devices.size();
  if (W < 1) throw Panic("process_sharded: no device given");
  const auto t0 = std::chrono::steady_clock::now();
  nimble_comm *comm = nullptr;
  check_dev(nimble_comm_create(devices.data(), W, &comm), "nimble_comm_create");
  struct CommGuard {
    nimble_comm *c;
    ~CommGuard() { nimble_comm_free(c); }
  } guard{comm};
  size_t batch = 1u << 19;
  if (const char *e = getenv("NIMBLE_FASTQ_BATCH")) batch = std::max<size_t>((size_t)strtoull(e, nullptr, 10), 1024);
  const bool paired = input_files.size() > 1;
  const std::string lengths = "Error -- read and reverse read files do not have matching lengths: ";
  // one library after the other, each over the whole input (src/process/fastq.rs:15)
  for (size_t li = 0; li < reference_indices.size(); ++li) {
    auto &ranks = reference_indices[li];
    if ((int)ranks.size() != W) throw Panic("process_sharded: one index per rank is needed");
    nimble_align_params prm;
    align::device_params(aligner_configs.at(li), &prm);
    uint32_t max_len = 0;
    {
      Cursor c1(input_files.at(0), false, batch, true);  // (batches travel packed: nimble_sharded_append_packed)
      std::unique_ptr<Cursor> c2;
      if (paired) c2.reset(new Cursor(input_files[1], true, batch, true));
      bool begun = false;
      // a failure between begin and end leaves no rank with an open call (the contexts are used again by the next library)
      struct OpenGuard {
        nimble_comm *comm;
        int W;
        bool armed;
        ~OpenGuard() {
          if (armed)
            for (int r = 0; r < W; ++r) (void)nimble_sharded_abort(comm, r);
        }
      } open_guard{comm, W, false};
      for (;;) {
        c1.fill();
        if (paired) c2->fill();
        const uint64_t n = paired ? std::min(c1.avail(), c2->avail()) : c1.avail();
        const uint32_t ml = std::max<uint32_t>(c1.b->data.max_len, paired ? c2->b->data.max_len : 0u);
        if (!begun) {
          max_len = std::max<uint32_t>(32, (ml + 31u) / 32u * 32u);
          on_every_rank(W, [&](int r) {
            check_dev(nimble_sharded_begin(comm, r, ranks[(size_t)r]->ctx(), &prm, paired ? 1 : 0, max_len),
                      "nimble_sharded_begin");
          });
          begun = true;
          open_guard.armed = true;
        }
        if (ml > max_len) {
          // a later batch holds a longer read: the records every rank has kept are widened in place, nothing is re-read
          max_len = (ml + 31u) / 32u * 32u;
          on_every_rank(W, [&](int r) { check_dev(nimble_sharded_grow(comm, r, max_len), "nimble_sharded_grow"); });
        }
        if (n) {
          const uint64_t *o1 = c1.b->data.offsets.data() + c1.used;
          const uint64_t *o2 = paired ? c2->b->data.offsets.data() + c2->used : nullptr;
          const uint8_t *b1 = c1.b->data.bases.data();
          const uint8_t *b2 = paired ? c2->b->data.bases.data() : nullptr;
          // one round: rank r takes reads [n r / W, n (r + 1) / W) of the slice -- as the 2-bit words the parser threads
          // made of them when the batch carries them (a quarter of the bytes over the link), else as text -- and the
          // packed reads go to their owners
          const bool words = c1.b->stride != 0 && (!paired || c2->b->stride != 0);
          on_every_rank(W, [&](int r) {
            const uint64_t lo = n * (uint64_t)r / (uint64_t)W, hi = n * (uint64_t)(r + 1) / (uint64_t)W;
            if (words) {
              const uint64_t a1 = c1.used + lo, a2 = paired ? c2->used + lo : 0;
              check_dev(nimble_sharded_append_packed(comm, r, c1.b->words.data() + a1 * (uint64_t)c1.b->stride,
                                                     c1.b->lens.data() + a1, c1.b->stride,
                                                     paired ? c2->b->words.data() + a2 * (uint64_t)c2->b->stride : nullptr,
                                                     paired ? c2->b->lens.data() + a2 : nullptr, paired ? c2->b->stride : 0u,
                                                     hi - lo),
                        "nimble_sharded_append_packed");
            } else {
              check_dev(nimble_sharded_append(comm, r, b1, o1 + lo, b2, paired ? o2 + lo : nullptr, hi - lo, 0,
                                              NIMBLE_MEM_HOST),
                        "nimble_sharded_append");
            }
          });
          c1.used += n;
          if (paired) c2->used += n;
          continue;
        }
        if (c1.at_end()) {
          if (!c1.b->error.empty()) throw Panic(c1.b->error);
          break;
        }
        throw Panic(c2->b->error.empty() ? lengths : c2->b->error);
      }
      open_guard.armed = false;
    }
    // every rank: the call over the reads it owns, its rows; the callsets are agreed by content (this is one
    // process: a shared dictionary), the counts summed by an all-reduce of one int64 vector
    std::mutex mu;
    std::map<std::vector<std::string>, size_t> dict;
    std::vector<std::vector<align::ScoreRow>> rows((size_t)W);
    on_every_rank(W, [&](int r) {
      uint64_t owned = 0;
      check_dev(nimble_sharded_end(comm, r, &owned), "nimble_sharded_end");
      align::CallOutput out = align::end_calls(owned, *ranks[(size_t)r], references.at(li), aligner_configs.at(li), 0);
      out.materialize();
      rows[(size_t)r] = std::move(out.rows);
      std::lock_guard<std::mutex> lk(mu);
      for (const auto &row : rows[(size_t)r]) dict.emplace(row.first, 0);
    });
    size_t k = 0;
    for (auto &kv : dict) kv.second = k++;  // std::map order == Vec<String> order == the order of the TSV
    std::vector<std::vector<int64_t>> vec((size_t)W, std::vector<int64_t>(dict.size(), 0));
    on_every_rank(W, [&](int r) {
      for (const auto &row : rows[(size_t)r]) vec[(size_t)r][dict.at(row.first)] += row.second;
      check_dev(nimble_counts_allreduce_host(comm, r, vec[(size_t)r].data(), vec[(size_t)r].size()),
                "nimble_counts_allreduce_host");
    });
    std::vector<align::ScoreRow> result;
    for (const auto &kv : dict) result.emplace_back(kv.first, (int32_t)vec[0][kv.second]);
    utils::write_to_tsv(result, output_paths.at(li));
  }
  if (getenv("NIMBLE_HOST_TIMING"))
    fprintf(stderr, "[nimble host] fastq pipeline (parse + device + tsv) %.3f s, %d ranks%s\n",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), W,
            nimble_comm_uses_rccl(comm) ? " over RCCL" : " on one device");
}

}  // namespace fastq
}  // namespace process
}  // namespace nimble
