// fastq.cpp -- FASTQ ingest and the FASTQ pipeline driver.
// Mirrors src/parse/fastq.rs:8-43 (niffler + bio::io::fastq::Reader -> DnaString per record) and
// src/process/fastq.rs:7-30 (one score::call per library over the whole file, then write_to_tsv).
//
// Ingest is overlapped with the device work: one reader thread per input file inflates / reads and parses
// records into batches; the pipeline thread hands batch i to the streamed call of every library
// (align::CallStream: copy + pack + align of that batch on the GPU) while batch i+1 is being parsed.  It is
// still ONE score::call per library: dedup and counting run once over everything at the end.
#include <zlib.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

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

inline size_t trimmed(const uint8_t *b, size_t len) {
  while (len > 0 && (b[len - 1] == ' ' || b[len - 1] == '\t' || b[len - 1] == '\r' || b[len - 1] == '\n' ||
                     b[len - 1] == '\f' || b[len - 1] == '\v'))
    --len;
  return len;
}

// One record of bio::io::fastq::Reader::read appended to `out`: '@' header, sequence lines up to the '+' line,
// then as many quality lines as there were sequence lines; sequence and quality lengths are not compared.
// Returns false at a clean EOF; throws on a malformed record.
bool read_record(LineSource &ln, FastqData &out, const char *malformed) {
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
  size_t qual_len = 0;
  for (size_t k = 0; k < seq_lines; ++k) {
    if (!ln.next(b, len)) break;
    qual_len += trimmed(b, len);
  }
  if (qual_len == 0) throw Panic(std::string(malformed) + ": Unable to read sequence");  // IncompleteRecord
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

// ---- batch reader: a thread parses ahead, batches are handed over in file order ----------------------
struct BatchReader::Impl {
  std::string path;
  bool is_mate;
  size_t batch_reads;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::unique_ptr<Batch>> ready, spare;
  bool stop = false;

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

BatchReader::BatchReader(const std::string &path, bool is_mate, size_t batch_reads) : impl_(new Impl()) {
  impl_->path = path;
  impl_->is_mate = is_mate;
  impl_->batch_reads = std::max<size_t>(batch_reads, 1);
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
  std::unique_lock<std::mutex> lk(impl_->mu);
  impl_->cv.wait(lk, [&] { return !impl_->ready.empty(); });
  std::unique_ptr<Batch> b = std::move(impl_->ready.front());
  impl_->ready.pop_front();
  return b;
}

void BatchReader::recycle(std::unique_ptr<Batch> b) {
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

// The reference's loop pulls R1 record i, then R2 record i (align.rs:513-541): the first file to hit an event
// (end of file or a malformed record) at the smaller record index decides how the run ends.
void streamed(const std::vector<std::string> &input_files,
              std::vector<std::unique_ptr<align::PseudoAligner>> &reference_indices,
              const std::vector<reference_library::Reference> &references,
              const std::vector<align::AlignFilterConfig> &aligner_configs,
              const std::vector<std::string> &output_paths, size_t batch_reads) {
  using parse::fastq::BatchReader;
  const bool paired = input_files.size() > 1;
  BatchReader rd1(input_files.at(0), false, batch_reads);
  std::unique_ptr<BatchReader> rd2;
  if (paired) rd2.reset(new BatchReader(input_files[1], true, batch_reads));

  std::vector<std::unique_ptr<align::CallStream>> streams;
  uint32_t stream_max_len = 0;
  const uint64_t INF = ~0ULL;
  const std::string lengths = "Error -- read and reverse read files do not have matching lengths: ";
  for (;;) {
    std::unique_ptr<BatchReader::Batch> b1 = rd1.next(), b2;
    if (paired) b2 = rd2->next();
    const uint64_t n1 = b1->data.n(), n2 = paired ? b2->data.n() : n1;
    // record index (inside this batch) at which each file ends or turns malformed; batches that are not the
    // last one of their file are full, so both batches start at the same record
    const uint64_t e1 = b1->last ? n1 : INF, e2 = paired && b2->last ? n2 : INF;
    const uint64_t n = std::min(n1, n2);
    if (streams.empty()) {
      const uint32_t ml = std::max<uint32_t>(b1->data.max_len, paired ? b2->data.max_len : 0u);
      stream_max_len = std::max<uint32_t>(32, (ml + 31u) / 32u * 32u);
      uint64_t cap = n;  // capacity from the bytes the first batch took on disk
      if (!b1->last && b1->raw_offset > 0 && n1 > 0)
        cap = (uint64_t)((double)file_size(input_files[0]) / (double)b1->raw_offset * (double)n1 * 1.05) + 1024;
      for (size_t i = 0; i < reference_indices.size(); ++i)
        streams.emplace_back(new align::CallStream(*reference_indices[i], aligner_configs.at(i), paired, stream_max_len,
                                                   std::max<uint64_t>(cap, n)));
    }
    if (b1->data.max_len > stream_max_len || (paired && b2->data.max_len > stream_max_len)) throw NeedWholeFile();
    if (n) {
      align::ReadBatch a, m;
      a.bases = b1->data.bases.data();
      a.offsets = b1->data.offsets.data();
      a.n = n;
      a.max_len = b1->data.max_len;
      if (paired) {
        m.bases = b2->data.bases.data();
        m.offsets = b2->data.offsets.data();
        m.n = n;
        m.max_len = b2->data.max_len;
      }
      for (auto &st : streams) st->append(a, paired ? &m : nullptr);
    }
    if (e1 != INF && e1 <= e2) {  // R1 is pulled first: its end of file ends the run, its bad record panics
      if (!b1->error.empty()) throw Panic(b1->error);
      break;
    }
    if (e2 != INF) {
      if (e2 == n1) {
        // R2's event sits right behind this (full) R1 batch: R1's next record decides whether R2 is pulled at all
        std::unique_ptr<BatchReader::Batch> nb = rd1.next();
        if (nb->last && nb->data.n() == 0) {
          if (!nb->error.empty()) throw Panic(nb->error);
          break;
        }
      }
      throw Panic(b2->error.empty() ? lengths : b2->error);
    }
    rd1.recycle(std::move(b1));
    if (paired) rd2->recycle(std::move(b2));
  }
  for (size_t i = 0; i < streams.size(); ++i) {
    align::CallOutput res = streams[i]->finish(references.at(i));
    res.materialize();
    std::sort(res.rows.begin(), res.rows.end(),
              [](const align::ScoreRow &a, const align::ScoreRow &b) { return a.first < b.first; });
    utils::write_to_tsv(res.rows, output_paths.at(i));
  }
}

void whole_file(const std::vector<std::string> &input_files,
                std::vector<std::unique_ptr<align::PseudoAligner>> &reference_indices,
                const std::vector<reference_library::Reference> &references,
                const std::vector<align::AlignFilterConfig> &aligner_configs,
                const std::vector<std::string> &output_paths) {
  // The reference re-opens the file(s) for every library (process/fastq.rs:15-23); parsing once and
  // re-using the in-memory reads is equivalent.
  parse::fastq::FastqData r1 = parse::fastq::read_fastq(input_files.at(0), false);
  parse::fastq::FastqData r2;
  const bool paired = input_files.size() > 1;
  if (paired) r2 = parse::fastq::read_fastq(input_files[1], true);
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

}  // namespace fastq
}  // namespace process
}  // namespace nimble
