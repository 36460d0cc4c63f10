// fastq.cpp -- FASTQ ingest and the FASTQ pipeline driver.
// Mirrors src/parse/fastq.rs:8-43 (niffler + bio::io::fastq::Reader -> DnaString per record) and
// src/process/fastq.rs:7-30 (one score::call per library over the whole file, then write_to_tsv).
#include <zlib.h>

#include <cstdio>
#include <cstring>

#include "nimble_host.hpp"

namespace nimble {
namespace parse {
namespace fastq {

namespace {

// whole-file reader with gzip auto-detection by magic bytes (what niffler::from_path does);
// gzread also passes plain files through unchanged
std::vector<uint8_t> slurp(const std::string &path) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) throw Panic("Error -- could not determine compression format for " + path);
  gzbuffer(f, 1 << 20);
  std::vector<uint8_t> data;
  std::vector<uint8_t> chunk(1 << 22);
  for (;;) {
    int got = gzread(f, chunk.data(), (unsigned)chunk.size());
    if (got < 0) {
      gzclose(f);
      throw Panic("Error -- could not determine compression format for " + path);
    }
    if (got == 0) break;
    data.insert(data.end(), chunk.begin(), chunk.begin() + got);
  }
  gzclose(f);
  return data;
}

struct Lines {
  const std::vector<uint8_t> &d;
  size_t p = 0;
  explicit Lines(const std::vector<uint8_t> &data) : d(data) {}
  // read_line: returns false at EOF; [b, e) excludes the line terminator; `raw_empty` mirrors an empty
  // String after read_line (true only at EOF)
  bool next(size_t &b, size_t &e) {
    if (p >= d.size()) return false;
    b = p;
    const uint8_t *nl = (const uint8_t *)memchr(d.data() + p, '\n', d.size() - p);
    size_t end = nl ? (size_t)(nl - d.data()) : d.size();
    p = nl ? end + 1 : end;
    e = end;
    return true;
  }
};

size_t trim_end(const std::vector<uint8_t> &d, size_t b, size_t e) {
  while (e > b && (d[e - 1] == ' ' || d[e - 1] == '\t' || d[e - 1] == '\r' || d[e - 1] == '\n' || d[e - 1] == '\f' ||
                   d[e - 1] == '\v'))
    --e;
  return e;
}

}  // namespace

FastqData read_fastq(const std::string &path, bool is_mate) {
  const char *malformed = is_mate ? "Error -- could not parse reverse read. Input R2 data malformed."
                                  : "Error -- could not parse read. Input R1 data malformed.";
  std::vector<uint8_t> data = slurp(path);
  FastqData out;
  out.offsets.push_back(0);
  out.bases.reserve(data.size() / 2);
  Lines ln(data);
  size_t b, e;
  // bio::io::fastq::Reader::read: '@' header, sequence lines up to the '+' line, then as many quality
  // lines as there were sequence lines; sequence and quality lengths are not compared
  while (ln.next(b, e)) {
    if (e == b || data[b] != '@') throw Panic(std::string(malformed) + ": Unable to read sequence");
    size_t seq_lines = 0;
    bool more = ln.next(b, e);
    while (more && !(e > b && data[b] == '+')) {
      size_t te = trim_end(data, b, e);
      out.bases.insert(out.bases.end(), data.begin() + (long)b, data.begin() + (long)te);
      ++seq_lines;
      more = ln.next(b, e);
    }
    size_t qual_len = 0;
    for (size_t k = 0; k < seq_lines; ++k) {
      if (!ln.next(b, e)) break;
      qual_len += trim_end(data, b, e) - b;
    }
    if (qual_len == 0) throw Panic(std::string(malformed) + ": Unable to read sequence");  // IncompleteRecord
    uint64_t len = out.bases.size() - out.offsets.back();
    if (len > out.max_len) out.max_len = (uint32_t)len;
    out.offsets.push_back(out.bases.size());
  }
  return out;
}

}  // namespace fastq
}  // namespace parse

namespace process {
namespace fastq {

void process(const std::vector<std::string> &input_files,
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

}  // namespace fastq
}  // namespace process
}  // namespace nimble
