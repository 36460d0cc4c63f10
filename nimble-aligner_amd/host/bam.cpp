// bam.cpp -- the BAM pipeline of the reference, without htslib: a BGZF + BAM record reader on zlib, the reference's UMI
// grouping, and process::bam::process on top of get_calls_umis (many UMI groups per device call).
//
//   parse::bam::Reader            what rust_htslib::bam::Reader::records() yields to the reference (the fields it reads)
//   parse::bam::SortedBamReader   src/parse/sorted_bam_reader.rs:6-186
//   parse::bam::UMIReader         src/parse/bam.rs:51-288 (BAM_FIELDS_TO_REPORT :9-49, strip_nonbio_regions :256-288)
//   process::bam::process         src/process/bam.rs:45-243 (+ align_umi_to_libraries :305-405, the TSV of :92-121)
//
// Quirks of the reference that are kept because they decide what is written:
//   * the last UMI group of a file is collected but never sent to the aligner once any group was sent before
//     (process/bam.rs:163-178: `final_umi && has_aligned` breaks before the send);
//   * the records of the LAST UMI of the file are not sorted by cell barcode (sorted_bam_reader.rs:33-116: the sort runs only
//     when the next UMI shows up);
//   * an unpaired read and its SKIP_ALIGN dummy come out dummy first (filter_paired_reads :138-150: neither is first in
//     template, so the pair is swapped);
//   * the columns headed r1_* carry the mate's BAM fields and filter reason, r2_* the first read's (process/bam.rs:103-116);
//   * integer tags (NH, HI, AS, nM) are reported empty: only Aux::String values are taken (parse/bam.rs:186-189).
//   * a UMI whose records all fall to the pairing filter ENDS the input: SortedBamReader::next reports the empty buffer as
//     an error, and UMIReader takes any error for the end of the file (sorted_bam_reader.rs:169-186, parse/bam.rs:113-117).
// The reference's rows leave its consumer pool in no fixed order; here they are written in UMI order, callsets sorted (the
// order of score::call), then the pairs without a call.
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <chrono>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <unordered_map>
#include <unordered_set>

#include "../../include/nimble_hip.h"
#include "nimble_host.hpp"

namespace nimble {
namespace parse {
namespace bam {

const char *const BAM_FIELDS_TO_REPORT[38] = {
    "QNAME", "QUAL", "REVERSE", "MATE_REVERSE", "PAIRED", "PROPER_PAIRED", "PAIR_ORIENTATION", "UNMAPPED",
    "MATE_UNMAPPED", "FIRST_IN_TEMPLATE", "LAST_IN_TEMPLATE", "STRAND", "MAPQ", "POS", "MATE_POS", "SEQ", "SEQ_LEN",
    "INSERT_SIZE", "QUALITY_FAILED", "SECONDARY", "DUPLICATE", "SUPPLEMENTARY", "NH", "HI", "AS", "GN", "TX", "AN", "nM",
    "fx", "RE", "CR", "CY", "CB", "UR", "UY", "UB", "SKIP_ALIGN"};

// ---- BGZF + BAM records -----------------------------------------------------------------------------------------
struct Reader::Impl {
  gzFile f = nullptr;  // BGZF is a series of gzip members: zlib's gz layer reads across them
  std::vector<uint8_t> buf;
  bool read_exact(void *dst, size_t n, bool &eof) {
    uint8_t *p = static_cast<uint8_t *>(dst);
    size_t got = 0;
    while (got < n) {
      const int r = gzread(f, p + got, (unsigned)std::min<size_t>(n - got, 1u << 30));
      if (r < 0) throw Panic("Error -- could not read BAM file (corrupt BGZF block)");
      if (r == 0) {
        int err = Z_OK;
        (void)gzerror(f, &err);
        if (err != Z_OK && err != Z_STREAM_END) throw Panic("0: Found truncated record");  // the file ends inside a block
        break;
      }
      got += (size_t)r;
    }
    eof = got == 0;
    return got == n;
  }
};

Reader::Reader(const std::string &path) : impl_(new Impl()) {
  impl_->f = gzopen(path.c_str(), "rb");
  if (!impl_->f) throw Panic("Error -- could not open BAM file " + path);
  gzbuffer(impl_->f, 1u << 20);
  bool eof = false;
  char magic[4];
  int32_t l_text = 0, n_ref = 0;
  if (!impl_->read_exact(magic, 4, eof) || memcmp(magic, "BAM\1", 4) != 0) throw Panic("Error -- " + path + " is not a BAM file");
  if (!impl_->read_exact(&l_text, 4, eof) || l_text < 0) throw Panic("Error -- truncated BAM header");
  impl_->buf.resize((size_t)l_text);
  if (l_text && !impl_->read_exact(impl_->buf.data(), (size_t)l_text, eof)) throw Panic("Error -- truncated BAM header");
  if (!impl_->read_exact(&n_ref, 4, eof) || n_ref < 0) throw Panic("Error -- truncated BAM header");
  for (int32_t i = 0; i < n_ref; ++i) {
    int32_t l_name = 0;
    if (!impl_->read_exact(&l_name, 4, eof) || l_name < 0) throw Panic("Error -- truncated BAM header");
    impl_->buf.resize((size_t)l_name + 4);
    if (!impl_->read_exact(impl_->buf.data(), (size_t)l_name + 4, eof)) throw Panic("Error -- truncated BAM header");
  }
}

Reader::~Reader() {
  if (impl_->f) gzclose(impl_->f);
}

bool Reader::next(Record &r) {
  bool eof = false;
  int32_t block = 0;
  if (!impl_->read_exact(&block, 4, eof)) {
    if (eof) return false;
    throw Panic("0: Found truncated record");  // parse/bam.rs:136-139
  }
  if (block < 32) throw Panic("0: Found truncated record");
  impl_->buf.resize((size_t)block);
  if (!impl_->read_exact(impl_->buf.data(), (size_t)block, eof)) throw Panic("0: Found truncated record");
  const uint8_t *p = impl_->buf.data();
  auto i32 = [&](size_t o) { int32_t v; memcpy(&v, p + o, 4); return v; };
  auto u16 = [&](size_t o) { uint16_t v; memcpy(&v, p + o, 2); return v; };
  r.tid = i32(0);
  r.pos = i32(4);
  const uint32_t l_read_name = p[8];
  r.mapq = p[9];
  const uint32_t n_cigar = u16(12);
  r.flag = u16(14);
  const uint32_t l_seq = (uint32_t)i32(16);
  r.mtid = i32(20);
  r.mpos = i32(24);
  r.tlen = i32(28);
  size_t o = 32;
  if (o + l_read_name + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq > (size_t)block) throw Panic("0: Found truncated record");
  r.qname.assign(reinterpret_cast<const char *>(p + o), l_read_name ? l_read_name - 1 : 0);  // NUL-terminated
  o += l_read_name + 4ull * n_cigar;
  static const char code[] = "=ACMGRSVTWYHKDBN";
  r.seq.resize(l_seq);
  for (uint32_t i = 0; i < l_seq; ++i) r.seq[i] = code[(p[o + i / 2] >> (i & 1 ? 0 : 4)) & 15];
  o += (l_seq + 1) / 2;
  r.qual.assign(reinterpret_cast<const char *>(p + o), l_seq);
  o += l_seq;
  r.aux.assign(p + o, p + block);
  return true;
}

// the value of a 'Z' tag (what rust-htslib hands out as Aux::String); any other type: not a string
bool Record::aux_string(const char *tag, std::string &out) const {
  if (strlen(tag) != 2) return false;  // rust-htslib: a tag has two characters, anything else is an error
  size_t o = 0;
  const size_t n = aux.size();
  while (o + 3 <= n) {
    const char t0 = (char)aux[o], t1 = (char)aux[o + 1], ty = (char)aux[o + 2];
    o += 3;
    size_t len = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': len = 1; break;
      case 's': case 'S': len = 2; break;
      case 'i': case 'I': case 'f': len = 4; break;
      case 'Z': case 'H': {
        size_t e = o;
        while (e < n && aux[e] != 0) ++e;
        if (e >= n) return false;
        if (t0 == tag[0] && t1 == tag[1]) {
          if (ty != 'Z') return false;
          out.assign(reinterpret_cast<const char *>(aux.data() + o), e - o);
          return true;
        }
        o = e + 1;
        continue;
      }
      case 'B': {
        if (o + 5 > n) return false;
        const char sub = (char)aux[o];
        uint32_t cnt;
        memcpy(&cnt, aux.data() + o + 1, 4);
        const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        len = 5 + w * (size_t)cnt;
        break;
      }
      default: return false;
    }
    if (t0 == tag[0] && t1 == tag[1]) return false;  // present, but not a string
    o += len;
  }
  // tags pushed by the reader itself (SKIP_ALIGN is no legal two-character tag; rust-htslib accepted it on push)
  return false;
}

// rust-htslib Record::read_pair_orientation
// The fifteen two-character tags of BAM_FIELDS_TO_REPORT in ONE walk over the aux data, each with the answer aux_string
// would give (the first entry of a tag decides: a string gives its value, any other type gives nothing, and nothing behind
// a malformed entry is seen).  have[k] = tag k is a string; out[k] its value.
static const char *const REPORT_TAGS[15] = {"NH", "HI", "AS", "GN", "TX", "AN", "nM", "fx", "RE", "CR", "CY", "CB", "UR", "UY", "UB"};
static void aux_report_tags(const Record &r, std::string *out, bool *have) {
  bool seen[15];
  for (int k = 0; k < 15; ++k) seen[k] = have[k] = false;
  const std::vector<uint8_t> &aux = r.aux;
  size_t o = 0;
  const size_t n = aux.size();
  while (o + 3 <= n) {
    const char t0 = (char)aux[o], t1 = (char)aux[o + 1], ty = (char)aux[o + 2];
    o += 3;
    int k = -1;
    for (int q = 0; q < 15; ++q)
      if (REPORT_TAGS[q][0] == t0 && REPORT_TAGS[q][1] == t1) {
        k = q;
        break;
      }
    size_t len = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': len = 1; break;
      case 's': case 'S': len = 2; break;
      case 'i': case 'I': case 'f': len = 4; break;
      case 'Z': case 'H': {
        size_t e = o;
        while (e < n && aux[e] != 0) ++e;
        if (e >= n) return;
        if (k >= 0 && !seen[k]) {
          seen[k] = true;
          if (ty == 'Z') {
            have[k] = true;
            out[k].assign(reinterpret_cast<const char *>(aux.data() + o), e - o);
          }
        }
        o = e + 1;
        continue;
      }
      case 'B': {
        if (o + 5 > n) return;
        const char sub = (char)aux[o];
        uint32_t cnt;
        memcpy(&cnt, aux.data() + o + 1, 4);
        const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        len = 5 + w * (size_t)cnt;
        break;
      }
      default: return;
    }
    if (k >= 0) seen[k] = true;  // present, but not a string
    o += len;
  }
}

static const char *pair_orientation(const Record &r) {
  const bool paired = r.flag & 0x1, unmapped = r.flag & 0x4, mate_unmapped = r.flag & 0x8;
  if (!(paired && !unmapped && !mate_unmapped && r.tid == r.mtid)) return "None";
  if (r.pos == r.mpos) return "None";
  const bool rev = r.flag & 0x10, mrev = r.flag & 0x20, first = r.flag & 0x40;
  int64_t p1, p2;
  bool f1, f2;
  if (first) { p1 = r.pos; p2 = r.mpos; f1 = !rev; f2 = !mrev; }
  else { p1 = r.mpos; p2 = r.pos; f1 = !mrev; f2 = !rev; }
  if (p1 < p2) return f1 ? (f2 ? "F1F2" : "F1R2") : (f2 ? "R1F2" : "R1R2");
  return f2 ? (f1 ? "F2F1" : "F2R1") : (f1 ? "R2F1" : "R2R1");
}

// ---- SortedBamReader (sorted_bam_reader.rs) ---------------------------------------------------------------------
SortedBamReader::SortedBamReader(const std::string &path, bool force_bam_paired)
    : reader_(path), force_bam_paired_(force_bam_paired) {}

static std::string umi_of(const Record &r) {  // corrected UB, else raw UR (sorted_bam_reader.rs:57-65)
  std::string u;
  if (r.aux_string("UB", u)) return u;
  if (r.aux_string("UR", u)) return u;
  throw Panic("Error -- Could not read UMI.");
}

void SortedBamReader::fill_buffer() {
  buffer_.clear();
  for (auto &r : next_records_) buffer_.push_back(std::move(r));
  next_records_.clear();
  current_umi_ = next_umi_;
  Record rec;
  while (reader_.next(rec)) {
    if (!(rec.flag & 0x1) && force_bam_paired_) continue;
    std::string cb;
    if (!rec.aux_string("CB", cb)) continue;
    const std::string umi = umi_of(rec);
    if (umi == "AAAAAAAAAA") continue;
    if (current_umi_.empty()) current_umi_ = umi;
    if (current_umi_ != umi) {
      // records of one UMI ordered by cell barcode (a stable sort, like Vec::sort_by)
      // (the barcode every buffered record was checked to have, read once when it came in)
      std::stable_sort(buffer_.begin(), buffer_.end(), [](const Record &a, const Record &b) { return a.cb < b.cb; });
      rec.cb = std::move(cb);
      next_records_.push_back(std::move(rec));
      next_umi_ = umi;
      return;
    }
    rec.cb = std::move(cb);
    buffer_.push_back(std::move(rec));
    rec = Record();
  }
}

void SortedBamReader::add_dummy_paired_reads() {
  std::vector<Record> out;
  out.reserve(buffer_.size());
  for (Record &r : buffer_) {
    r.skip_align = "FALSE";
    if (!(r.flag & 0x1)) {
      Record dummy = r;
      dummy.skip_align = "TRUE";
      out.push_back(std::move(r));
      out.push_back(std::move(dummy));
    } else {
      out.push_back(std::move(r));
    }
  }
  buffer_.swap(out);
}

void SortedBamReader::filter_paired_reads() {
  std::vector<Record> out;
  std::set<std::string> seen;
  size_t i = 0;
  while (i < buffer_.size()) {
    if (i + 1 >= buffer_.size()) break;
    if (buffer_[i].qname == buffer_[i + 1].qname) {
      seen.insert(buffer_[i].qname);
      if (buffer_[i].flag & 0x40) {
        out.push_back(std::move(buffer_[i]));
        out.push_back(std::move(buffer_[i + 1]));
      } else {
        out.push_back(std::move(buffer_[i + 1]));
        out.push_back(std::move(buffer_[i]));
      }
      i += 2;
    } else {
      puts("Warning: Unpaired qname!");
      if (seen.count(buffer_[i].qname))
        printf("Warning: Read with qname '\"%s\"' has been deleted but was seen before.\n", buffer_[i].qname.c_str());
      seen.insert(buffer_[i].qname);
      i += 1;
    }
  }
  buffer_.swap(out);
}

bool SortedBamReader::next(Record &out) {
  if (cursor_ >= buffer_.size()) {
    fill_buffer();
    if (!force_bam_paired_) add_dummy_paired_reads();
    filter_paired_reads();
    cursor_ = 0;
    if (buffer_.empty()) return false;  // (the reference reports BamTruncatedRecord here: its end-of-input signal)
  }
  out = std::move(buffer_[cursor_++]);
  return true;
}

// ---- UMIReader (parse/bam.rs) -----------------------------------------------------------------------------------
static const size_t CLIP_LENGTH = 13;

static std::string strip_nonbio_regions(const std::string &seq, bool rev_comp) {
  std::string s = seq;
  if (seq.size() == 124) s = rev_comp ? seq.substr(0, seq.size() - CLIP_LENGTH) : seq.substr(CLIP_LENGTH);
  // DnaString::from_acgt_bytes + to_string: upper-case A/C/G/T, anything else reads as 'A'
  for (char &c : s) {
    const char u = (char)(c & 0xDF);
    c = (u == 'A' || u == 'C' || u == 'G' || u == 'T') ? u : 'A';
  }
  return s;
}

static std::string strip_nonbio_regions_qual(const std::string &qual, bool rev_comp) {
  std::string q = qual;
  if (qual.size() == 124) q = rev_comp ? qual.substr(0, qual.size() - CLIP_LENGTH) : qual.substr(CLIP_LENGTH);
  if (rev_comp) std::reverse(q.begin(), q.end());
  return q;
}

static bool valid_utf8(const std::string &s) {
  for (unsigned char c : s)
    if (c >= 0x80) return false;  // Phred bytes are 0..93; anything else (0xFF = absent) is no ASCII
  return true;
}

UMIReader::UMIReader(const std::string &path, bool terminate_on_error, bool force_bam_paired)
    : reader_(path, force_bam_paired), terminate_on_error_(terminate_on_error) {}

bool UMIReader::next() { return !get_umi_from_bam(); }  // true = that was the final UMI

bool UMIReader::get_umi_from_bam() {
  current_umi_group = std::move(next_umi_group_);
  current_metadata_group = std::move(next_metadata_group_);
  current_umi = next_umi_;
  current_iteration_key_ = next_iteration_key_;
  current_cell_barcode = next_cell_barcode_;
  next_umi_group_.clear();
  next_metadata_group_.clear();
  next_umi_.clear();
  next_cell_barcode_.clear();
  next_iteration_key_.clear();
  Record record;
  for (;;) {
    if (!reader_.next(record)) return false;
    ++read_counter_;
    if (read_counter_ % 1000000 == 0) printf("Aligned reads %zu-%zu\n", read_counter_ - 1000000, read_counter_);
    // (one walk over the aux data for the fifteen reported tags, the UMI and the cell barcode; the 38 fields by position
    // in BAM_FIELDS_TO_REPORT -- a lookup by name per field and record was most of the pipeline's time)
    std::string tagv[15];
    bool tagh[15];
    aux_report_tags(record, tagv, tagh);
    if (!tagh[14] && !tagh[12]) throw Panic("Error -- Could not read UMI.");
    const std::string read_umi = tagh[14] ? tagv[14] : tagv[12];  // corrected UB, else raw UR
    if (!tagh[11]) throw Panic("Error Read without cell barcode, cannot excise read-mate.");
    const std::string &cb = tagv[11];
    const std::string cell = cb.size() >= 2 ? cb.substr(0, cb.size() - 2) : std::string();
    const std::string key = read_umi + cell;
    if (current_umi.empty()) current_umi = read_umi;
    if (current_iteration_key_.empty()) current_iteration_key_ = key;
    const bool rev = record.flag & 0x10;
    const std::string seq = strip_nonbio_regions(record.seq, rev);
    std::string qual = record.qual;
    if (!valid_utf8(qual)) {
      puts("QUAL parsing warning: invalid utf-8 sequence");
      qual.clear();
    }
    qual = strip_nonbio_regions_qual(qual, rev);
    std::vector<std::string> fields(38);
    auto b = [](bool v) { return std::string(v ? "true" : "false"); };
    fields[0] = record.qname;
    fields[1] = std::move(qual);
    fields[2] = b(rev);
    fields[3] = b(record.flag & 0x20);
    fields[4] = b(record.flag & 0x1);
    fields[5] = b(record.flag & 0x2);
    fields[6] = pair_orientation(record);
    fields[7] = b(record.flag & 0x4);
    fields[8] = b(record.flag & 0x8);
    fields[9] = b(record.flag & 0x40);
    fields[10] = b(record.flag & 0x80);
    fields[11] = rev ? "-" : "+";
    fields[12] = std::to_string(record.mapq);
    fields[13] = std::to_string((long long)record.pos);
    fields[14] = std::to_string((long long)record.mpos);
    fields[15] = seq;
    fields[16] = std::to_string(record.seq.size());
    fields[17] = std::to_string((long long)record.tlen);
    fields[18] = b(record.flag & 0x200);
    fields[19] = b(record.flag & 0x100);
    fields[20] = b(record.flag & 0x400);
    fields[21] = b(record.flag & 0x800);
    for (int k = 0; k < 15; ++k)
      if (tagh[k]) fields[22 + k] = std::move(tagv[k]);  // (tagv[11..14] are not used again below)
    fields[37] = record.skip_align;
    if (current_iteration_key_ == key) {
      current_umi_group.push_back(seq);
      current_metadata_group.push_back(std::move(fields));
      current_cell_barcode = cell;
    } else {
      next_umi_group_.push_back(seq);
      next_metadata_group_.push_back(std::move(fields));
      next_umi_ = read_umi;
      next_cell_barcode_ = cell;
      next_iteration_key_ = key;
      return true;
    }
  }
}

}  // namespace bam
}  // namespace parse

namespace process {
namespace bam {

// process/bam.rs:417-423
bool parse_str_as_bool(const std::string &v) {
  if (v == "true") return true;
  if (v == "false") return false;
  throw Panic("Could not parse revcomp field \"" + v + "\" as boolean");
}

// process/bam.rs:407-415
std::string reverse_comp_if_needed(const std::string &seq, bool reverse_comp) {
  return reverse_comp ? utils::revcomp(seq) : seq;
}

namespace {

std::string bam_data(const std::vector<std::string> &fields) {  // bam_data_values: all but QUAL (1) and SEQ (15)
  std::string s;
  bool first = true;
  for (size_t i = 0; i < fields.size(); ++i) {
    if (i == 1 || i == 15) continue;
    if (!first) s += '\t';
    s += fields[i];
    first = false;
  }
  return s;
}

std::string bam_header(const char *prefix) {
  std::string s;
  bool first = true;
  for (size_t i = 0; i < 38; ++i) {
    if (i == 1 || i == 15) continue;
    if (!first) s += '\t';
    s += prefix;
    s += '_';
    s += parse::bam::BAM_FIELDS_TO_REPORT[i];
    first = false;
  }
  return s;
}

// The output file as ONE gzip member (what flate2's GzEncoder writes, process/bam.rs:60-75) deflated by several threads the way
// pigz does it: the text is cut into blocks, each block is deflated raw and closed with a sync flush (the last with a
// finish), the blocks are written in order, and the member's CRC-32 is combined from the blocks'.  One zlib stream at
// level 6 wrote 60 MB of this very repetitive text per second and was the slowest stage of the pipeline.
class GzWriter {
 public:
  GzWriter(const std::string &path, unsigned threads) {
    f_ = fopen(path.c_str(), "wb");
    if (!f_) throw Panic("could not create " + path);
    static const unsigned char head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    fwrite(head, 1, 10, f_);
    for (unsigned t = 0; t < std::max(1u, threads); ++t) workers_.emplace_back([this] { work(); });
  }
  void write(const char *p, size_t n) {
    cur_.append(p, n);
    if (cur_.size() >= BLOCK) submit(false);
  }
  bool close() {  // true = everything written
    submit(true);
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return written_ == jobs_.size(); });
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
    workers_.clear();
    unsigned char tail[8];
    for (int k = 0; k < 4; ++k) {
      tail[k] = (unsigned char)(crc_ >> (8 * k));
      tail[4 + k] = (unsigned char)((uint32_t)total_ >> (8 * k));
    }
    bool ok = !failed_ && fwrite(tail, 1, 8, f_) == 8;
    ok = fclose(f_) == 0 && ok;
    f_ = nullptr;
    return ok;
  }
  ~GzWriter() {
    if (f_) (void)close();
  }

 private:
  static constexpr size_t BLOCK = 1u << 20;
  struct Job {
    std::string text, out;
    uint32_t crc = 0;
    bool last = false, done = false;
  };
  void submit(bool last) {
    std::unique_ptr<Job> j(new Job());
    j->text.swap(cur_);
    j->last = last;
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return jobs_.size() - written_ < 4 * workers_.size() + 4; });  // bounded memory
    jobs_.push_back(std::move(j));
    lk.unlock();
    cv_.notify_all();
  }
  void work() {
    for (;;) {
      Job *j = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || next_ < jobs_.size(); });
        if (next_ >= jobs_.size()) return;
        j = jobs_[next_++].get();
      }
      z_stream z;
      memset(&z, 0, sizeof z);
      bool ok = deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
      if (ok) {
        j->out.resize(deflateBound(&z, (uLong)j->text.size()) + 16);
        z.next_in = (Bytef *)j->text.data();
        z.avail_in = (uInt)j->text.size();
        z.next_out = (Bytef *)&j->out[0];
        z.avail_out = (uInt)j->out.size();
        const int rc = deflate(&z, j->last ? Z_FINISH : Z_SYNC_FLUSH);
        ok = j->last ? rc == Z_STREAM_END : (rc == Z_OK && z.avail_in == 0);
        j->out.resize(j->out.size() - z.avail_out);
        deflateEnd(&z);
      }
      j->crc = (uint32_t)crc32(0, (const Bytef *)j->text.data(), (uInt)j->text.size());
      std::unique_lock<std::mutex> lk(mu_);
      if (!ok) failed_ = true;
      j->done = true;
      // whoever completes the next block in file order writes it and the finished ones behind it
      while (written_ < jobs_.size() && jobs_[written_]->done) {
        Job &w = *jobs_[written_];
        if (fwrite(w.out.data(), 1, w.out.size(), f_) != w.out.size()) failed_ = true;
        crc_ = (uint32_t)crc32_combine(crc_, w.crc, (z_off_t)w.text.size());
        total_ += w.text.size();
        w.text = std::string();
        w.out = std::string();
        ++written_;
      }
      lk.unlock();
      cv_.notify_all();
    }
  }
  FILE *f_ = nullptr;
  std::string cur_;
  std::vector<std::unique_ptr<Job>> jobs_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_;
  size_t next_ = 0, written_ = 0;
  uint64_t total_ = 0;
  uint32_t crc_ = 0;
  bool stop_ = false, failed_ = false;
};

struct Group {  // one UMI x cell barcode: records 2k / 2k + 1 are a pair
  std::vector<std::string> seqs;
  std::vector<std::vector<std::string>> meta;
};

}  // namespace

void process(const std::vector<std::string> &input_files,
             std::vector<std::unique_ptr<align::PseudoAligner>> &reference_indices,
             const std::vector<reference_library::Reference> &references,
             const std::vector<align::AlignFilterConfig> &aligner_configs, const std::vector<std::string> &output_paths,
             size_t num_cores, bool force_bam_paired) {
  (void)num_cores;  // the reference's consumer pool: here the UMI groups of a batch share one device call
  const size_t n_lib = reference_indices.size();
  std::vector<std::unique_ptr<GzWriter>> out(n_lib);
  puts("Spawning logging thread.");
  const unsigned gz_threads = std::max(1u, std::min(parse::usable_cpus() / 2, 8u));
  for (size_t i = 0; i < n_lib; ++i)  // created / truncated; level 6 = flate2 Compression::default()
    out[i].reset(new GzWriter(output_paths.at(i), gz_threads));
  std::vector<bool> first_write(n_lib, true);
  auto write_line = [&](size_t lib, const std::string &line) {
    if (first_write[lib]) {
      printf("Writing header for file %zu\n", lib);
      const std::string h = "nimble_features\tnimble_score\t" + bam_header("r1") + "\t" + bam_header("r2") +
                            "\tr1_filter_forward\tr1_forward_score\tr1_filter_reverse\tr1_reverse_score\tr2_filter_forward"
                            "\tr2_forward_score\tr2_filter_reverse\tr2_reverse_score\ttriage_reason\taligndirection\n";
      out[lib]->write(h.data(), h.size());
      first_write[lib] = false;
    }
    out[lib]->write(line.data(), line.size());
  };

  // UMI groups are gathered into batches: one device call per batch and library, the group index is the segment
  size_t batch_pairs = 1u << 17;  // (small enough that the reader thread and the consumer overlap on ordinary files)
  if (const char *e = getenv("NIMBLE_BAM_BATCH")) batch_pairs = std::max<size_t>((size_t)strtoull(e, nullptr, 10), 1);
  std::vector<Group> groups;
  size_t pairs = 0;

  double t_prep = 0, t_call = 0, t_rows = 0;
  auto flush = [&]() {
    if (groups.empty()) return;
    const auto tp0 = std::chrono::steady_clock::now();
    // the call's inputs: R1 = record 2k, R2 = record 2k + 1, each reverse-complemented when the BAM says the read was
    // (process/bam.rs:245-303); the quality strings are already in read direction (parse/bam.rs:270-287)
    std::vector<uint8_t> b[2], q[2], skip[2];
    std::vector<uint64_t> off[2] = {{0}, {0}};
    std::vector<uint32_t> seg;
    std::vector<size_t> first_pair(groups.size() + 1, 0);
    uint32_t max_len = 1;
    for (size_t g = 0; g < groups.size(); ++g) {
      const Group &G = groups[g];
      first_pair[g] = seg.size();
      for (size_t k = 0; k + 1 < G.seqs.size(); k += 2) {
        for (int m = 0; m < 2; ++m) {
          const std::vector<std::string> &md = G.meta[k + m];
          const std::string s = reverse_comp_if_needed(G.seqs[k + m], parse_str_as_bool(md[2]));
          const std::string &ql = md[1];
          if (ql.size() != s.size())
            throw Panic("BAM record without usable qualities (" + md[0] + "): not supported by the MI355X build");
          b[m].insert(b[m].end(), s.begin(), s.end());
          q[m].insert(q[m].end(), ql.begin(), ql.end());
          off[m].push_back(b[m].size());
          skip[m].push_back(md[37] == "TRUE" ? 1 : 0);
          max_len = std::max<uint32_t>(max_len, (uint32_t)s.size());
        }
        seg.push_back((uint32_t)g);
      }
    }
    first_pair[groups.size()] = seg.size();
    const uint64_t n = seg.size();
    for (size_t lib = 0; lib < n_lib && n; ++lib) {
      align::ReadBatch a, m;
      a.bases = b[0].data();
      a.offsets = off[0].data();
      a.n = n;
      a.max_len = max_len;
      m.bases = b[1].data();
      m.offsets = off[1].data();
      m.n = n;
      m.max_len = max_len;
      align::UmiExtras ex;
      ex.segment = seg.data();
      ex.n_segments = (uint32_t)groups.size();
      ex.qual[0] = q[0].data();
      ex.qual[1] = q[1].data();
      ex.skip[0] = skip[0].data();
      ex.skip[1] = skip[1].data();
      const auto tc0 = std::chrono::steady_clock::now();
      if (lib == 0) t_prep += std::chrono::duration<double>(tc0 - tp0).count();
      align::UmiOutput res = align::get_calls_umis(a, &m, ex, *reference_indices[lib], references.at(lib),
                                                   aligner_configs.at(lib), true);
      const auto tr0 = std::chrono::steady_clock::now();
      t_call += std::chrono::duration<double>(tr0 - tc0).count();
      struct RowsLap {
        double &acc;
        std::chrono::steady_clock::time_point t0;
        ~RowsLap() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
      } rows_lap{t_rows, tr0};
      size_t row = 0;
      for (size_t g = 0; g < groups.size(); ++g) {
        const Group &G = groups[g];
        const size_t p0 = first_pair[g], np = first_pair[g + 1] - p0;
        const size_t row0 = row;
        while (row < res.rows.size() && res.rows[row].segment == (uint32_t)g) ++row;
        if (row == row0) continue;  // `if s.len() == 0 { results.push(vec![]) }`: nothing at all for this UMI
        // filter_reasons is keyed by the read key (R1 string + R2 string): a later pair with the same key replaces an
        // earlier one (align.rs:591-600)
        std::unordered_map<std::string, size_t> last_of_key;
        auto key_of = [&](size_t pair) {
          const uint64_t i = p0 + pair;
          return std::string(b[0].begin() + (long)off[0][i], b[0].begin() + (long)off[0][i + 1]) +
                 std::string(b[1].begin() + (long)off[1][i], b[1].begin() + (long)off[1][i + 1]);
        };
        for (size_t k = 0; k < np; ++k) last_of_key[key_of(k)] = k;
        std::unordered_set<std::string> scored_qnames;
        auto emit = [&](const std::string &features, int32_t count, size_t pair) {
          const std::vector<std::string> &m1 = G.meta[2 * pair], &m2 = G.meta[2 * pair + 1];
          const align::FilterRecord &fr = res.per_read[p0 + last_of_key.at(key_of(pair))];
          std::string line = features + "\t" + std::to_string(count) + "\t" + bam_data(m2) + "\t" + bam_data(m1) + "\t";
          line += std::string(align::to_string(fr.r2)) + "\t" + std::to_string(fr.score2) + "\t";  // "r1": the mate
          line += std::string(align::to_string(align::FilterReason::None)) + "\t0\t";
          line += std::string(align::to_string(fr.r1)) + "\t" + std::to_string(fr.score1) + "\t";  // "r2": the first read
          line += std::string(align::to_string(align::FilterReason::None)) + "\t0\t";
          line += std::string(align::to_string(fr.triage)) + "\tNone\n";
          write_line(lib, line);
        };
        for (size_t r = row0; r < row; ++r) {
          const align::UmiRow &R = res.rows[r];
          const size_t pair = R.representative - p0;
          scored_qnames.insert(G.meta[2 * pair][0]);  // score.1.1[0]: the first read's QNAME
          std::string f;
          for (size_t t = 0; t < R.features.size(); ++t) f += (t ? "," : "") + R.features[t];
          emit(f, R.count, pair);
        }
        for (size_t k = 0; k < np; ++k) {  // pairs that stand for no callset: an empty call (process/bam.rs:341-355)
          if (scored_qnames.count(G.meta[2 * k + 1][0])) continue;
          emit("", 0, k);
        }
      }
    }
    groups.clear();
    pairs = 0;
  };

  // The reader runs on a thread of its own, one batch ahead (the reference has a reader thread and a pool of consumers,
  // process/bam.rs:127-226): BGZF, record decoding and the 38 strings per record are the pipeline's largest cost and
  // overlap the device call and the writing of the batch before.
  puts("Spawning reader thread.");
  struct Batch {
    std::vector<Group> groups;
    std::string error;
    bool last = false;
  };
  std::mutex qmu;
  std::condition_variable qcv;
  std::deque<std::unique_ptr<Batch>> queue;
  bool quit = false;
  double t_read = 0;
  std::thread reader_thread([&] {
    std::unique_ptr<Batch> cur(new Batch());
    size_t cur_pairs = 0;
    auto push = [&](bool last) {
      cur->last = last;
      std::unique_lock<std::mutex> lk(qmu);
      qcv.wait(lk, [&] { return quit || queue.size() < 2; });
      if (quit) return false;
      queue.push_back(std::move(cur));
      lk.unlock();
      qcv.notify_all();
      cur.reset(new Batch());
      cur_pairs = 0;
      return true;
    };
    try {
      const auto t0 = std::chrono::steady_clock::now();
      parse::bam::UMIReader reader(input_files.at(0), false, force_bam_paired);
      bool has_aligned = false;
      for (;;) {
        const bool final_umi = reader.next();
        if (final_umi && has_aligned) {
          puts("Finished reading UMIs from input file.");
          break;
        }
        Group g;
        g.seqs = std::move(reader.current_umi_group);
        g.meta = std::move(reader.current_metadata_group);
        cur_pairs += g.seqs.size() / 2;
        cur->groups.push_back(std::move(g));
        if (cur_pairs >= batch_pairs && !push(false)) return;
        has_aligned = true;
      }
      t_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } catch (const std::exception &e) {
      cur->error = e.what();
    }
    (void)push(true);
  });
  double t_wait = 0, t_flush = 0;
  std::string failure;
  try {
    for (;;) {
      std::unique_ptr<Batch> bt;
      const auto tw = std::chrono::steady_clock::now();
      {
        std::unique_lock<std::mutex> lk(qmu);
        qcv.wait(lk, [&] { return !queue.empty(); });
        bt = std::move(queue.front());
        queue.pop_front();
      }
      qcv.notify_all();
      const auto tf = std::chrono::steady_clock::now();
      t_wait += std::chrono::duration<double>(tf - tw).count();
      groups = std::move(bt->groups);
      flush();
      t_flush += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf).count();
      if (!bt->error.empty()) failure = bt->error;
      if (bt->last) break;
    }
  } catch (...) {
    {
      std::lock_guard<std::mutex> lk(qmu);
      quit = true;
    }
    qcv.notify_all();
    reader_thread.join();
    throw;
  }
  reader_thread.join();
  if (!failure.empty()) throw Panic(failure);
  if (getenv("NIMBLE_HOST_TIMING"))
    fprintf(stderr, "[nimble host] bam pipeline: reader thread %.2f s (BGZF, records, UMI groups); consumer %.2f s waiting for it, "
            "%.2f s in calls and rows (%.2f preparing the calls' inputs, %.2f in the calls, %.2f writing rows)\n", t_read, t_wait,
            t_flush, t_prep, t_call, t_rows);
  for (size_t i = 0; i < n_lib; ++i) {
    if (out[i]->close()) printf("Successfully flushed and closed file %zu\n", i);
    else fprintf(stderr, "Error finishing GZIP for file %zu\n", i);
  }
  for (const std::string &p : output_paths) {  // validate_gzip (process/bam.rs:425-435)
    printf("Validating GZIP file: %s\n", p.c_str());
    gzFile f = gzopen(p.c_str(), "rb");
    char tmp[1 << 16];
    int r = 0;
    while (f && (r = gzread(f, tmp, sizeof tmp)) > 0) {
    }
    if (!f || r < 0) fprintf(stderr, "GZIP validation failed for %s\n", p.c_str());
    else printf("Validation successful for %s\n", p.c_str());
    if (f) gzclose(f);
  }
  puts("Logging thread terminating.");
  puts("Joined on logging; terminating.");
}

}  // namespace bam
}  // namespace process
}  // namespace nimble
