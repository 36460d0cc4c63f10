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
#include <atomic>
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

// what a record's Raw says about its body (lengths, aux offset, the CB / UB / UR string tags); false = the lengths do not fit
// the body ("truncated record").  Defined below the reader.
static bool describe_record(const uint8_t *body, uint32_t block, Raw &r);

// ---- BGZF + BAM records -----------------------------------------------------------------------------------------
// A BGZF file is a series of gzip members of at most 64 KiB, each with its compressed size in an extra field ("BC"): the
// members are independent, so a producer thread cuts the file into members and a handful of helpers inflate a batch of them
// at once, each into its place of one output buffer (the uncompressed size stands in a member's last four bytes).  The
// record decoder consumes the batches in order.  One zlib stream inflated 0.3 GB/s: 2.2 of the 5.3 s the reader thread spent
// on a 2 M-record file.  A gzip file without the extra field (not BGZF) goes through zlib's gz layer as before.
struct Reader::Impl {
  // --- not BGZF: zlib's gz layer reads across members
  gzFile f = nullptr;
  // --- BGZF
  FILE *fp = nullptr;
  struct Pre {
    uint32_t pos;  // where the record's block_size field stands inside the batch
    Raw raw;       // described already (by the inflate helpers, in parallel): off = 0
  };
  struct Chunk {
    std::vector<uint8_t> data;
    std::vector<Pre> recs;  // the records that lie inside this batch whole, in order (none: the decoder goes byte by byte)
    std::string error;   // raised when the consumer has used up `data`
    bool truncated = false, last = false;
  };
  size_t cur_rec = 0;
  std::thread producer;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::shared_ptr<Chunk>> queue;
  bool quit = false;
  std::shared_ptr<Chunk> cur;
  Hold hold;  // what the record returned last lies in: `cur`, or a buffer of its own for a record that straddled two batches
  size_t cur_at = 0;
  bool ended = false;
  double waited = 0;  // the record decoder waiting for inflated data
  std::vector<uint8_t> buf;

  ~Impl() {
    if (producer.joinable()) {
      {
        std::lock_guard<std::mutex> lk(mu);
        quit = true;
      }
      cv.notify_all();
      producer.join();
    }
    if (fp) fclose(fp);
    if (f) gzclose(f);
    if (getenv("NIMBLE_HOST_TIMING") && waited > 0)
      fprintf(stderr, "[nimble host] bam reader: %.2f s waiting for inflated BGZF blocks\n", waited);
  }

  struct Member {
    size_t at, size;     // inside the batch's compressed bytes
    size_t data_at;      // where the deflate stream starts inside the member
    uint32_t isize, crc;
    size_t out_at;
  };
  // size of the member that starts at p (n bytes available); 0 = not decidable yet (too few bytes), NOT_A_MEMBER = these
  // bytes are no BGZF member whatever follows
  static constexpr size_t NOT_A_MEMBER = ~(size_t)0;
  static size_t member_size(const uint8_t *p, size_t n, size_t &data_at) {
    if (n < 12) return 0;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return NOT_A_MEMBER;
    const size_t xlen = p[10] | ((size_t)p[11] << 8);
    if (n < 12 + xlen) return 0;
    size_t o = 12;
    const size_t xe = 12 + xlen;
    while (o + 4 <= xe) {
      const size_t slen = p[o + 2] | ((size_t)p[o + 3] << 8);
      if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= xe) {
        data_at = xe;
        return (size_t)(p[o + 4] | ((size_t)p[o + 5] << 8)) + 1;
      }
      o += 4 + slen;
    }
    return NOT_A_MEMBER;  // a gzip member without the BGZF size field
  }
  bool producer_dead = false;  // the producer ended with an exception: nothing more will come (under mu)
  void producer_failed(const char *what) {
    std::shared_ptr<Chunk> ch;
    try {
      ch = std::make_shared<Chunk>();
      ch->error = std::string("Error -- could not read BAM file (") + what + ")";
      ch->last = true;
    } catch (...) {
      ch.reset();  // (not even that much memory: the consumer is told through producer_dead alone)
    }
    {
      std::unique_lock<std::mutex> lk(mu);
      producer_dead = true;
      try {
        if (ch && !quit) queue.push_back(std::move(ch));
      } catch (...) {
      }
    }
    cv.notify_all();
  }
  void produce(unsigned helpers) {
    std::vector<uint8_t> comp;
    size_t have = 0;
    bool file_end = false;
    size_t BATCH_IN = 4u << 20;  // compressed bytes per batch (NIMBLE_BGZF_BATCH: tests make batches of a few members)
    if (const char *e = getenv("NIMBLE_BGZF_BATCH")) BATCH_IN = std::max<size_t>((size_t)strtoull(e, nullptr, 10), 1024);
    // Where records start in the inflated stream is known to this thread alone (it sees every byte in order): the records that
    // lie inside a batch whole are described here, by the helpers, so that the decoder thread -- the pipeline's narrowest
    // place at 250 ns a record -- only copies them.  `tail` = the bytes of the record that straddles into the next batch;
    // prescan < 0 = given up (a header that does not fit the first batch: everything goes byte by byte).
    int prescan = 0;  // 0 = header not seen yet, 1 = on, -1 = off
    std::vector<uint8_t> tail;
    for (;;) {
      std::shared_ptr<Chunk> ch(new Chunk());
      // fill the compressed window
      if (comp.size() < have + BATCH_IN) comp.resize(have + BATCH_IN);
      while (!file_end && have < BATCH_IN) {
        const size_t r = fread(comp.data() + have, 1, comp.size() - have, fp);
        if (r == 0) file_end = true;
        have += r;
      }
      // cut it into whole members
      std::vector<Member> ms;
      size_t at = 0, out_total = 0;
      while (at < have) {
        size_t data_at = 0;
        const size_t sz = member_size(comp.data() + at, have - at, data_at);
        if (sz == NOT_A_MEMBER || (sz != 0 && sz < data_at + 8)) {  // (what stands in front of it is delivered first)
          ch->error = "Error -- could not read BAM file (corrupt BGZF block)";
          at = have;
          break;
        }
        if (sz == 0 || at + sz > have) {
          if (file_end) {  // the file ends inside a member: what zlib reported as an unexpected end
            ch->truncated = true;
            at = have;
          }
          break;
        }
        Member m;
        m.at = at;
        m.size = sz;
        m.data_at = data_at;
        memcpy(&m.crc, comp.data() + at + sz - 8, 4);
        memcpy(&m.isize, comp.data() + at + sz - 4, 4);
        if (m.isize > (1u << 16)) {  // (a BGZF member holds at most 64 KiB)
          ch->error = "Error -- could not read BAM file (corrupt BGZF block)";
          at = have;
          break;
        }
        m.out_at = out_total;
        out_total += m.isize;
        ms.push_back(m);
        at += sz;
      }
      if (ms.empty() && !file_end && ch->error.empty()) {  // not one whole member in the window: a wider one
        BATCH_IN *= 2;
        continue;
      }
      ch->data.resize(out_total);
      std::atomic<size_t> next{0};
      std::atomic<size_t> first_bad{~(size_t)0};  // the first member that does not inflate to its size and CRC
      auto mark_bad = [&](size_t i) {
        size_t cur_min = first_bad.load();
        while (i < cur_min && !first_bad.compare_exchange_weak(cur_min, i)) {
        }
      };
      auto work = [&] {
        z_stream z;
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= ms.size()) return;
          const Member &m = ms[i];
          memset(&z, 0, sizeof z);
          if (inflateInit2(&z, -15) != Z_OK) {
            mark_bad(i);
            return;
          }
          z.next_in = comp.data() + m.at + m.data_at;
          z.avail_in = (uInt)(m.size - m.data_at - 8);
          // (a member that holds no data -- the 28-byte BGZF end-of-file marker -- may be the only member of a batch: the
          // batch's buffer is then empty and its data() null, which zlib refuses as an output pointer)
          Bytef nothing = 0;
          Bytef *const dst = m.isize ? ch->data.data() + m.out_at : &nothing;
          z.next_out = dst;
          z.avail_out = m.isize;
          const int rc = inflate(&z, Z_FINISH);
          const bool ok = rc == Z_STREAM_END && z.avail_out == 0 && (uint32_t)crc32(0, dst, m.isize) == m.crc;
          inflateEnd(&z);
          if (!ok) mark_bad(i);
        }
      };
      if (ms.size() > 4 && helpers > 1) {
        threads::run_beside(helpers - 1, work);
      } else {
        work();
      }
      const bool bad = first_bad.load() != ~(size_t)0;
      if (bad) {
        // what lies in front of the damaged member is good data: the decoder gets it and meets the error behind it (a reader
        // that stops earlier -- the reference's end-of-input quirks -- never sees it, as with one zlib stream)
        ch->data.resize(ms[first_bad.load()].out_at);
        ch->error = "Error -- could not read BAM file (corrupt BGZF block)";
        ch->truncated = false;
      }
      if (prescan >= 0) {
        const uint8_t *d = ch->data.data();
        const size_t n = ch->data.size();
        size_t pos = 0;
        bool ok = true;
        if (prescan == 0) {  // the BAM header: magic, text, reference names
          auto i32at = [&](size_t o, int32_t &v) {
            if (o + 4 > n) return false;
            memcpy(&v, d + o, 4);
            return true;
          };
          int32_t l_text = 0, n_ref = 0;
          ok = n >= 4 && memcmp(d, "BAM\1", 4) == 0 && i32at(4, l_text) && l_text >= 0 && i32at(8 + (size_t)l_text, n_ref) && n_ref >= 0;
          pos = 12 + (size_t)(ok ? l_text : 0);
          for (int32_t i = 0; ok && i < n_ref; ++i) {
            int32_t l_name = 0;
            ok = i32at(pos, l_name) && l_name >= 0;
            pos += 8 + (size_t)(ok ? l_name : 0);
          }
          ok = ok && pos <= n;
          prescan = ok ? 1 : -1;
        } else {
          // finish the straddling record: its size field may itself be split
          if (!tail.empty()) {
            while (tail.size() < 4 && pos < n) tail.push_back(d[pos++]);
            if (tail.size() >= 4) {
              int32_t block;
              memcpy(&block, tail.data(), 4);
              const size_t need = block < 0 ? 0 : 4 + (size_t)block;
              if (block < 32) {
                prescan = -1;  // (the decoder will report it)
              } else if (need - tail.size() <= n - pos) {
                pos += need - tail.size();
                tail.clear();
              } else {
                tail.insert(tail.end(), d + pos, d + n);  // (a record longer than a batch: keep collecting)
                pos = n;
              }
            }
          }
        }
        if (prescan == 1 && tail.empty()) {
          while (pos + 4 <= n) {
            int32_t block;
            memcpy(&block, d + pos, 4);
            if (block < 32) {
              prescan = -1;
              break;
            }
            if (pos + 4 + (size_t)block > n) break;
            Pre p;
            p.pos = (uint32_t)pos;
            ch->recs.push_back(p);
            pos += 4 + (size_t)block;
          }
          if (prescan == 1) tail.assign(d + pos, d + n);
          std::atomic<size_t> nx{0};
          std::atomic<bool> stop_at_bad{false};
          auto describe = [&] {
            for (;;) {
              const size_t i0 = nx.fetch_add(256);
              if (i0 >= ch->recs.size()) return;
              for (size_t i = i0; i < std::min(ch->recs.size(), i0 + 256); ++i) {
                Pre &p = ch->recs[i];
                int32_t block;
                memcpy(&block, d + p.pos, 4);
                if (!describe_record(d + p.pos + 4, (uint32_t)block, p.raw)) {
                  p.raw.len = Raw::NONE;  // (lengths that do not fit: the decoder reports it when it gets there)
                  stop_at_bad = true;
                }
              }
            }
          };
          if (ch->recs.size() > 1024 && helpers > 1) {
            threads::run_beside(helpers - 1, describe);
          } else {
            describe();
          }
          if (stop_at_bad) prescan = -1;
        }
      }
      memmove(comp.data(), comp.data() + at, have - at);
      have -= at;
      ch->last = !ch->error.empty() || ch->truncated || (file_end && have == 0);
      const bool last = ch->last;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit || queue.size() < 3; });
        if (quit) return;
        queue.push_back(std::move(ch));
      }
      cv.notify_all();
      if (last) return;
    }
  }
  // the next n bytes where they lie, consumed -- when the current batch holds all of them
  const uint8_t *contiguous(size_t n) {
    if (f || !cur || cur->data.size() - cur_at < n) return nullptr;
    const uint8_t *p = cur->data.data() + cur_at;
    cur_at += n;
    return p;
  }
  bool read_exact(void *dst, size_t n, bool &eof) {
    uint8_t *p = static_cast<uint8_t *>(dst);
    size_t got = 0;
    if (f) {
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
    while (got < n) {
      if (!more_data()) break;
      const size_t k = std::min(n - got, cur->data.size() - cur_at);
      memcpy(p + got, cur->data.data() + cur_at, k);
      cur_at += k;
      got += k;
    }
    eof = got == 0;
    return got == n;
  }
  // true = the current batch holds unread bytes (the next batch is fetched when the current one is used up); false = the
  // input has ended; an error that stands behind the last byte is raised here
  bool more_data() {
    for (;;) {
      if (cur && cur_at < cur->data.size()) return true;
      if (cur) {  // used up: what stood behind it?
        if (!cur->error.empty()) throw Panic(cur->error);
        if (cur->truncated) throw Panic("0: Found truncated record");  // the file ends inside a block
        if (cur->last) ended = true;
      }
      if (ended) return false;
      const auto tw = std::chrono::steady_clock::now();
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !queue.empty() || producer_dead; });
      waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
      if (queue.empty()) throw Panic("Error -- could not read BAM file (the reader thread failed)");
      cur = std::move(queue.front());
      queue.pop_front();
      cur_at = 0;
      cur_rec = 0;
      lk.unlock();
      cv.notify_all();
    }
  }
  // the record at the cursor as the producer described it (nullptr: not described -- it straddles two batches, or the
  // description was given up); consumed when returned
  const Pre *described(const uint8_t *&body) {
    if (f || !cur) return nullptr;
    std::vector<Pre> &rs = cur->recs;
    while (cur_rec < rs.size() && rs[cur_rec].pos < cur_at) ++cur_rec;
    if (cur_rec >= rs.size() || rs[cur_rec].pos != cur_at || rs[cur_rec].raw.len == Raw::NONE) return nullptr;
    const Pre *p = &rs[cur_rec++];
    body = cur->data.data() + p->pos + 4;
    cur_at = (size_t)p->pos + 4 + p->raw.len;
    return p;
  }
};

Reader::Reader(const std::string &path) : impl_(new Impl()) {
  impl_->fp = fopen(path.c_str(), "rb");
  if (!impl_->fp) throw Panic("Error -- could not open BAM file " + path);
  uint8_t head[64];
  const size_t hn = fread(head, 1, sizeof head, impl_->fp);
  size_t data_at = 0;
  const size_t first = Impl::member_size(head, hn, data_at);
  const bool bgzf = first != 0 && first != Impl::NOT_A_MEMBER;
  if (bgzf) {
    rewind(impl_->fp);
    unsigned helpers = std::max(1u, std::min(parse::usable_cpus() / 2, 8u));
    if (const char *e = getenv("NIMBLE_BGZF_THREADS")) helpers = (unsigned)std::max(1, atoi(e));
    Impl *im = impl_.get();
    // (nothing thrown inside the producer may reach the top of its thread -- that is std::terminate: it is handed to the
    // consumer as the input's error, like a damaged block)
    impl_->producer = std::thread([im, helpers] {
      try {
        im->produce(helpers);
      } catch (const std::exception &e) {
        im->producer_failed(e.what());
      } catch (...) {
        im->producer_failed("unknown failure");
      }
    });
  } else {
    fclose(impl_->fp);
    impl_->fp = nullptr;
    impl_->f = gzopen(path.c_str(), "rb");
    if (!impl_->f) throw Panic("Error -- could not open BAM file " + path);
    gzbuffer(impl_->f, 1u << 20);
  }
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

Reader::~Reader() {}

namespace {
inline int32_t ld_i32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
inline uint32_t ld_u16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
// fixed part of a record body
inline uint32_t b_flag(const uint8_t *b) { return ld_u16(b + 14); }
inline uint32_t b_lname(const uint8_t *b) { return b[8]; }
inline uint32_t b_lseq(const uint8_t *b) { return (uint32_t)ld_i32(b + 16); }
inline size_t b_seq_at(const uint8_t *b) { return 32 + (size_t)b_lname(b) + 4ull * ld_u16(b + 12); }
const char SEQ_CODE[] = "=ACMGRSVTWYHKDBN";
}  // namespace

const Hold &Reader::hold() const { return impl_->hold; }

bool Reader::next_raw(Raw &r) {
  bool eof = false;
  int32_t block = 0;
  if (!impl_->f && impl_->more_data()) {
    const uint8_t *body = nullptr;
    if (const Impl::Pre *p = impl_->described(body)) {
      r = p->raw;
      r.body = body;
      if (impl_->hold.get() != impl_->cur.get()) impl_->hold = impl_->cur;
      return true;
    }
  }
  if (!impl_->read_exact(&block, 4, eof)) {
    if (eof) return false;
    throw Panic("0: Found truncated record");  // parse/bam.rs:136-139
  }
  if (block < 32) throw Panic("0: Found truncated record");
  if (const uint8_t *src = impl_->contiguous((size_t)block)) {  // inside one inflated batch, only not described
    if (!describe_record(src, (uint32_t)block, r)) throw Panic("0: Found truncated record");
    r.body = src;
    if (impl_->hold.get() != impl_->cur.get()) impl_->hold = impl_->cur;
    return true;
  }
  // a record that straddles two batches (or a file that is no BGZF): its bytes are put together in a buffer of their own
  std::shared_ptr<std::vector<uint8_t>> own(new std::vector<uint8_t>((size_t)block));
  if (!impl_->read_exact(own->data(), (size_t)block, eof)) throw Panic("0: Found truncated record");
  if (!describe_record(own->data(), (uint32_t)block, r)) throw Panic("0: Found truncated record");
  r.body = own->data();
  impl_->hold = own;
  return true;
}

bool Reader::next(Record &r) {
  Raw raw;
  if (!next_raw(raw)) return false;
  const uint8_t *p = raw.body;
  r.tid = ld_i32(p);
  r.pos = ld_i32(p + 4);
  const uint32_t l_read_name = b_lname(p);
  r.mapq = p[9];
  r.flag = b_flag(p);
  const uint32_t l_seq = b_lseq(p);
  r.mtid = ld_i32(p + 20);
  r.mpos = ld_i32(p + 24);
  r.tlen = ld_i32(p + 28);
  r.qname.assign(reinterpret_cast<const char *>(p + 32), l_read_name ? l_read_name - 1 : 0);  // NUL-terminated
  size_t o = b_seq_at(p);
  r.seq.resize(l_seq);
  for (uint32_t i = 0; i < l_seq; ++i) r.seq[i] = SEQ_CODE[(p[o + i / 2] >> (i & 1 ? 0 : 4)) & 15];
  o += (l_seq + 1) / 2;
  r.qual.assign(reinterpret_cast<const char *>(p + o), l_seq);
  r.aux.assign(p + raw.aux, p + raw.len);
  return true;
}

// One walk over aux data for a set of two-character tags, each with the answer rust-htslib's Record::aux gives for an
// Aux::String: the first entry of a tag decides (a string gives its value, any other type gives nothing), and nothing behind
// a malformed entry is seen.  at[k] / len[k] = value of tag k inside `aux` (len NONE = no string).
static void aux_walk(const uint8_t *aux, size_t n, const char *const *tags, int n_tags, uint32_t *at, uint32_t *len) {
  bool seen[16];
  for (int k = 0; k < n_tags; ++k) {
    seen[k] = false;
    len[k] = Raw::NONE;
    at[k] = 0;
  }
  size_t o = 0;
  while (o + 3 <= n) {
    const char t0 = (char)aux[o], t1 = (char)aux[o + 1], ty = (char)aux[o + 2];
    o += 3;
    int k = -1;
    for (int q = 0; q < n_tags; ++q)
      if (tags[q][0] == t0 && tags[q][1] == t1) {
        k = q;
        break;
      }
    size_t l = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': l = 1; break;
      case 's': case 'S': l = 2; break;
      case 'i': case 'I': case 'f': l = 4; break;
      case 'Z': case 'H': {
        size_t e = o;
        while (e < n && aux[e] != 0) ++e;
        if (e >= n) return;
        if (k >= 0 && !seen[k]) {
          seen[k] = true;
          if (ty == 'Z') {
            at[k] = (uint32_t)o;
            len[k] = (uint32_t)(e - o);
          }
        }
        o = e + 1;
        continue;
      }
      case 'B': {
        if (o + 5 > n) return;
        const char sub = (char)aux[o];
        uint32_t cnt;
        memcpy(&cnt, aux + o + 1, 4);
        const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        l = 5 + w * (size_t)cnt;
        break;
      }
      default: return;
    }
    if (k >= 0) seen[k] = true;  // present, but not a string
    o += l;
  }
}

static bool describe_record(const uint8_t *p, uint32_t block, Raw &r) {
  r = Raw();
  r.len = block;
  const uint32_t l_seq = b_lseq(p);
  const size_t o = b_seq_at(p);
  if (o + (l_seq + 1) / 2 + (size_t)l_seq > (size_t)block) return false;
  r.aux = (uint32_t)(o + (l_seq + 1) / 2 + l_seq);
  r.l_seq = l_seq;
  r.flag = (uint16_t)b_flag(p);
  // the cell barcode and the UMI (corrected UB, else raw UR: sorted_bam_reader.rs:57-65) as SortedBamReader asks for them
  static const char *const T3[3] = {"CB", "UB", "UR"};
  uint32_t at[3], len[3];
  aux_walk(p + r.aux, block - r.aux, T3, 3, at, len);
  r.cb_len = len[0];
  r.cb = r.aux + at[0];
  const int u = len[1] != Raw::NONE ? 1 : (len[2] != Raw::NONE ? 2 : -1);
  r.umi_len = u < 0 ? Raw::NONE : len[u];
  r.umi = u < 0 ? 0 : r.aux + at[u];
  return true;
}

// the value of a 'Z' tag (what rust-htslib hands out as Aux::String); any other type: not a string
bool Record::aux_string(const char *tag, std::string &out) const {
  if (strlen(tag) != 2) return false;  // rust-htslib: a tag has two characters, anything else is an error
  uint32_t at, len;
  const char *tags[1] = {tag};
  aux_walk(aux.data(), aux.size(), tags, 1, &at, &len);
  if (len == Raw::NONE) return false;
  out.assign(reinterpret_cast<const char *>(aux.data() + at), len);
  return true;
}

static const char *const REPORT_TAGS[15] = {"NH", "HI", "AS", "GN", "TX", "AN", "nM", "fx", "RE", "CR", "CY", "CB", "UR", "UY", "UB"};

// rust-htslib Record::read_pair_orientation
static const char *pair_orientation(const uint8_t *b) {
  const uint32_t flag = b_flag(b);
  const int32_t tid = ld_i32(b), pos = ld_i32(b + 4), mtid = ld_i32(b + 20), mpos = ld_i32(b + 24);
  const bool paired = flag & 0x1, unmapped = flag & 0x4, mate_unmapped = flag & 0x8;
  if (!(paired && !unmapped && !mate_unmapped && tid == mtid)) return "None";
  if (pos == mpos) return "None";
  const bool rev = flag & 0x10, mrev = flag & 0x20, first = flag & 0x40;
  int64_t p1, p2;
  bool f1, f2;
  if (first) { p1 = pos; p2 = mpos; f1 = !rev; f2 = !mrev; }
  else { p1 = mpos; p2 = pos; f1 = !mrev; f2 = !rev; }
  if (p1 < p2) return f1 ? (f2 ? "F1F2" : "F1R2") : (f2 ? "R1F2" : "R1R2");
  return f2 ? (f1 ? "F2F1" : "F2R1") : (f1 ? "R2F1" : "R2R1");
}

// ---- a record's reported values, straight from its bytes --------------------------------------------------------
static const size_t CLIP_LENGTH = 13;

// strip_nonbio_regions (parse/bam.rs:256-288) + DnaString::from_acgt_bytes / to_string: a 124-base read loses its 13
// non-biological bases (at the end when the read is on the reverse strand), and whatever is no A/C/G/T reads as 'A'
void raw_sequence(const uint8_t *b, const Raw &r, std::string &out) {
  const uint32_t l_seq = b_lseq(b);
  const bool rev = b_flag(b) & 0x10;
  uint32_t lo = 0, hi = l_seq;
  if (l_seq == 124) {
    if (rev) hi = l_seq - CLIP_LENGTH;
    else lo = CLIP_LENGTH;
  }
  const uint8_t *sq = b + b_seq_at(b);
  out.resize(hi - lo);
  for (uint32_t i = lo; i < hi; ++i) {
    const char c = SEQ_CODE[(sq[i / 2] >> (i & 1 ? 0 : 4)) & 15];
    out[i - lo] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'A';
  }
}

void raw_quality(const uint8_t *b, const Raw &r, std::string &out) {
  out.clear();
  if (r.qual_bad) return;  // (warned about when the record came in; an empty string stays empty through clip and reverse)
  const uint32_t l_seq = b_lseq(b);
  const bool rev = b_flag(b) & 0x10;
  uint32_t lo = 0, hi = l_seq;
  if (l_seq == 124) {
    if (rev) hi = l_seq - CLIP_LENGTH;
    else lo = CLIP_LENGTH;
  }
  const uint8_t *q = b + b_seq_at(b) + (l_seq + 1) / 2;
  out.assign(reinterpret_cast<const char *>(q + lo), hi - lo);
  if (rev) std::reverse(out.begin(), out.end());
}

// The two of them as the call takes them (process/bam.rs:245-303): the read reverse-complemented back when the BAM holds its
// reverse strand, the qualities in read direction -- written straight into the call's input arrays, two bases per packed byte
// through a table (raw_sequence + utils::revcomp + raw_quality through strings took 750 ns a read).
bool raw_quality_is_text(const uint8_t *b, const Raw &r) {
  const uint8_t *q = b + b_seq_at(b) + (r.l_seq + 1) / 2;
  uint8_t any = 0;
  for (uint32_t i = 0; i < r.l_seq; ++i) any |= q[i];
  return !(any & 0x80);
}

void raw_call_input(const uint8_t *b, const Raw &r, uint8_t *bases, uint8_t *quals) {
  static const struct Tables {
    uint8_t fwd[16], rc[16];
    Tables() {
      for (int c = 0; c < 16; ++c) {
        const char ch = SEQ_CODE[c];
        const char a = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? ch : 'A';
        fwd[c] = (uint8_t)a;
        rc[c] = (uint8_t)(a == 'A' ? 'T' : a == 'C' ? 'G' : a == 'G' ? 'C' : 'A');
      }
    }
  } T;
  const uint32_t l_seq = r.l_seq;
  const bool rev = r.flag & 0x10;
  uint32_t lo = 0, hi = l_seq;
  if (l_seq == 124) {
    if (rev) hi = l_seq - CLIP_LENGTH;
    else lo = CLIP_LENGTH;
  }
  const uint8_t *sq = b + b_seq_at(b);
  const uint8_t *q = sq + (l_seq + 1) / 2;
  const uint32_t n = hi - lo;
  if (!rev) {
    for (uint32_t i = lo; i < hi; ++i) bases[i - lo] = T.fwd[(sq[i >> 1] >> ((i & 1) ? 0 : 4)) & 15];
    if (!r.qual_bad) memcpy(quals, q + lo, n);
  } else {
    // strip, then reverse-complement: base i of the kept stretch lands at n - 1 - (i - lo)
    for (uint32_t i = lo; i < hi; ++i) bases[n - 1 - (i - lo)] = T.rc[(sq[i >> 1] >> ((i & 1) ? 0 : 4)) & 15];
    // (the qualities were reversed once by the reader, parse/bam.rs:270-287, and stay that way)
    if (!r.qual_bad)
      for (uint32_t i = lo; i < hi; ++i) quals[n - 1 - (i - lo)] = q[i];
  }
}

namespace {
inline void put_bool(std::string &o, bool v) { o += v ? "true" : "false"; }
inline void put_int(std::string &o, long long v) {  // (snprintf took a microsecond per row: ten numbers a row)
  char tmp[24];
  char *e = tmp + sizeof tmp, *p = e;
  unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
  do {
    *--p = (char)('0' + u % 10);
    u /= 10;
  } while (u);
  if (v < 0) *--p = '-';
  o.append(p, (size_t)(e - p));
}
// field k of BAM_FIELDS_TO_REPORT appended to o; tags from one aux_walk over REPORT_TAGS.  (1 = QUAL and 15 = SEQ are made
// by raw_quality / raw_sequence.)
inline void put_field(std::string &o, int k, const uint8_t *b, const Raw &r, const uint32_t *tat, const uint32_t *tlen) {
  const uint32_t flag = b_flag(b);
  switch (k) {
    case 0: {
      const uint32_t ln = b_lname(b);
      o.append(reinterpret_cast<const char *>(b + 32), ln ? ln - 1 : 0);
      break;
    }
    case 2: put_bool(o, flag & 0x10); break;
    case 3: put_bool(o, flag & 0x20); break;
    case 4: put_bool(o, flag & 0x1); break;
    case 5: put_bool(o, flag & 0x2); break;
    case 6: o += pair_orientation(b); break;
    case 7: put_bool(o, flag & 0x4); break;
    case 8: put_bool(o, flag & 0x8); break;
    case 9: put_bool(o, flag & 0x40); break;
    case 10: put_bool(o, flag & 0x80); break;
    case 11: o += (flag & 0x10) ? '-' : '+'; break;
    case 12: put_int(o, b[9]); break;
    case 13: put_int(o, ld_i32(b + 4)); break;
    case 14: put_int(o, ld_i32(b + 24)); break;
    case 16: put_int(o, b_lseq(b)); break;
    case 17: put_int(o, ld_i32(b + 28)); break;
    case 18: put_bool(o, flag & 0x200); break;
    case 19: put_bool(o, flag & 0x100); break;
    case 20: put_bool(o, flag & 0x400); break;
    case 21: put_bool(o, flag & 0x800); break;
    case 37: o += r.skip == 2 ? "TRUE" : (r.skip == 1 ? "FALSE" : ""); break;
    default:
      if (k >= 22 && k < 37 && tlen[k - 22] != Raw::NONE)
        o.append(reinterpret_cast<const char *>(b + r.aux + tat[k - 22]), tlen[k - 22]);
  }
}
}  // namespace

void append_int(std::string &o, long long v) { put_int(o, v); }

void raw_fields(const uint8_t *b, const Raw &r, std::vector<std::string> &out) {
  uint32_t tat[15], tlen[15];
  aux_walk(b + r.aux, r.len - r.aux, REPORT_TAGS, 15, tat, tlen);
  out.assign(38, std::string());
  for (int k = 0; k < 38; ++k) {
    if (k == 1) raw_quality(b, r, out[1]);
    else if (k == 15) raw_sequence(b, r, out[15]);
    else put_field(out[(size_t)k], k, b, r, tat, tlen);
  }
}

void raw_row_fields(const uint8_t *b, const Raw &r, std::string &o) {
  uint32_t tat[15], tlen[15];
  aux_walk(b + r.aux, r.len - r.aux, REPORT_TAGS, 15, tat, tlen);
  bool first = true;
  for (int k = 0; k < 38; ++k) {
    if (k == 1 || k == 15) continue;
    if (!first) o += '\t';
    put_field(o, k, b, r, tat, tlen);
    first = false;
  }
}

// ---- SortedBamReader (sorted_bam_reader.rs) ---------------------------------------------------------------------
SortedBamReader::SortedBamReader(const std::string &path, bool force_bam_paired)
    : reader_(path), force_bam_paired_(force_bam_paired) {}

void SortedBamReader::fill_buffer() {
  buffer_.clear();
  holds_.clear();
  ++generation_;
  for (const Raw &r : next_records_) buffer_.push_back(r);  // (the first record of this UMI, read when the UMI before ended)
  if (!next_records_.empty()) holds_.push_back(next_hold_);
  next_records_.clear();
  next_hold_.reset();
  current_umi_ = next_umi_;
  Raw rec;
  for (;;) {
    if (!reader_.next_raw(rec)) break;
    const uint8_t *b = rec.body;
    bool keep = !(!(rec.flag & 0x1) && force_bam_paired_);
    if (keep) keep = rec.cb_len != Raw::NONE;  // no cell barcode: not reported
    if (keep) {
      if (rec.umi_len == Raw::NONE) throw Panic("Error -- Could not read UMI.");
      keep = !(rec.umi_len == 10 && memcmp(b + rec.umi, "AAAAAAAAAA", 10) == 0);
    }
    if (!keep) continue;
    const char *um = reinterpret_cast<const char *>(b + rec.umi);
    if (current_umi_.empty()) current_umi_.assign(um, rec.umi_len);
    if (current_umi_.size() != rec.umi_len || memcmp(current_umi_.data(), um, rec.umi_len) != 0) {
      // records of one UMI ordered by cell barcode (a stable sort, like Vec::sort_by)
      std::stable_sort(buffer_.begin(), buffer_.end(), [](const Raw &x, const Raw &y) {
        const int c = memcmp(x.body + x.cb, y.body + y.cb, std::min(x.cb_len, y.cb_len));
        return c < 0 || (c == 0 && x.cb_len < y.cb_len);
      });
      next_umi_.assign(um, rec.umi_len);
      next_records_.push_back(rec);
      next_hold_ = reader_.hold();
      return;
    }
    if (holds_.empty() || holds_.back().get() != reader_.hold().get()) holds_.push_back(reader_.hold());
    buffer_.push_back(rec);
  }
}

void SortedBamReader::add_dummy_paired_reads() {
  std::vector<Raw> out;
  out.reserve(buffer_.size());
  for (Raw &r : buffer_) {
    r.skip = 1;
    out.push_back(r);
    if (!(r.flag & 0x1)) {
      Raw dummy = r;  // (the same bytes: the dummy differs in the pushed tag alone)
      dummy.skip = 2;
      out.push_back(dummy);
    }
  }
  buffer_.swap(out);
}

void SortedBamReader::filter_paired_reads() {
  std::vector<Raw> out;
  out.reserve(buffer_.size());
  auto qn = [](const Raw &r) {
    const uint8_t *b = r.body;
    const uint32_t ln = b_lname(b);
    return std::string(reinterpret_cast<const char *>(b + 32), ln ? ln - 1 : 0);
  };
  auto same_name = [](const Raw &x, const Raw &y) {
    const uint8_t *a = x.body, *b = y.body;
    const uint32_t la = b_lname(a) ? b_lname(a) - 1 : 0, lb = b_lname(b) ? b_lname(b) - 1 : 0;
    return la == lb && memcmp(a + 32, b + 32, la) == 0;
  };
  // (`seen` = the names of everything in front of record i: only an unpaired read ever asks, so it is made when one shows up)
  std::set<std::string> seen;
  size_t seen_upto = 0;
  size_t i = 0;
  while (i < buffer_.size()) {
    if (i + 1 >= buffer_.size()) break;
    if (same_name(buffer_[i], buffer_[i + 1])) {
      if (buffer_[i].flag & 0x40) {
        out.push_back(buffer_[i]);
        out.push_back(buffer_[i + 1]);
      } else {
        out.push_back(buffer_[i + 1]);
        out.push_back(buffer_[i]);
      }
      i += 2;
    } else {
      puts("Warning: Unpaired qname!");
      for (; seen_upto < i; ++seen_upto) seen.insert(qn(buffer_[seen_upto]));
      const std::string name = qn(buffer_[i]);
      if (seen.count(name)) printf("Warning: Read with qname '\"%s\"' has been deleted but was seen before.\n", name.c_str());
      i += 1;
    }
  }
  buffer_.swap(out);
}

bool SortedBamReader::next(Raw &out) {
  if (cursor_ >= buffer_.size()) {
    fill_buffer();
    if (!force_bam_paired_) add_dummy_paired_reads();
    filter_paired_reads();
    cursor_ = 0;
    if (buffer_.empty()) return false;  // (the reference reports BamTruncatedRecord here: its end-of-input signal)
  }
  out = buffer_[cursor_++];
  return true;
}

// ---- UMIReader (parse/bam.rs) -----------------------------------------------------------------------------------
UMIReader::UMIReader(const std::string &path, bool terminate_on_error, bool force_bam_paired)
    : reader_(path, force_bam_paired), terminate_on_error_(terminate_on_error) {}

std::string UmiBatch::umi_of(size_t g) const {
  if (groups[g].count == 0) return std::string();
  const Raw &r = recs[groups[g].first];
  return std::string(reinterpret_cast<const char *>(r.body + r.umi), r.umi_len);
}
std::string UmiBatch::cell_of(size_t g) const {
  if (groups[g].count == 0) return std::string();
  const Raw &r = recs[groups[g].first + groups[g].count - 1];
  return std::string(reinterpret_cast<const char *>(r.body + r.cb), r.cb_len >= 2 ? r.cb_len - 2 : 0);
}

// A group starts with the record that ended the group before it (kept aside) and grows until a record of another (UMI, cell
// barcode) shows up; false = the input ended (the group is the last one).
bool UMIReader::next_group(UmiBatch &st) {
  UmiBatch::Group g;
  g.first = (uint32_t)st.recs.size();
  // the batch keeps what its records lie in: the holds of the sorted reader's current UMI, added once per UMI and batch
  auto keep_holds = [&](const std::vector<Hold> &hs) {
    for (const Hold &h : hs) {
      bool have = false;
      for (size_t k = st.holds.size(); k-- > 0 && k + 4 > st.holds.size();) have |= st.holds[k].get() == h.get();
      if (!have) st.holds.push_back(h);
    }
  };
  if (st.holds.empty()) merged_generation_ = ~0ULL;  // (a batch just begun)
  auto take = [&](const Raw &rec) {
    st.recs.push_back(rec);
    ++g.count;
  };
  if (have_pend_) {
    keep_holds(pend_holds_);
    take(pend_);
    have_pend_ = false;
  }
  current_iteration_key_ = next_iteration_key_;
  next_iteration_key_.clear();
  Raw rec;
  std::string key;
  bool more = false;
  for (;;) {
    if (!reader_.next(rec)) break;
    ++read_counter_;
    if (read_counter_ % 1000000 == 0) printf("Aligned reads %zu-%zu\n", read_counter_ - 1000000, read_counter_);
    const uint8_t *b = rec.body;
    // (SortedBamReader hands on only records with a UMI and a cell barcode: the reference's checks here cannot fail)
    const char *um = reinterpret_cast<const char *>(b + rec.umi);
    const char *cb = reinterpret_cast<const char *>(b + rec.cb);
    const size_t cell_len = rec.cb_len >= 2 ? rec.cb_len - 2 : 0;
    // the iteration key is UMI + cell barcode as ONE string; compared in place (the string is made when a group opens)
    auto make_key = [&] {
      key.assign(um, rec.umi_len);
      key.append(cb, cell_len);
    };
    if (current_iteration_key_.empty()) {
      make_key();
      current_iteration_key_ = key;
    }
    const std::string &ck = current_iteration_key_;
    const bool same = ck.size() == rec.umi_len + cell_len && memcmp(ck.data(), um, rec.umi_len) == 0 &&
                      memcmp(ck.data() + rec.umi_len, cb, cell_len) == 0;
    // (whether the quality bytes are text -- Phred bytes are 0..93, 0xFF = absent is no ASCII -- is looked at where the
    // qualities are copied: raw_call_input for the pipeline, UMIReader::next for the string form)
    if (same) {
      if (merged_generation_ != reader_.generation()) {
        keep_holds(reader_.holds());
        merged_generation_ = reader_.generation();
      }
      take(rec);
    } else {
      pend_ = rec;
      pend_holds_ = reader_.holds();
      have_pend_ = true;
      make_key();
      next_iteration_key_ = key;
      more = true;
      break;
    }
  }
  st.groups.push_back(g);
  return more;
}

bool UMIReader::next() {  // true = that was the final UMI
  UmiBatch st;
  const bool more = next_group(st);
  current_umi = st.umi_of(0);
  current_cell_barcode = st.cell_of(0);
  current_umi_group.assign(st.recs.size(), std::string());
  current_metadata_group.assign(st.recs.size(), std::vector<std::string>());
  for (size_t i = 0; i < st.recs.size(); ++i) {
    if (raw_quality_is_text(st.recs[i].body, st.recs[i])) {
      st.recs[i].qual_bad = 0;
    } else {
      puts("QUAL parsing warning: invalid utf-8 sequence");
      st.recs[i].qual_bad = 1;
    }
    raw_fields(st.recs[i].body, st.recs[i], current_metadata_group[i]);
    current_umi_group[i] = current_metadata_group[i][15];
  }
  return !more;
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
    if (const char *e = getenv("NIMBLE_GZIP_LEVEL")) level_ = std::min(9, std::max(1, atoi(e)));
    f_ = fopen(path.c_str(), "wb");
    if (!f_) throw Panic("could not create " + path);
    static const unsigned char head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    fwrite(head, 1, 10, f_);
    for (unsigned t = 0; t < std::max(1u, threads); ++t)
      if (!workers_.spawn([this] { work(); })) break;  // (fewer deflate threads than asked for: slower, not wrong)
    if (workers_.size() == 0) {
      fclose(f_);
      f_ = nullptr;
      throw Panic("could not start a thread for the gzip writer (thread limit reached)");
    }
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
    workers_.join();
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
  // validate_gzip (process/bam.rs:425-435) for a file this writer made: the file is read back and every deflate block --
  // they are independent, and their sizes are known here -- is inflated and checked against the size and CRC-32 of the text
  // it was made from, several blocks at a time; then the member's trailer.  (One gzread loop inflated the 2.4 GB of a 4 M-pair
  // run in 0.9 s, after everything else had finished.)
  bool validate(const std::string &path, unsigned threads) const {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char head[10];
    bool ok = fread(head, 1, 10, f) == 10 && head[0] == 0x1f && head[1] == 0x8b && head[2] == 8;
    std::atomic<bool> good{ok};
    const size_t GROUP = 64;  // blocks read per step (bounded memory), inflated in parallel
    for (size_t b0 = 0; b0 < blocks_.size() && good; b0 += GROUP) {
      const size_t b1 = std::min(blocks_.size(), b0 + GROUP);
      std::vector<std::string> comp(b1 - b0);
      for (size_t b = b0; b < b1 && good; ++b) {
        comp[b - b0].resize(blocks_[b].comp);
        if (fread(&comp[b - b0][0], 1, blocks_[b].comp, f) != blocks_[b].comp) good = false;
      }
      if (!good) break;
      std::atomic<size_t> next{b0};
      auto work = [&] {
        std::string out;
        for (;;) {
          const size_t b = next.fetch_add(1);
          if (b >= b1 || !good) return;
          const Block &B = blocks_[b];
          out.resize(B.raw + 1);
          z_stream z;
          memset(&z, 0, sizeof z);
          if (inflateInit2(&z, -15) != Z_OK) {
            good = false;
            return;
          }
          z.next_in = (Bytef *)comp[b - b0].data();
          z.avail_in = (uInt)B.comp;
          z.next_out = (Bytef *)&out[0];
          z.avail_out = (uInt)out.size();
          const int rc = inflate(&z, Z_SYNC_FLUSH);
          const bool last = b + 1 == blocks_.size();
          const bool fine = (last ? rc == Z_STREAM_END : (rc == Z_OK || rc == Z_BUF_ERROR)) && z.avail_in == 0 &&
                            z.total_out == B.raw && (uint32_t)crc32(0, (const Bytef *)out.data(), (uInt)B.raw) == B.crc;
          inflateEnd(&z);
          if (!fine) good = false;
        }
      };
      threads::run_beside(std::max(1u, threads) - 1, work);
    }
    unsigned char tail[8];
    if (good && fread(tail, 1, 8, f) == 8) {
      uint32_t crc = 0, isize = 0;
      for (int k = 0; k < 4; ++k) {
        crc |= (uint32_t)tail[k] << (8 * k);
        isize |= (uint32_t)tail[4 + k] << (8 * k);
      }
      if (crc != crc_ || isize != (uint32_t)total_) good = false;
      if (fgetc(f) != EOF) good = false;  // nothing behind the member
    } else {
      good = false;
    }
    fclose(f);
    return good;
  }

 private:
  static constexpr size_t BLOCK = 1u << 20;
  struct Block {
    uint64_t comp, raw;
    uint32_t crc;
  };
  std::vector<Block> blocks_;  // as written, in file order (validate)
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
      bool ok = deflateInit2(&z, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
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
        blocks_.push_back({(uint64_t)w.out.size(), (uint64_t)w.text.size(), w.crc});
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
  int level_ = 6;  // flate2's Compression::default()
  std::string cur_;
  std::vector<std::unique_ptr<Job>> jobs_;
  threads::Group workers_;
  std::mutex mu_;
  std::condition_variable cv_;
  size_t next_ = 0, written_ = 0;
  uint64_t total_ = 0;
  uint32_t crc_ = 0;
  bool stop_ = false, failed_ = false;
};


// a host buffer that keeps its size from batch to batch and is page-locked for the device's copy engine (pageable memory goes
// through the runtime's bounce buffers: 58 MB of bases and qualities per batch took 12 of a call's 16 ms)
struct PinBuf {
  uint8_t *p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  uint8_t *ensure(size_t n) {
    if (n <= cap) return p;
    release();
    cap = n + n / 4 + (1u << 20);
    p = static_cast<uint8_t *>(malloc(cap));
    if (!p) throw Panic("out of memory");
    pinned = nimble_pinned_register(p, cap) == 0;
    return p;
  }
  void release() {
    if (p && pinned) nimble_pinned_unregister(p);
    free(p);
    p = nullptr;
    cap = 0;
    pinned = false;
  }
  ~PinBuf() { release(); }
};

// body of a parallel loop over [0, n): `threads` workers take indices from a shared counter
template <class F>
void parallel_indices(size_t n, unsigned threads, F &&body) {
  if (n == 0) return;
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  std::string what;
  std::mutex mu;
  auto work = [&] {
    try {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n || failed) return;
        body(i);
      }
    } catch (const std::exception &e) {
      std::lock_guard<std::mutex> lk(mu);
      if (!failed) what = e.what();
      failed = true;
    }
  };
  const unsigned t = (unsigned)std::min<size_t>(std::max(1u, threads), n);
  threads::run_beside(t - 1, work);
  if (failed) throw Panic(what);
}

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
  parse::bam::UmiBatch store;
  size_t pairs = 0;

  double t_prep = 0, t_call = 0, t_rows = 0, t_write = 0, t_free = 0;
  PinBuf b_mem[2], q_mem[2];
  unsigned row_threads = std::max(1u, std::min(parse::usable_cpus() * 3 / 4, 12u));
  if (const char *e = getenv("NIMBLE_BAM_ROW_THREADS")) row_threads = (unsigned)std::max(1, atoi(e));
  auto flush = [&]() {
    if (store.groups.empty()) return;
    const size_t n_groups = store.groups.size();
    const auto tp0 = std::chrono::steady_clock::now();
    // the call's inputs: R1 = record 2k, R2 = record 2k + 1, each reverse-complemented when the BAM says the read was
    // (process/bam.rs:245-303); the quality strings are already in read direction (parse/bam.rs:270-287).  Lengths first
    // (one pass), then every pair's bases and qualities written into place by several threads.
    uint8_t *b[2] = {nullptr, nullptr}, *q[2] = {nullptr, nullptr};  // (not zero-filled: every byte is written below)
    std::vector<uint8_t> skip[2];
    std::vector<uint64_t> off[2] = {{0}, {0}};
    std::vector<uint32_t> seg;
    std::vector<size_t> first_pair(n_groups + 1, 0);
    std::vector<uint32_t> where;  // pair -> its first record
    uint32_t max_len = 1;
    for (size_t g = 0; g < n_groups; ++g) {
      const parse::bam::UmiBatch::Group &G = store.groups[g];
      first_pair[g] = seg.size();
      for (size_t k = 0; k + 1 < G.count; k += 2) {
        for (int m = 0; m < 2; ++m) {
          const parse::bam::Raw &R = store.recs[G.first + k + (size_t)m];
          const uint32_t len = R.l_seq == 124 ? R.l_seq - 13 : R.l_seq;
          off[m].push_back(off[m].back() + len);
          max_len = std::max<uint32_t>(max_len, len);
        }
        seg.push_back((uint32_t)g);
        where.push_back(G.first + (uint32_t)k);
      }
    }
    first_pair[n_groups] = seg.size();
    const uint64_t n = seg.size();
    for (int m = 0; m < 2; ++m) {
      b[m] = b_mem[m].ensure(off[m].back() + 1);
      q[m] = q_mem[m].ensure(off[m].back() + 1);
      skip[m].resize(n);
    }
    {
      const size_t CH = 4096;
      parallel_indices((n + CH - 1) / CH, row_threads, [&](size_t c) {
        for (size_t i = c * CH; i < std::min<size_t>(n, (c + 1) * CH); ++i) {
          for (int m = 0; m < 2; ++m) {
            const parse::bam::Raw &R = store.recs[where[i] + (size_t)m];
            if (!parse::bam::raw_quality_is_text(R.body, R)) {
              // (the reference warns and reads the field as empty; a call cannot trim a read without its qualities)
              puts("QUAL parsing warning: invalid utf-8 sequence");
              if (off[m][i + 1] != off[m][i]) {
                std::vector<std::string> md;
                parse::bam::raw_fields(R.body, R, md);
                throw Panic("BAM record without usable qualities (" + md[0] + "): not supported by the MI355X build");
              }
            }
            parse::bam::raw_call_input(R.body, R, b[m] + off[m][i], q[m] + off[m][i]);
            skip[m][i] = R.skip == 2 ? 1 : 0;
          }
        }
      });
    }
    for (size_t lib = 0; lib < n_lib && n; ++lib) {
      align::ReadBatch a, m;
      a.bases = b[0];
      a.offsets = off[0].data();
      a.n = n;
      a.max_len = max_len;
      m.bases = b[1];
      m.offsets = off[1].data();
      m.n = n;
      m.max_len = max_len;
      align::UmiExtras ex;
      ex.segment = seg.data();
      ex.n_segments = (uint32_t)n_groups;
      ex.qual[0] = q[0];
      ex.qual[1] = q[1];
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
      // rows of a group (res.rows is sorted by segment): the groups are independent, their text is made by several threads
      // and written in group order
      std::vector<size_t> row_first(n_groups + 1, 0);
      {
        size_t row = 0;
        for (size_t g = 0; g < n_groups; ++g) {
          row_first[g] = row;
          while (row < res.rows.size() && res.rows[row].segment == (uint32_t)g) ++row;
        }
        row_first[n_groups] = row;
      }
      const size_t GCH = 256;  // groups per piece of text
      std::vector<std::string> text((n_groups + GCH - 1) / GCH);
      parallel_indices(text.size(), row_threads, [&](size_t piece) {
        std::string &line = text[piece];
        std::vector<uint32_t> last;                          // (made once per piece, not per group: most groups are a pair or two)
        std::vector<std::pair<const char *, uint32_t>> scored;  // QNAMEs of the reads that stand for a callset
        std::string f;
        for (size_t g = piece * GCH; g < std::min(n_groups, (piece + 1) * GCH); ++g) {
          const parse::bam::Raw *const recs = store.recs.data() + store.groups[g].first;
          const size_t p0 = first_pair[g], np = first_pair[g + 1] - p0;
          const size_t row0 = row_first[g], row1 = row_first[g + 1];
          if (row1 == row0) continue;  // `if s.len() == 0 { results.push(vec![]) }`: nothing at all for this UMI
          // filter_reasons is keyed by the read key (R1 string + R2 string): a later pair with the same key replaces an
          // earlier one (align.rs:591-600)
          // last[k] = the last pair of the group with pair k's key
          last.assign(np, 0u);
          auto same_key = [&](size_t x, size_t y) {
            const uint64_t i = p0 + x, j = p0 + y;
            const uint64_t a0 = off[0][i + 1] - off[0][i], a1 = off[1][i + 1] - off[1][i];
            // (the key is the concatenation R1 + R2: equal as strings, not mate by mate -- compare it as one)
            if (a0 + a1 != (off[0][j + 1] - off[0][j]) + (off[1][j + 1] - off[1][j])) return false;
            if (a0 == off[0][j + 1] - off[0][j])
              return memcmp(b[0] + off[0][i], b[0] + off[0][j], a0) == 0 && memcmp(b[1] + off[1][i], b[1] + off[1][j], a1) == 0;
            std::string kx(reinterpret_cast<const char *>(b[0] + off[0][i]), a0), ky(reinterpret_cast<const char *>(b[0] + off[0][j]), off[0][j + 1] - off[0][j]);
            kx.append(reinterpret_cast<const char *>(b[1] + off[1][i]), a1);
            ky.append(reinterpret_cast<const char *>(b[1] + off[1][j]), off[1][j + 1] - off[1][j]);
            return kx == ky;
          };
          if (np <= 48) {
            for (size_t k = 0; k < np; ++k) {
              uint32_t l = (uint32_t)k;
              for (size_t j = np; j-- > k + 1;)
                if (same_key(k, j)) {
                  l = (uint32_t)j;
                  break;
                }
              last[k] = l;
            }
          } else {
            std::unordered_map<std::string, uint32_t> last_of_key;
            auto key_of = [&](size_t pair) {
              const uint64_t i = p0 + pair;
              std::string k(reinterpret_cast<const char *>(b[0] + off[0][i]), off[0][i + 1] - off[0][i]);
              k.append(reinterpret_cast<const char *>(b[1] + off[1][i]), off[1][i + 1] - off[1][i]);
              return k;
            };
            for (size_t k = 0; k < np; ++k) last_of_key[key_of(k)] = (uint32_t)k;
            for (size_t k = 0; k < np; ++k) last[k] = last_of_key.at(key_of(k));
          }
          auto qname_of = [&](size_t rec) {
            const uint8_t *body = recs[rec].body;
            const uint32_t ln = body[8];
            return std::make_pair(reinterpret_cast<const char *>(body + 32), ln ? ln - 1 : 0u);
          };
          scored.clear();
          auto emit = [&](const std::string &features, int32_t count, size_t pair) {
            const parse::bam::Raw &m1 = recs[2 * pair], &m2 = recs[2 * pair + 1];
            const align::FilterRecord &fr = res.per_read[p0 + last[pair]];
            line += features;
            line += '\t';
            parse::bam::append_int(line, count);
            line += '\t';
            parse::bam::raw_row_fields(m2.body, m2, line);
            line += '\t';
            parse::bam::raw_row_fields(m1.body, m1, line);
            line += '\t';
            line += align::to_string(fr.r2);  // "r1": the mate
            line += '\t';
            parse::bam::append_int(line, fr.score2);
            line += '\t';
            line += align::to_string(align::FilterReason::None);
            line += "\t0\t";
            line += align::to_string(fr.r1);  // "r2": the first read
            line += '\t';
            parse::bam::append_int(line, fr.score1);
            line += '\t';
            line += align::to_string(align::FilterReason::None);
            line += "\t0\t";
            line += align::to_string(fr.triage);
            line += "\tNone\n";
          };
          for (size_t r = row0; r < row1; ++r) {
            const align::UmiRow &R = res.rows[r];
            const size_t pair = R.representative - p0;
            scored.push_back(qname_of(2 * pair));  // score.1.1[0]: the first read's QNAME
            f.clear();
            const std::vector<std::string> &feat = res.features(R);
            for (size_t t = 0; t < feat.size(); ++t) {
              if (t) f += ',';
              f += feat[t];
            }
            emit(f, R.count, pair);
          }
          for (size_t k = 0; k < np; ++k) {  // pairs that stand for no callset: an empty call (process/bam.rs:341-355)
            const auto qn = qname_of(2 * k + 1);
            bool is_scored = false;
            for (const auto &sq : scored) is_scored |= sq.second == qn.second && memcmp(sq.first, qn.first, qn.second) == 0;
            if (is_scored) continue;
            emit("", 0, k);
          }
        }
      });
      const auto tw0 = std::chrono::steady_clock::now();
      for (const std::string &t : text)
        if (!t.empty()) write_line(lib, t);
      t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
    }
    const auto tf0 = std::chrono::steady_clock::now();
    store.clear();
    t_free += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
    pairs = 0;
  };

  // The reader runs on a thread of its own, one batch ahead (the reference has a reader thread and a pool of consumers,
  // process/bam.rs:127-226): BGZF, record decoding and the 38 strings per record are the pipeline's largest cost and
  // overlap the device call and the writing of the batch before.
  puts("Spawning reader thread.");
  struct Batch {
    parse::bam::UmiBatch store;
    std::string error;
    bool last = false;
  };
  std::mutex qmu;
  std::condition_variable qcv;
  std::deque<std::unique_ptr<Batch>> queue;
  bool quit = false;
  double t_read = 0, t_read_blocked = 0;
  std::thread reader_thread([&] {
    std::unique_ptr<Batch> cur(new Batch());
    size_t cur_pairs = 0;
    auto push = [&](bool last) {
      cur->last = last;
      const auto tb = std::chrono::steady_clock::now();
      std::unique_lock<std::mutex> lk(qmu);
      qcv.wait(lk, [&] { return quit || queue.size() < 2; });
      t_read_blocked += std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count();
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
        const bool final_umi = !reader.next_group(cur->store);
        if (final_umi && has_aligned) {
          // (the group just read is the file's last: collected, never sent -- process/bam.rs:163-178)
          const parse::bam::UmiBatch::Group last = cur->store.groups.back();
          cur->store.groups.pop_back();
          cur->store.recs.resize(last.first);
          puts("Finished reading UMIs from input file.");
          break;
        }
        cur_pairs += cur->store.groups.back().count / 2;
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
      store = std::move(bt->store);
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
    fprintf(stderr, "[nimble host] bam pipeline: reader thread %.2f s (BGZF, records, UMI groups; %.2f of it waiting for the consumer); consumer %.2f s waiting for it, "
            "%.2f s in calls and rows (%.2f preparing the calls' inputs, %.2f in the calls, %.2f making and writing rows, of which "
            "%.2f handing them to the gzip writer; %.2f freeing the batch)\n", t_read, t_read_blocked, t_wait, t_flush, t_prep, t_call,
            t_rows, t_write, t_free);
  for (size_t i = 0; i < n_lib; ++i) {
    if (out[i]->close()) printf("Successfully flushed and closed file %zu\n", i);
    else fprintf(stderr, "Error finishing GZIP for file %zu\n", i);
  }
  for (size_t i = 0; i < output_paths.size(); ++i) {  // validate_gzip (process/bam.rs:425-435)
    const std::string &p = output_paths[i];
    printf("Validating GZIP file: %s\n", p.c_str());
    if (i < n_lib && out[i]->validate(p, gz_threads)) printf("Validation successful for %s\n", p.c_str());
    else fprintf(stderr, "GZIP validation failed for %s\n", p.c_str());
  }
  puts("Logging thread terminating.");
  puts("Joined on logging; terminating.");
}

}  // namespace bam
}  // namespace process
}  // namespace nimble
