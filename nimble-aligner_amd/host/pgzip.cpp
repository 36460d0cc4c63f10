// pgzip.cpp -- a single gzip stream inflated by many threads.
//
// The reference opens .fastq.gz through niffler + flate2 (src/parse/fastq.rs:21-43): one thread, one stream.  A deflate
// stream has no index, but it can still be cut: a worker that starts in the middle (1) finds the start of a deflate block
// by trying bit positions until a block header and the symbols behind it make sense, and (2) decodes from there without
// knowing the 32 KiB of history the stream may refer back to -- a back-reference into that unknown window is kept as a
// SYMBOL ("byte k of the window") in a 16-bit output.  The chunks are joined in file order: the worker of chunk i keeps
// decoding until it stands, at a block boundary, exactly on the bit where chunk i + 1 began -- which proves that guess
// right (a guess it runs past was wrong: its work is dropped and chunk i simply goes on) -- then chunk i + 1's symbols are
// replaced by the now known bytes.  What comes out is the stream's content, byte for byte, and every member's CRC-32 and
// length are checked against its trailer as zlib would.  (The idea is that of pugz, Kerbiriou & Chikhi 2019; the code
// is this repository's own.)
#include <immintrin.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "nimble_host.hpp"

namespace nimble {
namespace parse {
unsigned usable_cpus() { return threads::usable_cpus(); }  // (csrc/threads.h: affinity mask, cgroup quota, NIMBLE_CPUS)
}  // namespace parse

namespace parse {
namespace pgzip {

void *huge_map(size_t bytes, void **map, size_t *map_bytes) {
  const size_t H = 2u << 20;
  const size_t want = ((bytes + H - 1) / H) * H + H;
  void *m = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (m == MAP_FAILED) throw Panic("out of memory (mmap refused a buffer)");
  void *a = (void *)(((uintptr_t)m + H - 1) & ~(uintptr_t)(H - 1));
  static const bool huge = getenv("NIMBLE_NO_HUGEPAGES") == nullptr;
  if (huge) (void)madvise(a, want - H, MADV_HUGEPAGE);
  *map = m;
  *map_bytes = want;
  return a;
}
void huge_unmap(void *map, size_t map_bytes) { munmap(map, map_bytes); }

namespace {

constexpr uint32_t WSIZE = 32768;
constexpr uint16_t SYM0 = 256;  // output value SYM0 + k = byte k of the unknown 32 KiB window in front of the chunk

struct BitReader {
  const uint8_t *base;
  size_t size;      // bytes
  size_t pos;       // next byte to load
  uint64_t buf = 0;
  uint32_t cnt = 0;  // valid bits in buf
  bool over = false;
  BitReader(const uint8_t *b, size_t n, uint64_t bit) : base(b), size(n), pos((size_t)(bit >> 3)) {
    const uint32_t skip = (uint32_t)(bit & 7);
    refill();
    if (skip) drop(skip);
  }
  inline void refill() {
    if (pos + 8 <= size) {  // one unaligned load; the bits above `cnt` it brings along are the stream's own next bits
      uint64_t w;
      memcpy(&w, base + pos, 8);
      buf |= w << cnt;
      pos += (63 - cnt) >> 3;
      cnt |= 56;
      return;
    }
    while (cnt <= 56) {
      if (pos < size) buf |= (uint64_t)base[pos] << cnt;
      else if (pos >= size + 16) over = true;  // far past the data: whoever decodes here is decoding zeros
      ++pos;
      cnt += 8;
    }
  }
  inline uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ULL << n) - 1)); }
  inline void drop(uint32_t n) { buf >>= n; cnt -= n; }
  inline uint32_t take(uint32_t n) {
    if (cnt < n) refill();
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  uint64_t bit_pos() const { return (uint64_t)pos * 8 - cnt; }  // (bytes past the end count as loaded zeros)
  bool past_end() const { return bit_pos() > (uint64_t)size * 8; }
};

// canonical Huffman decoding table: entry = symbol | length << 16, indexed by `bits` low bits (LSB-first codes reversed)
struct Huff {
  std::vector<uint32_t> tab;
  uint32_t bits = 0;
  // returns false when the lengths are no complete prefix code (over-subscribed or incomplete; a single code of length 1 is
  // allowed for distances as zlib allows it)
  bool build(const uint8_t *len, uint32_t n, bool allow_incomplete) {
    uint32_t count[16] = {0};
    for (uint32_t i = 0; i < n; ++i) count[len[i]]++;
    if (count[0] == n) return false;
    uint32_t maxl = 15;
    while (maxl && !count[maxl]) --maxl;
    int32_t left = 1;
    for (uint32_t l = 1; l <= 15; ++l) {
      left <<= 1;
      left -= (int32_t)count[l];
      if (left < 0) return false;
    }
    if (left > 0 && !(allow_incomplete && maxl == 1)) return false;  // zlib: incomplete only for a lone 1-bit code
    bits = maxl;
    tab.assign((size_t)1 << bits, 0);
    uint32_t next[16], code = 0;
    for (uint32_t l = 1; l <= 15; ++l) {
      code = (code + count[l - 1]) << 1;
      next[l] = code;
    }
    next[0] = 0;
    for (uint32_t s = 0; s < n; ++s) {
      const uint32_t l = len[s];
      if (!l) continue;
      uint32_t c = next[l]++, r = 0;
      for (uint32_t i = 0; i < l; ++i) r |= ((c >> i) & 1u) << (l - 1 - i);
      for (uint32_t k = r; k < tab.size(); k += 1u << l) tab[k] = s | (l << 16);
    }
    return true;
  }
};

const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// Two-level decoding table for the literal/length and the distance code.  A 15-bit flat table (128 KiB) falls out of the
// L1 cache and costs more to fill, once per block, than the block costs to decode; here the first level takes PRIMARY bits
// and the few longer codes go through small second-level tables behind it.  An entry carries what the decoder needs without
// a second lookup:
//   bits  7..0   how many bits to drop (first level: the code's length, or PRIMARY for a pointer; second level: the rest)
//   bits 12..8   number of extra bits (length / distance), or the width of the second-level table for a pointer
//   bit  13      end of block      bit 14  pointer to a second-level table     bit 15  literal
//   bits 31..16  the literal, the base length or distance, or the pointer's table offset
// and 0 marks a bit pattern no code has (or a symbol that must not appear: 286/287, 30/31, or -- in a trial decode that
// only accepts FASTQ text -- a literal that is no text).
constexpr uint32_t E_LIT = 0x8000u, E_SUB = 0x4000u, E_EOB = 0x2000u;
struct FastTable {
  std::vector<uint32_t> tab;
  uint32_t primary = 0;
  std::vector<uint8_t> sub_bits;
  std::vector<uint16_t> rev;

  static inline bool text(uint32_t c) { return c == '\n' || c == '\r' || c == '\t' || (c >= 32 && c < 127); }
  static uint32_t payload(uint32_t s, bool dist, bool strict) {
    if (dist) return s < 30 ? ((uint32_t)DBASE[s] << 16) | ((uint32_t)DEXT[s] << 8) : 0;
    if (s < 256) return strict && !text(s) ? 0 : (s << 16) | E_LIT;
    if (s == 256) return E_EOB;
    s -= 257;
    return s < 29 ? ((uint32_t)LBASE[s] << 16) | ((uint32_t)LEXT[s] << 8) : 0;
  }
  // false when the lengths are no complete prefix code (a lone 1-bit code is allowed, as zlib allows it); `complete` =
  // false skips that check (the fixed distance code has 30 of 32 codes)
  bool build(const uint8_t *len, uint32_t n, uint32_t primary_bits, bool dist, bool strict, bool complete = true) {
    uint32_t count[16] = {0};
    for (uint32_t i = 0; i < n; ++i) count[len[i]]++;
    if (count[0] == n) return false;
    uint32_t maxl = 15;
    while (maxl && !count[maxl]) --maxl;
    int32_t left = 1;
    for (uint32_t l = 1; l <= 15; ++l) {
      left <<= 1;
      left -= (int32_t)count[l];
      if (left < 0) return false;
    }
    if (complete && left > 0 && maxl != 1) return false;
    primary = std::min(primary_bits, maxl);
    const uint32_t psize = 1u << primary;
    tab.assign(psize, 0);
    uint32_t next[16], code = 0;
    next[0] = 0;
    for (uint32_t l = 1; l <= 15; ++l) {
      code = (code + count[l - 1]) << 1;
      next[l] = code;
    }
    rev.resize(n);
    const bool longer = maxl > primary;
    if (longer) sub_bits.assign(psize, 0);
    for (uint32_t s = 0; s < n; ++s) {
      const uint32_t l = len[s];
      if (!l) continue;
      uint32_t c = next[l]++, r = 0;
      for (uint32_t i = 0; i < l; ++i) r |= ((c >> i) & 1u) << (l - 1 - i);
      rev[s] = (uint16_t)r;
      if (l <= primary) {
        const uint32_t e = payload(s, dist, strict);
        if (e)
          for (uint32_t k = r; k < psize; k += 1u << l) tab[k] = e | l;
      } else {
        uint8_t &sb = sub_bits[r & (psize - 1)];
        sb = std::max<uint8_t>(sb, (uint8_t)(l - primary));
      }
    }
    if (!longer) return true;
    for (uint32_t s = 0; s < n; ++s) {
      const uint32_t l = len[s];
      if (l <= primary) continue;
      const uint32_t r = rev[s], pre = r & (psize - 1), sb = sub_bits[pre];
      if (!(tab[pre] & E_SUB)) {
        const uint32_t at = (uint32_t)tab.size();
        if (at > 0xFFFFu) return false;
        tab.resize(at + (1u << sb), 0);
        tab[pre] = (at << 16) | E_SUB | (sb << 8) | primary;
      }
      const uint32_t at = tab[pre] >> 16, e = payload(s, dist, strict);
      if (e)
        for (uint32_t k = r >> primary; k < (1u << sb); k += 1u << (l - primary)) tab[at + k] = e | (l - primary);
    }
    return true;
  }
};
constexpr uint32_t LIT_PRIMARY = 11, DIST_PRIMARY = 9;

struct FixedTables {
  FastTable lit, dist;
  FixedTables() {
    uint8_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    lit.build(l, 288, LIT_PRIMARY, false, false);
    uint8_t d[32];
    for (int i = 0; i < 32; ++i) d[i] = 5;
    dist.build(d, 32, DIST_PRIMARY, true, false);
  }
};
const FixedTables &fixed_tables() {
  static const FixedTables t;
  return t;
}
// the same, for a trial decode that only accepts text
const FixedTables &fixed_tables_strict() {
  static const FixedTables t = [] {
    FixedTables f;
    uint8_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    f.lit.build(l, 288, LIT_PRIMARY, false, true);
    return f;
  }();
  return t;
}

// One decoder position: decodes deflate blocks into 16-bit output (bytes, or SYM0 + k for bytes of the unknown window).
// out[0 .. WSIZE) stands for the window itself, the decoded data follows; `n` entries of `out` are in use (the vector is
// kept larger than that and cut to size by finish()).
struct Inflater {
  BitReader br;
  HugeBuf<uint16_t> &out;
  size_t n;
  bool strict_text;  // trial decoding: every literal must be a FASTQ character
  Huff hc;  // kept across blocks: the tables are re-filled, not re-allocated
  FastTable dyn_lit, dyn_dist;
  Inflater(const uint8_t *b, size_t sz, uint64_t bit, HugeBuf<uint16_t> &o, bool strict)
      : br(b, sz, bit), out(o), n(o.size()), strict_text(strict) {}
  void reset(uint64_t bit) {
    br = BitReader(br.base, br.size, bit);
    n = WSIZE;
  }
  void finish() { out.resize(n); }
  inline uint16_t *room(size_t need) {
    if (n + need > out.size()) out.resize(std::max(out.size() * 2, n + need + 65536));
    return out.data();
  }
  static inline bool text(uint32_t c) { return c == '\n' || c == '\r' || c == '\t' || (c >= 32 && c < 127); }

  // decode one block; *final = its BFINAL bit.  `limit`: stop early (trial) once n exceeds it.  false = not valid.
  bool block(bool *final, size_t limit = ~(size_t)0) {
    br.refill();
    *final = br.take(1) != 0;
    const uint32_t type = br.take(2);
    if (type == 3) return false;
    if (type == 0) {
      br.drop(br.cnt & 7);  // to the byte boundary
      br.refill();
      const uint32_t len = br.take(16), nlen = br.take(16);
      if ((len ^ nlen) != 0xFFFFu) return false;
      // the bit buffer holds whole bytes now: give them back and copy from the input
      size_t at = (size_t)(br.bit_pos() >> 3);
      if (at + len > br.size) return false;
      uint16_t *o = room(len) + n;
      const uint8_t *src = br.base + at;
      for (uint32_t i = 0; i < len; ++i) {
        if (strict_text && !text(src[i])) return false;
        o[i] = src[i];
      }
      n += len;
      br = BitReader(br.base, br.size, (uint64_t)(at + len) * 8);
      return true;
    }
    const FastTable *hl, *hd;
    if (type == 1) {
      const FixedTables &f = strict_text ? fixed_tables_strict() : fixed_tables();
      hl = &f.lit;
      hd = &f.dist;
    } else {
      br.refill();
      const uint32_t hlit = br.take(5) + 257, hdist = br.take(5) + 1, hclen = br.take(4) + 4;
      if (hlit > 286 || hdist > 30) return false;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl[19] = {0};
      for (uint32_t i = 0; i < hclen; ++i) {
        if (br.cnt < 3) br.refill();
        cl[order[i]] = (uint8_t)br.take(3);
      }
      if (!hc.build(cl, 19, false)) return false;
      uint8_t lens[286 + 30];
      uint32_t i = 0;
      while (i < hlit + hdist) {
        br.refill();
        const uint32_t e = hc.tab[br.peek(hc.bits)];
        if (!(e >> 16)) return false;
        br.drop(e >> 16);
        const uint32_t s = e & 0xFFFF;
        if (s < 16) {
          lens[i++] = (uint8_t)s;
        } else {
          uint32_t rep, val = 0;
          if (s == 16) {
            if (i == 0) return false;
            val = lens[i - 1];
            rep = 3 + br.take(2);
          } else if (s == 17) {
            rep = 3 + br.take(3);
          } else {
            rep = 11 + br.take(7);
          }
          if (i + rep > hlit + hdist) return false;
          while (rep--) lens[i++] = (uint8_t)val;
        }
      }
      if (lens[256] == 0) return false;  // no end-of-block code
      if (!dyn_lit.build(lens, hlit, LIT_PRIMARY, false, strict_text)) return false;
      if (!dyn_dist.build(lens + hlit, hdist, DIST_PRIMARY, true, false)) {
        // a block without any distance code (all literals) is legal: a table that refuses every distance
        bool none = true;
        for (uint32_t k = 0; k < hdist; ++k) none = none && lens[hlit + k] == 0;
        if (!none) return false;
        dyn_dist.primary = 1;
        dyn_dist.tab.assign(2, 0);
      }
      hl = &dyn_lit;
      hd = &dyn_dist;
    }
    const uint32_t *lt = hl->tab.data(), *dt = hd->tab.data();
    const uint64_t lmask = (1u << hl->primary) - 1, dmask = (1u << hd->primary) - 1;
    uint16_t *o = room(1024);
    size_t cap = out.size();
    for (;;) {
      if (n + 600 > cap) {
        o = room(65536);
        cap = out.size();
      }
      // one refill (at least 56 bits) covers a length code with its extra bits and a distance code with its own
      // (15 + 5 + 15 + 13 = 48), or three literals
      br.refill();
      if (br.over) return false;
      uint32_t e = lt[br.buf & lmask];
      if (e & E_LIT) {
        br.drop(e & 0xFF);
        o[n++] = (uint16_t)(e >> 16);
        e = lt[br.buf & lmask];
        if (e & E_LIT) {
          br.drop(e & 0xFF);
          o[n++] = (uint16_t)(e >> 16);
          e = lt[br.buf & lmask];
          if (e & E_LIT) {
            br.drop(e & 0xFF);
            o[n++] = (uint16_t)(e >> 16);
            if (n > limit) return true;
            continue;
          }
        }
        if (n > limit) return true;
        if (br.cnt < 48) br.refill();
      }
      if (e & E_SUB) {
        br.drop(e & 0xFF);
        e = lt[(e >> 16) + (uint32_t)(br.buf & ((1u << ((e >> 8) & 0x1F)) - 1))];
        if (e & E_LIT) {
          br.drop(e & 0xFF);
          o[n++] = (uint16_t)(e >> 16);
          continue;
        }
      }
      if (!e) return false;
      br.drop(e & 0xFF);
      if (e & E_EOB) break;
      uint32_t x = (e >> 8) & 0x1F;
      const uint32_t len = (e >> 16) + (uint32_t)(br.buf & ((1u << x) - 1));
      br.drop(x);
      uint32_t de = dt[br.buf & dmask];
      if (de & E_SUB) {
        br.drop(de & 0xFF);
        de = dt[(de >> 16) + (uint32_t)(br.buf & ((1u << ((de >> 8) & 0x1F)) - 1))];
      }
      if (!de) return false;
      br.drop(de & 0xFF);
      x = (de >> 8) & 0x1F;
      const uint32_t dist = (de >> 16) + (uint32_t)(br.buf & ((1u << x) - 1));
      br.drop(x);
      if (dist > n) return false;  // before the window (cannot happen: the data starts behind WSIZE window entries)
      uint16_t *dst = o + n;
      const uint16_t *src = dst - dist;
      if (dist >= 8) {
        for (uint32_t k = 0; k < len; k += 8) memcpy(dst + k, src + k, 16);  // (room() left slack behind n + len)
      } else {
        for (uint32_t k = 0; k < len; ++k) dst[k] = src[k];
      }
      n += len;
      if (n > limit) return true;
    }
    return !br.past_end();
  }
};

// gzip member header at byte `p`; returns the offset of the deflate data or 0 when there is no valid header
size_t gzip_header(const uint8_t *d, size_t n, size_t p) {
  if (p + 10 > n || d[p] != 0x1f || d[p + 1] != 0x8b || d[p + 2] != 8) return 0;
  const uint8_t flg = d[p + 3];
  size_t q = p + 10;
  if (flg & 4) {
    if (q + 2 > n) return 0;
    q += 2 + (d[q] | (d[q + 1] << 8));
  }
  if (flg & 8) {
    while (q < n && d[q]) ++q;
    ++q;
  }
  if (flg & 16) {
    while (q < n && d[q]) ++q;
    ++q;
  }
  if (flg & 2) q += 2;
  return q <= n ? q : 0;
}

}  // namespace

struct Reader::Impl {
  int fd = -1;
  const uint8_t *data = nullptr;
  size_t size = 0;
  unsigned threads = 1;
  size_t chunk = 0;

  struct Chunk {
    uint64_t nominal = 0;          // compressed byte offset the worker was given
    uint64_t start_bit = 0;        // block start it found (or, chunk 0, the first block of the first member)
    bool found = false;
    HugeBuf<uint16_t> sym;         // WSIZE window entries, then the decoded data up to `end_bit`
    uint64_t end_bit = 0;          // where decoding stopped: at a block boundary
    bool ended_stream = false;     // the worker met the end of the file
    std::vector<uint64_t> member_ends;  // positions in the chunk's data (index into sym - WSIZE) where a member ended,
    std::vector<uint32_t> member_crc, member_isize;  // ... with that member's trailer
    bool done = false;
  };
  std::vector<std::unique_ptr<Chunk>> chunks;
  std::mutex mu;
  std::condition_variable cv;
  threads::Group workers;
  std::atomic<size_t> next_job{0};
  size_t in_flight_limit = 0;
  size_t consumed = 0;  // chunks the consumer has taken (workers stay at most in_flight_limit ahead)
  bool stop = false;

  std::mutex spare_mu;
  std::vector<HugeBuf<uint16_t>> spare;  // symbol arrays handed back (Reader::recycle)
  void give_back(HugeBuf<uint16_t> &&b) {
    if (!b.capacity()) return;
    std::lock_guard<std::mutex> lk(spare_mu);
    if (spare.size() < (size_t)threads * 2 + 4) spare.push_back(std::move(b));
  }

  // consumer state
  size_t cur = 0;                     // chunk being emitted
  std::vector<uint8_t> window;        // last WSIZE bytes emitted (the history for the next chunk)
  bool finished = false;
  std::deque<Piece> ready;

  // hand a decoded stretch over as a piece and move the window on by it (only the last WSIZE bytes are resolved here; the
  // rest is resolved by whoever takes the piece, in parallel)
  void emit(Chunk &c) {
    Piece p;
    p.window = window;
    p.member_ends = std::move(c.member_ends);
    p.member_crc = std::move(c.member_crc);
    p.member_isize = std::move(c.member_isize);
    p.sym = std::move(c.sym);
    const size_t n = p.size();
    const size_t k = std::min<size_t>(n, WSIZE);
    std::vector<uint8_t> tail(k);
    const uint16_t *s = p.sym.data() + WSIZE + (n - k);
    for (size_t i = 0; i < k; ++i) tail[i] = s[i] < SYM0 ? (uint8_t)s[i] : p.window[s[i] - SYM0];
    if (k == WSIZE) {
      window = tail;
    } else if (k) {  // (a piece without output leaves the window as it is)
      memmove(window.data(), window.data() + k, WSIZE - k);
      memcpy(window.data() + WSIZE - k, tail.data(), k);
    }
    ready.push_back(std::move(p));
  }

  // decode sequentially from `at_bit` until `until_bit` is reached at a block boundary (or passed), or the stream ends;
  // what is decoded goes out as a piece.  Returns the bit position reached; *ended = the file's last member ended.
  uint64_t bridge(uint64_t at_bit, uint64_t until_bit, bool *ended) {
    Chunk br;
    br.sym.resize(WSIZE);
    for (uint32_t k = 0; k < WSIZE; ++k) br.sym[k] = (uint16_t)(SYM0 + k);
    Inflater inf(data, size, at_bit, br.sym, false);
    *ended = false;
    while (inf.br.bit_pos() < until_bit) {
      bool fin = false;
      if (!inf.block(&fin)) throw Panic("Error -- corrupt or truncated deflate data");
      if (!fin) continue;
      inf.br.drop(inf.br.cnt & 7);
      const size_t tr = (size_t)(inf.br.bit_pos() >> 3);
      if (tr + 8 > size) throw Panic("Error -- truncated gzip file");
      uint32_t crc32v, isize;
      memcpy(&crc32v, data + tr, 4);
      memcpy(&isize, data + tr + 4, 4);
      br.member_ends.push_back(inf.n - WSIZE);
      br.member_crc.push_back(crc32v);
      br.member_isize.push_back(isize);
      size_t nx = tr + 8;
      while (nx < size && data[nx] == 0) ++nx;
      if (nx >= size) {
        *ended = true;
        break;
      }
      const size_t h = gzip_header(data, size, nx);
      if (!h) throw Panic("Error -- trailing garbage in gzip file");
      inf.br = BitReader(data, size, (uint64_t)h * 8);
    }
    const uint64_t reached = *ended ? (uint64_t)size * 8 : inf.br.bit_pos();
    inf.finish();
    emit(br);
    return reached;
  }

  // chunk `cur` goes out in full: its own data and -- where the chunks behind it guessed wrong or found nothing -- the
  // data up to the start of the next chunk that guessed right, decoded here
  void join_step() {
    if (size == 0) {
      finished = true;
      return;
    }
    Chunk *ch;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return chunks[cur]->done; });
      ch = chunks[cur].get();
    }
    if (cur == 0 && !ch->found) throw Panic("Error -- could not determine compression format (no gzip header)");
    uint64_t at_bit = ch->end_bit;
    bool ended = ch->ended_stream;
    emit(*ch);
    size_t nxt = cur + 1;
    // join: the next chunk counts only if it began exactly where this stream stands, at a block boundary
    while (!ended && nxt < chunks.size()) {
      Chunk *nc;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return chunks[nxt]->done; });
        nc = chunks[nxt].get();
      }
      if (nc->found && nc->start_bit == at_bit) {  // proven: a block boundary of the real stream
        ++joined_direct;
        break;
      }
      if (nc->found && nc->start_bit > at_bit) {
        // the stream has not reached that guess yet (the block at our stop bit was none the search accepts, or an earlier
        // guess was dropped): decode on, here, to the first block boundary at or behind it
        at_bit = bridge(at_bit, nc->start_bit, &ended);
        ++bridged;
        if (ended || nc->start_bit == at_bit) break;
        // ran past the guess: it was no block boundary
      }
      give_back(std::move(nc->sym));  // found nothing, or a guess the stream has passed: that work is dropped
      ++nxt;
      {
        std::lock_guard<std::mutex> lk(mu);  // the workers may go on to the chunks behind it
        consumed = nxt;
      }
      cv.notify_all();
    }
    if (!ended && nxt >= chunks.size()) {
      bridge(at_bit, ~0ULL, &ended);  // no further chunk to join: the rest of the file, here
      if (!ended) throw Panic("Error -- truncated gzip file");
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      consumed = nxt;
    }
    cv.notify_all();
    cur = nxt;
    if (ended || cur >= chunks.size()) finished = true;
    if (finished && getenv("NIMBLE_GZIP_DEBUG")) {
      size_t found = 0;
      for (auto &c : chunks) found += c->found ? 1 : 0;
      fprintf(stderr, "[pgzip] %zu chunks, %zu found a block start, %zu joined without a bridge, %zu bridged\n", chunks.size(), found,
              joined_direct, bridged);
    }
  }
  size_t joined_direct = 0, bridged = 0;

  ~Impl() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    workers.join();
    if (data) munmap((void *)data, size);
    if (fd >= 0) close(fd);
  }

  // decode from (chunk c's start) until the nominal start region of later chunks ends or a limit of output is reached; the
  // worker stops at the first block boundary at or behind the bit where the NEXT chunk's search began
  void run_chunk(size_t c) {
    Chunk &ch = *chunks[c];
    {
      std::lock_guard<std::mutex> lk(spare_mu);  // memory that has been through once: its pages are there already
      if (!spare.empty()) {
        ch.sym = std::move(spare.back());
        spare.pop_back();
      }
    }
    ch.sym.reserve(WSIZE + 12 * chunk);  // (address space; deflate seldom does better than 1:8 on FASTQ)
    ch.sym.resize(WSIZE);
    for (uint32_t k = 0; k < WSIZE; ++k) ch.sym[k] = (uint16_t)(SYM0 + k);
    uint64_t bit = 0;
    if (c == 0) {
      const size_t h = gzip_header(data, size, 0);
      if (!h) {
        ch.found = false;
        return;
      }
      bit = (uint64_t)h * 8;
      ch.found = true;
    } else {
      // try bit positions from the nominal offset on: a non-final dynamic block whose header is valid and whose first
      // 8 KiB of output are text
      const uint64_t lo = ch.nominal * 8, hi = std::min<uint64_t>((ch.nominal + chunk) * 8, (uint64_t)size * 8);
      static thread_local HugeBuf<uint16_t> trial;  // (a trial only appends: the window placeholders stay)
      if (trial.empty()) {
        trial.resize(WSIZE);
        memcpy(trial.data(), ch.sym.data(), WSIZE * sizeof(uint16_t));
      }
      Inflater probe(data, size, lo, trial, true);
      // (deflate writers close a block every few ten kB of output at the latest; a chunk that shows no block start in its
      // first SEARCH bytes -- stored data, something incompressible -- is left to the stream that arrives from the left)
      const uint64_t SEARCH = 512u << 10;
      const uint64_t hi_s = std::min<uint64_t>(hi, lo + SEARCH * 8);
      for (uint64_t p = lo; p + 64 < hi_s && !ch.found; ++p) {
        const size_t by = (size_t)(p >> 3);
        if ((p & 7) == 0 && data[by] == 0x1f && data[by + 1] == 0x8b && data[by + 2] == 8) {
          // the start of a gzip member (files written in blocks -- bgzip -- are nothing but members): its first block,
          // final or not, with nothing in front of it to refer to
          const size_t h = gzip_header(data, size, by);
          if (h && h + 8 < size) {
            probe.reset((uint64_t)h * 8);
            bool fin = false;
            if (probe.block(&fin, WSIZE + 8192) && probe.n > WSIZE) {
              bit = (uint64_t)h * 8;
              ch.found = true;
              break;
            }
          }
        }
        const uint32_t hdr = ((uint32_t)data[by] | ((uint32_t)data[by + 1] << 8)) >> (p & 7);
        if ((hdr & 7u) != 4u) continue;  // BFINAL = 0, BTYPE = 2
        probe.reset(p);
        bool fin = false;
        if (!probe.block(&fin, WSIZE + 8192)) continue;
        if (probe.n < WSIZE + 1024) {
          // a short block: the next one must parse as well
          bool fin2 = false;
          if (fin || !probe.block(&fin2, WSIZE + 8192)) continue;
        }
        bit = p;
        ch.found = true;
      }
      if (!ch.found) return;
    }
    ch.start_bit = bit;
    Inflater inf(data, size, bit, ch.sym, false);
    const uint64_t stop_bit = c + 1 < chunks.size() ? chunks[c + 1]->nominal * 8 : ~0ULL;
    for (;;) {
      if (inf.br.bit_pos() >= stop_bit) break;  // the next chunk's worker searches from here on
      bool fin = false;
      if (!inf.block(&fin)) throw Panic("Error -- could not determine compression format / corrupt deflate data");
      if (fin) {
        // end of a member: trailer, then maybe another member
        inf.br.drop(inf.br.cnt & 7);
        const size_t tr = (size_t)(inf.br.bit_pos() >> 3);
        if (tr + 8 > size) throw Panic("Error -- truncated gzip file");
        uint32_t crc32v, isize;
        memcpy(&crc32v, data + tr, 4);
        memcpy(&isize, data + tr + 4, 4);
        ch.member_ends.push_back(inf.n - WSIZE);
        ch.member_crc.push_back(crc32v);
        ch.member_isize.push_back(isize);
        size_t nx = tr + 8;
        while (nx < size && data[nx] == 0) ++nx;  // padding between members
        if (nx >= size) {
          ch.ended_stream = true;
          ch.end_bit = (uint64_t)size * 8;
          inf.finish();
          return;
        }
        const size_t h = gzip_header(data, size, nx);
        if (!h) throw Panic("Error -- trailing garbage in gzip file");
        inf.br = BitReader(data, size, (uint64_t)h * 8);
      }
    }
    ch.end_bit = inf.br.bit_pos();
    inf.finish();
  }

  void work() {
    for (;;) {
      size_t c;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || (next_job.load() < chunks.size() && next_job.load() < consumed + in_flight_limit); });
        if (stop) return;
        c = next_job++;
      }
      std::string err;
      const auto t0 = std::chrono::steady_clock::now();
      try {
        run_chunk(c);
      } catch (const std::exception &e) {
        err = e.what();
        chunks[c]->found = false;
      }
      if (getenv("NIMBLE_GZIP_DEBUG"))
        fprintf(stderr, "[pgzip] chunk %zu: %.3f s, %zu symbols, found %d\n", c,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), chunks[c]->sym.size(),
                (int)chunks[c]->found);
      {
        std::lock_guard<std::mutex> lk(mu);
        chunks[c]->done = true;
      }
      cv.notify_all();
    }
  }
};

Reader::Reader(const std::string &path, unsigned threads) : impl_(new Impl()) {
  Impl &I = *impl_;
  I.fd = open(path.c_str(), O_RDONLY);
  if (I.fd < 0) throw Panic("Error -- could not determine compression format for " + path);
  struct stat st;
  if (fstat(I.fd, &st) != 0) throw Panic("Error -- could not determine compression format for " + path);
  I.size = (size_t)st.st_size;
  if (I.size) {
    void *m = mmap(nullptr, I.size, PROT_READ, MAP_PRIVATE, I.fd, 0);
    if (m == MAP_FAILED) throw Panic("Error -- could not determine compression format for " + path);
    I.data = (const uint8_t *)m;
    (void)madvise(m, I.size, MADV_SEQUENTIAL);
  }
  I.threads = std::max(1u, threads);
  // chunks small enough that every worker has several, large enough that finding a block start (a few hundred trial
  // decodes) stays a small part of the work
  I.chunk = std::min<size_t>(4u << 20, std::max<size_t>(256u << 10, I.size / ((size_t)I.threads * 4)));
  if (const char *e = getenv("NIMBLE_GZIP_CHUNK")) I.chunk = std::max<size_t>((size_t)strtoull(e, nullptr, 10), 1u << 16);
  const size_t n = std::max<size_t>(1, (I.size + I.chunk - 1) / I.chunk);
  for (size_t c = 0; c < n; ++c) {
    I.chunks.emplace_back(new Impl::Chunk());
    I.chunks.back()->nominal = c * I.chunk;
  }
  I.in_flight_limit = (size_t)I.threads + 2;
  I.window.assign(WSIZE, 0);
  for (unsigned t = 0; t < I.threads; ++t)
    if (!I.workers.spawn([this] { impl_->work(); })) break;  // (fewer inflate threads than asked for: slower, not wrong)
  if (I.workers.size() == 0) throw Panic("could not start a gzip reader thread (thread limit reached)");  // (~Impl cleans up)
}

Reader::~Reader() {}
void Reader::recycle(HugeBuf<uint16_t> &&sym) { impl_->give_back(std::move(sym)); }

// symbols -> bytes: a value below 256 is the byte itself, SYM0 + k is byte k of the window in front of the piece
// CRC-32 (the gzip polynomial) by carry-less multiplication: four 128-bit lanes folded 512 bits at a time, then reduced
// (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", Intel 2009; the constants are x^k mod P
// for the fold distances, bit-reflected).  zlib 1.2.11's table-driven crc32 runs near 1 GB/s per core, which made the
// checksum cost as much as the inflating it checks.  `crc` in and out are zlib's (not inverted) values.
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t crc0, const uint8_t *buf, size_t len) {
  // len >= 64 and a multiple of 16
  alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ULL, 0x01c6e41596ULL};  // fold by 512 bits
  alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ULL, 0x00ccaa009eULL};  // fold by 128 bits
  alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ULL, 0};                // 96 -> 64 bits
  alignas(16) static const uint64_t poly[2] = {0x01db710641ULL, 0x01f7011641ULL};  // P and the Barrett constant
  const __m128i *in = (const __m128i *)buf;
  __m128i a = _mm_loadu_si128(in), b = _mm_loadu_si128(in + 1), c = _mm_loadu_si128(in + 2), d = _mm_loadu_si128(in + 3);
  a = _mm_xor_si128(a, _mm_cvtsi32_si128((int)~crc0));
  __m128i k = _mm_load_si128((const __m128i *)k1k2);
  in += 4;
  len -= 64;
  while (len >= 64) {
    const __m128i al = _mm_clmulepi64_si128(a, k, 0x00), bl = _mm_clmulepi64_si128(b, k, 0x00);
    const __m128i cl = _mm_clmulepi64_si128(c, k, 0x00), dl = _mm_clmulepi64_si128(d, k, 0x00);
    a = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(a, k, 0x11), al), _mm_loadu_si128(in));
    b = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(b, k, 0x11), bl), _mm_loadu_si128(in + 1));
    c = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(c, k, 0x11), cl), _mm_loadu_si128(in + 2));
    d = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(d, k, 0x11), dl), _mm_loadu_si128(in + 3));
    in += 4;
    len -= 64;
  }
  k = _mm_load_si128((const __m128i *)k3k4);
  a = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(a, k, 0x11), _mm_clmulepi64_si128(a, k, 0x00)), b);
  a = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(a, k, 0x11), _mm_clmulepi64_si128(a, k, 0x00)), c);
  a = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(a, k, 0x11), _mm_clmulepi64_si128(a, k, 0x00)), d);
  while (len >= 16) {
    a = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(a, k, 0x11), _mm_clmulepi64_si128(a, k, 0x00)), _mm_loadu_si128(in));
    ++in;
    len -= 16;
  }
  // 128 -> 64 bits
  const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
  __m128i t = _mm_clmulepi64_si128(a, k, 0x10);
  a = _mm_xor_si128(_mm_srli_si128(a, 8), t);
  k = _mm_loadl_epi64((const __m128i *)k5k0);
  t = _mm_srli_si128(a, 4);
  a = _mm_and_si128(a, mask32);
  a = _mm_xor_si128(_mm_clmulepi64_si128(a, k, 0x00), t);
  // Barrett reduction to 32 bits
  k = _mm_load_si128((const __m128i *)poly);
  t = _mm_and_si128(a, mask32);
  t = _mm_clmulepi64_si128(t, k, 0x10);
  t = _mm_and_si128(t, mask32);
  t = _mm_clmulepi64_si128(t, k, 0x00);
  a = _mm_xor_si128(a, t);
  return ~(uint32_t)_mm_extract_epi32(a, 1);
}

uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len) {
  static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") &&
                           getenv("NIMBLE_NO_CLMUL") == nullptr;
  if (have && len >= 64) {
    const size_t body = len & ~(size_t)15;
    crc = crc32_clmul(crc, buf, body);
    buf += body;
    len -= body;
  }
  while (len) {
    const size_t step = std::min<size_t>(len, 1u << 30);
    crc = (uint32_t)crc32(crc, buf, (uInt)step);
    buf += step;
    len -= step;
  }
  return crc;
}

// symbols [from, to) of the piece as bytes into out[from .. to)
static void resolve_range(const Piece &p, size_t from, size_t to, uint8_t *out) {
  const uint16_t *s = p.sym.data() + WSIZE;
  const uint8_t *w = p.window.data();
  size_t i = from;
  const __m128i zero = _mm_setzero_si128();
  for (; i + 16 <= to; i += 16) {
    const __m128i lo = _mm_loadu_si128((const __m128i *)(s + i)), hi = _mm_loadu_si128((const __m128i *)(s + i + 8));
    const __m128i high_bytes = _mm_srli_epi16(_mm_or_si128(lo, hi), 8);
    if (_mm_movemask_epi8(_mm_cmpeq_epi8(high_bytes, zero)) == 0xFFFF) {  // sixteen plain bytes
      _mm_storeu_si128((__m128i *)(out + i), _mm_packus_epi16(lo, hi));
    } else {
      for (size_t k = i; k < i + 16; ++k) {
        const uint16_t v = s[k];
        out[k] = v < SYM0 ? (uint8_t)v : w[v - SYM0];
      }
    }
  }
  for (; i < to; ++i) {
    const uint16_t v = s[i];
    out[i] = v < SYM0 ? (uint8_t)v : w[v - SYM0];
  }
}

void resolve(const Piece &p, uint8_t *out) { resolve_range(p, 0, p.size(), out); }

// the same with the CRC-32 of out[from .. to) continued from `crc`, block by block while the bytes are still in the cache
uint32_t resolve_crc(const Piece &p, size_t from, size_t to, uint8_t *out, uint32_t crc) {
  const size_t BLOCK = 16u << 10;
  for (size_t q = from; q < to; q += BLOCK) {
    const size_t e = std::min(to, q + BLOCK);
    resolve_range(p, q, e, out);
    crc = crc32_fast(crc, out + q, e - q);
  }
  return crc;
}

// the next piece of the decompressed stream, in order (symbols unresolved, the window they refer to attached); false at
// the end
bool Reader::next(Piece &out) {
  Impl &I = *impl_;
  if (I.ready.empty() && !I.finished) I.join_step();
  if (I.ready.empty()) return false;
  out = std::move(I.ready.front());
  I.ready.pop_front();
  return true;
}

}  // namespace pgzip
}  // namespace parse
}  // namespace nimble
