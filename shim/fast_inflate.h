// fast_inflate.h -- a zlib-stream (RFC 1950 / 1951) decoder for the drop-in's probability PNG.
//
// The file hand-off of the reference (super4pcs_test.cc:58-80, base.cc:317) passes the per-pixel probabilities as a
// 16-bit PNG; inflating its 614 KB with zlib's inflate() was ~1 ms, the longest single step of a 1.8 ms call.  This
// decoder has the whole input and the whole output buffer in front of it, so it can refill its bit buffer eight
// bytes at a time, look a symbol up in one table probe (11 bits for literals / lengths, 8 for distances, second-level
// tables behind them; two literals from one probe where both codes fit in it) and copy matches in words.  It is
// RESUMABLE at any output position (run(limit) decodes until at least `limit` bytes exist -- up to five more), which is
// what lets the caller stop after the last image row it needs.
//
// Safety: every table probe and copy is bounded by the buffers given; a malformed stream returns false.  The caller
// checks the Adler-32 (zlib's adler32()) of a completely decoded stream against trailer() and falls back to zlib's own
// inflate on any failure (super4pcs_shim.cc).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace fastinf {

class Inflater {
 public:
  // `in` must stay valid; it is copied once with eight bytes of padding so that refills never read past its end
  Inflater(const unsigned char* in, size_t n_in, unsigned char* out, size_t n_out)
      : src_(in, in + n_in), out_(out), n_out_(n_out) {
    src_.resize(n_in + 16, 0);
    n_in_ = n_in;
  }

  // Decodes until at least `limit` output bytes exist (or the stream ends).  false = malformed / truncated stream
  // or output overflow.  May be called again with a larger limit.
  bool run(size_t limit) {
    if (limit > n_out_) limit = n_out_;
    if (!started_) {
      if (n_in_ < 6) return false;
      const unsigned cmf = src_[0], flg = src_[1];
      if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u || (flg & 32u)) return false;
      ip_ = 2;
      started_ = true;
    }
    while (opos_ < limit && !done_) {
      if (!in_block_) {
        if (!begin_block()) return false;
        continue;
      }
      if (btype_ == 0) {
        if (!stored(limit)) return false;
      } else {
        if (!huffman(limit, false)) return false;
      }
    }
    return true;
  }

  // Decodes to the end of the stream (no further output may appear beyond the buffer): needed for the trailer.
  bool finish() {
    if (!started_ && !run(0)) return false;
    while (!done_) {
      if (!in_block_) {
        if (!begin_block()) return false;
        continue;
      }
      if (btype_ == 0) {
        if (!stored(n_out_)) return false;
      } else {
        if (!huffman(n_out_, true)) return false;
      }
    }
    return true;
  }

  size_t produced() const { return opos_; }
  bool finished() const { return done_; }
  // the stream's own checksum (valid once finished()): big-endian Adler-32 after the last block
  bool trailer(uint32_t* adler) const {
    if (!done_) return false;
    size_t p = ip_ - (size_t)(bitcnt_ / 8);   // whole bytes still in the bit buffer were not consumed
    if (p + 4 > n_in_) return false;
    *adler = ((uint32_t)src_[p] << 24) | ((uint32_t)src_[p + 1] << 16) | ((uint32_t)src_[p + 2] << 8) | src_[p + 3];
    return true;
  }

 private:
  static constexpr int kLitBits = 11, kDistBits = 8;
  // table entry: bits 0..7 = code length to drop (0 = invalid), bit 8 = second level follows, bits 16.. = symbol or, for a
  // second-level pointer, (sub-table offset << 4 | sub-table bits)
  std::vector<unsigned char> src_;
  size_t n_in_ = 0, ip_ = 0;
  unsigned char* out_;
  size_t n_out_, opos_ = 0;
  uint64_t bitbuf_ = 0;
  int bitcnt_ = 0;
  bool started_ = false, done_ = false, in_block_ = false, last_ = false;
  int btype_ = 0;
  size_t stored_left_ = 0;
  std::vector<uint32_t> lit_, dist_;

  void refill() {   // at least 56 bits afterwards (the padding makes the 8-byte read safe)
    if (bitcnt_ <= 56 && ip_ + 8 <= src_.size()) {
      uint64_t w;
      std::memcpy(&w, src_.data() + ip_, 8);   // little-endian host (x86-64 / the GPU box)
      bitbuf_ |= w << bitcnt_;
      const int take = (63 - bitcnt_) >> 3;
      ip_ += (size_t)take;
      bitcnt_ += take * 8;
    }
  }
  uint32_t bits(int n) {   // n <= 32, after a refill
    const uint32_t v = (uint32_t)(bitbuf_ & ((n >= 32) ? 0xFFFFFFFFull : ((1ull << n) - 1ull)));
    bitbuf_ >>= n;
    bitcnt_ -= n;
    return v;
  }
  bool input_ok() const { return ip_ <= n_in_ + 8 && (ip_ - (size_t)(bitcnt_ > 0 ? bitcnt_ / 8 : 0)) <= n_in_; }

  static bool build(const unsigned char* len, int n, int root, std::vector<uint32_t>& tab) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[len[i]]++;
    count[0] = 0;
    int max_len = 15;
    while (max_len > 0 && count[max_len] == 0) --max_len;
    tab.assign((size_t)1 << root, 0u);
    if (max_len == 0) return true;   // no codes: every probe is invalid (an unused distance tree is legal)
    // over-subscription check (an incomplete set is accepted: zlib allows a single distance code)
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
      left = (left << 1) - count[l];
      if (left < 0) return false;
    }
    unsigned next_code[16] = {0};
    unsigned code = 0;
    for (int l = 1; l <= 15; ++l) {
      code = (code + (unsigned)count[l - 1]) << 1;
      next_code[l] = code;
    }
    // first pass: sizes of the second-level tables (one per distinct root-bit prefix of the long codes)
    std::vector<int> sub_bits((size_t)1 << root, 0);
    {
      unsigned nc[16];
      std::memcpy(nc, next_code, sizeof nc);
      for (int i = 0; i < n; ++i) {
        const int l = len[i];
        if (l <= root) { if (l) nc[l]++; continue; }
        const unsigned c = nc[l]++;
        unsigned rev = 0;
        for (int b = 0; b < l; ++b) rev |= ((c >> (l - 1 - b)) & 1u) << b;
        const unsigned prefix = rev & ((1u << root) - 1u);
        if (l - root > sub_bits[prefix]) sub_bits[prefix] = l - root;
      }
    }
    std::vector<uint32_t> sub_off((size_t)1 << root, 0u);
    for (size_t p = 0; p < sub_bits.size(); ++p) {
      if (!sub_bits[p]) continue;
      sub_off[p] = (uint32_t)tab.size();
      const uint32_t rel = sub_off[p] - (1u << root);   // the entry holds the offset behind the first level, in 12 bits
      if (rel >= 4096u) return false;   // (does not happen for 15-bit codes behind these roots)
      tab.resize(tab.size() + ((size_t)1 << sub_bits[p]), 0u);
      tab[p] = ((rel << 4 | (uint32_t)sub_bits[p]) << 16) | 0x100u | (uint32_t)root;
    }
    for (int i = 0; i < n; ++i) {
      const int l = len[i];
      if (!l) continue;
      const unsigned c = next_code[l]++;
      unsigned rev = 0;
      for (int b = 0; b < l; ++b) rev |= ((c >> (l - 1 - b)) & 1u) << b;
      if (l <= root) {
        const uint32_t e = ((uint32_t)i << 16) | (uint32_t)l;
        for (unsigned k = rev; k < (1u << root); k += 1u << l) tab[k] = e;
      } else {
        const unsigned prefix = rev & ((1u << root) - 1u);
        const int sb = sub_bits[prefix];
        const unsigned hi = rev >> root;
        const uint32_t e = ((uint32_t)i << 16) | (uint32_t)(l - root);
        for (unsigned k = hi; k < (1u << sb); k += 1u << (l - root)) tab[sub_off[prefix] + k] = e;
      }
    }
    return true;
  }

  bool begin_block() {
    refill();
    if (bitcnt_ < 3) return false;
    last_ = bits(1) != 0;
    btype_ = (int)bits(2);
    if (btype_ == 0) {
      bits(bitcnt_ & 7);   // to the byte boundary
      refill();
      if (bitcnt_ < 32) return false;
      const uint32_t len = bits(16), nlen = bits(16);
      if ((len ^ 0xFFFFu) != nlen) return false;
      stored_left_ = len;
    } else if (btype_ == 1) {
      unsigned char l[288];
      for (int i = 0; i < 144; ++i) l[i] = 8;
      for (int i = 144; i < 256; ++i) l[i] = 9;
      for (int i = 256; i < 280; ++i) l[i] = 7;
      for (int i = 280; i < 288; ++i) l[i] = 8;
      unsigned char d[30];
      for (int i = 0; i < 30; ++i) d[i] = 5;
      if (!build(l, 288, kLitBits, lit_) || !build(d, 30, kDistBits, dist_)) return false;
      mark_literals(lit_);
    } else if (btype_ == 2) {
      refill();
      if (bitcnt_ < 14) return false;
      const int hlit = (int)bits(5) + 257, hdist = (int)bits(5) + 1, hclen = (int)bits(4) + 4;
      if (hlit > 286 || hdist > 30) return false;
      static const unsigned char order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      unsigned char cl[19] = {0};
      for (int i = 0; i < hclen; ++i) {
        refill();
        if (bitcnt_ < 3) return false;
        cl[order[i]] = (unsigned char)bits(3);
      }
      std::vector<uint32_t> ct;
      if (!build(cl, 19, 7, ct)) return false;
      unsigned char l[286 + 30] = {0};
      int n = 0;
      while (n < hlit + hdist) {
        refill();
        if (bitcnt_ < 15 || !input_ok()) return false;
        const uint32_t e = ct[bitbuf_ & 127u];
        const int el = (int)(e & 0xFFu);
        if (el == 0 || (e & 0x100u)) return false;   // (code-length codes are at most 7 bits: no second level)
        bits(el);
        const int sym = (int)(e >> 16);
        if (sym < 16) {
          l[n++] = (unsigned char)sym;
        } else {
          int rep, val = 0;
          if (sym == 16) {
            if (n == 0) return false;
            val = l[n - 1];
            rep = 3 + (int)bits(2);
          } else if (sym == 17) {
            rep = 3 + (int)bits(3);
          } else {
            rep = 11 + (int)bits(7);
          }
          if (n + rep > hlit + hdist) return false;
          while (rep--) l[n++] = (unsigned char)val;
        }
      }
      if (l[256] == 0) return false;   // no end-of-block code
      if (!build(l, hlit, kLitBits, lit_) || !build(l + hlit, hdist, kDistBits, dist_)) return false;
      mark_literals(lit_);
    } else {
      return false;
    }
    in_block_ = true;
    return true;
  }

  bool stored(size_t limit) {
    // bytes still in the bit buffer come first (whole bytes: the block started on a byte boundary)
    while (stored_left_ && bitcnt_ >= 8 && opos_ < n_out_) {
      out_[opos_++] = (unsigned char)bits(8);
      --stored_left_;
    }
    if (stored_left_ && bitcnt_ >= 8) return false;   // output full with data left
    if (stored_left_) {
      // the bit buffer is empty (< 8 bits, and they are padding of the NEXT refill: none were consumed past a byte)
      ip_ -= (size_t)(bitcnt_ / 8);
      bitbuf_ = 0;
      bitcnt_ = 0;
      size_t n = stored_left_;
      if (opos_ + n > n_out_ || ip_ + n > n_in_) return false;
      const size_t want = limit > opos_ ? limit - opos_ : 0;
      if (n > want && want > 0) n = want;   // stop at the limit: the rest of the block comes with the next run()
      std::memcpy(out_ + opos_, src_.data() + ip_, n);
      opos_ += n;
      ip_ += n;
      stored_left_ -= n;
    }
    if (!stored_left_) end_block();
    return true;
  }

  void end_block() {
    in_block_ = false;
    if (last_) done_ = true;
  }

  // Entries of the literal / length table whose symbol is a plain literal get bit 9: the hot loop tests one bit.  Where the
  // bits behind such a literal decide a SECOND literal inside the same first-level probe (both codes together at most
  // kLitBits long: the short codes of the high bytes next to the long ones of the low bytes, in a Sub-filtered 16-bit
  // image), the entry carries both -- bit 10, the two literals in bits 16..31, the total length in bits 0..7 and the first
  // literal's own length in bits 11..14 for the places that take one symbol at a time.
  static void mark_literals(std::vector<uint32_t>& tab) {
    for (size_t k = 0; k < tab.size(); ++k) {
      const uint32_t e = tab[k];
      if (!(e & 0x100u) && (e & 0xFFu) != 0u && (e >> 16) < 256u) tab[k] = e | 0x200u;
    }
    const uint32_t root = 1u << kLitBits;
    std::vector<uint32_t> two(root, 0u);
    for (uint32_t k = 0; k < root; ++k) {
      const uint32_t e1 = tab[k];
      if (!(e1 & 0x200u)) continue;
      const uint32_t l1 = e1 & 0xFFu;
      if (l1 >= (uint32_t)kLitBits) continue;
      const uint32_t e2 = tab[k >> l1];   // the bits behind the first code, zeros above them
      if (!(e2 & 0x200u)) continue;
      const uint32_t l2 = e2 & 0xFFu;
      if (l1 + l2 > (uint32_t)kLitBits) continue;   // else the second code depends on bits this probe does not hold
      two[k] = ((e2 >> 16) << 24) | ((e1 >> 16) << 16) | (l1 << 11) | 0x400u | 0x200u | (l1 + l2);
    }
    for (uint32_t k = 0; k < root; ++k)
      if (two[k]) tab[k] = two[k];
  }

  // The decoder's state lives in locals for the length of the call (as members every store to the output -- an unsigned
  // char -- forces the bit buffer, its count and the input position back to memory and in again: ~8 ns per literal, three
  // times zlib's rate lost on an image that does not compress), and up to three first-level literals follow one refill.
  // Input exhaustion (bits taken from the zero padding) is checked where it matters -- at every match, before an end-of-
  // block is believed, and on the way out -- not per literal: what is decoded from padding stays inside the output
  // buffer and the call still fails.
  bool huffman(size_t limit, bool to_end) {
    static const unsigned short len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const unsigned char len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const unsigned short dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const unsigned char dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const uint32_t* const lit = lit_.data();
    const uint32_t* const dst = dist_.data();
    unsigned char* const out = out_;
    const unsigned char* const src = src_.data();
    const size_t n_out = n_out_, src_size = src_.size(), n_in = n_in_;
    const uint64_t lit_mask = (1u << kLitBits) - 1u;
    size_t op = opos_, ip = ip_;
    uint64_t bb = bitbuf_;
    int bc = bitcnt_;
    bool ok = true, block_done = false;
    auto input_ok = [&] { return ip <= n_in + 8 && (ip - (size_t)(bc > 0 ? bc / 8 : 0)) <= n_in; };
    while (op < limit || to_end) {
      // >= 56 bits: a length code (15) + its extra bits (5) + a distance code (15) + its extra bits (13) = 48
      if (bc <= 56 && ip + 8 <= src_size) {
        uint64_t w;
        std::memcpy(&w, src + ip, 8);   // little-endian host (x86-64 / the GPU box)
        bb |= w << bc;
        const int take = (63 - bc) >> 3;
        ip += (size_t)take;
        bc += take * 8;
      }
      uint32_t e = lit[bb & lit_mask];
      if ((e & 0x200u) && op + 8 <= n_out) {   // up to three probes of at most 11 bits, one or two literals each
        bb >>= (e & 0xFFu);
        bc -= (int)(e & 0xFFu);
        out[op] = (unsigned char)(e >> 16);
        out[op + 1] = (unsigned char)(e >> 24);   // (written either way: overwritten by what follows when the entry holds one)
        op += 1 + ((e >> 10) & 1u);
        e = lit[bb & lit_mask];
        if (e & 0x200u) {
          bb >>= (e & 0xFFu);
          bc -= (int)(e & 0xFFu);
          out[op] = (unsigned char)(e >> 16);
          out[op + 1] = (unsigned char)(e >> 24);
          op += 1 + ((e >> 10) & 1u);
          e = lit[bb & lit_mask];
          if (e & 0x200u) {
            bb >>= (e & 0xFFu);
            bc -= (int)(e & 0xFFu);
            out[op] = (unsigned char)(e >> 16);
            out[op + 1] = (unsigned char)(e >> 24);
            op += 1 + ((e >> 10) & 1u);
          }
        }
        continue;
      }
      if (e & 0x400u) e = (e & 0x00FF0000u) | 0x200u | ((e >> 11) & 0xFu);   // near the end of the buffer: its first literal alone
      if (e & 0x100u) {
        const uint32_t sub = e >> 16;
        bb >>= kLitBits;
        bc -= kLitBits;
        e = lit[(1u << kLitBits) + (sub >> 4) + (uint32_t)(bb & ((1u << (sub & 15u)) - 1u))];
        if (e & 0x100u) { ok = false; break; }
      }
      const int el = (int)(e & 0xFFu);
      if (el == 0) { ok = false; break; }
      bb >>= el;
      bc -= el;
      const uint32_t sym = e >> 16;
      if (sym < 256) {
        if (op >= n_out) { ok = false; break; }
        out[op++] = (unsigned char)sym;
        continue;
      }
      if (sym == 256) {
        block_done = true;
        break;
      }
      if (sym > 285) { ok = false; break; }
      const uint32_t li = sym - 257;
      size_t len = len_base[li] + (size_t)(bb & ((1u << len_extra[li]) - 1u));
      bb >>= len_extra[li];
      bc -= len_extra[li];
      uint32_t d = dst[bb & ((1u << kDistBits) - 1u)];
      if (d & 0x100u) {
        const uint32_t sub = d >> 16;
        bb >>= kDistBits;
        bc -= kDistBits;
        d = dst[(1u << kDistBits) + (sub >> 4) + (uint32_t)(bb & ((1u << (sub & 15u)) - 1u))];
        if (d & 0x100u) { ok = false; break; }
      }
      const int dl = (int)(d & 0xFFu);
      if (dl == 0) { ok = false; break; }
      bb >>= dl;
      bc -= dl;
      const uint32_t ds = d >> 16;
      if (ds > 29) { ok = false; break; }
      const size_t dist = dist_base[ds] + (size_t)(bb & ((1u << dist_extra[ds]) - 1u));
      bb >>= dist_extra[ds];
      bc -= dist_extra[ds];
      if (bc < 0 || !input_ok()) { ok = false; break; }   // ran past the input
      if (dist > op || op + len > n_out) { ok = false; break; }
      unsigned char* to = out + op;
      const unsigned char* from = to - dist;
      op += len;
      if (dist >= 8 && op + 8 <= n_out) {
        // word copies; the overshoot of up to seven bytes stays inside the buffer and is overwritten by what follows
        for (size_t k = 0; k < len; k += 8) {
          uint64_t w;
          std::memcpy(&w, from + k, 8);
          std::memcpy(to + k, &w, 8);
        }
      } else if ((dist == 1 || dist == 2 || dist == 4) && op + 8 <= n_out) {
        // a run of one byte (all a Z_RLE encoder, cv::imwrite's default strategy, ever emits), of one 16-bit or one 32-bit
        // value -- most of a probability image: the pattern in words (8 % dist == 0)
        uint64_t w;
        if (dist == 1) {
          w = (uint64_t)*from * 0x0101010101010101ull;
        } else if (dist == 2) {
          uint16_t v;
          std::memcpy(&v, from, 2);
          w = (uint64_t)v * 0x0001000100010001ull;
        } else {
          uint32_t v;
          std::memcpy(&v, from, 4);
          w = (uint64_t)v * 0x0000000100000001ull;
        }
        for (size_t k = 0; k < len; k += 8) std::memcpy(to + k, &w, 8);
      } else {
        for (size_t k = 0; k < len; ++k) to[k] = from[k];
      }
    }
    bitbuf_ = bb;
    bitcnt_ = bc;
    ip_ = ip;
    opos_ = op;
    if (!ok || !input_ok()) return false;
    if (block_done) end_block();
    return true;
  }
};

}  // namespace fastinf
