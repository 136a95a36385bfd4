// shim/test_parsers.cc -- the drop-in's file readers (file_readers.h, fast_inflate.h) under AddressSanitizer + UBSan.
//
//   make -C shim asan           builds shim/test_parsers_asan (g++ -fsanitize=address,undefined; CPU only, no GPU, no Eigen)
//   test_parsers_asan ply  <valid.ply> <n> <seed> <scratch file>     n seeded mutations of the file through read_ply
//   test_parsers_asan png  <valid.png> <n> <seed> <scratch file>     ... through read_png_gray
//   test_parsers_asan zlib <valid.png> <n> <seed> -                  ... of its IDAT stream through fastinf::Inflater,
//                                                                    odd output capacities, resumable steps, against zlib
//   test_parsers_asan file <kind> <path>                             one file as it is (the crafted headers)
//
// Every mutated file must be either refused (false) or read into containers consistent with what the reader reports;
// the sanitizers see every access, and the test runs with ASAN_OPTIONS=max_allocation_size_mb=256 so that a header
// that asks for more memory than a damaged 100 KB file can justify aborts the run (tests/test_parsers_fuzz.py).
// The reference exit(-1)s on files it cannot read (super4pcs_test.cc:58-80); the drop-in answers identity / score 0.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "file_readers.h"

typedef std::vector<unsigned char> Bytes;

static Bytes slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return Bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static bool spill(const char* path, const Bytes& b) {
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  f.write(reinterpret_cast<const char*>(b.data()), (std::streamsize)b.size());
  return (bool)f;
}

// header-aware mutations on top of plain byte damage: numbers in the first KB replaced by huge / negative / odd ones
static void mutate(Bytes& b, std::mt19937& g, bool text_header) {
  if (b.empty()) return;
  std::uniform_int_distribution<size_t> pos(0, b.size() - 1), hpos(0, std::min<size_t>(b.size(), 400) - 1);
  const int kind = (int)(g() % 10);
  const int n = 1 + (int)(g() % 4);
  for (int k = 0; k < n; ++k) {
    switch (kind) {
      case 0: b[pos(g)] ^= (unsigned char)(1u << (g() % 8)); break;
      case 1: b[hpos(g)] = (unsigned char)g(); break;
      case 2: b.resize(pos(g)); if (b.empty()) return; pos = std::uniform_int_distribution<size_t>(0, b.size() - 1); hpos = std::uniform_int_distribution<size_t>(0, std::min<size_t>(b.size(), 400) - 1); break;
      case 3: { const size_t p = pos(g), m = std::min<size_t>(b.size() - p, 1 + g() % 64); b.erase(b.begin() + (long)p, b.begin() + (long)(p + m)); if (b.empty()) return; pos = std::uniform_int_distribution<size_t>(0, b.size() - 1); hpos = std::uniform_int_distribution<size_t>(0, std::min<size_t>(b.size(), 400) - 1); break; }
      case 4: { const size_t p = pos(g); b.insert(b.begin() + (long)p, (size_t)(1 + g() % 64), (unsigned char)g()); break; }
      case 5:
        if (text_header) {   // a digit run of the header becomes a number the file cannot back
          static const char* big[] = {"2000000000", "99999999999999", "-5", "0", "4294967297", "1e9", "2147483648"};
          size_t p = hpos(g);
          while (p < b.size() && p < 400 && !(b[p] >= '0' && b[p] <= '9')) ++p;
          size_t q = p;
          while (q < b.size() && b[q] >= '0' && b[q] <= '9') ++q;
          if (q > p) {
            const char* s = big[g() % 7];
            b.erase(b.begin() + (long)p, b.begin() + (long)q);
            b.insert(b.begin() + (long)p, s, s + std::strlen(s));
          }
        } else {             // four bytes of the first chunks become a big-endian size
          const size_t p = 8 + g() % 40;
          if (p + 4 <= b.size()) { const uint32_t v = g() % 3 ? (uint32_t)g() : 0x7FFFFFFFu; b[p] = (unsigned char)(v >> 24); b[p + 1] = (unsigned char)(v >> 16); b[p + 2] = (unsigned char)(v >> 8); b[p + 3] = (unsigned char)v; }
        }
        break;
      case 6: { const size_t p = pos(g), m = std::min<size_t>(b.size() - p, 1 + g() % 256); Bytes c(b.begin() + (long)p, b.begin() + (long)(p + m)); b.insert(b.begin() + (long)pos(g), c.begin(), c.end()); break; }
      case 7: for (int j = 0; j < 16; ++j) b[pos(g)] = (unsigned char)g(); break;
      case 8: b[pos(g)] = 0; break;
      default: b[pos(g)] = 0xFF; break;
    }
  }
}

static int check_ply(const char* path) {
  shimio::Cloud c;
  if (!shimio::read_ply(path, c)) return 0;
  if (c.n < 0 || c.xyz.size() != (size_t)c.n * 3 || c.nrm.size() != (size_t)c.n * 3) {
    std::printf("FAIL read_ply: n %d with %zu / %zu floats\n", c.n, c.xyz.size(), c.nrm.size());
    return -1;
  }
  return 1;
}
static int check_png(const char* path) {
  std::vector<uint16_t> px;
  int rows = 0, cols = 0;
  if (!shimio::read_png_gray(path, px, rows, cols)) return 0;
  if (rows <= 0 || cols <= 0 || px.size() != (size_t)rows * cols) {
    std::printf("FAIL read_png_gray: %d x %d with %zu pixels\n", rows, cols, px.size());
    return -1;
  }
  return 1;
}

// the IDAT payload of a PNG (its chunks concatenated)
static Bytes idat_of(const Bytes& f) {
  Bytes z;
  size_t pos = 8;
  while (pos + 12 <= f.size()) {
    const uint32_t len = ((uint32_t)f[pos] << 24) | ((uint32_t)f[pos + 1] << 16) | ((uint32_t)f[pos + 2] << 8) | f[pos + 3];
    if (pos + 12 + (size_t)len > f.size()) break;
    if (std::memcmp(&f[pos + 4], "IDAT", 4) == 0) z.insert(z.end(), f.begin() + (long)pos + 8, f.begin() + (long)(pos + 8 + len));
    pos += 12 + (size_t)len;
  }
  return z;
}

// one stream through the decoder in resumable steps with capacity `cap`; whatever zlib makes of the same stream with the
// same capacity is the truth: where zlib succeeds completely and the decoder says it finished, the bytes must agree
static int check_zlib(const Bytes& z, size_t cap, std::mt19937& g) {
  Bytes mine(cap + 1, 0xA5), ref(cap + 1, 0x5A);
  fastinf::Inflater inf(z.data(), z.size(), mine.data(), cap);
  bool ok = true;
  size_t limit = 0;
  while (ok && limit < cap) {
    limit = std::min(cap, limit + 1 + g() % (cap / 4 + 1));
    ok = inf.run(limit);
    if (ok && inf.produced() < limit && !inf.finished()) ok = false;
    if (inf.finished()) break;
  }
  if (ok) ok = inf.finish();
  if (mine[cap] != 0xA5) {
    std::printf("FAIL inflater wrote past its buffer (cap %zu)\n", cap);
    return -1;
  }
  if (inf.produced() > cap) {
    std::printf("FAIL inflater reports %zu of %zu bytes\n", inf.produced(), cap);
    return -1;
  }
  z_stream zs;
  std::memset(&zs, 0, sizeof zs);
  if (inflateInit(&zs) != Z_OK) return -1;
  zs.next_in = const_cast<unsigned char*>(z.data());
  zs.avail_in = (uInt)z.size();
  zs.next_out = ref.data();
  zs.avail_out = (uInt)cap;
  const int rc = inflate(&zs, Z_FINISH);
  const size_t n_ref = cap - zs.avail_out;
  inflateEnd(&zs);
  if (ok && rc == Z_STREAM_END) {
    uint32_t want = 0;
    const bool sum_ok = inf.trailer(&want) && (uint32_t)adler32(adler32(0L, Z_NULL, 0), mine.data(), (uInt)inf.produced()) == want;
    // (a stream whose checksum is damaged is refused by zlib -- Z_DATA_ERROR, not reached here -- and by the caller of the
    // decoder, which compares trailer() itself)
    if (inf.produced() != n_ref || std::memcmp(mine.data(), ref.data(), n_ref) != 0) {
      std::printf("FAIL inflater: %zu bytes, zlib %zu, contents %s\n", inf.produced(), n_ref, inf.produced() == n_ref ? "differ" : "-");
      return -1;
    }
    return sum_ok ? 1 : 0;
  }
  return 0;
}

// adler32_of (the 32-bytes-at-a-time form where the CPU has AVX2) against zlib's own on seeded random buffers: every length
// 0 .. 300, lengths around the block size of the vector loop and around zlib's NMAX, all-0xFF buffers (the largest sums)
static int check_adler(unsigned seed) {
  std::mt19937 g(seed);
  std::vector<size_t> lens;
  for (size_t n = 0; n <= 300; ++n) lens.push_back(n);
  for (size_t c : {(size_t)5520, (size_t)5536, (size_t)5552, (size_t)11072, (size_t)65521, (size_t)614880, (size_t)1 << 20})
    for (int d = -33; d <= 33; d += 11) lens.push_back(c + d);
  int bad = 0;
  for (size_t n : lens) {
    for (int fill = 0; fill < 2; ++fill) {
      Bytes b(n + 3);
      for (auto& x : b) x = fill ? 0xFF : (unsigned char)(g() & 0xFF);
      const unsigned char* p = b.data() + (n % 3);   // (unaligned starts too)
      const uint32_t want = (uint32_t)adler32(adler32(0L, Z_NULL, 0), p, (uInt)n);
      if (shimio::adler32_of(p, n) != want) {
        std::fprintf(stderr, "adler32_of differs at length %zu (fill %d)\n", n, fill);
        ++bad;
      }
    }
  }
  std::printf("adler %zu lengths checked, %d differ\n", lens.size() * 2, bad);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc == 3 && std::strcmp(argv[1], "adler") == 0) return check_adler((unsigned)std::atoi(argv[2]));
  if (argc == 4 && std::strcmp(argv[1], "file") == 0) {
    const int r = std::strcmp(argv[2], "ply") == 0 ? check_ply(argv[3]) : check_png(argv[3]);
    std::printf("%s %s\n", argv[3], r > 0 ? "READ" : (r == 0 ? "REFUSED" : "BROKEN"));
    return r < 0 ? 1 : 0;
  }
  if (argc != 6) {
    std::fprintf(stderr, "usage: %s ply|png|zlib <valid file> <n> <seed> <scratch> | file ply|png <path>\n", argv[0]);
    return 2;
  }
  const std::string kind = argv[1];
  const Bytes base = slurp(argv[2]);
  const int n = std::atoi(argv[3]);
  std::mt19937 g((unsigned)std::atoi(argv[4]));
  const char* scratch = argv[5];
  if (base.empty()) {
    std::fprintf(stderr, "cannot read %s\n", argv[2]);
    return 2;
  }
  int read = 0, refused = 0;
  if (kind == "zlib") {
    const Bytes z0 = idat_of(base);
    // the untouched stream first: round trip exact at its true size and refused / truncated at odd ones
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    Bytes full(64u << 20);
    inflateInit(&zs);
    zs.next_in = const_cast<unsigned char*>(z0.data());
    zs.avail_in = (uInt)z0.size();
    zs.next_out = full.data();
    zs.avail_out = (uInt)full.size();
    if (inflate(&zs, Z_FINISH) != Z_STREAM_END) return 2;
    const size_t true_size = full.size() - zs.avail_out;
    inflateEnd(&zs);
    if (check_zlib(z0, true_size, g) != 1) {
      std::printf("FAIL the valid stream does not round-trip\n");
      return 1;
    }
    for (int i = 0; i < n; ++i) {
      Bytes z = z0;
      if (i % 3) mutate(z, g, false);
      const size_t cap = (i % 5 == 0) ? true_size : 1 + g() % (true_size + true_size / 8);
      const int r = check_zlib(z, cap, g);
      if (r < 0) return 1;
      (r ? read : refused)++;
    }
  } else {
    const bool ply = kind == "ply";
    for (int i = 0; i < n; ++i) {
      Bytes b = base;
      mutate(b, g, ply);
      if (!spill(scratch, b)) return 2;
      const int r = ply ? check_ply(scratch) : check_png(scratch);
      if (r < 0) return 1;
      (r ? read : refused)++;
    }
  }
  std::printf("%s: %d mutations, %d read, %d refused\nOK\n", kind.c_str(), n, read, refused);
  return 0;
}
