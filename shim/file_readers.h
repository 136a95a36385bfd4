// shim/file_readers.h -- the file hand-off of the drop-in (S4/super4pcs_test.cc:58-80, base.cc:317): PLY clouds and the
// 16-bit probability PNG, read the way the reference reads them.  Header-only and free of Eigen / libpgp so that the
// parsers can be built on their own under AddressSanitizer + UBSan and fed damaged files (shim/test_parsers.cc,
// `make -C shim asan`, tests/test_parsers_fuzz.py).
//
// Failure mode: every reader returns false on a file it cannot make sense of -- the drop-in then answers with the identity
// pose and score 0 where the reference calls exit(-1) (super4pcs_test.cc:58-80).  A reader never trusts a count of the
// header beyond what the file can hold: the vertex count of a PLY is bounded by the bytes that follow the header, the
// pixel count of a PNG by what its compressed data can inflate to (DEFLATE expands at most 1032 : 1), so a damaged header
// cannot make the ROS node allocate more than a small multiple of the file's own size.
#pragma once

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include <zlib.h>

#include <cerrno>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <mutex>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "fast_inflate.h"

namespace shimio {

struct Cloud {
  std::vector<float> xyz, nrm;  // n x 3 each (nrm zero when the file has none)
  int n = 0;
};

// ---------------------------------------------------------------------------------------------
// PLY: header-driven reader for what pcl::io::savePLYFile writes (ASCII by default, also
// binary_little_endian): vertex properties x y z [nx ny nz | normal_x normal_y normal_z] plus
// anything else, which is skipped (S4/io/io_ply.h reads the same columns, :270-277,311-317).
// ASCII files give the reference reader's positions and normals bit for bit
// (tests/test_ply_reader.py, against S4/io/io.cc compiled unmodified).  binary_little_endian is an
// extension: the reference's binary reader assumes all-float vertex records and misreads the files
// PCL writes with uchar colours (checked against the same build), so the node's ASCII default is
// the only form the two can be compared on.
// ---------------------------------------------------------------------------------------------
struct PlyProp {
  std::string name, type;
  int size = 0;
};

inline int ply_type_size(const std::string& t) {
  if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
  if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
  if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
  if (t == "double" || t == "float64") return 8;
  return 0;
}

inline double ply_read_bin(const unsigned char* p, const std::string& t) {
  if (t == "float" || t == "float32") { float v; std::memcpy(&v, p, 4); return v; }
  if (t == "double" || t == "float64") { double v; std::memcpy(&v, p, 8); return v; }
  if (t == "uchar" || t == "uint8") return *p;
  if (t == "char" || t == "int8") return *reinterpret_cast<const signed char*>(p);
  if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, p, 2); return v; }
  if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, p, 2); return v; }
  if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, p, 4); return v; }
  if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, p, 4); return v; }
  return 0;
}

// One decimal number of an ASCII vertex line as the FLOAT the reference reads: fscanf("%f", &float)
// (S4/io/io_ply.h:272-296) converts decimal -> float with ONE rounding (strtof).  Fast path (Clinger): at
// most 15 significant digits and a decimal exponent within +-22 make mantissa and power of ten both exact
// doubles, so one multiplication or division gives the correctly rounded DOUBLE; casting that to float is a
// second rounding and agrees with strtof unless the double sits on (or within one double-ulp of) the midpoint
// of two floats -- 29 discarded mantissa bits reading 0x0FFFFFFF .. 0x10000001 -- or outside the normal
// float range; those, long mantissas, nan / inf and hex floats go to strtof itself.
inline bool ascii_number(const char** pp, float* out) {
  static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char* p = *pp;
  while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v') ++p;
  const char* start = p;
  bool neg = false;
  if (*p == '+' || *p == '-') neg = *p++ == '-';
  {
    // the plain form first -- digits, an optional point, more digits, then white space: what "%.9g" writes for every
    // coordinate of a cloud in metres.  At most 19 digits cannot overflow 64 bits; below 10^15 the value is the same
    // exact mantissa the general path below would collect (leading zeros add nothing), and the same one division follows.
    const char* q = p;
    unsigned long long m = 0;
    while ((unsigned)(*q - '0') < 10u) m = m * 10 + (unsigned)(*q++ - '0');
    const char* const int_end = q;
    int frac = 0;
    if (*q == '.') {
      ++q;
      const char* const f0 = q;
      while ((unsigned)(*q - '0') < 10u) m = m * 10 + (unsigned)(*q++ - '0');
      frac = (int)(q - f0);
    }
    const int n_digits = (int)(int_end - p) + frac;
    if (n_digits >= 1 && n_digits <= 19 && frac <= 22 && m < 1000000000000000ull &&
        (*q == ' ' || *q == '\n' || *q == '\r' || *q == '\t' || *q == '\0')) {
      double d = (double)m;
      if (frac) d /= kPow10[frac];
      uint64_t bits;
      std::memcpy(&bits, &d, 8);
      const uint32_t low = (uint32_t)(bits & 0x1FFFFFFFull);
      const int e2 = (int)((bits >> 52) & 0x7FF) - 1023;
      if (m == 0 || (!(low >= 0x0FFFFFFFu && low <= 0x10000001u) && e2 >= -126 && e2 <= 126)) {
        *out = neg ? -(float)d : (float)d;
        *pp = q;
        return true;
      }
    }
  }
  unsigned long long mant = 0;
  int digits = 0, exp10 = 0;
  bool any = false, fast = true;
  while (*p >= '0' && *p <= '9') {
    any = true;
    if (mant || *p != '0') {
      if (digits < 15) { mant = mant * 10 + (unsigned)(*p - '0'); ++digits; }
      else fast = false;
    }
    ++p;
  }
  if (*p == '.') {
    ++p;
    while (*p >= '0' && *p <= '9') {
      any = true;
      if (mant || *p != '0') {
        if (digits < 15) { mant = mant * 10 + (unsigned)(*p - '0'); ++digits; --exp10; }
        else fast = false;
      } else {
        --exp10;   // a leading zero of the fraction
      }
      ++p;
    }
  }
  if (any && (*p == 'e' || *p == 'E')) {
    const char* q = p + 1;
    bool eneg = false;
    if (*q == '+' || *q == '-') eneg = *q++ == '-';
    if (*q >= '0' && *q <= '9') {
      int e = 0;
      while (*q >= '0' && *q <= '9') {
        if (e < 10000) e = e * 10 + (*q - '0');
        ++q;
      }
      exp10 += eneg ? -e : e;
      p = q;
    }
  }
  const bool end_ok = *p == '\0' || *p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v';
  if (any && fast && end_ok && exp10 >= -22 && exp10 <= 22) {
    double d = (double)mant;   // < 10^15 < 2^53: exact
    d = exp10 < 0 ? d / kPow10[-exp10] : d * kPow10[exp10];
    uint64_t bits;
    std::memcpy(&bits, &d, 8);
    const uint32_t low = (uint32_t)(bits & 0x1FFFFFFFull);          // the mantissa bits a float drops
    const int e2 = (int)((bits >> 52) & 0x7FF) - 1023;
    const bool midpoint = low >= 0x0FFFFFFFu && low <= 0x10000001u;
    if (mant == 0 || (!midpoint && e2 >= -126 && e2 <= 126)) {
      *out = neg ? -(float)d : (float)d;
      *pp = p;
      return true;
    }
  }
  char* end = nullptr;
  const float f = std::strtof(start, &end);
  if (end == start) return false;
  *out = f;
  *pp = end;
  return true;
}

// A whole file in memory, NUL-terminated, in one of a few buffers the process keeps between calls (a fresh zero-filled
// buffer of a 180 KB cloud is an mmap, its page faults and a munmap on every call): a reader takes whichever is free and
// falls back to a buffer of its own when all are held.  Buffers beyond kKeepBytes are not kept.
struct FileBytes {
  static const size_t kSlots = 4, kKeepBytes = 16u << 20;
  struct Slot {
    std::mutex mu;
    char* buf = nullptr;
    size_t cap = 0;
  };
  static Slot* slots() {
    static Slot s[kSlots];
    return s;
  }
  Slot* slot = nullptr;
  char* own = nullptr;
  char* data = nullptr;
  size_t len = 0;
  FileBytes() {}
  FileBytes(const FileBytes&) = delete;
  FileBytes& operator=(const FileBytes&) = delete;
  ~FileBytes() {
    if (slot) {
      if (slot->cap > kKeepBytes) { std::free(slot->buf); slot->buf = nullptr; slot->cap = 0; }
      slot->mu.unlock();
    }
    std::free(own);
  }
  char* room(size_t need) {
    if (!slot && !own) {
      Slot* s = slots();
      for (size_t k = 0; k < kSlots; ++k)
        if (s[k].mu.try_lock()) { slot = &s[k]; break; }
    }
    if (slot) {
      if (slot->cap < need) {
        char* nb = static_cast<char*>(std::realloc(slot->buf, need));
        if (!nb) return nullptr;
        slot->buf = nb;
        slot->cap = need;
      }
      return slot->buf;
    }
    char* nb = static_cast<char*>(std::realloc(own, need));
    if (nb) own = nb;
    return nb;
  }
  bool read(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    size_t cap = 1 << 16;
    if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) cap = (size_t)st.st_size + 1;
    len = 0;
    bool ok = true;
    for (;;) {
      data = room(cap + 1);
      if (!data) { ok = false; break; }
      const ssize_t got = ::read(fd, data + len, cap - len);
      if (got < 0) {
        if (errno == EINTR) continue;
        ok = false;
        break;
      }
      if (got == 0) break;
      len += (size_t)got;
      if (len == cap) cap *= 2;   // (longer than fstat said, or not a regular file)
    }
    ::close(fd);
    if (ok) data[len] = '\0';
    return ok;
  }
};

// the next white-space separated token of a header line [p, e)
inline std::string ply_token(const char*& p, const char* e) {
  while (p < e && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\f' || *p == '\v')) ++p;
  const char* b = p;
  while (p < e && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\f' || *p == '\v')) ++p;
  return std::string(b, p);
}

inline bool read_ply(const std::string& path, Cloud& out) {
  FileBytes file;
  if (!file.read(path)) return false;
  const char* p = file.data;
  const char* const end = file.data + file.len;
  if (file.len < 3 || std::memcmp(p, "ply", 3) != 0) return false;
  bool ascii = true, in_vertex = false, header_done = false;
  long n_vertex = 0;
  std::vector<PlyProp> props;
  {
    const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(end - p)));
    if (!nl) return false;   // (a header needs its end_header line)
    p = nl + 1;
  }
  while (p < end) {
    const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(end - p)));
    const char* e = nl ? nl : end;
    const char* q = p;
    p = nl ? nl + 1 : end;
    const std::string tok = ply_token(q, e);
    if (tok == "format") {
      const std::string f = ply_token(q, e);
      if (f == "ascii") ascii = true;
      else if (f == "binary_little_endian") ascii = false;
      else return false;
    } else if (tok == "element") {
      const std::string name = ply_token(q, e);
      const std::string cnt_s = ply_token(q, e);
      char* ce = nullptr;
      errno = 0;
      const long cnt = std::strtol(cnt_s.c_str(), &ce, 10);
      if (cnt_s.empty() || ce == cnt_s.c_str() || errno == ERANGE || cnt < 0) return false;
      in_vertex = name == "vertex";
      if (in_vertex) n_vertex = cnt;
    } else if (tok == "property" && in_vertex) {
      PlyProp pr;
      pr.type = ply_token(q, e);
      if (pr.type == "list") return false;
      pr.name = ply_token(q, e);
      pr.size = ply_type_size(pr.type);
      if (!pr.size) return false;
      props.push_back(pr);
    } else if (tok == "end_header") {
      header_done = true;
      break;
    }
  }
  int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1;
  for (size_t k = 0; k < props.size(); ++k) {
    const std::string& nm = props[k].name;
    if (nm == "x") ix = (int)k;
    else if (nm == "y") iy = (int)k;
    else if (nm == "z") iz = (int)k;
    else if (nm == "nx" || nm == "normal_x") inx = (int)k;
    else if (nm == "ny" || nm == "normal_y") iny = (int)k;
    else if (nm == "nz" || nm == "normal_z") inz = (int)k;
  }
  if (ix < 0 || iy < 0 || iz < 0 || n_vertex < 0 || n_vertex > INT_MAX || !header_done) return false;
  // `element vertex N` is only a claim: N rows need N x (bytes per row) of file behind the header -- the record size of
  // a binary file, at least one character and one separator per property of an ASCII one
  const size_t left = (size_t)(end - p);
  {
    size_t row = 0;
    for (size_t k = 0; k < props.size(); ++k) row += ascii ? 2 : (size_t)props[k].size;
    if (row == 0 || (unsigned long long)n_vertex > (unsigned long long)left / row + 1) return false;
  }
  out.n = (int)n_vertex;
  out.xyz.assign((size_t)n_vertex * 3, 0.f);
  out.nrm.assign((size_t)n_vertex * 3, 0.f);
  const bool normals = inx >= 0 && iny >= 0 && inz >= 0;
  if (ascii) {
    // numbers by ascii_number() straight from the file's bytes (NUL-terminated by FileBytes): stream extraction of
    // 43 000 doubles was 4.3 of the drop-in's 6 ms per object
    std::vector<float> vf(props.size());
    for (long i = 0; i < n_vertex; ++i) {
      for (size_t k = 0; k < props.size(); ++k)
        if (!ascii_number(&p, &vf[k])) return false;
      out.xyz[3 * i] = vf[ix]; out.xyz[3 * i + 1] = vf[iy]; out.xyz[3 * i + 2] = vf[iz];
      if (normals) {
        out.nrm[3 * i] = vf[inx]; out.nrm[3 * i + 1] = vf[iny]; out.nrm[3 * i + 2] = vf[inz];
      }
    }
  } else {
    size_t stride = 0;
    std::vector<size_t> off(props.size());
    for (size_t k = 0; k < props.size(); ++k) { off[k] = stride; stride += props[k].size; }
    if ((unsigned long long)n_vertex * stride > left) return false;
    std::vector<double> v(props.size());
    const unsigned char* rec = reinterpret_cast<const unsigned char*>(p);
    for (long i = 0; i < n_vertex; ++i, rec += stride) {
      for (size_t k = 0; k < props.size(); ++k) v[k] = ply_read_bin(rec + off[k], props[k].type);
      out.xyz[3 * i] = (float)v[ix]; out.xyz[3 * i + 1] = (float)v[iy]; out.xyz[3 * i + 2] = (float)v[iz];
      if (normals) {
        out.nrm[3 * i] = (float)v[inx]; out.nrm[3 * i + 1] = (float)v[iny]; out.nrm[3 * i + 2] = (float)v[inz];
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// PNG: 8/16-bit greyscale, non-interlaced (what cv::imwrite produces for the CV_16UC1
// probability image read back at base.cc:317).  zlib inflates, the five scanline filters are
// undone here.
// ---------------------------------------------------------------------------------------------
// `last_row` (optional): the decoder looks at it after every band of rows and stops once the rows up to it are
// done -- the caller publishes the last image row its points fall on as soon as it knows it (pgp_image_rows_needed),
// rows beyond stay zero and are never read.  The inflate of the whole 640 x 480 x 16 bit image is ~1 ms, the longest
// single step of the file hand-off.
// Adler-32 of a decoded image (what a zlib stream carries of its content): zlib's own routine runs at ~3 GB/s -- 0.2 ms
// of a 0.35 ms PNG hand-off for a 640 x 480 x 16 bit image --, the 32-bytes-at-a-time form at several times that.
// Same value by construction (sums modulo 65521, blocks short enough that nothing overflows); falls back to zlib's
// where the CPU has no AVX2.  tests/test_parsers_fuzz.py checks it against zlib's on random buffers.
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline uint32_t adler32_avx2(const unsigned char* p, size_t len) {
  uint64_t s1 = 1, s2 = 0;
  const __m256i weights = _mm256_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8,
                                           7, 6, 5, 4, 3, 2, 1);
  const __m256i ones16 = _mm256_set1_epi16(1), zero = _mm256_setzero_si256();
  while (len >= 32) {
    const size_t n = std::min(len, (size_t)5536) & ~(size_t)31;   // < NMAX = 5552: the 32-bit lanes cannot overflow
    __m256i v_s1 = zero, v_s2 = zero, v_before = zero;
    for (size_t i = 0; i < n; i += 32) {
      const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + i));
      v_before = _mm256_add_epi32(v_before, v_s1);                                          // the byte sums in front of this block
      v_s1 = _mm256_add_epi32(v_s1, _mm256_sad_epu8(b, zero));                              // four partial byte sums
      v_s2 = _mm256_add_epi32(v_s2, _mm256_madd_epi16(_mm256_maddubs_epi16(b, weights), ones16));   // sum of (32 - i) b_i
    }
    uint32_t t1[8], t2[8], tb[8];
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(t1), v_s1);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(t2), v_s2);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(tb), v_before);
    uint64_t bytes = 0, weighted = 0, before = 0;
    for (int k = 0; k < 8; ++k) {
      bytes += t1[k];
      weighted += t2[k];
      before += tb[k];
    }
    s2 = (s2 + s1 * n + 32 * before + weighted) % 65521u;
    s1 = (s1 + bytes) % 65521u;
    p += n;
    len -= n;
  }
  for (; len; --len) {
    s1 += *p++;
    s2 += s1;
  }
  s1 %= 65521u;
  s2 %= 65521u;
  return (uint32_t)((s2 << 16) | s1);
}
#endif
inline uint32_t adler32_of(const unsigned char* p, size_t len) {
#if defined(__x86_64__)
  static const bool has_avx2 = __builtin_cpu_supports("avx2");
  if (has_avx2) return adler32_avx2(p, len);
#endif
  uLong a = adler32(0L, Z_NULL, 0);
  while (len) {   // (zlib takes a 32-bit length)
    const size_t n = std::min(len, (size_t)1 << 30);
    a = adler32(a, p, (uInt)n);
    p += n;
    len -= n;
  }
  return (uint32_t)a;
}

// The inflated stream of the last image, kept between calls (one per process, taken by whoever gets it first): a fresh
// 600 KB vector per call is an mmap, 150 page faults and a munmap -- more than the inflate itself.
// Scanline work of the PNG reader, 16 bytes at a time (SSE2, part of x86-64 itself): what cv::imwrite produces is Sub on
// every row (grfmt_png.cpp sets PNG_FILTER_SUB), a running sum along the row per byte lane -- log-step inside a vector, the
// last pixel carried into the next -- and big-endian samples.  Byte-wise loops at -O2 made unfilter + byte swap 0.14 ms of
// a 0.35 ms hand-off for 640 x 480 x 16 bit.  The plain loops serve other targets and the ends of rows.
#if defined(__x86_64__)
inline size_t png_unsub_sse2(const unsigned char* in, unsigned char* cur, size_t stride, size_t bpp) {
  if (bpp != 1 && bpp != 2) return 0;
  __m128i carry = _mm_setzero_si128();
  size_t i = 0;
  for (; i + 16 <= stride; i += 16) {
    __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(in + i));
    if (bpp == 1) x = _mm_add_epi8(x, _mm_slli_si128(x, 1));
    x = _mm_add_epi8(x, _mm_slli_si128(x, 2));
    x = _mm_add_epi8(x, _mm_slli_si128(x, 4));
    x = _mm_add_epi8(x, _mm_slli_si128(x, 8));
    x = _mm_add_epi8(x, carry);
    _mm_storeu_si128(reinterpret_cast<__m128i*>(cur + i), x);
    if (bpp == 2) {
      carry = _mm_shufflehi_epi16(x, 0xFF);
      carry = _mm_unpackhi_epi64(carry, carry);
    } else {
      carry = _mm_srli_si128(x, 15);
      carry = _mm_unpacklo_epi8(carry, carry);
      carry = _mm_unpacklo_epi16(carry, carry);
      carry = _mm_shuffle_epi32(carry, 0);
    }
  }
  return i;
}
inline size_t png_unup_sse2(const unsigned char* in, const unsigned char* prev, unsigned char* cur, size_t stride) {
  size_t i = 0;
  for (; i + 16 <= stride; i += 16)
    _mm_storeu_si128(reinterpret_cast<__m128i*>(cur + i),
                     _mm_add_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(in + i)),
                                  _mm_loadu_si128(reinterpret_cast<const __m128i*>(prev + i))));
  return i;
}
inline int png_swap16_sse2(const unsigned char* row, uint16_t* out, int cols) {
  int c = 0;
  for (; c + 8 <= cols; c += 8) {
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + 2 * (size_t)c));
    _mm_storeu_si128(reinterpret_cast<__m128i*>(out + c), _mm_or_si128(_mm_slli_epi16(v, 8), _mm_srli_epi16(v, 8)));
  }
  return c;
}
#else
inline size_t png_unsub_sse2(const unsigned char*, unsigned char*, size_t, size_t) { return 0; }
inline size_t png_unup_sse2(const unsigned char*, const unsigned char*, unsigned char*, size_t) { return 0; }
inline int png_swap16_sse2(const unsigned char*, uint16_t*, int) { return 0; }
#endif

struct PngScratch {
  std::mutex mu;
  std::vector<unsigned char> raw;
};
inline PngScratch& png_scratch() {
  static PngScratch s;
  return s;
}

inline bool read_png_gray(const std::string& path, std::vector<uint16_t>& px, int& rows, int& cols,
                          const std::atomic<int>* last_row = nullptr) {
  FileBytes bytes;   // (one read(): a stream iterator takes 0.5 ms over the 400 KB of a dense image)
  if (!bytes.read(path)) return false;
  const unsigned char* const file = reinterpret_cast<const unsigned char*>(bytes.data);
  const size_t file_size = bytes.len;
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file_size < 8 || std::memcmp(file, sig, 8) != 0) return false;
  size_t pos = 8;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<unsigned char> idat;
  idat.reserve(file_size);
  auto be32 = [&](size_t p) { return ((uint32_t)file[p] << 24) | ((uint32_t)file[p + 1] << 16) | ((uint32_t)file[p + 2] << 8) | file[p + 3]; };
  while (pos + 12 <= file_size) {
    uint32_t len = be32(pos);
    const std::string type(reinterpret_cast<const char*>(file + pos + 4), 4);
    size_t data = pos + 8;
    if (data + len + 4 > file_size) return false;
    if (type == "IHDR") {
      if (len != 13) return false;
      const uint32_t wc = be32(data), hr = be32(data + 4);
      if (wc == 0 || hr == 0 || wc > 16384u || hr > 16384u) return false;   // (a camera image; also keeps every size below 2^31)
      cols = (int)wc;
      rows = (int)hr;
      depth = file[data + 8];
      ctype = file[data + 9];
      interlace = file[data + 12];
    } else if (type == "IDAT") {
      idat.insert(idat.end(), file + data, file + data + len);
    } else if (type == "IEND") {
      break;
    }
    pos = data + len + 4;
  }
  if (ctype != 0 || interlace != 0 || (depth != 8 && depth != 16) || rows <= 0 || cols <= 0) return false;
  const int bpp = depth / 8;
  const size_t stride = (size_t)cols * bpp;
  // the header's size against what the compressed data can hold at all: DEFLATE expands by at most 1032 : 1
  if (idat.size() > 0x7FFFFFFFu || (stride + 1) * (size_t)rows > 1032 * idat.size() + 64) return false;
  // the inflated stream: filter byte + filtered row, per row -- in the process's scratch when nobody else holds it
  PngScratch& scratch = png_scratch();
  std::unique_lock<std::mutex> scratch_lock(scratch.mu, std::try_to_lock);
  std::vector<unsigned char> raw_own;
  std::vector<unsigned char>& raw = scratch_lock.owns_lock() ? scratch.raw : raw_own;
  // (the process keeps the scratch between calls only at the size of camera frames: one 16384 x 16384 image would
  //  otherwise leave half a gigabyte resident in the node for good -- ADVICE r5; released before the lock is)
  struct TrimScratch {
    std::vector<unsigned char>* v;
    ~TrimScratch() {
      if (v && v->capacity() > ((size_t)16 << 20)) std::vector<unsigned char>().swap(*v);
    }
  } trim_scratch{scratch_lock.owns_lock() ? &scratch.raw : nullptr};
  raw.resize((stride + 1) * (size_t)rows);   // (every byte that is read below has been written by the inflate before)
  px.resize((size_t)rows * cols);            // (the rows that stay undecoded are zeroed at the end)
  int rows_done = 0;
  const size_t B = (size_t)bpp;
  // inflate: the decoder of fast_inflate.h, band by band (its history IS `raw`, which therefore stays as inflated: the
  // rows are unfiltered into a pair of row buffers).  zlib's own inflate takes over from the start if the decoder ever
  // refuses the stream, or if the Adler-32 of a completely decoded image does not match the stream's.
  fastinf::Inflater fast(idat.data(), idat.size(), raw.data(), raw.size());
  bool use_zlib = getenv("PGP_SHIM_ZLIB") != nullptr;   // A/B and test knob
  z_stream zs;
  std::memset(&zs, 0, sizeof zs);
  bool zs_open = false;
  auto zlib_to = [&](size_t limit) -> bool {   // raw[0, limit) inflated by zlib afterwards
    if (!zs_open) {
      if (inflateInit(&zs) != Z_OK) return false;
      zs_open = true;
      zs.next_in = idat.data();
      zs.avail_in = (uInt)idat.size();
      zs.next_out = raw.data();
    }
    const size_t have = (size_t)(zs.next_out - raw.data());
    if (limit <= have) return true;
    zs.avail_out = (uInt)(limit - have);
    while (zs.avail_out > 0) {
      const int rc = inflate(&zs, Z_NO_FLUSH);
      if (rc == Z_STREAM_END) break;
      if (rc != Z_OK) return false;
    }
    return zs.avail_out == 0;   // else the stream ended before the image did
  };
  // the row above, unfiltered: zeros above the first row, then either the inflated row itself (filter None: `raw` stays
  // as inflated) or one of two row buffers
  std::vector<unsigned char> row_zero(stride, 0), row_a(stride), row_b(stride);
  const unsigned char* prev = row_zero.data();
  const int band = 32;   // rows per inflate step
  bool ok = true;
  for (int r0 = 0; r0 < rows && ok; r0 += band) {
    const int r1 = std::min(rows, r0 + band);
    const size_t limit = (stride + 1) * (size_t)r1;
    if (!use_zlib) {
      bool good = fast.run(limit) && fast.produced() >= limit;
      if (good && r1 == rows) {   // the whole image: the stream ends here and carries the checksum of what was decoded
        uint32_t want = 0;
        good = fast.finish() && fast.produced() == raw.size() && fast.trailer(&want) &&
               adler32_of(raw.data(), raw.size()) == want;
      }
      if (!good) {
        use_zlib = true;     // zlib starts over, and so do the rows
        prev = row_zero.data();
        r0 = -band;
        rows_done = 0;
        continue;
      }
    } else if (!zlib_to(limit)) {
      ok = false;
      break;
    }
    // undo the scanline filters, one specialised loop per row (the filter type is per row; a switch inside the
    // per-byte loop made this the dearest part of the whole file hand-off: 1.7 ms of 2 at 640 x 480 x 16 bit)
    for (int r = r0; r < r1; ++r) {
      const unsigned char* in = raw.data() + (stride + 1) * (size_t)r + 1;
      unsigned char* cur = prev == row_a.data() ? row_b.data() : row_a.data();
      const unsigned char* row = cur;
      switch (in[-1]) {
        case 0: row = in; break;
        case 1: {
          size_t i = png_unsub_sse2(in, cur, stride, B);
          for (; i < stride; ++i) cur[i] = (unsigned char)(in[i] + (i >= B ? cur[i - B] : 0));
          break;
        }
        case 2: {
          size_t i = png_unup_sse2(in, prev, cur, stride);
          for (; i < stride; ++i) cur[i] = (unsigned char)(in[i] + prev[i]);
          break;
        }
        case 3:
          for (size_t i = 0; i < B && i < stride; ++i) cur[i] = (unsigned char)(in[i] + (prev[i] >> 1));
          for (size_t i = B; i < stride; ++i) cur[i] = (unsigned char)(in[i] + ((cur[i - B] + prev[i]) >> 1));
          break;
        case 4:
          for (size_t i = 0; i < B && i < stride; ++i) cur[i] = (unsigned char)(in[i] + prev[i]);   // a = c = 0: predictor b
          for (size_t i = B; i < stride; ++i) {
            const int a = cur[i - B], b2 = prev[i], c = prev[i - B];
            const int p = a + b2 - c, pa = std::abs(p - a), pb = std::abs(p - b2), pc = std::abs(p - c);
            cur[i] = (unsigned char)(in[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b2 : c)));
          }
          break;
        default: ok = false;
      }
      if (!ok) break;
      uint16_t* out = px.data() + (size_t)r * cols;
      if (depth == 16) {
        for (int cidx = png_swap16_sse2(row, out, cols); cidx < cols; ++cidx) out[cidx] = (uint16_t)((row[2 * cidx] << 8) | row[2 * cidx + 1]);
      } else {
        for (int cidx = 0; cidx < cols; ++cidx) out[cidx] = row[cidx];
      }
      prev = row;
      rows_done = r + 1;
    }
    if (ok && last_row && last_row->load(std::memory_order_acquire) < r1) break;   // every row anyone will read is done
  }
  if (zs_open) inflateEnd(&zs);
  if (rows_done < rows) std::fill(px.begin() + (size_t)rows_done * cols, px.end(), (uint16_t)0);
  return ok;
}

}  // namespace shimio
