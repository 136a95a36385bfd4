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

inline bool read_ply(const std::string& path, Cloud& out) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) return false;
  std::string line;
  if (!std::getline(f, line) || line.substr(0, 3) != "ply") return false;
  bool ascii = true, in_vertex = false;
  long n_vertex = 0;
  std::vector<PlyProp> props;
  while (std::getline(f, line)) {
    if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    std::istringstream ss(line);
    std::string tok;
    ss >> tok;
    if (tok == "format") {
      ss >> tok;
      if (tok == "ascii") ascii = true;
      else if (tok == "binary_little_endian") ascii = false;
      else return false;
    } else if (tok == "element") {
      std::string name;
      long cnt = -1;
      ss >> name >> cnt;
      if (ss.fail() || cnt < 0) return false;
      in_vertex = name == "vertex";
      if (in_vertex) n_vertex = cnt;
    } else if (tok == "property" && in_vertex) {
      PlyProp p;
      ss >> p.type;
      if (p.type == "list") return false;
      ss >> p.name;
      p.size = ply_type_size(p.type);
      if (!p.size) return false;
      props.push_back(p);
    } else if (tok == "end_header") {
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
  if (ix < 0 || iy < 0 || iz < 0 || n_vertex < 0 || n_vertex > INT_MAX || !f) return false;
  // `element vertex N` is only a claim: N rows need N x (bytes per row) of file behind the header -- the record size of
  // a binary file, at least one character and one separator per property of an ASCII one
  {
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streamoff left = f.tellg() - here;
    f.seekg(here);
    size_t row = 0;
    for (size_t k = 0; k < props.size(); ++k) row += ascii ? 2 : (size_t)props[k].size;
    if (left < 0 || row == 0 || (unsigned long long)n_vertex > (unsigned long long)left / row + 1) return false;
  }
  out.n = (int)n_vertex;
  out.xyz.assign((size_t)n_vertex * 3, 0.f);
  out.nrm.assign((size_t)n_vertex * 3, 0.f);
  std::vector<double> v(props.size());
  std::vector<float> vf(props.size());
  if (ascii) {
    // the vertex block in one read, numbers by ascii_number(): stream extraction of 43 000 doubles was 4.3 of
    // the drop-in's 6 ms per object
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streamoff len = f.tellg() - here;
    f.seekg(here);
    if (len < 0) return false;
    std::vector<char> text((size_t)len + 1);
    if (len > 0 && !f.read(text.data(), len)) return false;
    text[(size_t)len] = '\0';
    const char* p = text.data();
    for (long i = 0; i < n_vertex; ++i) {
      for (size_t k = 0; k < props.size(); ++k)
        if (!ascii_number(&p, &vf[k])) return false;
      out.xyz[3 * i] = vf[ix]; out.xyz[3 * i + 1] = vf[iy]; out.xyz[3 * i + 2] = vf[iz];
      if (inx >= 0 && iny >= 0 && inz >= 0) {
        out.nrm[3 * i] = vf[inx]; out.nrm[3 * i + 1] = vf[iny]; out.nrm[3 * i + 2] = vf[inz];
      }
    }
  } else {
    size_t stride = 0;
    std::vector<size_t> off(props.size());
    for (size_t k = 0; k < props.size(); ++k) { off[k] = stride; stride += props[k].size; }
    std::vector<unsigned char> buf(stride);
    for (long i = 0; i < n_vertex; ++i) {
      if (!f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)stride)) return false;
      for (size_t k = 0; k < props.size(); ++k) v[k] = ply_read_bin(buf.data() + off[k], props[k].type);
      out.xyz[3 * i] = (float)v[ix]; out.xyz[3 * i + 1] = (float)v[iy]; out.xyz[3 * i + 2] = (float)v[iz];
      if (inx >= 0 && iny >= 0 && inz >= 0) {
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
inline bool read_png_gray(const std::string& path, std::vector<uint16_t>& px, int& rows, int& cols,
                          const std::atomic<int>* last_row = nullptr) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) return false;
  std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return false;
  size_t pos = 8;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<unsigned char> idat;
  auto be32 = [&](size_t p) { return ((uint32_t)file[p] << 24) | ((uint32_t)file[p + 1] << 16) | ((uint32_t)file[p + 2] << 8) | file[p + 3]; };
  while (pos + 12 <= file.size()) {
    uint32_t len = be32(pos);
    std::string type(reinterpret_cast<char*>(&file[pos + 4]), 4);
    size_t data = pos + 8;
    if (data + len + 4 > file.size()) return false;
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
      idat.insert(idat.end(), file.begin() + data, file.begin() + data + len);
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
  std::vector<unsigned char> raw((stride + 1) * (size_t)rows);   // the inflated stream: filter byte + filtered row, per row
  px.assign((size_t)rows * cols, 0);
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
  std::vector<unsigned char> row_a(stride, 0), row_b(stride, 0);
  unsigned char* prev = row_a.data();   // the row above, unfiltered (zeros above the first row)
  unsigned char* cur = row_b.data();
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
               (uint32_t)adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size()) == want;
      }
      if (!good) {
        use_zlib = true;     // zlib starts over, and so do the rows
        std::fill(row_a.begin(), row_a.end(), 0);
        prev = row_a.data();
        cur = row_b.data();
        r0 = -band;
        continue;
      }
    } else if (!zlib_to(limit)) {
      ok = false;
      break;
    }
    // undo the scanline filters, one specialised loop per row (the filter type is per row; a switch inside the
    // per-byte loop made this the dearest part of the whole file hand-off: 1.7 ms of 2 at 640 x 480 x 16 bit)
    for (int r = r0; r < r1; ++r) {
      const unsigned char* in = raw.data() + (stride + 1) * (size_t)r;
      std::memcpy(cur, in + 1, stride);
      switch (in[0]) {
        case 0: break;
        case 1:
          for (size_t i = B; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + cur[i - B]);
          break;
        case 2:
          for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]);
          break;
        case 3:
          for (size_t i = 0; i < B && i < stride; ++i) cur[i] = (unsigned char)(cur[i] + (prev[i] >> 1));
          for (size_t i = B; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + ((cur[i - B] + prev[i]) >> 1));
          break;
        case 4:
          for (size_t i = 0; i < B && i < stride; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]);   // a = c = 0: predictor b
          for (size_t i = B; i < stride; ++i) {
            const int a = cur[i - B], b2 = prev[i], c = prev[i - B];
            const int p = a + b2 - c, pa = std::abs(p - a), pb = std::abs(p - b2), pc = std::abs(p - c);
            cur[i] = (unsigned char)(cur[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b2 : c)));
          }
          break;
        default: ok = false;
      }
      if (!ok) break;
      uint16_t* out = px.data() + (size_t)r * cols;
      if (depth == 16)
        for (int cidx = 0; cidx < cols; ++cidx) out[cidx] = (uint16_t)((cur[2 * cidx] << 8) | cur[2 * cidx + 1]);
      else
        for (int cidx = 0; cidx < cols; ++cidx) out[cidx] = cur[cidx];
      std::swap(prev, cur);
    }
    if (ok && last_row && last_row->load(std::memory_order_acquire) < r1) break;   // every row anyone will read is done
  }
  if (zs_open) inflateEnd(&zs);
  return ok;
}

}  // namespace shimio
