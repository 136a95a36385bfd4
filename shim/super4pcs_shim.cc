// shim/super4pcs_shim.cc -- libsuper4pcs.so replacement: exports the reference's own C++ entry
// point
//
//   void getProbableTransformsSuper4PCS(std::string, std::string, std::string,
//        std::pair<Eigen::Isometry3d,float>&, std::vector<std::pair<Eigen::Isometry3d,float>>&,
//        std::string, std::map<std::vector<int>, std::vector<std::pair<int,int>>>&, int,
//        Eigen::Matrix3f, std::string, std::string, std::vector<int>&)
//
// (defined in the reference at S4/super4pcs_test.cc:39-111, declared by its only caller at
// PPE/hypothesis_generation/ObjectPoseCandidateSet.cpp:5-9) on top of the C ABI of
// include/pgp.h.  The node links this in place of the reference library and is otherwise
// unchanged.  Everything data-parallel runs on the GPU (congruent quads, rigid fits, weighted
// LCP scoring); what stays here is what the reference keeps serial and RNG-driven:
//   file hand-off      PLY x3 + 16-bit probability PNG        super4pcs_test.cc:58-80, base.cc:317-340
//   init()             centring, per-point weights            base.cc:216-345
//   base selection     StoCS sampling over the PPF map        base.cc:600-792, 582-598, 150-160
//   base pairing       TryQuadrilateral / segment distance    base.cc:415-464, 81-148
//   bookkeeping        <=100 quads per base, running best     base.cc:1855-1874, 1885-1914
//
// Needs Eigen only for the types in the signature (build against the node's Eigen; this
// repository compiles it against the reference's vendored copy when /root/reference exists).
// Differences from the reference, all deliberate: read errors return identity/0 instead of
// exit(-1); hull.ply (hard-coded path, dead code) is not read; base selection gives up after
// 20 x 100 failed draws instead of looping forever; the best pose is taken from the
// un-truncated list (the reference indexes the truncated one, base.cc:1787-1790);
// PGP_SHIM_SEED fixes the RNG seed (the reference seeds from the clock).

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include <zlib.h>

#include <Eigen/Core>
#include <Eigen/Geometry>

#include "../include/pgp.h"
#include "super4pcs_shim.h"

namespace {

typedef float Scalar;
typedef Eigen::Matrix<Scalar, 3, 1> Vec3;

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

int ply_type_size(const std::string& t) {
  if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
  if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
  if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
  if (t == "double" || t == "float64") return 8;
  return 0;
}

double ply_read_bin(const unsigned char* p, const std::string& t) {
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

bool read_ply(const std::string& path, Cloud& out) {
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
      long cnt;
      ss >> name >> cnt;
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
  if (ix < 0 || iy < 0 || iz < 0 || n_vertex < 0) return false;
  out.n = (int)n_vertex;
  out.xyz.assign((size_t)n_vertex * 3, 0.f);
  out.nrm.assign((size_t)n_vertex * 3, 0.f);
  std::vector<double> v(props.size());
  if (ascii) {
    for (long i = 0; i < n_vertex; ++i) {
      for (size_t k = 0; k < props.size(); ++k)
        if (!(f >> v[k])) return false;
      out.xyz[3 * i] = (float)v[ix]; out.xyz[3 * i + 1] = (float)v[iy]; out.xyz[3 * i + 2] = (float)v[iz];
      if (inx >= 0 && iny >= 0 && inz >= 0) {
        out.nrm[3 * i] = (float)v[inx]; out.nrm[3 * i + 1] = (float)v[iny]; out.nrm[3 * i + 2] = (float)v[inz];
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
bool read_png_gray(const std::string& path, std::vector<uint16_t>& px, int& rows, int& cols) {
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
      cols = (int)be32(data);
      rows = (int)be32(data + 4);
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
  std::vector<unsigned char> raw((stride + 1) * (size_t)rows);
  uLongf raw_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;
  std::vector<unsigned char> prev(stride, 0), cur(stride);
  px.assign((size_t)rows * cols, 0);
  for (int r = 0; r < rows; ++r) {
    const unsigned char* in = raw.data() + (stride + 1) * (size_t)r;
    const int ft = in[0];
    for (size_t i = 0; i < stride; ++i) {
      int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int x = in[1 + i], rec;
      switch (ft) {
        case 0: rec = x; break;
        case 1: rec = x + a; break;
        case 2: rec = x + b; break;
        case 3: rec = x + ((a + b) >> 1); break;
        case 4: {
          int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          rec = x + pr;
          break;
        }
        default: return false;
      }
      cur[i] = (unsigned char)(rec & 0xFF);
    }
    for (int cidx = 0; cidx < cols; ++cidx)
      px[(size_t)r * cols + cidx] = depth == 16 ? (uint16_t)((cur[2 * cidx] << 8) | cur[2 * cidx + 1]) : cur[cidx];
    prev.swap(cur);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Host side of Match4PCSBase (serial, RNG-driven parts)
// ---------------------------------------------------------------------------------------------
struct Matcher {
  int nP = 0;
  std::vector<Vec3> P, Pn;            // centred scene + unit normals (sampled_P_3D_)
  std::vector<float> prob;            // orig_probabilities_
  std::map<std::vector<int>, std::vector<std::pair<int, int> > >* PPFMap = nullptr;
  int trans_disc = 5, rot_disc = 10;  // base.cc:303-304

  static int approximate_bin(int val, int disc) {  // base.cc:150-160
    int lower = val - (val % disc), upper = lower + disc;
    return (val - lower < upper - val) ? lower : upper;
  }

  void computePPF(int i1, int i2, std::vector<int>& ppf) const {  // base.cc:582-598
    Vec3 p1 = P[i1], p2 = P[i2], n1 = Pn[i1], n2 = Pn[i2];
    Vec3 u = p1 - p2;
    int f1 = int(u.norm() * 1000);
    int f2 = int(std::atan2(n1.cross(u).norm(), n1.dot(u)) * 180 / M_PI);
    int f3 = int(std::atan2(n2.cross(u).norm(), n2.dot(u)) * 180 / M_PI);
    int f4 = int(std::atan2(n1.cross(n2).norm(), n1.dot(n2)) * 180 / M_PI);
    ppf.push_back(approximate_bin(f1, trans_disc));
    ppf.push_back(approximate_bin(f2, rot_disc));
    ppf.push_back(approximate_bin(f3, rot_disc));
    ppf.push_back(approximate_bin(f4, rot_disc));
  }

  // base.cc:81-148
  static Scalar distSegmentToSegment(const Vec3& p1, const Vec3& p2, const Vec3& q1, const Vec3& q2,
                                     double& invariant1, double& invariant2) {
    static const double kSmallNumber = 0.0001;
    Vec3 u = p2 - p1, v = q2 - q1, w = p1 - q1;
    double a = u.dot(u), b = u.dot(v), c = v.dot(v), d = u.dot(w), e = v.dot(w);
    double f = a * c - b * b;
    double s1 = 0.0, s2 = f, t1 = 0.0, t2 = f;
    if (f < kSmallNumber) {
      s1 = 0.0; s2 = 1.0; t1 = e; t2 = c;
    } else {
      s1 = (b * e - c * d);
      t1 = (a * e - b * d);
      if (s1 < 0.0) { s1 = 0.0; t1 = e; t2 = c; }
      else if (s1 > s2) { s1 = s2; t1 = e + b; t2 = c; }
    }
    if (t1 < 0.0) {
      t1 = 0.0;
      if (-d < 0.0) s1 = 0.0;
      else if (-d > a) s1 = s2;
      else { s1 = -d; s2 = a; }
    } else if (t1 > t2) {
      t1 = t2;
      if ((-d + b) < 0.0) s1 = 0;
      else if ((-d + b) > a) s1 = s2;
      else { s1 = (-d + b); s2 = a; }
    }
    invariant1 = (std::abs(s1) < kSmallNumber ? 0.0 : s1 / s2);
    invariant2 = (std::abs(t1) < kSmallNumber ? 0.0 : t1 / t2);
    return (w + ((Scalar)invariant1 * u) - ((Scalar)invariant2 * v)).norm();
  }

  // base.cc:415-464: best of the 12 pairings of the 4 base points
  bool TryQuadrilateral(std::array<int, 4>& ids, Scalar& invariant1, Scalar& invariant2) const {
    Scalar min_distance = std::numeric_limits<Scalar>::max();
    int best[4] = {-1, -1, -1, -1};
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        if (i == j) continue;
        int k = 0;
        while (k == i || k == j) k++;
        int l = 0;
        while (l == i || l == j || l == k) l++;
        double li1, li2;
        Scalar sd = distSegmentToSegment(P[ids[i]], P[ids[j]], P[ids[k]], P[ids[l]], li1, li2);
        if (sd < min_distance) {
          min_distance = sd;
          best[0] = i; best[1] = j; best[2] = k; best[3] = l;
          invariant1 = (Scalar)li1;
          invariant2 = (Scalar)li2;
        }
      }
    if (best[0] < 0) return false;
    std::array<int, 4> tmp = ids;
    for (int k = 0; k < 4; ++k) ids[k] = tmp[best[k]];
    return true;
  }

  // base.cc:600-792: stochastic base sampling weighted by segmentation probability x existence
  // of the pair feature in the model's PPF map
  bool SelectQuadrilateralStoCS(std::default_random_engine& gen, std::array<int, 4>& ids, Scalar& inv1,
                                Scalar& inv2) const {
    std::vector<int> ppf;
    std::vector<float> cur(prob);
    std::discrete_distribution<int> d1(cur.begin(), cur.end());
    const int base1 = d1(gen);
    float sum = 0;
    bool present = false;
    for (int i = 0; i < nP; ++i) {
      if (i == base1 || cur[i] == 0) { cur[i] = 0; continue; }
      ppf.clear();
      computePPF(base1, i, ppf);
      float edge = PPFMap->find(ppf) == PPFMap->end() ? 0.f : 1.f;
      cur[i] = prob[i] * prob[base1] * edge;
      if (cur[i] != 0) present = true;
      sum += cur[i];
    }
    if (!present) return false;
    for (int i = 0; i < nP; ++i) cur[i] /= sum;
    std::discrete_distribution<int> d2(cur.begin(), cur.end());
    const int base2 = d2(gen);

    sum = 0;
    present = false;
    Vec3 v_1 = P[base2] - P[base1];
    for (int i = 0; i < nP; ++i) {
      Vec3 v_2 = P[i] - P[base1];
      float int_angle = std::acos(v_1.dot(v_2)) * 180 / M_PI;  // un-normalised dot, as the reference
      int_angle = std::min(int_angle, 180 - int_angle);
      if (i == base1 || i == base2 || cur[i] == 0 || int_angle < 30) { cur[i] = 0; continue; }
      ppf.clear();
      computePPF(base2, i, ppf);
      float edge = PPFMap->find(ppf) == PPFMap->end() ? 0.f : 1.f;
      cur[i] = cur[i] * prob[base2] * edge;
      if (cur[i] != 0) present = true;
      sum += cur[i];
    }
    if (!present) return false;
    for (int i = 0; i < nP; ++i) cur[i] /= sum;
    std::discrete_distribution<int> d3(cur.begin(), cur.end());
    const int base3 = d3(gen);

    sum = 0;
    present = false;
    const double x1 = P[base1](0), y1 = P[base1](1), z1 = P[base1](2);
    const double x2 = P[base2](0), y2 = P[base2](1), z2 = P[base2](2);
    const double x3 = P[base3](0), y3 = P[base3](1), z3 = P[base3](2);
    for (int i = 0; i < nP; ++i) {
      if (i == base1 || i == base2 || i == base3 || cur[i] == 0) { cur[i] = 0; continue; }
      Scalar denom = (-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
      if (denom != 0) {  // the 4th point must be close to the plane of the first three
        Scalar A = (-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / denom;
        Scalar B = (x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / denom;
        Scalar C = (-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / denom;
        Scalar planar = std::abs(A * P[i](0) + B * P[i](1) + C * P[i](2) - 1.0);
        if (planar > 0.01 || (P[i] - P[base1]).norm() < 0.01 || (P[i] - P[base2]).norm() < 0.01 ||
            (P[i] - P[base3]).norm() < 0.01) {
          cur[i] = 0;
          continue;
        }
      }
      ppf.clear();
      computePPF(base3, i, ppf);
      float edge = PPFMap->find(ppf) == PPFMap->end() ? 0.f : 1.f;
      cur[i] = cur[i] * prob[base3] * edge;
      if (cur[i] != 0) present = true;
      sum += cur[i];
    }
    if (!present) return false;
    for (int i = 0; i < nP; ++i) cur[i] /= sum;
    std::discrete_distribution<int> d4(cur.begin(), cur.end());
    const int base4 = d4(gen);
    ids = {base1, base2, base3, base4};
    TryQuadrilateral(ids, inv1, inv2);
    return true;
  }
};

void set_identity(std::pair<Eigen::Isometry3d, float>& h) {
  h.first.matrix().setIdentity();
  h.second = 0.f;
}

#define SHIM_PGP(call)                                                                       \
  do {                                                                                       \
    if ((call) != PGP_OK) {                                                                  \
      std::cerr << "[libsuper4pcs shim] " #call " failed: " << pgp_last_error() << std::endl; \
      if (ctx) pgp_destroy(ctx);                                                             \
      set_identity(bestHypothesis);                                                          \
      return;                                                                                \
    }                                                                                        \
  } while (0)

}  // namespace

// What the reference's reader + CleanInvalidNormals leave in Point3D::normal(): unit normals
// (Point3D::set_normal, shared4pcs.h:85-87), zero where the squared norm is < 0.01
// (utils/geometry.h:57-84).
static void clean_normals(std::vector<float>& n) {
  for (size_t i = 0; i + 2 < n.size(); i += 3) {
    Vec3 v(n[i], n[i + 1], n[i + 2]);
    v = v.normalized();                            // set_normal() while reading
    if (v.squaredNorm() < 0.01f) v.setZero();      // CleanInvalidNormals
    n[i] = v(0); n[i + 1] = v(1); n[i + 2] = v(2);
  }
}

// C-linkage probe for the tests: read one cloud the way the entry point does (reader + normal
// cleaning) into caller arrays; returns the point count or -1.
extern "C" int super4pcs_shim_read_cloud(const char* path, float* xyz, float* nrm, int cap) {
  Cloud c;
  if (!read_ply(path, c)) return -1;
  clean_normals(c.nrm);
  for (int i = 0; i < c.n && i < cap; ++i)
    for (int k = 0; k < 3; ++k) {
      xyz[3 * i + k] = c.xyz[3 * (size_t)i + k];
      nrm[3 * i + k] = c.nrm[3 * (size_t)i + k];
    }
  return c.n;
}

// C-linkage probe for the tests: decode a greyscale PNG into 16-bit samples (8-bit files are widened);
// returns 0, or -1 when the file cannot be read / is not a supported PNG, -2 when cap is too small.
extern "C" int super4pcs_shim_read_png(const char* path, unsigned short* px, int cap, int* rows, int* cols) {
  std::vector<uint16_t> v;
  int r = 0, c = 0;
  if (!read_png_gray(path, v, r, c)) return -1;
  *rows = r;
  *cols = c;
  if ((long long)r * c > cap) return -2;
  std::copy(v.begin(), v.end(), px);
  return 0;
}

bool super4pcs_shim_read_ply(const std::string& path, std::vector<float>& xyz, std::vector<float>& normals) {
  Cloud c;
  if (!read_ply(path, c)) return false;
  xyz.swap(c.xyz);
  normals.swap(c.nrm);
  return true;
}

bool super4pcs_shim_read_png16(const std::string& path, std::vector<unsigned short>& pixels, int& rows, int& cols) {
  std::vector<uint16_t> px;
  if (!read_png_gray(path, px, rows, cols)) return false;
  pixels.assign(px.begin(), px.end());
  return true;
}

void getProbableTransformsSuper4PCS(std::string input1, std::string input2, std::string input3,
                                    std::pair<Eigen::Isometry3d, float>& bestHypothesis,
                                    std::vector<std::pair<Eigen::Isometry3d, float> >& hypothesisSet,
                                    std::string probImagePath,
                                    std::map<std::vector<int>, std::vector<std::pair<int, int> > >& PPFMap,
                                    int max_count_ppf, Eigen::Matrix3f camIntrinsic, std::string objName,
                                    std::string scenePath, std::vector<int>& registered_points) {
  (void)max_count_ppf; (void)objName; (void)scenePath;
  set_identity(bestHypothesis);
  // ---- file hand-off (super4pcs_test.cc:58-80): set1 = segment, set2 = validation model, set3 = search model
  Cloud seg, qval, qsearch;
  if (!read_ply(input1, seg) || !read_ply(input2, qval) || !read_ply(input3, qsearch)) {
    std::cerr << "[libsuper4pcs shim] cannot read the input PLY files" << std::endl;
    return;
  }
  std::vector<uint16_t> px;
  int rows = 0, cols = 0;
  const bool have = read_png_gray(probImagePath, px, rows, cols);
  if (!have) std::cerr << "[libsuper4pcs shim] no probability image at " << probImagePath << ": weights = 1" << std::endl;
  const Super4PCSCloudView vs = {seg.xyz.data(), seg.nrm.data(), seg.n};
  const Super4PCSCloudView vq = {qval.xyz.data(), qval.nrm.data(), qval.n};
  const Super4PCSCloudView vqs = {qsearch.xyz.data(), qsearch.nrm.data(), qsearch.n};
  getProbableTransformsSuper4PCS(vs, vq, vqs, have ? px.data() : nullptr, rows, cols, bestHypothesis, hypothesisSet,
                                 PPFMap, camIntrinsic, registered_points);
}

void getProbableTransformsSuper4PCS(const Super4PCSCloudView& segment, const Super4PCSCloudView& model_validation,
                                    const Super4PCSCloudView& model_search, const unsigned short* prob_image,
                                    int rows, int cols,
                                    std::pair<Eigen::Isometry3d, float>& bestHypothesis,
                                    std::vector<std::pair<Eigen::Isometry3d, float> >& hypothesisSet,
                                    std::map<std::vector<int>, std::vector<std::pair<int, int> > >& PPFMap,
                                    Eigen::Matrix3f camIntrinsic, std::vector<int>& registered_points) {
  const float delta = 0.005f;              // super4pcs_test.cc:20
  const int max_number_of_bases = 100;     // base.cc:290
  const int max_sampled_csets = 100;       // base.cc:1858
  pgp_ctx* ctx = nullptr;
  set_identity(bestHypothesis);
  auto own = [](const Super4PCSCloudView& v) {   // the call centres and re-normalises: work on copies
    Cloud c;
    c.n = v.n > 0 && v.xyz ? v.n : 0;
    c.xyz.assign(v.xyz, v.xyz + 3 * (size_t)c.n);
    if (v.normals) c.nrm.assign(v.normals, v.normals + 3 * (size_t)c.n);
    else c.nrm.assign(3 * (size_t)c.n, 0.f);
    return c;
  };
  Cloud seg = own(segment), qval = own(model_validation), qsearch = own(model_search);
  if (seg.n == 0 || qval.n == 0 || qsearch.n == 0) return;
  clean_normals(seg.nrm);
  clean_normals(qval.nrm);

  // ---- init(): centring (base.cc:242-268)
  float cP[3], cQ[3];
  if (pgp_center(seg.xyz.data(), seg.n, qsearch.xyz.data(), qsearch.n, qval.xyz.data(), qval.n, cP, cQ) != PGP_OK) return;

  Matcher m;
  m.nP = seg.n;
  m.PPFMap = &PPFMap;
  m.P.resize(seg.n);
  m.Pn.resize(seg.n);
  for (int i = 0; i < seg.n; ++i) {
    m.P[i] = Vec3(seg.xyz[3 * i], seg.xyz[3 * i + 1], seg.xyz[3 * i + 2]);
    m.Pn[i] = Vec3(seg.nrm[3 * i], seg.nrm[3 * i + 1], seg.nrm[3 * i + 2]);
  }
  // ---- per-point weights from the probability image (base.cc:317-340)
  m.prob.assign(seg.n, 1.f);
  if (prob_image && rows > 0 && cols > 0) {
    float K[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) K[3 * r + c] = camIntrinsic(r, c);
    pgp_weights_from_image(seg.xyz.data(), seg.n, cP, K, prob_image, rows, cols, m.prob.data());
  }

  const auto t_start = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  // ---- device state: scene index, validation model, search model
  SHIM_PGP(pgp_create(&ctx, -1));
  SHIM_PGP(pgp_set_scene(ctx, seg.xyz.data(), seg.nrm.data(), m.prob.data(), seg.n, delta));
  SHIM_PGP(pgp_set_model(ctx, qval.xyz.data(), qval.nrm.data(), qval.n));
  SHIM_PGP(pgp_set_search_model(ctx, qsearch.xyz.data(), qsearch.n));

  const double ms_setup = ms_since(t_start);
  const auto t_bases = std::chrono::steady_clock::now();
  // ---- Step 1: base selection (base.cc:1831-1848)
  unsigned seed = (unsigned)std::chrono::system_clock::now().time_since_epoch().count();
  if (const char* s = getenv("PGP_SHIM_SEED")) { seed = (unsigned)strtoul(s, nullptr, 10); srand(seed); }
  std::default_random_engine generator(seed);
  struct Base { std::array<int, 4> ids; Scalar inv1, inv2; };
  std::vector<Base> bases;
  for (int attempt = 0; (int)bases.size() < max_number_of_bases && attempt < 20 * max_number_of_bases; ++attempt) {
    Base b;
    if (m.SelectQuadrilateralStoCS(generator, b.ids, b.inv1, b.inv2)) bases.push_back(b);
  }

  const double ms_bases = ms_since(t_bases);
  const auto t_cs = std::chrono::steady_clock::now();
  // ---- Step 2: congruent sets (base.cc:1855-1874, 1929-1993) -> (base, quad) pairs
  std::vector<int> base_ids, quad_ids;  // n x 4 each
  std::vector<int> quads;
  for (size_t bi = 0; bi < bases.size(); ++bi) {
    const Base& b = bases[bi];
    std::vector<int> ppf_1, ppf_6;
    m.computePPF(b.ids[0], b.ids[1], ppf_1);
    m.computePPF(b.ids[2], b.ids[3], ppf_6);
    auto it1 = PPFMap.find(ppf_1), it6 = PPFMap.find(ppf_6);
    if (it1 == PPFMap.end() || it6 == PPFMap.end() || it1->second.empty() || it6->second.empty()) continue;
    float base_xyz[12];
    for (int k = 0; k < 4; ++k)
      for (int d = 0; d < 3; ++d) base_xyz[3 * k + d] = m.P[b.ids[k]](d);
    static_assert(sizeof(std::pair<int, int>) == 2 * sizeof(int), "pair<int,int> must be two packed ints");
    const int* p1 = reinterpret_cast<const int*>(it1->second.data());
    const int* p6 = reinterpret_cast<const int*>(it6->second.data());
    int n_quads = 0;
    if (quads.size() < (size_t)4 << 16) quads.resize((size_t)4 << 16);
    for (;;) {
      const int cap = (int)(quads.size() / 4);
      SHIM_PGP(pgp_find_congruent(ctx, base_xyz, b.inv1, b.inv2, delta, p1, (int)it1->second.size(), p6,
                                  (int)it6->second.size(), quads.data(), cap, &n_quads));
      if (n_quads <= cap) break;
      quads.resize((size_t)n_quads * 4);
    }
    if (n_quads == 0) continue;
    std::vector<int> pick;
    if (n_quads < max_sampled_csets) {
      for (int j = 0; j < n_quads; ++j) pick.push_back(j);
    } else {  // 100 distinct random quads (the reference iterates an unordered_set; we sort)
      std::set<int> chosen;
      while ((int)chosen.size() < max_sampled_csets) chosen.insert(rand() % n_quads);
      pick.assign(chosen.begin(), chosen.end());
    }
    for (int j : pick) {
      for (int k = 0; k < 4; ++k) {
        base_ids.push_back(b.ids[k]);
        quad_ids.push_back(quads[4 * (size_t)j + k]);
      }
    }
  }
  const double ms_cs = ms_since(t_cs);
  const auto t_fit = std::chrono::steady_clock::now();
  const int n_pairs = (int)(base_ids.size() / 4);
  std::vector<float> T((size_t)n_pairs * 16);
  std::vector<double> pose((size_t)n_pairs * 16);
  std::vector<int> status(n_pairs);
  if (n_pairs > 0)
    SHIM_PGP(pgp_rigid_from_congruent(ctx, base_ids.data(), quad_ids.data(), n_pairs, cP, cQ, T.data(),
                                      pose.data(), status.data(), nullptr));
  // allTransforms / allPose hold only the fits that were pushed (base.cc:1467-1485)
  std::vector<float> allT;
  std::vector<std::pair<Eigen::Isometry3d, float> > allPose;
  for (int i = 0; i < n_pairs; ++i) {
    if (status[i] != 1) continue;
    allT.insert(allT.end(), T.begin() + 16 * (size_t)i, T.begin() + 16 * (size_t)i + 16);
    Eigen::Isometry3d iso;
    iso.matrix() = Eigen::Map<const Eigen::Matrix4d>(pose.data() + 16 * (size_t)i);
    allPose.push_back(std::make_pair(iso, 0.f));
  }

  // ---- Step 3: verification (base.cc:1885-1901), operMode = 1 -> WeightedVerify
  const int n_h = (int)allPose.size();
  const double ms_fit = ms_since(t_fit);
  if (getenv("PGP_SHIM_VERBOSE"))
    std::cerr << "[libsuper4pcs shim] bases " << bases.size() << ", congruent pairs " << n_pairs
              << ", transforms " << n_h << "; ms: setup " << ms_setup << ", base selection " << ms_bases
              << ", congruent sets " << ms_cs << ", rigid fits " << ms_fit << std::endl;
  std::vector<float> lcp(n_h);
  int best = -1;
  float best_lcp = 0.f;
  SHIM_PGP(pgp_score_lcp(ctx, allT.data(), n_h, PGP_MODE_WEIGHTED, 30.f, lcp.data(), nullptr, &best, &best_lcp));
  for (int i = 0; i < n_h; ++i) allPose[i].second = lcp[i];
  std::vector<int> selected(n_h > 0 ? n_h : 1);
  int n_sel = 0;
  pgp_running_best(lcp.data(), n_h, selected.data(), &n_sel);
  hypothesisSet.clear();   // the reference REPLACES the list by the running-best subsequence (allPose.clear(), base.cc:1903)
  for (int k = 0; k < n_sel; ++k) hypothesisSet.push_back(allPose[selected[k]]);  // base.cc:1903-1908
  if (best >= 0) {
    bestHypothesis = std::make_pair(allPose[best].first, best_lcp);
    registered_points.resize(qval.n);
    int n_reg = 0;
    SHIM_PGP(pgp_registered(ctx, allT.data() + 16 * (size_t)best, PGP_MODE_WEIGHTED, 30.f,
                            registered_points.data(), &n_reg));
    registered_points.resize(n_reg);
  } else {
    std::cout << "returning identity" << std::endl;  // base.cc:1791-1794
  }
  pgp_destroy(ctx);
}
