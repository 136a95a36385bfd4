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
// unchanged.  Steps 1-3 of Perform_N_steps run on the GPU behind the C ABI: base selection
// (pgp_select_bases: the weighting loops of SelectQuadrilateralStoCS, the point-pair features, the
// look-ups in the model's pair-feature table, the draws and TryQuadrilateral, many attempts per
// launch), congruent sets for all bases at once (pgp_find_congruent_batch), rigid fits
// (pgp_congruent_batch_fit) and weighted LCP scoring (pgp_score_lcp / pgp_multi_score_lcp).
// What stays here is the hand-off and the bookkeeping:
//   file hand-off      PLY x3 + 16-bit probability PNG        super4pcs_test.cc:58-80, base.cc:317-340
//   init()             centring, per-point weights            base.cc:216-345   (pgp_center, pgp_weights_from_image)
//   the host's engine  uniform variates of the draws, rand() of the <=100-quads sampling (base.cc:613-617,1858-1866)
//   bookkeeping        running best, output containers        base.cc:1885-1914
//
// Needs Eigen only for the types in the signature (build against the node's Eigen; this
// repository compiles it against the reference's vendored copy when /root/reference exists).
// Differences from the reference, all deliberate: read errors return identity/0 instead of
// exit(-1); hull.ply (hard-coded path, dead code) is not read; base selection gives up after
// 20 rounds of 128 attempts instead of looping forever; one engine feeds all attempts (the reference
// seeds a fresh one from the clock inside every attempt, base.cc:613-614); the best pose is taken from the
// un-truncated list (the reference indexes the truncated one, base.cc:1787-1790);
// PGP_SHIM_SEED fixes the RNG seed (the reference seeds from the clock).

#include <algorithm>
#include <atomic>
#include <array>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

#include <Eigen/Core>
#include <Eigen/Geometry>

#include "../include/pgp.h"
#include "fast_inflate.h"
#include "file_readers.h"
#include "super4pcs_shim.h"
#include "object_slots.h"
#include "frame_pool.h"

namespace {

typedef float Scalar;
typedef Eigen::Matrix<Scalar, 3, 1> Vec3;

using shimio::Cloud;
using shimio::read_ply;
using shimio::read_png_gray;
using shimio::ascii_number;

// The per-object device state and its leases: shim/object_slots.h (ObjectSlot, ShimState::acquire, SlotLease) -- a header
// without Eigen or HIP, so that `make -C shim tsan` runs the same code under ThreadSanitizer on a machine without a GPU.
// The node hands the SAME std::map of an object to every request (PPE/data_layer/Objects.cpp:31-49 fills it once), so its
// 10^4-10^5 keys are flattened and uploaded again only when a different map arrives: different address, size, first / last
// entry (keys, pair counts, first pairs) or target context.  A caller that rebuilds a map IN PLACE with the same size and the
// same two end entries would still be taken for the old one: such callers set PGP_SHIM_NO_CACHE=1 (every call gets a fresh
// context and uploads its map).
using shimstate::ObjectSlot;
using shimstate::ShimState;
using shimstate::SlotLease;
static unsigned long long cloud_hash(const std::vector<float>& a, const std::vector<float>& b) {
  unsigned long long h0 = 0x9E3779B97F4A7C15ull ^ (unsigned long long)a.size(), h1 = 0xC2B2AE3D27D4EB4Full ^ (unsigned long long)b.size();
  const uint32_t* w = reinterpret_cast<const uint32_t*>(a.data());
  for (size_t i = 0; i < a.size(); ++i) h0 = (h0 ^ w[i]) * 0x100000001B3ull + (h0 >> 29);
  w = reinterpret_cast<const uint32_t*>(b.data());
  for (size_t i = 0; i < b.size(); ++i) h1 = (h1 ^ w[i]) * 0x100000001B3ull + (h1 >> 31);
  return (h0 ^ (h1 * 0x9E3779B97F4A7C15ull)) | 1ull;
}

// set on the threads of getProbableTransformsSuper4PCSFrame: the call's quad sampling draws from a generator of its own
static thread_local bool t_private_rand = false;

ShimState& shim_state() {
  static ShimState* st = new ShimState;   // (never destroyed: see above)
  return *st;
}

int shim_device_count() {   // PGP_SHIM_DEVICES: 1 (default), a number, or "all"
  const char* v = getenv("PGP_SHIM_DEVICES");
  if (!v) return 1;
  if (!strcmp(v, "all")) return 0;
  const int n = atoi(v);
  return n > 0 ? n : 1;
}

void set_identity(std::pair<Eigen::Isometry3d, float>& h) {
  h.first.matrix().setIdentity();
  h.second = 0.f;
}

#define SHIM_PGP(call)                                                                       \
  do {                                                                                       \
    if ((call) != PGP_OK) {                                                                  \
      std::cerr << "[libsuper4pcs shim] " #call " failed: " << pgp_last_error() << std::endl; \
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

// C-linkage probe for the tests: the reader's number parser on a whitespace-separated list; returns how many
// numbers it produced (it stops at the first token it rejects).
extern "C" int super4pcs_shim_parse_numbers(const char* text, float* out, int cap) {
  const char* p = text;
  int n = 0;
  while (n < cap && ascii_number(&p, &out[n])) ++n;
  return n;
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

// C-linkage probe for the tests: the zlib-stream decoder of fast_inflate.h on its own, in steps of `step` output bytes
// (0: one run); returns the number of bytes produced, -1 when it refuses the stream, -2 on a checksum mismatch
extern "C" long long super4pcs_shim_inflate(const unsigned char* in, long long n_in, unsigned char* out, long long n_out, long long step) {
  fastinf::Inflater f(in, (size_t)n_in, out, (size_t)n_out);
  if (step > 0)
    for (long long lim = step; lim < n_out; lim += step)
      if (!f.run((size_t)lim)) return -1;
  if (!f.run((size_t)n_out) || !f.finish()) return -1;
  uint32_t want = 0;
  if (!f.trailer(&want)) return -1;
  if ((uint32_t)adler32(adler32(0L, Z_NULL, 0), out, (uInt)f.produced()) != want) return -2;
  return (long long)f.produced();
}

// the same with the decoder told (before it starts) that nothing beyond `last_row` will be read: rows up to
// last_row are decoded, rows of later bands stay zero
extern "C" int super4pcs_shim_read_png_rows(const char* path, int last_row, unsigned short* px, int cap, int* rows, int* cols) {
  std::vector<uint16_t> v;
  int r = 0, c = 0;
  const std::atomic<int> last(last_row);
  if (!read_png_gray(path, v, r, c, &last)) return -1;
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

// true when two of the n points coincide exactly (as floats: -0 == +0; a NaN point equals nothing)
static bool has_duplicate_points(const float* xyz, int n) {
  if (n < 2) return false;
  size_t cap = 1;
  while (cap < 2 * (size_t)n) cap <<= 1;
  std::vector<int> slot(cap, -1);
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + 3 * (size_t)i;
    if (p[0] != p[0] || p[1] != p[1] || p[2] != p[2]) continue;
    uint32_t b[3];
    for (int k = 0; k < 3; ++k) {
      const float v = p[k] == 0.f ? 0.f : p[k];   // -0 -> +0
      std::memcpy(&b[k], &v, 4);
    }
    size_t h = ((size_t)b[0] * 0x9E3779B1u) ^ ((size_t)b[1] * 0x85EBCA77u) ^ ((size_t)b[2] * 0xC2B2AE3Du);
    h = (h ^ (h >> 15)) & (cap - 1);
    for (;; h = (h + 1) & (cap - 1)) {
      const int j = slot[h];
      if (j < 0) {
        slot[h] = i;
        break;
      }
      const float* q = xyz + 3 * (size_t)j;
      if (q[0] == p[0] && q[1] == p[1] && q[2] == p[2]) return true;
    }
  }
  return false;
}

static void match_impl(const Super4PCSCloudView& segment, const Super4PCSCloudView& model_validation,
                       const Super4PCSCloudView& model_search,
                       const std::function<const unsigned short*(int*, int*, int)>& image,
                       std::pair<Eigen::Isometry3d, float>& bestHypothesis,
                       std::vector<std::pair<Eigen::Isometry3d, float> >& hypothesisSet,
                       std::map<std::vector<int>, std::vector<std::pair<int, int> > >& PPFMap,
                       Eigen::Matrix3f camIntrinsic, std::vector<int>& registered_points, bool want_rows = true);

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
  // the four files are parsed on four host threads (three ASCII PLYs + the PNG's inflate): the parse was
  // ~1.7 of the drop-in's 3.6 ms per object on one thread
  Cloud seg, qval, qsearch;
  // the decoded probability image: in a buffer the process keeps between calls when no other call holds it (a fresh
  // 600 KB vector per call is an mmap + 150 page faults + a munmap on the image thread's critical path)
  static std::mutex px_mu;
  static std::vector<uint16_t> px_kept;
  std::unique_lock<std::mutex> px_lock(px_mu, std::try_to_lock);
  std::vector<uint16_t> px_own;
  std::vector<uint16_t>& px = px_lock.owns_lock() ? px_kept : px_own;
  struct TrimKept {   // (kept between calls at the size of camera frames only: ADVICE r5; runs before the lock is released)
    std::vector<uint16_t>* v;
    ~TrimKept() {
      if (v && v->capacity() * sizeof(uint16_t) > ((size_t)16 << 20)) std::vector<uint16_t>().swap(*v);
    }
  } trim_kept{px_lock.owns_lock() ? &px_kept : nullptr};
  int rows = 0, cols = 0;
  bool ok1 = false, ok2 = false, ok3 = false, have = false;
  const auto t_files = std::chrono::steady_clock::now();
  double ms_part[4] = {0, 0, 0, 0}, ms_spawn = 0;
  auto timed = [&](int k, const std::function<void()>& fn) {
    const auto t0 = std::chrono::steady_clock::now();
    fn();
    ms_part[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  // A worker that throws (bad_alloc on a huge file) must not take the host node down through std::terminate: every
  // worker body catches and reports failure through its flag, and the guards join whatever is still joinable when
  // this frame unwinds -- the reference would propagate the exception, and so does this function after the join.
  struct JoinGuard {
    std::thread& t;
    ~JoinGuard() { if (t.joinable()) t.join(); }
  };
  auto guarded = [](bool* flag, const std::function<bool()>& body) {
    try {
      *flag = body();
    } catch (...) {
      *flag = false;
    }
  };
  // Starting a thread costs this one ~0.016 ms (MI355X host), so it starts ONE -- the validation model's reader, which
  // first starts the other two -- and is parsing the segment, the longest of the three clouds, 0.03 ms earlier.  (Threads
  // kept between calls start in 0.006 ms but run slower: woken next to the thread that wakes them, they share its core
  // -- segment 0.154 -> 0.22 ms, the call 0.90 -> 1.01 ms; measured, not kept.)  A reader that cannot get a thread runs
  // inline on the one that wanted to start it.
  // The PNG (inflate + unfilter) keeps decoding while the clouds are centred, uploaded and indexed; the match waits
  // for it only where the weights are first needed.
  std::atomic<int> last_row(1 << 30);   // until the match knows which rows its points fall on: all of them
  const std::function<void()> read_image = [&] { timed(3, [&] { guarded(&have, [&] { return read_png_gray(probImagePath, px, rows, cols, &last_row); }); }); };
  const std::function<void()> read_search = [&] { timed(2, [&] { guarded(&ok3, [&] { return read_ply(input3, qsearch); }); }); };
  const std::function<void()> read_validation = [&] { timed(1, [&] { guarded(&ok2, [&] { return read_ply(input2, qval); }); }); };
  auto on_a_thread = [](std::thread& t, const std::function<void()>& body) {
    try {
      t = std::thread(body);
    } catch (const std::system_error&) {
      body();
    }
  };
  std::thread t4, t3, t2;   // t4 and t3 are started by t2's thread: only looked at after t2 is joined
  JoinGuard g4{t4}, g3{t3}, g2{t2};   // (destroyed in reverse: t2 first)
  on_a_thread(t2, [&] {
    on_a_thread(t4, read_image);
    on_a_thread(t3, read_search);
    read_validation();
  });
  ms_spawn = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_files).count();
  timed(0, [&] { ok1 = read_ply(input1, seg); });
  if (t2.joinable()) t2.join();
  if (t3.joinable()) t3.join();
  if (!ok1 || !ok2 || !ok3) {
    if (t4.joinable()) t4.join();
    std::cerr << "[libsuper4pcs shim] cannot read the input PLY files" << std::endl;
    return;
  }
  const Super4PCSCloudView vs = {seg.xyz.data(), seg.nrm.data(), seg.n};
  const Super4PCSCloudView vq = {qval.xyz.data(), qval.nrm.data(), qval.n};
  const Super4PCSCloudView vqs = {qsearch.xyz.data(), qsearch.nrm.data(), qsearch.n};
  bool joined = false;
  auto image = [&](int* r, int* c, int last_needed) -> const unsigned short* {
    if (!joined) {
      last_row.store(last_needed, std::memory_order_release);
      if (t4.joinable()) t4.join();
      joined = true;
      if (getenv("PGP_SHIM_VERBOSE"))
        std::cerr << "[libsuper4pcs shim] file hand-off: image ready " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_files).count()
                  << " ms after the call (reader thread started " << ms_spawn << ", segment " << ms_part[0] << ", validation model " << ms_part[1] << ", search model " << ms_part[2]
                  << ", probability image " << ms_part[3] << " ms, in parallel)" << std::endl;
      if (!have) std::cerr << "[libsuper4pcs shim] no probability image at " << probImagePath << ": weights = 1" << std::endl;
    }
    *r = rows;
    *c = cols;
    return have ? px.data() : nullptr;
  };
  const double ms_parsed = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_files).count();
  match_impl(vs, vq, vqs, image, bestHypothesis, hypothesisSet, PPFMap, camIntrinsic, registered_points);
  if (t4.joinable()) t4.join();
  if (getenv("PGP_SHIM_VERBOSE"))
    std::cerr << "[libsuper4pcs shim] file hand-off: clouds parsed " << ms_parsed << " ms after the call, matched "
              << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_files).count() << std::endl;
}

void getProbableTransformsSuper4PCS(const Super4PCSCloudView& segment, const Super4PCSCloudView& model_validation,
                                    const Super4PCSCloudView& model_search, const unsigned short* prob_image,
                                    int rows, int cols,
                                    std::pair<Eigen::Isometry3d, float>& bestHypothesis,
                                    std::vector<std::pair<Eigen::Isometry3d, float> >& hypothesisSet,
                                    std::map<std::vector<int>, std::vector<std::pair<int, int> > >& PPFMap,
                                    Eigen::Matrix3f camIntrinsic, std::vector<int>& registered_points) {
  match_impl(segment, model_validation, model_search,
             [&](int* r, int* c, int) -> const unsigned short* { *r = rows; *c = cols; return prob_image; },
             bestHypothesis, hypothesisSet, PPFMap, camIntrinsic, registered_points, false);
}

// ---- the objects of one frame side by side --------------------------------------------------------------------------
// The node matches the objects of a frame one after the other (SceneCfg.cpp:379-402 -> ObjectPoseCandidateSet.cpp:53-68); a
// call is a chain of short device steps with the host in between (three round trips), so one object leaves most of the GPU
// and most of the call's wall-clock unused.  Here every job runs on a thread the process keeps (job k on worker k % 8), finds
// its object's context, models and pair-feature table from the frame before in the process's state (ShimState: one slot per
// object), and the jobs' device work overlaps on their contexts' streams.  Same results as the jobs called one by one
// (with a generator of the call's own for the quad sampling: PGP_SHIM_PRIVATE_RAND above).
namespace {
// (the kept worker threads: shim/frame_pool.h, HIP- and Eigen-free for the ThreadSanitizer build)
using shimstate::FramePool;
FramePool* frame_pool() {
  static FramePool* pool = FramePool::make([] { t_private_rand = true; });
  return pool;
}
}  // namespace

void getProbableTransformsSuper4PCSFrame(Super4PCSJob* jobs, int n_jobs) {
  if (!jobs || n_jobs <= 0) return;
  auto run = [jobs](int j) {
    Super4PCSJob& q = jobs[j];
    set_identity(q.bestHypothesis);
    q.hypothesisSet.clear();
    q.registered_points.clear();
    q.failed = false;
    if (!q.PPFMap) {
      q.failed = true;
      return;
    }
    try {
      getProbableTransformsSuper4PCS(q.segment, q.model_validation, q.model_search, q.prob_image, q.rows, q.cols, q.bestHypothesis,
                                     q.hypothesisSet, *q.PPFMap, q.camIntrinsic, q.registered_points);
    } catch (...) {
      q.failed = true;   // (what the single call would have thrown: bad_alloc; the other jobs of the frame are not lost)
    }
  };
  FramePool* pool = n_jobs > 1 && !getenv("PGP_SHIM_FRAME_SERIAL") ? frame_pool() : nullptr;
  if (!pool) {
    const bool before = t_private_rand;
    t_private_rand = true;
    for (int j = 0; j < n_jobs; ++j) run(j);
    t_private_rand = before;
    return;
  }
  std::lock_guard<std::mutex> frame(pool->use_mu);
  const int used = n_jobs < FramePool::kWorkers ? n_jobs : FramePool::kWorkers;
  for (int k = 0; k < used; ++k)
    pool->start(k, [=] {
      for (int j = k; j < n_jobs; j += FramePool::kWorkers) run(j);
    });
  for (int k = 0; k < used; ++k) pool->wait(k);
}

// `image` hands over the probability image when the weights are needed (the file entry point is still
// decoding it on another thread while the clouds are uploaded and indexed)
static void match_impl(const Super4PCSCloudView& segment, const Super4PCSCloudView& model_validation,
                       const Super4PCSCloudView& model_search,
                       const std::function<const unsigned short*(int*, int*, int)>& image,
                       std::pair<Eigen::Isometry3d, float>& bestHypothesis,
                       std::vector<std::pair<Eigen::Isometry3d, float> >& hypothesisSet,
                       std::map<std::vector<int>, std::vector<std::pair<int, int> > >& PPFMap,
                       Eigen::Matrix3f camIntrinsic, std::vector<int>& registered_points, bool want_rows) {
  const float delta = 0.005f;              // super4pcs_test.cc:20
  const int max_number_of_bases = 100;     // base.cc:290
  const int max_sampled_csets = 100;       // base.cc:1858
  pgp_ctx* ctx = nullptr;
  set_identity(bestHypothesis);
  // PGP_SHIM_VERBOSE: where a call's time goes (one PHASES line per call: tools/dropin_phases.py aggregates them)
  const bool verbose = getenv("PGP_SHIM_VERBOSE") != nullptr;
  auto t_mark = std::chrono::steady_clock::now();
  std::vector<std::pair<const char*, double> > phases;
  auto mark = [&](const char* name) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    phases.push_back(std::make_pair(name, std::chrono::duration<double, std::milli>(now - t_mark).count()));
    t_mark = now;
  };
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
  mark("copy+normals");

  // ---- init(): centring (base.cc:242-268)
  float cP[3], cQ[3];
  if (pgp_center(seg.xyz.data(), seg.n, qsearch.xyz.data(), qsearch.n, qval.xyz.data(), qval.n, cP, cQ) != PGP_OK) return;
  mark("centre");

  const auto t_start = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  // ---- device state: scene index, validation model, search model, pair-feature table
  ShimState local_state;
  ShimState& st = getenv("PGP_SHIM_NO_CACHE") ? local_state : shim_state();
  ObjectSlot* obj = nullptr;
  bool model_deferred = false;
  SlotLease lease;                                   // the object's slot is this call's until it returns
  std::unique_lock<std::mutex> single_lock;          // (the device group / single context of the modes that have one)
  // fingerprint of the caller's PPFMap (its two end entries): with its address and size, what an object is known by
  unsigned long long print = 0x9E3779B97F4A7C15ull;
  if (!PPFMap.empty()) {
    auto mix = [&print](long long v) { print = (print ^ (unsigned long long)v) * 0x100000001B3ull + (print >> 31); };
    const auto& a = *PPFMap.begin();
    const auto& b = *PPFMap.rbegin();
    for (int v : a.first) mix(v);
    mix((long long)a.second.size());
    if (!a.second.empty()) { mix(a.second.front().first); mix(a.second.front().second); mix(a.second.back().first); mix(a.second.back().second); }
    for (int v : b.first) mix(v);
    mix((long long)b.second.size());
    if (!b.second.empty()) { mix(b.second.front().first); mix(b.second.front().second); mix(b.second.back().first); mix(b.second.back().second); }
  }
  const int n_dev = shim_device_count();   // PGP_SHIM_DEVICES: 1 (default) | n | all
  // Exact distance ties go to the point the reference's kd-tree returns (pgp_set_exact_ties: its tree is then built
  // with the scene, +2-3 ms) -- by default only for a segment that HOLDS DUPLICATED POINTS, the one case in which
  // ties are not a one-in-10^7 event (a voxel-gridded or back-projected segment has none: ~10 us to find out).
  // PGP_SHIM_EXACT_TIES=1 / 0: always / never.
  bool exact_ties;
  if (const char* v = getenv("PGP_SHIM_EXACT_TIES")) exact_ties = atoi(v) != 0;
  else exact_ties = has_duplicate_points(seg.xyz.data(), seg.n);
  if (getenv("PGP_SHIM_VERBOSE"))
    std::cerr << "[libsuper4pcs shim] exact ties: " << (exact_ties ? "on" : "off") << std::endl;
  if (n_dev != 1) {
    // hypotheses sharded over the devices of the node (pgp_multi_*: RCCL all-reduce of the scores);
    // device 0's context of the group also serves the single-device steps
    if (&st != &local_state) single_lock = std::unique_lock<std::mutex>(st.single_mu);
    if (!st.group) SHIM_PGP(pgp_multi_create(&st.group, nullptr, n_dev));
    ctx = pgp_multi_context(st.group, 0);
    for (int d = 0; pgp_multi_context(st.group, d); ++d) SHIM_PGP(pgp_set_exact_ties(pgp_multi_context(st.group, d), exact_ties ? 1 : 0));
    SHIM_PGP(pgp_multi_set_scene(st.group, seg.xyz.data(), seg.nrm.data(), nullptr, seg.n, delta));
    SHIM_PGP(pgp_multi_set_model(st.group, qval.xyz.data(), qval.nrm.data(), qval.n));
  } else if (&st == &local_state) {
    if (!st.ctx) SHIM_PGP(pgp_create(&st.ctx, -1));
    ctx = st.ctx;
    SHIM_PGP(pgp_set_exact_ties(ctx, exact_ties ? 1 : 0));
    mark("ties+context");
    SHIM_PGP(pgp_set_scene(ctx, seg.xyz.data(), seg.nrm.data(), nullptr, seg.n, delta));
    mark("set_scene");
    SHIM_PGP(pgp_set_model(ctx, qval.xyz.data(), qval.nrm.data(), qval.n));
    mark("set_model");
  } else {
    // the object's own context: models and pair-feature table stay resident in it
    obj = st.acquire((const void*)&PPFMap, PPFMap.size(), print);
    lease.s = obj;
    if (!obj->ctx) SHIM_PGP(pgp_create(&obj->ctx, -1));
    ctx = obj->ctx;
    SHIM_PGP(pgp_set_exact_ties(ctx, exact_ties ? 1 : 0));
    mark("ties+context");
    SHIM_PGP(pgp_set_scene(ctx, seg.xyz.data(), seg.nrm.data(), nullptr, seg.n, delta));
    mark("set_scene");
    // (the validation model -- hashed, and sent again only when it changed -- is looked at while the base selection's
    //  kernel runs: ensure_model below; nothing before the verification reads it)
    model_deferred = true;
    mark("set_model");
  }
  if (obj) {
    static const std::vector<float> none;
    const unsigned long long sh = cloud_hash(qsearch.xyz, none);
    if (sh != obj->search_hash || !obj->map_loaded) {
      obj->search_hash = 0;
      obj->map_loaded = false;        // the table's pair lists index the search model: both go together
      SHIM_PGP(pgp_set_search_model(ctx, qsearch.xyz.data(), qsearch.n));
      obj->search_hash = sh;
    }
  } else {
    SHIM_PGP(pgp_set_search_model(ctx, qsearch.xyz.data(), qsearch.n));
  }
  mark("set_search_model");
  const bool map_stale = obj ? !obj->map_loaded
                             : (st.map_addr != (const void*)&PPFMap || st.map_size != PPFMap.size() || st.map_print != print ||
                                st.map_ctx != (const void*)ctx);
  if (map_stale) {
    // std::map<std::vector<int>, std::vector<std::pair<int,int>>> -> keys | counts | pairs
    static_assert(sizeof(std::pair<int, int>) == 2 * sizeof(int), "pair<int,int> must be two packed ints");
    std::vector<int> keys, counts, pairs;
    keys.reserve(4 * PPFMap.size());
    counts.reserve(PPFMap.size());
    for (const auto& kv : PPFMap) {
      if (kv.first.size() != 4) continue;   // computePPF always pushes four features
      keys.insert(keys.end(), kv.first.begin(), kv.first.end());
      counts.push_back((int)kv.second.size());
      const int* p = reinterpret_cast<const int*>(kv.second.data());
      pairs.insert(pairs.end(), p, p + 2 * kv.second.size());
    }
    SHIM_PGP(pgp_set_ppf_map(ctx, keys.data(), counts.data(), pairs.data(), (int)counts.size()));
    if (obj) {
      obj->map_loaded = true;
    } else {
      st.map_addr = &PPFMap;
      st.map_size = PPFMap.size();
      st.map_print = print;
      st.map_ctx = ctx;
    }
  }

  mark("ppf_map");
  // ---- per-point weights from the probability image (base.cc:317-340), once the image is there
  {
    int rows = 0, cols = 0;
    float K[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) K[3 * r + c] = camIntrinsic(r, c);
    // the last image row the segment's points fall on (whatever the image's size turns out to be): a decoder that
    // is still running stops there
    int row_lo = -1, row_hi = -1;
    if (want_rows)   // (only a decoder that is still running has a use for them: the file entry point)
      pgp_image_rows_needed(seg.xyz.data(), seg.n, cP, K, 1 << 24, 1 << 24, &row_lo, &row_hi);
    const unsigned short* prob_image = image(&rows, &cols, row_hi);
    if (prob_image && rows > 0 && cols > 0) {
      std::vector<float> prob(seg.n, 1.f);
      pgp_weights_from_image(seg.xyz.data(), seg.n, cP, K, prob_image, rows, cols, prob.data());
      if (st.group) SHIM_PGP(pgp_multi_set_scene_weights(st.group, prob.data(), seg.n));
      else SHIM_PGP(pgp_set_scene_weights(ctx, prob.data(), seg.n));
    }
  }
  mark("weights");
  const double ms_setup = ms_since(t_start);
  const auto t_bases = std::chrono::steady_clock::now();
  // ---- Step 1: base selection (base.cc:1831-1848): rounds of independent attempts, one launch each;
  // the four uniform variates of an attempt's std::discrete_distribution draws come from this engine
  unsigned seed = (unsigned)std::chrono::system_clock::now().time_since_epoch().count();
  if (const char* s = getenv("PGP_SHIM_SEED")) { seed = (unsigned)strtoul(s, nullptr, 10); srand(seed); }
  // rand() is one stream per PROCESS: calls that run side by side (getProbableTransformsSuper4PCSFrame) would interleave
  // their draws, so each of them -- and a single call under PGP_SHIM_PRIVATE_RAND=1, their checker -- draws its quad
  // samples from a generator of its own, seeded like the engine above
  unsigned private_rand = seed;
  const bool use_private_rand = t_private_rand || getenv("PGP_SHIM_PRIVATE_RAND") != nullptr;
  auto next_rand = [&]() -> int { return use_private_rand ? rand_r(&private_rand) : rand(); };
  std::default_random_engine generator(seed);
  const int attempts_per_round = 256;   // (the variates are drawn in attempt order, so the round size does not change which bases are taken;
                                        //  256 workgroups are one wave of the 256 CUs and the reference's 100 bases come out of ONE round: 0.41 -> 0.2 ms)
  int n_rounds = 0;
  std::vector<int> base_ids;       // n_bases x 4 (scene ids, TryQuadrilateral's order)
  std::vector<float> base_inv;     // n_bases x 2
  std::vector<int> base_rows;      // n_bases x 2: the table rows of pairs1 / pairs6 (they come home with the bases)
  {
    std::vector<double> u(4 * (size_t)attempts_per_round);
    std::vector<int> ids(4 * (size_t)attempts_per_round), status(attempts_per_round), rows(2 * (size_t)attempts_per_round);
    std::vector<float> inv(2 * (size_t)attempts_per_round);
    for (int round = 0; round < 20 && (int)(base_ids.size() / 4) < max_number_of_bases; ++round) {
      ++n_rounds;
      for (double& x : u) x = std::generate_canonical<double, 53>(generator);
      if (model_deferred) {
        // the first round in two halves: the ~0.1 ms kernel is queued, the host hashes the validation model meanwhile
        // (and sends it only if it changed: an upload would wait for the kernel), then collects the bases
        SHIM_PGP(pgp_select_bases_rows_begin(ctx, u.data(), attempts_per_round));
        model_deferred = false;
        const unsigned long long mh = cloud_hash(qval.xyz, qval.nrm);
        if (mh != obj->model_hash) {
          obj->model_hash = 0;
          SHIM_PGP(pgp_select_bases_rows_end(ctx, ids.data(), inv.data(), status.data(), rows.data()));
          SHIM_PGP(pgp_set_model(ctx, qval.xyz.data(), qval.nrm.data(), qval.n));
          obj->model_hash = mh;
        } else {
          SHIM_PGP(pgp_select_bases_rows_end(ctx, ids.data(), inv.data(), status.data(), rows.data()));
        }
      } else {
        SHIM_PGP(pgp_select_bases_rows(ctx, u.data(), attempts_per_round, ids.data(), inv.data(), status.data(), rows.data()));
      }
      for (int k = 0; k < attempts_per_round && (int)(base_ids.size() / 4) < max_number_of_bases; ++k) {
        if (status[k] != 1) continue;
        base_ids.insert(base_ids.end(), ids.begin() + 4 * k, ids.begin() + 4 * k + 4);
        base_inv.insert(base_inv.end(), inv.begin() + 2 * k, inv.begin() + 2 * k + 2);
        base_rows.insert(base_rows.end(), rows.begin() + 2 * k, rows.begin() + 2 * k + 2);
      }
    }
  }
  const int n_bases = (int)(base_ids.size() / 4);
  if (model_deferred) {   // (no round ran)
    const unsigned long long mh = cloud_hash(qval.xyz, qval.nrm);
    if (mh != obj->model_hash) {
      obj->model_hash = 0;
      SHIM_PGP(pgp_set_model(ctx, qval.xyz.data(), qval.nrm.data(), qval.n));
      obj->model_hash = mh;
    }
  }

  mark("base_selection");
  const double ms_bases = ms_since(t_bases);
  const auto t_cs = std::chrono::steady_clock::now();
  // ---- Step 2: congruent sets of every base in one pass (base.cc:1855-1874, 1929-1993), then the
  // reference's sampling of at most 100 quads per base -> (base, quad) picks
  std::vector<float> base_xyz(12 * (size_t)n_bases);
  for (int b = 0; b < n_bases; ++b)
    for (int k = 0; k < 4; ++k)
      for (int d = 0; d < 3; ++d) base_xyz[12 * (size_t)b + 3 * k + d] = seg.xyz[3 * (size_t)base_ids[4 * b + k] + d];
  std::vector<int> n_quads(n_bases > 0 ? n_bases : 1, 0);
  if (n_bases > 0)
    SHIM_PGP(pgp_find_congruent_batch_rows(ctx, base_ids.data(), base_xyz.data(), base_inv.data(), base_rows.data(), n_bases,
                                           delta, n_quads.data()));
  mark("congruent_sets");
  // The draw of at most 100 quads per base (base.cc:1858-1866).  By default it happens ON THE DEVICE, inside the call that fits
  // and verifies (pgp_congruent_batch_sample_fit_score_list: counter-based variates and Floyd's subset algorithm, one wave per
  // base, seeded from this call's seed -- the host draws nothing and uploads no picks; the device group's path draws the SAME
  // picks on the host, pgp_sample_quads).  PGP_SHIM_RAND=host: the host loop below, from the process's rand() (or, side by side
  // with other calls, a generator of the call's own) -- the streams of rounds 2-5, kept for their fixed-seed soaks.  The two
  // come out within 2 % of each other (the host's draw hides behind the device's key sort, the device's waits for it but
  // saves the picks' way up): profiles/r06_ab/device_draw.log.
  const char* rand_mode = getenv("PGP_SHIM_RAND");
  const bool host_draw = rand_mode && std::strcmp(rand_mode, "host") == 0;
  const bool device_draw = !st.group && !host_draw;
  // (64 bits of the call's seed: the clock's count, or PGP_SHIM_SEED -- the same picks call after call under a fixed seed)
  const unsigned long long draw_seed = getenv("PGP_SHIM_SEED") ? (unsigned long long)seed
                                                               : (unsigned long long)std::chrono::system_clock::now().time_since_epoch().count();
  std::vector<int> picks;   // (base, j) pairs
  picks.reserve(device_draw ? 0 : 2 * (size_t)n_bases * max_sampled_csets);
  std::vector<unsigned long long> seen;
  if (st.group && !host_draw && n_bases > 0) {   // (the device's draw, stated on the host: the group's fits take picks)
    picks.resize(2 * (size_t)n_bases * max_sampled_csets);
    int got = 0;
    SHIM_PGP(pgp_sample_quads(draw_seed, n_quads.data(), n_bases, max_sampled_csets, picks.data(), &got));
    picks.resize(2 * (size_t)got);
  }
  for (int b = 0; b < n_bases && host_draw; ++b) {
    const int nq = n_quads[b];
    if (nq < max_sampled_csets) {
      for (int j = 0; j < nq; ++j) { picks.push_back(b); picks.push_back(j); }
    } else if (nq <= (1 << 22)) {
      // 100 distinct random quads (the reference iterates an unordered_set; we take them in ascending order): the same
      // rand() stream and the same set as a std::set would hold, kept as a bitmap over the base's quads -- 100 bases x
      // 100 tree insertions were 0.17 ms of a 1.1 ms call
      const size_t nw = ((size_t)nq + 63) / 64;
      seen.assign(nw, 0ull);
      for (int n = 0; n < max_sampled_csets;) {
        const unsigned v = (unsigned)(next_rand() % nq);
        unsigned long long& w = seen[v >> 6];
        const unsigned long long m = 1ull << (v & 63);
        n += (w & m) ? 0 : 1;
        w |= m;
      }
      for (size_t k = 0; k < nw; ++k)
        for (unsigned long long w = seen[k]; w; w &= w - 1) { picks.push_back(b); picks.push_back((int)(k * 64 + (size_t)__builtin_ctzll(w))); }
    } else {
      std::set<int> chosen;
      while ((int)chosen.size() < max_sampled_csets) chosen.insert(next_rand() % nq);
      for (int j : chosen) { picks.push_back(b); picks.push_back(j); }
    }
  }
  mark("sample_quads");
  const double ms_cs = ms_since(t_cs);
  const auto t_fit = std::chrono::steady_clock::now();
  int n_pairs = (int)(picks.size() / 2);
  if (device_draw) {   // (known from the quad counts alone)
    n_pairs = 0;
    for (int b = 0; b < n_bases; ++b) n_pairs += n_quads[b] < max_sampled_csets ? n_quads[b] : max_sampled_csets;
  }
  if (!st.group) {
    // ---- single device: the fits never leave HBM -- fitted, verified (Step 3, base.cc:1885-1901, operMode = 1 ->
    // WeightedVerify) and walked there; scores and status come back, then the poses of the few hypotheses that are kept
    // hypothesisSet is the running-best list: its entries are decided on exact (reference-order) sums
    SHIM_PGP(pgp_set_exact_records(ctx, 1));
    // ONE call, ONE copy back: scores, the running-best walk over them, the poses it keeps, the best pose and the points it
    // registers (pgp_congruent_batch_fit_score_list).  A list beyond kListCap entries (never seen: a walk over n scores
    // keeps ~ln n of them) takes the calls this one replaces.
    const int kListCap = 256;
    std::vector<int> li(kListCap);
    std::vector<float> ls(kListCap), lT(16 * (size_t)kListCap);
    std::vector<double> lp(16 * (size_t)kListCap);
    int n_list = 0, n_pushed = 0, best_pick = -1, n_reg = 0;
    float best_lcp = 0.f, best_T[16];
    double best_pose[16];
    std::vector<int> reg(qval.n > 0 ? qval.n : 1);
    if (n_pairs > 0 && device_draw) {
      SHIM_PGP(pgp_congruent_batch_sample_fit_score_list(ctx, draw_seed, max_sampled_csets, base_ids.data(), cP, cQ, PGP_MODE_WEIGHTED, 30.f,
                                                         kListCap, &n_list, li.data(), ls.data(), lT.data(), lp.data(), &n_pushed,
                                                         &best_pick, &best_lcp, best_T, best_pose, reg.data(), &n_reg, nullptr, nullptr));
    } else if (n_pairs > 0) {
      SHIM_PGP(pgp_congruent_batch_fit_score_list(ctx, picks.data(), base_ids.data(), n_pairs, cP, cQ, PGP_MODE_WEIGHTED, 30.f, kListCap,
                                                  &n_list, li.data(), ls.data(), lT.data(), lp.data(), &n_pushed, &best_pick, &best_lcp,
                                                  best_T, best_pose, reg.data(), &n_reg));
    }
    const int n_h = n_pushed;
    mark("fit+score+records");
    if (verbose)
      std::cerr << "[libsuper4pcs shim] bases " << n_bases << ", congruent pairs " << n_pairs << ", transforms " << n_h
                << "; ms: setup " << ms_setup << ", base selection " << ms_bases << ", congruent sets " << ms_cs << std::endl;
    auto iso_of = [](const double* m16) {
      Eigen::Isometry3d iso;
      iso.matrix() = Eigen::Map<const Eigen::Matrix4d>(m16);
      return iso;
    };
    hypothesisSet.clear();   // the reference REPLACES the list by the running-best subsequence (allPose.clear(), base.cc:1903)
    if (n_list <= kListCap) {
      for (int k = 0; k < n_list; ++k) hypothesisSet.push_back(std::make_pair(iso_of(lp.data() + 16 * (size_t)k), ls[k]));  // base.cc:1903-1908
    } else {
      // the long way round: all scores, the walk on the host, the kept poses fetched
      if (device_draw && picks.size() != 2 * (size_t)n_pairs) {
        // (a list beyond kListCap records has never been seen; should it happen, the draw is repeated on the host -- the same
        //  picks: pgp_sample_quads is the device's draw)
        picks.resize(2 * (size_t)n_pairs);
        int got = 0;
        SHIM_PGP(pgp_sample_quads(draw_seed, n_quads.data(), n_bases, max_sampled_csets, picks.data(), &got));
      }
      std::vector<float> lcp_all(n_pairs);
      std::vector<int> status(n_pairs);
      SHIM_PGP(pgp_congruent_batch_fit_score(ctx, picks.data(), base_ids.data(), n_pairs, cP, cQ, PGP_MODE_WEIGHTED, 30.f,
                                             lcp_all.data(), status.data(), &best_pick, &best_lcp));
      std::vector<int> kept, selected(n_pairs);
      std::vector<float> lcp;
      for (int i = 0; i < n_pairs; ++i)
        if (status[i] == 1) {
          kept.push_back(i);
          lcp.push_back(lcp_all[i]);
        }
      int n_sel = 0;
      pgp_running_best(lcp.data(), (int)kept.size(), selected.data(), &n_sel);
      std::vector<int> want;
      for (int k = 0; k < n_sel; ++k) want.push_back(kept[selected[k]]);
      std::vector<float> Tf(16 * want.size() + 16);
      std::vector<double> posed(16 * want.size() + 16);
      if (!want.empty()) SHIM_PGP(pgp_congruent_batch_fetch(ctx, want.data(), (int)want.size(), Tf.data(), posed.data()));
      for (int k = 0; k < n_sel; ++k) hypothesisSet.push_back(std::make_pair(iso_of(posed.data() + 16 * (size_t)k), lcp[selected[k]]));
    }
    if (best_pick >= 0) {
      bestHypothesis = std::make_pair(iso_of(best_pose), best_lcp);
      registered_points.assign(reg.begin(), reg.begin() + n_reg);
    } else {
      std::cout << "returning identity" << std::endl;  // base.cc:1791-1794
    }
    mark("list+registered");
    if (verbose) {
      std::cerr << "[libsuper4pcs shim] PHASES";
      for (const auto& ph : phases) std::cerr << " " << ph.first << "=" << ph.second;
      std::cerr << " n_h=" << n_h << " rounds=" << n_rounds << " bases=" << n_bases << std::endl;
    }
    return;
  }
  // ---- several devices (PGP_SHIM_DEVICES): the fits come back and the group scores them
  std::vector<float> T((size_t)n_pairs * 16);
  std::vector<double> pose((size_t)n_pairs * 16);
  std::vector<int> status(n_pairs);
  if (n_pairs > 0)
    SHIM_PGP(pgp_congruent_batch_fit(ctx, picks.data(), base_ids.data(), n_pairs, cP, cQ, T.data(), pose.data(),
                                     status.data(), nullptr));
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
  mark("rigid_fits");
  const double ms_fit = ms_since(t_fit);
  if (getenv("PGP_SHIM_VERBOSE"))
    std::cerr << "[libsuper4pcs shim] bases " << n_bases << ", congruent pairs " << n_pairs
              << ", transforms " << n_h << "; ms: setup " << ms_setup << ", base selection " << ms_bases
              << ", congruent sets " << ms_cs << ", rigid fits " << ms_fit << std::endl;
  std::vector<float> lcp(n_h);
  int best = -1;
  float best_lcp = 0.f;
  // hypothesisSet is the running-best list: its entries are decided on exact (reference-order) sums
  SHIM_PGP(pgp_set_exact_records(ctx, 1));
  if (st.group)
    SHIM_PGP(pgp_multi_score_lcp(st.group, allT.data(), n_h, PGP_MODE_WEIGHTED, 30.f, lcp.data(), nullptr, &best,
                                 &best_lcp));
  else
    SHIM_PGP(pgp_score_lcp(ctx, allT.data(), n_h, PGP_MODE_WEIGHTED, 30.f, lcp.data(), nullptr, &best, &best_lcp));
  mark("score+records");
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
  mark("list+registered");
  if (verbose) {
    std::cerr << "[libsuper4pcs shim] PHASES";
    for (const auto& ph : phases) std::cerr << " " << ph.first << "=" << ph.second;
    std::cerr << " n_h=" << n_h << " rounds=" << n_rounds << " bases=" << n_bases << std::endl;
  }
}
