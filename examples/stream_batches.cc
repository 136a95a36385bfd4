// examples/stream_batches.cc -- list after list through the device group WITHOUT a host wait per list, from a C++ host.
//
//   g++ -O2 -std=c++11 -Iinclude examples/stream_batches.cc -Lphysimglobalpose_amd -lpgp
//       -Wl,-rpath,$PWD/physimglobalpose_amd -Wl,-rpath-link,/opt/rocm/lib -o stream_batches
//
// The verification loop of base.cc:1885-1901 over the hypothesis lists of successive objects / expansions: every list is left
// resident on the devices of a group (pgp_multi_upload_slot), one verification step per list is QUEUED (pgp_multi_enqueue_slot:
// every member scores its slice, the RCCL all-reduce of step i runs on a second stream under the scoring of step i + 1, member
// 0's arg-max behind that) and ONE pgp_multi_collect completes everything -- what bench.py's N > 1 headline times.  The program
// checks the streamed result against the synchronous call (pgp_multi_score_lcp) and against one context, bit for bit, and
// prints the group's make-up (pgp_multi_get_info: devices, ranks in the communicator).
//
//   ./stream_batches [hypotheses per list] [lists]        PGP_MULTI_EMULATE=4 ./stream_batches   (four members on one GPU)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "pgp.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    if ((call) != PGP_OK) {                                                  \
      std::fprintf(stderr, "%s failed: %s\n", #call, pgp_last_error());      \
      return 1;                                                              \
    }                                                                        \
  } while (0)

int main(int argc, char** argv) {
  const int n_h = argc > 1 ? std::atoi(argv[1]) : 2048;
  const int n_lists = argc > 2 ? std::atoi(argv[2]) : 6;
  if (n_h < 1 || n_lists < 1 || n_lists > 16) {
    std::fprintf(stderr, "usage: stream_batches [hypotheses per list >= 1] [lists 1..16]\n");
    return 2;
  }
  std::mt19937 gen(11);
  std::normal_distribution<float> normal(0.f, 1.f);
  // scene: 12 000 points on a sphere patch + clutter, outward normals; model: every fourth scene point of the object
  std::vector<float> P, Pn, Pw, Q, Qn;
  for (int i = 0; i < 12000; ++i) {
    float v[3] = {normal(gen), normal(gen), normal(gen)};
    const float n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const bool on_object = i < 9000;
    for (int k = 0; k < 3; ++k) {
      const float u = v[k] / n;
      P.push_back(on_object ? 0.15f * u : 0.4f * v[k]);
      Pn.push_back(u);
    }
    Pw.push_back(on_object ? 1.f : 0.2f);
    if (on_object && i % 4 == 0)
      for (int k = 0; k < 3; ++k) {
        Q.push_back(P[3 * (size_t)i + k]);
        Qn.push_back(Pn[3 * (size_t)i + k]);
      }
  }
  const int nP = 12000, nQ = (int)(Q.size() / 3);
  // lists of 4x4 column-major transforms: rotations about z of ~17 degrees + centimetre translations, the identity somewhere in
  // list 0 (poses within a millimetre of each other would all tie at the top and make every step take the exact near-tie
  // settlement -- correct, and a hundred times slower than the common case this example is about)
  std::vector<std::vector<float> > T((size_t)n_lists, std::vector<float>((size_t)n_h * 16, 0.f));
  for (int l = 0; l < n_lists; ++l)
    for (int h = 0; h < n_h; ++h) {
      float* m = &T[(size_t)l][16 * (size_t)h];
      const float a = (l == 0 && h == n_h / 3) ? 0.f : 0.3f * normal(gen);
      m[0] = std::cos(a); m[1] = std::sin(a); m[4] = -std::sin(a); m[5] = std::cos(a); m[10] = 1.f; m[15] = 1.f;
      for (int k = 0; k < 3; ++k) m[12 + k] = (l == 0 && h == n_h / 3) ? 0.f : 0.02f * normal(gen);
    }

  pgp_multi* grp = nullptr;
  CHECK(pgp_multi_create(&grp, nullptr, 0));            // every visible device (PGP_MULTI_EMULATE=n: n members on device 0)
  pgp_multi_info inf;
  CHECK(pgp_multi_get_info(grp, &inf));
  std::printf("group: %d member(s) in this process, world %d, RCCL ranks %d%s\n", inf.n_local, inf.world, inf.rccl_ranks,
              inf.emulated ? " (emulated on one device)" : "");
  CHECK(pgp_multi_set_scene(grp, P.data(), Pn.data(), Pw.data(), nP, 0.005f));
  CHECK(pgp_multi_set_model(grp, Q.data(), Qn.data(), nQ));
  for (int l = 0; l < n_lists; ++l) CHECK(pgp_multi_upload_slot(grp, l, T[(size_t)l].data(), n_h));

  std::vector<float> s_stream((size_t)n_h), s_sync((size_t)n_h), s_one((size_t)n_h);
  std::vector<int> c_stream((size_t)n_h), c_sync((size_t)n_h), c_one((size_t)n_h);
  int b_stream = -2, b_sync = -2, b_one = -2;
  float bs_stream = 0.f, bs_sync = 0.f, bs_one = 0.f;
  // warm up, then: every list once, no host wait in between; list 0 last, so that its result is the one collected
  for (int l = 0; l < n_lists; ++l) CHECK(pgp_multi_enqueue_slot(grp, l, PGP_MODE_WEIGHTED, 30.f));
  CHECK(pgp_multi_collect(grp, s_stream.data(), c_stream.data(), &b_stream, &bs_stream));
  const int rounds = 20;
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < rounds; ++r)
    for (int l = n_lists - 1; l >= 0; --l) CHECK(pgp_multi_enqueue_slot(grp, l, PGP_MODE_WEIGHTED, 30.f));
  CHECK(pgp_multi_collect(grp, s_stream.data(), c_stream.data(), &b_stream, &bs_stream));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("streamed: %d steps of %d hypotheses in %.3f ms = %.2f M hypotheses/s (one collect at the end)\n", rounds * n_lists, n_h,
              ms, 1e-3 * rounds * n_lists * n_h / ms);
  // the synchronous call on the same list, and one context
  CHECK(pgp_multi_score_lcp(grp, T[0].data(), n_h, PGP_MODE_WEIGHTED, 30.f, s_sync.data(), c_sync.data(), &b_sync, &bs_sync));
  pgp_ctx* one = nullptr;
  CHECK(pgp_create(&one, inf.devices[0]));
  CHECK(pgp_set_scene(one, P.data(), Pn.data(), Pw.data(), nP, 0.005f));
  CHECK(pgp_set_model(one, Q.data(), Qn.data(), nQ));
  CHECK(pgp_score_lcp(one, T[0].data(), n_h, PGP_MODE_WEIGHTED, 30.f, s_one.data(), c_one.data(), &b_one, &bs_one));
  const bool same = std::memcmp(s_stream.data(), s_sync.data(), (size_t)n_h * 4) == 0 && std::memcmp(s_stream.data(), s_one.data(), (size_t)n_h * 4) == 0 &&
                    std::memcmp(c_stream.data(), c_sync.data(), (size_t)n_h * 4) == 0 && std::memcmp(c_stream.data(), c_one.data(), (size_t)n_h * 4) == 0 &&
                    b_stream == b_sync && b_stream == b_one && bs_stream == bs_sync && bs_stream == bs_one;
  std::printf("best %d (score %.6f); streamed == synchronous == one context: %s\n", b_stream, bs_stream, same ? "yes" : "NO");
  CHECK(pgp_multi_get_info(grp, &inf));
  std::printf("exchanges issued: %lld\n", inf.exchanges);
  pgp_destroy(one);
  pgp_multi_destroy(grp);
  // (the identity itself does not win in weighted mode: model normals that DUPLICATE scene normals exactly make acos(dot > 1) a
  //  NaN for a fifth of the points, which the reference's gate rejects -- SURVEY hazard 4; a pose a hair off it wins)
  if (!same || b_stream < 0 || !(bs_stream > 0.5f)) return 1;
  std::printf("OK\n");
  return 0;
}
