// examples/score_poses.cc -- the C ABI from a C++ host, with no Python in the loop.
//
//   g++ -O2 -std=c++11 -Iinclude examples/score_poses.cc -Lphysimglobalpose_amd -lpgp
//       -Wl,-rpath,$PWD/physimglobalpose_amd -Wl,-rpath-link,/opt/rocm/lib -o score_poses
//
// Builds a small scene (points on a sphere patch + clutter), takes every third scene point as the
// model, scores N candidate poses -- the identity among random ones -- in plain and weighted mode,
// checks what must hold exactly (the identity scores 1.0 in plain mode and wins, running-best rule,
// registered ids of the identity = the model's own scene points) and prints the throughput of the
// host-pointer call.  What a maintainer's own call site looks like is in INTEGRATION.md section 2.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
  const int n_hyp = argc > 1 ? std::atoi(argv[1]) : 2048;
  std::mt19937 gen(7);
  std::normal_distribution<float> normal(0.f, 1.f);
  std::uniform_real_distribution<float> uni(-1.f, 1.f);
  // scene: 6000 points on a sphere of radius 0.15 (outward normals) + 2000 clutter points
  std::vector<float> P, Pn, Pw;
  for (int i = 0; i < 8000; ++i) {
    float v[3] = {normal(gen), normal(gen), normal(gen)};
    const float n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const bool on_object = i < 6000;
    for (int k = 0; k < 3; ++k) {
      const float u = v[k] / n;
      P.push_back(on_object ? 0.15f * u : 0.4f * uni(gen));
      Pn.push_back(u);
    }
    Pw.push_back(on_object ? 1.0f : 0.2f);
  }
  // model: every third object point (so the identity maps every model point ONTO a scene point)
  std::vector<float> Q, Qn;
  std::vector<int> q_scene_id;
  for (int i = 0; i < 6000; i += 3) {
    for (int k = 0; k < 3; ++k) {
      Q.push_back(P[3 * i + k]);
      Qn.push_back(Pn[3 * i + k]);
    }
    q_scene_id.push_back(i);
  }
  const int nP = (int)Pw.size(), nQ = (int)q_scene_id.size();
  // hypotheses: column-major 4x4 images; index 17 is the identity, the rest random rigid motions
  std::vector<float> T((size_t)n_hyp * 16, 0.f);
  const int id_index = n_hyp > 17 ? 17 : 0;
  for (int h = 0; h < n_hyp; ++h) {
    float* t = &T[(size_t)h * 16];
    float ax[3] = {normal(gen), normal(gen), normal(gen)};
    const float an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    const float ang = h == id_index ? 0.f : 0.6f * uni(gen), c = std::cos(ang), s = std::sin(ang);
    const float x = ax[0] / an, y = ax[1] / an, z = ax[2] / an;
    const float R[9] = {c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s,
                        y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s,
                        z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)};
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) t[r + 4 * cc] = R[3 * r + cc];
    for (int r = 0; r < 3; ++r) t[12 + r] = h == id_index ? 0.f : 0.03f * uni(gen);
    t[15] = 1.f;
  }

  pgp_ctx* ctx = nullptr;
  CHECK(pgp_create(&ctx, -1));
  const float delta = 0.005f;
  CHECK(pgp_set_scene(ctx, P.data(), Pn.data(), Pw.data(), nP, delta));
  CHECK(pgp_set_model(ctx, Q.data(), Qn.data(), nQ));
  std::vector<float> lcp(n_hyp), wlcp(n_hyp);
  std::vector<int> counts(n_hyp);
  int best = -1, wbest = -1;
  float best_lcp = 0.f, wbest_lcp = 0.f;
  CHECK(pgp_score_lcp(ctx, T.data(), n_hyp, PGP_MODE_PLAIN, 30.f, lcp.data(), counts.data(), &best, &best_lcp));
  CHECK(pgp_score_lcp(ctx, T.data(), n_hyp, PGP_MODE_WEIGHTED, 30.f, wlcp.data(), nullptr, &wbest, &wbest_lcp));
  int failures = 0;
  if (counts[id_index] != nQ || lcp[id_index] != 1.0f) { std::printf("FAIL identity count %d of %d\n", counts[id_index], nQ); ++failures; }
  if (best != id_index || best_lcp != 1.0f) { std::printf("FAIL best %d (%g), expected %d\n", best, best_lcp, id_index); ++failures; }
  if (wbest != id_index) { std::printf("FAIL weighted best %d, expected %d\n", wbest, id_index); ++failures; }
  for (int h = 0; h < n_hyp; ++h)
    if (wlcp[h] > lcp[h] + 1e-6f) { std::printf("FAIL weighted score above plain at %d\n", h); ++failures; break; }
  std::vector<int> sel(n_hyp), reg(nQ);
  int n_sel = 0, n_reg = 0;
  CHECK(pgp_running_best(lcp.data(), n_hyp, sel.data(), &n_sel));
  if (n_sel < 1 || sel[n_sel - 1] != id_index) { std::printf("FAIL running best ends on %d\n", n_sel ? sel[n_sel - 1] : -1); ++failures; }
  CHECK(pgp_registered(ctx, &T[(size_t)id_index * 16], PGP_MODE_PLAIN, 30.f, reg.data(), &n_reg));
  if (n_reg != nQ) { std::printf("FAIL registered %d of %d\n", n_reg, nQ); ++failures; }
  for (int i = 0; i < n_reg && i < nQ; ++i)
    if (reg[i] != q_scene_id[i]) { std::printf("FAIL registered id %d: %d != %d\n", i, reg[i], q_scene_id[i]); ++failures; break; }

  const int reps = 50;
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r)
    CHECK(pgp_score_lcp(ctx, T.data(), n_hyp, PGP_MODE_WEIGHTED, 30.f, wlcp.data(), nullptr, &wbest, &wbest_lcp));
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("scene %d, model %d, %d hypotheses: identity at %d scores %.3f plain / %.3f weighted; "
              "%.2f M hypotheses/s through the host-pointer call\n",
              nP, nQ, n_hyp, id_index, lcp[id_index], wlcp[id_index], reps * (double)n_hyp / sec / 1e6);
  CHECK(pgp_destroy(ctx));
  std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
  return failures ? 1 : 0;
}
