// examples/six_objects.cc -- BASELINE configs[3] from a C++ host: six objects of a frame, their hypothesis lists sharded
// over the GPUs of the node as ONE flat (object, hypothesis) space, then the best poses of every object refined by ICP
// with the poses sharded the same way -- the node's object loop (PPE/data_layer/SceneCfg.cpp:376-406) and the
// refinement of an expansion's children (PPE/hypothesis_verification/mcts/UCTSearch.cpp:200-266) through the C ABI:
//
//   pgp_multi_create / pgp_multi_add_object / pgp_multi_set_object_scene / _model     clouds replicated on every member
//   pgp_multi_score_objects      every member scores its share, ONE all-reduce of {scores | counts}, arg-max per object
//   pgp_multi_icp_refine         the (object, pose) space block-partitioned, one launch per member, results gathered
//
// and checks both against single-context calls (pgp_score_lcp, pgp_icp_refine) bit for bit.  On a one-GPU machine run it
// with PGP_MULTI_EMULATE=8 (eight logical members on the one device); with several GPUs the group is every visible one.
//
//   g++ -O2 -std=c++11 -Iinclude examples/six_objects.cc -Lphysimglobalpose_amd -lpgp
//       -Wl,-rpath,$PWD/physimglobalpose_amd -Wl,-rpath-link,/opt/rocm/lib -o six_objects
//   ./six_objects [hypotheses in total = 65536] [poses refined per object = 32]
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
    if ((call) < PGP_OK) {                                                   \
      std::fprintf(stderr, "%s failed: %s\n", #call, pgp_last_error());      \
      return 1;                                                              \
    }                                                                        \
  } while (0)

struct Object {
  std::vector<float> scene, scene_n, scene_w, model, model_n, T;   // xyz triples; T: n x 16 column-major
  int n_scene = 0, n_model = 0, n_hyp = 0;
  float R[9], t[3];   // ground-truth pose model -> scene
};

static void rot_about(const float axis[3], float ang, float R[9]) {
  const float n = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  const float x = axis[0] / n, y = axis[1] / n, z = axis[2] / n, c = std::cos(ang), s = std::sin(ang), k = 1 - c;
  const float M[9] = {c + x * x * k, x * y * k - z * s, x * z * k + y * s, y * x * k + z * s, c + y * y * k,
                      y * z * k - x * s, z * x * k - y * s, z * y * k + x * s, c + z * z * k};
  std::memcpy(R, M, sizeof M);
}

// column-major 4x4 of {A * B} for rigid {R, t} pairs
static void compose(const float Ra[9], const float ta[3], const float Rb[9], const float tb[3], float* out16) {
  std::memset(out16, 0, 64);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      float v = 0.f;
      for (int k = 0; k < 3; ++k) v += Ra[3 * r + k] * Rb[3 * k + c];
      out16[r + 4 * c] = v;
    }
    float v = ta[r];
    for (int k = 0; k < 3; ++k) v += Ra[3 * r + k] * tb[k];
    out16[12 + r] = v;
  }
  out16[15] = 1.f;
}

int main(int argc, char** argv) {
  const int n_total = argc > 1 ? std::atoi(argv[1]) : 65536, k_refine = argc > 2 ? std::atoi(argv[2]) : 32, n_obj = 6;
  const int share[6] = {8, 6, 6, 4, 4, 4};   // 32nds of the total: 16384, 12288, 12288, 8192, 8192, 8192 at 65 536
  const float delta = 0.005f;
  std::mt19937 gen(23);
  std::normal_distribution<float> normal(0.f, 1.f);
  std::uniform_real_distribution<float> uni(-1.f, 1.f);
  std::vector<Object> objs(n_obj);
  for (int o = 0; o < n_obj; ++o) {
    Object& ob = objs[o];
    ob.n_model = 3000;
    const float ax[3] = {0.07f + 0.008f * o, 0.045f, 0.03f + 0.004f * o};
    for (int i = 0; i < ob.n_model; ++i) {   // an ellipsoid with outward (unnormalised-ellipsoid) normals
      float v[3] = {normal(gen), normal(gen), normal(gen)};
      const float n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      float g[3], gn = 0.f;
      for (int d = 0; d < 3; ++d) {
        ob.model.push_back(ax[d] * v[d] / n);
        g[d] = v[d] / n / ax[d];
        gn += g[d] * g[d];
      }
      for (int d = 0; d < 3; ++d) ob.model_n.push_back(g[d] / std::sqrt(gn));
    }
    const float axis[3] = {uni(gen), uni(gen), uni(gen) + 1.5f};
    rot_about(axis, 0.5f + 0.4f * o, ob.R);
    ob.t[0] = 0.12f * (o % 3) - 0.12f;
    ob.t[1] = 0.1f * (o / 3) - 0.05f;
    ob.t[2] = 0.7f;
    // segment: every second model point under the pose, 0.4 mm noise, weight 1; clutter up to 20 000 points, weight 0.2
    for (int i = 0; i < ob.n_model; i += 2) {
      for (int r = 0; r < 3; ++r) {
        float p = ob.t[r] + 0.0004f * normal(gen), q = 0.f;
        for (int c = 0; c < 3; ++c) {
          p += ob.R[3 * r + c] * ob.model[3 * i + c];
          q += ob.R[3 * r + c] * ob.model_n[3 * i + c];
        }
        ob.scene.push_back(p);
        ob.scene_n.push_back(q);
      }
      ob.scene_w.push_back(1.f);
    }
    while ((int)ob.scene_w.size() < 20000) {
      float v[3] = {normal(gen), normal(gen), normal(gen)};
      const float n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      for (int r = 0; r < 3; ++r) {
        ob.scene.push_back(ob.t[r] + 0.25f * uni(gen));
        ob.scene_n.push_back(v[r] / n);
      }
      ob.scene_w.push_back(0.2f);
    }
    ob.n_scene = (int)ob.scene_w.size();
    // hypotheses: the pose perturbed by up to ~6 degrees / 1 cm (a quarter of them close: 1 degree / 2 mm)
    ob.n_hyp = (int)((long long)n_total * share[o] / 32);
    ob.T.assign((size_t)ob.n_hyp * 16, 0.f);
    for (int h = 0; h < ob.n_hyp; ++h) {
      const bool close = (h % 4) == 1;
      const float a[3] = {normal(gen), normal(gen), normal(gen)};
      float dR[9];
      rot_about(a, (close ? 0.017f : 0.1f) * uni(gen), dR);
      const float s = close ? 0.002f : 0.01f, dt[3] = {s * uni(gen), s * uni(gen), s * uni(gen)};
      float Rt[9], tt[3];
      for (int r = 0; r < 3; ++r) {   // {R, t} * {dR, dt}: perturb in the model frame
        for (int c = 0; c < 3; ++c) {
          Rt[3 * r + c] = 0.f;
          for (int k = 0; k < 3; ++k) Rt[3 * r + c] += ob.R[3 * r + k] * dR[3 * k + c];
        }
        tt[r] = ob.t[r];
        for (int k = 0; k < 3; ++k) tt[r] += ob.R[3 * r + k] * dt[k];
      }
      const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z[3] = {0, 0, 0};
      compose(Rt, tt, I, z, &ob.T[(size_t)h * 16]);
    }
  }

  // ---- the group: every visible device (or PGP_MULTI_EMULATE members), six objects replicated on each
  pgp_multi* grp = nullptr;
  CHECK(pgp_multi_create(&grp, nullptr, 0));
  const int n_dev = pgp_multi_size(grp);
  for (int o = 1; o < n_obj; ++o) CHECK(pgp_multi_add_object(grp));
  for (int o = 0; o < n_obj; ++o) {
    Object& ob = objs[o];
    CHECK(pgp_multi_set_object_scene(grp, o, ob.scene.data(), ob.scene_n.data(), ob.scene_w.data(), ob.n_scene, delta));
    CHECK(pgp_multi_set_object_model(grp, o, ob.model.data(), ob.model_n.data(), ob.n_model));
  }
  std::vector<const float*> Ts(n_obj);
  std::vector<int> n_h(n_obj), off(n_obj + 1, 0);
  for (int o = 0; o < n_obj; ++o) {
    Ts[o] = objs[o].T.data();
    n_h[o] = objs[o].n_hyp;
    off[o + 1] = off[o] + n_h[o];
  }
  const int N = off[n_obj];
  std::vector<float> scores((size_t)N), best_score(n_obj);
  std::vector<int> counts((size_t)N), best(n_obj);
  CHECK(pgp_multi_score_objects(grp, Ts.data(), n_h.data(), n_obj, PGP_MODE_WEIGHTED, 30.f, scores.data(), counts.data(),
                                best.data(), best_score.data()));
  const int reps = 10;
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r)
    CHECK(pgp_multi_score_objects(grp, Ts.data(), n_h.data(), n_obj, PGP_MODE_WEIGHTED, 30.f, scores.data(), counts.data(),
                                  best.data(), best_score.data()));
  const double sec_score = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;

  // ---- the same through six single contexts on device 0
  int failures = 0;
  std::vector<pgp_ctx*> one(n_obj, nullptr);
  for (int o = 0; o < n_obj; ++o) {
    Object& ob = objs[o];
    CHECK(pgp_create(&one[o], 0));
    CHECK(pgp_set_scene(one[o], ob.scene.data(), ob.scene_n.data(), ob.scene_w.data(), ob.n_scene, delta));
    CHECK(pgp_set_model(one[o], ob.model.data(), ob.model_n.data(), ob.n_model));
    std::vector<float> s((size_t)ob.n_hyp);
    std::vector<int> c((size_t)ob.n_hyp);
    int b = -1;
    float bs = 0.f;
    CHECK(pgp_score_lcp(one[o], ob.T.data(), ob.n_hyp, PGP_MODE_WEIGHTED, 30.f, s.data(), c.data(), &b, &bs));
    if (std::memcmp(s.data(), &scores[(size_t)off[o]], (size_t)ob.n_hyp * 4) != 0 ||
        std::memcmp(c.data(), &counts[(size_t)off[o]], (size_t)ob.n_hyp * 4) != 0 || b != best[o] || bs != best_score[o]) {
      std::printf("FAIL object %d: the group's scores differ from a single context's (best %d / %d)\n", o, best[o], b);
      ++failures;
    }
    if (b < 0 || bs < 0.2f) {
      std::printf("FAIL object %d: no hypothesis registers the object (best %d, score %g)\n", o, b, bs);
      ++failures;
    }
  }

  // ---- ICP of the k best poses per object, the (object, pose) space sharded over the members.  ICP moves the SEGMENT
  // onto the model: the initial guess is the inverse pose (UCTState.cpp:184-185)
  std::vector<std::vector<float>> seg(n_obj), guess(n_obj), energy(n_obj);
  std::vector<std::vector<int>> iters(n_obj);
  std::vector<pgp_multi_icp_job> jobs(n_obj);
  for (int o = 0; o < n_obj; ++o) {
    Object& ob = objs[o];
    seg[o].assign(ob.scene.begin(), ob.scene.begin() + 3 * (ob.n_model / 2));   // the object's own points
    // k best by score (a partial selection sort is enough here)
    std::vector<int> order;
    std::vector<char> taken((size_t)ob.n_hyp, 0);
    for (int k = 0; k < k_refine && k < ob.n_hyp; ++k) {
      int arg = -1;
      for (int h = 0; h < ob.n_hyp; ++h)
        if (!taken[h] && (arg < 0 || scores[(size_t)off[o] + h] > scores[(size_t)off[o] + arg])) arg = h;
      taken[arg] = 1;
      order.push_back(arg);
    }
    guess[o].assign(order.size() * 16, 0.f);
    for (size_t k = 0; k < order.size(); ++k) {
      const float* T = &ob.T[(size_t)order[k] * 16];
      float* G = &guess[o][k * 16];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) G[r + 4 * c] = T[c + 4 * r];   // R^T
        G[12 + r] = -(T[0 + 4 * r] * T[12] + T[1 + 4 * r] * T[13] + T[2 + 4 * r] * T[14]);
      }
      G[15] = 1.f;
    }
    energy[o].assign(order.size(), 0.f);
    iters[o].assign(order.size(), 0);
    jobs[o] = pgp_multi_icp_job{seg[o].data(), ob.n_model / 2, ob.model.data(), ob.n_model, guess[o].data(), (int)order.size(),
                                energy[o].data(), iters[o].data()};
  }
  const std::vector<std::vector<float>> guess0 = guess;
  pgp_icp_params prm = {30, 0.9f, 0.f, 1.f};
  CHECK(pgp_multi_icp_refine(grp, jobs.data(), n_obj, &prm));   // first call: builds every target's index
  for (int o = 0; o < n_obj; ++o) guess[o] = guess0[o];
  const auto t1 = std::chrono::steady_clock::now();
  CHECK(pgp_multi_icp_refine(grp, jobs.data(), n_obj, &prm));
  const double sec_icp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  for (int o = 0; o < n_obj; ++o) {
    std::vector<float> T = guess0[o], e(energy[o].size());
    std::vector<int> it(iters[o].size());
    CHECK(pgp_icp_refine(one[o], seg[o].data(), objs[o].n_model / 2, objs[o].model.data(), objs[o].n_model, T.data(), jobs[o].n,
                         &prm, e.data(), it.data()));
    if (std::memcmp(T.data(), guess[o].data(), T.size() * 4) != 0 || std::memcmp(e.data(), energy[o].data(), e.size() * 4) != 0 ||
        std::memcmp(it.data(), iters[o].data(), it.size() * 4) != 0) {
      std::printf("FAIL object %d: the group's refined poses differ from pgp_icp_refine's\n", o);
      ++failures;
    }
    // a refined pose must explain the segment to the noise level: sqrt(mean d2) of the kept 90 % well below 1 mm
    for (size_t k = 0; k < e.size(); ++k)
      if (!(std::sqrt(e[k]) < 0.001f)) {
        std::printf("FAIL object %d pose %zu: rms %.5f m after ICP\n", o, k, std::sqrt(e[k]));
        ++failures;
        break;
      }
  }
  std::printf("%d member(s), 6 objects (20 000-point segments, 3 000-point models), %d hypotheses: %.3f ms per scoring call = %.1f M "
              "hypotheses/s from host pointers; ICP of %d poses per object: %.3f ms\n",
              n_dev, N, 1e3 * sec_score, N / sec_score / 1e6, k_refine, 1e3 * sec_icp);
  for (int o = 0; o < n_obj; ++o) CHECK(pgp_destroy(one[o]));
  CHECK(pgp_multi_destroy(grp));
  std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
  return failures ? 1 : 0;
}
