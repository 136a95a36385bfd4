// examples/refine_objects.cc -- the resident chain of the node's per-object loop (PPE/data_layer/SceneCfg.cpp:379-402:
// for every object: hypotheses -> verification -> best poses -> ICP) through the C ABI, from a C++ host:
//   per object   pgp_score_lcp_device        scores of its hypotheses, in HBM
//                pgp_select_top_device       the k best, rigidly inverted (UCTState.cpp:184-185), in HBM
//   all objects  pgp_icp_refine_multi_device ONE launch refines the k x n_objects poses, each against its own model
// and checks that every refined transform equals, bit for bit, what pgp_icp_refine (host pointers, one object at a
// time) returns for the same guesses.  Only the HIP runtime API is used here (no kernels of its own):
//
//   g++ -O2 -std=c++11 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/refine_objects.cc
//       -Lphysimglobalpose_amd -lpgp -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/physimglobalpose_amd -Wl,-rpath,/opt/rocm/lib
#include <hip/hip_runtime_api.h>

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
#define HIP(call)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));               \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

struct Object {
  pgp_ctx* ctx = nullptr;
  std::vector<float> scene, scene_n, scene_w, model, model_n, T;   // xyz triples; T: n_hyp x 16 column-major
  float *d_T = nullptr, *d_scores = nullptr, *d_top = nullptr, *d_seg4 = nullptr, *d_model4 = nullptr;
  int *d_idx = nullptr, *d_n = nullptr, *d_iters = nullptr;
  int n_seg = 0;
};

static void rot_about(const float axis[3], float ang, float R[9]) {
  const float n = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  const float x = axis[0] / n, y = axis[1] / n, z = axis[2] / n, c = std::cos(ang), s = std::sin(ang), k = 1 - c;
  const float M[9] = {c + x * x * k, x * y * k - z * s, x * z * k + y * s, y * x * k + z * s, c + y * y * k,
                      y * z * k - x * s, z * x * k - y * s, z * y * k + x * s, c + z * z * k};
  std::memcpy(R, M, sizeof M);
}

int main(int argc, char** argv) {
  const int n_hyp = argc > 1 ? std::atoi(argv[1]) : 1024, k_top = 32, n_obj = 3;
  std::mt19937 gen(11);
  std::normal_distribution<float> normal(0.f, 1.f);
  std::uniform_real_distribution<float> uni(-1.f, 1.f);
  std::vector<Object> objs(n_obj);
  for (int o = 0; o < n_obj; ++o) {
    Object& ob = objs[o];
    // model: 1500 points on an ellipsoid; scene: the model under a pose + noise (weight 1) + clutter (weight 0.2)
    const float ax[3] = {0.06f + 0.01f * o, 0.04f, 0.03f};
    for (int i = 0; i < 1500; ++i) {
      float v[3] = {normal(gen), normal(gen), normal(gen)};
      const float n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      for (int d = 0; d < 3; ++d) {
        ob.model.push_back(ax[d] * v[d] / n);
        ob.model_n.push_back(v[d] / n);
      }
    }
    float R[9];
    const float axis[3] = {uni(gen), uni(gen), uni(gen) + 1.5f};
    rot_about(axis, 0.7f + 0.3f * o, R);
    const float t[3] = {0.1f * o, -0.05f, 0.6f};
    for (int i = 0; i < 1500; i += 2) {
      for (int r = 0; r < 3; ++r) {
        float p = t[r] + 0.0004f * normal(gen), q = 0.f;
        for (int c = 0; c < 3; ++c) {
          p += R[3 * r + c] * ob.model[3 * i + c];
          q += R[3 * r + c] * ob.model_n[3 * i + c];
        }
        ob.scene.push_back(p);
        ob.scene_n.push_back(q);
      }
      ob.scene_w.push_back(1.f);
    }
    ob.n_seg = (int)ob.scene_w.size();
    for (int i = 0; i < 600; ++i) {
      for (int d = 0; d < 3; ++d) {
        ob.scene.push_back(t[d] + 0.2f * uni(gen));
        ob.scene_n.push_back(d == 2 ? 1.f : 0.f);
      }
      ob.scene_w.push_back(0.2f);
    }
    // hypotheses (model -> scene): the true pose perturbed by up to ~3 degrees / 4 mm, and random ones
    for (int h = 0; h < n_hyp; ++h) {
      float Rp[9], Rh[9];
      const float a2[3] = {uni(gen), uni(gen), uni(gen)};
      rot_about(a2, h % 4 == 0 ? 0.05f * uni(gen) : 3.f * uni(gen), Rp);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rh[3 * r + c] = Rp[3 * r] * R[c] + Rp[3 * r + 1] * R[3 + c] + Rp[3 * r + 2] * R[6 + c];
      float M[16] = {0};
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) M[4 * c + r] = Rh[3 * r + c];
        M[12 + r] = t[r] + (h % 4 == 0 ? 0.004f : 0.1f) * uni(gen);
      }
      M[15] = 1.f;
      ob.T.insert(ob.T.end(), M, M + 16);
    }
    CHECK(pgp_create(&ob.ctx, 0));
    CHECK(pgp_set_scene(ob.ctx, ob.scene.data(), ob.scene_n.data(), ob.scene_w.data(), (int)ob.scene_w.size(), 0.005f));
    CHECK(pgp_set_model(ob.ctx, ob.model.data(), ob.model_n.data(), 1500));
    CHECK(pgp_reserve(ob.ctx, n_hyp));
    HIP(hipMalloc((void**)&ob.d_T, (size_t)n_hyp * 64));
    HIP(hipMalloc((void**)&ob.d_scores, (size_t)n_hyp * 4));
    HIP(hipMalloc((void**)&ob.d_top, (size_t)k_top * 64));
    HIP(hipMalloc((void**)&ob.d_idx, (size_t)k_top * 4));
    HIP(hipMalloc((void**)&ob.d_n, 4));
    HIP(hipMalloc((void**)&ob.d_iters, (size_t)k_top * 4));
    HIP(hipMalloc((void**)&ob.d_seg4, (size_t)ob.n_seg * 16));
    HIP(hipMalloc((void**)&ob.d_model4, (size_t)1500 * 16));
    HIP(hipMemcpy(ob.d_T, ob.T.data(), (size_t)n_hyp * 64, hipMemcpyHostToDevice));
    std::vector<float> s4((size_t)ob.n_seg * 4, 0.f), m4((size_t)1500 * 4, 0.f);
    for (int i = 0; i < ob.n_seg; ++i) std::memcpy(&s4[4 * i], &ob.scene[3 * i], 12);
    for (int i = 0; i < 1500; ++i) std::memcpy(&m4[4 * i], &ob.model[3 * i], 12);
    HIP(hipMemcpy(ob.d_seg4, s4.data(), s4.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(ob.d_model4, m4.data(), m4.size() * 4, hipMemcpyHostToDevice));
    CHECK(pgp_icp_target_token(ob.ctx, 1000 + o));   // the model does not change between calls
  }
  hipStream_t st;
  HIP(hipStreamCreate(&st));
  // ---- the chain, nothing but 4 bytes per object leaves the device before the refined poses do
  std::vector<pgp_icp_job> jobs(n_obj);
  for (int o = 0; o < n_obj; ++o) {
    Object& ob = objs[o];
    CHECK(pgp_score_lcp_device(ob.ctx, ob.d_T, n_hyp, PGP_MODE_WEIGHTED, 30.f, ob.d_scores, nullptr, nullptr, st));
    CHECK(pgp_select_top_device(ob.ctx, ob.d_T, ob.d_scores, n_hyp, k_top, 1, ob.d_top, ob.d_idx, ob.d_n, st));
    jobs[o] = pgp_icp_job{ob.ctx, ob.d_seg4, ob.n_seg, ob.d_model4, 1500, ob.d_top, k_top, nullptr, ob.d_iters};
  }
  // the guesses as selected (before refinement), for the host-pointer comparison below
  HIP(hipStreamSynchronize(st));
  std::vector<std::vector<float> > guess(n_obj, std::vector<float>((size_t)k_top * 16));
  for (int o = 0; o < n_obj; ++o) HIP(hipMemcpy(guess[o].data(), objs[o].d_top, (size_t)k_top * 64, hipMemcpyDeviceToHost));
  const pgp_icp_params prm = {30, 0.9f, 0.f, 1.f};
  CHECK(pgp_icp_refine_multi_device(jobs.data(), n_obj, &prm, st));
  HIP(hipStreamSynchronize(st));
  int bad = 0;
  for (int o = 0; o < n_obj; ++o) {
    Object& ob = objs[o];
    std::vector<float> got((size_t)k_top * 16), want = guess[o];
    std::vector<int> it_got(k_top), it_want(k_top);
    int n_pos = 0;
    HIP(hipMemcpy(got.data(), ob.d_top, got.size() * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(it_got.data(), ob.d_iters, (size_t)k_top * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(&n_pos, ob.d_n, 4, hipMemcpyDeviceToHost));
    CHECK(pgp_icp_refine(ob.ctx, ob.scene.data(), ob.n_seg, ob.model.data(), 1500, want.data(), k_top, &prm, nullptr, it_want.data()));
    const bool same = std::memcmp(got.data(), want.data(), got.size() * 4) == 0 && it_got == it_want;
    // the best refined pose maps the segment onto the model: its translation part is finite and the pose moved
    std::printf("object %d: %d of the top %d hypotheses scored > 0, multi-target launch == per-object host call: %s (first pose: %d iterations)\n",
                o, n_pos, k_top, same ? "yes" : "NO", it_got[0]);
    if (!same || n_pos != k_top) ++bad;
  }
  for (Object& ob : objs) pgp_destroy(ob.ctx);
  std::printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
