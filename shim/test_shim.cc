// shim/test_shim.cc -- stands in for the node side of the boundary
// (PPE/hypothesis_generation/ObjectPoseCandidateSet.cpp:53-68 + PPE/data_layer/Objects.cpp:31-49):
// loads PPFMap.txt the way Objects::readPPFMap does, calls getProbableTransformsSuper4PCS exactly
// as CongruentSetMatching::generate does, and prints the outputs for tests/test_shim_gpu.py.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

#include "super4pcs_shim.h"  // the in-memory overload + readers (SHIM_TEST_INMEMORY=1 uses them)

void getProbableTransformsSuper4PCS(std::string input1, std::string input2, std::string input3,
                                    std::pair<Eigen::Isometry3d, float>& bestHypothesis,
                                    std::vector<std::pair<Eigen::Isometry3d, float> >& hypothesisSet,
                                    std::string probImagePath,
                                    std::map<std::vector<int>, std::vector<std::pair<int, int> > >& PPFMap,
                                    int max_count_ppf, Eigen::Matrix3f camIntrinsic, std::string objName,
                                    std::string scenePath, std::vector<int>& registered_points);

int main(int argc, char** argv) {
  if (argc < 10) {
    std::fprintf(stderr, "usage: test_shim segment.ply model_val.ply model_search.ply prob.png PPFMap.txt fx fy cx cy\n");
    return 2;
  }
  std::map<std::vector<int>, std::vector<std::pair<int, int> > > PPFMap;
  {
    std::ifstream f(argv[5]);
    std::vector<int> key(4);
    int count;
    while (f >> key[0] >> key[1] >> key[2] >> key[3] >> count) {
      std::vector<std::pair<int, int> > pairs;
      for (int i = 0; i < count; ++i) {
        int a, b;
        f >> a >> b;
        pairs.push_back(std::make_pair(a, b));
      }
      PPFMap.insert(std::make_pair(key, pairs));
    }
  }
  Eigen::Matrix3f K = Eigen::Matrix3f::Identity();
  K(0, 0) = std::atof(argv[6]); K(1, 1) = std::atof(argv[7]); K(0, 2) = std::atof(argv[8]); K(1, 2) = std::atof(argv[9]);
  std::pair<Eigen::Isometry3d, float> best;
  best.first.matrix().setIdentity();
  best.second = 0;
  std::vector<std::pair<Eigen::Isometry3d, float> > hyps;
  std::vector<int> registered;
  // SHIM_TEST_REPEAT=n: the same call n times in one process (bench.py's drop_in row: the first call
  // pays context creation and the code-object load); every call starts from fresh output containers
  const int repeat = std::getenv("SHIM_TEST_REPEAT") ? std::max(1, std::atoi(std::getenv("SHIM_TEST_REPEAT"))) : 1;
  std::vector<double> elapsed;
  // SHIM_TEST_TWO_OBJECTS=1: a second object with its own PPFMap (a copy: another address) alternates with the first,
  // as the node's object loop does -- each keeps its table and models resident in its own context
  std::map<std::vector<int>, std::vector<std::pair<int, int> > > PPFMap2;
  const bool two = std::getenv("SHIM_TEST_TWO_OBJECTS") != nullptr;
  if (two) PPFMap2 = PPFMap;
  std::map<std::vector<int>, std::vector<std::pair<int, int> > >* maps[2] = {&PPFMap, two ? &PPFMap2 : &PPFMap};
  // SHIM_TEST_CHECK_SAME=1 (with PGP_SHIM_SEED): every call must return what the first one did -- best score and pose,
  // the list's scores, the registered points -- bit for bit (a soak for races between the calls' asynchronous parts)
  int n_same = std::getenv("SHIM_TEST_CHECK_SAME") ? 0 : -1;
  std::vector<double> first_sig;
  // SHIM_TEST_FRAME=n: n objects of one frame (the same clouds, each object with a pair-feature table of its own) through
  // getProbableTransformsSuper4PCSFrame, `repeat` frames; before them every object once through the single call (under
  // PGP_SHIM_PRIVATE_RAND, set by the test): FRAME_SAME counts the jobs whose outputs equal that call's bit for bit
  if (const char* fr = std::getenv("SHIM_TEST_FRAME")) {
    const int n_obj = std::max(1, std::atoi(fr));
    std::vector<float> sx, sn, vx, vn, qx, qn;
    std::vector<unsigned short> px;
    int rows = 0, cols = 0;
    if (!super4pcs_shim_read_ply(argv[1], sx, sn) || !super4pcs_shim_read_ply(argv[2], vx, vn) || !super4pcs_shim_read_ply(argv[3], qx, qn)) return 3;
    const bool have = super4pcs_shim_read_png16(argv[4], px, rows, cols);
    const Super4PCSCloudView sv = {sx.data(), sn.data(), (int)(sx.size() / 3)};
    const Super4PCSCloudView vv = {vx.data(), vn.data(), (int)(vx.size() / 3)};
    const Super4PCSCloudView qv = {qx.data(), qn.data(), (int)(qx.size() / 3)};
    auto signature = [](const std::pair<Eigen::Isometry3d, float>& b, const std::vector<std::pair<Eigen::Isometry3d, float> >& h,
                        const std::vector<int>& reg) {
      std::vector<double> sig;
      sig.push_back(b.second);
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) sig.push_back(b.first.matrix()(r, c));
      for (size_t i = 0; i < h.size(); ++i) sig.push_back(h[i].second);
      sig.push_back((double)reg.size());
      for (int v : reg) sig.push_back((double)v);
      return sig;
    };
    std::vector<std::map<std::vector<int>, std::vector<std::pair<int, int> > > > tables((size_t)n_obj, PPFMap);
    std::vector<Super4PCSJob> jobs((size_t)n_obj);
    for (int j = 0; j < n_obj; ++j) {
      jobs[j].segment = sv;
      jobs[j].model_validation = vv;
      jobs[j].model_search = qv;
      jobs[j].prob_image = have ? px.data() : nullptr;
      jobs[j].rows = rows;
      jobs[j].cols = cols;
      jobs[j].PPFMap = std::getenv("SHIM_TEST_FRAME_SAME_TABLE") && j == 0 ? &PPFMap : &tables[j];   // (probe knob)
      if (std::getenv("SHIM_TEST_FRAME_ONE_OBJECT")) jobs[j].PPFMap = &tables[0];   // every job the SAME object: the calls take turns
      jobs[j].camIntrinsic = K;
    }
    // (the single call they are compared with runs AFTER the frames: its context and streams would otherwise sit beside the
    //  workers' -- a process has four hardware queues, profiles/r05_ab/hardware_queues.log -- which a node that only ever
    //  calls the frame entry point does not have)
    std::vector<std::vector<double> > got_all;
    const int gap_us = std::getenv("SHIM_TEST_FRAME_GAP_US") ? std::atoi(std::getenv("SHIM_TEST_FRAME_GAP_US")) : 0;
    for (int rep = 0; rep < repeat; ++rep) {
      if (gap_us > 0) {   // the node's own work between two frames, as a busy wait
        const auto g0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g0).count() < gap_us) {}
      }
      const auto t0 = std::chrono::steady_clock::now();
      if (std::getenv("SHIM_TEST_FRAME_THREADS")) {
        // the reference's own commented-out form (SceneCfg.cpp:377,402-403): a fresh std::thread per object around the single
        // call, joined at the end of the loop -- the objects' contexts are the process's, so the new threads find them
        std::vector<std::thread> th;
        for (int j = 0; j < n_obj; ++j)
          th.emplace_back([&, j] {
            if (!std::strcmp(std::getenv("SHIM_TEST_FRAME_THREADS"), "files"))   // through the files, exactly the reference's signature
              getProbableTransformsSuper4PCS(argv[1], argv[2], argv[3], jobs[j].bestHypothesis, jobs[j].hypothesisSet, argv[4], *jobs[j].PPFMap, 0, K,
                                             "synthetic_object", "./", jobs[j].registered_points);
            else
              getProbableTransformsSuper4PCS(sv, vv, qv, jobs[j].prob_image, rows, cols, jobs[j].bestHypothesis, jobs[j].hypothesisSet, *jobs[j].PPFMap, K,
                                             jobs[j].registered_points);
          });
        for (std::thread& t : th) t.join();
      } else if (std::getenv("SHIM_TEST_FRAME_DIRECT"))   // (probe knob: the single call in this loop instead)
        getProbableTransformsSuper4PCS(sv, vv, qv, jobs[0].prob_image, rows, cols, jobs[0].bestHypothesis, jobs[0].hypothesisSet, *jobs[0].PPFMap, K,
                                       jobs[0].registered_points);
      else
      getProbableTransformsSuper4PCSFrame(jobs.data(), n_obj);
      elapsed.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      for (int j = 0; j < n_obj; ++j)
        got_all.push_back(jobs[j].failed ? std::vector<double>() : signature(jobs[j].bestHypothesis, jobs[j].hypothesisSet, jobs[j].registered_points));
    }
    getProbableTransformsSuper4PCS(sv, vv, qv, have ? px.data() : nullptr, rows, cols, best, hyps, PPFMap, K, registered);
    const std::vector<double> want = signature(best, hyps, registered);
    int same = 0;
    const int total = (int)got_all.size();
    for (const std::vector<double>& got : got_all)
      if (got.size() == want.size() && std::memcmp(got.data(), want.data(), got.size() * sizeof(double)) == 0) ++same;
    std::printf("FRAME_SAME %d of %d\n", same, total);
    std::printf("FRAME_MS");
    for (double e : elapsed) std::printf(" %.3f", e);
    std::printf("\nBEST_SCORE %.9g\n", best.second);
    return 0;
  }
  for (int rep = 0; rep < repeat; ++rep) {
    std::map<std::vector<int>, std::vector<std::pair<int, int> > >& PPFMapCall = *maps[rep & 1];
    best.first.matrix().setIdentity();
    best.second = 0;
    hyps.clear();
    registered.clear();
    if (std::getenv("PGP_SHIM_SEED")) std::srand((unsigned)std::atoi(std::getenv("PGP_SHIM_SEED")));
    const auto t0 = std::chrono::steady_clock::now();
    if (std::getenv("SHIM_TEST_INMEMORY")) {
      std::vector<float> sx, sn, vx, vn, qx, qn;
      std::vector<unsigned short> px;
      int rows = 0, cols = 0;
      if (!super4pcs_shim_read_ply(argv[1], sx, sn) || !super4pcs_shim_read_ply(argv[2], vx, vn) ||
          !super4pcs_shim_read_ply(argv[3], qx, qn)) return 3;
      const bool have = super4pcs_shim_read_png16(argv[4], px, rows, cols);
      const Super4PCSCloudView s = {sx.data(), sn.data(), (int)(sx.size() / 3)};
      const Super4PCSCloudView v = {vx.data(), vn.data(), (int)(vx.size() / 3)};
      const Super4PCSCloudView q = {qx.data(), qn.data(), (int)(qx.size() / 3)};
      const auto t1 = std::chrono::steady_clock::now();   // the in-memory caller holds its clouds already
      getProbableTransformsSuper4PCS(s, v, q, have ? px.data() : nullptr, rows, cols, best, hyps, PPFMapCall, K, registered);
      elapsed.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    } else {
      getProbableTransformsSuper4PCS(argv[1], argv[2], argv[3], best, hyps, argv[4], PPFMapCall, 0, K, "synthetic_object",
                                     "./", registered);
      elapsed.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    if (n_same >= 0) {
      std::vector<double> sig;
      sig.push_back(best.second);
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) sig.push_back(best.first.matrix()(r, c));
      for (size_t i = 0; i < hyps.size(); ++i) sig.push_back(hyps[i].second);
      sig.push_back((double)registered.size());
      for (int v : registered) sig.push_back((double)v);
      if (rep == 0) first_sig = sig;
      if (sig.size() == first_sig.size() && std::memcmp(sig.data(), first_sig.data(), sig.size() * sizeof(double)) == 0) ++n_same;
    }
  }
  if (n_same >= 0) std::printf("SAME_AS_FIRST %d of %d\n", n_same, repeat);
  std::printf("ELAPSED_MS");
  for (double e : elapsed) std::printf(" %.3f", e);
  std::printf("\n");
  std::printf("PPFMAP %zu\n", PPFMap.size());
  std::printf("BEST_SCORE %.9g\n", best.second);
  std::printf("BEST_POSE");
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) std::printf(" %.17g", best.first.matrix()(r, c));
  std::printf("\nHYPOTHESES %zu", hyps.size());
  for (size_t i = 0; i < hyps.size(); ++i) std::printf(" %.9g", hyps[i].second);
  std::printf("\nREGISTERED %zu\n", registered.size());
  return 0;
}
