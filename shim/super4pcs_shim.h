// shim/super4pcs_shim.h -- what libsuper4pcs.so (this repository's build) exports beyond the
// reference's own entry point.  The node needs none of this to link: the file-based
// getProbableTransformsSuper4PCS keeps the reference's exact signature
// (PPE/hypothesis_generation/ObjectPoseCandidateSet.cpp:5-9).
//
// the reference serialises three clouds to ASCII PLY and a 16-bit PNG per object
// per request (ObjectPoseCandidateSet.cpp:53-60) only to parse them again on the other side of
// the call (super4pcs_test.cc:58-80, base.cc:317).  The overload below takes the same data as
// plain arrays; a node that adopts it skips the disk round trip, everything else is unchanged.
#pragma once

#include <map>
#include <string>
#include <utility>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

struct Super4PCSCloudView {
  const float* xyz;      // n x 3, row-major (pclSegment / pclModel / pclModelSampled points)
  const float* normals;  // n x 3 or nullptr
  int n;
};

// Same outputs as the file-based entry point.  prob_image: rows x cols 16-bit probability image
// (value / 10000 = weight, base.cc:317-324) or nullptr for unit weights.
void getProbableTransformsSuper4PCS(const Super4PCSCloudView& segment, const Super4PCSCloudView& model_validation,
                                    const Super4PCSCloudView& model_search, const unsigned short* prob_image,
                                    int rows, int cols,
                                    std::pair<Eigen::Isometry3d, float>& bestHypothesis,
                                    std::vector<std::pair<Eigen::Isometry3d, float> >& hypothesisSet,
                                    std::map<std::vector<int>, std::vector<std::pair<int, int> > >& PPFMap,
                                    Eigen::Matrix3f camIntrinsic, std::vector<int>& registered_points);

// The objects of ONE frame side by side (the node matches them one after the other, SceneCfg.cpp:379-402): the jobs run on
// threads the library keeps, each object's context, models and pair-feature table stay resident from frame to frame (as they
// do for the single calls, from whichever thread those come), and the jobs' device work overlaps.  Inputs and outputs of a job are those of the overload above; `failed` is set where the
// single call would have thrown (the other jobs of the frame still complete).  Results equal the jobs called one by one,
// except that each call samples its quads from a generator of its own instead of the process-wide rand() (with
// PGP_SHIM_SEED fixed: the single calls under PGP_SHIM_PRIVATE_RAND=1, bit for bit).  PGP_SHIM_FRAME_SERIAL=1: job by job.
struct Super4PCSJob {
  Super4PCSCloudView segment, model_validation, model_search;
  const unsigned short* prob_image;   // rows x cols, or nullptr
  int rows, cols;
  std::map<std::vector<int>, std::vector<std::pair<int, int> > >* PPFMap;
  Eigen::Matrix3f camIntrinsic;
  std::pair<Eigen::Isometry3d, float> bestHypothesis;                      // out
  std::vector<std::pair<Eigen::Isometry3d, float> > hypothesisSet;         // out
  std::vector<int> registered_points;                                      // out
  bool failed;                                                             // out
};
void getProbableTransformsSuper4PCSFrame(Super4PCSJob* jobs, int n_jobs);

// The readers the file-based entry point uses (what pcl::io::savePLYFile / cv::imwrite produce).
bool super4pcs_shim_read_ply(const std::string& path, std::vector<float>& xyz, std::vector<float>& normals);
bool super4pcs_shim_read_png16(const std::string& path, std::vector<unsigned short>& pixels, int& rows, int& cols);
