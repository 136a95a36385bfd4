// oracle/ref_io_harness.cc -- TEST INFRASTRUCTURE, never shipped, never on the product path.
//
// The file hand-off of the drop-in boundary, on the REFERENCE'S OWN reader: this translation unit
// is linked with S4/io/io.cc compiled unmodified where it lies under /root/reference (it needs
// only the vendored Eigen; its OpenCV use is behind USE_OPENCV, which the reference's build does
// not define for this target either) and calls IOManager::ReadObject + Utils::CleanInvalidNormals
// exactly as getProbableTransformsSuper4PCS does for each input cloud (S4/super4pcs_test.cc:58-89).
// tests/test_ply_reader.py compares shim/super4pcs_shim.cc's reader with it on PLY files in the
// layout pcl::io::savePLYFile writes, and tests/golden/make_golden.py stores its outputs.

#include <string>
#include <vector>

#include "io/io.h"
#include "utils/geometry.h"

extern "C" {

// Returns the number of points (<= cap written), or -1 when the reader fails.
// xyz / nrm receive Point3D::pos() / Point3D::normal() -- what the matcher reads afterwards.
int ref_read_cloud(const char* path, float* xyz, float* nrm, int cap) {
  IOManager iomananger;
  std::vector<Point3D> set1;
  std::vector<Eigen::Matrix2f> tex_coords1;
  std::vector<typename Point3D::VectorType> normals1;
  std::vector<tripple> tris1;
  std::vector<std::string> mtls1;
  if (!iomananger.ReadObject(path, set1, tex_coords1, normals1, tris1, mtls1)) return -1;
  if (tris1.size() == 0) Super4PCS::Utils::CleanInvalidNormals(set1, normals1);
  const int n = (int)set1.size();
  for (int i = 0; i < n && i < cap; ++i) {
    for (int k = 0; k < 3; ++k) {
      xyz[3 * i + k] = set1[i].pos()(k);
      nrm[3 * i + k] = set1[i].normal()(k);
    }
  }
  return n;
}

}  // extern "C"
