// csrc/kd_ties.hip -- the reference's kd-tree, kept ONLY to settle exact distance ties the way it does.
//
// KdTree::doQueryRestrictedClosestIndex (S4/accelerators/kdtree.h:394-459) accepts a candidate when
// `sqdist <= cl_dist` (:424): among scene points at EXACTLY the same float distance from the query, the one it
// visits LAST wins, and the visiting order follows the tree -- leaves in the order of the descent (the query's side
// of every split plane first, :440-451), points inside a leaf in the order the build's in-place partition left them
// (split(), :522-538).  The grid index of this library has no such order; it breaks a tie by the lowest scene
// index.  Ties are rare on real data (about one query in 10^7 with two or more neighbours)
// except for duplicated scene points, which always tie.
//
// With pgp_set_exact_ties(ctx, 1) the scene set-up also builds the reference's tree on the host (same midpoint
// splits, same partition, same leaf sizes: createTree, :560-641) and keeps {nodes, points in tree order, original
// indices} on the device.  The scoring paths then notice when two different candidates share the minimal distance
// and ask the tree, by the reference's own descent, which one it would have returned (kd_restricted_nn in
// pgp_internal.h) -- a handful of queries per million, so the tree is never on the hot path.

#include "pgp_internal.h"

#include <cfloat>
#include <cstring>
#include <vector>

namespace pgp {

namespace {

constexpr unsigned kPointsPerCell = 64;   // KD_POINT_PER_CELL, kdtree.h:63
constexpr int kMaxDepth = 32;             // KD_MAX_DEPTH, kdtree.h:60

struct Builder {
  std::vector<float> pts;     // xyz, reordered in place
  std::vector<int> idx;       // original index of the point at each position
  std::vector<int4> nodes;    // inner: {bits(split value), first child, dim, parent << 1}; leaf: {start, size, 0, parent << 1 | 1}
  std::vector<int> parent;    // of every node (the root: itself): the device descent keeps no stack, it climbs

  void swap_pts(int a, int b) {
    for (int k = 0; k < 3; ++k) std::swap(pts[3 * (size_t)a + k], pts[3 * (size_t)b + k]);
    std::swap(idx[a], idx[b]);
  }

  // the two-pointer partition of kdtree.h:522-538: points below the split value to the front
  int split(int start, int end, int dim, float v) {
    int l = start, r = end - 1;
    for (; l < r; ++l, --r) {
      while (l < end && pts[3 * (size_t)l + dim] < v) l++;
      while (r >= start && pts[3 * (size_t)r + dim] >= v) r--;
      if (l > r) break;
      swap_pts(l, r);
    }
    if (l >= end) return l;   // (the reference reads one past the range here)
    return pts[3 * (size_t)l + dim] < v ? l + 1 : l;
  }

  void leaf(int node, int start, int size) { nodes[node] = make_int4(start, size, 0, (parent[node] << 1) | 1); }

  // kdtree.h:560-641: split the largest extent of the node's bounding box at its middle
  void create(int node, int start, int end, int level) {
    float mn[3] = {FLT_MAX / 2, FLT_MAX / 2, FLT_MAX / 2}, mx[3] = {-FLT_MAX / 2, -FLT_MAX / 2, -FLT_MAX / 2};
    for (int i = start; i < end; ++i)
      for (int k = 0; k < 3; ++k) {
        const float v = pts[3 * (size_t)i + k];
        if (v < mn[k]) mn[k] = v;   // a NaN coordinate changes neither bound (bbox.h:73-75)
        if (v > mx[k]) mx[k] = v;
      }
    float diag[3];
    for (int k = 0; k < 3; ++k) diag[k] = 0.5f * (mx[k] - mn[k]);
    int dim = 0;   // the first strict maximum (Eigen maxCoeff)
    for (int k = 1; k < 3; ++k)
      if (diag[k] > diag[dim]) dim = k;
    const float v = mn[dim] + ((mx[dim] - mn[dim]) / 2.0f);   // AlignedBox::center(), bbox.h:88-89
    const int mid = split(start, end, dim, v);
    const int first = (int)nodes.size();
    nodes.push_back(make_int4(0, 0, 0, 0));
    nodes.push_back(make_int4(0, 0, 0, 0));
    parent.push_back(node);
    parent.push_back(node);
    nodes[node] = make_int4(__builtin_bit_cast(int, v), first, dim, parent[node] << 1);
    if ((unsigned)(mid - start) <= kPointsPerCell || level >= kMaxDepth) leaf(first, start, mid - start);
    else create(first, start, mid, level + 1);
    if ((unsigned)(end - mid) <= kPointsPerCell || level >= kMaxDepth) leaf(first + 1, mid, end - mid);
    else create(first + 1, mid, end, level + 1);
  }
};

}  // namespace

// Builds the reference's tree over the n scene points (host) and uploads it.  ctx->kd_valid afterwards.
int build_kd_ties(pgp_ctx* ctx, const float* h_xyz, int n) {
  ctx->kd_valid = false;
  Builder b;
  b.pts.assign(h_xyz, h_xyz + 3 * (size_t)n);
  b.idx.resize((size_t)n);
  for (int i = 0; i < n; ++i) b.idx[i] = i;
  b.nodes.reserve(n > 0 ? 4 * (size_t)n / kPointsPerCell + 8 : 8);
  b.nodes.push_back(make_int4(0, 0, 0, 0));   // the root is an inner node even over an empty cloud (kdtree.h:362-367)
  b.parent.push_back(0);
  b.create(0, 0, n, 1);
  std::vector<float4> hp((size_t)std::max(n, 1));
  for (int i = 0; i < n; ++i)
    hp[i] = make_float4(b.pts[3 * (size_t)i], b.pts[3 * (size_t)i + 1], b.pts[3 * (size_t)i + 2], __builtin_bit_cast(float, b.idx[i]));
  int rc;
  if ((rc = ctx->d_kd_nodes.ensure(b.nodes.size() * sizeof(int4))) != PGP_OK) return rc;
  if ((rc = ctx->d_kd_pts.ensure(hp.size() * sizeof(float4))) != PGP_OK) return rc;
  PGP_HIP(hipMemcpyAsync(ctx->d_kd_nodes.p, b.nodes.data(), b.nodes.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
  PGP_HIP(hipMemcpyAsync(ctx->d_kd_pts.p, hp.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  PGP_HIP(hipStreamSynchronize(ctx->stream));   // the host vectors go out of scope
  ctx->kd_n_nodes = (int)b.nodes.size();
  ctx->kd_valid = true;
  return PGP_OK;
}

}  // namespace pgp
