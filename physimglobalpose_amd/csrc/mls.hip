// csrc/mls.hip -- moving-least-squares smoothing + normals of a segment cloud on gfx950.
//
// Replaces pcl::MovingLeastSquares<PointXYZRGB, PointXYZRGBNormal> as the node configures it
// (PPE/segmentation/Segmentation.cpp:239-246): setComputeNormals(true), setPolynomialFit(true) (order 2,
// PCL's default), setSearchRadius(0.02), kd-tree radius search, no upsampling.  PCL is not vendored
// (SURVEY 8c): this follows PCL 1.7's published MovingLeastSquares::computeMLSPointNormal step by step, in
// double precision where PCL uses double:
//   neighbours   = points with squared distance STRICTLY below (float)(radius^2) (FLANN radius search; float
//                  accumulation in x, y, z order, the point itself included); fewer than 3 -> the point is
//                  dropped (performProcessing)
//   plane        = centroid (pcl::compute3DCentroid) + eigenvector of the smallest eigenvalue of the
//                  un-normalised covariance (pcl::computeCovarianceMatrix, pcl::eigen33: scaled matrix,
//                  trigonometric roots, largest of the three row cross products)
//   projection   = the query point dropped onto that plane; curvature = |lambda_min / trace|
//   polynomial   = if there are >= 6 neighbours: heights f over the plane in the frame (u, v) =
//                  (normal x unitOrthogonal(normal), unitOrthogonal(normal)), weights exp(-d^2 / radius^2) with
//                  d^2 (rounded to float, as PCL stores it) measured from the PROJECTED point, terms
//                  (1, v, v^2, u, uv, u^2); normal equations P W P^T c = P W f solved by Cholesky (LLT);
//                  point += c[0] * normal; normal = plane normal - c[3] * u - c[1] * v (NOT re-normalised in
//                  PCL 1.7; the node normalises and flips all normals afterwards,
//                  PPE/hypothesis_generation/ObjectPoseCandidateSet.cpp:39-51)
// Differences from PCL that cannot be closed without its binary: PCL visits the neighbours in order of
// distance, this kernel cell by cell (then by point index) -- the double sums differ in their last bits --
// and libm's exp / atan2 / cos are the device's.  An independent numpy implementation (float64, eigh,
// weighted lstsq: tests/golden/make_mls_golden.py) agrees to 1e-6.
//
// Mapping: one thread per point; the cloud is sorted by cell (edge >= radius) with a STABLE radix sort, so a
// point's neighbours are the nine x-rows of three cells around it, each one contiguous range of the sorted
// array found by binary search -- no dense grid, any extent; the ranges are kept in registers and walked three
// times (centroid, covariance, polynomial sums).  Double precision VALU; a segment is 10^3..10^4 points.

#include "pgp_internal.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <cmath>

namespace pgp {

namespace {

struct MlsDesc {
  float ox, oy, oz, inv_h;
  int nx, ny, nz;
  float r2;          // (float)(radius * radius): FLANN's threshold
  double gauss;      // radius * radius: sqr_gauss_param_
};

__device__ __forceinline__ bool mls_cell(const MlsDesc& g, float x, float y, float z, int* cx, int* cy, int* cz) {
  const float fx = (x - g.ox) * g.inv_h, fy = (y - g.oy) * g.inv_h, fz = (z - g.oz) * g.inv_h;
  if (!(fx >= 0.f && fx < (float)g.nx && fy >= 0.f && fy < (float)g.ny && fz >= 0.f && fz < (float)g.nz)) return false;
  *cx = (int)fx;
  *cy = (int)fy;
  *cz = (int)fz;
  return true;
}

__global__ __launch_bounds__(256) void mls_keys(const float* __restrict__ xyz, int n, MlsDesc g,
                                                unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  // non-finite points sort to the end and are nobody's neighbour
  unsigned long long k = ~0ull;
  if (mls_cell(g, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], &cx, &cy, &cz))
    k = ((unsigned long long)cz * (unsigned long long)g.ny + (unsigned long long)cy) * (unsigned long long)g.nx + (unsigned long long)cx;
  keys[i] = k;
  vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void mls_gather(const float* __restrict__ xyz, const uint32_t* __restrict__ order, int n,
                                                  float4* __restrict__ sorted) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t i = order[k];
  sorted[k] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __uint_as_float(i));
}

__device__ __forceinline__ int lower_bound_key(const unsigned long long* __restrict__ keys, int n, unsigned long long v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// pcl::computeRoots2 / computeRoots (pcl/common/impl/eigen.hpp), double
__device__ __forceinline__ void roots2(double b, double c, double r[3]) {
  r[0] = 0.0;
  double d = b * b - 4.0 * c;
  if (d < 0.0) d = 0.0;   // no real roots: set to zero (numerical noise)
  const double sd = sqrt(d);
  r[2] = 0.5 * (b + sd);
  r[1] = 0.5 * (b - sd);
}

__device__ void roots3(const double m[6], double r[3]) {   // m = {xx, xy, xz, yy, yz, zz}
  const double c0 = m[0] * m[3] * m[5] + 2.0 * m[1] * m[2] * m[4] - m[0] * m[4] * m[4] - m[3] * m[2] * m[2] -
                    m[5] * m[1] * m[1];
  const double c1 = m[0] * m[3] - m[1] * m[1] + m[0] * m[5] - m[2] * m[2] + m[3] * m[5] - m[4] * m[4];
  const double c2 = m[0] + m[3] + m[5];
  if (fabs(c0) < DBL_EPSILON) {   // one root is 0 -> quadratic equation
    roots2(c2, c1, r);
    return;
  }
  const double s_inv3 = 1.0 / 3.0, s_sqrt3 = sqrt(3.0);
  const double c2_over_3 = c2 * s_inv3;
  double a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0) a_over_3 = 0.0;
  const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
  double q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0) q = 0.0;
  const double rho = sqrt(-a_over_3);
  const double theta = atan2(sqrt(-q), half_b) * s_inv3;
  const double cos_theta = cos(theta), sin_theta = sin(theta);
  r[0] = c2_over_3 + 2.0 * rho * cos_theta;
  r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  // sort in increasing order (PCL's three conditional swaps)
  if (r[0] >= r[1]) { const double t = r[0]; r[0] = r[1]; r[1] = t; }
  if (r[1] >= r[2]) {
    const double t = r[1]; r[1] = r[2]; r[2] = t;
    if (r[0] >= r[1]) { const double u = r[0]; r[0] = r[1]; r[1] = u; }
  }
  if (r[0] <= 0.0) roots2(c2, c1, r);   // a symmetric positive semi-definite matrix has no negative eigenvalue
}

// pcl::eigen33 (smallest eigenvalue + its eigenvector), double
__device__ void eigen33_smallest(const double cov[6], double* eval, double evec[3]) {
  double scale = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) scale = fmax(scale, fabs(cov[k]));
  if (scale <= DBL_MIN) scale = 1.0;
  double m[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) m[k] = cov[k] / scale;
  double r[3];
  roots3(m, r);
  *eval = r[0] * scale;
  const double a00 = m[0] - r[0], a11 = m[3] - r[0], a22 = m[5] - r[0];
  const double r0[3] = {a00, m[1], m[2]}, r1[3] = {m[1], a11, m[4]}, r2[3] = {m[2], m[4], a22};
  const double v1[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
  const double v2[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
  const double v3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  const double l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  const double l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  const double l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const double* v;
  double l;
  if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
  else { v = v3; l = l3; }
  const double s = sqrt(l);
  evec[0] = v[0] / s;
  evec[1] = v[1] / s;
  evec[2] = v[2] / s;
}

// Eigen::MatrixBase<Vector3d>::unitOrthogonal()
__device__ __forceinline__ void unit_orthogonal(const double n[3], double o[3]) {
  const double prec = 1e-12;   // NumTraits<double>::dummy_precision()
  const bool x_small = fabs(n[0]) <= fabs(n[2]) * prec, y_small = fabs(n[1]) <= fabs(n[2]) * prec;
  if (!x_small || !y_small) {
    const double inv = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1]);
    o[0] = -n[1] * inv;
    o[1] = n[0] * inv;
    o[2] = 0.0;
  } else {
    const double inv = 1.0 / sqrt(n[1] * n[1] + n[2] * n[2]);
    o[0] = 0.0;
    o[1] = -n[2] * inv;
    o[2] = n[1] * inv;
  }
}

// in-place Cholesky solve of the 6 x 6 system (lower triangle of A used), Eigen LLT semantics: a
// non-positive pivot yields NaN (sqrt of a negative) which the caller tests with isfinite(c[0])
__device__ void llt_solve6(double A[6][6], double b[6]) {
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
    for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    const double l = sqrt(d);
    A[j][j] = l;
    for (int i = j + 1; i < 6; ++i) {
      double s = A[i][j];
      for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k];
      A[i][j] = s / l;
    }
  }
  for (int i = 0; i < 6; ++i) {   // L y = b
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[i][k] * b[k];
    b[i] = s / A[i][i];
  }
  for (int i = 5; i >= 0; --i) {  // L^T x = y
    double s = b[i];
    for (int k = i + 1; k < 6; ++k) s -= A[k][i] * b[k];
    b[i] = s / A[i][i];
  }
}

struct Ranges {
  int b[9], e[9];
};

template <class F>
__device__ __forceinline__ void for_neighbours(const Ranges& rg, const float4* __restrict__ sorted, float px, float py,
                                               float pz, float r2, F f) {
#pragma unroll 1
  for (int r = 0; r < 9; ++r) {
    for (int k = rg.b[r]; k < rg.e[r]; ++k) {
      const float4 q = sorted[k];
      // FLANN L2_Simple: float accumulation in x, y, z order; strict <
      const float dx = __fsub_rn(px, q.x), dy = __fsub_rn(py, q.y), dz = __fsub_rn(pz, q.z);
      const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (d2 < r2) f(q);
    }
  }
}

__global__ __launch_bounds__(128) void mls_points(const float* __restrict__ xyz, int n, MlsDesc g,
                                                  const unsigned long long* __restrict__ keys,
                                                  const float4* __restrict__ sorted, int n_sorted,
                                                  float4* __restrict__ out_pn,     // {x, y, z, curvature}
                                                  float4* __restrict__ out_nrm,    // {nx, ny, nz, -}
                                                  uint32_t* __restrict__ valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
  valid[i] = 0u;
  int cx, cy, cz;
  if (!mls_cell(g, px, py, pz, &cx, &cy, &cz)) return;
  Ranges rg;
  {
    int r = 0;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy, ++r) {
        const int zz = cz + dz, yy = cy + dy;
        rg.b[r] = rg.e[r] = 0;
        if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny) continue;
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
        const unsigned long long row = ((unsigned long long)zz * (unsigned long long)g.ny + (unsigned long long)yy) * (unsigned long long)g.nx;
        rg.b[r] = lower_bound_key(keys, n_sorted, row + (unsigned long long)x0);
        rg.e[r] = lower_bound_key(keys, n_sorted, row + (unsigned long long)x1 + 1ull);
      }
  }
  // 1. centroid (pcl::compute3DCentroid, double)
  int cnt = 0;
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for_neighbours(rg, sorted, px, py, pz, g.r2, [&](const float4& q) {
    ++cnt;
    sx += (double)q.x;
    sy += (double)q.y;
    sz += (double)q.z;
  });
  if (cnt < 3) return;   // MovingLeastSquares::performProcessing skips the point
  const double mx = sx / (double)cnt, my = sy / (double)cnt, mz = sz / (double)cnt;
  // 2. covariance about the centroid, un-normalised (pcl::computeCovarianceMatrix)
  double cov[6] = {0, 0, 0, 0, 0, 0};
  for_neighbours(rg, sorted, px, py, pz, g.r2, [&](const float4& q) {
    const double dx = (double)q.x - mx, dy = (double)q.y - my, dz = (double)q.z - mz;
    cov[0] += dx * dx;
    cov[1] += dx * dy;
    cov[2] += dx * dz;
    cov[3] += dy * dy;
    cov[4] += dy * dz;
    cov[5] += dz * dz;
  });
  double eval, nrm[3];
  eigen33_smallest(cov, &eval, nrm);
  const double d_plane = -(nrm[0] * mx + nrm[1] * my + nrm[2] * mz);   // model_coefficients[3]
  double pt[3] = {(double)px, (double)py, (double)pz};
  const double dist = pt[0] * nrm[0] + pt[1] * nrm[1] + pt[2] * nrm[2] + d_plane;
  pt[0] -= dist * nrm[0];
  pt[1] -= dist * nrm[1];
  pt[2] -= dist * nrm[2];
  float curvature = (float)(cov[0] + cov[3] + cov[5]);
  if (curvature != 0.f) curvature = fabsf((float)(eval / (double)curvature));
  double out_n[3] = {nrm[0], nrm[1], nrm[2]};
  // 3. bivariate polynomial of order 2 over the plane
  if (cnt >= 6) {
    double v_ax[3], u_ax[3];
    unit_orthogonal(nrm, v_ax);
    u_ax[0] = nrm[1] * v_ax[2] - nrm[2] * v_ax[1];
    u_ax[1] = nrm[2] * v_ax[0] - nrm[0] * v_ax[2];
    u_ax[2] = nrm[0] * v_ax[1] - nrm[1] * v_ax[0];
    double A[6][6], b[6];
    for (int r = 0; r < 6; ++r) {
      b[r] = 0.0;
      for (int c = 0; c < 6; ++c) A[r][c] = 0.0;
    }
    for_neighbours(rg, sorted, px, py, pz, g.r2, [&](const float4& q) {
      const double dx = (double)q.x - pt[0], dy = (double)q.y - pt[1], dz = (double)q.z - pt[2];
      const float sqr = (float)(dx * dx + dy * dy + dz * dz);   // nn_sqr_dists is a float vector
      const double w = exp(-(double)sqr / g.gauss);
      const double u = dx * u_ax[0] + dy * u_ax[1] + dz * u_ax[2];
      const double v = dx * v_ax[0] + dy * v_ax[1] + dz * v_ax[2];
      const double f = dx * nrm[0] + dy * nrm[1] + dz * nrm[2];
      const double term[6] = {1.0, v, v * v, u, u * v, u * u};
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double wr = w * term[r];
        b[r] += wr * f;
#pragma unroll
        for (int c = 0; c <= r; ++c) A[r][c] += wr * term[c];
      }
    });
    llt_solve6(A, b);
    if (isfinite(b[0])) {
      pt[0] += b[0] * nrm[0];
      pt[1] += b[0] * nrm[1];
      pt[2] += b[0] * nrm[2];
      // partial derivatives at (0, 0): c[order + 1] along u, c[1] along v
      out_n[0] = nrm[0] - b[3] * u_ax[0] - b[1] * v_ax[0];
      out_n[1] = nrm[1] - b[3] * u_ax[1] - b[1] * v_ax[1];
      out_n[2] = nrm[2] - b[3] * u_ax[2] - b[1] * v_ax[2];
    }
  }
  out_pn[i] = make_float4((float)pt[0], (float)pt[1], (float)pt[2], curvature);
  out_nrm[i] = make_float4((float)out_n[0], (float)out_n[1], (float)out_n[2], 0.f);
  valid[i] = 1u;
}

__global__ __launch_bounds__(256) void mls_compact(const float4* __restrict__ pn, const float4* __restrict__ nr,
                                                   const uint32_t* __restrict__ rank, const uint32_t* __restrict__ valid_next,
                                                   int n, float* __restrict__ o_xyz, float* __restrict__ o_nrm,
                                                   float* __restrict__ o_curv, int* __restrict__ o_idx, int cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = rank[i];
  if (valid_next[i] == k || (int)k >= cap) return;   // rank[i + 1] == rank[i]: point i was dropped
  const float4 a = pn[i], b = nr[i];
  o_xyz[3 * (size_t)k] = a.x;
  o_xyz[3 * (size_t)k + 1] = a.y;
  o_xyz[3 * (size_t)k + 2] = a.z;
  if (o_nrm) {
    o_nrm[3 * (size_t)k] = b.x;
    o_nrm[3 * (size_t)k + 1] = b.y;
    o_nrm[3 * (size_t)k + 2] = b.z;
  }
  if (o_curv) o_curv[k] = a.w;
  if (o_idx) o_idx[k] = i;
}

}  // namespace

// d_xyz: n x 3 device floats.  Outputs (device, cap rows; d_nrm / d_curv / d_index nullable): the smoothed
// points, their normals, curvatures and the input index of every point that had >= 3 neighbours, in input
// order.  *n_out = their number (host).  Synchronises `st`.
int launch_mls(pgp_ctx* ctx, const float* d_xyz, int n, float radius, float* d_out_xyz, float* d_out_nrm,
               float* d_out_curv, int* d_out_index, int cap, int* n_out, hipStream_t st) {
  *n_out = 0;
  if (n <= 0) return PGP_OK;
  float mn[3], mx[3];
  int rc = device_bbox(ctx, d_xyz, n, 3, mn, mx, st);
  if (rc != PGP_OK) return rc;
  if (!(mn[0] <= mx[0])) return PGP_OK;   // no finite point
  MlsDesc g{};
  float maxabs = 0.f;
  for (int k = 0; k < 3; ++k) maxabs = fmaxf(maxabs, fmaxf(fabsf(mn[k]), fabsf(mx[k])));
  // cell edge >= radius, with room for the rounding of the float cell coordinate: a neighbour within the
  // radius is then at most one cell away on every axis
  float h = radius * 1.001f + 64.f * FLT_EPSILON * maxabs;
  for (;;) {
    const double nx = floor((double)(mx[0] - mn[0]) / h) + 2, ny = floor((double)(mx[1] - mn[1]) / h) + 2,
                 nz = floor((double)(mx[2] - mn[2]) / h) + 2;
    if (nx < 2097152.0 && ny < 2097152.0 && nz < 2097152.0) {   // 21 bits per axis in the 64-bit key
      g.nx = (int)nx;
      g.ny = (int)ny;
      g.nz = (int)nz;
      break;
    }
    h *= 2.f;
  }
  g.ox = mn[0];
  g.oy = mn[1];
  g.oz = mn[2];
  g.inv_h = 1.0f / h;
  g.r2 = (float)((double)radius * (double)radius);
  g.gauss = (double)radius * (double)radius;

  const size_t N = (size_t)n;
  size_t sort_bytes = 0;
  hipError_t he = rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                            (uint32_t*)nullptr, (uint32_t*)nullptr, N, 0, 64, st);
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_pairs (size query) failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  // keys_in | keys_out | vals_in | vals_out | sorted float4 | pn float4 | nr float4 | valid/rank (n + 1) | sort temp
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_kin = take(N * 8), o_kout = take(N * 8), o_vin = take(N * 4), o_vout = take(N * 4), o_sorted = take(N * 16),
               o_pn = take(N * 16), o_nr = take(N * 16), o_valid = take((N + 1) * 4), o_rank = take((N + 1) * 4),
               o_tmp = take(sort_bytes + 256);
  if ((rc = ctx->d_mls_ws.ensure(off)) != PGP_OK) return rc;
  if ((rc = ctx->d_scan_tmp.ensure(((N + 1) / 2048 + 2) * 4)) != PGP_OK) return rc;
  unsigned char* base = ctx->d_mls_ws.as<unsigned char>();
  unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(base + o_kin);
  unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(base + o_kout);
  uint32_t* vals_in = reinterpret_cast<uint32_t*>(base + o_vin);
  uint32_t* vals_out = reinterpret_cast<uint32_t*>(base + o_vout);
  float4* sorted = reinterpret_cast<float4*>(base + o_sorted);
  float4* pn = reinterpret_cast<float4*>(base + o_pn);
  float4* nr = reinterpret_cast<float4*>(base + o_nr);
  uint32_t* valid = reinterpret_cast<uint32_t*>(base + o_valid);
  uint32_t* rank = reinterpret_cast<uint32_t*>(base + o_rank);
  const dim3 gn((n + 255) / 256);
  hipLaunchKernelGGL(mls_keys, gn, dim3(256), 0, st, d_xyz, n, g, keys_in, vals_in);
  he = rocprim::radix_sort_pairs(base + o_tmp, sort_bytes, keys_in, keys_out, vals_in, vals_out, N, 0, 64, st);   // stable
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_pairs failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  hipLaunchKernelGGL(mls_gather, gn, dim3(256), 0, st, d_xyz, (const uint32_t*)vals_out, n, sorted);
  PGP_HIP(hipMemsetAsync(valid + N, 0, 4, st));
  hipLaunchKernelGGL(mls_points, dim3((n + 127) / 128), dim3(128), 0, st, d_xyz, n, g, (const unsigned long long*)keys_out,
                     (const float4*)sorted, n, pn, nr, valid);
  if ((rc = device_exclusive_scan(valid, rank, N + 1, ctx->d_scan_tmp.as<uint32_t>(), st)) != PGP_OK) return rc;
  hipLaunchKernelGGL(mls_compact, gn, dim3(256), 0, st, (const float4*)pn, (const float4*)nr, (const uint32_t*)rank,
                     (const uint32_t*)(rank + 1), n, d_out_xyz, d_out_nrm, d_out_curv, d_out_index, cap > 0 ? cap : 0);
  PGP_HIP(hipGetLastError());
  uint32_t total = 0;
  PGP_HIP(hipMemcpyAsync(&total, rank + N, 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  *n_out = (int)total;
  return PGP_OK;
}

}  // namespace pgp
