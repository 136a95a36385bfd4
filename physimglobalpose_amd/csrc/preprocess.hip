// csrc/preprocess.hip -- the steps either side of the scoring path that complete SURVEY 8(f):
//
//   voxel grid        pcl::VoxelGrid<PointXYZRGB> with leaf 0.01 in front of the segment
//                     (PPE/segmentation/Segmentation.cpp:234-237): centroid per occupied voxel, leaves
//                     in ascending voxel index (PCL's layout order).  PCL is not vendored:
//                     the published algorithm is followed (see orc_voxel_grid for the wording); PCL
//                     sorts with std::sort, so the order in which a voxel's points are added is
//                     unspecified there -- here it is the point index (stable radix sort).
//   Hausdorff         Match4PCSBase::c_dist_pose / c_dist_pose_mean (S4/algorithms/match4pcsBase.cc:
//                     1616-1655): directed Hausdorff distance (max, and the sum the reference calls
//                     mean) between the hull under two transforms, for a list of pose pairs.
//   device forms      pgp_set_scene_device: the scene from device arrays (image -> segment -> index
//                     without leaving HBM).
//
// Mapping.  Voxel grid: bounding box by a multi-block reduction, one 64-bit voxel key per point, a
// stable device radix sort of (key, index), run heads by a scan, then ONE thread per voxel adds its
// run in index order (runs are short: ~10^2 points at 1 mm pixel pitch and 1 cm leaves) -- float
// sums in a defined order, so the result is reproducible bit for bit.  Hausdorff: one wave per pose
// pair; the hull under T2 is staged in LDS (broadcast reads), each lane owns hull points under T1 and
// keeps the minimum squared distance (sqrt is monotone: min of the norms = norm of the min); the max
// folds over lanes in any order, the sum is added in hull order (v_readlane chain), as the reference.

#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it

#include "pgp_internal.h"

#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <cmath>
#include <vector>

namespace pgp {

namespace {

__device__ __forceinline__ float sqrt_rn(float z) { return (float)__dsqrt_rn((double)z); }

// ---- bounding box over finite points: per-block partials, the host folds them ------------------
template <int STRIDE>   // floats per point: 3 (packed xyz) or 4 (float4)
__global__ __launch_bounds__(256) void bbox_partial(const float* __restrict__ xyz, int n, float* __restrict__ out) {
  __shared__ float s_mn[3][4], s_mx[3][4];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float x = xyz[(size_t)STRIDE * i], y = xyz[(size_t)STRIDE * i + 1], z = xyz[(size_t)STRIDE * i + 2];
    if (!(fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX)) continue;   // NaN / inf
    mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
    mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
  }
  for (int k = 0; k < 3; ++k)
    for (int off = 32; off >= 1; off >>= 1) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
    }
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; ++k) {
      s_mn[k][threadIdx.x >> 6] = mn[k];
      s_mx[k][threadIdx.x >> 6] = mx[k];
    }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    out[6 * blockIdx.x + k] = fminf(fminf(s_mn[k][0], s_mn[k][1]), fminf(s_mn[k][2], s_mn[k][3]));
    out[6 * blockIdx.x + 3 + k] = fmaxf(fmaxf(s_mx[k][0], s_mx[k][1]), fmaxf(s_mx[k][2], s_mx[k][3]));
  }
}

struct VgDesc {
  float inv;           // 1 / leaf
  int min_b[3];
  long long div0, div01;
};

__global__ __launch_bounds__(256) void vg_keys(const float* __restrict__ xyz, int n, VgDesc g,
                                               unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
  unsigned long long key = ~0ull;   // non-finite points sort to the end and are dropped
  if (fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX) {
    // ijk = static_cast<int>(floor(p * inverse_leaf_size) - static_cast<float>(min_b))
    const int i0 = (int)__fsub_rn(floorf(__fmul_rn(x, g.inv)), (float)g.min_b[0]);
    const int i1 = (int)__fsub_rn(floorf(__fmul_rn(y, g.inv)), (float)g.min_b[1]);
    const int i2 = (int)__fsub_rn(floorf(__fmul_rn(z, g.inv)), (float)g.min_b[2]);
    key = (unsigned long long)((long long)i0 + (long long)i1 * g.div0 + (long long)i2 * g.div01);
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void vg_heads(const unsigned long long* __restrict__ keys, int n,
                                                uint32_t* __restrict__ head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    head[i] = 0u;   // scan sentinel
    return;
  }
  const unsigned long long k = keys[i];
  head[i] = (k != ~0ull && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

// one thread per sorted position that starts a voxel: add its run in index order
__global__ __launch_bounds__(256) void vg_centroids(const unsigned long long* __restrict__ keys,
                                                    const uint32_t* __restrict__ vals, const uint32_t* __restrict__ rank,
                                                    int n, const float* __restrict__ xyz, float* __restrict__ out,
                                                    uint32_t cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  if (k == ~0ull || (i > 0 && keys[i - 1] == k)) return;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  int j = i;
  for (; j < n && keys[j] == k; ++j) {
    const uint32_t p = vals[j];
    cx = __fadd_rn(cx, xyz[3 * (size_t)p]);
    cy = __fadd_rn(cy, xyz[3 * (size_t)p + 1]);
    cz = __fadd_rn(cz, xyz[3 * (size_t)p + 2]);
  }
  const uint32_t v = rank[i];   // exclusive scan of the head flags = leaf number
  if (v < cap) {
    const float cnt = (float)(j - i);
    out[3 * (size_t)v] = __fdiv_rn(cx, cnt);
    out[3 * (size_t)v + 1] = __fdiv_rn(cy, cnt);
    out[3 * (size_t)v + 2] = __fdiv_rn(cz, cnt);
  }
}

// ---- directed Hausdorff between the hull under two transforms -----------------------------------
constexpr int kHullMax = 4096;   // hull points staged in LDS (64 KB as float4)

__device__ __forceinline__ float xf_row(float a, float b, float c, float t, float q0, float q1, float q2) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, q0), __fmul_rn(b, q1)), __fmul_rn(c, q2)), t);
}

__global__ __launch_bounds__(64) void pose_hausdorff(const float4* __restrict__ hull, int n_hull,
                                                     const float* __restrict__ T, int n_poses,
                                                     const int2* __restrict__ pairs, int m, float* __restrict__ d_max,
                                                     float* __restrict__ d_sum) {
  extern __shared__ float4 s_q[];   // hull under T2
  const int pr = blockIdx.x;
  if (pr >= m) return;
  const int2 ij = pairs[pr];
  const int lane = threadIdx.x;
  if ((unsigned)ij.x >= (unsigned)n_poses || (unsigned)ij.y >= (unsigned)n_poses) {
    if (lane == 0) {
      d_max[pr] = __int_as_float(0x7FC00000);
      d_sum[pr] = __int_as_float(0x7FC00000);
    }
    return;
  }
  const float* A = T + 16 * (size_t)ij.x;
  const float* B = T + 16 * (size_t)ij.y;
  for (int j = lane; j < n_hull; j += 64) {
    const float4 h = hull[j];
    s_q[j] = make_float4(xf_row(B[0], B[4], B[8], B[12], h.x, h.y, h.z), xf_row(B[1], B[5], B[9], B[13], h.x, h.y, h.z),
                         xf_row(B[2], B[6], B[10], B[14], h.x, h.y, h.z), 0.f);
  }
  __syncthreads();
  float mx = 0.f;     // max_distance = 0
  float S = 0.f;      // mean_distance = 0 (kept by every lane: the chain below is wave-uniform)
  for (int base = 0; base < n_hull; base += 64) {
    const int ii = base + lane;
    float mind = FLT_MAX;   // min_distance = FLT_MAX; stays there when every dist is NaN or >= FLT_MAX
    if (ii < n_hull) {
      const float4 h = hull[ii];
      const float px = xf_row(A[0], A[4], A[8], A[12], h.x, h.y, h.z), py = xf_row(A[1], A[5], A[9], A[13], h.x, h.y, h.z),
                  pz = xf_row(A[2], A[6], A[10], A[14], h.x, h.y, h.z);
      float min2 = __int_as_float(0x7F800000);   // +inf
      for (int jj = 0; jj < n_hull; ++jj) {
        const float4 q = s_q[jj];
        const float dx = __fsub_rn(px, q.x), dy = __fsub_rn(py, q.y), dz = __fsub_rn(pz, q.z);
        const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dz, dz)));
        min2 = d2 < min2 ? d2 : min2;   // NaN never replaces the minimum, as `dist < min_distance`
      }
      const float dist = sqrt_rn(min2);
      mind = dist < FLT_MAX ? dist : FLT_MAX;
    }
    const float for_max = ii < n_hull ? mind : 0.f;
    float wmax = for_max;
    for (int off = 32; off >= 1; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
    mx = wmax > mx ? wmax : mx;
    // mean_distance += min_distance, in hull order
    const int cnt = min(64, n_hull - base);
    for (int k = 0; k < cnt; ++k) S = __fadd_rn(S, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mind), k)));
  }
  if (lane == 0) {
    d_max[pr] = mx;
    d_sum[pr] = S;
  }
}

// xyz (n x 3) [+ normals, weights] -> the context's float4 scene arrays
__global__ __launch_bounds__(256) void pack_scene(const float* __restrict__ xyz, const float* __restrict__ nrm,
                                                  const float* __restrict__ w, int n, float4* __restrict__ P,
                                                  float4* __restrict__ Pnw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  P[i] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float(i));
  const float ww = w ? w[i] : 1.0f;
  Pnw[i] = nrm ? make_float4(nrm[3 * (size_t)i], nrm[3 * (size_t)i + 1], nrm[3 * (size_t)i + 2], ww)
               : make_float4(0.f, 0.f, 0.f, ww);
}

int device_bbox3(pgp_ctx* ctx, const float* d_xyz, int n, int stride, float mn[3], float mx[3], hipStream_t st) {
  const int nb = 64;
  int rc = ctx->d_pre_ws.ensure((size_t)nb * 24 + 64);
  if (rc != PGP_OK) return rc;
  float* d_part = ctx->d_pre_ws.as<float>();
  if (stride == 3) hipLaunchKernelGGL(bbox_partial<3>, dim3(nb), dim3(256), 0, st, d_xyz, n, d_part);
  else hipLaunchKernelGGL(bbox_partial<4>, dim3(nb), dim3(256), 0, st, d_xyz, n, d_part);
  PGP_HIP(hipGetLastError());
  float part[64 * 6];
  PGP_HIP(hipMemcpyAsync(part, d_part, sizeof part, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  for (int k = 0; k < 3; ++k) {
    mn[k] = FLT_MAX;
    mx[k] = -FLT_MAX;
  }
  for (int b = 0; b < nb; ++b)
    for (int k = 0; k < 3; ++k) {
      mn[k] = fminf(mn[k], part[6 * b + k]);
      mx[k] = fmaxf(mx[k], part[6 * b + 3 + k]);
    }
  return PGP_OK;
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void scene_weights(float4* __restrict__ Pnw, const float* __restrict__ w, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Pnw[i].w = w[i];
}
}  // namespace

// ---- points of a segment that already-placed objects explain (UCTState::performTrICP, UCTState.cpp:142-174) ----
// The reference moves every placed object's model to its pose, builds one kd-tree over all of them and
// removes each segment point that has ANY such point within pointRemovalThreshold (8 mm, UCTState.cpp:9;
// FLANN's radius search keeps a neighbour when its squared distance is strictly below radius^2).
// Here: grid (segment chunks of 256, objects); a workgroup moves a tile of 1024 model points to the object's
// pose in LDS -- pose rows as ((r0 x + r1 y) + r2 z) + t, float -- and every lane tests its segment point
// against the tile (broadcast reads), leaving as soon as the whole wave is explained; explained[i] |= 1.
// Squared distance as FLANN's L2 accumulates it, (dx^2 + dy^2) + dz^2.  PCL / FLANN are not vendored:
// unpinned against their bits, pinned against oracle/preprocess_oracle.py.
namespace {
struct ExplainArgs {
  const float* seg;          // n x 3
  int n;
  const float* model;        // all objects' model points, 3 floats each
  const int* model_off;      // [K + 1] point offsets into `model`
  const float* T;            // K x 16 column-major: model -> the segment's frame
  float r2;
  unsigned int* explained;   // [n] zeroed by the caller
};
__global__ __launch_bounds__(256) void explained_points(ExplainArgs a) {
  __shared__ float4 s_m[1024];
  const int k = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float* G = a.T + 16 * (size_t)k;
  const int m0 = a.model_off[k], m1 = a.model_off[k + 1];
  const float qnan = __int_as_float(0x7FC00000);
  const float x = i < a.n ? a.seg[3 * (size_t)i] : qnan, y = i < a.n ? a.seg[3 * (size_t)i + 1] : qnan,
              z = i < a.n ? a.seg[3 * (size_t)i + 2] : qnan;
  bool hit = false;
  for (int t0 = m0; t0 < m1; t0 += 1024) {
    const int tn = min(1024, m1 - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < tn; j += 256) {
      const float* p = a.model + 3 * (size_t)(t0 + j);
      s_m[j] = make_float4(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(G[0], p[0]), __fmul_rn(G[4], p[1])), __fmul_rn(G[8], p[2])), G[12]),
                           __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(G[1], p[0]), __fmul_rn(G[5], p[1])), __fmul_rn(G[9], p[2])), G[13]),
                           __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(G[2], p[0]), __fmul_rn(G[6], p[1])), __fmul_rn(G[10], p[2])), G[14]), 0.f);
    }
    __syncthreads();
    if (__ballot(!hit && i < a.n) != 0ull) {   // a wave whose points are all explained only keeps the barriers
      for (int j = 0; j < tn; ++j) {
        const float4 m = s_m[j];
        const float dx = __fsub_rn(x, m.x), dy = __fsub_rn(y, m.y), dz = __fsub_rn(z, m.z);
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        hit |= d2 < a.r2;
      }
    }
  }
  if (hit && i < a.n) a.explained[i] = 1u;   // benign race across objects: every writer stores 1
}
}  // namespace

int launch_explained_points(pgp_ctx* ctx, const float* d_seg, int n, const float* d_model, const int* d_model_off,
                            const float* d_T, int n_obj, float radius, unsigned int* d_explained, hipStream_t st) {
  (void)ctx;
  if (n <= 0) return PGP_OK;
  PGP_HIP(hipMemsetAsync(d_explained, 0, (size_t)n * 4, st));
  if (n_obj <= 0) return PGP_OK;
  ExplainArgs a{d_seg, n, d_model, d_model_off, d_T, radius * radius, d_explained};
  hipLaunchKernelGGL(explained_points, dim3((n + 255) / 256, n_obj), dim3(256), 0, st, a);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

// the weights of the resident scene replaced (d_w: n device floats); enqueued on st
int launch_scene_weights(pgp_ctx* ctx, const float* d_w, int n, hipStream_t st) {
  if (n <= 0) return PGP_OK;
  hipLaunchKernelGGL(scene_weights, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_Pnw.as<float4>(), d_w, n);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

int device_bbox(pgp_ctx* ctx, const float* d_pts, int n, int stride, float mn[3], float mx[3], hipStream_t st) {
  return device_bbox3(ctx, d_pts, n, stride, mn, mx, st);
}

// d_xyz: n x 3 device floats; d_out: cap x 3; *n_out: number of leaves (host).  Synchronises `st`.
int launch_voxel_grid(pgp_ctx* ctx, const float* d_xyz, int n, float leaf, float* d_out, int cap, int* n_out,
                      hipStream_t st) {
  *n_out = 0;
  if (n <= 0) return PGP_OK;
  float mn[3], mx[3];
  int rc = device_bbox3(ctx, d_xyz, n, 3, mn, mx, st);
  if (rc != PGP_OK) return rc;
  if (!(mn[0] <= mx[0])) return PGP_OK;   // no finite point
  VgDesc g{};
  g.inv = 1.0f / leaf;
  long long div[3];
  for (int k = 0; k < 3; ++k) {
    g.min_b[k] = (int)floorf(mn[k] * g.inv);
    div[k] = (long long)(int)floorf(mx[k] * g.inv) - g.min_b[k] + 1;
  }
  if ((double)div[0] * (double)div[1] * (double)div[2] > 9.0e18) {
    set_error("pgp_voxel_grid: leaf %g is too small for the extent of the cloud", (double)leaf);
    return PGP_EINVAL;
  }
  g.div0 = div[0];
  g.div01 = div[0] * div[1];
  const size_t N = (size_t)n;
  size_t sort_bytes = 0;
  hipError_t he = rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                            (uint32_t*)nullptr, (uint32_t*)nullptr, N, 0, 64, st);
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_pairs (size query) failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  // keys_in | keys_out | vals_in | vals_out | head/rank (n + 1) | sort temp
  const size_t off_vals = 2 * N * 8, off_head = off_vals + 2 * N * 4, off_tmp = (off_head + (N + 1) * 4 + 255) & ~(size_t)255;
  if ((rc = ctx->d_vg_ws.ensure(off_tmp + sort_bytes + 256)) != PGP_OK) return rc;
  if ((rc = ctx->d_scan_tmp.ensure(((N + 1) / 2048 + 2) * 4)) != PGP_OK) return rc;
  unsigned char* base = ctx->d_vg_ws.as<unsigned char>();
  unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(base);
  unsigned long long* keys_out = keys_in + N;
  uint32_t* vals_in = reinterpret_cast<uint32_t*>(base + off_vals);
  uint32_t* vals_out = vals_in + N;
  uint32_t* head = reinterpret_cast<uint32_t*>(base + off_head);
  void* sort_tmp = base + off_tmp;
  const dim3 gn((n + 255) / 256), gn1((n + 1 + 255) / 256);
  hipLaunchKernelGGL(vg_keys, gn, dim3(256), 0, st, d_xyz, n, g, keys_in, vals_in);
  he = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_in, keys_out, vals_in, vals_out, N, 0, 64, st);   // stable
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_pairs failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  hipLaunchKernelGGL(vg_heads, gn1, dim3(256), 0, st, (const unsigned long long*)keys_out, n, head);
  if ((rc = device_exclusive_scan(head, head, N + 1, ctx->d_scan_tmp.as<uint32_t>(), st)) != PGP_OK) return rc;
  hipLaunchKernelGGL(vg_centroids, gn, dim3(256), 0, st, (const unsigned long long*)keys_out, (const uint32_t*)vals_out,
                     (const uint32_t*)head, n, d_xyz, d_out, (uint32_t)(cap > 0 ? cap : 0));
  PGP_HIP(hipGetLastError());
  uint32_t total = 0;
  PGP_HIP(hipMemcpyAsync(&total, head + N, 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  *n_out = (int)total;
  return PGP_OK;
}

int launch_pose_hausdorff(pgp_ctx* ctx, const float4* d_hull, int n_hull, const float* d_T, int n_poses,
                          const int2* d_pairs, int m, float* d_max, float* d_sum, hipStream_t st) {
  if (n_hull > kHullMax) {
    set_error("pgp_pose_hausdorff: at most %d hull points", kHullMax);
    return PGP_EINVAL;
  }
  if (m <= 0) return PGP_OK;
  const size_t lds = (size_t)(n_hull > 0 ? n_hull : 1) * sizeof(float4);
  if (!ctx->hd_attr_set) {
    PGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pose_hausdorff), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(kHullMax * sizeof(float4))));
    ctx->hd_attr_set = true;
  }
  hipLaunchKernelGGL(pose_hausdorff, dim3(m), dim3(64), lds, st, d_hull, n_hull, d_T, n_poses, d_pairs, m, d_max, d_sum);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

// The scene from DEVICE arrays (n x 3 floats; normals / weights nullable): packs them into the
// context's layout, finds the bounding box on the device and builds the index.  Synchronises `st`.
int set_scene_device(pgp_ctx* ctx, const float* d_xyz, const float* d_nrm, const float* d_w, int n, float delta,
                     hipStream_t st) {
  ctx->has_index = false;
  ctx->prob_cdf_valid = false;
  ctx->nP = n;
  ctx->has_scene_normals = d_nrm != nullptr;
  int rc;
  const size_t N = (size_t)(n > 0 ? n : 1);
  if ((rc = ctx->d_P.ensure(N * sizeof(float4))) != PGP_OK) return rc;
  if ((rc = ctx->d_Pnw.ensure(N * sizeof(float4))) != PGP_OK) return rc;
  float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
  if (n > 0) {
    hipLaunchKernelGGL(pack_scene, dim3((n + 255) / 256), dim3(256), 0, st, d_xyz, d_nrm, d_w, n, ctx->d_P.as<float4>(),
                       ctx->d_Pnw.as<float4>());
    PGP_HIP(hipGetLastError());
    if ((rc = device_bbox3(ctx, d_xyz, n, 3, mn, mx, st)) != PGP_OK) return rc;
    if (!(mn[0] <= mx[0]))
      for (int k = 0; k < 3; ++k) mn[k] = mx[k] = 0.f;
  }
  // build_index works on ctx->stream; the packing above ran on `st`
  PGP_HIP(hipStreamSynchronize(st));
  ctx->kd_valid = false;
  if (ctx->exact_ties) {   // the reference's tree is built on the host: one copy of the cloud back
    std::vector<float> h((size_t)3 * N);
    if (n > 0) PGP_HIP(hipMemcpy(h.data(), d_xyz, (size_t)n * 12, hipMemcpyDeviceToHost));
    if ((rc = build_kd_ties(ctx, h.data(), n)) != PGP_OK) return rc;
  }
  return build_index_bbox(ctx, mn, mx, delta);
}

}  // namespace pgp
