// csrc/backproject.hip -- depth image -> camera-frame point cloud, in scan order.
//
// Replaces, for one object mask, the step in front of the segment cloud:
//   utilities::readDepthImage        PPE/misc/utilities.cpp:47-61   raw 16-bit -> metres
//                                    (rotate right by 3, then / 10000, both in the reference)
//   objDepth = depthImage.mul(mask)  PPE/segmentation/Segmentation.cpp:219
//   utilities::convert3dUnOrganized  PPE/misc/utilities.cpp:190-206 (and the RGB twin :210-231)
// Arithmetic as written there, every operation in float:
//   x = (float)((v - cx) * depth / fx),  y = (float)((u - cy) * depth / fy),  z = depth
// for the pixels with 0.1 < depth < 2.0 (depth promoted to double for the comparison, as the
// literals are doubles), emitted in row-major order: the output is the same LIST as the reference's
// push_back loop.  Ordered compaction = per-workgroup counts, exclusive scan, ranked write.
//
// HBM-bound streaming: 2 or 4 B of depth (+1 B of mask) per pixel in, 12 B per kept pixel out.

#include "pgp_internal.h"

namespace pgp {

namespace {

constexpr int kBpThreads = 256;

template <bool RAW16>
__device__ __forceinline__ float depth_at(const void* img, size_t i) {
  if (RAW16) {
    unsigned short s = static_cast<const unsigned short*>(img)[i];
    s = (unsigned short)((s << 13) | (s >> 3));  // utilities.cpp:57 (APC encoding)
    return __fdiv_rn((float)s, 10000.0f);        // :59 (float)depthShort/10000
  }
  return static_cast<const float*>(img)[i];
}

__device__ __forceinline__ bool depth_ok(float d, double z_min, double z_max) {
  return (double)d > z_min && (double)d < z_max;  // utilities.cpp:197 / :218
}

// pass 1: kept pixels per workgroup; pass 2 (WRITE): ranked write behind the scanned offsets
template <bool RAW16, bool WRITE>
__global__ __launch_bounds__(kBpThreads) void backproject(const void* __restrict__ img,
                                                          const unsigned char* __restrict__ mask, int rows,
                                                          int cols, float fx, float fy, float cx, float cy,
                                                          double z_min, double z_max,
                                                          uint32_t* __restrict__ block_ctr,
                                                          float* __restrict__ xyz, uint32_t cap) {
  __shared__ uint32_t s_wave[kBpThreads / 64];
  const size_t n = (size_t)rows * cols;
  const size_t i = (size_t)blockIdx.x * kBpThreads + threadIdx.x;
  float d = 0.f;
  bool keep = false;
  if (i < n) {
    d = depth_at<RAW16>(img, i);
    if (mask && mask[i] == 0) d = 0.f;  // depthImage.mul(objMask): a masked-out pixel has depth 0
    keep = depth_ok(d, z_min, z_max);
  }
  const unsigned long long b = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = (uint32_t)__popcll(b);
  __syncthreads();
  if (!WRITE) {
    if (threadIdx.x == 0) block_ctr[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return;
  }
  if (!keep) return;
  uint32_t rank = block_ctr[blockIdx.x] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) rank += s_wave[w];
  if (rank >= cap) return;
  const int u = (int)(i / (size_t)cols), v = (int)(i - (size_t)u * cols);
  xyz[3 * (size_t)rank] = __fdiv_rn(__fmul_rn(__fsub_rn((float)v, cx), d), fx);
  xyz[3 * (size_t)rank + 1] = __fdiv_rn(__fmul_rn(__fsub_rn((float)u, cy), d), fy);
  xyz[3 * (size_t)rank + 2] = d;
}

}  // namespace

int launch_backproject(pgp_ctx* ctx, const void* d_img, bool raw16, const unsigned char* d_mask, int rows,
                       int cols, const float K[9], double z_min, double z_max, uint32_t* d_ctr,
                       uint32_t* d_scan_tmp, float* d_xyz, int cap, int* n_host, hipStream_t st) {
  (void)ctx;
  *n_host = 0;
  const size_t n = (size_t)rows * cols;
  if (n == 0) return PGP_OK;
  const int nb = (int)((n + kBpThreads - 1) / kBpThreads);
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];  // row-major 3x3
  PGP_HIP(hipMemsetAsync(d_ctr + nb, 0, sizeof(uint32_t), st));  // scan sentinel
  if (raw16)
    hipLaunchKernelGGL((backproject<true, false>), dim3(nb), dim3(kBpThreads), 0, st, d_img, d_mask, rows, cols, fx,
                       fy, cx, cy, z_min, z_max, d_ctr, (float*)nullptr, 0u);
  else
    hipLaunchKernelGGL((backproject<false, false>), dim3(nb), dim3(kBpThreads), 0, st, d_img, d_mask, rows, cols,
                       fx, fy, cx, cy, z_min, z_max, d_ctr, (float*)nullptr, 0u);
  int rc = device_exclusive_scan(d_ctr, d_ctr, (size_t)nb + 1, d_scan_tmp, st);
  if (rc != PGP_OK) return rc;
  if (raw16)
    hipLaunchKernelGGL((backproject<true, true>), dim3(nb), dim3(kBpThreads), 0, st, d_img, d_mask, rows, cols, fx,
                       fy, cx, cy, z_min, z_max, d_ctr, d_xyz, (uint32_t)cap);
  else
    hipLaunchKernelGGL((backproject<false, true>), dim3(nb), dim3(kBpThreads), 0, st, d_img, d_mask, rows, cols, fx,
                       fy, cx, cy, z_min, z_max, d_ctr, d_xyz, (uint32_t)cap);
  PGP_HIP(hipGetLastError());
  uint32_t total = 0;
  PGP_HIP(hipMemcpyAsync(&total, d_ctr + nb, sizeof total, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  *n_host = (int)total;
  return PGP_OK;
}

}  // namespace pgp
