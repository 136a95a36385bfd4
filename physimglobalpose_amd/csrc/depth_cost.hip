// csrc/depth_cost.hip -- MCTS leaf cost: rendered depth vs observed depth.
//
// Replaces UCTState::computeCost (PPE/hypothesis_verification/mcts/UCTState.cpp:93-116) for a batch
// of rendered images against one observed image (the per-expansion cost after physics;
// rendering itself stays on the host).  Per pixel, with d = |obs - ren| (float):
//   obScore  += obs > 0 && d > thr        renScore += ren > 0 && d > thr
//   intScore += obs > 0 && ren > 0 && d > thr        renderScore = obScore + renScore - intScore
// The reference counts in float; every count is <= rows*cols < 2^24, so integer counting gives
// the same value exactly.
//
// Streaming, HBM-bound by construction: 4 B of observed + 4 B of rendered depth per pixel and
// nothing else; 16-byte loads, ballot/popcount per wave, one atomic per block and image.
// Algorithmic bytes per image: 8 * rows * cols (the observed image is re-read per rendered image;
// at 640x480 it stays in L2).

#include "pgp_internal.h"

namespace pgp {

namespace {

__device__ __forceinline__ void tally(float o, float r, float thr, int* ob, int* re, int* in) {
  const float d = fabsf(__fsub_rn(o, r));
  const bool far = d > thr;
  *ob += (o > 0.f) & far;
  *re += (r > 0.f) & far;
  *in += (o > 0.f) & (r > 0.f) & far;
}

__global__ __launch_bounds__(256) void depth_cost(const float* __restrict__ obs, const float* __restrict__ ren,
                                                  int n_pix, float thr, int* __restrict__ counts /*[n][3]*/) {
  const int img = blockIdx.y;
  const float* r = ren + (size_t)img * n_pix;
  int ob = 0, re = 0, in = 0;
  const int n4 = n_pix >> 2;
  const bool aligned = (((uintptr_t)obs | (uintptr_t)r) & 15) == 0;
  if (aligned) {
    const float4* o4 = reinterpret_cast<const float4*>(obs);
    const float4* r4 = reinterpret_cast<const float4*>(r);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
      const float4 a = o4[i], b = r4[i];
      tally(a.x, b.x, thr, &ob, &re, &in);
      tally(a.y, b.y, thr, &ob, &re, &in);
      tally(a.z, b.z, thr, &ob, &re, &in);
      tally(a.w, b.w, thr, &ob, &re, &in);
    }
    for (int i = 4 * n4 + blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += gridDim.x * blockDim.x)
      tally(obs[i], r[i], thr, &ob, &re, &in);
  } else {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += gridDim.x * blockDim.x)
      tally(obs[i], r[i], thr, &ob, &re, &in);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    ob += __shfl_xor(ob, off, 64);
    re += __shfl_xor(re, off, 64);
    in += __shfl_xor(in, off, 64);
  }
  __shared__ int s[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s[wave][0] = ob; s[wave][1] = re; s[wave][2] = in; }
  __syncthreads();
  if (threadIdx.x < 3) {
    int v = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
    if (v) atomicAdd(&counts[3 * img + threadIdx.x], v);  // integer: order-independent
  }
}

}  // namespace

int launch_depth_cost(pgp_ctx* ctx, const float* d_obs, const float* d_ren, int n, int n_pix, float thr,
                      int* d_counts, hipStream_t stream) {
  (void)ctx;
  if (n <= 0) return PGP_OK;
  PGP_HIP(hipMemsetAsync(d_counts, 0, (size_t)n * 3 * sizeof(int), stream));
  if (n_pix <= 0) return PGP_OK;
  int bx = (n_pix / 4 + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;  // 640x480 / 4 / 256 = 300 -> 64 blocks x n images, grid-stride
  hipLaunchKernelGGL(depth_cost, dim3(bx, n), dim3(256), 0, stream, d_obs, d_ren, n_pix, thr, d_counts);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

}  // namespace pgp
