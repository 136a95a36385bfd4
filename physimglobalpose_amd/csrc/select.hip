// csrc/select.hip -- the k best-scoring hypotheses of a scored batch, as the initial guesses of the ICP that follows.
//
// The reference hands its best candidates to the refinement one object after the other on the host
// (PPE/hypothesis_verification/HypothesisSelection.cpp:248-257 reads the scores, UCTState.cpp:184-185 inverts the
// pose: `tform = pose.inverse()` -- ICP moves the SEGMENT onto the MODEL).  Here scores and transforms are in HBM
// already; this file keeps the hand-off there: sort the score bits (stable radix sort, descending: among equal
// scores the lower hypothesis index comes first, as a stable host argsort gives), gather the first k transforms
// and -- on request -- write their rigid inverses {R^T, -R^T t} (double arithmetic, rounded once).
#include "pgp_internal.h"

#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it

#include <rocprim/rocprim.hpp>

namespace pgp {
namespace {

__global__ __launch_bounds__(256) void top_keys(const float* __restrict__ scores, int n, unsigned* __restrict__ keys,
                                                int* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = scores[i];
  keys[i] = s > 0.f ? __float_as_uint(s) : 0u;   // NaN and <= 0 never qualify (best_LCP_ starts at 0, strict >)
  idx[i] = i;
}

__global__ __launch_bounds__(256) void top_gather(const float* __restrict__ T, const unsigned* __restrict__ keys,
                                                  const int* __restrict__ idx, int n, int k, int invert,
                                                  float* __restrict__ T_out, int* __restrict__ idx_out, int* __restrict__ n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {   // entries with a positive score among the first k (the sort put the zeros last)
    int m = 0;
    for (int j = 0; j < k && j < n; ++j) m += keys[j] != 0u ? 1 : 0;
    *n_out = m;
  }
  if (i >= k) return;
  const bool ok = i < n && keys[i] != 0u;
  const int h = ok ? idx[i] : -1;
  if (idx_out) idx_out[i] = h;
  float* o = T_out + 16 * (size_t)i;
  if (!ok) {
    for (int q = 0; q < 16; ++q) o[q] = (q % 5 == 0) ? 1.f : 0.f;   // identity: a harmless guess
    return;
  }
  const float* m = T + 16 * (size_t)h;   // column-major: m[4 c + r]
  if (!invert) {
    for (int q = 0; q < 16; ++q) o[q] = m[q];
    return;
  }
  // rigid inverse: rotation transposed, translation -R^T t
  const double tx = m[12], ty = m[13], tz = m[14];
  for (int r = 0; r < 3; ++r) {
    // row r of R^T = column r of R
    const double a = m[4 * r], b = m[4 * r + 1], c = m[4 * r + 2];
    o[r] = (float)a;           // (R^T)(r, 0) = R(0, r) -> column 0, row r
    o[4 + r] = (float)b;
    o[8 + r] = (float)c;
    o[12 + r] = (float)(-(a * tx + b * ty + c * tz));
  }
  o[3] = o[7] = o[11] = 0.f;
  o[15] = 1.f;
}

}  // namespace

int launch_select_top(pgp_ctx* ctx, const float* d_T, const float* d_scores, int n, int k, int invert, float* d_T_out,
                      int* d_idx_out, int* d_n_out, hipStream_t st) {
  if (k <= 0) return PGP_OK;
  size_t sort_bytes = 0;
  hipError_t he = rocprim::radix_sort_pairs_desc(nullptr, sort_bytes, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr,
                                                 (int*)nullptr, (size_t)(n > 0 ? n : 1), 0, 32, st);
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_pairs_desc (size query) failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  const size_t nn = (size_t)(n > 0 ? n : 1);
  int rc = ctx->d_top_ws.ensure(nn * 16 + sort_bytes + 512);
  if (rc != PGP_OK) return rc;
  unsigned* keys_in = ctx->d_top_ws.as<unsigned>();
  unsigned* keys_out = keys_in + nn;
  int* idx_in = reinterpret_cast<int*>(keys_out + nn);
  int* idx_out = idx_in + nn;
  void* tmp = reinterpret_cast<unsigned char*>(idx_out + nn) + 256 - ((nn * 16) & 255);
  if (n > 0) {
    hipLaunchKernelGGL(top_keys, dim3((n + 255) / 256), dim3(256), 0, st, d_scores, n, keys_in, idx_in);
    he = rocprim::radix_sort_pairs_desc(tmp, sort_bytes, keys_in, keys_out, idx_in, idx_out, (size_t)n, 0, 32, st);
    if (he != hipSuccess) {
      set_error("rocprim::radix_sort_pairs_desc failed: %s", hipGetErrorString(he));
      return PGP_EHIP;
    }
  }
  hipLaunchKernelGGL(top_gather, dim3((k + 255) / 256), dim3(256), 0, st, d_T, (const unsigned*)keys_out, (const int*)idx_out, n, k,
                     invert, d_T_out, d_idx_out, d_n_out);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

}  // namespace pgp
