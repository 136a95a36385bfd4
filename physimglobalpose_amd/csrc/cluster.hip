// csrc/cluster.hip -- greedy pose clustering of a scored hypothesis list.
//
// Replaces HypothesisSelection::greedyClustering (PPE/hypothesis_verification/HypothesisSelection.cpp:
// 66-115) with its pose distance utilities::getPoseError (PPE/misc/utilities.cpp:514-548):
//   1. prune   : keep scores > accept_fraction * best_score                        (:70-77)
//   2. order   : score descending (:83); equal scores stay in index order here -- the reference's
//                std::sort leaves that order unspecified
//   3. greedy  : a candidate is kept unless an EARLIER kept one is within (rot_thresh, trans_thresh)
//                of it, getPoseError(candidate, kept)                                (:88-109)
// (the `cluster_it.second += ...` at :102 updates a by-value copy, so scores are not merged).
//
// GPU form: the greedy pass is a non-maximum suppression.  All m(m-1)/2 pair tests are independent
// and go first (one wave = one candidate x 64 earlier poses, ballot -> one 64-bit word of a
// lower-triangular bit matrix); the sequential part then only ANDs words: per tile of 64
// candidates, hits against representatives of earlier tiles are found in parallel, and the
// 64-step dependency chain inside the tile runs on one diagonal word per candidate in registers.
//
// Arithmetic follows the reference's Eigen expressions (3x3 cofactor inverse, size-3 redux order
// x0 + (x1 + x2), Shoemake quaternion extraction, Euler angles in double); float division and
// sqrt are the correctly rounded forms.  atan2 / asin are the device's double-precision routines:
// a pair whose error lies within an ulp-of-double of a threshold could decide differently from
// glibc -- the tests measure the margin of every fixture (oracle/pgp_oracle.c:orc_pose_error).

#include <cstring>

#include "pgp_internal.h"

#include <rocprim/rocprim.hpp>

namespace pgp {

namespace {

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ float sqrt_rn(float z) { return (float)__dsqrt_rn((double)z); }

constexpr int kPoseStride = 12;  // per sorted pose: 9 floats of a 3x3 (row-major) + translation

// float -> uint32 whose unsigned order is the float order (NaN never reaches here)
__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(256) void cluster_keys(const float* __restrict__ scores, int n, float bar,
                                                    unsigned long long* __restrict__ keys,
                                                    int* __restrict__ m_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool pass = false;
  unsigned long long key = 0ull;
  if (i < n) {
    const float s = scores[i];
    pass = s > bar;  // HypothesisSelection.cpp:75
    // ascending sort of (~score, index) = score descending, index ascending
    key = ((unsigned long long)~orderable(s) << 32) | (uint32_t)i;
  }
  // only the poses above the bar are kept (compacted: the sort then handles m keys, not n); their order in
  // `keys` is whatever the atomics made it -- the keys are unique, so the sorted list does not depend on it
  const unsigned long long b = __ballot(pass);
  if (b == 0ull) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(m_out, __popcll(b));
  base = __builtin_amdgcn_readfirstlane(base);
  if (pass) keys[base + __popcll(b & ((1ull << lane) - 1ull))] = key;
}

__device__ __forceinline__ float cof3(const float* a, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return sub(mul(a[3 * i1 + j1], a[3 * i2 + j2]), mul(a[3 * i1 + j2], a[3 * i2 + j1]));
}

// utilities.cpp:523 testRot.inverse(): Eigen compute_inverse<.,.,3> (row-major a, inv)
__device__ __forceinline__ void inverse3(const float* a, float* inv) {
  const float c0 = cof3(a, 0, 0), c1 = cof3(a, 1, 0), c2 = cof3(a, 2, 0);
  const float det = add(mul(c0, a[0]), add(mul(c1, a[3]), mul(c2, a[6])));
  const float invdet = fdiv(1.0f, det);
  inv[0] = mul(c0, invdet);
  inv[1] = mul(c1, invdet);
  inv[2] = mul(c2, invdet);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    inv[3 + j] = mul(cof3(a, j, 1), invdet);
    inv[6 + j] = mul(cof3(a, j, 2), invdet);
  }
}

// Sorted pose tables.  inv_c: AoS, the candidate role (wave-uniform reads): inverse rotation + t.
// rot_r: SoA [12][m], the "cluster" role (lane-indexed reads): rotation + t.
__global__ __launch_bounds__(256) void cluster_gather(const unsigned long long* __restrict__ keys_sorted,
                                                      const float* __restrict__ T, int m,
                                                      float* __restrict__ inv_c, float* __restrict__ rot_r,
                                                      int* __restrict__ idx_sorted) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  const int id = (int)(uint32_t)keys_sorted[c];
  idx_sorted[c] = id;
  const float* t = T + 16 * (size_t)id;  // column-major 4x4
  float a[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) a[3 * i + j] = t[i + 4 * j];
  float inv[9];
  inverse3(a, inv);
  float* o = inv_c + (size_t)kPoseStride * c;
#pragma unroll
  for (int k = 0; k < 9; ++k) o[k] = inv[k];
  o[9] = t[12];
  o[10] = t[13];
  o[11] = t[14];
#pragma unroll
  for (int k = 0; k < 9; ++k) rot_r[(size_t)k * m + c] = a[k];
  rot_r[(size_t)9 * m + c] = t[12];
  rot_r[(size_t)10 * m + c] = t[13];
  rot_r[(size_t)11 * m + c] = t[14];
}

struct Sym {
  float x, y, z;
};

__device__ __forceinline__ float fold(float e, float sym) {
  // utilities.cpp:528-542
  float v = fdiv(mul(e, 180.0f), 3.14159274101257324f);
  v = fabsf(v);
  if (sym == 90.f) {
    v = fabsf(sub(v, 90.f));
    v = fminf(v, sub(90.f, v));
  } else if (sym == 180.f) {
    v = fminf(v, sub(180.f, v));
  } else if (sym == 360.f) {
    v = 0.f;
  }
  return v;
}

// translation part of getPoseError (:545-547 pow(float, int) promotes to double): independent of the rotations
__device__ __forceinline__ float pose_trans_error(const float* tc, const float* tg) {
  const double dx = (double)sub(tg[0], tc[0]), dy = (double)sub(tg[1], tc[1]), dz = (double)sub(tg[2], tc[2]);
  return (float)__dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
}

// rotation part of getPoseError(test = candidate (inverse rotation iv), gt = g)
__device__ __forceinline__ float pose_rot_error(const float* iv, const float* g, Sym sym) {
  float d[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)  // :524, size-3 redux
      d[3 * i + j] = add(mul(iv[3 * i], g[j]), add(mul(iv[3 * i + 1], g[3 + j]), mul(iv[3 * i + 2], g[6 + j])));
  // :525 Eigen quaternionbase_assign_impl<.,3,3>
  float x, y, z, w;
  float t = add(d[0], add(d[4], d[8]));
  if (t > 0.f) {
    t = sqrt_rn(add(t, 1.0f));
    w = mul(0.5f, t);
    t = fdiv(0.5f, t);
    x = mul(sub(d[7], d[5]), t);
    y = mul(sub(d[2], d[6]), t);
    z = mul(sub(d[3], d[1]), t);
  } else {
    // i = arg max of the diagonal (first maximum), j = (i+1)%3, k = (j+1)%3; spelled out per case
    // so that no array is indexed at run time (scratch)
    int i = 0;
    if (d[4] > d[0]) i = 1;
    if (d[8] > (i ? d[4] : d[0])) i = 2;
#define PGP_QCASE(I, J, K, QI, QJ, QK)                                              \
  {                                                                                 \
    t = sqrt_rn(add(sub(sub(d[4 * I], d[4 * J]), d[4 * K]), 1.0f));                 \
    QI = mul(0.5f, t);                                                              \
    t = fdiv(0.5f, t);                                                              \
    w = mul(sub(d[3 * K + J], d[3 * J + K]), t);                                    \
    QJ = mul(add(d[3 * J + I], d[3 * I + J]), t);                                   \
    QK = mul(add(d[3 * K + I], d[3 * I + K]), t);                                   \
  }
    if (i == 0) PGP_QCASE(0, 1, 2, x, y, z)
    else if (i == 1) PGP_QCASE(1, 2, 0, y, z, x)
    else PGP_QCASE(2, 0, 1, z, x, y)
#undef PGP_QCASE
  }
  // :335-356 toEulerianAngle: float products and sums, the rest in double
  const double sinr = 2.0 * (double)add(mul(w, x), mul(y, z));
  const double cosr = 1.0 - 2.0 * (double)add(mul(x, x), mul(y, y));
  const float e0 = (float)atan2(sinr, cosr);
  const double sinp = 2.0 * (double)sub(mul(w, y), mul(z, x));
  const float e1 = fabs(sinp) >= 1 ? (float)copysign(1.57079632679489661923, sinp) : (float)asin(sinp);
  const double siny = 2.0 * (double)add(mul(w, z), mul(x, y));
  const double cosy = 1.0 - 2.0 * (double)add(mul(y, y), mul(z, z));
  const float e2 = (float)atan2(siny, cosy);
  return fdiv(add(add(fold(e0, sym.x), fold(e1, sym.y)), fold(e2, sym.z)), 3.0f);
}

// getPoseError(test = candidate (inverse rotation iv, translation tc), gt = g / tg)
__device__ __forceinline__ void pose_error(const float* iv, const float* tc, const float* g, const float* tg,
                                           Sym sym, float* rot_err, float* trans_err) {
  *rot_err = pose_rot_error(iv, g, sym);
  *trans_err = pose_trans_error(tc, tg);
}

__global__ __launch_bounds__(256) void pose_error_pairs(const float* __restrict__ test, const float* __restrict__ gt,
                                                        int n, Sym sym, float* __restrict__ rot_err,
                                                        float* __restrict__ trans_err) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float* a16 = test + 16 * (size_t)p;
  const float* g16 = gt + 16 * (size_t)p;
  float a[9], g[9], iv[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      a[3 * i + j] = a16[i + 4 * j];
      g[3 * i + j] = g16[i + 4 * j];
    }
  inverse3(a, iv);
  const float tc[3] = {a16[12], a16[13], a16[14]}, tg[3] = {g16[12], g16[13], g16[14]};
  float re, te;
  pose_error(iv, tc, g, tg, sym, &re, &te);
  rot_err[p] = re;
  trans_err[p] = te;
}

// bits[c * W + w] bit b = candidate c is within the thresholds of sorted pose r = 64 w + b, r < c.
// One block per candidate; its 4 waves stride over the words of the row.
__global__ __launch_bounds__(256) void cluster_pair_bits(const float* __restrict__ inv_c,
                                                         const float* __restrict__ rot_r, int m, int W, Sym sym,
                                                         float rot_thresh, float trans_thresh,
                                                         unsigned long long* __restrict__ bits) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* cp = inv_c + (size_t)kPoseStride * c;
  float iv[9], tc[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) iv[k] = cp[k];
  tc[0] = cp[9];
  tc[1] = cp[10];
  tc[2] = cp[11];
  const int nw = (c + 63) >> 6;  // words holding some r < c
  for (int w = wave; w < nw; w += 4) {
    const int r = 64 * w + lane;
    // The translation test first: it does not involve the rotations, and among scored hypotheses few pairs are
    // within centimetres of each other -- a wave none of whose 64 pairs passes it skips the quaternion and the
    // three double-precision inverse trigonometric functions of the rotation error altogether.
    bool hit = false;
    if (r < c) {
      float tg[3];
      tg[0] = rot_r[(size_t)9 * m + r];
      tg[1] = rot_r[(size_t)10 * m + r];
      tg[2] = rot_r[(size_t)11 * m + r];
      hit = pose_trans_error(tc, tg) < trans_thresh;
    }
    if (__ballot(hit) != 0ull) {
      if (hit) {
        float g[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k] = rot_r[(size_t)k * m + r];
        hit = pose_rot_error(iv, g, sym) < rot_thresh;  // HypothesisSelection.cpp:99: both below their thresholds
      }
    }
    const unsigned long long b = __ballot(hit);
    if (lane == 0) bits[(size_t)c * W + w] = b;
  }
  // the diagonal word of a candidate with c % 64 == 0 holds no r < c: define it
  if ((c & 63) == 0 && threadIdx.x == 0) bits[(size_t)c * W + (c >> 6)] = 0ull;
}

// The sequential pass.  One block (16 waves).  keep[] (LDS) = representatives so far, by sorted position.
// Tile t = candidates 64 t .. 64 t + 63.  A candidate joins the representative at the LOWEST position it is
// adjacent to, else it becomes one; its row of the bit matrix is needed (a) against the representatives of
// tiles 0 .. t-2, (b) against those of tile t-1, (c) against the tile's own earlier candidates (the diagonal
// word).  Only (c) is serial.  The work is software-pipelined across tiles with ONE barrier per tile:
//   wave 0, iteration t    : (b) with the word it prefetched, then requests tile t+1's two words, then the
//                            64-step walk (c) of tile t under those loads' latency
//   waves 1-15, iteration t: (a) for tile t+1 -- keep[0 .. t-1] are final -- with the rows they requested
//                            during iteration t-1; then request tile t+2's rows
// so no global round trip is ever waited for with nothing else to do (a single resident block has no other
// wave to hide it behind).  911 us (round 1) -> 449 us (16 waves, batched row loads, v_readlane in the walk)
// -> pipelined: see DESIGN.md section 4.
constexpr int kGreedyThreads = 1024;
constexpr int kGreedyA = kGreedyThreads / 64 - 1;           // 15 look-ahead waves
constexpr int kGreedyPer = (64 + kGreedyA - 1) / kGreedyA;  // 5 candidates each
__global__ __launch_bounds__(kGreedyThreads) void cluster_greedy(const unsigned long long* __restrict__ bits, int m, int W,
                                                      const int* __restrict__ idx_sorted, int* __restrict__ rep_out,
                                                      int* __restrict__ assign, int* __restrict__ n_rep_out) {
  extern __shared__ unsigned long long keep[];  // W words
  __shared__ int pre[2][64];                    // first hit among tiles <= t-2 per candidate of tile t, or -1
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_tiles = (m + 63) >> 6;
  for (int w = threadIdx.x; w < W; w += blockDim.x) keep[w] = 0ull;
  if (threadIdx.x < 128) pre[threadIdx.x >> 6][threadIdx.x & 63] = -1;
  __syncthreads();

  // ---- wave 0: the walk ----
  unsigned long long w_prev = 0ull, w_diag = 0ull;   // candidate's words t-1 and t (tile t), prefetched
  unsigned long long kw_prev = 0ull;                 // keep[t-1]
  int n_rep = 0;
  // ---- waves 1-15: rows (first 64 words) of the candidates they will test next ----
  unsigned long long x_next[kGreedyPer];
#pragma unroll
  for (int k = 0; k < kGreedyPer; ++k) x_next[k] = 0ull;
  if (wave == 0) {
    const int c = lane;
    w_diag = c < m ? bits[(size_t)c * W + 0] : 0ull;
  } else {
    // rows of tile 1 (tested during iteration 0 against nothing: tile 1 has no tile <= -1 ... kept uniform)
#pragma unroll
    for (int k = 0; k < kGreedyPer; ++k) {
      const int slot = (wave - 1) + kGreedyA * k, c = 64 + slot;
      x_next[k] = (slot < 64 && c < m && lane < W) ? bits[(size_t)c * W + lane] : 0ull;
    }
  }

  for (int t = 0; t < n_tiles; ++t) {
    const int c0 = t << 6;
    if (wave == 0) {
      const int c = c0 + lane;
      int my_pre = pre[t & 1][lane];                       // (a), computed during iteration t-1
      if (my_pre < 0 && t >= 1) {                          // (b)
        const unsigned long long x = w_prev & kw_prev;
        if (x) my_pre = 64 * (t - 1) + (__ffsll((long long)x) - 1);
      }
      const unsigned long long diag = w_diag;
      // tile t+1's words t and t+1, in flight under the walk
      const int cn = c + 64;
      if (t + 1 < n_tiles) {
        w_prev = cn < m ? bits[(size_t)cn * W + t] : 0ull;
        w_diag = cn < m ? bits[(size_t)cn * W + t + 1] : 0ull;
      }
      unsigned long long kw = 0ull;
      int my_assign = my_pre;
      const int jn = m - c0 < 64 ? m - c0 : 64;
      for (int j = 0; j < jn; ++j) {
        // j is wave-uniform: v_readlane (a few cycles) instead of a cross-lane shuffle through the LDS
        // crossbar (a dependent ~100-cycle trip, twice per step, 64 steps per tile)
        const unsigned dlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(diag & 0xFFFFFFFFull), j);
        const unsigned dhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(diag >> 32), j);
        const unsigned long long dj = (((unsigned long long)dhi << 32) | dlo) & kw;
        const int pj = __builtin_amdgcn_readlane(my_pre, j);
        int a;
        if (pj >= 0) {
          a = pj;
        } else if (dj) {
          a = c0 + (__ffsll((long long)dj) - 1);
        } else {
          a = c0 + j;
          kw |= 1ull << j;
          if (lane == 0) rep_out[n_rep] = idx_sorted[c0 + j];
          ++n_rep;
        }
        if (lane == j) my_assign = a;
      }
      if (c < m) assign[idx_sorted[c]] = idx_sorted[my_assign];
      if (lane == 0) keep[t] = kw;
      kw_prev = kw;
    } else if (t + 1 < n_tiles) {
      // (a) for tile t+1: words 0 .. t-1
      unsigned long long x_load[kGreedyPer];
#pragma unroll
      for (int k = 0; k < kGreedyPer; ++k) {   // tile t+2's rows, used in the next iteration
        const int slot = (wave - 1) + kGreedyA * k, c = c0 + 128 + slot;
        x_load[k] = (slot < 64 && c < m && lane < W) ? bits[(size_t)c * W + lane] : 0ull;
      }
      const unsigned long long kp0 = lane < t && lane < W ? keep[lane] : 0ull;   // words < t only
#pragma unroll
      for (int k = 0; k < kGreedyPer; ++k) {
        const int slot = (wave - 1) + kGreedyA * k;   // wave-uniform
        if (slot >= 64) continue;
        const int c = c0 + 64 + slot;
        int first = -1;
        if (c < m) {
          unsigned long long xk = x_next[k] & kp0;
          unsigned long long any = __ballot(xk != 0ull);
          if (any) {
            const int src = __ffsll((long long)any) - 1;
            first = __shfl(64 * lane + (__ffsll((long long)xk) - 1), src, 64);
          } else {
            for (int wb = 64; wb < t; wb += 64) {   // more than 4096 poses: the words past the prefetched 64
              const int w = wb + lane;
              xk = (w < t) ? (bits[(size_t)c * W + w] & keep[w]) : 0ull;
              any = __ballot(xk != 0ull);
              if (any) {
                const int src = __ffsll((long long)any) - 1;
                first = __shfl(64 * w + (__ffsll((long long)xk) - 1), src, 64);
                break;
              }
            }
          }
        }
        if (lane == 0) pre[(t + 1) & 1][slot] = first;
      }
#pragma unroll
      for (int k = 0; k < kGreedyPer; ++k) x_next[k] = x_load[k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_rep_out = n_rep;
}

// ---- the same pass for m <= 4096 poses (W <= 64 words): every candidate's state lives in a register ----------
// Thread tid owns candidates tid, tid + 1024, ... (<= 4): `pre` = position of the first representative of an
// EARLIER tile it is adjacent to, or -1.  Tile t's 64 candidates are then the lanes of ONE wave, wave t mod 16,
// which walks the tile; the walk itself is the minimal recurrence
//     kw |= (diag_j & kw) == 0 && !pre_j ? 1 << j : 0          (j = 0 .. 63)
// on the scalar unit, fully unrolled (two v_readlane + s_and / s_cselect / s_or per step, no branch), since a
// candidate's diagonal word only holds bits below j: assignments and the representative list follow from the
// final kw in parallel.  After the barrier every thread PUSHES tile t's representatives onto its own later
// candidates (word t of their rows, requested one iteration earlier; two register sets alternate so that no
// copy waits for a load).  The barrier waits for LDS only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads()
// would drain the prefetches.  449 us (two barriers, a branchy 64-step walk) -> see DESIGN.md section 4.
constexpr int kSmallMax = 4096;
__global__ __launch_bounds__(kGreedyThreads) void cluster_greedy_small(const unsigned long long* __restrict__ bits, int m, int W,
                                                            const int* __restrict__ idx_sorted, int* __restrict__ rep_out,
                                                            int* __restrict__ assign, int* __restrict__ n_rep_out) {
  __shared__ unsigned long long keep[kSmallMax / 64];
  __shared__ int n_rep_s;
  __shared__ int s_idx[kSmallMax];   // idx_sorted: the walker's stores must not wait for a dependent global load
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  const int n_tiles = (m + 63) >> 6;
  for (int c = tid; c < m; c += kGreedyThreads) s_idx[c] = idx_sorted[c];
  constexpr int K = kSmallMax / kGreedyThreads;   // 4 candidates per thread
  int pre[K];
#pragma unroll
  for (int k = 0; k < K; ++k) pre[k] = -1;
  if (tid == 0) n_rep_s = 0;
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // word `t` of the rows of this thread's candidates that lie in tiles > t (0 otherwise)
  auto load_words = [&](int t, unsigned long long (&wd)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int c = tid + kGreedyThreads * k;
      wd[k] = (t < n_tiles && c < m && c >= 64 * (t + 1)) ? bits[(size_t)c * W + t] : 0ull;
    }
  };
  // diagonal word of the candidate this lane holds in tile t (if this wave walks tile t)
  auto load_diag = [&](int t) -> unsigned long long {
    const int c = 64 * t + lane;
    return (t < n_tiles && (t & 15) == wave && c < m) ? bits[(size_t)c * W + t] : 0ull;
  };
  auto iteration = [&](const int t, const unsigned long long (&w_prev)[K], unsigned long long (&w_next)[K],
                       const unsigned long long diag, unsigned long long& diag_next) {
    // 1. push tile t-1's representatives (keep[t-1] is final since the last barrier)
    if (t >= 1) {
      const unsigned long long kp = keep[t - 1];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const unsigned long long x = w_prev[k] & kp;
        if (pre[k] < 0 && x) pre[k] = 64 * (t - 1) + (__ffsll((long long)x) - 1);
      }
    }
    // 2. requests for the next iteration's push and the next walk of this wave
    load_words(t, w_next);
    diag_next = load_diag(t + 1);
    // 3. the walk, by the wave that owns tile t
    if ((t & 15) == wave) {
      const int kk = t >> 4, c0 = t << 6, c = c0 + lane;
      const int my_pre = kk == 0 ? pre[0] : kk == 1 ? pre[1] : kk == 2 ? pre[2] : pre[3];
      const unsigned long long cannot = __ballot(my_pre >= 0 || c >= m);   // lanes that cannot become representatives
      const unsigned dlo = (unsigned)(diag & 0xFFFFFFFFull), dhi = (unsigned)(diag >> 32);
      unsigned long long kw = 0ull;
      if (__ballot(((cannot >> lane) & 1ull) == 0ull && (diag & ~cannot) != 0ull) == 0ull) {
        // no eligible candidate of the tile is adjacent to an eligible earlier one of the same tile (the usual case:
        // adjacency is sparse): every eligible candidate becomes a representative, no walk
        kw = ~cannot;
      } else {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
          const unsigned long long dj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, j) << 32) |
                                        (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dlo, j);
          const unsigned long long bit = ~cannot & (1ull << j);
          kw |= (dj & kw) ? 0ull : bit;
        }
      }
      // assignments and representatives from the final kw (bits >= j of diag_j are zero)
      const unsigned long long dk = diag & kw;
      const int a = my_pre >= 0 ? my_pre : dk ? c0 + (__ffsll((long long)dk) - 1) : c;
      if (c < m) assign[s_idx[c]] = s_idx[a];
      const int n0 = n_rep_s;
      if ((kw >> lane) & 1ull) rep_out[n0 + __popcll(kw & ((1ull << lane) - 1ull))] = s_idx[c];
      if (lane == 0) {
        keep[t] = kw;
        n_rep_s = n0 + __popcll(kw);
      }
    }
    lds_barrier();
  };
  unsigned long long wA[K], wB[K], dA, dB;
#pragma unroll
  for (int k = 0; k < K; ++k) wA[k] = wB[k] = 0ull;
  dA = load_diag(0);
  dB = 0ull;
  __syncthreads();
  for (int t = 0; t < n_tiles; t += 2) {
    iteration(t, wA, wB, dA, dB);
    if (t + 1 < n_tiles) iteration(t + 1, wB, wA, dB, dA);
  }
  if (tid == 0) *n_rep_out = n_rep_s;
}

// ---- few clusters: one round per representative ---------------------------------------------------------------
// With the reference's pruning (scores above half the best) the survivors sit around a handful of poses: 3612
// survivors in 3 clusters on the configs[2] workload, for which the bit matrix holds 6.5 M pair tests and the walk
// 57 tiles.  The greedy pass is then cheaper ROUND BY ROUND: the first unassigned candidate in score order is the
// next representative (every earlier one has joined an earlier representative), every unassigned later candidate is
// tested against it alone and joins it if within the thresholds -- the lowest-positioned representative it is
// adjacent to, as the reference's loop over the clusters in creation order finds it (HypothesisSelection.cpp:88-109).
// One workgroup; a round costs the tests of the unassigned candidates (translation first) + one barrier pair.
// Stops after `max_rounds` representatives with *done = 0: the caller then runs the bit-matrix pass instead.
constexpr int kRoundThreads = 1024;
constexpr int kRoundMaxM = 16 * kRoundThreads;   // one assigned-bit per candidate in a 16-bit mask per thread
__global__ __launch_bounds__(kRoundThreads) void cluster_rounds(const float* __restrict__ inv_c, const float* __restrict__ rot_r,
                                                                int m, Sym sym, float rot_thresh, float trans_thresh,
                                                                const int* __restrict__ idx_sorted, int max_rounds,
                                                                int* __restrict__ rep_out, int* __restrict__ assign,
                                                                int* __restrict__ n_rep_out, int* __restrict__ done) {
  __shared__ int s_left;
  __shared__ int s_next[3];   // round k reduces into slot k % 3; slot (k + 2) % 3 is re-armed meanwhile
  const int tid = threadIdx.x;
  unsigned assigned = 0u;   // bit k: candidate tid + 1024 k has a representative (or does not exist)
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (tid + kRoundThreads * k >= m) assigned |= 1u << k;
  if (tid < 3) s_next[tid] = 0x7FFFFFFF;
  __syncthreads();
  int r = 0, n_rep = 0;
  for (;;) {
    // candidate r is the next representative
    const int rid = idx_sorted[r];
    if (tid == (r & (kRoundThreads - 1))) {
      assigned |= 1u << (r / kRoundThreads);
      rep_out[n_rep] = rid;
      assign[rid] = rid;
    }
    ++n_rep;
    float g[9], tg[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = rot_r[(size_t)k * m + r];
    tg[0] = rot_r[(size_t)9 * m + r];
    tg[1] = rot_r[(size_t)10 * m + r];
    tg[2] = rot_r[(size_t)11 * m + r];
    int first_left = 0x7FFFFFFF;
    for (int k = 0; k < 16; ++k) {
      if ((assigned >> k) & 1u) continue;
      const int c = tid + kRoundThreads * k;   // c > r: every candidate before r is assigned
      const float* cp = inv_c + (size_t)kPoseStride * c;
      const float tc[3] = {cp[9], cp[10], cp[11]};
      bool hit = pose_trans_error(tc, tg) < trans_thresh;
      if (hit) {
        float iv[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) iv[q] = cp[q];
        hit = pose_rot_error(iv, g, sym) < rot_thresh;
      }
      if (hit) {
        assigned |= 1u << k;
        assign[idx_sorted[c]] = rid;
      } else if (c < first_left) {
        first_left = c;
      }
    }
    // the first candidate nobody has taken yet
    int* slot = &s_next[n_rep % 3];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) first_left = min(first_left, __shfl_xor(first_left, off, 64));
    if ((tid & 63) == 0 && first_left != 0x7FFFFFFF) atomicMin(slot, first_left);
    __syncthreads();
    r = *slot;
    // many clusters ahead (eight representatives have taken less than half of the candidates): leave early
    if (n_rep == 8 && r != 0x7FFFFFFF) {
      if (tid == 0) s_left = 0;
      __syncthreads();
      int left = __popc(~assigned & 0xFFFFu);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) left += __shfl_xor(left, off, 64);
      if ((tid & 63) == 0) atomicAdd(&s_left, left);
      __syncthreads();
      if (s_left > m / 2) n_rep = max_rounds;   // reported as "not done" below
    }
    // the slot of the round after next: last read before this barrier, next written after the next one
    if (tid == 0) s_next[(n_rep + 2) % 3] = 0x7FFFFFFF;
    if (r == 0x7FFFFFFF || n_rep >= max_rounds) break;
  }
  if (tid == 0) {
    *n_rep_out = n_rep;
    *done = r == 0x7FFFFFFF ? 1 : 0;
  }
}

}  // namespace

int launch_pose_error(pgp_ctx* ctx, const float* d_test, const float* d_gt, int n, const float sym[3],
                      float* d_rot, float* d_trans, hipStream_t st) {
  (void)ctx;
  if (n <= 0) return PGP_OK;
  const Sym s = {sym[0], sym[1], sym[2]};
  hipLaunchKernelGGL(pose_error_pairs, dim3((n + 255) / 256), dim3(256), 0, st, d_test, d_gt, n, s, d_rot, d_trans);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

// d_T [n][16], d_scores [n] -> d_rep [n] (first *n_rep entries valid), d_assign [n], d_counts {m, n_rep}
int launch_cluster(pgp_ctx* ctx, const float* d_T, const float* d_scores, int n, float best_score,
                   const float sym[3], const pgp_cluster_params* prm, int* d_rep, int* d_assign,
                   int* h_m, int* h_n_rep, hipStream_t st) {
  *h_m = 0;
  *h_n_rep = 0;
  if (n <= 0) return PGP_OK;
  const float bar = prm->accept_fraction * best_score;  // float product as at HypothesisSelection.cpp:75
  size_t sort_bytes = 0;
  hipError_t he = rocprim::radix_sort_keys(nullptr, sort_bytes, (unsigned long long*)nullptr,
                                           (unsigned long long*)nullptr, (size_t)n, 0, 64, st);
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_keys (size query) failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  int rc = ctx->d_cl_keys.ensure((size_t)n * 16 + sort_bytes + 512);
  if (rc != PGP_OK) return rc;
  unsigned long long* keys_in = ctx->d_cl_keys.as<unsigned long long>();
  unsigned long long* keys_out = keys_in + n;
  int* d_cnt = reinterpret_cast<int*>(keys_out + n);  // {m, n_rep}
  void* sort_tmp = reinterpret_cast<unsigned char*>(d_cnt) + 256;
  PGP_HIP(hipMemsetAsync(d_cnt, 0, 4 * sizeof(int), st));
  PGP_HIP(hipMemsetAsync(d_assign, 0xFF, (size_t)n * sizeof(int), st));
  hipLaunchKernelGGL(cluster_keys, dim3((n + 255) / 256), dim3(256), 0, st, d_scores, n, bar, keys_in, d_cnt);
  int m = 0;
  {
    HostOut out(ctx, st);   // (a 4-byte copy into pageable memory keeps the host 12 us longer than one into pinned memory)
    if ((rc = out.to(&m, d_cnt, sizeof(int))) != PGP_OK || (rc = out.sync()) != PGP_OK) return rc;
  }
  *h_m = m;
  if (m == 0) return PGP_OK;
  // the m survivors only (typically a tenth of the batch: one block sort instead of a dozen merge passes)
  size_t sort_bytes_m = sort_bytes;
  he = rocprim::radix_sort_keys(sort_tmp, sort_bytes_m, keys_in, keys_out, (size_t)m, 0, 64, st);
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_keys failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  const int W = (m + 63) / 64;
  if ((size_t)W * 8 > 60 * 1024) {  // keep[] lives in LDS
    set_error("pgp_cluster_poses: %d hypotheses pass the score bar; at most %d are supported", m, 60 * 1024 / 8 * 64);
    return PGP_EINVAL;
  }
  const size_t pose_bytes = ((size_t)m * kPoseStride * 4 + 255) & ~(size_t)255;
  const size_t idx_bytes = ((size_t)m * 4 + 255) & ~(size_t)255;
  if ((rc = ctx->d_cl_ws.ensure(2 * pose_bytes + idx_bytes + (size_t)m * W * 8)) != PGP_OK) return rc;
  unsigned char* base = ctx->d_cl_ws.as<unsigned char>();
  float* inv_c = reinterpret_cast<float*>(base);
  float* rot_r = reinterpret_cast<float*>(base + pose_bytes);
  int* idx_sorted = reinterpret_cast<int*>(base + 2 * pose_bytes);
  unsigned long long* bits = reinterpret_cast<unsigned long long*>(base + 2 * pose_bytes + idx_bytes);
  hipLaunchKernelGGL(cluster_gather, dim3((m + 255) / 256), dim3(256), 0, st, keys_out, d_T, m, inv_c, rot_r,
                     idx_sorted);
  const Sym s = {sym[0], sym[1], sym[2]};
  // few clusters (the usual outcome of the reference's pruning): one round per representative
  int max_rounds = 64;
  if (const char* v = getenv("PGP_CLUSTER_ROUNDS")) max_rounds = atoi(v);   // A/B knob; 0 = bit matrix only
  if (m <= kRoundMaxM && max_rounds > 0) {
    hipLaunchKernelGGL(cluster_rounds, dim3(1), dim3(kRoundThreads), 0, st, (const float*)inv_c, (const float*)rot_r, m, s,
                       prm->rot_thresh_deg, prm->trans_thresh, (const int*)idx_sorted, max_rounds, d_rep, d_assign,
                       d_cnt + 1, d_cnt + 2);
    PGP_HIP(hipGetLastError());
    int res[2] = {0, 0};
    {
      HostOut out(ctx, st);
      if ((rc = out.to(res, d_cnt + 1, 2 * sizeof(int))) != PGP_OK || (rc = out.sync()) != PGP_OK) return rc;
    }
    if (res[1]) {
      *h_n_rep = res[0];
      return PGP_OK;
    }
    // more clusters than rounds: start over with the bit matrix (it overwrites every assignment made so far)
    PGP_HIP(hipMemsetAsync(d_assign, 0xFF, (size_t)n * sizeof(int), st));
  }
  hipLaunchKernelGGL(cluster_pair_bits, dim3(m), dim3(256), 0, st, inv_c, rot_r, m, W, s, prm->rot_thresh_deg,
                     prm->trans_thresh, bits);
  if (m <= kSmallMax)
    hipLaunchKernelGGL(cluster_greedy_small, dim3(1), dim3(kGreedyThreads), 0, st, bits, m, W, idx_sorted, d_rep, d_assign,
                       d_cnt + 1);
  else
    hipLaunchKernelGGL(cluster_greedy, dim3(1), dim3(kGreedyThreads), (size_t)W * 8, st, bits, m, W, idx_sorted, d_rep,
                       d_assign, d_cnt + 1);
  PGP_HIP(hipGetLastError());
  HostOut out(ctx, st);
  if ((rc = out.to(h_n_rep, d_cnt + 1, sizeof(int))) != PGP_OK) return rc;
  return out.sync();
}

}  // namespace pgp
