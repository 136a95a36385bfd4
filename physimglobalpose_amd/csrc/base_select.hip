// csrc/base_select.hip -- Step 1 of Perform_N_steps on gfx950: stochastic base selection.
//
// Replaces, for MANY independent attempts at once (the reference seeds a fresh engine inside every
// call, base.cc:613-614, so attempts do not share state):
//   Match4PCSBase::SelectQuadrilateralStoCS   S4/algorithms/match4pcsBase.cc:600-792
//   Match4PCSBase::computePPF                 :582-598        approximate_bin  :150-160
//   Match4PCSBase::TryQuadrilateral           :415-464        distSegmentToSegment  :81-148
//   the presence test PPFMap->find(ppf_) of the model's pair-feature table (PPE/data_layer/Objects.cpp:31-49)
//
// Mapping: one 256-thread workgroup per attempt walks the three weighting loops -- a lane per scene
// point computes the point-pair feature against the last chosen base point (one cross product, three
// dot products, a correctly rounded square root and a division per angle), probes the device hash
// set of the model's feature keys and writes the stage weight; the reference's sequential float
// `sum_probabilities` is reproduced by ONE wave that loads 64 weights per instruction and adds them
// lane by lane (v_readlane + v_add: the additions stay in the reference's order, the loads are
// coalesced); the draw (std::discrete_distribution) is an inverse-CDF look-up on a block-wide double
// prefix sum with a uniform variate the HOST drew from its engine (two engine calls per variate,
// exactly what discrete_distribution::operator() consumes); TryQuadrilateral's twelve pairings run
// on twelve lanes.  One launch returns every attempt's four ids, its invariants and a status.
//
// Float parity (pinned on the Eigen-typed harness, tests/golden/stocs.npz): stage weights are
// bit-identical to the reference loops', including the normalisation by the sequential sum.
// The three angles of a feature go through atan2f in the reference.  glibc evaluates
// atan2f(y >= 0, x) as a function of the single float r = |y / x| and the sign of x; the feature only
// needs the 10-degree bin of int(angle * 180 / pi), so the host finds, by bisection over ITS OWN
// atan2f (ppf_thresholds), the nine ratios at which that bin changes for x > 0 and for x < 0, and the
// device compares the correctly rounded ratio against them: no device transcendental is involved
// and the bins agree with the libm of the machine the node runs on.  The same device is used for
// the 30-degree internal-angle test of the third point (acosf of an un-normalised dot product,
// base.cc:664-665): the thresholds of lcp_score.hip's gate.

#include "pgp_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace pgp {

namespace {

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float sqrt_rn(float z) { return (float)__dsqrt_rn((double)z); }
__device__ __forceinline__ float sum3(float a, float b, float c) { return add(a, add(b, c)); }  // Eigen: a + (b + c)
__device__ __forceinline__ float dot(V3 a, V3 b) { return sum3(mul(a.x, b.x), mul(a.y, b.y), mul(a.z, b.z)); }
__device__ __forceinline__ float norm(V3 v) { return sqrt_rn(dot(v, v)); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {sub(a.x, b.x), sub(a.y, b.y), sub(a.z, b.z)}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {sub(mul(a.y, b.z), mul(a.z, b.y)), sub(mul(a.z, b.x), mul(a.x, b.z)), sub(mul(a.x, b.y), mul(a.y, b.x))};
}
__device__ __forceinline__ V3 ld3(const float4* __restrict__ a, int i) {
  const float4 v = a[i];
  return {v.x, v.y, v.z};
}

struct PpfTable {
  const unsigned long long* keys;  // open addressing, ~0ull = empty
  const uint32_t* value;           // key index (row of the CSR pair lists)
  uint32_t mask;
  int shift;
  float tpos[9], tneg[9];          // ratio thresholds of the 10-degree bins, x > 0 / x < 0
  int trans_disc;                  // 5 (mm)
};

__host__ __device__ inline int approximate_bin(int val, int disc) {  // base.cc:150-160
  const int lower = val - (val % disc), upper = lower + disc;
  return (val - lower < upper - val) ? lower : upper;
}

// approximate_bin(int(atan2f(y, x) * 180 / M_PI), 10) for y >= 0; -1 when the reference's value
// cannot be a table key (NaN inputs)
__device__ __forceinline__ int angle_bin(const PpfTable& t, float y, float x) {
  if (!(y == y) || !(x == x)) return -1;
  if (y == 0.f) return (__float_as_uint(x) >> 31) ? 180 : 0;   // atan2f(0, -0 or negative) = pi
  if (x == 0.f) return 90;
  const float r = __fdiv_rn(y, fabsf(x));
  int c = 0;
  if (x > 0.f) {
#pragma unroll
    for (int b = 0; b < 9; ++b) c += r >= t.tpos[b] ? 1 : 0;
    return 10 * c;
  }
#pragma unroll
  for (int b = 0; b < 9; ++b) c += r >= t.tneg[b] ? 1 : 0;
  return 180 - 10 * c;
}

// computePPF(i1, i2) packed as f1 << 24 | f2 << 16 | f3 << 8 | f4, or ~0ull when it is no key
__device__ __forceinline__ unsigned long long ppf_key(const PpfTable& t, V3 p1, V3 n1, V3 p2, V3 n2, int* f) {
  const V3 u = vsub(p1, p2);
  const int f1 = approximate_bin((int)mul(norm(u), 1000.0f), t.trans_disc);
  const int f2 = angle_bin(t, norm(cross(n1, u)), dot(n1, u));
  const int f3 = angle_bin(t, norm(cross(n2, u)), dot(n2, u));
  const int f4 = angle_bin(t, norm(cross(n1, n2)), dot(n1, n2));
  if (f) {
    f[0] = f1;
    f[1] = f2;
    f[2] = f3;
    f[3] = f4;
  }
  if (f1 < 0 || f2 < 0 || f3 < 0 || f4 < 0) return ~0ull;
  return ((unsigned long long)(unsigned)f1 << 24) | ((unsigned long long)f2 << 16) | ((unsigned long long)f3 << 8) |
         (unsigned long long)f4;
}

__device__ __forceinline__ int table_find(const PpfTable& t, unsigned long long key) {
  if (key == ~0ull || t.mask == 0u) return -1;
  uint32_t s = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> t.shift) & t.mask;
  for (;;) {   // load factor <= 0.5: an empty slot always ends the probe
    const unsigned long long k = t.keys[s];
    if (k == key) return (int)t.value[s];
    if (k == ~0ull) return -1;
    s = (s + 1) & t.mask;
  }
}

struct SelectArgs {
  const float4* P;     // {x, y, z, id}
  const float4* Pnw;   // {nx, ny, nz, prob}
  int n;
  PpfTable tab;
  float gate_lo, gate_hi;   // int_angle < 30 <=> dot in [gate_lo, 1] or [-1, gate_hi]
  const double* prob_cdf;   // inclusive double prefix sums of prob [n] (the first draw)
  const double* u;          // [n_attempts][4] uniform variates in [0, 1)
  float* cur;               // [n_attempts][n] curr_probabilities_
  int4* ids;
  float2* inv;
  int* status;              // 1 ok; 0 no candidate left at stage 2 / 3 / 4 (the reference returns false)
  int2* rows;               // nullable: the table rows of the base's two edges (ids 0-1, ids 2-3), what ExtractCongruentSet looks up
};

// ---- the three weighting loops, one point per call (base.cc:625-652, 662-699, 713-769) ----------
__device__ __forceinline__ float stage2_weight(const SelectArgs& a, int i, float cur_i, int b1, V3 pb, V3 nb,
                                               float prob_b) {
  if (i == b1 || cur_i == 0.f) return 0.f;
  const float4 pn = a.Pnw[i];
  const unsigned long long key = ppf_key(a.tab, pb, nb, ld3(a.P, i), {pn.x, pn.y, pn.z}, nullptr);
  const float edge = table_find(a.tab, key) >= 0 ? 1.f : 0.f;
  return mul(mul(pn.w, prob_b), edge);   // orig[i] * orig[base1] * edge
}

__device__ __forceinline__ float stage3_weight(const SelectArgs& a, int i, float cur_i, int b1, int b2, V3 p1,
                                               V3 v_1, V3 pb, V3 nb, float prob_b) {
  const V3 v_2 = vsub(ld3(a.P, i), p1);
  const float d = dot(v_1, v_2);
  // int_angle = min(a, 180 - a) < 30 with a = acosf(d) * 180 / pi: thresholds on d (NaN: not excluded)
  const bool small_angle = (d >= a.gate_lo && d <= 1.0f) || (d <= a.gate_hi && d >= -1.0f);
  if (i == b1 || i == b2 || cur_i == 0.f || small_angle) return 0.f;
  const float4 pn = a.Pnw[i];
  const unsigned long long key = ppf_key(a.tab, pb, nb, ld3(a.P, i), {pn.x, pn.y, pn.z}, nullptr);
  const float edge = table_find(a.tab, key) >= 0 ? 1.f : 0.f;
  return mul(mul(cur_i, prob_b), edge);  // cur[i] * orig[base2] * edge
}

struct Plane {
  float denom, A, B, C;
};
__device__ __forceinline__ Plane fit_plane(V3 q1, V3 q2, V3 q3) {  // base.cc:731-745, double arithmetic
  const double x1 = q1.x, y1 = q1.y, z1 = q1.z, x2 = q2.x, y2 = q2.y, z2 = q2.z, x3 = q3.x, y3 = q3.y, z3 = q3.z;
  Plane p;
  p.denom = (float)(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
  const double dd = (double)p.denom;
  p.A = (float)((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / dd);
  p.B = (float)((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / dd);
  p.C = (float)((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / dd);
  return p;
}

__device__ __forceinline__ float stage4_weight(const SelectArgs& a, int i, float cur_i, int b1, int b2, int b3,
                                               V3 q1, V3 q2, V3 q3, Plane pl, V3 nb, float prob_b) {
  if (i == b1 || i == b2 || i == b3 || cur_i == 0.f) return 0.f;
  const V3 p = ld3(a.P, i);
  if (pl.denom != 0.f) {
    // Scalar planar_distance = std::abs(A*x + B*y + C*z - 1.0): float products and sums, double minus
    const float lin = add(add(mul(pl.A, p.x), mul(pl.B, p.y)), mul(pl.C, p.z));
    const float planar = (float)fabs((double)lin - 1.0);
    if ((double)planar > 0.01 || (double)norm(vsub(p, q1)) < 0.01 || (double)norm(vsub(p, q2)) < 0.01 ||
        (double)norm(vsub(p, q3)) < 0.01)
      return 0.f;
  }
  const float4 pn = a.Pnw[i];
  const unsigned long long key = ppf_key(a.tab, q3, nb, p, {pn.x, pn.y, pn.z}, nullptr);
  const float edge = table_find(a.tab, key) >= 0 ? 1.f : 0.f;
  return mul(mul(cur_i, prob_b), edge);  // cur[i] * orig[base3] * edge
}

// Sequential float sum of w[0..n) in index order, by wave 0 (call with all threads of the block):
// 64 coalesced loads per trip, then 64 ordered additions (v_readlane + v_add); `sum += 0` is the
// identity, so skipping nothing changes nothing.  Result valid in every thread after the barrier.
__device__ float sequential_sum(const float* __restrict__ w, int n, float* s_bcast) {
  if (threadIdx.x < 64) {
    float S = 0.f;
    for (int base = 0; base < n; base += 64) {
      const int i = base + (int)threadIdx.x;
      const float v = i < n ? w[i] : 0.f;
      unsigned long long m = __ballot(v != 0.f);
      if (m == 0ull) continue;
      if (__popcll(m) > 40) {
#pragma unroll
        for (int k = 0; k < 64; ++k) S = __fadd_rn(S, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)));
      } else {
        // few survivors of the stage's constraints (the usual case after stage 2): only the non-zero lanes, in lane order
        // -- the chain of dependent adds is what this sum costs (~10 cycles each)
        while (m) {
          const int k = __builtin_ctzll(m);
          m &= m - 1ull;
          S = __fadd_rn(S, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)));
        }
      }
    }
    if (threadIdx.x == 0) *s_bcast = S;
  }
  __syncthreads();
  const float r = *s_bcast;
  __syncthreads();
  return r;
}

// Inverse-CDF draw over w[0..n) (weights >= 0, at least one > 0): the first index with a positive
// weight whose inclusive prefix sum (double, fixed block-wide association) reaches u * total.
// All threads call.
// 1024 threads per attempt: the stage weights of a ~2000-point segment are two trips per thread, not eight
constexpr int kSelThreads = 1024;
// The association is FIXED at kDrawT segments whatever the block size (threads beyond them only keep the barriers).
constexpr int kDrawT = 256;
__device__ int draw_index(const float* __restrict__ w, int n, double u, double* s_part, int* s_pick) {
  constexpr int T = kDrawT;
  const int per = (n + T - 1) / T;
  const bool mine = threadIdx.x < (unsigned)T;
  const int lo = mine ? min((int)threadIdx.x * per, n) : n, hi = min(lo + per, n);
  double local = 0.0;
  for (int i = lo; i < hi; ++i) local += (double)w[i];
  if (mine) s_part[threadIdx.x] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    double run = 0.0;
#pragma unroll 8
    for (int t = 0; t < T; ++t) {   // exclusive offsets, in thread order
      const double v = s_part[t];
      s_part[t] = run;
      run += v;
    }
    s_part[T] = run;
    *s_pick = 0x7FFFFFFF;
  }
  __syncthreads();
  const double target = u * s_part[T];
  double run = mine ? s_part[threadIdx.x] : 0.0;
  int pick = 0x7FFFFFFF;
  for (int i = lo; i < hi; ++i) {
    const double wi = (double)w[i];
    run += wi;
    if (wi > 0.0 && run >= target) {
      pick = i;
      break;
    }
  }
  if (pick != 0x7FFFFFFF) atomicMin(s_pick, pick);   // every later segment qualifies too: the lowest wins
  __syncthreads();
  if (*s_pick == 0x7FFFFFFF && threadIdx.x == 0) {   // u * total rounded past the last partial sum
    int p = -1;
    for (int i = n - 1; i >= 0 && p < 0; --i)
      if (w[i] > 0.f) p = i;
    *s_pick = p;
  }
  __syncthreads();
  const int r = *s_pick;
  __syncthreads();
  return r;
}

// Closest approach of the segments P(s) = p1 + s u and Q(t) = q1 + t v, s, t in [0, 1]: the parameters
// are kept as fractions sN / sD and tN / tD of the 2 x 2 system's determinant and clamped edge by edge
// (the classic closed form the reference's distSegmentToSegment follows, base.cc:81-148).  Arithmetic
// as TryQuadrilateral instantiates it: float dot products widened to double, double fractions, the
// invariants narrowed to float before the residual vector is formed.  Returns |w + s u - t v|.
__device__ float seg_seg(V3 p1, V3 p2, V3 q1, V3 q2, double* invariant1, double* invariant2) {
  const double tiny = 0.0001;
  const V3 u = vsub(p2, p1), v = vsub(q2, q1), w = vsub(p1, q1);
  const double uu = dot(u, u), uv = dot(u, v), vv = dot(v, v), uw = dot(u, w), vw = dot(v, w);
  const double det = uu * vv - uv * uv;
  double sN, sD = det, tN, tD = det;
  if (det < tiny) {            // (almost) parallel: pin s = 0 and project q-side only
    sN = 0.0;
    sD = 1.0;
    tN = vw;
    tD = vv;
  } else {                      // interior solution of the unconstrained problem, then the s-edges
    sN = uv * vw - vv * uw;
    tN = uu * vw - uv * uw;
    if (sN < 0.0) {
      sN = 0.0;
      tN = vw;
      tD = vv;
    } else if (sN > sD) {
      sN = sD;
      tN = vw + uv;
      tD = vv;
    }
  }
  if (tN < 0.0) {               // t-edges: re-solve s on the edge t = 0 or t = 1
    tN = 0.0;
    const double m = -uw;
    if (m < 0.0) sN = 0.0;
    else if (m > uu) sN = sD;
    else {
      sN = m;
      sD = uu;
    }
  } else if (tN > tD) {
    tN = tD;
    const double m = -uw + uv;
    if (m < 0.0) sN = 0;
    else if (m > uu) sN = sD;
    else {
      sN = m;
      sD = uu;
    }
  }
  *invariant1 = fabs(sN) < tiny ? 0.0 : sN / sD;
  *invariant2 = fabs(tN) < tiny ? 0.0 : tN / tD;
  const float fs = (float)*invariant1, ft = (float)*invariant2;   // Eigen narrows the double scalars
  const V3 r = {sub(add(w.x, mul(fs, u.x)), mul(ft, v.x)), sub(add(w.y, mul(fs, u.y)), mul(ft, v.y)),
                sub(add(w.z, mul(fs, u.z)), mul(ft, v.z))};
  return norm(r);
}

// TryQuadrilateral on lanes 0..11 of the calling wave (all 64 lanes call): pairing p enumerates
// (i, j) in the reference's loop order; the first minimum wins (strict <).
__device__ void try_quadrilateral(const float4* __restrict__ P, int ids[4], float* inv1, float* inv2) {
  const int lane = threadIdx.x & 63;
  const int p = lane < 12 ? lane : 0;
  const int i = p / 3;
  int j = p % 3;
  j += j >= i ? 1 : 0;
  int k = 0;
  while (k == i || k == j) k++;
  int l = 0;
  while (l == i || l == j || l == k) l++;
  V3 q[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) q[t] = ld3(P, ids[t]);
  auto pick = [&](int t) { return t == 0 ? q[0] : (t == 1 ? q[1] : (t == 2 ? q[2] : q[3])); };
  double li1, li2;
  float sd = seg_seg(pick(i), pick(j), pick(k), pick(l), &li1, &li2);
  if (lane >= 12 || !(sd == sd)) sd = __int_as_float(0x7F800000);   // NaN never passes `<`
  // arg-min with the lowest pairing index on ties: key = (distance bits, pairing)
  unsigned long long key = ((unsigned long long)__float_as_uint(sd) << 8) | (unsigned)p;   // sd >= 0
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(key, off, 64);
    key = o < key ? o : key;
  }
  key = __shfl(key, 0, 64);   // lanes 0..15 hold the minimum of lanes 0..15
  const int bp = (int)(key & 0xFFu);
  const bool valid = (key >> 8) < 0x7F800000ull && (key >> 8) < (unsigned long long)__float_as_uint(3.4028234663852886e38f);
  const float f1 = __shfl((float)li1, bp, 64), f2 = __shfl((float)li2, bp, 64);
  if (valid) {   // min_distance starts at FLT_MAX: a pairing at exactly FLT_MAX or beyond never wins
    const int bi = bp / 3;
    int bj = bp % 3;
    bj += bj >= bi ? 1 : 0;
    int bk = 0;
    while (bk == bi || bk == bj) bk++;
    int bl = 0;
    while (bl == bi || bl == bj || bl == bk) bl++;
    const int t0 = ids[0], t1 = ids[1], t2 = ids[2], t3 = ids[3];
    auto sel = [&](int t) { return t == 0 ? t0 : (t == 1 ? t1 : (t == 2 ? t2 : t3)); };
    ids[0] = sel(bi);
    ids[1] = sel(bj);
    ids[2] = sel(bk);
    ids[3] = sel(bl);
    *inv1 = f1;
    *inv2 = f2;
  }
}

// kSelLdsPoints: an attempt's stage weights live in LDS up to this many points (the sequential sums are one wave's 64-wide
// trips over them, each waiting for its load: from memory that wait, not the chain of additions, was most of a stage)
constexpr int kSelLdsPoints = 12288;
// (8 waves per SIMD asked for: 106 scalar registers had capped the kernel at 7, one workgroup of 1024 threads per CU; with
//  78 two fit, so the base selections of two objects of a frame can share the compute units -- neutral for a single call)
template <bool LDS>
__global__ __launch_bounds__(kSelThreads, 8) void select_bases(SelectArgs a) {
  extern __shared__ float s_cur[];
  __shared__ double s_part[257];
  __shared__ int s_pick;
  __shared__ float s_sum;
  __shared__ int s_present;
  const int att = blockIdx.x;
  const double* u = a.u + 4 * (size_t)att;
  float* cur = LDS ? s_cur : a.cur + (size_t)att * a.n;
  const int n = a.n;
  auto fail = [&]() {
    if (threadIdx.x == 0) {
      a.status[att] = 0;
      a.ids[att] = make_int4(-1, -1, -1, -1);
      a.inv[att] = make_float2(0.f, 0.f);
      if (a.rows) a.rows[att] = make_int2(-1, -1);
    }
  };
  // ---- point 1: discrete_distribution over orig_probabilities_ (its double prefix sums are shared)
  if (threadIdx.x == 0) {
    const double total = a.prob_cdf[n - 1], target = u[0] * total;
    int lo = 0, hi = n - 1;   // first index with cdf >= target
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (a.prob_cdf[mid] >= target) hi = mid;
      else lo = mid + 1;
    }
    while (lo < n - 1 && a.Pnw[lo].w == 0.f) ++lo;   // a zero-weight point is never drawn
    s_pick = (total > 0.0 && a.Pnw[lo].w != 0.f) ? lo : -1;
  }
  __syncthreads();
  const int b1 = s_pick;
  __syncthreads();
  if (b1 < 0) {
    fail();
    return;
  }
  // ---- point 2
  if (threadIdx.x == 0) s_present = 0;
  __syncthreads();
  {
    const V3 pb = ld3(a.P, b1);
    const float4 nb4 = a.Pnw[b1];
    const V3 nb = {nb4.x, nb4.y, nb4.z};
    bool any = false;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float w = stage2_weight(a, i, a.Pnw[i].w, b1, pb, nb, nb4.w);
      cur[i] = w;
      any |= w != 0.f;
    }
    if (any) s_present = 1;
  }
  __syncthreads();
  if (!s_present) {
    fail();
    return;
  }
  float sum = sequential_sum(cur, n, &s_sum);
  for (int i = threadIdx.x; i < n; i += blockDim.x) cur[i] = __fdiv_rn(cur[i], sum);
  __syncthreads();
  const int b2 = draw_index(cur, n, u[1], s_part, &s_pick);
#if defined(PGP_SEL_STOP) && PGP_SEL_STOP == 2   // timing experiment (wrong results): the kernel up to the second point
  fail();
  return;
#endif
  // ---- point 3
  if (threadIdx.x == 0) s_present = 0;
  __syncthreads();
  {
    const V3 p1 = ld3(a.P, b1), pb = ld3(a.P, b2);
    const V3 v_1 = vsub(pb, p1);
    const float4 nb4 = a.Pnw[b2];
    const V3 nb = {nb4.x, nb4.y, nb4.z};
    bool any = false;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float w = stage3_weight(a, i, cur[i], b1, b2, p1, v_1, pb, nb, nb4.w);
      cur[i] = w;
      any |= w != 0.f;
    }
    if (any) s_present = 1;
  }
  __syncthreads();
  if (!s_present) {
    fail();
    return;
  }
  sum = sequential_sum(cur, n, &s_sum);
  for (int i = threadIdx.x; i < n; i += blockDim.x) cur[i] = __fdiv_rn(cur[i], sum);
  __syncthreads();
  const int b3 = draw_index(cur, n, u[2], s_part, &s_pick);
#if defined(PGP_SEL_STOP) && PGP_SEL_STOP == 3
  fail();
  return;
#endif
  // ---- point 4
  if (threadIdx.x == 0) s_present = 0;
  __syncthreads();
  {
    const V3 q1 = ld3(a.P, b1), q2 = ld3(a.P, b2), q3 = ld3(a.P, b3);
    const Plane pl = fit_plane(q1, q2, q3);
    const float4 nb4 = a.Pnw[b3];
    const V3 nb = {nb4.x, nb4.y, nb4.z};
    bool any = false;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float w = stage4_weight(a, i, cur[i], b1, b2, b3, q1, q2, q3, pl, nb, nb4.w);
      cur[i] = w;
      any |= w != 0.f;
    }
    if (any) s_present = 1;
  }
  __syncthreads();
  if (!s_present) {
    fail();
    return;
  }
  sum = sequential_sum(cur, n, &s_sum);
  for (int i = threadIdx.x; i < n; i += blockDim.x) cur[i] = __fdiv_rn(cur[i], sum);
  __syncthreads();
  const int b4 = draw_index(cur, n, u[3], s_part, &s_pick);
#if defined(PGP_SEL_STOP) && PGP_SEL_STOP == 4
  fail();
  return;
#endif
  // ---- pairing + invariants
  if (threadIdx.x < 64) {
    int ids[4] = {b1, b2, b3, b4};
    float i1 = 0.f, i2 = 0.f;
    try_quadrilateral(a.P, ids, &i1, &i2);
    // the rows of the pair-feature table that hold pairs1 / pairs6 of this base (computePPF of its two edges,
    // base.cc:1970-1981): they ride home with the base, so that pgp_find_congruent_batch_rows need not ask for them
    int row = -1;
    if (a.rows && threadIdx.x < 2) {
      const int i = threadIdx.x == 0 ? ids[0] : ids[2], j = threadIdx.x == 0 ? ids[1] : ids[3];
      const float4 n1 = a.Pnw[i], n2 = a.Pnw[j];
      row = table_find(a.tab, ppf_key(a.tab, ld3(a.P, i), {n1.x, n1.y, n1.z}, ld3(a.P, j), {n2.x, n2.y, n2.z}, nullptr));
    }
    const int row6 = __shfl(row, 1, 64);
    if (threadIdx.x == 0) {
      a.ids[att] = make_int4(ids[0], ids[1], ids[2], ids[3]);
      a.inv[att] = make_float2(i1, i2);
      a.status[att] = 1;
      if (a.rows) a.rows[att] = make_int2(row, row6);
    }
  }
}

// ---- inspection / batched helpers (tests, and the rest of the drop-in) ---------------------------
__global__ __launch_bounds__(256) void ppf_features(SelectArgs a, const int2* __restrict__ pairs, int m,
                                                    int4* __restrict__ f_out, int* __restrict__ row_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const int2 pr = pairs[t];
  int f[4] = {-1, -1, -1, -1};
  int row = -1;
  if ((unsigned)pr.x < (unsigned)a.n && (unsigned)pr.y < (unsigned)a.n) {
    const float4 n1 = a.Pnw[pr.x], n2 = a.Pnw[pr.y];
    const unsigned long long key = ppf_key(a.tab, ld3(a.P, pr.x), {n1.x, n1.y, n1.z}, ld3(a.P, pr.y), {n2.x, n2.y, n2.z}, f);
    row = table_find(a.tab, key);
  }
  f_out[t] = make_int4(f[0], f[1], f[2], f[3]);
  if (row_out) row_out[t] = row;
}

// one weighting loop for given base points: cur in/out (one block; normalised like the reference)
__global__ __launch_bounds__(256) void stage_weights(SelectArgs a, int stage, int b1, int b2, int b3, float* cur,
                                                     float* sum_out, int* present_out) {
  __shared__ float s_sum;
  __shared__ int s_present;
  const int n = a.n;
  if (threadIdx.x == 0) s_present = 0;
  __syncthreads();
  bool any = false;
  if (stage == 2) {
    const V3 pb = ld3(a.P, b1);
    const float4 nb4 = a.Pnw[b1];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float w = stage2_weight(a, i, cur[i], b1, pb, {nb4.x, nb4.y, nb4.z}, nb4.w);
      cur[i] = w;
      any |= w != 0.f;
    }
  } else if (stage == 3) {
    const V3 p1 = ld3(a.P, b1), pb = ld3(a.P, b2);
    const float4 nb4 = a.Pnw[b2];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float w = stage3_weight(a, i, cur[i], b1, b2, p1, vsub(pb, p1), pb, {nb4.x, nb4.y, nb4.z}, nb4.w);
      cur[i] = w;
      any |= w != 0.f;
    }
  } else {
    const V3 q1 = ld3(a.P, b1), q2 = ld3(a.P, b2), q3 = ld3(a.P, b3);
    const Plane pl = fit_plane(q1, q2, q3);
    const float4 nb4 = a.Pnw[b3];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float w = stage4_weight(a, i, cur[i], b1, b2, b3, q1, q2, q3, pl, {nb4.x, nb4.y, nb4.z}, nb4.w);
      cur[i] = w;
      any |= w != 0.f;
    }
  }
  if (any) s_present = 1;
  __syncthreads();
  const float sum = sequential_sum(cur, n, &s_sum);
  if (threadIdx.x == 0) {
    *sum_out = sum;
    *present_out = s_present;
  }
  if (s_present)
    for (int i = threadIdx.x; i < n; i += blockDim.x) cur[i] = __fdiv_rn(cur[i], sum);
}

__global__ __launch_bounds__(64) void base_invariants(const float4* __restrict__ P, int n, int4* ids, int m,
                                                      float2* __restrict__ inv, int* __restrict__ ok) {
  const int b = blockIdx.x;
  if (b >= m) return;
  const int4 v = ids[b];
  int t[4] = {v.x, v.y, v.z, v.w};
  const bool good = (unsigned)t[0] < (unsigned)n && (unsigned)t[1] < (unsigned)n && (unsigned)t[2] < (unsigned)n &&
                    (unsigned)t[3] < (unsigned)n;
  if (!good) {
    if (threadIdx.x == 0) {
      ok[b] = -1;
      inv[b] = make_float2(0.f, 0.f);
    }
    return;
  }
  float i1 = 0.f, i2 = 0.f;
  try_quadrilateral(P, t, &i1, &i2);
  if (threadIdx.x == 0) {
    ids[b] = make_int4(t[0], t[1], t[2], t[3]);
    inv[b] = make_float2(i1, i2);
    ok[b] = 1;
  }
}

// approximate_bin(int(atan2f(r, sign) * 180 / M_PI), 10) on the host's libm
int host_angle_bin(float r, float sign) {
  const float a = std::atan2(r, sign);
  return approximate_bin(int(a * 180 / M_PI), 10);
}

float next_up(float f) {
  uint32_t b;
  std::memcpy(&b, &f, 4);
  ++b;
  std::memcpy(&f, &b, 4);
  return f;
}

}  // namespace

// The nine ratios r = y / |x| at which the reference's 10-degree angle bin changes, per sign of x,
// found by bisection over the host's atan2f (monotone in r; checked around every threshold).
int ppf_thresholds(float tpos[9], float tneg[9]) {
  for (int sgn = 0; sgn < 2; ++sgn) {
    const float sx = sgn == 0 ? 1.0f : -1.0f;
    for (int b = 1; b <= 9; ++b) {
      // x > 0: bins rise 0 -> 90 with r; first r whose bin >= 10 b.  x < 0: bins fall 180 -> 90; first r
      // whose bin <= 180 - 10 b.
      auto reached = [&](float r) {
        const int v = host_angle_bin(r, sx);
        return sgn == 0 ? v >= 10 * b : v <= 180 - 10 * b;
      };
      uint32_t lo = 0u, hi = 0x7F800000u;   // +0 .. +inf as ordered bit patterns
      float fh;
      std::memcpy(&fh, &hi, 4);
      if (!reached(fh)) {
        set_error("ppf_thresholds: atan2f(inf, %g) does not reach bin %d", (double)sx, b);
        return PGP_EINVAL;
      }
      while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        float fm;
        std::memcpy(&fm, &mid, 4);
        if (reached(fm)) hi = mid;
        else lo = mid;
      }
      float t;
      std::memcpy(&t, &hi, 4);
      float zero = 0.f;
      if (reached(zero)) t = 0.f;
      // monotone around the threshold?
      float below = t, above = t;
      for (int k = 0; k < 2048; ++k) {
        if (below > 0.f) {
          uint32_t bb;
          std::memcpy(&bb, &below, 4);
          --bb;
          std::memcpy(&below, &bb, 4);
          if (reached(below)) {
            set_error("ppf_thresholds: host atan2f is not monotone near r = %g", (double)t);
            return PGP_EINVAL;
          }
        }
        above = next_up(above);
        if (above == above && !std::isinf(above) && !reached(above)) {
          set_error("ppf_thresholds: host atan2f is not monotone near r = %g", (double)t);
          return PGP_EINVAL;
        }
      }
      (sgn == 0 ? tpos : tneg)[b - 1] = t;
    }
  }
  return PGP_OK;
}

namespace {

int fill_select_args(pgp_ctx* ctx, SelectArgs* a) {
  if (ctx->nP <= 0 || !ctx->d_P.p || !ctx->has_scene_normals) {
    set_error("base selection needs a scene with normals: call pgp_set_scene first");
    return PGP_ESTATE;
  }
  if (!ctx->ppf_ready) {
    set_error("no pair-feature table: call pgp_set_ppf_map first");
    return PGP_ESTATE;
  }
  a->P = ctx->d_P.as<float4>();
  a->Pnw = ctx->d_Pnw.as<float4>();
  a->n = ctx->nP;
  a->tab.keys = ctx->d_ppf_keys.as<unsigned long long>();
  a->tab.value = ctx->d_ppf_val.as<uint32_t>();
  a->tab.mask = ctx->ppf_mask;
  a->tab.shift = ctx->ppf_shift;
  a->tab.trans_disc = 5;   // base.cc:303
  std::memcpy(a->tab.tpos, ctx->ppf_tpos, sizeof a->tab.tpos);
  std::memcpy(a->tab.tneg, ctx->ppf_tneg, sizeof a->tab.tneg);
  if (ctx->gate_deg_cached != 30.f) {
    gate_thresholds(30.f, &ctx->gate_lo, &ctx->gate_hi);
    ctx->gate_deg_cached = 30.f;
  }
  a->gate_lo = ctx->gate_lo;
  a->gate_hi = ctx->gate_hi;
  return PGP_OK;
}

}  // namespace

int set_ppf_map(pgp_ctx* ctx, const int* keys, const int* counts, const int* pairs, int n_keys) {
  ctx->csb_fit_m = 0;
  ctx->csb_nb = 0;   // a resident congruent batch indexes the OLD pair lists
  int rc = ppf_thresholds(ctx->ppf_tpos, ctx->ppf_tneg);
  if (rc != PGP_OK) return rc;
  uint32_t size = 16;
  int lg = 4;
  while (size < 2u * (uint32_t)std::max(n_keys, 1)) {
    size <<= 1;
    ++lg;
  }
  std::vector<unsigned long long> tk(size, ~0ull);
  std::vector<uint32_t> tv(size, 0u);
  std::vector<uint32_t> off((size_t)n_keys + 1, 0u);
  for (int k = 0; k < n_keys; ++k) {
    const int* f = keys + 4 * (size_t)k;
    off[k + 1] = off[k] + (uint32_t)(counts ? std::max(counts[k], 0) : 0);
    if (f[0] < 0 || f[1] < 0 || f[1] > 255 || f[2] < 0 || f[2] > 255 || f[3] < 0 || f[3] > 255) continue;   // unreachable key
    const unsigned long long key = ((unsigned long long)(unsigned)f[0] << 24) | ((unsigned long long)f[1] << 16) |
                                   ((unsigned long long)f[2] << 8) | (unsigned long long)f[3];
    uint32_t s = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - lg)) & (size - 1);
    while (tk[s] != ~0ull && tk[s] != key) s = (s + 1) & (size - 1);
    if (tk[s] == ~0ull) {   // std::map keeps the first insertion of a key
      tk[s] = key;
      tv[s] = (uint32_t)k;
    }
  }
  hipStream_t st = ctx->stream;
  if ((rc = ctx->d_ppf_keys.ensure((size_t)size * 8)) != PGP_OK) return rc;
  if ((rc = ctx->d_ppf_val.ensure((size_t)size * 4)) != PGP_OK) return rc;
  if ((rc = ctx->d_ppf_off.ensure(((size_t)n_keys + 1) * 4)) != PGP_OK) return rc;
  const size_t n_pairs = off[n_keys];
  if ((rc = ctx->d_ppf_pairs.ensure(std::max<size_t>(n_pairs, 1) * 8)) != PGP_OK) return rc;
  PGP_HIP(hipMemcpyAsync(ctx->d_ppf_keys.p, tk.data(), (size_t)size * 8, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(ctx->d_ppf_val.p, tv.data(), (size_t)size * 4, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(ctx->d_ppf_off.p, off.data(), ((size_t)n_keys + 1) * 4, hipMemcpyHostToDevice, st));
  if (n_pairs && pairs) PGP_HIP(hipMemcpyAsync(ctx->d_ppf_pairs.p, pairs, n_pairs * 8, hipMemcpyHostToDevice, st));
  PGP_HIP(hipStreamSynchronize(st));
  ctx->ppf_mask = size - 1;
  ctx->ppf_shift = 64 - lg;
  ctx->ppf_n_keys = n_keys;
  ctx->ppf_n_pairs = (long long)n_pairs;
  ctx->ppf_off_host.assign(off.begin(), off.end());
  ctx->ppf_ready = true;
  return PGP_OK;
}

// phase 0: the whole call; 1: queue only (the variates go up from a pinned image of the selection's own, the kernel is
// launched, nothing is waited for: pgp_select_bases_rows_begin); 2: bring the results of a phase-1 call home (.._end)
int launch_select_bases(pgp_ctx* ctx, const double* h_u, int n_attempts, int* h_ids, float* h_inv, int* h_status,
                        int* h_rows, hipStream_t st, int phase) {
  if (phase == 2) {
    if (ctx->sel_begun_A <= 0) {
      set_error("pgp_select_bases_rows_end: no selection has been begun");
      return PGP_ESTATE;
    }
    n_attempts = ctx->sel_begun_A;
    ctx->sel_begun_A = 0;
  } else {
    ctx->sel_begun_A = 0;   // (a begun selection that nobody collected is dropped)
  }
  SelectArgs a{};
  int rc = fill_select_args(ctx, &a);
  if (rc != PGP_OK) return rc;
  const int n = ctx->nP;
  // the first draw's distribution: double prefix sums of orig_probabilities_ (host: O(n), once per scene)
  if (!ctx->prob_cdf_valid) {
    std::vector<float4> hn((size_t)n);
    PGP_HIP(hipMemcpyAsync(hn.data(), ctx->d_Pnw.p, (size_t)n * 16, hipMemcpyDeviceToHost, st));
    PGP_HIP(hipStreamSynchronize(st));
    std::vector<double> cdf((size_t)n);
    double run = 0.0;
    for (int i = 0; i < n; ++i) {
      run += (double)hn[i].w;
      cdf[i] = run;
    }
    if ((rc = ctx->d_prob_cdf.ensure((size_t)n * 8)) != PGP_OK) return rc;
    PGP_HIP(hipMemcpyAsync(ctx->d_prob_cdf.p, cdf.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    PGP_HIP(hipStreamSynchronize(st));
    ctx->prob_cdf_valid = true;
  }
  const size_t A = (size_t)n_attempts;
  // workspace: u (double) | ids (int4) | inv (float2) | status (int) | rows (int2) | cur (float [A][n])
  const size_t bytes = A * 32 + A * 16 + A * 8 + A * 4 + 8 + A * 8 + A * (size_t)n * 4 + 512;
  if ((rc = ctx->d_sel_ws.ensure(bytes)) != PGP_OK) return rc;
  unsigned char* base = ctx->d_sel_ws.as<unsigned char>();
  double* d_u = reinterpret_cast<double*>(base);
  int4* d_ids = reinterpret_cast<int4*>(base + A * 32);
  float2* d_inv = reinterpret_cast<float2*>(base + A * 48);
  int* d_status = reinterpret_cast<int*>(base + A * 56);
  // (8-byte records: behind ids | inv | status = 28 B per attempt, which ends on an odd word when the attempts are odd in number)
  const size_t rows_off = (A * 28 + 7) & ~(size_t)7;
  int2* d_rows = reinterpret_cast<int2*>(base + A * 32 + rows_off);
  float* d_cur = reinterpret_cast<float*>(base + ((A * 32 + rows_off + A * 8 + 255) & ~(size_t)255));
  // the variates go up from the pinned area the results come home to (the call synchronises before it returns): a
  // copy out of the caller's pageable array costs the host ~10 us whatever its size
  HostOut out(ctx, st);
  if (phase == 2) goto collect;
  {
    unsigned char* up = nullptr;
    if (phase == 1) {
      // (an image that outlives this call: the stage kernel may run after it has returned)
      if (ctx->h_sel_cap < A * 32) {
        if (ctx->h_sel_pin) (void)hipHostFree(ctx->h_sel_pin);
        ctx->h_sel_pin = nullptr;
        ctx->h_sel_cap = 0;
        PGP_HIP(hipHostMalloc(&ctx->h_sel_pin, A * 32 + 4096, hipHostMallocDefault));
        ctx->h_sel_cap = A * 32 + 4096;
      }
      up = static_cast<unsigned char*>(ctx->h_sel_pin);
    } else {
      up = out.room(A * 32);
    }
    if (up) std::memcpy(up, h_u, A * 32);
    if (up) {
      if ((rc = stage_to_device(st, d_u, up, A * 32)) != PGP_OK) return rc;
    } else {
      PGP_HIP(hipMemcpyAsync(d_u, h_u, A * 32, hipMemcpyHostToDevice, st));
    }
  }
  a.prob_cdf = ctx->d_prob_cdf.as<double>();
  a.u = d_u;
  a.cur = d_cur;
  a.ids = d_ids;
  a.inv = d_inv;
  a.status = d_status;
  a.rows = (h_rows || phase == 1) ? d_rows : nullptr;   // (a begun selection always writes them: its collector may ask)
  if (getenv("PGP_SEL_GLOBAL") || a.n > kSelLdsPoints)   // (A/B knob; segments beyond the LDS form)
    hipLaunchKernelGGL(select_bases<false>, dim3(n_attempts), dim3(kSelThreads), 0, st, a);
  else
    hipLaunchKernelGGL(select_bases<true>, dim3(n_attempts), dim3(kSelThreads), (size_t)a.n * 4, st, a);
  PGP_HIP(hipGetLastError());
  if (phase == 1) {
    ctx->sel_begun_A = n_attempts;
    return PGP_OK;
  }
collect:
  // ids | inv | status | rows lie back to back in the workspace: ONE copy back, into pinned memory (pgp::HostOut)
  const unsigned char* got = nullptr;
  if ((rc = out.fetch(&got, d_ids, h_rows ? rows_off + A * 8 : A * 28)) != PGP_OK || (rc = out.sync()) != PGP_OK) return rc;
  std::memcpy(h_ids, got, A * 16);
  std::memcpy(h_inv, got + A * 16, A * 8);
  std::memcpy(h_status, got + A * 24, A * 4);
  if (h_rows) std::memcpy(h_rows, got + rows_off, A * 8);
  return PGP_OK;
}

int launch_ppf_features(pgp_ctx* ctx, const int* h_pairs, int m, int* h_f, int* h_row, hipStream_t st) {
  SelectArgs a{};
  int rc = fill_select_args(ctx, &a);
  if (rc != PGP_OK) return rc;
  const size_t M = (size_t)m;
  if ((rc = ctx->d_sel_ws.ensure(M * 8 + M * 16 + M * 4 + 64)) != PGP_OK) return rc;
  unsigned char* base = ctx->d_sel_ws.as<unsigned char>();
  int4* d_f = reinterpret_cast<int4*>(base);
  int2* d_pairs = reinterpret_cast<int2*>(base + M * 16);
  int* d_row = reinterpret_cast<int*>(base + M * 24);
  PGP_HIP(hipMemcpyAsync(d_pairs, h_pairs, M * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(ppf_features, dim3((m + 255) / 256), dim3(256), 0, st, a, (const int2*)d_pairs, m, d_f, d_row);
  PGP_HIP(hipGetLastError());
  HostOut out(ctx, st);
  if (h_f && (rc = out.to(h_f, d_f, M * 16)) != PGP_OK) return rc;   // (null: the caller wants the rows only)
  if (h_row && (rc = out.to(h_row, d_row, M * 4)) != PGP_OK) return rc;
  return out.sync();
}

int launch_stage_weights(pgp_ctx* ctx, int stage, int b1, int b2, int b3, float* h_cur, float* h_sum, int* h_present,
                         hipStream_t st) {
  SelectArgs a{};
  int rc = fill_select_args(ctx, &a);
  if (rc != PGP_OK) return rc;
  const int n = ctx->nP;
  auto bad = [&](int b) { return b < 0 || b >= n; };
  if (stage < 2 || stage > 4 || bad(b1) || (stage >= 3 && bad(b2)) || (stage >= 4 && bad(b3))) {
    set_error("pgp_stocs_stage_weights: bad stage / base ids");
    return PGP_EINVAL;
  }
  if ((rc = ctx->d_sel_ws.ensure((size_t)n * 4 + 64)) != PGP_OK) return rc;
  float* d_cur = ctx->d_sel_ws.as<float>();
  float* d_sum = d_cur + n;
  int* d_present = reinterpret_cast<int*>(d_sum + 1);
  PGP_HIP(hipMemcpyAsync(d_cur, h_cur, (size_t)n * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(stage_weights, dim3(1), dim3(256), 0, st, a, stage, b1, b2, b3, d_cur, d_sum, d_present);
  PGP_HIP(hipGetLastError());
  HostOut out(ctx, st);
  if ((rc = out.to(h_cur, d_cur, (size_t)n * 4)) != PGP_OK || (rc = out.to(h_sum, d_sum, 4)) != PGP_OK ||
      (rc = out.to(h_present, d_present, 4)) != PGP_OK)
    return rc;
  return out.sync();
}

int launch_base_invariants(pgp_ctx* ctx, int* h_ids, int m, float* h_inv, int* h_ok, hipStream_t st) {
  if (ctx->nP <= 0 || !ctx->d_P.p) {
    set_error("pgp_base_invariants: no scene");
    return PGP_ESTATE;
  }
  const size_t M = (size_t)m;
  int rc = ctx->d_sel_ws.ensure(M * 16 + M * 8 + M * 4 + 64);
  if (rc != PGP_OK) return rc;
  unsigned char* base = ctx->d_sel_ws.as<unsigned char>();
  int4* d_ids = reinterpret_cast<int4*>(base);
  float2* d_inv = reinterpret_cast<float2*>(base + M * 16);
  int* d_ok = reinterpret_cast<int*>(base + M * 24);
  PGP_HIP(hipMemcpyAsync(d_ids, h_ids, M * 16, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(base_invariants, dim3(m), dim3(64), 0, st, ctx->d_P.as<float4>(), ctx->nP, d_ids, m, d_inv, d_ok);
  PGP_HIP(hipGetLastError());
  HostOut out(ctx, st);
  if ((rc = out.to(h_ids, d_ids, M * 16)) != PGP_OK || (rc = out.to(h_inv, d_inv, M * 8)) != PGP_OK ||
      (rc = out.to(h_ok, d_ok, M * 4)) != PGP_OK)
    return rc;
  return out.sync();
}

}  // namespace pgp
