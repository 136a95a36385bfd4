// csrc/congruent.hip -- congruent-set extraction on gfx950.
//
// Replaces, for one base at a time (the host keeps the base sampling RNG):
//   MatchSuper4PCS::ExtractPairs               S4/algorithms/super4pcs.cc:193-236
//     + PairCreationFunctor::process           S4/pairCreationFunctor.h:167-253
//     + IntersectionFunctor (octree raster)    S4/accelerators/pairExtraction/intersectionFunctor.h:105-234
//   MatchSuper4PCS::FindCongruentQuadrilaterals  S4/algorithms/super4pcs.cc:78-187
//     + IndexedNormalSet<Point,3,7,float>      S4/accelerators/normalset.{h,hpp}
//
// ExtractPairs: the reference rasterises a sphere of radius d around every point through an
// octree to find all i > j with | |q_i - q_j| - d | <= eps.  On the GPU the search model (<= a few
// thousand points) is scanned exhaustively: one wave per row i, lanes stride j < i, ballot +
// popcount give each hit its output slot, so the list comes out in (i, j) lexicographic order,
// (j,i) then (i,j) per hit -- the same SET as the reference's functor (tests/golden/congruent_*).
//
// FindCongruentQuadrilaterals: the reference hashes the invariant point e1 = p1 + inv1 (p2-p1) of
// every P-pair into an eps-grid cell x 7^3 direction bin, then for every Q-pair looks at the ONE
// cell of its own invariant point e2, "renders" a cone of half-angle alpha around the pair
// direction into the 343 direction bins, and accepts the P-pairs of the coloured bins whose
// world-space invariant points are within sqrt(delta) (sic: squared distance vs delta,
// super4pcs.cc:170).  Here: P-pairs are bucketed by a hash of their cell (count / scan / fill),
// one lane per Q-pair recomputes the cone mask (11 x 32 bits in VGPRs) and walks its bucket;
// hits are (P-pair id, Q-pair id) keys, sorted with a device radix sort = the reference's
// std::set order, so `congruent_quads[k]` means the same quad in both implementations.
//
// Float parity: all index arithmetic (unit-cube coordinates, cell and bin indices, quaternion
// rotation of the cone samples, normalisation) is evaluated in the reference's order with
// separately rounded operations and correctly rounded sqrt/divide.  The transcendental part of
// the cone (acos, atan, sin, cos of per-BASE constants) is evaluated once per call on the host
// with the host libm -- exactly what the reference does -- and passed in as a <= 56-entry table.

#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it

#include "pgp_internal.h"

#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <cmath>
#include <chrono>
#include <vector>

namespace pgp {

namespace {

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }
// correctly rounded float sqrt (see rigid_fit.hip: __fsqrt_rn is 1 ulp off on ~15 % of inputs)
__device__ __forceinline__ float sqrt_rn(float z) { return (float)__dsqrt_rn((double)z); }

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ float sqnorm(V3 v) { return add(mul(v.x, v.x), add(mul(v.y, v.y), mul(v.z, v.z))); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {sub(a.x, b.x), sub(a.y, b.y), sub(a.z, b.z)}; }
__device__ __forceinline__ V3 normalized(V3 v) {  // Eigen normalized(): z > 0 ? v / sqrt(z) : v
  float z = sqnorm(v);
  if (z > 0.f) {
    float n = sqrt_rn(z);
    return {fdiv(v.x, n), fdiv(v.y, n), fdiv(v.z, n)};
  }
  return v;
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {sub(mul(a.y, b.z), mul(a.z, b.y)), sub(mul(a.z, b.x), mul(a.x, b.z)), sub(mul(a.x, b.y), mul(a.y, b.x))};
}
__device__ __forceinline__ V3 ld3(const float4* __restrict__ a, int i) {
  float4 v = a[i];
  return {v.x, v.y, v.z};
}
// p1 + inv * (p2 - p1)
__device__ __forceinline__ V3 lerp_pt(V3 p1, V3 p2, float inv) {
  V3 d = vsub(p2, p1);
  return {add(p1.x, mul(inv, d.x)), add(p1.y, mul(inv, d.y)), add(p1.z, mul(inv, d.z))};
}

// IndexedNormalSet::indexNormal (normalset.h:100-104, utils.h:141-148): 7 bins per axis
__device__ __forceinline__ int normal_bin(V3 n, float nepsilon) {
  int ix = (int)fdiv(add(fdiv(n.x, 2.0f), 0.5f), nepsilon);
  int iy = (int)fdiv(add(fdiv(n.y, 2.0f), 0.5f), nepsilon);
  int iz = (int)fdiv(add(fdiv(n.z, 2.0f), 0.5f), nepsilon);
  if ((unsigned)ix > 6u || (unsigned)iy > 6u || (unsigned)iz > 6u) return -1;
  return iz * 49 + iy * 7 + ix;
}

// IndexedNormalSet::indexPos: int(p / epsilon) per axis in an eg^3 grid; -1 outside
__device__ __forceinline__ long long pos_cell(V3 p, float epsilon, int eg) {
  float cx = fdiv(p.x, epsilon), cy = fdiv(p.y, epsilon), cz = fdiv(p.z, epsilon);
  if (!(cx > -1.0f && cx < (float)eg && cy > -1.0f && cy < (float)eg && cz > -1.0f && cz < (float)eg)) return -1;
  int ix = (int)cx, iy = (int)cy, iz = (int)cz;  // truncation toward zero, as int(coord)
  if (ix < 0 || iy < 0 || iz < 0) return -1;
  return ((long long)iz * eg + iy) * eg + ix;
}

__device__ __forceinline__ unsigned bucket_of(long long cell, unsigned mask) {
  unsigned long long h = (unsigned long long)cell * 0x9E3779B97F4A7C15ull;
  return (unsigned)(h >> 40) & mask;
}

// ---------------- ExtractPairs ----------------
template <bool FILL>
__global__ __launch_bounds__(256) void pair_rows(const float4* __restrict__ Qw, int n, double pair_distance,
                                                 double eps, uint32_t* __restrict__ row_cnt,
                                                 const uint32_t* __restrict__ row_start,
                                                 int2* __restrict__ out, uint32_t cap) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const V3 qi = ld3(Qw, i);
  uint32_t running = 0;
  const uint32_t base = FILL ? row_start[i] : 0u;
  for (int j0 = 0; j0 < i; j0 += 64) {
    const int j = j0 + lane;
    bool hit = false;
    if (j < i) {
      V3 d = vsub(qi, ld3(Qw, j));
      float dist = sqrt_rn(sqnorm(d));  // (q.pos() - p.pos()).norm()
      hit = !(fabs((double)dist - pair_distance) > eps);
    }
    unsigned long long m = __ballot(hit);
    if (FILL && hit) {
      uint32_t slot = base + running + __popcll(m & ((1ull << lane) - 1ull));
      if (2ull * slot < cap) out[2 * slot] = make_int2(j, i);          // pairs->emplace_back(j, i)
      if (2ull * slot + 1 < cap) out[2 * slot + 1] = make_int2(i, j);  // pairs->emplace_back(i, j)
    }
    running += __popcll(m);
  }
  if (!FILL && lane == 0) row_cnt[i] = running;
}

// ---------------- FindCongruentQuadrilaterals ----------------
struct ConeTable {
  int nb;          // nbSample
  float v[56][3];  // (sinAlpha cos(theta_a), sinAlpha sin(theta_a), cosAlpha)
};

int cone_for_base(const float base[12], float* cone /*[56][3]*/);   // defined with the batch launcher below

struct CsArgs {
  const float4* Qw;  // world (centred) search model
  const float4* Qu;  // unit-cube image
  int nQs;
  const int2* Pp;
  int nP;
  const int2* Qp;
  int nQ;
  float inv1, inv2, threshold, epsilon, nepsilon;
  int eg;
  unsigned bmask;
  uint32_t* bucket_cnt;          // [nb+1]
  const uint32_t* bucket_start;  // [nb+1]
  int4* entries;                 // {cell_lo, cell_hi, bin, id}
  uint32_t* q_cnt;               // [nQ+1]
  const uint32_t* q_start;
  unsigned long long* keys;
  uint32_t cap;
};

template <bool FILL>
__global__ __launch_bounds__(256) void p_entries(CsArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nP) return;
  int2 pr = a.Pp[i];
  if ((unsigned)pr.x >= (unsigned)a.nQs || (unsigned)pr.y >= (unsigned)a.nQs) return;
  V3 p1 = ld3(a.Qu, pr.x), p2 = ld3(a.Qu, pr.y);
  V3 n = normalized(vsub(p2, p1));
  long long c = pos_cell(lerp_pt(p1, p2, a.inv1), a.epsilon, a.eg);
  int b = normal_bin(n, a.nepsilon);
  if (c < 0 || b < 0) return;  // addElement returns false
  unsigned bk = bucket_of(c, a.bmask);
  uint32_t slot = atomicAdd(&a.bucket_cnt[bk], 1u);
  if (FILL) a.entries[a.bucket_start[bk] + slot] = make_int4((int)(c & 0xFFFFFFFFll), (int)(c >> 32), b, i);
}

template <bool FILL>
__global__ __launch_bounds__(128) void q_match(CsArgs a, ConeTable cone) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nQ) return;
  int2 qr = a.Qp[i];
  uint32_t found = 0;
  const uint32_t out0 = FILL ? a.q_start[i] : 0u;
  if ((unsigned)qr.x < (unsigned)a.nQs && (unsigned)qr.y < (unsigned)a.nQs) {
    V3 p1 = ld3(a.Qu, qr.x), p2 = ld3(a.Qu, qr.y);
    long long c = pos_cell(lerp_pt(p1, p2, a.inv2), a.epsilon, a.eg);
    if (c >= 0) {
      unsigned bk = bucket_of(c, a.bmask);
      uint32_t s = a.bucket_start[bk], e = a.bucket_start[bk + 1];
      // is there anything of this cell in the bucket?  (angularGrid(p) == NULL otherwise)
      bool any = false;
      for (uint32_t k = s; k < e && !any; ++k) {
        int4 en = a.entries[k];
        any = en.x == (int)(c & 0xFFFFFFFFll) && en.y == (int)(c >> 32);
      }
      if (any) {
        // q.setFromTwoVectors((0,0,1), queryn)   (Eigen Quaternion.h:577-610)
        V3 queryn = normalized(vsub(p2, p1));
        V3 v1 = normalized(queryn);
        float cq = add(mul(v1.x, 0.f), add(mul(v1.y, 0.f), mul(v1.z, 1.f)));
        V3 qv;
        float qw;
        if (cq < add(-1.0f, 1e-5f)) {
          // nearly opposite to +z: the reference takes the axis from a 2x3 SVD; any unit axis
          // orthogonal to z is a valid half-turn, we take x (documented divergence)
          float cc = cq > -1.0f ? cq : -1.0f;
          float w2 = mul(add(1.0f, cc), 0.5f);
          qw = sqrt_rn(w2);
          qv = {sqrt_rn(sub(1.0f, w2)), 0.f, 0.f};
        } else {
          V3 axis = {sub(mul(0.f, v1.z), mul(1.f, v1.y)), sub(mul(1.f, v1.x), mul(0.f, v1.z)),
                     sub(mul(0.f, v1.y), mul(0.f, v1.x))};
          float sq = sqrt_rn(mul(add(1.0f, cq), 2.0f));
          float invs = fdiv(1.0f, sq);
          qv = {mul(axis.x, invs), mul(axis.y, invs), mul(axis.z, invs)};
          qw = mul(sq, 0.5f);
        }
        // cone "rendering" into the 343 direction bins (normalset.hpp:186-196)
        uint32_t colored[11];
#pragma unroll
        for (int w = 0; w < 11; ++w) colored[w] = 0u;
        for (int s2 = 0; s2 < cone.nb; ++s2) {
          V3 v = {cone.v[s2][0], cone.v[s2][1], cone.v[s2][2]};
          V3 uv = cross(qv, v);  // q * v = v + w*uv + vec x uv, uv = 2 (vec x v)
          uv = {add(uv.x, uv.x), add(uv.y, uv.y), add(uv.z, uv.z)};
          V3 c2 = cross(qv, uv);
          V3 r = {add(add(v.x, mul(qw, uv.x)), c2.x), add(add(v.y, mul(qw, uv.y)), c2.y),
                  add(add(v.z, mul(qw, uv.z)), c2.z)};
          int id = normal_bin(normalized(r), a.nepsilon);
          if (id >= 0) {
#pragma unroll
            for (int w = 0; w < 11; ++w)
              if (w == (id >> 5)) colored[w] |= 1u << (id & 31);
          }
        }
        // world-space invariant point of the query pair
        V3 queryQ = lerp_pt(ld3(a.Qw, qr.x), ld3(a.Qw, qr.y), a.inv2);
        for (uint32_t k = s; k < e; ++k) {
          int4 en = a.entries[k];
          if (en.x != (int)(c & 0xFFFFFFFFll) || en.y != (int)(c >> 32)) continue;
          uint32_t word = 0;
#pragma unroll
          for (int w = 0; w < 11; ++w)
            if (w == (en.z >> 5)) word = colored[w];
          if (!((word >> (en.z & 31)) & 1u)) continue;
          int2 pp = a.Pp[en.w];
          V3 w1 = ld3(a.Qw, pp.x), w2 = ld3(a.Qw, pp.y);
          V3 dd = vsub(w2, w1);  // invPoint = pp1 + (pp2 - pp1) * invariant1
          V3 ip = {add(w1.x, mul(dd.x, a.inv1)), add(w1.y, mul(dd.y, a.inv1)), add(w1.z, mul(dd.z, a.inv1))};
          if (sqnorm(vsub(queryQ, ip)) <= a.threshold) {  // squared distance vs delta, sic
            if (FILL && out0 + found < a.cap)
              a.keys[out0 + found] = ((unsigned long long)(unsigned)en.w << 32) | (unsigned)i;
            ++found;
          }
        }
      }
    }
  }
  if (!FILL) a.q_cnt[i] = found;
}

__global__ __launch_bounds__(256) void emit_quads(const unsigned long long* __restrict__ keys, uint32_t n,
                                                  const int2* __restrict__ Pp, const int2* __restrict__ Qp,
                                                  int4* __restrict__ quads) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  unsigned long long key = keys[k];
  int2 p = Pp[(uint32_t)(key >> 32)], q = Qp[(uint32_t)(key & 0xFFFFFFFFull)];
  quads[k] = make_int4(p.x, p.y, q.x, q.y);
}


// ---------------- FindCongruentQuadrilaterals for MANY bases in one pass ----------------
// The drop-in extracts congruent sets for ~100 bases per object (base.cc:1855-1874).  Driven base by
// base that is ~12 small launches, two host syncs and two list uploads each (24 ms per object,
// round 1).  Here every base's pair lists are ranges of the device-resident pair-feature table
// (pgp_set_ppf_map: pairs1 = PPFMap[ppf(b0,b1)], pairs6 = PPFMap[ppf(b2,b3)], base.cc:1970-1981),
// one thread handles one (base, pair), the P-pairs of all bases share ONE bucket table (the base id
// is part of the hash and of the entry), and the match keys (base | P-pair | Q-pair) are sorted
// once -- per base that is the reference's std::set order again.  Same arithmetic per pair as the
// single-base kernels above; identical lists (tests/test_congruent_batch_gpu.py).
struct BatchBase {
  float inv1, inv2;
  uint32_t p_off, p_cnt, q_off, q_cnt;   // ranges of the table's pair array
  uint32_t p_flat, q_flat;               // first flat thread index of this base's P / Q pairs
  int cone_nb;
  int pad;
};

struct CsBatchArgs {
  const float4* Qw;
  const float4* Qu;
  int nQs;
  const int2* pairs;          // the table's pair array
  const BatchBase* bases;
  int nb;
  const float* cones;         // [nb][56][3]
  float threshold, epsilon, nepsilon;
  int eg;
  unsigned bmask;
  // the P entries of a bucket are a LIST (head[bucket] -> next[entry] -> ... -> -1, entry = flat P index): filled in ONE
  // pass with an atomic exchange on the head -- a count pass, a scan of the buckets and a fill pass before (seven dependent
  // small launches and fills, ~45 us of a 0.2 ms phase); the matches are sorted afterwards, so the order in which a walk
  // meets the entries of a bucket does not matter
  int* head;                  // [bmask + 1]
  int* next;                  // [total_p]
  int4* entries;              // [total_p] {cell_lo, cell_hi, bin | base << 16, P-pair index}
  unsigned long long* keys;   // base << 48 | P-pair << 24 | Q-pair
  uint32_t total_p, total_q;
  uint32_t* n_keys;           // bq_match: matches appended so far (may pass key_cap: the host then grows and repeats)
  uint32_t* base_cnt;         // [nb] matches per base
  uint32_t key_cap;
};

__device__ __forceinline__ int base_of(const BatchBase* __restrict__ bases, int nb, uint32_t t, bool q_side) {
  int lo = 0, hi = nb - 1;   // last base whose first flat index is <= t
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    const uint32_t f = q_side ? bases[mid].q_flat : bases[mid].p_flat;
    if (f <= t) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ unsigned bucket_of_b(long long cell, int b, unsigned mask) {
  unsigned long long h = ((unsigned long long)cell * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)(b + 1) * 0xC2B2AE3D27D4EB4Full);
  h *= 0xD6E8FEB86659FD93ull;
  return (unsigned)(h >> 40) & mask;
}

// heads = -1, match counters = 0: one launch instead of two fills split by the runtime into four
__global__ __launch_bounds__(256) void bp_init(int* __restrict__ head, uint32_t n_head, uint32_t* __restrict__ counters, uint32_t n_counters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_head) head[t] = -1;
  if (t < n_counters) counters[t] = 0u;
}

__global__ __launch_bounds__(256) void bp_entries(CsBatchArgs a) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.total_p) return;
  const int b = base_of(a.bases, a.nb, t, false);
  const BatchBase B = a.bases[b];
  const uint32_t i = t - B.p_flat;
  const int2 pr = a.pairs[B.p_off + i];
  if ((unsigned)pr.x >= (unsigned)a.nQs || (unsigned)pr.y >= (unsigned)a.nQs) return;
  V3 p1 = ld3(a.Qu, pr.x), p2 = ld3(a.Qu, pr.y);
  V3 n = normalized(vsub(p2, p1));
  long long c = pos_cell(lerp_pt(p1, p2, B.inv1), a.epsilon, a.eg);
  int bin = normal_bin(n, a.nepsilon);
  if (c < 0 || bin < 0) return;  // addElement returns false
  unsigned bk = bucket_of_b(c, b, a.bmask);
  a.entries[t] = make_int4((int)(c & 0xFFFFFFFFll), (int)(c >> 32), bin | (b << 16), (int)i);
  a.next[t] = atomicExch(&a.head[bk], (int)t);   // (the entry is read by the NEXT launch only)
}

// ONE pass: a thread counts its matches, reserves that many key slots (one atomic per thread that found any, one more on
// its base's counter) and walks its bucket a second time to write them -- the keys are sorted afterwards, so where a
// thread's keys land does not matter, and the expensive part of a thread (the cone of <= 56 rotated samples) is done once.
// (Before: a counting launch, a scan over all Q pairs and a second full launch; profiles/r04_dropin_kernels.txt.)
// does Q pair t fall into a cell that holds a P entry of its base?  (the cheap part of a match: two loads and a bucket walk)
__device__ __forceinline__ bool bq_has_cell(const CsBatchArgs& a, const uint32_t t, const int b, int* first) {
  const BatchBase B = a.bases[b];
  const int2 qr = a.pairs[B.q_off + (t - B.q_flat)];
  if ((unsigned)qr.x >= (unsigned)a.nQs || (unsigned)qr.y >= (unsigned)a.nQs) return false;
  const V3 p1 = ld3(a.Qu, qr.x), p2 = ld3(a.Qu, qr.y);
  const long long c = pos_cell(lerp_pt(p1, p2, B.inv2), a.epsilon, a.eg);
  if (c < 0) return false;
  const unsigned bk = bucket_of_b(c, b, a.bmask);
  const int clo = (int)(c & 0xFFFFFFFFll), chi = (int)(c >> 32);
  for (int k = a.head[bk]; k >= 0; k = a.next[k]) {
    const int4 en = a.entries[k];
    if (en.x == clo && en.y == chi && (en.z >> 16) == b) {
      *first = k;   // the walks of the match start at the first entry of the cell
      return true;
    }
  }
  return false;
}

// The cone of a Q pair (<= 56 samples rotated onto its direction, each binned: the expensive part of a match, ~300
// instructions a sample): lane `part` of G takes the samples part, part + G, ...; the caller ORs the lanes' bit sets.
__device__ __forceinline__ void bq_cone_mask(const CsBatchArgs& a, const uint32_t t, const int b, const int part, const int G,
                                             uint32_t colored[11]) {
  const BatchBase B = a.bases[b];
  const int2 qr = a.pairs[B.q_off + (t - B.q_flat)];
  const float* cone = a.cones + (size_t)b * 168;
  const V3 p1 = ld3(a.Qu, qr.x), p2 = ld3(a.Qu, qr.y);
  // q.setFromTwoVectors((0,0,1), queryn)   (Eigen Quaternion.h:577-610)
  V3 queryn = normalized(vsub(p2, p1));
  V3 v1 = normalized(queryn);
  float cq = add(mul(v1.x, 0.f), add(mul(v1.y, 0.f), mul(v1.z, 1.f)));
  V3 qv;
  float qw;
  if (cq < add(-1.0f, 1e-5f)) {
    float cc = cq > -1.0f ? cq : -1.0f;
    float w2 = mul(add(1.0f, cc), 0.5f);
    qw = sqrt_rn(w2);
    qv = {sqrt_rn(sub(1.0f, w2)), 0.f, 0.f};
  } else {
    V3 axis = {sub(mul(0.f, v1.z), mul(1.f, v1.y)), sub(mul(1.f, v1.x), mul(0.f, v1.z)),
               sub(mul(0.f, v1.y), mul(0.f, v1.x))};
    float sq = sqrt_rn(mul(add(1.0f, cq), 2.0f));
    float invs = fdiv(1.0f, sq);
    qv = {mul(axis.x, invs), mul(axis.y, invs), mul(axis.z, invs)};
    qw = mul(sq, 0.5f);
  }
  for (int s2 = part; s2 < B.cone_nb; s2 += G) {
    V3 v = {cone[3 * s2], cone[3 * s2 + 1], cone[3 * s2 + 2]};
    V3 uv = cross(qv, v);
    uv = {add(uv.x, uv.x), add(uv.y, uv.y), add(uv.z, uv.z)};
    V3 c2 = cross(qv, uv);
    V3 r = {add(add(v.x, mul(qw, uv.x)), c2.x), add(add(v.y, mul(qw, uv.y)), c2.y),
            add(add(v.z, mul(qw, uv.z)), c2.z)};
    int id = normal_bin(normalized(r), a.nepsilon);
    if (id >= 0) {
#pragma unroll
      for (int w = 0; w < 11; ++w)
        if (w == (id >> 5)) colored[w] |= 1u << (id & 31);
    }
  }
}

// The walk of a Q pair over the P entries of its cell, with the cone's bit set (LDS).  WRITE = false: counts the matches
// (returned) and keeps the first kBqKeep of them in kept[]; WRITE = true: writes the keys of all matches from keys[out0] on
// (a thread with more than kBqKeep matches walks a second time).
constexpr int kBqKeep = 4;
template <bool WRITE>
__device__ __forceinline__ uint32_t bq_walk(const CsBatchArgs& a, const uint32_t t, const int b, const int s,
                                            const uint32_t* __restrict__ colored, uint32_t kept[kBqKeep], uint32_t out0) {
  const BatchBase B = a.bases[b];
  const uint32_t i = t - B.q_flat;
  const int2 qr = a.pairs[B.q_off + i];
  const V3 p1 = ld3(a.Qu, qr.x), p2 = ld3(a.Qu, qr.y);
  const long long c = pos_cell(lerp_pt(p1, p2, B.inv2), a.epsilon, a.eg);
  const int clo = (int)(c & 0xFFFFFFFFll), chi = (int)(c >> 32);
  const V3 queryQ = lerp_pt(ld3(a.Qw, qr.x), ld3(a.Qw, qr.y), B.inv2);
  uint32_t n = 0;
  for (int k = s, nk; k >= 0; k = nk) {
    nk = a.next[k];
    const int4 en = a.entries[k];
    if (en.x != clo || en.y != chi || (en.z >> 16) != b) continue;
    const int bin = en.z & 0xFFFF;
    const uint32_t word = colored[bin >> 5];
    if (!((word >> (bin & 31)) & 1u)) continue;
    const int2 pp = a.pairs[B.p_off + (uint32_t)en.w];
    const V3 w1 = ld3(a.Qw, pp.x), w2 = ld3(a.Qw, pp.y);
    const V3 dd = vsub(w2, w1);
    const V3 ip = {add(w1.x, mul(dd.x, B.inv1)), add(w1.y, mul(dd.y, B.inv1)), add(w1.z, mul(dd.z, B.inv1))};
    if (sqnorm(vsub(queryQ, ip)) <= a.threshold) {
      if (!WRITE) {
#pragma unroll
        for (int q = 0; q < kBqKeep; ++q)
          if ((uint32_t)q == n) kept[q] = (uint32_t)en.w;
      } else if (out0 + n < a.key_cap) {
        a.keys[out0 + n] = ((unsigned long long)(unsigned)b << 48) | ((unsigned long long)(unsigned)en.w << 24) | (unsigned long long)i;
      }
      ++n;
    }
  }
  return n;
}

// The Q pairs of a workgroup that have a cell at all are gathered first, so that the cone (<= 56 rotated samples, the
// expensive part) runs on full waves: about one Q pair in four has one, scattered over the waves.
__global__ __launch_bounds__(256) void bq_match(CsBatchArgs a) {
  __shared__ uint32_t s_items[256];
  __shared__ int s_first[256];
  __shared__ uint32_t s_col[256][11];   // the cones' bit sets (352 direction bins)
  __shared__ unsigned short s_base[256];
  __shared__ uint32_t s_wcnt[4];
  __shared__ int s_b0, s_b1;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the bases of the workgroup's first and last Q pair: nearly always the same one (a base has ~1000 Q pairs), so a thread
  // searches a range of one
  if (threadIdx.x == 0) s_b0 = base_of(a.bases, a.nb, blockIdx.x * blockDim.x, true);
  if (threadIdx.x == 64) s_b1 = base_of(a.bases, a.nb, min(blockIdx.x * blockDim.x + blockDim.x - 1u, a.total_q - 1u), true);
  __syncthreads();
  int b = s_b0;
  if (t < a.total_q)
    for (const int b1 = s_b1; b < b1 && a.bases[b + 1].q_flat <= t;) ++b;   // last base whose first flat index is <= t
  int first = -1;
  const bool any = t < a.total_q && bq_has_cell(a, t, b, &first);
  const unsigned long long m = __ballot(any);
  if (lane == 0) s_wcnt[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < wave; ++w) off += s_wcnt[w];
  const uint32_t n_any = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
  if (any) {
    const uint32_t slot = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    s_items[slot] = t;
    s_first[slot] = first;
    s_base[slot] = (unsigned short)b;   // at most 65535 bases per call
  }
  __syncthreads();
  // the cones: kConeLanes lanes per Q pair (a wave of 64 pairs walked its 56 samples for ~35 us while most of the chip had
  // nothing to do: the workgroups of this launch are fewer than two per compute unit), their bit sets ORed over the group
  constexpr int kConeLanes = 4;
  for (uint32_t it0 = 0; it0 < n_any; it0 += 256 / kConeLanes) {
    const uint32_t item = it0 + threadIdx.x / kConeLanes;
    uint32_t colored[11];
#pragma unroll
    for (int w = 0; w < 11; ++w) colored[w] = 0u;
    if (item < n_any) bq_cone_mask(a, s_items[item], (int)s_base[item], (int)(threadIdx.x & (kConeLanes - 1)), kConeLanes, colored);
#pragma unroll
    for (int off = 1; off < kConeLanes; off <<= 1)
#pragma unroll
      for (int w = 0; w < 11; ++w) colored[w] |= __shfl_xor(colored[w], off, 64);
    if (item < n_any && (threadIdx.x & (kConeLanes - 1)) == 0)
#pragma unroll
      for (int w = 0; w < 11; ++w) s_col[item][w] = colored[w];
  }
  __syncthreads();
  // The walks: count first, then ONE reservation of key slots per workgroup (and one add on the base's counter where the
  // whole workgroup works on one base: nearly always).  A reservation per thread put ~8000 atomic adds on ONE word: 75 of the
  // launch's 87 us went into that queue (PGP_BQ_ABLATE builds: look-up 20, cones +9, walks +75 us of the stage).
  uint32_t kept[kBqKeep];
  uint32_t found = 0;
  const bool mine = threadIdx.x < n_any;
  const uint32_t my_t = mine ? s_items[threadIdx.x] : 0u;
  const int my_b = mine ? (int)s_base[threadIdx.x] : 0;
  if (mine) found = bq_walk<false>(a, my_t, my_b, s_first[threadIdx.x], s_col[threadIdx.x], kept, 0u);
  uint32_t incl = found;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  __syncthreads();   // (s_wcnt is read above by every wave)
  if (lane == 63) s_wcnt[wave] = incl;
  __syncthreads();
  __shared__ uint32_t s_out0;
  const uint32_t total = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
  if (total == 0u) return;
  const bool one_base = s_b0 == s_b1;
  if (threadIdx.x == 0) {
    s_out0 = atomicAdd(a.n_keys, total);
    if (one_base) atomicAdd(&a.base_cnt[s_b0], total);
  }
  if (!one_base && found) atomicAdd(&a.base_cnt[my_b], found);
  __syncthreads();
  uint32_t out0 = s_out0 + incl - found;
  for (int w = 0; w < wave; ++w) out0 += s_wcnt[w];
  if (found == 0u) return;
  if (found <= (uint32_t)kBqKeep) {
    const uint32_t i = my_t - a.bases[my_b].q_flat;
#pragma unroll
    for (int q = 0; q < kBqKeep; ++q)
      if ((uint32_t)q < found && out0 + q < a.key_cap)
        a.keys[out0 + q] = ((unsigned long long)(unsigned)my_b << 48) | ((unsigned long long)kept[q] << 24) | (unsigned long long)i;
  } else {
    (void)bq_walk<true>(a, my_t, my_b, s_first[threadIdx.x], s_col[threadIdx.x], kept, out0);
  }
}

// starts of the bases in the sorted keys = exclusive prefix sums of their match counts (one workgroup; nb is a few hundred)
__global__ __launch_bounds__(256) void batch_base_starts(const uint32_t* __restrict__ base_cnt, int nb,
                                                         uint32_t* __restrict__ base_start) {
  __shared__ uint32_t s_sum[256];
  const int per = (nb + 255) / 256, lo = min((int)threadIdx.x * per, nb), hi = min(lo + per, nb);
  uint32_t sum = 0;
  for (int b = lo; b < hi; ++b) sum += base_cnt[b];
  s_sum[threadIdx.x] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (int t = 0; t < (int)threadIdx.x; ++t) run += s_sum[t];
  for (int b = lo; b < hi; ++b) {
    base_start[b] = run;
    run += base_cnt[b];
  }
  if (threadIdx.x == 255) base_start[nb] = run;   // the last thread's running sum is the total
}

// picks (base, j) -> the j-th quad of that base in the reference's order
__global__ void batch_gather(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ base_start,
                             const BatchBase* __restrict__ bases, const int2* __restrict__ pairs,
                             const int2* __restrict__ picks, int m, int4* __restrict__ quads) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const int2 pk = picks[t];
  const BatchBase B = bases[pk.x];
  const unsigned long long key = keys[base_start[pk.x] + (uint32_t)pk.y];
  const int2 p = pairs[B.p_off + (uint32_t)((key >> 24) & 0xFFFFFFull)], q = pairs[B.q_off + (uint32_t)(key & 0xFFFFFFull)];
  quads[t] = make_int4(p.x, p.y, q.x, q.y);
}


// ---------------- classic 4PCS quad search (Match4PCS::FindCongruentQuadrilaterals) ----------------
// S4/algorithms/4pcs.cc:61-103: the invariant points e1 = p1 + invariant1 (p2 - p1) of the P-pairs go
// into a kd-tree; for every Q-pair the tree is asked for all e1 with |e1 - e2|^2 < distance_threshold2
// (strict, S4/accelerators/kdtree.h:491; the UNSQUARED threshold is compared with the squared
// distance, as in the Super4PCS variant) and each hit id emits (P_pairs[id / 2], Q_pairs[i]) -- the
// `id / 2` is the reference's (its tree holds ONE point per pair, so the quad names the wrong pair for
// every odd id); it is reproduced as written.  On the device the range query is a tiled scan: a
// thread per Q-pair, the invariant points of 1024 P-pairs per LDS tile, hits emitted in (i, id) order.
__global__ __launch_bounds__(256) void invariant_points(const float4* __restrict__ Qw, int nQs, const int2* __restrict__ Pp,
                                                        int nP, float inv1, float4* __restrict__ e1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nP) return;
  const int2 pr = Pp[i];
  const float qnan = __int_as_float(0x7FC00000);
  if ((unsigned)pr.x >= (unsigned)nQs || (unsigned)pr.y >= (unsigned)nQs) {
    e1[i] = make_float4(qnan, qnan, qnan, 0.f);   // never within any range
    return;
  }
  const V3 p = lerp_pt(ld3(Qw, pr.x), ld3(Qw, pr.y), inv1);
  e1[i] = make_float4(p.x, p.y, p.z, 0.f);
}

template <bool FILL>
__global__ __launch_bounds__(256) void range_match(const float4* __restrict__ Qw, int nQs, const float4* __restrict__ e1,
                                                   int nP, const int2* __restrict__ Pp, const int2* __restrict__ Qp, int nQ,
                                                   float inv2, float sqdist, uint32_t* __restrict__ q_cnt,
                                                   const uint32_t* __restrict__ q_start, int4* __restrict__ quads,
                                                   uint32_t cap) {
  __shared__ float4 s_e[1024];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < nQ;
  int2 qr = make_int2(0, 0);
  V3 e2 = {0.f, 0.f, 0.f};
  bool ok = false;
  if (live) {
    qr = Qp[i];
    ok = (unsigned)qr.x < (unsigned)nQs && (unsigned)qr.y < (unsigned)nQs;
    if (ok) e2 = lerp_pt(ld3(Qw, qr.x), ld3(Qw, qr.y), inv2);
  }
  uint32_t found = 0;
  const uint32_t out0 = (FILL && live) ? q_start[i] : 0u;
  for (int t0 = 0; t0 < nP; t0 += 1024) {
    const int tn = min(1024, nP - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < tn; j += blockDim.x) s_e[j] = e1[t0 + j];
    __syncthreads();
    if (!ok) continue;
    for (int j = 0; j < tn; ++j) {
      const float4 e = s_e[j];
      const V3 d = {sub(e2.x, e.x), sub(e2.y, e.y), sub(e2.z, e.z)};
      if (sqnorm(d) < sqdist) {   // strict; NaN never
        if (FILL && out0 + found < cap) {
          const int2 pp = Pp[(t0 + j) / 2];   // sic: P_pairs[id / 2]
          quads[out0 + found] = make_int4(pp.x, pp.y, qr.x, qr.y);
        }
        ++found;
      }
    }
  }
  if (!FILL && live) q_cnt[i] = found;
}

}  // namespace

// PairCreationFunctor::synch3DContent (pairCreationFunctor.h:102-138): centre + ratio of the
// unit-cube normalisation, and the normalised points.  Host, O(n), float/double as the reference.
void unit_cube_image(const float* xyz, int n, float gcenter[3], float* ratio, std::vector<float4>* unit) {
  float mn[3] = {FLT_MAX / 2, FLT_MAX / 2, FLT_MAX / 2}, mx[3] = {-FLT_MAX / 2, -FLT_MAX / 2, -FLT_MAX / 2};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      float v = xyz[3 * (size_t)i + k];
      if (v < mn[k]) mn[k] = v;
      if (v > mx[k]) mx[k] = v;
    }
  float ext[3];
  for (int k = 0; k < 3; ++k) {
    ext[k] = mx[k] - mn[k];
    gcenter[k] = mn[k] + (ext[k] / 2.0f);
  }
  double r = (double)ext[2] + 0.001, w = (double)ext[1] + 0.001, h = (double)ext[0] + 0.001;
  double m = w > h ? w : h;
  m = r > m ? r : m;
  *ratio = (float)m;
  unit->resize((size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    float u[3];
    for (int k = 0; k < 3; ++k) {
      float d = xyz[3 * (size_t)i + k] - gcenter[k];
      u[k] = d / *ratio + 0.5f;  // worldToUnit: (p - gcenter) / ratio + half
    }
    (*unit)[i] = make_float4(u[0], u[1], u[2], 0.f);
  }
}

int launch_extract_pairs(pgp_ctx* ctx, float pair_distance, float eps, int* d_pairs, int cap,
                         int* n_pairs_host, hipStream_t st) {
  const int n = ctx->nQs;
  *n_pairs_host = 0;
  if (n <= 0 || !ctx->d_Qs.p) {
    set_error("no search model: call pgp_set_search_model first");
    return PGP_ESTATE;
  }
  int rc;
  if ((rc = ctx->d_cs_cnt.ensure(((size_t)n + 1) * 8 + 64)) != PGP_OK) return rc;
  if ((rc = ctx->d_scan_tmp.ensure(((size_t)n / 2048 + 2) * 4)) != PGP_OK) return rc;
  uint32_t* cnt = ctx->d_cs_cnt.as<uint32_t>();
  uint32_t* start = cnt + (n + 1);
  PGP_HIP(hipMemsetAsync(cnt, 0, ((size_t)n + 1) * 4, st));
  dim3 grid((n + 3) / 4);
  hipLaunchKernelGGL(pair_rows<false>, grid, dim3(256), 0, st, ctx->d_Qs.as<float4>(), n,
                     (double)pair_distance, (double)eps, cnt, (const uint32_t*)nullptr, (int2*)nullptr, 0u);
  if ((rc = device_exclusive_scan(cnt, start, (size_t)n + 1, ctx->d_scan_tmp.as<uint32_t>(), st)) != PGP_OK) return rc;
  uint32_t total = 0;
  PGP_HIP(hipMemcpyAsync(&total, start + n, 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  *n_pairs_host = (int)(2u * total);
  if (total > 0 && cap > 0)
    hipLaunchKernelGGL(pair_rows<true>, grid, dim3(256), 0, st, ctx->d_Qs.as<float4>(), n,
                       (double)pair_distance, (double)eps, cnt, (const uint32_t*)start,
                       reinterpret_cast<int2*>(d_pairs), (uint32_t)cap);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

int launch_find_congruent(pgp_ctx* ctx, const float base[12], float inv1, float inv2, float threshold,
                          const int* d_Pp, int nP, const int* d_Qp, int nQ, int* d_quads, int cap,
                          int* n_quads_host, hipStream_t st) {
  *n_quads_host = 0;
  if (ctx->nQs <= 0 || !ctx->d_Qs.p) {
    set_error("no search model: call pgp_set_search_model first");
    return PGP_ESTATE;
  }
  if (nP <= 0 || nQ <= 0) return PGP_OK;
  CsArgs a{};
  const float eps = threshold / ctx->cs_ratio;          // getNormalizedEpsilon
  const int gridDepth = (int)(-std::log2(eps));          // normalset.h:116
  if (!(eps > 0.f) || gridDepth < 0 || gridDepth > 20) {
    set_error("find_congruent: threshold %g gives an unusable grid depth %d", (double)threshold, gridDepth);
    return PGP_EINVAL;
  }
  a.eg = (int)std::pow(2, gridDepth);
  a.epsilon = 1.f / (float)a.eg;
  a.nepsilon = (float)((double)(1.0f / 7.0f) + 0.00001);  // normalset.h:88
  // per-base constants, host libm exactly as the reference evaluates them (cone_for_base)
  ConeTable cone{};
  cone.nb = cone_for_base(base, &cone.v[0][0]);
  // ---- buffers -------------------------------------------------------------------------------
  unsigned nbk = 1024;
  while (nbk < 2u * (unsigned)nP && nbk < (1u << 24)) nbk <<= 1;
  a.bmask = nbk - 1;
  int rc;
  const size_t n_cnt = (size_t)nbk + 1 + (size_t)nQ + 1;
  if ((rc = ctx->d_cs_cnt.ensure(n_cnt * 8 + 64)) != PGP_OK) return rc;
  if ((rc = ctx->d_cs_entries.ensure((size_t)nP * 16 + 16)) != PGP_OK) return rc;
  if ((rc = ctx->d_scan_tmp.ensure((n_cnt / 2048 + 2) * 4)) != PGP_OK) return rc;
  uint32_t* bcnt = ctx->d_cs_cnt.as<uint32_t>();
  uint32_t* bstart = bcnt + (nbk + 1);
  uint32_t* qcnt = bstart + (nbk + 1);
  uint32_t* qstart = qcnt + (nQ + 1);
  a.Qw = ctx->d_Qs.as<float4>();
  a.Qu = ctx->d_Qs_unit.as<float4>();
  a.nQs = ctx->nQs;
  a.Pp = reinterpret_cast<const int2*>(d_Pp);
  a.nP = nP;
  a.Qp = reinterpret_cast<const int2*>(d_Qp);
  a.nQ = nQ;
  a.inv1 = inv1;
  a.inv2 = inv2;
  a.threshold = threshold;
  a.bucket_cnt = bcnt;
  a.bucket_start = bstart;
  a.entries = ctx->d_cs_entries.as<int4>();
  a.q_cnt = qcnt;
  a.q_start = qstart;
  a.cap = (uint32_t)(cap > 0 ? cap : 0);
  uint32_t* scan_tmp = ctx->d_scan_tmp.as<uint32_t>();
  const dim3 gp((nP + 255) / 256), gq((nQ + 127) / 128);
  PGP_HIP(hipMemsetAsync(bcnt, 0, ((size_t)nbk + 1) * 4, st));
  hipLaunchKernelGGL(p_entries<false>, gp, dim3(256), 0, st, a);
  if ((rc = device_exclusive_scan(bcnt, bstart, (size_t)nbk + 1, scan_tmp, st)) != PGP_OK) return rc;
  PGP_HIP(hipMemsetAsync(bcnt, 0, ((size_t)nbk + 1) * 4, st));
  hipLaunchKernelGGL(p_entries<true>, gp, dim3(256), 0, st, a);
  PGP_HIP(hipMemsetAsync(qcnt + nQ, 0, 4, st));
  hipLaunchKernelGGL(q_match<false>, gq, dim3(128), 0, st, a, cone);
  if ((rc = device_exclusive_scan(qcnt, qstart, (size_t)nQ + 1, scan_tmp, st)) != PGP_OK) return rc;
  uint32_t total = 0;
  PGP_HIP(hipMemcpyAsync(&total, qstart + nQ, 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  *n_quads_host = (int)total;
  if (total == 0 || cap <= 0) return PGP_OK;
  // all matches are materialised and sorted, then the first `cap` are emitted: truncation keeps
  // the reference's (id, i) order
  size_t sort_bytes = 0;
  hipError_t he = rocprim::radix_sort_keys(nullptr, sort_bytes, (unsigned long long*)nullptr,
                                           (unsigned long long*)nullptr, (size_t)total, 0, 64, st);
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_keys (size query) failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  ctx->csb_fit_m = 0;
  ctx->csb_nb = 0;   // the batch's sorted keys (if any) are overwritten here
  if ((rc = ctx->d_cs_keys.ensure((size_t)total * 16 + sort_bytes + 256)) != PGP_OK) return rc;
  unsigned long long* keys_in = ctx->d_cs_keys.as<unsigned long long>();
  unsigned long long* keys_out = keys_in + total;
  void* sort_tmp = keys_out + total;
  a.keys = keys_in;
  a.cap = total;
  hipLaunchKernelGGL(q_match<true>, gq, dim3(128), 0, st, a, cone);
  he = rocprim::radix_sort_keys(sort_tmp, sort_bytes, keys_in, keys_out, (size_t)total, 0, 64, st);
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_keys failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  const uint32_t n_emit = total < (uint32_t)cap ? total : (uint32_t)cap;
  hipLaunchKernelGGL(emit_quads, dim3((n_emit + 255) / 256), dim3(256), 0, st,
                     (const unsigned long long*)keys_out, n_emit, a.Pp, a.Qp, reinterpret_cast<int4*>(d_quads));
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}


namespace {

// Per-base constants of FindCongruentQuadrilaterals that need the host libm (super4pcs.cc:107-109,
// normalset.hpp:175-196): the cone of half-angle acos(<u01, u23>) as <= 56 sample vectors.
int cone_for_base(const float base[12], float* cone /*[56][3]*/) {
  auto normalized_h = [](const float v[3], float o[3]) {
    float x = v[0] * v[0], y = v[1] * v[1], z = v[2] * v[2];
    float t = y + z;
    float s = x + t;
    if (s > 0.f) {
      float n = std::sqrt(s);
      o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
    } else {
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    }
  };
  float d01[3], d23[3], u01[3], u23[3];
  for (int k = 0; k < 3; ++k) {
    d01[k] = base[3 + k] - base[k];
    d23[k] = base[9 + k] - base[6 + k];
  }
  normalized_h(d01, u01);
  normalized_h(d23, u23);
  float cosAlpha;
  {
    float x = u01[0] * u23[0], y = u01[1] * u23[1], z = u01[2] * u23[2];
    float t = y + z;
    cosAlpha = x + t;
  }
  const float alpha = std::acos(cosAlpha);
  const float perimeter = (float)((double)2.0f * M_PI * (double)std::atan(alpha));
  const float nbf = 2 * std::ceil(perimeter * 7.0f / 2.0f);
  const unsigned nb = (nbf == nbf && nbf > 0.f && nbf <= 56.f) ? (unsigned)nbf : 0u;
  const float angleStep = (float)((double)2.0f * M_PI / (double)(float)nb);
  const float sinAlpha = std::sin(alpha);
  // cos / sin of the steps depend on nb alone (an even count up to 56): kept per thread -- a hundred bases were 6000 libm
  // calls, most of the 27 us this host step took between two device phases of pgp_find_congruent_batch.  Same floats.
  struct Steps {
    bool have = false;
    float c[56], s[56];
  };
  static thread_local Steps steps[57];
  Steps& t = steps[nb];
  if (!t.have) {
    for (unsigned s = 0; s < nb; ++s) {
      const float theta = (float)s * angleStep;
      t.c[s] = std::cos(theta);
      t.s[s] = std::sin(theta);
    }
    t.have = true;
  }
  for (unsigned s = 0; s < nb; ++s) {
    cone[3 * s] = sinAlpha * t.c[s];
    cone[3 * s + 1] = sinAlpha * t.s[s];
    cone[3 * s + 2] = cosAlpha;
  }
  return (int)nb;
}

}  // namespace

// Phase 1 + 2 for all bases: leaves the sorted match keys and the per-base starts in the context.
int launch_find_congruent_batch(pgp_ctx* ctx, const int* h_base_ids, const float* h_base_xyz, const float* h_inv,
                                const int* h_rows, int nb, float threshold, int* h_n_quads, hipStream_t st) {
  ctx->csb_fit_m = 0;
  ctx->csb_nb = 0;
  if (ctx->nQs <= 0 || !ctx->d_Qs.p) {
    set_error("no search model: call pgp_set_search_model first");
    return PGP_ESTATE;
  }
  if (!ctx->ppf_ready || ctx->ppf_n_pairs <= 0) {
    set_error("no pair-feature table with pair lists: call pgp_set_ppf_map(keys, counts, pairs) first");
    return PGP_ESTATE;
  }
  if (nb > 65535) {
    set_error("pgp_find_congruent_batch: at most 65535 bases per call");
    return PGP_EINVAL;
  }
  for (int b = 0; b < nb; ++b) h_n_quads[b] = 0;
  if (nb == 0) return PGP_OK;
  // PGP_CS_TIMING=1: host time of the call's stages on stderr (each stage then ends with a stream synchronisation)
  static const bool timing = getenv("PGP_CS_TIMING") && atoi(getenv("PGP_CS_TIMING")) != 0;
  auto t_last = std::chrono::steady_clock::now();
  auto stage = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[congruent batch] %-28s %7.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_last).count());
    t_last = now;
  };
  // ---- which rows of the table are pairs1 / pairs6 of every base: computePPF on the device
  std::vector<int> edge_pairs((size_t)nb * 4), rows((size_t)nb * 2);
  for (int b = 0; b < nb; ++b) {
    edge_pairs[4 * b] = h_base_ids[4 * b];       // computePPF(base_id1, base_id2)
    edge_pairs[4 * b + 1] = h_base_ids[4 * b + 1];
    edge_pairs[4 * b + 2] = h_base_ids[4 * b + 2];   // computePPF(base_id3, base_id4)
    edge_pairs[4 * b + 3] = h_base_ids[4 * b + 3];
  }
  int rc = PGP_OK;
  if (h_rows) {   // the caller holds them already (pgp_select_bases_rows): one launch and one round trip less
    const int n_rows = (int)ctx->ppf_off_host.size() - 1;
    for (int k = 0; k < 2 * nb; ++k) {
      if (h_rows[k] < -1 || h_rows[k] >= n_rows) {
        set_error("pgp_find_congruent_batch_rows: row %d of base %d is not a row of the table (%d rows)", h_rows[k], k / 2, n_rows);
        return PGP_EINVAL;
      }
      rows[(size_t)k] = h_rows[k];
    }
  } else {
    rc = launch_ppf_features(ctx, edge_pairs.data(), 2 * nb, nullptr, rows.data(), st);
    if (rc != PGP_OK) return rc;
  }
  stage("pair features of the bases");
  const float eps = threshold / ctx->cs_ratio;          // getNormalizedEpsilon
  const int gridDepth = (int)(-std::log2(eps));          // normalset.h:116
  if (!(eps > 0.f) || gridDepth < 0 || gridDepth > 20) {
    set_error("find_congruent: threshold %g gives an unusable grid depth %d", (double)threshold, gridDepth);
    return PGP_EINVAL;
  }
  std::vector<BatchBase> hb((size_t)nb);
  std::vector<float> cones((size_t)nb * 168, 0.f);
  uint64_t tp = 0, tq = 0;
  for (int b = 0; b < nb; ++b) {
    BatchBase& B = hb[b];
    B.inv1 = h_inv[2 * b];
    B.inv2 = h_inv[2 * b + 1];
    const int r1 = rows[2 * b], r6 = rows[2 * b + 1];
    B.p_off = B.p_cnt = B.q_off = B.q_cnt = 0;
    if (r1 >= 0 && r6 >= 0) {   // pairs1.size() == 0 || pairs6.size() == 0 -> no quads for this base
      const uint32_t c1 = ctx->ppf_off_host[r1 + 1] - ctx->ppf_off_host[r1], c6 = ctx->ppf_off_host[r6 + 1] - ctx->ppf_off_host[r6];
      if (c1 > 0 && c6 > 0) {
        B.p_off = ctx->ppf_off_host[r1];
        B.p_cnt = c1;
        B.q_off = ctx->ppf_off_host[r6];
        B.q_cnt = c6;
      }
    }
    if (B.p_cnt >= (1u << 24) || B.q_cnt >= (1u << 24)) {
      set_error("pgp_find_congruent_batch: a pair list exceeds 2^24 entries");
      return PGP_EINVAL;
    }
    B.p_flat = (uint32_t)tp;
    B.q_flat = (uint32_t)tq;
    tp += B.p_cnt;
    tq += B.q_cnt;
    B.cone_nb = B.p_cnt ? cone_for_base(h_base_xyz + 12 * (size_t)b, cones.data() + 168 * (size_t)b) : 0;
    B.pad = 0;
  }
  if (tp >= (1ull << 31) || tq >= (1ull << 31)) {
    set_error("pgp_find_congruent_batch: more than 2^31 pairs in one batch");
    return PGP_EINVAL;
  }
  if (tp == 0 || tq == 0) return PGP_OK;
  stage("rows + cones (host)");
  CsBatchArgs a{};
  a.eg = (int)std::pow(2, gridDepth);
  a.epsilon = 1.f / (float)a.eg;
  a.nepsilon = (float)((double)(1.0f / 7.0f) + 0.00001);  // normalset.h:88
  unsigned nbk = 1024;
  while (nbk < 2u * (unsigned)tp && nbk < (1u << 26)) nbk <<= 1;
  a.bmask = nbk - 1;
  // bucket heads | entry links | {matches appended, matches per base}
  const size_t n_cnt = (size_t)nbk + (size_t)tp + (size_t)nb + 4;
  if ((rc = ctx->d_cs_cnt.ensure(n_cnt * 4 + 64)) != PGP_OK) return rc;
  if ((rc = ctx->d_cs_entries.ensure((size_t)tp * 16 + 16)) != PGP_OK) return rc;
  // bases | cones | base_start, in one staging buffer
  const size_t bb = ((size_t)nb * sizeof(BatchBase) + 255) & ~(size_t)255, cb = ((size_t)nb * 168 * 4 + 255) & ~(size_t)255;
  if ((rc = ctx->d_csb.ensure(bb + cb + ((size_t)nb + 1) * 4 + 64)) != PGP_OK) return rc;
  unsigned char* sb = ctx->d_csb.as<unsigned char>();
  BatchBase* d_bases = reinterpret_cast<BatchBase*>(sb);
  float* d_cones = reinterpret_cast<float*>(sb + bb);
  uint32_t* d_base_start = reinterpret_cast<uint32_t*>(sb + bb + cb);
  // bases and cones in ONE copy (a pageable copy costs the call ~10 us whatever its size); the staging vector lives until
  // the stream is synchronised below
  // (in the context's pinned area when there is room: the pageable form costs the host ~10 us)
  HostOut staging(ctx, st);
  std::vector<unsigned char> stage_h;
  unsigned char* stage_p = staging.room(bb + cb);
  if (!stage_p) {
    stage_h.assign(bb + cb, 0);
    stage_p = stage_h.data();
  }
  std::memcpy(stage_p, hb.data(), (size_t)nb * sizeof(BatchBase));
  std::memcpy(stage_p + bb, cones.data(), (size_t)nb * 168 * 4);
  if (stage_h.empty()) {   // (the pinned area: by a kernel on the stream)
    if ((rc = stage_to_device(st, d_bases, stage_p, bb + cb)) != PGP_OK) return rc;
  } else {
    PGP_HIP(hipMemcpyAsync(d_bases, stage_p, bb + cb, hipMemcpyHostToDevice, st));
  }
  int* head = ctx->d_cs_cnt.as<int>();
  int* next = head + nbk;
  uint32_t* qcnt = reinterpret_cast<uint32_t*>(next + tp);
  a.Qw = ctx->d_Qs.as<float4>();
  a.Qu = ctx->d_Qs_unit.as<float4>();
  a.nQs = ctx->nQs;
  a.pairs = ctx->d_ppf_pairs.as<int2>();
  a.bases = d_bases;
  a.nb = nb;
  a.cones = d_cones;
  a.threshold = threshold;
  a.head = head;
  a.next = next;
  a.entries = ctx->d_cs_entries.as<int4>();
  a.total_p = (uint32_t)tp;
  a.total_q = (uint32_t)tq;
  const dim3 gp((unsigned)((tp + 255) / 256)), gq((unsigned)((tq + 255) / 256));
  const uint32_t n_init = std::max<uint32_t>(nbk, (uint32_t)nb + 1);
  hipLaunchKernelGGL(bp_init, dim3((n_init + 255) / 256), dim3(256), 0, st, head, (uint32_t)nbk, qcnt, (uint32_t)nb + 1);
  hipLaunchKernelGGL(bp_entries, gp, dim3(256), 0, st, a);
  stage("P entries into buckets");
  // the matches: appended to a key array sized by a guess (twice the last batch's total, at least 64 k); a batch that
  // outgrows it is matched again into a larger one
  std::vector<uint32_t> starts((size_t)nb + 1);
  uint32_t total = 0;
  uint32_t* d_nkeys = qcnt;            // [1] appended so far | [nb] per base (the Q counters of the two-pass form)
  uint32_t* d_base_cnt = qcnt + 1;
  size_t sort_bytes = 0;
  bool repeat = false;
  // (PGP_CS_KEY_CAP=n: the first guess, for the test that makes a batch outgrow it)
  static const size_t cap_env = getenv("PGP_CS_KEY_CAP") ? (size_t)std::max(1, atoi(getenv("PGP_CS_KEY_CAP"))) : 0;
  for (size_t cap = cap_env ? cap_env : std::max<size_t>((size_t)ctx->csb_cap_hint, (size_t)1 << 16);;) {
    hipError_t he = rocprim::radix_sort_keys(nullptr, sort_bytes, (unsigned long long*)nullptr,
                                             (unsigned long long*)nullptr, cap, 0, 64, st);
    if (he != hipSuccess) {
      set_error("rocprim::radix_sort_keys (size query) failed: %s", hipGetErrorString(he));
      return PGP_EHIP;
    }
    if ((rc = ctx->d_cs_keys.ensure(cap * 16 + sort_bytes + 256)) != PGP_OK) return rc;
    a.keys = ctx->d_cs_keys.as<unsigned long long>();
    a.key_cap = (uint32_t)cap;
    a.n_keys = d_nkeys;
    a.base_cnt = d_base_cnt;
    if (repeat) PGP_HIP(hipMemsetAsync(d_nkeys, 0, ((size_t)nb + 1) * 4, st));   // (the first round: bp_init did)
    repeat = true;
    hipLaunchKernelGGL(bq_match, gq, dim3(256), 0, st, a);
    hipLaunchKernelGGL(batch_base_starts, dim3(1), dim3(256), 0, st, (const uint32_t*)d_base_cnt, nb, d_base_start);
    {
      HostOut out(ctx, st);
      if ((rc = out.to(starts.data(), d_base_start, ((size_t)nb + 1) * 4)) != PGP_OK || (rc = out.sync()) != PGP_OK) return rc;
    }
    total = starts[nb];
    if ((size_t)total <= cap) {
      ctx->csb_keys_off = (uint32_t)cap;
      break;
    }
    cap = (size_t)total + (size_t)total / 4;   // every key counted: this one fits
  }
  stage("Q matches");
  if (timing) fprintf(stderr, "[congruent batch] %d bases, %llu P pairs, %llu Q pairs, %u buckets, %u matches\n", nb,
                      (unsigned long long)tp, (unsigned long long)tq, nbk, total);
  ctx->csb_cap_hint = std::max<uint32_t>(2u * total, 1u << 16);
  for (int b = 0; b < nb; ++b) h_n_quads[b] = (int)(starts[b + 1] - starts[b]);
  if (total == 0) {   // a valid, empty batch: every pick is out of range
    ctx->csb_starts = starts;
    ctx->csb_nb = nb;
    ctx->csb_total = 0;
    return PGP_OK;
  }
  unsigned long long* keys_in = ctx->d_cs_keys.as<unsigned long long>();
  unsigned long long* keys_out = keys_in + ctx->csb_keys_off;
  void* sort_tmp = keys_out + ctx->csb_keys_off;
  // (Round 6 tried the keys in order WITHOUT the device-wide radix sort -- scattered to their base's segment, every segment sorted
  //  by one workgroup in LDS (bitonic): ~190 us against rocPRIM's ~35, because ONE base of the drop-in's case holds 4389 of the
  //  13 928 keys and a bitonic sort of 8192 64-bit keys by 256 threads is 91 passes over LDS; profiles/r06_ab/device_draw.log.)
  hipError_t he = rocprim::radix_sort_keys(sort_tmp, sort_bytes, keys_in, keys_out, (size_t)total, 0, 64, st);
  if (he != hipSuccess) {
    set_error("rocprim::radix_sort_keys failed: %s", hipGetErrorString(he));
    return PGP_EHIP;
  }
  PGP_HIP(hipGetLastError());
  stage("sort");
  // only now do the sorted keys exist: the batch becomes visible to pgp_congruent_batch_quads / _fit
  ctx->csb_starts = starts;
  ctx->csb_nb = nb;
  ctx->csb_total = total;
  return PGP_OK;
}

// picks[m][2] = (base, j) -> d_quads[m] (int4), on the sorted keys left by the call above
int launch_congruent_batch_gather(pgp_ctx* ctx, const int* h_picks, int m, int4* d_quads, hipStream_t st, const int2* d_picks_there) {
  if (ctx->csb_nb <= 0 || (int)ctx->csb_starts.size() != ctx->csb_nb + 1) {
    set_error("no congruent batch: call pgp_find_congruent_batch first (a later pgp_find_congruent, "
              "pgp_set_ppf_map or pgp_set_search_model discards it)");
    return PGP_ESTATE;
  }
  if (!h_picks && !d_picks_there) {
    set_error("congruent batch: no picks");
    return PGP_EINVAL;
  }
  // (picks that were DRAWN on the device from the batch's own quad counts -- pgp_api.hip sample_quads_kernel -- name quads
  //  that exist by construction: no host copy to check)
  for (int k = 0; h_picks && k < m; ++k) {   // a bad pick would read past the sorted keys on the device
    const int b = h_picks[2 * (size_t)k], j = h_picks[2 * (size_t)k + 1];
    if (b < 0 || b >= ctx->csb_nb) {
      set_error("congruent batch: pick %d names base %d of %d", k, b, ctx->csb_nb);
      return PGP_EINVAL;
    }
    const uint32_t nq = ctx->csb_starts[b + 1] - ctx->csb_starts[b];
    if (j < 0 || (uint32_t)j >= nq) {
      set_error("congruent batch: pick %d names quad %d of %u of base %d", k, j, nq, b);
      return PGP_EINVAL;
    }
  }
  const int2* d_picks = d_picks_there;   // (the caller has queued their upload already, with other things of its own)
  if (!d_picks) {
    int rc = ctx->d_csb_picks.ensure((size_t)m * 8 + 16);
    if (rc != PGP_OK) return rc;
    PGP_HIP(hipMemcpyAsync(ctx->d_csb_picks.p, h_picks, (size_t)m * 8, hipMemcpyHostToDevice, st));
    d_picks = ctx->d_csb_picks.as<int2>();
  }
  const int nb = ctx->csb_nb;
  const size_t bb = ((size_t)nb * sizeof(BatchBase) + 255) & ~(size_t)255, cb = ((size_t)nb * 168 * 4 + 255) & ~(size_t)255;
  unsigned char* sb = ctx->d_csb.as<unsigned char>();
  const BatchBase* d_bases = reinterpret_cast<const BatchBase*>(sb);
  const uint32_t* d_base_start = reinterpret_cast<const uint32_t*>(sb + bb + cb);
  const unsigned long long* keys_out = ctx->d_cs_keys.as<unsigned long long>() + ctx->csb_keys_off;
  hipLaunchKernelGGL(batch_gather, dim3((m + 255) / 256), dim3(256), 0, st, keys_out, d_base_start, d_bases,
                     (const int2*)ctx->d_ppf_pairs.as<int2>(), d_picks, m, d_quads);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}


// where the resident batch's per-base starts (nb + 1 words) live on the device
const uint32_t* congruent_batch_starts_device(pgp_ctx* ctx) {
  const int nb = ctx->csb_nb;
  const size_t bb = ((size_t)nb * sizeof(BatchBase) + 255) & ~(size_t)255, cb = ((size_t)nb * 168 * 4 + 255) & ~(size_t)255;
  return reinterpret_cast<const uint32_t*>(ctx->d_csb.as<unsigned char>() + bb + cb);
}

int launch_find_congruent_4pcs(pgp_ctx* ctx, float inv1, float inv2, float threshold, const int* d_Pp, int nP,
                               const int* d_Qp, int nQ, int* d_quads, int cap, int* n_quads_host, hipStream_t st) {
  *n_quads_host = 0;
  if (ctx->nQs <= 0 || !ctx->d_Qs.p) {
    set_error("no search model: call pgp_set_search_model first");
    return PGP_ESTATE;
  }
  if (nP <= 0 || nQ <= 0) return PGP_OK;
  int rc;
  if ((rc = ctx->d_cs_entries.ensure((size_t)nP * 16 + 16)) != PGP_OK) return rc;
  if ((rc = ctx->d_cs_cnt.ensure(((size_t)nQ + 1) * 8 + 64)) != PGP_OK) return rc;
  if ((rc = ctx->d_scan_tmp.ensure(((size_t)nQ / 2048 + 2) * 4)) != PGP_OK) return rc;
  float4* e1 = ctx->d_cs_entries.as<float4>();
  uint32_t* qcnt = ctx->d_cs_cnt.as<uint32_t>();
  uint32_t* qstart = qcnt + (nQ + 1);
  const float4* Qw = ctx->d_Qs.as<float4>();
  const int2* Pp = reinterpret_cast<const int2*>(d_Pp);
  const int2* Qp = reinterpret_cast<const int2*>(d_Qp);
  hipLaunchKernelGGL(invariant_points, dim3((nP + 255) / 256), dim3(256), 0, st, Qw, ctx->nQs, Pp, nP, inv1, e1);
  PGP_HIP(hipMemsetAsync(qcnt + nQ, 0, 4, st));
  const dim3 gq((nQ + 255) / 256);
  hipLaunchKernelGGL(range_match<false>, gq, dim3(256), 0, st, Qw, ctx->nQs, (const float4*)e1, nP, Pp, Qp, nQ, inv2,
                     threshold, qcnt, (const uint32_t*)nullptr, (int4*)nullptr, 0u);
  if ((rc = device_exclusive_scan(qcnt, qstart, (size_t)nQ + 1, ctx->d_scan_tmp.as<uint32_t>(), st)) != PGP_OK) return rc;
  uint32_t total = 0;
  PGP_HIP(hipMemcpyAsync(&total, qstart + nQ, 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  *n_quads_host = (int)total;
  if (total == 0 || cap <= 0) return PGP_OK;
  hipLaunchKernelGGL(range_match<true>, gq, dim3(256), 0, st, Qw, ctx->nQs, (const float4*)e1, nP, Pp, Qp, nQ, inv2,
                     threshold, qcnt, (const uint32_t*)qstart, reinterpret_cast<int4*>(d_quads), (uint32_t)cap);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

}  // namespace pgp
