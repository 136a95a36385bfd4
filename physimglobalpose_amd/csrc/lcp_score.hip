// csrc/lcp_score.hip -- batched LCP scoring of pose hypotheses on gfx950.
//
// Replaces, for a whole batch of transforms at once:
//   Match4PCSBase::Verify          S4/algorithms/match4pcsBase.cc:1699-1731  (PGP_MODE_PLAIN)
//   Match4PCSBase::WeightedVerify  S4/algorithms/match4pcsBase.cc:1733-1766  (PGP_MODE_WEIGHTED)
//   the verification loop + best bookkeeping of Perform_N_steps  :1885-1901
//   KdTree::doQueryRestrictedClosestIndex  S4/accelerators/kdtree.h:394-459  (via grid_index.hip)
//
// Mapping to the machine
//   lane            = one validation-model point (kept in VGPRs for the whole block: it is
//                     re-used by every hypothesis of the block's chunk, so it never needs LDS)
//   workgroup       = 256 model points (4 wave64) x a chunk of `hpb` hypotheses (8 at C2; the last
//                     10 % of a batch in chunks of 2 so that the launch ends on short workgroups)
//   hypothesis      = wave-uniform: its 4x4 is read with scalar loads (one 64 B line) and lives
//                     in SGPRs; the transform is 9 mul + 9 add per lane, no contraction
//   cell look-up    = one v_fma per axis leaves the lattice cell in the mantissa (cell_bits), the block
//                     coordinates are cut out with v_bfe: 9 VALU from position to word address
//   inlier count    = __ballot + popcount per wave, packed per group of 4 hypotheses on the SCALAR unit;
//                     weighted mode parks each lane's registered weight in LDS and reduces a group at
//                     once; 4 wave totals combined in LDS in fixed order, one (tile, hypothesis) partial
//                     per block, summed by finalize_scores in tile order => no atomics on the data
//                     path, bit-reproducible run to run
//   block -> (chunk, tile) is XCD-aware: blocks that share blockIdx%8 (one XCD, one L2) work on
//                     the same hypothesis chunks, so the cells those poses touch stay in that L2.
//
// Float parity: every arithmetic expression that decides an inlier is evaluated in the order
// Eigen evaluates it in the reference (pinned in oracle/, tests/test_oracle_vs_ref.py), each
// operation rounded separately (__fmul_rn/__fadd_rn; the file is also built -ffp-contract=off):
//   x'_r = ((m_r0*q0 + m_r1*q1) + m_r2*q2) + m_r3          (mat * q.homogeneous()).head<3>()
//   d2   = dx*dx + (dy*dy + dz*dz),  inlier iff d2 <= delta*delta    kdtree.h:423-424
//   n'_r = m_r0*n0 + (m_r1*n1 + m_r2*n2)                    mat.block<3,3>(0,0) * normal
//   dot  = a0*b0 + (a1*b1 + a2*b2)
// The acos/fold/<30 gate of base.cc:1756-1758 is monotone in the dot product, so it is applied
// as two thresholds found by bisection over the host libm's acosf (gate_thresholds): no device
// transcendental can disagree with the host at the gate boundary.

#include "pgp_internal.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#if defined(PGP_CAND8) && PGP_CAND8
#include <hip/hip_fp16.h>
#include <map>
#endif

namespace pgp {

namespace {

constexpr int kTile = 256;    // model points per workgroup (512 = 8 waves: +6 us per C2 step at its best hpb)
constexpr int kMaxHpb = 64;   // hypotheses per workgroup (LDS partial slots)

#if defined(PGP_ABLATE) && PGP_ABLATE == 10
// timing experiment: where a wave's time goes.  s_memtime stamps around the phases of a trip, summed per wave
// in SGPRs and added to g_phase at the end (read by pgp_debug_phase_cycles).  The stamps wait for the scalar
// loads in flight, so the figures are a decomposition, not the undisturbed kernel.
constexpr int kPhaseWaves = 65536;
__device__ unsigned long long g_phase[kPhaseWaves][8];   // one row per wave of the launch: no atomics (they would dominate)
#define PGP_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PGP_PHASE(idx, t0, t1) ph[idx] += (t1) - (t0)
#else
#define PGP_STAMP(var)
#define PGP_PHASE(idx, t0, t1)
#endif

struct Xf {  // one hypothesis, wave-uniform (SGPRs)
  float m00, m10, m20, m01, m11, m21, m02, m12, m22, m03, m13, m23;
};

__device__ __forceinline__ Xf load_xf(const float* __restrict__ T, uint32_t h) {
  const float4* c = reinterpret_cast<const float4*>(T + 16u * h);   // < 2^28 hypotheses  // column-major 4x4
  float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
  Xf x;
  x.m00 = c0.x; x.m10 = c0.y; x.m20 = c0.z;
  x.m01 = c1.x; x.m11 = c1.y; x.m21 = c1.z;
  x.m02 = c2.x; x.m12 = c2.y; x.m22 = c2.z;
  x.m03 = c3.x; x.m13 = c3.y; x.m23 = c3.z;
  return x;
}

__device__ __forceinline__ float xf_row(float a, float b, float c, float t, float q0, float q1, float q2) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, q0), __fmul_rn(b, q1)), __fmul_rn(c, q2)), t);
}

__device__ __forceinline__ float rot_row(float a, float b, float c, float n0, float n1, float n2) {
  return __fadd_rn(__fmul_rn(a, n0), __fadd_rn(__fmul_rn(b, n1), __fmul_rn(c, n2)));
}

__device__ __forceinline__ float sqdist(float x, float y, float z, float4 p) {
  float dx = __fsub_rn(x, p.x), dy = __fsub_rn(y, p.y), dz = __fsub_rn(z, p.z);
  return __fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dz, dz)));
}

// Sum over the 64 lanes in a FIXED association, on the DPP cross-lane path (no LDS traffic, unlike
// __shfl_xor = ds_bpermute): four row-local steps (quad swaps, half-row and row mirrors -- after
// them every lane of a 16-lane row holds the row total), then the four row totals (r0+r1)+(r2+r3).
template <int CTRL>
__device__ __forceinline__ float dpp_step(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
  return __fadd_rn(v, __int_as_float(moved));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_step<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_step<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_step<0x141>(v);  // row_half_mirror
  v = dpp_step<0x140>(v);  // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return __fadd_rn(__fadd_rn(r0, r1), __fadd_rn(r2, r3));
}

// Cell lookup: candidate run [s, e) for position (x,y,z); empty outside the grid / in empty cells.
// Branch-free on purpose: the kernel is instruction-issue bound and every `if` costs an exec-mask
// save / branch / restore triple; both loads are issued unconditionally from always-valid addresses
// (entry 0 when the position is outside the grid or the cell is empty) and the result is selected.
// Returns the run as (start, count).
// No validity test: the truncated cell coordinates are CLAMPED into the grid.  A position outside
// the grid is farther than delta from every scene point (choose_grid pads the box by r + 1 cells),
// so whatever candidates the boundary cell it is clamped to may hold fail the exact distance test;
// NaN converts to 0 (v_cvt_i32_f32) and fails `d2 <= eps` as well.  The outermost cell layer is
// empty in practice (the dilation reaches r cells), so such positions cost one word lookup.
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b_uniform, uint32_t c) {
  // full-rate 24-bit multiply-add (32/64-bit integer multiplies issue at a quarter of the rate);
  // operands: every axis has <= 1024 cells (grid_index.hip kMaxDim), i.e. <= 256 blocks
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}

// float -> int, truncating, with the hardware's total semantics (saturates, NaN -> 0); a C cast is
// undefined for those inputs and __float2int_rz adds a redundant v_trunc_f32
__device__ __forceinline__ int cvt_rz(float f) {
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}

// min(max(v, 0), hi) in one instruction (hi >= 0 is wave-uniform)
__device__ __forceinline__ int clamp0(int v, int hi_uniform) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi_uniform));
  return r;
}

// ---- instruction selection for the scoring loop ------------------------------------------------------
// Issue cost per wave64 instruction on gfx950, measured (tools/valu_rates.hip, profiles/r02_valu_rates.json):
// ~2.5 cycles for v_add/sub/mul_f32, v_add_u32, v_and/or/xor, v_lshrrev, v_mov with VGPR operands; ~4.2
// cycles for everything else that matters here (any VOP3: fma, mad24, med3, bfe, bcnt, lshl_or; cvt; cmp;
// DPP; the packed-f32 ops; ANY op with an SGPR source) -- and ~23 cycles for v_cndmask_b32 in its VOP2 form
// (implicit VCC), against 4.3 for the VOP3 form with the mask in an SGPR pair.  The helpers below keep
// selects in the VOP3 form; wave-uniform per-hypothesis counts stay on the scalar unit.

// mask bit of this lane ? a : b, as v_cndmask_b32_e64 with the mask in an SGPR pair
__device__ __forceinline__ float sel_mask(unsigned long long m, float a, float b) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ uint32_t sel_mask(unsigned long long m, uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}

// Cell number of one coordinate in the low mantissa bits: t = x * inv_h + (1.5 * 2^23 - k0) is rounded ONCE,
// to the integer round(x * inv_h) - k0 (+ 1.5 * 2^23) -- the exact nearest lattice cell, no float error at
// all -- and its bit pattern is 0x4B400000 + cell.  Positive floats order like their bit patterns and
// negative floats / NaN patterns are negative or huge integers, so ONE signed clamp of the bits does what
// truncation + two-sided clamp did: two instructions per axis instead of four (subtract, scale, truncate,
// clamp).  The clamp is to [0, n_max - 1] with n_max the LONGEST axis, the same two registers for all three
// axes (a VOP3 instruction reads one SGPR; per-axis bounds would cost three more registers, and this kernel
// sits on the 64-VGPR / 78-SGPR limits of 8 waves per SIMD): on a shorter axis a position beyond the grid
// can therefore land up to n_max - n cells outside it, and the word index is limited once more
// (v_min_u32).  Such a position ALIASES onto some word inside the array -- exact: whatever candidates the
// word lists fail the reference's float distance test (a position outside the grid is farther than
// delta from every scene point, choose_grid pads r + 1 cells; NaN fails every comparison).  Without
// any clamp the look-up was as fast, but far-out positions scattered over the whole array (+40 % HBM
// traffic, profiles/r02_pmc_noclamp.json).
constexpr int kMagicBits = 0x4B400000;   // bit pattern of 1.5 * 2^23
__device__ __forceinline__ uint32_t cell_bits(float x, float inv_h_vgpr, float c_uniform, int lo_vgpr, int hi_uniform) {
  float t;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(x), "v"(inv_h_vgpr), "s"(c_uniform));
  int b;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(b) : "v"(__float_as_int(t)), "v"(lo_vgpr), "s"(hi_uniform));
  return (uint32_t)b;
}

// Sparse form of the index (grid_index.hip): {occupancy bits, rank base} of block (bx, by, bz) from the
// open-addressing table; a block that is not in the table holds no candidate.  The table is at most one
// eighth full, so a look-up ends after 1.07 - 1.15 entries on average, hit or miss.
__device__ __forceinline__ void block_probe(const GridDesc& g, const uint4* __restrict__ tab, uint32_t bx,
                                            uint32_t by, uint32_t bz, unsigned* bits, unsigned* base) {
  const uint32_t key = block_key(g, bx, by, bz);
  uint32_t i = block_hash(g, key);
  uint4 e = tab[i];
  while (e.x != key && e.x != kBlockEmpty) {
    i = (i + 1u) & g.tab_mask;
    e = tab[i];
  }
  const bool hit = e.x == key;
  *bits = hit ? e.y : 0u;
  *base = hit ? e.z : 0u;
}

template <bool WAVE_SKIP = false>
__device__ __forceinline__ void cell_run(const GridDesc& g, const uint2* __restrict__ words,
                                         const uint2* __restrict__ occ_run, float x, float y,
                                         float z, uint32_t* s, uint32_t* n, bool live = true) {
  const float fx = (x - g.ox) * g.inv_h, fy = (y - g.oy) * g.inv_h, fz = (z - g.oz) * g.inv_h;
  const int cx = clamp0(cvt_rz(fx), g.nx - 1);
  const int cy = clamp0(cvt_rz(fy), g.ny - 1);
  const int cz = clamp0(cvt_rz(fz), g.nz - 1);
  // blocked numbering (pgp_internal.h grid_word / grid_bit): word = 4 x 4 x 2 block, bit = cell in it
  unsigned lo, base;
  if (g.sparse) {
    block_probe(g, reinterpret_cast<const uint4*>(words), (uint32_t)cx >> 2, (uint32_t)cy >> 2, (uint32_t)cz >> 1,
                &lo, &base);
  } else {
    const uint32_t brow = mad24((uint32_t)cz >> 1, (uint32_t)g.nby, (uint32_t)cy >> 2);
    const uint32_t wi = mad24(brow, (uint32_t)g.nbx, (uint32_t)cx >> 2);
    const unsigned long long wv = reinterpret_cast<const unsigned long long*>(words)[wi];
    lo = (unsigned)(wv & 0xFFFFFFFFull);
    base = (unsigned)(wv >> 32);
  }
  const int bit = ((cz & 1) << 4) | ((cy & 3) << 2) | (cx & 3);
  const bool occ = live & (((lo >> bit) & 1u) != 0u);
  if (WAVE_SKIP && __ballot(occ) == 0ull) {  // wave-uniform: no lane has a candidate run
    *s = 0u;
    *n = 0u;
    return;
  }
  const uint32_t k = occ ? base + __popc(lo & ((1u << bit) - 1u)) : 0u;
  const unsigned long long rv = reinterpret_cast<const unsigned long long*>(occ_run)[k];  // {start, count}
  const uint32_t st = (unsigned)(rv & 0xFFFFFFFFull), cnt = (unsigned)(rv >> 32);
  *s = occ ? st : 0u;
  *n = occ ? cnt : 0u;
}

// Nearest candidate with d2 <= sq_eps; ties -> lowest scene index (the reference's tie rule
// depends on kd-tree leaf order, kdtree.h:424; ties are measure-zero on real data).
__device__ __forceinline__ void nn_update(float4 p, float x, float y, float z, float* best, int* bid, bool* tie) {
  float d2 = sqdist(x, y, z, p);
  int id = __float_as_int(p.w);
  // two DIFFERENT scene points at the running minimum (a later, smaller distance makes the flag a false alarm:
  // harmless, the tree is only asked more often than needed)
  *tie = *tie || (d2 == *best && *bid >= 0 && id != *bid);
  if (d2 < *best || (d2 == *best && (*bid < 0 || id < *bid))) {
    *best = d2;
    *bid = id;
  }
}

// Four candidates per trip: the loads are independent, so one memory round trip serves four
// tests (the run is contiguous; indices past the end are clamped to the last element, which is
// harmless for "exists" and for arg-min with the lowest-index tie rule).
__device__ __forceinline__ int nearest_in_run(const float4* __restrict__ cand, uint32_t s, uint32_t e,
                                              float x, float y, float z, float sq_eps, bool* tied = nullptr) {
  float best = sq_eps;
  int bid = -1;
  bool tie = false;
  for (uint32_t j = s; j < e; j += 4) {
    uint32_t last = e - 1;
    float4 p0 = cand[j], p1 = cand[min(j + 1, last)], p2 = cand[min(j + 2, last)], p3 = cand[min(j + 3, last)];
    nn_update(p0, x, y, z, &best, &bid, &tie);
    nn_update(p1, x, y, z, &best, &bid, &tie);
    nn_update(p2, x, y, z, &best, &bid, &tie);
    nn_update(p3, x, y, z, &best, &bid, &tie);
  }
  if (tied) *tied = tie;
  return bid;
}

__device__ __forceinline__ bool any_in_run(const float4* __restrict__ cand, uint32_t s, uint32_t e,
                                           float x, float y, float z, float sq_eps) {
  bool hit = false;
  for (uint32_t j = s; j < e && !hit; j += 4) {
    uint32_t last = e - 1;
    float4 p0 = cand[j], p1 = cand[min(j + 1, last)], p2 = cand[min(j + 2, last)], p3 = cand[min(j + 3, last)];
    float d0 = sqdist(x, y, z, p0), d1 = sqdist(x, y, z, p1), d2 = sqdist(x, y, z, p2), d3 = sqdist(x, y, z, p3);
    hit = (d0 <= sq_eps) | (d1 <= sq_eps) | (d2 <= sq_eps) | (d3 <= sq_eps);
  }
  return hit;
}

__device__ __forceinline__ bool gate_ok(float dot, float lo, float hi) {
  // aligned branch: lo <= dot <= 1 ; anti-parallel branch (the fold of base.cc:1757): -1 <= dot <= hi
  // dot outside [-1,1] -> acos = NaN -> rejected (SURVEY hazard 4)
  return (dot >= lo && dot <= 1.0f) || (dot <= hi && dot >= -1.0f);
}

struct ScoreArgs {
  GridDesc g;
  const uint2* words;
  const uint2* occ_run;
  const float4* cand;
  const float4* Pnw;
  const float4* Q;
  const float4* Qn;
  int nQ;
  const float* T;
  int n_h, hpb, n_tiles, n_chunks;
  int n_big, hpb_tail;  // chunks [0, n_big) hold hpb hypotheses each, the rest hpb_tail (chunk_range)
  float sq_eps, gate_lo, gate_hi;
  float cell_c[3];      // 1.5 * 2^23 - k0 per axis (cell_bits)
  uint32_t last_word;   // number of occupancy words - 1
  int cell_hi;          // kMagicBits + (cells of the longest axis) - 1
  uint2* partial;       // [n_tiles][n_h] {inlier count, bits of the weight sum (weighted only)}: ONE 8-byte
                        // load per (tile, hypothesis) in finalize_scores
  // pgp_set_exact_ties: the reference's kd-tree over the scene (kd_ties.hip), asked only when two different
  // candidates share the minimal distance; null = ties go to the lowest scene index
  const int4* kd_nodes;
  const float4* kd_pts;
};

// Fused finalisation (PGP_FUSED builds, launch_score): the (tile, hypothesis) partials are ADDED to per-hypothesis
// accumulator words whose upper 16 bits count the tiles that have arrived -- data and ticket in ONE returning atomic --,
// and whoever arrives last turns the total into the score; no second launch.  The LAST parameter of the flat kernels,
// read through the kernel-argument pointer at the very end of a workgroup only (fuse_args): as members of ScoreArgs
// these fields were fetched at the top of the kernel and sat in scalar registers through the whole scoring loop.
struct FuseArgs {
  unsigned long long* acc;        // [2][stride]: {arrivals << 48 | inlier count}, {arrivals << 48 | fixed-point weight sum}
  float* scores;
  int* counts;                    // nullable
  unsigned long long* best_key;
  unsigned long long* runner_key;
  unsigned int* done;
  int* best;
  double fx_inv;                  // 2^-shift
  float fx_scale;                 // 2^shift of the fixed-point weight sums
  int acc_stride;
};
[[maybe_unused]] constexpr size_t kFuseArgsOffset = ((sizeof(ScoreArgs) + 7) & ~(size_t)7) + 5 * sizeof(void*);
[[maybe_unused]] __device__ __forceinline__ FuseArgs fuse_args() {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned char* kp = (const unsigned char*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));   // the loads below stay below
  return *reinterpret_cast<const FuseArgs*>(kp + kFuseArgsOffset);
#else
  return FuseArgs{};
#endif
}

// KdTree::doQueryRestrictedClosestIndex (kdtree.h:394-459) on the uploaded tree: the same descent (the query's side
// of a split first, the other side only while its plane is closer than the best so far, strictly), the same
// inclusive test inside a leaf, so that among equal distances the point the reference visits last is returned.
__device__ __attribute__((noinline)) int kd_restricted_nn(const int4* __restrict__ nodes, const float4* __restrict__ pts,
                                                          float x, float y, float z, float sq_eps) {
  // The reference keeps a stack of {node, squared plane distance}: the near child inherits its parent's entry (which has just
  // passed the test, with the same best), so it is ALWAYS visited; the far child's entry is the squared offset to the parent's
  // plane, tested against the best when the near subtree is done.  Both are functions of the parent alone, so the same order of
  // visits is walked here WITHOUT a stack, climbing through parent links (node.w >> 1) -- a 512-byte stack per lane was the only
  // scratch memory of every kernel that can meet a tie (finalize_scores among them, launched with every scoring call).
  int cl_id = -1;
  float cl_dist = sq_eps;
  if (!(0.f < cl_dist)) return cl_id;   // the root's own entry {0, 0.f} fails the test
  int node = 0, child = 0;
  bool down = true;
  for (;;) {
    const int4 nd = nodes[node];
    if (down) {
      if (nd.w & 1) {   // leaf {start, size}
        for (int i = nd.x; i < nd.x + nd.y; ++i) {
          const float4 p = pts[i];
          const float d2 = sqdist(x, y, z, p);
          if (d2 <= cl_dist) {
            cl_dist = d2;
            cl_id = __float_as_int(p.w);
          }
        }
        child = node;
        node = nd.w >> 1;
        down = false;
      } else {          // inner {bits(split), first child, dim}: the query's side first
        const float q = nd.z == 0 ? x : (nd.z == 1 ? y : z);
        const float new_off = __fsub_rn(q, __int_as_float(nd.x));
        node = new_off < 0.f ? nd.y : nd.y + 1;
      }
    } else {            // back in an inner node, from `child`
      const float q = nd.z == 0 ? x : (nd.z == 1 ? y : z);
      const float new_off = __fsub_rn(q, __int_as_float(nd.x));
      const int near = new_off < 0.f ? nd.y : nd.y + 1;
      if (child == near && __fmul_rn(new_off, new_off) < cl_dist) {
        node = new_off < 0.f ? nd.y + 1 : nd.y;
        down = true;
        continue;
      }
      if (node == 0) break;
      child = node;
      node = nd.w >> 1;
    }
  }
  return cl_id;
}

// nearest candidate of a run by the rule in force: lowest scene index on exact ties, or the reference's
// (kd_restricted_nn) when its tree is there
__device__ __forceinline__ int nearest_by_rule(const ScoreArgs& a, uint32_t s, uint32_t e, float x, float y, float z) {
  bool tied = false;
  int id = nearest_in_run(a.cand, s, e, x, y, z, a.sq_eps, &tied);
  if (tied && a.kd_nodes) id = kd_restricted_nn(a.kd_nodes, a.kd_pts, x, y, z, a.sq_eps);
  return id;
}

// Hypothesis range of a chunk.  The last chunks are smaller: workgroups are dispatched in block
// order, so the batch ends on short workgroups and the idle tail of the launch shrinks.
__device__ __forceinline__ void chunk_range(const ScoreArgs& a, int chunk, int* h0, int* h1) {
  const int big = min(chunk, a.n_big), small = chunk - big;
  *h0 = big * a.hpb + small * a.hpb_tail;
  *h1 = min(*h0 + (chunk < a.n_big ? a.hpb : a.hpb_tail), a.n_h);
}

// U hypotheses are in flight per lane: their word loads, then their offset loads, then their
// candidate trips are issued back to back, so a wave keeps U independent dependency chains in
// the memory system instead of one (the kernel is latency-bound: 79 % of wave cycles were
// s_waitcnt at U = 1, profiles/r01_a_*).
template <int MODE, int U>
__global__ __launch_bounds__(kTile) void score_hypotheses(ScoreArgs a) {
  __shared__ int s_cnt[kTile / 64][kMaxHpb];
  __shared__ float s_sum[kTile / 64][kMaxHpb];

  // XCD-aware decode: consecutive blocks are dealt round-robin over the 8 XCDs, so blocks that
  // agree in blockIdx%8 share an L2; give each XCD its own hypothesis chunks.
  const int L = blockIdx.x;
  const int xcd = L & 7, seq = L >> 3;
  const int chunk = (seq / a.n_tiles) * 8 + xcd;
  const int tile = seq % a.n_tiles;
  if (chunk >= a.n_chunks) return;  // whole block exits together (padding blocks)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = tile * kTile + threadIdx.x;
  const bool live = qi < a.nQ;
  float4 q = live ? a.Q[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 qn = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == PGP_MODE_WEIGHTED && live) qn = a.Qn[qi];

  int h0, h1;
  chunk_range(a, chunk, &h0, &h1);
  for (int hb = h0; hb < h1; hb += U) {
    Xf m[U];
    float x[U], y[U], z[U];
    uint32_t s[U], e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int h = min(hb + u, h1 - 1);  // tail: recompute the last one, result discarded
      m[u] = load_xf(a.T, h);
      x[u] = xf_row(m[u].m00, m[u].m01, m[u].m02, m[u].m03, q.x, q.y, q.z);
      y[u] = xf_row(m[u].m10, m[u].m11, m[u].m12, m[u].m13, q.x, q.y, q.z);
      z[u] = xf_row(m[u].m20, m[u].m21, m[u].m22, m[u].m23, q.x, q.y, q.z);
      cell_run(a.g, a.words, a.occ_run, x[u], y[u], z[u], &s[u], &e[u], live);
      e[u] += s[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool hit = false;
      float wsum = 0.f;
      if (MODE == PGP_MODE_PLAIN) {
        hit = any_in_run(a.cand, s[u], e[u], x[u], y[u], z[u], a.sq_eps);
      } else {
        int id = nearest_by_rule(a, s[u], e[u], x[u], y[u], z[u]);
        if (id >= 0) {
          float nx = rot_row(m[u].m00, m[u].m01, m[u].m02, qn.x, qn.y, qn.z);
          float ny = rot_row(m[u].m10, m[u].m11, m[u].m12, qn.x, qn.y, qn.z);
          float nz = rot_row(m[u].m20, m[u].m21, m[u].m22, qn.x, qn.y, qn.z);
          float4 pn = a.Pnw[id];
          float dot = __fadd_rn(__fmul_rn(pn.x, nx), __fadd_rn(__fmul_rn(pn.y, ny), __fmul_rn(pn.z, nz)));
          if (gate_ok(dot, a.gate_lo, a.gate_hi)) {
            hit = true;
            wsum = pn.w;
          }
        }
      }
      unsigned long long mask = __ballot(hit);
      if (MODE == PGP_MODE_WEIGHTED && mask) wsum = wave_sum(wsum);  // fixed association, same every run
      if (lane == 0 && hb + u < h1) {
        s_cnt[wave][hb + u - h0] = __popcll(mask);
        if (MODE == PGP_MODE_WEIGHTED) s_sum[wave][hb + u - h0] = wsum;
      }
    }
  }
  __syncthreads();
  const int hh = threadIdx.x;
  if (hh < h1 - h0) {
    int c = 0;
#pragma unroll
    for (int w = 0; w < kTile / 64; ++w) c += s_cnt[w][hh];
    float f = 0.f;
    if (MODE == PGP_MODE_WEIGHTED) {
#pragma unroll
      for (int w = 0; w < kTile / 64; ++w) f += s_sum[w][hh];
    }
    a.partial[(size_t)tile * a.n_h + h0 + hh] = make_uint2((uint32_t)c, __float_as_uint(f));
  }
}

#if defined(PGP_CAND8) && PGP_CAND8
// ---- the candidate FORMAT experiment (VERDICT r4 task 4; build knob PGP_CAND8, `make variantf FILE=lcp_score`) ----
// 8-byte candidates instead of the 16-byte float4 {x, y, z, id}: three fp16 offsets from the CENTRE of the cell whose list the
// candidate sits in + a 16-bit scene index -- half the candidate footprint (22.6 -> 11.3 MB at C2), so that finer cells fit the
// footprint the 0.85-delta grid has today.  The query is expressed relative to the same centre (two VALU per axis and trip),
// a candidate is decoded with three conversions and two shifts.  fp16 offsets carry ~4 um of error at these cell sizes:
// inlier decisions next to the radius can differ from the exact kernel's, so this is a TIMING AND COUNTER experiment, not a
// product path (a product would keep this test as a conservative reject and re-test survivors on the float4 array).
// Scenes of at most 65 535 points, dense block array, flat kernels only.
__device__ __forceinline__ float4 cand8_fetch(const float4* __restrict__ cand, uint32_t idx) {
  const uint2 raw = reinterpret_cast<const uint2*>(cand)[idx];
  const float hx = __half2float(__ushort_as_half((unsigned short)(raw.x & 0xFFFFu)));
  const float hy = __half2float(__ushort_as_half((unsigned short)(raw.x >> 16)));
  const float hz = __half2float(__ushort_as_half((unsigned short)(raw.y & 0xFFFFu)));
  return make_float4(hx, hy, hz, __int_as_float((int)(raw.y >> 16)));
}

__global__ __launch_bounds__(256) void pack_cand8(GridDesc g, const uint2* __restrict__ words, const uint2* __restrict__ occ_run,
                                                  const float4* __restrict__ cand, uint2* __restrict__ out, uint32_t n_words) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t wi = tid >> 5, bit = tid & 31u;
  if (wi >= n_words) return;
  const uint2 w = words[wi];
  if (!((w.x >> bit) & 1u)) return;
  const uint2 run = occ_run[w.y + __popc(w.x & ((1u << bit) - 1u))];
  const uint32_t bx = wi % (uint32_t)g.nbx, by = (wi / (uint32_t)g.nbx) % (uint32_t)g.nby, bz = wi / ((uint32_t)g.nbx * (uint32_t)g.nby);
  // lattice numbers of the cell: round(p * inv_h) for the points inside it (grid_index.hip); the centre is L * h
  const float Lx = (float)((int)(4u * bx + (bit & 3u)) + g.k0x), Ly = (float)((int)(4u * by + ((bit >> 2) & 3u)) + g.k0y),
              Lz = (float)((int)(2u * bz + (bit >> 4)) + g.k0z);
  const float cx = Lx * g.h, cy = Ly * g.h, cz = Lz * g.h;
  for (uint32_t k = 0; k < run.y; ++k) {
    const float4 p = cand[run.x + k];
    const unsigned hx = __half_as_ushort(__float2half_rn(p.x - cx)), hy = __half_as_ushort(__float2half_rn(p.y - cy)),
                   hz = __half_as_ushort(__float2half_rn(p.z - cz));
    out[run.x + k] = make_uint2(hx | (hy << 16), hz | ((unsigned)__float_as_int(p.w) << 16));
  }
}
#endif

// ---- wave-flattened candidate phase --------------------------------------------------------------
// What binds the per-lane kernel above (profiles/r01_c_*): the vector L1 is busy for the whole
// kernel (TCP_GATE_EN ~ kernel duration) at ~25 busy cycles per vector-memory WAVE-INSTRUCTION,
// almost independent of how many lanes or lines the instruction touches; only ~22 % of the lanes
// own a candidate run (mean 9.4 entries, p90 23), so the per-lane walk spends 5-6 load
// instructions per hypothesis on ~132 candidates.  Two experiments that kept the per-lane walk
// (software pipelining across hypotheses; 16-lane groups per run) issued MORE such instructions
// and were slower (157 and 203 us vs 125 us).  Here the runs of a wave are flattened: the lanes
// that own a run are compacted (in lane order) into an LDS table, the slots of the concatenated
// runs are handed out by a DPP prefix scan of the run lengths, each run's first slot is marked in
// an LDS bit array, and lane (w mod 64) tests candidate slot w after resolving its owner with a
// ballot and a popcount over those marks (flat_batch) -- so a load instruction serves 64 useful
// candidates and a wave-iteration needs ceil(W/64) ~ 2 of them.  Per-owner results are combined
// through LDS (store / 64-bit min) and read back by the owner.
// Arithmetic per (query, candidate) pair is unchanged; results are identical.
constexpr int kFlatCap = 1024;  // candidate slots per wave-iteration served by the flat path

// Inclusive prefix sum over the 64 lanes on the DPP path: a shift-and-add scan inside each 16-lane
// row (zeros are shifted in at the row start), then the row totals are carried across rows
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_scan_step(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  v = dpp_scan_step<0x111, 0xF>(v);  // row_shr:1
  v = dpp_scan_step<0x112, 0xF>(v);  // row_shr:2
  v = dpp_scan_step<0x114, 0xF>(v);  // row_shr:4
  v = dpp_scan_step<0x118, 0xF>(v);  // row_shr:8
  v = dpp_scan_step<0x142, 0xA>(v);  // row_bcast:15 -> rows 1, 3
  v = dpp_scan_step<0x143, 0xC>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}

// NC chunks of 64 candidate slots of the wave's concatenated runs (see score_hypotheses_flat).
// Slot w belongs to the owner with the greatest run start <= w.  Run starts are marked as bits
// (marks[w >> 6] bit w & 63) and owners are numbered in start order, so the owner of slot w is
//   (#owners starting before the chunk) + popcount(marks word of the chunk & bits <= lane) - 1:
// one uniform LDS read, a ballot and four VALU per chunk -- no per-slot owner table to fill.
template <int MODE, int NC, bool TIES = false>
__device__ __forceinline__ void flat_batch(const ScoreArgs& a, const float4* __restrict__ cand,
                                           const float4* ent, unsigned long long* res,
                                           const unsigned long long* marks, uint32_t start_key, uint32_t W,
                                           uint32_t w0, int lane, uint32_t le_lo, uint32_t le_hi,
                                           unsigned long long* res_hi = nullptr) {
  // a lane past the last slot resolves to the LAST owner and repeats its last candidate (valid owner, valid
  // candidate, the SAME cache line); it is masked out of the result: no exec-mask branches in the batch.
  // (Letting those lanes read on past the run saved the clamp and cost 33 % more HBM traffic: a wave-
  // iteration with 6 slots then touched a whole KB of lines nobody needs, profiles/r02_pmc_noclamp.json.)
  int o[NC];
  uint32_t we[NC];
  float4 en[NC], p[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const uint32_t wb = w0 + 64 * c;                       // wave-uniform
    const unsigned long long word = marks[wb >> 6];        // same address in every lane: one broadcast read
    const int before = __popcll(__ballot(start_key < wb)); // owners whose run starts before this chunk
    o[c] = before - 1 + __popc((uint32_t)word & le_lo) + __popc((uint32_t)(word >> 32) & le_hi);
    we[c] = min(wb + lane, W - 1u);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) en[c] = ent[o[c]];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#if defined(PGP_ABLATE) && PGP_ABLATE == 1
    p[c] = cand[lane];            // timing experiment: one contiguous, always-cached 1 KB instead of the gather
#elif defined(PGP_CAND8) && PGP_CAND8
    p[c] = cand8_fetch(cand, __float_as_uint(en[c].w) + we[c]);   // the candidate FORMAT experiment (see cand8_fetch)
#else
    p[c] = cand[__float_as_uint(en[c].w) + we[c]];
#endif
  }
  if (MODE == PGP_MODE_WEIGHTED) {
    // keep the 16-byte loads whole: the id (.w) is only used by lanes with an in-range candidate, and
    // the compiler otherwise narrows the load to 12 bytes and sinks a dependent 4-byte load of the id
    // into that branch -- one more L1/L2 round trip on the critical chain of every batch with a hit
#pragma unroll
    for (int c = 0; c < NC; ++c) asm volatile("" ::"v"(p[c].w));
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float d2 = sqdist(en[c].x, en[c].y, en[c].z, p[c]);
#if defined(PGP_ABLATE) && PGP_ABLATE == 3
    if (d2 == -1.0f) {            // timing experiment: no result traffic
#else
    if (w0 + 64 * c + lane < W && d2 <= a.sq_eps) {
#endif
      if (MODE == PGP_MODE_PLAIN) {
        res[o[c]] = 1ull;  // benign race: every writer stores the same value
      } else if (!TIES) {
        atomicMin(&res[o[c]], ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(p[c].w));
      } else {
        // pgp_set_exact_ties: a second minimum with the scene index complemented keeps the HIGHEST index at the
        // minimal distance; when it differs from the lowest, two scene points tie and the owner asks the reference's
        // tree (kd_restricted_nn).  (A returning atomic + a flag store instead of the second minimum costs the same:
        // weighted step 103 -> 124 us at C2 either way, tools/exact_ties_cost.py.)
        const unsigned long long dk = (unsigned long long)__float_as_uint(d2) << 32;
        const unsigned id = (unsigned)__float_as_int(p[c].w);
        atomicMin(&res[o[c]], dk | id);
        atomicMin(&res_hi[o[c]], dk | (unsigned)~id);
      }
    }
  }
}

// Hypotheses whose per-lane registered weights are parked in LDS before ONE reduction of the group
// (weighted mode): a 64-lane DPP tree per hypothesis cost 12 VALU per wave-trip; four rows of 64 floats
// are summed by 16 lanes each (4 sequential adds, then 4 DPP steps) for 9 VALU per FOUR trips.  (Eight rows
// would halve that again, but their 8 KB push the workgroup past 20 KB of LDS = 7 workgroups per CU.)
constexpr int kSumGroup = 4;

// Build knobs of the flat kernel (A/B through `make variant`, tools/ab_step.sh):
//   PGP_SHORTRUN  longest run a wave's owner lanes test by themselves (0: every trip takes the flat path)
//   PGP_PLAIN_NC  widest batch in plain mode, in chunks of 64 slots.  Four chunks used all 64 VGPRs: with the
//                 short-trip path they spill (plain 84 -> 104 us per step); three leave 60.
#ifndef PGP_SHORTRUN
#define PGP_SHORTRUN 1
#endif
#ifndef PGP_PLAIN_NC
#define PGP_PLAIN_NC 3
#endif

// The read-only arrays are separate __restrict__ kernel parameters: inside the by-value struct
// hipcc could not prove them invariant next to the LDS atomics and fetched the wave-uniform 4x4
// with FOUR vector loads per hypothesis instead of scalar loads.
template <int MODE, int NCW, bool SPARSE, bool TIES = false>   // NCW: widest batch (chunks of 64 slots) in weighted mode
__device__ __forceinline__ void score_flat_body(const ScoreArgs& a, const float* __restrict__ Tm,
                                                const uint2* __restrict__ words,
                                                const uint2* __restrict__ occ_run,
                                                const float4* __restrict__ cand,
                                                const float4* __restrict__ Pnw) {
  constexpr bool kW = MODE == PGP_MODE_WEIGHTED;
  __shared__ int s_cnt[kTile / 64][kMaxHpb];
  __shared__ float s_sum[kTile / 64][kMaxHpb];
  __shared__ float4 s_ent[kTile / 64][64];                 // {x', y', z', bits(run start - prefix)}
  __shared__ unsigned long long s_res[kTile / 64][64];     // plain: 0/1 ; weighted: min key
  // run-start bits of the wave's concatenated runs; 4 spare words: the last batch may look one
  // to three chunks past the end (they stay zero)
  __shared__ unsigned long long s_marks[kTile / 64][kFlatCap / 64 + 4];
  __shared__ float s_w[kW ? kTile / 64 : 1][kW ? kSumGroup : 1][64];   // registered weight per (hypothesis, lane)
  __shared__ float4 s_qn[kW ? kTile : 1];                                 // model normals of the tile
  // pgp_set_exact_ties: min key with the scene index complemented (= the highest index at the minimal distance)
  __shared__ unsigned long long s_res_hi[TIES ? kTile / 64 : 1][TIES ? 64 : 1];
  unsigned long long* res_hi = s_res_hi[TIES ? (threadIdx.x >> 6) : 0];

  const int L = blockIdx.x;
  const int xcd = L & 7, seq = L >> 3;
  const int chunk = (seq / a.n_tiles) * 8 + xcd;
  const int tile = seq % a.n_tiles;
  if (chunk >= a.n_chunks) return;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = tile * kTile + threadIdx.x;
  const bool live = qi < a.nQ;
  // a lane past the end of the model carries a NaN point: it lands in some cell and can never pass
  // `d2 <= eps`, so the loop needs no liveness test
  const float qnan = __int_as_float(0x7FC00000);
  float4 q = live ? a.Q[qi] : make_float4(qnan, qnan, qnan, 0.f);
  // weighted mode: the model normal is needed once per trip with a neighbour; it waits in LDS (a linear,
  // conflict-free 16-byte read) instead of holding three registers through the candidate phase
  if (kW) s_qn[threadIdx.x] = live ? a.Qn[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4* ent = s_ent[wave];
  unsigned long long* res = s_res[wave];
  unsigned long long* marks = s_marks[wave];
  if (lane < kFlatCap / 64 + 4) marks[lane] = 0ull;
  const unsigned long long le_mask = (2ull << lane) - 1ull;  // bits 0..lane (lane 63: all ones)
  const uint32_t le_lo = (uint32_t)le_mask, le_hi = (uint32_t)(le_mask >> 32);
  const unsigned long long* words64 = reinterpret_cast<const unsigned long long*>(words);
  const unsigned long long* run64 = reinterpret_cast<const unsigned long long*>(occ_run);
  // wave-uniform constants of cell_bits that must sit in VGPRs (one SGPR per VOP3 instruction)
  const float inv_h_v = a.g.inv_h;
  const int cell_lo_v = kMagicBits;

  int h0, h1;
  chunk_range(a, chunk, &h0, &h1);
  // Per group of kSumGroup hypotheses, on the scalar unit: the wave's inlier counts, one byte each (a count
  // is <= 64), and which rows of s_w were written (a wave-iteration in which no lane has a neighbour writes
  // nothing).  Scalar and vector instructions cost the SIMD the same ~4 issue cycles each here and add up
  // (tools/valu_rates.hip; a wave-iteration with every cell empty takes (VALU + SALU) x 4 cycles), so the
  // per-trip bookkeeping is kept to a handful of scalar instructions and off the empty path.
  unsigned long long cnt_pack = 0ull;
  uint32_t wrote = 0u;
#if defined(PGP_ABLATE) && PGP_ABLATE == 10
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

  // one hypothesis (slot hs = h - h0 of the chunk) under its 4x4 `m`
  auto trip = [&](const int hs, const Xf& m) {
    const int gs = hs & (kSumGroup - 1);   // slot in the group
    PGP_STAMP(t_a);
#if defined(PGP_ABLATE) && PGP_ABLATE == 7
    // timing experiment: no transform (and the 8 KB of words of experiment 4)
    const float x = q.x + m.m03, y = q.y + m.m13, z = q.z + m.m23;
#else
    const float x = xf_row(m.m00, m.m01, m.m02, m.m03, q.x, q.y, q.z);
    const float y = xf_row(m.m10, m.m11, m.m12, m.m13, q.x, q.y, q.z);
    const float z = xf_row(m.m20, m.m21, m.m22, m.m23, q.x, q.y, q.z);
#endif
    // cell -> occupancy word (blocked numbering, pgp_internal.h grid_word / grid_bit)
    const uint32_t bx = cell_bits(x, inv_h_v, a.cell_c[0], cell_lo_v, a.cell_hi);
    const uint32_t by = cell_bits(y, inv_h_v, a.cell_c[1], cell_lo_v, a.cell_hi);
    const uint32_t bz = cell_bits(z, inv_h_v, a.cell_c[2], cell_lo_v, a.cell_hi);
    uint32_t lo, base;
    if (SPARSE) {
      // sparse form: the block's entry of the hashed table (grid_index.hip).  A position outside the grid is
      // clamped to the LARGEST axis only (cell_hi), so its key may be any 31-bit number: whatever block that
      // names holds candidates far from the position, and the distance test below rejects them.
      const uint32_t key = (__builtin_amdgcn_ubfe(bx, 2, 12) | (__builtin_amdgcn_ubfe(by, 2, 12) << a.g.key_sy) |
                            (__builtin_amdgcn_ubfe(bz, 1, 13) << a.g.key_sz)) & 0x7FFFFFFFu;
      const uint4* tab = reinterpret_cast<const uint4*>(words);
      // two entries in flight at once: a second DEPENDENT load (the wave waits for its slowest lane) is then
      // needed by ~1 % of the look-ups instead of ~10 % (table at most one eighth full): 32.7 -> 38.5 M hyp/s at
      // C2 forced sparse with 4 entries per block, 40.9 M with 8 (tools/sparse_index_time.py)
      uint32_t i = block_hash(a.g, key);
      uint4 e = tab[i];
      const uint4 e1 = tab[(i + 1u) & a.g.tab_mask];
      if (e.x != key && e.x != kBlockEmpty) {
        e = e1;
        ++i;
        while (e.x != key && e.x != kBlockEmpty) {
          i = (i + 1u) & a.g.tab_mask;
          e = tab[i];
        }
      }
      const bool hit = e.x == key;
      lo = hit ? e.y : 0u;
      base = e.z;
    } else {
    const uint32_t brow = mad24(__builtin_amdgcn_ubfe(bz, 1, 9), (uint32_t)a.g.nby, __builtin_amdgcn_ubfe(by, 2, 8));
    const uint32_t wi = min(mad24(brow, (uint32_t)a.g.nbx, __builtin_amdgcn_ubfe(bx, 2, 8)), a.last_word);
#if defined(PGP_ABLATE) && (PGP_ABLATE == 4 || PGP_ABLATE == 7)
    const unsigned long long wv = words64[wi & 1023u];   // timing experiment: occupancy words from 8 KB
#elif defined(PGP_ABLATE) && PGP_ABLATE == 6
    const unsigned long long wv = (unsigned long long)(wi >> 31);   // timing experiment: no word load (all empty)
#elif defined(PGP_ABLATE) && PGP_ABLATE == 8
    const unsigned long long wv = words64[wi] & ~0xFFFFFFFFull;    // timing experiment: real word load, all empty
#else
    const unsigned long long wv = words64[wi];
#endif
    lo = (uint32_t)wv;
    base = (uint32_t)(wv >> 32);
    }
    const uint32_t bit = (bx & 3u) | ((by & 3u) << 2) | ((bz & 1u) << 4);
    // lanes whose cell holds a candidate run; the second look-up runs for those lanes only
    const bool occ = __builtin_amdgcn_ubfe(lo, bit, 1) != 0u;
    const unsigned long long am = __ballot(occ);
    PGP_STAMP(t_b);   // the occupancy word has arrived
    PGP_PHASE(0, t_a, t_b);
    if (am == 0ull) return;   // 35 % of the wave-iterations at C2 end here: count 0, no row of s_w
    uint32_t s = 0u, len = 0u;
    if (occ) {
      const uint32_t k = base + __popc(__builtin_amdgcn_ubfe(lo, 0, bit));
      const unsigned long long rv = run64[k];  // {start, count}, count >= 1
      s = (uint32_t)rv;
      len = (uint32_t)(rv >> 32);
    }
#if defined(PGP_CAND8) && PGP_CAND8
    // the query relative to the centre of its cell: L = round(x * inv_h) sits in the clamped lattice word, centre = L * h
    const float h_v = a.g.h;
    const float xq = __fmaf_rn(-__fsub_rn(__uint_as_float(bx), a.cell_c[0]), h_v, x), yq = __fmaf_rn(-__fsub_rn(__uint_as_float(by), a.cell_c[1]), h_v, y),
                zq = __fmaf_rn(-__fsub_rn(__uint_as_float(bz), a.cell_c[2]), h_v, z);
#else
    const float xq = x, yq = y, zq = z;
#endif
    uint32_t rlo = kW ? 0xFFFFFFFFu : 0u;
    // Short trips (round 3, profiles/r03_ab/shortrun.log): when every run of the wave holds at most
    // PGP_SHORTRUN candidates, each owner lane tests its own -- no slot scan, no owner table, no LDS round
    // trips.  With 1: weighted 101.7 -> 99.2 us per step, plain 84.5 -> 83.6 (with three-chunk batches, below);
    // 2: 100.4 / 84.2; 3: 101.8; 4: 109 -- testing N candidates in every owner lane soon costs more than the
    // trips it takes off the flat path.  (Not in the weighted kernel over the sparse table: it would spill.)
    constexpr int kShort = (SPARSE && kW) ? 0 : PGP_SHORTRUN;
    const bool short_trip = kShort > 0 && __ballot(len > (uint32_t)kShort) == 0ull;
    if (short_trip) {
      if (occ) {
        const uint32_t last = s + len - 1u;
        float4 pc[kShort > 0 ? kShort : 1];
#pragma unroll
#if defined(PGP_CAND8) && PGP_CAND8
        for (int k = 0; k < kShort; ++k) pc[k] = cand8_fetch(cand, min(s + (uint32_t)k, last));
#else
        for (int k = 0; k < kShort; ++k) pc[k] = cand[min(s + (uint32_t)k, last)];
#endif
        unsigned long long best = ~0ull;
#pragma unroll
        for (int k = 0; k < kShort; ++k) {
          const float d2 = sqdist(xq, yq, zq, pc[k]);
          if (!kW) {
            if (d2 <= a.sq_eps) rlo = 1u;
          } else {
            const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(pc[k].w);
            if (d2 <= a.sq_eps && key < best) best = key;
          }
        }
        if (kW) rlo = (uint32_t)best;
      }
    } else {
    // slot allocation in the concatenated run of the wave, in LANE order: an inclusive DPP scan of
    // the run lengths (no LDS); the total lands in an SGPR, so everything below branches scalar.
    const uint32_t incl = wave_inclusive_scan(len);
    const uint32_t W = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);   // >= 1
    const uint32_t pre = incl - len;
    PGP_STAMP(t_c);   // run descriptors + scan
    PGP_PHASE(1, t_b, t_c);
    // result of the candidate phase per owner lane (rlo): plain 0 / 1, weighted the scene id of the nearest
    // candidate within delta (all ones = -1: none)
    if (W <= (uint32_t)kFlatCap) {
      // owners are numbered in lane order = start order: # owning lanes below this one (v_mbcnt)
      const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
      if (occ) {
        // store (start - prefix) so that slot w maps to candidate (start - prefix) + w
        ent[r] = make_float4(xq, yq, zq, __uint_as_float(s - pre));
        res[r] = kW ? ~0ull : 0ull;
        if (TIES) res_hi[r] = ~0ull;
        atomicOr(&marks[pre >> 6], 1ull << (pre & 63u));
      }
      __builtin_amdgcn_wave_barrier();
      PGP_STAMP(t_d);   // owner table written
      PGP_PHASE(2, t_c, t_d);
      const uint32_t start_key = sel_mask(am, pre, 0xFFFFFFFFu);
      // batches of NC chunks of 64 slots: owner resolution, then ALL candidate loads, then tests.
      // Half of the non-empty wave-iterations need a single chunk (median W = 6 at C2), so the
      // batch width follows what is left instead of always issuing four chunks.
      for (uint32_t w0 = 0; w0 < W;) {
        const uint32_t left = W - w0;
#if PGP_PLAIN_NC >= 4
        if (!kW && left > 128) {
          flat_batch<MODE, 4, TIES>(a, cand, ent, res, marks, start_key, W, w0, lane, le_lo, le_hi, res_hi);
          w0 += 256;
        } else
#endif
        if ((kW || PGP_PLAIN_NC == 3) && NCW >= 3 && left > 128) {
          flat_batch<MODE, 3, TIES>(a, cand, ent, res, marks, start_key, W, w0, lane, le_lo, le_hi, res_hi);
          w0 += 192;
        } else if (left > 64) {
          flat_batch<MODE, 2, TIES>(a, cand, ent, res, marks, start_key, W, w0, lane, le_lo, le_hi, res_hi);
          w0 += 128;
        } else {
          flat_batch<MODE, 1, TIES>(a, cand, ent, res, marks, start_key, W, w0, lane, le_lo, le_hi, res_hi);
          w0 += 64;
        }
      }
      __builtin_amdgcn_wave_barrier();
      PGP_STAMP(t_e);   // all batches done
      PGP_PHASE(3, t_d, t_e);
      if ((uint32_t)lane < ((W + 63u) >> 6)) {   // clear the bits for the next iteration
        // (the zero is made HERE: as an ordinary constant the compiler parked a 64-bit zero in two VGPRs for the whole
        // scoring loop and, at the register ceiling, spilled it to scratch memory and reloaded it on every trip)
        unsigned long long zero = 0ull;
        asm volatile("" : "+v"(zero));
        marks[lane] = zero;
      }
      // the low word of the owner's result: plain 0 / 1; weighted the id of the minimum key, -1 if none
      if (occ) rlo = reinterpret_cast<const uint32_t*>(res)[2 * r];
      if (TIES && kW) {   // tied candidates at the minimum: the reference's tree decides (rare)
#if defined(PGP_ABLATE) && PGP_ABLATE == 12   // ties seen (both minima kept) but never resolved: the cost of SEEING them alone
        if (occ && rlo != 0xFFFFFFFFu && reinterpret_cast<const uint32_t*>(res_hi)[2 * r] != ~rlo) rlo = ~reinterpret_cast<const uint32_t*>(res_hi)[2 * r];
#else
        if (occ && rlo != 0xFFFFFFFFu && reinterpret_cast<const uint32_t*>(res_hi)[2 * r] != ~rlo)
          rlo = (uint32_t)kd_restricted_nn(a.kd_nodes, a.kd_pts, x, y, z, a.sq_eps);
#endif
      }
    } else {
      // oversized wave-iteration (very dense scene): per-lane walk
      if (!kW) rlo = any_in_run(a.cand, s, s + len, x, y, z, a.sq_eps) ? 1u : 0u;
      else rlo = TIES ? (uint32_t)nearest_by_rule(a, s, s + len, x, y, z) : (uint32_t)nearest_in_run(a.cand, s, s + len, x, y, z, a.sq_eps);
    }
    }
    PGP_STAMP(t_f);   // results read back
    unsigned long long hm;   // lanes whose model point registers under this hypothesis
    if (!kW) {
      hm = __ballot(rlo != 0u);
    } else {
      const int nn_id = (int)rlo;
      if (__ballot(nn_id >= 0) == 0ull) return;   // no lane has a neighbour within delta
      float dot = 2.0f, pw = 0.0f;   // dot = 2 fails the gate
#if defined(PGP_ABLATE) && PGP_ABLATE == 5
      if (nn_id == -2) {        // timing experiment: no normal gate, no weight gather
#else
      if (nn_id >= 0) {
#endif
        const float4 pn = Pnw[nn_id];
        const float4 qn = s_qn[threadIdx.x];
        const float nx = rot_row(m.m00, m.m01, m.m02, qn.x, qn.y, qn.z);
        const float ny = rot_row(m.m10, m.m11, m.m12, qn.x, qn.y, qn.z);
        const float nz = rot_row(m.m20, m.m21, m.m22, qn.x, qn.y, qn.z);
        dot = __fadd_rn(__fmul_rn(pn.x, nx), __fadd_rn(__fmul_rn(pn.y, ny), __fmul_rn(pn.z, nz)));
        pw = pn.w;
      }
      hm = __ballot(gate_ok(dot, a.gate_lo, a.gate_hi));
      // this lane's registered weight (0 when it does not register), reduced once per group
      s_w[wave][gs][lane] = sel_mask(hm, pw, 0.0f);
      wrote |= 1u << gs;
    }
    cnt_pack |= (unsigned long long)(uint32_t)__popcll(hm) << (8 * gs);
    PGP_STAMP(t_g);   // gate + weight parked
    PGP_PHASE(4, t_f, t_g);
    PGP_PHASE(5, t_b, t_g);   // the whole non-empty part
  };

  // after the last hypothesis of a group (slots g0s .. g0s + n - 1 of the chunk): lane 16k publishes slot k
  auto publish = [&](const int g0s, const int n) {
    int pl = lane;
    asm volatile("" : "+v"(pl));   // the lane arithmetic is redone here, once per group, not held in registers
    const int k = pl >> 4;
    const bool pub = (pl & 15) == 0 && k < n;
    if (pub) s_cnt[wave][g0s + k] = (int)((cnt_pack >> (8 * k)) & 0xFFull);
    if (kW) {
      // lanes 16k .. 16k+15 (one DPP row) sum row k of the group: 4 consecutive floats each, in order, then
      // the four DPP steps that fold a row (xor 1, xor 2, half-row mirror, row mirror).  A FIXED association:
      // same bits every run; the reference's own (sequential) order is restored for near-ties by
      // finalize_scores.
      float t = 0.0f;
      if (wrote != 0u) {
        __builtin_amdgcn_wave_barrier();
        const float4 v0 = *reinterpret_cast<const float4*>(&s_w[wave][k][(pl & 15) * 4]);
        t = __fadd_rn(__fadd_rn(__fadd_rn(v0.x, v0.y), v0.z), v0.w);
        t = dpp_step<0xB1>(t);   // quad_perm [1,0,3,2]
        t = dpp_step<0x4E>(t);   // quad_perm [2,3,0,1]
        t = dpp_step<0x141>(t);  // row_half_mirror
        t = dpp_step<0x140>(t);  // row_mirror
        // a row nobody wrote holds an older group's values: its sum is 0
        t = ((wrote >> k) & 1u) != 0u ? t : 0.0f;
        __builtin_amdgcn_wave_barrier();
      }
      if (pub) s_sum[wave][g0s + k] = t;
    }
    cnt_pack = 0ull;
    wrote = 0u;
  };

  // the hypothesis' 4x4 for the NEXT trip is requested (scalar loads) at the top of the current one, so
  // that its latency runs under this trip's vector loads instead of in front of the next transform.
  // (Two trips per pass with a register set each -- no copies -- measured the same and cost 8 SGPRs.)
  const int n_slots = h1 - h0;
  Xf m_pre = load_xf(Tm, (uint32_t)h0);
  for (int hs = 0; hs < n_slots; ++hs) {
    const Xf m = m_pre;
    m_pre = load_xf(Tm, (uint32_t)(h0 + min(hs + 1, n_slots - 1)));
    trip(hs, m);
    if (((hs & (kSumGroup - 1)) == kSumGroup - 1) || hs == n_slots - 1) publish(hs & ~(kSumGroup - 1), (hs & (kSumGroup - 1)) + 1);
  }
#if defined(PGP_ABLATE) && PGP_ABLATE == 10
  if (lane == 0) {
    const int wid = (blockIdx.x * (kTile / 64) + wave) % kPhaseWaves;
    for (int k = 0; k < 8; ++k) g_phase[wid][k] = ph[k];
    g_phase[wid][7] = 1ull;   // this wave ran
  }
#endif
  __syncthreads();
  const int hh = threadIdx.x;
#if defined(PGP_FUSED) && PGP_FUSED
  if (threadIdx.x >= 64) return;   // the block's <= 64 hypotheses live in wave 0
  {
    const FuseArgs z = fuse_args();
    unsigned long long key = 0;
    bool fin = false;
    if (hh < h1 - h0) {
      int c = 0;
#pragma unroll
      for (int w = 0; w < kTile / 64; ++w) c += s_cnt[w][hh];
      float f = 0.f;
      if (kW) {
#pragma unroll
        for (int w = 0; w < kTile / 64; ++w) f += s_sum[w][hh];
      }
      const int h = h0 + hh;
      const unsigned long long one = 1ull << 48, low = one - 1ull;
      const unsigned long long last_mark = (unsigned long long)(a.n_tiles - 1);
      // both adds are in flight together: ONE round trip to the memory side
      unsigned long long oldB = 0, oldA = 0;
      const unsigned long long fx = kW ? (unsigned long long)__float2ll_rn(__fmul_rn(f, z.fx_scale)) : 0ull;
      if (kW) oldB = __hip_atomic_fetch_add(&z.acc[z.acc_stride + h], one | fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool want_a = !kW || z.counts != nullptr;
      if (want_a) oldA = __hip_atomic_fetch_add(&z.acc[h], one | (unsigned long long)(uint32_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float score = 0.f;
      if (want_a && (oldA >> 48) == last_mark) {
        const int total = (int)((oldA & low) + (unsigned long long)(uint32_t)c);
        z.acc[h] = 0ull;   // re-armed for the next launch (nobody else touches it in this one)
        if (z.counts) z.counts[h] = total;
        if (!kW) {
          score = __fdiv_rn((float)total, (float)a.nQ);
          fin = true;
        }
      }
      if (kW && (oldB >> 48) == last_mark) {
        const unsigned long long sum = (oldB & low) + fx;
        z.acc[z.acc_stride + h] = 0ull;
        score = __fdiv_rn(__double2float_rn((double)sum * z.fx_inv), (float)a.nQ);
        fin = true;
      }
      if (fin) {
        __hip_atomic_store(&z.scores[h], score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (score > 0.f) key = ((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)h);
      }
    }
    const unsigned long long fm = __ballot(fin);
    if (fm == 0ull) return;      // 19 of 20 blocks end here
    // the hypotheses this block finalised: their top-2 keys, then the launch's (two atomics + the arrival count)
    const unsigned long long mykey = key;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      unsigned long long o = __shfl_xor(key, off, 64);
      key = o > key ? o : key;
    }
    unsigned long long key2 = mykey == key ? 0ull : mykey;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      unsigned long long o = __shfl_xor(key2, off, 64);
      key2 = o > key2 ? o : key2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through score stores of this wave are complete
    if (threadIdx.x == 0) {
      if (key) {
        const unsigned long long old = atomicMax(z.best_key, key);
        const unsigned long long push = old < key ? old : key;
        if (push) atomicMax(z.runner_key, push);
      }
      if (key2) atomicMax(z.runner_key, key2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned n_fin = (unsigned)__popcll(fm);
      const bool last = atomicAdd(z.done, n_fin) + n_fin == (unsigned)a.n_h;
      if (last) {
        const unsigned long long kk = atomicExch(z.best_key, 0ull);
        atomicExch(z.runner_key, 0ull);
        atomicExch(z.done, 0u);
        // (experiment: weighted near-ties are NOT settled here)
        if (kk == 0) {
          z.best[0] = -1;
          z.best[1] = 0;
        } else {
          z.best[0] = (int)(0xFFFFFFFFu - (unsigned)(kk & 0xFFFFFFFFull));
          z.best[1] = (int)(unsigned)(kk >> 32);
        }
      }
    }
  }
#else
  if (hh < h1 - h0) {
    int c = 0;
#pragma unroll
    for (int w = 0; w < kTile / 64; ++w) c += s_cnt[w][hh];
    float f = 0.f;
    if (kW) {
#pragma unroll
      for (int w = 0; w < kTile / 64; ++w) f += s_sum[w][hh];
    }
    a.partial[(size_t)tile * a.n_h + h0 + hh] = make_uint2((uint32_t)c, __float_as_uint(f));
  }
#endif
#if defined(PGP_ABLATE) && PGP_ABLATE == 9
  // timing experiment: what a per-block agent-scope release + ticket would cost (fused finalize)
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned int*>(a.partial) + 2 * chunk + 1, 1u);
#endif
}

// 8 waves per SIMD: 64 VGPRs, 78 SGPRs, 20 KB of LDS per workgroup.  All three limits were hit while this
// kernel was shaped (tools/ab_step.sh A/B runs): a 7-wave build with 72 VGPRs is 6 %
// slower, three-chunk weighted batches with spills 12 % slower, 8 KB more LDS = 7 workgroups per CU 7 %.
template <int MODE>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void score_hypotheses_flat(
    ScoreArgs a, const float* __restrict__ Tm, const uint2* __restrict__ words, const uint2* __restrict__ occ_run,
    const float4* __restrict__ cand, const float4* __restrict__ Pnw, FuseArgs) {
  score_flat_body<MODE, 3, false>(a, Tm, words, occ_run, cand, Pnw);
}

// pgp_set_exact_ties, weighted mode: the same kernel with the tie flags and the reference's tree for tied candidates
template <bool SPARSE>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void score_hypotheses_flat_ties(
    ScoreArgs a, const float* __restrict__ Tm, const uint2* __restrict__ words, const uint2* __restrict__ occ_run,
    const float4* __restrict__ cand, const float4* __restrict__ Pnw, FuseArgs) {
  score_flat_body<PGP_MODE_WEIGHTED, 3, SPARSE, true>(a, Tm, words, occ_run, cand, Pnw);
}

// the same kernel over the sparse block table (grid_index.hip): `words` is the uint4 table
template <int MODE>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void score_hypotheses_flat_sparse(
    ScoreArgs a, const float* __restrict__ Tm, const uint2* __restrict__ words, const uint2* __restrict__ occ_run,
    const float4* __restrict__ cand, const float4* __restrict__ Pnw, FuseArgs) {
  score_flat_body<MODE, 3, true>(a, Tm, words, occ_run, cand, Pnw);
}

// One model point (Morton position i) under one transform: the scene id it registers to (after
// the normal gate in weighted mode), or -1.  Per-lane walk of the candidate run.
template <int MODE>
__device__ __forceinline__ int point_hit(const ScoreArgs& a, const Xf& m, int i) {
  const float4 q = a.Q[i];
  const float x = xf_row(m.m00, m.m01, m.m02, m.m03, q.x, q.y, q.z);
  const float y = xf_row(m.m10, m.m11, m.m12, m.m13, q.x, q.y, q.z);
  const float z = xf_row(m.m20, m.m21, m.m22, m.m23, q.x, q.y, q.z);
  uint32_t s, e;
  cell_run(a.g, a.words, a.occ_run, x, y, z, &s, &e, true);
  e += s;
  int id = nearest_by_rule(a, s, e, x, y, z);
  if (MODE == PGP_MODE_WEIGHTED && id >= 0) {
    const float4 qn = a.Qn[i];
    const float nx = rot_row(m.m00, m.m01, m.m02, qn.x, qn.y, qn.z);
    const float ny = rot_row(m.m10, m.m11, m.m12, qn.x, qn.y, qn.z);
    const float nz = rot_row(m.m20, m.m21, m.m22, qn.x, qn.y, qn.z);
    const float4 pn = a.Pnw[id];
    const float dot = __fadd_rn(__fmul_rn(pn.x, nx), __fadd_rn(__fmul_rn(pn.y, ny), __fmul_rn(pn.z, nz)));
    if (!gate_ok(dot, a.gate_lo, a.gate_hi)) id = -1;
  }
  return id;
}

// One transform, per-model-point result in ORIGINAL model order (the Q arrays are Morton-sorted;
// q.w carries the original index): hit id after the gate, or -1.
template <int MODE>
__global__ __launch_bounds__(256) void registered_points(ScoreArgs a, int* __restrict__ hits) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nQ) return;
  const Xf m = load_xf(a.T, 0);
  hits[__float_as_int(a.Q[i].w)] = point_hit<MODE>(a, m, i);
}

// Match4PCSBase::getRegisteredModel (base.cc:347-375; not called by ComputeTransformation): the
// verifier's loop over ANOTHER cloud (sampled_Q_3D_, the search model) with the 30-degree gate applied
// to the directed angle -- the fold `min(a, 180 - a)` is commented out there (:368), so anti-parallel
// normals do NOT register.  Per point, in cloud order: scene id or -1.
__global__ __launch_bounds__(256) void registered_model(ScoreArgs a, const float4* __restrict__ q_xyz,
                                                        const float4* __restrict__ q_nrm, int n,
                                                        int* __restrict__ hits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Xf m = load_xf(a.T, 0);
  const float4 q = q_xyz[i];
  const float x = xf_row(m.m00, m.m01, m.m02, m.m03, q.x, q.y, q.z);
  const float y = xf_row(m.m10, m.m11, m.m12, m.m13, q.x, q.y, q.z);
  const float z = xf_row(m.m20, m.m21, m.m22, m.m23, q.x, q.y, q.z);
  uint32_t s, e;
  cell_run(a.g, a.words, a.occ_run, x, y, z, &s, &e, true);
  e += s;
  int id = nearest_by_rule(a, s, e, x, y, z);
  if (id >= 0) {
    const float4 qn = q_nrm[i];
    const float nx = rot_row(m.m00, m.m01, m.m02, qn.x, qn.y, qn.z);
    const float ny = rot_row(m.m10, m.m11, m.m12, qn.x, qn.y, qn.z);
    const float nz = rot_row(m.m20, m.m21, m.m22, qn.x, qn.y, qn.z);
    const float4 pn = a.Pnw[id];
    const float dot = __fadd_rn(__fmul_rn(pn.x, nx), __fadd_rn(__fmul_rn(pn.y, ny), __fmul_rn(pn.z, nz)));
    if (!(dot >= a.gate_lo && dot <= 1.0f)) id = -1;   // acos(dot) * 180 / pi < gate, no fold; NaN rejected
  }
  hits[i] = id;
}

// Sum the per-tile partials in tile order, form the score exactly as the reference does
// (Scalar(good_points)/Scalar(number_of_points), base.cc:1730 / weighted_match/Scalar(n), :1765)
// and fold the batch arg-max: key = score bits << 32 | ~index, so the maximum key is the highest
// score at its LOWEST index = what `lcp > best_LCP_` (strict, base.cc:1891) ends on.
// The last block to finish (ticket) publishes {best index, best score bits} and re-arms the key
// and the ticket for the next call, so a scoring call is two launches: score + finalize.
//
// Weighted mode, returned best pose (base.cc:1759,1891): the reference accumulates
// `weighted_match += w[hit]` sequentially in model order, this library in a fixed tree (per wave,
// per tile), so two hypotheses whose scores differ by less than that re-association could swap
// places.  Size of the effect: each of the reference's c <= nQ additions rounds by at most half an
// ulp of a partial sum <= S = score * nQ, a random walk with standard deviation of about
// 0.41 * 2^-24 * S * sqrt(c) <= 0.41 * 2^-24 * sqrt(nQ) * score per score (3.7e-7 at the C2 sizes;
// largest deviation seen over 1.4e5 hypotheses there: 2e-6); the tree's own error is a log2(nQ) / nQ
// fraction of that.  Hypotheses within refine_tol() = TEN such deviations of the maximum (6.2e-6
// at C2) are candidates: if there is more than one (the launch's top-2 keys tell), each of them
// (in index order, at most kRefineCap) is re-scored EXACTLY as the reference does it -- registered
// weights scattered to original model order, then ONE lane per candidate adds them sequentially
// in float -- its score entry is overwritten with that value, and the arg-max is taken over the
// exact values with the reference's strict `>`.  With a single candidate the arg-max cannot depend
// on the association and nothing more runs.
constexpr int kRefineCap = 128;
constexpr int kRefineGroup = 4;         // candidates settled together (one summation lane each)
constexpr int kSeqChunk = 1024;         // floats per candidate staged through LDS per step
constexpr int kSeqStride4 = kSeqChunk / 4 + 16;   // float4 per staged row: sixteen of padding (seq_sum_rows reads ahead)

__device__ __forceinline__ float refine_tol(float best_score, int nQ) {
  const float rel = fmaxf(4.1f * 5.9604645e-8f * __fsqrt_rn((float)nQ), 4.8e-7f);
  return best_score * rel;
}

// Sequential float sums of G <= kRefineGroup rows of nQ4 floats (row g at seq + g * nQ4), lane g's return value = row g's sum
// added in index order from +0.0f (base.cc:1737-1759).  All 256 threads of the block, wave g serving row g: the rows pass
// through LDS in chunks of kSeqChunk floats with their ZEROS DROPPED in order (x + 0 == x for every x the sum can hold: it
// starts at +0.0f and can not become -0.0f; a point that does not register contributes +0.0f, and that is most of them for
// any pose but a very good one), the next chunk's global loads are in flight (registers) while lanes 0..G-1 add the current
// one, and the LDS reads run eight 16-byte requests ahead of the adds -- the lane's time is then its chain of dependent adds
// (~10 cycles each on this part) over the REGISTERED points only, not a load round trip per four model points (16 us per
// 1500 points before, profiles/r04_dropin_kernels.txt).
__device__ __forceinline__ float seq_sum_rows(const float* __restrict__ seq, int nQ4, int G, float4 (*s_stage)[kSeqStride4],
                                              int* s_cnt) {
  static_assert(kRefineGroup == 4 && kSeqChunk % 64 == 0, "one wave of the 256-thread block per row");
  constexpr int kPer = kSeqChunk / 64;   // floats per lane and chunk
  if (nQ4 <= 0) return 0.0f;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const float* row = seq + (size_t)min(g, G - 1) * nQ4;
  float regs[kPer];
  auto fetch = [&](int c0) {   // every load issued, none behind a branch, none consumed before park()
    const int len = min(kSeqChunk, nQ4 - c0);
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = k * 64 + lane;
      regs[k] = row[i < len ? c0 + i : 0];
    }
  };
  auto park = [&](int c0) {    // the chunk's non-zero values, in order, then zeros up to a whole group of the adding loop
    const int len = min(kSeqChunk, nQ4 - c0);
    float* dst = reinterpret_cast<float*>(s_stage[g]);
    int n = 0;
    if (g < G) {
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const bool nz = k * 64 + lane < len && regs[k] != 0.0f;   // (a NaN weight is kept and poisons the sum as it must)
        const unsigned long long m = __ballot(nz);
        if (nz) dst[n + __popcll(m & ((1ull << lane) - 1ull))] = regs[k];
        n += __popcll(m);
      }
      dst[n + lane] = 0.0f;   // rows hold kSeqChunk + 64 floats
      if (lane == 0) s_cnt[g] = n;
    }
  };
  float S = 0.0f;
  fetch(0);
  park(0);
  __syncthreads();
  for (int c0 = 0; c0 < nQ4; c0 += kSeqChunk) {
    const bool more = c0 + kSeqChunk < nQ4;
    if (more) fetch(c0 + kSeqChunk);
    if ((int)threadIdx.x < G) {
      const float4* v = s_stage[threadIdx.x];
      const int n4 = (s_cnt[threadIdx.x] + 63) / 64 * 16;   // whole groups of sixteen float4
      float4 A[8], B[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) A[k] = v[k];
      for (int i = 0; i < n4; i += 16) {
        // B is requested before A's 32 dependent adds and A's next eight before B's (rows are padded by sixteen float4, so
        // the last request needs no clamp)
#pragma unroll
        for (int k = 0; k < 8; ++k) B[k] = v[i + 8 + k];
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise gathers all sixteen reads at the top and waits for them)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          S = __fadd_rn(S, A[k].x);
          S = __fadd_rn(S, A[k].y);
          S = __fadd_rn(S, A[k].z);
          S = __fadd_rn(S, A[k].w);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) A[k] = v[i + 16 + k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          S = __fadd_rn(S, B[k].x);
          S = __fadd_rn(S, B[k].y);
          S = __fadd_rn(S, B[k].z);
          S = __fadd_rn(S, B[k].w);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (more) {
      park(c0 + kSeqChunk);
      __syncthreads();
    }
  }
  return S;
}

// Exact scores of G <= kRefineGroup hypotheses hs[0..G) -> s_out[0..G).  All threads of the block.
__device__ void refine_exact_group(const ScoreArgs& a, const int* hs, int G, float* __restrict__ seq,
                                   float4 (*s_stage)[kSeqStride4], int* s_cnt, float* s_out) {
  const int nQ4 = (a.nQ + 3) & ~3;
  // registered weights of every candidate, scattered to ORIGINAL model order: two model points per
  // thread and trip, so that their lookup chains overlap
  for (int g = 0; g < G; ++g) {
    const Xf m = load_xf(a.T, hs[g]);
    float* row = seq + (size_t)g * nQ4;
    for (int i0 = threadIdx.x; i0 < nQ4; i0 += 2 * blockDim.x) {
      const int i1 = i0 + blockDim.x;
      const int id0 = i0 < a.nQ ? point_hit<PGP_MODE_WEIGHTED>(a, m, i0) : -1;
      const int id1 = i1 < a.nQ ? point_hit<PGP_MODE_WEIGHTED>(a, m, i1) : -1;
      // adding +0.0f is the identity on the reference's running sum (which starts at +0.0f)
      const float w0 = id0 >= 0 ? a.Pnw[id0].w : 0.0f, w1 = id1 >= 0 ? a.Pnw[id1].w : 0.0f;
      if (i0 < a.nQ) row[__float_as_int(a.Q[i0].w)] = w0;
      else row[i0] = 0.0f;   // padding up to a multiple of four
      if (i1 < a.nQ) row[__float_as_int(a.Q[i1].w)] = w1;
      else if (i1 < nQ4) row[i1] = 0.0f;
    }
  }
  __syncthreads();
  const float S = seq_sum_rows(seq, nQ4, G, s_stage, s_cnt);
  if ((int)threadIdx.x < G) s_out[threadIdx.x] = __fdiv_rn(S, (float)a.nQ);   // base.cc:1765
  __syncthreads();
}

// Called by all 256 threads of ONE block with kk = the arg-max key over the (tree-summed) score
// vector: publishes {best index, best score bits}, settling weighted near-ties exactly (above).
__device__ __forceinline__ void settle_and_publish(const ScoreArgs& a, int n_h, int mode, int refine, float* scores,
                                                const unsigned long long kk, int* __restrict__ best,
                                                float* __restrict__ seq, unsigned long long* s_key) {
  if (kk == 0) {
    if (threadIdx.x == 0) {
      best[0] = -1;
      best[1] = 0;  // best_LCP_ = 0.0f
    }
    return;
  }
  int bi = (int)(0xFFFFFFFFu - (unsigned)(kk & 0xFFFFFFFFull));
  float bs = __uint_as_float((unsigned)(kk >> 32));
  if (mode == PGP_MODE_WEIGHTED && refine) {
    __shared__ int s_red[4];
    __shared__ float4 s_stage[kRefineGroup][kSeqStride4];
    __shared__ int s_seqcnt[kRefineGroup];
    __shared__ float s_exact[kRefineGroup];
    __shared__ int s_cand[kRefineGroup];
    const float thr = bs - refine_tol(bs, a.nQ);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cnt = 0;
    for (int i = threadIdx.x; i < n_h; i += blockDim.x) cnt += scores[i] >= thr ? 1 : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) s_red[wave] = cnt;
    __syncthreads();
    cnt = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    if (cnt >= 2) {
      int cur = -1, done = 0, ebi = -1;
      float ebest = 0.0f;
      bool more = true;
      while (more && done < kRefineCap) {
        // the next (up to) kRefineGroup candidates in index order
        int G = 0;
        for (; G < kRefineGroup && done + G < kRefineCap; ++G) {
          int nxt = 0x7FFFFFFF;
          for (int i = cur + 1 + threadIdx.x; i < n_h; i += blockDim.x)
            if (scores[i] >= thr) { nxt = i; break; }
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) nxt = min(nxt, __shfl_xor(nxt, off, 64));
          if (lane == 0) s_red[wave] = nxt;
          __syncthreads();
          nxt = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
          __syncthreads();
          if (nxt == 0x7FFFFFFF) { more = false; break; }
          if (threadIdx.x == 0) s_cand[G] = nxt;
          cur = nxt;
        }
        __syncthreads();
        if (G == 0) break;
        refine_exact_group(a, s_cand, G, seq, s_stage, s_seqcnt, s_exact);
        for (int g = 0; g < G; ++g) {
          const float e = s_exact[g];
          const int h = s_cand[g];
          if (threadIdx.x == 0) scores[h] = e;   // the reference's value, bit for bit
          if (e > ebest) {                        // strict >, ascending index (base.cc:1891)
            ebest = e;
            ebi = h;
          }
        }
        done += G;
        __syncthreads();
      }
      // candidates past the cap (pathological: > kRefineCap near-equal hypotheses) keep their
      // tree-summed values and can only take over with a strictly greater one
      unsigned long long rest = 0;
      for (int i = cur + 1 + threadIdx.x; i < n_h; i += blockDim.x) {
        const float v = scores[i];
        if (v >= thr && v > ebest) {
          const unsigned long long kx = ((unsigned long long)__float_as_uint(v) << 32) | (0xFFFFFFFFu - (unsigned)i);
          rest = kx > rest ? kx : rest;
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(rest, off, 64);
        rest = o > rest ? o : rest;
      }
      if (lane == 0) s_key[wave] = rest;
      __syncthreads();
      rest = s_key[0];
      for (int w = 1; w < 4; ++w) rest = s_key[w] > rest ? s_key[w] : rest;
      if (rest) {
        ebi = (int)(0xFFFFFFFFu - (unsigned)(rest & 0xFFFFFFFFull));
        ebest = __uint_as_float((unsigned)(rest >> 32));
      }
      if (ebi >= 0) {
        bi = ebi;
        bs = ebest;
      }
    }
  }
  if (threadIdx.x == 0) {
    best[0] = bi;
    best[1] = (int)__float_as_uint(bs);
  }
}

// The host-pointer call's results on their way home WITHOUT a copy engine and without the end-of-kernel signal: every block
// also stores its scores and counts straight into the call's pinned landing area (system-scope stores, complete before the
// block's ticket), and the launch's last block writes the best and then a completion word the host polls (pgp_api.hip
// pgp_score_lcp).  A device-to-host DMA behind the kernel costs ~10 us whatever its size and the stream's completion signal a
// few more: 13 us of a 144 us call (profiles/r06_ab/fused_exact_tile_order.log).  best[2] = 1: a weighted near-tie was settled
// on the device (scores rewritten there): the host fetches the arrays the old way for that call.
struct HostPub {
  float* scores;   // NULL: nothing is published
  int* counts;
  int* best;       // {index, score bits, settled-on-device, spare}
  unsigned int* flag;
  unsigned int flag_value;
};

// The kernels' FIRST parameter, read back through the kernel-argument pointer where the rare settlement path needs it: taken
// from the by-value parameter, its 50 words were fetched at the top of the kernel and held in scalar registers through the
// common path (42-51 scalar spills in finalize_scores / settle_best_kernel).
__device__ __forceinline__ const ScoreArgs& score_args_in_kernarg() {
#if defined(__HIP_DEVICE_COMPILE__)
  const ScoreArgs* ap = (const ScoreArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ap));   // the loads stay where they are used
  return *ap;
#else
  static const ScoreArgs none{};
  return none;
#endif
}

__global__ __launch_bounds__(256) void finalize_scores(ScoreArgs /* read through score_args_in_kernarg() */, const uint2* __restrict__ partial,
                                                       int n_tiles, int n_h, int nQ, int mode, int refine,
                                                       float* scores, int* __restrict__ counts,
                                                       unsigned long long* best_key, unsigned long long* runner_key,
                                                       unsigned int* ticket, int* __restrict__ best,
                                                       float* __restrict__ seq, HostPub hp) {
  int h = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long key = 0;
  if (h < n_h) {
    int c = 0;
    float f = 0.f;
    // the partials were written by the kernel before this one, on every die: each load is a trip to the
    // memory side, so sixteen tiles' loads are issued together and then added IN TILE ORDER (a dependent
    // load-add chain took 20 round trips, ~9 of this kernel's 12 us at C2)
    for (int t0 = 0; t0 < n_tiles; t0 += 16) {
      uint2 pp[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) pp[j] = partial[(size_t)min(t0 + j, n_tiles - 1) * n_h + h];
      // (the C2 model has 20 tiles: the second batch's four loads go out before the first batch is consumed)
      uint2 pq[4];
      const bool tail4 = n_tiles - (t0 + 16) > 0 && n_tiles - (t0 + 16) <= 4;
      if (tail4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pq[j] = partial[(size_t)min(t0 + 16 + j, n_tiles - 1) * n_h + h];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (t0 + j < n_tiles) {
          c += (int)pp[j].x;
          f += __uint_as_float(pp[j].y);
        }
      }
      if (tail4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (t0 + 16 + j < n_tiles) {
            c += (int)pq[j].x;
            f += __uint_as_float(pq[j].y);
          }
        }
        break;
      }
    }
    float score = (mode == PGP_MODE_PLAIN) ? __fdiv_rn((float)c, (float)nQ) : __fdiv_rn(f, (float)nQ);
    // write-through (global_store ... sc1): the score leaves this XCD's L2 for memory, where the block that
    // settles a near-tie -- on whatever die -- can read it after ONE acquire of its own, taken only in that
    // rare case.  (A plain store + a release fence per block + an acquire in the last block cost this launch
    // two ~3.5 us fences on its critical chain, every call: MI355X_MICROARCH.md, __threadfence row.)
    __hip_atomic_store(&scores[h], score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (counts) counts[h] = c;
    if (hp.scores) {
      __hip_atomic_store(&hp.scores[h], score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&hp.counts[h], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (score > 0.f)  // NaN and <= 0 never become best (best_LCP_ starts at 0, strict >)
      key = ((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)h);
  }
  // block top-2: the wave maximum, then the maximum of what is left (keys > 0 are unique: they carry h)
  const unsigned long long mykey = key;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned long long o = __shfl_xor(key, off, 64);
    key = o > key ? o : key;
  }
  unsigned long long key2 = mykey == key ? 0ull : mykey;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned long long o = __shfl_xor(key2, off, 64);
    key2 = o > key2 ? o : key2;
  }
  __shared__ unsigned long long s_key[4], s_key2[4];
  __shared__ unsigned long long s_kk, s_kk2;
  __shared__ int s_last;
  if ((threadIdx.x & 63) == 0) {
    s_key[threadIdx.x >> 6] = key;
    s_key2[threadIdx.x >> 6] = key2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through score stores are complete ...
  __syncthreads();                                    // ... and so are every wave's of this block, before its ticket
  if (threadIdx.x == 0) {
    unsigned long long k = 0, k2 = 0;
    for (int w = 0; w < 4; ++w) {
      const unsigned long long a1 = s_key[w], a2 = s_key2[w];
      if (a1 > k) { k2 = k > a2 ? k : a2; k = a1; }
      else if (a1 > k2) k2 = a1;
      // a2 <= a1: it can only displace the runner-up
      if (a2 > k2 && a2 != k) k2 = a2;
    }
    // global top-2 with two atomics per block (device scope: performed at L2, visible to every XCD):
    // every value except the final maximum is pushed to the runner-up slot, either by the block that
    // displaced it from the first slot or by its own block when it lost there
    if (k) {
      const unsigned long long old = atomicMax(best_key, k);
      const unsigned long long push = old < k ? old : k;
      if (push) atomicMax(runner_key, push);
    }
    if (k2) atomicMax(runner_key, k2);
    // device-scope atomics are performed at the memory side and need no fence to be seen by the other dies;
    // they must only have been performed before this block's ticket is
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    if (last) {
      s_kk = atomicExch(best_key, 0ull);  // read the result and re-arm
      s_kk2 = atomicExch(runner_key, 0ull);
      atomicExch(ticket, 0u);
    }
    s_last = last ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last block of the launch: publish (and, in weighted mode, settle near-ties exactly) ----
  const unsigned long long kk = s_kk, kk2 = s_kk2;
  const bool near_tie = mode == PGP_MODE_WEIGHTED && refine && kk2 != 0ull &&
                        __uint_as_float((unsigned)(kk2 >> 32)) >=
                            __uint_as_float((unsigned)(kk >> 32)) - refine_tol(__uint_as_float((unsigned)(kk >> 32)), nQ);
  if (near_tie) {
    // the other blocks' scores are in memory (write-through stores, drained before their tickets); this CU's
    // L1 may hold stale lines of them: one agent-scope acquire by one lane, completed before anyone reads
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#if !(defined(PGP_ABLATE) && PGP_ABLATE == 11)   // 11: what the common launch would cost without the settlement in its code object
    settle_and_publish(score_args_in_kernarg(), n_h, mode, refine, scores, kk, best, seq, s_key);
#endif
    if (hp.scores) {   // (settled values and the best live in device memory: the host copies them back for this call)
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_store(&hp.best[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(hp.flag, hp.flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  } else if (threadIdx.x == 0) {
    const int bi = kk == 0 ? -1 : (int)(0xFFFFFFFFu - (unsigned)(kk & 0xFFFFFFFFull));
    const int bs = kk == 0 ? 0 : (int)(unsigned)(kk >> 32);   // (none: best_LCP_ = 0.0f)
    best[0] = bi;
    best[1] = bs;
    if (hp.scores) {
      __hip_atomic_store(&hp.best[0], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&hp.best[1], bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&hp.best[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // Every block's host stores were complete (acknowledged: system-scope stores are written through) before its ticket;
      // these three before the word the host polls.  No release FENCE: at system scope that is a write-back of everything
      // dirty in this die's L2 -- the launch's 0.6 MB of partials among it -- for the sake of stores that never were in it.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(hp.flag, hp.flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The same publication over a COMPLETE score vector that was assembled elsewhere (the slices of
// several devices after the all-reduce, pgp_settle_best_device): one block finds the arg-max key
// and settles near-ties with this device's copy of the clouds and the full transform list.
__global__ __launch_bounds__(256) void settle_best_kernel(ScoreArgs /* read through score_args_in_kernarg() */, int n_h, int mode, int refine,
                                                          float* scores, int* __restrict__ best,
                                                          float* __restrict__ seq, int nQ) {
  __shared__ unsigned long long s_key[4], s_key2[4];
  unsigned long long key = 0, key2 = 0;   // this thread's best and second-best keys (keys > 0 are unique: they carry h)
  // sixteen scores per thread in flight: ONE block walks the complete vector of a device group (65 536 scores at configs[3]),
  // and one load per trip made that walk 256 dependent round trips -- ~60 us behind every exchange (round 5)
  for (int h0 = threadIdx.x; h0 < n_h; h0 += 16 * (int)blockDim.x) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int h = h0 + u * (int)blockDim.x;
      v[u] = h < n_h ? scores[h] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int h = h0 + u * (int)blockDim.x;
      if (v[u] > 0.f) {   // (a slot past the end holds 0)
        const unsigned long long k = ((unsigned long long)__float_as_uint(v[u]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)h);
        if (k > key) {
          key2 = key;
          key = k;
        } else if (k > key2) {
          key2 = k;
        }
      }
    }
  }
  // the block's top two: merge the lanes' pairs (the runner-up of a merge is the larger of the loser and the winner's second)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(key, off, 64), o2 = __shfl_xor(key2, off, 64);
    const unsigned long long hi = o > key ? o : key, lo = o > key ? key : o;
    const unsigned long long s2 = o > key ? o2 : key2;   // the winner's own second
    key = hi;
    key2 = lo > s2 ? lo : s2;
  }
  if ((threadIdx.x & 63) == 0) {
    s_key[threadIdx.x >> 6] = key;
    s_key2[threadIdx.x >> 6] = key2;
  }
  __syncthreads();
  key = 0;
  key2 = 0;
  for (int w = 0; w < 4; ++w) {
    const unsigned long long a1 = s_key[w], a2 = s_key2[w];
    if (a1 > key) {
      key2 = key > a2 ? key : a2;
      key = a1;
    } else {
      if (a1 > key2) key2 = a1;
    }
  }
  __syncthreads();
  // the walk of settle_and_publish over all scores (its count of the values within the tolerance of the maximum) only when the
  // runner-up is within it -- the same decision finalize_scores takes from its two keys
  const bool near_tie = mode == PGP_MODE_WEIGHTED && refine && key != 0ull && key2 != 0ull &&
                        __uint_as_float((unsigned)(key2 >> 32)) >=
                            __uint_as_float((unsigned)(key >> 32)) - refine_tol(__uint_as_float((unsigned)(key >> 32)), nQ);
  if (near_tie) {
    settle_and_publish(score_args_in_kernarg(), n_h, mode, refine, scores, key, best, seq, s_key);
  } else if (threadIdx.x == 0) {
    if (key == 0) {
      best[0] = -1;
      best[1] = 0;  // best_LCP_ = 0.0f
    } else {
      best[0] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
      best[1] = (int)(unsigned)(key >> 32);
    }
  }
}

// ---- exact scores at every decision of the running-best walk (base.cc:1891-1908) ---------------------------
// The reference keeps hypothesis i when lcp_i > best so far (strict); the list of those records is what the
// drop-in returns as hypothesisSet.  With tree-summed weighted scores an entry within the re-association error
// of the then-running maximum could enter or stay out differently from the reference (DESIGN.md section 2,
// divergence iv).  Three small launches (opt-in, pgp_set_exact_records) find every NEAR-RECORD -- a
// hypothesis whose tree score reaches the running maximum of the tree scores before it minus twice the
// settlement tolerance --, compute the registered weight of every (near-record, model point) in parallel and
// overwrite each near-record's score with the reference's own sequential sum, one lane per near-record.  The running maximum is always attained at a record, every record and every
// hypothesis that could displace or tie one is a near-record, and their values are now the reference's bit
// for bit: the host walk over the score vector (pgp_running_best) returns the reference's list.  About
// ln(n_h) + a few candidates per call; past kRecordCap (pathological) the rest keep their tree values.
constexpr int kRecordCap = 128;

// 1. the near-records, in index order (one block)
__global__ __launch_bounds__(256) void records_find(int n_h, int nQ, const float* __restrict__ scores,
                                                    int* __restrict__ list, int* __restrict__ count) {
  // 4096 scores per pass, 16 consecutive ones per thread held in registers: ONE batch of loads, the running maximum in front of
  // every thread's run by a prefix-max over the block, the positions of its near-records by a prefix sum (19 us as three
  // passes of dependent loads with serial prefix loops, profiles/r04_dropin_kernels.txt)
  constexpr int kPer = 16;
  __shared__ float s_wmax[4];
  __shared__ int s_wcnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float carry = 0.f;   // records are > 0 (best_LCP_ starts at 0)
  int base = 0;
  for (int s0 = 0; s0 < n_h; s0 += 256 * kPer) {
    float v[kPer];
    const int lo = s0 + tid * kPer;
#pragma unroll
    for (int k = 0; k < kPer; ++k) v[k] = lo + k < n_h ? scores[lo + k] : 0.f;
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < kPer; ++k) m = fmaxf(m, v[k]);
    float inc = m;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float o = __shfl_up(inc, off, 64);
      if (lane >= off) inc = fmaxf(inc, o);
    }
    if (lane == 63) s_wmax[wave] = inc;
    float run = __shfl_up(inc, 1, 64);   // running maximum of the tree scores before this thread's run
    if (lane == 0) run = 0.f;
    __syncthreads();
    run = fmaxf(run, carry);
    for (int w = 0; w < wave; ++w) run = fmaxf(run, s_wmax[w]);
    carry = fmaxf(carry, fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3])));
    unsigned mask = 0;
    {
      float r = run;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const float x = v[k];
        if (x > 0.f && x >= r - 2.0f * refine_tol(fmaxf(r, x), nQ)) mask |= 1u << k;
        r = fmaxf(r, x);
      }
    }
    const int cnt = __popc(mask);
    int pre = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(pre, off, 64);
      if (lane >= off) pre += o;
    }
    if (lane == 63) s_wcnt[wave] = pre;
    __syncthreads();
    int off = base + pre - cnt;
    for (int w = 0; w < wave; ++w) off += s_wcnt[w];
    base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    while (mask) {
      const int k = __ffs(mask) - 1;
      mask &= mask - 1;
      if (off < kRecordCap) list[off] = lo + k;
      ++off;
    }
    __syncthreads();   // s_wmax / s_wcnt are rewritten by the next pass
  }
  if (tid == 0) *count = min(base, kRecordCap);
}

// 2. the registered weight of every (near-record, model point), one row of nQ4 floats per near-record in
//    ORIGINAL model order (+0.0f where the point does not register: the identity on the running sum)
__global__ __launch_bounds__(256) void records_weights(ScoreArgs a, const int* __restrict__ list,
                                                       const int* __restrict__ count, float* __restrict__ seq) {
  const int k = blockIdx.y;
  if (k >= *count) return;
  const int nQ4 = (a.nQ + 3) & ~3;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nQ4) return;
  float* row = seq + (size_t)k * nQ4;
  if (i >= a.nQ) {   // padding up to a multiple of four
    row[i] = 0.0f;
    return;
  }
  const Xf m = load_xf(a.T, (uint32_t)list[k]);
  const int id = point_hit<PGP_MODE_WEIGHTED>(a, m, i);
  row[__float_as_int(a.Q[i].w)] = id >= 0 ? a.Pnw[id].w : 0.0f;
}

// 3. a block per kRefineGroup near-records: their rows are staged through LDS (the whole block loads, at
//    bandwidth) and ONE lane per near-record adds its row sequentially in model order (base.cc:1759), then
//    overwrites the score with the reference's value (base.cc:1765).  Blocks run side by side, so the time is
//    one row's 5000 dependent adds whatever the number of near-records.
__global__ __launch_bounds__(256) void records_sum(int nQ, const int* __restrict__ list, const int* __restrict__ count,
                                                   const float* __restrict__ seq, float* __restrict__ scores) {
  __shared__ float4 s_stage[kRefineGroup][kSeqStride4];
  __shared__ int s_cnt[kRefineGroup];
  const int g0 = blockIdx.x * kRefineGroup;
  const int n = *count;
  if (g0 >= n) return;
  const int G = min(kRefineGroup, n - g0);
  const int nQ4 = (nQ + 3) & ~3;
  const float S = seq_sum_rows(seq + (size_t)g0 * nQ4, nQ4, G, s_stage, s_cnt);
  if ((int)threadIdx.x < G) scores[list[g0 + threadIdx.x]] = __fdiv_rn(S, (float)nQ);
}

// ---- Verify's early termination, reproduced (base.cc:1699-1731, opt-in: pgp_set_verify_early_out) -----------
// The reference's plain verifier stops a hypothesis as soon as it can no longer beat the running best:
//   terminate_value = (int)(best_LCP_ * N);  after model point i:  if (N - i + good < terminate_value) break;
// and returns good / N at that point, so what it writes to allPose[i].second for a hypothesis that is not a
// new best is a LOWER BOUND that depends on the order of the batch and of the model.  The scoring kernel
// counts every hypothesis completely (its counts are the true ones; best index and best score never differ,
// since a hypothesis that would become the best can not terminate early).  With the option on, a second
// pass rewrites the scores and counts of the others to the reference's values:
//   * best_LCP_ in front of hypothesis h = the maximum of the TRUE scores before it (a terminated
//     hypothesis returns less than the best it was compared with), an exclusive prefix maximum;
//   * a hypothesis whose true count reaches its terminate_value runs to the end (N - i + good_i >
//     good_final >= terminate_value for every i): untouched;
//   * the others walk the model in ORIGINAL order, 64 points per wave-step (hit test = the per-lane
//     candidate walk of pgp_registered), prefix counts by ballot, and stop at the first i that satisfies
//     the reference's test.
// Cost: up to one more pass over (hypotheses x model), typically half of it; results equal the reference's
// Verify bit for bit (tests/golden/*.npz early_out_scores, from the Eigen harness).
__global__ __launch_bounds__(1024) void early_out_terminate_values(const int* __restrict__ counts, int n_h, int nQ,
                                                                    int* __restrict__ tv) {
  __shared__ float s_wave[16];
  __shared__ float s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0.f;   // best_LCP_ starts at 0
  __syncthreads();
  for (int base = 0; base < n_h; base += 1024) {
    const int h = base + tid;
    const float sc = h < n_h ? __fdiv_rn((float)counts[h], (float)nQ) : 0.f;
    float incl = sc;   // inclusive prefix maximum inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float o = __shfl_up(incl, off, 64);
      if (lane >= off) incl = fmaxf(incl, o);
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    float before = s_carry;   // maximum of everything in front of this wave
    for (int w = 0; w < wave; ++w) before = fmaxf(before, s_wave[w]);
    float excl = __shfl_up(incl, 1, 64);
    excl = lane == 0 ? before : fmaxf(before, excl);
    // (int)(best_LCP_ * number_of_points): float x float(size_t), truncated
    if (h < n_h) tv[h] = (int)__fmul_rn(excl, (float)nQ);
    __syncthreads();
    if (tid == 1023) s_carry = fmaxf(before, incl);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void early_out_walk(ScoreArgs a, const int* __restrict__ qpos, const int* __restrict__ tv,
                                                      float* __restrict__ scores, int* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int h = blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per hypothesis
  if (h >= a.n_h) return;
  const int term = tv[h], N = a.nQ;
  if (counts[h] >= term) return;   // runs to the end in the reference too: the true count stands
  const Xf m = load_xf(a.T, (uint32_t)h);
  int good = 0;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    const bool hit = i < N && point_hit<PGP_MODE_PLAIN>(a, m, qpos[i]) >= 0;
    const unsigned long long bm = __ballot(hit);
    const int good_i = good + __popcll(bm & ((2ull << lane) - 1ull));   // hits among points 0 .. i
    const unsigned long long stop = __ballot(i < N && N - i + good_i < term);
    if (stop) {
      const int first = __ffsll((long long)stop) - 1;
      const int g = __shfl(good_i, first, 64);
      if (lane == 0) {
        counts[h] = g;
        scores[h] = __fdiv_rn((float)g, (float)N);
      }
      return;
    }
    good += __popcll(bm);
  }
}

__global__ void publish_none(int* __restrict__ best) {  // empty hypothesis list (base.cc:1791-1794)
  best[0] = -1;
  best[1] = 0;
}

// base.cc:1756-1758 evaluated on the host, exactly as the reference evaluates it.
bool gate_pred(float dot, float gate_deg) {
  float ac = std::acos(dot);
  float ac180 = ac * 180;
  float angle_n = (float)((double)ac180 / M_PI);
  angle_n = std::min(angle_n, std::fabs(180 - angle_n));
  return angle_n < gate_deg;
}

// float <-> monotone integer key
int32_t f2key(float f) {
  int32_t i;
  std::memcpy(&i, &f, 4);
  return i >= 0 ? i : (int32_t)(0x80000000u - (uint32_t)i);
}
float key2f(int32_t k) {
  int32_t i = k >= 0 ? k : (int32_t)(0x80000000u - (uint32_t)k);
  float f;
  std::memcpy(&f, &i, 4);
  return f;
}

// ev0 / ev1 (both or neither): start / stop events attached to the dispatch itself
// (hipExtLaunchKernelGGL) -- the kernel's own begin / end timestamps, without the two barrier
// packets that hipEventRecord before and after the launch put on the stream (those cost the
// C2 step 8 us, 7 % of its throughput, when per-kernel timing was on).
#if defined(PGP_CAND8) && PGP_CAND8
const float4* cand8_override = nullptr;   // experiment: the packed candidates of the context being launched
#endif
void launch_variant(int mode, int unroll, dim3 grid, hipStream_t stream, const ScoreArgs& a, hipEvent_t ev0,
                    hipEvent_t ev1, const FuseArgs& fz = FuseArgs{}) {
  // default by measurement at C2 (tools/tune.py): wave-flattened 112 us plain / 157 us weighted vs
  // per-lane walk (U = 2) 125 / 170 us
  // (a scene so far from the origin that its lattice numbers leave the mantissa trick's range takes the
  // per-lane kernel, which finds cells by subtraction and truncation)
  if (unroll <= 0 && a.g.magic_ok && a.kd_nodes && mode == PGP_MODE_WEIGHTED) {   // exact ties (pgp_set_exact_ties)
    if (a.g.sparse)
      hipExtLaunchKernelGGL(score_hypotheses_flat_ties<true>, grid, dim3(kTile), 0, stream, ev0, ev1, 0, a, a.T, a.words,
                            a.occ_run, a.cand, a.Pnw, fz);
    else
      hipExtLaunchKernelGGL(score_hypotheses_flat_ties<false>, grid, dim3(kTile), 0, stream, ev0, ev1, 0, a, a.T, a.words,
                            a.occ_run, a.cand, a.Pnw, fz);
    return;
  }
  if (unroll <= 0 && a.g.magic_ok && a.g.sparse) {  // wave-flattened candidate phase over the sparse block table
    if (mode == PGP_MODE_PLAIN)
      hipExtLaunchKernelGGL(score_hypotheses_flat_sparse<PGP_MODE_PLAIN>, grid, dim3(kTile), 0, stream, ev0, ev1, 0, a,
                            a.T, a.words, a.occ_run, a.cand, a.Pnw, fz);
    else
      hipExtLaunchKernelGGL(score_hypotheses_flat_sparse<PGP_MODE_WEIGHTED>, grid, dim3(kTile), 0, stream, ev0, ev1, 0,
                            a, a.T, a.words, a.occ_run, a.cand, a.Pnw, fz);
    return;
  }
  if (unroll <= 0 && a.g.magic_ok) {  // wave-flattened candidate phase (dense block array)
    const float4* cand_arg = a.cand;
#if defined(PGP_CAND8) && PGP_CAND8
    cand_arg = cand8_override ? cand8_override : a.cand;
#endif
    if (mode == PGP_MODE_PLAIN)
      hipExtLaunchKernelGGL(score_hypotheses_flat<PGP_MODE_PLAIN>, grid, dim3(kTile), 0, stream, ev0, ev1, 0, a, a.T,
                            a.words, a.occ_run, cand_arg, a.Pnw, fz);
    else
      hipExtLaunchKernelGGL(score_hypotheses_flat<PGP_MODE_WEIGHTED>, grid, dim3(kTile), 0, stream, ev0, ev1, 0, a,
                            a.T, a.words, a.occ_run, cand_arg, a.Pnw, fz);
    return;
  }
#define PGP_LAUNCH(M, UU) \
  hipExtLaunchKernelGGL((score_hypotheses<M, UU>), grid, dim3(kTile), 0, stream, ev0, ev1, 0, a)
  // per-lane walk, two hypotheses unrolled (U = 1, 4, 8 measured within 2 % or slower)
  if (mode == PGP_MODE_PLAIN) PGP_LAUNCH(PGP_MODE_PLAIN, 2);
  else PGP_LAUNCH(PGP_MODE_WEIGHTED, 2);
#undef PGP_LAUNCH
}

int fill_args(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg, ScoreArgs* a) {
  if (!ctx->has_index) {
    set_error("no scene index: call pgp_set_scene first");
    return PGP_ESTATE;
  }
  if (ctx->nQ <= 0 && n_h > 0 && !ctx->d_Q.p) {
    set_error("no model: call pgp_set_model first");
    return PGP_ESTATE;
  }
  if (mode != PGP_MODE_PLAIN && mode != PGP_MODE_WEIGHTED) {
    set_error("unknown mode %d", mode);
    return PGP_EINVAL;
  }
  if (mode == PGP_MODE_WEIGHTED && !(ctx->has_scene_normals && ctx->has_model_normals)) {
    set_error("weighted mode needs scene and model normals");
    return PGP_ESTATE;
  }
  if (mode == PGP_MODE_WEIGHTED && ctx->gate_deg_cached != gate_deg) {
    gate_thresholds(gate_deg, &ctx->gate_lo, &ctx->gate_hi);
    ctx->gate_deg_cached = gate_deg;
  }
  a->g = ctx->grid;
  a->cell_c[0] = 12582912.0f - (float)ctx->grid.k0x;
  a->cell_c[1] = 12582912.0f - (float)ctx->grid.k0y;
  a->cell_c[2] = 12582912.0f - (float)ctx->grid.k0z;
  a->cell_hi = kMagicBits + std::max(ctx->grid.nx, std::max(ctx->grid.ny, ctx->grid.nz)) - 1;
  a->last_word = (uint32_t)ctx->grid.nbx * (uint32_t)ctx->grid.nby * (uint32_t)ctx->grid.nbz - 1u;
  a->words = ctx->grid.sparse ? ctx->d_blocktab.as<uint2>() : ctx->d_bitmap.as<uint2>();
  a->occ_run = ctx->d_occ_start.as<uint2>();
  const bool ties = ctx->exact_ties && ctx->kd_valid;
  a->kd_nodes = ties ? ctx->d_kd_nodes.as<int4>() : nullptr;
  a->kd_pts = ties ? ctx->d_kd_pts.as<float4>() : nullptr;
  a->cand = ctx->d_cand.as<float4>();
  a->Pnw = ctx->d_Pnw.as<float4>();
  a->Q = ctx->d_Q.as<float4>();
  a->Qn = ctx->d_Qn.as<float4>();
  a->nQ = ctx->nQ;
  a->T = d_T;
  a->n_h = n_h;
  a->n_tiles = tiles_for(ctx->nQ);
  // delta*delta in float, as `sq_eps = epsilon*epsilon` (base.cc:1711)
  a->sq_eps = ctx->delta * ctx->delta;
  a->gate_lo = ctx->gate_lo;
  a->gate_hi = ctx->gate_hi;
  return PGP_OK;
}

}  // namespace

int tiles_for(int nQ) { return nQ > 0 ? (nQ + kTile - 1) / kTile : 1; }

void gate_thresholds(float gate_deg, float* c_aligned_min, float* c_anti_max) {
  // pred(d) is true on [-1, hi] U [lo, 1] (acosf is monotone): bisect both edges over the
  // ordered float keys.  Degenerate gates collapse naturally (gate >= 90: everything in
  // [-1,1] passes; gate <= 0: nothing does).
  *c_aligned_min = 2.f;   // nothing passes
  *c_anti_max = -2.f;
  if (!(gate_deg == gate_deg)) return;
  if (gate_pred(1.0f, gate_deg)) {
    int32_t lo = f2key(0.0f), hi = f2key(1.0f);  // pred(hi) true; find first true in [0,1]
    if (gate_pred(0.0f, gate_deg)) {
      *c_aligned_min = 0.0f;
    } else {
      while (hi - lo > 1) {
        int32_t mid = lo + (hi - lo) / 2;
        if (gate_pred(key2f(mid), gate_deg)) hi = mid; else lo = mid;
      }
      *c_aligned_min = key2f(hi);
    }
  }
  if (gate_pred(-1.0f, gate_deg)) {
    int32_t lo = f2key(-1.0f), hi = f2key(-0.0f);  // pred(lo) true; find last true in [-1,-0]
    if (gate_pred(-0.0f, gate_deg)) {
      *c_anti_max = 0.0f;
    } else {
      while (hi - lo > 1) {
        int32_t mid = lo + (hi - lo) / 2;
        if (gate_pred(key2f(mid), gate_deg)) lo = mid; else hi = mid;
      }
      *c_anti_max = key2f(lo);
    }
  }
}

int launch_score(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg,
                 float* d_scores, int* d_counts, int* d_best, hipStream_t stream, ScoreHostOut* host) {
  ScoreArgs a{};
  if (host) host->published = false;
  int rc = await_index(ctx, stream);
  if (rc == PGP_OK) rc = fill_args(ctx, d_T, n_h, mode, gate_deg, &a);
  if (rc != PGP_OK) return rc;
  if (n_h > ctx->cap_h) {
    set_error("n_h %d exceeds reserved capacity %d (pgp_reserve)", n_h, ctx->cap_h);
    return PGP_ESTATE;
  }
  unsigned long long* key = ctx->d_best.as<unsigned long long>();   // zeroed at create, re-armed
  int* best_local = reinterpret_cast<int*>(key + 1);                 // by every finalize launch
  unsigned int* ticket = reinterpret_cast<unsigned int*>(key + 2);
  if (n_h > 0) {
    // ~40 workgroups per CU (measured best at C2: hpb 8 -> 125 us vs 136 us at hpb 20); at least
    // 4 hypotheses per block to amortise the model-point load
    long long work = (long long)n_h * a.n_tiles;
    int hpb = ctx->hpb_override > 0 ? ctx->hpb_override : (int)(work / 10240);
    // small batches: a workgroup's trips run one after the other (~1.2 us each), so with few workgroups per
    // CU the launch lasts as long as its longest workgroup: 2 hypotheses per block up to 512, 3 up to 768, 4
    // beyond (tools/small_batch.py: 512 hypotheses 29.4 -> 26.1 us, 256: 23.1 -> 19.9 us; 1024: 38.9 at 4, 40.0 at 2)
    if (hpb < 4 && ctx->hpb_override <= 0) {
      hpb = (int)(work / 5120);
      hpb = hpb < 2 ? 2 : (hpb > 4 ? 4 : hpb);
    }
    // more than 16 per block never paid (tools/tune.py: 16 384 hypotheses 313 us at 8..16 vs 325 us
    // at 32; 65 536 hypotheses 1168 us at 16 vs 1201 us at 64); an explicit PGP_HPB may go to 64
    if (ctx->hpb_override <= 0 && hpb > 16) hpb = 16;
    if (hpb > kMaxHpb) hpb = kMaxHpb;
    a.hpb = hpb;
    // the last tail_pct % of the hypotheses go in chunks of hpb_tail (C2, tools/tune.py: 10 % in chunks
    // of 2 -> 87-88 us plain / 114-116 us weighted against 90.5 / 119.5 us with uniform chunks; 20-40 %
    // or chunks of 1, 3, 4 gain less)
    static const int tail_pct = getenv("PGP_TAIL_PCT") ? atoi(getenv("PGP_TAIL_PCT")) : 10;
    static const int tail_hpb = getenv("PGP_TAIL_HPB") ? atoi(getenv("PGP_TAIL_HPB")) : 2;
    a.hpb_tail = tail_hpb > 0 && tail_hpb <= hpb ? tail_hpb : hpb;
    int n_tail_h = (int)((long long)n_h * (tail_pct < 0 ? 0 : tail_pct > 100 ? 100 : tail_pct) / 100);
    a.n_big = (n_h - n_tail_h) / hpb;
    n_tail_h = n_h - a.n_big * hpb;
    a.n_chunks = a.n_big + (n_tail_h + a.hpb_tail - 1) / a.hpb_tail;
    a.partial = ctx->d_partial.as<uint2>();
    int chunks_pad = (a.n_chunks + 7) / 8 * 8;
    dim3 grid((unsigned)(chunks_pad * a.n_tiles));
#if defined(PGP_CAND8) && PGP_CAND8
    {
      // experiment only: the packed copy of this context's candidate lists, rebuilt when the index changed (recognised by
      // its array, its size and the grid's origin); never freed
      struct Packed { DevBuf buf; const void* cand = nullptr; long long n = -1; float ox = 0.f, h = 0.f; };
      static std::map<pgp_ctx*, Packed> packed;
      Packed& pk = packed[ctx];
      cand8_override = nullptr;
      if (!ctx->grid.sparse && ctx->grid.magic_ok && ctx->nP <= 65535 && !(ctx->exact_ties && ctx->kd_valid)) {
        if (pk.cand != ctx->d_cand.p || pk.n != ctx->n_cand || pk.ox != ctx->grid.ox || pk.h != ctx->grid.h) {
          if ((rc = finish_index(ctx)) != PGP_OK) return rc;
          if ((rc = pk.buf.ensure(((size_t)ctx->n_cand + 256) * 8)) != PGP_OK) return rc;
          const uint32_t n_words = (uint32_t)ctx->grid.nbx * (uint32_t)ctx->grid.nby * (uint32_t)ctx->grid.nbz;
          hipLaunchKernelGGL(pack_cand8, dim3((unsigned)(((size_t)n_words * 32 + 255) / 256)), dim3(256), 0, stream, ctx->grid,
                             (const uint2*)a.words, (const uint2*)a.occ_run, (const float4*)a.cand, pk.buf.as<uint2>(), n_words);
          pk.cand = ctx->d_cand.p;
          pk.n = ctx->n_cand;
          pk.ox = ctx->grid.ox;
          pk.h = ctx->grid.h;
        }
        cand8_override = reinterpret_cast<const float4*>(pk.buf.p);
      }
    }
#endif
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->timing > 0 && (ctx->timing_seq++ % (unsigned)ctx->timing) == 0) {
      if (ctx->ev_used + 2 > ctx->ev.size()) {
        for (int k = 0; k < 2; ++k) {
          hipEvent_t e;
          PGP_HIP(hipEventCreate(&e));
          ctx->ev.push_back(e);
        }
      }
      ev0 = ctx->ev[ctx->ev_used];
      ev1 = ctx->ev[ctx->ev_used + 1];
      ctx->ev_used += 2;
    }
#if defined(PGP_FUSED) && PGP_FUSED
    const bool fused = ctx->unroll <= 0 && a.g.magic_ok;
    FuseArgs fz{};
    if (fused) {
      fz.acc = ctx->d_acc.as<unsigned long long>();
      fz.acc_stride = ctx->cap_h;
      fz.scores = d_scores;
      fz.counts = d_counts ? d_counts : (ctx->verify_early_out && mode == PGP_MODE_PLAIN ? ctx->d_counts.as<int>() : nullptr);
      fz.best_key = key;
      fz.runner_key = key + 3;
      fz.done = ticket;
      fz.best = d_best ? d_best : best_local;
      // the weight sums in fixed point: nQ * w_max * 2^shift < 2^47 (w_max = 1: probabilities, base.cc:317-324)
      int shift = 46 - std::ilogb((double)std::max(a.nQ, 1));
      shift = shift > 40 ? 40 : shift;
      fz.fx_scale = std::ldexp(1.0f, shift);
      fz.fx_inv = std::ldexp(1.0, -shift);
    }
    launch_variant(mode, ctx->unroll, grid, stream, a, ev0, ev1, fz);
    if (!fused)
#else
    launch_variant(mode, ctx->unroll, grid, stream, a, ev0, ev1);
#endif
    hipLaunchKernelGGL(finalize_scores, dim3((n_h + 255) / 256), dim3(256), 0, stream, a,
                       (const uint2*)a.partial, a.n_tiles, n_h, a.nQ,
                       mode, ctx->refine_best ? 1 : 0, d_scores,
                       d_counts ? d_counts : (ctx->verify_early_out && mode == PGP_MODE_PLAIN ? ctx->d_counts.as<int>() : nullptr),
                       key, key + 3, ticket,
                       d_best ? d_best : best_local, ctx->d_seq.as<float>(),
                       host ? HostPub{host->scores, host->counts, host->best, host->flag, host->flag_value} : HostPub{});
    if (host) host->published = true;
  } else {
    hipLaunchKernelGGL(publish_none, dim3(1), dim3(1), 0, stream, d_best ? d_best : best_local);
  }
  PGP_HIP(hipGetLastError());
  if (ctx->exact_records && n_h > 0) return launch_settle_records(ctx, d_T, n_h, mode, gate_deg, d_scores, stream);
  if (ctx->verify_early_out && mode == PGP_MODE_PLAIN && n_h > 0)
    return launch_verify_early_out(ctx, d_T, n_h, d_scores, d_counts ? d_counts : ctx->d_counts.as<int>(), stream);
  return PGP_OK;
}

// scores / counts of a COMPLETE plain-mode batch (true counts in) rewritten to what the reference's Verify
// returns with its early termination; queue-only
int launch_verify_early_out(pgp_ctx* ctx, const float* d_T, int n_h, float* d_scores, int* d_counts, hipStream_t stream) {
  if (n_h <= 0) return PGP_OK;
  ScoreArgs a{};
  int rc = await_index(ctx, stream);
  if (rc == PGP_OK) rc = fill_args(ctx, d_T, n_h, PGP_MODE_PLAIN, 30.f, &a);
  if (rc != PGP_OK) return rc;
  if (ctx->d_eo_ws.cap < (size_t)n_h * sizeof(int) || !ctx->d_Qpos.p) {
    set_error("verify early-out: workspace not reserved (pgp_reserve / pgp_set_model)");
    return PGP_ESTATE;
  }
  int* tv = ctx->d_eo_ws.as<int>();
  hipLaunchKernelGGL(early_out_terminate_values, dim3(1), dim3(1024), 0, stream, (const int*)d_counts, n_h, a.nQ, tv);
  hipLaunchKernelGGL(early_out_walk, dim3((n_h + 3) / 4), dim3(256), 0, stream, a, (const int*)ctx->d_Qpos.as<int>(),
                     (const int*)tv, d_scores, d_counts);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

int launch_settle_best(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg, float* d_scores,
                       int* d_best, hipStream_t stream, float* seq_ws) {
  if (n_h <= 0) {
    hipLaunchKernelGGL(publish_none, dim3(1), dim3(1), 0, stream, d_best);
    PGP_HIP(hipGetLastError());
    return PGP_OK;
  }
  ScoreArgs a{};
  int rc = await_index(ctx, stream);
  if (rc == PGP_OK) rc = fill_args(ctx, d_T, n_h, mode, gate_deg, &a);
  if (rc != PGP_OK) return rc;
  hipLaunchKernelGGL(settle_best_kernel, dim3(1), dim3(256), 0, stream, a, n_h, mode, ctx->refine_best ? 1 : 0,
                     d_scores, d_best, seq_ws ? seq_ws : ctx->d_seq.as<float>(), a.nQ);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

int records_workspace_bytes(int nQ) { return (int)(((size_t)(((nQ > 0 ? nQ : 1) + 3) & ~3) * kRecordCap + kRecordCap + 64) * 4); }

int launch_settle_records(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg, float* d_scores,
                          hipStream_t stream) {
  if (n_h <= 0 || mode != PGP_MODE_WEIGHTED) return PGP_OK;   // plain counts are exact already
  ScoreArgs a{};
  int rc = await_index(ctx, stream);
  if (rc == PGP_OK) rc = fill_args(ctx, d_T, n_h, mode, gate_deg, &a);
  if (rc != PGP_OK) return rc;
  if (ctx->d_rec_ws.cap < (size_t)records_workspace_bytes(ctx->nQ)) {
    set_error("exact records: workspace not reserved (pgp_set_exact_records after pgp_set_model)");
    return PGP_ESTATE;
  }
  int* list = ctx->d_rec_ws.as<int>();          // [kRecordCap] near-records | count | weights
  int* count = list + kRecordCap;
  float* seq = reinterpret_cast<float*>(list + kRecordCap + 64);
  hipLaunchKernelGGL(records_find, dim3(1), dim3(256), 0, stream, n_h, a.nQ, (const float*)d_scores, list, count);
  hipLaunchKernelGGL(records_weights, dim3((a.nQ + 3 + 255) / 256, kRecordCap), dim3(256), 0, stream, a, (const int*)list,
                     (const int*)count, seq);
  hipLaunchKernelGGL(records_sum, dim3((kRecordCap + kRefineGroup - 1) / kRefineGroup), dim3(256), 0, stream, a.nQ,
                     (const int*)list, (const int*)count, (const float*)seq, d_scores);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

// Neighbour count of every scene point inside the scene's own index (segment pre-processing,
// pgp_radius_outlier_filter): strict d2 < r2 as FLANN's radius search, the point itself included,
// squared distance accumulated in x, y, z order as FLANN's L2_Simple does.
__global__ __launch_bounds__(256) void count_neighbours(GridDesc g, const uint2* __restrict__ words,
                                                        const uint2* __restrict__ occ_run,
                                                        const float4* __restrict__ cand,
                                                        const float4* __restrict__ P, int nP, float r2,
                                                        int* __restrict__ counts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nP) return;
  const float4 p = P[i];
  uint32_t s, e;
  cell_run(g, words, occ_run, p.x, p.y, p.z, &s, &e, true);
  e += s;
  int k = 0;
  for (uint32_t j = s; j < e; ++j) {
    const float4 c = cand[j];
    const float dx = __fsub_rn(p.x, c.x), dy = __fsub_rn(p.y, c.y), dz = __fsub_rn(p.z, c.z);
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    k += d2 < r2;
  }
  counts[i] = k;
}

int launch_count_neighbours(pgp_ctx* ctx, float radius, int* d_counts, hipStream_t stream) {
  if (!ctx->has_index) {
    set_error("no scene index: call pgp_set_scene first");
    return PGP_ESTATE;
  }
  if (radius > ctx->delta) {
    set_error("radius %g exceeds the index radius %g", (double)radius, (double)ctx->delta);
    return PGP_EINVAL;
  }
  if (ctx->nP == 0) return PGP_OK;
  {
    const int rc = await_index(ctx, stream);
    if (rc != PGP_OK) return rc;
  }
  hipLaunchKernelGGL(count_neighbours, dim3((ctx->nP + 255) / 256), dim3(256), 0, stream, ctx->grid,
                     ctx->grid.sparse ? ctx->d_blocktab.as<uint2>() : ctx->d_bitmap.as<uint2>(), ctx->d_occ_start.as<uint2>(), ctx->d_cand.as<float4>(),
                     ctx->d_P.as<float4>(), ctx->nP, radius * radius, d_counts);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

int launch_registered_model(pgp_ctx* ctx, const float* d_T16, const float4* d_q, const float4* d_qn, int n,
                            float gate_deg, int* d_hits, hipStream_t stream) {
  if (!ctx->has_scene_normals) {
    set_error("pgp_registered_model needs scene normals");
    return PGP_ESTATE;
  }
  if (!ctx->has_index) {
    set_error("no scene index: call pgp_set_scene first");
    return PGP_ESTATE;
  }
  if (ctx->gate_deg_cached != gate_deg) {
    gate_thresholds(gate_deg, &ctx->gate_lo, &ctx->gate_hi);
    ctx->gate_deg_cached = gate_deg;
  }
  {
    const int rc = await_index(ctx, stream);
    if (rc != PGP_OK) return rc;
  }
  ScoreArgs a{};
  a.g = ctx->grid;
  a.words = ctx->grid.sparse ? ctx->d_blocktab.as<uint2>() : ctx->d_bitmap.as<uint2>();
  a.occ_run = ctx->d_occ_start.as<uint2>();
  a.cand = ctx->d_cand.as<float4>();
  a.Pnw = ctx->d_Pnw.as<float4>();
  a.T = d_T16;
  a.n_h = 1;
  a.sq_eps = ctx->delta * ctx->delta;
  a.gate_lo = ctx->gate_lo;
  a.gate_hi = ctx->gate_hi;
  if (n == 0) return PGP_OK;
  hipLaunchKernelGGL(registered_model, dim3((n + 255) / 256), dim3(256), 0, stream, a, d_q, d_qn, n, d_hits);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

int launch_registered(pgp_ctx* ctx, const float* d_T16, int mode, float gate_deg, int* d_hits,
                      hipStream_t stream) {
  ScoreArgs a{};
  int rc = await_index(ctx, stream);
  if (rc == PGP_OK) rc = fill_args(ctx, d_T16, 1, mode, gate_deg, &a);
  if (rc != PGP_OK) return rc;
  if (ctx->nQ == 0) return PGP_OK;
  dim3 grid((ctx->nQ + 255) / 256);
  if (mode == PGP_MODE_PLAIN)
    hipLaunchKernelGGL(registered_points<PGP_MODE_PLAIN>, grid, dim3(256), 0, stream, a, d_hits);
  else
    hipLaunchKernelGGL(registered_points<PGP_MODE_WEIGHTED>, grid, dim3(256), 0, stream, a, d_hits);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

#if defined(PGP_ABLATE) && PGP_ABLATE == 10
extern "C" int pgp_debug_phase_cycles(unsigned long long* out16, int reset) {
  // sums over the waves of the LAST launch(es) since the last reset: out16[0..6] phase ticks, out16[7] waves
  std::vector<unsigned long long> h((size_t)kPhaseWaves * 8);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_phase), h.size() * 8) != hipSuccess) return -1;
  for (int k = 0; k < 16; ++k) out16[k] = 0;
  for (int w = 0; w < kPhaseWaves; ++w)
    for (int k = 0; k < 8; ++k) out16[k] += h[(size_t)w * 8 + k];
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h.data(), h.size() * 8) != hipSuccess) return -1;
  }
  return 0;
}
#endif

}  // namespace pgp
