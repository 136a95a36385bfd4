// csrc/rigid_fit.hip -- batched rigid transform from congruent (base, quad) pairs.
//
// Replaces Match4PCSBase::ComputeRigidTransformFromCongruentPair
// (S4/algorithms/match4pcsBase.cc:1411-1488) and ComputeRigidTransformation (:1504-1614,
// computeScale = false, max_angle < 0 as shipped) for a whole list of congruent quads at once:
// one thread per pair, the 4+4 points gathered by index from the resident scene / search-model
// arrays, output = the centred 4x4 that the verification loop scores (`allTransforms`,
// base.cc:1468) and the de-centred double pose returned to the node (`allPose`, :1484).
//
// Float parity: every expression is evaluated in Eigen's order with separately rounded
// operations, IEEE sqrt and divide (sqrt_rn below, __fdiv_rn) -- the centred transform and the rms
// are bit-identical to the reference expressions (pinned via oracle/, tests/golden/rigid_fit.npz).
// The de-centred translation uses the linear part where the reference multiplies the SVD polar
// factors rot*scale (computeRotationScaling, base.cc:1480): equal to ~1e-7.
//
// Work per pair is ~200 flops on 96 gathered bytes: launch-latency bound at the reference's
// batch sizes (<= 10^4 pairs); it lives on the device so that congruent-set -> fit -> score
// never leaves HBM.

#include "pgp_internal.h"

namespace pgp {

namespace {

struct V3 {
  float x, y, z;
};

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

// Correctly rounded float sqrt.  ROCm 7.2's __fsqrt_rn / sqrtf are 1 ulp off for ~15 % of inputs
// on gfx950 (measured: 632 495 of 4 194 304 random values differ from IEEE sqrtf), the double
// square root is exact, and rounding a 53-bit correctly rounded root to 24 bits is innocuous.
__device__ __forceinline__ float sqrt_rn(float z) { return (float)__dsqrt_rn((double)z); }

__device__ __forceinline__ float sum3(float a, float b, float c) { return add(a, add(b, c)); }  // a + (b + c)
__device__ __forceinline__ float sqnorm(V3 v) { return sum3(mul(v.x, v.x), mul(v.y, v.y), mul(v.z, v.z)); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return sum3(mul(a.x, b.x), mul(a.y, b.y), mul(a.z, b.z)); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {sub(a.x, b.x), sub(a.y, b.y), sub(a.z, b.z)}; }

__device__ __forceinline__ V3 normalized(V3 v) {  // Eigen normalize(): if z > 0: v /= sqrt(z)
  float z = sqnorm(v);
  if (z > 0.f) {
    float n = sqrt_rn(z);
    v.x = __fdiv_rn(v.x, n);
    v.y = __fdiv_rn(v.y, n);
    v.z = __fdiv_rn(v.z, n);
  }
  return v;
}

__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {sub(mul(a.y, b.z), mul(a.z, b.y)), sub(mul(a.z, b.x), mul(a.x, b.z)), sub(mul(a.x, b.y), mul(a.y, b.x))};
}

// Gram-Schmidt frame of base.cc:1532-1546; false on the degenerate exits
__device__ __forceinline__ bool frame(V3 a0, V3 a1, V3 a2, V3* v1, V3* v2, V3* v3) {
  V3 e1 = vsub(a1, a0);
  if (sqnorm(e1) == 0.f) return false;
  e1 = normalized(e1);
  V3 d = vsub(a2, a0);
  float proj = dot(d, e1);
  V3 e2 = {sub(d.x, mul(proj, e1.x)), sub(d.y, mul(proj, e1.y)), sub(d.z, mul(proj, e1.z))};
  if (sqnorm(e2) == 0.f) return false;
  e2 = normalized(e2);
  *v1 = e1;
  *v2 = e2;
  *v3 = cross(e1, e2);
  return true;
}

__device__ __forceinline__ V3 ld3(const float4* __restrict__ a, int i) {
  float4 v = a[i];
  return {v.x, v.y, v.z};
}

__device__ __forceinline__ float comp(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

__global__ __launch_bounds__(128) void rigid_from_congruent(const float4* __restrict__ P, int nP,
                                                            const float4* __restrict__ Qs, int nQs,
                                                            const int4* __restrict__ base_ids,
                                                            const int4* __restrict__ quad_ids, int n,
                                                            float cPx, float cPy, float cPz, float cQx,
                                                            float cQy, float cQz, float* __restrict__ T,
                                                            double* __restrict__ pose,
                                                            int* __restrict__ status,
                                                            float* __restrict__ rms_out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float kNaN = __int_as_float(0x7fc00000);
  float* Ti = T + 16 * (size_t)i;
  double* Pi = pose ? pose + 16 * (size_t)i : nullptr;
  // a rejected / degenerate pair leaves a NaN transform: it scores 0 in the verification loop
  auto fail = [&](int code, float rms) {
    for (int k = 0; k < 16; ++k) Ti[k] = kNaN;
    if (Pi)
      for (int k = 0; k < 16; ++k) Pi[k] = (double)kNaN;
    status[i] = code;
    if (rms_out) rms_out[i] = rms;
  };
  int4 b = base_ids[i], c = quad_ids[i];
  if ((unsigned)b.x >= (unsigned)nP || (unsigned)b.y >= (unsigned)nP || (unsigned)b.z >= (unsigned)nP ||
      (unsigned)c.x >= (unsigned)nQs || (unsigned)c.y >= (unsigned)nQs || (unsigned)c.z >= (unsigned)nQs) {
    fail(-1, 1e9f);  // index out of range (the reference would read out of bounds)
    return;
  }
  V3 p0 = ld3(P, b.x), p1 = ld3(P, b.y), p2 = ld3(P, b.z);
  V3 q0 = ld3(Qs, c.x), q1 = ld3(Qs, c.y), q2 = ld3(Qs, c.z);
  // centroids of the first three points (base.cc:1431,1453-1455): ((a + b) + c) / 3
  V3 c1 = {__fdiv_rn(add(add(p0.x, p1.x), p2.x), 3.f), __fdiv_rn(add(add(p0.y, p1.y), p2.y), 3.f),
           __fdiv_rn(add(add(p0.z, p1.z), p2.z), 3.f)};
  V3 c2 = {__fdiv_rn(add(add(q0.x, q1.x), q2.x), 3.f), __fdiv_rn(add(add(q0.y, q1.y), q2.y), 3.f),
           __fdiv_rn(add(add(q0.z, q1.z), q2.z), 3.f)};
  V3 fp1, fp2, fp3, fq1, fq2, fq3;
  if (!frame(p0, p1, p2, &fp1, &fp2, &fp3) || !frame(q0, q1, q2, &fq1, &fq2, &fq3)) {
    fail(2, 1e9f);
    return;
  }
  // rotation = rotate_p^T * rotate_q : R(r,c) = p1_r q1_c + (p2_r q2_c + p3_r q3_c)
  float R[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
      R[r][cc] = sum3(mul(comp(fp1, r), comp(fq1, cc)), mul(comp(fp2, r), comp(fq2, cc)), mul(comp(fp3, r), comp(fq3, cc)));
  // discard non-orthogonal solutions: diag(R*R) - 1 > 1e-6 (base.cc:1563)
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float d = sum3(mul(R[r][0], R[0][r]), mul(R[r][1], R[1][r]), mul(R[r][2], R[2][r]));
    if (sub(d, 1.0f) > 1e-6f) {
      fail(0, 1e9f);
      return;
    }
  }
  // rms over the three pairs, divided by pairs.size() = 4 (base.cc:1587-1599)
  float rms = 0.f;
  V3 pp[3] = {p0, p1, p2}, qq[3] = {q0, q1, q2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    V3 first = vsub({mul(1.0f, qq[k].x), mul(1.0f, qq[k].y), mul(1.0f, qq[k].z)}, c2);
    V3 tr = {sum3(mul(R[0][0], first.x), mul(R[0][1], first.y), mul(R[0][2], first.z)),
             sum3(mul(R[1][0], first.x), mul(R[1][1], first.y), mul(R[1][2], first.z)),
             sum3(mul(R[2][0], first.x), mul(R[2][1], first.y), mul(R[2][2], first.z))};
    V3 e = {add(sub(tr.x, pp[k].x), c1.x), add(sub(tr.y, pp[k].y), c1.y), add(sub(tr.z, pp[k].z), c1.z)};
    rms = add(rms, sqrt_rn(sqnorm(e)));
  }
  rms = __fdiv_rn(rms, 4.0f);
  if (!(rms >= 0.f)) {  // base.cc:1467 `ok && rms >= 0`
    fail(0, rms);
    return;
  }
  // etrans = translate(c1) * rotate(R) * translate(-c2) (base.cc:1601-1611)
  float t[3], tw[3];
  V3 u = {add(c2.x, cQx), add(c2.y, cQy), add(c2.z, cQz)};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float s = sum3(mul(R[r][0], -c2.x), mul(R[r][1], -c2.y), mul(R[r][2], -c2.z));
    t[r] = add(comp(c1, r), s);
    // de-centring (base.cc:1474-1482): (c1 + cP) - L (c2 + cQ)
    float sw = sum3(mul(R[r][0], u.x), mul(R[r][1], u.y), mul(R[r][2], u.z));
    float cp = r == 0 ? cPx : (r == 1 ? cPy : cPz);
    tw[r] = sub(add(comp(c1, r), cp), sw);
  }
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) {
#pragma unroll
    for (int r = 0; r < 3; ++r) Ti[4 * cc + r] = R[r][cc];
    Ti[4 * cc + 3] = 0.f;
  }
  Ti[12] = t[0];
  Ti[13] = t[1];
  Ti[14] = t[2];
  Ti[15] = 1.f;
  if (Pi) {
    for (int k = 0; k < 12; ++k) Pi[k] = (double)Ti[k];
    Pi[12] = (double)tw[0];
    Pi[13] = (double)tw[1];
    Pi[14] = (double)tw[2];
    Pi[15] = 1.0;
  }
  status[i] = 1;
  if (rms_out) rms_out[i] = rms;
}

}  // namespace

int launch_rigid(pgp_ctx* ctx, const int* d_base_ids, const int* d_quad_ids, int n,
                 const float cP[3], const float cQ[3], float* d_T, double* d_pose, int* d_status,
                 float* d_rms, hipStream_t stream) {
  if (ctx->nP <= 0 || !ctx->d_P.p) {
    set_error("no scene: call pgp_set_scene first");
    return PGP_ESTATE;
  }
  if (ctx->nQs <= 0 || !ctx->d_Qs.p) {
    set_error("no search model: call pgp_set_search_model first");
    return PGP_ESTATE;
  }
  if (n <= 0) return PGP_OK;
  hipLaunchKernelGGL(rigid_from_congruent, dim3((n + 127) / 128), dim3(128), 0, stream,
                     ctx->d_P.as<float4>(), ctx->nP, ctx->d_Qs.as<float4>(), ctx->nQs,
                     reinterpret_cast<const int4*>(d_base_ids), reinterpret_cast<const int4*>(d_quad_ids), n,
                     cP[0], cP[1], cP[2], cQ[0], cQ[1], cQ[2], d_T, d_pose, d_status, d_rms);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

}  // namespace pgp
