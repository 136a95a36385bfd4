// csrc/render.hip -- depth images of posed objects, rendered on the device for the MCTS leaf states.
//
// Replaces the OpenGL pass the search reaches through UCTState::render
// (PPE/hypothesis_verification/mcts/UCTState.cpp:44-72 -> src/3rdparty/depth_sim/src/renderScene.cpp:45-72,
// pcl::simulation::RangeLikelihood): the state's newest object is drawn with the camera's intrinsics,
// depths beyond 1 m are dropped (renderScene.cpp:69 `depth_image.setTo(0, depth_image > 1)`), and the
// result is laid over the parent state's image wherever it is nearer (UCTState.cpp:62-68).  The images
// stay in HBM: pgp_depth_cost_device (depth_cost.hip = UCTState::computeCost, :93-116) reads them there,
// so a batch of leaf states is rendered and costed without one image crossing PCIe.
//
// OpenGL's rasteriser is not specified to the bit (fill rule on shared edges, depth-buffer quantisation,
// driver), so there is nothing to be bit-identical WITH; this file states its own rules and
// tests/_checkers.py restates them in numpy float32, operation for operation:
//   camera point   c = R v + t           rows as ((r0 x + r1 y) + r2 z) + t, float
//   pixel          px = (fx c.x) / c.z + cx,  py = (fy c.y) / c.z + cy;  dropped unless c.z > z_near
//   splat (no triangles): the pixel (rint(px), rint(py)) takes min(depth, c.z)
//   triangle: every pixel centre (i + 0.5, j + 0.5) inside or on the edge of the projected triangle
//             (edge function E(a, b, p) = (b.x - a.x)(p.y - a.y) - (b.y - a.y)(p.x - a.x), all three of
//             the sign of the area or zero) takes min(depth, z) with the perspective-correct
//             z = area / (E0 / z0 + E1 / z1 + E2 / z2)
//   near plane: a triangle with one or two vertices at c.z <= z_near is CLIPPED against z = z_clip (z_near, or 1e-4
//             when z_near is 0) in camera space (Sutherland-Hodgman on the one plane; OpenGL clips too: a table or a
//             wall that runs past the camera must not vanish): with A in front and B behind,
//                 t = (z_clip - A.z) / (B.z - A.z),  P = {A.x + t (B.x - A.x), A.y + t (B.y - A.y), z_clip},
//             one vertex in front (A; B, C behind, cyclic order kept): triangle (A, AB, AC); two in front (A, B; C
//             behind): triangles (A, B, BC) and (A, BC, AC); the pieces are projected and filled like any triangle.
//   large triangles: a triangle whose pixel box exceeds 4096 pixels is filled by a whole workgroup (second launch,
//             256 threads striding over the box) instead of one thread -- same fragments, same image.
//   finally        a fragment is kept if z <= z_max; untouched pixels keep the parent's value, else 0.
// Depth is a positive float, so its bit pattern orders like its value and ONE 32-bit atomic min per
// fragment is the whole z-buffer: the image does not depend on the order of the fragments.
// Memory-bound by design: 16 B per projected vertex, 4 B per fragment; the per-pose vertex transform
// runs once per vertex (not once per triangle corner).

#include "pgp_internal.h"

namespace pgp {

namespace {

constexpr uint32_t kInfBits = 0x7F800000u;

struct RenderArgs {
  const float* verts;    // n_vert x stride floats
  int stride, n_vert;
  const int* tris;       // n_tri x 3, nullable
  int n_tri;
  const float* T;        // n x 16 column-major: model -> camera
  int n;
  int rows, cols;
  float fx, fy, cx, cy, z_near, z_max;
  const float* parent;   // nullable
  size_t parent_stride;  // 0: one parent for every image
  float4* proj;          // [n][n_vert] {px, py, z, valid}
  uint32_t* out;         // [n][rows * cols] depth bits
  float z_clip;          // the clipping plane (z_near, or 1e-4 when z_near is 0)
  // triangles (and clipped pieces) with a large pixel box: filled by render_big, one workgroup each
  float4* big;           // [big_cap][3] projected vertices {px, py, z, image}
  unsigned* big_count;
  unsigned big_cap;
};
constexpr int kBigPixels = 4096;

__device__ __forceinline__ float row3(float a, float b, float c, float t, float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, x), __fmul_rn(b, y)), __fmul_rn(c, z)), t);
}

__global__ __launch_bounds__(256) void render_init(RenderArgs a) {
  const size_t n_pix = (size_t)a.rows * a.cols;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pix) return;
  const int img = blockIdx.y;
  uint32_t v = kInfBits;
  if (a.parent) {
    const float p = a.parent[(size_t)img * a.parent_stride + i];
    if (p > 0.f) v = __float_as_uint(p);
  }
  a.out[(size_t)img * n_pix + i] = v;
}

__global__ __launch_bounds__(256) void render_project(RenderArgs a) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= a.n_vert) return;
  const int img = blockIdx.y;
  const float* G = a.T + 16 * (size_t)img;
  const float* p = a.verts + (size_t)v * a.stride;
  const float x = row3(G[0], G[4], G[8], G[12], p[0], p[1], p[2]);
  const float y = row3(G[1], G[5], G[9], G[13], p[0], p[1], p[2]);
  const float z = row3(G[2], G[6], G[10], G[14], p[0], p[1], p[2]);
  float4 o = make_float4(0.f, 0.f, z, 0.f);
  if (z > a.z_near) {   // false for NaN
    o.x = __fadd_rn(__fdiv_rn(__fmul_rn(a.fx, x), z), a.cx);
    o.y = __fadd_rn(__fdiv_rn(__fmul_rn(a.fy, y), z), a.cy);
    o.w = 1.f;
  }
  a.proj[(size_t)img * a.n_vert + v] = o;
}

__global__ __launch_bounds__(256) void render_splat(RenderArgs a) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= a.n_vert) return;
  const int img = blockIdx.y;
  const float4 p = a.proj[(size_t)img * a.n_vert + v];
  if (p.w == 0.f || !(p.z <= a.z_max)) return;
  const float fu = rintf(p.x), fv = rintf(p.y);
  if (!(fu >= 0.f && fu < (float)a.cols && fv >= 0.f && fv < (float)a.rows)) return;
  atomicMin(&a.out[((size_t)img * a.rows + (int)fv) * a.cols + (int)fu], __float_as_uint(p.z));
}

__device__ __forceinline__ float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
  return __fsub_rn(__fmul_rn(__fsub_rn(bx, ax), __fsub_rn(py, ay)), __fmul_rn(__fsub_rn(by, ay), __fsub_rn(px, ax)));
}

// pixel box of a projected triangle; false: nothing to fill
struct TriBox {
  int x0, x1, y0, y1;
  float area, sgn;
};
__device__ __forceinline__ bool tri_box(const RenderArgs& a, const float4 v0, const float4 v1, const float4 v2, TriBox* b) {
  b->area = edge_fn(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
  if (!(b->area != 0.f)) return false;   // degenerate (or NaN)
  b->sgn = b->area > 0.f ? 1.f : -1.f;
  // pixel centres (i + 0.5) inside [min, max]:  i >= min - 0.5, i <= max - 0.5
  const float minx = fminf(v0.x, fminf(v1.x, v2.x)), maxx = fmaxf(v0.x, fmaxf(v1.x, v2.x));
  const float miny = fminf(v0.y, fminf(v1.y, v2.y)), maxy = fmaxf(v0.y, fmaxf(v1.y, v2.y));
  if (!(maxx >= 0.f && maxy >= 0.f && minx <= (float)a.cols && miny <= (float)a.rows)) return false;
  b->x0 = (int)fmaxf(ceilf(__fsub_rn(minx, 0.5f)), 0.f);
  b->x1 = (int)fminf(floorf(__fsub_rn(maxx, 0.5f)), (float)(a.cols - 1));
  b->y0 = (int)fmaxf(ceilf(__fsub_rn(miny, 0.5f)), 0.f);
  b->y1 = (int)fminf(floorf(__fsub_rn(maxy, 0.5f)), (float)(a.rows - 1));
  return b->x0 <= b->x1 && b->y0 <= b->y1;
}
__device__ __forceinline__ void tri_pixel(const RenderArgs& a, const float4 v0, const float4 v1, const float4 v2, const TriBox& b,
                                          int px, int py, uint32_t* out) {
  const float cx = __fadd_rn((float)px, 0.5f), cy = __fadd_rn((float)py, 0.5f);
  const float e0 = __fmul_rn(b.sgn, edge_fn(v1.x, v1.y, v2.x, v2.y, cx, cy));
  const float e1 = __fmul_rn(b.sgn, edge_fn(v2.x, v2.y, v0.x, v0.y, cx, cy));
  const float e2 = __fmul_rn(b.sgn, edge_fn(v0.x, v0.y, v1.x, v1.y, cx, cy));
  if (!(e0 >= 0.f && e1 >= 0.f && e2 >= 0.f)) return;
  const float den = __fadd_rn(__fadd_rn(__fdiv_rn(e0, v0.z), __fdiv_rn(e1, v1.z)), __fdiv_rn(e2, v2.z));
  const float z = __fdiv_rn(__fmul_rn(b.sgn, b.area), den);
  if (!(z > a.z_near && z <= a.z_max)) return;   // also drops NaN / inf (a pixel exactly on all three edges)
  atomicMin(&out[(size_t)py * a.cols + px], __float_as_uint(z));
}
// one projected triangle: filled by this thread, or queued for a workgroup when its pixel box is large
__device__ __forceinline__ void tri_emit(const RenderArgs& a, int img, const float4 v0, const float4 v1, const float4 v2) {
  TriBox b;
  if (!tri_box(a, v0, v1, v2, &b)) return;
  if ((long long)(b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1) > kBigPixels && a.big) {
    const unsigned slot = atomicAdd(a.big_count, 1u);
    if (slot < a.big_cap) {
      a.big[3 * (size_t)slot] = make_float4(v0.x, v0.y, v0.z, __int_as_float(img));
      a.big[3 * (size_t)slot + 1] = v1;
      a.big[3 * (size_t)slot + 2] = v2;
      return;
    }   // (a full queue: this thread fills the triangle itself, slowly but correctly)
  }
  uint32_t* out = a.out + (size_t)img * a.rows * a.cols;
  for (int py = b.y0; py <= b.y1; ++py)
    for (int px = b.x0; px <= b.x1; ++px) tri_pixel(a, v0, v1, v2, b, px, py, out);
}

// camera-space point of vertex v (the operations of render_project)
__device__ __forceinline__ float4 cam_point(const RenderArgs& a, int img, int v) {
  const float* G = a.T + 16 * (size_t)img;
  const float* p = a.verts + (size_t)v * a.stride;
  return make_float4(row3(G[0], G[4], G[8], G[12], p[0], p[1], p[2]), row3(G[1], G[5], G[9], G[13], p[0], p[1], p[2]),
                     row3(G[2], G[6], G[10], G[14], p[0], p[1], p[2]), 0.f);
}
// the point of the edge A (in front) -> B (behind) on the clipping plane, projected
__device__ __forceinline__ float4 clip_edge(const RenderArgs& a, const float4 A, const float4 B) {
  const float t = __fdiv_rn(__fsub_rn(a.z_clip, A.z), __fsub_rn(B.z, A.z));
  const float x = __fadd_rn(A.x, __fmul_rn(t, __fsub_rn(B.x, A.x)));
  const float y = __fadd_rn(A.y, __fmul_rn(t, __fsub_rn(B.y, A.y)));
  return make_float4(__fadd_rn(__fdiv_rn(__fmul_rn(a.fx, x), a.z_clip), a.cx), __fadd_rn(__fdiv_rn(__fmul_rn(a.fy, y), a.z_clip), a.cy),
                     a.z_clip, 1.f);
}

__global__ __launch_bounds__(256) void render_tris(RenderArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n_tri) return;
  const int img = blockIdx.y;
  const int i0 = a.tris[3 * (size_t)t], i1 = a.tris[3 * (size_t)t + 1], i2 = a.tris[3 * (size_t)t + 2];
  if ((unsigned)i0 >= (unsigned)a.n_vert || (unsigned)i1 >= (unsigned)a.n_vert || (unsigned)i2 >= (unsigned)a.n_vert) return;
  const float4* P = a.proj + (size_t)img * a.n_vert;
  const float4 v0 = P[i0], v1 = P[i1], v2 = P[i2];
  const int front = (v0.w != 0.f ? 1 : 0) + (v1.w != 0.f ? 1 : 0) + (v2.w != 0.f ? 1 : 0);
  if (front == 3) {
    tri_emit(a, img, v0, v1, v2);
    return;
  }
  if (front == 0) return;
  // the near plane cuts the triangle: rotate so that A is in front and the vertex before it is not, then clip
  int ia = i0, ib = i1, ic = i2;
  float4 pa = v0, pb = v1, pc = v2;
  for (int k = 0; k < 2 && !(pa.w != 0.f && pc.w == 0.f); ++k) {
    const int ti = ia;
    ia = ib; ib = ic; ic = ti;
    const float4 tp = pa;
    pa = pb; pb = pc; pc = tp;
  }
  const float4 A = cam_point(a, img, ia), B = cam_point(a, img, ib), C = cam_point(a, img, ic);
  if (!(A.z == A.z && B.z == B.z && C.z == C.z)) return;   // a NaN vertex: nothing to draw
  if (front == 1) {          // A in front; B, C behind
    tri_emit(a, img, pa, clip_edge(a, A, B), clip_edge(a, A, C));
  } else {                   // A, B in front; C behind
    const float4 bc = clip_edge(a, B, C), ac = clip_edge(a, A, C);
    tri_emit(a, img, pa, pb, bc);
    tri_emit(a, img, pa, bc, ac);
  }
}

// the queued large triangles: a workgroup each, its 256 threads striding over the pixel box
__global__ __launch_bounds__(256) void render_big(RenderArgs a) {
  const unsigned n = min(*a.big_count, a.big_cap);
  for (unsigned e = blockIdx.x; e < n; e += gridDim.x) {
    float4 v0 = a.big[3 * (size_t)e];
    const float4 v1 = a.big[3 * (size_t)e + 1], v2 = a.big[3 * (size_t)e + 2];
    const int img = __float_as_int(v0.w);
    v0.w = 1.f;
    TriBox b;
    if (!tri_box(a, v0, v1, v2, &b)) continue;
    uint32_t* out = a.out + (size_t)img * a.rows * a.cols;
    const int w = b.x1 - b.x0 + 1;
    const long long total = (long long)w * (b.y1 - b.y0 + 1);
    for (long long k = threadIdx.x; k < total; k += blockDim.x) tri_pixel(a, v0, v1, v2, b, b.x0 + (int)(k % w), b.y0 + (int)(k / w), out);
  }
}

__global__ __launch_bounds__(256) void render_finish(uint32_t* __restrict__ out, size_t n_total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total) return;
  if (out[i] == kInfBits) out[i] = 0u;   // nothing drawn here and no parent: depth 0 = no surface
}

__global__ __launch_bounds__(256) void cost_scores(const int* __restrict__ counts, int n, float* __restrict__ scores) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // renderScore = obScore + renScore - intScore in float (UCTState.cpp:115); the tallies are < 2^24
  scores[i] = __fsub_rn(__fadd_rn((float)counts[3 * i], (float)counts[3 * i + 1]), (float)counts[3 * i + 2]);
}

}  // namespace

int launch_render_depth(pgp_ctx* ctx, const float* d_verts, int stride, int n_vert, const int* d_tris, int n_tri,
                        const float* d_T, int n, const pgp_camera* cam, const float* d_parent, size_t parent_stride,
                        float* d_depth, hipStream_t st) {
  if (n <= 0) return PGP_OK;
  if (cam->rows <= 0 || cam->cols <= 0 || (size_t)cam->rows * cam->cols > (size_t)1 << 28) {
    set_error("render: bad image size %d x %d", cam->rows, cam->cols);
    return PGP_EINVAL;
  }
  if (n > 65535) {
    set_error("render: at most 65535 images per call (%d)", n);
    return PGP_EINVAL;
  }
  if (stride != 3 && stride != 4) {
    set_error("render: vertex stride must be 3 or 4 floats");
    return PGP_EINVAL;
  }
  RenderArgs a{};
  a.verts = d_verts;
  a.stride = stride;
  a.n_vert = n_vert;
  a.tris = d_tris;
  a.n_tri = d_tris ? n_tri : 0;
  a.T = d_T;
  a.n = n;
  a.rows = cam->rows;
  a.cols = cam->cols;
  a.fx = cam->fx;
  a.fy = cam->fy;
  a.cx = cam->cx;
  a.cy = cam->cy;
  a.z_near = cam->z_near > 0.f ? cam->z_near : 0.f;
  a.z_clip = a.z_near > 0.f ? a.z_near : 1e-4f;
  a.z_max = cam->z_max > 0.f ? cam->z_max : 3.0e38f;
  a.parent = d_parent;
  a.parent_stride = parent_stride;
  a.out = reinterpret_cast<uint32_t*>(d_depth);
  const size_t n_pix = (size_t)a.rows * a.cols;
  int rc;
  const size_t proj_bytes = ((size_t)n * (size_t)(n_vert > 0 ? n_vert : 1) * 16 + 255) & ~(size_t)255;
  a.big_cap = d_tris ? 16384u : 0u;
  if ((rc = ctx->d_render_ws.ensure(proj_bytes + (size_t)a.big_cap * 48 + 256)) != PGP_OK) return rc;
  a.proj = ctx->d_render_ws.as<float4>();
  a.big_count = reinterpret_cast<unsigned*>(ctx->d_render_ws.as<unsigned char>() + proj_bytes);
  a.big = a.big_cap ? reinterpret_cast<float4*>(ctx->d_render_ws.as<unsigned char>() + proj_bytes + 256) : nullptr;
  hipLaunchKernelGGL(render_init, dim3((unsigned)((n_pix + 255) / 256), n), dim3(256), 0, st, a);
  if (n_vert > 0) {
    const dim3 gv((n_vert + 255) / 256, n);
    hipLaunchKernelGGL(render_project, gv, dim3(256), 0, st, a);
    if (a.tris) {
      if (a.n_tri > 0) {
        PGP_HIP(hipMemsetAsync(a.big_count, 0, 4, st));
        hipLaunchKernelGGL(render_tris, dim3((a.n_tri + 255) / 256, n), dim3(256), 0, st, a);
        hipLaunchKernelGGL(render_big, dim3(1024), dim3(256), 0, st, a);   // (returns at once when nothing was queued)
      }
    } else {
      hipLaunchKernelGGL(render_splat, gv, dim3(256), 0, st, a);
    }
  }
  hipLaunchKernelGGL(render_finish, dim3((unsigned)((n_pix * n + 255) / 256)), dim3(256), 0, st, a.out, n_pix * (size_t)n);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

int launch_cost_scores(const int* d_counts, int n, float* d_scores, hipStream_t st) {
  if (n <= 0) return PGP_OK;
  hipLaunchKernelGGL(cost_scores, dim3((n + 255) / 256), dim3(256), 0, st, d_counts, n, d_scores);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

}  // namespace pgp
