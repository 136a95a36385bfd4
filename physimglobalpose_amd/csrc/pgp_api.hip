// csrc/pgp_api.hip -- the C ABI of include/pgp.h: context, uploads, host-pointer entry points.
// No CPU fallback anywhere: every scoring entry point ends in a HIP kernel launch or an error.

#include "pgp_internal.h"

#include <chrono>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

namespace pgp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

int DevBuf::ensure(size_t bytes) {
  if (bytes <= cap && p) return PGP_OK;
  if (bytes == 0) bytes = 16;
  if (p) {
    hipError_t e = hipFree(p);
    (void)e;
    p = nullptr;
    cap = 0;
  }
  // grow geometrically so repeated calls with slowly growing sizes do not reallocate each time
  size_t want = bytes + bytes / 4;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    p = nullptr;
    set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? PGP_ENOMEM : PGP_EHIP;
  }
  cap = want;
  return PGP_OK;
}

void DevBuf::release() {
  if (p) {
    hipError_t e = hipFree(p);
    (void)e;
  }
  p = nullptr;
  cap = 0;
}

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) {
      hipError_t e = hipSetDevice(prev);
      (void)e;
    }
  }
};

// Entry-point guard of a context: selects its device and orders this call behind whatever an earlier
// *_device call queued on the CALLER's stream -- those calls return without synchronising while the
// kernels they launched still read the context's arrays (clouds, index, workspaces), and the host-pointer
// entry points rewrite those arrays on the context's own non-blocking stream.  The *_device call only sets
// a flag; the next host-pointer call (synchronous by contract anyway) drains the device first.  Two things
// that were tried and dropped: an event recorded after every *_device call cost the scoring step 3 us (a
// barrier packet between back-to-back steps); recording it lazily here, on the caller's stream, is unsafe --
// the caller may have destroyed that stream, and HIP dereferences the stale handle.
struct CtxGuard : DeviceGuard {
  // join = false for the *_device entry points themselves: they queue on the caller's stream, behind whatever
  // the caller queued there before, and must not synchronise anything while the caller may be capturing a graph
  // (A call that returned on a completion WORD in host memory -- pgp_score_lcp, pgp::publish_and_wait -- leaves a kernel that
  //  has made its last memory access and is merely retiring: nothing to order against.)
  explicit CtxGuard(pgp_ctx* ctx, bool join = true) : DeviceGuard(ctx->device) {
    if (join && ok && ctx->device_work_pending) {
      (void)hipDeviceSynchronize();
      ctx->device_work_pending = false;
    }
  }
};

// after a *_device entry point has queued work on `stream`
inline void note_device_work(pgp_ctx* ctx, hipStream_t stream) {
  if (stream == ctx->stream) return;
  // a stream that is being captured into a graph runs nothing now; whoever replays the graph orders the
  // replays against later calls on the context
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return;
  ctx->device_work_pending = true;
}

// after the host has synchronised a stream that was made to wait for the side-stream index build (await_index):
// the build is through, so its flag is cleared and its counts are taken over here -- a *_device call that follows
// (possibly on a capturing stream, which can neither query nor wait for the build's event) then finds nothing pending
inline int index_settled(pgp_ctx* ctx) { return ctx->index_pending ? finish_index(ctx) : PGP_OK; }

inline uint32_t spread10(uint32_t v) {
  v &= 1023u;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

}  // namespace
}  // namespace pgp

using namespace pgp;

extern "C" {

int pgp_version(void) { return 200; }

const char* pgp_last_error(void) { return g_err; }

int pgp_create(pgp_ctx** out, int device_id) {
  if (!out) {
    set_error("pgp_create: out is NULL");
    return PGP_EINVAL;
  }
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device available (%s); libpgp has no CPU fallback",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return PGP_ENODEV;
  }
  if (device_id < 0) {
    if (hipGetDevice(&device_id) != hipSuccess) device_id = 0;
  }
  if (device_id >= n) {
    set_error("device %d out of range (%d devices)", device_id, n);
    return PGP_EINVAL;
  }
  DeviceGuard guard(device_id);
  if (!guard.ok) {
    set_error("hipSetDevice(%d) failed", device_id);
    return PGP_ENODEV;
  }
  pgp_ctx* ctx = new pgp_ctx();
  ctx->device = device_id;
  {
    int cus = 0, coop = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0 &&
        hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device_id) == hipSuccess && coop)
      ctx->n_cus = cus;   // 0: the kernels whose workgroups wait for each other stay off (icp.hip launch_resident)
  }
  if (const char* v = getenv("PGP_UNROLL")) ctx->unroll = atoi(v);
  if (const char* v = getenv("PGP_HPB")) ctx->hpb_override = atoi(v);
  if (const char* v = getenv("PGP_REFINE")) ctx->refine_best = atoi(v) != 0;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete ctx;
    return PGP_ENODEV;
  }
  int rc = ctx->d_best.ensure(32);
  if (rc == PGP_OK && hipMemset(ctx->d_best.p, 0, 32) != hipSuccess) rc = PGP_EHIP;  // key + ticket armed
  if (rc != PGP_OK) {
    (void)hipStreamDestroy(ctx->stream);
    ctx->d_best.release();
    delete ctx;
    return rc;
  }
  *out = ctx;
  return PGP_OK;
}

int pgp_destroy(pgp_ctx* ctx) {
  if (!ctx) return PGP_OK;
  CtxGuard guard(ctx);
  ctx->deferred_build = nullptr;   // (a build that was never queued)
  if (ctx->stream) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    (void)e;
  }
  {
    hipError_t e = hipSuccess;
    if (ctx->build_stream) {
      e = hipStreamSynchronize(ctx->build_stream);
      if (ctx->build_stream_own) e = hipStreamDestroy(ctx->build_stream);   // (else the device's shared one: it stays)
    }
    if (ctx->ev_index) e = hipEventDestroy(ctx->ev_index);
    if (ctx->ev_build0) e = hipEventDestroy(ctx->ev_build0);
    if (ctx->h_build_counts) e = hipHostFree(ctx->h_build_counts);
    (void)e;
    ctx->build_stream = nullptr;
    ctx->ev_index = ctx->ev_build0 = nullptr;
    ctx->h_build_counts = nullptr;
    ctx->index_pending = false;
  }
  DevBuf* bufs[] = {&ctx->d_P, &ctx->d_Pnw, &ctx->d_cell_start, &ctx->d_cell_tmp, &ctx->d_scan_tmp, &ctx->d_build_scan,
                    &ctx->d_bitmap, &ctx->d_blocktab, &ctx->d_kd_nodes, &ctx->d_kd_pts, &ctx->d_occ_start, &ctx->d_cand, &ctx->d_Q, &ctx->d_Qn, &ctx->d_Qpos, &ctx->d_eo_ws, &ctx->d_T, &ctx->d_partial,
                    &ctx->d_scores, &ctx->d_counts, &ctx->d_best, &ctx->d_rec_ws, &ctx->d_hits, &ctx->d_seq, &ctx->d_Qs, &ctx->d_ids,
                    &ctx->d_rig, &ctx->d_icp_src, &ctx->d_icp_tgt, &ctx->d_icp_tgt_n, &ctx->d_icp_grid, &ctx->d_icp_T, &ctx->d_icp_out, &ctx->d_icp_ws, &ctx->d_icp_x, &ctx->d_Qs_unit, &ctx->d_cs_cnt, &ctx->d_cs_entries, &ctx->d_cs_keys,
                    &ctx->d_cs_pairs, &ctx->d_cs_out, &ctx->d_ppf_keys, &ctx->d_ppf_val, &ctx->d_ppf_off, &ctx->d_ppf_pairs, &ctx->d_prob_cdf, &ctx->d_sel_ws, &ctx->d_csb, &ctx->d_csb_picks, &ctx->d_pre_ws, &ctx->d_vg_ws, &ctx->d_mls_ws, &ctx->d_pre_io, &ctx->d_depth, &ctx->d_render_ws, &ctx->d_render_io, &ctx->d_cl_keys, &ctx->d_cl_ws, &ctx->d_cl_io, &ctx->d_bp, &ctx->d_top_ws, &ctx->d_acc, &ctx->d_pub_ticket};
  for (DevBuf* b : bufs) b->release();
  ctx->d_out.release();
  if (ctx->h_pin) {
    hipError_t e = hipHostFree(ctx->h_pin);
    (void)e;
    ctx->h_pin = nullptr;
  }
  if (ctx->h_sel_pin) {
    hipError_t e = hipHostFree(ctx->h_sel_pin);
    (void)e;
    ctx->h_sel_pin = nullptr;
  }
  if (ctx->h_flag) {
    hipError_t e = hipHostFree(ctx->h_flag);
    (void)e;
    ctx->h_flag = nullptr;
  }
  if (ctx->h_out) {
    hipError_t e = hipHostFree(ctx->h_out);
    (void)e;
    ctx->h_out = nullptr;
  }
  if (ctx->h_w_pin) {
    hipError_t e = hipHostFree(ctx->h_w_pin);
    (void)e;
    ctx->h_w_pin = nullptr;
  }
  if (ctx->h_s_pin) {
    hipError_t e = hipHostFree(ctx->h_s_pin);
    (void)e;
    ctx->h_s_pin = nullptr;
  }
  if (ctx->ev_s) {
    hipError_t e = hipEventDestroy(ctx->ev_s);
    (void)e;
    ctx->ev_s = nullptr;
  }
  if (ctx->ev_w) {
    hipError_t e = hipEventDestroy(ctx->ev_w);
    (void)e;
    ctx->ev_w = nullptr;
  }
  for (hipEvent_t e : ctx->ev) {
    hipError_t r = hipEventDestroy(e);
    (void)r;
  }
  if (ctx->stream) {
    hipError_t e = hipStreamDestroy(ctx->stream);
    (void)e;
  }
  delete ctx;
  return PGP_OK;
}

int pgp_center(float* P, int nP, float* Qs, int nQs, float* Qv, int nQv, float centroid_P[3],
               float centroid_Q[3]) {
  if (nP < 0 || nQs < 0 || nQv < 0 || (nP && !P) || (nQs && !Qs) || (nQv && !Qv) || !centroid_P ||
      !centroid_Q) {
    set_error("pgp_center: bad argument");
    return PGP_EINVAL;
  }
  // base.cc:242-249: centroid += pos (index order, float), then /= Scalar(n)
  float cP[3] = {0.f, 0.f, 0.f}, cQ[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < nP; ++i)
    for (int k = 0; k < 3; ++k) cP[k] += P[3 * (size_t)i + k];
  for (int k = 0; k < 3; ++k) cP[k] /= (float)nP;
  for (int i = 0; i < nQs; ++i)
    for (int k = 0; k < 3; ++k) cQ[k] += Qs[3 * (size_t)i + k];
  for (int k = 0; k < 3; ++k) cQ[k] /= (float)nQs;
  // base.cc:256-264
  for (int i = 0; i < nP; ++i)
    for (int k = 0; k < 3; ++k) P[3 * (size_t)i + k] -= cP[k];
  for (int i = 0; i < nQs; ++i)
    for (int k = 0; k < 3; ++k) Qs[3 * (size_t)i + k] -= cQ[k];
  for (int i = 0; i < nQv; ++i)
    for (int k = 0; k < 3; ++k) Qv[3 * (size_t)i + k] -= cQ[k];
  for (int k = 0; k < 3; ++k) {
    centroid_P[k] = cP[k];
    centroid_Q[k] = cQ[k];
  }
  return PGP_OK;
}

namespace {
// the pixel of one scene point as Match4PCSBase::init computes it (base.cc:317-340); false = outside the image
inline bool image_pixel(const float* p, const float cP[3], const float K[9], int rows, int cols, int* row, int* col) {
  // b_ii.pos() += centroid_P_ (float), then double x1,y1,z1 -> Eigen::Vector3f(x1,y1,z1) (back to float)
  const float x = p[0] + cP[0], y = p[1] + cP[1], z = p[2] + cP[2];
  // camIntrinsic * v: Eigen evaluates a 3x3 * 3x1 float product as k0*x + (k1*y + k2*z) per row
  float u[3];
  for (int r = 0; r < 3; ++r) {
    float a = K[3 * r] * x, b = K[3 * r + 1] * y, c = K[3 * r + 2] * z;
    float bc = b + c;
    u[r] = a + bc;
  }
  const float fc = u[0] / u[2], fr = u[1] / u[2];
  // int col = point2D[0]/point2D[2]: truncation toward zero; guard what the reference leaves undefined
  const bool ok = fc == fc && fr == fr && fc > -1.f && fr > -1.f && fc < (float)cols && fr < (float)rows;
  if (!ok) return false;
  *col = (int)fc;
  *row = (int)fr;
  return *col >= 0 && *row >= 0 && *col < cols && *row < rows;
}
}  // namespace

int pgp_weights_from_image(const float* P, int n, const float cP[3], const float K[9],
                           const unsigned short* img, int rows, int cols, float* weights) {
  if (n < 0 || rows < 0 || cols < 0 || (n > 0 && (!P || !weights)) || !cP || !K || (rows * cols > 0 && !img)) {
    set_error("pgp_weights_from_image: bad argument");
    return PGP_EINVAL;
  }
  for (int i = 0; i < n; ++i) {
    int row = 0, col = 0;
    weights[i] = image_pixel(P + 3 * (size_t)i, cP, K, rows, cols, &row, &col) ? (float)img[(size_t)row * cols + col] / 10000 : 0.f;
  }
  return PGP_OK;
}

int pgp_image_rows_needed(const float* P, int n, const float cP[3], const float K[9], int rows, int cols, int* row_min,
                          int* row_max) {
  if (n < 0 || rows < 0 || cols < 0 || (n > 0 && !P) || !cP || !K || !row_min || !row_max) {
    set_error("pgp_image_rows_needed: bad argument");
    return PGP_EINVAL;
  }
  int lo = -1, hi = -1;
  for (int i = 0; i < n; ++i) {
    int row = 0, col = 0;
    if (!image_pixel(P + 3 * (size_t)i, cP, K, rows, cols, &row, &col)) continue;
    lo = lo < 0 || row < lo ? row : lo;
    hi = row > hi ? row : hi;
  }
  *row_min = lo;
  *row_max = hi;
  return PGP_OK;
}

int pgp_set_scene(pgp_ctx* ctx, const float* xyz, const float* nrm, const float* weight, int n,
                  float delta) {
  if (!ctx || n < 0 || (n > 0 && !xyz) || !(delta > 0.f) || !std::isfinite(delta)) {
    set_error("pgp_set_scene: bad argument (n=%d, delta=%g)", n, (double)delta);
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);   // (*_device calls still queued on a caller's stream read the arrays replaced below: the guard drains them)
  // What else can still touch those arrays: the PREVIOUS scene's index build on the side stream (it reads the points) -- the
  // uploads below are ordered behind it on the context's stream, where everything else that rewrites the scene is queued
  // anyway.  (Up to round 5 an unconditional hipDeviceSynchronize stood here; behind a call that returned on its completion
  // word -- the drop-in's last step -- it waited ~30 us for a kernel that had long made its last access to retire.)
  static const bool sync_always = getenv("PGP_SET_SCENE_SYNC") && atoi(getenv("PGP_SET_SCENE_SYNC")) != 0;
  if (sync_always) PGP_HIP(hipDeviceSynchronize());
  else if (ctx->index_pending && ctx->ev_index && !ctx->deferred_build) PGP_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_index, 0));
  ctx->has_index = false;
  ctx->prob_cdf_valid = false;
  ctx->nP = n;
  ctx->has_scene_normals = nrm != nullptr;
  // packed {x, y, z, id} / {nx, ny, nz, w}: through a pinned staging buffer of this call's own for scenes up to 4 MB (the
  // uploads are then queued and nobody waits for them here: an event tells the side-stream index build when they are
  // through), through temporaries and a synchronisation beyond
  const size_t N = (size_t)std::max(n, 1), bytes = N * sizeof(float4);
  const bool staged = 2 * bytes <= ((size_t)4 << 20);
  std::vector<float4> tmp;
  float4 *hp, *hn;
  int rc;
  if (staged) {
    if (ctx->ev_s) PGP_HIP(hipEventSynchronize(ctx->ev_s));   // the previous scene's copies out of the staging buffer
    else PGP_HIP(hipEventCreateWithFlags(&ctx->ev_s, hipEventDisableTiming));
    if (2 * bytes > ctx->h_s_cap) {
      if (ctx->h_s_pin) {
        hipError_t e = hipHostFree(ctx->h_s_pin);
        (void)e;
        ctx->h_s_pin = nullptr;
        ctx->h_s_cap = 0;
      }
      PGP_HIP(hipHostMalloc(&ctx->h_s_pin, 2 * bytes + bytes / 2, hipHostMallocDefault));
      ctx->h_s_cap = 2 * bytes + bytes / 2;
    }
    hp = static_cast<float4*>(ctx->h_s_pin);
    hn = hp + N;
  } else {
    tmp.resize(2 * N);
    hp = tmp.data();
    hn = hp + N;
  }
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + 3 * (size_t)i;
    hp[i] = make_float4(p[0], p[1], p[2], __builtin_bit_cast(float, i));
    float w = weight ? weight[i] : 1.0f;
    if (nrm)
      hn[i] = make_float4(nrm[3 * (size_t)i], nrm[3 * (size_t)i + 1], nrm[3 * (size_t)i + 2], w);
    else
      hn[i] = make_float4(0.f, 0.f, 0.f, w);
  }
  if ((rc = ctx->d_P.ensure(bytes)) != PGP_OK) return rc;
  if ((rc = ctx->d_Pnw.ensure(bytes)) != PGP_OK) return rc;
  if (staged) {   // (out of the pinned image: by a kernel on the stream, no copy-engine hand-over in front of the next launch)
    if ((rc = stage_to_device(ctx->stream, ctx->d_P.p, hp, (size_t)n * sizeof(float4))) != PGP_OK) return rc;
    if ((rc = stage_to_device(ctx->stream, ctx->d_Pnw.p, hn, (size_t)n * sizeof(float4))) != PGP_OK) return rc;
  } else {
    PGP_HIP(hipMemcpyAsync(ctx->d_P.p, hp, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    PGP_HIP(hipMemcpyAsync(ctx->d_Pnw.p, hn, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  }
  ctx->scene_upload_pending = false;
  if (staged) {
    PGP_HIP(hipEventRecord(ctx->ev_s, ctx->stream));
    ctx->scene_upload_pending = true;
  } else {
    PGP_HIP(hipStreamSynchronize(ctx->stream));
  }
  ctx->kd_valid = false;
  if (ctx->exact_ties && (rc = build_kd_ties(ctx, xyz, n)) != PGP_OK) return rc;
  return build_index(ctx, xyz, delta);
}

int pgp_set_scene_weights(pgp_ctx* ctx, const float* weight, int n) {
  if (!ctx || n < 0 || (n > 0 && !weight)) {
    set_error("pgp_set_scene_weights: bad argument");
    return PGP_EINVAL;
  }
  if (n != ctx->nP) {
    set_error("pgp_set_scene_weights: %d weights for a scene of %d points", n, ctx->nP);
    return PGP_ESTATE;
  }
  if (n == 0) return PGP_OK;
  if (!ctx->has_index) {
    set_error("pgp_set_scene_weights: no scene (pgp_set_scene first)");
    return PGP_ESTATE;
  }
  // a pgp_score_lcp_device / pgp_settle_records_device call still queued on the caller's stream reads these weights:
  // the guard drains the device when such a call has been noted (note_device_work).  Not unconditionally: an index
  // build still running on the context's side stream (small scenes) reads the points only, and the caller's next
  // steps -- base selection, congruent sets -- are meant to run beside it.
  CtxGuard guard(ctx);
  int rc = ctx->d_pre_io.ensure((size_t)n * 4);
  if (rc != PGP_OK) return rc;
  if ((rc = ctx->d_prob_cdf.ensure((size_t)n * 8)) != PGP_OK) return rc;
  // The weights and -- base selection draws its first point from them -- their double prefix sums (the sequence
  // launch_select_bases would otherwise read back and add up on its first call) go up from a pinned staging buffer of
  // this call's own: nothing of the caller's is read after the return, and nothing waits for the copies here (the
  // caller's next step, the base selection, queues behind them on the stream).
  const size_t off_cdf = ((size_t)n * 4 + 63) & ~(size_t)63, need = off_cdf + (size_t)n * 8;
  if (ctx->ev_w) PGP_HIP(hipEventSynchronize(ctx->ev_w));   // the previous call's copies out of the staging buffer
  else PGP_HIP(hipEventCreateWithFlags(&ctx->ev_w, hipEventDisableTiming));
  if (need > ctx->h_w_cap) {
    if (ctx->h_w_pin) {
      hipError_t e = hipHostFree(ctx->h_w_pin);
      (void)e;
      ctx->h_w_pin = nullptr;
      ctx->h_w_cap = 0;
    }
    PGP_HIP(hipHostMalloc(&ctx->h_w_pin, need + need / 4, hipHostMallocDefault));
    ctx->h_w_cap = need + need / 4;
  }
  unsigned char* pin = static_cast<unsigned char*>(ctx->h_w_pin);
  std::memcpy(pin, weight, (size_t)n * 4);
  double* cdf = reinterpret_cast<double*>(pin + off_cdf);
  double run = 0.0;
  for (int i = 0; i < n; ++i) {
    run += (double)weight[i];
    cdf[i] = run;
  }
  ctx->prob_cdf_valid = false;
  if ((rc = stage_to_device(ctx->stream, ctx->d_pre_io.p, pin, (size_t)n * 4)) != PGP_OK) return rc;
  if ((rc = launch_scene_weights(ctx, ctx->d_pre_io.as<float>(), n, ctx->stream)) != PGP_OK) return rc;
  if ((rc = stage_to_device(ctx->stream, ctx->d_prob_cdf.p, cdf, (size_t)n * 8)) != PGP_OK) return rc;
  PGP_HIP(hipEventRecord(ctx->ev_w, ctx->stream));
  ctx->prob_cdf_valid = true;
  return PGP_OK;
}

int pgp_set_model(pgp_ctx* ctx, const float* xyz, const float* nrm, int n) {
  if (!ctx || n < 0 || (n > 0 && !xyz)) {
    set_error("pgp_set_model: bad argument (n=%d)", n);
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  PGP_HIP(hipDeviceSynchronize());   // see pgp_set_scene
  ctx->nQ = n;
  ctx->has_model_normals = nrm != nullptr;
  // Morton order: neighbouring lanes hold neighbouring model points, so under any rigid
  // transform a wave's 64 queries fall into neighbouring cells (coalesced bitmap / offset /
  // candidate reads, similar run lengths => little divergence).  Scores are sums over the
  // model, so the order is free; q.w keeps the original index for pgp_registered.
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      float v = xyz[3 * (size_t)i + k];
      if (std::isfinite(v)) {
        mn[k] = std::min(mn[k], v);
        mx[k] = std::max(mx[k], v);
      }
    }
  float scale[3];
  for (int k = 0; k < 3; ++k) scale[k] = (mx[k] > mn[k]) ? 1023.0f / (mx[k] - mn[k]) : 0.f;
  std::vector<uint32_t> code((size_t)std::max(n, 1));
  for (int i = 0; i < n; ++i) {
    uint32_t c[3];
    for (int k = 0; k < 3; ++k) {
      float v = (xyz[3 * (size_t)i + k] - mn[k]) * scale[k];
      c[k] = std::isfinite(v) ? (uint32_t)std::min(1023.f, std::max(0.f, v)) : 0u;
    }
    code[i] = spread10(c[0]) | (spread10(c[1]) << 1) | (spread10(c[2]) << 2);
  }
  std::vector<int> order((size_t)n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return code[a] != code[b] ? code[a] < code[b] : a < b; });
  std::vector<float4> hq((size_t)std::max(n, 1)), hn((size_t)std::max(n, 1));
  for (int s = 0; s < n; ++s) {
    int i = order[s];
    hq[s] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2],
                        __builtin_bit_cast(float, i));
    hn[s] = nrm ? make_float4(nrm[3 * (size_t)i], nrm[3 * (size_t)i + 1], nrm[3 * (size_t)i + 2], 0.f)
                : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int rc;
  if ((rc = ctx->d_Q.ensure(hq.size() * sizeof(float4))) != PGP_OK) return rc;
  if ((rc = ctx->d_Qn.ensure(hn.size() * sizeof(float4))) != PGP_OK) return rc;
  if ((rc = ctx->d_Qpos.ensure(hq.size() * sizeof(int))) != PGP_OK) return rc;
  std::vector<int> hpos((size_t)std::max(n, 1), 0);
  for (int s = 0; s < n; ++s) hpos[order[s]] = s;
  if ((rc = ctx->d_hits.ensure(hq.size() * sizeof(int))) != PGP_OK) return rc;
  if ((rc = ctx->d_seq.ensure(4 * (hq.size() + 4) * sizeof(float))) != PGP_OK) return rc;  // kRefineGroup rows
  if (ctx->exact_records && (rc = ctx->d_rec_ws.ensure((size_t)records_workspace_bytes(n))) != PGP_OK) return rc;
  PGP_HIP(hipMemcpyAsync(ctx->d_Q.p, hq.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  PGP_HIP(hipMemcpyAsync(ctx->d_Qn.p, hn.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  PGP_HIP(hipMemcpyAsync(ctx->d_Qpos.p, hpos.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  PGP_HIP(hipStreamSynchronize(ctx->stream));
  // the per-tile partial buffer depends on the model size
  if (ctx->cap_h > 0) {
    int cap = ctx->cap_h;
    ctx->cap_h = 0;
    return pgp_reserve(ctx, cap);
  }
  return PGP_OK;
}

static int reserve_impl(pgp_ctx* ctx, int max_hypotheses);

}  // extern "C"

namespace pgp {
namespace {
// n16 16-byte vectors, then n_tail (< 4) words
__global__ __launch_bounds__(256) void stage_from_host(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, int n_tail) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) dst[i] = src[i];
  else if (i < n16 + (size_t)n_tail)
    reinterpret_cast<uint32_t*>(dst + n16)[i - n16] = reinterpret_cast<const uint32_t*>(src + n16)[i - n16];
}

struct PubArgs {
  const uint32_t* src[kPubMaxItems];
  uint32_t* dst[kPubMaxItems];
  uint32_t words[kPubMaxItems];
  int n;
  unsigned int* flag;
  unsigned int value;
  unsigned int* ticket;   // device word, zero between launches: the workgroup that draws the last number writes the flag
};
// The items word by word into host memory (system-scope stores: written through, acknowledged), then the word the host polls.
// One workgroup per ~32 KB (up to 32): every workgroup's stores are complete before it draws its ticket, the last one writes
// the word and puts the ticket back.
__global__ __launch_bounds__(1024) void publish_items(PubArgs p) {
  const uint32_t t0 = blockIdx.x * 1024u + threadIdx.x, step = gridDim.x * 1024u;
  for (int k = 0; k < p.n; ++k) {
    const uint32_t* __restrict__ s = p.src[k];
    uint32_t* d = p.dst[k];
    const uint32_t nw = p.words[k];
    if ((((uintptr_t)s | (uintptr_t)d) & 7u) == 0) {   // two words per store
      const unsigned long long* __restrict__ s2 = reinterpret_cast<const unsigned long long*>(s);
      unsigned long long* d2 = reinterpret_cast<unsigned long long*>(d);
      for (uint32_t i = t0; i < nw / 2; i += step) __hip_atomic_store(&d2[i], s2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((nw & 1u) && t0 == 0) __hip_atomic_store(&d[nw - 1], s[nw - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      for (uint32_t i = t0; i < nw; i += step) __hip_atomic_store(&d[i], s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    bool last = gridDim.x == 1;
    if (!last) {
      last = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
      if (last) __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (last) __hip_atomic_store(p.flag, p.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace

bool publish_usable(const PubItem* items, int n) {
  static const bool off = getenv("PGP_HOSTOUT_KERNEL") && atoi(getenv("PGP_HOSTOUT_KERNEL")) == 0;
  if (off || n <= 0 || n > kPubMaxItems) return false;
  size_t total = 0;
  for (int k = 0; k < n; ++k) {
    if (((reinterpret_cast<uintptr_t>(items[k].d_src) | reinterpret_cast<uintptr_t>(items[k].h_dst) | items[k].bytes) & 3u) != 0) return false;
    total += items[k].bytes;
  }
  return total <= kPubMaxBytes;
}

// Queues the publishing kernel behind whatever `st` holds and waits for its completion word.  The word is the context's
// (one wait at a time per context); a wait that runs past a generous bound falls back to the stream itself.
int publish_and_wait(pgp_ctx* ctx, hipStream_t st, const PubItem* items, int n) {
  if (!ctx->h_flag) {
    PGP_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_flag), 64, hipHostMallocDefault));
    *ctx->h_flag = 0u;
  }
  PubArgs a{};
  a.n = n;
  for (int k = 0; k < n; ++k) {
    a.src[k] = static_cast<const uint32_t*>(items[k].d_src);
    a.dst[k] = static_cast<uint32_t*>(items[k].h_dst);
    a.words[k] = (uint32_t)(items[k].bytes / 4);
  }
  a.flag = ctx->h_flag;
  a.value = ++ctx->flag_seq ? ctx->flag_seq : ++ctx->flag_seq;
  size_t total = 0;
  for (int k = 0; k < n; ++k) total += items[k].bytes;
  const unsigned groups = (unsigned)std::min<size_t>(32, std::max<size_t>(1, (total + (32u << 10) - 1) / (32u << 10)));
  if (groups > 1) {
    if (!ctx->d_pub_ticket.p) {
      int rc = ctx->d_pub_ticket.ensure(64);
      if (rc != PGP_OK) return rc;
      PGP_HIP(hipMemsetAsync(ctx->d_pub_ticket.p, 0, 64, st));
    }
    a.ticket = ctx->d_pub_ticket.as<unsigned int>();
  }
  hipLaunchKernelGGL(publish_items, dim3(groups), dim3(1024), 0, st, a);
  PGP_HIP(hipGetLastError());
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(250);
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(ctx->h_flag, __ATOMIC_ACQUIRE) == a.value) {
      return PGP_OK;
    }
    __builtin_ia32_pause();
    if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() >= t_end) break;
  }
  PGP_HIP(hipStreamSynchronize(st));
  if (__atomic_load_n(ctx->h_flag, __ATOMIC_ACQUIRE) != a.value) {
    set_error("publish_and_wait: the stream finished without the completion word");
    return PGP_EHIP;
  }
  return PGP_OK;
}

// Small inputs on their way from a PINNED host image to the device, fetched by a KERNEL on the stream that is about to use
// them instead of a copy-engine transfer in front of that stream's kernels: the hand-over from the copy engine to the compute
// queue costs ~15 us whatever the size (tools/per_call_probe.py: the 4096-hypothesis host-pointer call 140 -> 126 us, the
// 1024-hypothesis one 70 -> 55 us; profiles/r06_ab/stage_kernel.log), a kernel behind a kernel ~2.5 us.  Up to 2 MB (a
// kernel reads host memory at PCIe speed like the engine does, but occupies compute units while it waits); beyond that,
// unaligned, or with PGP_STAGE_KERNEL=0: hipMemcpyAsync as before.
int stage_to_device(hipStream_t st, void* d_dst, const void* h_pinned, size_t bytes) {
  if (bytes == 0) return PGP_OK;
  static const bool off = getenv("PGP_STAGE_KERNEL") && atoi(getenv("PGP_STAGE_KERNEL")) == 0;
  const bool aligned = ((reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(h_pinned)) & 15u) == 0 && (bytes & 3u) == 0;
  if (off || !aligned || bytes > (2u << 20)) {
    PGP_HIP(hipMemcpyAsync(d_dst, h_pinned, bytes, hipMemcpyHostToDevice, st));
    return PGP_OK;
  }
  const size_t n16 = bytes / 16;
  const int n_tail = (int)((bytes & 15u) / 4);
  hipLaunchKernelGGL(stage_from_host, dim3((unsigned)((n16 + (size_t)n_tail + 255) / 256)), dim3(256), 0, st,
                     static_cast<const uint4*>(h_pinned), static_cast<uint4*>(d_dst), n16, n_tail);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}
}  // namespace pgp

extern "C" {

int pgp_reserve(pgp_ctx* ctx, int max_hypotheses) {
  if (!ctx || max_hypotheses < 0) {
    set_error("pgp_reserve: bad argument");
    return PGP_EINVAL;
  }
  return reserve_impl(ctx, max_hypotheses);
}

static int reserve_impl(pgp_ctx* ctx, int max_hypotheses) {
  CtxGuard guard(ctx);
  if (int rc = flush_deferred_build(ctx)) return rc;
  if (ctx->index_pending && hipEventQuery(ctx->ev_index) == hipSuccess) {
    const int rc = finish_index(ctx);   // a finished side-stream build: noticed without waiting
    if (rc != PGP_OK) return rc;
  } else if (ctx->index_pending) {
    (void)hipGetLastError();   // hipErrorNotReady is not an error
  }
  if (max_hypotheses <= ctx->cap_h) return PGP_OK;
  PGP_HIP(hipDeviceSynchronize());   // the workspaces replaced below may be in use by queued launches
  if (int rc = index_settled(ctx)) return rc;
  int rc;
  size_t cap = (size_t)max_hypotheses;
  size_t tiles = (size_t)tiles_for(ctx->nQ);
  if ((rc = ctx->d_T.ensure(cap * 16 * sizeof(float))) != PGP_OK) return rc;
  if ((rc = ctx->d_partial.ensure(tiles * cap * (sizeof(int) + sizeof(float)))) != PGP_OK) return rc;
  if ((rc = ctx->d_scores.ensure(cap * sizeof(float))) != PGP_OK) return rc;
  if ((rc = ctx->d_counts.ensure(cap * sizeof(int))) != PGP_OK) return rc;
  if ((rc = ctx->d_eo_ws.ensure(cap * sizeof(int) + 64)) != PGP_OK) return rc;
  if ((rc = ctx->d_acc.ensure(cap * 16 + 256)) != PGP_OK) return rc;   // the near word + one ticket per chunk (<= cap / 2 + 2)
  PGP_HIP(hipMemset(ctx->d_acc.p, 0, cap * 16 + 256));
  ctx->cap_h = max_hypotheses;
  return PGP_OK;
}

int pgp_score_lcp_device(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg,
                         float* d_scores, int* d_counts, int* d_best, void* stream) {
  if (!ctx || n_h < 0 || (n_h > 0 && (!d_T || !d_scores))) {
    set_error("pgp_score_lcp_device: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx, false);
  const int rc = launch_score(ctx, d_T, n_h, mode, gate_deg, d_scores, d_counts, d_best,
                              static_cast<hipStream_t>(stream));
  note_device_work(ctx, static_cast<hipStream_t>(stream));
  return rc;
}

int pgp_set_exact_ties(pgp_ctx* ctx, int on) {
  if (!ctx) {
    set_error("pgp_set_exact_ties: ctx is NULL");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  ctx->exact_ties = on != 0;
  if (!ctx->exact_ties) ctx->kd_valid = false;   // the next pgp_set_scene builds the tree again if asked to
  return PGP_OK;
}

int pgp_set_exact_records(pgp_ctx* ctx, int on) {
  if (!ctx) {
    set_error("pgp_set_exact_records: ctx is NULL");
    return PGP_EINVAL;
  }
  if (on) {
    CtxGuard guard(ctx);
    int rc = ctx->d_rec_ws.ensure((size_t)records_workspace_bytes(ctx->nQ));
    if (rc != PGP_OK) return rc;
  }
  ctx->exact_records = on != 0;
  return PGP_OK;
}

int pgp_set_verify_early_out(pgp_ctx* ctx, int on) {
  if (!ctx) {
    set_error("pgp_set_verify_early_out: ctx is NULL");
    return PGP_EINVAL;
  }
  ctx->verify_early_out = on != 0;
  return PGP_OK;
}

int pgp_verify_early_out_device(pgp_ctx* ctx, const float* d_T, int n_h, float* d_scores, int* d_counts, void* stream) {
  if (!ctx || n_h < 0 || (n_h > 0 && (!d_T || !d_scores || !d_counts))) {
    set_error("pgp_verify_early_out_device: bad argument");
    return PGP_EINVAL;
  }
  if (n_h > ctx->cap_h) {
    set_error("pgp_verify_early_out_device: %d hypotheses exceed the reserved capacity %d (pgp_reserve)", n_h, ctx->cap_h);
    return PGP_ESTATE;
  }
  CtxGuard guard(ctx, false);
  const int rc = launch_verify_early_out(ctx, d_T, n_h, d_scores, d_counts, static_cast<hipStream_t>(stream));
  note_device_work(ctx, static_cast<hipStream_t>(stream));
  return rc;
}

int pgp_settle_records_device(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg, float* d_scores,
                              void* stream) {
  if (!ctx || n_h < 0 || (n_h > 0 && (!d_T || !d_scores))) {
    set_error("pgp_settle_records_device: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx, false);
  // queue-only entry: no allocation here (a hipFree / hipMalloc would synchronise the device and break a
  // capture); the workspace is reserved by pgp_set_exact_records / pgp_set_model
  if (ctx->d_rec_ws.cap < (size_t)records_workspace_bytes(ctx->nQ)) {
    set_error("pgp_settle_records_device: workspace not reserved (call pgp_set_exact_records(ctx, 1) after pgp_set_model)");
    return PGP_ESTATE;
  }
  int rc = launch_settle_records(ctx, d_T, n_h, mode, gate_deg, d_scores, static_cast<hipStream_t>(stream));
  note_device_work(ctx, static_cast<hipStream_t>(stream));
  return rc;
}

int pgp_settle_best_device(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg,
                           float* d_scores, int* d_best, void* stream) {
  if (!ctx || n_h < 0 || !d_best || (n_h > 0 && (!d_T || !d_scores))) {
    set_error("pgp_settle_best_device: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx, false);
  const int rc = launch_settle_best(ctx, d_T, n_h, mode, gate_deg, d_scores, d_best, static_cast<hipStream_t>(stream));
  note_device_work(ctx, static_cast<hipStream_t>(stream));
  return rc;
}

int pgp_score_lcp(pgp_ctx* ctx, const float* T, int n_h, int mode, float gate_deg, float* scores,
                  int* counts, int* best_index, float* best_score) {
  if (!ctx || n_h < 0 || (n_h > 0 && (!T || !scores))) {
    set_error("pgp_score_lcp: bad argument");
    return PGP_EINVAL;
  }
  // PGP_CALL_PHASES=1: where a call's wall time goes, printed to stderr every 100 calls (diagnostic)
  static const bool phases = getenv("PGP_CALL_PHASES") != nullptr;
  static thread_local double ph_acc[8] = {0};
  static thread_local int ph_n = 0;
  double ph_t = 0.0;
  int ph_k = 0;
  auto ph_now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto ph_mark = [&] {
    if (!phases) return;
    const double t = ph_now();
    if (ph_k > 0) ph_acc[ph_k - 1] += t - ph_t;
    ph_t = t;
    ++ph_k;
  };
  ph_mark();
  CtxGuard guard(ctx);
  int rc = reserve_impl(ctx, n_h);
  if (rc != PGP_OK) return rc;
  hipStream_t st = ctx->stream;
  ph_mark();   // [0] guard + reserve
  // The caller's arrays are pageable (std::vector storage in the node): copying them through a
  // pinned staging buffer of our own makes the two transfers plain DMA (one H2D, ONE D2H of
  // scores | counts | best) instead of the runtime's chunked pageable path -- 199 -> ~150 us per
  // 4096-hypothesis call.
  const size_t nT = (size_t)n_h * 16 * sizeof(float);
  const size_t out_bytes = (size_t)n_h * 8 + 8;
  const size_t pin_need = nT + out_bytes + 64 + 256;   // + {index, score bits, near, spare} and the completion word
  if (pin_need > ctx->h_pin_cap) {
    if (ctx->h_pin) {
      hipError_t e = hipHostFree(ctx->h_pin);
      (void)e;
      ctx->h_pin = nullptr;
      ctx->h_pin_cap = 0;
    }
    const size_t want = pin_need + pin_need / 4;
    PGP_HIP(hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault));
    ctx->h_pin_cap = want;
  }
  if ((rc = ctx->d_out.ensure(out_bytes)) != PGP_OK) return rc;
  unsigned char* pin = static_cast<unsigned char*>(ctx->h_pin);
  unsigned char* pin_out = pin + ((nT + 63) & ~(size_t)63);
  float* d_scores = ctx->d_out.as<float>();
  int* d_counts = reinterpret_cast<int*>(d_scores + n_h);
  int* d_best = d_counts + n_h;
  if (n_h > 0) {
    std::memcpy(pin, T, nT);
    ph_mark();   // [1] transforms into the pinned image
    if ((rc = stage_to_device(st, ctx->d_T.p, pin, nT)) != PGP_OK) return rc;
    ph_mark();   // [2] H2D queued
  }
  // The results' way back.  finalize_scores writes scores, counts and the best straight into the pinned landing area, and a
  // completion word behind them (lcp_score.hip HostPub): no device-to-host copy (a DMA of its own behind the kernel: ~10 us
  // whatever its size) and no wait for the end-of-kernel signal -- the host polls the word.  The opt-in passes over the score
  // vector (exact records, Verify's early termination), a weighted near-tie settled on the device and PGP_HOST_RESULTS=0
  // copy back as before.
  static const bool host_results = !(getenv("PGP_HOST_RESULTS") && atoi(getenv("PGP_HOST_RESULTS")) == 0);
  unsigned char* pin_tail = pin_out + ((out_bytes + 63) & ~(size_t)63);
  ScoreHostOut ho{};
  ho.scores = reinterpret_cast<float*>(pin_out);
  ho.counts = reinterpret_cast<int*>(pin_out + (size_t)n_h * 4);
  ho.best = reinterpret_cast<int*>(pin_tail);
  ho.flag = reinterpret_cast<unsigned int*>(pin_tail + 64);
  ho.flag_value = ++ctx->flag_seq ? ctx->flag_seq : ++ctx->flag_seq;   // never 0
  const bool want_host = host_results && n_h > 0 && !ctx->exact_records && !(ctx->verify_early_out && mode == PGP_MODE_PLAIN);
  if (want_host) {
    // (the word must not hold this call's value from the area's earlier life: the area is re-allocated when it grows)
    __atomic_store_n(ho.flag, 0u, __ATOMIC_RELAXED);
  }
  rc = launch_score(ctx, ctx->d_T.as<float>(), n_h, mode, gate_deg, d_scores, d_counts, d_best, st, want_host ? &ho : nullptr);
  if (rc != PGP_OK) return rc;
  ph_mark();     // [3] kernels queued
  bool via_host = want_host && ho.published;
  if (via_host) {
    ph_mark();   // [4] (no copy to queue)
    // the kernel's own launch is ~0.1 ms per 4096 hypotheses; past a generous bound the stream is asked instead
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20 + n_h / 64);
    bool seen = false;
    for (unsigned spins = 0;; ++spins) {
      if (__atomic_load_n(ho.flag, __ATOMIC_ACQUIRE) == ho.flag_value) {
        seen = true;
        break;
      }
      __builtin_ia32_pause();
      if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() >= t_end) break;
    }
    if (!seen) {
      PGP_HIP(hipStreamSynchronize(st));
      if (__atomic_load_n(ho.flag, __ATOMIC_ACQUIRE) != ho.flag_value) {
        set_error("pgp_score_lcp: the scoring launch finished without publishing its results");
        return PGP_EHIP;
      }
    }
    ph_mark();   // [5] the wait
    if (ho.best[2] != 0) via_host = false;   // weighted near-tie (rare): settled on the device, its arrays come the old way
  }
  if (!via_host) {
    PGP_HIP(hipMemcpyAsync(pin_out, d_scores, out_bytes, hipMemcpyDeviceToHost, st));
    ph_mark();     // [4] D2H queued
    PGP_HIP(hipStreamSynchronize(st));
      ph_mark();     // [5] the wait
  }
  if ((rc = index_settled(ctx)) != PGP_OK) return rc;
  if (n_h > 0) {
    std::memcpy(scores, pin_out, (size_t)n_h * sizeof(float));
    if (counts) std::memcpy(counts, pin_out + (size_t)n_h * 4, (size_t)n_h * sizeof(int));
  }
  int best[2];
  std::memcpy(best, via_host ? pin_tail : pin_out + (size_t)n_h * 8, sizeof best);
  if (best_index) *best_index = best[0];
  if (best_score) std::memcpy(best_score, &best[1], 4);
  ph_mark();     // [6] results out of the pinned image
  if (phases && ++ph_n == 100) {
    fprintf(stderr, "pgp_score_lcp phases (us, mean of 100 calls, n_h %d): reserve %.1f | stage T %.1f | queue H2D %.1f | queue kernels %.1f | "
                    "queue D2H %.1f | wait %.1f | copy out %.1f\n", n_h, ph_acc[0] / 100, ph_acc[1] / 100, ph_acc[2] / 100, ph_acc[3] / 100,
            ph_acc[4] / 100, ph_acc[5] / 100, ph_acc[6] / 100);
    for (double& a : ph_acc) a = 0;
    ph_n = 0;
  }
  return PGP_OK;
}

int pgp_registered(pgp_ctx* ctx, const float* T16, int mode, float gate_deg, int* ids, int* n) {
  if (!ctx || !T16 || !n || (ctx->nQ > 0 && !ids)) {
    set_error("pgp_registered: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  int rc = pgp_reserve(ctx, 1);
  if (rc != PGP_OK) return rc;
  hipStream_t st = ctx->stream;
  PGP_HIP(hipMemcpyAsync(ctx->d_T.p, T16, 16 * sizeof(float), hipMemcpyHostToDevice, st));
  rc = launch_registered(ctx, ctx->d_T.as<float>(), mode, gate_deg, ctx->d_hits.as<int>(), st);
  if (rc != PGP_OK) return rc;
  HostOut out(ctx, st);
  const unsigned char* got = nullptr;
  if ((rc = out.fetch(&got, ctx->d_hits.p, (size_t)ctx->nQ * sizeof(int))) != PGP_OK || (rc = out.sync()) != PGP_OK) return rc;
  const int* hits = reinterpret_cast<const int*>(got);
  if ((rc = index_settled(ctx)) != PGP_OK) return rc;
  int k = 0;
  for (int i = 0; i < ctx->nQ; ++i)
    if (hits[i] >= 0) ids[k++] = hits[i];  // model-point order, as push_back in base.cc:1760
  *n = k;
  return PGP_OK;
}

int pgp_registered_model(pgp_ctx* ctx, const float* T16, const float* q_xyz, const float* q_nrm, int n,
                         float gate_deg, int* ids, int* n_ids) {
  if (!ctx || !T16 || !n_ids || n < 0 || (n > 0 && (!q_xyz || !q_nrm || !ids))) {
    set_error("pgp_registered_model: bad argument");
    return PGP_EINVAL;
  }
  *n_ids = 0;
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  std::vector<float4> hq((size_t)n), hn((size_t)n);
  for (int i = 0; i < n; ++i) {
    hq[i] = make_float4(q_xyz[3 * (size_t)i], q_xyz[3 * (size_t)i + 1], q_xyz[3 * (size_t)i + 2], 0.f);
    hn[i] = make_float4(q_nrm[3 * (size_t)i], q_nrm[3 * (size_t)i + 1], q_nrm[3 * (size_t)i + 2], 0.f);
  }
  const size_t N = (size_t)n;
  int rc = ctx->d_pre_io.ensure(N * 36 + 64 + 256);
  if (rc != PGP_OK) return rc;
  unsigned char* base = ctx->d_pre_io.as<unsigned char>();
  float4* d_q = reinterpret_cast<float4*>(base);
  float4* d_n = d_q + N;
  int* d_hits = reinterpret_cast<int*>(d_n + N);
  float* d_T = reinterpret_cast<float*>(d_hits + N);
  d_T = reinterpret_cast<float*>(((uintptr_t)d_T + 63) & ~(uintptr_t)63);
  PGP_HIP(hipMemcpyAsync(d_q, hq.data(), N * 16, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(d_n, hn.data(), N * 16, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(d_T, T16, 64, hipMemcpyHostToDevice, st));
  rc = launch_registered_model(ctx, d_T, d_q, d_n, n, gate_deg, d_hits, st);
  if (rc != PGP_OK) return rc;
  HostOut out(ctx, st);
  const unsigned char* got = nullptr;
  if ((rc = out.fetch(&got, d_hits, N * 4)) != PGP_OK || (rc = out.sync()) != PGP_OK) return rc;
  const int* hits = reinterpret_cast<const int*>(got);
  if ((rc = index_settled(ctx)) != PGP_OK) return rc;
  int k = 0;
  for (int i = 0; i < n; ++i)
    if (hits[i] >= 0) ids[k++] = hits[i];
  *n_ids = k;
  return PGP_OK;
}

int pgp_find_congruent_4pcs(pgp_ctx* ctx, float invariant1, float invariant2, float threshold, const int* P_pairs,
                            int nP, const int* Q_pairs, int nQ, int* quads, int cap, int* n_quads) {
  if (!ctx || !n_quads || nP < 0 || nQ < 0 || cap < 0 || (nP > 0 && !P_pairs) || (nQ > 0 && !Q_pairs) ||
      (cap > 0 && !quads)) {
    set_error("pgp_find_congruent_4pcs: bad argument");
    return PGP_EINVAL;
  }
  *n_quads = 0;
  if (nP == 0 || nQ == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = ctx->d_cs_pairs.ensure(((size_t)nP + nQ) * 8)) != PGP_OK) return rc;
  if ((rc = ctx->d_cs_out.ensure((size_t)std::max(cap, 1) * 16)) != PGP_OK) return rc;
  int* d_Pp = ctx->d_cs_pairs.as<int>();
  int* d_Qp = d_Pp + 2 * (size_t)nP;
  PGP_HIP(hipMemcpyAsync(d_Pp, P_pairs, (size_t)nP * 8, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(d_Qp, Q_pairs, (size_t)nQ * 8, hipMemcpyHostToDevice, st));
  int total = 0;
  rc = launch_find_congruent_4pcs(ctx, invariant1, invariant2, threshold, d_Pp, nP, d_Qp, nQ, ctx->d_cs_out.as<int>(), cap,
                                  &total, st);
  if (rc != PGP_OK) return rc;
  const int n_copy = std::min(total, cap);
  if (n_copy > 0) PGP_HIP(hipMemcpyAsync(quads, ctx->d_cs_out.p, (size_t)n_copy * 16, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  *n_quads = total;
  return PGP_OK;
}

int pgp_running_best(const float* scores, int n_h, int* selected, int* n_selected) {
  if (n_h < 0 || (n_h > 0 && (!scores || !selected)) || !n_selected) {
    set_error("pgp_running_best: bad argument");
    return PGP_EINVAL;
  }
  float best = 0.f;  // best_LCP_ = 0.0 (base.cc:308)
  int k = 0;
  for (int i = 0; i < n_h; ++i)
    if (scores[i] > best) {
      best = scores[i];
      selected[k++] = i;
    }
  *n_selected = k;
  return PGP_OK;
}

int pgp_set_search_model(pgp_ctx* ctx, const float* xyz, int n) {
  if (!ctx || n < 0 || (n > 0 && !xyz)) {
    set_error("pgp_set_search_model: bad argument (n=%d)", n);
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  PGP_HIP(hipDeviceSynchronize());   // see pgp_set_scene
  ctx->csb_fit_m = 0;
  ctx->csb_nb = 0;   // a resident congruent batch belongs to the old search model
  std::vector<float4> hq((size_t)std::max(n, 1));
  for (int i = 0; i < n; ++i)
    hq[i] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2],
                        __builtin_bit_cast(float, i));
  int rc = ctx->d_Qs.ensure(hq.size() * sizeof(float4));
  if (rc != PGP_OK) return rc;
  // unit-cube image for the congruent-set accelerators (pairCreationFunctor.h:102-138)
  std::vector<float4> hu;
  unit_cube_image(xyz, n, ctx->cs_gcenter, &ctx->cs_ratio, &hu);
  if ((rc = ctx->d_Qs_unit.ensure(hu.size() * sizeof(float4))) != PGP_OK) return rc;
  PGP_HIP(hipMemcpyAsync(ctx->d_Qs_unit.p, hu.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  PGP_HIP(hipMemcpyAsync(ctx->d_Qs.p, hq.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  PGP_HIP(hipStreamSynchronize(ctx->stream));
  ctx->nQs = n;
  return PGP_OK;
}

int pgp_set_ppf_map(pgp_ctx* ctx, const int* keys, const int* counts, const int* pairs, int n_keys) {
  if (!ctx || n_keys < 0 || (n_keys > 0 && !keys)) {
    set_error("pgp_set_ppf_map: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  PGP_HIP(hipDeviceSynchronize());
  return set_ppf_map(ctx, keys, counts, pairs, n_keys);
}

int pgp_select_bases(pgp_ctx* ctx, const double* u, int n_attempts, int* ids, float* invariants, int* status) {
  if (!ctx || n_attempts < 0 || (n_attempts > 0 && (!u || !ids || !invariants || !status))) {
    set_error("pgp_select_bases: bad argument");
    return PGP_EINVAL;
  }
  if (n_attempts == 0) return PGP_OK;
  CtxGuard guard(ctx);
  return launch_select_bases(ctx, u, n_attempts, ids, invariants, status, nullptr, ctx->stream);
}

int pgp_select_bases_rows(pgp_ctx* ctx, const double* u, int n_attempts, int* ids, float* invariants, int* status, int* rows) {
  if (!ctx || n_attempts < 0 || (n_attempts > 0 && (!u || !ids || !invariants || !status || !rows))) {
    set_error("pgp_select_bases_rows: bad argument");
    return PGP_EINVAL;
  }
  if (n_attempts == 0) return PGP_OK;
  CtxGuard guard(ctx);
  return launch_select_bases(ctx, u, n_attempts, ids, invariants, status, rows, ctx->stream);
}

int pgp_select_bases_rows_begin(pgp_ctx* ctx, const double* u, int n_attempts) {
  if (!ctx || n_attempts <= 0 || !u) {
    set_error("pgp_select_bases_rows_begin: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  return launch_select_bases(ctx, u, n_attempts, nullptr, nullptr, nullptr, nullptr, ctx->stream, 1);
}

int pgp_select_bases_rows_end(pgp_ctx* ctx, int* ids, float* invariants, int* status, int* rows) {
  if (!ctx || !ids || !invariants || !status) {
    set_error("pgp_select_bases_rows_end: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  return launch_select_bases(ctx, nullptr, 0, ids, invariants, status, rows, ctx->stream, 2);
}

int pgp_ppf_features(pgp_ctx* ctx, const int* pairs, int m, int* features, int* rows) {
  if (!ctx || m < 0 || (m > 0 && (!pairs || !features))) {
    set_error("pgp_ppf_features: bad argument");
    return PGP_EINVAL;
  }
  if (m == 0) return PGP_OK;
  CtxGuard guard(ctx);
  return launch_ppf_features(ctx, pairs, m, features, rows, ctx->stream);
}

int pgp_stocs_stage_weights(pgp_ctx* ctx, int stage, int base1, int base2, int base3, float* cur, float* sum,
                            int* present) {
  if (!ctx || !cur || !sum || !present) {
    set_error("pgp_stocs_stage_weights: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  return launch_stage_weights(ctx, stage, base1, base2, base3, cur, sum, present, ctx->stream);
}

int pgp_base_invariants(pgp_ctx* ctx, int* ids, int m, float* invariants, int* ok) {
  if (!ctx || m < 0 || (m > 0 && (!ids || !invariants || !ok))) {
    set_error("pgp_base_invariants: bad argument");
    return PGP_EINVAL;
  }
  if (m == 0) return PGP_OK;
  CtxGuard guard(ctx);
  return launch_base_invariants(ctx, ids, m, invariants, ok, ctx->stream);
}

int pgp_rigid_from_congruent_device(pgp_ctx* ctx, const int* d_base_ids, const int* d_quad_ids, int n,
                                    const float centroid_P[3], const float centroid_Q[3], float* d_T,
                                    double* d_pose, int* d_status, float* d_rms, void* stream) {
  if (!ctx || n < 0 || !centroid_P || !centroid_Q || (n > 0 && (!d_base_ids || !d_quad_ids || !d_T || !d_status))) {
    set_error("pgp_rigid_from_congruent_device: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx, false);
  const int rc = launch_rigid(ctx, d_base_ids, d_quad_ids, n, centroid_P, centroid_Q, d_T, d_pose, d_status, d_rms,
                              static_cast<hipStream_t>(stream));
  note_device_work(ctx, static_cast<hipStream_t>(stream));
  return rc;
}

int pgp_rigid_from_congruent(pgp_ctx* ctx, const int* base_ids, const int* quad_ids, int n,
                             const float centroid_P[3], const float centroid_Q[3], float* T,
                             double* pose, int* status, float* rms) {
  if (!ctx || n < 0 || !centroid_P || !centroid_Q || (n > 0 && (!base_ids || !quad_ids || !T || !status))) {
    set_error("pgp_rigid_from_congruent: bad argument");
    return PGP_EINVAL;
  }
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)n;
  int rc;
  if ((rc = ctx->d_ids.ensure(N * 32)) != PGP_OK) return rc;
  // layout of the staged outputs: pose (double, first for alignment) | T | rms | status
  ctx->csb_fit_m = 0;   // d_rig is rewritten below: the fits a pgp_congruent_batch_fetch could name are gone
  if ((rc = ctx->d_rig.ensure(N * (128 + 64 + 4 + 4))) != PGP_OK) return rc;
  int* d_b = ctx->d_ids.as<int>();
  int* d_q = d_b + 4 * N;
  double* d_pose = ctx->d_rig.as<double>();
  float* d_T = reinterpret_cast<float*>(d_pose + 16 * N);
  float* d_rms = d_T + 16 * N;
  int* d_status = reinterpret_cast<int*>(d_rms + N);
  PGP_HIP(hipMemcpyAsync(d_b, base_ids, N * 16, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(d_q, quad_ids, N * 16, hipMemcpyHostToDevice, st));
  rc = launch_rigid(ctx, d_b, d_q, n, centroid_P, centroid_Q, d_T, d_pose, d_status, d_rms, st);
  if (rc != PGP_OK) return rc;
  {
    // (through the pinned landing area where there is room, one wait for all: pgp::HostOut)
    HostOut out(ctx, st);
    int rc_out;
    if ((rc_out = out.to(status, d_status, N * 4)) != PGP_OK || (rms && (rc_out = out.to(rms, d_rms, N * 4)) != PGP_OK) ||
        (rc_out = out.to(T, d_T, N * 64)) != PGP_OK || (pose && (rc_out = out.to(pose, d_pose, N * 128)) != PGP_OK) ||
        (rc_out = out.sync()) != PGP_OK)
      return rc_out;
  }
  return PGP_OK;
}

int pgp_extract_pairs(pgp_ctx* ctx, float pair_distance, float eps, int* pairs, int cap, int* n_pairs) {
  if (!ctx || !n_pairs || cap < 0 || (cap > 0 && !pairs) || !(eps >= 0.f)) {
    set_error("pgp_extract_pairs: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  int rc = ctx->d_cs_out.ensure((size_t)std::max(cap, 1) * 8);
  if (rc != PGP_OK) return rc;
  int total = 0;
  rc = launch_extract_pairs(ctx, pair_distance, eps, ctx->d_cs_out.as<int>(), cap, &total, st);
  if (rc != PGP_OK) return rc;
  int n_copy = std::min(total, cap);
  if (n_copy > 0)
    PGP_HIP(hipMemcpyAsync(pairs, ctx->d_cs_out.p, (size_t)n_copy * 8, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  *n_pairs = total;
  return PGP_OK;
}

int pgp_find_congruent(pgp_ctx* ctx, const float* base, float invariant1, float invariant2, float threshold,
                       const int* P_pairs, int nP, const int* Q_pairs, int nQ, int* quads, int cap,
                       int* n_quads) {
  if (!ctx || !base || !n_quads || nP < 0 || nQ < 0 || cap < 0 || (nP > 0 && !P_pairs) ||
      (nQ > 0 && !Q_pairs) || (cap > 0 && !quads)) {
    set_error("pgp_find_congruent: bad argument");
    return PGP_EINVAL;
  }
  *n_quads = 0;
  if (nP == 0 || nQ == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = ctx->d_cs_pairs.ensure(((size_t)nP + nQ) * 8)) != PGP_OK) return rc;
  if ((rc = ctx->d_cs_out.ensure((size_t)std::max(cap, 1) * 16)) != PGP_OK) return rc;
  int* d_Pp = ctx->d_cs_pairs.as<int>();
  int* d_Qp = d_Pp + 2 * (size_t)nP;
  PGP_HIP(hipMemcpyAsync(d_Pp, P_pairs, (size_t)nP * 8, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(d_Qp, Q_pairs, (size_t)nQ * 8, hipMemcpyHostToDevice, st));
  int total = 0;
  rc = launch_find_congruent(ctx, base, invariant1, invariant2, threshold, d_Pp, nP, d_Qp, nQ,
                             ctx->d_cs_out.as<int>(), cap, &total, st);
  if (rc != PGP_OK) return rc;
  int n_copy = std::min(total, cap);
  if (n_copy > 0)
    PGP_HIP(hipMemcpyAsync(quads, ctx->d_cs_out.p, (size_t)n_copy * 16, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  *n_quads = total;
  return PGP_OK;
}

int pgp_find_congruent_batch(pgp_ctx* ctx, const int* base_ids, const float* base_xyz, const float* invariants,
                             int n_bases, float threshold, int* n_quads) {
  if (!ctx || n_bases < 0 || (n_bases > 0 && (!base_ids || !base_xyz || !invariants || !n_quads))) {
    set_error("pgp_find_congruent_batch: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  return launch_find_congruent_batch(ctx, base_ids, base_xyz, invariants, nullptr, n_bases, threshold, n_quads, ctx->stream);
}

int pgp_find_congruent_batch_rows(pgp_ctx* ctx, const int* base_ids, const float* base_xyz, const float* invariants,
                                  const int* rows, int n_bases, float threshold, int* n_quads) {
  if (!ctx || n_bases < 0 || (n_bases > 0 && (!base_ids || !base_xyz || !invariants || !rows || !n_quads))) {
    set_error("pgp_find_congruent_batch_rows: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  return launch_find_congruent_batch(ctx, base_ids, base_xyz, invariants, rows, n_bases, threshold, n_quads, ctx->stream);
}

int pgp_congruent_batch_quads(pgp_ctx* ctx, const int* picks, int m, int* quads) {
  if (!ctx || m < 0 || (m > 0 && (!picks || !quads))) {
    set_error("pgp_congruent_batch_quads: bad argument");
    return PGP_EINVAL;
  }
  if (m == 0) return PGP_OK;
  CtxGuard guard(ctx);
  int rc = ctx->d_cs_out.ensure((size_t)m * 16);
  if (rc != PGP_OK) return rc;
  rc = launch_congruent_batch_gather(ctx, picks, m, ctx->d_cs_out.as<int4>(), ctx->stream);
  if (rc != PGP_OK) return rc;
  PGP_HIP(hipMemcpyAsync(quads, ctx->d_cs_out.p, (size_t)m * 16, hipMemcpyDeviceToHost, ctx->stream));
  PGP_HIP(hipStreamSynchronize(ctx->stream));
  return PGP_OK;
}

int pgp_congruent_batch_fit(pgp_ctx* ctx, const int* picks, const int* base_ids, int m, const float centroid_P[3],
                            const float centroid_Q[3], float* T, double* pose, int* status, float* rms) {
  if (!ctx || m < 0 || !centroid_P || !centroid_Q || (m > 0 && (!picks || !base_ids || !T || !status))) {
    set_error("pgp_congruent_batch_fit: bad argument");
    return PGP_EINVAL;
  }
  if (m == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)m;
  int rc;
  if ((rc = ctx->d_ids.ensure(N * 32)) != PGP_OK) return rc;
  ctx->csb_fit_m = 0;   // d_rig is rewritten below: the fits a pgp_congruent_batch_fetch could name are gone
  if ((rc = ctx->d_rig.ensure(N * (128 + 64 + 4 + 4))) != PGP_OK) return rc;
  int* d_b = ctx->d_ids.as<int>();
  int* d_q = d_b + 4 * N;
  // the base of every pick (scene ids), staged from the host; the quads never leave the device
  std::vector<int> hb(4 * N);
  for (size_t k = 0; k < N; ++k) {
    const int b = picks[2 * k];
    if (b < 0 || b >= ctx->csb_nb) {
      set_error("pgp_congruent_batch_fit: pick %zu names base %d of %d", k, b, ctx->csb_nb);
      return PGP_EINVAL;
    }
    for (int j = 0; j < 4; ++j) hb[4 * k + j] = base_ids[4 * (size_t)b + j];
  }
  PGP_HIP(hipMemcpyAsync(d_b, hb.data(), N * 16, hipMemcpyHostToDevice, st));
  rc = launch_congruent_batch_gather(ctx, picks, m, reinterpret_cast<int4*>(d_q), st);
  if (rc != PGP_OK) return rc;
  double* d_pose = ctx->d_rig.as<double>();
  float* d_T = reinterpret_cast<float*>(d_pose + 16 * N);
  float* d_rms = d_T + 16 * N;
  int* d_status = reinterpret_cast<int*>(d_rms + N);
  rc = launch_rigid(ctx, d_b, d_q, m, centroid_P, centroid_Q, d_T, d_pose, d_status, d_rms, st);
  if (rc != PGP_OK) return rc;
  {
    // (through the pinned landing area where there is room, one wait for all: pgp::HostOut)
    HostOut out(ctx, st);
    int rc_out;
    if ((rc_out = out.to(status, d_status, N * 4)) != PGP_OK || (rms && (rc_out = out.to(rms, d_rms, N * 4)) != PGP_OK) ||
        (rc_out = out.to(T, d_T, N * 64)) != PGP_OK || (pose && (rc_out = out.to(pose, d_pose, N * 128)) != PGP_OK) ||
        (rc_out = out.sync()) != PGP_OK)
      return rc_out;
  }   // also: hb is a stack-owned staging vector
  return PGP_OK;
}

namespace {
// a fit the reference does not push (status != 1: base.cc:1467-1485) is replaced by a transform that registers
// nothing -- identity rotation, translation 1e6 -- so that it scores 0 and can never enter the running-best walk
__global__ __launch_bounds__(256) void mask_failed_fits(const int* __restrict__ status, int m, float* __restrict__ T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m || status[i] == 1) return;
  float* o = T + 16 * (size_t)i;
  for (int q = 0; q < 16; ++q) o[q] = (q % 5 == 0) ? 1.f : 0.f;
  o[12] = o[13] = o[14] = 1e6f;
}
// 16 lanes per kept fit: its float transform and its double pose into a contiguous staging block
__global__ __launch_bounds__(64) void gather_fits(const int* __restrict__ index, int k, const float* __restrict__ T,
                                                  const double* __restrict__ pose, float* __restrict__ T_out,
                                                  double* __restrict__ pose_out) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 4), q = threadIdx.x & 15;
  if (j >= k) return;
  const size_t h = (size_t)index[j];
  T_out[16 * (size_t)j + q] = T[16 * h + q];
  pose_out[16 * (size_t)j + q] = pose[16 * h + q];
}

// The running-best walk of the verification loop (base.cc:1891-1908: hypothesis i enters when lcp_i > best so far, strict,
// best_LCP_ = 0 to begin with) over the FINAL scores on the device -- what pgp_running_best does on the host, entry for
// entry.  One block; 16 consecutive scores per thread, the running maximum in front of a thread's run by a prefix-max over
// the block.  list[0] = number of records (all of them), list[1] = number of fits that were pushed (status == 1),
// list[2 .. 2 + cap) = the first cap records' indices, rec_score[cap] their scores.
__global__ __launch_bounds__(256) void records_walk(int n, const float* __restrict__ scores, const int* __restrict__ status, int cap,
                                                    int* __restrict__ list, float* __restrict__ rec_score) {
  constexpr int kPer = 16;
  __shared__ float s_wmax[4];
  __shared__ int s_wcnt[4], s_wkept[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float carry = 0.f;
  int base = 0, kept = 0;
  for (int s0 = 0; s0 < n; s0 += 256 * kPer) {
    float v[kPer];
    const int lo = s0 + tid * kPer;
    int my_kept = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      v[k] = lo + k < n ? scores[lo + k] : 0.f;
      my_kept += (lo + k < n && status[lo + k] == 1) ? 1 : 0;
    }
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < kPer; ++k) m = fmaxf(m, v[k]);
    float inc = m;
    int ik = my_kept;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float o = __shfl_up(inc, off, 64);
      const int ok = __shfl_up(ik, off, 64);
      if (lane >= off) {
        inc = fmaxf(inc, o);
        ik += ok;
      }
    }
    float excl = __shfl_up(inc, 1, 64);
    if (lane == 0) excl = 0.f;
    if (lane == 63) {
      s_wmax[wave] = inc;
      s_wkept[wave] = ik;
    }
    __syncthreads();
    float run = fmaxf(carry, excl);
    for (int w = 0; w < wave; ++w) run = fmaxf(run, s_wmax[w]);
    unsigned mask = 0u;
    int c = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      if (v[k] > run) {
        run = v[k];
        mask |= 1u << k;
        ++c;
      }
    int ic = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(ic, off, 64);
      if (lane >= off) ic += o;
    }
    if (lane == 63) s_wcnt[wave] = ic;
    __syncthreads();
    int at = base + ic - c;
    for (int w = 0; w < wave; ++w) at += s_wcnt[w];
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      if (mask & (1u << k)) {
        if (at < cap) {
          list[2 + at] = lo + k;
          rec_score[at] = v[k];
        }
        ++at;
      }
    base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    kept += s_wkept[0] + s_wkept[1] + s_wkept[2] + s_wkept[3];
    carry = fmaxf(fmaxf(fmaxf(carry, s_wmax[0]), fmaxf(s_wmax[1], s_wmax[2])), s_wmax[3]);
    __syncthreads();
  }
  if (tid == 0) {
    list[0] = base;
    list[1] = kept;
  }
}

// entries 0 .. min(records, cap) - 1: the records of `list`; entry cap: the best hypothesis (best[0] < 0: zeros)
__global__ __launch_bounds__(64) void gather_fits_list(const int* __restrict__ list, int cap, const int* __restrict__ best,
                                                       const float* __restrict__ T, const double* __restrict__ pose,
                                                       float* __restrict__ T_out, double* __restrict__ pose_out) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 4), q = threadIdx.x & 15;
  if (j > cap) return;
  const int n_rec = list[0] < cap ? list[0] : cap;
  int h = -1;
  if (j < n_rec) h = list[2 + j];
  else if (j == cap) h = best[0];
  T_out[16 * (size_t)j + q] = h >= 0 ? T[16 * (size_t)h + q] : 0.f;
  pose_out[16 * (size_t)j + q] = h >= 0 ? pose[16 * (size_t)h + q] : 0.0;
}
}  // namespace

int pgp_congruent_batch_fit_score(pgp_ctx* ctx, const int* picks, const int* base_ids, int m, const float centroid_P[3],
                                  const float centroid_Q[3], int mode, float gate_deg, float* scores, int* status,
                                  int* best_index, float* best_score) {
  if (!ctx || m < 0 || !centroid_P || !centroid_Q || (m > 0 && (!picks || !base_ids || !scores || !status))) {
    set_error("pgp_congruent_batch_fit_score: bad argument");
    return PGP_EINVAL;
  }
  if (best_index) *best_index = -1;
  if (best_score) *best_score = 0.f;
  if (m == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)m;
  int rc;
  if ((rc = pgp_reserve(ctx, m)) != PGP_OK) return rc;
  if ((rc = ctx->d_ids.ensure(N * 32)) != PGP_OK) return rc;
  ctx->csb_fit_m = 0;   // d_rig is rewritten below: the fits a pgp_congruent_batch_fetch could name are gone
  if ((rc = ctx->d_rig.ensure(N * (128 + 64 + 4 + 4))) != PGP_OK) return rc;
  if ((rc = ctx->d_out.ensure(N * 8 + 8)) != PGP_OK) return rc;
  // pinned staging: [bases of the picks | picks] in, [scores | status | best] out
  const size_t in_bytes = N * 16, out_bytes = N * 8 + 8, pin_need = in_bytes + out_bytes + 128;
  if (pin_need > ctx->h_pin_cap) {
    if (ctx->h_pin) {
      hipError_t e = hipHostFree(ctx->h_pin);
      (void)e;
      ctx->h_pin = nullptr;
      ctx->h_pin_cap = 0;
    }
    const size_t want = pin_need + pin_need / 4;
    PGP_HIP(hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault));
    ctx->h_pin_cap = want;
  }
  unsigned char* pin = static_cast<unsigned char*>(ctx->h_pin);
  int* hb = reinterpret_cast<int*>(pin);
  for (size_t k = 0; k < N; ++k) {
    const int b = picks[2 * k];
    if (b < 0 || b >= ctx->csb_nb) {
      set_error("pgp_congruent_batch_fit_score: pick %zu names base %d of %d", k, b, ctx->csb_nb);
      return PGP_EINVAL;
    }
    for (int j = 0; j < 4; ++j) hb[4 * k + j] = base_ids[4 * (size_t)b + j];
  }
  int* d_b = ctx->d_ids.as<int>();
  int* d_q = d_b + 4 * N;
  PGP_HIP(hipMemcpyAsync(d_b, hb, in_bytes, hipMemcpyHostToDevice, st));
  rc = launch_congruent_batch_gather(ctx, picks, m, reinterpret_cast<int4*>(d_q), st);
  if (rc != PGP_OK) return rc;
  double* d_pose = ctx->d_rig.as<double>();
  float* d_T = reinterpret_cast<float*>(d_pose + 16 * N);
  float* d_rms = d_T + 16 * N;
  int* d_status = reinterpret_cast<int*>(d_rms + N);
  rc = launch_rigid(ctx, d_b, d_q, m, centroid_P, centroid_Q, d_T, d_pose, d_status, d_rms, st);
  if (rc != PGP_OK) return rc;
  hipLaunchKernelGGL(mask_failed_fits, dim3((m + 255) / 256), dim3(256), 0, st, (const int*)d_status, m, d_T);
  float* d_scores = ctx->d_out.as<float>();
  int* d_best = reinterpret_cast<int*>(d_scores + 2 * N);
  rc = launch_score(ctx, d_T, m, mode, gate_deg, d_scores, nullptr, d_best, st);
  if (rc != PGP_OK) return rc;
  unsigned char* pin_out = pin + ((in_bytes + 63) & ~(size_t)63);
  PGP_HIP(hipMemcpyAsync(pin_out, d_scores, N * 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipMemcpyAsync(pin_out + N * 4, d_status, N * 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipMemcpyAsync(pin_out + N * 8, d_best, 8, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  std::memcpy(scores, pin_out, N * 4);
  std::memcpy(status, pin_out + N * 4, N * 4);
  int best[2];
  std::memcpy(best, pin_out + N * 8, sizeof best);
  if (best_index) *best_index = best[0];
  if (best_score) std::memcpy(best_score, &best[1], 4);
  ctx->csb_fit_m = m;
  return PGP_OK;
}

int pgp_congruent_batch_fetch(pgp_ctx* ctx, const int* index, int k, float* T, double* pose) {
  if (!ctx || k < 0 || (k > 0 && !index)) {
    set_error("pgp_congruent_batch_fetch: bad argument");
    return PGP_EINVAL;
  }
  if (k == 0) return PGP_OK;
  CtxGuard guard(ctx);
  const size_t N = (size_t)ctx->csb_fit_m;
  if (N == 0 || ctx->d_rig.cap < N * (128 + 64 + 4 + 4)) {
    set_error("pgp_congruent_batch_fetch: no fits resident (pgp_congruent_batch_fit_score first)");
    return PGP_ESTATE;
  }
  const double* d_pose = ctx->d_rig.as<double>();
  const float* d_T = reinterpret_cast<const float*>(d_pose + 16 * N);
  for (int j = 0; j < k; ++j)
    if (index[j] < 0 || (size_t)index[j] >= N) {
      set_error("pgp_congruent_batch_fetch: index %d of %zu", index[j], N);
      return PGP_EINVAL;
    }
  // gathered on the device, ONE copy back through the pinned buffer (a 64-byte copy into pageable memory per
  // entry cost 15 us each: 0.3 ms for the twenty entries of a running-best list)
  const size_t K = (size_t)k, pin_need = K * (4 + 128 + 64) + 256;
  int rc;
  if ((rc = ctx->d_cs_out.ensure(K * (4 + 128 + 64) + 256)) != PGP_OK) return rc;
  if (pin_need > ctx->h_pin_cap) {
    if (ctx->h_pin) {
      hipError_t e = hipHostFree(ctx->h_pin);
      (void)e;
      ctx->h_pin = nullptr;
      ctx->h_pin_cap = 0;
    }
    PGP_HIP(hipHostMalloc(&ctx->h_pin, pin_need + pin_need / 4, hipHostMallocDefault));
    ctx->h_pin_cap = pin_need + pin_need / 4;
  }
  unsigned char* pin = static_cast<unsigned char*>(ctx->h_pin);
  unsigned char* dev = ctx->d_cs_out.as<unsigned char>();
  const size_t off_pose = (K * 4 + 127) & ~(size_t)127, off_T = off_pose + K * 128;
  std::memcpy(pin, index, K * 4);
  PGP_HIP(hipMemcpyAsync(dev, pin, K * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(gather_fits, dim3((k + 3) / 4), dim3(64), 0, ctx->stream, reinterpret_cast<const int*>(dev), k, d_T, d_pose,
                     reinterpret_cast<float*>(dev + off_T), reinterpret_cast<double*>(dev + off_pose));
  PGP_HIP(hipMemcpyAsync(pin + off_pose, dev + off_pose, K * (128 + 64), hipMemcpyDeviceToHost, ctx->stream));
  PGP_HIP(hipStreamSynchronize(ctx->stream));
  if (pose) std::memcpy(pose, pin + off_pose, K * 128);
  if (T) std::memcpy(T, pin + off_T, K * 64);
  return PGP_OK;
}

}  // extern "C"

namespace pgp {
namespace {
// The drop-in's draw of at most `cap` quads per base (base.cc:1858-1866) as a function of (seed, base, quads of the base)
// alone.  Every base has a COUNTER-BASED generator of its own: variate i is splitmix64's finaliser of state + (i + 1) * gamma, the
// state mixed out of the seed and the base's number -- so a wave computes all `cap` variates at once.  The subset is Floyd's: for
// i = 0 .. cap-1, with j = n - cap + i: t = (variate i >> 33) % (j + 1) (31-bit values, as rand() gives); t joins the sample
// unless it is in it already, in which case j does.  Exactly `cap` steps whatever n (drawing until `cap` different values have
// come takes ~5 n draws when n is barely above cap), every subset equally likely; handed out in ascending order; a base with
// fewer than `cap` quads hands out all of them.  The reference draws from the process's rand(), seeded from the clock: any
// uniform sample of `cap` distinct quads is its behaviour.  sample_quads_kernel and pgp_sample_quads: bit for bit the same picks.
__host__ __device__ inline unsigned long long sample_state(unsigned long long seed, int base) {
  return (seed ^ 0xD1B54A32D192ED03ull) + (unsigned long long)(base + 1) * 0xBF58476D1CE4E5B9ull;
}
__host__ __device__ inline unsigned int sample_variate(unsigned long long state, int i) {
  unsigned long long z = state + (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned int)(z >> 33);
}
constexpr int kSampleMax = 128;   // two slots per lane of the base's wave

// one wave per base: where the base's picks start (the sum of the earlier bases' counts), then its draw
__global__ __launch_bounds__(64) void sample_quads_kernel(const uint32_t* __restrict__ base_start, int nb, int cap, unsigned long long seed,
                                                          const int4* __restrict__ base_ids, int2* __restrict__ picks,
                                                          int4* __restrict__ pick_bases) {
  const int b = blockIdx.x, lane = threadIdx.x;
  uint32_t off = 0;
  for (int i = lane; i < b; i += 64) off += min(base_start[i + 1] - base_start[i], (uint32_t)cap);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) off += __shfl_xor(off, o, 64);
  const uint32_t nq = base_start[b + 1] - base_start[b];
  const int4 ids = base_ids[b];
  if (nq < (uint32_t)cap) {
    for (uint32_t j = lane; j < nq; j += 64) {
      picks[off + j] = make_int2(b, (int)j);
      pick_bases[off + j] = ids;
    }
    return;
  }
  // this lane's two candidates of Floyd's steps i = lane and i = lane + 64, all lanes at once
  const unsigned long long x = sample_state(seed, b);
  const uint32_t j0 = nq - (uint32_t)cap + (uint32_t)lane, j1 = j0 + 64u;
  const uint32_t t0 = lane < cap ? sample_variate(x, lane) % (j0 + 1u) : 0u;
  const uint32_t t1 = lane + 64 < cap ? sample_variate(x, lane + 64) % (j1 + 1u) : 0u;
  uint32_t s0 = 0xFFFFFFFFu, s1 = 0xFFFFFFFFu;   // slot `lane` and slot `lane + 64` (no quad is number 2^32 - 1)
  for (int i = 0; i < cap; ++i) {
    // (i is the same in every lane: v_readlane, not a cross-lane permute in the middle of the chain)
    const uint32_t t = (uint32_t)(i < 64 ? __builtin_amdgcn_readlane((int)t0, i) : __builtin_amdgcn_readlane((int)t1, i - 64));
    const uint32_t v = __ballot(s0 == t || s1 == t) != 0ull ? nq - (uint32_t)cap + (uint32_t)i : t;
    if (lane == (i & 63)) {
      if (i < 64) s0 = v;
      else s1 = v;
    }
  }
  // ascending: a value's place is the number of smaller ones
  uint32_t r0 = 0, r1 = 0;
  for (int i = 0; i < cap; ++i) {
    const uint32_t u = (uint32_t)(i < 64 ? __builtin_amdgcn_readlane((int)s0, i) : __builtin_amdgcn_readlane((int)s1, i - 64));
    r0 += u < s0 ? 1u : 0u;
    r1 += u < s1 ? 1u : 0u;
  }
  if (lane < cap) {
    picks[off + r0] = make_int2(b, (int)s0);
    pick_bases[off + r0] = ids;
  }
  if (lane + 64 < cap) {
    picks[off + r1] = make_int2(b, (int)s1);
    pick_bases[off + r1] = ids;
  }
}

struct SampleSpec {
  bool on;
  unsigned long long seed;
  int cap;
  int* picks_out;   // nullable: the picks the device drew, [m][2]
  int* m_out;       // nullable
};

int fit_score_list_impl(pgp_ctx* ctx, const int* picks, const int* base_ids, int m, const float centroid_P[3],
                        const float centroid_Q[3], int mode, float gate_deg, int list_cap, int* n_list,
                        int* list_index, float* list_score, float* list_T, double* list_pose, int* n_pushed,
                        int* best_index, float* best_score, float* best_T, double* best_pose, int* registered,
                        int* n_registered, const SampleSpec& smp);
}  // namespace
}  // namespace pgp

extern "C" {

int pgp_sample_quads(unsigned long long seed, const int* n_quads, int n_bases, int max_per_base, int* picks, int* n_picks) {
  if (n_bases < 0 || (n_bases > 0 && !n_quads) || max_per_base < 1 || max_per_base > kSampleMax || !n_picks) {
    set_error("pgp_sample_quads: bad argument (1 <= max_per_base <= %d)", kSampleMax);
    return PGP_EINVAL;
  }
  int k = 0;
  std::vector<unsigned int> got;
  for (int b = 0; b < n_bases; ++b) {
    const int nq = n_quads[b];
    if (nq < 0) {
      set_error("pgp_sample_quads: base %d has %d quads", b, nq);
      return PGP_EINVAL;
    }
    if (nq < max_per_base) {
      for (int j = 0; j < nq; ++j, ++k)
        if (picks) { picks[2 * (size_t)k] = b; picks[2 * (size_t)k + 1] = j; }
      continue;
    }
    got.clear();
    const unsigned long long x = sample_state(seed, b);
    for (int i = 0; i < max_per_base; ++i) {   // Floyd's steps (see sample_quads_kernel)
      const unsigned int j = (unsigned int)(nq - max_per_base + i);
      const unsigned int t = sample_variate(x, i) % (j + 1u);
      got.push_back(std::find(got.begin(), got.end(), t) == got.end() ? t : j);
    }
    std::sort(got.begin(), got.end());
    for (unsigned int v : got) {
      if (picks) { picks[2 * (size_t)k] = b; picks[2 * (size_t)k + 1] = (int)v; }
      ++k;
    }
  }
  *n_picks = k;
  return PGP_OK;
}

int pgp_congruent_batch_fit_score_list(pgp_ctx* ctx, const int* picks, const int* base_ids, int m, const float centroid_P[3],
                                       const float centroid_Q[3], int mode, float gate_deg, int list_cap, int* n_list,
                                       int* list_index, float* list_score, float* list_T, double* list_pose, int* n_pushed,
                                       int* best_index, float* best_score, float* best_T, double* best_pose, int* registered,
                                       int* n_registered) {
  if (m > 0 && !picks) {
    set_error("pgp_congruent_batch_fit_score_list: bad argument");
    return PGP_EINVAL;
  }
  return fit_score_list_impl(ctx, picks, base_ids, m, centroid_P, centroid_Q, mode, gate_deg, list_cap, n_list, list_index, list_score,
                             list_T, list_pose, n_pushed, best_index, best_score, best_T, best_pose, registered, n_registered,
                             SampleSpec{false, 0ull, 0, nullptr, nullptr});
}

int pgp_congruent_batch_sample_fit_score_list(pgp_ctx* ctx, unsigned long long seed, int max_per_base, const int* base_ids,
                                              const float centroid_P[3], const float centroid_Q[3], int mode, float gate_deg,
                                              int list_cap, int* n_list, int* list_index, float* list_score, float* list_T,
                                              double* list_pose, int* n_pushed, int* best_index, float* best_score, float* best_T,
                                              double* best_pose, int* registered, int* n_registered, int* picks_out, int* n_picks) {
  if (max_per_base < 1 || max_per_base > kSampleMax) {
    set_error("pgp_congruent_batch_sample_fit_score_list: 1 <= max_per_base <= %d", kSampleMax);
    return PGP_EINVAL;
  }
  if (n_picks) *n_picks = 0;
  return fit_score_list_impl(ctx, nullptr, base_ids, 0, centroid_P, centroid_Q, mode, gate_deg, list_cap, n_list, list_index,
                             list_score, list_T, list_pose, n_pushed, best_index, best_score, best_T, best_pose, registered,
                             n_registered, SampleSpec{true, seed, max_per_base, picks_out, n_picks});
}

}  // extern "C"

namespace pgp {
namespace {
int fit_score_list_impl(pgp_ctx* ctx, const int* picks, const int* base_ids, int m, const float centroid_P[3],
                        const float centroid_Q[3], int mode, float gate_deg, int list_cap, int* n_list,
                        int* list_index, float* list_score, float* list_T, double* list_pose, int* n_pushed,
                        int* best_index, float* best_score, float* best_T, double* best_pose, int* registered,
                        int* n_registered, const SampleSpec& smp) {
  if (!ctx || m < 0 || !centroid_P || !centroid_Q || list_cap < 0 || list_cap > 4096 || !n_list || !n_registered ||
      ((m > 0 || smp.on) && !base_ids) || (list_cap > 0 && (!list_index || !list_score || !list_T || !list_pose))) {
    set_error("pgp_congruent_batch_fit_score_list: bad argument");
    return PGP_EINVAL;
  }
  if (smp.on) {
    // the picks are drawn on the device; how many there will be follows from the resident batch's quad counts
    if (ctx->csb_nb <= 0 || (int)ctx->csb_starts.size() != ctx->csb_nb + 1) {
      set_error("pgp_congruent_batch_sample_fit_score_list: no congruent batch resident (pgp_find_congruent_batch first)");
      return PGP_ESTATE;
    }
    long long total = 0;
    for (int b = 0; b < ctx->csb_nb; ++b)
      total += std::min<long long>((long long)(ctx->csb_starts[b + 1] - ctx->csb_starts[b]), smp.cap);
    if (total > 0x3FFFFFFF) {
      set_error("pgp_congruent_batch_sample_fit_score_list: more than 2^30 picks");
      return PGP_EINVAL;
    }
    m = (int)total;
    if (smp.m_out) *smp.m_out = m;
  }
  *n_list = 0;
  *n_registered = 0;
  if (n_pushed) *n_pushed = 0;
  if (best_index) *best_index = -1;
  if (best_score) *best_score = 0.f;
  if (m == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)m, C = (size_t)list_cap, nQ = (size_t)std::max(ctx->nQ, 0);
  int rc;
  if ((rc = pgp_reserve(ctx, m)) != PGP_OK) return rc;
  if ((rc = ctx->d_ids.ensure(N * 40 + 64 + (size_t)std::max(ctx->csb_nb, 0) * 16 + 64)) != PGP_OK) return rc;   // bases of the picks | picks | quads | ids of the bases
  ctx->csb_fit_m = 0;
  if ((rc = ctx->d_rig.ensure(N * (128 + 64 + 4 + 4))) != PGP_OK) return rc;
  // scores | (counts) | best -- then what goes home in ONE copy: best {index, score bits} | list {records, pushed, indices}
  // | record scores | poses (records, best) | transforms (records, best) | hits of the best pose
  const size_t off_best = N * 8, off_list = off_best + 16, off_rs = off_list + (C + 2) * 4, off_pose = (off_rs + C * 4 + 127) & ~(size_t)127,
               off_T = off_pose + (C + 1) * 128, off_hits = off_T + (C + 1) * 64, total = off_hits + nQ * 4 + 64;
  if ((rc = ctx->d_out.ensure(total)) != PGP_OK) return rc;
  const size_t in_bytes = smp.on ? (((size_t)std::max(ctx->csb_nb, 0) * 16 + 63) & ~(size_t)63) : N * 24;   // bases of the picks | picks
  const size_t home = total - off_best, pin_need = in_bytes + home + (smp.on && smp.picks_out ? N * 8 + 64 : 0) + 256;
  if (pin_need > ctx->h_pin_cap) {
    if (ctx->h_pin) {
      hipError_t e = hipHostFree(ctx->h_pin);
      (void)e;
      ctx->h_pin = nullptr;
      ctx->h_pin_cap = 0;
    }
    const size_t want = pin_need + pin_need / 4;
    PGP_HIP(hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault));
    ctx->h_pin_cap = want;
  }
  unsigned char* pin = static_cast<unsigned char*>(ctx->h_pin);
  int* hb = reinterpret_cast<int*>(pin);
  // device: bases of the picks [4 N] | picks [2 N] (ONE copy) | quads [4 N] | (sampled form) the bases' ids [4 nb]
  int* d_b = ctx->d_ids.as<int>();
  int* d_pk = d_b + 4 * N;
  int* d_q = d_pk + ((2 * N + 3) & ~(size_t)3);   // (int4 records: 16-byte aligned)
  if (!smp.on) {
    for (size_t k = 0; k < N; ++k) {
      const int b = picks[2 * k];
      if (b < 0 || b >= ctx->csb_nb) {
        set_error("pgp_congruent_batch_fit_score_list: pick %zu names base %d of %d", k, b, ctx->csb_nb);
        return PGP_EINVAL;
      }
      for (int j = 0; j < 4; ++j) hb[4 * k + j] = base_ids[4 * (size_t)b + j];
    }
    std::memcpy(hb + 4 * N, picks, N * 8);
    if ((rc = stage_to_device(st, d_b, hb, in_bytes)) != PGP_OK) return rc;
    rc = launch_congruent_batch_gather(ctx, picks, m, reinterpret_cast<int4*>(d_q), st, reinterpret_cast<const int2*>(d_pk));
    if (rc != PGP_OK) return rc;
  } else {
    // the bases' ids go up (16 B each), the picks are drawn where the quads are: one wave per base
    const int nb = ctx->csb_nb;
    int* d_base_ids = d_q + 4 * N;
    std::memcpy(hb, base_ids, (size_t)nb * 16);
    if ((rc = stage_to_device(st, d_base_ids, hb, (size_t)nb * 16)) != PGP_OK) return rc;
    const uint32_t* d_base_start = congruent_batch_starts_device(ctx);
    hipLaunchKernelGGL(sample_quads_kernel, dim3((unsigned)nb), dim3(64), 0, st, d_base_start, nb, smp.cap, smp.seed,
                       reinterpret_cast<const int4*>(d_base_ids), reinterpret_cast<int2*>(d_pk), reinterpret_cast<int4*>(d_b));
    PGP_HIP(hipGetLastError());
    rc = launch_congruent_batch_gather(ctx, nullptr, m, reinterpret_cast<int4*>(d_q), st, reinterpret_cast<const int2*>(d_pk));
    if (rc != PGP_OK) return rc;
  }
  double* d_pose = ctx->d_rig.as<double>();
  float* d_T = reinterpret_cast<float*>(d_pose + 16 * N);
  float* d_rms = d_T + 16 * N;
  int* d_status = reinterpret_cast<int*>(d_rms + N);
  rc = launch_rigid(ctx, d_b, d_q, m, centroid_P, centroid_Q, d_T, d_pose, d_status, d_rms, st);
  if (rc != PGP_OK) return rc;
  hipLaunchKernelGGL(mask_failed_fits, dim3((m + 255) / 256), dim3(256), 0, st, (const int*)d_status, m, d_T);
  unsigned char* dev = ctx->d_out.as<unsigned char>();
  float* d_scores = reinterpret_cast<float*>(dev);
  int* d_best = reinterpret_cast<int*>(dev + off_best);
  int* d_list = reinterpret_cast<int*>(dev + off_list);
  float* d_rs = reinterpret_cast<float*>(dev + off_rs);
  double* d_pose_out = reinterpret_cast<double*>(dev + off_pose);
  float* d_T_out = reinterpret_cast<float*>(dev + off_T);
  int* d_hits = reinterpret_cast<int*>(dev + off_hits);
  rc = launch_score(ctx, d_T, m, mode, gate_deg, d_scores, nullptr, d_best, st);
  if (rc != PGP_OK) return rc;
  // the walk, the poses it keeps, the best pose and the points it registers: all behind the scores on the stream
  hipLaunchKernelGGL(records_walk, dim3(1), dim3(256), 0, st, m, (const float*)d_scores, (const int*)d_status, list_cap, d_list, d_rs);
  hipLaunchKernelGGL(gather_fits_list, dim3((list_cap + 1 + 3) / 4), dim3(64), 0, st, (const int*)d_list, list_cap, (const int*)d_best,
                     (const float*)d_T, (const double*)d_pose, d_T_out, d_pose_out);
  PGP_HIP(hipGetLastError());
  if (nQ > 0 && (rc = launch_registered(ctx, d_T_out + 16 * C, mode, gate_deg, d_hits, st)) != PGP_OK) return rc;
  unsigned char* pin_out = pin + ((in_bytes + 63) & ~(size_t)63);
  {
    // one kernel brings everything home and writes the word the host polls (pgp::publish_and_wait); the build that was put
    // off goes to its stream first, as before any wait of the host
    if (ctx->deferred_build && (rc = flush_deferred_build(ctx)) != PGP_OK) return rc;
    unsigned char* pin_picks = pin_out + ((home + 63) & ~(size_t)63);
    const bool want_picks = smp.on && smp.picks_out;
    const PubItem items[2] = {{dev + off_best, pin_out, home}, {d_pk, pin_picks, N * 8}};
    const int n_items = want_picks ? 2 : 1;
    if (publish_usable(items, n_items)) {
      if ((rc = publish_and_wait(ctx, st, items, n_items)) != PGP_OK) return rc;
    } else {
      PGP_HIP(hipMemcpyAsync(pin_out, dev + off_best, home, hipMemcpyDeviceToHost, st));
      if (want_picks) PGP_HIP(hipMemcpyAsync(pin_picks, d_pk, N * 8, hipMemcpyDeviceToHost, st));
      PGP_HIP(hipStreamSynchronize(st));
    }
    if (want_picks) std::memcpy(smp.picks_out, pin_picks, N * 8);
  }
  if ((rc = index_settled(ctx)) != PGP_OK) return rc;
  const unsigned char* h = pin_out - off_best;   // (offsets as on the device)
  int best[2], lst[2];
  std::memcpy(best, h + off_best, sizeof best);
  std::memcpy(lst, h + off_list, sizeof lst);
  *n_list = lst[0];
  if (n_pushed) *n_pushed = lst[1];
  if (best_index) *best_index = best[0];
  if (best_score) std::memcpy(best_score, &best[1], 4);
  const size_t n_rec = (size_t)std::min(std::max(lst[0], 0), list_cap);
  if (n_rec) {
    std::memcpy(list_index, h + off_list + 8, n_rec * 4);
    std::memcpy(list_score, h + off_rs, n_rec * 4);
    std::memcpy(list_pose, h + off_pose, n_rec * 128);
    std::memcpy(list_T, h + off_T, n_rec * 64);
  }
  if (best[0] >= 0) {
    if (best_pose) std::memcpy(best_pose, h + off_pose + C * 128, 128);
    if (best_T) std::memcpy(best_T, h + off_T + C * 64, 64);
    if (registered) {
      const int* hits = reinterpret_cast<const int*>(h + off_hits);
      int k = 0;
      for (size_t i = 0; i < nQ; ++i)
        if (hits[i] >= 0) registered[k++] = hits[i];   // model-point order, as push_back in base.cc:1760
      *n_registered = k;
    }
  }
  ctx->csb_fit_m = m;
  return PGP_OK;
}
}  // namespace
}  // namespace pgp

extern "C" {

static pgp_icp_options options_of(const pgp_icp_params* p) {
  pgp_icp_options o;
  pgp_icp_default_options(&o);
  o.max_iterations = p->max_iterations;
  o.trim_fraction = p->trim_fraction;
  o.max_corr_dist = p->max_corr_dist;
  o.energy_ratio = p->energy_ratio > 0.f ? p->energy_ratio : 1.f;   // setNewToOldEnergyRatio(1.f), UCTState.cpp:139
  return o;
}

int pgp_icp_default_options(pgp_icp_options* opt) {
  if (!opt) {
    set_error("pgp_icp_default_options: opt is NULL");
    return PGP_EINVAL;
  }
  opt->max_iterations = 100;
  opt->trim_fraction = 1.f;
  opt->max_corr_dist = 0.f;
  opt->energy_ratio = 1.f;
  opt->error_metric = 0;
  opt->transformation_epsilon = -1.f;
  opt->relative_mse = 0.f;
  opt->absolute_mse = -1.f;
  opt->min_diff_rot = 0.f;
  opt->min_diff_trans = 0.f;
  opt->smooth_length = 0;
  opt->nn_search = 0;
  return PGP_OK;
}

int pgp_icp_refine_ex_device(pgp_ctx* ctx, const float* d_src4, int n_src, const float* d_tgt4,
                             const float* d_tgt_n4, int n_tgt, float* d_T, int n, const pgp_icp_options* opt,
                             float* d_energy, int* d_iters, void* stream) {
  if (!ctx || !opt || n < 0 || n_src < 0 || n_tgt < 0 || (n > 0 && (!d_src4 || !d_tgt4 || !d_T))) {
    set_error("pgp_icp_refine_ex_device: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx, false);
  const int rc = launch_icp(ctx, reinterpret_cast<const float4*>(d_src4), n_src, reinterpret_cast<const float4*>(d_tgt4),
                    reinterpret_cast<const float4*>(d_tgt_n4), n_tgt, d_T, n, opt, d_energy, d_iters,
                    static_cast<hipStream_t>(stream), ctx->icp_user_token);
  note_device_work(ctx, static_cast<hipStream_t>(stream));
  return rc;
}

int pgp_select_top_device(pgp_ctx* ctx, const float* d_T, const float* d_scores, int n, int k, int invert,
                          float* d_T_out, int* d_index_out, int* d_n_out, void* stream) {
  if (!ctx || n < 0 || k < 0 || (k > 0 && (!d_T_out || !d_n_out)) || (n > 0 && (!d_T || !d_scores))) {
    set_error("pgp_select_top_device: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx, false);
  const int rc = launch_select_top(ctx, d_T, d_scores, n, k, invert, d_T_out, d_index_out, d_n_out, static_cast<hipStream_t>(stream));
  note_device_work(ctx, static_cast<hipStream_t>(stream));
  return rc;
}

int pgp_icp_refine_multi_device(const pgp_icp_job* jobs, int n_jobs, const pgp_icp_params* params, void* stream) {
  if (n_jobs < 0 || (n_jobs > 0 && !jobs) || !params) {
    set_error("pgp_icp_refine_multi_device: bad argument");
    return PGP_EINVAL;
  }
  if (n_jobs == 0) return PGP_OK;
  std::vector<IcpJob> v((size_t)n_jobs);
  for (int j = 0; j < n_jobs; ++j) {
    const pgp_icp_job& q = jobs[j];
    if (!q.ctx || q.n < 0 || q.n_src < 0 || q.n_tgt < 0 || (q.n > 0 && (!q.d_src4 || !q.d_tgt4 || !q.d_T))) {
      set_error("pgp_icp_refine_multi_device: bad job %d", j);
      return PGP_EINVAL;
    }
    v[j] = IcpJob{q.ctx, reinterpret_cast<const float4*>(q.d_src4), q.n_src, reinterpret_cast<const float4*>(q.d_tgt4), q.n_tgt,
                  q.d_T, q.n, q.d_energy, q.d_iters, q.ctx->icp_user_token};
  }
  const pgp_icp_options o = options_of(params);
  CtxGuard guard(jobs[0].ctx, false);
  const int rc = launch_icp_multi(v.data(), n_jobs, &o, static_cast<hipStream_t>(stream));
  for (int j = 0; j < n_jobs; ++j) note_device_work(jobs[j].ctx, static_cast<hipStream_t>(stream));
  return rc;
}

int pgp_icp_target_token(pgp_ctx* ctx, unsigned long long token) {
  if (!ctx) {
    set_error("pgp_icp_target_token: ctx is NULL");
    return PGP_EINVAL;
  }
  ctx->icp_user_token = token;
  return PGP_OK;
}

int pgp_icp_refine_device(pgp_ctx* ctx, const float* d_src4, int n_src, const float* d_tgt4, int n_tgt,
                          float* d_T, int n, const pgp_icp_params* params, float* d_energy,
                          int* d_iters, void* stream) {
  if (!params) {
    set_error("pgp_icp_refine_device: bad argument");
    return PGP_EINVAL;
  }
  const pgp_icp_options o = options_of(params);
  return pgp_icp_refine_ex_device(ctx, d_src4, n_src, d_tgt4, nullptr, n_tgt, d_T, n, &o, d_energy, d_iters, stream);
}

}  // extern "C"

namespace pgp {

pgp_icp_options icp_options_of(const pgp_icp_params* p) { return options_of(p); }

namespace {
std::vector<float4> pack4(const float* xyz, int m) {
  std::vector<float4> v((size_t)std::max(m, 1));
  for (int i = 0; i < m; ++i) v[i] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], 0.f);
  return v;
}
// The target is the object's model, the same from call to call (TrimmedICP::init builds its search
// structure once per model: UCTState.cpp:137-139): a hash of its coordinates tells whether the copy
// and the index already resident on the device are this target's, and the upload + build are skipped.
unsigned long long hash_of(const float* xyz, int m) {
  // eight independent multiply-xor chains over the 64-bit words (a single chain over 32-bit words is latency-bound: 20 us
  // at 5000 points; four of them still took ~0.1 ms on the 100 000-point table of the scene alignment, SceneCfg.cpp:135-141)
  const size_t e32 = 3 * (size_t)m, e = e32 / 2;
  const unsigned char* bytes = reinterpret_cast<const unsigned char*>(xyz);
  unsigned long long h[8] = {0x9E3779B97F4A7C15ull ^ (unsigned long long)m, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull,
                             0x85EBCA77C2B2AE63ull, 0xD6E8FEB86659FD93ull, 0xFF51AFD7ED558CCDull, 0xC4CEB9FE1A85EC53ull};
  size_t i = 0;
  for (; i + 8 <= e; i += 8) {
    unsigned long long w[8];
    std::memcpy(w, bytes + 8 * i, 64);
#pragma GCC unroll 8
    for (int k = 0; k < 8; ++k) h[k] = (h[k] ^ w[k]) * 0x100000001B3ull + (h[k] >> (27 + k % 5));
  }
  for (; i < e; ++i) {
    unsigned long long w;
    std::memcpy(&w, bytes + 8 * i, 8);
    h[0] = (h[0] ^ w) * 0x100000001B3ull + (h[0] >> 29);
  }
  if (e32 & 1) {
    uint32_t w;
    std::memcpy(&w, bytes + 4 * (e32 - 1), 4);
    h[1] = (h[1] ^ w) * 0x100000001B3ull + (h[1] >> 31);
  }
  unsigned long long hsh = h[0];
  for (int k = 1; k < 8; ++k) hsh = (hsh ^ (h[k] << (7 * k) | h[k] >> (64 - 7 * k))) * 0x9E3779B97F4A7C15ull;
  return hsh | 1ull;   // never 0
}
}  // namespace

// One host-pointer ICP job staged in the context's buffers on `st` (no launch yet): ONE device block
// [src | T | energy | iters] and its pinned host image -- ONE copy in ([src | T]) and ONE copy out
// ([T | energy | iters]): the caller's arrays are pageable, and five pageable copies cost more than ten
// ICP iterations of one pose.  The target goes to d_icp_tgt unless its hash says it is resident.
int icp_host_stage(pgp_ctx* ctx, const float* src_xyz, int n_src, const float* tgt_xyz, int n_tgt, const float* T, int n,
                   hipStream_t st, IcpHostStage* out) {
  int rc;
  IcpHostStage g{};
  g.off_T = (size_t)std::max(n_src, 1) * 16;
  g.off_e = g.off_T + (size_t)n * 64;
  g.off_i = g.off_e + (size_t)n * 4;
  g.total = g.off_i + (size_t)n * 4;
  if ((rc = ctx->d_icp_src.ensure(g.total)) != PGP_OK) return rc;
  if ((rc = ctx->d_icp_tgt.ensure((size_t)std::max(n_tgt, 1) * 16)) != PGP_OK) return rc;
  if (g.total + 64 > ctx->h_pin_cap) {
    if (ctx->h_pin) {
      hipError_t e = hipHostFree(ctx->h_pin);
      (void)e;
      ctx->h_pin = nullptr;
      ctx->h_pin_cap = 0;
    }
    const size_t want = g.total + g.total / 4 + 64;
    PGP_HIP(hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault));
    ctx->h_pin_cap = want;
  }
  unsigned char* pin = static_cast<unsigned char*>(ctx->h_pin);
  {
    float4* ps = reinterpret_cast<float4*>(pin);
    for (int i = 0; i < n_src; ++i)
      ps[i] = make_float4(src_xyz[3 * (size_t)i], src_xyz[3 * (size_t)i + 1], src_xyz[3 * (size_t)i + 2], 0.f);
    std::memcpy(pin + g.off_T, T, (size_t)n * 64);
  }
  const unsigned long long tok = hash_of(tgt_xyz, n_tgt);
  if (!(tok == ctx->icp_host_token && n_tgt == ctx->icp_host_ntgt)) {
    const std::vector<float4> ht = pack4(tgt_xyz, n_tgt);
    ctx->icp_host_token = 0;
    ctx->icp_idx_valid = false;
    ctx->icp_grid_valid = false;
    PGP_HIP(hipMemcpyAsync(ctx->d_icp_tgt.p, ht.data(), (size_t)n_tgt * 16, hipMemcpyHostToDevice, st));
    PGP_HIP(hipStreamSynchronize(st));   // ht is a temporary
    ctx->icp_host_token = tok;
    ctx->icp_host_ntgt = n_tgt;
  }
  unsigned char* dev = ctx->d_icp_src.as<unsigned char>();
  g.d_src = reinterpret_cast<const float4*>(dev);
  g.d_tgt = ctx->d_icp_tgt.as<float4>();
  g.d_T = reinterpret_cast<float*>(dev + g.off_T);
  g.d_energy = reinterpret_cast<float*>(dev + g.off_e);
  g.d_iters = reinterpret_cast<int*>(dev + g.off_i);
  g.token = tok;
  if ((rc = stage_to_device(st, dev, pin, g.off_e)) != PGP_OK) return rc;
  *out = g;
  return PGP_OK;
}

// the results' way back: queued behind the job's kernels on `st` ...
int icp_host_collect_enqueue(pgp_ctx* ctx, const IcpHostStage& g, hipStream_t st) {
  unsigned char* pin = static_cast<unsigned char*>(ctx->h_pin);
  unsigned char* dev = ctx->d_icp_src.as<unsigned char>();
  PGP_HIP(hipMemcpyAsync(pin + g.off_T, dev + g.off_T, g.total - g.off_T, hipMemcpyDeviceToHost, st));
  return PGP_OK;
}
// ... and, once `st` is synchronised, out of the pinned image into the caller's arrays
void icp_host_collect(pgp_ctx* ctx, const IcpHostStage& g, int n, float* T, float* energy, int* iters) {
  const unsigned char* pin = static_cast<const unsigned char*>(ctx->h_pin);
  std::memcpy(T, pin + g.off_T, (size_t)n * 64);
  if (energy) std::memcpy(energy, pin + g.off_e, (size_t)n * 4);
  if (iters) std::memcpy(iters, pin + g.off_i, (size_t)n * 4);
}

}  // namespace pgp

namespace pgp {
void icp_scene_form_off(bool off);   // icp.hip: this thread's next launch_icp calls take the host-driven scene-sized form
}

extern "C" {

int pgp_icp_refine_ex(pgp_ctx* ctx, const float* src_xyz, int n_src, const float* tgt_xyz, const float* tgt_nrm,
                      int n_tgt, float* T, int n, const pgp_icp_options* opt, float* energy, int* iters) {
  if (!ctx || !opt || n < 0 || n_src < 0 || n_tgt < 0 ||
      (n > 0 && (!T || (n_src > 0 && !src_xyz) || (n_tgt > 0 && !tgt_xyz)))) {
    set_error("pgp_icp_refine_ex: bad argument");
    return PGP_EINVAL;
  }
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  int rc;
  IcpHostStage g{};
  if ((rc = icp_host_stage(ctx, src_xyz, n_src, tgt_xyz, n_tgt, T, n, st, &g)) != PGP_OK) return rc;
  const float4* d_n = nullptr;
  if (tgt_nrm) {
    const unsigned long long ntok = hash_of(tgt_nrm, n_tgt);
    if ((rc = ctx->d_icp_tgt_n.ensure((size_t)std::max(n_tgt, 1) * 16)) != PGP_OK) return rc;
    if (ntok != ctx->icp_host_ntoken) {
      const std::vector<float4> hn = pack4(tgt_nrm, n_tgt);
      ctx->icp_host_ntoken = 0;
      PGP_HIP(hipMemcpyAsync(ctx->d_icp_tgt_n.p, hn.data(), (size_t)n_tgt * 16, hipMemcpyHostToDevice, st));
      PGP_HIP(hipStreamSynchronize(st));
      ctx->icp_host_ntoken = ntok;
    }
    d_n = ctx->d_icp_tgt_n.as<float4>();
  }
  rc = launch_icp(ctx, g.d_src, n_src, g.d_tgt, d_n, n_tgt, g.d_T, n, opt, g.d_energy, g.d_iters, st, g.token);
  if (rc != PGP_OK) return rc;
  // the results' way home: one publishing kernel + the completion word (pgp::publish_and_wait) where they are small and aligned,
  // else a copy and the stream
  auto home = [&]() -> int {
    const PubItem item{ctx->d_icp_src.as<unsigned char>() + g.off_T, static_cast<unsigned char*>(ctx->h_pin) + g.off_T, g.total - g.off_T};
    if (publish_usable(&item, 1)) return publish_and_wait(ctx, st, &item, 1);
    const int r = icp_host_collect_enqueue(ctx, g, st);
    if (r != PGP_OK) return r;
    PGP_HIP(hipStreamSynchronize(st));
    return PGP_OK;
  };
  if ((rc = home()) != PGP_OK) return rc;
  {
    // A pose the scene-sized one-launch form gave up on (iteration count -1: the device was held by somebody else for
    // seconds and the pose's units did not arrive, icp.hip icp_scene_persist) -- the caller's transforms are still
    // untouched: the whole job once more, host-driven.
    const int* it = reinterpret_cast<const int*>(static_cast<const unsigned char*>(ctx->h_pin) + g.off_i);
    bool lost = false;
    for (int i = 0; i < n; ++i) lost = lost || it[i] < 0;
    if (lost) {
      struct Off {
        Off() { pgp::icp_scene_form_off(true); }
        ~Off() { pgp::icp_scene_form_off(false); }
      } off;
      if ((rc = icp_host_stage(ctx, src_xyz, n_src, tgt_xyz, n_tgt, T, n, st, &g)) != PGP_OK) return rc;
      rc = launch_icp(ctx, g.d_src, n_src, g.d_tgt, d_n, n_tgt, g.d_T, n, opt, g.d_energy, g.d_iters, st, g.token);
      if (rc != PGP_OK) return rc;
      if ((rc = home()) != PGP_OK) return rc;
    }
  }
  icp_host_collect(ctx, g, n, T, energy, iters);
  return PGP_OK;
}

int pgp_icp_refine(pgp_ctx* ctx, const float* src_xyz, int n_src, const float* tgt_xyz, int n_tgt,
                   float* T, int n, const pgp_icp_params* params, float* energy, int* iters) {
  if (!params) {
    set_error("pgp_icp_refine: bad argument");
    return PGP_EINVAL;
  }
  const pgp_icp_options o = options_of(params);
  return pgp_icp_refine_ex(ctx, src_xyz, n_src, tgt_xyz, nullptr, n_tgt, T, n, &o, energy, iters);
}

int pgp_radius_outlier_filter(pgp_ctx* ctx, const float* xyz, const float* nrm, int n, float radius,
                              int min_neighbors, unsigned char* keep, float* nrm_out, int* n_kept) {
  if (!ctx || n < 0 || (n > 0 && (!xyz || !keep)) || !(radius > 0.f) || !n_kept) {
    set_error("pgp_radius_outlier_filter: bad argument");
    return PGP_EINVAL;
  }
  *n_kept = 0;
  if (n == 0) return PGP_OK;
  // the scene index with delta = radius answers "points within radius" for the cloud itself
  int rc = pgp_set_scene(ctx, xyz, nullptr, nullptr, n, radius);
  if (rc != PGP_OK) return rc;
  CtxGuard guard(ctx);
  if ((rc = ctx->d_counts.ensure((size_t)n * sizeof(int))) != PGP_OK) return rc;
  rc = launch_count_neighbours(ctx, radius, ctx->d_counts.as<int>(), ctx->stream);
  if (rc != PGP_OK) return rc;
  std::vector<int> k((size_t)n);
  PGP_HIP(hipMemcpyAsync(k.data(), ctx->d_counts.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PGP_HIP(hipStreamSynchronize(ctx->stream));
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    keep[i] = k[i] > min_neighbors ? 1 : 0;  // PCL 1.7: `k <= min_pts_radius_` -> outlier
    kept += keep[i];
    if (nrm && nrm_out) {
      // pcl::flipNormalTowardsViewpoint(p, 0,0,0, n): flip when (vp - p).n < 0, then / magnitude
      const float* p = xyz + 3 * (size_t)i;
      float nx = nrm[3 * (size_t)i], ny = nrm[3 * (size_t)i + 1], nz = nrm[3 * (size_t)i + 2];
      const float vx = 0.f - p[0], vy = 0.f - p[1], vz = 0.f - p[2];
      const float cos_theta = vx * nx + vy * ny + vz * nz;
      if (cos_theta < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
      const float mag = std::sqrt(nx * nx + ny * ny + nz * nz);
      nrm_out[3 * (size_t)i] = nx / mag;
      nrm_out[3 * (size_t)i + 1] = ny / mag;
      nrm_out[3 * (size_t)i + 2] = nz / mag;
    }
  }
  *n_kept = kept;
  return PGP_OK;
}

int pgp_unexplained_segment(pgp_ctx* ctx, const float* seg_xyz, int n, const float* model_xyz, const int* model_offsets,
                            const float* T, int n_objects, float radius, unsigned char* keep, int* n_kept) {
  if (!ctx || n < 0 || n_objects < 0 || (n > 0 && (!seg_xyz || !keep)) || !(radius > 0.f) ||
      (n_objects > 0 && (!model_offsets || !T))) {
    set_error("pgp_unexplained_segment: bad argument");
    return PGP_EINVAL;
  }
  if (n_kept) *n_kept = 0;
  if (n == 0) return PGP_OK;
  int n_model = 0;
  for (int k = 0; k < n_objects; ++k) {
    if (model_offsets[k] < 0 || model_offsets[k + 1] < model_offsets[k]) {
      set_error("pgp_unexplained_segment: model_offsets must be non-decreasing from 0");
      return PGP_EINVAL;
    }
    n_model = model_offsets[k + 1];
  }
  if (n_model > 0 && !model_xyz) {
    set_error("pgp_unexplained_segment: model_xyz is NULL");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t off_m = up((size_t)n * 12), off_o = off_m + up((size_t)std::max(n_model, 1) * 12),
               off_T = off_o + up(((size_t)n_objects + 1) * 4), off_e = off_T + up((size_t)std::max(n_objects, 1) * 64),
               total = off_e + (size_t)n * 4;
  int rc = ctx->d_pre_io.ensure(total);
  if (rc != PGP_OK) return rc;
  unsigned char* d = ctx->d_pre_io.as<unsigned char>();
  PGP_HIP(hipMemcpyAsync(d, seg_xyz, (size_t)n * 12, hipMemcpyHostToDevice, st));
  if (n_objects > 0) {
    if (n_model > 0) PGP_HIP(hipMemcpyAsync(d + off_m, model_xyz, (size_t)n_model * 12, hipMemcpyHostToDevice, st));
    PGP_HIP(hipMemcpyAsync(d + off_o, model_offsets, ((size_t)n_objects + 1) * 4, hipMemcpyHostToDevice, st));
    PGP_HIP(hipMemcpyAsync(d + off_T, T, (size_t)n_objects * 64, hipMemcpyHostToDevice, st));
  }
  rc = launch_explained_points(ctx, reinterpret_cast<const float*>(d), n, reinterpret_cast<const float*>(d + off_m),
                               reinterpret_cast<const int*>(d + off_o), reinterpret_cast<const float*>(d + off_T), n_objects,
                               radius, reinterpret_cast<unsigned int*>(d + off_e), st);
  if (rc != PGP_OK) return rc;
  std::vector<unsigned int> he((size_t)n);
  PGP_HIP(hipMemcpyAsync(he.data(), d + off_e, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    keep[i] = he[i] ? 0 : 1;
    kept += keep[i];
  }
  if (n_kept) *n_kept = kept;
  return PGP_OK;
}

int pgp_backproject_depth(pgp_ctx* ctx, const void* image, int raw16, const unsigned char* mask, int rows,
                          int cols, const float K[9], double z_min, double z_max, float* xyz_out, int cap,
                          int* n_out) {
  if (!ctx || rows < 0 || cols < 0 || cap < 0 || !K || !n_out || (cap > 0 && !xyz_out) ||
      ((size_t)rows * cols > 0 && !image) || (size_t)rows * cols > ((size_t)1 << 30)) {
    set_error("pgp_backproject_depth: bad argument");
    return PGP_EINVAL;
  }
  *n_out = 0;
  const size_t n = (size_t)rows * cols;
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t img_bytes = (n * (raw16 ? 2 : 4) + 255) & ~(size_t)255;
  const size_t mask_bytes = (n + 255) & ~(size_t)255;
  const size_t nb = (n + 255) / 256;
  const size_t ctr_bytes = ((nb + 1) * 4 + 255) & ~(size_t)255;
  const size_t scan_bytes = ((nb / 2048 + 4) * 4 + 255) & ~(size_t)255;
  const size_t xyz_bytes = (size_t)cap * 12;
  int rc = ctx->d_bp.ensure(img_bytes + mask_bytes + ctr_bytes + scan_bytes + xyz_bytes + 256);
  if (rc != PGP_OK) return rc;
  unsigned char* base = ctx->d_bp.as<unsigned char>();
  void* d_img = base;
  unsigned char* d_mask = mask ? base + img_bytes : nullptr;
  uint32_t* d_ctr = reinterpret_cast<uint32_t*>(base + img_bytes + mask_bytes);
  uint32_t* d_scan = reinterpret_cast<uint32_t*>(base + img_bytes + mask_bytes + ctr_bytes);
  float* d_xyz = reinterpret_cast<float*>(base + img_bytes + mask_bytes + ctr_bytes + scan_bytes);
  PGP_HIP(hipMemcpyAsync(d_img, image, n * (raw16 ? 2 : 4), hipMemcpyHostToDevice, st));
  if (mask) PGP_HIP(hipMemcpyAsync(d_mask, mask, n, hipMemcpyHostToDevice, st));
  int total = 0;
  rc = launch_backproject(ctx, d_img, raw16 != 0, d_mask, rows, cols, K, z_min, z_max, d_ctr, d_scan, d_xyz, cap,
                          &total, st);
  if (rc != PGP_OK) return rc;
  *n_out = total;
  const int n_copy = total < cap ? total : cap;
  if (n_copy > 0) {
    HostOut out(ctx, st);
    if ((rc = out.to(xyz_out, d_xyz, (size_t)n_copy * 12)) != PGP_OK || (rc = out.sync()) != PGP_OK) return rc;
  }
  return PGP_OK;
}

int pgp_set_scene_device(pgp_ctx* ctx, const float* d_xyz, const float* d_nrm, const float* d_weight, int n,
                         float delta, void* stream) {
  if (!ctx || n < 0 || (n > 0 && !d_xyz) || !(delta > 0.f) || !std::isfinite(delta)) {
    set_error("pgp_set_scene_device: bad argument (n=%d, delta=%g)", n, (double)delta);
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx, false);
  hipStream_t st = static_cast<hipStream_t>(stream);
  PGP_HIP(hipStreamSynchronize(st));   // the producer of d_xyz
  PGP_HIP(hipDeviceSynchronize());     // queued launches may still read the arrays replaced below
  return set_scene_device(ctx, d_xyz, d_nrm, d_weight, n, delta, st);
}

int pgp_voxel_grid_device(pgp_ctx* ctx, const float* d_xyz, int n, float leaf, float* d_out_xyz, int cap,
                          int* n_out, void* stream) {
  if (!ctx || n < 0 || cap < 0 || !n_out || !(leaf > 0.f) || !std::isfinite(leaf) || (n > 0 && !d_xyz) ||
      (cap > 0 && !d_out_xyz)) {
    set_error("pgp_voxel_grid_device: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx, false);
  return launch_voxel_grid(ctx, d_xyz, n, leaf, d_out_xyz, cap, n_out, static_cast<hipStream_t>(stream));
}

int pgp_voxel_grid(pgp_ctx* ctx, const float* xyz, int n, float leaf, float* out_xyz, int cap, int* n_out) {
  if (!ctx || n < 0 || cap < 0 || !n_out || !(leaf > 0.f) || !std::isfinite(leaf) || (n > 0 && !xyz) ||
      (cap > 0 && !out_xyz)) {
    set_error("pgp_voxel_grid: bad argument");
    return PGP_EINVAL;
  }
  *n_out = 0;
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t in_b = ((size_t)n * 12 + 255) & ~(size_t)255;
  int rc = ctx->d_pre_io.ensure(in_b + (size_t)std::max(cap, 1) * 12 + 64);
  if (rc != PGP_OK) return rc;
  float* d_in = ctx->d_pre_io.as<float>();
  float* d_out = reinterpret_cast<float*>(ctx->d_pre_io.as<unsigned char>() + in_b);
  PGP_HIP(hipMemcpyAsync(d_in, xyz, (size_t)n * 12, hipMemcpyHostToDevice, st));
  rc = launch_voxel_grid(ctx, d_in, n, leaf, d_out, cap, n_out, st);
  if (rc != PGP_OK) return rc;
  const int n_copy = *n_out < cap ? *n_out : cap;
  if (n_copy > 0) {
    HostOut out(ctx, st);
    if ((rc = out.to(out_xyz, d_out, (size_t)n_copy * 12)) != PGP_OK || (rc = out.sync()) != PGP_OK) return rc;
  }
  return PGP_OK;
}

int pgp_mls_normals_device(pgp_ctx* ctx, const float* d_xyz, int n, float radius, float* d_out_xyz, float* d_out_nrm,
                           float* d_out_curvature, int* d_out_index, int cap, int* n_out, void* stream) {
  if (!ctx || n < 0 || cap < 0 || !n_out || !(radius > 0.f) || !std::isfinite(radius) || (n > 0 && !d_xyz) ||
      (cap > 0 && !d_out_xyz)) {
    set_error("pgp_mls_normals_device: bad argument");
    return PGP_EINVAL;
  }
  *n_out = 0;
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx, false);
  return launch_mls(ctx, d_xyz, n, radius, d_out_xyz, d_out_nrm, d_out_curvature, d_out_index, cap, n_out,
                    static_cast<hipStream_t>(stream));
}

int pgp_mls_normals(pgp_ctx* ctx, const float* xyz, int n, float radius, float* out_xyz, float* out_nrm,
                    float* out_curvature, int* out_index, int cap, int* n_out) {
  if (!ctx || n < 0 || cap < 0 || !n_out || !(radius > 0.f) || !std::isfinite(radius) || (n > 0 && !xyz) ||
      (cap > 0 && !out_xyz)) {
    set_error("pgp_mls_normals: bad argument");
    return PGP_EINVAL;
  }
  *n_out = 0;
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t rows = (size_t)std::max(cap, 1);
  const size_t b_in = ((size_t)n * 12 + 255) & ~(size_t)255, b_3 = (rows * 12 + 255) & ~(size_t)255,
               b_1 = (rows * 4 + 255) & ~(size_t)255;
  int rc = ctx->d_pre_io.ensure(b_in + 2 * b_3 + 2 * b_1);
  if (rc != PGP_OK) return rc;
  unsigned char* base = ctx->d_pre_io.as<unsigned char>();
  float* d_in = reinterpret_cast<float*>(base);
  float* d_x = reinterpret_cast<float*>(base + b_in);
  float* d_n = reinterpret_cast<float*>(base + b_in + b_3);
  float* d_c = reinterpret_cast<float*>(base + b_in + 2 * b_3);
  int* d_i = reinterpret_cast<int*>(base + b_in + 2 * b_3 + b_1);
  PGP_HIP(hipMemcpyAsync(d_in, xyz, (size_t)n * 12, hipMemcpyHostToDevice, st));
  rc = launch_mls(ctx, d_in, n, radius, d_x, d_n, d_c, d_i, cap, n_out, st);
  if (rc != PGP_OK) return rc;
  const size_t m = (size_t)(*n_out < cap ? *n_out : cap);
  if (m > 0) {
    // through the context's pinned landing area, one wait for all four (a copy into the caller's pageable array is a
    // synchronisation of its own: four of them were 160 us of a 0.52 ms call, tools/call_timeline.sh mls)
    HostOut out(ctx, st);
    if ((rc = out.to(out_xyz, d_x, m * 12)) != PGP_OK) return rc;
    if (out_nrm && (rc = out.to(out_nrm, d_n, m * 12)) != PGP_OK) return rc;
    if (out_curvature && (rc = out.to(out_curvature, d_c, m * 4)) != PGP_OK) return rc;
    if (out_index && (rc = out.to(out_index, d_i, m * 4)) != PGP_OK) return rc;
    if ((rc = out.sync()) != PGP_OK) return rc;
  }
  return PGP_OK;
}

int pgp_pose_hausdorff(pgp_ctx* ctx, const float* hull_xyz, int n_hull, const float* T, int n_poses,
                       const int* pairs, int m, float* dist_max, float* dist_sum) {
  if (!ctx || n_hull < 0 || n_poses < 0 || m < 0 || (n_hull > 0 && !hull_xyz) || (n_poses > 0 && !T) ||
      (m > 0 && (!pairs || !dist_max))) {
    set_error("pgp_pose_hausdorff: bad argument");
    return PGP_EINVAL;
  }
  if (m == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  std::vector<float4> hh((size_t)std::max(n_hull, 1));
  for (int i = 0; i < n_hull; ++i) hh[i] = make_float4(hull_xyz[3 * (size_t)i], hull_xyz[3 * (size_t)i + 1], hull_xyz[3 * (size_t)i + 2], 0.f);
  const size_t b_h = (hh.size() * 16 + 255) & ~(size_t)255, b_T = ((size_t)std::max(n_poses, 1) * 64 + 255) & ~(size_t)255,
               b_p = ((size_t)m * 8 + 255) & ~(size_t)255, b_o = ((size_t)m * 4 + 255) & ~(size_t)255;
  int rc = ctx->d_pre_io.ensure(b_h + b_T + b_p + 2 * b_o);
  if (rc != PGP_OK) return rc;
  unsigned char* base = ctx->d_pre_io.as<unsigned char>();
  float4* d_h = reinterpret_cast<float4*>(base);
  float* d_T = reinterpret_cast<float*>(base + b_h);
  int2* d_p = reinterpret_cast<int2*>(base + b_h + b_T);
  float* d_mx = reinterpret_cast<float*>(base + b_h + b_T + b_p);
  float* d_sm = reinterpret_cast<float*>(base + b_h + b_T + b_p + b_o);
  PGP_HIP(hipMemcpyAsync(d_h, hh.data(), (size_t)n_hull * 16, hipMemcpyHostToDevice, st));
  if (n_poses > 0) PGP_HIP(hipMemcpyAsync(d_T, T, (size_t)n_poses * 64, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(d_p, pairs, (size_t)m * 8, hipMemcpyHostToDevice, st));
  rc = launch_pose_hausdorff(ctx, d_h, n_hull, d_T, n_poses, d_p, m, d_mx, d_sm, st);
  if (rc != PGP_OK) return rc;
  {
    HostOut out(ctx, st);
    int rc_out;
    if ((rc_out = out.to(dist_max, d_mx, (size_t)m * 4)) != PGP_OK || (dist_sum && (rc_out = out.to(dist_sum, d_sm, (size_t)m * 4)) != PGP_OK) ||
        (rc_out = out.sync()) != PGP_OK)   // (the wait also covers hh, a stack-owned staging vector)
      return rc_out;
  }
  return PGP_OK;
}

int pgp_backproject_depth_device(pgp_ctx* ctx, const void* d_image, int raw16, const unsigned char* d_mask,
                                 int rows, int cols, const float K[9], double z_min, double z_max,
                                 float* d_xyz_out, int cap, int* n_out, void* stream) {
  if (!ctx || rows < 0 || cols < 0 || cap < 0 || !K || !n_out || (cap > 0 && !d_xyz_out) ||
      ((size_t)rows * cols > 0 && !d_image) || (size_t)rows * cols > ((size_t)1 << 30)) {
    set_error("pgp_backproject_depth_device: bad argument");
    return PGP_EINVAL;
  }
  *n_out = 0;
  const size_t n = (size_t)rows * cols;
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx, false);
  const size_t nb = (n + 255) / 256;
  const size_t ctr_bytes = ((nb + 1) * 4 + 255) & ~(size_t)255;
  const size_t scan_bytes = ((nb / 2048 + 4) * 4 + 255) & ~(size_t)255;
  int rc = ctx->d_bp.ensure(ctr_bytes + scan_bytes + 256);
  if (rc != PGP_OK) return rc;
  uint32_t* d_ctr = ctx->d_bp.as<uint32_t>();
  uint32_t* d_scan = reinterpret_cast<uint32_t*>(ctx->d_bp.as<unsigned char>() + ctr_bytes);
  return launch_backproject(ctx, d_image, raw16 != 0, d_mask, rows, cols, K, z_min, z_max, d_ctr, d_scan, d_xyz_out, cap,
                            n_out, static_cast<hipStream_t>(stream));
}

int pgp_cluster_poses_device(pgp_ctx* ctx, const float* d_T, const float* d_scores, int n_h, float best_score,
                             const float sym_deg[3], const pgp_cluster_params* params, int* d_rep_index,
                             int* d_assignment, int* n_rep, void* stream) {
  if (!ctx || n_h < 0 || !n_rep || !sym_deg || (n_h > 0 && (!d_T || !d_scores || !d_rep_index || !d_assignment))) {
    set_error("pgp_cluster_poses_device: bad argument");
    return PGP_EINVAL;
  }
  *n_rep = 0;
  if (n_h == 0) return PGP_OK;
  const pgp_cluster_params dflt = {0.5f, 10.f, 0.02f};  // HypothesisSelection.cpp:70,99
  if (!params) params = &dflt;
  CtxGuard guard(ctx, false);
  int m = 0;
  return launch_cluster(ctx, d_T, d_scores, n_h, best_score, sym_deg, params, d_rep_index, d_assignment, &m, n_rep,
                        static_cast<hipStream_t>(stream));
}

int pgp_depth_cost(pgp_ctx* ctx, const float* observed, const float* rendered, int n, int rows, int cols,
                   float threshold, float* render_score, int* counts) {
  if (!ctx || n < 0 || rows < 0 || cols < 0 || (n > 0 && (!observed || !rendered || !render_score))) {
    set_error("pgp_depth_cost: bad argument");
    return PGP_EINVAL;
  }
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t npix = (size_t)rows * cols;
  const size_t img_bytes = (npix * 4 + 15) & ~(size_t)15;
  int rc = ctx->d_depth.ensure(img_bytes * ((size_t)n + 1) + (size_t)n * 12 + 64);
  if (rc != PGP_OK) return rc;
  float* d_obs = ctx->d_depth.as<float>();
  float* d_ren = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(d_obs) + img_bytes);
  int* d_counts = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(d_ren) + (size_t)n * npix * 4 + 16);
  d_counts = reinterpret_cast<int*>(((uintptr_t)d_counts + 15) & ~(uintptr_t)15);
  if (npix > 0) {
    PGP_HIP(hipMemcpyAsync(d_obs, observed, npix * 4, hipMemcpyHostToDevice, st));
    PGP_HIP(hipMemcpyAsync(d_ren, rendered, (size_t)n * npix * 4, hipMemcpyHostToDevice, st));
  }
  rc = launch_depth_cost(ctx, d_obs, d_ren, n, (int)npix, threshold, d_counts, st);
  if (rc != PGP_OK) return rc;
  std::vector<int> hc((size_t)n * 3);
  PGP_HIP(hipMemcpyAsync(hc.data(), d_counts, hc.size() * 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) {
    render_score[i] = (float)hc[3 * i] + (float)hc[3 * i + 1] - (float)hc[3 * i + 2];  // UCTState.cpp:115
    if (counts) {
      counts[3 * i] = hc[3 * i];
      counts[3 * i + 1] = hc[3 * i + 1];
      counts[3 * i + 2] = hc[3 * i + 2];
    }
  }
  return PGP_OK;
}

int pgp_depth_cost_device(pgp_ctx* ctx, const float* d_observed, const float* d_rendered, int n, int rows, int cols,
                          float threshold, int* d_counts, float* d_scores, void* stream) {
  if (!ctx || n < 0 || rows < 0 || cols < 0 || (n > 0 && (!d_observed || !d_rendered || !d_counts))) {
    set_error("pgp_depth_cost_device: bad argument");
    return PGP_EINVAL;
  }
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx, false);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = launch_depth_cost(ctx, d_observed, d_rendered, n, (int)((size_t)rows * cols), threshold, d_counts, st);
  if (rc == PGP_OK && d_scores) rc = launch_cost_scores(d_counts, n, d_scores, st);
  note_device_work(ctx, st);
  return rc;
}

int pgp_render_depth_device(pgp_ctx* ctx, const float* d_vertices, int vertex_stride, int n_vert, const int* d_triangles,
                            int n_tri, const float* d_T, int n, const pgp_camera* cam, const float* d_parent,
                            size_t parent_stride, float* d_depth, void* stream) {
  if (!ctx || !cam || n < 0 || n_vert < 0 || n_tri < 0 || (n > 0 && (!d_T || !d_depth)) || (n_vert > 0 && !d_vertices)) {
    set_error("pgp_render_depth_device: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx, false);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc = launch_render_depth(ctx, d_vertices, vertex_stride, n_vert, d_triangles, n_tri, d_T, n, cam, d_parent,
                                     parent_stride, d_depth, st);
  note_device_work(ctx, st);
  return rc;
}

int pgp_render_depth(pgp_ctx* ctx, const float* vertices, int n_vert, const int* triangles, int n_tri, const float* T,
                     int n, const pgp_camera* cam, const float* parent, float* depth) {
  if (!ctx || !cam || n < 0 || n_vert < 0 || n_tri < 0 || (n > 0 && (!T || !depth)) || (n_vert > 0 && !vertices) ||
      cam->rows <= 0 || cam->cols <= 0) {
    set_error("pgp_render_depth: bad argument");
    return PGP_EINVAL;
  }
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t n_pix = (size_t)cam->rows * cam->cols;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t off_t = up((size_t)std::max(n_vert, 1) * 12), off_T = off_t + up((size_t)std::max(n_tri, 1) * 12),
               off_p = off_T + up((size_t)n * 64), off_o = off_p + up(n_pix * 4), total = off_o + n_pix * 4 * (size_t)n;
  int rc = ctx->d_render_io.ensure(total);
  if (rc != PGP_OK) return rc;
  unsigned char* d = ctx->d_render_io.as<unsigned char>();
  if (n_vert > 0) PGP_HIP(hipMemcpyAsync(d, vertices, (size_t)n_vert * 12, hipMemcpyHostToDevice, st));
  if (triangles && n_tri > 0) PGP_HIP(hipMemcpyAsync(d + off_t, triangles, (size_t)n_tri * 12, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(d + off_T, T, (size_t)n * 64, hipMemcpyHostToDevice, st));
  if (parent) PGP_HIP(hipMemcpyAsync(d + off_p, parent, n_pix * 4, hipMemcpyHostToDevice, st));
  rc = launch_render_depth(ctx, reinterpret_cast<const float*>(d), 3, n_vert,
                           triangles ? reinterpret_cast<const int*>(d + off_t) : nullptr, n_tri,
                           reinterpret_cast<const float*>(d + off_T), n, cam,
                           parent ? reinterpret_cast<const float*>(d + off_p) : nullptr, 0,
                           reinterpret_cast<float*>(d + off_o), st);
  if (rc != PGP_OK) return rc;
  PGP_HIP(hipMemcpyAsync(depth, d + off_o, n_pix * 4 * (size_t)n, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  return PGP_OK;
}

int pgp_cluster_poses(pgp_ctx* ctx, const float* T, const float* scores, int n_h, float best_score,
                      const float sym_deg[3], const pgp_cluster_params* params, int* rep_index, int cap,
                      int* n_rep, int* assignment) {
  if (!ctx || n_h < 0 || cap < 0 || !n_rep || !sym_deg || (n_h > 0 && (!T || !scores)) || (cap > 0 && !rep_index)) {
    set_error("pgp_cluster_poses: bad argument");
    return PGP_EINVAL;
  }
  *n_rep = 0;
  if (n_h == 0) return PGP_OK;
  const pgp_cluster_params dflt = {0.5f, 10.f, 0.02f};  // HypothesisSelection.cpp:70,99
  if (!params) params = &dflt;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t nT = (size_t)n_h * 64, nS = ((size_t)n_h * 4 + 63) & ~(size_t)63;
  int rc = ctx->d_cl_io.ensure(nT + 3 * nS);
  if (rc != PGP_OK) return rc;
  unsigned char* base = ctx->d_cl_io.as<unsigned char>();
  float* d_T = reinterpret_cast<float*>(base);
  float* d_s = reinterpret_cast<float*>(base + nT);
  int* d_rep = reinterpret_cast<int*>(base + nT + nS);
  int* d_assign = reinterpret_cast<int*>(base + nT + 2 * nS);
  PGP_HIP(hipMemcpyAsync(d_T, T, nT, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(d_s, scores, (size_t)n_h * 4, hipMemcpyHostToDevice, st));
  int m = 0, nr = 0;
  rc = launch_cluster(ctx, d_T, d_s, n_h, best_score, sym_deg, params, d_rep, d_assign, &m, &nr, st);
  if (rc != PGP_OK) return rc;
  *n_rep = nr;
  const int n_copy = nr < cap ? nr : cap;
  HostOut out(ctx, st);
  if (n_copy > 0 && (rc = out.to(rep_index, d_rep, (size_t)n_copy * 4)) != PGP_OK) return rc;
  if (assignment && (rc = out.to(assignment, d_assign, (size_t)n_h * 4)) != PGP_OK) return rc;
  return out.sync();
}

int pgp_pose_error(pgp_ctx* ctx, const float* test, const float* gt, int n, const float sym_deg[3],
                   float* rot_err_deg, float* trans_err) {
  if (!ctx || n < 0 || !sym_deg || (n > 0 && (!test || !gt || !rot_err_deg || !trans_err))) {
    set_error("pgp_pose_error: bad argument");
    return PGP_EINVAL;
  }
  if (n == 0) return PGP_OK;
  CtxGuard guard(ctx);
  hipStream_t st = ctx->stream;
  const size_t nT = (size_t)n * 64, nS = ((size_t)n * 4 + 63) & ~(size_t)63;
  int rc = ctx->d_cl_io.ensure(2 * nT + 2 * nS);
  if (rc != PGP_OK) return rc;
  unsigned char* base = ctx->d_cl_io.as<unsigned char>();
  float* d_a = reinterpret_cast<float*>(base);
  float* d_g = reinterpret_cast<float*>(base + nT);
  float* d_r = reinterpret_cast<float*>(base + 2 * nT);
  float* d_t = reinterpret_cast<float*>(base + 2 * nT + nS);
  PGP_HIP(hipMemcpyAsync(d_a, test, nT, hipMemcpyHostToDevice, st));
  PGP_HIP(hipMemcpyAsync(d_g, gt, nT, hipMemcpyHostToDevice, st));
  rc = launch_pose_error(ctx, d_a, d_g, n, sym_deg, d_r, d_t, st);
  if (rc != PGP_OK) return rc;
  {
    HostOut out(ctx, st);
    int rc_out;
    if ((rc_out = out.to(rot_err_deg, d_r, (size_t)n * 4)) != PGP_OK || (rc_out = out.to(trans_err, d_t, (size_t)n * 4)) != PGP_OK ||
        (rc_out = out.sync()) != PGP_OK)
      return rc_out;
  }
  return PGP_OK;
}

int pgp_set_kernel_timing(pgp_ctx* ctx, int enable) {
  if (!ctx) {
    set_error("pgp_set_kernel_timing: ctx is NULL");
    return PGP_EINVAL;
  }
  ctx->timing = enable > 0 ? enable : 0;
  ctx->timing_seq = 0;
  return PGP_OK;
}

int pgp_get_kernel_timing(pgp_ctx* ctx, int* launches, float* total_ms, int reset) {
  if (!ctx || !launches || !total_ms) {
    set_error("pgp_get_kernel_timing: bad argument");
    return PGP_EINVAL;
  }
  CtxGuard guard(ctx);
  float sum = 0.f;
  for (size_t k = 0; k + 1 < ctx->ev_used; k += 2) {
    PGP_HIP(hipEventSynchronize(ctx->ev[k + 1]));
    float ms = 0.f;
    PGP_HIP(hipEventElapsedTime(&ms, ctx->ev[k], ctx->ev[k + 1]));
    sum += ms;
  }
  *launches = (int)(ctx->ev_used / 2);
  *total_ms = sum;
  if (reset) ctx->ev_used = 0;
  return PGP_OK;
}

int pgp_get_index_info(pgp_ctx* ctx, pgp_index_info* info) {
  if (!ctx || !info) {
    set_error("pgp_get_index_info: bad argument");
    return PGP_EINVAL;
  }
  std::memset(info, 0, sizeof *info);
  info->n_scene = ctx->nP;
  info->n_model = ctx->nQ;
  if (ctx->has_index) {
    CtxGuard guard(ctx);
    const int rc = finish_index(ctx);   // a build on the side stream: its counts are known when it is through
    if (rc != PGP_OK) return rc;
    info->grid_nx = ctx->grid.nx;
    info->grid_ny = ctx->grid.ny;
    info->grid_nz = ctx->grid.nz;
    info->cell_size = ctx->grid.h;
    info->delta = ctx->delta;
    info->n_cells = ctx->n_cells;
    info->n_candidates = ctx->n_cand;
    info->n_occupied = ctx->n_occ;
    const size_t table = ctx->grid.sparse ? ((size_t)ctx->grid.tab_mask + 1) * 16
                                          : (size_t)ctx->grid.nbx * ctx->grid.nby * ctx->grid.nbz * 8;
    info->bytes_index = (long long)(table + (size_t)ctx->n_occ * 8 + (size_t)ctx->n_cand * 16);
    info->sparse = ctx->grid.sparse;
    info->n_blocks = ctx->grid.sparse ? ctx->n_blocks : (long long)ctx->grid.nbx * ctx->grid.nby * ctx->grid.nbz;
    info->build_ms = ctx->build_ms;
  }
  return PGP_OK;
}

}  // extern "C"
