// csrc/icp.hip -- batched (trimmed) point-to-point ICP refinement on gfx950.
//
// Replaces the ICP inner loop the reference reaches through PCL / libpointmatcher:
//   pcl::recognition::TrimmedICP::align    PPE/hypothesis_verification/mcts/UCTState.cpp:137-139,194
//                                          PPE/misc/utilities.cpp:666-676
//   pcl::IterativeClosestPoint::align      PPE/misc/utilities.cpp:697-703, PPE/data_layer/SceneCfg.cpp:101,135-141
// PCL and libpointmatcher are not vendored in the reference (version unpinned, sources
// absent), so there is no reference arithmetic to reproduce: this file implements the published
// algorithm with the parameters the call sites set, and oracle/pgp_oracle.c restates the same
// definition on the CPU ("parity unpinned" against PCL; pinned against our own oracle):
//
//   G <- initial guess (source frame -> target frame);  E_old <- +inf
//   do   x_i = G s_i ; (j_i, d2_i) = nearest target point of x_i (exhaustive, ties: lowest j)
//        S   = the k = |trim * n| source points with smallest d2 (ties: lowest i), or all with
//              d2 <= max_corr^2 when a correspondence cap is set
//        E   = mean_{i in S} d2_i   (|S| = k is constant when trimming: same test as PCL's sum)
//        G   <- argmin_G sum_{i in S} |G s_i - m_{j_i}|^2      (Horn 1987, closed form)
//   while (E / E_old < ratio, E_old <- E, iterations < max)     [ratio = 1: UCTState.cpp:139]
//
// Mapping to the machine, two forms with identical results:
//   split (default)  per iteration, icp_nn_split searches the correspondences with a grid of
//                    (source chunks x target chunks x poses) workgroups -- 4 source points per lane
//                    as two packed-fp32 register pairs, a 512-point target chunk in LDS read four
//                    targets per trip, the bound seeded from the previous iteration, one 64-bit
//                    atomic-min key per source point -- and icp_refine<true> (one workgroup per pose) selects,
//                    reduces and solves; the host enqueues the iterations and tests a device
//                    counter every fourth one;
//   persistent       icp_refine<false>: ONE WORKGROUP (1024 threads) PER POSE runs every iteration
//                    itself (no host involvement, graph-capturable), below.
//   * NN search is a tiled exhaustive scan: a tile of 4096 target points is staged in LDS (64 KB)
//     with coalesced 16-B loads, every lane keeps R = 4 transformed source points in VGPRs and
//     reads each target point once as an LDS broadcast (ds_read_b128, same address in all lanes:
//     conflict-free) => 256 distance tests per LDS read per wave; VALU-bound by design.
//   * trimming is an exact radix select on the float bits of d2 (4 passes x 256-bin LDS
//     histogram), deterministic tie handling by an ordered block scan;
//   * the 3x3 cross-covariance and centroids are accumulated in f64 per thread and reduced
//     across the block in a fixed tree (bit-reproducible), then one lane solves Horn's 4x4
//     symmetric eigenproblem with cyclic Jacobi sweeps.
// Algorithmic bytes per pose-iteration: 12|src| + 12|tgt| + 48 + 64.
//
// Variants the other call sites need (pgp_icp_options, all on the same kernels):
//   * stop rules of pcl::registration::DefaultConvergenceCriteria as IterativeClosestPoint sets them
//     (greedy_bfs/State.cpp:139-142: max_corr, 50 iterations, transformation epsilon 1e-8;
//     utilities.cpp:697-703: 100 iterations, defaults): the update of an iteration is tested for
//     cos(angle) >= 1 - eps and |t|^2 <= eps, and the mean squared correspondence distance for a
//     relative / absolute change below a threshold;
//   * point-to-plane (pcl::IterativeClosestPointWithNormals, utilities.cpp:709-739): the linearised
//     least squares of TransformationEstimationPointToPlaneLLS -- 6 x 6 normal equations accumulated
//     in f64 over the selected pairs, solved by Gaussian elimination with partial pivoting, the
//     update built from the three angles and applied on the left of the current transform;
//   * libpointmatcher's chain (utilities.cpp:744-838): exact nearest neighbour (its kd-tree runs with
//     epsilon 3.16, an approximation this library does not make), TrimmedDistOutlierFilter 0.75 =
//     trim_fraction, DifferentialTransformationChecker = smoothed rotation / translation change;
//   * a uniform-grid nearest-neighbour search (icp_nn_grid) when a correspondence cap is set and the
//     clouds are scene-sized (PPE/data_layer/SceneCfg.cpp:101,135-141: table ICP, max_corr 0.01):
//     cells of edge max_corr, 27 cells per query, identical (d2, lowest j) results to the scan.

#include "pgp_internal.h"

#include <cfloat>
#include <cstdlib>
#include <vector>

namespace pgp {

namespace {

constexpr int kIcpThreads = 1024;
constexpr int kIcpR = 4;                 // source points per lane per sweep
constexpr int kTgtTile = 4096;           // target points per LDS tile (64 KB)
constexpr int kRedPlane = 28;            // point-to-plane: count + 21 (upper triangle of AtA) + 6 (Atb)
constexpr int kMaxSmooth = 8;            // history length of the differential checker

struct IcpArgs {
  const float4* src;   // [n_src] {x,y,z,-}
  const float4* tgt;   // [n_tgt]
  int n_src, n_tgt;
  float* T;            // [n][16] in/out, column-major
  int n;
  int max_iter, k_trim;
  float max_corr2;     // < 0: unlimited
  float ratio;
  float* ws_d2;        // [n][n_src]
  int* ws_j;           // [n][n_src]
  float* energy;       // [n] (nullable)
  int* iters;          // [n] (nullable)
  // split path (few poses): correspondences come from icp_nn_split through 64-bit keys
  unsigned long long* ws_key;  // [n][n_src]  (d2 bits << 32) | j, ~0 = none
  double* st_E;        // [n] previous mean squared distance
  int* st_it;          // [n] iterations done
  int* st_done;        // [n] 1 = converged / stopped
  int* n_done;         // [1]
  // variants (pgp_icp_options)
  const float4* tgt_n; // target normals {nx,ny,nz,-} (point-to-plane), nullable
  int metric;          // 0 point-to-point, 1 point-to-plane
  float t_eps;         // >= 0: stop when the update has cos(angle) >= 1 - eps and |t|^2 <= eps
  float rel_mse;       // > 0: stop when |E - E_old| / E_old < rel_mse
  float abs_mse;       // >= 0: stop when |E - E_old| < abs_mse
  float diff_rot, diff_trans;  // > 0: libpointmatcher DifferentialTransformationChecker thresholds
  int smooth;          // its smoothLength (1..kMaxSmooth)
  double* st_hist;     // [n][kMaxSmooth + 1][7]: quaternion + translation of the last absolute transforms
  // grid search (icp_nn_grid)
  float gox, goy, goz, ginv_h;
  int gnx, gny, gnz;
  const uint32_t* gcell_start;   // [cells + 1]
  const float4* gpts;            // target points sorted by cell, .w = bits(original index)
};

__device__ __forceinline__ float row_xf(float a, float b, float c, float t, float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, x), __fmul_rn(b, y)), __fmul_rn(c, z)), t);
}

// cyclic Jacobi on a symmetric 4x4 (double); returns the eigenvector of the largest eigenvalue.
// The rotation planes (p, r) and the inner index are fully unrolled, so every A[..][..] / V[..][..] has
// compile-time indices and the two matrices live in VGPRs (run-time indexed private arrays would go
// to scratch, cdna_hip_programming.md rule 20; round 1 kept them in LDS, where the single working
// lane paid an LDS round trip per element: ~60 of the 96 us of an icp_refine launch).  Same
// operations in the same order as before (and as oracle/pgp_oracle.c jacobi_eig4).
__device__ void largest_eigvec4(double (&A)[4][4], double q[4]) {
  double V[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) V[p][r] = p == r ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 16; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = p + 1; r < 4; ++r) off += A[p][r] * A[p][r];
    if (off < 1e-300) break;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = p + 1; r < 4; ++r) {
        const double apr = A[p][r];
        if (apr != 0.0) {
          const double theta = (A[r][r] - A[p][p]) / (2.0 * apr);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
          for (int k = 0; k < 4; ++k) {  // A <- A J
            const double akp = A[k][p], akr = A[k][r];
            A[k][p] = c * akp - sn * akr;
            A[k][r] = sn * akp + c * akr;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {  // A <- J^T A
            const double apk = A[p][k], ark = A[r][k];
            A[p][k] = c * apk - sn * ark;
            A[r][k] = sn * apk + c * ark;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double vkp = V[k][p], vkr = V[k][r];
            V[k][p] = c * vkp - sn * vkr;
            V[k][r] = sn * vkp + c * vkr;
          }
        }
      }
  }
  int best = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const double ak = k == 1 ? A[1][1] : (k == 2 ? A[2][2] : A[3][3]);
    const double ab = best == 0 ? A[0][0] : (best == 1 ? A[1][1] : (best == 2 ? A[2][2] : A[3][3]));
    if (ak > ab) best = k;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = best == 0 ? V[k][0] : (best == 1 ? V[k][1] : (best == 2 ? V[k][2] : V[k][3]));
}

// Horn's closed form from the f64 sums over the selected pairs:
// red = {n, sx,sy,sz, mx,my,mz, Sxx,Sxy,Sxz, Syx,Syy,Syz, Szx,Szy,Szz}  (S_ab = sum s_a m_b)
__device__ void solve_rigid(const double* red, float* G) {
  double n = red[0];
  if (!(n >= 1.0)) return;  // nothing selected: keep G
  double sb[3] = {red[1] / n, red[2] / n, red[3] / n}, mb[3] = {red[4] / n, red[5] / n, red[6] / n};
  const double Sxx = red[7] - n * sb[0] * mb[0], Sxy = red[8] - n * sb[0] * mb[1], Sxz = red[9] - n * sb[0] * mb[2];
  const double Syx = red[10] - n * sb[1] * mb[0], Syy = red[11] - n * sb[1] * mb[1], Syz = red[12] - n * sb[1] * mb[2];
  const double Szx = red[13] - n * sb[2] * mb[0], Szy = red[14] - n * sb[2] * mb[1], Szz = red[15] - n * sb[2] * mb[2];
  double N[4][4];
  N[0][0] = Sxx + Syy + Szz; N[0][1] = Syz - Szy;       N[0][2] = Szx - Sxz;        N[0][3] = Sxy - Syx;
  N[1][0] = Syz - Szy;       N[1][1] = Sxx - Syy - Szz; N[1][2] = Sxy + Syx;        N[1][3] = Szx + Sxz;
  N[2][0] = Szx - Sxz;       N[2][1] = Sxy + Syx;       N[2][2] = -Sxx + Syy - Szz; N[2][3] = Syz + Szy;
  N[3][0] = Sxy - Syx;       N[3][1] = Szx + Sxz;       N[3][2] = Syz + Szy;        N[3][3] = -Sxx - Syy + Szz;
  double q[4];
  largest_eigvec4(N, q);
  double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (!(nq > 0.0)) return;
  double w = q[0] / nq, x = q[1] / nq, y = q[2] / nq, z = q[3] / nq;
  double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                    {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                    {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double t = mb[r] - (R[r][0] * sb[0] + R[r][1] * sb[1] + R[r][2] * sb[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) G[4 * c + r] = (float)R[r][c];
    G[12 + r] = (float)t;
  }
  G[3] = G[7] = G[11] = 0.f;
  G[15] = 1.f;
}

// Point-to-plane update (pcl::registration::TransformationEstimationPointToPlaneLLS): the sums are
// over the selected pairs with x = G s (the source as currently placed), m its target point, n the
// target normal: row = (n x x ... precisely a = nz xy - ny xz, b = nx xz - nz xx, c = ny xx - nx xy,
// nx, ny, nz), rhs = n . (m - x).  red = {count, upper triangle of AtA (21, row-major), Atb (6)}.
// Solves AtA p = Atb, builds the update from (alpha, beta, gamma, tx, ty, tz) and sets G <- D G.
__device__ void solve_plane(const double* red, float* G) {
  if (!(red[0] >= 3.0)) return;
  double A[6][7];
  int t = 1;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) {
      A[r][c] = red[t];
      A[c][r] = red[t];
      ++t;
    }
  for (int r = 0; r < 6; ++r) A[r][6] = red[22 + r];
  for (int col = 0; col < 6; ++col) {   // Gaussian elimination, partial pivoting
    int piv = col;
    for (int r = col + 1; r < 6; ++r)
      if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
    if (!(fabs(A[piv][col]) > 1e-300)) return;   // singular: keep G
    if (piv != col)
      for (int c = 0; c < 7; ++c) {
        const double tmp = A[col][c];
        A[col][c] = A[piv][c];
        A[piv][c] = tmp;
      }
    for (int r = col + 1; r < 6; ++r) {
      const double f = A[r][col] / A[col][col];
      for (int c = col; c < 7; ++c) A[r][c] -= f * A[col][c];
    }
  }
  double x[6];
  for (int r = 5; r >= 0; --r) {
    double v = A[r][6];
    for (int c = r + 1; c < 6; ++c) v -= A[r][c] * x[c];
    x[r] = v / A[r][r];
  }
  const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
  const double D[3][4] = {{cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca, x[3]},
                          {sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca, x[4]},
                          {-sb, cb * sa, cb * ca, x[5]}};
  double Gn[3][4];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      double v = D[r][0] * (double)G[4 * c] + D[r][1] * (double)G[4 * c + 1] + D[r][2] * (double)G[4 * c + 2];
      if (c == 3) v += D[r][3];
      Gn[r][c] = v;
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) G[4 * c + r] = (float)Gn[r][c];
  G[3] = G[7] = G[11] = 0.f;
  G[15] = 1.f;
}

// rotation part of a column-major 4x4 (float) as a unit quaternion (w, x, y, z), double
__device__ void quat_of(const float* G, double q[4]) {
  const double m00 = G[0], m10 = G[1], m20 = G[2], m01 = G[4], m11 = G[5], m21 = G[6], m02 = G[8], m12 = G[9], m22 = G[10];
  const double tr = m00 + m11 + m22;
  if (tr > 0.0) {
    const double s = sqrt(tr + 1.0) * 2.0;
    q[0] = 0.25 * s; q[1] = (m21 - m12) / s; q[2] = (m02 - m20) / s; q[3] = (m10 - m01) / s;
  } else if (m00 > m11 && m00 > m22) {
    const double s = sqrt(1.0 + m00 - m11 - m22) * 2.0;
    q[0] = (m21 - m12) / s; q[1] = 0.25 * s; q[2] = (m01 + m10) / s; q[3] = (m02 + m20) / s;
  } else if (m11 > m22) {
    const double s = sqrt(1.0 + m11 - m00 - m22) * 2.0;
    q[0] = (m02 - m20) / s; q[1] = (m01 + m10) / s; q[2] = 0.25 * s; q[3] = (m12 + m21) / s;
  } else {
    const double s = sqrt(1.0 + m22 - m00 - m11) * 2.0;
    q[0] = (m10 - m01) / s; q[1] = (m02 + m20) / s; q[2] = (m12 + m21) / s; q[3] = 0.25 * s;
  }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n > 0.0)
    for (int k = 0; k < 4; ++k) q[k] /= n;
}

// Does the iteration that turned G_old into G_new end the loop?  (thread 0)
//   DefaultConvergenceCriteria: update D = G_new G_old^-1 with cos(angle) >= 1 - eps and |t|^2 <= eps;
//   relative / absolute change of the mean squared correspondence distance;
//   DifferentialTransformationChecker: mean over the last `smooth` iterations of the angular distance
//   between consecutive absolute rotations and of the distance between consecutive translations.
__device__ bool converged_extra(const IcpArgs& a, int pose, int it_done, const float* G_old, const float* G_new,
                                double E, double E_old) {
  bool stop = false;
  if (a.t_eps >= 0.f) {
    // D = G_new * inverse(G_old), rigid: R_D = R_n R_o^T, t_D = t_n - R_D t_o
    double R[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        R[r][c] = (double)G_new[r] * G_old[c] + (double)G_new[4 + r] * G_old[4 + c] + (double)G_new[8 + r] * G_old[8 + c];
    double tsq = 0.0;
    for (int r = 0; r < 3; ++r) {
      const double td = (double)G_new[12 + r] - (R[r][0] * G_old[12] + R[r][1] * G_old[13] + R[r][2] * G_old[14]);
      tsq += td * td;
    }
    const double cos_angle = 0.5 * (R[0][0] + R[1][1] + R[2][2] - 1.0);
    if (cos_angle >= 1.0 - (double)a.t_eps && tsq <= (double)a.t_eps) stop = true;
  }
  if (a.rel_mse > 0.f && E_old < (double)FLT_MAX && fabs(E - E_old) / E_old < (double)a.rel_mse) stop = true;
  if (a.abs_mse >= 0.f && E_old < (double)FLT_MAX && fabs(E - E_old) < (double)a.abs_mse) stop = true;
  if (a.smooth > 0 && a.st_hist) {
    double* H = a.st_hist + (size_t)pose * (kMaxSmooth + 1) * 7;
    const int L = a.smooth;
    // ring of the last L + 1 absolute transforms; slot of iteration k (1-based) is k % (L + 1)
    double q[4];
    quat_of(G_new, q);
    double* cur = H + (size_t)(it_done % (L + 1)) * 7;
    for (int k = 0; k < 4; ++k) cur[k] = q[k];
    for (int k = 0; k < 3; ++k) cur[4 + k] = G_new[12 + k];
    if (it_done > L) {   // rotations.size() > smoothLength
      double cr = 0.0, ct = 0.0;
      for (int k = 0; k < L; ++k) {
        const double* x = H + (size_t)((it_done - k) % (L + 1)) * 7;
        const double* y = H + (size_t)((it_done - k - 1) % (L + 1)) * 7;
        const double d = fabs(x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3]);
        cr += 2.0 * acos(d > 1.0 ? 1.0 : d);   // Quaternion::angularDistance
        ct += sqrt((x[4] - y[4]) * (x[4] - y[4]) + (x[5] - y[5]) * (x[5] - y[5]) + (x[6] - y[6]) * (x[6] - y[6]));
      }
      if (cr / L < (double)a.diff_rot && ct / L < (double)a.diff_trans) stop = true;
    }
  }
  return stop;
}

// SPLIT = false: persistent kernel, all iterations of one pose in one workgroup (many poses).
// SPLIT = true : one iteration's selection + update for one pose; the correspondences were
//                produced by icp_nn_split over many workgroups (few poses: a single pose would
//                otherwise run its exhaustive search on one CU of 256).  Same arithmetic, same
//                reduction tree: both paths give identical results.
template <bool SPLIT>
__global__ __launch_bounds__(kIcpThreads) void icp_refine(IcpArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  float4* s_tgt = reinterpret_cast<float4*>(smem);                         // kTgtTile float4
  double* s_red = reinterpret_cast<double*>(smem);                         // aliases the tile
  __shared__ float s_G[16];
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_scan[kIcpThreads / 64];
  __shared__ unsigned s_prefix, s_kleft, s_carry;
  __shared__ double s_energy, s_energy_old;
  __shared__ int s_continue;
  __shared__ double s_sum[kRedPlane + 1];
  __shared__ unsigned s_sel_bin, s_sel_acc;
  __shared__ float s_G_old[16];

  const int pose = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Tg = a.T + 16 * (size_t)pose;
  float* d2w = a.ws_d2 + (size_t)pose * a.n_src;
  int* jw = a.ws_j + (size_t)pose * a.n_src;
  if (SPLIT && a.st_done[pose]) return;  // whole block, uniform
  if (tid < 16) s_G[tid] = Tg[tid];
  if (tid == 0) {
    // PCL: energy starts at numeric_limits<float>::max()
    s_energy_old = SPLIT ? a.st_E[pose] : (double)FLT_MAX;
    s_energy = 0.0;
  }
  __syncthreads();

  int it = SPLIT ? a.st_it[pose] : 0;
  for (;;) {
    if (SPLIT) {
      // decode the keys left by icp_nn_split and re-arm them for the next iteration
      unsigned long long* kw = a.ws_key + (size_t)pose * a.n_src;
      for (int i = tid; i < a.n_src; i += kIcpThreads) {
        const unsigned long long k = kw[i];
        kw[i] = ~0ull;
        d2w[i] = k == ~0ull ? FLT_MAX : __uint_as_float((unsigned)(k >> 32));
        jw[i] = k == ~0ull ? -1 : (int)(unsigned)(k & 0xFFFFFFFFull);
      }
    } else {
    // ---- 1. correspondences: exhaustive NN of G*s_i in the target, tiled through LDS --------
    const float g00 = s_G[0], g10 = s_G[1], g20 = s_G[2], g01 = s_G[4], g11 = s_G[5], g21 = s_G[6],
                g02 = s_G[8], g12 = s_G[9], g22 = s_G[10], g03 = s_G[12], g13 = s_G[13], g23 = s_G[14];
    for (int base = 0; base < a.n_src; base += kIcpThreads * kIcpR) {
      float x[kIcpR], y[kIcpR], z[kIcpR], best[kIcpR];
      int bj[kIcpR];
#pragma unroll
      for (int r = 0; r < kIcpR; ++r) {
        int i = base + r * kIcpThreads + tid;
        float4 s = i < a.n_src ? a.src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        x[r] = row_xf(g00, g01, g02, g03, s.x, s.y, s.z);
        y[r] = row_xf(g10, g11, g12, g13, s.x, s.y, s.z);
        z[r] = row_xf(g20, g21, g22, g23, s.x, s.y, s.z);
        best[r] = FLT_MAX;
        bj[r] = -1;
      }
      for (int t0 = 0; t0 < a.n_tgt; t0 += kTgtTile) {
        const int tn = min(kTgtTile, a.n_tgt - t0);
        __syncthreads();  // previous tile fully consumed
        for (int j = tid; j < tn; j += kIcpThreads) s_tgt[j] = a.tgt[t0 + j];
        __syncthreads();
        for (int j = 0; j < tn; ++j) {
          const float4 m = s_tgt[j];  // broadcast read
#pragma unroll
          for (int r = 0; r < kIcpR; ++r) {
            float dx = __fsub_rn(x[r], m.x), dy = __fsub_rn(y[r], m.y), dz = __fsub_rn(z[r], m.z);
            float d2 = __fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dz, dz)));
            if (d2 < best[r]) {  // strict: the lowest j wins ties
              best[r] = d2;
              bj[r] = t0 + j;
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kIcpR; ++r) {
        int i = base + r * kIcpThreads + tid;
        if (i < a.n_src) {
          d2w[i] = best[r];
          jw[i] = bj[r];
        }
      }
    }
    }  // !SPLIT
    __syncthreads();  // d2w/jw visible to the block (same workgroup: global writes + barrier)

    // ---- 2. selection threshold: k-th smallest d2 by radix select on the float bits ---------
    unsigned thr_key = 0xFFFFFFFFu, ties_to_take = 0xFFFFFFFFu;  // default: take everything
    if (a.max_corr2 < 0.f && a.k_trim < a.n_src) {
      if (tid == 0) {
        s_prefix = 0;
        s_kleft = (unsigned)a.k_trim;  // rank (1-based) of the element we look for
      }
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < a.n_src; i += kIcpThreads) {
          unsigned key = __float_as_uint(d2w[i]);
          if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // the bin that holds the kleft-th element: inclusive scan of the 256 counts on 4 waves (the
        // serial walk of thread 0 cost an LDS round trip per bin, four times per iteration)
        unsigned hv = 0, incl = 0;
        if (tid < 256) {
          hv = s_hist[tid];
          incl = hv;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
          }
          if (lane == 63) s_scan[wave] = incl;
        }
        if (tid == 0) {
          s_sel_bin = 255u;
          s_sel_acc = 0xFFFFFFFFu;
        }
        __syncthreads();
        if (tid < 256) {
          unsigned woff = 0;
          for (int w = 0; w < wave; ++w) woff += s_scan[w];
          incl += woff;
          const unsigned excl = incl - hv, kleft = s_kleft;
          if (excl < kleft && kleft <= incl) {   // exactly one bin (counts are non-negative)
            s_sel_bin = (unsigned)tid;
            s_sel_acc = excl;
          }
        }
        __syncthreads();
        if (tid == 0) {
          unsigned acc = s_sel_acc;
          if (acc == 0xFFFFFFFFu) acc = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];   // rank beyond the population
          s_kleft = s_kleft - acc;
          s_prefix = prefix | (s_sel_bin << shift);
        }
        __syncthreads();
      }
      thr_key = s_prefix;        // the k-th smallest key
      ties_to_take = s_kleft;    // how many elements equal to it belong to the k smallest
    }

    // ---- 3. f64 sums over the selected pairs (ordered tie handling), fixed-tree reduction ----
    double acc[kRedPlane];
#pragma unroll
    for (int k = 0; k < kRedPlane; ++k) acc[k] = 0.0;
    double e_acc = 0.0;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < a.n_src; base += kIcpThreads) {
      const int i = base + tid;
      unsigned key = 0xFFFFFFFFu;
      float d2 = 0.f;
      if (i < a.n_src) {
        d2 = d2w[i];
        key = __float_as_uint(d2);
      }
      bool sel;
      if (a.max_corr2 >= 0.f) {
        sel = i < a.n_src && d2 <= a.max_corr2;
      } else if (thr_key == 0xFFFFFFFFu) {
        sel = i < a.n_src;
      } else {
        // ties at the threshold are taken in index order: ordered exclusive scan of the flags
        bool tie = i < a.n_src && key == thr_key;
        unsigned long long bm = __ballot(tie);
        unsigned before = __popcll(bm & ((1ull << lane) - 1ull));
        if (lane == 0) s_scan[wave] = __popcll(bm);
        __syncthreads();
        unsigned woff = s_carry;
        for (int w = 0; w < wave; ++w) woff += s_scan[w];
        sel = i < a.n_src && (key < thr_key || (tie && woff + before < ties_to_take));
        __syncthreads();
        if (tid == 0) {
          unsigned tot = 0;
          for (int w = 0; w < kIcpThreads / 64; ++w) tot += s_scan[w];
          s_carry += tot;
        }
        __syncthreads();
      }
      const int jm = i < a.n_src ? jw[i] : -1;
      if (sel && jm >= 0 && a.metric == 1) {
        const float4 s = a.src[i];
        const float4 m = a.tgt[jm];
        const float4 nn = a.tgt_n[jm];
        const double sx = row_xf(s_G[0], s_G[4], s_G[8], s_G[12], s.x, s.y, s.z);
        const double sy = row_xf(s_G[1], s_G[5], s_G[9], s_G[13], s.x, s.y, s.z);
        const double sz = row_xf(s_G[2], s_G[6], s_G[10], s_G[14], s.x, s.y, s.z);
        const double nx = nn.x, ny = nn.y, nz = nn.z;
        const double row[6] = {nz * sy - ny * sz, nx * sz - nz * sx, ny * sx - nx * sy, nx, ny, nz};
        const double rhs = nx * (double)m.x + ny * (double)m.y + nz * (double)m.z - nx * sx - ny * sy - nz * sz;
        acc[0] += 1.0;
        int t = 1;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = r; c < 6; ++c) acc[t++] += row[r] * row[c];
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[22 + r] += row[r] * rhs;
        e_acc += (double)d2;
      } else if (sel && jm >= 0) {  // jm < 0: a non-finite transformed point has no neighbour
        float4 s = a.src[i];
        float4 m = a.tgt[jm];
        acc[0] += 1.0;
        acc[1] += s.x; acc[2] += s.y; acc[3] += s.z;
        acc[4] += m.x; acc[5] += m.y; acc[6] += m.z;
        acc[7] += (double)s.x * m.x; acc[8] += (double)s.x * m.y; acc[9] += (double)s.x * m.z;
        acc[10] += (double)s.y * m.x; acc[11] += (double)s.y * m.y; acc[12] += (double)s.y * m.z;
        acc[13] += (double)s.z * m.x; acc[14] += (double)s.z * m.y; acc[15] += (double)s.z * m.z;
        e_acc += (double)d2;
      }
    }
    // wave butterfly, then the 16 wave results through LDS (aliases the target tile: all reads of
    // the tile finished before the barrier after step 1)
    // point-to-point uses the first 16 sums only (the branch is wave-uniform and folds away for k < 16)
#pragma unroll
    for (int k = 0; k < kRedPlane; ++k)
      if (k < 16 || a.metric == 1)
        for (int off = 32; off >= 1; off >>= 1) acc[k] += __shfl_xor(acc[k], off, 64);
    for (int off = 32; off >= 1; off >>= 1) e_acc += __shfl_xor(e_acc, off, 64);
    __syncthreads();
    if (lane == 0) {
      for (int k = 0; k < kRedPlane; ++k) s_red[wave * (kRedPlane + 1) + k] = acc[k];
      s_red[wave * (kRedPlane + 1) + kRedPlane] = e_acc;
    }
    __syncthreads();
    if (tid <= kRedPlane) {   // one thread per sum, waves added in order (as thread 0 did alone before)
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < kIcpThreads / 64; ++w) v += s_red[w * (kRedPlane + 1) + tid];
      s_sum[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
      const double* red = s_sum;
      // progress is judged on the mean squared distance of the selected pairs; with a fixed
      // trim count this is PCL's energy test (E/E_old), and it stays meaningful when a
      // correspondence cap lets |S| change between iterations
      const double E = red[0] >= 1.0 ? red[kRedPlane] / red[0] : 0.0;
      // ---- 4. closed-form update, then the progress tests (PCL order: update first) ----------
      for (int k = 0; k < 16; ++k) s_G_old[k] = s_G[k];
      if (a.metric == 1) solve_plane(red, s_G);
      else solve_rigid(red, s_G);
      const double E_old = s_energy_old;
      s_energy = E;
      s_energy_old = E;
      bool go = it + 1 < a.max_iter;
      if (a.ratio > 0.f && !(E / E_old < (double)a.ratio)) go = false;   // TrimmedICP's energy ratio
      if (red[0] < 1.0) go = false;                                        // no correspondences left
      if (converged_extra(a, pose, it + 1, s_G_old, s_G, E, E_old)) go = false;
      s_continue = go ? 1 : 0;
    }
    __syncthreads();
    ++it;
    if (SPLIT || !s_continue) break;
  }
  if (tid < 16) Tg[tid] = s_G[tid];
  if (tid == 0) {
    if (a.energy) a.energy[pose] = (float)s_energy;
    if (a.iters) a.iters[pose] = it;
    if (SPLIT) {
      a.st_E[pose] = s_energy;
      a.st_it[pose] = it;
      if (!s_continue) {
        a.st_done[pose] = 1;
        atomicAdd(a.n_done, 1);
      }
    }
  }
}

// Split-path correspondences: grid (source chunks, target chunks, poses); a block keeps 4 source
// points per lane in VGPRs, stages its 1024-point target chunk in LDS and publishes each source
// point's best (d2, j) with one 64-bit atomic min -- min over the key is min d2, then lowest j,
// i.e. exactly what the strict `<` scan of the persistent kernel returns.
constexpr int kNnThreads = 128;   // 512 source points per workgroup: 2500 sources -> 5 workgroups, 98 % full
constexpr int kNnTgt = 512;    // 8 KB of LDS per 2-wave workgroup: 8 waves per SIMD resident

__global__ __launch_bounds__(kNnThreads) void icp_nn_split(IcpArgs a) {
  __shared__ float4 s_t[kNnTgt];
  const int pose = blockIdx.z;
  if (a.st_done[pose]) return;
  const float* G = a.T + 16 * (size_t)pose;
  const float g00 = G[0], g10 = G[1], g20 = G[2], g01 = G[4], g11 = G[5], g21 = G[6], g02 = G[8], g12 = G[9],
              g22 = G[10], g03 = G[12], g13 = G[13], g23 = G[14];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * kNnThreads * kIcpR;
  const int t0 = blockIdx.y * kNnTgt;
  const int tn = min(kNnTgt, a.n_tgt - t0);
  // the tile is padded to a multiple of four with NaN points (a NaN distance never improves)
  const int tn4 = (tn + 3) & ~3;
  const float qnan = __int_as_float(0x7FC00000);
  for (int j = tid; j < tn4; j += kNnThreads) s_t[j] = j < tn ? a.tgt[t0 + j] : make_float4(qnan, qnan, qnan, 0.f);
  // Two source points per packed-fp32 register pair: the three differences, three squares and two
  // sums of a (source, target) pair are v_pk_* instructions shared by two pairs (4 VALU per pair
  // instead of 8; same operations, each rounded separately -- the file is built -ffp-contract=off).
  // The best-so-far update (two selects per pair) runs only when some lane of the wave improved:
  // after the first few hundred targets of a tile that is rare.
  typedef float v2f __attribute__((ext_vector_type(2)));
  static_assert(kIcpR == 4, "two register pairs per lane");
  v2f X[2], Y[2], Z[2];
  float best[kIcpR];
  int bj[kIcpR];
  const int* jprev = a.ws_j + (size_t)pose * a.n_src;  // correspondences of the previous iteration (-1: none)
#pragma unroll
  for (int r = 0; r < kIcpR; ++r) {
    int i = base + r * kNnThreads + tid;
    float4 s = i < a.n_src ? a.src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float sx = row_xf(g00, g01, g02, g03, s.x, s.y, s.z);
    const float sy = row_xf(g10, g11, g12, g13, s.x, s.y, s.z);
    const float sz = row_xf(g20, g21, g22, g23, s.x, s.y, s.z);
    X[r >> 1][r & 1] = sx;
    Y[r >> 1][r & 1] = sy;
    Z[r >> 1][r & 1] = sz;
    // Bound from the previous iteration: the distance to last iteration's correspondence is an upper
    // bound on this iteration's minimum, so the search starts just ABOVE it (next float up, so that
    // this very candidate and every tie with it are still found, in scan order) instead of at
    // FLT_MAX -- after that, improvements (the slow path below) are rare from the first target on,
    // and the workgroups of tiles that hold nothing closer write no key at all.
    best[r] = FLT_MAX;
    bj[r] = -1;
    const int jp = i < a.n_src ? jprev[i] : -1;
    if (jp >= 0) {
      const float4 m = a.tgt[jp];
      const float dx = __fsub_rn(sx, m.x), dy = __fsub_rn(sy, m.y), dz = __fsub_rn(sz, m.z);
      const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dz, dz)));
      if (d2 < FLT_MAX) best[r] = __uint_as_float(__float_as_uint(d2) + 1u);  // d2 >= 0: next float up
    }
  }
  __syncthreads();
  for (int j = 0; j < tn4; j += 4) {
    // four targets per trip: their LDS reads are issued together, ahead of the arithmetic
    float4 m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = s_t[j + k];
    v2f d2[4][2];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const v2f dx = X[p] - m[k].x, dy = Y[p] - m[k].y, dz = Z[p] - m[k].z;
        d2[k][p] = dx * dx + (dy * dy + dz * dz);
        any |= (d2[k][p].x < best[2 * p]) | (d2[k][p].y < best[2 * p + 1]);
      }
    if (__ballot(any)) {  // in target order, against the bound as it tightens
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const bool c0 = d2[k][p].x < best[2 * p], c1 = d2[k][p].y < best[2 * p + 1];
          best[2 * p] = c0 ? d2[k][p].x : best[2 * p];
          best[2 * p + 1] = c1 ? d2[k][p].y : best[2 * p + 1];
          bj[2 * p] = c0 ? t0 + j + k : bj[2 * p];
          bj[2 * p + 1] = c1 ? t0 + j + k : bj[2 * p + 1];
        }
    }
  }
  unsigned long long* kw = a.ws_key + (size_t)pose * a.n_src;
#pragma unroll
  for (int r = 0; r < kIcpR; ++r) {
    int i = base + r * kNnThreads + tid;
    if (i < a.n_src && bj[r] >= 0)
      atomicMin(&kw[i], ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned)bj[r]);
  }
}


// ---- uniform-grid nearest neighbour for capped searches on scene-sized clouds -------------------
// Cells of edge h >= max_corr over the target's bounding box, every target point in exactly one
// cell (counting sort by cell id); a query visits the 27 cells around its own.  Every point within
// max_corr of the query lies in them, so whenever the scan's nearest neighbour passes the cap the two
// searches return the same (d2, lowest j); when it does not, the pair is unselected either way.
__device__ __forceinline__ int grid_cell_of(const IcpArgs& a, float x, float y, float z, int* cx, int* cy, int* cz) {
  const float fx = (x - a.gox) * a.ginv_h, fy = (y - a.goy) * a.ginv_h, fz = (z - a.goz) * a.ginv_h;
  if (!(fx >= -1.f && fx < (float)(a.gnx + 1) && fy >= -1.f && fy < (float)(a.gny + 1) && fz >= -1.f && fz < (float)(a.gnz + 1)))
    return 0;   // farther than one cell from the grid (or NaN): nothing within max_corr
  *cx = (int)floorf(fx);
  *cy = (int)floorf(fy);
  *cz = (int)floorf(fz);
  return 1;
}

template <bool FILL>
__global__ __launch_bounds__(256) void grid_scatter(IcpArgs a, uint32_t* __restrict__ cell_ctr, float4* __restrict__ pts) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.n_tgt) return;
  const float4 p = a.tgt[j];
  int cx, cy, cz;
  if (!grid_cell_of(a, p.x, p.y, p.z, &cx, &cy, &cz)) return;
  if (cx < 0 || cy < 0 || cz < 0 || cx >= a.gnx || cy >= a.gny || cz >= a.gnz) return;
  const size_t c = ((size_t)cz * a.gny + cy) * a.gnx + cx;
  const uint32_t slot = atomicAdd(&cell_ctr[c], 1u);
  if (FILL) pts[a.gcell_start[c] + slot] = make_float4(p.x, p.y, p.z, __int_as_float(j));
}

__global__ __launch_bounds__(256) void icp_nn_grid(IcpArgs a) {
  const int pose = blockIdx.y;
  if (a.st_done[pose]) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_src) return;
  const float* G = a.T + 16 * (size_t)pose;
  const float4 s = a.src[i];
  const float x = row_xf(G[0], G[4], G[8], G[12], s.x, s.y, s.z);
  const float y = row_xf(G[1], G[5], G[9], G[13], s.x, s.y, s.z);
  const float z = row_xf(G[2], G[6], G[10], G[14], s.x, s.y, s.z);
  float best = FLT_MAX;
  int bj = -1;
  int cx, cy, cz;
  if (grid_cell_of(a, x, y, z, &cx, &cy, &cz)) {
    for (int dz = -1; dz <= 1; ++dz) {
      const int zz = cz + dz;
      if (zz < 0 || zz >= a.gnz) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        const int yy = cy + dy;
        if (yy < 0 || yy >= a.gny) continue;
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, a.gnx - 1);
        if (x0 > x1) continue;
        // the three x-neighbours are consecutive cells: one contiguous range of points
        const size_t c0 = ((size_t)zz * a.gny + yy) * a.gnx + x0;
        const uint32_t b = a.gcell_start[c0], e = a.gcell_start[c0 + (x1 - x0) + 1];
        for (uint32_t k = b; k < e; ++k) {
          const float4 m = a.gpts[k];
          const float dx = __fsub_rn(x, m.x), dyy = __fsub_rn(y, m.y), dzz = __fsub_rn(z, m.z);
          const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dyy, dyy), __fmul_rn(dzz, dzz)));
          const int j = __float_as_int(m.w);
          if (d2 < best || (d2 == best && j < bj)) {   // the scan's rule: smallest d2, then lowest j
            best = d2;
            bj = j;
          }
        }
      }
    }
  }
  if (bj >= 0)
    a.ws_key[(size_t)pose * a.n_src + i] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bj;
}

}  // namespace

int launch_icp(pgp_ctx* ctx, const float4* d_src, int n_src, const float4* d_tgt, const float4* d_tgt_n, int n_tgt,
               float* d_T, int n, const pgp_icp_options* prm, float* d_energy, int* d_iters, hipStream_t stream) {
  if (n <= 0) return PGP_OK;
  if (n_src <= 0 || n_tgt <= 0) {
    set_error("icp: empty source or target cloud");
    return PGP_EINVAL;
  }
  if (prm->error_metric == 1 && !d_tgt_n) {
    set_error("icp: the point-to-plane metric needs target normals");
    return PGP_EINVAL;
  }
  if (prm->error_metric != 0 && prm->error_metric != 1) {
    set_error("icp: unknown error metric %d", prm->error_metric);
    return PGP_EINVAL;
  }
  // the pose index rides on gridDim.z / .y (<= 65535): larger batches go in slices
  constexpr int kMaxPoses = 32768;
  if (n > kMaxPoses) {
    for (int off = 0; off < n; off += kMaxPoses) {
      const int m = n - off < kMaxPoses ? n - off : kMaxPoses;
      int rc = launch_icp(ctx, d_src, n_src, d_tgt, d_tgt_n, n_tgt, d_T + 16 * (size_t)off, m, prm,
                          d_energy ? d_energy + off : nullptr, d_iters ? d_iters + off : nullptr, stream);
      if (rc != PGP_OK) return rc;
    }
    return PGP_OK;
  }
  IcpArgs a{};
  a.src = d_src;
  a.tgt = d_tgt;
  a.tgt_n = d_tgt_n;
  a.n_src = n_src;
  a.n_tgt = n_tgt;
  a.T = d_T;
  a.n = n;
  a.max_iter = prm->max_iterations > 0 ? prm->max_iterations : 100;
  float tf = prm->trim_fraction;
  if (!(tf > 0.f) || tf > 1.f) tf = 1.f;
  // float numPoints = trim * size; align(..., abs(numPoints), ...) -> int (UCTState.cpp:176,194)
  int k = (int)fabsf(tf * (float)n_src);
  if (k < 1) k = 1;
  if (k > n_src) k = n_src;
  a.k_trim = k;
  a.max_corr2 = prm->max_corr_dist > 0.f ? prm->max_corr_dist * prm->max_corr_dist : -1.f;
  a.ratio = prm->energy_ratio;            // <= 0: the energy-ratio test is off
  a.metric = prm->error_metric;
  a.t_eps = prm->transformation_epsilon;  // < 0: off
  a.rel_mse = prm->relative_mse;
  a.abs_mse = prm->absolute_mse;
  a.diff_rot = prm->min_diff_rot;
  a.diff_trans = prm->min_diff_trans;
  a.smooth = (prm->min_diff_rot > 0.f && prm->min_diff_trans > 0.f)
                 ? (prm->smooth_length < 1 ? 1 : (prm->smooth_length > kMaxSmooth ? kMaxSmooth : prm->smooth_length))
                 : 0;
  int rc;
  size_t need = (size_t)n * n_src;
  // measured (tools/icp_time.py, 2500 x 5000, 10 iterations): the split path wins at every batch
  // size tried -- 1 pose 1.2 vs 7.7 ms, 64 poses 3.6 vs 12.8 ms, 256 poses 9.4 vs 12.9 ms -- so it
  // is the default; PGP_ICP_SPLIT=0 selects the single-launch persistent kernel (fully
  // asynchronous, graph-capturable)
  bool split = true;
  if (const char* v = getenv("PGP_ICP_SPLIT")) split = atoi(v) != 0;
  // grid search: only with a correspondence cap; by default when the scan would be >= 2^27 tests per pose
  bool use_grid = false;
  if (a.max_corr2 >= 0.f) {
    if (prm->nn_search == 2) use_grid = true;
    else if (prm->nn_search == 0) use_grid = (double)n_src * (double)n_tgt >= 134217728.0;
  } else if (prm->nn_search == 2) {
    set_error("icp: the grid search needs max_corr_dist > 0");
    return PGP_EINVAL;
  }
  if (use_grid || a.smooth > 0) split = true;   // both live on the host-driven path
  const size_t hist_bytes = a.smooth > 0 ? (size_t)n * (kMaxSmooth + 1) * 7 * 8 : 0;
  const size_t state_bytes = split ? need * 8 + (size_t)n * 16 + 64 + hist_bytes + 64 : 0;
  if ((rc = ctx->d_icp_ws.ensure(need * 8 + state_bytes + 64)) != PGP_OK) return rc;
  a.ws_d2 = ctx->d_icp_ws.as<float>();
  a.ws_j = reinterpret_cast<int*>(a.ws_d2 + need);
  a.energy = d_energy;
  a.iters = d_iters;
  const size_t lds = (size_t)kTgtTile * sizeof(float4);
  if (!ctx->icp_attr_set) {  // per context = per device (function attributes are per device)
    PGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(icp_refine<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(icp_refine<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->icp_attr_set = true;
  }
  if (!split) {
    hipLaunchKernelGGL(icp_refine<false>, dim3(n), dim3(kIcpThreads), lds, stream, a);
    PGP_HIP(hipGetLastError());
    return PGP_OK;
  }
  // ---- split path: per iteration, correspondences over many workgroups + one update workgroup
  unsigned char* p = reinterpret_cast<unsigned char*>(a.ws_j + need);
  p = reinterpret_cast<unsigned char*>(((uintptr_t)p + 15) & ~(uintptr_t)15);
  a.ws_key = reinterpret_cast<unsigned long long*>(p);
  a.st_E = reinterpret_cast<double*>(a.ws_key + need);
  a.st_it = reinterpret_cast<int*>(a.st_E + n);
  a.st_done = a.st_it + n;
  a.n_done = a.st_done + n;
  a.st_hist = a.smooth > 0 ? reinterpret_cast<double*>(((uintptr_t)(a.n_done + 1) + 15) & ~(uintptr_t)15) : nullptr;
  PGP_HIP(hipMemsetAsync(a.ws_key, 0xFF, need * 8, stream));
  PGP_HIP(hipMemsetAsync(a.ws_j, 0xFF, need * 4, stream));  // no previous correspondence yet
  PGP_HIP(hipMemsetAsync(a.st_it, 0, (size_t)n * 8 + 4, stream));
  if (a.st_hist) PGP_HIP(hipMemsetAsync(a.st_hist, 0, hist_bytes, stream));
  {
    std::vector<double> e0((size_t)n, (double)FLT_MAX);
    PGP_HIP(hipMemcpyAsync(a.st_E, e0.data(), (size_t)n * 8, hipMemcpyHostToDevice, stream));
    PGP_HIP(hipStreamSynchronize(stream));  // e0 is a stack temporary
  }
  if (use_grid) {
    // ---- the target's grid: bounding box (device), cell edge >= max_corr (grown to keep <= 2^26 cells)
    float bb[6];
    if ((rc = device_bbox(ctx, reinterpret_cast<const float*>(d_tgt), n_tgt, 4, bb, bb + 3, stream)) != PGP_OK) return rc;
    if (!(bb[0] <= bb[3])) bb[0] = bb[1] = bb[2] = bb[3] = bb[4] = bb[5] = 0.f;   // no finite target point
    float maxabs = 0.f;
    for (int q = 0; q < 6; ++q) maxabs = fmaxf(maxabs, fabsf(bb[q]));
    // margin for the rounding of the cell coordinate: a point within max_corr of a query is at most
    // one cell away from the query's cell
    float h = prm->max_corr_dist * 1.001f + 64.f * FLT_EPSILON * maxabs;
    for (;;) {
      // one spare cell per axis: the float cell coordinate of a point on the upper face may round up
      const double nx = floor((double)(bb[3] - bb[0]) / h) + 2, ny = floor((double)(bb[4] - bb[1]) / h) + 2,
                   nz = floor((double)(bb[5] - bb[2]) / h) + 2;
      if (nx * ny * nz <= 67108864.0) {
        a.gnx = (int)nx;
        a.gny = (int)ny;
        a.gnz = (int)nz;
        break;
      }
      h *= 1.26f;
    }
    a.gox = bb[0];
    a.goy = bb[1];
    a.goz = bb[2];
    a.ginv_h = 1.0f / h;
    const size_t cells = (size_t)a.gnx * a.gny * a.gnz;
    const size_t off_pts = ((cells + 1) * 8 + 64 + 255) & ~(size_t)255;
    if ((rc = ctx->d_icp_grid.ensure(off_pts + (size_t)n_tgt * 16 + 64)) != PGP_OK) return rc;
    if ((rc = ctx->d_scan_tmp.ensure(((cells + 1) / 2048 + 2) * 4)) != PGP_OK) return rc;
    unsigned char* gb = ctx->d_icp_grid.as<unsigned char>();
    uint32_t* ctr = reinterpret_cast<uint32_t*>(gb + 64);
    uint32_t* start = ctr + (cells + 1);
    float4* pts = reinterpret_cast<float4*>(gb + off_pts);
    a.gcell_start = start;
    a.gpts = pts;
    const dim3 gt((n_tgt + 255) / 256);
    PGP_HIP(hipMemsetAsync(ctr, 0, (cells + 1) * 4, stream));
    hipLaunchKernelGGL(grid_scatter<false>, gt, dim3(256), 0, stream, a, ctr, (float4*)nullptr);
    if ((rc = device_exclusive_scan(ctr, start, cells + 1, ctx->d_scan_tmp.as<uint32_t>(), stream)) != PGP_OK) return rc;
    PGP_HIP(hipMemsetAsync(ctr, 0, (cells + 1) * 4, stream));
    hipLaunchKernelGGL(grid_scatter<true>, gt, dim3(256), 0, stream, a, ctr, pts);
    PGP_HIP(hipGetLastError());
  }
  const dim3 gnn((n_src + kNnThreads * kIcpR - 1) / (kNnThreads * kIcpR), (n_tgt + kNnTgt - 1) / kNnTgt, n);
  const dim3 ggrid((n_src + 255) / 256, n);
  for (int it = 0; it < a.max_iter; ++it) {
    if (use_grid) hipLaunchKernelGGL(icp_nn_grid, ggrid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(icp_nn_split, gnn, dim3(kNnThreads), 0, stream, a);
    hipLaunchKernelGGL(icp_refine<true>, dim3(n), dim3(kIcpThreads), lds, stream, a);
    if (it == 0) PGP_HIP(hipGetLastError());   // a bad launch configuration shows on the first pair
    if ((it & 3) == 3) {  // every 4 iterations: has every pose stopped?
      int done = 0;
      PGP_HIP(hipMemcpyAsync(&done, a.n_done, 4, hipMemcpyDeviceToHost, stream));
      PGP_HIP(hipStreamSynchronize(stream));
      if (done >= n) break;
    }
  }
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

}  // namespace pgp
