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
#include <cstring>
#include <mutex>
#include <vector>

namespace pgp {

namespace {

#ifndef PGP_ICP_THREADS
#define PGP_ICP_THREADS 1024   // A/B knob (round 5): 512 = 256 registers per lane, twice the points per thread (results' low bits differ:
#endif                         // the sums' tree follows the thread count) -- profiles/r05_ab/icp_512_threads.log
constexpr int kIcpThreads = PGP_ICP_THREADS;
constexpr int kIcpBase = 1024;           // the points-per-thread classes below are in units of this many points
constexpr int kPS = kIcpBase / kIcpThreads;
static_assert(kIcpThreads == 1024 || kIcpThreads == 512, "workgroup shapes of the ICP kernels");
constexpr int kIcpR = 4;                 // source points per lane per sweep
constexpr int kTgtTile = 4096;           // target points per LDS tile (64 KB)
constexpr int kRedPlane = 28;            // point-to-plane: count + 21 (upper triangle of AtA) + 6 (Atb)
constexpr int kMaxSmooth = 8;            // history length of the differential checker
constexpr int kSumR = 4;                 // consecutive source points a thread sums per block of kSumR x kIcpThreads

// ---- exact nearest-neighbour index over the STATIC target cloud ---------------------------------
// The target (the object model: UCTState.cpp:137-139, utilities.cpp:666-676 build a TrimmedICP over it
// once and query it for every iteration) does not move, so one index serves all poses and iterations:
//   * a uniform grid of cells numbered row-major (x fastest), the target points sorted by cell (.w
//     keeps the original index), 16-bit cell starts: the cells [x0, x1] of one (y, z) row are ONE
//     contiguous run of points;
//   * per cell a REPRESENTATIVE: the target point nearest to the cell's centre.  A query's distance to
//     the representative of its own (clamped) cell is an upper bound U on its nearest-neighbour
//     distance -- a true candidate, so the search only has to visit the rows that meet the ball of
//     radius U, and the answer (smallest d2, then lowest original index: the exhaustive scan's rule) is
//     exact however loose U is.  The previous iteration's correspondence tightens U further.
//     Cost grows with U (a query visits (2U/h + 1)^2 rows), i.e. the index is at its best once the
//     poses are within a centimetre or two; a lane's loop nest is rows x points and nothing deeper,
//     because a wave pays the PRODUCT of the per-level maximum trip counts of its lanes (a four-level
//     nest over occupancy blocks measured 16 000 VALU instructions per wave-query, this one ~600).
// The whole index is one byte image (<= ~150 KB at 5000 points) that a workgroup copies into LDS.
struct NnGeom {
  float ox, oy, oz;
  float hx, inv_hx;      // cell edge along x (the direction of a row): short, the chord of a search ball is cut to it
  float h, inv_h;        // cell edge along y and z: long -- a search pays per ROW it visits (~45 instructions and
                         // an LDS round trip each, mostly to find the row empty), and there are (2U/h)^2 of them
  int nx, ny, nz;        // cells per axis
  int n_cells;
  int strip_shift;       // cell >> strip_shift = its strip (<= 512 strips): the secondary key of the query sort
  uint32_t off_start, off_rep, bytes;   // byte offsets inside the image (points at 0)
};

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
  // exact index of the static target (icp_nn_index / icp_persist_index)
  NnGeom nn;
  const unsigned char* nn_image; // the LDS image of the index, in HBM: points | start16 | rep16 | mask
  int* ws_pos;                   // [n][n_src] position (in the image's point order) of the last correspondence
  int nn_image_in_lds;           // 1: a workgroup copies the image into LDS; 0: it is read where it lies (L2)
  // several workgroups per pose (icp_persist_index, few poses): each searches its share of the source points,
  // the shares meet in HBM once per iteration
  int wgs_per_pose;              // 1, 2 or 4
  unsigned long long* x_buf;     // [2][n][n_src] (d2 bits << 32) | position, ping-pong by iteration parity
  unsigned* x_ctr;               // [n] arrivals of the pose's workgroups (monotone; zeroed before the launch)
  unsigned* x_ticks;             // [2][n][4] search time of every share, published with it
  unsigned solo_ticks;           // every share searched faster than this: the pose goes on in ONE workgroup
  int slot_budget;               // A/B knob (PGP_ICP_SLOTS): lane slots up to which queries get more lanes; 0 = one pass
  int rows_mode;                 // A/B knob (PGP_ICP_ROWS): phase B's lanes dealt by rows -- 0: where it pays (default), 1: always, 2: never
  // vicinity graph of the target (nnidx_vic_*): per image position the kVicK nearest other target points and a
  // radius inside which no unlisted point lies -- resolves a query next to a known candidate without a search
  const uint4* nn_vic;           // [n_tgt] or nullptr
  int first_walk;                // moves downhill on the graph in the FIRST iteration (no previous correspondence yet)
  // lost passes (the helping launch, PGP_ICP_HELP): a workgroup sets *x_lost; the follow-up launch (run_if = x_lost, one
  // workgroup per pose, starting again from the transforms saved in T_save) repairs the call.  (The clustered launch
  // repairs a lost meeting inside itself: workgroup 0 of the pose goes on alone.)
  unsigned* x_lost;              // [1] in the library's own workspace
  unsigned* x_done;              // [n] workgroups of the pose that have left: the last one zeroes the pose's counters for the next call
  unsigned* x_abandon;           // [n] set by the first workgroup whose wait for a meeting ran out: the partners leave at once
  unsigned wait_ticks;           // floor of every wait's clock bound (100 MHz ticks; launch_icp: 3 ms, PGP_ICP_WAIT_MS)
  int force_lost;                // test knob (PGP_ICP_FORCE_LOST): every pose's first meeting is declared lost
  float* T_save;                 // [n][16]: clustered launch: part 0 stores the pose's initial transform here
  const float* T_in;             // where a pose's initial transform is read (T itself, or T_save in the repair launch)
  const unsigned* run_if;        // non-null: the whole launch returns at once unless *run_if != 0
  // helping (icp_persist_help: one workgroup per pose, all resident): a workgroup whose pose has finished takes
  // search passes of poses that are still running -- see HelpPub
  unsigned char* help;           // the poses' publication blocks, help_stride bytes each
  size_t help_stride;
  unsigned long long* help_ctl;  // [n] (iteration tag << 32) | next unclaimed slot; tag 0 = nothing to take
  unsigned* help_nslots;         // [n] slots of the published iteration
  unsigned* help_done;           // [n] passes completed
  unsigned* help_finished;       // [1] poses that are through all their iterations
  int dbg_pose;                  // diagnostic builds (PGP_ICP_STAMPS): the pose whose phases are timed (PGP_ICP_DBG_POSE)
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

// The same eigenvector without iterating over rotations: Horn's N is symmetric and traceless, so its
// characteristic polynomial is l^4 + c2 l^2 + c1 l + c0 with c2 = -tr(N^2)/2, c1 = -tr(N^3)/3, c0 = det N
// (Newton's identities); it is convex and increasing beyond its largest root, so Newton's iteration from
// the Gershgorin bound descends monotonically onto that root (Horn 1987 section 4; Theobald 2005), and
// every non-zero column of adj(N - l I) is the eigenvector.  ~400 f64 operations against ~50 Jacobi
// rotations of four divisions / square roots each: 27 us -> 2 us of ONE lane per ICP iteration
// (tools/icp_phases.py).  Returns false -- the caller falls back to the Jacobi sweeps -- when the largest
// eigenvalue is (nearly) double: the adjugate then vanishes and the rotation is not unique.
__device__ __forceinline__ double det3(double a, double b, double c, double d, double e, double f, double g, double h,
                                       double i) {
  return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}
__device__ bool largest_eigvec4_direct(const double (&N)[4][4], double q[4]) {
  const double a00 = N[0][0], a01 = N[0][1], a02 = N[0][2], a03 = N[0][3], a11 = N[1][1], a12 = N[1][2], a13 = N[1][3],
               a22 = N[2][2], a23 = N[2][3], a33 = N[3][3];
  // N^2 (symmetric)
  const double m00 = a00 * a00 + a01 * a01 + a02 * a02 + a03 * a03, m01 = a00 * a01 + a01 * a11 + a02 * a12 + a03 * a13,
               m02 = a00 * a02 + a01 * a12 + a02 * a22 + a03 * a23, m03 = a00 * a03 + a01 * a13 + a02 * a23 + a03 * a33,
               m11 = a01 * a01 + a11 * a11 + a12 * a12 + a13 * a13, m12 = a01 * a02 + a11 * a12 + a12 * a22 + a13 * a23,
               m13 = a01 * a03 + a11 * a13 + a12 * a23 + a13 * a33, m22 = a02 * a02 + a12 * a12 + a22 * a22 + a23 * a23,
               m23 = a02 * a03 + a12 * a13 + a22 * a23 + a23 * a33, m33 = a03 * a03 + a13 * a13 + a23 * a23 + a33 * a33;
  const double p2 = m00 + m11 + m22 + m33;
  const double p3 = (m00 * a00 + m11 * a11 + m22 * a22 + m33 * a33) +
                    2.0 * (m01 * a01 + m02 * a02 + m03 * a03 + m12 * a12 + m13 * a13 + m23 * a23);
  const double det = a00 * det3(a11, a12, a13, a12, a22, a23, a13, a23, a33) - a01 * det3(a01, a12, a13, a02, a22, a23, a03, a23, a33) +
                     a02 * det3(a01, a11, a13, a02, a12, a23, a03, a13, a33) - a03 * det3(a01, a11, a12, a02, a12, a22, a03, a13, a23);
  const double c2 = -0.5 * p2, c1 = -p3 / 3.0, c0 = det;
  double lam = fmax(fmax(fabs(a00) + fabs(a01) + fabs(a02) + fabs(a03), fabs(a01) + fabs(a11) + fabs(a12) + fabs(a13)),
                    fmax(fabs(a02) + fabs(a12) + fabs(a22) + fabs(a23), fabs(a03) + fabs(a13) + fabs(a23) + fabs(a33)));
  if (!(lam > 0.0)) return false;
  for (int it = 0; it < 64; ++it) {
    const double l2 = lam * lam;
    const double P = (l2 + c2) * l2 + c1 * lam + c0, dP = (4.0 * l2 + 2.0 * c2) * lam + c1;
    if (!(dP > 0.0)) break;
    const double nl = lam - P / dP;
    if (!(nl < lam)) break;   // monotone from above: no further descent = converged to the rounding of P
    lam = nl;
  }
  const double b00 = a00 - lam, b11 = a11 - lam, b22 = a22 - lam, b33 = a33 - lam;
  // adj(B), B = N - lam I symmetric: cofactor(i, j) = (-1)^(i+j) det(B without row i and column j)
  const double k00 = det3(b11, a12, a13, a12, b22, a23, a13, a23, b33), k11 = det3(b00, a02, a03, a02, b22, a23, a03, a23, b33),
               k22 = det3(b00, a01, a03, a01, b11, a13, a03, a13, b33), k33 = det3(b00, a01, a02, a01, b11, a12, a02, a12, b22);
  const double k01 = -det3(a01, a12, a13, a02, b22, a23, a03, a23, b33), k02 = det3(a01, b11, a13, a02, a12, a23, a03, a13, b33),
               k03 = -det3(a01, b11, a12, a02, a12, b22, a03, a13, a23), k12 = -det3(b00, a01, a03, a02, a12, a23, a03, a13, b33),
               k13 = det3(b00, a01, a02, a02, a12, b22, a03, a13, a23), k23 = -det3(b00, a01, a02, a01, b11, a12, a03, a13, a23);
  // adj(B) = kappa v v^T: the column with the largest diagonal entry is the best conditioned one
  const double d0 = fabs(k00), d1 = fabs(k11), d2 = fabs(k22), d3 = fabs(k33);
  const double dm = fmax(fmax(d0, d1), fmax(d2, d3));
  if (!(dm > 1e-6 * lam * lam * lam)) return false;
  if (d0 == dm) { q[0] = k00; q[1] = k01; q[2] = k02; q[3] = k03; }
  else if (d1 == dm) { q[0] = k01; q[1] = k11; q[2] = k12; q[3] = k13; }
  else if (d2 == dm) { q[0] = k02; q[1] = k12; q[2] = k22; q[3] = k23; }
  else { q[0] = k03; q[1] = k13; q[2] = k23; q[3] = k33; }
  return true;
}

// Horn's closed form from the f64 sums over the selected pairs:
// red = {n, sx,sy,sz, mx,my,mz, Sxx,Sxy,Sxz, Syx,Syy,Syz, Szx,Szy,Szz}  (S_ab = sum s_a m_b)
__device__ __attribute__((noinline)) void solve_rigid(const double* red, float* G) {
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
  if (!largest_eigvec4_direct(N, q)) largest_eigvec4(N, q);
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
__device__ __attribute__((noinline)) void solve_plane(const double* red, float* G) {
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
// what the extra stop rules read of IcpArgs, handed over BY VALUE: a reference to the kernel's argument struct made every
// kernel that calls this keep a 440-byte copy of it in scratch memory (written by every thread at the start of the launch,
// read back by lane 0 through memory round trips: 17.6 of the 88 us of icp_refine<true> on the reference's table alignment)
struct StopRules {
  float t_eps, rel_mse, abs_mse, diff_rot, diff_trans;
  int smooth;
  double* st_hist;
};
__device__ __forceinline__ StopRules stop_rules_of(const IcpArgs& a) {
  return StopRules{a.t_eps, a.rel_mse, a.abs_mse, a.diff_rot, a.diff_trans, a.smooth, a.st_hist};
}
__device__ __attribute__((noinline)) bool converged_extra(const StopRules a, int pose, int it_done, const float* G_old, const float* G_new,
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

// Sum of one double per lane over the wave, the same value in every lane, on the DPP path: four steps inside
// each row of 16 lanes (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: two register
// moves + one v_add_f64 each, no LDS), then the four row sums are read with v_readlane and added in row order.
// A FIXED tree: ((r0 + r1) + r2) + r3 of balanced 16-lane trees -- every kernel of this file reduces with it,
// so they agree bit for bit.  (__shfl_xor on doubles = 12 ds_bpermute round trips per value; the 17 sums of an
// iteration spent ~5 us in them.)
template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int plo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  const int phi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return v + __hiloint2double(phi, plo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v = dpp_add_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add_f64<0x141>(v);   // row_half_mirror
  v = dpp_add_f64<0x140>(v);   // row_mirror: every lane of a row holds the row's sum
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return ((r0 + r1) + r2) + r3;
}

// Inclusive prefix sum over the 64 lanes on the DPP path: four row_shr steps inside the rows of 16, then the row
// totals are carried by row_bcast15 (rows 1, 3) and row_bcast31 (rows 2, 3) -- six additions, no LDS.  (__shfl_up
// is a ds_bpermute round trip per step; the scans of the query sort and of the selection sit between barriers
// where nothing hides it.)
__device__ __forceinline__ unsigned wave_scan_incl_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast15 into rows 1 and 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast31 into rows 2 and 3
  return v;
}

constexpr int kPartStride = 32;   // doubles per (pose, block) in the partial sums of icp_sums_partial (kRedPlane + 1 = 29 used)

// SPLIT = false: persistent kernel, all iterations of one pose in one workgroup (many poses).
// SPLIT = true : one iteration's selection + update for one pose; the correspondences were
//                produced by icp_nn_split over many workgroups (few poses: a single pose would
//                otherwise run its exhaustive search on one CU of 256).  Same arithmetic, same
//                reduction tree: both paths give identical results.
// PART (host-driven path, scenes beyond 4096 points, no trimming): the sums of the iteration were formed by icp_sums_partial,
// one workgroup per block of 4096 points, and are only added up here in block order (a single workgroup walking 30 000
// points -- the reference's table alignment -- spent 52 of its 71 us in that walk, profiles/r04_ab/icp_refine_ablation.log)
template <bool SPLIT, bool PART = false>
__global__ __launch_bounds__(kIcpThreads) void icp_refine(IcpArgs a, const double* __restrict__ part = nullptr, int n_blk = 0) {
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
    if (SPLIT && PART) {
      // (icp_sums_partial has decoded and re-armed the keys of its blocks)
    } else if (SPLIT) {
      // decode the keys left by icp_nn_split and re-arm them for the next iteration
      unsigned long long* kw = a.ws_key + (size_t)pose * a.n_src;
      // four keys per trip, their loads issued together (a 30 000-point scene is 30 trips of a load-then-store chain
      // otherwise: the reference's table alignment spent a third of this kernel here)
      for (int i0 = tid; i0 < a.n_src; i0 += 4 * kIcpThreads) {
        unsigned long long k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = i0 + u * kIcpThreads < a.n_src ? kw[i0 + u * kIcpThreads] : ~0ull;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * kIcpThreads;
          if (i >= a.n_src) break;
          kw[i] = ~0ull;
          d2w[i] = k[u] == ~0ull ? FLT_MAX : __uint_as_float((unsigned)(k[u] >> 32));
          jw[i] = k[u] == ~0ull ? -1 : (int)(unsigned)(k[u] & 0xFFFFFFFFull);
        }
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
    // Thread t sums the points 4 t .. 4 t + 3 of every block of 4 x 1024 points, in index order (the persistent
    // indexed kernel maps its points the same way, so both add the same numbers in the same order): a cloud of
    // 1756 points then sits on 7 of the 16 waves, and only those pay the 17 wave-level f64 trees.
#if defined(PGP_REFINE_ABLATE) && PGP_REFINE_ABLATE >= 4   // timing experiments (wrong results): 4 no sums loop
    for (int b0 = 0; b0 < 0; b0 += kSumR * kIcpThreads) {
#else
    for (int b0 = 0; b0 < (PART ? 0 : a.n_src); b0 += kSumR * kIcpThreads) {
#endif
      unsigned key[kSumR];
      float d2v[kSumR];
      bool sel[kSumR];
#pragma unroll
      for (int r = 0; r < kSumR; ++r) {
        const int i = b0 + kSumR * tid + r;
        key[r] = 0xFFFFFFFFu;
        d2v[r] = 0.f;
        if (i < a.n_src) {
          d2v[r] = d2w[i];
          key[r] = __float_as_uint(d2v[r]);
        }
      }
      if (a.max_corr2 >= 0.f) {
#pragma unroll
        for (int r = 0; r < kSumR; ++r) sel[r] = b0 + kSumR * tid + r < a.n_src && d2v[r] <= a.max_corr2;
      } else if (thr_key == 0xFFFFFFFFu) {
#pragma unroll
        for (int r = 0; r < kSumR; ++r) sel[r] = b0 + kSumR * tid + r < a.n_src;
      } else {
        // ties at the threshold are taken in index order: ordered exclusive scan of the per-thread tie counts
        unsigned mine = 0;
#pragma unroll
        for (int r = 0; r < kSumR; ++r) mine += (b0 + kSumR * tid + r < a.n_src && key[r] == thr_key) ? 1u : 0u;
        const unsigned incl = wave_scan_incl_u32(mine);
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        unsigned rank = s_carry + incl - mine;
        for (int w = 0; w < wave; ++w) rank += s_scan[w];
#pragma unroll
        for (int r = 0; r < kSumR; ++r) {
          const bool in = b0 + kSumR * tid + r < a.n_src;
          const bool tie = in && key[r] == thr_key;
          sel[r] = in && (key[r] < thr_key || (tie && rank < ties_to_take));
          rank += tie ? 1u : 0u;
        }
        __syncthreads();
        if (tid == 0) {
          unsigned tot = 0;
          for (int w = 0; w < kIcpThreads / 64; ++w) tot += s_scan[w];
          s_carry += tot;
        }
        __syncthreads();
      }
#pragma unroll
      for (int r = 0; r < kSumR; ++r) {
      const int i = b0 + kSumR * tid + r;
      const float d2 = d2v[r];
      const int jm = i < a.n_src ? jw[i] : -1;
      if (sel[r] && jm >= 0 && a.metric == 1) {
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
        for (int rr = 0; rr < 6; ++rr)
#pragma unroll
          for (int c = rr; c < 6; ++c) acc[t++] += row[rr] * row[c];
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) acc[22 + rr] += row[rr] * rhs;
        e_acc += (double)d2;
      } else if (sel[r] && jm >= 0) {  // jm < 0: a non-finite transformed point has no neighbour
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
    }
    // wave butterfly, then the 16 wave results through LDS (aliases the target tile: all reads of
    // the tile finished before the barrier after step 1)
    // point-to-point uses the first 16 sums only (the branch is wave-uniform and folds away for k < 16)
#if !(defined(PGP_REFINE_ABLATE) && PGP_REFINE_ABLATE >= 3)   // 3: no wave sums
#pragma unroll
    for (int k = 0; k < kRedPlane; ++k)
      if (k < 16 || a.metric == 1)
        acc[k] = wave_sum_f64(acc[k]);
    e_acc = wave_sum_f64(e_acc);
#endif
    __syncthreads();
    if (lane == 0) {
      for (int k = 0; k < kRedPlane; ++k) s_red[wave * (kRedPlane + 1) + k] = acc[k];
      s_red[wave * (kRedPlane + 1) + kRedPlane] = e_acc;
    }
    __syncthreads();
    if (PART) {
      if (tid <= kRedPlane) {   // one thread per sum, the blocks' sums added in block order
        double v = 0.0;
        for (int b = 0; b < n_blk; ++b) v += part[((size_t)pose * n_blk + b) * kPartStride + tid];
        s_sum[tid] = v;
      }
    } else if (tid <= kRedPlane) {   // one thread per sum, waves added in order (as thread 0 did alone before)
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
#if !(defined(PGP_REFINE_ABLATE) && PGP_REFINE_ABLATE >= 2)   // 2: no solve
      if (a.metric == 1) solve_plane(red, s_G);
      else solve_rigid(red, s_G);
#endif
      const double E_old = s_energy_old;
      s_energy = E;
      s_energy_old = E;
      bool go = it + 1 < a.max_iter;
      if (a.ratio > 0.f && !(E / E_old < (double)a.ratio)) go = false;   // TrimmedICP's energy ratio
#if !(defined(PGP_REFINE_ABLATE) && PGP_REFINE_ABLATE >= 1)   // 1: no stop rules (runs to max_iter)
      if (red[0] < 1.0) go = false;                                        // no correspondences left
      if (converged_extra(stop_rules_of(a), pose, it + 1, s_G_old, s_G, E, E_old)) go = false;
#endif
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

// One block of kSumR * kIcpThreads = 4096 points of one pose: decodes (and re-arms) the block's keys, selects (cap or
// everything: this path is not taken when a trimming threshold has to be found first) and forms the block's f64 sums with the
// accumulation and the reduction tree of icp_refine -- for a cloud of one block the numbers are icp_refine's own.
__global__ __launch_bounds__(kIcpThreads) void icp_sums_partial(IcpArgs a, double* __restrict__ part, int n_blk) {
  __shared__ double s_red[(kIcpThreads / 64) * (kRedPlane + 1)];
  __shared__ float s_G[16];
  const int pose = blockIdx.y, blk = blockIdx.x;
  if (a.st_done[pose]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 16) s_G[tid] = a.T[16 * (size_t)pose + tid];
  __syncthreads();
  unsigned long long* kw = a.ws_key + (size_t)pose * a.n_src;
  float* d2w = a.ws_d2 + (size_t)pose * a.n_src;
  int* jw = a.ws_j + (size_t)pose * a.n_src;
  const int b0 = blk * kSumR * kIcpThreads;
  unsigned long long kv[kSumR];
#pragma unroll
  for (int r = 0; r < kSumR; ++r) {
    const int i = b0 + kSumR * tid + r;
    kv[r] = i < a.n_src ? kw[i] : ~0ull;
  }
  float d2v[kSumR];
  int jmv[kSumR];
  bool sel[kSumR];
#pragma unroll
  for (int r = 0; r < kSumR; ++r) {
    const int i = b0 + kSumR * tid + r;
    d2v[r] = kv[r] == ~0ull ? FLT_MAX : __uint_as_float((unsigned)(kv[r] >> 32));
    jmv[r] = kv[r] == ~0ull ? -1 : (int)(unsigned)(kv[r] & 0xFFFFFFFFull);
    if (i < a.n_src) {
      kw[i] = ~0ull;
      d2w[i] = d2v[r];
      jw[i] = jmv[r];
    }
    sel[r] = i < a.n_src && (a.max_corr2 >= 0.f ? d2v[r] <= a.max_corr2 : true);
  }
  double acc[kRedPlane];
#pragma unroll
  for (int k = 0; k < kRedPlane; ++k) acc[k] = 0.0;
  double e_acc = 0.0;
#pragma unroll
  for (int r = 0; r < kSumR; ++r) {
    const int i = b0 + kSumR * tid + r;
    const float d2 = d2v[r];
    const int jm = jmv[r];
    if (sel[r] && jm >= 0 && a.metric == 1) {
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
      for (int rr = 0; rr < 6; ++rr)
#pragma unroll
        for (int c = rr; c < 6; ++c) acc[t++] += row[rr] * row[c];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) acc[22 + rr] += row[rr] * rhs;
      e_acc += (double)d2;
    } else if (sel[r] && jm >= 0) {
      const float4 s = a.src[i];
      const float4 m = a.tgt[jm];
      acc[0] += 1.0;
      acc[1] += s.x; acc[2] += s.y; acc[3] += s.z;
      acc[4] += m.x; acc[5] += m.y; acc[6] += m.z;
      acc[7] += (double)s.x * m.x; acc[8] += (double)s.x * m.y; acc[9] += (double)s.x * m.z;
      acc[10] += (double)s.y * m.x; acc[11] += (double)s.y * m.y; acc[12] += (double)s.y * m.z;
      acc[13] += (double)s.z * m.x; acc[14] += (double)s.z * m.y; acc[15] += (double)s.z * m.z;
      e_acc += (double)d2;
    }
  }
#pragma unroll
  for (int k = 0; k < kRedPlane; ++k)
    if (k < 16 || a.metric == 1)
      acc[k] = wave_sum_f64(acc[k]);
  e_acc = wave_sum_f64(e_acc);
  if (lane == 0) {
    for (int k = 0; k < kRedPlane; ++k) s_red[wave * (kRedPlane + 1) + k] = acc[k];
    s_red[wave * (kRedPlane + 1) + kRedPlane] = e_acc;
  }
  __syncthreads();
  if (tid <= kRedPlane) {   // one thread per sum, waves added in order
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < kIcpThreads / 64; ++w) v += s_red[w * (kRedPlane + 1) + tid];
    part[((size_t)pose * n_blk + blk) * kPartStride + tid] = v;
  }
}

// Split-path correspondences: grid (source chunks, target chunks, poses); a block keeps 4 source
// points per lane in VGPRs, stages its 1024-point target chunk in LDS and publishes each source
// point's best (d2, j) with one 64-bit atomic min -- min over the key is min d2, then lowest j,
// i.e. exactly what the strict `<` scan of the persistent kernel returns.
constexpr int kNnThreads = 128;   // 512 source points per workgroup: 2500 sources -> 5 workgroups, 98 % full
constexpr int kNnTgt = 512;    // 8 KB of LDS per 2-wave workgroup: 8 waves per SIMD resident

// LIST: only the queries list[pose][0 .. cnt[pose]) (those the open grid search could not settle, icp_nn_grid_open)
template <bool LIST>
__global__ __launch_bounds__(kNnThreads) void icp_nn_split(IcpArgs a, const int* __restrict__ list, const int* __restrict__ cnt) {
  __shared__ float4 s_t[kNnTgt];
  const int pose = blockIdx.z;
  if (a.st_done[pose]) return;
  const int n_q = LIST ? cnt[pose] : a.n_src;
  if (LIST && (int)(blockIdx.x * kNnThreads * kIcpR) >= n_q) return;   // the grid is sized for every query unsettled
  const int* qlist = LIST ? list + (size_t)pose * a.n_src : nullptr;
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
  int qi[kIcpR];   // the query of slot r (-1: none)
#pragma unroll
  for (int r = 0; r < kIcpR; ++r) {
    const int slot = base + r * kNnThreads + tid;
    int i = slot < n_q ? (LIST ? qlist[slot] : slot) : -1;
    qi[r] = i;
    float4 s = i >= 0 ? a.src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
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
    const int jp = i >= 0 ? jprev[i] : -1;
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
    const int i = qi[r];
    if (i >= 0 && bj[r] >= 0)
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

// Nearest target point of (x, y, z) among the 27 cells around its own, by the 16 lanes of a DPP row: lane r < 9 walks the
// three x-neighbours of row (dz, dy) = (r / 3 - 1, r % 3 - 1) -- consecutive cells, one contiguous range of points --, then the
// row's minimum key (d2 bits << 32 | j: smallest d2, then lowest j, the scan's rule; ~0: nothing) is formed by four
// exchanges.  (One lane per query walked ~300 points through a chain of dependent loads: 73 us per iteration for the
// reference's table alignment, 30 000 queries on 118 workgroups.)
__device__ __forceinline__ unsigned long long grid_nn27(const IcpArgs& a, float x, float y, float z, int lane16) {
  float best = FLT_MAX;
  int bj = -1;
  int cx, cy, cz;
  if (lane16 < 9 && grid_cell_of(a, x, y, z, &cx, &cy, &cz)) {
    const int zz = cz + lane16 / 3 - 1, yy = cy + lane16 % 3 - 1;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, a.gnx - 1);
    if (zz >= 0 && zz < a.gnz && yy >= 0 && yy < a.gny && x0 <= x1) {
      const size_t c0 = ((size_t)zz * a.gny + yy) * a.gnx + x0;
      const uint32_t b = a.gcell_start[c0], e = a.gcell_start[c0 + (x1 - x0) + 1];
      // four points per trip, their loads issued together (one point per trip was a chain of ~30 dependent L2 round trips
      // per lane); the comparisons stay in point order, so the winner is the one-by-one walk's
#ifndef PGP_GRID_UNROLL
#define PGP_GRID_UNROLL 4
#endif
      for (uint32_t k = b; k < e; k += PGP_GRID_UNROLL) {
        float4 m[PGP_GRID_UNROLL];
#pragma unroll
        for (int u = 0; u < PGP_GRID_UNROLL; ++u) m[u] = a.gpts[min(k + (uint32_t)u, e - 1u)];
#pragma unroll
        for (int u = 0; u < PGP_GRID_UNROLL; ++u) {
          const float dx = __fsub_rn(x, m[u].x), dyy = __fsub_rn(y, m[u].y), dzz = __fsub_rn(z, m[u].z);
          const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dyy, dyy), __fmul_rn(dzz, dzz)));
          const int j = __float_as_int(m[u].w);
          if (k + (uint32_t)u < e && (d2 < best || (d2 == best && j < bj))) {   // the scan's rule: smallest d2, then lowest j
            best = d2;
            bj = j;
          }
        }
      }
    }
  }
  unsigned long long key = bj >= 0 ? ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bj : ~0ull;   // d2 >= 0: bits order as values
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(key, off, 16);
    key = o < key ? o : key;
  }
  return key;
}

// The same search with the candidates of the 27 cells FLATTENED over L lanes (round 5): the nine (dz, dy) rows are nine
// contiguous ranges of points; lane l of the query's group tests the candidates l, l + L, l + 2 L ... of their concatenation.
// With one lane per row, a planar target (a table top) kept three lanes busy with ~28 points each while six found their
// rows empty; flattened, sixteen lanes take ~5 each and eight ~10.  The key minimum is a total order on (d2, j), so however
// the candidates are dealt out the group's minimum is the row-wise walk's -- and the scan's -- key.
template <int L, int U = 4>   // U: candidates per lane whose loads are in flight together
__device__ __forceinline__ unsigned long long grid_nn27_flat(const IcpArgs& a, float x, float y, float z, int laneL) {
  static_assert(L == 8 || L == 16, "a group is a power-of-two slice of a DPP row");
  uint32_t rb[9], pre[10];
  pre[0] = 0u;
  int cx, cy, cz;
  const bool in = grid_cell_of(a, x, y, z, &cx, &cy, &cz) != 0;
  const int x0 = in ? max(cx - 1, 0) : 0, x1 = in ? min(cx + 1, a.gnx - 1) : -1;
  uint32_t re[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {   // every lane of the group reads the same eighteen starts: one round trip, broadcast lines
    const int zz = cz + r / 3 - 1, yy = cy + r % 3 - 1;
    const bool ok = in && zz >= 0 && zz < a.gnz && yy >= 0 && yy < a.gny && x0 <= x1;
    const size_t c0 = ok ? ((size_t)zz * a.gny + yy) * a.gnx + x0 : 0;
    rb[r] = a.gcell_start[c0];
    re[r] = ok ? a.gcell_start[c0 + (x1 - x0) + 1] : rb[r];
  }
#pragma unroll
  for (int r = 0; r < 9; ++r) pre[r + 1] = pre[r] + (re[r] - rb[r]);
  const uint32_t T = pre[9];
  float best = FLT_MAX;
  int bj = -1;
  for (uint32_t t0 = (uint32_t)laneL; t0 < T; t0 += (uint32_t)(U * L)) {
    float4 m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t t = min(t0 + (uint32_t)(u * L), T - 1u);
      // the row of candidate t: the last r with pre[r] <= t
      uint32_t base = rb[0], p0 = 0u;
#pragma unroll
      for (int r = 1; r < 9; ++r) {
        const bool ge = t >= pre[r];
        base = ge ? rb[r] : base;
        p0 = ge ? pre[r] : p0;
      }
      m[u] = a.gpts[base + (t - p0)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float dx = __fsub_rn(x, m[u].x), dyy = __fsub_rn(y, m[u].y), dzz = __fsub_rn(z, m[u].z);
      const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dyy, dyy), __fmul_rn(dzz, dzz)));
      const int j = __float_as_int(m[u].w);
      if (t0 + (uint32_t)(u * L) < T && (d2 < best || (d2 == best && j < bj))) {   // the scan's rule: smallest d2, then lowest j
        best = d2;
        bj = j;
      }
    }
  }
  unsigned long long key = bj >= 0 ? ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bj : ~0ull;   // d2 >= 0: bits order as values
#pragma unroll
  for (int off = L / 2; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(key, off, L);
    key = o < key ? o : key;
  }
  return key;
}

constexpr int kGridLanes = 16;   // lanes per query in icp_nn_grid / icp_nn_grid_open

__global__ __launch_bounds__(256) void icp_nn_grid(IcpArgs a) {
  const int pose = blockIdx.y;
  if (a.st_done[pose]) return;
  const int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) / kGridLanes), lane16 = threadIdx.x & (kGridLanes - 1);
  if (i >= a.n_src) return;   // whole rows: a row's lanes share i
  const float* G = a.T + 16 * (size_t)pose;
  const float4 s = a.src[i];
  const float x = row_xf(G[0], G[4], G[8], G[12], s.x, s.y, s.z);
  const float y = row_xf(G[1], G[5], G[9], G[13], s.x, s.y, s.z);
  const float z = row_xf(G[2], G[6], G[10], G[14], s.x, s.y, s.z);
  const unsigned long long key = grid_nn27_flat<kGridLanes>(a, x, y, z, lane16);
  if (lane16 == 0 && key != ~0ull) a.ws_key[(size_t)pose * a.n_src + i] = key;
}

// The same search WITHOUT a correspondence cap, for targets beyond the exact index's 65 535 points (round 4): the 27
// cells around a query hold every target point within r_safe of it, so a nearest neighbour found at d2 <= r2_safe is the
// nearest of the whole target (same d2, same lowest-j rule as the scan) and is written as the query's key; a query whose
// 27 cells hold nothing that close is appended to its pose's list and settled by the exhaustive scan over the listed
// queries only (icp_nn_split<true>).  Which queries end on the list does not change any result.
__global__ __launch_bounds__(256) void icp_nn_grid_open(IcpArgs a, float r2_safe, int* __restrict__ list, int* __restrict__ cnt) {
  const int pose = blockIdx.y;
  if (a.st_done[pose]) return;
  const int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) / kGridLanes), lane16 = threadIdx.x & (kGridLanes - 1);
  if (i >= a.n_src) return;
  const float* G = a.T + 16 * (size_t)pose;
  const float4 s = a.src[i];
  const float x = row_xf(G[0], G[4], G[8], G[12], s.x, s.y, s.z);
  const float y = row_xf(G[1], G[5], G[9], G[13], s.x, s.y, s.z);
  const float z = row_xf(G[2], G[6], G[10], G[14], s.x, s.y, s.z);
  const unsigned long long key = grid_nn27(a, x, y, z, lane16);
  if (lane16 != 0) return;
  if (key != ~0ull && __uint_as_float((unsigned)(key >> 32)) <= r2_safe) {
    a.ws_key[(size_t)pose * a.n_src + i] = key;
  } else if (x == x && y == y && z == z) {   // (a non-finite query has no neighbour in the scan either: no key)
    const int slot = atomicAdd(&cnt[pose], 1);
    list[(size_t)pose * a.n_src + slot] = i;
  }
}


// ---- the scene-sized capped ICP as ONE launch (round 5) --------------------------------------------------------------
// The reference's one LIVE ICP call aligns the whole scene to the table model (SceneCfg.cpp:101,135-141: ~30 000 source
// points, a 100 000-point target, correspondence cap 1 cm, <= 50 iterations).  Host-driven, an iteration was three launches
// (icp_nn_grid, icp_sums_partial, icp_refine<true, true>) and every fourth a host synchronisation: 63 us of which 42 in
// kernels.  Here ONE launch of resident workgroups (launch_resident) runs every iteration, with the arithmetic of those three
// kernels operation for operation (same grid search, same 4-points-per-lane accumulation, same wave tree, the same
// order of the wave and block sums), so transforms, energies and iteration counts are theirs bit for bit:
//   chunk  = 16 consecutive source points: one round of a 256-thread workgroup, 16 lanes per query (grid_nn27); keys go
//            to memory write-through;
//   unit   = 256 consecutive points = the 16 chunks whose keys one WAVE of icp_sums_partial would sum: the workgroup
//            whose chunk arrives last at the unit's ticket forms that wave's sums (wave 0, lanes as icp_sums_partial's);
//   pose   = the workgroup whose unit arrives last at the pose's ticket adds the unit sums in icp_sums_partial's wave
//            order and icp_refine's block order, solves, applies the stop rules and publishes the next transform;
// everybody else polls the pose's state word (iterations completed | done << 31).  Three tickets deep, no grid-wide
// barrier, no host round trip.  A ticket is a returning agent-scope add behind the arriver's drained write-through
// stores (MI355X_MICROARCH.md, handoff-flag); whoever draws the last number reads the others' data with agent-scope loads.
struct SceneArgs {
  unsigned long long* keys;   // [n][n_src] (d2 bits << 32) | j, ~0 = no neighbour in the 27 cells
  double* W;                  // [n][n_units16][kPartStride]: the unit sums; units past the cloud stay zero
  unsigned* unit_ctr;         // [n][n_units16] chunk arrivals, monotone over the iterations
  unsigned* pose_ctr;         // [n] unit arrivals, monotone
  unsigned* state;            // [n] iterations completed | done << 31
  double* E_old;              // [n] mean squared distance of the previous iteration
  unsigned* lost;             // [1] set when a wait ran into its clock bound
  int n_chunks, n_units, n_units16, n_blk;
  int poll_sleep;             // s_sleep argument between two polls of a pose's state word
  int n_upd;                  // workgroups 0 .. n_upd-1 are updaters (pose p: updater p % n_upd), the rest workers
  int force_lost;             // test knob (PGP_ICP_FORCE_LOST): every pose's first wait for its units counts as run out
  unsigned wait_ticks;        // floor of every wait's clock bound (100 MHz ticks; 3 ms, PGP_ICP_WAIT_MS)
  unsigned long long* dbg;    // PGP_SCENE_STAMPS builds: [64 iterations][16] clock stamps (100 MHz) of updater 0 and of three workers
};
#ifdef PGP_SCENE_STAMPS
#define SCENE_STAMP(slot) do { if (it < 64) z.dbg[it * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SCENE_STAMP(slot) do { } while (0)
#endif
constexpr int kSceneThreads = 256;
constexpr int kSceneLanes = 8;                         // lanes per query (grid_nn27_flat)
constexpr int kSceneQ = kSceneThreads / kSceneLanes;   // queries per chunk: 32
constexpr int kSceneCpu = 256 / kSceneQ;               // chunks per unit: 8

template <class T>
__device__ __forceinline__ T agent_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ void agent_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int kSceneRep = 32;       // copies of a pose's state word, one 128-byte line each: 1000 polling workgroups on ONE word
constexpr int kSceneRepStride = 32; // (in words) queue behind each other at the memory side and hold up the updater's own traffic
constexpr int kSceneUpdMax = 8;     // workgroups in the updater role

// Two ROLES in one launch.  Workgroups 0 .. n_upd-1 are updaters: pose p belongs to updater p % n_upd, which waits for the
// pose's unit counter, adds the unit sums in the reference order, solves, applies the stop rules and publishes the next
// transform -- the code with the solvers' calls and ~120 registers.  All other workgroups are workers: search their
// chunks, close units.  (One body for both had the compiler keep the solvers' calling-convention spills and 150-190
// registers around the search loop: 3 waves per SIMD and scratch traffic in the hot path; measured 76 us per iteration
// against 57 host-driven.)
#ifdef PGP_SCENE_WAVES   // A/B knob (make variantf FILE=icp): force the register budget of N waves per SIMD
#define PGP_SCENE_ATTR __attribute__((amdgpu_waves_per_eu(PGP_SCENE_WAVES, PGP_SCENE_WAVES)))
#else
#define PGP_SCENE_ATTR
#endif
template <int METRIC>
__global__ __launch_bounds__(kSceneThreads) PGP_SCENE_ATTR void icp_scene_persist(IcpArgs a, SceneArgs z) {
  __shared__ float s_G[16], s_G_old[16];
  __shared__ int s_flag[2];
  __shared__ double s_sum[kRedPlane + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  static_assert(kSumR == 4 && kSceneQ * kSceneCpu == 256, "a unit is one wave of icp_sums_partial: 64 lanes x 4 consecutive points");
  if ((int)blockIdx.x < z.n_upd) {
    // ================= updater =================
    __shared__ double s_blk[8][kRedPlane + 1];   // the block sums of one pass (eight blocks), formed by four waves
    __shared__ int s_upd[2];
    unsigned long long upd_longest = 0;   // (thread 0) the longest wait for a pose's units that ended well, in ticks
    for (int it = 0;; ++it) {
      bool any = false;
      for (int pose = blockIdx.x; pose < a.n; pose += z.n_upd) {
        // (this workgroup's own last publication: the same answer in every wave)
        unsigned st = agent_load(&z.state[(size_t)pose * kSceneRep * kSceneRepStride]);
        if (st >> 31) continue;
        any = true;
        // ---- every unit of the pose has arrived (their sums are in memory: written through before the arrival)
        const unsigned target = (unsigned)z.n_units * (unsigned)(it + 1);
        if (tid == 0) {
          // The bound follows the WORK: 64 x the longest wait this updater has seen so far, a few milliseconds at least
          // (z.wait_ticks).  An iteration is ~25 us; up to round 5 a fixed 2 s stood here -- with the workgroups of two
          // processes each holding a part of the device, every lost wait cost the caller those 2 s.
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          const unsigned long long bound = max((unsigned long long)z.wait_ticks, 64ull * upd_longest);
          int lost = 0;
          while (agent_load(&z.pose_ctr[pose]) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > bound) {
              lost = 1;
              break;
            }
          }
          if (!lost) upd_longest = max(upd_longest, __builtin_amdgcn_s_memrealtime() - t0);
          if (z.force_lost && it == 0) lost = 1;
          s_upd[0] = lost;
        }
        __syncthreads();
        if (s_upd[0]) {
          if (tid == 0) {
            agent_store(z.lost, 1u);
            if (a.iters) a.iters[pose] = -1;
          }
          if (tid < kSceneRep) agent_store(&z.state[((size_t)pose * kSceneRep + tid) * kSceneRepStride], 0x80000000u);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();   // (every wave reads the state word at the top of the next pass)
          continue;
        }
        if (tid == 0 && pose == 0) SCENE_STAMP(0);   // every unit has arrived
        if (tid < 16) s_G[tid] = agent_load(&a.T[16 * (size_t)pose + tid]);
        // ---- the pose's sums: icp_sums_partial adds its 16 waves in order from 0.0, icp_refine<true, true> the blocks in
        // order from 0.0.  Eight blocks per pass: lane k + 32 (b & 1) of wave b / 2 adds the sixteen wave sums of block b
        // (their loads in flight together); thread k then adds the pass's blocks in order.
        double total = 0.0;
        for (int bb = 0; bb < z.n_blk; bb += 8) {
          const int k = lane & 31, b = bb + 2 * wave + (lane >> 5);
          double w16[16];
#pragma unroll
          for (int w = 0; w < 16; ++w)
            w16[w] = (b < z.n_blk && k <= kRedPlane) ? agent_load(&z.W[((size_t)pose * z.n_units16 + 16 * b + w) * kPartStride + k]) : 0.0;
          double v = 0.0;
#pragma unroll
          for (int w = 0; w < 16; ++w) v += w16[w];
          if (k <= kRedPlane) s_blk[b - bb][k] = v;
          __syncthreads();
          if (tid <= kRedPlane)
            for (int q = 0; q < 8 && bb + q < z.n_blk; ++q) total += s_blk[q][tid];
          __syncthreads();
        }
        if (tid <= kRedPlane) s_sum[tid] = total;
        __syncthreads();
        // ---- closed-form update and the progress tests (icp_refine's, thread 0's)
        int go_i = 0;
        if (tid == 0 && pose == 0) SCENE_STAMP(1);   // sums added
        if (tid == 0) {
          const double* red = s_sum;
          const double E = red[0] >= 1.0 ? red[kRedPlane] / red[0] : 0.0;
          for (int q = 0; q < 16; ++q) s_G_old[q] = s_G[q];
          if constexpr (METRIC == 1) solve_plane(red, s_G);
          else solve_rigid(red, s_G);
          if (pose == 0) SCENE_STAMP(2);   // solved
          const double E_old = it == 0 ? (double)FLT_MAX : z.E_old[pose];   // PCL: energy starts at numeric_limits<float>::max()
          bool go = it + 1 < a.max_iter;
          if (a.ratio > 0.f && !(E / E_old < (double)a.ratio)) go = false;
          if (red[0] < 1.0) go = false;
          if (converged_extra(stop_rules_of(a), pose, it + 1, s_G_old, s_G, E, E_old)) go = false;
          z.E_old[pose] = E;   // (this workgroup's own: nobody else reads it)
          if (!go) {
            if (a.energy) a.energy[pose] = (float)E;
            if (a.iters) a.iters[pose] = it + 1;
          }
          go_i = go ? 1 : 0;
          s_upd[1] = go_i;
        }
        __syncthreads();
        go_i = s_upd[1];
        if (tid < 16) agent_store(&a.T[16 * (size_t)pose + tid], s_G[tid]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (wave 0 holds both the transform's stores and the state's)
        // every copy of the state word by ONE store instruction, a line each
        if (tid < kSceneRep)
          agent_store(&z.state[((size_t)pose * kSceneRep + tid) * kSceneRepStride], (unsigned)(it + 1) | (go_i ? 0u : 0x80000000u));
        if (tid == 0 && pose == 0) SCENE_STAMP(3);   // published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // (every wave reads the state word at the top of the next pass)
      }
      if (!any) break;
    }
    return;
  }
  // ================= worker =================
  const int items = a.n * z.n_chunks, n_work = (int)gridDim.x - z.n_upd, me = (int)blockIdx.x - z.n_upd;
  const unsigned* my_state = z.state + (size_t)(me % kSceneRep) * kSceneRepStride;
  unsigned long long wrk_longest = 0;   // (thread 0) the longest wait for a state word that ended well, in ticks
  for (int it = 0;; ++it) {
    bool any = false;
    for (int item = me; item < items; item += n_work) {
      const int pose = item / z.n_chunks, chunk = item - pose * z.n_chunks;
      // (the chunk's source point does not depend on the transform: its round trip passes under the wait)
      const int qi = chunk * kSceneQ + tid / kSceneLanes, laneL = tid & (kSceneLanes - 1);
      const float4 sq = a.src[min(qi, a.n_src - 1)];
      // ---- the pose's transform of this iteration: published by the updater when it closed the previous one
      if (tid == 0) {
        unsigned st = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        // (twice the updaters' floor: an updater that is there gives up -- and says so in the state word -- before its workers do)
        const unsigned long long bound = max(2ull * (unsigned long long)z.wait_ticks, 64ull * wrk_longest);
        for (;;) {
          st = agent_load(&my_state[(size_t)pose * kSceneRep * kSceneRepStride]);
          if ((st >> 31) != 0u || (st & 0x7FFFFFFFu) >= (unsigned)it) break;
          for (int q = 0; q < z.poll_sleep; ++q) __builtin_amdgcn_s_sleep(8);
          if (__builtin_amdgcn_s_memrealtime() - t0 > bound) {
            // The pose's updater is not there (its workgroup never became resident: another process holds the compute
            // units): the pose is ABANDONED for everybody -- the state word says done in all its copies, so that no other
            // worker (and no later item of this one) waits for it again, and the iteration count says lost (the host-pointer
            // calls redo the job host-driven; an updater that turns up late finds the pose closed).
            agent_store(z.lost, 1u);
            if (a.iters) a.iters[pose] = -1;
            for (int r = 0; r < kSceneRep; ++r) agent_store(&z.state[((size_t)pose * kSceneRep + r) * kSceneRepStride], 0x80000000u);
            st = 0x80000000u;
            break;
          }
        }
        if ((st >> 31) == 0u) wrk_longest = max(wrk_longest, __builtin_amdgcn_s_memrealtime() - t0);
        s_flag[0] = (int)(st >> 31);
      }
      __syncthreads();
      const bool pose_done = s_flag[0] != 0;
#ifdef PGP_SCENE_STAMPS
      const int dslot = me == 0 ? (item == me ? 4 : 8) : -1;
      if (tid == 0 && dslot >= 0) SCENE_STAMP(dslot);       // the state word said go
      if (tid == 0 && it == 10 && me < 2048) z.dbg[1024 + me] = __builtin_amdgcn_s_memrealtime();   // every worker's go
#endif
      if (!pose_done && tid < 16) s_G[tid] = agent_load(&a.T[16 * (size_t)pose + tid]);
      __syncthreads();   // (also: everybody has read s_flag[0] before thread 0 writes it again)
      if (pose_done) continue;
      any = true;
      // ---- 1. the chunk's correspondences (icp_nn_grid's arithmetic)
      {
        const int i = qi;
        if (i < a.n_src) {   // whole groups: a group's lanes share i
          const float4 s = sq;
          const float x = row_xf(s_G[0], s_G[4], s_G[8], s_G[12], s.x, s.y, s.z);
          const float y = row_xf(s_G[1], s_G[5], s_G[9], s_G[13], s.x, s.y, s.z);
          const float zq = row_xf(s_G[2], s_G[6], s_G[10], s_G[14], s.x, s.y, s.z);
          const unsigned long long key = grid_nn27_flat<kSceneLanes, 4>(a, x, y, zq, laneL);
          if (laneL == 0) agent_store(&z.keys[(size_t)pose * a.n_src + i], key);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's keys have left ...
      __syncthreads();                                    // ... and so have every wave's, before the chunk's ticket
#ifdef PGP_SCENE_STAMPS
      if (tid == 0 && dslot >= 0) SCENE_STAMP(dslot + 1);   // searched, keys out
      if (tid == 0 && it == 10 && me < 2048) z.dbg[1024 + 2048 + me] = __builtin_amdgcn_s_memrealtime();   // every worker's chunk searched
#endif
      const int u = chunk / kSceneCpu;
      if (tid == 0) {
        const unsigned per = (unsigned)min(kSceneCpu, z.n_chunks - kSceneCpu * u);
        const unsigned old = __hip_atomic_fetch_add(&z.unit_ctr[(size_t)pose * z.n_units16 + u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_flag[1] = old + 1u == per * (unsigned)(it + 1) ? 1 : 0;
      }
      __syncthreads();
#ifdef PGP_SCENE_STAMPS
      if (tid == 0 && dslot >= 0) SCENE_STAMP(dslot + 2);   // ticket drawn
#endif
#ifdef PGP_SCENE_STAMPS
#endif
      if (!s_flag[1] || wave != 0) continue;   // (the other waves wait at the next item's barrier while wave 0 closes the unit)
      // ---- 2. the unit's sums: lane l holds the points 256 u + 4 l .. + 3, as thread 64 w + l of icp_sums_partial does
      const int b0 = 256 * u;
      unsigned long long kv[kSumR];
#pragma unroll
      for (int r = 0; r < kSumR; ++r) {
        const int i = b0 + kSumR * lane + r;
        kv[r] = i < a.n_src ? agent_load(&z.keys[(size_t)pose * a.n_src + i]) : ~0ull;
      }
      constexpr int kNs = METRIC == 1 ? kRedPlane : 16;   // sums in use
      double acc[kNs];
#pragma unroll
      for (int k = 0; k < kNs; ++k) acc[k] = 0.0;
      double e_acc = 0.0;
#pragma unroll
      for (int r = 0; r < kSumR; ++r) {
        const int i = b0 + kSumR * lane + r;
        const float d2 = kv[r] == ~0ull ? FLT_MAX : __uint_as_float((unsigned)(kv[r] >> 32));
        const int jm = kv[r] == ~0ull ? -1 : (int)(unsigned)(kv[r] & 0xFFFFFFFFull);
        const bool sel = i < a.n_src && (a.max_corr2 >= 0.f ? d2 <= a.max_corr2 : true);
        if constexpr (METRIC == 1) {
          if (sel && jm >= 0) {
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
            for (int rr = 0; rr < 6; ++rr)
#pragma unroll
              for (int c = rr; c < 6; ++c) acc[t++] += row[rr] * row[c];
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) acc[22 + rr] += row[rr] * rhs;
            e_acc += (double)d2;
          }
        } else if (sel && jm >= 0) {
          const float4 s = a.src[i];
          const float4 m = a.tgt[jm];
          acc[0] += 1.0;
          acc[1] += s.x; acc[2] += s.y; acc[3] += s.z;
          acc[4] += m.x; acc[5] += m.y; acc[6] += m.z;
          acc[7] += (double)s.x * m.x; acc[8] += (double)s.x * m.y; acc[9] += (double)s.x * m.z;
          acc[10] += (double)s.y * m.x; acc[11] += (double)s.y * m.y; acc[12] += (double)s.y * m.z;
          acc[13] += (double)s.z * m.x; acc[14] += (double)s.z * m.y; acc[15] += (double)s.z * m.z;
          e_acc += (double)d2;
        }
      }
#pragma unroll
      for (int k = 0; k < kNs; ++k) acc[k] = wave_sum_f64(acc[k]);
      e_acc = wave_sum_f64(e_acc);
      double* Wu = z.W + ((size_t)pose * z.n_units16 + u) * kPartStride;
      {
        // every lane holds every sum (wave_sum_f64): lane k stores sum k -- ONE store instruction over three lines instead
        // of 29 single-lane write-through stores, each a fabric write of its own
        double mine = lane == kRedPlane ? e_acc : 0.0;
#pragma unroll
        for (int k = 0; k < kNs; ++k) mine = lane == k ? acc[k] : mine;
        if (lane <= kRedPlane) agent_store(&Wu[lane], mine);
      }
#ifdef PGP_SCENE_STAMPS
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the unit's sums have left before its arrival is counted
      if (lane == 0) __hip_atomic_fetch_add(&z.pose_ctr[pose], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef PGP_SCENE_STAMPS
      if (lane == 0 && it < 64) atomicMax(&z.dbg[it * 16 + 12], __builtin_amdgcn_s_memrealtime());   // the LAST unit closed (any worker)
#endif
    }
    if (!any) break;
  }
}

// ---- the exact index: build ----------------------------------------------------------------------
__device__ __forceinline__ int nn_axis(float v, float o, float inv_h, int n) {
  // monotone in v (every step is): the cells of [x - r, x + r] bracket the cell of any point in between
  float f = __fmul_rn(__fsub_rn(v, o), inv_h);
  f = fminf(fmaxf(f, 0.f), (float)(n - 1));   // NaN -> 0
  return (int)f;
}

template <bool FILL>
__global__ __launch_bounds__(256) void nnidx_scatter(const float4* __restrict__ tgt, int n_tgt, NnGeom g,
                                                     uint32_t* __restrict__ ctr, const uint32_t* __restrict__ start,
                                                     float4* __restrict__ pts) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_tgt) return;
  const float4 p = tgt[j];
  const uint32_t c = ((uint32_t)nn_axis(p.z, g.oz, g.inv_h, g.nz) * (uint32_t)g.ny + (uint32_t)nn_axis(p.y, g.oy, g.inv_h, g.ny)) *
                         (uint32_t)g.nx + (uint32_t)nn_axis(p.x, g.ox, g.inv_hx, g.nx);
  const uint32_t slot = atomicAdd(&ctr[c], 1u);
  if (FILL) pts[start[c] + slot] = make_float4(p.x, p.y, p.z, __int_as_float(j));
}

// representative of every cell: exhaustive nearest point (in image order) of the cell's centre;
// grid (cells / 256, target chunks of 512), one 64-bit atomic-min key per cell
__global__ __launch_bounds__(256) void nnidx_rep(const float4* __restrict__ pts, int n_tgt, NnGeom g,
                                                 unsigned long long* __restrict__ key) {
  __shared__ float4 s_t[512];
  const int t0 = blockIdx.y * 512, tn = min(512, n_tgt - t0);
  for (int k = threadIdx.x; k < tn; k += 256) s_t[k] = pts[t0 + k];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= g.n_cells) return;
  const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
  const float x = g.ox + ((float)cx + 0.5f) * g.hx, y = g.oy + ((float)cy + 0.5f) * g.h, z = g.oz + ((float)cz + 0.5f) * g.h;
  float best = FLT_MAX;
  int bp = -1;
  for (int k = 0; k < tn; ++k) {
    const float4 m = s_t[k];
    const float dx = x - m.x, dy = y - m.y, dz = z - m.z;
    const float d2 = dx * dx + (dy * dy + dz * dz);
    if (d2 < best) {
      best = d2;
      bp = t0 + k;
    }
  }
  if (bp >= 0) atomicMin(&key[c], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bp);
}

// the 16-bit tables of the image
__global__ __launch_bounds__(256) void nnidx_pack(NnGeom g, const uint32_t* __restrict__ start,
                                                  const unsigned long long* __restrict__ key,
                                                  unsigned char* __restrict__ image) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  uint16_t* s16 = reinterpret_cast<uint16_t*>(image + g.off_start);
  uint16_t* r16 = reinterpret_cast<uint16_t*>(image + g.off_rep);
  if (c <= g.n_cells) s16[c] = (uint16_t)start[c];
  if (c < g.n_cells) {
    const unsigned long long k = key[c];
    r16[c] = k == ~0ull ? (uint16_t)0 : (uint16_t)(k & 0xFFFFull);   // no finite point at all: any position
  }
}

// ---- the vicinity graph: per target point its kVicK nearest other target points + a clearance radius ---------
// Why: once a pose is within a few millimetres, a query's nearest neighbour is the previous iteration's
// correspondence p or a point next to it, and the row search below only CONFIRMS that (its cost, ~1000
// instructions per query, is the price of exactness).  The triangle inequality confirms it for the price of
// kVicK + 1 distance tests: let N(p) be the kVicK target points nearest to p and R(p) a lower bound on |p - p'|
// for every target point p' outside C = {p} u N(p) (the distance to the (kVicK+1)-th nearest, rounded down).
// With c the best candidate of C under the scan's rule, a = |x - p|, b = |x - c|:  every p' outside C has
// |x - p'| >= R(p) - a, so  a + b < R(p)  proves that c is the exact answer (no outsider can even tie).  The
// float test keeps a relative margin of 1e-4 on a and b and 1e-3 on R against the ~3e-7 rounding of the computed
// squared distances, so a query that passes has the same (d2, lowest original index) as the exhaustive scan.
// A query that fails goes through the row search as before, with the graph's best candidate as its bound
// (tighter than the cell representative: fewer rows).  The centre p is the better of the query's previous
// correspondence and its cell's representative.
// Record (uint4): seven 16-bit image positions + the upper 16 bits of the float R (truncated = rounded down).
constexpr int kVicK = 7;
constexpr int kVicChunk = 1024;       // target points per block of the partial pass
constexpr int kVicMaxTargets = 16384; // brute-force build: 2.7e8 tests at most

__device__ __forceinline__ void vic_insert(float (&bd)[kVicK + 1], int (&bp)[kVicK + 1], float d2, int pos) {
  if (d2 < bd[kVicK]) {   // false for NaN; +inf and FLT_MAX never enter (a non-finite point is nobody's neighbour)
    bd[kVicK] = d2;
    bp[kVicK] = pos;
#pragma unroll
    for (int i = kVicK; i > 0; --i) {
      const bool sw = bd[i] < bd[i - 1];
      const float lo = sw ? bd[i] : bd[i - 1], hi = sw ? bd[i - 1] : bd[i];
      const int plo = sw ? bp[i] : bp[i - 1], phi = sw ? bp[i - 1] : bp[i];
      bd[i - 1] = lo;
      bd[i] = hi;
      bp[i - 1] = plo;
      bp[i] = phi;
    }
  }
}

// grid (points / 64, chunks): the kVicK + 1 nearest points of one chunk for every point (image order)
__global__ __launch_bounds__(64) void nnidx_vic_partial(const float4* __restrict__ pts, int n_tgt, float2* __restrict__ part) {
  __shared__ float4 s_t[kVicChunk];
  const int c0 = blockIdx.y * kVicChunk, cn = min(kVicChunk, n_tgt - c0);
  for (int k = threadIdx.x; k < cn; k += 64) s_t[k] = pts[c0 + k];
  __syncthreads();
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n_tgt) return;
  const float4 me = pts[p];
  float bd[kVicK + 1];
  int bp[kVicK + 1];
#pragma unroll
  for (int i = 0; i <= kVicK; ++i) {
    bd[i] = FLT_MAX;
    bp[i] = p;
  }
  for (int k = 0; k < cn; ++k) {
    const float4 m = s_t[k];
    const float dx = me.x - m.x, dy = me.y - m.y, dz = me.z - m.z;
    float d2 = dx * dx + (dy * dy + dz * dz);
    if (c0 + k == p) d2 = FLT_MAX;   // not its own neighbour
    vic_insert(bd, bp, d2, c0 + k);
  }
  float2* o = part + ((size_t)blockIdx.y * n_tgt + p) * (kVicK + 1);
#pragma unroll
  for (int i = 0; i <= kVicK; ++i) o[i] = make_float2(bd[i], __int_as_float(bp[i]));
}

__global__ __launch_bounds__(64) void nnidx_vic_merge(const float2* __restrict__ part, int n_tgt, int n_chunks,
                                                      uint4* __restrict__ vic) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n_tgt) return;
  float bd[kVicK + 1];
  int bp[kVicK + 1];
#pragma unroll
  for (int i = 0; i <= kVicK; ++i) {
    bd[i] = FLT_MAX;
    bp[i] = p;
  }
  for (int c = 0; c < n_chunks; ++c) {
    const float2* o = part + ((size_t)c * n_tgt + p) * (kVicK + 1);
#pragma unroll
    for (int i = 0; i <= kVicK; ++i) {
      const float2 v = o[i];
      vic_insert(bd, bp, v.x, __float_as_int(v.y));
    }
  }
  // fewer than kVicK + 1 other (finite) points: everything is listed, nothing lies outside -> R = +inf
  const float R = bd[kVicK] < FLT_MAX ? __builtin_sqrtf(bd[kVicK]) * 0.999f : __int_as_float(0x7F800000);
  const unsigned r16 = __float_as_uint(R) >> 16;   // truncation: never above R
  uint4 rec;
  rec.x = (unsigned)bp[0] | ((unsigned)bp[1] << 16);
  rec.y = (unsigned)bp[2] | ((unsigned)bp[3] << 16);
  rec.z = (unsigned)bp[4] | ((unsigned)bp[5] << 16);
  rec.w = (unsigned)bp[6] | (r16 << 16);
  vic[p] = rec;
}
static_assert(kVicK == 7, "the record holds seven positions and the radius");

// ---- the exact index: query (all tables in LDS) ----------------------------------------------------
struct NnLds {
  const float4* pts;
  const uint16_t* start;
  const uint16_t* rep;
  // per query of this workgroup (persist across the iterations of the persistent kernel)
  float* d2;          // squared distance of the correspondence (FLT_MAX: none)
  uint16_t* pos;      // its position in the image (0xFFFF: none)
  uint16_t* order;    // the queries sorted by estimated search cost (dearest first), then by where they fall
  uint32_t* bins;     // kNnBins 16-bit counters / offsets of that sort, two per word
#ifdef PGP_ICP_STAMPS
  unsigned* dbg;      // diagnostic build: rows / live rows / points / lane slots / queries
#endif
};
constexpr int kNnClasses = 16;
constexpr int kNnStrips = 512 / kPS;                 // runs of neighbouring cells a class is subdivided into (8 counters per thread, 16 classes)
constexpr int kNnBins = kNnClasses * kNnStrips;      // 16 KB of 16-bit counters

// (d2, original index) as ONE unsigned 64-bit key: d2 >= +0, so the float bits order like the values,
// and key < best is exactly the scan's rule (smaller d2, then lower index).  The initial key
// (FLT_MAX, 0) rejects NaN, +inf and d2 == FLT_MAX as the scan's strict `<` against FLT_MAX does.
__device__ __forceinline__ void nn_consider(float x, float y, float z, const float4 m, int pos,
                                            unsigned long long& best, int& bpos) {
  const float dx = __fsub_rn(x, m.x), dy = __fsub_rn(y, m.y), dz = __fsub_rn(z, m.z);
  const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dz, dz)));
  const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(m.w);
  if (key < best) {
    best = key;
    bpos = pos;
  }
}
constexpr unsigned long long kNnNone = (unsigned long long)0x7F7FFFFFu << 32;   // (FLT_MAX, 0)

__device__ __forceinline__ int nn_cell_of(const NnGeom& g, float x, float y, float z) {
  return (nn_axis(z, g.oz, g.inv_h, g.nz) * g.ny + nn_axis(y, g.oy, g.inv_h, g.ny)) * g.nx + nn_axis(x, g.ox, g.inv_hx, g.nx);
}

// The box of cells that can hold a point with computed d2 <= best: every such point lies within r of
// the query in each coordinate (1e-4 relative + 4e-7 |coordinates| cover the rounding of d2 and of x -/+ r).
struct NnBox {
  int x0, x1, y0, y1, z0, z1;
};
__device__ __forceinline__ NnBox nn_box(const NnGeom& g, float x, float y, float z, float best_d2) {
  const float r = sqrtf(best_d2) * 1.0001f + 4e-7f * (fabsf(x) + fabsf(y) + fabsf(z));
  NnBox b;
  b.x0 = nn_axis(x - r, g.ox, g.inv_hx, g.nx);
  b.x1 = nn_axis(x + r, g.ox, g.inv_hx, g.nx);
  b.y0 = nn_axis(y - r, g.oy, g.inv_h, g.ny);
  b.y1 = nn_axis(y + r, g.oy, g.inv_h, g.ny);
  b.z0 = nn_axis(z - r, g.oz, g.inv_h, g.nz);
  b.z1 = nn_axis(z + r, g.oz, g.inv_h, g.nz);
  return b;
}

// Phase A helpers.  nn_class: the cost class (0 cheapest .. 14) of the row search that starts from the bound d2.
__device__ __forceinline__ int nn_cost(const NnBox& b) { return (b.y1 - b.y0 + 1) * (b.z1 - b.z0 + 1) * (((b.x1 - b.x0) >> 2) + 3); }
// (round 5, tools/icp_lane_balance.py: the dearest lane of a wave-pass carries about twice the mean lane's work; ordering
// the queries of a class by the next two bits of their cost did not change that -- neighbouring queries' costs differ by
// their ROW lengths, which the box cost does not see -- and timed equal or slower: profiles/r05_ab/icp_lane_balance.log)
__device__ __forceinline__ int nn_class(const NnGeom& g, float x, float y, float z, float bound_d2, unsigned* rows = nullptr) {
  const NnBox b = nn_box(g, x, y, z, bound_d2);
  if (rows) *rows = (unsigned)((b.y1 - b.y0 + 1) * (b.z1 - b.z0 + 1));
  const int cost = nn_cost(b);   // rows x (row overhead + cells)
  const int c = 31 - __clz(cost);   // cost >= 3
  return c < kNnClasses - 2 ? c : kNnClasses - 2;
}
// One move downhill on the vicinity graph: the seven neighbours of the centre (record rec; best / bpos hold it) are
// considered; true when a neighbour is nearer (best / bpos then hold that neighbour, the next centre).
__device__ __forceinline__ bool nn_vic_move(const NnLds& t, const uint4 rec, float x, float y, float z,
                                            unsigned long long& best, int& bpos) {
  const int p = bpos;
  const int n0 = rec.x & 0xFFFFu, n1 = rec.x >> 16, n2 = rec.y & 0xFFFFu, n3 = rec.y >> 16, n4 = rec.z & 0xFFFFu,
            n5 = rec.z >> 16, n6 = rec.w & 0xFFFFu;
  nn_consider(x, y, z, t.pts[n0], n0, best, bpos);
  nn_consider(x, y, z, t.pts[n1], n1, best, bpos);
  nn_consider(x, y, z, t.pts[n2], n2, best, bpos);
  nn_consider(x, y, z, t.pts[n3], n3, best, bpos);
  nn_consider(x, y, z, t.pts[n4], n4, best, bpos);
  nn_consider(x, y, z, t.pts[n5], n5, best, bpos);
  nn_consider(x, y, z, t.pts[n6], n6, best, bpos);
  return bpos != p;
}

// The check on the vicinity graph (see nnidx_vic_*): the seven neighbours of the centre p (record rec; best / bpos
// hold p on entry) are considered; returns true when a + b < R(p): nothing outside {p} u N(p) comes closer than
// best or ties with it, i.e. best / bpos are the exhaustive scan's answer.
__device__ __forceinline__ bool nn_vic_check(const NnLds& t, const uint4 rec, float x, float y, float z,
                                             unsigned long long& best, int& bpos) {
  const float a2 = __uint_as_float((unsigned)(best >> 32));   // best IS p here: |x - p|^2
  const int n0 = rec.x & 0xFFFFu, n1 = rec.x >> 16, n2 = rec.y & 0xFFFFu, n3 = rec.y >> 16, n4 = rec.z & 0xFFFFu,
            n5 = rec.z >> 16, n6 = rec.w & 0xFFFFu;
  const float4 m0 = t.pts[n0], m1 = t.pts[n1], m2 = t.pts[n2], m3 = t.pts[n3], m4 = t.pts[n4], m5 = t.pts[n5], m6 = t.pts[n6];
  nn_consider(x, y, z, m0, n0, best, bpos);
  nn_consider(x, y, z, m1, n1, best, bpos);
  nn_consider(x, y, z, m2, n2, best, bpos);
  nn_consider(x, y, z, m3, n3, best, bpos);
  nn_consider(x, y, z, m4, n4, best, bpos);
  nn_consider(x, y, z, m5, n5, best, bpos);
  nn_consider(x, y, z, m6, n6, best, bpos);
  const float R = __uint_as_float(rec.w & 0xFFFF0000u);
  const float a = __builtin_sqrtf(a2), b = __builtin_sqrtf(__uint_as_float((unsigned)(best >> 32)));
  return a * 1.0001f + b * 1.0001f < R;
}

// Phase B for one query, shared by a group of L lanes (a power of two, consecutive lanes of one wave):
// lane `sub` takes the rows sub, sub + L, ... of the search box (or, without a bound, every L-th target
// point); the group's results are merged by the caller.  The row loop is the cost of a far query
// (~(2U/h + 1)^2 pi/4 rows): everything in it works in CELL UNITS on values prepared once per query.
// (returns the lane's work in instruction units -- ~45 per row, ~12 per point -- in PGP_ICP_STAMPS builds, 0 otherwise)
// box_d2 >= 0: the box is cut for THAT squared distance (every lane of a query's group must enumerate the same rows, while
// `best` may already hold a nearer point another lane found); scan_all: the query has no bound at all.
__device__ __forceinline__ unsigned nn_search(const NnGeom& g, const NnLds& t, int n_tgt, float x, float y, float z,
                                              int sub, int L, unsigned long long& best, int& bpos, float box_d2 = -1.f,
                                              bool scan_all = false) {
  if (box_d2 < 0.f ? bpos < 0 : scan_all) {
    for (int k = sub; k < n_tgt; k += L) nn_consider(x, y, z, t.pts[k], k, best, bpos);
    return 0u;
  }
  const float bound = box_d2 < 0.f ? __uint_as_float((unsigned)(best >> 32)) : box_d2;
  const NnBox b = nn_box(g, x, y, z, bound);
  const float mag = fabsf(x) + fabsf(y) + fabsf(z) + fabsf(g.ox) + fabsf(g.oy) + fabsf(g.oz);
  // query in cell units relative to the grid origin; a cell c spans [c, c + 1].  Slack of the row tests:
  // 1e-4 cell for the float cell boundaries (values <= 2^7 cells carry <= 2^-16 cell of rounding)
  // + the rounding of the coordinates themselves.
  // x in ITS cell units (the chord is cut in cells of x), y and z in theirs
  const float fx = (x - g.ox) * g.inv_hx, fy = (y - g.oy) * g.inv_h, fz = (z - g.oz) * g.inv_h;
  const float slack = 2e-4f + 2e-6f * mag * g.inv_h, slack_x = 2e-4f + 2e-6f * mag * g.inv_hx;
  const float inv_h2 = g.inv_h * g.inv_h, yz_to_x = g.h * g.inv_hx;
  const bool wide = b.x1 - b.x0 >= 2;
  const int nyb = b.y1 - b.y0 + 1;
  // (cz, cy) of this lane's first row, then steps of L rows
  int iz = sub / nyb;
  int cz = b.z0 + iz, cy = b.y0 + (sub - iz * nyb);
  const int step_z = L / nyb, step_y = L - step_z * nyb;
#ifdef PGP_ICP_STAMPS
  unsigned dbg_rows = 0, dbg_live = 0, dbg_pts = 0;
#endif
  // (round 5: the same loop one row ahead -- the next row's chord cut with the bound before this row's points, its two `start`
  // reads in flight beside this row's point reads -- timed equal on every row of tools/icp_quick.py: profiles/r05_ab/icp_lane_balance.log)
  while (cz <= b.z1) {
#ifdef PGP_ICP_STAMPS
    ++dbg_rows;
#endif
    // distance (cells) from the query to the row's (y, z) square: max(|f - (c + 1/2)| - 1/2 - slack, 0)
    const float gy = fmaxf(fabsf(fy - ((float)cy + 0.5f)) - (0.5f + slack), 0.f);
    const float gz = fmaxf(fabsf(fz - ((float)cz + 0.5f)) - (0.5f + slack), 0.f);
    const float lim = __uint_as_float((unsigned)(best >> 32)) * inv_h2 * 1.0001f;   // best d2 in cells^2, rounded up
    const float rem = lim - (gy * gy + gz * gz);
    if (rem >= 0.f) {   // else the whole row lies beyond the best distance so far
      int x0 = b.x0, x1 = b.x1;
      if (wide) {   // the chord of the ball on this row instead of the box's full width
        const float hw = __builtin_sqrtf(rem) * yz_to_x * 1.0001f + slack_x;   // half chord in cells of x
        x0 = max(x0, (int)fmaxf(fx - hw, 0.f));
        x1 = min(x1, (int)fminf(fx + hw, (float)(g.nx - 1)));
      }
      if (x0 <= x1) {
        const int row = (cz * g.ny + cy) * g.nx;
        const int kb = t.start[row + x0], ke = t.start[row + x1 + 1];
#ifdef PGP_ICP_STAMPS
        ++dbg_live;
        dbg_pts += ke - kb;
#endif
        for (int k = kb; k < ke; k += 4) {   // four LDS reads in flight; the last point of a short run counts again
          const int k1 = min(k + 1, ke - 1), k2 = min(k + 2, ke - 1), k3 = min(k + 3, ke - 1);
          const float4 m0 = t.pts[k], m1 = t.pts[k1], m2 = t.pts[k2], m3 = t.pts[k3];
          nn_consider(x, y, z, m0, k, best, bpos);
          nn_consider(x, y, z, m1, k1, best, bpos);
          nn_consider(x, y, z, m2, k2, best, bpos);
          nn_consider(x, y, z, m3, k3, best, bpos);
        }
      }
    }
    cy += step_y;
    cz += step_z;
    if (cy > b.y1) {
      cy -= nyb;
      ++cz;
    }
  }
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 2
  if (t.dbg) {
    atomicAdd(&t.dbg[0], dbg_rows);
    atomicAdd(&t.dbg[1], dbg_live);
    atomicAdd(&t.dbg[2], dbg_pts);
    atomicAdd(&t.dbg[3], 1u);
    if (sub == 0) atomicAdd(&t.dbg[4], 1u);
  }
#endif
#ifdef PGP_ICP_STAMPS
  return 45u * dbg_rows + 12u * dbg_pts;
#else
  return 0u;
#endif
}

// LDS layout of a workgroup that answers up to n_q_cap queries: [image |] d2[n_q_cap] | pos[n_q_cap] | order[n_q_cap] | bins.
// IMG_LDS = false: the image stays where it was built (HBM, L2-resident) -- targets whose image does not fit a
// CU's LDS (beyond ~6000 points); the tables are then read with global loads, everything else is the same code.
template <bool IMG_LDS>
__device__ __forceinline__ NnLds nn_load_image(const IcpArgs& a, unsigned char* smem, int n_q_cap, int tid, int nthreads) {
  NnLds t;
  size_t base = 0;
  if (IMG_LDS) {
    const uint4* src = reinterpret_cast<const uint4*>(a.nn_image);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const int n16 = (int)(a.nn.bytes >> 4);
    for (int k = tid; k < n16; k += nthreads) dst[k] = src[k];
    t.pts = reinterpret_cast<const float4*>(smem);
    t.start = reinterpret_cast<const uint16_t*>(smem + a.nn.off_start);
    t.rep = reinterpret_cast<const uint16_t*>(smem + a.nn.off_rep);
    base = a.nn.bytes;
  } else {
    t.pts = reinterpret_cast<const float4*>(a.nn_image);
    t.start = reinterpret_cast<const uint16_t*>(a.nn_image + a.nn.off_start);
    t.rep = reinterpret_cast<const uint16_t*>(a.nn_image + a.nn.off_rep);
  }
  t.d2 = reinterpret_cast<float*>(smem + base);
  t.pos = reinterpret_cast<uint16_t*>(t.d2 + n_q_cap);
  t.order = t.pos + n_q_cap;
  t.bins = reinterpret_cast<uint32_t*>(smem + ((base + 8 * (size_t)n_q_cap + 15) & ~(size_t)15));
#ifdef PGP_ICP_STAMPS
  t.dbg = nullptr;
#endif
  return t;
}
__host__ __device__ inline size_t nn_lds_bytes(uint32_t image_bytes, int n_q, bool image_in_lds = true) {
  return (((image_in_lds ? (size_t)image_bytes : 0) + 8 * (size_t)n_q + 15) & ~(size_t)15) + 2 * (size_t)kNnBins;
}
// source point q of the workgroup (q_base + q of the cloud): an L2 read.  (A copy of the workgroup's source
// points in LDS, 12 B each, measured +-1 %: the searches are bound by the LDS gather rate, not by this read.)
__device__ __forceinline__ float4 nn_src(const IcpArgs& a, const NnLds&, int q_base, int q) { return a.src[q_base + q]; }

// All n_q queries of a workgroup (source points q_base .. q_base + n_q - 1 under the pose G), in two
// phases with the work BALANCED in between: a wave pays for its dearest lane (and, in a loop nest, for
// the per-level maxima of all its lanes), and a few far queries per wave -- segmentation bleed, the rim
// of a misaligned segment -- made every wave as slow as an exhaustive scan.  Phase A bounds every query
// and files it under the log2 of its search cost (rows x cells of its box); a query of class c then gets
// 2^(c - kNnBaseClass) LANES (1 .. 64, consecutive in one wave) that share its rows and merge their
// results with cross-lane exchanges: every lane of phase B carries about the same work.
// Needs R >= ceil(n_q / NT).  In: t.pos[q] = previous correspondence (0xFFFF none).
// Out: t.d2[q], t.pos[q].  Ends with a barrier.
constexpr int kNnBaseClass = 6;   // <= 127 cost units: one lane
__device__ __forceinline__ int nn_class_lanes_log2(int c, int base = kNnBaseClass) {
  const int l = c - base;
  return l < 0 ? 0 : (l > 6 ? 6 : l);
}
constexpr int kNnFew = 64;        // up to this many unanswered queries skip the sort: 16 lanes each, one pass
#ifndef PGP_NN_ROWS
#define PGP_NN_ROWS 0             // phase B's lanes dealt by exact row counts instead of cost classes: 0 nowhere (default), 1 in the kernels with
#endif                            // several workgroups per pose, 2 everywhere.  Built, same bits, measured: profiles/r05_ab/icp_rows_dealt.log
struct NnSched {
  unsigned cnt[kNnClasses + 1];    // offset of every class in `order` (dearest class first); [kNnClasses] = n_q
  unsigned slot_end[kNnClasses];   // end of the class's lane slots
  unsigned n_slots;
  unsigned wave_sum[16];
  unsigned search_ticks;           // time of the search loop below as thread 0 saw it (100 MHz ticks)
  unsigned n_unres;                // queries the vicinity graph did not answer: phase B's population
  unsigned all_rows, max_rows;     // rows of their search boxes, summed / the largest box beyond 2048 rows (PGP_NN_ROWS)
  uint16_t few[kNnFew];            // the first of them, in arrival order: the short path of phase B
  unsigned help_s0[2];             // helping: the pass this workgroup has claimed (ping-pong across turns)
  unsigned help_avail;             // helping: some workgroup of the launch is through with its pose
  int base;                        // classes above it get 2^(class - base) lanes (kNnBaseClass, lower when lanes would idle)
};
// ---- helping: search passes of a slow pose taken by workgroups whose own pose has finished -----------------------
// With one workgroup per pose a launch lasts as long as its slowest pose (256 poses from up to 6 cm off: 240 .. 1050 us,
// mean 680: a third of the chip's time idle).  What makes a pose slow is its search -- 2000 queries still far from
// the surface, three passes of 1024 lane slots at ~20 us each -- and the passes of one iteration are independent.
// So the owner of such an iteration PUBLISHES it in HBM (the sorted slot table, the queries' bounds, the pose) and
// takes passes off a counter; a workgroup that is through with its own pose polls the counters of the others and
// takes passes too: it holds the same target image in LDS and needs nothing else.  Results come back through HBM,
// the owner waits for the passes it did not run itself.  The answer of a query does not depend on who searched
// for it (exact nearest neighbour), so results are the bits of the unhelped kernel.
//   ctl[p] = (tag << 32) | next unclaimed slot, tag = iteration + 1 while open, 0 when closed: a helper claims a pass
//   by compare-and-swap on the whole word, so a pass of iteration k + 1 is never taken with the tables of iteration k;
//   done[p] counts completed passes.  Nobody waits for anybody except the owner for passes that were CLAIMED, and a
//   claimed pass is always finished: no cycle of waits; every spin is bounded by a clock all the same (a lost pass
//   sets the lost flag and the repair launch behind the helping launch redoes the call).
struct HelpPub {                     // one per pose, in HBM (write-through stores, agent-scope loads)
  unsigned n_unres, base, pad0, pad1;
  unsigned cnt[kNnClasses + 1];
  unsigned slot_end[kNnClasses];
  float G[16];
  // followed by: order[n_src] (u32) | bound[n_src] (u64, by query) | result[n_src] (u64, by query)
};
__host__ __device__ inline size_t help_off_order() { return (sizeof(HelpPub) + 255) & ~(size_t)255; }
__host__ __device__ inline size_t help_off_bound(int n_src) { return help_off_order() + (((size_t)n_src * 4 + 255) & ~(size_t)255); }
__host__ __device__ inline size_t help_off_result(int n_src) { return help_off_bound(n_src) + (((size_t)n_src * 8 + 255) & ~(size_t)255); }
__host__ __device__ inline size_t help_bytes(int n_src) { return help_off_result(n_src) + (((size_t)n_src * 8 + 255) & ~(size_t)255); }

template <class V>
__device__ __forceinline__ void st_agent(V* p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class V>
__device__ __forceinline__ V ld_agent(const V* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ONE pass of 1024 lane slots.  tab: the slot table in LDS (the owner's own, or a helper's copy); from_pub: the
// queries' order and bounds are read from the publication (a helper) instead of this workgroup's LDS arrays (the
// owner); to_pub: the results go to the publication (a published iteration) instead of the LDS arrays (an owner's
// unpublished single pass).  ONE instance per kernel: the search it inlines is the bulk of the kernel's code.
template <int NT>
__device__ __forceinline__ void help_pass(const IcpArgs& a, const NnLds& t, const NnSched* tab, const float* G, unsigned char* pub,
                                          unsigned s0, unsigned n_slots, int tid, bool from_pub, bool to_pub) {
  const float g00 = G[0], g10 = G[1], g20 = G[2], g01 = G[4], g11 = G[5], g21 = G[6], g02 = G[8], g12 = G[9],
              g22 = G[10], g03 = G[12], g13 = G[13], g23 = G[14];
  const unsigned* p_order = reinterpret_cast<const unsigned*>(pub + help_off_order());
  const unsigned long long* p_bound = reinterpret_cast<const unsigned long long*>(pub + help_off_bound(a.n_src));
  unsigned long long* p_result = reinterpret_cast<unsigned long long*>(pub + help_off_result(a.n_src));
  const unsigned sl = s0 + (unsigned)tid;
  const bool valid = sl < n_slots;
  unsigned long long best = kNnNone;
  int bpos = -1, q = 0, lg = 0, sub = 0;
  if (valid) {
    int c = kNnClasses - 1;
    while (sl >= tab->slot_end[c]) --c;
    lg = nn_class_lanes_log2(c, tab->base);
    const unsigned first = c == kNnClasses - 1 ? 0u : tab->slot_end[c + 1];
    const unsigned rel = sl - first;
    sub = (int)(rel & ((1u << lg) - 1u));
    const unsigned at = tab->cnt[c] + (rel >> lg);
    q = !from_pub ? (int)t.order[at] : (int)ld_agent(&p_order[at]);
    const float4 s = a.src[q];
    const float x = row_xf(g00, g01, g02, g03, s.x, s.y, s.z), y = row_xf(g10, g11, g12, g13, s.x, s.y, s.z),
                z = row_xf(g20, g21, g22, g23, s.x, s.y, s.z);
    unsigned pp, d2b;
    if (!from_pub) {
      pp = t.pos[q];
      d2b = __float_as_uint(t.d2[q]);
    } else {
      const unsigned long long b = ld_agent(&p_bound[q]);
      pp = (unsigned)(b & 0xFFFFull);
      d2b = (unsigned)(b >> 32);
    }
    bpos = pp == 0xFFFFu ? -1 : (int)pp;
    if (bpos >= 0) best = ((unsigned long long)d2b << 32) | (unsigned)__float_as_int(t.pts[bpos].w);
    nn_search(a.nn, t, a.n_tgt, x, y, z, sub, 1 << lg, best, bpos);
  }
  const bool b0 = __ballot(lg >= 1) != 0ull, b1 = __ballot(lg >= 2) != 0ull, b2 = __ballot(lg >= 3) != 0ull,
             b3 = __ballot(lg >= 4) != 0ull, b4 = __ballot(lg >= 5) != 0ull, b5 = __ballot(lg >= 6) != 0ull;
  const bool need[6] = {b0, b1, b2, b3, b4, b5};
#pragma unroll
  for (int st = 0; st < 6; ++st) {
    if (!need[st]) break;   // wave-uniform
    const int off = 1 << st;
    const unsigned lo = __shfl_xor((unsigned)best, off, 64), hi = __shfl_xor((unsigned)(best >> 32), off, 64);
    const int pp = __shfl_xor(bpos, off, 64);
    const unsigned long long pk = ((unsigned long long)hi << 32) | lo;
    if (off < (1 << lg) && pk < best) {
      best = pk;
      bpos = pp;
    }
  }
  if (valid && sub == 0) {
    if (to_pub) {   // {d2 bits, position}: what the owner copies into its LDS arrays after the last pass
      st_agent(&p_result[q], ((unsigned long long)(bpos < 0 ? 0x7F7FFFFFu : (unsigned)(best >> 32)) << 32) |
                                 (unsigned long long)(bpos < 0 ? 0xFFFFu : (unsigned)bpos));
    } else {
      t.d2[q] = bpos < 0 ? FLT_MAX : __uint_as_float((unsigned)(best >> 32));
      t.pos[q] = (uint16_t)(bpos < 0 ? 0xFFFF : bpos);
    }
  }
}

// The caller may hand over a SHARE of the queries (several workgroups per pose): the cloud is dealt in blocks of
// eight consecutive queries, block k to workgroup k % P -- eight 8-byte meeting records are one 64-byte line, so a
// line of the meeting buffer is written by ONE workgroup (interleaving single queries made every 8-byte store a
// partial line: 4.6 x the algorithmic write traffic).  Local query j of part `part` is nn_share_query(j, part, P);
// t.d2 / t.pos / t.order are indexed by the query itself.
__host__ __device__ __forceinline__ int nn_share_query(int j, int part, int P) {
  return P == 1 ? j : (((j >> 3) * P + part) << 3) + (j & 7);
}
__host__ __device__ __forceinline__ int nn_share_count(int n, int part, int P) {
  if (P == 1) return n;
  const int nb = (n + 7) >> 3;                       // blocks of eight, the last one possibly short
  if (part >= nb) return 0;
  const int mine = (nb - part + P - 1) / P;
  return mine * 8 - ((nb - 1) % P == part ? nb * 8 - n : 0);
}
__host__ __device__ __forceinline__ bool nn_share_owns(int q, int part, int P) { return P == 1 || ((q >> 3) % P) == part; }

template <int NT, int R, bool HELP = false, bool ROWS = false>
__device__ __forceinline__ void nn_all_queries(const IcpArgs& a, const NnLds& t, const float* G, int q_base, int n_q,
                                               NnSched* sch /* LDS */, int tid, int part = 0, int P = 1, int help_pose = 0,
                                               unsigned help_tag = 0, int* lost = nullptr, int first_walk = 0) {
  const float g00 = G[0], g10 = G[1], g20 = G[2], g01 = G[4], g11 = G[5], g21 = G[6], g02 = G[8], g12 = G[9],
              g22 = G[10], g03 = G[12], g13 = G[13], g23 = G[14];
  static_assert(kNnBins / 2 == 4 * NT, "four counter words per thread");
  {   // zero the sort's counters
    uint4* b4 = reinterpret_cast<uint4*>(t.bins);
    b4[tid] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) {
      sch->n_unres = 0;
      sch->all_rows = 0;
      sch->max_rows = 0;
    }
  }
  __syncthreads();
#ifdef PGP_ICP_STAMPS
  unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
#define PGP_NN_STAMP(k) do { if (tid == 0 && t.dbg) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); t.dbg[k] += (unsigned)(now - st0); st0 = now; } } while (0)
#else
#define PGP_NN_STAMP(k) do { } while (0)
#endif
  // Phase A: every query is bounded -- and, next to a known candidate, ANSWERED -- on the vicinity graph
  // (nn_vic_check).  The unanswered ones are filed for phase B under
  // sort key = (15 - class) * 512 + strip: dearest class first (the lane groups of phase B need whole classes in
  // a row), and inside a class the queries of one strip -- a run of neighbouring cells of the index -- side by
  // side: the 64 lanes of a wave then walk the SAME rows of the same points, which LDS serves as broadcasts
  // instead of 64-way gathers (phase B is bound by the LDS gather rate).  One counting-sort pass over 8192
  // 16-bit counters; the order inside a bin is whatever the atomics made it.
  const int strip_shift = a.nn.strip_shift;
  constexpr unsigned kNoTag = 0xFFFFFFFFu;
  unsigned tag[R];   // key << 16 | rank inside the bin; kNoTag: not in the sort
  // Memory round trips are what phase A costs when few queries are on the lanes (a share of a pose): the R source
  // points of a thread and the graph records of their previous correspondences -- where the check of a nearly
  // converged pose happens -- are requested at once; a query whose cell representative is the better start
  // candidate (the first iteration; a correspondence gone stale) requests that record in a second round, again
  // all R together.  (A walk downhill on the graph before the check -- up to four steps -- was measured: every
  // step costs ~110 instructions per query and a dependent L2 read, more than the rows it saves: 2.45 vs 2.60 M
  // pose-iterations/s at 256 poses; one step it is.)
  const uint4* __restrict__ vic = a.nn_vic;
  float qx[R], qy[R], qz[R];
  unsigned long long best[R];
  int bpos[R];
  uint4 rec[R];
  bool live[R];
  {
    float4 sp[R];
    int pp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = r * NT + tid;
      live[r] = j < n_q;
      const int q = live[r] ? nn_share_query(j, part, P) : 0;
      sp[r] = nn_src(a, t, q_base, q);
      const unsigned v = t.pos[q];
      pp[r] = live[r] && v != 0xFFFFu ? (int)v : -1;
      if (vic) rec[r] = vic[pp[r] < 0 ? 0 : pp[r]];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      best[r] = kNnNone;
      bpos[r] = -1;
      if (live[r]) {
        const float4 s = sp[r];
        const float x = row_xf(g00, g01, g02, g03, s.x, s.y, s.z), y = row_xf(g10, g11, g12, g13, s.x, s.y, s.z),
                    z = row_xf(g20, g21, g22, g23, s.x, s.y, s.z);
        qx[r] = x;
        qy[r] = y;
        qz[r] = z;
        // candidates: the representative of the query's cell and the previous correspondence
        const int p0 = t.rep[nn_cell_of(a.nn, x, y, z)];
        nn_consider(x, y, z, t.pts[p0], p0, best[r], bpos[r]);
        if (pp[r] >= 0) nn_consider(x, y, z, t.pts[pp[r]], pp[r], best[r], bpos[r]);
        if (vic && bpos[r] >= 0 && bpos[r] != pp[r]) rec[r] = vic[bpos[r]];
      }
    }
  }
  // The FIRST iteration has no previous correspondence: the only candidate is the representative of the query's
  // cell.  There -- and only there: the walk lost when every iteration paid for it -- the candidate may first walk
  // downhill on the graph (first_walk moves, the R queries of a thread together) before the check runs at the walk's
  // end.  Measured (profiles/r04_ab/icp_first_walk_sweep.log): 0 .. 5 moves are within 1 % of each other on poses
  // that start centimetres off (the representative's bound is not what makes their first search dear: the empty
  // ball around a query 4 cm from the surface is) and one move is worth ~3 % on poses that start millimetres off.
  if (vic && first_walk > 0) {
    for (int hop = 0; hop < first_walk; ++hop) {
      bool moved[R];
      bool any = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        moved[r] = live[r] && bpos[r] >= 0 && nn_vic_move(t, rec[r], qx[r], qy[r], qz[r], best[r], bpos[r]);
        any = any || moved[r];
      }
      if (!__any(any)) break;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (moved[r]) rec[r] = vic[bpos[r]];
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    tag[r] = kNoTag;
    bool unres = false;
    unsigned q_rows = 0;
    const int q = nn_share_query(r * NT + tid, part, P);
    if (live[r]) {
      const bool proven = vic && bpos[r] >= 0 && nn_vic_check(t, rec[r], qx[r], qy[r], qz[r], best[r], bpos[r]);
      t.d2[q] = bpos[r] < 0 ? FLT_MAX : __uint_as_float((unsigned)(best[r] >> 32));
      t.pos[q] = (uint16_t)(bpos[r] < 0 ? 0xFFFF : bpos[r]);
      if (!proven) {
        unres = true;
        // no candidate at all (a non-finite or astronomically far query): the plain scan, class 15
        q_rows = ((unsigned)a.n_tgt + 63u) >> 6;
        const int c = bpos[r] < 0 ? kNnClasses - 1 : nn_class(a.nn, qx[r], qy[r], qz[r], __uint_as_float((unsigned)(best[r] >> 32)), &q_rows);
        const int cell = nn_cell_of(a.nn, qx[r], qy[r], qz[r]);
        const unsigned key = (unsigned)(kNnClasses - 1 - c) * kNnStrips + (unsigned)min(cell >> strip_shift, kNnStrips - 1);
        const unsigned old = atomicAdd(&t.bins[key >> 1], (key & 1u) ? 0x10000u : 1u);
        tag[r] = (key << 16) | ((key & 1u) ? (old >> 16) : (old & 0xFFFFu));
      }
    }
    // count the queries left for phase B (one atomic per wave and sweep) and list the first kNnFew of them
    const unsigned long long um = __ballot(unres);
    if (um) {
      unsigned base = 0;
      if ((tid & 63) == 0) base = atomicAdd(&sch->n_unres, (unsigned)__popcll(um));
      base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
      const unsigned slot = base + (unsigned)__popcll(um & ((1ull << (tid & 63)) - 1ull));
      if (unres && slot < (unsigned)kNnFew) sch->few[slot] = (uint16_t)q;
      if (ROWS) {   // the rows of the boxes (a lane's share of phase B): one add per wave and sweep
        const unsigned incl = wave_scan_incl_u32(unres ? q_rows : 0u);
        if ((tid & 63) == 63) atomicAdd(&sch->all_rows, incl);
      }
    }
  }
  __syncthreads();
  PGP_NN_STAMP(5);
  const unsigned n_unres = sch->n_unres;
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 1
  if (tid == 0 && t.dbg) t.dbg[3] += n_unres;   // queries the graph left to the search, summed over the iterations
#endif
  if (n_unres == 0) {   // every query answered on the graph (the usual iteration of a converged pose)
    if (tid == 0) sch->search_ticks = 0;
    __syncthreads();    // n_unres is zeroed by the next call: everybody has read it
    return;
  }
  if (HELP && NT == 16 * kNnFew && n_unres <= (unsigned)kNnFew) {
    // (the helping kernel keeps ONE inlined copy of the search: the handful goes through help_pass as one class of
    //  16-lane groups -- class 15 with base 11)
    if (tid < kNnClasses) {
      sch->cnt[tid] = 0;
      sch->slot_end[tid] = tid == kNnClasses - 1 ? 16u * n_unres : 16u * n_unres;
    }
    if (tid == 0) {
      sch->base = kNnClasses - 1 - 4;
      sch->search_ticks = 1;
    }
    if ((unsigned)tid < n_unres) t.order[tid] = sch->few[tid];
    __syncthreads();
    help_pass<NT>(a, t, sch, G, nullptr, 0u, 16u * n_unres, tid, false, false);
    __syncthreads();
    return;
  }
  if (!HELP && NT == 16 * kNnFew && n_unres <= (unsigned)kNnFew) {
    // A handful of stragglers (the graph answers all but ~30 of 1800 queries of a pose that is millimetres off):
    // no sort, no slot table -- query i takes lanes 16 i .. 16 i + 15, which share its rows; one pass, one barrier.
    const int grp = tid >> 4, sub = tid & 15;
    unsigned long long best = kNnNone;
    int bpos = -1, q = 0;
    const bool valid = (unsigned)grp < n_unres;
    if (valid) {
      q = sch->few[grp];
      const float4 s = nn_src(a, t, q_base, q);
      const float x = row_xf(g00, g01, g02, g03, s.x, s.y, s.z), y = row_xf(g10, g11, g12, g13, s.x, s.y, s.z),
                  z = row_xf(g20, g21, g22, g23, s.x, s.y, s.z);
      const unsigned pp = t.pos[q];
      bpos = pp == 0xFFFFu ? -1 : (int)pp;
      if (bpos >= 0) best = ((unsigned long long)__float_as_uint(t.d2[q]) << 32) | (unsigned)__float_as_int(t.pts[bpos].w);
      nn_search(a.nn, t, a.n_tgt, x, y, z, sub, 16, best, bpos);
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {   // the 16 lanes of a group are aligned: partners lane ^ off
      const unsigned lo = __shfl_xor((unsigned)best, off, 64), hi = __shfl_xor((unsigned)(best >> 32), off, 64);
      const int pp = __shfl_xor(bpos, off, 64);
      const unsigned long long pk = ((unsigned long long)hi << 32) | lo;
      if (pk < best) {
        best = pk;
        bpos = pp;
      }
    }
    if (valid && sub == 0) {
      t.d2[q] = bpos < 0 ? FLT_MAX : __uint_as_float((unsigned)(best >> 32));
      t.pos[q] = (uint16_t)(bpos < 0 ? 0xFFFF : bpos);
    }
    if (tid == 0) sch->search_ticks = 1;
    __syncthreads();
    return;
  }
  {   // exclusive scan of the 8192 counters, 8 per thread, in key order
    uint4* b4 = reinterpret_cast<uint4*>(t.bins);
    const uint4 w = b4[tid];
    const unsigned c0 = w.x & 0xFFFFu, c1 = w.x >> 16, c2 = w.y & 0xFFFFu, c3 = w.y >> 16, c4 = w.z & 0xFFFFu, c5 = w.z >> 16,
                   c6 = w.w & 0xFFFFu, c7 = w.w >> 16;
    const unsigned mine = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    const unsigned incl = wave_scan_incl_u32(mine);
    if ((tid & 63) == 63) sch->wave_sum[tid >> 6] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int wv = 0; wv < (tid >> 6); ++wv) base += sch->wave_sum[wv];
    unsigned e = base + incl - mine;   // queries in front of this thread's first bin
    // a class owns 512 bins = 64 threads = one wave (1024 threads): its first thread holds the class offset
    constexpr int kTpc = kNnStrips / 8;
    if ((tid & (kTpc - 1)) == 0) sch->cnt[kNnClasses - 1 - tid / kTpc] = e;
    uint4 o4;
    o4.x = e | ((e + c0) << 16);
    e += c0 + c1;
    o4.y = e | ((e + c2) << 16);
    e += c2 + c3;
    o4.z = e | ((e + c4) << 16);
    e += c4 + c5;
    o4.w = e | ((e + c6) << 16);
    b4[tid] = o4;
  }
  __syncthreads();
  if (tid == 0) {   // the lane slots of the classes, dearest first
    // With few queries (a share of a pose's points, a short cloud, what the graph left over) one lane per query
    // leaves most of the workgroup idle while the wave that holds the dearest class works alone (measured: 35 us
    // of a 40 us search with 625 queries on 1024 lanes): lower the class from which queries get 2, 4, ... lanes
    // for as long as everything still fits ONE pass of the workgroup.
    unsigned members[kNnClasses];   // queries per class (registers: the loops below are unrolled)
#pragma unroll
    for (int c = 0; c < kNnClasses; ++c) members[c] = (c == 0 ? n_unres : sch->cnt[c - 1]) - sch->cnt[c];
    auto count_slots = [&](int bs) {
      unsigned sl = 0;
#pragma unroll
      for (int c = 0; c < kNnClasses; ++c) sl += members[c] << nn_class_lanes_log2(c, bs);
      return sl;
    };
    int base = kNnBaseClass;
    unsigned total = count_slots(base);
    while (base > 0) {
      const unsigned wider = count_slots(base - 1);
      if (wider > (unsigned)(a.slot_budget > 0 ? a.slot_budget : NT)) break;
      --base;
      total = wider;
    }
    sch->n_slots = total;
    sch->base = base;
    if (HELP) sch->help_avail = ld_agent(a.help_finished);   // nobody idle yet: nothing is published (no overhead)
    unsigned slots = 0;
#pragma unroll
    for (int c = kNnClasses - 1; c >= 0; --c) {
      slots += members[c] << nn_class_lanes_log2(c, base);
      sch->slot_end[c] = slots;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (tag[r] != kNoTag) {
      const int q = nn_share_query(r * NT + tid, part, P);
      const unsigned key = tag[r] >> 16;
      const unsigned off = (key & 1u) ? (t.bins[key >> 1] >> 16) : (t.bins[key >> 1] & 0xFFFFu);
      t.order[off + (tag[r] & 0xFFFFu)] = (uint16_t)q;
    }
  }
  __syncthreads();
  PGP_NN_STAMP(6);
  const unsigned n_slots = sch->n_slots;
  if (HELP && n_slots > (unsigned)NT && sch->help_avail != 0u) {
    // ---- more than one pass: published, so that idle workgroups can take passes (see HelpPub)
    unsigned char* pub = a.help + (size_t)help_pose * a.help_stride;
    HelpPub* hp = reinterpret_cast<HelpPub*>(pub);
    unsigned* p_order = reinterpret_cast<unsigned*>(pub + help_off_order());
    unsigned long long* p_bound = reinterpret_cast<unsigned long long*>(pub + help_off_bound(a.n_src));
    const unsigned long long* p_result = reinterpret_cast<const unsigned long long*>(pub + help_off_result(a.n_src));
    for (unsigned i = (unsigned)tid; i < n_unres; i += NT) {
      const unsigned q = t.order[i];
      st_agent(&p_order[i], q);
      st_agent(&p_bound[q], ((unsigned long long)__float_as_uint(t.d2[q]) << 32) | (unsigned long long)t.pos[q]);
    }
    if (tid < kNnClasses) {
      st_agent(&hp->cnt[tid], sch->cnt[tid]);
      st_agent(&hp->slot_end[tid], sch->slot_end[tid]);
    }
    if (tid < 16) st_agent(&hp->G[tid], G[tid]);
    if (tid == 0) {
      st_agent(&hp->base, (unsigned)sch->base);
      st_agent(&hp->n_unres, n_unres);
      st_agent(&a.help_nslots[help_pose], n_slots);
      st_agent(&a.help_done[help_pose], 0u);
    }
    __builtin_amdgcn_s_waitcnt(0);   // everything above has left before the counter opens
    __syncthreads();
    if (tid == 0) {
      st_agent(&a.help_ctl[help_pose], (unsigned long long)help_tag << 32);
      __builtin_amdgcn_s_waitcnt(0);
    }
    const unsigned long long search_t0 = __builtin_amdgcn_s_memrealtime();
    for (int turn = 0;; ++turn) {
      if (tid == 0)
        sch->help_s0[turn & 1] = (unsigned)(__hip_atomic_fetch_add(&a.help_ctl[help_pose], (unsigned long long)NT, __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFFFFFull);
      __syncthreads();
      const unsigned s0 = sch->help_s0[turn & 1];
      if (s0 >= n_slots) break;
      help_pass<NT>(a, t, sch, G, pub, s0, n_slots, tid, false, true);
      __builtin_amdgcn_s_waitcnt(0);   // this pass's results have left
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(&a.help_done[help_pose], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {   // the passes others took: wait for them (claimed passes are always finished)
      const unsigned n_passes = (n_slots + NT - 1) / NT;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (ld_agent(&a.help_done[help_pose]) < n_passes) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > max(200000ull, 4ull * (unsigned long long)a.wait_ticks)) {   // (claimed passes are being worked on by RUNNING workgroups)
          if (lost) *lost = 1;
          break;
        }
      }
      st_agent(&a.help_ctl[help_pose], 0ull);   // closed
      sch->search_ticks = (unsigned)(__builtin_amdgcn_s_memrealtime() - search_t0);
    }
    __syncthreads();
    for (unsigned i = (unsigned)tid; i < n_unres; i += NT) {
      const unsigned q = t.order[i];
      const unsigned long long v = ld_agent(&p_result[q]);
      t.d2[q] = __uint_as_float((unsigned)(v >> 32));
      t.pos[q] = (uint16_t)(v & 0xFFFFull);
    }
    __syncthreads();
    return;
  }
  if (HELP) {   // a single pass, or nobody there to help: this workgroup's own passes, results straight into LDS
    for (unsigned s0 = 0; s0 < n_slots; s0 += NT) help_pass<NT>(a, t, sch, G, nullptr, s0, n_slots, tid, false, false);
    if (tid == 0) sch->search_ticks = 1;
    __syncthreads();
    return;
  }
  if (ROWS && !HELP && a.rows_mode != 2) {
    // ---- phase B, lanes dealt by ROWS (round 5).  The class scheme above gives a query 2^k lanes from the log2 of its box
    // cost: per-lane work then spreads over a factor of two inside a class alone, and tools/icp_lane_balance.py measured
    // the dearest lane of a wave-pass at 1.85 - 2.3 x the mean.  Here the rows of the queries' boxes are counted exactly,
    // `per` = ceil(all rows / NT) rows is a lane's share, query j gets L_j = ceil(rows_j / per) consecutive lane slots
    // (any number, in the sort's dearest-first order) and lane `sub` of the group takes its rows sub, sub + L_j, ... --
    // at most `per` of them.  The lanes of a group need not sit in one wave: their results meet in a 64-bit minimum in
    // LDS, key = d2 | original index (16 bits: an indexed target has at most 65 535 points) | position, the same order as
    // nn_consider's.  The sort's bins are free by now: keys (8 B) and slot prefixes (4 B) of up to NT queries per batch.
    // Exact nearest neighbours either way: same bits as every other schedule.
    static_assert((size_t)kNnBins * 2 >= (size_t)NT * 12, "keys + slot prefixes of a batch fit the sort's bins");
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(t.bins);
    unsigned* slotpre = reinterpret_cast<unsigned*>(t.bins) + 2 * NT;
    constexpr unsigned long long kNoKey = ((unsigned long long)0x7F7FFFFFu << 32) | 0xFFFFFFFFull;
    const unsigned long long search_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned per = max((sch->all_rows + (unsigned)NT - 1u) / (unsigned)NT, 1u);
    for (unsigned b0 = 0; b0 < n_unres; b0 += NT) {
      const unsigned nb = min((unsigned)NT, n_unres - b0);
      unsigned rows = 0;
      int q_mine = 0;
      if ((unsigned)tid < nb) {
        q_mine = t.order[b0 + tid];
        const unsigned pp = t.pos[q_mine];
        if (pp == 0xFFFFu) {   // no candidate at all: the plain scan, dealt in runs of 64 points
          rows = ((unsigned)a.n_tgt + 63u) >> 6;
          keys[tid] = kNoKey;
        } else {
          const float4 s = nn_src(a, t, q_base, q_mine);
          const float x = row_xf(g00, g01, g02, g03, s.x, s.y, s.z), y = row_xf(g10, g11, g12, g13, s.x, s.y, s.z),
                      z = row_xf(g20, g21, g22, g23, s.x, s.y, s.z);
          const float d2 = t.d2[q_mine];
          const NnBox bx = nn_box(a.nn, x, y, z, d2);
          rows = (unsigned)((bx.y1 - bx.y0 + 1) * (bx.z1 - bx.z0 + 1));
          keys[tid] = ((unsigned long long)__float_as_uint(d2) << 32) | ((unsigned long long)((unsigned)__float_as_int(t.pts[pp].w) & 0xFFFFu) << 16) |
                      (unsigned long long)pp;
        }
      }
      const unsigned lanes = rows ? (rows + per - 1u) / per : 0u;   // (all batches share one `per`: phase A summed every box)
      {   // exclusive scan of the lane counts in the sort's order
        const unsigned incl = wave_scan_incl_u32(lanes);
        if ((tid & 63) == 63) sch->wave_sum[tid >> 6] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv) base += sch->wave_sum[wv];
        slotpre[tid] = base + incl - lanes;
      }
      unsigned n_slots = 0;
      for (int wv = 0; wv < NT / 64; ++wv) n_slots += sch->wave_sum[wv];
      __syncthreads();
      for (unsigned s0 = 0; s0 < n_slots; s0 += NT) {
        const unsigned sl = s0 + (unsigned)tid;
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 6
        unsigned lane_cost = 0;
#endif
        if (sl < n_slots) {
          // the query of slot sl: the last j with slotpre[j] <= sl
          unsigned j = 0;
#pragma unroll
          for (unsigned step = (unsigned)NT / 2; step; step >>= 1) {
            const unsigned cand = j + step;
            if (cand < nb && slotpre[cand] <= sl) j = cand;
          }
          const unsigned first = slotpre[j], L = (j + 1 < nb ? slotpre[j + 1] : n_slots) - first;
          const int q = t.order[b0 + j];
          const float4 s = nn_src(a, t, q_base, q);
          const unsigned long long key = keys[j];
          const float x = row_xf(g00, g01, g02, g03, s.x, s.y, s.z), y = row_xf(g10, g11, g12, g13, s.x, s.y, s.z),
                      z = row_xf(g20, g21, g22, g23, s.x, s.y, s.z);
          const bool scan_all = t.pos[q] == 0xFFFFu;
          unsigned long long best = kNnNone;
          int bpos = -1;
          if ((key & 0xFFFFull) != 0xFFFFull) {
            best = (key & 0xFFFFFFFF00000000ull) | ((key >> 16) & 0xFFFFull);
            bpos = (int)(key & 0xFFFFull);
          }
          if (scan_all) {   // run `sub + k L` of 64 points
            for (unsigned run = sl - first; run * 64u < (unsigned)a.n_tgt; run += L) {
              const int k_end = min((int)(run * 64u) + 64, a.n_tgt);
              for (int k = (int)(run * 64u); k < k_end; ++k) nn_consider(x, y, z, t.pts[k], k, best, bpos);
            }
          } else {
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 6
            lane_cost =
#endif
            nn_search(a.nn, t, a.n_tgt, x, y, z, (int)(sl - first), (int)L, best, bpos, t.d2[q], false);
          }
          if (bpos >= 0) {
            const unsigned long long nk = (best & 0xFFFFFFFF00000000ull) | ((best & 0xFFFFull) << 16) | (unsigned long long)(unsigned)bpos;
            if (nk < key) atomicMin(&keys[j], nk);
          }
        }
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 6
        {   // how well a wave's lanes are balanced (tools/icp_lane_balance.py)
          unsigned mx = lane_cost, sm = lane_cost;
          for (int off = 32; off >= 1; off >>= 1) {
            mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
            sm += (unsigned)__shfl_xor((int)sm, off, 64);
          }
          if ((tid & 63) == 0 && t.dbg && sm > 0) {
            atomicAdd(&t.dbg[0], mx);
            atomicAdd(&t.dbg[1], sm / 64u);
            atomicAdd(&t.dbg[2], 1u);
          }
        }
#endif
      }
      __syncthreads();
      if ((unsigned)tid < nb) {
        const unsigned long long k = keys[tid];
        const unsigned pp = (unsigned)(k & 0xFFFFull);
        t.d2[q_mine] = pp == 0xFFFFu ? FLT_MAX : __uint_as_float((unsigned)(k >> 32));
        t.pos[q_mine] = (uint16_t)pp;
      }
      __syncthreads();   // keys and slot prefixes belong to the next batch (and, after the last, to the next sort)
    }
    if (tid == 0) sch->search_ticks = max((unsigned)(__builtin_amdgcn_s_memrealtime() - search_t0), 1u);
    return;
  }
  const int lane_base = sch->base;
  int c = kNnClasses - 1;   // class of the current slot: slots only grow
  // a slot's query and its source point (an L2 read behind an LDS read) are fetched one trip ahead
  int nq = 0, nlg = 0, nsub = 0;
  float4 ns = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](unsigned sl) {
    if (sl < n_slots) {
      while (sl >= sch->slot_end[c]) --c;
      nlg = nn_class_lanes_log2(c, lane_base);
      const unsigned first = c == kNnClasses - 1 ? 0u : sch->slot_end[c + 1];
      const unsigned rel = sl - first;
      nsub = (int)(rel & ((1u << nlg) - 1u));
      nq = t.order[sch->cnt[c] + (rel >> nlg)];
      ns = nn_src(a, t, q_base, nq);
    }
  };
  fetch((unsigned)tid);
  const unsigned long long search_t0 = __builtin_amdgcn_s_memrealtime();
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 3
  const unsigned long long wv0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (unsigned s0 = 0; s0 < n_slots; s0 += NT) {   // uniform trip count: the exchanges below need whole waves
    const unsigned sl = s0 + (unsigned)tid;
    const bool valid = sl < n_slots;
    unsigned long long best = kNnNone;
    int bpos = -1;
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 6
    unsigned lane_cost = 0;
#endif
    const int q = nq, lg = valid ? nlg : 0, sub = nsub;
    const float4 s = ns;
    fetch(sl + NT);
    if (valid) {
      const float x = row_xf(g00, g01, g02, g03, s.x, s.y, s.z), y = row_xf(g10, g11, g12, g13, s.x, s.y, s.z),
                  z = row_xf(g20, g21, g22, g23, s.x, s.y, s.z);
      const unsigned pp = t.pos[q];
      bpos = pp == 0xFFFFu ? -1 : (int)pp;
      if (bpos >= 0) best = ((unsigned long long)__float_as_uint(t.d2[q]) << 32) | (unsigned)__float_as_int(t.pts[bpos].w);
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 6
      lane_cost = nn_search(a.nn, t, a.n_tgt, x, y, z, sub, 1 << lg, best, bpos);
#else
      nn_search(a.nn, t, a.n_tgt, x, y, z, sub, 1 << lg, best, bpos);
#endif
    }
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 6
    {   // how well a wave's lanes are balanced: the wave pays its dearest lane, 64 times
      unsigned mx = lane_cost, sm = lane_cost;
      for (int off = 32; off >= 1; off >>= 1) {
        mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
        sm += (unsigned)__shfl_xor((int)sm, off, 64);
      }
      if ((tid & 63) == 0 && t.dbg && sm > 0) {
        atomicAdd(&t.dbg[0], mx);          // critical path of the wave-pass (instruction units)
        atomicAdd(&t.dbg[1], sm / 64u);    // what a perfectly balanced wave would pay
        atomicAdd(&t.dbg[2], 1u);
      }
      lane_cost = 0;
    }
#endif
    // merge the lanes of a group: class regions start at multiples of their group size, so the partners
    // lane ^ off (off < L) work on the same query
    // (most waves hold single-lane queries only: no exchange at all)
    const bool b0 = __ballot(lg >= 1) != 0ull, b1 = __ballot(lg >= 2) != 0ull, b2 = __ballot(lg >= 3) != 0ull,
               b3 = __ballot(lg >= 4) != 0ull, b4 = __ballot(lg >= 5) != 0ull, b5 = __ballot(lg >= 6) != 0ull;
    const bool need[6] = {b0, b1, b2, b3, b4, b5};
#pragma unroll
    for (int st = 0; st < 6; ++st) {
      if (!need[st]) break;   // wave-uniform
      const int off = 1 << st;
      const unsigned lo = __shfl_xor((unsigned)best, off, 64), hi = __shfl_xor((unsigned)(best >> 32), off, 64);
      const int pp = __shfl_xor(bpos, off, 64);
      const unsigned long long pk = ((unsigned long long)hi << 32) | lo;
      if (off < (1 << lg) && pk < best) {
        best = pk;
        bpos = pp;
      }
    }
    if (valid && sub == 0) {
      t.d2[q] = bpos < 0 ? FLT_MAX : __uint_as_float((unsigned)(best >> 32));
      t.pos[q] = (uint16_t)(bpos < 0 ? 0xFFFF : bpos);
    }
  }
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 3
  if ((tid & 63) == 0 && t.dbg) {   // per-wave time in the search loop: sum over waves, maximum, wave 0's
    const unsigned dtw = (unsigned)(__builtin_amdgcn_s_memrealtime() - wv0);
    atomicAdd(&t.dbg[0], dtw);
    atomicMax(&t.dbg[1], dtw);
    if (tid == 0) t.dbg[2] += dtw;
    if (tid == NT - 64) t.dbg[3] += dtw;
  }
#endif
  __syncthreads();
  if (tid == 0) sch->search_ticks = (unsigned)(__builtin_amdgcn_s_memrealtime() - search_t0);
  PGP_NN_STAMP(7);
#undef PGP_NN_STAMP
}

// Split-path correspondences through the index: grid (source chunks of 1024, poses); the workgroup
// copies the image into LDS and answers its 1024 queries.  Same keys as icp_nn_split.
constexpr int kIdxThreads = kIcpThreads;   // (shares the sort of nn_all_queries, whose bins follow the thread count)
template <bool IMG_LDS>
__global__ __launch_bounds__(kIdxThreads) void icp_nn_index(IcpArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ NnSched s_sch;
  const int pose = blockIdx.y;
  if (a.st_done[pose]) return;
  const int tid = threadIdx.x;
  const int q_base = blockIdx.x * kIdxThreads, n_q = min(kIdxThreads, a.n_src - q_base);
  const NnLds t = nn_load_image<IMG_LDS>(a, smem, kIdxThreads, tid, kIdxThreads);
  const size_t o = (size_t)pose * a.n_src + q_base + tid;
  if (tid < n_q) {
    const int pp = a.ws_pos[o];
    t.pos[tid] = (uint16_t)(pp < 0 ? 0xFFFF : pp);
  }
  __syncthreads();
  nn_all_queries<kIdxThreads, 1>(a, t, a.T + 16 * (size_t)pose, q_base, n_q, &s_sch, tid);
  if (tid < n_q) {
    const unsigned pp = t.pos[tid];
    a.ws_pos[o] = pp == 0xFFFFu ? -1 : (int)pp;
    a.ws_key[o] = pp == 0xFFFFu ? ~0ull
                                : (((unsigned long long)__float_as_uint(t.d2[tid]) << 32) | (unsigned)__float_as_int(t.pts[pp].w));
  }
}

// ONE WORKGROUP PER POSE, every iteration inside the launch, correspondences through the index in LDS:
// the image is copied once; squared distances and correspondences of the pose's source points live in
// LDS from search to selection to sums, so an iteration reads only the source cloud (L2) from memory
// (the target points of the sums are read from the image).  Selection, sums, their reduction tree,
// the closed-form update and the stop rules are those of icp_refine, operation for operation: the two
// kernels (and the exhaustive searches) give bit-identical transforms, energies and iteration counts.
constexpr int kPiR = 4 * kPS;   // source points per thread: n_src <= 4096
constexpr int kSelRank = 512;   // keys of the threshold's 12-bit bin that are ranked by comparison (else: 8-bit radix passes)
// CLUSTER = false: one workgroup per pose, the meeting code (and its arguments) compiled out -- the kernel is at its
// 128-register ceiling, and every value less to keep is a spill less per iteration.
// TRIM_ONLY = true: the TrimmedICP form (UCTState.cpp:137-139: trimming, energy ratio, no correspondence cap and none
// of the extra stop rules) -- the hot form; its branches on the other forms' options are gone at compile time.
// PIR: source points per thread (n_src <= PIR x 1024): every per-thread array of the search's phase A, of the
// selection and of the sums has PIR entries -- a 1756-point segment takes PIR = 2 and half the registers of PIR = 4.
// A workgroup of a clustered launch leaves its pose (its share published and the pose handed to workgroup 0; or the pose
// finished): the LAST of the pose's workgroups to leave puts the pose's arrival counter back to zero -- nobody polls it any
// more -- so the next clustered launch on this context starts without a memset in front of it (a fill is ~3 us of GPU time
// and ~9 us of dependency latency on a call that lasts 200).
__device__ __forceinline__ void cluster_leave(const IcpArgs& a, int pose, int tid) {
  if (tid != 0) return;
  const unsigned left = __hip_atomic_fetch_add(&a.x_done[pose], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (left + 1u == (unsigned)a.wgs_per_pose) {
    __hip_atomic_store(&a.x_ctr[pose], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.x_abandon[pose], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.x_done[pose], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int METRIC, bool IMG_LDS, bool CLUSTER, bool TRIM_ONLY, int PIR, bool HELP = false>
__device__ __forceinline__ void icp_persist_body(const IcpArgs a) {   // BY VALUE: a reference to the kernel argument costs 16 -> 83 spilled registers
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double s_red[(kIcpThreads / 64) * (kRedPlane + 1)];
  __shared__ float s_G[16];
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_scan[kIcpThreads / 64];
  __shared__ unsigned s_sel_prefix[2], s_sel_kleft[2], s_sel_nties;
  __shared__ unsigned s_sel_bin, s_sel_m, s_sel_cnt, s_sel_thr, s_sel_take;
  __shared__ unsigned s_tie[64];
  static_assert(PIR >= 1 && PIR * (kIcpThreads / 64) <= 64, "one wave scans the tie counts");
  __shared__ double s_energy, s_energy_old;
  __shared__ int s_continue;
  __shared__ double s_sum[kRedPlane + 1];
  __shared__ float s_G_old[16];
  __shared__ NnSched s_sch;

  // the repair launch behind a helping launch (launch_icp, PGP_ICP_HELP): nothing to do unless a pass was lost
  if (!CLUSTER && a.run_if && *a.run_if == 0u) return;   // (the helping launch's repair launch: one workgroup per pose)

  // several workgroups per pose (few poses in flight: 64 poses would use 64 of the 256 CUs): workgroup `part`
  // searches its share of the source points (nn_share_query); the shares meet in HBM (x_buf) once per iteration,
  // everything after the search runs in every workgroup of the pose on the same data -- same bits, same decisions
  // Sharing pays while the search is long (poses centimetres off: 60 -> 24 us per iteration); once every share is
  // searched in ~10 us the meeting (~10 us) costs more than it saves, so the pose goes on in workgroup 0 alone
  // and the others leave.  The switch follows measured time -- the results do not depend on it.
  int P = CLUSTER ? a.wgs_per_pose : 1;
  // block b = part * n + pose: workgroup 0 of every pose (the one that may finish the pose alone) comes from the
  // first n blocks, which the dispatcher deals round-robin over the 8 XCDs (b = pose * P + part put all of them
  // on XCDs 0 and 4: +10 % on a call whose poses soon go solo)
  const int part = CLUSTER ? (int)blockIdx.x / a.n : 0, pose = (int)blockIdx.x - part * a.n;
  int n_share = nn_share_count(a.n_src, part, P);
  __shared__ int s_lost, s_solo;
  __shared__ unsigned s_slowest;   // (thread 0) the slowest share / longest wait of the pose's meetings so far, in ticks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Tg = a.T + 16 * (size_t)pose;
  if (tid == 0) {
    s_lost = s_solo = 0;
    s_slowest = 0u;
  }
#if defined(PGP_ICP_STAMPS)
  const unsigned long long k_start = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef PGP_ICP_STAMPS
  __shared__ unsigned s_dbg[8];
  if (tid < 8) s_dbg[tid] = 0;
  NnLds t = nn_load_image<IMG_LDS>(a, smem, a.n_src, tid, kIcpThreads);
  t.dbg = s_dbg;
#else
  const NnLds t = nn_load_image<IMG_LDS>(a, smem, a.n_src, tid, kIcpThreads);
#endif
  for (int q = tid; q < a.n_src; q += kIcpThreads) t.pos[q] = 0xFFFF;   // no previous correspondence yet
  if (tid < 16) {
    const float v = (!CLUSTER && a.T_in ? a.T_in : a.T)[16 * (size_t)pose + tid];
    s_G[tid] = v;
    if ((CLUSTER || HELP) && a.T_save && part == 0) a.T_save[16 * (size_t)pose + tid] = v;   // what a repair launch starts from
  }
  if (tid == 0) {
    s_energy_old = (double)FLT_MAX;   // PCL: energy starts at numeric_limits<float>::max()
    s_energy = 0.0;
  }
  __syncthreads();

  int it = 0;
#ifdef PGP_ICP_STAMPS   // diagnostic build only (tools/icp_phases.py): thread 0 of pose 0 times the phases (cycles, summed
  // over the iterations) and reports them in energy[1..5] (poses 1..7 leave their energies alone)
  unsigned long long st_prev = 0, st_acc[6] = {0, 0, 0, 0, 0, 0};
#define PGP_STAMP(k) do { if (pose == a.dbg_pose && tid == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); \
    if ((k) > 0) st_acc[k] += now - st_prev; st_prev = now; } } while (0)
#else
#define PGP_STAMP(k) do { } while (0)
#endif
  for (;;) {
    PGP_STAMP(0);
    // ---- 1. correspondences ---------------------------------------------------------------------
    nn_all_queries<kIcpThreads, PIR, HELP, PGP_NN_ROWS == 2 || (PGP_NN_ROWS == 1 && CLUSTER)>(a, t, s_G, 0, n_share, &s_sch, tid, part, P, pose, (unsigned)(it + 1), &s_lost,
                                            it == 0 ? a.first_walk : 0);
    if (HELP && s_lost) break;
    if (CLUSTER && P > 1) {
      // publish this share (write-through, agent scope: the partners may sit on other XCDs), meet, read theirs.
      // Consecutive lanes store consecutive records of a block of eight: whole 64-byte lines.
      unsigned long long* xb = a.x_buf + ((size_t)(it & 1) * a.n + pose) * a.n_src;
      for (int j = tid; j < n_share; j += kIcpThreads) {
        const int q = nn_share_query(j, part, P);
        __hip_atomic_store(&xb[q], ((unsigned long long)__float_as_uint(t.d2[q]) << 32) | (unsigned long long)t.pos[q],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      unsigned* xt = a.x_ticks + ((size_t)(it & 1) * a.n + pose) * 4;
      if (tid == 0) __hip_atomic_store(&xt[part], s_sch.search_ticks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);   // the stores have left before the arrival below is counted
      __syncthreads();
      if (tid == 0) {
        const unsigned target = (unsigned)P * (unsigned)(it + 1);
        // relaxed on both sides: the shares are written through and waited for above and are read with
        // agent-scope loads below, so no cache has to be written back or invalidated here (an acquire in the
        // polling loop invalidated the XCD's L2 for every workgroup on it: selection + sums 10.7 -> 13.4 us)
        __hip_atomic_fetch_add(&a.x_ctr[pose], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The grid fits the device (launch_resident: one workgroup per CU), so the partners arrive -- at once on an idle
        // device, as compute units come free behind somebody else's kernel; the clock bound turns a device that stays
        // taken for seconds into a pose finished by one workgroup instead of a hang.
        // The bound follows the WORK: 64 x the slowest share of the previous meeting, a few
        // milliseconds at least (a.wait_ticks) -- not the 2 s of round 5, which two processes holding half the device each
        // could run into meeting after meeting.  And whoever runs out of it says so in the pose's ABANDON word, which every
        // waiting partner polls: the late ones leave at once instead of spinning out a bound of their own.
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long bound = max((unsigned long long)a.wait_ticks, 64ull * (unsigned long long)s_slowest);
        while (__hip_atomic_load(&a.x_ctr[pose], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (__hip_atomic_load(&a.x_abandon[pose], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            s_lost = 1;
            break;
          }
          if (__builtin_amdgcn_s_memrealtime() - t0 > bound) {
            __hip_atomic_store(&a.x_abandon[pose], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_lost = 1;
            break;
          }
        }
        if (a.force_lost && it == 0) s_lost = 1;
        unsigned slowest = 0;
        for (int k = 0; k < P; ++k)
          slowest = max(slowest, __hip_atomic_load(&xt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (!s_lost) s_slowest = max(slowest, (unsigned)(__builtin_amdgcn_s_memrealtime() - t0));
        s_solo = slowest < a.solo_ticks ? 1 : 0;   // the same P numbers in every workgroup: the same decision
      }
      __syncthreads();
      if (s_lost) {
        // A partner did not arrive in time (another process holding the chip's compute units for seconds): workgroup 0
        // goes on ALONE from this very iteration -- it holds every query's previous correspondence (each meeting hands
        // all shares to everybody) and searches all queries again under the same transform --, the others leave.  Exact
        // nearest neighbours whoever searches: the bits of the undisturbed launch.  (A workgroup that comes late finds
        // the counter short of its next target, times out in its turn and leaves, or -- workgroup 0 -- goes on alone.)
        if (part != 0) {
          cluster_leave(a, pose, tid);
          return;
        }
        __syncthreads();
        if (tid == 0) s_lost = 0;
        P = 1;
        n_share = a.n_src;
        continue;   // (the loop's head searches again: the iteration count has not moved)
      }
      if (s_solo && part != 0) {         // this share is published; workgroup 0 finishes the pose
        cluster_leave(a, pose, tid);
        return;
      }
      for (int q = tid; q < a.n_src; q += kIcpThreads) {
        if (nn_share_owns(q, part, P)) continue;
        const unsigned long long v = __hip_atomic_load(&xb[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t.d2[q] = __uint_as_float((unsigned)(v >> 32));
        t.pos[q] = (uint16_t)(v & 0xFFFFull);
      }
      __syncthreads();
      if (s_solo) {   // from the next iteration on: every query here, nobody to meet
        P = 1;
        n_share = a.n_src;
      }
    }
    PGP_STAMP(1);

    // the source points of the sums (step 3): requested now, so that their L2 round trip passes under the selection
    // (thread t sums the points 4 t .. 4 t + 3, as icp_refine does: see there)
    float4 sreg[kSumR];
#pragma unroll
    for (int r = 0; r < kSumR; ++r) sreg[r] = a.src[min(kSumR * tid + r, a.n_src - 1)];

    // ---- 2. selection threshold: the k-th smallest d2 (the keys are the float bits: d2 >= +0) ----------------
    unsigned thr_key = 0xFFFFFFFFu, ties_to_take = 0xFFFFFFFFu;  // default: take everything
    if ((TRIM_ONLY || a.max_corr2 < 0.f) && a.k_trim < a.n_src) {
      // One histogram over the upper 12 key bits (sign, exponent, four mantissa bits: 4096 counters in the
      // search's sort bins, free by now) finds the bin of the k-th key; the keys of that bin -- a few dozen of
      // 2500 -- are compacted and RANKED by comparison, each by its own thread.  Six barriers instead of the
      // thirteen of four 8-bit radix passes; same threshold, same tie count.  A bin with more than kSelRank
      // keys (a cloud of coincident points: hundreds of d2 = 0) takes the radix passes below.
      uint32_t* hist = t.bins;
      reinterpret_cast<uint4*>(hist)[tid] = make_uint4(0u, 0u, 0u, 0u);
      if (tid == 0) s_sel_cnt = 0;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < PIR; ++r) {
        const int i = r * kIcpThreads + tid;
        if (i < a.n_src) atomicAdd(&hist[__float_as_uint(t.d2[i]) >> 19], 1u);
      }
      __syncthreads();
      const uint4 hv = reinterpret_cast<const uint4*>(hist)[tid];
      const unsigned mine = hv.x + hv.y + hv.z + hv.w;
      const unsigned incl = wave_scan_incl_u32(mine);
      if (lane == 63) s_scan[wave] = incl;
      __syncthreads();
      {
        unsigned woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_scan[w];
        unsigned excl = woff + incl - mine;
        const unsigned kk = (unsigned)a.k_trim;
        if (excl < kk && kk <= excl + mine) {   // exactly one thread: its four bins hold the k-th key
          unsigned b = 4u * (unsigned)tid, m = hv.x;
          if (kk > excl + m) { excl += m; m = hv.y; ++b;
            if (kk > excl + m) { excl += m; m = hv.z; ++b;
              if (kk > excl + m) { excl += m; m = hv.w; ++b; } } }
          s_sel_bin = b;
          s_sel_m = m;
          s_sel_kleft[0] = kk - excl;   // rank of the key inside its bin (1-based)
        }
      }
      __syncthreads();
      const unsigned sel_bin = s_sel_bin, sel_m = s_sel_m, kleft = s_sel_kleft[0];
      if (sel_m <= (unsigned)kSelRank) {
        uint32_t* cand = hist;   // the histogram is dead: every thread read its counters before the last barrier
#pragma unroll
        for (int r = 0; r < PIR; ++r) {
          const int i = r * kIcpThreads + tid;
          if (i < a.n_src) {
            const unsigned key = __float_as_uint(t.d2[i]);
            if ((key >> 19) == sel_bin) cand[atomicAdd(&s_sel_cnt, 1u)] = key;
          }
        }
        __syncthreads();
        if ((unsigned)tid < sel_m) {
          const unsigned c = cand[tid];
          unsigned lt = 0, le = 0;
          for (unsigned j = 0; j < sel_m; ++j) {   // broadcast reads
            const unsigned v = cand[j];
            lt += v < c ? 1u : 0u;
            le += v <= c ? 1u : 0u;
          }
          if (lt < kleft && kleft <= le) {   // c is the k-th key (every thread that holds a copy of it writes the same)
            s_sel_thr = c;
            s_sel_take = kleft - lt;         // how many keys equal to it belong to the k smallest
            s_sel_nties = le - lt;           // the keys EQUAL to the threshold
          }
        }
        __syncthreads();
        thr_key = s_sel_thr;
        ties_to_take = s_sel_take;
      } else {
        // Three barriers per pass: count | scan inside the waves | pick the bin.  The thread that owns the bin
        // writes the next pass's prefix and rank itself (ping-pong slots: its neighbours still read this pass's),
        // and every bin is zeroed by its owner on the way out.
        if (tid == 0) {
          s_sel_prefix[0] = 0;
          s_sel_kleft[0] = (unsigned)a.k_trim;
        }
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        for (int pass = 0; pass < 4; ++pass) {
          const int shift = 24 - 8 * pass;
          const unsigned prefix = s_sel_prefix[pass & 1], kl = s_sel_kleft[pass & 1];
          const unsigned mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
          for (int i = tid; i < a.n_src; i += kIcpThreads) {
            const unsigned key = __float_as_uint(t.d2[i]);
            if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
          }
          __syncthreads();
          unsigned h8 = 0, inc8 = 0;
          if (tid < 256) {
            h8 = s_hist[tid];
            inc8 = h8;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
              const unsigned tt = __shfl_up(inc8, off, 64);
              if (lane >= off) inc8 += tt;
            }
            if (lane == 63) s_scan[wave] = inc8;
          }
          __syncthreads();
          if (tid < 256) {
            unsigned woff = 0;
            for (int w = 0; w < wave; ++w) woff += s_scan[w];
            inc8 += woff;
            const unsigned excl = inc8 - h8;
            if (excl < kl && kl <= inc8) {          // the bin that holds the kl-th key: exactly one thread
              s_sel_kleft[(pass + 1) & 1] = kl - excl;
              s_sel_prefix[(pass + 1) & 1] = prefix | ((unsigned)tid << shift);
              s_sel_nties = h8;                           // after the last pass: the keys EQUAL to the threshold
            } else if (tid == 255 && kl > inc8) {      // fewer keys than the rank asks for (cannot happen: k <= n)
              s_sel_kleft[(pass + 1) & 1] = kl - inc8;
              s_sel_prefix[(pass + 1) & 1] = prefix | (255u << shift);
              s_sel_nties = 0xFFFFFFFFu;
            }
            s_hist[tid] = 0;
          }
          __syncthreads();
        }
        thr_key = s_sel_prefix[0];
        ties_to_take = s_sel_kleft[0];
      }
    }

    PGP_STAMP(2);
    // ---- 3. f64 sums over the selected pairs (ordered tie handling), fixed-tree reduction ----
    constexpr int kNs = METRIC == 1 ? kRedPlane : 16;   // sums in use
    double acc[kNs];
#pragma unroll
    for (int k = 0; k < kNs; ++k) acc[k] = 0.0;
    double e_acc = 0.0;
    // ties at the threshold are taken in index order (as icp_refine does): exclusive scan of the per-thread
    // tie counts -- inside the waves on the DPP path, across them through LDS
    const bool ranked = (TRIM_ONLY || a.max_corr2 < 0.f) && thr_key != 0xFFFFFFFFu;
    // the usual case: every key equal to the threshold is taken (one such key, the k-th itself) -- no ranking
    const bool all_ties = ranked && ties_to_take >= s_sel_nties;
    const bool wave_has_points = kSumR * 64 * wave < a.n_src;   // this wave's 256 consecutive points exist
    unsigned rank = 0;
    if (ranked && !all_ties) {
      unsigned mine = 0;
#pragma unroll
      for (int r = 0; r < kSumR; ++r) {
        const int i = kSumR * tid + r;
        mine += (i < a.n_src && __float_as_uint(t.d2[i]) == thr_key) ? 1u : 0u;
      }
      const unsigned incl = wave_scan_incl_u32(mine);
      if (lane == 63) s_tie[wave] = incl;
      __syncthreads();
      rank = incl - mine;
      for (int w = 0; w < wave; ++w) rank += s_tie[w];
    }
    if (wave_has_points) {
#pragma unroll
    for (int r = 0; r < kSumR; ++r) {
      const int i = kSumR * tid + r;
      unsigned key = 0xFFFFFFFFu;
      float d2 = 0.f;
      if (i < a.n_src) {
        d2 = t.d2[i];
        key = __float_as_uint(d2);
      }
      bool sel;
      if (!TRIM_ONLY && a.max_corr2 >= 0.f) {
        sel = i < a.n_src && d2 <= a.max_corr2;
      } else if (!ranked) {
        sel = i < a.n_src;
      } else {
        const bool tie = i < a.n_src && key == thr_key;
        sel = i < a.n_src && (key < thr_key || (tie && (all_ties || rank < ties_to_take)));
        rank += tie ? 1u : 0u;
      }
      const unsigned pm = i < a.n_src ? (unsigned)t.pos[i] : 0xFFFFu;   // 0xFFFF: a non-finite point has no neighbour
      if constexpr (METRIC == 1) {
        if (sel && pm != 0xFFFFu) {
          const float4 s = sreg[r];
          const float4 m = t.pts[pm];
          const float4 nn = a.tgt_n[__float_as_int(m.w)];
          const double px = row_xf(s_G[0], s_G[4], s_G[8], s_G[12], s.x, s.y, s.z);
          const double py = row_xf(s_G[1], s_G[5], s_G[9], s_G[13], s.x, s.y, s.z);
          const double pz = row_xf(s_G[2], s_G[6], s_G[10], s_G[14], s.x, s.y, s.z);
          const double nx = nn.x, ny = nn.y, nz = nn.z;
          const double row[6] = {nz * py - ny * pz, nx * pz - nz * px, ny * px - nx * py, nx, ny, nz};
          const double rhs = nx * (double)m.x + ny * (double)m.y + nz * (double)m.z - nx * px - ny * py - nz * pz;
          acc[0] += 1.0;
          int tt = 1;
#pragma unroll
          for (int rr = 0; rr < 6; ++rr)
#pragma unroll
            for (int c = rr; c < 6; ++c) acc[tt++] += row[rr] * row[c];
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) acc[22 + rr] += row[rr] * rhs;
          e_acc += (double)d2;
        }
      } else if (sel && pm != 0xFFFFu) {
        const float4 s = sreg[r];
        const float4 m = t.pts[pm];
        const float s0 = s.x, s1 = s.y, s2 = s.z;
        acc[0] += 1.0;
        acc[1] += s0; acc[2] += s1; acc[3] += s2;
        acc[4] += m.x; acc[5] += m.y; acc[6] += m.z;
        acc[7] += (double)s0 * m.x; acc[8] += (double)s0 * m.y; acc[9] += (double)s0 * m.z;
        acc[10] += (double)s1 * m.x; acc[11] += (double)s1 * m.y; acc[12] += (double)s1 * m.z;
        acc[13] += (double)s2 * m.x; acc[14] += (double)s2 * m.y; acc[15] += (double)s2 * m.z;
        e_acc += (double)d2;
      }
    }
    // (a wave without points holds zeros: its trees would return the same zeros)
#pragma unroll
    for (int k = 0; k < kNs; ++k)
      acc[k] = wave_sum_f64(acc[k]);
    e_acc = wave_sum_f64(e_acc);
    }
    // (no barrier here: s_red was last read before the previous iteration's closing barriers)
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < kRedPlane; ++k) s_red[wave * (kRedPlane + 1) + k] = k < kNs ? acc[k < kNs ? k : 0] : 0.0;
      s_red[wave * (kRedPlane + 1) + kRedPlane] = e_acc;
    }
    __syncthreads();
    if (tid <= kRedPlane) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < kIcpThreads / 64; ++w) v += s_red[w * (kRedPlane + 1) + tid];
      s_sum[tid] = v;
    }
    __syncthreads();
    PGP_STAMP(3);
    if (tid == 0) {
      const double* red = s_sum;
      const double E = red[0] >= 1.0 ? red[kRedPlane] / red[0] : 0.0;
      for (int k = 0; k < 16; ++k) s_G_old[k] = s_G[k];
      if constexpr (METRIC == 1) solve_plane(red, s_G);
      else solve_rigid(red, s_G);
      PGP_STAMP(4);
      const double E_old = s_energy_old;
      s_energy = E;
      s_energy_old = E;
      bool go = it + 1 < a.max_iter;
      if (a.ratio > 0.f && !(E / E_old < (double)a.ratio)) go = false;
      if (red[0] < 1.0) go = false;
      // (the TrimmedICP form has none of the extra rules: no call, no spills around it)
      if (!TRIM_ONLY && (a.t_eps >= 0.f || a.rel_mse > 0.f || a.abs_mse >= 0.f || a.smooth > 0) &&
          converged_extra(stop_rules_of(a), pose, it + 1, s_G_old, s_G, E, E_old))
        go = false;
      s_continue = go ? 1 : 0;
      PGP_STAMP(5);
    }
    __syncthreads();
    ++it;
    if (!s_continue) break;
  }
  if (HELP) {
    if (!s_lost) {
      if (tid < 16) Tg[tid] = s_G[tid];
      if (tid == 0) {
        if (a.energy) a.energy[pose] = (float)s_energy;
        if (a.iters) a.iters[pose] = it;
      }
    } else if (tid == 0) {
      __hip_atomic_store(a.x_lost, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- this pose is through: take search passes of the poses that are still running (see HelpPub)
    __shared__ int h_pose;
    __shared__ unsigned h_s0, h_ns;
    __shared__ unsigned long long h_pick[kIcpThreads / 64];
    if (tid == 0) {
      __builtin_amdgcn_s_waitcnt(0);
      __hip_atomic_fetch_add(a.help_finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long h_t0 = __builtin_amdgcn_s_memrealtime();   // (thread 0 reads it) the last time this helper had something to do
    for (;;) {
      // every thread looks at one pose's counter; the lowest open one is tried
      unsigned long long c = 0;
      unsigned ns = 0;
      bool open = false;
      if (tid < a.n && tid != pose) {
        c = ld_agent(&a.help_ctl[tid]);
        if ((c >> 32) != 0ull) {
          ns = ld_agent(&a.help_nslots[tid]);
          open = (unsigned)(c & 0xFFFFFFFFull) < ns;
        }
      }
      const unsigned long long bm = __ballot(open);
      if (lane == 0) h_pick[wave] = bm;
      __syncthreads();
      int cand = -1;
      for (int w = 0; w < kIcpThreads / 64 && cand < 0; ++w)
        if (h_pick[w]) cand = 64 * w + __builtin_ctzll(h_pick[w]);
      if (tid == 0) h_pose = -1;
      __syncthreads();
      if (cand >= 0) {
        if (tid == cand) {   // the thread that read this pose's counter claims a pass with it
          unsigned long long expected = c;
          if (__hip_atomic_compare_exchange_strong(&a.help_ctl[cand], &expected, c + (unsigned long long)kIcpThreads, __ATOMIC_RELAXED,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            h_pose = cand;
            h_s0 = (unsigned)(c & 0xFFFFFFFFull);
            h_ns = ns;
          }
        }
        __syncthreads();
        const int hp_pose = h_pose;
        if (hp_pose >= 0) {
          unsigned char* pub = a.help + (size_t)hp_pose * a.help_stride;
          const HelpPub* hp = reinterpret_cast<const HelpPub*>(pub);
          if (tid < kNnClasses) {
            s_sch.cnt[tid] = ld_agent(&hp->cnt[tid]);
            s_sch.slot_end[tid] = ld_agent(&hp->slot_end[tid]);
          }
          if (tid < 16) s_G[tid] = ld_agent(&hp->G[tid]);
          if (tid == 0) s_sch.base = (int)ld_agent(&hp->base);
          __syncthreads();
          help_pass<kIcpThreads>(a, t, &s_sch, s_G, pub, h_s0, h_ns, tid, true, true);
          __builtin_amdgcn_s_waitcnt(0);
          __syncthreads();
          if (tid == 0) __hip_atomic_fetch_add(&a.help_done[hp_pose], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          h_t0 = __builtin_amdgcn_s_memrealtime();
        }
        continue;   // (a lost race for the pass: look again at once)
      }
      // nothing open: done when every pose is through -- or when nobody has asked for help for a few milliseconds: an idle
      // helper holds a compute unit that a workgroup of this launch which is not resident yet (or another process) waits for
      if (tid == 0)
        h_pose = (ld_agent(a.help_finished) >= (unsigned)a.n ||
                  __builtin_amdgcn_s_memrealtime() - h_t0 > (unsigned long long)a.wait_ticks) ? -2 : -1;
      __syncthreads();
      if (h_pose == -2) break;
      __builtin_amdgcn_s_sleep(127);   // ~3 us between looks: 255 idle workgroups must not crowd the counters
    }
    return;
  }
  // (every workgroup of a clustered launch runs and leaves sooner or later -- also the one that came too late for a meeting --,
  //  so the counters are back at zero when the launch is over)
  if (CLUSTER) cluster_leave(a, pose, tid);
  if (CLUSTER && part != 0) return;
  if (tid < 16) Tg[tid] = s_G[tid];
  if (tid == 0) {
#if defined(PGP_ICP_STAMPS) && PGP_ICP_STAMPS == 4
    // level 4: every pose reports its whole time in the kernel (ticks) instead of its energy
    if (a.energy) a.energy[pose] = (float)(__builtin_amdgcn_s_memrealtime() - k_start);
    if (a.iters) a.iters[pose] = it;
    return;
#endif
#ifdef PGP_ICP_STAMPS
    if (pose == a.dbg_pose && a.energy && a.n >= 16) {
      for (int k = 1; k < 6; ++k) a.energy[k] = (float)st_acc[k];
      for (int k = 0; k < 8; ++k) a.energy[8 + k] = (float)s_dbg[k];
    }
    // poses 16.. report their whole time in the kernel (ticks) instead of their energy: tools/icp_phases.py picks
    // the slowest of them for a second run
    if (a.energy && pose >= 16) a.energy[pose] = (float)(__builtin_amdgcn_s_memrealtime() - k_start);
    else if (a.energy && (pose == a.dbg_pose || a.n < 16)) a.energy[pose] = (float)s_energy;
#else
    if (a.energy) a.energy[pose] = (float)s_energy;
#endif
    if (a.iters) a.iters[pose] = it;
  }
}

template <int METRIC, bool IMG_LDS, bool CLUSTER, bool TRIM_ONLY, int PIR>
__global__ __launch_bounds__(kIcpThreads) void icp_persist_index(IcpArgs a) {
  icp_persist_body<METRIC, IMG_LDS, CLUSTER, TRIM_ONLY, PIR>(a);
}

// one workgroup per pose, all resident, finished workgroups helping the running ones (see HelpPub)
template <bool TRIM_ONLY, int PIR>
__global__ __launch_bounds__(kIcpThreads) void icp_persist_help(IcpArgs a) {
  icp_persist_body<0, true, false, TRIM_ONLY, PIR, true>(a);
}

// SEVERAL (source, target) pairs in ONE launch: the children of an MCTS expansion belong to different objects
// (UCTSearch.cpp:200-266 -> UCTState.cpp:121-204), and the node's per-object loop (SceneCfg.cpp:379-402) refines
// the candidates of every object of a frame -- 3 x 64 poses are three clustered launches one after the other
// through the single-target entry, or 192 workgroups of one launch here.  Workgroup = one pose; its target (index
// image, geometry, vicinity graph), its segment and its slice of the transform / energy / iteration arrays come
// from the descriptor of the pose's job -- a table that travels in the kernel arguments (no upload, no sync).
constexpr int kIcpMultiMax = 8;
struct IcpTarget {
  const float4* src;
  const unsigned char* nn_image;
  const uint4* nn_vic;
  float* T;              // the job's arrays, offset so that the launch-wide pose number indexes them
  float* energy;
  int* iters;
  NnGeom nn;
  int n_src, n_tgt, k_trim, pad;
};
struct IcpMulti {
  int n_jobs;
  int first[kIcpMultiMax + 1];   // poses [first[j], first[j + 1]) belong to job j
  IcpTarget tg[kIcpMultiMax];
};
template <bool TRIM_ONLY, int PIR>
__global__ __launch_bounds__(kIcpThreads) void icp_persist_multi(IcpArgs a, IcpMulti mt) {
  const int pose = (int)blockIdx.x;
  int j = 0;
  for (int k = 1; k < mt.n_jobs; ++k) j = pose >= mt.first[k] ? k : j;   // uniform: scalar loads and compares
  const IcpTarget& tg = mt.tg[j];
  IcpArgs b = a;
  b.src = tg.src;
  b.nn_image = tg.nn_image;
  b.nn_vic = tg.nn_vic;
  b.T = tg.T;
  b.energy = tg.energy;
  b.iters = tg.iters;
  b.nn = tg.nn;
  b.n_src = tg.n_src;
  b.n_tgt = tg.n_tgt;
  b.k_trim = tg.k_trim;
  icp_persist_body<0, true, false, TRIM_ONLY, PIR>(b);
}

}  // namespace

// The instantiations of icp_persist_index.  Point-to-point with the image in LDS (the hot ones) exist per
// (several workgroups per pose, TrimmedICP form, points per thread 2 / 3 / 4); point-to-plane and the image read
// from L2 in the general form only.
constexpr int kPersistKernels = 18;
static const void* persist_kernel_at(int k) {
#define PGP_PK(M, I, C, T, R) reinterpret_cast<const void*>(icp_persist_index<M, I, C, T, (R) * kPS>)
  static const void* const tab[kPersistKernels] = {
      PGP_PK(0, true, false, false, 2), PGP_PK(0, true, false, false, 3), PGP_PK(0, true, false, false, 4),
      PGP_PK(0, true, false, true, 2),  PGP_PK(0, true, false, true, 3),  PGP_PK(0, true, false, true, 4),
      PGP_PK(0, true, true, false, 2),  PGP_PK(0, true, true, false, 3),  PGP_PK(0, true, true, false, 4),
      PGP_PK(0, true, true, true, 2),   PGP_PK(0, true, true, true, 3),   PGP_PK(0, true, true, true, 4),
      PGP_PK(1, true, false, false, 4), PGP_PK(1, true, true, false, 4),
      PGP_PK(0, false, false, false, 4), PGP_PK(0, false, true, false, 4),
      PGP_PK(1, false, false, false, 4), PGP_PK(1, false, true, false, 4)};
#undef PGP_PK
  return tab[k];
}
static const void* persist_kernel(int metric, bool img_lds, bool cluster, bool trim_only, int n_src) {
  if (metric != 1 && img_lds) {
    const int r = n_src <= 2 * kIcpBase ? 0 : (n_src <= 3 * kIcpBase ? 1 : 2);
    return persist_kernel_at((cluster ? 6 : 0) + (trim_only ? 3 : 0) + r);
  }
  if (metric == 1 && img_lds) return persist_kernel_at(cluster ? 13 : 12);
  return persist_kernel_at(14 + (metric == 1 ? 2 : 0) + (cluster ? 1 : 0));
}

static const void* help_kernel(bool trim_only, int pir) {
  static const void* const tab[6] = {
      reinterpret_cast<const void*>(icp_persist_help<false, 2 * kPS>), reinterpret_cast<const void*>(icp_persist_help<false, 3 * kPS>),
      reinterpret_cast<const void*>(icp_persist_help<false, 4 * kPS>), reinterpret_cast<const void*>(icp_persist_help<true, 2 * kPS>),
      reinterpret_cast<const void*>(icp_persist_help<true, 3 * kPS>),  reinterpret_cast<const void*>(icp_persist_help<true, 4 * kPS>)};
  return tab[(trim_only ? 3 : 0) + (pir <= 2 ? 0 : (pir == 3 ? 1 : 2))];
}
static const void* multi_kernel(bool trim_only, int pir) {
  static const void* const tab[6] = {
      reinterpret_cast<const void*>(icp_persist_multi<false, 2 * kPS>), reinterpret_cast<const void*>(icp_persist_multi<false, 3 * kPS>),
      reinterpret_cast<const void*>(icp_persist_multi<false, 4 * kPS>), reinterpret_cast<const void*>(icp_persist_multi<true, 2 * kPS>),
      reinterpret_cast<const void*>(icp_persist_multi<true, 3 * kPS>),  reinterpret_cast<const void*>(icp_persist_multi<true, 4 * kPS>)};
  return tab[(trim_only ? 3 : 0) + (pir <= 2 ? 0 : (pir == 3 ? 1 : 2))];
}

// dynamic-LDS limits of the ICP kernels: per context = per device (function attributes are per device)
static int ensure_icp_attrs(pgp_ctx* ctx) {
  if (ctx->icp_attr_set) return PGP_OK;
  const size_t lds = (size_t)kTgtTile * sizeof(float4);
  PGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(icp_refine<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  PGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(icp_refine<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  PGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(icp_refine<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const void* big[] = {reinterpret_cast<const void*>(icp_nn_index<true>), reinterpret_cast<const void*>(icp_nn_index<false>)};
  for (const void* f : big) PGP_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
  for (int v = 0; v < kPersistKernels; ++v)
    PGP_HIP(hipFuncSetAttribute(persist_kernel_at(v), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
  for (int v = 0; v < 6; ++v) {
    PGP_HIP(hipFuncSetAttribute(multi_kernel(v >= 3, 2 + v % 3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
    PGP_HIP(hipFuncSetAttribute(help_kernel(v >= 3, 2 + v % 3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
  }
  ctx->icp_attr_set = true;
  return PGP_OK;
}

// Builds the exact index of the target in ctx->d_icp_grid (image | counters | starts | keys) when its
// image fits one workgroup's LDS.  *fits = false (and PGP_OK): the caller keeps the exhaustive search.
static int build_nn_index(pgp_ctx* ctx, const float4* d_tgt, int n_tgt, int n_q, IcpArgs* a, bool* fits, hipStream_t stream,
                          unsigned long long token) {
  *fits = false;
  static_assert(sizeof(NnGeom) <= sizeof(ctx->icp_idx_geom), "NnGeom outgrew its slot in the context");
  constexpr int kLdsBytes = 160 * 1024, kScratch = 8 * 1024;   // static LDS of icp_persist_index + slack
  bool force_global = false;
  if (const char* v = getenv("PGP_ICP_IMAGE")) force_global = !strcmp(v, "global");   // A/B knob
  if (token != 0 && ctx->icp_idx_valid && ctx->icp_idx_token == token && ctx->icp_idx_tgt == (const void*)d_tgt &&
      ctx->icp_idx_ntgt == n_tgt) {
    NnGeom g;
    memcpy(&g, ctx->icp_idx_geom, sizeof g);
    const bool in_lds = !force_global && (long long)nn_lds_bytes(g.bytes, n_q) + kScratch <= kLdsBytes;
    if (in_lds || (long long)nn_lds_bytes(g.bytes, n_q, false) + kScratch <= kLdsBytes) {   // the resident index serves this call too
      a->nn = g;
      a->nn_image = ctx->d_icp_grid.as<unsigned char>();
      a->nn_image_in_lds = in_lds ? 1 : 0;
      a->nn_vic = ctx->icp_idx_vic_off ? reinterpret_cast<const uint4*>(a->nn_image + ctx->icp_idx_vic_off) : nullptr;
      *fits = true;
      return PGP_OK;
    }
  }
  ctx->icp_idx_valid = false;
  ctx->icp_grid_valid = false;   // (d_icp_grid holds either the index image or the capped search's grid)
  if (n_tgt < 1 || n_tgt > 65535) return PGP_OK;                // 16-bit positions
  if ((long long)nn_lds_bytes(0, n_q, false) + kScratch > kLdsBytes) return PGP_OK;   // the per-query arrays alone do not fit
  // n_q = queries a workgroup keeps in LDS (8 B each).  When the image (16 B per point + 4 B per cell) fits beside
  // them it is copied into LDS; else it is read from L2 and the cell count is bounded by the 16-bit tables only.
  const long long avail = (long long)kLdsBytes - kScratch - 16ll * n_tgt - 8ll * n_q - 2ll * kNnBins - 128;
  // per cell: 2 B start + 2 B representative
  long long budget = avail / 4 - 64;
  bool in_lds = !force_global && avail >= 4096 && budget >= 64 && budget * 8 >= n_tgt;   // else cells would hold > 8 points on average
  if (!in_lds) budget = 32768;
  if (budget > 32768) budget = 32768;
  float bb[6];
  int rc;
  if ((rc = device_bbox(ctx, reinterpret_cast<const float*>(d_tgt), n_tgt, 4, bb, bb + 3, stream)) != PGP_OK) return rc;
  if (!(bb[0] <= bb[3])) bb[0] = bb[1] = bb[2] = bb[3] = bb[4] = bb[5] = 0.f;   // no finite target point
  double ext[3];
  for (int k = 0; k < 3; ++k) ext[k] = fmax((double)bb[3 + k] - (double)bb[k], 1e-6);
  // cell edge from the spacing of a surface sampling of the box: ~3 points per occupied cell
  const double area = 2.0 * (ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0]);
  double h = 1.7 * sqrt(area / (double)n_tgt);
  if (const char* v = getenv("PGP_ICP_CELL")) h *= atof(v);       // tuning: cell edge multiplier
  // cells are plates: a * h in y and z, h / a^2 along x (same volume, same points per cell)
  double aspect = 1.8;   // measured flat over 1.7 .. 2 (tools/icp_time.py), 20 % faster than cubes
  if (const char* v = getenv("PGP_ICP_ASPECT")) aspect = fmax(1.0, atof(v));
  NnGeom g{};
  double hx, hyz;
  for (;;) {
    hx = h / (aspect * aspect);
    hyz = h * aspect;
    const long long nx = (long long)floor(ext[0] / hx) + 1, ny = (long long)floor(ext[1] / hyz) + 1, nz = (long long)floor(ext[2] / hyz) + 1;
    if (nx * ny * nz <= budget) {
      g.nx = (int)nx;
      g.ny = (int)ny;
      g.nz = (int)nz;
      break;
    }
    h *= 1.05;
  }
  g.h = (float)hyz;
  g.inv_h = 1.0f / g.h;
  g.hx = (float)hx;
  g.inv_hx = 1.0f / g.hx;
  g.ox = bb[0];
  g.oy = bb[1];
  g.oz = bb[2];
  g.n_cells = g.nx * g.ny * g.nz;
  g.strip_shift = 0;
  while ((g.n_cells >> g.strip_shift) > kNnStrips) ++g.strip_shift;
  g.off_start = (uint32_t)n_tgt * 16u;
  g.off_rep = (g.off_start + (uint32_t)(g.n_cells + 1) * 2u + 15u) & ~15u;
  g.bytes = (g.off_rep + (uint32_t)g.n_cells * 2u + 15u) & ~15u;
  if (in_lds && (long long)nn_lds_bytes(g.bytes, n_q) + kScratch > kLdsBytes) in_lds = false;
  const size_t nc1 = (size_t)g.n_cells + 1;
  const size_t off_ctr = ((size_t)g.bytes + 255) & ~(size_t)255, off_st = off_ctr + ((nc1 * 4 + 255) & ~(size_t)255),
               off_key = off_st + ((nc1 * 4 + 255) & ~(size_t)255),
               off_vic = (off_key + (size_t)g.n_cells * 8 + 255) & ~(size_t)255;
  bool want_vic = n_tgt <= kVicMaxTargets;
  if (const char* v = getenv("PGP_ICP_VIC")) want_vic = want_vic && atoi(v) != 0;   // A/B knob: 0 = searches only
  const int vic_chunks = (n_tgt + kVicChunk - 1) / kVicChunk;
  if ((rc = ctx->d_icp_grid.ensure(off_vic + (want_vic ? (size_t)n_tgt * 16 : 0) + 256)) != PGP_OK) return rc;
  if (want_vic && (rc = ctx->d_icp_ws.ensure((size_t)vic_chunks * n_tgt * (kVicK + 1) * 8 + 64)) != PGP_OK) return rc;
  if ((rc = ctx->d_scan_tmp.ensure((nc1 / 2048 + 2) * 4)) != PGP_OK) return rc;
  unsigned char* image = ctx->d_icp_grid.as<unsigned char>();
  uint32_t* ctr = reinterpret_cast<uint32_t*>(image + off_ctr);
  uint32_t* start = reinterpret_cast<uint32_t*>(image + off_st);
  unsigned long long* key = reinterpret_cast<unsigned long long*>(image + off_key);
  const dim3 gt((n_tgt + 255) / 256);
  PGP_HIP(hipMemsetAsync(ctr, 0, nc1 * 4, stream));
  hipLaunchKernelGGL(nnidx_scatter<false>, gt, dim3(256), 0, stream, d_tgt, n_tgt, g, ctr, (const uint32_t*)nullptr,
                     (float4*)nullptr);
  if ((rc = device_exclusive_scan(ctr, start, nc1, ctx->d_scan_tmp.as<uint32_t>(), stream)) != PGP_OK) return rc;
  PGP_HIP(hipMemsetAsync(ctr, 0, nc1 * 4, stream));
  PGP_HIP(hipMemsetAsync(key, 0xFF, (size_t)g.n_cells * 8, stream));
  hipLaunchKernelGGL(nnidx_scatter<true>, gt, dim3(256), 0, stream, d_tgt, n_tgt, g, ctr, (const uint32_t*)start,
                     reinterpret_cast<float4*>(image));
  hipLaunchKernelGGL(nnidx_rep, dim3((g.n_cells + 255) / 256, (n_tgt + 511) / 512), dim3(256), 0, stream,
                     reinterpret_cast<const float4*>(image), n_tgt, g, key);
  hipLaunchKernelGGL(nnidx_pack, dim3((g.n_cells + 1 + 255) / 256), dim3(256), 0, stream, g, (const uint32_t*)start,
                     (const unsigned long long*)key, image);
  if (want_vic) {   // the vicinity graph over the points in image order
    float2* part = ctx->d_icp_ws.as<float2>();
    uint4* vic = reinterpret_cast<uint4*>(image + off_vic);
    hipLaunchKernelGGL(nnidx_vic_partial, dim3((n_tgt + 63) / 64, vic_chunks), dim3(64), 0, stream,
                       reinterpret_cast<const float4*>(image), n_tgt, part);
    hipLaunchKernelGGL(nnidx_vic_merge, dim3((n_tgt + 63) / 64), dim3(64), 0, stream, (const float2*)part, n_tgt, vic_chunks, vic);
  }
  PGP_HIP(hipGetLastError());
  a->nn = g;
  a->nn_image = image;
  a->nn_image_in_lds = in_lds ? 1 : 0;
  a->nn_vic = want_vic ? reinterpret_cast<const uint4*>(image + off_vic) : nullptr;
  ctx->icp_idx_vic_off = want_vic ? off_vic : 0;
  *fits = true;
  memcpy(ctx->icp_idx_geom, &g, sizeof g);
  ctx->icp_idx_valid = token != 0;
  ctx->icp_idx_token = token;
  ctx->icp_idx_tgt = (const void*)d_tgt;
  ctx->icp_idx_ntgt = n_tgt;
  ctx->icp_idx_nq = n_q;
  return PGP_OK;
}

// Clustered launches (several workgroups per pose that meet every iteration) must not interleave with each other on
// a device: two of them, each half resident, would wait for partners that cannot be scheduled (until the clock bound of
// their waits frees them).  This chain rules it out for every stream of THIS process: each launch of resident
// workgroups waits for the previous one's completion event (no host synchronisation).
namespace {
struct CoopChain {
  std::mutex mu;
  hipEvent_t last[64] = {};
};
CoopChain g_coop;
}  // namespace

// The host-pointer call redoes a scene-sized job whose one-launch form reported a pose as lost (iteration count -1: its
// units did not arrive within the clock bound) with the host-driven iterations: this thread's next launch_icp calls.
// The floor of the clock bounds of the kernels whose workgroups wait for each other, in ticks of the 100 MHz counter: 3 ms --
// two orders above an iteration (~25 us), three below the 2 s of round 5.  PGP_ICP_WAIT_MS overrides it (tests).
static unsigned icp_wait_ticks() {   // (read at every launch: a test changes it between calls)
  double ms = 3.0;
  if (const char* v = getenv("PGP_ICP_WAIT_MS")) ms = atof(v) > 0.0 ? atof(v) : ms;
  const double t = ms * 1e5;
  return (unsigned)(t < 100.0 ? 100.0 : (t > 4.0e9 ? 4.0e9 : t));
}

static thread_local bool t_scene_form_off = false;
void icp_scene_form_off(bool off) { t_scene_form_off = off; }

// A launch whose workgroups wait for each other (the clustered, the helping and the scene-sized kernels).  By default a
// PLAIN launch of a grid that fits on the device at the kernel's occupancy: on an idle or lightly shared device every
// workgroup is resident at once; behind somebody else's long kernel the late ones arrive when it ends, and every wait in
// these kernels is bounded by a clock (a lost meeting is finished by the pose's first workgroup; a lost unit of the scene-
// sized form ends the pose with iteration count -1, and the host-pointer call redoes the job host-driven).  hipLaunchCooperativeKernel gives the same launch a runtime check and a queue of its own -- and
// that queue, once it exists, makes the hardware scheduler time-slice the device between processes: every OTHER process
// on the GPU (the node's segmentation network, say) then meets stalls of ~11 ms although this one is idle (measured:
// tools/child_under_parent.py, profiles/r05_ab/cooperative_queue_stalls.log).  PGP_COOPERATIVE_LAUNCH=1: the runtime's form.
static hipError_t launch_resident(pgp_ctx* ctx, const void* fn, unsigned grid, unsigned threads, void** params, unsigned lds,
                                  hipStream_t stream) {
  const char* coop = getenv("PGP_COOPERATIVE_LAUNCH");
  if (coop && atoi(coop) != 0) return hipLaunchCooperativeKernel(fn, dim3(grid), dim3(threads), params, lds, stream);
  struct Known {
    const void* fn;
    unsigned threads, lds;
    int per_cu;
  };
  static std::mutex mu;
  static std::vector<Known> known;
  int per_cu = -1;
  {
    std::lock_guard<std::mutex> lk(mu);
    for (const Known& k : known)
      if (k.fn == fn && k.threads == threads && k.lds == lds) per_cu = k.per_cu;
    if (per_cu < 0) {
      int v = 0;
      const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, (int)threads, lds);
      if (e != hipSuccess) return e;
      per_cu = v;
      known.push_back(Known{fn, threads, lds, v});
    }
  }
  if ((long long)grid > (long long)per_cu * ctx->n_cus) return hipErrorCooperativeLaunchTooLarge;
  return hipLaunchKernel(fn, dim3(grid), dim3(threads), params, lds, stream);
}

// the options of a call (pgp_icp_options) as kernel arguments; k_trim = the trimmed count of a cloud of n_src points
static int icp_trim_count(const pgp_icp_options* prm, int n_src) {
  float tf = prm->trim_fraction;
  if (!(tf > 0.f) || tf > 1.f) tf = 1.f;
  // float numPoints = trim * size; align(..., abs(numPoints), ...) -> int (UCTState.cpp:176,194)
  int k = (int)fabsf(tf * (float)n_src);
  if (k < 1) k = 1;
  if (k > n_src) k = n_src;
  return k;
}
static void icp_option_args(const pgp_icp_options* prm, int n_src, IcpArgs* a) {
  a->max_iter = prm->max_iterations > 0 ? prm->max_iterations : 100;
  a->k_trim = icp_trim_count(prm, n_src);
  a->max_corr2 = prm->max_corr_dist > 0.f ? prm->max_corr_dist * prm->max_corr_dist : -1.f;
  a->ratio = prm->energy_ratio;            // <= 0: the energy-ratio test is off
  a->metric = prm->error_metric;
  a->t_eps = prm->transformation_epsilon;  // < 0: off
  a->rel_mse = prm->relative_mse;
  a->abs_mse = prm->absolute_mse;
  a->diff_rot = prm->min_diff_rot;
  a->diff_trans = prm->min_diff_trans;
  a->smooth = (prm->min_diff_rot > 0.f && prm->min_diff_trans > 0.f)
                  ? (prm->smooth_length < 1 ? 1 : (prm->smooth_length > kMaxSmooth ? kMaxSmooth : prm->smooth_length))
                  : 0;
}
static bool icp_trim_only(const IcpArgs& a) {
  return !(a.max_corr2 >= 0.f) && !(a.t_eps >= 0.f || a.rel_mse > 0.f || a.abs_mse >= 0.f || a.smooth > 0);
}

int launch_icp(pgp_ctx* ctx, const float4* d_src, int n_src, const float4* d_tgt, const float4* d_tgt_n, int n_tgt,
               float* d_T, int n, const pgp_icp_options* prm, float* d_energy, int* d_iters, hipStream_t stream,
               unsigned long long tgt_token) {
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
                          d_energy ? d_energy + off : nullptr, d_iters ? d_iters + off : nullptr, stream, tgt_token);
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
  icp_option_args(prm, n_src, &a);
  a.first_walk = 1;
  if (const char* v = getenv("PGP_ICP_FIRST_WALK")) a.first_walk = atoi(v) < 0 ? 0 : atoi(v);   // A/B knob
  a.wait_ticks = icp_wait_ticks();
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
  // exact index of the static target: the default whenever its image fits a workgroup's LDS.  The
  // exhaustive searches stay as the checker paths (nn_search 1, PGP_ICP_NN=scan, or PGP_ICP_SPLIT set).
  bool use_index = false, persist_index = false;
  {
    const char* env_nn = getenv("PGP_ICP_NN");
    const bool legacy = getenv("PGP_ICP_SPLIT") != nullptr;
    bool want = !use_grid && !legacy && (prm->nn_search == 0 || prm->nn_search == 3);
    if (env_nn && !strcmp(env_nn, "scan")) want = false;
    if (env_nn && !strcmp(env_nn, "index") && !use_grid && prm->nn_search != 1) want = true;
    bool want_persist = n_src <= kPiR * kIcpThreads;
    if (const char* v = getenv("PGP_ICP_PERSIST")) want_persist = want_persist && atoi(v) != 0;
    if (want) {
      // one persistent workgroup per pose keeps all n_src correspondences in LDS; the host-driven path 1024
      if (want_persist) {
        if ((rc = build_nn_index(ctx, d_tgt, n_tgt, n_src, &a, &use_index, stream, tgt_token)) != PGP_OK) return rc;
        persist_index = use_index;
      }
      if (!use_index && (rc = build_nn_index(ctx, d_tgt, n_tgt, kIdxThreads, &a, &use_index, stream, tgt_token)) != PGP_OK) return rc;
      if (!use_index && prm->nn_search == 3) {
        set_error("icp: the target (%d points) does not fit the LDS index", n_tgt);
        return PGP_EINVAL;
      }
    }
  }
  if (use_grid || a.smooth > 0 || use_index) split = true;   // all live on the host-driven path
  const size_t hist_bytes = a.smooth > 0 ? (size_t)n * (kMaxSmooth + 1) * 7 * 8 : 0;
  a.energy = d_energy;
  a.iters = d_iters;
  const size_t lds = (size_t)kTgtTile * sizeof(float4);
  if ((rc = ensure_icp_attrs(ctx)) != PGP_OK) return rc;
  if (persist_index) {
    // everything of an iteration lives in LDS: the workspace holds the pointmatcher history only
    if (a.smooth > 0) {
      if ((rc = ctx->d_icp_ws.ensure(hist_bytes + 64)) != PGP_OK) return rc;
      a.st_hist = ctx->d_icp_ws.as<double>();
      PGP_HIP(hipMemsetAsync(a.st_hist, 0, hist_bytes, stream));
    }
    const bool il = a.nn_image_in_lds != 0;
    const size_t plds = nn_lds_bytes(a.nn.bytes, n_src, il);
    const bool trim_only = icp_trim_only(a);
    const void* fn = persist_kernel(a.metric, il, false, trim_only, n_src);
    const void* fn_cluster = persist_kernel(a.metric, il, true, trim_only, n_src);
    // Few poses: 2 or 4 workgroups per pose share the search (64 poses alone would use 64 of the 256 CUs).  They
    // meet once per iteration, so all of them should be resident at once: a grid that fits the device, one workgroup
    // per CU, chained behind the previous such launch (launch_resident, g_coop).  Not while the stream is being
    // captured (the chain's event wait is not for a graph) and not with the pointmatcher history.
    a.wgs_per_pose = 1;
    if (const char* v = getenv("PGP_ICP_DBG_POSE")) a.dbg_pose = atoi(v);
    if (const char* v = getenv("PGP_ICP_SLOTS")) a.slot_budget = atoi(v);
    if (const char* v = getenv("PGP_ICP_ROWS")) a.rows_mode = atoi(v);
    int want_wgs = n * 4 <= ctx->n_cus ? 4 : (n * 2 <= ctx->n_cus ? 2 : 1);
    if (const char* v = getenv("PGP_ICP_WGS")) want_wgs = atoi(v) == 4 ? 4 : (atoi(v) == 2 ? 2 : 1);   // A/B knob
    if (want_wgs > 1 && a.smooth == 0 && n * want_wgs <= ctx->n_cus && n_src >= 64 * want_wgs) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusActive;
      if (cap == hipStreamCaptureStatusNone) a.wgs_per_pose = want_wgs;
    }
    if (getenv("PGP_ICP_DEBUG")) fprintf(stderr, "icp: n_cus %d want %d -> %d\n", ctx->n_cus, want_wgs, a.wgs_per_pose);
    if (a.wgs_per_pose > 1) {
      const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
      // The chain first: the previous clustered launch of this device (any stream of this process) has finished
      // before this one's buffers are (re)allocated, its counters zeroed and its workgroups started.
      std::lock_guard<std::mutex> chain(g_coop.mu);
      if (g_coop.last[dev]) PGP_HIP(hipStreamWaitEvent(stream, g_coop.last[dev], 0));
      else PGP_HIP(hipEventCreateWithFlags(&g_coop.last[dev], hipEventDisableTiming));
      // meeting buffers | arrival counters | workgroups that have left | search ticks
      const size_t xbytes = 2 * need * 8, ctr_words = 3 * (size_t)n + 4;
      const void* x_before = ctx->d_icp_x.p;
      const size_t x_cap_before = ctx->d_icp_x.cap;
      if ((rc = ctx->d_icp_x.ensure(xbytes + ctr_words * 4 + (size_t)n * 32 + 64)) != PGP_OK) return rc;
      a.x_buf = ctx->d_icp_x.as<unsigned long long>();
      a.x_ctr = reinterpret_cast<unsigned*>(a.x_buf + 2 * need);
      a.x_done = a.x_ctr + n + 4;
      a.x_abandon = a.x_done + n;
      if (getenv("PGP_ICP_FORCE_LOST")) a.force_lost = 1;   // test knob: the first meeting of every pose counts as lost
      a.x_ticks = a.x_ctr + ctr_words;
      a.solo_ticks = 1100;   // 11 us (tools/icp_time.py, PGP_ICP_SOLO_TICKS sweep)
      if (const char* v = getenv("PGP_ICP_SOLO_TICKS")) a.solo_ticks = (unsigned)atoi(v);
      // The counters are zero whenever a clustered launch on this context is over (every pose's last workgroup to leave
      // puts them back, lost meeting or not: cluster_leave): a fill only when these words were not the counters of the
      // last such launch (first use, a grown buffer, another pose count, the helping launch's layout in between).
      const bool same_words = ctx->icp_x_clean && ctx->d_icp_x.p == x_before && ctx->d_icp_x.cap == x_cap_before &&
                              ctx->icp_x_n == n && ctx->icp_x_need == (size_t)need;
      if (!same_words) PGP_HIP(hipMemsetAsync(a.x_ctr, 0, ctr_words * 4, stream));
      ctx->icp_x_clean = true;
      ctx->icp_x_n = n;
      ctx->icp_x_need = (size_t)need;
      void* params[] = {&a};
      hipError_t e = launch_resident(ctx, fn_cluster, (unsigned)(n * a.wgs_per_pose), kIcpThreads, params, (unsigned)plds, stream);
      if (getenv("PGP_ICP_DEBUG"))
        fprintf(stderr, "icp: %d poses x %d workgroups, resident launch: %s\n", n, a.wgs_per_pose, hipGetErrorString(e));
      if (e == hipSuccess) {
        // Nothing behind the kernel: a meeting that is lost (another process holding the chip's compute units for
        // seconds) makes the pose's workgroup 0 go on alone inside the launch (icp_persist_body), so the caller --
        // host-pointer or device-pointer API -- always gets refined transforms.  (Rounds 3-4 queued a repair launch
        // behind every clustered launch for that case: 1.6 us + ~12 us of dependency latency per call.)
        PGP_HIP(hipEventRecord(g_coop.last[dev], stream));
        PGP_HIP(hipGetLastError());
        return PGP_OK;
      }
      (void)hipGetLastError();   // the grid does not fit on this device at the kernel's occupancy: one workgroup per pose
      a.wgs_per_pose = 1;
    }
    // One workgroup per pose and all of them resident at once (129 .. 256 poses on 256 compute units): a workgroup
    // that is through with its pose takes search passes of the poses still running (HelpPub) -- the launch then
    // lasts about as long as the MEAN pose, not the slowest.  Launched and chained like the clustered launches
    // (launch_resident); a repair launch behind it (the clustered launch needs none any more).
    // MEASURED SLOWER and therefore OFF unless PGP_ICP_HELP=1 (profiles/r04_ab/icp_helping.log: 256 poses from far
    // 0.84 -> 1.00 ms, from near 0.25 -> 0.28 ms, same bits): the helped kernel's own passes lose the one-trip-ahead
    // prefetch of the plain slot loop, a published iteration costs ~10 us of write-through traffic and waiting, and
    // idle workgroups only exist once the fast poses are through -- when the slow ones have few far iterations left.
    bool want_help = a.wgs_per_pose == 1 && n >= 2 && n <= ctx->n_cus && a.metric == 0 && il && a.smooth == 0;
    {
      const char* v = getenv("PGP_ICP_HELP");
      want_help = want_help && v && atoi(v) != 0;
    }
    if (want_help) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) want_help = false;
    }
    if (want_help) {
      const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
      std::lock_guard<std::mutex> chain(g_coop.mu);
      if (g_coop.last[dev]) PGP_HIP(hipStreamWaitEvent(stream, g_coop.last[dev], 0));
      else PGP_HIP(hipEventCreateWithFlags(&g_coop.last[dev], hipEventDisableTiming));
      // counters (claim words | slot counts | passes done | finished, lost) | saved transforms | publications
      const size_t N = (size_t)n, hdr = (N * 16 + 8 + 255) & ~(size_t)255, save = (N * 64 + 255) & ~(size_t)255, hb = help_bytes(n_src);
      if ((rc = ctx->d_icp_x.ensure(hdr + save + N * hb + 256)) != PGP_OK) return rc;
      ctx->icp_x_clean = false;   // (the clustered launch's counters live in the same buffer)
      unsigned char* base = ctx->d_icp_x.as<unsigned char>();
      IcpArgs h = a;
      h.help_ctl = reinterpret_cast<unsigned long long*>(base);
      h.help_nslots = reinterpret_cast<unsigned*>(base + N * 8);
      h.help_done = h.help_nslots + N;
      h.help_finished = h.help_done + N;
      h.x_lost = h.help_finished + 1;
      h.T_save = reinterpret_cast<float*>(base + hdr);
      h.help = base + hdr + save;
      h.help_stride = hb;
      PGP_HIP(hipMemsetAsync(base, 0, hdr, stream));
      void* hparams[] = {&h};
      hipError_t e = launch_resident(ctx, help_kernel(trim_only, n_src <= 2 * kIcpBase ? 2 : (n_src <= 3 * kIcpBase ? 3 : 4)),
                                     (unsigned)n, kIcpThreads, hparams, (unsigned)plds, stream);
      if (getenv("PGP_ICP_DEBUG")) fprintf(stderr, "icp: %d poses, helping launch: %s\n", n, hipGetErrorString(e));
      if (e == hipSuccess) {
        IcpArgs fix = a;
        fix.wgs_per_pose = 1;
        fix.run_if = h.x_lost;
        fix.T_in = h.T_save;
        fix.T_save = nullptr;
        void* fparams[] = {&fix};
        PGP_HIP(hipLaunchKernel(fn, dim3(n), dim3(kIcpThreads), fparams, plds, stream));
        PGP_HIP(hipEventRecord(g_coop.last[dev], stream));
        PGP_HIP(hipGetLastError());
        return PGP_OK;
      }
      (void)hipGetLastError();   // the grid does not fit here: the launch without helpers below
    }
    void* params[] = {&a};
    PGP_HIP(hipLaunchKernel(fn, dim3(n), dim3(kIcpThreads), params, plds, stream));
    PGP_HIP(hipGetLastError());
    return PGP_OK;
  }
  // Targets beyond the exact index's 65 535 points, no correspondence cap: a uniform grid answers every query that has a
  // neighbour within a safe radius, the exhaustive scan only the others (icp_nn_grid_open; PGP_ICP_OPEN_GRID=0: the scan alone)
  bool open_grid = split && !use_grid && !use_index && a.max_corr2 < 0.f && n_tgt > 65535 && prm->nn_search != 1;
  if (const char* v = getenv("PGP_ICP_OPEN_GRID")) open_grid = open_grid && atoi(v) != 0;
  // Scenes beyond one block of 4096 points, nothing to trim (a cap, or every pair kept): the iteration's sums are formed by one
  // workgroup per block (icp_sums_partial) and icp_refine only adds them up (PGP_ICP_PART=0: one workgroup walks the scene)
  const int n_blk = (n_src + kSumR * kIcpThreads - 1) / (kSumR * kIcpThreads);
  bool part_sums = split && n_blk > 1 && !(a.max_corr2 < 0.f && a.k_trim < a.n_src);
  if (const char* v = getenv("PGP_ICP_PART")) part_sums = part_sums && atoi(v) != 0;
  const size_t part_bytes = part_sums ? (size_t)n * n_blk * kPartStride * 8 + 64 : 0;
  const size_t state_bytes = split ? need * 8 + (size_t)n * 16 + 64 + hist_bytes + 64 + (use_index ? need * 4 + 64 : 0) +
                                         (open_grid ? need * 4 + (size_t)n * 4 + 128 : 0) + part_bytes : 0;
  if ((rc = ctx->d_icp_ws.ensure(need * 8 + state_bytes + 64)) != PGP_OK) return rc;
  a.ws_d2 = ctx->d_icp_ws.as<float>();
  a.ws_j = reinterpret_cast<int*>(a.ws_d2 + need);
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
  if (use_index) a.ws_pos = reinterpret_cast<int*>(((uintptr_t)(a.n_done + 1) + hist_bytes + 31) & ~(uintptr_t)15);
  // the state of the host-driven iterations (not needed -- and its host synchronisation not paid -- by the one-launch form below)
  auto init_split_state = [&]() -> int {
    if (use_index) PGP_HIP(hipMemsetAsync(a.ws_pos, 0xFF, need * 4, stream));   // no previous correspondence yet
    PGP_HIP(hipMemsetAsync(a.ws_key, 0xFF, need * 8, stream));
    PGP_HIP(hipMemsetAsync(a.ws_j, 0xFF, need * 4, stream));  // no previous correspondence yet
    PGP_HIP(hipMemsetAsync(a.st_it, 0, (size_t)n * 8 + 4, stream));
    if (a.st_hist) PGP_HIP(hipMemsetAsync(a.st_hist, 0, hist_bytes, stream));
    std::vector<double> e0((size_t)n, (double)FLT_MAX);
    PGP_HIP(hipMemcpyAsync(a.st_E, e0.data(), (size_t)n * 8, hipMemcpyHostToDevice, stream));
    PGP_HIP(hipStreamSynchronize(stream));  // e0 is a stack temporary
    return PGP_OK;
  };
  // (up to 64 poses: the reference's call has one; the unit sums of many poses would be gigabytes)
  bool scene_persist = use_grid && n_blk > 1 && !(a.max_corr2 < 0.f && a.k_trim < a.n_src) && a.smooth == 0 && ctx->n_cus > 0 && n <= 64 &&
                       !t_scene_form_off;
  if (const char* v = getenv("PGP_ICP_PART")) scene_persist = scene_persist && atoi(v) != 0;
  if (const char* v = getenv("PGP_ICP_SCENE_PERSIST")) scene_persist = scene_persist && atoi(v) != 0;
  if (scene_persist) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) scene_persist = false;
  }
  if (!scene_persist && (rc = init_split_state()) != PGP_OK) return rc;
  double* d_part = nullptr;
  if (part_sums) {   // the last part_bytes of the workspace
    unsigned char* end = ctx->d_icp_ws.as<unsigned char>() + need * 8 + state_bytes + 64;
    d_part = reinterpret_cast<double*>(((uintptr_t)(end - part_bytes) + 47) & ~(uintptr_t)15);
  }
  int* open_list = nullptr;
  int* open_cnt = nullptr;
  float open_r2 = 0.f;
  if (open_grid) {
    unsigned char* q = reinterpret_cast<unsigned char*>(a.n_done + 1) + hist_bytes + 64;
    q = reinterpret_cast<unsigned char*>(((uintptr_t)q + 15) & ~(uintptr_t)15);
    open_list = reinterpret_cast<int*>(q);
    open_cnt = open_list + need;
  }
  // The grid of the capped search stays valid across calls like the exact index does: same (target pointer, size, token
  // != 0) and the same cap = the same cells (the reference aligns every frame's scene to the SAME table model with the same
  // cap, SceneCfg.cpp:101,135-141: bounding box, two scatters and a scan -- ~0.15 ms and a host synchronisation -- per call).
  const bool grid_cached = use_grid && !open_grid && tgt_token != 0 && ctx->icp_grid_valid && ctx->icp_grid_token == tgt_token &&
                           ctx->icp_grid_tgt == (const void*)d_tgt && ctx->icp_grid_ntgt == n_tgt && ctx->icp_grid_cap == prm->max_corr_dist;
  if (grid_cached) {
    a.gox = ctx->icp_grid_geom[0];
    a.goy = ctx->icp_grid_geom[1];
    a.goz = ctx->icp_grid_geom[2];
    a.ginv_h = ctx->icp_grid_geom[3];
    a.gnx = ctx->icp_grid_n[0];
    a.gny = ctx->icp_grid_n[1];
    a.gnz = ctx->icp_grid_n[2];
    const size_t cells = (size_t)a.gnx * a.gny * a.gnz;
    const size_t off_pts = ((cells + 1) * 8 + 64 + 255) & ~(size_t)255;
    unsigned char* gb = ctx->d_icp_grid.as<unsigned char>();
    a.gcell_start = reinterpret_cast<uint32_t*>(gb + 64) + (cells + 1);
    a.gpts = reinterpret_cast<float4*>(gb + off_pts);
  }
  if ((use_grid || open_grid) && !grid_cached) {
    ctx->icp_grid_valid = false;
    ctx->icp_idx_valid = false;   // d_icp_grid is about to hold the search's grid
    // ---- the target's grid: bounding box (device), cell edge >= max_corr (grown to keep <= 2^26 cells)
    float bb[6];
    if ((rc = device_bbox(ctx, reinterpret_cast<const float*>(d_tgt), n_tgt, 4, bb, bb + 3, stream)) != PGP_OK) return rc;
    if (!(bb[0] <= bb[3])) bb[0] = bb[1] = bb[2] = bb[3] = bb[4] = bb[5] = 0.f;   // no finite target point
    float maxabs = 0.f;
    for (int q = 0; q < 6; ++q) maxabs = fmaxf(maxabs, fabsf(bb[q]));
    // margin for the rounding of the cell coordinate: a point within max_corr of a query is at most
    // one cell away from the query's cell
    const float margin = 64.f * FLT_EPSILON * maxabs;
    float h = prm->max_corr_dist * 1.001f + margin;
    if (open_grid) {
      // no cap to size the cells by: twice the spacing of n_tgt points spread over the bounding box's surface (a table,
      // a room's walls), so that whatever lies within ~2 spacings of the target -- every query of a pose that is not far
      // off -- is settled by the grid, from a few dozen candidates.  Measured (tools/icp_big_target.py, 30 000 x 100 000,
      // 10 iterations): 1.5 / 2 / 3 / 4 / 6 spacings -> 2.63 / 2.32 / 2.94 / 3.84 / 6.67 ms from 4 mm off and 3.86 / 4.12 /
      // 4.69 / 5.53 / 8.11 ms from 8 cm off, against 6.3 ms for the exhaustive scan alone.
      const double ex = (double)bb[3] - bb[0], ey = (double)bb[4] - bb[1], ez = (double)bb[5] - bb[2];
      const double area = 2.0 * (ex * ey + ey * ez + ez * ex);
      double spacings = 2.0;
      if (const char* v = getenv("PGP_ICP_OPEN_CELL")) spacings = atof(v) > 0.0 ? atof(v) : spacings;   // A/B knob
      h = (float)(spacings * std::sqrt(std::max(area, 1e-12) / (double)n_tgt)) + margin;
      if (!(h > 0.f) || !std::isfinite(h)) h = 1.f;
    }
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
    if (open_grid) {
      // every target point within r of a query lies in the 27 cells around the query's: r = (h - margin) / 1.001, taken a
      // little smaller still (the comparison is on squared distances as the kernels round them)
      const double r = ((double)h - (double)margin) / 1.001;
      open_r2 = (float)(r * r * (1.0 - 1e-5));
      if (!(open_r2 > 0.f)) open_r2 = 0.f;
      if (getenv("PGP_ICP_DEBUG"))
        fprintf(stderr, "icp: open grid over %d target points: cell %.4g, %d x %d x %d cells, settles neighbours within %.4g\n", n_tgt,
                (double)h, a.gnx, a.gny, a.gnz, r);
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
    if (use_grid && !open_grid && tgt_token != 0) {
      ctx->icp_grid_valid = true;
      ctx->icp_grid_token = tgt_token;
      ctx->icp_grid_tgt = (const void*)d_tgt;
      ctx->icp_grid_ntgt = n_tgt;
      ctx->icp_grid_cap = prm->max_corr_dist;
      ctx->icp_grid_geom[0] = a.gox;
      ctx->icp_grid_geom[1] = a.goy;
      ctx->icp_grid_geom[2] = a.goz;
      ctx->icp_grid_geom[3] = a.ginv_h;
      ctx->icp_grid_n[0] = a.gnx;
      ctx->icp_grid_n[1] = a.gny;
      ctx->icp_grid_n[2] = a.gnz;
    }
  }
  // The capped grid search with sums by block -- the shape of the reference's table alignment -- as ONE launch of resident
  // workgroups (icp_scene_persist): every iteration on the device, no host round trip.  PGP_ICP_SCENE_PERSIST=0: the host-
  // driven iterations below (the checker of that kernel: same bits); also taken while the stream is being captured and with
  // the pointmatcher history.
  {
    if (scene_persist) {
      SceneArgs z{};
      z.n_chunks = (n_src + kSceneQ - 1) / kSceneQ;
      z.n_units = (n_src + 255) / 256;
      z.n_blk = n_blk;
      z.n_units16 = n_blk * 16;
      const size_t N = (size_t)n, w_bytes = N * z.n_units16 * kPartStride * 8, uc_bytes = (N * z.n_units16 * 4 + 63) & ~(size_t)63;
      const size_t st_bytes = N * kSceneRep * kSceneRepStride * 4, pc_bytes = (N * 4 + 63) & ~(size_t)63, eo_bytes = (N * 8 + 63) & ~(size_t)63;
      const size_t tail = pc_bytes + st_bytes + eo_bytes + 64;
      const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
      std::lock_guard<std::mutex> chain(g_coop.mu);   // launches of resident workgroups of one process never overlap on a device
      if (g_coop.last[dev]) PGP_HIP(hipStreamWaitEvent(stream, g_coop.last[dev], 0));
      else PGP_HIP(hipEventCreateWithFlags(&g_coop.last[dev], hipEventDisableTiming));
      if ((rc = ctx->d_icp_x.ensure(w_bytes + uc_bytes + tail + 256)) != PGP_OK) return rc;
      ctx->icp_x_clean = false;   // (the clustered launch's counters live in the same buffer)
      unsigned char* xb = ctx->d_icp_x.as<unsigned char>();
      z.W = reinterpret_cast<double*>(xb);
      z.unit_ctr = reinterpret_cast<unsigned*>(xb + w_bytes);
      z.pose_ctr = reinterpret_cast<unsigned*>(xb + w_bytes + uc_bytes);
      z.state = reinterpret_cast<unsigned*>(xb + w_bytes + uc_bytes + pc_bytes);
      z.E_old = reinterpret_cast<double*>(xb + w_bytes + uc_bytes + pc_bytes + st_bytes);
      z.lost = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(z.E_old) + eo_bytes);
      z.keys = a.ws_key;
#ifdef PGP_SCENE_STAMPS
      static unsigned long long* d_dbg = nullptr;
      if (!d_dbg) PGP_HIP(hipMalloc(&d_dbg, (1024 + 4096) * 8));
      PGP_HIP(hipMemsetAsync(d_dbg, 0, (1024 + 4096) * 8, stream));
      z.dbg = d_dbg;
#endif
      PGP_HIP(hipMemsetAsync(xb, 0, w_bytes + uc_bytes + tail, stream));
      int per_cu = 0;
      const void* fn_scene = a.metric == 1 ? reinterpret_cast<const void*>(icp_scene_persist<1>) : reinterpret_cast<const void*>(icp_scene_persist<0>);
      if (a.metric == 1) PGP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, icp_scene_persist<1>, kSceneThreads, 0));
      else PGP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, icp_scene_persist<0>, kSceneThreads, 0));
      const long long items = (long long)n * z.n_chunks, room = (long long)per_cu * ctx->n_cus;
      z.n_upd = std::min(n, kSceneUpdMax);
      unsigned grid = (unsigned)std::max<long long>(z.n_upd + 1, std::min(items + z.n_upd, room));
      if (const char* v = getenv("PGP_ICP_SCENE_WGS")) grid = std::max((unsigned)z.n_upd + 1u, std::min(grid, (unsigned)atoi(v)));   // A/B knob
      z.poll_sleep = 1;
      z.wait_ticks = icp_wait_ticks();
      if (const char* v = getenv("PGP_ICP_SCENE_SLEEP")) z.poll_sleep = atoi(v);
      if (getenv("PGP_ICP_FORCE_LOST")) z.force_lost = 1;
      void* params[] = {&a, &z};
      const hipError_t e = room > z.n_upd ? launch_resident(ctx, fn_scene, grid, kSceneThreads, params, 0, stream) : hipErrorInvalidValue;
      if (getenv("PGP_ICP_DEBUG"))
        fprintf(stderr, "icp: scene-sized capped ICP in one launch: %d poses x %d chunks on %u workgroups (%d per CU): %s\n", n, z.n_chunks,
                grid, per_cu, hipGetErrorString(e));
      if (e == hipSuccess) {
        PGP_HIP(hipEventRecord(g_coop.last[dev], stream));
        PGP_HIP(hipGetLastError());
#ifdef PGP_SCENE_STAMPS
        {
          std::vector<unsigned long long> h(1024 + 4096);
          PGP_HIP(hipStreamSynchronize(stream));
          PGP_HIP(hipMemcpy(h.data(), d_dbg, (1024 + 4096) * 8, hipMemcpyDeviceToHost));
          {   // iteration 10: every worker's go and searched times relative to the publication of iteration 9
            const unsigned long long pub = h[9 * 16 + 3];
            std::vector<double> go, se;
            for (int w = 0; w < 2048; ++w)
              if (h[1024 + w]) {
                go.push_back((double)((long long)(h[1024 + w] - pub)) * 0.01);
                se.push_back((double)((long long)(h[1024 + 2048 + w] - pub)) * 0.01);
              }
            if (!go.empty()) {
              auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
              fprintf(stderr, "iteration 10, %zu workers: go min %.2f median %.2f p90 %.2f max %.2f | searched min %.2f median %.2f p90 %.2f p99 %.2f max %.2f us\n",
                      go.size(), pct(go, 0), pct(go, 0.5), pct(go, 0.9), pct(go, 1), pct(se, 0), pct(se, 0.5), pct(se, 0.9), pct(se, 0.99), pct(se, 1));
              int slow = 0;
              for (size_t w = 0; w < se.size() && slow < 12; ++w)
                if (se[w] > pct(se, 0.97)) { fprintf(stderr, "  worker %zu: go %.2f searched %.2f\n", w, go[w], se[w]); ++slow; }
            }
          }
          // per iteration, in 10 ns ticks relative to the previous publication: worker 0's two chunks (go, searched, ticket),
          // the last unit closed, the updater (units in, sums added, solved, published)
          unsigned long long prev = h[4];
          for (int it = 0; it < 64 && h[it * 16 + 3]; ++it) {
            const unsigned long long* r = &h[it * 16];
            auto d = [&](unsigned long long v) { return v ? (double)((long long)(v - prev)) * 0.01 : -1.0; };
            fprintf(stderr, "it %2d  w0 chunk A: go %6.2f searched %6.2f ticket %6.2f | last unit closed %6.2f | updater: units in %6.2f summed %6.2f solved %6.2f "
                            "published %6.2f us\n", it, d(r[4]), d(r[5]), d(r[6]), d(r[12]), d(r[0]), d(r[1]), d(r[2]), d(r[3]));
            prev = r[3];
          }
        }
#endif
        return PGP_OK;
      }
      (void)hipGetLastError();   // the grid does not fit here: the host-driven iterations
      if ((rc = init_split_state()) != PGP_OK) return rc;
    }
  }
  const dim3 gnn((n_src + kNnThreads * kIcpR - 1) / (kNnThreads * kIcpR), (n_tgt + kNnTgt - 1) / kNnTgt, n);
  const dim3 ggrid((unsigned)(((size_t)n_src * kGridLanes + 255) / 256), n);
  for (int it = 0; it < a.max_iter; ++it) {
    if (use_grid) hipLaunchKernelGGL(icp_nn_grid, ggrid, dim3(256), 0, stream, a);
    else if (open_grid) {
      PGP_HIP(hipMemsetAsync(open_cnt, 0, (size_t)n * 4, stream));
      hipLaunchKernelGGL(icp_nn_grid_open, ggrid, dim3(256), 0, stream, a, open_r2, open_list, open_cnt);
      hipLaunchKernelGGL(icp_nn_split<true>, gnn, dim3(kNnThreads), 0, stream, a, (const int*)open_list, (const int*)open_cnt);
    } else if (use_index && a.nn_image_in_lds)
      hipLaunchKernelGGL(icp_nn_index<true>, dim3((n_src + kIdxThreads - 1) / kIdxThreads, n), dim3(kIdxThreads),
                         nn_lds_bytes(a.nn.bytes, kIdxThreads, true), stream, a);
    else if (use_index)
      hipLaunchKernelGGL(icp_nn_index<false>, dim3((n_src + kIdxThreads - 1) / kIdxThreads, n), dim3(kIdxThreads),
                         nn_lds_bytes(a.nn.bytes, kIdxThreads, false), stream, a);
    else hipLaunchKernelGGL(icp_nn_split<false>, gnn, dim3(kNnThreads), 0, stream, a, (const int*)nullptr, (const int*)nullptr);
    if (part_sums) {
      hipLaunchKernelGGL(icp_sums_partial, dim3(n_blk, n), dim3(kIcpThreads), 0, stream, a, d_part, n_blk);
      hipLaunchKernelGGL((icp_refine<true, true>), dim3(n), dim3(kIcpThreads), lds, stream, a, (const double*)d_part, n_blk);
    } else {
      hipLaunchKernelGGL((icp_refine<true, false>), dim3(n), dim3(kIcpThreads), lds, stream, a, (const double*)nullptr, 0);
    }
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

// Several (segment, target) jobs in ONE launch (icp_persist_multi).  Falls back to one launch_icp per job whenever
// the single launch cannot serve them all: a target whose index does not fit LDS, a segment beyond 4096 points,
// the point-to-plane metric, the pointmatcher history, more than kIcpMultiMax jobs.  Same results either way.
int launch_icp_multi(const IcpJob* jobs, int n_jobs, const pgp_icp_options* prm, hipStream_t stream) {
  if (n_jobs <= 0) return PGP_OK;
  for (int j = 0; j < n_jobs; ++j) {
    const IcpJob& q = jobs[j];
    if (!q.ctx || q.n < 0 || (q.n > 0 && (!q.d_src || !q.d_tgt || !q.d_T || q.n_src <= 0 || q.n_tgt <= 0))) {
      set_error("icp (multi): bad job %d", j);
      return PGP_EINVAL;
    }
    if (q.ctx->device != jobs[0].ctx->device) {
      set_error("icp (multi): the jobs' contexts live on different devices (%d, %d)", jobs[0].ctx->device, q.ctx->device);
      return PGP_EINVAL;
    }
  }
  IcpArgs a{};
  icp_option_args(prm, 1, &a);
  a.first_walk = 1;
  if (const char* v = getenv("PGP_ICP_FIRST_WALK")) a.first_walk = atoi(v) < 0 ? 0 : atoi(v);
  a.wait_ticks = icp_wait_ticks();
  bool one_launch = n_jobs >= 2 && n_jobs <= kIcpMultiMax && a.metric == 0 && a.smooth == 0 && prm->nn_search != 1 &&
                    prm->nn_search != 2 && !getenv("PGP_ICP_NN") && !getenv("PGP_ICP_PERSIST") && !getenv("PGP_ICP_SPLIT");
  if (const char* v = getenv("PGP_ICP_MULTI")) one_launch = one_launch && atoi(v) != 0;   // A/B knob: 0 = job by job
  // a context keeps ONE target index (d_icp_grid): two jobs with poses on the same context would have the second build
  // overwrite -- or reallocate -- the image the first job's descriptor points to.  Such jobs run one after the other
  // (each launch_icp builds its index in stream order behind the previous job's kernel).
  for (int j = 0; j < n_jobs && one_launch; ++j)
    for (int i = 0; i < j && one_launch; ++i)
      if (jobs[i].n > 0 && jobs[j].n > 0 && jobs[i].ctx == jobs[j].ctx) one_launch = false;
  IcpMulti mt{};
  int total = 0, max_src = 0, rc;
  size_t lds = 0;
  for (int j = 0; j < n_jobs && one_launch; ++j) {
    const IcpJob& q = jobs[j];
    mt.first[j] = total;
    if (q.n == 0) continue;
    if (q.n_src > kPiR * kIcpThreads) {
      one_launch = false;
      break;
    }
    IcpArgs aj{};
    bool fits = false;
    if ((rc = build_nn_index(q.ctx, q.d_tgt, q.n_tgt, q.n_src, &aj, &fits, stream, q.token)) != PGP_OK) return rc;
    if (!fits || !aj.nn_image_in_lds) {
      one_launch = false;
      break;
    }
    IcpTarget& tg = mt.tg[j];
    tg.src = q.d_src;
    tg.nn_image = aj.nn_image;
    tg.nn_vic = aj.nn_vic;
    tg.T = q.d_T - 16 * (ptrdiff_t)total;
    tg.energy = q.d_energy ? q.d_energy - total : nullptr;
    tg.iters = q.d_iters ? q.d_iters - total : nullptr;
    tg.nn = aj.nn;
    tg.n_src = q.n_src;
    tg.n_tgt = q.n_tgt;
    tg.k_trim = icp_trim_count(prm, q.n_src);
    total += q.n;
    max_src = q.n_src > max_src ? q.n_src : max_src;
    const size_t l = nn_lds_bytes(aj.nn.bytes, q.n_src, true);
    lds = l > lds ? l : lds;
  }
  if (one_launch && total > 0 && total <= 32768) {
    mt.n_jobs = n_jobs;
    for (int j = n_jobs; j <= kIcpMultiMax; ++j) mt.first[j] = total;
    // (jobs without poses keep first[j] = first[j + 1]: no workgroup ever selects them)
    for (int j = n_jobs - 1; j >= 0; --j)
      if (jobs[j].n == 0) mt.first[j] = mt.first[j + 1];
    a.n = total;
    a.wgs_per_pose = 1;
    a.energy = nullptr;   // per job, through the descriptor
    if ((rc = ensure_icp_attrs(jobs[0].ctx)) != PGP_OK) return rc;
    const int pir = max_src <= 2 * kIcpBase ? 2 : (max_src <= 3 * kIcpBase ? 3 : 4);
    void* params[] = {&a, &mt};
    PGP_HIP(hipLaunchKernel(multi_kernel(icp_trim_only(a), pir), dim3(total), dim3(kIcpThreads), params, lds, stream));
    PGP_HIP(hipGetLastError());
    return PGP_OK;
  }
  for (int j = 0; j < n_jobs; ++j) {
    const IcpJob& q = jobs[j];
    if ((rc = launch_icp(q.ctx, q.d_src, q.n_src, q.d_tgt, nullptr, q.n_tgt, q.d_T, q.n, prm, q.d_energy, q.d_iters, stream,
                         q.token)) != PGP_OK)
      return rc;
  }
  return PGP_OK;
}

}  // namespace pgp
