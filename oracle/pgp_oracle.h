/* oracle/pgp_oracle.h -- CPU restatement of the reference's pose-hypothesis scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (physimglobalpose_amd/, include/pgp.h) never does.
 *
 * Parity status: PARTIALLY PINNED.
 *   - nearest-neighbour query + kd-tree build: pinned bit-for-bit (returned index, including
 *     ties) against the reference's own kdtree.h compiled here (oracle/_ref, `make ref`);
 *   - Verify / WeightedVerify / rigid fit: the reference translation unit (match4pcsBase.cc)
 *     needs OpenCV + boost, absent from this image => unbuildable without stand-ins.  Their loop
 *     bodies are restated in oracle/ref_harness.cc on the reference's vendored Eigen (so the
 *     float evaluation order is Eigen's) and THIS file is pinned bit-for-bit against that.
 *   - The reference ships no golden vectors for the path.
 *
 * Citations: S4 = /root/reference/src/3rdparty/super4pcs/src/super4pcs, base.cc =
 * S4/algorithms/match4pcsBase.cc.
 *
 * Conventions: clouds are n x 3 row-major float; a transform is the reference's
 * Eigen::Matrix<float,4,4> memory image = 16 floats COLUMN-major (element (r,c) at [4*c+r]),
 * i.e. exactly one entry of `allTransforms` (base.cc:1468).
 */
#ifndef PGP_ORACLE_H
#define PGP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_kdtree orc_kdtree;

/* S4/accelerators/kdtree.h:355-370 (finalize), :522-538 (split), :560-641 (createTree);
 * 64 points per leaf, max depth 32 (kdtree.h:60-63). */
orc_kdtree* orc_kd_build(const float* xyz, int n);
void orc_kd_free(orc_kdtree* t);
int orc_kd_num_nodes(const orc_kdtree* t);
/* S4/accelerators/kdtree.h:394-459 (doQueryRestrictedClosestIndex): index of the closest point
 * with d2 <= sqdist (inclusive), -1 if none. Re-entrant (stack is local, unlike kdtree.h:311). */
int orc_kd_query(const orc_kdtree* t, const float q[3], float sqdist);
/* Semantic definition of the same query by exhaustive scan; ties -> lowest index. */
int orc_brute_query(const float* xyz, int n, const float q[3], float sqdist);

/* (mat * q.homogeneous()).head<3>() with Eigen's evaluation order (base.cc:1718,1750):
 * out_r = ((m_r0*q0 + m_r1*q1) + m_r2*q2) + m_r3, every operation rounded to float. */
void orc_transform_point(const float T[16], const float q[3], float out[3]);
/* mat.block<3,3>(0,0) * n (base.cc:1755): out_r = m_r0*n0 + (m_r1*n1 + m_r2*n2). */
void orc_rotate_normal(const float T[16], const float n[3], float out[3]);
/* (a-b).squaredNorm() (kdtree.h:423): dx*dx + (dy*dy + dz*dz). */
float orc_sqdist(const float a[3], const float b[3]);
/* a.dot(b) (base.cc:1756): a0*b0 + (a1*b1 + a2*b2). */
float orc_dot(const float a[3], const float b[3]);
/* The normal gate of base.cc:1756-1758 as a predicate on the dot product:
 * a = (float)((double)(acosf(d)*180.f)/M_PI); a = fminf(a, fabsf(180.f-a)); return a < gate. */
int orc_normal_gate(float dot, float gate_deg);

/* base.cc:1699-1731 (Verify).  use_nn: 0 = kd-tree (kd != NULL), 1 = brute force over P_xyz.
 * early_out != 0 keeps the reference's termination test against best_lcp (base.cc:1708,1725).
 * hit_ids (nullable, nQ ints) receives the NN index per model point (-1 = no inlier; entries
 * after an early-out are left untouched).  Returns good/nQ as float; *good_out = count. */
float orc_verify(const orc_kdtree* kd, const float* P_xyz, int nP, const float* Q_xyz, int nQ,
                 const float T[16], float delta, float best_lcp, int early_out,
                 int* good_out, int* hit_ids);

/* base.cc:1733-1766 (WeightedVerify).  registered (nullable, capacity nQ) receives the P ids
 * in model-point order, *n_registered their number.  gate_deg = 30 in the reference. */
float orc_weighted_verify(const orc_kdtree* kd, const float* P_xyz, const float* P_nrm,
                          const float* P_w, int nP, const float* Q_xyz, const float* Q_nrm, int nQ,
                          const float T[16], float delta, float gate_deg,
                          int* registered, int* n_registered);

/* base.cc:1885-1901 (verification loop of Perform_N_steps) over n_h transforms, no early-out
 * for mode 1 (weighted, the live operMode=1) and -- unless early_out is set -- none for mode 0
 * either (equal work per hypothesis).  scores[n_h]; *best_index = the index that
 * last satisfied `lcp > best` (strict), -1 if none; selected/n_selected (nullable) = the
 * running-best subsequence of base.cc:1903-1908.  threads > 1 splits hypotheses over OpenMP
 * threads (only legal without early-out; the bookkeeping is then replayed serially). */
void orc_score_batch(const orc_kdtree* kd, const float* P_xyz, const float* P_nrm, const float* P_w,
                     int nP, const float* Q_xyz, const float* Q_nrm, int nQ,
                     const float* T, int n_h, float delta, int mode, float gate_deg,
                     int early_out, int threads,
                     float* scores, int* best_index, int* selected, int* n_selected);

/* base.cc:242-268 (init): centre P on centroid(P); Q_search and Q_val on centroid(Q_search). */
void orc_center(float* P_xyz, int nP, float* Qs_xyz, int nQs, float* Qv_xyz, int nQv,
                float centroid_P[3], float centroid_Q[3]);

/* base.cc:1411-1488 (ComputeRigidTransformFromCongruentPair) + base.cc:1504-1614
 * (ComputeRigidTransformation, computeScale = false, max_angle < 0 as shipped) for ONE pair:
 * p[4][3] base points (centred P frame), q[4][3] congruent quad points (centred Q frame).
 * Returns 1 = transform pushed (T_centred: 16 floats col-major; pose: 16 doubles col-major),
 * 0 = rejected (non-orthogonal), 2 = degenerate input (the reference's `return kLargeNumber`).
 * The centred transform is bit-exact Eigen order; the de-centred translation uses the linear
 * part itself where the reference multiplies the polar factors rot*scale of an SVD
 * (computeRotationScaling, base.cc:1480-1481): equal up to float rounding (~1e-7). */
int orc_rigid_from_pair(const float* p, const float* q, const float centroid_P[3],
                        const float centroid_Q[3], float* T_centred, double* pose, float* rms_out);

/* ---- congruent-set extraction (S4/algorithms/super4pcs.cc, S4/pairCreationFunctor.h,
 * S4/accelerators/normalset.{h,hpp}) -------------------------------------------------------- */
typedef struct orc_cs orc_cs;
/* PairCreationFunctor::synch3DContent (pairCreationFunctor.h:102-138): unit-cube image of the
 * (centred) search model Q. */
orc_cs* orc_cs_create(const float* Qs_xyz, int n);
void orc_cs_free(orc_cs* s);
/* MatchSuper4PCS::ExtractPairs (super4pcs.cc:193-236) + PairCreationFunctor::process
 * (pairCreationFunctor.h:167-253) with the options the fork sets (no normal / colour /
 * translation / angle gates): every i > j with |(float)|q_i - q_j| - d| <= eps (evaluated in
 * double as in the reference), emitted as (j,i) then (i,j), in (i,j) lexicographic order.  The
 * reference finds them through an octree rasterisation whose emission order differs; parity is
 * at the SET level.  Returns the pair count (may exceed cap; only cap pairs are written). */
int orc_cs_extract_pairs(const orc_cs* s, float pair_distance, float eps, int* pairs_out, int cap);
/* MatchSuper4PCS::FindCongruentQuadrilaterals (super4pcs.cc:78-187) incl. IndexedNormalSet
 * <Point,3,7,float> (normalset.hpp:114-131 add, :166-214 cone query): base[4][3] = base_3D_
 * positions, P_pairs / Q_pairs flat (first,second) ids into the search model.  Quads are written
 * in the reference's order (sorted by (P-pair id, Q-pair id)); returns their number. */
int orc_cs_find_congruent(const orc_cs* s, const float* base, float invariant1, float invariant2,
                          float threshold, const int* P_pairs, int nP, const int* Q_pairs, int nQ,
                          int* quads_out, int cap);

/* Trimmed point-to-point ICP, the library's own statement of the algorithm behind
 * pcl::recognition::TrimmedICP::align / pcl::IterativeClosestPoint::align (call sites:
 * PPE/hypothesis_verification/mcts/UCTState.cpp:137-139,194; PPE/misc/utilities.cpp:666-703).
 * PARITY UNPINNED against PCL (not vendored, version unpinned: SURVEY 8c); this restatement pins
 * the HIP kernel.  Definition: see physimglobalpose_amd/csrc/icp.hip header.  T (16 floats,
 * col-major, source->target) is refined in place; returns the iteration count; *energy = final
 * mean squared distance of the selected pairs. */
int orc_icp(const float* src_xyz, int n_src, const float* tgt_xyz, int n_tgt, float* T,
            int max_iterations, float trim_fraction, float max_corr_dist, float energy_ratio,
            float* energy);

/* Pose distance of utilities::getPoseError (PPE/misc/utilities.cpp:514-548): test / gt are 4x4
 * col-major float images (what convertToMatrix :276-280 produces), sym = the object's symInfo in
 * degrees per axis (0 = none, 90 / 180 / 360).  *rot_err = mean |Euler angle| of test^-1 * gt in
 * degrees after the symmetry folds, *trans_err = translation distance. */
void orc_pose_error(const float test[16], const float gt[16], const float sym[3], float* rot_err,
                    float* trans_err);

/* HypothesisSelection::greedyClustering (PPE/hypothesis_verification/HypothesisSelection.cpp:66-115):
 * keep scores > accept_fraction * best_score, order by score descending (STABLE here: equal scores
 * stay in index order; the reference's std::sort leaves that order unspecified), then greedily keep
 * a candidate unless getPoseError(candidate, kept) < (rot_thresh, trans_thresh) for an earlier
 * kept one.  rep_out receives the kept hypothesis ids in output order; assign (nullable, n)
 * receives for every hypothesis the id of the representative that absorbed it (itself for a
 * representative, -1 if pruned).  Returns the number of representatives. */
int orc_greedy_cluster(const float* T, const float* scores, int n, float best_score, float accept_fraction,
                       const float sym[3], float rot_thresh, float trans_thresh, int* rep_out, int* assign);

/* Depth image -> cloud (PPE/misc/utilities.cpp:47-61 decode, Segmentation.cpp:219 mask,
 * utilities.cpp:190-206 back-projection), scan order.  image: raw 16-bit samples (raw16) or float
 * metres; mask nullable; K row-major 3x3.  Returns the number of points written (<= rows*cols). */
int orc_backproject(const void* image, int raw16, const unsigned char* mask, int rows, int cols,
                    const float K[9], double z_min, double z_max, float* xyz_out);

/* pcl::VoxelGrid as Segmentation.cpp:234-237 sets it up (published PCL 1.7 algorithm; points of a
 * voxel are added in index order): centroids in ascending voxel index; returns their number. */
int orc_voxel_grid(const float* xyz, int n, float leaf, float* out_xyz, int cap);

/* pcl::MovingLeastSquares (polynomial order 2, normals, no upsampling; Segmentation.cpp:239-246), PCL 1.7's
 * published algorithm restated (PCL is not vendored): smoothed points, un-normalised normals, curvatures and
 * input indices of the points with >= 3 neighbours, in input order; returns their number. */
int orc_mls(const float* xyz, int n, float radius, float* out_xyz, float* out_nrm, float* out_curv, int* out_idx,
            int cap);
/* c_dist_pose (max) and c_dist_pose_mean (sum), base.cc:1616-1655 */
void orc_pose_hausdorff(const float* hull, int n_hull, const float T1[16], const float T2[16], float* d_max,
                        float* d_sum);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
