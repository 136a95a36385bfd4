/* include/pgp.h -- C ABI of the MI355X-native pose-hypothesis scoring path (libpgp.so).
 *
 * This is the drop-in boundary for the hot path of cmitash/PhysimGlobalPose: everything the
 * reference computes between "a list of candidate transforms exists" and "every candidate has an
 * LCP score and the best one is known" (S4/algorithms/match4pcsBase.cc:1885-1901, with
 * S4 = src/3rdparty/super4pcs/src/super4pcs), plus the steps either side of it as they are
 * added (rigid fit from congruent pairs, ICP refinement).  Plain pointers and sizes only.
 *
 * Frames and layouts are the reference's own, so its containers can be passed without copies:
 *   - a cloud is n x 3 row-major float (pos() of each Point3D, S4/shared4pcs.h:61-111);
 *   - a transform is the memory image of Eigen::Matrix<float,4,4>: 16 floats, COLUMN-major
 *     (element (r,c) at [4*c + r]); `allTransforms.data()` (base.cc:1468) is such an array;
 *   - scoring happens in the CENTRED frames of Match4PCSBase::init (base.cc:242-268): the scene
 *     cloud minus centroid_P, the validation model minus centroid_Q.  pgp_center() does that.
 *
 * Every function returns 0 on success or a negative PGP_E* code; pgp_last_error() then holds a
 * message for the calling thread.  A context is bound to one HIP device; calls on one context
 * must not overlap (the reference is single-threaded: main.cpp:212), distinct contexts are
 * independent.  The *_device entry points queue their kernels on the caller's stream and return; a
 * context serves ONE such stream at a time (its workspaces are shared), and any later call on the
 * host-pointer call on the context -- pgp_set_model, pgp_set_scene, pgp_score_lcp ... -- first waits for
 * the queued work (a device synchronisation), so the arrays a queued kernel reads are never rewritten under it.
 * There is NO CPU fallback: without a usable HIP device pgp_create() fails.
 */
#ifndef PGP_H
#define PGP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgp_ctx pgp_ctx;

enum {
  PGP_OK = 0,
  PGP_EINVAL = -1,   /* bad argument (null pointer, negative size, NaN delta ...) */
  PGP_ENODEV = -2,   /* no usable HIP device / runtime error at create */
  PGP_EHIP = -3,     /* a HIP call failed */
  PGP_ESTATE = -4,   /* scene / model / index not set for this call */
  PGP_ENOMEM = -5
};

/* Scoring mode: which reference verifier is reproduced. */
enum {
  PGP_MODE_PLAIN = 0,    /* Match4PCSBase::Verify, base.cc:1699-1731, WITHOUT the early-out
                            (every hypothesis gets its true inlier count) */
  PGP_MODE_WEIGHTED = 1  /* Match4PCSBase::WeightedVerify, base.cc:1733-1766 (live operMode=1) */
};

int pgp_version(void);
const char* pgp_last_error(void);

/* Replaces the construction of the per-call MatchSuper4PCS object (S4/super4pcs_test.cc:102).
 * device_id < 0 selects the current HIP device. */
int pgp_create(pgp_ctx** out, int device_id);
int pgp_destroy(pgp_ctx* ctx);

/* Replaces Match4PCSBase::init's centring loop (base.cc:242-268): in place, float arithmetic in
 * the reference's order.  P is centred on centroid(P); Qs (search model) and Qv (validation
 * model) on centroid(Qs).  Pure host helper (O(n)); any of the clouds may be empty. */
int pgp_center(float* P_xyz, int nP, float* Qs_xyz, int nQs, float* Qv_xyz, int nQv,
               float centroid_P[3], float centroid_Q[3]);

/* Replaces the "priority based sampling" loop of Match4PCSBase::init (base.cc:317-340): the
 * per-scene-point weight orig_probabilities_[i] = probImg(row, col) with probImg = u16 / 10000
 * and (col,row) = int(K * (p_i + centroid_P)) / z, float arithmetic in the reference's order.
 * P_xyz is the CENTRED cloud; K is the 3x3 intrinsic matrix, row-major; img is the decoded
 * 16-bit probability image (rows x cols, row-major).  Pixels outside the image give weight 0
 * (the reference reads out of bounds there).  Pure host helper. */
int pgp_weights_from_image(const float* P_xyz, int n, const float centroid_P[3], const float K[9],
                           const unsigned short* img, int rows, int cols, float* weights);

/* The image rows pgp_weights_from_image reads for these points (the same arithmetic, so the answer is exact):
 * *row_min / *row_max over the points that fall inside a rows x cols image, -1 / -1 when none does.  A caller
 * that decodes the probability image while the clouds are being set up (the drop-in's file hand-off, base.cc:317)
 * can stop decoding after row_max.  Pure host helper. */
int pgp_image_rows_needed(const float* P_xyz, int n, const float centroid_P[3], const float K[9], int rows,
                          int cols, int* row_min, int* row_max);

/* Replaces `sampled_P_3D_ = P` + initKdTree() + orig_probabilities_ (base.cc:235,270,1046-1056,
 * 327-340): uploads the (centred) scene cloud and builds the device spatial index for inlier
 * radius `delta` (options_.delta, S4/super4pcs_test.cc:20).  nrm and weight may be NULL
 * (weights default to 1; PGP_MODE_WEIGHTED then needs normals and fails without them).
 * The index is a uniform grid of cell 0.85 delta with dilated candidate lists; its block table is a dense array
 * for scenes within 1024 cells per axis and 2^26 cells, a hashed table of the occupied blocks beyond (up to
 * 16 384 cells per axis at that cell size; pgp_index_info.sparse says which).  Answers are identical.
 * Host pointers, synchronous. */
int pgp_set_scene(pgp_ctx* ctx, const float* xyz, const float* nrm, const float* weight, int n,
                  float delta);

/* Replaces the weights of the resident scene (n must be the scene's point count) without touching
 * its points or index: the per-point probabilities come from an image (base.cc:317-340) that a caller
 * can still be decoding while pgp_set_scene builds the index from the coordinates.  Host pointer,
 * synchronous.  pgp_multi_set_scene_weights: the same on every device of a group. */
int pgp_set_scene_weights(pgp_ctx* ctx, const float* weight, int n);

/* Replaces `validation_Q_3D = Q_validation` (base.cc:237).  Host pointers, synchronous. */
int pgp_set_model(pgp_ctx* ctx, const float* xyz, const float* nrm, int n);

/* Replaces the verification loop of Perform_N_steps (base.cc:1885-1901) for n_h transforms.
 *   scores[n_h]     : lcp per hypothesis (inliers / nQ, or weighted sum / nQ), as allPose[i].second
 *   counts[n_h]     : (nullable) integer inlier count per hypothesis (for weighted mode: the
 *                     number of registered points)
 *   best_index      : (nullable) what best_lcp_index ends as: the lowest index attaining the
 *                     maximum score if that maximum is > 0, else -1 (base.cc:1891,309)
 *   best_score      : (nullable) best_LCP_
 * gate_deg is the normal-agreement gate of base.cc:1758 (30 in the reference; ignored in plain
 * mode).  Host pointers, synchronous. */
int pgp_score_lcp(pgp_ctx* ctx, const float* T, int n_h, int mode, float gate_deg,
                  float* scores, int* counts, int* best_index, float* best_score);

/* Same computation with DEVICE pointers, enqueued on `stream` (a hipStream_t; NULL = default
 * stream) and not synchronised: for callers that keep the hypothesis batch resident (bench.py,
 * multi-GPU sharding).  d_best (nullable) receives 2 ints: {best_index, float bits of best
 * score}.  No allocation happens inside when n_h <= the capacity reserved by
 * pgp_reserve(); otherwise PGP_ESTATE.
 * A context serves ONE stream at a time: its workspaces (partials, arg-max key, ticket) are
 * shared by every queued call, so the *_device calls of one context must all go to the same
 * stream (or be ordered by the caller's events); different contexts are independent.
 * pgp_set_scene / pgp_set_model / pgp_set_search_model / a growing pgp_reserve wait for
 * everything queued on the device (hipDeviceSynchronize) before they replace arrays a queued
 * launch may still be reading.
 * Weighted mode: hypotheses within 4.1 * 2^-24 * sqrt(nQ) * best_score of the maximum (ten standard
 * deviations of the reference's own summation error; 6.2e-6 at 5000 model points) are re-summed on
 * the device in the reference's order (sequential float adds in model order, base.cc:1759) and
 * their score entries overwritten with that value, so best_index is the reference's also under
 * near-ties; all other weighted scores carry the library's fixed summation tree (within 2e-6 of
 * the reference at those sizes). */
int pgp_reserve(pgp_ctx* ctx, int max_hypotheses);
int pgp_score_lcp_device(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg,
                         float* d_scores, int* d_counts, int* d_best, void* stream);

/* Arg-max over a COMPLETE score vector assembled from several partial calls (the slices of
 * several devices after the score all-reduce, or several batches of one object): publishes
 * d_best = {best_index, float bits of best score} with the rule of base.cc:1891 (lowest index of
 * the maximum, -1 when the maximum is not > 0) and, in weighted mode, the exact near-tie
 * settlement described above -- d_T must hold ALL n_h transforms and the context the clouds the
 * scores were computed on; settled entries of d_scores are overwritten with the reference-order
 * value.  Device pointers, enqueued on `stream`. */
int pgp_settle_best_device(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg,
                           float* d_scores, int* d_best, void* stream);

/* Replaces `registered_indices = temp_registered_indices` (base.cc:1897): the scene-point ids
 * matched under ONE transform, in model-point order.  ids has capacity nQ; *n receives the
 * count.  In plain mode every inlier's nearest scene point is reported. */
int pgp_registered(pgp_ctx* ctx, const float* T16, int mode, float gate_deg, int* ids, int* n);

/* Replaces Match4PCSBase::getRegisteredModel (base.cc:347-375; defined but not called by
 * ComputeTransformation): the scene ids registered by the points of ANOTHER cloud with normals
 * (sampled_Q_3D_, the search model) under one transform, with the gate on the DIRECTED normal angle
 * (acos(dot) * 180 / pi < gate_deg; the fold of WeightedVerify is commented out there, :368).
 * q_xyz / q_nrm: n x 3; ids (capacity n) in cloud order; *n_ids their number.  Host pointers. */
int pgp_registered_model(pgp_ctx* ctx, const float* T16, const float* q_xyz, const float* q_nrm, int n,
                         float gate_deg, int* ids, int* n_ids);

/* Opt-in: plain-mode scores and counts as the reference's Verify returns them, WITH its early termination
 * (base.cc:1708,1725-1727: a hypothesis that can no longer beat the running best stops, and its
 * allPose[i].second is the fraction counted so far -- an order-dependent lower bound).  Off (default):
 * every hypothesis is counted completely.  Best index and best score are the same either way.  Costs up to
 * one more pass over the batch.  Applies to pgp_score_lcp / pgp_score_lcp_device in PGP_MODE_PLAIN;
 * pgp_verify_early_out_device applies it to a complete vector assembled elsewhere (d_scores / d_counts hold
 * the TRUE values of all n_h hypotheses, e.g. after the all-reduce of a device group). */
int pgp_set_verify_early_out(pgp_ctx* ctx, int on);
int pgp_verify_early_out_device(pgp_ctx* ctx, const float* d_T, int n_h, float* d_scores, int* d_counts, void* stream);

/* Weighted mode, the running-best LIST (base.cc:1891-1908, what the drop-in returns as hypothesisSet): with
 * pgp_set_exact_records(ctx, 1) every later weighted scoring call on the context also re-scores, exactly as
 * the reference sums (sequential float adds in model order), every hypothesis whose score comes within the
 * summation tolerance of the running maximum before it -- the records and whatever could displace or tie one;
 * pgp_running_best over the returned scores is then the reference's list, entry for entry.  Costs three
 * small launches per call (~60 us at 4096 hypotheses x 5000 model points); off by default (the best pose is
 * exact either way).  Call it after pgp_set_model (it reserves a workspace of 128 rows of model size).
 * pgp_settle_records_device does the same for a score vector assembled elsewhere (the slices of several
 * devices), like pgp_settle_best_device. */
int pgp_set_exact_records(pgp_ctx* ctx, int on);

/* Exact distance ties.  KdTree::doQueryRestrictedClosestIndex accepts a candidate when `sqdist <= cl_dist`
 * (kdtree.h:424): of several scene points at exactly the same float distance from a query it returns the one its
 * descent visits last (the query's side of every split plane first, points of a leaf in the order the build left
 * them).  This library's index breaks such a tie by the lowest scene index instead -- the registered point, hence
 * the normal gate and the weight of that model point, can then differ from the reference's (about one query in
 * 10^7 with two neighbours on real clouds; every query near a DUPLICATED scene point).  With on != 0, every later
 * pgp_set_scene / pgp_set_scene_device also builds the reference's tree on the host (its own splits, partition
 * and leaf sizes, kdtree.h:522-641; ~5 ms per 50 000 points) and the scoring paths ask it whenever two different
 * candidates share the minimal distance: scores, registered points and records then follow the reference on ties
 * as well.  Default off.  Takes effect at the next scene set-up. */
int pgp_set_exact_ties(pgp_ctx* ctx, int on);
int pgp_settle_records_device(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg, float* d_scores,
                              void* stream);

/* Running-best subsequence of base.cc:1891-1908 over a score vector (host helper): writes the
 * indices i with scores[i] > max(scores[0..i-1], 0) to selected (capacity n_h). */
int pgp_running_best(const float* scores, int n_h, int* selected, int* n_selected);

/* Introspection for DESIGN.md / bench.py: sizes of the device index of the current scene. */
typedef struct {
  int n_scene, n_model;
  int grid_nx, grid_ny, grid_nz;
  float cell_size, delta;
  long long n_cells, n_candidates;      /* entries in the dilated per-cell candidate lists */
  long long n_occupied;                 /* cells with a non-empty candidate list */
  long long bytes_index;                /* query-time index: words + occupied offsets + lists */
  float build_ms;                       /* device time of the last index build */
  int sparse;                           /* 1: hashed table of the occupied 4x4x2 blocks, 0: dense block array */
  long long n_blocks;                   /* blocks in the table (sparse) or in the dense array */
} pgp_index_info;
int pgp_get_index_info(pgp_ctx* ctx, pgp_index_info* info);

/* Replaces `sampled_Q_3D_ = Q` (base.cc:236): the sparse search model whose points the
 * congruent quads index.  Host pointer, synchronous. */
int pgp_set_search_model(pgp_ctx* ctx, const float* xyz, int n);

/* ---- Step 1 of Perform_N_steps: stochastic base selection (base.cc:600-792) on the device -------
 * pgp_set_ppf_map replaces the node's hand-over of the model's pair-feature table
 * (std::map<std::vector<int>, std::vector<std::pair<int,int>>> PPFMap, PPE/data_layer/Objects.cpp:31-49;
 * parameter of getProbableTransformsSuper4PCS): keys[n_keys][4] = the discretised features
 * (computePPF + approximate_bin, base.cc:582-598,150-160), counts[n_keys] (nullable) and
 * pairs[sum(counts)][2] (nullable) the pair list of every key in key order (ids into the search
 * model).  The first occurrence of a key counts, as std::map::insert.  The keys become a device
 * hash set; the ratio thresholds that stand in for atan2f are measured on the host's libm here. */
int pgp_set_ppf_map(pgp_ctx* ctx, const int* keys, const int* counts, const int* pairs, int n_keys);

/* Replaces n_attempts calls of Match4PCSBase::SelectQuadrilateralStoCS (base.cc:600-792; the
 * reference seeds a fresh engine per call, so attempts are independent) + TryQuadrilateral
 * (:415-464) in ONE launch, on the scene set by pgp_set_scene (positions, normals, weights =
 * orig_probabilities_) and the table set by pgp_set_ppf_map.  u[n_attempts][4]: the uniform variates
 * in [0,1) of the four std::discrete_distribution draws, taken by the CALLER from its own engine
 * (std::generate_canonical<double, 53>, what discrete_distribution::operator() consumes).
 * ids[n_attempts][4]: scene ids of the base in TryQuadrilateral's order; invariants[n_attempts][2];
 * status[n_attempts]: 1 = base found, 0 = the reference would have returned false (no candidate
 * left for the 2nd / 3rd / 4th point).  Host pointers, synchronous. */
int pgp_select_bases(pgp_ctx* ctx, const double* u, int n_attempts, int* ids, float* invariants, int* status);
/* The same, and rows[n_attempts][2]: the rows of the pair-feature table under which ExtractCongruentSet finds pairs1 and
 * pairs6 of the base -- PPFMap->find(computePPF(ids[0], ids[1])) and (ids[2], ids[3]), base.cc:1970-1981; -1: not a key --
 * computed where the base is selected, for pgp_find_congruent_batch_rows below (one launch and one round trip less per
 * object than asking for them again). */
int pgp_select_bases_rows(pgp_ctx* ctx, const double* u, int n_attempts, int* ids, float* invariants, int* status, int* rows);
/* pgp_select_bases_rows in two halves, for a caller with host work of its own to do meanwhile (the drop-in hashes the
 * validation model while the selection's ~0.1 ms kernel runs): _begin copies the variates, queues the launch and returns;
 * _end waits and returns what pgp_select_bases_rows returns (rows nullable).  Between the two, only calls that leave the
 * scene, its weights and the pair-feature table alone (pgp_set_model); any other selection call drops the begun one. */
int pgp_select_bases_rows_begin(pgp_ctx* ctx, const double* u, int n_attempts);
int pgp_select_bases_rows_end(pgp_ctx* ctx, int* ids, float* invariants, int* status, int* rows);

/* The pieces of the above, for parity tests and for callers that keep their own sampling loop:
 * pgp_ppf_features: computePPF for m (i, j) pairs of scene ids -> features[m][4] (-1: not a key,
 *   NaN inputs) and rows[m] (nullable): index of the key in the table or -1 (PPFMap->find);
 * pgp_stocs_stage_weights: ONE weighting loop of SelectQuadrilateralStoCS (stage 2: base.cc:625-652,
 *   3: :662-699, 4: :713-769) for given base points; cur[nP] in = curr_probabilities_ entering the
 *   loop (stage 2: orig_probabilities_), out = the values the reference leaves (divided by the
 *   sequential float sum when a candidate is present); *sum, *present as in the reference;
 * pgp_base_invariants: TryQuadrilateral for m bases given as scene ids[m][4] (reordered in place),
 *   invariants[m][2], ok[m] (1, or -1 for a bad id). */
int pgp_ppf_features(pgp_ctx* ctx, const int* pairs, int m, int* features, int* rows);
int pgp_stocs_stage_weights(pgp_ctx* ctx, int stage, int base1, int base2, int base3, float* cur, float* sum,
                            int* present);
int pgp_base_invariants(pgp_ctx* ctx, int* ids, int m, float* invariants, int* ok);

/* Replaces ComputeRigidTransformFromCongruentPair + ComputeRigidTransformation
 * (base.cc:1411-1488, 1504-1614) for n congruent pairs.  base_ids[n][4] index the scene cloud
 * (base_3D_ ids), quad_ids[n][4] the search model; only the first three of each are used, as in
 * the reference.  Outputs (host, caller-owned, any of pose/rms nullable):
 *   T[n][16]     centred transform, column-major float = what allTransforms receives (:1468)
 *   pose[n][16]  de-centred transform as double, column-major = allPose[i].first (:1484)
 *   status[n]    1 pushed | 0 rejected (non-orthogonal / rms) | 2 degenerate input | -1 bad index
 *   rms[n]
 * Entries with status != 1 hold NaN transforms (they score 0); the reference simply does not
 * push them -- compact by status to reproduce its lists. */
int pgp_rigid_from_congruent(pgp_ctx* ctx, const int* base_ids, const int* quad_ids, int n,
                             const float centroid_P[3], const float centroid_Q[3], float* T,
                             double* pose, int* status, float* rms);
/* Same with device pointers on `stream` (ids as int4 arrays; d_pose / d_rms nullable). */
int pgp_rigid_from_congruent_device(pgp_ctx* ctx, const int* d_base_ids, const int* d_quad_ids, int n,
                                    const float centroid_P[3], const float centroid_Q[3], float* d_T,
                                    double* d_pose, int* d_status, float* d_rms, void* stream);

/* Replaces MatchSuper4PCS::ExtractPairs (S4/algorithms/super4pcs.cc:193-236; operMode 0) over the
 * search model set by pgp_set_search_model: every ordered pair (a,b), a != b, with
 * | |q_a - q_b| - pair_distance | <= eps (no normal / colour / angle gates, as the fork sets
 * them), written as (j,i),(i,j) for i > j in (i,j) order.  pairs: cap x 2 ints; *n_pairs = the
 * full count (may exceed cap).  Same SET as the reference's octree functor. */
int pgp_extract_pairs(pgp_ctx* ctx, float pair_distance, float eps, int* pairs, int cap, int* n_pairs);

/* Replaces MatchSuper4PCS::FindCongruentQuadrilaterals (super4pcs.cc:78-187) for one base:
 * base[4][3] = positions of base_3D_ (row-major), invariants of the base, threshold =
 * distance_factor * delta (base.cc:1989-1990), P_pairs / Q_pairs = the two pair lists (flat
 * (first,second) ids into the search model: pairs1 / pairs6 of base.cc:1970-1981).  quads:
 * cap x 4 ints in the reference's order (its std::set of (P-pair id, Q-pair id));
 * *n_quads = the full count. */
int pgp_find_congruent(pgp_ctx* ctx, const float* base, float invariant1, float invariant2, float threshold,
                       const int* P_pairs, int nP, const int* Q_pairs, int nQ, int* quads, int cap,
                       int* n_quads);

/* Replaces Match4PCS::FindCongruentQuadrilaterals (S4/algorithms/4pcs.cc:61-103; the classic 4PCS
 * matcher, not instantiated by the node): for every Q-pair i, every P-pair id whose invariant point
 * e1 = p1 + invariant1 (p2 - p1) lies at SQUARED distance < threshold (strict; kdtree.h:491) of
 * e2 = q1 + invariant2 (q2 - q1) emits (P_pairs[id / 2], Q_pairs[i]) -- `id / 2` as the reference
 * writes it.  Quads come out in (i, id) order (the reference's kd-tree leaves the order inside one i
 * unspecified).  Pair lists as in pgp_find_congruent; *n_quads = full count. */
int pgp_find_congruent_4pcs(pgp_ctx* ctx, float invariant1, float invariant2, float threshold, const int* P_pairs,
                            int nP, const int* Q_pairs, int nQ, int* quads, int cap, int* n_quads);

/* Replaces the loop `for (auto base_it: baseSet) ExtractCongruentSet(base_it)` of Perform_N_steps
 * (base.cc:1855-1874 -> :1929-1993, operMode 1) for ALL bases of an object in one pass: for base b,
 * pairs1 = PPFMap[computePPF(id0, id1)], pairs6 = PPFMap[computePPF(id2, id3)] are looked up in the
 * device table of pgp_set_ppf_map (which must have been given the pair lists), then
 * FindCongruentQuadrilaterals(invariant1, invariant2, threshold, ...) as in pgp_find_congruent.
 * base_ids[n_bases][4]: scene ids in TryQuadrilateral's order; base_xyz[n_bases][4][3]: their
 * (centred) positions; invariants[n_bases][2].  n_quads[n_bases] receives every base's full quad
 * count; the sorted quad lists stay ON THE DEVICE until the next call:
 *   pgp_congruent_batch_quads copies the picked quads (picks[m][2] = (base, j): the j-th quad of
 *     that base in the reference's order) to the host, quads[m][4];
 *   pgp_congruent_batch_fit runs ComputeRigidTransformFromCongruentPair (as
 *     pgp_rigid_from_congruent) on the picked (base, quad) pairs without the quads leaving the
 *     device: the reference's sampling of <= 100 quads per base (base.cc:1858-1872) decides the picks. */
int pgp_find_congruent_batch(pgp_ctx* ctx, const int* base_ids, const float* base_xyz, const float* invariants,
                             int n_bases, float threshold, int* n_quads);
/* pgp_find_congruent_batch for a caller that already holds rows[n_bases][2] (pgp_select_bases_rows, or pgp_ppf_features of
 * the edges (ids 0-1), (ids 2-3)): same quads, same order.  A row outside [-1, rows of the table) is PGP_EINVAL. */
int pgp_find_congruent_batch_rows(pgp_ctx* ctx, const int* base_ids, const float* base_xyz, const float* invariants,
                                  const int* rows, int n_bases, float threshold, int* n_quads);
int pgp_congruent_batch_quads(pgp_ctx* ctx, const int* picks, int m, int* quads);
int pgp_congruent_batch_fit(pgp_ctx* ctx, const int* picks, const int* base_ids, int m, const float centroid_P[3],
                            const float centroid_Q[3], float* T, double* pose, int* status, float* rms);

/* pgp_congruent_batch_fit and the verification of its fits WITHOUT a round trip of the transforms: the fits stay in
 * HBM, every pick is scored there ([Weighted]Verify, base.cc:1885-1901; a fit the reference does not push -- status
 * != 1, base.cc:1467-1485 -- scores 0 and can therefore never enter the strict-`>` walk), and only scores[m],
 * status[m] and the best {index, score} come back.  With pgp_set_exact_records the scores at the walk's decisions are
 * the reference's own sums.  pgp_congruent_batch_fetch then returns the float transform and the double pose of the few
 * picks the caller keeps (the running-best list, the best pose): index[k] -> T[k][16], pose[k][16], either nullable. */
int pgp_congruent_batch_fit_score(pgp_ctx* ctx, const int* picks, const int* base_ids, int m, const float centroid_P[3],
                                  const float centroid_Q[3], int mode, float gate_deg, float* scores, int* status,
                                  int* best_index, float* best_score);
int pgp_congruent_batch_fetch(pgp_ctx* ctx, const int* index, int k, float* T, double* pose);
/* pgp_congruent_batch_fit_score and everything the caller of Perform_N_steps reads afterwards, in ONE call with ONE copy
 * back and ONE synchronisation: the running-best walk of the verification loop (base.cc:1891-1908, what pgp_running_best
 * computes on the host -- run with pgp_set_exact_records for the reference's own decisions) on the device, the poses it
 * keeps, the best pose (base.cc:1787-1790) and the scene points the best pose registers (pgp_registered).
 *   n_list: the number of records; list_index / list_score / list_T [.][16] / list_pose [.][16] hold the first
 *     min(n_list, list_cap) of them in walk order (list_cap <= 4096; n_list > list_cap: call pgp_congruent_batch_fetch
 *     for the rest -- the scores stay on the device, the fits too);
 *   n_pushed (nullable): the number of fits the reference pushes (status == 1, base.cc:1467-1485);
 *   best_index = -1: no hypothesis scored above 0 -- best_T / best_pose / registered are then left alone, n_registered = 0;
 *   registered: room for every model point (pgp_set_model's n).  Nullable outputs: n_pushed, best_*, registered. */
int pgp_congruent_batch_fit_score_list(pgp_ctx* ctx, const int* picks, const int* base_ids, int m, const float centroid_P[3],
                                       const float centroid_Q[3], int mode, float gate_deg, int list_cap, int* n_list,
                                       int* list_index, float* list_score, float* list_T, double* list_pose, int* n_pushed,
                                       int* best_index, float* best_score, float* best_T, double* best_pose, int* registered,
                                       int* n_registered);

/* The same call with the sampling of the quads (base.cc:1858-1866: at most 100 random quads per base) done ON THE DEVICE, where
 * the quad counts are: no picks to draw on the host between the congruent sets and the fits, none to upload.  The draw is a
 * function of (seed, base, the base's quad count) alone -- a counter-based splitmix64 generator per base feeding Floyd's subset
 * algorithm (max_per_base steps whatever the count, every subset equally likely), handed out in ascending order; a base with
 * fewer quads hands out all of them --, so the bases are drawn side by side, and pgp_sample_quads is the same function on the
 * host (the reference draws from rand() seeded from the clock: any uniform sample of max_per_base distinct quads is its
 * behaviour).  Works on the batch pgp_find_congruent_batch[_rows]
 * left resident; base_ids[n_bases][4] as there.  picks_out (nullable, room for n_bases x max_per_base x 2) / n_picks
 * (nullable) return what was drawn: the picks a pgp_congruent_batch_fit_score_list call would need for the same result.
 * 1 <= max_per_base <= 128. */
int pgp_congruent_batch_sample_fit_score_list(pgp_ctx* ctx, unsigned long long seed, int max_per_base, const int* base_ids,
                                              const float centroid_P[3], const float centroid_Q[3], int mode, float gate_deg,
                                              int list_cap, int* n_list, int* list_index, float* list_score, float* list_T,
                                              double* list_pose, int* n_pushed, int* best_index, float* best_score, float* best_T,
                                              double* best_pose, int* registered, int* n_registered, int* picks_out, int* n_picks);
/* Host helper: the picks that call draws, from the quad counts alone.  picks (nullable: count only) [sum][2] = (base, quad). */
int pgp_sample_quads(unsigned long long seed, const int* n_quads, int n_bases, int max_per_base, int* picks, int* n_picks);

/* ICP refinement.  Replaces the inner loop behind pcl::recognition::TrimmedICP::align
 * (PPE/hypothesis_verification/mcts/UCTState.cpp:137-139,194; PPE/misc/utilities.cpp:666-676) and
 * pcl::IterativeClosestPoint::align (utilities.cpp:697-703; PPE/data_layer/SceneCfg.cpp:101,135-141)
 * for a batch of n initial guesses at once.  PCL is not vendored in the reference, so the
 * arithmetic is this library's own statement of the algorithm: exact nearest neighbour
 * (an index over the static target, identical to an exhaustive search: smallest d2, then lowest
 * index), keep the |trim*n_src| closest pairs (or those within max_corr_dist), Horn's
 * closed-form rigid update, repeat while mean-squared-distance / previous < energy_ratio. */
typedef struct {
  int max_iterations;    /* <= 0: 100 (utilities.cpp:698); TrimmedICP itself is unbounded */
  float trim_fraction;   /* (0,1]: k = (int)|trim * n_src| (UCTState.cpp:176,194); 1 = use all */
  float max_corr_dist;   /* > 0: drop pairs farther than this instead of trimming (State.cpp:139) */
  float energy_ratio;    /* setNewToOldEnergyRatio: 1.0 at the call sites (UCTState.cpp:139) */
} pgp_icp_params;

/* src: points that are moved (the scene segment in the reference), tgt: the cloud searched for
 * neighbours (the model).  T[n][16]: column-major float, source frame -> target frame; initial
 * guesses in, refined transforms out (UCTState.cpp:184-203 inverts the pose before and after).
 * energy[n] (nullable): final mean squared distance of the selected pairs; iters[n] (nullable).
 * Host pointers, synchronous. */
int pgp_icp_refine(pgp_ctx* ctx, const float* src_xyz, int n_src, const float* tgt_xyz, int n_tgt,
                   float* T, int n, const pgp_icp_params* params, float* energy, int* iters);
/* Device pointers: d_src / d_tgt are float4 arrays {x,y,z,-}; enqueued on `stream`.  Default
 * (target index fits LDS, n_src <= 4096): the index is built (the call synchronises once, for the
 * target's bounding box) and ONE launch runs every iteration of every pose.  Otherwise the
 * iterations are driven from the host (many workgroups per pose) and the call synchronises the
 * stream every four iterations to test for convergence.  While few poses are in flight (n x 4 or
 * n x 2 <= the device's compute units) that ONE launch has 4 or 2 workgroups per pose sharing the search, one per
 * compute unit (a plain launch of a grid that fits the device -- PGP_COOPERATIVE_LAUNCH=1: hipLaunchCooperativeKernel,
 * whose queue makes the hardware scheduler time-slice the GPU between processes; not on a stream that is being
 * captured; PGP_ICP_WGS=1 switches the sharing off).  Should the
 * workgroups of a pose ever fail to meet (another process holding the GPU's compute units for seconds), the pose's
 * first workgroup searches every query again and finishes the pose alone, inside the same launch -- the caller always
 * receives refined transforms, the same bits.  Clustered launches of one process never overlap on a device.
 * The scene-sized capped form in one launch (n_src > 4096, a correspondence cap, no trimming, n <= 64) reports a pose
 * whose work did not arrive within the same clock bound with d_iters = -1 and leaves its transform where it was; the
 * host-pointer calls then redo the job with the host-driven iterations themselves.
 * Checker paths, same results:
 * PGP_ICP_NN=scan (exhaustive search), PGP_ICP_PERSIST=0 (index, host-driven iterations),
 * PGP_ICP_SPLIT=0/1 (the exhaustive persistent / host-driven kernels). */
int pgp_icp_refine_device(pgp_ctx* ctx, const float* d_src4, int n_src, const float* d_tgt4, int n_tgt,
                          float* d_T, int n, const pgp_icp_params* params, float* d_energy,
                          int* d_iters, void* stream);

/* SEVERAL (segment, target) pairs refined by ONE launch: the children of an MCTS expansion belong to different objects
 * (PPE/hypothesis_verification/mcts/UCTSearch.cpp:200-266 -> UCTState.cpp:121-204) and the node's object loop
 * (PPE/data_layer/SceneCfg.cpp:379-402) refines the candidates of every object of a frame -- through
 * pgp_icp_refine_device that is one launch per object, each wanting the whole chip.  Every job names the context
 * that keeps its target's index (pgp_icp_target_token applies per context), its clouds and its own transform /
 * energy / iteration arrays; all contexts on one device.  Results are bit-identical to one pgp_icp_refine_device
 * call per job -- which is also what happens when the single launch cannot serve the jobs (more than 8 of them, a
 * target whose index does not fit a compute unit's LDS, a segment beyond 4096 points, or two jobs that name the SAME
 * context: a context keeps one target index, so give every job of a launch its own context when the single launch
 * matters).  Enqueued on `stream`. */
typedef struct {
  pgp_ctx* ctx;
  const float* d_src4;   /* float4 {x,y,z,-} [n_src] */
  int n_src;
  const float* d_tgt4;   /* float4 {x,y,z,-} [n_tgt] */
  int n_tgt;
  float* d_T;            /* [n][16] in/out */
  int n;
  float* d_energy;       /* [n], nullable */
  int* d_iters;          /* [n], nullable */
} pgp_icp_job;
int pgp_icp_refine_multi_device(const pgp_icp_job* jobs, int n_jobs, const pgp_icp_params* params, void* stream);

/* The k best-scoring hypotheses of a scored batch, in HBM: d_T_out[k][16] = their transforms in descending score
 * order (equal scores: lower index first -- a stable argsort of -score), inverted as rigid transforms {R^T, -R^T t}
 * when invert != 0 (UCTState.cpp:184-185 hands `pose.inverse()` to ICP); d_index_out[k] (nullable) = their indices,
 * *d_n_out = how many of the k have a score > 0 (the rest: index -1, identity).  The hand-off
 * HypothesisSelection.cpp:248-257 -> UCTState::performTrICP without a trip to the host.  Enqueued on `stream`. */
int pgp_select_top_device(pgp_ctx* ctx, const float* d_T, const float* d_scores, int n, int k, int invert,
                          float* d_T_out, int* d_index_out, int* d_n_out, void* stream);

/* The index over the target is built per call unless the caller vouches that the target has not
 * changed: after pgp_icp_target_token(ctx, token != 0), *_device calls with the same (d_tgt pointer,
 * n_tgt, token) reuse the resident index (the reference builds TrimmedICP's search structure once per
 * model, UCTState.cpp:137-139).  token 0 (default) = rebuild on every call.  The host-pointer calls do
 * this by themselves with a hash of the target's coordinates. */
int pgp_icp_target_token(pgp_ctx* ctx, unsigned long long token);

/* The other ICP forms of the call sites, on the same kernels (csrc/icp.hip).  Fields <= 0 / < 0 switch
 * a rule off as noted; pgp_icp_default_options() fills the TrimmedICP form of pgp_icp_params.
 *   greedy_bfs/State.cpp:139-142   pcl::IterativeClosestPoint, setMaxCorrespondenceDistance(max_corr),
 *                                  50 iterations, setTransformationEpsilon(1e-8):
 *                                  {50, 1, max_corr, 0, 0, 1e-8f, 0, 1e-12f, 0, 0, 0, 0}
 *   misc/utilities.cpp:697-703     pcl::IterativeClosestPoint, 100 iterations, PCL defaults:
 *                                  {100, 1, 0, 0, 0, 0.f, 0, 1e-12f, 0, 0, 0, 0}
 *   misc/utilities.cpp:709-739     pcl::IterativeClosestPointWithNormals: the same with error_metric 1
 *   misc/utilities.cpp:744-838     libpointmatcher chain: {100, 0.75f, 0, 0, 0, -1, 0, -1, 0.001f, 0.005f, 4, 0}
 *   data_layer/SceneCfg.cpp:135-141  table ICP on scene-sized clouds: max_corr 0.01, 50 iterations,
 *                                  transformation epsilon 1e-9 (nn_search 0 picks the grid search there) */
typedef struct {
  int max_iterations;            /* <= 0: 100 */
  float trim_fraction;           /* (0,1]: keep the |trim * n_src| closest pairs; 1 (or <= 0) = all */
  float max_corr_dist;           /* > 0: pairs farther than this are dropped (instead of trimming) */
  float energy_ratio;            /* > 0: go on only while E / E_old < ratio (TrimmedICP); <= 0: off */
  int error_metric;              /* 0 point-to-point (Horn closed form), 1 point-to-plane (linearised
                                    least squares; needs target normals) */
  float transformation_epsilon;  /* >= 0: stop when an iteration's update has cos(angle) >= 1 - eps and
                                    |t|^2 <= eps (pcl DefaultConvergenceCriteria); < 0: off */
  float relative_mse;            /* > 0: stop when |E - E_old| / E_old < this; <= 0: off */
  float absolute_mse;            /* >= 0: stop when |E - E_old| < this (PCL: 1e-12); < 0: off */
  float min_diff_rot;            /* > 0 with min_diff_trans > 0: libpointmatcher's                  */
  float min_diff_trans;          /*   DifferentialTransformationChecker on the last smooth_length    */
  int smooth_length;             /*   iterations (1..8)                                              */
  int nn_search;                 /* 0 auto, 1 exhaustive scan, 2 uniform grid (needs max_corr_dist > 0),
                                    3 exact index of the static target (its image in LDS up to ~6000 points,
                                    read from L2 beyond; fails above 65 535 points); auto = 3 when it applies,
                                    else 2 for capped scene-sized searches, else 1.  All return identical results. */
} pgp_icp_options;
int pgp_icp_default_options(pgp_icp_options* opt);
/* tgt_nrm: n_tgt x 3 unit normals of the target (nullable unless error_metric is 1).  Otherwise as
 * pgp_icp_refine / pgp_icp_refine_device (d_tgt_n4: float4 {nx,ny,nz,-}). */
int pgp_icp_refine_ex(pgp_ctx* ctx, const float* src_xyz, int n_src, const float* tgt_xyz, const float* tgt_nrm,
                      int n_tgt, float* T, int n, const pgp_icp_options* opt, float* energy, int* iters);
int pgp_icp_refine_ex_device(pgp_ctx* ctx, const float* d_src4, int n_src, const float* d_tgt4,
                             const float* d_tgt_n4, int n_tgt, float* d_T, int n, const pgp_icp_options* opt,
                             float* d_energy, int* d_iters, void* stream);

/* Segment pre-processing in front of the path (PPE/hypothesis_generation/ObjectPoseCandidateSet.cpp:
 * 28-51): pcl::RadiusOutlierRemoval(radius 0.03, min neighbours 10) followed by
 * flipNormalTowardsViewpoint((0,0,0)) + re-normalisation.  PCL is not vendored, so the
 * filter follows PCL 1.7 / FLANN as published: k = number of points with squared distance
 * STRICTLY below radius^2 (the point itself included), keep iff k > min_neighbors.
 * xyz / nrm: n x 3 (nrm nullable); keep[n] receives 0/1; nrm_out (nullable, n x 3) the flipped,
 * re-normalised normals of ALL points (callers compact by keep).  *n_kept = number kept (the
 * node bails out when <= 30 remain, :34-37).  Builds a temporary index in the context: call it
 * before pgp_set_scene.  Host pointers, synchronous. */
int pgp_radius_outlier_filter(pgp_ctx* ctx, const float* xyz, const float* nrm, int n, float radius,
                              int min_neighbors, unsigned char* keep, float* nrm_out, int* n_kept);

/* The part of a segment that the objects already placed do NOT explain, in front of the ICP refinement of
 * the next object (UCTState::performTrICP, PPE/hypothesis_verification/mcts/UCTState.cpp:142-174): a segment
 * point is dropped when any point of any placed object's model, moved to that object's pose, lies within
 * `radius` of it (pointRemovalThreshold = 0.008, UCTState.cpp:9; strictly below, as FLANN's radius search).
 *   seg_xyz: n x 3; model_xyz: the placed objects' model points one object after the other;
 *   model_offsets[n_objects + 1]: object k owns points model_offsets[k] .. model_offsets[k + 1] - 1;
 *   T: n_objects x 16 column-major float, model -> the segment's frame (the objects' poses);
 *   keep[n]: 1 = unexplained (stays in the cloud handed to pgp_icp_refine), *n_kept (nullable) their number.
 * n_objects == 0 keeps everything (the reference skips the step for the first object).  Host pointers,
 * synchronous.  PCL / FLANN are not vendored: unpinned against their bits. */
int pgp_unexplained_segment(pgp_ctx* ctx, const float* seg_xyz, int n, const float* model_xyz, const int* model_offsets,
                            const float* T, int n_objects, float radius, unsigned char* keep, int* n_kept);

/* pgp_set_scene with DEVICE pointers (d_xyz: n x 3 floats; d_nrm: n x 3 or NULL; d_weight: n or NULL):
 * the segment a previous device call produced (pgp_backproject_depth_device, pgp_voxel_grid_device)
 * becomes the scene without a host round trip.  Enqueued on `stream`, which is synchronised (the
 * index build needs the bounding box and two counts on the host). */
int pgp_set_scene_device(pgp_ctx* ctx, const float* d_xyz, const float* d_nrm, const float* d_weight, int n,
                         float delta, void* stream);

/* Replaces pcl::VoxelGrid<PointXYZRGB> with setLeafSize(leaf, leaf, leaf) in front of the segment
 * (PPE/segmentation/Segmentation.cpp:234-237; positions only).  PCL is not vendored: PCL 1.7's
 * published algorithm -- bounds over the finite points, voxel index (ijk relative to
 * floor(min * 1/leaf)) = i + j * div_x + k * div_x * div_y, one centroid (float sum / count) per
 * occupied voxel, leaves in ascending voxel index.  PCL leaves the order in which a voxel's points
 * are added unspecified (std::sort); here it is the point index, so results are reproducible.
 * out_xyz receives min(*n_out, cap) leaves; *n_out their full number.  Host pointers, synchronous;
 * the _device form takes n x 3 / cap x 3 device arrays and synchronises `stream` for the count. */
int pgp_voxel_grid(pgp_ctx* ctx, const float* xyz, int n, float leaf, float* out_xyz, int cap, int* n_out);
int pgp_voxel_grid_device(pgp_ctx* ctx, const float* d_xyz, int n, float leaf, float* d_out_xyz, int cap,
                          int* n_out, void* stream);

/* Replaces pcl::MovingLeastSquares<PointXYZRGB, PointXYZRGBNormal> as PPE/segmentation/Segmentation.cpp:
 * 239-246 configures it: setComputeNormals(true), setPolynomialFit(true) (order 2), setSearchRadius(radius =
 * 0.02), kd-tree radius search, no upsampling -- the step that gives the (voxel-gridded) segment its normals.
 * PCL is not vendored: PCL 1.7's published computeMLSPointNormal, in double where PCL uses double
 * (csrc/mls.hip lists the steps).  A point with fewer than 3 neighbours inside the radius (itself included)
 * is dropped, as performProcessing does; with fewer than 6 it keeps the plane's projection and normal.
 * Outputs, in input order, min(*n_out, cap) rows: out_xyz the smoothed positions, out_nrm (nullable) the
 * normals -- NOT re-normalised and not oriented, as PCL 1.7 leaves them (the node normalises and flips them
 * towards the camera afterwards, ObjectPoseCandidateSet.cpp:39-51 = pgp_radius_outlier_filter) --,
 * out_curvature (nullable) = |lambda_min / trace|, out_index (nullable) the input index of each row.
 * Host pointers, synchronous; the _device form takes device arrays and synchronises `stream` for the count. */
int pgp_mls_normals(pgp_ctx* ctx, const float* xyz, int n, float radius, float* out_xyz, float* out_nrm,
                    float* out_curvature, int* out_index, int cap, int* n_out);
int pgp_mls_normals_device(pgp_ctx* ctx, const float* d_xyz, int n, float radius, float* d_out_xyz, float* d_out_nrm,
                           float* d_out_curvature, int* d_out_index, int cap, int* n_out, void* stream);

/* Replaces Match4PCSBase::c_dist_pose and c_dist_pose_mean (S4/algorithms/match4pcsBase.cc:1616-1655)
 * for m pairs of poses: hull_xyz[n_hull][3] = hull_Q_3D (<= 4096 points), T[n_poses][16] =
 * allTransforms (column-major), pairs[m][2] = (index_1, index_2).  dist_max[m] = the directed
 * Hausdorff distance max_i min_j |T1 h_i - T2 h_j|; dist_sum[m] (nullable) = sum_i min_j (what the
 * reference calls the mean distance), added in hull order.  Host pointers, synchronous. */
int pgp_pose_hausdorff(pgp_ctx* ctx, const float* hull_xyz, int n_hull, const float* T, int n_poses,
                       const int* pairs, int m, float* dist_max, float* dist_sum);

/* pgp_backproject_depth with DEVICE pointers: d_image (rows x cols, u16 or f32), d_mask (nullable),
 * d_xyz_out (cap x 3).  *n_out (host) = full count; `stream` is synchronised for it. */
int pgp_backproject_depth_device(pgp_ctx* ctx, const void* d_image, int raw16, const unsigned char* d_mask,
                                 int rows, int cols, const float K[9], double z_min, double z_max,
                                 float* d_xyz_out, int cap, int* n_out, void* stream);

/* Depth image -> camera-frame cloud in front of the segment (PPE/misc/utilities.cpp:47-61 decode,
 * PPE/segmentation/Segmentation.cpp:219 mask, utilities.cpp:190-206 / 210-231 back-projection).
 * image: rows x cols, either the raw 16-bit PNG samples (raw16 != 0: rotated right by 3 and divided
 * by 10000 as the reference does) or depth in metres as float (raw16 == 0); mask (nullable, rows x
 * cols bytes): a pixel with mask == 0 is dropped; K: 3x3 row-major intrinsics.  Pixels with
 * z_min < depth < z_max (0.1, 2.0 in the reference) are emitted IN ROW-MAJOR ORDER as
 * x = (v - cx) * depth / fx, y = (u - cy) * depth / fy, z = depth, float operations in that order:
 * the reference's list, bit for bit.  xyz_out receives min(*n_out, cap) points; *n_out the full
 * count.  Host pointers, synchronous. */
int pgp_backproject_depth(pgp_ctx* ctx, const void* image, int raw16, const unsigned char* mask, int rows,
                          int cols, const float K[9], double z_min, double z_max, float* xyz_out, int cap,
                          int* n_out);

/* Replaces UCTState::computeCost (PPE/hypothesis_verification/mcts/UCTState.cpp:93-116) for n
 * rendered depth images against one observed image (all rows x cols float, row-major, metres):
 * render_score[i] = obScore + renScore - intScore with the pixel tests of the reference and
 * threshold = explanationThreshold (0.01 there, UCTState.cpp:8).  counts (nullable) receives the
 * three integer tallies per image.  Host pointers, synchronous. */
int pgp_depth_cost(pgp_ctx* ctx, const float* observed, const float* rendered, int n, int rows, int cols,
                   float threshold, float* render_score, int* counts);

/* Device-pointer form of pgp_depth_cost: d_observed (rows x cols) and d_rendered (n x rows x cols) are
 * float images in HBM, d_counts (n x 3 ints, {obScore, renScore, intScore}) and d_scores (nullable, n
 * floats = renderScore) are written there too; enqueued on `stream`, no synchronisation, no allocation. */
int pgp_depth_cost_device(pgp_ctx* ctx, const float* d_observed, const float* d_rendered, int n, int rows, int cols,
                          float threshold, int* d_counts, float* d_scores, void* stream);

/* Depth images of a posed object for a batch of MCTS leaf states, rendered in HBM.  Replaces the OpenGL
 * pass behind UCTState::render (PPE/hypothesis_verification/mcts/UCTState.cpp:44-72 ->
 * src/3rdparty/depth_sim/src/renderScene.cpp:45-72): the object is drawn under each of the n poses with
 * the camera's intrinsics, fragments beyond z_max are dropped (renderScene.cpp:69: 1 m), and every
 * image starts from the parent state's image, the nearer surface winning (UCTState.cpp:62-68).
 *   vertices: n_vert x vertex_stride floats (stride 3 or 4), object frame
 *   triangles: n_tri x 3 vertex indices, or NULL = every vertex is splatted into its nearest pixel
 *   T: n x 16 column-major float, object -> camera frame (convertToWorld/convertToCamera done by the caller)
 *   parent (nullable): parent_stride == 0: ONE rows x cols image under all n; else image i at parent + i * parent_stride
 *   depth: n x rows x cols floats, metres, 0 = no surface
 * The rasterisation rules (pixel centres, inclusive edges, perspective-correct depth, atomic-min z-buffer)
 * are stated in csrc/render.hip; OpenGL's are implementation-defined, so parity with the reference's
 * renderer is unpinned and tests/_checkers.py restates the rules in numpy float32, bit for bit. */
typedef struct {
  int rows, cols;
  float fx, fy, cx, cy;
  float z_near;   /* points / triangles with a vertex at z <= z_near are dropped (<= 0: 0) */
  float z_max;    /* fragments with z > z_max are dropped (<= 0: no limit); the reference uses 1.0 */
} pgp_camera;
int pgp_render_depth_device(pgp_ctx* ctx, const float* d_vertices, int vertex_stride, int n_vert, const int* d_triangles,
                            int n_tri, const float* d_T, int n, const pgp_camera* cam, const float* d_parent,
                            size_t parent_stride, float* d_depth, void* stream);
/* Host pointers, synchronous (tests, small callers): parent is ONE image or NULL. */
int pgp_render_depth(pgp_ctx* ctx, const float* vertices, int n_vert, const int* triangles, int n_tri, const float* T,
                     int n, const pgp_camera* cam, const float* parent, float* depth);

/* Replaces HypothesisSelection::greedyClustering (PPE/hypothesis_verification/HypothesisSelection.cpp:
 * 66-115) and its pose distance utilities::getPoseError (PPE/misc/utilities.cpp:514-548) for one
 * object's scored hypothesis list.  T: n_h x 16 col-major float (the Matrix4f images convertToMatrix
 * yields, utilities.cpp:276-280), scores: n_h (allPose[i].second), best_score = bestHypothesis.second,
 * sym_deg = the object's symInfo (per axis 0 / 90 / 180 / 360, PPE/data_layer/Objects.hpp:22).
 * params NULL = the reference's constants {0.5, 10, 0.02} (:70,:99).  Hypotheses with
 * score > accept_fraction * best_score are visited in descending score order (equal scores in
 * index order; the reference's std::sort leaves ties unspecified) and kept unless an earlier kept
 * one is within both thresholds.  rep_index (cap entries) receives the kept hypothesis ids in
 * that order = clusteredHypothesisSet; *n_rep their number (may exceed cap: the list is then
 * truncated); assignment (nullable, n_h) the id of the representative that absorbed each
 * hypothesis (itself for a representative, -1 when pruned).  The `+=` at :102 acts on a copy in
 * the reference, so cluster scores are the representatives' own scores[rep_index[k]].
 * Host pointers, synchronous. */
typedef struct pgp_cluster_params {
  float accept_fraction; /* 0.5  */
  float rot_thresh_deg;  /* 10   */
  float trans_thresh;    /* 0.02 */
} pgp_cluster_params;
int pgp_cluster_poses(pgp_ctx* ctx, const float* T, const float* scores, int n_h, float best_score,
                      const float sym_deg[3], const pgp_cluster_params* params, int* rep_index, int cap,
                      int* n_rep, int* assignment);
/* pgp_cluster_poses with DEVICE pointers (d_T n_h x 16, d_scores n_h, d_rep_index n_h ints,
 * d_assignment n_h ints): *n_rep (host) = number of representatives; `stream` is synchronised. */
int pgp_cluster_poses_device(pgp_ctx* ctx, const float* d_T, const float* d_scores, int n_h, float best_score,
                             const float sym_deg[3], const pgp_cluster_params* params, int* d_rep_index,
                             int* d_assignment, int* n_rep, void* stream);
/* The pose distance alone, for n pairs (test[i], gt[i]) of 4x4 col-major transforms. */
int pgp_pose_error(pgp_ctx* ctx, const float* test, const float* gt, int n, const float sym_deg[3],
                   float* rot_err_deg, float* trans_err);

/* ---- several GPUs of one node (north_star; SceneCfg.cpp:376-406 and
 * HypothesisSelection.cpp:248-257 are the consumers) -----------------------------------------------
 * A pgp_multi is a group of devices in ONE process: one pgp_ctx, one host thread and one stream per
 * device, the clouds and the index replicated on each.  pgp_multi_score_lcp block-partitions the
 * n_h hypotheses (pgp_multi_slice: contiguous slices, sizes differing by at most one), every device
 * scores its slice into a zero-initialised full-length vector, ONE RCCL all-reduce(sum) over xGMI
 * (scores as float, counts as int32, grouped) leaves every device with all of them, and device 0
 * takes the arg-max (pgp_settle_best_device) and copies the arrays back.  Outputs are those of
 * pgp_score_lcp on one device, bit for bit (plain) / with the same summation tree (weighted).
 * device_ids NULL = devices 0 .. n_dev-1; n_dev <= 0 = every visible device.  RCCL (librccl.so.1)
 * is bound at run time and only when the group has more than one device (or
 * PGP_MULTI_FORCE_COLLECTIVE=1).  Every member's worker thread queues its slice, its own ncclAllReduce (one
 * communicator per device) and, on member 0, the arg-max and the copy back: one rendezvous of the calling thread per
 * call (PGP_MULTI_COLL=grouped issues the collective for all members from the calling thread inside one group instead).
 * Member 0 holds all transforms (it settles near-ties across slices), the others copy their slice only.
 * Calls on one pgp_multi must not overlap. */
typedef struct pgp_multi pgp_multi;
int pgp_multi_create(pgp_multi** out, const int* device_ids, int n_dev);
int pgp_multi_destroy(pgp_multi* m);
int pgp_multi_size(const pgp_multi* m);
pgp_ctx* pgp_multi_context(pgp_multi* m, int k);       /* device k's context, for the other entry points */
int pgp_multi_slice(int n_total, int k, int n_dev, int* lo, int* hi);   /* host helper: device k's [lo, hi) */
int pgp_multi_set_scene(pgp_multi* m, const float* xyz, const float* nrm, const float* weight, int n,
                        float delta);
int pgp_multi_set_scene_weights(pgp_multi* m, const float* weight, int n);
int pgp_multi_set_model(pgp_multi* m, const float* xyz, const float* nrm, int n);
int pgp_multi_score_lcp(pgp_multi* m, const float* T, int n_h, int mode, float gate_deg, float* scores,
                        int* counts, int* best_index, float* best_score);
/* The two halves of pgp_multi_score_lcp, for callers that score one resident batch repeatedly
 * (bench.py): copy the transforms to every device once, then score what is there. */
int pgp_multi_upload(pgp_multi* m, const float* T, int n_h);
int pgp_multi_score_uploaded(pgp_multi* m, int mode, float gate_deg, float* scores, int* counts,
                             int* best_index, float* best_score);
/* ---- a group that holds SEVERAL objects (BASELINE configs[3]: 6 objects, 64 k hypotheses over 8 GPUs) -----
 * The node loops over the objects of a frame (PPE/data_layer/SceneCfg.cpp:376-406), each with its own segment (scene
 * side), validation model and hypothesis list.  An OBJECT of a group is one such (scene, model) pair, replicated on
 * every member; object 0 exists from pgp_multi_create on -- the single-object entry points above act on it --
 * and pgp_multi_add_object appends one (returns its id >= 1, or a negative PGP_E* code).
 * pgp_multi_score_objects flattens (object, hypothesis) into ONE index space (object after object), gives member k the
 * contiguous share pgp_multi_slice(sum n_h, k, ...) of it -- pgp_multi_flat_slices lists that share as (object, lo, hi)
 * pieces --, every member scores its pieces on the objects' contexts into a zeroed full-length vector, ONE all-reduce
 * for the concatenated {scores | counts} of all objects, and member 0 takes every object's arg-max with the near-tie
 * settlement of pgp_settle_best_device (pgp_set_exact_records / pgp_set_verify_early_out on
 * pgp_multi_object_context(m, obj, 0) apply as for one object).
 *   T[n_obj], n_h[n_obj]: every object's transform list (objects 0 .. n_obj-1 of the group; a count may be 0)
 *   scores / counts (nullable): flat, sum(n_h) entries, object after object -- per object what pgp_score_lcp returns
 *   best_index[n_obj] (nullable): per object, relative to the object's list; best_score[n_obj] (nullable). */
int pgp_multi_add_object(pgp_multi* m);
int pgp_multi_objects(const pgp_multi* m);
pgp_ctx* pgp_multi_object_context(pgp_multi* m, int obj, int k);
int pgp_multi_set_object_scene(pgp_multi* m, int obj, const float* xyz, const float* nrm, const float* weight, int n,
                               float delta);
int pgp_multi_set_object_scene_weights(pgp_multi* m, int obj, const float* weight, int n);
int pgp_multi_set_object_model(pgp_multi* m, int obj, const float* xyz, const float* nrm, int n);
int pgp_multi_set_object_search_model(pgp_multi* m, int obj, const float* xyz, int n);
int pgp_multi_set_object_ppf_map(pgp_multi* m, int obj, const int* keys, const int* counts, const int* pairs, int n_keys);
int pgp_multi_flat_slices(const int* n_h, int n_obj, int k, int n_dev, int* obj, int* lo, int* hi, int* n_pieces);
int pgp_multi_score_objects(pgp_multi* m, const float* const* T, const int* n_h, int n_obj, int mode, float gate_deg,
                            float* scores, int* counts, int* best_index, float* best_score);
int pgp_multi_upload_objects(pgp_multi* m, const float* const* T, const int* n_h, int n_obj);
int pgp_multi_score_objects_uploaded(pgp_multi* m, int mode, float gate_deg, float* scores, int* counts, int* best_index,
                                     float* best_score);

/* ICP over the group: the poses to refine are sharded like the hypotheses ("shard poses-to-refine the same way;
 * no collective except the final gather").  The jobs are the (segment, target) pairs of pgp_icp_refine -- the children of
 * an MCTS expansion (PPE/hypothesis_verification/mcts/UCTSearch.cpp:200-266 -> UCTState.cpp:121-204), the objects of a
 * frame -- with HOST pointers; the flat (job, pose) space is block-partitioned over the members
 * (pgp_multi_flat_slices), every member refines its pieces in ONE launch (pgp_icp_refine_multi_device; the target of job
 * j keeps its index on the member's j-th ICP context from call to call) and writes its poses' results straight into the
 * caller's arrays.  Results are those of one pgp_icp_refine per job on one context, bit for bit. */
typedef struct {
  const float* src_xyz;  /* n_src x 3: the points that are moved (the segment) */
  int n_src;
  const float* tgt_xyz;  /* n_tgt x 3: the cloud searched for neighbours (the model) */
  int n_tgt;
  float* T;              /* [n][16] in/out */
  int n;
  float* energy;         /* [n], nullable */
  int* iters;            /* [n], nullable */
} pgp_multi_icp_job;
int pgp_multi_icp_refine(pgp_multi* m, const pgp_multi_icp_job* jobs, int n_jobs, const pgp_icp_params* params);

/* Congruent sets over the group: the bases of an object are sharded ("shard by base (100 bases/object);
 * gather variable-length transform lists on host"; the loop base.cc:1855-1874).  Member k runs pgp_find_congruent_batch
 * for the bases pgp_multi_slice(n_bases, k, ...) on its context of object `obj` (which needs the object's scene, search
 * model and pair-feature table: pgp_multi_set_object_*), n_quads[n_bases] is gathered, and the sorted quad lists stay on
 * the member that owns the base.  pgp_multi_congruent_batch_quads / _fit route every pick (base, j) to that member and
 * return the results in the caller's pick order -- what pgp_congruent_batch_quads / _fit return on one context
 * (base_ids[n_bases][4] as there). */
int pgp_multi_find_congruent_batch(pgp_multi* m, int obj, const int* base_ids, const float* base_xyz,
                                   const float* invariants, int n_bases, float threshold, int* n_quads);
int pgp_multi_congruent_batch_quads(pgp_multi* m, int obj, const int* picks, int cnt, int* quads);
int pgp_multi_congruent_batch_fit(pgp_multi* m, int obj, const int* picks, const int* base_ids, int cnt,
                                  const float centroid_P[3], const float centroid_Q[3], float* T, double* pose,
                                  int* status, float* rms);

/* ---- a group that spans several PROCESSES (one process per GPU under a launcher: python -m torch.distributed.run, mpirun) ----
 * The same group, its members spread over processes: this process holds n_local members (devices device_ids[0 ..
 * n_local-1]) which are ranks rank0 .. rank0 + n_local - 1 of `world`.  One process calls pgp_multi_unique_id
 * (ncclGetUniqueId; 128 bytes) and hands the id to the others by whatever channel the launcher offers (a torch.distributed
 * store, MPI_Bcast, a file); every process then calls pgp_multi_create_ranked with it (ncclCommInitRank: the calls meet).
 * Every process uploads the SAME clouds and the SAME complete hypothesis list; rank r scores the slice
 * pgp_multi_slice(n_h, r, world), the all-reduce leaves every process with all scores, and member 0 of EVERY process takes
 * the arg-max -- each process returns what pgp_score_lcp returns on one device.  The scoring entry points
 * (pgp_multi_score_lcp, _upload, _score_uploaded, _score_objects*, the streaming form below) work on such a group; the
 * calls that gather into the caller's arrays without a collective (pgp_multi_icp_refine, pgp_multi_find_congruent_batch,
 * ...) need a single-process group and return PGP_ESTATE otherwise.  The consumer is the same per-object loop
 * (SceneCfg.cpp:376-406) in a node that was started once per GPU. */
int pgp_multi_unique_id(void* id128);
int pgp_multi_create_ranked(pgp_multi** out, const int* device_ids, int n_local, int rank0, int world, const void* id128);

/* What the group is made of.  rccl_ranks = ncclCommCount of the communicator the exchange runs on (0: the group has no
 * communicator -- one member without PGP_MULTI_FORCE_COLLECTIVE, or an emulated group); exchanges = all-reduces (or, in
 * an emulated group, sum kernels) issued so far. */
typedef struct {
  int n_local, world, rank0;
  int rccl_ranks;
  int emulated;
  int devices[16];        /* the first 16 local members' devices */
  long long exchanges;
} pgp_multi_info;
int pgp_multi_get_info(pgp_multi* m, pgp_multi_info* info);

/* ---- streaming form: the verification loop batch after batch without a host wait per batch (base.cc:1885-1901 run over
 * the lists of successive objects / expansions; bench.py's N > 1 headline) --------------------------------------------
 * pgp_multi_upload_slot leaves a hypothesis list resident in one of 16 slots (object 0's clouds; synchronous).
 * pgp_multi_enqueue_slot queues one scoring step over a slot and returns: every member scores its slice into one of two
 * {scores | counts} vectors on its stream, the all-reduce runs on a SECOND stream of the member -- under the scoring of
 * the next step --, and member 0's arg-max (near-tie settlement, exact records, Verify's early termination as set on
 * member 0's context) follows it on that second stream, with a settlement workspace of its own.  pgp_multi_collect completes everything queued and
 * returns the LAST step's arrays, bit for bit those of pgp_multi_score_lcp on the same list.  pgp_multi_upload_slot
 * between an enqueue and its collect returns PGP_ESTATE. */
int pgp_multi_upload_slot(pgp_multi* m, int slot, const float* T, int n_h);
int pgp_multi_enqueue_slot(pgp_multi* m, int slot, int mode, float gate_deg);
int pgp_multi_collect(pgp_multi* m, float* scores, int* counts, int* best_index, float* best_score);

/* Host wall clock of the last scoring call in ms: upload (pinned copy + H2D enqueue), enqueue
 * (kernels + collective issued on every device), total (until the results are back). */
int pgp_multi_last_timing(pgp_multi* m, float* upload_ms, float* enqueue_ms, float* total_ms);

/* Per-kernel timing for bench.py's roofline line.  enable = N >= 1: every Nth pgp_score_lcp[_device]
 * call (the 1st, N+1st, ...) attaches a start and a stop HIP event to its dominant kernel's dispatch
 * (score_hypotheses), on the SAME stream it is launched on; 0 switches it off.  Timing a launch costs
 * the stream about 8 us (the runtime serialises around a timed dispatch), which is why a throughput
 * measurement samples (N = 8 in bench.py) instead of timing every step.  pgp_get_kernel_timing
 * synchronises those events and returns the number of timed launches and the sum of their
 * durations since the last reset. */
int pgp_set_kernel_timing(pgp_ctx* ctx, int enable);
int pgp_get_kernel_timing(pgp_ctx* ctx, int* launches, float* total_ms, int reset);

#ifdef __cplusplus
}
#endif
#endif /* PGP_H */
