/* oracle/pgp_oracle.c -- CPU restatement of the reference's scoring path.  TEST INFRASTRUCTURE
 * ONLY (see pgp_oracle.h for the parity status and the import rule).
 *
 * Every float operation below is written as a separately-rounded C expression in the exact
 * order Eigen 3.3.90 evaluates the corresponding reference expression on x86-64/SSE2 (pinned
 * empirically against oracle/_ref, see tests/test_oracle_vs_ref.py).  Build with
 * -ffp-contract=off (oracle/Makefile) so no FMA is formed.
 */
#include "pgp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KD_MAX_DEPTH 32       /* kdtree.h:60 */
#define KD_POINT_PER_CELL 64  /* kdtree.h:63 */

/* ---- elementary expressions ------------------------------------------------------------- */

void orc_transform_point(const float T[16], const float q[3], float out[3]) {
  for (int r = 0; r < 3; ++r) {
    float a = T[0 + r] * q[0];
    float b = T[4 + r] * q[1];
    float c = T[8 + r] * q[2];
    float s = a + b;
    s = s + c;
    out[r] = s + T[12 + r];
  }
}

void orc_rotate_normal(const float T[16], const float n[3], float out[3]) {
  for (int r = 0; r < 3; ++r) {
    float a = T[0 + r] * n[0];
    float b = T[4 + r] * n[1];
    float c = T[8 + r] * n[2];
    float bc = b + c;
    out[r] = a + bc;
  }
}

float orc_sqdist(const float a[3], const float b[3]) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  float t = yy + zz;
  return xx + t;
}

float orc_dot(const float a[3], const float b[3]) {
  float x = a[0] * b[0], y = a[1] * b[1], z = a[2] * b[2];
  float t = y + z;
  return x + t;
}

int orc_normal_gate(float dot, float gate_deg) {
  /* base.cc:1756: float angle_n = std::acos(<float>)*180/M_PI;  -> acosf, float*int->float,
   * then / (double)M_PI in double, narrowed to float on assignment. */
  float ac = acosf(dot);
  float ac180 = ac * 180.0f;
  float angle_n = (float)((double)ac180 / M_PI);
  /* base.cc:1757: std::min(angle_n, fabs(180-angle_n)) -- all float (SURVEY hazard 5).
   * std::min(a,b) = (b < a) ? b : a, so a NaN angle_n stays NaN. */
  float other = fabsf(180.0f - angle_n);
  float m = (other < angle_n) ? other : angle_n;
  return m < gate_deg; /* NaN -> false (hazard 4) */
}

/* ---- kd-tree (kdtree.h) ------------------------------------------------------------------ */

typedef struct {
  int leaf;
  /* internal */
  float splitValue;
  int firstChildId;
  int dim;
  /* leaf */
  int start;
  int size;
} kd_node;

struct orc_kdtree {
  float* pts; /* reordered copy, n x 3 (mPoints) */
  int* idx;   /* original ids (mIndices) */
  int n;
  kd_node* nodes;
  int n_nodes, cap_nodes;
};

static int kd_push_node(orc_kdtree* t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 64;
    t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
  }
  kd_node* nd = &t->nodes[t->n_nodes];
  memset(nd, 0, sizeof(*nd));
  return t->n_nodes++;
}

static void kd_swap(orc_kdtree* t, int a, int b) {
  float tmp[3];
  memcpy(tmp, t->pts + 3 * a, sizeof tmp);
  memcpy(t->pts + 3 * a, t->pts + 3 * b, sizeof tmp);
  memcpy(t->pts + 3 * b, tmp, sizeof tmp);
  int ti = t->idx[a];
  t->idx[a] = t->idx[b];
  t->idx[b] = ti;
}

/* kdtree.h:522-538 */
static int kd_split(orc_kdtree* t, int start, int end, int dim, float splitValue) {
  int l = start, r = end - 1;
  for (; l < r; ++l, --r) {
    while (l < end && t->pts[3 * l + dim] < splitValue) l++;
    while (r >= start && t->pts[3 * r + dim] >= splitValue) r--;
    if (l > r) break;
    kd_swap(t, l, r);
  }
  if (l >= end) return l; /* the reference would read one past the range here (UB) */
  return (t->pts[3 * l + dim] < splitValue) ? l + 1 : l;
}

/* kdtree.h:560-641 */
static void kd_create(orc_kdtree* t, int nodeId, int start, int end, int level) {
  float mn[3] = {FLT_MAX / 2, FLT_MAX / 2, FLT_MAX / 2};      /* bbox.h:63-64 */
  float mx[3] = {-FLT_MAX / 2, -FLT_MAX / 2, -FLT_MAX / 2};
  for (int i = start; i < end; ++i)
    for (int k = 0; k < 3; ++k) {
      float v = t->pts[3 * i + k];
      if (v < mn[k]) mn[k] = v; /* bbox.h:73-75: select(q < min, q, min) */
      if (v > mx[k]) mx[k] = v;
    }
  float diag[3];
  for (int k = 0; k < 3; ++k) diag[k] = 0.5f * (mx[k] - mn[k]);
  int dim = 0; /* Eigen maxCoeff(&dim): first strict maximum */
  for (int k = 1; k < 3; ++k)
    if (diag[k] > diag[dim]) dim = k;
  float splitValue = mn[dim] + ((mx[dim] - mn[dim]) / 2.0f); /* bbox.h:88-89 center() */

  t->nodes[nodeId].dim = dim;
  t->nodes[nodeId].splitValue = splitValue;
  int midId = kd_split(t, start, end, dim, splitValue);

  int first = kd_push_node(t);
  kd_push_node(t);
  t->nodes[nodeId].firstChildId = first;

  if ((unsigned)(midId - start) <= KD_POINT_PER_CELL || level >= KD_MAX_DEPTH) {
    t->nodes[first].leaf = 1;
    t->nodes[first].start = start;
    t->nodes[first].size = midId - start; /* reference narrows to unsigned short (kdtree.h:155) */
  } else {
    t->nodes[first].leaf = 0;
    kd_create(t, first, start, midId, level + 1);
  }
  if ((unsigned)(end - midId) <= KD_POINT_PER_CELL || level >= KD_MAX_DEPTH) {
    t->nodes[first + 1].leaf = 1;
    t->nodes[first + 1].start = midId;
    t->nodes[first + 1].size = end - midId;
  } else {
    t->nodes[first + 1].leaf = 0;
    kd_create(t, first + 1, midId, end, level + 1);
  }
}

orc_kdtree* orc_kd_build(const float* xyz, int n) {
  orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof(*t));
  t->n = n;
  t->pts = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  t->idx = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  memcpy(t->pts, xyz, sizeof(float) * 3 * (size_t)n);
  for (int i = 0; i < n; ++i) t->idx[i] = i;
  int root = kd_push_node(t); /* kdtree.h:362-363: one node, leaf = 0 */
  t->nodes[root].leaf = 0;
  kd_create(t, 0, 0, n, 1); /* kdtree.h:367 */
  return t;
}

void orc_kd_free(orc_kdtree* t) {
  if (!t) return;
  free(t->pts);
  free(t->idx);
  free(t->nodes);
  free(t);
}

int orc_kd_num_nodes(const orc_kdtree* t) { return t->n_nodes; }

/* kdtree.h:394-459 */
int orc_kd_query(const orc_kdtree* t, const float q[3], float sqdist) {
  struct { int nodeId; float sq; } stack[64];
  int cl_id = -1;
  float cl_dist = sqdist;
  stack[0].nodeId = 0;
  stack[0].sq = 0.f;
  unsigned count = 1;
  while (count) {
    int top = (int)count - 1;
    const kd_node* node = &t->nodes[stack[top].nodeId];
    if (stack[top].sq < cl_dist) {
      if (node->leaf) {
        --count;
        const int end = node->start + node->size;
        for (int i = node->start; i < end; ++i) {
          const float d = orc_sqdist(q, t->pts + 3 * i);
          if (d <= cl_dist) { /* inclusive; a later equal distance replaces the earlier */
            cl_dist = d;
            cl_id = t->idx[i];
          }
        }
      } else {
        const float new_off = q[node->dim] - node->splitValue;
        if (new_off < 0.) {
          stack[count].nodeId = node->firstChildId;
          stack[top].nodeId = node->firstChildId + 1;
        } else {
          stack[count].nodeId = node->firstChildId + 1;
          stack[top].nodeId = node->firstChildId;
        }
        stack[count].sq = stack[top].sq;
        stack[top].sq = new_off * new_off;
        ++count;
      }
    } else {
      --count;
    }
  }
  return cl_id;
}

int orc_brute_query(const float* xyz, int n, const float q[3], float sqdist) {
  int best = -1;
  float bd = sqdist;
  for (int i = 0; i < n; ++i) {
    float d = orc_sqdist(q, xyz + 3 * i);
    if (best < 0 ? d <= bd : d < bd) {
      bd = d;
      best = i;
    }
  }
  return best;
}

/* ---- Verify / WeightedVerify ------------------------------------------------------------- */

static int nn_query(const orc_kdtree* kd, const float* P_xyz, int nP, const float q[3], float sq) {
  return kd ? orc_kd_query(kd, q, sq) : orc_brute_query(P_xyz, nP, q, sq);
}

float orc_verify(const orc_kdtree* kd, const float* P_xyz, int nP, const float* Q_xyz, int nQ,
                 const float T[16], float delta, float best_lcp, int early_out,
                 int* good_out, int* hit_ids) {
  const float epsilon = delta;
  int good_points = 0;
  const size_t number_of_points = (size_t)nQ;
  const int terminate_value = (int)(best_lcp * number_of_points); /* float * size_t -> float -> int */
  const float sq_eps = epsilon * epsilon;
  for (int i = 0; i < nQ; ++i) {
    float x[3];
    orc_transform_point(T, Q_xyz + 3 * i, x);
    int resId = nn_query(kd, P_xyz, nP, x, sq_eps);
    if (hit_ids) hit_ids[i] = resId;
    if (resId != -1) good_points++;
    /* base.cc:1725: size_t arithmetic, compared with an int promoted to size_t */
    if (early_out && (number_of_points - (size_t)i + (size_t)good_points) < (size_t)terminate_value) break;
  }
  if (good_out) *good_out = good_points;
  return (float)good_points / (float)number_of_points;
}

float orc_weighted_verify(const orc_kdtree* kd, const float* P_xyz, const float* P_nrm,
                          const float* P_w, int nP, const float* Q_xyz, const float* Q_nrm, int nQ,
                          const float T[16], float delta, float gate_deg,
                          int* registered, int* n_registered) {
  const float sq_eps = delta * delta;
  float weighted_match = 0;
  int nreg = 0;
  for (int i = 0; i < nQ; ++i) {
    float x[3];
    orc_transform_point(T, Q_xyz + 3 * i, x);
    int resId = nn_query(kd, P_xyz, nP, x, sq_eps);
    if (resId != -1) {
      float n_q[3];
      orc_rotate_normal(T, Q_nrm + 3 * i, n_q);
      float d = orc_dot(P_nrm + 3 * resId, n_q);
      if (orc_normal_gate(d, gate_deg)) {
        weighted_match += P_w ? P_w[resId] : 1.0f;
        if (registered) registered[nreg] = resId;
        nreg++;
      }
    }
  }
  if (n_registered) *n_registered = nreg;
  return weighted_match / (float)(size_t)nQ;
}

void orc_score_batch(const orc_kdtree* kd, const float* P_xyz, const float* P_nrm, const float* P_w,
                     int nP, const float* Q_xyz, const float* Q_nrm, int nQ,
                     const float* T, int n_h, float delta, int mode, float gate_deg,
                     int early_out, int threads,
                     float* scores, int* best_index, int* selected, int* n_selected) {
  if (mode == 0 && early_out) {
    /* order-dependent: strictly serial, as base.cc:1888-1901 */
    float best = 0.f;
    int bi = -1, ns = 0;
    for (int h = 0; h < n_h; ++h) {
      float lcp = orc_verify(kd, P_xyz, nP, Q_xyz, nQ, T + 16 * (size_t)h, delta, best, 1, NULL, NULL);
      scores[h] = lcp;
      if (lcp > best) {
        best = lcp;
        bi = h;
        if (selected) selected[ns] = h;
        ns++;
      }
    }
    if (best_index) *best_index = bi;
    if (n_selected) *n_selected = ns;
    return;
  }
  if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
#endif
  for (int h = 0; h < n_h; ++h) {
    const float* Th = T + 16 * (size_t)h;
    if (mode == 0)
      scores[h] = orc_verify(kd, P_xyz, nP, Q_xyz, nQ, Th, delta, 0.f, 0, NULL, NULL);
    else
      scores[h] = orc_weighted_verify(kd, P_xyz, P_nrm, P_w, nP, Q_xyz, Q_nrm, nQ, Th, delta,
                                      gate_deg, NULL, NULL);
  }
  float best = 0.f; /* base.cc:308 best_LCP_ = 0.0 */
  int bi = -1, ns = 0;
  for (int h = 0; h < n_h; ++h) {
    if (scores[h] > best) {
      best = scores[h];
      bi = h;
      if (selected) selected[ns] = h;
      ns++;
    }
  }
  if (best_index) *best_index = bi;
  if (n_selected) *n_selected = ns;
}

/* ---- init(): centring (base.cc:242-268) -------------------------------------------------- */

void orc_center(float* P_xyz, int nP, float* Qs_xyz, int nQs, float* Qv_xyz, int nQv,
                float centroid_P[3], float centroid_Q[3]) {
  float cP[3] = {0, 0, 0}, cQ[3] = {0, 0, 0};
  for (int i = 0; i < nP; ++i)
    for (int k = 0; k < 3; ++k) cP[k] += P_xyz[3 * i + k];
  for (int k = 0; k < 3; ++k) cP[k] /= (float)nP;
  for (int i = 0; i < nQs; ++i)
    for (int k = 0; k < 3; ++k) cQ[k] += Qs_xyz[3 * i + k];
  for (int k = 0; k < 3; ++k) cQ[k] /= (float)nQs;
  for (int i = 0; i < nP; ++i)
    for (int k = 0; k < 3; ++k) P_xyz[3 * i + k] -= cP[k];
  for (int i = 0; i < nQs; ++i)
    for (int k = 0; k < 3; ++k) Qs_xyz[3 * i + k] -= cQ[k];
  for (int i = 0; i < nQv; ++i)
    for (int k = 0; k < 3; ++k) Qv_xyz[3 * i + k] -= cQ[k];
  for (int k = 0; k < 3; ++k) {
    centroid_P[k] = cP[k];
    centroid_Q[k] = cQ[k];
  }
}

/* ---- rigid fit from a congruent pair (base.cc:1411-1488, 1504-1614) ---------------------- */

static float sqnorm3(const float v[3]) { /* Eigen squaredNorm: x*x + (y*y + z*z) */
  float x = v[0] * v[0], y = v[1] * v[1], z = v[2] * v[2];
  float t = y + z;
  return x + t;
}
static float dot3(const float a[3], const float b[3]) {
  float x = a[0] * b[0], y = a[1] * b[1], z = a[2] * b[2];
  float t = y + z;
  return x + t;
}
static void normalize3(float v[3]) { /* Eigen normalize(): z = squaredNorm; if z > 0: v /= sqrt(z) */
  float z = sqnorm3(v);
  if (z > 0.f) {
    float n = sqrtf(z);
    v[0] = v[0] / n;
    v[1] = v[1] / n;
    v[2] = v[2] / n;
  }
}
static void cross3(const float a[3], const float b[3], float o[3]) {
  float t0 = a[1] * b[2], t1 = a[2] * b[1];
  float t2 = a[2] * b[0], t3 = a[0] * b[2];
  float t4 = a[0] * b[1], t5 = a[1] * b[0];
  o[0] = t0 - t1;
  o[1] = t2 - t3;
  o[2] = t4 - t5;
}
/* Gram-Schmidt frame of base.cc:1532-1546; returns 0 on the degenerate exits */
static int frame3(const float a0[3], const float a1[3], const float a2[3], float v1[3], float v2[3],
                  float v3[3]) {
  float d[3], proj;
  for (int k = 0; k < 3; ++k) v1[k] = a1[k] - a0[k];
  if (sqnorm3(v1) == 0) return 0;
  normalize3(v1);
  for (int k = 0; k < 3; ++k) d[k] = a2[k] - a0[k];
  proj = dot3(d, v1);
  for (int k = 0; k < 3; ++k) {
    float t = proj * v1[k];
    v2[k] = d[k] - t;
  }
  if (sqnorm3(v2) == 0) return 0;
  normalize3(v2);
  cross3(v1, v2, v3);
  return 1;
}

int orc_rigid_from_pair(const float* p, const float* q, const float centroid_P[3],
                        const float centroid_Q[3], float* T_centred, double* pose, float* rms_out) {
  const float kLargeNumber = 1e9f, kSmallNumber = 1e-6f;
  float c1[3], c2[3];
  for (int k = 0; k < 3; ++k) { /* (b1 + b2 + b3) / 3 */
    float s1 = p[k] + p[3 + k];
    s1 = s1 + p[6 + k];
    c1[k] = s1 / 3.0f;
    float s2 = q[k] + q[3 + k];
    s2 = s2 + q[6 + k];
    c2[k] = s2 / 3.0f;
  }
  if (rms_out) *rms_out = kLargeNumber;
  float p1[3], p2[3], p3[3], q1[3], q2[3], q3[3];
  if (!frame3(p, p + 3, p + 6, p1, p2, p3)) return 2;
  if (!frame3(q, q + 3, q + 6, q1, q2, q3)) return 2;
  /* rotation = rotate_p^T * rotate_q : R(i,j) = p1_i q1_j + (p2_i q2_j + p3_i q3_j) */
  float R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float a = p1[i] * q1[j], b = p2[i] * q2[j], c = p3[i] * q3[j];
      float bc = b + c;
      R[i][j] = a + bc;
    }
  for (int i = 0; i < 3; ++i) { /* diag(R*R) - 1 > 1e-6 -> reject */
    float a = R[i][0] * R[0][i], b = R[i][1] * R[1][i], c = R[i][2] * R[2][i];
    float bc = b + c;
    float d = a + bc;
    if (d - 1.0f > kSmallNumber) return 0;
  }
  float rms = 0.f;
  for (int i = 0; i < 3; ++i) {
    float first[3], tr[3], e[3];
    for (int k = 0; k < 3; ++k) first[k] = 1.0f * q[3 * i + k] - c2[k];
    for (int r = 0; r < 3; ++r) {
      float a = R[r][0] * first[0], b = R[r][1] * first[1], c = R[r][2] * first[2];
      float bc = b + c;
      tr[r] = a + bc;
    }
    for (int k = 0; k < 3; ++k) {
      float t = tr[k] - p[3 * i + k];
      e[k] = t + c1[k];
    }
    rms += sqrtf(sqnorm3(e));
  }
  rms /= 4.0f;
  if (rms_out) *rms_out = rms;
  /* etrans = translate(c1) * rotate(R) * translate(-c2): t_r = c1_r + (R_r0*(-c2_0) + (R_r1*(-c2_1) + R_r2*(-c2_2))) */
  float t[3];
  for (int r = 0; r < 3; ++r) {
    float a = R[r][0] * (-c2[0]), b = R[r][1] * (-c2[1]), c = R[r][2] * (-c2[2]);
    float bc = b + c;
    float s = a + bc;
    t[r] = c1[r] + s;
  }
  if (!(rms >= 0.f)) return 0; /* base.cc:1467 `ok && rms >= 0` (false for NaN) */
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) T_centred[4 * c + r] = (r == c) ? 1.f : 0.f;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T_centred[4 * c + r] = R[r][c];
    T_centred[12 + r] = t[r];
  }
  /* de-centring (base.cc:1474-1482): col(3) = (c1 + cP) - L*(c2 + cQ) */
  float u[3], tw[3];
  for (int k = 0; k < 3; ++k) u[k] = c2[k] + centroid_Q[k];
  for (int r = 0; r < 3; ++r) {
    float a = R[r][0] * u[0], b = R[r][1] * u[1], c = R[r][2] * u[2];
    float bc = b + c;
    float s = a + bc;
    float l = c1[r] + centroid_P[r];
    tw[r] = l - s;
  }
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) pose[4 * c + r] = (double)T_centred[4 * c + r];
  for (int r = 0; r < 3; ++r) pose[12 + r] = (double)tw[r];
  return 1;
}

/* ---- congruent-set extraction ---------------------------------------------------------------- */

struct orc_cs {
  int n;
  float* world; /* n x 3, sampled_Q_3D_ positions */
  float* unit;  /* n x 3, pcfunctor_.points (unit cube) */
  float gcenter[3];
  float ratio;
};

orc_cs* orc_cs_create(const float* Q, int n) {
  orc_cs* s = (orc_cs*)calloc(1, sizeof(*s));
  s->n = n;
  s->world = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  s->unit = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  memcpy(s->world, Q, sizeof(float) * 3 * (size_t)n);
  float mn[3] = {FLT_MAX / 2, FLT_MAX / 2, FLT_MAX / 2}, mx[3] = {-FLT_MAX / 2, -FLT_MAX / 2, -FLT_MAX / 2};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      float v = Q[3 * i + k];
      if (v < mn[k]) mn[k] = v;
      if (v > mx[k]) mx[k] = v;
    }
  float ext[3];
  for (int k = 0; k < 3; ++k) {
    ext[k] = mx[k] - mn[k];
    s->gcenter[k] = mn[k] + (ext[k] / 2.0f); /* bbox.center() */
  }
  /* pairCreationFunctor.h:122-124: std::max(depth+0.001, max(width+0.001, height+0.001)) in double */
  double r = (double)ext[2] + 0.001, w = (double)ext[1] + 0.001, h = (double)ext[0] + 0.001;
  double m = w > h ? w : h;
  m = r > m ? r : m;
  s->ratio = (float)m;
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) { /* worldToUnit: (p - gcenter) / ratio + half */
      float d = Q[3 * i + k] - s->gcenter[k];
      float u = d / s->ratio;
      s->unit[3 * i + k] = u + 0.5f;
    }
  return s;
}

void orc_cs_free(orc_cs* s) {
  if (!s) return;
  free(s->world);
  free(s->unit);
  free(s);
}

int orc_cs_extract_pairs(const orc_cs* s, float pair_distance, float eps, int* out, int cap) {
  int n = 0;
  const double pd = (double)pair_distance, pe = (double)eps; /* functor members are double */
  for (int i = 0; i < s->n; ++i)
    for (int j = 0; j < i; ++j) {
      float d[3];
      for (int k = 0; k < 3; ++k) d[k] = s->world[3 * i + k] - s->world[3 * j + k];
      const float distance = sqrtf(sqnorm3(d));
      if (fabs((double)distance - pd) > pe) continue;
      if (n < cap) { out[2 * n] = j; out[2 * n + 1] = i; }
      ++n;
      if (n < cap) { out[2 * n] = i; out[2 * n + 1] = j; }
      ++n;
    }
  return n;
}

static void vnormalized(const float v[3], float o[3]) { /* Eigen normalized(): z > 0 ? v / sqrt(z) : v */
  float z = sqnorm3(v);
  if (z > 0.f) {
    float nrm = sqrtf(z);
    o[0] = v[0] / nrm; o[1] = v[1] / nrm; o[2] = v[2] / nrm;
  } else {
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  }
}

/* IndexedNormalSet::indexNormal: int((n/2 + 1/2) / _nepsilon) per dim, 7 bins, _nepsilon = 1/7 + 1e-5 */
static int normal_bin(const float n[3]) {
  const float nepsilon = (float)((double)(1.0f / 7.0f) + 0.00001);
  int idx[3];
  for (int k = 0; k < 3; ++k) {
    float h = n[k] / 2.0f;
    float c = h + 0.5f;
    c = c / nepsilon;
    idx[k] = (int)c;
    if (idx[k] < 0 || idx[k] > 6) return -1; /* the reference does not validate (UB) */
  }
  return idx[2] * 49 + idx[1] * 7 + idx[0];
}

static long long pos_cell(const float p[3], float epsilon, int eg) {
  long long idx[3];
  for (int k = 0; k < 3; ++k) {
    float c = p[k] / epsilon;
    idx[k] = (long long)(int)c;
    if (idx[k] < 0 || idx[k] >= eg) return -1; /* outside the unit cube: never for inv in [0,1] */
  }
  return (idx[2] * eg + idx[1]) * eg + idx[0];
}

typedef struct { long long cell; int bin; int id; } cs_entry;
static int cs_entry_cmp(const void* a, const void* b) {
  const cs_entry* x = (const cs_entry*)a;
  const cs_entry* y = (const cs_entry*)b;
  if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
  return (x->id > y->id) - (x->id < y->id);
}
typedef struct { int id, i; } cs_match;
static int cs_match_cmp(const void* a, const void* b) {
  const cs_match* x = (const cs_match*)a;
  const cs_match* y = (const cs_match*)b;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return (x->i > y->i) - (x->i < y->i);
}

int orc_cs_find_congruent(const orc_cs* s, const float* base, float inv1, float inv2, float threshold,
                          const int* Pp, int nP, const int* Qp, int nQ, int* quads, int cap) {
  /* alpha = (b1-b0).normalized().dot((b3-b2).normalized())  (super4pcs.cc:107-109) */
  float d01[3], d23[3], u01[3], u23[3];
  for (int k = 0; k < 3; ++k) {
    d01[k] = base[3 + k] - base[k];
    d23[k] = base[9 + k] - base[6 + k];
  }
  vnormalized(d01, u01);
  vnormalized(d23, u23);
  const float cosAlpha = dot3(u01, u23);
  /* IndexedNormalSet ctor (normalset.h:114-122) */
  const float eps = threshold / s->ratio; /* getNormalizedEpsilon */
  const int gridDepth = (int)(-log2f(eps));
  const int eg = (int)pow(2, gridDepth);
  const float epsilon = 1.f / (float)eg;

  cs_entry* ent = (cs_entry*)malloc(sizeof(cs_entry) * (size_t)(nP > 0 ? nP : 1));
  int ne = 0;
  for (int i = 0; i < nP; ++i) {
    const float* p1 = s->unit + 3 * Pp[2 * i];
    const float* p2 = s->unit + 3 * Pp[2 * i + 1];
    float d[3], nrm[3], pos[3];
    for (int k = 0; k < 3; ++k) d[k] = p2[k] - p1[k];
    vnormalized(d, nrm);
    for (int k = 0; k < 3; ++k) {
      float t = inv1 * d[k];
      pos[k] = p1[k] + t;
    }
    long long c = pos_cell(pos, epsilon, eg);
    int b = normal_bin(nrm);
    if (c < 0 || b < 0) continue;
    ent[ne].cell = c;
    ent[ne].bin = b;
    ent[ne].id = i;
    ++ne;
  }
  qsort(ent, (size_t)ne, sizeof(cs_entry), cs_entry_cmp);

  /* cone "rendering" constants (normalset.hpp:176-182); NaN alpha -> no samples (the reference
   * converts NaN to unsigned there, which is undefined) */
  const float alpha = acosf(cosAlpha);
  const float perimeter = (float)((double)2.0f * M_PI * (double)atanf(alpha));
  const float nbf = 2 * ceilf(perimeter * 7.0f / 2.0f);
  const unsigned nbSample = (nbf == nbf && nbf > 0.f && nbf < 1e6f) ? (unsigned)nbf : 0u;
  const float angleStep = (float)((double)2.0f * M_PI / (double)(float)nbSample);
  const float sinAlpha = sinf(alpha);

  size_t mcap = 1024, nm = 0;
  cs_match* mt = (cs_match*)malloc(sizeof(cs_match) * mcap);
  for (int i = 0; i < nQ; ++i) {
    const int a = Qp[2 * i], b = Qp[2 * i + 1];
    const float* p1 = s->unit + 3 * a;
    const float* p2 = s->unit + 3 * b;
    float d[3], query[3], queryn[3], queryQ[3];
    for (int k = 0; k < 3; ++k) {
      d[k] = p2[k] - p1[k];
      float t = inv2 * d[k];
      query[k] = p1[k] + t;
      float dw = s->world[3 * b + k] - s->world[3 * a + k];
      float tw = inv2 * dw;
      queryQ[k] = s->world[3 * a + k] + tw;
    }
    vnormalized(d, queryn);
    long long c = pos_cell(query, epsilon, eg);
    if (c < 0) continue;
    /* range of entries in the same cell */
    int lo = 0, hi = ne;
    while (lo < hi) { int mid = (lo + hi) / 2; if (ent[mid].cell < c) lo = mid + 1; else hi = mid; }
    int first = lo;
    hi = ne;
    while (lo < hi) { int mid = (lo + hi) / 2; if (ent[mid].cell <= c) lo = mid + 1; else hi = mid; }
    int last = lo;
    if (first == last) continue; /* angularGrid(p) == NULL */
    /* q.setFromTwoVectors((0,0,1), queryn)  (Eigen Quaternion.h:577-610) */
    float v1[3];
    vnormalized(queryn, v1);
    float cq = v1[2]; /* v1.dot((0,0,1)) = x*0 + (y*0 + z*1) */
    {
      float x = v1[0] * 0.f, y = v1[1] * 0.f, z = v1[2] * 1.f;
      float t = y + z;
      cq = x + t;
    }
    float qv[3], qw;
    if (cq < -1.0f + 1e-5f) {
      /* nearly opposite: the reference solves a 2x3 SVD for the axis; any unit axis orthogonal to
       * z gives a valid half-turn -- we take x (documented divergence, |dir + z| < 0.26 deg) */
      float cc = cq > -1.0f ? cq : -1.0f;
      float w2 = (1.0f + cc) * 0.5f;
      qw = sqrtf(w2);
      float sv = sqrtf(1.0f - w2);
      qv[0] = sv; qv[1] = 0.f; qv[2] = 0.f;
    } else {
      /* axis = (0,0,1) x v1 = (0*v1z - 1*v1y, 1*v1x - 0*v1z, 0*v1y - 0*v1x) */
      float axis[3];
      axis[0] = 0.f * v1[2] - 1.f * v1[1];
      axis[1] = 1.f * v1[0] - 0.f * v1[2];
      axis[2] = 0.f * v1[1] - 0.f * v1[0];
      float sq = sqrtf((1.0f + cq) * 2.0f);
      float invs = 1.0f / sq;
      qv[0] = axis[0] * invs; qv[1] = axis[1] * invs; qv[2] = axis[2] * invs;
      qw = sq * 0.5f;
    }
    unsigned char colored[343];
    memset(colored, 0, sizeof colored);
    for (unsigned a2 = 0; a2 != nbSample; ++a2) {
      float theta = (float)a2 * angleStep;
      float v[3] = {sinAlpha * cosf(theta), sinAlpha * sinf(theta), cosAlpha};
      /* q * v = v + w*uv + vec x uv, uv = 2 (vec x v)  (Quaternion.h:470-480) */
      float uv[3], c2[3], r[3], dir[3];
      cross3(qv, v, uv);
      for (int k = 0; k < 3; ++k) uv[k] = uv[k] + uv[k];
      cross3(qv, uv, c2);
      for (int k = 0; k < 3; ++k) {
        float t = qw * uv[k];
        float t2 = v[k] + t;
        r[k] = t2 + c2[k];
      }
      vnormalized(r, dir);
      int id = normal_bin(dir);
      if (id >= 0) colored[id] = 1;
    }
    for (int e = first; e < last; ++e) {
      if (!colored[ent[e].bin]) continue;
      const int id = ent[e].id;
      const float* pp1 = s->world + 3 * Pp[2 * id];
      const float* pp2 = s->world + 3 * Pp[2 * id + 1];
      float diff[3];
      for (int k = 0; k < 3; ++k) {
        float dd = pp2[k] - pp1[k];
        float t = dd * inv1;
        float ip = pp1[k] + t;
        diff[k] = queryQ[k] - ip;
      }
      if (sqnorm3(diff) <= threshold) { /* squared distance vs delta, sic (super4pcs.cc:170) */
        if (nm == mcap) { mcap *= 2; mt = (cs_match*)realloc(mt, sizeof(cs_match) * mcap); }
        mt[nm].id = id;
        mt[nm].i = i;
        ++nm;
      }
    }
  }
  qsort(mt, nm, sizeof(cs_match), cs_match_cmp);
  for (size_t k = 0; k < nm; ++k)
    if ((int)k < cap) {
      quads[4 * k] = Pp[2 * mt[k].id];
      quads[4 * k + 1] = Pp[2 * mt[k].id + 1];
      quads[4 * k + 2] = Qp[2 * mt[k].i];
      quads[4 * k + 3] = Qp[2 * mt[k].i + 1];
    }
  free(ent);
  free(mt);
  return (int)nm;
}

/* ---- trimmed ICP (own definition; PCL absent) --------------------------------------------- */

typedef struct { float d2; int i; } icp_pair;

static int icp_cmp(const void* a, const void* b) {
  const icp_pair* x = (const icp_pair*)a;
  const icp_pair* y = (const icp_pair*)b;
  if (x->d2 < y->d2) return -1;
  if (x->d2 > y->d2) return 1;
  return (x->i > y->i) - (x->i < y->i); /* ties: lowest source index first */
}

static void jacobi_eig4(double A[4][4], double q[4]) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 16; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 4; ++p)
      for (int r = p + 1; r < 4; ++r) off += A[p][r] * A[p][r];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int r = p + 1; r < 4; ++r) {
        double apr = A[p][r];
        if (apr == 0.0) continue;
        double theta = (A[r][r] - A[p][p]) / (2.0 * apr);
        double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) {
          double akp = A[k][p], akr = A[k][r];
          A[k][p] = c * akp - s * akr;
          A[k][r] = s * akp + c * akr;
        }
        for (int k = 0; k < 4; ++k) {
          double apk = A[p][k], ark = A[r][k];
          A[p][k] = c * apk - s * ark;
          A[r][k] = s * apk + c * ark;
        }
        for (int k = 0; k < 4; ++k) {
          double vkp = V[k][p], vkr = V[k][r];
          V[k][p] = c * vkp - s * vkr;
          V[k][r] = s * vkp + c * vkr;
        }
      }
  }
  int best = 0;
  for (int k = 1; k < 4; ++k)
    if (A[k][k] > A[best][best]) best = k;
  for (int k = 0; k < 4; ++k) q[k] = V[k][best];
}

int orc_icp(const float* src, int n_src, const float* tgt, int n_tgt, float* T, int max_iterations,
            float trim_fraction, float max_corr_dist, float energy_ratio, float* energy_out) {
  if (max_iterations <= 0) max_iterations = 100;
  if (!(trim_fraction > 0.f) || trim_fraction > 1.f) trim_fraction = 1.f;
  int k = (int)fabsf(trim_fraction * (float)n_src);
  if (k < 1) k = 1;
  if (k > n_src) k = n_src;
  const float cap2 = max_corr_dist > 0.f ? max_corr_dist * max_corr_dist : -1.f;
  const double ratio = energy_ratio > 0.f ? (double)energy_ratio : 1.0;
  icp_pair* pr = (icp_pair*)malloc(sizeof(icp_pair) * (size_t)(n_src > 0 ? n_src : 1));
  int* nn = (int*)malloc(sizeof(int) * (size_t)(n_src > 0 ? n_src : 1));
  double E_old = (double)FLT_MAX, E = 0.0;
  int it = 0;
  for (;;) {
    for (int i = 0; i < n_src; ++i) {
      float x[3];
      orc_transform_point(T, src + 3 * i, x);
      float best = FLT_MAX;
      int bj = -1;
      for (int j = 0; j < n_tgt; ++j) {
        float d = orc_sqdist(x, tgt + 3 * j);
        if (d < best) {
          best = d;
          bj = j;
        }
      }
      pr[i].d2 = best;
      pr[i].i = i;
      nn[i] = bj;
    }
    int n_sel;
    if (cap2 >= 0.f) {
      n_sel = 0;
      for (int i = 0; i < n_src; ++i)
        if (pr[i].d2 <= cap2) pr[n_sel++] = pr[i];
    } else {
      if (k < n_src) qsort(pr, (size_t)n_src, sizeof(icp_pair), icp_cmp);
      n_sel = k;
    }
    double red[16] = {0};
    double e = 0.0;
    for (int s = 0; s < n_sel; ++s) {
      int i = pr[s].i, j = nn[i];
      if (j < 0) continue;
      const float* a = src + 3 * i;
      const float* m = tgt + 3 * j;
      red[0] += 1.0;
      for (int c = 0; c < 3; ++c) {
        red[1 + c] += a[c];
        red[4 + c] += m[c];
        for (int d = 0; d < 3; ++d) red[7 + 3 * c + d] += (double)a[c] * m[d];
      }
      e += (double)pr[s].d2;
    }
    E = red[0] >= 1.0 ? e / red[0] : 0.0;
    if (red[0] >= 1.0) {
      double n = red[0], sb[3], mb[3], S[3][3];
      for (int c = 0; c < 3; ++c) {
        sb[c] = red[1 + c] / n;
        mb[c] = red[4 + c] / n;
      }
      for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) S[c][d] = red[7 + 3 * c + d] - n * sb[c] * mb[d];
      double N[4][4] = {
          {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
          {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
          {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
          {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
      double q[4];
      jacobi_eig4(N, q);
      double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      if (nq > 0.0) {
        double w = q[0] / nq, x = q[1] / nq, y = q[2] / nq, z = q[3] / nq;
        double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                          {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                          {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
        for (int r = 0; r < 3; ++r) {
          double t = mb[r] - (R[r][0] * sb[0] + R[r][1] * sb[1] + R[r][2] * sb[2]);
          for (int c = 0; c < 3; ++c) T[4 * c + r] = (float)R[r][c];
          T[12 + r] = (float)t;
        }
        T[3] = T[7] = T[11] = 0.f;
        T[15] = 1.f;
      }
    }
    ++it;
    int go = (it < max_iterations) && (E / E_old < ratio);
    E_old = E;
    if (!go) break;
  }
  if (energy_out) *energy_out = (float)E;
  free(pr);
  free(nn);
  return it;
}

/* ---- hypothesis clustering -------------------------------------------------------------- */

/* Eigen 3.3.90 LU/InverseImpl.h (compute_inverse<.,.,3>): cofactor expansion along column 0. */
static float cof3(const float m[3][3], int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
}

void orc_pose_error(const float test[16], const float gt[16], const float sym[3], float* rot_err,
                    float* trans_err) {
  float a[3][3], g[3][3], inv[3][3], d[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a[i][j] = test[i + 4 * j];
      g[i][j] = gt[i + 4 * j];
    }
  /* utilities.cpp:523 testRot.inverse() */
  float c0 = cof3(a, 0, 0), c1 = cof3(a, 1, 0), c2 = cof3(a, 2, 0);
  float det = c0 * a[0][0] + (c1 * a[1][0] + c2 * a[2][0]);
  float invdet = 1.0f / det;
  inv[0][0] = c0 * invdet; inv[0][1] = c1 * invdet; inv[0][2] = c2 * invdet;
  for (int j = 0; j < 3; ++j) {
    inv[1][j] = cof3(a, j, 1) * invdet;
    inv[2][j] = cof3(a, j, 2) * invdet;
  }
  /* :524 rotdiff = testRot * gtRot (size-3 redux: x0 + (x1 + x2)) */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      d[i][j] = inv[i][0] * g[0][j] + (inv[i][1] * g[1][j] + inv[i][2] * g[2][j]);
  /* :525 Quaternionf(rotdiff): Eigen Geometry/Quaternion.h quaternionbase_assign_impl<.,3,3> */
  float q[4]; /* x y z w */
  float t = d[0][0] + (d[1][1] + d[2][2]);
  if (t > 0.0f) {
    t = sqrtf(t + 1.0f);
    q[3] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (d[2][1] - d[1][2]) * t;
    q[1] = (d[0][2] - d[2][0]) * t;
    q[2] = (d[1][0] - d[0][1]) * t;
  } else {
    int i = 0;
    if (d[1][1] > d[0][0]) i = 1;
    if (d[2][2] > d[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(d[i][i] - d[j][j] - d[k][k] + 1.0f);
    q[i] = 0.5f * t;
    t = 0.5f / t;
    q[3] = (d[k][j] - d[j][k]) * t;
    q[j] = (d[j][i] + d[i][j]) * t;
    q[k] = (d[k][i] + d[i][k]) * t;
  }
  /* :335-356 toEulerianAngle: float products / sums, then double */
  float x = q[0], y = q[1], z = q[2], w = q[3], e[3];
  double sinr = +2.0 * (double)(w * x + y * z);
  double cosr = +1.0 - 2.0 * (double)(x * x + y * y);
  e[0] = (float)atan2(sinr, cosr);
  double sinp = +2.0 * (double)(w * y - z * x);
  if (fabs(sinp) >= 1)
    e[1] = (float)copysign(M_PI / 2, sinp);
  else
    e[1] = (float)asin(sinp);
  double siny = +2.0 * (double)(w * z + x * y);
  double cosy = +1.0 - 2.0 * (double)(y * y + z * z);
  e[2] = (float)atan2(siny, cosy);
  /* :528-542 */
  for (int dim = 0; dim < 3; ++dim) {
    float v = e[dim] * 180.0f / (float)M_PI;
    v = fabsf(v);
    if (sym[dim] == 90) {
      v = fabsf(v - 90);
      v = fminf(v, 90 - v);
    } else if (sym[dim] == 180) {
      v = fminf(v, 180 - v);
    } else if (sym[dim] == 360) {
      v = 0;
    }
    e[dim] = v;
  }
  *rot_err = (e[0] + e[1] + e[2]) / 3;
  /* :545-547 pow(float, int) promotes to double */
  double dx = (double)(gt[12] - test[12]), dy = (double)(gt[13] - test[13]), dz = (double)(gt[14] - test[14]);
  *trans_err = (float)sqrt(dx * dx + dy * dy + dz * dz);
}

int orc_greedy_cluster(const float* T, const float* scores, int n, float best_score, float accept_fraction,
                       const float sym[3], float rot_thresh, float trans_thresh, int* rep_out, int* assign) {
  int* order = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int m = 0, n_rep = 0;
  float bar = accept_fraction * best_score;
  for (int i = 0; i < n; ++i) {
    if (assign) assign[i] = -1;
    if (scores[i] > bar) order[m++] = i;
  }
  /* stable insertion sort by score descending (test sizes are small) */
  for (int i = 1; i < m; ++i) {
    int v = order[i], j = i - 1;
    while (j >= 0 && scores[order[j]] < scores[v]) {
      order[j + 1] = order[j];
      --j;
    }
    order[j + 1] = v;
  }
  for (int c = 0; c < m; ++c) {
    int cand = order[c], hit = -1;
    for (int r = 0; r < n_rep && hit < 0; ++r) {
      float re, te;
      orc_pose_error(T + 16 * (size_t)cand, T + 16 * (size_t)rep_out[r], sym, &re, &te);
      if (re < rot_thresh && te < trans_thresh) hit = rep_out[r];
    }
    if (hit < 0) {
      rep_out[n_rep++] = cand;
      hit = cand;
    }
    if (assign) assign[cand] = hit;
  }
  free(order);
  return n_rep;
}

int orc_backproject(const void* image, int raw16, const unsigned char* mask, int rows, int cols,
                    const float K[9], double z_min, double z_max, float* xyz_out) {
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  int n = 0;
  for (int u = 0; u < rows; ++u)
    for (int v = 0; v < cols; ++v) {
      const size_t i = (size_t)u * cols + v;
      float depth;
      if (raw16) {
        unsigned short s = ((const unsigned short*)image)[i];
        s = (unsigned short)(s << 13 | s >> 3);       /* utilities.cpp:57 */
        depth = (float)s / 10000;                      /* :59 */
      } else {
        depth = ((const float*)image)[i];
      }
      if (mask && mask[i] == 0) depth = 0.f;           /* Segmentation.cpp:219 depthImage.mul(objMask) */
      if (depth > z_min && depth < z_max) {            /* utilities.cpp:197: double comparison */
        xyz_out[3 * (size_t)n] = (float)((v - cx) * depth / fx);
        xyz_out[3 * (size_t)n + 1] = (float)((u - cy) * depth / fy);
        xyz_out[3 * (size_t)n + 2] = depth;
        ++n;
      }
    }
  return n;
}

/* pcl::VoxelGrid<PointXYZRGB>::applyFilter as Segmentation.cpp:234-237 configures it (leaf 0.01 on
 * every axis, all data down-sampled, no minimum count), published algorithm of PCL 1.7 (PCL is not
 * vendored: SURVEY 8c): bounds over the finite points, inverse leaf = 1 / leaf (float),
 * min_b = (int)floor(min_p * inv), div_b = max_b - min_b + 1, voxel index
 * ijk0 + ijk1 * div_b0 + ijk2 * div_b0 * div_b1 with ijk = (int)(floor(p * inv) - (float)min_b);
 * points sorted by voxel index, one centroid per voxel = float sum of its points / count, leaves in
 * ascending voxel index.  PCL sorts with std::sort, which leaves the order of the points INSIDE a
 * voxel unspecified; this restatement (and the HIP path) add them in point-index order. */
typedef struct { long long key; int idx; } orc_vg_item;
static int orc_vg_cmp(const void* a, const void* b) {
  const orc_vg_item* x = (const orc_vg_item*)a;
  const orc_vg_item* y = (const orc_vg_item*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
int orc_voxel_grid(const float* xyz, int n, float leaf, float* out_xyz, int cap) {
  const float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  int finite = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + 3 * (size_t)i;
    if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
    ++finite;
    for (int k = 0; k < 3; ++k) {
      if (p[k] < mn[k]) mn[k] = p[k];
      if (p[k] > mx[k]) mx[k] = p[k];
    }
  }
  if (!finite) return 0;
  int min_b[3], div_b[3];
  for (int k = 0; k < 3; ++k) {
    min_b[k] = (int)floorf(mn[k] * inv);
    div_b[k] = (int)floorf(mx[k] * inv) - min_b[k] + 1;
  }
  orc_vg_item* it = (orc_vg_item*)malloc(sizeof(orc_vg_item) * (size_t)finite);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + 3 * (size_t)i;
    if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
    const int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
    const int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
    const int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
    it[m].key = (long long)i0 + (long long)i1 * div_b[0] + (long long)i2 * div_b[0] * (long long)div_b[1];
    it[m].idx = i;
    ++m;
  }
  qsort(it, (size_t)m, sizeof(orc_vg_item), orc_vg_cmp);
  int n_out = 0;
  for (int a = 0; a < m;) {
    int b = a;
    float c[3] = {0.f, 0.f, 0.f};
    while (b < m && it[b].key == it[a].key) {
      const float* p = xyz + 3 * (size_t)it[b].idx;
      c[0] += p[0];
      c[1] += p[1];
      c[2] += p[2];
      ++b;
    }
    const float cnt = (float)(b - a);
    if (n_out < cap) {
      out_xyz[3 * (size_t)n_out] = c[0] / cnt;
      out_xyz[3 * (size_t)n_out + 1] = c[1] / cnt;
      out_xyz[3 * (size_t)n_out + 2] = c[2] / cnt;
    }
    ++n_out;
    a = b;
  }
  free(it);
  return n_out;
}

/* Match4PCSBase::c_dist_pose / c_dist_pose_mean (base.cc:1616-1655): directed Hausdorff distance
 * (maximum, and the SUM the reference calls mean) from the hull under T1 to the hull under T2. */
void orc_pose_hausdorff(const float* hull, int n_hull, const float T1[16], const float T2[16], float* d_max,
                        float* d_sum) {
  float max_distance = 0, mean_distance = 0;
  for (int ii = 0; ii < n_hull; ++ii) {
    float min_distance = FLT_MAX;
    float p[3], q[3];
    orc_transform_point(T1, hull + 3 * (size_t)ii, p);
    for (int jj = 0; jj < n_hull; ++jj) {
      orc_transform_point(T2, hull + 3 * (size_t)jj, q);
      const float dist = sqrtf(orc_sqdist(p, q));
      if (dist < min_distance) min_distance = dist;
    }
    if (min_distance > max_distance) max_distance = min_distance;
    mean_distance += min_distance;
  }
  *d_max = max_distance;
  *d_sum = mean_distance;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- moving least squares (pcl::MovingLeastSquares as PPE/segmentation/Segmentation.cpp:239-246 sets it up) ----
 * PCL is not vendored: PCL 1.7's published MovingLeastSquares::computeMLSPointNormal (polynomial fit of
 * order 2, no upsampling, normals on), pcl::eigen33 / computeRoots, Eigen's unitOrthogonal and LLT, restated
 * step by step in double where PCL uses double.  Neighbours: brute force, squared distance in float (x, y, z
 * order) STRICTLY below (float)(radius^2) as FLANN's radius search, visited in index order (PCL: by
 * distance; the double sums differ in their last bits).  Points with fewer than 3 neighbours are dropped. */
static void mls_roots2(double b, double c, double r[3]) {
  r[0] = 0.0;
  double d = b * b - 4.0 * c;
  if (d < 0.0) d = 0.0;
  double sd = sqrt(d);
  r[2] = 0.5 * (b + sd);
  r[1] = 0.5 * (b - sd);
}

static void mls_roots3(const double m[6], double r[3]) { /* m = {xx, xy, xz, yy, yz, zz} */
  double c0 = m[0] * m[3] * m[5] + 2.0 * m[1] * m[2] * m[4] - m[0] * m[4] * m[4] - m[3] * m[2] * m[2] - m[5] * m[1] * m[1];
  double c1 = m[0] * m[3] - m[1] * m[1] + m[0] * m[5] - m[2] * m[2] + m[3] * m[5] - m[4] * m[4];
  double c2 = m[0] + m[3] + m[5];
  if (fabs(c0) < DBL_EPSILON) {
    mls_roots2(c2, c1, r);
    return;
  }
  const double s_inv3 = 1.0 / 3.0, s_sqrt3 = sqrt(3.0);
  double c2_over_3 = c2 * s_inv3;
  double a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0) a_over_3 = 0.0;
  double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
  double q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0) q = 0.0;
  double rho = sqrt(-a_over_3);
  double theta = atan2(sqrt(-q), half_b) * s_inv3;
  double ct = cos(theta), st = sin(theta);
  r[0] = c2_over_3 + 2.0 * rho * ct;
  r[1] = c2_over_3 - rho * (ct + s_sqrt3 * st);
  r[2] = c2_over_3 - rho * (ct - s_sqrt3 * st);
  if (r[0] >= r[1]) { double t = r[0]; r[0] = r[1]; r[1] = t; }
  if (r[1] >= r[2]) {
    double t = r[1]; r[1] = r[2]; r[2] = t;
    if (r[0] >= r[1]) { double u = r[0]; r[0] = r[1]; r[1] = u; }
  }
  if (r[0] <= 0.0) mls_roots2(c2, c1, r);
}

static void mls_eigen33(const double cov[6], double* eval, double evec[3]) {
  double scale = 0.0, m[6], r[3];
  for (int k = 0; k < 6; ++k) scale = fmax(scale, fabs(cov[k]));
  if (scale <= DBL_MIN) scale = 1.0;
  for (int k = 0; k < 6; ++k) m[k] = cov[k] / scale;
  mls_roots3(m, r);
  *eval = r[0] * scale;
  double r0[3] = {m[0] - r[0], m[1], m[2]}, r1[3] = {m[1], m[3] - r[0], m[4]}, r2[3] = {m[2], m[4], m[5] - r[0]};
  double v1[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
  double v2[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
  double v3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  double l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2], l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2],
         l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const double* v = v3;
  double l = l3;
  if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
  double s = sqrt(l);
  for (int k = 0; k < 3; ++k) evec[k] = v[k] / s;
}

static void mls_unit_orthogonal(const double n[3], double o[3]) {
  const double prec = 1e-12;
  int x_small = fabs(n[0]) <= fabs(n[2]) * prec, y_small = fabs(n[1]) <= fabs(n[2]) * prec;
  if (!x_small || !y_small) {
    double inv = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1]);
    o[0] = -n[1] * inv; o[1] = n[0] * inv; o[2] = 0.0;
  } else {
    double inv = 1.0 / sqrt(n[1] * n[1] + n[2] * n[2]);
    o[0] = 0.0; o[1] = -n[2] * inv; o[2] = n[1] * inv;
  }
}

static void mls_llt6(double A[6][6], double b[6]) {
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
    for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    double l = sqrt(d);
    A[j][j] = l;
    for (int i = j + 1; i < 6; ++i) {
      double s = A[i][j];
      for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k];
      A[i][j] = s / l;
    }
  }
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[i][k] * b[k];
    b[i] = s / A[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < 6; ++k) s -= A[k][i] * b[k];
    b[i] = s / A[i][i];
  }
}

int orc_mls(const float* xyz, int n, float radius, float* out_xyz, float* out_nrm, float* out_curv, int* out_idx,
            int cap) {
  const float r2 = (float)((double)radius * (double)radius);
  const double gauss = (double)radius * (double)radius;
  int* nb = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + 3 * (size_t)i;
    if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
    int cnt = 0;
    for (int j = 0; j < n; ++j) {
      const float* q = xyz + 3 * (size_t)j;
      float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
      float d2 = (dx * dx + dy * dy) + dz * dz;
      if (d2 < r2) nb[cnt++] = j;
    }
    if (cnt < 3) continue;
    double sx = 0, sy = 0, sz = 0;
    for (int k = 0; k < cnt; ++k) {
      const float* q = xyz + 3 * (size_t)nb[k];
      sx += (double)q[0]; sy += (double)q[1]; sz += (double)q[2];
    }
    double mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
    double cov[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < cnt; ++k) {
      const float* q = xyz + 3 * (size_t)nb[k];
      double dx = (double)q[0] - mx, dy = (double)q[1] - my, dz = (double)q[2] - mz;
      cov[0] += dx * dx; cov[1] += dx * dy; cov[2] += dx * dz; cov[3] += dy * dy; cov[4] += dy * dz; cov[5] += dz * dz;
    }
    double eval, nr[3];
    mls_eigen33(cov, &eval, nr);
    double dpl = -(nr[0] * mx + nr[1] * my + nr[2] * mz);
    double pt[3] = {(double)p[0], (double)p[1], (double)p[2]};
    double dist = pt[0] * nr[0] + pt[1] * nr[1] + pt[2] * nr[2] + dpl;
    for (int k = 0; k < 3; ++k) pt[k] -= dist * nr[k];
    float curv = (float)(cov[0] + cov[3] + cov[5]);
    if (curv != 0.f) curv = fabsf((float)(eval / (double)curv));
    double on[3] = {nr[0], nr[1], nr[2]};
    if (cnt >= 6) {
      double va[3], ua[3], A[6][6], b[6];
      mls_unit_orthogonal(nr, va);
      ua[0] = nr[1] * va[2] - nr[2] * va[1];
      ua[1] = nr[2] * va[0] - nr[0] * va[2];
      ua[2] = nr[0] * va[1] - nr[1] * va[0];
      memset(A, 0, sizeof(A));
      memset(b, 0, sizeof(b));
      for (int k = 0; k < cnt; ++k) {
        const float* q = xyz + 3 * (size_t)nb[k];
        double dx = (double)q[0] - pt[0], dy = (double)q[1] - pt[1], dz = (double)q[2] - pt[2];
        float sqr = (float)(dx * dx + dy * dy + dz * dz);
        double w = exp(-(double)sqr / gauss);
        double u = dx * ua[0] + dy * ua[1] + dz * ua[2], v = dx * va[0] + dy * va[1] + dz * va[2];
        double f = dx * nr[0] + dy * nr[1] + dz * nr[2];
        double t[6] = {1.0, v, v * v, u, u * v, u * u};
        for (int r = 0; r < 6; ++r) {
          double wr = w * t[r];
          b[r] += wr * f;
          for (int c = 0; c <= r; ++c) A[r][c] += wr * t[c];
        }
      }
      mls_llt6(A, b);
      if (isfinite(b[0])) {
        for (int k = 0; k < 3; ++k) {
          pt[k] += b[0] * nr[k];
          on[k] = nr[k] - b[3] * ua[k] - b[1] * va[k];
        }
      }
    }
    if (m < cap) {
      for (int k = 0; k < 3; ++k) {
        out_xyz[3 * (size_t)m + k] = (float)pt[k];
        if (out_nrm) out_nrm[3 * (size_t)m + k] = (float)on[k];
      }
      if (out_curv) out_curv[m] = curv;
      if (out_idx) out_idx[m] = i;
    }
    ++m;
  }
  free(nb);
  return m;
}
