/*
 * icp_oracle.c -- CPU restatement of the libpointmatcher chain laser_slam configures.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see icp_oracle.h for the full statement).
 *
 * Reference anchors (paths relative to /root/reference):
 *   module chain + parameters ...... laser_slam/configurations/icp_default.yaml:1-29
 *   default chain .................. laser_slam/src/laser_track.cpp:18-21 (icp_.setDefault())
 *   call sites ..................... laser_slam/src/laser_track.cpp:496,
 *                                    laser_slam/src/incremental_estimator.cpp:108
 *   rigid check / correct .......... laser_slam/include/laser_slam/common.hpp:136-149
 * Module semantics: libpointmatcher (ethz-asl/libpointmatcher, un-pinned in
 * dependencies.rosinstall:26-28) + libnabo (:23-25), restated from their published algorithm
 * (SURVEY.md Appendix A).
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off -fopenmp).
 */
#include "icp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small helpers */

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* column-major 4x4: M(r,c) = m[c*4+r] */
#define M4(m, r, c) ((m)[(c) * 4 + (r)])

static void mat4_identity(float* m) {
  memset(m, 0, 16 * sizeof(float));
  m[0] = m[5] = m[10] = m[15] = 1.0f;
}

/* out = a*b, plain float, k ascending, no contraction (compiled with -ffp-contract=off). */
static void mat4_mul(const float* a, const float* b, float* out) {
  float t[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float s = M4(a, r, 0) * M4(b, 0, c);
      s = s + M4(a, r, 1) * M4(b, 1, c);
      s = s + M4(a, r, 2) * M4(b, 2, c);
      s = s + M4(a, r, 3) * M4(b, 3, c);
      M4(t, r, c) = s;
    }
  memcpy(out, t, sizeof(t));
}

static inline float dist2_def(float dx, float dy, float dz) {
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

static inline void xform_def(const float* T, const float* p, float* o) {
  const float x = p[0], y = p[1], z = p[2];
  o[0] = fmaf(M4(T, 0, 2), z, fmaf(M4(T, 0, 1), y, fmaf(M4(T, 0, 0), x, M4(T, 0, 3))));
  o[1] = fmaf(M4(T, 1, 2), z, fmaf(M4(T, 1, 1), y, fmaf(M4(T, 1, 0), x, M4(T, 1, 3))));
  o[2] = fmaf(M4(T, 2, 2), z, fmaf(M4(T, 2, 1), y, fmaf(M4(T, 2, 0), x, M4(T, 2, 3))));
}

void lso_config_yaml(lso_config* c) { /* icp_default.yaml:1-27 */
  c->reading_sampling_prob = 0.5f;  /* :3  */
  c->surface_normal_knn = 10;       /* :7  */
  c->surface_normal_ratio = 0.5f;   /* module default (not in yaml) */
  c->trim_ratio = 0.75f;            /* :16 */
  c->max_iterations = 40;           /* :23 */
  c->min_diff_rot = 0.001f;         /* :25 */
  c->min_diff_trans = 0.01f;        /* :26 */
  c->smooth_length = 4;             /* :27 */
  c->accum_double = 0;
  c->num_threads = 1;
}

void lso_config_default(lso_config* c) { /* ICP::setDefault(), laser_track.cpp:20 */
  c->reading_sampling_prob = 0.75f;
  c->surface_normal_knn = 7;
  c->surface_normal_ratio = 0.5f;
  c->trim_ratio = 0.85f;
  c->max_iterations = 40;
  c->min_diff_rot = 0.001f;
  c->min_diff_trans = 0.001f;
  c->smooth_length = 3;
  c->accum_double = 0;
  c->num_threads = 1;
}

/* ------------------------------------------------------------------ RigidTransformation */

void lso_transform_points(const float T[16], const float* xyz1, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float o[3];
    xform_def(T, xyz1 + 4 * i, o);
    out[4 * i + 0] = o[0];
    out[4 * i + 1] = o[1];
    out[4 * i + 2] = o[2];
    out[4 * i + 3] = xyz1[4 * i + 3];
  }
}

void lso_rotate_normals(const float T[16], const float* nrm, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    const float x = nrm[3 * i], y = nrm[3 * i + 1], z = nrm[3 * i + 2];
    out[3 * i + 0] = fmaf(M4(T, 0, 2), z, fmaf(M4(T, 0, 1), y, M4(T, 0, 0) * x));
    out[3 * i + 1] = fmaf(M4(T, 1, 2), z, fmaf(M4(T, 1, 1), y, M4(T, 1, 0) * x));
    out[3 * i + 2] = fmaf(M4(T, 2, 2), z, fmaf(M4(T, 2, 1), y, M4(T, 2, 0) * x));
  }
}

static float det3(const float* T) {
  const float a = M4(T, 0, 0), b = M4(T, 0, 1), c = M4(T, 0, 2);
  const float d = M4(T, 1, 0), e = M4(T, 1, 1), f = M4(T, 1, 2);
  const float g = M4(T, 2, 0), h = M4(T, 2, 1), i = M4(T, 2, 2);
  return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}

/* RigidTransformation::checkParameters: |1 - det(R)| <= 1e-3 (via common.hpp:143). */
int lso_check_rigid(const float T[16]) {
  const float eps = 0.001f;
  return fabsf(1.0f - det3(T)) <= eps;
}

/* RigidTransformation::correctParameters (via common.hpp:146): normalise columns,
 * c0' = c1 x c2, c1' = c2 x c0', c2' = c2. */
void lso_correct_rigid(const float T[16], float out[16]) {
  float c[3][3];
  for (int k = 0; k < 3; ++k) {
    const float x = M4(T, 0, k), y = M4(T, 1, k), z = M4(T, 2, k);
    const float n = sqrtf(x * x + y * y + z * z);
    c[k][0] = x / n; c[k][1] = y / n; c[k][2] = z / n;
  }
  float c0[3] = {c[1][1] * c[2][2] - c[1][2] * c[2][1], c[1][2] * c[2][0] - c[1][0] * c[2][2],
                 c[1][0] * c[2][1] - c[1][1] * c[2][0]};
  float c1[3] = {c[2][1] * c0[2] - c[2][2] * c0[1], c[2][2] * c0[0] - c[2][0] * c0[2],
                 c[2][0] * c0[1] - c[2][1] * c0[0]};
  memcpy(out, T, 16 * sizeof(float));
  for (int r = 0; r < 3; ++r) {
    M4(out, r, 0) = c0[r];
    M4(out, r, 1) = c1[r];
    M4(out, r, 2) = c[2][r];
  }
}

/* ------------------------------------------------------------------ K1 RandomSampling */

int64_t lso_random_sampling(int64_t n, float prob, int64_t seed, int64_t* keep_idx) {
  if (seed >= 0) srand((unsigned)seed);
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float r = (float)rand() / (float)RAND_MAX;
    if (r < prob) keep_idx[m++] = i;
  }
  return m;
}

/* ------------------------------------------------------------------ nth_element on indices */

static inline void swap_i32(int32_t* a, int32_t* b) { int32_t t = *a; *a = *b; *b = t; }

/* Rearranges idx[lo,hi) so that idx[nth] holds the element that would be there if sorted by
 * coordinate `dim`, everything before is <= and everything after is >=. */
static void nth_element_idx(int32_t* idx, int64_t lo, int64_t hi, int64_t nth, const float* xyz1,
                            int dim) {
#define KEY(i) (xyz1[4 * (int64_t)(i) + dim])
  while (hi - lo > 1) {
    /* median of three pivot */
    int64_t mid = lo + (hi - lo) / 2;
    if (KEY(idx[mid]) < KEY(idx[lo])) swap_i32(&idx[mid], &idx[lo]);
    if (KEY(idx[hi - 1]) < KEY(idx[lo])) swap_i32(&idx[hi - 1], &idx[lo]);
    if (KEY(idx[hi - 1]) < KEY(idx[mid])) swap_i32(&idx[hi - 1], &idx[mid]);
    const float pivot = KEY(idx[mid]);
    int64_t i = lo, j = hi - 1;
    while (i <= j) {
      while (KEY(idx[i]) < pivot) ++i;
      while (KEY(idx[j]) > pivot) --j;
      if (i <= j) { swap_i32(&idx[i], &idx[j]); ++i; --j; }
    }
    /* [lo..j] <= pivot, [i..hi) >= pivot, (j,i) == pivot */
    if (nth <= j) hi = j + 1;
    else if (nth >= i) lo = i;
    else return;
  }
#undef KEY
}

/* ------------------------------------------------------------------ K2 SamplingSurfaceNormal */

/* Stable merge sort of idx[lo,hi) by coordinate `dim` (equal coordinates keep their order; -0 == +0).
 * std::nth_element leaves the order among equal keys and inside each half unspecified; the restatement
 * fixes it as "stable sort, split at the median" so that every implementation of the chain (this one,
 * the device filter) builds bit-identical boxes, box orders and rand() consumption orders. */
static void stable_sort_idx(int32_t* idx, int32_t* tmp, int64_t lo, int64_t hi, const float* xyz1, int dim) {
  const int64_t n = hi - lo;
  if (n < 2) return;
  if (n <= 16) { /* insertion sort */
    for (int64_t i = lo + 1; i < hi; ++i) {
      const int32_t v = idx[i];
      const float kv = xyz1[4 * (int64_t)v + dim];
      int64_t j = i;
      while (j > lo && xyz1[4 * (int64_t)idx[j - 1] + dim] > kv) { idx[j] = idx[j - 1]; --j; }
      idx[j] = v;
    }
    return;
  }
  const int64_t mid = lo + n / 2;
  stable_sort_idx(idx, tmp, lo, mid, xyz1, dim);
  stable_sort_idx(idx, tmp, mid, hi, xyz1, dim);
  int64_t a = lo, b = mid, o = lo;
  while (a < mid && b < hi) {
    if (xyz1[4 * (int64_t)idx[b] + dim] < xyz1[4 * (int64_t)idx[a] + dim]) tmp[o++] = idx[b++];
    else tmp[o++] = idx[a++];
  }
  while (a < mid) tmp[o++] = idx[a++];
  while (b < hi) tmp[o++] = idx[b++];
  memcpy(idx + lo, tmp + lo, sizeof(int32_t) * (size_t)n);
}

typedef struct ssn_ctx {
  const float* xyz1;
  int32_t* idx;
  int32_t* tmp;
  int knn;
  float ratio;
  float* out_xyz1;
  float* out_nrm;
  int64_t n_out;
  unsigned char* keep; /* per ORIGINAL index: in indicesToKeep */
  float* nrm_of;       /* per original index: the normal of its box */
} ssn_ctx;

/* Jacobi eigen-decomposition of a symmetric 3x3 (double).  V columns = eigenvectors. */
static void jacobi3(double a[3][3], double w[3], double v[3][3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i][j] = (i == j);
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-18 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { /* A <- A J */
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* A <- J^T A */
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
}

/* Numerical rank of a 3x3 float matrix the way Eigen::FullPivHouseholderQR::rank() counts it: upstream's
 * fuseRange drops a box unless C.fullPivHouseholderQr().rank() + 1 >= 3 (libpointmatcher, from knowledge -- see
 * "restatement choices" in icp_oracle.h).  Threshold: |pivot| > max |pivot| * eps * 3; a remaining corner whose
 * largest entry is <= eps * 3 times the first one ends the factorisation. */
static int rank3f(const float c[3][3]) {
  float m[3][3];
  memcpy(m, c, sizeof(m));
  const float prec = FLT_EPSILON * 3.0f;
  float pivots[3] = {0, 0, 0};
  float max_pivot = 0.f, first_corner = 0.f;
  int n_pivots = 3;
  for (int k = 0; k < 3; ++k) {
    /* full pivoting: the largest remaining entry goes to (k, k) */
    int br = k, bc = k;
    float big = -1.f;
    for (int i = k; i < 3; ++i)
      for (int j = k; j < 3; ++j)
        if (fabsf(m[i][j]) > big) { big = fabsf(m[i][j]); br = i; bc = j; }
    if (k == 0) first_corner = big;
    if (big <= first_corner * prec) { n_pivots = k; break; }
    for (int j = 0; j < 3; ++j) { float t = m[k][j]; m[k][j] = m[br][j]; m[br][j] = t; }
    for (int i = 0; i < 3; ++i) { float t = m[i][k]; m[i][k] = m[i][bc]; m[i][bc] = t; }
    /* Householder reflector of column k (rows k..2): beta = -sign(x0) |x| */
    float below = 0.f;
    for (int i = k + 1; i < 3; ++i) below += m[i][k] * m[i][k];
    const float x0 = m[k][k];
    float beta = x0, tau = 0.f, ess[3] = {0, 0, 0};
    if (below > FLT_MIN) {
      beta = sqrtf(x0 * x0 + below);
      if (x0 >= 0.f) beta = -beta;
      for (int i = k + 1; i < 3; ++i) ess[i] = m[i][k] / (x0 - beta);
      tau = (beta - x0) / beta;
    }
    pivots[k] = fabsf(beta);
    if (pivots[k] > max_pivot) max_pivot = pivots[k];
    for (int j = k + 1; j < 3; ++j) {
      float dot = m[k][j];
      for (int i = k + 1; i < 3; ++i) dot += ess[i] * m[i][j];
      m[k][j] -= tau * dot;
      for (int i = k + 1; i < 3; ++i) m[i][j] -= tau * dot * ess[i];
    }
  }
  int r = 0;
  for (int k = 0; k < n_pivots; ++k) r += (pivots[k] > max_pivot * prec);
  return r;
}

static void ssn_fuse(ssn_ctx* s, int64_t first, int64_t last) {
  const int64_t cnt = last - first;
  if (cnt <= 0) return;
  float mean[3] = {0, 0, 0};
  for (int64_t i = first; i < last; ++i)
    for (int d = 0; d < 3; ++d) mean[d] += s->xyz1[4 * (int64_t)s->idx[i] + d];
  for (int d = 0; d < 3; ++d) mean[d] /= (float)cnt;
  float C[3][3] = {{0}};
  for (int64_t i = first; i < last; ++i) {
    float nn[3];
    for (int d = 0; d < 3; ++d) nn[d] = s->xyz1[4 * (int64_t)s->idx[i] + d] - mean[d];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) C[a][b] += nn[a] * nn[b];
  }
  if (rank3f(C) + 1 < 3) return; /* box dropped: too degenerate for a normal */
  double a[3][3], w[3], v[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = C[i][j];
  jacobi3(a, w, v);
  int k = 0;
  if (w[1] < w[k]) k = 1;
  if (w[2] < w[k]) k = 2;
  double nx = v[0][k], ny = v[1][k], nz = v[2][k];
  const double nl = sqrt(nx * nx + ny * ny + nz * nz);
  nx /= nl; ny /= nl; nz /= nl;
  for (int64_t i = first; i < last; ++i) { /* samplingMethod 0: the draws are taken in box-traversal order */
    const float r = (float)rand() / (float)RAND_MAX;
    if (r < s->ratio) {
      /* indicesToKeep.push_back(k); normals->col(k) = normal -- the point keeps its ORIGINAL column until the
       * final compaction (lso_sampling_surface_normal below) */
      const int64_t src = s->idx[i];
      s->keep[src] = 1;
      s->nrm_of[3 * src + 0] = (float)nx;
      s->nrm_of[3 * src + 1] = (float)ny;
      s->nrm_of[3 * src + 2] = (float)nz;
    }
  }
}

static void ssn_build(ssn_ctx* s, int64_t first, int64_t last, const float* minb,
                      const float* maxb) {
  const int64_t count = last - first;
  if (count <= s->knn) { ssn_fuse(s, first, last); return; }
  int cut = 0;
  float ext = maxb[0] - minb[0];
  for (int d = 1; d < 3; ++d)
    if (maxb[d] - minb[d] > ext) { ext = maxb[d] - minb[d]; cut = d; }
  const int64_t right = count / 2, left = count - right;
  stable_sort_idx(s->idx, s->tmp, first, last, s->xyz1, cut);
  const float cutval = s->xyz1[4 * (int64_t)s->idx[first + left] + cut];
  float lmax[3] = {maxb[0], maxb[1], maxb[2]}, rmin[3] = {minb[0], minb[1], minb[2]};
  lmax[cut] = cutval;
  rmin[cut] = cutval;
  ssn_build(s, first, first + left, minb, lmax);
  ssn_build(s, first + left, last, rmin, maxb);
}

int64_t lso_sampling_surface_normal(const float* xyz1, int64_t n, int knn, float ratio,
                                    int64_t seed, float* out_xyz1, float* out_normals) {
  if (n <= 0) return 0;
  if (seed >= 0) srand((unsigned)seed);
  ssn_ctx s;
  s.xyz1 = xyz1; s.knn = knn; s.ratio = ratio;
  s.out_xyz1 = out_xyz1; s.out_nrm = out_normals; s.n_out = 0;
  s.idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  s.tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  float minb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, maxb[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = 0; i < n; ++i) {
    s.idx[i] = (int32_t)i;
    for (int d = 0; d < 3; ++d) {
      const float v = xyz1[4 * i + d];
      if (v < minb[d]) minb[d] = v;
      if (v > maxb[d]) maxb[d] = v;
    }
  }
  s.keep = (unsigned char*)calloc((size_t)n, 1);
  s.nrm_of = (float*)malloc(sizeof(float) * 3 * (size_t)n);
  ssn_build(&s, 0, n, minb, maxb);
  /* "Bring the data we keep to the front of the arrays": upstream sorts indicesToKeep ascending before it compacts
   * (std::sort in inPlaceFilter; from knowledge of libpointmatcher, restatement choice 10 in icp_oracle.h), so the
   * filtered cloud is in ORIGINAL index order, every point with its box's normal */
  for (int64_t i = 0; i < n; ++i)
    if (s.keep[i]) {
      const int64_t o = s.n_out++;
      memcpy(s.out_xyz1 + 4 * o, xyz1 + 4 * i, 4 * sizeof(float));
      memcpy(s.out_nrm + 3 * o, s.nrm_of + 3 * i, 3 * sizeof(float));
    }
  free(s.keep);
  free(s.nrm_of);
  free(s.idx);
  free(s.tmp);
  return s.n_out;
}

/* ------------------------------------------------------------------ K4/K6 kd-tree (libnabo) */

#define KD_BUCKET 8

typedef struct kd_node {
  int32_t a;     /* inner: left child,  leaf: first point  */
  int32_t b;     /* inner: right child, leaf: point count  */
  float cutval;
  int32_t dim;   /* -1 for leaves */
} kd_node;

typedef struct kd_tree {
  kd_node* nodes;
  int64_t n_nodes, cap_nodes;
  float* pts;      /* leaf-ordered x,y,z,(bitcast original index) */
  int32_t* idx;
  int64_t n;
  const float* src;
} kd_tree;

static int32_t kd_new_node(kd_tree* t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
  }
  return (int32_t)t->n_nodes++;
}

static int32_t kd_build_rec(kd_tree* t, int64_t lo, int64_t hi) {
  const int32_t me = kd_new_node(t);
  if (hi - lo <= KD_BUCKET) {
    t->nodes[me].a = (int32_t)lo;
    t->nodes[me].b = (int32_t)(hi - lo);
    t->nodes[me].dim = -1;
    t->nodes[me].cutval = 0.f;
    return me;
  }
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = lo; i < hi; ++i)
    for (int d = 0; d < 3; ++d) {
      const float v = t->src[4 * (int64_t)t->idx[i] + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  int cut = 0;
  for (int d = 1; d < 3; ++d)
    if (mx[d] - mn[d] > mx[cut] - mn[cut]) cut = d;
  const int64_t mid = lo + (hi - lo) / 2;
  nth_element_idx(t->idx, lo, hi, mid, t->src, cut);
  const float cutval = t->src[4 * (int64_t)t->idx[mid] + cut];
  const int32_t l = kd_build_rec(t, lo, mid);
  const int32_t r = kd_build_rec(t, mid, hi);
  t->nodes[me].a = l;
  t->nodes[me].b = r;
  t->nodes[me].dim = cut;
  t->nodes[me].cutval = cutval; /* left: <= cutval, right: >= cutval */
  return me;
}

void* lso_kdtree_build(const float* ref_xyz1, int64_t nr) {
  kd_tree* t = (kd_tree*)calloc(1, sizeof(kd_tree));
  t->n = nr;
  t->src = ref_xyz1;
  if (nr <= 0) return t;
  t->idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)nr);
  for (int64_t i = 0; i < nr; ++i) t->idx[i] = (int32_t)i;
  kd_build_rec(t, 0, nr);
  t->pts = (float*)malloc(sizeof(float) * 4 * (size_t)nr);
  for (int64_t i = 0; i < nr; ++i) {
    const int64_t s = t->idx[i];
    t->pts[4 * i + 0] = ref_xyz1[4 * s + 0];
    t->pts[4 * i + 1] = ref_xyz1[4 * s + 1];
    t->pts[4 * i + 2] = ref_xyz1[4 * s + 2];
    memcpy(&t->pts[4 * i + 3], &t->idx[i], sizeof(int32_t));
  }
  t->src = NULL;
  return t;
}

void lso_kdtree_free(void* tree) {
  kd_tree* t = (kd_tree*)tree;
  if (!t) return;
  free(t->nodes); free(t->pts); free(t->idx); free(t);
}

typedef struct kd_query {
  const kd_tree* t;
  float q[3];
  float best;
  int32_t best_id;
} kd_query;

/* rd = squared distance from q to the cell of `node` (double, lower bound).  A cell is pruned only
 * when rd*(1-4e-7) > best so that float rounding of the candidate distance can never hide a
 * point whose computed dist^2 would have been smaller. */
static void kd_search(kd_query* s, int32_t node, double rd, double off[3]) {
  const kd_node* nd = &s->t->nodes[node];
  if (nd->dim < 0) {
    const float* p = s->t->pts + 4 * (int64_t)nd->a;
    for (int i = 0; i < nd->b; ++i, p += 4) {
      const float d = dist2_def(s->q[0] - p[0], s->q[1] - p[1], s->q[2] - p[2]);
      if (d < s->best) {
        s->best = d;
        memcpy(&s->best_id, p + 3, sizeof(int32_t));
      }
    }
    return;
  }
  const int cd = nd->dim;
  const double old_off = off[cd];
  const double new_off = (double)s->q[cd] - (double)nd->cutval;
  int32_t near_child, far_child;
  if (new_off > 0) { near_child = nd->b; far_child = nd->a; }
  else             { near_child = nd->a; far_child = nd->b; }
  kd_search(s, near_child, rd, off);
  const double new_rd = rd - old_off * old_off + new_off * new_off;
  if (new_rd * (1.0 - 4e-7) <= (double)s->best) {
    off[cd] = new_off;
    kd_search(s, far_child, new_rd, off);
    off[cd] = old_off;
  }
}

void lso_kdtree_nn(const void* tree, const float* q_xyz1, int64_t nq, int32_t* ids, float* d2,
                   int num_threads) {
  const kd_tree* t = (const kd_tree*)tree;
  if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(dynamic, 2048) num_threads(num_threads)
  for (int64_t i = 0; i < nq; ++i) {
    kd_query s;
    s.t = t;
    s.q[0] = q_xyz1[4 * i]; s.q[1] = q_xyz1[4 * i + 1]; s.q[2] = q_xyz1[4 * i + 2];
    s.best = INFINITY;
    s.best_id = -1; /* InvalidId */
    if (t->n > 0) {
      double off[3] = {0, 0, 0};
      kd_search(&s, 0, 0.0, off);
    }
    ids[i] = s.best_id;
    d2[i] = s.best;
  }
}

void lso_brute_nn(const float* ref, int64_t nr, const float* q, int64_t nq, int32_t* ids,
                  float* d2) {
  for (int64_t i = 0; i < nq; ++i) {
    float best = INFINITY;
    int32_t bid = -1;
    for (int64_t j = 0; j < nr; ++j) {
      const float d = dist2_def(q[4 * i] - ref[4 * j], q[4 * i + 1] - ref[4 * j + 1],
                                q[4 * i + 2] - ref[4 * j + 2]);
      if (d < best) { best = d; bid = (int32_t)j; }
    }
    ids[i] = bid;
    d2[i] = best;
  }
}

/* ------------------------------------------------------------------ K7 TrimmedDist */

static void nth_float(float* v, int64_t lo, int64_t hi, int64_t nth) {
  while (hi - lo > 1) {
    int64_t mid = lo + (hi - lo) / 2;
    float a = v[lo], b = v[mid], c = v[hi - 1];
    float pivot = (a < b) ? ((b < c) ? b : (a < c ? c : a)) : ((a < c) ? a : (b < c ? c : b));
    int64_t i = lo, j = hi - 1;
    while (i <= j) {
      while (v[i] < pivot) ++i;
      while (v[j] > pivot) --j;
      if (i <= j) { float t = v[i]; v[i] = v[j]; v[j] = t; ++i; --j; }
    }
    if (nth <= j) hi = j + 1;
    else if (nth >= i) lo = i;
    else return;
  }
}

int lso_trim_limit(const float* d2, int64_t n, float ratio, float* limit) {
  float* v = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i)
    if (d2[i] != INFINITY) v[m++] = d2[i];
  if (m == 0) { free(v); return LSO_NO_CONVERGENCE; } /* "no outlier to filter" */
  int64_t k = (int64_t)((float)m * ratio); /* values.size() * ratio, truncated */
  if (k >= m) k = m - 1;
  nth_float(v, 0, m, k);
  *limit = v[k];
  free(v);
  return LSO_OK;
}

/* ------------------------------------------------------------------ K8 PointToPlane */

static int llt_solve6f(const double A[36], const double b[6], float x[6]) {
  float L[6][6], y[6], bf[6];
  memset(L, 0, sizeof(L));
  for (int i = 0; i < 6; ++i) bf[i] = (float)b[i];
  for (int j = 0; j < 6; ++j) {
    float s = (float)A[j * 6 + j];
    for (int k = 0; k < j; ++k) s = s - L[j][k] * L[j][k];
    if (!(s > 0.f)) return LSO_NO_CONVERGENCE;
    L[j][j] = sqrtf(s);
    for (int i = j + 1; i < 6; ++i) {
      float t = (float)A[i * 6 + j];
      for (int k = 0; k < j; ++k) t = t - L[i][k] * L[j][k];
      L[i][j] = t / L[j][j];
    }
  }
  for (int i = 0; i < 6; ++i) {
    float t = bf[i];
    for (int k = 0; k < i; ++k) t = t - L[i][k] * y[k];
    y[i] = t / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    float t = y[i];
    for (int k = i + 1; k < 6; ++k) t = t - L[k][i] * x[k];
    x[i] = t / L[i][i];
  }
  for (int i = 0; i < 6; ++i)
    if (isnan(x[i])) return LSO_NO_CONVERGENCE;
  return LSO_OK;
}

/* dT from x = [rotation vector; translation]: Eigen AngleAxis(|r|, r/|r|).toRotationMatrix(). */
static void delta_from_x(const float x[6], float dT[16]) {
  mat4_identity(dT);
  const float ang = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (ang > 0.f && isfinite(ang)) {
    const float ax = x[0] / ang, ay = x[1] / ang, az = x[2] / ang;
    const float s = sinf(ang), c = cosf(ang);
    const float sx = s * ax, sy = s * ay, sz = s * az;
    const float c1x = (1.f - c) * ax, c1y = (1.f - c) * ay, c1z = (1.f - c) * az;
    float tmp;
    tmp = c1x * ay; M4(dT, 0, 1) = tmp - sz; M4(dT, 1, 0) = tmp + sz;
    tmp = c1x * az; M4(dT, 0, 2) = tmp + sy; M4(dT, 2, 0) = tmp - sy;
    tmp = c1y * az; M4(dT, 1, 2) = tmp - sx; M4(dT, 2, 1) = tmp + sx;
    M4(dT, 0, 0) = c1x * ax + c;
    M4(dT, 1, 1) = c1y * ay + c;
    M4(dT, 2, 2) = c1z * az + c;
  } /* else rotation := I (x == 0 gives NaN axis upstream, replaced by identity) */
  M4(dT, 0, 3) = x[3];
  M4(dT, 1, 3) = x[4];
  M4(dT, 2, 3) = x[5];
}

int lso_point_to_plane(const float* p_xyz1, const float* ref_xyz1, const float* ref_nrm,
                       const int32_t* ids, const float* d2, float limit, int64_t nq,
                       int accum_double, double A[36], double b[6], double x[6], float dT[16],
                       int64_t* n_used) {
  double Ad[21], bd[6];
  float Af[21], bfl[6];
  memset(Ad, 0, sizeof(Ad)); memset(bd, 0, sizeof(bd));
  memset(Af, 0, sizeof(Af)); memset(bfl, 0, sizeof(bfl));
  int64_t used = 0;
  for (int64_t i = 0; i < nq; ++i) {
    if (!(d2[i] <= limit) || ids[i] < 0) continue; /* w == 0 */
    const float* p = p_xyz1 + 4 * i;
    const float* q = ref_xyz1 + 4 * (int64_t)ids[i];
    const float* n = ref_nrm + 3 * (int64_t)ids[i];
    float J[6];
    J[0] = p[1] * n[2] - p[2] * n[1];
    J[1] = p[2] * n[0] - p[0] * n[2];
    J[2] = p[0] * n[1] - p[1] * n[0];
    J[3] = n[0]; J[4] = n[1]; J[5] = n[2];
    const float r = (p[0] - q[0]) * n[0] + (p[1] - q[1]) * n[1] + (p[2] - q[2]) * n[2];
    int k = 0;
    if (accum_double) {
      for (int a = 0; a < 6; ++a) {
        for (int c = a; c < 6; ++c) Ad[k++] += (double)J[a] * (double)J[c];
        bd[a] -= (double)J[a] * (double)r;
      }
    } else {
      for (int a = 0; a < 6; ++a) {
        for (int c = a; c < 6; ++c) Af[k++] += J[a] * J[c];
        bfl[a] -= J[a] * r;
      }
    }
    ++used;
  }
  *n_used = used;
  int k = 0;
  for (int a = 0; a < 6; ++a) {
    for (int c = a; c < 6; ++c, ++k) {
      const double v = accum_double ? Ad[k] : (double)Af[k];
      A[a * 6 + c] = v;
      A[c * 6 + a] = v;
    }
    b[a] = accum_double ? bd[a] : (double)bfl[a];
  }
  if (used == 0) return LSO_NO_CONVERGENCE; /* "no point to minimize" */
  float xf[6];
  const int rc = llt_solve6f(A, b, xf);
  if (rc != LSO_OK) return rc;
  for (int i = 0; i < 6; ++i) x[i] = xf[i];
  delta_from_x(xf, dT);
  return LSO_OK;
}

/* ------------------------------------------------------------------ K9 checkers */

/* Eigen Quaternion(Matrix3) */
static void quat_from_R(const float* T, float q[4] /* w,x,y,z */) {
  const float m00 = M4(T, 0, 0), m11 = M4(T, 1, 1), m22 = M4(T, 2, 2);
  float t = m00 + m11 + m22;
  if (t > 0.f) {
    t = sqrtf(t + 1.0f);
    q[0] = 0.5f * t;
    t = 0.5f / t;
    q[1] = (M4(T, 2, 1) - M4(T, 1, 2)) * t;
    q[2] = (M4(T, 0, 2) - M4(T, 2, 0)) * t;
    q[3] = (M4(T, 1, 0) - M4(T, 0, 1)) * t;
  } else {
    int i = 0;
    if (m11 > m00) i = 1;
    if (m22 > M4(T, i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(M4(T, i, i) - M4(T, j, j) - M4(T, k, k) + 1.0f);
    q[1 + i] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (M4(T, k, j) - M4(T, j, k)) * t;
    q[1 + j] = (M4(T, j, i) + M4(T, i, j)) * t;
    q[1 + k] = (M4(T, k, i) + M4(T, i, k)) * t;
  }
}

/* Eigen >= 3.3 angularDistance: d = a * conj(b); 2*atan2(|d.vec|, |d.w|). */
static float quat_angular_distance(const float a[4], const float b[4]) {
  const float bw = b[0], bx = -b[1], by = -b[2], bz = -b[3];
  const float w = a[0] * bw - a[1] * bx - a[2] * by - a[3] * bz;
  const float x = a[0] * bx + a[1] * bw + a[2] * bz - a[3] * by;
  const float y = a[0] * by + a[2] * bw + a[3] * bx - a[1] * bz;
  const float z = a[0] * bz + a[3] * bw + a[1] * by - a[2] * bx;
  return 2.0f * atan2f(sqrtf(x * x + y * y + z * z), fabsf(w));
}

/* Test hook: the rotation metric of DifferentialTransformationChecker between two 4x4 transforms (column major),
 * i.e. Eigen's Quaternion(R_a).angularDistance(Quaternion(R_b)) as restated above.  Pinned against scipy's
 * Rotation.magnitude() by tests/golden/make_golden_independent.py. */
float lso_rotation_distance(const float Ta[16], const float Tb[16]) {
  float qa[4], qb[4];
  quat_from_R(Ta, qa);
  quat_from_R(Tb, qb);
  return quat_angular_distance(qa, qb);
}

typedef struct checkers {
  int counter, max_iter, smooth;
  float lim_rot, lim_trans;
  float (*quats)[4];
  float (*trans)[3];
  int n_hist;
} checkers;

static void checkers_init(checkers* c, const lso_config* cfg, const float* T) {
  c->counter = 0;
  c->max_iter = cfg->max_iterations;
  c->smooth = cfg->smooth_length;
  c->lim_rot = cfg->min_diff_rot;
  c->lim_trans = cfg->min_diff_trans;
  c->quats = malloc(sizeof(float[4]) * (size_t)(cfg->max_iterations + 2));
  c->trans = malloc(sizeof(float[3]) * (size_t)(cfg->max_iterations + 2));
  c->n_hist = 0;
  quat_from_R(T, c->quats[0]);
  for (int d = 0; d < 3; ++d) c->trans[0][d] = M4(T, d, 3);
  c->n_hist = 1;
}

/* returns LSO_OK / LSO_NO_CONVERGENCE; *iterate cleared to stop; *by_diff set if differential. */
static int checkers_check(checkers* c, const float* T, int* iterate, int* by_diff) {
  c->counter++;
  if (c->counter >= c->max_iter) { *iterate = 0; return LSO_OK; } /* MaxNumIterationsReached */
  quat_from_R(T, c->quats[c->n_hist]);
  for (int d = 0; d < 3; ++d) c->trans[c->n_hist][d] = M4(T, d, 3);
  c->n_hist++;
  float cv0 = 0.f, cv1 = 0.f;
  if (c->n_hist > c->smooth) {
    for (int i = c->n_hist - 1; i >= c->n_hist - c->smooth; --i) {
      cv0 += fabsf(quat_angular_distance(c->quats[i], c->quats[i - 1]));
      const float dx = c->trans[i][0] - c->trans[i - 1][0];
      const float dy = c->trans[i][1] - c->trans[i - 1][1];
      const float dz = c->trans[i][2] - c->trans[i - 1][2];
      cv1 += fabsf(sqrtf(dx * dx + dy * dy + dz * dz));
    }
    cv0 /= (float)c->smooth;
    cv1 /= (float)c->smooth;
    if (cv0 < c->lim_rot && cv1 < c->lim_trans) { *iterate = 0; *by_diff = 1; }
  }
  if (isnan(cv0) || isnan(cv1)) return LSO_NO_CONVERGENCE;
  return LSO_OK;
}

static void checkers_free(checkers* c) { free(c->quats); free(c->trans); }

/* ------------------------------------------------------------------ ICP::compute */

int lso_icp_compute(const lso_config* cfg, const float* reading_xyz1, int64_t nq,
                    const float* ref_xyz1, const float* ref_nrm, int64_t nr,
                    const float T_init[16], float T_out[16], lso_stats* stats,
                    lso_iter_trace* trace, int trace_cap) {
  lso_stats st;
  memset(&st, 0, sizeof(st));
  memcpy(T_out, T_init, 16 * sizeof(float));
  if (nq <= 0 || nr <= 0) { if (stats) *stats = st; return LSO_NO_CONVERGENCE; }
  double t0 = now_ms();

  /* step 2: centre the reference on its mean (double sum, rounded to float). */
  double sm[3] = {0, 0, 0};
  for (int64_t i = 0; i < nr; ++i)
    for (int d = 0; d < 3; ++d) sm[d] += ref_xyz1[4 * i + d];
  float mean[3];
  for (int d = 0; d < 3; ++d) mean[d] = (float)(sm[d] / (double)nr);
  float* ref = (float*)malloc(sizeof(float) * 4 * (size_t)nr);
  for (int64_t i = 0; i < nr; ++i) {
    for (int d = 0; d < 3; ++d) ref[4 * i + d] = ref_xyz1[4 * i + d] - mean[d];
    ref[4 * i + 3] = ref_xyz1[4 * i + 3];
  }
  float T_refIn_refMean[16];
  mat4_identity(T_refIn_refMean);
  for (int d = 0; d < 3; ++d) M4(T_refIn_refMean, d, 3) = mean[d];

  /* step 3: matcher->init */
  void* tree = lso_kdtree_build(ref, nr);

  /* step 5: T_refMean_dataIn = T_refIn_refMean^-1 * T_init ; reading <- that * reading */
  float T_refMean_dataIn[16];
  memcpy(T_refMean_dataIn, T_init, 16 * sizeof(float));
  for (int d = 0; d < 3; ++d) M4(T_refMean_dataIn, d, 3) = M4(T_init, d, 3) - mean[d];
  float* reading = (float*)malloc(sizeof(float) * 4 * (size_t)nq);
  lso_transform_points(T_refMean_dataIn, reading_xyz1, nq, reading);
  st.t_build_ms = now_ms() - t0;
  t0 = now_ms();

  /* step 6: the loop */
  float T_iter[16];
  mat4_identity(T_iter);
  checkers ck;
  checkers_init(&ck, cfg, T_iter);
  float* step = (float*)malloc(sizeof(float) * 4 * (size_t)nq);
  int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)nq);
  float* d2 = (float*)malloc(sizeof(float) * (size_t)nq);
  int iterate = 1, by_diff = 0, rc = LSO_OK, it = 0;
  while (iterate) {
    lso_transform_points(T_iter, reading, nq, step);                       /* 6a */
    lso_kdtree_nn(tree, step, nq, ids, d2, cfg->num_threads);              /* 6b */
    float limit;
    rc = lso_trim_limit(d2, nq, cfg->trim_ratio, &limit);                  /* 6c */
    if (rc != LSO_OK) break;
    double A[36], b[6], x[6];
    float dT[16];
    int64_t used = 0;
    rc = lso_point_to_plane(step, ref, ref_nrm, ids, d2, limit, nq, cfg->accum_double, A, b, x,
                            dT, &used);                                    /* 6d */
    if (rc != LSO_OK) break;
    mat4_mul(dT, T_iter, T_iter);
    if (trace && it < trace_cap) {
      memcpy(trace[it].T_iter, T_iter, sizeof(T_iter));
      trace[it].limit = limit;
      trace[it].n_used = used;
      memcpy(trace[it].A, A, sizeof(A));
      memcpy(trace[it].b, b, sizeof(b));
      memcpy(trace[it].x, x, sizeof(x));
    }
    st.final_limit = limit;
    st.final_n_used = used;
    ++it;
    rc = checkers_check(&ck, T_iter, &iterate, &by_diff);                  /* 6e */
    if (rc != LSO_OK) break;
  }
  st.iterations = it;
  st.converged = by_diff;
  st.t_loop_ms = now_ms() - t0;
  checkers_free(&ck);

  if (rc == LSO_OK) { /* step 7 */
    float tmp[16];
    mat4_mul(T_iter, T_refMean_dataIn, tmp);
    mat4_mul(T_refIn_refMean, tmp, T_out);
  }
  lso_kdtree_free(tree);
  free(ref); free(reading); free(step); free(ids); free(d2);
  if (stats) *stats = st;
  return rc;
}

int lso_icp_compute_full(const lso_config* cfg, const float* reading_xyz1, int64_t nq,
                         const float* ref_xyz1, int64_t nr, const float T_init[16],
                         int64_t seed, float T_out[16], lso_stats* stats) {
  memcpy(T_out, T_init, 16 * sizeof(float));
  if (nq <= 0 || nr <= 0) return LSO_NO_CONVERGENCE;
  if (seed >= 0) srand((unsigned)seed);
  const double t0 = now_ms();
  /* step 1: reference filters (K2) */
  float* rf = (float*)malloc(sizeof(float) * 4 * (size_t)nr);
  float* rn = (float*)malloc(sizeof(float) * 3 * (size_t)nr);
  const int64_t nrf = lso_sampling_surface_normal(ref_xyz1, nr, cfg->surface_normal_knn,
                                                  cfg->surface_normal_ratio, -1, rf, rn);
  /* step 4: reading filters (K1) */
  int64_t* keep = (int64_t*)malloc(sizeof(int64_t) * (size_t)nq);
  int64_t nqf;
  if (cfg->reading_sampling_prob < 0.f) { /* no readingDataPointsFilters module: every point, no rand() call */
    nqf = nq;
    for (int64_t i = 0; i < nq; ++i) keep[i] = i;
  } else {
    nqf = lso_random_sampling(nq, cfg->reading_sampling_prob, -1, keep);
  }
  float* qf = (float*)malloc(sizeof(float) * 4 * (size_t)(nqf > 0 ? nqf : 1));
  for (int64_t i = 0; i < nqf; ++i) memcpy(qf + 4 * i, reading_xyz1 + 4 * keep[i], 16);
  const double tf = now_ms() - t0;
  const int rc = lso_icp_compute(cfg, qf, nqf, rf, rn, nrf, T_init, T_out, stats, NULL, 0);
  if (stats) stats->t_filter_ms = tf;
  free(rf); free(rn); free(keep); free(qf);
  return rc;
}

/* ------------------------------------------------------------------ local-map maintenance (N4) */

int64_t lso_cylinder_filter(const float* xyz1, int64_t n, const float center[3], double radius_m,
                            double height_m, int remove_point_inside, float* out_xyz1) {
  const double radius_squared = radius_m * radius_m; /* pow(radius_m, 2.0) */
  const double height_halved_m = height_m / 2.0;
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float dx = xyz1[4 * i] - center[0], dy = xyz1[4 * i + 1] - center[1];
    const float dz = fabsf(xyz1[4 * i + 2] - center[2]);
    const double r2 = (double)dx * (double)dx + (double)dy * (double)dy; /* pow(float, 2.0) promotes to double */
    const int inside = r2 <= radius_squared && (double)dz <= height_halved_m;
    const int outside = r2 >= radius_squared || (double)dz >= height_halved_m;
    if (remove_point_inside ? outside : inside) memcpy(out_xyz1 + 4 * (m++), xyz1 + 4 * i, 16);
  }
  return m;
}

typedef struct { int32_t idx; int32_t pt; } lso_vox_pair;
static int lso_vox_cmp(const void* a, const void* b) {
  const lso_vox_pair* x = (const lso_vox_pair*)a; const lso_vox_pair* y = (const lso_vox_pair*)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return x->pt < y->pt ? -1 : (x->pt > y->pt); /* input order inside a voxel */
}

int64_t lso_voxel_grid(const float* xyz1, int64_t n, const float leaf[3], int min_points, float* out_xyz1) {
  if (n <= 0) return 0;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      const float v = xyz1[4 * i + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  float inv[3];
  int minb[3], divb[3];
  for (int d = 0; d < 3; ++d) {
    inv[d] = 1.0f / leaf[d];
    minb[d] = (int)floorf(mn[d] * inv[d]);
    const int maxb = (int)floorf(mx[d] * inv[d]);
    divb[d] = maxb - minb[d] + 1;
  }
  if ((int64_t)divb[0] * (int64_t)divb[1] * (int64_t)divb[2] > 2147483647ll) return -1;
  const int mul1 = divb[0], mul2 = divb[0] * divb[1];
  lso_vox_pair* pr = (lso_vox_pair*)malloc(sizeof(lso_vox_pair) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const int i0 = (int)(floorf(xyz1[4 * i] * inv[0]) - (float)minb[0]);
    const int i1 = (int)(floorf(xyz1[4 * i + 1] * inv[1]) - (float)minb[1]);
    const int i2 = (int)(floorf(xyz1[4 * i + 2] * inv[2]) - (float)minb[2]);
    pr[i].idx = i0 + i1 * mul1 + i2 * mul2;
    pr[i].pt = (int32_t)i;
  }
  qsort(pr, (size_t)n, sizeof(lso_vox_pair), lso_vox_cmp);
  int64_t m = 0;
  for (int64_t a = 0; a < n;) {
    int64_t b = a;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    while (b < n && pr[b].idx == pr[a].idx) {
      sx += xyz1[4 * (int64_t)pr[b].pt]; sy += xyz1[4 * (int64_t)pr[b].pt + 1]; sz += xyz1[4 * (int64_t)pr[b].pt + 2];
      ++b;
    }
    if (b - a >= (int64_t)min_points) {
      const float cnt = (float)(b - a);
      out_xyz1[4 * m] = sx / cnt; out_xyz1[4 * m + 1] = sy / cnt; out_xyz1[4 * m + 2] = sz / cnt; out_xyz1[4 * m + 3] = 1.0f;
      ++m;
    }
    a = b;
  }
  free(pr);
  return m;
}

/* ------------------------------------------------------------------ input filter chain (K0) */
/* laser_slam/src/laser_track.cpp:146  input_filters_.apply(scan.scan).  DataPointsFilters::apply runs the
 * filters in order and throws ConvergenceError("no points to filter") when one of them receives an empty cloud. */
int64_t lso_apply_point_filters(lso_point_filter* filters, int n_filters, const float* xyz1, int64_t n,
                                int64_t seed, float* out_xyz1) {
  if (seed >= 0) srand((unsigned)seed);
  if (n <= 0) return n_filters > 0 ? -1 : 0;  /* empty cloud into a non-empty chain: ConvergenceError upstream */
  float* cur = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  memcpy(cur, xyz1, sizeof(float) * 4 * (size_t)n);
  int64_t m = n;
  for (int k = 0; k < n_filters; ++k) {
    lso_point_filter* f = &filters[k];
    if (m == 0) { free(cur); return -1; }
    int64_t o = 0;
    uint32_t step = 1, phase = 0;
    if (f->type == 4) { /* FixStepSamplingDataPointsFilter::inPlaceFilter */
      double st = f->state > 0.0 ? f->state : (double)f->v[0];
      const int istep = (int)st < 1 ? 1 : (int)st;
      step = (uint32_t)istep;
      phase = (uint32_t)rand() % (uint32_t)istep;
      const double delta = (double)f->v[0] * (double)f->v[2] - (double)f->v[0];
      st *= (double)f->v[2];
      if (delta >= 0 && st > (double)f->v[1]) st = (double)f->v[1];
      if (delta < 0 && st < (double)f->v[1]) st = (double)f->v[1];
      f->state = st;
    }
    for (int64_t i = 0; i < m; ++i) {
      const float x = cur[4 * i], y = cur[4 * i + 1], z = cur[4 * i + 2];
      int keep = 1;
      if (f->type == 1 || f->type == 2) {
        /* MaxDist / MinDist (upstream, from knowledge): radial branch = norm against |limit| for both; single-axis
         * branch = SIGNED coordinate < maxDist for MaxDist, |coordinate| > minDist for MinDist */
        const float c = f->dim == 0 ? x : f->dim == 1 ? y : z;
        if (f->dim < 0) {
          const float val = sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
          keep = f->type == 1 ? (val < fabsf(f->v[0])) : (val > fabsf(f->v[0]));
        } else {
          keep = f->type == 1 ? (c < f->v[0]) : (fabsf(c) > f->v[0]);
        }
      } else if (f->type == 3) {
        const int in = x > f->v[0] && x < f->v[1] && y > f->v[2] && y < f->v[3] && z > f->v[4] && z < f->v[5];
        keep = f->flag ? !in : in;
      } else if (f->type == 4) {
        keep = (uint64_t)i >= phase && ((uint64_t)i - phase) % step == 0;
      } else if (f->type == 5) {
        const float r = (float)rand() / (float)RAND_MAX;
        keep = r < f->v[0];
      } else if (f->type == 6) { /* RemoveNaNDataPointsFilter: !(col == col).all() over the feature rows (x, y, z, pad) */
        const float w = cur[4 * i + 3];
        keep = x == x && y == y && z == z && w == w;
      }
      if (keep) { if (o != i) memcpy(cur + 4 * o, cur + 4 * i, 16); ++o; }
    }
    m = o;
  }
  memcpy(out_xyz1, cur, sizeof(float) * 4 * (size_t)m);
  free(cur);
  return m;
}
