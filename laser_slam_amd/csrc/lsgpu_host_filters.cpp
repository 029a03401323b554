// lsgpu_host_filters.cpp -- host-side modules of the chain (the reference runs them on the CPU
// inside PointMatcher::ICP::compute, laser_slam/src/laser_track.cpp:496):
//   RandomSamplingDataPointsFilter           laser_slam/configurations/icp_default.yaml:1-3
//   SamplingSurfaceNormalDataPointsFilter    laser_slam/configurations/icp_default.yaml:5-7
//   RigidTransformation check / correct      laser_slam/include/laser_slam/common.hpp:136-149
// GPU versions of the two filters are SURVEY.md §8f row N1/N3 ("next").
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/lsgpu_icp.h"

namespace {

struct SurfaceNormalBuilder {
  const float* xyz1;
  std::vector<int32_t> idx;
  int knn;
  float ratio;
  float* out_xyz1;
  float* out_nrm;
  int64_t n_out = 0;

  float coord(int32_t i, int d) const { return xyz1[4 * (int64_t)i + d]; }

  // symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations (double)
  static void eig3(double a[3][3], double w[3], double v[3][3]) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
      const double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
      const double dia = std::fabs(a[0][0]) + std::fabs(a[1][1]) + std::fabs(a[2][2]);
      if (off <= 1e-300 || off <= 1e-18 * dia) break;
      for (int p = 0; p < 2; ++p)
        for (int q = p + 1; q < 3; ++q) {
          if (a[p][q] == 0.0) continue;
          const double th = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
          const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
          const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
          for (int k = 0; k < 3; ++k) {
            const double x = a[k][p], y = a[k][q];
            a[k][p] = c * x - s * y; a[k][q] = s * x + c * y;
          }
          for (int k = 0; k < 3; ++k) {
            const double x = a[p][k], y = a[q][k];
            a[p][k] = c * x - s * y; a[q][k] = s * x + c * y;
          }
          for (int k = 0; k < 3; ++k) {
            const double x = v[k][p], y = v[k][q];
            v[k][p] = c * x - s * y; v[k][q] = s * x + c * y;
          }
        }
    }
    for (int i = 0; i < 3; ++i) w[i] = a[i][i];
  }

  // rank with full pivoting, threshold = max pivot * eps * 3 (FullPivLU default)
  static int rank3(const float c[3][3]) {
    float m[3][3];
    std::memcpy(m, c, sizeof(m));
    float piv[3] = {0, 0, 0}, maxpiv = 0.f;
    for (int k = 0; k < 3; ++k) {
      int pr = k, pc = k;
      float best = -1.f;
      for (int i = k; i < 3; ++i)
        for (int j = k; j < 3; ++j)
          if (std::fabs(m[i][j]) > best) { best = std::fabs(m[i][j]); pr = i; pc = j; }
      if (best <= 0.f) break;
      for (int j = 0; j < 3; ++j) std::swap(m[k][j], m[pr][j]);
      for (int i = 0; i < 3; ++i) std::swap(m[i][k], m[i][pc]);
      piv[k] = std::fabs(m[k][k]);
      maxpiv = std::max(maxpiv, piv[k]);
      for (int i = k + 1; i < 3; ++i) {
        const float f = m[i][k] / m[k][k];
        for (int j = k; j < 3; ++j) m[i][j] -= f * m[k][j];
      }
    }
    const float thr = maxpiv * FLT_EPSILON * 3.0f;
    return (piv[0] > thr) + (piv[1] > thr) + (piv[2] > thr);
  }

  void fuse(int64_t first, int64_t last) {
    const int64_t cnt = last - first;
    if (cnt <= 0) return;
    float mean[3] = {0, 0, 0};
    for (int64_t i = first; i < last; ++i)
      for (int d = 0; d < 3; ++d) mean[d] += coord(idx[i], d);
    for (int d = 0; d < 3; ++d) mean[d] /= (float)cnt;
    float C[3][3] = {};
    for (int64_t i = first; i < last; ++i) {
      float e[3];
      for (int d = 0; d < 3; ++d) e[d] = coord(idx[i], d) - mean[d];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) C[a][b] += e[a] * e[b];
    }
    if (rank3(C) + 1 < 3) return;  // degenerate box: dropped
    double a[3][3], w[3], v[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) a[i][j] = C[i][j];
    eig3(a, w, v);
    int k = 0;
    if (w[1] < w[k]) k = 1;
    if (w[2] < w[k]) k = 2;
    double n[3] = {v[0][k], v[1][k], v[2][k]};
    const double nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (double& c : n) c /= nl;
    for (int64_t i = first; i < last; ++i) {  // samplingMethod 0: random subset keeps the box normal
      const float r = (float)std::rand() / (float)RAND_MAX;
      if (r < ratio) {
        const int64_t o = n_out++;
        std::memcpy(out_xyz1 + 4 * o, xyz1 + 4 * (int64_t)idx[i], 16);
        for (int d = 0; d < 3; ++d) out_nrm[3 * o + d] = (float)n[d];
      }
    }
  }

  void build(int64_t first, int64_t last, const float* minb, const float* maxb) {
    const int64_t count = last - first;
    if (count <= knn) { fuse(first, last); return; }
    int cut = 0;
    for (int d = 1; d < 3; ++d)
      if (maxb[d] - minb[d] > maxb[cut] - minb[cut]) cut = d;
    const int64_t right = count / 2, left = count - right;
    std::nth_element(idx.begin() + first, idx.begin() + first + left, idx.begin() + last,
                     [&](int32_t a, int32_t b) { return coord(a, cut) < coord(b, cut); });
    const float cutval = coord(idx[first + left], cut);
    float lmax[3] = {maxb[0], maxb[1], maxb[2]}, rmin[3] = {minb[0], minb[1], minb[2]};
    lmax[cut] = cutval; rmin[cut] = cutval;
    build(first, first + left, minb, lmax);
    build(first + left, last, rmin, maxb);
  }
};

float det3(const float* T) {
  auto m = [&](int r, int c) { return T[c * 4 + r]; };
  return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) -
         m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
         m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
}

}  // namespace

extern "C" {

int64_t lsgpu_filter_random_sampling(int64_t n, float prob, int64_t seed, int64_t* keep_idx) {
  if (seed >= 0) std::srand((unsigned)seed);
  if (n <= 0 || !keep_idx) return 0;
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float r = (float)std::rand() / (float)RAND_MAX;
    if (r < prob) keep_idx[m++] = i;
  }
  return m;
}

int64_t lsgpu_filter_sampling_surface_normal(const float* xyz1, int64_t n, int knn, float ratio,
                                             int64_t seed, float* out_xyz1, float* out_normals) {
  if (n <= 0 || !xyz1 || !out_xyz1 || !out_normals || knn < 3) return 0;
  if (seed >= 0) std::srand((unsigned)seed);
  SurfaceNormalBuilder b;
  b.xyz1 = xyz1; b.knn = knn; b.ratio = ratio; b.out_xyz1 = out_xyz1; b.out_nrm = out_normals;
  b.idx.resize((size_t)n);
  std::iota(b.idx.begin(), b.idx.end(), 0);
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      mn[d] = std::min(mn[d], xyz1[4 * i + d]);
      mx[d] = std::max(mx[d], xyz1[4 * i + d]);
    }
  b.build(0, n, mn, mx);
  return b.n_out;
}

int lsgpu_check_rigid(const float T[16]) { return std::fabs(1.0f - det3(T)) <= 0.001f; }

void lsgpu_correct_rigid(const float T[16], float out[16]) {
  float c[3][3];
  for (int k = 0; k < 3; ++k) {
    const float x = T[k * 4], y = T[k * 4 + 1], z = T[k * 4 + 2];
    const float n = std::sqrt(x * x + y * y + z * z);
    c[k][0] = x / n; c[k][1] = y / n; c[k][2] = z / n;
  }
  const float c0[3] = {c[1][1] * c[2][2] - c[1][2] * c[2][1], c[1][2] * c[2][0] - c[1][0] * c[2][2],
                       c[1][0] * c[2][1] - c[1][1] * c[2][0]};
  const float c1[3] = {c[2][1] * c0[2] - c[2][2] * c0[1], c[2][2] * c0[0] - c[2][0] * c0[2],
                       c[2][0] * c0[1] - c[2][1] * c0[0]};
  std::memcpy(out, T, 16 * sizeof(float));
  for (int r = 0; r < 3; ++r) { out[r] = c0[r]; out[4 + r] = c1[r]; out[8 + r] = c[2][r]; }
}

}  // extern "C"
