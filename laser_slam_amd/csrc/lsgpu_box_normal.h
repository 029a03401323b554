// lsgpu_box_normal.h -- the per-box arithmetic of SamplingSurfaceNormalDataPointsFilter
// (laser_slam/configurations/icp_default.yaml:5-7, run inside icp_.compute,
// laser_slam/src/laser_track.cpp:496), shared by the host filter (lsgpu_host_filters.cpp) and the
// device filter (lsgpu_ssn.hip.h): mean and covariance of the box in float, rank test (FullPivHouseholderQR,
// default threshold), eigenvector of the smallest eigenvalue by cyclic Jacobi rotations in double.
// Compiled with -ffp-contract=off; only IEEE + - * / sqrt fabs are used, so host and device agree
// bit for bit when they visit the box's points in the same order.
#pragma once
#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#define LSGPU_BN_HD __host__ __device__ inline
#else
#define LSGPU_BN_HD inline
#endif

namespace lsgpu {
namespace boxnormal {

// symmetric 3x3 eigen-decomposition (double); v columns = eigenvectors, w = eigenvalues
LSGPU_BN_HD void eig3(double a[3][3], double w[3], double v[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double dia = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-18 * dia) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double th = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
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

// Numerical rank as Eigen's FullPivHouseholderQR reports it -- upstream's fuseRange tests
// C.fullPivHouseholderQr().rank() + 1 >= dim (libpointmatcher SamplingSurfaceNormal.cpp, from knowledge; rounds 1-2 used
// FullPivLU pivots here, same threshold, different pivot values -- they differ only on borderline boxes).  Per step: the
// largest |entry| of the remaining corner is brought to (k, k) by a row and a column swap, the corner is declared
// negligible if it is <= eps * 3 times the very first one, a Householder reflection zeroes the column below the
// diagonal and |beta| (the column's norm) is the pivot; rank = pivots > max pivot * eps * 3.
LSGPU_BN_HD int rank3(const float c[3][3]) {
  float m[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m[i][j] = c[i][j];
  const float prec = FLT_EPSILON * 3.0f;
  float piv[3] = {0.f, 0.f, 0.f}, maxpiv = 0.f, biggest = 0.f;
  int nonzero = 3;
  for (int k = 0; k < 3; ++k) {
    int pr = k, pc = k;
    float corner = -1.f;
    for (int i = k; i < 3; ++i)
      for (int j = k; j < 3; ++j)
        if (fabsf(m[i][j]) > corner) { corner = fabsf(m[i][j]); pr = i; pc = j; }
    if (k == 0) biggest = corner;
    if (corner <= biggest * prec) { nonzero = k; break; }   // isMuchSmallerThan(corner, biggest, precision)
    for (int j = 0; j < 3; ++j) { const float t = m[k][j]; m[k][j] = m[pr][j]; m[pr][j] = t; }
    for (int i = 0; i < 3; ++i) { const float t = m[i][k]; m[i][k] = m[i][pc]; m[i][pc] = t; }
    // makeHouseholderInPlace on rows k..2 of column k
    float tail2 = 0.f;
    for (int i = k + 1; i < 3; ++i) tail2 += m[i][k] * m[i][k];
    const float c0 = m[k][k];
    float beta = c0, tau = 0.f, v[3] = {0.f, 0.f, 0.f};
    if (tail2 > FLT_MIN) {
      beta = sqrtf(c0 * c0 + tail2);
      if (c0 >= 0.f) beta = -beta;
      for (int i = k + 1; i < 3; ++i) v[i] = m[i][k] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    piv[k] = fabsf(beta);
    maxpiv = piv[k] > maxpiv ? piv[k] : maxpiv;
    for (int j = k + 1; j < 3; ++j) {     // apply (I - tau v v^T), v = (1, essential part), to the rest of the corner
      float sdot = m[k][j];
      for (int i = k + 1; i < 3; ++i) sdot += v[i] * m[i][j];
      m[k][j] -= tau * sdot;
      for (int i = k + 1; i < 3; ++i) m[i][j] -= tau * sdot * v[i];
    }
  }
  const float thr = maxpiv * prec;
  int r = 0;
  for (int k = 0; k < nonzero; ++k) r += (piv[k] > thr);
  return r;
}

// Normal of one box.  point(i, d) -> coordinate d of the box's i-th point, in the box's order.
// false: the box is too degenerate for a normal (it is dropped).
template <class PointFn>
LSGPU_BN_HD bool box_normal(int cnt, PointFn point, float out[3]) {
  if (cnt <= 0) return false;
  float mean[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < cnt; ++i)
    for (int d = 0; d < 3; ++d) mean[d] += point(i, d);
  for (int d = 0; d < 3; ++d) mean[d] /= (float)cnt;
  float C[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  for (int i = 0; i < cnt; ++i) {
    float e[3];
    for (int d = 0; d < 3; ++d) e[d] = point(i, d) - mean[d];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) C[a][b] += e[a] * e[b];
  }
  if (rank3(C) + 1 < 3) return false;
  double a[3][3], w[3], v[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = C[i][j];
  eig3(a, w, v);
  int k = 0;
  if (w[1] < w[k]) k = 1;
  if (w[2] < w[k]) k = 2;
  double n[3] = {v[0][k], v[1][k], v[2][k]};
  const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  for (int d = 0; d < 3; ++d) out[d] = (float)(n[d] / nl);
  return true;
}

}  // namespace boxnormal
}  // namespace lsgpu
