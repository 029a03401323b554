// lsgpu_host_math.h -- the O(1) arithmetic of one ICP iteration, shared by host and device code
// (k_icp_update runs it on the GPU so that the loop needs no per-iteration host round trip; the host
// uses it for the final composition).
//
// libpointmatcher pieces restated here (selected by laser_slam/configurations/icp_default.yaml):
//   PointToPlaneErrorMinimizer solve + AngleAxis update ............ yaml:18-19
//   CounterTransformationChecker / DifferentialTransformationChecker  yaml:21-27
//   final composition T_refIn_refMean * T_iter * T_refMean_dataIn
// All float, column-major 4x4 (PointMatcher<float>::TransformationParameters,
// laser_slam/include/laser_slam/common.hpp:14).  Compiled with -ffp-contract=off; only IEEE
// + - * / sqrt are used, except sin/cos/atan2 which are evaluated in double and rounded to float
// (within 1 ulp of libm's float versions, identical in all but ~1e-9 of the cases).
#pragma once
#include <cmath>
#include <cstring>

#if defined(__HIPCC__)
#define LSGPU_HD __host__ __device__ inline
#else
#define LSGPU_HD inline
#endif

namespace lsgpu {
namespace hostmath {

LSGPU_HD float at(const float* m, int r, int c) { return m[c * 4 + r]; }
LSGPU_HD void set(float* m, int r, int c, float v) { m[c * 4 + r] = v; }

LSGPU_HD void identity4(float* m) {
  for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f;
}

// out = a * b (out may alias an input)
LSGPU_HD void mul4(const float* a, const float* b, float* out) {
  float t[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float s = at(a, r, 0) * at(b, 0, c);
      s = s + at(a, r, 1) * at(b, 1, c);
      s = s + at(a, r, 2) * at(b, 2, c);
      s = s + at(a, r, 3) * at(b, 3, c);
      t[c * 4 + r] = s;
    }
  for (int i = 0; i < 16; ++i) out[i] = t[i];
}

// device accumulator layout: 21 upper-tri (row major, a<=c), 6 rhs, count, sum r^2
LSGPU_HD void unpack_normal_eq(const double* ne, double A[36], double b[6]) {
  int k = 0;
  for (int a = 0; a < 6; ++a)
    for (int c = a; c < 6; ++c, ++k) A[a * 6 + c] = A[c * 6 + a] = ne[k];
  for (int a = 0; a < 6; ++a) b[a] = ne[21 + a];
}

// x = A.llt().solve(b) in float (the minimiser works in PointMatcher<float>).
LSGPU_HD bool llt_solve6(const double A[36], const double b[6], float x[6]) {
  float L[6][6];
  float y[6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) L[i][j] = 0.f;
  for (int j = 0; j < 6; ++j) {
    float s = (float)A[j * 6 + j];
    for (int k = 0; k < j; ++k) s = s - L[j][k] * L[j][k];
    if (!(s > 0.f)) return false;
    L[j][j] = sqrtf(s);
    for (int i = j + 1; i < 6; ++i) {
      float t = (float)A[i * 6 + j];
      for (int k = 0; k < j; ++k) t = t - L[i][k] * L[j][k];
      L[i][j] = t / L[j][j];
    }
  }
  for (int i = 0; i < 6; ++i) {
    float t = (float)b[i];
    for (int k = 0; k < i; ++k) t = t - L[i][k] * y[k];
    y[i] = t / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    float t = y[i];
    for (int k = i + 1; k < 6; ++k) t = t - L[k][i] * x[k];
    x[i] = t / L[i][i];
  }
  for (int i = 0; i < 6; ++i)
    if (x[i] != x[i]) return false;
  return true;
}

LSGPU_HD float sin_f32(float a) { return (float)sin((double)a); }
LSGPU_HD float cos_f32(float a) { return (float)cos((double)a); }
LSGPU_HD float atan2_f32(float y, float x) { return (float)atan2((double)y, (double)x); }

// dT = [AngleAxis(|w|, w/|w|), t] with x = [w; t]; a zero rotation vector gives identity.
LSGPU_HD void delta_from_x(const float x[6], float* dT) {
  identity4(dT);
  const float ang = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (ang > 0.f && ang <= 3.4e38f) {
    const float ax = x[0] / ang, ay = x[1] / ang, az = x[2] / ang;
#if defined(__HIP_DEVICE_COMPILE__)
    double sd, cd;   // one range reduction for both (the device library's sin / cos are this same routine)
    sincos((double)ang, &sd, &cd);
    const float s = (float)sd, c = (float)cd;
#else
    const float s = sin_f32(ang), c = cos_f32(ang);
#endif
    const float sx = s * ax, sy = s * ay, sz = s * az;
    const float kx = (1.f - c) * ax, ky = (1.f - c) * ay, kz = (1.f - c) * az;
    float t = kx * ay; set(dT, 0, 1, t - sz); set(dT, 1, 0, t + sz);
    t = kx * az;       set(dT, 0, 2, t + sy); set(dT, 2, 0, t - sy);
    t = ky * az;       set(dT, 1, 2, t - sx); set(dT, 2, 1, t + sx);
    set(dT, 0, 0, kx * ax + c);
    set(dT, 1, 1, ky * ay + c);
    set(dT, 2, 2, kz * az + c);
  }
  set(dT, 0, 3, x[3]); set(dT, 1, 3, x[4]); set(dT, 2, 3, x[5]);
}

// Eigen Quaternion(Matrix3): q = {w, x, y, z}
LSGPU_HD void quat_from_rotation(const float* T, float q[4]) {
  float t = at(T, 0, 0) + at(T, 1, 1) + at(T, 2, 2);
  if (t > 0.f) {
    t = sqrtf(t + 1.0f);
    q[0] = 0.5f * t;
    t = 0.5f / t;
    q[1] = (at(T, 2, 1) - at(T, 1, 2)) * t;
    q[2] = (at(T, 0, 2) - at(T, 2, 0)) * t;
    q[3] = (at(T, 1, 0) - at(T, 0, 1)) * t;
  } else {
    int i = 0;
    if (at(T, 1, 1) > at(T, 0, 0)) i = 1;
    if (at(T, 2, 2) > at(T, i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(at(T, i, i) - at(T, j, j) - at(T, k, k) + 1.0f);
    q[1 + i] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (at(T, k, j) - at(T, j, k)) * t;
    q[1 + j] = (at(T, j, i) + at(T, i, j)) * t;
    q[1 + k] = (at(T, k, i) + at(T, i, k)) * t;
  }
}

// angle of a * conj(b): 2 atan2(|vec|, |w|)   (Eigen >= 3.3 angularDistance)
LSGPU_HD float angular_distance(const float a[4], const float b[4]) {
  const float bw = b[0], bx = -b[1], by = -b[2], bz = -b[3];
  const float w = a[0] * bw - a[1] * bx - a[2] * by - a[3] * bz;
  const float x = a[0] * bx + a[1] * bw + a[2] * bz - a[3] * by;
  const float y = a[0] * by + a[2] * bw + a[3] * bx - a[1] * bz;
  const float z = a[0] * bz + a[3] * bw + a[1] * by - a[2] * bx;
  return 2.0f * atan2_f32(sqrtf(x * x + y * y + z * z), fabsf(w));
}

// Counter (first) then Differential, in the order icp_default.yaml:21-27 lists them.  The history
// lives in caller-provided arrays of (max_iter + 2) entries: hist[i] = {qw,qx,qy,qz, tx,ty,tz, |rotation to entry i-1|}.
struct CheckerState {
  int counter;
  int n_hist;
};

LSGPU_HD void checker_push(CheckerState* s, float* hist, const float* T) {
  float* e = hist + 8 * s->n_hist;
  quat_from_rotation(T, e);
  e[4] = at(T, 0, 3); e[5] = at(T, 1, 3); e[6] = at(T, 2, 3);
  // slot 7: the rotation between this entry and its predecessor -- the differential checker looks at every pair
  // `smooth` times over consecutive iterations; computed once, here (an atan2 evaluated in double each)
  e[7] = s->n_hist > 0 ? fabsf(angular_distance(e, e - 8)) : 0.f;
  s->n_hist++;
}

// false => NaN (ConvergenceError).  *iterate is cleared when a checker says stop.
// (out2: the two smoothed changes the differential checker compared with its limits, 0 while its history is short --
// the device's loop state carries them to the host, which sizes its next group of launches from them)
LSGPU_HD bool checker_check(CheckerState* s, float* hist, int max_iter, int smooth, float lim_rot,
                            float lim_trans, const float* T, bool* iterate, bool* by_diff, float* out2 = nullptr) {
  if (out2) { out2[0] = 0.f; out2[1] = 0.f; }
  if (++s->counter >= max_iter) { *iterate = false; return true; }  // MaxNumIterationsReached
  checker_push(s, hist, T);
  float rot = 0.f, trans = 0.f;
  const int n = s->n_hist;
  if (n > smooth) {
    for (int i = n - 1; i >= n - smooth; --i) {
      const float* a = hist + 8 * i;
      const float* b = hist + 8 * (i - 1);
      rot += a[7];   // = fabsf(angular_distance(a, b)), see checker_push
      const float dx = a[4] - b[4], dy = a[5] - b[5], dz = a[6] - b[6];
      trans += fabsf(sqrtf(dx * dx + dy * dy + dz * dz));
    }
    rot /= (float)smooth;
    trans /= (float)smooth;
    if (out2) { out2[0] = rot; out2[1] = trans; }
    if (rot < lim_rot && trans < lim_trans) { *iterate = false; *by_diff = true; }
  }
  return !(rot != rot || trans != trans);
}

}  // namespace hostmath
}  // namespace lsgpu
