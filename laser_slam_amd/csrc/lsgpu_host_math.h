// lsgpu_host_math.h -- the O(1) host arithmetic of one ICP iteration.
//
// libpointmatcher pieces restated here (selected by laser_slam/configurations/icp_default.yaml):
//   PointToPlaneErrorMinimizer solve + AngleAxis update ............ yaml:18-19
//   CounterTransformationChecker / DifferentialTransformationChecker  yaml:21-27
//   final composition T_refIn_refMean * T_iter * T_refMean_dataIn
// All float, column-major 4x4 (PointMatcher<float>::TransformationParameters,
// laser_slam/include/laser_slam/common.hpp:14).  Compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <array>

namespace lsgpu {
namespace hostmath {

inline float& at(float* m, int r, int c) { return m[c * 4 + r]; }
inline float at(const float* m, int r, int c) { return m[c * 4 + r]; }

inline void identity4(float* m) {
  for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f;
}

// out = a * b (out may alias an input)
inline void mul4(const float* a, const float* b, float* out) {
  float t[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float s = at(a, r, 0) * at(b, 0, c);
      s = s + at(a, r, 1) * at(b, 1, c);
      s = s + at(a, r, 2) * at(b, 2, c);
      s = s + at(a, r, 3) * at(b, 3, c);
      t[c * 4 + r] = s;
    }
  std::memcpy(out, t, sizeof(t));
}

// device accumulator layout: 21 upper-tri (row major, a<=c), 6 rhs, count, sum r^2
inline void unpack_normal_eq(const double* ne, double A[36], double b[6]) {
  int k = 0;
  for (int a = 0; a < 6; ++a)
    for (int c = a; c < 6; ++c, ++k) A[a * 6 + c] = A[c * 6 + a] = ne[k];
  for (int a = 0; a < 6; ++a) b[a] = ne[21 + a];
}

// x = A.llt().solve(b) in float (the minimiser works in PointMatcher<float>).
inline bool llt_solve6(const double A[36], const double b[6], float x[6]) {
  float L[6][6] = {};
  float y[6];
  for (int j = 0; j < 6; ++j) {
    float s = (float)A[j * 6 + j];
    for (int k = 0; k < j; ++k) s = s - L[j][k] * L[j][k];
    if (!(s > 0.f)) return false;
    L[j][j] = std::sqrt(s);
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
    if (std::isnan(x[i])) return false;
  return true;
}

// dT = [AngleAxis(|w|, w/|w|), t] with x = [w; t]; a zero rotation vector gives identity.
inline void delta_from_x(const float x[6], float* dT) {
  identity4(dT);
  const float ang = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (ang > 0.f && std::isfinite(ang)) {
    const float ax = x[0] / ang, ay = x[1] / ang, az = x[2] / ang;
    const float s = std::sin(ang), c = std::cos(ang);
    const float sx = s * ax, sy = s * ay, sz = s * az;
    const float kx = (1.f - c) * ax, ky = (1.f - c) * ay, kz = (1.f - c) * az;
    float t = kx * ay; at(dT, 0, 1) = t - sz; at(dT, 1, 0) = t + sz;
    t = kx * az;       at(dT, 0, 2) = t + sy; at(dT, 2, 0) = t - sy;
    t = ky * az;       at(dT, 1, 2) = t - sx; at(dT, 2, 1) = t + sx;
    at(dT, 0, 0) = kx * ax + c;
    at(dT, 1, 1) = ky * ay + c;
    at(dT, 2, 2) = kz * az + c;
  }
  at(dT, 0, 3) = x[3]; at(dT, 1, 3) = x[4]; at(dT, 2, 3) = x[5];
}

struct Quat { float w, x, y, z; };

inline Quat quat_from_rotation(const float* T) {
  Quat q;
  float* v = &q.x;  // x,y,z contiguous
  float t = at(T, 0, 0) + at(T, 1, 1) + at(T, 2, 2);
  if (t > 0.f) {
    t = std::sqrt(t + 1.0f);
    q.w = 0.5f * t;
    t = 0.5f / t;
    q.x = (at(T, 2, 1) - at(T, 1, 2)) * t;
    q.y = (at(T, 0, 2) - at(T, 2, 0)) * t;
    q.z = (at(T, 1, 0) - at(T, 0, 1)) * t;
  } else {
    int i = 0;
    if (at(T, 1, 1) > at(T, 0, 0)) i = 1;
    if (at(T, 2, 2) > at(T, i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(at(T, i, i) - at(T, j, j) - at(T, k, k) + 1.0f);
    v[i] = 0.5f * t;
    t = 0.5f / t;
    q.w = (at(T, k, j) - at(T, j, k)) * t;
    v[j] = (at(T, j, i) + at(T, i, j)) * t;
    v[k] = (at(T, k, i) + at(T, i, k)) * t;
  }
  return q;
}

// angle of a * conj(b): 2 atan2(|vec|, |w|)
inline float angular_distance(const Quat& a, const Quat& b) {
  const float bw = b.w, bx = -b.x, by = -b.y, bz = -b.z;
  const float w = a.w * bw - a.x * bx - a.y * by - a.z * bz;
  const float x = a.w * bx + a.x * bw + a.y * bz - a.z * by;
  const float y = a.w * by + a.y * bw + a.z * bx - a.x * bz;
  const float z = a.w * bz + a.z * bw + a.x * by - a.y * bx;
  return 2.0f * std::atan2(std::sqrt(x * x + y * y + z * z), std::fabs(w));
}

// Counter (first) then Differential, in the order icp_default.yaml:21-27 lists them.
class Checkers {
 public:
  Checkers(int max_iter, int smooth, float lim_rot, float lim_trans, const float* T0)
      : max_iter_(max_iter), smooth_(smooth), lim_rot_(lim_rot), lim_trans_(lim_trans) {
    push(T0);
  }
  // false => NaN (ConvergenceError).  *iterate is cleared when a checker says stop.
  bool check(const float* T, bool* iterate, bool* by_diff) {
    if (++counter_ >= max_iter_) { *iterate = false; return true; }  // MaxNumIterationsReached
    push(T);
    float rot = 0.f, trans = 0.f;
    const int n = (int)quats_.size();
    if (n > smooth_) {
      for (int i = n - 1; i >= n - smooth_; --i) {
        rot += std::fabs(angular_distance(quats_[i], quats_[i - 1]));
        const float dx = trans_[i][0] - trans_[i - 1][0];
        const float dy = trans_[i][1] - trans_[i - 1][1];
        const float dz = trans_[i][2] - trans_[i - 1][2];
        trans += std::fabs(std::sqrt(dx * dx + dy * dy + dz * dz));
      }
      rot /= (float)smooth_;
      trans /= (float)smooth_;
      if (rot < lim_rot_ && trans < lim_trans_) { *iterate = false; *by_diff = true; }
    }
    return !(std::isnan(rot) || std::isnan(trans));
  }

 private:
  void push(const float* T) {
    quats_.push_back(quat_from_rotation(T));
    trans_.push_back({at(T, 0, 3), at(T, 1, 3), at(T, 2, 3)});
  }
  int max_iter_, smooth_, counter_ = 0;
  float lim_rot_, lim_trans_;
  std::vector<Quat> quats_;
  std::vector<std::array<float, 3>> trans_;
};

}  // namespace hostmath
}  // namespace lsgpu
