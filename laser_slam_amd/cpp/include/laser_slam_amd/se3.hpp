// se3.hpp -- minimal rigid transform (unit quaternion + position, double) standing in for
// kindr::minimal::QuatTransformationTemplate<double>, the `SE3` of the reference
// (laser_slam/include/laser_slam/common.hpp:17).  Only what the ICP path touches: compose, inverse,
// 4x4 matrix in/out, renormalising construction from a rotation matrix
// (convertTransformationMatrixToSE3, common.hpp:263-269).
#pragma once
#include <array>
#include <cmath>

namespace laser_slam_amd {

class SE3 {
 public:
  SE3() : q_{1, 0, 0, 0}, p_{0, 0, 0} {}
  SE3(const std::array<double, 4>& q_wxyz, const std::array<double, 3>& p) : q_(q_wxyz), p_(p) { normalize(); }

  // rotation matrix (row major 3x3, nearly orthonormal) + position; renormalises the quaternion
  static SE3 fromRotationAndPosition(const double R[9], const double p[3]) {
    SE3 t;
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
      double s = std::sqrt(tr + 1.0) * 2;
      t.q_ = {0.25 * s, (R[7] - R[5]) / s, (R[2] - R[6]) / s, (R[3] - R[1]) / s};
    } else if (R[0] > R[4] && R[0] > R[8]) {
      double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
      t.q_ = {(R[7] - R[5]) / s, 0.25 * s, (R[1] + R[3]) / s, (R[2] + R[6]) / s};
    } else if (R[4] > R[8]) {
      double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
      t.q_ = {(R[2] - R[6]) / s, (R[1] + R[3]) / s, 0.25 * s, (R[5] + R[7]) / s};
    } else {
      double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
      t.q_ = {(R[3] - R[1]) / s, (R[2] + R[6]) / s, (R[5] + R[7]) / s, 0.25 * s};
    }
    t.p_ = {p[0], p[1], p[2]};
    t.normalize();
    return t;
  }

  // 4x4 float column-major (PointMatcher TransformationParameters) -> SE3   (common.hpp:263-269)
  static SE3 fromTransformationMatrix(const float T[16]) {
    double R[9], p[3];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) R[r * 3 + c] = (double)T[c * 4 + r];
      p[r] = (double)T[12 + r];
    }
    return fromRotationAndPosition(R, p);
  }

  void rotationMatrix(double R[9]) const {
    const double w = q_[0], x = q_[1], y = q_[2], z = q_[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
  }

  // getTransformationMatrix().cast<float>(): 4x4 float column major
  std::array<float, 16> transformationMatrixF() const {
    double R[9];
    rotationMatrix(R);
    std::array<float, 16> T{};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T[c * 4 + r] = (float)R[r * 3 + c];
      T[12 + r] = (float)p_[r];
    }
    T[15] = 1.f;
    return T;
  }

  SE3 inverse() const {
    SE3 t;
    t.q_ = {q_[0], -q_[1], -q_[2], -q_[3]};
    double R[9];
    t.rotationMatrix(R);
    for (int r = 0; r < 3; ++r) t.p_[r] = -(R[r * 3] * p_[0] + R[r * 3 + 1] * p_[1] + R[r * 3 + 2] * p_[2]);
    return t;
  }

  SE3 operator*(const SE3& o) const {
    SE3 t;
    const auto& a = q_; const auto& b = o.q_;
    t.q_ = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
    double R[9];
    rotationMatrix(R);
    for (int r = 0; r < 3; ++r)
      t.p_[r] = R[r * 3] * o.p_[0] + R[r * 3 + 1] * o.p_[1] + R[r * 3 + 2] * o.p_[2] + p_[r];
    t.normalize();
    return t;
  }

  // ---- minimal manifold operations for the pose graph (pose_graph.hpp): minkindr's decoupled
  // [translation; rotation vector] chart
  // this (+) d = this * SE3(exp(d[3..5]), d[0..2])
  SE3 retract(const double d[6]) const {
    const double th = std::sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    SE3 e;
    if (th < 1e-12) {
      e.q_ = {1.0, 0.5 * d[3], 0.5 * d[4], 0.5 * d[5]};
    } else {
      const double s = std::sin(0.5 * th) / th;
      e.q_ = {std::cos(0.5 * th), s * d[3], s * d[4], s * d[5]};
    }
    e.p_ = {d[0], d[1], d[2]};
    e.normalize();
    return (*this) * e;
  }
  // d with this (+) d == other: [position; rotation vector] of this^-1 * other
  void localCoordinates(const SE3& other, double d[6]) const {
    const SE3 e = inverse() * other;
    std::array<double, 4> q = e.q_;
    if (q[0] < 0) for (auto& v : q) v = -v;
    const double vn = std::sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double k = 2.0;  // angle / |v| for small angles
    if (vn > 1e-12) k = 2.0 * std::atan2(vn, q[0]) / vn;
    d[0] = e.p_[0]; d[1] = e.p_[1]; d[2] = e.p_[2];
    d[3] = k * q[1]; d[4] = k * q[2]; d[5] = k * q[3];
  }

  const std::array<double, 4>& quaternion() const { return q_; }
  const std::array<double, 3>& position() const { return p_; }

  // interpolation between two nodes (DiscreteSE3Curve::evaluate between keys): slerp + lerp
  static SE3 interpolate(const SE3& a, const SE3& b, double alpha) {
    std::array<double, 4> qb = b.q_;
    double dot = a.q_[0] * qb[0] + a.q_[1] * qb[1] + a.q_[2] * qb[2] + a.q_[3] * qb[3];
    if (dot < 0) { dot = -dot; for (auto& v : qb) v = -v; }
    double wa = 1 - alpha, wb = alpha;
    if (dot < 0.9995) {
      const double th = std::acos(dot), s = std::sin(th);
      wa = std::sin((1 - alpha) * th) / s; wb = std::sin(alpha * th) / s;
    }
    SE3 t;
    for (int i = 0; i < 4; ++i) t.q_[i] = wa * a.q_[i] + wb * qb[i];
    for (int i = 0; i < 3; ++i) t.p_[i] = (1 - alpha) * a.p_[i] + alpha * b.p_[i];
    t.normalize();
    return t;
  }

 private:
  void normalize() {
    const double n = std::sqrt(q_[0] * q_[0] + q_[1] * q_[1] + q_[2] * q_[2] + q_[3] * q_[3]);
    if (n > 0) for (auto& v : q_) v /= n;
  }
  std::array<double, 4> q_;  // w, x, y, z
  std::array<double, 3> p_;
};

}  // namespace laser_slam_amd
