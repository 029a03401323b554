// PARSE-CHECK STAND-IN (tests/cpp/mock/README.md): kindr::minimal::QuatTransformationTemplate as common.hpp:17-18 names it
#pragma once
#include <Eigen/Dense>
namespace kindr { namespace minimal {
template <class S> class RotationQuaternionTemplate {
 public:
  RotationQuaternionTemplate();
  RotationQuaternionTemplate(S w, S x, S y, S z);
  S w() const; S x() const; S y() const; S z() const;
};
template <class S> class QuatTransformationTemplate {
 public:
  typedef RotationQuaternionTemplate<S> Rotation;
  typedef Eigen::Matrix<S, 3, 1> Position;
  typedef Eigen::Matrix<S, 4, 4> TransformationMatrix;
  QuatTransformationTemplate();
  QuatTransformationTemplate(const Rotation& q_A_B, const Position& A_t_A_B);
  const Rotation& getRotation() const;
  const Position& getPosition() const;
  TransformationMatrix getTransformationMatrix() const;
  QuatTransformationTemplate inverse() const;
  QuatTransformationTemplate operator*(const QuatTransformationTemplate& rhs) const;
};
}}  // namespace kindr::minimal
// the mock Position needs (x, y, z) construction
namespace Eigen {
template <> class Matrix<double, 3, 1> {
 public:
  Matrix();
  Matrix(double x, double y, double z);
  double& operator[](std::ptrdiff_t i);
  const double& operator[](std::ptrdiff_t i) const;
};
}  // namespace Eigen
