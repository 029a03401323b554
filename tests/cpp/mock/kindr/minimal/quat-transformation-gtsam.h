// PARSE-CHECK STAND-IN (tests/cpp/mock/README.md): the expression helpers laser_track.cpp:435-456 / incremental_estimator.cpp:119-123 call
#pragma once
#include <gtsam/nonlinear/Expression.h>
#include <kindr/minimal/quat-transformation.h>
namespace kindr { namespace minimal {
typedef QuatTransformationTemplate<double> QuatTransformation;
gtsam::Expression<QuatTransformation> inverse(const gtsam::Expression<QuatTransformation>& T);
gtsam::Expression<QuatTransformation> compose(const gtsam::Expression<QuatTransformation>& T1, const gtsam::Expression<QuatTransformation>& T2);
}}  // namespace kindr::minimal
