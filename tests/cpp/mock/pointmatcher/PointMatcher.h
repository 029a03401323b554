// PARSE-CHECK STAND-IN (tests/cpp/mock/README.md): PointMatcher<T>::DataPoints / TransformationParameters as common.hpp:14-15 uses them
#pragma once
#include <Eigen/Dense>
#include <stdexcept>
#include <string>
#include <vector>
template <class T> struct PointMatcher {
  typedef Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> Matrix;
  typedef Matrix TransformationParameters;
  struct ConvergenceError : std::runtime_error { explicit ConvergenceError(const std::string& reason); };
  struct DataPoints {
    struct Label { Label(const std::string& text = "", const std::size_t span = 0); std::string text; std::size_t span; };
    struct Labels : std::vector<Label> {};
    DataPoints();
    DataPoints(const Labels& featureLabels, const Labels& descriptorLabels, const std::size_t pointCount);
    Matrix features;
    Labels featureLabels;
    Matrix descriptors;
    Labels descriptorLabels;
  };
};
