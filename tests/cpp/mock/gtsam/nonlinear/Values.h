// PARSE-CHECK STAND-IN (tests/cpp/mock/README.md): gtsam::Values / Key / KeySet / noise models as the overlay uses them.
#pragma once
#include <Eigen/Dense>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <set>
#include <vector>
namespace gtsam {
typedef std::uint64_t Key;
typedef Eigen::Matrix<double, 6, 1> Vector6;
typedef Eigen::MatrixXd Matrix;
class KeySet : public std::set<Key> {};
template <class T> using FastVector = std::vector<T>;
typedef FastVector<std::size_t> FactorIndices;
class Value {
 public:
  template <class T> const T& cast() const;
};
class Values {
 public:
  struct ConstKeyValuePair { const Key key; const Value& value; };
  struct const_iterator {
    ConstKeyValuePair operator*() const;
    const_iterator& operator++();
    bool operator!=(const const_iterator&) const;
  };
  Values();
  Values(const Values&);
  Values& operator=(const Values&);
  template <class T> void insert(Key j, const T& val);
  void clear();
  std::size_t size() const;
  const_iterator begin() const;
  const_iterator end() const;
};
namespace noiseModel {
class Base { public: typedef std::shared_ptr<Base> shared_ptr; virtual ~Base(); };
class Diagonal : public Base { public: static std::shared_ptr<Diagonal> Sigmas(const Vector6& sigmas, bool smart = true); };
namespace mEstimator {
class Base { public: typedef std::shared_ptr<Base> shared_ptr; virtual ~Base(); };
class Cauchy : public Base { public: static std::shared_ptr<Cauchy> Create(double k); };
}  // namespace mEstimator
class Robust : public Base {
 public:
  static std::shared_ptr<Robust> Create(const std::shared_ptr<mEstimator::Base>& robust, const std::shared_ptr<Base>& noise);
};
}  // namespace noiseModel
typedef noiseModel::Base::shared_ptr SharedNoiseModel;
}  // namespace gtsam
