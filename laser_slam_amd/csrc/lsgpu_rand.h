// lsgpu_rand.h -- the random draws of the sampling filters.
//
// RandomSamplingDataPointsFilter and SamplingSurfaceNormalDataPointsFilter
// (laser_slam/configurations/icp_default.yaml:1-7) draw `(float)std::rand() / (float)RAND_MAX` per
// point.  The library keeps its OWN generator with the sequence std::srand(seed) + std::rand() gives
// on glibc (the additive-feedback generator "TYPE_3": 31 words, r[i] = r[i-3] + r[i-31], output
// r >> 1, seeded by the Lehmer recurrence 16807 and 310 discarded outputs), so that
//   * a seeded filter call selects exactly the points the sequential CPU chain selects,
//   * no global libc state is touched (handles on different threads do not race on rand()),
//   * the draws of a whole cloud cost ~2 ns each and can be produced while the device works.
// seed >= 0 reseeds the stream; seed < 0 continues it (like calling rand() again).  An unseeded
// stream starts as srand(1), the C default.  tests/test_abi.py compares the sequence with libc's.
#pragma once
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

namespace lsgpu {

class DrawStream {
 public:
  DrawStream() { reseed_locked(1u); }

  // the next k draws, as floats in [0, 1]; seed >= 0: reseed first
  void take(int64_t seed, size_t k, float* out) {
    std::lock_guard<std::mutex> g(mu_);
    if (seed >= 0) reseed_locked((unsigned)seed);
    for (size_t i = 0; i < k; ++i) out[i] = (float)next_locked() / 2147483648.0f;  // (float)RAND_MAX == 2^31
  }

  static DrawStream& global() {
    static DrawStream s;
    return s;
  }

 private:
  void reseed_locked(unsigned seed) {
    int32_t word = seed ? (int32_t)seed : 1;
    r_[0] = word;
    for (int i = 1; i < 31; ++i) {
      const int32_t hi = word / 127773, lo = word % 127773;
      word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      r_[i] = word;
    }
    f_ = 3; b_ = 0;
    for (int i = 0; i < 310; ++i) (void)next_locked();
  }
  uint32_t next_locked() {
    const uint32_t v = (uint32_t)r_[f_] + (uint32_t)r_[b_];
    r_[f_] = (int32_t)v;
    f_ = f_ + 1 == 31 ? 0 : f_ + 1;
    b_ = b_ + 1 == 31 ? 0 : b_ + 1;
    return v >> 1;
  }
  std::mutex mu_;
  int32_t r_[31];
  int f_ = 3, b_ = 0;
};

}  // namespace lsgpu
