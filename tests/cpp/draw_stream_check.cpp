// The library's draw stream (csrc/lsgpu_rand.h) against glibc's srand/rand: the speculative use of the device filters
// -- begin(kmax) produces more draws than are consumed, commit(k < kmax) consumes k of them, the stream continues from
// there -- including the parallel path of large requests (segment start states by jump-ahead).
#include "lsgpu_rand.h"

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

int main() {
  int bad = 0;
  const size_t cases[][2] = {{1000, 1000}, {1000, 0}, {1000, 517}, {300000, 299999}, {300000, 65536}, {300000, 65535},
                             {1200000, 777777}, {1200000, 1200000}, {262144, 131072},
                             {1310720, 1310720}, {1310720, 196608}, {1200000, 196607}, {1200000, 2048}, {1200000, 1198081}, {1200000, 0}};
  unsigned seed = 42;
  for (const auto& c : cases) {
    const size_t kmax = c[0], k = c[1];
    std::vector<float> got(kmax), tail(300);
    lsgpu::DrawStream::global().begin((int64_t)seed, kmax, got.data());
    lsgpu::DrawStream::global().commit(k);
    lsgpu::DrawStream::global().take(-1, tail.size(), tail.data());
    srand(seed);
    size_t mism = 0;
    for (size_t i = 0; i < k; ++i) mism += got[i] != (float)rand() / (float)RAND_MAX;
    for (size_t i = 0; i < tail.size(); ++i) mism += tail[i] != (float)rand() / (float)RAND_MAX;
    if (mism) { std::printf("kmax %zu k %zu: %zu mismatches\n", kmax, k, mism); ++bad; }
    ++seed;
  }
  // the two-step form of lsgpu_icp_compute's DrawAhead: the owner locks, a helper thread generates, the owner commits --
  // one request serving two consecutive filters (k1 draws of the first kmax1, then k2 more right behind them)
  const size_t two[][4] = {{5000, 3210, 4000, 4000}, {1100000, 900001, 1000000, 1000000}, {70000, 0, 70000, 12345}};
  for (const auto& c : two) {
    const size_t kmax1 = c[0], k1 = c[1], kmax2 = c[2], k2 = c[3];
    std::vector<float> got(kmax1 + kmax2), tail(100);
    lsgpu::DrawStream::global().lock((int64_t)seed);
    float* dst = got.data();
    const size_t kmax = kmax1 + kmax2;
    std::thread worker([dst, kmax] { lsgpu::DrawStream::global().generate(kmax, dst); });
    worker.join();
    lsgpu::DrawStream::global().commit(k1 + k2);   // the second filter's draws start at k1
    lsgpu::DrawStream::global().take(-1, tail.size(), tail.data());
    srand(seed);
    size_t mism = 0;
    for (size_t i = 0; i < k1 + k2; ++i) mism += got[i] != (float)rand() / (float)RAND_MAX;
    for (size_t i = 0; i < tail.size(); ++i) mism += tail[i] != (float)rand() / (float)RAND_MAX;
    if (mism) { std::printf("two-step kmax %zu k %zu: %zu mismatches\n", kmax, k1 + k2, mism); ++bad; }
    ++seed;
  }
  // round 6: the caller is told as soon as the first k_first draws exist (the reference filter's share leaves for the
  // device while the reading filter's is still being produced) -- they must be final at that moment, whichever threads
  // of the pool filled them, and the whole request must still be the rand() sequence afterwards
  const size_t early[][2] = {{2092367, 1046335}, {4185022, 3139020}, {1048576, 65536}, {600000, 599999}, {100000, 50000}, {2000000, 2000000}};
  for (const auto& c : early) {
    const size_t kmax = c[0], kfirst = c[1];
    std::vector<float> got(kmax), first_copy;
    lsgpu::DrawStream::global().lock((int64_t)seed);
    float* dst = got.data();
    int calls = 0;
    std::thread worker([&] {
      lsgpu::DrawStream::global().generate(kmax, dst, kfirst, [&] { ++calls; first_copy.assign(dst, dst + kfirst); });
    });
    worker.join();
    lsgpu::DrawStream::global().commit(kmax);
    srand(seed);
    size_t mism = calls == 1 ? 0 : 1;
    for (size_t i = 0; i < kmax; ++i) {
      const float want = (float)rand() / (float)RAND_MAX;
      mism += got[i] != want;
      if (i < kfirst && i < first_copy.size()) mism += first_copy[i] != want;
    }
    mism += first_copy.size() != kfirst;
    if (mism) { std::printf("early kmax %zu kfirst %zu: %zu mismatches (%d calls)\n", kmax, kfirst, mism, calls); ++bad; }
    ++seed;
  }
  std::printf(bad ? "DRAWS_FAIL\n" : "DRAWS_OK\n");
  return bad ? 1 : 0;
}
