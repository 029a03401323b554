// The library's draw stream (csrc/lsgpu_rand.h) against glibc's srand/rand: the speculative use of the device filters
// -- begin(kmax) produces more draws than are consumed, commit(k < kmax) consumes k of them, the stream continues from
// there -- including the parallel path of large requests (segment start states by jump-ahead).
#include "lsgpu_rand.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main() {
  int bad = 0;
  const size_t cases[][2] = {{1000, 1000}, {1000, 0}, {1000, 517}, {300000, 299999}, {300000, 65536}, {300000, 65535},
                             {1200000, 777777}, {1200000, 1200000}, {262144, 131072}};
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
  std::printf(bad ? "DRAWS_FAIL\n" : "DRAWS_OK\n");
  return bad ? 1 : 0;
}
