#include "lsgpu_rand.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
int main() {
  // sequential reference: glibc itself
  const size_t K = 3100000, C = 2900001;
  std::vector<float> out(K), ref(K + 5000);
  srand(5);
  for (size_t i = 0; i < C; ++i) ref[i] = (float)rand() / (float)RAND_MAX;
  std::vector<float> cont(5000);
  for (auto& v : cont) v = (float)rand() / (float)RAND_MAX;
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    lsgpu::DrawStream::global().begin(5, K, out.data());
    lsgpu::DrawStream::global().commit(C);
    auto t1 = std::chrono::steady_clock::now();
    std::vector<float> c2(5000);
    lsgpu::DrawStream::global().take(-1, 5000, c2.data());
    size_t bad = 0;
    for (size_t i = 0; i < C; ++i) bad += out[i] != ref[i];
    size_t bad2 = 0;
    for (size_t i = 0; i < 5000; ++i) bad2 += c2[i] != cont[i];
    printf("3.1M draws: %.2f ms, mismatches %zu, continuation mismatches %zu\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), bad, bad2);
  }
}
