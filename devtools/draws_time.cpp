#include "../laser_slam_amd/csrc/lsgpu_rand.h"
#include <chrono>
#include <cstdio>
int main() {
  using namespace lsgpu;
  std::vector<float> out(4u << 20);
  for (size_t k : {200000ul, 1046335ul, 2092367ul, 3139020ul}) {
    for (int rep = 0; rep < 4; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      DrawStream::global().lock(0);
      DrawStream::global().generate(k, out.data());
      auto t1 = std::chrono::steady_clock::now();
      DrawStream::global().commit(k / 2);
      auto t2 = std::chrono::steady_clock::now();
      if (rep == 3) printf("k %zu: generate %.3f ms, commit %.3f ms\n", k, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
  }
}
