// lsgpu_rand.h -- the random draws of the sampling filters.
//
// RandomSamplingDataPointsFilter and SamplingSurfaceNormalDataPointsFilter
// (laser_slam/configurations/icp_default.yaml:1-7) draw `(float)std::rand() / (float)RAND_MAX` per
// point.  The library keeps its OWN generator with the sequence std::srand(seed) + std::rand() gives
// on glibc (the additive-feedback generator "TYPE_3": 31 words, r[i] = r[i-3] + r[i-31], output
// r >> 1, seeded by the Lehmer recurrence 16807 and 310 discarded outputs), so that
//   * a seeded filter call selects exactly the points the sequential CPU chain selects,
//   * no global libc state is touched (handles on different threads do not race on rand()),
//   * the draws of a whole cloud can be produced while the device works: ~0.8-1.3 ns each on one core, and for requests of
//     8 segments (half a million draws) or more on up to 8 threads -- the recurrence is linear over Z/2^32, so the state
//     65536 draws ahead is one 31 x 31 matrix product away (jump-ahead), and the segments are filled independently
//     (3.1 M draws of a three-scan sub-map: 4.2 ms -> 0.7 ms; it used to be the longest item of a LaserTrack scan).
//     Round 6: the threads are a standing pool (spawning eight cost 0.1 ms of a 0.5 ms request), they take the segments
//     IN ORDER, and the caller can be told as soon as the first k_first draws are there -- the reference filter's share
//     of a compute's draws goes to the device while the reading filter's is still being produced (the device used to
//     idle 0.3 ms in front of k_ssn_select, the whole request's production and upload ahead of it).
// seed >= 0 reseeds the stream; seed < 0 continues it (like calling rand() again).  An unseeded
// stream starts as srand(1), the C default.  tests/test_abi.py compares the sequence with libc's.
#pragma once
#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lsgpu {

class DrawStream {
 public:
  DrawStream() { reseed_locked(1u); }

  // the next k draws, as floats in [0, 1]; seed >= 0: reseed first
  void take(int64_t seed, size_t k, float* out) {
    std::lock_guard<std::mutex> g(mu_);
    if (seed >= 0) reseed_locked((unsigned)seed);
    generate_locked(k, out);
    commit_locked(k);
  }

  // one raw draw: the integer std::rand() would return (FixStepSampling's `rand() % step`)
  uint32_t take_raw(int64_t seed) {
    std::lock_guard<std::mutex> g(mu_);
    if (seed >= 0) reseed_locked((unsigned)seed);
    generate_locked(1, nullptr);
    const uint32_t r = raw_[31] >> 1;
    commit_locked(1);
    return r;
  }

  // Speculative use: begin() locks the stream and produces up to kmax draws WITHOUT consuming them,
  // commit(k) consumes the first k <= kmax of them and unlocks.  (The device filters learn how many
  // draws the sequential filter would have made only after their kernels ran; the draws themselves are
  // produced while those kernels run.)
  void begin(int64_t seed, size_t kmax, float* out) {
    mu_.lock();
    if (seed >= 0) reseed_locked((unsigned)seed);
    generate_locked(kmax, out);
  }
  void commit(size_t k) {
    commit_locked(k);
    mu_.unlock();
  }
  // The same in two steps, for draws produced on a helper thread: the OWNER calls lock() and later commit(); generate()
  // may run on any one thread in between (the owner joins it before commit()).
  void lock(int64_t seed) {
    mu_.lock();
    if (seed >= 0) reseed_locked((unsigned)seed);
  }
  // (k_first / on_first: called on this thread once draws [0, k_first) are in `out` -- the rest follows)
  void generate(size_t kmax, float* out, size_t k_first = 0, const std::function<void()>& on_first = nullptr) {
    generate_locked(kmax, out, k_first, on_first);
  }

  static DrawStream& global() {
    static DrawStream s;
    return s;
  }

 private:
  void reseed_locked(unsigned seed) {
    int32_t word = seed ? (int32_t)seed : 1;
    uint32_t r[31 + 310];
    r[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
      const int32_t hi = word / 127773, lo = word % 127773;
      word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      r[i] = (uint32_t)word;
    }
    // glibc starts with the front pointer 3 words ahead of the rear one: in history order (oldest
    // first) the table reads r[3], r[4], ..., r[30], r[0], r[1], r[2]; then 310 outputs are discarded
    uint32_t w[31 + 310];
    for (int i = 0; i < 31; ++i) w[i] = r[(i + 3) % 31];
    // the first three sums use the seed table's r[0..2] as the "3 steps ago" values
    for (int i = 0; i < 310; ++i) w[i + 31] = w[i] + w[i + 28];
    for (int i = 0; i < 31; ++i) hist_[i] = w[310 + i];
    raw_.clear();
  }
  // ---- jump-ahead.  State = the last 31 raw words, oldest first; one step: s' = (s[1..30], s[0] + s[28]).
  static constexpr size_t kSeg = 65536;
  struct Mat { uint32_t m[31][31]; };
  static void mat_mul(const Mat& a, const Mat& b, Mat* c) {
    for (int i = 0; i < 31; ++i)
      for (int j = 0; j < 31; ++j) {
        uint32_t t = 0;
        for (int k = 0; k < 31; ++k) t += a.m[i][k] * b.m[k][j];
        c->m[i][j] = t;
      }
  }
  static const Mat& jump() {   // M^kSeg by repeated squaring (kSeg = 2^16), built once
    static const Mat J = [] {
      Mat a{}, b{};
      for (int r = 0; r < 30; ++r) a.m[r][r + 1] = 1u;
      a.m[30][0] = 1u; a.m[30][28] = 1u;
      for (size_t p = 1; p < kSeg; p <<= 1) { mat_mul(a, a, &b); a = b; }
      return a;
    }();
    return J;
  }
  // (`snaps`: the state in front of every kSub-th draw of the segment, so that commit() replays at most kSub steps)
  static constexpr size_t kSub = 2048;
  static void fill_segment(const uint32_t* state, size_t len, float* out, std::vector<uint32_t>* scratch, uint32_t* snaps) {
    scratch->resize(31 + len);
    uint32_t* w = scratch->data();
    for (int i = 0; i < 31; ++i) w[i] = state[i];
    for (size_t i = 0; i < len; ++i) w[i + 31] = w[i] + w[i + 28];
    for (size_t i = 0; i < len; ++i) out[i] = (float)(w[i + 31] >> 1) / 2147483648.0f;
    for (size_t j = 0; j * kSub <= len; ++j)
      for (int i = 0; i < 31; ++i) snaps[j * 31 + i] = w[j * kSub + i];
  }
  // ---- a standing pool for the parallel requests (the stream is locked while it works: one job at a time)
  struct Pool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::thread> threads;
    std::function<void()> job;     // every worker runs it once per generation
    unsigned generation = 0;
    int running = 0;
    bool stop = false;
    explicit Pool(size_t n) {
      for (size_t t = 0; t < n; ++t)
        threads.emplace_back([this] {
          unsigned seen = 0;
          for (;;) {
            std::function<void()> f;
            {
              std::unique_lock<std::mutex> lk(mu);
              cv.wait(lk, [&] { return stop || generation != seen; });
              if (stop) return;
              seen = generation;
              f = job;
            }
            f();
            {
              std::lock_guard<std::mutex> lk(mu);
              --running;
            }
            cv.notify_all();
          }
        });
    }
    ~Pool() {
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      cv.notify_all();
      for (auto& t : threads) t.join();
    }
    void start(const std::function<void()>& f) {
      { std::lock_guard<std::mutex> lk(mu); job = f; running = (int)threads.size(); ++generation; }
      cv.notify_all();
    }
    void wait() {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return running == 0; });
    }
  };
  static Pool& pool() {
    static Pool p(std::min<size_t>(7, std::max(1u, std::thread::hardware_concurrency()) - 1));   // (+ the calling thread)
    return p;
  }
  // raw_[0..31) = the last 31 raw words (oldest first), raw_[31 + i] = the i-th not yet consumed word
  void generate_locked(size_t k, float* out, size_t k_first = 0, const std::function<void()>& on_first = nullptr) {
    seg_states_.clear();
    if (out && k >= 8 * kSeg && pool().threads.size() > 0) {   // large request: segment start states by jump-ahead, segments filled in parallel
      const size_t nseg = (k + kSeg - 1) / kSeg;
      const Mat& J = jump();
      seg_states_.resize((nseg + 1) * 31);
      for (int i = 0; i < 31; ++i) seg_states_[i] = hist_[i];
      for (size_t s = 0; s < nseg; ++s) {
        const uint32_t* a = &seg_states_[s * 31];
        uint32_t* b = &seg_states_[(s + 1) * 31];
        for (int i = 0; i < 31; ++i) {
          uint32_t t = 0;
          for (int c = 0; c < 31; ++c) t += J.m[i][c] * a[c];
          b[i] = t;
        }
      }
      constexpr size_t kSnapsPerSeg = kSeg / kSub + 1;
      sub_states_.resize(nseg * kSnapsPerSeg * 31);
      // the segments are taken in order (one counter); done[] says which are complete
      std::atomic<size_t> next{0};
      std::vector<std::atomic<unsigned char>> done(nseg);
      for (auto& d : done) d.store(0, std::memory_order_relaxed);
      auto work = [&] {
        std::vector<uint32_t> scratch;
        for (;;) {
          const size_t sg = next.fetch_add(1, std::memory_order_relaxed);
          if (sg >= nseg) return;
          fill_segment(&seg_states_[sg * 31], std::min(kSeg, k - sg * kSeg), out + sg * kSeg, &scratch,
                       &sub_states_[sg * kSnapsPerSeg * 31]);
          done[sg].store(1, std::memory_order_release);
        }
      };
      pool().start(work);
      const size_t first_segs = on_first ? std::min(nseg, (k_first + kSeg - 1) / kSeg) : 0;
      if (on_first && first_segs < nseg) {
        // this thread helps with the first part's segments, then tells the caller, then helps with the rest
        std::vector<uint32_t> scratch;
        for (;;) {
          const size_t seen = next.load(std::memory_order_relaxed);
          if (seen >= first_segs) break;
          const size_t sg = next.fetch_add(1, std::memory_order_relaxed);
          if (sg >= nseg) break;
          fill_segment(&seg_states_[sg * 31], std::min(kSeg, k - sg * kSeg), out + sg * kSeg, &scratch,
                       &sub_states_[sg * kSnapsPerSeg * 31]);
          done[sg].store(1, std::memory_order_release);
        }
        for (size_t sg = 0; sg < first_segs; ++sg)
          while (!done[sg].load(std::memory_order_acquire)) std::this_thread::yield();
        on_first();
      }
      work();
      pool().wait();
      if (on_first && first_segs >= nseg) on_first();
      seg_k_ = k;
      return;
    }
    raw_.resize(31 + k);
    uint32_t* w = raw_.data();
    for (int i = 0; i < 31; ++i) w[i] = hist_[i];
    for (size_t i = 0; i < k; ++i) w[i + 31] = w[i] + w[i + 28];
    if (out)
      for (size_t i = 0; i < k; ++i) out[i] = (float)(w[i + 31] >> 1) / 2147483648.0f;  // (float)RAND_MAX == 2^31
    if (on_first) on_first();
  }
  void commit_locked(size_t k) {
    if (!seg_states_.empty()) {   // (parallel request: replay from the last snapshot in front of draw k, < kSub steps)
      constexpr size_t kSnapsPerSeg = kSeg / kSub + 1;
      size_t sg = k / kSeg, in_seg = k % kSeg;
      if (sg * kSeg >= seg_k_ && sg > 0) { sg -= 1; in_seg = kSeg; }   // k == the end of the last (full) segment
      const size_t j = in_seg / kSub, r = in_seg % kSub;
      std::vector<uint32_t> w(31 + r);
      for (int i = 0; i < 31; ++i) w[i] = sub_states_[(sg * kSnapsPerSeg + j) * 31 + i];
      for (size_t i = 0; i < r; ++i) w[i + 31] = w[i] + w[i + 28];
      for (int i = 0; i < 31; ++i) hist_[i] = w[r + i];
      seg_states_.clear();
      return;
    }
    for (int i = 0; i < 31; ++i) hist_[i] = raw_[k + i];
  }
  std::mutex mu_;
  uint32_t hist_[31];
  std::vector<uint32_t> raw_;
  std::vector<uint32_t> seg_states_;  // parallel request: state at the start of every segment (+ one past the end)
  std::vector<uint32_t> sub_states_;  //   and in front of every kSub-th draw of every segment
  size_t seg_k_ = 0;
};

}  // namespace lsgpu
