// lsgpu_policy.h -- the launch policy of lsgpu_icp_align as an explicit state machine: WHAT gets enqueued next.
//
// ICP::compute's loop (laser_slam/src/laser_track.cpp:496 -> PointMatcher::ICP::compute, icp_default.yaml:9-27) runs on
// the device; the host only decides which kernels the next iteration is made of and when to look at the loop state.
// Rounds 2-4 kept those decisions in the lambdas of lsgpu_icp_align and inside run_knn (about thirty booleans); a defect
// shipped in round 4 because of it: the iteration enqueued behind a look consumed the direction index's re-pricing
// count before the look read it, so an alignment priced off the index never returned to it.  This header holds every
// such decision in one place, without HIP and without the handle, so that tests/cpp/policy_check.cpp can drive it on the
// CPU: the sequence of iterations of an alignment, the hand-over to the direction index, pricing, re-pricing and the
// repeat paths.  lsgpu_icp.hip executes what this header decides; it takes no launch decision of its own.
//
// None of the decisions changes a result (every path is an exact search / an exact order statistic); they decide what
// an iteration costs.
#pragma once
#include <cmath>
#include <cstdint>

namespace lsgpu {
namespace policy {

struct Config {               // constant during an alignment (from Tuning and the handle)
  int cone_from = 2;          // first iteration that may search the direction index
  int wide_iters = 3;         // the first iterations still have wide balls
  int group = 6;              // iterations between two looks at the loop state
  int enq_limit = 0;          // guards against a device that never finishes
  bool predict_select = true, commit_select = true, comm_commit = true;
  bool two_pass_select = false;   // the select stops after its second pass, the normal-equation kernel settles the rest (fused select)
  bool lookahead = true;      // one iteration enqueued behind a look's state copy
  bool comm = false;          // split-scan mode (RCCL): per-shard tables, no look-ahead
  bool seed_cap = true, cap_enabled = true;
  bool cone_probe = true;
  float cone_heavy_share = 0.07f;    // >= 2: the index is never priced
  float cone_max_occupancy = 7.f;
  double straggler_share = 0.02;     // lanes the index could not serve, per settled iteration, above which it is dropped
};

struct Iteration {            // one enqueued iteration: what the search, the select and the normal equations are told
  bool knn = true;            // false: select + normal equations + update on the distances already there (a missed prediction)
  bool seed = false, capped = true, wide = false;
  bool predicted = false;     // first half of the select in the search kernel's epilogue
  bool committed = false;     // ... and no select launch at all
  bool full_select = true;    // a select that is launched runs all three passes
  bool cone_iter = false;     // this search may go through the direction index
  bool dense_wait = false;    // (its first one: a denser reference waits one iteration more)
  bool price = false;         // this search prices the index for the one behind it
  int ordinal = 0;            // its place in the alignment (0 = the seeded first iteration)
};

enum class KnnKernel { Cone, ConeProbe, Tile };

// Across alignments of one handle (a track registers scan after scan of the same surroundings): the last voxel-grid search
// in front of the index's first use and the first settled search through the index are timed (two event pairs per
// alignment).  Where the index is clearly SLOWER than that earlier, wider voxel search, it is not paying at all -- a wall a
// metre from the sensor seen from three poses: measured 890 - 1107 us per search through the index against 344 - 425 for the
// voxel search two iterations before it (and 396 - 693 for the voxel grid in its place), with a price check that saw
// nothing (the heavy lanes are few, the launch is as long as its slowest wave) -- and the following alignments of the
// handle leave it alone, then try again.  The margin is wide on purpose: on a dense three-scan map the index' first
// search still has balls of centimetres and costs about what the voxel search before it did (188 - 364 us against ~250),
// yet it wins every later iteration (67 against 107 us).
constexpr int kIndexRestAligns = 8;
inline bool index_not_paying(float voxel_us, float index_us) { return voxel_us > 0.f && index_us > 1.5f * voxel_us; }

enum class LookVerdict { Continue, RepeatUncapped, RepeatSelect, Done, GiveUp };

struct LookInput {            // what a look at the loop state shows (IcpState fields)
  int done = 0, status = 0, iter = 0, sel_streak = 0;
  float chk_rot = 0.f, chk_trans = 0.f, lim_rot = 0.f, lim_trans = 0.f;   // the differential checker's smoothed changes and limits (0: not known)
  float chk_rot_prev = 0.f, chk_trans_prev = 0.f;                          // ... one iteration earlier
  unsigned long long stragglers = 0;
  int64_t nq = 0;
  int status_cap_failed = 100, status_sel_failed = 101;
};

struct State {
  // ---- the loop
  int enq = 0;                // iterations enqueued so far = ordinal of the next one
  int since_check = 0;        // ... since the last look
  bool first_select = true;
  bool commit_ok = false;
  int committed_iterations = 0, sel_retries = 0, cap_retries = 0;
  // ---- the direction index of the current reference, as this alignment sees it
  bool cone_ok = false;       // built (or being built)
  bool cone_decided = false;  // its occupancy has been looked at (per reference)
  bool cone_dense = false;    // ... and is above the limit: never used for this reference
  float cone_occupancy = 0.f;
  bool cone_off = false;      // this alignment stopped using it
  bool cone_off_price = false;  // ... because of its price: priced again by the last launch in front of every later look
  bool price_pending = false;   // a priced launch's counters are on their way to the host
  float cone_heavy = -1.f;      // last priced share of heavy lanes (-1: not priced)
  int cone_launches = 0;
  int look_iter = 0;
  unsigned long long look_strag = 0;
  // ---- the end of the alignment, as far as the checker's trend shows it: launches enqueued behind the iteration that
  // raises `done` exit at once but still cost ~10 us each, and a group of six put four or five of them there
  int group_now = 0;            // iterations of the group being enqueued (0: Config::group)
  float trend_rot = 0.f, trend_trans = 0.f;   // the smoothed changes at the last look ...
  int trend_iter = -1;          // ... and its iteration

  void begin_align(bool index_built, bool decided, bool dense, float occupancy) {
    *this = State();
    cone_ok = index_built; cone_decided = decided; cone_dense = dense; cone_occupancy = occupancy;
  }

  // ---- what the next iteration is made of
  Iteration plan(const Config& c, bool seed, bool capped, bool wide, bool knn, bool price_next) {
    Iteration it;
    it.knn = knn; it.seed = seed; it.capped = capped; it.wide = wide;
    // capped launches without a wave-per-query pass may fold the first half of the select into the search kernel; in the
    // split-scan mode only committed iterations do (the counts are per shard: one grouped exchange sums them)
    const bool can_commit = c.commit_select && commit_ok && !first_select && (!c.comm || c.comm_commit);
    it.predicted = c.predict_select && knn && capped && !wide && (!c.comm || can_commit);
    it.committed = it.predicted && can_commit;
    it.full_select = !c.two_pass_select;
    it.cone_iter = knn && !seed && capped && enq >= c.cone_from;
    it.dense_wait = enq == c.cone_from;
    it.price = knn && (enq == c.cone_from - 1 || price_next);
    it.ordinal = enq;
    if (!it.committed) first_select = false; else ++committed_iterations;
    return it;
  }

  // the loop between two looks: true -> enqueue a plain iteration (fills `it`), false -> look at the loop state
  bool next_in_group(const Config& c, Iteration* it) {
    if (!(enq < c.enq_limit && since_check < (group_now > 0 ? group_now : c.group))) return false;
    // an alignment the index was too dear for prices it again with the LAST launch in front of the look: the counters
    // travel in front of the look's state copy
    const bool price_next = cone_off_price && since_check == (group_now > 0 ? group_now : c.group) - 1;
    *it = plan(c, false, true, enq < c.wide_iters, true, price_next);
    ++enq; ++since_check;
    return true;
  }
  // the iteration that goes out behind a look's state copy (carries the decisions of the PREVIOUS look)
  bool lookahead_iteration(const Config& c, Iteration* it) {
    if (!(c.lookahead && !c.comm && enq < c.enq_limit)) return false;
    *it = plan(c, false, true, enq < c.wide_iters, true, false);
    ++enq;
    return true;
  }

  // ---- inside one search: the direction index
  bool pricing(const Config& c, const Iteration& it, bool in_loop) const {
    return it.price && it.capped && in_loop && !it.seed && cone_ok && (!cone_off || cone_off_price) && !cone_dense &&
           c.cone_heavy_share < 2.f;
  }
  void priced() { price_pending = true; }
  // first search that may use the index for this reference: the host needs the build's occupancy first
  bool wants_occupancy(const Iteration& it, bool in_loop) const {
    return it.capped && it.cone_iter && in_loop && cone_ok && !cone_off && !cone_decided;
  }
  void set_occupancy(const Config& c, float points_per_bin) {
    cone_occupancy = points_per_bin;
    cone_dense = points_per_bin > c.cone_max_occupancy;
    cone_decided = true;
  }
  // the FIRST price of an alignment is consumed by the first search that may use the index.  A RE-pricing count
  // (cone_off_price) is not: it belongs to the look in front of which it was launched -- the look-ahead iteration that
  // is enqueued behind that look's copy must leave it alone (the round-4 defect)
  bool wants_first_price(const Iteration& it) const { return it.cone_iter && price_pending && !cone_off_price; }
  void set_first_price(const Config& c, float heavy_share) {
    price_pending = false;
    cone_heavy = heavy_share;
    if (heavy_share > c.cone_heavy_share) { cone_off = true; cone_off_price = true; }
  }
  KnnKernel kernel(const Config& c, const Iteration& it, bool in_loop) {
    bool cone_iter = it.cone_iter;
    // a denser reference keeps the index out of one more iteration (its third search still has balls of centimetres)
    if (cone_iter && cone_decided && cone_occupancy > 3.f && it.dense_wait) cone_iter = false;
    if (it.capped && cone_iter && in_loop && cone_ok && !cone_off && !cone_dense) {
      ++cone_launches;
      return it.wide && c.cone_probe ? KnnKernel::ConeProbe : KnnKernel::Cone;
    }
    return KnnKernel::Tile;
  }

  // ---- a look at the loop state.  `repriced_share` < 0: no re-pricing count arrived with this look
  bool wants_reprice() const { return price_pending && cone_off_price; }
  LookVerdict on_look(const Config& c, const LookInput& s, int enqueued_ahead, float repriced_share) {
    since_check = enqueued_ahead;
    commit_ok = s.sel_streak >= 1 && s.status == 0;   // (a miss below clears it until the streak is rebuilt)
    // lanes the index cannot serve search the voxel grid one by one and are counted as stragglers: a handful on the
    // clouds it is made for; where they are not, the rest of this alignment goes back to the voxel grid
    const int settled_from = c.cone_from > c.wide_iters ? c.cone_from : c.wide_iters;
    if (cone_ok && !cone_off && look_iter >= settled_from && s.iter > look_iter &&
        (double)(s.stragglers - look_strag) > c.straggler_share * (double)s.nq * (double)(s.iter - look_iter))
      cone_off = true;
    look_iter = s.iter; look_strag = s.stragglers;
    group_now = estimate_group(c, s, enqueued_ahead);
    if (wants_reprice() && repriced_share >= 0.f) {   // priced again by the last launch in front of this look: cheap enough by now?
      cone_heavy = repriced_share;
      if (repriced_share <= c.cone_heavy_share) { cone_off = false; cone_off_price = false; }
      price_pending = false;
    }
    if (s.done && s.status == s.status_cap_failed) { ++cap_retries; return LookVerdict::RepeatUncapped; }
    if (s.done && s.status == s.status_sel_failed) { ++sel_retries; commit_ok = false; return LookVerdict::RepeatSelect; }
    if (s.done) return LookVerdict::Done;
    if (enq >= c.enq_limit) return LookVerdict::GiveUp;
    return LookVerdict::Continue;
  }
  // How many iterations the next group should hold: the smoothed changes shrink geometrically towards their limits (the
  // factor per iteration from this look and the last, between 0.5 and 0.97); the alignment ends with the first iteration
  // at which BOTH are below.  One iteration of margin; never more than Config::group, never fewer than one.
  int estimate_group(const Config& c, const LookInput& s, int enqueued_ahead) {
    int g = c.group;
    const bool known = s.chk_trans > 0.f && s.chk_rot >= 0.f && s.lim_trans > 0.f && s.lim_rot > 0.f;
    const bool two_looks = trend_iter >= 0 && s.iter > trend_iter && trend_trans > 0.f;
    if (known && (two_looks || s.chk_trans_prev > 0.f)) {
      // the factor per iteration: from this look and the last one, or -- at a first look -- from the last two iterations
      const int span = two_looks ? s.iter - trend_iter : 1;
      const float then_t = two_looks ? trend_trans : s.chk_trans_prev, then_r = two_looks ? trend_rot : s.chk_rot_prev;
      auto left = [&](float now, float then, float lim) -> float {
        if (!(now > lim)) return 0.f;
        float f = then > 0.f && now < then ? std::pow(now / then, 1.f / (float)span) : 0.97f;
        f = f < 0.5f ? 0.5f : f > 0.97f ? 0.97f : f;
        return std::log(now / lim) / std::log(1.f / f);
      };
      const float rem = std::fmax(left(s.chk_trans, then_t, s.lim_trans), left(s.chk_rot, then_r, s.lim_rot));
      // the group counts the iteration that is already out behind this look (next_in_group starts at `enqueued_ahead`):
      // the iterations still needed + one of margin
      (void)enqueued_ahead;
      const int want = (int)std::ceil(rem) + 1;
      g = want < 1 ? 1 : want > c.group ? c.group : want;
    }
    if (known) { trend_rot = s.chk_rot; trend_trans = s.chk_trans; trend_iter = s.iter; }
    return g;
  }
  // the repeat paths: the cap prediction failed (repeat the iteration uncapped), the select prediction missed (the
  // distances stand: select + normal equations again, no search)
  Iteration repeat_uncapped(const Config& c) {
    Iteration it = plan(c, false, false, true, true, false);
    ++enq; since_check = 1;
    return it;
  }
  Iteration repeat_select(const Config& c) {
    Iteration it = plan(c, false, true, false, false, false);
    it.full_select = true;   // (whatever voided the iteration: this time the limit comes from the select itself)
    since_check = 1;
    return it;
  }
};

}  // namespace policy
}  // namespace lsgpu
