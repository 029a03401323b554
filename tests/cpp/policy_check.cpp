// policy_check.cpp -- the launch policy of lsgpu_icp_align (csrc/lsgpu_policy.h) driven on the CPU: WHAT gets enqueued
// next, without a GPU.  A toy "device" stands in for the loop state: it finishes after a given number of iterations and
// reports the price counters the test scripts for it.  Checked here:
//   * the plain sequence of an alignment (seeded first iteration, the pricing launch in front of the index's first use,
//     the hand-over to the direction index, groups of six with a look-ahead iteration behind every look);
//   * an alignment whose first price is too high stays on the voxel grid, prices again with the LAST launch in front of
//     every look, and RETURNS to the index once a look finds it cheap -- direction_index_launches grows after that look
//     (the round-4 defect: the look-ahead iteration consumed the re-pricing count, the index never came back);
//   * the look-ahead iteration never consumes a re-pricing count;
//   * the repeat paths (failed cap prediction, missed select prediction) and the committed select;
//   * the split-scan mode: no look-ahead, predicted select only in committed iterations.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../laser_slam_amd/csrc/lsgpu_policy.h"

using namespace lsgpu::policy;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "policy_check: %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

struct Enq { Iteration it; KnnKernel kern; bool priced; bool lookahead; };

struct Sim {
  Config c;
  State s;
  std::vector<Enq> log;
  int device_done_after = 20;          // the device raises `done` once this many searches ran
  std::vector<float> reprice_shares;   // what the re-pricing counts say, look by look
  float first_price = 0.01f;
  float occupancy = 2.0f;
  size_t look_no = 0;
  int searches = 0;
  int sel_streak_from = 1 << 30;       // iteration from which the device reports a steady limit
  int fail_cap_at = -1, fail_sel_at = -1;
  float trend0 = 0.f, trend_factor = 0.f;   // the checker's smoothed translation change: trend0 * factor^iteration (0: the device reports none)

  void run_search(const Iteration& it, bool lookahead) {
    // exactly what run_knn does with the policy (lsgpu_icp.hip)
    const bool pricing = s.pricing(c, it, true);
    if (s.wants_occupancy(it, true)) s.set_occupancy(c, occupancy);
    if (s.wants_first_price(it)) s.set_first_price(c, first_price);
    const KnnKernel k = s.kernel(c, it, true);
    if (k == KnnKernel::Tile && pricing) s.priced();
    log.push_back({it, k, pricing && k == KnnKernel::Tile, lookahead});
    ++searches;
  }
  void enqueue(const Iteration& it, bool lookahead = false) {
    if (it.knn) run_search(it, lookahead);
    else log.push_back({it, KnnKernel::Tile, false, lookahead});
  }
  // returns the number of looks
  int align(bool index_built) {
    s.begin_align(index_built, false, false, 0.f);
    enqueue(s.plan(c, true, c.seed_cap && c.cap_enabled, true, true, false));
    s.enq = 1; s.since_check = 1;
    int looks = 0;
    for (;;) {
      Iteration it;
      if (s.next_in_group(c, &it)) { enqueue(it); continue; }
      ++looks;
      const bool repriced = s.wants_reprice();
      const int iter_at_copy = searches;          // the state copy sees the iterations enqueued in front of it
      int ahead = 0;
      Iteration la;
      if (s.lookahead_iteration(c, &la)) { enqueue(la, true); ahead = 1; }
      LookInput li;
      li.nq = 1000000;
      li.iter = iter_at_copy < device_done_after ? iter_at_copy : device_done_after;
      li.done = iter_at_copy >= device_done_after;
      li.sel_streak = li.iter >= sel_streak_from ? 3 : 0;
      if (trend0 > 0.f) { li.lim_rot = 1e-5f; li.lim_trans = 1e-4f; li.chk_rot = 1e-6f; li.chk_trans = trend0 * std::pow(trend_factor, (float)li.iter); }
      if (fail_cap_at >= 0 && iter_at_copy > fail_cap_at) { li.done = 1; li.status = li.status_cap_failed; li.iter = fail_cap_at; fail_cap_at = -1; }
      else if (fail_sel_at >= 0 && iter_at_copy > fail_sel_at) { li.done = 1; li.status = li.status_sel_failed; li.iter = fail_sel_at; fail_sel_at = -1; }
      float share = -1.f;
      if (repriced) { share = look_no < reprice_shares.size() ? reprice_shares[look_no] : 1.f; }
      ++look_no;
      const LookVerdict v = s.on_look(c, li, ahead, share);
      if (v == LookVerdict::RepeatUncapped) { enqueue(s.repeat_uncapped(c)); continue; }
      if (v == LookVerdict::RepeatSelect) { enqueue(s.repeat_select(c)); continue; }
      if (v == LookVerdict::Done) return looks;
      CHECK(v == LookVerdict::Continue);
      CHECK(looks < 100);
    }
  }
};

static Config base_config() {
  Config c;
  c.enq_limit = 8 * 40 + 64;
  return c;
}

int main() {
  {  // ---- the plain sequence
    Sim m; m.c = base_config(); m.device_done_after = 20;
    const int looks = m.align(true);
    CHECK(m.log[0].it.seed && m.log[0].it.capped && m.log[0].it.wide && m.log[0].kern == KnnKernel::Tile && !m.log[0].priced);
    CHECK(!m.log[1].it.seed && m.log[1].it.wide && m.log[1].kern == KnnKernel::Tile && m.log[1].priced);   // iteration cone_from - 1 prices
    CHECK(m.log[2].kern == KnnKernel::ConeProbe && m.log[2].it.dense_wait);                               // iteration 2: the index, still wide: with the probe
    for (size_t i = 3; i < m.log.size(); ++i) CHECK(m.log[i].kern == KnnKernel::Cone && !m.log[i].priced);
    CHECK(m.s.cone_decided && !m.s.cone_dense && !m.s.cone_off && m.s.cone_heavy == m.first_price);
    // groups of six: looks after 6, 12 (+1 ahead each), ... ; the iteration behind every look is marked
    CHECK(m.log[6].lookahead && m.log[12].lookahead && !m.log[5].lookahead && !m.log[7].lookahead);
    CHECK(looks == 4 && m.s.cone_launches == (int)m.log.size() - 2);
    for (auto& e : m.log) CHECK(!e.it.committed);   // the device never reported a steady limit
  }
  {  // ---- the last groups follow the checker's trend: the smoothed change shrinks by 0.86 per iteration and falls below its
     // limit at iteration 32 -- no more than two launches may be left behind the end (a fixed group of six leaves up to six)
    Sim m; m.c = base_config(); m.trend0 = 1e-4f / std::pow(0.86f, 31.5f); m.trend_factor = 0.86f; m.device_done_after = 32;
    const int looks = m.align(true);
    CHECK((int)m.log.size() >= 32 && (int)m.log.size() <= 32 + 2);
    CHECK(looks >= 6 && looks <= 10);
    Sim f; f.c = base_config(); f.device_done_after = 32;     // no trend reported: groups of six as ever
    f.align(true);
    CHECK((int)f.log.size() > 32 + 2);
    // a trend that stalls (factor 1) must not shrink the groups to nothing for ever: the estimate is clamped, groups stay >= 1
    Sim st; st.c = base_config(); st.trend0 = 5e-4f; st.trend_factor = 1.0f; st.device_done_after = 40;
    CHECK(st.align(true) < 100);
  }
  {  // ---- a denser reference: the index waits one iteration more; a too dense one is never used
    Sim m; m.c = base_config(); m.occupancy = 4.3f; m.align(true);
    CHECK(m.log[2].kern == KnnKernel::Tile && m.log[3].kern == KnnKernel::Cone);
    Sim d; d.c = base_config(); d.occupancy = 11.8f; d.align(true);
    for (auto& e : d.log) CHECK(e.kern == KnnKernel::Tile);
    CHECK(d.s.cone_dense && d.s.cone_launches == 0);
    Sim n; n.c = base_config(); n.align(false);      // no index at all: nothing priced, nothing waited for
    for (auto& e : n.log) CHECK(e.kern == KnnKernel::Tile && !e.priced);
  }
  {  // ---- priced off, priced again before every look, back on the index once it is cheap (the round-4 defect)
    Sim m; m.c = base_config(); m.device_done_after = 30;
    m.first_price = 0.30f;                       // > 0.07: off
    m.reprice_shares = {0.20f, 0.10f, 0.03f};    // looks 0, 1: still dear; look 2: cheap
    m.align(true);
    CHECK(m.log[1].priced && m.log[2].kern == KnnKernel::Tile);     // consumed by iteration 2: too dear
    // every group's LAST launch in front of a look prices; the look-ahead iteration behind the look never does
    std::vector<size_t> look_ahead_at;
    for (size_t i = 0; i < m.log.size(); ++i) if (m.log[i].lookahead) look_ahead_at.push_back(i);
    CHECK(look_ahead_at.size() >= 4);
    for (size_t q = 0; q < 3; ++q) {
      const size_t la = look_ahead_at[q];
      CHECK(m.log[la - 1].priced);               // the launch in front of the look
      CHECK(!m.log[la].priced);                  // the one behind it
    }
    // looks 0 and 1 leave it on the voxel grid; the look-ahead iteration of look 2 still carries the previous look's
    // decision (voxel grid); everything after it searches the index again
    for (size_t i = 2; i <= look_ahead_at[2]; ++i) CHECK(m.log[i].kern == KnnKernel::Tile);
    int cone_after = 0;
    for (size_t i = look_ahead_at[2] + 1; i < m.log.size(); ++i) { CHECK(m.log[i].kern == KnnKernel::Cone && !m.log[i].priced); ++cone_after; }
    CHECK(cone_after >= 5 && m.s.cone_launches == cone_after);     // direction_index_launches grows after the look
    CHECK(!m.s.cone_off && !m.s.cone_off_price && m.s.cone_heavy == 0.03f);
  }
  {  // ---- an alignment that never gets cheap stays off, and keeps pricing
    Sim m; m.c = base_config(); m.first_price = 0.5f; m.reprice_shares = {0.5f, 0.4f, 0.3f, 0.2f}; m.align(true);
    for (auto& e : m.log) CHECK(e.kern == KnnKernel::Tile);
    CHECK(m.s.cone_off && m.s.cone_off_price && m.s.cone_launches == 0 && m.s.cone_heavy > 0.07f);
  }
  {  // ---- LSGPU_CONE_HEAVY_SHARE=2: never priced
    Sim m; m.c = base_config(); m.c.cone_heavy_share = 2.f; m.first_price = 0.9f; m.align(true);
    for (auto& e : m.log) CHECK(!e.priced);
    CHECK(m.log[2].kern == KnnKernel::ConeProbe && m.s.cone_heavy == -1.f);
  }
  {  // ---- stragglers: the index is dropped when too many lanes could not be served
    Config c = base_config();
    State s; s.begin_align(true, true, false, 2.f);
    LookInput li; li.nq = 1000; li.iter = 6; li.stragglers = 5;
    CHECK(s.on_look(c, li, 1, -1.f) == LookVerdict::Continue && !s.cone_off);   // first look: only remembers
    li.iter = 12; li.stragglers = 5 + 6 * 30;                                   // 30 per iteration = 3 % of the queries
    CHECK(s.on_look(c, li, 1, -1.f) == LookVerdict::Continue && s.cone_off && !s.cone_off_price);
  }
  {  // ---- committed select: once a look reports a steady limit the select launches stop; a miss repeats the select only
    Sim m; m.c = base_config(); m.device_done_after = 26; m.sel_streak_from = 5; m.fail_sel_at = 14; m.align(true);
    int committed = 0, select_only = 0;
    for (auto& e : m.log) { committed += e.it.committed; select_only += !e.it.knn; if (e.it.committed) CHECK(e.it.predicted); }
    CHECK(committed > 5 && select_only == 1 && m.s.sel_retries == 1 && m.s.committed_iterations == committed);
    for (size_t i = 0; i < 7; ++i) CHECK(!m.log[i].it.committed);   // nothing commits before the first look said so
    // the iteration right behind the repeated select is not committed (the streak has to be rebuilt)
    for (size_t i = 0; i + 1 < m.log.size(); ++i) if (!m.log[i].it.knn) CHECK(!m.log[i].it.committed && !m.log[i + 1].it.committed);
  }
  {  // ---- failed cap prediction: one uncapped, wide repeat, then the loop carries on
    Sim m; m.c = base_config(); m.device_done_after = 18; m.fail_cap_at = 8; m.align(true);
    int uncapped = 0;
    for (size_t i = 1; i < m.log.size(); ++i) if (!m.log[i].it.capped) { ++uncapped; CHECK(m.log[i].it.wide && m.log[i].kern == KnnKernel::Tile); }
    CHECK(uncapped == 1 && m.s.cap_retries == 1);
  }
  {  // ---- split-scan mode: no look-ahead; the search kernels fold the select in only in committed iterations
    Sim m; m.c = base_config(); m.c.comm = true; m.sel_streak_from = 5; m.device_done_after = 20; m.align(false);
    for (auto& e : m.log) { CHECK(!e.lookahead); CHECK(e.it.predicted == e.it.committed); }
    Sim n; n.c = base_config(); n.c.comm = true; n.c.comm_commit = false; n.sel_streak_from = 0; n.align(false);
    for (auto& e : n.log) CHECK(!e.it.predicted && !e.it.committed);
  }
  {  // ---- is the index paying at all on this handle's clouds?  (two timed launches per alignment, across alignments)
    CHECK(!index_not_paying(228.f, 60.f));     // the benchmark pair: voxel search of iteration 1 against the index' first settled search
    CHECK(!index_not_paying(124.f, 53.f));     // an ordinary scan of the track drive
    CHECK(index_not_paying(425.f, 955.f));     // a wall a metre from the sensor (scan 19 of the drive)
    CHECK(index_not_paying(344.f, 890.f));
    CHECK(!index_not_paying(250.f, 364.f));    // a dense three-scan map: the index' first search is no faster yet, the later ones are
    CHECK(!index_not_paying(0.f, 50.f));       // nothing timed: nothing decided
    CHECK(kIndexRestAligns >= 2);
  }
  std::printf("policy_check ok\n");
  return 0;
}
