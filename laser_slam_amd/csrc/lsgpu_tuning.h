// lsgpu_tuning.h -- the ONE place where liblsgpu_icp.so looks at its environment.
//
// Every LSGPU_* variable is read once per process by tuning(), range-checked (an out-of-range or unparsable value is
// reported on stderr and replaced by the default -- it never aborts a registration), and kept in this struct; the rest
// of the library reads the struct.  None of the switches changes a result: they move work between exact paths
// (tests/test_gpu_parity.py::test_experiment_switches_do_not_change_results) or size scratch resources.  A variable
// that starts with LSGPU_ and is not listed here is reported once (a typo would otherwise silently run the default).
//
// Product switches (defaults are the measured optima, DESIGN.md "switches"):
//   LSGPU_QUERY_ORDER      -1   order of the queries inside the waves: -1 automatic, 0 Morton, 1 spherical cells
//   LSGPU_Q_ELEV / _Q_SECT  0   elevation bin / azimuth sector of the spherical cells in degrees (0: from the density)
//   LSGPU_GAP             0.002 metres searched beyond the current best in capped launches (keep-match bound)
//   LSGPU_BUDGET           512  chunks in a spread wave's cell block above which its lanes search on their own (settled launches;
//                               measured 128 / 256 / 512 / 1024: 89.9 / 88.3 / 86.4 / 91.5 us per launch, profiles/r03_knn_variants.txt)
//   LSGPU_BUDGET_WIDE     1024  the same threshold in the wide launches (first iterations)
//   LSGPU_WIDE_ITERS         3  first iterations of an align whose wide-ball spread waves go to the wave-per-query pass
//   LSGPU_ROUTE_R         0.02  ... if their largest ball exceeds this (metres)
//   LSGPU_ROUTE_CHUNKS    1024  ... or the cell block holds more chunks than this
//   LSGPU_ROUTE_HEAVY_MAX 1024  ... but only the first so many such tiles of a launch (a ticket); the others stay in the tile kernel.  A few
//                               heavy tiles are a launch's tail, thousands (a 0.29 m / 1.5 deg guess on an 8-scan local map,
//                               devtools/split_first.py: 468 k + 165 k queries handed over in the first two searches, 3.8 + 1.2 ms)
//                               cost the wave-per-query pass the map's density once per query.  -1: all of them (rounds 3-5)
//   LSGPU_ROUTE_DENSE     512   a spread wave whose cell block holds more chunks than this goes there whatever its balls (a wall
//                               next to the sensor: 64 lanes each walking 1 700 chunk boxes held one wave for 550 us)
//   LSGPU_NO_PREDICT            always the three-pass select (no first two passes in the search epilogue)
//   LSGPU_NO_COMMIT             keep launching the select kernels when the limit is steady
//   LSGPU_NO_COMM_COMMIT        the same, only for handles with a communicator (split-scan mode)
//   LSGPU_NO_SEED_CAP           first search uncapped (no quantile of the seed distances)
//   LSGPU_NO_FRONT              spread tiles go through the separate row pass instead of the front of the tile launch
//   LSGPU_FRONT_GUESS     2048  tiles the front of the grid is sized for before the host has seen the list
//   LSGPU_NO_LAZY               wide launches (first iterations) use the plain tile kernel instead of the instantiation that re-tests chunks before fetching them
//   LSGPU_NO_SPLIT              settled launches of the voxel search evaluate every candidate in all 64 lanes (no lane split, lsgpu_knn.hip.h)
//   LSGPU_NO_SIDE_STREAM        lsgpu_icp_compute: reading filter + query order AFTER the grid build, on the same stream (not beside it)
//   LSGPU_NO_LOOKAHEAD          no iteration enqueued behind the copy of the loop state: the device idles while the host looks at it
//   LSGPU_NO_ROUTE_ALL          (with NO_FRONT) settled spread waves search per lane inside the tile kernel
//   LSGPU_NO_ROWQ               (with NO_FRONT) handed-over queries go to the wave-per-query kernel
//   LSGPU_ROWQ_BLOCKS     2048  (with NO_FRONT) grid of the row pass
//   LSGPU_SORT_ITEMS         0  keys per thread of the radix passes (0: by size; 4, 8, 16)
//   LSGPU_SSN_FULL_SORT         the reference filter's levels as whole-cloud sorts by (segment, coordinate) (rounds 1-3) instead of segmented sorts
//   LSGPU_SSN_GLOBAL            every level of the reference filter as a global sort (no in-LDS finish; implies LSGPU_SSN_SORT_LEVELS)
//   LSGPU_SSN_SORT_LEVELS       the upper levels of the reference filter with a segmented sort per level (round 4, lsgpu_segsort.hip.h) instead of
//                               the sort-free levels of lsgpu_ssn_select.hip.h (exact median by selection + one stable partition)
//   LSGPU_SSN_OLD_FINISH        the last levels with k_ssn_finish (rounds 2-4: 2048 points per workgroup, a radix sort per level) instead of
//                               k_ssn_tree (presorted axes, a stable partition per level); continues a sorted order: implies LSGPU_SSN_SORT_LEVELS
//   LSGPU_SSN_ROOT        8192  points per workgroup of k_ssn_tree (2048, 4096, 8192)
//   LSGPU_NE_BLOCKS        256  blocks of k_normal_eq_loop (64 .. 2048)
//   LSGPU_SPLIT_UPDATE          the per-iteration update as its own launch (profiling)
//   LSGPU_COMM_TIMEOUT_MS 30000 bound on every stream wait of the split-scan mode
//   LSGPU_CONE_HEAVY_STEPS / _SHARE  price check before the first search through the index: lanes whose windows would hold more
//                               than STEPS (1024) evaluation steps are heavy; more than SHARE (0.07; 2 = never) of them: voxel grid
//   LSGPU_NO_CONE_PROBE         no probe of the query's own direction in the index's wide launches (iterations < LSGPU_WIDE_ITERS)
//   LSGPU_NO_CONE               settled launches search the voxel grid (k_knn_tile) instead of the direction index (k_knn_cone)
//   LSGPU_CONE_FROM          2  first iteration of an align that searches the direction index (0 and 1 have balls as wide as the
//                               initial guess is off: measured 464 us for iteration 1 on the benchmark pair against 230 with the voxel grid)
//   LSGPU_CONE_MAX_OCC       7  reference points per occupied bin of the direction index above which the settled launches use the voxel grid
//                               (local maps of K 1 M-point scans: K = 3 -> 4.3: 67 us per launch against 107; K = 5 -> 8.5: 147 / 140; K = 6 -> 9.8: 249 / 161)
//   LSGPU_CONE_ROWS        128  rows (bins of the sine of the elevation) of the direction index
//   LSGPU_CONE_COLS       8192  columns (bins of the pseudo-azimuth) of the direction index
//   LSGPU_KNN_DBG            0  ablation flags of the -DLSGPU_KNN_STATS build (ignored by the product build)
// Experiment switches, compiled in only with -DLSGPU_EXPERIMENTS (measured-slower variants kept as the record of what was
// tried: DESIGN.md "Rejected after measurement"); the product build reports them as unknown:
//   LSGPU_KNN_ROWS (0/1/2), LSGPU_KNN_LANE, LSGPU_SPARSE_LANES, LSGPU_TILE_WAVES (1/4), LSGPU_XCD_SWIZZLE,
//   LSGPU_ROCPRIM_SORT (rocPRIM's radix sort instead of lsgpu_sort.hip.h: the library sort as a cross-check)

#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern char** environ;

namespace lsgpu {

struct Tuning {
  int query_order = -1;
  float q_elev = 0.f, q_sect = 0.f;
  float gap = 0.002f;
  int chunk_budget = 512, chunk_budget_wide = 1024;
  int wide_iters = 3;
  float route_r = 0.02f;
  int route_chunks = 1024;
  int route_heavy_max = 1024;      // LSGPU_ROUTE_HEAVY_MAX
  int route_dense = 512;           // ... and a SPREAD wave whose cell block holds more chunks than this goes there whatever its balls (LSGPU_ROUTE_DENSE)
  bool predict_select = true, commit_select = true, comm_commit = true, seed_cap = true;
  int sel_amb_cap = 256;      // ... setting aside at most this many distances of the limit's slice (<= kSelAmbCap; a test knob: the sums' order changes with it)
  bool two_pass_select = true;   // ... also where the select kernels run: they stop after their second pass (LSGPU_THREE_PASS_SELECT)
  bool short_last_group = true;   // the host sizes a group of launches from the checker's trend (LSGPU_FULL_GROUPS: always six)
  bool fused_select = true;   // the normal-equation kernel finds the trim limit itself (LSGPU_NO_FUSED_SELECT: the select kernels / the window table)
  bool front = true;
  bool lazy_need = true;
  bool lane_split = true;    // LSGPU_NO_SPLIT: k_knn_tile<1, false> instead of <1, false, true> for the settled voxel searches
  bool lookahead = true;     // LSGPU_NO_LOOKAHEAD: the host waits for the whole stream when it looks at the loop state
  bool index_rest = true;    // LSGPU_NO_INDEX_REST: the handle never rests the direction index on the evidence of two timed searches
                             // (the one launch decision that depends on wall-clock timings: off => the same kernels every run)
  bool side_stream = true;   // LSGPU_NO_SIDE_STREAM: lsgpu_icp_compute runs the reading's filter + query order after the grid build instead of beside it
  int front_guess = 2048;
  bool route_all = true, rowq = true;
  int rowq_blocks = 2048;
  int sort_items = 0;
  bool ssn_global = false;
  bool ssn_full_sort = false;
  bool ssn_old_finish = false;
  bool ssn_sort_levels = false;
  int ssn_root = 0;   // points per root of k_ssn_tree (2048 / 4096 / 8192); 0: by the cloud's size, so that there are about as many roots as compute units
  int ne_blocks = 256;
  bool split_update = false;
  double comm_timeout_ms = 30000.0;
  int knn_dbg = 0;
  bool cone = true;
  bool cone_probe = true;
  int cone_rows = 128, cone_cols = 8192, cone_from = 2;
  float cone_max_occupancy = 7.0f;
  float cone_heavy_steps = 1024.f;   // the index's price check: a lane whose windows would hold more steps of four than this is heavy
  float cone_heavy_share = 0.07f;   // ... and an align with more than this share of heavy lanes among the searching ones keeps the voxel grid
#ifdef LSGPU_EXPERIMENTS
  int knn_rows = 0;
  bool knn_lane = false;
  int sparse_lanes = 0;
  int tile_waves = 1;
  int xcd_swizzle = 0;
  bool rocprim_sort = false;
#endif
};

namespace tuning_detail {
inline bool flag(const char* name) { return getenv(name) != nullptr; }
inline double number(const char* name, double def, double lo, double hi) {
  const char* e = getenv(name);
  if (!e) return def;
  char* end = nullptr;
  const double v = strtod(e, &end);
  if (end == e || *end != '\0' || !(v >= lo && v <= hi)) {
    fprintf(stderr, "liblsgpu_icp: %s=%s is not a number in [%g, %g]; using the default %g\n", name, e, lo, hi, def);
    return def;
  }
  return v;
}
inline Tuning read() {
  Tuning t;
  t.query_order = (int)number("LSGPU_QUERY_ORDER", -1, -1, 1);
  t.q_elev = (float)number("LSGPU_Q_ELEV", 0, 0, 45);
  t.q_sect = (float)number("LSGPU_Q_SECT", 0, 0, 45);
  t.gap = (float)number("LSGPU_GAP", 0.002, 0, 1);
  t.chunk_budget = (int)number("LSGPU_BUDGET", 512, 1, 1 << 20);
  t.chunk_budget_wide = (int)number("LSGPU_BUDGET_WIDE", 1024, 1, 1 << 20);
  t.wide_iters = (int)number("LSGPU_WIDE_ITERS", 3, 0, 1 << 20);
  t.route_r = (float)number("LSGPU_ROUTE_R", 0.02, 1e-6, 1e6);
  t.route_chunks = (int)number("LSGPU_ROUTE_CHUNKS", 1024, 1, 1 << 30);
  t.route_heavy_max = (int)number("LSGPU_ROUTE_HEAVY_MAX", 1024, -1, 1 << 30);
  t.route_dense = (int)number("LSGPU_ROUTE_DENSE", 512, 1, 1 << 30);
  t.split_update = flag("LSGPU_SPLIT_UPDATE");
  t.predict_select = !flag("LSGPU_NO_PREDICT") && !t.split_update;
  t.commit_select = !flag("LSGPU_NO_COMMIT");
  t.fused_select = !flag("LSGPU_NO_FUSED_SELECT");
  t.short_last_group = !flag("LSGPU_FULL_GROUPS");
  t.two_pass_select = !flag("LSGPU_THREE_PASS_SELECT");
  t.sel_amb_cap = (int)number("LSGPU_SEL_AMB_CAP", 256, 0, 256);
  t.comm_commit = !flag("LSGPU_NO_COMM_COMMIT");
  t.seed_cap = !flag("LSGPU_NO_SEED_CAP");
  t.front = !flag("LSGPU_NO_FRONT");
  t.lazy_need = !flag("LSGPU_NO_LAZY");
  t.lane_split = !flag("LSGPU_NO_SPLIT");
  t.side_stream = !flag("LSGPU_NO_SIDE_STREAM");
  t.lookahead = !flag("LSGPU_NO_LOOKAHEAD");
  t.index_rest = !flag("LSGPU_NO_INDEX_REST");
  t.front_guess = (int)number("LSGPU_FRONT_GUESS", 2048, 0, 8192);
  t.route_all = !flag("LSGPU_NO_ROUTE_ALL");
  t.rowq = !flag("LSGPU_NO_ROWQ");
  t.rowq_blocks = (int)number("LSGPU_ROWQ_BLOCKS", 2048, 1, 65535);
  t.sort_items = (int)number("LSGPU_SORT_ITEMS", 0, 0, 16);
  if (t.sort_items != 0 && t.sort_items != 4 && t.sort_items != 8 && t.sort_items != 16) {
    fprintf(stderr, "liblsgpu_icp: LSGPU_SORT_ITEMS must be 4, 8 or 16; choosing by size\n");
    t.sort_items = 0;
  }
  t.ssn_global = flag("LSGPU_SSN_GLOBAL");
  t.ssn_full_sort = flag("LSGPU_SSN_FULL_SORT");
  t.ssn_old_finish = flag("LSGPU_SSN_OLD_FINISH");
  t.ssn_sort_levels = flag("LSGPU_SSN_SORT_LEVELS");
  t.ssn_root = (int)number("LSGPU_SSN_ROOT", 0, 0, 8192);
  if (t.ssn_root != 0 && t.ssn_root != 2048 && t.ssn_root != 4096 && t.ssn_root != 8192) {
    fprintf(stderr, "liblsgpu_icp: LSGPU_SSN_ROOT must be 2048, 4096 or 8192; choosing by size\n");
    t.ssn_root = 0;
  }
  t.ne_blocks = (int)number("LSGPU_NE_BLOCKS", 256, 64, 2048);
  t.comm_timeout_ms = number("LSGPU_COMM_TIMEOUT_MS", 30000, 1, 1e9);
  t.knn_dbg = (int)number("LSGPU_KNN_DBG", 0, 0, 1 << 20);
  t.cone = !flag("LSGPU_NO_CONE");
  t.cone_probe = !flag("LSGPU_NO_CONE_PROBE");
  t.cone_from = (int)number("LSGPU_CONE_FROM", 2, 1, 1 << 20);
  t.cone_max_occupancy = (float)number("LSGPU_CONE_MAX_OCC", 7.0, 0.0, 1e9);
  t.cone_heavy_steps = (float)number("LSGPU_CONE_HEAVY_STEPS", 1024.0, 0.0, 1e9);
  t.cone_heavy_share = (float)number("LSGPU_CONE_HEAVY_SHARE", 0.07, 0.0, 2.0);
  t.cone_rows = (int)number("LSGPU_CONE_ROWS", 128, 8, 1024);
  t.cone_cols = (int)number("LSGPU_CONE_COLS", 8192, 64, 65536) & ~3;
  static const char* known[] = {"LSGPU_QUERY_ORDER", "LSGPU_Q_ELEV", "LSGPU_Q_SECT", "LSGPU_GAP", "LSGPU_BUDGET", "LSGPU_BUDGET_WIDE", "LSGPU_WIDE_ITERS",
                                "LSGPU_ROUTE_R", "LSGPU_ROUTE_CHUNKS", "LSGPU_ROUTE_HEAVY_MAX", "LSGPU_ROUTE_DENSE", "LSGPU_SPLIT_UPDATE", "LSGPU_NO_PREDICT", "LSGPU_NO_COMMIT",
                                "LSGPU_NO_COMM_COMMIT", "LSGPU_NO_SEED_CAP", "LSGPU_NO_FRONT", "LSGPU_NO_LAZY", "LSGPU_NO_SPLIT", "LSGPU_NO_SIDE_STREAM", "LSGPU_NO_LOOKAHEAD", "LSGPU_NO_INDEX_REST", "LSGPU_CELLS_SPLIT", "LSGPU_NO_FUSED_SELECT", "LSGPU_FULL_GROUPS", "LSGPU_THREE_PASS_SELECT", "LSGPU_SEL_AMB_CAP", "LSGPU_FRONT_GUESS", "LSGPU_NO_ROUTE_ALL",
                                "LSGPU_NO_ROWQ", "LSGPU_ROWQ_BLOCKS", "LSGPU_SORT_ITEMS", "LSGPU_SSN_GLOBAL", "LSGPU_SSN_FULL_SORT", "LSGPU_SSN_OLD_FINISH", "LSGPU_SSN_ROOT", "LSGPU_SSN_SORT_LEVELS",
                                "LSGPU_NE_BLOCKS", "LSGPU_COMM_TIMEOUT_MS", "LSGPU_KNN_DBG", "LSGPU_NO_CONE", "LSGPU_NO_CONE_PROBE", "LSGPU_CONE_ROWS", "LSGPU_CONE_COLS", "LSGPU_CONE_FROM", "LSGPU_CONE_MAX_OCC", "LSGPU_CONE_HEAVY_STEPS", "LSGPU_CONE_HEAVY_SHARE",
                                // read by the Python / C++ hosts and the test drivers, not by this library:
                                "LSGPU_SO", "LSGPU_STATS_SO", "LSGPU_BATCH_POOLS", "LSGPU_BATCH_UNIQ", "LSGPU_GS_DEBUG", "LSGPU_GOLDEN_DIR", "LSGPU_SEQ_PERTURB", "LSGPU_SEQ_POSES", "LSGPU_TEST_INPUT_FILTERS",
#ifdef LSGPU_EXPERIMENTS
                                "LSGPU_KNN_ROWS", "LSGPU_KNN_LANE", "LSGPU_SPARSE_LANES", "LSGPU_TILE_WAVES", "LSGPU_XCD_SWIZZLE", "LSGPU_ROCPRIM_SORT",
#endif
                                nullptr};
#ifdef LSGPU_EXPERIMENTS
  t.knn_rows = (int)number("LSGPU_KNN_ROWS", 0, 0, 2);
  t.knn_lane = flag("LSGPU_KNN_LANE");
  t.sparse_lanes = (int)number("LSGPU_SPARSE_LANES", 0, 0, 64);
  t.tile_waves = (int)number("LSGPU_TILE_WAVES", 1, 1, 4) == 4 ? 4 : 1;
  t.xcd_swizzle = (int)number("LSGPU_XCD_SWIZZLE", 0, 0, 1 << 16);
  t.rocprim_sort = flag("LSGPU_ROCPRIM_SORT");
  // the front rows only exist in the one-wave tile kernel; the row-wise experiment does not fill the window table
  if (t.knn_lane || t.knn_rows != 0 || t.tile_waves == 4 || t.sparse_lanes != 0) t.front = false;
  if (t.knn_rows != 0) t.commit_select = false;
#endif
  for (char** e = environ; e && *e; ++e) {
    if (strncmp(*e, "LSGPU_", 6) != 0) continue;
    const char* eq = strchr(*e, '=');
    const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
    bool ok = false;
    for (const char** k = known; *k; ++k) ok = ok || (strlen(*k) == len && strncmp(*k, *e, len) == 0);
    if (!ok) fprintf(stderr, "liblsgpu_icp: unknown switch %.*s ignored (see csrc/lsgpu_tuning.h)\n", (int)len, *e);
  }
  return t;
}
}  // namespace tuning_detail

inline const Tuning& tuning() {
  static const Tuning t = tuning_detail::read();
  return t;
}

}  // namespace lsgpu
