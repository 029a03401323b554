// lsgpu_ssn_tree.hip.h -- the lower levels of SamplingSurfaceNormalDataPointsFilter's box tree inside ONE workgroup
// (laser_slam/configurations/icp_default.yaml:5-7; the filter PointMatcher::ICP::compute applies to the reference at
// laser_slam/src/laser_track.cpp:496).  Round 5; replaces k_ssn_finish (lsgpu_ssn.hip.h) as the default.
//
// What the levels compute (lsgpu_ssn.hip.h): every segment that still holds more than knn points is put into the STABLE
// order of its cut coordinate and halved; equal coordinates keep the order they had -- the rule the oracle and the host
// filter follow.  k_ssn_finish did that literally: per level four LDS radix passes over the 32-bit coordinate and one over
// the segment number, 40 passes and ~290 barriers for 2048 points, 226 us per workgroup (profiles/r04_bench.stats.txt).
//
// Here the three axes are sorted ONCE, when the workgroup starts (the classic presorted kd-tree build), and a level is
// two stable partitions instead of a sort:
//   list[d]   the root's points in the stable order of coordinate d (ties in the order the workgroup found them);
//             every segment owns the SAME index range [start, start + count) in all three lists
//   rank[d]   dense rank of a point's coordinate d among the root's points (equal coordinates <=> equal rank)
//   cur_pos   position of a point in the list that IS its segment's current order
// Level step for a segment that cuts along a:
//   1. list[a] restricted to the segment is the stable sort by coordinate a -- except inside runs of EQUAL coordinates,
//      whose members must follow the segment's current order (that is what "stable" means).  If the current order is
//      list[a] already (the parent cut along a as well) or the order the workgroup started from (ties in the lists are
//      in that order by construction) there is nothing to do; otherwise every member of a tie run counts the members of
//      its run that come before it in the current order (cur_pos) and moves there.  Runs are short on real clouds (a pair
//      now and then); a degenerate cloud costs O(run) per point, never a wrong order.
//   2. cur_pos <- positions in list[a]; the left child is the first count - count/2 of them.
//   3. the other two lists are partitioned stably by child (one packed block-wide scan for both), so that every child
//      again owns one index range in all three lists.
// devtools-free check of exactly this scheme against the chain of stable sorts, heavy ties included: the CPU suite's
// tests/test_oracle.py::test_presorted_lists_equal_the_chain_of_stable_sorts (numpy model of the steps above).
// The workgroup holds up to B = 8192 points (1024 threads, 152 KB of the CU's 160 KB LDS; 4096 / 2048 for the A/B
// switch), i.e. two more levels than k_ssn_finish's 2048 leave the global segmented sorts (~100 us each at 1 M points).
// Same boxes, same order, same normals as before: the bit-exact filter tests and the switch test (LSGPU_SSN_OLD_FINISH)
// compare the two.
#pragma once
#include "lsgpu_ssn.hip.h"

namespace lsgpu {

struct alignas(8) TreeSeg {   // 8 bytes, local to the workgroup
  uint16_t start, count;
  uint8_t cut;            // cut axis (meaningful while count > knn)
  uint8_t ord;            // axis whose list is this segment's current order; 0xFF: the order the workgroup started from
  uint16_t pad;
};
struct alignas(8) TreeBox { float lo[3], hi[3]; };

template <int B>
struct SsnTreeLds {
  static constexpr int T = B / 8;        // threads
  static constexpr int W = T / 64;       // waves
  static constexpr int S = B / 8;        // leaf segments at most (host: levels inside the workgroup <= log2(B / 8))
  static constexpr int SP = B / 16;      // parents of the last level at most
  uint16_t rank[3][B];
  uint16_t list[3][B];
  union {
    struct { uint32_t key[B]; uint32_t cnt[W][256]; } pre;                                        // presort of one axis
    struct { uint16_t cur_pos[B]; uint16_t sof[B]; TreeBox box[SP]; uint32_t segbase[S]; } tree;  // the levels
  } u;
  TreeSeg seg[S];
  uint32_t wsum[W];
  uint32_t wtot[4];
  uint32_t kmin[3], kmax[3];
};

#ifdef LSGPU_KNN_STATS   // stats build: shader-clock stamps of workgroup 0 (devtools/tree_phases.py)
__device__ unsigned long long g_tree_dbg[64];
#define LSGPU_TREE_T(n) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tree_dbg[n] = (unsigned long long)clock64(); } while (0)
#else
#define LSGPU_TREE_T(n) do { } while (0)
#endif

// (a barrier that waits for the LDS only -- s_waitcnt lgkmcnt(0) + s_barrier, so that the level's global gather of the cut
// value stays in flight across the partition -- was measured no faster: 1.010 against 0.994 ms per 1 M-point filter)
#define LSGPU_TREE_SYNC() __syncthreads()

// the value of the lane before / behind this one (DPP wave shift, no LDS traffic); lane 0 / 63 get 0
__device__ __forceinline__ uint32_t tree_lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ uint32_t tree_lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false); }
// inclusive prefix sum over the 64 lanes: DPP row scans + the three row totals (no LDS traffic)
__device__ __forceinline__ uint32_t tree_wave_scan(uint32_t v, int lane) {
  v = row_scan_incl_u32(v);
  const uint32_t t0 = rl_u(v, 15), t1 = rl_u(v, 31), t2 = rl_u(v, 47);
  return v + (lane >= 16 ? t0 : 0u) + (lane >= 32 ? t1 : 0u) + (lane >= 48 ? t2 : 0u);
}

template <int B>
__global__ __launch_bounds__(B / 8) void k_ssn_tree(const float4* __restrict__ p, uint32_t* __restrict__ idx,
                                                    const SsnSeg* __restrict__ segs, int knn, int rem,
                                                    uint32_t* __restrict__ seg_of, SsnSeg* __restrict__ segs_out,
                                                    const int* __restrict__ root_axis /* nullable */,
                                                    const uint32_t* __restrict__ root_sig /* nullable */) {
  using Lds = SsnTreeLds<B>;
  constexpr int T = Lds::T, W = Lds::W;
  __shared__ Lds L;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const SsnSeg root = segs[blockIdx.x];
  const int cnt = (int)root.count;
  const unsigned long long lt = (1ull << lane) - 1ull;
  // the axis the root's order already follows (the upper levels cut along it last): its list is the order the workgroup
  // finds the points in -- no sort (a third of the presort)
  const int sorted_axis = root_axis ? root_axis[blockIdx.x] : -1;

  // ---- load: ordered keys of this thread's 8 points, per axis.  Local id of a point = its position in the order the
  // workgroup found the root in (idx[root.start ..]); radix ownership: wave w, group it, lane -> w * 512 + it * 64 + lane
  LSGPU_TREE_T(0);
  uint32_t kx[8], ky[8], kz[8];
  if (tid < 3) { L.kmin[tid] = 0xFFFFFFFFu; L.kmax[tid] = 0u; }
  {
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = w * 512 + it * 64 + lane;
      kx[it] = ky[it] = kz[it] = 0u;
      if (e < cnt) {
        const float4 v = p[idx[root.start + e]];
        kx[it] = float_order_key(v.x); ky[it] = float_order_key(v.y); kz[it] = float_order_key(v.z);
        mn[0] = min(mn[0], kx[it]); mx[0] = max(mx[0], kx[it]);
        mn[1] = min(mn[1], ky[it]); mx[1] = max(mx[1], ky[it]);
        mn[2] = min(mn[2], kz[it]); mx[2] = max(mx[2], kz[it]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const uint32_t a = ~wave_max_u32(~mn[d]), b = wave_max_u32(mx[d]);
      if (lane == 0) { atomicMin(&L.kmin[d], a); atomicMax(&L.kmax[d], b); }
    }
    __syncthreads();
  }

  // ---- presort: per axis a stable LSD radix sort of the local ids by (key - min), 8 bits per pass, only the passes
  // the key range needs; the ids ping-pong between list[d] and rank[d], the keys stay where they are
  LSGPU_TREE_T(1);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const uint32_t kmin = L.kmin[d];
    const uint32_t range = cnt > 0 ? L.kmax[d] - kmin : 0u;
    const int bits = range ? 32 - __clz((int)range) : 0;
    const int P = d == sorted_axis ? 0 : (bits + 7) >> 3;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = w * 512 + it * 64 + lane;
      const uint32_t k = d == 0 ? kx[it] : d == 1 ? ky[it] : kz[it];
      if (e < cnt) L.u.pre.key[e] = k - kmin;
    }
    __syncthreads();
    for (int j = 0; j < P; ++j) {
      uint16_t* dst = ((P - 1 - j) & 1) ? L.rank[d] : L.list[d];
      const uint16_t* src = ((P - 1 - j) & 1) ? L.list[d] : L.rank[d];   // (not read in pass 0)
      const int shift = 8 * j;
#pragma unroll
      for (int q = 0; q < 4; ++q) (&L.u.pre.cnt[0][0])[tid + q * T] = 0u;
      __syncthreads();
      uint32_t es[8], dg[8], rk[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int i = w * 512 + it * 64 + lane;
        const bool valid = i < cnt;
        const uint32_t e = valid ? (j == 0 ? (uint32_t)i : (uint32_t)src[i]) : 0u;
        const uint32_t dgt = valid ? (L.u.pre.key[e] >> shift) & 255u : 0u;
        es[it] = e; dg[it] = dgt;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const bool bit = (dgt >> b) & 1u;
          const unsigned long long m = __ballot(bit);
          peers &= bit ? m : ~m;
        }
        const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
        uint32_t old = 0u;
        if (valid && lane == leader) {
          old = L.u.pre.cnt[w][dgt];
          L.u.pre.cnt[w][dgt] = old + (uint32_t)__popcll(peers);
        }
        old = (uint32_t)__shfl((int)old, leader, 64);
        rk[it] = old + (uint32_t)__popcll(peers & lt);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
      }
      __syncthreads();
      uint32_t tot = 0u, incl = 0u;
      if (tid < 256) {   // thread = digit: its total over the waves, then the exclusive scan over the digits
#pragma unroll
        for (int ww = 0; ww < W; ++ww) tot += L.u.pre.cnt[ww][tid];
        incl = wave_scan_incl_u32(tot, lane);
        if (lane == 63) L.wtot[w] = incl;
      }
      __syncthreads();
      if (tid < 256) {
        uint32_t run = incl - tot;
        for (int ww = 0; ww < w; ++ww) run += L.wtot[ww];
#pragma unroll
        for (int ww = 0; ww < W; ++ww) {
          const uint32_t c = L.u.pre.cnt[ww][tid];
          L.u.pre.cnt[ww][tid] = run;
          run += c;
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int i = w * 512 + it * 64 + lane;
        if (i < cnt) dst[L.u.pre.cnt[w][dg[it]] + rk[it]] = (uint16_t)es[it];
      }
      __syncthreads();
    }
    if (P == 0) {   // sorted already, or one value on this axis: the order the workgroup started from
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int i = w * 512 + it * 64 + lane;
        if (i < cnt) L.list[d][i] = (uint16_t)i;
      }
      __syncthreads();
    }
    LSGPU_TREE_T(2 + 2 * d);
#ifdef LSGPU_KNN_STATS
    if (blockIdx.x == 0 && tid == 0) g_tree_dbg[40 + d] = (unsigned long long)P;
#endif
    // dense ranks: a wave walks its 512 consecutive positions of the sorted list, 64 at a time
    {
      uint32_t e[8], r[8];
      uint32_t carry = 0u;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = w * 512 + k * 64 + lane;
        e[k] = 0u;
        uint32_t kk = 0u;
        if (i < cnt) { e[k] = L.list[d][i]; kk = L.u.pre.key[e[k]]; }
        uint32_t pk = tree_lane_prev(kk);
        if (lane == 0 && i > 0 && i < cnt) pk = L.u.pre.key[L.list[d][i - 1]];
        const uint32_t f = (i < cnt && i > 0 && kk != pk) ? 1u : 0u;
        const uint32_t incl = tree_wave_scan(f, lane);
        r[k] = carry + incl;
        carry += rl_u(incl, 63);
      }
      if (lane == 0) L.wsum[w] = carry;
      __syncthreads();
      uint32_t before = 0u;
      for (int ww = 0; ww < w; ++ww) before += L.wsum[ww];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (w * 512 + k * 64 + lane < cnt) L.rank[d][e[k]] = (uint16_t)(before + r[k]);
      __syncthreads();
    }
    LSGPU_TREE_T(3 + 2 * d);
  }

  // ---- the levels.  Ownership as in the radix passes: wave w, slab k, lane -> position w * 512 + k * 64 + lane, so that
  // the 64 lanes of every LDS access by position touch consecutive uint16 (the first version gave a thread 8 consecutive
  // positions: 16-byte lane stride, 4-way bank conflicts on every access, 27 k cycles per level)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = w * 512 + k * 64 + lane;
    L.u.tree.cur_pos[i] = (uint16_t)i;
    L.u.tree.sof[i] = 0;
  }
  if (tid == 0) {
    TreeSeg r0;
    r0.start = 0; r0.count = (uint16_t)cnt; r0.cut = (uint8_t)ssn_cut_axis(root); r0.ord = 0xFF; r0.pad = 0;
    L.seg[0] = r0;
    TreeBox b0;
#pragma unroll
    for (int d = 0; d < 3; ++d) { b0.lo[d] = root.lo[d]; b0.hi[d] = root.hi[d]; }
    L.u.tree.box[0] = b0;
  }
  __syncthreads();
  // ---- the initial order.  Roots that come from the sort-free upper levels (lsgpu_ssn_select.hip.h) arrive as SETS in
  // original-index order with a signature: the axes they were cut along, most recent first.  Their current order -- what
  // the chain of stable sorts would have left -- is (key on the signature's first axis, on its second, on its third, index):
  // the list of the first axis is that order up to its tie runs, which are put right here by the same comparator (dense
  // ranks stand in for the keys); cur_pos is taken from it and the levels below carry on as for any other segment.
  const uint32_t rsig = root_sig ? root_sig[blockIdx.x] : 0xFFFFFFFFu;
  const uint32_t ro1 = rsig & 0xFFu, ro2 = (rsig >> 8) & 0xFFu, ro3 = (rsig >> 16) & 0xFFu;
  if (ro1 < 3u) {     // (the same for every thread of the workgroup)
    uint16_t* la = L.list[ro1];
    const uint16_t* ra = L.rank[ro1];
    const uint16_t* r2 = L.rank[ro2 < 3u ? ro2 : 0u];
    const uint16_t* r3 = L.rank[ro3 < 3u ? ro3 : 0u];
    uint32_t ek[8], np[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = w * 512 + k * 64 + lane;
      ek[k] = 0u; np[k] = (uint32_t)i;
      const bool act = i < cnt;
      if (act) ek[k] = la[i];
      const uint32_t r = act ? (uint32_t)ra[ek[k]] : 0xFFFFFFFFu;
      uint32_t rp = tree_lane_prev(r), rn = tree_lane_next(r);
      if (lane == 0 && act && i > 0) rp = ra[la[i - 1]];
      if (lane == 63 && act && i + 1 < cnt) rn = ra[la[i + 1]];
      if (act && ((i > 0 && rp == r) || (i + 1 < cnt && rn == r))) {
        int lo = i, hi = i + 1;
        while (lo > 0 && ra[la[lo - 1]] == r) --lo;
        while (hi < cnt && ra[la[hi]] == r) ++hi;
        const uint32_t e = ek[k];
        const uint32_t e2 = ro2 < 3u ? (uint32_t)r2[e] : 0u, e3 = ro3 < 3u ? (uint32_t)r3[e] : 0u;
        uint32_t c = 0u;
        for (int j = lo; j < hi; ++j) {
          const uint32_t f = la[j];
          const uint32_t f2 = ro2 < 3u ? (uint32_t)r2[f] : 0u, f3 = ro3 < 3u ? (uint32_t)r3[f] : 0u;
          const bool less = f2 != e2 ? f2 < e2 : f3 != e3 ? f3 < e3 : f < e;
          c += less ? 1u : 0u;
        }
        np[k] = (uint32_t)lo + c;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = w * 512 + k * 64 + lane;
      if (i < cnt) { la[np[k]] = (uint16_t)ek[k]; L.u.tree.cur_pos[ek[k]] = (uint16_t)np[k]; }
    }
    if (tid == 0) L.seg[0].ord = (uint8_t)ro1;
    __syncthreads();
  }
  const uint32_t base_seg = (uint32_t)blockIdx.x << rem;
  for (int l = 0; l < rem; ++l) {
    const int ns = 1 << l;
    const bool last = l + 1 == rem;
    LSGPU_TREE_T(8 + l);
    // this thread's 8 positions: their segments
    uint32_t sk[8];
    TreeSeg sg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = w * 512 + k * 64 + lane;
      sk[k] = L.u.tree.sof[i];
      sg[k] = L.seg[i < cnt ? sk[k] : 0u];
    }
    // step 1: tie runs of list[cut] into the segment's current order
    uint32_t ek[8], np[8];
    bool upd[8], fix[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = w * 512 + k * 64 + lane;
      const bool active = i < cnt && (int)sg[k].count > knn;
      const int a = sg[k].cut;
      upd[k] = active && sg[k].ord != a;
      fix[k] = upd[k] && sg[k].ord != 0xFF;
      ek[k] = 0u; np[k] = (uint32_t)i;
      const uint16_t* la = L.list[a];
      const uint16_t* ra = L.rank[a];
      if (upd[k]) ek[k] = la[i];
      if (__ballot(fix[k])) {   // (wave-uniform: no segment of this slab re-sorts, nothing to look at)
        const uint32_t r = fix[k] ? (uint32_t)ra[ek[k]] : 0xFFFFFFFFu;
        // the neighbours' ranks: lanes of the same segment hold them (same segment -> same cut axis, same `fix`); the
        // slab's first and last lane read theirs
        const int s0 = sg[k].start, s1 = s0 + sg[k].count;
        uint32_t rp = tree_lane_prev(r), rn = tree_lane_next(r);
        if (lane == 0 && fix[k] && i > s0) rp = ra[la[i - 1]];
        if (lane == 63 && fix[k] && i + 1 < s1) rn = ra[la[i + 1]];
        const bool tie = fix[k] && ((i > s0 && rp == r) || (i + 1 < s1 && rn == r));
        if (tie) {
          int lo = i, hi = i + 1;
          while (lo > s0 && ra[la[lo - 1]] == r) --lo;
          while (hi < s1 && ra[la[hi]] == r) ++hi;
          const uint32_t cp = L.u.tree.cur_pos[ek[k]];
          uint32_t c = 0u;
          for (int j = lo; j < hi; ++j) c += L.u.tree.cur_pos[la[j]] < cp ? 1u : 0u;
          np[k] = (uint32_t)lo + c;
        }
      }
    }
    LSGPU_TREE_SYNC();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (fix[k]) L.list[sg[k].cut][np[k]] = (uint16_t)ek[k];
      if (upd[k]) L.u.tree.cur_pos[ek[k]] = (uint16_t)np[k];
    }
    LSGPU_TREE_SYNC();
    // the segments' own threads: the point the cut value comes from (a global gather, in flight across the partition)
    TreeSeg ps;
    TreeBox pb;
    uint32_t gi = 0u;
    ps.start = 0; ps.count = 0; ps.cut = 0; ps.ord = 0xFF; ps.pad = 0;
    if (tid < ns) {
      ps = L.seg[tid];
      pb = L.u.tree.box[tid];
      if ((int)ps.count > knn) {
        const uint32_t left = (uint32_t)ps.count - (uint32_t)ps.count / 2u;
        gi = idx[root.start + L.list[ps.cut][ps.start + left]];
      }
    }
    // step 3: stable partition of the other two lists by child; one packed scan (low half: axis cut + 1, high: cut + 2)
    uint32_t e1[8], e2[8], xk[8], vk[8];
    uint32_t carry = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = w * 512 + k * 64 + lane;
      const bool active = i < cnt && (int)sg[k].count > knn;
      uint32_t v = 0u;
      e1[k] = e2[k] = 0u;
      if (active) {
        const int a = sg[k].cut;
        const int d1 = a == 2 ? 0 : a + 1, d2 = a == 0 ? 2 : a - 1;
        const uint32_t left = (uint32_t)sg[k].count - (uint32_t)sg[k].count / 2u;
        e1[k] = L.list[d1][i]; e2[k] = L.list[d2][i];
        const uint32_t f1 = ((uint32_t)L.u.tree.cur_pos[e1[k]] - sg[k].start) >= left ? 1u : 0u;
        const uint32_t f2 = ((uint32_t)L.u.tree.cur_pos[e2[k]] - sg[k].start) >= left ? 1u : 0u;
        v = f1 | (f2 << 16);
      }
      vk[k] = v;
      const uint32_t incl = tree_wave_scan(v, lane);
      xk[k] = carry + incl - v;      // exclusive, inside the wave's 512 positions
      carry += rl_u(incl, 63);
    }
    if (lane == 0) L.wsum[w] = carry;
    LSGPU_TREE_SYNC();
    uint32_t before = 0u;
    for (int ww = 0; ww < w; ++ww) before += L.wsum[ww];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      xk[k] += before;
      const int i = w * 512 + k * 64 + lane;
      if (i < cnt && (int)sg[k].count > knn && i == (int)sg[k].start) L.u.tree.segbase[sk[k]] = xk[k];
    }
    LSGPU_TREE_SYNC();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = w * 512 + k * 64 + lane;
      uint32_t nsof = 2u * sk[k];
      if (i < cnt && (int)sg[k].count > knn) {
        const int a = sg[k].cut;
        const int d1 = a == 2 ? 0 : a + 1, d2 = a == 0 ? 2 : a - 1;
        const uint32_t left = (uint32_t)sg[k].count - (uint32_t)sg[k].count / 2u;
        const uint32_t rel = xk[k] - L.u.tree.segbase[sk[k]];   // (no borrow between the halves: both prefixes are monotone)
        const uint32_t r1 = rel & 0xFFFFu, r2 = rel >> 16;
        const uint32_t off = (uint32_t)i - sg[k].start;
        const uint32_t f1 = vk[k] & 1u, f2 = vk[k] >> 16;
        const uint32_t p1 = f1 ? sg[k].start + left + r1 : sg[k].start + (off - r1);
        const uint32_t p2 = f2 ? sg[k].start + left + r2 : sg[k].start + (off - r2);
        L.list[d1][p1] = (uint16_t)e1[k];
        L.list[d2][p2] = (uint16_t)e2[k];
        nsof += off >= left ? 1u : 0u;
      }
      L.u.tree.sof[i] = (uint16_t)nsof;
    }
    // the children (2s, 2s + 1); a finished segment carries over as child 2s
    if (tid < ns) {
      TreeSeg ca = ps, cb = ps;
      TreeBox ba = pb, bb = pb;
      if ((int)ps.count > knn) {
        const int cut = ps.cut;
        const uint32_t right = (uint32_t)ps.count / 2u, left = (uint32_t)ps.count - right;
        const float cutval = coord_of(p[gi], cut);
        SsnSeg ta, tb;
#pragma unroll
        for (int d = 0; d < 3; ++d) {   // (selects, not ba.hi[cut]: a dynamically indexed member would live in scratch memory)
          ba.hi[d] = d == cut ? cutval : ba.hi[d];
          bb.lo[d] = d == cut ? cutval : bb.lo[d];
          ta.lo[d] = ba.lo[d]; ta.hi[d] = ba.hi[d]; tb.lo[d] = bb.lo[d]; tb.hi[d] = bb.hi[d];
        }
        ca.count = (uint16_t)left; ca.cut = (uint8_t)ssn_cut_axis(ta); ca.ord = (uint8_t)cut;
        cb.start = (uint16_t)(ps.start + left); cb.count = (uint16_t)right; cb.cut = (uint8_t)ssn_cut_axis(tb); cb.ord = (uint8_t)cut;
      } else {
        cb.start = (uint16_t)(ps.start + ps.count); cb.count = 0;
      }
      L.seg[2 * tid] = ca;
      L.seg[2 * tid + 1] = cb;
      if (!last) {
        L.u.tree.box[2 * tid] = ba;
        L.u.tree.box[2 * tid + 1] = bb;
      } else {
        SsnSeg oa, ob;
        oa.start = root.start + ca.start; oa.count = ca.count;
        ob.start = root.start + cb.start; ob.count = cb.count;
#pragma unroll
        for (int d = 0; d < 3; ++d) { oa.lo[d] = ba.lo[d]; oa.hi[d] = ba.hi[d]; ob.lo[d] = bb.lo[d]; ob.hi[d] = bb.hi[d]; }
        segs_out[base_seg + 2u * (uint32_t)tid] = oa;
        segs_out[base_seg + 2u * (uint32_t)tid + 1u] = ob;
      }
    }
    LSGPU_TREE_SYNC();
  }

  // ---- out: every leaf in its current order; the workgroup's range of idx is read completely before it is written
  LSGPU_TREE_T(30);
  uint32_t g[8], so[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = w * 512 + k * 64 + lane;
    g[k] = 0u; so[k] = 0u;
    if (i < cnt) {
      so[k] = L.u.tree.sof[i];
      const TreeSeg leaf = L.seg[so[k]];
      const uint32_t e = leaf.ord == 0xFF ? (uint32_t)i : (uint32_t)L.list[leaf.ord][i];
      g[k] = idx[root.start + e];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = w * 512 + k * 64 + lane;
    if (i < cnt) {
      idx[root.start + i] = g[k];
      seg_of[root.start + i] = base_seg + so[k];
    }
  }
  LSGPU_TREE_T(31);
}

}  // namespace lsgpu
