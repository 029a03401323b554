// lsgpu_ssn_select.hip.h -- the UPPER levels of SamplingSurfaceNormalDataPointsFilter's box tree (segments too large for
// one workgroup of k_ssn_tree) WITHOUT any sort (icp_default.yaml:5-7; the reference filter of PointMatcher::ICP::compute,
// laser_slam/src/laser_track.cpp:496).  Round 5.
//
// Rounds 1-4 sorted at every level (lsgpu_segsort.hip.h: a segmented stable radix sort of the cut coordinate, 4 passes of 3
// launches + key / plan / split / assign kernels = 16 dependent launches and ~100 us per level at 1 M points, whatever the
// amount of data).  What a level actually needs is much less:
//   * a segment is split at the MEDIAN of its points in the stable order of the cut coordinate;
//   * the order INSIDE the halves only matters as the tie-break of later levels -- and the chain of stable sorts has a
//     closed form: a segment's current order is the lexicographic order of (key on the axis it was cut along last, key on
//     the one before, key on the third, original index): its SIGNATURE, three bytes.
// So the upper levels keep no order at all.  A segment is a SET of points -- held in original-index order by stable
// partitions -- with a signature, and a level is
//   k_gs_hist<1>               256-bin histogram of the cut coordinate over the range the segment's POINTS span (the box can be
//   k_gs_hist<2>               far wider: a patch of ground keeps the cloud's z range until z is cut), summed per segment with
//                              one atomic per block and bin; the bin the median falls into is found by every block of the NEXT
//                              kernel in its prologue (a 256-entry scan: cheaper than a launch); the same again inside that
//                              bin (16 bits of the range resolved: a handful of candidates per segment remain);
//   k_gs_collect               the candidates of the last bin -> a small per-segment list; every block's count of points that
//                              go left for sure;
//   k_gs_select                ONE workgroup per segment: the exact median of the candidates in the total order above (key on
//                              the cut axis, keys on the signature's other axes, original index) -- ranks by ballots up to 256
//                              candidates, a radix select over the tuple's bytes beyond --, the children's boxes and
//                              signatures, the candidates' share of the blocks' left counts and their exclusive prefix;
//   k_gs_part                  the stable partition by "before the median in that order": points and their three keys move
//                              to the other buffer set, left half first; on the way the key range of either child on ITS cut
//                              axis (four atomics per block).
// Five launches per level, every access by position coalesced, no key ever sorted.  Segment sizes are static (exact
// halving): the host knows every level's number of blocks, k_gs_plan writes the block tables when the cloud's size changes.
// k_ssn_tree takes the sets over: it presorts its three axes anyway; the list of the signature's first axis gets its tie
// runs ordered by the same comparator once (lsgpu_ssn_tree.hip.h, "initial order"), after which the workgroup's own
// bookkeeping (cur_pos) carries on.  The scheme is modelled step for step in tests/ssn_tree_model.py (select_then_tree) and
// checked there against the chain of stable sorts the restatement defines, heavy ties included.
// Limits: the candidate list has room for kGsCandRoom candidates per segment of the LAST level -- as many as such a segment
// has points -- and the levels above share the same total, so a segment's candidates always fit (a wall square to a frame
// axis puts 12 000 points of a sub-map into one bin of the first level; a lattice a fifth of a segment).  What cannot
// happen -- more candidates than room, a key outside its segment's range, counts that do not add up -- raises a flag and the
// host repeats the filter with the segmented sorts (LSGPU_SSN_SORT_LEVELS selects them always; LSGPU_GS_DEBUG prints the
// reason).  Bit-identical to them and to the oracle.
#pragma once
#include "lsgpu_ssn.hip.h"
#include "lsgpu_ssn_tree.hip.h"

namespace lsgpu {

// (stats build: stamps of the first level's k_gs_select, g_tree_dbg[48..55]; devtools/tree_phases.py)
#ifdef LSGPU_KNN_STATS
#define LSGPU_SEL_T(n) do { if (gridDim.x == 1 && threadIdx.x == 0) g_tree_dbg[n] = (unsigned long long)clock64(); } while (0)
#else
#define LSGPU_SEL_T(n) do { } while (0)
#endif

constexpr uint32_t kGsTile = 2048u;       // positions per block of the level kernels (256 threads x 8)
constexpr int kGsPartThreads = 512;        // k_gs_part: 4 positions per thread (512 blocks of 256 threads left a 1 M-point level two waves per SIMD)
constexpr uint32_t kGsCandCap = 2048u;    // candidates k_gs_select holds in LDS (more: it works on the global list)
constexpr uint32_t kGsCandRoom = 16384u;  // room in the candidate list per segment of the LAST level (>= its points: it cannot
                                          // overflow; the levels above share the same total, i.e. their segments' sizes too)
constexpr uint32_t kGsNoAxis = 0xFFu;
constexpr uint32_t kGsSelBlocks = 8192u;  // k_gs_select counts in LDS for segments of up to this many blocks (16.7 M points)

struct GsSet {                 // one buffer set: the points of every segment (original-index order inside it) and their keys
  uint32_t* e;
  uint32_t* k[3];
};
// (selects, not g.k[d]: a dynamically indexed kernel argument would be copied to scratch memory)
__device__ __forceinline__ uint32_t* gs_k(const GsSet& g, int d) { return d == 0 ? g.k[0] : d == 1 ? g.k[1] : g.k[2]; }

struct GsBlock {               // 32 bytes, computed on the host (segment sizes are static)
  uint32_t first, count;       // positions [first, first + count)
  uint32_t seg;                // its segment ...
  uint32_t seg_start, seg_count;
  uint32_t fb, nb;             // ... and that segment's blocks [fb, fb + nb)
  uint32_t pad;
};
struct GsSegBlocks { uint32_t fb, nb, start, count; };   // per segment (host)

struct GsMedian { uint32_t ka, k1, k2, e; };   // the median in the segment's total order: cut-axis key, the other keys, index

// a segment's signature: the axes it was cut along, most recent first, one byte each (kGsNoAxis: none); 0xFFFFFFFF at the root
__host__ __device__ __forceinline__ uint32_t gs_sig_push(uint32_t sig, uint32_t a) {
  const uint32_t o1 = sig & 0xFFu, o2 = (sig >> 8) & 0xFFu, o3 = (sig >> 16) & 0xFFu;
  if (o1 == a) return sig;
  const uint32_t r2 = o2 == a ? o3 : o2;               // the others, a removed (three axes: at most two remain)
  return a | (o1 << 8) | (r2 << 16) | 0xFF000000u;
}
// the two axes that follow the cut axis in the segment's order (kGsNoAxis: none)
__device__ __forceinline__ void gs_other_axes(uint32_t sig, uint32_t a, uint32_t& x1, uint32_t& x2) {
  const uint32_t o1 = sig & 0xFFu, o2 = (sig >> 8) & 0xFFu, o3 = (sig >> 16) & 0xFFu;
  if (o1 == a) { x1 = o2; x2 = o3; }
  else { x1 = o1; x2 = o2 == a ? o3 : o2; }
}
__device__ __forceinline__ bool gs_less(uint32_t ka, uint32_t k1, uint32_t k2, uint32_t e, const GsMedian& m) {
  return ka != m.ka ? ka < m.ka : k1 != m.k1 ? k1 < m.k1 : k2 != m.k2 ? k2 < m.k2 : e < m.e;
}

// The block tables of all levels, on the device (one workgroup per level).  Segment sizes are static -- a segment of c
// points splits into c - c / 2 and c / 2 -- so the host knows every level's number of blocks (a few hundred integer
// operations) and where its rows start; writing the 14 000 rows of a 3.1 M-point cloud on the host and copying them was
// 0.10 - 0.12 ms per call with the device idle, and a track's sub-map has another size at every scan.
constexpr int kGsPlanLevels = 24;
constexpr uint32_t kGsPlanSegs = 8192u;    // segments per level this kernel handles (LDS); more: host tables
struct GsPlanArgs { uint32_t first[kGsPlanLevels]; };
__global__ __launch_bounds__(1024) void k_gs_plan(uint32_t n, GsPlanArgs lvl, GsBlock* __restrict__ tab, GsSegBlocks* __restrict__ sblk) {
  __shared__ uint32_t fbs[kGsPlanSegs + 1];     // first block of every segment of this level (+ the total)
  __shared__ uint2 sc[kGsPlanSegs];             // (start, count)
  __shared__ uint32_t wsum[16], carry_sh;
  const uint32_t L = blockIdx.x, ns = 1u << L;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_sh = 0u;
  __syncthreads();
  for (uint32_t s0 = 0; s0 < ns; s0 += 1024u) {
    const uint32_t sg = s0 + threadIdx.x;
    uint32_t st = 0u, c = n;
    for (int bit = (int)L - 1; bit >= 0; --bit) {   // the path from the root: left child first
      const uint32_t left = c - c / 2u;
      if ((sg >> bit) & 1u) { st += left; c -= left; } else c = left;
    }
    const uint32_t nb = sg < ns ? (c + kGsTile - 1u) / kGsTile : 0u;
    const uint32_t incl = wave_scan_incl_u32(nb, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = carry_sh;
    for (int ww = 0; ww < w; ++ww) before += wsum[ww];
    if (sg < ns) { fbs[sg] = before + incl - nb; sc[sg] = make_uint2(st, c); }
    __syncthreads();
    if (threadIdx.x == 1023) carry_sh = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) fbs[ns] = carry_sh;
  __syncthreads();
  const uint32_t total = fbs[ns];
  for (uint32_t sg = threadIdx.x; sg < ns; sg += 1024u)
    sblk[(ns - 1u) + sg] = GsSegBlocks{fbs[sg], fbs[sg + 1u] - fbs[sg], sc[sg].x, sc[sg].y};
  GsBlock* __restrict__ t = tab + lvl.first[L];
  for (uint32_t j = threadIdx.x; j < total; j += 1024u) {
    uint32_t lo = 0u, hi = ns;                    // the segment whose blocks hold row j: fbs[lo] <= j < fbs[lo + 1]
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (fbs[mid] <= j) lo = mid; else hi = mid; }
    const uint32_t k = j - fbs[lo], st = sc[lo].x, c = sc[lo].y;
    const uint32_t off = k * kGsTile;
    t[j] = GsBlock{st + off, min(kGsTile, c - off), lo, st, c, fbs[lo], fbs[lo + 1u] - fbs[lo], 0u};
  }
}

// keys and identity: the root set (the other set's point array gets valid ids too: whatever an aborted level leaves behind
// must stay dereferenceable for the kernels queued behind it); the root's key range on its cut axis = the cloud's bounds;
// block 0 also clears what the first level accumulates into (its two histograms, its candidate count, the error words)
// and gives the root its empty signature -- five fills in front of the first level otherwise
__global__ __launch_bounds__(256) void k_gs_init(const float4* __restrict__ p, int n, GsSet out, uint32_t* __restrict__ other_e,
                                                 const SsnSeg* __restrict__ root, const uint32_t* __restrict__ bb,
                                                 uint2* __restrict__ rng, uint32_t* __restrict__ gh1_root,
                                                 uint32_t* __restrict__ gh2_root, uint32_t* __restrict__ cand_n_root,
                                                 uint32_t* __restrict__ sig_root, uint32_t* __restrict__ err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0) {
    gh1_root[threadIdx.x] = 0u; gh2_root[threadIdx.x] = 0u;
    if (threadIdx.x < 8) err[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
      const int a = ssn_cut_axis(root[0]); rng[0] = make_uint2(bb[a], bb[3 + a]);
      cand_n_root[0] = 0u; sig_root[0] = 0xFFFFFFFFu;   // the root was cut along no axis yet
    }
  }
  if (i >= n) return;
  const float4 v = p[i];
  out.e[i] = (uint32_t)i; other_e[i] = (uint32_t)i;
  out.k[0][i] = float_order_key(v.x); out.k[1][i] = float_order_key(v.y); out.k[2][i] = float_order_key(v.z);
}

// What a block needs to know about its segment's cut: the axis and the map from a key to one of 65536 bins.  The bins are
// uniform in the COORDINATE over the range the segment's points span, not in the key: keys are float bit patterns, a range
// that crosses zero or a few binades spends most of its key space where no point is (measured: 3 241 of a segment's 16 349
// points -- ground returns within two centimetres, the range stretched to 7.7 m by a wall -- in one of 65536 key-space
// bins; 50 in a coordinate-space bin).  All that exactness needs of the map is that it is monotone: x1 < x2 => bin(x1) <=
// bin(x2), which float subtraction, multiplication by a positive constant and truncation are.
struct GsCut { int a; uint32_t klo, khi; float xlo, scale; };
__device__ __forceinline__ GsCut gs_cut(const SsnSeg& sg, const uint2 rng /* keys of the segment's points on the cut axis: min, max */) {
  GsCut c;
  c.a = ssn_cut_axis(sg);
  c.klo = rng.x; c.khi = rng.y >= rng.x ? rng.y : rng.x;
  c.xlo = float_from_order_key(c.klo);
  const float ext = float_from_order_key(c.khi) - c.xlo;
  c.scale = ext > 0.f ? 65536.0f / ext : 0.f;
  return c;
}
__device__ __forceinline__ uint32_t gs_bin16(const GsCut& c, uint32_t k) {
  const float t = (float_from_order_key(k) - c.xlo) * c.scale;
  return (uint32_t)fminf(fmaxf(t, 0.f), 65535.f);
}

// The bin of a 256-bin segment histogram that holds rank `target`, and the number of points in front of it: every block
// that needs it computes it itself (whole workgroup of 256 threads; LDS scratch of 8 words; ~1 us).  bin = 0xFFFFFFFF if the
// rank is not in the histogram (cannot happen unless an earlier kernel failed).
__device__ __forceinline__ void gs_find_bin(const uint32_t* __restrict__ gh, uint32_t target, uint32_t* sh /* >= 8 */,
                                            uint32_t& bin, uint32_t& below) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool act = threadIdx.x < 256u;           // (workgroups of more than 256 threads: the first four waves do it)
  const uint32_t tot = act ? gh[threadIdx.x] : 0u;
  const uint32_t incl = wave_scan_incl_u32(tot, lane);
  if (threadIdx.x == 0) { sh[4] = 0xFFFFFFFFu; sh[5] = 0u; }
  if (act && lane == 63) sh[w] = incl;
  __syncthreads();
  uint32_t before = 0u;
  for (int ww = 0; ww < w && ww < 4; ++ww) before += sh[ww];
  const uint32_t excl = before + incl - tot;
  if (act && tot != 0u && excl <= target && target < excl + tot) { sh[4] = threadIdx.x; sh[5] = excl; }
  __syncthreads();
  bin = sh[4]; below = sh[5];
  __syncthreads();
}

// PASS 1: the upper eight bits of every point's bin; PASS 2: the lower eight of the points inside the median's first bin.
// gh1 / gh2: per segment 256 counters (zeroed by the level above), one atomic per block and occupied bin.
template <int PASS>
__global__ __launch_bounds__(256) void k_gs_hist(const GsBlock* __restrict__ tab, const SsnSeg* __restrict__ segs,
                                                 const uint2* __restrict__ rng, GsSet in, uint32_t* __restrict__ gh1,
                                                 uint32_t* __restrict__ gh2, uint32_t* __restrict__ err) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh[8];
  hist[threadIdx.x] = 0u;
  const GsBlock sb = tab[blockIdx.x];
  const GsCut c = gs_cut(segs[sb.seg], rng[sb.seg]);
  const uint32_t* __restrict__ ka = gs_k(in, c.a);
  uint32_t mb1 = 0u, below1 = 0u;
  if (PASS == 2) gs_find_bin(gh1 + (size_t)sb.seg * 256u, sb.seg_count - sb.seg_count / 2u, sh, mb1, below1);
  else __syncthreads();
  bool bad = PASS == 2 && mb1 == 0xFFFFFFFFu;
  for (uint32_t j = threadIdx.x; j < sb.count; j += 256u) {
    const uint32_t k = ka[sb.first + j];
    bad = bad || k < c.klo || k > c.khi;
    const uint32_t b16 = gs_bin16(c, k), b1 = b16 >> 8;
    if (PASS == 1) atomicAdd(&hist[b1], 1u);
    else if (b1 == mb1) atomicAdd(&hist[b16 & 255u], 1u);
  }
  if (bad) { err[0] = 1u; err[1] = 1u; err[2] = sb.seg; }   // a point outside its segment's range / no bin: cannot happen; the host would fall back
  __syncthreads();
  const uint32_t v = hist[threadIdx.x];
  if (v) atomicAdd(&(PASS == 1 ? gh1 : gh2)[(size_t)sb.seg * 256u + threadIdx.x], v);
}

// candidates of the median's last bin -> the segment's list (tuple + block); the block's count of points that go left for sure
__global__ __launch_bounds__(256) void k_gs_collect(const GsBlock* __restrict__ tab, const SsnSeg* __restrict__ segs,
                                                    const uint2* __restrict__ rng,
                                                    const uint32_t* __restrict__ sig, GsSet in, const uint32_t* __restrict__ gh1,
                                                    const uint32_t* __restrict__ gh2,
                                                    uint32_t* __restrict__ cand_n, GsMedian* __restrict__ cand,
                                                    uint32_t* __restrict__ cand_blk, uint32_t cap /* candidates per segment at this level */,
                                                    uint32_t* __restrict__ cl) {
  __shared__ uint32_t ws[4];
  __shared__ uint32_t sh[8];
  const GsBlock sb = tab[blockIdx.x];
  const GsCut c = gs_cut(segs[sb.seg], rng[sb.seg]);
  const uint32_t leftn = sb.seg_count - sb.seg_count / 2u;
  uint32_t mb1, below1, mb2, below2;
  gs_find_bin(gh1 + (size_t)sb.seg * 256u, leftn, sh, mb1, below1);
  gs_find_bin(gh2 + (size_t)sb.seg * 256u, leftn - below1, sh, mb2, below2);
  uint32_t x1, x2;
  gs_other_axes(sig[sb.seg], (uint32_t)c.a, x1, x2);
  const uint32_t* __restrict__ ka = gs_k(in, c.a);
  uint32_t left = 0u;
  // (uniform trip count: the waves allocate their candidates' slots together -- ONE atomic on the segment's counter per
  //  wave and round instead of one per candidate: a wall square to the cut axis is twelve thousand candidates in one
  //  segment, and same-address atomics are served one after the other, ~12 ns each)
  const int lane = threadIdx.x & 63;
  for (uint32_t j0 = 0; j0 < sb.count; j0 += 256u) {
    const uint32_t j = j0 + threadIdx.x;
    const bool live = j < sb.count;
    const uint32_t i = sb.first + (live ? j : 0u);
    const uint32_t k = live ? ka[i] : 0u;
    const uint32_t b16 = gs_bin16(c, k), b1 = b16 >> 8, b2 = b16 & 255u;
    const bool is_cand = live && b1 == mb1 && b2 == mb2;
    if (live && (b1 < mb1 || (b1 == mb1 && b2 < mb2))) ++left;
    const unsigned long long m = __ballot(is_cand);
    if (m != 0ull) {
      uint32_t base = 0u;
      if (lane == 0) base = atomicAdd(&cand_n[sb.seg], (uint32_t)__popcll(m));
      base = (uint32_t)__shfl((int)base, 0);
      const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (is_cand && slot < cap) {
        GsMedian md;
        md.ka = k; md.e = in.e[i];
        md.k1 = x1 != kGsNoAxis ? gs_k(in, (int)x1)[i] : 0u;
        md.k2 = x2 != kGsNoAxis ? gs_k(in, (int)x2)[i] : 0u;
        cand[(size_t)sb.seg * cap + slot] = md;
        cand_blk[(size_t)sb.seg * cap + slot] = blockIdx.x;
      }
    }
  }
  left = wave_sum_u32(left);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = left;
  __syncthreads();
  if (threadIdx.x == 0) cl[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// ONE workgroup per segment: the exact median among the candidates, the children, the candidates' share of the left counts
__global__ __launch_bounds__(1024) void k_gs_select(const GsSegBlocks* __restrict__ sblk, const float4* __restrict__ p,
                                                   const SsnSeg* __restrict__ segs, const uint32_t* __restrict__ sig,
                                                   uint32_t* __restrict__ gh1, uint32_t* __restrict__ gh2,
                                                   uint32_t* __restrict__ gh1_next, uint32_t* __restrict__ gh2_next,
                                                   uint32_t* __restrict__ cand_n,
                                                   const GsMedian* __restrict__ cand, const uint32_t* __restrict__ cand_blk, uint32_t cap,
                                                   uint32_t* __restrict__ cl, uint32_t* __restrict__ clp, GsMedian* __restrict__ median,
                                                   SsnSeg* __restrict__ out, uint32_t* __restrict__ sig_out,
                                                   uint32_t* __restrict__ cand_n_next, uint2* __restrict__ rng_next,
                                                   uint32_t* __restrict__ err) {
  __shared__ GsMedian lc[kGsCandCap];    // 32 KB
  __shared__ uint32_t med_slot;
  __shared__ GsMedian med_sh;
  __shared__ uint32_t sh[8];
  __shared__ uint32_t rh[256];
  __shared__ uint32_t lcnt[kGsSelBlocks];   // candidates that go left, per block of the segment (32 KB)
  __shared__ uint32_t lblk[kGsCandCap];     // the candidates' blocks (a global load per candidate inside the rank loops: 2 us each, one after the other)
  LSGPU_SEL_T(48);
  const uint32_t s = blockIdx.x;
  const GsSegBlocks q = sblk[s];
  const uint32_t n = cand_n[s];
  const uint32_t left = q.count - q.count / 2u;
  uint32_t mb1, below1, mb2, below2;
  gs_find_bin(gh1 + (size_t)s * 256u, left, sh, mb1, below1);
  gs_find_bin(gh2 + (size_t)s * 256u, left - below1, sh, mb2, below2);
  const uint32_t target = left - below1 - below2;    // rank of the median among the candidates
  LSGPU_SEL_T(49);
  // the children's histograms for the next level (this segment's own have been read by every kernel that needs them)
  if (threadIdx.x < 256u) {
    gh1_next[(size_t)(2u * s) * 256u + threadIdx.x] = 0u; gh1_next[(size_t)(2u * s + 1u) * 256u + threadIdx.x] = 0u;
    gh2_next[(size_t)(2u * s) * 256u + threadIdx.x] = 0u; gh2_next[(size_t)(2u * s + 1u) * 256u + threadIdx.x] = 0u;
  }
  if (threadIdx.x == 0) {
    med_slot = 0xFFFFFFFFu;
    cand_n_next[2u * s] = 0u; cand_n_next[2u * s + 1u] = 0u;
    rng_next[2u * s] = make_uint2(0xFFFFFFFFu, 0u); rng_next[2u * s + 1u] = make_uint2(0xFFFFFFFFu, 0u);   // (k_gs_part fills them)
    if (n > cap || target >= n || mb1 == 0xFFFFFFFFu || mb2 == 0xFFFFFFFFu) { err[0] = 1u; err[1] = 3u; err[2] = s; err[3] = n; err[4] = target; }   // too many equal keys around the median: the host falls back
  }
  const uint32_t c = min(n, cap);
  const GsMedian* __restrict__ gc = cand + (size_t)s * cap;          // this segment's candidates ...
  const uint32_t* __restrict__ gb = cand_blk + (size_t)s * cap;     // ... and their blocks
  const bool in_lds = c <= kGsCandCap;                               // (more: the radix select below works on the global list)
  // the candidates that go left are counted per block of the segment: in LDS (a few hundred device-scope atomics from this
  // one CU, their old values waited for, were ~10 us of the first two levels), in the global counts only for a segment of
  // more blocks than the LDS array has words
  const bool lds_counts = q.nb <= kGsSelBlocks;
  if (lds_counts) for (uint32_t b = threadIdx.x; b < q.nb; b += blockDim.x) lcnt[b] = 0u;
  if (in_lds) for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) { lc[i] = gc[i]; lblk[i] = gb[i]; }
  __syncthreads();
  LSGPU_SEL_T(50);
#ifdef LSGPU_KNN_STATS
  if (gridDim.x == 1 && threadIdx.x == 0) g_tree_dbg[54] = c;
#endif
  uint32_t arrived = 0u;
  if (c <= 256u) {
    // up to 256 candidates (all but the first levels of a sub-map): ranks by c x c comparisons, the LANES over the other
    // candidate (registers), a wave per candidate, counted by ballots.  (One thread per candidate looping over the others
    // in LDS was 12 us at 181 candidates: three waves, nothing to hide the LDS latency behind.)
    const int lane = threadIdx.x & 63;
    const uint32_t w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    GsMedian oj[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const uint32_t j = (uint32_t)(ch * 64 + lane);
      oj[ch] = j < c ? lc[j] : GsMedian{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // (less than nothing)
    }
    for (uint32_t i = w; i < c; i += nw) {
      const GsMedian me = lc[i];
      uint32_t r = 0u;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
        if ((uint32_t)(ch * 64) < c) r += (uint32_t)__popcll(__ballot(gs_less(oj[ch].ka, oj[ch].k1, oj[ch].k2, oj[ch].e, me)));
      if (lane == 0) {
        if (r < target) {                                                                      // this candidate goes left
          const uint32_t blk = lblk[i];
          if (lds_counts) atomicAdd(&lcnt[blk - q.fb], 1u); else arrived ^= atomicAdd(&cl[blk], 1u);
        }
        if (r == target) { med_slot = i; med_sh = me; }
      }
    }
  } else {
    // a thousand and more (a 1 M-point segment has 16 points per bin on average, many more where the sensor stands; a wall
    // square to an axis of the frame puts ten thousand points of a three-scan sub-map into one 3.7 mm bin): the c x c
    // comparisons all run on this one CU -- 17 us at 1 000 candidates, 75 at 2 000, whatever the number of threads.
    // Instead: the candidate of rank `target` by a radix select over the 16 bytes of the tuple (cut-axis key, the two
    // other keys, index; most significant first; it stops as soon as one candidate is left: after the third or fourth
    // byte unless the keys tie), then ONE comparison per candidate.  A candidate is still in the running while its
    // leading bytes are the ones chosen so far; up to kGsCandCap candidates are read from LDS, more from the global list.
    GsMedian ch{0u, 0u, 0u, 0u};                         // the bytes chosen so far
    uint32_t rem = target;
    bool found = false;
    for (int pass = 0; pass < 16 && !found; ++pass) {
      const int field = pass >> 2, shift = 24 - 8 * (pass & 3);
      const uint32_t himask = shift == 24 ? 0u : 0xFFFFFFFFu << (shift + 8);   // the current field's bytes already chosen
      if (threadIdx.x < 256u) rh[threadIdx.x] = 0u;
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
        const GsMedian o = in_lds ? lc[i] : gc[i];
        const bool before_ok = (field < 1 || o.ka == ch.ka) && (field < 2 || o.k1 == ch.k1) && (field < 3 || o.k2 == ch.k2);
        const uint32_t f = field == 0 ? o.ka : field == 1 ? o.k1 : field == 2 ? o.k2 : o.e;
        const uint32_t cf = field == 0 ? ch.ka : field == 1 ? ch.k1 : field == 2 ? ch.k2 : ch.e;
        if (before_ok && ((f ^ cf) & himask) == 0u) atomicAdd(&rh[(f >> shift) & 255u], 1u);
      }
      __syncthreads();
      uint32_t bin, below;
      gs_find_bin(rh, rem, sh, bin, below);
      if (bin == 0xFFFFFFFFu) break;                     // (rank outside the candidates: flagged above)
      rem -= below;
      found = rh[bin] == 1u;
      const uint32_t add = bin << shift;
      if (field == 0) ch.ka |= add; else if (field == 1) ch.k1 |= add; else if (field == 2) ch.k2 |= add; else ch.e |= add;
      if (found) {                                       // one candidate has these leading bytes: the median
        const uint32_t lomask = shift == 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu << shift;
        for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
          const GsMedian o = in_lds ? lc[i] : gc[i];
          const bool before_ok = (field < 1 || o.ka == ch.ka) && (field < 2 || o.k1 == ch.k1) && (field < 3 || o.k2 == ch.k2);
          const uint32_t f = field == 0 ? o.ka : field == 1 ? o.k1 : field == 2 ? o.k2 : o.e;
          const uint32_t cf = field == 0 ? ch.ka : field == 1 ? ch.k1 : field == 2 ? ch.k2 : ch.e;
          if (before_ok && ((f ^ cf) & lomask) == 0u) { med_slot = i; med_sh = o; }
        }
      }
      __syncthreads();                                   // (rh is cleared again at the top; med_sh is complete)
    }
    if (med_slot != 0xFFFFFFFFu) {
      const GsMedian m = med_sh;
      for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
        const GsMedian o = in_lds ? lc[i] : gc[i];
        if (gs_less(o.ka, o.k1, o.k2, o.e, m)) {
          const uint32_t blk = in_lds ? lblk[i] : gb[i];
          if (lds_counts) atomicAdd(&lcnt[blk - q.fb], 1u); else arrived ^= atomicAdd(&cl[blk], 1u);
        }
      }
    }
  }
  // (the atomics' old values waited for = they have been performed where the loads below look; no __threadfence(): on
  //  gfx950 that is a write-back and an invalidation of the XCD's whole L2)
  asm volatile("" ::"v"(arrived));
  __syncthreads();
  LSGPU_SEL_T(51);
  // the blocks' left counts are complete: their exclusive prefix over the segment's blocks, for k_gs_part
  {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool act = threadIdx.x < 256u;
    uint32_t carry = 0u;
    for (uint32_t b0 = 0; b0 < q.nb; b0 += 256u) {
      const uint32_t b = b0 + threadIdx.x;
      const uint32_t v = act && b < q.nb ? (lds_counts ? cl[q.fb + b] + lcnt[b] : __hip_atomic_load(&cl[q.fb + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0u;
      const uint32_t incl = wave_scan_incl_u32(v, lane);
      if (act && lane == 63) sh[w] = incl;
      __syncthreads();
      uint32_t before = carry;
      for (int ww = 0; ww < w && ww < 4; ++ww) before += sh[ww];
      if (act && b < q.nb) clp[q.fb + b] = before + incl - v;
      carry += sh[0] + sh[1] + sh[2] + sh[3];
      __syncthreads();
    }
  }
  LSGPU_SEL_T(52);
  if (threadIdx.x == 0) {
    // (no median: an error has been flagged above and the host repeats the filter -- but everything queued behind this
    //  level, k_ssn_tree included, runs first and must find segments that are what the static halving says they are)
    const bool have = med_slot != 0xFFFFFFFFu;
    const GsMedian m = have ? med_sh : GsMedian{0u, 0u, 0u, 0u};
    median[s] = m;
    const SsnSeg sg = segs[s];
    const int cut = ssn_cut_axis(sg);
    SsnSeg a = sg, b = sg;
    const float cutval = have ? coord_of(p[m.e], cut) : coord_of(make_float4(sg.hi[0], sg.hi[1], sg.hi[2], 0.f), cut);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      a.hi[d] = d == cut ? cutval : a.hi[d];
      b.lo[d] = d == cut ? cutval : b.lo[d];
    }
    a.start = q.start; a.count = left;
    b.start = q.start + left; b.count = q.count - left;
    out[2u * s] = a; out[2u * s + 1u] = b;
    const uint32_t sv = gs_sig_push(sig[s], (uint32_t)cut);
    sig_out[2u * s] = sv; sig_out[2u * s + 1u] = sv;
  }
  LSGPU_SEL_T(53);
}

// the stable partition: every point (and its three keys) to its child's range of the OUT buffers, left half first
__global__ __launch_bounds__(kGsPartThreads) void k_gs_part(const GsBlock* __restrict__ tab, const SsnSeg* __restrict__ segs,
                                                 const uint32_t* __restrict__ sig, GsSet in, GsSet out,
                                                 const GsMedian* __restrict__ median, const uint32_t* __restrict__ clp,
                                                 const SsnSeg* __restrict__ segs_next, uint2* __restrict__ rng_next,
                                                 uint32_t* __restrict__ err) {
  constexpr int kW = kGsPartThreads / 64, kI = (int)kGsTile / kGsPartThreads;   // waves; positions per thread
  __shared__ uint32_t ws[kW];
  __shared__ uint32_t rr[kW][4];
  __shared__ uint32_t stage[4][kGsTile];   // 32 KB
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const GsBlock sb = tab[blockIdx.x];
  const SsnSeg sg = segs[sb.seg];
  const int a = ssn_cut_axis(sg);
  uint32_t x1, x2;
  gs_other_axes(sig[sb.seg], (uint32_t)a, x1, x2);
  const GsMedian m = median[sb.seg];
  const uint32_t base_left = clp[blockIdx.x];     // left points of the segment's earlier blocks (k_gs_select)
  const uint32_t left_total = sb.seg_count - sb.seg_count / 2u;
  const uint32_t base_right = (sb.first - sb.seg_start) - base_left;
  // wave w owns 64 kI consecutive positions of the block, 64 at a time: stable ranks need positions in order
  uint32_t ev[kI], kx[kI], ky[kI], kz[kI], xl[kI];
  uint32_t carry = 0u;
  // the children's own cut axes (k_gs_select has written their boxes): the key range of either child's points on it
  const int aL = ssn_cut_axis(segs_next[2u * sb.seg]), aR = ssn_cut_axis(segs_next[2u * sb.seg + 1u]);
  uint32_t mnL = 0xFFFFFFFFu, mxL = 0u, mnR = 0xFFFFFFFFu, mxR = 0u;
#pragma unroll
  for (int k = 0; k < kI; ++k) {
    const uint32_t j = (uint32_t)(w * (64 * kI) + k * 64 + lane);
    uint32_t v = 0u;
    ev[k] = kx[k] = ky[k] = kz[k] = 0u;
    if (j < sb.count) {
      const uint32_t i = sb.first + j;
      ev[k] = in.e[i]; kx[k] = in.k[0][i]; ky[k] = in.k[1][i]; kz[k] = in.k[2][i];
      const uint32_t ka = a == 0 ? kx[k] : a == 1 ? ky[k] : kz[k];
      const uint32_t k1 = x1 == kGsNoAxis ? 0u : x1 == 0u ? kx[k] : x1 == 1u ? ky[k] : kz[k];
      const uint32_t k2 = x2 == kGsNoAxis ? 0u : x2 == 0u ? kx[k] : x2 == 1u ? ky[k] : kz[k];
      v = gs_less(ka, k1, k2, ev[k], m) ? 1u : 0u;
      if (v) { const uint32_t kc = aL == 0 ? kx[k] : aL == 1 ? ky[k] : kz[k]; mnL = min(mnL, kc); mxL = max(mxL, kc); }
      else   { const uint32_t kc = aR == 0 ? kx[k] : aR == 1 ? ky[k] : kz[k]; mnR = min(mnR, kc); mxR = max(mxR, kc); }
    }
    const uint32_t incl = tree_wave_scan(v, lane);
    xl[k] = ((carry + incl - v) << 1) | v;      // left points in front of this one inside the wave's range; its own side
    carry += rl_u(incl, 63);
  }
  __syncthreads();
  mnL = ~wave_max_u32(~mnL); mxL = wave_max_u32(mxL); mnR = ~wave_max_u32(~mnR); mxR = wave_max_u32(mxR);
  if (lane == 0) { ws[w] = carry; rr[w][0] = mnL; rr[w][1] = mxL; rr[w][2] = mnR; rr[w][3] = mxR; }
  __syncthreads();
  if (threadIdx.x < 4) {
    // (an atomic only where this block improves on what it can see of the range so far: hundreds of blocks of one segment
    // on the same four words serialise in the L2 otherwise; a stale look costs an atomic that changes nothing)
    const int q = (int)threadIdx.x;
    const bool is_min = (q & 1) == 0;
    uint32_t v = rr[0][q];
    for (int ww = 1; ww < kW; ++ww) v = is_min ? min(v, rr[ww][q]) : max(v, rr[ww][q]);
    uint32_t* word = reinterpret_cast<uint32_t*>(&rng_next[2u * sb.seg + (uint32_t)(q >> 1)]) + (q & 1);
    const uint32_t seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (is_min ? v < seen : v > seen) { if (is_min) atomicMin(word, v); else atomicMax(word, v); }
  }
  uint32_t before = 0u;
  for (int ww = 0; ww < w; ++ww) before += ws[ww];
  uint32_t nleft = 0u;                                         // this block's left points
  for (int ww = 0; ww < kW; ++ww) nleft += ws[ww];
  // through LDS: the block's left points first, then its right ones, each in order -- the stores below are then two runs
  // of consecutive addresses per array instead of 64 lanes alternating between two places
#pragma unroll
  for (int k = 0; k < kI; ++k) {
    const uint32_t j = (uint32_t)(w * (64 * kI) + k * 64 + lane);
    if (j < sb.count) {
      const uint32_t lbefore = before + (xl[k] >> 1);            // left points of this block in front of j
      const uint32_t slot = (xl[k] & 1u) ? lbefore : nleft + (j - lbefore);
      stage[0][slot] = ev[k]; stage[1][slot] = kx[k]; stage[2][slot] = ky[k]; stage[3][slot] = kz[k];
    }
  }
  __syncthreads();
  const uint32_t dst_left = sb.seg_start + base_left, dst_right = sb.seg_start + left_total + base_right;
  const uint32_t pos0 = sb.first - sb.seg_start;   // (the counts of a consistent level: nothing below can leave the segment)
  const bool bad = base_left > pos0 || base_left + nleft > left_total ||
                   (pos0 - base_left) + (sb.count - nleft) > sb.seg_count - left_total;
  if (bad) {   // (only behind an error the earlier kernels have flagged: stay inside the segment -- the host repeats the filter, the
               //  kernels queued behind this one must find valid ids)
    if (threadIdx.x == 0) { err[0] = 1u; err[5] = 4u; err[6] = sb.seg; err[7] = dst_left; }
    for (uint32_t t = threadIdx.x; t < sb.count; t += (uint32_t)kGsPartThreads) {
      const uint32_t d = sb.first + t;
      out.e[d] = stage[0][t]; out.k[0][d] = stage[1][t]; out.k[1][d] = stage[2][t]; out.k[2][d] = stage[3][t];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < kI; ++k) {
    const uint32_t t = (uint32_t)(k * kGsPartThreads) + threadIdx.x;
    if (t < sb.count) {
      const uint32_t d = t < nleft ? dst_left + t : dst_right + (t - nleft);
      out.e[d] = stage[0][t]; out.k[0][d] = stage[1][t]; out.k[1][d] = stage[2][t]; out.k[2][d] = stage[3][t];
    }
  }
}

}  // namespace lsgpu
