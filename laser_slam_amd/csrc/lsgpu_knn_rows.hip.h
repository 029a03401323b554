// lsgpu_knn_rows.hip.h -- the steady-state form of the exact 1-NN correspondence search
// (KDTreeMatcher::findClosests knn 1 / epsilon 0, laser_slam/configurations/icp_default.yaml:9-12, once per
// iteration of icp_.compute at laser_slam/src/laser_track.cpp:496).
//
// Once ICP has settled (capped launches with per-query lower bounds, lsgpu_knn.hip.h) about three quarters of
// the queries are *skips*: their previous match is provably still the unique nearest neighbour (keep) or
// everything is provably beyond the trim cap (far).  k_knn_tile still paid a full broadcast evaluation for a
// wave in which a single lane had to search.  Here the work is split:
//   k_knn_classify  streams over ALL queries (36 B read, 8 B written per query): new distance to the current
//                   match, displacement bound, keep / far decision; skips are final, the others are appended to
//                   a compact work list (block-local order preserved, one atomic per 1024 queries).
//   k_knn_rows      one wave = 64 SEARCHING queries = four DPP rows of 16.  Compacted lanes are four times
//                   further apart than the raw queries, so a chunk that one lane needs is useless to most of the
//                   wave: every row therefore owns its region (<= 4x4x4 cells of the pyramid), its chunk list and
//                   its need list, and an evaluation round stages FOUR chunks -- one per row -- through LDS-DMA
//                   (pass p = points 16p..16p+15 of each row's chunk) and every row evaluates its own.  Per
//                   searching lane the candidate count stays what it was, per wave four times fewer lanes idle.
// Results are independent of how the queries are grouped (canonical ties, lsgpu_knn.hip.h), so the order in
// which blocks append to the work list does not matter.
#pragma once
#include "lsgpu_knn.hip.h"

namespace lsgpu {

constexpr int kClassifyPerBlock = 1024;  // queries per k_knn_classify block (256 threads x 4)

__global__ __launch_bounds__(256) void k_knn_classify(KnnArgs a) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t base_sh;
  Mat34 T; float cap2;
  if (!iter_params(a.st, a.T, a.cap2, 1, T, cap2)) return;  // (uniform: every thread reads the same state)
  Mat34 To;
#pragma unroll
  for (int i = 0; i < 12; ++i) To.m[i] = a.st->T_rows_prev[i];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int blk0 = blockIdx.x * kClassifyPerBlock;
  const bool predict = a.sel_below && a.st->sel_mode;
  const uint32_t w_lo = predict ? a.st->sel_lo : 0u, w_span = predict ? a.st->sel_span : 0u, w_sh = predict ? (uint32_t)a.st->sel_shift : 0u;
  bool search[4];
  uint32_t pos[4];
  uint32_t below_cnt = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = blk0 + u * 256 + tid;
    const bool act = j < a.nq;
    search[u] = false;
    bool fin = false;
    float dfin = 0.f;
    if (act) {
      const float4 r = a.rdq[j];
      const float4 mp = a.prev[j];
      const float lb_in = a.lb[j];
      const float3 q = xform(T, r.x, r.y, r.z);
      const float ub = dist2(q.x - mp.x, q.y - mp.y, q.z - mp.z);
      // same bound arithmetic as k_knn_tile: lb holds for T_rows_prev, the query moved by delta since
      const float3 qo = xform(To, r.x, r.y, r.z);
      const float ddx = q.x - qo.x, ddy = q.y - qo.y, ddz = q.z - qo.z;
      const float delta = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) * (1.0f + 1e-5f) + 1e-7f;
      const float lbn = fmaxf(lb_in * (1.0f - 1e-6f) - delta, 0.f);
      const float lb2 = lbn * lbn;
      const bool keep = ub * (1.0f + 1e-5f) < lb2;
      const bool far = fminf(ub, lb2) > cap2 * (1.0f + 1e-5f);
      a.lb[j] = lbn;  // searching lanes: k_knn_rows reads it back as their carried-over bound
      if (keep || far) { a.d2[j] = ub; fin = true; dfin = ub; }  // match, index and warm start stay as they are
      search[u] = !(keep || far);
    }
    const unsigned long long bal = __ballot(search[u]);
    pos[u] = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[u * 4 + w] = (uint32_t)__popcll(bal);
    if (predict) {  // first half of the trimmed-distance select for the distances that are final here
      const uint32_t bits = __float_as_uint(dfin);
      below_cnt += (uint32_t)__popcll(__ballot(fin && bits < w_lo));
      if (fin && bits >= w_lo && bits - w_lo < w_span) atomicAdd(&a.sel_hist2[(bits - w_lo) >> w_sh], 1u);
    }
  }
  __syncthreads();
  uint32_t run = 0, basev[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 16; ++i) {  // list order inside the block = query order: (u, wave, lane)
    const uint32_t c = wsum[i];
    if ((i & 3) == w) basev[i >> 2] = run;
    run += c;
  }
  // a block's segment starts and ends on a multiple of 16: a DPP row of k_knn_rows never mixes the queries of two
  // blocks (which may sit anywhere in the scan); the filler entries are marked invalid
  const uint32_t run16 = (run + 15u) & ~15u;
  if (tid == 0) base_sh = run ? atomicAdd(a.work_count, run16) : 0u;
  __syncthreads();
  const uint32_t gb = base_sh;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (search[u]) a.work[gb + basev[u] + pos[u]] = (uint32_t)(blk0 + u * 256 + tid);
  if ((uint32_t)tid < run16 - run) a.work[gb + run + (uint32_t)tid] = 0xFFFFFFFFu;
  if (predict && lane == 0 && below_cnt)
    atomicAdd(&a.sel_below[((blockIdx.x * 4u + (uint32_t)w) & (kSelBelowSlots - 1)) * kSelBelowStride], below_cnt);
}

// ---------------------------------------------------------------- row-wise search
constexpr int kRowListCap = 192;    // chunk ids queued per row; a row whose region holds more searches per lane
constexpr int kRowStreamCap = 192;  // candidate slots queued per row before an evaluation is forced (3 rounds of 64)

struct RowsLds {
  // one 4 KiB buffer, two uses that never overlap in time:
  //   evaluation round: slots 16p..16p+15 of each row's candidate stream at buf[64 p + lane]  (p = 0..3)
  //   cull batch      : the chunk descriptor fetched by lane L at buf[2L], buf[2L+1]
  float4 buf[4 * 64];
  uint32_t list[4][kRowListCap];
  // The row's candidate STREAM: the points of every chunk some lane of the row needs, chunk after chunk, each
  // chunk rounded up to a multiple of 4 slots with far pad points.  Rows evaluate 64 slots per round whatever
  // the chunk sizes are (the average chunk holds 17 points: one chunk per row and round left most slots empty).
  uint32_t pidx[4][kRowStreamCap];  // slot -> index into pts
};

// Row r evaluates the `n4` slots (multiple of 4) of its stream staged in this round: slot t sits at
// buf[64 (t / 16) + 16 r + t % 16]; same arithmetic and bookkeeping as tile_eval_slot.  Rows whose stream is
// shorter see far pad points.  gslot = stream position of the round's first slot.
__device__ __forceinline__ void rows_eval(const float4* __restrict__ rbase, uint32_t gslot, uint32_t n4, float qx,
                                          float qy, float qz, float& best, float& sec, int& gs) {
  const f32x2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
  for (uint32_t t = 0; t < n4; t += 4) {
    const float4* p = rbase + ((t >> 4) * 64u + (t & 15u));
    const float4 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
    const f32x2 d0 = dist2_pair(q2x, q2y, q2z, f32x2{c0.x, c1.x}, f32x2{c0.y, c1.y}, f32x2{c0.z, c1.z});
    const f32x2 d1 = dist2_pair(q2x, q2y, q2z, f32x2{c2.x, c3.x}, f32x2{c2.y, c3.y}, f32x2{c2.z, c3.z});
    const float m4 = fminf(fminf(fminf(d0.x, d0.y), d1.x), d1.y);
    sec = __builtin_amdgcn_fmed3f(best, m4, sec);
    if (m4 < best) { best = m4; gs = (int)(gslot + t); }
  }
}

template <bool COMPACT>
__global__ __launch_bounds__(64, 4) void k_knn_rows(KnnArgs a) {
  __shared__ RowsLds lds;
  const int lane = threadIdx.x, row = lane >> 4, k16 = lane & 15;
  const uint32_t count = COMPACT ? *a.work_count : (uint32_t)a.nq;
  const uint32_t tile = blockIdx.x;
  if (tile * 64u >= count) return;
#ifdef LSGPU_KNN_STATS
  const long long t_begin = clock64();
  uint32_t st_rounds = 0, st_needs = 0, st_batches = 0, st_groups = 0;
#endif
  const uint32_t idx = tile * 64u + (uint32_t)lane;
  uint32_t jw = idx < count ? (COMPACT ? a.work[idx] : idx) : 0xFFFFFFFFu;
  const bool act = jw != 0xFFFFFFFFu;  // (segments of the work list are padded to multiples of 16 with invalid entries)
  const int j = act ? (int)jw : 0;
  float4 rraw = make_float4(0.f, 0.f, 0.f, 0.f), mp = rraw;
  float lbn = 0.f;
  if (act) {
    rraw = a.rdq[j];
    mp = a.prev[j];
    if (COMPACT && a.lb) lbn = a.lb[j];  // already reduced by this iteration's displacement (k_knn_classify)
  }
  Mat34 T; float cap2;
  if (!iter_params(a.st, a.T, a.cap2, a.use_state_cap, T, cap2)) return;
  const float cap2s = cap2 * kCapSearchMargin2;
  const GridDev& g = a.g;
  float qx = 0.f, qy = 0.f, qz = 0.f, ub = 0.f, best = INFINITY, sec = INFINITY;
  int bi = -1, grp = -1;
  if (act) {
    const float3 q = xform(T, rraw.x, rraw.y, rraw.z);
    qx = q.x; qy = q.y; qz = q.z;
    bi = __float_as_int(mp.w);
    ub = dist2(qx - mp.x, qy - mp.y, qz - mp.z);
  }
  const float gap = a.use_state_cap ? a.gap : 0.f;
  const float R = sqrtf(prune_lim(ub, gap, cap2s)) * (1.0f + 1e-5f) + 1e-7f;
  const bool straggler = act && !(R <= a.r_cap);  // uncapped launches only: wide balls go to k_knn_fallback
  const bool ing = act && !straggler;

  // ---- the row's region: query box, largest ball, pyramid level with <= 4x4x4 cells, one hash probe per cell
  const float rlx = row_min(ing ? qx : INFINITY), rhx = row_max(ing ? qx : -INFINITY);
  const float rly = row_min(ing ? qy : INFINITY), rhy = row_max(ing ? qy : -INFINITY);
  const float rlz = row_min(ing ? qz : INFINITY), rhz = row_max(ing ? qz : -INFINITY);
  const float Rmax = row_max(ing ? R : 0.f);
  float maxbest = row_max(ing ? prune_lim(ub, gap, cap2s) : 0.f);
  const bool rowing = rhx >= rlx;  // some lane of this row searches
  uint32_t cs[4] = {0u, 0u, 0u, 0u}, ce[4] = {0u, 0u, 0u, 0u};
  {
    const int lim = (1 << (g.bits + g.fine)) - 1;
    const float pad = Rmax + kFineSlack * g.hf;
    int flx = 0, fly = 0, flz = 0, fhx = 0, fhy = 0, fhz = 0, l = 0;
    if (rowing) {
      flx = fine_coord(rlx - pad, g.ox, g.inv_hf, lim); fhx = fine_coord(rhx + pad, g.ox, g.inv_hf, lim);
      fly = fine_coord(rly - pad, g.oy, g.inv_hf, lim); fhy = fine_coord(rhy + pad, g.oy, g.inv_hf, lim);
      flz = fine_coord(rlz - pad, g.oz, g.inv_hf, lim); fhz = fine_coord(rhz + pad, g.oz, g.inv_hf, lim);
      for (int sh = g.fine; l < g.bits; ++l, ++sh)
        if ((fhx >> sh) - (flx >> sh) < 4 && (fhy >> sh) - (fly >> sh) < 4 && (fhz >> sh) - (flz >> sh) < 4) break;
    }
    // rows may sit on different levels: one pass per distinct level keeps the table base / mask scalar
    unsigned long long todo = __ballot(rowing);
    while (todo) {
      const int L = __builtin_amdgcn_readlane(l, __ffsll((long long)todo) - 1);
      const bool mine = rowing && l == L;
      todo &= ~__ballot(mine);
      if (mine) {
        const int sh = g.fine + L;
        const int x0 = flx >> sh, y0 = fly >> sh, z0 = flz >> sh;
        const int nx = (fhx >> sh) - x0 + 1, ny = (fhy >> sh) - y0 + 1, nz = (fhz >> sh) - z0 + 1;
        const uint32_t mask = g.mask[L];
        const uint4* tab = reinterpret_cast<const uint4*>(g.tab[L]);
        const int cx = k16 & 3, cy = k16 >> 2;  // cell (cx, cy, i) of the block, i = 0..3: all four probes in flight
        uint4 en[4];
        uint32_t slot[4];
        bool want[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          want[i] = cx < nx && cy < ny && i < nz;
          slot[i] = cell_hash((uint32_t)(x0 + cx), (uint32_t)(y0 + cy), (uint32_t)(z0 + i)) & mask;
          en[i] = make_uint4(kEmpty, 0u, 0u, 0u);
          if (want[i]) en[i] = tab[slot[i]];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (!want[i]) continue;
          const uint32_t xy = (uint32_t)(x0 + cx) | ((uint32_t)(y0 + cy) << 16), zz = (uint32_t)(z0 + i);
          uint4 e = en[i];
          uint32_t sl = slot[i];
          while (!(((e.x ^ xy) | (e.y ^ zz)) == 0u) && e.x != kEmpty) {  // collision chain (rare)
            sl = (sl + 1) & mask;
            e = tab[sl];
          }
          if (e.x != kEmpty) { cs[i] = e.z; ce[i] = e.w; }
        }
      }
    }
  }
  // ---- flatten the cells' chunk ranges into the row's list
  const uint32_t nch = (ce[0] - cs[0]) + (ce[1] - cs[1]) + (ce[2] - cs[2]) + (ce[3] - cs[3]);
  const uint32_t incl = row_scan_incl_u32(nch);
  const uint32_t tot = row_sum_u32(nch);
  const bool listrow = rowing && tot <= (uint32_t)kRowListCap;
  const bool lanesearch = ing && tot > (uint32_t)kRowListCap;  // over-full region: each lane searches its own ball
  if (listrow) {
    uint32_t o = incl - nch;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      for (uint32_t ch = cs[i]; ch < ce[i]; ++ch) lds.list[row][o++] = ch;
  }
  const uint32_t nbmax = wave_max_u32(listrow ? tot : 0u);

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  uint32_t n_slots = 0;  // length of the row's candidate stream (same value in the 16 lanes of a row)
  // Evaluate what the rows have queued, 64 slots per row and round: four LDS-DMA passes of 1 KiB stage slots
  // 16p..16p+15 of EVERY row's stream, then each row evaluates its own; afterwards the bounds are tightened.
  auto evaluate = [&]() {
    const uint32_t maxslots = wave_max_u32(n_slots);
    int gs = -1;  // stream position of the group of 4 holding the new best, if any
    for (uint32_t s0 = 0; s0 < maxslots; s0 += 64u) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // nothing still reads the bytes the DMA overwrites
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (s0 + (uint32_t)(p * 16) < maxslots) {
          const uint32_t sl = s0 + (uint32_t)(p * 16 + k16);
          // slots past the row's stream fetch a far pad point: every slot that is read is defined
          const uint32_t gi = sl < n_slots ? lds.pidx[row][sl] : (uint32_t)a.pad_index;
          __builtin_amdgcn_global_load_lds((gptr_t)(a.pts + gi), (lptr_t)&lds.buf[p * 64], 16, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t n4 = maxslots - s0 < 64u ? maxslots - s0 : 64u;  // (stream lengths are multiples of 4)
#ifdef LSGPU_KNN_STATS
      st_groups += n4 >> 2; ++st_rounds;
#endif
      if (s0 < n_slots) rows_eval(&lds.buf[row * 16], s0, n4, qx, qy, qz, best, sec, gs);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (gs >= 0) grp = (int)lds.pidx[row][gs];  // the group's first point (4 consecutive points of one chunk)
    n_slots = 0;
    maxbest = row_max(ing ? prune_lim(fminf(best, ub), gap, cap2s) : 0.f);
  };

  // ---- cull 16 queued chunks per row and step (lane = chunk) against the row's query box, test the survivors
  // per lane against each lane's own bound, append what some lane of the row needs to the row's stream
  auto load_desc = [&](uint32_t base, float4& b0, float4& b1) -> bool {
    const uint32_t e = base + (uint32_t)k16;
    const bool v = listrow && e < tot;
    b0 = make_float4(0.f, 0.f, 0.f, 0.f); b1 = b0;
    if (v) {
      const float4* cd = reinterpret_cast<const float4*>(a.chunks + lds.list[row][e]);
      b0 = cd[0]; b1 = cd[1];
    }
    return v;
  };
  float4 nb0, nb1;
  bool nvalid = load_desc(0u, nb0, nb1);
  for (uint32_t base = 0; base < nbmax; base += 16u) {
    const float4 b0 = nb0, b1 = nb1;
    const bool valid = nvalid;
#ifdef LSGPU_KNN_STATS
    ++st_batches;
#endif
    if (base + 16u < nbmax) nvalid = load_desc(base + 16u, nb0, nb1);  // next batch's descriptors in flight
    bool pass = false;
    if (valid) {
      const float gx = fmaxf(fmaxf(b0.x - rhx, rlx - b1.x), 0.f);
      const float gy = fmaxf(fmaxf(b0.y - rhy, rly - b1.y), 0.f);
      const float gz = fmaxf(fmaxf(b0.z - rhz, rlz - b1.z), 0.f);
      pass = (gx * gx + gy * gy + gz * gz) * kPruneShrink <= maxbest;
    }
    const unsigned long long bal = __ballot(pass);
    uint32_t m16 = (uint32_t)(bal >> (row * 16)) & 0xFFFFu;  // the row's survivors still to be looked at
    while (__ballot(m16 != 0u)) {
      lds.buf[2 * lane] = b0;      // (again after an evaluation in between: the staging buffer is shared)
      lds.buf[2 * lane + 1] = b1;
      const float limq = ing ? prune_lim(fminf(best, ub), gap, cap2s) : 0.f;  // bounds as of now; they only tighten
      bool stalled = false;  // the row's stream is full: evaluate first, then carry on with the same survivor
      while (__ballot(m16 != 0u && !stalled)) {
        const bool has = m16 != 0u && !stalled;
        const uint32_t kk = has ? (uint32_t)(__ffs((int)m16) - 1) : 0u;
        const float4 c0 = lds.buf[2 * (row * 16 + (int)kk)], c1 = lds.buf[2 * (row * 16 + (int)kk) + 1];
        const bool need = has && ing && box_dist2(c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, qx, qy, qz) * kPruneShrink <= limq;
        const unsigned long long nbal = __ballot(need);
        const bool rneed = has && ((uint32_t)(nbal >> (row * 16)) & 0xFFFFu) != 0u;
        const uint32_t st = __float_as_uint(c0.w), cnt = __float_as_uint(c1.w), cnt4 = (cnt + 3u) & ~3u;
        if (rneed && n_slots + cnt4 > (uint32_t)kRowStreamCap) {
          stalled = true;
        } else {
          if (rneed) {
            for (uint32_t o = (uint32_t)k16; o < cnt4; o += 16u)
              lds.pidx[row][n_slots + o] = o < cnt ? st + o : (uint32_t)a.pad_index;
            n_slots += cnt4;
#ifdef LSGPU_KNN_STATS
            if (k16 == 0) ++st_needs;
#endif
          }
          if (has) m16 &= m16 - 1u;
        }
      }
      if (__ballot(stalled)) evaluate();
    }
    const bool last = base + 16u >= nbmax;
    if (last && __ballot(n_slots != 0u)) evaluate();
  }

  if (lanesearch) {  // tracks the exact index itself
    const int before = bi;
    best = ub;
    lane_ball_search(a, cap2s, qx, qy, qz, best, bi);
    if (bi != before) { const float4 p = a.pts[bi]; mp = make_float4(p.x, p.y, p.z, __int_as_float(bi)); }
  }
  if (act) {
    float nb;  // new lower bound on the distance to every point other than the (new) match
    if (lanesearch) {
      nb = best <= cap2s ? sqrtf(best) * (1.0f - 1e-6f) : fmaxf(lbn, sqrtf(cap2s) * (1.0f - 1e-5f));
    } else if (!ing) {
      best = ub;  // straggler: k_knn_fallback overwrites this result
      nb = lbn;
    } else {
      const float lim_f = prune_lim(fminf(best, ub), gap, cap2s);  // every unevaluated point is beyond this
      if (best <= ub) {  // the evaluated minimum (the warm-start point itself unless something beat it)
        float s4 = INFINITY;
        mp = tile_resolve_match(a, grp, qx, qy, qz, best, mp, s4);
        float others = fminf(fminf(sec, s4), lim_f);
        bool same = __float_as_int(mp.w) == bi;
        if (sec == best || (!same && best == ub)) {  // a second point at exactly this distance: smallest index
          mp = canonical_tie(a, qx, qy, qz, best, mp);
          same = __float_as_int(mp.w) == bi;
        }
        if (!same) others = fminf(others, ub);  // (covers a warm-start point whose chunk was not needed)
        nb = sqrtf(others) * (1.0f - 1e-5f);
        if (same) nb = fmaxf(nb, lbn);
      } else {  // the warm-start point's chunk was beyond the cap and nothing closer exists
        nb = fmaxf(sqrtf(fminf(best, lim_f)) * (1.0f - 1e-5f), lbn);
        best = ub;
      }
    }
    a.ids[j] = __float_as_int(mp.w);
    a.d2[j] = best;
    a.prev[j] = mp;
    if (a.lb) a.lb[j] = nb;
    if (straggler) a.strag[atomicAdd(a.strag_count, 1u)] = (uint32_t)j;
  }
  if (a.sel_below && a.st->sel_mode) {  // first half of the trimmed-distance select (see k_knn_tile)
    const uint32_t bits = __float_as_uint(best), w_lo = a.st->sel_lo;
    const unsigned long long below = __ballot(act && bits < w_lo);
    if (act && bits >= w_lo && bits - w_lo < a.st->sel_span) atomicAdd(&a.sel_hist2[(bits - w_lo) >> a.st->sel_shift], 1u);
    if (lane == 0 && below)
      atomicAdd(&a.sel_below[(tile & (kSelBelowSlots - 1)) * kSelBelowStride], (uint32_t)__popcll(below));
  }
#ifdef LSGPU_KNN_STATS
  {
    const uint32_t n_ls = (uint32_t)__popcll(__ballot(lanesearch));
    if (lane == 0 && a.dbg_wave)
      a.dbg_wave[tile] = make_uint4((uint32_t)(clock64() - t_begin),
                                    (st_rounds << 16) | (rl_u(st_needs, 0) + rl_u(st_needs, 16) + rl_u(st_needs, 32) + rl_u(st_needs, 48)),
                                    (st_groups << 8) | st_batches, (n_ls << 16) | (nbmax & 0xFFFFu));
  }
#endif
}

}  // namespace lsgpu
