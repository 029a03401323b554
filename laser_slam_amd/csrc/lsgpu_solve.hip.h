// lsgpu_solve.hip.h -- TrimmedDistOutlierFilter (icp_default.yaml:14-16) as an exact radix select and
// PointToPlaneErrorMinimizer (icp_default.yaml:18-19) as a deterministic normal-equation reduction.
#pragma once
#include "lsgpu_common.hip.h"
#include "lsgpu_host_math.h"
#include "../../include/lsgpu_icp.h"

namespace lsgpu {

// ---------------------------------------------------------------- trimmed-distance order statistic
// Exact radix select on the float bit pattern of d2 (non-negative floats order like their bits):
// pass 1 bits [31:20], pass 2 bits [19:9], pass 3 bits [8:0].
struct SelState {
  uint32_t prefix;  // selected high bits so far
  uint32_t k;       // rank still to find inside the selected bin
};

// Whole block (256 threads): find bin b with cum(b) <= k < cum(b)+hist[b] -- parallel: each thread
// owns nbins/256 consecutive bins, block-wide inclusive scan of the per-thread sums, the one thread
// whose range contains rank k walks its own bins.
__device__ void find_bin(const uint32_t* __restrict__ hist, int nbins, uint32_t k, uint32_t* bin,
                         uint32_t* krem, uint32_t* sh /* >= 260 words */) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int per = nbins / 256;  // nbins is a multiple of 256, per <= 16
  uint32_t c[16];
  uint32_t loc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { c[i] = i < per ? hist[t * per + i] : 0u; loc += c[i]; }
  uint32_t incl = loc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) sh[w] = incl;
  if (t == 0) { sh[256] = (uint32_t)(nbins - 1); sh[257] = 0u; }  // k beyond the total: last bin
  __syncthreads();
  uint32_t base = 0;
  for (int i = 0; i < w; ++i) base += sh[i];
  incl += base;
  const uint32_t excl = incl - loc;
  if (excl <= k && k < incl) {
    uint32_t cum = excl;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i < per) {
        if (k < cum + c[i]) { sh[256] = (uint32_t)(t * per + i); sh[257] = k - cum; break; }
        cum += c[i];
      }
    }
  }
  __syncthreads();
  *bin = sh[256];
  *krem = sh[257];
  __syncthreads();
}

// One pass of a select kernel over the distances: 16 of them per thread and step, their four 16-byte loads in flight
// together (one float per thread and step was a chain of sixteen dependent round trips: 9.5 us for 4 MB).
template <class F>
__device__ __forceinline__ void hist_sweep(const float* __restrict__ d2, int n, F&& f) {
  const int stride = gridDim.x * 256;
  if (reinterpret_cast<uintptr_t>(d2) & 15u) {   // (a caller's unaligned array: one by one)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) f(__float_as_uint(d2[i]));
    return;
  }
  const int n4 = n >> 2;
  const float4* __restrict__ d4 = reinterpret_cast<const float4*>(d2);
  for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * stride; v[u] = i < n4 ? d4[i] : make_float4(-1.f, -1.f, -1.f, -1.f); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u * stride < n4) { f(__float_as_uint(v[u].x)); f(__float_as_uint(v[u].y)); f(__float_as_uint(v[u].z)); f(__float_as_uint(v[u].w)); }
    }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < (n & 3)) f(__float_as_uint(d2[(n4 << 2) + threadIdx.x]));
}

__global__ __launch_bounds__(256) void k_hist1(const float* __restrict__ d2, int n,
                                               uint32_t* __restrict__ hist,
                                               const IcpState* __restrict__ ist, int predicted) {
  __shared__ uint32_t sh[kHistBins];
  if (ist && ist->done) return;
  if (predicted && ist->sel_mode) return;  // the kNN kernel of this iteration did passes 1 and 2
  for (int i = threadIdx.x; i < kHistBins; i += 256) sh[i] = 0;
  __syncthreads();
  hist_sweep(d2, n, [&](uint32_t b) { atomicAdd(&sh[b >> 20], 1u); });
  __syncthreads();
  for (int i = threadIdx.x; i < kHistBins; i += 256)
    if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// PASS 2: shift 9, 11 bits, parent = hist1 ; PASS 3: shift 0, 9 bits, parent = hist2
template <int PASS>
__global__ __launch_bounds__(256) void k_hist_refine(const float* __restrict__ d2, int n,
                                                     const uint32_t* __restrict__ parent,
                                                     const SelState* __restrict__ st_in,
                                                     SelState* __restrict__ st_out,
                                                     uint32_t* __restrict__ hist,
                                                     const IcpState* __restrict__ ist, int predicted,
                                                     uint32_t* __restrict__ sel_aux) {
  __shared__ uint32_t sh[kHistBins];
  __shared__ uint32_t sc[260];
  if (ist && ist->done) return;
  SelState in = *st_in;
  if (predicted && ist->sel_mode) {
    if (PASS == 2) return;  // done by the kNN kernel
    if (ist->sel_wide) return;  // fused select: the normal-equation kernel settles the rest
    // PASS 3 after a predicted first half: `parent` (the 11-bit histogram inside the predicted bin) and the
    // counters of smaller distances come from the kNN kernel; st_in[-1] is the select's input {0, k}.  The rank must
    // fall inside the bin, otherwise the prediction failed and the iteration is repeated with the full select.
    in = st_in[-1];
    uint32_t below = 0, inside = 0;
    for (int i = threadIdx.x; i < kSelBelowSlots; i += 256) below += sel_aux[i * kSelBelowStride];
    for (int i = threadIdx.x; i < kHistBins; i += 256) inside += parent[i];
    below = wave_sum_u32(below); inside = wave_sum_u32(inside);
    if ((threadIdx.x & 63) == 0) { sc[threadIdx.x >> 6] = below; sc[4 + (threadIdx.x >> 6)] = inside; }
    __syncthreads();
    below = sc[0] + sc[1] + sc[2] + sc[3];
    inside = sc[4] + sc[5] + sc[6] + sc[7];
    __syncthreads();
    if (in.k < below || in.k - below >= inside) {
      if (blockIdx.x == 0 && threadIdx.x == 0) sel_aux[kSelFailFlag] = 1u;
      return;
    }
    in.prefix = ist->sel_bin1;
    in.k -= below;
  }
  uint32_t bin, krem;
  find_bin(parent, kHistBins, in.k, &bin, &krem, sc);
  const uint32_t prefix = (PASS == 2) ? bin : ((in.prefix << 11) | bin);
  if (blockIdx.x == 0 && threadIdx.x == 0) { st_out->prefix = prefix; st_out->k = krem; }
  for (int i = threadIdx.x; i < kHistBins; i += 256) sh[i] = 0;
  __syncthreads();
  constexpr int SH_HI = (PASS == 2) ? 20 : 9;
  constexpr int SH_LO = (PASS == 2) ? 9 : 0;
  constexpr uint32_t MASK = (PASS == 2) ? 0x7FFu : 0x1FFu;
  hist_sweep(d2, n, [&](uint32_t b) { if ((b >> SH_HI) == prefix) atomicAdd(&sh[(b >> SH_LO) & MASK], 1u); });
  __syncthreads();
  for (int i = threadIdx.x; i < kHistBins; i += 256)
    if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// limit bits = prefix(23 high bits) << 9 | bin
__device__ __forceinline__ float select_limit(const uint32_t* __restrict__ hist3,
                                              const SelState* __restrict__ st, uint32_t* sc) {
  const SelState in = *st;
  uint32_t bin, krem;
  find_bin(hist3, kHistBins, in.k, &bin, &krem, sc);
  return __uint_as_float((in.prefix << 9) | bin);
}

__global__ __launch_bounds__(256) void k_limit_out(const uint32_t* __restrict__ hist3,
                                                   const SelState* __restrict__ st,
                                                   float* __restrict__ out) {
  __shared__ uint32_t sc[260];
  const float lim = select_limit(hist3, st, sc);
  if (threadIdx.x == 0) *out = lim;
}

// First iteration: the select just ran over the SEED distances (upper bounds of the true ones); its result is a
// valid cap for the first search (limit(true distances) <= limit(upper bounds)).
__global__ __launch_bounds__(256) void k_seed_cap(uint32_t* __restrict__ hist /* 3 x kHistBins */,
                                                  const SelState* __restrict__ st, IcpState* __restrict__ ist) {
  __shared__ uint32_t sc[260];
  if (ist->done) return;
  // (two passes of the select ran: the UPPER edge of the second pass's bin that holds the order statistic -- a bound
  // is all the cap has to be, and that edge lies within 6e-5 of the exact value; st = the state after the first pass)
  const SelState in = *st;
  uint32_t bin, krem;
  find_bin(hist + kHistBins, kHistBins, in.k, &bin, &krem, sc);
  const float lim = __uint_as_float((((in.prefix << 11) | bin) << 9) | 0x1FFu);
  if (threadIdx.x == 0) ist->cap2 = lim;
  for (int i = threadIdx.x; i < 3 * kHistBins; i += 256) hist[i] = 0u;   // re-armed for the iteration's own select
}

// Start of an align: the loop state, the checker history's first entry, the select's input and every per-iteration
// scratch table in ONE launch (there used to be two uploads and eight memsets, 4-5 us each on the stream).
struct AlignInitArgs {
  IcpState state;
  float chk0[8];
  SelState sel0;
  IcpState* state_dev; float* chk_hist; SelState* sel;
  uint32_t *counters3, *ne_ticket, *sel_aux, *spread_flag, *spread_cnt, *sel_win, *hist;
  int n_sel_aux, n_spread_flag, n_sel_win;
};
__global__ __launch_bounds__(256) void k_align_init(AlignInitArgs a) {
  const int t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  if (blockIdx.x == 0) {
    constexpr int kWords = (int)(sizeof(IcpState) / sizeof(uint32_t));
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.state);
    if ((int)threadIdx.x < kWords) reinterpret_cast<uint32_t*>(a.state_dev)[threadIdx.x] = src[threadIdx.x];
    if (threadIdx.x < 8) a.chk_hist[threadIdx.x] = a.chk0[threadIdx.x];
    if (threadIdx.x == 8) a.sel[0] = a.sel0;
    if (threadIdx.x >= 16 && threadIdx.x < 19) a.counters3[threadIdx.x - 16] = 0u;
    if (threadIdx.x >= 32 && threadIdx.x < 40) a.ne_ticket[threadIdx.x - 32] = 0u;
    if (threadIdx.x >= 64 && threadIdx.x < 66) a.spread_cnt[threadIdx.x - 64] = 0u;
  }
  for (int i = t; i < a.n_sel_aux; i += stride) a.sel_aux[i] = 0u;
  for (int i = t; i < a.n_spread_flag; i += stride) a.spread_flag[i] = 0u;
  for (int i = t; i < a.n_sel_win; i += stride) a.sel_win[i] = 0u;
  for (int i = t; i < 3 * kHistBins; i += stride) a.hist[i] = 0u;
}

// ---------------------------------------------------------------- point-to-plane normal equations
// Per pair with weight 1: J = [p x n ; n] (float, as libpointmatcher), r = (p - q).n ;
// accumulate 21 upper-tri J J^T, 6 of -J r, count, r^2 in double.  Per-block partials, then a
// single-block fixed-order reduction => bitwise reproducible.
constexpr int kNe = 29;
constexpr int kNeGroup = 16;   // k_normal_eq_loop: blocks per first-level reduction group

template <bool IDS_ORIG, bool LIMIT_DEV>
__global__ __launch_bounds__(256) void k_normal_eq(const float4* __restrict__ rdq, int nq, Mat34 T,
                                                   const int* __restrict__ ids,
                                                   const float* __restrict__ d2,
                                                   const float4* __restrict__ pts,
                                                   const float4* __restrict__ nrm,
                                                   const uint32_t* __restrict__ inv,
                                                   const uint32_t* __restrict__ hist3,
                                                   const SelState* __restrict__ st, float limit_val,
                                                   float* __restrict__ limit_out,
                                                   double* __restrict__ partials) {
  __shared__ uint32_t sc[260];
  __shared__ double red[4][kNe];
  float limit = limit_val;
  if (LIMIT_DEV) {
    limit = select_limit(hist3, st, sc);
    if (blockIdx.x == 0 && threadIdx.x == 0) *limit_out = limit;
  }
  double acc[kNe];
#pragma unroll
  for (int k = 0; k < kNe; ++k) acc[k] = 0.0;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < nq; j += gridDim.x * 256) {
    const float d = d2[j];
    int id = ids[j];
    if (!(d <= limit) || id < 0) continue;
    if (IDS_ORIG) id = (int)inv[id];
    const float4 r = rdq[j];
    const float3 p = xform(T, r.x, r.y, r.z);
    const float4 q = pts[id];
    const float4 n = nrm[id];
    float J[6];
    J[0] = p.y * n.z - p.z * n.y;
    J[1] = p.z * n.x - p.x * n.z;
    J[2] = p.x * n.y - p.y * n.x;
    J[3] = n.x; J[4] = n.y; J[5] = n.z;
    const float res = (p.x - q.x) * n.x + (p.y - q.y) * n.y + (p.z - q.z) * n.z;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = a; c < 6; ++c) acc[k++] += (double)J[a] * (double)J[c];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] -= (double)J[a] * (double)res;
    acc[27] += 1.0;
    acc[28] += (double)res * (double)res;
  }
#pragma unroll
  for (int k = 0; k < kNe; ++k) acc[k] = wave_sum(acc[k]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < kNe; ++k) red[w][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < kNe)
    partials[(size_t)blockIdx.x * 32 + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

constexpr float kCapFactor = 1.1f;  // next search cap = this x the current trim limit (verified each iteration)

#ifdef LSGPU_KNN_STATS
// stats build only: where k_normal_eq_loop spends its time.  [0] launches, [1..3] sum over blocks of the cycles in
// {prologue (limit), main loop, wave+block reduce and hand-off}, [4] blocks, [5] first block start (100 MHz wall clock,
// re-armed by the last block), [6] latest end of a main loop, [7] sum of (last loop end - first start), [8] sum of
// (kernel end - last loop end), [9] sum of (first loop START - first start) i.e. prologue wall time of the earliest block
__device__ unsigned long long g_ne_dbg[24] = {0, 0, 0, 0, 0, ~0ull, 0, 0, 0, 0, ~0ull, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
__device__ unsigned long long g_ne_tail[4];   // (wall clock at points of the last block's tail)
#endif
// ---------------------------------------------------------------- per-iteration update (device side)
// One lane: 6x6 float LLT solve, AngleAxis update, T_iter <- dT * T_iter, Counter + Differential
// checkers, trace record, next cap.  Same code as the host (lsgpu_host_math.h).
__device__ inline void icp_update_lane(IcpState* st, const double* ne_out, float* chk_hist,
                                       lsgpu_iter_trace* trace, int trace_cap, int capped_launch,
                                       uint32_t* sel_aux, int sel_failed = -1 /* -1: read (and clear) the flag in sel_aux */) {
  if (st->done) return;
  if (sel_failed < 0) {
    sel_failed = (sel_aux && sel_aux[kSelFailFlag]) ? 1 : 0;
    if (sel_failed) sel_aux[kSelFailFlag] = 0u;
  }
  if (sel_failed) {
    // the predicted select missed (the limit left its 12-bit bin): nothing of this iteration is usable
    st->sel_mode = 0;
    st->sel_fails += 1;
    st->status = kStatusSelFailed;  // the distances of this iteration stand: the host re-runs the full select on them
    st->done = 1;
    return;
  }
  const float limit = (float)ne_out[29];
  const unsigned long long nstrag = (unsigned long long)ne_out[30];
  if (capped_launch && st->cap2 < INFINITY && !(limit <= st->cap2)) {
    st->status = kStatusCapFailed;  // the order statistic is not among exact values: repeat uncapped
    st->done = 1;
    return;
  }
#ifdef LSGPU_KNN_STATS
  const long long u0 = clock64();
#endif
  st->stragglers += nstrag;
  const long long used = (long long)ne_out[27];
  if (used <= 0) { st->status = LSGPU_NO_CONVERGENCE; st->err_code = 1; st->done = 1; return; }
  double A[36], b[6];
  hostmath::unpack_normal_eq(ne_out, A, b);
  float x[6], dT[16], Tn[16];
  if (!hostmath::llt_solve6(A, b, x)) { st->status = LSGPU_NO_CONVERGENCE; st->err_code = 2; st->done = 1; return; }
#ifdef LSGPU_KNN_STATS
  const long long u1 = clock64();
#endif
  hostmath::delta_from_x(x, dT);
  hostmath::mul4(dT, st->T_iter, Tn);
  for (int i = 0; i < 16; ++i) st->T_iter[i] = Tn[i];
  for (int i = 0; i < 12; ++i) st->T_rows_prev[i] = st->T_rows[i];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) st->T_rows[r * 4 + c] = Tn[c * 4 + r];
#ifdef LSGPU_KNN_STATS
  const long long u2 = clock64();
#endif
  const int it = st->iter;
  st->last_limit = limit; st->last_used = used;
  if (it < trace_cap) {
    lsgpu_iter_trace& tr = trace[it];
    for (int i = 0; i < 16; ++i) tr.T_iter[i] = Tn[i];
    tr.limit = limit; tr.n_used = used;
    for (int i = 0; i < 36; ++i) tr.A[i] = A[i];
    for (int i = 0; i < 6; ++i) { tr.b[i] = b[i]; tr.x[i] = x[i]; }
    tr.knn_main_us = 0.f; tr.knn_fallback_us = 0.f; tr.stragglers = (uint32_t)nstrag; tr.reserved = (uint32_t)ne_out[31];
  }
#ifdef LSGPU_KNN_STATS
  const long long u3 = clock64();
#endif
  {
    const uint32_t lb = __float_as_uint(limit), b1 = lb >> 20, b2 = (lb >> 9) & 0x7FFu;
    if (st->sel_wide) {
      // the next limit is looked for in the octave that starts at 0.7 x this one; armed (and the streak counted) once a
      // limit has moved by less than - 25 % / + 20 %: the steps of an alignment shrink, so will the limit's
      const float prev = st->prev_limit;
      const bool armed = limit > 1e-30f && limit < 1e30f && prev < 1e30f && limit >= kSelArmLo * prev && limit <= kSelArmHi * prev && st->sel_fails < 2;
      // the window: as far around this limit as four times its last move plus 1 %, at most [0.7, 1.4] x (a narrow window
      // means few lanes of the search add to the slice histogram; the counters of the distances below it cost one atomic per wave)
      const float move = armed ? fabsf(limit - prev) / prev : 1.f;
      const float f_lo = fmaxf(kSelWideLo, 1.f - 4.f * move - 0.01f), f_hi = fminf(2.f * kSelWideLo, 1.f + 4.f * move + 0.01f);
      const uint32_t lo_s = __float_as_uint(limit * f_lo) >> kSelSliceShift, hi_s = (__float_as_uint(limit * f_hi) >> kSelSliceShift) + 1u;
      st->sel_lo = lo_s << kSelSliceShift;
      st->sel_span = (hi_s - lo_s < (uint32_t)kSelSlices ? hi_s - lo_s : (uint32_t)kSelSlices) << kSelSliceShift;
      st->sel_shift = kSelSliceShift;
      st->sel_streak = armed ? st->sel_streak + 1 : 0;
      st->sel_mode = armed ? 1 : 0;
    } else {
      const uint32_t moved = b2 > st->sel_bin2 ? b2 - st->sel_bin2 : st->sel_bin2 - b2;
      st->sel_streak = (b1 == st->sel_bin1 && moved <= (uint32_t)kSelStreakBins) ? st->sel_streak + 1 : 0;
      st->sel_mode = (b1 == st->sel_bin1) ? 1 : 0;  // predict only a bin that has just been confirmed
      st->sel_lo = b1 << 20; st->sel_span = 1u << 20; st->sel_shift = 9;
    }
    st->sel_bin1 = b1;
    st->sel_bin2 = b2;
  }
  st->prev_limit = limit;
  st->cap2 = st->cap_enabled ? limit * kCapFactor : INFINITY;
  st->iter = it + 1;
  hostmath::CheckerState cs{st->counter, st->n_hist};
  bool iterate = true, by_diff = false;
  float chk2[2];
  const bool ok = hostmath::checker_check(&cs, chk_hist, st->max_iter, st->smooth, st->lim_rot,
                                          st->lim_trans, Tn, &iterate, &by_diff, chk2);
  st->chk_rot_prev = st->chk_rot; st->chk_trans_prev = st->chk_trans;
  st->chk_rot = chk2[0]; st->chk_trans = chk2[1];
  st->counter = cs.counter; st->n_hist = cs.n_hist;
#ifdef LSGPU_KNN_STATS
  {
    const long long u4 = clock64();
    g_ne_dbg[12] += (unsigned long long)(u1 - u0); g_ne_dbg[13] += (unsigned long long)(u2 - u1);
    g_ne_dbg[14] += (unsigned long long)(u3 - u2); g_ne_dbg[15] += (unsigned long long)(u4 - u3);
  }
#endif
  if (!ok) { st->status = LSGPU_NO_CONVERGENCE; st->err_code = 3; st->done = 1; return; }
  if (!iterate) { st->converged = by_diff ? 1 : 0; st->done = 1; }
}

// stand-alone launch: used when something sits between the normal equations and the update (the RCCL
// all-reduce of the split-scan mode); otherwise the last block of k_normal_eq_loop runs the update itself
__global__ __launch_bounds__(64) void k_icp_update(IcpState* __restrict__ st,
                                                   const double* __restrict__ ne_out,
                                                   float* __restrict__ chk_hist,
                                                   lsgpu_iter_trace* __restrict__ trace, int trace_cap,
                                                   int capped_launch, uint32_t* __restrict__ sel_aux) {
  if (threadIdx.x == 0) icp_update_lane(st, ne_out, chk_hist, trace, trace_cap, capped_launch, sel_aux);
}


// The align loop's variant: the matched point comes coalesced from the warm-start array (xyz + sorted
// index of every query's neighbour, written by the kNN kernels), only the normal is gathered.  The
// LAST block to finish (ticket; partial sums exchanged with agent-scope accesses) reduces the block
// partials in a fixed order, publishes {29 sums, limit, straggler count} and re-arms the per-iteration
// scratch (histograms, straggler counter, ticket) so the next iteration needs no memset launches.
#ifndef LSGPU_NE_UNROLL
#define LSGPU_NE_UNROLL 8
#endif
constexpr int kNeUnroll = LSGPU_NE_UNROLL;  // points whose loads are in flight together, per lane
__global__ __launch_bounds__(256) void k_normal_eq_loop(const float4* __restrict__ rdq, int nq,
                                                        IcpState* __restrict__ ist,
                                                        const float4* __restrict__ match,
                                                        const float* __restrict__ d2,
                                                        const float4* __restrict__ nrm,
                                                        uint32_t* __restrict__ hist,  // 3 x kHistBins
                                                        const SelState* __restrict__ st,
                                                        uint32_t* __restrict__ strag_count,
                                                        uint32_t* __restrict__ ticket,  // [0] groups done, [1 + g] blocks of group g done
                                                        double* __restrict__ partials,
                                                        double* __restrict__ gpartials,
                                                        double* __restrict__ out /* 32 doubles */,
                                                        float* __restrict__ chk_hist,
                                                        lsgpu_iter_trace* __restrict__ trace, int trace_cap,
                                                        int capped_launch, int fuse_update,
                                                        uint32_t* __restrict__ sel_aux,
                                                        uint32_t* __restrict__ hist3w /* committed select: window table */,
                                                        int committed,
                                                        uint32_t* __restrict__ spread_cnt /* front rows of k_knn_tile (nullable) */,
                                                        int predicted /* the search of this iteration counted against the select's window if IcpState::sel_mode */,
                                                        uint32_t* __restrict__ amb_cnt, uint2* __restrict__ amb_key /* kSelAmbCap x {query, distance bits} */,
                                                        double* __restrict__ amb_val /* kSelAmbCap x 32 */,
                                                        int amb_cap /* <= kSelAmbCap: a slice with more distances is summed in place */,
                                                        int two_pass /* the select stopped after its second pass: the limit's slice is known, the limit is not */) {
  __shared__ uint32_t sc[260];
  __shared__ double fin[32];
  __shared__ double red[8][33];
  __shared__ int is_last;
  if (ist->done) return;
  // fused select (IcpState::sel_wide, lsgpu_common.hip.h): the search kernel left {distances below the window, slice
  // histogram}; no select kernel has run (committed) or they all exited at once (predicted and armed)
  const bool wide = ist->sel_wide && amb_cnt;
  const bool fused = wide && (committed || (predicted && ist->sel_mode));
  const bool plain2 = wide && !fused && two_pass;
  const bool resolve = fused || plain2;   // the limit is one of the distances set aside: the last block finds it
  uint32_t s_slice = 0xFFFFFFFFu;   // the slice whose distances are set aside (none: 0xFFFFFFFF -- no distance is that large)
  uint32_t f_krem2 = 0u, f_cnt2 = 0u;   // rank of the limit inside its slice (fused), distances in the slice
#ifdef LSGPU_KNN_STATS
  const long long c0 = clock64();
  if (threadIdx.x == 0) atomicMin(&g_ne_dbg[5], (unsigned long long)wall_clock64());
#endif
  Mat34 T;
#pragma unroll
  for (int i = 0; i < 12; ++i) T.m[i] = ist->T_rows[i];
  float limit;
  bool sel_ok = true;
  if (fused) {
    // every block: the slice that holds the order statistic, the rank inside it, how many distances it holds -- one batch
    // of loads (the thread's 16 slices, the counters of the distances below the window), one scan
    __shared__ uint32_t wsum[4], fres[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint4* h4 = reinterpret_cast<const uint4*>(hist + kHistBins) + t * (kSelSlices / 256 / 4);
    static_assert(kSelSlices == 256 * 16, "16 slices per thread");
    uint4 c4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c4[i] = h4[i];
    uint32_t bel = t < kSelBelowSlots ? sel_aux[t * kSelBelowStride] : 0u;
    const uint32_t k = st[-2].k;  // sel[0] = {0, rank}: constant during an align
    uint32_t c[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[4 * i] = c4[i].x; c[4 * i + 1] = c4[i].y; c[4 * i + 2] = c4[i].z; c[4 * i + 3] = c4[i].w; }
    uint32_t loc = 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) loc += c[i];
    uint32_t incl = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    bel = wave_sum_u32(bel);   // (the counters sit in the first wave's lanes)
    if (lane == 63) wsum[w] = incl;
    if (t == 0) { fres[0] = 0u; fres[1] = 0u; fres[2] = 0u; fres[3] = bel; }
    __syncthreads();
    const uint32_t below = fres[3], inside = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    uint32_t base = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += i < w ? wsum[i] : 0u;
    incl += base;
    const uint32_t excl = incl - loc;
    const bool in_window = k >= below && k - below < inside;
    const uint32_t kk = k - below;
    if (in_window && excl <= kk && kk < incl) {
      uint32_t cum = excl;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (kk < cum + c[i]) { fres[0] = (uint32_t)(t * 16 + i); fres[1] = kk - cum; fres[2] = c[i]; break; }
        cum += c[i];
      }
    }
    __syncthreads();
    limit = 0.f;
    if (!in_window) {
      sel_ok = false;
    } else {
      f_krem2 = fres[1]; f_cnt2 = fres[2];
      s_slice = (ist->sel_lo >> kSelSliceShift) + fres[0];
      if (f_cnt2 > (uint32_t)amb_cap || f_krem2 >= f_cnt2) sel_ok = false;   // (a fuller slice is summed in place: the select runs in full)
    }
  } else if (plain2) {
    // the select's first two passes ran: its second table covers the limit's 12-bit float bin in steps of 2^9, a slice
    // is four of its bins
    const SelState in = st[-1];
    uint32_t bin2, krem;
    find_bin(hist + kHistBins, kHistBins, in.k, &bin2, &krem, sc);
    const uint32_t i0 = bin2 & ~3u;
    uint32_t cnt = 0u, before = 0u;
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) { const uint32_t c = hist[kHistBins + i0 + i]; cnt += c; before += i0 + i < bin2 ? c : 0u; }
    static_assert(kSelSliceShift == 11, "a slice = four bins of the second pass");
    s_slice = (in.prefix << 9) | (bin2 >> 2);
    f_cnt2 = cnt; f_krem2 = krem + before;
    limit = 0.f;
    if (cnt > (uint32_t)amb_cap) sel_ok = false;   // (a fuller slice is summed in place: the select runs its third pass)
  } else if (committed) {
    // No select kernel ran: the search kernels left {counts below the last limit's 12-bit bin, the 11-bit histogram
    // inside it, the 9-bit histograms of a window of second-level bins around it}.  Every block derives the order
    // statistic from them (3 short scans); a rank outside the bin or the window voids the iteration.
    __shared__ uint32_t cnt[8];
    uint32_t below = 0, inside = 0;
    for (int i = threadIdx.x; i < kSelBelowSlots; i += 256) below += sel_aux[i * kSelBelowStride];
    for (int i = threadIdx.x; i < kHistBins; i += 256) inside += hist[kHistBins + i];
    below = wave_sum_u32(below); inside = wave_sum_u32(inside);
    if ((threadIdx.x & 63) == 0) { cnt[threadIdx.x >> 6] = below; cnt[4 + (threadIdx.x >> 6)] = inside; }
    __syncthreads();
    below = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    inside = cnt[4] + cnt[5] + cnt[6] + cnt[7];
    __syncthreads();
    const uint32_t k = st[-2].k;  // sel[0] = {0, rank}: constant during an align
    limit = 0.f;
    if (k < below || k - below >= inside) {
      sel_ok = false;
    } else {
      uint32_t bin2, krem2, bin3, krem3;
      find_bin(hist + kHistBins, kHistBins, k - below, &bin2, &krem2, sc);
      const uint32_t d = bin2 - ist->sel_bin2 + (uint32_t)kSelWinHalf;
      if (d >= (uint32_t)kSelWinRows) {
        sel_ok = false;
      } else {
        find_bin(hist3w + d * 512u, 512, krem2, &bin3, &krem3, sc);
        limit = __uint_as_float((ist->sel_bin1 << 20) | (bin2 << 9) | bin3);
      }
    }
  } else {
    limit = select_limit(hist + 2 * kHistBins, st, sc);
    if (wide) {
      // the same sum as the fused select's: the inliers of the limit's slice are set aside and added in query order --
      // if the slice holds no more than the fused path could have set aside (its count: four bins of the select's
      // second table, which covers the limit's 12-bit float bin in steps of 2^9)
      const uint32_t lb = __float_as_uint(limit), sl = lb >> kSelSliceShift;
      const uint32_t i0 = (sl << (kSelSliceShift - 9)) & (uint32_t)(kHistBins - 1);
      uint32_t cnt = 0u;
      for (uint32_t i = 0; i < (1u << (kSelSliceShift - 9)); ++i) cnt += hist[kHistBins + i0 + i];
      if (cnt <= (uint32_t)amb_cap) { s_slice = sl; f_cnt2 = cnt; }   // (f_cnt2: at most this many will be set aside)
    }
  }
#ifdef LSGPU_KNN_STATS
  const long long c1 = clock64();
  if (threadIdx.x == 0) atomicMin(&g_ne_dbg[10], (unsigned long long)wall_clock64());
#endif
  double acc[kNe];
#pragma unroll
  for (int k = 0; k < kNe; ++k) acc[k] = 0.0;
  // four points per step: their loads (distance, match, query, gathered normal) are issued together, the sums
  // are still taken in index order, so the result does not depend on the unrolling
  const int stride = gridDim.x * 256;
  for (int j0 = blockIdx.x * 256 + threadIdx.x; j0 < (sel_ok ? nq : 0); j0 += kNeUnroll * stride) {
    float dd[kNeUnroll]; float4 qq[kNeUnroll], rr[kNeUnroll], nn[kNeUnroll]; bool use[kNeUnroll], amb[kNeUnroll];
#pragma unroll
    for (int u = 0; u < kNeUnroll; ++u) {
      const int j = j0 + u * stride;
      use[u] = j < nq;
      dd[u] = use[u] ? d2[j] : INFINITY;
      {
        const uint32_t sl = __float_as_uint(dd[u]) >> kSelSliceShift;
        const bool in_s = sl == s_slice;
        // fused: below the limit's slice = an inlier, inside it = set aside (the exact limit is one of those);
        // otherwise the limit is known and only the slice's inliers are set aside
        const bool inl = resolve ? sl <= s_slice : dd[u] <= limit;
        amb[u] = use[u] && inl && in_s;
        use[u] = use[u] && inl;
      }
      qq[u] = use[u] ? match[j] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
      rr[u] = use[u] ? rdq[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      use[u] = use[u] && __float_as_int(qq[u].w) >= 0;
    }
#pragma unroll
    // (the normal is gathered: the 16 MB normal array stays in the L2s / Infinity Cache over an alignment, and a copy of
    // the match's normal kept next to the match -- coalesced 16 B per query, written by the search kernels -- measured
    // SLOWER: 31.5 -> 34.6 us per launch, profiles/r03b_bench.json; it adds 16 MB of HBM stream to save cache hits)
    for (int u = 0; u < kNeUnroll; ++u) nn[u] = use[u] ? nrm[__float_as_int(qq[u].w)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < kNeUnroll; ++u) {
    if (!use[u]) continue;
    const float4 q = qq[u];
    const float4 r = rr[u];
    const float3 p = xform(T, r.x, r.y, r.z);
    const float4 n = nn[u];
    float J[6];
    J[0] = p.y * n.z - p.z * n.y;
    J[1] = p.z * n.x - p.x * n.z;
    J[2] = p.x * n.y - p.y * n.x;
    J[3] = n.x; J[4] = n.y; J[5] = n.z;
    const float res = (p.x - q.x) * n.x + (p.y - q.y) * n.y + (p.z - q.z) * n.z;
    if (amb[u]) {   // its contribution waits for the exact limit (the last block adds it, or not)
      const uint32_t slot = __hip_atomic_fetch_add(amb_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot < (uint32_t)kSelAmbCap) {
        double* v = amb_val + (size_t)slot * 32;
        int kk = 0;
        for (int a = 0; a < 6; ++a)
          for (int c = a; c < 6; ++c) __hip_atomic_store(&v[kk++], (double)J[a] * (double)J[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int a = 0; a < 6; ++a) __hip_atomic_store(&v[21 + a], -((double)J[a] * (double)res), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&v[27], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&v[28], (double)res * (double)res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(&amb_key[slot]),
                           ((unsigned long long)__float_as_uint(dd[u]) << 32) | (unsigned long long)(uint32_t)(j0 + u * stride),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      continue;
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = a; c < 6; ++c) acc[k++] += (double)J[a] * (double)J[c];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] -= (double)J[a] * (double)res;
    acc[27] += 1.0;
    acc[28] += (double)res * (double)res;
    }
  }
#ifdef LSGPU_KNN_STATS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long c2 = clock64();
  if (threadIdx.x == 0) atomicMax(&g_ne_dbg[6], (unsigned long long)wall_clock64());
#endif
#pragma unroll
  for (int k = 0; k < kNe; ++k) acc[k] = wave_sum(acc[k]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < kNe; ++k) red[w][k] = acc[k];
  }
  __syncthreads();
  // Hand-off between blocks WITHOUT fences: the partial sums are written and read with agent-scope (sc1) accesses,
  // which go past the per-CU L1 and the per-XCD L2 on both sides (MI355X_MICROARCH.md, inter-workgroup visibility:
  // "sc1 stores and loads both sides"); a release / acquire fence pair costs 1.7-6.5 us per block on this chip and
  // every block would pay it.  s_waitcnt vmcnt(0) orders a block's stores before its ticket.
  if (threadIdx.x < kNe)
    __hip_atomic_store(&partials[(size_t)blockIdx.x * 32 + threadIdx.x],
                       ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- publish.  The LAST block to finish (one ticket) sums the block partials in a fixed order -- thread (g, c) adds
  // the rows of slice g of column c in row order, the 8 slice sums are added in slice order -- so the result is
  // bitwise reproducible whichever block ends up doing it; all loads of a thread's slice are in flight together:
  // one ticket and one load round trip after the last block's own loop.  (A two-level variant -- groups of 16 blocks,
  // then the groups -- paid two more dependent agent-scope round trips; measured 24 us from the last loop end to the
  // kernel's end, of which 8 us in the update lane, before this and the LDS staging below.)
  // The fence-free form leans on gfx9 behaviour (stores are counted in vmcnt, sc1 accesses are served by the memory
  // side of the L2): it is compiled for gfx942 / gfx950 only, any other target gets the release / acquire pair of the
  // HIP memory model (tests/test_gpu_parity.py::test_experiment_switches_do_not_change_results compares the two builds digest by digest).
#if (defined(__gfx950__) || defined(__gfx942__)) && !defined(LSGPU_NE_FENCED)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1);
#ifdef LSGPU_KNN_STATS
    const long long c3 = clock64();
    atomicAdd(&g_ne_dbg[1], (unsigned long long)(c1 - c0)); atomicAdd(&g_ne_dbg[2], (unsigned long long)(c2 - c1));
    atomicAdd(&g_ne_dbg[3], (unsigned long long)(c3 - c2)); atomicAdd(&g_ne_dbg[4], 1ull);
#endif
  }
  __syncthreads();
  if (!is_last) return;
#ifdef LSGPU_KNN_STATS
  if (threadIdx.x == 0) g_ne_tail[0] = (unsigned long long)wall_clock64();
#endif
#if !((defined(__gfx950__) || defined(__gfx942__)) && !defined(LSGPU_NE_FENCED))
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  // the loop state and the select's failure flag travel with the same round trip as the partials: the update lane
  // then works on an LDS copy (a dozen dependent global round trips of one lane otherwise)
  __shared__ IcpState st_sh;
  __shared__ uint32_t fail_sh, cnt_sh[2];
  constexpr int kStWords = (int)(sizeof(IcpState) / sizeof(uint32_t));
  uint32_t* sw = reinterpret_cast<uint32_t*>(&st_sh);
  uint32_t* gw = reinterpret_cast<uint32_t*>(ist);
  uint32_t st_word[(kStWords + 255) / 256];
  if (fuse_update) {
#pragma unroll
    for (int i = 0; i < (kStWords + 255) / 256; ++i)
      st_word[i] = (int)threadIdx.x + 256 * i < kStWords ? gw[threadIdx.x + 256 * i] : 0u;
  }
  // the distances set aside: their keys travel with the same round trip as the partials (the whole area, whatever its fill)
  const bool s_on = sel_ok && s_slice != 0xFFFFFFFFu;
  __shared__ __attribute__((aligned(16))) uint32_t akj[kSelAmbCap];
  __shared__ __attribute__((aligned(16))) uint32_t akb[kSelAmbCap];
  __shared__ uint32_t aord[kSelAmbCap];
  __shared__ uint32_t alim_sh, am_sh;
  __shared__ double red2[8][33];
  constexpr int kAmbLds = 64;           // slots whose contributions also travel with that round trip (fuller: one more trip for the rest)
  __shared__ double aval[kAmbLds * 32];
  uint32_t an = 0u;
  unsigned long long akey = 0ull;
  double av[kAmbLds * 32 / 256];
  if (s_on) {
    // (only slots this launch can have written -- the prologue knows how many distances the slice holds; a slot nobody wrote
    // comes from far away, and loads return in order.  Clamped addresses, not predicated loads: a select on a loaded value
    // would wait for it here, in front of the partials' loads)
    an = __hip_atomic_load(amb_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t kmax = f_cnt2 ? f_cnt2 - 1u : 0u, vmax = kmax * 32u + 31u;
    akey = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&amb_key[threadIdx.x < kmax ? threadIdx.x : kmax]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int u = 0; u < kAmbLds * 32 / 256; ++u) {
      const uint32_t i = threadIdx.x + 256u * (uint32_t)u;
      av[u] = __hip_atomic_load(&amb_val[i < vmax ? i : vmax], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  uint32_t fail_word = 0u, ns_word = 0u, nw_word = 0u;
  if (threadIdx.x == 255) {
    if (sel_aux) fail_word = sel_aux[kSelFailFlag];
    ns_word = __hip_atomic_load(strag_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    nw_word = __hip_atomic_load(strag_count + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  {
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;          // 8 slices of rows
    const uint32_t per = (gridDim.x + 7u) / 8u;
    const uint32_t r0 = (uint32_t)slice * per, r1 = r0 + per < gridDim.x ? r0 + per : gridDim.x;
    double t = 0.0;
    uint32_t r = r0;
    for (; r + 16 <= r1; r += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        v[u] = __hip_atomic_load(&partials[(size_t)(r + u) * 32 + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 16; ++u) t += v[u];
    }
    for (; r < r1; ++r)
      t += __hip_atomic_load(&partials[(size_t)r * 32 + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[slice][col] = t;
  }
  if (fuse_update) {
#pragma unroll
    for (int i = 0; i < (kStWords + 255) / 256; ++i)
      if ((int)threadIdx.x + 256 * i < kStWords) sw[threadIdx.x + 256 * i] = st_word[i];
  }
  if (s_on) {
#pragma unroll
    for (int u = 0; u < kAmbLds * 32 / 256; ++u) aval[threadIdx.x + 256 * u] = av[u];
    const uint32_t n_in = an < (uint32_t)kSelAmbCap ? an : (uint32_t)kSelAmbCap;
    akj[threadIdx.x] = threadIdx.x < n_in ? (uint32_t)akey : 0xFFFFFFFFu;            // (slots beyond the fill: keys that
    akb[threadIdx.x] = threadIdx.x < n_in ? (uint32_t)(akey >> 32) : 0xFFFFFFFFu;    //  rank behind everything)
    if (threadIdx.x == 0) { alim_sh = resolve ? 0xFFFFFFFFu : __float_as_uint(limit); am_sh = 0u; }
  }
  __syncthreads();
#ifdef LSGPU_KNN_STATS
  if (threadIdx.x == 0) g_ne_tail[1] = (unsigned long long)wall_clock64();
#endif
  bool amb_ok = true;
  if (s_on) {
    if (resolve) amb_ok = an == f_cnt2;   // (every distance counted into the slice has been seen here)
    const uint32_t n = amb_ok ? (an < (uint32_t)kSelAmbCap ? an : (uint32_t)kSelAmbCap) : 0u;
    const uint32_t mb = akb[threadIdx.x], mj = akj[threadIdx.x];
    // eight keys per step (the padding behind the fill ranks behind every real key)
    if (resolve) {   // the limit is the distance of rank f_krem2 inside its slice
      if (threadIdx.x < n) {
        uint32_t cl = 0u, cle = 0u;
        for (uint32_t e0 = 0; e0 < n; e0 += 8u) {
          const uint4 b0 = *reinterpret_cast<const uint4*>(&akb[e0]), b1 = *reinterpret_cast<const uint4*>(&akb[e0 + 4u]);
          const uint32_t eb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int q = 0; q < 8; ++q) { cl += eb[q] < mb ? 1u : 0u; cle += eb[q] <= mb ? 1u : 0u; }
        }
        if (cl <= f_krem2 && f_krem2 < cle) alim_sh = mb;
      }
      __syncthreads();
    }
    const uint32_t lim_bits = alim_sh;
    amb_ok = amb_ok && lim_bits != 0xFFFFFFFFu;
    limit = __uint_as_float(lim_bits);
    // the slice's inliers in the order of their queries, whatever order they were set aside in: row r of that order is
    // added by group r mod 8, the groups' sums in group order
    if (threadIdx.x < n && mb <= lim_bits) {
      uint32_t jr = 0u;
      for (uint32_t e0 = 0; e0 < n; e0 += 8u) {
        const uint4 b0 = *reinterpret_cast<const uint4*>(&akb[e0]), b1 = *reinterpret_cast<const uint4*>(&akb[e0 + 4u]);
        const uint4 j0 = *reinterpret_cast<const uint4*>(&akj[e0]), j1 = *reinterpret_cast<const uint4*>(&akj[e0 + 4u]);
        const uint32_t eb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, ej[8] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) jr += (eb[q] <= lim_bits && ej[q] < mj) ? 1u : 0u;
      }
      aord[jr] = threadIdx.x;
      atomicAdd(&am_sh, 1u);
    }
    __syncthreads();
    {
      const uint32_t m = am_sh;
      const uint32_t col = threadIdx.x & 31u, grp = threadIdx.x >> 5;
      double t = 0.0;
      for (uint32_t r = grp; r < m; r += 8u) {
        const uint32_t e = aord[r];
        t += e < (uint32_t)kAmbLds ? aval[e * 32u + col]
                                   : __hip_atomic_load(&amb_val[(size_t)e * 32 + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      red2[grp][col] = t;
    }
    __syncthreads();
  }
#ifdef LSGPU_KNN_STATS
  if (threadIdx.x == 0) g_ne_tail[2] = (unsigned long long)wall_clock64();
#endif
  if (threadIdx.x == 255) {
    fail_sh = fail_word | ((sel_ok && amb_ok) ? 0u : 1u);  // (a missed committed / fused select: same handling as a missed prediction)
    cnt_sh[0] = ns_word; cnt_sh[1] = nw_word;
  }
  __syncthreads();
  if (threadIdx.x < kNe) {
    double t = red[0][threadIdx.x];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][threadIdx.x];
    if (s_on) {
#pragma unroll
      for (int g = 0; g < 8; ++g) t += red2[g][threadIdx.x];
    }
    out[threadIdx.x] = t; fin[threadIdx.x] = t;
  }
  if (threadIdx.x == 32) {
    if (wide) __hip_atomic_store(amb_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double ns = (double)cnt_sh[0];
    out[29] = (double)limit; fin[29] = (double)limit;
    out[30] = ns; fin[30] = ns;
    const double nw = (double)cnt_sh[1];
#ifdef LSGPU_KNN_STATS
    const double nw_dbg = s_on ? (double)an : nw;   // (stats build: the trace's `searching` field shows how many distances were set aside)
    out[31] = nw_dbg; fin[31] = nw_dbg;
#else
    out[31] = nw; fin[31] = nw;  // queries that had to search in this iteration (k_knn_classify), for the trace
#endif
    __hip_atomic_store(strag_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(strag_count + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // work-list length (k_knn_classify)
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (spread_cnt) {  // tiles found spread in this iteration's search join the front rows from the next launch on
      const uint32_t n = __hip_atomic_load(spread_cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t nc = n < (uint32_t)kFrontMax ? n : (uint32_t)kFrontMax;
      __hip_atomic_store(spread_cnt, nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (fuse_update) st_sh.n_spread = nc; else ist->n_spread = nc;
    }
  }
#ifdef LSGPU_KNN_STATS
  unsigned long long w_pre = 0;
  if (threadIdx.x == 0) w_pre = (unsigned long long)wall_clock64();
#endif
  __syncthreads();
  // wave 0's first lane advances the loop state; the other waves re-arm the iteration's scratch meanwhile (every
  // block has read hist3 / the window table by now)
  if (threadIdx.x == 0) {
    if (fuse_update) icp_update_lane(&st_sh, fin, chk_hist, trace, trace_cap, capped_launch, nullptr, (int)fail_sh);
    else if (sel_aux && fail_sh) sel_aux[kSelFailFlag] = 1u;   // the stand-alone update kernel reads it there
  } else if (threadIdx.x >= 64) {
    const int t = (int)threadIdx.x - 64;
    for (int i = t; i < 3 * kHistBins; i += 192) hist[i] = 0u;
    if (sel_aux && t < kSelBelowSlots) sel_aux[t * kSelBelowStride] = 0u;
    if (sel_aux && fuse_update && t == kSelBelowSlots) sel_aux[kSelFailFlag] = 0u;
    if (hist3w) {  // the window table of the committed select (written only by launches that carry it)
      uint4* w4 = reinterpret_cast<uint4*>(hist3w);
      for (int i = t; i < kSelWinRows * 512 / 4; i += 192) w4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  if (fuse_update) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (kStWords + 255) / 256; ++i)
      if ((int)threadIdx.x + 256 * i < kStWords) gw[threadIdx.x + 256 * i] = sw[threadIdx.x + 256 * i];
  }
#ifdef LSGPU_KNN_STATS
  if (threadIdx.x == 0) {
    const unsigned long long wend = (unsigned long long)wall_clock64();
    const unsigned long long w0 = g_ne_dbg[5], w2 = g_ne_dbg[6], w1 = g_ne_dbg[10];
    g_ne_dbg[0] += 1ull; g_ne_dbg[7] += w2 - w0; g_ne_dbg[8] += wend - w2; g_ne_dbg[9] += w1 - w0;
    g_ne_dbg[11] += wend - w_pre;  // the update lane alone
    g_ne_dbg[16] += g_ne_tail[0] - w2; g_ne_dbg[17] += g_ne_tail[1] - g_ne_tail[0]; g_ne_dbg[18] += g_ne_tail[2] - g_ne_tail[1]; g_ne_dbg[19] += w_pre - g_ne_tail[2];
    g_ne_dbg[5] = ~0ull; g_ne_dbg[6] = 0ull; g_ne_dbg[10] = ~0ull;
  }
#endif
}

// 1024 threads = 32 groups of 32: group r sums rows r, r+32, ... of its column, then the 32 group
// sums are added in fixed order => deterministic, and ~30x faster than one thread per column.
__global__ __launch_bounds__(1024) void k_ne_final(const double* __restrict__ partials, int nblocks,
                                                   double* __restrict__ out) {
  __shared__ double sh[32][33];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  double s = 0.0;
  if (col < kNe)
    for (int b = grp; b < nblocks; b += 32) s += partials[(size_t)b * 32 + col];
  sh[grp][col] = s;
  __syncthreads();
  if (threadIdx.x < kNe) {
    double t = 0.0;
    for (int r = 0; r < 32; ++r) t += sh[r][threadIdx.x];
    out[threadIdx.x] = t;
  }
}

}  // namespace lsgpu
