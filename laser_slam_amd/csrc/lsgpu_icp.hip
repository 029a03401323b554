// lsgpu_icp.hip -- C-ABI shim (include/lsgpu_icp.h) over the gfx950 kernels.
//
// Host-side control flow restates PointMatcher::ICP::compute as called from
// laser_slam/src/laser_track.cpp:496 and laser_slam/src/incremental_estimator.cpp:108:
//   set_reference : steps 2-3 (centre on mean, matcher init)
//   align         : steps 5-7 (move reading by T_refMean_dataIn, iterate, compose)
// Kernels: lsgpu_grid.hip.h, lsgpu_knn.hip.h, lsgpu_solve.hip.h.  No CPU fallback exists: every failure is returned to the caller.
#include <cstring>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <chrono>
#include <functional>
#include <system_error>
#include <thread>

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is opened lazily (dlopen) by lsgpu_icp_comm_init
#ifdef LSGPU_EXPERIMENTS
#include <rocprim/device/device_radix_sort.hpp>   // experiments build only, behind LSGPU_ROCPRIM_SORT: the library sort as a cross-check of lsgpu_sort.hip.h
#endif

#include "../../include/lsgpu_icp.h"
#include "lsgpu_grid.hip.h"
#include "lsgpu_tuning.h"
#include "lsgpu_policy.h"
#include "lsgpu_knn.hip.h"
#include "lsgpu_cone.hip.h"
#ifdef LSGPU_EXPERIMENTS
#include "lsgpu_knn_rows.hip.h"   // measured-slower variants, kept as the record of what was tried
#endif
#include "lsgpu_solve.hip.h"
#include "lsgpu_host_math.h"
#include "lsgpu_ssn.hip.h"
#include "lsgpu_ssn_tree.hip.h"
#include "lsgpu_ssn_select.hip.h"
#include "lsgpu_sort.hip.h"
#include "lsgpu_scan.hip.h"
#include "lsgpu_rand.h"

using namespace lsgpu;

#define HIPC(expr)                                                                      \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      char buf__[256];                                                                  \
      snprintf(buf__, sizeof buf__, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,        \
               hipGetErrorString(e__));                                                 \
      h->err = buf__;                                                                   \
      (void)hipGetLastError();                                                          \
      return LSGPU_HIP_ERROR;                                                           \
    }                                                                                   \
  } while (0)

namespace {

// RCCL entry points, resolved at run time so that liblsgpu_icp.so itself has no RCCL dependency.
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*GroupStart)() = nullptr;           // optional (fused exchange of the committed select)
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
      api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
      api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.lib, "ncclCommAbort"));
      api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
      api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.lib, "ncclGroupStart"));
      api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.lib, "ncclGroupEnd"));
      if (!api.GroupStart || !api.GroupEnd) { api.GroupStart = nullptr; api.GroupEnd = nullptr; }
      if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) api.lib = nullptr;
    }
  }
  return api.lib ? &api : nullptr;
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t want = n + n / 8 + 256;
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

bool is_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// A few words for the host, in ONE launch: the kernel stores into the pinned (host-coherent, device-mapped) staging words
// themselves.  (Each hipMemcpyAsync of four bytes is a 4 - 5 us slot of its own on the stream: four of them behind the
// reference filter, three behind the cell counts -- on chains of 0.8 and 0.25 ms.)
struct ToHost { const uint32_t* src[6]; uint32_t* dst[6]; uint32_t n[6]; };
__global__ __launch_bounds__(64) void k_to_host(ToHost c) {
#pragma unroll
  for (int e = 0; e < 6; ++e)
    for (uint32_t i = threadIdx.x; i < c.n[e]; i += 64u) c.dst[e][i] = c.src[e][i];
}
static void to_host_add(ToHost* c, int* used, const void* src, void* dst, size_t bytes) {
  c->src[*used] = static_cast<const uint32_t*>(src); c->dst[*used] = static_cast<uint32_t*>(dst); c->n[*used] = (uint32_t)(bytes / 4);
  ++*used;
}

// (Uploading a PINNED cloud with a copy kernel over its device mapping instead of hipMemcpyAsync was measured slower:
// 5.93 vs 5.48 ms per 1 M-point compute from host buffers; what had made the pinned path the slower one was two copies
// sharing the link -- lsgpu_icp_compute now queues the reading's copy behind the reference's.)

Mat34 to_mat34(const float* Tcm) {  // column-major 4x4 -> rows
  Mat34 m;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) m.m[r * 4 + c] = Tcm[c * 4 + r];
  return m;
}

double wall_ms() {
  return std::chrono::duration<double, std::milli>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline int nblk(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

struct lsgpu_icp;
static int wait_stream(lsgpu_icp* h);

namespace {

}  // namespace

struct lsgpu_icp {
  lsgpu_icp_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;   // lsgpu_icp_compute: the reading's H2D, overlapped with the reference filter
  float4* upload_into = nullptr;       // lsgpu_icp_compute_clouds_upload: that H2D goes into the reading's slot instead of the staging buffer
  bool upload_done = false;            // ... and has been enqueued (every return path of lsgpu_icp_compute drains the copy stream)
  hipEvent_t copy_done = nullptr;
  hipEvent_t ref_up_done = nullptr;    // the reference's H2D on `stream`: the reading's copy queues behind it
  std::string err;

  // reference (steps 2-3)
  int64_t nr = 0;
  float mean[3] = {0, 0, 0};
  GridDev grid;
  int ls = 0;  // start level of the main kNN pass
  lsgpu_icp_info info;
  DevBuf<float4> ref_in;   DevBuf<float> nrm_in;
  // Scratch of the sorts and scans.  Two sets: lsgpu_icp_compute orders the queries on a second stream while the
  // reference grid is built on the first (`sc` / `cur` = the set and the stream the helpers below enqueue on).
  struct SortScratch {
    DevBuf<uint64_t> keys, keys_alt;
    DevBuf<uint32_t> vals, vals_alt;
    DevBuf<char> sort_tmp;
    DevBuf<uint32_t> sort_hist;  // radix sort: 256 x blocks digit histograms + 256 digit totals
    void release() { keys.release(); keys_alt.release(); vals.release(); vals_alt.release(); sort_tmp.release(); sort_hist.release(); }
  };
  SortScratch scr_main, scr_side;
  SortScratch* sc = &scr_main;
  hipStream_t cur = nullptr;           // == stream except while lsgpu_icp_compute enqueues its side work
  hipStream_t draw_stream = nullptr;   // H2D of the filters' draws, issued by the helper thread that produces them
  hipEvent_t draws_done = nullptr, draws_first_done = nullptr;
  hipStream_t side_stream = nullptr;   // lsgpu_icp_compute: reading filter + query order, beside the grid build
  hipEvent_t side_done = nullptr;
  int side_totals_slot = 0;            // scan_totals staging: the side path uses its own words of h_pinned
  std::function<int()> hook_before_ref_sync, hook_after_grid;   // set by lsgpu_icp_compute: before set_reference waits for its cell counts / once the whole grid build is enqueued
  const float* prepared_rd = nullptr;  // queries already ordered by the side path (consumed by the next align)
  int64_t prepared_nq = 0;
  DevBuf<float4> pts, nrm;
  DevBuf<uint32_t> ref_inv;
  DevBuf<HashEntry> tables;
  DevBuf<uint32_t> flags, cidx, bounds;
  DevBuf<ChunkDesc> chunks, chunk_groups;
  DevBuf<float> soa;            // chunk-blocked SoA copy of pts (k_soa_fill)
  DevBuf<uint32_t> soa_base, soa_cnt4, soa_first;
  uint32_t nchunks = 0;
  // direction index of the reference (lsgpu_cone.hip.h): the settled launches of an align search it instead of the voxel grid
  DevBuf<float> cone_soa;
  DevBuf<uint32_t> cone_tab, cone_map;
  DevBuf<float4> cone_rowz;
  ConeDev cone;
  bool cone_ok = false;       // built (or being built on the side stream: cone_pending) for the current reference
  bool defer_cone = false;    // lsgpu_icp_compute: set_reference leaves the build to the side stream
  bool cone_build_in_align = false;   // ... and the next align enqueues it there behind its first iteration
  bool cone_pending = false;  // the loop's stream has not yet waited for cone_done
  hipEvent_t cone_done = nullptr;
  DevBuf<uint32_t> cone_occ;          // occupied (row, column) bins of the index
  hipEvent_t cone_occ_ready = nullptr;
  bool cone_decided = false;          // ... looked at (per reference): cone_dense says whether the reference is too dense in direction
  bool cone_dense = false;
  float cone_occupancy = 0.f;         // points per occupied bin
  float cone_zeta_lo = 0.f, cone_zeta_hi = 0.f;
  bool cone_origin_inside = false;
  uint32_t* h_cone_occ = nullptr;     // pinned word the build's occupancy count travels to
  hipEvent_t price_ready = nullptr;   // the price check's two counters are on the host
  DevBuf<uint32_t> price_cnt;         // kPriceSlots x kPriceStride words (lsgpu_knn.hip.h: ConePrice::count)
  uint32_t* h_price = nullptr;        // ... and their pinned copy
  // launch policy of the running align (lsgpu_policy.h): every decision about what is enqueued next lives there
  policy::Config pol_cfg;
  policy::State pol;
  // is the index paying on this handle's clouds?  (policy::index_not_paying; two event pairs per alignment)
  hipEvent_t ev_pay[4] = {nullptr, nullptr, nullptr, nullptr};
  bool pay_voxel_timed = false, pay_index_timed = false;
  int index_rest = 0;         // alignments that still leave the index alone
  int64_t index_rest_nr = 0;  // ... decided on a reference of this size (another size: the judgement starts over)
  float pay_voxel_us = 0.f, pay_index_us = 0.f;   // the two timings behind it (lsgpu_icp_get_policy_info)
  int ssn_sort_fallbacks = 0, ssn_calls = 0;
  DevBuf<uint4> knn_dbg_wave;
  DevBuf<unsigned long long> knn_dbg;  // LSGPU_KNN_STATS builds: 8 counters
  DevBuf<uint2> cell_cache;  // ntiles x 64
  DevBuf<ulonglong2> cell_tags;
  uint32_t cache_gen = 1;
  ncclComm_t comm = nullptr;  // split-scan mode (lsgpu_icp_comm_init)
  int comm_rank = 0, comm_size = 1;
  DevBuf<long long> comm_tmp;
  int dbg_launch_no = 0;     // kNN launches since the last prepare_queries (stats build ablations)
  DevBuf<float> lb;          // per-query lower bound on the NN distance
  DevBuf<IcpState> state;    // loop state of the running align (device)
  DevBuf<float> chk_hist;    // checker history: 8 floats x (max_iterations + 2)
  DevBuf<lsgpu_iter_trace> trace_dev;
  DevBuf<float4> prev;       // warm start of every query: its current match {xyz, sorted index}
  DevBuf<RefStats> stat_partials;
  DevBuf<GeomDev> geom;      // grid geometry derived on the device (k_ref_stats_final)
  DevBuf<uint32_t> counters;  // [0..16] cell counts, [32] straggler count
  DevBuf<uint32_t> ang_cells; // angular occupancy of the reading (query order decision)
  uint32_t n_spread_host = 0;   // length of the spread list as of the last state fetch
  bool n_spread_known = false;
  DevBuf<uint32_t> spread_flag, spread_list, spread_cnt;  // front rows of the tile kernel (tiles whose queries share no candidates)
  DevBuf<uint32_t> sel_aux;   // predicted select: kSelBelowSlots counters + failure flag
  DevBuf<uint32_t> sel_win;   // committed select (split-scan mode): kSelWinRows x 512 window histogram
  DevBuf<uint2> amb_key;      // fused select: the distances of the limit's slice the normal equations set aside ...
  DevBuf<double> amb_val;     // ... and their contributions (kSelAmbCap x 32)
#ifdef LSGPU_EXPERIMENTS
  DevBuf<uint32_t> work;      // compacted list of searching queries (k_knn_classify -> k_knn_rows)
#endif

  // device filters (lsgpu_ssn.hip.h)
  DevBuf<SsnSeg> ssn_seg_a, ssn_seg_b;
  DevBuf<int> ssn_axis_a, ssn_axis_b;       // per segment: the axis its current order follows (segmented level sorts)
  DevBuf<uint32_t> ssn_seg_fb;
  DevBuf<SegBlock> ssn_blocktab;
  DevBuf<uint32_t> ssn_seg_of, ssn_box_pts, ssn_box_base, ssn_keep, ssn_out_pos, ssn_bb, ssn_bounds_ws;
  DevBuf<float> ssn_box_normal, ssn_draws;
  // sort-free upper levels of the reference filter (lsgpu_ssn_select.hip.h)
  DevBuf<uint32_t> gs_e[2], gs_k[6];          // two buffer sets: points + their three ordered keys
  DevBuf<GsBlock> gs_tab;                     // block tables of all levels (host-computed: segment sizes are static)
  DevBuf<GsSegBlocks> gs_sblk;
  DevBuf<uint32_t> gs_hist;                   // per segment 2 x 256 counters, two levels (ping-pong)
  DevBuf<GsMedian> gs_cand, gs_median;
  DevBuf<uint32_t> gs_cand_blk, gs_cand_n, gs_cl, gs_err;
  DevBuf<uint2> gs_rng;                       // per segment: the key range of its points on its cut axis (two levels)
  uint32_t* h_gs_err = nullptr;               // pinned
  int64_t gs_plan_n = -1;                     // the cloud size / level count the tables on the device were made for
  int gs_plan_levels = -1;
  std::vector<uint32_t> gs_lvl_first, gs_lvl_blocks;
  DevBuf<float4> flt_in, flt_in2, flt_ref, flt_rd;
  DevBuf<float> flt_nrm;
  float* draws_pinned = nullptr;  // host staging of the filter draws (pinned: async H2D)
  std::vector<DevBuf<float4>> clouds;  // lsgpu_cloud_upload slots
  std::vector<int64_t> cloud_n;        // -1: empty
  DevBuf<float4> submap;               // assembled reference of lsgpu_icp_compute_clouds
  size_t draws_pinned_cap = 0;

  // reading
  int64_t nq = 0;
  DevBuf<float4> q_in, rdq;
  DevBuf<int> ids;   DevBuf<float> d2;
  DevBuf<int> ids_io; DevBuf<float> d2_io;
  DevBuf<uint32_t> strag;
  DevBuf<uint32_t> hist;      // 3 * kHistBins
  DevBuf<SelState> sel;       // [0] input rank, [1] after pass 2, [2] after pass 3
  DevBuf<double> ne_partials; // kNeBlocks * 32
  DevBuf<double> ne_gpartials;  // (kNeBlocksMax / kNeGroup) * 32: first-level sums of k_normal_eq_loop
  DevBuf<uint32_t> ne_tickets;  // 1 + kNeBlocksMax / kNeGroup
  DevBuf<double> ne_out;      // 32 (29 + limit slot)
  DevBuf<float> limit_dev;
  double* h_pinned = nullptr; // 64 doubles of pinned host staging
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_state = nullptr;   // lsgpu_icp_align: completion of a loop-state copy (the stream goes on behind it)
  // lsgpu_icp_align may return with ONE more iteration queued on `stream` behind its last look at the loop state (it
  // exits at once: the state says `done`).  Work the next call starts on the handle's OTHER streams is ordered behind
  // it through this event, so that nothing ever depends on what that launch does not touch.
  hipEvent_t ev_tail = nullptr;
  bool tail_pending = false;
  struct KnnEv { hipEvent_t a, b, c, d, e; bool second; };  // before kNN, after the main pass, after the wave-per-query pass (recorded only if one was launched: `second`), after the select, after the normal equations
  std::vector<KnnEv> knn_events;   // pool, reused across aligns
  // split-scan mode with profile_kernels: an event pair around every RCCL call of the loop (stats.t_comm_ms)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> comm_events;
  size_t comm_events_used = 0;
  bool time_comm = false;
  size_t knn_events_used = 0;
  std::vector<lsgpu_iter_trace> trace;
  size_t trace_on_device = 0;   // records of the last alignment still in trace_dev (fetched by lsgpu_icp_get_trace)
  std::vector<std::pair<float, float>> trace_knn_us;   // their search timings (profiled runs), merged in on the fetch
};

// Wait for the handle's stream.  Plain handles block; a handle with a communicator polls with a deadline
// (LSGPU_COMM_TIMEOUT_MS, default 30 s): if a peer rank died inside the loop the collectives never complete, the
// communicator is aborted and the caller gets LSGPU_HIP_ERROR instead of a hang.
static int wait_stream(lsgpu_icp* h) {
  if (!h->comm) { HIPC(hipStreamSynchronize(h->stream)); return LSGPU_OK; }
  const double limit_ms = tuning().comm_timeout_ms;
  const double t0 = wall_ms();
  for (int spin = 0;; ++spin) {
    const hipError_t e = hipStreamQuery(h->stream);
    if (e == hipSuccess) return LSGPU_OK;
    if (e != hipErrorNotReady) { (void)hipGetLastError(); h->err = std::string("stream: ") + hipGetErrorString(e); return LSGPU_HIP_ERROR; }
    (void)hipGetLastError();
    if (wall_ms() - t0 > limit_ms) {
      h->err = "split-scan: a collective did not complete in time (peer rank lost?); communicator aborted";
      RcclApi* api = rccl_api();
      if (api && api->CommAbort) (void)api->CommAbort(h->comm); else if (api) (void)api->CommDestroy(h->comm);
      h->comm = nullptr; h->comm_size = 1; h->comm_rank = 0;
      return LSGPU_HIP_ERROR;
    }
    if (spin > 200) std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

// event pair around a collective (only when the align is profiled)
static void comm_mark(lsgpu_icp* h, bool begin) {
  if (!h->time_comm) return;
  if (begin) {
    if (h->comm_events_used == h->comm_events.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { (void)hipGetLastError(); h->time_comm = false; return; }
      h->comm_events.emplace_back(a, b);
    }
    (void)hipEventRecord(h->comm_events[h->comm_events_used].first, h->stream);
  } else {
    (void)hipEventRecord(h->comm_events[h->comm_events_used++].second, h->stream);
  }
}

static void order_after_tail(lsgpu_icp* h, hipStream_t s) {
  if (h->tail_pending && s && h->ev_tail && hipStreamWaitEvent(s, h->ev_tail, 0) != hipSuccess) (void)hipGetLastError();
}

static constexpr int kNeBlocksMax = 2048;
static const int kNeBlocks = tuning().ne_blocks;   // 64 .. kNeBlocksMax (lsgpu_tuning.h)
static constexpr int kStatBlocks = 512;
static constexpr int kHistBlocks = 256;
static constexpr int kFallbackBlocksSettled = 1024;
static constexpr int kFallbackBlocks = 8192;  // x 4 waves: one query per wave for up to 32 k stragglers, round robin beyond

extern "C" {

void lsgpu_icp_config_yaml(lsgpu_icp_config* c) {  // icp_default.yaml:14-27
  std::memset(c, 0, sizeof(*c));
  c->trim_ratio = 0.75f;
  c->max_iterations = 40;
  c->min_diff_rot = 0.001f;
  c->min_diff_trans = 0.01f;
  c->smooth_length = 4;
  c->cell_size = 0.f;
}

void lsgpu_icp_config_default(lsgpu_icp_config* c) {  // ICP::setDefault(), laser_track.cpp:20
  std::memset(c, 0, sizeof(*c));
  c->trim_ratio = 0.85f;
  c->max_iterations = 40;
  c->min_diff_rot = 0.001f;
  c->min_diff_trans = 0.001f;
  c->smooth_length = 3;
  c->cell_size = 0.f;
}

void lsgpu_chain_config_yaml(lsgpu_chain_config* c) {  // icp_default.yaml:1-7
  std::memset(c, 0, sizeof(*c));
  c->reading_prob = 0.5f; c->ssn_knn = 10; c->ssn_ratio = 0.5f; c->seed = -1;
}

void lsgpu_chain_config_default(lsgpu_chain_config* c) {  // ICP::setDefault(), laser_track.cpp:20
  std::memset(c, 0, sizeof(*c));
  c->reading_prob = 0.75f; c->ssn_knn = 7; c->ssn_ratio = 0.5f; c->seed = -1;
}

int lsgpu_abi_version(void) { return LSGPU_ABI_VERSION; }

const char* lsgpu_strerror(int code) {
  switch (code) {
    case LSGPU_OK: return "ok";
    case LSGPU_NO_CONVERGENCE: return "ICP did not converge (PointMatcher::ConvergenceError)";
    case LSGPU_BAD_CONFIG: return "bad configuration";
    case LSGPU_HIP_ERROR: return "HIP runtime error";
    case LSGPU_BAD_ARG: return "bad argument";
    default: return "unknown";
  }
}

const char* lsgpu_last_error(lsgpu_icp* h) { return h ? h->err.c_str() : "null handle"; }

int lsgpu_icp_create(const lsgpu_icp_config* cfg, int device, lsgpu_icp** out) {
  if (!cfg || !out) return LSGPU_BAD_ARG;
  *out = nullptr;
  if (!(cfg->trim_ratio > 0.f && cfg->trim_ratio <= 1.f) || cfg->max_iterations < 1 ||
      cfg->smooth_length < 1)
    return LSGPU_BAD_CONFIG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    (void)hipGetLastError();
    return LSGPU_HIP_ERROR;  // no silent CPU path: the caller must see that there is no GPU
  }
  lsgpu_icp* h = new lsgpu_icp();
  h->cfg = *cfg;
  h->device = device;
  std::memset(&h->grid, 0, sizeof(h->grid));
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc((void**)&h->h_pinned, 128 * sizeof(double), hipHostMallocDefault) != hipSuccess ||
      hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    (void)hipGetLastError();
    delete h;
    return LSGPU_HIP_ERROR;
  }
  h->cur = h->stream;
  *out = h;
  return LSGPU_OK;
}

void lsgpu_icp_destroy(lsgpu_icp* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm && rccl_api()) (void)rccl_api()->CommDestroy(h->comm);
  h->comm_tmp.release();
  for (auto& c : h->clouds) c.release();
  h->submap.release();
  h->ref_in.release(); h->nrm_in.release(); h->scr_main.release(); h->scr_side.release();
#ifdef LSGPU_EXPERIMENTS
  h->work.release();
#endif
  h->pts.release();
  h->cone_soa.release(); h->cone_occ.release(); h->cone_tab.release(); h->cone_map.release(); h->cone_rowz.release();
  h->nrm.release(); h->ref_inv.release(); h->tables.release(); h->flags.release(); h->cidx.release(); h->bounds.release(); h->chunks.release(); h->chunk_groups.release(); h->soa.release(); h->soa_base.release(); h->soa_cnt4.release(); h->soa_first.release(); h->prev.release(); h->state.release(); h->lb.release(); h->cell_cache.release(); h->cell_tags.release(); h->ssn_seg_a.release(); h->ssn_seg_b.release(); h->ssn_axis_a.release(); h->ssn_axis_b.release(); h->ssn_seg_fb.release(); h->ssn_blocktab.release(); h->ssn_seg_of.release(); h->ssn_box_pts.release(); h->ssn_box_base.release(); h->ssn_keep.release(); h->ssn_out_pos.release(); h->ssn_bb.release(); h->ssn_bounds_ws.release(); h->ssn_box_normal.release(); h->ssn_draws.release(); h->flt_in.release(); h->flt_in2.release(); h->flt_ref.release(); h->flt_rd.release(); h->flt_nrm.release(); h->chk_hist.release(); h->trace_dev.release(); h->knn_dbg.release(); h->knn_dbg_wave.release(); h->stat_partials.release(); h->geom.release();
  h->counters.release(); h->price_cnt.release(); h->ang_cells.release(); h->sel_aux.release(); h->sel_win.release(); h->amb_key.release(); h->amb_val.release(); h->spread_flag.release(); h->spread_list.release(); h->spread_cnt.release(); h->q_in.release(); h->rdq.release(); h->ids.release(); h->d2.release();
  h->ids_io.release(); h->d2_io.release(); h->strag.release(); h->hist.release();
  h->sel.release(); h->ne_partials.release(); h->ne_gpartials.release(); h->ne_tickets.release(); h->ne_out.release(); h->limit_dev.release();
  for (auto& e : h->comm_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : h->knn_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); (void)hipEventDestroy(e.c); (void)hipEventDestroy(e.d); (void)hipEventDestroy(e.e); }
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_state) (void)hipEventDestroy(h->ev_state);
  if (h->ev_tail) (void)hipEventDestroy(h->ev_tail);
  if (h->cone_done) (void)hipEventDestroy(h->cone_done);
  if (h->cone_occ_ready) (void)hipEventDestroy(h->cone_occ_ready);
  for (auto& e : h->ev_pay) if (e) (void)hipEventDestroy(e);
  if (h->h_pinned) (void)hipHostFree(h->h_pinned);
  if (h->h_price) (void)hipHostFree(h->h_price);
  if (h->h_cone_occ) (void)hipHostFree(h->h_cone_occ);
  if (h->h_gs_err) (void)hipHostFree(h->h_gs_err);
  for (auto& bf : h->gs_e) bf.release();
  for (auto& bf : h->gs_k) bf.release();
  h->gs_tab.release(); h->gs_sblk.release(); h->gs_hist.release(); h->gs_cand.release(); h->gs_median.release();
  h->gs_cand_blk.release(); h->gs_cand_n.release(); h->gs_cl.release(); h->gs_err.release(); h->gs_rng.release();
  if (h->draws_pinned) (void)hipHostFree(h->draws_pinned);
  if (h->copy_done) (void)hipEventDestroy(h->copy_done);
  if (h->ref_up_done) (void)hipEventDestroy(h->ref_up_done);
  if (h->draws_done) (void)hipEventDestroy(h->draws_done);
  if (h->draws_first_done) (void)hipEventDestroy(h->draws_first_done);
  if (h->draw_stream) { (void)hipStreamSynchronize(h->draw_stream); (void)hipStreamDestroy(h->draw_stream); }
  if (h->side_done) (void)hipEventDestroy(h->side_done);
  if (h->side_stream) { (void)hipStreamSynchronize(h->side_stream); (void)hipStreamDestroy(h->side_stream); }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

}  // extern "C"

// ---------------------------------------------------------------- internals

static int scan_u32(lsgpu_icp* h, const uint32_t* in, uint32_t* out, size_t n, bool inclusive = false, bool nonzero = false);

template <int ITEMS>
static void radix_pass(lsgpu_icp* h, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, int64_t n,
                       int shift, uint32_t mask, int nblocks) {
  uint32_t* bh = h->sc->sort_hist.p;
  uint32_t* dtot = bh + (size_t)256 * nblocks;
  // (the scan as the tail of k_rs_hist -- last block, ticket -- was measured 2 % slower end to end than this launch)
  hipLaunchKernelGGL(k_rs_hist<ITEMS>, dim3(nblocks), dim3(256), 0, h->cur, kin, n, shift, mask, bh, nblocks);
  hipLaunchKernelGGL(k_rs_scan, dim3(256), dim3(256), 0, h->cur, bh, nblocks, dtot);
  hipLaunchKernelGGL(k_rs_scatter<ITEMS>, dim3(nblocks), dim3(256), 0, h->cur, kin, vin, kout, vout, n, shift, mask,
                     bh, dtot, nblocks);
}

// Stable sort of (h->sc->keys, h->sc->vals)[0..n) by the low `nbits` key bits; the result is in h->sc->keys_alt / h->sc->vals_alt
// (callers re-read those members: the two buffer pairs may have changed places).
static int sort_pairs(lsgpu_icp* h, int64_t n, int nbits) {
  HIPC(h->sc->keys_alt.reserve(n));
  HIPC(h->sc->vals_alt.reserve(n));
#ifdef LSGPU_EXPERIMENTS
  const bool lib_sort = tuning().rocprim_sort;
#else
  const bool lib_sort = false;
#endif
  if (!lib_sort) {   // own radix sort (lsgpu_sort.hip.h)
    const int items_env = tuning().sort_items;
    const int items = items_env ? items_env : n >= (1 << 21) ? 16 : n >= (1 << 19) ? 8 : 4;
    const int nblocks = (int)((n + 256 * items - 1) / (256 * items));
    HIPC(h->sc->sort_hist.reserve((size_t)256 * nblocks + 256));
    const int passes = std::max(1, (nbits + 7) / 8);   // (no key bits: one pass over an all-zero digit = a stable copy)
    uint64_t *kin = h->sc->keys.p, *kout = h->sc->keys_alt.p;
    uint32_t *vin = h->sc->vals.p, *vout = h->sc->vals_alt.p;
    for (int p = 0; p < passes; ++p) {
      const int shift = 8 * p, width = std::max(0, std::min(8, nbits - shift));
      const uint32_t mask = (1u << width) - 1u;
      if (items == 16) radix_pass<16>(h, kin, vin, kout, vout, n, shift, mask, nblocks);
      else if (items == 8) radix_pass<8>(h, kin, vin, kout, vout, n, shift, mask, nblocks);
      else radix_pass<4>(h, kin, vin, kout, vout, n, shift, mask, nblocks);
      std::swap(kin, kout); std::swap(vin, vout);
    }
    HIPC(hipGetLastError());
    if ((passes & 1) == 0) { std::swap(h->sc->keys, h->sc->keys_alt); std::swap(h->sc->vals, h->sc->vals_alt); }  // result is in `keys`
    return LSGPU_OK;
  }
#ifdef LSGPU_EXPERIMENTS
  size_t bytes = 0;
  HIPC(rocprim::radix_sort_pairs(nullptr, bytes, h->sc->keys.p, h->sc->keys_alt.p, h->sc->vals.p,
                                 h->sc->vals_alt.p, (size_t)n, 0, nbits, h->cur));
  HIPC(h->sc->sort_tmp.reserve(bytes));
  bytes = h->sc->sort_tmp.cap;
  HIPC(rocprim::radix_sort_pairs((void*)h->sc->sort_tmp.p, bytes, h->sc->keys.p, h->sc->keys_alt.p, h->sc->vals.p,
                                 h->sc->vals_alt.p, (size_t)n, 0, nbits, h->cur));
#endif
  return LSGPU_OK;  // sorted: keys_alt / vals_alt
}

// Stage a cloud on the device: returns a device pointer valid on h->stream.
static int stage_points(lsgpu_icp* h, const float* src, int64_t n, DevBuf<float4>& buf,
                        const float4** out) {
  if (is_device_ptr(src)) { *out = reinterpret_cast<const float4*>(src); return LSGPU_OK; }
  HIPC(buf.reserve(n));
  HIPC(hipMemcpyAsync(buf.p, src, (size_t)n * 16, hipMemcpyHostToDevice, h->cur));
  *out = buf.p;
  return LSGPU_OK;
}

static int ensure_loop_buffers(lsgpu_icp* h, int64_t nq) {
  HIPC(h->ids.reserve(nq));
  HIPC(h->d2.reserve(nq));
  HIPC(h->strag.reserve(nq));
  HIPC(h->hist.reserve(3 * kHistBins));
  HIPC(h->sel.reserve(4));
  HIPC(h->ne_partials.reserve((size_t)kNeBlocksMax * 32));
  HIPC(h->ne_gpartials.reserve((size_t)(kNeBlocksMax / kNeGroup + 1) * 32));
  HIPC(h->ne_tickets.reserve((size_t)(kNeBlocksMax / kNeGroup + 2)));
  HIPC(h->ne_out.reserve(32));
  HIPC(h->limit_dev.reserve(4));
  HIPC(h->counters.reserve(64));
  return LSGPU_OK;
}

// Sort the reading coarsely (own frame), move it by T (rows) -> h->rdq (w = caller index).
// (gather == false: everything but the last kernel, which needs T -- lsgpu_icp_compute launches it once the reference's
// mean is known)
static int prepare_queries(lsgpu_icp* h, const float* q_xyz1, int64_t nq, const Mat34& T, bool gather = true) {
  const float4* src = nullptr;
  int rc = stage_points(h, q_xyz1, nq, h->q_in, &src);
  if (rc) return rc;
  HIPC(h->sc->keys.reserve(nq));
  HIPC(h->sc->vals.reserve(nq));
  HIPC(h->rdq.reserve(nq));
  HIPC(h->prev.reserve(nq));
  HIPC(h->lb.reserve(nq));
#ifdef LSGPU_EXPERIMENTS
  HIPC(h->work.reserve(nq));
#endif
  // order of the queries inside the waves: chosen on the device from the cloud's angular sampling density
  const int qorder = tuning().query_order;   // -1: automatic
  const float qelev = tuning().q_elev, qsect = tuning().q_sect;   // 0: automatic
  HIPC(h->ang_cells.reserve(kDecCells + 8));
  HIPC(hipMemsetAsync(h->ang_cells.p, 0, (kDecCells + 8) * sizeof(uint32_t), h->cur));
  if (qorder != 0) hipLaunchKernelGGL(k_query_ang_hist, dim3(nblk(nq)), dim3(256), 0, h->cur, src, nq, h->ang_cells.p);
  hipLaunchKernelGGL(k_query_order, dim3(1), dim3(1024), 0, h->cur, h->ang_cells.p, qorder, qelev, qsect);
  hipLaunchKernelGGL(k_query_keys, dim3(nblk(nq)), dim3(256), 0, h->cur, src, nq, h->sc->keys.p, h->sc->vals.p,
                     h->ang_cells.p + kDecCells + 2);
  rc = sort_pairs(h, nq, 48);
  if (rc) return rc;
  if (gather) {
    hipLaunchKernelGGL(k_query_gather, dim3(nblk(nq)), dim3(256), 0, h->cur, src, nq,
                       h->sc->vals_alt.p, T, h->rdq.p);
    HIPC(hipGetLastError());
  }
  h->nq = nq;
  {  // per-tile probe cache: new generation, tags cleared when (re)allocated
    const size_t nt = (size_t)((nq + 63) / 64);
    const bool grow = nt > h->cell_tags.cap;
    HIPC(h->cell_cache.reserve(nt * 64));
    HIPC(h->cell_tags.reserve(nt));
    if (grow) HIPC(hipMemsetAsync(h->cell_tags.p, 0, h->cell_tags.cap * sizeof(ulonglong2), h->cur));
    if (++h->cache_gen == 0) h->cache_gen = 1;
  }
  h->dbg_launch_no = 0;
  return ensure_loop_buffers(h, nq);
}

static int run_select(lsgpu_icp* h, const float* d2, int n, uint32_t k, bool zero_hist, const IcpState* st,
                      bool use_comm, bool predicted, int passes = 3);

static float price_share(const lsgpu_icp* h) {   // heavy lanes / searching lanes of the priced launch (its counters are on the host)
  uint64_t heavy = 0, searching = 0;
  for (int i = 0; i < kPriceSlots; ++i) { heavy += h->h_price[i * kPriceStride]; searching += h->h_price[i * kPriceStride + 1]; }
  return searching ? (float)((double)heavy / (double)searching) : 0.f;
}

static KnnArgs knn_args(lsgpu_icp* h, const Mat34& T) {
  KnnArgs a;
  a.rdq = h->rdq.p; a.nq = (int)h->nq; a.T = T; a.g = h->grid; a.pts = h->pts.p;
  a.chunks = h->chunks.p; a.cgroups = h->chunk_groups.p; a.soa = reinterpret_cast<const float4*>(h->soa.p); a.chunk_soa = h->soa_base.p; a.ids = h->ids.p; a.d2 = h->d2.p; a.prev = h->prev.p;
  a.strag = h->strag.p; a.strag_count = h->counters.p + 32;
  a.r_cap = 1.0f; a.group_r = 0.75f; a.cap2 = INFINITY; a.st = nullptr; a.use_state_cap = 0; a.lb = nullptr;
  a.spread_route_r = 0.f; a.route_chunks = 1 << 30; a.route_dense = 1 << 30; a.route_heavy_max = -1; a.sel_hist2 = nullptr; a.sel_below = nullptr;
  a.sel_hist3w = nullptr; a.sel_force = 0; a.write_all = 1; a.spread_flag = nullptr; a.spread_list = nullptr; a.spread_cnt = nullptr; a.front_blocks = 0;
  a.price.count = nullptr;
  a.gap = tuning().gap;
  a.ntiles = (int)((h->nq + 63) / 64); a.pad_index = (int)h->nr;
  a.chunk_budget = tuning().chunk_budget;
  a.cell_cache = h->cell_cache.p; a.cell_tags = h->cell_tags.p; a.cache_gen = h->cache_gen;
#ifdef LSGPU_EXPERIMENTS
  a.sparse_lanes = 0; a.xcd_swizzle = 0; a.work = h->work.p; a.work_count = h->counters.p + 34;
#endif
  a.dbg = h->knn_dbg.p;
  a.dbg_wave = h->knn_dbg_wave.p;
  { a.dbg_flags = tuning().knn_dbg;
    if ((a.dbg_flags & (64 | 128 | 256 | 512 | 1024 | 2048)) && h->dbg_launch_no < 6) a.dbg_flags = 0; }  // early-exit ablations from launch 6 on
  return a;
}

// findClosests for the queries in h->rdq moved by T: fills h->ids (sorted-reference index), h->d2.
// seed: the queries have no warm start yet (first iteration of an align, or the kernel-level API).
// findClosests for the queries in h->rdq: fills h->ids (sorted-reference index), h->d2, h->prev.
//   T / st   : the transform comes from the argument (kernel-level API) or from the loop state
//   seed     : the queries have no warm start yet (first iteration, kernel-level API)
//   capped   : search cap from the loop state (exact below cap, see lsgpu_knn.hip.h); otherwise uncapped,
//              followed by the straggler fallback
//   it       : what the launch policy decided for this iteration (lsgpu_policy.h); the kernel-level API passes a seeded,
//              uncapped, wide search outside any loop
static int run_knn(lsgpu_icp* h, const Mat34& T, const IcpState* st, const policy::Iteration& it, bool timed,
                   uint32_t seed_rank = 0xFFFFFFFFu) {
  const bool seed = it.seed, capped = it.capped, wide = it.wide, predicted = it.predicted, committed = it.committed;
  policy::State& pol = h->pol;
  const policy::Config& pc = h->pol_cfg;
  const int nq = (int)h->nq;
  KnnArgs a = knn_args(h, T);
  h->dbg_launch_no++;
  a.st = st;
  a.lb = st ? h->lb.p : nullptr;
  a.use_state_cap = capped ? 1 : 0;
  a.write_all = (seed || !capped || !st) ? 1 : 0;
  if (capped) a.r_cap = INFINITY;  // capped balls are never larger than the cap: no straggler by radius
  // `wide`: the balls may still be large (first iterations of an align, retries, kernel-level API): spread
  // waves with wide balls go to the wave-per-query pass, which is launched after the tile kernel
  const Tuning& tn = tuning();
  // settled launches (capped, balls already small): EVERY spread wave hands its lanes on -- 64 divergent per-lane
  // searches held single waves for 190 k cycles, the tail of a 46 k-cycle launch
  const bool settled = capped && !wide && st && tn.route_all;
  a.spread_route_r = wide ? tn.route_r : settled ? 1e-30f : 0.f;
  a.chunk_budget = wide ? tn.chunk_budget_wide : tn.chunk_budget;
#ifdef LSGPU_EXPERIMENTS
  a.sparse_lanes = settled && tn.rowq ? tn.sparse_lanes : 0;
#endif
  // front rows: spread tiles are remembered from the first searches on and searched row-wise by the first workgroups
  // of the settled launches themselves -- no hand-over, no second launch (LSGPU_NO_FRONT: the separate row pass)
  const bool front = tn.front && st && h->spread_cnt.p;
  if (front) {
    a.spread_flag = h->spread_flag.p; a.spread_list = h->spread_list.p; a.spread_cnt = h->spread_cnt.p;
    // (the front is sized from the list's length as the host last saw it -- it travels with the state every few
    // iterations --, from a guess before that: surplus workgroups exit at once, a listed tile beyond the front, or one
    // that turns spread late, searches per lane inside the kernel)
    if (settled) {
      const int tiles = h->n_spread_known ? (int)h->n_spread_host : std::min(tn.front_guess, a.ntiles);
      a.front_blocks = kFrontPerTile * tiles;
      a.spread_route_r = 0.f;
    }
  }
  if (predicted && !wide && capped && st) { a.sel_hist2 = h->hist.p + kHistBins; a.sel_below = h->sel_aux.p; }
  if (committed && a.sel_below) { a.sel_hist3w = (h->comm || !tuning().fused_select) ? h->sel_win.p : nullptr; a.sel_force = 1; }   // (the window table: aligned mode only)
  a.route_chunks = tn.route_chunks;
  a.route_heavy_max = st ? tn.route_heavy_max : -1;   // (the ticket lives in the loop's scratch, re-armed every iteration)
  a.route_dense = tn.route_dense;
  // the search before the first one through the direction index prices the index (lsgpu_knn.hip.h: cone_price)
  const bool pricing = pol.pricing(pc, it, st != nullptr);
  if (pricing) {
    constexpr size_t kPriceBytes = (size_t)kPriceSlots * kPriceStride * sizeof(uint32_t);
    HIPC(h->price_cnt.reserve((size_t)kPriceSlots * kPriceStride));
    if (!h->h_price) HIPC(hipHostMalloc((void**)&h->h_price, kPriceBytes, hipHostMallocDefault));
    HIPC(hipMemsetAsync(h->price_cnt.p, 0, kPriceBytes, h->stream));
    a.price.ox = h->cone.ox; a.price.oy = h->cone.oy; a.price.oz = h->cone.oz; a.price.rs = h->cone.rs; a.price.cs = h->cone.cs;
    a.price.dens4 = (float)(0.25 * (double)h->nr / ((double)h->cone.rows * (double)h->cone.cols));
    a.price.heavy = tn.cone_heavy_steps; a.price.count = h->price_cnt.p;
  }
  if (seed && !st) HIPC(hipMemsetAsync(a.strag_count, 0, sizeof(uint32_t), h->stream));  // (align: k_align_init, then re-armed by k_normal_eq_loop)
  if (seed) hipLaunchKernelGGL(k_knn_seed, dim3(nblk(nq)), dim3(256), 0, h->stream, a);
  if (seed && capped && st && seed_rank != 0xFFFFFFFFu) {
    // first iteration of an align: cap = trim quantile of the seed distances (k_knn_seed left them in d2)
    const int rs = run_select(h, h->d2.p, nq, seed_rank, false /* k_align_init armed the tables */, st, true, false, 2);
    if (rs) return rs;
    hipLaunchKernelGGL(k_seed_cap, dim3(1), dim3(256), 0, h->stream, h->hist.p, h->sel.p + 1, h->state.p);
  }
  // two launches of every alignment are timed for policy::index_not_paying (not in profiled runs: they time everything)
  const bool pay_probe = !timed && st && pol.cone_ok && h->ev_pay[0];
  const bool pay_voxel = pay_probe && !seed && it.ordinal == pc.cone_from - 1 && !h->pay_voxel_timed;
  if (pay_voxel) HIPC(hipEventRecord(h->ev_pay[0], h->stream));
  lsgpu_icp::KnnEv* ev = nullptr;
  if (timed) {
    if (h->knn_events_used == h->knn_events.size()) {
      lsgpu_icp::KnnEv n{};
      HIPC(hipEventCreate(&n.a)); HIPC(hipEventCreate(&n.b)); HIPC(hipEventCreate(&n.c));
      HIPC(hipEventCreate(&n.d)); HIPC(hipEventCreate(&n.e));
      h->knn_events.push_back(n);
    }
    ev = &h->knn_events[h->knn_events_used++];
    ev->second = false;
    HIPC(hipEventRecord(ev->a, h->stream));
  }
  if (pol.wants_occupancy(it, st != nullptr)) {
    // first search through the index for this reference: is it worth it?  The windows of a lane grow with the number of
    // reference points per direction -- measured on local maps of K scans of 1 M points (devtools/cone_density.py), kNN
    // per settled launch: K = 1: 48 us against 82 with the voxel grid, K = 3: 69 / 107, K = 8: 423 / 186.  Points per
    // occupied bin: 2, 3, 8.  (The wait is for a copy queued behind the build; the device is in the first iterations.)
    HIPC(hipEventSynchronize(h->cone_occ_ready));
    const uint32_t occ = *h->h_cone_occ;
    pol.set_occupancy(pc, occ ? (float)((double)h->nr / (double)occ) : 1e9f);
    h->cone_occupancy = pol.cone_occupancy; h->cone_dense = pol.cone_dense; h->cone_decided = true;   // (per reference: a reused reference keeps the verdict)
  }
  if (pol.wants_first_price(it)) {
    // ... and what would this align pay for it?  The search before this one counted the lanes whose windows would be long
    // (wide balls next to O: a wall a metre from the sensor; a sparse reading on a dense map that converges in steps of
    // millimetres): the index evaluates each lane's windows alone, the voxel kernel shares one candidate stream among 64
    // lanes whose balls overlap -- measured on such clouds 1.6-2.0 ms per search against 0.6.  A RE-pricing
    // count (an alignment that was priced off) is not taken here: it belongs to the look it was launched in front of.
    HIPC(hipEventSynchronize(h->price_ready));
    pol.set_first_price(pc, price_share(h));
  }
  const policy::KnnKernel kern = pol.kernel(pc, it, st != nullptr);
  if (kern != policy::KnnKernel::Tile) {
    // settled launch: every lane searches its own windows of the direction-sorted reference (lsgpu_cone.hip.h)
    if (h->cone_pending) {   // (built on the side stream beside the first iterations: lsgpu_icp_compute)
      HIPC(hipStreamWaitEvent(h->stream, h->cone_done, 0));
      h->cone_pending = false;
    }
    a.front_blocks = 0;
    const bool pay_index = pay_probe && kern == policy::KnnKernel::Cone && h->pay_voxel_timed && !h->pay_index_timed;
    if (pay_index) HIPC(hipEventRecord(h->ev_pay[2], h->stream));
    if (kern == policy::KnnKernel::ConeProbe)   // balls as wide as the last ICP step: a probe of the query's own direction first
      hipLaunchKernelGGL((k_knn_cone<LSGPU_CONE_WAVES, true>), dim3((a.ntiles + LSGPU_CONE_WAVES - 1) / LSGPU_CONE_WAVES), dim3(LSGPU_CONE_WAVES * 64), (size_t)h->cone.rows * sizeof(float4), h->stream, a, h->cone);
    else
      hipLaunchKernelGGL((k_knn_cone<LSGPU_CONE_WAVES, false>), dim3((a.ntiles + LSGPU_CONE_WAVES - 1) / LSGPU_CONE_WAVES), dim3(LSGPU_CONE_WAVES * 64), (size_t)h->cone.rows * sizeof(float4), h->stream, a, h->cone);
    if (timed) HIPC(hipEventRecord(ev->b, h->stream));
    if (pay_index) { HIPC(hipEventRecord(h->ev_pay[3], h->stream)); h->pay_index_timed = true; }
    HIPC(hipGetLastError());
    return LSGPU_OK;
  }
#ifdef LSGPU_EXPERIMENTS
  // measured-slower variants (DESIGN.md "Rejected after measurement"), compiled only into the experiments build:
  //   knn_rows 1: settled launches classify first and search row-wise on the compacted list; 2: k_knn_rows also stands
  //   in for k_knn_tile everywhere else; knn_lane: one lane per query in capped launches; tile_waves 4; xcd_swizzle
  if (tn.knn_rows >= 1 && capped && !wide && st && !tn.knn_lane) {
    hipLaunchKernelGGL(k_knn_classify, dim3((nq + kClassifyPerBlock - 1) / kClassifyPerBlock), dim3(256), 0, h->stream, a);
    hipLaunchKernelGGL(k_knn_rows<true>, dim3(a.ntiles), dim3(64), 0, h->stream, a);
    if (timed) { HIPC(hipEventRecord(ev->b, h->stream)); ev->second = true; HIPC(hipEventRecord(ev->c, h->stream)); }
    HIPC(hipGetLastError());
    return LSGPU_OK;
  }
  if (tn.knn_rows >= 2 && !tn.knn_lane) {
    a.spread_route_r = 0.f;  // (no routing in the row-wise kernel; stragglers by radius still go to the fallback)
    hipLaunchKernelGGL(k_knn_rows<false>, dim3(a.ntiles), dim3(64), 0, h->stream, a);
    if (timed) HIPC(hipEventRecord(ev->b, h->stream));
    if (!capped) hipLaunchKernelGGL(k_knn_fallback, dim3(kFallbackBlocks), dim3(256), 0, h->stream, a);
    if (timed) { ev->second = true; HIPC(hipEventRecord(ev->c, h->stream)); }
    HIPC(hipGetLastError());
    return LSGPU_OK;
  }
  if (capped && tn.knn_lane) {
    hipLaunchKernelGGL(k_knn_lane, dim3(nblk(nq)), dim3(256), 0, h->stream, a);
    if (timed) { HIPC(hipEventRecord(ev->b, h->stream)); ev->second = true; HIPC(hipEventRecord(ev->c, h->stream)); }
    HIPC(hipGetLastError());
    return LSGPU_OK;
  }
  a.xcd_swizzle = tn.xcd_swizzle;
  if (tn.tile_waves == 4)
    hipLaunchKernelGGL((k_knn_tile<4, false>), dim3((a.ntiles + 3) / 4), dim3(256), 0, h->stream, a);
  else
#endif
  if (wide && tn.lazy_need)   // balls still as wide as the last ICP step: the instantiation that re-tests chunks before fetching them
    hipLaunchKernelGGL((k_knn_tile<1, true>), dim3(a.ntiles + a.front_blocks), dim3(64), 0, h->stream, a);
  else if (!wide && tn.lane_split)   // settled: most tiles have fewer than 32 searching lanes -- their candidates are shared out over the idle ones
    hipLaunchKernelGGL((k_knn_tile<1, false, true>), dim3(a.ntiles + a.front_blocks), dim3(64), 0, h->stream, a);
  else
    hipLaunchKernelGGL((k_knn_tile<1, false>), dim3(a.ntiles + a.front_blocks), dim3(64), 0, h->stream, a);
  if (timed) HIPC(hipEventRecord(ev->b, h->stream));
  // stragglers (balls > r_cap) only exist in uncapped launches; a settled launch without front rows hands a few
  // thousand queries at most to the row pass (one DPP row per query)
  bool second = true;
  if (a.front_blocks > 0) {
    second = false;   // nothing was handed over
  } else if (settled && tn.rowq) {
    hipLaunchKernelGGL(k_knn_rowq, dim3(tn.rowq_blocks), dim3(256), 0, h->stream, a);
  } else if (!capped || a.spread_route_r > 0.f) {
    hipLaunchKernelGGL(k_knn_fallback, dim3(settled ? kFallbackBlocksSettled : kFallbackBlocks), dim3(256), 0, h->stream, a);
  } else {
    second = false;
  }
  if (timed && second) { ev->second = true; HIPC(hipEventRecord(ev->c, h->stream)); }   // (an event pair around nothing still reads ~5 us)
  if (pay_voxel) { HIPC(hipEventRecord(h->ev_pay[1], h->stream)); h->pay_voxel_timed = true; }
  if (pricing) {
    if (!h->price_ready) HIPC(hipEventCreateWithFlags(&h->price_ready, hipEventDisableTiming));
    HIPC(hipMemcpyAsync(h->h_price, h->price_cnt.p, (size_t)kPriceSlots * kPriceStride * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipEventRecord(h->price_ready, h->stream));
    pol.priced();
  }
  HIPC(hipGetLastError());
  return LSGPU_OK;
}

#define RCCLC(expr)                                                                     \
  do {                                                                                  \
    ncclResult_t r__ = (expr);                                                          \
    if (r__ != ncclSuccess) {                                                           \
      h->err = std::string("RCCL: ") + (rccl_api()->GetErrorString ? rccl_api()->GetErrorString(r__) : "error"); \
      return LSGPU_HIP_ERROR;                                                           \
    }                                                                                   \
  } while (0)

// TrimmedDist order statistic of d2[0..n) -> rank k; leaves hist3 + sel[2] for select_limit().
static int run_select(lsgpu_icp* h, const float* d2, int n, uint32_t k, bool zero_hist, const IcpState* st,
                      bool use_comm, bool predicted, int passes) {
  if (zero_hist) HIPC(hipMemsetAsync(h->hist.p, 0, 3 * kHistBins * sizeof(uint32_t), h->stream));
  if (zero_hist) {  // sel[0] = {0, k}: constant during an align, uploaded once
    SelState s0{0u, k};
    std::memcpy(h->h_pinned + 56, &s0, sizeof(s0));
    HIPC(hipMemcpyAsync(h->sel.p, h->h_pinned + 56, sizeof(SelState), hipMemcpyHostToDevice, h->stream));
  }
  const int nb = std::min(kHistBlocks, nblk(n));
  const int pr = predicted && st ? 1 : 0;
  hipLaunchKernelGGL(k_hist1, dim3(nb), dim3(256), 0, h->stream, d2, n, h->hist.p, st, pr);
  if (use_comm && h->comm) { comm_mark(h, true); RCCLC(rccl_api()->AllReduce(h->hist.p, h->hist.p, kHistBins, ncclUint32, ncclSum, h->comm, h->stream)); comm_mark(h, false); }
  hipLaunchKernelGGL(k_hist_refine<2>, dim3(nb), dim3(256), 0, h->stream, d2, n, h->hist.p,
                     h->sel.p, h->sel.p + 1, h->hist.p + kHistBins, st, pr, h->sel_aux.p);
  if (use_comm && h->comm) { comm_mark(h, true); RCCLC(rccl_api()->AllReduce(h->hist.p + kHistBins, h->hist.p + kHistBins, kHistBins, ncclUint32, ncclSum, h->comm, h->stream)); comm_mark(h, false); }
  if (passes < 3) { HIPC(hipGetLastError()); return LSGPU_OK; }   // (fused select: k_normal_eq_loop settles the limit inside its slice)
  hipLaunchKernelGGL(k_hist_refine<3>, dim3(nb), dim3(256), 0, h->stream, d2, n,
                     h->hist.p + kHistBins, h->sel.p + 1, h->sel.p + 2, h->hist.p + 2 * kHistBins, st, pr,
                     h->sel_aux.p);
  if (use_comm && h->comm) { comm_mark(h, true); RCCLC(rccl_api()->AllReduce(h->hist.p + 2 * kHistBins, h->hist.p + 2 * kHistBins, kHistBins, ncclUint32, ncclSum, h->comm, h->stream)); comm_mark(h, false); }
  HIPC(hipGetLastError());
  return LSGPU_OK;
}

// Direction index of the current reference (lsgpu_cone.hip.h) on the handle's current stream / sort scratch: keys of
// the Morton-sorted points, three radix passes, SoA copy + position map + (row, column) table + zeta range per row.
// is a direction index built for the current reference at all?  (host-side facts only)
static bool cone_wanted(const lsgpu_icp* h) {
  const int64_t nr = h->nr;
  if (!(tuning().cone && h->cone_origin_inside && nr >= 1024)) return false;
  // a reference with more points than 0.6 x the occupancy limit x the number of bins cannot come out below the limit
  // (measured: 4.3 / 6.3 / 8.5 points per occupied bin at 3.0 / 4.0 / 5.0 per bin): spare it the build (1.8 ms at 8 M points)
  return !((double)nr > 0.6 * (double)tuning().cone_max_occupancy * (double)tuning().cone_rows * (double)tuning().cone_cols);
}
static int build_cone_index(lsgpu_icp* h) {
  const int64_t nr = h->nr;
  h->cone_ok = false;
  if (!cone_wanted(h)) return LSGPU_OK;
  ConeDev c;
  std::memset(&c, 0, sizeof(c));
  c.ox = -h->mean[0]; c.oy = -h->mean[1]; c.oz = -h->mean[2];
  c.rows = tuning().cone_rows; c.cols = tuning().cone_cols;
  const float zr = std::max(h->cone_zeta_hi - h->cone_zeta_lo, 1e-3f);
  c.z0 = h->cone_zeta_lo - 1e-5f - 1e-4f * zr;
  c.rs = (float)c.rows / (zr * 1.0002f + 2e-5f);
  c.cs = (float)c.cols * 0.25f;
  const size_t npad = (((size_t)nr + 3) & ~(size_t)3) + kConePad, nkeys = (size_t)c.rows * (size_t)c.cols;
  HIPC(h->cone_soa.reserve((size_t)kConeGF4 * npad));
  HIPC(h->cone_map.reserve(npad));
  HIPC(h->cone_tab.reserve(nkeys + 1)); HIPC(h->cone_rowz.reserve((size_t)c.rows));
  HIPC(h->sc->keys.reserve(nr)); HIPC(h->sc->vals.reserve(nr));
  c.soa = reinterpret_cast<const float4*>(h->cone_soa.p); c.map = h->cone_map.p; c.tab = h->cone_tab.p; c.rowz = h->cone_rowz.p;
  hipLaunchKernelGGL(k_cone_keys, dim3(nblk(nr)), dim3(256), 0, h->cur, h->pts.p, nr, c,
                     h->sc->keys.p, h->sc->vals.p);
  int nbits = 1;
  while (((size_t)1 << nbits) < nkeys) ++nbits;
  const int rc = sort_pairs(h, nr, nbits);
  if (rc) return rc;
  HIPC(h->cone_occ.reserve(1));
  HIPC(hipMemsetAsync(h->cone_occ.p, 0, sizeof(uint32_t), h->cur));
  hipLaunchKernelGGL(k_cone_gather, dim3(nblk((int64_t)npad)), dim3(256), 0, h->cur, h->pts.p, h->sc->vals_alt.p,
                     h->sc->keys_alt.p, nr, c, h->cone_soa.p, h->cone_map.p, h->cone_tab.p);
  hipLaunchKernelGGL(k_cone_rows, dim3(c.rows), dim3(256), 0, h->cur, c, h->cone_rowz.p, h->cone_occ.p);
  HIPC(hipGetLastError());
  // the number of occupied bins travels to the host behind the build; lsgpu_icp_align looks at it before its first
  // search through the index (the device is busy with the first two iterations by then)
  if (!h->cone_occ_ready) HIPC(hipEventCreateWithFlags(&h->cone_occ_ready, hipEventDisableTiming));
  // (its own pinned word -- the 128-double staging block also carries the loop state, the scan totals and the grid's
  // words --, and an earlier build's copy may still be in flight on the side stream when the next one starts: wait for it
  // before the word is reset)
  if (!h->h_cone_occ) HIPC(hipHostMalloc((void**)&h->h_cone_occ, 64, hipHostMallocDefault));
  else if (hipEventSynchronize(h->cone_occ_ready) != hipSuccess) (void)hipGetLastError();
  uint32_t* ho = h->h_cone_occ;
  *ho = 0u;
  HIPC(hipMemcpyAsync(ho, h->cone_occ.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->cur));
  HIPC(hipEventRecord(h->cone_occ_ready, h->cur));
  h->cone = c;
  h->cone_ok = true;
  h->cone_decided = false;
  return LSGPU_OK;
}

static uint32_t trim_rank(int64_t n, float ratio) {
  int64_t k = (int64_t)((float)n * ratio);  // values.size() * ratio, truncated
  if (k >= n) k = n - 1;
  if (k < 0) k = 0;
  return (uint32_t)k;
}

extern "C" {

int lsgpu_icp_get_reference_mean(lsgpu_icp* h, float mean[3]) {
  if (!h || !mean) return LSGPU_BAD_ARG;
  if (h->nr <= 0) return LSGPU_BAD_ARG;
  std::memcpy(mean, h->mean, 3 * sizeof(float));
  return LSGPU_OK;
}

int lsgpu_icp_set_reference(lsgpu_icp* h, const float* ref_xyz1, const float* ref_normals,
                            int64_t nr) {
  if (!h) return LSGPU_BAD_ARG;
  h->err.clear();
  if (!ref_xyz1 || nr <= 0 || nr > 0x7FFFFFF0ll) { h->err = "set_reference: empty or oversize cloud"; h->nr = 0; return LSGPU_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  h->nr = 0;
  const float4* src = nullptr;
  int rc = stage_points(h, ref_xyz1, nr, h->ref_in, &src);
  if (rc) return rc;
  const float* nsrc = nullptr;
  if (ref_normals) {
    if (is_device_ptr(ref_normals)) nsrc = ref_normals;
    else {
      HIPC(h->nrm_in.reserve(3 * nr));
      HIPC(hipMemcpyAsync(h->nrm_in.p, ref_normals, (size_t)nr * 12, hipMemcpyHostToDevice, h->stream));
      nsrc = h->nrm_in.p;
    }
  }
  // ---- mean + bounding box (step 2) and the grid geometry, all on the device: nothing below waits for the host
  // until the cell counts are needed (one round trip per set_reference; there used to be two)
  HIPC(h->stat_partials.reserve(kStatBlocks + 1));
  HIPC(h->geom.reserve(1));
  const int sb = std::min(kStatBlocks, nblk(nr));
  hipLaunchKernelGGL(k_ref_stats, dim3(sb), dim3(256), 0, h->stream, src, nr, h->stat_partials.p);
  hipLaunchKernelGGL(k_ref_stats_final, dim3(1), dim3(64), 0, h->stream, h->stat_partials.p, sb,
                     h->stat_partials.p + kStatBlocks, nr, h->cfg.cell_size, h->geom.p);
  // ---- keys, sort, gather: 16 key bits per axis in total (`bits` address level-0 cells, `fine` order points inside
  // a cell; the split is chosen by k_ref_stats_final), i.e. always 48 key bits
  HIPC(h->sc->keys.reserve(nr)); HIPC(h->sc->vals.reserve(nr));
  HIPC(h->pts.reserve(nr + 8)); HIPC(h->nrm.reserve(nr)); HIPC(h->ref_inv.reserve(nr));
  hipLaunchKernelGGL(k_ref_keys, dim3(nblk(nr)), dim3(256), 0, h->stream, src, nr, h->geom.p, h->sc->keys.p, h->sc->vals.p);
  rc = sort_pairs(h, nr, 48);
  if (rc) return rc;
  hipLaunchKernelGGL(k_ref_gather, dim3(nblk(nr + 8)), dim3(256), 0, h->stream, src, nsrc, nr,
                     h->sc->vals_alt.p, h->geom.p, h->pts.p, h->nrm.p, h->ref_inv.p);
  // ---- chunks: flags -> inclusive scan -> bounds
  HIPC(h->flags.reserve(nr)); HIPC(h->cidx.reserve(nr));
  hipLaunchKernelGGL(k_chunk_flags, dim3(nblk(nr)), dim3(256), 0, h->stream, h->sc->keys_alt.p, nr, h->geom.p,
                     h->flags.p);
  rc = scan_u32(h, h->flags.p, h->cidx.p, (size_t)nr, /*inclusive*/ true);
  if (rc) return rc;
  // ---- cell counts per level (+ chunk count, + the geometry) -> host, to size the tables
  HIPC(h->counters.reserve(64));
  HIPC(hipMemsetAsync(h->counters.p, 0, 64 * sizeof(uint32_t), h->stream));
  hipLaunchKernelGGL(k_cells_count, dim3(std::min(512, nblk(nr))), dim3(256), 0, h->stream, h->sc->keys_alt.p, nr, h->geom.p,
                     h->counters.p);
  uint32_t* hc = reinterpret_cast<uint32_t*>(h->h_pinned);
  GeomDev* hg = reinterpret_cast<GeomDev*>(h->h_pinned + 16);
  static_assert(sizeof(GeomDev) <= 16 * sizeof(double), "geometry staging");
  {
    static_assert(sizeof(GeomDev) % 4 == 0, "copied word by word");
    ToHost c{}; int used = 0;
    to_host_add(&c, &used, h->counters.p, hc, kMaxLevels * sizeof(uint32_t));
    to_host_add(&c, &used, h->cidx.p + (nr - 1), hc + 20, sizeof(uint32_t));
    to_host_add(&c, &used, h->geom.p, hg, sizeof(GeomDev));
    hipLaunchKernelGGL(k_to_host, dim3(1), dim3(64), 0, h->stream, c);
    HIPC(hipGetLastError());
  }
  if (h->hook_before_ref_sync) {   // lsgpu_icp_compute: the reading's side of the work is enqueued on its own stream now
    rc = h->hook_before_ref_sync();
    if (rc) return rc;
  }
  HIPC(hipStreamSynchronize(h->stream));
  if (hg->bad) { h->err = "set_reference: non-finite coordinates"; return LSGPU_BAD_ARG; }
  for (int d = 0; d < 3; ++d) h->mean[d] = hg->mean[d];
  const int bits = hg->bits, fine = hg->fine;
  const float h0 = hg->h0;
  GridDev g;
  std::memset(&g, 0, sizeof(g));
  g.ox = hg->ox; g.oy = hg->oy; g.oz = hg->oz;
  g.h0 = h0; g.hf = hg->hf; g.inv_hf = hg->inv_hf; g.fine = fine; g.bits = bits;
  const uint32_t nchunks = hc[20];
  size_t total = 0, off[kMaxLevels];
  uint32_t cap[kMaxLevels], ncell[kMaxLevels];
  for (int l = 0; l <= bits; ++l) ncell[l] = hc[l];
  for (int l = 0; l <= bits; ++l) {
    uint32_t c = 4;
    while (c < 4u * ncell[l]) c <<= 1;   // (load <= 0.25: the longest probe chain of the benchmark scan's tables is 7 slots)
    cap[l] = c; off[l] = total; total += c;
  }
  HIPC(h->bounds.reserve((size_t)nchunks + 1));
  HIPC(h->chunks.reserve(nchunks));
  hipLaunchKernelGGL(k_chunk_bounds, dim3(nblk(nr)), dim3(256), 0, h->stream, h->flags.p, h->cidx.p,
                     nr, h->bounds.p);
  hipLaunchKernelGGL(k_chunk_boxes, dim3((nchunks + 3) / 4), dim3(256), 0, h->stream, h->pts.p,
                     h->bounds.p, nchunks, h->chunks.p);
  // chunk-blocked SoA copy for the broadcast evaluation: <= 3 rounding slots per chunk
  HIPC(h->soa_cnt4.reserve(nchunks)); HIPC(h->soa_first.reserve(nchunks)); HIPC(h->soa_base.reserve(nchunks));
  HIPC(h->soa.reserve(3 * ((size_t)nr + 3 * (size_t)nchunks) + 16));
  HIPC(h->chunk_groups.reserve((size_t)nchunks / kChunkGroup + 1));
  hipLaunchKernelGGL(k_chunk_cnt4, dim3((nchunks + 255) / 256), dim3(256), 0, h->stream, h->bounds.p, nchunks,
                     h->soa_cnt4.p, (const ChunkDesc*)h->chunks.p, h->chunk_groups.p);
  rc = scan_u32(h, h->soa_cnt4.p, h->soa_first.p, nchunks);
  if (rc) return rc;
  hipLaunchKernelGGL(k_soa_fill, dim3((nchunks + 3) / 4), dim3(256), 0, h->stream, h->pts.p, h->bounds.p,
                     h->soa_first.p, nchunks, h->soa.p, h->soa_base.p);
  HIPC(h->tables.reserve(total));
  HIPC(hipMemsetAsync(h->tables.p, 0xFF, total * sizeof(HashEntry), h->stream));
  TableSet ts;
  std::memset(&ts, 0, sizeof(ts));
  for (int l = 0; l <= bits; ++l) {
    ts.tab[l] = h->tables.p + off[l]; ts.mask[l] = cap[l] - 1;
    g.tab[l] = ts.tab[l]; g.mask[l] = ts.mask[l];
  }
  if (getenv("LSGPU_CELLS_SPLIT")) {   // (dev: one launch per level, to time them)
    for (int l = 0; l <= bits; ++l)
      hipLaunchKernelGGL(k_cells_fill, dim3((nchunks + 255) / 256, 1), dim3(256), 0, h->stream, h->sc->keys_alt.p, h->bounds.p, nchunks, fine, bits, ts, l);
  } else
  hipLaunchKernelGGL(k_cells_fill, dim3((nchunks + 255) / 256, bits + 1), dim3(256), 0, h->stream, h->sc->keys_alt.p,
                     h->bounds.p, nchunks, fine, bits, ts, 0);
  HIPC(hipGetLastError());  // (no sync: align / knn follow on the same stream)
  h->grid = g;
  h->nr = nr;
  h->nchunks = nchunks;
  if (h->hook_after_grid) {    // lsgpu_icp_compute: the mean is known and this stream is busy -- now the queries' side
    rc = h->hook_after_grid();
    if (rc) return rc;
  }
  // ---- direction index for the settled launches (after k_cells_fill: its sort reuses the Morton keys' buffers).
  // lsgpu_icp_compute builds it on its side stream instead, beside the first iterations of the loop (defer_cone).
  h->cone_ok = false; h->cone_pending = false; h->cone_build_in_align = false;
  h->cone_zeta_lo = hg->zeta_lo; h->cone_zeta_hi = hg->zeta_hi; h->cone_origin_inside = hg->origin_inside != 0;
  if (!h->defer_cone) {
    rc = build_cone_index(h);
    if (rc) return rc;
  }
  std::memset(&h->info, 0, sizeof(h->info));
  h->info.n_reference = nr;
  h->info.bits_per_axis = bits;
  h->info.fine_bits = fine;
  h->info.cell_size = h0;
  h->info.n_chunks = nchunks;
  for (int l = 0; l <= bits; ++l) h->info.cells[l] = ncell[l];
  h->info.table_bytes = total * sizeof(HashEntry);
  return LSGPU_OK;
}

int lsgpu_comm_get_unique_id(void* id) {
  if (!id) return LSGPU_BAD_ARG;
  RcclApi* api = rccl_api();
  if (!api) return LSGPU_HIP_ERROR;
  ncclUniqueId u;
  if (api->GetUniqueId(&u) != ncclSuccess) return LSGPU_HIP_ERROR;
  static_assert(sizeof(u) == LSGPU_COMM_ID_BYTES, "unique id size");
  std::memcpy(id, &u, sizeof(u));
  return LSGPU_OK;
}

int lsgpu_icp_comm_init(lsgpu_icp* h, int rank, int nranks, const void* id) {
  if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return LSGPU_BAD_ARG;
  h->err.clear();
  RcclApi* api = rccl_api();
  if (!api) { h->err = "librccl.so.1 could not be loaded"; return LSGPU_HIP_ERROR; }
  HIPC(hipSetDevice(h->device));
  if (h->comm) { (void)api->CommDestroy(h->comm); h->comm = nullptr; }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  HIPC(h->comm_tmp.reserve(4));  // entry handshake of every align (allocated here: the handshake itself must not fail locally)
  RCCLC(api->CommInitRank(&h->comm, nranks, u, rank));
  h->comm_rank = rank; h->comm_size = nranks;
  return LSGPU_OK;
}

int lsgpu_icp_get_policy_info(lsgpu_icp* h, lsgpu_policy_info* out) {
  if (!h || !out) return LSGPU_BAD_ARG;
  std::memset(out, 0, sizeof(*out));
  out->index_rest = h->index_rest; out->pay_voxel_us = h->pay_voxel_us; out->pay_index_us = h->pay_index_us;
  out->ssn_sort_fallbacks = h->ssn_sort_fallbacks; out->ssn_calls = h->ssn_calls;
  return LSGPU_OK;
}

int lsgpu_icp_get_info(lsgpu_icp* h, lsgpu_icp_info* out) {
  if (!h || !out || h->nr <= 0) return LSGPU_BAD_ARG;
  *out = h->info;
  return LSGPU_OK;
}

int lsgpu_knn(lsgpu_icp* h, const float* query_xyz1, int64_t nq, const float T[16], int32_t* ids,
              float* d2) {
  if (!h) return LSGPU_BAD_ARG;
  h->err.clear();
  if (h->nr <= 0) { h->err = "knn: no reference set"; return LSGPU_BAD_ARG; }
  if (nq == 0) return LSGPU_OK;
  if (!query_xyz1 || !ids || !d2 || nq < 0 || nq > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const Mat34 Tm = to_mat34(T ? T : I);
  // the transform is applied inside the search kernels; sorting uses the raw coordinates
  const Mat34 Id = to_mat34(I);
  int rc = prepare_queries(h, query_xyz1, nq, Id);
  if (rc) return rc;
  policy::Iteration probe;   // a seeded, uncapped, wide search outside any loop
  probe.seed = true; probe.capped = false; probe.wide = true;
  h->pol.begin_align(false, false, false, 0.f);
  rc = run_knn(h, Tm, nullptr, probe, false);
  if (rc) return rc;
  const bool dev_out = is_device_ptr(ids);
  int* ids_o = ids; float* d2_o = d2;
  if (!dev_out) {
    HIPC(h->ids_io.reserve(nq)); HIPC(h->d2_io.reserve(nq));
    ids_o = h->ids_io.p; d2_o = h->d2_io.p;
  }
  hipLaunchKernelGGL(k_knn_unpermute, dim3(nblk(nq)), dim3(256), 0, h->stream, h->rdq.p, (int)nq,
                     h->ids.p, h->d2.p, h->pts.p, ids_o, d2_o);
  HIPC(hipGetLastError());
  if (!dev_out) {
    HIPC(hipMemcpyAsync(ids, ids_o, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipMemcpyAsync(d2, d2_o, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream));
  }
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

int lsgpu_trim_limit(lsgpu_icp* h, const float* d2, int64_t n, float ratio, float* limit) {
  if (!h || !limit) return LSGPU_BAD_ARG;
  h->err.clear();
  if (n <= 0 || !d2) return LSGPU_NO_CONVERGENCE;  // "no outlier to filter"
  if (n > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  int rc = ensure_loop_buffers(h, 1);
  if (rc) return rc;
  const float* src = d2;
  if (!is_device_ptr(d2)) {
    HIPC(h->d2_io.reserve(n));
    HIPC(hipMemcpyAsync(h->d2_io.p, d2, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    src = h->d2_io.p;
  }
  rc = run_select(h, src, (int)n, trim_rank(n, ratio), true, nullptr, false, false);
  if (rc) return rc;
  hipLaunchKernelGGL(k_limit_out, dim3(1), dim3(256), 0, h->stream, h->hist.p + 2 * kHistBins,
                     h->sel.p + 2, h->limit_dev.p);
  float* hl = reinterpret_cast<float*>(h->h_pinned);
  HIPC(hipMemcpyAsync(hl, h->limit_dev.p, 4, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  *limit = *hl;
  return LSGPU_OK;
}

int lsgpu_normal_eq(lsgpu_icp* h, const float* query_xyz1, int64_t nq, const float T[16],
                    const int32_t* ids, const float* d2, float limit, double out[29]) {
  if (!h || !out) return LSGPU_BAD_ARG;
  h->err.clear();
  if (h->nr <= 0) { h->err = "normal_eq: no reference set"; return LSGPU_BAD_ARG; }
  if (nq <= 0 || !query_xyz1 || !ids || !d2 || nq > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  int rc = ensure_loop_buffers(h, nq);
  if (rc) return rc;
  const float4* q = nullptr;
  rc = stage_points(h, query_xyz1, nq, h->q_in, &q);
  if (rc) return rc;
  const int* idp = ids; const float* dp = d2;
  if (!is_device_ptr(ids)) {
    HIPC(h->ids_io.reserve(nq));
    HIPC(hipMemcpyAsync(h->ids_io.p, ids, (size_t)nq * 4, hipMemcpyHostToDevice, h->stream));
    idp = h->ids_io.p;
  }
  if (!is_device_ptr(d2)) {
    HIPC(h->d2_io.reserve(nq));
    HIPC(hipMemcpyAsync(h->d2_io.p, d2, (size_t)nq * 4, hipMemcpyHostToDevice, h->stream));
    dp = h->d2_io.p;
  }
  float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const Mat34 Tm = to_mat34(T ? T : I);
  const int nb = std::min(kNeBlocks, nblk(nq));
  hipLaunchKernelGGL((k_normal_eq<true, false>), dim3(nb), dim3(256), 0, h->stream, q, (int)nq, Tm,
                     idp, dp, h->pts.p, h->nrm.p, h->ref_inv.p, (const uint32_t*)nullptr,
                     (const SelState*)nullptr, limit, (float*)nullptr, h->ne_partials.p);
  hipLaunchKernelGGL(k_ne_final, dim3(1), dim3(1024), 0, h->stream, h->ne_partials.p, nb, h->ne_out.p);
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(h->h_pinned, h->ne_out.p, kNe * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  std::memcpy(out, h->h_pinned, kNe * sizeof(double));
  return LSGPU_OK;
}

int lsgpu_transform_points(lsgpu_icp* h, const float T[16], const float* xyz1, int64_t n,
                           float* out) {
  if (!h || !T || !out) return LSGPU_BAD_ARG;
  h->err.clear();
  if (n == 0) return LSGPU_OK;
  if (!xyz1 || n < 0) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  const float4* src = nullptr;
  int rc = stage_points(h, xyz1, n, h->q_in, &src);
  if (rc) return rc;
  const bool dev_out = is_device_ptr(out);
  float4* dst = reinterpret_cast<float4*>(out);
  if (!dev_out) { HIPC(h->rdq.reserve(n)); dst = h->rdq.p; }
  hipLaunchKernelGGL(k_transform, dim3(nblk(n)), dim3(256), 0, h->stream, src, n, to_mat34(T), dst);
  HIPC(hipGetLastError());
  if (!dev_out) HIPC(hipMemcpyAsync(out, dst, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

int lsgpu_rotate_descriptors(lsgpu_icp* h, const float T[16], const float* desc3, int64_t n, float* out) {
  if (!h || !T) return LSGPU_BAD_ARG;
  h->err.clear();
  if (n == 0) return LSGPU_OK;
  if (!desc3 || !out || n < 0 || n > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  if (!lsgpu_check_rigid(T)) { h->err = "rotate_descriptors: the matrix is not rigid"; return LSGPU_BAD_ARG; }   // TransformationError upstream
  HIPC(hipSetDevice(h->device));
  const float* src = desc3;
  if (!is_device_ptr(desc3)) {
    HIPC(h->flt_nrm.reserve(3 * n));
    HIPC(hipMemcpyAsync(h->flt_nrm.p, desc3, (size_t)n * 12, hipMemcpyHostToDevice, h->stream));
    src = h->flt_nrm.p;
  }
  const bool dev_out = is_device_ptr(out);
  float* dst = out;
  if (!dev_out) { HIPC(h->ssn_box_normal.reserve(3 * n)); dst = h->ssn_box_normal.p; }
  hipLaunchKernelGGL(k_rotate3, dim3(nblk(n)), dim3(256), 0, h->stream, src, n, to_mat34(T), dst);
  HIPC(hipGetLastError());
  if (!dev_out) HIPC(hipMemcpyAsync(out, dst, (size_t)n * 12, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

}  // extern "C"

// prefix sum of n uint32 on the handle's current stream (lsgpu_scan.hip.h; in == out is fine)
static int scan_u32(lsgpu_icp* h, const uint32_t* in, uint32_t* out, size_t n, bool inclusive, bool nonzero) {
  if (n == 0) return LSGPU_OK;
  const int nb = (int)((n + kScanTile - 1) / kScanTile);
  uint32_t* sums = nullptr;
  if (nb > 1) {
    HIPC(h->sc->sort_tmp.reserve((size_t)nb * sizeof(uint32_t)));
    sums = reinterpret_cast<uint32_t*>(h->sc->sort_tmp.p);
    if (nonzero) hipLaunchKernelGGL(k_scan_sums<true>, dim3(nb), dim3(256), 0, h->cur, in, n, sums);
    else hipLaunchKernelGGL(k_scan_sums<false>, dim3(nb), dim3(256), 0, h->cur, in, n, sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, h->cur, sums, nb);
  }
  if (nonzero) hipLaunchKernelGGL((k_scan_write<false, true>), dim3(nb), dim3(256), 0, h->cur, in, out, n, (const uint32_t*)sums);   // (exclusive: the one use)
  else if (inclusive) hipLaunchKernelGGL((k_scan_write<true, false>), dim3(nb), dim3(256), 0, h->cur, in, out, n, (const uint32_t*)sums);
  else hipLaunchKernelGGL((k_scan_write<false, false>), dim3(nb), dim3(256), 0, h->cur, in, out, n, (const uint32_t*)sums);
  HIPC(hipGetLastError());
  return LSGPU_OK;
}

// Up to `kmax` draws of the library stream -> h->ssn_draws, speculatively: the stream stays locked until
// the caller commits the number the sequential filter would have consumed (DrawStream::commit).
static int upload_draws_begin(lsgpu_icp* h, int64_t seed, size_t kmax) {
  if (kmax + 1 > h->draws_pinned_cap) {
    if (h->draws_pinned) (void)hipHostFree(h->draws_pinned);
    h->draws_pinned = nullptr; h->draws_pinned_cap = 0;
    HIPC(hipHostMalloc((void**)&h->draws_pinned, (kmax + 1) * sizeof(float), hipHostMallocDefault));
    h->draws_pinned_cap = kmax + 1;
  }
  HIPC(h->ssn_draws.reserve(kmax + 1));
  DrawStream::global().begin(seed, kmax, h->draws_pinned);
  if (kmax) {
    const hipError_t e = hipMemcpyAsync(h->ssn_draws.p, h->draws_pinned, kmax * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) { DrawStream::global().commit(0); HIPC(e); }
  }
  return LSGPU_OK;
}

// The same, produced AHEAD on a helper thread while the calling thread enqueues the kernels that come before the draws
// are needed (the ~1.3 ns a draw costs used to sit on the stream's critical path: the device idled ~0.3 ms per 200 k
// points in front of k_ssn_select and again in front of k_draw_select, profiles/r03_bench.stats.txt).  begin() locks
// the stream and starts the helper, ready() joins it and enqueues the H2D of all draws on the handle's stream, the
// destructor consumes `used` draws -- what the sequential filters would have made -- and unlocks.  One DrawAhead can
// serve consecutive filters: lsgpu_icp_compute draws for the reference filter and the reading filter in one go, the
// reading filter's draws start where the reference filter's end (known once its kernels ran).
struct DrawAhead {
  lsgpu_icp* h = nullptr;
  size_t kmax = 0, used = 0, kfirst = 0;
  bool open = false, waited = false, unlocked_early = false;
  hipError_t upload_err = hipSuccess;
  std::atomic<int> first_sent{0};   // the first `kfirst` draws are on their way to the device (their own event)
  std::thread worker;
  DrawAhead() = default;
  DrawAhead(const DrawAhead&) = delete;
  DrawAhead& operator=(const DrawAhead&) = delete;
  // (k_first: draws [0, k_first) are sent off and can be waited for -- ready(k_first) -- before the rest exists)
  int begin(lsgpu_icp* hh, int64_t seed, size_t k, size_t k_first = 0) {
    h = hh; kmax = k; kfirst = (k_first && k_first < k) ? k_first : 0;
    first_sent.store(0, std::memory_order_relaxed);
    if (kmax + 1 > h->draws_pinned_cap) {
      if (h->draws_pinned) (void)hipHostFree(h->draws_pinned);
      h->draws_pinned = nullptr; h->draws_pinned_cap = 0;
      HIPC(hipHostMalloc((void**)&h->draws_pinned, (kmax + 1) * sizeof(float), hipHostMallocDefault));
      h->draws_pinned_cap = kmax + 1;
    }
    HIPC(h->ssn_draws.reserve(kmax + 1));
    if (!h->draw_stream) HIPC(hipStreamCreateWithFlags(&h->draw_stream, hipStreamNonBlocking));
    if (!h->draws_done) HIPC(hipEventCreateWithFlags(&h->draws_done, hipEventDisableTiming));
    if (!h->draws_first_done) HIPC(hipEventCreateWithFlags(&h->draws_first_done, hipEventDisableTiming));
    order_after_tail(h, h->draw_stream);
    DrawStream::global().lock(seed);
    open = true;
    if (kmax) {
      // the helper produces the draws and sends them off on a stream of their own right away: the H2D (8 MB for two
      // 1 M-point clouds) used to sit on the filter's stream in front of k_ssn_select, 0.18 ms of idle device
      lsgpu_icp* hh2 = h;
      auto produce = [hh2, k, this] {
        hipError_t e = hipSetDevice(hh2->device);
        const size_t kf = kfirst;
        if (kf) {
          // the first filter's share leaves as soon as it exists; the rest is still being produced
          DrawStream::global().generate(k, hh2->draws_pinned, kf, [&] {
            if (e == hipSuccess) e = hipMemcpyAsync(hh2->ssn_draws.p, hh2->draws_pinned, kf * sizeof(float), hipMemcpyHostToDevice, hh2->draw_stream);
            if (e == hipSuccess) e = hipEventRecord(hh2->draws_first_done, hh2->draw_stream);
            upload_err = e;
            first_sent.store(1, std::memory_order_release);
          });
          if (e == hipSuccess) e = hipMemcpyAsync(hh2->ssn_draws.p + kf, hh2->draws_pinned + kf, (k - kf) * sizeof(float), hipMemcpyHostToDevice, hh2->draw_stream);
        } else {
          DrawStream::global().generate(k, hh2->draws_pinned);
          if (e == hipSuccess) e = hipMemcpyAsync(hh2->ssn_draws.p, hh2->draws_pinned, k * sizeof(float), hipMemcpyHostToDevice, hh2->draw_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(hh2->draws_done, hh2->draw_stream);
        upload_err = e;
        first_sent.store(1, std::memory_order_release);
      };
      try {
        worker = std::thread(produce);
      } catch (const std::system_error&) {   // no thread to be had: produce the draws here
        produce();
      }
    }
    return LSGPU_OK;
  }
  int ready(size_t upto = ~(size_t)0) {   // the stream the caller enqueues on (h->cur) waits for the draws [0, upto)
    if (kmax && kfirst && upto <= kfirst) {   // the first part has its own event: no need for the rest to exist yet
      while (!first_sent.load(std::memory_order_acquire)) std::this_thread::yield();
      if (upload_err != hipSuccess) { (void)hipGetLastError(); HIPC(upload_err); }
      HIPC(hipStreamWaitEvent(h->cur, h->draws_first_done, 0));
      return LSGPU_OK;
    }
    if (worker.joinable()) worker.join();
    if (kmax) {
      if (upload_err != hipSuccess) { (void)hipGetLastError(); HIPC(upload_err); }
      HIPC(hipStreamWaitEvent(h->cur, h->draws_done, 0));
      waited = true;
    }
    return LSGPU_OK;
  }
  // The number of draws the filters will have consumed is known: consume them and unlock NOW -- the values are
  // already on their way to the device, the filters that still have to run only read them there.  (The process-wide
  // stream used to stay locked from the first filter to the end of the last, i.e. across the whole grid build and its
  // host round trip: compute calls on other handles -- other robots' tracks, a loop-closure ICP -- took turns there.)
  void commit_now(size_t total) {
    if (worker.joinable()) worker.join();
    if (open) DrawStream::global().commit(std::min(total, kmax));
    if (open) unlocked_early = true;
    open = false;
  }
  void finish() {   // consume + unlock now (the stream must not stay locked while the ICP loop runs)
    if (worker.joinable()) worker.join();
    if (open) DrawStream::global().commit(std::min(used, kmax));
    if ((open || unlocked_early) && kmax && !waited && h->draw_stream) (void)hipStreamSynchronize(h->draw_stream);   // (nobody waited for the upload: the staging buffer must be free on return)
    open = false; unlocked_early = false;
  }
  ~DrawAhead() { finish(); }
};

// totals of two exclusive scans (last scanned value + last input) -> four pinned words, behind the scans ...
// (`extra`: up to two more ranges that travel with them)
static int scan_totals_enqueue(lsgpu_icp* h, const uint32_t* in_a, const uint32_t* sc_a, size_t na,
                               const uint32_t* in_b, const uint32_t* sc_b, size_t nb,
                               const void* extra_src = nullptr, void* extra_dst = nullptr, size_t extra_bytes = 0) {
  uint32_t* hp = reinterpret_cast<uint32_t*>(h->h_pinned + 100 + 4 * h->side_totals_slot);
  hp[0] = hp[1] = hp[2] = hp[3] = 0;
  ToHost c{}; int used = 0;
  if (in_a) {
    to_host_add(&c, &used, in_a + (na - 1), hp, 4);
    to_host_add(&c, &used, sc_a + (na - 1), hp + 1, 4);
  }
  to_host_add(&c, &used, in_b + (nb - 1), hp + 2, 4);
  to_host_add(&c, &used, sc_b + (nb - 1), hp + 3, 4);
  if (extra_src) to_host_add(&c, &used, extra_src, extra_dst, extra_bytes);
  hipLaunchKernelGGL(k_to_host, dim3(1), dim3(64), 0, h->cur, c);
  HIPC(hipGetLastError());
  return LSGPU_OK;
}
// ... and the host's wait for them (synchronises the current stream)
static int scan_totals_wait(lsgpu_icp* h, uint32_t* tot_a, uint32_t* tot_b, bool b_flags = false) {
  const uint32_t* hp = reinterpret_cast<const uint32_t*>(h->h_pinned + 100 + 4 * h->side_totals_slot);
  HIPC(hipStreamSynchronize(h->cur));
  *tot_a = hp[0] + hp[1];
  *tot_b = (b_flags ? (hp[2] ? 1u : 0u) : hp[2]) + hp[3];   // (b_flags: the scanned words are flags, any non-zero word counts 1)
  return LSGPU_OK;
}
static int scan_totals(lsgpu_icp* h, const uint32_t* in_a, const uint32_t* sc_a, size_t na,
                       const uint32_t* in_b, const uint32_t* sc_b, size_t nb, uint32_t* tot_a, uint32_t* tot_b,
                       const void* extra_src = nullptr, void* extra_dst = nullptr, size_t extra_bytes = 0, bool b_flags = false) {
  const int rc = scan_totals_enqueue(h, in_a, sc_a, na, in_b, sc_b, nb, extra_src, extra_dst, extra_bytes);
  return rc ? rc : scan_totals_wait(h, tot_a, tot_b, b_flags);
}

// SamplingSurfaceNormal on device memory: src (n points) -> out_xyz1 / out_nrm (device, room for n)
// (`ahead`: draws begun by the caller, this filter's first at ahead->used; nullptr: the filter draws for itself)
static int ssn_device(lsgpu_icp* h, const float4* src, int64_t n, int knn, float ratio, int64_t seed,
                      float4* out_xyz1, float* out_nrm, int64_t* n_out, DrawAhead* ahead = nullptr, bool force_sort_levels = false) {
  *n_out = 0;
  if (!force_sort_levels) ++h->ssn_calls;
  DrawAhead own;
  if (!ahead) {   // at most one draw per point, produced while the levels below are enqueued and run
    const int rc0 = own.begin(h, seed, (size_t)n);
    if (rc0) return rc0;
    ahead = &own;
  }
  const size_t first_draw = ahead->used;
  int levels = 0;
  for (int64_t c = n; c > knn; c -= c / 2) ++levels;  // the largest child keeps count - count / 2 points
  const size_t nseg = (size_t)1 << levels;
  HIPC(h->ssn_seg_a.reserve(nseg));
  HIPC(h->ssn_seg_b.reserve(nseg));
  HIPC(h->ssn_seg_of.reserve(n));
  HIPC(h->ssn_box_pts.reserve(nseg));
  HIPC(h->ssn_box_base.reserve(nseg));
  HIPC(h->ssn_box_normal.reserve(3 * nseg));
  HIPC(h->ssn_keep.reserve(n));
  HIPC(h->ssn_out_pos.reserve(n));
  HIPC(h->ssn_bb.reserve(8));
  HIPC(h->sc->keys.reserve(n));
  HIPC(h->sc->vals.reserve(n));
  {   // the cloud's bounds and the root segment in one launch (k_ssn_bounds_root: a ticket that is zero between calls)
    const bool fresh = h->ssn_bounds_ws.cap == 0;
    HIPC(h->ssn_bounds_ws.reserve(8 + 6 * kSsnBoundsBlocks));
    if (fresh) HIPC(hipMemsetAsync(h->ssn_bounds_ws.p, 0, 8 * sizeof(uint32_t), h->stream));
    hipLaunchKernelGGL(k_ssn_bounds_root, dim3(std::min(nblk(n), kSsnBoundsBlocks)), dim3(256), 0, h->stream, src, (int)n, h->ssn_bounds_ws.p,
                       h->ssn_bb.p, h->ssn_seg_a.p);
  }
  SsnSeg* cur = h->ssn_seg_a.p;
  SsnSeg* nxt = h->ssn_seg_b.p;
  const uint32_t* idx = nullptr;
  const int* root_axis = nullptr;   // per root of the in-workgroup levels: the axis its order follows (segmented level sorts)
  // levels [0, glevels) with global sorts; the rest inside one workgroup per segment once a segment fits
  // its LDS (<= kSsnLdsMax points, <= kSsnLdsLevels levels to go)
  int glevels = 0;
  // (k_ssn_tree: up to ssn_root points and log2(ssn_root / 8) levels per workgroup; k_ssn_finish, LSGPU_SSN_OLD_FINISH: 2048 / 8)
  const bool tree_finish = !tuning().ssn_old_finish;
  // (one workgroup per root: 8192-point roots leave half the chip idle on a scan of a million points -- 128 roots, 248 us --
  // where 4096-point roots and one more global level take 50 us less; a three-scan sub-map has 383 roots of 8192)
  const int root_auto = n >= 200ll * 8192 ? 8192 : n >= 200ll * 4096 ? 4096 : 2048;
  const int root_max = tree_finish ? (tuning().ssn_root ? tuning().ssn_root : root_auto) : kSsnLdsMax;
  int root_levels = kSsnLdsLevels;
  if (tree_finish) { root_levels = 0; while ((8 << root_levels) < root_max) ++root_levels; }
  {
    const bool lds_finish = !tuning().ssn_global;
    int64_t c = n;
    while (glevels < levels && !(lds_finish && c <= root_max && levels - glevels <= root_levels)) { c -= c / 2; ++glevels; }
  }
  // the sort-free levels hand SETS over (points in original-index order + a signature): only k_ssn_tree knows how to take
  // them; the round-4 finish kernel and the all-global mode continue a sorted order, so they imply the sorted levels
  const bool select_levels = !force_sort_levels && !tuning().ssn_sort_levels && !tuning().ssn_full_sort && tree_finish &&
                             !tuning().ssn_global && glevels > 0;
  const uint32_t* root_sig = nullptr;   // per root of the in-workgroup levels: its signature (sort-free upper levels)
  if (tuning().ssn_full_sort) {   // rounds 1-3: the whole cloud sorted by (segment, coordinate) at every level
    for (int L = 0; L < glevels; ++L) {
      hipLaunchKernelGGL(k_ssn_keys, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n, idx,
                         L ? h->ssn_seg_of.p : (const uint32_t*)nullptr, cur, knn, h->sc->keys.p, h->sc->vals.p);
      int rc = sort_pairs(h, n, 32 + L);
      if (rc) return rc;
      idx = h->sc->vals_alt.p;
      const int ns = 1 << L;
      hipLaunchKernelGGL(k_ssn_split, dim3(nblk(ns)), dim3(256), 0, h->stream, src, idx, cur, ns, knn, nxt,
                         (const int*)nullptr, (int*)nullptr);
      hipLaunchKernelGGL(k_ssn_assign, dim3(nblk(n)), dim3(256), 0, h->stream, (int)n, cur, knn, h->ssn_seg_of.p, L == 0 ? 1 : 0);
      std::swap(cur, nxt);
    }
  } else if (select_levels) {
    // Sort-free upper levels (lsgpu_ssn_select.hip.h): a segment is a set + a signature, a level is the exact median in the
    // segment's total order (two 8-bit histogram passes over the range its points span + a per-segment selection among the
    // candidates left) and one stable partition: five launches per level, no sort.
    if (h->gs_plan_n != n || h->gs_plan_levels != glevels) {   // block tables: static for a cloud size
      // every level's number of blocks and first row (sizes of a level: c -> c - c / 2, c / 2)
      std::vector<uint32_t> sizes{(uint32_t)n}, next;
      h->gs_lvl_first.clear(); h->gs_lvl_blocks.clear();
      uint32_t rows = 0, segs = 0;
      for (int L = 0; L < glevels; ++L) {
        uint32_t fb = 0;
        next.clear();
        for (uint32_t c : sizes) {
          fb += (c + kGsTile - 1u) / kGsTile;
          next.push_back(c - c / 2u); next.push_back(c / 2u);
        }
        h->gs_lvl_first.push_back(rows); h->gs_lvl_blocks.push_back(fb);
        rows += fb; segs += (uint32_t)sizes.size();
        sizes.swap(next);
      }
      HIPC(h->gs_tab.reserve(rows)); HIPC(h->gs_sblk.reserve(segs));
      if (glevels <= kGsPlanLevels && ((size_t)1 << (glevels - 1)) <= kGsPlanSegs) {   // the rows themselves: on the device
        GsPlanArgs pa{};
        for (int L = 0; L < glevels; ++L) pa.first[L] = h->gs_lvl_first[L];
        hipLaunchKernelGGL(k_gs_plan, dim3(glevels), dim3(1024), 0, h->stream, (uint32_t)n, pa, h->gs_tab.p, h->gs_sblk.p);
        HIPC(hipGetLastError());
      } else {                                            // (more than 8192 segments in a level: 67 M points)
        std::vector<GsBlock> tab;
        std::vector<GsSegBlocks> sblk;
        std::vector<std::pair<uint32_t, uint32_t>> segsz{{0u, (uint32_t)n}};
        for (int L = 0; L < glevels; ++L) {
          uint32_t fb = 0;
          std::vector<std::pair<uint32_t, uint32_t>> nx;
          for (uint32_t sidx = 0; sidx < segsz.size(); ++sidx) {
            const uint32_t st = segsz[sidx].first, c = segsz[sidx].second;
            const uint32_t nb = (c + kGsTile - 1u) / kGsTile;
            sblk.push_back(GsSegBlocks{fb, nb, st, c});
            for (uint32_t k = 0; k < nb; ++k)
              tab.push_back(GsBlock{st + k * kGsTile, std::min(kGsTile, c - k * kGsTile), sidx, st, c, fb, nb, 0u});
            fb += nb;
            const uint32_t left = c - c / 2u;
            nx.push_back({st, left}); nx.push_back({st + left, c - left});
          }
          segsz.swap(nx);
        }
        // (pageable source: the copy returns when the staging is done)
        HIPC(hipMemcpyAsync(h->gs_tab.p, tab.data(), tab.size() * sizeof(GsBlock), hipMemcpyHostToDevice, h->stream));
        HIPC(hipMemcpyAsync(h->gs_sblk.p, sblk.data(), sblk.size() * sizeof(GsSegBlocks), hipMemcpyHostToDevice, h->stream));
        HIPC(hipStreamSynchronize(h->stream));
      }
      h->gs_plan_n = n; h->gs_plan_levels = glevels;
    }
    const size_t nseg_g = (size_t)1 << glevels;
    int cap = 1;
    for (uint32_t v : h->gs_lvl_blocks) cap = std::max(cap, (int)v);
    for (auto& bf : h->gs_e) HIPC(bf.reserve(n));
    for (auto& bf : h->gs_k) HIPC(bf.reserve(n));
    HIPC(h->gs_hist.reserve(4 * 256 * (nseg_g + 1))); HIPC(h->gs_median.reserve(nseg_g)); HIPC(h->gs_cand_n.reserve(2 * nseg_g + 2));
    const size_t cand_room = std::max<size_t>(nseg_g / 2, 1) * kGsCandRoom;
    HIPC(h->gs_cand.reserve(cand_room)); HIPC(h->gs_cand_blk.reserve(cand_room));
    HIPC(h->gs_cl.reserve((size_t)2 * cap)); HIPC(h->gs_err.reserve(8)); HIPC(h->gs_rng.reserve(2 * nseg_g + 2));
    HIPC(h->ssn_axis_a.reserve(nseg_g)); HIPC(h->ssn_axis_b.reserve(nseg_g));
    if (!h->h_gs_err) HIPC(hipHostMalloc((void**)&h->h_gs_err, 64, hipHostMallocDefault));
    uint32_t* cand_n[2] = {h->gs_cand_n.p, h->gs_cand_n.p + nseg_g + 1};
    uint32_t* sig_cur = reinterpret_cast<uint32_t*>(h->ssn_axis_a.p);
    uint32_t* sig_nxt = reinterpret_cast<uint32_t*>(h->ssn_axis_b.p);
    GsSet in, out;
    in.e = h->gs_e[0].p; out.e = h->gs_e[1].p;
    for (int d = 0; d < 3; ++d) { in.k[d] = h->gs_k[d].p; out.k[d] = h->gs_k[3 + d].p; }
    uint2* rng_cur = h->gs_rng.p;
    uint2* rng_nxt = h->gs_rng.p + nseg_g + 1;
    // per segment two 256-bin histograms; a level zeroes its children's (the root's: here)
    uint32_t* gh1[2] = {h->gs_hist.p, h->gs_hist.p + 256 * (nseg_g + 1)};
    uint32_t* gh2[2] = {h->gs_hist.p + 2 * 256 * (nseg_g + 1), h->gs_hist.p + 3 * 256 * (nseg_g + 1)};
    hipLaunchKernelGGL(k_gs_init, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n, in, out.e, (const SsnSeg*)cur, (const uint32_t*)h->ssn_bb.p, rng_cur,
                       gh1[0], gh2[0], cand_n[0], sig_cur, h->gs_err.p);
    for (int L = 0; L < glevels; ++L) {
      const int ns = 1 << L, nb = (int)h->gs_lvl_blocks[L], par = L & 1;
      const GsBlock* tab = h->gs_tab.p + h->gs_lvl_first[L];
      const GsSegBlocks* sblk = h->gs_sblk.p + (ns - 1);
      hipLaunchKernelGGL(k_gs_hist<1>, dim3(nb), dim3(256), 0, h->stream, tab, cur, (const uint2*)rng_cur, in, gh1[par], gh2[par], h->gs_err.p);
      hipLaunchKernelGGL(k_gs_hist<2>, dim3(nb), dim3(256), 0, h->stream, tab, cur, (const uint2*)rng_cur, in, gh1[par], gh2[par], h->gs_err.p);
      // (candidates per segment: the list's room shared out among the level's segments -- kGsCandRoom at the last level, i.e.
      //  as many as the segment has points, and so at every level above it: a wall square to a frame axis puts 12 000 points
      //  of a sub-map into one bin of the first levels, a lattice a fifth of a segment)
      const uint32_t lvl_cap = (uint32_t)std::min<size_t>(cand_room / (size_t)ns, (size_t)1 << 22);
      hipLaunchKernelGGL(k_gs_collect, dim3(nb), dim3(256), 0, h->stream, tab, cur, (const uint2*)rng_cur, (const uint32_t*)sig_cur, in,
                         (const uint32_t*)gh1[par], (const uint32_t*)gh2[par], cand_n[par], h->gs_cand.p, h->gs_cand_blk.p, lvl_cap, h->gs_cl.p);
      hipLaunchKernelGGL(k_gs_select, dim3(ns), dim3(L < 3 ? 1024 : 256), 0, h->stream, sblk, src, cur, (const uint32_t*)sig_cur, gh1[par], gh2[par],
                         gh1[par ^ 1], gh2[par ^ 1], cand_n[par], (const GsMedian*)h->gs_cand.p, (const uint32_t*)h->gs_cand_blk.p, lvl_cap, h->gs_cl.p,
                         h->gs_cl.p + cap, h->gs_median.p, nxt, sig_nxt, cand_n[par ^ 1], rng_nxt, h->gs_err.p);
      hipLaunchKernelGGL(k_gs_part, dim3(nb), dim3(kGsPartThreads), 0, h->stream, tab, cur, (const uint32_t*)sig_cur, in, out,
                         (const GsMedian*)h->gs_median.p, (const uint32_t*)(h->gs_cl.p + cap), (const SsnSeg*)nxt, rng_nxt, h->gs_err.p);
      std::swap(in, out);
      std::swap(cur, nxt);
      std::swap(sig_cur, sig_nxt);
      std::swap(rng_cur, rng_nxt);
    }
    idx = in.e;          // every root's points, in original-index order
    root_sig = sig_cur;
    HIPC(hipGetLastError());
  } else if (glevels > 0) {
    // segmented sorts (lsgpu_segsort.hip.h): per level only the segments that cut along a new axis, four passes of
    // (uint32 key, uint32 index) pairs, every segment inside its own range of the arrays
    const size_t nseg_g = (size_t)1 << glevels;
    const int cap = (int)(n / kSegTile + (int64_t)(nseg_g / 2) + 2);
    HIPC(h->sc->vals_alt.reserve(n));
    HIPC(h->ssn_axis_a.reserve(nseg_g)); HIPC(h->ssn_axis_b.reserve(nseg_g)); HIPC(h->ssn_seg_fb.reserve(nseg_g));
    HIPC(h->ssn_blocktab.reserve((size_t)cap)); HIPC(h->sc->sort_hist.reserve((size_t)256 * cap + 256 + 4));
    uint32_t* keyA = reinterpret_cast<uint32_t*>(h->sc->keys.p);
    uint32_t* keyB = keyA + n;
    uint32_t* valA = h->sc->vals.p;
    uint32_t* valB = h->sc->vals_alt.p;
    uint32_t* bh = h->sc->sort_hist.p;
    uint32_t* dtot = bh + (size_t)256 * cap;
    uint32_t* nblocks_dev = dtot + 256;
    int* ax_cur = h->ssn_axis_a.p;
    int* ax_nxt = h->ssn_axis_b.p;
    HIPC(hipMemsetAsync(ax_cur, 0xFF, sizeof(int), h->stream));   // the root's order follows no axis (-1)
    for (int L = 0; L < glevels; ++L) {
      const int ns = 1 << L;
      const int grid = (int)std::min<int64_t>(cap, n / kSegTile + ns + 1);
      hipLaunchKernelGGL(k_ssn_plan, dim3(1), dim3(256), 0, h->stream, cur, ns, knn, ax_cur, h->ssn_seg_fb.p,
                         h->ssn_blocktab.p, nblocks_dev);
      hipLaunchKernelGGL(k_ssn_keys32, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n, valA, h->ssn_seg_of.p, cur,
                         h->ssn_seg_fb.p, L == 0 ? 1 : 0, keyA);
      for (int pass = 0; pass < 4; ++pass) {
        const uint32_t* kin = (pass & 1) ? keyB : keyA; const uint32_t* vin = (pass & 1) ? valB : valA;
        uint32_t* kout = (pass & 1) ? keyA : keyB; uint32_t* vout = (pass & 1) ? valA : valB;
        hipLaunchKernelGGL(k_seg_hist<kSegItems>, dim3(grid), dim3(256), 0, h->stream, kin, h->ssn_blocktab.p, nblocks_dev, 8 * pass, bh, cap);
        hipLaunchKernelGGL(k_seg_scan, dim3(256), dim3(256), 0, h->stream, bh, cap, nblocks_dev, dtot);
        hipLaunchKernelGGL((k_seg_scatter<kSegItems>), dim3(grid), dim3(256), 0, h->stream, kin, vin, kout, vout,
                           h->ssn_blocktab.p, nblocks_dev, 8 * pass, bh, dtot, cap);
      }
      idx = valA;
      hipLaunchKernelGGL(k_ssn_split, dim3(nblk(ns)), dim3(256), 0, h->stream, src, idx, cur, ns, knn, nxt, (const int*)ax_cur, ax_nxt);
      hipLaunchKernelGGL(k_ssn_assign, dim3(nblk(n)), dim3(256), 0, h->stream, (int)n, cur, knn, h->ssn_seg_of.p, L == 0 ? 1 : 0);
      std::swap(cur, nxt);
      std::swap(ax_cur, ax_nxt);
    }
    root_axis = ax_cur;
    HIPC(hipGetLastError());
  }
  if (glevels < levels) {
    if (glevels == 0) {  // the whole cloud fits one workgroup: identity order to start from
      hipLaunchKernelGGL(k_ssn_keys, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n, (const uint32_t*)nullptr,
                         (const uint32_t*)nullptr, cur, knn, h->sc->keys.p, h->sc->vals.p);
      idx = h->sc->vals.p;
    }
    uint32_t* idx_rw = const_cast<uint32_t*>(idx);
    if (!tree_finish)
      hipLaunchKernelGGL(k_ssn_finish, dim3(1 << glevels), dim3(256), 0, h->stream, src, idx_rw, cur, knn, levels - glevels, h->ssn_seg_of.p, nxt);
    else if (root_max == 8192)
      hipLaunchKernelGGL(k_ssn_tree<8192>, dim3(1 << glevels), dim3(1024), 0, h->stream, src, idx_rw, cur, knn, levels - glevels, h->ssn_seg_of.p, nxt, root_axis, root_sig);
    else if (root_max == 4096)
      hipLaunchKernelGGL(k_ssn_tree<4096>, dim3(1 << glevels), dim3(512), 0, h->stream, src, idx_rw, cur, knn, levels - glevels, h->ssn_seg_of.p, nxt, root_axis, root_sig);
    else
      hipLaunchKernelGGL(k_ssn_tree<2048>, dim3(1 << glevels), dim3(256), 0, h->stream, src, idx_rw, cur, knn, levels - glevels, h->ssn_seg_of.p, nxt, root_axis, root_sig);
    std::swap(cur, nxt);
  }
  if (levels == 0) {  // a single box: identity order
    hipLaunchKernelGGL(k_ssn_keys, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, cur, knn, h->sc->keys.p, h->sc->vals.p);
    HIPC(hipMemsetAsync(h->ssn_seg_of.p, 0, (size_t)n * 4, h->stream));
    idx = h->sc->vals.p;
  }
  hipLaunchKernelGGL(k_ssn_boxes, dim3((int)((nseg + 127) / 128)), dim3(128), 0, h->stream, src, idx, cur, (int)nseg,
                     h->ssn_box_normal.p, h->ssn_box_pts.p);
  HIPC(hipGetLastError());
  int rc = scan_u32(h, h->ssn_box_pts.p, h->ssn_box_base.p, nseg);
  if (rc) return rc;
  // the draws: at most one per point; produced on the helper thread while the kernels above were enqueued
  rc = ahead->ready(first_draw + (size_t)n);
  if (rc) return rc;
  hipLaunchKernelGGL(k_ssn_select, dim3(nblk(n)), dim3(256), 0, h->stream, (int)n, idx, h->ssn_seg_of.p, cur,
                     h->ssn_box_pts.p, h->ssn_box_base.p, h->ssn_draws.p + first_draw, ratio, h->ssn_keep.p);
  rc = scan_u32(h, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n, false, /*nonzero*/ true);
  if (rc == LSGPU_OK) {
    hipLaunchKernelGGL(k_ssn_emit, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n,
                       h->ssn_box_normal.p, h->ssn_keep.p, h->ssn_out_pos.p, out_xyz1, out_nrm);
  }
  uint32_t n_draws = 0, kept = 0;
  if (rc == LSGPU_OK)
    // (the sort-free levels' error words travel with the totals)
    rc = scan_totals(h, h->ssn_box_pts.p, h->ssn_box_base.p, nseg, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n, &n_draws, &kept,
                     select_levels ? h->gs_err.p : nullptr, h->h_gs_err, 8 * sizeof(uint32_t), /*b_flags*/ true);
  if (rc) return rc;
  if (select_levels && *h->h_gs_err) {
    if (getenv("LSGPU_GS_DEBUG")) fprintf(stderr, "lsgpu: sort-free levels gave up (n %lld): code %u seg %u a %u b %u | part %u seg %u dst %u\n", (long long)n,
                                          h->h_gs_err[1], h->h_gs_err[2], h->h_gs_err[3], h->h_gs_err[4], h->h_gs_err[5], h->h_gs_err[6], h->h_gs_err[7]);
    // thousands of equal coordinates around a median (more candidates than a workgroup selects among): the segmented
    // sorts do not care -- the same filter again with them (same draws: nothing has been consumed yet).  Counted:
    // lsgpu_icp_get_policy_info shows a handle that pays the filter twice on every call.
    ++h->ssn_sort_fallbacks;
    return ssn_device(h, src, n, knn, ratio, seed, out_xyz1, out_nrm, n_out, ahead, true);
  }
  ahead->used = first_draw + (size_t)n_draws;  // dropped boxes drew nothing
  HIPC(hipGetLastError());
  *n_out = kept;
  return LSGPU_OK;
}

// RandomSampling on device memory (order preserved)
// (defer_wait: everything is enqueued, the host's wait for the number of points kept is left to random_sampling_wait --
// lsgpu_icp_compute puts the rest of the grid build on the other stream in between)
static int random_sampling_wait(lsgpu_icp* h, int64_t* n_out) {
  uint32_t unused = 0, kept = 0;
  const int rc = scan_totals_wait(h, &unused, &kept);
  if (rc) return rc;
  *n_out = kept;
  return LSGPU_OK;
}
static int random_sampling_device(lsgpu_icp* h, const float4* src, int64_t n, float prob, int64_t seed,
                                  float4* out_xyz1, int64_t* n_out, DrawAhead* ahead = nullptr, bool defer_wait = false) {
  *n_out = 0;
  HIPC(h->ssn_keep.reserve(n));
  HIPC(h->ssn_out_pos.reserve(n));
  DrawAhead own;
  int rc = LSGPU_OK;
  if (!ahead) {
    rc = own.begin(h, seed, (size_t)n);
    if (rc) return rc;
    ahead = &own;
  }
  const size_t first_draw = ahead->used;
  ahead->used = first_draw + (size_t)n;  // one draw per point, whatever happens next
  rc = ahead->ready();
  if (rc) return rc;
  hipLaunchKernelGGL(k_draw_select, dim3(nblk(n)), dim3(256), 0, h->cur, (int)n, h->ssn_draws.p + first_draw, prob, h->ssn_keep.p);
  rc = scan_u32(h, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n);
  if (rc) return rc;
  hipLaunchKernelGGL(k_compact_points, dim3(nblk(n)), dim3(256), 0, h->cur, src, (int)n, h->ssn_keep.p,
                     h->ssn_out_pos.p, out_xyz1);
  HIPC(hipGetLastError());
  rc = scan_totals_enqueue(h, nullptr, nullptr, 0, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n);
  if (rc || defer_wait) return rc;
  return random_sampling_wait(h, n_out);
}

extern "C" {

int lsgpu_icp_filter_reference(lsgpu_icp* h, const float* xyz1, int64_t n, int knn, float ratio,
                               int64_t seed, float* out_xyz1, float* out_normals, int64_t* n_out) {
  if (!h || !n_out || !out_xyz1 || !out_normals) return LSGPU_BAD_ARG;
  h->err.clear();
  *n_out = 0;
  if (knn < 3 || knn > kSsnMaxKnn) { h->err = "filter_reference: knn must be in [3, 32]"; return LSGPU_BAD_ARG; }
  if (seed >= 0) DrawStream::global().take(seed, 0, nullptr);
  if (n <= 0 || !xyz1) return LSGPU_OK;
  if (n > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  const float4* src = nullptr;
  int rc = stage_points(h, xyz1, n, h->flt_in, &src);
  if (rc) return rc;
  const bool dev_x = is_device_ptr(out_xyz1), dev_n = is_device_ptr(out_normals);
  float4* ox = reinterpret_cast<float4*>(out_xyz1);
  float* on = out_normals;
  if (!dev_x) { HIPC(h->flt_ref.reserve(n)); ox = h->flt_ref.p; }
  if (!dev_n) { HIPC(h->flt_nrm.reserve(3 * n)); on = h->flt_nrm.p; }
  rc = ssn_device(h, src, n, knn, ratio, -1, ox, on, n_out);
  if (rc) return rc;
  if (!dev_x && *n_out) HIPC(hipMemcpyAsync(out_xyz1, ox, (size_t)*n_out * 16, hipMemcpyDeviceToHost, h->stream));
  if (!dev_n && *n_out) HIPC(hipMemcpyAsync(out_normals, on, (size_t)*n_out * 12, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

int lsgpu_icp_filter_reading(lsgpu_icp* h, const float* xyz1, int64_t n, float prob, int64_t seed,
                             float* out_xyz1, int64_t* n_out) {
  if (!h || !n_out || !out_xyz1) return LSGPU_BAD_ARG;
  h->err.clear();
  *n_out = 0;
  if (seed >= 0) DrawStream::global().take(seed, 0, nullptr);
  if (n <= 0 || !xyz1) return LSGPU_OK;
  if (n > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  const float4* src = nullptr;
  int rc = stage_points(h, xyz1, n, h->flt_in, &src);
  if (rc) return rc;
  const bool dev_x = is_device_ptr(out_xyz1);
  float4* ox = reinterpret_cast<float4*>(out_xyz1);
  if (!dev_x) { HIPC(h->flt_rd.reserve(n)); ox = h->flt_rd.p; }
  rc = random_sampling_device(h, src, n, prob, -1, ox, n_out);
  if (rc) return rc;
  if (!dev_x && *n_out) HIPC(hipMemcpyAsync(out_xyz1, ox, (size_t)*n_out * 16, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

int lsgpu_icp_compute(lsgpu_icp* h, const float* reading_xyz1, int64_t nq, const float* reference_xyz1,
                      int64_t nr, const float T_init[16], const lsgpu_chain_config* chain,
                      float T_out[16], lsgpu_icp_stats* stats) {
  if (!h || !T_init || !T_out || !chain) return LSGPU_BAD_ARG;
  h->err.clear();
  std::memcpy(T_out, T_init, 16 * sizeof(float));
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (chain->ssn_knn < 3 || chain->ssn_knn > kSsnMaxKnn) { h->err = "compute: ssn_knn must be in [3, 32]"; return LSGPU_BAD_CONFIG; }
  if (chain->seed >= 0) DrawStream::global().take(chain->seed, 0, nullptr);
  if (nq <= 0 || nr <= 0 || !reading_xyz1 || !reference_xyz1) { h->err = "compute: empty cloud"; return LSGPU_NO_CONVERGENCE; }
  if (nq > 0x7FFFFFF0ll || nr > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  const double t0 = wall_ms();
  // the draws of both filters, produced on a helper thread from now on: at most one per reference point, then one per
  // reading point
  DrawAhead draws;
  {
    const int rc0 = draws.begin(h, -1, (size_t)nr + (chain->reading_prob < 0.f ? (size_t)0 : (size_t)nq), (size_t)nr);   // (the reference filter's share first)
    if (rc0) return rc0;
  }
  // step 1: reference filter (yaml:5-7)
  const float4* src = nullptr;
  int rc = stage_points(h, reference_xyz1, nr, h->flt_in, &src);
  if (rc) return rc;
  // A reading handed over in HOST memory crosses PCIe while the reference is being filtered: its own stream, and its own
  // host thread, because a copy from pageable memory keeps the calling thread until the last chunk is staged
  // (SURVEY.md §8d counts H2D in scans/s).  The loop's stream waits for it right before the reading filter.
  std::thread uploader;
  hipError_t upload_err = hipSuccess;
  const bool overlap_upload = !is_device_ptr(reading_xyz1);
  if (overlap_upload) {
    if (!h->copy_stream) HIPC(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!h->copy_done) HIPC(hipEventCreateWithFlags(&h->copy_done, hipEventDisableTiming));
    float4* up_dst = h->upload_into;     // (a cloud slot: lsgpu_icp_compute_clouds_upload)
    if (!up_dst) { HIPC(h->flt_in2.reserve(nq)); up_dst = h->flt_in2.p; }
    order_after_tail(h, h->copy_stream);
    // the reference's own upload goes first: it is in front of everything, the reading is not needed before the reading
    // filter, and two copies at once share the link (from pinned buffers they did: the pinned path was the slower one)
    if (src != reinterpret_cast<const float4*>(reference_xyz1)) {
      if (!h->ref_up_done) HIPC(hipEventCreateWithFlags(&h->ref_up_done, hipEventDisableTiming));
      HIPC(hipEventRecord(h->ref_up_done, h->stream));
      HIPC(hipStreamWaitEvent(h->copy_stream, h->ref_up_done, 0));
    }
    auto upload = [&, up_dst] {
      hipError_t e = hipSetDevice(h->device);
      if (e == hipSuccess) e = hipMemcpyAsync(up_dst, reading_xyz1, (size_t)nq * 16, hipMemcpyHostToDevice, h->copy_stream);
      if (e == hipSuccess) e = hipEventRecord(h->copy_done, h->copy_stream);
      upload_err = e;
      if (e == hipSuccess) h->upload_done = true;
    };
    try {
      uploader = std::thread(upload);
    } catch (const std::system_error&) {   // no thread to be had: copy here (no overlap)
      upload();
    }
  }
  // every return path joins the uploader and drains its stream: a pinned host reading is read by DMA, and the caller
  // may free or reuse it as soon as this call has returned, error or not (after a completed alignment the copy is long
  // over and the wait returns at once)
  struct Joiner {
    std::thread& t; lsgpu_icp* h; bool started;
    ~Joiner() {
      if (t.joinable()) t.join();
      if (started && h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    }
  } joiner{uploader, h, overlap_upload};
  HIPC(h->flt_ref.reserve(nr));
  HIPC(h->flt_nrm.reserve(3 * nr));
  int64_t nrf = 0, nqf = 0;
  rc = ssn_device(h, src, nr, chain->ssn_knn, chain->ssn_ratio, -1, h->flt_ref.p, h->flt_nrm.p, &nrf, &draws);
  if (rc) return rc;
  if (nrf <= 0) { h->err = "compute: the reference filter left no point"; h->nr = 0; return LSGPU_NO_CONVERGENCE; }
  // the reading filter draws once per reading point, whatever it keeps: the total is known, the stream can go
  draws.commit_now(draws.used + (chain->reading_prob < 0.f ? (size_t)0 : (size_t)nq));
  // steps 2-4.  The grid build (steps 2-3, h->stream) and the reading's side -- its filter (step 4) and the ordering of
  // the queries (the first part of step 5) -- do not depend on each other: the latter is enqueued on a second stream,
  // with its own sort scratch, from inside set_reference (right before its one host round trip), and the queries are
  // moved into the reference's frame as soon as set_reference knows the mean.  Both are chains of short launches that
  // leave most of the chip idle; side by side the shorter one disappears (LSGPU_NO_SIDE_STREAM: one after the other).
  const bool side = tuning().side_stream && !h->comm;
  const float4* rd_src = nullptr;
  const float4* rd_dev = nullptr;
  auto reading_ready = [&](hipStream_t on) -> int {   // the reading's upload, if it is ours, has to be there
    if (overlap_upload) {
      if (uploader.joinable()) uploader.join();
      if (upload_err != hipSuccess) { h->err = std::string("compute: reading upload: ") + hipGetErrorString(upload_err); (void)hipGetLastError(); return LSGPU_HIP_ERROR; }
      HIPC(hipStreamWaitEvent(on, h->copy_done, 0));
      rd_src = h->upload_into ? h->upload_into : h->flt_in2.p;
    } else {
      rd_src = reinterpret_cast<const float4*>(reading_xyz1);
    }
    return LSGPU_OK;
  };
  auto reading_filter = [&](bool defer_wait) -> int {   // step 4: reading filter (yaml:1-3), on h->cur
    rd_dev = rd_src;
    if (chain->reading_prob < 0.f) {
      // no readingDataPointsFilters section: upstream runs no module at all -- every point, NO rand() call (a
      // RandomSampling module with prob 1 would consume nq draws and drop the points whose draw rounds to 1.0f)
      nqf = nq;
      return LSGPU_OK;
    }
    HIPC(h->flt_rd.reserve(nq));
    const int r = random_sampling_device(h, rd_src, nq, chain->reading_prob, -1, h->flt_rd.p, &nqf, &draws, defer_wait);
    rd_dev = h->flt_rd.p;
    return r;
  };
  struct SideGuard {   // whatever happens, the helpers go back to the main stream and the side stream is drained
    lsgpu_icp* h; bool used = false;
    void enter() { used = true; h->cur = h->side_stream; h->sc = &h->scr_side; h->side_totals_slot = 1; }
    void leave() { h->cur = h->stream; h->sc = &h->scr_main; h->side_totals_slot = 0; }
    ~SideGuard() {
      leave();
      h->hook_before_ref_sync = nullptr; h->hook_after_grid = nullptr;
      if (used && h->side_stream) (void)hipStreamSynchronize(h->side_stream);
    }
  } side_guard{h};
  bool side_prepared = false;
  if (side) {
    // (Tried: this stream at the lowest stream priority, so that the loop's short kernels never wait for a compute unit behind
    // the direction index's build.  Measured slower, 5.27 -> 5.41..5.60 ms per compute from host buffers, level from resident
    // ones: the reading's filter and query order run here too and ARE on the critical path.)
    if (!h->side_stream) HIPC(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    if (!h->side_done) HIPC(hipEventCreateWithFlags(&h->side_done, hipEventDisableTiming));
    order_after_tail(h, h->side_stream);
    // Order of the host's work (round 5, from rocprofv3's timeline of a step).  The host thread that enqueues both chains
    // is the scarce resource here (~4 us per launch), and what it waits for decides what idles:
    //   1. the reading's filter goes out NOW, in front of the whole grid build (it only needs the draws and the upload);
    //   2. the ordering of the queries (~25 launches) goes out while the first half of the grid build runs, right before
    //      set_reference waits for its cell counts -- the number of reading points kept is long there;
    //   3. the second half of the grid build goes out right behind that wait;
    //   4. the move of the queries into the reference's frame (it needs the mean) behind it.
    // Before, the host enqueued the reading's whole side -- a wait and ~60 launches -- between the grid's two halves: the
    // loop's stream sat idle for 0.3 ms behind the cell counts, then the loop waited for the queries.
    side_guard.enter();
    rc = reading_ready(h->side_stream);
    if (!rc) rc = reading_filter(/*defer_wait*/ true);
    side_guard.leave();
    if (rc) return rc;
    h->hook_before_ref_sync = [&]() -> int {
      side_guard.enter();
      int r = LSGPU_OK;
      if (chain->reading_prob >= 0.f) r = random_sampling_wait(h, &nqf);   // (the number of points kept)
      if (!r && nqf > 0) r = prepare_queries(h, reinterpret_cast<const float*>(rd_dev), nqf, Mat34{}, /*gather*/ false);
      side_guard.leave();
      return r;
    };
    h->hook_after_grid = [&]() -> int {
      if (nqf <= 0) return LSGPU_OK;
      float T_rm_in[16];
      std::memcpy(T_rm_in, T_init, sizeof(T_rm_in));
      for (int d = 0; d < 3; ++d) T_rm_in[12 + d] = T_init[12 + d] - h->mean[d];   // (as lsgpu_icp_align, step 5)
      hipLaunchKernelGGL(k_query_gather, dim3(nblk(nqf)), dim3(256), 0, h->side_stream, rd_dev, nqf, h->scr_side.vals_alt.p,
                         to_mat34(T_rm_in), h->rdq.p);
      HIPC(hipGetLastError());
      HIPC(hipEventRecord(h->side_done, h->side_stream));
      side_prepared = true;
      return LSGPU_OK;
    };
  }
  h->defer_cone = side;
  rc = lsgpu_icp_set_reference(h, reinterpret_cast<const float*>(h->flt_ref.p), h->flt_nrm.p, nrf);
  h->hook_before_ref_sync = nullptr; h->hook_after_grid = nullptr;
  h->defer_cone = false;
  if (rc) return rc;
  if (side) {
    // the direction index of the reference is not needed before the loop's third search: lsgpu_icp_align enqueues its
    // build on the side stream (behind the queries' order, with that stream's sort scratch) once the loop's first
    // iteration is out -- beside the first two iterations instead of in front of the loop (0.16 ms per 1 M-point
    // compute), and without holding the host back from starting the loop (round 5)
    if (!h->cone_done) HIPC(hipEventCreateWithFlags(&h->cone_done, hipEventDisableTiming));
    h->cone_build_in_align = true;
  }
  if (!side) {
    rc = reading_ready(h->stream);
    if (!rc) rc = reading_filter(false);
    if (rc) return rc;
  }
  draws.finish();
  const double t_filters = wall_ms() - t0;
  if (nqf <= 0) { h->err = "compute: the reading filter left no point"; return LSGPU_NO_CONVERGENCE; }
  if (side_prepared) { h->prepared_rd = reinterpret_cast<const float*>(rd_dev); h->prepared_nq = nqf; }
  // steps 5-7
  rc = lsgpu_icp_align(h, reinterpret_cast<const float*>(rd_dev), nqf, T_init, T_out, stats);
  if (stats) stats->t_reserved[0] = t_filters;
  return rc;
}

int lsgpu_filter_cylinder(lsgpu_icp* h, const float* xyz1, int64_t n, const float center[3], double radius_m,
                          double height_m, int remove_point_inside, float* out_xyz1, int64_t* n_out) {
  if (!h || !n_out || !center || !out_xyz1) return LSGPU_BAD_ARG;
  h->err.clear();
  *n_out = 0;
  if (n <= 0 || !xyz1) return LSGPU_OK;
  if (n > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  const float4* src = nullptr;
  int rc = stage_points(h, xyz1, n, h->flt_in, &src);
  if (rc) return rc;
  const bool dev_x = is_device_ptr(out_xyz1);
  float4* ox = reinterpret_cast<float4*>(out_xyz1);
  if (!dev_x) { HIPC(h->flt_rd.reserve(n)); ox = h->flt_rd.p; }
  HIPC(h->ssn_keep.reserve(n));
  HIPC(h->ssn_out_pos.reserve(n));
  hipLaunchKernelGGL(k_cylinder_select, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n, center[0], center[1],
                     center[2], radius_m * radius_m, height_m / 2.0, remove_point_inside, h->ssn_keep.p);
  rc = scan_u32(h, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n);
  if (rc) return rc;
  hipLaunchKernelGGL(k_compact_points, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n, h->ssn_keep.p,
                     h->ssn_out_pos.p, ox);
  HIPC(hipGetLastError());
  uint32_t unused = 0, kept = 0;
  rc = scan_totals(h, nullptr, nullptr, 0, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n, &unused, &kept);
  if (rc) return rc;
  *n_out = kept;
  if (!dev_x && kept) HIPC(hipMemcpyAsync(out_xyz1, ox, (size_t)kept * 16, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

int lsgpu_filter_voxel_grid(lsgpu_icp* h, const float* xyz1, int64_t n, const float leaf[3], int min_points,
                            float* out_xyz1, int64_t* n_out) {
  if (!h || !n_out || !leaf || !out_xyz1) return LSGPU_BAD_ARG;
  h->err.clear();
  *n_out = 0;
  if (!(leaf[0] > 0.f && leaf[1] > 0.f && leaf[2] > 0.f)) { h->err = "voxel_grid: leaf size must be positive"; return LSGPU_BAD_ARG; }
  if (n <= 0 || !xyz1) return LSGPU_OK;
  if (n > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  const float4* src = nullptr;
  int rc = stage_points(h, xyz1, n, h->flt_in, &src);
  if (rc) return rc;
  // getMinMax3D
  HIPC(h->ssn_bb.reserve(8));
  HIPC(hipMemsetAsync(h->ssn_bb.p, 0xFF, 12, h->stream));
  HIPC(hipMemsetAsync(h->ssn_bb.p + 3, 0, 12, h->stream));
  hipLaunchKernelGGL(k_ssn_bounds, dim3(std::min(nblk(n), 256)), dim3(256), 0, h->stream, src, (int)n, h->ssn_bb.p);
  uint32_t* hb = reinterpret_cast<uint32_t*>(h->h_pinned + 104);
  HIPC(hipMemcpyAsync(hb, h->ssn_bb.p, 24, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  auto from_key = [](uint32_t k) { const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; std::memcpy(&f, &u, 4); return f; };
  float inv[3];
  int minb[3], divb[3];
  for (int d = 0; d < 3; ++d) {
    inv[d] = 1.0f / leaf[d];
    minb[d] = (int)std::floor(from_key(hb[d]) * inv[d]);
    const int maxb = (int)std::floor(from_key(hb[3 + d]) * inv[d]);
    divb[d] = maxb - minb[d] + 1;
  }
  if ((int64_t)divb[0] * (int64_t)divb[1] * (int64_t)divb[2] > 2147483647ll) {
    h->err = "voxel_grid: leaf size too small for the cloud, the voxel index would overflow";
    return LSGPU_BAD_ARG;
  }
  HIPC(h->sc->keys.reserve(n));
  HIPC(h->sc->vals.reserve(n));
  HIPC(h->ssn_keep.reserve(n));
  HIPC(h->ssn_out_pos.reserve(n));
  HIPC(h->ssn_seg_of.reserve(n));   // voxel head flags
  HIPC(h->flt_ref.reserve(n));      // centroids by sorted position
  hipLaunchKernelGGL(k_voxel_keys, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n, inv[0], inv[1], inv[2],
                     minb[0], minb[1], minb[2], divb[0], divb[0] * divb[1], h->sc->keys.p, h->sc->vals.p);
  rc = sort_pairs(h, n, 31);  // stable: equal voxels keep input order
  if (rc) return rc;
  hipLaunchKernelGGL(k_voxel_heads, dim3(nblk(n)), dim3(256), 0, h->stream, h->sc->keys_alt.p, (int)n, h->ssn_seg_of.p);
  hipLaunchKernelGGL(k_voxel_centroids, dim3(nblk(n)), dim3(256), 0, h->stream, src, h->sc->keys_alt.p, h->sc->vals_alt.p, (int)n,
                     min_points, h->ssn_seg_of.p, h->flt_ref.p, h->ssn_keep.p);
  rc = scan_u32(h, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n);
  if (rc) return rc;
  const bool dev_x = is_device_ptr(out_xyz1);
  float4* ox = reinterpret_cast<float4*>(out_xyz1);
  if (!dev_x) { HIPC(h->flt_rd.reserve(n)); ox = h->flt_rd.p; }
  hipLaunchKernelGGL(k_compact_points, dim3(nblk(n)), dim3(256), 0, h->stream, h->flt_ref.p, (int)n, h->ssn_keep.p,
                     h->ssn_out_pos.p, ox);
  HIPC(hipGetLastError());
  uint32_t unused = 0, kept = 0;
  rc = scan_totals(h, nullptr, nullptr, 0, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n, &unused, &kept);
  if (rc) return rc;
  *n_out = kept;
  if (!dev_x && kept) HIPC(hipMemcpyAsync(out_xyz1, ox, (size_t)kept * 16, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

int lsgpu_apply_point_filters(lsgpu_icp* h, lsgpu_point_filter* filters, int n_filters, const float* xyz1,
                              int64_t n, int64_t seed, float* out_xyz1, int64_t* n_out) {
  if (!h || !n_out || n_filters < 0 || (n_filters > 0 && !filters) || n < 0 || (n > 0 && (!xyz1 || !out_xyz1))) return LSGPU_BAD_ARG;
  h->err.clear();
  *n_out = 0;
  if (n > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  for (int k = 0; k < n_filters; ++k) {
    const lsgpu_point_filter& f = filters[k];
    const bool ok = (f.type == LSGPU_FILTER_MAX_DIST || f.type == LSGPU_FILTER_MIN_DIST) ? (f.dim >= -1 && f.dim <= 2)
                    : f.type == LSGPU_FILTER_BOUNDING_BOX ? true
                    : f.type == LSGPU_FILTER_FIX_STEP_SAMPLING ? (f.v[0] >= 1.f && f.v[1] >= 1.f && f.v[2] > 0.f)
                    : f.type == LSGPU_FILTER_RANDOM_SAMPLING ? (f.v[0] >= 0.f && f.v[0] <= 1.f)
                    : f.type == LSGPU_FILTER_REMOVE_NAN ? true : false;
    if (!ok) { h->err = "apply_point_filters: unknown filter type or parameter out of range"; return LSGPU_BAD_CONFIG; }
  }
  if (seed >= 0) DrawStream::global().take(seed, 0, nullptr);
  // DataPointsFilters::apply: an empty CHAIN is a no-op; otherwise the first filter that is handed an empty cloud throws
  // ConvergenceError("no points to filter") -- also when the cloud came in empty
  if (n == 0) {
    if (n_filters == 0) return LSGPU_OK;
    h->err = "apply_point_filters: no points to filter";
    return LSGPU_NO_CONVERGENCE;
  }
  HIPC(hipSetDevice(h->device));
  const float4* cur = nullptr;
  int rc = stage_points(h, xyz1, n, h->flt_in, &cur);
  if (rc) return rc;
  HIPC(h->flt_ref.reserve(n));
  HIPC(h->flt_rd.reserve(n));
  HIPC(h->ssn_keep.reserve(n));
  HIPC(h->ssn_out_pos.reserve(n));
  float4* ping = h->flt_ref.p;
  float4* pong = h->flt_rd.p;
  int64_t m = n;
  for (int k = 0; k < n_filters; ++k) {
    lsgpu_point_filter& f = filters[k];
    if (m == 0) { h->err = "apply_point_filters: no points to filter"; return LSGPU_NO_CONVERGENCE; }
    PointFilterDev d;
    std::memset(&d, 0, sizeof(d));
    d.type = f.type; d.dim = f.dim; d.flag = f.flag;
    for (int i = 0; i < 6; ++i) d.v[i] = f.v[i];
    d.step = 1; d.phase = 0;
    if (f.type == LSGPU_FILTER_FIX_STEP_SAMPLING) {
      double step = f.state > 0.0 ? f.state : (double)f.v[0];
      const int istep = std::max(1, (int)step);
      d.step = (uint32_t)istep;
      d.phase = DrawStream::global().take_raw(-1) % (uint32_t)istep;  // rand() % iStep
      const double delta = (double)f.v[0] * (double)f.v[2] - (double)f.v[0];
      step *= (double)f.v[2];
      if (delta >= 0 && step > (double)f.v[1]) step = (double)f.v[1];
      if (delta < 0 && step < (double)f.v[1]) step = (double)f.v[1];
      f.state = step;
    } else if (f.type == LSGPU_FILTER_RANDOM_SAMPLING) {
      rc = upload_draws_begin(h, -1, (size_t)m);
      if (rc) return rc;
      DrawStream::global().commit((size_t)m);  // one draw per point
    }
    hipLaunchKernelGGL(k_point_filter_select, dim3(nblk(m)), dim3(256), 0, h->stream, cur, (int)m, d, h->ssn_draws.p,
                       h->ssn_keep.p);
    rc = scan_u32(h, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)m);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compact_points, dim3(nblk(m)), dim3(256), 0, h->stream, cur, (int)m, h->ssn_keep.p,
                       h->ssn_out_pos.p, ping);
    HIPC(hipGetLastError());
    uint32_t unused = 0, kept = 0;
    rc = scan_totals(h, nullptr, nullptr, 0, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)m, &unused, &kept);
    if (rc) return rc;
    m = kept;
    cur = ping;
    std::swap(ping, pong);
  }
  *n_out = m;
  if (m) HIPC(hipMemcpyAsync(out_xyz1, cur, (size_t)m * 16, hipMemcpyDefault, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

int lsgpu_cloud_from_pointcloud2(lsgpu_icp* h, const unsigned char* data, int64_t n, int point_step, int off_x,
                                 int off_y, int off_z, int is_bigendian, int drop_non_finite, float* out_xyz1,
                                 int64_t* n_out) {
  if (!h || !n_out || n < 0 || (n > 0 && (!data || !out_xyz1))) return LSGPU_BAD_ARG;
  h->err.clear();
  *n_out = 0;
  const int lo = std::min(off_x, std::min(off_y, off_z)), hi = std::max(off_x, std::max(off_y, off_z));
  if (point_step < 12 || lo < 0 || hi + 4 > point_step) { h->err = "from_pointcloud2: x/y/z fields do not fit a record"; return LSGPU_BAD_ARG; }
  if (n == 0) return LSGPU_OK;
  if (n > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  HIPC(hipSetDevice(h->device));
  const unsigned char* src = data;
  if (!is_device_ptr(data)) {  // the message's byte block crosses PCIe once, as it is
    HIPC(h->sc->sort_tmp.reserve((size_t)n * (size_t)point_step));
    HIPC(hipMemcpyAsync(h->sc->sort_tmp.p, data, (size_t)n * (size_t)point_step, hipMemcpyHostToDevice, h->stream));
    src = reinterpret_cast<const unsigned char*>(h->sc->sort_tmp.p);
  }
  HIPC(h->flt_in.reserve(n));
  HIPC(h->ssn_keep.reserve(n));
  HIPC(h->ssn_out_pos.reserve(n));
  hipLaunchKernelGGL(k_pc2_unpack, dim3(nblk(n)), dim3(256), 0, h->stream, src, (int)n, point_step, off_x, off_y, off_z,
                     is_bigendian ? 1 : 0, drop_non_finite ? 1 : 0, h->flt_in.p, h->ssn_keep.p);
  const bool dev_x = is_device_ptr(out_xyz1);
  int64_t m = n;
  const float4* res = h->flt_in.p;
  if (drop_non_finite) {
    HIPC(h->flt_rd.reserve(n));
    int rc = scan_u32(h, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compact_points, dim3(nblk(n)), dim3(256), 0, h->stream, h->flt_in.p, (int)n, h->ssn_keep.p,
                       h->ssn_out_pos.p, h->flt_rd.p);
    uint32_t unused = 0, kept = 0;
    rc = scan_totals(h, nullptr, nullptr, 0, h->ssn_keep.p, h->ssn_out_pos.p, (size_t)n, &unused, &kept);
    if (rc) return rc;
    m = kept;
    res = h->flt_rd.p;
  }
  HIPC(hipGetLastError());
  *n_out = m;
  if (m) HIPC(hipMemcpyAsync(out_xyz1, res, (size_t)m * 16, dev_x ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

int lsgpu_cloud_to_pointxyz(lsgpu_icp* h, const float* xyz1, int64_t n, unsigned char* out_data) {
  if (!h || n < 0 || (n > 0 && (!xyz1 || !out_data))) return LSGPU_BAD_ARG;
  h->err.clear();
  if (n == 0) return LSGPU_OK;
  HIPC(hipSetDevice(h->device));
  const float4* src = nullptr;
  int rc = stage_points(h, xyz1, n, h->flt_in, &src);
  if (rc) return rc;
  const bool dev_o = is_device_ptr(out_data);
  float4* dst = reinterpret_cast<float4*>(out_data);
  if (!dev_o) { HIPC(h->flt_rd.reserve(n)); dst = h->flt_rd.p; }
  hipLaunchKernelGGL(k_pc2_pack, dim3(nblk(n)), dim3(256), 0, h->stream, src, n, dst);
  HIPC(hipGetLastError());
  if (!dev_o) HIPC(hipMemcpyAsync(out_data, dst, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return LSGPU_OK;
}

int lsgpu_cloud_upload(lsgpu_icp* h, int slot, const float* xyz1, int64_t n) {
  if (!h || slot < 0 || slot >= (1 << 20) || n < 0 || n > 0x7FFFFFF0ll || (n > 0 && !xyz1)) return LSGPU_BAD_ARG;
  h->err.clear();
  HIPC(hipSetDevice(h->device));
  if ((size_t)slot >= h->clouds.size()) { h->clouds.resize((size_t)slot + 1); h->cloud_n.resize((size_t)slot + 1, -1); }
  h->cloud_n[slot] = -1;   // (whatever the slot held is gone from here on: a failed copy must not leave a half-written cloud behind)
  HIPC(h->clouds[slot].reserve(n ? n : 1));
  if (n) HIPC(hipMemcpyAsync(h->clouds[slot].p, xyz1, (size_t)n * 16, hipMemcpyDefault, h->stream));
  HIPC(hipStreamSynchronize(h->stream));  // the caller's buffer is free again on return
  h->cloud_n[slot] = n;
  return LSGPU_OK;
}

int lsgpu_cloud_release(lsgpu_icp* h, int slot) {
  if (!h || slot < 0) return LSGPU_BAD_ARG;
  if ((size_t)slot >= h->clouds.size() || h->cloud_n[slot] < 0) return LSGPU_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->clouds[slot].release();
  h->cloud_n[slot] = -1;
  return LSGPU_OK;
}

int lsgpu_cloud_size(lsgpu_icp* h, int slot, int64_t* n) {
  if (!h || !n || slot < 0) return LSGPU_BAD_ARG;
  *n = (size_t)slot < h->clouds.size() ? h->cloud_n[slot] : -1;
  return LSGPU_OK;
}

// sub-map assembly (laser_track.cpp:474-486) on the device: submap = concat_i ( T_i * cloud ref_slots[i] ), `total` points
static int assemble_submap(lsgpu_icp* h, const int* ref_slots, const float* ref_T, int n_ref, int64_t total) {
  HIPC(h->submap.reserve(total));
  int64_t off = 0;
  for (int i = 0; i < n_ref; ++i) {
    const int64_t n = h->cloud_n[ref_slots[i]];
    if (!n) continue;
    const float* T = ref_T ? ref_T + 16 * i : nullptr;
    bool identity = !T;
    if (T) {
      identity = true;
      for (int k = 0; k < 16; ++k) identity = identity && T[k] == ((k % 5 == 0) ? 1.f : 0.f);
    }
    if (identity) {
      HIPC(hipMemcpyAsync(h->submap.p + off, h->clouds[ref_slots[i]].p, (size_t)n * 16, hipMemcpyDeviceToDevice, h->stream));
    } else {
      if (!lsgpu_check_rigid(T)) { h->err = "compute_clouds: a sub-map transform is not rigid"; return LSGPU_BAD_ARG; }
      hipLaunchKernelGGL(k_transform, dim3(nblk(n)), dim3(256), 0, h->stream, h->clouds[ref_slots[i]].p, n,
                         to_mat34(T), h->submap.p + off);
    }
    off += n;
  }
  HIPC(hipGetLastError());
  return LSGPU_OK;
}

int lsgpu_icp_compute_clouds(lsgpu_icp* h, int reading_slot, const int* ref_slots, const float* ref_T,
                             int n_ref, const float T_init[16], const lsgpu_chain_config* chain,
                             float T_out[16], lsgpu_icp_stats* stats) {
  if (!h || !T_init || !T_out || !chain || n_ref < 0 || (n_ref > 0 && !ref_slots)) return LSGPU_BAD_ARG;
  h->err.clear();
  std::memcpy(T_out, T_init, 16 * sizeof(float));
  if (stats) std::memset(stats, 0, sizeof(*stats));
  auto have = [&](int s) { return s >= 0 && (size_t)s < h->clouds.size() && h->cloud_n[s] >= 0; };
  if (!have(reading_slot)) { h->err = "compute_clouds: empty reading slot"; return LSGPU_BAD_ARG; }
  int64_t total = 0;
  for (int i = 0; i < n_ref; ++i) {
    if (!have(ref_slots[i])) { h->err = "compute_clouds: empty reference slot"; return LSGPU_BAD_ARG; }
    total += h->cloud_n[ref_slots[i]];
  }
  if (total > 0x7FFFFFF0ll) return LSGPU_BAD_ARG;
  if (total <= 0 || h->cloud_n[reading_slot] <= 0) {
    if (chain->seed >= 0) DrawStream::global().take(chain->seed, 0, nullptr);
    h->err = "compute: empty cloud";
    return LSGPU_NO_CONVERGENCE;
  }
  HIPC(hipSetDevice(h->device));
  {
    const int rca = assemble_submap(h, ref_slots, ref_T, n_ref, total);
    if (rca) return rca;
  }
  const int rc = lsgpu_icp_compute(h, reinterpret_cast<const float*>(h->clouds[reading_slot].p), h->cloud_n[reading_slot],
                                   reinterpret_cast<const float*>(h->submap.p), total, T_init, chain, T_out, stats);
  return rc;  // (the assembly is stream-ordered: its time is part of compute's filter time, stats->t_reserved[0])
}

// lsgpu_cloud_upload(reading_slot) + lsgpu_icp_compute_clouds in one call: the new scan crosses PCIe WHILE the sub-map --
// the scans already in HBM -- is assembled and filtered (its own stream and host thread, as the host reading of
// lsgpu_icp_compute).  LaserTrack::localScanToSubMap matches every new scan exactly once, right after it arrived
// (laser_track.cpp:112-119, 466-519): uploaded first and matched afterwards, the 0.3 ms of a 1 M-point scan's copy were
// spent with the device idle, inside the reference's own timed region (scan_matching_times_).  The slot holds the scan on
// return whatever the registration's outcome, like the two calls one after the other.
int lsgpu_icp_compute_clouds_upload(lsgpu_icp* h, int reading_slot, const float* reading_xyz1, int64_t nq,
                                    const int* ref_slots, const float* ref_T, int n_ref, const float T_init[16],
                                    const lsgpu_chain_config* chain, float T_out[16], lsgpu_icp_stats* stats) {
  if (!h || !T_init || !T_out || !chain || n_ref < 0 || (n_ref > 0 && !ref_slots)) return LSGPU_BAD_ARG;
  if (reading_slot < 0 || reading_slot >= (1 << 20) || nq < 0 || nq > 0x7FFFFFF0ll || (nq > 0 && !reading_xyz1)) return LSGPU_BAD_ARG;
  auto have = [&](int s) { return s >= 0 && (size_t)s < h->clouds.size() && h->cloud_n[s] >= 0; };
  bool fused = nq > 0 && !is_device_ptr(reading_xyz1);
  int64_t total = 0;
  for (int i = 0; i < n_ref && fused; ++i) {
    fused = have(ref_slots[i]) && ref_slots[i] != reading_slot;
    if (fused) total += h->cloud_n[ref_slots[i]];
  }
  fused = fused && total > 0 && total <= 0x7FFFFFF0ll && chain->ssn_knn >= 3 && chain->ssn_knn <= kSsnMaxKnn;
  if (!fused) {   // nothing to overlap (or nothing to match against): the two calls, one after the other
    const int rcu = lsgpu_cloud_upload(h, reading_slot, reading_xyz1, nq);
    if (rcu) return rcu;
    return lsgpu_icp_compute_clouds(h, reading_slot, ref_slots, ref_T, n_ref, T_init, chain, T_out, stats);
  }
  h->err.clear();
  std::memcpy(T_out, T_init, 16 * sizeof(float));
  if (stats) std::memset(stats, 0, sizeof(*stats));
  HIPC(hipSetDevice(h->device));
  if ((size_t)reading_slot >= h->clouds.size()) { h->clouds.resize((size_t)reading_slot + 1); h->cloud_n.resize((size_t)reading_slot + 1, -1); }
  h->cloud_n[reading_slot] = -1;
  HIPC(h->clouds[reading_slot].reserve(nq));
  {
    const int rca = assemble_submap(h, ref_slots, ref_T, n_ref, total);
    if (rca) {   // (a sub-map transform that is not rigid: the scan is stored all the same)
      const std::string why = h->err;
      const int rcu = lsgpu_cloud_upload(h, reading_slot, reading_xyz1, nq);
      h->err = why;
      return rcu ? rcu : rca;
    }
  }
  h->upload_into = h->clouds[reading_slot].p;
  h->upload_done = false;
  const int rc = lsgpu_icp_compute(h, reading_xyz1, nq, reinterpret_cast<const float*>(h->submap.p), total, T_init, chain, T_out, stats);
  h->upload_into = nullptr;
  if (!h->upload_done) {   // the call returned before its upload went out: the plain copy, so that the slot holds the scan
    const std::string why = h->err;
    HIPC(hipMemcpyAsync(h->clouds[reading_slot].p, reading_xyz1, (size_t)nq * 16, hipMemcpyDefault, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    h->err = why;
  }
  h->upload_done = false;
  h->cloud_n[reading_slot] = nq;
  return rc;
}

int lsgpu_icp_align(lsgpu_icp* h, const float* reading_xyz1, int64_t nq, const float T_init[16],
                    float T_out[16], lsgpu_icp_stats* stats) {
  if (!h || !T_init || !T_out) return LSGPU_BAD_ARG;
  h->err.clear();
  std::memcpy(T_out, T_init, 16 * sizeof(float));
  lsgpu_icp_stats st;
  std::memset(&st, 0, sizeof(st));
  if (stats) *stats = st;
  h->trace.clear(); h->trace_on_device = 0;
  // queries that lsgpu_icp_compute ordered and moved on its side stream belong to THIS call and to no later one, whatever
  // way it ends (a guess that is refused below would otherwise leave queries moved by that guess to the next call with the
  // same pointer and size)
  const float* const prepared_rd = h->prepared_rd;
  const int64_t prepared_nq = h->prepared_nq;
  h->prepared_rd = nullptr; h->prepared_nq = 0;
  // Local reasons not to start.  In the split-scan mode they are NOT returned yet: a rank that left here would
  // leave its peers blocked in the first collective, so every rank first takes part in the entry handshake below.
  int local_rc = LSGPU_OK;
  if (h->nr <= 0 || nq <= 0 || !reading_xyz1) {  // empty cloud: ConvergenceError upstream
    h->err = "align: empty reading or no reference";
    local_rc = LSGPU_NO_CONVERGENCE;
  } else if (nq > 0x7FFFFFF0ll) {
    local_rc = LSGPU_BAD_ARG;
  } else if (!lsgpu_check_rigid(T_init)) {
    // step 5 moves the reading with RigidTransformation::compute, which throws TransformationError for such a matrix
    // (after both filters have run, as here when the call came through lsgpu_icp_compute)
    h->err = "align: the initial guess is not a rigid transformation (|1 - det R| > 1e-3)";
    local_rc = LSGPU_BAD_ARG;
  }
  if (local_rc && !h->comm) { h->cone_build_in_align = false; return local_rc; }
  HIPC(hipSetDevice(h->device));
  const double t0 = wall_ms();
  h->knn_events_used = 0;
  h->comm_events_used = 0;
  h->time_comm = h->comm != nullptr && h->cfg.profile_kernels != 0;
  h->tail_pending = false;   // (from here on everything is behind it on h->stream itself)

  // step 5: T_refMean_dataIn = T_refIn_refMean^-1 * T_init (pure translation inverse)
  float T_rm_in[16];
  std::memcpy(T_rm_in, T_init, sizeof(T_rm_in));
  for (int d = 0; d < 3; ++d) T_rm_in[12 + d] = T_init[12 + d] - h->mean[d];
  // (lsgpu_icp_compute may have ordered and moved these very queries on its side stream already: the loop's stream
  // only has to wait for that)
  const bool prepared = !local_rc && !h->comm && prepared_rd == reading_xyz1 && prepared_nq == nq;
  int rc = local_rc;
  if (!rc) {
    if (prepared) HIPC(hipStreamWaitEvent(h->stream, h->side_done, 0));
    else rc = prepare_queries(h, reading_xyz1, nq, to_mat34(T_rm_in));
  }
  if (rc && !h->comm) return rc;
  int64_t nq_total = nq;
  if (h->comm) {
    // entry handshake: {shard size, cannot-start flag} summed over the ranks.  TrimmedDist ranks over ALL matches, so
    // the rank uses the global count; and either every rank enters the loop or none does.
    long long* hn = reinterpret_cast<long long*>(h->h_pinned + 60);
    hn[0] = rc ? 0 : (long long)nq;
    hn[1] = rc ? 1 : 0;
    HIPC(hipMemcpyAsync(h->comm_tmp.p, hn, 2 * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    RCCLC(rccl_api()->AllReduce(h->comm_tmp.p, h->comm_tmp.p + 2, 2, ncclInt64, ncclSum, h->comm, h->stream));
    HIPC(hipMemcpyAsync(hn, h->comm_tmp.p + 2, 2 * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    const int w = wait_stream(h);
    if (w) return w;
    nq_total = (int64_t)hn[0];
    if (rc) return rc;
    if (hn[1] != 0) {
      h->err = "align (split-scan): another rank could not start; all ranks give up together";
      return LSGPU_NO_CONVERGENCE;
    }
  }

  // step 6: the loop runs on the device.  Every iteration is {kNN, select x3, normal equations,
  // update}; k_icp_update solves, moves T_iter, runs the checkers and raises `done`, after which the
  // remaining enqueued launches exit immediately.  The host only looks at the state every few iterations.
  const int max_it = h->cfg.max_iterations;
  HIPC(h->state.reserve(1));
  HIPC(h->chk_hist.reserve((size_t)8 * (max_it + 2)));
  HIPC(h->trace_dev.reserve((size_t)max_it));
  IcpState* hst = reinterpret_cast<IcpState*>(h->h_pinned + 64);  // pinned staging (<= 512 B); [0..47] D2H, [48..63] H2D
  static_assert(sizeof(IcpState) <= 64 * sizeof(double), "IcpState staging");
  std::memset(hst, 0, sizeof(IcpState));
  hostmath::identity4(hst->T_iter);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) hst->T_rows[r * 4 + c] = hst->T_iter[c * 4 + r];
  hst->prev_limit = INFINITY; hst->cap2 = INFINITY;
  hst->cap_enabled = h->cfg.reserved[0] == 0 ? 1 : 0;
  hst->max_iter = max_it; hst->smooth = h->cfg.smooth_length;
  hst->lim_rot = h->cfg.min_diff_rot; hst->lim_trans = h->cfg.min_diff_trans;
  AlignInitArgs ia{};
  {  // checkers.init(T_iter): history starts with the identity
    hostmath::CheckerState cs{0, 0};
    hostmath::checker_push(&cs, ia.chk0, hst->T_iter);
    hst->counter = cs.counter; hst->n_hist = cs.n_hist;
  }
  const uint32_t k = trim_rank(nq_total, h->cfg.trim_ratio);
  HIPC(h->sel_aux.reserve(kSelFailFlag + 4));
  HIPC(h->spread_flag.reserve((size_t)((nq + 63) / 64))); HIPC(h->spread_list.reserve(kFrontMax)); HIPC(h->spread_cnt.reserve(2));
  HIPC(h->sel_win.reserve((size_t)kSelWinRows * 512));
  HIPC(h->amb_key.reserve((size_t)kSelAmbCap)); HIPC(h->amb_val.reserve((size_t)kSelAmbCap * 32));
  hst->sel_wide = (!h->comm && tuning().fused_select) ? 1 : 0;
  h->n_spread_host = 0; h->n_spread_known = false;
  ia.state = *hst;
  ia.sel0 = SelState{0u, k};   // sel[0] = {0, rank}: constant during an align
  ia.state_dev = h->state.p; ia.chk_hist = h->chk_hist.p; ia.sel = h->sel.p;
  ia.counters3 = h->counters.p + 32;   // stragglers, (unused), work-list length
  ia.ne_ticket = h->ne_tickets.p; ia.sel_aux = h->sel_aux.p; ia.spread_flag = h->spread_flag.p;
  ia.spread_cnt = h->spread_cnt.p; ia.sel_win = h->sel_win.p; ia.hist = h->hist.p;
  ia.n_sel_aux = kSelFailFlag + 4; ia.n_spread_flag = (int)((nq + 63) / 64); ia.n_sel_win = kSelWinRows * 512;
  hipLaunchKernelGGL(k_align_init, dim3(64), dim3(256), 0, h->stream, ia);
  HIPC(hipGetLastError());

  const int nb = std::min(kNeBlocks, nblk(nq));
  const bool timed = h->cfg.profile_kernels != 0;
  const Mat34 Tdummy = to_mat34(hst->T_iter);
  std::vector<size_t> ev_of_launch;  // event index of every enqueued iteration
  const bool split_update = tuning().split_update;  // (profiling: the update as its own launch)
  // The launch policy (lsgpu_policy.h) decides what every iteration is made of and when the host looks at the loop
  // state; this function executes its decisions.  (tests/cpp/policy_check.cpp drives the same state machine on the CPU.)
  policy::Config& pc = h->pol_cfg;
  pc = policy::Config();
  pc.cone_from = tuning().cone_from; pc.wide_iters = tuning().wide_iters; pc.group = 6;
  pc.enq_limit = 8 * max_it + 64;   // (only guards against a device that never finishes, see below)
  pc.predict_select = tuning().predict_select; pc.commit_select = tuning().commit_select; pc.comm_commit = tuning().comm_commit;
  pc.lookahead = tuning().lookahead; pc.comm = h->comm != nullptr;
  pc.seed_cap = tuning().seed_cap; pc.cap_enabled = h->cfg.reserved[0] == 0;
  pc.two_pass_select = !h->comm && tuning().fused_select && tuning().two_pass_select;
  pc.cone_probe = tuning().cone_probe; pc.cone_heavy_share = tuning().cone_heavy_share; pc.cone_max_occupancy = tuning().cone_max_occupancy;
  policy::State& pol = h->pol;
  if (h->cone_build_in_align) { h->cone_ok = cone_wanted(h); h->cone_decided = false; }   // (its build follows the first iteration, below)
  // a handle whose last alignments found the index slower than the voxel grid leaves it alone for a while (and spares
  // itself the build when that is still to come)
  // (a reference of a markedly different size is another scene or another sub-map depth: the judgement starts over)
  if (h->index_rest > 0 && (h->nr > 2 * h->index_rest_nr || 2 * h->nr < h->index_rest_nr)) h->index_rest = 0;
  const bool index_rests = h->index_rest > 0 && h->cone_ok;
  if (index_rests) { --h->index_rest; if (h->cone_build_in_align) { h->cone_build_in_align = false; h->cone_ok = false; } }
  pol.begin_align(h->cone_ok && !index_rests, h->cone_decided, h->cone_dense, h->cone_occupancy);
  h->pay_voxel_timed = h->pay_index_timed = false;
  if (!h->ev_pay[0]) for (auto& e : h->ev_pay) HIPC(hipEventCreate(&e));
  auto enqueue_iteration = [&](const policy::Iteration& itn) -> int {
    // itn.knn == false: only select + normal equations + update on the distances already there (after a missed
    // prediction).  RCCL mode: the per-shard tables of a *committed* iteration are summed over the ranks in ONE grouped
    // all-reduce (the host knows beforehand that no select kernel will run: sel_streak comes from the global limit,
    // so every rank takes the same decision); un-committed iterations there run the plain three-pass select.
    int r = LSGPU_OK;
    if (itn.knn) {
      r = run_knn(h, Tdummy, h->state.p, itn, timed, itn.seed && itn.capped ? k : 0xFFFFFFFFu);  // 6a+6b
      if (r) return r;
      ev_of_launch.push_back(h->knn_events_used ? h->knn_events_used - 1 : 0);
    }
    if (itn.committed && h->comm) {  // one exchange for the whole select: {counts below, 11-bit histogram, window table}
      RcclApi* api = rccl_api();
      comm_mark(h, true);
      if (api->GroupStart) RCCLC(api->GroupStart());
      RCCLC(api->AllReduce(h->sel_aux.p, h->sel_aux.p, kSelFailFlag, ncclUint32, ncclSum, h->comm, h->stream));
      RCCLC(api->AllReduce(h->hist.p + kHistBins, h->hist.p + kHistBins, kHistBins, ncclUint32, ncclSum, h->comm, h->stream));
      RCCLC(api->AllReduce(h->sel_win.p, h->sel_win.p, (size_t)kSelWinRows * 512, ncclUint32, ncclSum, h->comm, h->stream));
      if (api->GroupEnd) RCCLC(api->GroupEnd());
      comm_mark(h, false);
    }
    if (!itn.committed) {
      r = run_select(h, h->d2.p, (int)nq, k, false /* armed by k_align_init / k_seed_cap / the previous k_normal_eq_loop */,
                     h->state.p, true, itn.predicted, itn.full_select ? 3 : 2);       // 6c
      if (r) return r;
    }
    lsgpu_icp::KnnEv* ev = (timed && itn.knn && h->knn_events_used) ? &h->knn_events[h->knn_events_used - 1] : nullptr;
    if (ev) HIPC(hipEventRecord(ev->d, h->stream));
    hipLaunchKernelGGL(k_normal_eq_loop, dim3(nb), dim3(256), 0, h->stream, h->rdq.p, (int)nq,
                       h->state.p, h->prev.p, h->d2.p, h->nrm.p, h->hist.p, h->sel.p + 2,
                       h->counters.p + 32, h->ne_tickets.p, h->ne_partials.p, h->ne_gpartials.p, h->ne_out.p,
                       h->chk_hist.p, h->trace_dev.p, max_it, itn.capped ? 1 : 0, (h->comm || split_update) ? 0 : 1,
                       h->sel_aux.p, (h->comm || !tuning().fused_select) ? h->sel_win.p : nullptr, itn.committed ? 1 : 0, h->spread_cnt.p,
                       (itn.predicted && itn.knn) ? 1 : 0, h->sel_aux.p + kSelFailFlag + 2, h->amb_key.p, h->amb_val.p, tuning().sel_amb_cap,
                       (!itn.full_select && !itn.committed) ? 1 : 0);   // 6d (+6e)
    if (h->comm || split_update) {   // split scan: every rank gets the sums over all shards (the limit, slot 29, is already global)
      if (h->comm) comm_mark(h, true);
      if (h->comm && rccl_api()->AllReduce(h->ne_out.p, h->ne_out.p, kNe, ncclDouble, ncclSum, h->comm, h->stream) != ncclSuccess) {
        h->err = "RCCL all-reduce of the normal equations failed";
        return LSGPU_HIP_ERROR;
      }
      if (h->comm) comm_mark(h, false);
      hipLaunchKernelGGL(k_icp_update, dim3(1), dim3(64), 0, h->stream, h->state.p, h->ne_out.p,
                         h->chk_hist.p, h->trace_dev.p, max_it, itn.capped ? 1 : 0, h->sel_aux.p);           // 6d+6e
    }
    if (ev) HIPC(hipEventRecord(ev->e, h->stream));
    return hipGetLastError() == hipSuccess ? LSGPU_OK : LSGPU_HIP_ERROR;
  };
  // A look at the loop state.  Look-ahead (not in the split-scan mode): ONE more iteration is enqueued behind the copy
  // before the host waits for it, so the device works through the round trip instead of idling (~25 us + a cold first
  // launch per look); that iteration carries the decisions of the previous look, like the later iterations of any group,
  // and exits at once if the state it finds says `done`.
  if (pc.lookahead && !pc.comm && !h->ev_state) HIPC(hipEventCreateWithFlags(&h->ev_state, hipEventDisableTiming));
  auto fetch_state = [&](int* enqueued_ahead) -> int {
    *enqueued_ahead = 0;
    HIPC(hipMemcpyAsync(hst, h->state.p, sizeof(IcpState), hipMemcpyDeviceToHost, h->stream));
    policy::Iteration ahead_it;
    if (pol.lookahead_iteration(pc, &ahead_it)) {
      HIPC(hipEventRecord(h->ev_state, h->stream));
      const int r = enqueue_iteration(ahead_it);
      if (r) return r;
      *enqueued_ahead = 1;
      HIPC(hipEventSynchronize(h->ev_state));
      return LSGPU_OK;
    }
    return wait_stream(h);  // (bounded in the split-scan mode: a dead peer must not hang this rank)
  };

  // iteration 0: seeded; capped by the trim quantile of the seed distances (a guaranteed bound: no retry can follow)
  rc = enqueue_iteration(pol.plan(pc, true, pc.seed_cap && pc.cap_enabled, true, true, false));
  if (rc) return rc;
  pol.enq = 1; pol.since_check = 1;
  if (h->cone_build_in_align) {
    // lsgpu_icp_compute left the direction index's build to this point: the device is busy with the first search, the
    // ~15 launches of the build go to the side stream (its own sort scratch) while it is
    h->cone_build_in_align = false;
    h->cur = h->side_stream; h->sc = &h->scr_side;
    int rb = build_cone_index(h);
    if (!rb && h->cone_ok) {
      if (hipEventRecord(h->cone_done, h->side_stream) != hipSuccess) { (void)hipGetLastError(); rb = LSGPU_HIP_ERROR; h->err = "align: event record"; }
      h->cone_pending = true;
    }
    h->cur = h->stream; h->sc = &h->scr_main;
    if (rb) return rb;
  }
  // The device decides when the loop ends (CounterTransformationChecker raises `done` after max_iterations at the
  // latest); the host keeps feeding groups of launches until it sees `done`.  Launches enqueued behind an
  // iteration that had to be repeated exit at once, so the number of enqueues is NOT bounded by max_iterations;
  // the policy's enq_limit only guards against a device that never finishes.
  for (;;) {
    policy::Iteration itn;
    if (pol.next_in_group(pc, &itn)) {
      rc = enqueue_iteration(itn);
      if (rc) return rc;
      continue;
    }
    int ahead = 0;
    const bool repriced = pol.wants_reprice();   // (the last launch in front of this look priced the index again: its counters
    rc = fetch_state(&ahead);                    //  were copied in front of the state, they are on the host when the state is)
    if (rc) return rc;
    h->n_spread_host = hst->n_spread; h->n_spread_known = true;
    policy::LookInput li;
    li.done = hst->done; li.status = hst->status; li.iter = hst->iter; li.sel_streak = hst->sel_streak;
    if (hst->sel_fails >= 2) pc.two_pass_select = false;   // (slices fuller than the normal equations can set aside: the select's third pass is back)
    li.stragglers = hst->stragglers; li.nq = nq; li.status_cap_failed = kStatusCapFailed; li.status_sel_failed = kStatusSelFailed;
    if (tuning().short_last_group) { li.chk_rot = hst->chk_rot; li.chk_trans = hst->chk_trans; li.lim_rot = hst->lim_rot; li.lim_trans = hst->lim_trans; li.chk_rot_prev = hst->chk_rot_prev; li.chk_trans_prev = hst->chk_trans_prev; }
    const policy::LookVerdict verdict = pol.on_look(pc, li, ahead, repriced ? price_share(h) : -1.f);
    if (verdict == policy::LookVerdict::RepeatUncapped) {
      // the cap prediction failed for iteration hst->iter: repeat it uncapped, then carry on
      hst->done = 0; hst->status = 0;
      HIPC(hipMemcpyAsync(h->state.p, hst, sizeof(IcpState), hipMemcpyHostToDevice, h->stream));
      HIPC(hipStreamSynchronize(h->stream));
      rc = enqueue_iteration(pol.repeat_uncapped(pc));
      if (rc) return rc;
      continue;
    }
    if (verdict == policy::LookVerdict::RepeatSelect) {
      // the limit left the predicted 12-bit bin in iteration hst->iter: its distances stand, the full select
      // and everything after it run again
      hst->sel_streak = 0;
      hst->done = 0; hst->status = 0;
      HIPC(hipMemcpyAsync(h->state.p, hst, sizeof(IcpState), hipMemcpyHostToDevice, h->stream));
      HIPC(hipStreamSynchronize(h->stream));
      rc = enqueue_iteration(pol.repeat_select(pc));
      if (rc) return rc;
      continue;
    }
    if (verdict == policy::LookVerdict::Done) {
      if (ahead) {   // one launch is still queued behind the look that saw `done`
        if (!h->ev_tail) HIPC(hipEventCreateWithFlags(&h->ev_tail, hipEventDisableTiming));
        HIPC(hipEventRecord(h->ev_tail, h->stream));
        h->tail_pending = true;
      }
      break;
    }
    if (verdict == policy::LookVerdict::GiveUp) {
      h->err = "align: the device loop did not finish";
      return LSGPU_HIP_ERROR;
    }
  }
  st.cap_retries = pol.cap_retries;
  const int sel_retries = pol.sel_retries, committed_iterations = pol.committed_iterations;
  if (h->pay_voxel_timed && h->pay_index_timed) {   // (both launches are long over: the loop's end was seen behind them)
    float t_voxel = 0.f, t_index = 0.f;
    if (hipEventElapsedTime(&t_voxel, h->ev_pay[0], h->ev_pay[1]) == hipSuccess && hipEventElapsedTime(&t_index, h->ev_pay[2], h->ev_pay[3]) == hipSuccess) {
      h->pay_voxel_us = t_voxel * 1e3f; h->pay_index_us = t_index * 1e3f;
      if (tuning().index_rest && policy::index_not_paying(t_voxel * 1e3f, t_index * 1e3f)) { h->index_rest = policy::kIndexRestAligns; h->index_rest_nr = h->nr; }
    } else {
      (void)hipGetLastError();
    }
  }
  const int it = hst->iter;
  rc = hst->status;
  if (rc == LSGPU_NO_CONVERGENCE)
    h->err = hst->err_code == 1 ? "no point to minimize" : hst->err_code == 2 ? "normal matrix not positive definite"
                                                                          : "NaN in transformation checker";
  st.iterations = it;
  st.converged = hst->converged;
  st.stragglers = (int64_t)hst->stragglers;
  float T_iter[16];
  std::memcpy(T_iter, hst->T_iter, sizeof(T_iter));
  // The per-iteration trace stays in device memory until lsgpu_icp_get_trace asks for it (the synchronous copy cost every
  // alignment 37 us -- 5 % of a 200 k-point pair -- for a record nobody but the tests and the profiler reads); the two
  // statistics that came out of it travel with the loop state.
  h->trace.clear();
  h->trace_on_device = (size_t)std::min(it, max_it);
  h->trace_knn_us.clear();
  if (it > 0) { st.final_limit = hst->last_limit; st.final_n_used = (int64_t)hst->last_used; }
  if (rc == LSGPU_OK) {  // step 7
    float Tmean[16], tmp[16];
    hostmath::identity4(Tmean);
    for (int d = 0; d < 3; ++d) Tmean[12 + d] = h->mean[d];
    hostmath::mul4(T_iter, T_rm_in, tmp);
    hostmath::mul4(Tmean, tmp, T_out);
  }
  // kNN timing: launches that actually ran are the first `it` (+ retries) enqueued ones
  {
    size_t t = 0;
    for (size_t i = 0; i < h->knn_events_used && i < ev_of_launch.size(); ++i) {
      const auto& e = h->knn_events[ev_of_launch[i]];
      float m1 = 0.f, m2 = 0.f;
      if (hipEventElapsedTime(&m1, e.a, e.b) != hipSuccess || (e.second && hipEventElapsedTime(&m2, e.b, e.c) != hipSuccess)) {
        (void)hipGetLastError();
        continue;
      }
      if (st.knn_launches >= it + st.cap_retries) break;  // the rest exited immediately (enqueued past the end)
      if (m1 < 0.02f) continue;  // exited immediately (enqueued behind an iteration that had to be repeated)
      st.t_knn_main_ms += m1; st.t_knn_fallback_ms += m2; st.t_knn_ms += m1 + m2; st.knn_launches++;
      float m3 = 0.f, m4 = 0.f;
      if (hipEventElapsedTime(&m3, e.second ? e.c : e.b, e.d) == hipSuccess && hipEventElapsedTime(&m4, e.d, e.e) == hipSuccess) { st.t_select_ms += m3; st.t_ne_ms += m4; }
      else (void)hipGetLastError();
      if (t < h->trace_on_device) { h->trace_knn_us.push_back({m1 * 1e3f, m2 * 1e3f}); ++t; }
    }
  }
  if (h->time_comm) {   // time inside the collectives (launches enqueued behind the end exit at once and add next to nothing)
    for (size_t i = 0; i < h->comm_events_used; ++i) {
      float m = 0.f;
      if (hipEventElapsedTime(&m, h->comm_events[i].first, h->comm_events[i].second) == hipSuccess) { st.t_comm_ms += m; st.comm_calls++; }
      else (void)hipGetLastError();
    }
  }
  st.direction_index_launches = h->pol.cone_launches;   // (enqueued; those behind the end of the loop exited at once)
  st.direction_index_occupancy = h->cone_decided ? h->cone_occupancy : 0.f;
  st.direction_index_heavy_share = h->pol.cone_heavy;
  st.pad_ = sel_retries;  // (select predictions that missed; informational)
  st.committed_select_iterations = committed_iterations;
  st.spread_tiles = (int)hst->n_spread;
  st.t_total_ms = wall_ms() - t0;
  if (stats) *stats = st;
  return rc;
}

// dev only (LSGPU_KNN_STATS build): per-wave records of the last k_knn_tile launch / global counters
int lsgpu_icp_align_batch(lsgpu_icp* const* handles, int n_handles, int64_t n_pairs,
                          const float* const* reference_xyz1, const float* const* reference_normals,
                          const int64_t* n_reference, const float* const* reading_xyz1,
                          const int64_t* n_reading, const float* T_init, float* T_out,
                          lsgpu_icp_stats* stats, int* rc) {
  if (!handles || n_handles < 1 || n_pairs < 0) return LSGPU_BAD_ARG;
  if (n_pairs == 0) return LSGPU_OK;
  if (!reference_xyz1 || !reference_normals || !n_reference || !reading_xyz1 || !n_reading || !T_init || !T_out)
    return LSGPU_BAD_ARG;
  for (int k = 0; k < n_handles; ++k)
    if (!handles[k] || handles[k]->device != handles[0]->device || handles[k]->comm) return LSGPU_BAD_ARG;
  std::vector<int> codes((size_t)n_pairs, LSGPU_OK);
  // handle k owns pairs k, k + n_handles, ...: a static assignment, each pair is one independent,
  // deterministic {set_reference, align} on that handle's stream
  auto worker = [&](int k) {
    lsgpu_icp* h = handles[k];
    // Reference reuse (the part that is exact): a pair that names the SAME reference buffers as the
    // previous pair of this handle (same pointers, same size -- many readings against one map, loop-closure candidates
    // against one sub-map) keeps the centred, sorted reference, its chunks and its cell tables: steps 2-3 of ICP::compute
    // depend on the reference alone.  Bit-identical with rebuilding (tests/test_gpu_parity.py::
    // test_align_batch_reuses_a_shared_reference).
    const float* have_ref = nullptr; const float* have_nrm = nullptr; int64_t have_n = -1;
    for (int64_t i = k; i < n_pairs; i += n_handles) {
      float* To = T_out + 16 * i;
      const float* Ti = T_init + 16 * i;
      std::memcpy(To, Ti, 16 * sizeof(float));
      if (stats) std::memset(&stats[i], 0, sizeof(lsgpu_icp_stats));
      int c = LSGPU_OK;
      const bool same = have_n > 0 && reference_xyz1[i] == have_ref && reference_normals[i] == have_nrm && n_reference[i] == have_n;
      if (!same) {
        c = lsgpu_icp_set_reference(h, reference_xyz1[i], reference_normals[i], n_reference[i]);
        have_ref = reference_xyz1[i]; have_nrm = reference_normals[i]; have_n = c == LSGPU_OK ? n_reference[i] : -1;
      }
      if (c == LSGPU_OK) c = lsgpu_icp_align(h, reading_xyz1[i], n_reading[i], Ti, To, stats ? &stats[i] : nullptr);
      if (stats) stats[i].reference_reused = same ? 1 : 0;
      codes[(size_t)i] = c;
    }
  };
  const int nt = (int)std::min<int64_t>(n_handles, n_pairs);
  std::vector<std::thread> pool;
  for (int k = 1; k < nt; ++k) pool.emplace_back(worker, k);
  worker(0);
  for (auto& t : pool) t.join();
  int ret = LSGPU_OK;
  for (int64_t i = 0; i < n_pairs; ++i) {
    const int c = codes[(size_t)i];
    if (rc) rc[i] = c;
    if (c != LSGPU_OK && c != LSGPU_NO_CONVERGENCE) { if (ret == LSGPU_OK || ret == LSGPU_NO_CONVERGENCE) ret = c; }
    else if (c == LSGPU_NO_CONVERGENCE && ret == LSGPU_OK) ret = c;
  }
  return ret;
}

int lsgpu_dev_knn_wave_stats(lsgpu_icp* h, unsigned int* out, int nwaves) {
  if (!h) return LSGPU_BAD_ARG;
  if (!out) { HIPC(h->knn_dbg_wave.reserve((size_t)nwaves)); HIPC(hipMemset(h->knn_dbg_wave.p, 0, (size_t)nwaves * 16)); return LSGPU_OK; }
  HIPC(hipStreamSynchronize(h->stream));
  HIPC(hipMemcpy(out, h->knn_dbg_wave.p, (size_t)nwaves * 16, hipMemcpyDeviceToHost));
  return LSGPU_OK;
}
#ifdef LSGPU_KNN_STATS
int lsgpu_dev_tree_phases(lsgpu_icp* h, unsigned long long out[64]) {  // stats build only (devtools/tree_phases.py)
  if (!h) return LSGPU_BAD_ARG;
  HIPC(hipStreamSynchronize(h->stream));
  HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tree_dbg), 512));
  return LSGPU_OK;
}
int lsgpu_dev_cone_phases(lsgpu_icp* h, unsigned int* out, int ntiles) {  // stats build only (devtools/cone_phases.py)
  // out == nullptr: (re)arm the record buffer for `ntiles` waves per iteration; otherwise copy the records back
  if (!h || ntiles <= 0) return LSGPU_BAD_ARG;
  static uint32_t* rec = nullptr;
  static size_t rec_words = 0;
  const size_t words = (size_t)kConeRecIters * (size_t)ntiles * 16;
  HIPC(hipStreamSynchronize(h->stream));
  if (!out) {
    if (words > rec_words) { if (rec) HIPC(hipFree(rec)); HIPC(hipMalloc((void**)&rec, words * 4)); rec_words = words; }
    HIPC(hipMemset(rec, 0, words * 4));
    HIPC(hipMemcpyToSymbol(HIP_SYMBOL(g_cone_rec), &rec, sizeof(rec)));
    return LSGPU_OK;
  }
  if (!rec || words > rec_words) return LSGPU_BAD_ARG;
  HIPC(hipMemcpy(out, rec, words * 4, hipMemcpyDeviceToHost));
  return LSGPU_OK;
}
int lsgpu_dev_ne_phases(lsgpu_icp* h, unsigned long long out[24]) {  // stats build only (devtools/ne_phases.py)
  if (!h) return LSGPU_BAD_ARG;
  HIPC(hipStreamSynchronize(h->stream));
  HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ne_dbg), 192));
  return LSGPU_OK;
}
#endif
int lsgpu_dev_knn_counters(lsgpu_icp* h, unsigned long long out[8]) {
  if (!h) return LSGPU_BAD_ARG;
  if (!h->knn_dbg.p) {
    HIPC(h->knn_dbg.reserve(8));
    HIPC(hipMemset(h->knn_dbg.p, 0, 64));
    std::memset(out, 0, 64);
    return LSGPU_OK;
  }
  HIPC(hipStreamSynchronize(h->stream));
  HIPC(hipMemcpy(out, h->knn_dbg.p, 64, hipMemcpyDeviceToHost));
  HIPC(hipMemset(h->knn_dbg.p, 0, 64));
  return LSGPU_OK;
}

int lsgpu_icp_get_trace(lsgpu_icp* h, lsgpu_iter_trace* out, int cap) {
  if (!h || !out || cap <= 0) return 0;
  if (h->trace_on_device) {   // the last alignment's records are still where the device wrote them
    h->trace.resize(h->trace_on_device);
    if (hipSetDevice(h->device) != hipSuccess ||
        hipMemcpy(h->trace.data(), h->trace_dev.p, h->trace.size() * sizeof(lsgpu_iter_trace), hipMemcpyDeviceToHost) != hipSuccess) {
      (void)hipGetLastError();
      h->trace.clear();
    }
    for (size_t t = 0; t < h->trace.size() && t < h->trace_knn_us.size(); ++t) {
      h->trace[t].knn_main_us = h->trace_knn_us[t].first; h->trace[t].knn_fallback_us = h->trace_knn_us[t].second;
    }
    h->trace_on_device = 0;
  }
  const int n = std::min<int>(cap, (int)h->trace.size());
  std::memcpy(out, h->trace.data(), (size_t)n * sizeof(lsgpu_iter_trace));
  return n;
}

}  // extern "C"
