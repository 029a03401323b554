/*
 * lsgpu_icp.h -- C ABI of the MI355X-native ICP scan-matching path (liblsgpu_icp.so).
 *
 * The reference exposes no FFI for this path: `icp_` is a concrete PointMatcher::ICP member
 * (laser_slam/include/laser_slam/laser_track.hpp:217, incremental_estimator.hpp:70) and the two
 * call sites are
 *     icp_.compute(last_scan.scan, sub_map, T_init)        laser_slam/src/laser_track.cpp:496
 *     icp_.compute(sub_map_b, sub_map_a, T_init)           laser_slam/src/incremental_estimator.cpp:108
 * This header is the seam a maintainer binds instead (INTEGRATION.md shows the C++ shim): one
 * handle == one `icp_` member == one device + one HIP stream.  Handles are independent (tracks run
 * concurrently, laser_track.hpp:211 holds one mutex per track); a handle is not thread safe.
 *
 * Data layout == PointMatcher<float>::DataPoints (laser_slam/include/laser_slam/common.hpp:14-17):
 *   features    (dim+1) x N column major  ->  AoS x,y,z,1 float, 16 B / point  ("xyz1")
 *   descriptors "normals" 3 x N column major -> 12 B / point
 *   TransformationParameters 4x4 float column major; p_reference = T * p_reading.
 * Every pointer argument may be host memory or device (HBM) memory of the handle's device; the
 * library detects which.  Host buffers are copied, never retained.  No exceptions cross this ABI.
 *
 * Stream contract: every call does its device work on the HANDLE'S OWN stream (created non-blocking: it does not
 * synchronise with the null stream or with any stream of the caller) and returns after that work has completed, so
 * outputs are valid on return.  Device INPUTS must be complete before the call: a buffer still being written by
 * the caller's stream has to be synchronised first (hipStreamSynchronize / an event wait on the producing stream).
 * The Python twin (laser_slam_amd/icp.py) synchronises torch's current stream before every call that passes a
 * device tensor.
 */
#ifndef LSGPU_ICP_H_
#define LSGPU_ICP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSGPU_ABI_VERSION 4

/* Return codes.  NO_CONVERGENCE is PointMatcher::ConvergenceError: laser_track.cpp:499-502 catches it
 * and keeps the odometry guess; incremental_estimator.cpp:108 lets it propagate. */
enum {
  LSGPU_OK = 0,
  LSGPU_NO_CONVERGENCE = 1,
  LSGPU_BAD_CONFIG = 2,
  LSGPU_HIP_ERROR = 3,
  LSGPU_BAD_ARG = 4
};

typedef struct lsgpu_icp lsgpu_icp;

/* Device-side part of the chain in laser_slam/configurations/icp_default.yaml:9-27. */
typedef struct lsgpu_icp_config {
  float trim_ratio;       /* TrimmedDistOutlierFilter ratio            yaml:16  (0.75)  */
  int   max_iterations;   /* CounterTransformationChecker              yaml:23  (40)    */
  float min_diff_rot;     /* Differential... minDiffRotErr   [rad]     yaml:25  (0.001) */
  float min_diff_trans;   /* Differential... minDiffTransErr [m]       yaml:26  (0.01)  */
  int   smooth_length;    /* Differential... smoothLength              yaml:27  (4)     */
  float cell_size;        /* finest voxel edge [m]; <= 0: automatic                      */
  int   profile_kernels;  /* 1: HIP-event time every kNN launch (see lsgpu_icp_stats)    */
  int   reserved[8];      /* reserved[0] = 1 disables the trimmed-radius cap (debug)       */
} lsgpu_icp_config;

/* icp_default.yaml values / ICP::setDefault() values (laser_track.cpp:17,20). */
void lsgpu_icp_config_yaml(lsgpu_icp_config* c);
void lsgpu_icp_config_default(lsgpu_icp_config* c);

typedef struct lsgpu_icp_stats {
  int     iterations;
  int     converged;         /* 1 stopped by the differential checker, 0 by the counter */
  float   final_limit;       /* last trimmed squared-distance limit                     */
  int64_t final_n_used;      /* pairs with weight 1 in the last iteration               */
  int64_t stragglers;        /* queries resolved by the exact fallback search, summed   */
  double  t_total_ms;        /* host wall time of the call                              */
  double  t_knn_ms;          /* sum of kNN time, main + fallback (HIP events; profile_kernels=1) */
  int     knn_launches;
  double  t_knn_main_ms;     /* k_knn_main only                                         */
  double  t_knn_fallback_ms; /* k_knn_fallback only                                     */
  int     cap_retries;       /* iterations repeated because the radius-cap prediction failed */
  int     pad_;              /* iterations whose predicted select missed and was redone in full (info)  */
  double  t_reserved[1];     /* lsgpu_icp_compute: milliseconds in the two filters + set_reference */
  double  t_select_ms;       /* sum over the iterations: trimmed-distance select kernels (profile_kernels=1) */
  double  t_ne_ms;           /* sum over the iterations: normal equations + solve + checkers (profile_kernels=1) */
  int     committed_select_iterations;  /* iterations whose trim limit came from the search kernels' own tables (no select launch) */
  int     spread_tiles;                 /* 64-query tiles whose queries share no candidates (searched row-wise by the front of the tile kernel) */
  int     reference_reused;             /* lsgpu_icp_align_batch: 1 if this pair kept the previous pair's reference structures (no set_reference) */
  int     comm_calls;                   /* split-scan mode, profile_kernels=1: RCCL calls of the loop ... */
  double  t_comm_ms;                    /* ... and the time between their first and last kernel on the stream, summed */
  int     direction_index_launches;     /* searches served by the direction index (k_knn_cone) instead of the voxel grid */
  float   direction_index_occupancy;    /* reference points per occupied bin of that index (0: not built / not looked at);
                                           above LSGPU_CONE_MAX_OCC (7) the settled searches stay on the voxel grid */
  float   direction_index_heavy_share;  /* share of the searching queries whose windows in that index would be long (priced by the
                                           search before its first use, again before every later look at the loop state while
                                           it keeps the alignment off the index; -1: not priced); above LSGPU_CONE_HEAVY_SHARE
                                           (csrc/lsgpu_tuning.h: 0.07) the alignment stays on the voxel grid */
} lsgpu_icp_stats;

/* One record per iteration (optional parity/debug trace; replaces the VTKFileInspector dump of
 * icp_default.yaml:32-40). */
typedef struct lsgpu_iter_trace {
  float   T_iter[16];
  float   limit;
  int64_t n_used;
  double  A[36];
  double  b[6];
  double  x[6];
  float   knn_main_us;      /* k_knn_tile duration (HIP events; 0 unless profile_kernels) */
  float   knn_fallback_us;  /* k_knn_fallback duration                                    */
  uint32_t stragglers;      /* queries resolved by the fallback in this iteration         */
  uint32_t reserved;        /* development counter: heavy tiles a wide launch (first iterations) counted -- the first 1024 go to the
                             * wave-per-query pass; 0 in the settled iterations */
} lsgpu_iter_trace;

int  lsgpu_icp_create(const lsgpu_icp_config* cfg, int device, lsgpu_icp** out);
void lsgpu_icp_destroy(lsgpu_icp* h);

/* Steps 2-3 of ICP::compute: centre the (already filtered) reference on its mean, build the voxel
 * grid.  `normals` = the descriptor SamplingSurfaceNormalDataPointsFilter attached (yaml:5-7). */
int lsgpu_icp_set_reference(lsgpu_icp* h, const float* ref_xyz1, const float* ref_normals, int64_t nr);

/* Steps 5-7 of ICP::compute on the (already filtered) reading.  T_out = T_init on failure.  A T_init that is not rigid
 * (lsgpu_check_rigid: |1 - det R| > 1e-3) is LSGPU_BAD_ARG -- step 5 is a RigidTransformation::compute, which throws
 * PointMatcher's TransformationError for it; neither call site of the reference corrects its guess
 * (laser_track.cpp:489-496, incremental_estimator.cpp:92-108). */
int lsgpu_icp_align(lsgpu_icp* h, const float* reading_xyz1, int64_t nq, const float T_init[16],
                    float T_out[16], lsgpu_icp_stats* stats);

/* ---- many independent scan pairs on one GPU (BASELINE config 3) --------------------
 * Replaces a loop of `icp_.compute` calls over independent pairs (laser_slam/src/laser_track.cpp:496 has
 * no state across calls).  Pair i runs {set_reference, align} on handles[i % n_handles]; the handles work
 * concurrently, each on its own HIP stream driven by its own host thread, so that small clouds (200 k
 * points do not fill 256 CUs) overlap on the device.  Results do not depend on n_handles.  All handles
 * must live on the same device.  rc[i] receives pair i's return code (LSGPU_NO_CONVERGENCE leaves
 * T_out[i] = T_init[i], like lsgpu_icp_align); the function returns the first non-OK, non-NO_CONVERGENCE
 * code, else LSGPU_NO_CONVERGENCE if any pair failed to converge, else LSGPU_OK.
 * T_init / T_out: 16 floats per pair, column major.  stats and rc may be NULL.
 * Reference reuse: a pair whose reference_xyz1[i], reference_normals[i] and n_reference[i] equal those of the previous
 * pair of the same handle (pair i - n_handles) skips set_reference -- the sorted reference, chunks and cell tables depend
 * on the reference alone (stats[i].reference_reused = 1 for such a pair).  Results are bit-identical either way. */
int lsgpu_icp_align_batch(lsgpu_icp* const* handles, int n_handles, int64_t n_pairs,
                          const float* const* reference_xyz1, const float* const* reference_normals,
                          const int64_t* n_reference, const float* const* reading_xyz1,
                          const int64_t* n_reading, const float* T_init, float* T_out,
                          lsgpu_icp_stats* stats, int* rc);

/* ---- one scan pair split over several GPUs (BASELINE config 4) -------------------------
 * Every rank holds the whole reference (set_reference with the same cloud) and ITS shard of the reading.
 * After lsgpu_icp_comm_init, lsgpu_icp_align treats its `reading_xyz1` as the local shard: per iteration
 * the three select histograms (3 x 2048 u32) and the 29 normal-equation sums are all-reduced over RCCL on
 * the handle's stream, so every rank computes the same limit, the same 6x6 system and the same T.
 * The unique id comes from rank 0 (lsgpu_comm_get_unique_id) and travels to the other ranks by any
 * means (torch.distributed broadcast in laser_slam_amd/sharding.py).  Every rank needs >= 1 point. */
#define LSGPU_COMM_ID_BYTES 128
int lsgpu_comm_get_unique_id(void* id /* LSGPU_COMM_ID_BYTES */);
int lsgpu_icp_comm_init(lsgpu_icp* h, int rank, int nranks, const void* id);

/* Per-iteration records of the last align; returns the number written.  The records stay in device memory until this
 * call fetches them (one synchronous copy); the next alignment on the handle overwrites them. */
int lsgpu_icp_get_trace(lsgpu_icp* h, lsgpu_iter_trace* out, int cap);

/* Geometry of the voxel-hash pyramid built by the last set_reference (for roofline accounting). */
typedef struct lsgpu_icp_info {
  int64_t  n_reference;
  int      bits_per_axis;      /* level 0 has 2^bits cells per axis        */
  int      fine_bits;          /* key bits per axis below level 0           */
  float    cell_size;          /* level-0 edge [m]                          */
  uint32_t n_chunks;           /* <=64-point chunks (32 B descriptor each)  */
  uint32_t cells[17];          /* occupied cells per level                  */
  uint64_t table_bytes;        /* hash tables, all levels                   */
} lsgpu_icp_info;
int lsgpu_icp_get_info(lsgpu_icp* h, lsgpu_icp_info* out);

/* What the handle's launch policy remembers ACROSS calls (nothing of it changes a result; it decides what a call costs).
 * The per-alignment decisions are in lsgpu_icp_stats; these outlive an alignment. */
typedef struct lsgpu_policy_info {
  int   index_rest;          /* alignments that will still leave the direction index alone: the handle found it slower than
                                the voxel grid (two timed searches per alignment); LSGPU_NO_INDEX_REST=1 switches the
                                judgement off, a reference of another size resets it */
  float pay_voxel_us;        /* the two timings of the last alignment that took them (0: none yet): the voxel-grid search in */
  float pay_index_us;        /* front of the index's first use / the first settled search through the index */
  int   ssn_sort_fallbacks;  /* reference filters this handle had to repeat with the segmented sorts because the sort-free
                                levels gave up (more candidates around a median than a workgroup selects among) */
  int   ssn_calls;           /* reference filters run by this handle */
  int   reserved[3];
} lsgpu_policy_info;
int lsgpu_icp_get_policy_info(lsgpu_icp* h, lsgpu_policy_info* out);

/* The float mean subtracted from the reference (T_refIn_refMean translation). */
int lsgpu_icp_get_reference_mean(lsgpu_icp* h, float mean[3]);

/* ---- kernel-level entry points (parity tests / profiling); all in the reference-MEAN frame ---- */

/* KDTreeMatcher::findClosests knn=1 eps=0 (yaml:9-12): ids index the reference as given to
 * set_reference, d2 = squared distance.  T (may be NULL = identity) is applied to each query on load. */
int lsgpu_knn(lsgpu_icp* h, const float* query_xyz1, int64_t nq, const float T[16], int32_t* ids,
              float* d2);
/* TrimmedDistOutlierFilter (yaml:14-16): limit = sorted(d2)[floor(n*ratio)]. */
int lsgpu_trim_limit(lsgpu_icp* h, const float* d2, int64_t n, float ratio, float* limit);
/* PointToPlaneErrorMinimizer accumulation (yaml:18-19): out = 21 upper-tri of sum J J^T (row major
 * order a<=c), 6 of -sum J r, sum w, sum w r^2  -> double[29]. */
int lsgpu_normal_eq(lsgpu_icp* h, const float* query_xyz1, int64_t nq, const float T[16],
                    const int32_t* ids, const float* d2, float limit, double out[29]);
/* RigidTransformation::compute on features (laser_track.cpp:265,485): out = T * xyz1. */
int lsgpu_transform_points(lsgpu_icp* h, const float T[16], const float* xyz1, int64_t n, float* out);
/* RigidTransformation::compute on a 3-row descriptor of the cloud (`normals`, `observationDirections`: the descriptors
 * upstream rotates, laser_track.cpp:265,485,630,643 when stored scans carry them): out = R * d, 3 floats per point
 * (a 3 x N column-major matrix), same fma chain as the points without the translation.  LSGPU_BAD_ARG if T is not rigid
 * (TransformationError upstream). */
int lsgpu_rotate_descriptors(lsgpu_icp* h, const float T[16], const float* desc3, int64_t n, float* out);

/* ---- the whole of ICP::compute on the device ----------------------------------
 * The sampling filters of icp_default.yaml:1-7 as device kernels, and the complete call
 * `icp_.compute(reading, reference, T_init)` (laser_slam/src/laser_track.cpp:496,
 * incremental_estimator.cpp:108) = reference filter, set_reference, reading filter, align, without the
 * clouds leaving the GPU.  Draws: the library's own stream with the std::srand/std::rand sequence of
 * glibc (csrc/lsgpu_rand.h); seed >= 0 reseeds it, seed < 0 continues it.  The device filters, the host
 * filters below and the oracle produce the same points, in the same order, with the same normals. */
typedef struct lsgpu_chain_config {
  float   reading_prob;     /* RandomSamplingDataPointsFilter.prob            yaml:2-3 (0.5); < 0: NO reading filter
                             * module (a yaml without readingDataPointsFilters): every point, no draw consumed */
  int     ssn_knn;          /* SamplingSurfaceNormalDataPointsFilter.knn      yaml:6-7 (10)   */
  float   ssn_ratio;        /* SamplingSurfaceNormalDataPointsFilter.ratio    yaml:6-7 (0.5)  */
  int     pad_;
  int64_t seed;             /* >= 0: reseed before the reference filter; < 0: continue        */
} lsgpu_chain_config;
void lsgpu_chain_config_yaml(lsgpu_chain_config* c);     /* icp_default.yaml values           */
void lsgpu_chain_config_default(lsgpu_chain_config* c);  /* ICP::setDefault(): 0.75, 7, 0.5   */

/* SamplingSurfaceNormalDataPointsFilter on the device.  3 <= knn <= 32.  out_xyz1 (4 floats/pt) and
 * out_normals (3 floats/pt) need room for n points; host or device pointers. */
int lsgpu_icp_filter_reference(lsgpu_icp* h, const float* xyz1, int64_t n, int knn, float ratio,
                               int64_t seed, float* out_xyz1, float* out_normals, int64_t* n_out);
/* RandomSamplingDataPointsFilter on the device: keeps point i iff draw_i < prob, order preserved. */
int lsgpu_icp_filter_reading(lsgpu_icp* h, const float* xyz1, int64_t n, float prob, int64_t seed,
                             float* out_xyz1, int64_t* n_out);
/* ICP::compute.  Returns like lsgpu_icp_align (LSGPU_NO_CONVERGENCE also when a filter leaves no
 * point); stats->t_reserved[0] = milliseconds spent in the two filters + set_reference. */
int lsgpu_icp_compute(lsgpu_icp* h, const float* reading_xyz1, int64_t nq, const float* reference_xyz1,
                      int64_t nr, const float T_init[16], const lsgpu_chain_config* chain,
                      float T_out[16], lsgpu_icp_stats* stats);

/* ---- clouds kept in HBM between calls (persistent sub-maps) ------------------------
 * LaserTrack::localScanToSubMap (laser_slam/src/laser_track.cpp:466-519) rebuilds its sub-map at every scan
 * from the last `nscan_in_sub_map` scans: a 4x4 * 4xN transform and a concatenate per extra scan, all on the
 * host, then hands reading and sub-map to icp_.compute.  Here a scan is uploaded once into a numbered slot of
 * the handle; lsgpu_icp_compute_clouds assembles the sub-map on the device (out_i = T_i * cloud_i with the
 * arithmetic of lsgpu_transform_points, concatenated in the order given) and runs the whole ICP::compute on it.
 * Slots are small non-negative integers chosen by the caller; uploading to a used slot replaces its cloud. */
int lsgpu_cloud_upload(lsgpu_icp* h, int slot, const float* xyz1, int64_t n);
int lsgpu_cloud_release(lsgpu_icp* h, int slot);
int lsgpu_cloud_size(lsgpu_icp* h, int slot, int64_t* n);   /* n = -1: empty slot */
/* reading = cloud `reading_slot`; reference = concat_i ( T_i * cloud ref_slots[i] ), ref_T = 16 floats per
 * reference cloud, column major (NULL: all identity).  Otherwise exactly lsgpu_icp_compute. */
int lsgpu_icp_compute_clouds(lsgpu_icp* h, int reading_slot, const int* ref_slots, const float* ref_T,
                             int n_ref, const float T_init[16], const lsgpu_chain_config* chain,
                             float T_out[16], lsgpu_icp_stats* stats);
/* lsgpu_cloud_upload(h, reading_slot, reading_xyz1, nq) followed by lsgpu_icp_compute_clouds(h, reading_slot, ...), in one
 * call: the new scan (host memory) crosses PCIe WHILE the sub-map is assembled and filtered.  This is the call shape of
 * LaserTrack::processLaserScan -> localScanToSubMap (laser_slam/src/laser_track.cpp:112-119, 466-519): every scan is matched
 * exactly once, right after it arrived, against scans that are already resident; uploaded first and matched afterwards, a
 * 1 M-point scan's copy is 0.3 ms of idle device inside the reference's own timed region (scan_matching_times_,
 * laser_track.cpp:128, 208-209).  Same results, same draws, same return codes as the two calls one after the other; the
 * slot holds the scan on return whatever the registration's outcome.  `reading_slot` must not be one of `ref_slots`
 * (then, and for a device pointer or an empty cloud, the two calls are simply made one after the other). */
int lsgpu_icp_compute_clouds_upload(lsgpu_icp* h, int reading_slot, const float* reading_xyz1, int64_t nq,
                                    const int* ref_slots, const float* ref_T, int n_ref, const float T_init[16],
                                    const lsgpu_chain_config* chain, float T_out[16], lsgpu_icp_stats* stats);

/* ---- local-map maintenance on the device --------------------------------------------
 * What the ROS worker does to its local map between scans (laser_slam_ros/src/laser_slam_worker.cpp:415-488,
 * 522-540): cylindrical crop around the robot, voxel-grid down-sampling, rigid re-transform after a loop
 * closure (= lsgpu_transform_points).  Inputs / outputs: host or device pointers, 4 floats per point. */
/* applyCylindricalFilter (laser_slam_ros/include/laser_slam_ros/common.hpp:194-223): keeps the points with
 * (x-cx)^2 + (y-cy)^2 <= r^2 and |z-cz| <= height/2 (remove_point_inside: the points with >= in either test).
 * Order preserved.  out_xyz1 needs room for n points. */
int lsgpu_filter_cylinder(lsgpu_icp* h, const float* xyz1, int64_t n, const float center[3], double radius_m,
                          double height_m, int remove_point_inside, float* out_xyz1, int64_t* n_out);
/* pcl::VoxelGrid<PointXYZ> (laser_slam_worker.cpp:70-72, 439-440): one centroid (float sums in input order,
 * divided by the count) per voxel with >= min_points points, voxels in ascending index order.  LSGPU_BAD_ARG if
 * the voxel index would overflow an int (PCL refuses such leaf sizes as well). */
int lsgpu_filter_voxel_grid(lsgpu_icp* h, const float* xyz1, int64_t n, const float leaf[3], int min_points,
                            float* out_xyz1, int64_t* n_out);

/* ---- the input filter chain -------------------------------------------------
 * LaserTrack loads a libpointmatcher DataPointsFilters chain from `icp_input_filters_file` (laser_slam/src/
 * laser_track.cpp:24-30, LOG(FATAL) if the file cannot be opened) and applies it to every incoming scan before the
 * scan is stored or matched (laser_track.cpp:81, :146: input_filters_.apply(scan.scan)).  The filters below run on the
 * device, one after the other, each on the output of the previous one, order of the surviving points preserved:
 *   MaxDistDataPointsFilter      dim -1: keep |p| <  |maxDist|        dim 0..2: keep  p[dim]  <  maxDist (SIGNED, as upstream)
 *   MinDistDataPointsFilter      dim -1: keep |p| >  |minDist|        dim 0..2: keep |p[dim]| >  minDist
 *   BoundingBoxDataPointsFilter  inside = xMin < x < xMax && ...;     keeps inside (removeInside 0) or outside (1)
 *   FixStepSamplingDataPointsFilter  keeps points phase, phase + step, ...; phase = rand() % step; afterwards
 *                                step *= stepMult, clamped at endStep (the step persists from scan to scan: `state`)
 *   RandomSamplingDataPointsFilter   keeps point i iff draw_i < prob
 *   RemoveNaNDataPointsFilter        drops the points with a NaN among their four feature rows x, y, z, pad (Inf stays)
 * (MaxPointCountDataPointsFilter is NOT offered: which points it keeps depends on the libpointmatcher version --
 *  std::random_shuffle in the 1.2 line, sequential selection sampling later -- and on the C++ library's shuffle; it
 *  cannot be restated from the reference, which configures none.  Unknown module names are configuration errors.)
 * |p| = sqrt(fma(z,z,fma(y,y,x*x))) in float.  Draws: the library's glibc-sequence stream (see lsgpu_icp_compute);
 * seed >= 0 reseeds it before the first filter.  LSGPU_NO_CONVERGENCE if a filter is handed an empty cloud
 * (PointMatcher::ConvergenceError "no points to filter" upstream); an empty chain copies the cloud. */
enum {
  LSGPU_FILTER_MAX_DIST = 1,
  LSGPU_FILTER_MIN_DIST = 2,
  LSGPU_FILTER_BOUNDING_BOX = 3,
  LSGPU_FILTER_FIX_STEP_SAMPLING = 4,
  LSGPU_FILTER_RANDOM_SAMPLING = 5,
  LSGPU_FILTER_REMOVE_NAN = 6
};
typedef struct lsgpu_point_filter {
  int    type;     /* LSGPU_FILTER_*                                                                          */
  int    dim;      /* Max/MinDist: -1 radial, 0..2 one axis                                                   */
  int    flag;     /* BoundingBox: removeInside                                                               */
  int    pad_;
  float  v[6];     /* MaxDist {maxDist}  MinDist {minDist}  BoundingBox {xMin,xMax,yMin,yMax,zMin,zMax}        */
                   /* FixStepSampling {startStep,endStep,stepMult}  RandomSampling {prob}                      */
  double state;    /* FixStepSampling: current step (0: start at startStep); updated by every apply            */
} lsgpu_point_filter;
int lsgpu_apply_point_filters(lsgpu_icp* h, lsgpu_point_filter* filters, int n_filters, const float* xyz1,
                              int64_t n, int64_t seed, float* out_xyz1, int64_t* n_out);

/* ---- the ROS message surface of the scan path ----------------------------------
 * LaserSlamWorker::scanCallback turns the incoming sensor_msgs/PointCloud2 into DataPoints with
 * PointMatcher_ros::rosMsgToPointMatcherCloud<float> (laser_slam_ros/src/laser_slam_worker.cpp:125) and publishes
 * clouds through lpmToPcl / pcl::toROSMsg (laser_slam_ros/include/laser_slam_ros/common.hpp:159-191).  Both are
 * pure layout changes and run on the device so that a scan crosses PCIe once, as the message's byte block.
 *   from: `data` = the message's data block (n_points records of point_step bytes, host or device memory), the
 *         FLOAT32 fields x, y, z at byte offsets off_x/y/z inside a record (any alignment), optionally byte-swapped
 *         (is_bigendian); out = x,y,z,1 per point.  drop_non_finite (= !is_dense): records with a NaN / Inf
 *         coordinate are removed, order preserved.
 *   to  : x,y,z,1 -> the data block of a PointCloud2 / pcl::PointCloud<pcl::PointXYZ> with fields x@0 y@4 z@8,
 *         point_step 16 (what pcl::toROSMsg emits for PointXYZ; the 4th float of a record is padding, written 1). */
int lsgpu_cloud_from_pointcloud2(lsgpu_icp* h, const unsigned char* data, int64_t n_points, int point_step, int off_x,
                                 int off_y, int off_z, int is_bigendian, int drop_non_finite, float* out_xyz1,
                                 int64_t* n_out);
int lsgpu_cloud_to_pointxyz(lsgpu_icp* h, const float* xyz1, int64_t n, unsigned char* out_data /* 16 n bytes */);

/* ---- host-side versions of the two filters (same output as the device filters) and O(1) helpers ---- */

/* RandomSamplingDataPointsFilter (yaml:1-3): keep i iff draw_i < prob; seed as above. */
int64_t lsgpu_filter_random_sampling(int64_t n, float prob, int64_t seed, int64_t* keep_idx);
/* SamplingSurfaceNormalDataPointsFilter (yaml:5-7), samplingMethod 0, keepNormals 1. */
int64_t lsgpu_filter_sampling_surface_normal(const float* xyz1, int64_t n, int knn, float ratio,
                                             int64_t seed, float* out_xyz1, float* out_normals);
/* RigidTransformation::checkParameters / correctParameters (common.hpp:136-149). */
int  lsgpu_check_rigid(const float T[16]);
void lsgpu_correct_rigid(const float T[16], float out[16]);
/* The rotation metric of DifferentialTransformationChecker (yaml:24-27) between two 4x4 transforms (column major):
 * Quaternion(R_a).angularDistance(Quaternion(R_b)) = 2 atan2(|vec|, |w|) of q_a * conj(q_b), in float -- the same code
 * the device-side checker runs (csrc/lsgpu_host_math.h). */
float lsgpu_rotation_distance(const float Ta[16], const float Tb[16]);

const char* lsgpu_strerror(int code);
const char* lsgpu_last_error(lsgpu_icp* h); /* detail of the last failure on this handle */
int         lsgpu_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
