/*
 * icp_oracle.h -- CPU restatement of the ICP chain laser_slam configures.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (laser_slam_amd/, include/)
 * links, loads or calls this library; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, and there only as checker / baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in libpointmatcher +
 * libnabo + Eigen (dependencies.rosinstall:23-28, un-vendored, un-pinned), none
 * of which is under /root/reference or installed here, and the reference holds
 * no golden vectors (laser_slam/test/test_empty.cpp:3-5 is its whole test
 * suite).  This file restates the published libpointmatcher algorithm for the
 * module chain of laser_slam/configurations/icp_default.yaml:1-29 and is
 * anchored on the reference's call sites (laser_slam/src/laser_track.cpp:496,
 * laser_slam/src/incremental_estimator.cpp:108).  It is cross-checked against
 * brute force, scipy.cKDTree (dev only) and analytic known-answer scenes.
 *
 * Conventions (laser_slam/include/laser_slam/common.hpp:14-17):
 *   clouds are PointMatcher<float>::DataPoints.features, (dim+1) x N column
 *   major  ==  AoS  x,y,z,1  (16 B / point);  normals are 3 x N column major.
 *   4x4 transforms are column major float (Eigen default).
 *   T maps reading -> reference:  p_ref = T * p_reading.
 *
 * RESTATEMENT CHOICES -- where upstream's behaviour is not derivable from /root/reference and this file (and the
 * product, which shares the choice) had to pick one.  devtools/dump_for_upstream.py writes the inputs, the filters'
 * outputs and this file's per-iteration trace in a form a real libpointmatcher build can replay (INTEGRATION.md,
 * "Diffing against a real libpointmatcher"); [in brackets] the dumped file / column in which a wrong choice shows.  Each line names what would change if the choice were wrong and
 * the test that pins the choice to the mathematics (not to upstream).  A session with libpointmatcher available
 * should diff exactly these:
 *   1. rank test of a box (SamplingSurfaceNormal::fuseRange): FullPivHouseholderQR::rank() + 1 >= 3, default
 *      threshold eps * 3 (rounds 1-2 restated it with FullPivLU pivots).  Wrong => borderline-thin boxes kept /
 *      dropped differently (a few points of 1 M).  tests: test_independent_known_answers (collinear boxes dropped,
 *      planar ones kept, normals == numpy.linalg.eigh).  [reference_filtered.csv: number of rows]
 *   2. split of a box: std::nth_element leaves ties and the order inside the halves unspecified; here "stable sort,
 *      split at the median".  Wrong => other box memberships where coordinates tie, another order in which the boxes'
 *      points take their rand() draws (ratio < 1: other points kept).  tests: test_filter_golden_vectors_reproduce,
 *      test_device_reference_filter_is_bit_identical (pin device == host == oracle, not upstream);
 *      test_surface_normal_filter_boxes_against_a_numpy_recursion (the stated rule as a plain numpy recursion).  [reference_filtered.csv: row order, normals]
 *   3. kd-tree ties (libnabo): implementation defined => any nearest point is valid; the product returns the
 *      smallest index of its own order.  tests: _check_nn / test_golden_vectors accept equal-distance alternatives.  [none: the trace is the same for every
 *      valid tie order]
 *   4. TrimmedDistOutlierFilter: index floor(float(n) * ratio) over the matched pairs, weight 1 iff d2 <= limit
 *      (inclusive).  Wrong ('<') => pairs at exactly the limit lose their weight (>= 1 pair, more with ties).
 *      tests: test_independent_known_answers (numpy.partition index), test_trim_limit_is_order_statistic.
 *      [oracle_trace.csv: limit, n_used]
 *   5. reference mean: accumulated in double, rounded to float (Eigen's rowwise().sum() in float would differ in the
 *      last bits of the centring).  Affects T at the 1e-6 level.  [oracle_trace.csv: last digits of T00..T33]
 *   6. minimiser: A and b from float J, accumulated in double (accum_double 1) or float (0, as upstream);
 *      x = A.llt().solve(b) in float; LLT failure => ConvergenceError (newer upstream falls back to a QR / SVD
 *      min-norm solve).  tests: test_independent_known_answers (numpy Cholesky), test_point_to_plane_matches_numpy_lstsq.
 *      [oracle_trace.csv: T00..T33 from row 0 on]
 *   7. DifferentialTransformationChecker: Eigen >= 3.3 angularDistance = 2 atan2(|vec|, |w|) of q_a conj(q_b) (older
 *      Eigen: 2 acos(|dot|), equal to ~1e-7 rad).  tests: test_independent_known_answers (scipy Rotation.magnitude).  [oracle_trace.csv: number of rows]
 *   8. rand(): glibc's srand/rand sequence; draw = (float)rand() / (float)RAND_MAX, kept iff draw < prob; one stream
 *      consumed in the order reference filter -> reading filter -> (input filters, per scan).  A caller that wants
 *      reproducible runs reseeds per call (seed >= 0); upstream never seeds.  tests: test_draw_stream_is_the_glibc_
 *      rand_sequence; the sequence tests run both "reseed per call" and "one continuing stream".
 *      On ERROR paths the product's stream position may differ from a libpointmatcher process': lsgpu_icp_compute
 *      consumes the reading filter's draws as soon as the reference filter has run (the total is known), so a failure
 *      after that point (HIP error, empty grid) leaves them consumed although the filter never ran.  A guess that is not
 *      rigid is refused AFTER both filters have consumed their draws, as upstream does (the C++ facade hands it to the
 *      device all the same; tests/cpp/shim_check.cpp compares the stream positions); only a facade without a device, or with
 *      the LSGPU_TEST_SEAMS compute override, refuses it before any draw.  No call site of laser_slam continues after either.
 *      [reading_filtered.csv, and reference_filtered.csv for ratio < 1: which rows]
 *   9. MaxDist / MinDist input filters: radial branch compares the norm with |limit| (both), one-axis branch compares
 *      the SIGNED coordinate (MaxDist) / the absolute one (MinDist); an empty cloud into a non-empty chain throws.
 *      tests: test_input_filters_upstream_asymmetries.  [input_filtered.csv: rows]
 *  10. output order of SamplingSurfaceNormal: upstream's inPlaceFilter takes the draws in box-traversal order, collects the
 *      kept indices, then does std::sort(indicesToKeep) and compacts the cloud in place ("bring the data we keep to the
 *      front of the arrays") -- the filtered cloud is in ascending ORIGINAL index, every point with its box's normal
 *      (from knowledge of libpointmatcher, like choice 1; rounds 1-5 emitted in traversal order, the round-5 verdict's
 *      catch).  Wrong => the same points in another row order: other ids, another float order of the reference mean
 *      (T at the 1e-6 level).  tests: test_surface_normal_filter_boxes_against_a_numpy_recursion (point for point against
 *      a numpy recursion that sorts the kept indices), test_filter_golden_vectors_reproduce, test_device_reference_filter_
 *      is_bit_identical.  [reference_filtered.csv: row order]
 *  11. eigenvectors of a box: upstream runs Eigen::EigenSolver<Matrix3f> (general real solver, float) on C and takes the
 *      real part of the eigenvector of the smallest eigenvalue; here a cyclic Jacobi iteration in double on the float C,
 *      normalised in double, rounded to float (lsgpu_box_normal.h / jacobi3: the same source on host and device).  The
 *      two agree to float rounding (1e-6 in the normal's components, sign included only up to the solver's convention:
 *      the point-to-plane error is even in the normal); a near-isotropic box (two or three equal eigenvalues) may pick
 *      another vector of the degenerate eigenspace.  tests: test_independent_known_answers (numpy.linalg.eigh).
 *      [reference_filtered.csv: columns nx, ny, nz]
 * The COMPOSITION of ICP::compute (frames, left-multiplied update, checker window, steps 1-7 of SURVEY.md A.1) is pinned
 * against a float64 numpy / scipy ICP that shares no code with this file:
 * test_oracle_loop_against_an_independent_numpy_icp.
 *
 * Arithmetic definitions shared with the HIP path (so integer results are
 * bit-comparable):
 *   transform : x' = fma(m02,z, fma(m01,y, fma(m00,x, m03)))   (per row)
 *   dist^2    : fma(dz,dz, fma(dy,dy, dx*dx))
 */
#ifndef LS_ICP_ORACLE_H_
#define LS_ICP_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* icp_default.yaml:1-29 (values) / ICP::setDefault (laser_track.cpp:18-21). */
typedef struct lso_config {
  float reading_sampling_prob;  /* RandomSamplingDataPointsFilter prob   (yaml 0.5, default 0.75); < 0: no reading filter module */
  int   surface_normal_knn;     /* SamplingSurfaceNormal knn             (yaml 10,  default 7)    */
  float surface_normal_ratio;   /* SamplingSurfaceNormal ratio           (default 0.5)            */
  float trim_ratio;             /* TrimmedDistOutlierFilter ratio        (yaml 0.75, default .85) */
  int   max_iterations;         /* CounterTransformationChecker          (40)                     */
  float min_diff_rot;           /* DifferentialTransformationChecker minDiffRotErr   (0.001)      */
  float min_diff_trans;         /* DifferentialTransformationChecker minDiffTransErr (yaml 0.01)  */
  int   smooth_length;          /* DifferentialTransformationChecker smoothLength    (yaml 4)     */
  int   accum_double;           /* 0: float A,b as libpointmatcher; 1: double accumulation        */
  int   num_threads;            /* OpenMP threads for the NN query loop (libnabo: omp if built so)*/
} lso_config;

void lso_config_yaml(lso_config* c);     /* icp_default.yaml values   */
void lso_config_default(lso_config* c);  /* ICP::setDefault() values  */

/* One record per ICP iteration, for CPU-vs-GPU trace diffs. */
typedef struct lso_iter_trace {
  float  T_iter[16];   /* after this iteration's left-multiply */
  float  limit;        /* trimmed squared-distance limit       */
  int64_t n_used;      /* number of weights == 1               */
  double A[36];        /* normal matrix (row-major, symmetric) */
  double b[6];
  double x[6];         /* solution [rot(3); trans(3)]          */
} lso_iter_trace;

typedef struct lso_stats {
  int   iterations;
  int   converged;        /* 1: differential checker stopped it, 0: counter */
  float final_limit;
  int64_t final_n_used;
  double t_filter_ms, t_build_ms, t_loop_ms;
} lso_stats;

enum { LSO_OK = 0, LSO_NO_CONVERGENCE = 1, LSO_BAD_ARG = 2 };

/* RigidTransformation::compute on features / normals descriptor. */
void lso_transform_points(const float T[16], const float* xyz1, int64_t n, float* out_xyz1);
void lso_rotate_normals(const float T[16], const float* nrm, int64_t n, float* out_nrm);
/* RigidTransformation::checkParameters / correctParameters (common.hpp:136-149). */
int  lso_check_rigid(const float T[16]);
float lso_rotation_distance(const float Ta[16], const float Tb[16]); /* DifferentialTransformationChecker's rotation metric */
void lso_correct_rigid(const float T[16], float out[16]);

/* RandomSamplingDataPointsFilter: keep i iff rand()/RAND_MAX < prob.  If seed>=0 srand(seed) first. */
int64_t lso_random_sampling(int64_t n, float prob, int64_t seed, int64_t* keep_idx);
/* SamplingSurfaceNormalDataPointsFilter (samplingMethod 0, keepNormals 1). Returns n_out. */
int64_t lso_sampling_surface_normal(const float* xyz1, int64_t n, int knn, float ratio,
                                    int64_t seed, float* out_xyz1, float* out_normals);

/* KDTreeMatcher knn=1 epsilon=0: exact nearest neighbour, squared distances. */
void* lso_kdtree_build(const float* ref_xyz1, int64_t nr);
void  lso_kdtree_free(void* tree);
void  lso_kdtree_nn(const void* tree, const float* q_xyz1, int64_t nq, int32_t* ids, float* d2,
                    int num_threads);
void  lso_brute_nn(const float* ref_xyz1, int64_t nr, const float* q_xyz1, int64_t nq,
                   int32_t* ids, float* d2);

/* TrimmedDistOutlierFilter: limit = values[floor(n_valid*ratio)] of finite d2; w = d2 <= limit. */
int lso_trim_limit(const float* d2, int64_t n, float ratio, float* limit);

/* PointToPlaneErrorMinimizer: builds A (6x6), b (6) over pairs with d2<=limit, solves (float LLT),
 * returns dT (4x4 col major).  p_xyz1 = reading already moved by T_iter.                           */
int lso_point_to_plane(const float* p_xyz1, const float* ref_xyz1, const float* ref_nrm,
                       const int32_t* ids, const float* d2, float limit, int64_t nq,
                       int accum_double, double A[36], double b[6], double x[6], float dT[16],
                       int64_t* n_used);

/* Steps 2..7 of ICP::compute on ALREADY FILTERED clouds (reference carries normals). */
int lso_icp_compute(const lso_config* cfg, const float* reading_xyz1, int64_t nq,
                    const float* ref_xyz1, const float* ref_nrm, int64_t nr,
                    const float T_init[16], float T_out[16], lso_stats* stats,
                    lso_iter_trace* trace, int trace_cap);

/* Whole ICP::compute incl. both filter chains (K1, K2) -- laser_track.cpp:496. */
int lso_icp_compute_full(const lso_config* cfg, const float* reading_xyz1, int64_t nq,
                         const float* ref_xyz1, int64_t nr, const float T_init[16],
                         int64_t seed, float T_out[16], lso_stats* stats);

/* ---- local-map maintenance of the ROS worker ------------------------------------
 * applyCylindricalFilter, laser_slam_ros/include/laser_slam_ros/common.hpp:194-223: keep (or remove) the
 * points inside the vertical cylinder |(x,y) - c| <= radius, |z - cz| <= height / 2.  Order preserved. */
int64_t lso_cylinder_filter(const float* xyz1, int64_t n, const float center[3], double radius_m,
                            double height_m, int remove_point_inside, float* out_xyz1);
/* pcl::VoxelGrid<PointXYZ> as configured at laser_slam_ros/src/laser_slam_worker.cpp:70-72 and applied at
 * :439-440: one centroid per voxel holding >= min_points points, voxels in ascending index order
 * (index = i + j * div_x + k * div_x * div_y).  PCL sorts (index, point) pairs with an unstable sort, so the
 * order of the float additions inside a voxel is unspecified upstream; here it is the input order.
 * Returns the number of output points, or -1 if the index would overflow an int (PCL refuses as well). */
int64_t lso_voxel_grid(const float* xyz1, int64_t n, const float leaf[3], int min_points, float* out_xyz1);

/* ---- the input filter chain: laser_slam/src/laser_track.cpp:24-30 loads a libpointmatcher
 * DataPointsFilters chain, :81 and :146 apply it to every incoming scan.  The chain file is not in the reference
 * repository; this restates the published semantics of the five filters the HIP path supports (libpointmatcher
 * DataPointsFilters/{MaxDist,MinDist,BoundingBox,FixStepSampling,RandomSampling}.cpp), one after the other, each on
 * the output of the previous one.  Field meaning as in include/lsgpu_icp.h (lsgpu_point_filter); draws: libc rand().
 * Returns the number of output points, or -1 if a filter is handed an empty cloud ("no points to filter"). */
typedef struct lso_point_filter {
  int    type;    /* 1 MaxDist, 2 MinDist, 3 BoundingBox, 4 FixStepSampling, 5 RandomSampling */
  int    dim;
  int    flag;
  int    pad_;
  float  v[6];
  double state;
} lso_point_filter;
int64_t lso_apply_point_filters(lso_point_filter* filters, int n_filters, const float* xyz1, int64_t n,
                                int64_t seed, float* out_xyz1);

#ifdef __cplusplus
}
#endif
#endif
