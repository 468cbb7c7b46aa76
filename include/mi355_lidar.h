/* mi355_lidar.h — C-ABI of the MI355X-native LiDAR odometry front-end (libmi355_lidar.so).
 *
 * This is the drop-in boundary for the hot path of RyuYamamoto/lidar_graph_slam: the PCL / fast_gicp / ndt_omp
 * calls made by the `points_prefiltering` and `lidar_scan_matcher` nodes (and the loop-closure registration of
 * `graph_based_slam`).  Every entry point names the reference interface it replaces (file:line under the
 * reference tree).  extern "C", plain pointers and sizes, opaque handles, no C++/torch types, no exceptions:
 * every function returns 0 on success and a negative MI355_E* code otherwise; mi355_last_error() gives the text.
 *
 * Clouds are (base pointer, count, byte stride): xyz as 3 floats at offset 0, intensity as a float at
 * `intensity_offset` bytes — stride 32 / intensity 16 for pcl::PointXYZI, stride 16 / intensity 12 for the packed
 * float4 device layout.  Host-pointer entry points copy in/out; the *_dev entry points take device pointers
 * (e.g. torch tensor .data_ptr()) in the packed float4 layout and run on the context's HIP stream.
 * One handle may be used from one thread at a time; different handles are independent.
 */
#ifndef MI355_LIDAR_H_
#define MI355_LIDAR_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header changes layout or an entry point changes meaning: 2 = mi355_pointcloud2 gained
 * `data_size` (mandatory) and a non-FLOAT32 intensity field is ignored as pcl::fromROSMsg ignores it; 3 = mi355_batch_diagnostics,
 * mi355_multi_lane_allocations, mi355_multi_batch_register_clouds, mi355_batch_set_*_views; 4 = mi355_reg_gicp_trial_cost (additions only);
 * 5 = mi355_result gained `evaluations` (+ a reserved word: 392 -> 400 bytes; n_inliers is an inlier count again for every engine) and
 * mi355_batch_get_cloud was added.
 * mi355_abi_version() returns the library's: a caller compares it with the MI355_ABI_VERSION it was compiled against. */
#define MI355_ABI_VERSION 5
int mi355_abi_version(void);

#define MI355_OK 0
#define MI355_EINVAL (-1)   /* bad argument */
#define MI355_EHIP (-2)     /* HIP runtime error (text in mi355_last_error) */
#define MI355_ECAPACITY (-3)/* output buffer too small */
#define MI355_ESTATE (-4)   /* call order (e.g. align before set_target) */

typedef struct mi355_ctx mi355_ctx;
typedef struct mi355_reg mi355_reg;
typedef struct mi355_batch mi355_batch;

/* Cloud layout descriptor. */
typedef struct mi355_cloud_view {
  const void* data;
  size_t n;
  size_t stride;            /* bytes between points */
  size_t intensity_offset;  /* bytes from point start to the intensity float */
} mi355_cloud_view;

/* registration_method of lidar_scan_matcher.param.yaml:4 / lidar_scan_matcher.cpp:37,55 */
enum { MI355_METHOD_FAST_GICP = 0, MI355_METHOD_NDT_OMP = 1,
       MI355_METHOD_ICP = 2, /* pcl::IterativeClosestPoint: the back end's default (graph_based_slam.param.yaml:9, cpp:142-151) */
       MI355_METHOD_GICP = 3 /* pclomp::GeneralizedIterativeClosestPoint, PCL 1.8's BFGS GICP (lidar_scan_matcher.cpp:73-96,
                                graph_based_slam.cpp:120-141) */ };
/* fast_gicp::RegularizationMethod (default PLANE) */
enum { MI355_REG_NONE = 0, MI355_REG_MIN_EIG = 1, MI355_REG_NORMALIZED_MIN_EIG = 2, MI355_REG_PLANE = 3, MI355_REG_FROBENIUS = 4 };
/* fast_gicp::LSQ_OPTIMIZER_TYPE (default LevenbergMarquardt) */
enum { MI355_OPT_LM = 0, MI355_OPT_GN = 1 };

/* fast_gicp::FastGICP setters — lidar_scan_matcher.cpp:48-52 (values: lidar_scan_matcher.param.yaml:9-11,18-19)
 * and graph_based_slam.cpp:94-98. */
typedef struct mi355_gicp_params {
  int k_correspondences;              /* setCorrespondenceRandomness(20) */
  int max_iterations;                 /* setMaximumIterations(64) */
  double transformation_epsilon;      /* setTransformationEpsilon(0.01) */
  double rotation_epsilon;            /* fast_gicp default 2e-3 */
  double max_correspondence_distance; /* setMaxCorrespondenceDistance(2.0) */
  int regularization_method;          /* MI355_REG_PLANE */
  int optimizer;                      /* MI355_OPT_LM */
  int lm_max_iterations;              /* 10 */
  double lm_init_lambda_factor;       /* 1e-9 */
  int force_iterations;               /* 1 = benchmark config "20 GICP iters": run exactly max_iterations — neither the
                                         convergence test nor an exhausted LM trial loop ends the optimisation */
  int num_threads;                    /* setNumThreads — accepted for interface parity, unused on the GPU */
} mi355_gicp_params;

/* pclomp::NormalDistributionsTransform setters — lidar_scan_matcher.cpp:65-70 (yaml:9-11,14-15),
 * graph_based_slam.cpp:112-117. Neighbourhood search is DIRECT7 (lidar_scan_matcher.cpp:69). */
typedef struct mi355_ndt_params {
  double transformation_epsilon; /* 0.01 */
  double step_size;              /* 0.1 */
  double resolution;             /* 2.0 front end / 1.0 back end */
  int max_iterations;            /* 64 */
  double outlier_ratio;          /* PCL default 0.55 */
  int min_points_per_voxel;      /* 6 */
  double min_covar_eigvalue_mult;/* 0.01 */
  int legacy_mt_skip;            /* 1 (default): the More-Thuente loop never iterates, as in ndt_omp / PCL 1.8
                                    (`interval_converged = (step_max - step_min) > 0`); 0: PCL >= 1.10's line search */
  int num_threads;               /* unused on the GPU */
} mi355_ndt_params;

/* pcl::IterativeClosestPoint as graph_based_slam.cpp:142-151 configures it (the setters' values are the defaults here). */
typedef struct mi355_icp_params {
  double max_correspondence_distance; /* setMaxCorrespondenceDistance(30): a pair is kept iff d2 <= max^2 */
  int max_iterations;                 /* setMaximumIterations(100); reaching it counts as converged (DefaultConvergenceCriteria) */
  double transformation_epsilon;      /* setTransformationEpsilon(1e-8): compared with the SQUARED incremental translation */
  double euclidean_fitness_epsilon;   /* setEuclideanFitnessEpsilon(1e-6): relative change of the mean squared distance */
  double rotation_epsilon;            /* 0: PCL's default, cos(angle) >= 1 - transformation_epsilon */
  int ransac_iterations;              /* setRANSACIterations(0): only 0 (no outlier rejector) is built */
} mi355_icp_params;

/* pclomp::GeneralizedIterativeClosestPoint as the nodes configure it (lidar_scan_matcher.cpp:84-93,
 * graph_based_slam.cpp:131-138); what they do not set keeps PCL's default. */
typedef struct mi355_gicp_bfgs_params {
  double max_correspondence_distance; /* setMaxCorrespondenceDistance(correspondence_distance): kept iff d2 < max^2 */
  int max_iterations;                 /* setMaximumIterations: outer (re-correspondence) iterations */
  int max_optimizer_iterations;       /* setMaximumOptimizerIterations(max_optimizer_iteration -> 20): BFGS steps per outer iteration */
  double transformation_epsilon;      /* setTransformationEpsilon: converged when every translation entry moved less than this ... */
  double rotation_epsilon;            /* ... and every rotation entry less than this (PCL default 2e-3) */
  double euclidean_fitness_epsilon;   /* setEuclideanFitnessEpsilon: stored, not used by GICP's own convergence test (as upstream) */
  int k_correspondences;              /* PCL default 20: neighbours of a covariance (singular values replaced by 1, 1, 1e-3) */
  int ransac_iterations;              /* setRANSACIterations(0): only 0 is built */
} mi355_gicp_bfgs_params;

/* points_prefiltering parameters — points_prefiltering.launch.xml:2-13. */
typedef struct mi355_prefilter_params {
  double min_distance;   /* min_distance_cloud 1.0; <= 0 disables the distance filter (points_prefiltering.cpp:102-112) */
  double leaf_size;      /* leaf_size 0.1 (BASELINE: 0.2); <= 0 disables VoxelGrid (cpp:114-121) */
  int mean_k;            /* mean_k 30; <= 0 disables StatisticalOutlierRemoval (cpp:132-140) */
  double stddev_mul;     /* stddev 1.2 */
} mi355_prefilter_params;

/* Result of one alignment: what hasConverged()/getFinalTransformation()/getFitnessScore() expose
 * (lidar_scan_matcher.cpp:167,172; graph_based_slam.cpp:320-322). */
typedef struct mi355_result {
  float T[16];              /* final_transformation_, row-major 4x4 */
  int converged;
  int iterations;
  double error;             /* GICP: last accepted sum of Mahalanobis errors; NDT: final score */
  double H[36];             /* GICP: final_hessian_; NDT: last Hessian */
  double trans_probability; /* NDT */
  int n_source;             /* source points actually registered (after any prefilter) */
  int n_target;
  int n_inliers;            /* FAST_GICP / GICP: correspondences within max_correspondence_distance at the last linearisation /
                               evaluation; 0 for NDT and ICP */
  int status;               /* batch mode: MI355_PAIR_OK, or why the pair was not registered (T = the guess, converged = 0) */
  int evaluations;          /* NDT: computeDerivatives / computeHessian passes over the source the pair asked for; GICP: cost
                               evaluations of the BFGS line searches; 0 for the other engines (ABI 5) */
  int reserved;
} mi355_result;
#define MI355_PAIR_OK 0
#define MI355_PAIR_TOO_FEW_POINTS 1 /* source or target holds fewer than k_correspondences points after the prefilter
                                       (mi355_reg_set_source / _target reject such clouds with MI355_EINVAL) */

void mi355_gicp_default_params(mi355_gicp_params* p);
void mi355_ndt_default_params(mi355_ndt_params* p);
void mi355_prefilter_default_params(mi355_prefilter_params* p);
void mi355_icp_default_params(mi355_icp_params* p);
void mi355_gicp_bfgs_default_params(mi355_gicp_bfgs_params* p);

/* ---- context ------------------------------------------------------------------------------------------- */
mi355_ctx* mi355_create(int device_id);
void mi355_destroy(mi355_ctx* ctx);
const char* mi355_last_error(const mi355_ctx* ctx);
/* own_stream == 0: run on the caller's HIP stream `hip_stream` exactly as given — a NULL handle is the legacy default stream
 * (torch.cuda.current_stream().cuda_stream is 0 for torch's default stream), so the library's work is ordered with what the
 * caller enqueued there.  own_stream != 0: the library creates a private non-blocking stream (the state after mi355_create);
 * inputs handed to the *_dev entry points must then be complete before the call. */
int mi355_set_stream(mi355_ctx* ctx, void* hip_stream, int own_stream);
int mi355_synchronize(mi355_ctx* ctx);
/* NN-grid cell edge in metres for the voxel-hash searches (default 0.5). */
int mi355_set_grid_cell(mi355_ctx* ctx, float cell);
/* device name / CU count, for the benchmark log */
int mi355_device_info(mi355_ctx* ctx, char* name, size_t name_cap, int* compute_units);

/* ---- filters: replace the PCL filter objects of points_prefiltering.cpp:146-148 ------------------------ */
/* PointsPreFiltering::distance_filter (points_prefiltering.cpp:102-112). out: packed float4. */
int mi355_distance_filter(mi355_ctx* ctx, const mi355_cloud_view* in, double min_distance, float* out, size_t out_cap, size_t* out_n);
/* pcl::VoxelGrid::setLeafSize + filter (points_prefiltering.cpp:118-120; graph_based_slam.cpp:61,311-313). */
int mi355_voxel_grid(mi355_ctx* ctx, const mi355_cloud_view* in, float leaf, float* out, size_t out_cap, size_t* out_n, int* overflow);
/* pcl::StatisticalOutlierRemoval::setMeanK/setStddevMulThresh/filter (points_prefiltering.cpp:136-139).
 * keep_mask (n bytes) and mean_dist (n floats) are optional diagnostics. */
int mi355_sor(mi355_ctx* ctx, const mi355_cloud_view* in, int mean_k, double stddev_mul, float* out, size_t out_cap, size_t* out_n,
              uint8_t* keep_mask, float* mean_dist);
/* PointsPreFiltering::callback_sensor_points pipeline (points_prefiltering.cpp:65-87): distance -> voxel -> SOR, on device. */
int mi355_prefilter(mi355_ctx* ctx, const mi355_cloud_view* in, const mi355_prefilter_params* p, float* out, size_t out_cap, size_t* out_n);
/* pcl::transformPointCloud (lidar_scan_matcher.cpp:282,291). */
int mi355_transform(mi355_ctx* ctx, const mi355_cloud_view* in, const float T16[16], float* out);

/* ---- stage-level entry points (parity tests of the search / covariance kernels) ------------------------ */
/* Exact k-NN of `queries` in `pts` (pcl::search::KdTree::nearestKSearch semantics: ascending d2; ties -> lower index). */
int mi355_knn(mi355_ctx* ctx, const mi355_cloud_view* pts, const mi355_cloud_view* queries, int k, int* idx, float* d2);
/* fast_gicp calculate_covariances: per-point k-NN covariance + regularisation; cov6 = (xx,xy,xz,yy,yz,zz) per point. */
int mi355_covariances(mi355_ctx* ctx, const mi355_cloud_view* pts, int k, int method, float* cov6);

/* ---- registration object: replaces pcl::Registration<PointXYZI,PointXYZI>::Ptr registration_ ----------- */
/* new fast_gicp::FastGICP / new pclomp::NormalDistributionsTransform (lidar_scan_matcher.cpp:38-39,56-57;
 * graph_based_slam.cpp:84-85,103-104) / new pcl::IterativeClosestPoint (graph_based_slam.cpp:143-144).
 * params: mi355_gicp_params*, mi355_ndt_params* or mi355_icp_params* by method (NULL = defaults). */
mi355_reg* mi355_reg_create(mi355_ctx* ctx, int method, const void* params);
void mi355_reg_destroy(mi355_reg* reg);
int mi355_reg_set_params(mi355_reg* reg, const void* params);
/* setInputTarget (lidar_scan_matcher.cpp:149,212).  ALWAYS rebuilds: the node mutates the same cloud in place and
 * re-passes the same pointer (cpp:187,209,212), so pointer identity must not short-circuit. */
int mi355_reg_set_target(mi355_reg* reg, const mi355_cloud_view* cloud);
/* setInputSource (lidar_scan_matcher.cpp:162). */
int mi355_reg_set_source(mi355_reg* reg, const mi355_cloud_view* cloud);
/* align(output, guess) + hasConverged + getFinalTransformation (lidar_scan_matcher.cpp:165-172; guess NULL = identity as
 * graph_based_slam.cpp:318).  aligned_out (optional): source transformed by the result, packed float4, n_source points. */
int mi355_reg_align(mi355_reg* reg, const float guess16[16], mi355_result* result, float* aligned_out);
/* getFitnessScore(max_range) (graph_based_slam.cpp:321): mean squared 1-NN distance of the aligned source. */
int mi355_reg_fitness(mi355_reg* reg, double max_range, double* score);
/* One linearisation at pose T (row-major 4x4 double): the 6x6 normal equations of FastGICP::linearize, for parity tests.
 * corr (optional, n_source ints): target index per source point in the CALLER's point order, -1 = none. */
int mi355_reg_gicp_linearize(mi355_reg* reg, const double T16[16], double* H36, double* b6, double* error, int* corr);
/* FastGICP::compute_error(T_trial) with the correspondences and Mahalanobis matrices of a linearisation at T_lin (fast_gicp_impl.hpp,
 * LsqRegistration::step_lm's trial cost), for parity tests: model_cost = as the optimisation prices its trials (from the sums of the
 * linearisation pass, no pass over the points); direct_cost = the reference's loop over the points. */
int mi355_reg_gicp_trial_cost(mi355_reg* reg, const double T_lin16[16], const double T_trial16[16], double* model_cost, double* direct_cost);
/* NDT computeDerivatives at p6 = (tx,ty,tz,roll,pitch,yaw); hessian_fp64 selects computeHessian's fp64 path. */
int mi355_reg_ndt_derivatives(mi355_reg* reg, const double p6[6], double* score, double* grad6, double* hess36, int hessian_fp64);
/* NDT leaves (VoxelGridCovariance) for parity: arrays sized by mi355_reg_ndt_num_cells(); sorted by linear voxel index. */
int mi355_reg_ndt_num_cells(mi355_reg* reg, size_t* n_cells);
int mi355_reg_ndt_cells(mi355_reg* reg, int64_t* leaf_idx, int* leaf_n, double* mean3, double* cov9, double* icov9, int* min_b3, int* div_b3);

/* ---- device-resident clouds: the key-frame ring --------------------------------------------- */
/* The odometry node rebuilds its sub-map at every key frame by transforming up to 20 stored key-frame clouds on the CPU and
 * concatenating them (lidar_scan_matcher.cpp:187-212); the back end does the same around a loop candidate
 * (graph_based_slam.cpp:297-309).  A mi355_cloud keeps a cloud in HBM: uploaded once, transformed / concatenated by kernels,
 * handed to a registration object without crossing PCIe again.  A cloud that has been a GICP source carries its per-point
 * covariances; transforms rotate them (C' = R C R^T), so a sub-map assembled with carry_covariances needs only its spatial
 * index rebuilt, not the k-NN covariance pass over ~3 x 10^5 points. */
typedef struct mi355_cloud mi355_cloud;
mi355_cloud* mi355_cloud_upload(mi355_ctx* ctx, const mi355_cloud_view* points);
void mi355_cloud_destroy(mi355_cloud* cloud);
size_t mi355_cloud_size(const mi355_cloud* cloud);
int mi355_cloud_has_covariances(const mi355_cloud* cloud);
/* packed float4, the caller's point order; capacity in points */
int mi355_cloud_download(mi355_cloud* cloud, float* out_packed, size_t capacity);
/* pcl::transformPointCloud (lidar_scan_matcher.cpp:275-294) on the device: a NEW cloud */
mi355_cloud* mi355_cloud_transform(mi355_cloud* cloud, const float T16[16]);
/* T_0 cloud_0 ++ T_1 cloud_1 ++ ... in list order (lidar_scan_matcher.cpp:199-210): a NEW cloud.  carry_covariances != 0
 * requires every input to carry covariances (MI355_ESTATE otherwise) and gives the result the rotated ones. */
mi355_cloud* mi355_cloud_concat(mi355_ctx* ctx, mi355_cloud* const* clouds, const float* T16s, int n, int carry_covariances);
/* fast_gicp's calculate_covariances for a cloud that has not been a registration source (key frame #0 of the odometry node is
 * the first TARGET): computed once, kept with the cloud. */
int mi355_cloud_compute_covariances(mi355_cloud* cloud, int k_correspondences, int regularization_method);
/* PointsPreFiltering's chain (points_prefiltering.cpp:66-100; VoxelGrid alone = graph_based_slam.cpp:311-313) on a
 * device-resident cloud: a NEW device cloud, nothing crosses PCIe but the resulting point count. */
mi355_cloud* mi355_cloud_prefilter(mi355_cloud* cloud, const mi355_prefilter_params* params);
/* setInputSource / setInputTarget from a device cloud.  FAST_GICP: a source cloud without covariances gets them computed and
 * cached (same k / regularisation as the registration object); a target cloud that carries covariances uses them instead
 * of recomputing over the union (carry_covariances above), one that does not gets them computed as mi355_reg_set_target does. */
int mi355_reg_set_source_cloud(mi355_reg* reg, mi355_cloud* cloud);
int mi355_reg_set_target_cloud(mi355_reg* reg, mi355_cloud* cloud);

/* ---- wire / on-disk formats ------------------------------------------------------------------ */
/* sensor_msgs/PointCloud2 as the nodes receive and publish it (pcl::fromROSMsg at lidar_scan_matcher.cpp:127,196,203 and
 * points_prefiltering.cpp:70; pcl::toROSMsg at lidar_scan_matcher.cpp:145,196 and graph_based_slam.cpp:462): the raw message
 * bytes cross PCIe once and are unpacked / packed by a kernel.  Field offsets and datatypes are the message's `fields`
 * entries for x, y, z, intensity (sensor_msgs/PointField: INT8=1 UINT8=2 INT16=3 UINT16=4 INT32=5 UINT32=6 FLOAT32=7 FLOAT64=8);
 * x, y, z must be FLOAT32; an absent intensity field is off_intensity = -1 (intensity 0, as fromROSMsg leaves it), and so is an
 * intensity field whose datatype is not FLOAT32: pcl::fromROSMsg maps a field only when name, datatype and count match the
 * point type's (pcl::PointXYZI::intensity is a float) and otherwise warns and leaves the default 0 - nothing is converted. */
typedef struct mi355_pointcloud2 {
  const void* data;      /* msg.data */
  uint32_t width, height;
  uint32_t point_step, row_step;
  int32_t off_x, off_y, off_z, off_intensity;
  uint8_t intensity_datatype;
  uint8_t is_bigendian;  /* must be 0 (every ROS 2 platform is little-endian) */
  size_t data_size;      /* msg.data.size(): must cover row_step * height bytes (MI355_EINVAL otherwise - an inconsistent message
                            is refused, not read past its end).  ABI note: this field was appended in round 3 (MI355_ABI_VERSION 2);
                            a caller built against the older struct, or one that leaves it 0, is refused for every non-empty message */
} mi355_pointcloud2;
mi355_cloud* mi355_cloud_from_pointcloud2(mi355_ctx* ctx, const mi355_pointcloud2* msg);
/* pcl::toROSMsg(pcl::PointCloud<pcl::PointXYZI>): point_step 32, x@0 y@4 z@8 (data[3] = 1.0f @12) intensity@16, padding 0;
 * width = n, height = 1, row_step = 32 n.  out: 32 n bytes. */
int mi355_cloud_to_pointcloud2(mi355_cloud* cloud, void* out_data, size_t capacity_bytes);
/* pcl::io::savePCDFile (graph_based_slam.cpp:499: ASCII by default, binary_mode = 0): FIELDS x y z intensity, SIZE 4 4 4 4,
 * TYPE F F F F, WIDTH n, HEIGHT 1, VIEWPOINT 0 0 0 1 0 0 0, values with 8 significant digits.  Host-only (no GPU needed). */
int mi355_pcd_write(const char* path, const mi355_cloud_view* points, int binary);
int mi355_cloud_save_pcd(mi355_cloud* cloud, const char* path, int binary);

/* ---- batch mode: many independent scan pairs per launch (BASELINE configs 2 and 5) --------------------- */
/* n_sources source clouds and n_targets target clouds, each up to `capacity` points; pair p aligns source p to
 * target tgt_of_src[p] (graph_based_slam.cpp:311-322 generalised from 1 candidate per tick to all candidates).
 * method: MI355_METHOD_FAST_GICP (params = mi355_gicp_params), MI355_METHOD_NDT_OMP (mi355_ndt_params), MI355_METHOD_ICP
 * (mi355_icp_params; the back end's yaml default) or MI355_METHOD_GICP (mi355_gicp_bfgs_params; pclomp's BFGS GICP) — the four engines
 * GraphBasedSLAM::get_registration builds (graph_based_slam.cpp:80-152).  With NDT_OMP every pair's Newton / More-Thuente state, with
 * ICP every pair's fit and convergence criteria, with GICP every pair's line search and BFGS update live in device memory and all pairs
 * advance together - one pass over the sources of the whole batch per evaluation, nothing read back in between; mi355_batch_align then
 * BLOCKS until the last pair is done (FAST_GICP only enqueues).  Results: T, converged, iterations, error (NDT: the score; ICP: the
 * mean squared correspondence distance; GICP: the minimiser's last cost, n_inliers the correspondences of the last evaluation), H and
 * trans_probability (NDT); fitness as FAST_GICP. */
mi355_batch* mi355_batch_create(mi355_ctx* ctx, int method, const void* params, int n_sources, int n_targets, size_t capacity);
void mi355_batch_destroy(mi355_batch* b);
/* Device-resident inputs: packed float4 [n_clouds][capacity] + host counts.  prefilter (optional) is applied on device
 * (distance + voxel; SOR when mean_k > 0).  Enqueues work; does not synchronise.  LIFETIME of d_points: the preparation chains that read
 * it run on internal side streams forked from the batch's stream at the call and joined by the next mi355_batch_align - the buffer must
 * stay valid and unchanged until the stream has passed THAT align (with mi355_batch_set_side_streams(b, 0): until it has passed this
 * call).  After that the batch holds its own copies of everything it needs, later aligns included. */
int mi355_batch_set_targets_dev(mi355_batch* b, const void* d_points, const int* counts, const mi355_prefilter_params* prefilter);
int mi355_batch_set_sources_dev(mi355_batch* b, const void* d_points, const int* counts, const mi355_prefilter_params* prefilter);
/* Host-pointer variants (PCIe copy included). */
int mi355_batch_set_targets(mi355_batch* b, const float* points, const int* counts, const mi355_prefilter_params* prefilter);
int mi355_batch_set_sources(mi355_batch* b, const float* points, const int* counts, const mi355_prefilter_params* prefilter);
/* Host clouds as views, one per slot (any stride / intensity offset; at most `capacity` points each): packed through pinned
 * staging and sent a run of slots per copy, the packing of the next run overlapping the transfer of the last.  The caller's buffers
 * are free when the call returns. */
int mi355_batch_set_targets_views(mi355_batch* b, const mi355_cloud_view* views, const mi355_prefilter_params* prefilter);
int mi355_batch_set_sources_views(mi355_batch* b, const mi355_cloud_view* views, const mi355_prefilter_params* prefilter);
/* The same from device clouds (one per slot, at most `capacity` points each): copied slot by slot on the device. */
int mi355_batch_set_targets_clouds(mi355_batch* b, mi355_cloud* const* clouds, const mi355_prefilter_params* prefilter);
int mi355_batch_set_sources_clouds(mi355_batch* b, mi355_cloud* const* clouds, const mi355_prefilter_params* prefilter);
/* tgt_of_src: n_sources ints (NULL = identity mapping, requires n_targets == n_sources); guesses: n_sources x 16 floats (NULL = I).
 * Enqueues every alignment; does not synchronise. */
int mi355_batch_align(mi355_batch* b, const int* tgt_of_src, const float* guesses16, int with_fitness);
/* Synchronises and copies results (n_sources entries) and optional fitness scores. */
int mi355_batch_results(mi355_batch* b, mi355_result* results, double* fitness);
/* Use only the first n_sources / n_targets slots of the batch (1 <= live <= what mi355_batch_create was given) until the next call:
 * a caller that keeps ONE batch object for requests of varying size (LoopClosureCore, the lanes of mi355_multi) pays nothing for
 * the slots a request does not fill.  Invalidates the sources, targets and results that were set before.
 * The launch geometry (blocks per pair, which search kernel) stays that of the slots the batch was CREATED with, and with it the order
 * of a pair's fp64 partial sums: a request registered on a larger batch object equals the same request on one created for its size to
 * summation order (poses to ~1e-6 m), not bit for bit; two requests on the same object are bit-equal. */
int mi355_batch_set_live(mi355_batch* b, int n_sources, int n_targets);
/* Parity probe of update_correspondences (fast_gicp F3), after mi355_batch_align of a FAST_GICP batch: the correspondences the
 * INCREMENTAL search holds at the final poses (margins: DESIGN 4.3) against a search of every source point from scratch
 * at the same poses.  *n_mismatch must be 0: the incremental search is exact, not approximate.  Leaves the results untouched. */
int mi355_batch_verify_correspondences(mi355_batch* b, long long* n_checked, long long* n_mismatch);
/* Diagnostics of the last mi355_batch_align of a FAST_GICP batch, for the benchmark log (synchronises the batch; every output is
 * optional).  first_converged[n_sources]: the iterations a pair had run when fast_gicp's is_converged (lsq_registration_impl.hpp,
 * SURVEY F9) first held, -1 = never - with force_iterations that is what the pair would have run without it.
 * knn_points[2] / knn_handed_over[2] (0 = targets, 1 = sources): the points of each side's last covariance launch and how many
 * of them the wave-cooperative tile search handed over to the one-wave-per-query kernel. */
int mi355_batch_diagnostics(mi355_batch* b, int* first_converged, long long* knn_points, long long* knn_handed_over);
/* Parity probe of the batched preparation chain (prefilter -> voxel hash -> k-NN covariances), after mi355_batch_set_targets* /
 * _sources*: cloud `index` of the targets (targets != 0) or sources AS THE BATCH REGISTERS IT - the points that survived the prefilter, in
 * the prefilter's output order (pcl::VoxelGrid: ascending voxel index), and, for the engines that use them (FAST_GICP, GICP), their
 * regularised covariances (6 floats xx xy xz yy yz zz per point; out_cov6 may be NULL).  Synchronises the batch.  *n = the point count;
 * nothing is copied when it exceeds `capacity` (MI355_ECAPACITY).  Not available for the NDT targets (they are kept as voxel Gaussians). */
int mi355_batch_get_cloud(mi355_batch* b, int targets, int index, float* out_xyzi, float* out_cov6, size_t capacity, size_t* n);
/* Per-kernel-family timing for the benchmark's roofline block.  With timing enabled the library records HIP events on the
 * launching stream around every launch of the hot kernels (k_knn_cov_tile, k_gicp_correspond, k_gicp_linearize, ...; short
 * kernels that always run back to back, e.g. the 12 launches of one radix sort, share one bracket).
 * mi355_batch_kernel_times synchronises the batch, returns one entry per family (total milliseconds and number of brackets
 * since the last call); *n_out is the number of families that ran.  The counters are reset only when every family fitted into
 * `capacity` - a caller whose array was too small calls again with *n_out entries and loses nothing. */
typedef struct mi355_kernel_time {
  char name[64];
  double total_ms;
  int launches;
  int reserved;
} mi355_kernel_time;
/* Where mi355_batch_set_targets* / _sources* run their preparation chains: 1 = on two internal side streams that overlap each other
 * (joined by mi355_batch_align), 0 = on the batch's own stream, one after the other, -1 = the library default (1).  Results are
 * identical; 0 is what a profiler wants when it asks how long a kernel takes by itself. */
int mi355_batch_set_side_streams(mi355_batch* b, int mode);
int mi355_batch_set_timing(mi355_batch* b, int enable);
int mi355_batch_kernel_times(mi355_batch* b, mi355_kernel_time* out, int capacity, int* n_out);

/* Run this batch on its own HIP stream instead of the context's (use_context_stream != 0 switches back).  Batches on
 * different streams overlap on the GPU: while one batch is in its (latency-bound) LM iterations the next one can be
 * prefiltered and indexed - the double buffering a streaming caller (one batch of loop-closure candidates / scan pairs
 * after the other) wants.  Device pointers handed to mi355_batch_set_*_dev must be valid in THAT stream's order. */
int mi355_batch_set_stream(mi355_batch* batch, void* hip_stream, int use_context_stream);

/* ---- multi-device: one host process, one context + stream per device (SURVEY 8b "ctx_or_multi", 8e) ---------- */
/* The loop-closure back end is a C++ node (graph_based_slam.cpp:238-352) and registers one candidate per tick (:311-322);
 * BASELINE configs[4] registers thousands of independent candidates at once, sharded over the GPUs of one node.  A mi355_multi
 * owns one mi355_ctx (its stream, its buffers) per entry of device_ids — the same ordinal may appear twice: two independent
 * lanes on one device.  Pairs are dealt to lanes by target (sub-map) so that every sub-map is uploaded, filtered and indexed
 * once: targets by descending pair count (ties: lower target id), each to the least-loaded lane (ties: lower lane) — the rule
 * of lidar_graph_slam_amd/sharding.py, whose torchrun form (one process per GPU, results all-gathered over RCCL) produces the
 * same partition.  Here the process is one, so the only exchange step between lanes — the fixed-size mi355_result
 * records, 400 bytes per pair — is a D2H copy per lane; every lane is driven by its own host thread.  (This host-view entry point
 * also uploads every source and target cloud of the call to its lane; callers whose clouds are device-resident use
 * mi355_multi_context + mi355_cloud_* + mi355_multi_batch_register_clouds and pay that once.)  No pair crosses lanes: results are those of a single batch, pair for pair. */
typedef struct mi355_multi mi355_multi;
mi355_multi* mi355_multi_create(const int* device_ids, int n_devices);
void mi355_multi_destroy(mi355_multi* m);
const char* mi355_multi_last_error(const mi355_multi* m);
int mi355_multi_device_count(const mi355_multi* m);
/* lane i's context (for device clouds / filters on that device); owned by the multi object */
mi355_ctx* mi355_multi_context(mi355_multi* m, int lane);
/* how many times lane i's batch object has been (re)created: 1 after the first registration and unchanged while later requests
 * fit what it was sized for and ask for the same method / parameters (monitoring; -1 = bad lane) */
int mi355_multi_lane_allocations(const mi355_multi* m, int lane);
/* the partition alone (host only): lane_of_pair[p] for n_pairs pairs over n_lanes lanes */
int mi355_multi_shard_by_target(const int* tgt_of_src, int n_pairs, int n_lanes, int* lane_of_pair);
/* mi355_batch_* over all lanes in one call: pair p aligns sources[p] to targets[tgt_of_src[p]] (host views, any stride) from
 * guesses16 + 16 p (NULL = identity, graph_based_slam.cpp:318).  method / params as mi355_batch_create; the prefilters
 * (optional) run on the device (graph_based_slam.cpp:311-313: VoxelGrid 0.5 m on the sub-map).  results: n_sources entries;
 * fitness (optional, with_fitness != 0): getFitnessScore per pair (:321); lane_of_pair (optional): where each pair ran.
 * Every lane keeps its batch object between calls (grow-only). */
int mi355_multi_batch_register(mi355_multi* m, int method, const void* params, const mi355_cloud_view* sources, int n_sources,
                               const mi355_cloud_view* targets, int n_targets, const int* tgt_of_src, const float* guesses16,
                               const mi355_prefilter_params* source_prefilter, const mi355_prefilter_params* target_prefilter, int with_fitness,
                               mi355_result* results, double* fitness, int* lane_of_pair);
/* The same for clouds that already live on the devices (mi355_cloud_* on mi355_multi_context(m, lane)): source p and the sub-map it
 * registers against must belong to the lane mi355_multi_shard_by_target deals pair p to (MI355_EINVAL otherwise) - a caller that
 * keeps its key frames device-resident per lane, as LoopClosureCore does, uploads nothing here. */
int mi355_multi_batch_register_clouds(mi355_multi* m, int method, const void* params, mi355_cloud* const* sources, int n_sources,
                                      mi355_cloud* const* targets, int n_targets, const int* tgt_of_src, const float* guesses16,
                                      const mi355_prefilter_params* source_prefilter, const mi355_prefilter_params* target_prefilter,
                                      int with_fitness, mi355_result* results, double* fitness, int* lane_of_pair);

#ifdef __cplusplus
}
#endif
#endif /* MI355_LIDAR_H_ */
