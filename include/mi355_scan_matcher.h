/* mi355_scan_matcher.h — C-ABI of ScanMatcherCore, the ROS-free restatement of LidarScanMatcher::callback_cloud
 * (/root/reference/lidar_scan_matcher/src/lidar_scan_matcher.cpp:122-250) with its registration on the MI355X.
 * This is the per-scan odometry loop of BASELINE config 4 (scan-to-keyframe stream): what the node's subscriber
 * callback does between "PointCloud2 decoded" and "pose published".  Parameters are the node's ROS parameters
 * (lidar_scan_matcher.param.yaml:3-19). */
#ifndef MI355_SCAN_MATCHER_H_
#define MI355_SCAN_MATCHER_H_
#include "mi355_lidar.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355_scan_matcher mi355_scan_matcher;

typedef struct mi355_scan_matcher_params {
  int registration_method;            /* MI355_METHOD_FAST_GICP | MI355_METHOD_NDT_OMP | MI355_METHOD_GICP (yaml:4; lidar_scan_matcher.cpp:34-96) */
  double displacement;                /* yaml:5 -> 1.0: key-frame distance gate (cpp:183) */
  int max_scan_accumulate_num;        /* yaml:6 -> 20: key frames in the sliding sub-map (cpp:199) */
  int max_iteration;                  /* yaml:9 -> 64 */
  int omp_num_thread;                 /* yaml:10 -> 4 (no effect on the GPU) */
  double transformation_epsilon;      /* yaml:11 -> 0.01 */
  double step_size;                   /* yaml:14 -> 0.1 */
  double ndt_resolution;              /* yaml:15 -> 2.0 */
  double max_correspondence_distance; /* yaml:18 -> 2.0 */
  int correspondence_randomness;      /* yaml:19 -> 20 */
  int reuse_key_frame_covariances;    /* 0 (the default of every wrapper, and what a drop-in must pass): target covariances are
                                         recomputed over the merged sub-map at every key frame, as fast_gicp does
                                         (lidar_scan_matcher.cpp:212 -> setInputTarget) = the reference's arithmetic.
                                         1: opt-in optimisation, NOT the reference's arithmetic: a key frame's GICP covariances
                                         are computed once and rotated into every sub-map (C' = R C R^T).  The
                                         sub-map itself is assembled on the device either way. */
  double correspondence_distance;     /* yaml:22 -> 5.0 ("GICP": setMaxCorrespondenceDistance, cpp:88) */
  int max_optimizer_iteration;        /* yaml:25 -> 20  ("GICP": setMaximumOptimizerIterations, cpp:91) */
} mi355_scan_matcher_params;

typedef struct mi355_scan_output {
  int first_scan;      /* the scan became the initial target; the node publishes no pose for it (cpp:133-160) */
  int converged;       /* 0 => the node drops the scan and keeps the previous pose (cpp:167-170) */
  int new_key_frame;
  float pose[16];      /* translation_: map <- base, row-major (cpp:172) */
  long key_frame_id;
  size_t target_points;
  int iterations;
  long n_key_frames;
  double accum_distance;
} mi355_scan_output;

mi355_scan_matcher* mi355_scan_matcher_create(int device_id, const mi355_scan_matcher_params* params);
void mi355_scan_matcher_destroy(mi355_scan_matcher* h);
const char* mi355_scan_matcher_last_error(const mi355_scan_matcher* h);
/* One subscriber callback (cpp:122-250).  base_to_sensor16: the TF looked up at cpp:129-130 (NULL = identity). */
int mi355_scan_matcher_callback_cloud(mi355_scan_matcher* h, const mi355_cloud_view* cloud, const float base_to_sensor16[16], mi355_scan_output* out);
/* Current sliding sub-map (target_cloud_), packed float4; out_packed may be NULL to query the size. */
int mi355_scan_matcher_target(mi355_scan_matcher* h, float* out_packed, size_t cap, size_t* n);

#ifdef __cplusplus
}
#endif
#endif
