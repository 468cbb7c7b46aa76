/* mi355_loop_closure.h — C-ABI of LoopClosureCore, the ROS-free restatement of the registration half of
 * GraphBasedSLAM::optimization_callback (/root/reference/graph_based_slam/src/graph_based_slam.cpp:238-328): candidate
 * gate (:256-280), +-N key-frame sub-map (:297-309), 0.5 m VoxelGrid (:61,:311-313), align from identity (:315-318),
 * getFitnessScore acceptance (:321-328) and the relative pose handed to the pose graph (:330-341).
 * The reference closes ONE loop per 1 Hz tick; here any number of query key frames go through one batched launch
 * sequence (BASELINE config 5).  The pose graph itself (GTSAM iSAM2, :331-349) stays with the caller. */
#ifndef MI355_LOOP_CLOSURE_H_
#define MI355_LOOP_CLOSURE_H_
#include "mi355_lidar.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355_loop_closure mi355_loop_closure;

typedef struct mi355_loop_closure_params {
  int search_key_frame_num;               /* graph_based_slam.param.yaml:4 -> 20: sub-map = candidate +- this many key frames */
  double score_threshold;                 /* yaml:6 -> 0.3 (:328) */
  double search_for_candidate_threshold;  /* yaml:7 -> 15.0 m (:273) */
  double accumulate_distance_threshold;   /* yaml:8 -> 100.0 m (:266) */
  float submap_leaf;                      /* 0.5 m, hard-coded at :61 */
  int max_iteration;                      /* yaml:12 -> 100 */
  double transformation_epsilon;          /* yaml:14 -> 0.01 */
  double max_correspondence_distance;     /* yaml:21 -> 2.0 */
  int correspondence_randomness;          /* yaml:23 -> 20 */
  int registration_method;                /* yaml:9 -> "ICP" = MI355_METHOD_ICP (pcl::IterativeClosestPoint with the values of
                                             graph_based_slam.cpp:145-149); ICP, MI355_METHOD_FAST_GICP (:80-100),
                                             MI355_METHOD_NDT_OMP (:101-119) and MI355_METHOD_GICP (:120-141, pclomp's BFGS
                                             GICP) all register the candidates of a lane in one batch */
  double step_size;                       /* yaml:17 -> 0.1 (NDT_OMP) */
  double ndt_resolution;                  /* yaml:18 -> 1.0 (NDT_OMP) */
  double correspondence_distance;         /* yaml:26 -> 2.0 (GICP) */
  int max_optimizer_iteration;            /* yaml:23 -> 20 (GICP) */
  int ransac_iteration;                   /* yaml:29 -> 0 (GICP: setRANSACIterations, graph_based_slam.cpp:131,139; ICP hard-codes 0 at
                                             :149).  Only 0 - no RANSAC correspondence rejector - is built: create fails otherwise */
} mi355_loop_closure_params;

typedef struct mi355_loop_result {
  int query_id;        /* key frame treated as "latest" */
  int candidate_id;    /* min_id of :256-280, or -1: no candidate, nothing registered */
  int converged;       /* hasConverged() */
  int accepted;        /* converged && !(score_threshold < fitness)  (:328) */
  int iterations;
  double fitness;      /* getFitnessScore() (:321) */
  float transform[16]; /* getFinalTransformation() (:320), row-major */
  double between[16];  /* pose_from.between(pose_to): pose_from = transform * pose(query), pose_to = pose(candidate) (:331-340) */
  size_t target_points; /* sub-map size after the 0.5 m VoxelGrid */
} mi355_loop_result;

void mi355_loop_closure_default_params(mi355_loop_closure_params* p);
/* No GPU is touched before the first mi355_loop_closure_close call. */
mi355_loop_closure* mi355_loop_closure_create(int device_id, const mi355_loop_closure_params* params);
/* The same over several devices from one process (one context + stream per device): the candidates of one
 * mi355_loop_closure_close call are dealt to the devices by sub-map (the rule of mi355_multi_shard_by_target), every device
 * registers its share with one batch launch sequence under its own host thread, results come back in query order.  An ordinal
 * may appear twice (two lanes on one device).  Results are those of the single-device object, query for query. */
mi355_loop_closure* mi355_loop_closure_create_multi(const int* device_ids, int n_devices, const mi355_loop_closure_params* params);
void mi355_loop_closure_destroy(mi355_loop_closure* h);
const char* mi355_loop_closure_last_error(const mi355_loop_closure* h);
/* key_frame_callback's bookkeeping (:354-380 without the graph factors): cloud in the key frame's own frame (KeyFrame.msg),
 * pose7 = position xyz + quaternion xyzw (geometry_msgs/Pose), accumulated path length.  Returns the key frame id (>= 0). */
int mi355_loop_closure_add_key_frame(mi355_loop_closure* h, const mi355_cloud_view* cloud, const double pose7[7], double accum_distance);
/* Overwrite a key frame's pose (adjust_pose after an optimisation, :417-432). */
int mi355_loop_closure_set_pose(mi355_loop_closure* h, int key_frame_id, const double pose7[7]);
/* The candidate gate alone (host only): *candidate_id = min_id or -1. */
int mi355_loop_closure_candidate(const mi355_loop_closure* h, int query_id, int* candidate_id);
/* Register every query key frame against the sub-map around its candidate, all pairs in one batch. */
int mi355_loop_closure_close(mi355_loop_closure* h, const int* query_ids, int n_queries, mi355_loop_result* results);

#ifdef __cplusplus
}
#endif
#endif
