"""TEST INFRASTRUCTURE: an independent NumPy restatement of the registration half of GraphBasedSLAM::optimization_callback
(/root/reference/graph_based_slam/src/graph_based_slam.cpp:238-341) that registers with the ORACLE, used to check the C++
LoopClosureCore + HIP batch registration.  Shares no code with csrc/host/loop_closure_core.cpp."""
import numpy as np

from oracle import oracle_py as O
from ref_scan_matcher import matrix_to_pose, pose_to_matrix


class RefLoopClosure:
    def __init__(self, search_key_frame_num=20, score_threshold=0.3, search_for_candidate_threshold=15.0, accumulate_distance_threshold=100.0,
                 submap_leaf=0.5, max_iteration=100, transformation_epsilon=0.01, max_correspondence_distance=2.0, correspondence_randomness=20,
                 registration_method="ICP", step_size=0.1, ndt_resolution=1.0, correspondence_distance=2.0, max_optimizer_iteration=20):
        self.method = registration_method
        self.np = O.ndt_params(resolution=ndt_resolution, step_size=step_size, transformation_epsilon=transformation_epsilon, max_iterations=max_iteration)
        self.bp = O.gicp_bfgs_params(max_correspondence_distance=correspondence_distance, max_iterations=max_iteration,
                                     max_optimizer_iterations=max_optimizer_iteration, transformation_epsilon=transformation_epsilon)
        self.ip = O.icp_params()  # graph_based_slam.cpp:145-149: 30 m, 100 iterations, 1e-8, 1e-6, no RANSAC
        self.n, self.score, self.cand_r, self.accum = search_key_frame_num, score_threshold, search_for_candidate_threshold, accumulate_distance_threshold
        self.leaf = submap_leaf
        self.gp = O.gicp_params(k_correspondences=correspondence_randomness, max_iterations=max_iteration, transformation_epsilon=transformation_epsilon,
                                max_correspondence_distance=max_correspondence_distance)
        self.kf = []  # (cloud, position, quaternion, accum_distance)

    def add_key_frame(self, cloud, position, quaternion, accum_distance):
        self.kf.append((np.asarray(cloud, np.float32), np.asarray(position, np.float64), np.asarray(quaternion, np.float64), float(accum_distance)))
        return len(self.kf) - 1

    def candidate(self, q):  # :256-280
        _, p_l, _, a_l = self.kf[q]
        best, best_d = -1, np.inf
        for i, (_, p, _, a) in enumerate(self.kf):
            if (a_l - a) < self.accum:
                continue
            d = float(np.linalg.norm(p_l - p))
            if d < self.cand_r and d < best_d:
                best, best_d = i, d
        return best

    def close_loop(self, q):  # :282-341 for one query
        c = self.candidate(q)
        if c < 0:
            return {"query_id": q, "candidate_id": -1}
        parts = []
        for k in range(c - self.n, c + self.n + 1):
            if 0 <= k < len(self.kf):
                cloud, p, quat, _ = self.kf[k]
                parts.append(O.transform(cloud, pose_to_matrix(p, quat)))
        target = O.voxel_grid(np.concatenate(parts), self.leaf)[0]
        cloud, p, quat, _ = self.kf[q]
        M_q = pose_to_matrix(p, quat)
        source = O.transform(cloud, M_q)
        if self.method == "FAST_GICP":
            r = O.gicp_align(source, target, self.gp)
        elif self.method == "NDT_OMP":
            r = O.ndt_align(source, target, self.np)
        elif self.method == "GICP":
            r = O.gicp_bfgs_align(source, target, self.bp)
        else:
            r = O.icp_align(source, target, self.ip)
        fitness = O.fitness(source, target, r["T"])
        accepted = bool(r["converged"]) and not (self.score < fitness)
        pf, qf = matrix_to_pose(np.asarray(r["T"], np.float32) @ M_q)
        A = pose_to_matrix(pf, qf).astype(np.float64)
        A[:3, :3] = _rot(qf)
        A[:3, 3] = pf
        B = np.eye(4)
        B[:3, :3] = _rot(self.kf[c][2])
        B[:3, 3] = self.kf[c][1]
        return {"query_id": q, "candidate_id": c, "converged": bool(r["converged"]), "accepted": accepted, "fitness": fitness, "T": r["T"],
                "between": np.linalg.inv(A) @ B, "target_points": len(target), "source": source, "target": target}


def _rot(q):
    from scipy.spatial.transform import Rotation

    return Rotation.from_quat(np.asarray(q, np.float64)).as_matrix()
