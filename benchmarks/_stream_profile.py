import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from lidar_graph_slam_amd import synth
from lidar_graph_slam_amd.filters import PointsPreFiltering
from lidar_graph_slam_amd.scan_matcher import LidarScanMatcher
scene = synth.Scene(1234)
poses = synth.trajectory(30)
pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
scans = [pre.callback_sensor_points(synth.cast_scan(scene, P, seed=10 + i)) for i, P in enumerate(poses)]
sm = LidarScanMatcher(registration_method="FAST_GICP")
ts = []
for s in scans:
    t0 = time.perf_counter(); o = sm.callback_cloud(s); ts.append((time.perf_counter() - t0, o["new_key_frame"], o["target_points"], o["iterations"]))
for t in ts: print(f"{t[0]*1e3:7.2f} ms  kf={int(t[1])} target={t[2]} it={t[3]}")
print("mean ms", 1e3*np.mean([t[0] for t in ts[1:]]))
