import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from lidar_graph_slam_amd import synth
from lidar_graph_slam_amd.filters import PointsPreFiltering
from lidar_graph_slam_amd.registration import FastGICP
s, t, T = synth.make_pair()
pre = PointsPreFiltering(leaf_size=0.2, mean_k=0, min_distance_cloud=1.0)
s2, t2 = pre.callback_sensor_points(s), pre.callback_sensor_points(t)
g = FastGICP(max_iterations=20, force_iterations=1)
g.setInputTarget(t2)
def one():
    g.setInputSource(s2)
    return g.align()
for _ in range(3): one()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); one(); ts.append(time.perf_counter() - t0)
print("single pair set_source+align ms median", 1e3 * float(np.median(ts)))
ts = []
for _ in range(10):
    g.setInputSource(s2)
    t0 = time.perf_counter(); g.align(); ts.append(time.perf_counter() - t0)
print("align only ms median", 1e3 * float(np.median(ts)))
