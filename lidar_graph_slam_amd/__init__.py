"""lidar_graph_slam_amd — MI355X-native LiDAR odometry front-end (hot path of RyuYamamoto/lidar_graph_slam).

Only the hot path lives here: the HIP kernels + C-ABI (csrc/, libmi355_lidar.so), and a thin host mirror of the
reference's PCL-facing interface (filters.py, registration.py, scan_matcher.py).  Importing the package does not
load the HIP library; the first API call does, and fails loudly if it is missing or no GPU is present.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
