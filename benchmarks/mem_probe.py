import sys, os
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
from lidar_graph_slam_amd.registration import BatchRegistration
free0, total = torch.cuda.mem_get_info()
b = BatchRegistration(1024, 1024, 65536, max_iterations=20, force_iterations=1)
free1, _ = torch.cuda.mem_get_info()
print("batch_create(1024 pairs, cap 65536): %.2f GB = %.2f MB per pair" % ((free0 - free1) / 2**30, (free0 - free1) / 2**20 / 1024))
