"""The torso frame of bench.py (build_torso_frame) N times on one stream: the workload of scripts/gpu_torso_trace.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from real3dportrait_amd import TriPlaneGenerator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
G = TriPlaneGenerator().to(dev).eval()
frame, fl = bench.build_torso_frame(torch, dev, G)
for i in range(3):
    frame(i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(n):
    frame(i)
torch.cuda.synchronize()
print("torso frame: %.4f ms (%d frames, %.1f GFLOP of convs each)" % ((time.perf_counter() - t0) / n * 1e3, n, fl / 1e9))
