"""The torso frame of bench.py (to_plane_cnn -> planes -> rays -> fused SuperresolutionHybrid8XDC_Warp.forward) alone -- target for rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=8)
frame, flops = bench.build_torso_frame(torch, dev, G)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for t in range(2): frame(t)
torch.cuda.synchronize()
import time; t0 = time.perf_counter()
for t in range(reps): frame(t)
torch.cuda.synchronize(); print("torso frame: %.3f ms (%d frames)" % ((time.perf_counter() - t0) / reps * 1e3, reps))
