"""How long does the host take to ENQUEUE one frame (python + ctypes + torch allocations)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
G, clip, dec, scene = bench.build_scene(torch, torch.device("cuda", 0), n_frames=64)
ring = torch.zeros(64, 512, 512, 3, dtype=torch.uint8, device="cuda")
for i in range(5): clip.render_u8(i, out=ring[i:i+1])
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(50): clip.render_u8(i, out=ring[i:i+1])
t_enq = (time.perf_counter() - t) / 50
torch.cuda.synchronize()
t_tot = (time.perf_counter() - t) / 50
print("host enqueue per frame: %.3f ms; wall per frame incl. GPU: %.3f ms" % (t_enq * 1e3, t_tot * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(50): clip.render_u8(i, out=ring[i:i+1])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
