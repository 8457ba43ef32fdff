"""BASELINE configs[2]: a 5 s driving clip at 25 fps (125 frames), frame-sharded over the ranks of one node, gathered to
rank 0 over RCCL and written by frames.write_frames as packed rgb24 (there is no ffmpeg in the image; R3D_CLIP_FMT = raw | npy | ppm | png).  Synthetic per-frame inputs of
the reference's shapes: planes_t = cano + secc_t (seeded per frame index), cameras from a smoothed yaw sweep.

    python scripts/render_clip.py                     # 1 GPU
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/render_clip.py     # one node
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from real3dportrait_amd import TriPlaneGenerator, synth
from real3dportrait_amd.frames import PipelinedClipRenderer, gather_frames, shard_frames, write_frames

T_FRAMES = int(os.environ.get("R3D_CLIP_FRAMES", 125))
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
if world > 1:
    dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
Tn = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
G = TriPlaneGenerator().cuda().eval()
dec = synth.synth_decoder(0, sigma_bias=6.0)
with torch.no_grad():
    for p, v in zip((G.decoder.net[0].weight, G.decoder.net[0].bias, G.decoder.net[2].weight, G.decoder.net[2].bias), dec):
        p.copy_(Tn(v))
cano = Tn(synth.synth_planes(0, N=1))
residuals = [Tn(synth.synth_planes(100 + t, N=1, scale=0.1)) for t in range(8)]          # secc_t, cycled
yaw = np.convolve(np.sin(np.linspace(0, 2 * np.pi, T_FRAMES)) * 0.35, np.ones(5) / 5, mode="same")
cams = Tn(np.stack([synth.look_at_camera(float(y), 0.0) for y in yaw]).astype(np.float32))
ws = torch.ones(1, 14, 512, device="cuda")
clip = PipelinedClipRenderer(G, cano, residuals, cams, ws, base_seed=1, n_streams=3)
lo, hi = shard_frames(T_FRAMES, world, rank)
per = (T_FRAMES + world - 1) // world
ring = torch.zeros(per, 512, 512, 3, dtype=torch.uint8, device="cuda")
for t in range(lo, min(lo + 2, hi)):
    clip.render_u8(t, out=ring[0:1])                      # warm-up (prepack, workspaces)
clip.sync(); torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
for i, t in enumerate(range(lo, hi)):
    clip.render_u8(t, out=ring[i:i + 1])
clip.sync()
frames = gather_frames(ring, T_FRAMES)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
if rank == 0:
    out = os.environ.get("R3D_CLIP_OUT", "/tmp/clip_u8.raw")          # rgb24: ffmpeg -f rawvideo -pix_fmt rgb24 -s 512x512 -r 25 -i clip_u8.raw ...
    write_frames(frames, out, os.environ.get("R3D_CLIP_FMT", "raw"))
    print("clip: %d frames on %d GPU(s) in %.1f ms = %.1f frames/s (%.1fx real time at 25 fps) -> %s %s" %
          (T_FRAMES, world, dt * 1e3, T_FRAMES / dt, T_FRAMES / dt / 25.0, out, tuple(frames.shape)))
if world > 1:
    dist.destroy_process_group()
