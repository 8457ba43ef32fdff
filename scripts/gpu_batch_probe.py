"""Frames per launch probe (round 5): the head frame's ray kernel + SR with B frames in ONE launch of every kernel, per-frame time on 1 and 3 streams.
Planes are laid out ahead of the timed loop (the layout kernel is per frame either way)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from real3dportrait_amd.frames import clone_generator_shell
from real3dportrait_amd.superresolution import const_bound

dev = torch.device("cuda:0")
G, clip, dec, (cano, residuals, cams) = bench.build_scene(torch, dev, n_frames=128)


def make_worker(Gs, B):
    ren, sr = Gs.renderer, Gs.superresolution
    ren.noise_mode = "hash"; ren.need_depth = False
    ws = torch.ones(B, 14, 512, device=dev)
    planes = torch.cat([cano + residuals[i % len(residuals)] for i in range(B)], 0)
    nhwc = ren.prepare_planes(planes)
    out = torch.empty(B, 512, 512, 3, dtype=torch.uint8, device=dev)

    def frame(t):
        cam = cams[t * B:(t + 1) * B]
        spec = sr.split_input_spec(ws, B, dev)
        feat, depth, wsum, valid = ren.forward_camera(nhwc, Gs.decoder, cam[:, :16].view(-1, 4, 4), cam[:, 16:25].view(-1, 3, 3), 128, Gs.rendering_kwargs, _split_for=spec)
        fimg = feat.permute(0, 2, 1).reshape(B, 32, 128, 128)
        x = feat._r3d_split
        sr(fimg[:, :3].contiguous(), x, ws, noise_mode="none", _u8_out=out, _need_img=False)
        return out
    return frame


for B in (1, 2, 3, 4, 6):
    for ns in (1, 3):
        shells = [G] + [clone_generator_shell(G) for _ in range(ns - 1)]
        workers = [make_worker(g, B) for g in shells]
        streams = [torch.cuda.Stream() for _ in range(ns)]
        n = max(6, 96 // B)

        def run(n):
            for i in range(n):
                with torch.cuda.stream(streams[i % ns]):
                    workers[i % ns](i % (128 // B))
            for s in streams:
                torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:
            run(8)
        best = 1e9
        for rep in range(3):
            t1 = time.perf_counter(); run(n); best = min(best, time.perf_counter() - t1)
        print("B=%d streams=%d: %.4f ms per frame (%.0f frames/s)" % (B, ns, best / (n * B) * 1e3, n * B / best), flush=True)
