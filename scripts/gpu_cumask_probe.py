"""CU-masked streams (hipExtStreamCreateWithCUMask): the ray kernel of frame t+1 on one set of CUs, the SR of frame t on the rest.
Frames/s for several splits, next to unmasked 2-stream and the product's 3-stream pipelining.  VERDICT r01 item 6."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from real3dportrait_amd.frames import PipelinedClipRenderer, ClipRenderer, clone_generator_shell
dev = torch.device("cuda", 0)
G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=64)
cano, residuals, cams = scene
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ring = torch.zeros(K, 512, 512, 3, dtype=torch.uint8, device=dev)
hip = ctypes.CDLL("libamdhip64.so")

def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[32 * w + b]) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)

def run_split(sR, sS, label, reps=3):
    """render part of every frame on sR, SR part on sS (one generator shell: each side's buffers are touched by one stream only)"""
    c = ClipRenderer(clone_generator_shell(G), cano, residuals, cams, clip.ws, base_seed=clip.base_seed)
    sr = c.G.superresolution
    best = 0.0
    for rep in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(K):
            with torch.cuda.stream(sR):
                fimg = c._features(t)
                ev = torch.cuda.Event(); ev.record(sR)
            sS.wait_event(ev)
            with torch.cuda.stream(sS):
                fimg.record_stream(sS)
                sr(fimg[:, :3], fimg, c.ws, noise_mode="none", _u8_out=ring[t:t + 1], _need_img=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep: best = max(best, K / dt)
    print("%-58s %8.1f frames/s" % (label, best)); sys.stdout.flush()
    return best

def run_product(n):
    pipe = PipelinedClipRenderer(G, cano, residuals, cams, clip.ws, base_seed=clip.base_seed, n_streams=n)
    best = 0.0
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(K): pipe.render_u8(t, out=ring[t:t + 1])
        pipe.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep: best = max(best, K / dt)
    print("%-58s %8.1f frames/s" % ("product: %d streams, no masks" % n, best)); sys.stdout.flush()

run_product(1); run_product(3)
run_split(torch.cuda.Stream(), torch.cuda.Stream(), "render | SR on two plain streams")
for nr in (48, 64, 80, 96):
    # contiguous: the first nr mask bits to the ray kernel
    bits = [i < nr for i in range(256)]
    run_split(masked_stream(bits), masked_stream([not b for b in bits]), "contiguous bits: render %d CUs | SR %d CUs" % (nr, 256 - nr))
    # interleaved: every (256/nr)-th bit
    step = 256.0 / nr
    chosen = set(int(i * step) for i in range(nr))
    bits = [i in chosen for i in range(256)]
    run_split(masked_stream(bits), masked_stream([not b for b in bits]), "interleaved bits: render %d CUs | SR %d CUs" % (nr, 256 - nr))
