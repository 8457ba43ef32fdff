"""Frame-sharded clip rendering: frames of a driving sequence are independent units (the loop body at
inference/real3d_infer.py:480-492 depends only on per-frame inputs and clip constants), so rank r of W renders
the contiguous chunk [r*ceil(T/W), (r+1)*ceil(T/W)) with NO data-path collective, and the uint8 frames are
re-assembled on rank 0 with one gather (RCCL over xGMI when the backend is 'nccl'; each peer has its own
link to the root, so a gather -- not a ring -- is the natural collective here).

The host logic is backend-agnostic (tests run it with gloo on CPU and a fake frame renderer).
"""
import torch
import torch.distributed as dist


def shard_frames(num_frames, world_size, rank):
    """Contiguous chunking: rank r gets [lo, hi); the last ranks may get fewer (or zero) frames."""
    per = (num_frames + world_size - 1) // world_size
    lo = min(rank * per, num_frames)
    hi = min(lo + per, num_frames)
    return lo, hi


def frame_seed(base_seed, frame_index):
    """Sampling-noise seed of a frame: a function of the frame index only, so a sharded run renders the
    same pixels as a serial run (the reference consumes one global generator sequentially)."""
    return (int(base_seed) * 0x9E3779B97F4A7C15 + int(frame_index) * 0xD1B54A32D192ED03 + 1) & 0xFFFFFFFFFFFFFFFF


class ClipGatherer:
    """Re-assembles a frame-sharded clip on rank 0 into ONE pre-allocated buffer [world, per, H, W, 3] (no per-call allocation, no
    concatenation: with contiguous chunks of `per` frames, frame t of the clip is row t of the flattened buffer).

    backend 'r3d'   : r3d_gather_frames behind the C ABI (grouped ncclSend / ncclRecv on the caller's stream, csrc/r3d_comm.hip); the
                      RCCL communicator is created here, its 128-byte id travels over the torch.distributed process group;
    backend 'torch' : torch.distributed.gather into views of the same buffer (any backend; gloo in the CPU tests).
    Without an initialised process group the local ring is the clip."""

    def __init__(self, per, frame_hw=(512, 512), device="cuda", backend="torch", group=None):
        self.per, self.hw, self.group, self.backend = int(per), tuple(frame_hw), group, backend
        self.dist = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist else 1
        self.rank = dist.get_rank(group) if self.dist else 0
        self.root_buf, self._comm, self._lib = None, None, None
        if not self.dist:
            return
        if self.rank == 0:
            self.root_buf = torch.empty(self.world, self.per, self.hw[0], self.hw[1], 3, dtype=torch.uint8, device=device)
        if backend == "r3d":
            import ctypes
            from . import _lib
            self._lib = _lib
            lib = _lib.load()
            uid = (ctypes.c_char * 128)()
            if self.rank == 0:
                _lib.check(lib.r3d_comm_unique_id(ctypes.cast(uid, ctypes.c_void_p)), "comm_unique_id")
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = (ctypes.c_char * 128).from_buffer_copy(box[0])
            comm = ctypes.c_void_p()
            _lib.check(lib.r3d_comm_init(ctypes.cast(uid, ctypes.c_void_p), self.rank, self.world, ctypes.byref(comm)), "comm_init")
            self._comm = comm
        elif backend != "torch":
            raise ValueError("backend must be 'r3d' or 'torch'")

    def gather(self, local, num_frames):
        """local: uint8 [per, H, W, 3] on every rank (tail rows of the last chunks unused).  Returns the clip [num_frames, H, W, 3] (a
        view of the pre-allocated buffer) on rank 0, None elsewhere."""
        assert tuple(local.shape) == (self.per, self.hw[0], self.hw[1], 3) and local.dtype == torch.uint8 and local.is_contiguous()
        if not self.dist:
            return local[:num_frames]
        if self.backend == "r3d":
            lib = self._lib.load()
            self._lib.check(lib.r3d_gather_frames(self._comm, local.data_ptr(), local.numel(),
                                                  None if self.root_buf is None else self.root_buf.data_ptr(), 0, self._lib.stream_ptr()),
                            "gather_frames")
        else:
            dist.gather(local, [self.root_buf[r] for r in range(self.world)] if self.rank == 0 else None, dst=0, group=self.group)
        if self.rank != 0:
            return None
        return self.root_buf.view(self.world * self.per, self.hw[0], self.hw[1], 3)[:num_frames]

    def close(self):
        if self._comm is not None:
            self._lib.load().r3d_comm_destroy(self._comm)
            self._comm = None


_GATHERERS = {}      # (shape, device, group key) -> (weakref to the group or None, ClipGatherer)


def clear_gatherers():
    """Drop the receive buffers gather_frames() keeps between calls (world * per * H * W * 3 bytes of device memory per shape)."""
    for _, g in _GATHERERS.values():
        g.close()
    _GATHERERS.clear()


def gather_frames(local, num_frames, group=None):
    """local: uint8 [per, H, W, 3] on every rank (per = ceil(T/W), tail rows unused).  Returns a NEW uint8 [T, H, W, 3] tensor on rank 0
    and None elsewhere.  The receive buffer is kept per (shape, device, process group) and reused by the next call -- the result is a
    copy of it, so a second call cannot overwrite a clip handed out earlier (use ClipGatherer directly for the zero-copy view);
    clear_gatherers() releases the buffers."""
    import weakref
    if not (dist.is_available() and dist.is_initialized()):
        return local[:num_frames].clone()
    key = (tuple(local.shape), str(local.device), id(group))
    ent = _GATHERERS.get(key)
    if ent is not None and group is not None and ent[0]() is not group:      # id() of a destroyed group was recycled
        ent[1].close()
        ent = None
    if ent is None:
        try:
            ref = weakref.ref(group) if group is not None else (lambda: None)
        except TypeError:                                                   # (a group type without weak references: key on id only)
            ref = (lambda g=group: g)
        ent = _GATHERERS[key] = (ref, ClipGatherer(local.shape[0], local.shape[1:3], local.device, "torch", group))
    out = ent[1].gather(local.contiguous(), num_frames)
    return None if out is None else out.clone()


def render_clip_sharded(render_frame, num_frames, frame_hw=(512, 512), device="cuda", group=None):
    """render_frame(t) -> uint8 [H, W, 3] tensor on `device` for global frame index t.
    Every rank renders its chunk into a device-resident ring, then one gather assembles the clip."""
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    per = (num_frames + world - 1) // world
    lo, hi = shard_frames(num_frames, world, rank)
    ring = torch.zeros(per, frame_hw[0], frame_hw[1], 3, dtype=torch.uint8, device=device)
    for i, t in enumerate(range(lo, hi)):
        ring[i] = render_frame(t)
    return gather_frames(ring, num_frames, group)


def clone_generator_shell(G):
    """A second generator shell that SHARES G's parameters (decoder, SR layers) but owns its own operator objects,
    i.e. its own workspaces / style buffers -- what a second HIP stream needs to render another frame concurrently."""
    from .superresolution import SuperresolutionHybrid8XDC
    from .triplane import TriPlaneGenerator
    dev = next(G.parameters()).device
    G2 = TriPlaneGenerator(hp=G.hparams, backbone=G.backbone)
    G2.decoder = G.decoder
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True)
    for name in ("block0", "block1"):
        src, dst = getattr(G.superresolution, name), getattr(sr, name)
        dst.conv0, dst.conv1, dst.torgb = src.conv0, src.conv1, src.torgb
        dst.precision = src.precision
    G2.superresolution = sr
    G2.rendering_kwargs = dict(G.rendering_kwargs)
    G2.neural_rendering_resolution = G.neural_rendering_resolution
    return G2.to(dev).eval()


class ClipRenderer:
    """Per-frame driver around a (HIP) TriPlaneGenerator: planes_t = cano + residual_t (the reference's
    secc2plane residual add, modules/real3d/secc_img2plane.py:73-81, fused into the layout kernel),
    render + SR, clamp, uint8 HWC conversion on device."""

    def __init__(self, generator, cano_planes, residuals, cameras, ws, base_seed=0, precision=None):
        """precision: SR precision to set on the generator's blocks -- None = leave them as they are (library default 'f16mx'),
        'throughput' = superresolution.THROUGHPUT_SR_PRECISION (the default 'f16mx', unless R3D_SR_PRECISION names another),
        or a precision name ('f16x3' = the fp32-class tier, 'f32')."""
        import os
        from . import _lib
        from .superresolution import THROUGHPUT_SR_PRECISION, set_sr_precision
        self._lib = _lib
        self.G = generator
        if precision == "throughput":
            precision = os.environ.get("R3D_SR_PRECISION", THROUGHPUT_SR_PRECISION)
        set_sr_precision(generator.superresolution, precision)
        self.cano = cano_planes            # [1,3,32,H,W]
        self.residuals = residuals         # list of [1,3,32,H,W] (cycled) or None
        self.cameras = cameras             # [T,25]
        self.ws = ws
        self.base_seed = base_seed
        self.G.renderer.noise_mode = "hash"

    def planes_for(self, t):
        add = self.residuals[t % len(self.residuals)] if self.residuals else None
        nhwc = self.G.renderer.prepare_planes(self.cano, add)
        nhwc._r3d_nhwc = True
        return nhwc

    def _features(self, t):
        """Frame t up to the 128^2 feature image (the SR input) + its known bound (|x| <= 1.002 by construction)."""
        from .superresolution import const_bound
        G = self.G
        G.renderer.seed = frame_seed(self.base_seed, t)
        cam = self.cameras[t: t + 1]
        ren = G.renderer
        R = G.neural_rendering_resolution
        sr = G.superresolution
        # the ray kernel writes the SR's first operand itself (SPLIT copy of the feature image) when the SR takes 128^2 inputs as they come
        from .superresolution import SuperresolutionHybrid8XDC
        spec = sr.split_input_spec(self.ws, 1, cam.device) if (isinstance(sr, SuperresolutionHybrid8XDC) and R == sr.input_resolution) else None
        keep, ren.need_depth = ren.need_depth, False          # only the frames leave this driver: no depth image, no clamp launch
        try:
            feat, depth, wsum, valid = ren.forward_camera(self.planes_for(t), G.decoder, cam[:, :16].view(-1, 4, 4), cam[:, 16:25].view(-1, 3, 3),
                                                          R, G.rendering_kwargs, _split_for=spec)
        finally:
            ren.need_depth = keep                             # (G.synthesis() on the same generator still gets its depth)
        fimg = feat.permute(0, 2, 1).reshape(1, 32, R, R).contiguous()
        fimg._r3d_bound = const_bound(1.01, 1, fimg.device)
        fimg._r3d_split = getattr(feat, "_r3d_split", None)
        return fimg

    def render_image(self, t):
        fimg = self._features(t)
        x = fimg._r3d_split if fimg._r3d_split is not None else fimg
        return self.G.superresolution(fimg[:, :3], x, self.ws, noise_mode="none")

    def render_u8(self, t, out=None):
        """Frame t as uint8 [H,W,3] (into `out` [1,H,W,3] if given).  With the f16x3 SR the clamp -> uint8 conversion of
        real3d_infer.py:472,518-522 is fused into the last block's toRGB kernel (no fp32 image round trip)."""
        lib = self._lib.load()
        fimg = self._features(t)
        sr = self.G.superresolution
        if out is None:
            out = torch.empty(1, 512, 512, 3, dtype=torch.uint8, device=fimg.device)
        x = fimg._r3d_split if fimg._r3d_split is not None else fimg
        if sr.block0.precision in ("f16x3", "f16mx"):
            sr(fimg[:, :3], x, self.ws, noise_mode="none", _u8_out=out, _need_img=False)
        else:
            img = sr(fimg[:, :3], x, self.ws, noise_mode="none").contiguous()
            N, _, H, W = img.shape
            self._lib.check(lib.r3d_frames_to_u8(self._lib.ptr(img), N, H, W, self._lib.ptr(out), self._lib.stream_ptr()),
                            "frames_to_u8")
        return out[0] if out.dim() == 4 else out


class PipelinedClipRenderer:
    """Frames are independent, so consecutive frames are issued round-robin on `n_streams` HIP streams, each with its
    own operator workspaces (clone_generator_shell): the latency-bound ray kernel of frame t+1 overlaps the MFMA-bound
    SR convolutions of frame t.  `sync()` joins the side streams back into the caller's stream."""

    def __init__(self, generator, cano_planes, residuals, cameras, ws, base_seed=0, n_streams=2, precision=None):
        self.streams = [torch.cuda.Stream() for _ in range(n_streams)]
        first = ClipRenderer(generator, cano_planes, residuals, cameras, ws, base_seed, precision)      # sets the precision the shells then copy
        shells = [clone_generator_shell(generator) for _ in range(n_streams - 1)]
        self.clips = [first] + [ClipRenderer(g, cano_planes, residuals, cameras, ws, base_seed, None) for g in shells]
        self._k = 0

    def render_u8(self, t, out=None):
        i = self._k % len(self.streams)
        self._k += 1
        st = self.streams[i]
        cur = torch.cuda.current_stream()
        # every frame: inputs prepared on / ring slots still being read from the caller's stream (a gather, a consumer) are
        # ordered before this frame's kernels (an event wait; cheap)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            res = self.clips[i].render_u8(t, out=out)
        if out is None:
            res.record_stream(cur)                            # allocated in the side stream's pool, consumed by the caller
        return res

    def sync(self):
        cur = torch.cuda.current_stream()
        for st in self.streams:
            cur.wait_stream(st)


class StreamPipeline:
    """Round-robin issue of independent per-frame calls on several HIP streams.  `workers` = one callable per stream, each owning
    its module shells / workspaces (parameters may be shared): worker i renders the frames i, i+n, i+2n, ...  What it buys is the
    fill of kernel tails and of under-filled launches with other frames' work (the torso frame of bench.py: +31 % on 3 streams).
    Results are allocated in the side streams' pools and handed to the caller's stream with record_stream; `sync()` joins the streams."""

    def __init__(self, workers):
        self.workers = list(workers)
        self.streams = [torch.cuda.Stream() for _ in self.workers]
        self._k = 0

    def submit(self, *args, **kwargs):
        i = self._k % len(self.workers)
        self._k += 1
        st, cur = self.streams[i], torch.cuda.current_stream()
        st.wait_stream(cur)                                   # inputs prepared on the caller's stream are ordered before this frame
        with torch.cuda.stream(st):
            res = self.workers[i](*args, **kwargs)
        if torch.is_tensor(res):
            res.record_stream(cur)
        return res

    def sync(self):
        cur = torch.cuda.current_stream()
        for st in self.streams:
            cur.wait_stream(st)


# ---------------------------------------------------------------------------------------------------------------------
# frame writer: the reference hands uint8 HWC frames to imageio / ffmpeg (inference/real3d_infer.py:472-473,
# 520-525); there is no ffmpeg in this image, so the clip is written as raw frames a maintainer can pipe into it.
# ---------------------------------------------------------------------------------------------------------------------
def _png_bytes(frame):
    """Minimal PNG encoder (8-bit RGB, filter 0, one IDAT) on zlib + struct: enough for lossless frame dumps without imageio / PIL."""
    import struct
    import zlib
    h, w, _ = frame.shape
    raw = b"".join(b"\x00" + frame[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_frames(frames, path, fmt="raw"):
    """frames: uint8 [T,H,W,3] (torch tensor on any device, or numpy).  fmt:
      'raw'  one file of packed rgb24 frames (`ffmpeg -f rawvideo -pix_fmt rgb24 -s WxH -r 25 -i clip.raw out.mp4` is the
             imageio.get_writer(..., format='FFMPEG', codec='h264') of real3d_infer.py:522)
      'npy'  numpy array file
      'ppm' / 'png'  one image per frame under the directory `path` (frame_00000.ppm, ...)
    Returns the list of files written."""
    import os
    import numpy as np
    if hasattr(frames, "detach"):
        frames = frames.detach().cpu().numpy()
    frames = np.ascontiguousarray(frames)
    assert frames.dtype == np.uint8 and frames.ndim == 4 and frames.shape[-1] == 3, (frames.dtype, frames.shape)
    T, H, W, _ = frames.shape
    if fmt == "raw":
        frames.tofile(path)
        return [path]
    if fmt == "npy":
        np.save(path, frames)
        return [path if path.endswith(".npy") else path + ".npy"]
    if fmt not in ("ppm", "png"):
        raise ValueError("fmt must be 'raw', 'npy', 'ppm' or 'png'")
    os.makedirs(path, exist_ok=True)
    out = []
    for t in range(T):
        fn = os.path.join(path, "frame_%05d.%s" % (t, fmt))
        with open(fn, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (W, H) + frames[t].tobytes() if fmt == "ppm" else _png_bytes(frames[t]))
        out.append(fn)
    return out
