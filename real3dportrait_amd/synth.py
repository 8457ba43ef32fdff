"""Deterministic synthetic inputs for tests and benchmarks (no checkpoints / datasets exist offline).

Everything is derived from a counter-based integer hash (splitmix64) evaluated in numpy, so the
same (seed, shape) gives bit-identical arrays in the build container, on the GPU box and on every
rank of a sharded run -- independent of torch / numpy RNG stream versions.

Shapes follow the reference: tri-planes [N,3,32,256,256] (modules/img2plane/img2plane_model.py:72-82),
decoder 32->64->33 (modules/eg3ds/models/triplane.py:166-176), SR parameter names of
SuperresolutionHybrid8XDC (modules/eg3ds/models/superresolution.py:331-346), camera = 16 c2w + 9
intrinsics floats (modules/eg3ds/camera_utils/pose_sampler.py:28-36; focal 4.2647, radius 2.7).
"""
import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_uniform(seed, n, stream=0):
    """n float32 values in [0,1) with 24 random bits each."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        ctr = np.arange(n, dtype=np.uint64) + base
        bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def hash_unitvar(seed, shape, stream=0):
    """Zero-mean unit-variance float32 array (Irwin-Hall sum of 4 uniforms; bounded, bell-shaped)."""
    n = int(np.prod(shape))
    u = hash_uniform(seed, 4 * n, stream).reshape(4, n)
    s = (u[0] + u[1]) + (u[2] + u[3])
    return ((s - np.float32(2.0)) * np.float32(math.sqrt(3.0))).astype(np.float32).reshape(shape)


def synth_noise(seed, shape, stream=7):
    return hash_uniform(seed, int(np.prod(shape)), stream).reshape(shape)


def synth_planes(seed, N=1, C=32, H=256, W=256, scale=1.0):
    return (hash_unitvar(seed, (N, 3, C, H, W), stream=1) * np.float32(scale)).astype(np.float32)


def synth_decoder(seed, C=32, HID=64, OUT=33, sigma_bias=0.0):
    """Raw OSGDecoder parameters (net.0.weight, net.0.bias, net.2.weight, net.2.bias).
    `sigma_bias` shifts the density pre-activation ("dense" variant of SURVEY 8d)."""
    w1 = hash_unitvar(seed, (HID, C), stream=11)
    b1 = hash_unitvar(seed, (HID,), stream=12) * np.float32(0.1)
    w2 = hash_unitvar(seed, (OUT, HID), stream=13)
    b2 = hash_unitvar(seed, (OUT,), stream=14) * np.float32(0.1)
    b2[0] += np.float32(sigma_bias)
    return w1, b1, w2, b2


def synth_sr_block(seed, cin, cout, w_dim=512, stream0=100):
    """conv0/conv1/torgb -> (weight, bias, affine.weight, affine.bias) like SynthesisBlock."""
    def layer(ci, co, k, s):
        return (hash_unitvar(seed, (co, ci, k, k), stream=s),
                hash_unitvar(seed, (co,), stream=s + 1) * np.float32(0.1),
                hash_unitvar(seed, (ci, w_dim), stream=s + 2),
                (np.ones((ci,), np.float32) + hash_unitvar(seed, (ci,), stream=s + 3) * np.float32(0.05)))
    return {"conv0": layer(cin, cout, 3, stream0),
            "conv1": layer(cout, cout, 3, stream0 + 10),
            "torgb": layer(cout, 3, 1, stream0 + 20)}


def synth_sr_params(seed, channels=32, mid=256, last=128, w_dim=512):
    """Parameters of SuperresolutionHybrid8XDC: block0 (channels->mid), block1 (mid->last)."""
    return [synth_sr_block(seed, channels, mid, w_dim, 100), synth_sr_block(seed, mid, last, w_dim, 200)]


# channel plans of the torso / background fusion stacks (modules/real3d/super_resolution/sr_with_ref.py:24-63):
# (in_channels, out_channels, kernel_size, leaky_relu_after)
FUSION_STACKS = {
    "torso_encoder": [(64, 256, 1, False)],
    "bg_encoder": [(3, 64, 3, True), (64, 256, 3, True), (256, 256, 3, False)],
    "fuse_head_torso_convs": [(512, 256, 3, True), (256, 256, 3, False)],
    "fuse_fg_bg_convs": [(512, 64, 1, True), (64, 256, 3, True), (256, 256, 3, False)],
}


# SegFormerSECC2PlaneBackbone.to_plane_cnn (modules/real3d/segformer.py:691-700); "up" = UpsamplingBilinear2d(2) before the conv
TO_PLANE_CNN = [(256, 256, 3, True), (256, 256, 3, True), (256, 256, 3, True), (256, 96, 3, False)]
TO_PLANE_CNN_UP_BEFORE = 3          # the x2 bilinear up-sampling sits in front of layer 3


def synth_conv_stack(seed, plan, stream0=300):
    """[(weight [co,ci,k,k], bias [co])] for a FUSION_STACKS plan; weights ~ N(0, 1/(ci k k)) so activations stay O(1)."""
    out = []
    for i, (ci, co, k, _) in enumerate(plan):
        w = hash_unitvar(seed, (co, ci, k, k), stream=stream0 + 2 * i) * np.float32(1.0 / math.sqrt(ci * k * k))
        b = hash_unitvar(seed, (co,), stream=stream0 + 2 * i + 1) * np.float32(0.1)
        out.append((w.astype(np.float32), b.astype(np.float32)))
    return out


def look_at_camera(yaw=0.0, pitch=0.0, radius=2.7, lookat=(0.0, 0.0, 0.2), focal=4.2647):
    """camera[25] = flattened OpenCV-convention cam2world (4x4) + normalised intrinsics (3x3)."""
    la = np.asarray(lookat, np.float64)
    origin = la + radius * np.array([math.sin(yaw) * math.cos(pitch), math.sin(pitch),
                                     math.cos(yaw) * math.cos(pitch)])
    fwd = la - origin
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 1.0, 0.0])
    right = -np.cross(up, fwd)
    right /= np.linalg.norm(right)
    up2 = np.cross(fwd, right)
    up2 /= np.linalg.norm(up2)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up2, fwd, origin
    K = np.array([[focal, 0, 0.5], [0, focal, 0.5], [0, 0, 1.0]])
    return np.concatenate([c2w.reshape(-1), K.reshape(-1)]).astype(np.float32)


def camera_sweep(n, yaw_lo=-0.4, yaw_hi=0.4, pitch=0.0):
    yaws = np.linspace(yaw_lo, yaw_hi, n) if n > 1 else np.array([0.0])
    return np.stack([look_at_camera(float(y), pitch) for y in yaws]).astype(np.float32)


# ---- host mirror of the device's counter-based sampling noise (csrc/r3d_common.h: mix32 / hash_uniform) ---------------------------
# noise_mode='hash' of ImportanceRenderer replaces the two RNG draws of the reference (torch.rand_like at
# modules/eg3ds/volumetric_rendering/renderer.py:226, torch.rand at :281) by a function of (seed, stream, index): stream 0 = the coarse
# jitter, index = ray * Nc + k; stream 1 = the importance u, index = ray * Nf + j; ray = n * M + m.  tests/test_gpu_pinned_config.py
# holds this mirror bit-identical to the kernel (hash mode == the same arrays injected) and feeds them to the oracle.
def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def device_hash_uniform(seed, stream, idx):
    """float32 [0,1) values of r3d::hash_uniform(seed, stream, idx) for an array of 64-bit indices."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h0 = np.uint32((seed & 0xFFFFFFFF) ^ ((0x9E3779B9 * (int(stream) + 1)) & 0xFFFFFFFF))
        h = _mix32(np.full(1, h0, np.uint32))
        h = _mix32(h ^ np.uint32(seed >> 32))
        h = _mix32(h ^ (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        h = _mix32(h ^ (idx >> np.uint64(32)).astype(np.uint32) ^ np.uint32(0x85EBCA6B))
    return ((h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def render_hash_noise(seed, rays, Nc, Nf):
    """(noise_coarse [len(rays), Nc], u_fine [len(rays), Nf]) the ray kernel derives for the global ray indices `rays` (n * M + m)."""
    rays = np.asarray(rays, dtype=np.uint64).reshape(-1, 1)
    nc = device_hash_uniform(seed, 0, rays * np.uint64(Nc) + np.arange(Nc, dtype=np.uint64)[None, :])
    uf = device_hash_uniform(seed, 1, rays * np.uint64(max(Nf, 1)) + np.arange(max(Nf, 1), dtype=np.uint64)[None, :])[:, :Nf]
    return nc, uf
