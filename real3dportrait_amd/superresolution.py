"""Drop-in replacements for the reference's super-resolution modules, backed by libr3d_hip.so.

Mirrors (same constructor arguments, state_dict keys/shapes, forward signatures):
  * SynthesisLayer / ToRGBLayer parameter containers   modules/eg3ds/models/networks_stylegan2.py:286-373
  * SynthesisBlock.forward(x, img, ws, **kw) -> (x, img)  networks_stylegan2.py:377-476 (architecture 'skip',
    in_channels != 0, fp32, noise_mode in {'none','const' with zero strength})
  * SynthesisBlockNoUp                                   modules/eg3ds/models/superresolution.py:159-258
  * SuperresolutionHybrid8XDC.forward(rgb, x, ws, **kw) -> rgb   modules/eg3ds/models/superresolution.py:331-359
  * torch.nn.Conv2d / nn.Sequential stacks of the torso model and to_plane_cnn   (Conv2d, ConvStack)

A checkpoint saved from the reference loads with strict=True (keys: block{0,1}.{conv0,conv1,torgb}.
{weight,bias,affine.weight,affine.bias}, conv*.noise_strength, buffers noise_const / resample_filter).

Inference only: inputs are detached and no autograd graph is built (the reference runs this path under torch.no_grad(),
inference/real3d_infer.py:435,479).

fp16 range management (f16x3 precision, see include/r3d_hip.h "fp16 range management"): every activation that travels in
the SPLIT fp16 hi/lo format is stored times an exact power of two derived from a guaranteed bound on max|x|.  The bound
of an fp32 input is measured on the device (r3d_absmax) unless the tensor carries `_r3d_bound` (a device float [N]
tensor); it is propagated through a chain of layers by one r3d_chain_fold launch per forward.  A SPLIT tensor is tagged
with the module it was scaled for (`_r3d_for`): only that module may consume it.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .volumetric_rendering import _FC, _f32c

_SQRT2 = float(np.sqrt(2.0))


def _setup_filter():
    f = torch.tensor([1.0, 3.0, 3.0, 1.0])
    f = torch.outer(f, f)
    return f / f.sum()          # upfirdn2d.setup_filter([1,3,3,1]) (ops/upfirdn2d.py:72-116)


# ---------------------------------------------------------------------------------------------------------------------
# bounds
# ---------------------------------------------------------------------------------------------------------------------
class _BoundMeter:
    """max|x| per sample of an fp32 device tensor (r3d_absmax), two alternating result slots so that no memset launch is
    needed (each call clears the slot of the next one)."""

    def __init__(self):
        self._slots = None
        self._k = 0

    def __call__(self, x):
        lib = _lib.load()
        N = x.shape[0]
        if self._slots is None or self._slots.shape[1] != N or self._slots.device != x.device:
            self._slots = torch.zeros(2, N, device=x.device, dtype=torch.float32)
            self._k = 0
        cur, nxt = self._slots[self._k & 1], self._slots[(self._k + 1) & 1]
        self._k += 1
        _lib.check(lib.r3d_absmax(_lib.ptr(x), x.numel() // N, N, cur.data_ptr(), nxt.data_ptr(), _lib.stream_ptr()), "absmax")
        return cur


_CONST_BOUNDS = {}


def const_bound(value, N, device):
    """A constant bound tensor (cached): for inputs whose range is known by construction, e.g. the renderer's feature image
    (|x| <= 1.002: sigmoid(y) * 1.002 - 0.001 composited with weights summing to <= 1, then * 2 - 1)."""
    key = (float(value), int(N), str(device))
    t = _CONST_BOUNDS.get(key)
    if t is None:
        t = torch.full((N,), float(value), device=device, dtype=torch.float32)
        # The tensor is shared by every stream of the process (PipelinedClipRenderer / StreamPipeline shells read it in r3d_chain_fold on
        # their own streams, ordered only against the caller's stream): finish the fill before anybody can see it.  Once per key.
        torch.cuda.current_stream(t.device).synchronize()
        t._r3d_const = True          # never written again: a fold whose only inputs are such bounds and cached styles can be reused
        _CONST_BOUNDS[key] = t
    return t


# Precision policy (round 5).  A module built by anybody -- a user, patch_model() on a reference model, frames.ClipRenderer,
# bench.py -- computes in 'f16mx': fp32 operands as fp16 hi + lo, hi*hi on the f16 MFMA, the two cross products on the block-scaled 8-bit
# MFMA, fp32 accumulate.  It earned the default when its activation records became OCP e5m2 (r3d_sr_f16x3.hip "f16mx"): every reference
# golden, the benchmarked frame and the heavy-tail sweeps (spikes of 2^6 .. 2^14 sigma, near and far field) sit at 1.3e-5 .. 6.3e-5 of max|ref|
# against the 2e-4 SURVEY 8(d) states; until round 4 (e4m3 records with one exponent per tensor) the far field left the tolerance at 2^14 sigma.
# 'f16x3' (every product as 3 fp16 MFMA terms: <= 1.3e-6, the fp32-class tier) and 'f32' (exact fp32 MFMA) are selected by name;
# R3D_SR_PRECISION overrides the default for a process.  DESIGN 4.2c states the tiers.
DEFAULT_SR_PRECISION = "f16mx"
THROUGHPUT_SR_PRECISION = "f16mx"          # what ClipRenderer(precision='throughput') asks for: the default since round 5
FP32_CLASS_SR_PRECISION = "f16x3"
# A/B switch: 1 = the head network re-folds block1's conv1 operand every frame from max|block0 output| measured in block0's epilogue (rounds 3-4: the
# e4m3 records needed the operand within one layer of a measurement).  Round 5: the e5m2 records have the fp16 hi plane's exponent range, the
# propagated bound of the main fold serves f16mx as it always served f16x3 -- one launch (~10 us on one stream) and one atomic per frame less.
_MX_TAIL_FOLD = os.environ.get("R3D_MX_TAIL_FOLD", "0") == "1"
_MX_UPCONV = os.environ.get("R3D_MX_UPCONV", "1") != "0"      # A/B switch: 0 = f16mx keeps block1's up-sampling conv on the 3-term fp16 split


def set_sr_precision(module, precision):
    """Set `.precision` on every SR block (SynthesisBlock / SynthesisBlockNoUp) and every HIP Conv2d under `module` (the torso / background
    fusion stacks of SuperresolutionHybrid8XDC_Warp; a 1x1 conv or one fed with fp32 has no fp8 path and computes as 'f16x3' under
    'f16mx').  None: leave as constructed.  Returns the module."""
    if precision is not None:
        if precision not in SynthesisBlock._PREC:
            raise ValueError("SR precision must be one of %s, got %r" % (sorted(SynthesisBlock._PREC), precision))
        for m in module.modules():
            if isinstance(m, (SynthesisBlock, Conv2d)):
                m.precision = precision
    return module

MAX_DEPTH = 3      # a stored fp16 operand may be at most this many conv layers away from a measured / known max|x|


# f16mx main loops only within this many layers of a measured bound (ConvStack.forward).  Round 4: 2 -- the e4m3 records carried one exponent
# per tensor and went subnormal three propagated bounds (~15 binades) from a measurement.  Round 5: the e5m2 records have the fp16 hi plane's own
# exponent range, so an MX operand is usable wherever a SPLIT operand is (MAX_DEPTH), including the hand-offs between a stack and a block
# (fuse_fg_bg_convs -> block1: depth 3); measured at depth 3: to_plane_cnn 2.9e-5 (2.2e-5 at depth 2), torso frame 4.0e-5 of max|ref|.
MX_MAX_DEPTH = int(os.environ.get("R3D_MX_MAX_DEPTH", str(MAX_DEPTH)))


def bound_of(x, meter, layers=1):
    """(bound, depth) of an fp32 activation that is about to enter a chain of `layers` conv layers folded together.  The propagated
    bound loosens ~5 binades per layer and the fp16 window has ~16, so a tag (`_r3d_bound`, `_r3d_depth` = layers since the last
    measurement) is only trusted while the deepest operand of the chain stays within MAX_DEPTH layers of a measurement; otherwise
    max|x| is measured on the device (depth 0)."""
    b = getattr(x, "_r3d_bound", None)
    d = int(getattr(x, "_r3d_depth", 0))
    if b is not None and d + layers - 1 <= MAX_DEPTH:
        return b, d
    return meter(x), 0


def _tag(y, bound, depth):
    y._r3d_bound, y._r3d_depth = bound, depth
    return y


def _keep_tags(x):
    """_f32c (detach) drops python attributes: carry the range tags over."""
    y = _f32c(x)
    if y is not x:
        b = getattr(x, "_r3d_bound", None)
        if b is not None:
            _tag(y, b, int(getattr(x, "_r3d_depth", 0)))
        fmt = getattr(x, "_r3d_fmt", None)
        if fmt is not None:
            y._r3d_fmt = fmt
    return y


def chain_fold(ops, N, ext, zero=()):
    """One r3d_chain_fold launch: ops = list of _lib.ChainOp, ext = list of device float[N] bound tensors, zero = absmax slots
    (device float[N]) that kernels launched after this fold will measure into (`_x_absmax` / `_y_absmax`)."""
    lib = _lib.load()
    assert 1 <= len(ops) <= _lib.CHAIN_MAX_OPS and len(ext) <= _lib.CHAIN_MAX_EXT and len(zero) <= _lib.CHAIN_MAX_ZERO
    arr = (_lib.ChainOp * len(ops))(*ops)
    for t in list(ext) + list(zero):
        assert t.is_cuda and t.dtype == torch.float32 and t.numel() >= N and t.is_contiguous()
    ptrs = (ctypes.c_void_p * max(1, len(ext)))(*[t.data_ptr() for t in ext])
    zptrs = (ctypes.c_void_p * max(1, len(zero)))(*[t.data_ptr() for t in zero])
    _lib.check(lib.r3d_chain_fold(ctypes.cast(arr, ctypes.c_void_p), len(ops), N, ctypes.cast(ptrs, ctypes.c_void_p), len(ext),
                                  ctypes.cast(zptrs, ctypes.c_void_p), len(zero), _lib.stream_ptr()), "chain_fold")


class SynthesisLayer(nn.Module):
    """Parameter container with the reference's names/shapes (networks_stylegan2.py:286-320)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True,
                 activation="lrelu", resample_filter=(1, 3, 3, 1), conv_clamp=None, channels_last=False, **_):
        super().__init__()
        assert kernel_size == 3 and activation == "lrelu" and tuple(resample_filter) == (1, 3, 3, 1)
        self.in_channels, self.out_channels, self.w_dim = in_channels, out_channels, w_dim
        self.resolution, self.up, self.use_noise, self.conv_clamp = resolution, up, use_noise, conv_clamp
        self.register_buffer("resample_filter", _setup_filter())
        self.affine = _FC(w_dim, in_channels, bias_init=1.0)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        if use_noise:
            self.register_buffer("noise_const", torch.randn(resolution, resolution))
            self.noise_strength = nn.Parameter(torch.zeros([]))
        self.bias = nn.Parameter(torch.zeros(out_channels))


class ToRGBLayer(nn.Module):
    """networks_stylegan2.py:352-364."""

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        assert kernel_size == 1 and out_channels == 3
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.affine = _FC(w_dim, in_channels, bias_init=1.0)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))


class SynthesisBlock(nn.Module):
    """networks_stylegan2.py:377-476 for the configuration the SR path instantiates: in_channels != 0,
    architecture 'skip', up=2 conv0 + conv1 + toRGB, fp32.

    forward(x, img, ws, ...) -> (x, img) like the reference.  `x` may be NCHW (reference layout) or a
    channel-blocked tensor produced by a previous block (tagged `_r3d_fmt`); the returned x is
    channel-blocked / SPLIT when `self.out_format` says so (used inside SuperresolutionHybrid8XDC) and
    NCHW otherwise (drop-in use, e.g. sr_with_ref.py:83,124)."""

    _UP = 1        # 1: conv0 up=2 + upsample2d RGB skip; 0: SynthesisBlockNoUp

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture="skip",
                 resample_filter=(1, 3, 3, 1), conv_clamp=256, use_fp16=False, fp16_channels_last=False,
                 fused_modconv_default=True, **layer_kwargs):
        super().__init__()
        if in_channels == 0 or architecture != "skip" or img_channels != 3:
            raise NotImplementedError("HIP SynthesisBlock covers the SR configuration only "
                                      "(in_channels != 0, architecture 'skip', 3 image channels)")
        if use_fp16:
            raise NotImplementedError("use_fp16 SR blocks: the Real3D shells run SR in fp32 "
                                      "(img2plane_baseline.py:102); fp16 is not built")
        self.in_channels, self.out_channels, self.w_dim = in_channels, out_channels, w_dim
        self.resolution, self.img_channels, self.is_last, self.architecture = resolution, img_channels, is_last, architecture
        self.use_fp16 = False
        self.fused_modconv_default = fused_modconv_default
        self.register_buffer("resample_filter", _setup_filter())
        self.num_conv, self.num_torgb = 2, 1
        self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2 if self._UP else 1,
                                    conv_clamp=conv_clamp, **layer_kwargs)
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution,
                                    conv_clamp=conv_clamp, **layer_kwargs)
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
        self.conv_clamp = conv_clamp
        self.out_format = "nchw"       # 'nchw' (reference layout) | 'cb8' | 'split' (f16x3 hand-off, needs _next)
        self.return_x = True           # False: skip materialising x (last block of SuperresolutionHybrid8XDC)
        # 'f16mx' (library default): fp32 operands as fp16 hi + lo with the cross products of the block's convs on the block-scaled 8-bit MFMA
        #          (e5m2 activation x e4m3 weight records, error ~2^-15 of each product: <= 6.3e-5 of max|ref| on every reference golden and
        #          heavy-tail sweep, tests/test_gpu_pinned_config.py; 2 instead of 3 matrix passes per MAC);
        # 'f16x3': fp32-accurate 3-term split on the f16 matrix pipe (<= 1.3e-6 over the operand sweeps);
        # 'f32': exact fp32 MFMA.
        # Override per module (`.precision`, set_sr_precision()) or for the process with R3D_SR_PRECISION.
        self.precision = os.environ.get("R3D_SR_PRECISION", DEFAULT_SR_PRECISION)
        self._prepacked = None
        self._prepack_key = None
        self._styles = None
        self._styles_key = None
        self._styles_ws = None         # keeps the ws tensor of the cached styles alive (its address cannot be recycled)
        self._workspace = None
        self._meter = _BoundMeter()
        self._depth_in = 0             # layers between the last measurement and this block's input (set by whoever folds it)
        self._fold_epoch = 0           # bumped by every chain_op(): lets a caller see that nobody re-folded the block since its own fold

    def _buf(self, name, nbytes, dev):
        t = getattr(self, name)
        if t is None or t.numel() < nbytes or t.device != dev:
            t = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            setattr(self, name, t)
        return t

    _FMT = {"none": -1, "nchw": 0, "cb8": 1, "split": 2, "split_mx": 3}      # split_mx: SPLIT with fp8 records in the lo plane (f16mx hand-off between up blocks)
    _PREC = {"f32": 0, "f16x3": 1, "f16mx": 2}

    def _prec(self):
        return self._PREC[self.precision]

    def wants_mx(self):
        """True when a producer of this block's SPLIT input should leave fp8 records in the lo plane (out_format 'split_mx'): the block's
        conv0 then runs the f16mx main loop (the up-sampling conv of SynthesisBlock, the plain 3x3 conv0 of SynthesisBlockNoUp)."""
        return self._prec() == 2 and self.in_channels % 16 == 0 and (_MX_UPCONV or not self._UP)

    def _clamp(self):
        return -1.0 if self.conv_clamp is None else float(self.conv_clamp)

    def prepare(self, ws, dev=None, ws_key=None):
        """Static weight re-layout (cached on parameter versions) + the per-sample style / demodulation vectors for
        `ws` [N,3,w_dim] (cached on the identity + version of `ws_key`, default `ws` itself, and the parameter versions:
        a clip renders every frame with the same ws).  Returns (prepacked, styles) device buffers.  The FOLDED vectors the
        f16x3 kernels read are written by `fold()` / chain_fold, every forward."""
        lib = _lib.load()
        Cin, Cout = self.in_channels, self.out_channels
        st = _lib.stream_ptr()
        c0, c1, tr = self.conv0, self.conv1, self.torgb
        params = (c0.weight, c0.bias, c0.affine.weight, c0.affine.bias, c1.weight, c1.bias, c1.affine.weight, c1.affine.bias,
                  tr.weight, tr.bias, tr.affine.weight, tr.affine.bias)
        keep = [_f32c(t) for t in params]
        dev = keep[0].device if dev is None else dev
        prec = self._prec()
        key = (keep[0].data_ptr(), c0.weight._version, keep[4].data_ptr(), c1.weight._version, str(dev), prec)
        if self._prepack_key != key:
            pre = self._buf("_prepacked", int(lib.r3d_sr_block_prepacked_bytes(Cin, Cout)), dev)
            _lib.check(lib.r3d_sr_block_prepack(Cin, Cout, _lib.ptr(keep[0]), _lib.ptr(keep[4]), _lib.ptr(pre), prec, st),
                       "sr_block_prepack")
            self._prepack_key = key
        src = ws if ws_key is None else ws_key
        N = ws.shape[0]
        skey = (id(src), src._version, tuple(src.shape), N, str(dev)) + tuple((t.data_ptr(), p._version) for t, p in zip(keep, params))
        if self._styles_key != skey or self._styles_ws is not src:
            ws = _f32c(ws)
            assert ws.shape[1] == self.num_conv + self.num_torgb and ws.shape[2] == self.w_dim, ws.shape
            styles = self._buf("_styles", int(lib.r3d_sr_block_styles_bytes(N, Cin, Cout)), dev)
            _lib.check(lib.r3d_sr_block_styles(_lib.ptr(ws), N, self.w_dim, Cin, Cout, *[_lib.ptr(t) for t in keep],
                                               _lib.ptr(styles), st), "sr_block_styles")
            self._styles_key, self._styles_ws = skey, src
        return self._prepacked, self._styles

    def styles_stride(self):
        return int(_lib.load().r3d_sr_block_styles_bytes(1, self.in_channels, self.out_channels)) // 4

    def in_scale(self):
        """(device float view, per-sample stride): the folded conv0 style vector = what a producer of this block's SPLIT
        input multiplies by (start of the styles buffer)."""
        return self._styles.view(torch.float32), self.styles_stride()

    def chain_op(self, src_a=-1, src_b=_lib.CHAIN_SRC_NONE, tail=False):
        """tail=True: re-fold only the conv1 operand from a MEASURED max|block input| (R3D_CHAIN_SR_BLOCK_TAIL)."""
        self._fold_epoch += 1        # whoever asks for the op launches a fold that rewrites this block's folded vectors
        return _lib.ChainOp(kind=_lib.CHAIN_SR_BLOCK_TAIL if tail else _lib.CHAIN_SR_BLOCK, Cin=self.in_channels, Cout=self.out_channels, ksize=3, act=1, gain=_SQRT2,
                            clamp=self._clamp(), src_a=src_a, src_b=src_b, scales=self._styles.data_ptr(), prepacked=None, bias=None)

    def bound_out(self, N):
        """Device view [N] of the bound on this block's output x written by the last fold (valid until the next fold)."""
        lib = _lib.load()
        off = int(lib.r3d_sr_block_bound_offset(self.in_channels, self.out_channels))
        # N > 1: the per-sample bounds sit one styles record apart; consumers (r3d_chain_fold ext bounds) read a dense float[N]
        return self._styles.view(torch.float32).as_strided((N,), (self.styles_stride(),), off).contiguous()

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, noise_mode="random",
                _prepared=None, _next=None, _folded=False, _u8_out=None, _need_img=True, _x_absmax=None, **layer_kwargs):
        lib = _lib.load()
        if noise_mode == "random" or (noise_mode == "const" and
                                      (float(self.conv0.noise_strength) != 0 or float(self.conv1.noise_strength) != 0)):
            raise NotImplementedError("noise_mode=%r: the inference path uses 'none' "
                                      "(img2plane_baseline.py:113)" % noise_mode)
        if img is None:
            raise NotImplementedError("img=None (first block of a synthesis network) is not on the SR path")
        x_fmt = getattr(x, "_r3d_fmt", "nchw")
        if x_fmt in ("split", "split_mx"):
            if getattr(x, "_r3d_for", None) is not self:
                raise RuntimeError("SPLIT activation was scaled for a different consumer")
            _folded = True
            x = x.contiguous()
        else:
            x = _keep_tags(x)
        img = _f32c(img)
        N = img.shape[0]
        Hin, Win = img.shape[-2], img.shape[-1]
        Cin, Cout = self.in_channels, self.out_channels
        dev = img.device
        st = _lib.stream_ptr()
        prec = self._prec()
        pre, styles = _prepared if _prepared is not None else self.prepare(ws, dev)
        if prec >= 1 and not _folded:
            # f16mx wants the conv1 operand within one layer of a measurement: only a measured / known input bound will do
            bx, self._depth_in = bound_of(x, self._meter, layers=2 if prec == 1 else MAX_DEPTH + 1)
            chain_fold([self.chain_op(-1)], N, [bx])
        need = int(lib.r3d_sr_block_workspace_bytes(N, Cin, Cout, Hin, Win))
        work = self._buf("_workspace", need, dev)
        OH, OW = (2 * Hin, 2 * Win) if self._UP else (Hin, Win)
        img_out = torch.empty(N, 3, OH, OW, device=dev, dtype=torch.float32) if (_need_img or _u8_out is None) else None
        out_fmt = self.out_format if self.return_x else "none"
        next_scale, next_stride = None, 0
        if out_fmt == "none":
            x_out = None
        elif out_fmt in ("split", "split_mx"):
            if _next is None:
                raise RuntimeError("out_format=%r needs the consumer (`_next`: its folded in-multiplier)" % out_fmt)
            if out_fmt == "split_mx" and not (prec == 2 and _next.wants_mx()):
                # mixed per-module precision: only an f16mx block writes the fp8 records, and only an f16mx consumer reads them
                out_fmt = "split"
            next_scale, next_stride = _next.in_scale()
            x_out = torch.empty(N, 2, Cout // 8, OH, OW, 8, device=dev, dtype=torch.float16)
        elif out_fmt == "cb8":
            x_out = torch.empty(N, Cout // 8, OH, OW, 8, device=dev, dtype=torch.float32)
        else:
            x_out = torch.empty(N, Cout, OH, OW, device=dev, dtype=torch.float32)
        _lib.check(lib.r3d_sr_block_forward(_lib.ptr(pre), _lib.ptr(styles), N, Cin, Cout, Hin, Win, self._UP,
                                            _lib.ptr(x), self._FMT[x_fmt], _lib.ptr(img), self._clamp(), _lib.ptr(x_out), self._FMT[out_fmt],
                                            _lib.ptr(next_scale), next_stride, _lib.ptr(img_out), _lib.ptr(_u8_out), _lib.ptr(_x_absmax), prec,
                                            _lib.ptr(work), need, st), "sr_block_forward")
        if x_out is not None:
            if out_fmt != "nchw":
                x_out._r3d_fmt = out_fmt
            if out_fmt in ("split", "split_mx"):
                x_out._r3d_for = _next
            elif prec >= 1:
                _tag(x_out, self.bound_out(N), self._depth_in + 2)
        return x_out, img_out


class SynthesisBlockNoUp(SynthesisBlock):
    """modules/eg3ds/models/superresolution.py:159-258 (architecture 'skip', in_channels != 0, fp32): conv0 and conv1
    are both plain modulated 3x3 convs and the RGB skip is `img.add_(torgb(x))` at the block's resolution (the
    upsample2d call is commented out in the reference, :241-243).  Used by SuperresolutionHybrid8XDC_Warp as
    head_torso_block (modules/real3d/super_resolution/sr_with_ref.py:54).  f16x3 precision only."""
    _UP = 0


class Conv2d(nn.Module):
    """torch.nn.Conv2d(in_channels, out_channels, k, 1, padding=k//2) on the HIP conv kernel (r3d_conv_forward), with
    torch's parameter names/shapes so the reference's state_dict loads unchanged.  Covers what the torso / background
    fusion stacks of sr_with_ref.py:24-63 use: k in {1, 3}, stride 1, 'same' zero padding, bias."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        s = stride[0] if isinstance(stride, (tuple, list)) else stride
        pd = padding[0] if isinstance(padding, (tuple, list)) else padding
        if k not in (1, 3) or s != 1 or pd != k // 2:
            raise NotImplementedError("HIP Conv2d covers k in {1,3}, stride 1, padding k//2 (got k=%r s=%r p=%r)" % (k, s, pd))
        if out_channels % 4:
            raise NotImplementedError("HIP Conv2d needs out_channels %% 4 == 0 (got %d)" % out_channels)
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = in_channels, out_channels, (k, k), (1, 1), (pd, pd)
        ref = nn.Conv2d(in_channels, out_channels, k, 1, pd, bias=bias)       # torch's default init
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone()) if bias else None
        self._prepacked = None
        self._prepack_key = None
        self._workspace = None
        self._scales = None
        self._bias32 = None
        self._meter = _BoundMeter()
        self._depth_in = 0
        # 'f16mx' (library default): a 3x3 conv whose SPLIT input carries 8-bit records (R3D_FMT_SPLIT_MX) runs its cross products on the
        # block-scaled 8-bit MFMA; the PRODUCER of the input asks `wants_mx()` and writes the records | 'f16x3' (fp32-class).  'f32' = 'f16x3'
        # here (the plain convs have no exact-f32 kernel).  R3D_SR_PRECISION / set_sr_precision() as for the SR blocks.
        self.precision = os.environ.get("R3D_SR_PRECISION", DEFAULT_SR_PRECISION)

    _buf = SynthesisBlock._buf
    _FMT = SynthesisBlock._FMT

    def wants_mx(self):
        return self.precision == "f16mx" and self.kernel_size[0] == 3 and self.in_channels % 16 == 0

    def prepare(self, N, dev):
        """Static weight re-layout (cached on the parameter version) and this call's scales buffer."""
        lib = _lib.load()
        Cin, Cout, k = self.in_channels, self.out_channels, self.kernel_size[0]
        w = _f32c(self.weight)
        key = (w.data_ptr(), self.weight._version, str(dev))
        if self._prepack_key != key:
            pre = self._buf("_prepacked", int(lib.r3d_conv_prepacked_bytes(Cin, Cout, k)), dev)
            _lib.check(lib.r3d_conv_prepack(_lib.ptr(w), Cin, Cout, k, _lib.ptr(pre), _lib.stream_ptr()), "conv_prepack")
            self._prepack_key = key
        self._bias32 = _f32c(self.bias) if self.bias is not None else None
        self._buf("_scales", int(lib.r3d_conv_scales_bytes(N, Cin, Cout)), dev)

    def in_scale(self):
        N1 = int(_lib.load().r3d_conv_scales_bytes(1, self.in_channels, self.out_channels)) // 4
        return self._scales.view(torch.float32), N1

    def chain_op(self, src_a=-1, src_b=_lib.CHAIN_SRC_NONE, negative_slope=None):
        return _lib.ChainOp(kind=_lib.CHAIN_CONV, Cin=self.in_channels, Cout=self.out_channels, ksize=self.kernel_size[0],
                            act=0 if negative_slope is None else 1, gain=1.0, clamp=-1.0, src_a=src_a, src_b=src_b,
                            scales=self._scales.data_ptr(), prepacked=self._prepacked.data_ptr(),
                            bias=None if self._bias32 is None else self._bias32.data_ptr())

    def bound_out(self, N):
        lib = _lib.load()
        off = int(lib.r3d_conv_scales_bound_offset(self.in_channels, self.out_channels))
        return self._scales.view(torch.float32).as_strided((N,), (self.in_scale()[1],), off).contiguous()     # dense float[N] (a copy for N > 1)

    @staticmethod
    def _shape(x, x_fmt):
        if x_fmt == "nchw":
            return x.shape
        if x_fmt == "cb8":
            return x.shape[0], x.shape[1] * 8, x.shape[2], x.shape[3]
        return x.shape[0], x.shape[2] * 8, x.shape[3], x.shape[4]

    def can_blend(self, a, b):
        """May `cat([a * m, b * (1 - m)])` be fused into this conv (r3d_conv_forward_blend)?  1x1, both sources channel-blocked fp32, Cin % 64 == 0."""
        return (self.kernel_size[0] == 1 and getattr(a, "_r3d_fmt", None) == "cb8" and getattr(b, "_r3d_fmt", None) == "cb8"
                and (a.shape[1] + b.shape[1]) * 8 == self.in_channels and self.in_channels % 64 == 0)

    def forward(self, x, negative_slope=None, out_format="nchw", _next=None, _folded=False, _y_absmax=None, _blend=None):
        """negative_slope: fuse a following torch.nn.LeakyReLU(negative_slope); out_format 'nchw' | 'cb8' | 'split'
        ('split' needs `_next`, the consumer module, folded by the caller in the same chain).
        _blend = (a, b, mask) instead of x: the input is cat([a * mask, b * (1 - mask)], dim=1) (sr_with_ref.py:113-114), computed inside the
        kernel (see can_blend; the caller has folded this layer for max(bound(a), bound(b)))."""
        lib = _lib.load()
        if _blend is not None:
            a, b, mask = _blend
            assert _folded and self.can_blend(a, b), "fused blend: a folded 1x1 conv over two 'cb8' sources"
            a, b, mask = a.contiguous(), b.contiguous(), _f32c(mask)
            N, H, W = a.shape[0], a.shape[2], a.shape[3]
            assert tuple(b.shape[:1] + b.shape[2:4]) == (N, H, W) and tuple(mask.shape) == (N, 1, H, W), (a.shape, b.shape, mask.shape)
            y, next_scale, next_stride = self._alloc_out(N, H, W, out_format, _next, a.device)
            _lib.check(lib.r3d_conv_forward_blend(_lib.ptr(self._prepacked), _lib.ptr(self._scales), _lib.ptr(self._bias32), N, a.shape[1] * 8, b.shape[1] * 8,
                                                  self.out_channels, H, W, _lib.ptr(a), _lib.ptr(b), _lib.ptr(mask),
                                                  0 if negative_slope is None else 1, float(negative_slope or 0.0), 1.0, -1.0,
                                                  _lib.ptr(y), self._FMT[out_format], _lib.ptr(next_scale), next_stride, _lib.ptr(_y_absmax), _lib.stream_ptr()),
                       "conv_forward_blend")
            return self._tag_out(y, out_format, _next, N)
        x_fmt = getattr(x, "_r3d_fmt", "nchw")
        if x_fmt in ("split", "split_mx"):
            if getattr(x, "_r3d_for", None) is not self:
                raise RuntimeError("SPLIT activation was scaled for a different consumer")
            _folded = True
            x = x.contiguous()
        else:
            x = _keep_tags(x)
        Cin, Cout, k = self.in_channels, self.out_channels, self.kernel_size[0]
        N, C, H, W = self._shape(x, x_fmt)
        if C != Cin:
            raise RuntimeError("Conv2d: expected input with %d channels, got %d" % (Cin, C))
        dev = x.device
        st = _lib.stream_ptr()
        if not _folded:
            self.prepare(N, dev)
            bx, self._depth_in = bound_of(x, self._meter, layers=1)
            chain_fold([self.chain_op(-1, negative_slope=negative_slope)], N, [bx])
        need = int(lib.r3d_conv_workspace_bytes(N, Cin, H, W))
        work = self._buf("_workspace", need, dev) if x_fmt not in ("split", "split_mx") else None
        y, next_scale, next_stride = self._alloc_out(N, H, W, out_format, _next, dev)
        act = 0 if negative_slope is None else 1
        _lib.check(lib.r3d_conv_forward(_lib.ptr(self._prepacked), _lib.ptr(self._scales), _lib.ptr(self._bias32), N, Cin, Cout, H, W, k,
                                        _lib.ptr(x), self._FMT[x_fmt], act, float(negative_slope or 0.0), 1.0, -1.0,
                                        _lib.ptr(y), self._FMT[out_format], _lib.ptr(next_scale), next_stride, _lib.ptr(_y_absmax),
                                        _lib.ptr(work), need if work is not None else 0, st), "conv_forward")
        return self._tag_out(y, out_format, _next, N)

    def forward_cat(self, x, dst, chan_off, mask, mask_invert, _next, negative_slope=None):
        """This (already folded) 1x1 conv's output as channels [chan_off, chan_off + out_channels) of the concatenated SPLIT / SPLIT_MX tensor `dst`
        ([N, 2, C_total / 8, H, W, 8] halfs, tagged `_r3d_fmt`), times mask (or 1 - mask) per pixel and the in-multiplier of `_next`, the consumer of
        `dst` (r3d_conv_forward_cat): `x_torso = torso_encoder(hid)` + its half of `cat([x * a, x_torso * (1 - a)])` (sr_with_ref.py:88,104)."""
        lib = _lib.load()
        x_fmt = getattr(x, "_r3d_fmt", "nchw")
        assert x_fmt in ("nchw", "cb8") and self.kernel_size[0] == 1, "forward_cat: a 1x1 conv over an fp32 input"
        x = _f32c(x)
        N, C, H, W = self._shape(x, x_fmt)
        assert C == self.in_channels and tuple(dst.shape[:2]) == (N, 2) and tuple(dst.shape[3:]) == (H, W, 8) and tuple(mask.shape) == (N, 1, H, W)
        need = int(lib.r3d_conv_workspace_bytes(N, C, H, W))
        work = self._buf("_workspace", need, x.device)
        ns, stride = _next.in_scale()
        _lib.check(lib.r3d_conv_forward_cat(_lib.ptr(self._prepacked), _lib.ptr(self._scales), _lib.ptr(self._bias32), N, C, self.out_channels, H, W, 1,
                                            _lib.ptr(x), self._FMT[x_fmt], 0 if negative_slope is None else 1, float(negative_slope or 0.0), 1.0, -1.0,
                                            _lib.ptr(dst), self._FMT[dst._r3d_fmt], dst.shape[2] * 8, chan_off, _lib.ptr(_f32c(mask)), int(bool(mask_invert)),
                                            _lib.ptr(ns), stride, _lib.ptr(work), need, _lib.stream_ptr()), "conv_forward_cat")
        return dst

    def _alloc_out(self, N, H, W, out_format, _next, dev):
        Cout = self.out_channels
        next_scale, next_stride = None, 0
        if out_format in ("split", "split_mx"):      # split_mx: fp8 records in the lo plane, for an f16mx SynthesisBlock that consumes y
            assert _next is not None, "out_format='split' needs the consumer (its folded in-multiplier)"
            next_scale, next_stride = _next.in_scale()
            y = torch.empty(N, 2, Cout // 8, H, W, 8, device=dev, dtype=torch.float16)
        elif out_format == "cb8":
            y = torch.empty(N, Cout // 8, H, W, 8, device=dev, dtype=torch.float32)
        else:
            y = torch.empty(N, Cout, H, W, device=dev, dtype=torch.float32)
        return y, next_scale, next_stride

    def _tag_out(self, y, out_format, _next, N):
        if out_format != "nchw":
            y._r3d_fmt = out_format
        if out_format in ("split", "split_mx"):
            y._r3d_for = _next
        else:
            _tag(y, self.bound_out(N), self._depth_in + 1)
        return y


def upsample2x_bilinear(x, out_format="split", _next=None):
    """torch.nn.UpsamplingBilinear2d(scale_factor=2.) (align_corners=True) on a channel-blocked fp32 activation
    (r3d_upsample2x_bilinear); output 'split' / 'split_mx' (input of the conv `_next`, already folded) or 'cb8'."""
    lib = _lib.load()
    assert getattr(x, "_r3d_fmt", None) == "cb8", "upsample2x_bilinear takes the 'cb8' output of a Conv2d"
    bx, dx = getattr(x, "_r3d_bound", None), int(getattr(x, "_r3d_depth", 0))
    x = x.contiguous()
    N, C8, H, W, _ = x.shape
    next_scale, next_stride = None, 0
    if out_format in ("split", "split_mx"):
        assert _next is not None
        next_scale, next_stride = _next.in_scale()
        y = torch.empty(N, 2, C8, 2 * H, 2 * W, 8, device=x.device, dtype=torch.float16)
    else:
        y = torch.empty(N, C8, 2 * H, 2 * W, 8, device=x.device, dtype=torch.float32)
    _lib.check(lib.r3d_upsample2x_bilinear(_lib.ptr(x), N, C8 * 8, H, W, _lib.ptr(y), SynthesisBlock._FMT[out_format],
                                           _lib.ptr(next_scale), next_stride, _lib.stream_ptr()), "upsample2x_bilinear")
    y._r3d_fmt = out_format
    if out_format in ("split", "split_mx"):
        y._r3d_for = _next
    elif bx is not None:
        _tag(y, bx, dx)            # a convex combination does not raise the bound
    return y


def resize_bilinear(x, size, antialias=True):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=False, antialias=antialias) on the HIP kernel (ATen's separable
    antialiased weights; superresolution.py:351-355, sr_with_ref.py:71-82,110)."""
    lib = _lib.load()
    x = _f32c(x)
    N, C, H, W = x.shape
    OH, OW = size
    y = torch.empty(N, C, OH, OW, device=x.device, dtype=torch.float32)
    _lib.check(lib.r3d_resize_bilinear(_lib.ptr(x), N * C, H, W, _lib.ptr(y), OH, OW, int(bool(antialias)), _lib.stream_ptr()), "resize_bilinear")
    return y


_BLEND_METERS = {}


def blend_cat(a, b, mask, consumer, ws=None, _folded_head=None, _b_channels=None, _dst=None):
    """(b = None with _b_channels = Cb and _dst = the concatenated tensor: only the `a` part is written; Conv2d.forward_cat fills channels [Ca, Ca + Cb).)
    cat([a * mask, b * (1 - mask)], dim=1) (sr_with_ref.py:104,114,126,136) written directly as the SPLIT input of
    `consumer` (a ConvStack / Conv2d / SynthesisBlock[NoUp]; for a block pass its `ws` [N,3,w_dim]): the consumer chain is folded
    here from max(bound(a), bound(b)) (mask in [0,1]), then r3d_blend_cat_to_split multiplies by its in-multiplier.
    a, b: NCHW fp32 or 'cb8'-tagged tensors; mask [N,1,H,W].  _folded_head: the consumer's first module when the caller has
    already folded it (inside a larger chain)."""
    lib = _lib.load()

    def desc(t):
        fmt = getattr(t, "_r3d_fmt", "nchw")
        assert fmt in ("nchw", "cb8"), fmt
        t = _keep_tags(t)
        if fmt == "nchw":
            return t, 0, t.shape[1], t.shape[0], t.shape[2], t.shape[3]
        return t, 1, t.shape[1] * 8, t.shape[0], t.shape[2], t.shape[3]
    a, fa, Ca, N, H, W = desc(a)
    if b is None:
        assert _folded_head is not None and _b_channels
        fb, Cb, Nb, Hb, Wb = 0, int(_b_channels), N, H, W
    else:
        b, fb, Cb, Nb, Hb, Wb = desc(b)
    mask = _f32c(mask)
    assert (N, H, W) == (Nb, Hb, Wb) and tuple(mask.shape) == (N, 1, H, W), (a.shape, b.shape, mask.shape)
    if _folded_head is not None:
        head = _folded_head
    else:
        meters = _BLEND_METERS.setdefault(id(consumer), (_BoundMeter(), _BoundMeter()))
        L = consumer.num_layers()
        (ba, da), (bb, db) = bound_of(a, meters[0], L), bound_of(b, meters[1], L)
        head = consumer.fold_for_input(N, a.device, [ba, bb], ws=ws, depth=max(da, db))
    ns, stride = head.in_scale()
    y = _dst if _dst is not None else torch.empty(N, 2, (Ca + Cb) // 8, H, W, 8, device=a.device, dtype=torch.float16)
    assert tuple(y.shape) == (N, 2, (Ca + Cb) // 8, H, W, 8) and y.is_contiguous()
    fmt = "split_mx" if head.wants_mx() else "split"       # an f16mx consumer: fp8 records in the lo plane
    _lib.check(lib.r3d_blend_cat_to_split(_lib.ptr(a), fa, Ca, _lib.ptr(b), fb, Cb, _lib.ptr(mask), N, H, W, _lib.ptr(y), SynthesisBlock._FMT[fmt],
                                          _lib.ptr(ns), stride, _lib.stream_ptr()), "blend_cat_to_split")
    y._r3d_fmt = fmt
    y._r3d_for = head
    return y


def _fold_single(module, N, dev, bounds, ws=None, negative_slope=None, depth=0):
    """Fold one module (Conv2d or SynthesisBlock) whose input bound is the max of `bounds` (1 or 2 device tensors)."""
    module._depth_in = depth
    if isinstance(module, SynthesisBlock):
        module.prepare(ws, dev)
        op = module.chain_op(-1, -2 if len(bounds) > 1 else _lib.CHAIN_SRC_NONE)
    else:
        module.prepare(N, dev)
        op = module.chain_op(-1, -2 if len(bounds) > 1 else _lib.CHAIN_SRC_NONE, negative_slope=negative_slope)
    chain_fold([op], N, bounds)
    return module


def _block_fold_for_input(self, N, dev, bounds, ws=None, depth=0):
    assert ws is not None, "folding a SynthesisBlock needs its ws"
    return _fold_single(self, N, dev, bounds, ws=ws, depth=depth)


SynthesisBlock.fold_for_input = _block_fold_for_input
SynthesisBlock.num_layers = lambda self: 2


class ConvStack(nn.Sequential):
    """An nn.Sequential of Conv2d / LeakyReLU [/ UpsamplingBilinear2d(2)] modules (the shape of torso_encoder, bg_encoder,
    fuse_head_torso_convs, fuse_fg_bg_convs, sr_with_ref.py:24-63, and of SegFormerSECC2PlaneBackbone.to_plane_cnn,
    modules/real3d/segformer.py:691-700) evaluated on the HIP conv kernel: each LeakyReLU is fused into the
    preceding conv's epilogue and intermediate activations stay in the fp16 hi/lo SPLIT format (no fp32 round trip).
    state_dict keys are the reference's ('0.weight', '0.bias', '2.weight', ...).

    ConvStack.from_torch(seq) converts a torch nn.Sequential with loaded weights."""

    @classmethod
    def from_torch(cls, seq):
        mods = []
        for m in seq:
            if isinstance(m, nn.Conv2d):
                c = Conv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, bias=m.bias is not None)
                c.load_state_dict(m.state_dict())
                mods.append(c.to(m.weight.device))
            elif isinstance(m, nn.LeakyReLU):
                mods.append(nn.LeakyReLU(m.negative_slope))
            elif isinstance(m, nn.UpsamplingBilinear2d) and float(m.scale_factor) == 2.0:
                mods.append(nn.UpsamplingBilinear2d(scale_factor=2.0))
            else:
                raise NotImplementedError("ConvStack: unsupported module %s" % type(m).__name__)
        return cls(*mods)

    def _plan(self):
        """[(conv, negative_slope, upsample_after)] in execution order."""
        mods = list(self)
        plan, i = [], 0
        while i < len(mods):
            m = mods[i]
            if not isinstance(m, Conv2d):
                raise NotImplementedError("ConvStack: %s without a preceding Conv2d" % type(m).__name__)
            slope, step = None, 1
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU):
                slope, step = mods[i + 1].negative_slope, 2
            up = i + step < len(mods) and isinstance(mods[i + step], nn.UpsamplingBilinear2d)
            if up:
                if m.out_channels % 16 or i + step + 1 >= len(mods):
                    raise NotImplementedError("ConvStack: UpsamplingBilinear2d must sit between two convs with C % 16 == 0")
                step += 1
            plan.append((m, slope, up))
            i += step
        return plan

    def num_layers(self):
        return len(self._plan())

    def chain_ops(self, N, dev, src_a=-1, src_b=_lib.CHAIN_SRC_NONE, base=0, depth=0):
        """The stack's layers as r3d_chain_fold ops: the first reads (src_a, src_b), layer k reads op base + k - 1.  Returns
        (ops, first conv, index of the last op).  depth: layers between the last measurement and the stack's input."""
        plan = self._plan()
        ops = []
        for k, (m, slope, _) in enumerate(plan):
            m.prepare(N, dev)
            m._depth_in = depth + k
            ops.append(m.chain_op(src_a, src_b, negative_slope=slope) if k == 0 else m.chain_op(base + k - 1, negative_slope=slope))
        return ops, plan[0][0], base + len(plan) - 1

    def fold_for_input(self, N, dev, bounds, ws=None, depth=0):
        """Fold the whole stack (one launch) for an input whose bound is the max of `bounds`; returns the first conv (the module a
        producer of the stack's SPLIT input scales for)."""
        ops, head, _ = self.chain_ops(N, dev, -1, -2 if len(bounds) > 1 else _lib.CHAIN_SRC_NONE, depth=depth)
        chain_fold(ops, N, bounds)
        return head

    def forward(self, x, out_format="nchw", _next=None, _y_absmax=None, _blend=None):
        """out_format of the LAST conv: 'nchw' (default, reference layout) | 'cb8' | 'split' (scaled for `_next`, already folded);
        _y_absmax: device float[N] slot the last conv measures max|y| into (zeroed by a preceding fold);
        _blend = (a, b, mask) with x = None: the stack's input is cat([a * mask, b * (1 - mask)]) and its first conv computes it (Conv2d.can_blend;
        the caller has folded the stack for the two sources, as for blend_cat(_folded_head=...))."""
        plan = self._plan()
        x_fmt = getattr(x, "_r3d_fmt", "nchw")
        dx = int(getattr(x, "_r3d_depth", 0))
        if _blend is not None:
            assert x is None
            dx = 0            # as for blend_cat's untagged SPLIT output: the caller folded the stack from the sources' bounds
        elif x_fmt in ("split", "split_mx"):
            if getattr(x, "_r3d_for", None) is not plan[0][0]:
                raise RuntimeError("SPLIT activation was scaled for a different consumer")
        else:
            if not hasattr(self, "_meter_obj"):
                object.__setattr__(self, "_meter_obj", _BoundMeter())
            x = _keep_tags(x)
            bx, dx = bound_of(x, self._meter_obj, len(plan))
            self.fold_for_input(x.shape[0], x.device, [bx], depth=dx)
        for k, (m, slope, up) in enumerate(plan):
            nxt = plan[k + 1][0] if k + 1 < len(plan) else None
            # The fp8 records carry ONE exponent per tensor (the fold's bound), and a bound propagated through the layers' L1 norms loosens by
            # ~5 binades per layer: three layers from a measurement the typical operand sits 2^15 under its bound, xh8 = e4m3(hi * 2^-7) is a
            # subnormal and the layer's cross products are noise (to_plane_cnn's last conv at full size: 3.4e-4 of max|ref| instead of 2.8e-5,
            # tests/test_gpu_mx.py) -- so a layer whose operand is more than MX_MAX_DEPTH layers from a measured bound runs f16x3.
            mx_next = nxt is not None and nxt.wants_mx() and m.out_channels % 16 == 0 and dx + k + 1 <= MX_MAX_DEPTH
            bl = _blend if k == 0 else None
            if up:
                x = upsample2x_bilinear(m(x, negative_slope=slope, out_format="cb8", _folded=True, _blend=bl), "split_mx" if mx_next else "split", _next=nxt)
            elif nxt is not None and m.out_channels % 16 == 0:
                x = m(x, negative_slope=slope, out_format="split_mx" if mx_next else "split", _next=nxt, _folded=True, _blend=bl)
            elif nxt is not None:
                raise NotImplementedError("ConvStack: inner layers need out_channels % 16 == 0")
            else:
                x = m(x, negative_slope=slope, out_format=out_format, _next=_next, _folded=True, _y_absmax=_y_absmax, _blend=bl)
        return x


def _conv_fold_for_input(self, N, dev, bounds, ws=None, depth=0):
    return _fold_single(self, N, dev, bounds, depth=depth)


Conv2d.fold_for_input = _conv_fold_for_input
Conv2d.num_layers = lambda self: 1


class SuperresolutionHybrid8XDC(nn.Module):
    """superresolution.py:331-359: 128^2 x 32ch -> 512^2 RGB through two SynthesisBlocks (32->256 @256, 256->128 @512)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, large_sr=False, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        if large_sr:
            raise NotImplementedError("large_sr variants are not on the released inference path")
        use_fp16 = sr_num_fp16_res > 0
        if use_fp16:
            raise NotImplementedError("sr_num_fp16_res > 0: the Real3D shells pass 0 (img2plane_baseline.py:102)")
        self.input_resolution = 128
        self.sr_antialias = sr_antialias
        self.block0 = SynthesisBlock(channels, 256, w_dim=512, resolution=256, img_channels=3, is_last=False,
                                     use_fp16=False, conv_clamp=None, **block_kwargs)
        self.block1 = SynthesisBlock(256, 128, w_dim=512, resolution=512, img_channels=3, is_last=True,
                                     use_fp16=False, conv_clamp=None, **block_kwargs)
        self.block1.return_x = False           # forward() only returns rgb (:359)
        self._meter = _BoundMeter()
        self._ws3 = None                       # (ws, version, ws[:, -1:].repeat(1, 3, 1)): a clip renders every frame with one ws
        self._mx_slot = None                   # f16mx: max|block0 output| measured in its conv1 epilogue
        self._fold_sig, self._fold_bx = None, None

    def _ws_last3(self, ws):
        c = self._ws3
        if c is not None and c[0] is ws and c[1] == ws._version:
            return c[2]
        ws3 = ws[:, -1:, :].repeat(1, 3, 1)          # :349
        self._ws3 = (ws, ws._version, ws3)
        return ws3

    FEATURE_BOUND = 1.01      # |renderer feature image| <= 1.002 by construction (sigmoid * 1.002 - 0.001 composited with weights summing to <= 1)

    def _prepare_and_fold(self, ws, N, dev, bx, dx):
        """Style vectors of both blocks (cached per ws) + the range fold for an input bounded by bx (skipped while nothing it depends on
        changed).  Returns (ws3, prep0, prep1, x_absmax slot or None)."""
        ws3 = self._ws_last3(ws)
        b0, b1 = self.block0, self.block1
        b1.precision = b0.precision
        prep0 = b0.prepare(ws3, dev, ws_key=ws)
        prep1 = b1.prepare(ws3, dev, ws_key=ws)
        mx = b0.precision == "f16mx" and _MX_TAIL_FOLD
        x_absmax = None
        if b0.precision in ("f16x3", "f16mx"):
            # one fold launch for both blocks; block0's conv1 epilogue then emits its output already multiplied by block1.conv0's
            # folded styles and split into fp16 hi/lo planes, so block1 stages its input with plain copies
            b0._depth_in, b1._depth_in = dx, dx + 2
            if mx:      # R3D_MX_TAIL_FOLD=1: block1's conv1 operand within one layer of a measurement: block0 measures max|x0| in its epilogue
                if self._mx_slot is None or self._mx_slot.shape[0] != N or self._mx_slot.device != dev:
                    self._mx_slot = torch.zeros(N, device=dev, dtype=torch.float32)
                x_absmax = self._mx_slot
            # The fold is a function of (bx, both blocks' style vectors).  With a constant bound (the renderer's feature image,
            # const_bound) and the per-clip style cache unchanged, the folded vectors of the previous frame are still in place:
            # skip the launch (13 us per frame).  f16mx: the per-frame tail fold (forward) re-arms the max|x0| slot and only rewrites
            # block1's conv1 operand vectors, which it rewrites again next frame before they are read: the main fold can be skipped too.
            if not getattr(bx, "_r3d_const", False) or self._fold_sig != self._sig(bx, dx):
                chain_fold([b0.chain_op(-1), b1.chain_op(0)], N, [bx], zero=[x_absmax] if mx else ())
                self._fold_sig = self._sig(bx, dx)
                self._fold_bx = bx           # keeps id(bx) from being recycled
        return ws3, prep0, prep1, x_absmax

    def _sig(self, bx, dx):
        b0, b1 = self.block0, self.block1
        return (id(bx), dx, b0._styles_key, b1._styles_key, b0._fold_epoch, b1._fold_epoch, b0.precision)

    def split_input_spec(self, ws, N, dev):
        """For a producer that writes this network's input directly in the SPLIT format (the ray kernel, r3d_render_forward `split_out`):
        makes sure the styles and the fold for the renderer's feature image (|x| <= FEATURE_BOUND) are in place and returns
        (folded input multiplier as a float tensor, per-sample stride in floats, consuming block); None for the exact-f32 precision."""
        if self.block0.precision not in ("f16x3", "f16mx"):
            return None
        self._prepare_and_fold(ws, N, dev, const_bound(self.FEATURE_BOUND, N, dev), 0)
        scale, stride = self.block0.in_scale()
        return scale, stride, self.block0

    def forward(self, rgb, x, ws, _u8_out=None, _need_img=True, **block_kwargs):
        """_u8_out: optional uint8 [N,512,512,3] tensor that receives clamp(-1,1) -> ((x+1)/2*255).int() of the result, fused
        into the last block's toRGB kernel (the conversion real3d_infer.py:472,518-522 does per frame); with
        _need_img=False the fp32 image is not materialised and None is returned.
        x may be the renderer's SPLIT copy of the feature image (produced for `split_input_spec`): no conversion launch then."""
        b0, b1 = self.block0, self.block1
        x_is_split = getattr(x, "_r3d_fmt", None) == "split"
        if x_is_split:
            N, dev = x.shape[0], x.device
            bx, dx = const_bound(self.FEATURE_BOUND, N, dev), 0          # the bound split_input_spec folded for
        else:
            if x.shape[-1] != self.input_resolution:      # :351-355: any other neural-rendering resolution is resampled to 128^2 first
                sz = (self.input_resolution, self.input_resolution)
                x, rgb = resize_bilinear(x, sz, self.sr_antialias), resize_bilinear(rgb, sz, self.sr_antialias)
            N, dev = x.shape[0], x.device
            bx, dx = None, 0
            if b0.precision in ("f16x3", "f16mx"):
                x = _keep_tags(x)
                bx, dx = bound_of(x, self._meter, layers=4)
        ws3, prep0, prep1, x_absmax = self._prepare_and_fold(ws, N, dev, bx, dx)
        mx = x_absmax is not None              # (f16mx with R3D_MX_TAIL_FOLD=1)
        if b0.precision in ("f16x3", "f16mx"):
            # f16mx: block0's conv1 epilogue leaves fp8 records in the lo plane and block1's up-sampling conv runs its cross products on them
            b0.out_format, nxt = ("split_mx" if b1.wants_mx() else "split"), b1
        else:
            b0.out_format, nxt = "cb8", None
        x, rgb = b0(x, rgb, ws3, _prepared=prep0, _next=nxt, _folded=True, _x_absmax=x_absmax, **block_kwargs)
        if mx:
            # tail fold: block1's conv1 operand from the MEASURED max|x0|; clears the slot for the next frame (the kernel reads its
            # bounds before it zeroes) -- our own fold, so the main fold's signature stays valid
            fresh = self._fold_sig == self._sig(bx, dx)
            chain_fold([b1.chain_op(-1, tail=True)], x.shape[0], [x_absmax], zero=[x_absmax])
            if fresh:
                self._fold_sig = self._sig(bx, dx)
        x, rgb = b1(x, rgb, ws3, _prepared=prep1, _folded=True, _u8_out=_u8_out, _need_img=_need_img, **block_kwargs)
        return rgb
