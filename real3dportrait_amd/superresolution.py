"""Drop-in replacements for the reference's super-resolution modules, backed by libr3d_hip.so.

Mirrors (same constructor arguments, state_dict keys/shapes, forward signatures):
  * SynthesisLayer / ToRGBLayer parameter containers   modules/eg3ds/models/networks_stylegan2.py:286-373
  * SynthesisBlock.forward(x, img, ws, **kw) -> (x, img)  networks_stylegan2.py:377-476 (architecture 'skip',
    in_channels != 0, fp32, noise_mode in {'none','const' with zero strength})
  * SuperresolutionHybrid8XDC.forward(rgb, x, ws, **kw) -> rgb   modules/eg3ds/models/superresolution.py:331-359

A checkpoint saved from the reference loads with strict=True (keys: block{0,1}.{conv0,conv1,torgb}.
{weight,bias,affine.weight,affine.bias}, conv*.noise_strength, buffers noise_const / resample_filter).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .volumetric_rendering import _FC, _f32c


def _setup_filter():
    f = torch.tensor([1.0, 3.0, 3.0, 1.0])
    f = torch.outer(f, f)
    return f / f.sum()          # upfirdn2d.setup_filter([1,3,3,1]) (ops/upfirdn2d.py:72-116)


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
    channel-blocked tensor produced by a previous block (tagged `_r3d_cb8`); the returned x is
    channel-blocked by default when `self.blocked_output` (used inside SuperresolutionHybrid8XDC) and
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
        # 'f16x3': fp32-accurate 3-term fp16 split on the f16 matrix pipe (default, ~5x faster);
        # 'f32': exact fp32 MFMA.  Override per module or with R3D_SR_PRECISION.
        self.precision = os.environ.get("R3D_SR_PRECISION", "f16x3")
        self._prepacked = None
        self._prepack_key = None
        self._styles = None
        self._workspace = None

    def _buf(self, name, nbytes, dev):
        t = getattr(self, name)
        if t is None or t.numel() < nbytes or t.device != dev:
            t = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            setattr(self, name, t)
        return t

    _FMT = {"none": -1, "nchw": 0, "cb8": 1, "split": 2}

    def prepare(self, ws, dev=None):
        """Static weight re-layout (cached on parameter versions) + per-forward style/demodulation vectors for
        `ws` [N,3,w_dim].  Returns (prepacked, styles) device buffers; styles starts with the conv0 style vector
        (what a producer needs to emit this block's input in SPLIT format)."""
        lib = _lib.load()
        ws = _f32c(ws)
        N = ws.shape[0]
        dev = ws.device if dev is None else dev
        Cin, Cout = self.in_channels, self.out_channels
        assert ws.shape[1] == self.num_conv + self.num_torgb and ws.shape[2] == self.w_dim, ws.shape
        st = _lib.stream_ptr()
        c0, c1, tr = self.conv0, self.conv1, self.torgb
        keep = [_f32c(t) for t in (c0.weight, c0.bias, c0.affine.weight, c0.affine.bias,
                                   c1.weight, c1.bias, c1.affine.weight, c1.affine.bias,
                                   tr.weight, tr.bias, tr.affine.weight, tr.affine.bias)]
        prec = {"f32": 0, "f16x3": 1}[self.precision]
        key = (keep[0].data_ptr(), c0.weight._version, keep[4].data_ptr(), c1.weight._version, str(dev), prec)
        if self._prepack_key != key:
            pre = self._buf("_prepacked", int(lib.r3d_sr_block_prepacked_bytes(Cin, Cout)), dev)
            _lib.check(lib.r3d_sr_block_prepack(Cin, Cout, _lib.ptr(keep[0]), _lib.ptr(keep[4]), _lib.ptr(pre), prec, st),
                       "sr_block_prepack")
            self._prepack_key = key
        styles = self._buf("_styles", int(lib.r3d_sr_block_styles_bytes(N, Cin, Cout)), dev)
        _lib.check(lib.r3d_sr_block_styles(_lib.ptr(ws), N, self.w_dim, Cin, Cout, *[_lib.ptr(t) for t in keep],
                                           _lib.ptr(styles), st), "sr_block_styles")
        return self._prepacked, styles

    def styles_stride(self):
        return int(_lib.load().r3d_sr_block_styles_bytes(1, self.in_channels, self.out_channels)) // 4

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, noise_mode="random",
                _prepared=None, _next=None, **layer_kwargs):
        lib = _lib.load()
        if noise_mode == "random" or (noise_mode == "const" and
                                      (float(self.conv0.noise_strength) != 0 or float(self.conv1.noise_strength) != 0)):
            raise NotImplementedError("noise_mode=%r: the inference path uses 'none' "
                                      "(img2plane_baseline.py:113)" % noise_mode)
        if img is None:
            raise NotImplementedError("img=None (first block of a synthesis network) is not on the SR path")
        x_fmt = getattr(x, "_r3d_fmt", "nchw")
        x = x.contiguous() if x_fmt == "split" else _f32c(x)
        img = _f32c(img)
        N = img.shape[0]
        Hin, Win = img.shape[-2], img.shape[-1]
        Cin, Cout = self.in_channels, self.out_channels
        dev = img.device
        st = _lib.stream_ptr()
        prec = {"f32": 0, "f16x3": 1}[self.precision]
        pre, styles = _prepared if _prepared is not None else self.prepare(ws, dev)
        need = int(lib.r3d_sr_block_workspace_bytes(N, Cin, Cout, Hin, Win))
        work = self._buf("_workspace", need, dev)
        OH, OW = (2 * Hin, 2 * Win) if self._UP else (Hin, Win)
        img_out = torch.empty(N, 3, OH, OW, device=dev, dtype=torch.float32)
        out_fmt = self.out_format if self.return_x else "none"
        next_scale, next_stride = None, 0
        if out_fmt == "none":
            x_out = None
        elif out_fmt == "split":
            assert _next is not None, "out_format='split' needs the consumer's styles (scaled hand-off)"
            next_scale, next_stride = _next
            x_out = torch.empty(N, 2, Cout // 8, OH, OW, 8, device=dev, dtype=torch.float16)
        elif out_fmt == "cb8":
            x_out = torch.empty(N, Cout // 8, OH, OW, 8, device=dev, dtype=torch.float32)
        else:
            x_out = torch.empty(N, Cout, OH, OW, device=dev, dtype=torch.float32)
        clamp = -1.0 if self.conv_clamp is None else float(self.conv_clamp)
        _lib.check(lib.r3d_sr_block_forward(_lib.ptr(pre), _lib.ptr(styles), N, Cin, Cout, Hin, Win, self._UP,
                                            _lib.ptr(x), self._FMT[x_fmt], _lib.ptr(img), clamp, _lib.ptr(x_out), self._FMT[out_fmt],
                                            _lib.ptr(next_scale), next_stride, _lib.ptr(img_out), prec,
                                            _lib.ptr(work), need, st), "sr_block_forward")
        if x_out is not None and out_fmt != "nchw":
            x_out._r3d_fmt = out_fmt
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

    _buf = SynthesisBlock._buf
    _FMT = SynthesisBlock._FMT

    def forward(self, x, negative_slope=None, out_format="nchw"):
        """negative_slope: fuse a following torch.nn.LeakyReLU(negative_slope); out_format 'nchw' | 'cb8' | 'split'."""
        lib = _lib.load()
        x_fmt = getattr(x, "_r3d_fmt", "nchw")
        x = x.contiguous() if x_fmt == "split" else _f32c(x)
        Cin, Cout, k = self.in_channels, self.out_channels, self.kernel_size[0]
        if x_fmt == "nchw":
            N, C, H, W = x.shape
        elif x_fmt == "cb8":
            N, C, H, W = x.shape[0], x.shape[1] * 8, x.shape[2], x.shape[3]
        else:
            N, C, H, W = x.shape[0], x.shape[2] * 8, x.shape[3], x.shape[4]
        if C != Cin:
            raise RuntimeError("Conv2d: expected input with %d channels, got %d" % (Cin, C))
        dev = x.device
        st = _lib.stream_ptr()
        w = _f32c(self.weight)
        key = (w.data_ptr(), self.weight._version, str(dev))
        if self._prepack_key != key:
            pre = self._buf("_prepacked", int(lib.r3d_conv_prepacked_bytes(Cin, Cout, k)), dev)
            _lib.check(lib.r3d_conv_prepack(_lib.ptr(w), Cin, Cout, k, _lib.ptr(pre), st), "conv_prepack")
            self._prepack_key = key
        need = int(lib.r3d_conv_workspace_bytes(N, Cin, H, W))
        work = self._buf("_workspace", need, dev) if x_fmt != "split" else None
        if out_format == "split":
            y = torch.empty(N, 2, Cout // 8, H, W, 8, device=dev, dtype=torch.float16)
        elif out_format == "cb8":
            y = torch.empty(N, Cout // 8, H, W, 8, device=dev, dtype=torch.float32)
        else:
            y = torch.empty(N, Cout, H, W, device=dev, dtype=torch.float32)
        b = _f32c(self.bias) if self.bias is not None else None
        act = 0 if negative_slope is None else 1
        _lib.check(lib.r3d_conv_forward(_lib.ptr(self._prepacked), N, Cin, Cout, H, W, k, _lib.ptr(x), self._FMT[x_fmt],
                                        None, 0, None, 0, _lib.ptr(b), 0, act, float(negative_slope or 0.0), 1.0, -1.0,
                                        _lib.ptr(y), self._FMT[out_format], None, 0, _lib.ptr(work), need if work is not None else 0, st),
                   "conv_forward")
        if out_format != "nchw":
            y._r3d_fmt = out_format
        return y


def upsample2x_bilinear(x, out_format="split"):
    """torch.nn.UpsamplingBilinear2d(scale_factor=2.) (align_corners=True) on a channel-blocked fp32 activation
    (r3d_upsample2x_bilinear); output 'split' (input of the next conv) or 'cb8'."""
    lib = _lib.load()
    assert getattr(x, "_r3d_fmt", None) == "cb8", "upsample2x_bilinear takes the 'cb8' output of a Conv2d"
    x = x.contiguous()
    N, C8, H, W, _ = x.shape
    if out_format == "split":
        y = torch.empty(N, 2, C8, 2 * H, 2 * W, 8, device=x.device, dtype=torch.float16)
    else:
        y = torch.empty(N, C8, 2 * H, 2 * W, 8, device=x.device, dtype=torch.float32)
    _lib.check(lib.r3d_upsample2x_bilinear(_lib.ptr(x), N, C8 * 8, H, W, _lib.ptr(y), SynthesisBlock._FMT[out_format], None, 0,
                                           _lib.stream_ptr()), "upsample2x_bilinear")
    y._r3d_fmt = out_format
    return y


def blend_cat(a, b, mask):
    """cat([a * mask, b * (1 - mask)], dim=1) (sr_with_ref.py:104,114,126,136) written directly as the SPLIT input of the
    next Conv2d / ConvStack (r3d_blend_cat_to_split).  a, b: NCHW fp32 or 'cb8'-tagged tensors; mask [N,1,H,W]."""
    lib = _lib.load()

    def desc(t):
        fmt = getattr(t, "_r3d_fmt", "nchw")
        assert fmt in ("nchw", "cb8"), fmt
        t = _f32c(t)
        if fmt == "nchw":
            return t, 0, t.shape[1], t.shape[0], t.shape[2], t.shape[3]
        return t, 1, t.shape[1] * 8, t.shape[0], t.shape[2], t.shape[3]
    a, fa, Ca, N, H, W = desc(a)
    b, fb, Cb, Nb, Hb, Wb = desc(b)
    mask = _f32c(mask)
    assert (N, H, W) == (Nb, Hb, Wb) and tuple(mask.shape) == (N, 1, H, W), (a.shape, b.shape, mask.shape)
    y = torch.empty(N, 2, (Ca + Cb) // 8, H, W, 8, device=a.device, dtype=torch.float16)
    _lib.check(lib.r3d_blend_cat_to_split(_lib.ptr(a), fa, Ca, _lib.ptr(b), fb, Cb, _lib.ptr(mask), N, H, W, _lib.ptr(y),
                                          _lib.stream_ptr()), "blend_cat_to_split")
    y._r3d_fmt = "split"
    return y


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

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if not isinstance(m, Conv2d):
                raise NotImplementedError("ConvStack: %s without a preceding Conv2d" % type(m).__name__)
            slope, step = None, 1
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU):
                slope, step = mods[i + 1].negative_slope, 2
            nxt = mods[i + step] if i + step < len(mods) else None
            if isinstance(nxt, nn.UpsamplingBilinear2d):
                if m.out_channels % 16 or i + step + 1 >= len(mods):
                    raise NotImplementedError("ConvStack: UpsamplingBilinear2d must sit between two convs with C % 16 == 0")
                x = upsample2x_bilinear(m(x, negative_slope=slope, out_format="cb8"), "split")
                step += 1
            else:
                fmt = "split" if (nxt is not None and m.out_channels % 16 == 0) else "nchw"
                x = m(x, negative_slope=slope, out_format=fmt)
            i += step
        return x


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

    def forward(self, rgb, x, ws, **block_kwargs):
        ws = ws[:, -1:, :].repeat(1, 3, 1)
        if x.shape[-1] != self.input_resolution:      # cold path, same ATen op as the reference (:351-355)
            x = F.interpolate(x, size=(self.input_resolution, self.input_resolution), mode="bilinear",
                              align_corners=False, antialias=self.sr_antialias)
            rgb = F.interpolate(rgb, size=(self.input_resolution, self.input_resolution), mode="bilinear",
                                align_corners=False, antialias=self.sr_antialias)
        # block1's style vectors first: block0's conv1 epilogue emits its output already scaled by block1.conv0's
        # styles and split into fp16 hi/lo planes (f16x3), so block1 stages its input with plain copies
        self.block1.precision = self.block0.precision
        prep1 = self.block1.prepare(ws)
        if self.block0.precision == "f16x3":
            self.block0.out_format = "split"
            nxt = (prep1[1].view(torch.float32), self.block1.styles_stride())
        else:
            self.block0.out_format, nxt = "cb8", None
        x, rgb = self.block0(x, rgb, ws, _next=nxt, **block_kwargs)
        x, rgb = self.block1(x, rgb, ws, _prepared=prep1, **block_kwargs)
        return rgb
