"""Drop-in replacements for the reference's volume-rendering operators, backed by libr3d_hip.so.

Mirrors (same class names, call signatures, option keys, return values):
  * RaySampler.forward(cam2world_matrix, intrinsics, resolution)
        modules/eg3ds/volumetric_rendering/ray_sampler.py:24-63
  * ImportanceRenderer.forward(planes, decoder, ray_origins, ray_directions, rendering_options)
        modules/eg3ds/volumetric_rendering/renderer.py:118-167
  * ImportanceRenderer.run_model(planes, decoder, sample_coordinates, sample_directions, options)
        renderer.py:169-188
  * OSGDecoder (parameter container with the checkpoint keys net.0.weight/bias, net.2.weight/bias)
        modules/eg3ds/models/triplane.py:166-189, modules/img2plane/triplane.py:122-146

There is no eager/PyTorch fallback: tensors must live on the GPU and the HIP library must be built.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


class RaySampler(nn.Module):
    """ray_sampler.py:18-63.  (N,4,4) cam2world + (N,3,3) normalised intrinsics -> origins, dirs [N,R*R,3]."""

    def __init__(self):
        super().__init__()

    def forward(self, cam2world_matrix, intrinsics, resolution):
        lib = _lib.load()
        c2w, K = _f32c(cam2world_matrix), _f32c(intrinsics)
        N, R = c2w.shape[0], int(resolution)
        origins = torch.empty(N, R * R, 3, device=c2w.device, dtype=torch.float32)
        dirs = torch.empty_like(origins)
        _lib.check(lib.r3d_raygen(_lib.ptr(c2w), _lib.ptr(K), N, R, _lib.ptr(origins), _lib.ptr(dirs),
                                  _lib.stream_ptr()), "raygen")
        return origins, dirs


class _FC(nn.Module):
    """FullyConnectedLayer parameter layout (networks_stylegan2.py:99-131), linear activation only."""

    def __init__(self, in_features, out_features, lr_multiplier=1.0, bias_init=0.0):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.randn(out_features, in_features) / lr_multiplier)
        self.bias = nn.Parameter(torch.full([out_features], float(bias_init)))
        self.weight_gain = lr_multiplier / (in_features ** 0.5)
        self.bias_gain = lr_multiplier

    def forward(self, x):   # utility only; the renderer never calls this, it reads the parameters
        return torch.addmm((self.bias * self.bias_gain).unsqueeze(0), x, (self.weight * self.weight_gain).t())


class OSGDecoder(nn.Module):
    """Same constructor / state_dict keys as the reference decoder (triplane.py:166-176).

    The HIP renderer evaluates this MLP inside the fused ray kernel from the raw parameters; `forward`
    is kept (plain torch) only for callers that decode pre-sampled features outside the renderer."""

    def __init__(self, n_features=32, options=None):
        super().__init__()
        options = options or {"decoder_lr_mul": 1, "decoder_output_dim": 32}
        self.hidden_dim = 64
        self.net = nn.Sequential(
            _FC(n_features, self.hidden_dim, lr_multiplier=options["decoder_lr_mul"]),
            nn.Softplus(),
            _FC(self.hidden_dim, 1 + options["decoder_output_dim"], lr_multiplier=options["decoder_lr_mul"]))

    def forward(self, sampled_features, ray_directions=None):
        x = sampled_features.mean(1)
        N, M, C = x.shape
        x = self.net(x.reshape(N * M, C)).view(N, M, -1)
        return {"rgb": torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001, "sigma": x[..., 0:1]}


def decoder_params(decoder):
    """Raw (w1, b1, w2, b2) of any module shaped like OSGDecoder (ours or the reference's)."""
    l0, l2 = decoder.net[0], decoder.net[2]
    for l in (l0, l2):
        if abs(float(getattr(l, "bias_gain", 1.0)) - 1.0) > 1e-12 or \
           abs(float(l.weight_gain) * (l.weight.shape[1] ** 0.5) - 1.0) > 1e-6:
            raise NotImplementedError("decoder_lr_mul != 1 is not supported by the fused HIP decoder")
    w1, b1, w2, b2 = _f32c(l0.weight), _f32c(l0.bias), _f32c(l2.weight), _f32c(l2.bias)
    if tuple(w1.shape) != (64, 32) or tuple(w2.shape) != (33, 64):
        raise NotImplementedError("fused HIP decoder is built for 32 -> 64 -> 33, got %s, %s"
                                  % (tuple(w1.shape), tuple(w2.shape)))
    return w1, b1, w2, b2


class ImportanceRenderer(nn.Module):
    """renderer.py:107-297 restated as one fused HIP kernel per frame (see csrc/r3d_render.hip).

    rendering_options keys read (same as the reference): ray_start, ray_end (must both be 'auto': the
    numeric branch is broken upstream), box_warp, depth_resolution,
    depth_resolution_importance, disparity_space_sampling (must be False), clamp_mode ('softplus'),
    white_back, density_noise (must be 0 / absent at inference).

    Sampling noise: by default the two draws of the reference (torch.rand_like at renderer.py:226 and
    torch.rand at renderer.py:281) are drawn here with the same shapes, in the same order, from torch's
    generator, so a seeded run consumes the RNG exactly like the reference does.  Pass explicit tensors
    through `self.noise_override = (noise_coarse, u_fine)` (parity tests), or set
    `self.noise_mode = 'hash'` to let the kernel derive the noise from (`self.seed`, ray, sample) -- the
    mode the frame-sharded driver uses, because it makes a frame independent of its rank."""

    def __init__(self, hp=None):
        super().__init__()
        self.hparams = dict(hp) if hp is not None else {}
        self.triplane_feature_type = self.hparams.get("triplane_feature_type", "triplane")
        # 'trigrid' / 'trigrid_v2': planes are [N,3,C*D,H,W] volumes sampled tri-linearly (renderer.py:78-89,180-181);
        # the reference reads the depth from hparams['triplane_depth'] (default 1)
        self.triplane_depth = int(self.hparams.get("triplane_depth", 1)) if self.triplane_feature_type in ("trigrid", "trigrid_v2") else 1
        self.noise_mode = "torch"
        self.noise_override = None
        self.seed = 0
        self._plane_cache = None       # (source tensor, version, nhwc tensor)
        self.rgb_channel_major = True  # memory layout of the colours (the returned tensor is [N,M,32] either way)
        self.need_depth = True         # False: forward returns depth = None and skips the depth-clamp launch (ClipRenderer: frames only)
        self._workspace = None

    # -- plane layout ---------------------------------------------------------------------------------
    SECC_PLANE_FLIPS = 53      # R3D_SECC_PLANE_FLIPS: planes 0,1 flipped along H, plane 2 along H and W (segformer.py:722-728)

    def prepare_planes(self, planes, add=None, add_flip=0):
        """NCHW [N,3,C,H,W] (+ optional per-frame residual, secc_img2plane.py:76-77) -> channel-last.
        add_flip = SECC_PLANE_FLIPS when `add` is the raw to_plane_cnn output ([N,96,H,W], before the torch.flip calls of
        SegFormerSECC2PlaneBackbone.forward): the flips are applied while the residual is read."""
        lib = _lib.load()
        planes = _f32c(planes)
        N, P, CD, H, W = planes.shape
        D = self.triplane_depth
        assert P == 3, "expected tri-planes [N,3,C,H,W]"
        if CD != 32 * D:
            raise NotImplementedError("HIP renderer is built for 32 feature channels (x triplane_depth %d), got %d" % (D, CD))
        C = CD // D
        # one allocation: the channel-last planes, then the per-block |max| partials of the layout pass (the bound the renderer's fp16
        # range fold needs, r3d_hip.h "plane_absmax") -- the tensor handed back is the view of the first part
        numel = N * 3 * D * H * W * C
        npart = int(lib.r3d_planes_absmax_partials(N, C, H, W, D))
        buf = torch.empty(numel + npart, device=planes.device, dtype=torch.float32)
        out = buf[:numel].view((N, 3, H, W, C) if D == 1 else (N, 3, D, H, W, C))
        part = buf[numel:]
        addc = _f32c(add).reshape(planes.shape) if add is not None else None
        nwritten = ctypes.c_int(0)
        _lib.check(lib.r3d_planes_to_nhwc(_lib.ptr(planes), _lib.ptr(addc), _lib.ptr(out), N, C, H, W, D,
                                          int(add_flip), part.data_ptr(), ctypes.byref(nwritten), _lib.stream_ptr()), "planes_to_nhwc")
        out._r3d_nhwc = True
        out._r3d_absmax = (part, int(nwritten.value), out._version)
        return out

    @staticmethod
    def _absmax_of(planes_nhwc):
        """(partials pointer, count) for r3d_render_forward / r3d_run_model, or (None, 0) = "measure the bound inside the call".  The
        partials of prepare_planes are only trusted while the tensor is the one that pass wrote: an in-place update afterwards bumps
        its version, and a stale bound would let the kernel's unguarded fp16 split (split8_bounded) overflow silently."""
        tag = getattr(planes_nhwc, "_r3d_absmax", None)
        if tag is None or tag[2] != planes_nhwc._version:
            return None, 0
        return tag[0].data_ptr(), tag[1]

    def _planes_nhwc(self, planes):
        """Channel-last copy of `planes`, cached for static planes.  The cache entry holds the SOURCE tensor itself and is valid
        only for that very object at the same version: a freed tensor's address (and a fresh tensor's version 0) can be handed
        to the next frame's planes by the caching allocator, so an address/version key alone would render stale planes."""
        if getattr(planes, "_r3d_nhwc", False):
            return planes
        c = self._plane_cache
        if c is not None and c[0] is planes and c[1] == planes._version:
            return c[2]
        out = self.prepare_planes(planes)
        self._plane_cache = (planes, planes._version, out)
        return out

    def _check_options(self, opts):
        if self.triplane_feature_type not in ("triplane", "trigrid", "trigrid_v2"):
            raise NotImplementedError("triplane_feature_type=%r (3dgrid) is not built" % self.triplane_feature_type)
        if not (opts.get("ray_start") == "auto" and opts.get("ray_end") == "auto"):
            raise NotImplementedError("only ray_start == ray_end == 'auto' is supported "
                                      "(the numeric branch raises UnboundLocalError upstream)")
        if opts.get("disparity_space_sampling", False):
            raise NotImplementedError("disparity_space_sampling")
        if opts.get("clamp_mode", "softplus") != "softplus":
            raise AssertionError("MipRayMarcher only supports `clamp_mode`=`softplus`!")
        if opts.get("density_noise", 0) > 0:
            raise NotImplementedError("density_noise is a training-only branch")

    # -- forward ----------------------------------------------------------------------------------------
    def forward_camera(self, planes, decoder, cam2world_matrix, intrinsics, resolution, rendering_options, _split_for=None):
        """RaySampler.forward + forward in one call: the rays are generated inside the kernels (r3d_render_forward's camera mode) with the
        instruction sequence of r3d_raygen -- identical pixels, one launch and two [N,M,3] arrays less.  Same return value as forward.
        _split_for = (scale float tensor, per-sample stride in floats, consumer module): the colours are ALSO written as the consumer's
        SPLIT operand (fp16 hi / lo, times its folded input multiplier); the tensor comes back as `rgb._r3d_split`."""
        return self.forward(planes, decoder, None, None, rendering_options, _camera=(_f32c(cam2world_matrix), _f32c(intrinsics), int(resolution)),
                            _split_for=_split_for)

    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options, _camera=None, _split_for=None):
        lib = _lib.load()
        self._check_options(rendering_options)
        planes_nhwc = self._planes_nhwc(planes)
        N, H, W = planes_nhwc.shape[0], planes_nhwc.shape[-3], planes_nhwc.shape[-2]
        D = self.triplane_depth
        assert planes_nhwc.dim() == (5 if D == 1 else 6) and (D == 1 or planes_nhwc.shape[2] == D), planes_nhwc.shape
        if _camera is not None:
            c2w, K, R = _camera
            assert c2w.shape[0] == N and tuple(c2w.shape[1:]) == (4, 4) and tuple(K.shape) == (N, 3, 3), (c2w.shape, K.shape)
            o = d = None
            M = R * R
        else:
            c2w = K = None
            o, d = _f32c(ray_origins), _f32c(ray_directions)
            M = o.shape[1]
        Nc = int(rendering_options["depth_resolution"])
        Nf = int(rendering_options["depth_resolution_importance"])
        w1, b1, w2, b2 = decoder_params(decoder)
        dev = planes_nhwc.device

        noise_c = u_f = None
        if self.noise_override is not None:
            noise_c, u_f = self.noise_override
            noise_c = _f32c(noise_c).reshape(N, M, Nc)
            u_f = _f32c(u_f).reshape(N * M, Nf) if Nf > 0 else None
        elif self.noise_mode == "torch":
            noise_c = torch.rand(N, M, Nc, 1, device=dev, dtype=torch.float32)     # == rand_like(depths_coarse)
            u_f = torch.rand(N * M, Nf, device=dev, dtype=torch.float32) if Nf > 0 else None
        elif self.noise_mode != "hash":
            raise ValueError("noise_mode must be 'torch' or 'hash'")

        # the kernel writes the colours channel-major ([N,32,M] = the NCHW feature image synthesis() is about to build); the tensor
        # handed back is the [N,M,32] VIEW of it, so the reference's permute(0,2,1).reshape(N,32,R,R).contiguous() is free
        if self.rgb_channel_major:
            rgb_cm = torch.empty(N, 32, M, device=dev, dtype=torch.float32)
            rgb = rgb_cm.permute(0, 2, 1)
        else:
            rgb_cm = rgb = torch.empty(N, M, 32, device=dev, dtype=torch.float32)
        depth = torch.empty(N, M, 1, device=dev, dtype=torch.float32) if self.need_depth else None
        wsum = torch.empty(N, M, 1, device=dev, dtype=torch.float32)
        valid = torch.empty(N, M, 1, device=dev, dtype=torch.bool)
        need = int(lib.r3d_render_workspace_bytes(N, M, Nc, Nf))
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != dev:
            self._workspace = torch.empty(need, device=dev, dtype=torch.uint8)
        part, npart = self._absmax_of(planes_nhwc)        # None (caller's own layout, or updated in place since): measured inside the call
        x_split, sp_scale, sp_stride = None, None, 0
        if _split_for is not None:
            sp_scale, sp_stride, consumer = _split_for
            R = int(round(M ** 0.5))
            assert R * R == M, "the SPLIT copy is an image: M must be a square"
            x_split = torch.empty(N, 2, 4, R, R, 8, device=dev, dtype=torch.float16)
        _lib.check(lib.r3d_render_forward(
            _lib.ptr(planes_nhwc), N, H, W, D, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2),
            _lib.ptr(o), _lib.ptr(d), M, Nc, Nf, float(rendering_options["box_warp"]),
            int(bool(rendering_options.get("white_back", False))),
            _lib.ptr(noise_c), _lib.ptr(u_f), int(self.seed) & 0xFFFFFFFFFFFFFFFF,
            _lib.ptr(rgb_cm), int(self.rgb_channel_major), _lib.ptr(depth), _lib.ptr(wsum), _lib.ptr(valid),
            part, npart, _lib.ptr(c2w), _lib.ptr(K),
            _lib.ptr(x_split), None if sp_scale is None else sp_scale.data_ptr(), int(sp_stride),
            _lib.ptr(self._workspace), need, _lib.stream_ptr()), "render_forward")
        if x_split is not None:
            x_split._r3d_fmt, x_split._r3d_for = "split", consumer
            rgb._r3d_split = x_split
        return rgb, depth, wsum, valid

    def run_model(self, planes, decoder, sample_coordinates, sample_directions, options):
        lib = _lib.load()
        if self.triplane_feature_type not in ("triplane", "trigrid", "trigrid_v2"):
            raise NotImplementedError("triplane_feature_type=%r" % self.triplane_feature_type)
        if options.get("density_noise", 0) > 0:
            raise NotImplementedError("density_noise is a training-only branch")
        planes_nhwc = self._planes_nhwc(planes)
        N, H, W = planes_nhwc.shape[0], planes_nhwc.shape[-3], planes_nhwc.shape[-2]
        D = self.triplane_depth
        assert planes_nhwc.dim() == (5 if D == 1 else 6) and (D == 1 or planes_nhwc.shape[2] == D), planes_nhwc.shape
        coords = _f32c(sample_coordinates)
        npts = coords.shape[1]
        w1, b1, w2, b2 = decoder_params(decoder)
        rgb = torch.empty(N, npts, 32, device=coords.device, dtype=torch.float32)
        sigma = torch.empty(N, npts, 1, device=coords.device, dtype=torch.float32)
        need = int(lib.r3d_run_model_workspace_bytes())
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != coords.device:
            self._workspace = torch.empty(need, device=coords.device, dtype=torch.uint8)
        part, npart = self._absmax_of(planes_nhwc)
        _lib.check(lib.r3d_run_model(_lib.ptr(planes_nhwc), N, H, W, D, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2),
                                     _lib.ptr(b2), _lib.ptr(coords), npts, float(options["box_warp"]),
                                     _lib.ptr(rgb), _lib.ptr(sigma), part, npart,
                                     _lib.ptr(self._workspace), need, _lib.stream_ptr()), "run_model")
        return {"rgb": rgb, "sigma": sigma}
