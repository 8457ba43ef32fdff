"""Generator shell with the reference's `TriPlaneGenerator.synthesis()` contract, HIP hot path inside.

Mirrors modules/eg3ds/models/triplane.py:23-163 for everything downstream of the tri-plane tensor:
`synthesis(ws, camera, cond=None, update_emas=False, cache_backbone=False, use_cached_backbone=False,
**synthesis_kwargs)` returns {'image','image_raw','image_depth','image_feature','plane'} with the same shapes.
The tri-plane producer (StyleGAN2 backbone / img2plane / secc2plane) is a cold PyTorch encoder and stays
outside: pass it as `backbone` (any module with `.synthesis(ws, **kw) -> [N,96,256,256]`) or inject planes
through `_last_planes` + use_cached_backbone=True exactly like the reference's inference loop does
(inference/real3d_infer.py:486-489).

`patch_model(model)` swaps ray_sampler / renderer / superresolution of an already-constructed reference
model (TriPlaneGenerator, OSAvatar_Img2plane, OSAvatarSECC_Img2plane[_Torso]) for the HIP operators,
keeping its parameters (strict state_dict copy).
"""
import copy

import torch
import torch.nn as nn

from .superresolution import SuperresolutionHybrid8XDC, SynthesisBlock, const_bound
from .volumetric_rendering import ImportanceRenderer, OSGDecoder, RaySampler

DEFAULT_HPARAMS = {           # egs/egs_bases/eg3d/base.yaml:20-44 (+ 'auto' limits as every os_avatar config sets)
    "final_resolution": 512, "neural_rendering_resolution": 128, "num_samples_coarse": 48, "num_samples_fine": 48,
    "ray_near": "auto", "ray_far": "auto", "box_warp": 1.0, "w_dim": 512, "base_channel": 32768, "max_channel": 512,
    "ones_ws_for_sr": True, "mask_invalid_rays": False, "triplane_feature_type": "triplane",
}


class TriPlaneGenerator(nn.Module):
    def __init__(self, hp=None, backbone=None):
        super().__init__()
        self.hparams = copy.copy(DEFAULT_HPARAMS)
        if hp is not None:
            self.hparams.update(hp)
        hparams = self.hparams
        self.camera_dim = 25
        self.w_dim = hparams["w_dim"]
        self.img_resolution = hparams["final_resolution"]
        self.img_channels = 3
        self.renderer = ImportanceRenderer(hp=hparams)
        self.renderer.triplane_feature_type = "triplane"
        self.ray_sampler = RaySampler()
        self.neural_rendering_resolution = hparams["neural_rendering_resolution"]
        self.backbone = backbone
        self.decoder = OSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32})
        self.rendering_kwargs = {
            "image_resolution": hparams["final_resolution"], "disparity_space_sampling": False,
            "clamp_mode": "softplus", "c_scale": 1.0, "superresolution_noise_mode": "none", "sr_antialias": True,
            "depth_resolution": hparams["num_samples_coarse"],
            "depth_resolution_importance": hparams["num_samples_fine"],
            "ray_start": hparams["ray_near"], "ray_end": hparams["ray_far"], "box_warp": hparams["box_warp"],
            "avg_camera_radius": 2.7, "avg_camera_pivot": [0, 0, 0.2], "white_back": False,
        }
        self.superresolution = SuperresolutionHybrid8XDC(
            channels=32, img_resolution=self.img_resolution, sr_num_fp16_res=0, sr_antialias=True,
            channel_base=hparams["base_channel"], channel_max=hparams["max_channel"],
            fused_modconv_default="inference_only")
        self._last_planes = None
        self._ones_ws = None

    # -- the reference's public surface (modules/eg3ds/models/triplane.py:73-163) -------------------------------------
    def mapping(self, z, camera, cond=None, truncation_psi=0.7, truncation_cutoff=None, update_emas=False):
        """triplane.py:73-88: the mapping network lives in the (cold, PyTorch) tri-plane producer."""
        if self.backbone is None or not hasattr(self.backbone, "mapping"):
            raise RuntimeError("mapping() needs a backbone with a mapping network (the StyleGAN2 producer is a cold encoder, out of scope)")
        ws = self.backbone.mapping(z, camera * self.rendering_kwargs.get("c_scale", 0), truncation_psi=truncation_psi,
                                   truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        if self.hparams.get("gen_cond_mode", "none") == "mapping":        # triplane.py:85-87
            d_ws = self.backbone.cond_mapping(cond, None, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff,
                                              update_emas=update_emas)
            ws = ws * 0.5 + d_ws * 0.5
        return ws

    def _planes(self, ws, update_emas, cache_backbone, use_cached_backbone, synthesis_kwargs):
        if use_cached_backbone and self._last_planes is not None:
            planes = self._last_planes
        elif self.backbone is None:
            raise RuntimeError("no tri-plane producer: pass backbone=... or set _last_planes and use_cached_backbone=True")
        else:
            planes = self.backbone.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
        if cache_backbone:
            self._last_planes = planes
        return planes.view(len(planes), 3, -1, planes.shape[-2], planes.shape[-1])

    def _ray_images(self, planes, camera, sr_ws=None):
        """Rays -> fused HIP renderer -> the 'raw' neural-rendered images [N,C,R,R] (triplane.py:96-127 / secc_img2plane.py:99-130).
        sr_ws: the ws the SR will be called with next; when the SR is the HIP one and takes R^2 inputs as they come, the ray kernel
        also writes its first operand (images["feature_split"], ImportanceRenderer.forward `_split_for`)."""
        R = self.neural_rendering_resolution
        c2w, K = camera[:, :16].view(-1, 4, 4), camera[:, 16:25].view(-1, 3, 3)
        split = None
        if type(self.ray_sampler) is RaySampler and type(self.renderer) is ImportanceRenderer:
            # both operators are the HIP ones: the rays are generated inside the render launches (same pixels, triplane.py:96-99 + :113)
            sr = self.superresolution
            spec = None
            if (sr_ws is not None and isinstance(sr, SuperresolutionHybrid8XDC) and R == sr.input_resolution
                    and not self.hparams.get("mask_invalid_rays", False)):          # (the mask edits the fp32 image afterwards)
                spec = sr.split_input_spec(sr_ws, c2w.shape[0], c2w.device)
            feat, depth, wsum, valid = self.renderer.forward_camera(planes, self.decoder, c2w, K, R, self.rendering_kwargs, _split_for=spec)
            split = getattr(feat, "_r3d_split", None)
        else:
            origins, directions = self.ray_sampler(c2w, K, R)
            feat, depth, wsum, valid = self.renderer(planes, self.decoder, origins, directions, self.rendering_kwargs)   # [N, R*R, C]
        N = feat.shape[0]
        to_img = lambda t: t.transpose(1, 2).reshape(N, t.shape[-1], R, R)
        images = {"feature": to_img(feat).contiguous(), "depth": to_img(depth), "weights": to_img(wsum).contiguous(), "feature_split": split}
        if self.hparams.get("mask_invalid_rays", False):
            # feature <- -1 and depth <- min valid depth on rays that miss the box (triplane.py:123-126), without the
            # reference's host sync (.item()) and boolean indexing
            keep = valid.reshape(N, 1, R, R)
            d = images["depth"]
            dmin = torch.where(keep, d, torch.full_like(d, float("inf"))).amin()
            images["feature"] = torch.where(keep, images["feature"], torch.full_like(images["feature"], -1.0))
            images["depth"] = torch.where(keep, d, dmin)
        return images

    def _ws_for_sr(self, ws):
        if not self.hparams["ones_ws_for_sr"]:
            return ws
        c = self._ones_ws
        if c is None or c.shape != ws.shape or c.device != ws.device:
            c = self._ones_ws = torch.ones_like(ws)       # one persistent tensor: the SR blocks cache their style vectors on it
        return c

    def synthesis(self, ws, camera, cond=None, update_emas=False, cache_backbone=False, use_cached_backbone=False,
                  **synthesis_kwargs):
        """triplane.py:90-138.  Returns the reference's keys plus 'weights_img' (the name the Real3D shells use,
        secc_img2plane.py:132)."""
        planes = self._planes(ws, update_emas, cache_backbone, use_cached_backbone, synthesis_kwargs)
        sr_ws = self._ws_for_sr(ws)
        sr_kwargs = {k: v for k, v in synthesis_kwargs.items() if k != "noise_mode"}
        im = self._ray_images(planes, camera, sr_ws=None if sr_kwargs else sr_ws)
        feature = im["feature"]
        # |feature| <= 1.002 by construction (sigmoid * 1.002 - 0.001 composited with weights summing to <= 1, then * 2 - 1):
        # the SR's fp16 range fold uses this bound instead of measuring it
        feature._r3d_bound, feature._r3d_depth = const_bound(1.01, feature.shape[0], feature.device), 0
        x = im["feature_split"] if im.get("feature_split") is not None else feature
        sr_image = self.superresolution(feature[:, :3], x, sr_ws,
                                        noise_mode=self.rendering_kwargs["superresolution_noise_mode"], **sr_kwargs)
        return {"image": sr_image.clamp(-1, 1), "image_raw": feature[:, :3].clamp(-1, 1), "image_depth": im["depth"],
                "image_feature": feature[:, 3:], "plane": planes, "weights_img": im["weights"]}

    def sample(self, coordinates, directions, z, camera, cond=None, truncation_psi=1, truncation_cutoff=None, update_emas=False,
               **synthesis_kwargs):
        """triplane.py:140-148."""
        ws = self.mapping(z, camera, cond=cond, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.sample_mixed(coordinates, directions, ws, update_emas=update_emas, **synthesis_kwargs)

    def sample_mixed(self, coordinates, directions, ws, truncation_psi=1, truncation_cutoff=None, update_emas=False,
                     **synthesis_kwargs):
        """triplane.py:150-156."""
        planes = self.backbone.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
        planes = planes.view(len(planes), 3, -1, planes.shape[-2], planes.shape[-1])
        return self.renderer.run_model(planes, self.decoder, coordinates, directions, self.rendering_kwargs)

    def forward(self, z, camera, cond=None, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None,
                update_emas=False, cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        """triplane.py:158-163: mapping -> synthesis."""
        ws = self.mapping(z, camera, cond=cond, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, camera, cond=cond, update_emas=update_emas, cache_backbone=cache_backbone,
                              use_cached_backbone=use_cached_backbone, **synthesis_kwargs)


def _copy_sr_block(dst, src):
    missing = dst.load_state_dict(src.state_dict(), strict=True)
    return missing


def _reference_hparams(model):
    """The reference keeps its configuration in one global dict (utils/commons/hparams.py) that its modules read at call time;
    some shells also keep a copy as `self.hparams`."""
    import sys
    hp = {}
    mod = sys.modules.get("utils.commons.hparams")
    if mod is not None and isinstance(getattr(mod, "hparams", None), dict):
        hp.update(mod.hparams)
    if isinstance(getattr(model, "hparams", None), dict):
        hp.update(model.hparams)
    return hp


def patch_model(model, fuse_warp_sr=True, precision=None):
    """Swap the hot-path operators of a constructed reference model for the HIP ones (in place).  INFERENCE ONLY: the HIP modules
    detach their inputs and build no autograd graph (the reference runs this path under torch.no_grad(), real3d_infer.py:435,479).
    precision: SR precision of the installed blocks (None = the library default 'f16mx': inside the 2e-4 of SURVEY 8(d) on every golden and heavy-tail sweep,
    what bench.py measures; 'f16x3' = the fp32-class tier, 1.3e-6; superresolution.py "Precision policy").

    * model.ray_sampler  -> RaySampler            (created at img2plane_baseline.py:106 / triplane.py:38)
    * model.renderer     -> ImportanceRenderer    (img2plane_baseline.py:104-105 / triplane.py:36-37)
    * model.superresolution (vanilla SuperresolutionHybrid8XDC) -> HIP SuperresolutionHybrid8XDC, or, for the
      torso model whose superresolution is SuperresolutionHybrid8XDC_Warp (secc_img2plane_torso.py:10-11),
      its .block0 / .block1 (called at sr_with_ref.py:83,124 as block(x, img, ws, **kw) -> (x, img)), its torso / background
      fusion stacks (nn.Sequential -> ConvStack, :24-63) and head_torso_block (-> SynthesisBlockNoUp); with fuse mode 'v2' and
      fuse_warp_sr=True its forward is replaced by the fused HIP evaluation (real3dportrait_amd/sr_with_ref.py).
      With hparams weight_fuse=False the reference calls block1(x, None, ws) (sr_with_ref.py:161), which the HIP block does not
      cover: block1 is then left untouched.
    * <backbone>.to_plane_cnn (segformer.py:691-700) -> ConvStack.
    Parameters are copied with strict key matching; the decoder module is left untouched (the renderer reads
    decoder.net[0|2].{weight,bias} directly)."""
    import types
    dev = next(model.parameters()).device
    hp = _reference_hparams(model)
    old_r = model.renderer
    new_r = ImportanceRenderer(hp=getattr(old_r, "hparams", None))
    new_r.triplane_feature_type = getattr(old_r, "triplane_feature_type", "triplane")
    model.renderer = new_r
    model.ray_sampler = RaySampler()
    sr = model.superresolution
    if type(sr).__name__ == "SuperresolutionHybrid8XDC":
        new_sr = SuperresolutionHybrid8XDC(channels=sr.block0.in_channels, img_resolution=512, sr_num_fp16_res=0,
                                           sr_antialias=sr.sr_antialias).to(dev)
        new_sr.load_state_dict(sr.state_dict(), strict=True)
        model.superresolution = new_sr
    else:
        weight_fuse = bool(hp.get("weight_fuse", True))
        for name in ("block0", "block1"):
            old = getattr(sr, name, None)
            if old is None or type(old).__name__ != "SynthesisBlock" or (name == "block1" and not weight_fuse):
                continue
            new = SynthesisBlock(old.in_channels, old.conv1.out_channels, w_dim=old.w_dim, resolution=old.resolution,
                                 img_channels=old.img_channels, is_last=old.is_last, use_fp16=False,
                                 conv_clamp=old.conv1.conv_clamp).to(dev)
            new.load_state_dict(old.state_dict(), strict=True)
            setattr(sr, name, new)
        _patch_fusion_stacks(sr, dev)
        if (fuse_warp_sr and weight_fuse and hp.get("htbsr_head_weight_fuse_mode") == "v2" and hasattr(sr, "torso_model")
                and type(getattr(sr, "head_torso_block", None)).__name__ == "SynthesisBlockNoUp"
                and type(sr.head_torso_block).__module__.startswith("real3dportrait_amd")):
            from . import sr_with_ref
            # snapshot of the keys the fused forward reads -- only those that are SET (absent keys keep the reference's defaults, e.g.
            # torso_model_version 'v1', sr_with_ref.py:84); the reference reads its global hparams at call time, so patch_model must
            # run after set_hparams()
            sr._r3d_state = sr_with_ref.WarpSRState(
                {k: hp[k] for k in ("weight_fuse", "htbsr_head_weight_fuse_mode", "htbsr_head_threshold", "torso_model_version") if k in hp})
            sr._r3d_reference_forward = sr.forward           # the exact-f32 precision runs the reference forward over the patched sub-modules
            sr.forward = types.MethodType(sr_with_ref.forward_v2, sr)
            sr.split_input_spec = types.MethodType(sr_with_ref.warp_split_input_spec, sr)
    for owner in (getattr(model, "secc_img2plane_backbone", None), getattr(model, "img2plane_backbone", None)):
        _patch_sequential(owner, "to_plane_cnn", dev)       # per-frame plane producer tail (segformer.py:691-700)
    from .superresolution import set_sr_precision
    set_sr_precision(model.superresolution, precision)
    return model


def _patch_sequential(owner, name, dev):
    """owner.<name>: nn.Sequential of Conv2d / LeakyReLU / UpsamplingBilinear2d(2) -> ConvStack (same state_dict keys, NCHW in
    and out, so the surrounding reference code is unchanged).  Stacks the HIP conv does not cover are left alone."""
    from .superresolution import ConvStack
    seq = getattr(owner, name, None) if owner is not None else None
    if seq is None or type(seq).__name__ != "Sequential":
        return False
    try:
        setattr(owner, name, ConvStack.from_torch(seq).to(dev))
    except NotImplementedError:
        return False
    return True


def _patch_fusion_stacks(sr, dev):
    """SuperresolutionHybrid8XDC_Warp (sr_with_ref.py:24-63): the torso / background fusion stacks and head_torso_block."""
    from .superresolution import SynthesisBlockNoUp
    for name in ("torso_encoder", "bg_encoder", "fuse_head_torso_convs", "fuse_fg_bg_convs"):
        _patch_sequential(sr, name, dev)
    old = getattr(sr, "head_torso_block", None)
    if old is not None and type(old).__name__ == "SynthesisBlockNoUp":
        new = SynthesisBlockNoUp(old.in_channels, old.conv1.out_channels, w_dim=old.w_dim, resolution=old.resolution,
                                 img_channels=old.img_channels, is_last=old.is_last, use_fp16=False,
                                 conv_clamp=old.conv1.conv_clamp).to(dev)
        new.load_state_dict(old.state_dict(), strict=True)
        sr.head_torso_block = new
