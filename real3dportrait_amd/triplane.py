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

from .superresolution import SuperresolutionHybrid8XDC, SynthesisBlock
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

    def synthesis(self, ws, camera, cond=None, update_emas=False, cache_backbone=False, use_cached_backbone=False,
                  **synthesis_kwargs):
        hparams = self.hparams
        ret = {}
        cam2world_matrix = camera[:, :16].view(-1, 4, 4)
        intrinsics = camera[:, 16:25].view(-1, 3, 3)
        R = self.neural_rendering_resolution
        ray_origins, ray_directions = self.ray_sampler(cam2world_matrix, intrinsics, R)
        N, M, _ = ray_origins.shape
        if use_cached_backbone and self._last_planes is not None:
            planes = self._last_planes
        else:
            if self.backbone is None:
                raise RuntimeError("no tri-plane producer: pass backbone=... or set _last_planes and "
                                   "use_cached_backbone=True")
            planes = self.backbone.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
        if cache_backbone:
            self._last_planes = planes
        planes = planes.view(len(planes), 3, -1, planes.shape[-2], planes.shape[-1])

        feature_samples, depth_samples, weights_samples, is_ray_valid = self.renderer(
            planes, self.decoder, ray_origins, ray_directions, self.rendering_kwargs)

        feature_image = feature_samples.permute(0, 2, 1).reshape(N, feature_samples.shape[-1], R, R).contiguous()
        depth_image = depth_samples.permute(0, 2, 1).reshape(N, 1, R, R)
        if hparams.get("mask_invalid_rays", False):
            mask = is_ray_valid.reshape(N, 1, R, R)
            feature_image[~mask.repeat(1, feature_image.shape[1], 1, 1)] = -1
            depth_image[~mask] = depth_image[mask].min().item()
        rgb_image = feature_image[:, :3]
        ws_to_sr = torch.ones_like(ws) if hparams["ones_ws_for_sr"] else ws
        sr_image = self.superresolution(
            rgb_image, feature_image, ws_to_sr, noise_mode=self.rendering_kwargs["superresolution_noise_mode"],
            **{k: synthesis_kwargs[k] for k in synthesis_kwargs if k != "noise_mode"})
        rgb_image = rgb_image.clamp(-1, 1)
        sr_image = sr_image.clamp(-1, 1)
        ret.update({"image": sr_image, "image_raw": rgb_image, "image_depth": depth_image,
                    "image_feature": feature_image[:, 3:], "plane": planes, "weights_image":
                    weights_samples.permute(0, 2, 1).reshape(N, 1, R, R)})
        return ret

    def sample_mixed(self, coordinates, directions, ws, truncation_psi=1, truncation_cutoff=None, update_emas=False,
                     **synthesis_kwargs):
        planes = self.backbone.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
        planes = planes.view(len(planes), 3, 32, planes.shape[-2], planes.shape[-1])
        return self.renderer.run_model(planes, self.decoder, coordinates, directions, self.rendering_kwargs)

    def forward(self, ws, camera, **kw):
        return self.synthesis(ws, camera, **kw)


def _copy_sr_block(dst, src):
    missing = dst.load_state_dict(src.state_dict(), strict=True)
    return missing


def patch_model(model):
    """Swap the hot-path operators of a constructed reference model for the HIP ones (in place).

    * model.ray_sampler  -> RaySampler            (created at img2plane_baseline.py:106 / triplane.py:38)
    * model.renderer     -> ImportanceRenderer    (img2plane_baseline.py:104-105 / triplane.py:36-37)
    * model.superresolution (vanilla SuperresolutionHybrid8XDC) -> HIP SuperresolutionHybrid8XDC, or, for the
      torso model whose superresolution is SuperresolutionHybrid8XDC_Warp (secc_img2plane_torso.py:10-11),
      its .block0 / .block1 (called at sr_with_ref.py:83,124 as block(x, img, ws, **kw) -> (x, img)), its torso / background
      fusion stacks (nn.Sequential -> ConvStack, :24-63) and head_torso_block (-> SynthesisBlockNoUp).
    * <backbone>.to_plane_cnn (segformer.py:691-700) -> ConvStack.
    Parameters are copied with strict key matching; the decoder module is left untouched (the renderer reads
    decoder.net[0|2].{weight,bias} directly)."""
    dev = next(model.parameters()).device
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
        for name in ("block0", "block1"):
            old = getattr(sr, name, None)
            if old is None or type(old).__name__ != "SynthesisBlock":
                continue
            new = SynthesisBlock(old.in_channels, old.conv1.out_channels, w_dim=old.w_dim, resolution=old.resolution,
                                 img_channels=old.img_channels, is_last=old.is_last, use_fp16=False,
                                 conv_clamp=old.conv1.conv_clamp).to(dev)
            new.load_state_dict(old.state_dict(), strict=True)
            setattr(sr, name, new)
        _patch_fusion_stacks(sr, dev)
    for owner in (getattr(model, "secc_img2plane_backbone", None), getattr(model, "img2plane_backbone", None)):
        _patch_sequential(owner, "to_plane_cnn", dev)       # per-frame plane producer tail (segformer.py:691-700)
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
