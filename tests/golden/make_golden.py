#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own PyTorch code on CPU.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

The reference has no tests or golden vectors for this path, so these files ARE the
pin: tests/test_oracle_golden.py replays them against oracle/r3d_oracle.c (CPU) and
tests/test_gpu_parity.py replays them against the HIP path (GPU).

The two RNG draws inside the reference renderer (torch.rand_like at renderer.py:226 and torch.rand at
renderer.py:281) are replaced, for the duration of the call, by functions returning pre-drawn arrays
that are stored with the outputs, so every implementation sees identical sampling noise.
"""
import contextlib
import os
import sys

import numpy as np

REF = os.environ.get("R3D_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import torch  # noqa: E402

from real3dportrait_amd import synth  # noqa: E402

from modules.eg3ds.models.networks_stylegan2 import SynthesisBlock  # noqa: E402
from modules.eg3ds.models.superresolution import SuperresolutionHybrid8XDC  # noqa: E402
from modules.eg3ds.models.triplane import OSGDecoder, TriPlaneGenerator  # noqa: E402
from modules.eg3ds.volumetric_rendering.ray_sampler import RaySampler  # noqa: E402
from modules.eg3ds.volumetric_rendering.renderer import ImportanceRenderer  # noqa: E402
from modules.eg3ds.volumetric_rendering import math_utils  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


@contextlib.contextmanager
def injected_noise(noise_c, u_f):
    """Make the renderer's two RNG draws return our arrays (shape-checked)."""
    real_rand_like, real_rand = torch.rand_like, torch.rand
    state = {"c": 0, "f": 0}

    def fake_rand_like(t, *a, **k):
        assert tuple(t.shape[:3]) == tuple(noise_c.shape[:3]), (t.shape, noise_c.shape)
        state["c"] += 1
        return torch.from_numpy(noise_c).reshape(t.shape).clone()

    def fake_rand(*shape, **k):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert shape == tuple(u_f.shape), (shape, u_f.shape)
        state["f"] += 1
        return torch.from_numpy(u_f).clone()

    torch.rand_like, torch.rand = fake_rand_like, fake_rand
    try:
        yield state
    finally:
        torch.rand_like, torch.rand = real_rand_like, real_rand


def make_decoder(dec_np):
    dec = OSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32}).eval()
    with torch.no_grad():
        dec.net[0].weight.copy_(torch.from_numpy(dec_np[0]))
        dec.net[0].bias.copy_(torch.from_numpy(dec_np[1]))
        dec.net[2].weight.copy_(torch.from_numpy(dec_np[2]))
        dec.net[2].bias.copy_(torch.from_numpy(dec_np[3]))
    return dec


def load_block(block, p):
    with torch.no_grad():
        for name in ("conv0", "conv1", "torgb"):
            layer = getattr(block, name)
            w, b, aw, ab = p[name]
            layer.weight.copy_(torch.from_numpy(w))
            layer.bias.copy_(torch.from_numpy(b))
            layer.affine.weight.copy_(torch.from_numpy(aw))
            layer.affine.bias.copy_(torch.from_numpy(ab))


def opts(Nc, Nf, box_warp=1.0, white_back=False):
    return {"ray_start": "auto", "ray_end": "auto", "box_warp": box_warp, "depth_resolution": Nc,
            "depth_resolution_importance": Nf, "disparity_space_sampling": False, "clamp_mode": "softplus",
            "white_back": white_back}


def render_case(name, planes, cams, R, Nc, Nf, dec_seed, noise_seed, box_warp=1.0, white_back=False,
                sigma_bias=0.0, store_planes=True, planes_spec=None, trigrid_depth=1):
    cams = np.asarray(cams, np.float32).reshape(-1, 25)
    N, M = cams.shape[0], R * R
    dec_np = synth.synth_decoder(dec_seed, sigma_bias=sigma_bias)
    dec = make_decoder(dec_np)
    noise_c = synth.synth_noise(noise_seed, (N, M, Nc, 1), stream=7)
    u_f = synth.synth_noise(noise_seed, (N * M, max(Nf, 1)), stream=8)[:, :Nf].copy() if Nf > 0 else np.zeros((N * M, 0), np.float32)
    cam_t = torch.from_numpy(cams)
    c2w, K = cam_t[:, :16].view(-1, 4, 4), cam_t[:, 16:].view(-1, 3, 3)
    with torch.no_grad():
        o, d = RaySampler()(c2w, K, R)
        rs, re = math_utils.get_ray_limits_box(o, d, box_side_length=box_warp)
        hp = {"enable_rescale_plane_regulation": False, "triplane_feature_type": "triplane"}
        if trigrid_depth > 1:       # sample_from_trigrids branch (renderer.py:180-181), depth from hparams
            hp = {"enable_rescale_plane_regulation": False, "triplane_feature_type": "trigrid", "triplane_depth": trigrid_depth}
        ren = ImportanceRenderer(hp=hp).eval()
        with injected_noise(noise_c, u_f) as st:
            rgb, depth, wsum, valid = ren(torch.from_numpy(planes), dec, o, d, opts(Nc, Nf, box_warp, white_back))
        assert st["c"] == 1 and st["f"] == (1 if Nf > 0 else 0)
    out = dict(cams=cams, R=R, Nc=Nc, Nf=Nf, box_warp=np.float32(box_warp), white_back=int(white_back), triplane_depth=trigrid_depth,
               dec_w1=dec_np[0], dec_b1=dec_np[1], dec_w2=dec_np[2], dec_b2=dec_np[3],
               noise_c=noise_c, u_f=u_f,
               origins=o.numpy(), dirs=d.numpy(), raw_start=rs.numpy(), raw_end=re.numpy(),
               rgb=rgb.numpy(), depth=depth.numpy(), wsum=wsum.numpy(), valid=valid.numpy())
    if store_planes:
        out["planes"] = planes
    else:
        out["planes_spec"] = np.asarray(planes_spec, np.float64)   # (seed, N, C, H, W, scale)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "rgb", rgb.shape, "valid frac", float(valid.float().mean()), "wsum mean", float(wsum.mean()))


def run_model_case(name, trigrid_depth=1):
    planes = synth.synth_planes(21, N=2, H=32, W=32) if trigrid_depth == 1 else \
        synth.hash_unitvar(25, (2, 3, 32 * trigrid_depth, 16, 16), stream=1)
    dec_np = synth.synth_decoder(22)
    dec = make_decoder(dec_np)
    coords = (synth.synth_noise(23, (2, 777, 3)) - 0.5) * 1.3     # includes points outside the +-0.5 box
    hp = {"enable_rescale_plane_regulation": False, "triplane_feature_type": "triplane"}
    if trigrid_depth > 1:
        hp = {"enable_rescale_plane_regulation": False, "triplane_feature_type": "trigrid_v2", "triplane_depth": trigrid_depth}
    ren = ImportanceRenderer(hp=hp).eval()
    with torch.no_grad():
        out = ren.run_model(torch.from_numpy(planes), dec, torch.from_numpy(coords), None, opts(16, 0))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), planes=planes, coords=coords.astype(np.float32),
                        dec_w1=dec_np[0], dec_b1=dec_np[1], dec_w2=dec_np[2], dec_b2=dec_np[3],
                        rgb=out["rgb"].numpy(), sigma=out["sigma"].numpy(), box_warp=np.float32(1.0), triplane_depth=trigrid_depth)
    print(name, out["rgb"].shape)


def sr_small_case(name):
    """The reference's SynthesisBlock pair (32->256 @ res 32, 256->128 @ res 64): same layers as
    SuperresolutionHybrid8XDC (superresolution.py:342-345) at 1/8 spatial size."""
    seed = 31
    params = synth.synth_sr_params(seed)
    kw = dict(w_dim=512, img_channels=3, use_fp16=False, conv_clamp=None, channel_base=32768, channel_max=512,
              fused_modconv_default="inference_only")
    b0 = SynthesisBlock(32, 256, resolution=32, is_last=False, **kw).eval()
    b1 = SynthesisBlock(256, 128, resolution=64, is_last=True, **kw).eval()
    load_block(b0, params[0])
    load_block(b1, params[1])
    x = synth.hash_unitvar(seed, (1, 32, 16, 16), stream=1)
    rgb = synth.hash_unitvar(seed, (1, 3, 16, 16), stream=2) * np.float32(0.5)
    ws = (np.ones((1, 3, 512), np.float32) + synth.hash_unitvar(seed, (1, 3, 512), stream=3) * np.float32(0.1))
    with torch.no_grad():
        x0, r0 = b0(torch.from_numpy(x), torch.from_numpy(rgb), torch.from_numpy(ws), noise_mode="none")
        x1, r1 = b1(x0, r0, torch.from_numpy(ws), noise_mode="none")
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, x=x, rgb=rgb, ws=ws,
                        x0=x0.numpy()[:, ::4], rgb0=r0.numpy(), x1=x1.numpy()[:, ::8], rgb1=r1.numpy())
    print(name, r1.shape, float(r1.abs().mean()))


def sr_full_case(name):
    """SuperresolutionHybrid8XDC 128^2 -> 512^2 with all-ones ws (what Real3D feeds it,
    secc_img2plane_torso.py:15 / triplane.py:131-132)."""
    seed = 41
    params = synth.synth_sr_params(seed)
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True,
                                   channel_base=32768, channel_max=512, fused_modconv_default="inference_only").eval()
    load_block(sr.block0, params[0])
    load_block(sr.block1, params[1])
    x = synth.hash_unitvar(seed, (1, 32, 128, 128), stream=1)
    rgb = x[:, :3].copy()
    ws = np.ones((1, 14, 512), np.float32)
    with torch.no_grad():
        out = sr(torch.from_numpy(rgb), torch.from_numpy(x), torch.from_numpy(ws), noise_mode="none").numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, strided=out[:, :, ::4, ::4],
                        corner=out[:, :, :96, :96], tail=out[:, :, -64:, -64:],
                        mean=np.float64(out.mean()), absmean=np.float64(np.abs(out).mean()))
    print(name, out.shape, float(np.abs(out).mean()))


def synthesis_case(name):
    """TriPlaneGenerator.synthesis (modules/eg3ds/models/triplane.py:90-138) at the reference default:
    R=128, 48+48 samples, SR to 512^2, cached planes, ones ws."""
    from utils.commons.hparams import set_hparams, hparams
    set_hparams(os.path.join(REF, "egs/egs_bases/eg3d/base.yaml"), print_hparams=False)
    hparams.update(ray_near="auto", ray_far="auto", ones_ws_for_sr=True, enable_rescale_plane_regulation=False)
    G = TriPlaneGenerator().eval()
    seed = 51
    dec_np = synth.synth_decoder(seed, sigma_bias=4.0)
    with torch.no_grad():
        G.decoder.net[0].weight.copy_(torch.from_numpy(dec_np[0]))
        G.decoder.net[0].bias.copy_(torch.from_numpy(dec_np[1]))
        G.decoder.net[2].weight.copy_(torch.from_numpy(dec_np[2]))
        G.decoder.net[2].bias.copy_(torch.from_numpy(dec_np[3]))
    params = synth.synth_sr_params(seed)
    load_block(G.superresolution.block0, params[0])
    load_block(G.superresolution.block1, params[1])
    planes = synth.synth_planes(seed, N=1)
    G._last_planes = torch.from_numpy(planes).view(1, 96, 256, 256)
    cam = synth.look_at_camera(0.25, -0.1)[None]
    R, Nc, Nf = 128, 48, 48
    noise_c = synth.synth_noise(seed, (1, R * R, Nc, 1), stream=7)
    u_f = synth.synth_noise(seed, (R * R, Nf), stream=8)
    with torch.no_grad(), injected_noise(noise_c, u_f):
        out = G.synthesis(torch.ones(1, G.backbone.num_ws, 512), torch.from_numpy(cam), use_cached_backbone=True,
                          noise_mode="none")
    img = out["image"].numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, cam=cam, R=R, Nc=Nc, Nf=Nf,
                        image_strided=img[:, :, ::4, ::4], image_corner=img[:, :, :96, :96],
                        image_raw=out["image_raw"].numpy(), image_depth=out["image_depth"].numpy(),
                        image_feature_strided=out["image_feature"].numpy()[:, ::4],
                        absmean=np.float64(np.abs(img).mean()))
    print(name, img.shape, float(np.abs(img).mean()))


def torch_stack(plan, params):
    """The reference builds these stacks inline from torch.nn modules (sr_with_ref.py:24-63); same construction."""
    mods = []
    for (ci, co, k, lrelu), (w, b) in zip(plan, params):
        conv = torch.nn.Conv2d(ci, co, k, 1, padding=k // 2)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w))
            conv.bias.copy_(torch.from_numpy(b))
        mods.append(conv)
        if lrelu:
            mods.append(torch.nn.LeakyReLU())
    return torch.nn.Sequential(*mods).eval()


def fusion_case(name, R=24):
    """Convolutional part of SuperresolutionHybrid8XDC_Warp.forward, fuse mode 'v2' (sr_with_ref.py:101-123) at
    R x R instead of 256 x 256: torso_encoder, bg_encoder, alpha-cat, fuse_head_torso_convs, head_torso_block
    (the reference's SynthesisBlockNoUp), occlusion-cat, fuse_fg_bg_convs.  The warp-based torso model that produces
    deformed_torso_hid / occlusion is out of scope; its outputs are synthetic inputs here."""
    from modules.eg3ds.models.superresolution import SynthesisBlockNoUp
    seed = 57
    stacks = {k: torch_stack(plan, synth.synth_conv_stack(seed, plan, 300 + 20 * i))
              for i, (k, plan) in enumerate(synth.FUSION_STACKS.items())}
    blk = SynthesisBlockNoUp(256, 256, w_dim=512, resolution=R, img_channels=3, is_last=False, use_fp16=False,
                             conv_clamp=None, channel_base=32768, channel_max=512,
                             fused_modconv_default="inference_only").eval()
    load_block(blk, synth.synth_sr_block(seed, 256, 256, 512, 400))
    t = lambda shape, s, g=1.0: torch.from_numpy(synth.hash_unitvar(seed, shape, stream=s) * np.float32(g))
    x_head, hid, bg, rgb, rgb_torso = t((1, 256, R, R), 1), t((1, 64, R, R), 2), t((1, 3, R, R), 3, 0.5), t((1, 3, R, R), 4, 0.5), t((1, 3, R, R), 5, 0.5)
    alpha = torch.from_numpy(synth.synth_noise(seed, (1, 1, R, R), stream=6))
    occ = torch.from_numpy(synth.synth_noise(seed, (1, 1, R, R), stream=8))
    ws = torch.from_numpy(np.ones((1, 3, 512), np.float32) + synth.hash_unitvar(seed, (1, 3, 512), stream=9) * np.float32(0.1))
    with torch.no_grad():
        x_torso = stacks["torso_encoder"](hid)
        x_bg = stacks["bg_encoder"](bg)
        rgb1 = rgb * alpha + rgb_torso * (1 - alpha)
        x1 = stacks["fuse_head_torso_convs"](torch.cat([x_head * alpha, x_torso * (1 - alpha)], dim=1))
        x2, rgb2 = blk(x1, rgb1.clone(), ws, noise_mode="none")
        x3 = stacks["fuse_fg_bg_convs"](torch.cat([x2 * occ, x_bg * (1 - occ)], dim=1))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, R=R,
                        x_torso=x_torso.numpy()[:, ::4], x_bg=x_bg.numpy()[:, ::4], x1=x1.numpy()[:, ::4],
                        x2=x2.numpy()[:, ::4], rgb2=rgb2.numpy(), x3=x3.numpy()[:, ::4])
    print(name, x3.shape, float(x3.abs().mean()), float(rgb2.abs().mean()))


def toplane_case(name, r=24):
    """Per-frame plane producer tail: SegFormerSECC2PlaneBackbone.to_plane_cnn (segformer.py:691-700) on a
    synthetic fused feature map, the view / flip / stack of its forward (:721-729) and the cano + secc add
    (secc_img2plane.py:76-77), at r x r -> 2r x 2r instead of 128 -> 256.  The Sequential is built exactly as the
    reference builds it (plain torch.nn modules); the MiT encoder in front of it is a cold stage and out of scope."""
    seed = 71
    plan = synth.TO_PLANE_CNN
    params = synth.synth_conv_stack(seed, plan, 500)
    mods = []
    for i, ((ci, co, k, lrelu), (w, b)) in enumerate(zip(plan, params)):
        if i == synth.TO_PLANE_CNN_UP_BEFORE:
            mods.append(torch.nn.UpsamplingBilinear2d(scale_factor=2.))
        conv = torch.nn.Conv2d(ci, co, k, 1, padding=1)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w)); conv.bias.copy_(torch.from_numpy(b))
        mods.append(conv)
        if lrelu:
            mods.append(torch.nn.LeakyReLU(negative_slope=0.01, inplace=True))
    cnn = torch.nn.Sequential(*mods).eval()
    feat = torch.from_numpy(synth.hash_unitvar(seed, (1, 256, r, r), stream=1))
    cano = torch.from_numpy(synth.hash_unitvar(seed, (1, 3, 32, 2 * r, 2 * r), stream=2))
    with torch.no_grad():
        planes = cnn(feat)
        planes = planes.view(len(planes), 3, -1, planes.shape[-2], planes.shape[-1])
        pxy, pxz, pzy = torch.flip(planes[:, 0], [2]), torch.flip(planes[:, 1], [2]), torch.flip(planes[:, 2], [2, 3])
        secc = torch.stack([pxy, pxz, pzy], dim=1)
        out = cano + secc
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, r=r, planes=out.numpy())
    print(name, out.shape, float(secc.abs().mean()))


def sr_cfg5_case(name):
    """BASELINE config 5 (stress): the SR at 2x spatial size, 256^2 -> 512^2 -> 1024^2.  SuperresolutionHybrid8XDC itself asserts a
    512 output (superresolution.py:334) and SynthesisBlock checks its input size against `resolution` (networks_stylegan2.py:447), so
    the same two reference SynthesisBlocks are built with resolution 512 / 1024 (the convolutions are resolution-agnostic with
    noise_mode='none'; SURVEY 8d cfg 5) and chained exactly like superresolution.py:348-359 does."""
    seed = 45
    params = synth.synth_sr_params(seed)
    kw = dict(w_dim=512, img_channels=3, use_fp16=False, conv_clamp=None, channel_base=32768, channel_max=512,
              fused_modconv_default="inference_only")
    b0 = SynthesisBlock(32, 256, resolution=512, is_last=False, **kw).eval()
    b1 = SynthesisBlock(256, 128, resolution=1024, is_last=True, **kw).eval()
    load_block(b0, params[0])
    load_block(b1, params[1])
    x = synth.hash_unitvar(seed, (1, 32, 256, 256), stream=1)
    rgb = x[:, :3].copy()
    ws = torch.ones(1, 14, 512)[:, -1:, :].repeat(1, 3, 1)
    with torch.no_grad():
        x0, r0 = b0(torch.from_numpy(x), torch.from_numpy(rgb), ws, noise_mode="none")
        x1, out = b1(x0, r0, ws, noise_mode="none")
    out = out.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, strided=out[:, :, ::8, ::8], corner=out[:, :, :64, :64],
                        tail=out[:, :, -48:, -48:], mid=out[:, :, 480:544, 480:544], absmean=np.float64(np.abs(out).mean()))
    print(name, out.shape, float(np.abs(out).mean()))


def fusion_full_case(name):
    """fusion_case at the reference size 256 x 256 (sr_with_ref.py:101-123, fuse mode v2): exercises the multi-tile / Cin = 512
    multi-stage paths.  Only strided slices and crops are stored."""
    from modules.eg3ds.models.superresolution import SynthesisBlockNoUp
    seed, R = 59, 256
    stacks = {k: torch_stack(plan, synth.synth_conv_stack(seed, plan, 300 + 20 * i))
              for i, (k, plan) in enumerate(synth.FUSION_STACKS.items())}
    blk = SynthesisBlockNoUp(256, 256, w_dim=512, resolution=R, img_channels=3, is_last=False, use_fp16=False,
                             conv_clamp=None, channel_base=32768, channel_max=512,
                             fused_modconv_default="inference_only").eval()
    load_block(blk, synth.synth_sr_block(seed, 256, 256, 512, 400))
    t = lambda shape, s, g=1.0: torch.from_numpy(synth.hash_unitvar(seed, shape, stream=s) * np.float32(g))
    x_head, hid, bg, rgb, rgb_torso = t((1, 256, R, R), 1), t((1, 64, R, R), 2), t((1, 3, R, R), 3, 0.5), t((1, 3, R, R), 4, 0.5), t((1, 3, R, R), 5, 0.5)
    alpha = torch.from_numpy(synth.synth_noise(seed, (1, 1, R, R), stream=6))
    occ = torch.from_numpy(synth.synth_noise(seed, (1, 1, R, R), stream=8))
    ws = torch.from_numpy(np.ones((1, 3, 512), np.float32) + synth.hash_unitvar(seed, (1, 3, 512), stream=9) * np.float32(0.1))
    with torch.no_grad():
        x_torso = stacks["torso_encoder"](hid)
        x_bg = stacks["bg_encoder"](bg)
        rgb1 = rgb * alpha + rgb_torso * (1 - alpha)
        x1 = stacks["fuse_head_torso_convs"](torch.cat([x_head * alpha, x_torso * (1 - alpha)], dim=1))
        x2, rgb2 = blk(x1, rgb1.clone(), ws, noise_mode="none")
        x3 = stacks["fuse_fg_bg_convs"](torch.cat([x2 * occ, x_bg * (1 - occ)], dim=1))
    sl = lambda v: v.numpy()[:, ::16, ::8, ::8]
    cr = lambda v: v.numpy()[:, ::32, 120:152, 232:]           # a crop that straddles tile borders and the right image edge
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, R=R,
                        x_torso=sl(x_torso), x_bg=sl(x_bg), x1=sl(x1), x2=sl(x2), x3=sl(x3), rgb2=rgb2.numpy()[:, :, ::4, ::4],
                        x1_crop=cr(x1), x3_crop=cr(x3))
    print(name, x3.shape, float(x3.abs().mean()), float(rgb2.abs().mean()))


def toplane_full_case(name):
    """toplane_case at the reference size 128^2 -> 256^2 (segformer.py:691-700,721-729 + secc_img2plane.py:76-77)."""
    seed, r = 73, 128
    plan = synth.TO_PLANE_CNN
    params = synth.synth_conv_stack(seed, plan, 500)
    mods = []
    for i, ((ci, co, k, lrelu), (w, b)) in enumerate(zip(plan, params)):
        if i == synth.TO_PLANE_CNN_UP_BEFORE:
            mods.append(torch.nn.UpsamplingBilinear2d(scale_factor=2.))
        conv = torch.nn.Conv2d(ci, co, k, 1, padding=1)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(w)); conv.bias.copy_(torch.from_numpy(b))
        mods.append(conv)
        if lrelu:
            mods.append(torch.nn.LeakyReLU(negative_slope=0.01, inplace=True))
    cnn = torch.nn.Sequential(*mods).eval()
    feat = torch.from_numpy(synth.hash_unitvar(seed, (1, 256, r, r), stream=1))
    cano = torch.from_numpy(synth.hash_unitvar(seed, (1, 3, 32, 2 * r, 2 * r), stream=2))
    with torch.no_grad():
        planes = cnn(feat)
        planes = planes.view(len(planes), 3, -1, planes.shape[-2], planes.shape[-1])
        pxy, pxz, pzy = torch.flip(planes[:, 0], [2]), torch.flip(planes[:, 1], [2]), torch.flip(planes[:, 2], [2, 3])
        out = (cano + torch.stack([pxy, pxz, pzy], dim=1)).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, r=r, strided=out[:, :, :, ::8, ::8], corner=out[:, :, ::4, :40, :40],
                        tail=out[:, :, ::4, -40:, -40:])
    print(name, out.shape, float(np.abs(out).mean()))


def synthesis_mask_case(name):
    """TriPlaneGenerator.synthesis with hparams['mask_invalid_rays'] = True (triplane.py:123-126) and a camera pushed sideways so that a
    large share of the rays misses the box."""
    from utils.commons.hparams import set_hparams, hparams
    set_hparams(os.path.join(REF, "egs/egs_bases/eg3d/base.yaml"), print_hparams=False)
    hparams.update(ray_near="auto", ray_far="auto", ones_ws_for_sr=True, enable_rescale_plane_regulation=False, mask_invalid_rays=True)
    G = TriPlaneGenerator().eval()
    seed = 53
    dec_np = synth.synth_decoder(seed, sigma_bias=4.0)
    with torch.no_grad():
        G.decoder.net[0].weight.copy_(torch.from_numpy(dec_np[0])); G.decoder.net[0].bias.copy_(torch.from_numpy(dec_np[1]))
        G.decoder.net[2].weight.copy_(torch.from_numpy(dec_np[2])); G.decoder.net[2].bias.copy_(torch.from_numpy(dec_np[3]))
    params = synth.synth_sr_params(seed)
    load_block(G.superresolution.block0, params[0])
    load_block(G.superresolution.block1, params[1])
    G._last_planes = torch.from_numpy(synth.synth_planes(seed, N=1)).view(1, 96, 256, 256)
    cam = synth.look_at_camera(0.1, 0.05)
    cam[3] += 0.45
    cam = cam[None]
    R, Nc, Nf = 128, 48, 48
    noise_c = synth.synth_noise(seed, (1, R * R, Nc, 1), stream=7)
    u_f = synth.synth_noise(seed, (R * R, Nf), stream=8)
    with torch.no_grad(), injected_noise(noise_c, u_f):
        out = G.synthesis(torch.ones(1, G.backbone.num_ws, 512), torch.from_numpy(cam), use_cached_backbone=True, noise_mode="none")
    hparams.update(mask_invalid_rays=False)
    img = out["image"].numpy()
    raw = out["image_raw"].numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, cam=cam, R=R, Nc=Nc, Nf=Nf,
                        image_strided=img[:, :, ::4, ::4], image_raw=raw, image_depth=out["image_depth"].numpy(),
                        image_feature_strided=out["image_feature"].numpy()[:, ::4],
                        masked_frac=np.float64((raw == -1).mean()))
    print(name, img.shape, "masked frac", float((raw == -1).mean()))


def warp_sr_case(name):
    """The reference's SuperresolutionHybrid8XDC_Warp.forward (sr_with_ref.py:67-137, fuse mode v2, the shipped torso configuration
    egs/os_avatar/real3d_orig/secc_img2plane_torso_orig.yaml) at full size, with its face-vid2vid `torso_model` replaced by the
    deterministic stand-in of tests/warp_mock.py (that network is out of scope and needs .cuda())."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import ref_stubs
    import warp_mock
    ref_stubs.install()
    from utils.commons.hparams import set_hparams, hparams
    set_hparams(os.path.join(REF, "egs/os_avatar/real3d_orig/secc_img2plane_torso_orig.yaml"), print_hparams=False)
    assert hparams["htbsr_head_weight_fuse_mode"] == "v2" and hparams["torso_model_version"] == "v2"
    from modules.real3d.super_resolution.sr_with_ref import SuperresolutionHybrid8XDC_Warp
    sr = SuperresolutionHybrid8XDC_Warp(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True, channel_base=32768,
                                        channel_max=512, fused_modconv_default="inference_only").eval()
    sr.torso_model = warp_mock.MockTorso()
    warp_mock.load_warp_params(sr, load_block)
    i = {k: torch.from_numpy(v) for k, v in warp_mock.warp_inputs().items()}
    with torch.no_grad():
        out, ret = sr(i["x"][:, :3].contiguous(), i["x"], i["ws"], i["ref_torso_rgb"], i["ref_bg_rgb"], i["weights_img"], None, None, None,
                      noise_mode="none")
    out = out.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=warp_mock.SEED, threshold=np.float64(hparams["htbsr_head_threshold"]),
                        strided=out[:, :, ::4, ::4], corner=out[:, :, :96, :96], tail=out[:, :, -64:, -64:],
                        absmean=np.float64(np.abs(out).mean()))
    print(name, out.shape, float(np.abs(out).mean()))


def warp_sr_two_stage_case(name):
    """The reference's two-stage entry of the same module (sr_with_ref.py:164-218: infer_forward_stage1 -> infer_forward_stage2; unused by
    real3d_infer.py, kept by the mirror class): same parameters, inputs and stand-in torso network as warp_sr_case."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import ref_stubs
    import warp_mock
    ref_stubs.install()
    from utils.commons.hparams import set_hparams, hparams
    set_hparams(os.path.join(REF, "egs/os_avatar/real3d_orig/secc_img2plane_torso_orig.yaml"), print_hparams=False)
    from modules.real3d.super_resolution.sr_with_ref import SuperresolutionHybrid8XDC_Warp
    sr = SuperresolutionHybrid8XDC_Warp(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True, channel_base=32768,
                                        channel_max=512, fused_modconv_default="inference_only").eval()
    sr.torso_model = warp_mock.MockTorso()
    warp_mock.load_warp_params(sr, load_block)
    i = {k: torch.from_numpy(v) for k, v in warp_mock.warp_inputs().items()}
    with torch.no_grad():
        ret = sr.infer_forward_stage1(i["x"][:, :3].contiguous(), i["x"], i["ws"], i["ref_torso_rgb"], i["ref_bg_rgb"], i["weights_img"], None, None, None,
                                      noise_mode="none")
        x0 = ret["x"].numpy().copy()
        out, ret = sr.infer_forward_stage2(ret, noise_mode="none")
    out = out.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=warp_mock.SEED, strided=out[:, :, ::4, ::4], corner=out[:, :, :96, :96],
                        tail=out[:, :, -64:, -64:], absmean=np.float64(np.abs(out).mean()), x0_strided=x0[:, ::8, ::8, ::8],
                        keys=np.array(sorted(k for k in ret if not k.startswith("_"))))
    print(name, out.shape, float(np.abs(out).mean()))


def warp_sr_v1_case(name):
    """SuperresolutionHybrid8XDC_Warp.forward with htbsr_head_weight_fuse_mode = 'v1' (sr_with_ref.py:92-104: the direct alpha blend of x and x_torso;
    not a shipped configuration): same parameters, inputs and stand-in torso network as warp_sr_case."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import ref_stubs
    import warp_mock
    ref_stubs.install()
    from utils.commons.hparams import set_hparams, hparams
    set_hparams(os.path.join(REF, "egs/os_avatar/real3d_orig/secc_img2plane_torso_orig.yaml"), print_hparams=False)
    hparams["htbsr_head_weight_fuse_mode"] = "v1"
    from modules.real3d.super_resolution.sr_with_ref import SuperresolutionHybrid8XDC_Warp
    sr = SuperresolutionHybrid8XDC_Warp(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True, channel_base=32768,
                                        channel_max=512, fused_modconv_default="inference_only").eval()
    sr.torso_model = warp_mock.MockTorso()
    warp_mock.load_warp_params(sr, load_block)
    i = {k: torch.from_numpy(v) for k, v in warp_mock.warp_inputs().items()}
    with torch.no_grad():
        out, ret = sr(i["x"][:, :3].contiguous(), i["x"], i["ws"], i["ref_torso_rgb"], i["ref_bg_rgb"], i["weights_img"], None, None, None,
                      noise_mode="none")
    hparams["htbsr_head_weight_fuse_mode"] = "v2"
    out = out.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=warp_mock.SEED, threshold=np.float64(hparams["htbsr_head_threshold"]),
                        strided=out[:, :, ::4, ::4], corner=out[:, :, :96, :96], tail=out[:, :, -64:, -64:],
                        absmean=np.float64(np.abs(out).mean()))
    print(name, out.shape, float(np.abs(out).mean()))


def render_r512_case(name, Nf):
    """BASELINE configs[1] read literally: a 512x512 NEURAL render (R = 512) with 48 (+48) depth samples on full-size planes.  The
    reference renders all 262 144 rays; the fixture keeps the rays of every 8th row and column (the global couplings -- min / max of
    the valid ray starts, the global depth clamp -- still come from the full render).  Planes, decoder and sampling noise are
    regenerated from their seeds by the tests (50 MB of noise is not stored).  With Nf = 48 the feature image is also sent through
    the reference SuperresolutionHybrid8XDC, whose first step resamples any R != 128 input to 128^2 (superresolution.py:351-355)."""
    R, Nc, seed = 512, 48, 61
    planes = synth.synth_planes(seed, N=1)
    cams = np.asarray([synth.look_at_camera(0.12, -0.07)], np.float32).reshape(-1, 25)
    N, M = 1, R * R
    dec_np = synth.synth_decoder(seed + 1, sigma_bias=3.0)
    dec = make_decoder(dec_np)
    noise_c = synth.synth_noise(seed + 2, (N, M, Nc, 1), stream=7)
    u_f = synth.synth_noise(seed + 2, (N * M, max(Nf, 1)), stream=8)[:, :Nf].copy() if Nf > 0 else np.zeros((N * M, 0), np.float32)
    cam_t = torch.from_numpy(cams)
    with torch.no_grad():
        o, d = RaySampler()(cam_t[:, :16].view(-1, 4, 4), cam_t[:, 16:].view(-1, 3, 3), R)
        ren = ImportanceRenderer(hp={"enable_rescale_plane_regulation": False, "triplane_feature_type": "triplane"}).eval()
        with injected_noise(noise_c, u_f) as st:
            rgb, depth, wsum, valid = ren(torch.from_numpy(planes), dec, o, d, opts(Nc, Nf))
    idx = (np.arange(0, R, 8)[:, None] * R + np.arange(0, R, 8)[None, :]).reshape(-1)
    out = dict(cams=cams, R=R, Nc=Nc, Nf=Nf, seed=seed, ray_index=idx.astype(np.int64),
               rgb=rgb.numpy()[:, idx], depth=depth.numpy()[:, idx], wsum=wsum.numpy()[:, idx], valid=valid.numpy()[:, idx],
               valid_frac=np.float64(valid.float().mean()), depth_min=np.float64(depth.min()), depth_max=np.float64(depth.max()))
    if Nf > 0:
        params = synth.synth_sr_params(seed + 3)
        sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True,
                                       channel_base=32768, channel_max=512, fused_modconv_default="inference_only").eval()
        load_block(sr.block0, params[0]); load_block(sr.block1, params[1])
        feat = rgb.permute(0, 2, 1).reshape(1, 32, R, R).contiguous()
        with torch.no_grad():
            img = sr(feat[:, :3], feat, torch.ones(1, 14, 512), noise_mode="none").numpy()
        out.update(sr_seed=seed + 3, sr_strided=img[:, :, ::4, ::4], sr_corner=img[:, :, :96, :96])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "rgb", rgb.shape, "valid frac", float(valid.float().mean()), "wsum mean", float(wsum.mean()))


def sr_resize_case(name):
    """SuperresolutionHybrid8XDC fed with a neural render that is NOT 128^2: the reference first resamples x and rgb to 128^2 with the
    antialiased bilinear filter (superresolution.py:351-355): one down-sampling (192^2) and one up-sampling (80^2) input."""
    seed = 71
    params = synth.synth_sr_params(seed)
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True,
                                   channel_base=32768, channel_max=512, fused_modconv_default="inference_only").eval()
    load_block(sr.block0, params[0]); load_block(sr.block1, params[1])
    out = dict(seed=seed)
    for tag, r in (("down", 192), ("up", 80)):
        x = synth.hash_unitvar(seed, (1, 32, r, r), stream=3 if tag == "down" else 4)
        with torch.no_grad():
            img = sr(torch.from_numpy(x[:, :3].copy()), torch.from_numpy(x), torch.ones(1, 14, 512), noise_mode="none").numpy()
        out["r_" + tag] = r
        out["strided_" + tag] = img[:, :, ::4, ::4]
        out["corner_" + tag] = img[:, :, :64, :64]
        print(name, tag, img.shape, float(np.abs(img).mean()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def main():
    which = sys.argv[1:] or ["render", "run_model", "sr_small", "sr_full", "synthesis", "fusion", "toplane", "sr_cfg5", "fusion_full",
                             "toplane_full", "synthesis_mask", "warp_sr", "warp_sr_two_stage", "warp_sr_v1", "render_r512", "sr_resize"]
    if "render_r512" in which:
        render_r512_case("render_g_r512_48p0", 0)
        render_r512_case("render_h_r512_48p48_sr", 48)
    if "sr_resize" in which:
        sr_resize_case("sr_resize_a")
    if "warp_sr" in which:
        warp_sr_case("warp_sr_a")
    if "warp_sr_two_stage" in which:
        warp_sr_two_stage_case("warp_sr_two_stage_a")
    if "warp_sr_v1" in which:
        warp_sr_v1_case("warp_sr_v1_a")
    if "sr_cfg5" in which:
        sr_cfg5_case("sr_cfg5_a")
    if "fusion_full" in which:
        fusion_full_case("fusion_full_a")
    if "toplane_full" in which:
        toplane_full_case("toplane_full_a")
    if "synthesis_mask" in which:
        synthesis_mask_case("synthesis_mask_a")
    if "render" in which:
        small = synth.synth_planes(1, N=1, H=32, W=32)
        render_case("render_a_r16_16p16", small, [synth.look_at_camera(0.0, 0.0)], 16, 16, 16, 2, 3)
        small2 = synth.synth_planes(4, N=2, H=32, W=32)
        render_case("render_b_n2_r16_48p48", small2, synth.camera_sweep(2, -0.3, 0.3), 16, 48, 48, 5, 6,
                    sigma_bias=4.0)
        # camera pushed sideways: a large share of rays misses the box (exercises the fix-up, Q2/Q3)
        cam = synth.look_at_camera(0.1, 0.05)
        cam[3] += 0.45
        render_case("render_c_invalid_r16_16p0", small, [cam], 16, 16, 0, 7, 8)
        grid = (synth.hash_unitvar(17, (1, 3, 32 * 3, 24, 24), stream=1)).astype(np.float32)
        render_case("render_f_trigrid_d3_r12_20p12", grid, [synth.look_at_camera(0.15, -0.1)], 12, 20, 12, 14, 15,
                    sigma_bias=2.0, trigrid_depth=3)
        render_case("render_e_white_r12_32p16_bw", small, [synth.look_at_camera(-0.2, 0.15)], 12, 32, 16, 9, 10,
                    box_warp=1.2, white_back=True, sigma_bias=2.0)
        big = synth.synth_planes(11, N=1)
        render_case("render_d_cfg1_r64_16p16", big, [synth.look_at_camera(0.0, 0.0)], 64, 16, 16, 12, 13,
                    store_planes=False, planes_spec=(11, 1, 32, 256, 256, 1.0), sigma_bias=3.0)
    if "run_model" in which:
        run_model_case("run_model_a")
        run_model_case("run_model_b_trigrid_d3", trigrid_depth=3)
    if "sr_small" in which:
        sr_small_case("sr_small_a")
    if "sr_full" in which:
        sr_full_case("sr_full_a")
    if "synthesis" in which:
        synthesis_case("synthesis_ref_a")
    if "fusion" in which:
        fusion_case("fusion_a")
    if "toplane" in which:
        toplane_case("toplane_a")


if __name__ == "__main__":
    main()
