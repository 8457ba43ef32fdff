"""Shared by tests/golden/make_golden.py (reference side, CPU) and tests/test_gpu_warp_sr.py (HIP side): a deterministic stand-in for
the face-vid2vid warp network `torso_model` of SuperresolutionHybrid8XDC_Warp (modules/real3d/facev2v_warp/model2.py; a cold-ish
PyTorch encoder, out of scope here) and the synthetic inputs / parameters of the fusion forward."""
import numpy as np
import torch

from real3dportrait_amd import synth

SEED = 91


class MockTorso(torch.nn.Module):
    """forward(ref_torso_rgb_256, segmap, kp_s, kp_d, rgb_256, weights_256, cal_loss, target_torso_mask) -> (rgb_torso, ret) like
    WarpBasedTorsoModelMediaPipe.forward (torso_model_version 'v2').  The outputs depend on the resized inputs, so the 128 -> 256 and
    512 -> 256 antialiased resizes of sr_with_ref.py:77-82 are part of what the golden pins."""

    def __init__(self, seed=SEED):
        super().__init__()
        self.seed = seed

    def forward(self, ref_torso_rgb_256, segmap, kp_s, kp_d, rgb_256, weights_256=None, cal_loss=True, target_torso_mask=None):
        N, dev = rgb_256.shape[0], rgb_256.device
        t = lambda shape, s, g=1.0: torch.from_numpy(synth.hash_unitvar(self.seed, shape, stream=s) * np.float32(g)).to(dev)
        rgb_torso = t((N, 3, 256, 256), 31, 0.3) + 0.4 * ref_torso_rgb_256 + 0.2 * rgb_256 * weights_256
        hid = t((N, 64, 256, 256), 32)
        occ = torch.from_numpy(synth.synth_noise(self.seed, (N, 1, 64, 64), stream=33)).to(dev)
        return rgb_torso, {"deformed_torso_hid": hid, "occlusion_2": occ}

    # the two-stage entry (WarpBasedTorsoModelMediaPipe.infer_forward_stage1 / _stage2, called by sr_with_ref.py:182,196): stage 1 returns the
    # dict, stage 2 the torso image computed from what stage 1 left in it
    def infer_forward_stage1(self, ref_torso_rgb_256, segmap, kp_s, kp_d, rgb_256, cal_loss=True):
        N, dev = rgb_256.shape[0], rgb_256.device
        rgb_torso, ret = self.forward(ref_torso_rgb_256, segmap, kp_s, kp_d, rgb_256, torch.ones(N, 1, 256, 256, device=dev) * 0.5)
        ret["_rgb_torso"] = rgb_torso
        return ret

    def infer_forward_stage2(self, ret):
        return ret["_rgb_torso"]


def warp_inputs(seed=SEED, N=1):
    t = lambda shape, s, g=1.0: synth.hash_unitvar(seed, shape, stream=s) * np.float32(g)
    return {
        "x": t((N, 32, 128, 128), 1), "ref_torso_rgb": t((N, 3, 512, 512), 3, 0.5), "ref_bg_rgb": t((N, 3, 512, 512), 4, 0.5),
        "weights_img": synth.synth_noise(seed, (N, 1, 128, 128), stream=5),
        "ws": (np.ones((N, 14, 512), np.float32) + synth.hash_unitvar(seed, (N, 14, 512), stream=6) * np.float32(0.1)),
    }


def load_warp_params(sr, load_block, seed=SEED, to=lambda a: torch.from_numpy(np.ascontiguousarray(a))):
    """Deterministic parameters for block0 / block1 / head_torso_block and the four conv stacks (module attribute names of
    sr_with_ref.py:24-63), for the reference module and for ours alike."""
    load_block(sr.block0, synth.synth_sr_block(seed, 32, 256, 512, 100))
    load_block(sr.block1, synth.synth_sr_block(seed, 256, 128, 512, 200))
    if hasattr(sr, "head_torso_block"):            # (the reference builds it and fuse_head_torso_convs for fuse modes v2 / v3 only, sr_with_ref.py:36-55)
        load_block(sr.head_torso_block, synth.synth_sr_block(seed, 256, 256, 512, 400))
    with torch.no_grad():
        for i, (name, plan) in enumerate(synth.FUSION_STACKS.items()):
            if not hasattr(sr, name):
                continue
            convs = [m for m in getattr(sr, name) if hasattr(m, "weight")]
            for m, (w, b) in zip(convs, synth.synth_conv_stack(seed, plan, 300 + 20 * i)):
                m.weight.copy_(to(w)); m.bias.copy_(to(b))
