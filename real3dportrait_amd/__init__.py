"""MI355X-native tri-plane NeRF volume renderer + super-resolution for Real3D-Portrait's per-frame path.

Operators keep the reference's names and signatures (see each module's docstring for file:line):
    RaySampler, ImportanceRenderer, OSGDecoder          (volumetric_rendering.py)
    SynthesisBlock, SuperresolutionHybrid8XDC           (superresolution.py)
    SynthesisBlockNoUp, Conv2d, ConvStack                (superresolution.py: torso/background fusion convs)
    TriPlaneGenerator (.synthesis contract), patch_model (triplane.py)
    render_clip_sharded                                  (frames.py: frame sharding + RCCL gather)
All compute goes through libr3d_hip.so (include/r3d_hip.h); there is no eager/CPU fallback.
"""
__version__ = "0.1.0"


def __getattr__(name):      # lazy: importing the package (e.g. for synth) must not require torch+GPU
    if name in ("RaySampler", "ImportanceRenderer", "OSGDecoder"):
        from . import volumetric_rendering as m
        return getattr(m, name)
    if name in ("SynthesisBlock", "SuperresolutionHybrid8XDC", "SynthesisBlockNoUp", "Conv2d", "ConvStack"):
        from . import superresolution as m
        return getattr(m, name)
    if name in ("TriPlaneGenerator", "patch_model"):
        from . import triplane as m
        return getattr(m, name)
    if name in ("render_clip_sharded", "shard_frames"):
        from . import frames as m
        return getattr(m, name)
    raise AttributeError(name)
