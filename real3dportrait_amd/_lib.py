"""ctypes binding of libr3d_hip.so (the C ABI declared in include/r3d_hip.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing an
operator raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C real3dportrait_amd/csrc`.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("R3D_LIB") or os.path.join(_HERE, "lib", "libr3d_hip.so")     # R3D_LIB: an experiment build (scripts/)
CSRC = os.path.join(_HERE, "csrc")

c_void_p, c_int, c_float, c_size_t, c_uint64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_size_t, ctypes.c_uint64)
P = c_void_p

# name -> (restype, argtypes); mirrors include/r3d_hip.h one to one
SIGNATURES = {
    "r3d_version": (c_int, []),
    "r3d_last_error": (ctypes.c_char_p, []),
    "r3d_planes_to_nhwc": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, ctypes.POINTER(c_int), P]),
    "r3d_planes_absmax_partials": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "r3d_conv_forward_cat": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, c_float, c_float, c_float,
                                     P, c_int, c_int, c_int, P, c_int, P, c_size_t, P, c_size_t, P]),
    "r3d_conv_forward_blend": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_float, c_float, c_float,
                                       P, c_int, P, c_size_t, P, P]),
    "r3d_blend_cat_to_split": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, P, c_int, P, c_size_t, P]),
    "r3d_upsample2x_bilinear": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P, c_size_t, P]),
    "r3d_raygen": (c_int, [P, P, c_int, c_int, P, P, P]),
    "r3d_render_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "r3d_render_forward": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_int,
                                   P, P, c_uint64, P, c_int, P, P, P, P, c_int, P, P, P, P, c_size_t, P, c_size_t, P]),
    "r3d_run_model": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, c_float, P, P, P, c_int, P, c_size_t, P]),
    "r3d_run_model_workspace_bytes": (c_size_t, []),
    "r3d_sr_block_prepacked_bytes": (c_size_t, [c_int, c_int]),
    "r3d_sr_block_styles_bytes": (c_size_t, [c_int, c_int, c_int]),
    "r3d_sr_block_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "r3d_sr_block_prepack": (c_int, [c_int, c_int, P, P, P, c_int, P]),
    "r3d_sr_block_styles": (c_int, [P, c_int, c_int, c_int, c_int] + [P] * 12 + [P, P]),
    "r3d_sr_block_forward": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_float, P, c_int, P, c_size_t,
                                     P, P, P, c_int, P, c_size_t, P]),
    "r3d_chain_fold": (c_int, [P, c_int, c_int, P, c_int, P, c_int, P]),
    "r3d_sr_block_bound_offset": (c_size_t, [c_int, c_int]),
    "r3d_conv_scales_bound_offset": (c_size_t, [c_int, c_int]),
    "r3d_conv_scales_bytes": (c_size_t, [c_int, c_int, c_int]),
    "r3d_absmax": (c_int, [P, c_size_t, c_int, P, P, P]),
    "r3d_conv_prepacked_bytes": (c_size_t, [c_int, c_int, c_int]),
    "r3d_conv_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "r3d_conv_prepack": (c_int, [P, c_int, c_int, c_int, P, P]),
    "r3d_conv_forward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int,
                                 c_int, c_float, c_float, c_float, P, c_int, P, c_size_t, P, P, c_size_t, P]),
    "r3d_frames_to_u8": (c_int, [P, c_int, c_int, c_int, P, P]),
    "r3d_resize_bilinear": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, c_int, P]),
    "r3d_blend": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "r3d_person_occlusion": (c_int, [P, P, c_float, c_size_t, P, P]),
    "r3d_comm_unique_id": (c_int, [P]),
    "r3d_comm_init": (c_int, [P, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "r3d_comm_destroy": (c_int, [P]),
    "r3d_gather_frames": (c_int, [P, P, c_size_t, P, c_int, P]),
    "r3d_profile_configure": (c_int, [ctypes.c_uint32]),
    "r3d_profile_reset": (c_int, []),
    "r3d_profile_read": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]),
    "r3d_profile_clock": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ulonglong)]),
    "r3d_event_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "r3d_event_record": (c_int, [P, P]),
    "r3d_event_elapsed_ms": (c_int, [P, P, ctypes.POINTER(c_float)]),
    "r3d_event_destroy": (c_int, [P]),
}



class ChainOp(ctypes.Structure):
    """r3d_chain_op of include/r3d_hip.h (one layer of an r3d_chain_fold chain)."""
    _fields_ = [("kind", c_int), ("Cin", c_int), ("Cout", c_int), ("ksize", c_int), ("act", c_int),
                ("gain", c_float), ("clamp", c_float), ("src_a", c_int), ("src_b", c_int),
                ("scales", c_void_p), ("prepacked", c_void_p), ("bias", c_void_p)]


CHAIN_SR_BLOCK, CHAIN_CONV, CHAIN_SR_BLOCK_TAIL, CHAIN_SRC_NONE, CHAIN_MAX_OPS, CHAIN_MAX_EXT, CHAIN_MAX_ZERO = 0, 1, 2, -1000, 12, 4, 4

ABI_VERSION = 61          # r3d_version() of the library this table mirrors (include/r3d_hip.h)
_lib = None


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def load():
    """Load libr3d_hip.so; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of the process (one runtime, one
    # set of streams/devices), so make sure it is mapped before our library asks the loader for that SONAME.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "real3dportrait_amd: HIP extension %s not built -- run `make -C %s` (there is no CPU/eager fallback)"
            % (LIB_PATH, CSRC))
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError("real3dportrait_amd: cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the ABI and this table drift apart
        fn.restype, fn.argtypes = res, args
    if lib.r3d_version() != ABI_VERSION:      # a stale in-tree build (real3dportrait_amd/lib/ is not tracked): arguments would be shifted silently
        raise RuntimeError("real3dportrait_amd: %s reports ABI %d, this package binds ABI %d -- rebuild it (`make -C %s`)"
                           % (LIB_PATH, lib.r3d_version(), ABI_VERSION, CSRC))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().r3d_last_error()
        raise RuntimeError("libr3d_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Raw device pointer of a contiguous fp32/uint8/bool CUDA(HIP) tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda, "libr3d_hip operates on device tensors only"
    assert t.is_contiguous()
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
