"""CPU: the C-ABI shared library loads and exports every symbol include/r3d_hip.h declares; argument
validation (which runs before any HIP call) reports errors through return codes + r3d_last_error."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "r3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(r3d_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    from real3dportrait_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libr3d_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "python binding table misses %s" % s
    assert set(_lib.SIGNATURES) == set(syms)


def test_version_and_sizes():
    from real3dportrait_amd import _lib
    lib = _lib.load()
    assert lib.r3d_version() >= 10
    assert lib.r3d_render_workspace_bytes(1, 16384, 48, 48) >= 2 * 16384 * 4
    assert lib.r3d_sr_block_prepacked_bytes(32, 256) == (2 * 9 * 32 * 256 + 9 * 256 * 256 + 4 * 256) * 4   # conv0 in two layouts + conv1 + 2 weight-row tails
    assert lib.r3d_conv_prepacked_bytes(3, 64, 3) == (9 * 16 * 128 + 2 * 128) * 4            # padded to 16 x 128 + tail
    assert lib.r3d_conv_scales_bytes(2, 3, 64) >= 2 * (16 + 128 + 3) * 4
    assert lib.r3d_sr_block_bound_offset(32, 256) * 4 < lib.r3d_sr_block_styles_bytes(1, 32, 256)
    assert lib.r3d_sr_block_styles_bytes(2, 32, 256) > 2 * (32 + 4 * 256) * 4
    assert lib.r3d_sr_block_workspace_bytes(1, 32, 256, 128, 128) > 256 * 257 * 257 * 4


def test_argument_errors_are_reported_not_thrown():
    from real3dportrait_amd import _lib
    lib = _lib.load()
    rc = lib.r3d_raygen(None, None, 1, 16, None, None, None)
    assert rc == -1 and b"raygen" in lib.r3d_last_error()
    one = ctypes.c_void_p(64)      # never dereferenced: validation fails first
    rc = lib.r3d_render_forward(one, 1, 32, 32, 1, one, one, one, one, one, one, 256, 200, 0, 1.0, 0, None, None, 0,
                                one, 0, one, one, one, one, 1 << 20, None)
    assert rc == -1 and b"depth_resolution" in lib.r3d_last_error()
    rc = lib.r3d_render_forward(one, 1, 32, 32, 1, one, one, one, one, one, one, 256, 48, 48, 1.0, 0, None, None, 0,
                                one, 1, None, one, one, one, 8, None)
    assert rc == -2 and b"workspace" in lib.r3d_last_error()
    rc = lib.r3d_sr_block_prepack(30, 256, one, one, one, 1, None)
    assert rc == -1 and b"multiple of 8" in lib.r3d_last_error()
    rc2 = lib.r3d_sr_block_forward(one, one, 1, 32, 256, 16, 16, 1, one, 2, one, -1.0, None, -1, None, 0, one, None, None, 0, one, 1 << 40, None)
    assert rc2 == -1 and b"format" in lib.r3d_last_error()      # SPLIT input needs the f16x3 precision
    rc2 = lib.r3d_sr_block_forward(one, one, 1, 32, 256, 16, 16, 0, one, 0, one, -1.0, None, -1, None, 0, one, None, None, 0, one, 1 << 40, None)
    assert rc2 == -1 and b"up=0" in lib.r3d_last_error()        # SynthesisBlockNoUp is f16x3 only
    rc2 = lib.r3d_sr_block_forward(one, one, 1, 32, 256, 16, 16, 1, one, 0, one, -1.0, None, -1, None, 0, one, one, None, 0, one, 1 << 40, None)
    assert rc2 == -1 and b"uint8" in lib.r3d_last_error()       # the fused uint8 output is f16x3 only
    rc2 = lib.r3d_conv_forward(one, one, None, 1, 64, 30, 16, 16, 3, one, 0, 0, 0.0, 1.0, -1.0, one, 0, None, 0, None, one, 1 << 40, None)
    assert rc2 == -1 and b"multiple of 4" in lib.r3d_last_error()
    rc2 = lib.r3d_conv_forward(one, one, None, 1, 64, 32, 16, 16, 5, one, 0, 0, 0.0, 1.0, -1.0, one, 0, None, 0, None, one, 1 << 40, None)
    assert rc2 == -1 and b"bad argument" in lib.r3d_last_error()
    rc2 = lib.r3d_conv_forward(one, one, None, 1, 64, 32, 16, 16, 3, one, 0, 0, 0.0, 1.0, -1.0, one, 0, None, 0, None, one, 8, None)
    assert rc2 == -2 and b"workspace" in lib.r3d_last_error()
    # chain fold: a layer may only read the bound of an EARLIER op or of an external slot that exists
    op = _lib.ChainOp(kind=_lib.CHAIN_CONV, Cin=16, Cout=128, ksize=3, act=0, gain=1.0, clamp=-1.0, src_a=0, src_b=_lib.CHAIN_SRC_NONE,
                      scales=64, prepacked=64, bias=None)
    arr = (_lib.ChainOp * 1)(op)
    rc2 = lib.r3d_chain_fold(ctypes.cast(arr, ctypes.c_void_p), 1, 1, None, 0, None, 0, None)
    assert rc2 == -1 and b"bound source" in lib.r3d_last_error()
    arr[0].src_a = -1
    rc2 = lib.r3d_chain_fold(ctypes.cast(arr, ctypes.c_void_p), 1, 1, None, 0, None, 0, None)
    assert rc2 == -1 and b"bound source" in lib.r3d_last_error()
    rc2 = lib.r3d_absmax(None, 16, 1, one, None, None)
    assert rc2 == -1 and b"absmax" in lib.r3d_last_error()
    rc = lib.r3d_sr_block_prepack(30, 256, one, one, one, 1, None)
    try:
        _lib.check(rc, "prepack")
        assert False
    except RuntimeError as e:
        assert "multiple of 8" in str(e)


def test_product_has_no_oracle_dependency():
    """The oracle is test infrastructure: nothing under real3dportrait_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "real3dportrait_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "r3d_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
