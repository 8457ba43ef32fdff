"""CPU tests of the host mirror of the kernel's counter-based sampling noise (real3dportrait_amd/synth.py: device_hash_uniform,
render_hash_noise).  The benchmarked configuration renders with noise_mode='hash' (frames.ClipRenderer), so the oracle can only check
it if the host can produce the same jitter: here the mirror is held against the DEVICE SOURCE itself -- mix32 / hash_uniform are cut out
of csrc/r3d_common.h and compiled for the host with gcc (integer arithmetic and one exact int -> float conversion: no target dependence) --
and its statistics are checked; tests/test_gpu_pinned_config.py holds it against the kernel on the GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np

from real3dportrait_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_build_of_device_hash(tmp_path):
    src = open(os.path.join(ROOT, "real3dportrait_amd", "csrc", "r3d_common.h")).read()
    m = re.search(r"(__device__ __forceinline__ uint32_t mix32.*?\n}\n)(__device__ __forceinline__ float hash_uniform.*?\n}\n)", src, re.S)
    assert m, "mix32 / hash_uniform not found in r3d_common.h"
    c = ("#include <stdint.h>\n#define __device__\n#define __forceinline__ static inline\n" + m.group(1) + m.group(2) +
         "void fill(uint64_t seed, uint32_t stream, const uint64_t* idx, long n, float* out)"
         "{ for (long i = 0; i < n; ++i) out[i] = hash_uniform(seed, stream, idx[i]); }\n")
    cfile, so = os.path.join(tmp_path, "devhash.c"), os.path.join(tmp_path, "devhash.so")
    open(cfile, "w").write(c)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, cfile])
    return ctypes.CDLL(so)


def test_mirror_equals_the_device_source(tmp_path):
    lib = _host_build_of_device_hash(str(tmp_path))
    rng = np.random.default_rng(1)
    idx = np.concatenate([np.arange(5000, dtype=np.uint64), rng.integers(0, 2 ** 63, 5000, dtype=np.uint64) * np.uint64(2) + np.uint64(1),
                          np.array([2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 64 - 1], dtype=np.uint64)])
    for seed in (0, 1, 0xFFFFFFFF, 0x123456789ABCDEF0, 2 ** 64 - 1):
        for stream in (0, 1, 7):
            out = np.empty(idx.size, np.float32)
            lib.fill(ctypes.c_uint64(seed), ctypes.c_uint32(stream), idx.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(idx.size),
                     out.ctypes.data_as(ctypes.c_void_p))
            assert np.array_equal(out, synth.device_hash_uniform(seed, stream, idx)), (seed, stream)


def test_jitter_statistics():
    """U[0,1) sanity of what the headline configuration samples with: range, mean, variance, per-sample-slot and per-ray means, 16-bin
    uniformity (chi^2), independence of the two streams and of neighbouring rays / frames."""
    from real3dportrait_amd.frames import frame_seed
    M, Nc, Nf = 128 * 128, 48, 48
    nc, uf = synth.render_hash_noise(frame_seed(0, 3), np.arange(M), Nc, Nf)
    for a in (nc, uf):
        assert a.dtype == np.float32 and a.min() >= 0.0 and a.max() < 1.0
        assert abs(a.mean() - 0.5) < 1.5e-3 and abs(a.var() - 1.0 / 12.0) < 1e-3
        assert np.abs(a.mean(axis=0) - 0.5).max() < 0.012          # every sample slot: sigma = 0.29 / sqrt(16384) = 2.3e-3
        assert np.abs(a.mean(axis=1) - 0.5).max() < 0.25           # every ray: sigma = 0.29 / sqrt(48) = 0.042
        hist = np.bincount((a.reshape(-1) * 16).astype(np.int64), minlength=16).astype(np.float64)
        chi2 = ((hist - a.size / 16.0) ** 2 / (a.size / 16.0)).sum()
        assert chi2 < 50.0, chi2                                   # 15 degrees of freedom: P(chi2 > 50) ~ 1e-5
        assert abs(np.corrcoef(a[:-1].reshape(-1), a[1:].reshape(-1))[0, 1]) < 5e-3       # neighbouring rays
        assert abs(np.corrcoef(a[:, :-1].reshape(-1), a[:, 1:].reshape(-1))[0, 1]) < 5e-3   # neighbouring samples
    assert abs(np.corrcoef(nc.reshape(-1), uf.reshape(-1))[0, 1]) < 5e-3                  # the two draws of a frame
    nc2, _ = synth.render_hash_noise(frame_seed(0, 4), np.arange(M), Nc, Nf)              # the next frame
    assert abs(np.corrcoef(nc.reshape(-1), nc2.reshape(-1))[0, 1]) < 5e-3 and not np.array_equal(nc, nc2)


def test_noise_is_a_function_of_the_global_ray_index():
    """A ray's jitter does not depend on which rays are asked for with it (what makes a frame independent of its shard)."""
    a, b = synth.render_hash_noise(99, np.arange(1000), 48, 48)
    sub = np.array([3, 999, 17, 500])
    a2, b2 = synth.render_hash_noise(99, sub, 48, 48)
    assert np.array_equal(a[sub], a2) and np.array_equal(b[sub], b2)
    a0, b0 = synth.render_hash_noise(99, sub, 48, 0)
    assert b0.shape == (4, 0) and np.array_equal(a0, a2)
