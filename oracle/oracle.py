"""ctypes binding of oracle/r3d_oracle.c (numpy in / numpy out).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libr3d_oracle.so")


def build_oracle(force=False):
    """Compile r3d_oracle.c with gcc (seconds).  Returns the .so path."""
    src = os.path.join(_HERE, "r3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Oracle:
    def __init__(self):
        self.lib = ctypes.CDLL(build_oracle())
        self.lib.r3d_oracle_version.restype = ctypes.c_int
        self.lib.r3d_oracle_num_threads.restype = ctypes.c_int
        self.lib.r3d_oracle_render.restype = ctypes.c_int
        self.lib.r3d_oracle_sr_block.restype = ctypes.c_int

    @property
    def num_threads(self):
        return int(self.lib.r3d_oracle_num_threads())

    def raygen(self, c2w, K, R):
        c2w, K = _f(c2w).reshape(-1, 16), _f(K).reshape(-1, 9)
        N = c2w.shape[0]
        o = np.empty((N, R * R, 3), np.float32)
        d = np.empty((N, R * R, 3), np.float32)
        self.lib.r3d_oracle_raygen(_p(c2w), _p(K), N, R, _p(o), _p(d))
        return o, d

    def ray_limits(self, origins, dirs, box_warp):
        o, d = _f(origins), _f(dirs)
        n = o.size // 3
        rs, re = np.empty(n, np.float32), np.empty(n, np.float32)
        v = np.empty(n, np.uint8)
        self.lib.r3d_oracle_ray_limits(_p(o), _p(d), n, ctypes.c_float(box_warp), _p(rs), _p(re), _p(v))
        shp = o.shape[:-1]
        return rs.reshape(shp), re.reshape(shp), v.reshape(shp).astype(bool)

    def _set_depth(self, planes, triplane_depth):
        """planes [N,3,C*D,H,W]; returns the feature count C the C functions take."""
        self.lib.r3d_oracle_set_triplane_depth(int(triplane_depth))
        assert planes.shape[2] % triplane_depth == 0
        return planes.shape[2] // triplane_depth

    def run_model(self, planes, dec, coords, box_warp=1.0, triplane_depth=1):
        planes, coords = _f(planes), _f(coords)
        N, _, _, H, W = planes.shape
        C = self._set_depth(planes, triplane_depth)
        npts = coords.shape[1]
        w1, b1, w2, b2 = (_f(x) for x in dec)
        HID, OUT = w1.shape[0], w2.shape[0]
        rgb = np.empty((N, npts, OUT - 1), np.float32)
        sig = np.empty((N, npts, 1), np.float32)
        self.lib.r3d_oracle_run_model(_p(planes), N, C, H, W, _p(w1), _p(b1), _p(w2), _p(b2), HID, OUT,
                                      ctypes.c_float(box_warp), _p(coords), npts, _p(rgb), _p(sig))
        self.lib.r3d_oracle_set_triplane_depth(1)
        return rgb, sig

    def render(self, planes, dec, origins, dirs, Nc, Nf, noise_c, u_f, box_warp=1.0, white_back=False,
               debug=False, triplane_depth=1):
        planes, o, d = _f(planes), _f(origins), _f(dirs)
        N, _, _, H, W = planes.shape
        C = self._set_depth(planes, triplane_depth)
        M = o.shape[1]
        w1, b1, w2, b2 = (_f(x) for x in dec)
        HID, OUT = w1.shape[0], w2.shape[0]
        noise_c = _f(noise_c).reshape(-1)
        u_f = _f(u_f).reshape(-1) if Nf > 0 else np.zeros(1, np.float32)
        assert noise_c.size == N * M * Nc
        rgb = np.empty((N, M, OUT - 1), np.float32)
        depth = np.empty((N, M, 1), np.float32)
        wsum = np.empty((N, M, 1), np.float32)
        valid = np.empty((N, M, 1), np.uint8)
        dc = np.empty((N, M, Nc), np.float32) if debug else None
        df = np.empty((N, M, max(Nf, 1)), np.float32) if debug else None
        sc = np.empty((N, M, Nc), np.float32) if debug else None
        rc = self.lib.r3d_oracle_render(_p(planes), N, C, H, W, _p(w1), _p(b1), _p(w2), _p(b2), HID, OUT,
                                        _p(o), _p(d), M, Nc, Nf, ctypes.c_float(box_warp), int(white_back),
                                        _p(noise_c), _p(u_f), _p(rgb), _p(depth), _p(wsum), _p(valid),
                                        _p(dc), _p(df), _p(sc))
        self.lib.r3d_oracle_set_triplane_depth(1)
        if rc != 0:
            raise RuntimeError("r3d_oracle_render failed rc=%d" % rc)
        out = (rgb, depth, wsum, valid.astype(bool))
        if debug:
            return out + ({"depths_coarse": dc, "depths_fine": df[..., :Nf], "sigma_coarse": sc},)
        return out

    def conv2d(self, x, w, b=None, slope=None):
        """torch.nn.Conv2d(k in {1,3}, stride 1, padding k//2) [+ LeakyReLU(slope)] for one image [Ci,H,W]."""
        x, w = _f(x), _f(w)
        b = None if b is None else _f(b)
        Ci, H, W = x.shape
        Co, k = w.shape[0], w.shape[-1]
        y = np.empty((Co, H, W), np.float32)
        rc = self.lib.r3d_oracle_conv2d(_p(x), Ci, H, W, _p(w), _p(b) if b is not None else None, Co, k,
                                        ctypes.c_float(-1.0 if slope is None else slope), _p(y))
        if rc != 0:
            raise RuntimeError("r3d_oracle_conv2d failed")
        return y

    def upsample2x_bilinear(self, x):
        x = _f(x)
        C, H, W = x.shape
        y = np.empty((C, 2 * H, 2 * W), np.float32)
        self.lib.r3d_oracle_upsample2x_bilinear(_p(x), C, H, W, _p(y))
        return y

    def to_plane_cnn(self, x, plan, params, up_before):
        """SegFormerSECC2PlaneBackbone.to_plane_cnn + the view/flip/stack of forward (segformer.py:691-700,721-729):
        x [256,h,w] -> secc planes [3,32,2h,2w]."""
        for i, ((ci, co, k, lrelu), (w, b)) in enumerate(zip(plan, params)):
            if i == up_before:
                x = self.upsample2x_bilinear(x)
            x = self.conv2d(x, w, b, 0.01 if lrelu else None)
        p = x.reshape(3, -1, x.shape[-2], x.shape[-1])
        return np.stack([p[0][:, ::-1, :], p[1][:, ::-1, :], p[2][:, ::-1, ::-1]]).astype(np.float32)

    def conv_stack(self, x, plan, params):
        """A FUSION_STACKS plan [(ci, co, k, lrelu)] with params [(w, b)]; torch.nn.LeakyReLU() default slope 0.01."""
        for (ci, co, k, lrelu), (w, b) in zip(plan, params):
            x = self.conv2d(x, w, b, 0.01 if lrelu else None)
        return x

    def sr_block(self, x, img, params, ws3, clamp=None, up=True):
        """params: dict with conv0/conv1/torgb -> (weight, bias, affine_w, affine_b).  up=False: SynthesisBlockNoUp."""
        x, img, ws3 = _f(x), _f(img), _f(ws3)
        Ci, H, W = x.shape
        Co = params["conv1"][0].shape[0]
        WD = ws3.shape[-1]
        m = 2 if up else 1
        xo = np.empty((Co, m * H, m * W), np.float32)
        io = np.empty((3, m * H, m * W), np.float32)
        args = []
        for k in ("conv0", "conv1", "torgb"):
            args += [_f(t) for t in params[k]]
        keep = args
        fn = self.lib.r3d_oracle_sr_block if up else self.lib.r3d_oracle_sr_block_noup
        rc = fn(_p(x), _p(img), Ci, Co, H, W, WD, _p(ws3),
                                          *[_p(a) for a in keep],
                                          ctypes.c_float(-1.0 if clamp is None else clamp), _p(xo), _p(io))
        if rc != 0:
            raise RuntimeError("r3d_oracle_sr_block failed")
        return xo, io

    def superresolution(self, rgb, x, blocks, ws):
        """SuperresolutionHybrid8XDC.forward (superresolution.py:348-359) for one image:
        ws[-1] repeated 3x (:349); block0 then block1.  blocks = [params0, params1]."""
        ws3 = np.repeat(_f(ws)[-1:, :], 3, axis=0)
        for p in blocks:
            x, rgb = self.sr_block(x, rgb, p, ws3)
        return rgb
