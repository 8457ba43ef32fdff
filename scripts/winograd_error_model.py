"""numpy model: error of F(2,3) Winograd (along x) against the direct 3x3 conv, both in the f16x3 / f16mx arithmetic of r3d_sr_f16x3.hip, vs fp64.
One output row of a 3x3 conv with C input channels: K = 9 C products per output direct, 6 C transformed products per output with F(2,3).
Input transform in fp32 (v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3) BEFORE the hi / lo split; weights transformed in fp64 at "prepack"
(u0 = g0, u1 = (g0 + g1 + g2) / 2, u2 = (g0 - g1 + g2) / 2, u3 = g2) and row-normalised like the direct weights; output transform in fp32
(y0 = m0 + m1 + m2, y1 = m1 - m2 - m3).  Usage: python scripts/winograd_error_model.py > profiles/r05/winograd_error_model.txt"""
import numpy as np
from mx_format_model import e4m3, e5m2, split_f16


def products(xh, xl, wh, wl, mode):
    """sum over K of x * w in the two shipped arithmetics (fp32 accumulate modelled in fp64: the accumulation error is common to both forms)"""
    main = xh @ wh
    if mode == "f16x3":
        return main + xh @ wl + xl @ wh
    return main + (e5m2(xh) @ e4m3(wl * 2.0 ** 8) + e5m2(xl * 2.0 ** 11) @ e4m3(wh * 2.0 ** -3)) * 2.0 ** -8


def main():
    rng = np.random.default_rng(1)
    C, W = 256, 130                                        # channels, input columns (128 outputs)
    print("# err = max|y - y_fp64| / max|y_fp64| over 128 outputs x 64 couts of one row; C = %d (K = %d direct, %d transformed per output)" % (C, 9 * C, 6 * C))
    print("# case                      f16x3 direct  f16x3 F(2,3)  f16mx direct  f16mx F(2,3)")
    for name, spike in (("gaussian", 0), ("spike 2^6", 6), ("spike 2^10", 10), ("spike 2^14", 14), ("lrelu (one-sided)", -1)):
        x = rng.standard_normal((3, C, W))
        if spike > 0:
            x[1, 7, 40] = 2.0 ** spike
        if spike < 0:
            x = np.where(x > 0, x, 0.2 * x) * 1.4
        g = rng.standard_normal((64, 3, C, 3)) / np.sqrt(9 * C)         # [cout][ky][c][kx]
        bound = 2.0 ** np.ceil(np.log2(np.abs(x).max()))
        # ---- direct: operand scaled so that |x| < 2^15 (the fold), weight rows normalised to [2^10, 2^11)
        xs = x * (2.0 ** 15 / bound) * 0.999
        ref = np.zeros((64, W - 2))
        for kx in range(3):
            ref += np.einsum("okc,kcw->ow", g[..., kx], x[:, :, kx:kx + W - 2])
        out = {}
        for mode in ("f16x3", "f16mx"):
            y = np.zeros((64, W - 2))
            for co in range(64):
                wrow = g[co]                                           # [ky][c][kx]
                kw = 10 - np.floor(np.log2(np.abs(wrow).max()))
                wh, wl = split_f16(wrow * 2.0 ** kw)
                acc = np.zeros(W - 2)
                xh, xl = split_f16(xs.astype(np.float32).astype(np.float64))
                for kx in range(3):
                    A_h = xh[:, :, kx:kx + W - 2].reshape(3 * C, -1).T; A_l = xl[:, :, kx:kx + W - 2].reshape(3 * C, -1).T
                    acc += products(A_h, A_l, wh[..., kx].reshape(-1), wl[..., kx].reshape(-1), mode)
                y[co] = acc * 2.0 ** -kw * (bound / 2.0 ** 15 / 0.999)
            out[(mode, "direct")] = np.abs(y - ref).max() / np.abs(ref).max()
        # ---- F(2,3): the transformed operand can reach 2 |x|: one more bit of head room in the fold
        xs = x * (2.0 ** 14 / bound) * 0.999
        d = xs.astype(np.float32).astype(np.float64)
        P = (W - 2) // 2
        d0, d1, d2, d3 = d[:, :, 0:2 * P:2], d[:, :, 1:2 * P + 1:2], d[:, :, 2:2 * P + 2:2], d[:, :, 3:2 * P + 3:2]
        V = [(d0 - d2), (d1 + d2), (d2 - d1), (d1 - d3)]
        V = [v.astype(np.float32).astype(np.float64) for v in V]
        for mode in ("f16x3", "f16mx"):
            y = np.zeros((64, W - 2))
            for co in range(64):
                gg = g[co]                                             # [ky][c][kx]
                U = [gg[..., 0], (gg[..., 0] + gg[..., 1] + gg[..., 2]) / 2, (gg[..., 0] - gg[..., 1] + gg[..., 2]) / 2, gg[..., 2]]
                kw = 10 - np.floor(np.log2(max(np.abs(u).max() for u in U)))
                M = []
                for p in range(4):
                    uh, ul = split_f16(U[p] * 2.0 ** kw)
                    vh, vl = split_f16(V[p])
                    M.append(products(vh.reshape(3 * C, -1).T, vl.reshape(3 * C, -1).T, uh.reshape(-1), ul.reshape(-1), mode).astype(np.float32).astype(np.float64))
                y[co, 0:2 * P:2] = (M[0] + M[1] + M[2]) * 2.0 ** -kw * (bound / 2.0 ** 14 / 0.999)
                y[co, 1:2 * P:2] = (M[1] - M[2] - M[3]) * 2.0 ** -kw * (bound / 2.0 ** 14 / 0.999)
            out[(mode, "wino")] = np.abs(y - ref).max() / np.abs(ref).max()
        print("  %-24s  %.2e      %.2e      %.2e      %.2e" % (name, out[("f16x3", "direct")], out[("f16x3", "wino")], out[("f16mx", "direct")], out[("f16mx", "wino")]))


if __name__ == "__main__":
    main()
