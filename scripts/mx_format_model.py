"""numpy model of the f16mx correction records (r3d_sr_f16x3.hip, "f16mx"): error of a K = 9 x 256 dot product whose cross terms
xh*wl + xl*wh go through 8-bit records, as a function of how far the tensor's bound (the one exponent the records carry) sits above the
typical activation.  Three activation-record formats:
    e4m3/tensor   round 4: xh8 = e4m3(hi * 2^-7), xl8 = e4m3(lo * 2^4)  (one exponent per tensor: the range fold's bound 2^15)
    e4m3/group    the MX format proper: e4m3 with one E8M0 scale per (pixel, 16 channels), scale = the group's max exponent
    e5m2          round 5: xh8 = e5m2(hi), xl8 = e5m2(lo * 2^11)
Weights: e4m3 of the row-normalised hi / lo (unchanged).  Run on the CPU: python scripts/mx_format_model.py > profiles/r05/mx_format_model.txt
"""
import numpy as np


def quant(v, mbits, emin, vmax):
    """round-to-nearest-even to a float format with `mbits` stored mantissa bits, minimum normal exponent emin, saturating at vmax"""
    v = np.asarray(v, np.float64)
    a = np.abs(v)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.maximum(e, emin)
    q = np.ldexp(np.rint(np.ldexp(a, (mbits - e).astype(int))), (e - mbits).astype(int))
    return np.sign(v) * np.minimum(q, vmax)


def e4m3(v): return quant(v, 3, -6, 448.0)
def e5m2(v): return quant(v, 2, -14, 57344.0)


def split_f16(v):
    hi = v.astype(np.float16).astype(np.float64)
    lo = (v - hi).astype(np.float16).astype(np.float64)
    return hi, lo


def main():
    rng = np.random.default_rng(0)
    P, K = 4096, 9 * 256
    w = rng.standard_normal((K,)) / np.sqrt(K)
    kw = 10 - np.floor(np.log2(np.abs(w).max()))                  # the row stored as w * 2^kw, max in [2^10, 2^11)
    ws = w * 2.0 ** kw
    wh, wl = split_f16(ws)
    wl8, wh8 = e4m3(wl * 2.0 ** 8), e4m3(wh * 2.0 ** -3)
    print("# err = max|y - y_fp64| / max|y_fp64| over %d typical pixels; K = %d; x ~ N(0,1) * 2^(15 - gap): the tensor's bound is 2^gap above sigma" % (P, K))
    print("# gap   f16x3      e4m3/tensor  e4m3/group   e5m2")
    for gap in (3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25):
        x = rng.standard_normal((P, K)) * 2.0 ** (15 - gap)
        xh, xl = split_f16(x)
        ref = x @ ws
        main_ = xh @ wh
        y3 = main_ + xh @ wl + xl @ wh
        a = (e4m3(xh * 2.0 ** -7) @ wl8 + e4m3(xl * 2.0 ** 4) @ wh8) * 0.5
        g = x.reshape(P, K // 16, 16)
        sc = 2.0 ** (np.floor(np.log2(np.abs(g).max(-1, keepdims=True))) - 8)            # group max -> [2^8, 2^9) < 448
        gh, gl = xh.reshape(g.shape), xl.reshape(g.shape)
        b = ((e4m3(gh / sc) * sc).reshape(P, K) @ wl8 * 2.0 ** -8 + (e4m3(gl * 2.0 ** 11 / sc) * sc).reshape(P, K) @ wh8 * 2.0 ** -8)
        c = (e5m2(xh) @ wl8 + e5m2(xl * 2.0 ** 11) @ wh8) * 2.0 ** -8
        m = np.abs(ref).max()
        print("  %2d   %.2e   %.2e     %.2e     %.2e" % (gap, np.abs(y3 - ref).max() / m, np.abs(main_ + a - ref).max() / m,
                                                          np.abs(main_ + b - ref).max() / m, np.abs(main_ + c - ref).max() / m))


if __name__ == "__main__":
    main()
