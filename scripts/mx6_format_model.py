"""numpy model: what the f16mx cross products (xh*wl + xl*wh, r3d_sr_f16x3.hip "f16mx") would lose on the 6- and 4-bit operands of
v_mfma_scale_f32_32x32x64_f8f6f4, which gfx950 issues at TWICE the fp8 rate when BOTH operands are fp6 / fp4 (8 passes instead of 16: the five
cross-product MFMAs of a 16-channel stage are 320 of the direct conv's 608 matrix cycles -> 448).  Formats (OCP MX):
    e5m2 | e4m3     today: activation records e5m2 (per-element exponent), weight records e4m3 of the row-normalised weights; no block scales
    e2m3 (fp6)      3 mantissa bits, 2 exponent bits (max 7.5, subnormal step 1/8): 6 bits of range under the block's scale
    e3m2 (bf6)      2 mantissa bits, 3 exponent bits (max 28, subnormal step 1/16)
    e2m1 (fp4)      1 mantissa bit (max 6, subnormal 0.5)
with one E8M0 scale per MFMA lane operand = per (pixel | cout, tap, 16 channels): the 32 K-elements [xh (16) | xl * 2^11 (16)] of an activation
record, [wl * 2^11 | wh] of a weight record, share it (the instruction takes the scale per lane: a VGPR byte next to the 24 data bytes of a record).
Inputs: Gaussian activations under a bound 2^gap above sigma (the sweep of mx_format_model.py), and the dense heavy tail of
tests/test_gpu_pinned_config.py::test_sr_block_dense_heavy_tail (log-normal, sigma = 4 binades ... per element).
Run on the CPU: python scripts/mx6_format_model.py > profiles/r06/mx6_format_model.txt"""
import numpy as np

from mx_format_model import e4m3, e5m2, quant, split_f16

FMT = {"e2m3": (3, 0, 7.5, 2), "e3m2": (2, -2, 28.0, 4), "e2m1": (1, 0, 6.0, 2)}      # mantissa bits, min normal exponent, max, exponent of the max


def block_quant(v, fmt):
    """v [..., 32]: OCP MX block quantisation: scale = 2^(floor(log2(max|v|)) - emax), elements to the format (saturating); returns the de-quantised values"""
    mb, emin, vmax, emax = FMT[fmt]
    m = np.abs(v).max(-1, keepdims=True)
    sc = 2.0 ** (np.floor(np.log2(np.maximum(m, 1e-300))) - emax)
    return quant(v / sc, mb, emin, vmax) * sc


def cross(x, ws, fa, fw):
    """the two correction sums on block-scaled records of formats fa (activations) / fw (weights); blocks = 16 channels (K is ordered tap-major)"""
    P, K = x.shape
    xh, xl = split_f16(x)
    wh, wl = split_f16(ws)
    G = K // 16
    bx = np.concatenate([xh.reshape(P, G, 16), xl.reshape(P, G, 16) * 2.0 ** 11], -1)
    bw = np.concatenate([wl.reshape(G, 16) * 2.0 ** 11, wh.reshape(G, 16)], -1)
    if fa == "e5m2":
        qx = np.concatenate([e5m2(bx[..., :16]), e5m2(bx[..., 16:])], -1)
    else:
        qx = block_quant(bx, fa)
    if fw == "e4m3":
        qw = np.concatenate([e4m3(bw[..., :16] * 2.0 ** -3), e4m3(bw[..., 16:] * 2.0 ** -3)], -1) * 8.0
    else:
        qw = block_quant(bw, fw)
    return np.einsum("pgk,gk->p", qx, qw) * 2.0 ** -11, xh @ wh


def report(name, x, ws):
    ref = x @ ws
    m = np.abs(ref).max()
    xh, xl = split_f16(x); wh, wl = split_f16(ws)
    y3 = xh @ wh + xh @ wl + xl @ wh
    row = ["%.2e" % (np.abs(y3 - ref).max() / m)]
    for fa, fw in (("e5m2", "e4m3"), ("e2m3", "e2m3"), ("e3m2", "e2m3"), ("e3m2", "e3m2"), ("e2m1", "e2m1")):
        c, main_ = cross(x, ws, fa, fw)
        e = np.abs(main_ + c - ref)
        row.append("%.2e (med %.1e)" % (e.max() / m, np.median(e) / np.median(np.abs(ref))))
    print("%-22s %s" % (name, "   ".join(row)))


def main():
    rng = np.random.default_rng(0)
    P, K = 2048, 9 * 256
    w = rng.standard_normal((K,)) / np.sqrt(K)
    kw = 10 - np.floor(np.log2(np.abs(w).max()))
    ws = w * 2.0 ** kw
    print("# max|y - y_fp64| / max|y_fp64| over %d pixels (median error / median |y| in brackets); K = %d; activation x weight record formats" % (P, K))
    print("# %-20s %-10s %-24s %-24s %-24s %-24s %-24s" % ("input", "f16x3", "e5m2 x e4m3 (today)", "e2m3 x e2m3", "e3m2 x e2m3", "e3m2 x e3m2", "e2m1 x e2m1 (fp4)"))
    for gap in (3, 9, 15, 21):
        report("gaussian, gap 2^%d" % gap, rng.standard_normal((P, K)) * 2.0 ** (15 - gap), ws)
    for sig in (2.0, 4.0):
        x = rng.standard_normal((P, K)) * 2.0 ** (rng.standard_normal((P, K)) * sig)
        x *= 2.0 ** 14 / np.abs(x).max()
        report("log-normal, %g binades" % sig, x, ws)
    # heavy-tailed WEIGHTS too (a row with a few dominant taps): the block scale follows the group, the e4m3 records follow the row
    w2 = rng.standard_normal((K,)) * 2.0 ** (rng.standard_normal((K,)) * 2.0)
    ws2 = w2 * 2.0 ** (10 - np.floor(np.log2(np.abs(w2).max())))
    report("gaussian x heavy row", rng.standard_normal((P, K)) * 2.0 ** 6, ws2)


if __name__ == "__main__":
    main()
