// "f16x3" super-resolution block: fp32-accurate convolution on the f16 matrix pipe of gfx950.
//
// Numerics.  Every fp32 operand is split into two fp16 terms, x = hi + lo with |x - hi - lo| <= 2^-24|x|
// (lo subnormals are kept, probe: scripts/probes/mfma_f16_probe.hip), and each product is evaluated as
// hi*hi + hi*lo + lo*hi with fp32 accumulation inside v_mfma_f32_32x32x16_f16: three MFMAs at 16x the
// f32-MFMA rate = 5.3x the exact-f32 kernel at ~1e-7 relative error per dot product (fp32-rounding class).
//
// Dataflow.  Activations travel between kernels already multiplied by the consumer's style vector and already
// split: "SPLIT" format = two planes [C/8][H][W][8 halfs] (hi plane, lo plane; same bytes as fp32).  The producer
// (input conversion, the FIR kernel, the previous conv's epilogue) does the scale + split once per element;
// the conv kernel's halo-patch staging is then a pure 16-byte copy global -> registers -> LDS, prefetched one
// stage ahead under the MFMAs.  Weights are split once at prepack time ([tap][ci/8][cout][8 hi | 8 lo]) and
// stream from L2 into A-operand registers one tap ahead.  Epilogue: demodulation * acc + bias -> lrelu*sqrt2
// (+clamp) -> any of {fp32 channel-blocked, SPLIT scaled by the next layer's styles} + the block's toRGB partial
// sums (128 couts per block) so the 134 MB activation of the last layer never goes to HBM.
//
// Block = 128 couts x (16x16) pixels, 4 waves of 64 couts x 128 pixels (2x4 MFMA tiles, 128 accumulators).
// Behaviour restated from modules/eg3ds/models/networks_stylegan2.py:37-94,286-373,429-473 and
// modules/eg3ds/torch_utils/ops/{conv2d_resample.py:116-133, upfirdn2d.py:171-215,317-354, bias_act.py:93-122}.
#include "r3d_sr_common.h"

namespace r3d {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
static constexpr int F_TILE_H = 16, F_TILE_W = 16;
static constexpr int F_PATCH_H = F_TILE_H + 2, F_PATCH_W = F_TILE_W + 2, F_PATCH_PIX = F_PATCH_H * F_PATCH_W;   // 324

__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo)
{
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    hi = (_Float16)c;
    lo = (_Float16)(v - (float)hi);
}

// ---- weights: [tap][ci/8][cout][8 hi | 8 lo] -------------------------------------------------------------------
__global__ void sr_prepack_f16_kernel(const float* __restrict__ w, int Ci, int Cout, uint4* __restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (tap, chunk, cout)
    const size_t total = (size_t)9 * (Ci / 8) * Cout;
    if (e >= total) return;
    const int co = e % Cout;
    const int chunk = (e / Cout) % (Ci / 8);
    const int tap = (int)(e / Cout / (Ci / 8));
    const float* src = w + ((size_t)co * Ci + chunk * 8) * 9 + tap;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) { _Float16 a, b; split1(src[9 * j], a, b); hi[j] = a; lo[j] = b; }
    out[2 * e] = *reinterpret_cast<uint4*>(&hi);
    out[2 * e + 1] = *reinterpret_cast<uint4*>(&lo);
}

// ---- input conversion: fp32 (NCHW or CB8) * style -> SPLIT ------------------------------------------------------
__global__ void to_split_kernel(const float* __restrict__ src, int cb8, const float* __restrict__ scale, size_t scale_stride_n,
                                uint4* __restrict__ dst, int C, int HW)
{
    const int n = blockIdx.z, cb = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float v[8];
    if (cb8) {
        const float4* s4 = reinterpret_cast<const float4*>(src + (((size_t)n * (C / 8) + cb) * HW + p) * 8);
        const float4 a = s4[0], b = s4[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const float* s = src + ((size_t)n * C + cb * 8) * HW + p;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = s[(size_t)c * HW];
    }
    const float* sc = scale + n * scale_stride_n + cb * 8;
    h8 hi, lo;
#pragma unroll
    for (int c = 0; c < 8; ++c) { _Float16 a, b; split1(v[c] * sc[c], a, b); hi[c] = a; lo[c] = b; }
    const size_t plane = (size_t)(C / 8) * HW;
    uint4* d = dst + (size_t)n * 2 * plane + (size_t)cb * HW + p;
    d[0] = *reinterpret_cast<uint4*>(&hi);
    d[plane] = *reinterpret_cast<uint4*>(&lo);
}

// ---- conv ----------------------------------------------------------------------------------------------------
struct Conv2Args {
    const uint4* x; size_t x_stride_n;        // SPLIT input (hi plane, lo plane), per-n stride in uint4 units
    const uint4* wp;                          // split prepacked weights
    const float* out_scale; const float* bias; size_t vec_stride_n;    // demod d[cout], bias[cout] (styles buffer)
    float* y_f32; size_t y_f32_stride_n; int OH, OW;                   // fp32 CB8 output (T buffer / x_out) or null
    uint4* y_split; size_t y_split_stride_n;                           // SPLIT output scaled by next_scale, or null
    const float* next_scale; size_t next_scale_stride_n;
    const float* wrgb; float* rgb_partial; size_t rgbp_stride_n;       // toRGB partials [Cout/128][3][OH*OW] or null
    int Cin, Cout, H, W, nphase, act; float clamp;
    ConvPhase ph[4];
};

template <int NTAPS, int CPS, bool FULL_EPI>   // CPS = chunk pairs (K=16 MFMA steps) staged per barrier; FULL_EPI: act/split/toRGB epilogue
__device__ __forceinline__ void conv2_block(const Conv2Args& a, const ConvPhase& ph, int n, uint4* patch /* [2][2*CPS][324] */)
{
    constexpr int NCH = 2 * CPS;                              // channel blocks per stage
    constexpr int STAGE_ELEMS = 2 * NCH * F_PATCH_PIX;         // uint4 per stage (hi + lo)
    constexpr int NPF = (STAGE_ELEMS + 255) / 256;
    const int tiles_x = (ph.outW + F_TILE_W - 1) / F_TILE_W;
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int i0 = ty * F_TILE_H, j0 = tx * F_TILE_W;
    const int m0 = blockIdx.y * BLOCK_M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    const int nchunks = a.Cin >> 3;
    const size_t plane = (size_t)nchunks * a.H * a.W;
    const uint4* X = a.x + (size_t)n * a.x_stride_n;
    const uint4* WP = a.wp;

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // N-tile pixel of this lane: rows (2nt, 2nt+1) of the wave's 8 rows; the odd row's columns are rotated by 2 so
    // that each ds_read_b128 lane group covers 16 distinct 16-byte LDS slots with the 18-pixel patch row stride
    const int prow = li >> 4, pcol = ((li & 15) - 2 * prow) & 15;
    int boff[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) boff[nt] = (wn * 8 + nt * 2 + prow + 1) * F_PATCH_W + (pcol + 1);

    // staging: element e = tid + 256k  ->  (plane, chunk, patch pixel); recomputed per stage to keep VGPRs for the MFMAs
    const int chunk_stride = a.H * a.W;
    uint4 pf[NPF];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int e = threadIdx.x + 256 * k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (e < STAGE_ELEMS) {
                const int pl = e / (NCH * F_PATCH_PIX);
                const int rem = e - pl * (NCH * F_PATCH_PIX);
                const int c = rem / F_PATCH_PIX, pp = rem - c * F_PATCH_PIX;
                const int py = pp / F_PATCH_W, px = pp - py * F_PATCH_W;
                const int iy = i0 + py - 1, ix = j0 + px - 1;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c0 + c < nchunks)
                    v = X[(size_t)pl * plane + (size_t)(c0 + c) * chunk_stride + iy * a.W + ix];
            }
            pf[k] = v;
        }
    };
    auto load_a = [&](int tapw, int cbase, h8 (&ah)[2], h8 (&al)[2]) {
        const int cb = cbase + h;
        const bool dead = cb >= nchunks;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const size_t wi = (((size_t)tapw * nchunks + (dead ? 0 : cb)) * a.Cout + (m0 + 64 * wm + 32 * mt + li)) * 2;
            uint4 q0 = WP[wi], q1 = WP[wi + 1];
            if (dead) { q0 = make_uint4(0, 0, 0, 0); q1 = q0; }
            ah[mt] = *reinterpret_cast<h8*>(&q0); al[mt] = *reinterpret_cast<h8*>(&q1);
        }
    };

    prefetch(0);
    h8 ah[2], al[2];
    load_a(ph.widx[0], 0, ah, al);
    const uint4* patch_hi = patch;
    const uint4* patch_lo = patch + NCH * F_PATCH_PIX;
    for (int c0 = 0; c0 < nchunks; c0 += NCH) {
        __syncthreads();                                   // everyone is done reading the previous stage
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int e = threadIdx.x + 256 * k;
            if (e < STAGE_ELEMS) patch[e] = pf[k];
        }
        __syncthreads();
        if (c0 + NCH < nchunks) prefetch(c0 + NCH);        // next stage's loads fly under this stage's MFMAs
#pragma unroll
        for (int cp = 0; cp < CPS; ++cp) {
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                // A operands one (tap, chunk-pair) ahead
                h8 nh[2], nl[2];
                {
                    int nt_ = t + 1, ncp = cp, nc0 = c0;
                    if (nt_ == NTAPS) { nt_ = 0; ++ncp; if (ncp == CPS) { ncp = 0; nc0 += NCH; } }
                    if (nc0 < nchunks) load_a(ph.widx[nt_], nc0 + 2 * ncp, nh, nl);
                    else { nh[0] = ah[0]; nh[1] = ah[1]; nl[0] = al[0]; nl[1] = al[1]; }
                }
                const int toff = ph.dy[t] * F_PATCH_W + ph.dx[t] + (2 * cp + h) * F_PATCH_PIX;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    uint4 r0 = patch_hi[boff[nt] + toff];
                    uint4 r1 = patch_lo[boff[nt] + toff];
                    const h8 bh = *reinterpret_cast<h8*>(&r0), bl = *reinterpret_cast<h8*>(&r1);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
                    }
                }
                ah[0] = nh[0]; ah[1] = nh[1]; al[0] = nl[0]; al[1] = nl[1];
            }
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    const float* B = a.bias ? a.bias + (size_t)n * a.vec_stride_n : nullptr;
    const float* D = a.out_scale + (size_t)n * a.vec_stride_n;
    const float* NS = a.next_scale ? a.next_scale + (size_t)n * a.next_scale_stride_n : nullptr;
    const bool do_rgb = FULL_EPI && a.rgb_partial != nullptr;
    float rgbp[4][3];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { rgbp[nt][0] = 0.f; rgbp[nt][1] = 0.f; rgbp[nt][2] = 0.f; }
    const size_t oplane = (size_t)(a.Cout >> 3) * a.OH * a.OW;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int i = i0 + wn * 8 + nt * 2 + prow, j = j0 + pcol;
        const bool inside = i < ph.outH && j < ph.outW;
        const int oy = i * ph.oy_mul + ph.oy_add, ox = j * ph.ox_mul + ph.ox_add;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = m0 + 64 * wm + 32 * mt + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[mt][nt][4 * g + r] * D[co + r];
                    if (FULL_EPI && a.act) {
                        t += B[co + r];
                        t = (t < 0.f ? t * 0.2f : t) * 1.4142135623730951f;
                        if (a.clamp >= 0.f) t = fminf(fmaxf(t, -a.clamp), a.clamp);
                    }
                    v[r] = t;
                }
                if (do_rgb) {
                    const float4 w0 = *reinterpret_cast<const float4*>(a.wrgb + (size_t)n * a.vec_stride_n + co);
                    const float4 w1 = *reinterpret_cast<const float4*>(a.wrgb + (size_t)n * a.vec_stride_n + a.Cout + co);
                    const float4 w2 = *reinterpret_cast<const float4*>(a.wrgb + (size_t)n * a.vec_stride_n + 2 * a.Cout + co);
                    rgbp[nt][0] += v[0] * w0.x + v[1] * w0.y + v[2] * w0.z + v[3] * w0.w;
                    rgbp[nt][1] += v[0] * w1.x + v[1] * w1.y + v[2] * w1.z + v[3] * w1.w;
                    rgbp[nt][2] += v[0] * w2.x + v[1] * w2.y + v[2] * w2.z + v[3] * w2.w;
                }
                if (!inside) continue;
                const size_t pix = ((size_t)(co >> 3) * a.OH + oy) * a.OW + ox;
                if (a.y_f32)
                    *reinterpret_cast<float4*>(a.y_f32 + (size_t)n * a.y_f32_stride_n + pix * 8 + (co & 7)) = make_float4(v[0], v[1], v[2], v[3]);
                if (FULL_EPI && a.y_split) {
                    h4 hi, lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { _Float16 x0, x1; split1(v[r] * NS[co + r], x0, x1); hi[r] = x0; lo[r] = x1; }
                    uint2* dst = reinterpret_cast<uint2*>(a.y_split + (size_t)n * a.y_split_stride_n + pix) + ((co & 7) >> 2);
                    dst[0] = *reinterpret_cast<uint2*>(&hi);
                    dst[2 * oplane] = *reinterpret_cast<uint2*>(&lo);
                }
            }
    }
    if (do_rgb) {
        // reduce the two lane halves (disjoint couts), then the two cout-waves through LDS (reusing the patch)
        float* red = reinterpret_cast<float*>(patch);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int o = 0; o < 3; ++o) rgbp[nt][o] += __shfl_xor(rgbp[nt][o], 32);
        __syncthreads();
        if (wm == 1 && h == 0) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int o = 0; o < 3; ++o) red[((wn * 4 + nt) * 3 + o) * 32 + li] = rgbp[nt][o];
        }
        __syncthreads();
        if (wm == 0 && h == 0) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int i = i0 + wn * 8 + nt * 2 + prow, j = j0 + pcol;
                if (i < ph.outH && j < ph.outW) {
                    const int oy = i * ph.oy_mul + ph.oy_add, ox = j * ph.ox_mul + ph.ox_add;
#pragma unroll
                    for (int o = 0; o < 3; ++o)
                        a.rgb_partial[(size_t)n * a.rgbp_stride_n + ((size_t)blockIdx.y * 3 + o) * a.OH * a.OW + (size_t)oy * a.OW + ox] =
                            rgbp[nt][o] + red[((wn * 4 + nt) * 3 + o) * 32 + li];
                }
            }
        }
    }
}

// plain 3x3 conv: 9 taps, 16 input channels per barrier
__global__ __launch_bounds__(256, 2) void conv_mfma_f16x3_kernel(Conv2Args a)
{
    __shared__ uint4 patch[2 * 2 * 1 * F_PATCH_PIX];
    const int n = blockIdx.z;
    conv2_block<9, 1, true>(a, a.ph[0], n, patch);
}

// stride-2 transposed conv phases (4/2/2/1 taps)
__global__ __launch_bounds__(256, 2) void tconv_mfma_f16x3_kernel(Conv2Args a)
{
    __shared__ uint4 patch[2 * 2 * 1 * F_PATCH_PIX];
    const int n = blockIdx.z / a.nphase, p = blockIdx.z - n * a.nphase;
    const ConvPhase& ph = a.ph[p];
    const int tiles = ((ph.outW + F_TILE_W - 1) / F_TILE_W) * ((ph.outH + F_TILE_H - 1) / F_TILE_H);
    if ((int)blockIdx.x >= tiles) return;
    switch (ph.ntaps) {
        case 4: conv2_block<4, 1, false>(a, ph, n, patch); break;
        case 2: conv2_block<2, 1, false>(a, ph, n, patch); break;
        default: conv2_block<1, 1, false>(a, ph, n, patch); break;
    }
}

// ---- FIR 4x4 (gain 4, pad 1) + bias + lrelu*sqrt2 on the transposed-conv output; writes SPLIT scaled by the next
// conv's styles.  T fp32 [C/8][2H+1][2W+1][8] -> SPLIT [C/8][2H][2W].  One thread: one output pixel x 8 channels.
__global__ void fir_bias_act_split_kernel(const float* __restrict__ T, size_t t_stride_n, const float* __restrict__ bias,
                                          const float* __restrict__ next_scale, size_t vec_stride_n,
                                          uint4* __restrict__ y, size_t y_stride_n, int C, int OH, int OW, float clamp)
{
    const int n = blockIdx.z, cb = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= OH * OW) return;
    const int oy = p / OW, ox = p - oy * OW;
    const int TH = OH + 1, TW = OW + 1;
    const float* Tn = T + (size_t)n * t_stride_n + (size_t)cb * TH * TW * 8;
    const float f1[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
        const int ty = oy + aa - 1;
        if (ty < 0 || ty >= TH) continue;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const int tx = ox + bb - 1;
            if (tx < 0 || tx >= TW) continue;
            const float4* s4 = reinterpret_cast<const float4*>(Tn + ((size_t)ty * TW + tx) * 8);
            const float4 u = s4[0], v = s4[1];
            const float w = f1[aa] * f1[bb];
            acc[0] += u.x * w; acc[1] += u.y * w; acc[2] += u.z * w; acc[3] += u.w * w;
            acc[4] += v.x * w; acc[5] += v.y * w; acc[6] += v.z * w; acc[7] += v.w * w;
        }
    }
    const float* b = bias + (size_t)n * vec_stride_n + cb * 8;
    const float* ns = next_scale + (size_t)n * vec_stride_n + cb * 8;
    h8 hi, lo;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float t = acc[c] + b[c];
        t = (t < 0.f ? t * 0.2f : t) * 1.4142135623730951f;
        if (clamp >= 0.f) t = fminf(fmaxf(t, -clamp), clamp);
        _Float16 x0, x1; split1(t * ns[c], x0, x1); hi[c] = x0; lo[c] = x1;
    }
    const size_t plane = (size_t)(C / 8) * OH * OW;
    uint4* d = y + (size_t)n * y_stride_n + (size_t)cb * OH * OW + p;
    d[0] = *reinterpret_cast<uint4*>(&hi);
    d[plane] = *reinterpret_cast<uint4*>(&lo);
}

// ---- image finalize: img_out = upsample2d(img_in) + bias + sum_m partial[m]  (networks_stylegan2.py:463-469) ----
__global__ void rgb_finalize_kernel(const float* __restrict__ img_prev, const float* __restrict__ partial, size_t part_stride_n,
                                    int nparts, const float* __restrict__ brgb, size_t vec_stride_n,
                                    float* __restrict__ img_out, int H, int W, float clamp)
{
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, xx = p - y * W;
    const int Hh = H >> 1, Wh = W >> 1;
    const int ky = y >> 1, kx = xx >> 1;
    int r0, r1, c0, c1; float wy0, wy1, wx0, wx1;
    if (y & 1) { r0 = ky; r1 = ky + 1; wy0 = 0.75f; wy1 = 0.25f; } else { r0 = ky - 1; r1 = ky; wy0 = 0.25f; wy1 = 0.75f; }
    if (xx & 1) { c0 = kx; c1 = kx + 1; wx0 = 0.75f; wx1 = 0.25f; } else { c0 = kx - 1; c1 = kx; wx0 = 0.25f; wx1 = 0.75f; }
    const bool vr0 = r0 >= 0 && r0 < Hh, vr1 = r1 >= 0 && r1 < Hh, vc0 = c0 >= 0 && c0 < Wh, vc1 = c1 >= 0 && c1 < Wh;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = brgb[(size_t)n * vec_stride_n + c];
        for (int m = 0; m < nparts; ++m) t += partial[(size_t)n * part_stride_n + ((size_t)m * 3 + c) * H * W + p];
        if (clamp >= 0.f) t = fminf(fmaxf(t, -clamp), clamp);
        const float* I = img_prev + ((size_t)n * 3 + c) * Hh * Wh;
        float up = 0.f;
        if (vr0 && vc0) up += I[(size_t)r0 * Wh + c0] * (wy0 * wx0);
        if (vr0 && vc1) up += I[(size_t)r0 * Wh + c1] * (wy0 * wx1);
        if (vr1 && vc0) up += I[(size_t)r1 * Wh + c0] * (wy1 * wx0);
        if (vr1 && vc1) up += I[(size_t)r1 * Wh + c1] * (wy1 * wx1);
        img_out[((size_t)n * 3 + c) * H * W + p] = up + t;
    }
}

__global__ void cb8_to_nchw2_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW)
{
    const int n = blockIdx.z, cb = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float4* s = reinterpret_cast<const float4*>(src + (((size_t)n * (C / 8) + cb) * HW + p) * 8);
    const float4 a = s[0], b = s[1];
    float* d = dst + ((size_t)n * C + cb * 8) * HW + p;
    d[0] = a.x; d[(size_t)HW] = a.y; d[(size_t)2 * HW] = a.z; d[(size_t)3 * HW] = a.w;
    d[(size_t)4 * HW] = b.x; d[(size_t)5 * HW] = b.y; d[(size_t)6 * HW] = b.z; d[(size_t)7 * HW] = b.w;
}

// ---- host ------------------------------------------------------------------------------------------------------
int sr_prepack_f16x3(int Cin, int Cout, const float* c0_w, const float* c1_w, void* prepacked, hipStream_t st)
{
    float* out = reinterpret_cast<float*>(prepacked);
    const size_t m0 = (size_t)9 * (Cin / 8) * Cout, m1 = (size_t)9 * (Cout / 8) * Cout;
    hipLaunchKernelGGL(sr_prepack_f16_kernel, dim3((unsigned)((m0 + 255) / 256)), dim3(256), 0, st, c0_w, Cin, Cout,
                       reinterpret_cast<uint4*>(out));
    hipLaunchKernelGGL(sr_prepack_f16_kernel, dim3((unsigned)((m1 + 255) / 256)), dim3(256), 0, st, c1_w, Cout, Cout,
                       reinterpret_cast<uint4*>(out + (size_t)9 * Cin * Cout));
    return check_launch("sr_block_prepack");
}

static void launch_conv2(const Conv2Args& a, int tiles, int N, hipStream_t st)
{
    dim3 grid(tiles, a.Cout / BLOCK_M, N * a.nphase);
    if (a.nphase == 1) hipLaunchKernelGGL(conv_mfma_f16x3_kernel, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(tconv_mfma_f16x3_kernel, grid, dim3(256), 0, st, a);
}

int sr_block_forward_f16x3(const void* prepacked, const void* styles, int N, int Cin, int Cout, int Hin, int Win,
                           const void* x, int x_format, const float* img, float clamp,
                           void* x_out, int x_out_format, const float* next_scale, size_t next_scale_stride,
                           float* img_out, void* workspace, size_t workspace_bytes, hipStream_t st)
{
    (void)workspace_bytes;
    const SrStyleLayout L = sr_style_layout(Cin, Cout);
    const float* pk = reinterpret_cast<const float*>(styles);
    const float* wpk = reinterpret_cast<const float*>(prepacked);
    const int OH = 2 * Hin, OW = 2 * Win, TH = OH + 1, TW = OW + 1;
    char* wsb = reinterpret_cast<char*>(workspace);
    uint4* xin = reinterpret_cast<uint4*>(wsb); wsb += align256((size_t)N * Cin * Hin * Win * 4);
    float* T = reinterpret_cast<float*>(wsb);   wsb += align256((size_t)N * Cout * TH * TW * 4);
    uint4* y0 = reinterpret_cast<uint4*>(wsb);  wsb += align256((size_t)N * Cout * OH * OW * 4);
    float* xo = reinterpret_cast<float*>(wsb);                    // fp32 CB8 x (only when an fp32 x_out is requested)
    float* rgbp = reinterpret_cast<float*>(wsb + align256((size_t)N * Cout * OH * OW * 4));

    const uint4* xs = reinterpret_cast<const uint4*>(x);
    if (x_format != R3D_FMT_SPLIT) {
        ProfScope ps(R3D_PROF_LAYOUT, st);
        hipLaunchKernelGGL(to_split_kernel, dim3((Hin * Win + 255) / 256, Cin / 8, N), dim3(256), 0, st,
                           reinterpret_cast<const float*>(x), x_format == R3D_FMT_CB8 ? 1 : 0, pk + L.s0, L.total, xin, Cin, Hin * Win);
        xs = xin;
    }
    // ---- conv0: stride-2 transposed conv as 4 phases -> T (demodulated, fp32) ------------------------------------
    {
        Conv2Args a = {};
        a.x = xs; a.x_stride_n = (size_t)Cin / 8 * Hin * Win * 2;
        a.wp = reinterpret_cast<const uint4*>(wpk);
        a.out_scale = pk + L.d0; a.bias = nullptr; a.vec_stride_n = L.total;
        a.y_f32 = T; a.y_f32_stride_n = (size_t)Cout * TH * TW; a.OH = TH; a.OW = TW;
        a.Cin = Cin; a.Cout = Cout; a.H = Hin; a.W = Win; a.nphase = 4; a.act = 0; a.clamp = -1.f;
        sr_fill_tconv_phases(a.ph, Hin, Win);
        int maxtiles = 0;
        for (int p = 0; p < 4; ++p) {
            const int tiles = ((a.ph[p].outW + F_TILE_W - 1) / F_TILE_W) * ((a.ph[p].outH + F_TILE_H - 1) / F_TILE_H);
            if (tiles > maxtiles) maxtiles = tiles;
        }
        ProfScope ps(R3D_PROF_CONV, st);
        launch_conv2(a, maxtiles, N, st);
    }
    {
        ProfScope ps(R3D_PROF_FIR, st);
        hipLaunchKernelGGL(fir_bias_act_split_kernel, dim3((OH * OW + 255) / 256, Cout / 8, N), dim3(256), 0, st,
                           T, (size_t)Cout * TH * TW, pk + L.b0, pk + L.s1, L.total, y0, (size_t)Cout / 8 * OH * OW * 2, Cout, OH, OW, clamp);
    }
    // ---- conv1 (3x3) + bias/lrelu + toRGB partials (+ optional x outputs) ------------------------------------------
    const bool want_f32 = x_out && (x_out_format == R3D_FMT_NCHW || x_out_format == R3D_FMT_CB8);
    {
        Conv2Args a = {};
        a.x = y0; a.x_stride_n = (size_t)Cout / 8 * OH * OW * 2;
        a.wp = reinterpret_cast<const uint4*>(wpk + (size_t)9 * Cin * Cout);
        a.out_scale = pk + L.d1; a.bias = pk + L.b1; a.vec_stride_n = L.total;
        a.OH = OH; a.OW = OW;
        if (want_f32) { a.y_f32 = (x_out_format == R3D_FMT_CB8) ? reinterpret_cast<float*>(x_out) : xo; a.y_f32_stride_n = (size_t)Cout * OH * OW; }
        if (x_out && x_out_format == R3D_FMT_SPLIT) {
            a.y_split = reinterpret_cast<uint4*>(x_out); a.y_split_stride_n = (size_t)Cout / 8 * OH * OW * 2;
            a.next_scale = next_scale; a.next_scale_stride_n = next_scale_stride;
        }
        a.wrgb = pk + L.wrgb; a.rgb_partial = rgbp; a.rgbp_stride_n = (size_t)(Cout / BLOCK_M) * 3 * OH * OW;
        a.Cin = Cout; a.Cout = Cout; a.H = OH; a.W = OW; a.nphase = 1; a.act = 1; a.clamp = clamp;
        sr_fill_conv3x3_phase(a.ph, OH, OW);
        const int tiles = ((OW + F_TILE_W - 1) / F_TILE_W) * ((OH + F_TILE_H - 1) / F_TILE_H);
        ProfScope ps(R3D_PROF_CONV, st);
        launch_conv2(a, tiles, N, st);
    }
    {
        ProfScope ps(R3D_PROF_TORGB, st);
        hipLaunchKernelGGL(rgb_finalize_kernel, dim3((OH * OW + 255) / 256, N), dim3(256), 0, st, img, rgbp,
                           (size_t)(Cout / BLOCK_M) * 3 * OH * OW, Cout / BLOCK_M, pk + L.brgb, L.total, img_out, OH, OW, clamp);
    }
    if (x_out && x_out_format == R3D_FMT_NCHW) {
        ProfScope ps(R3D_PROF_LAYOUT, st);
        hipLaunchKernelGGL(cb8_to_nchw2_kernel, dim3((OH * OW + 255) / 256, Cout / 8, N), dim3(256), 0, st, xo,
                           reinterpret_cast<float*>(x_out), Cout, OH * OW);
    }
    return check_launch("sr_block_forward(f16x3)");
}

}  // namespace r3d
