// "f16x3" super-resolution block: fp32-accurate convolution on the f16 matrix pipe of gfx950.
//
// Numerics.  Every fp32 operand is split into two fp16 terms, x = hi + lo with |x - hi - lo| <= 2^-24|x|
// (lo subnormals are kept, probe: scripts/probes/mfma_f16_probe.hip), and each product is evaluated as
// hi*hi + hi*lo + lo*hi with fp32 accumulation inside v_mfma_f32_32x32x16_f16: three MFMAs at 16x the
// f32-MFMA rate = 5.3x the exact-f32 kernel at ~1e-7 relative error per dot product (fp32-rounding class).
//
// Dataflow.  Activations travel between kernels already multiplied by the consumer's style vector and already
// split: "SPLIT" format = two planes [C/8][H][W][8 halfs] (hi plane, lo plane; same bytes as fp32).  The producer
// (input conversion, the up-sampling conv's epilogue, the previous conv's epilogue) does the scale + split once per
// element.  Weights are split once at prepack time.  Every operand byte reaches LDS by LDS-DMA (global_load_lds_dwordx4
// from per-lane addresses, inline asm): no staging VGPRs, no ds_write, loads of stage s+1 in flight during stage s.
//
// Kernels.
//   conv_mfma_f16x3_kernel    plain 3x3 conv (conv3x3_dma_block): 128 couts x 16x16 px per block, 8 waves x (64 x 64),
//                             double-buffered patch, two-tap weight sub-stages, one barrier per sub-stage; epilogue =
//                             demodulation * acc + bias -> lrelu*sqrt2 (+clamp) -> {fp32 CB8, fp32 NCHW, SPLIT} + toRGB partials
//   upconv_fir_f16x3_kernel   up=2 layers: transposed conv (4 output phases) + FIR 4x4 + bias + lrelu + split, fused;
//                             32 couts x 16x16 grid x 4 phases per block, the fp32 T never leaves the CU
//   conv1x1_mfma_f16x3_kernel / tconv_mfma_f16x3_kernel (conv2_block): 1x1 convs of the fusion stacks; the per-phase
//                             transposed conv is kept behind R3D_UPCONV=0 for A/B runs with fir_bias_act_split_kernel
//   to_split / upsample2x_bilinear / rgb_finalize / prepack kernels: layout and glue
// Behaviour restated from modules/eg3ds/models/networks_stylegan2.py:37-94,286-373,429-473 and
// modules/eg3ds/torch_utils/ops/{conv2d_resample.py:116-133, upfirdn2d.py:171-215,317-354, bias_act.py:93-122}.
#include <stdlib.h>

#include "r3d_sr_common.h"
#include "r3d_stamps.h"

namespace r3d {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
static constexpr int F_TILE_H = 16, F_TILE_W = 16;
static constexpr int F_PATCH_H = F_TILE_H + 2, F_PATCH_W = F_TILE_W + 2, F_PATCH_PIX = F_PATCH_H * F_PATCH_W;   // 324

__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo)
{
    v = as_rounded(v);
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    hi = (_Float16)c;
    lo = (_Float16)(v - (float)hi);
}

// hot epilogues: the range fold (r3d_chain_fold) guarantees |v| < 2^15, so the saturation of split1 is dead weight there
__device__ __forceinline__ void split1_folded(float v, _Float16& hi, _Float16& lo)
{
    v = as_rounded(v);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// ---- "f16mx": the two 2^-11-sized correction products on the block-scaled fp8 MFMA (R3D_SR_F16MX) -------------------------------------
// A product is x*w = xh*wh + (xh*wl + xl*wh) (+ xl*wl ~ 2^-22, dropped as in f16x3).  f16x3 spends two f16 MFMAs per K = 16 on the bracket;
// here the bracket of TWO taps x 16 channels is ONE v_mfma_scale_f32_32x32x64_f8f6f4 (2x the f16 rate):
//     B lane (pixel, h):  32 bytes  [ xh8 (16 ch) | xl8 (16 ch) ]  of tap h        xh8 = bf8(hi),         xl8 = bf8(lo * 2^11)       OCP e5m2
//     A lane (cout,  h):  32 bytes  [ wl8 (16 ch) | wh8 (16 ch) ]  of tap h        wl8 = fp8(wl * 2^8),   wh8 = fp8(wh * 2^-3)       OCP e4m3
// (operand layout probed on the GPU: lane l holds row / column l % 32 and k = 32 * (l / 32) + byte, scripts/probes/mx_correction_probe.hip;
// the instruction takes the element format per operand: cbsz = 0 (e4m3) for A, blgp = 1 (e5m2) for B); the E8M0 scale operands supply the
// common factor 2^-8.
// Round 5: the ACTIVATION records are e5m2, stored at the top of its range.  Until round 4 they were e4m3 (xh8 = fp8(hi * 2^-7)), which has
// 14 binades of normal range under the fold's bound: the records carry one exponent per tensor, and with the bound 2^17 or more above the
// typical pixel (a 2^10 .. 2^14 sigma spike on top of the ~5 binades a propagated L1 bound is loose by) xh8 went subnormal and the typical
// outputs fell to the TF32 class (far-field 1.6e-4 / 3.6e-4 of max|ref| in tests/test_gpu_pinned_config.py -- why f16mx could not be the
// library default).  e5m2 holds hi < 2^15 directly (max 57 344) with 29 binades of NORMAL range below it -- a per-element exponent does what
// per-(pixel, group) E8M0 scale bytes would have done, without a byte to transport or a scale operand to load -- at one mantissa bit less:
// the correction is accurate to 2^-3 of a 2^-11-sized term instead of 2^-4, i.e. ~2^-15 of the product (numpy model of both formats over
// bounds 2^3 .. 2^25 above the typical value: e4m3 1.0e-5 -> 3e-4 of max|ref| from 2^17 on, e5m2 1.7e-5 flat; profiles/r05/mx_format_model.txt).
// The weight records stay e4m3: a weight row is normalised by its own maximum (2^kw[co]) at prepack time.
// In memory the fp8 records take the place of the fp16 lo plane with the SAME addressing: "lo" chunk 2G holds xh8 (wl8) of the 16
// channels 16G..16G+15, chunk 2G+1 holds xl8 (wh8) -- so every DMA of the f16x3 kernels is unchanged.
// Byte order inside a 16-byte record (round 4): dword d = 2 h + p holds channels 8 p + 4 h .. + 3 (p = which 8-channel chunk of the group,
// h = which half of that chunk), activations and weights alike (the K index of a dot product may be permuted freely).  The producers' lanes
// hold 4 channels = (chunk, half), and the two chunks of a group come from consecutive iterations of their loops: with this order a lane
// writes 8 contiguous bytes and a lane pair a whole record per store -- the straight order (dword 2 p + h) made every record two 8-byte
// read-modify-writes (block1.conv0 wrote 153 MB for 134 MB, profiles/r04/pmc_summary.txt).
__device__ __host__ __forceinline__ int mx_rec_chan(int dword) { return 8 * (dword & 1) + 4 * (dword >> 1); }     // first channel of a record dword
#ifndef R3D_TAPS_CT
#define R3D_TAPS_CT 3            // experiment switch (bisect): bit 0 = compile-time tap offsets in the f16 part, bit 1 = in the fp8 part
#endif
#ifndef R3D_MX_FREE_SCHED
#define R3D_MX_FREE_SCHED 0   // experiment switch: 1 = no scheduling fences around the fp8 part of a sub-stage
#endif
#ifndef R3D_MX_DRAIN
#define R3D_MX_DRAIN 0        // experiment switch: 1 = the MX conv waits for ALL its DMAs at every sub-stage (the spilling build's behaviour)
#endif
#ifndef R3D_MX_ACT_E4M3
#define R3D_MX_ACT_E4M3 0     // experiment switch (A/B builds): 1 = the round-4 activation records, e4m3 of hi * 2^-7 / lo * 2^4
#endif
static constexpr float kMxWl = 256.0f /* 2^8 */, kMxWh = 0.125f /* 2^-3 */;
#if R3D_MX_ACT_E4M3
static constexpr int kMxFmtB = 0;                               // blgp: B operand (activation records) OCP e4m3
static constexpr int kMxScaleA = 127, kMxScaleB = 126;          // E8M0 bytes: 2^0 * 2^-1
static constexpr float kMxXh = 0.0078125f /* 2^-7 */, kMxXl = 16.0f /* 2^4 */;
#else
static constexpr int kMxFmtB = 1;                               // blgp: B operand (activation records) OCP e5m2
static constexpr int kMxScaleA = 127, kMxScaleB = 119;          // E8M0 bytes: 2^0 * 2^-8
static constexpr float kMxXh = 1.0f, kMxXl = 2048.0f /* 2^11 */;
#endif

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d)      // OCP e4m3: the weight records
{
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}
__device__ __forceinline__ unsigned pack4_x8(float a, float b, float c, float d)       // the activation records (e5m2; e4m3 in the A/B build)
{
#if R3D_MX_ACT_E4M3
    return pack4_fp8(a, b, c, d);
#else
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
    return (unsigned)v;
#endif
}

// ---- per-cout statistics of a weight tensor [CoutReal][row_len]: tail = {2^-kw[co], sum|w[co]|} (ConvTail), padded couts -> {1, 0}.
// kw[co] = weight_row_exp(max|w[co]|): the row is stored as w * 2^kw (max in [2^10, 2^11)) and the conv epilogue multiplies by
// 2^-kw -- exact, and it makes the fp16 hi/lo split independent of the overall magnitude of the trained weights.
__global__ void weight_row_stats_kernel(const float* __restrict__ w, int row_len, int CoutReal, int Cout, float* __restrict__ tail)
{
    const int co = blockIdx.x;
    const ConvTail T = conv_tail_layout(Cout);
    float mx = 0.f, l1 = 0.f;
    if (co < CoutReal)
        for (int i = threadIdx.x; i < row_len; i += blockDim.x) { const float v = fabsf(w[(size_t)co * row_len + i]); mx = fmaxf(mx, v); l1 += v; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, d)); l1 += __shfl_xor(l1, d); }
    __shared__ float rm[4], rl[4];
    if ((threadIdx.x & 63) == 0) { rm[threadIdx.x >> 6] = mx; rl[threadIdx.x >> 6] = l1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
        tail[T.winv + co] = pow2f(-weight_row_exp(mx));
        tail[T.l1 + co] = rl[0] + rl[1] + rl[2] + rl[3];
    }
}

// ---- weights: [tap][ci/8][hi|lo][cout] x 8 halfs (hi and lo in separate 16-byte planes: conflict-free ds_read_b128) ----
// MX variant of the same layout (R3D_SR_F16MX): the hi rows are unchanged; the "lo" row of chunk 2G holds wl8 of the 16 channels of
// group G, the "lo" row of chunk 2G+1 their wh8 (one thread per (tap, 16-channel group, cout)).
__global__ void sr_prepack_mx_kernel(const float* __restrict__ w, int CiReal, int CoutReal, int ntaps, int Ci, int Cout,
                                     const float* __restrict__ winv, uint4* __restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (tap, group, cout)
    const size_t total = (size_t)ntaps * (Ci / 16) * Cout;
    if (e >= total) return;
    const int co = e % Cout;
    const int grp = (e / Cout) % (Ci / 16);
    const int tap = (int)(e / Cout / (Ci / 16));
    const float ws = 1.0f / winv[co];
    float hi[16], lo[16];
    h8 h0, h1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ci = grp * 16 + j;
        const float v = (co < CoutReal && ci < CiReal) ? w[((size_t)co * CiReal + ci) * ntaps + tap] * ws : 0.f;
        _Float16 a, b; split1(v, a, b);
        hi[j] = (float)a; lo[j] = v - (float)a;
        if (j < 8) h0[j] = a; else h1[j - 8] = a;
    }
    uint4 wl8, wh8;
    unsigned* pl = reinterpret_cast<unsigned*>(&wl8); unsigned* ph = reinterpret_cast<unsigned*>(&wh8);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = mx_rec_chan(q);
        pl[q] = pack4_fp8(lo[c] * kMxWl, lo[c + 1] * kMxWl, lo[c + 2] * kMxWl, lo[c + 3] * kMxWl);
        ph[q] = pack4_fp8(hi[c] * kMxWh, hi[c + 1] * kMxWh, hi[c + 2] * kMxWh, hi[c + 3] * kMxWh);
    }
    const size_t base = ((size_t)tap * (Ci / 8) + 2 * grp) * 2 * Cout + co;      // [tap][chunk][hi|lo][cout]
    out[base] = *reinterpret_cast<uint4*>(&h0);
    out[base + Cout] = wl8;
    out[base + 2 * Cout] = *reinterpret_cast<uint4*>(&h1);
    out[base + 3 * Cout] = wh8;
}

__global__ void sr_prepack_f16_kernel(const float* __restrict__ w, int CiReal, int CoutReal, int ntaps, int Ci, int Cout,
                                      const float* __restrict__ winv, uint4* __restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (tap, chunk, cout) in the PADDED index space
    const size_t total = (size_t)ntaps * (Ci / 8) * Cout;
    if (e >= total) return;
    const int co = e % Cout;
    const int chunk = (e / Cout) % (Ci / 8);
    const int tap = (int)(e / Cout / (Ci / 8));
    const float ws = 1.0f / winv[co];                                    // 2^kw[co] (exact reciprocal of a power of two)
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = chunk * 8 + j;
        const float v = (co < CoutReal && ci < CiReal) ? w[((size_t)co * CiReal + ci) * ntaps + tap] * ws : 0.f;
        _Float16 a, b; split1(v, a, b); hi[j] = a; lo[j] = b;
    }
    const size_t base = ((size_t)tap * (Ci / 8) + chunk) * 2 * Cout + co;
    out[base] = *reinterpret_cast<uint4*>(&hi);
    out[base + Cout] = *reinterpret_cast<uint4*>(&lo);
}

// ---- input conversion: fp32 (NCHW or CB8) * style -> SPLIT; channels >= Creal are zero padding (Cin = 3 -> 8) -------
__global__ void to_split_kernel(const float* __restrict__ src, int cb8, const float* __restrict__ scale, size_t scale_stride_n,
                                uint4* __restrict__ dst, int C, int Creal, int HW)
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
        const float* s = src + ((size_t)n * Creal + cb * 8) * HW + p;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (cb * 8 + c < Creal) ? s[(size_t)c * HW] : 0.f;
    }
    h8 hi, lo;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float sc = (scale && cb * 8 + c < Creal) ? scale[n * scale_stride_n + cb * 8 + c] : 1.f;
        _Float16 a, b; split_scaled(v[c], sc, a, b); hi[c] = a; lo[c] = b;
    }
    const size_t plane = (size_t)(C / 8) * HW;
    uint4* d = dst + (size_t)n * 2 * plane + (size_t)cb * HW + p;
    d[0] = *reinterpret_cast<uint4*>(&hi);
    d[plane] = *reinterpret_cast<uint4*>(&lo);
}

__device__ uint4 g_zero16[1];      // DMA source of zero-filled (out-of-image / padding) patch slots

// LDS-DMA of one wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1 KB).  Inline asm so
// that hipcc does not serialise the ds_reads of the other buffer behind it (waits are the explicit vmcnt(0) per stage).
__device__ __forceinline__ void dma64(const uint4* gsrc, uint4* lds_dst_uniform)
{
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds_dst_uniform);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(d) : "memory");
}

// saddr form: 64 lanes x 16 B from (wave-uniform base + per-lane 32-bit byte offset): one VGPR per distinct lane pattern instead of a
// 64-bit address pair per source.  The masked variant skips the lanes whose bit in `mask` is clear (their LDS slots keep what they
// hold: the caller zero-fills halo slots once per block).
__device__ __forceinline__ void dma64s(const void* base_uniform, unsigned voff_bytes, uint4* lds_dst_uniform)
{
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds_dst_uniform);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(base_uniform), "s"(d) : "memory");
}
__device__ __forceinline__ void dma64s_masked(const void* base_uniform, unsigned voff_bytes, unsigned long long mask, uint4* lds_dst_uniform)
{
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds_dst_uniform);
    unsigned keep; unsigned long long save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_and_saveexec_b64 %1, %5\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(save) : "v"(voff_bytes), "s"(base_uniform), "s"(d), "s"(mask) : "memory", "scc");
}

// ---- conv ----------------------------------------------------------------------------------------------------
struct Conv2Args {
    const uint4* x; size_t x_stride_n;        // SPLIT input (hi plane, lo plane), per-n stride in uint4 units
    const uint4* wp;                          // split prepacked weights [tap][Cin/8][hi|lo][Cout]
    const float* out_scale; size_t out_scale_stride_n;     // per-cout multiplier (demodulation) or null (= 1)
    const float* bias; size_t bias_stride_n;               // per-cout bias or null
    float* y_f32; size_t y_f32_stride_n; int OH, OW;       // fp32 CB8 output (T buffer / x_out) or null
    float* y_nchw; size_t y_nchw_stride_n;                 // fp32 NCHW output or null
    uint4* y_split; size_t y_split_stride_n;               // SPLIT output scaled by next_scale (null = 1), or null
    int y_split_mx;                                        // 1: the lo plane of y_split holds fp8 records (R3D_FMT_SPLIT_MX) for an f16mx consumer
    // y_split as ONE PART of a channel concatenation (r3d_conv_forward_cat, round 6): the destination has y_cat_chunks 8-channel chunks per plane (0 = this conv's
    // own Cout / 8), this conv's couts start at chunk y_cat_off (even), and every value is multiplied by its pixel's y_mask (or 1 - y_mask) before the
    // consumer's multiplier -- `torch.cat([.., x_torso * (1 - alpha)], dim=1)` written by the conv that produces x_torso
    const float* y_mask; int y_mask_invert; int y_cat_chunks, y_cat_off;
    const float* next_scale; size_t next_scale_stride_n;
    unsigned* y_absmax;                                    // [N] max |activated output| (uint bits of a non-negative float) or null
    const float* wrgb; size_t wrgb_stride_n; float* rgb_partial; size_t rgbp_stride_n;   // toRGB partials [Cout/128][3][OH*OW] or null
    int Cin, Cout, CoutReal, H, W, nphase;    // Cout: padded to 128 (weight layout); CoutReal: channels that exist in the outputs
    int act; float act_slope, act_gain, clamp; // act: leaky-relu(slope) * gain after the bias; clamp < 0: off
    unsigned long long* clk;                  // prof_clock_slot(R3D_PROF_CONV) or null (conv3x3_dma_block samples it)
    int order;                                // conv3x3_dma_block: 0 = (tile, cout tile) = (blockIdx.x, blockIdx.y); 1 / 2 = the cout tiles of a pixel tile adjacent in dispatch order on one XCD
    // conv1x1_blend_f16x3_kernel: the operand x = cat([bl_a * m, bl_b * (1 - m)]) * in_scale is computed while it is staged, never materialised --
    // bl_a / bl_b fp32 channel-blocked [N][C/8][H*W][8] (Ca + Cb = Cin), bl_mask [N][H*W], bl_scale [N] x stride: the layer's folded in-multiplier (uniform
    // over the channels for a plain conv layer: element 0 is read)
    const float* bl_a; const float* bl_b; const float* bl_mask; const float* bl_scale; size_t bl_scale_stride_n; int bl_Ca;
    ConvPhase ph[4];
};

// Epilogue of a 128-cout x 16x16-pixel block held as acc[2][NT] per wave: demodulation * acc (+ bias -> lrelu * gain ->
// clamp) -> any of {fp32 channel-blocked, fp32 NCHW, SPLIT scaled by the next layer's styles} + toRGB partial sums.
// toRGB partials: one plane per 64-cout half (index m0 / 64 + wm); rgb_finalize_kernel adds them up.
// The per-cout vectors (out_scale, bias, next_scale, 3 toRGB weight rows) of the block's 128 couts are staged in LDS ONCE, after
// the main loop: loaded from global inside the group loop they put an `s_waitcnt vmcnt(0)` -- which on gfx9 also waits for every
// store issued before it -- in front of each of the 16 groups (48 serialised memory round trips = the 29 k-cycle epilogue that
// round 1 measured); from LDS the loop has no global loads, so its stores stream out back to back.
// The caller has passed a __syncthreads() after its last LDS read; `ev` may alias the main loop's buffers.
static constexpr int EV_STRIDE = BLOCK_M;              // floats per staged vector
#ifndef R3D_EPI_WIDE
#define R3D_EPI_WIDE 1        // A/B switch: 0 = the 8-byte SPLIT stores of rounds 1-5 (same values, same bytes)
#endif
// PIXMAP = 1 (experiment): acc[mt][nt] holds column parity nt of the column pairs -- lane li <-> (row 4 wn + li / 8, columns 2 (li % 8) + nt).
// PIXMAP != 0 (conv_wino_f16x3_kernel): the accumulators carry the transformed operands' factor 1/4 (r3d_sr_wino.h); 2 = the ordinary pixel map.
template <bool FULL_EPI, int WN, int NT, int PIXMAP = 0, bool CAT = false>     // CAT: the concatenation-part output (y_mask, y_cat_*) is compiled in -- the 1x1 conv only: the 3x3 kernels sit at their register limit
__device__ __forceinline__ void conv_epilogue(const Conv2Args& a, const ConvPhase& ph, int n, f32x16 (&acc)[2][NT],
                                              int i0, int j0, int m0, float* ev)
{
    // The instantiations of this epilogue (tile shapes, MX) must agree bit for bit -- a layer may run on 16x16 or 8x16 tiles depending on
    // the batch size -- so nothing here is left to the compiler's contraction choices: every fma is spelled out.
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, h = lane >> 5;
    const int prow = li >> 4, pcol = ((li & 15) - 2 * prow) & 15;
    const int row0 = wn * 2 * NT;
    const bool do_rgb = FULL_EPI && a.rgb_partial != nullptr;
    {   // stage: vector v of cout c -> ev[v * EV_STRIDE + c]; v = 0 out_scale (1), 1 bias (0), 2 next_scale (1), 3..5 toRGB rows (0)
        constexpr int NTHR = 128 * WN;
        for (int e = threadIdx.x; e < 6 * BLOCK_M; e += NTHR) {
            const int v = e / BLOCK_M, c = e - v * BLOCK_M, co = m0 + c;
            float val = (v == 0 || v == 2) ? 1.f : 0.f;
            if (co < a.CoutReal) {
                if (v == 0) { if (a.out_scale) val = a.out_scale[(size_t)n * a.out_scale_stride_n + co]; }
                else if (v == 1) { if (FULL_EPI && a.bias) val = a.bias[(size_t)n * a.bias_stride_n + co]; }
                else if (v == 2) { if (FULL_EPI && a.y_split && a.next_scale) val = a.next_scale[(size_t)n * a.next_scale_stride_n + co]; }
                else if (do_rgb) val = a.wrgb[(size_t)n * a.wrgb_stride_n + (size_t)(v - 3) * a.CoutReal + co];
            }
            if (PIXMAP && v == 0) val *= 4.0f;
            ev[e] = val;
        }
        __syncthreads();
    }
    float vmax = 0.f;
    unsigned rec_h[NT], rec_l[NT];                        // y_split_mx: the even chunk's record dwords wait for the odd chunk's (one 8-byte store)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { rec_h[nt] = 0u; rec_l[nt] = 0u; }
    float rgbp[NT][3];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { rgbp[nt][0] = 0.f; rgbp[nt][1] = 0.f; rgbp[nt][2] = 0.f; }
    const size_t oplane = (size_t)(CAT && a.y_cat_chunks ? a.y_cat_chunks : a.CoutReal >> 3) * a.OH * a.OW;
    // Address arithmetic in 32 bits with the per-channel-plane part on the scalar unit: plane index and plane stride are wave-uniform,
    // the pixel offset is computed once per N tile (the 64-bit multiplies per store group were a third of the epilogue's VALU cycles).
    // host: (Cout/8) * OH * OW * 8 < 2^32 for every layer that gets here (checked in launch_conv2)
    const unsigned cs = (unsigned)a.OH * (unsigned)a.OW;
    const int wm_u = __builtin_amdgcn_readfirstlane(wm);
    unsigned p0[NT];
    bool inside_nt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int i = PIXMAP == 1 ? i0 + 4 * wn + (li >> 3) : i0 + row0 + nt * 2 + prow, j = PIXMAP == 1 ? j0 + 2 * (li & 7) + nt : j0 + pcol;
        inside_nt[nt] = i < ph.outH && j < ph.outW;
        p0[nt] = (unsigned)(i * ph.oy_mul + ph.oy_add) * (unsigned)a.OW + (unsigned)(j * ph.ox_mul + ph.ox_add);
    }
    float* Yf = a.y_f32 ? a.y_f32 + (size_t)n * a.y_f32_stride_n + ph.out_off : nullptr;
    float* Yn = (FULL_EPI && a.y_nchw) ? a.y_nchw + (size_t)n * a.y_nchw_stride_n : nullptr;
    uint4* Ys = (FULL_EPI && a.y_split) ? a.y_split + (size_t)n * a.y_split_stride_n + (CAT ? (size_t)a.y_cat_off * a.OH * a.OW : 0) : nullptr;   // (y_cat_off even: record pairs stay aligned)
    float ym[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ym[nt] = 1.f;
    const bool masked = CAT && FULL_EPI && a.y_mask != nullptr;
    if (masked) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float mk = inside_nt[nt] ? a.y_mask[(size_t)n * cs + p0[nt]] : 0.f;
            ym[nt] = a.y_mask_invert ? 1.0f - mk : mk;
        }
    }
    const bool want_max = FULL_EPI && a.y_absmax != nullptr;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cu = 64 * wm_u + 32 * mt + 8 * g;                // wave-uniform part of the cout index
            const int cl = cu + 4 * h, co = m0 + cl;
            if (co >= a.CoutReal) continue;                           // channels that only exist as weight padding
            const unsigned pbase = (unsigned)((m0 + cu) >> 3) * cs;   // first element of this group's 8-channel plane (scalar)
            const float4 d4 = *reinterpret_cast<const float4*>(ev + cl);
            const float4 b4 = *reinterpret_cast<const float4*>(ev + EV_STRIDE + cl);
            const float4 s4 = *reinterpret_cast<const float4*>(ev + 2 * EV_STRIDE + cl);
            const float4 w0 = *reinterpret_cast<const float4*>(ev + 3 * EV_STRIDE + cl);
            const float4 w1 = *reinterpret_cast<const float4*>(ev + 4 * EV_STRIDE + cl);
            const float4 w2 = *reinterpret_cast<const float4*>(ev + 5 * EV_STRIDE + cl);
            const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
            constexpr bool WIDE = R3D_EPI_WIDE && (NT % 2 == 0);    // (the 8-row tiles have one N tile per wave: 8-byte stores)
            uint2 hiw[NT], low[NT];
            unsigned xh8w[NT], xl8w[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool inside = inside_nt[nt];
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t;
                    if (FULL_EPI) {
                        t = __builtin_fmaf(acc[mt][nt][4 * g + r], dv[r], bv[r]);
                        if (a.act) t = (t < 0.f ? t * a.act_slope : t) * a.act_gain;
                        if (a.clamp >= 0.f) t = fminf(fmaxf(t, -a.clamp), a.clamp);
                    }
                    else t = acc[mt][nt][4 * g + r] * dv[r];
                    v[r] = t;
                }
                if (do_rgb) {
                    rgbp[nt][0] = __builtin_fmaf(v[3], w0.w, __builtin_fmaf(v[2], w0.z, __builtin_fmaf(v[1], w0.y, __builtin_fmaf(v[0], w0.x, rgbp[nt][0]))));
                    rgbp[nt][1] = __builtin_fmaf(v[3], w1.w, __builtin_fmaf(v[2], w1.z, __builtin_fmaf(v[1], w1.y, __builtin_fmaf(v[0], w1.x, rgbp[nt][1]))));
                    rgbp[nt][2] = __builtin_fmaf(v[3], w2.w, __builtin_fmaf(v[2], w2.z, __builtin_fmaf(v[1], w2.y, __builtin_fmaf(v[0], w2.x, rgbp[nt][2]))));
                }
                if (inside) {
                if (want_max) vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                const unsigned pix = pbase + p0[nt];
                if (Yf) *reinterpret_cast<float4*>(Yf + (size_t)pix * 8 + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
                if (Yn) {
                    float* yn = Yn + (size_t)(unsigned)(m0 + cu) * cs + (size_t)((unsigned)(4 * h) * cs + p0[nt]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) yn[(size_t)r * cs] = v[r];
                }
                }
                if (Ys) {
                    h4 hi, lo;
                    float lf[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sv_ = as_rounded(masked ? v[r] * ym[nt] * sv[r] : v[r] * sv[r]);      // (blend_cat_to_split_kernel's order: the same bits)
                        const _Float16 x0 = (_Float16)sv_;
                        lf[r] = sv_ - (float)x0;
                        hi[r] = x0; lo[r] = (_Float16)lf[r];
                    }
                    if constexpr (WIDE) {
                    // 16-byte stores (round 6): the two lane halves hold the two 8-byte halves of a pixel's 16-byte word; the values wait for the N-tile
                    // pair's v_permlane32_swap below (the `inside` test moves to the store)
                    hiw[nt] = *reinterpret_cast<uint2*>(&hi);
                    if (a.y_split_mx) {
                        xh8w[nt] = pack4_x8((float)hi[0] * kMxXh, (float)hi[1] * kMxXh, (float)hi[2] * kMxXh, (float)hi[3] * kMxXh);
                        xl8w[nt] = pack4_x8(lf[0] * kMxXl, lf[1] * kMxXl, lf[2] * kMxXl, lf[3] * kMxXl);
                    } else low[nt] = *reinterpret_cast<uint2*>(&lo);
                    } else {
                    if (!inside) continue;
                    const unsigned pix = pbase + p0[nt];
                    uint2* dst = reinterpret_cast<uint2*>(Ys + pix) + h;
                    dst[0] = *reinterpret_cast<uint2*>(&hi);
                    if (a.y_split_mx) {
                        // fp8 records in place of the lo words (see "f16mx" at the top): lo chunk 2G <- xh8, 2G+1 <- xl8 of the 16 channels of
                        // group G; this lane's 4 couts are bytes [8 (chunk & 1) + 4 h, +4) of both records
                        const unsigned xh8 = pack4_x8((float)hi[0] * kMxXh, (float)hi[1] * kMxXh, (float)hi[2] * kMxXh, (float)hi[3] * kMxXh);
                        const unsigned xl8 = pack4_x8(lf[0] * kMxXl, lf[1] * kMxXl, lf[2] * kMxXl, lf[3] * kMxXl);
                        const unsigned c8 = (unsigned)(m0 + cu) >> 3;
                        if (!(c8 & 1u)) { rec_h[nt] = xh8; rec_l[nt] = xl8; }          // (g even; g + 1 is the group's other chunk, same mt, same pixels)
                        else {
                            uint2* rec = reinterpret_cast<uint2*>(Ys + oplane + (size_t)((c8 & ~1u) * cs + p0[nt])) + h;      // dwords 2 h, 2 h + 1
                            rec[0] = make_uint2(rec_h[nt], xh8);
                            rec[2 * (size_t)cs] = make_uint2(rec_l[nt], xl8);
                        }
                    } else {
                        dst[2 * oplane] = *reinterpret_cast<uint2*>(&lo);
                    }
                    }
                }
            }
            if constexpr (WIDE) if (Ys) {
                // N tiles (nt, nt + 1) = two pixels per lane: lane half h = 0 hands its nt + 1 values to its partner (lane + 32) and receives the partner's nt
                // values -- v_permlane32_swap: new a = {a[0..31], b[0..31]}, new b = {a[32..63], b[32..63]} -- so every lane holds all 8 couts of ONE pixel
                // (nt + h) and stores 16 bytes: half the store instructions, whole 16-byte words (the epilogue's store tail is issue bound; same bytes, same values)
                const unsigned c8 = (unsigned)(m0 + cu) >> 3;
#pragma unroll
                for (int nt = 0; nt < NT; nt += 2) {
                    const bool ins = h ? inside_nt[nt + 1] : inside_nt[nt];
                    const unsigned pp = h ? p0[nt + 1] : p0[nt];
                    auto swp = [](unsigned x, unsigned y, unsigned& ox, unsigned& oy) { const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); ox = r[0]; oy = r[1]; };
                    uint4 w;
                    swp(hiw[nt].x, hiw[nt + 1].x, w.x, w.z); swp(hiw[nt].y, hiw[nt + 1].y, w.y, w.w);
                    if (ins) Ys[pbase + pp] = w;
                    if (a.y_split_mx) {
                        if (!(c8 & 1u)) { rec_h[nt] = xh8w[nt]; rec_l[nt] = xl8w[nt]; rec_h[nt + 1] = xh8w[nt + 1]; rec_l[nt + 1] = xl8w[nt + 1]; }   // (g even: wait for the group's other chunk)
                        else {
                            uint4 rh, rl;                              // record dwords (2 h, 2 h + 1) = (even chunk, odd chunk) of this lane half's 4 couts
                            swp(rec_h[nt], rec_h[nt + 1], rh.x, rh.z); swp(xh8w[nt], xh8w[nt + 1], rh.y, rh.w);
                            swp(rec_l[nt], rec_l[nt + 1], rl.x, rl.z); swp(xl8w[nt], xl8w[nt + 1], rl.y, rl.w);
                            uint4* rec = Ys + oplane + (size_t)((c8 & ~1u) * cs + pp);
                            if (ins) { rec[0] = rh; rec[cs] = rl; }
                        }
                    } else {
                        swp(low[nt].x, low[nt + 1].x, w.x, w.z); swp(low[nt].y, low[nt + 1].y, w.y, w.w);
                        if (ins) Ys[oplane + pbase + pp] = w;
                    }
                }
            }
        }
    if (want_max) {
        // one atomic per BLOCK (the staged vectors in `ev` are no longer read): atomics on one word serialise in L2 at ~12 ns each, and
        // all blocks of a launch reach this point together -- one per wave (4 096 per launch) showed as +18 us per kernel
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
        __syncthreads();
        if (lane == 0) ev[wave] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = ev[0];
            for (int w = 1; w < 2 * WN; ++w) m = fmaxf(m, ev[w]);
            if (m > 0.f) atomicMax(a.y_absmax + n, __float_as_uint(m));
        }
    }
    if (do_rgb) {
        // add the two lane halves (disjoint couts); each 64-cout wave row stores its own partial plane
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int o = 0; o < 3; ++o) rgbp[nt][o] += __shfl_xor(rgbp[nt][o], 32);
        if (h == 0) {
            float* P = a.rgb_partial + (size_t)n * a.rgbp_stride_n + (size_t)(m0 / 64 + wm) * 3 * a.OH * a.OW;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int i = PIXMAP == 1 ? i0 + 4 * wn + (li >> 3) : i0 + row0 + nt * 2 + prow, j = PIXMAP == 1 ? j0 + 2 * (li & 7) + nt : j0 + pcol;
                if (i < ph.outH && j < ph.outW) {
                    const int oy = i * ph.oy_mul + ph.oy_add, ox = j * ph.ox_mul + ph.ox_add;
#pragma unroll
                    for (int o = 0; o < 3; ++o) P[(size_t)o * a.OH * a.OW + (size_t)oy * a.OW + ox] = rgbp[nt][o];
                }
            }
        }
    }
}

// One block: 128 couts x 16x16 pixels, WN x 2 waves; wave (wm, wn) owns couts [64wm, +64) and pixel rows
// [2*NT*wn, +2*NT) as 2 x NT MFMA tiles (WN=2,NT=4: 4 waves x 128 accumulators; WN=4,NT=2: 8 waves x 64 accumulators
// = 4 waves/SIMD at 2 blocks/CU).  Per stage = 16 input channels (two 8-channel blocks = one K=16 MFMA step per tap):
// the halo patch (B operand, hi+lo) sits in LDS for all taps of the stage; the weights (A operand) go through LDS in
// sub-stages of SUB taps.  All global loads are register-prefetched one (sub-)stage ahead, and the weight loads are
// issued BEFORE the patch loads, so the compiler's counted s_waitcnt vmcnt(N) at the weight ds_write leaves the slow
// (MALL/HBM) patch loads in flight under the MFMAs.
template <int NTAPS, bool FULL_EPI, int WN, int NT>
__device__ __forceinline__ void conv2_block(const Conv2Args& a, const ConvPhase& ph, int n, uint4* lds)
{
    static_assert(WN * NT == 8, "16 pixel rows per block");
    constexpr int NTHR = 128 * WN;
    constexpr int SUB = NTAPS >= 9 ? 3 : (NTAPS >= 2 ? 2 : 1);        // taps per weight sub-stage
    constexpr int NSUB = (NTAPS + SUB - 1) / SUB;                     // sub-stages per stage
    constexpr int B_ELEMS = 2 * 2 * F_PATCH_PIX;                      // uint4: planes x chunks x pixels = 1296
    constexpr int NPF = (B_ELEMS + NTHR - 1) / NTHR;
    constexpr int A_ELEMS = SUB * 2 * 256;                            // uint4 per sub-stage: (tap, half-chunk) x (cout, hi|lo)
    constexpr int A_BUF = 3 * 2 * 256;                                // uint4 per weight buffer (sized for SUB = 3)
    constexpr int NAR = A_ELEMS / NTHR;
    uint4* patchB = lds;                                              // [plane][chunk][324]
    uint4* bufA = lds + B_ELEMS;                                      // 2 x [SUB][2 chunks][hi|lo][128 couts], filled by LDS-DMA
    const int tiles_x = (ph.outW + F_TILE_W - 1) / F_TILE_W;
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int i0 = ty * F_TILE_H, j0 = tx * F_TILE_W;
    const int m0 = blockIdx.y * BLOCK_M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, h = lane >> 5;
    const int nchunks = a.Cin >> 3;
    const size_t plane = (size_t)nchunks * a.H * a.W;
    const uint4* X = a.x + (size_t)n * a.x_stride_n;
    const uint4* WP = a.wp;

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // N-tile pixel of this lane: rows (2nt, 2nt+1) of the wave's rows; the odd row's columns are rotated by 2 so
    // that each ds_read_b128 lane group covers 16 distinct 16-byte LDS slots with the 18-pixel patch row stride
    const int prow = li >> 4, pcol = ((li & 15) - 2 * prow) & 15;
    const int row0 = wn * 2 * NT;
    int boff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) boff[nt] = (row0 + nt * 2 + prow + 1) * F_PATCH_W + (pcol + 1) + h * F_PATCH_PIX;
    const int aoff = h * 256 + 64 * wm + li;                          // [ts][hc = h][hi|lo][128 couts]: + ts*512 + mt*32 (+128 for lo)

    const int chunk_stride = a.H * a.W;
    uint4 pf[NPF];
    unsigned pf_off[NPF];                                             // per-thread patch element -> offset within a 2-chunk slab
    unsigned pf_valid = 0;                                            // bit k: element k is inside the image (else zero fill)
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int e = threadIdx.x + NTHR * k;
        unsigned off = 0;
        if (e < B_ELEMS) {
            const int pl = e / (2 * F_PATCH_PIX);
            const int rem = e - pl * (2 * F_PATCH_PIX);
            const int c = rem / F_PATCH_PIX, pp = rem - c * F_PATCH_PIX;
            const int py = pp / F_PATCH_W, px = pp - py * F_PATCH_W;
            const int iy = i0 + py - 1, ix = j0 + px - 1;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                off = (unsigned)(pl * plane) + (unsigned)(c * chunk_stride + iy * a.W + ix);
                pf_valid |= 1u << k;
            }
        }
        pf_off[k] = off;
    }
    auto load_patch = [&](int c0) {                                   // global -> registers (written to LDS one stage later)
        const uint4* Xs = X + (size_t)c0 * chunk_stride;              // wave-uniform base + 32-bit per-lane offset
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            // always issue exactly NPF loads (the counted s_waitcnt vmcnt(NPF) below relies on it); halo -> zero
            uint4 v = Xs[pf_off[k]];
            if (!(pf_valid & (1u << k))) v = make_uint4(0, 0, 0, 0);
            pf[k] = v;
        }
    };
    // weights: global -> LDS by DMA (global_load_lds_dwordx4: LDS address = wave-uniform base + lane*16; the
    // sub-stage image is linear in e = tid + NTHR*k, so a wave's 64 lanes fill one contiguous KB); no VGPRs, no ds_write
    const int tid_seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);      // 128-cout segment of this wave
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned tid_lo = threadIdx.x & 127;
    auto dma_weights = [&](int c0, int sub, uint4* dstA) {
#pragma unroll
        for (int k = 0; k < NAR; ++k) {
            const int seg = (NTHR >> 7) * k + tid_seg, t = sub * SUB + (seg >> 2), hc = (seg >> 1) & 1, hl = seg & 1;   // wave-uniform
            if (t < NTAPS) {
                const uint4* src = WP + (((size_t)ph.widx[t] * nchunks + (c0 + hc)) * 2 + hl) * a.Cout + m0;     // uniform
                // inline asm: hipcc would otherwise put s_waitcnt vmcnt(0) in front of every ds_read while an LDS-DMA it
                // knows about is in flight (no alias info inside one LDS array); waits for these DMAs are the explicit
                // counted s_waitcnt vmcnt below (cdna_hip_programming.md section 5.7)
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)(dstA + NTHR * k + 64 * wave_u));
                const uint4* gsrc = src + tid_lo;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
            }
        }
    };

    int g = 0;                                                        // global sub-stage counter (weight buffer parity)
    dma_weights(0, 0, bufA);
    load_patch(0);
    for (int c0 = 0; c0 < nchunks; c0 += 2) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub, ++g) {
            if (sub == 0) {
                __syncthreads();                                       // previous stage's patch reads are done
#pragma unroll
                for (int k = 0; k < NPF; ++k) {
                    const int e = threadIdx.x + NTHR * k;
                    if (e < B_ELEMS) patchB[e] = pf[k];
                }
            }
            // weights of this sub-stage have landed (own DMAs) -> barrier -> everybody's have
            // (the NPF patch loads of the next stage were issued AFTER these DMAs during sub-stage 0: leave them in flight)
            if (NSUB > 1 && sub == 1 && c0 + 2 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPF) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const uint4* curA = bufA + (g & 1) * A_BUF;
            uint4* nxtA = bufA + ((g + 1) & 1) * A_BUF;
            // next weights first, then (once per stage) the next patch: in-order vmcnt lets the weights be waited for
            // while the slow patch loads are still in flight
            {
            if (sub + 1 < NSUB) dma_weights(c0, sub + 1, nxtA);
            else if (c0 + 2 < nchunks) dma_weights(c0 + 2, 0, nxtA);
            if (sub == 0 && c0 + 2 < nchunks) load_patch(c0 + 2);
            }
#pragma unroll
            for (int ts = 0; ts < SUB; ++ts) {
                const int t = sub * SUB + ts;
                if (t < NTAPS) {
                    h8 ah[2], al[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        uint4 q0 = curA[ts * 512 + aoff + mt * 32], q1 = curA[ts * 512 + aoff + mt * 32 + 128];
                        ah[mt] = *reinterpret_cast<h8*>(&q0); al[mt] = *reinterpret_cast<h8*>(&q1);
                    }
                    const int toff = ph.dy[t] * F_PATCH_W + ph.dx[t];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        uint4 r0 = patchB[boff[nt] + toff];
                        uint4 r1 = patchB[boff[nt] + toff + 2 * F_PATCH_PIX];
                        const h8 bh = *reinterpret_cast<h8*>(&r0), bl = *reinterpret_cast<h8*>(&r1);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    conv_epilogue<FULL_EPI, WN, NT, 0, NTAPS == 1>(a, ph, n, acc, i0, j0, m0, reinterpret_cast<float*>(lds));
}

// ---- plain 3x3 conv, LDS-DMA pipeline ------------------------------------------------------------------------------
// Same tile as conv2_block<9, ., 4, 2> (128 couts x 16x16 pixels, 8 waves x (64 couts x 64 px)) but every byte reaches LDS
// by global_load_lds (no staging VGPRs, no ds_write): the register-staged version spilled its third patch prefetch and
// waited for it right after issuing it, once per stage.
//   * patch (B operand, 16 channels, hi|lo): double-buffered, 21 x 64-slot DMA segments (out-of-image slots read a zero
//     block); patch(s+1) is issued as soon as stage s-1 is finished with that buffer, two sub-stages ahead of its first use.
//   * weights (A operand): sub-stages of TWO taps ([ts][chunk][hi|lo][128 couts] = 16 KB, two DMAs per wave), double-
//     buffered.  The tap sequence runs across stages (9 taps = 4.5 sub-stages), so the loop is unrolled over 9 sub-stages
//     = 2 stages; an odd last stage skips the missing taps.
//   * one barrier per sub-stage (24 MFMAs per wave): wait for this sub-stage's weights (counted vmcnt: a patch DMA issued
//     after them stays in flight), barrier, issue the next sub-stage's DMAs, compute.
// LDS: 2 x 21.5 KB patch + 2 x 16 KB weights = 75.8 KB -> 2 blocks/CU.
static constexpr int P_SEGS = 21, P_BUF = P_SEGS * 64;               // 1344 uint4 per patch buffer (1296 used)
static constexpr int W2_BUF = 2 * 2 * 2 * 128;                        // 1024 uint4 per 2-tap weight sub-stage
static constexpr int F3_LDS_UINT4 = 2 * P_BUF + 2 * W2_BUF;           // 4736 uint4 = 75.8 KB

typedef int i8v __attribute__((ext_vector_type(8)));

// TH = pixel rows per tile: 16 (the shape above), or 8 for layers whose 16x16 tiling leaves most of the chip idle (the 128^2 layers of
// to_plane_cnn: 128 blocks for 512 block slots): 128 couts x 8x16 px per block, 8 waves x (64 couts x 32 px), twice the blocks, same
// K order per output (bit-identical results).  LDS 2 x 12.3 KB patch + 2 x 16 KB weights = 57 KB.
template <bool FULL_EPI, bool MX = false, int TH = F_TILE_H>
__device__ __forceinline__ void conv3x3_dma_block(const Conv2Args& a, const ConvPhase& ph, int n, uint4* lds)
{
    static_assert(TH == 16 || TH == 8, "tile rows");
    constexpr int WN = 4, NT = TH / 8;
    constexpr int PATCH_PIX = (TH + 2) * F_PATCH_W;                  // 324 | 180
    constexpr int PSEGS = (4 * PATCH_PIX + 63) / 64;                 // 21 | 12 DMA segments of 64 slots
    constexpr int PBUF = PSEGS * 64;
    constexpr int KMAX = (PSEGS + 7) / 8;                             // patch DMAs per wave and stage: 3 | 2 (waves >= PSEGS - 8 (KMAX - 1): one less)
    static_assert(PBUF == (TH == 16 ? P_BUF : 768), "patch buffer");
    uint4* pbuf = lds;                                                // [2][PBUF]
    uint4* wbuf = lds + 2 * PBUF;                                     // [2][W2_BUF]
    const int tiles_x = (ph.outW + F_TILE_W - 1) / F_TILE_W;
    int tile = blockIdx.x, cgi = blockIdx.y;
    if (a.order) {
        // Workgroups are dispatched in linear order (x fastest), round-robin over the 8 XCDs.  With (tile, cout tile) = (x, y) the second
        // cout tile of a 256-cout layer re-reads every input patch ~0.1 ms after the first: by then it has left the XCD's 4 MB L2 and
        // comes from HBM again, and horizontally adjacent tiles sit on different XCDs, so the 288-byte patch rows (18 pixels, not
        // line-aligned) are fetched as whole 128-byte lines by each of them: measured 318 MB for the 137 MB of block0.conv1.  Here the cout
        // tiles of one pixel tile are consecutive slots of ONE XCD, and (order 2) an XCD owns a contiguous band of tiles, so halo
        // lines and the second cout tile hit its L2: 172 MB (order 1, tiles interleaved over the XCDs: 233 MB).  Time-neutral: the
        // kernel is MFMA / power bound.  (host: only when the tile count is a multiple of 8)
        const int G = gridDim.y, b = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = b & 7, slot = b >> 3;
        cgi = slot % G;
        const int t = slot / G;
        tile = a.order == 1 ? t * 8 + xcd : xcd * (gridDim.x >> 3) + t;
    }
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int i0 = ty * TH, j0 = tx * F_TILE_W;
    const int m0 = cgi * BLOCK_M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, h = lane >> 5;
    const int nchunks = a.Cin >> 3, nst = a.Cin >> 4;
    const int chunk_stride = a.H * a.W;
    const size_t plane = (size_t)nchunks * chunk_stride;
    const uint4* X = a.x + (size_t)n * a.x_stride_n;
    const uint4* WP = a.wp + m0 + (wave_u & 1) * 64 + lane;           // this wave's 64-cout half of the 128-cout rows

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int prow = li >> 4, pcol = ((li & 15) - 2 * prow) & 15;
    const int row0 = wn * 2 * NT;
    int boff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) boff[nt] = (row0 + nt * 2 + prow + 1) * F_PATCH_W + (pcol + 1) + h * PATCH_PIX;
    const int aoff = h * 256 + 64 * wm + li;                          // [ts][chunk = h][hi|lo][128]: + ts*512 + mt*32 (+128 for lo)

    // patch DMA: wave w fills segments w, w+8, w+16 (< PSEGS); slot e = 64*seg + lane -> (plane, chunk, (TH+2)x18 pixel)
    unsigned pf_off[3];
    unsigned pf_valid = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int e = (8 * k + wave_u) * 64 + lane;
        unsigned off = 0;
        if (e < 4 * PATCH_PIX) {
            const int pl = e / (2 * PATCH_PIX), rem = e - pl * (2 * PATCH_PIX);
            const int c = rem / PATCH_PIX, pp = rem - c * PATCH_PIX;
            const int py = pp / F_PATCH_W, px = pp - py * F_PATCH_W;
            const int iy = i0 + py - 1, ix = j0 + px - 1;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                off = (unsigned)(pl * plane) + (unsigned)(c * chunk_stride + iy * a.W + ix);
                pf_valid |= 1u << k;
            }
        }
        pf_off[k] = off;
    }
    const bool three = wave_u < PSEGS - 8 * (KMAX - 1);               // these waves issue KMAX patch DMAs per stage, the others KMAX - 1
    // MX (register budget: 16 more operand VGPRs than f16x3): saddr-form DMAs -- no per-source 64-bit address pairs -- with the
    // out-of-image slots zero-filled ONCE here (both buffers) and skipped by every stage's DMA (exec mask) instead of reading a zero block
    unsigned long long pmask[3] = {0, 0, 0};
    unsigned pf_boff[3] = {0, 0, 0};
    const unsigned lane16 = (unsigned)lane * 16u;
    if constexpr (MX) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            pmask[k] = __ballot((pf_valid >> k) & 1u);
            pf_boff[k] = pf_off[k] * 16u;
            if ((k < KMAX - 1 || three) && !((pf_valid >> k) & 1u)) {
                pbuf[64 * (8 * k + wave_u) + lane] = make_uint4(0, 0, 0, 0);
                pbuf[PBUF + 64 * (8 * k + wave_u) + lane] = make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();                                              // zero fill done before any DMA is in flight
    }
    auto dma_patch = [&](int st, uint4* dst) {
        const uint4* Xs = X + (size_t)(2 * st) * chunk_stride;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < KMAX - 1 || three) {
                if constexpr (MX) {
                    // (a fully out-of-image segment still issues one DMA -- of the zero block -- so that the counted vmcnt waits hold)
                    if (pmask[k]) dma64s_masked(Xs, pf_boff[k], pmask[k], dst + 64 * (8 * k + wave_u));
                    else dma64(g_zero16, dst + 64 * (8 * k + wave_u));
                } else {
                    dma64((pf_valid & (1u << k)) ? Xs + pf_off[k] : g_zero16, dst + 64 * (8 * k + wave_u));
                }
            }
    };
    // weights of linear tap T (stage T / 9, tap T % 9) -> slot ts of a sub-stage buffer; wave w moves (chunk, hi|lo, cout half)
    // = bits (2, 1, 0) of w for both taps of the sub-stage
    auto dma_weights2 = [&](int T0, uint4* dst) {
#pragma unroll
        for (int ts = 0; ts < 2; ++ts) {
            const int T = T0 + ts;
            if (T < 9 * nst) {
                const int st = T / 9, t = T - 9 * st;
                const int hc = (wave_u >> 2) & 1, hl = (wave_u >> 1) & 1;
                if constexpr (MX)
                    dma64s(a.wp + m0 + (wave_u & 1) * 64 + (((size_t)t * nchunks + (2 * st + hc)) * 2 + hl) * a.Cout, lane16,
                           dst + ts * 512 + wave_u * 64);
                else
                    dma64(WP + (((size_t)t * nchunks + (2 * st + hc)) * 2 + hl) * a.Cout, dst + ts * 512 + wave_u * 64);
            } else if (MX && ts == 1 && T0 < 9 * nst) {
                dma64(g_zero16, dst + ts * 512 + wave_u * 64);       // the fp8 pair-MFMA reads both taps: a missing second tap contributes 0
            }
        }
    };

    // prologue: patch(0) and the first weight sub-stage
    R3D_STAMP_DECL;
    if (blockIdx.x == (gridDim.x >> 1) && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) clk_begin(kernarg_clk<Conv2Args>());
    dma_weights2(0, wbuf);
    dma_patch(0, pbuf);
    R3D_STAMP(4);
    for (int sp = 0; sp < nst; sp += 2) {
#pragma unroll
        for (int uu = 0; uu < 9; ++uu) {
            const int T0 = 9 * sp + 2 * uu;                           // first linear tap of this sub-stage
            if (T0 >= 9 * nst) break;
            // a patch DMA was issued AFTER the weights this sub-stage needs (at the top of uu = 0 / 5): leave it in flight.
            // (Counted waits need a kernel without scratch traffic -- it shares vmcnt.  Until the end of round 3 the MX instantiation
            // spilled: the tap offsets came from the ConvPhase argument, so hipcc kept 18 per-tap LDS addresses in VGPRs; as compile-time
            // constants they are ds_read immediates: 128 + 23 spilled -> 119 VGPRs, 125 -> 117 for f16x3, and R3D_MX_DRAIN=1 is the old behaviour.)
            const bool patch_behind = !(MX && R3D_MX_DRAIN) && ((uu == 1 && sp + 1 < nst) || (uu == 6 && sp + 2 < nst));
            if (patch_behind) {
                if constexpr (KMAX == 3) { if (three) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                else { if (three) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ... and every LDS read this wave has issued is complete.  Not implied by anything: hipcc may (and, for the fp8 part of the MX
            // instantiation, did) sink the MFMAs that consume the last operand reads of a sub-stage below this barrier, and its lgkmcnt wait
            // with them -- the wave then passes the barrier with reads of weight buffer `par` in flight, and the DMAs issued right after
            // the barrier (by any wave) refill that buffer.  L2-warm weights land in 250-400 cycles; next to a ray-kernel block that keeps the
            // CU's LDS queue busy a read can take longer: 55 of 1 920 pipelined frames were off by one count in a few bytes.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int par = (uu + (sp >> 1)) & 1;                     // 9 sub-stages per stage pair: the buffer parity alternates between pairs
            const uint4* curW = wbuf + par * W2_BUF;
            dma_weights2(T0 + 2, wbuf + (par ^ 1) * W2_BUF);
            if (uu == 0 && sp + 1 < nst) dma_patch(sp + 1, pbuf + PBUF);
            if (uu == 5 && sp + 2 < nst) dma_patch(sp + 2, pbuf);
#pragma unroll
            for (int ts = 0; ts < 2; ++ts) {
                const int Tl = 2 * uu + ts;                           // 0..17 within the stage pair (compile time after unrolling)
                const int sl = Tl / 9, t = Tl - 9 * sl;
                if (sp + sl < nst) {
                    const uint4* curP = pbuf + sl * PBUF;
                    h8 ah[2], al[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        uint4 q0 = curW[ts * 512 + aoff + mt * 32];
                        ah[mt] = *reinterpret_cast<h8*>(&q0);
                        if (!MX) { uint4 q1 = curW[ts * 512 + aoff + mt * 32 + 128]; al[mt] = *reinterpret_cast<h8*>(&q1); }
                    }
#if R3D_TAPS_CT & 1
                    const int toff = (t / 3 - 1) * F_PATCH_W + (t % 3 - 1);       // the plain 3x3 taps (sr_fill_conv3x3_phase): compile-time LDS offsets
#else
                    const int toff = ph.dy[t] * F_PATCH_W + ph.dx[t];
#endif
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        uint4 r0 = curP[boff[nt] + toff];
                        const h8 bh = *reinterpret_cast<h8*>(&r0);
                        h8 bl;
                        if (!MX) { uint4 r1 = curP[boff[nt] + toff + 2 * PATCH_PIX]; bl = *reinterpret_cast<h8*>(&r1); }
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            if (!MX) {
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
                            }
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
                        }
                    }
                }
            }
            if constexpr (MX) {
                // the correction products of BOTH taps of this sub-stage in one K = 64 fp8 MFMA per tile: lane half h <-> tap ts = h.
                // (a missing second tap -- last sub-stage of an odd stage count -- has all-zero weight rows, see dma_weights2.)
                // One B record and one A record live at a time (8 + 8 VGPRs): the kernel has 64 registers besides its accumulators.
                const int Tl0 = 2 * uu, Tl1 = 2 * uu + 1;
                const int sl0 = Tl0 / 9, t0 = Tl0 - 9 * sl0, sl1 = Tl1 / 9, t1 = Tl1 - 9 * sl1;
#if R3D_TAPS_CT & 2
                const int toffa = (t0 / 3 - 1) * F_PATCH_W + (t0 % 3 - 1), toffb = (t1 / 3 - 1) * F_PATCH_W + (t1 % 3 - 1);
#else
                const int toffa = ph.dy[t0] * F_PATCH_W + ph.dx[t0], toffb = ph.dy[t1] * F_PATCH_W + ph.dx[t1];
#endif
#if !R3D_MX_FREE_SCHED
                __builtin_amdgcn_sched_barrier(0);                  // fp8 operand loads stay behind this sub-stage's f16 MFMAs (register budget)
#endif
                int hh = h;
#if !R3D_MX_FREE_SCHED
                asm volatile("" : "+v"(hh));                        // keep the per-sub-stage address arithmetic inside the loop (9 hoisted VGPRs spill)
#endif
                // this lane half's tap: its stage's "lo" plane, pixel slot without the f16 chunk offset h * F_PATCH_PIX
                const int p8 = 2 * PATCH_PIX + sl0 * PBUF + toffa + hh * ((sl1 - sl0) * PBUF + toffb - toffa - PATCH_PIX);
                const uint4* P8 = pbuf + p8;
                const uint4* W8 = curW + 128 + aoff + hh * 256;                                           // [ts = h][chunk][lo row][cout]: aoff has h * 256
                i8v a8[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const uint4 q0 = W8[mt * 32], q1 = W8[mt * 32 + 256];
                    a8[mt] = (i8v){(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, (int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w};
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const uint4 r0 = P8[boff[nt]], r1 = P8[boff[nt] + PATCH_PIX];
                    const i8v b8 = (i8v){(int)r0.x, (int)r0.y, (int)r0.z, (int)r0.w, (int)r1.x, (int)r1.y, (int)r1.z, (int)r1.w};
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[mt], b8, acc[mt][nt], 0, kMxFmtB, 0, kMxScaleA, 0, kMxScaleB);
                    }
                }
#if !R3D_MX_FREE_SCHED
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
    }
    __syncthreads();
    R3D_STAMP(5);
    conv_epilogue<FULL_EPI, WN, NT>(a, ph, n, acc, i0, j0, m0, reinterpret_cast<float*>(lds));
    if (blockIdx.x == (gridDim.x >> 1) && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) clk_end(kernarg_clk<Conv2Args>());
    R3D_STAMP(6);
#ifdef R3D_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    R3D_STAMP(7);
    if ((threadIdx.x & 63) == 0) { for (int i_ = 4; i_ < 8; ++i_) atomicAdd(&r3d::g_stamps[8 + i_], st_acc_[i_]); atomicAdd(&r3d::g_stamps[30], 1ull); }
    R3D_STAMP_CLOCKS(26);
#endif
}

#include "r3d_sr_wino.h"

static constexpr int F_LDS_UINT4 = 2 * 2 * F_PATCH_PIX + 2 * 3 * 2 * 256;      // patch (20.7 KB) + 2 x 3-tap weight sub-stage (2 x 24.6 KB)

// plain 3x3 conv (9 taps)
template <int WN, int NT, int OCC, bool MX = false>
__global__ __launch_bounds__(128 * WN, OCC) void conv_mfma_f16x3_kernel(Conv2Args a)
{
    if constexpr (WN == 4 && NT == 2) {
        __shared__ uint4 lds[F3_LDS_UINT4];
        conv3x3_dma_block<true, MX>(a, a.ph[0], blockIdx.z, lds);
    } else {
        __shared__ uint4 lds[F_LDS_UINT4];
        conv2_block<9, true, WN, NT>(a, a.ph[0], blockIdx.z, lds);
    }
}

// the same conv on 8x16-pixel tiles (conv3x3_dma_block TH = 8), for launches that would fill less than half of the chip's block slots
static constexpr int F3S_LDS_UINT4 = 2 * 768 + 2 * W2_BUF;           // 57.3 KB
template <bool MX = false>
__global__ __launch_bounds__(512, 4) void conv_mfma_f16x3_rows8_kernel(Conv2Args a)
{
    __shared__ uint4 lds[F3S_LDS_UINT4];
    conv3x3_dma_block<true, MX, 8>(a, a.ph[0], blockIdx.z, lds);
}

// 1x1 conv with the full epilogue (nn.Conv2d(k=1) layers of the torso/background fusion stack)
template <int WN, int NT, int OCC>
__global__ __launch_bounds__(128 * WN, OCC) void conv1x1_mfma_f16x3_kernel(Conv2Args a)
{
    __shared__ uint4 lds[F_LDS_UINT4];
    conv2_block<1, true, WN, NT>(a, a.ph[0], blockIdx.z, lds);
}

// ---- 1x1 conv with the alpha / occlusion blend + channel concatenation of its operand fused in (round 6) ------------------------------------------------------
// `x = torch.cat([x * m, x_bg * (1 - m)], dim=1)` followed by the 1x1 conv of fuse_fg_bg_convs (modules/real3d/super_resolution/sr_with_ref.py:113-114, :56-62).
// Until round 5: r3d_blend_cat_to_split wrote the 512-channel operand in SPLIT form (134 MB at 256^2; 45 us) and conv1x1_mfma_f16x3_kernel read it back with the 18 x 18 halo
// patch of the 3x3 convs it shares its block routine with (270 MB moved for the 134 MB operand, one 16-channel stage in flight per block: 52 us).  Here one thread owns one
// (8-channel chunk, pixel) of the stage: 32 bytes of fp32 from bl_a or bl_b, times mask and in-multiplier, split into fp16 hi + lo (the arithmetic of
// blend_cat_to_split_kernel), straight into the LDS patch; FOUR stages of loads are in flight per thread (the kernel is a stream: 134 MB in, 17 MB out, 4.3 GFLOP), the
// weights ride in the same ring (global -> VGPR -> LDS), every load is visible to hipcc's vmcnt bookkeeping.  Block = 128 couts x 16 x 16 px, 8 waves of 64 couts x 64 px in
// the lane <-> pixel map of conv_epilogue (which it ends with: all output formats, bias, activation, max|y|); one block per CU (the grid of the 256^2 layer: 256 blocks).
static constexpr int BL_PATCH = 2 * 2 * 256;                          // uint4 per patch buffer: [hi|lo][chunk][16 x 16 px], odd rows rotated by 2 slots (conflict-free B reads)
static constexpr int BL_DEPTH = 4;                                    // stages of operand loads in flight
__global__ __launch_bounds__(512, 2) void conv1x1_blend_f16x3_kernel(Conv2Args a)
{
    __shared__ uint4 lds[2 * BL_PATCH + 2 * 512];                     // (the epilogue's staging, 6 * BLOCK_M floats, fits)
    constexpr int WN = 4, NT = 2;
    const ConvPhase& ph = a.ph[0];
    const int n = blockIdx.z;
    const int tiles_x = (ph.outW + F_TILE_W - 1) / F_TILE_W;
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int i0 = ty * F_TILE_H, j0 = tx * F_TILE_W;
    const int m0 = blockIdx.y * BLOCK_M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, h = lane >> 5;
    const int nst = a.Cin >> 4, hw = a.H * a.W;
    const int ca8 = a.bl_Ca >> 3, cb8 = (a.Cin - a.bl_Ca) >> 3;

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // this thread's share of a stage's operand: chunk tc of the stage, pixel (py, px) of the tile
    const int tc = threadIdx.x >> 8, py = (threadIdx.x >> 4) & 15, px = threadIdx.x & 15;
    const bool inside = i0 + py < a.H && j0 + px < a.W;
    const size_t pix = inside ? (size_t)(i0 + py) * a.W + (j0 + px) : 0;
    const float m = inside ? a.bl_mask[(size_t)n * hw + pix] : 0.f;
    const float sc = a.bl_scale[(size_t)n * a.bl_scale_stride_n];
    const float ma = inside ? m : 0.f, mb = inside ? 1.0f - m : 0.f;      // (x * m) * s in blend_cat_to_split_kernel's order: bit-identical operand, s is a power of two
    const int wslot = tc * 256 + py * 16 + ((px + 2 * (py & 1)) & 15);
    auto src_of = [&](int st) -> const float4* {                       // chunk 2 st + tc of the concatenation: bl_a's channels first, then bl_b's
        const int cc = 2 * st + tc;
        const float* base = cc < ca8 ? a.bl_a + ((size_t)n * ca8 + cc) * hw * 8 : a.bl_b + ((size_t)n * cb8 + (cc - ca8)) * hw * 8;
        return reinterpret_cast<const float4*>(base + pix * 8);
    };
    float4 raw[BL_DEPTH][2];
    auto issue = [&](int st, float4 (&r)[2]) {                         // no branch: a stage past the end re-reads the last one (never staged), a pixel outside reads pixel 0 (times 0)
        const float4* s4 = src_of(st < nst ? st : nst - 1); r[0] = s4[0]; r[1] = s4[1];
    };
    auto stage_patch = [&](int st, const float4 (&r)[2], uint4* P) {   // raw -> fp16 hi + lo -> LDS patch
        const float f = (2 * st + tc) < ca8 ? ma : mb;
        const float x[8] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w};
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = as_rounded(x[j] * f * sc);
            const float cl = fminf(fmaxf(t, -65504.f), 65504.f);
            hi[j] = (_Float16)cl; lo[j] = (_Float16)(t - (float)hi[j]);
        }
        P[wslot] = *reinterpret_cast<uint4*>(&hi);
        P[512 + wslot] = *reinterpret_cast<uint4*>(&lo);
    };
    // weights of a stage: [chunk 2 st, 2 st + 1][hi|lo][Cout] -> this block's 128 couts of the four rows, one uint4 per thread, in the SAME ring as the operand (the
    // vmcnt counter retires in order: a weight load issued later than an operand load and needed earlier would drain the ring -- the first version of this kernel
    // loaded A operands global -> VGPR one stage ahead and ran at 3.3 TB/s, an effective depth of one stage)
    const uint4* WP = a.wp + (size_t)(threadIdx.x >> 7) * a.Cout + m0 + (threadIdx.x & 127);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 wraw[BL_DEPTH];
    auto issue_w = [&](int st, u32x4& r) { r = *reinterpret_cast<const u32x4*>(WP + (size_t)(st < nst ? st : nst - 1) * 4 * a.Cout); };
    uint4* const Abuf = lds + 2 * BL_PATCH;                           // [parity][row = chunk * 2 + (hi|lo)][128 couts]
    const int aoff = h * 256 + 64 * wm + li;                           // + hl * 128 + 32 * mt
    // B slot of this lane in N tile nt: rows (2 nt, 2 nt + 1) of the wave's four rows, columns rotated by 2 on odd rows like the patch
    const int prow = li >> 4, row0 = wn * 2 * NT;
    int boff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) boff[nt] = h * 256 + (row0 + nt * 2 + prow) * 16 + (li & 15);

#pragma unroll
    for (int d = 0; d < BL_DEPTH; ++d) { issue(d, raw[d]); issue_w(d, wraw[d]); }
    stage_patch(0, raw[0], lds);
    *reinterpret_cast<u32x4*>(Abuf + threadIdx.x) = wraw[0];
    issue(BL_DEPTH, raw[0]); issue_w(BL_DEPTH, wraw[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // stage s: MFMAs on patch[s & 1] | operand of stage s + 1 -> patch[(s + 1) & 1] | loads of stage s + 1 + BL_DEPTH; the ring slot of a stage is (stage % BL_DEPTH): the
    // loop is unrolled over BL_DEPTH stages so that every slot is a fixed set of registers (a rotating copy would wait for loads still in flight)
    for (int s0 = 0; s0 < nst; s0 += BL_DEPTH) {
#pragma unroll
        for (int d = 0; d < BL_DEPTH; ++d) {
            const int s = s0 + d;                                     // (nst % BL_DEPTH == 0: the launcher's precondition Cin % 64 == 0)
            {
                const uint4* P = lds + (d & 1) * BL_PATCH;            // (BL_DEPTH is even: the patch parity of stage s is d & 1)
                const uint4* A = Abuf + (d & 1) * 512 + aoff;
                h8 ah[2], al[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) { const uint4 q0 = A[32 * mt], q1 = A[128 + 32 * mt]; ah[mt] = *reinterpret_cast<const h8*>(&q0); al[mt] = *reinterpret_cast<const h8*>(&q1); }
                h8 bh[NT], bl[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { const uint4 r0 = P[boff[nt]], r1 = P[512 + boff[nt]]; bh[nt] = *reinterpret_cast<const h8*>(&r0); bl[nt] = *reinterpret_cast<const h8*>(&r1); }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    }
                stage_patch(s + 1, raw[(d + 1) % BL_DEPTH], lds + ((d + 1) & 1) * BL_PATCH);   // (after the last stage: into the buffer nobody reads any more)
                *reinterpret_cast<u32x4*>(Abuf + ((d + 1) & 1) * 512 + threadIdx.x) = wraw[(d + 1) % BL_DEPTH];
                issue(s + 1 + BL_DEPTH, raw[(d + 1) % BL_DEPTH]); issue_w(s + 1 + BL_DEPTH, wraw[(d + 1) % BL_DEPTH]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
    }
    conv_epilogue<true, WN, NT>(a, ph, n, acc, i0, j0, m0, reinterpret_cast<float*>(lds));
}

// stride-2 transposed conv phases (4/2/2/1 taps)
template <int WN, int NT, int OCC>
__global__ __launch_bounds__(128 * WN, OCC) void tconv_mfma_f16x3_kernel(Conv2Args a)
{
    __shared__ uint4 lds[F_LDS_UINT4];
    const int n = blockIdx.z / a.nphase, p = blockIdx.z - n * a.nphase;
    const ConvPhase& ph = a.ph[p];
    const int tiles = ((ph.outW + F_TILE_W - 1) / F_TILE_W) * ((ph.outH + F_TILE_H - 1) / F_TILE_H);
    if ((int)blockIdx.x >= tiles) return;
    switch (ph.ntaps) {
        case 4: conv2_block<4, false, WN, NT>(a, ph, n, lds); break;
        case 2: conv2_block<2, false, WN, NT>(a, ph, n, lds); break;
        default: conv2_block<1, false, WN, NT>(a, ph, n, lds); break;
    }
}

// ---- fused up-sampling conv: conv_transpose2d(stride 2) + FIR 4x4 (gain 4, pad 1) + bias + lrelu*sqrt2 -> SPLIT --------
// (conv2d_resample.py:116-133 up=2 path + bias_act of SynthesisLayer.forward, networks_stylegan2.py:329-341.)
// The transposed conv is evaluated as its four output phases T_p(i,j) = sum_{taps (ky,kx) = p mod 2} x(i - ky/2, j - kx/2) W[ky][kx]
// (row 2i+pa, col 2j+pb of the (2H+1)x(2W+1) result).  One block owns ALL FOUR phases of a 16x16 grid of (i,j) for 32
// couts, so the input patch is read once for the 9 taps (the per-phase kernel re-read it four times: that launch was
// memory-bound at 28 % of the MFMA pipe) and the fp32 T never leaves the CU: it is FIR-filtered out of LDS into the
// 28x28 outputs the grid fully determines (halo recompute (16/14)^2 = 1.31x MFMA work), biased, activated, scaled by
// the next conv's styles, split and stored.  Out-of-image inputs are zero-filled, which makes the rows r = -1 and
// r = 2H+1 that the FIR's zero padding touches come out as exact zeros.
//   4 waves x (32 couts x 64 grid points x 4 phases) = 8 accumulator tiles (128 VGPRs) at 2 waves/SIMD, 2 blocks/CU.
//   Per 16-channel stage a wave reads the 4 shifted B windows once (16 ds_read_b128) and 9 taps of A (18) for 54 MFMAs.
static constexpr int U_TILE = 14;                                   // inputs (= output quads) per tile side
static constexpr int U_WSTAGE = 9 * 2 * 2 * 32;                     // uint4 per weight stage: [tap][chunk][hi|lo][32 couts]
static constexpr int U_PW = 17, U_PPLANE = 320;                     // patch: 17x17 pixels per (plane, chunk), padded to 5 x 64 slots
static constexpr int U_PATCH = 4 * U_PPLANE;                        // uint4: [hi|lo][chunk][320]
static constexpr int U_STAGE = U_PATCH + U_WSTAGE;                  // 2432 uint4 per stage buffer
static constexpr int U_LDS_UINT4 = 2 * U_STAGE + 2;                 // double-buffered: 77.8 KB (epilogue slice: 2048 uint4) + a 32-byte zero record (MXIN)
static constexpr int U_ZERO = 2 * U_STAGE;                          // the zero A record of the unpaired centre tap's second K half


struct UpArgs {
    const uint4* x; size_t x_stride_n;          // SPLIT input [hi|lo][Cin/8][H][W]
    const uint4* wp;                            // [Cout/32][Cin/16][tap][chunk][hi|lo][32] (sr_prepack_up_kernel)
    const float* out_scale; const float* bias; const float* next_scale; size_t vec_stride_n;
    uint4* y; size_t y_stride_n;                // SPLIT output [hi|lo][Cout/8][2H][2W], scaled by next_scale
    int Cin, Cout, H, W, tiles_x, ntiles, tiles_per_xcd;
    float clamp;
    unsigned long long* clk;                    // prof_clock_slot(R3D_PROF_UPCONV) or null
};

__global__ void sr_prepack_up_kernel(const float* __restrict__ w, int Cin, int Cout, const float* __restrict__ winv, uint4* __restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)9 * (Cin / 8) * Cout * 2;
    if (e >= total) return;
    const int co = e & 31, hl = (e >> 5) & 1, hc = (e >> 6) & 1;
    const int t = (int)((e >> 7) % 9);
    const size_t r = (e >> 7) / 9;
    const int nst = Cin / 16;
    const int st = (int)(r % nst), cg = (int)(r / nst);
    const int cout = cg * 32 + co;
    h8 v8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = st * 16 + hc * 8 + j;
        _Float16 a, b; split1(w[((size_t)cout * Cin + ci) * 9 + t] * (1.0f / winv[cout]), a, b);
        v8[j] = hl ? b : a;
    }
    out[e] = *reinterpret_cast<uint4*>(&v8);
}

// The same layout for an input that arrives as R3D_FMT_SPLIT_MX (upconv_fir_f16x3_kernel<.., .., MXIN = true>): the hi rows are unchanged, the
// "lo" row of chunk 0 holds wl8 = fp8(lo * 2^8) of the stage's 16 channels, the "lo" row of chunk 1 their wh8 = fp8(hi * 2^-3) -- the A
// record [wl8 | wh8] of a (tap, cout) is rows 1 and 3 of that tap's four 32-cout rows.
__global__ void sr_prepack_up_mx_kernel(const float* __restrict__ w, int Cin, int Cout, const float* __restrict__ winv, uint4* __restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)9 * (Cin / 8) * Cout * 2;
    if (e >= total) return;
    const int co = e & 31, hl = (e >> 5) & 1, hc = (e >> 6) & 1;
    const int t = (int)((e >> 7) % 9);
    const size_t r = (e >> 7) / 9;
    const int nst = Cin / 16;
    const int st = (int)(r % nst), cg = (int)(r / nst);
    const int cout = cg * 32 + co;
    const float ws = 1.0f / winv[cout];
    if (!hl) {
        h8 v8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { _Float16 a, b; split1(w[((size_t)cout * Cin + st * 16 + hc * 8 + j) * 9 + t] * ws, a, b); v8[j] = a; }
        out[e] = *reinterpret_cast<uint4*>(&v8);
        return;
    }
    uint4 rec;
    unsigned* pr = reinterpret_cast<unsigned*>(&rec);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = w[((size_t)cout * Cin + st * 16 + mx_rec_chan(q) + j) * 9 + t] * ws;
            _Float16 a, b; split1(v, a, b);
            f[j] = hc ? (float)a * kMxWh : (v - (float)a) * kMxWl;
        }
        pr[q] = pack4_fp8(f[0], f[1], f[2], f[3]);
    }
    out[e] = rec;
}

// MX = true (R3D_SR_F16MX): the output feeds the f16mx 3x3 conv: hi plane as usual, and in place of the fp16 lo words the fp8 records
// (lo chunk 2G <- xh8 of channels 16G..16G+15, lo chunk 2G+1 <- xl8): slice g (8 couts) owns bytes [8 (g & 1), +8) of both words.
// MXIN = true: the input is R3D_FMT_SPLIT_MX (hi plane + fp8 records, written by the MX epilogue of the previous block's conv1) and the
// main loop spends, per 16-channel stage and N tile, 9 f16 MFMAs (hi * hi) + 5 fp8 K = 64 MFMAs -- the cross products of the tap pairs
// (0,2) (6,8) -> phase 0, (1,7) -> phase 1, (3,5) -> phase 2 and of the centre tap 4 (phase 3, its second K half reads a zero record) --
// instead of 27 f16 MFMAs: 608 instead of 864 matrix cycles.  Lane half h <-> the pair's tap h, as in conv3x3_dma_block<.., MX>.
// NW = waves per block: 4 (two N tiles = 128 accumulator registers per wave, 2 waves per SIMD: the shape of the MFMA-bound layers) or
// 8 (one N tile, 64 accumulators, ~110 VGPRs: 4 waves per SIMD at the same 2 blocks per CU) for layers with a handful of K stages --
// block0.conv0 (Cin = 32: two stages) is ALL epilogue, and the epilogue's dependent LDS -> VALU -> store chains ran at 30 % of the VALU
// rate with two waves per SIMD (round 4: 39.8 -> see DESIGN 4.2).  Same tile, same K order per output: bit-identical results.
template <bool CLAMP, bool MX = false, bool MXIN = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void upconv_fir_f16x3_kernel(UpArgs a)
{
    constexpr int NTL = 8 / NW;                                     // N tiles (2 grid rows x 16 columns) per wave
    constexpr int NTHR = 64 * NW;
    R3D_STAMP_DECL;
    __shared__ uint4 lds[U_LDS_UINT4];
    const int G = a.Cout >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;         // the cout groups of one tile share an XCD (one L2)
    const int tl = slot / G, cg = slot - tl * G;
    const int tile = xcd * a.tiles_per_xcd + tl;
    if (tile >= a.ntiles) return;
    const int n = blockIdx.y;
    if (blockIdx.x == (gridDim.x >> 1) && n == 0 && threadIdx.x == 0) clk_begin(kernarg_clk<UpArgs>());
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int i0 = ty * U_TILE, j0 = tx * U_TILE;                   // first input row / col of the tile; grid point g <-> i0 - 1 + g
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int nst = a.Cin >> 4;
    const int chunk_stride = a.H * a.W;
    const size_t plane = (size_t)(a.Cin >> 3) * chunk_stride;
    const uint4* X = a.x + (size_t)n * a.x_stride_n;

    f32x16 acc[4][NTL];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][nt][r] = 0.f;

    // grid point of this lane in N tile nt: row 4*wave + 2*nt + prow, col pcol (odd rows rotated by the row stride mod 16:
    // every ds_read_b128 lane group then covers 16 distinct 16-byte slots)
    const int prow = li >> 4, pcol = ((li & 15) - prow) & 15;
    const int row0 = wave * 2 * NTL;
    int boff[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) boff[nt] = (row0 + nt * 2 + prow + 1) * U_PW + (pcol + 1) + h * U_PPLANE;
    const int aoff = U_PATCH + h * 64 + li;
    // MXIN: per-lane slots of the fp8 operands.  B record of (pixel, tap_h) = the lo-plane words of chunk 0 (xh8) and chunk 1 (xl8) at the
    // tap's window shift: pairs (0,2) (3,5) differ by one column, (1,7) by one row, (6,8) = one row up and one column.  A record of
    // (cout, tap_h) = rows 1 and 3 of the tap: taps of a pair are 2 or 6 taps apart.
    int b8c[NTL], b8r[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        const int pix = (row0 + nt * 2 + prow + 1) * U_PW + (pcol + 1) + 2 * U_PPLANE;
        b8c[nt] = pix - h; b8r[nt] = pix - h * U_PW;
    }
    const int a8d2 = U_PATCH + 32 + li + h * 256, a8d6 = U_PATCH + 32 + li + h * 768;
    const int a8c = U_PATCH + 32 + li + 4 * 128;                   // centre tap: lanes h = 1 read the zero record instead
    if (MXIN && tid < 2) lds[U_ZERO + tid] = make_uint4(0, 0, 0, 0);

    // patch DMA: wave w fills segment sg = 4k + w (k = 0..4) of the 20 x 64-slot patch image; slot idx = (sg % 5) * 64 + lane of
    // (plane, chunk) = sg / 5 holds patch pixel (idx / 17, idx % 17) <-> input (i0 - 2 + py, j0 - 2 + px); slots outside the
    // image / beyond 289 read the zero block
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr int PK = (20 + NW - 1) / NW;                          // patch DMA rounds: 20 segments of 64 slots over NW waves
    constexpr int WKF = 18 / NW, WKR = 18 - WKF * NW;               // weight DMA: 18 segments = WKF full rounds + WKR waves of a last one
    unsigned pf_off[PK];
    unsigned pf_valid = 0;
#pragma unroll
    for (int k = 0; k < PK; ++k) {
        const int sg = NW * k + wave_u, pc = sg / 5, idx = (sg - pc * 5) * 64 + lane;
        const int py = idx / U_PW, px = idx - py * U_PW;
        const int iy = i0 - 2 + py, ix = j0 - 2 + px;
        unsigned off = 0;
        if (sg < 20 && idx < U_PW * U_PW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
            off = (unsigned)((pc >> 1) * plane) + (unsigned)((pc & 1) * chunk_stride + iy * a.W + ix);
            pf_valid |= 1u << k;
        }
        pf_off[k] = off;
    }
    const uint4* WP = a.wp + (size_t)cg * nst * U_WSTAGE;
    auto dma_stage = [&](int st, uint4* buf) {
        const uint4* src = WP + (size_t)st * U_WSTAGE + tid;        // weights: 1152 uint4, linear: 18 segments of 64 lanes
#pragma unroll
        for (int k = 0; k <= WKF; ++k)
            if (k < WKF || wave_u < WKR) dma64(src + NTHR * k, buf + U_PATCH + NTHR * k + 64 * wave_u);
        const uint4* Xs = X + (size_t)(2 * st) * chunk_stride;
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            if (NW * k + wave_u < 20) {
                const uint4* g = (pf_valid & (1u << k)) ? Xs + pf_off[k] : g_zero16;
                dma64(g, buf + 64 * (NW * k + wave_u));
            }
        }
    };
    dma_stage(0, lds);
    R3D_STAMP(0);
    i8v pa68, pa17, pb68[NTL], pb17[NTL];                           // MXIN: operand set P4 of the previous stage (see the main loop)
    for (int st = 0; st < nst; ++st) {
#ifdef R3D_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); R3D_STAMP(1);      // stamps build: split the stage's cycles into compute (1) | own DMA wait (4) | barrier (5)
#endif
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // this stage's DMAs (issued one stage ago) have landed; this wave's LDS reads too (see conv3x3_dma_block)
        R3D_STAMP(4);
        __builtin_amdgcn_s_barrier();                               // ... everybody's; and the other buffer's readers are done
        R3D_STAMP(5);
        asm volatile("" ::: "memory");
        const uint4* cur = lds + (st & 1) * U_STAGE;
        if (st + 1 < nst) dma_stage(st + 1, lds + ((st + 1) & 1) * U_STAGE);
        if constexpr (MXIN) {
            // hi * hi on the f16 pipe, window by window (a B window is read once for its 4 / 2 / 2 / 1 taps); then the cross products: one K = 64 fp8
            // MFMA per tap pair and N tile -- taps (0, 2), (3, 5) and the centre tap meet the SAME B records (windows (0,0) / (0,1) of the lane halves).
            // Two waves per SIMD do not hide an LDS round trip per MFMA (hipcc's own schedule was "2 reads, wait, 1 MFMA" a dozen times per stage:
            // 3.1 k cycles for 1.2 k of MFMA), so the stage is four operand sets, each read while the previous set's MFMAs run:
            //   R(P1) | M(P4 of the PREVIOUS stage: operands in registers since before the barrier) | R(P2) | M(P1) | R(P3) | M(P2) | R(P4) | M(P3)
            //   P1 = windows 0, 1 (taps 0 1 3 4 | 2 5), P2 = windows 2, 3 (taps 6 7 | 8), P3 = pairs (0,2) (3,5) (4,-), P4 = pairs (6,8) (1,7).
            // Per accumulator the MFMA order is the unpipelined one (f16 taps in window order, then its fp8 pairs): bit-identical results.
            auto ld8 = [&](const uint4* p0, const uint4* p1) {
                const uint4 q0 = *p0, q1 = *p1;
                return (i8v){(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, (int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w};
            };
            auto ldh = [&](const uint4* p0) { uint4 q = *p0; return *reinterpret_cast<h8*>(&q); };
            const uint4* curA = cur + aoff;
            const int cb = (st & 1) * U_STAGE;
            // ---- R(P1)
            h8 B1[2][NTL], A1[6];
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) { B1[0][nt] = ldh(cur + boff[nt]); B1[1][nt] = ldh(cur + boff[nt] - 1); }
            {
                constexpr int T1[6] = {0, 1, 3, 4, 2, 5};
#pragma unroll
                for (int k = 0; k < 6; ++k) A1[k] = ldh(curA + T1[k] * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- M(P4) of the previous stage
            if (st > 0) {
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa68, pb68[nt], acc[0][nt], 0, kMxFmtB, 0, kMxScaleA, 0, kMxScaleB);
                    acc[1][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa17, pb17[nt], acc[1][nt], 0, kMxFmtB, 0, kMxScaleA, 0, kMxScaleB);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- R(P2)
            h8 B2[2][NTL], A2[3];
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) { B2[0][nt] = ldh(cur + boff[nt] - U_PW); B2[1][nt] = ldh(cur + boff[nt] - U_PW - 1); }
#pragma unroll
            for (int k = 0; k < 3; ++k) A2[k] = ldh(curA + (6 + k) * 128);
            __builtin_amdgcn_sched_barrier(0);
            // ---- M(P1): window 0 -> taps 0 1 3 4 = phases 0 1 2 3; window 1 -> taps 2 5 = phases 0 2
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) acc[k][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[k], B1[0][nt], acc[k][nt], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) acc[2 * k][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[4 + k], B1[1][nt], acc[2 * k][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- R(P3)
            const i8v a02 = ld8(lds + cb + a8d2 + 0 * 128, lds + cb + a8d2 + 0 * 128 + 64);
            const i8v a35 = ld8(lds + cb + a8d2 + 3 * 128, lds + cb + a8d2 + 3 * 128 + 64);
            const i8v a4 = ld8(lds + (h ? U_ZERO : cb + a8c), lds + (h ? U_ZERO + 1 : cb + a8c + 64));   // tap 4 alone: zero A record for the K half h = 1 (whatever B it meets)
            i8v b3[NTL];
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) b3[nt] = ld8(cur + b8c[nt], cur + b8c[nt] + U_PPLANE);
            __builtin_amdgcn_sched_barrier(0);
            // ---- M(P2): window 2 -> taps 6 7 = phases 0 1; window 3 -> tap 8 = phase 0
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) acc[k][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2[k], B2[0][nt], acc[k][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2[2], B2[1][nt], acc[0][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- R(P4): taps (6, 8) = windows (1,0) / (1,1); taps (1, 7) = windows (0,0) / (1,0)
            pa68 = ld8(lds + cb + a8d2 + 6 * 128, lds + cb + a8d2 + 6 * 128 + 64);
            pa17 = ld8(lds + cb + a8d6 + 1 * 128, lds + cb + a8d6 + 1 * 128 + 64);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                pb68[nt] = ld8(cur + b8c[nt] - U_PW, cur + b8c[nt] - U_PW + U_PPLANE);
                pb17[nt] = ld8(cur + b8r[nt], cur + b8r[nt] + U_PPLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- M(P3)
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                acc[0][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a02, b3[nt], acc[0][nt], 0, kMxFmtB, 0, kMxScaleA, 0, kMxScaleB);
                acc[2][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a35, b3[nt], acc[2][nt], 0, kMxFmtB, 0, kMxScaleA, 0, kMxScaleB);
                acc[3][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b3[nt], acc[3][nt], 0, kMxFmtB, 0, kMxScaleA, 0, kMxScaleB);
            }
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
#pragma unroll
        for (int win = 0; win < 4; ++win) {                         // input shift (sy, sx): x(i - sy, j - sx)
            const int sy = win >> 1, sx = win & 1;
            const int toff = -(sy * U_PW + sx);
            h8 bh[NTL], bl[NTL];
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                uint4 r0 = cur[boff[nt] + toff];
                uint4 r1 = cur[boff[nt] + toff + 2 * U_PPLANE];
                bh[nt] = *reinterpret_cast<h8*>(&r0); bl[nt] = *reinterpret_cast<h8*>(&r1);
            }
#pragma unroll
            for (int ky = 2 * sy; ky <= (sy ? 2 : 1); ++ky)
#pragma unroll
                for (int kx = 2 * sx; kx <= (sx ? 2 : 1); ++kx) {
                    const int t = ky * 3 + kx, p = (ky & 1) * 2 + (kx & 1);
                    uint4 q0 = cur[t * 128 + aoff], q1 = cur[t * 128 + aoff + 32];
                    const h8 ah = *reinterpret_cast<h8*>(&q0), al = *reinterpret_cast<h8*>(&q1);
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt) {
                        acc[p][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[nt], acc[p][nt], 0, 0, 0);
                        acc[p][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[nt], acc[p][nt], 0, 0, 0);
                        acc[p][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nt], acc[p][nt], 0, 0, 0);
                    }
                }
        }
    }
    if constexpr (MXIN) {                                           // M(P4) of the last stage
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            acc[0][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa68, pb68[nt], acc[0][nt], 0, kMxFmtB, 0, kMxScaleA, 0, kMxScaleB);
            acc[1][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa17, pb17[nt], acc[1][nt], 0, kMxFmtB, 0, kMxScaleA, 0, kMxScaleB);
        }
    }

    // ---- epilogue.  T (demodulated, fp32, in the accumulators) -> horizontal 4-tap pass IN REGISTERS -> H through LDS in 4 slices of
    // 8 couts ([pa][half][16 rows][28 cols], double-buffered: one barrier per slice) -> vertical 4-tap pass + bias + lrelu*sqrt2
    // (+clamp) -> * next styles -> fp16 hi/lo -> y.  Separable (8 instead of 16 FMAs per output), arithmetic on float pairs.
    // The horizontal pass needs T at grid columns X, X+1, X+2 of the SAME grid row, and a grid row is one 16-lane DPP row of the
    // accumulator layout (lane = pixel), so the neighbours arrive by row rotations (v_mov_dpp row_ror:15 / :14; the rotation also covers
    // the odd rows, whose columns are stored rotated by one lane).  Round 1 wrote T to LDS and ran the pass LDS -> LDS: 56 b128 LDS
    // operations per thread and slice, now 28 -- the ablations of round 2 had the epilogue at 56 % of this kernel.
    R3D_STAMP(1);
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
    // (Round 3 tried an XOR swizzle of the H columns against the write pattern's even-slot stride and a software pipeline of the four slices
    // -- hpass(g + 1) in front of vpass(g): both bit-identical and both time-neutral, SQ_LDS_BANK_CONFLICT unchanged (profiles/r03): the
    // epilogue is VALU-issue bound, ~460 wave64 instructions per slice at 2 waves per SIMD.  Not kept.)
    float4* hls0 = reinterpret_cast<float4*>(lds);                  // H: 2 x [pa][half][gy][32] (28 used), column swizzled by 8*half
    const int co0 = cg * 32;
    const float* D = a.out_scale + (size_t)n * a.vec_stride_n + co0;
    const float* Bv = a.bias + (size_t)n * a.vec_stride_n + co0;
    const float* NS = a.next_scale + (size_t)n * a.vec_stride_n + co0;
    const int OH = 2 * a.H, OW = 2 * a.W;
    const size_t oplane = (size_t)(a.Cout >> 3) * OH * OW;
    // all per-cout vectors of the four slices in one batch of loads, waited for ONCE here: loaded inside the slice loop,
    // hipcc re-waited with s_waitcnt vmcnt(0) in front of every conditional store group (loads and stores share vmcnt), which
    // serialised the epilogue on store latency
    // (round 4: the FIR runs UNNORMALISED -- taps (1, 3, 3, 1) per axis instead of (1, 3, 3, 1) / 4 -- on the raw accumulators, and the
    // demodulation d[co] / 16 is one factor of the fma that adds the bias in the vertical pass: two multiplies per T value and one per H
    // value less, and the horizontal pass is 3 instead of 4 operations per output.  The epilogue is VALU-issue bound)
    float4 dv4[4], bv4[4], nv4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        dv4[g] = *reinterpret_cast<const float4*>(D + 8 * g + 4 * (tid & 1));
        bv4[g] = *reinterpret_cast<const float4*>(Bv + 8 * g + 4 * (tid & 1));
        nv4[g] = *reinterpret_cast<const float4*>(NS + 8 * g + 4 * (tid & 1));
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
        asm volatile("" :: "v"(dv4[g].x), "v"(dv4[g].w), "v"(bv4[g].x), "v"(bv4[g].w), "v"(nv4[g].x), "v"(nv4[g].w));
    __syncthreads();                                                // the main loop's LDS reads are done
    unsigned mxk_h[1024 / NTHR][2], mxk_l[1024 / NTHR][2];
#pragma unroll
    for (int k = 0; k < 1024 / NTHR; ++k) { mxk_h[k][0] = mxk_h[k][1] = 0u; mxk_l[k][0] = mxk_l[k][1] = 0u; }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4* hls = hls0 + (g & 1) * 2048;
        // horizontal pass of this slice: (T[pa][pb=0], T[pa][pb=1]) at this lane's grid point -> (H column 2X, H column 2X+1)
        //   T column 2X+1+cc = phase (cc+1)&1 at grid column X + (cc+1)/2;  4 H(dx) = (T[dx] + T[dx+3]) + 3 (T[dx+1] + T[dx+2])
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                float hx[2][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t0 = acc[2 * pa][nt][4 * g + r], t1 = acc[2 * pa + 1][nt][4 * g + r];
                    // lane i <- lane (i + 1) % 16 / (i + 2) % 16 of its 16-lane row: row_ror:15 = 0x12F, row_ror:14 = 0x12E (a rotation has no
                    // invalid source lane: mov_dpp with bound_ctrl, which hipcc folds into the consuming v_add / v_fmac as a DPP operand)
                    const float ts = t0 + t1;                                        // (t0 + t1) one column on = t0p1 + t1p1: one rotation for both
                    const float tsp1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ts), 0x12F, 0xF, 0xF, true));
                    const float t0p1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t0), 0x12F, 0xF, 0xF, true));
                    const float t1p1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t1), 0x12F, 0xF, 0xF, true));
                    const float t0p2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t0), 0x12E, 0xF, 0xF, true));
                    const float t1p2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t1), 0x12E, 0xF, 0xF, true));
                    hx[0][r] = __builtin_fmaf(tsp1, 3.0f, t1 + t0p2);
                    hx[1][r] = __builtin_fmaf(t1p1 + t0p2, 3.0f, t0p1 + t1p2);
                }
                const int gy = row0 + nt * 2 + prow;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
                    hls[((pa * 2 + h) * 16 + gy) * 32 + ((2 * pcol + dx + 8 * h) & 31)] = make_float4(hx[dx][0], hx[dx][1], hx[dx][2], hx[dx][3]);
            }
        __syncthreads();                                            // (buffer g & 1 was last read by slice g - 2's vertical pass: done before barrier g - 1)
        {   // vertical: item (row pair rp, column oc, half): H rows 2rp+1 .. 2rp+5 -> y rows 2rp, 2rp+1 (4 couts each)
            const float4 b4 = bv4[g], n4 = nv4[g], d4 = dv4[g];
            const f2 ba = f2{b4.x, b4.y}, bb = f2{b4.z, b4.w};
            const f2 da = f2{d4.x, d4.y} * 0.0625f, db = f2{d4.z, d4.w} * 0.0625f;      // demodulation / 16 (the two unnormalised FIR passes)
            const float m = 1.4142135623730951f;
            const f2 ma = f2{n4.x, n4.y} * m, mb = f2{n4.z, n4.w} * m;
            uint4* d = a.y + (size_t)n * a.y_stride_n + (size_t)((co0 >> 3) + g) * OH * OW;
#pragma unroll
            for (int k = 0; k < 1024 / NTHR; ++k) {
                const int q = tid + NTHR * k;
                const int half = q & 1, oc = (q >> 1) & 31, rp = q >> 6;
                const bool live = rp < U_TILE && oc < 2 * U_TILE;
                f2 ha[5], hb[5];
#pragma unroll
                for (int rr = 0; rr < 5; ++rr) {                    // H row 2rp+1+rr = phase (rr+1)&1 at grid row rp + (rr+1)/2
                    const int gyy = min(rp, U_TILE - 1) + ((rr + 1) >> 1);   // (clamped: dead items read valid LDS)
                    const float4 t4 = hls[((((rr + 1) & 1) * 2 + half) * 16 + gyy) * 32 + ((oc + 8 * half) & 31)];
                    ha[rr] = f2{t4.x, t4.y}; hb[rr] = f2{t4.z, t4.w};
                }
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    f2 va = ((ha[dy] + ha[dy + 3]) + (ha[dy + 1] + ha[dy + 2]) * 3.0f) * da + ba;
                    f2 vb = ((hb[dy] + hb[dy + 3]) + (hb[dy + 1] + hb[dy + 2]) * 3.0f) * db + bb;
                    va = __builtin_elementwise_max(va, va * 0.2f);  // leaky relu (slope 0.2); sqrt(2) gain is inside ma/mb
                    vb = __builtin_elementwise_max(vb, vb * 0.2f);
                    if constexpr (CLAMP) {                          // conv_clamp acts on the gained activation
                        const f2 lim = f2{a.clamp, a.clamp} * 0.7071067811865476f;
                        va = __builtin_elementwise_min(__builtin_elementwise_max(va, -lim), lim);
                        vb = __builtin_elementwise_min(__builtin_elementwise_max(vb, -lim), lim);
                    }
                    va = va * ma;                                   // |.| < 2^15 by the range fold (r3d_chain_fold): no saturation guard needed
                    vb = vb * mb;
                    asm("" : "+v"(va), "+v"(vb));                   // as_rounded(): hi and lo below come from these fp32 values, not from their factors
                    const hh2 hia = __builtin_convertvector(va, hh2), hib = __builtin_convertvector(vb, hh2);
                    const hh2 loa = __builtin_convertvector(va - __builtin_convertvector(hia, f2), hh2);
                    const hh2 lob = __builtin_convertvector(vb - __builtin_convertvector(hib, f2), hh2);
                    // lanes 2j (couts 0-3) and 2j+1 (couts 4-7) hold the same pixel: each stores its 8-byte half of the 16-byte
                    // hi word and of the lo word (a wave store covers 32 pixels x 16 contiguous bytes per plane)
                    const uint2 hw = make_uint2(*reinterpret_cast<const unsigned*>(&hia), *reinterpret_cast<const unsigned*>(&hib));
                    const int oy = 2 * (i0 + rp) + dy, ox = 2 * j0 + oc;
                    if constexpr (MX) {
                        const f2 fha = __builtin_convertvector(hia, f2), fhb = __builtin_convertvector(hib, f2);
                        const f2 fla = va - fha, flb = vb - fhb;
                        const unsigned xh8 = pack4_x8(fha.x * kMxXh, fha.y * kMxXh, fhb.x * kMxXh, fhb.y * kMxXh);
                        const unsigned xl8 = pack4_x8(fla.x * kMxXl, fla.y * kMxXl, flb.x * kMxXl, flb.y * kMxXl);
                        if (!(g & 1)) { mxk_h[k][dy] = xh8; mxk_l[k][dy] = xl8; }      // the group's even chunk: its record dwords wait for slice g + 1
                        if (live && oy < OH && ox < OW) {
                            const size_t pix = (size_t)oy * OW + ox;
                            reinterpret_cast<uint2*>(d + pix)[half] = hw;
                            if (g & 1) {
                                // fp8 records of the 16-channel group (cg * 2 + (g >> 1)): even lo chunk = xh8, odd lo chunk = xl8; this lane's dwords
                                // 2 half, 2 half + 1 (mx_rec_chan): the lane pair writes the whole 16-byte record
                                uint4* rec = a.y + (size_t)n * a.y_stride_n + oplane + (size_t)((co0 >> 3) + (g & ~1)) * OH * OW + pix;    // lo plane
                                reinterpret_cast<uint2*>(rec)[half] = make_uint2(mxk_h[k][dy], xh8);
                                reinterpret_cast<uint2*>(rec + (size_t)OH * OW)[half] = make_uint2(mxk_l[k][dy], xl8);
                            }
                        }
                    } else {
                        const uint2 lw = make_uint2(*reinterpret_cast<const unsigned*>(&loa), *reinterpret_cast<const unsigned*>(&lob));
                        if (live && oy < OH && ox < OW) {
                            uint2* d2 = reinterpret_cast<uint2*>(d + (size_t)oy * OW + ox) + half;
                            d2[0] = hw;
                            d2[2 * oplane] = lw;
                        }
                    }
                }
            }
        }
    }
    R3D_STAMP(2);
    if (blockIdx.x == (gridDim.x >> 1) && blockIdx.y == 0 && threadIdx.x == 0) clk_end(kernarg_clk<UpArgs>());
#ifdef R3D_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    R3D_STAMP(3);
    R3D_STAMP_FLUSH(6, 1);
#endif
}
R3D_STAMP_READER(r3d_debug_stamps_sr)

// ---- FIR 4x4 (gain 4, pad 1) + bias + lrelu*sqrt2 on the transposed-conv output; writes SPLIT scaled by the next
// conv's styles.  T is PHASE-MAJOR: T[p=(r&1)*2+(c&1)][C/8][Hin+1][Win+1][8] holds row r, col c of the (2Hin+1)x(2Win+1)
// transposed-conv result (the conv epilogue's stores are then contiguous).  One thread = a 2x2 output quad x 8
// channels from the 5x5 T window rows 2Y-1..2Y+3, cols 2X-1..2X+3 (25 loads for 4 outputs instead of 64).
// (An XCD-banded block order and 2x4 strips per thread were measured slower: 164 / 240 us vs 152 us per frame.)
__global__ void fir_bias_act_split_kernel(const float* __restrict__ T, size_t t_stride_n, const float* __restrict__ bias,
                                          const float* __restrict__ next_scale, size_t vec_stride_n,
                                          uint4* __restrict__ y, size_t y_stride_n, int C, int Hin, int Win, float clamp)
{
    const int n = blockIdx.z, cb = blockIdx.y;
    const int bx = blockIdx.x;
    const int Wq = Win;
    const int q = bx * blockDim.x + threadIdx.x;
    if (q >= Hin * Wq) return;
    const int Y = q / Wq, X = q - Y * Wq;
    const int PH = Hin + 1, PW = Win + 1, OH = 2 * Hin, OW = 2 * Win;
    const size_t pplane = (size_t)(C / 8) * PH * PW * 8;
    const float* Tn = T + (size_t)n * t_stride_n + (size_t)cb * PH * PW * 8;
    const float f1[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    float acc[2][2][8];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[a_][b_][c] = 0.f;
#pragma unroll
    for (int rr = 0; rr < 5; ++rr) {
        const int r = 2 * Y - 1 + rr;
        if (r < 0 || r > 2 * Hin) continue;
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) {
            const int c = 2 * X - 1 + cc;
            if (c < 0 || c > 2 * Win) continue;
            const float4* s4 = reinterpret_cast<const float4*>(Tn + (size_t)((r & 1) * 2 + (c & 1)) * pplane + ((size_t)(r >> 1) * PW + (c >> 1)) * 8);
            const float4 u = s4[0], v = s4[1];
            const float tv[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int a_ = rr - dy;                 // tap index: r - (2Y+dy) + 1
                if (a_ < 0 || a_ > 3) continue;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int b_ = cc - dx;
                    if (b_ < 0 || b_ > 3) continue;
                    const float w = f1[a_] * f1[b_];
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) acc[dy][dx][ch] += tv[ch] * w;
                }
            }
        }
    }
    const float4* b4 = reinterpret_cast<const float4*>(bias + (size_t)n * vec_stride_n + cb * 8);
    const float4* n4 = reinterpret_cast<const float4*>(next_scale + (size_t)n * vec_stride_n + cb * 8);
    const float4 b0 = b4[0], b1 = b4[1], n0 = n4[0], n1 = n4[1];
    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, nv[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
    const size_t plane = (size_t)(C / 8) * OH * OW;
    uint4* d = y + (size_t)n * y_stride_n + (size_t)cb * OH * OW;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            h8 hi, lo;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                float t = acc[dy][dx][ch] + bv[ch];
                t = (t < 0.f ? t * 0.2f : t) * 1.4142135623730951f;
                if (clamp >= 0.f) t = fminf(fmaxf(t, -clamp), clamp);
                _Float16 x0, x1; split1(t * nv[ch], x0, x1); hi[ch] = x0; lo[ch] = x1;
            }
            const size_t p = (size_t)(2 * Y + dy) * OW + (2 * X + dx);
            d[p] = *reinterpret_cast<uint4*>(&hi);
            d[plane + p] = *reinterpret_cast<uint4*>(&lo);
        }
}

// ---- image finalize: img_out = upsample2d(img_in) + bias + sum_m partial[m]  (networks_stylegan2.py:463-469) ----
// img_u8 != null: the block is the last one of the network and the frame leaves as uint8 HWC: clamp(-1,1) (the
// `sr_image.clamp(-1, 1)` of triplane.py:136) then ((x + 1) / 2 * 255).int() (inference/real3d_infer.py:472,518-522), fused here
// so that the fp32 image makes no extra round trip; img_out may then be null.
__global__ void rgb_finalize_kernel(const float* __restrict__ img_prev, const float* __restrict__ partial, size_t part_stride_n,
                                    int nparts, const float* __restrict__ brgb, size_t vec_stride_n,
                                    float* __restrict__ img_out, uint8_t* __restrict__ img_u8, int H, int W, float clamp, int up)
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
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = brgb[(size_t)n * vec_stride_n + c];
        for (int m = 0; m < nparts; ++m) t += partial[(size_t)n * part_stride_n + ((size_t)m * 3 + c) * H * W + p];
        if (clamp >= 0.f) t = fminf(fmaxf(t, -clamp), clamp);
        float upv;
        if (up) {
            const float* I = img_prev + ((size_t)n * 3 + c) * Hh * Wh;
            upv = 0.f;
            if (vr0 && vc0) upv += I[(size_t)r0 * Wh + c0] * (wy0 * wx0);
            if (vr0 && vc1) upv += I[(size_t)r0 * Wh + c1] * (wy0 * wx1);
            if (vr1 && vc0) upv += I[(size_t)r1 * Wh + c0] * (wy1 * wx0);
            if (vr1 && vc1) upv += I[(size_t)r1 * Wh + c1] * (wy1 * wx1);
        } else {
            upv = img_prev[((size_t)n * 3 + c) * H * W + p];        // SynthesisBlockNoUp: img.add_(y) at the same resolution
        }
        rgb[c] = upv + t;
        if (img_out) img_out[((size_t)n * 3 + c) * H * W + p] = rgb[c];
    }
    if (img_u8) {
        uint8_t* d = img_u8 + ((size_t)n * H * W + p) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) d[c] = frame_u8(rgb[c]);
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

// ---- alpha / occlusion blend + channel concatenation as the input conversion of the next conv --------------------------
// y = cat([a * m, b * (1 - m)], dim=1) with a per-pixel mask m [N,1,H,W]: the `x = torch.cat([x * head_torso_alpha,
// x_torso * (1 - head_torso_alpha)], dim=1)` and `torch.cat([x * person_occlusion, x_bg * (1 - person_occlusion)])` steps of
// SuperresolutionHybrid8XDC_Warp.forward (modules/real3d/super_resolution/sr_with_ref.py:104,114,126,136), written
// directly in the SPLIT format of fuse_head_torso_convs / fuse_fg_bg_convs' first conv (no fp32 concat tensor).
// blockIdx.y < Ca/8: channels of a (scaled by m); else channels of b (scaled by 1 - m).
// mx: the lo plane receives the fp8 records of R3D_FMT_SPLIT_MX (lo chunk 2G <- xh8 of the 16 channels of group G, lo chunk 2G + 1 <- xl8;
// this thread's 8-channel chunk is half (cb & 1) of both records)
__global__ void blend_cat_to_split_kernel(const float* __restrict__ a, int a_cb8, int Ca, const float* __restrict__ b, int b_cb8, int Cb,
                                          const float* __restrict__ mask, uint4* __restrict__ dst, int HW,
                                          const float* __restrict__ next_scale, size_t next_scale_stride_n, int mx)
{
    const int n = blockIdx.z, cb = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const bool first = cb < Ca / 8;
    const float* src = first ? a : b;
    const int C = first ? Ca : Cb, c8 = first ? cb : cb - Ca / 8, cb8 = first ? a_cb8 : b_cb8;
    const float m = mask[(size_t)n * HW + p];
    const float sc = first ? m : 1.0f - m;
    float v[8];
    if (cb8) {
        const float4* s4 = reinterpret_cast<const float4*>(src + (((size_t)n * (C / 8) + c8) * HW + p) * 8);
        const float4 x0 = s4[0], x1 = s4[1];
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    } else {
        const float* s1 = src + ((size_t)n * C + c8 * 8) * HW + p;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = s1[(size_t)c * HW];
    }
    h8 hi, lo;
    float hf[8], lf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float ns = next_scale ? next_scale[n * next_scale_stride_n + cb * 8 + c] : 1.f;   // consumer's input multiplier (2^e)
        const float t = as_rounded(v[c] * sc * ns);
        const float cl = fminf(fmaxf(t, -65504.f), 65504.f);
        hi[c] = (_Float16)cl; hf[c] = (float)hi[c]; lf[c] = t - hf[c]; lo[c] = (_Float16)lf[c];
    }
    const size_t plane = (size_t)((Ca + Cb) / 8) * HW;
    uint4* d = dst + (size_t)n * 2 * plane + (size_t)cb * HW + p;
    d[0] = *reinterpret_cast<uint4*>(&hi);
    if (mx) {
        unsigned* rec = reinterpret_cast<unsigned*>(dst + (size_t)n * 2 * plane + plane + (size_t)(cb & ~1) * HW + p) + (cb & 1);      // dwords p and 2 + p
        rec[0] = pack4_x8(hf[0] * kMxXh, hf[1] * kMxXh, hf[2] * kMxXh, hf[3] * kMxXh);
        rec[2] = pack4_x8(hf[4] * kMxXh, hf[5] * kMxXh, hf[6] * kMxXh, hf[7] * kMxXh);
        rec[4 * (size_t)HW] = pack4_x8(lf[0] * kMxXl, lf[1] * kMxXl, lf[2] * kMxXl, lf[3] * kMxXl);
        rec[4 * (size_t)HW + 2] = pack4_x8(lf[4] * kMxXl, lf[5] * kMxXl, lf[6] * kMxXl, lf[7] * kMxXl);
    } else {
        d[plane] = *reinterpret_cast<uint4*>(&lo);
    }
}

int blend_cat_to_split_f16x3(const float* a, int a_format, int Ca, const float* b, int b_format, int Cb, const float* mask,
                             int N, int H, int W, void* y_split, int y_format, const float* next_scale, size_t next_scale_stride, hipStream_t st)
{
    ProfScope ps(R3D_PROF_LAYOUT, st);
    // b == nullptr: only the `a` part is written (the other part comes from r3d_conv_forward_cat); the planes still hold (Ca + Cb) / 8 chunks
    hipLaunchKernelGGL(blend_cat_to_split_kernel, dim3((H * W + 255) / 256, (b ? Ca + Cb : Ca) / 8, N), dim3(256), 0, st,
                       a, a_format == R3D_FMT_CB8 ? 1 : 0, Ca, b, b_format == R3D_FMT_CB8 ? 1 : 0, Cb, mask,
                       reinterpret_cast<uint4*>(y_split), H * W, next_scale, next_scale_stride, y_format == R3D_FMT_SPLIT_MX ? 1 : 0);
    return check_launch("blend_cat_to_split");
}

// ---- torch.nn.UpsamplingBilinear2d(scale_factor=2) (align_corners=True) between two conv layers of to_plane_cnn
// (modules/real3d/segformer.py:691-700): fp32 channel-blocked in -> SPLIT (or fp32 channel-blocked) out at 2H x 2W.
// Source index = dst * (in - 1) / (out - 1), lambda in fp32, as ATen's upsample_bilinear2d (area_pixel_compute_scale).
__global__ void upsample2x_bilinear_kernel(const float* __restrict__ x, uint4* __restrict__ y_split, float* __restrict__ y_cb8,
                                           const float* __restrict__ next_scale, size_t next_scale_stride_n, int C, int H, int W, int mx)
{
    const int n = blockIdx.z, cb = blockIdx.y;
    const int OH = 2 * H, OW = 2 * W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= OH * OW) return;
    const int oy = p / OW, ox = p - oy * OW;
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f, sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const float fy = sh * oy, fx = sw * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* X = x + ((size_t)n * (C / 8) + cb) * H * W * 8;
    float v[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const float4 a = *reinterpret_cast<const float4*>(X + ((size_t)y0 * W + x0) * 8 + 4 * half);
        const float4 b = *reinterpret_cast<const float4*>(X + ((size_t)y0 * W + x1) * 8 + 4 * half);
        const float4 c = *reinterpret_cast<const float4*>(X + ((size_t)y1 * W + x0) * 8 + 4 * half);
        const float4 d = *reinterpret_cast<const float4*>(X + ((size_t)y1 * W + x1) * 8 + 4 * half);
        v[4 * half + 0] = hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x);
        v[4 * half + 1] = hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y);
        v[4 * half + 2] = hy * (hx * a.z + lx * b.z) + ly * (hx * c.z + lx * d.z);
        v[4 * half + 3] = hy * (hx * a.w + lx * b.w) + ly * (hx * c.w + lx * d.w);
    }
    if (y_cb8) {
        float4* d = reinterpret_cast<float4*>(y_cb8 + (((size_t)n * (C / 8) + cb) * OH * OW + p) * 8);
        d[0] = make_float4(v[0], v[1], v[2], v[3]); d[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (y_split) {
        h8 hi, lo;
        float hf[8], lf[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float sc = next_scale ? next_scale[n * next_scale_stride_n + cb * 8 + c] : 1.f;
            const float t = as_rounded(v[c] * sc);                 // (= split1, with the fp32 residual kept for the fp8 records)
            const float cl = fminf(fmaxf(t, -65504.f), 65504.f);
            hi[c] = (_Float16)cl; hf[c] = (float)hi[c]; lf[c] = t - hf[c]; lo[c] = (_Float16)lf[c];
        }
        const size_t plane = (size_t)(C / 8) * OH * OW, HW = (size_t)OH * OW;
        uint4* d = y_split + (size_t)n * 2 * plane + (size_t)cb * HW + p;
        d[0] = *reinterpret_cast<uint4*>(&hi);
        if (mx) {                                                   // R3D_FMT_SPLIT_MX: this chunk's dwords (cb & 1) and 2 + (cb & 1) of the group's two records (blend_cat_to_split_kernel)
            unsigned* rec = reinterpret_cast<unsigned*>(y_split + (size_t)n * 2 * plane + plane + (size_t)(cb & ~1) * HW + p) + (cb & 1);
            rec[0] = pack4_x8(hf[0] * kMxXh, hf[1] * kMxXh, hf[2] * kMxXh, hf[3] * kMxXh);
            rec[2] = pack4_x8(hf[4] * kMxXh, hf[5] * kMxXh, hf[6] * kMxXh, hf[7] * kMxXh);
            rec[4 * HW] = pack4_x8(lf[0] * kMxXl, lf[1] * kMxXl, lf[2] * kMxXl, lf[3] * kMxXl);
            rec[4 * HW + 2] = pack4_x8(lf[4] * kMxXl, lf[5] * kMxXl, lf[6] * kMxXl, lf[7] * kMxXl);
        } else {
            d[plane] = *reinterpret_cast<uint4*>(&lo);
        }
    }
}

int upsample2x_bilinear_f16x3(const float* x_cb8, int N, int C, int H, int W, void* y, int y_format,
                              const float* next_scale, size_t next_scale_stride, hipStream_t st)
{
    ProfScope ps(R3D_PROF_LAYOUT, st);
    hipLaunchKernelGGL(upsample2x_bilinear_kernel, dim3((4 * H * W + 255) / 256, C / 8, N), dim3(256), 0, st, x_cb8,
                       (y_format == R3D_FMT_SPLIT || y_format == R3D_FMT_SPLIT_MX) ? reinterpret_cast<uint4*>(y) : nullptr,
                       y_format == R3D_FMT_CB8 ? reinterpret_cast<float*>(y) : nullptr, next_scale, next_scale_stride, C, H, W,
                       y_format == R3D_FMT_SPLIT_MX ? 1 : 0);
    return check_launch("upsample2x_bilinear");
}

// ---- host ------------------------------------------------------------------------------------------------------
// A/B switch of the Winograd F(2,3) conv (r3d_sr_wino.h): R3D_CONV_WINO=0 keeps every plain 3x3 conv on the direct kernels
static int wino_mode() { static const int v = getenv("R3D_CONV_WINO") ? atoi(getenv("R3D_CONV_WINO")) : 3; return v; }   // 0 off, 1 both precisions, 2 f16mx only, 3 f16x3 only
// ... and the shapes it takes: whole 16 x 16-pixel tiles, 16-channel stages, 128-cout blocks
static bool wino_shape_ok(int Cin, int Cout, int H, int W)
{
    // (the kernel addresses one sample's SPLIT activation and the weight pack through 32-bit buffer offsets)
    return wino_mode() && (H & 15) == 0 && (W & 15) == 0 && (Cin & 15) == 0 && (Cout % BLOCK_M) == 0 && (size_t)Cin * H * W * 4 < ((size_t)1 << 31) && (size_t)48 * Cin * Cout < ((size_t)1 << 31);
}
// float offset of conv1's Winograd pack inside an SR block's prepacked buffer (after everything sr_prepack_f16x3 wrote before round 6)
static size_t sr_wino_offset(int Cin, int Cout) { return (size_t)4 * 9 * Cin * Cout + (size_t)9 * Cout * Cout + 2 * conv_tail_layout(Cout).total; }

// prepacked = conv0 (plain layout) ++ conv1 ++ conv0 (fused up-conv layout) ++ ConvTail(conv0) ++ ConvTail(conv1) ++ conv0 (up-conv layout
// with fp8 records, for R3D_FMT_SPLIT_MX inputs; written for R3D_SR_F16MX only) ++ conv0 (plain layout with fp8 records) ++ conv1 (Winograd pack)
int sr_prepack_f16x3(int Cin, int Cout, const float* c0_w, const float* c1_w, void* prepacked, hipStream_t st, bool mx)
{
    float* out = reinterpret_cast<float*>(prepacked);
    const size_t m0 = (size_t)9 * (Cin / 8) * Cout, m1 = (size_t)9 * (Cout / 8) * Cout;
    const ConvTail T = conv_tail_layout(Cout);
    float* tail0 = out + (size_t)2 * 9 * Cin * Cout + (size_t)9 * Cout * Cout;
    float* tail1 = tail0 + T.total;
    hipLaunchKernelGGL(weight_row_stats_kernel, dim3(Cout), dim3(256), 0, st, c0_w, Cin * 9, Cout, Cout, tail0);
    hipLaunchKernelGGL(weight_row_stats_kernel, dim3(Cout), dim3(256), 0, st, c1_w, Cout * 9, Cout, Cout, tail1);
    hipLaunchKernelGGL(sr_prepack_f16_kernel, dim3((unsigned)((m0 + 255) / 256)), dim3(256), 0, st, c0_w, Cin, Cout, 9, Cin, Cout,
                       tail0 + T.winv, reinterpret_cast<uint4*>(out));
    if (mx)         // conv1 consumes the fp8 records of the up-sampling conv's epilogue (R3D_SR_F16MX)
        hipLaunchKernelGGL(sr_prepack_mx_kernel, dim3((unsigned)((m1 / 2 + 255) / 256)), dim3(256), 0, st, c1_w, Cout, Cout, 9, Cout, Cout,
                           tail1 + T.winv, reinterpret_cast<uint4*>(out + (size_t)9 * Cin * Cout));
    else
        hipLaunchKernelGGL(sr_prepack_f16_kernel, dim3((unsigned)((m1 + 255) / 256)), dim3(256), 0, st, c1_w, Cout, Cout, 9, Cout, Cout,
                           tail1 + T.winv, reinterpret_cast<uint4*>(out + (size_t)9 * Cin * Cout));
    // conv0 again in the fused up-conv layout (SynthesisBlock; the plain layout above serves SynthesisBlockNoUp)
    const size_t mu = (size_t)9 * (Cin / 8) * Cout * 2;
    hipLaunchKernelGGL(sr_prepack_up_kernel, dim3((unsigned)((mu + 255) / 256)), dim3(256), 0, st, c0_w, Cin, Cout, tail0 + T.winv,
                       reinterpret_cast<uint4*>(out + (size_t)9 * Cin * Cout + (size_t)9 * Cout * Cout));
    if (mx) {
        hipLaunchKernelGGL(sr_prepack_up_mx_kernel, dim3((unsigned)((mu + 255) / 256)), dim3(256), 0, st, c0_w, Cin, Cout, tail0 + T.winv,
                           reinterpret_cast<uint4*>(tail1 + T.total));
        if ((Cin & 15) == 0)        // conv0 in the plain layout with fp8 records: SynthesisBlockNoUp's conv0 on an R3D_FMT_SPLIT_MX input
            hipLaunchKernelGGL(sr_prepack_mx_kernel, dim3((unsigned)((m0 / 2 + 255) / 256)), dim3(256), 0, st, c0_w, Cin, Cout, 9, Cin, Cout,
                               tail0 + T.winv, reinterpret_cast<uint4*>(tail1 + T.total + (size_t)9 * Cin * Cout));
    }
    {   // conv1 for conv_wino_f16x3_kernel: 12 transformed tap matrices [Cout / 128][Cout / 16][wave 8][WG_WBLK]
        const size_t mw = (size_t)(Cout >> 7) * (Cout >> 4) * 8 * 3 * 2 * 32;
        hipLaunchKernelGGL(sr_prepack_wino_kernel, dim3((unsigned)((mw + 255) / 256)), dim3(256), 0, st, c1_w, Cout, Cout, Cout, Cout, tail1 + T.winv,
                           reinterpret_cast<uint4*>(out + sr_wino_offset(Cin, Cout)), mx ? 1 : 0);
    }
    return check_launch("sr_block_prepack");
}

// tuning switch (experiment): extra dynamic LDS per block of the two big kernels, e.g. 24576 -> one block per CU, which leaves room for a
// ray-kernel block of another stream on the same CU
static unsigned lds_pad() { static const unsigned v = getenv("R3D_SR_LDS_PAD") ? (unsigned)atoi(getenv("R3D_SR_LDS_PAD")) : 0u; return v; }

// wino: a.wp is a Winograd pack (sr_prepack_wino_kernel), a.x plain SPLIT; mx selects the f16mx main loop
static void launch_conv2(Conv2Args& a, int tiles, int N, hipStream_t st, bool mx = false, bool wino = false)
{
    a.clk = prof_clock_slot(R3D_PROF_CONV);
    dim3 grid(tiles, a.Cout / BLOCK_M, N * a.nphase);
    if (wino) {
        static const int order = getenv("R3D_CONV_ORDER") ? atoi(getenv("R3D_CONV_ORDER")) : 2;
        a.order = (tiles & 7) == 0 ? order : 0;
        if (mx) hipLaunchKernelGGL(conv_wino_f16x3_kernel<true>, grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL(conv_wino_f16x3_kernel<false>, grid, dim3(512), 0, st, a);
        return;
    }
    static const int rows8 = getenv("R3D_CONV_ROWS8") ? atoi(getenv("R3D_CONV_ROWS8")) : 1;   // A/B switch: 0 = always 16x16 tiles
    if (rows8 && (rows8 == 1 || (rows8 == 2 && !a.rgb_partial) || (rows8 == 3 && a.rgb_partial)) && a.nphase == 1 && a.ph[0].ntaps == 9 && (size_t)tiles * grid.y * grid.z <= 256) {
        // under-filled launch (<= half of the 512 block slots): 8x16-pixel tiles, twice the blocks (bit-identical results)
        const int t8 = ((a.ph[0].outW + F_TILE_W - 1) / F_TILE_W) * ((a.ph[0].outH + 7) / 8);
        a.order = (t8 & 7) == 0 ? 2 : 0;
        if (mx) hipLaunchKernelGGL(conv_mfma_f16x3_rows8_kernel<true>, dim3(t8, grid.y, grid.z), dim3(512), 0, st, a);     // (round 4: to_plane_cnn's 128^2 layers on MX)
        else hipLaunchKernelGGL(conv_mfma_f16x3_rows8_kernel<false>, dim3(t8, grid.y, grid.z), dim3(512), 0, st, a);
        return;
    }
    static const int shape = getenv("R3D_CONV_SHAPE") ? atoi(getenv("R3D_CONV_SHAPE")) : 0;   // tuning switch, default 0
    static const int order = getenv("R3D_CONV_ORDER") ? atoi(getenv("R3D_CONV_ORDER")) : 2;   // tuning switch (0: plain (x, y) order)
    a.order = (tiles & 7) == 0 ? order : 0;
    const int kind = a.nphase > 1 ? 2 : (a.ph[0].ntaps == 9 ? 0 : 1);      // 0: 3x3 conv, 1: 1x1 conv, 2: transposed-conv phases
    if (shape == 1 && !mx) {    // 4 waves x (64 couts x 128 px), 2 blocks/CU (the switch has no MX instantiation: an R3D_FMT_SPLIT_MX input keeps the default shape)
        if (kind == 0) hipLaunchKernelGGL((conv_mfma_f16x3_kernel<2, 4, 2>), grid, dim3(256), 0, st, a);
        else if (kind == 1) hipLaunchKernelGGL((conv1x1_mfma_f16x3_kernel<2, 4, 2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((tconv_mfma_f16x3_kernel<2, 4, 2>), grid, dim3(256), 0, st, a);
    } else {                    // 8 waves x (64 couts x 64 px): 4 waves/SIMD at 2 blocks/CU
        if (kind == 0 && mx) hipLaunchKernelGGL((conv_mfma_f16x3_kernel<4, 2, 4, true>), grid, dim3(512), 0, st, a);
        else if (kind == 0) hipLaunchKernelGGL((conv_mfma_f16x3_kernel<4, 2, 4>), grid, dim3(512), lds_pad(), st, a);
        else if (kind == 1) hipLaunchKernelGGL((conv1x1_mfma_f16x3_kernel<4, 2, 4>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((tconv_mfma_f16x3_kernel<4, 2, 4>), grid, dim3(512), 0, st, a);
    }
}

static int tiles_of(int H, int W) { return ((W + F_TILE_W - 1) / F_TILE_W) * ((H + F_TILE_H - 1) / F_TILE_H); }

int sr_block_forward_f16x3(const void* prepacked, const void* styles, int N, int Cin, int Cout, int Hin, int Win, int up,
                           const void* x, int x_format, const float* img, float clamp,
                           void* x_out, int x_out_format, const float* next_scale, size_t next_scale_stride,
                           float* img_out, uint8_t* img_u8, float* x_absmax, void* workspace, size_t workspace_bytes, hipStream_t st, bool mx)
{
    (void)workspace_bytes;
    const SrStyleLayout L = sr_style_layout(Cin, Cout);
    const float* pk = reinterpret_cast<const float*>(styles);
    const float* wpk = reinterpret_cast<const float*>(prepacked);
    const int OH = up ? 2 * Hin : Hin, OW = up ? 2 * Win : Win;
    char* wsb = reinterpret_cast<char*>(workspace);
    uint4* xin = reinterpret_cast<uint4*>(wsb); wsb += align256((size_t)N * Cin * Hin * Win * 4);
    const int PH = Hin + 1, PW = Win + 1;
    const size_t pplane = (size_t)Cout * PH * PW;                 // floats per phase plane
    float* T = reinterpret_cast<float*>(wsb);   wsb += align256((size_t)N * 4 * pplane * 4);
    uint4* y0 = reinterpret_cast<uint4*>(wsb);  wsb += align256((size_t)N * Cout * 4 * Hin * Win * 4);
    float* xo = reinterpret_cast<float*>(wsb);                    // fp32 CB8 x (only when an fp32 x_out is requested)
    float* rgbp = reinterpret_cast<float*>(wsb + align256((size_t)N * Cout * 4 * Hin * Win * 4));

    const uint4* xs = reinterpret_cast<const uint4*>(x);
    const bool mx_in = x_format == R3D_FMT_SPLIT_MX;          // (validated by the caller: up = 1, R3D_SR_F16MX)
    if (x_format != R3D_FMT_SPLIT && !mx_in) {
        ProfScope ps(R3D_PROF_LAYOUT, st);
        hipLaunchKernelGGL(to_split_kernel, dim3((Hin * Win + 255) / 256, Cin / 8, N), dim3(256), 0, st,
                           reinterpret_cast<const float*>(x), x_format == R3D_FMT_CB8 ? 1 : 0, pk + L.s0f, L.total, xin, Cin, Cin, Hin * Win);
        xs = xin;
    }
    // conv1 on the Winograd F(2,3) kernel: its operand is transformed in fp32 inside the kernel, so conv0 hands over plain SPLIT (no fp8 records)
    const bool wino1 = wino_shape_ok(Cout, Cout, OH, OW) && (wino_mode() == 1 || (wino_mode() == 2 && mx) || (wino_mode() == 3 && !mx));
    const bool mx0 = mx && !wino1;                            // does conv0's epilogue write the fp8 records of conv1's operand?
    static const int fused_up = getenv("R3D_UPCONV") ? atoi(getenv("R3D_UPCONV")) : 1;   // A/B switch: 0 = per-phase T-conv + FIR kernel
    if (up && fused_up) {
        // ---- conv0: fused transposed conv + FIR + bias + lrelu -> SPLIT (one kernel, T stays on chip) ----------------
        UpArgs u = {};
        u.x = xs; u.x_stride_n = (size_t)Cin / 8 * Hin * Win * 2;
        u.wp = reinterpret_cast<const uint4*>(wpk + (size_t)9 * Cin * Cout + (size_t)9 * Cout * Cout);
        if (mx_in) u.wp = reinterpret_cast<const uint4*>(wpk + (size_t)2 * 9 * Cin * Cout + (size_t)9 * Cout * Cout + 2 * conv_tail_layout(Cout).total);
        u.out_scale = pk + L.d0f; u.bias = pk + L.b0; u.next_scale = pk + L.s1f; u.vec_stride_n = L.total;
        u.y = y0; u.y_stride_n = (size_t)Cout / 8 * OH * OW * 2;
        u.Cin = Cin; u.Cout = Cout; u.H = Hin; u.W = Win;
        u.tiles_x = (Win + U_TILE - 1) / U_TILE;
        u.ntiles = u.tiles_x * ((Hin + U_TILE - 1) / U_TILE);
        u.tiles_per_xcd = (u.ntiles + 7) / 8;
        u.clamp = clamp;
        u.clk = prof_clock_slot(R3D_PROF_UPCONV);
        ProfScope ps(R3D_PROF_UPCONV, st);
        const dim3 ugrid(8 * u.tiles_per_xcd * (Cout / 32), N);
        // (NW = 8 -- one N tile per wave, 4 waves per SIMD -- was built for block0.conv0, the all-epilogue launch, and measured NEUTRAL in
        // round 4 (0.173 vs 0.169-0.177 ms per frame for the family, bit-identical results), as was a first-round stagger of the CU's second
        // block: that launch is bound by the LDS-DMA fill rate of its two K stages and by the VALU work of its epilogue one after the other,
        // not by latency.  The instantiation stays available behind R3D_UPCONV_NW8=1 for experiments.)
        static const int up8 = getenv("R3D_UPCONV_NW8") ? atoi(getenv("R3D_UPCONV_NW8")) : 0;
        const bool nw8 = up8 && Cin <= 64 && !mx_in && !mx0 && clamp < 0.f;
        if (mx_in && clamp >= 0.f && mx0) hipLaunchKernelGGL((upconv_fir_f16x3_kernel<true, true, true>), ugrid, dim3(256), 0, st, u);
        else if (mx_in && mx0) hipLaunchKernelGGL((upconv_fir_f16x3_kernel<false, true, true>), ugrid, dim3(256), 0, st, u);
        else if (mx_in && clamp >= 0.f) hipLaunchKernelGGL((upconv_fir_f16x3_kernel<true, false, true>), ugrid, dim3(256), 0, st, u);
        else if (mx_in) hipLaunchKernelGGL((upconv_fir_f16x3_kernel<false, false, true>), ugrid, dim3(256), 0, st, u);
        else if (nw8) hipLaunchKernelGGL((upconv_fir_f16x3_kernel<false, false, false, 8>), ugrid, dim3(512), 0, st, u);
        else if (mx0 && clamp >= 0.f) hipLaunchKernelGGL((upconv_fir_f16x3_kernel<true, true>), ugrid, dim3(256), 0, st, u);
        else if (mx0) hipLaunchKernelGGL((upconv_fir_f16x3_kernel<false, true>), ugrid, dim3(256), 0, st, u);
        else if (clamp >= 0.f) hipLaunchKernelGGL(upconv_fir_f16x3_kernel<true>, ugrid, dim3(256), 0, st, u);
        else hipLaunchKernelGGL(upconv_fir_f16x3_kernel<false>, ugrid, dim3(256), lds_pad(), st, u);
    } else if (up) {
        // ---- conv0: stride-2 transposed conv as 4 phases -> T (demodulated, fp32), then FIR + bias + lrelu -> SPLIT ----
        {
            Conv2Args a = {};
            a.x = xs; a.x_stride_n = (size_t)Cin / 8 * Hin * Win * 2;
            a.wp = reinterpret_cast<const uint4*>(wpk);
            a.out_scale = pk + L.d0f; a.out_scale_stride_n = L.total;
            a.y_f32 = T; a.y_f32_stride_n = 4 * pplane; a.OH = PH; a.OW = PW;
            a.Cin = Cin; a.Cout = Cout; a.CoutReal = Cout; a.H = Hin; a.W = Win; a.nphase = 4; a.act = 0; a.clamp = -1.f;
            sr_fill_tconv_phases(a.ph, Hin, Win);
            int maxtiles = 0;
            for (int p = 0; p < 4; ++p) {                               // phase-major T: contiguous stores per phase
                a.ph[p].oy_mul = 1; a.ph[p].oy_add = 0; a.ph[p].ox_mul = 1; a.ph[p].ox_add = 0; a.ph[p].out_off = p * pplane;
                const int tiles = tiles_of(a.ph[p].outH, a.ph[p].outW);
                if (tiles > maxtiles) maxtiles = tiles;
            }
            ProfScope ps(R3D_PROF_CONV, st);
            launch_conv2(a, maxtiles, N, st);
        }
        ProfScope ps(R3D_PROF_UPCONV, st);
        hipLaunchKernelGGL(fir_bias_act_split_kernel, dim3((Hin * Win + 255) / 256, Cout / 8, N), dim3(256), 0, st,
                           T, 4 * pplane, pk + L.b0, pk + L.s1f, L.total, y0, (size_t)Cout / 8 * OH * OW * 2, Cout, Hin, Win, clamp);
    } else {
        // ---- conv0 of SynthesisBlockNoUp (superresolution.py:159-258): plain modulated 3x3 conv -> SPLIT for conv1 ----
        Conv2Args a = {};
        a.x = xs; a.x_stride_n = (size_t)Cin / 8 * Hin * Win * 2;
        a.wp = reinterpret_cast<const uint4*>(wpk);
        a.out_scale = pk + L.d0f; a.out_scale_stride_n = L.total; a.bias = pk + L.b0; a.bias_stride_n = L.total;
        a.OH = OH; a.OW = OW;
        a.y_split = y0; a.y_split_stride_n = (size_t)Cout / 8 * OH * OW * 2; a.next_scale = pk + L.s1f; a.next_scale_stride_n = L.total;
        a.y_split_mx = mx0 ? 1 : 0;                                  // f16mx: conv1 reads fp8 records (the Winograd kernel: plain SPLIT)
        if (mx_in) a.wp = reinterpret_cast<const uint4*>(wpk + (size_t)3 * 9 * Cin * Cout + (size_t)9 * Cout * Cout + 2 * conv_tail_layout(Cout).total);
        a.Cin = Cin; a.Cout = Cout; a.CoutReal = Cout; a.H = Hin; a.W = Win; a.nphase = 1;
        a.act = 1; a.act_slope = 0.2f; a.act_gain = 1.4142135623730951f; a.clamp = clamp;
        sr_fill_conv3x3_phase(a.ph, OH, OW);
        ProfScope ps(R3D_PROF_CONV, st);
        launch_conv2(a, tiles_of(OH, OW), N, st, mx_in);
    }
    // ---- conv1 (3x3) + bias/lrelu + toRGB partials (+ optional x outputs) ------------------------------------------
    {
        Conv2Args a = {};
        a.x = y0; a.x_stride_n = (size_t)Cout / 8 * OH * OW * 2;
        a.wp = reinterpret_cast<const uint4*>(wpk + (size_t)9 * Cin * Cout);
        a.out_scale = pk + L.d1f; a.out_scale_stride_n = L.total; a.bias = pk + L.b1; a.bias_stride_n = L.total;
        a.OH = OH; a.OW = OW;
        if (x_out && x_out_format == R3D_FMT_CB8) { a.y_f32 = reinterpret_cast<float*>(x_out); a.y_f32_stride_n = (size_t)Cout * OH * OW; }
        if (x_out && x_out_format == R3D_FMT_NCHW) { a.y_nchw = reinterpret_cast<float*>(x_out); a.y_nchw_stride_n = (size_t)Cout * OH * OW; }
        if (x_out && (x_out_format == R3D_FMT_SPLIT || x_out_format == R3D_FMT_SPLIT_MX)) {
            a.y_split = reinterpret_cast<uint4*>(x_out); a.y_split_stride_n = (size_t)Cout / 8 * OH * OW * 2;
            a.next_scale = next_scale; a.next_scale_stride_n = next_scale_stride;
            a.y_split_mx = x_out_format == R3D_FMT_SPLIT_MX ? 1 : 0;
        }
        a.y_absmax = reinterpret_cast<unsigned*>(x_absmax);
        a.wrgb = pk + L.wrgb; a.wrgb_stride_n = L.total; a.rgb_partial = rgbp; a.rgbp_stride_n = (size_t)(Cout / 64) * 3 * OH * OW;
        a.Cin = Cout; a.Cout = Cout; a.CoutReal = Cout; a.H = OH; a.W = OW; a.nphase = 1;
        a.act = 1; a.act_slope = 0.2f; a.act_gain = 1.4142135623730951f; a.clamp = clamp;
        sr_fill_conv3x3_phase(a.ph, OH, OW);
        if (wino1) a.wp = reinterpret_cast<const uint4*>(wpk + sr_wino_offset(Cin, Cout));
        ProfScope ps(R3D_PROF_CONV, st);
        launch_conv2(a, tiles_of(OH, OW), N, st, mx, wino1);
    }
    (void)xo;
    {
        ProfScope ps(R3D_PROF_TORGB, st);
        hipLaunchKernelGGL(rgb_finalize_kernel, dim3((OH * OW + 255) / 256, N), dim3(256), 0, st, img, rgbp,
                           (size_t)(Cout / 64) * 3 * OH * OW, Cout / 64, pk + L.brgb, L.total, img_out, img_u8, OH, OW, clamp, up);
    }
    return check_launch("sr_block_forward(f16x3)");
}

// ---- generic convolution layer (nn.Conv2d k = 1 | 3, stride 1, padding k/2, + bias + optional LeakyReLU) on the same kernel:
// the torso / background fusion stacks of SuperresolutionHybrid8XDC_Warp (modules/real3d/super_resolution/sr_with_ref.py:24-63)
static inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

size_t conv_prepacked_bytes_f16x3(int Cin, int Cout, int ksize)
{
    const int Co = pad_to(Cout, BLOCK_M);
    const size_t w = (size_t)(ksize * ksize) * pad_to(Cin, 16) * Co;
    // 3x3: the weights again with fp8 records (R3D_FMT_SPLIT_MX inputs) and as the 12 transformed tap matrices of the Winograd F(2,3) kernel (plain SPLIT inputs, r3d_sr_wino.h)
    return (w + conv_tail_layout(Co).total + (ksize == 3 ? w + (size_t)12 * pad_to(Cin, 16) * Co : 0)) * sizeof(float);
}

// prepacked = split weights (rows pre-scaled by 2^kw[co]) ++ ConvTail {2^-kw[co], sum|w[co]|} ++ (3x3 only) the weights in the f16mx layout
int conv_prepack_f16x3(const float* w, int Cin, int Cout, int ksize, void* prepacked, hipStream_t st)
{
    const int Ci = pad_to(Cin, 16), Co = pad_to(Cout, BLOCK_M), nt = ksize * ksize;
    const size_t m = (size_t)nt * (Ci / 8) * Co;
    float* tail = reinterpret_cast<float*>(prepacked) + (size_t)nt * Ci * Co;
    ProfScope ps(R3D_PROF_PACK, st);
    hipLaunchKernelGGL(weight_row_stats_kernel, dim3(Co), dim3(256), 0, st, w, Cin * nt, Cout, Co, tail);
    hipLaunchKernelGGL(sr_prepack_f16_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, w, Cin, Cout, nt, Ci, Co,
                       tail + conv_tail_layout(Co).winv, reinterpret_cast<uint4*>(prepacked));
    if (ksize == 3) {
        hipLaunchKernelGGL(sr_prepack_mx_kernel, dim3((unsigned)((m / 2 + 255) / 256)), dim3(256), 0, st, w, Cin, Cout, nt, Ci, Co,
                           tail + conv_tail_layout(Co).winv, reinterpret_cast<uint4*>(tail + conv_tail_layout(Co).total));
        const size_t mw = (size_t)(Co >> 7) * (Ci >> 4) * 8 * 3 * 2 * 32;
        hipLaunchKernelGGL(sr_prepack_wino_kernel, dim3((unsigned)((mw + 255) / 256)), dim3(256), 0, st, w, Cin, Cout, Ci, Co, tail + conv_tail_layout(Co).winv,
                           reinterpret_cast<uint4*>(tail + conv_tail_layout(Co).total + (size_t)nt * Ci * Co), 0);      // (a plain SPLIT input: the f16x3 form)
    }
    return check_launch("conv_prepack");
}

size_t conv_workspace_bytes_f16x3(int N, int Cin, int H, int W)
{
    return align256((size_t)N * pad_to(Cin, 16) * H * W * 4) + 256;
}

// scales: per-sample ConvScales vectors written by r3d_conv_chain_scales (in_vec: multiplier of the fp32 -> SPLIT input
// conversion; out_vec: epilogue multiplier 2^-kw[co] * 2^-e_in); bias [Cout] shared by the batch (or null)
int conv_forward_f16x3(const void* prepacked, const float* scales, size_t scales_stride, const float* bias,
                       int N, int Cin, int Cout, int H, int W, int ksize,
                       const void* x, int x_format, int act, float slope, float gain, float clamp,
                       void* y, int y_format, const float* next_scale, size_t next_scale_stride, float* y_absmax,
                       void* workspace, hipStream_t st, const ConvCat* cat)
{
    const int Ci = pad_to(Cin, 16), Co = pad_to(Cout, BLOCK_M);
    const ConvScales S = conv_scales_layout(Ci, Co);
    const uint4* xs = reinterpret_cast<const uint4*>(x);
    const bool mx_in = x_format == R3D_FMT_SPLIT_MX;          // (validated by the caller: ksize 3, Cin % 16 == 0): the f16mx main loop
    if (x_format != R3D_FMT_SPLIT && !mx_in) {
        uint4* xin = reinterpret_cast<uint4*>(workspace);
        ProfScope ps(R3D_PROF_LAYOUT, st);
        hipLaunchKernelGGL(to_split_kernel, dim3((H * W + 255) / 256, Ci / 8, N), dim3(256), 0, st,
                           reinterpret_cast<const float*>(x), x_format == R3D_FMT_CB8 ? 1 : 0, scales + S.in_vec, scales_stride, xin, Ci, Cin, H * W);
        xs = xin;
    }
    Conv2Args a = {};
    a.x = xs; a.x_stride_n = (size_t)Ci / 8 * H * W * 2;
    a.wp = reinterpret_cast<const uint4*>(prepacked);
    if (mx_in) a.wp = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(prepacked) + (size_t)ksize * ksize * Ci * Co + conv_tail_layout(Co).total);
    a.out_scale = scales + S.out_vec; a.out_scale_stride_n = scales_stride; a.bias = bias; a.bias_stride_n = 0;
    a.OH = H; a.OW = W;
    if (y_format == R3D_FMT_CB8) { a.y_f32 = reinterpret_cast<float*>(y); a.y_f32_stride_n = (size_t)Cout * H * W; }
    else if (y_format == R3D_FMT_NCHW) { a.y_nchw = reinterpret_cast<float*>(y); a.y_nchw_stride_n = (size_t)Cout * H * W; }
    else { a.y_split = reinterpret_cast<uint4*>(y); a.y_split_stride_n = (size_t)Cout / 8 * H * W * 2; a.next_scale = next_scale; a.next_scale_stride_n = next_scale_stride;
           a.y_split_mx = y_format == R3D_FMT_SPLIT_MX ? 1 : 0;
           if (cat) {      // y is the whole concatenated tensor [N][hi|lo][C_total / 8][H][W][8]; next_scale the consumer's whole in-multiplier vector
               a.y_split_stride_n = (size_t)cat->C_total / 8 * H * W * 2; a.y_cat_chunks = cat->C_total / 8; a.y_cat_off = cat->chan_off / 8;
               a.y_mask = cat->mask; a.y_mask_invert = cat->mask_invert;
               if (next_scale) a.next_scale = next_scale + cat->chan_off;
           } }
    a.y_absmax = reinterpret_cast<unsigned*>(y_absmax);
    a.Cin = Ci; a.Cout = Co; a.CoutReal = Cout; a.H = H; a.W = W; a.nphase = 1;
    a.act = act; a.act_slope = slope; a.act_gain = gain; a.clamp = clamp;
    if (ksize == 3) sr_fill_conv3x3_phase(a.ph, H, W);
    else {
        ConvPhase& p = a.ph[0];
        p.outH = H; p.outW = W; p.oy_mul = 1; p.oy_add = 0; p.ox_mul = 1; p.ox_add = 0; p.out_off = 0;
        p.ntaps = 1; p.dy[0] = 0; p.dx[0] = 0; p.widx[0] = 0;
    }
    // a plain SPLIT operand (the f16x3 precision, or a producer that does not write records): Winograd F(2,3) when the shape allows it
    const bool wino = ksize == 3 && !mx_in && wino_shape_ok(Ci, Co, H, W) && (wino_mode() == 1 || wino_mode() == 3);
    if (wino) a.wp = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(prepacked) + (size_t)2 * 9 * Ci * Co + conv_tail_layout(Co).total);
    ProfScope ps(R3D_PROF_CONV, st);
    launch_conv2(a, tiles_of(H, W), N, st, mx_in, wino);
    return check_launch("conv_forward");
}

// cat([a * mask, b * (1 - mask)]) -> 1x1 conv in one kernel (conv1x1_blend_f16x3_kernel); the caller validated: ksize 1, a / b CB8, Ca % 8 == Cb % 8 == 0, (Ca + Cb) % 64 == 0
int conv_forward_blend_f16x3(const void* prepacked, const float* scales, size_t scales_stride, const float* bias,
                             int N, int Ca, int Cb, int Cout, int H, int W, const float* xa, const float* xb, const float* mask,
                             int act, float slope, float gain, float clamp,
                             void* y, int y_format, const float* next_scale, size_t next_scale_stride, float* y_absmax, hipStream_t st)
{
    const int Ci = Ca + Cb, Co = pad_to(Cout, BLOCK_M);
    const ConvScales S = conv_scales_layout(Ci, Co);
    Conv2Args a = {};
    a.wp = reinterpret_cast<const uint4*>(prepacked);
    a.bl_a = xa; a.bl_b = xb; a.bl_mask = mask; a.bl_scale = scales + S.in_vec; a.bl_scale_stride_n = scales_stride; a.bl_Ca = Ca;
    a.out_scale = scales + S.out_vec; a.out_scale_stride_n = scales_stride; a.bias = bias; a.bias_stride_n = 0;
    a.OH = H; a.OW = W;
    if (y_format == R3D_FMT_CB8) { a.y_f32 = reinterpret_cast<float*>(y); a.y_f32_stride_n = (size_t)Cout * H * W; }
    else if (y_format == R3D_FMT_NCHW) { a.y_nchw = reinterpret_cast<float*>(y); a.y_nchw_stride_n = (size_t)Cout * H * W; }
    else { a.y_split = reinterpret_cast<uint4*>(y); a.y_split_stride_n = (size_t)Cout / 8 * H * W * 2; a.next_scale = next_scale; a.next_scale_stride_n = next_scale_stride;
           a.y_split_mx = y_format == R3D_FMT_SPLIT_MX ? 1 : 0; }
    a.y_absmax = reinterpret_cast<unsigned*>(y_absmax);
    a.Cin = Ci; a.Cout = Co; a.CoutReal = Cout; a.H = H; a.W = W; a.nphase = 1;
    a.act = act; a.act_slope = slope; a.act_gain = gain; a.clamp = clamp;
    ConvPhase& p = a.ph[0];
    p.outH = H; p.outW = W; p.oy_mul = 1; p.oy_add = 0; p.ox_mul = 1; p.ox_add = 0; p.out_off = 0;
    p.ntaps = 1; p.dy[0] = 0; p.dx[0] = 0; p.widx[0] = 0;
    ProfScope ps(R3D_PROF_CONV, st);
    hipLaunchKernelGGL(conv1x1_blend_f16x3_kernel, dim3(tiles_of(H, W), Co / BLOCK_M, N), dim3(512), 0, st, a);
    return check_launch("conv_forward_blend");
}

}  // namespace r3d
