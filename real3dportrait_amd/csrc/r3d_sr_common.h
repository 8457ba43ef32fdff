// Declarations shared by the f32 (r3d_sr.hip) and f16x3 (r3d_sr_f16x3.hip) super-resolution kernels.
#pragma once
#include "r3d_common.h"

namespace r3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
static constexpr int BLOCK_M = 128;                    // output channels per conv block
static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- fp16 range management of the f16x3 path (exact power-of-two pre-scaling) --------------------------------------
// The 3-term fp16 split x = hi + lo is fp32-accurate while the STORED tensor has rms >= 2^-3 and max < 65504 (16 binades);
// the reference runs these layers in fp32 with conv_clamp=None, i.e. with no range limit at all.  Every fp16 operand is
// therefore stored times a power of two chosen from a GUARANTEED bound, and the factor is taken out again in fp32:
//   weights      row co is stored as w * 2^kw[co] with max|w[co]| * 2^kw in [2^10, 2^11)          (static, at prepack)
//   activations  tensor A (per sample) is stored as A * s'[ci], s' = s * 2^e, with  max|s'| * B(A) in [2^14, 2^15),
//                B(A) >= max|A| propagated layer to layer:  B_out = gain * max_co(|b[co]| + d[co] * sum|w[co] s| * B_in)
//   epilogue     acc * (d[co] * 2^-kw[co] * 2^-e)
// All factors are powers of two, so the result equals the unscaled computation up to the rounding of the lo terms.
static constexpr int kWeightTargetExp = 10;            // stored max|w[co]| in [2^10, 2^11)
static constexpr int kActTargetExp = 15;               // stored activation bound in [2^14, 2^15)

// exponent e with 2^(e-1) <= m < 2^e for finite m > 0 (frexp exponent); 0 for m == 0 / non-finite
__host__ __device__ inline int pow2_ceil_exp(float m)
{
    if (!(m > 0.f) || !(m < 3.0e38f)) return 0;
    int e;
    (void)frexpf(m, &e);
    return e;
}
__host__ __device__ inline float pow2f(int e) { return ldexpf(1.0f, e < -120 ? -120 : (e > 120 ? 120 : e)); }
__host__ __device__ inline int weight_row_exp(float rowmax) { return rowmax > 0.f ? kWeightTargetExp + 1 - pow2_ceil_exp(rowmax) : 0; }
// multiplier exponent for an activation with bound B whose per-channel multipliers have max |s|max (0 -> treated as 1)
__host__ __device__ inline int act_exp(float B, float smax)
{
    const int eb = B > 0.f ? pow2_ceil_exp(B) : 0, es = smax > 0.f ? pow2_ceil_exp(smax) : 0;
    return kActTargetExp - eb - es;
}

// Per-sample vectors of one SR block (floats).  The f16x3 kernels read the FOLDED vectors (s0f, s1f, d0f, d1f); the exact-f32
// kernels read the raw ones.  s0f comes first: a producer that emits this block's input in SPLIT format multiplies by the
// vector at the start of the buffer.
struct SrStyleLayout { size_t s0f, s1f, d0f, d1f, s0, s1, s2, d0, d1, wrgb, b0, b1, brgb, c0, c1, wi0, wi1, meta, total; };
static __host__ __device__ inline SrStyleLayout sr_style_layout(int Cin, int Cout)
{
    SrStyleLayout L;
    size_t o = 0;
    L.s0f = o; o += Cin;
    L.s1f = o; o += Cout;
    L.d0f = o; o += Cout;
    L.d1f = o; o += Cout;
    L.s0 = o; o += Cin;
    L.s1 = o; o += Cout;
    L.s2 = o; o += Cout;
    L.d0 = o; o += Cout;
    L.d1 = o; o += Cout;
    L.wrgb = o; o += (size_t)3 * Cout;
    L.b0 = o; o += Cout;
    L.b1 = o; o += Cout;
    L.brgb = o; o += 4;
    L.c0 = o; o += Cout;            // bound coefficient of conv0: d0[co] * sum_{ci,k} |w0[co,ci,k] s0[ci]|
    L.c1 = o; o += Cout;
    L.wi0 = o; o += Cout;           // 2^-kw[co] of conv0 / conv1 (same rule as the prepack kernels)
    L.wi1 = o; o += Cout;
    L.meta = o; o += 8;             // [0] bound_in, [1] bound after conv0, [2] bound_out, [3] e0, [4] e1, [5] smax0, [6] smax1
    L.total = (o + 3) & ~(size_t)3;
    return L;
}
enum { SR_META_BOUND_IN = 0, SR_META_BOUND_MID = 1, SR_META_BOUND_OUT = 2, SR_META_E0 = 3, SR_META_E1 = 4, SR_META_SMAX0 = 5, SR_META_SMAX1 = 6 };

// Tail of a prepacked plain conv (r3d_conv_prepack): per padded cout {2^-kw, sum|w|, max|w|}
struct ConvTail { size_t winv, l1, total; };
static __host__ __device__ inline ConvTail conv_tail_layout(int CoutPadded)
{
    ConvTail T; T.winv = 0; T.l1 = CoutPadded; T.total = 2 * (size_t)CoutPadded; return T;
}
// Per-sample scale vectors of one plain conv call (r3d_conv_chain_scales): in_vec[CinPadded] (what the producer of this conv's
// SPLIT input multiplies by), out_vec[CoutPadded] (epilogue), meta {bound_in, bound_out, e_in}
struct ConvScales { size_t in_vec, out_vec, meta, total; };
static __host__ __device__ inline ConvScales conv_scales_layout(int CinPadded, int CoutPadded)
{
    ConvScales S; S.in_vec = 0; S.out_vec = CinPadded; S.meta = (size_t)CinPadded + CoutPadded; S.total = (S.meta + 4 + 3) & ~(size_t)3; return S;
}

// A "phase" is a set of taps writing to a strided output lattice (plain conv: 1 phase of 9 taps;
// stride-2 transposed conv: 4 phases of 4/2/2/1 taps).
struct ConvPhase {
    int outH, outW;          // logical output extent (i in [0,outH), j in [0,outW))
    int oy_mul, oy_add, ox_mul, ox_add;   // stored at (i*oy_mul+oy_add, j*ox_mul+ox_add)
    int ntaps;
    int dy[9], dx[9], widx[9];            // input pixel = (i+dy, j+dx); weight tap index
    size_t out_off;                       // float offset of this phase's output plane (phase-major T layout), else 0
};

void sr_fill_tconv_phases(ConvPhase* ph, int Hin, int Win);
void sr_fill_conv3x3_phase(ConvPhase* ph, int H, int W);

// f16x3 implementation (r3d_sr_f16x3.hip)
int sr_prepack_f16x3(int Cin, int Cout, const float* c0_w, const float* c1_w, void* prepacked, hipStream_t st, bool mx);
int sr_block_forward_f16x3(const void* prepacked, const void* styles, int N, int Cin, int Cout, int Hin, int Win, int up,
                           const void* x, int x_format, const float* img, float clamp,
                           void* x_out, int x_out_format, const float* next_scale, size_t next_scale_stride,
                           float* img_out, uint8_t* img_u8, float* x_absmax, void* workspace, size_t workspace_bytes, hipStream_t st, bool mx);

size_t conv_prepacked_bytes_f16x3(int Cin, int Cout, int ksize);
int conv_prepack_f16x3(const float* w, int Cin, int Cout, int ksize, void* prepacked, hipStream_t st);
size_t conv_workspace_bytes_f16x3(int N, int Cin, int H, int W);
// the conv's SPLIT output as one part of a channel concatenation (r3d_conv_forward_cat): channels [chan_off, chan_off + Cout) of C_total, times mask or 1 - mask per pixel
struct ConvCat { const float* mask; int mask_invert; int C_total; int chan_off; };
int conv_forward_f16x3(const void* prepacked, const float* scales, size_t scales_stride, const float* bias,
                       int N, int Cin, int Cout, int H, int W, int ksize,
                       const void* x, int x_format, int act, float slope, float gain, float clamp,
                       void* y, int y_format, const float* next_scale, size_t next_scale_stride, float* y_absmax,
                       void* workspace, hipStream_t st, const ConvCat* cat = nullptr);

int conv_forward_blend_f16x3(const void* prepacked, const float* scales, size_t scales_stride, const float* bias,
                             int N, int Ca, int Cb, int Cout, int H, int W, const float* xa, const float* xb, const float* mask,
                             int act, float slope, float gain, float clamp,
                             void* y, int y_format, const float* next_scale, size_t next_scale_stride, float* y_absmax, hipStream_t st);

int upsample2x_bilinear_f16x3(const float* x_cb8, int N, int C, int H, int W, void* y, int y_format,
                              const float* next_scale, size_t next_scale_stride, hipStream_t st);

int blend_cat_to_split_f16x3(const float* a, int a_format, int Ca, const float* b, int b_format, int Cb, const float* mask,
                             int N, int H, int W, void* y_split, int y_format, const float* next_scale, size_t next_scale_stride, hipStream_t st);

}  // namespace r3d
