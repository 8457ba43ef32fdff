// Declarations shared by the f32 (r3d_sr.hip) and f16x3 (r3d_sr_f16x3.hip) super-resolution kernels.
#pragma once
#include "r3d_common.h"

namespace r3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
static constexpr int BLOCK_M = 128;                    // output channels per conv block
static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct SrStyleLayout { size_t s0, s1, s2, d0, d1, wrgb, b0, b1, brgb, total; };
static __host__ __device__ inline SrStyleLayout sr_style_layout(int Cin, int Cout)
{
    SrStyleLayout L;
    size_t o = 0;
    L.s0 = o; o += Cin;
    L.s1 = o; o += Cout;
    L.s2 = o; o += Cout;
    L.d0 = o; o += Cout;
    L.d1 = o; o += Cout;
    L.wrgb = o; o += (size_t)3 * Cout;
    L.b0 = o; o += Cout;
    L.b1 = o; o += Cout;
    L.brgb = o; o += 4;
    L.total = (o + 3) & ~(size_t)3;
    return L;
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
int sr_prepack_f16x3(int Cin, int Cout, const float* c0_w, const float* c1_w, void* prepacked, hipStream_t st);
int sr_block_forward_f16x3(const void* prepacked, const void* styles, int N, int Cin, int Cout, int Hin, int Win, int up,
                           const void* x, int x_format, const float* img, float clamp,
                           void* x_out, int x_out_format, const float* next_scale, size_t next_scale_stride,
                           float* img_out, void* workspace, size_t workspace_bytes, hipStream_t st);

size_t conv_prepacked_bytes_f16x3(int Cin, int Cout, int ksize);
int conv_prepack_f16x3(const float* w, int Cin, int Cout, int ksize, void* prepacked, hipStream_t st);
size_t conv_workspace_bytes_f16x3(int N, int Cin, int H, int W);
int conv_forward_f16x3(const void* prepacked, int N, int Cin, int Cout, int H, int W, int ksize,
                       const void* x, int x_format, const float* in_scale, size_t in_scale_stride,
                       const float* out_scale, size_t out_scale_stride, const float* bias, size_t bias_stride,
                       int act, float slope, float gain, float clamp,
                       void* y, int y_format, const float* next_scale, size_t next_scale_stride,
                       void* workspace, hipStream_t st);

int upsample2x_bilinear_f16x3(const float* x_cb8, int N, int C, int H, int W, void* y, int y_format,
                              const float* next_scale, size_t next_scale_stride, hipStream_t st);

int blend_cat_to_split_f16x3(const float* a, int a_format, int Ca, const float* b, int b_format, int Cb, const float* mask,
                             int N, int H, int W, void* y_split, hipStream_t st);

}  // namespace r3d
