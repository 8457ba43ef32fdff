// StyleGAN2-style super-resolution SynthesisBlock for gfx950 (MI355X), fp32 in / fp32 accumulate.
//
// Behaviour restated from the upstream repo: modules/eg3ds/models/networks_stylegan2.py:37-94 (modulated_conv2d),
// :286-373 (SynthesisLayer / ToRGBLayer), :429-473 (SynthesisBlock.forward);
// modules/eg3ds/torch_utils/ops/conv2d_resample.py:116-133 (up=2 = conv_transpose2d stride 2 + 4x4 FIR, gain 4);
// ops/upfirdn2d.py:72-116,171-215,317-354 (FIR taps outer([1,3,3,1])/64, upsample2d); ops/bias_act.py:93-122.
//
// MI355X design:
//  * activations live channel-blocked "CB8": [N][C/8][H][W][8] fp32, so a pixel's 8-channel group is one
//    32-byte unit: patch staging, MFMA operand reads (ds_read_b128) and epilogue stores are all 16-byte
//    vectors and a wave's epilogue store covers 1 KB contiguous;
//  * every 3x3 / transposed conv is ONE implicit-GEMM kernel on v_mfma_f32_32x32x2_f32 computed
//    transposed, D[cout][pixel] = sum_tap W_tap[cout][ci] X[ci][pixel+off(tap)]: the 10x18 input patch of a
//    block's 8x16 pixel tile is staged once in LDS and re-read at 9 shifted positions (9x less global
//    traffic than im2col), weights stream from L2 straight into A-operand registers (packed so that a
//    wave's load is 1 KB contiguous), bias + lrelu*sqrt(2) (+clamp) run in the epilogue;
//  * the stride-2 transposed conv runs as 4 output phases with 4/2/2/1 taps (9 tap-GEMMs per 4 output
//    pixels: no multiply-by-zero work), followed by a fused FIR4x4 + bias + lrelu kernel;
//  * modulation/demodulation is a per-forward packing kernel (no host sync, any ws), not a grouped conv.
#include "r3d_sr_common.h"

namespace r3d {


static constexpr int TILE_H = 8, TILE_W = 16;          // output pixels per block
static constexpr int PATCH_H = TILE_H + 2, PATCH_W = TILE_W + 2, PATCH_PIX = PATCH_H * PATCH_W;   // 180
static constexpr int CHUNKS_PER_STAGE = 4;             // 4 x 8 = 32 input channels per LDS stage

// -------------------------------------------------------------------------------------------------
// layout kernels
// -------------------------------------------------------------------------------------------------
__global__ void nchw_to_cb8_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW)
{
    // one thread per (cb, pixel): gathers 8 channels
    const int n = blockIdx.z, cb = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* s = src + ((size_t)n * C + cb * 8) * HW + p;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = s[(size_t)c * HW];
    float4* d = reinterpret_cast<float4*>(dst + (((size_t)n * (C / 8) + cb) * HW + p) * 8);
    d[0] = make_float4(v[0], v[1], v[2], v[3]);
    d[1] = make_float4(v[4], v[5], v[6], v[7]);
}

__global__ void cb8_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW)
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

// -------------------------------------------------------------------------------------------------
// Modulated convolution without per-frame weight rewriting.
//   reference (networks_stylegan2.py:62-70):  y = conv(x, W*s*d),  s = affine(w),  d = rsqrt(sum (W*s)^2 + 1e-8)
//   here:                                     y = d[co] * conv(s[ci]*x, W)
// so the conv weights are re-laid out ONCE (r3d_sr_block_prepack: [tap][ci/8][cout][ci%8], a wave's A-operand
// load is 1 KB contiguous) and each forward only computes the small vectors s (styles), d (demodulation),
// the 3xC modulated toRGB weights and copies the biases (r3d_sr_block_styles).
// styles buffer (floats, per batch item): see SrStyleLayout
// -------------------------------------------------------------------------------------------------
// static re-layout: prepacked = [conv0: 9][Cin/8][Cout][8] ++ [conv1: 9][Cout/8][Cout][8]; one thread per float4
__global__ void sr_prepack_kernel(const float* __restrict__ w, int Ci, int Cout, float* __restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (tap, chunk, cout, half)
    const size_t total = (size_t)9 * (Ci / 8) * Cout * 2;
    if (e >= total) return;
    const int half = e & 1;
    const int co = (e >> 1) % Cout;
    const int chunk = ((e >> 1) / Cout) % (Ci / 8);
    const int tap = (int)((e >> 1) / Cout / (Ci / 8));
    const int ci = chunk * 8 + half * 4;
    const float* src = w + ((size_t)co * Ci + ci) * 9 + tap;
    reinterpret_cast<float4*>(out)[e] = make_float4(src[0], src[9], src[18], src[27]);
}

// grid (ceil(max(Cin,Cout)/4), 3 layers, N), one wave per style row: styles = affine(w)
// (networks_stylegan2.py:326; FullyConnectedLayer :99-131); toRGB styles carry the 1/sqrt(Cin) gain (:366)
__global__ void sr_styles_kernel(const float* __restrict__ ws3, int WD, int Cin, int Cout,
                                 const float* __restrict__ aw0, const float* __restrict__ ab0,
                                 const float* __restrict__ aw1, const float* __restrict__ ab1,
                                 const float* __restrict__ aw2, const float* __restrict__ ab2,
                                 float* __restrict__ styles, size_t stride_n)
{
    const int layer = blockIdx.y, n = blockIdx.z;
    const SrStyleLayout L = sr_style_layout(Cin, Cout);
    const float* aw = layer == 0 ? aw0 : (layer == 1 ? aw1 : aw2);
    const float* ab = layer == 0 ? ab0 : (layer == 1 ? ab1 : ab2);
    const int C = layer == 0 ? Cin : Cout;
    const int lane = threadIdx.x & 63, c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    float* out = styles + n * stride_n + (layer == 0 ? L.s0 : (layer == 1 ? L.s1 : L.s2));
    const float* w = ws3 + ((size_t)n * 3 + layer) * WD;
    const float g = rsqrtf((float)WD);
    const float post = layer == 2 ? rsqrtf((float)Cout) : 1.0f;
    float acc = 0.f;
    for (int j = lane; j < WD; j += 64) acc += w[j] * (aw[(size_t)c * WD + j] * g);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) out[c] = (acc + ab[c]) * post;
}

// grid (Cout + 1, N, 2): blocks [0, Cout) of layer z: one BLOCK per output channel: d = rsqrt(sum_{ci,k} (W*s)^2 + 1e-8)
// (networks_stylegan2.py:65-70); block Cout of layer 0: modulated toRGB weights (:366-368) + bias copies.
// Also per cout, for the fp16 range management of the f16x3 path (r3d_sr_common.h): the bound coefficient
// c[co] = d[co] * sum |W*s|  (|conv output| <= c[co] * max|input|) and the weight-row factor 2^-kw[co].
__global__ void sr_demod_kernel(int Cin, int Cout, const float* __restrict__ w0, const float* __restrict__ w1,
                                const float* __restrict__ wrgb, const float* __restrict__ b0, const float* __restrict__ b1,
                                const float* __restrict__ brgb, float* __restrict__ styles, size_t stride_n)
{
    const int co = blockIdx.x, n = blockIdx.y, layer = blockIdx.z;
    const SrStyleLayout L = sr_style_layout(Cin, Cout);
    float* base = styles + n * stride_n;
    if (co == Cout) {
        if (layer == 0)
            for (int i = threadIdx.x; i < 3 * Cout; i += blockDim.x) {
                base[L.wrgb + i] = wrgb[i] * base[L.s2 + (i % Cout)];
                if (i < Cout) { base[L.b0 + i] = b0[i]; base[L.b1 + i] = b1[i]; }
                if (i < 3) base[L.brgb + i] = brgb[i];
            }
        return;
    }
    const int Ci = layer == 0 ? Cin : Cout;
    const float* W = (layer == 0 ? w0 : w1) + (size_t)co * Ci * 9;
    const float* st = base + (layer == 0 ? L.s0 : L.s1);
    float ss = 0.f, l1 = 0.f, mx = 0.f;
    for (int i = threadIdx.x; i < Ci * 9; i += blockDim.x) {
        const float wv = W[i], v = wv * st[i / 9];
        ss += v * v; l1 += fabsf(v); mx = fmaxf(mx, fabsf(wv));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ss += __shfl_xor(ss, d); l1 += __shfl_xor(l1, d); mx = fmaxf(mx, __shfl_xor(mx, d)); }
    __shared__ float red[3][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ss; red[1][threadIdx.x >> 6] = l1; red[2][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float d = rsqrtf(red[0][0] + red[0][1] + red[0][2] + red[0][3] + 1e-8f);
        base[(layer == 0 ? L.d0 : L.d1) + co] = d;
        base[(layer == 0 ? L.c0 : L.c1) + co] = d * (red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        base[(layer == 0 ? L.wi0 : L.wi1) + co] = pow2f(-weight_row_exp(fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]))));
    }
}

// ---- range fold (r3d_chain_fold): one block per sample walks a chain of layers, propagating a guaranteed bound on max|x| and
// writing each layer's power-of-two folded multipliers (see r3d_sr_common.h).  Runs once per forward in front of the convs.
struct ChainArgs {
    r3d_chain_op ops[R3D_CHAIN_MAX_OPS];
    const float* ext[R3D_CHAIN_MAX_EXT];
    float* zero[R3D_CHAIN_MAX_ZERO];
    int nops, N, nzero, next;
};

__device__ __forceinline__ float block_max(float v, float* red)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void chain_fold_kernel(ChainArgs a)
{
    __shared__ float red[4];
    __shared__ float bounds[R3D_CHAIN_MAX_OPS];
    __shared__ float extv[R3D_CHAIN_MAX_EXT];
    const int n = blockIdx.x, tid = threadIdx.x;
    // external bounds are read BEFORE the zero slots are cleared: a slot may be both (a fold that consumes a measured maximum and
    // re-arms the slot for the next frame's measurement, e.g. the f16mx tail fold)
    if (tid < a.next) extv[tid] = fabsf(a.ext[tid][n]);
    // One load phase (round 6): the chain below is a sequence of DEPENDENT round trips (a layer's bound needs the previous layer's), two per SR layer and one per conv
    // layer, each a cold miss after the convs in between have flushed the L2 -- 10 us for a three-conv + block chain, six folds per torso frame.  Every vector any op
    // will read is touched here first, all loads independent and in flight together: the chain then runs on L2 hits.
    {
        float touch = 0.f;
        for (int k = 0; k < a.nops; ++k) {
            const r3d_chain_op& op = a.ops[k];
            if (op.kind == R3D_CHAIN_SR_BLOCK || op.kind == R3D_CHAIN_SR_BLOCK_TAIL) {
                const SrStyleLayout L = sr_style_layout(op.Cin, op.Cout);
                const float* base = reinterpret_cast<const float*>(op.scales) + (size_t)n * L.total;
                for (int i = tid; i < op.Cin; i += 256) touch += base[L.s0 + i];
                for (int i = tid; i < op.Cout; i += 256)
                    touch += base[L.s1 + i] + base[L.d0 + i] + base[L.d1 + i] + base[L.wi0 + i] + base[L.wi1 + i] + base[L.b0 + i] + base[L.b1 + i] + base[L.c0 + i] + base[L.c1 + i];
            } else {
                const int Ci = (op.Cin + 15) / 16 * 16, Co = (op.Cout + BLOCK_M - 1) / BLOCK_M * BLOCK_M;
                const ConvTail T = conv_tail_layout(Co);
                const float* tail = reinterpret_cast<const float*>(op.prepacked) + (size_t)(op.ksize * op.ksize) * Ci * Co;
                for (int i = tid; i < Co; i += 256) touch += tail[T.winv + i] + tail[T.l1 + i] + (op.bias && i < op.Cout ? op.bias[i] : 0.f);
            }
        }
        if (touch == 1.2345e-33f && a.nzero < 0) extv[0] = touch;     // (never true: keeps the loads)
    }
    __syncthreads();
    if (tid < a.nzero) a.zero[tid][n] = 0.f;                  // absmax slots of the tensors the following kernels measure
    for (int k = 0; k < a.nops; ++k) {
        const r3d_chain_op& op = a.ops[k];
        float B = 0.f;
        const int srcs[2] = {op.src_a, op.src_b};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int s = srcs[j];
            if (s >= 0) B = fmaxf(B, bounds[s]);
            else if (s >= -R3D_CHAIN_MAX_EXT) B = fmaxf(B, extv[-1 - s]);
        }
        float Bout;
        if (op.kind == R3D_CHAIN_SR_BLOCK || op.kind == R3D_CHAIN_SR_BLOCK_TAIL) {
            const SrStyleLayout L = sr_style_layout(op.Cin, op.Cout);
            float* base = reinterpret_cast<float*>(op.scales) + (size_t)n * L.total;
            float Bin = B;
            if (op.kind == R3D_CHAIN_SR_BLOCK_TAIL) {
                // B is a measured max|block input|: tighten the bound after conv0 with it, keep layer 0's multipliers as they are
                float bo = 0.f;
                for (int i = tid; i < op.Cout; i += 256) bo = fmaxf(bo, fabsf(base[L.b0 + i]) + base[L.c0 + i] * B);
                bo = block_max(bo, red) * op.gain;
                if (op.clamp >= 0.f) bo = fminf(bo, op.clamp);
                Bin = fminf(bo, base[L.meta + SR_META_BOUND_MID]);
            }
            for (int layer = (op.kind == R3D_CHAIN_SR_BLOCK_TAIL ? 1 : 0); layer < 2; ++layer) {
                const int Ci = layer ? op.Cout : op.Cin;
                const size_t so = layer ? L.s1 : L.s0, sf = layer ? L.s1f : L.s0f, d_ = layer ? L.d1 : L.d0, df = layer ? L.d1f : L.d0f;
                const size_t c_ = layer ? L.c1 : L.c0, wi = layer ? L.wi1 : L.wi0, b_ = layer ? L.b1 : L.b0;
                float sm = 0.f;
                for (int i = tid; i < Ci; i += 256) sm = fmaxf(sm, fabsf(base[so + i]));
                sm = block_max(sm, red);
                const int e = act_exp(Bin, sm);
                const float up = pow2f(e), dn = pow2f(-e);
                for (int i = tid; i < Ci; i += 256) base[sf + i] = base[so + i] * up;
                float bo = 0.f;
                for (int i = tid; i < op.Cout; i += 256) {
                    base[df + i] = base[d_ + i] * base[wi + i] * dn;
                    bo = fmaxf(bo, fabsf(base[b_ + i]) + base[c_ + i] * Bin);
                }
                bo = block_max(bo, red) * op.gain;
                if (op.clamp >= 0.f) bo = fminf(bo, op.clamp);
                if (tid == 0) {
                    base[L.meta + (layer ? SR_META_E1 : SR_META_E0)] = (float)e;
                    base[L.meta + (layer ? SR_META_SMAX1 : SR_META_SMAX0)] = sm;
                    base[L.meta + (layer ? SR_META_BOUND_OUT : SR_META_BOUND_MID)] = bo;
                    if (!layer) base[L.meta + SR_META_BOUND_IN] = Bin;
                }
                Bin = bo;
            }
            Bout = Bin;
        } else {
            const int Ci = (op.Cin + 15) / 16 * 16, Co = (op.Cout + BLOCK_M - 1) / BLOCK_M * BLOCK_M;
            const ConvScales S = conv_scales_layout(Ci, Co);
            const ConvTail T = conv_tail_layout(Co);
            float* base = reinterpret_cast<float*>(op.scales) + (size_t)n * S.total;
            const float* tail = reinterpret_cast<const float*>(op.prepacked) + (size_t)(op.ksize * op.ksize) * Ci * Co;
            const int e = act_exp(B, 1.0f);
            const float up = pow2f(e), dn = pow2f(-e);
            for (int i = tid; i < Ci; i += 256) base[S.in_vec + i] = up;
            float bo = 0.f;
            for (int i = tid; i < Co; i += 256) {
                base[S.out_vec + i] = tail[T.winv + i] * dn;
                if (i < op.Cout) bo = fmaxf(bo, (op.bias ? fabsf(op.bias[i]) : 0.f) + tail[T.l1 + i] * B);
            }
            bo = block_max(bo, red);
            if (op.act) bo *= op.gain;
            if (op.clamp >= 0.f) bo = fminf(bo, op.clamp);
            if (tid == 0) { base[S.meta + 0] = B; base[S.meta + 1] = bo; base[S.meta + 2] = (float)e; }
            Bout = bo;
        }
        if (tid == 0) bounds[k] = Bout;
        __syncthreads();
    }
}

// ---- per-sample max |x| of an fp32 tensor (the measured bound at an fp32 -> SPLIT conversion).  `out` must be zero on entry;
// `zero_next` (the slot of the NEXT call, ping-pong) is cleared here so that no separate memset is needed.
__global__ void absmax_kernel(const float* __restrict__ x, size_t count, unsigned* __restrict__ out, unsigned* __restrict__ zero_next)
{
    const int n = blockIdx.y;
    if (zero_next && blockIdx.x == 0 && threadIdx.x == 0) zero_next[n] = 0u;
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)n * count);
    const size_t n4 = (count & 3) ? 0 : (count >> 2);                  // vector path only when every sample stays 16-byte aligned
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {                      // four independent 16-byte loads in flight per lane
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        const float ma = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
        const float mb = fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)));
        const float mc = fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w)));
        const float md = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
        m = fmaxf(m, fmaxf(fmaxf(ma, mb), fmaxf(mc, md)));
    }
    for (; i < n4; i += stride) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (!n4) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[(size_t)n * count + i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    // ONE atomic per block: atomics on one word serialise in L2 at ~12 ns each -- round 2's one-per-wave version (8 192 atomics for a 16 MB
    // tensor) took 97 us for a pass that needs 3 us of HBM time
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(&out[n], __float_as_uint(m));           // non-negative floats order like unsigned ints; NaN never wins
    }
}

// -------------------------------------------------------------------------------------------------
// implicit-GEMM conv on f32 MFMA.  A "phase" is a set of taps writing to a strided output lattice.
// -------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* x;  size_t x_stride_n;       // CB8 input  [Cin/8][H][W][8]
    const float* wp;                          // static pre-packed weights [9][Cin/8][Cout][8]
    const float* in_scale;                    // styles  s[ci]  (per batch item, stride vec_stride_n)
    const float* out_scale;                   // demod   d[cout]
    const float* bias;                        // [Cout] or nullptr
    size_t vec_stride_n;
    float* y; size_t y_stride_n;              // CB8 output [Cout/8][OH][OW][8]
    int Cin, Cout, H, W, OH, OW;
    int nphase, act; float clamp;
    ConvPhase ph[4];
};

template <int NTAPS>
__device__ __forceinline__ void conv_block(const ConvArgs& a, const ConvPhase& ph, int n, float* patch)
{
    const int tiles_x = (ph.outW + TILE_W - 1) / TILE_W;
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int i0 = ty * TILE_H, j0 = tx * TILE_W;
    const int m0 = blockIdx.y * BLOCK_M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;          // wave tile: couts [m0+64wm, +64), rows [4wn, 4wn+4)
    const int li = lane & 31, h = lane >> 5;
    const int nchunks = a.Cin >> 3;
    const float* X = a.x + (size_t)n * a.x_stride_n;
    const float4* WP = reinterpret_cast<const float4*>(a.wp);
    const float4* SC = reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.vec_stride_n);

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // B-operand patch offsets (float4 units within one chunk's patch) per N-tile and tap
    int boff[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int row = wn * 4 + nt * 2 + (li >> 4), col = li & 15;
        boff[nt] = ((row + 1) * PATCH_W + (col + 1)) * 2 + h;
    }
    const float4* patch4 = reinterpret_cast<const float4*>(patch);

    for (int c0 = 0; c0 < nchunks; c0 += CHUNKS_PER_STAGE) {
        const int nc = min(CHUNKS_PER_STAGE, nchunks - c0);
        __syncthreads();
        // ---- stage the (TILE+halo) patch of nc channel blocks: zero outside the image ------------
        for (int e = threadIdx.x; e < nc * PATCH_PIX * 2; e += blockDim.x) {
            const int half = e & 1, pp = (e >> 1) % PATCH_PIX, c = (e >> 1) / PATCH_PIX;
            const int py = pp / PATCH_W, px = pp - py * PATCH_W;
            const int iy = i0 + py - 1, ix = j0 + px - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                v = *reinterpret_cast<const float4*>(X + (((size_t)(c0 + c) * a.H + iy) * a.W + ix) * 8 + half * 4);
                const float4 sc = SC[(c0 + c) * 2 + half];          // modulation: x * s[ci]
                v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
            }
            reinterpret_cast<float4*>(patch)[(c * PATCH_PIX + pp) * 2 + half] = v;
        }
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
            // A fragments for every tap of this channel block straight from L2
            float4 af[NTAPS][2];
#pragma unroll
            for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    af[t][mt] = WP[(((size_t)ph.widx[t] * nchunks + (c0 + c)) * a.Cout + (m0 + 64 * wm + 32 * mt + li)) * 2 + h];
            const float4* pc = patch4 + c * PATCH_PIX * 2;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int toff = (ph.dy[t] * PATCH_W + ph.dx[t]) * 2;
                float4 bf[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bf[nt] = pc[boff[nt] + toff];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                reinterpret_cast<const float*>(&af[t][mt])[k], reinterpret_cast<const float*>(&bf[nt])[k],
                                acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias + lrelu(0.2)*sqrt(2) (+clamp)  (bias_act.py:93-122), CB8 store ------------
    float* Y = a.y + (size_t)n * a.y_stride_n;
    const float* B = a.bias ? a.bias + (size_t)n * a.vec_stride_n : nullptr;
    const float* D = a.out_scale + (size_t)n * a.vec_stride_n;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int i = i0 + wn * 4 + nt * 2 + (li >> 4), j = j0 + (li & 15);
        if (i >= ph.outH || j >= ph.outW) continue;
        const int oy = i * ph.oy_mul + ph.oy_add, ox = j * ph.ox_mul + ph.ox_add;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = m0 + 64 * wm + 32 * mt + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[mt][nt][4 * g + r] * D[co + r];      // demodulation
                    if (a.act) {
                        t += B[co + r];
                        t = (t < 0.f ? t * 0.2f : t) * 1.4142135623730951f;
                        if (a.clamp >= 0.f) t = fminf(fmaxf(t, -a.clamp), a.clamp);
                    }
                    v[r] = t;
                }
                *reinterpret_cast<float4*>(Y + ((((size_t)(co >> 3)) * a.OH + oy) * a.OW + ox) * 8 + (co & 7)) =
                    make_float4(v[0], v[1], v[2], v[3]);
            }
    }
}

__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) float patch[CHUNKS_PER_STAGE * PATCH_PIX * 8];
    const int n = blockIdx.z / a.nphase, p = blockIdx.z - n * a.nphase;
    const ConvPhase& ph = a.ph[p];
    const int tiles = ((ph.outW + TILE_W - 1) / TILE_W) * ((ph.outH + TILE_H - 1) / TILE_H);
    if ((int)blockIdx.x >= tiles) return;
    switch (ph.ntaps) {
        case 9: conv_block<9>(a, ph, n, patch); break;
        case 4: conv_block<4>(a, ph, n, patch); break;
        case 2: conv_block<2>(a, ph, n, patch); break;
        default: conv_block<1>(a, ph, n, patch); break;
    }
}

// -------------------------------------------------------------------------------------------------
// FIR 4x4 (outer([1,3,3,1])/64 * gain 4, pad 1) + bias + lrelu*sqrt(2)  on the transposed-conv output
// T [C/8][2H+1][2W+1][8] -> y [C/8][2H][2W][8]      (conv2d_resample.py:130, upfirdn2d.py:171-215)
// One thread: one output pixel x 4 channels.
// -------------------------------------------------------------------------------------------------
__global__ void fir_bias_act_kernel(const float* __restrict__ T, size_t t_stride_n, const float* __restrict__ bias,
                                    size_t bias_stride_n, float* __restrict__ y, size_t y_stride_n,
                                    int C, int OH, int OW, float clamp)
{
    const int n = blockIdx.z;
    const int cb = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;     // (pixel, half)
    if (e >= OH * OW * 2) return;
    const int half = e & 1, p = e >> 1;
    const int oy = p / OW, ox = p - oy * OW;
    const int TH = OH + 1, TW = OW + 1;
    const float* Tn = T + (size_t)n * t_stride_n + (size_t)cb * TH * TW * 8 + half * 4;
    const float f1[4] = {0.25f, 0.75f, 0.75f, 0.25f};        // [1,3,3,1]/8 * 2 per axis
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
        const int ty = oy + aa - 1;
        if (ty < 0 || ty >= TH) continue;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const int tx = ox + bb - 1;
            if (tx < 0 || tx >= TW) continue;
            const float4 v = *reinterpret_cast<const float4*>(Tn + ((size_t)ty * TW + tx) * 8);
            const float w = f1[aa] * f1[bb];
            acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
        }
    }
    const float* b = bias + (size_t)n * bias_stride_n + cb * 8 + half * 4;
    float o[4] = {acc.x + b[0], acc.y + b[1], acc.z + b[2], acc.w + b[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = o[r];
        t = (t < 0.f ? t * 0.2f : t) * 1.4142135623730951f;
        if (clamp >= 0.f) t = fminf(fmaxf(t, -clamp), clamp);
        o[r] = t;
    }
    *reinterpret_cast<float4*>(y + (size_t)n * y_stride_n + (((size_t)cb * OH + oy) * OW + ox) * 8 + half * 4) =
        make_float4(o[0], o[1], o[2], o[3]);
}

// -------------------------------------------------------------------------------------------------
// toRGB (1x1 modulated conv, no demod, linear bias_act; networks_stylegan2.py:365-370) fused with the
// RGB-skip upsample2d (upfirdn2d.py:317-354) and the add (:463-469).
// x CB8 [C/8][H][W][8]; img_prev NCHW [3][H/2][W/2]; img_out NCHW [3][H][W]
// -------------------------------------------------------------------------------------------------
__global__ void torgb_upsample_kernel(const float* __restrict__ x, size_t x_stride_n,
                                      const float* __restrict__ wrgb, const float* __restrict__ brgb, size_t pk_stride_n,
                                      const float* __restrict__ img_prev, float* __restrict__ img_out,
                                      int C, int H, int W, float clamp)
{
    extern __shared__ float sw[];      // [3][C]
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) sw[i] = wrgb[(size_t)n * pk_stride_n + i];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, xx = p - y * W;
    const float* X = x + (size_t)n * x_stride_n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int ncb = C >> 3;
    for (int cb = 0; cb < ncb; ++cb) {
        const float4* s = reinterpret_cast<const float4*>(X + ((size_t)cb * H * W + p) * 8);
        const float4 u = s[0], v = s[1];
        const float xv[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            a0 += xv[c] * sw[cb * 8 + c];
            a1 += xv[c] * sw[C + cb * 8 + c];
            a2 += xv[c] * sw[2 * C + cb * 8 + c];
        }
    }
    const float* br = brgb + (size_t)n * pk_stride_n;
    float o[3] = {a0 + br[0], a1 + br[1], a2 + br[2]};
    if (clamp >= 0.f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = fminf(fmaxf(o[c], -clamp), clamp);
    }
    // upsample2d: zero-insert x2, pad (2,1), FIR [1,3,3,1]^2/64 * 4: per axis even -> (1/4, 3/4) on (k-1, k),
    // odd -> (3/4, 1/4) on (k, k+1) with k = coord>>1
    const int Hh = H >> 1, Wh = W >> 1;
    const int ky = y >> 1, kx = xx >> 1;
    int r0, r1, c0, c1; float wy0, wy1, wx0, wx1;
    if (y & 1) { r0 = ky; r1 = ky + 1; wy0 = 0.75f; wy1 = 0.25f; } else { r0 = ky - 1; r1 = ky; wy0 = 0.25f; wy1 = 0.75f; }
    if (xx & 1) { c0 = kx; c1 = kx + 1; wx0 = 0.75f; wx1 = 0.25f; } else { c0 = kx - 1; c1 = kx; wx0 = 0.25f; wx1 = 0.75f; }
    const bool vr0 = r0 >= 0 && r0 < Hh, vr1 = r1 >= 0 && r1 < Hh, vc0 = c0 >= 0 && c0 < Wh, vc1 = c1 >= 0 && c1 < Wh;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* I = img_prev + ((size_t)n * 3 + c) * Hh * Wh;
        float up = 0.f;
        if (vr0 && vc0) up += I[(size_t)r0 * Wh + c0] * (wy0 * wx0);
        if (vr0 && vc1) up += I[(size_t)r0 * Wh + c1] * (wy0 * wx1);
        if (vr1 && vc0) up += I[(size_t)r1 * Wh + c0] * (wy1 * wx0);
        if (vr1 && vc1) up += I[(size_t)r1 * Wh + c1] * (wy1 * wx1);
        img_out[((size_t)n * 3 + c) * H * W + p] = up + o[c];
    }
}


}  // namespace r3d

using namespace r3d;

extern "C" size_t r3d_sr_block_prepacked_bytes(int Cin, int Cout)
{
    // conv0 (plain layout) + conv1 + conv0 again in the fused up-conv layout (f16x3) + the two per-cout weight-row tails + conv0 with fp8
    // records in the up-conv layout and in the plain layout (R3D_SR_F16MX: SynthesisBlock / SynthesisBlockNoUp on R3D_FMT_SPLIT_MX inputs)
    // + conv1 again as the 12 transformed tap matrices of the Winograd F(2,3) kernel (r3d_sr_wino.h)
    return ((size_t)4 * 9 * Cin * Cout + (size_t)9 * Cout * Cout + 2 * conv_tail_layout(Cout).total + (size_t)12 * Cout * Cout) * sizeof(float);
}

extern "C" size_t r3d_sr_block_styles_bytes(int N, int Cin, int Cout)
{
    return (size_t)N * sr_style_layout(Cin, Cout).total * sizeof(float);
}

extern "C" size_t r3d_sr_block_workspace_bytes(int N, int Cin, int Cout, int Hin, int Win)
{
    const size_t xin = align256((size_t)N * Cin * Hin * Win * 4);
    const size_t T = align256((size_t)N * Cout * 4 * (Hin + 1) * (Win + 1) * 4);
    const size_t y0 = align256((size_t)N * Cout * 4 * Hin * Win * 4);
    const size_t xo = align256((size_t)N * Cout * 4 * Hin * Win * 4);
    const size_t rgbp = align256((size_t)N * (Cout / 64) * 3 * 4 * Hin * Win * 4);     // toRGB partial planes, one per 64 couts
    return xin + T + y0 + xo + rgbp;
}

namespace r3d {
// T[2i+pa][2j+pb] = sum_{ky = pa (mod 2), kx = pb (mod 2)} x[i - ky/2][j - kx/2] w[ky][kx]   (conv_transpose2d stride 2)
void sr_fill_tconv_phases(ConvPhase* ph, int Hin, int Win)
{
    for (int pa = 0; pa < 2; ++pa)
        for (int pb = 0; pb < 2; ++pb) {
            ConvPhase& p = ph[pa * 2 + pb];
            p.outH = Hin + (pa == 0); p.outW = Win + (pb == 0);
            p.oy_mul = 2; p.oy_add = pa; p.ox_mul = 2; p.ox_add = pb; p.out_off = 0;
            p.ntaps = 0;
            for (int ky = pa; ky < 3; ky += 2)
                for (int kx = pb; kx < 3; kx += 2) {
                    p.dy[p.ntaps] = -(ky >> 1); p.dx[p.ntaps] = -(kx >> 1); p.widx[p.ntaps] = ky * 3 + kx;
                    ++p.ntaps;
                }
        }
}
void sr_fill_conv3x3_phase(ConvPhase* ph, int H, int W)
{
    ConvPhase& p = ph[0];
    p.outH = H; p.outW = W; p.oy_mul = 1; p.oy_add = 0; p.ox_mul = 1; p.ox_add = 0; p.ntaps = 9; p.out_off = 0;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) { p.dy[ky * 3 + kx] = ky - 1; p.dx[ky * 3 + kx] = kx - 1; p.widx[ky * 3 + kx] = ky * 3 + kx; }
}
}  // namespace r3d

static int sr_check_dims(const char* what, int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0 || (Cin & 7) || (Cout % BLOCK_M)) {
        set_error("%s: Cin %d must be a multiple of 8 and Cout %d a multiple of %d", what, Cin, Cout, BLOCK_M);
        return R3D_ERR_INVALID_ARG;
    }
    return R3D_OK;
}

extern "C" int r3d_sr_block_prepack(int Cin, int Cout, const float* c0_w, const float* c1_w, void* prepacked,
                                    int precision, r3d_stream_t stream)
{
    if (precision < R3D_SR_F32 || precision > R3D_SR_F16MX) { set_error("sr_block_prepack: unknown precision %d", precision); return R3D_ERR_INVALID_ARG; }
    if (!c0_w || !c1_w || !prepacked) { set_error("sr_block_prepack: NULL pointer"); return R3D_ERR_INVALID_ARG; }
    if (int rc = sr_check_dims("sr_block_prepack", Cin, Cout)) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* out = reinterpret_cast<float*>(prepacked);
    ProfScope ps(R3D_PROF_PACK, st);
    if (precision != R3D_SR_F32) return sr_prepack_f16x3(Cin, Cout, c0_w, c1_w, prepacked, st, precision == R3D_SR_F16MX);
    const size_t n0 = (size_t)9 * (Cin / 8) * Cout * 2, n1 = (size_t)9 * (Cout / 8) * Cout * 2;
    hipLaunchKernelGGL(sr_prepack_kernel, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, st, c0_w, Cin, Cout, out);
    hipLaunchKernelGGL(sr_prepack_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, c1_w, Cout, Cout,
                       out + (size_t)9 * Cin * Cout);
    return check_launch("sr_block_prepack");
}

extern "C" int r3d_sr_block_styles(const float* ws3, int N, int WD, int Cin, int Cout,
                                   const float* c0_w, const float* c0_b, const float* c0_aw, const float* c0_ab,
                                   const float* c1_w, const float* c1_b, const float* c1_aw, const float* c1_ab,
                                   const float* rgb_w, const float* rgb_b, const float* rgb_aw, const float* rgb_ab,
                                   void* styles, r3d_stream_t stream)
{
    if (!ws3 || !c0_w || !c0_b || !c0_aw || !c0_ab || !c1_w || !c1_b || !c1_aw || !c1_ab || !rgb_w || !rgb_b || !rgb_aw || !rgb_ab || !styles) {
        set_error("sr_block_styles: NULL pointer"); return R3D_ERR_INVALID_ARG;
    }
    if (N <= 0 || WD <= 0) { set_error("sr_block_styles: bad shape"); return R3D_ERR_INVALID_ARG; }
    if (int rc = sr_check_dims("sr_block_styles", Cin, Cout)) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* sv = reinterpret_cast<float*>(styles);
    const size_t stride = sr_style_layout(Cin, Cout).total;
    ProfScope ps(R3D_PROF_PACK, st);
    hipLaunchKernelGGL(sr_styles_kernel, dim3(((Cin > Cout ? Cin : Cout) + 3) / 4, 3, N), dim3(256), 0, st, ws3, WD, Cin, Cout, c0_aw, c0_ab, c1_aw, c1_ab, rgb_aw, rgb_ab, sv, stride);
    hipLaunchKernelGGL(sr_demod_kernel, dim3(Cout + 1, N, 2), dim3(256), 0, st, Cin, Cout, c0_w, c1_w, rgb_w, c0_b, c1_b, rgb_b, sv, stride);
    return check_launch("sr_block_styles");
}

extern "C" int r3d_sr_block_forward(const void* prepacked, const void* styles, int N, int Cin, int Cout, int Hin, int Win, int up,
                                    const void* x, int x_format, const float* img, float clamp,
                                    void* x_out, int x_out_format, const float* next_scale, size_t next_scale_stride,
                                    float* img_out, uint8_t* img_u8, float* x_absmax, int precision,
                                    void* workspace, size_t workspace_bytes, r3d_stream_t stream)
{
    if (precision < R3D_SR_F32 || precision > R3D_SR_F16MX) { set_error("sr_block_forward: unknown precision %d", precision); return R3D_ERR_INVALID_ARG; }
    if ((img_u8 || x_absmax) && precision == R3D_SR_F32) { set_error("sr_block_forward: the fused uint8 output / x_absmax need R3D_SR_F16X3"); return R3D_ERR_INVALID_ARG; }
    if (!prepacked || !styles || !x || !img || (!img_out && !img_u8) || N <= 0 || Hin <= 0 || Win <= 0 || (Cin & 15) || (Cout % BLOCK_M) || (up != 0 && up != 1)) {
        set_error("sr_block_forward: bad argument"); return R3D_ERR_INVALID_ARG;
    }
    if (!up && precision == R3D_SR_F32) { set_error("sr_block_forward: up=0 (SynthesisBlockNoUp) needs R3D_SR_F16X3"); return R3D_ERR_INVALID_ARG; }
    {   // the kernels index one sample's activation with 32 bits
        const size_t ohw = (size_t)(up ? 4 : 1) * Hin * Win, cmax = (size_t)(Cin > Cout ? Cin : Cout);
        if (cmax * ohw >= ((size_t)1 << 32)) { set_error("sr_block_forward: activation of %zu elements per sample exceeds the 32-bit index range", cmax * ohw); return R3D_ERR_INVALID_ARG; }
    }
    if (!workspace || workspace_bytes < r3d_sr_block_workspace_bytes(N, Cin, Cout, Hin, Win)) {
        set_error("sr_block_forward: workspace too small"); return R3D_ERR_WORKSPACE;
    }
    if (!x_out) x_out_format = R3D_FMT_NONE;
    const bool f16 = precision != R3D_SR_F32;
    const bool mxp = precision == R3D_SR_F16MX;
    if (x_format < R3D_FMT_NCHW || x_format > R3D_FMT_SPLIT_MX || x_out_format < R3D_FMT_NONE || x_out_format > R3D_FMT_SPLIT_MX ||
        (!f16 && (x_format >= R3D_FMT_SPLIT || x_out_format >= R3D_FMT_SPLIT)) ||
        ((x_format == R3D_FMT_SPLIT_MX || x_out_format == R3D_FMT_SPLIT_MX) && !mxp) ||
        (x_out_format >= R3D_FMT_SPLIT && !next_scale)) {
        set_error("sr_block_forward: unsupported activation format (x %d, x_out %d, precision %d)", x_format, x_out_format, precision);
        return R3D_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    if (f16)
        return sr_block_forward_f16x3(prepacked, styles, N, Cin, Cout, Hin, Win, up, x, x_format, img, clamp, x_out, x_out_format,
                                      next_scale, next_scale_stride, img_out, img_u8, x_absmax, workspace, workspace_bytes, st, precision == R3D_SR_F16MX);

    const SrStyleLayout L = sr_style_layout(Cin, Cout);
    const float* pk = reinterpret_cast<const float*>(styles);
    const float* wpk = reinterpret_cast<const float*>(prepacked);
    const int OH = 2 * Hin, OW = 2 * Win, TH = OH + 1, TW = OW + 1;
    char* wsb = reinterpret_cast<char*>(workspace);
    float* xin = reinterpret_cast<float*>(wsb); wsb += align256((size_t)N * Cin * Hin * Win * 4);
    float* T = reinterpret_cast<float*>(wsb);   wsb += align256((size_t)N * Cout * 4 * (Hin + 1) * (Win + 1) * 4);
    float* y0 = reinterpret_cast<float*>(wsb);  wsb += align256((size_t)N * Cout * OH * OW * 4);
    float* xo = reinterpret_cast<float*>(wsb);

    const float* xcb = reinterpret_cast<const float*>(x);
    if (x_format == R3D_FMT_NCHW) {
        ProfScope ps(R3D_PROF_LAYOUT, st);
        hipLaunchKernelGGL(nchw_to_cb8_kernel, dim3((Hin * Win + 255) / 256, Cin / 8, N), dim3(256), 0, st, xcb, xin, Cin, Hin * Win);
        xcb = xin;
    }
    // ---- conv0: transposed conv stride 2 as 4 phases -> T ------------------------------------------------
    {
        ConvArgs a;
        a.x = xcb; a.x_stride_n = (size_t)Cin * Hin * Win;
        a.wp = wpk; a.in_scale = pk + L.s0; a.out_scale = pk + L.d0; a.vec_stride_n = L.total;
        a.bias = nullptr;
        a.y = T; a.y_stride_n = (size_t)Cout * TH * TW;
        a.Cin = Cin; a.Cout = Cout; a.H = Hin; a.W = Win; a.OH = TH; a.OW = TW;
        a.nphase = 4; a.act = 0; a.clamp = -1.f;
        sr_fill_tconv_phases(a.ph, Hin, Win);
        int maxtiles = 0;
        for (int p = 0; p < 4; ++p) {
            const int tiles = ((a.ph[p].outW + TILE_W - 1) / TILE_W) * ((a.ph[p].outH + TILE_H - 1) / TILE_H);
            if (tiles > maxtiles) maxtiles = tiles;
        }
        ProfScope ps(R3D_PROF_CONV, st);
        hipLaunchKernelGGL(conv_mfma_kernel, dim3(maxtiles, Cout / BLOCK_M, N * 4), dim3(256), 0, st, a);
    }
    {
    ProfScope ps(R3D_PROF_UPCONV, st);
    hipLaunchKernelGGL(fir_bias_act_kernel, dim3((OH * OW * 2 + 255) / 256, Cout / 8, N), dim3(256), 0, st,
                       T, (size_t)Cout * TH * TW, pk + L.b0, L.total, y0, (size_t)Cout * OH * OW, Cout, OH, OW, clamp);
    }
    // ---- conv1: 3x3 pad 1 --------------------------------------------------------------------------------
    float* xo_cb = (x_out_format == R3D_FMT_CB8) ? reinterpret_cast<float*>(x_out) : xo;
    {
        ConvArgs a;
        a.x = y0; a.x_stride_n = (size_t)Cout * OH * OW;
        a.wp = wpk + (size_t)9 * Cin * Cout; a.in_scale = pk + L.s1; a.out_scale = pk + L.d1; a.vec_stride_n = L.total;
        a.bias = pk + L.b1;
        a.y = xo_cb; a.y_stride_n = (size_t)Cout * OH * OW;
        a.Cin = Cout; a.Cout = Cout; a.H = OH; a.W = OW; a.OH = OH; a.OW = OW;
        a.nphase = 1; a.act = 1; a.clamp = clamp;
        sr_fill_conv3x3_phase(a.ph, OH, OW);
        const int tiles = ((OW + TILE_W - 1) / TILE_W) * ((OH + TILE_H - 1) / TILE_H);
        ProfScope ps(R3D_PROF_CONV, st);
        hipLaunchKernelGGL(conv_mfma_kernel, dim3(tiles, Cout / BLOCK_M, N), dim3(256), 0, st, a);
    }
    // ---- toRGB + skip upsample ---------------------------------------------------------------------------
    {
    ProfScope ps(R3D_PROF_TORGB, st);
    hipLaunchKernelGGL(torgb_upsample_kernel, dim3((OH * OW + 255) / 256, N), dim3(256), 3 * Cout * sizeof(float), st,
                       xo_cb, (size_t)Cout * OH * OW, pk + L.wrgb, pk + L.brgb, L.total, img, img_out, Cout, OH, OW, clamp);
    }
    if (x_out_format == R3D_FMT_NCHW) {
        ProfScope ps(R3D_PROF_LAYOUT, st);
        hipLaunchKernelGGL(cb8_to_nchw_kernel, dim3((OH * OW + 255) / 256, Cout / 8, N), dim3(256), 0, st, xo_cb,
                           reinterpret_cast<float*>(x_out), Cout, OH * OW);
    }
    return check_launch("sr_block_forward");
}

// ---- generic convolution layer (see include/r3d_hip.h) -----------------------------------------------------------
extern "C" size_t r3d_conv_prepacked_bytes(int Cin, int Cout, int ksize) { return r3d::conv_prepacked_bytes_f16x3(Cin, Cout, ksize); }
extern "C" size_t r3d_conv_workspace_bytes(int N, int Cin, int H, int W) { return r3d::conv_workspace_bytes_f16x3(N, Cin, H, W); }
extern "C" size_t r3d_conv_scales_bytes(int N, int Cin, int Cout)
{
    return (size_t)N * conv_scales_layout((Cin + 15) / 16 * 16, (Cout + BLOCK_M - 1) / BLOCK_M * BLOCK_M).total * sizeof(float);
}
extern "C" size_t r3d_conv_scales_bound_offset(int Cin, int Cout)
{
    return conv_scales_layout((Cin + 15) / 16 * 16, (Cout + BLOCK_M - 1) / BLOCK_M * BLOCK_M).meta + 1;
}
extern "C" size_t r3d_sr_block_bound_offset(int Cin, int Cout) { return sr_style_layout(Cin, Cout).meta + SR_META_BOUND_OUT; }

extern "C" int r3d_conv_prepack(const float* weight, int Cin, int Cout, int ksize, void* prepacked, r3d_stream_t stream)
{
    using namespace r3d;
    if (!weight || !prepacked || Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3)) { set_error("conv_prepack: bad argument"); return R3D_ERR_INVALID_ARG; }
    return conv_prepack_f16x3(weight, Cin, Cout, ksize, prepacked, (hipStream_t)stream);
}

extern "C" int r3d_chain_fold(const r3d_chain_op* ops, int nops, int N, const float* const* ext_bounds, int n_ext,
                              float* const* zero_slots, int n_zero, r3d_stream_t stream)
{
    using namespace r3d;
    if (!ops || nops <= 0 || nops > R3D_CHAIN_MAX_OPS || N <= 0 || n_ext < 0 || n_ext > R3D_CHAIN_MAX_EXT || (n_ext && !ext_bounds) ||
        n_zero < 0 || n_zero > R3D_CHAIN_MAX_ZERO || (n_zero && !zero_slots)) {
        set_error("chain_fold: bad argument (1..%d ops, 0..%d external bounds, 0..%d zero slots)", R3D_CHAIN_MAX_OPS, R3D_CHAIN_MAX_EXT, R3D_CHAIN_MAX_ZERO);
        return R3D_ERR_INVALID_ARG;
    }
    ChainArgs a = {};
    a.nops = nops; a.N = N; a.nzero = n_zero; a.next = n_ext;
    for (int j = 0; j < n_zero; ++j) {
        if (!zero_slots[j]) { set_error("chain_fold: zero slot %d is NULL", j); return R3D_ERR_INVALID_ARG; }
        a.zero[j] = zero_slots[j];
    }
    for (int j = 0; j < n_ext; ++j) {
        if (!ext_bounds[j]) { set_error("chain_fold: external bound %d is NULL", j); return R3D_ERR_INVALID_ARG; }
        a.ext[j] = ext_bounds[j];
    }
    for (int k = 0; k < nops; ++k) {
        const r3d_chain_op& op = ops[k];
        const int srcs[2] = {op.src_a, op.src_b};
        for (int j = 0; j < 2; ++j) {
            const int sidx = srcs[j];
            const bool ok = sidx == R3D_CHAIN_SRC_NONE || (sidx >= 0 && sidx < k) || (sidx < 0 && sidx >= -n_ext);
            if (!ok) { set_error("chain_fold: op %d reads bound source %d (must be an earlier op or an external bound)", k, sidx); return R3D_ERR_INVALID_ARG; }
        }
        if (op.src_a == R3D_CHAIN_SRC_NONE && op.src_b == R3D_CHAIN_SRC_NONE) { set_error("chain_fold: op %d has no input bound", k); return R3D_ERR_INVALID_ARG; }
        if ((op.kind != R3D_CHAIN_SR_BLOCK && op.kind != R3D_CHAIN_CONV && op.kind != R3D_CHAIN_SR_BLOCK_TAIL) || !op.scales || op.Cin <= 0 || op.Cout <= 0 ||
            (op.kind == R3D_CHAIN_CONV && (!op.prepacked || (op.ksize != 1 && op.ksize != 3)))) {
            set_error("chain_fold: op %d is malformed", k); return R3D_ERR_INVALID_ARG;
        }
        a.ops[k] = op;
        if (a.ops[k].src_a == R3D_CHAIN_SRC_NONE) a.ops[k].src_a = -R3D_CHAIN_MAX_EXT - 1;     // "no source" inside the kernel
        if (a.ops[k].src_b == R3D_CHAIN_SRC_NONE) a.ops[k].src_b = -R3D_CHAIN_MAX_EXT - 1;
    }
    ProfScope ps(R3D_PROF_PACK, (hipStream_t)stream);
    hipLaunchKernelGGL(chain_fold_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("chain_fold");
}

extern "C" int r3d_absmax(const float* x, size_t count_per_sample, int N, float* out, float* zero_next, r3d_stream_t stream)
{
    using namespace r3d;
    if (!x || !out || N <= 0 || count_per_sample == 0) { set_error("absmax: bad argument"); return R3D_ERR_INVALID_ARG; }
    size_t blocks = (count_per_sample / 4 + 255) / 256;
    if (blocks > 512) blocks = 512;                     // 2 blocks per CU, one atomic each; each lane streams 4 x 16 B per iteration
    if (blocks < 1) blocks = 1;
    ProfScope ps(R3D_PROF_LAYOUT, (hipStream_t)stream);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, x, count_per_sample,
                       reinterpret_cast<unsigned*>(out), reinterpret_cast<unsigned*>(zero_next));
    return check_launch("absmax");
}

extern "C" int r3d_conv_forward(const void* prepacked, const void* scales, const float* bias,
                                int N, int Cin, int Cout, int H, int W, int ksize,
                                const void* x, int x_format, int act, float act_slope, float act_gain, float clamp,
                                void* y, int y_format, const float* next_scale, size_t next_scale_stride, float* y_absmax,
                                void* workspace, size_t workspace_bytes, r3d_stream_t stream)
{
    using namespace r3d;
    if (!prepacked || !scales || !x || !y || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (ksize != 1 && ksize != 3)) {
        set_error("conv_forward: bad argument"); return R3D_ERR_INVALID_ARG;
    }
    if (x_format < R3D_FMT_NCHW || x_format > R3D_FMT_SPLIT_MX || y_format < R3D_FMT_NCHW || y_format > R3D_FMT_SPLIT_MX ||
        (y_format == R3D_FMT_SPLIT_MX && (Cout & 15)) || (x_format == R3D_FMT_SPLIT_MX && (ksize != 3 || (Cin & 15)))) {           // (SPLIT_MX out: fp8 records for an f16mx SR block that consumes y; 16-channel groups)
        set_error("conv_forward: unsupported activation format (x %d, y %d)", x_format, y_format); return R3D_ERR_INVALID_ARG;
    }
    if (Cout & 3) { set_error("conv_forward: Cout = %d must be a multiple of 4", Cout); return R3D_ERR_INVALID_ARG; }
    {   // the kernels index one sample's activation with 32 bits (Cout padded to the 128-cout block)
        const size_t cmax = (size_t)((Cin > Cout ? Cin : Cout) + BLOCK_M);
        if (cmax * H * W >= ((size_t)1 << 32)) { set_error("conv_forward: activation of %zu elements per sample exceeds the 32-bit index range", cmax * H * W); return R3D_ERR_INVALID_ARG; }
    }
    if ((x_format != R3D_FMT_NCHW && (Cin & 15)) || (y_format != R3D_FMT_NCHW && (Cout & 7))) {
        set_error("conv_forward: blocked formats need Cin %% 16 == 0 and Cout %% 8 == 0 (Cin %d, Cout %d)", Cin, Cout); return R3D_ERR_INVALID_ARG;
    }
    if (x_format < R3D_FMT_SPLIT && (!workspace || workspace_bytes < r3d_conv_workspace_bytes(N, Cin, H, W))) {
        set_error("conv_forward: workspace too small"); return R3D_ERR_WORKSPACE;
    }
    const size_t stride = conv_scales_layout((Cin + 15) / 16 * 16, (Cout + BLOCK_M - 1) / BLOCK_M * BLOCK_M).total;
    return conv_forward_f16x3(prepacked, reinterpret_cast<const float*>(scales), stride, bias, N, Cin, Cout, H, W, ksize, x, x_format,
                              act, act_slope, act_gain, clamp, y, y_format, next_scale, next_scale_stride, y_absmax,
                              workspace, (hipStream_t)stream);
}

extern "C" int r3d_conv_forward_cat(const void* prepacked, const void* scales, const float* bias,
                                    int N, int Cin, int Cout, int H, int W, int ksize,
                                    const void* x, int x_format, int act, float act_slope, float act_gain, float clamp,
                                    void* y_cat, int y_format, int C_total, int chan_off, const float* mask, int mask_invert,
                                    const float* next_scale, size_t next_scale_stride,
                                    void* workspace, size_t workspace_bytes, r3d_stream_t stream)
{
    using namespace r3d;
    if (!prepacked || !scales || !x || !y_cat || !mask || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) {
        set_error("conv_forward_cat: bad argument"); return R3D_ERR_INVALID_ARG;
    }
    if (ksize != 1) { set_error("conv_forward_cat: ksize %d: only the 1x1 conv kernel carries the concatenation epilogue", ksize); return R3D_ERR_INVALID_ARG; }
    if (y_format != R3D_FMT_SPLIT && y_format != R3D_FMT_SPLIT_MX) { set_error("conv_forward_cat: y_format %d must be SPLIT or SPLIT_MX", y_format); return R3D_ERR_INVALID_ARG; }
    if ((Cout & 15) || (chan_off & 15) || (C_total & 15) || chan_off < 0 || chan_off + Cout > C_total) {
        set_error("conv_forward_cat: Cout %d, chan_off %d, C_total %d must be multiples of 16 with chan_off + Cout <= C_total", Cout, chan_off, C_total); return R3D_ERR_INVALID_ARG;
    }
    if (x_format < R3D_FMT_NCHW || x_format > R3D_FMT_SPLIT_MX || (x_format == R3D_FMT_SPLIT_MX && (ksize != 3 || (Cin & 15))) || (x_format != R3D_FMT_NCHW && (Cin & 15))) {
        set_error("conv_forward_cat: unsupported input (format %d, Cin %d)", x_format, Cin); return R3D_ERR_INVALID_ARG;
    }
    {
        const size_t cmax = (size_t)((Cin > C_total ? Cin : C_total) + BLOCK_M);
        if (cmax * H * W >= ((size_t)1 << 32)) { set_error("conv_forward_cat: activation of %zu elements per sample exceeds the 32-bit index range", cmax * H * W); return R3D_ERR_INVALID_ARG; }
    }
    if (x_format < R3D_FMT_SPLIT && (!workspace || workspace_bytes < r3d_conv_workspace_bytes(N, Cin, H, W))) {
        set_error("conv_forward_cat: workspace too small"); return R3D_ERR_WORKSPACE;
    }
    const size_t stride = conv_scales_layout((Cin + 15) / 16 * 16, (Cout + BLOCK_M - 1) / BLOCK_M * BLOCK_M).total;
    const ConvCat cat = {mask, mask_invert ? 1 : 0, C_total, chan_off};
    return conv_forward_f16x3(prepacked, reinterpret_cast<const float*>(scales), stride, bias, N, Cin, Cout, H, W, ksize, x, x_format,
                              act, act_slope, act_gain, clamp, y_cat, y_format, next_scale, next_scale_stride, nullptr,
                              workspace, (hipStream_t)stream, &cat);
}

extern "C" int r3d_conv_forward_blend(const void* prepacked, const void* scales, const float* bias,
                                      int N, int Ca, int Cb, int Cout, int H, int W,
                                      const float* a, const float* b, const float* mask,
                                      int act, float act_slope, float act_gain, float clamp,
                                      void* y, int y_format, const float* next_scale, size_t next_scale_stride, float* y_absmax, r3d_stream_t stream)
{
    using namespace r3d;
    if (!prepacked || !scales || !a || !b || !mask || !y || N <= 0 || Ca <= 0 || Cb <= 0 || Cout <= 0 || H <= 0 || W <= 0) {
        set_error("conv_forward_blend: bad argument"); return R3D_ERR_INVALID_ARG;
    }
    if ((Ca & 7) || (Cb & 7) || ((Ca + Cb) & 63)) {
        set_error("conv_forward_blend: Ca %d and Cb %d must be multiples of 8 and their sum a multiple of 64", Ca, Cb); return R3D_ERR_INVALID_ARG;
    }
    if (y_format < R3D_FMT_NCHW || y_format > R3D_FMT_SPLIT_MX || (y_format == R3D_FMT_SPLIT_MX && (Cout & 15)) || (y_format != R3D_FMT_NCHW && (Cout & 7)) || (Cout & 3)) {
        set_error("conv_forward_blend: unsupported output (format %d, Cout %d)", y_format, Cout); return R3D_ERR_INVALID_ARG;
    }
    {
        const size_t cmax = (size_t)((Ca + Cb > Cout ? Ca + Cb : Cout) + BLOCK_M);
        if (cmax * H * W >= ((size_t)1 << 32)) { set_error("conv_forward_blend: activation of %zu elements per sample exceeds the 32-bit index range", cmax * H * W); return R3D_ERR_INVALID_ARG; }
    }
    const size_t stride = conv_scales_layout(Ca + Cb, (Cout + BLOCK_M - 1) / BLOCK_M * BLOCK_M).total;
    return conv_forward_blend_f16x3(prepacked, reinterpret_cast<const float*>(scales), stride, bias, N, Ca, Cb, Cout, H, W, a, b, mask,
                                    act, act_slope, act_gain, clamp, y, y_format, next_scale, next_scale_stride, y_absmax, (hipStream_t)stream);
}

extern "C" int r3d_upsample2x_bilinear(const float* x_cb8, int N, int C, int H, int W, void* y, int y_format,
                                       const float* next_scale, size_t next_scale_stride, r3d_stream_t stream)
{
    using namespace r3d;
    if (!x_cb8 || !y || N <= 0 || C <= 0 || (C & 7) || H <= 0 || W <= 0) { set_error("upsample2x_bilinear: bad argument (C %d must be a multiple of 8)", C); return R3D_ERR_INVALID_ARG; }
    if (y_format != R3D_FMT_CB8 && y_format != R3D_FMT_SPLIT && y_format != R3D_FMT_SPLIT_MX) { set_error("upsample2x_bilinear: y_format %d must be CB8, SPLIT or SPLIT_MX", y_format); return R3D_ERR_INVALID_ARG; }
    if (y_format == R3D_FMT_SPLIT_MX && (C & 15)) { set_error("upsample2x_bilinear: SPLIT_MX needs C %% 16 == 0 (C %d)", C); return R3D_ERR_INVALID_ARG; }
    return upsample2x_bilinear_f16x3(x_cb8, N, C, H, W, y, y_format, next_scale, next_scale_stride, (hipStream_t)stream);
}

extern "C" int r3d_blend_cat_to_split(const float* a, int a_format, int Ca, const float* b, int b_format, int Cb, const float* mask,
                                      int N, int H, int W, void* y_split, int y_format, const float* next_scale, size_t next_scale_stride,
                                      r3d_stream_t stream)
{
    if (y_format != R3D_FMT_SPLIT && y_format != R3D_FMT_SPLIT_MX) { set_error("blend_cat_to_split: y_format %d must be SPLIT or SPLIT_MX", y_format); return R3D_ERR_INVALID_ARG; }
    using namespace r3d;
    if (!a || !mask || !y_split || N <= 0 || H <= 0 || W <= 0 || Ca <= 0 || Cb <= 0 || (Ca & 7) || (Cb & 7) || ((Ca + Cb) & 15) || (!b && (Ca & 15))) {
        set_error("blend_cat_to_split: bad argument (Ca %d, Cb %d: multiples of 8, sum a multiple of 16; b = NULL needs Ca %% 16 == 0)", Ca, Cb); return R3D_ERR_INVALID_ARG;
    }
    if ((a_format != R3D_FMT_NCHW && a_format != R3D_FMT_CB8) || (b && b_format != R3D_FMT_NCHW && b_format != R3D_FMT_CB8)) {
        set_error("blend_cat_to_split: inputs must be NCHW or CB8 (a %d, b %d)", a_format, b_format); return R3D_ERR_INVALID_ARG;
    }
    return blend_cat_to_split_f16x3(a, a_format, Ca, b, b_format, Cb, mask, N, H, W, y_split, y_format, next_scale, next_scale_stride, (hipStream_t)stream);
}
