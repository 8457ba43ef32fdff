// Winograd F(2,3) along x on the f16mx MFMA mix of the plain 3x3 convs: what the TRANSFORMED mix sustains on this board.
// Per pair of output columns the direct conv spends 18 tap-products, F(2,3) spends 12 (3 kernel rows x 4 transform positions): 1.5x fewer matrix
// passes for the same outputs, at the price of (i) the input transform  v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3  in fp32 on the
// LDS-resident patch BEFORE the hi / lo split (2 V elements per input element; shared by the block's waves: each wave transforms 1/8 of the tile),
// (ii) twice the accumulators per output (4 positions per column pair), (iii) the output transform in the epilogue (not modelled: 3 adds per output).
//   W0  the conv's LDS-fed mix as it is (power_ceiling_probe L1): 4 waves/SIMD, 64 accumulators per wave
//   W1  W0 + the transform side load per sub-stage: 4 ds_read_b128 of patch columns, fp16 -> fp32, add / sub, re-split into hi + e5m2 records
//       (~36 VALU), 2 ds_write_b128 of V, one s_barrier per stage (2 sub-stages) -- same 4 waves/SIMD x 64 accumulators (half-size N tiles)
//   W2  W1 at 2 waves/SIMD x 128 accumulators per wave (the full-size N tile with the doubled accumulators): half the A reads per MFMA
// "effective" = algorithmic rate of the transformed mix x 1.5 = what the direct conv would have to run at to match.
// build: hipcc -O3 --offload-arch=gfx950 -o winograd_mix_probe winograd_mix_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_clk[2];

// finite operands whatever the bits: f16 pairs with the exponent field forced to [8, 15] (|x| in 2^-7 .. 2), fp8 e4m3 bytes with exponent bits forced away from NaN
__device__ __forceinline__ unsigned f16pair(unsigned r) { return (r & 0x9fff9fffu) | 0x20002000u; }
__device__ __forceinline__ unsigned fp8quad(unsigned r) { return (r & 0xbfbfbfbfu) | 0x20202020u; }


__device__ __forceinline__ unsigned pk_bf8(float a, float b, float c, float d) {
    int v = 0; v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, v, false); v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true); return (unsigned)v;
}
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// the transform of one lane's share: two input columns (hi + lo, 8 channels = one uint4 each) -> two V elements (hi words + e5m2 records)
__device__ __forceinline__ void transform_pair(const u4 dh0, const u4 dl0, const u4 dh1, const u4 dl1, u4& vh, u4& vr)
{
    const h8 h0 = *(const h8*)&dh0, l0 = *(const h8*)&dl0, h1 = *(const h8*)&dh1, l1 = *(const h8*)&dl1;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (float)h0[j] + (float)l0[j]; b[j] = (float)h1[j] + (float)l1[j]; }
    h8 oh; float lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = j < 4 ? a[j] - b[j] : a[j] + b[j];               // half the channels through each of the two forms (d0 - d2 / d1 + d2)
        asm("" : "+v"(v));
        const _Float16 hh = (_Float16)v; oh[j] = hh; lo[j] = (v - (float)hh) * 2048.0f;
    }
    vh = *(const u4*)&oh;
    float hf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) hf[j] = (float)oh[j];
    vr = (u4){pk_bf8(hf[0], hf[1], hf[2], hf[3]), pk_bf8(hf[4], hf[5], hf[6], hf[7]), pk_bf8(lo[0], lo[1], lo[2], lo[3]), pk_bf8(lo[4], lo[5], lo[6], lo[7])};
}

template <int VAR, int TILES>
__global__ __launch_bounds__(512, VAR == 2 ? 2 : 4) void wino_kernel(float* out, int iters, unsigned seed)
{
    __shared__ u4 lds[2048 + 1024];                                 // 32 KB of random operands + 16 KB the transform writes V into
    const int lane = threadIdx.x & 63;
    unsigned s = seed ^ (blockIdx.x * 512u + threadIdx.x) * 2654435761u;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    for (int i = threadIdx.x; i < 3072; i += 512) lds[i] = (u4){f16pair(rnd()), f16pair(rnd()), fp8quad(rnd()), fp8quad(rnd())};
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = clock64(); r0 = wall_clock64(); }
    f16v c[TILES];
    for (int t = 0; t < TILES; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int base = ((it * 4 + u) * 67 + lane) & 2047;
            if (VAR >= 1) {                                         // the transform side load of this sub-stage
                const u4 dh0 = lds[(base + 5) & 2047], dl0 = lds[(base + 133) & 2047], dh1 = lds[(base + 261) & 2047], dl1 = lds[(base + 389) & 2047];
                u4 vh, vr;
                transform_pair(dh0, dl0, dh1, dl1, vh, vr);
                lds[2048 + ((threadIdx.x + 512 * (u & 1)) & 1023)] = (u4){f16pair(vh[0]), f16pair(vh[1]), f16pair(vh[2]), f16pair(vh[3])};
                lds[2048 + ((threadIdx.x + 512 * (u & 1) + 256) & 1023)] = (u4){f16pair(vr[0]), f16pair(vr[1]), fp8quad(vr[2]), fp8quad(vr[3])};
                if (u & 1) __syncthreads();                         // the stage's V is complete before the next stage's MFMAs read it
            }
            constexpr int NF = TILES == 4 ? 16 : 24;                // ds_read_b128 per sub-stage: 4 tiles -> 2 A x 2 B x 4; 8 tiles -> 2 A x 4 B x 4
            u4 f[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) f[j] = lds[(VAR >= 1 && (j & 2)) ? 2048 + ((base + 128 * j) & 1023) : ((base + 128 * j) & 2047)];
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const int ai = t & 1, bi = t >> 1, nb = TILES / 2;
                const u4 A0 = f[ai], B0 = f[2 + bi], A1 = f[2 + nb + ai], B1 = f[4 + nb + bi];
                c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&A0, *(const h8*)&B0, c[t], 0, 0, 0);
                c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&A1, *(const h8*)&B1, c[t], 0, 0, 0);
                const u4 q0 = f[4 + 2 * nb + ai], q1 = f[6 + 2 * nb + ai], p0 = f[8 + 2 * nb + bi], p1 = f[8 + 3 * nb + bi];
                const i8v A8 = {(int)fp8quad(q0[0]), (int)fp8quad(q0[1]), (int)q0[2], (int)q0[3], (int)fp8quad(q1[0]), (int)fp8quad(q1[1]), (int)q1[2], (int)q1[3]};
                const i8v B8 = {(int)fp8quad(p0[0]), (int)fp8quad(p0[1]), (int)p0[2], (int)p0[3], (int)fp8quad(p1[0]), (int)fp8quad(p1[1]), (int)p1[2], (int)p1[3]};
                c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8, B8, c[t], 0, 1, 0, 127, 0, 119);
            }
        }
    }
    float acc = 0;
    for (int t = 0; t < TILES; ++t) for (int r = 0; r < 16; ++r) acc += c[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&g_clk[0], clock64() - c0); atomicAdd(&g_clk[1], wall_clock64() - r0); }
}

struct Smi { std::atomic<bool> stop{false}; std::vector<double> w, mhz; };
static void poll(Smi* s)
{
    while (!s->stop.load()) {
        FILE* p = popen("rocm-smi --showpower --showclocks 2>/dev/null", "r");
        if (!p) return;
        char line[512]; double w = -1, m = -1;
        while (fgets(line, sizeof(line), p)) {
            const char* q;
            if ((q = strstr(line, "Power (W):"))) w = atof(q + 10);
            if ((q = strstr(line, "sclk clock level:")) && (q = strchr(q, '('))) m = atof(q + 1);
        }
        pclose(p);
        if (w > 0 && m > 0) { s->w.push_back(w); s->mhz.push_back(m); }
    }
}

template <int VAR, int TILES> void run(const char* name, float* d, double seconds)
{
    const int iters = TILES == 4 ? 1500 : 750, blocks = TILES == 4 ? 512 : 256, threads = 512;
    unsigned long long z[2] = {0, 0};
    wino_kernel<VAR, TILES><<<blocks, threads>>>(d, 10, 1u); hipDeviceSynchronize();
    hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z));
    Smi smi; std::thread th(poll, &smi);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    double ms_sum = 0; int n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int k = 0; k < 8; ++k) wino_kernel<VAR, TILES><<<blocks, threads>>>(d, iters, 7u + n * 8 + k);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms_sum += ms; n += 8;
    }
    smi.stop.store(true); th.join();
    unsigned long long c[2]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c));
    const double ms = ms_sum / n;
    const double steps = (double)blocks * (threads / 64) * iters * 4 * TILES;       // (2 taps x 16 channels x one 32x32 tile) units
    const double alg = steps * 2.0 * 32 * 32 * 32;                           // algorithmic FLOPs (32 MAC-channels per unit and output)
    double w = 0, m = 0; size_t k0 = smi.w.size() > 2 ? 1 : 0;
    for (size_t i = k0; i < smi.w.size(); ++i) { w += smi.w[i]; m += smi.mhz[i]; }
    const size_t ns = smi.w.size() - k0;
    printf("%-72s %.3f ms/launch  %5.0f TFLOP/s of mix (x1.5 = %5.0f effective, %.3f of 2500)  waves at %.3f GHz  socket %4.0f W  sclk %4.0f MHz (%zu samples)\n", name, ms,
           alg / (ms * 1e-3) / 1e12, (VAR ? 1.5 : 1.0) * alg / (ms * 1e-3) / 1e12, (VAR ? 1.5 : 1.0) * alg / (ms * 1e-3) / 1e12 / 2500.0, c[1] ? (double)c[0] / (double)c[1] * 0.1 : 0.0,
           ns ? w / ns : 0.0, ns ? m / ns : 0.0, ns);
}

int main(int argc, char** argv)
{
    const double sec = argc > 1 ? atof(argv[1]) : 3.0;             // seconds per variant
    float* d; hipMalloc(&d, 1 << 24);
    run<0, 4>("W0: the direct conv's LDS-fed f16mx mix (4 waves/SIMD x 64 acc)", d, sec);
    run<1, 4>("W1: F(2,3) mix + transform side load (4 waves/SIMD x 64 acc)", d, sec);
    run<2, 8>("W2: F(2,3) mix + transform side load (2 waves/SIMD x 128 acc)", d, sec);
    return 0;
}
