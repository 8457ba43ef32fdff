// What the f16mx MFMA mix of the dominant conv (per 2 taps x 16 channels and 32x32 tile: 2 x v_mfma_f32_32x32x16_f16 + 1 x
// v_mfma_scale_f32_32x32x64_f8f6f4, fp8) sustains on THIS board when nothing but the matrix pipe (and then its LDS feed) is busy -- rate, the clock
// the waves actually ran at (s_memtime / s_memrealtime) and the socket power rocm-smi reports meanwhile.  DESIGN 4.4a: the MFMA peaks are quoted at
// 2.4 GHz; the board clocks to its 1.4 kW budget.
//   R0  operands in registers, CONSTANT          (the round-2 probe: data that never toggles draws the least power)
//   R1  operands in registers, re-randomised by a few VALU xors per step (full-entropy mantissas, finite values)
//   L1  operands read from LDS every step at the conv's ratio (16 ds_read_b128 per 12 MFMAs), random LDS contents, no DMA, no barrier
// 512 blocks x 512 threads (4 waves/SIMD at 2 blocks/CU, 64 accumulators per wave: the conv's shape), ~3 s per variant.
// build: hipcc -O3 --offload-arch=gfx950 -o power_ceiling_probe power_ceiling_probe.hip -lpthread
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

template <int VAR>
__global__ __launch_bounds__(512, 4) void mix_kernel(float* out, int iters, unsigned seed)
{
    __shared__ u4 lds[2048];                                        // L1: 32 KB of random operands
    const int lane = threadIdx.x & 63;
    unsigned s = seed ^ (blockIdx.x * 512u + threadIdx.x) * 2654435761u;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    if (VAR == 2) {
        for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = (u4){f16pair(rnd()), f16pair(rnd()), fp8quad(rnd()), fp8quad(rnd())};
        __syncthreads();
    }
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = clock64(); r0 = wall_clock64(); }
    u4 ah = {f16pair(rnd()), f16pair(rnd()), f16pair(rnd()), f16pair(rnd())}, bh = {f16pair(rnd()), f16pair(rnd()), f16pair(rnd()), f16pair(rnd())};
    i8v a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = (int)fp8quad(rnd()); b8[j] = (int)fp8quad(rnd()); }
    f16v c[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {                               // one conv sub-stage (2 taps x 16 channels) for the wave's 4 tiles: 8 f16 + 4 fp8 MFMAs
            if (VAR == 1 || VAR >= 3) {                             // new bits in every operand register (24 xors per 512 MFMA cycles)
                const unsigned k = rnd();
#pragma unroll
                for (int j = 0; j < 4; ++j) { ah[j] = f16pair(ah[j] ^ (k * (2 * j + 1))); bh[j] = f16pair(bh[j] ^ (k * (2 * j + 3))); }
#pragma unroll
                for (int j = 0; j < 8; ++j) { a8[j] = (int)fp8quad((unsigned)a8[j] ^ (k * (j + 5))); b8[j] = (int)fp8quad((unsigned)b8[j] ^ (k * (j + 11))); }
            }
            if (VAR == 2) {                                         // 16 ds_read_b128: 4 A + 4 B fragments for the f16 part, 4 + 4 for the fp8 records
                const int base = ((it * 4 + u) * 67 + lane) & 2047;
                u4 f[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) f[j] = lds[(base + 128 * j) & 2047];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u4 A0 = f[t & 1], B0 = f[2 + (t >> 1)], A1 = f[4 + (t & 1)], B1 = f[6 + (t >> 1)];
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&A0, *(const h8*)&B0, c[t], 0, 0, 0);
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&A1, *(const h8*)&B1, c[t], 0, 0, 0);
                    const u4 q0 = f[8 + (t & 1)], q1 = f[10 + (t & 1)], p0 = f[12 + (t >> 1)], p1 = f[14 + (t >> 1)];
                    const i8v A8 = {(int)fp8quad(q0[0]), (int)fp8quad(q0[1]), (int)q0[2], (int)q0[3], (int)fp8quad(q1[0]), (int)fp8quad(q1[1]), (int)q1[2], (int)q1[3]};
                    const i8v B8 = {(int)fp8quad(p0[0]), (int)fp8quad(p0[1]), (int)p0[2], (int)p0[3], (int)fp8quad(p1[0]), (int)fp8quad(p1[1]), (int)p1[2], (int)p1[3]};
                    c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8, B8, c[t], 0, 0, 0, 116, 0, 127);
                }
            } else if (VAR == 3) {                                  // the mix with the cross products on fp6 e2m3 (same K = 64, half the passes; operands: 24 of the 32 bytes)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&ah, *(const h8*)&bh, c[t], 0, 0, 0);
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&bh, *(const h8*)&ah, c[t], 0, 0, 0);
                    c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[t], 2, 2, 0, 116, 0, 127);
                }
            } else if (VAR == 4) {                                  // f16 MFMAs only (4 per tile: the same matrix cycles as the mix)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&ah, *(const h8*)&bh, c[t], 0, 0, 0);
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&bh, *(const h8*)&ah, c[t], 0, 0, 0);
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&ah, *(const h8*)&bh, c[t], 0, 0, 0);
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&bh, *(const h8*)&ah, c[t], 0, 0, 0);
                }
            } else if (VAR == 5) {                                  // fp8 K = 64 MFMAs only (2 per tile: the same matrix cycles)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[t], 0, 0, 0, 116, 0, 127);
                    c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, c[t], 0, 0, 0, 116, 0, 127);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&ah, *(const h8*)&bh, c[t], 0, 0, 0);
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8*)&bh, *(const h8*)&ah, c[t], 0, 0, 0);
                    c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[t], 0, 0, 0, 116, 0, 127);
                }
            }
        }
    }
    float acc = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc += c[t][r];
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

template <int VAR> void run(const char* name, float* d, double seconds)
{
    const int iters = 1500, blocks = 512, threads = 512;
    unsigned long long z[2] = {0, 0};
    mix_kernel<VAR><<<blocks, threads>>>(d, 10, 1u); hipDeviceSynchronize();
    hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z));
    Smi smi; std::thread th(poll, &smi);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    double ms_sum = 0; int n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int k = 0; k < 8; ++k) mix_kernel<VAR><<<blocks, threads>>>(d, iters, 7u + n * 8 + k);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms_sum += ms; n += 8;
    }
    smi.stop.store(true); th.join();
    unsigned long long c[2]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c));
    const double ms = ms_sum / n;
    const double steps = (double)blocks * (threads / 64) * iters * 16;       // (2 taps x 16 channels x one 32x32 tile) units
    const double alg = steps * 2.0 * 32 * 32 * 32;                           // algorithmic FLOPs (32 MAC-channels per unit and output)
    double w = 0, m = 0; size_t k0 = smi.w.size() > 2 ? 1 : 0;
    for (size_t i = k0; i < smi.w.size(); ++i) { w += smi.w[i]; m += smi.mhz[i]; }
    const size_t ns = smi.w.size() - k0;
    printf("%-64s %.3f ms/launch  %5.0f algorithmic TFLOP/s = %.3f of 2500 (pipe %.3f)  waves at %.3f GHz  socket %4.0f W  sclk %4.0f MHz (%zu samples)\n", name, ms,
           alg / (ms * 1e-3) / 1e12, alg / (ms * 1e-3) / 1e12 / 2500.0, 2.0 * alg / (ms * 1e-3) / 1e12 / 2500.0, c[1] ? (double)c[0] / (double)c[1] * 0.1 : 0.0,
           ns ? w / ns : 0.0, ns ? m / ns : 0.0, ns);
}

int main(int argc, char** argv)
{
    const double sec = argc > 1 ? atof(argv[1]) : 3.0;             // seconds per variant
    float* d; hipMalloc(&d, 1 << 24);
    run<0>("R0: f16mx MFMA mix, operands in registers, constant", d, sec);
    run<1>("R1: ... operands in registers, re-randomised every sub-stage", d, sec);
    run<2>("L1: ... operands from LDS (16 ds_read_b128 per 12 MFMAs), random", d, sec);
    if (argc > 2) {                                                 // the instruction types on their own (TFLOP/s figures of these lines: read "matrix passes", see below)
        run<3>("X6: the mix with the cross products on fp6 (registers, random)", d, sec);
        run<4>("XF: f16 MFMAs only, 4 per tile and sub-stage (registers, random)", d, sec);
        run<5>("X8: fp8 K=64 MFMAs only, 2 per tile and sub-stage (registers, random)", d, sec);
    }
    return 0;
}
