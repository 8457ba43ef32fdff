// Probe for DESIGN section 8 ("cheaper correction terms"): what would the f16x3 convolution's matrix work cost if the two correction
// products (hi*lo, lo*hi: 2^-11-sized) ran on the block-scaled fp8 / fp6 MFMA instead of the f16 MFMA?
//   mix A (today)    per 32 input channels x 1 tap x one 32x32 tile: 6 x v_mfma_f32_32x32x16_f16          (hi*hi, hi*lo, lo*hi, K = 16 each)
//   mix B            2 x f16 (hi*hi) + 1 x v_mfma_scale_f32_32x32x64_f8f6f4, fp8 e4m3:  K = 64 = [xh8 | xl8] . [wl8 | wh8]
//   mix C            2 x f16 (hi*hi) + 1 x the same instruction with fp6 e2m3 operands
// Part 1 checks the operand layout assumption of the K = 64 instruction (lane l: row / column l % 32, k = 32 * (l / 32) + byte index) against a
// CPU product on small integers; part 2 times the three mixes from registers at 4 waves/SIMD, 256 CUs x 2 blocks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));

static uint8_t e4m3(int v) {          // small integers |v| <= 8 are exact in OCP e4m3 (bias 7, 3 mantissa bits)
    if (v == 0) return 0;
    const int s = v < 0, a = abs(v);
    int e = 0; while ((1 << (e + 1)) <= a) ++e;               // a in [2^e, 2^(e+1))
    const int m = ((a << 3) >> e) & 7;
    return (uint8_t)((s << 7) | ((e + 7) << 3) | m);
}

__global__ void layout_kernel(const uint8_t* A /*[32][64]*/, const uint8_t* B /*[64][32]*/, float* C /*[32][32]*/, int scale)
{
    const int l = threadIdx.x, row = l & 31, kh = l >> 5;
    i8v a, b;
    for (int w = 0; w < 8; ++w) {
        uint32_t wa = 0, wb = 0;
        for (int j = 0; j < 4; ++j) {
            const int k = 32 * kh + 4 * w + j;
            wa |= (uint32_t)A[row * 64 + k] << (8 * j);
            wb |= (uint32_t)B[k * 32 + row] << (8 * j);
        }
        a[w] = (int)wa; b[w] = (int)wb;
    }
    f16v c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    // cbsz = blgp = 0: fp8 e4m3 for A and B; E8M0 scale bytes (127 = 2^0) in byte 0 of the scale registers
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale, 0, 127);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];     // C/D map of the 32x32 shapes
}

template <int MIX>
__global__ __launch_bounds__(512, 4) void mix_kernel(float* out, int iters)
{
    h8 ah, al, bh, bl;
    i8v a8, b8;
    for (int j = 0; j < 8; ++j) {
        ah[j] = (_Float16)(threadIdx.x * 0.001f + j); al[j] = (_Float16)(j * 0.0005f); bh[j] = (_Float16)(1.0f / (j + 1)); bl[j] = (_Float16)(0.00025f * j);
        a8[j] = 0x38403c30 + 0x01010101 * (threadIdx.x & 7) + j; b8[j] = 0x3a34423c + 0x01010101 * j;      // arbitrary finite fp8 / fp6 bit patterns
    }
    f16v c[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {                          // one "32 channels x 1 tap" step for each of 4 accumulator tiles
                c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[t], 0, 0, 0);
                c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[t], 0, 0, 0);
                if (MIX == 0) {
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c[t], 0, 0, 0); c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c[t], 0, 0, 0);
                    c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c[t], 0, 0, 0); c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c[t], 0, 0, 0);
                } else if (MIX == 1) {
                    c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[t], 0, 0, 0, 116, 0, 127);       // fp8 e4m3, scale 2^-11
                } else {
                    c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[t], 2, 2, 0, 116, 0, 127);       // fp6 e2m3
                }
            }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += c[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MIX> void run(const char* name, float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1500, blocks = 512, threads = 512;
    mix_kernel<MIX><<<blocks, threads>>>(d, 10); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); mix_kernel<MIX><<<blocks, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double steps = (double)blocks * (threads / 64) * iters * 16;       // (32 channels x 1 tap x 32x32 tile) steps
    const double alg = steps * 2.0 * 32 * 32 * 32;                           // algorithmic FLOPs of those steps
    printf("%-44s %.3f ms  %.0f algorithmic TFLOP/s (f16x3 mix = 1.00 -> %.2fx needs the first line)\n", name, best, alg / (best * 1e-3) / 1e12, 0.0);
}

int main()
{
    // ---- part 1: layout -------------------------------------------------------------------------------------------------
    uint8_t hA[32 * 64], hB[64 * 32]; int iA[32 * 64], iB[64 * 32];
    srand(7);
    for (int i = 0; i < 32 * 64; ++i) { iA[i] = rand() % 9 - 4; hA[i] = e4m3(iA[i]); iB[i] = rand() % 9 - 4; hB[i] = e4m3(iB[i]); }
    uint8_t *dA, *dB; float* dC; hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    for (int scale : {127, 116}) {
        layout_kernel<<<1, 64>>>(dA, dB, dC, scale);
        float hC[32 * 32]; hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
        double maxerr = 0, sc = scale == 127 ? 1.0 : 1.0 / 2048.0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ref = 0; for (int k = 0; k < 64; ++k) ref += (double)iA[i * 64 + k] * iB[k * 32 + j];
            const double e = fabs(hC[i * 32 + j] - ref * sc); if (e > maxerr) maxerr = e;
        }
        printf("layout check (lane l: row l%%32, k = 32*(l/32) + byte), scale_a byte %d: max |C - A.B * 2^%d| = %g\n", scale, scale - 127, maxerr);
    }
    // ---- part 2: rates -----------------------------------------------------------------------------------------------------
    float* d; hipMalloc(&d, 1 << 24);
    run<0>("A: 6 x f16 (today's f16x3)", d);
    run<1>("B: 2 x f16 + 1 x scaled fp8 K=64", d);
    run<2>("C: 2 x f16 + 1 x scaled fp6 K=64", d);
    return 0;
}
