// Probe: throughput of v_mfma_f32_32x32x16_f16 under different accumulator dependency patterns and occupancies.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int PATTERN>
__global__ __launch_bounds__(512, 4) void k(float* out, int iters) {
    h8 a0, a1, b0, b1;
    for (int j = 0; j < 8; ++j) { a0[j] = (_Float16)(threadIdx.x * 0.001f + j); a1[j] = (_Float16)(j * 0.5f); b0[j] = (_Float16)(1.0f / (j + 1)); b1[j] = (_Float16)(0.25f * j); }
    f16v c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 0; c2[r] = 0; c3[r] = 0; }
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0) {          // as in the conv: two accumulators alternate, 3 products each (dependent distance 2)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c3, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c3, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c3, 0, 0, 0);
            }
        } else if (PATTERN == 1) {   // four accumulators round-robin (dependent distance 4)
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c3, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c3, 0, 0, 0);
            }
        } else {                     // one accumulator (fully dependent)
#pragma unroll
            for (int u = 0; u < 24; ++u) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
        }
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int P> void run(const char* name, int blocks, int threads, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<P><<<blocks, threads>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<P><<<blocks, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * (threads / 64) * iters * 24;
    const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-28s blocks %4d x %3d thr: %.3f ms  %.0f TFLOP/s (%.1f%% of 2500)\n", name, blocks, threads, ms, tf, tf / 25.0);
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    for (int thr : {256, 512}) for (int bpc : {1, 2}) {
        const int blocks = 256 * bpc;
        run<0>("2 acc alternating x3", blocks, thr, d); run<1>("4 acc round-robin", blocks, thr, d); run<2>("1 acc dependent", blocks, thr, d);
    }
    return 0;
}
