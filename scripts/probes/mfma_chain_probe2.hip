// Probe 2: MFMA throughput vs the number of consecutive MFMAs issued on the same accumulator (chain length L),
// cycling over 4 accumulators, with varying A/B operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int L, bool SAMEOPS>
__global__ __launch_bounds__(512, 4) void k(float* out, int iters) {
    h8 a[2], b[2];
    for (int j = 0; j < 8; ++j) { a[0][j] = (_Float16)(threadIdx.x * 0.001f + j); a[1][j] = (_Float16)(j * 0.5f); b[0][j] = (_Float16)(1.0f / (j + 1)); b[1][j] = (_Float16)(0.25f * j); }
    f16v c[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 24 / (4 * L) + (24 % (4 * L) ? 1 : 0); ++rep)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int l = 0; l < L; ++l)
                    c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[SAMEOPS ? 0 : (l & 1)], b[SAMEOPS ? 0 : ((l >> 1) & 1)], c[q], 0, 0, 0);
    }
    float s = 0; for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int L, bool S> void run(float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 512, threads = 512;
    const int per_iter = (24 / (4 * L) + (24 % (4 * L) ? 1 : 0)) * 4 * L;
    k<L, S><<<blocks, threads>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<L, S><<<blocks, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * (threads / 64) * iters * per_iter * 32768.0 / (ms * 1e-3) / 1e12;
    printf("chain L=%2d %s: %.0f TFLOP/s (%.1f%% of 2500)\n", L, S ? "same A/B " : "vary A/B ", tf, tf / 25.0);
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    run<1, false>(d); run<2, false>(d); run<3, false>(d); run<4, false>(d); run<6, false>(d); run<8, false>(d); run<12, false>(d); run<24, false>(d);
    run<1, true>(d); run<3, true>(d); run<6, true>(d);
    return 0;
}
