// Probe 3: the f16x3 conv inner loop in isolation (LDS operand reads + MFMAs, no global traffic), 8 waves/block,
// 2 blocks/CU, to find the MFMA/LDS issue pattern that reaches the matrix-pipe peak.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
template <int V>
__global__ __launch_bounds__(512, 4) void k(float* out, int iters) {
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = make_uint4(0x3c003c00u + i, 0x3c003800u, 0x38003c00u, 0x3c003c00u);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* A = lds + (wave & 1) * 64 + (lane & 31) + (lane >> 5) * 256;      // hi at +0, lo at +128; mt*32
    const uint4* B = lds + 1024 + (wave >> 1) * 40 + lane;                           // hi at +0, lo at +648; nt*36
    f16v c[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) c[m][n][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int ao = t * 512, bo = t * 19;
            if (V == 0) {            // current: A reads, then per nt: B reads + 6 MFMAs alternating accumulators
                uint4 q[2][2];
                for (int m = 0; m < 2; ++m) { q[m][0] = A[ao + m * 32]; q[m][1] = A[ao + m * 32 + 128]; }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    uint4 r0 = B[bo + n * 36], r1 = B[bo + n * 36 + 648];
                    h8 bh = *(h8*)&r0, bl = *(h8*)&r1;
#pragma unroll
                    for (int m = 0; m < 2; ++m) { MF(*(h8*)&q[m][1], bh, c[m][n]); }
#pragma unroll
                    for (int m = 0; m < 2; ++m) { MF(*(h8*)&q[m][0], bl, c[m][n]); }
#pragma unroll
                    for (int m = 0; m < 2; ++m) { MF(*(h8*)&q[m][0], bh, c[m][n]); }
                }
            } else if (V == 1) {     // all 8 operand reads first, then 12 MFMAs in chains of 3
                uint4 q[2][2], r[2][2];
                for (int m = 0; m < 2; ++m) { q[m][0] = A[ao + m * 32]; q[m][1] = A[ao + m * 32 + 128]; }
                for (int n = 0; n < 2; ++n) { r[n][0] = B[bo + n * 36]; r[n][1] = B[bo + n * 36 + 648]; }
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        MF(*(h8*)&q[m][1], *(h8*)&r[n][0], c[m][n]); MF(*(h8*)&q[m][0], *(h8*)&r[n][1], c[m][n]); MF(*(h8*)&q[m][0], *(h8*)&r[n][0], c[m][n]);
                    }
            } else {                 // V == 2: same as 1 but with an explicit scheduling fence between taps
                uint4 q[2][2], r[2][2];
                for (int m = 0; m < 2; ++m) { q[m][0] = A[ao + m * 32]; q[m][1] = A[ao + m * 32 + 128]; }
                for (int n = 0; n < 2; ++n) { r[n][0] = B[bo + n * 36]; r[n][1] = B[bo + n * 36 + 648]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        MF(*(h8*)&q[m][1], *(h8*)&r[n][0], c[m][n]); MF(*(h8*)&q[m][0], *(h8*)&r[n][1], c[m][n]); MF(*(h8*)&q[m][0], *(h8*)&r[n][0], c[m][n]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0; for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += c[m][n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V> void run(float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1500, blocks = 512, threads = 512;
    k<V><<<blocks, threads>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<V><<<blocks, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * (threads / 64) * iters * 36 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("variant %d: %.3f ms %.0f TFLOP/s executed (%.1f%% of 2500)\n", V, ms, tf, tf / 25.0);
}
int main() { float* d; hipMalloc(&d, 1 << 24); run<0>(d); run<1>(d); run<2>(d); run<0>(d); run<1>(d); return 0; }
