// Probe: v_mfma_f32_32x32x16_f16 operand layout + fp16 subnormal handling + 3-term split accuracy.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// D[32x32] = A[32x16] * B[16x32]; A row-major [i][k], B [k][j]
__global__ void k(const float* A, const float* B, float* D, int mode) {
    int l = threadIdx.x, i = l & 31, h = l >> 5;
    h8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
        float a = A[i * 16 + 8 * h + j], b = B[(8 * h + j) * 32 + i];
        _Float16 x = (_Float16)a; ah[j] = x; al[j] = (_Float16)(a - (float)x);
        _Float16 y = (_Float16)b; bh[j] = y; bl[j] = (_Float16)(b - (float)y);
    }
    f16v c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    if (mode >= 1) { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0); }
    for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * h; D[row * 32 + i] = c[r]; }
}
int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32]; float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    for (int scale_i = 0; scale_i < 3; ++scale_i) {
        float sc = scale_i == 0 ? 1.f : (scale_i == 1 ? 1e-2f : 1e-4f);
        srand(1);
        for (int i = 0; i < 512; ++i) { hA[i] = sc * (rand() / (float)RAND_MAX * 2 - 1); hB[i] = (rand() / (float)RAND_MAX * 2 - 1); }
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            k<<<1, 64>>>(dA, dB, dD, mode); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
            double maxerr = 0, maxref = 0;
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double r = 0; for (int kk = 0; kk < 16; ++kk) r += (double)hA[i * 16 + kk] * hB[kk * 32 + j];
                maxerr = fmax(maxerr, fabs(r - hD[i * 32 + j])); maxref = fmax(maxref, fabs(r)); }
            printf("scale %g mode %d (1=3-term split): max abs err %.3e  max|ref| %.3e  rel %.3e\n", sc, mode, maxerr, maxref, maxerr / maxref);
        }
    }
    return 0;
}
