// Host driver for ISA-level bisection of the co-execution anomaly: loads victim kernels from code objects (.hsaco, assembled
// from mutated copies of a failing kernel's device assembly by scripts/probes/asm_bisect.py) and runs each one alone and next to the MFMA
// aggressor of coexec_probe.hip on another stream.  usage: coexec_mod <kernel-name> <iters> a.hsaco [b.hsaco ...]
// victim signature: (unsigned* err, int iters, int salt, int H, int W)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void agg_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters, int src_mask)
{
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* mine = lds + wave * 256;
    for (int i = lane; i < 256; i += 64) mine[i] = src[(blockIdx.x * 512 + wave * 64 + i) & src_mask];
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    uint4 ra = mine[lane], rb = mine[64 + lane];
    for (int it = 0; it < iters; ++it) {
        ra = mine[(lane + it) & 63]; rb = mine[64 + ((lane + 2 * it) & 63)];
        const h8 a = *reinterpret_cast<h8*>(&ra), b = *reinterpret_cast<h8*>(&rb);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)ra.x;
}

int main(int argc, char** argv)
{
    if (argc < 4) { printf("usage: coexec_mod <kernel> <iters> a.hsaco ...\n"); return 2; }
    const char* kname = argv[1];
    int iters = atoi(argv[2]);
    hipStream_t sa, sv;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    const int src_elems = 1 << 18;
    std::vector<_Float16> hsrc((size_t)src_elems * 8);
    for (size_t i = 0; i < hsrc.size(); ++i) hsrc[i] = (_Float16)(0.01f * (float)((int)(i * 2654435761u >> 20) % 200 - 100));
    uint4* dsrc; float* dout; unsigned* derr;
    CK(hipMalloc(&dsrc, (size_t)src_elems * 16)); CK(hipMemcpy(dsrc, hsrc.data(), (size_t)src_elems * 16, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, 1024 * 512 * 4)); CK(hipMalloc(&derr, 4096 * sizeof(unsigned)));
    CK(hipFuncSetAttribute((const void*)agg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    for (int f = 3; f < argc; ++f) {
        hipModule_t mod; hipFunction_t fn;
        if (hipModuleLoad(&mod, argv[f]) != hipSuccess) { printf("%s: cannot load\n", argv[f]); continue; }
        if (hipModuleGetFunction(&fn, mod, kname) != hipSuccess) { printf("%s: no kernel %s\n", argv[f], kname); continue; }
        unsigned res[2][20];
        for (int with_agg = 0; with_agg < 2; ++with_agg) {
            CK(hipMemsetAsync(derr, 0, 4096 * sizeof(unsigned), sv)); CK(hipStreamSynchronize(sv));
            if (with_agg)
                for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(agg_kernel, dim3(1024), dim3(512), 76 * 1024, sa, dsrc, dout, 150, src_elems - 1);
            for (int rep = 0; rep < 6; ++rep) {
                int H = 256, W = 256, it = iters, salt = rep;
                void* args[] = {&derr, &it, &salt, &H, &W};
                CK(hipModuleLaunchKernel(fn, 2048, 1, 1, 256, 1, 1, 0, sv, args, nullptr));
            }
            CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
            CK(hipMemcpy(res[with_agg], derr, sizeof(res[0]), hipMemcpyDeviceToHost));
        }
        if (getenv("DUMP")) {          // surgery builds store 4 dwords per lane at byte 1024 + 16 * lane when their compare fires
            unsigned d[256]; CK(hipMemcpy(d, derr + 256, sizeof(d), hipMemcpyDeviceToHost));
            if (getenv("DUMP")[0] == 'y') { printf("   per-lane word:"); for (int l = 0; l < 64; ++l) printf(" %x", d[4 * l]); printf("\n"); }
            else if (getenv("DUMP")[0] == 'x') { for (int l = 0; l < 64; l += 21) printf("   lane %2d: VALU view %08x  SALU view %08x\n", l, d[4 * l], d[4 * l + 1]); }
            else for (int l = 0; l < 64; ++l) if (d[4 * l] | d[4 * l + 1]) { float f[4]; memcpy(f, d + 4 * l, 16);
                printf("   lane %2d: %.9g (%08x)  %.9g (%08x)  %.9g  %.9g\n", l, f[0], d[4 * l], f[1], d[4 * l + 1], f[2], f[3]); }
        }
        printf("%-40s alone %9u | next to MFMA load %9u  (rows %u %u %u %u | idx %u w %u | plane %u %u %u | tap %u %u %u %u)\n", argv[f], res[0][0], res[1][0], res[1][1], res[1][2], res[1][3], res[1][4],
               res[1][8], res[1][9], res[1][10], res[1][11], res[1][12], res[1][13], res[1][14], res[1][15], res[1][16]);
        CK(hipModuleUnload(mod));
    }
    return 0;
}
