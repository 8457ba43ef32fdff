// Synthetic "co-resident load" kernels for the determinism bisect of DESIGN 4.1 (scripts/gpu_debug_determinism.py, `agg:<mode>`):
// a block shaped like the SR conv kernel (512 threads, <= 128 VGPRs, 2 blocks per CU) whose inner loop is assembled from feature bits
//   1  f16 MFMAs (v_mfma_f32_32x32x16_f16, 16 per iteration and wave)          2  LDS-DMA (global_load_lds_dwordx4 through m0, the SR's dma64)
//   4  ds_read_b128 operand reads from LDS                                      8  one __syncthreads per iteration
//   16 76 KB of LDS per block (the SR conv's footprint; otherwise 8 KB)          32 packed-f32 VALU work instead of / next to the MFMAs
// so that the ingredient of the SR kernels that changes the ray kernel variant's results can be isolated.  Not product code.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma64(const uint4* gsrc, uint4* lds_dst_uniform)
{
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) uint4*)lds_dst_uniform);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(d) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void agg_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters, int src_mask)
{
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* mine = lds + wave * 256;                       // 4 KB per wave (8 KB total is the small footprint; the big one only pads)
    for (int i = lane; i < 256; i += 64) mine[i] = src[(blockIdx.x * 512 + wave * 64 + i) & src_mask];
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    uint4 ra = mine[lane], rb = mine[64 + lane];
    float2 pk = make_float2(1.0f + lane, 0.5f);
    for (int it = 0; it < iters; ++it) {
        if (MODE & 2) {
            dma64(src + ((blockIdx.x * 8191 + it * 512 + wave * 64 + lane) & src_mask), mine + 128 * (it & 1));
        }
        if (MODE & 4) { ra = mine[(lane + it) & 63]; rb = mine[64 + ((lane + 2 * it) & 63)]; }
        const h8 a = *reinterpret_cast<h8*>(&ra), b = *reinterpret_cast<h8*>(&rb);
        if (MODE & 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        }
        if (MODE & 32) {
#pragma unroll
            for (int k = 0; k < 64; ++k) { pk.x = pk.x * 1.0000001f + pk.y; pk.y = pk.y * 0.9999999f + pk.x * 1e-9f; }
        }
        if (MODE & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE & 8) __syncthreads();
    }
    float s = pk.x + pk.y;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)ra.x;
}

template <int MODE>
static int launch(const void* src, int src_elems, float* out, int grid, int iters, hipStream_t st)
{
    const size_t lds = (MODE & 16) ? 76 * 1024 : 8 * 1024;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)agg_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr_done = true; }
    hipLaunchKernelGGL((agg_kernel<MODE>), dim3(grid), dim3(512), lds, st, (const uint4*)src, out, iters, src_elems - 1);
    return (int)hipGetLastError();
}

// src: device buffer of src_elems uint4 (power of two, >= 64 Ki); out: float[grid * 512]
extern "C" int agg_launch(int mode, const void* src, int src_elems, float* out, int grid, int iters, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
#define C_(M) case M: return launch<M>(src, src_elems, out, grid, iters, st);
        C_(17) C_(25) C_(29) C_(31) C_(30) C_(28) C_(22) C_(16) C_(15) C_(9) C_(1) C_(48) C_(56) C_(49)
#undef C_
    }
    return -1;
}
