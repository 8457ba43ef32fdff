// Shared helpers for the gfx950 kernels (device + host side of libr3d_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/r3d_hip.h"

namespace r3d {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Optional per-launch-site timing with HIP events recorded ON THE LAUNCH STREAM (r3d_profile_* in r3d_hip.h).
void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);
struct ProfScope {
    int id; hipStream_t st;
    ProfScope(int id_, hipStream_t st_) : id(id_), st(st_) { prof_begin(id, st); }
    ~ProfScope() { prof_end(id, st); }
};
// The shader clock a profiled kernel actually runs at (r3d_profile_clock): when the family's bit is set its launches get a 4-word slot, and
// ONE wave of each launch (wave 0 of a block in the middle of the grid) stores (s_memtime, s_memrealtime) when it starts and adds the two
// deltas to words 0 / 1 when it ends -- cycles / ticks x the constant wall-clock rate = the frequency the power management held during the
// kernel.  nullptr (family not profiled): the kernel pays one uniform branch.  Launches of one family on several streams at once share the
// slot's start words: sample with one stream.
unsigned long long* prof_clock_slot(int id);
__device__ __forceinline__ void clk_begin(unsigned long long* c) { if (c) { c[2] = clock64(); c[3] = wall_clock64(); } }
__device__ __forceinline__ void clk_end(unsigned long long* c) { if (c) { atomicAdd(c, clock64() - c[2]); atomicAdd(c + 1, wall_clock64() - c[3]); } }
// Args::clk of a kernel whose FIRST parameter is an Args struct, read from the kernarg segment at the point of use (only the sampled wave
// executes the load; nothing stays live across the kernel -- a by-value `a.clk` cost the register-tight kernels a spill)
template <class Args>
__device__ __forceinline__ unsigned long long* kernarg_clk()
{
    typedef const __attribute__((address_space(4))) Args* KA;
    KA ap = (KA)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    return ap->clk;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- ordered-int encoding so float min/max can use integer atomics --------------------------------
__device__ __forceinline__ int f2ord(float f) {
    int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int k) {
    return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff);
}

// ---- fast transcendentals on the hardware exp2/log2 (<= ~1 ulp of the result scale) ---------------
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
// torch.nn.Softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplus20(float x) {
    const float r = fmaxf(x, 0.0f) + flog(1.0f + fexp(-fabsf(x)));
    return x > 20.0f ? x : r;
}
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + fexp(-x)); }

// ---- output side: clamp(-1,1) then ((x + 1) / 2 * 255).int() -> uint8 (inference/real3d_infer.py:472,518-522).  (x + 1) / 2 is exact,
// so fl((x + 1) * 127.5) is the same single rounding as fl(((x + 1) / 2) * 255); truncation toward zero; NaN -> 0.
__device__ __forceinline__ uint8_t frame_u8(float v) {
    v = fminf(fmaxf(v, -1.0f), 1.0f);
    return (uint8_t)(int)((v + 1.0f) * 127.5f);
}

// ---- "this fp32 value, as rounded, and nothing else".  Every fp32 -> fp16 hi/lo split starts with it.  Without it hipcc (contract = fast)
// may re-derive the value from its factors at one use and not at another: for  v = a * b; hi = fp16(v); lo = fp16(v - float(hi))  it
// emitted hi (stored) = v_cvt_pk_f16_f32(fl32(a * b)) but lo = v_fma_mixlo_f16(a, b, -h') with h' = v_fma_mixlo_f16(a, b, 0), the SINGLY
// rounded product: where the double rounding of `hi` and the single rounding of h' disagree, hi + lo is off by a whole fp16 ulp of hi
// (2^-11 relative, rare) -- a 16x loss of accuracy of SynthesisBlockNoUp found by the range sweeps.  tests/test_abi.py lints the
// device assembly for the fused-rounding form.
__device__ __forceinline__ float as_rounded(float v) { asm("" : "+v"(v)); return v; }

// ---- fp32 -> SPLIT (fp16 hi + fp16 lo) of x * s, the conversion every producer of an SR block's first operand performs (to_split_kernel in
// r3d_sr_f16x3.hip, the ray kernel's split_out in r3d_render.hip).  The residual comes from the exact product (one fma).  The fma is
// opaque to the compiler on purpose: left to itself it folds fptrunc(fma) into v_fma_mixlo_f16 at some sites and not at others
// (single vs double rounding: a 1-ulp difference of lo on fp16 ties, seen as a 3e-5 difference of the SR image between two
// producers), and the producers must agree bit for bit.
__device__ __forceinline__ void split_scaled(float x, float s, _Float16& hi, _Float16& lo) {
    const float p = x * s;
    hi = (_Float16)fminf(fmaxf(p, -65504.f), 65504.f);
    const float h32 = (float)hi;
    float r;
    asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(r) : "v"(x), "v"(s), "v"(h32));
    lo = (_Float16)r;
}

// ---- counter-based uniform [0,1) (used when the caller passes no noise tensors) ---------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint32_t stream, uint64_t idx) {
    uint32_t h = mix32((uint32_t)seed ^ 0x9E3779B9U * (stream + 1));
    h = mix32(h ^ (uint32_t)(seed >> 32));
    h = mix32(h ^ (uint32_t)idx);
    h = mix32(h ^ (uint32_t)(idx >> 32) ^ 0x85ebca6bU);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

}  // namespace r3d
