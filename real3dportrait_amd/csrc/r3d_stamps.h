// Instrumented EXPERIMENT builds only (-DR3D_STAMPS; never the shipped library): per-phase cycle accumulators of a kernel.
// A wave adds the s_memtime deltas of its phases into registers and one lane adds them to the translation unit's table at the end
// (R3D_STAMP_READER(name) defines the extern "C" function that reads and clears it).  s_memtime itself costs ~10 % of the wave cycles
// and the accumulators cost registers: read the SPLIT, not the total.
#pragma once
#ifdef R3D_STAMPS
namespace r3d { static __device__ unsigned long long g_stamps[32]; }
#define R3D_STAMP_DECL unsigned long long st_acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long st_r0_ = wall_clock64(); \
                       unsigned long long st_t_ = clock64(); const unsigned long long st_c0_ = st_t_
// the shader clock the wave actually ran at: slots (slot, slot + 1) -- 28 / 29 from R3D_STAMP_FLUSH -- accumulate the wave's s_memtime cycles and its s_memrealtime ticks (constant rate,
// hipDeviceAttributeWallClockRate); call once per wave, at the end
#define R3D_STAMP_CLOCKS(slot) do { if ((threadIdx.x & 63) == 0) { atomicAdd(&r3d::g_stamps[slot], clock64() - st_c0_); atomicAdd(&r3d::g_stamps[(slot) + 1], wall_clock64() - st_r0_); } } while (0)
#define R3D_STAMP(i) do { const unsigned long long n_ = clock64(); st_acc_[i] += n_ - st_t_; st_t_ = n_; } while (0)
#define R3D_STAMP_FLUSH(nphase, units) do { R3D_STAMP_CLOCKS(28); if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < (nphase); ++i_) atomicAdd(&r3d::g_stamps[i_], st_acc_[i_]); \
                                                atomicAdd(&r3d::g_stamps[31], (unsigned long long)(units)); } } while (0)
// out[0..30] = summed cycles per phase over all waves, out[31] = units; clears the table
#define R3D_STAMP_READER(name) extern "C" int name(unsigned long long* host) { \
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(r3d::g_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return -1; \
    unsigned long long z[32] = {}; return hipMemcpyToSymbol(HIP_SYMBOL(r3d::g_stamps), z, sizeof(z)) == hipSuccess ? 0 : -1; }
#else
#define R3D_STAMP_DECL do { } while (0)
#define R3D_STAMP(i) do { } while (0)
#define R3D_STAMP_FLUSH(nphase, units) do { } while (0)
#define R3D_STAMP_CLOCKS(slot) do { } while (0)
#define R3D_STAMP_READER(name)
#endif
