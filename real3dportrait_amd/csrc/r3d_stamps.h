// Instrumented EXPERIMENT builds only (-DR3D_STAMPS; never the shipped library): per-phase cycle accumulators of a kernel.
// A wave adds the s_memtime deltas of its phases into registers and one lane adds them to the translation unit's table at the end
// (R3D_STAMP_READER(name) defines the extern "C" function that reads and clears it).  s_memtime itself costs ~10 % of the wave cycles
// and the accumulators cost registers: read the SPLIT, not the total.
#pragma once
#ifdef R3D_STAMPS
namespace r3d { static __device__ unsigned long long g_stamps[32]; }
#define R3D_STAMP_DECL unsigned long long st_acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t_ = clock64()
#define R3D_STAMP(i) do { const unsigned long long n_ = clock64(); st_acc_[i] += n_ - st_t_; st_t_ = n_; } while (0)
#define R3D_STAMP_FLUSH(nphase, units) do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < (nphase); ++i_) atomicAdd(&r3d::g_stamps[i_], st_acc_[i_]); \
                                                atomicAdd(&r3d::g_stamps[31], (unsigned long long)(units)); } } while (0)
// out[0..30] = summed cycles per phase over all waves, out[31] = units; clears the table
#define R3D_STAMP_READER(name) extern "C" int name(unsigned long long* host) { \
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(r3d::g_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return -1; \
    unsigned long long z[32] = {}; return hipMemcpyToSymbol(HIP_SYMBOL(r3d::g_stamps), z, sizeof(z)) == hipSuccess ? 0 : -1; }
#else
#define R3D_STAMP_DECL do { } while (0)
#define R3D_STAMP(i) do { } while (0)
#define R3D_STAMP_FLUSH(nphase, units) do { } while (0)
#define R3D_STAMP_READER(name)
#endif
