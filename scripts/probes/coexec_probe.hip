// Minimal reproducer hunt for the "result depends on what else is resident" anomaly: small VICTIM kernels, each looping over
// one instruction pattern whose result is checked in-kernel against a value computed without that pattern, run (a) alone and (b) while
// an AGGRESSOR kernel (f16 MFMAs + LDS operand reads + one barrier per iteration, shaped like the SR conv: 512 threads, 76 KB LDS)
// occupies the same CUs from another stream.  Any mismatch count > 0 in (b) with 0 in (a) names the pattern.
//   build: hipcc -O3 --offload-arch=gfx950 -o scripts/probes/bin/coexec_probe scripts/probes/coexec_probe.hip      run: coexec_probe [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- aggressor (scripts/probes/aggressor.hip mode 29 / 25 / 17 / 49) ------------------------------------------------------------------
template <int MODE>
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
    float2 pk = make_float2(1.0f + lane, 0.5f);
    for (int it = 0; it < iters; ++it) {
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
        if (MODE & 8) __syncthreads();
    }
    float s = pk.x + pk.y;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)ra.x;
}

// ---- victims ------------------------------------------------------------------------------------------------------------------------
// err[0] = mismatching (lane, iteration) events, err[1 + r] = events in DPP row r (lanes 16r .. 16r+15)
__device__ __forceinline__ void report(unsigned* err, int bad, int lane)
{
    if (bad) { atomicAdd(&err[0], (unsigned)bad); atomicAdd(&err[1 + (lane >> 4)], (unsigned)bad); }
}

// pattern 0..2: VALU write -> s_nop -> v_mov_b32_dpp quad_perm read of that VGPR, at 2 / 3 / 6 wait states (2 = the documented minimum)
template <int WAIT>
__global__ __launch_bounds__(256) void victim_raw_dpp(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63;
    const int L = lane * 0x9E3779B1u + salt * 77 + blockIdx.x;                  // per-lane value
    const int Lq = __shfl(L, (lane & ~3) | 1);                                  // what quad_perm:[1,1,1,1] must deliver (+ i)
    int bad = 0;
    for (int i = 0; i < iters; ++i) {
        int x0, x1, x2, x3, y0, y1, y2, y3;
        if (WAIT == 2)
            asm volatile("v_add_u32 %0, %8, %9\n\tv_add_u32 %1, %8, %10\n\tv_mov_b32_dpp %4, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32 %2, %8, %11\n\tv_mov_b32_dpp %5, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32 %3, %8, %12\n\tv_mov_b32_dpp %6, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1\n\tv_mov_b32_dpp %7, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
                         : "v"(L), "v"(4 * i), "v"(4 * i + 1), "v"(4 * i + 2), "v"(4 * i + 3));
        else if (WAIT == 3)
            asm volatile("v_add_u32 %0, %8, %9\n\tv_add_u32 %1, %8, %10\n\tv_add_u32 %2, %8, %11\n\tv_mov_b32_dpp %4, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32 %3, %8, %12\n\tv_mov_b32_dpp %5, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 0\n\tv_mov_b32_dpp %6, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 0\n\tv_mov_b32_dpp %7, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
                         : "v"(L), "v"(4 * i), "v"(4 * i + 1), "v"(4 * i + 2), "v"(4 * i + 3));
        else
            asm volatile("v_add_u32 %0, %8, %9\n\tv_add_u32 %1, %8, %10\n\tv_add_u32 %2, %8, %11\n\tv_add_u32 %3, %8, %12\n\ts_nop 5\n\t"
                         "v_mov_b32_dpp %4, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %5, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %6, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %7, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
                         : "v"(L), "v"(4 * i), "v"(4 * i + 1), "v"(4 * i + 2), "v"(4 * i + 3));
        bad += (y0 != Lq + 4 * i) + (y1 != Lq + 4 * i + 1) + (y2 != Lq + 4 * i + 2) + (y3 != Lq + 4 * i + 3);
    }
    report(err, bad, lane);
}

// pattern 3: the C++ idiom of the ray kernel's quad-shared gather: lane q of a quad computes the bilinear taps of plane min(q, 2),
// the quad shares them by quad_perm DPP, every lane also computes all three planes' taps itself and compares
struct Tap { int idx; float w; };
__device__ __forceinline__ void plane_taps(float u, float v, int H, int W, int plane_base4, Tap t[4])
{
    const float ix = ((u + 1.0f) * (float)W - 1.0f) * 0.5f, iy = ((v + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx1 = ix - x0f, fy1 = iy - y0f, fx0 = (x0f + 1.0f) - ix, fy0 = (y0f + 1.0f) - iy;
    const float xc = fminf(fmaxf(x0f, -2.0f), (float)W + 1.0f), yc = fminf(fmaxf(y0f, -2.0f), (float)H + 1.0f);
    const int x0 = (int)xc, y0 = (int)yc;
    const bool okc = (xc == x0f) && (yc == y0f);
    const bool vx0 = okc && x0 >= 0 && x0 < W, vx1 = okc && x0 + 1 >= 0 && x0 + 1 < W;
    const bool vy0 = okc && y0 >= 0 && y0 < H, vy1 = okc && y0 + 1 >= 0 && y0 + 1 < H;
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1), ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
    const int ra = __mul24(ya, W), rb = __mul24(yb, W);
    t[0].idx = plane_base4 + (ra + xa) * 8; t[0].w = (vx0 && vy0) ? fx0 * fy0 : 0.0f;
    t[1].idx = plane_base4 + (ra + xb) * 8; t[1].w = (vx1 && vy0) ? fx1 * fy0 : 0.0f;
    t[2].idx = plane_base4 + (rb + xa) * 8; t[2].w = (vx0 && vy1) ? fx0 * fy1 : 0.0f;
    t[3].idx = plane_base4 + (rb + xb) * 8; t[3].w = (vx1 && vy1) ? fx1 * fy1 : 0.0f;
}
// V bits: 1 = positions without transcendental ops; 2 = no DPP (each lane compares the select-computed taps of plane min(q,2) with its own
// taps of that plane); 4 = categories only for idx; extra error slots: err[8] idx mismatches, err[9] weight mismatches, err[10+p] plane p,
// err[13+k] tap k
template <int V>
__global__ __launch_bounds__(256) void victim_quad_taps(unsigned* err, int iters, int salt, int H, int W)
{
    const int lane = threadIdx.x & 63, q = lane & 3, s = lane >> 2;
    const int HW8 = H * W * 8;
    int bad = 0, bad_idx = 0, bad_w = 0, bad_p[3] = {0, 0, 0}, bad_k[4] = {0, 0, 0, 0};
    float rec_s = 0.f, rec_o = 0.f, rec_u = 0.f, rec_v = 0.f; int rec_p = 0;
    float t = 0.001f * (float)(blockIdx.x + salt);
    unsigned h = (blockIdx.x * 64 + s) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        float qx, qy, qz;
        if (V & 1) {
            h = h * 1664525u + 1013904223u;
            qx = (float)((int)(h >> 8) & 0xFFFF) * (2.1f / 65536.f) - 1.05f; qy = (float)((int)(h >> 4) & 0xFFFF) * (2.1f / 65536.f) - 1.05f;
            qz = (float)((int)(h >> 14) & 0xFFFF) * (2.1f / 65536.f) - 1.05f;
        } else {
            t += 0.000137f;
            qx = __sinf(t + 0.37f * s) * 1.05f; qy = __cosf(1.7f * t + 0.11f * s) * 1.05f; qz = __sinf(2.3f * t - 0.05f * s) * 1.05f;   // same inside a quad
        }
        const float us[3] = {qx, qx, qz}, vs[3] = {qy, qz, qx};
        const int pq = q < 2 ? q : 2;
        Tap tq[4];
        plane_taps(pq == 2 ? qz : qx, pq == 0 ? qy : (pq == 1 ? qz : qx), H, W, pq * HW8, tq);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            Tap ts[4], tr[4];
            plane_taps(us[p], vs[p], H, W, p * HW8, tr);
            if (V & 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { ts[k].idx = pq == p ? tq[k].idx : tr[k].idx; ts[k].w = pq == p ? tq[k].w : tr[k].w; }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ts[k].idx = p == 0 ? __builtin_amdgcn_mov_dpp(tq[k].idx, 0x00, 0xF, 0xF, true) : p == 1 ? __builtin_amdgcn_mov_dpp(tq[k].idx, 0x55, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(tq[k].idx, 0xAA, 0xF, 0xF, true);
                    const int wi = __builtin_bit_cast(int, tq[k].w);
                    ts[k].w = __builtin_bit_cast(float, p == 0 ? __builtin_amdgcn_mov_dpp(wi, 0x00, 0xF, 0xF, true) : p == 1 ? __builtin_amdgcn_mov_dpp(wi, 0x55, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(wi, 0xAA, 0xF, 0xF, true));
                }
            }
            int f = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int fi = ts[k].idx ^ tr[k].idx, fw = __float_as_int(ts[k].w) ^ __float_as_int(tr[k].w);
                f |= fi | fw; bad_idx += fi != 0; bad_w += fw != 0; bad_k[k] += (fi | fw) != 0;
                if (V & 8) { const bool m = fw != 0 && k == 2; rec_s = m ? ts[k].w : rec_s; rec_o = m ? tr[k].w : rec_o; rec_u = m ? us[p] : rec_u; rec_v = m ? vs[p] : rec_v; rec_p = m ? p : rec_p; }
            }
            bad += f != 0; bad_p[p] += f != 0;
        }
    }
    report(err, bad, lane);
    if (bad) {
        atomicAdd(&err[8], (unsigned)bad_idx); atomicAdd(&err[9], (unsigned)bad_w);
        for (int p = 0; p < 3; ++p) atomicAdd(&err[10 + p], (unsigned)bad_p[p]);
        for (int k = 0; k < 4; ++k) atomicAdd(&err[13 + k], (unsigned)bad_k[k]);
        if (V & 8) {
            const unsigned slot = atomicAdd(&err[20], 1u);
            if (slot < 12) { unsigned* r = err + 24 + 8 * slot; r[0] = lane; r[1] = rec_p * 4 + 2; r[2] = __float_as_int(rec_s); r[3] = __float_as_int(rec_o);
                             r[4] = __float_as_int(rec_u); r[5] = __float_as_int(rec_v); r[6] = 0; r[7] = 0; }
        }
    }
}

// pattern 4: VMEM address registers overwritten by the VALU right after the load is issued (WAR on the address operand)
__global__ __launch_bounds__(256) void victim_vmem_war(unsigned* err, int iters, int salt, const int* __restrict__ table, int mask)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const int idx = (h >> 8) & mask;
        int v, off = idx * 4;
        // saddr form: the 32-bit offset VGPR is overwritten by the very next instruction
        asm volatile("global_load_dword %0, %1, %2\n\tv_mov_b32 %1, 0\n\ts_waitcnt vmcnt(0)" : "=&v"(v), "+v"(off) : "s"(table) : "memory");
        bad += v != (idx ^ 0x5A5A5A5A);
    }
    report(err, bad, lane);
}

// pattern 5: DPP row scan as the shipped ray kernel's march uses it (row_shr:1,2,4,8 + row_bcast:15 / :31), checked against __shfl_up scan
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, ROW_MASK, 0xF, false); }
__global__ __launch_bounds__(256) void victim_dpp_scan(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const int x = (h >> 12) & 0xFFFF;
        int v = x;
        v += dpp_i<0x111>(0, v); v += dpp_i<0x112>(0, v); v += dpp_i<0x114>(0, v); v += dpp_i<0x118>(0, v);
        v += dpp_i<0x142, 0xA>(0, v); v += dpp_i<0x143, 0xC>(0, v);
        int r = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(r, d); if (lane >= d) r += o; }
        bad += v != r;
    }
    report(err, bad, lane);
}

// pattern 6: lane-dependent v_cndmask selects on long-lived SGPR lane masks + quad_perm share (selects only, values from registers)
__global__ __launch_bounds__(256) void victim_select_share(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63, q = lane & 3, s = lane >> 2;
    const int pq = q < 2 ? q : 2;
    int bad = 0;
    unsigned h = (blockIdx.x * 64 + s) * 2654435761u + salt;      // same inside a quad
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const int a = h >> 3, b = h ^ 0x3C3C3C3C, c = h * 3 + 1;
        const int mine = pq == 0 ? a : (pq == 1 ? b : c);
        const int s0 = __builtin_amdgcn_mov_dpp(mine, 0x00, 0xF, 0xF, true), s1 = __builtin_amdgcn_mov_dpp(mine, 0x55, 0xF, 0xF, true), s2 = __builtin_amdgcn_mov_dpp(mine, 0xAA, 0xF, 0xF, true);
        bad += (s0 != a) + (s1 != b) + (s2 != c);
    }
    report(err, bad, lane);
}

// pattern 7: the ray kernel's one-plane-in-flight gather (csrc/r3d_render.hip gather_sample) evaluated both ways -- every lane computing all
// three planes' taps itself (the shipped kernel) and the quad-shared taps (the -DR3D_ABLATE=4096 experiment) -- on a real channel-last plane
// buffer; the 8 gathered channels must agree bit for bit.
template <bool SHARED>
__device__ __forceinline__ void gather_sample(const float4* __restrict__ planes4, int H, int W, int q, float qx, float qy, float qz, float x[8])
{
    const int HW8 = H * W * 8;
    const float us[3] = {qx, qx, qz}, vs[3] = {qy, qz, qx};
    float acc[3][8];
    const float4* __restrict__ pl = planes4 + 2 * q;
    const int pq = q < 2 ? q : 2;
    Tap tq[4];
    if (SHARED) plane_taps(pq == 2 ? qz : qx, pq == 0 ? qy : (pq == 1 ? qz : qx), H, W, pq * HW8, tq);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        Tap t[4];
        if (SHARED) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k].idx = p == 0 ? __builtin_amdgcn_mov_dpp(tq[k].idx, 0x00, 0xF, 0xF, true) : p == 1 ? __builtin_amdgcn_mov_dpp(tq[k].idx, 0x55, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(tq[k].idx, 0xAA, 0xF, 0xF, true);
                const int wi = __builtin_bit_cast(int, tq[k].w);
                t[k].w = __builtin_bit_cast(float, p == 0 ? __builtin_amdgcn_mov_dpp(wi, 0x00, 0xF, 0xF, true) : p == 1 ? __builtin_amdgcn_mov_dpp(wi, 0x55, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(wi, 0xAA, 0xF, 0xF, true));
            }
        } else plane_taps(us[p], vs[p], H, W, p * HW8, t);
        float4 lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = pl[t[i].idx]; hi[i] = pl[t[i].idx + 1]; }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[p][c] = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float w = t[k].w;
            const float4 a = lo[k], b = hi[k];
            acc[p][0] += a.x * w; acc[p][1] += a.y * w; acc[p][2] += a.z * w; acc[p][3] += a.w * w;
            acc[p][4] += b.x * w; acc[p][5] += b.y * w; acc[p][6] += b.z * w; acc[p][7] += b.w * w;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = (acc[0][c] + acc[1][c] + acc[2][c]) * (1.0f / 3.0f);
}
__global__ __launch_bounds__(256, 2) void victim_gather(unsigned* err, int iters, int salt, const float4* __restrict__ planes4, int H, int W)
{
    const int lane = threadIdx.x & 63, q = lane & 3, s = lane >> 2;
    int bad = 0;
    float t = 0.001f * (float)(blockIdx.x + salt);
    for (int i = 0; i < iters; ++i) {
        t += 0.000137f;
        const float qx = __sinf(t + 0.37f * s) * 1.05f, qy = __cosf(1.7f * t + 0.11f * s) * 1.05f, qz = __sinf(2.3f * t - 0.05f * s) * 1.05f;
        float xa[8], xb[8];
        gather_sample<true>(planes4, H, W, q, qx, qy, qz, xa);
        __builtin_amdgcn_sched_barrier(0);
        gather_sample<false>(planes4, H, W, q, qx, qy, qz, xb);
        __builtin_amdgcn_sched_barrier(0);
        int f = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) f |= __float_as_int(xa[c]) ^ __float_as_int(xb[c]);
        bad += f != 0;
    }
    report(err, bad, lane);
}

// pattern 8 / 9: WAR on a 64-bit SGPR lane mask: v_cndmask reads s[m:m+1], the NEXT instruction is a SALU write of s[m:m+1] (the
// sequence hipcc emits all the time: "v_cndmask_b32_e64 v, 0, v, s[10:11]; s_and_b64 s[10:11], ..."); GAP = wait states in between
template <int GAP>
__global__ __launch_bounds__(256) void victim_sgpr_war(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const int a = h >> 3, b = ~a;
        int r0, r1, r2, r3; unsigned long long m;
        if (GAP == 0)
            asm volatile("s_mov_b64 %4, -1\n\ts_nop 3\n\tv_cndmask_b32_e64 %0, %5, %6, %4\n\ts_mov_b64 %4, 0\n\ts_nop 3\n\t"
                         "v_cndmask_b32_e64 %1, %5, %6, %4\n\ts_mov_b64 %4, -1\n\ts_nop 3\n\t"
                         "v_cndmask_b32_e64 %2, %5, %6, %4\n\ts_not_b64 %4, %4\n\ts_nop 3\n\t"
                         "v_cndmask_b32_e64 %3, %5, %6, %4\n\ts_mov_b64 %4, -1"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&s"(m) : "v"(a), "v"(b) : "scc");
        else
            asm volatile("s_mov_b64 %4, -1\n\ts_nop 3\n\tv_cndmask_b32_e64 %0, %5, %6, %4\n\ts_nop 1\n\ts_mov_b64 %4, 0\n\ts_nop 3\n\t"
                         "v_cndmask_b32_e64 %1, %5, %6, %4\n\ts_nop 1\n\ts_mov_b64 %4, -1\n\ts_nop 3\n\t"
                         "v_cndmask_b32_e64 %2, %5, %6, %4\n\ts_nop 1\n\ts_not_b64 %4, %4\n\ts_nop 3\n\t"
                         "v_cndmask_b32_e64 %3, %5, %6, %4\n\ts_nop 1\n\ts_mov_b64 %4, -1"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&s"(m) : "v"(a), "v"(b) : "scc");
        bad += (r0 != b) + (r1 != a) + (r2 != b) + (r3 != a);
    }
    report(err, bad, lane);
}

// pattern 10 / 11: RAW from a VALU-written SGPR lane mask to a SALU read: v_cmp writes s[m:m+1], the NEXT instruction is a SALU that
// reads it ("v_cmp_eq_f32_e64 s[8:9], ...; s_and_b64 s[38:39], s[6:7], s[8:9]"); the copy is then used by a v_cndmask
template <int GAP>
__global__ __launch_bounds__(256) void victim_sgpr_raw(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const int a = h >> 3, b = ~a;
        const int thr = (i * 7 + salt) & 63;                    // wave-uniform threshold: mask = lanes < thr
        int r0, r1; unsigned long long m, c;
        if (GAP == 0)
            asm volatile("v_cmp_lt_u32_e64 %2, %6, %7\n\ts_mov_b64 %3, %2\n\ts_nop 3\n\tv_cndmask_b32_e64 %0, %4, %5, %3\n\t"
                         "v_cmp_ge_u32_e64 %2, %6, %7\n\ts_mov_b64 %3, %2\n\ts_nop 3\n\tv_cndmask_b32_e64 %1, %4, %5, %3"
                         : "=&v"(r0), "=&v"(r1), "=&s"(m), "=&s"(c) : "v"(a), "v"(b), "v"(lane), "v"(thr));
        else
            asm volatile("v_cmp_lt_u32_e64 %2, %6, %7\n\ts_nop 1\n\ts_mov_b64 %3, %2\n\ts_nop 3\n\tv_cndmask_b32_e64 %0, %4, %5, %3\n\t"
                         "v_cmp_ge_u32_e64 %2, %6, %7\n\ts_nop 1\n\ts_mov_b64 %3, %2\n\ts_nop 3\n\tv_cndmask_b32_e64 %1, %4, %5, %3"
                         : "=&v"(r0), "=&v"(r1), "=&s"(m), "=&s"(c) : "v"(a), "v"(b), "v"(lane), "v"(thr));
        bad += (r0 != (lane < thr ? b : a)) + (r1 != (lane >= thr ? b : a));
    }
    report(err, bad, lane);
}

// pattern 15..20: WAR between two INDEPENDENT back-to-back VALU instructions, the first a packed-f32 op: I1 reads v[b:b+1], I2 (the
// next instruction, not dependent on I1's result) overwrites v[b:b+1].  K selects the instruction pair:
//   0: pk_mul (op_sel cross) ; pk_mul (op_sel cross) -> b     (the pair hipcc emits in plane_taps)      1: the same with "s_nop 0" in between
//   2: pk_mul ; pk_mul -> b, no op_sel                       3: pk_mul (cross) ; v_mov_b32 -> b.lo and b.hi
//   4: v_mul_f32 x2 (not packed) ; pk_mul -> b               5: pk_mul (cross) ; "s_nop 1" ; pk_mul -> b
template <int K>
__global__ __launch_bounds__(256) void victim_pk_war(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        float2 a = make_float2((float)(int)(h >> 10) * 1e-3f + 1.0f, (float)(int)((h >> 5) & 0xFFFF) * 1e-2f + 3.0f);
        float2 b = make_float2((float)(int)((h >> 7) & 0xFFF) * 0.25f + 0.5f, (float)(int)((h >> 3) & 0xFFF) * 0.125f + 7.0f);
        const float2 b_in = b;
        float2 r;
        if (K == 0)
            asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]\n\tv_pk_mul_f32 %1, %2, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r), "+v"(b) : "v"(a));
        else if (K == 1)
            asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]\n\ts_nop 0\n\tv_pk_mul_f32 %1, %2, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r), "+v"(b) : "v"(a));
        else if (K == 2)
            asm volatile("v_pk_mul_f32 %0, %2, %1\n\tv_pk_mul_f32 %1, %2, %1" : "=&v"(r), "+v"(b) : "v"(a));
        else if (K == 3)
            asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]\n\tv_pk_mov_b32 %1, %2, %2 op_sel:[0,1]" : "=&v"(r), "+v"(b) : "v"(a));
        else if (K == 4)
            asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]\n\tv_pk_add_f32 %1, %2, %2" : "=&v"(r), "+v"(b) : "v"(a));
        else
            asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]\n\ts_nop 1\n\tv_pk_mul_f32 %1, %2, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r), "+v"(b) : "v"(a));
        // expected I1 result from the ORIGINAL b: cross: lo = a.hi * b.lo, hi = a.lo * b.hi; plain (K == 2): lo = a.lo * b.lo, hi = a.hi * b.hi
        const float e0 = K == 2 ? a.x * b_in.x : a.y * b_in.x, e1 = K == 2 ? a.y * b_in.y : a.x * b_in.y;
        bad += (__float_as_int(r.x) != __float_as_int(e0)) + (__float_as_int(r.y) != __float_as_int(e1));
        h += __float_as_int(b.x) & 1;            // keep I2's result alive
    }
    report(err, bad, lane);
}

// pattern 22..25: WAW on an SGPR pair: a VALU instruction writes a lane mask (v_cmp / the carry-out of v_addc_co), a LATER SALU instruction
// overwrites the same SGPR pair (ordinary register reuse: "v_cmp_.. s[6:7]; <uses>; s_and_b64 s[6:7], ..; v_cndmask .., s[6:7]"), and a VALU
// instruction then reads it.  GAP = independent VALU instructions between the VALU write and the SALU write.
template <int GAP>
__global__ __launch_bounds__(256) void victim_sgpr_waw(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const int a = h >> 3, b = ~a;
        int r0, r1, t0, t1;
        // v_cmp writes m = all ones (0 <= lane) ; [GAP independent VALU instructions] ; s_mov m = 0 ; s_nop ; v_cndmask reads m -> must select
        // src0 (a) in every lane; then the mirror image: v_cmp writes 0, s_mov writes -1, v_cndmask must select src1 (b)
        if (GAP == 0)
            asm volatile("v_cmp_le_u32_e64 s[70:71], 0, %[ln]\n\ts_mov_b64 s[70:71], 0\n\ts_nop 7\n\tv_cndmask_b32_e64 %[r0], %[a], %[b], s[70:71]\n\t"
                         "v_cmp_gt_u32_e64 s[70:71], 0, %[ln]\n\ts_mov_b64 s[70:71], -1\n\ts_nop 7\n\tv_cndmask_b32_e64 %[r1], %[a], %[b], s[70:71]"
                         : [r0] "=&v"(r0), [r1] "=&v"(r1), [t0] "=&v"(t0), [t1] "=&v"(t1) : [a] "v"(a), [b] "v"(b), [ln] "v"(lane) : "s70", "s71");
        else if (GAP == 2)
            asm volatile("v_cmp_le_u32_e64 s[70:71], 0, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\ts_mov_b64 s[70:71], 0\n\ts_nop 7\n\tv_cndmask_b32_e64 %[r0], %[a], %[b], s[70:71]\n\t"
                         "v_cmp_gt_u32_e64 s[70:71], 0, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\ts_mov_b64 s[70:71], -1\n\ts_nop 7\n\tv_cndmask_b32_e64 %[r1], %[a], %[b], s[70:71]"
                         : [r0] "=&v"(r0), [r1] "=&v"(r1), [t0] "=&v"(t0), [t1] "=&v"(t1) : [a] "v"(a), [b] "v"(b), [ln] "v"(lane) : "s70", "s71");
        else
            asm volatile("v_cmp_le_u32_e64 s[70:71], 0, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\ts_mov_b64 s[70:71], 0\n\ts_nop 7\n\tv_cndmask_b32_e64 %[r0], %[a], %[b], s[70:71]\n\t"
                         "v_cmp_gt_u32_e64 s[70:71], 0, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\tv_add_u32 %[t0], 1, %[ln]\n\tv_add_u32 %[t1], 2, %[ln]\n\ts_mov_b64 s[70:71], -1\n\ts_nop 7\n\tv_cndmask_b32_e64 %[r1], %[a], %[b], s[70:71]"
                         : [r0] "=&v"(r0), [r1] "=&v"(r1), [t0] "=&v"(t0), [t1] "=&v"(t1) : [a] "v"(a), [b] "v"(b), [ln] "v"(lane) : "s70", "s71");
        bad += (r0 != a) + (r1 != b);
        if (GAP) h += (t0 ^ t1) & 1;
    }
    report(err, bad, lane);
}

// pattern 25..: RAW through an SGPR pair between two VALU instructions: v_cmp writes a lane mask, a VALU instruction reads it WAIT wait
// states later -- as a v_cndmask mask and as a data operand.  LLVM's GCNHazardRecognizer enforces 2 wait states here on gfx940 / gfx950
// (VALUWriteSGPRVALUReadWaitstates, hasVDecCoExecHazard).  The mask alternates between all ones and zero so that a stale read shows.
template <int WAIT>
__global__ __launch_bounds__(256) void victim_sgpr_valu_raw(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const int a = h >> 3, b = ~a;
        int r0, r1, d0, d1;
#define R3D_RAW(NOPS) asm volatile("v_cmp_le_u32_e64 s[70:71], 0, %[ln]\n\t" NOPS "v_cndmask_b32_e64 %[r0], %[a], %[b], s[70:71]\n\ts_nop 7\n\t" \
                                   "v_cmp_gt_u32_e64 s[70:71], 0, %[ln]\n\t" NOPS "v_cndmask_b32_e64 %[r1], %[a], %[b], s[70:71]\n\ts_nop 7\n\t" \
                                   "v_cmp_le_u32_e64 s[70:71], 0, %[ln]\n\t" NOPS "v_lshrrev_b32_e64 %[d0], 16, s71\n\ts_nop 7\n\t" \
                                   "v_cmp_gt_u32_e64 s[70:71], 0, %[ln]\n\t" NOPS "v_lshrrev_b32_e64 %[d1], 16, s71\n\ts_nop 7" \
                                   : [r0] "=&v"(r0), [r1] "=&v"(r1), [d0] "=&v"(d0), [d1] "=&v"(d1) : [a] "v"(a), [b] "v"(b), [ln] "v"(lane) : "s70", "s71")
        if (WAIT == 0) R3D_RAW("");
        else if (WAIT == 1) R3D_RAW("s_nop 0\n\t");
        else if (WAIT == 2) R3D_RAW("s_nop 1\n\t");
        else if (WAIT == 3) R3D_RAW("s_nop 2\n\t");
        else if (WAIT == 4) R3D_RAW("s_nop 3\n\t");
        else if (WAIT == 6) R3D_RAW("s_nop 5\n\t");
        else if (WAIT == 8) R3D_RAW("s_nop 7\n\t");
        else if (WAIT == 16) R3D_RAW("s_nop 15\n\t");
        else R3D_RAW("s_nop 15\n\ts_nop 15\n\t");
#undef R3D_RAW
        bad += (r0 != b) + (r1 != a) + (d0 != 0xFFFF) + (d1 != 0);
    }
    report(err, bad, lane);
}

// pattern 34..37: ONE instruction: an IN-PLACE packed-f32 multiply whose halves are crossed by op_sel, so that each half reads the register
// the other half writes:   v_pk_mul_f32 v[b:b+1], v[a:a+1], v[b:b+1] op_sel:[0,1] op_sel_hi:[1,0]   ->  b.lo' = a.lo * b.hi, b.hi' = a.hi * b.lo
// (hipcc emits exactly this in bilinear-weight code: fx1*fy0 / fx0*fy1 from the (fx, fy) pairs).  K: 0 in place crossed; 1 crossed, not in
// place; 2 in place, not crossed; 3 in-place crossed v_pk_fma_f32 (+0); 4 in-place crossed v_pk_add_f32
template <int K>
__global__ __launch_bounds__(256) void victim_pk_inplace(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63;
    int bad = 0, bad_lo = 0, bad_hi = 0;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const float2 a = make_float2((float)(int)(h >> 10) * 1e-3f + 1.0f, (float)(int)((h >> 5) & 0xFFFF) * 1e-2f + 3.0f);
        const float2 b_in = make_float2((float)(int)((h >> 7) & 0xFFF) * 0.25f + 0.5f, (float)(int)((h >> 3) & 0xFFF) * 0.125f + 7.0f);
        float2 b = b_in, r = b_in, z = make_float2(0.f, 0.f);
        float e0, e1;
        if (K == 0) { asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(b) : "v"(a)); r = b; e0 = a.x * b_in.y; e1 = a.y * b_in.x; }
        else if (K == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.x * b_in.y; e1 = a.y * b_in.x; }
        else if (K == 2) { asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(b) : "v"(a)); r = b; e0 = a.x * b_in.x; e1 = a.y * b_in.y; }
        else if (K == 3) { asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(b) : "v"(a), "v"(z)); r = b; e0 = a.x * b_in.y; e1 = a.y * b_in.x; }
        else { asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(b) : "v"(a)); r = b; e0 = a.x + b_in.y; e1 = a.y + b_in.x; }
        const int f0 = __float_as_int(r.x) != __float_as_int(e0), f1 = __float_as_int(r.y) != __float_as_int(e1);
        bad += f0 | f1; bad_lo += f0; bad_hi += f1;
    }
    report(err, bad, lane);
    if (bad) { atomicAdd(&err[8], (unsigned)bad_lo); atomicAdd(&err[9], (unsigned)bad_hi); }
}

// pattern 39..44: the crossed packed multiply (not in place) with WAIT wait states between the producers of its operands and the instruction
// (is it a forwarding hazard?) and with different op_sel forms.  K: 0..3 = op_sel:[0,1] op_sel_hi:[1,0] after s_nop {0,1,3,15};
// 4 = op_sel:[1,0] op_sel_hi:[1,1] (src0 crossed for the low half only); 5 = op_sel_hi:[0,0] (both halves read the LOW dwords: broadcast)
template <int K>
__global__ __launch_bounds__(256) void victim_pk_opsel(unsigned* err, int iters, int salt)
{
    const int lane = threadIdx.x & 63;
    int bad = 0, bad_lo = 0, bad_hi = 0;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const float2 a = make_float2((float)(int)(h >> 10) * 1e-3f + 1.0f, (float)(int)((h >> 5) & 0xFFFF) * 1e-2f + 3.0f);
        const float2 b = make_float2((float)(int)((h >> 7) & 0xFFF) * 0.25f + 0.5f, (float)(int)((h >> 3) & 0xFFF) * 0.125f + 7.0f);
        float2 r; float e0, e1;
        if (K == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.x * b.y; e1 = a.y * b.x; }
        else if (K == 1) { asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.x * b.y; e1 = a.y * b.x; }
        else if (K == 2) { asm volatile("s_nop 3\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.x * b.y; e1 = a.y * b.x; }
        else if (K == 3) { asm volatile("s_nop 15\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 15" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.x * b.y; e1 = a.y * b.x; }
        else if (K == 4) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.y * b.x; e1 = a.y * b.y; }
        else if (K == 5) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,0]" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.x * b.x; e1 = a.x * b.x; }
        else if (K == 6) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.y * b.x; e1 = a.x * b.y; }
        else if (K == 7) { asm volatile("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=&v"(r) : "v"(a), "v"(b)); e0 = __builtin_fmaf(a.x, a.x, b.y); e1 = __builtin_fmaf(a.y, a.y, b.x); }
        else if (K == 8) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.x; e1 = b.x; }
        else if (K == 9) { float2 bs = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__float_as_int(b.x))), __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__float_as_int(b.y))));
                           asm volatile("s_nop 3\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(r) : "v"(a), "s"(bs)); e0 = a.x * bs.y; e1 = a.y * bs.y; }
        else if (K == 10) { float2 bs = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__float_as_int(b.x))), __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__float_as_int(b.y))));
                           asm volatile("s_nop 3\n\tv_pk_mul_f32 %0, %2, %1 op_sel:[1,0]" : "=&v"(r) : "v"(a), "s"(bs)); e0 = a.x * bs.y; e1 = a.y * bs.y; }
        else { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=&v"(r) : "v"(a), "v"(b)); e0 = a.y * b.y; e1 = a.x * b.x; }
        const int f0 = __float_as_int(r.x) != __float_as_int(e0), f1 = __float_as_int(r.y) != __float_as_int(e1);
        bad += f0 | f1; bad_lo += f0; bad_hi += f1;
    }
    report(err, bad, lane);
    if (bad) { atomicAdd(&err[8], (unsigned)bad_lo); atomicAdd(&err[9], (unsigned)bad_hi); }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int agg_mode_sel = argc > 2 ? atoi(argv[2]) : 29;
    hipStream_t sa, sv;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    const int src_elems = 1 << 18;
    std::vector<_Float16> hsrc((size_t)src_elems * 8);
    for (size_t i = 0; i < hsrc.size(); ++i) hsrc[i] = (_Float16)(0.01f * (float)((int)(i * 2654435761u >> 20) % 200 - 100));
    uint4* dsrc; float* dout; unsigned* derr; int* dtab;
    CK(hipMalloc(&dsrc, (size_t)src_elems * 16)); CK(hipMemcpy(dsrc, hsrc.data(), (size_t)src_elems * 16, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, 1024 * 512 * 4)); CK(hipMalloc(&derr, 256 * sizeof(unsigned)));
    const int tmask = (1 << 20) - 1;
    std::vector<int> htab(tmask + 1); for (int i = 0; i <= tmask; ++i) htab[i] = i ^ 0x5A5A5A5A;
    CK(hipMalloc(&dtab, (tmask + 1) * 4)); CK(hipMemcpy(dtab, htab.data(), (tmask + 1) * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)agg_kernel<29>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void*)agg_kernel<25>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void*)agg_kernel<17>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void*)agg_kernel<49>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    const char* names[] = {"VALU -> quad_perm DPP read, 2 wait states", "VALU -> quad_perm DPP read, 3 wait states", "VALU -> quad_perm DPP read, 6+ wait states",
                           "C++ quad-shared bilinear taps vs own taps", "global_load, address VGPR reuse", "DPP row scan (row_shr / row_bcast) vs shfl scan",
                           "v_cndmask lane selects + quad_perm share", "full gather: quad-shared taps vs own taps (8 channels)",
                           "v_cndmask reads SGPR mask; SALU overwrites it NEXT (WAR)", "same, 2 wait states in between",
                           "v_cmp writes SGPR mask; SALU reads it NEXT (RAW)", "same, 2 wait states in between",
                           "quad taps, positions without transcendentals", "quad taps, NO DPP (select-computed vs own taps)", "quad taps, no DPP, no transcendentals",
                           "pk_mul(cross) ; pk_mul(cross) overwrites its source", "same + s_nop 0 between", "pk_mul ; pk_mul overwrites source (no op_sel)",
                           "pk_mul(cross) ; v_pk_mov_b32 overwrites source", "pk_mul(cross) ; v_pk_add_f32 overwrites source", "pk_mul(cross) ; s_nop 1 ; pk_mul overwrites source", "quad taps + branch-free record of a mismatching tap-2 weight",
                           "v_cmp writes SGPR mask; SALU overwrites it NEXT (WAW); VALU reads", "same, 2 VALU instructions before the SALU write", "same, 8 VALU instructions before the SALU write",
                           "v_cmp -> SGPR -> VALU read, 0 wait states", "v_cmp -> SGPR -> VALU read, 1 wait state", "v_cmp -> SGPR -> VALU read, 2 wait states (LLVM's rule)",
                           "v_cmp -> SGPR -> VALU read, 3 wait states", "v_cmp -> SGPR -> VALU read, 4 wait states", "v_cmp -> SGPR -> VALU read, 6 wait states",
                           "v_cmp -> SGPR -> VALU read, 8 wait states", "v_cmp -> SGPR -> VALU read, 16 wait states", "v_cmp -> SGPR -> VALU read, 32 wait states",
                           "ONE in-place crossed v_pk_mul_f32", "crossed v_pk_mul_f32, not in place", "in-place v_pk_mul_f32, not crossed",
                           "in-place crossed v_pk_fma_f32", "in-place crossed v_pk_add_f32",
                           "crossed pk_mul, 0 wait states after its producers", "crossed pk_mul, s_nop 0 first", "crossed pk_mul, s_nop 3 first", "crossed pk_mul between s_nop 15",
                           "pk_mul op_sel:[1,0] op_sel_hi:[1,1] (src0 hi -> low half)", "pk_mul op_sel_hi:[0,0] (broadcast of the low dwords)",
                           "pk_mul op_sel:[1,0] op_sel_hi:[0,1] (src0 crossed both ways)", "pk_fma op_sel:[0,0,1] (src2 hi -> low half)", "v_pk_mov_b32 op_sel:[0,1] op_sel_hi:[1,0]",
                           "pk_mul v, v, SGPR pair op_sel:[0,1]", "pk_mul v, SGPR pair, v op_sel:[1,0] (the swapped form)", "pk_mul op_sel:[1,1] op_sel_hi:[0,0]"};
    float4* dplanes; const int PH = 256, PW = 256;
    {
        std::vector<float> hp((size_t)3 * PH * PW * 32);
        unsigned hh = 12345u;
        for (size_t i = 0; i < hp.size(); ++i) { hh = hh * 1664525u + 1013904223u; hp[i] = (float)((int)(hh >> 9) % 2001 - 1000) * 1e-3f; }
        CK(hipMalloc(&dplanes, hp.size() * 4)); CK(hipMemcpy(dplanes, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    }
    const int grid_v = 2048;
    for (int with_agg = 0; with_agg < 2; ++with_agg) {
        printf("== victims %s\n", with_agg ? "next to the MFMA aggressor (other stream)" : "alone");
        const int only = argc > 3 ? atoi(argv[3]) : -1;
        for (int pat = 0; pat < 51; ++pat) {
            if (only >= 0 && pat != only && !(only == 99 && (pat == 3 || (pat >= 12 && pat < 15) || pat == 21)) && !(only == 98 && pat >= 15 && pat < 21) && !(only == 97 && pat >= 22 && pat < 25) && !(only == 96 && pat >= 25 && pat < 34) && !(only == 95 && pat >= 34 && pat < 39) && !(only == 94 && pat >= 39 && pat < 45) && !(only == 93 && pat >= 45)) continue;
            CK(hipMemsetAsync(derr, 0, 256 * sizeof(unsigned), sv)); CK(hipStreamSynchronize(sv));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            if (with_agg)
                for (int k = 0; k < 40; ++k) {
                    const size_t lds = 76 * 1024;
                    if (agg_mode_sel == 25) hipLaunchKernelGGL((agg_kernel<25>), dim3(1024), dim3(512), lds, sa, dsrc, dout, 150, src_elems - 1);
                    else if (agg_mode_sel == 17) hipLaunchKernelGGL((agg_kernel<17>), dim3(1024), dim3(512), lds, sa, dsrc, dout, 150, src_elems - 1);
                    else if (agg_mode_sel == 49) hipLaunchKernelGGL((agg_kernel<49>), dim3(1024), dim3(512), lds, sa, dsrc, dout, 150, src_elems - 1);
                    else hipLaunchKernelGGL((agg_kernel<29>), dim3(1024), dim3(512), lds, sa, dsrc, dout, 150, src_elems - 1);
                }
            CK(hipEventRecord(e0, sv));
            for (int rep = 0; rep < 6; ++rep) {
                switch (pat) {
                    case 0: hipLaunchKernelGGL((victim_raw_dpp<2>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 1: hipLaunchKernelGGL((victim_raw_dpp<3>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 2: hipLaunchKernelGGL((victim_raw_dpp<6>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 3: hipLaunchKernelGGL((victim_quad_taps<0>), dim3(grid_v), dim3(256), 0, sv, derr, iters / 8, rep, 256, 256); break;
                    case 12: hipLaunchKernelGGL((victim_quad_taps<1>), dim3(grid_v), dim3(256), 0, sv, derr, iters / 8, rep, 256, 256); break;
                    case 13: hipLaunchKernelGGL((victim_quad_taps<2>), dim3(grid_v), dim3(256), 0, sv, derr, iters / 8, rep, 256, 256); break;
                    case 21: hipLaunchKernelGGL((victim_quad_taps<9>), dim3(grid_v), dim3(256), 0, sv, derr, iters / 8, rep, 256, 256); break;
                    case 14: hipLaunchKernelGGL((victim_quad_taps<3>), dim3(grid_v), dim3(256), 0, sv, derr, iters / 8, rep, 256, 256); break;
                    case 4: hipLaunchKernelGGL(victim_vmem_war, dim3(grid_v), dim3(256), 0, sv, derr, iters / 8, rep, dtab, tmask); break;
                    case 5: hipLaunchKernelGGL(victim_dpp_scan, dim3(grid_v), dim3(256), 0, sv, derr, iters / 4, rep); break;
                    case 6: hipLaunchKernelGGL(victim_select_share, dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 8: hipLaunchKernelGGL((victim_sgpr_war<0>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 9: hipLaunchKernelGGL((victim_sgpr_war<2>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 10: hipLaunchKernelGGL((victim_sgpr_raw<0>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 11: hipLaunchKernelGGL((victim_sgpr_raw<2>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 15: hipLaunchKernelGGL((victim_pk_war<0>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 16: hipLaunchKernelGGL((victim_pk_war<1>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 17: hipLaunchKernelGGL((victim_pk_war<2>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 18: hipLaunchKernelGGL((victim_pk_war<3>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 19: hipLaunchKernelGGL((victim_pk_war<4>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 20: hipLaunchKernelGGL((victim_pk_war<5>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 22: hipLaunchKernelGGL((victim_sgpr_waw<0>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 23: hipLaunchKernelGGL((victim_sgpr_waw<2>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 24: hipLaunchKernelGGL((victim_sgpr_waw<8>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 25: hipLaunchKernelGGL((victim_sgpr_valu_raw<0>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 26: hipLaunchKernelGGL((victim_sgpr_valu_raw<1>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 27: hipLaunchKernelGGL((victim_sgpr_valu_raw<2>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 28: hipLaunchKernelGGL((victim_sgpr_valu_raw<3>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 29: hipLaunchKernelGGL((victim_sgpr_valu_raw<4>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 30: hipLaunchKernelGGL((victim_sgpr_valu_raw<6>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 31: hipLaunchKernelGGL((victim_sgpr_valu_raw<8>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 32: hipLaunchKernelGGL((victim_sgpr_valu_raw<16>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 33: hipLaunchKernelGGL((victim_sgpr_valu_raw<32>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 34: hipLaunchKernelGGL((victim_pk_inplace<0>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 35: hipLaunchKernelGGL((victim_pk_inplace<1>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 36: hipLaunchKernelGGL((victim_pk_inplace<2>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 37: hipLaunchKernelGGL((victim_pk_inplace<3>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 38: hipLaunchKernelGGL((victim_pk_inplace<4>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 39: hipLaunchKernelGGL((victim_pk_opsel<0>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 40: hipLaunchKernelGGL((victim_pk_opsel<1>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 41: hipLaunchKernelGGL((victim_pk_opsel<2>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 42: hipLaunchKernelGGL((victim_pk_opsel<3>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 43: hipLaunchKernelGGL((victim_pk_opsel<4>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 44: hipLaunchKernelGGL((victim_pk_opsel<5>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 45: hipLaunchKernelGGL((victim_pk_opsel<6>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 46: hipLaunchKernelGGL((victim_pk_opsel<7>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 47: hipLaunchKernelGGL((victim_pk_opsel<8>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 48: hipLaunchKernelGGL((victim_pk_opsel<9>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 49: hipLaunchKernelGGL((victim_pk_opsel<10>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 50: hipLaunchKernelGGL((victim_pk_opsel<11>), dim3(grid_v), dim3(256), 0, sv, derr, iters, rep); break;
                    case 7: hipLaunchKernelGGL(victim_gather, dim3(grid_v), dim3(256), 0, sv, derr, iters / 16, rep, dplanes, PH, PW); break;
                }
            }
            CK(hipEventRecord(e1, sv));
            CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr[20]; CK(hipMemcpy(herr, derr, sizeof(herr), hipMemcpyDeviceToHost));
            printf("  [%d] %-52s mismatches %10u   by DPP row: %u %u %u %u   (%.1f ms)\n", pat, names[pat], herr[0], herr[1], herr[2], herr[3], herr[4], ms);
            if (herr[8] | herr[9]) printf("        [8] %u  [9] %u | plane %u %u %u | tap %u %u %u %u\n", herr[8], herr[9], herr[10], herr[11], herr[12], herr[13], herr[14], herr[15], herr[16]);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
    }
    return 0;
}
