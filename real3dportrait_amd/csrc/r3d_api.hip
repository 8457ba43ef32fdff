// Error handling, version, event helpers and the output-side conversion kernel of libr3d_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <utility>
#include <vector>

#include "r3d_common.h"

namespace r3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
        return R3D_ERR_LAUNCH;
    }
    return R3D_OK;
}

// ---- event-pair profiling of launch sites -------------------------------------------------------------
// Process-global by design (a bench tool), but safe to use from several host threads driving different streams: the state is guarded by
// a mutex and the events are created in r3d_profile_configure (a pool per family), not inside a timed region; a launch site that finds the
// pool exhausted is simply not bracketed.
static std::mutex g_prof_mu;
static uint32_t g_prof_mask = 0;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev[R3D_PROF_COUNT];
static size_t g_prof_used[R3D_PROF_COUNT] = {0};
static constexpr size_t kProfPool = 1024;       // event pairs pre-created per enabled family
static unsigned long long* g_prof_clk = nullptr;   // device: [R3D_PROF_COUNT][4] (cycles, ticks, start cycles, start ticks), see prof_clock_slot

unsigned long long* prof_clock_slot(int id)
{
    return (g_prof_clk && (g_prof_mask & (1u << id))) ? g_prof_clk + 4 * id : nullptr;
}

void prof_begin(int id, hipStream_t st)
{
    if (!(g_prof_mask & (1u << id))) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_used[id] >= g_prof_ev[id].size()) return;
    (void)hipEventRecord(g_prof_ev[id][g_prof_used[id]].first, st);
}
void prof_end(int id, hipStream_t st)
{
    if (!(g_prof_mask & (1u << id))) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_used[id] >= g_prof_ev[id].size()) return;
    (void)hipEventRecord(g_prof_ev[id][g_prof_used[id]].second, st);
    ++g_prof_used[id];
}

// clamp(-1,1) -> uint8 HWC  (inference/real3d_infer.py:495-521 does this on the host after the loop)
__global__ void frames_to_u8_kernel(const float* __restrict__ img, int HW, uint8_t* __restrict__ out)
{
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* s = img + (size_t)n * 3 * HW;
    uint8_t* d = out + ((size_t)n * HW + p) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = frame_u8(s[(size_t)c * HW + p]);
}

// ---- small image ops of SuperresolutionHybrid8XDC_Warp.forward (modules/real3d/super_resolution/sr_with_ref.py:67-137) -------------
// F.interpolate(mode='bilinear', align_corners=False, antialias=A): ATen's separable "aa" weights (UpSampleKernel.cpp
// _compute_indices_weights_aa): scale = in / out, support = max(scale, 1), centre = scale * (i + 0.5), taps
// [int(centre - support + 0.5), int(centre + support + 0.5)) clipped to the image, triangle weights (1 - |t| / max(scale, 1))
// normalised to sum 1.  antialias = 0 -> support 1 (plain bilinear with edge clamping, which the same formulas give).
__device__ __forceinline__ void aa_window(int i, float scale, int in_size, int antialias, int& lo, int& cnt, float& inv)
{
    const float sup = (antialias && scale >= 1.0f) ? scale : 1.0f;
    const float centre = scale * ((float)i + 0.5f);
    lo = max((int)(centre - sup + 0.5f), 0);
    cnt = min((int)(centre + sup + 0.5f), in_size) - lo;
    inv = (antialias && scale >= 1.0f) ? 1.0f / scale : 1.0f;
}
__device__ __forceinline__ float aa_tri(float t) { t = fabsf(t); return t < 1.0f ? 1.0f - t : 0.0f; }

__global__ void resize_bilinear_kernel(const float* __restrict__ x, int H, int W, float* __restrict__ y, int OH, int OW, int antialias)
{
    const int plane = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= OH * OW) return;
    const int oy = p / OW, ox = p - oy * OW;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    int y0, ny, x0, nx; float iy, ix;
    aa_window(oy, sy, H, antialias, y0, ny, iy);
    aa_window(ox, sx, W, antialias, x0, nx, ix);
    const float cy = sy * ((float)oy + 0.5f), cx = sx * ((float)ox + 0.5f);
    float wys = 0.f, wxs = 0.f;
    for (int j = 0; j < ny; ++j) wys += aa_tri(((float)(j + y0) - cy + 0.5f) * iy);
    for (int k = 0; k < nx; ++k) wxs += aa_tri(((float)(k + x0) - cx + 0.5f) * ix);
    const float* X = x + (size_t)plane * H * W;
    float acc = 0.f;
    for (int j = 0; j < ny; ++j) {
        const float wy = aa_tri(((float)(j + y0) - cy + 0.5f) * iy) / wys;
        float row = 0.f;
        for (int k = 0; k < nx; ++k) row += X[(size_t)(y0 + j) * W + x0 + k] * (aa_tri(((float)(k + x0) - cx + 0.5f) * ix) / wxs);
        acc += wy * row;
    }
    y[(size_t)plane * OH * OW + p] = acc;
}

// out = a * m + b * (1 - m)  (`rgb * head_torso_alpha + rgb_torso * (1 - head_torso_alpha)`, sr_with_ref.py:103,113,125,135)
__global__ void blend_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ mask, int C, int HW,
                             float* __restrict__ out)
{
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float m = mask[(size_t)n * HW + p];
    for (int c = 0; c < C; ++c) {
        const size_t i = ((size_t)n * C + c) * HW + p;
        out[i] = a[i] * m + b[i] * (1.0f - m);
    }
}

// person_occlusion = clamp(torso_occlusion + head_occlusion, 0, 1), head_occlusion = alpha with values > threshold set to 1
// (sr_with_ref.py:107-112,117-122)
__global__ void person_occlusion_kernel(const float* __restrict__ alpha, const float* __restrict__ torso_occ, float thr, size_t n,
                                        float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = alpha[i];
    out[i] = fminf(fmaxf(torso_occ[i] + (a > thr ? 1.0f : a), 0.0f), 1.0f);
}

}  // namespace r3d

using namespace r3d;

extern "C" int r3d_version(void) { return 61; }   // 0.6.1: r3d_conv_forward_blend; 0.6.0: r3d_sr_block_prepacked_bytes grew by conv1's Winograd F(2,3) pack; r3d_render_workspace_bytes follows the real grid (0.5.0: e5m2 activation records); real3dportrait_amd/_lib.py checks the number

extern "C" int r3d_resize_bilinear(const float* x, int planes, int H, int W, float* y, int OH, int OW, int antialias, r3d_stream_t stream)
{
    if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) { set_error("resize_bilinear: bad argument"); return R3D_ERR_INVALID_ARG; }
    ProfScope ps(R3D_PROF_LAYOUT, (hipStream_t)stream);
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((OH * OW + 255) / 256, planes), dim3(256), 0, (hipStream_t)stream, x, H, W, y, OH, OW, antialias);
    return check_launch("resize_bilinear");
}

extern "C" int r3d_blend(const float* a, const float* b, const float* mask, int N, int C, int H, int W, float* out, r3d_stream_t stream)
{
    if (!a || !b || !mask || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) { set_error("blend: bad argument"); return R3D_ERR_INVALID_ARG; }
    ProfScope ps(R3D_PROF_LAYOUT, (hipStream_t)stream);
    hipLaunchKernelGGL(blend_kernel, dim3((H * W + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, a, b, mask, C, H * W, out);
    return check_launch("blend");
}

extern "C" int r3d_person_occlusion(const float* alpha, const float* torso_occlusion, float head_threshold, size_t count, float* out,
                                    r3d_stream_t stream)
{
    if (!alpha || !torso_occlusion || !out || count == 0) { set_error("person_occlusion: bad argument"); return R3D_ERR_INVALID_ARG; }
    ProfScope ps(R3D_PROF_LAYOUT, (hipStream_t)stream);
    hipLaunchKernelGGL(person_occlusion_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, alpha, torso_occlusion,
                       head_threshold, count, out);
    return check_launch("person_occlusion");
}
extern "C" const char* r3d_last_error(void) { return g_err; }

extern "C" int r3d_frames_to_u8(const float* img, int N, int H, int W, uint8_t* out, r3d_stream_t stream)
{
    if (!img || !out || N <= 0 || H <= 0 || W <= 0) { set_error("frames_to_u8: bad argument"); return R3D_ERR_INVALID_ARG; }
    dim3 grid((H * W + 255) / 256, N);
    ProfScope ps(R3D_PROF_LAYOUT, (hipStream_t)stream);
    hipLaunchKernelGGL(frames_to_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, H * W, out);
    return check_launch("frames_to_u8");
}

extern "C" int r3d_profile_configure(uint32_t mask)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < R3D_PROF_COUNT; ++i)
        if (mask & (1u << i))
            while (g_prof_ev[i].size() < kProfPool) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { set_error("profile_configure: hipEventCreate failed"); return R3D_ERR_LAUNCH; }
                g_prof_ev[i].push_back(std::make_pair(a, b));
            }
    if (mask && !g_prof_clk) {
        if (hipMalloc(&g_prof_clk, sizeof(unsigned long long) * 4 * R3D_PROF_COUNT) != hipSuccess ||
            hipMemset(g_prof_clk, 0, sizeof(unsigned long long) * 4 * R3D_PROF_COUNT) != hipSuccess) { g_prof_clk = nullptr; set_error("profile_configure: clock slots"); return R3D_ERR_LAUNCH; }
    }
    g_prof_mask = mask;
    return R3D_OK;
}
extern "C" int r3d_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < R3D_PROF_COUNT; ++i) g_prof_used[i] = 0;
    if (g_prof_clk && (hipDeviceSynchronize() != hipSuccess || hipMemset(g_prof_clk, 0, sizeof(unsigned long long) * 4 * R3D_PROF_COUNT) != hipSuccess)) { set_error("profile_reset: clock slots"); return R3D_ERR_LAUNCH; }
    return R3D_OK;
}
extern "C" int r3d_profile_clock(int id, double* ghz, unsigned long long* cycles)
{
    if (id < 0 || id >= R3D_PROF_COUNT || !ghz) { set_error("profile_clock: bad argument"); return R3D_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    *ghz = 0.0; if (cycles) *cycles = 0;
    if (!g_prof_clk) return R3D_OK;
    unsigned long long h[4];
    int dev = 0, khz = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, g_prof_clk + 4 * id, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess ||
        hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) { set_error("profile_clock: query failed"); return R3D_ERR_LAUNCH; }
    if (h[1] && khz > 0) *ghz = (double)h[0] / (double)h[1] * (double)khz * 1e-6;
    if (cycles) *cycles = h[0];
    return R3D_OK;
}
extern "C" int r3d_profile_read(int id, double* total_ms, int* launches)
{
    if (id < 0 || id >= R3D_PROF_COUNT || !total_ms || !launches) { set_error("profile_read: bad argument"); return R3D_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double tot = 0.0;
    for (size_t i = 0; i < g_prof_used[id]; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof_ev[id][i].second) != hipSuccess ||
            hipEventElapsedTime(&ms, g_prof_ev[id][i].first, g_prof_ev[id][i].second) != hipSuccess) {
            set_error("profile_read: event query failed"); return R3D_ERR_LAUNCH;
        }
        tot += ms;
    }
    *total_ms = tot; *launches = (int)g_prof_used[id];
    return R3D_OK;
}

extern "C" int r3d_event_create(void** ev)
{
    hipEvent_t e;
    if (!ev || hipEventCreate(&e) != hipSuccess) { set_error("event_create failed"); return R3D_ERR_LAUNCH; }
    *ev = (void*)e;
    return R3D_OK;
}
extern "C" int r3d_event_record(void* ev, r3d_stream_t stream)
{
    if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) { set_error("event_record failed"); return R3D_ERR_LAUNCH; }
    return R3D_OK;
}
extern "C" int r3d_event_elapsed_ms(void* start, void* stop, float* ms)
{
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess || hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
        set_error("event_elapsed failed"); return R3D_ERR_LAUNCH;
    }
    return R3D_OK;
}
extern "C" int r3d_event_destroy(void* ev) { (void)hipEventDestroy((hipEvent_t)ev); return R3D_OK; }
