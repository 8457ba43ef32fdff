// Error handling, version, event helpers and the output-side conversion kernel of libr3d_hip.so.
#include <stdarg.h>
#include <stdio.h>

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
static uint32_t g_prof_mask = 0;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev[R3D_PROF_COUNT];
static size_t g_prof_used[R3D_PROF_COUNT] = {0};

void prof_begin(int id, hipStream_t st)
{
    if (!(g_prof_mask & (1u << id))) return;
    if (g_prof_used[id] == g_prof_ev[id].size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        g_prof_ev[id].push_back(std::make_pair(a, b));
    }
    (void)hipEventRecord(g_prof_ev[id][g_prof_used[id]].first, st);
}
void prof_end(int id, hipStream_t st)
{
    if (!(g_prof_mask & (1u << id))) return;
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
    for (int c = 0; c < 3; ++c) {
        float v = s[(size_t)c * HW + p];
        v = fminf(fmaxf(v, -1.0f), 1.0f);
        d[c] = (uint8_t)((v + 1.0f) * 127.5f);     // ((x+1)/2*255).to(uint8): truncation
    }
}

}  // namespace r3d

using namespace r3d;

extern "C" int r3d_version(void) { return 10; }   // 0.1.0
extern "C" const char* r3d_last_error(void) { return g_err; }

extern "C" int r3d_frames_to_u8(const float* img, int N, int H, int W, uint8_t* out, r3d_stream_t stream)
{
    if (!img || !out || N <= 0 || H <= 0 || W <= 0) { set_error("frames_to_u8: bad argument"); return R3D_ERR_INVALID_ARG; }
    dim3 grid((H * W + 255) / 256, N);
    ProfScope ps(R3D_PROF_LAYOUT, (hipStream_t)stream);
    hipLaunchKernelGGL(frames_to_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, H * W, out);
    return check_launch("frames_to_u8");
}

extern "C" int r3d_profile_configure(uint32_t mask) { g_prof_mask = mask; return R3D_OK; }
extern "C" int r3d_profile_reset(void)
{
    for (int i = 0; i < R3D_PROF_COUNT; ++i) g_prof_used[i] = 0;
    return R3D_OK;
}
extern "C" int r3d_profile_read(int id, double* total_ms, int* launches)
{
    if (id < 0 || id >= R3D_PROF_COUNT || !total_ms || !launches) { set_error("profile_read: bad argument"); return R3D_ERR_INVALID_ARG; }
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
