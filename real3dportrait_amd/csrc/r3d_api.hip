// Error handling, version, event helpers and the output-side conversion kernel of libr3d_hip.so.
#include <stdarg.h>
#include <stdio.h>

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
    hipLaunchKernelGGL(frames_to_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, H * W, out);
    return check_launch("frames_to_u8");
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
extern "C" int r3d_event_destroy(void* ev) { hipEventDestroy((hipEvent_t)ev); return R3D_OK; }
