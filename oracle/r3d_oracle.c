/*
 * r3d_oracle.c -- CPU restatement (plain C, fp32) of the Real3D-Portrait tri-plane
 * NeRF render + super-resolution hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load this library; the product path
 * (real3dportrait_amd/) never does and fails loudly without the HIP extension.
 *
 * Parity pinning: the reference repo has no tests/golden vectors for this path
 * (SURVEY.md section 4), so this restatement is pinned against outputs of the reference's
 * own PyTorch modules run on CPU in the build container (tests/golden/make_golden.py
 * imports them from /root/reference and commits the vectors; tests/test_oracle_golden.py
 * replays them against this file).
 *
 * Every function cites the reference file:line (relative to the upstream repo root)
 * whose behaviour it restates.  Nothing here is copied: the reference is Python/PyTorch,
 * this is scalar C written from the behavioural spec.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define R3D_API __attribute__((visibility("default")))

R3D_API int r3d_oracle_version(void) { return 1; }

R3D_API int r3d_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------
 * A1  ray generation
 * modules/eg3ds/volumetric_rendering/ray_sampler.py:24-63
 *   uv = (arange(R) * (1/R) + 0.5/R), x = column index (fastest), y = row index (:43-44)
 *   x_lift = (x - cx + cy*sk/fy - sk*y/fy) / fx ; y_lift = (y - cy)/fy  (:51-52)
 *   world = c2w @ [x_lift, y_lift, 1, 1]  (:54-56); dir = normalize(world - cam) (:58-59)
 *   origin = cam position broadcast (:61)
 * c2w [N,4,4] row-major, K [N,3,3] row-major; outputs [N, R*R, 3].
 * ---------------------------------------------------------------------------------- */
R3D_API void r3d_oracle_raygen(const float* c2w, const float* K, int N, int R,
                               float* origins, float* dirs)
{
    const int M = R * R;
    const float inv_r = (float)(1.0 / (double)R);
    const float half_r = (float)(0.5 / (double)R);
    for (int n = 0; n < N; ++n) {
        const float* C = c2w + 16 * n;
        const float* Kn = K + 9 * n;
        const float fx = Kn[0], sk = Kn[1], cx = Kn[2], fy = Kn[4], cy = Kn[5];
        const float camx = C[3], camy = C[7], camz = C[11];
        for (int m = 0; m < M; ++m) {
            const int i = m / R, j = m % R;
            const float xc = (float)j * inv_r + half_r;
            const float yc = (float)i * inv_r + half_r;
            const float xl = (xc - cx + cy * sk / fy - sk * yc / fy) / fx;
            const float yl = (yc - cy) / fy;
            float w[3];
            for (int r = 0; r < 3; ++r)
                w[r] = C[4 * r + 0] * xl + C[4 * r + 1] * yl + C[4 * r + 2] * 1.0f + C[4 * r + 3] * 1.0f;
            float dx = w[0] - camx, dy = w[1] - camy, dz = w[2] - camz;
            float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
            if (nrm < 1e-12f) nrm = 1e-12f; /* F.normalize eps */
            float* o = origins + 3 * ((size_t)n * M + m);
            float* d = dirs + 3 * ((size_t)n * M + m);
            o[0] = camx; o[1] = camy; o[2] = camz;
            d[0] = dx / nrm; d[1] = dy / nrm; d[2] = dz / nrm;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * A2  ray / axis-aligned box limits + the renderer's invalid-ray fix-up
 * modules/eg3ds/volumetric_rendering/math_utils.py:46-98  (slab test, invalid -> (-1,-2))
 * modules/eg3ds/volumetric_rendering/renderer.py:121-126  (valid = end > start; invalid rays
 *   get start = min(valid starts), end = max(valid STARTS) -- sic, both from ray_start)
 * nrays = N*M.  valid_out: 1 byte per ray.
 * ---------------------------------------------------------------------------------- */
static void ray_box(const float* o, const float* d, float half, float* tmin_o, float* tmax_o)
{
    float inv[3], lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        inv[a] = 1.0f / d[a];
        const int sgn = inv[a] < 0.0f;
        const float bmin = -half, bmax = half;
        lo[a] = ((sgn ? bmax : bmin) - o[a]) * inv[a];
        hi[a] = ((sgn ? bmin : bmax) - o[a]) * inv[a];
    }
    int valid = 1;
    float tmin = lo[0], tmax = hi[0];
    if (tmin > hi[1] || lo[1] > tmax) valid = 0;
    tmin = fmaxf(tmin, lo[1]);   /* torch.max propagates NaN; fmaxf does not -- NaN only for d==0 & o on slab */
    tmax = fminf(tmax, hi[1]);
    if (tmin > hi[2] || lo[2] > tmax) valid = 0;
    tmin = fmaxf(tmin, lo[2]);
    tmax = fminf(tmax, hi[2]);
    if (!valid) { tmin = -1.0f; tmax = -2.0f; }
    *tmin_o = tmin; *tmax_o = tmax;
}

R3D_API void r3d_oracle_ray_limits(const float* origins, const float* dirs, int nrays, float box_warp,
                                   float* ray_start, float* ray_end, uint8_t* valid_out)
{
    const float half = box_warp / 2.0f;
    int any = 0;
    float gmin = INFINITY, gmax = -INFINITY;
    for (int r = 0; r < nrays; ++r) {
        ray_box(origins + 3 * (size_t)r, dirs + 3 * (size_t)r, half, &ray_start[r], &ray_end[r]);
        const int v = ray_end[r] > ray_start[r];
        valid_out[r] = (uint8_t)v;
        if (v) {
            any = 1;
            if (ray_start[r] < gmin) gmin = ray_start[r];
            if (ray_start[r] > gmax) gmax = ray_start[r];
        }
    }
    if (any)
        for (int r = 0; r < nrays; ++r)
            if (!valid_out[r]) { ray_start[r] = gmin; ray_end[r] = gmax; }
}

/* ------------------------------------------------------------------------------------
 * A4  tri-plane bilinear sampling (grid_sample: bilinear, zeros padding, align_corners=False)
 * modules/eg3ds/volumetric_rendering/renderer.py:65-75 (scale by 2/box_warp :71, projection
 *   :49-63 with the plane axes of :30-47 -> plane0 (x,y), plane1 (x,z), plane2 (z,x); the first
 *   projected coordinate indexes the LAST (width) axis).
 * planes: [3, C, H, W] (one batch item, NCHW as the reference holds them).
 * feat_out[3][C]
 * ---------------------------------------------------------------------------------- */
/* tri-grid depth (triplane_feature_type 'trigrid' / 'trigrid_v2', hparams triplane_depth): 1 = plain tri-planes.
 * The reference reads it from a module-global hparams dict (renderer.py:181); so does this restatement. */
static int g_triplane_depth = 1;
R3D_API void r3d_oracle_set_triplane_depth(int d) { g_triplane_depth = d < 1 ? 1 : d; }

/* sample_from_trigrids (renderer.py:78-89): planes [3][C*D][H][W] viewed as [3][C][D][H][W] (channel c*D + d), 5-D
 * F.grid_sample (tri-linear, zeros padding, align_corners=False) at the projected coordinates: (u, v) as for tri-planes,
 * third coordinate = z, y, y (coordinates . inv(plane_axes), renderer.py:29-45,60-62); grid x -> W, y -> H, z -> D. */
static void sample_trigrids(const float* planes, int C, int D, int H, int W, const float* us, const float* vs, const float* ws,
                            float* feat_out)
{
    for (int p = 0; p < 3; ++p) {
        const float ix = ((us[p] + 1.0f) * (float)W - 1.0f) / 2.0f;
        const float iy = ((vs[p] + 1.0f) * (float)H - 1.0f) / 2.0f;
        const float iz = ((ws[p] + 1.0f) * (float)D - 1.0f) / 2.0f;
        const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
        const int x0 = (x0f >= -2.0f && x0f <= (float)W + 1.0f) ? (int)x0f : -5;
        const int y0 = (y0f >= -2.0f && y0f <= (float)H + 1.0f) ? (int)y0f : -5;
        const int z0 = (z0f >= -2.0f && z0f <= (float)D + 1.0f) ? (int)z0f : -5;
        const float* P = planes + (size_t)p * C * D * H * W;
        for (int c = 0; c < C; ++c) {
            float acc = 0.0f;
            for (int dz = 0; dz < 2; ++dz)
                for (int dy = 0; dy < 2; ++dy)
                    for (int dx = 0; dx < 2; ++dx) {
                        const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                        if (x < 0 || x >= W || y < 0 || y >= H || z < 0 || z >= D) continue;
                        const float wx = dx ? ix - x0f : (x0f + 1.0f) - ix;
                        const float wy = dy ? iy - y0f : (y0f + 1.0f) - iy;
                        const float wz = dz ? iz - z0f : (z0f + 1.0f) - iz;
                        acc += P[(((size_t)c * D + z) * H + y) * W + x] * (wx * wy * wz);
                    }
            feat_out[p * C + c] = acc;
        }
    }
}

static void sample_planes(const float* planes, int C, int H, int W, float box_warp,
                          const float* xyz, float* feat_out /* [3*C] */)
{
    const float s = (float)(2.0 / (double)box_warp);
    const float q[3] = { xyz[0] * s, xyz[1] * s, xyz[2] * s };
    const float us[3] = { q[0], q[0], q[2] };
    const float vs[3] = { q[1], q[2], q[0] };
    if (g_triplane_depth > 1) {
        const float ws[3] = { q[2], q[1], q[1] };
        sample_trigrids(planes, C, g_triplane_depth, H, W, us, vs, ws, feat_out);
        return;
    }
    for (int p = 0; p < 3; ++p) {
        const float ix = ((us[p] + 1.0f) * (float)W - 1.0f) / 2.0f;
        const float iy = ((vs[p] + 1.0f) * (float)H - 1.0f) / 2.0f;
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
        const float wnw = (x1f - ix) * (y1f - iy);
        const float wne = (ix - x0f) * (y1f - iy);
        const float wsw = (x1f - ix) * (iy - y0f);
        const float wse = (ix - x0f) * (iy - y0f);
        /* out-of-range coordinates (incl. NaN/inf) contribute nothing */
        const int x0 = (x0f >= -2.0f && x0f <= (float)W + 1.0f) ? (int)x0f : -5;
        const int y0 = (y0f >= -2.0f && y0f <= (float)H + 1.0f) ? (int)y0f : -5;
        const int x1 = x0 + 1, y1 = y0 + 1;
        const int vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
        const int vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
        const float* P = planes + (size_t)p * C * H * W;
        for (int c = 0; c < C; ++c) {
            const float* Pc = P + (size_t)c * H * W;
            float acc = 0.0f;
            if (vx0 && vy0) acc += Pc[(size_t)y0 * W + x0] * wnw;
            if (vx1 && vy0) acc += Pc[(size_t)y0 * W + x1] * wne;
            if (vx0 && vy1) acc += Pc[(size_t)y1 * W + x0] * wsw;
            if (vx1 && vy1) acc += Pc[(size_t)y1 * W + x1] * wse;
            feat_out[p * C + c] = acc;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * A5  OSGDecoder
 * modules/eg3ds/models/triplane.py:177-189 (== modules/img2plane/triplane.py:133-146):
 *   mean over the 3 planes (:179); FullyConnectedLayer 32->64 (networks_stylegan2.py:99-131,
 *   weight_gain = 1/sqrt(in), addmm :127); torch.nn.Softplus (beta 1, threshold 20);
 *   FullyConnectedLayer 64->33; rgb = sigmoid(y[1:])*(1+2*0.001)-0.001 (:187); sigma = y[0] (:188).
 * w1 [HID, C], b1 [HID], w2 [OUT, HID], b2 [OUT] are the RAW module parameters
 * (decoder.net.0.weight etc.); gains are applied here like the reference does at call time.
 * ---------------------------------------------------------------------------------- */
static inline float softplus20(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

static void decode(const float* feat3 /*[3*C]*/, int C, int HID, int OUT,
                   const float* w1, const float* b1, const float* w2, const float* b2,
                   float* sigma, float* rgb /* [OUT-1] */)
{
    float x[64], h[128];
    const float g1 = (float)(1.0 / sqrt((double)C));
    const float g2 = (float)(1.0 / sqrt((double)HID));
    for (int c = 0; c < C; ++c) x[c] = (feat3[c] + feat3[C + c] + feat3[2 * C + c]) / 3.0f;
    for (int j = 0; j < HID; ++j) {
        float acc = b1[j];
        for (int c = 0; c < C; ++c) acc += x[c] * (w1[j * C + c] * g1);
        h[j] = softplus20(acc);
    }
    for (int o = 0; o < OUT; ++o) {
        float acc = b2[o];
        for (int j = 0; j < HID; ++j) acc += h[j] * (w2[o * HID + j] * g2);
        if (o == 0) *sigma = acc;
        else rgb[o - 1] = (1.0f / (1.0f + expf(-acc))) * 1.002f - 0.001f;
    }
}

/* run_model: modules/eg3ds/volumetric_rendering/renderer.py:169-188 (inference branches only:
 * sample_from_planes :179 + decoder :185; view directions are ignored by the decoder).
 * planes [N,3,C,H,W]; coords [N, npts, 3]; rgb_out [N,npts,OUT-1]; sigma_out [N,npts]. */
R3D_API void r3d_oracle_run_model(const float* planes, int N, int C, int H, int W,
                                  const float* w1, const float* b1, const float* w2, const float* b2,
                                  int HID, int OUT, float box_warp,
                                  const float* coords, int npts, float* rgb_out, float* sigma_out)
{
    for (int n = 0; n < N; ++n) {
        const float* Pn = planes + (size_t)n * 3 * C * g_triplane_depth * H * W;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < npts; ++i) {
            float feat[3 * 64];
            const size_t idx = (size_t)n * npts + i;
            sample_planes(Pn, C, H, W, box_warp, coords + 3 * idx, feat);
            decode(feat, C, HID, OUT, w1, b1, w2, b2, &sigma_out[idx], rgb_out + idx * (OUT - 1));
        }
    }
}

/* ------------------------------------------------------------------------------------
 * A6  MipRayMarcher2.run_forward  modules/eg3ds/volumetric_rendering/ray_marcher.py:25-57
 * One ray, S samples in the order given (the caller sorts for the merged pass).
 * Returns unclamped composite depth (NaN -> +inf applied); the GLOBAL clamp to
 * [min(depths), max(depths)] (:50) is applied by the caller over the whole call.
 * weights_out: S-1 values.  rgb_out: CO channels, already scaled to (-1,1) (:55).
 * ---------------------------------------------------------------------------------- */
static void march_ray(const float* colors /*[S][CO]*/, const float* dens /*[S]*/, const float* depths /*[S]*/,
                      int S, int CO, int white_back,
                      float* rgb_out, float* depth_out, float* wsum_out, float* weights_out)
{
    float T = 1.0f;           /* cumprod of [1, 1-alpha+1e-10][: -1] */
    float wsum = 0.0f, dsum = 0.0f;
    for (int c = 0; c < CO; ++c) rgb_out[c] = 0.0f;
    for (int k = 0; k < S - 1; ++k) {
        const float delta = depths[k + 1] - depths[k];
        const float dmid = (dens[k] + dens[k + 1]) / 2.0f;
        const float tmid = (depths[k] + depths[k + 1]) / 2.0f;
        const float sp = softplus20(dmid - 1.0f);
        const float alpha = 1.0f - expf(-(sp * delta));
        const float w = alpha * T;
        T = T * (1.0f - alpha + 1e-10f);
        weights_out[k] = w;
        wsum += w;
        dsum += w * tmid;
        for (int c = 0; c < CO; ++c)
            rgb_out[c] += w * ((colors[(size_t)k * CO + c] + colors[(size_t)(k + 1) * CO + c]) / 2.0f);
    }
    float dep = dsum / wsum;
    if (isnan(dep)) dep = INFINITY;       /* torch.nan_to_num(x, float('inf')) */
    if (isinf(dep)) dep = dep > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f; /* posinf/neginf default */
    *depth_out = dep;
    *wsum_out = wsum;
    for (int c = 0; c < CO; ++c) {
        float v = rgb_out[c];
        if (white_back) v = v + 1.0f - wsum;
        rgb_out[c] = v * 2.0f - 1.0f;
    }
}

/* ------------------------------------------------------------------------------------
 * A7  importance sampling  modules/eg3ds/volumetric_rendering/renderer.py:234-297
 *   max_pool1d(k2,s1,pad1) -> avg_pool1d(k2,s1) -> +0.01 (:245-247); bins = depth midpoints (:249);
 *   sample_pdf over weights[1:-1] (+1e-5), cdf with leading 0, searchsorted(right=True),
 *   below/above clamps, denom<eps -> 1, linear interpolation (:271-296).
 * ---------------------------------------------------------------------------------- */
static void importance_ray(const float* z /*[Nc]*/, const float* w /*[Nc-1]*/, int Nc,
                           const float* u /*[Nf]*/, int Nf, float* out /*[Nf]*/)
{
    float a[512], s[512], bins[512], cdf[512];
    const int nw = Nc - 1;
    for (int i = 0; i < Nc; ++i) {            /* max_pool1d, -inf padding */
        const float l = (i - 1 >= 0) ? w[i - 1] : -INFINITY;
        const float r = (i < nw) ? w[i] : -INFINITY;
        a[i] = l > r ? l : r;
    }
    for (int i = 0; i < nw; ++i) s[i] = (a[i] + a[i + 1]) / 2.0f + 0.01f;   /* avg_pool1d + 0.01 */
    for (int i = 0; i < nw; ++i) bins[i] = 0.5f * (z[i] + z[i + 1]);
    const int ns = Nc - 3;                   /* weights[:, 1:-1] */
    float tot = 0.0f;
    for (int i = 0; i < ns; ++i) tot += (s[i + 1] + 1e-5f);
    cdf[0] = 0.0f;
    float run = 0.0f;
    for (int i = 0; i < ns; ++i) { run += (s[i + 1] + 1e-5f) / tot; cdf[i + 1] = run; }
    const int ncdf = ns + 1;
    for (int j = 0; j < Nf; ++j) {
        int ind = 0;                          /* searchsorted right: #cdf <= u */
        while (ind < ncdf && cdf[ind] <= u[j]) ++ind;
        const int below = ind - 1 > 0 ? ind - 1 : 0;
        const int above = ind < ns ? ind : ns;
        float denom = cdf[above] - cdf[below];
        if (denom < 1e-5f) denom = 1.0f;
        out[j] = bins[below] + (u[j] - cdf[below]) / denom * (bins[above] - bins[below]);
    }
}

/* stable insertion ordering of indices by depth (torch.sort ascending, renderer.py:202) */
static void argsort_depth(const float* t, int n, int* idx)
{
    for (int i = 0; i < n; ++i) idx[i] = i;
    for (int i = 1; i < n; ++i) {
        const int v = idx[i];
        int j = i - 1;
        while (j >= 0 && t[idx[j]] > t[v]) { idx[j + 1] = idx[j]; --j; }
        idx[j + 1] = v;
    }
}

/* ------------------------------------------------------------------------------------
 * A10  ImportanceRenderer.forward ('auto' limits)  renderer.py:118-167
 *   limits + fix-up (:121-126) -> sample_stratified tensor branch (:223-226, math_utils.linspace
 *   :101-118) -> coarse run_model (:135-142) -> ray_marcher for weights (:147) -> sample_importance
 *   (:149) -> fine run_model (:150-157) -> unify_samples sort/gather (:197-207) -> ray_marcher (:163)
 *   -> (rgb, depth, weights.sum(2), is_ray_valid) (:167).
 * The two torch RNG draws (rand_like :226, rand :281) are INJECTED: noise_c [N*M*Nc], u_f [N*M*Nf].
 * Outputs: rgb [N*M*CO], depth [N*M], wsum [N*M], valid [N*M] (bytes).
 * Optional debug outputs (may be NULL): depths_c [N*M*Nc], depths_f [N*M*Nf], sigma_c [N*M*Nc].
 * ---------------------------------------------------------------------------------- */
R3D_API int r3d_oracle_render(const float* planes, int N, int C, int H, int W,
                              const float* w1, const float* b1, const float* w2, const float* b2,
                              int HID, int OUT,
                              const float* origins, const float* dirs, int M,
                              int Nc, int Nf, float box_warp, int white_back,
                              const float* noise_c, const float* u_f,
                              float* rgb, float* depth, float* wsum, uint8_t* valid,
                              float* dbg_depths_c, float* dbg_depths_f, float* dbg_sigma_c)
{
    const int CO = OUT - 1;
    const int nrays = N * M;
    if (Nc < 4 || Nc > 256 || Nf < 0 || Nf > 256 || C > 64 || HID > 128) return -1;
    float* rs = (float*)malloc(sizeof(float) * nrays);
    float* re = (float*)malloc(sizeof(float) * nrays);
    r3d_oracle_ray_limits(origins, dirs, nrays, box_warp, rs, re, valid);

    const int S = Nc + Nf;
    float gmin = INFINITY, gmax = -INFINITY;   /* global depth range of the FINAL marcher call */

#pragma omp parallel
    {
        float* col = (float*)malloc(sizeof(float) * (size_t)S * CO);
        float* den = (float*)malloc(sizeof(float) * S);
        float* t = (float*)malloc(sizeof(float) * S);
        float* col2 = (float*)malloc(sizeof(float) * (size_t)S * CO);
        float* den2 = (float*)malloc(sizeof(float) * S);
        float* t2 = (float*)malloc(sizeof(float) * S);
        float* wts = (float*)malloc(sizeof(float) * S);
        int* order = (int*)malloc(sizeof(int) * S);
        float feat[3 * 64];
        float lmin = INFINITY, lmax = -INFINITY;
#pragma omp for schedule(dynamic, 64)
        for (int r = 0; r < nrays; ++r) {
            const int n = r / M;
            const float* Pn = planes + (size_t)n * 3 * C * g_triplane_depth * H * W;
            const float* o = origins + 3 * (size_t)r;
            const float* d = dirs + 3 * (size_t)r;
            const float start = rs[r], end = re[r];
            /* stratified depths: linspace + jitter */
            const float delta = (end - start) / (float)(Nc - 1);
            for (int k = 0; k < Nc; ++k) {
                const float step = (float)k / (float)(Nc - 1);
                float tk = start + step * (end - start);
                tk += noise_c[(size_t)r * Nc + k] * delta;
                t[k] = tk;
                float xyz[3] = { o[0] + tk * d[0], o[1] + tk * d[1], o[2] + tk * d[2] };
                sample_planes(Pn, C, H, W, box_warp, xyz, feat);
                decode(feat, C, HID, OUT, w1, b1, w2, b2, &den[k], col + (size_t)k * CO);
            }
            if (dbg_depths_c) memcpy(dbg_depths_c + (size_t)r * Nc, t, sizeof(float) * Nc);
            if (dbg_sigma_c) memcpy(dbg_sigma_c + (size_t)r * Nc, den, sizeof(float) * Nc);
            float rgb_r[64], dep_r, ws_r;
            if (Nf > 0) {
                march_ray(col, den, t, Nc, CO, white_back, rgb_r, &dep_r, &ws_r, wts);
                importance_ray(t, wts, Nc, u_f + (size_t)r * Nf, Nf, t + Nc);
                if (dbg_depths_f) memcpy(dbg_depths_f + (size_t)r * Nf, t + Nc, sizeof(float) * Nf);
                for (int k = Nc; k < S; ++k) {
                    float xyz[3] = { o[0] + t[k] * d[0], o[1] + t[k] * d[1], o[2] + t[k] * d[2] };
                    sample_planes(Pn, C, H, W, box_warp, xyz, feat);
                    decode(feat, C, HID, OUT, w1, b1, w2, b2, &den[k], col + (size_t)k * CO);
                }
                argsort_depth(t, S, order);
                for (int k = 0; k < S; ++k) {
                    t2[k] = t[order[k]];
                    den2[k] = den[order[k]];
                    memcpy(col2 + (size_t)k * CO, col + (size_t)order[k] * CO, sizeof(float) * CO);
                }
                march_ray(col2, den2, t2, S, CO, white_back, rgb_r, &dep_r, &ws_r, wts);
                for (int k = 0; k < S; ++k) { if (t2[k] < lmin) lmin = t2[k]; if (t2[k] > lmax) lmax = t2[k]; }
            } else {
                march_ray(col, den, t, Nc, CO, white_back, rgb_r, &dep_r, &ws_r, wts);
                for (int k = 0; k < Nc; ++k) { if (t[k] < lmin) lmin = t[k]; if (t[k] > lmax) lmax = t[k]; }
            }
            memcpy(rgb + (size_t)r * CO, rgb_r, sizeof(float) * CO);
            depth[r] = dep_r;
            wsum[r] = ws_r;
        }
#pragma omp critical
        { if (lmin < gmin) gmin = lmin; if (lmax > gmax) gmax = lmax; }
        free(col); free(den); free(t); free(col2); free(den2); free(t2); free(wts); free(order);
    }
    /* ray_marcher.py:50  clamp to the global [min, max] of all depths in the call */
    for (int r = 0; r < nrays; ++r) {
        float v = depth[r];
        if (v < gmin) v = gmin;
        if (v > gmax) v = gmax;
        depth[r] = v;
    }
    free(rs); free(re);
    return 0;
}

/* ====================================================================================
 * A9  Super-resolution (StyleGAN2 synthesis blocks), fp32, eval, noise_mode='none'
 * ==================================================================================== */

/* styles = affine(w): FullyConnectedLayer(w_dim -> Cin, bias_init=1), weight_gain 1/sqrt(w_dim)
 * modules/eg3ds/models/networks_stylegan2.py:99-131, :312, :326 */
static void affine_styles(const float* aw /*[Cin][WD]*/, const float* ab /*[Cin]*/, const float* wvec /*[WD]*/,
                          int Cin, int WD, float* styles)
{
    const float g = (float)(1.0 / sqrt((double)WD));
    for (int c = 0; c < Cin; ++c) {
        float acc = ab[c];
        for (int j = 0; j < WD; ++j) acc += wvec[j] * (aw[(size_t)c * WD + j] * g);
        styles[c] = acc;
    }
}

/* modulated weights w'' = W*s*rsqrt(sum (W*s)^2 + 1e-8)  networks_stylegan2.py:62-70 */
static void modulate(const float* Wt /*[Co][Ci][k][k]*/, const float* styles, int Co, int Ci, int kk,
                     int demod, float* out)
{
    for (int o = 0; o < Co; ++o) {
        float ss = 0.0f;
        for (int c = 0; c < Ci; ++c)
            for (int t = 0; t < kk; ++t) {
                const float v = Wt[((size_t)o * Ci + c) * kk + t] * styles[c];
                out[((size_t)o * Ci + c) * kk + t] = v;
                ss += v * v;
            }
        if (demod) {
            const float d = 1.0f / sqrtf(ss + 1e-8f);
            for (int c = 0; c < Ci; ++c)
                for (int t = 0; t < kk; ++t) out[((size_t)o * Ci + c) * kk + t] *= d;
        }
    }
}

/* plain 3x3 correlation, pad 1 (F.conv2d; conv2d_resample.py:139-141 fast path) NCHW one image.
 * Output-stationary bands of BAND rows x one output channel (the band of y stays in L1 while all input channels and taps stream
 * past); every output element is still accumulated in the order (c, ky, kx), so the result does not depend on BAND or threads. */
/* Grow-only scratch buffers that survive between calls (the oracle is single-caller test infrastructure): a fresh 270 MB calloc per
 * convolution costs more page faults than the convolution costs arithmetic on a many-core host. */
static float* scratch_buf(int slot, size_t floats)
{
    static float* buf[4];
    static size_t cap[4];
    if (cap[slot] < floats) {
        free(buf[slot]);
        buf[slot] = (float*)malloc(sizeof(float) * floats);
        cap[slot] = buf[slot] ? floats : 0;
    }
    return buf[slot];
}
/* zero-padded copy [C][H + 2][Wp] of x [C][H][W] (interior at row + 1, column + 1) */
static float* padded_copy(int slot, const float* x, int Ci, int Hh, int Ww, int Wp)
{
    const int Hp = Hh + 2;
    float* xp = scratch_buf(slot, (size_t)Ci * Hp * Wp);
    if (!xp) return NULL;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < Ci; ++c)
        for (int yy = 0; yy < Hp; ++yy) {
            float* row = xp + ((size_t)c * Hp + yy) * Wp;
            if (yy == 0 || yy == Hp - 1) { memset(row, 0, sizeof(float) * (size_t)Wp); continue; }
            row[0] = 0.0f;
            memcpy(row + 1, x + ((size_t)c * Hh + yy - 1) * Ww, sizeof(float) * (size_t)Ww);
            memset(row + 1 + Ww, 0, sizeof(float) * (size_t)(Wp - 1 - Ww));
        }
    return xp;
}

#define R3D_BAND 8
#define R3D_OB 4          /* output channels per register tile */
#define R3D_XB 16         /* pixels per register tile */
/* Register-tiled: a tile of R3D_OB output channels x R3D_XB pixels of one row stays in registers while all input channels and taps stream
 * past a zero-padded copy of the input (so the taps need no bounds tests; an out-of-image tap adds w * 0).  Every output element is still
 * accumulated from zero in the order (c, ky, kx): the result is bit-identical to the plain loop nest and independent of the tiling. */
static void conv3x3(const float* x, int Ci, int Hh, int Ww, const float* w /*[Co][Ci][3][3]*/, int Co, float* y)
{
    const int Hp = Hh + 2, Wp = Ww + 2 + R3D_XB;        /* + R3D_XB: the last (partial) pixel tile reads past the row end inside the copy */
    float* xp = padded_copy(0, x, Ci, Hh, Ww, Wp);
    if (!xp) return;
    const int nob = (Co + R3D_OB - 1) / R3D_OB;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ob = 0; ob < nob; ++ob)
        for (int yy = 0; yy < Hh; ++yy) {
            const int o0 = ob * R3D_OB, on = o0 + R3D_OB <= Co ? R3D_OB : Co - o0;
            for (int xb = 0; xb < Ww; xb += R3D_XB) {
                float acc[R3D_OB][R3D_XB];
                for (int o = 0; o < R3D_OB; ++o)
                    for (int i = 0; i < R3D_XB; ++i) acc[o][i] = 0.0f;
                for (int c = 0; c < Ci; ++c) {
                    const float* xc = xp + ((size_t)c * Hp + yy) * Wp + xb;
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const float* xr = xc + (size_t)ky * Wp + kx;
                            for (int o = 0; o < on; ++o) {
                                const float wv = w[((size_t)(o0 + o) * Ci + c) * 9 + ky * 3 + kx];
#pragma omp simd
                                for (int i = 0; i < R3D_XB; ++i) acc[o][i] += wv * xr[i];
                            }
                        }
                }
                const int nx = xb + R3D_XB <= Ww ? R3D_XB : Ww - xb;
                for (int o = 0; o < on; ++o) memcpy(y + ((size_t)(o0 + o) * Hh + yy) * Ww + xb, acc[o], sizeof(float) * (size_t)nx);
            }
        }
}

/* up=2 path of conv2d_resample (conv2d_resample.py:116-133 with padding=1, f 4x4):
 * conv_transpose2d(stride 2, no padding, weight used un-flipped) -> [Co][2H+1][2W+1],
 * then upfirdn2d(pad 1,1,1,1, gain 4) with f = outer([1,3,3,1])/64 (upfirdn2d.py:72-116,171-215).
 * T is built in bands of T rows (output-stationary, accumulation order (c, ky, kx) per element as before), then filtered. */
static void upconv3x3(const float* x, int Ci, int Hh, int Ww, const float* w /*[Co][Ci][3][3]*/, int Co, float* y /*[Co][2H][2W]*/)
{
    const int Th = 2 * Hh + 1, Tw = 2 * Ww + 1;
    const int Oh = 2 * Hh, Ow = 2 * Ww;
    static const float f1[4] = { 1.0f, 3.0f, 3.0f, 1.0f };
    float f2[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) f2[a][b] = (f1[a] * f1[b] / 64.0f) * 4.0f;
    float* Tall = scratch_buf(1, (size_t)Co * Th * Tw);
    if (!Tall) return;
    /* T[o][2 iy + ky][2 ix + kx] += w[o][c][ky][kx] * x[c][iy][ix].  Register-tiled like conv3x3: one T row of R3D_OB output channels is
     * built in two column-parity halves (even columns take kx = 0 from ix = j and kx = 2 from ix = j - 1; odd columns kx = 1), from a
     * zero-padded copy of x; per T element the terms arrive in the order (c, ky, kx), as in the plain scatter loop. */
    const int Hp = Hh + 2, Wp = Ww + 2 + R3D_XB;
    float* xp = padded_copy(2, x, Ci, Hh, Ww, Wp);
    if (!xp) return;
    const int nob = (Co + R3D_OB - 1) / R3D_OB;
    const int ne = Ww + 1;                               /* even columns tx = 2 j, j = 0..Ww; odd columns tx = 2 j + 1, j = 0..Ww-1 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int ob = 0; ob < nob; ++ob)
        for (int ty = 0; ty < Th; ++ty) {
            const int o0 = ob * R3D_OB, on = o0 + R3D_OB <= Co ? R3D_OB : Co - o0;
            /* rows of x feeding this T row: ky with (ty - ky) even and 0 <= (ty - ky) / 2 < Hh; padded row index = iy + 1 */
            int kys[2], iys[2], nk = 0;
            for (int ky = 0; ky < 3; ++ky) { const int d = ty - ky; if (d >= 0 && !(d & 1) && d / 2 < Hh) { kys[nk] = ky; iys[nk] = d / 2; ++nk; } }
            for (int jb = 0; jb < ne; jb += R3D_XB) {
                float ae[R3D_OB][R3D_XB], ao[R3D_OB][R3D_XB];
                for (int o = 0; o < R3D_OB; ++o)
                    for (int i = 0; i < R3D_XB; ++i) { ae[o][i] = 0.0f; ao[o][i] = 0.0f; }
                for (int c = 0; c < Ci; ++c)
                    for (int k = 0; k < nk; ++k) {
                        const float* xr = xp + ((size_t)c * Hp + iys[k] + 1) * Wp + 1 + jb;       /* xr[i] = x[c][iy][jb + i], xr[-1] = left neighbour / 0 */
                        for (int o = 0; o < on; ++o) {
                            const float* wk = w + ((size_t)(o0 + o) * Ci + c) * 9 + kys[k] * 3;
                            const float w0 = wk[0], w1 = wk[1], w2 = wk[2];
#pragma omp simd
                            for (int i = 0; i < R3D_XB; ++i) { ae[o][i] += w0 * xr[i]; ae[o][i] += w2 * xr[i - 1]; ao[o][i] += w1 * xr[i]; }
                        }
                    }
                for (int o = 0; o < on; ++o) {
                    float* tr = Tall + ((size_t)(o0 + o) * Th + ty) * Tw;
                    for (int i = 0; i < R3D_XB && jb + i < ne; ++i) {
                        tr[2 * (jb + i)] = ae[o][i];
                        if (jb + i < Ww) tr[2 * (jb + i) + 1] = ao[o][i];
                    }
                }
            }
        }
#pragma omp parallel for collapse(2) schedule(static)
    for (int o = 0; o < Co; ++o)
        for (int yy = 0; yy < Oh; ++yy) {
            const float* T = Tall + (size_t)o * Th * Tw;
            float* yo = y + (size_t)o * Oh * Ow;
            for (int xx = 0; xx < Ow; ++xx) {
                float acc = 0.0f;
                for (int a = 0; a < 4; ++a) {
                    const int ty = yy + a - 1;
                    if (ty < 0 || ty >= Th) continue;
                    for (int b = 0; b < 4; ++b) {
                        const int tx = xx + b - 1;
                        if (tx < 0 || tx >= Tw) continue;
                        acc += T[(size_t)ty * Tw + tx] * f2[a][b];
                    }
                }
                yo[(size_t)yy * Ow + xx] = acc;
            }
        }
}

/* bias_act: lrelu(alpha .2)*sqrt(2) or linear  (bias_act.py:93-122; networks_stylegan2.py:339-341) */
static void bias_act_inplace(float* x, int Cc, size_t hw, const float* b, int lrelu, float clampv)
{
    const float gain = lrelu ? (float)sqrt(2.0) : 1.0f;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < Cc; ++c) {
        float* xc = x + (size_t)c * hw;
        const float bc = b[c];
        for (size_t i = 0; i < hw; ++i) {
            float v = xc[i] + bc;
            if (lrelu) { v = v < 0.0f ? v * 0.2f : v; v = v * gain; }
            if (clampv >= 0.0f) { if (v > clampv) v = clampv; if (v < -clampv) v = -clampv; }
            xc[i] = v;
        }
    }
}

/* upsample2d of the RGB skip (upfirdn2d.py:317-354): zero-insert x2, pad (2,1), FIR 4x4 * 4 */
static void upsample2d_rgb(const float* img, int Cc, int Hh, int Ww, float* out /*[C][2H][2W]*/)
{
    static const float f1[4] = { 1.0f, 3.0f, 3.0f, 1.0f };
    const int Oh = 2 * Hh, Ow = 2 * Ww;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < Cc; ++c)
        for (int yy = 0; yy < Oh; ++yy)
            for (int xx = 0; xx < Ow; ++xx) {
                float acc = 0.0f;
                for (int a = 0; a < 4; ++a) {
                    const int uy = yy + a - 2;          /* index into zero-inserted image */
                    if (uy < 0 || uy >= Oh || (uy & 1)) continue;
                    for (int b = 0; b < 4; ++b) {
                        const int ux = xx + b - 2;
                        if (ux < 0 || ux >= Ow || (ux & 1)) continue;
                        /* filter is flipped for true convolution; [1,3,3,1] is symmetric */
                        acc += img[((size_t)c * Hh + uy / 2) * Ww + ux / 2] * ((f1[3 - a] * f1[3 - b] / 64.0f) * 4.0f);
                    }
                }
                out[((size_t)c * Oh + yy) * Ow + xx] = acc;
            }
}

/* torch.nn.Conv2d(Ci, Co, k, 1, padding=k/2) [+ torch.nn.LeakyReLU(slope)] as the torso / background fusion stacks
 * of SuperresolutionHybrid8XDC_Warp build them (modules/real3d/super_resolution/sr_with_ref.py:24-63), one image NCHW.
 * k in {1, 3}; slope < 0 = no activation. */
R3D_API int r3d_oracle_conv2d(const float* x, int Ci, int Hh, int Ww, const float* w /*[Co][Ci][k][k]*/, const float* b /*[Co] or NULL*/,
                              int Co, int k, float slope, float* y)
{
    const size_t hw = (size_t)Hh * Ww;
    if (k == 3) conv3x3(x, Ci, Hh, Ww, w, Co, y);
    else if (k == 1) {
#pragma omp parallel for schedule(static)
        for (int o = 0; o < Co; ++o) {
            float* yo = y + (size_t)o * hw;
            memset(yo, 0, sizeof(float) * hw);
            for (int c = 0; c < Ci; ++c) {
                const float wv = w[(size_t)o * Ci + c];
                const float* xc = x + (size_t)c * hw;
                for (size_t i = 0; i < hw; ++i) yo[i] += wv * xc[i];
            }
        }
    } else return -1;
    for (int o = 0; o < Co; ++o) {
        float* yo = y + (size_t)o * hw;
        const float bo = b ? b[o] : 0.0f;
        for (size_t i = 0; i < hw; ++i) {
            float v = yo[i] + bo;
            if (slope >= 0.0f && v < 0.0f) v *= slope;
            yo[i] = v;
        }
    }
    return 0;
}

/* torch.nn.UpsamplingBilinear2d(scale_factor=2): F.interpolate(mode='bilinear', align_corners=True), as used inside
 * SegFormerSECC2PlaneBackbone.to_plane_cnn (modules/real3d/segformer.py:698).  ATen upsample_bilinear2d: source index =
 * dst * (in-1)/(out-1), neighbours clamped at the last row/col.  x [C][H][W] -> y [C][2H][2W]. */
R3D_API int r3d_oracle_upsample2x_bilinear(const float* x, int Cc, int Hh, int Ww, float* y)
{
    const int Oh = 2 * Hh, Ow = 2 * Ww;
    const float sh = Oh > 1 ? (float)(Hh - 1) / (float)(Oh - 1) : 0.0f, sw = Ow > 1 ? (float)(Ww - 1) / (float)(Ow - 1) : 0.0f;
    for (int c = 0; c < Cc; ++c)
        for (int oy = 0; oy < Oh; ++oy) {
            const float fy = sh * oy;
            const int y0 = (int)fy, y1 = y0 + (y0 < Hh - 1 ? 1 : 0);
            const float ly = fy - y0, hy = 1.0f - ly;
            for (int ox = 0; ox < Ow; ++ox) {
                const float fx = sw * ox;
                const int x0 = (int)fx, x1 = x0 + (x0 < Ww - 1 ? 1 : 0);
                const float lx = fx - x0, hx = 1.0f - lx;
                const float* xc = x + (size_t)c * Hh * Ww;
                y[((size_t)c * Oh + oy) * Ow + ox] = hy * (hx * xc[y0 * Ww + x0] + lx * xc[y0 * Ww + x1]) +
                                                     ly * (hx * xc[y1 * Ww + x0] + lx * xc[y1 * Ww + x1]);
            }
        }
    return 0;
}

/* toRGB 1x1 mod-conv for a chunk of pixels: acc[i] = sum_c x[c][i] * w[c], every pixel accumulated from zero in channel order (as the
 * per-pixel loop did), but walking each channel plane contiguously */
#define R3D_PXB 256
static void torgb_chunk(const float* x, size_t plane, const float* w, int Cc, size_t n, float* acc)
{
    for (size_t i = 0; i < n; ++i) acc[i] = 0.0f;
    for (int c = 0; c < Cc; ++c) {
        const float wv = w[c];
        const float* xc = x + (size_t)c * plane;
#pragma omp simd
        for (size_t i = 0; i < n; ++i) acc[i] += xc[i] * wv;
    }
}

/* One SynthesisBlockNoUp (architecture 'skip', in_channels != 0)  modules/eg3ds/models/superresolution.py:215-250:
 *   conv0 :233 and conv1 :234 are both up=1 SynthesisLayers, img.add_(torgb(x)) :247 with no upsample (:241-243).
 * x [Ci][H][W], img [3][H][W] -> x_out [Co][H][W], img_out [3][H][W]. */
R3D_API int r3d_oracle_sr_block_noup(const float* x, const float* img, int Ci, int Co, int Hh, int Ww, int WD,
                                     const float* ws3,
                                     const float* c0_w, const float* c0_b, const float* c0_aw, const float* c0_ab,
                                     const float* c1_w, const float* c1_b, const float* c1_aw, const float* c1_ab,
                                     const float* rgb_w, const float* rgb_b, const float* rgb_aw, const float* rgb_ab,
                                     float clampv, float* x_out, float* img_out)
{
    const size_t hw = (size_t)Hh * Ww;
    float* st = (float*)malloc(sizeof(float) * (Ci > Co ? Ci : Co));
    float* wm0 = (float*)malloc(sizeof(float) * (size_t)Co * Ci * 9);
    float* wm1 = (float*)malloc(sizeof(float) * (size_t)Co * Co * 9);
    float* wm2 = (float*)malloc(sizeof(float) * (size_t)3 * Co);
    float* t0 = (float*)malloc(sizeof(float) * (size_t)Co * hw);
    if (!st || !wm0 || !wm1 || !wm2 || !t0) return -1;

    affine_styles(c0_aw, c0_ab, ws3 + 0 * WD, Ci, WD, st);
    modulate(c0_w, st, Co, Ci, 9, 1, wm0);
    conv3x3(x, Ci, Hh, Ww, wm0, Co, t0);
    bias_act_inplace(t0, Co, hw, c0_b, 1, clampv);

    affine_styles(c1_aw, c1_ab, ws3 + 1 * WD, Co, WD, st);
    modulate(c1_w, st, Co, Co, 9, 1, wm1);
    conv3x3(t0, Co, Hh, Ww, wm1, Co, x_out);
    bias_act_inplace(x_out, Co, hw, c1_b, 1, clampv);

    affine_styles(rgb_aw, rgb_ab, ws3 + 2 * WD, Co, WD, st);
    const float g = (float)(1.0 / sqrt((double)Co));
    for (int c = 0; c < Co; ++c) st[c] *= g;
    modulate(rgb_w, st, 3, Co, 1, 0, wm2);
    for (int o = 0; o < 3; ++o) {
        float* yo = img_out + (size_t)o * hw;
        const float* io = img + (size_t)o * hw;
#pragma omp parallel for schedule(static)
        for (size_t i0 = 0; i0 < hw; i0 += R3D_PXB) {
            float accv[R3D_PXB];
            const size_t n = i0 + R3D_PXB <= hw ? R3D_PXB : hw - i0;
            torgb_chunk(x_out + i0, hw, wm2 + (size_t)o * Co, Co, n, accv);
            for (size_t i = 0; i < n; ++i) {
                float v = accv[i] + rgb_b[o];
                if (clampv >= 0.0f) { if (v > clampv) v = clampv; if (v < -clampv) v = -clampv; }
                yo[i0 + i] = io[i0 + i] + v;
            }
        }
    }
    free(st); free(wm0); free(wm1); free(wm2); free(t0);
    return 0;
}

/* One SynthesisBlock (architecture 'skip', in_channels != 0)  networks_stylegan2.py:429-473
 *   conv0 (up=2) :459, conv1 :460, upsample2d(img) :463-465, torgb add :466-469.
 * Parameter pointers are the raw module tensors; ws3 = [3][WD] (conv0, conv1, torgb).
 * x [Ci][H][W], img [3][H][W] -> x_out [Co][2H][2W], img_out [3][2H][2W]. */
R3D_API int r3d_oracle_sr_block(const float* x, const float* img, int Ci, int Co, int Hh, int Ww, int WD,
                                const float* ws3,
                                const float* c0_w, const float* c0_b, const float* c0_aw, const float* c0_ab,
                                const float* c1_w, const float* c1_b, const float* c1_aw, const float* c1_ab,
                                const float* rgb_w, const float* rgb_b, const float* rgb_aw, const float* rgb_ab,
                                float clampv, float* x_out, float* img_out)
{
    const int Oh = 2 * Hh, Ow = 2 * Ww;
    const size_t ohw = (size_t)Oh * Ow;
    float* st = (float*)malloc(sizeof(float) * (Ci > Co ? Ci : Co));
    float* wm0 = (float*)malloc(sizeof(float) * (size_t)Co * Ci * 9);
    float* wm1 = (float*)malloc(sizeof(float) * (size_t)Co * Co * 9);
    float* wm2 = (float*)malloc(sizeof(float) * (size_t)3 * Co);
    float* t0 = (float*)malloc(sizeof(float) * (size_t)Co * ohw);
    if (!st || !wm0 || !wm1 || !wm2 || !t0) return -1;

    affine_styles(c0_aw, c0_ab, ws3 + 0 * WD, Ci, WD, st);
    modulate(c0_w, st, Co, Ci, 9, 1, wm0);
    upconv3x3(x, Ci, Hh, Ww, wm0, Co, t0);
    bias_act_inplace(t0, Co, ohw, c0_b, 1, clampv);

    affine_styles(c1_aw, c1_ab, ws3 + 1 * WD, Co, WD, st);
    modulate(c1_w, st, Co, Co, 9, 1, wm1);
    conv3x3(t0, Co, Oh, Ow, wm1, Co, x_out);
    bias_act_inplace(x_out, Co, ohw, c1_b, 1, clampv);

    /* toRGB: styles * 1/sqrt(Cin), 1x1 mod-conv without demod, linear bias_act (:365-370) */
    affine_styles(rgb_aw, rgb_ab, ws3 + 2 * WD, Co, WD, st);
    const float g = (float)(1.0 / sqrt((double)Co));
    for (int c = 0; c < Co; ++c) st[c] *= g;
    modulate(rgb_w, st, 3, Co, 1, 0, wm2);
    upsample2d_rgb(img, 3, Hh, Ww, img_out);
    for (int o = 0; o < 3; ++o) {
        float* yo = img_out + (size_t)o * ohw;
#pragma omp parallel for schedule(static)
        for (size_t i0 = 0; i0 < ohw; i0 += R3D_PXB) {
            float accv[R3D_PXB];
            const size_t n = i0 + R3D_PXB <= ohw ? R3D_PXB : ohw - i0;
            torgb_chunk(x_out + i0, ohw, wm2 + (size_t)o * Co, Co, n, accv);
            for (size_t i = 0; i < n; ++i) {
                float v = accv[i] + rgb_b[o];
                if (clampv >= 0.0f) { if (v > clampv) v = clampv; if (v < -clampv) v = -clampv; }
                yo[i0 + i] += v;
            }
        }
    }
    free(st); free(wm0); free(wm1); free(wm2); free(t0);
    return 0;
}
