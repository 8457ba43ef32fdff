/*
 * r3d_hip.h -- C ABI of libr3d_hip.so: MI355X (gfx950) native tri-plane NeRF render + super-resolution.
 *
 * This is the drop-in boundary for the per-frame hot path of Real3D-Portrait.  The reference has no
 * native renderer (it is ~25 eager ATen ops per pass) and three pybind11/CUDA plugins for the SR
 * blocks; every entry point below names the reference interface it replaces (file:line relative to the
 * upstream repo).  All pointers are BORROWED raw device pointers (fp32 unless noted), all tensors are
 * contiguous, `stream` is a hipStream_t passed as void*.  No function synchronises, allocates or frees
 * device memory.  Every function returns 0 on success or a negative r3d_status; the message of the last
 * failure on the calling thread is available from r3d_last_error().
 *
 * Reference-side binding: see INTEGRATION.md (ctypes stub a maintainer would add).
 */
#ifndef R3D_HIP_H
#define R3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* r3d_stream_t; /* hipStream_t */

enum r3d_status {
    R3D_OK = 0,
    R3D_ERR_INVALID_ARG = -1,   /* shape / pointer / option outside what the kernels support */
    R3D_ERR_WORKSPACE = -2,     /* workspace pointer NULL or too small */
    R3D_ERR_LAUNCH = -3,        /* HIP launch / runtime error (message has hipGetErrorString) */
    R3D_ERR_UNSUPPORTED = -4    /* valid in the reference, not implemented here (e.g. trigrid) */
};

#define R3D_FEATURES 32         /* tri-plane channels C            (img2plane_model.py:72-82)        */
#define R3D_HIDDEN 64           /* OSGDecoder hidden width         (modules/eg3ds/models/triplane.py:169) */
#define R3D_DECODER_OUT 33      /* 1 density + 32 colour features  (triplane.py:175)                 */

int r3d_version(void);
const char* r3d_last_error(void);

/* --- plane layout -------------------------------------------------------------------------------
 * Re-lays tri-planes from the reference's NCHW [N,3,C,H,W] to channel-last [N,3,H,W,C] so that each
 * bilinear tap is one contiguous 128-byte read, optionally fusing the per-frame `cano + secc` add.
 * Replaces: the plane add in OSAvatarSECC_Img2plane.cal_plane_given_cano (modules/real3d/
 * secc_img2plane.py:73-81) + the reshape at modules/eg3ds/volumetric_rendering/renderer.py:68.
 * `add` may be NULL.  add_flip: bit 2k flips plane k of `add` along H, bit 2k+1 along W, while it is read -- the
 * torch.flip calls SegFormerSECC2PlaneBackbone.forward applies to its conv output (modules/real3d/segformer.py:722-728:
 * planes 0,1 along H, plane 2 along H and W = 0b110101 = 53) when `add` is the raw to_plane_cnn output; 0 = none.
 * depth > 1: tri-grids (triplane_feature_type 'trigrid' / 'trigrid_v2', renderer.py:78-89): planes_nchw is
 * [N,3,C*depth,H,W] with channel c*depth + d (the reference's .view(N*3, C, D, H, W)) -> [N,3,depth,H,W,C]. */
#define R3D_SECC_PLANE_FLIPS 53
int r3d_planes_to_nhwc(const float* planes_nchw, const float* add_nchw, float* planes_nhwc,
                       int N, int C, int H, int W, int depth, int add_flip, float* absmax_partials, int* n_partials,
                       r3d_stream_t stream);
/* absmax_partials (may be NULL): device float[r3d_planes_absmax_partials(...)] that receives one max|value| per thread block of the
 * layout pass (no extra pass over the 25 MB, no atomics); *n_partials (host int) = how many were written.  They feed the renderer's
 * fp16 range fold (r3d_render_forward / r3d_run_model `plane_absmax`). */
size_t r3d_planes_absmax_partials(int N, int C, int H, int W, int depth);

/* --- A1 ray generation --------------------------------------------------------------------------
 * Replaces RaySampler.forward(cam2world[N,4,4], intrinsics[N,3,3], resolution)
 * (modules/eg3ds/volumetric_rendering/ray_sampler.py:24-63).  origins/dirs: [N, R*R, 3]. */
int r3d_raygen(const float* c2w, const float* intrinsics, int N, int R,
               float* origins, float* dirs, r3d_stream_t stream);

/* --- A2..A10 volume render ----------------------------------------------------------------------
 * Replaces ImportanceRenderer.forward(planes, decoder, ray_origins, ray_directions, rendering_options)
 * (modules/eg3ds/volumetric_rendering/renderer.py:118-167) with ray_start == ray_end == 'auto',
 * disparity_space_sampling False, clamp_mode 'softplus', feature type 'triplane': box limits + fix-up
 * (math_utils.py:46-98, renderer.py:121-126), stratified depths (:209-232), tri-plane sampling (:65-75),
 * OSGDecoder (modules/eg3ds/models/triplane.py:177-189), MipRayMarcher2 (ray_marcher.py:25-57),
 * importance resampling (:234-297), sort-merge (:197-207) and the final composite.
 *
 *   planes_nhwc  [N,3,H,W,32]      from r3d_planes_to_nhwc
 *   triplane_depth 1: tri-planes, bilinear (sample_from_planes, renderer.py:65-75);  D > 1: tri-grids [N,3,D,H,W,32],
 *                tri-linear with zero padding (sample_from_trigrids, renderer.py:78-89; hparams triplane_depth = 3)
 *   w1,b1,w2,b2  RAW decoder parameters decoder.net.0.{weight[64,32],bias[64]}, decoder.net.2.{weight[33,64],
 *                bias[33]}; the FullyConnectedLayer gains 1/sqrt(fan_in) are applied inside
 *   origins,dirs [N,M,3]
 *   Nc, Nf       depth_resolution / depth_resolution_importance (4 <= Nc <= 96, 0 <= Nf <= 96)
 *   noise_c      [N,M,Nc] U[0,1) stratification jitter (the reference's torch.rand_like, renderer.py:226)
 *   u_f          [N*M,Nf] U[0,1) importance draws       (the reference's torch.rand,      renderer.py:281)
 *                either may be NULL: the kernel then draws from a counter-based hash of
 *                (seed, ray, sample), which makes a frame's noise independent of how frames are sharded
 *   rgb [N,M,32] (rgb_channel_major = 0: what the reference's renderer returns) or [N,32,M] (rgb_channel_major = 1: the NCHW
 *                feature image TriPlaneGenerator.synthesis builds from it with permute(0,2,1).reshape(N,32,R,R), triplane.py:121-122,
 *                written directly so that no transposition pass runs per frame)
 *   depth [N,M] or NULL (no depth image: the global-range clamp launch is skipped), wsum [N,M], valid [N,M] (1 byte, is_ray_valid)
 *   plane_absmax device float[n_plane_absmax] whose maximum bounds |planes_nhwc| (the partials of r3d_planes_to_nhwc), or NULL / 0: the
 *                bound is then measured here (one extra pass over the planes).  fp16 range management of the decoder: the MLP runs on
 *                the f16 matrix pipe with fp32 operands split into fp16 hi + lo; one small launch per call (decoder_fold_kernel) derives
 *                exact power-of-two factors from this bound and from the weights (features * 2^b against W1 * 2^-b, and -- only when a
 *                bound would reach 2^15 -- 2^c on the hidden values / 2^d on the colour rows), so that the result equals the reference's
 *                unbounded fp32 evaluation for any magnitude of planes and weights whose pre-activations fit fp32.
 *                The bound must be RIGOROUS (max of the array >= max |planes_nhwc| as the kernel will read them): the per-sample fp16
 *                split has no saturation guard, so a stale or too small bound yields inf / NaN, not a clamp.  Pass NULL whenever the
 *                planes were modified after the partials were written (the Python operator does: it ties them to the tensor version).
 *   workspace    r3d_render_workspace_bytes() bytes of device scratch, 64-byte aligned.  Per-ray limits and the decoder's fold record; when Nc or Nf
 *                exceeds 48 AND Nf > 0 (the kernel shapes with more than three 16-sample tiles per pass and a fine pass) also 24 KB per wave of the
 *                launch's grid (min(512, rays / 4 rounded up to 8) blocks x 4 waves: 48 MB for a full grid), in which a wave parks the colours of the
 *                ray it is rendering between the decode and the composite (ABI 0.5.0: they do not fit the register file; 0.6.0: sized from the grid).
 */
size_t r3d_render_workspace_bytes(int N, int M, int Nc, int Nf);
int r3d_render_forward(const float* planes_nhwc, int N, int H, int W, int triplane_depth,
                       const float* w1, const float* b1, const float* w2, const float* b2,
                       const float* origins, const float* dirs, int M,
                       int Nc, int Nf, float box_warp, int white_back,
                       const float* noise_c, const float* u_f, uint64_t seed,
                       float* rgb, int rgb_channel_major, float* depth, float* wsum, uint8_t* valid,
                       const float* plane_absmax, int n_plane_absmax,
                       const float* cam2world, const float* intrinsics,
                       void* split_out, const float* split_scale, size_t split_scale_stride,
                       void* workspace, size_t workspace_bytes, r3d_stream_t stream);
/* Camera mode (origins = dirs = NULL, cam2world [N,4,4] + intrinsics [N,3,3] given, M = R * R): RaySampler.forward is evaluated inside
 * the limits pass and the render kernel with r3d_raygen's instruction sequence -- the same pixels as r3d_raygen + the ray arrays, without
 * the launch and the two [N,M,3] round trips.  cam2world / intrinsics are ignored when origins / dirs are given (pass NULL).
 * split_out (may be NULL): a second copy of the colours in R3D_FMT_SPLIT ([N][hi|lo][4][M][8] fp16), multiplied by split_scale[n][c]
 * (stride split_scale_stride floats; = the start of the consuming SR block's styles buffer AFTER r3d_chain_fold): the input of
 * r3d_sr_block_forward(x_format = R3D_FMT_SPLIT) without a conversion launch. */

/* Replaces ImportanceRenderer.run_model(planes, decoder, sample_coordinates, sample_directions, options)
 * (renderer.py:169-188; inference branches) -- the point-query used by .sample() (triplane.py:140-148).
 * coords [N,npts,3] -> rgb [N,npts,32], sigma [N,npts]. */
int r3d_run_model(const float* planes_nhwc, int N, int H, int W, int triplane_depth,
                  const float* w1, const float* b1, const float* w2, const float* b2,
                  const float* coords, int npts, float box_warp,
                  float* rgb, float* sigma, const float* plane_absmax, int n_plane_absmax,
                  void* workspace, size_t workspace_bytes, r3d_stream_t stream);
/* plane_absmax / n_plane_absmax: as for r3d_render_forward; workspace: r3d_run_model_workspace_bytes() bytes, 16-byte aligned */
size_t r3d_run_model_workspace_bytes(void);

/* --- A12..A14 super-resolution ------------------------------------------------------------------
 * One StyleGAN2 SynthesisBlock (architecture 'skip', up=2, fp32, eval, noise_mode 'none'):
 * replaces SynthesisBlock.forward (modules/eg3ds/models/networks_stylegan2.py:429-473) including
 * modulated_conv2d (:37-94), conv2d_resample up=2 path (torch_utils/ops/conv2d_resample.py:116-133),
 * and the three plugin calls it makes: bias_act_plugin.bias_act (ops/bias_act.cpp:36),
 * upfirdn2d_plugin.upfirdn2d (ops/upfirdn2d.cpp:20) for the post-T-conv FIR and the RGB-skip upsample2d.
 *
 * Modulated convolution is evaluated as  y = d[cout] * conv(s[ci] * x, W)  (== conv(x, W*s*d), :62-70), so:
 * r3d_sr_block_prepack: STATIC re-layout of conv0/conv1 weights ([tap][ci/8][cout][ci%8]); call once per
 *   parameter update.  prepacked: r3d_sr_block_prepacked_bytes(Cin,Cout) bytes.
 * r3d_sr_block_styles: per-forward small vectors for one batch of style inputs: styles s = affine(w) (:326),
 *   demodulation d = rsqrt(sum (W*s)^2 + 1e-8) (:65-70), modulated toRGB weights (:366-368), biases.
 *   ws3 [N,3,WD] (conv0, conv1, torgb);  *_w/_b/_aw/_ab = layer.weight / .bias / .affine.weight / .affine.bias;
 *   styles: r3d_sr_block_styles_bytes(N,Cin,Cout) bytes.
 * r3d_sr_block_forward:
 *   x [N,Cin,Hin,Win] in x_format, img [N,3,Hin,Win] NCHW fp32
 *   -> x_out in x_out_format (may be NULL / R3D_FMT_NONE), img_out [N,3,2Hin,2Win] NCHW fp32.
 *   up = 1: SynthesisBlock (conv0 up=2, RGB skip through upsample2d).  up = 0: SynthesisBlockNoUp
 *   (modules/eg3ds/models/superresolution.py:159-258: conv0 is a plain modulated 3x3 conv and img_out = img + toRGB(x)
 *   at the input resolution; img and img_out are [N,3,Hin,Win]); R3D_SR_F16X3 only.  Cin % 16 == 0, Cout % 128 == 0.
 *   Activation formats: R3D_FMT_NCHW fp32 (the reference layout); R3D_FMT_CB8 fp32 channel-blocked [N,C/8,H,W,8];
 *   R3D_FMT_SPLIT (f16x3 only): two fp16 planes [N][hi|lo][C/8][H][W][8] holding the activation ALREADY MULTIPLIED
 *   by the consumer conv's style vector -- as input it must have been scaled with this block's conv0 styles; as
 *   output it is scaled with `next_scale` ([N][Cout] floats, stride next_scale_stride: the next block's conv0 styles,
 *   i.e. the start of that block's styles buffer).
 *   R3D_FMT_SPLIT_MX (R3D_SR_F16MX): R3D_FMT_SPLIT whose lo plane holds, byte for byte in its place, the 8-bit correction
 *   records of the f16mx precision: lo chunk 2G: xh8 = e5m2(hi) of channels 16G..16G+15, lo chunk 2G+1: xl8 = e5m2(lo * 2^11) (OCP e5m2 since ABI
 *   0.5.0 -- a per-element exponent with the fp16 hi plane's range; 0.4.0: e4m3 of hi * 2^-7 / lo * 2^4, one exponent per tensor).
 *   As x_out_format the block's conv1 epilogue writes them; as x_format the block's up-sampling conv runs its cross products on the
 *   block-scaled 8-bit MFMA (2 instead of 3 matrix passes per MAC).  A producer / consumer pair must agree (same next_scale as SPLIT).
 *   clamp < 0 disables conv_clamp (the fp32 configuration Real3D uses, img2plane_baseline.py:102-104).
 *   img_u8 (may be NULL; R3D_SR_F16X3 only): [N,OH,OW,3] uint8 -- the block is the last one of the network and the frame leaves
 *   as clamp(-1,1) -> ((x + 1) / 2 * 255).int() (triplane.py:136 + inference/real3d_infer.py:472,518-522), fused into the
 *   toRGB finalize kernel; img_out may then be NULL.
 *   x_absmax (may be NULL; R3D_SR_F16X3 only): device float[N], atomically maxed with |x_out| (see r3d_chain_fold).
 *   R3D_SR_F16X3 reads the FOLDED vectors of the styles buffer: call r3d_chain_fold (below) after r3d_sr_block_styles and
 *   before r3d_sr_block_forward, every forward.
 *
 * precision: R3D_SR_F32   exact fp32 on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain);
 *            R3D_SR_F16X3 fp32-accurate on the f16 matrix pipe: every operand is split x = hi + lo (two fp16
 *                         terms, 2^-24 relative) and hi*hi + hi*lo + lo*hi is accumulated in fp32 by
 *                         v_mfma_f32_32x32x16_f16 (3 MFMAs at 16x the f32 rate; ~1e-7 relative per dot product).
 *            R3D_SR_F16MX (what the Python operators pass by default): as F16X3, but in the 3x3 convs whose input arrives as R3D_FMT_SPLIT_MX the two
 *                         2^-11-sized correction products hi*lo + lo*hi of two taps x 16 channels are ONE block-scaled 8-bit MFMA
 *                         (v_mfma_scale_f32_32x32x64_f8f6f4: activation records OCP e5m2, weight records OCP e4m3, 2x the f16 rate) instead of
 *                         four f16 MFMAs: 1.5x fewer matrix cycles.  The correction is then accurate to e5m2 rounding, i.e. ~2^-15 of each
 *                         product (between fp32's 2^-24 and TF32's 2^-11); parity: <= 5e-5 * max|ref| per block on the reference goldens,
 *                         <= 6.3e-5 near and far field on the heavy-tail sweeps (spikes of 2^6 .. 2^14 sigma) against the 2e-4 tolerance of
 *                         the path, <= 1e-4 over the 2^-20..2^14 operand sweeps vs fp64 (tests/test_gpu_mx.py, tests/test_gpu_pinned_config.py).
 *                         The records have the hi plane's exponent range: no measured bound is needed beyond what R3D_SR_F16X3 needs
 *                         (R3D_CHAIN_SR_BLOCK_TAIL stays available for chains deeper than three layers).
 *            The prepacked buffer is precision-specific (same size).
 *            Algorithm (ABI 0.6.0): the block's plain 3x3 conv (conv1) has a second implementation, Winograd F(2,3) along x with the input transform in
 *            the kernel (csrc/r3d_sr_wino.h: 12 instead of 18 matrix products per pair of output columns; the prepacked buffer carries the 12
 *            transformed tap matrices).  It is taken for R3D_SR_F16X3 when the output is whole 16 x 16 tiles (-13 % per launch, fp32-class:
 *            error x 1.4 of the direct form's 7e-8); for R3D_SR_F16MX the direct kernel is faster and stays.  Process-wide switch:
 *            R3D_CONV_WINO = 0 (never) | 1 (both precisions) | 2 (f16mx only) | 3 (f16x3 only, the default). */
enum r3d_sr_precision { R3D_SR_F32 = 0, R3D_SR_F16X3 = 1, R3D_SR_F16MX = 2 };
enum r3d_act_format { R3D_FMT_NONE = -1, R3D_FMT_NCHW = 0, R3D_FMT_CB8 = 1, R3D_FMT_SPLIT = 2, R3D_FMT_SPLIT_MX = 3 };
size_t r3d_sr_block_prepacked_bytes(int Cin, int Cout);
size_t r3d_sr_block_styles_bytes(int N, int Cin, int Cout);
size_t r3d_sr_block_workspace_bytes(int N, int Cin, int Cout, int Hin, int Win);
int r3d_sr_block_prepack(int Cin, int Cout, const float* c0_w, const float* c1_w, void* prepacked,
                         int precision, r3d_stream_t stream);
int r3d_sr_block_styles(const float* ws3, int N, int WD, int Cin, int Cout,
                        const float* c0_w, const float* c0_b, const float* c0_aw, const float* c0_ab,
                        const float* c1_w, const float* c1_b, const float* c1_aw, const float* c1_ab,
                        const float* rgb_w, const float* rgb_b, const float* rgb_aw, const float* rgb_ab,
                        void* styles, r3d_stream_t stream);
int r3d_sr_block_forward(const void* prepacked, const void* styles, int N, int Cin, int Cout, int Hin, int Win, int up,
                         const void* x, int x_format, const float* img, float clamp,
                         void* x_out, int x_out_format, const float* next_scale, size_t next_scale_stride,
                         float* img_out, uint8_t* img_u8, float* x_absmax, int precision,
                         void* workspace, size_t workspace_bytes, r3d_stream_t stream);

/* --- fp16 range management of the R3D_SR_F16X3 path ----------------------------------------------------------------
 * The reference runs these layers in fp32 with conv_clamp=None (no range limit).  The f16x3 kernels keep fp32 accuracy while a
 * stored fp16 operand tensor has rms >= 2^-3 and max < 65504, so every operand is stored times an exact power of two:
 * weight rows at prepack time (max|w[co]| -> [2^10, 2^11)), activations per sample from a guaranteed bound B >= max|x| that is
 * propagated layer to layer (B_out = gain * max_co(|b[co]| + d[co] * sum|w[co] * s| * B_in)); the factors are taken out again in
 * the fp32 epilogue.  r3d_chain_fold walks a chain of layers ON THE DEVICE (one launch, no host sync) and writes each layer's
 * folded multipliers: SR blocks into their styles buffer, plain convs into their `scales` buffer (r3d_conv_scales_bytes).
 *   src_a / src_b: where the layer's input bound comes from: k >= 0 = output bound of op k (k < this op's index);
 *   -1 - j = ext_bounds[j] (device float[N]: a measured r3d_absmax result, a previous chain's bound, or a constant);
 *   R3D_CHAIN_SRC_NONE = unused.  Two sources = a concatenation / blend of two tensors (r3d_blend_cat_to_split): the max.
 * The bound loosens ~5 binades per layer (10 binades of slack remain after one layer): re-measure (r3d_absmax on an fp32 tensor)
 * after at most 3 chained layers.  r3d_*_bound_offset: float offset, within one sample's buffer, of the op's output bound.
 * Measuring without an extra pass: r3d_sr_block_forward / r3d_conv_forward take `y_absmax` (device float[N] or NULL): the conv
 * epilogue then atomically maxes |y| into it; the slot must be zero beforehand -- pass it in `zero_slots` of the r3d_chain_fold
 * launch that precedes the producer (up to R3D_CHAIN_MAX_ZERO slots of N floats each are cleared by the fold kernel).  The kernel
 * reads its external bounds BEFORE it clears: a slot may be an ext bound and a zero slot of the same fold (consume the measured
 * maximum of this frame, re-arm the slot for the next frame). */
#define R3D_CHAIN_MAX_ZERO 4
#define R3D_CHAIN_MAX_OPS 12
#define R3D_CHAIN_MAX_EXT 4
#define R3D_CHAIN_SRC_NONE (-1000)
enum r3d_chain_kind {
    R3D_CHAIN_SR_BLOCK = 0,       /* both layers of an SR block from the bound of the block input */
    R3D_CHAIN_CONV = 1,           /* a plain conv layer */
    R3D_CHAIN_SR_BLOCK_TAIL = 2   /* re-fold only the block's second layer (conv1 operand) from a MEASURED max|block input|: the
                                     block's input was already written with the multipliers of an earlier R3D_CHAIN_SR_BLOCK fold */
};
typedef struct r3d_chain_op {
    int kind;                 /* r3d_chain_kind */
    int Cin, Cout, ksize;     /* ksize: R3D_CHAIN_CONV only */
    int act;                  /* R3D_CHAIN_CONV: activation after the bias (|act(t)| <= gain * |t|); SR blocks always activate */
    float gain, clamp;        /* act gain (sqrt(2) for SR blocks); clamp < 0: off */
    int src_a, src_b;
    void* scales;             /* SR block: styles buffer (r3d_sr_block_styles).  conv: r3d_conv_scales_bytes() buffer */
    const void* prepacked;    /* conv: r3d_conv_prepack buffer (its tail holds 2^-kw[co] and sum|w[co]|) */
    const float* bias;        /* conv: [Cout] or NULL */
} r3d_chain_op;
int r3d_chain_fold(const r3d_chain_op* ops, int nops, int N, const float* const* ext_bounds, int n_ext,
                   float* const* zero_slots, int n_zero, r3d_stream_t stream);
size_t r3d_sr_block_bound_offset(int Cin, int Cout);
size_t r3d_conv_scales_bound_offset(int Cin, int Cout);
/* out[n] = max |x[n, :]| over count_per_sample floats.  `out` (float[N]) must be zero on entry; zero_next (float[N], may be NULL)
 * is cleared by the same launch so that two alternating slots need no memset. */
int r3d_absmax(const float* x, size_t count_per_sample, int N, float* out, float* zero_next, r3d_stream_t stream);

/* --- plain convolution layers around the SR blocks --------------------------------------
 * replaces torch.nn.Conv2d(Cin, Cout, k, 1, padding=k//2) [+ torch.nn.LeakyReLU] as used by the torso / background
 * fusion stacks of SuperresolutionHybrid8XDC_Warp (modules/real3d/super_resolution/sr_with_ref.py:24-63:
 * torso_encoder, bg_encoder, fuse_head_torso_convs, fuse_fg_bg_convs) and by to_plane_cnn (modules/real3d/segformer.py:691-700),
 * on the f16x3 convolution kernel of the SR blocks (fp32-accurate, see r3d_sr_precision).
 *   y = act(conv_k(x, W) + bias[co]),  act(t) = (t < 0 ? act_slope * t : t) * act_gain when act != 0, then clamp to +-clamp
 *   when clamp >= 0.  ksize 1 | 3.  weight [Cout,Cin,k,k] fp32 (torch layout).  Cout % 4 == 0.  bias [Cout] or NULL.
 *   scales: this layer's r3d_conv_scales_bytes() buffer, filled by r3d_chain_fold for this forward.
 *   x / y formats as for the SR blocks; blocked formats (CB8, SPLIT) need Cin % 16 == 0 / Cout % 8 == 0, NCHW takes
 *   any Cin (zero padded to 16 inside).  A SPLIT x must have been written with this layer's in-multiplier (the start of
 *   `scales`); a SPLIT y is multiplied by next_scale = the CONSUMER's in-multiplier vector (start of its scales / styles
 *   buffer, already folded), NULL = 1.  y_format = R3D_FMT_SPLIT_MX (Cout % 16 == 0): a SPLIT y whose lo plane holds the fp8 records
 *   the consumer's f16mx main loop reads (an R3D_SR_F16MX SynthesisBlock[NoUp], or the next r3d_conv_forward called with
 *   x_format = R3D_FMT_SPLIT_MX).  x_format = R3D_FMT_SPLIT_MX (ksize 3, Cin % 16 == 0): this conv's cross products run on the
 *   block-scaled fp8 MFMA (precision tier of R3D_SR_F16MX); r3d_conv_prepack writes both weight layouts, the input format selects.
 *   workspace (r3d_conv_workspace_bytes) is only used for non-SPLIT inputs. */
size_t r3d_conv_prepacked_bytes(int Cin, int Cout, int ksize);
size_t r3d_conv_workspace_bytes(int N, int Cin, int H, int W);
size_t r3d_conv_scales_bytes(int N, int Cin, int Cout);
int r3d_conv_prepack(const float* weight, int Cin, int Cout, int ksize, void* prepacked, r3d_stream_t stream);
int r3d_conv_forward(const void* prepacked, const void* scales, const float* bias,
                     int N, int Cin, int Cout, int H, int W, int ksize,
                     const void* x, int x_format, int act, float act_slope, float act_gain, float clamp,
                     void* y, int y_format, const float* next_scale, size_t next_scale_stride, float* y_absmax,
                     void* workspace, size_t workspace_bytes, r3d_stream_t stream);

/* Alpha / occlusion blend + channel concatenation, emitted as the SPLIT input of the next conv:
 *   y = cat([a * mask, b * (1 - mask)], dim=1),  a [N,Ca,H,W], b [N,Cb,H,W] (NCHW or CB8 fp32), mask [N,1,H,W]
 * replaces `torch.cat([x * head_torso_alpha, x_torso * (1 - head_torso_alpha)], dim=1)` and the person_occlusion / x_bg
 * twin in SuperresolutionHybrid8XDC_Warp.forward (modules/real3d/super_resolution/sr_with_ref.py:104,114,126,136).
 * y_split: [N][hi|lo][(Ca+Cb)/8][H][W][8] halfs, multiplied by next_scale ([N][Ca+Cb], the consumer's in-multiplier; NULL = 1);
 * y_format R3D_FMT_SPLIT, or R3D_FMT_SPLIT_MX (fp8 records in the lo plane) when the consumer runs the f16mx main loop (since 0.4.0).
 * Ca % 8 == Cb % 8 == 0, (Ca + Cb) % 16 == 0.  b = NULL (since 0.6.1, Ca % 16 == 0): only the `a` part is written, the planes still hold
 * (Ca + Cb) / 8 chunks -- the `b` part is the output of r3d_conv_forward_cat. */
int r3d_blend_cat_to_split(const float* a, int a_format, int Ca, const float* b, int b_format, int Cb, const float* mask,
                           int N, int H, int W, void* y_split, int y_format, const float* next_scale, size_t next_scale_stride,
                           r3d_stream_t stream);

/* A conv layer that writes its output as ONE PART of such a concatenation (since 0.6.1): channels [chan_off, chan_off + Cout) of the C_total-channel
 * SPLIT / SPLIT_MX tensor y_cat [N][hi|lo][C_total/8][H][W][8], every value times mask (mask_invert = 0) or 1 - mask (1) of its pixel and the
 * consumer's in-multiplier next_scale[chan_off + co] -- `x_torso = self.torso_encoder(hid)` followed by its half of
 * `torch.cat([x * alpha, x_torso * (1 - alpha)], dim=1)` (sr_with_ref.py:88,104) without the fp32 x_torso in between; the other part comes from
 * r3d_blend_cat_to_split called with b = NULL (which then writes only channels [0, Ca) of the Ca + Cb).  Bit-identical to the two-step form.
 * ksize = 1 (the 3x3 kernels sit at their register limit and do not carry this epilogue); Cout, chan_off, C_total multiples of 16; other
 * arguments as r3d_conv_forward (next_scale: the consumer's whole vector, indexed by the concatenated channel). */
int r3d_conv_forward_cat(const void* prepacked, const void* scales, const float* bias,
                         int N, int Cin, int Cout, int H, int W, int ksize,
                         const void* x, int x_format, int act, float act_slope, float act_gain, float clamp,
                         void* y_cat, int y_format, int C_total, int chan_off, const float* mask, int mask_invert,
                         const float* next_scale, size_t next_scale_stride,
                         void* workspace, size_t workspace_bytes, r3d_stream_t stream);

/* The same blend + concatenation FUSED into the 1x1 conv that consumes it (since 0.6.1):
 *   y = act(conv_1x1(cat([a * mask, b * (1 - mask)], dim=1), W) + bias)
 * = `x = torch.cat([x * person_occlusion, x_bg * (1 - person_occlusion)], dim=1); x = self.fuse_fg_bg_convs(x)` up to the first layer of
 * fuse_fg_bg_convs (modules/real3d/super_resolution/sr_with_ref.py:113-114 / :125-126, the Sequential of :56-62) -- and the head / torso twin
 * when that Sequential starts with a 1x1 conv.  The 512-channel operand is never written: the kernel blends, scales by the layer's in-multiplier
 * and splits while it stages (bit-identical to r3d_blend_cat_to_split followed by r3d_conv_forward; 134 MB read instead of 67 + 134 MB
 * written + 270 MB read at 256^2).  a [N,Ca/8,H,W,8], b [N,Cb/8,H,W,8] R3D_FMT_CB8 fp32, mask [N,1,H,W]; Ca % 8 == Cb % 8 == 0,
 * (Ca + Cb) % 64 == 0.  prepacked / scales / bias / act... / y... as r3d_conv_forward with ksize 1 and Cin = Ca + Cb; the in-multiplier a
 * conv layer's r3d_chain_fold writes is one power of two for all channels (element 0 of `scales` is read). */
int r3d_conv_forward_blend(const void* prepacked, const void* scales, const float* bias,
                           int N, int Ca, int Cb, int Cout, int H, int W,
                           const float* a, const float* b, const float* mask,
                           int act, float act_slope, float act_gain, float clamp,
                           void* y, int y_format, const float* next_scale, size_t next_scale_stride, float* y_absmax, r3d_stream_t stream);

/* torch.nn.UpsamplingBilinear2d(scale_factor=2) (align_corners=True), the resampling step inside to_plane_cnn
 * (modules/real3d/segformer.py:691-700), between two r3d_conv_forward layers: x fp32 channel-blocked [N,C/8,H,W,8] ->
 * y at 2H x 2W in R3D_FMT_CB8, R3D_FMT_SPLIT or R3D_FMT_SPLIT_MX (scaled by next_scale, NULL = 1; SPLIT_MX -- the consumer runs the f16mx
 * main loop -- needs C % 16 == 0).  C % 8 == 0. */
int r3d_upsample2x_bilinear(const float* x_cb8, int N, int C, int H, int W, void* y, int y_format,
                            const float* next_scale, size_t next_scale_stride, r3d_stream_t stream);

/* --- small image ops of the torso / background fusion forward (modules/real3d/super_resolution/sr_with_ref.py:67-137) --------------
 * r3d_resize_bilinear: torch.nn.functional.interpolate(x, size=(OH, OW), mode='bilinear', align_corners=False, antialias=A) on
 *   `planes` = N*C contiguous H x W planes (the 128 -> 256 / 512 -> 256 / occlusion resizes of :76-81,110,120).
 * r3d_blend: out = a * mask + b * (1 - mask), a, b [N,C,H,W], mask [N,1,H,W]  (`rgb * alpha + rgb_torso * (1 - alpha)` :103,113).
 * r3d_person_occlusion: clamp(torso_occlusion + where(alpha > head_threshold, 1, alpha), 0, 1)  (:107-112,117-122). */
int r3d_resize_bilinear(const float* x, int planes, int H, int W, float* y, int OH, int OW, int antialias, r3d_stream_t stream);
int r3d_blend(const float* a, const float* b, const float* mask, int N, int C, int H, int W, float* out, r3d_stream_t stream);
int r3d_person_occlusion(const float* alpha, const float* torso_occlusion, float head_threshold, size_t count, float* out,
                         r3d_stream_t stream);

/* --- output side --------------------------------------------------------------------------------
 * clamp(-1,1) -> (x+1)*127.5 -> uint8 HWC, the conversion real3d_infer.py:495-521 does on the host
 * after the frame loop; used to build the per-rank frame ring that is gathered over RCCL.
 * img [N,3,H,W] fp32 -> out [N,H,W,3] uint8. */
int r3d_frames_to_u8(const float* img, int N, int H, int W, uint8_t* out, r3d_stream_t stream);

/* --- frame gather over RCCL / xGMI ---------------------------------------------------------------------------
 * One process per GPU renders a contiguous chunk of the clip into a device-resident uint8 ring; r3d_gather_frames re-assembles the
 * clip on `root`: every rank sends `bytes_per_rank` bytes of `local`, the root receives world x bytes_per_rank into root_buf in rank
 * order (grouped ncclSend / ncclRecv on `stream`; no reduction, each peer uses its own xGMI link to the root).  The reference has
 * no counterpart: its frame loop is serial (inference/real3d_infer.py:480-492).
 *   r3d_comm_unique_id: rank 0 fills a 128-byte id that the caller ships to the other ranks (any side channel);
 *   r3d_comm_init: collective, on the calling thread's current HIP device.  RCCL is resolved with dlopen at first use (a copy that
 *   is already mapped, e.g. torch's, is reused); R3D_ERR_UNSUPPORTED if no librccl.so can be found. */
#define R3D_COMM_ID_BYTES 128
int r3d_comm_unique_id(void* id128);
int r3d_comm_init(const void* id128, int rank, int world, void** comm);
int r3d_comm_destroy(void* comm);
int r3d_gather_frames(void* comm, const uint8_t* local, size_t bytes_per_rank, uint8_t* root_buf, int root, r3d_stream_t stream);

/* Time a region with HIP events ON THE GIVEN STREAM (bench.py's roofline leg): returns elapsed ms
 * between two events; handles are opaque. */
int r3d_event_create(void** ev);
int r3d_event_record(void* ev, r3d_stream_t stream);
int r3d_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int r3d_event_destroy(void* ev);

/* --- per-kernel-family timing (bench.py's roofline leg) -------------------------------------------
 * When a family's bit is set in `mask`, every launch site of that family is bracketed by a HIP event pair
 * recorded on the launch stream.  r3d_profile_read() synchronises on the recorded events and returns the
 * summed duration and the number of bracketed launches since the last r3d_profile_reset(). */
enum r3d_prof_id {
    R3D_PROF_RENDER = 0,     /* render_kernel<..> (fused ray kernel)                       */
    R3D_PROF_CONV = 1,       /* conv_mfma_kernel (3x3 / transposed-conv implicit GEMM)     */
    R3D_PROF_UPCONV = 2,     /* up-sampling conv: fused transposed conv + FIR (f16x3); FIR (f32)  */
    R3D_PROF_TORGB = 3,      /* torgb_upsample_kernel                                      */
    R3D_PROF_PACK = 4,       /* SR weight modulation / packing kernels                     */
    R3D_PROF_LAYOUT = 5,     /* layout kernels (planes_to_nhwc, nchw<->cb8, frames_to_u8)  */
    R3D_PROF_MISC = 6,       /* raygen, limits, clamp, init                                */
    R3D_PROF_COUNT = 7
};
int r3d_profile_configure(uint32_t mask);
int r3d_profile_reset(void);
int r3d_profile_read(int id, double* total_ms, int* launches);
/* The shader clock the family's sampled kernels ran at since the last reset, in GHz (0 if none was sampled): one wave per launch of
 * render_kernel (RENDER), the plain 3x3 f16x3/f16mx conv (CONV) and the fused up-sampling conv (UPCONV) reads s_memtime and
 * s_memrealtime at its start and end -- cycles / ticks x hipDeviceAttributeWallClockRate.  The peaks bench.py prices against assume the
 * 2.4 GHz peak engine clock; under an MFMA-dense kernel the power management holds less.  `cycles` (optional): the summed s_memtime
 * cycles of the sampled waves.  Synchronises the device.  Sample with ONE stream (the start stamps are per family, not per launch). */
int r3d_profile_clock(int id, double* ghz, unsigned long long* cycles);

#ifdef __cplusplus
}
#endif
#endif /* R3D_HIP_H */
