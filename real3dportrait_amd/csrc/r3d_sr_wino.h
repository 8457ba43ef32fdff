// Winograd F(2,3) along x for the plain 3x3 convolutions of the SR blocks (block0.conv1, block1.conv1: modulated_conv2d of
// modules/eg3ds/models/networks_stylegan2.py:37-94 with up = down = 1, padding 1) -- included by r3d_sr_f16x3.hip.
//
// Per pair of output columns (x, x+1) and kernel row ky the direct form spends 6 tap-products, F(2,3) spends 4:
//     d0..d3 = in(x-1 .. x+2)          V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3          (input transform, fp32)
//     g0..g2 = w[ky][0..2]             U0 = g0        U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2     (weight transform, at prepack)
//     M_p = sum_{ci, ky} U_p[ky] V_p(row + ky)          y(x) = M0 + M1 + M2      y(x+1) = M1 - M2 - M3   (output transform, epilogue)
// i.e. four independent "3x1 convolutions" over half-width images V_p: 12 instead of 18 matrix products per column pair.
//
// Why the shape below.  A wave that held all four positions of its pixels would use every A (weight) and B (V) operand
// for ONE 32x32 tile: 6 ds_read_b128 per (f16 + f16 + fp8) tile unit, 75 % of the CU's LDS bandwidth before any transform traffic.
// So a wave owns ONE position: wave w = 4 wm + pos computes M_pos for 64 couts (wm) x all 128 column pairs of the 16 x 16-pixel tile
// = 2 x 4 tiles of 32 x 32 = 128 accumulator registers, two waves per SIMD, one block per CU.  Then
//   * B (V_pos) is shared by two waves, A (U_pos of the wave's 64 couts) by NOBODY: the weights go global -> VGPR directly (coalesced
//     16-byte loads of a prepacked per-wave stream, L2-resident), never through LDS;
//   * the input transform is done once per block: the raw SPLIT patch (fp16 hi + lo) is staged in LDS (global -> VGPR -> ds_write, half a stage
//     apart: every load of the kernel is visible to hipcc's vmcnt bookkeeping), each wave transforms the V rows of ITS position (its 8-channel chunk
//     wm of the 16-channel stage) and writes V as fp16 hi + {fp16 lo | e5m2 records};
//   * the four positions of a pixel meet in the epilogue through LDS (two rounds of 128 KB), after which the wave holds the ordinary
//     64 couts x 64 pixels of conv_epilogue, in the direct kernel's lane <-> pixel map.
// K stage = 16 input channels.  f16mx: per (ky, tile) one f16 MFMA (hi * hi) and the cross products of (ky0 | ky1) and (ky2 | zero) on
// the K = 64 fp8 MFMA (lane half h <-> kernel row h): 3 x 32 + 2 x 64 = 224 matrix cycles per (stage, tile) against 2 x 9 x 64 / 4 = 288
// for the same outputs in the direct kernel (1.29 x; pairing ky2 with the next stage's ky0 would make it 192).  f16x3: 9 f16 MFMAs = 288
// against 432 (1.5 x).
// Range: the fold guarantees |x| < 2^15; V' = (a +- b) / 2 keeps |V'| < 2^15 and U' = U 2^(kw-1) keeps |U'| < 2^11 (|U| <= 1.5 max|g|);
// the factor 4 is taken out with the per-cout epilogue multiplier (exact).
// The transform of stage s + 1 rides inside the matrix pipeline of stage s in three 8-channel pieces (stage_body).
// Measured (profiles/r06/winograd_kernel_findings.txt): f16x3 -13.7 % per launch against the direct kernel (the default there), f16mx +9 % (instruction-issue
// bound at two waves per SIMD: the transform is 180 VALU instructions per wave and stage; stays on the direct kernel, also end to end: 1 650 vs 1 570 frames/s).
// WG_X_* / WG_BD / WG_PD / WG_RL / WG_RS: experiment switches of that analysis (timing-only variants, prefetch depths); the defaults are the product.

static constexpr int WG_VPLANE = 18 * 8;                            // uint4 slots of one V plane: [V row 18][column pair 8]
static constexpr int WG_VPOS = 4 * WG_VPLANE;                       // per position: hi chunk 0 | hi chunk 1 | (lo chunk 0 | lo chunk 1) or (rec_h | rec_l)
static constexpr int WG_VBUF = 4 * WG_VPOS;                         // 2304 uint4 = 36 KB per stage
static constexpr int WG_RPLANE = 18 * 8;                            // raw patch plane k = (hi|lo, chunk, column parity): [row 18][8], column index rotated by k
static constexpr int WG_REXTRA = 8 * WG_RPLANE;                     // the ninth column index of every plane: [row 18][plane 8]
static constexpr int WG_RBUF = 9 * WG_RPLANE;                       // 1296 uint4 = 20.25 KB per stage
static constexpr int WG_RSEGS = (WG_RBUF + 63) / 64;                // 21 segments of 64 slots (one per wave and store instruction)
static constexpr int WG_RSTRIDE = WG_RSEGS * 64;                    // 1344: the last segment's tail lanes land in padding
static constexpr int WG_LDS_UINT4 = 8192;                           // main loop: 2 x 2304 + 2 x 1344 = 7296; epilogue exchange: 8192 (128 KB)
#ifndef WG_X_PIECE
#define WG_X_PIECE 0                // timing experiments: 1 = transform pieces without their LDS traffic, 2 = without their VALU work
#endif
#ifndef WG_PD
#define WG_PD 3                     // a transform piece is consumed this many (odd: 1 or 3) steps after its LDS reads were issued
#endif
#ifndef WG_BD
#define WG_BD 1                     // B operands are read this many steps ahead of their MFMAs
#endif
#ifndef WG_RL
#define WG_RL 4
#define WG_RS 10
#endif
static constexpr int WG_WBLK = 768;                                 // uint4 of weights per (cout tile, stage, wave)

// d = (float)half(x, HI) * m + c  and  d = (float)half(x, HI) * m + (float)half(c, HI): hipcc selects v_fma_mix_f32 for these (one instruction instead of
// v_cvt_f32_f16 + v_fma_f32) as long as the multiplier is not a literal it can fold (+-1).  Plain expressions, not inline asm: hipcc's hazard recogniser
// does not look inside an asm statement, and with the VALU work interleaved between MFMAs (sched_group_barrier) the asm form computed garbage.
typedef _Float16 wg_hh2 __attribute__((ext_vector_type(2)));
template <int HI>
__device__ __forceinline__ float mix_hf(unsigned x, float m, float c) { return __builtin_fmaf((float)__builtin_bit_cast(wg_hh2, x)[HI], m, c); }
template <int HI>
__device__ __forceinline__ float mix_hh(unsigned x, float m, unsigned c) { return __builtin_fmaf((float)__builtin_bit_cast(wg_hh2, x)[HI], m, (float)__builtin_bit_cast(wg_hh2, c)[HI]); }

// ---- weights: U = G g per kernel row, times 2^(kw[co] - 1), split and laid out as the per-wave streams the kernel loads ----------------
// block (ct, st, w = 4 wm + pos) of WG_WBLK uint4, cout = 128 ct + 64 wm + 32 mt + li, channels 16 st ..:
//   f16x3:  [(ky 3, mt 2, hi|lo 2)][lane 64]                          lane = 32 chunk + li
//   f16mx:  [(ky 3, mt 2)][lane 64] hi   ++   [(mt 2, wl8|wh8 2)][lane 64] records of kernel row h = lane / 32 (ky 0 | ky 1)
//                                        ++   [(mt 2, wl8|wh8 2)][li 32]   records of ky 2 (lanes h = 1 of that MFMA read a zero block)
// Record bytes: channel j of the stage at byte j (the V records of the kernel use the same order).
// (w is [CoutReal][CinReal][3][3]; Cin / Cout are the padded sizes of the pack -- multiples of 16 / 128 --, channels beyond the real ones are zero)
__global__ void sr_prepack_wino_kernel(const float* __restrict__ w, int CinReal, int CoutReal, int Cin, int Cout, const float* __restrict__ winv, uint4* __restrict__ out, int mx)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nst = Cin >> 4;
    const size_t total = (size_t)(Cout >> 7) * nst * 8 * 3 * 2 * 32;
    if (e >= total) return;
    const int li = e & 31, mt = (e >> 5) & 1;
    const int ky = (int)((e >> 6) % 3);
    size_t r = (e >> 6) / 3;
    const int w8 = (int)(r & 7); r >>= 3;
    const int st = (int)(r % nst), ct = (int)(r / nst);
    const int pos = w8 & 3, wm = w8 >> 2;
    const int co = ct * 128 + wm * 64 + mt * 32 + li;
    const double ws = 0.5 / (double)winv[co];                       // 2^(kw - 1)
    float u[16], hif[16], lof[16];
    h8 hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ci = st * 16 + j;
        const bool real = co < CoutReal && ci < CinReal;
        const float* g = w + (((size_t)(real ? co : 0) * CinReal + (real ? ci : 0)) * 3 + ky) * 3;
        const double g0 = real ? g[0] : 0.0, g1 = real ? g[1] : 0.0, g2 = real ? g[2] : 0.0;
        const double uu = pos == 0 ? g0 : (pos == 1 ? 0.5 * (g0 + g1 + g2) : (pos == 2 ? 0.5 * (g0 - g1 + g2) : g2));
        u[j] = (float)(uu * ws);
        _Float16 a, b; split1(u[j], a, b);
        hi[j >> 3][j & 7] = a; lo[j >> 3][j & 7] = b;
        hif[j] = (float)a; lof[j] = u[j] - (float)a;
    }
    uint4* blk = out + ((size_t)(ct * nst + st) * 8 + w8) * WG_WBLK;
    if (!mx) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            blk[((ky * 2 + mt) * 2 + 0) * 64 + c * 32 + li] = *reinterpret_cast<uint4*>(&hi[c]);
            blk[((ky * 2 + mt) * 2 + 1) * 64 + c * 32 + li] = *reinterpret_cast<uint4*>(&lo[c]);
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) blk[(ky * 2 + mt) * 64 + c * 32 + li] = *reinterpret_cast<uint4*>(&hi[c]);
    uint4 wl8, wh8;
    unsigned* pl = reinterpret_cast<unsigned*>(&wl8); unsigned* phh = reinterpret_cast<unsigned*>(&wh8);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        pl[q] = pack4_fp8(lof[4 * q] * kMxWl, lof[4 * q + 1] * kMxWl, lof[4 * q + 2] * kMxWl, lof[4 * q + 3] * kMxWl);
        phh[q] = pack4_fp8(hif[4 * q] * kMxWh, hif[4 * q + 1] * kMxWh, hif[4 * q + 2] * kMxWh, hif[4 * q + 3] * kMxWh);
    }
    if (ky < 2) {
        blk[384 + (mt * 2 + 0) * 64 + ky * 32 + li] = wl8;
        blk[384 + (mt * 2 + 1) * 64 + ky * 32 + li] = wh8;
    } else {
        blk[640 + (mt * 2 + 0) * 32 + li] = wl8;
        blk[640 + (mt * 2 + 1) * 32 + li] = wh8;
    }
}

template <bool MX>
__global__ __launch_bounds__(512, 2) void conv_wino_f16x3_kernel(Conv2Args a)
{
    __shared__ uint4 lds[WG_LDS_UINT4];
    const ConvPhase& ph = a.ph[0];
    const int n = blockIdx.z;
    const int tiles_x = (ph.outW + F_TILE_W - 1) / F_TILE_W;
    int tile = blockIdx.x, cgi = blockIdx.y;
    if (a.order) {                                                   // XCD-aware block order of conv3x3_dma_block
        const int G = gridDim.y, b = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = b & 7, slot = b >> 3;
        cgi = slot % G;
        const int t = slot / G;
        tile = a.order == 1 ? t * 8 + xcd : xcd * (gridDim.x >> 3) + t;
    }
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int i0 = ty * F_TILE_H, j0 = tx * F_TILE_W;
    const int m0 = cgi * BLOCK_M;
    const int lane = threadIdx.x & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos = wave_u & 3, wm = wave_u >> 2;
    const int li = lane & 31, h = lane >> 5;
    const int nst = a.Cin >> 4, nchunks = a.Cin >> 3;
    const int chunk_stride = a.H * a.W;
    const size_t plane = (size_t)nchunks * chunk_stride;
    const uint4* X = a.x + (size_t)n * a.x_stride_n;
    if (blockIdx.x == (gridDim.x >> 1) && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) clk_begin(kernarg_clk<Conv2Args>());

    // ---- raw patch staging: slot e of a raw buffer <-> (plane k = 4 hl + 2 chunk + parity, row, column index idx), patch column 2 idx + parity.
    // Global -> VGPR (buffer loads: out-of-image slots point past num_records and read zeros) -> ds_write_b128, half a stage apart, NOT LDS-DMA:
    // an LDS-DMA issued from inline asm is invisible to hipcc's vmcnt bookkeeping, and every counted wait hipcc emits for a weight load older than
    // the DMA then waits for the DMA too -- the first version stalled on the patch's HBM latency at the top of every stage.
    int pf_voff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = (8 * k + wave_u) * 64 + lane;
        int off = (int)0x80000000;
        if (e < WG_RBUF) {
            int kp, row, idx;
            if (e < WG_REXTRA) { kp = e / WG_RPLANE; const int rem = e - kp * WG_RPLANE; row = rem >> 3; idx = ((rem & 7) - kp) & 7; }
            else { const int rem = e - WG_REXTRA; row = rem >> 3; kp = rem & 7; idx = 8; }
            const int hl = kp >> 2, c = (kp >> 1) & 1, par = kp & 1;
            const int iy = i0 - 1 + row, ix = j0 - 1 + 2 * idx + par;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) off = (int)(((unsigned)(hl * plane) + (unsigned)(c * chunk_stride + iy * a.W + ix)) * 16u);
        }
        pf_voff[k] = off;
    }
    const bool three = wave_u < WG_RSEGS - 16;                        // waves 0..4 stage three 64-slot segments per stage, the others two
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(X), 0, (unsigned)(2 * plane) * 16u, 0x00020000);   // host: a sample is < 2 GB
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto raw_load = [&](int st, u32x4 (&rr)[3]) {
        const unsigned soff = (unsigned)(2 * st * chunk_stride) * 16u;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < 2 || three) rr[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, pf_voff[k], (int)soff, 0);
    };
    auto raw_store = [&](uint4* dst, const u32x4 (&rr)[3]) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < 2 || three) dst[64 * (8 * k + wave_u) + lane] = make_uint4(rr[k][0], rr[k][1], rr[k][2], rr[k][3]);
    };
    uint4* const vbuf = lds;                                          // [2][WG_VBUF]
    uint4* const rbuf = lds + 2 * WG_VBUF;                            // [2][WG_RSTRIDE]

    // ---- input transform of this wave: position pos, chunk wm; item = (V row, column pair), 3 passes of 8 rows (the last: rows 16, 17)
    // V = A + sgn B with (parity, column index offset) of A and B:  pos 0: (0,0) - (0,1)   1: (1,0) + (0,1)   2: (0,1) - (1,0)   3: (1,0) - (1,1)
    const int parA = (pos == 1 || pos == 3) ? 1 : 0, offA = pos == 2 ? 1 : 0;
    const int parB = (pos >= 2) ? 1 : 0, offB = (pos == 2) ? 0 : 1;
    const float sgnh = pos == 1 ? 0.5f : -0.5f;
    float halfv = 0.5f, neg1 = -1.0f, sgnv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(pos == 1 ? 0x3f000000 : (int)0xbf000000));
    asm volatile("" : "+s"(halfv), "+s"(neg1), "+s"(sgnv));           // multipliers of the mix ops: one SGPR operand each (constant-bus limit 1)
    typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const hh2 half2 = {(_Float16)0.5f, (_Float16)0.5f};
    const hh2 sgn2 = {(_Float16)sgnh, (_Float16)sgnh};
    const int r8 = lane >> 3, p8 = lane & 7;
    auto raw_slot = [&](int kp, int off) {                            // slot of (plane kp, row r8, column index p8 + off) for pass 0
        const int idx = p8 + off;
        return idx < 8 ? kp * WG_RPLANE + r8 * 8 + ((idx + kp) & 7) : WG_REXTRA + r8 * 8 + kp;
    };
    const int kA = 2 * wm + parA, kB = 2 * wm + parB;                 // hi planes; + 4 for the lo planes
    const int sAh = raw_slot(kA, offA), sAl = raw_slot(kA + 4, offA), sBh = raw_slot(kB, offB), sBl = raw_slot(kB + 4, offB);
    const int vw_slot = pos * WG_VPOS + r8 * 8 + p8;                  // + plane * WG_VPLANE + pass * 64
    const int d2 = (16 + (r8 & 1) - r8) * 8 - 128;                    // pass 2: every lane takes row 16 + (r8 & 1) (rows 16, 17 eight times over: same values to the same slots)

    // The transform of a stage in three PIECES (pass p: V rows 8 p .. 8 p + 7, pass 2 = rows 16, 17; the wave's 8-channel chunk): four 16-byte LDS reads
    // issued in step 2 p of the matrix pipeline below, ~70 VALU instructions spread between the MFMAs of step 2 p + 1, then one 16-byte and two 8-byte
    // LDS writes.  (A version with six 4-channel pieces -- 8-byte reads, 8- and 4-byte writes -- spent 2 100 LDS cycles per stage and CU on them; this 980.)
    // Unconditional (the last stage transforms stale bytes into a V buffer nobody reads): a branch would cut the scheduling region the interleave needs.
    uint4 tr[2][4];                                                   // two pieces in flight: piece p is read in step 2 p and consumed WG_PD steps later
    auto tp_load = [&](int bufi, int pass) {
        const uint4* R = rbuf + bufi * WG_RSTRIDE;
        const int o = pass == 2 ? d2 : 0;
#if WG_X_PIECE == 1
        (void)R; (void)o;
#pragma unroll
        for (int q = 0; q < 4; ++q) { tr[pass & 1][q] = make_uint4(0x3c003c00u + lane, 0x3c003c00u + pass, 0x38003800u, 0x34003400u + q); asm volatile("" : "+v"(tr[pass & 1][q].x), "+v"(tr[pass & 1][q].y), "+v"(tr[pass & 1][q].z), "+v"(tr[pass & 1][q].w)); }
#else
        tr[pass & 1][0] = R[sAh + o + pass * 64]; tr[pass & 1][1] = R[sAl + o + pass * 64]; tr[pass & 1][2] = R[sBh + o + pass * 64]; tr[pass & 1][3] = R[sBl + o + pass * 64];
#endif
    };
    auto tp_compute = [&](int bufi, int pass) {
        uint4* V = vbuf + bufi * WG_VBUF + vw_slot + pass * 64 + (pass == 2 ? d2 : 0);
        const unsigned* pah = reinterpret_cast<const unsigned*>(&tr[pass & 1][0]); const unsigned* pal = reinterpret_cast<const unsigned*>(&tr[pass & 1][1]);
        const unsigned* pbh = reinterpret_cast<const unsigned*>(&tr[pass & 1][2]); const unsigned* pbl = reinterpret_cast<const unsigned*>(&tr[pass & 1][3]);
#if WG_X_PIECE == 2
        V[wm * WG_VPLANE] = make_uint4(pah[0] ^ pbh[0], pah[1] ^ pbh[1], pah[2] ^ pbh[2], pah[3] ^ pbh[3]);
        reinterpret_cast<uint2*>(V + 2 * WG_VPLANE)[wm] = make_uint2(pal[0], pal[1]);
        reinterpret_cast<uint2*>(V + 3 * WG_VPLANE)[wm] = make_uint2(pbl[0], pbl[1]);
        return;
#endif
        float v[8], lo[8];
        unsigned hw[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const hh2 la = __builtin_bit_cast(hh2, pal[d]) * half2;
            const hh2 ls = __builtin_elementwise_fma(__builtin_bit_cast(hh2, pbl[d]), sgn2, la);       // (lo_a + sgn lo_b) / 2, fp16 (2^-22 of x)
            const unsigned lsu = __builtin_bit_cast(unsigned, ls);
            const float t0 = mix_hh<0>(pbh[d], sgnv, lsu), t1 = mix_hh<1>(pbh[d], sgnv, lsu);
            v[2 * d] = mix_hf<0>(pah[d], halfv, t0);
            v[2 * d + 1] = mix_hf<1>(pah[d], halfv, t1);
            const hh2 hi2 = __builtin_convertvector((f2){v[2 * d], v[2 * d + 1]}, hh2);
            hw[d] = __builtin_bit_cast(unsigned, hi2);
            lo[2 * d] = mix_hf<0>(hw[d], neg1, v[2 * d]);
            lo[2 * d + 1] = mix_hf<1>(hw[d], neg1, v[2 * d + 1]);
        }
#if WG_X_PIECE == 1
        {
            const unsigned r0 = pack4_x8(v[0], v[1], v[2], v[3]), r1 = pack4_x8(v[4], v[5], v[6], v[7]), r2 = pack4_x8(lo[0] * kMxXl, lo[1] * kMxXl, lo[2] * kMxXl, lo[3] * kMxXl), r3 = pack4_x8(lo[4] * kMxXl, lo[5] * kMxXl, lo[6] * kMxXl, lo[7] * kMxXl);
            asm volatile("" :: "v"(hw[0]), "v"(hw[1]), "v"(hw[2]), "v"(hw[3]), "v"(r0), "v"(r1), "v"(r2), "v"(r3));
            return;
        }
#endif
        V[wm * WG_VPLANE] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        if constexpr (MX) {
            // e5m2 records of V (hi part straight from the fp32 value) and of lo * 2^11: chunk wm = bytes [8 wm, +8) of the two 16-byte records of the stage
            reinterpret_cast<uint2*>(V + 2 * WG_VPLANE)[wm] = make_uint2(pack4_x8(v[0] * kMxXh, v[1] * kMxXh, v[2] * kMxXh, v[3] * kMxXh), pack4_x8(v[4] * kMxXh, v[5] * kMxXh, v[6] * kMxXh, v[7] * kMxXh));
            reinterpret_cast<uint2*>(V + 3 * WG_VPLANE)[wm] = make_uint2(pack4_x8(lo[0] * kMxXl, lo[1] * kMxXl, lo[2] * kMxXl, lo[3] * kMxXl), pack4_x8(lo[4] * kMxXl, lo[5] * kMxXl, lo[6] * kMxXl, lo[7] * kMxXl));
        } else {
            unsigned lw[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) lw[d] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo[2 * d], lo[2 * d + 1]}, hh2));
            V[(2 + wm) * WG_VPLANE] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
    };
    auto transform = [&](int bufi) {                                  // raw[bufi] -> V[bufi] in one go (prologue)
#pragma unroll
        for (int k = 0; k < 3; ++k) { tp_load(bufi, k); tp_compute(bufi, k); }
    };

    // ---- accumulators: M_pos of couts [64 wm, +64) x (16 rows x 8 column pairs) = acc[mt][nt], N tile nt = rows 4 nt .. 4 nt + 3
    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // A operands: this wave's stream [ct][st][w] (WG_WBLK uint4 per stage), per-lane pointer
    // (buffer loads: one VGPR of lane offset for every load of the kernel, the rest of the address on the scalar unit; lanes that must read
    // zeros -- the second K half of the unpaired ky 2 records -- point past num_records)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.wp), 0, (unsigned)((a.Cout >> 7) * nst * 8 * WG_WBLK) * 16u, 0x00020000);
    const unsigned wa_blk = (unsigned)((cgi * nst * 8 + wave_u) * WG_WBLK) * 16u;      // byte offset of this wave's block of stage 0 (host: the pack is < 2 GB)
    const unsigned wa_stage = 8u * WG_WBLK * 16u;
    const int lane16 = lane * 16, z16 = h ? (int)0x80000000 : li * 16;
    auto ldw = [&](int voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, (int)soff, 0); };
    auto ldw_h8 = [&](int voff, unsigned soff) { return __builtin_bit_cast(h8, ldw(voff, soff)); };
    auto ldw_i8 = [&](int voff, unsigned s0, unsigned s1) {
        const auto q0 = ldw(voff, s0), q1 = ldw(voff, s1);
        return (i8v){(int)q0[0], (int)q0[1], (int)q0[2], (int)q0[3], (int)q1[0], (int)q1[1], (int)q1[2], (int)q1[3]};
    };
    auto ld_h8 = [](const uint4* p) { const uint4 q = *p; return *reinterpret_cast<const h8*>(&q); };
    auto ld_i8 = [](const uint4* p0, const uint4* p1) {
        const uint4 q0 = *p0, q1 = *p1;
        return (i8v){(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, (int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w};
    };
    // carried across stages (16 registers): MX: the f16 rows ky 0, ky 1;  f16x3: hi and lo of ky 0
    h8 ca[2], cb[2];
    auto load_a_carried = [&](int st) {
        const unsigned W = wa_blk + (unsigned)st * wa_stage;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if constexpr (MX) { ca[mt] = ldw_h8(lane16, W + ((0 * 2 + mt) * 64) * 16); cb[mt] = ldw_h8(lane16, W + ((1 * 2 + mt) * 64) * 16); }
            else { ca[mt] = ldw_h8(lane16, W + (((0 * 2 + mt) * 2 + 0) * 64) * 16); cb[mt] = ldw_h8(lane16, W + (((0 * 2 + mt) * 2 + 1) * 64) * 16); }
        }
    };
    const int boff = pos * WG_VPOS + li;                              // B slot: + plane * WG_VPLANE + (4 nt + ky) * 8

    // One stage = a pipeline of STEPS of 128 (f16mx) / 192 (f16x3) matrix cycles.  Step s: the B operands of step s + 1 are read (and, at a few
    // steps, A operands loaded ~512 matrix cycles ahead of their use; and the LDS reads of transform piece s issued) | the step's MFMAs | the VALU
    // part of transform piece s (next stage's V).  The fences keep hipcc from hoisting every read and load of the stage to its top (485 spilled
    // registers in the first version).  Both waves of a SIMD run the same sequence and cover each other's LDS / L2 latencies.
    // (First version: one wave's whole transform under the other's whole matrix phase, the waves of a SIMD in opposite orders -- a single wave
    // then has to keep the matrix pipe fed alone, and a B read issued one step ahead does not arrive in time: the matrix phase took 2.06 x its MFMA time.)
    auto stage_body = [&](int st, int bufi) {
        const uint4* V = vbuf + bufi * WG_VBUF + boff;
        const unsigned W = wa_blk + (unsigned)st * wa_stage;
        const int nxt = bufi ^ 1;
        u32x4 rr[3];                                                  // raw patch of stage st + 2 on its way to raw[bufi] (which the pieces of stage st - 1 have finished with)
#define WG_SB __builtin_amdgcn_sched_barrier(0)
        if constexpr (MX) {
            // steps: F(ky 0) x 2 | F(ky 1) x 2 | E(ky0 | ky1) x 4 | F(ky 2) x 2 | E(ky2 | zero) x 4;  F = f16 hi * hi on two N tiles, E = fp8 cross products on one
            h8 bf[WG_BD + 1][2], ak2[2];
            i8v b8[WG_BD + 1], a8p0[2], a8p1[2];
#ifdef WG_X_NOA
            ak2[0] = ca[0]; ak2[1] = ca[1]; a8p0[0] = a8p0[1] = a8p1[0] = a8p1[1] = (i8v){lane, li, h, 1, 2, 3, 4, 5};
#endif
            auto rd_step = [&](int s) {                                // the B operands of step s
                const bool isE = (s >= 4 && s < 8) || s >= 10;
                if (!isE) {
                    const int ky = s < 4 ? s >> 1 : 2, np = s & 1;
#pragma unroll
                    for (int j = 0; j < 2; ++j) bf[s % (WG_BD + 1)][j] = ld_h8(V + h * WG_VPLANE + (4 * (2 * np + j) + ky) * 8);
                } else {
                    const int nt = s < 8 ? s - 4 : s - 10;
                    const int kyh = s < 8 ? h : 2;                    // kernel row of this lane half
                    b8[s % (WG_BD + 1)] = ld_i8(V + 2 * WG_VPLANE + (4 * nt + kyh) * 8, V + 3 * WG_VPLANE + (4 * nt + kyh) * 8);
                }
            };
#pragma unroll
            for (int s = 0; s < WG_BD; ++s) rd_step(s);
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                WG_SB;
                if (s + WG_BD < 14) rd_step(s + WG_BD);
#ifndef WG_X_NOA
                if (s == 0) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) a8p0[mt] = ldw_i8(lane16, W + (384 + (mt * 2 + 0) * 64) * 16, W + (384 + (mt * 2 + 1) * 64) * 16);
                }
                if (s == 4) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) ak2[mt] = ldw_h8(lane16, W + ((2 * 2 + mt) * 64) * 16);
                }
                if (s == 6) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)                    // ky 2 alone in its pair: lanes h = 1 (the second K half) read zeros (out of range)
                        a8p1[mt] = ldw_i8(z16, W + (640 + (mt * 2 + 0) * 32) * 16, W + (640 + (mt * 2 + 1) * 32) * 16);
                }
                if (s == 10) load_a_carried(min(st + 1, nst - 1));      // (last stage: loads its own operands again, unused)
#endif
#ifndef WG_X_NORAW
                if (s == WG_RL) raw_load(min(st + 2, nst - 1), rr);
                if (s == WG_RS) raw_store(rbuf + bufi * WG_RSTRIDE, rr);
#endif
#ifndef WG_X_NOPIECE
                if (s < 6 && !(s & 1)) tp_load(nxt, s >> 1);
#endif
                const bool isE = (s >= 4 && s < 8) || s >= 10;
                if (!isE) {
                    const int np = s & 1;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            const h8 A = s < 2 ? ca[mt] : (s < 4 ? cb[mt] : ak2[mt]);
                            acc[mt][2 * np + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, bf[s % (WG_BD + 1)][j], acc[mt][2 * np + j], 0, 0, 0);
                        }
                } else {
                    const int nt = s < 8 ? s - 4 : s - 10;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(s < 8 ? a8p0[mt] : a8p1[mt], b8[s % (WG_BD + 1)], acc[mt][nt], 0, kMxFmtB, 0, kMxScaleA, 0, kMxScaleB);
                }
#ifndef WG_X_NOPIECE
                if (s >= WG_PD && s < 6 + WG_PD && ((s - WG_PD) & 1) == 0) tp_compute(nxt, (s - WG_PD) >> 1);
#endif
#ifndef WG_X_NOSGB
                // order inside the step: LDS reads and global loads first, then the VALU work of the piece spread between the MFMAs, LDS writes last
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
                if (!isE) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 19, 0); }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 38, 0); }
                }
                __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
#endif
            }
            WG_SB;
        } else {
            // f16x3: step = (kernel row ky, N tile nt): lo(U) hi(V) + hi(U) lo(V) + hi(U) hi(V) on both cout tiles = 6 MFMAs
            h8 bh[2], bl[2], ah1[2], al1[2], ah2[2], al2[2];
            auto rd_step = [&](int s) {
                const int ky = s >> 2, nt = s & 3;
                bh[s & 1] = ld_h8(V + h * WG_VPLANE + (4 * nt + ky) * 8);
                bl[s & 1] = ld_h8(V + (2 + h) * WG_VPLANE + (4 * nt + ky) * 8);
            };
            rd_step(0);
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                WG_SB;
                if (s + 1 < 12) rd_step(s + 1);
                if (s == 0) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) { ah1[mt] = ldw_h8(lane16, W + (((1 * 2 + mt) * 2 + 0) * 64) * 16); al1[mt] = ldw_h8(lane16, W + (((1 * 2 + mt) * 2 + 1) * 64) * 16); }
                }
                if (s == 4) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) { ah2[mt] = ldw_h8(lane16, W + (((2 * 2 + mt) * 2 + 0) * 64) * 16); al2[mt] = ldw_h8(lane16, W + (((2 * 2 + mt) * 2 + 1) * 64) * 16); }
                }
                if (s == 8) load_a_carried(min(st + 1, nst - 1));
                if (s == 6) raw_load(min(st + 2, nst - 1), rr);
                if (s == 10) raw_store(rbuf + bufi * WG_RSTRIDE, rr);
                if (s < 6 && !(s & 1)) tp_load(nxt, s >> 1);
                const int ky = s >> 2, nt = s & 3;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const h8 AH = ky == 0 ? ca[mt] : (ky == 1 ? ah1[mt] : ah2[mt]), AL = ky == 0 ? cb[mt] : (ky == 1 ? al1[mt] : al2[mt]);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, bh[s & 1], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, bl[s & 1], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, bh[s & 1], acc[mt][nt], 0, 0, 0);
                }
                if (s >= WG_PD && s < 6 + WG_PD && ((s - WG_PD) & 1) == 0) tp_compute(nxt, (s - WG_PD) >> 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
#pragma unroll
                for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 13, 0); }
                __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
            }
            WG_SB;
        }
#undef WG_SB
    };

    R3D_STAMP_DECL;
    // ---- prologue: raw(0), raw(1) -> LDS; first-half operands of stage 0; V(0)
    {
        u32x4 r0[3], r1[3];
        raw_load(0, r0);
        if (nst > 1) raw_load(1, r1);
        load_a_carried(0);
        raw_store(rbuf, r0);
        if (nst > 1) raw_store(rbuf + WG_RSTRIDE, r1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    transform(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    R3D_STAMP(4);
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        stage_body(st, cur);
        R3D_STAMP(5);
        // V(st + 1) and raw(st + 2) written, this wave's reads of V(st) / raw(st + 1) complete -> everybody's (the carried weight loads stay in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        R3D_STAMP(7);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        R3D_STAMP(8);
    }

    // ---- the four positions of a pixel meet: two rounds (mt) through LDS; wave (pos, wm) ends up with output rows [4 pos, +4) of its couts
    f32x16 out[2][2];
#ifdef WG_NO_EXCHANGE
    for (int mt = 0; mt < 2; ++mt) for (int q = 0; q < 2; ++q) out[mt][q] = acc[mt][q] + acc[mt][q + 2];
#else
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        f32x4* Xc = reinterpret_cast<f32x4*>(lds);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Xc[((wave_u * 4 + nt) * 4 + q) * 64 + lane] = (f32x4){acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
        __syncthreads();
        // read back in the pixel map of conv_epilogue (lane li <-> row 2 nt + li / 16, column (li % 16 - 2 (li / 16)) % 16 of the wave's four rows): a
        // lane takes ONE column of its pair -- even: (M1 + M2) + M0, odd: (M1 - M2) - M3 -- so the stores below are the direct kernel's contiguous
        // 16-pixel rows (with the lane <-> column-pair map the first version kept, every store instruction wrote every other pixel: 59 k cycles of epilogue)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int prow = li >> 4, pcol = ((li & 15) - 2 * prow) & 15, par = pcol & 1;
            const int lw = (2 * nt + prow) * 8 + (pcol >> 1) + 32 * h;
            const float sg = par ? -1.0f : 1.0f;
            const f32x4* Xr = Xc + (size_t)((wm * 4) * 4 + pos) * 4 * 64 + lw;               // + p * 1024 + q * 64
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 R0 = Xr[(par ? 3 : 0) * 1024 + q * 64], M1 = Xr[1 * 1024 + q * 64], M2 = Xr[2 * 1024 + q * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) out[mt][nt][4 * q + r] = __builtin_fmaf(sg, R0[r], __builtin_fmaf(sg, M2[r], M1[r]));
            }
        }
        __syncthreads();
    }
#endif
    R3D_STAMP(9);
    conv_epilogue<true, 4, 2, 2>(a, ph, n, out, i0, j0, m0, reinterpret_cast<float*>(lds));
    R3D_STAMP(10);
    if (blockIdx.x == (gridDim.x >> 1) && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) clk_end(kernarg_clk<Conv2Args>());
#ifdef R3D_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    R3D_STAMP(11);
    if ((threadIdx.x & 63) == 0) { for (int i_ = 4; i_ < 12; ++i_) atomicAdd(&r3d::g_stamps[8 + i_], st_acc_[i_]); atomicAdd(&r3d::g_stamps[30], 1ull); }
    R3D_STAMP_CLOCKS(26);
#endif
}
