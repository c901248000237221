// Pointwise (1x1) convolution of float32 plans on the 16-BIT matrix pipe, float32-grade (round 4; see pointwise.hip for the GEMM view
// and the dispatcher, mbr.hip "SPLIT form" for the arithmetic).  On gfx950 v_mfma_f32_16x16x4_f32 delivers 64 FLOP / clk / SIMD - the
// packed-FMA rate, on the same lanes - and the float32 GEMMs of the heads sat at 0.4-0.7 of that pipe; v_mfma_f32_16x16x32_f16 has
// 16 x the rate.  Every float32 operand is cut into two float16 planes, x = h + 2^-11 m (h = f16(x), m = f16((x - h) 2^11): 22
// significant bits, x - h exact), and a product takes three MFMAs - h h' into one accumulator, h m' + m h' into a second one that
// joins with 2^-11 in the epilogue (the dropped m m' is below 2^-24 |x| |w|): measured against float64 the results carry the same
// error as the float32-MFMA kernel's.  The planes are cut ONCE per element, on the way from the fetch registers into LDS
// (5 VALU operations per pair of values; weights too - their matrix stays float32 in the plan), so everything in front of the
// LDS store - gathers, concat, up-sampling, pooling, SE gate - is pw_kernel's code, and the epilogue is too.
// Precondition: |x|, |w| < 65504 (beyond it the planes are inf and the result NaN / inf - or, behind a ReLU6 epilogue, a clamped value).
// A 32-wide k chunk per barrier pair (one MFMA step); tile shapes and index as pw_kernel's.
#include "pws_common.h"

#define PWS_RPP (256 / PWS_KQ)     // rows loaded per pass of the 256 threads

// blocks per CU the register allocator must make room for (two accumulator sets per tile pair)
constexpr int pws_min_blocks(int pt, int ct) {
    const int tiles = pt * ct;
    return tiles >= 16 ? 1 : (tiles >= 8 || (pt == 4 && ct == 1)) ? 2 : tiles >= 4 ? 3 : 4;   // (256 x 16: eight fetch passes in two register sets)
}

template <int PT, int CT, int WM, int WN, bool SIMPLE>
__global__ __launch_bounds__(256, pws_min_blocks(PT, CT)) void pws_kernel(PwArgs a) {
    constexpr bool DW = false;
    constexpr int BM = 16 * PT * WM;
    constexpr int BN = 16 * CT * WN;
    constexpr int A_PASSES = BM / PWS_RPP;
    constexpr int B_PASSES = (BN + PWS_RPP - 1) / PWS_RPP;
    // two float16 planes per operand: [rows][PWS_LD halves] each (32 k + 8 halves of padding: rows 80 bytes apart)
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * (BM + BN) * PWS_LD];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];  // the tile's BN scale | shift (read by the epilogue)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    // 1-D grid walked in XCD-contiguous order with the cout tile fastest: the cout tiles of one pixel
    // tile run back to back on one XCD, so the activation tile is re-read from that XCD's L2.
    const unsigned ntn = (a.N + BN - 1) / BN;
    const unsigned L = yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (int)(L / ntn) * BM;
    const int n0 = (int)(L % ntn) * BN;
    const int kp = a.S.kp;

    // the tile's BatchNorm scale / shift go to LDS now (behind the k loop's barriers by the time they are read):
    // fetched in the epilogue they would cost every tile an L2 round trip with nothing left to hide it
    if (tid < BN) {
        const int n = n0 + tid < a.N ? n0 + tid : a.N - 1;
        ss[tid] = a.scale ? a.scale[n] : 1.f;
        ss[BN + tid] = a.shift ? a.shift[n] : 0.f;
    }

    // loader mapping: quad kq of row lr (+64 per pass)
    const int lr = tid / PWS_KQ;
    constexpr int MODE = SIMPLE ? 2 : 0;
    const bool gated = SIMPLE && !DW && a.gate != nullptr;
    PwRow<MODE> row[A_PASSES];
    pw_unroll<A_PASSES>([&](auto P) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        row[p].init(a, m0 + lr + p * PWS_RPP);
        if constexpr (SIMPLE && !DW)
            if (!gated) row[p].grow = a.wt;  // ungated: the gate load becomes a (cached, ignored) weight quad
    });
    const float* brow[B_PASSES];
    pw_unroll<B_PASSES>([&](auto P) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        const int n = n0 + lr + p * PWS_RPP;
        brow[p] = a.wt + (size_t)(n < a.N ? n : 0) * kp;  // rows beyond N feed couts that are never stored
    });

    const int g = lane >> 4, li = lane & 15;
    f32x4 acc[CT][PT], ac1[CT][PT];   // h h' | h m' + m h' (joins with 2^-11 in the epilogue)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) { acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f}; ac1[c][p] = acc[c][p]; }
    pws_k_loop<256, PT, CT, WM, WN, MODE, A_PASSES, B_PASSES>(a, row, brow, gated, lds, acc, ac1);

    // ---- epilogue: (pre-BN addend,) BN scale/shift, activation, (residual,) (2x2 max,) store: 4 consecutive couts
    // per lane.  Branches are uniform or guard stores only; every load is unconditional (pw_load_quad): a load under
    // a per-lane branch is followed by its own s_waitcnt, one L2 round trip per element group with nothing to hide it.
    const bool vec_out = (a.out_ld & 3) == 0;
    const bool vec_res = (a.res_ld & 3) == 0, vec_pre = (a.pre_ld & 3) == 0;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int nl = (wn * CT + c) * 16 + g * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + nl);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + BN + nl);
#pragma unroll
        for (int p = 0; p < PT; ++p)
            pw_finish_quad(a, __builtin_elementwise_fma(ac1[c][p], (f32x4){0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f}, acc[c][p]), sc, sh, m0 + (wm * PT + p) * 16 + li, n0 + nl, li, vec_out, vec_res, vec_pre);
    }
}



template <int PT, int CT, int WM, int WN>
static int launch_split_cfg(const PwArgs& a, hipStream_t s) {
    constexpr int BM = 16 * PT * WM, BN = 16 * CT * WN;
    dim3 grid((unsigned)((a.M + BM - 1) / BM) * (unsigned)((a.N + BN - 1) / BN));
    const bool simple = a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY;
    static char nm[2][48];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pws_kernel<%d,%d,%d,%d,0>", PT, CT, WM, WN) +
                              snprintf(nm[1], sizeof(nm[1]), "pws_kernel<%d,%d,%d,%d,1>", PT, CT, WM, WN);
    (void)nm_len;
    yr_note_kernel(nm[simple ? 1 : 0]);
    if (simple) hipLaunchKernelGGL((pws_kernel<PT, CT, WM, WN, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pws_kernel<PT, CT, WM, WN, false>), grid, dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// The K-SPLIT form (round 5, the passes of one or two images: se_reduced bit 17 of a POINTWISE op, set by the compiler's 'nohead_k'
// variant for the maps of the heads and the last backbone stages).  At 169 .. 2704 pixels a conv is a handful of workgroups, each ONE latency
// chain: pws_kernel walks its k chunks one barrier pair at a time, ~165 instructions of staging per chunk and wave for three MFMAs
// (block_14_project, 720 deep: 23 chunks, 15 us; several chunks per barrier pair with all their loads in flight - built, bit-identical,
// measured - is no faster: the chain is instructions, not round trips).
// Here a workgroup is ONE 16 x 16 output tile and its four waves SPLIT THE K RANGE: wave w takes chunks w, w + 4, ..., fetches its
// operands straight into the MFMA fragment layout (lane (li, g): row li, k = 8 g .. 8 g + 7 of the chunk - two quads per operand, all
// loads of PWK_G chunks in flight at once), cuts the planes in registers and multiplies: no LDS, no barrier in the loop.  The four
// partial accumulator pairs meet in LDS in wave order; wave 0 runs pw_kernel's epilogue.  The sums are grouped differently from
// pws_kernel's (by wave), so the form belongs to the PLAN (a batch of that variant equals its images run one by one through it), never to the
// tuner.  At four images and wide layers the tiles' re-fetches cost more than the shorter chains save (MobileNetV2 x1.4 @512).  Same operand planes, same float32 accumulation: the error against float64 is the split form's.
#define PWK_G 3     // chunks a wave has in flight

template <bool SIMPLE>
__global__ __launch_bounds__(256) void pwk_kernel(PwArgs a) {
    constexpr int MODE = SIMPLE ? 2 : 0;
    __shared__ f32x4 red[3][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const unsigned ntn = (a.N + 15) / 16;
    const unsigned L = yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (int)(L / ntn) * 16, n0 = (int)(L % ntn) * 16;
    const int kp = a.S.kp, nch = (kp + PWS_BK - 1) / PWS_BK;
    const bool gated = SIMPLE && a.gate != nullptr;
    PwRow<MODE> row;
    row.init(a, m0 + li);
    if constexpr (SIMPLE)
        if (!gated) row.grow = a.wt;   // ungated: the gate load becomes a (cached, ignored) weight quad
    const float* brow = a.wt + (size_t)(n0 + li < a.N ? n0 + li : 0) * kp;   // rows beyond N feed couts that are never stored
    // wave 0's BatchNorm terms: issued now, read after the loop
    float scq[4], shq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + g * 4 + r < a.N ? n0 + g * 4 + r : a.N - 1;
        scq[r] = a.scale ? a.scale[n] : 1.f;
        shq[r] = a.shift ? a.shift[n] : 0.f;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, ac1 = {0.f, 0.f, 0.f, 0.f};
    auto k_loop = [&](auto pools_tag) __attribute__((always_inline)) {
        constexpr bool POOLS = decltype(pools_tag)::value;
        for (int c0 = wave; c0 < nch; c0 += 4 * PWK_G) {
            float4 xa[PWK_G][2], xg[PWK_G][2], wb[PWK_G][2];
            int cv[PWK_G][2];
            pw_unroll<PWK_G>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                pw_unroll<2>([&](auto Q) __attribute__((always_inline)) {
                    constexpr int q = decltype(Q)::value;
                    const int kraw = (c0 + 4 * j) * PWS_BK + g * 8 + q * 4;     // (beyond kp for a dead chunk: clamped addresses, cv = 0)
                    row.template issue<POOLS>(a, kraw, kp, xa[j][q], xg[j][q], cv[j][q]);
                    wb[j][q] = *reinterpret_cast<const float4*>(brow + (kraw < kp ? kraw : kp - 4));
                });
            });
            pw_unroll<PWK_G>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                if (c0 + 4 * j < nch) {      // (wave-uniform)
                    const float4 x0 = gated ? pw_finish<2>(xa[j][0], xg[j][0], cv[j][0]) : pw_finish<1>(xa[j][0], xg[j][0], cv[j][0]);
                    const float4 x1 = gated ? pw_finish<2>(xa[j][1], xg[j][1], cv[j][1]) : pw_finish<1>(xa[j][1], xg[j][1], cv[j][1]);
                    unsigned h[8], m[8];
                    yr_cut2(x0.x, x0.y, h[0], m[0]);
                    yr_cut2(x0.z, x0.w, h[1], m[1]);
                    yr_cut2(x1.x, x1.y, h[2], m[2]);
                    yr_cut2(x1.z, x1.w, h[3], m[3]);
                    yr_cut2(wb[j][0].x, wb[j][0].y, h[4], m[4]);
                    yr_cut2(wb[j][0].z, wb[j][0].w, h[5], m[5]);
                    yr_cut2(wb[j][1].x, wb[j][1].y, h[6], m[6]);
                    yr_cut2(wb[j][1].z, wb[j][1].w, h[7], m[7]);
                    const pws_u4 xh = {h[0], h[1], h[2], h[3]}, xm = {m[0], m[1], m[2], m[3]}, wh = {h[4], h[5], h[6], h[7]}, wm = {m[4], m[5], m[6], m[7]};
                    acc = pws_mfma(wh, xh, acc);
                    ac1 = pws_mfma(wh, xm, ac1);
                    ac1 = pws_mfma(wm, xh, ac1);
                }
            });
        }
    };
    bool pooled = false;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < YR_MAX_SRC; ++i)
            pooled |= a.S.s[i].xform == YR_X_MAXPOOL2 || a.S.s[i].xform == YR_X_MAXPOOL4;
    }
    if (pooled) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    if (wave > 0) { red[wave - 1][0][lane] = acc; red[wave - 1][1][lane] = ac1; }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) { acc += red[w][0][lane]; ac1 += red[w][1][lane]; }    // (in wave order: the result does not depend on timing)
    const f32x4 sc = {scq[0], scq[1], scq[2], scq[3]}, sh = {shq[0], shq[1], shq[2], shq[3]};
    pw_finish_quad(a, __builtin_elementwise_fma(ac1, (f32x4){0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f}, acc), sc, sh, m0 + li, n0 + g * 4, li,
                   (a.out_ld & 3) == 0, (a.res_ld & 3) == 0, (a.pre_ld & 3) == 0);
}

int yr_pw_launch_ksplit(const PwArgs& a, hipStream_t s) {
    dim3 grid((unsigned)((a.M + 15) / 16) * (unsigned)((a.N + 15) / 16));
    const bool simple = a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY;
    yr_note_kernel(simple ? "pwk_kernel<1>" : "pwk_kernel<0>");
    if (simple) hipLaunchKernelGGL((pwk_kernel<true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pwk_kernel<false>), grid, dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

int yr_pw_launch_split(int shape, const PwArgs& a, hipStream_t s) {
    switch (shape) {
        case 0: return launch_split_cfg<4, 1, 4, 1>(a, s);
        case 1: return launch_split_cfg<2, 2, 4, 1>(a, s);
        case 2: return launch_split_cfg<2, 3, 4, 1>(a, s);
        case 3: return launch_split_cfg<4, 2, 2, 2>(a, s);
        case 4: return launch_split_cfg<2, 5, 4, 1>(a, s);
        case 5: return launch_split_cfg<4, 3, 2, 2>(a, s);
        case 6: return launch_split_cfg<4, 4, 2, 2>(a, s);
        case 7: return launch_split_cfg<1, 1, 4, 1>(a, s);
        case 8: return launch_split_cfg<1, 2, 4, 1>(a, s);
        case 9: return launch_split_cfg<1, 3, 4, 1>(a, s);
        case 10: return launch_split_cfg<1, 4, 4, 1>(a, s);
        case 11: return launch_split_cfg<1, 5, 4, 1>(a, s);
        case 12: return launch_split_cfg<1, 6, 4, 1>(a, s);
        case 13: return launch_split_cfg<1, 8, 4, 1>(a, s);
        default: yr_set_error("pointwise (split form): shape %d out of range", shape); return YR_ERR_ARG;
    }
}
