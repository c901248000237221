// Fused inverted-residual block, float32 (split form), WEIGHT-STREAMING formulation (YR_OP_MBR with k bit 6: the blocks whose
// two 1x1 convolutions do not fit one CU's register file - MobileNetV2 x0.75 block_11..15: 72 -> 432 -> 72 | 120, 120 -> 720 -> 120;
// x1.4: 88 -> 528 -> 88 | 136, 136 -> 816 -> 136; [3P] via reference code/yolo3/override.py:290-341):
//   expand 1x1 + BN + ReLU6 -> depthwise 3x3 (stride 1|2, TF SAME) + BN + ReLU6 -> project 1x1 + BN (+ residual)
// in ONE launch, the 6x-wide expanded tensor and the depthwise map never leave the CU.
//
// mbr.hip keeps the WEIGHTS stationary in registers and walks the pixels past them; that needs all A fragments of both
// convolutions in one CU's register file (1208 registers per lane for 72 -> 432 -> 72 in float16 planes: 309 KB of 512 KB).  Here the
// PIXELS are stationary and the weights stream past them, flash-attention style:
//   * a workgroup owns a strip of 16 input columns x NW * ROWS consecutive input rows; a WAVE owns ROWS (1 | 2) of those rows for the
//     whole kernel: their float16 planes (the expand conv's B operands, cut once) and the float32 accumulators of the projection
//     of ITS output rows stay in registers from the first to the last expanded channel;
//   * the expanded channels are processed in PAIRS OF TILES (32 channels = one K = 32 step of the projection).  Per pair, the
//     fragments of both convolutions, the depthwise taps and the BN shifts arrive as ONE contiguous chunk through LDS-direct
//     buffer loads (two chunk buffers: chunk q + 1 is in flight while chunk q is used); every wave reads each fragment from LDS
//     once and uses it for all its rows;
//   * expand (v_mfma_f32_16x16x32_f16, three per float32 product: mbr_common.h) -> BN shift / ReLU6 on the result registers; the
//     MFMA result layout - a lane holds 4 consecutive channels of one pixel, a DPP row = the strip's 16 pixels - is what the
//     depthwise taps want (row_shr / row_shl, mbr_dw_row) and what the projection's B operand wants;
//   * the vertical taps need the rows of the NEIGHBOUR waves: every wave parks its first and last expanded row of the pair in LDS,
//     one barrier, and reads the row above its first / below its last one (1 KB per tile and row);
//   * depthwise + BN + ReLU6 in registers, cut into planes, 3 MFMAs per cout tile into the resident accumulators.
// Two barriers per tile pair (chunk landed | rows parked), 14 pairs for 432 expanded channels.  HBM traffic = block input + output
// (+ the weight stream from L2: one pass per workgroup).  Per wave and pair: 66 MFMAs, ~46 ds_read_b128, ~240 VALU (ROWS = 2).
// Nothing here depends on a tuned parameter: the sums are grouped by the SHAPE (results are the same for every batch / tuner state).
//
// Geometry.  Stride 1: lane px of a DPP row = input column 14 strip - 1 + px, outputs at lanes 1..14; the workgroup of segment s owns
// input rows s (NR - 2) .. + NR - 1 (NR = NW ROWS) and emits the rows whose two neighbours it owns or that lie at the image border.
// Stride 2 (ROWS = 2): even input columns in lanes 0..7, odd ones in 8..15 (mbr_dw_row2), wave w = output row y0 + w = input rows
// 2 y - pad_t, + 1, and + 2 from the wave below; outputs in lanes 0..6.
#include "yr_common.h"
#include <type_traits>

#include "mbr_common.h"

struct MbkArgs {
    const float* x; float* out;
    const float* wa;   // NQ chunks of CHB bytes: see compiler.mbk_pack
    const float* bp;   // project BN shift [16 TO]
    int H, W, Ho, Wo, ld_in, ld_out, pad_t, pad_l, strips, segs;
    unsigned wa_bytes;
};

typedef __attribute__((address_space(3))) void* mbk_lds_ptr;
#ifdef MBK_TIMING     // (python tools/relink.py mbk.hip -DMBK_TIMING; tools/mbk_timing.py: per-phase shader-clock totals of every wave, written behind the last image of the output - a probe build)
#define MBK_T(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define MBK_T(i)
#endif

// (EXP: ablations of tools/mbk_probe.py, -DMBK_EXPERIMENT builds only - wrong results: 1 no depthwise taps, 2 no expand MFMAs, 3 no project
//  MFMAs, 4 no barrier in the loop, 5 no chunk loads in the loop, 6 no fragment reads from LDS in the loop)
template <int CIN, int CEXP, int COUT, int S, int ROWS, int NW, bool RES, int EXP = 0>
__global__ __launch_bounds__(64 * NW) void mbk_kernel(MbkArgs a) {
    constexpr int T = CEXP / 16, NQ = (T + 1) / 2, TO = (COUT + 15) / 16, NKE = (CIN + 31) / 32, NOUT = 14 / S;
    constexpr int EXB = 2 * NKE * 2 * 1024, PRB = TO * 2 * 1024, CHB = EXB + PRB + 2048;     // bytes of one chunk: expand | project | 2 x [11][16] floats (padded to 2 KB)
    constexpr int NPIECE = CHB / 1024, NR = NW * ROWS;
    constexpr bool ILV = EXP != 7;        // (EXP 7: the loop body as the compiler orders it - the A/B of tools/mbk_probe.py)
    constexpr bool SKEW = false;      // (measured: opposite stream orders on the two waves of a SIMD - 45 -> 50 us on block_11; see DESIGN)
    static_assert(S == 1 || ROWS == 2, "stride 2: a wave owns the two input rows of its output row");
    static_assert(CEXP % 16 == 0 && COUT % 4 == 0 && CIN % 8 == 0, "widths");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];     // [3][CHB] chunk buffers | [2][NW + 2][NPR][2][64] v4f parked rows
    constexpr int NPR = S == 1 && ROWS > 1 ? 2 : 1;      // rows a wave parks per tile: its first and (if it is another one) its last
    constexpr int XCB = (NW + 2) * NPR * 2 * 1024;       // (slots 0 and NW + 1 stay zero: the rows beyond the workgroup's - every neighbour read is unconditional)
    char* const xch = lds_raw + 3 * CHB;
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int seg = bid % a.segs; bid /= a.segs;
    const int strip = bid % a.strips;
    const int b = bid / a.strips;
#ifdef MBK_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif

    // ---- rows of this wave, columns of this lane
    int rin[ROWS];            // input rows (may lie outside the image: zero rows)
    int yo;                   // S == 2: the output row;  S == 1: output row of input row i is rin[i]
    bool emit[ROWS];          // S == 1: row i is emitted;  S == 2: emit[0] only
    if constexpr (S == 1) {
        const int ri0 = seg * (NR - 2), out0 = seg == 0 ? 0 : ri0 + 1;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int r = ri0 + ROWS * w + i;
            rin[i] = r;
            emit[i] = r >= out0 && r < a.H && (r + 1 < ri0 + NR || r + 1 >= a.H);
        }
        yo = 0;
    } else {
        yo = seg * (NW - 1) + w;
        rin[0] = 2 * yo - a.pad_t; rin[1] = rin[0] + 1;
        emit[0] = yo < a.Ho && (w < NW - 1 || rin[0] + 2 >= a.H);
        emit[1] = false;
    }
    const int podd = S == 2 ? px >> 3 : 0;
    const int xin = S * NOUT * strip - a.pad_l + (S == 2 ? 2 * (px & 7) + podd : px);
    const int xc = min(max(xin, 0), a.W - 1);
    const bool col_in = xin >= 0 && xin < a.W;
    const int xo = NOUT * strip + (S == 2 ? (px & 7) : px - 1);
    const bool out_lane = (S == 2 ? px < 7 : (px >= 1 && px <= 14)) && xo < a.Wo;

    const mbr_rsrc xsrc = mbr_make_rsrc(a.x + (size_t)b * a.H * a.W * a.ld_in, (unsigned)(a.H * a.W * a.ld_in) * 4u);
    const mbr_rsrc osrc = mbr_make_rsrc(a.out + (size_t)b * a.Ho * a.Wo * a.ld_out, (unsigned)(a.Ho * a.Wo * a.ld_out) * 4u);
    const mbr_rsrc wsrc = mbr_make_rsrc(a.wa, a.wa_bytes);

    // ---- the weight stream: chunk q -> buffer q & 1, 1 KB pieces shared out over the waves
    auto issue_chunk = [&](const int q) {
        char* dst = lds_raw + (q % 3) * CHB;
        const int qs = min(q, NQ - 1);
#pragma unroll
        for (int u = 0; u < (NPIECE + NW - 1) / NW; ++u) {
            const int p = u * NW + w;
            if (NPIECE % NW == 0 || p < NPIECE)      // (wave-uniform)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (mbk_lds_ptr)(dst + p * 1024), 16, (unsigned)(qs * CHB + p * 1024 + lane * 16), 0, 0, 0);
        }
    };
    constexpr int NU = (NPIECE + NW - 1) / NW;      // pieces of a chunk per wave
    auto issue_piece = [&](const int q, const int u) {
        const int p = u * NW + w;
        if (NPIECE % NW == 0 || p < NPIECE)          // (wave-uniform)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (mbk_lds_ptr)(lds_raw + (q % 3) * CHB + p * 1024), 16, (unsigned)(min(q, NQ - 1) * CHB + p * 1024 + lane * 16), 0, 0, 0);
    };
    issue_chunk(0);
    issue_chunk(NQ > 1 ? 1 : 0);
    if (w < 2) {     // the zero rows above the first and below the last wave, both buffers
#pragma unroll
        for (int u = 0; u < 2 * NPR * 2; ++u)
            reinterpret_cast<v4f*>(xch + (u / (NPR * 2)) * XCB + (w * (NW + 1) * NPR * 2 + u % (NPR * 2)) * 1024)[lane] = (v4f){0.f, 0.f, 0.f, 0.f};
    }

    // ---- the wave's pixels: float16 planes of its rows, cut once (step c = the lane's channels 32 c + 8 mg .. + 7; zeros beyond the block input)
    mbs_u4 xh[ROWS][NKE], xm[ROWS][NKE];
    float hr[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const bool row_in = rin[i] >= 0 && rin[i] < a.H;
        hr[i] = row_in && col_in ? 6.f : 0.f;
        const unsigned so = (unsigned)min(max(rin[i], 0), a.H - 1) * (unsigned)(a.W * a.ld_in) * 4u;
        v4f lo[NKE], hi[NKE];
#pragma unroll
        for (int c = 0; c < NKE; ++c) {
            const unsigned off = 32 * c + 8 * mg < CIN ? ((unsigned)xc * (unsigned)a.ld_in + 32u * c + 8u * mg) * 4u : MBR_DEAD;
            lo[c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, so, 0));
            hi[c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off == MBR_DEAD ? MBR_DEAD : off + 16u, so, 0));
        }
#pragma unroll
        for (int c = 0; c < NKE; ++c) {
            const float v[8] = {lo[c][0], lo[c][1], lo[c][2], lo[c][3], hi[c][0], hi[c][1], hi[c][2], hi[c][3]};
            mbs_split8(v, xh[i][c], xm[i][c]);
        }
    }

    constexpr int NE = S == 1 ? ROWS : 1;      // output rows of a wave
    v4f P[NE][TO], P1[NE][TO];
#pragma unroll
    for (int i = 0; i < NE; ++i)
#pragma unroll
        for (int t = 0; t < TO; ++t) { P[i][t] = (v4f){0.f, 0.f, 0.f, 0.f}; P1[i][t] = P[i][t]; }
    const v4f k11 = (v4f){0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f};

    // ---- expand of tile pair q (chunk buffer cb): 2 tiles x ROWS rows, BN shift as the accumulators' first value, ReLU6 (upper bound 0
    // outside the image)
    auto expand = [&](const char* cb, v4f (&ec)[ROWS][2]) {
        const u4* fe = reinterpret_cast<const u4*>(cb) + lane;                    // [2 tiles][NKE][2 planes][64]
        const v4f* tb0 = reinterpret_cast<const v4f*>(cb + EXB + PRB) + mg;       // tile j: tb0 + 44 j;  [11][16] floats = 44 v4f
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const v4f se = tb0[44 * j + 40];
            v4f e1[ROWS];
#pragma unroll
            for (int i = 0; i < ROWS; ++i) { ec[i][j] = se; e1[i] = (v4f){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int c = 0; c < NKE; ++c) {
                if (EXP == 2) continue;
                const u4 wh = EXP == 6 ? xh[0][c] : fe[((j * NKE + c) * 2 + 0) * 64], wm = EXP == 6 ? xm[0][c] : fe[((j * NKE + c) * 2 + 1) * 64];
#pragma unroll
                for (int i = 0; i < ROWS; ++i) ec[i][j] = mbs_mfma(wh, xh[i][c], ec[i][j]);
#pragma unroll
                for (int i = 0; i < ROWS; ++i) e1[i] = mbs_mfma(wh, xm[i][c], e1[i]);
#pragma unroll
                for (int i = 0; i < ROWS; ++i) e1[i] = mbs_mfma(wm, xh[i][c], e1[i]);
            }
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                ec[i][j] = __builtin_elementwise_fma(e1[i], k11, ec[i][j]);
#pragma unroll
                for (int s = 0; s < 4; ++s) ec[i][j][s] = __builtin_amdgcn_fmed3f(ec[i][j][s], 0.f, hr[i]);
            }
        }
    };
    // ---- the rows the neighbour waves need: parked in LDS (buffer q & 1), fetched behind the barrier
    auto park_rows = [&](const int q, const v4f (&ec)[ROWS][2]) {
        v4f* park = reinterpret_cast<v4f*>(xch + (q & 1) * XCB) + lane;        // [NW][2: first | last row][2 tiles][64]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            park[(((w + 1) * NPR + 0) * 2 + j) * 64] = ec[0][j];
            if (NPR > 1) park[(((w + 1) * NPR + 1) * 2 + j) * 64] = ec[ROWS - 1][j];
        }
    };
    // ---- depthwise 3x3 + BN + ReLU6 of pair q (tap rows applied to both output rows of the wave while they are in registers), then
    // the projection: the lane's 8 depthwise results of the pair are its 8 k values of this step
    auto dw_project = [&](const int q, const char* cb, const v4f (&ec)[ROWS][2]) {
        const u4* fp = reinterpret_cast<const u4*>(cb + EXB) + lane;              // [TO][2 planes][64]
        const v4f* tb0 = reinterpret_cast<const v4f*>(cb + EXB + PRB) + mg;
        const v4f* park = reinterpret_cast<const v4f*>(xch + (q & 1) * XCB) + lane;
        v4f above[2], below[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            above[j] = S == 1 ? park[((w * NPR + NPR - 1) * 2 + j) * 64] : (v4f){0.f, 0.f, 0.f, 0.f};     // slot w = wave w - 1
            below[j] = park[(((w + 2) * NPR + 0) * 2 + j) * 64];
        }
        v4f d[NE][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const v4f* tb = tb0 + 44 * j;
            if constexpr (S == 1) {
#pragma unroll
                for (int i = 0; i < ROWS; ++i) d[i][j] = tb[36];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const v4f w0 = tb[12 * ky], w1 = tb[12 * ky + 4], w2 = tb[12 * ky + 8];
#pragma unroll
                    for (int i = 0; i < ROWS; ++i) {
                        const int rr = i + ky - 1;      // the wave's row index the tap row reads (-1: above, ROWS: below)
                        if (EXP == 1) { d[i][j] += (rr < 0 ? above[j] : rr >= ROWS ? below[j] : ec[rr < 0 ? 0 : rr >= ROWS ? 0 : rr][j]) * w1; continue; }
                        mbr_dw_row(d[i][j], rr < 0 ? above[j] : rr >= ROWS ? below[j] : ec[rr < 0 ? 0 : rr >= ROWS ? 0 : rr][j], w0, w1, w2);
                    }
                }
            } else {
                d[0][j] = tb[36];
                mbr_dw_row2<false>(d[0][j], ec[0][j], tb[0], tb[4], tb[8]);
                mbr_dw_row2<false>(d[0][j], ec[1][j], tb[12], tb[16], tb[20]);
                mbr_dw_row2<false>(d[0][j], below[j], tb[24], tb[28], tb[32]);
            }
#pragma unroll
            for (int i = 0; i < NE; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) d[i][j][s] = __builtin_amdgcn_fmed3f(d[i][j][s], 0.f, 6.f);
        }
        mbs_u4 bh[NE], bm[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const float v[8] = {d[i][0][0], d[i][0][1], d[i][0][2], d[i][0][3], d[i][1][0], d[i][1][1], d[i][1][2], d[i][1][3]};
            mbs_split8(v, bh[i], bm[i]);
        }
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            if (EXP == 3) { P[0][t] += __builtin_bit_cast(v4f, bh[0]) + __builtin_bit_cast(v4f, bm[NE - 1]); continue; }
            const u4 wh = EXP == 6 ? bm[0] : fp[(t * 2 + 0) * 64], wm = EXP == 6 ? bh[0] : fp[(t * 2 + 1) * 64];
#pragma unroll
            for (int i = 0; i < NE; ++i) P[i][t] = mbs_mfma(wh, bh[i], P[i][t]);
#pragma unroll
            for (int i = 0; i < NE; ++i) P1[i][t] = mbs_mfma(wh, bm[i], P1[i][t]);
#pragma unroll
            for (int i = 0; i < NE; ++i) P1[i][t] = mbs_mfma(wm, bh[i], P1[i][t]);
        }
    };

    // ---- the loop body proper: the depthwise stage + projection of pair q and the expand conv of pair q + 1 as ONE hand-interleaved
    // instruction stream.  Left to the compiler the body is [all VALU][all MFMAs] and the phases add up (ablations): the
    // matrix pipe idles under the taps, the VALU under the MFMAs, and every fragment read is waited for in front of its MFMA.  Here a
    // SLICE = one MFMA + a third of a tap row (4 VALU), fenced by sched_barrier(0) so the order survives; the LDS reads of a group of
    // slices (fragments of one k step, taps of one tap row) are issued a group ahead.
    auto fused_step = [&](const int q, const char* cbq, const char* cbn, const v4f (&ec)[ROWS][2], v4f (&en)[ROWS][2]) {
        constexpr int MPG = 3 * ROWS, NMG = 2 * NKE, NM = NMG * MPG;          // expand: MFMAs per (tile, k step) group, groups, MFMAs
        constexpr int PPG = 3 * NE, NDG = 6, NV = NDG * PPG;                  // depthwise: parts per (tile, tap row) group, groups, parts
        constexpr int NS = NM > NV ? NM : NV;
        const u4* fe = reinterpret_cast<const u4*>(cbn) + lane;
        const v4f* tbn = reinterpret_cast<const v4f*>(cbn + EXB + PRB) + mg;
        const u4* fp = reinterpret_cast<const u4*>(cbq + EXB) + lane;
        const v4f* tbq = reinterpret_cast<const v4f*>(cbq + EXB + PRB) + mg;
        const v4f* park = reinterpret_cast<const v4f*>(xch + (q & 1) * XCB) + lane;
        v4f above[2], below[2], d[NE][2], e1[ROWS];
        u4 fr[2][2];          // fragment ring: [group & 1][plane]
        v4f tp[2][3];         // tap ring: [group & 1][kx]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            above[j] = S == 1 ? park[((w * NPR + NPR - 1) * 2 + j) * 64] : (v4f){0.f, 0.f, 0.f, 0.f};     // slot w = wave w - 1
            below[j] = park[(((w + 2) * NPR + 0) * 2 + j) * 64];
        }
        fr[0][0] = fe[0]; fr[0][1] = fe[64];
        tp[0][0] = tbq[0]; tp[0][1] = tbq[4]; tp[0][2] = tbq[8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const v4f sh = tbq[44 * j + 36], se = tbn[44 * j + 40];
#pragma unroll
            for (int i = 0; i < NE; ++i) d[i][j] = sh;
#pragma unroll
            for (int i = 0; i < ROWS; ++i) en[i][j] = se;
        }
        __builtin_amdgcn_sched_barrier(0);
        MBK_T(2)
        mbk_for<NS>([&](auto SS) {
            constexpr int sl = decltype(SS)::value;
            // chunk q + 2 goes out piece by piece in the shadow of the MFMAs (an LDS-direct load costs 100+ cycles of issue at the top
            // of the turn, where nothing hides it: tools/mbk_timing.py) - into the buffer chunk q - 1 was read from, free since the barrier
            if constexpr (EXP != 5) {
                mbk_for<NU>([&](auto UU) {
                    constexpr int u = decltype(UU)::value;
                    if constexpr ((2 * u + 1) * NS / (2 * NU) == sl) issue_piece(q + 2, u);
                });
            }
            if constexpr (sl < NM) {
                constexpr int g = sl / MPG, r = sl % MPG, kind = r / ROWS, i = r % ROWS, j = g / NKE, c = g % NKE;
                if constexpr (r == 0 && g + 1 < NMG) {
                    fr[(g + 1) & 1][0] = fe[((g + 1) * 2 + 0) * 64];
                    fr[(g + 1) & 1][1] = fe[((g + 1) * 2 + 1) * 64];
                }
                if constexpr (c == 0 && kind == 1) e1[i] = (v4f){0.f, 0.f, 0.f, 0.f};
                if constexpr (EXP == 2) {}
                else if constexpr (kind == 0) en[i][j] = mbs_mfma(fr[g & 1][0], xh[i][c], en[i][j]);
                else if constexpr (kind == 1) e1[i] = mbs_mfma(fr[g & 1][0], xm[i][c], e1[i]);
                else e1[i] = mbs_mfma(fr[g & 1][1], xh[i][c], e1[i]);
                if constexpr (c == NKE - 1 && kind == 2) en[i][j] = __builtin_elementwise_fma(e1[i], k11, en[i][j]);   // (the clamp waits for phase 3)
            }
            if constexpr (sl < NV) {
                constexpr int dg = sl / PPG, pr = sl % PPG, i = pr / 3, part = pr % 3, j = dg / 3, ky = dg % 3;
                if constexpr (pr == 0 && dg + 1 < NDG) {
                    constexpr int j1 = (dg + 1) / 3, ky1 = (dg + 1) % 3;
                    tp[(dg + 1) & 1][0] = tbq[44 * j1 + 12 * ky1];
                    tp[(dg + 1) & 1][1] = tbq[44 * j1 + 12 * ky1 + 4];
                    tp[(dg + 1) & 1][2] = tbq[44 * j1 + 12 * ky1 + 8];
                }
                if constexpr (EXP == 1) { if (part == 0) d[i][j] += ec[i][j] * tp[dg & 1][1] + above[j] + below[j]; }
                else if constexpr (S == 1) {
                    constexpr int rr = i + ky - 1;      // the wave's row the tap row reads (-1: above, ROWS: below)
                    // (part 0 = the centre tap kx = 1, part 1 = the left neighbour kx = 0, part 2 = the right one kx = 2)
                    mbk_dw_part(part, d[i][j], rr < 0 ? above[j] : rr >= ROWS ? below[j] : ec[rr < 0 ? 0 : rr >= ROWS ? 0 : rr][j], tp[dg & 1][part == 0 ? 1 : part == 1 ? 0 : 2]);
                } else {
                    mbk_dw2_part(part, d[0][j], ky == 0 ? ec[0][j] : ky == 1 ? ec[1][j] : below[j], tp[dg & 1][part]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        MBK_T(3)
        // phase 2 (VALU): ReLU6 of the depthwise results, the float16 planes of the pair's 8 values per lane
        u4 pf[2][2];
        pf[0][0] = fp[0]; pf[0][1] = fp[64];
        mbs_u4 bh[NE], bm[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) d[i][j][s] = __builtin_amdgcn_fmed3f(d[i][j][s], 0.f, 6.f);
            const float v[8] = {d[i][0][0], d[i][0][1], d[i][0][2], d[i][0][3], d[i][1][0], d[i][1][1], d[i][1][2], d[i][1][3]};
            mbs_split8(v, bh[i], bm[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        MBK_T(4)
        // phase 3: the projection's MFMAs with the ReLU6 of the new expanded rows (4 v_med3 per slice) in their shadow
        constexpr int MPT = 3 * NE, NM3 = TO * MPT, NV3 = ROWS * 2;
        mbk_for<(NM3 > NV3 ? NM3 : NV3)>([&](auto SS) {
            constexpr int sl = decltype(SS)::value;
            if constexpr (sl < NM3) {
                constexpr int t = sl / MPT, r = sl % MPT, kind = r / NE, i = r % NE;
                if constexpr (r == 0 && t + 1 < TO) {
                    pf[(t + 1) & 1][0] = fp[((t + 1) * 2 + 0) * 64];
                    pf[(t + 1) & 1][1] = fp[((t + 1) * 2 + 1) * 64];
                }
                if constexpr (EXP == 3) { if (kind == 0) P[i][t] += __builtin_bit_cast(v4f, bh[i] + pf[t & 1][0]) + __builtin_bit_cast(v4f, bm[i] + pf[t & 1][1]); }
                else if constexpr (kind == 0) P[i][t] = mbs_mfma(pf[t & 1][0], bh[i], P[i][t]);
                else if constexpr (kind == 1) P1[i][t] = mbs_mfma(pf[t & 1][0], bm[i], P1[i][t]);
                else P1[i][t] = mbs_mfma(pf[t & 1][1], bh[i], P1[i][t]);
            }
            if constexpr (sl < NV3) {
                constexpr int i = sl / 2, j = sl % 2;
#pragma unroll
                for (int s = 0; s < 4; ++s) en[i][j][s] = __builtin_amdgcn_fmed3f(en[i][j][s], 0.f, hr[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        MBK_T(5)
    };

    // ONE barrier per tile pair: behind barrier q every wave has parked its rows of pair q and chunk q + 1 has landed, so the depthwise
    // stage + projection of pair q and the expand conv of pair q + 1 - independent instruction streams (VALU | MFMA) - run in the same
    // stretch of code; chunk q + 2 goes into the buffer chunk q - 1 was read from (three chunk buffers, two sets of parked rows).
    v4f ec[ROWS][2];
    __builtin_amdgcn_s_waitcnt(0x0f70);         // vmcnt(0): chunks 0 and 1 (this wave's pieces) and the pixel rows
    __syncthreads();
    expand(lds_raw, ec);
    park_rows(0, ec);
    MBK_T(0)
    for (int q = 0; q + 1 < NQ; ++q) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        if (EXP != 4) __syncthreads();
        MBK_T(1)
        if (EXP != 5 && !ILV) issue_chunk(q + 2);              // (the interleaved body issues it inside its slices; beyond the last chunk: the last one again, into a buffer nobody reads any more)
        v4f en[ROWS][2];
        // The two streams in OPPOSITE order on the two waves that share a SIMD (waves w and w + 4: a workgroup's waves go to the SIMDs
        // cyclically): between two barriers all waves run the same code from the same starting line, so with one order both waves of a
        // SIMD would want the VALU at the same time and then the matrix pipe at the same time.
        if (ILV) {
            fused_step(q, lds_raw + (q % 3) * CHB, lds_raw + ((q + 1) % 3) * CHB, ec, en);
        } else if (SKEW && (w & 4)) {
            expand(lds_raw + ((q + 1) % 3) * CHB, en);
            __builtin_amdgcn_sched_barrier(0);
            dw_project(q, lds_raw + (q % 3) * CHB, ec);
        } else {
            dw_project(q, lds_raw + (q % 3) * CHB, ec);
            __builtin_amdgcn_sched_barrier(0);
            expand(lds_raw + ((q + 1) % 3) * CHB, en);
        }
        park_rows(q + 1, en);
#pragma unroll
        for (int i = 0; i < ROWS; ++i) { ec[i][0] = en[i][0]; ec[i][1] = en[i][1]; }
        MBK_T(6)
    }
    // the residual and the projection's BN shift are fetched BEFORE the last pair's depthwise stage and projection: their round trip
    // (a load nobody waits for until the stores) runs under ~1500 cycles of work instead of in front of the stores
    v4f res[NE][RES ? TO : 1], psh[TO];
    unsigned oaddr[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int y = S == 1 ? rin[i] : yo;
        const bool live = emit[i] && out_lane;
        const unsigned opix = ((unsigned)max(y, 0) * (unsigned)a.Wo + (unsigned)xo);
        oaddr[i] = live ? opix * (unsigned)a.ld_out * 4u : MBR_DEAD;
        if constexpr (RES) {
#pragma unroll
            for (int t = 0; t < TO; ++t)
                res[i][t] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, live && 16 * t + 4 * mg < COUT ? (opix * (unsigned)a.ld_in + 16u * t + 4u * mg) * 4u : MBR_DEAD, 0, 0));
        }
    }
#pragma unroll
    for (int t = 0; t < TO; ++t) psh[t] = *reinterpret_cast<const v4f*>(a.bp + 16 * t + 4 * mg);
    __builtin_amdgcn_sched_barrier(0);     // (the loads are issued HERE)
    __syncthreads();
    dw_project(NQ - 1, lds_raw + ((NQ - 1) % 3) * CHB, ec);

    // ---- BN shift, residual, store (16 bytes per lane and cout tile)
#pragma unroll
    for (int i = 0; i < NE; ++i) {
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            const int co = 16 * t + 4 * mg;
            v4f v = __builtin_elementwise_fma(P1[i][t], k11, P[i][t]) + psh[t];
            if constexpr (RES) v += res[i][t];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), osrc, oaddr[i] != MBR_DEAD && co < COUT ? oaddr[i] + (unsigned)co * 4u : MBR_DEAD, 0, 0);
        }
    }
#ifdef MBK_TIMING
    MBK_T(7)
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (lane < 8) reinterpret_cast<unsigned*>(a.out)[(size_t)(gridDim.x / (a.strips * a.segs)) * a.Ho * a.Wo * a.ld_out + ((size_t)blockIdx.x * NW + w) * 8 + lane] = (unsigned)tacc[lane];
#endif
}

// segments a map of H input rows is cut into (the same formula as the kernel's row rule; compiler.mbk_segs)
static int mbk_segs(int S, int H, int Ho, int pad_t, int NW, int ROWS) {
    if (S == 1) {
        const int NR = NW * ROWS;
        return H <= NR ? 1 : (H - NR + NR - 3) / (NR - 2) + 1;
    }
    int s = 0;
    for (;; ++s) {   // the last wave of a segment emits only when the row below it lies outside the image
        const int ylast = s * (NW - 1) + NW - 1;
        const int end = (2 * ylast - pad_t + 2 >= H) ? ylast : ylast - 1;
        if (end >= Ho - 1) break;
    }
    return s + 1;
}

template <int CIN, int CEXP, int COUT, int S, int ROWS, int NW, bool RES, int EXP = 0>
static int launch_mbk(const MbkArgs& a0, int batch, hipStream_t s) {
    MbkArgs a = a0;
    constexpr int T = CEXP / 16, NQ = (T + 1) / 2, TO = (COUT + 15) / 16, NKE = (CIN + 31) / 32, NOUT = 14 / S;
    constexpr int CHB = 2 * NKE * 2 * 1024 + TO * 2 * 1024 + 2048;
    a.strips = (a.Wo + NOUT - 1) / NOUT;
    a.segs = mbk_segs(S, a.H, a.Ho, a.pad_t, NW, ROWS);
    a.wa_bytes = (unsigned)NQ * CHB;
    const size_t lds = (size_t)3 * CHB + (size_t)2 * (NW + 2) * (S == 1 && ROWS > 1 ? 2 : 1) * 2 * 1024;
    static char nm[64];
    static const int nm_len = snprintf(nm, sizeof(nm), "mbk_kernel<%d,%d,%d,%d,%d,%d,%d>", CIN, CEXP, COUT, S, ROWS, NW, (int)RES);
    (void)nm_len;
    yr_note_kernel(nm);
    auto kern = mbk_kernel<CIN, CEXP, COUT, S, ROWS, NW, RES, EXP>;
    static bool attr_set_dev[64] = {};   // per device
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool& attr_set = attr_set_dev[cur_dev & 63];
    if (!attr_set) {
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(batch * a.strips * a.segs)), dim3(64 * NW), lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// YR_OP_MBR with k bit 6 (and bit 7: the operands are float16 planes): the weight-streaming form.  Op fields as YR_OP_MBR; k bits 8-15 =
// waves per workgroup, bits 16-23 = rows per wave (both fixed by the plan: compiler.MBK_SHAPES).  wgt = NQ = ceil(Cexp / 32) chunks, one per
// pair of expanded tiles (2 q, 2 q + 1), each [2 tiles][NKE][2 planes][64 lanes][8 halves] expand fragments (mbs_pack's) |
// [TO][2 planes][64][8] project fragments of the pair | [2 tiles][11][16] float32 (taps x BN scale | depthwise BN shift | expand BN shift)
// | zeros up to a multiple of 1 KB;  b2 = project BN shift [16 TO];  wgt2 unused.
int yr_launch_mbk(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.dtype == YR_F32 && op.out_dtype == YR_F32, "mbk: float32 plans only");
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY && op.src[0].dtype == YR_F32, "mbk: needs one float32 identity source");
    const yr_src& in = op.src[0];
    YR_REQUIRE((op.k & 0x3f) == 3 && (op.k & 0xc0) == 0xc0 && (op.stride == 1 || op.stride == 2) && op.act == YR_ACT_RELU6, "mbk: 3x3, stride 1|2, ReLU6, split + streamed form");
    YR_REQUIRE(in.ptr && op.out && op.wgt && op.b2, "mbk: null pointer");
    YR_REQUIRE(in.ld % 4 == 0 && op.out_ld % 4 == 0 && in.c == op.cin && in.ld >= in.c && op.out_ld >= op.cout, "mbk: channel strides");
    YR_REQUIRE(((uintptr_t)in.ptr) % 16 == 0 && ((uintptr_t)op.out) % 16 == 0 && ((uintptr_t)op.wgt) % 16 == 0, "mbk: pointers must be 16-byte aligned");
    MbkArgs a;
    a.x = (const float*)in.ptr; a.out = (float*)op.out; a.wa = op.wgt; a.bp = op.b2;
    a.H = in.h; a.W = in.w; a.Ho = (in.h + op.stride - 1) / op.stride; a.Wo = (in.w + op.stride - 1) / op.stride;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "mbk: output dims mismatch");
    a.ld_in = in.ld; a.ld_out = op.out_ld;
    const int pth = (a.Ho - 1) * op.stride + 3 - in.h, ptw = (a.Wo - 1) * op.stride + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    const bool res = op.res != nullptr;
    if (res) YR_REQUIRE(op.res == in.ptr && op.stride == 1 && in.c == op.cout, "mbk: the residual must be the block input (stride 1, Cin == Cout)");
    a.strips = a.segs = 0; a.wa_bytes = 0;
    const int nw = (op.k >> 8) & 0xff, rows = (op.k >> 16) & 0xff;
#ifdef MBK_EXPERIMENT
    if (const char* e = getenv("YR_MBK_EXP")) {
        switch (atoi(e)) {
            case 1: return launch_mbk<72, 432, 72, 1, 2, 8, true, 1>(a, batch, s);
            case 2: return launch_mbk<72, 432, 72, 1, 2, 8, true, 2>(a, batch, s);
            case 3: return launch_mbk<72, 432, 72, 1, 2, 8, true, 3>(a, batch, s);
            case 4: return launch_mbk<72, 432, 72, 1, 2, 8, true, 4>(a, batch, s);
            case 5: return launch_mbk<72, 432, 72, 1, 2, 8, true, 5>(a, batch, s);
            case 6: return launch_mbk<72, 432, 72, 1, 2, 8, true, 6>(a, batch, s);
            case 7: return launch_mbk<72, 432, 72, 1, 2, 8, true, 7>(a, batch, s);
        }
    }
#endif
#define MBK_CASE(CIN, CEXP, COUT, S, ROWS, NW, RES)                                                                            \
    if (in.c == CIN && op.se_reduced == CEXP && op.cout == COUT && op.stride == S && res == RES && nw == NW && rows == ROWS)      \
        return launch_mbk<CIN, CEXP, COUT, S, ROWS, NW, RES>(a, batch, s);
    MBK_CASE(48, 288, 48, 1, 2, 8, true)       // MobileNetV2 x0.75 block_7..9 (26 x 26)
    MBK_CASE(48, 288, 72, 1, 2, 8, false)      // block_10
    MBK_CASE(72, 432, 72, 1, 2, 8, true)       // block_11, 12 (26 x 26)
    MBK_CASE(72, 432, 120, 2, 2, 8, false)     // block_13 (26 x 26 -> 13 x 13)
    MBK_CASE(120, 720, 120, 1, 1, 8, true)     // block_14, 15 (13 x 13)
    // (measured and dropped, block_11 at 64 images: 4 waves x 2 rows - two workgroups per CU - 47 us against 46; 16 waves x 1 row 47;
    //  8 waves x 1 row 51 (two generations of workgroups); block_14: 4 waves 78 us, 2 rows per wave 58 against 49 - profiles/r06_mbk_probe.txt)
#undef MBK_CASE
    yr_set_error("mbk: block %d -> %d -> %d stride %d res %d (nw %d, rows %d) is not built", in.c, op.se_reduced, op.cout, op.stride, (int)res, nw, rows);
    return YR_ERR_ARG;
}
