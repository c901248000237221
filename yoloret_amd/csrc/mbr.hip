// Fused inverted-residual block, float32, ROW-WALKING REGISTER-CHAINED formulation (YR_OP_MBR):
//   expand 1x1 + BN + ReLU6 -> depthwise 3x3 (stride 1|2, TF SAME) + BN + ReLU6 -> project 1x1 + BN (+ residual)
// (MobileNetV2 block_1..15 [3P], reference code/yolo3/override.py:290-341; SE-free MBConv, efficientnet.py:467-536).
//
// Both 1x1 convolutions run on v_mfma_f32_16x16x4_f32 (exact float32 FMA chains) and the expanded tensor never leaves the
// register file - not even for LDS:
//   * a wave owns a STRIP of 16 adjacent input columns and walks down the rows.  The expand GEMM of one row is
//     D[expanded channel][pixel] = We * X with the 16 pixels of the strip row as the MFMA's N dimension; X comes straight
//     from global memory in operand layout (lane (pixel p, k group g) loads channels 16c + 4g .. +3 as one 16-byte load
//     and uses component s in MFMA step 4c + s; the weights are permuted the same way by the host).  The BN shift is the
//     accumulator's initial value (the scale is folded into the weights), ReLU6 one v_med3 whose upper bound is 0 for
//     lanes / rows outside the image (= TF's zero padding of the depthwise input).
//   * the MFMA result layout IS the layout the depthwise conv wants and the layout the projection's B operand wants:
//     a lane holds 4 consecutive channels of ONE pixel, the 16 lanes of a DPP row are the 16 pixels.  The three
//     horizontal taps are the lane itself and its row_shr:1 / row_shl:1 neighbours (DPP, no LDS), the three vertical
//     taps are the last three rows of the walk, kept in registers (a ring of two rows + the new one).  The depthwise
//     result (BN shift as the first addend, v_med3) is used AS IS as the B operand of the projection MFMAs.
//   * the A operands (expand and project weights of the wave's expanded-channel tiles) are STATIONARY in registers for the
//     whole walk; only the nine depthwise taps per tile come from a small LDS table.
//   * the NW waves of a workgroup share one strip and split the expanded channels (tiles of 16); their partial projections
//     of an output row meet in LDS (one barrier per output row, two buffers), the wave that finishes a cout tile adds the
//     BN shift and the residual and stores 16 bytes per lane.
// HBM traffic = block input + output; LDS traffic = 10 x 16 bytes per lane per tile-row + the partial sums.
// Strip geometry: lane l of a DPP row is input column base + l; outputs sit at lanes S*j + 1, j < 14 / S (the lanes whose
// three taps are inside the strip): 14 (stride 1) or 7 (stride 2) output columns per strip.
#include "yr_common.h"
#include <type_traits>

#include "mbr_common.h"

// (experiment, tools/relink.py mbr.hip -DMBR_EXP_ONE_TAP: every tap of the stride-1 block kernel reads the SAME table entry - wrong
//  results, one LDS read per tile and row instead of ten: what the tap reads cost)
#ifdef MBR_EXP_ONE_TAP
#define MBR_TB(i) tb[0]
#else
#define MBR_TB(i) tb[i]
#endif

struct MbrArgs {
    const float* x; float* out;
    const float* wa;   // A fragments: [T][KE + 4 * TO][64 lanes]
    const float* wt;   // per expanded tile [T][11][16]: nine depthwise taps (times the BN scale) | depthwise BN shift | expand BN shift
    const float* bp;   // project BN shift [16 * TO] (the scale is folded into the project weights)
    int H, W, Ho, Wo, ld_in, ld_out, pad_t, pad_l, strips, segs, seg_rows;
};

template <int CIN, int CEXP, int COUT, int S, int NW, bool RES, int NT, bool SP>
__device__ __forceinline__ void mbr_body(const MbrArgs& a, const int t0, const int w, float* lds) {
    constexpr int T = CEXP / 16, TO = (COUT + 15) / 16, NMAIN = CIN / 16, TAIL = CIN % 16, KE = NMAIN * 4 + TAIL / 4;
    constexpr int NREG = KE + 4 * TO, NOUT = 14 / S;
    constexpr int NKE = (CIN + 31) / 32, NP = (NT + 1) / 2;   // SP: K = 32 steps of the expand conv, tile pairs of this wave
    static_assert(TAIL == 0 || TAIL == 8, "block input width must be 16 n or 16 n + 8");
    static_assert(CEXP % 16 == 0 && COUT % 4 == 0, "widths");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    // ---- which strip segment
    int bid = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int seg = bid % a.segs; bid /= a.segs;
    const int strip = bid % a.strips;
    const int b = bid / a.strips;
    const int yo0 = seg * a.seg_rows, yo1 = min(yo0 + a.seg_rows, a.Ho);
    // stride 1: lane = input column, outputs at lanes 1..14; stride 2: even columns in lanes 0..7, odd ones in 8..15, the outputs
    // of an even output row in lanes 0..6, of the odd row below it in lanes 8..14 (see mbr_dw_row2)
    const int podd = S == 2 ? px >> 3 : 0;
    const int xin = S * NOUT * strip - a.pad_l + (S == 2 ? 2 * (px & 7) + podd : px);
    const int xc = min(max(xin, 0), a.W - 1);
    const float hi = (xin >= 0 && xin < a.W) ? 6.f : 0.f;
    const int jo = S == 2 ? (px & 7) : px - 1, xo = NOUT * strip + jo;
    const bool out_lane = (S == 2 ? (px & 7) < 7 : (px >= 1 && px <= 14)) && xo < a.Wo;

    // ---- stationary A fragments of this wave's tiles
    float we[SP ? 1 : NT][SP ? 1 : KE], wp[SP ? 1 : NT][TO][4];
    mbs_u4 weh[SP ? NT : 1][NKE], wem[SP ? NT : 1][NKE], wph[SP ? NP : 1][TO], wpm[SP ? NP : 1][TO];
    v4f se[NT];
    if constexpr (!SP) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float* p = a.wa + ((size_t)(t0 + j) * NREG) * 64 + lane;
#pragma unroll
            for (int q = 0; q < KE; ++q) we[j][q] = p[q * 64];
#pragma unroll
            for (int t = 0; t < TO; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) wp[j][t][s] = p[(KE + 4 * t + s) * 64];
        }
    } else {   // [T][NKE][2 planes][64 lanes] x 16 bytes, then per (wave's tile pair)[TO][2 planes][64] (compiler.mbs_pack)
        const mbs_u4* pe = reinterpret_cast<const mbs_u4*>(a.wa);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int c = 0; c < NKE; ++c) {
                weh[j][c] = pe[(((size_t)(t0 + j) * NKE + c) * 2 + 0) * 64 + lane];
                wem[j][c] = pe[(((size_t)(t0 + j) * NKE + c) * 2 + 1) * 64 + lane];
            }
        constexpr int TT = CEXP / 16, NTL_ = TT / NW, R_ = TT % NW, NTH_ = NTL_ + (R_ ? 1 : 0);
        const int pair0 = w < R_ ? w * ((NTH_ + 1) / 2) : R_ * ((NTH_ + 1) / 2) + (w - R_) * ((NTL_ + 1) / 2);   // the pairs of the waves before this one
        const mbs_u4* pp = pe + (size_t)TT * NKE * 2 * 64;
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int t = 0; t < TO; ++t) {
                wph[q][t] = pp[(((size_t)(pair0 + q) * TO + t) * 2 + 0) * 64 + lane];
                wpm[q][t] = pp[(((size_t)(pair0 + q) * TO + t) * 2 + 1) * 64 + lane];
            }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) se[j] = *reinterpret_cast<const v4f*>(a.wt + (size_t)(t0 + j) * MBR_TAB + 160 + 4 * mg);
    // ---- the depthwise table of all tiles -> LDS
    float* tab = lds;
    for (int i = threadIdx.x; i < T * MBR_TAB; i += 64 * NW) tab[i] = a.wt[i];
    v4f* red = reinterpret_cast<v4f*>(lds + T * MBR_TAB);   // [2][NW][TO][64]
    __syncthreads();

    // Every global access of the walk is UNCONDITIONAL, through buffer descriptors (dead lanes pass an offset beyond
    // num_records): the compiler can then COUNT the outstanding operations, and the wait for the next row's pixels does
    // not wait for the store just issued (a store under a per-lane branch turns every later wait into vmcnt(0)).
    const mbr_rsrc xsrc = mbr_make_rsrc(a.x + (size_t)b * a.H * a.W * a.ld_in, (unsigned)(a.H * a.W * a.ld_in) * 4u);
    const mbr_rsrc osrc = mbr_make_rsrc(a.out + (size_t)b * a.Ho * a.Wo * a.ld_out, (unsigned)(a.Ho * a.Wo * a.ld_out) * 4u);
    const int rbeg = S * yo0 - a.pad_t, nout = yo1 - yo0;
    const unsigned xoff = ((unsigned)xc * (unsigned)a.ld_in + 4u * mg) * 4u, xtoff = ((unsigned)xc * (unsigned)a.ld_in + 16u * NMAIN + 2u * mg) * 4u;
    const unsigned xrow = (unsigned)(a.W * a.ld_in) * 4u;
    // the B operands of a row: two register sets used alternately (the loads of row r + 1 are issued at the start of row r)
    struct XRow { v4f m[SP ? 2 * NKE : (NMAIN > 0 ? NMAIN : 1)]; v2f t; };
    XRow xa, xb;
    // SP: step c takes the lane's channels 32 c + 8 g .. + 7 (two 16-byte loads; a group beyond the block input reads zeros)
    unsigned xsoff[SP ? NKE : 1];
    if constexpr (SP) {
#pragma unroll
        for (int c = 0; c < NKE; ++c) xsoff[c] = 32 * c + 8 * mg < CIN ? ((unsigned)xc * (unsigned)a.ld_in + 32u * c + 8u * mg) * 4u : MBR_DEAD;
    }
    auto load_row = [&](XRow& x, int r) {
        const unsigned so = (unsigned)min(max(r, 0), a.H - 1) * xrow;
        if constexpr (SP) {
#pragma unroll
            for (int c = 0; c < NKE; ++c) {
                x.m[2 * c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xsoff[c], so, 0));
                x.m[2 * c + 1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xsoff[c] == MBR_DEAD ? MBR_DEAD : xsoff[c] + 16u, so, 0));
            }
        } else {
#pragma unroll
            for (int c = 0; c < NMAIN; ++c) x.m[c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff + 64u * c, so, 0));
            if constexpr (TAIL != 0) x.t = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(xsrc, xtoff, so, 0));
        }
    };
    load_row(xa, rbeg);
    // the cout tiles this wave finishes (tile t = w + tt * NW): BN shift once, the residual one row ahead of its use
    constexpr int NF = (TO + NW - 1) / NW;
    v4f fsh[NF], fres[NF];
    bool flive[NF];
#pragma unroll
    for (int tt = 0; tt < NF; ++tt) {
        const int t = w + tt * NW, co = 16 * t + 4 * mg;
        flive[tt] = t < TO && out_lane && co < COUT;
        fsh[tt] = *reinterpret_cast<const v4f*>(a.bp + (co < 16 * TO ? co : 0));
        fres[tt] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    // the last two expanded rows of this wave's tiles (after ReLU6, zero outside the image): with the new row, the three tap rows
    v4f ea[NT], eb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { ea[j] = (v4f){0.f, 0.f, 0.f, 0.f}; eb[j] = ea[j]; }
    int buf = 0;

    // one input row r = rbeg + k of the walk.  Stride 1: PH = 1 if it is the last tap row of output row yo (rows r - 2, r - 1, r), else 0.
    // Stride 2: PH = 0 warms the ring up (row rbeg); PH = 1..4 are the four input rows of a PAIR of output rows (yo, yo + 1): taps
    // 0-1-2 of row yo are the previous row 4 (ea) and rows 1, 2; those of row yo + 1 are rows 2, 3, 4; row 4 projects and stores.
    v4f d2[S == 2 ? NT : 1];
    auto row = [&](auto ph_c, const int k, const int yo, const XRow& xc_, XRow& xn_) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr bool EMIT = S == 2 ? PH == 4 : PH == 1;
        const int r = rbeg + k;
        load_row(xn_, r + 1);
        float xq[SP ? 1 : KE];
        mbs_u4 xh[SP ? NKE : 1], xm[SP ? NKE : 1];
        if constexpr (SP) {
#pragma unroll
            for (int c = 0; c < NKE; ++c) {
                const float v[8] = {xc_.m[2 * c][0], xc_.m[2 * c][1], xc_.m[2 * c][2], xc_.m[2 * c][3], xc_.m[2 * c + 1][0], xc_.m[2 * c + 1][1], xc_.m[2 * c + 1][2], xc_.m[2 * c + 1][3]};
                mbs_split8(v, xh[c], xm[c]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < NMAIN; ++c) { xq[4 * c] = xc_.m[c][0]; xq[4 * c + 1] = xc_.m[c][1]; xq[4 * c + 2] = xc_.m[c][2]; xq[4 * c + 3] = xc_.m[c][3]; }
            if constexpr (TAIL != 0) { xq[4 * NMAIN] = xc_.t[0]; xq[4 * NMAIN + 1] = xc_.t[1]; }
        }
        if constexpr (EMIT && RES) {
#pragma unroll
            for (int tt = 0; tt < NF; ++tt) {
                const unsigned off = flive[tt] ? (((unsigned)yo * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_in + 16u * (w + tt * NW) + 4u * mg) * 4u : MBR_DEAD;
                fres[tt] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
            }
        }
        const float hr = (r >= 0 && r < a.H) ? hi : 0.f;
        // ---- expand: NT independent accumulator chains, step-major
        v4f ec[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) ec[j] = se[j];
        if constexpr (SP) {
            v4f e1[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) e1[j] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NKE; ++c) {
#pragma unroll
                for (int j = 0; j < NT; ++j) ec[j] = mbs_mfma(weh[j][c], xh[c], ec[j]);
#pragma unroll
                for (int j = 0; j < NT; ++j) e1[j] = mbs_mfma(weh[j][c], xm[c], e1[j]);
#pragma unroll
                for (int j = 0; j < NT; ++j) e1[j] = mbs_mfma(wem[j][c], xh[c], e1[j]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) ec[j] = __builtin_elementwise_fma(e1[j], (v4f){0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f}, ec[j]);   // (an FMA by name: the library is built without contraction)
        } else {
#pragma unroll
            for (int q = 0; q < KE; ++q)
#pragma unroll
                for (int j = 0; j < NT; ++j) ec[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(we[j][q], xq[q], ec[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) ec[j][i] = __builtin_amdgcn_fmed3f(ec[j][i], 0.f, hr);

        v4f P[TO], P1[SP ? TO : 1];
        if constexpr (EMIT) {
#pragma unroll
            for (int t = 0; t < TO; ++t) P[t] = (v4f){0.f, 0.f, 0.f, 0.f};
            if constexpr (SP) {
#pragma unroll
                for (int t = 0; t < TO; ++t) P1[t] = (v4f){0.f, 0.f, 0.f, 0.f};
            }
        }
        // SP: the projection of one tile pair - the 8 depthwise results a lane holds are its 8 k values of the step
        auto project_pair = [&](const int q, const v4f dA, const v4f dB) {
            const float v[8] = {dA[0], dA[1], dA[2], dA[3], dB[0], dB[1], dB[2], dB[3]};
            mbs_u4 bh, bm;
            mbs_split8(v, bh, bm);
#pragma unroll
            for (int t = 0; t < TO; ++t) {
                P[t] = mbs_mfma(wph[q][t], bh, P[t]);
                P1[t] = mbs_mfma(wph[q][t], bm, P1[t]);
                P1[t] = mbs_mfma(wpm[q][t], bh, P1[t]);
            }
        };
        if constexpr (S == 2 && PH != 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const v4f* tb = reinterpret_cast<const v4f*>(tab + (t0 + j) * MBR_TAB) + mg;
                if constexpr (PH == 1) {
                    d2[j] = tb[36];   // the BN shift is the first addend, in every lane
                    mbr_dw_row2<false>(d2[j], ea[j], tb[0], tb[4], tb[8]);
                    mbr_dw_row2<false>(d2[j], ec[j], tb[12], tb[16], tb[20]);
                } else if constexpr (PH == 2) {
                    mbr_dw_row2<false>(d2[j], ec[j], tb[24], tb[28], tb[32]);
                    mbr_dw_row2<true>(d2[j], ec[j], tb[0], tb[4], tb[8]);
                } else if constexpr (PH == 3) {
                    mbr_dw_row2<true>(d2[j], ec[j], tb[12], tb[16], tb[20]);
                } else {
                    mbr_dw_row2<true>(d2[j], ec[j], tb[24], tb[28], tb[32]);
                    v4f d = d2[j];
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_fmed3f(d[i], 0.f, 6.f);
                    if constexpr (SP) {
                        d2[j] = d;
                        if (j % 2 == 1) project_pair(j / 2, d2[j - 1], d2[j]);   // (j is a constant once unrolled)
                        else if (j == NT - 1) project_pair(j / 2, d2[j], (v4f){0.f, 0.f, 0.f, 0.f});
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int t = 0; t < TO; ++t) P[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[j][t][s], d[s], P[t], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (S == 1 && EMIT) {
            v4f dprev = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const v4f* tb = reinterpret_cast<const v4f*>(tab + (t0 + j) * MBR_TAB) + mg;
                v4f d = MBR_TB(36);   // the BN shift is the first addend
                mbr_dw_row(d, ea[j], MBR_TB(0), MBR_TB(4), MBR_TB(8));
                mbr_dw_row(d, eb[j], MBR_TB(12), MBR_TB(16), MBR_TB(20));
                mbr_dw_row(d, ec[j], MBR_TB(24), MBR_TB(28), MBR_TB(32));
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_fmed3f(d[i], 0.f, 6.f);
                if constexpr (SP) {
                    if (j % 2 == 1) project_pair(j / 2, dprev, d);   // (j is a constant once unrolled)
                    else if (j == NT - 1) project_pair(j / 2, d, (v4f){0.f, 0.f, 0.f, 0.f});
                    dprev = d;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < TO; ++t) P[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[j][t][s], d[s], P[t], 0, 0, 0);
                }
            }
        }
        if constexpr (EMIT && SP) {
#pragma unroll
            for (int t = 0; t < TO; ++t) P[t] = __builtin_elementwise_fma(P1[t], (v4f){0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f}, P[t]);
        }
        if constexpr (EMIT) {
            const int yl = yo + podd;   // (stride 2: lanes 8..15 hold the row below)
            const bool rlive = S == 1 || yl < yo1;
            const unsigned opix = ((unsigned)yl * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_out * 4u;
            if constexpr (NW == 1) {
#pragma unroll
                for (int t = 0; t < TO; ++t) {
                    v4f v = P[t] + fsh[t];
                    if (RES) v += fres[t];   // (after the sum: the load issued at the row's start is not waited for before here)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), osrc, flive[t] && rlive ? opix + (16u * t + 4u * mg) * 4u : MBR_DEAD, 0, 0);
                }
            } else {
                v4f* rb = red + buf * (NW * TO * 64);
#pragma unroll
                for (int t = 0; t < TO; ++t) rb[(w * TO + t) * 64 + lane] = P[t];
                __syncthreads();
#pragma unroll
                for (int tt = 0; tt < NF; ++tt) {
                    const int t = w + tt * NW;
                    v4f v = fsh[tt];
                    if (t < TO) {   // (wave-uniform; LDS reads only - the store below stays unconditional)
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) v += rb[(ww * TO + t) * 64 + lane];
                    }
                    if (RES) v += fres[tt];   // last: the residual load issued at the row's start is waited for only here
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), osrc, flive[tt] && rlive ? opix + (16u * t + 4u * mg) * 4u : MBR_DEAD, 0, 0);
                }
                buf ^= 1;
            }
        }
        if constexpr (S == 1) {
#pragma unroll
            for (int j = 0; j < NT; ++j) { ea[j] = eb[j]; eb[j] = ec[j]; }
        } else if constexpr (PH == 0 || PH == 4) {
#pragma unroll
            for (int j = 0; j < NT; ++j) ea[j] = ec[j];
        }
    };
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    if constexpr (S == 2) {
        constexpr std::integral_constant<int, 2> P2{};
        constexpr std::integral_constant<int, 3> P3{};
        constexpr std::integral_constant<int, 4> P4{};
        row(P0, 0, 0, xa, xb);
        for (int i = 0; i < nout; i += 2) {   // a pair of output rows per turn (an odd segment's last pair stores its even row only)
            row(P1, 2 * i + 1, 0, xb, xa);
            row(P2, 2 * i + 2, 0, xa, xb);
            row(P3, 2 * i + 3, 0, xb, xa);
            row(P4, 2 * i + 4, yo0 + i, xa, xb);
        }
    } else {   // rows 0, 1 warm the ring up; then every input row emits
        row(P0, 0, 0, xa, xb);
        row(P0, 1, 0, xb, xa);
        int i = 0;
        for (; i + 1 < nout; i += 2) {
            row(P1, i + 2, yo0 + i, xa, xb);
            row(P1, i + 3, yo0 + i + 1, xb, xa);
        }
        if (i < nout) row(P1, i + 2, yo0 + i, xa, xb);
    }
}

template <int CIN, int CEXP, int COUT, int S, int NW, bool RES, bool SP, int MW>
__global__ __launch_bounds__(64 * NW, MW) void mbr_kernel(MbrArgs a) {
    constexpr int T = CEXP / 16, NTL = T / NW, R = T % NW, NTH = NTL + (R ? 1 : 0);
    static_assert(NTL >= 1, "more waves than expanded tiles");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (R != 0) {
        if (w < R) { mbr_body<CIN, CEXP, COUT, S, NW, RES, NTH, SP>(a, w * NTH, w, lds); return; }
    }
    mbr_body<CIN, CEXP, COUT, S, NW, RES, NTL, SP>(a, R * NTH + (w - R) * NTL, w, lds);
}

template <int CIN, int CEXP, int COUT, int S, int NW, bool RES, bool SP = false>
static int launch_mbr(const MbrArgs& a0, int batch, int want_segs, hipStream_t s) {
    MbrArgs a = a0;
    constexpr int T = CEXP / 16, TO = (COUT + 15) / 16, NOUT = 14 / S;
    a.strips = (a.Wo + NOUT - 1) / NOUT;
    // segments: enough workgroups for ~2 generations of the chip's SIMDs, not so many that the 2 halo rows matter
    const int walks = batch * a.strips;
    int segs = (2 * 1024 + walks * NW - 1) / (walks * NW);
    const int max_segs = (a.Ho + 5) / 6;
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    if (want_segs > 0) segs = want_segs < a.Ho ? want_segs : a.Ho;
    a.seg_rows = (a.Ho + segs - 1) / segs;
    if (S == 2) a.seg_rows += a.seg_rows & 1;   // (output rows are processed in pairs)
    a.segs = (a.Ho + a.seg_rows - 1) / a.seg_rows;
    const size_t lds = (size_t)T * MBR_TAB * 4 + (NW > 1 ? (size_t)2 * NW * TO * 64 * 16 : 0);
    static char nm[64];
    static const int nm_len = snprintf(nm, sizeof(nm), "mbr_kernel<%d,%d,%d,%d,%d,%d,%d>", CIN, CEXP, COUT, S, NW, (int)RES, (int)SP);   // (... SP = 1: the split form)
    (void)nm_len;
    yr_note_kernel(nm);
    // waves per SIMD the register allocator must leave room for, from an estimate of what a wave holds: the stationary
    // fragments + BN shifts, ring and new row, projection accumulators, two rows of pixel operands, ~44 others
    constexpr int NTH = (T + NW - 1) / NW, KE = CIN / 4;
    constexpr int NKE = (CIN + 31) / 32, NPH = (NTH + 1) / 2;
    constexpr int EST = SP ? NTH * (8 * NKE + 4 + 12 + 4) + NPH * 8 * TO + 8 * TO + 16 * NKE + 8 * NKE + 60
                           : NTH * (KE + 4 * TO + 4 + 12) + 4 * TO + 2 * KE + 44;
    // (SP: measured register counts - the stride-1 three-wave kernels fit 168, the others spill there)
    constexpr int MW = SP ? (S == 1 && NW <= 3 ? 3 : 2) : EST <= 164 ? 3 : EST <= 250 ? 2 : 1;
    static_assert(NW <= 4 * MW, "a workgroup's waves must fit one CU at this register budget");
    auto kern = mbr_kernel<CIN, CEXP, COUT, S, NW, RES, SP, MW>;   // waves per SIMD the register allocator must leave room for
    static bool attr_set_dev[64] = {};   // per device: a process that drives several GPUs needs the attribute on each
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool& attr_set = attr_set_dev[cur_dev & 63];
    if (!attr_set && lds > 48 * 1024) {
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(batch * a.strips * a.segs)), dim3(64 * NW), lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// YR_OP_MBE: the first two thirds of the block - expand 1x1 + BN + ReLU6 -> depthwise 3x3 + BN + ReLU6 - in the same
// register-chained form, for blocks whose three A-fragment sets do not fit one CU's register file (MobileNetV2 x0.75 block_11
// on: 72 -> 432 -> 72 and wider).  Without the projection nothing couples the expanded channels: every WAVE is on its own - a
// strip segment x NT expanded tiles - and the depthwise map is stored (16 bytes per lane and tile); the projection stays a
// pointwise op.  The 6x-wide EXPAND output (written once, read once: half of the unfused chain's bytes) never exists.
struct MbeArgs {
    const float* x; float* out;
    const float* wa;   // expand A fragments [T][KE][64]
    const float* wt;   // [T][11][16]: depthwise taps x BN scale | depthwise BN shift | expand BN shift
    int H, W, Ho, Wo, ld_in, ld_out, pad_t, pad_l, strips, segs, seg_rows, T, groups, nwaves;
};

template <int CIN, int S, int NT, int MW, bool SP>
__global__ __launch_bounds__(256, MW) void mbe_kernel(MbeArgs a) {
    constexpr int NMAIN = CIN / 16, TAIL = CIN % 16, KE = NMAIN * 4 + TAIL / 4, NOUT = 14 / S;
    constexpr int NKE = (CIN + 31) / 32;   // SP (the split form, see mbs_split8): K = 32 steps of the expand conv
    static_assert(TAIL == 0 || TAIL == 8, "block input width must be 16 n or 16 n + 8");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float tab[];
    for (int i = threadIdx.x; i < a.T * MBR_TAB; i += 256) tab[i] = a.wt[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    int gw = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (gw >= a.nwaves) return;
    const int g = gw % a.groups; gw /= a.groups;      // (the tile groups of one strip segment are neighbours: same pixels, L1 / L2)
    const int seg = gw % a.segs; gw /= a.segs;
    const int strip = gw % a.strips;
    const int b = gw / a.strips;
    const int t0 = g * NT;
    const int yo0 = seg * a.seg_rows, yo1 = min(yo0 + a.seg_rows, a.Ho);
    const int xin = S * NOUT * strip - a.pad_l + px;
    const int xc = min(max(xin, 0), a.W - 1);
    const float hi = (xin >= 0 && xin < a.W) ? 6.f : 0.f;
    const int xo = NOUT * strip + (px - 1) / S;
    const bool out_lane = px >= 1 && px <= 14 && (px - 1) % S == 0 && xo < a.Wo;

    float we[SP ? 1 : NT][SP ? 1 : KE];
    mbs_u4 weh[SP ? NT : 1][NKE], wem[SP ? NT : 1][NKE];
    v4f se[NT];
    unsigned ooff[NT];   // byte offset of this lane's 4 channels of tile j within a pixel, or dead
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = min(t0 + j, a.T - 1);           // (a short last group recomputes the last tile; its stores are dead)
        if constexpr (SP) {
            const mbs_u4* pe = reinterpret_cast<const mbs_u4*>(a.wa) + ((size_t)t * NKE) * 2 * 64 + lane;
#pragma unroll
            for (int c = 0; c < NKE; ++c) { weh[j][c] = pe[(2 * c) * 64]; wem[j][c] = pe[(2 * c + 1) * 64]; }
        } else {
            const float* p = a.wa + ((size_t)t * KE) * 64 + lane;
#pragma unroll
            for (int q = 0; q < KE; ++q) we[j][q] = p[q * 64];
        }
        se[j] = *reinterpret_cast<const v4f*>(a.wt + (size_t)t * MBR_TAB + 160 + 4 * mg);
        ooff[j] = (t0 + j < a.T && out_lane) ? (16u * (t0 + j) + 4u * mg) * 4u : MBR_DEAD;
    }
    const mbr_rsrc xsrc = mbr_make_rsrc(a.x + (size_t)b * a.H * a.W * a.ld_in, (unsigned)(a.H * a.W * a.ld_in) * 4u);
    const mbr_rsrc osrc = mbr_make_rsrc(a.out + (size_t)b * a.Ho * a.Wo * a.ld_out, (unsigned)(a.Ho * a.Wo * a.ld_out) * 4u);
    const int rbeg = S * yo0 - a.pad_t, nout = yo1 - yo0;
    const unsigned xoff = ((unsigned)xc * (unsigned)a.ld_in + 4u * mg) * 4u, xtoff = ((unsigned)xc * (unsigned)a.ld_in + 16u * NMAIN + 2u * mg) * 4u;
    const unsigned xrow = (unsigned)(a.W * a.ld_in) * 4u;
    struct XRow { v4f m[SP ? 2 * NKE : (NMAIN > 0 ? NMAIN : 1)]; v2f t; };
    XRow xa, xb;
    unsigned xsoff[SP ? NKE : 1];   // SP: step c takes the lane's channels 32 c + 8 g .. + 7 (a group beyond the block input reads zeros)
    if constexpr (SP) {
#pragma unroll
        for (int c = 0; c < NKE; ++c) xsoff[c] = 32 * c + 8 * mg < CIN ? ((unsigned)xc * (unsigned)a.ld_in + 32u * c + 8u * mg) * 4u : MBR_DEAD;
    }
    auto load_row = [&](XRow& x, int r) {
        const unsigned so = (unsigned)min(max(r, 0), a.H - 1) * xrow;
        if constexpr (SP) {
#pragma unroll
            for (int c = 0; c < NKE; ++c) {
                x.m[2 * c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xsoff[c], so, 0));
                x.m[2 * c + 1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xsoff[c] == MBR_DEAD ? MBR_DEAD : xsoff[c] + 16u, so, 0));
            }
        } else {
#pragma unroll
            for (int c = 0; c < NMAIN; ++c) x.m[c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff + 64u * c, so, 0));
            if constexpr (TAIL != 0) x.t = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(xsrc, xtoff, so, 0));
        }
    };
    load_row(xa, rbeg);
    v4f ea[NT], eb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { ea[j] = (v4f){0.f, 0.f, 0.f, 0.f}; eb[j] = ea[j]; }

    auto row = [&](auto emit_c, const int k, const int yo, const XRow& xc_, XRow& xn_) {
        constexpr bool EMIT = decltype(emit_c)::value;
        const int r = rbeg + k;
        load_row(xn_, r + 1);
        const float hr = (r >= 0 && r < a.H) ? hi : 0.f;
        v4f ec[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) ec[j] = se[j];
        if constexpr (SP) {
            v4f e1[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) e1[j] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NKE; ++c) {
                const float v[8] = {xc_.m[2 * c][0], xc_.m[2 * c][1], xc_.m[2 * c][2], xc_.m[2 * c][3], xc_.m[2 * c + 1][0], xc_.m[2 * c + 1][1], xc_.m[2 * c + 1][2], xc_.m[2 * c + 1][3]};
                mbs_u4 xh, xm;
                mbs_split8(v, xh, xm);
#pragma unroll
                for (int j = 0; j < NT; ++j) ec[j] = mbs_mfma(weh[j][c], xh, ec[j]);
#pragma unroll
                for (int j = 0; j < NT; ++j) e1[j] = mbs_mfma(weh[j][c], xm, e1[j]);
#pragma unroll
                for (int j = 0; j < NT; ++j) e1[j] = mbs_mfma(wem[j][c], xh, e1[j]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) ec[j] = __builtin_elementwise_fma(e1[j], (v4f){0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f}, ec[j]);   // (an FMA by name: the library is built without contraction)
        } else {
            float xq[KE];
#pragma unroll
            for (int c = 0; c < NMAIN; ++c) { xq[4 * c] = xc_.m[c][0]; xq[4 * c + 1] = xc_.m[c][1]; xq[4 * c + 2] = xc_.m[c][2]; xq[4 * c + 3] = xc_.m[c][3]; }
            if constexpr (TAIL != 0) { xq[4 * NMAIN] = xc_.t[0]; xq[4 * NMAIN + 1] = xc_.t[1]; }
#pragma unroll
            for (int q = 0; q < KE; ++q)
#pragma unroll
                for (int j = 0; j < NT; ++j) ec[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(we[j][q], xq[q], ec[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) ec[j][i] = __builtin_amdgcn_fmed3f(ec[j][i], 0.f, hr);
        if constexpr (EMIT) {
            const unsigned opix = ((unsigned)yo * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_out * 4u;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const v4f* tb = reinterpret_cast<const v4f*>(tab + min(t0 + j, a.T - 1) * MBR_TAB) + mg;
                v4f d = tb[36];
                mbr_dw_row(d, ea[j], tb[0], tb[4], tb[8]);
                mbr_dw_row(d, eb[j], tb[12], tb[16], tb[20]);
                mbr_dw_row(d, ec[j], tb[24], tb[28], tb[32]);
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_fmed3f(d[i], 0.f, 6.f);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, d), osrc, ooff[j] == MBR_DEAD ? MBR_DEAD : opix + ooff[j], 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) { ea[j] = eb[j]; eb[j] = ec[j]; }
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    if constexpr (S == 2) {
        row(N, 0, 0, xa, xb);
        for (int i = 0; i < nout; ++i) {
            row(N, 2 * i + 1, 0, xb, xa);
            row(Y, 2 * i + 2, yo0 + i, xa, xb);
        }
    } else {
        row(N, 0, 0, xa, xb);
        row(N, 1, 0, xb, xa);
        int i = 0;
        for (; i + 1 < nout; i += 2) {
            row(Y, i + 2, yo0 + i, xa, xb);
            row(Y, i + 3, yo0 + i + 1, xb, xa);
        }
        if (i < nout) row(Y, i + 2, yo0 + i, xa, xb);
    }
}

template <int CIN, int S, int NT, bool SP = false>
static int launch_mbe(const MbeArgs& a0, int batch, int want_segs, hipStream_t s) {
    MbeArgs a = a0;
    constexpr int NOUT = 14 / S, KE = SP ? 8 * ((CIN + 31) / 32) : CIN / 4;
    constexpr int EST = NT * (KE + 16 + (SP ? 4 : 0)) + 2 * KE + (SP ? 8 : 0) + 44;      // registers a wave holds (stationary fragments, ring, two rows of pixels, ~44 others)
    constexpr int MW = EST <= 150 ? 3 : EST <= 250 ? 2 : 1;
    a.strips = (a.Wo + NOUT - 1) / NOUT;
    a.groups = (a.T + NT - 1) / NT;
    const int walks = batch * a.strips * a.groups;             // waves at one segment per strip
    int segs = (3 * 1024 + walks - 1) / walks;                 // ~3 waves per SIMD of the chip
    const int max_segs = (a.Ho + 5) / 6;
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    if (want_segs > 0) segs = want_segs < a.Ho ? want_segs : a.Ho;
    a.seg_rows = (a.Ho + segs - 1) / segs;
    a.segs = (a.Ho + a.seg_rows - 1) / a.seg_rows;
    a.nwaves = batch * a.strips * a.segs * a.groups;
    const size_t lds = (size_t)a.T * MBR_TAB * 4;
    YR_REQUIRE(lds <= 64 * 1024, "mbe: %d expanded channels exceed the depthwise table's LDS budget", a.T * 16);
    static char nm[48];
    static const int nm_len = snprintf(nm, sizeof(nm), "mbe_kernel<%d,%d,%d,%d,%d>", CIN, S, NT, MW, (int)SP);   // (... SP = 1: the split form)
    (void)nm_len;
    yr_note_kernel(nm);
    auto kern = mbe_kernel<CIN, S, NT, MW, SP>;
    static bool attr_set_dev[64] = {};   // per device
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool& attr_set = attr_set_dev[cur_dev & 63];
    if (!attr_set) {
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.nwaves + 3) / 4)), dim3(256), lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// op fields: src[0] = block input (float32, c % 16 in {0, 8}); cout = Cexp = width of the stored depthwise map (multiple of 16);
// k = 3 | segs << 16; stride 1 | 2; act = ReLU6 (both activations).  wgt = expand A fragments [T][KE][64] (register order of
// YR_OP_MBR, rho < KE); wgt2 = [T][11][16] as YR_OP_MBR.
int yr_launch_mbe(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.dtype == YR_F32 && op.out_dtype == YR_F32, "mbe: float32 plans only");
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY && op.src[0].dtype == YR_F32, "mbe: needs one float32 identity source");
    const yr_src& in = op.src[0];
    YR_REQUIRE((op.k & 0x7f) == 3 && (op.stride == 1 || op.stride == 2) && op.act == YR_ACT_RELU6, "mbe: 3x3, stride 1|2, ReLU6");
    YR_REQUIRE(in.ptr && op.out && op.wgt && op.wgt2 && op.res == nullptr, "mbe: null pointer (or a residual)");
    YR_REQUIRE(in.ld % 4 == 0 && op.out_ld % 4 == 0 && in.c == op.cin && in.ld >= in.c && op.out_ld >= op.cout && op.cout % 16 == 0, "mbe: channel strides / widths");
    YR_REQUIRE(((uintptr_t)in.ptr) % 16 == 0 && ((uintptr_t)op.out) % 16 == 0, "mbe: pointers must be 16-byte aligned");
    MbeArgs a;
    a.x = (const float*)in.ptr; a.out = (float*)op.out; a.wa = op.wgt; a.wt = op.wgt2;
    a.H = in.h; a.W = in.w; a.Ho = (in.h + op.stride - 1) / op.stride; a.Wo = (in.w + op.stride - 1) / op.stride;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "mbe: output dims mismatch");
    a.ld_in = in.ld; a.ld_out = op.out_ld; a.T = op.cout / 16;
    const int pth = (a.Ho - 1) * op.stride + 3 - in.h, ptw = (a.Wo - 1) * op.stride + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.strips = a.segs = a.seg_rows = a.groups = a.nwaves = 0;
    const int segs = (op.k >> 16) & 0xff;
    // k bit 7: the SPLIT form - wgt holds the float16 planes [T][NKE][2][64 lanes][8 halves] of YR_OP_MBR's split form (expand part)
    // (NTS: tiles per wave of the split form - its two rows of pixel operands are 16 registers per K = 32 step)
#define MBE_CASE(CIN, NT, NTS)                                                                \
    if (in.c == CIN && (op.k & 0x80)) return op.stride == 1 ? launch_mbe<CIN, 1, NTS, true>(a, batch, segs, s) : launch_mbe<CIN, 2, NTS, true>(a, batch, segs, s); \
    if (in.c == CIN) return op.stride == 1 ? launch_mbe<CIN, 1, NT>(a, batch, segs, s) : launch_mbe<CIN, 2, NT>(a, batch, segs, s);
    MBE_CASE(48, 3, 3)
    MBE_CASE(72, 3, 2)      // MobileNetV2 x0.75 block_11..13
    MBE_CASE(88, 2, 2)      // x1.4 block_7..10
    MBE_CASE(120, 2, 2)     // x0.75 block_14, 15
    MBE_CASE(136, 2, 1)     // x1.4 block_11..13
#undef MBE_CASE
    if (in.c == 224 && !(op.k & 0x80)) return op.stride == 1 ? launch_mbe<224, 1, 1>(a, batch, segs, s) : launch_mbe<224, 2, 1>(a, batch, segs, s);   // x1.4 block_14, 15
    yr_set_error("mbe: block input width %d is not built", in.c);
    return YR_ERR_ARG;
}

// op fields: src[0] = block input (float32, c % 8 == 0, c % 16 in {0, 8}); se_reduced = Cexp (multiple of 16); k = 3 | split << 7 | nw << 8 | segs << 16
// (nw: waves per workgroup, segs: row segments per strip; 0 = the library's choice); stride 1 | 2; act = ReLU6; res (optional) = the block input.  Parameters (float32):
//   wgt  = A fragments [T = Cexp/16][KE + 4 TO][64]: register rho of lane (m = l % 16, g = l / 16) of tile j:
//          rho < KE (expand step q = rho): We[16 j + m][kperm(q, g)] * expand BN scale, kperm(4 c + s, g) = 16 c + 4 g + s for the
//          full 16-channel chunks, 16 n + 2 g + s (s < 2) for a trailing 8;  rho = KE + 4 t + s: Wp[16 t + m][16 j + 4 g + s] * project
//          BN scale (0 beyond cout);
//   wgt2 = [T][11][16]: depthwise taps (ky, kx) times the depthwise BN scale | depthwise BN shift | expand BN shift;
//   b2   = project BN shift [16 TO].
// split = 1: the SPLIT form (both 1x1 convolutions on the 16-bit matrix pipe, every operand as two float16 planes: see mbs_split8) -
//   wgt  = the float32 words that hold [T][NKE = ceil(cin / 32)][2 planes][64 lanes][8 halves]: We[16 j + m][32 c + 8 g + i] * BN scale
//          (h plane, then m plane = f16((w - h) 2^11)), followed by, per tile pair (tA, tB) of the nw waves in order (a wave pairs ITS tiles;
//          an odd last one pairs with nothing), [TO][2 planes][64][8]: Wp[16 t + m][16 tA + 4 g + i] (i < 4) | Wp[16 t + m][16 tB + 4 g + i - 4];
//          nw must be the value the fragments were packed for.  wgt2, b2 as above.
int yr_launch_mbr(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.dtype == YR_F32 && op.out_dtype == YR_F32, "mbr: float32 plans only");
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY && op.src[0].dtype == YR_F32, "mbr: needs one float32 identity source");
    const yr_src& in = op.src[0];
    YR_REQUIRE((op.k & 0x7f) == 3 && (op.stride == 1 || op.stride == 2) && op.act == YR_ACT_RELU6, "mbr: 3x3, stride 1|2, ReLU6");
    YR_REQUIRE(in.ptr && op.out && op.wgt && op.wgt2 && op.b2, "mbr: null pointer");
    YR_REQUIRE(in.ld % 4 == 0 && op.out_ld % 4 == 0 && in.c == op.cin && in.ld >= in.c && op.out_ld >= op.cout, "mbr: channel strides");
    YR_REQUIRE(((uintptr_t)in.ptr) % 16 == 0 && ((uintptr_t)op.out) % 16 == 0, "mbr: pointers must be 16-byte aligned");
    MbrArgs a;
    a.x = (const float*)in.ptr; a.out = (float*)op.out; a.wa = op.wgt; a.wt = op.wgt2; a.bp = op.b2;
    a.H = in.h; a.W = in.w; a.Ho = (in.h + op.stride - 1) / op.stride; a.Wo = (in.w + op.stride - 1) / op.stride;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "mbr: output dims mismatch");
    a.ld_in = in.ld; a.ld_out = op.out_ld;
    const int pth = (a.Ho - 1) * op.stride + 3 - in.h, ptw = (a.Wo - 1) * op.stride + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    const bool res = op.res != nullptr;
    if (res) YR_REQUIRE(op.res == in.ptr && op.stride == 1 && in.c == op.cout, "mbr: the residual must be the block input (stride 1, Cin == Cout)");
    a.strips = a.segs = a.seg_rows = 0;
    const int nw = (op.k >> 8) & 0xff;
    if (op.k & 0x80) {   // the SPLIT form (bit 7 of k): wgt holds float16 planes packed for exactly this many waves (compiler.mbs_pack)
#define MBS_CASE(CIN, CEXP, COUT, S, NW, RES)                                                              \
    if (in.c == CIN && op.se_reduced == CEXP && op.cout == COUT && op.stride == S && res == RES && nw == NW) \
        return launch_mbr<CIN, CEXP, COUT, S, NW, RES, true>(a, batch, (op.k >> 16) & 0xff, s);
        MBS_CASE(16, 96, 24, 2, 3, false)
        MBS_CASE(16, 96, 24, 2, 2, false)     // (block_1 @208: 0.188 ms against 0.195-0.204 with three waves - one partner less at the row barrier)
        MBS_CASE(24, 144, 24, 1, 3, true)
        MBS_CASE(24, 144, 24, 2, 3, false)
        MBS_CASE(24, 144, 48, 2, 3, false)
        MBS_CASE(48, 288, 48, 1, 6, true)
        MBS_CASE(48, 288, 72, 1, 6, false)
        MBS_CASE(24, 144, 32, 2, 3, false)     // MobileNetV2 x1.4 block_1
        MBS_CASE(32, 192, 32, 1, 4, true)      // x1.4 block_2
        MBS_CASE(32, 192, 48, 2, 4, false)     // x1.4 block_3
#undef MBS_CASE
        yr_set_error("mbr (split form): block %d -> %d -> %d stride %d res %d nw %d is not built", in.c, op.se_reduced, op.cout, op.stride, (int)res, nw);
        return YR_ERR_ARG;
    }
#define MBR_CASE(CIN, CEXP, COUT, S, NW, RES)                                                              \
    if (in.c == CIN && op.se_reduced == CEXP && op.cout == COUT && op.stride == S && res == RES && (nw == 0 || nw == NW)) \
        return launch_mbr<CIN, CEXP, COUT, S, NW, RES>(a, batch, (op.k >> 16) & 0xff, s);
    MBR_CASE(16, 96, 24, 2, 2, false)      // MobileNetV2 x0.75 block_1
    MBR_CASE(16, 96, 24, 2, 3, false)
    MBR_CASE(24, 144, 24, 1, 3, true)      // block_2, 4, 5
    MBR_CASE(24, 144, 24, 2, 3, false)     // block_3 (x0.75: 32 * 0.75 = 24 outputs)
    MBR_CASE(24, 144, 48, 2, 3, false)     // block_6
    MBR_CASE(24, 144, 32, 2, 3, false)     // MobileNetV2 x1.4 block_1 / x1.0 block_3
    MBR_CASE(32, 192, 32, 1, 4, true)      // x1.4 block_2 / x1.0 block_4, 5
    MBR_CASE(32, 192, 48, 2, 4, false)     // x1.4 block_3
    MBR_CASE(48, 288, 48, 1, 6, true)      // block_7..9
    MBR_CASE(48, 288, 48, 1, 8, true)
    MBR_CASE(48, 288, 72, 1, 6, false)     // block_10
    MBR_CASE(48, 288, 72, 1, 8, false)
#undef MBR_CASE
    yr_set_error("mbr: block %d -> %d -> %d stride %d res %d (nw %d) is not built", in.c, op.se_reduced, op.cout, op.stride, (int)res, nw);
    return YR_ERR_ARG;
}
