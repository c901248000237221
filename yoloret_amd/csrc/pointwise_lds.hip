// Pointwise (1x1) convolution, LDS-staged MFMA kernel (see pointwise.hip for the GEMM view, the direct
// variant and the dispatcher).  A 16-wide k chunk is staged in LDS by coalesced loads (4 adjacent lanes = 64
// bytes of a row); lane group g reads k = k0+4g..4g+3 as one ds_read_b128 and uses component s in MFMA step s.
// The next chunk's (small tiles: the next two chunks') global loads are issued before the current chunk's MFMAs
// (register double buffering); every load is unconditional, so the compiler waits for them by count.
#include <type_traits>

#include "pw_common.h"

#ifndef PW_BK
#define PW_BK 16                 // k depth staged per barrier pair (multiple of 16; measured: 32 is 6 % and 64 is 13 % slower end to end)
#endif
#ifndef PW_PF2_MAX_TILES
#define PW_PF2_MAX_TILES 4       // tiles (PT*CT) per wave up to which TWO k chunks are prefetched (measured end to end:
                                 // 3 and 4 are equal, 6 is 1 % and "all" 1.5 % slower - the second register set costs occupancy)
#endif
#define PW_KQ (PW_BK / 4)        // float4 quads per staged row
#define PW_RPP (256 / PW_KQ)     // rows loaded per pass of the 256 threads
#define PW_LDS_LD (PW_BK + 4)    // padded row stride (floats): an odd number of 16-byte slots

// PT/CT: 16-wide pixel / cout MFMA tiles per wave; WM x WN waves (WM*WN == 4).
// SIMPLE: one identity source (optionally SE-gated) -> a pixel's channels are one contiguous row (the
// common case: every backbone expand/project and most head convs); otherwise the generic gather
// through per-source row pointers (upsample / maxpool / concat folded into the loads).
// Blocks per CU the register allocator must make room for.  Left alone the compiler is generous (84-172 registers);
// asked, it fits the same code into 54-128 without spilling (the exceptions below are the shapes that would spill;
// 5 blocks for the 4-7 tile shapes measured no better than 4).
constexpr int pw_min_blocks(int pt, int ct, bool simple) {
    const int tiles = pt * ct;
    return tiles >= 16 ? 2 : (pt == 4 && ct == 1 && !simple) ? 2 : (pt == 4 && (ct == 1 || (!simple && ct >= 3))) ? 3 : tiles >= 4 ? 4 : (simple ? 6 : 5);
}

// DW (with SIMPLE): the single source is read through a 3x3 depthwise conv + BN + activation (YR_X_DW3, PwDwRow):
// nine tap quads are fetched per activation quad and reduced in the stage; the depthwise weights, scale and shift of
// ALL k live in dynamic LDS (11 * kp floats, loaded once per workgroup).
template <int PT, int CT, int WM, int WN, bool SIMPLE, bool DW = false>
__global__ __launch_bounds__(256, DW ? 3 : pw_min_blocks(PT, CT, SIMPLE)) void pw_kernel(PwArgs a) {
    static_assert(!DW || SIMPLE, "a depthwise-folded source is a single source");
    constexpr int BM = 16 * PT * WM;
    constexpr int BN = 16 * CT * WN;
    constexpr int A_PASSES = BM / PW_RPP;
    constexpr int B_PASSES = (BN + PW_RPP - 1) / PW_RPP;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * PW_LDS_LD];
    float* As = lds;                   // [BM][PW_LDS_LD] activations
    float* Bs = lds + BM * PW_LDS_LD;  // [BN][PW_LDS_LD] weights
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
    const int lr = tid / PW_KQ, kq = tid % PW_KQ;
    constexpr int MODE = SIMPLE ? 2 : 0;
    constexpr int TAPS = DW ? 9 : 1;
    const bool gated = SIMPLE && !DW && a.gate != nullptr;
    typename std::conditional<DW, PwDwRow, PwRow<MODE>>::type row[A_PASSES];
    pw_unroll<A_PASSES>([&](auto P) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        row[p].init(a, m0 + lr + p * PW_RPP);
        if constexpr (SIMPLE && !DW)
            if (!gated) row[p].grow = a.wt;  // ungated: the gate load becomes a (cached, ignored) weight quad
    });
    extern __shared__ __attribute__((aligned(16))) float dwl[];  // DW: [9][kp] weights | [kp] scale | [kp] shift
    if constexpr (DW) {
        for (int i = tid * 4; i < 11 * kp; i += 256 * 4) {
            const float* src = i < 9 * kp ? a.dw_w + i : (i < 10 * kp ? a.dw_scale + (i - 9 * kp) : a.dw_shift + (i - 10 * kp));
            *reinterpret_cast<float4*>(dwl + i) = *reinterpret_cast<const float4*>(src);
        }
        __syncthreads();
    }
    const float* brow[B_PASSES];
    pw_unroll<B_PASSES>([&](auto P) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        const int n = n0 + lr + p * PW_RPP;
        brow[p] = a.wt + (size_t)(n < a.N ? n : 0) * kp;  // rows beyond N feed couts that are never stored
    });

    const int g = lane >> 4, li = lane & 15;
    f32x4 acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The k loop, instantiated with and without pooled-source support: only the rare pooled gathers pay for the
    // branches (and the vmcnt(0) waits they force) around the extra taps.
    auto k_loop = [&](auto pools_tag) __attribute__((always_inline)) {
        constexpr bool POOLS = decltype(pools_tag)::value;
        // fetch() only ISSUES loads (raw values + the pixel's gate quad), all of them unconditional (PwRow::issue):
        // masking and the gate multiply happen in stage(), one or two chunks later, right before the LDS store.
        // Touching the loaded registers inside fetch() would put the s_waitcnt - a full L2/HBM round trip - in
        // front of the MFMAs of every k chunk.
        struct Regs {
            float4 ra[A_PASSES][TAPS], rg[A_PASSES], rb[B_PASSES];
            int cv[A_PASSES];  // valid channels in the fetched quad (<= 0: none)
        };
        auto fetch = [&](int k0, Regs& R) __attribute__((always_inline)) {
            const int kraw = k0 + kq * 4;
            const int k = kraw < kp ? kraw : kp - 4;
            pw_unroll<A_PASSES>([&](auto P) __attribute__((always_inline)) {
                constexpr int p = decltype(P)::value;
                if constexpr (DW) row[p].issue(a, kraw, kp, R.ra[p], R.cv[p]);
                else row[p].template issue<POOLS>(a, kraw, kp, R.ra[p][0], R.rg[p], R.cv[p]);
            });
            pw_unroll<B_PASSES>([&](auto P) __attribute__((always_inline)) {
                constexpr int p = decltype(P)::value;
                R.rb[p] = *reinterpret_cast<const float4*>(brow[p] + k);
            });
        };

        // One k chunk: registers -> LDS, barrier, refill the register set with the chunk DEPTH ahead, fragments + MFMA,
        // barrier.  With DEPTH 2 two chunks of global loads are in flight per wave; the loop body is two steps on
        // alternating register sets and every fetch is unconditional, so the compiler counts the outstanding loads
        // exactly and a step waits only for ITS set.  A dead step (odd chunk count) stages zeros and skips the MFMAs.
        constexpr int DEPTH = (!DW && PT * CT <= PW_PF2_MAX_TILES) ? 2 : 1;  // (nine tap quads per activation quad: one set only)
        auto step = [&](int k0, Regs& R, bool live) __attribute__((always_inline)) {
            pw_unroll<A_PASSES>([&](auto P) __attribute__((always_inline)) {
                constexpr int p = decltype(P)::value;
                float4 v;
                if constexpr (DW) {
                    const int kraw = k0 + kq * 4;
                    v = row[p].finish(a, R.ra[p], R.cv[p], dwl, kp, kraw < kp ? kraw : kp - 4);
                } else {
                    v = gated ? pw_finish<2>(R.ra[p][0], R.rg[p], R.cv[p]) : pw_finish<1>(R.ra[p][0], R.rg[p], R.cv[p]);
                }
                *reinterpret_cast<float4*>(As + (lr + p * PW_RPP) * PW_LDS_LD + kq * 4) = v;
            });
            pw_unroll<B_PASSES>([&](auto P) __attribute__((always_inline)) {
                constexpr int p = decltype(P)::value;
                const float4 v = R.rb[p];
                if ((p + 1) * PW_RPP <= BN || lr + p * PW_RPP < BN)  // only a partial last pass tests the lane
                    *reinterpret_cast<float4*>(Bs + (lr + p * PW_RPP) * PW_LDS_LD + kq * 4) = v;
            });
            __syncthreads();
            fetch(k0 + DEPTH * PW_BK, R);
            // fragments + MFMA, 16 k per sub-step (sub-steps wholly beyond kp are skipped: uniform)
            if (live) {
#pragma unroll
                for (int kk = 0; kk < PW_BK; kk += 16) {
                    if (k0 + kk >= kp) break;
                    f32x4 wf[CT], xf[PT];
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        wf[c] = *reinterpret_cast<const f32x4*>(Bs + ((wn * CT + c) * 16 + li) * PW_LDS_LD + kk + g * 4);
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        xf[p] = *reinterpret_cast<const f32x4*>(As + ((wm * PT + p) * 16 + li) * PW_LDS_LD + kk + g * 4);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int p = 0; p < PT; ++p)
                                acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][s], xf[p][s], acc[c][p], 0, 0, 0);
                }
            }
            __syncthreads();
        };
        Regs R0;
        fetch(0, R0);
        if constexpr (DEPTH == 2) {
            Regs R1;
            __builtin_amdgcn_sched_barrier(0);  // R0's loads must be issued first: the loop waits for them by COUNT
            fetch(PW_BK, R1);
            __builtin_amdgcn_sched_barrier(0);
            for (int k0 = 0; k0 < kp; k0 += 2 * PW_BK) {
                step(k0, R0, true);
                step(k0 + PW_BK, R1, k0 + PW_BK < kp);
            }
        } else {
            for (int k0 = 0; k0 < kp; k0 += PW_BK) step(k0, R0, true);
        }
    };
    bool pooled = false;
    if (!SIMPLE) {
#pragma unroll
        for (int i = 0; i < YR_MAX_SRC; ++i)
            pooled |= a.S.s[i].xform == YR_X_MAXPOOL2 || a.S.s[i].xform == YR_X_MAXPOOL4;
    }
    if (pooled) k_loop(std::true_type{});
    else k_loop(std::false_type{});

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
            pw_finish_quad(a, acc[c][p], sc, sh, m0 + (wm * PT + p) * 16 + li, n0 + nl, li, vec_out, vec_res, vec_pre);
    }
}


template <int PT, int CT, int WM, int WN>
static int launch_cfg(const PwArgs& a, hipStream_t s) {
    constexpr int BM = 16 * PT * WM, BN = 16 * CT * WN;
    dim3 grid((unsigned)((a.M + BM - 1) / BM) * (unsigned)((a.N + BN - 1) / BN));
    if (a.dw_w != nullptr) {  // depthwise-folded source: built for the 64-row shapes with 3..8 cout tiles
        if constexpr (PT == 1 && WM == 4 && CT >= 3) {
            static char dnm[48];
            static const int dnm_len = snprintf(dnm, sizeof(dnm), "pw_kernel<%d,%d,%d,%d,1,1>", PT, CT, WM, WN);   // (spelled like the symbol: SIMPLE, DW)
            (void)dnm_len;
            yr_note_kernel(dnm);
            hipLaunchKernelGGL((pw_kernel<PT, CT, WM, WN, true, true>), grid, dim3(256), (size_t)11 * a.S.kp * sizeof(float), s, a);
            YR_LAUNCH_CHECK();
            return YR_OK;
        } else {
            yr_set_error("pointwise: no depthwise-folded kernel for tile shape %dx%d", BM, BN);
            return YR_ERR_ARG;
        }
    }
    const bool simple = a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY;
    static char nm[2][48];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pw_kernel<%d,%d,%d,%d,0,0>", PT, CT, WM, WN) +
                              snprintf(nm[1], sizeof(nm[1]), "pw_kernel<%d,%d,%d,%d,1,0>", PT, CT, WM, WN);
    (void)nm_len;
    yr_note_kernel(nm[simple ? 1 : 0]);
    if (simple) hipLaunchKernelGGL((pw_kernel<PT, CT, WM, WN, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pw_kernel<PT, CT, WM, WN, false>), grid, dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

int yr_pw_launch_lds(int shape, const PwArgs& a, hipStream_t s) {
    if (a.dw_w != nullptr && (shape < 9 || shape > 13)) {  // any other request: the narrowest built shape that covers N
        shape = a.N <= 48 ? 9 : a.N <= 64 ? 10 : a.N <= 80 ? 11 : a.N <= 96 ? 12 : 13;
    }
    switch (shape) {
        case 0: return launch_cfg<4, 1, 4, 1>(a, s);
        case 1: return launch_cfg<2, 2, 4, 1>(a, s);
        case 2: return launch_cfg<2, 3, 4, 1>(a, s);
        case 3: return launch_cfg<4, 2, 2, 2>(a, s);
        case 4: return launch_cfg<2, 5, 4, 1>(a, s);
        case 5: return launch_cfg<4, 3, 2, 2>(a, s);
        case 6: return launch_cfg<4, 4, 2, 2>(a, s);
        case 7: return launch_cfg<1, 1, 4, 1>(a, s);
        case 8: return launch_cfg<1, 2, 4, 1>(a, s);
        case 9: return launch_cfg<1, 3, 4, 1>(a, s);
        case 10: return launch_cfg<1, 4, 4, 1>(a, s);
        case 11: return launch_cfg<1, 5, 4, 1>(a, s);
        case 12: return launch_cfg<1, 6, 4, 1>(a, s);
        case 13: return launch_cfg<1, 8, 4, 1>(a, s);
        default: yr_set_error("pointwise: LDS shape %d out of range", shape); return YR_ERR_ARG;
    }
}
