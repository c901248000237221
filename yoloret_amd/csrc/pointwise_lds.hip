// Pointwise (1x1) convolution, LDS-staged MFMA kernel (see pointwise.hip for the GEMM view, the direct
// variant and the dispatcher).  A 16-wide k chunk is staged in LDS by coalesced loads (4 adjacent lanes = 64
// bytes of a row); lane group g reads k = k0+4g..4g+3 as one ds_read_b128 and uses component s in MFMA step s.
// The next chunk's global loads are issued before the current chunk's MFMAs (register double buffering).
#include "pw_common.h"

#ifndef PW_BK
#define PW_BK 16                 // k depth staged per barrier pair (multiple of 16; measured: 32 is 6 % and 64 is 13 % slower end to end)
#endif
#ifndef PW_PF2_MAX_TILES
#define PW_PF2_MAX_TILES 0       // tiles (PT*CT) per wave up to which TWO k chunks are prefetched (measured: never pays here)
#endif
#define PW_KQ (PW_BK / 4)        // float4 quads per staged row
#define PW_RPP (256 / PW_KQ)     // rows loaded per pass of the 256 threads
#define PW_LDS_LD (PW_BK + 4)    // padded row stride (floats): an odd number of 16-byte slots

// PT/CT: 16-wide pixel / cout MFMA tiles per wave; WM x WN waves (WM*WN == 4).
// SIMPLE: one identity source (optionally SE-gated) -> a pixel's channels are one contiguous row (the
// common case: every backbone expand/project and most head convs); otherwise the generic gather
// through per-source row pointers (upsample / maxpool / concat folded into the loads).
template <int PT, int CT, int WM, int WN, bool SIMPLE>
__global__ __launch_bounds__(256) void pw_kernel(PwArgs a) {
    constexpr int BM = 16 * PT * WM;
    constexpr int BN = 16 * CT * WN;
    constexpr int A_PASSES = BM / PW_RPP;
    constexpr int B_PASSES = (BN + PW_RPP - 1) / PW_RPP;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * PW_LDS_LD];
    float* As = lds;                   // [BM][PW_LDS_LD] activations
    float* Bs = lds + BM * PW_LDS_LD;  // [BN][PW_LDS_LD] weights

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

    // loader mapping: quad kq of row lr (+64 per pass)
    const int lr = tid / PW_KQ, kq = tid % PW_KQ;
    bool pv[A_PASSES];
    const float* arow[A_PASSES];               // SIMPLE: the pixel's row
    const float* grow[A_PASSES];               // SE gate row of the pixel's image (or null)
    const float* srow[A_PASSES][YR_MAX_SRC];   // generic: per-source row pointer of the pixel (xform folded in)
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
        const int m = m0 + lr + p * PW_RPP;
        pv[p] = m < a.M;
        const int mm = pw_pixel_of_row(a, pv[p] ? m : 0);
        const int hw = a.H * a.W;
        const int b = mm / hw;
        grow[p] = a.gate ? a.gate + (size_t)b * a.gate_ld : nullptr;
        if (SIMPLE) {
            arow[p] = a.S.s[0].ptr + (size_t)mm * a.S.s[0].ld;
        } else {
            arow[p] = nullptr;
            const int rem = mm - b * hw;
            const int y = rem / a.W, x = rem - y * a.W;
#pragma unroll
            for (int si = 0; si < YR_MAX_SRC; ++si) {
                const DSrc& d = a.S.s[si];
                int sy = y, sx = x;
                if (d.xform == YR_X_UP2) { sy = y >> 1; sx = x >> 1; }
                else if (d.xform == YR_X_MAXPOOL2) { sy = y * 2; sx = x * 2; }
                else if (d.xform == YR_X_MAXPOOL4) { sy = y * 4; sx = x * 4; }
                srow[p][si] = d.ptr + ((size_t)(b * d.h + sy) * d.w + sx) * d.ld;
            }
        }
    }
    const float* brow[B_PASSES];
    bool bvld[B_PASSES];
#pragma unroll
    for (int p = 0; p < B_PASSES; ++p) {
        const int n = n0 + lr + p * PW_RPP;
        bvld[p] = (lr + p * PW_RPP < BN) && n < a.N;
        brow[p] = a.wt + (size_t)(bvld[p] ? n : 0) * kp;
    }

    // fetch() only ISSUES loads (raw values + the pixel's gate quad); masking and the gate multiply happen in
    // stage(), one iteration later, right before the LDS store.  Touching the loaded registers inside fetch()
    // would put the s_waitcnt - a full L2/HBM round trip - in front of the MFMAs of every k chunk.
    struct Regs {
        float4 ra[A_PASSES], rg[A_PASSES], rb[B_PASSES];
        int cv[A_PASSES];  // valid channels in the fetched quad (<= 0: none)
    };
    auto fetch = [&](int k0, Regs& R) {
        const int k = k0 + kq * 4;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 gt = make_float4(1.f, 1.f, 1.f, 1.f);
            int cvalid = 0;
            if (pv[p] && k < kp) {
                if (SIMPLE) {
                    v = *reinterpret_cast<const float4*>(arow[p] + k);
                    cvalid = a.S.s[0].c - k;
                } else {
                    // segment of this quad (kbase of unused segments is huge), then a pre-offset row pointer
                    int si = 0;
#pragma unroll
                    for (int i = 1; i < YR_MAX_SRC; ++i)
                        if (k >= a.S.s[i].kbase) si = i;
                    const float* rp = srow[p][0];
                    int kb = a.S.s[0].kbase, cc = a.S.s[0].c, xf = a.S.s[0].xform, sw = a.S.s[0].w, sld = a.S.s[0].ld;
#pragma unroll
                    for (int i = 1; i < YR_MAX_SRC; ++i)
                        if (si == i) { rp = srow[p][i]; kb = a.S.s[i].kbase; cc = a.S.s[i].c; xf = a.S.s[i].xform; sw = a.S.s[i].w; sld = a.S.s[i].ld; }
                    rp += k - kb;
                    v = *reinterpret_cast<const float4*>(rp);
                    if (xf == YR_X_MAXPOOL2) {  // pooled sources are reduced here (the only path that waits in fetch):
                        // the three other taps are issued together - ONE round trip, not one per tap
                        const float4 v1 = *reinterpret_cast<const float4*>(rp + sld);
                        const float4 v2 = *reinterpret_cast<const float4*>(rp + (size_t)sw * sld);
                        const float4 v3 = *reinterpret_cast<const float4*>(rp + ((size_t)sw + 1) * sld);
                        v = yr_max4(yr_max4(v, v1), yr_max4(v2, v3));
                    } else if (xf == YR_X_MAXPOOL4) {
                        for (int dy = 0; dy < 4; ++dy)
                            for (int dx = 0; dx < 4; ++dx)
                                v = yr_max4(v, *reinterpret_cast<const float4*>(rp + ((size_t)dy * sw + dx) * sld));
                    }
                    cvalid = cc - (k - kb);
                }
                if (grow[p] != nullptr) gt = *reinterpret_cast<const float4*>(grow[p] + k);
            }
            R.ra[p] = v; R.rg[p] = gt; R.cv[p] = cvalid;
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bvld[p] && k < kp) v = *reinterpret_cast<const float4*>(brow[p] + k);
            R.rb[p] = v;
        }
    };
    // the fetched quad with pad lanes zeroed (the source's pad lanes and the gate's may hold anything) and gated
    auto staged = [&](const Regs& R, int p) {
        float4 v = R.ra[p];
        const float4 gt = R.rg[p];
        const int cvalid = R.cv[p];
        v.x = cvalid > 0 ? v.x * gt.x : 0.f;
        v.y = cvalid > 1 ? v.y * gt.y : 0.f;
        v.z = cvalid > 2 ? v.z * gt.z : 0.f;
        v.w = cvalid > 3 ? v.w * gt.w : 0.f;
        return v;
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15;
    // One k chunk: registers -> LDS, barrier, refill the register set with the chunk DEPTH ahead, fragments + MFMA,
    // barrier.  Small tiles (<= 4 MFMA tiles per wave: <= 512 matrix cycles per chunk) keep two chunks of global
    // loads in flight, larger ones one (their MFMA phase already covers an L2 round trip; the extra registers cost
    // occupancy: measured).
    constexpr int DEPTH = (PT * CT <= PW_PF2_MAX_TILES) ? 2 : 1;
    auto step = [&](int k0, Regs& R) {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p)
            *reinterpret_cast<float4*>(As + (lr + p * PW_RPP) * PW_LDS_LD + kq * 4) = staged(R, p);
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p)
            if (lr + p * PW_RPP < BN) *reinterpret_cast<float4*>(Bs + (lr + p * PW_RPP) * PW_LDS_LD + kq * 4) = R.rb[p];
        __syncthreads();
        if (k0 + DEPTH * PW_BK < kp) fetch(k0 + DEPTH * PW_BK, R);
        // fragments + MFMA, 16 k per sub-step (sub-steps wholly beyond kp are skipped: uniform)
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
        __syncthreads();
    };
    Regs R0;
    fetch(0, R0);
    if constexpr (DEPTH == 2) {
        Regs R1;
        fetch(PW_BK, R1);  // beyond kp: zeros, never staged
        for (int k0 = 0; k0 < kp; k0 += 2 * PW_BK) {
            step(k0, R0);
            if (k0 + PW_BK < kp) step(k0 + PW_BK, R1);
        }
    } else {
        for (int k0 = 0; k0 < kp; k0 += PW_BK) step(k0, R0);
    }

    // ---- epilogue: BN scale/shift, activation, residual, store (4 consecutive couts per lane)
    const bool vec_out = (a.out_ld & 3) == 0;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = n0 + (wn * CT + c) * 16 + g * 4;
        if (n >= a.N) continue;
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < a.N) {
                if (a.scale) sc[r] = a.scale[n + r];
                if (a.shift) sh[r] = a.shift[n + r];
            }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = m0 + (wm * PT + p) * 16 + li;
            if (m >= a.M) continue;
            float v[4], pa[4];
            if (a.pre) {  // uniform: the low-resolution share of a hoisted concat conv joins the accumulator before BN
                pw_pre_addend(a, m, n, pa);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[c][p][r] += pa[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = yr_apply_act(__builtin_fmaf(acc[c][p][r], sc[r], sh[r]), a.act);
            if (a.res) {
                const float* rp = a.res + (size_t)m * a.res_ld + n;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < a.N) v[r] += rp[r];
            }
            int orow = m;
            if (a.pool) {  // uniform: MaxPooling2D(2) across the 4 adjacent lanes of a window (see pw_common.h)
                pw_pool4(v);
                if (li & 3) continue;
                orow = m >> 2;
            }
            float* op = a.out + (size_t)orow * a.out_ld + n;
            if (vec_out && n + 3 < a.N) {
                *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (n + 3 < a.N) {
                *reinterpret_cast<f32x4u*>(op) = (f32x4u){v[0], v[1], v[2], v[3]};  // dense rows: dword-aligned 16-byte store
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < a.N) op[r] = v[r];
            }
        }
    }
}


template <int PT, int CT, int WM, int WN>
static int launch_cfg(const PwArgs& a, hipStream_t s) {
    constexpr int BM = 16 * PT * WM, BN = 16 * CT * WN;
    dim3 grid((unsigned)((a.M + BM - 1) / BM) * (unsigned)((a.N + BN - 1) / BN));
    const bool simple = a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY;
    static char nm[2][48];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pw_kernel<%d,%d,%d,%d,0>", PT, CT, WM, WN) +
                              snprintf(nm[1], sizeof(nm[1]), "pw_kernel<%d,%d,%d,%d,1>", PT, CT, WM, WN);
    (void)nm_len;
    yr_note_kernel(nm[simple ? 1 : 0]);
    if (simple) hipLaunchKernelGGL((pw_kernel<PT, CT, WM, WN, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pw_kernel<PT, CT, WM, WN, false>), grid, dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

int yr_pw_launch_lds(int shape, const PwArgs& a, hipStream_t s) {
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
