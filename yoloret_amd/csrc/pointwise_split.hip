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
#include <type_traits>

#include "pw_common.h"

#define PWS_BK 32
#ifndef PWS_PF2_MAX_TILES
#define PWS_PF2_MAX_TILES 4      // tiles (PT * CT) per wave up to which TWO k chunks are prefetched (8: the second register set costs a wave per SIMD - pointwise family 0.81 -> 0.89 ms on c2)
#endif
#define PWS_KQ (PWS_BK / 4)        // float4 quads per staged row
#define PWS_RPP (256 / PWS_KQ)     // rows loaded per pass of the 256 threads
#define PWS_LD (PWS_BK + 8)        // halves per LDS row: 80 bytes, an odd number of 16-byte slots

typedef _Float16 pws_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 pws_h8 __attribute__((ext_vector_type(8)));
typedef unsigned pws_u4 __attribute__((ext_vector_type(4)));
typedef unsigned pws_u2 __attribute__((ext_vector_type(2)));
typedef float pws_f2 __attribute__((ext_vector_type(2)));

// four float32 values -> four halves of the h plane and four of the m plane at the same position
__device__ __forceinline__ void pws_store(_Float16* ph, _Float16* pm, int off, const float4 v) {
    const pws_f2 a = (pws_f2){v.x, v.y}, b = (pws_f2){v.z, v.w};
    const pws_h2 ha = __builtin_convertvector(a, pws_h2), hb = __builtin_convertvector(b, pws_h2);
    const pws_h2 ma = __builtin_convertvector((a - __builtin_convertvector(ha, pws_f2)) * 2048.0f, pws_h2);
    const pws_h2 mb = __builtin_convertvector((b - __builtin_convertvector(hb, pws_f2)) * 2048.0f, pws_h2);
    *reinterpret_cast<pws_u2*>(ph + off) = (pws_u2){__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
    *reinterpret_cast<pws_u2*>(pm + off) = (pws_u2){__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb)};
}
__device__ __forceinline__ f32x4 pws_mfma(pws_u4 a, pws_u4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pws_h8, a), __builtin_bit_cast(pws_h8, b), c, 0, 0, 0);
}

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
    _Float16* Ah = lds;                       // [BM][PWS_LD] activations, h plane
    _Float16* Am = lds + BM * PWS_LD;         //                           m plane = f16((x - h) 2^11)
    _Float16* Bh = lds + 2 * BM * PWS_LD;     // [BN][PWS_LD] weights
    _Float16* Bm = lds + (2 * BM + BN) * PWS_LD;
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
    const int lr = tid / PWS_KQ, kq = tid % PWS_KQ;
    constexpr int MODE = SIMPLE ? 2 : 0;
    constexpr int TAPS = DW ? 9 : 1;
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
                row[p].template issue<POOLS>(a, kraw, kp, R.ra[p][0], R.rg[p], R.cv[p]);
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
        constexpr int DEPTH = PT * CT <= PWS_PF2_MAX_TILES ? 2 : 1;   // (two register sets of fetched chunks in flight)
        auto step = [&](int k0, Regs& R, bool live) __attribute__((always_inline)) {
            pw_unroll<A_PASSES>([&](auto P) __attribute__((always_inline)) {
                constexpr int p = decltype(P)::value;
                const float4 v = gated ? pw_finish<2>(R.ra[p][0], R.rg[p], R.cv[p]) : pw_finish<1>(R.ra[p][0], R.rg[p], R.cv[p]);
                pws_store(Ah, Am, (lr + p * PWS_RPP) * PWS_LD + kq * 4, v);
            });
            pw_unroll<B_PASSES>([&](auto P) __attribute__((always_inline)) {
                constexpr int p = decltype(P)::value;
                const float4 v = R.rb[p];
                if ((p + 1) * PWS_RPP <= BN || lr + p * PWS_RPP < BN)  // only a partial last pass tests the lane
                    pws_store(Bh, Bm, (lr + p * PWS_RPP) * PWS_LD + kq * 4, v);
            });
            __syncthreads();
            fetch(k0 + DEPTH * PWS_BK, R);
            // fragments (8 halves of each plane per lane: k = 8 g .. 8 g + 7 of the chunk) + three MFMAs per tile pair
            if (live) {
                pws_u4 wh[CT], wm_[CT], xh[PT], xm[PT];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    wh[c] = *reinterpret_cast<const pws_u4*>(Bh + ((wn * CT + c) * 16 + li) * PWS_LD + g * 8);
                    wm_[c] = *reinterpret_cast<const pws_u4*>(Bm + ((wn * CT + c) * 16 + li) * PWS_LD + g * 8);
                }
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    xh[p] = *reinterpret_cast<const pws_u4*>(Ah + ((wm * PT + p) * 16 + li) * PWS_LD + g * 8);
                    xm[p] = *reinterpret_cast<const pws_u4*>(Am + ((wm * PT + p) * 16 + li) * PWS_LD + g * 8);
                }
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p) acc[c][p] = pws_mfma(wh[c], xh[p], acc[c][p]);
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p) ac1[c][p] = pws_mfma(wh[c], xm[p], ac1[c][p]);
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int p = 0; p < PT; ++p) ac1[c][p] = pws_mfma(wm_[c], xh[p], ac1[c][p]);
            }
            __syncthreads();
        };
        Regs R0;
        fetch(0, R0);
        if constexpr (DEPTH == 2) {
            Regs R1;
            __builtin_amdgcn_sched_barrier(0);  // R0's loads must be issued first: the loop waits for them by COUNT
            fetch(PWS_BK, R1);
            __builtin_amdgcn_sched_barrier(0);
            for (int k0 = 0; k0 < kp; k0 += 2 * PWS_BK) {
                step(k0, R0, true);
                step(k0 + PWS_BK, R1, k0 + PWS_BK < kp);
            }
        } else {
            for (int k0 = 0; k0 < kp; k0 += PWS_BK) step(k0, R0, true);
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
            pw_finish_quad(a, ac1[c][p] * 0.00048828125f + acc[c][p], sc, sh, m0 + (wm * PT + p) * 16 + li, n0 + nl, li, vec_out, vec_res, vec_pre);
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
