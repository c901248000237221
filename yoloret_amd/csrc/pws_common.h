// The k loop of the SPLIT pointwise GEMM (float32 operands as two float16 planes on v_mfma_f32_16x16x32_f16; see
// pointwise_split.hip for the arithmetic), shared by pws_kernel and the head-block kernel (headblock.hip): global fetch through
// PwRow (gathers, concat, up-sampling, pooling, SE gate) -> planes cut on the way into LDS -> fragments -> three MFMAs per tile.
#pragma once
#include <type_traits>

#include "pw_common.h"

#define PWS_BK 32
#ifndef PWS_PF2_MAX_TILES
#define PWS_PF2_MAX_TILES 4      // tiles (PT * CT) per wave up to which TWO k chunks are prefetched (8: the second register set costs a wave per SIMD - pointwise family 0.81 -> 0.89 ms on c2)
#endif
#define PWS_KQ (PWS_BK / 4)        // float4 quads per staged row
#define PWS_LD (PWS_BK + 8)        // halves per LDS row: 80 bytes, an odd number of 16-byte slots

typedef _Float16 pws_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 pws_h8 __attribute__((ext_vector_type(8)));
typedef unsigned pws_u4 __attribute__((ext_vector_type(4)));
typedef unsigned pws_u2 __attribute__((ext_vector_type(2)));
typedef float pws_f2 __attribute__((ext_vector_type(2)));

// four float32 values -> four halves of the h plane and four of the m plane at the same position
__device__ __forceinline__ void pws_store(_Float16* ph, _Float16* pm, int off, const float4 v) {
    unsigned ha, ma, hb, mb;
    yr_cut2(v.x, v.y, ha, ma);     // (4 operations per pair: yr_common.h)
    yr_cut2(v.z, v.w, hb, mb);
    *reinterpret_cast<pws_u2*>(ph + off) = (pws_u2){ha, hb};
    *reinterpret_cast<pws_u2*>(pm + off) = (pws_u2){ma, mb};
}
__device__ __forceinline__ f32x4 pws_mfma(pws_u4 a, pws_u4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pws_h8, a), __builtin_bit_cast(pws_h8, b), c, 0, 0, 0);
}

// The whole k loop of one workgroup tile.  NTH threads = WM x WN waves, each PT x CT MFMA tiles; row[] / brow[]: the activation
// rows and weight rows this thread fetches (pass p: tile row lr + p * (NTH / 8)); lds: 2 * (BM + BN) * PWS_LD halves.
// acc / ac1: h h' | h m' + m h' (the caller joins them with 2^-11).  Ends with a barrier: the LDS may be reused at once.
template <int NTH, int PT, int CT, int WM, int WN, int MODE, int A_PASSES, int B_PASSES>
__device__ __forceinline__ void pws_k_loop(const PwArgs& a, PwRow<MODE> (&row)[A_PASSES], const float* (&brow)[B_PASSES], const bool gated,
                                           _Float16* lds, f32x4 (&acc)[CT][PT], f32x4 (&ac1)[CT][PT]) {
    constexpr int RPP = NTH / PWS_KQ;     // rows loaded per pass of the NTH threads
    constexpr int BM = 16 * PT * WM, BN = 16 * CT * WN;
    static_assert(A_PASSES * RPP == BM && B_PASSES == (BN + RPP - 1) / RPP, "loader passes");
    _Float16* Ah = lds;                       // [BM][PWS_LD] activations, h plane
    _Float16* Am = lds + BM * PWS_LD;         //                           m plane = f16((x - h) 2^11)
    _Float16* Bh = lds + 2 * BM * PWS_LD;     // [BN][PWS_LD] weights
    _Float16* Bm = lds + (2 * BM + BN) * PWS_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int lr = tid / PWS_KQ, kq = tid % PWS_KQ;
    const int g = lane >> 4, li = lane & 15;
    const int kp = a.S.kp;

    // The k loop, instantiated with and without pooled-source support: only the rare pooled gathers pay for the
    // branches (and the vmcnt(0) waits they force) around the extra taps.
    auto k_loop = [&](auto pools_tag) __attribute__((always_inline)) {
        constexpr bool POOLS = decltype(pools_tag)::value;
        // fetch() only ISSUES loads (raw values + the pixel's gate quad), all of them unconditional (PwRow::issue):
        // masking and the gate multiply happen in stage(), one or two chunks later, right before the LDS store.
        // Touching the loaded registers inside fetch() would put the s_waitcnt - a full L2/HBM round trip - in
        // front of the MFMAs of every k chunk.
        struct Regs {
            float4 ra[A_PASSES][1], rg[A_PASSES], rb[B_PASSES];
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
                pws_store(Ah, Am, (lr + p * RPP) * PWS_LD + kq * 4, v);
            });
            pw_unroll<B_PASSES>([&](auto P) __attribute__((always_inline)) {
                constexpr int p = decltype(P)::value;
                const float4 v = R.rb[p];
                if ((p + 1) * RPP <= BN || lr + p * RPP < BN)  // only a partial last pass tests the lane
                    pws_store(Bh, Bm, (lr + p * RPP) * PWS_LD + kq * 4, v);
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
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < YR_MAX_SRC; ++i)
            pooled |= a.S.s[i].xform == YR_X_MAXPOOL2 || a.S.s[i].xform == YR_X_MAXPOOL4;
    }
    if (pooled) k_loop(std::true_type{});
    else k_loop(std::false_type{});
}
