// Device helpers shared by the register-chained float32 kernels (mbr.hip: YR_OP_MBR / YR_OP_MBE; headwalk.hip: YR_OP_HEAD's
// walking form): DPP depthwise tap rows on MFMA result registers, buffer descriptors, the float16-plane split of the SPLIT form.
#pragma once
#include "yr_common.h"
#include <type_traits>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define MBR_TAB 176   // floats of one tile's LDS table

__device__ __forceinline__ float mbr_shr1(float v) {   // lane l <- lane l - 1 of the same 16-lane row, 0 at the row start
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float mbr_shl1(float v) {   // lane l <- lane l + 1, 0 at the row end
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));
}

// One tap ROW of the 3x3 depthwise conv for the lane's 4 channels: acc[i] += e[i] * w1[i] + shr(e[i]) * w0[i] + shl(e[i]) * w2[i].
// The centre tap needs no shift and runs as two v_pk_fma_f32 (round 5: 10 instead of 12 VALU slots per row - and four operands
// fewer in the asm statement, which ended the scratch spills of the three-wave block kernels).  For the outer taps the DPP shift
// rides on the multiply-add's first operand (v_fmac_f32_dpp: no v_mov_dpp, no extra register) and the four
// channels' chains are interleaved tap-major, so a dependent instruction sits four slots behind its producer.  hipcc 7.2 does
// not fold update_dpp into the fma (left to itself: 6 v_mov_b32_dpp + hazard nops per channel, one chain after the other).
// s_nop 1: a VALU write of e[] must be two wait states ahead of a DPP read (the hazard recogniser does not look into asm).
// (A packed form - v_pk_fma_f32 on row_shr / row_shl copies made once per row, scatter order - needs 26 instead of 36 VALU
// slots per tile row, but its register tuples made hipcc 7.2 spill 100-600 bytes per lane on the stride-1 kernels: measured
// slower everywhere it spilled, equal elsewhere.)
#define MBR_DPP(ctl) " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void mbr_dw_row(v4f& acc, const v4f e, const v4f w0, const v4f w1, const v4f w2) {
    acc = __builtin_elementwise_fma(e, w1, acc);     // the centre tap needs no shift: two v_pk_fma_f32 instead of four v_fmac_f32
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %4, %8" MBR_DPP("row_shr:1")
        "v_fmac_f32_dpp %1, %5, %9" MBR_DPP("row_shr:1")
        "v_fmac_f32_dpp %2, %6, %10" MBR_DPP("row_shr:1")
        "v_fmac_f32_dpp %3, %7, %11" MBR_DPP("row_shr:1")
        "v_fmac_f32_dpp %0, %4, %12" MBR_DPP("row_shl:1")
        "v_fmac_f32_dpp %1, %5, %13" MBR_DPP("row_shl:1")
        "v_fmac_f32_dpp %2, %6, %14" MBR_DPP("row_shl:1")
        "v_fmac_f32_dpp %3, %7, %15" MBR_DPP("row_shl:1")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]),
          "v"(w2[0]), "v"(w2[1]), "v"(w2[2]), "v"(w2[3]));
    acc = (v4f){a0, a1, a2, a3};
}

// STRIDE 2, PAIRED OUTPUT ROWS.  With lane = input column, a stride-2 strip has its 7 outputs in every other lane and the projection
// MFMAs run at 7 useful columns of 16.  The expand conv does not care which pixel sits in which lane, so a stride-2 strip loads the
// EVEN input columns E_0..E_7 into lanes 0..7 of a DPP row and the ODD ones O_0..O_7 into lanes 8..15; output column j needs
// E_j, O_j, E_j+1:
//   * an even output row computes in lanes 0..6 (own lane, row_shl:8, row_shl:1) and writes banks 0-1 only (bank_mask:0x3),
//   * the next (odd) output row computes in lanes 8..14 (row_shr:8, own lane, row_shr:7) and writes banks 2-3 only,
// into the SAME accumulator registers: one clamp and ONE set of projection MFMAs (14 useful columns of 16) per pair of output rows.
#define MBR_DPPM(ctl, bank) " " ctl " row_mask:0xf bank_mask:" bank " bound_ctrl:1\n\t"
#define MBR_DW2(c0, c1, c2, bank)                                                                                                    \
    asm("s_nop 1\n\t"                                                                                                                \
        "v_fmac_f32_dpp %0, %4, %8" MBR_DPPM(c0, bank) "v_fmac_f32_dpp %1, %5, %9" MBR_DPPM(c0, bank)                               \
        "v_fmac_f32_dpp %2, %6, %10" MBR_DPPM(c0, bank) "v_fmac_f32_dpp %3, %7, %11" MBR_DPPM(c0, bank)                             \
        "v_fmac_f32_dpp %0, %4, %12" MBR_DPPM(c1, bank) "v_fmac_f32_dpp %1, %5, %13" MBR_DPPM(c1, bank)                             \
        "v_fmac_f32_dpp %2, %6, %14" MBR_DPPM(c1, bank) "v_fmac_f32_dpp %3, %7, %15" MBR_DPPM(c1, bank)                             \
        "v_fmac_f32_dpp %0, %4, %16" MBR_DPPM(c2, bank) "v_fmac_f32_dpp %1, %5, %17" MBR_DPPM(c2, bank)                             \
        "v_fmac_f32_dpp %2, %6, %18" MBR_DPPM(c2, bank) "v_fmac_f32_dpp %3, %7, %19" MBR_DPPM(c2, bank)                             \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                                                                     \
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]),                                \
          "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]), "v"(w2[0]), "v"(w2[1]), "v"(w2[2]), "v"(w2[3]))
template <bool ODD>
__device__ __forceinline__ void mbr_dw_row2(v4f& acc, const v4f e, const v4f w0, const v4f w1, const v4f w2) {
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    if constexpr (!ODD) MBR_DW2("quad_perm:[0,1,2,3]", "row_shl:8", "row_shl:1", "0x3");
    else MBR_DW2("row_shr:8", "quad_perm:[0,1,2,3]", "row_shr:7", "0xc");
    acc = (v4f){a0, a1, a2, a3};
}

typedef __amdgpu_buffer_rsrc_t mbr_rsrc;
__device__ __forceinline__ mbr_rsrc mbr_make_rsrc(const void* base, unsigned bytes) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#define MBR_DEAD 0x7f000000u   // a byte offset beyond every descriptor's num_records: the load returns zeros, the store is dropped

// ---- SPLIT form (SP, round 4): the two 1x1 convolutions on the 16-bit matrix pipe with float32-grade operands.  On gfx950 the
// float32 MFMA runs on the VALU's FMA lanes (its cycles ADD to the depthwise stage's), v_mfma_f32_16x16x32_f16 has its own pipe and
// 8 x the rate.  Every float32 operand is cut into two float16 planes, x = h + 2^-11 m with h = f16(x), m = f16((x - h) 2^11)
// (x - h is exact; 22 significant bits, the scaled plane never leaves the normal range for |x| > 2^-13 and degrades gracefully
// below), and a product needs three MFMAs - h h' into one accumulator, h m' + m h' into a second that joins with 2^-11 at the
// end (the dropped m m' term is below 2^-24 of |x| |w|).  The weights' planes are cut by the host (compiler.mbs_pack), the
// pixels' and the depthwise results' in registers (5 VALU operations per pair of values).  Precondition: |x| < 65504 for the
// block input (a float16 plane has no more range; beyond it the result is undefined - the planes become inf, the sums NaN, and the
// ReLU6 behind the expand conv turns that into 0 or 6); the depthwise
// results are ReLU6'd.  One K = 32 step takes 8 channels per lane: the block input's channels 32 c + 8 g .. + 7, and for the
// projection the four channels of expanded tile 2 q and the four of tile 2 q + 1 a lane holds after the depthwise stage.
typedef _Float16 mbs_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 mbs_h8 __attribute__((ext_vector_type(8)));
typedef unsigned mbs_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mbs_split8(const float (&v)[8], mbs_u4& h, mbs_u4& m) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned hh, mm;
        yr_cut2(v[2 * p], v[2 * p + 1], hh, mm);     // (4 operations per pair: yr_common.h)
        h[p] = hh;
        m[p] = mm;
    }
}
__device__ __forceinline__ v4f mbs_mfma(mbs_u4 a, mbs_u4 b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mbs_h8, a), __builtin_bit_cast(mbs_h8, b), c, 0, 0, 0);
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) - the slices below index registers with the counter
template <int N, class F>
__device__ __forceinline__ void mbk_for(F&& f) {
    if constexpr (N > 0) {
        mbk_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// mbr_dw_row in THREE parts of equal VALU weight, so that a part can sit between two MFMAs: the centre tap (two v_pk_fma_f32), the left
// neighbour's tap, the right neighbour's (four v_fmac_f32_dpp each).  No s_nop in front: the rows these read were written by VALU
// instructions slices ago (or came from LDS).
__device__ __forceinline__ void mbk_dw_part(const int part, v4f& acc, const v4f e, const v4f wt) {
    if (part == 0) { acc = __builtin_elementwise_fma(e, wt, acc); return; }
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    if (part == 1)
        asm("v_fmac_f32_dpp %0, %4, %8" MBR_DPP("row_shr:1") "v_fmac_f32_dpp %1, %5, %9" MBR_DPP("row_shr:1")
            "v_fmac_f32_dpp %2, %6, %10" MBR_DPP("row_shr:1") "v_fmac_f32_dpp %3, %7, %11" MBR_DPP("row_shr:1")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(wt[0]), "v"(wt[1]), "v"(wt[2]), "v"(wt[3]));
    else
        asm("v_fmac_f32_dpp %0, %4, %8" MBR_DPP("row_shl:1") "v_fmac_f32_dpp %1, %5, %9" MBR_DPP("row_shl:1")
            "v_fmac_f32_dpp %2, %6, %10" MBR_DPP("row_shl:1") "v_fmac_f32_dpp %3, %7, %11" MBR_DPP("row_shl:1")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(wt[0]), "v"(wt[1]), "v"(wt[2]), "v"(wt[3]));
    acc = (v4f){a0, a1, a2, a3};
}
// ... and of the stride-2 tap row (even output row: own lane = E_j, row_shl:8 = O_j, row_shl:1 = E_j+1; lanes 0..7 written)
#define MBK_DW2P(ctl)                                                                                                                 \
    asm("v_fmac_f32_dpp %0, %4, %8" MBR_DPPM(ctl, "0x3") "v_fmac_f32_dpp %1, %5, %9" MBR_DPPM(ctl, "0x3")                                \
        "v_fmac_f32_dpp %2, %6, %10" MBR_DPPM(ctl, "0x3") "v_fmac_f32_dpp %3, %7, %11" MBR_DPPM(ctl, "0x3")                              \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(wt[0]), "v"(wt[1]), "v"(wt[2]), "v"(wt[3]))
__device__ __forceinline__ void mbk_dw2_part(const int part, v4f& acc, const v4f e, const v4f wt) {
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    if (part == 0) MBK_DW2P("quad_perm:[0,1,2,3]");
    else if (part == 1) MBK_DW2P("row_shl:8");
    else MBK_DW2P("row_shl:1");
    acc = (v4f){a0, a1, a2, a3};
}

// rows of a head map a walking wave takes per segment (headwalk.hip / headwalk_h.hip; shape only: the squeeze-excite sums are
// grouped by (strip, segment))
static inline int hw_seg_rows(int H) { return H <= 16 ? H : (H + ((H + 12) / 13) - 1) / ((H + 12) / 13); }
