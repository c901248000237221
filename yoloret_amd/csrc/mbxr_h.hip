// YR_OP_MBX on 16-bit activations, ROW-WALKING and REGISTER-CHAINED (the float32 twin is mbr.hip: mbe_kernel):
//   expand 1x1 + BN + act (v_mfma_f32_16x16x32_bf16 / _f16) -> depthwise KxK (K = 3 | 5, stride 1 | 2, TF SAME) + BN + act,
// the depthwise map stored (16-bit) and, for the squeeze of squeeze-excite (reference code/yolo3/efficientnet.py:406-438,
// 467-536: tf.reduce_mean over H, W), the per-channel sums of what was stored.
//
// The LDS-tiled form (mbh.hip MODE 1) moves every expanded value through LDS K times and holds one to three workgroups per CU
// (PMC round 3: 51-57 % of its wave time waiting, 0.04-0.08 of any pipe).  Here NOTHING of the walk touches LDS:
//   * a wave owns a strip of 16 input columns x NT expanded tiles of 16 channels and walks down the rows; the expand GEMM of
//     a row has the strip's pixels as the MFMA's N dimension, the pixel operand comes from global memory as it lies there
//     (lane (pixel p, k group g): the 8 consecutive channels 32 c + 8 g .. + 7 = one 16-byte load = one MFMA B operand);
//   * the MFMA result layout (a lane: 4 consecutive channels of one pixel; the 16 lanes of a DPP row: the 16 pixels) is the
//     layout the depthwise conv wants: horizontal taps by DPP row shifts riding on the multiply-add's operand
//     (v_fmac_f32_dpp), vertical taps = the last K rows of the walk in registers;
//   * expand weights (A fragments), BN parameters and the K x K taps of the wave's tiles are STATIONARY in registers;
//   * the matrix pipe is its own pipe for 16-bit operands, so the kernel is bound by the depthwise VALU work alone:
//     K*K multiply-adds + two activations + the rounding per value - what the arithmetic itself costs.
// Same op, same parameter layouts as mbh.hip MODE 1 (yoloret_hip.h: YR_OP_MBX); the forced "tile" th = 255 selects this form
// and tw = row segments per strip (0: the launcher's choice) - yr_autotune times it next to the LDS-tiled tiles.
// Numerics: expand accumulates in float32, BN in float32, activation (swish: hardware exp2 + rcp, as mbh.hip), the expanded
// value stays float32 (never rounded), depthwise in float32 with the BN scale folded into the taps, one rounding at the store.
#include "yr_common.h"
#include <type_traits>
#include <cstdlib>

typedef float xr_f4 __attribute__((ext_vector_type(4)));
typedef unsigned xr_u4 __attribute__((ext_vector_type(4)));
typedef unsigned xr_u2 __attribute__((ext_vector_type(2)));
template <class T> using xr_v8 = T __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t xr_rsrc;

struct MbxrArgs {
    const void* x; void* out; const void* we; const float* prm;
    float* part; int ld_part, rows_cap;
    int H, W, Ho, Wo, Cin, CexpP, ld_in, ld_out, KP, pad_t, pad_l;
    int strips, segs, seg_rows, T, groups, nwaves;
    int qrows, nquanta;   // the squeeze sums leave per (strip, quantum of qrows output rows): grouped by the map's SHAPE, whatever the segments
};

__device__ __forceinline__ xr_rsrc xr_make_rsrc(const void* base, unsigned bytes) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#define XR_DEAD 0x7f000000u

template <class T>
__device__ __forceinline__ xr_f4 xr_mfma(xr_u4 w, xr_u4 x, xr_f4 acc) {
    if constexpr (yr_elem<T>::dtype == YR_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xr_v8<__bf16>, w), __builtin_bit_cast(xr_v8<__bf16>, x), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xr_v8<_Float16>, w), __builtin_bit_cast(xr_v8<_Float16>, x), acc, 0, 0, 0);
}

// ... and the 16-deep step (v_mfma_f32_16x16x16_{bf16,f16}): B = the lane's 4 values of k = 4 g .. 4 g + 3
template <class T>
__device__ __forceinline__ xr_f4 xr_mfma16(xr_u2 w, xr_u2 x, xr_f4 acc) {
    typedef short xr_s4 __attribute__((ext_vector_type(4)));
    typedef _Float16 xr_h4 __attribute__((ext_vector_type(4)));
    if constexpr (yr_elem<T>::dtype == YR_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(xr_s4, w), __builtin_bit_cast(xr_s4, x), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(xr_h4, w), __builtin_bit_cast(xr_h4, x), acc, 0, 0, 0);
}

// one tap ROW of the depthwise conv for the lane's 4 channels (see mbr.hip: the DPP shift rides on v_fmac's first operand,
// the four channels' chains interleaved tap-major; s_nop 1 covers the VALU-write -> DPP-read hazard the compiler cannot see)
#define XR_DPP(ctl) " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void xr_row3(xr_f4& acc, const xr_f4 e, const xr_f4 w0, const xr_f4 w1, const xr_f4 w2) {
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %4, %8" XR_DPP("row_shr:1") "v_fmac_f32_dpp %1, %5, %9" XR_DPP("row_shr:1")
        "v_fmac_f32_dpp %2, %6, %10" XR_DPP("row_shr:1") "v_fmac_f32_dpp %3, %7, %11" XR_DPP("row_shr:1")
        "v_fmac_f32 %0, %4, %12\n\t" "v_fmac_f32 %1, %5, %13\n\t" "v_fmac_f32 %2, %6, %14\n\t" "v_fmac_f32 %3, %7, %15\n\t"
        "v_fmac_f32_dpp %0, %4, %16" XR_DPP("row_shl:1") "v_fmac_f32_dpp %1, %5, %17" XR_DPP("row_shl:1")
        "v_fmac_f32_dpp %2, %6, %18" XR_DPP("row_shl:1") "v_fmac_f32_dpp %3, %7, %19" XR_DPP("row_shl:1")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]),
          "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]), "v"(w2[0]), "v"(w2[1]), "v"(w2[2]), "v"(w2[3]));
    acc = (xr_f4){a0, a1, a2, a3};
}
// five taps: two asm blocks (an asm statement takes at most 30 operands): taps 0..2 at row_shr:2, row_shr:1, 0; taps 3..4 at row_shl:1, row_shl:2
__device__ __forceinline__ void xr_row5(xr_f4& acc, const xr_f4 e, const xr_f4 w0, const xr_f4 w1, const xr_f4 w2, const xr_f4 w3, const xr_f4 w4) {
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %4, %8" XR_DPP("row_shr:2") "v_fmac_f32_dpp %1, %5, %9" XR_DPP("row_shr:2")
        "v_fmac_f32_dpp %2, %6, %10" XR_DPP("row_shr:2") "v_fmac_f32_dpp %3, %7, %11" XR_DPP("row_shr:2")
        "v_fmac_f32_dpp %0, %4, %12" XR_DPP("row_shr:1") "v_fmac_f32_dpp %1, %5, %13" XR_DPP("row_shr:1")
        "v_fmac_f32_dpp %2, %6, %14" XR_DPP("row_shr:1") "v_fmac_f32_dpp %3, %7, %15" XR_DPP("row_shr:1")
        "v_fmac_f32 %0, %4, %16\n\t" "v_fmac_f32 %1, %5, %17\n\t" "v_fmac_f32 %2, %6, %18\n\t" "v_fmac_f32 %3, %7, %19\n\t"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]),
          "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]), "v"(w2[0]), "v"(w2[1]), "v"(w2[2]), "v"(w2[3]));
    asm("v_fmac_f32_dpp %0, %4, %8" XR_DPP("row_shl:1") "v_fmac_f32_dpp %1, %5, %9" XR_DPP("row_shl:1")
        "v_fmac_f32_dpp %2, %6, %10" XR_DPP("row_shl:1") "v_fmac_f32_dpp %3, %7, %11" XR_DPP("row_shl:1")
        "v_fmac_f32_dpp %0, %4, %12" XR_DPP("row_shl:2") "v_fmac_f32_dpp %1, %5, %13" XR_DPP("row_shl:2")
        "v_fmac_f32_dpp %2, %6, %14" XR_DPP("row_shl:2") "v_fmac_f32_dpp %3, %7, %15" XR_DPP("row_shl:2")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(w3[0]), "v"(w3[1]), "v"(w3[2]), "v"(w3[3]),
          "v"(w4[0]), "v"(w4[1]), "v"(w4[2]), "v"(w4[3]));
    acc = (xr_f4){a0, a1, a2, a3};
}

template <int ACT>
__device__ __forceinline__ float xr_act(float v, float hi) {   // hi: 6 (relu6) / 1 (swish) inside the image, 0 outside
    if constexpr (ACT == 0) return __builtin_amdgcn_fmed3f(v, 0.0f, hi);
    else return hi * (v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f)));
}

// The same on the 4 channels of a lane at once, the multiplies and adds as PACKED float32 instructions (v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32: two values per issue slot, each IEEE-exact like its scalar form, so results are bit-identical to xr_act).  These
// kernels are bound by VALU issue, and with a 3x3 depthwise conv the two swishes are 60 % of it (tools/valu_rate.hip: v_exp_f32 and
// v_rcp_f32 cost two slots each on gfx950, everything else one): 9 -> 6.5 slots per expanded value, 8 -> 6 per stored value.
typedef float xr_f2 __attribute__((ext_vector_type(2)));
template <int ACT>
__device__ __forceinline__ xr_f4 xr_act4(const xr_f4 v, const float hi) {
    xr_f4 o;
    if constexpr (ACT == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_fmed3f(v[i], 0.0f, hi);
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const xr_f2 x = {v[2 * h], v[2 * h + 1]};
            const xr_f2 u = x * (xr_f2){-1.44269504088896341f, -1.44269504088896341f};
            const xr_f2 e = (xr_f2){__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + (xr_f2){1.0f, 1.0f};
            const xr_f2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
            const xr_f2 y = (xr_f2){hi, hi} * (x * r);
            o[2 * h] = y[0]; o[2 * h + 1] = y[1];
        }
    }
    return o;
}
// ... behind the expand conv's BatchNorm: act(d * scale + shift)
template <int ACT>
__device__ __forceinline__ xr_f4 xr_bn_act4(const xr_f4 d, const xr_f4 sc, const xr_f4 sh, const float hi) {
    xr_f4 t;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const xr_f2 y = __builtin_elementwise_fma((xr_f2){d[2 * h], d[2 * h + 1]}, (xr_f2){sc[2 * h], sc[2 * h + 1]}, (xr_f2){sh[2 * h], sh[2 * h + 1]});
        t[2 * h] = y[0]; t[2 * h + 1] = y[1];
    }
    return xr_act4<ACT>(t, hi);
}

// STRIDE 2, PAIRED OUTPUT ROWS (as mbr.hip): the expand conv does not care which pixel sits in which lane, so a stride-2 strip
// loads the EVEN input columns E_0..E_7 into lanes 0..7 of a DPP row and the ODD ones O_0..O_7 into lanes 8..15.  Tap dx of output
// column j is E_(j + dx/2) (dx even) or O_(j + (dx-1)/2) (dx odd).  An even output row accumulates in lanes 0..7 (bank_mask 0x3:
// own lane, row_shl:8, row_shl:1, row_shl:9, row_shl:2), the odd row below it in lanes 8..15 of the SAME registers (bank_mask
// 0xc: row_shr:8, own lane, row_shr:7, row_shl:1, row_shr:6): one activation, one rounding, one store (and in the whole-block
// kernels one projection and one barrier) per PAIR of output rows, all 16 lanes of it useful (14 / 12 of 16 for 3x3 / 5x5).
#define XR_DPPM(ctl, bank) " " ctl " row_mask:0xf bank_mask:" bank " bound_ctrl:1\n\t"
#define XR_PAIR3(c0, c1, c2, bank)                                                                                                   \
    asm("s_nop 1\n\t"                                                                                                                \
        "v_fmac_f32_dpp %0, %4, %8" XR_DPPM(c0, bank) "v_fmac_f32_dpp %1, %5, %9" XR_DPPM(c0, bank)                                 \
        "v_fmac_f32_dpp %2, %6, %10" XR_DPPM(c0, bank) "v_fmac_f32_dpp %3, %7, %11" XR_DPPM(c0, bank)                               \
        "v_fmac_f32_dpp %0, %4, %12" XR_DPPM(c1, bank) "v_fmac_f32_dpp %1, %5, %13" XR_DPPM(c1, bank)                               \
        "v_fmac_f32_dpp %2, %6, %14" XR_DPPM(c1, bank) "v_fmac_f32_dpp %3, %7, %15" XR_DPPM(c1, bank)                               \
        "v_fmac_f32_dpp %0, %4, %16" XR_DPPM(c2, bank) "v_fmac_f32_dpp %1, %5, %17" XR_DPPM(c2, bank)                               \
        "v_fmac_f32_dpp %2, %6, %18" XR_DPPM(c2, bank) "v_fmac_f32_dpp %3, %7, %19" XR_DPPM(c2, bank)                               \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                                                                     \
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(w[0][0]), "v"(w[0][1]), "v"(w[0][2]), "v"(w[0][3]),                        \
          "v"(w[1][0]), "v"(w[1][1]), "v"(w[1][2]), "v"(w[1][3]), "v"(w[2][0]), "v"(w[2][1]), "v"(w[2][2]), "v"(w[2][3]))
#define XR_PAIR2(c3, c4, bank)                                                                                                       \
    asm("v_fmac_f32_dpp %0, %4, %8" XR_DPPM(c3, bank) "v_fmac_f32_dpp %1, %5, %9" XR_DPPM(c3, bank)                                 \
        "v_fmac_f32_dpp %2, %6, %10" XR_DPPM(c3, bank) "v_fmac_f32_dpp %3, %7, %11" XR_DPPM(c3, bank)                               \
        "v_fmac_f32_dpp %0, %4, %12" XR_DPPM(c4, bank) "v_fmac_f32_dpp %1, %5, %13" XR_DPPM(c4, bank)                               \
        "v_fmac_f32_dpp %2, %6, %14" XR_DPPM(c4, bank) "v_fmac_f32_dpp %3, %7, %15" XR_DPPM(c4, bank)                               \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                                                                     \
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(w[3][0]), "v"(w[3][1]), "v"(w[3][2]), "v"(w[3][3]),                        \
          "v"(w[4][0]), "v"(w[4][1]), "v"(w[4][2]), "v"(w[4][3]))
// one tap row (K taps at w[0..K-1]) of the paired form: ODD = the odd output row of the pair (lanes 8..15)
template <int K, bool ODD>
__device__ __forceinline__ void xr_row_pair(xr_f4& acc, const xr_f4 e, const xr_f4* w) {
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    if constexpr (K == 3) {
        if constexpr (!ODD) XR_PAIR3("quad_perm:[0,1,2,3]", "row_shl:8", "row_shl:1", "0x3");
        else XR_PAIR3("row_shr:8", "quad_perm:[0,1,2,3]", "row_shr:7", "0xc");
    } else {
        if constexpr (!ODD) { XR_PAIR3("quad_perm:[0,1,2,3]", "row_shl:8", "row_shl:1", "0x3"); XR_PAIR2("row_shl:9", "row_shl:2", "0x3"); }
        else { XR_PAIR3("row_shr:8", "quad_perm:[0,1,2,3]", "row_shr:7", "0xc"); XR_PAIR2("row_shl:1", "row_shr:6", "0xc"); }
    }
    acc = (xr_f4){a0, a1, a2, a3};
}

// 5x5 depthwise taps WITHOUT a register per tap: the 25 taps of a channel live in TWO registers, lane p of every 16-lane row
// holding tap p (register 0) / tap 16 + p (register 1) of the lane's channel, and a multiply-add reads the tap it needs through
// DPP row_newbcast:p (gfx90a+: lane p of the row, broadcast to the row; tools/dpp_bcast_test.hip).  DPP modifies one operand
// only, so the horizontal shift cannot ride on the same instruction: the five COLUMN sums S_dx = sum_ky tap(ky, dx) * e_ky are
// accumulated unshifted and the shifts are applied once per output row, to the sums:
//   out = shift + S_2 + shr2(S_0) + shr1(S_1) + shl1(S_3) + shl2(S_4)
// 30 instead of 25 instructions per channel and output row, 8 instead of 100 tap registers per tile, no LDS table (an LDS table
// read where it is used costs 25 ds_read_b128 per tile and output row: 0.31 ms on EfficientNet-lite0's stage-3 entry, LDS-bound).
template <int KY> __device__ __forceinline__ void xr_bc5_row(float (&S)[5][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1);
template <> __device__ __forceinline__ void xr_bc5_row<0>(float (&S)[5][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_mul_f32_dpp %0, %16, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %17, %13 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %18, %14 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %19, %15 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %4, %16, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %5, %17, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %6, %18, %14 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %7, %19, %15 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %8, %16, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %9, %17, %13 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %10, %18, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %11, %19, %15 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(S[0][0]), "=&v"(S[0][1]), "=&v"(S[0][2]), "=&v"(S[0][3]), "=&v"(S[1][0]), "=&v"(S[1][1]), "=&v"(S[1][2]), "=&v"(S[1][3]), "=&v"(S[2][0]), "=&v"(S[2][1]), "=&v"(S[2][2]), "=&v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
    asm("v_mul_f32_dpp %0, %12, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %13, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %14, %10 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %15, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %4, %12, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %5, %13, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %6, %14, %10 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %7, %15, %11 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(S[3][0]), "=&v"(S[3][1]), "=&v"(S[3][2]), "=&v"(S[3][3]), "=&v"(S[4][0]), "=&v"(S[4][1]), "=&v"(S[4][2]), "=&v"(S[4][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
template <> __device__ __forceinline__ void xr_bc5_row<1>(float (&S)[5][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_fmac_f32_dpp %0, %16, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %16, %12 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %17, %13 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %18, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %19, %15 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %8, %16, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %9, %17, %13 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %10, %18, %14 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %11, %19, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[2][2]), "+v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
    asm("v_fmac_f32_dpp %0, %12, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %13, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %14, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %15, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %12, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %13, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %14, %10 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %15, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(S[3][0]), "+v"(S[3][1]), "+v"(S[3][2]), "+v"(S[3][3]), "+v"(S[4][0]), "+v"(S[4][1]), "+v"(S[4][2]), "+v"(S[4][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
template <> __device__ __forceinline__ void xr_bc5_row<2>(float (&S)[5][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_fmac_f32_dpp %0, %16, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %16, %12 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %17, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %18, %14 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %19, %15 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %8, %16, %12 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %9, %17, %13 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %10, %18, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %11, %19, %15 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[2][2]), "+v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
    asm("v_fmac_f32_dpp %0, %12, %8 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %13, %9 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %14, %10 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %15, %11 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %12, %8 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %13, %9 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %14, %10 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %15, %11 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(S[3][0]), "+v"(S[3][1]), "+v"(S[3][2]), "+v"(S[3][3]), "+v"(S[4][0]), "+v"(S[4][1]), "+v"(S[4][2]), "+v"(S[4][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
template <> __device__ __forceinline__ void xr_bc5_row<3>(float (&S)[5][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_fmac_f32_dpp %0, %16, %12 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %20, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %21, %13 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %22, %14 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %23, %15 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %8, %20, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %9, %21, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %10, %22, %14 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %11, %23, %15 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[2][2]), "+v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
    asm("v_fmac_f32_dpp %0, %16, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %17, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %18, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %19, %11 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %16, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %17, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %18, %10 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %19, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(S[3][0]), "+v"(S[3][1]), "+v"(S[3][2]), "+v"(S[3][3]), "+v"(S[4][0]), "+v"(S[4][1]), "+v"(S[4][2]), "+v"(S[4][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
template <> __device__ __forceinline__ void xr_bc5_row<4>(float (&S)[5][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_fmac_f32_dpp %0, %20, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %21, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %22, %14 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %23, %15 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %20, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %21, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %22, %14 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %23, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %8, %20, %12 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %9, %21, %13 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %10, %22, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %11, %23, %15 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[2][2]), "+v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
    asm("v_fmac_f32_dpp %0, %16, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %17, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %18, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %19, %11 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %16, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %17, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %18, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %19, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(S[3][0]), "+v"(S[3][1]), "+v"(S[3][2]), "+v"(S[3][3]), "+v"(S[4][0]), "+v"(S[4][1]), "+v"(S[4][2]), "+v"(S[4][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
__device__ __forceinline__ xr_f4 xr_bc5_finish(const float (&S)[5][4], const xr_f4 shift) {
    float o0 = shift[0] + S[2][0], o1 = shift[1] + S[2][1], o2 = shift[2] + S[2][2], o3 = shift[3] + S[2][3];
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %5, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %6, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %7, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %9, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %10, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %11, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %12, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %13, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %14, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %15, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %16, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %17, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %18, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %19, %3 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3)
        : "v"(S[0][0]), "v"(S[0][1]), "v"(S[0][2]), "v"(S[0][3]), "v"(S[1][0]), "v"(S[1][1]), "v"(S[1][2]), "v"(S[1][3]), "v"(S[3][0]), "v"(S[3][1]), "v"(S[3][2]), "v"(S[3][3]), "v"(S[4][0]), "v"(S[4][1]), "v"(S[4][2]), "v"(S[4][3]));
    return (xr_f4){o0, o1, o2, o3};
}

// ... and the broadcast-tap form for PAIRED output rows (stride 2, even input columns E in lanes 0..7, odd ones O in 8..15).  The
// column sums live at their INPUT lane: S_dx with dx even is only ever needed at E lanes, with dx odd only at O lanes, so S_0 | S_1
// share a register (lanes 0..7 | 8..15, bank masks), S_2 | S_3 the next, S_4 a third: three registers per channel and output row of
// the pair.  An input row feeds the even output row's sums with its taps ky and the odd row's sums with ky - 2 (same code, other array).
template <int KY> __device__ __forceinline__ void xr_bc5_half_row(float (&S)[3][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1);
template <> __device__ __forceinline__ void xr_bc5_half_row<0>(float (&S)[3][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_fmac_f32_dpp %0, %16, %12 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:0 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %0, %16, %12 row_newbcast:1 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:1 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:1 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:1 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %4, %16, %12 row_newbcast:2 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %5, %17, %13 row_newbcast:2 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %6, %18, %14 row_newbcast:2 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %7, %19, %15 row_newbcast:2 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %4, %16, %12 row_newbcast:3 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %5, %17, %13 row_newbcast:3 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %6, %18, %14 row_newbcast:3 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %7, %19, %15 row_newbcast:3 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %8, %16, %12 row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %9, %17, %13 row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %10, %18, %14 row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %11, %19, %15 row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
        : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[2][2]), "+v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
template <> __device__ __forceinline__ void xr_bc5_half_row<1>(float (&S)[3][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_fmac_f32_dpp %0, %16, %12 row_newbcast:5 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:5 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:5 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:5 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %0, %16, %12 row_newbcast:6 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:6 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:6 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:6 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %4, %16, %12 row_newbcast:7 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %5, %17, %13 row_newbcast:7 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %6, %18, %14 row_newbcast:7 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %7, %19, %15 row_newbcast:7 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %4, %16, %12 row_newbcast:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %5, %17, %13 row_newbcast:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %6, %18, %14 row_newbcast:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %7, %19, %15 row_newbcast:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %8, %16, %12 row_newbcast:9 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %9, %17, %13 row_newbcast:9 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %10, %18, %14 row_newbcast:9 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %11, %19, %15 row_newbcast:9 row_mask:0xf bank_mask:0x3\n\t"
        : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[2][2]), "+v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
template <> __device__ __forceinline__ void xr_bc5_half_row<2>(float (&S)[3][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_fmac_f32_dpp %0, %16, %12 row_newbcast:10 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:10 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:10 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:10 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %0, %16, %12 row_newbcast:11 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:11 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:11 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:11 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %4, %16, %12 row_newbcast:12 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %5, %17, %13 row_newbcast:12 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %6, %18, %14 row_newbcast:12 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %7, %19, %15 row_newbcast:12 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %4, %16, %12 row_newbcast:13 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %5, %17, %13 row_newbcast:13 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %6, %18, %14 row_newbcast:13 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %7, %19, %15 row_newbcast:13 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %8, %16, %12 row_newbcast:14 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %9, %17, %13 row_newbcast:14 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %10, %18, %14 row_newbcast:14 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %11, %19, %15 row_newbcast:14 row_mask:0xf bank_mask:0x3\n\t"
        : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[2][2]), "+v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
template <> __device__ __forceinline__ void xr_bc5_half_row<3>(float (&S)[3][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_fmac_f32_dpp %0, %16, %12 row_newbcast:15 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %1, %17, %13 row_newbcast:15 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %2, %18, %14 row_newbcast:15 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %3, %19, %15 row_newbcast:15 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %0, %20, %12 row_newbcast:0 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %1, %21, %13 row_newbcast:0 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %2, %22, %14 row_newbcast:0 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %3, %23, %15 row_newbcast:0 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %4, %20, %12 row_newbcast:1 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %5, %21, %13 row_newbcast:1 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %6, %22, %14 row_newbcast:1 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %7, %23, %15 row_newbcast:1 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %4, %20, %12 row_newbcast:2 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %5, %21, %13 row_newbcast:2 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %6, %22, %14 row_newbcast:2 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %7, %23, %15 row_newbcast:2 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %8, %20, %12 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %9, %21, %13 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %10, %22, %14 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %11, %23, %15 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\t"
        : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[2][2]), "+v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
template <> __device__ __forceinline__ void xr_bc5_half_row<4>(float (&S)[3][4], const xr_f4 e, const xr_f4 t0, const xr_f4 t1) {
    asm("v_fmac_f32_dpp %0, %20, %12 row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %1, %21, %13 row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %2, %22, %14 row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %3, %23, %15 row_newbcast:4 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %0, %20, %12 row_newbcast:5 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %1, %21, %13 row_newbcast:5 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %2, %22, %14 row_newbcast:5 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %3, %23, %15 row_newbcast:5 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %4, %20, %12 row_newbcast:6 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %5, %21, %13 row_newbcast:6 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %6, %22, %14 row_newbcast:6 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %7, %23, %15 row_newbcast:6 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %4, %20, %12 row_newbcast:7 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %5, %21, %13 row_newbcast:7 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %6, %22, %14 row_newbcast:7 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %7, %23, %15 row_newbcast:7 row_mask:0xf bank_mask:0xc\n\t"
        "v_fmac_f32_dpp %8, %20, %12 row_newbcast:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %9, %21, %13 row_newbcast:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %10, %22, %14 row_newbcast:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_fmac_f32_dpp %11, %23, %15 row_newbcast:8 row_mask:0xf bank_mask:0x3\n\t"
        : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[2][2]), "+v"(S[2][3])
        : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(t0[0]), "v"(t0[1]), "v"(t0[2]), "v"(t0[3]), "v"(t1[0]), "v"(t1[1]), "v"(t1[2]), "v"(t1[3]));
}
// the pair's depthwise results: the even row's in lanes 0..7, the odd row's in lanes 8..15
__device__ __forceinline__ xr_f4 xr_bc5_pair_finish(const float (&SA)[3][4], const float (&SB)[3][4], const xr_f4 shift) {
    float o0 = shift[0], o1 = shift[1], o2 = shift[2], o3 = shift[3];
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %0 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %5, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %6, %2 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %7, %3 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:8 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %5, %1 row_shl:8 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %6, %2 row_shl:8 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %7, %3 row_shl:8 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %8, %0 row_shl:1 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %9, %1 row_shl:1 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %10, %2 row_shl:1 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %11, %3 row_shl:1 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %8, %0 row_shl:9 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %9, %1 row_shl:9 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %10, %2 row_shl:9 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %11, %3 row_shl:9 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %12, %0 row_shl:2 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %13, %1 row_shl:2 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %14, %2 row_shl:2 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %15, %3 row_shl:2 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %16, %0 row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %17, %1 row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %18, %2 row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %19, %3 row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %16, %0 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %17, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %18, %2 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %19, %3 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %20, %0 row_shr:7 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %21, %1 row_shr:7 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %22, %2 row_shr:7 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %23, %3 row_shr:7 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %20, %0 row_shl:1 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %21, %1 row_shl:1 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %22, %2 row_shl:1 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %23, %3 row_shl:1 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %24, %0 row_shr:6 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %25, %1 row_shr:6 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %26, %2 row_shr:6 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %27, %3 row_shr:6 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
        : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3)
        : "v"(SA[0][0]), "v"(SA[0][1]), "v"(SA[0][2]), "v"(SA[0][3]), "v"(SA[1][0]), "v"(SA[1][1]), "v"(SA[1][2]), "v"(SA[1][3]), "v"(SA[2][0]), "v"(SA[2][1]), "v"(SA[2][2]), "v"(SA[2][3]), "v"(SB[0][0]), "v"(SB[0][1]), "v"(SB[0][2]), "v"(SB[0][3]), "v"(SB[1][0]), "v"(SB[1][1]), "v"(SB[1][2]), "v"(SB[1][3]), "v"(SB[2][0]), "v"(SB[2][1]), "v"(SB[2][2]), "v"(SB[2][3]));
    return (xr_f4){o0, o1, o2, o3};
}

// K: depthwise kernel, S: stride, ACT: 0 relu6 / 1 swish (both activations), NC: 32-channel chunks of the block input, NT: tiles per wave
template <class T, int K, int S, int ACT, int NC, int NT, int MW>
__global__ __launch_bounds__(256, MW) void mbxr_kernel(MbxrArgs a) {
    constexpr int KK = K * K, PAD = (K - 1) / 2, NOUT = (16 - K) / S + 1;
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    int gw = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (gw >= a.nwaves) return;
    const int g = gw % a.groups; gw /= a.groups;
    const int seg = gw % a.segs; gw /= a.segs;
    const int strip = gw % a.strips;
    const int b = gw / a.strips;
    const int t0 = g * NT;
    const int yo0 = seg * a.seg_rows, yo1 = min(yo0 + a.seg_rows, a.Ho);
    constexpr bool BC5 = K == 5 && NC >= 4;
    constexpr bool PAIR = S == 2 && !BC5;   // stride 2: pairs of output rows, even input columns in lanes 0..7, odd ones in 8..15 (xr_row_pair)
    const int podd = PAIR ? px >> 3 : 0;
    const int xin = S * NOUT * strip - a.pad_l + (PAIR ? 2 * (px & 7) + podd : px);
    const int xc = min(max(xin, 0), a.W - 1);
    constexpr float HI = ACT == 0 ? 6.f : 1.f;
    const float hi = (xin >= 0 && xin < a.W) ? HI : 0.f;
    const int jo = PAIR ? (px & 7) : (px - PAD) / S, xo = NOUT * strip + jo;
    const bool out_lane = (PAIR ? true : (px >= PAD && (px - PAD) % S == 0)) && jo < NOUT && xo < a.Wo;
    const float omask = out_lane ? 1.f : 0.f;

    // ---- stationary: expand A fragments, BN rows, taps (times the depthwise BN scale) of this wave's tiles
    xr_u4 aw[NT][NC];
    // 5x5 behind four input chunks: the broadcast-tap form (lane p of a row holds tap p / tap 16 + p, xr_bc5_row) - a register per tap
    // leaves those blocks two waves per SIMD at 242 registers; with fewer chunks the 20 % more instructions of that form cost more
    // than the registers (measured, batch 128: 24 -> 144 s2 0.227 vs 0.270 ms; 112 -> 672 s2 0.126 vs 0.107)
    xr_f4 es[NT], eh[NT], dh[NT], tp[NT][BC5 ? 2 : KK], ssum[NT];
    unsigned ooff[NT];
    bool tlive[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        tlive[j] = t0 + j < a.T;
        const int t = min(t0 + j, a.T - 1);
        const int ch = 16 * t + 4 * mg;                                   // this lane's 4 expanded channels of tile j
        const char* wrow = reinterpret_cast<const char*>(a.we) + ((size_t)(16 * t + px) * a.KP + 8 * mg) * 2;
#pragma unroll
        for (int c = 0; c < NC; ++c) aw[j][c] = *reinterpret_cast<const xr_u4*>(wrow + 64 * c);
        const xr_f4 dsc = *reinterpret_cast<const xr_f4*>(a.prm + (size_t)KK * a.CexpP + ch);
        dh[j] = *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(KK + 1) * a.CexpP + ch);
        es[j] = *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(KK + 2) * a.CexpP + ch);
        eh[j] = *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(KK + 3) * a.CexpP + ch);
        if constexpr (BC5) {
            tp[j][0] = *reinterpret_cast<const xr_f4*>(a.prm + (size_t)px * a.CexpP + ch) * dsc;
            tp[j][1] = 16 + px < KK ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(16 + px) * a.CexpP + ch) * dsc : (xr_f4){0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int q = 0; q < KK; ++q) tp[j][q] = *reinterpret_cast<const xr_f4*>(a.prm + (size_t)q * a.CexpP + ch) * dsc;
        }
        ssum[j] = (xr_f4){0.f, 0.f, 0.f, 0.f};
        ooff[j] = (tlive[j] && out_lane) ? (unsigned)ch * 2u : XR_DEAD;
    }
    const xr_rsrc xsrc = xr_make_rsrc(reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.ld_in, (unsigned)(a.H * a.W * a.ld_in) * 2u);
    const xr_rsrc osrc = xr_make_rsrc(reinterpret_cast<T*>(a.out) + (size_t)b * a.Ho * a.Wo * a.ld_out, (unsigned)(a.Ho * a.Wo * a.ld_out) * 2u);
    const int rbeg = S * yo0 - a.pad_t, nout = yo1 - yo0;
    unsigned xoff[NC];     // the lane's 16 bytes of chunk c within the row; k groups beyond the input's channels read zeros (dead offset)
#pragma unroll
    for (int c = 0; c < NC; ++c) xoff[c] = (32 * c + 8 * mg < a.Cin) ? ((unsigned)xc * (unsigned)a.ld_in + 32u * c + 8u * mg) * 2u : XR_DEAD;
    const unsigned xrow = (unsigned)(a.W * a.ld_in) * 2u;
    struct XRow { xr_u4 m[NC]; };
    XRow xa, xb;
    auto load_row = [&](XRow& x, int r) {
        const unsigned so = (unsigned)min(max(r, 0), a.H - 1) * xrow;
#pragma unroll
        for (int c = 0; c < NC; ++c) x.m[c] = __builtin_bit_cast(xr_u4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[c], so, 0));
    };
    load_row(xa, rbeg);
    xr_f4 ring[NT][K - 1];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < K - 1; ++q) ring[j][q] = (xr_f4){0.f, 0.f, 0.f, 0.f};

    // ---- the squeeze: the wave's channel sums over one QUANTUM of output rows -> row (strip, quantum) of the partial-sum buffer.  Segments
    // are whole quanta (launch_mbxr), so which wave sums a quantum changes with the tuned segmentation, WHAT is summed in which order
    // does not: the gate - and with it the model's result - is the same for every tuning table and batch size.
    const xr_rsrc psrc = xr_make_rsrc(a.part != nullptr ? a.part + (size_t)b * a.rows_cap * a.ld_part : reinterpret_cast<const float*>(a.x), a.part != nullptr ? (unsigned)(a.rows_cap * a.ld_part) * 4u : 0u);
    // (the lane that stores a tile's sums: pixel 0 of its DPP row; byte offset of the lane's 4 channels of tile j within a buffer row: pch + 64 j)
    const unsigned pch = (unsigned)(16 * t0 + 4 * mg) * 4u;
    auto poff = [&](const int j) { return (tlive[j] && px == 0 && 16 * (t0 + j) + 4 * mg < a.ld_part) ? pch + 64u * j : XR_DEAD; };
    if (a.part != nullptr && strip == 0 && seg == 0) {     // rows no (strip, quantum) owns: zero (the buffer is sized for the LDS-tiled form's smallest tile)
        for (int rr = a.strips * a.nquanta; rr < a.rows_cap; ++rr)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                __builtin_amdgcn_raw_buffer_store_b128((xr_u4){0u, 0u, 0u, 0u}, psrc, poff(j) == XR_DEAD ? XR_DEAD : (unsigned)(rr * a.ld_part) * 4u + poff(j), 0, 0);
    }
    auto flush = [&](const int qi) {
        const unsigned prow_ = (unsigned)((strip * a.nquanta + qi) * a.ld_part) * 4u;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            xr_f4 v = ssum[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // fixed tree over the 16 pixels of the DPP row (rotations by 8, 4, 2, 1: every lane ends with the total): deterministic
                float t = v[i];
                asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(t));
                v[i] = t;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xr_u4, v), psrc, poff(j) == XR_DEAD ? XR_DEAD : prow_ + poff(j), 0, 0);
            ssum[j] = (xr_f4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto row = [&](auto emit_c, const int k, const int yo, const XRow& xc_, XRow& xn_) {
        constexpr bool EMIT = decltype(emit_c)::value;
        const int r = rbeg + k;
        load_row(xn_, r + 1);
        const float hr = (r >= 0 && r < a.H) ? hi : 0.f;
        xr_f4 ec[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            xr_f4 d = (xr_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) d = xr_mfma<T>(aw[j][c], xc_.m[c], d);
            ec[j] = xr_bn_act4<ACT>(d, es[j], eh[j], hr);
        }
        if constexpr (EMIT) {
            const unsigned opix = ((unsigned)yo * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_out * 2u;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                xr_f4 d = dh[j];
                if constexpr (K == 3) {
                    xr_row3(d, ring[j][0], tp[j][0], tp[j][1], tp[j][2]);
                    xr_row3(d, ring[j][1], tp[j][3], tp[j][4], tp[j][5]);
                    xr_row3(d, ec[j], tp[j][6], tp[j][7], tp[j][8]);
                } else if constexpr (!BC5) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) xr_row5(d, ring[j][q], tp[j][5 * q], tp[j][5 * q + 1], tp[j][5 * q + 2], tp[j][5 * q + 3], tp[j][5 * q + 4]);
                    xr_row5(d, ec[j], tp[j][20], tp[j][21], tp[j][22], tp[j][23], tp[j][24]);
                } else {
                    float cs[5][4];   // the five column sums
                    xr_bc5_row<0>(cs, ring[j][0], tp[j][0], tp[j][1]);
                    xr_bc5_row<1>(cs, ring[j][1], tp[j][0], tp[j][1]);
                    xr_bc5_row<2>(cs, ring[j][2], tp[j][0], tp[j][1]);
                    xr_bc5_row<3>(cs, ring[j][3], tp[j][0], tp[j][1]);
                    xr_bc5_row<4>(cs, ec[j], tp[j][0], tp[j][1]);
                    d = xr_bc5_finish(cs, d);
                }
                typedef T t4 __attribute__((ext_vector_type(4)));
                const xr_f4 v = xr_act4<ACT>(d, HI);
                const t4 o = __builtin_convertvector(v, t4);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xr_u2, o), osrc, ooff[j] == XR_DEAD ? XR_DEAD : opix + ooff[j], 0, 0);
                const xr_f4 stored = __builtin_convertvector(o, xr_f4);   // the squeeze sums what was STORED (rounded)
#pragma unroll
                for (int i = 0; i < 4; ++i) ssum[j][i] = __builtin_fmaf(stored[i], omask, ssum[j][i]);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q + 1 < K - 1; ++q) ring[j][q] = ring[j][q + 1];
            ring[j][K - 2] = ec[j];
        }
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    if constexpr (PAIR) {
        // A pair of output rows (y, y + 1) reads input rows rho = 0 .. K + 1 (counted from 2 y - pad_t): row y taps ky = rho, row
        // y + 1 taps ky = rho - 2.  The first K - 2 of them are the last rows of the previous pair (cr), four are new; the
        // accumulators of the pair (d2) take every row's contribution as soon as the row is expanded.
        xr_f4 d2[NT], cr[NT][K - 2];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < K - 2; ++q) cr[j][q] = (xr_f4){0.f, 0.f, 0.f, 0.f};
        auto prow = [&](auto ph_c, const int k, const int yo, const XRow& xc_, XRow& xn_) {
            constexpr int PH = decltype(ph_c)::value;   // 0: warm-up row; 1..4: the pair's new rows rho = K - 3 + PH
            const int r = rbeg + k;
            load_row(xn_, r + 1);
            const float hr = (r >= 0 && r < a.H) ? hi : 0.f;
            xr_f4 ec[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                xr_f4 d = (xr_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NC; ++c) d = xr_mfma<T>(aw[j][c], xc_.m[c], d);
                ec[j] = xr_bn_act4<ACT>(d, es[j], eh[j], hr);
            }
            if constexpr (PH >= 1) {
                constexpr int RHO = K - 3 + PH;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (PH == 1) {
                        d2[j] = dh[j];   // the BN shift is the first addend, in every lane
#pragma unroll
                        for (int c = 0; c < K - 2; ++c) {
                            xr_row_pair<K, false>(d2[j], cr[j][c], &tp[j][K * c]);
                            if (c >= 2) xr_row_pair<K, true>(d2[j], cr[j][c], &tp[j][K * (c >= 2 ? c - 2 : 0)]);
                        }
                    }
                    if constexpr (RHO < K) xr_row_pair<K, false>(d2[j], ec[j], &tp[j][K * (RHO < K ? RHO : 0)]);
                    if constexpr (RHO >= 2) xr_row_pair<K, true>(d2[j], ec[j], &tp[j][K * (RHO >= 2 ? RHO - 2 : 0)]);
                }
            }
            if constexpr (PH == 4) {
                const int yl = yo + podd;   // (lanes 8..15 hold the row below)
                const bool rlive = yl < yo1;
                const float om = rlive ? omask : 0.f;
                const unsigned opix = ((unsigned)yl * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_out * 2u;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    typedef T t4 __attribute__((ext_vector_type(4)));
                    const xr_f4 v = xr_act4<ACT>(d2[j], HI);
                    const t4 o = __builtin_convertvector(v, t4);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xr_u2, o), osrc, (ooff[j] == XR_DEAD || !rlive) ? XR_DEAD : opix + ooff[j], 0, 0);
                    const xr_f4 stored = __builtin_convertvector(o, xr_f4);   // the squeeze sums what was STORED (rounded)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ssum[j][i] = __builtin_fmaf(stored[i], om, ssum[j][i]);
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int q = 0; q + 1 < K - 2; ++q) cr[j][q] = cr[j][q + 1];
                cr[j][K - 3] = ec[j];
            }
        };
        constexpr std::integral_constant<int, 0> P0{};
        constexpr std::integral_constant<int, 1> P1{};
        constexpr std::integral_constant<int, 2> P2{};
        constexpr std::integral_constant<int, 3> P3{};
        constexpr std::integral_constant<int, 4> P4{};
        int k = 0;   // K - 2 warm-up rows (odd: afterwards the current row's operands are in xb)
#pragma unroll
        for (int q = 0; q < (K - 2) / 2; ++q) { prow(P0, k, 0, xa, xb); prow(P0, k + 1, 0, xb, xa); k += 2; }
        prow(P0, k, 0, xa, xb); k += 1;
#pragma nounroll
        for (int q0 = 0; q0 < nout; q0 += a.qrows) {     // quantum by quantum (a.qrows is even; without squeeze sums: one quantum = the segment)
            const int q1 = min(q0 + a.qrows, nout);
            for (int i = q0; i < q1; i += 2) {   // (an odd segment's last pair stores its even row only)
                prow(P1, k, 0, xb, xa);
                prow(P2, k + 1, 0, xa, xb);
                prow(P3, k + 2, 0, xb, xa);
                prow(P4, k + 3, yo0 + i, xa, xb);
                k += 4;
            }
            if (a.part != nullptr) flush((yo0 + q0) / a.qrows);
        }
    } else {
    // rows 0 .. K - S - 1 warm the ring up; then every output row takes S input rows, the last of which emits
    int k = 0;
    if constexpr ((K - S) % 2 == 0) {
#pragma unroll
        for (int q = 0; q < (K - S) / 2; ++q) { row(N, k, 0, xa, xb); row(N, k + 1, 0, xb, xa); k += 2; }
    } else {
#pragma unroll
        for (int q = 0; q < (K - S) / 2; ++q) { row(N, k, 0, xa, xb); row(N, k + 1, 0, xb, xa); k += 2; }
        row(N, k, 0, xa, xb); k += 1;
    }
#pragma nounroll
    for (int q0 = 0; q0 < nout; q0 += a.qrows) {     // quantum by quantum (a.qrows is even; without squeeze sums: one quantum = the segment)
        const int q1 = min(q0 + a.qrows, nout);
        if constexpr (S == 2) {   // (an odd warm-up: the current row's operands are in xb)
            for (int i = q0; i < q1; ++i) {
                row(N, k, 0, xb, xa);
                row(Y, k + 1, yo0 + i, xa, xb);
                k += 2;
            }
        } else {
            int i = q0;
            for (; i + 1 < q1; i += 2) {
                row(Y, k, yo0 + i, xa, xb);
                row(Y, k + 1, yo0 + i + 1, xb, xa);
                k += 2;
            }
            if (i < q1) row(Y, k, yo0 + i, xa, xb);     // (only the image's last quantum has an odd number of rows)
        }
        if (a.part != nullptr) flush((yo0 + q0) / a.qrows);
    }
    }
}

template <class T, int K, int S, int ACT, int NC, int NT>
static int launch_mbxr(const MbxrArgs& a0, int batch, int want_segs, hipStream_t s) {
    MbxrArgs a = a0;
    constexpr int NOUT = (16 - K) / S + 1;
    constexpr bool BC5 = K == 5 && NC >= 4;
    constexpr int EST = NT * (4 * NC + 16 + (BC5 ? 8 : 4 * K * K) + 4 * (K - 1) + 8) + 8 * NC + 48 + (BC5 ? 20 : 0);
    constexpr int MW = EST <= 120 ? 4 : EST <= 160 ? 3 : EST <= 250 ? 2 : 1;
    a.strips = (a.Wo + NOUT - 1) / NOUT;
    a.groups = (a.T + NT - 1) / NT;
    const int walks = batch * a.strips * a.groups;
    int segs = (3 * 1024 + walks - 1) / walks;
    const int max_segs = (a.Ho + 5) / 6;
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    if (want_segs > 0) segs = want_segs < a.Ho ? want_segs : a.Ho;
    a.seg_rows = (a.Ho + segs - 1) / segs;
    if (S == 2 && !BC5) a.seg_rows += a.seg_rows & 1;   // (output rows are processed in pairs)
    a.qrows = a.seg_rows + (a.seg_rows & 1); a.nquanta = 0;     // (no sums: the quantum loop runs once)
    if (a.part != nullptr) {
        // one row of the partial-sum buffer per (strip, quantum): the smallest even quantum from 4 rows on that fits the buffer - a
        // function of the shape alone; a segment is a whole number of quanta
        int q = 4;
        while (a.strips * ((a.Ho + q - 1) / q) > a.rows_cap && q < a.Ho + 1) q += 2;
        YR_REQUIRE(a.strips * ((a.Ho + q - 1) / q) <= a.rows_cap, "mbxr: %d strips exceed the %d rows of the partial-sum buffer", a.strips, a.rows_cap);
        a.qrows = q; a.nquanta = (a.Ho + q - 1) / q;
        a.seg_rows = (a.seg_rows + q - 1) / q * q;
    }
    a.segs = (a.Ho + a.seg_rows - 1) / a.seg_rows;
    a.nwaves = batch * a.strips * a.segs * a.groups;
    static char nm[64];
    static const int nm_len = snprintf(nm, sizeof(nm), "mbxr_kernel<%s,%d,%d,%d,%d,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, S, ACT, NC, NT, MW);
    (void)nm_len;
    yr_note_kernel(nm);
    hipLaunchKernelGGL((mbxr_kernel<T, K, S, ACT, NC, NT, MW>), dim3((unsigned)((a.nwaves + 3) / 4)), dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T, int K, int S, int ACT>
static int launch_mbxr_nc(const MbxrArgs& a, int batch, int segs, hipStream_t s) {
    constexpr int NT = K == 3 ? 2 : 1;
    switch (a.KP / 32) {
        case 1: return launch_mbxr<T, K, S, ACT, 1, NT>(a, batch, segs, s);
        case 2: return launch_mbxr<T, K, S, ACT, 2, NT>(a, batch, segs, s);
        case 3: return launch_mbxr<T, K, S, ACT, 3, NT>(a, batch, segs, s);
        case 4: return launch_mbxr<T, K, S, ACT, 4, 2>(a, batch, segs, s);   // (5x5: the broadcast-tap form, two tiles per wave)
        default: yr_set_error("mbxr: %d input channels are not built", a.Cin); return YR_ERR_ARG;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The network entry of the squeeze-excite EfficientNets in the same form (YR_OP_STEMBLOCK without a projection: stem Conv2D
// 3x3 stride 2 + BN + act -> stage-1 depthwise 3x3 + BN + act, the map stored, its squeeze sums per 14 x 14 tile; reference
// code/yolo3/efficientnet.py:636-645 and the first MBConvBlock, :467-536 with expand_ratio 1).  The stem is ONE
// v_mfma_f32_16x16x32 per 16 stem channels: K = 27 taps in the order of stemblock_h.hip - k group g < 3 = image row 2y + g, the
// first 8 of its 9 (kx, c) values = 32 contiguous bytes of the image; group 3 = value 8 of the three rows + zeros - the image
// rounded to the 16-bit type like every MFMA operand.  Parameters as the float32-pipe kernel takes them (stemblock.hip: packed
// per channel pair, BN scale folded in); the A fragment is gathered (and rounded) from them once per wave.  Even image sizes,
// float32 images; a wave = one 14 x 14 output tile (strip of 14 columns x 14 rows: one row of the squeeze-sum buffer), all C1
// channels (NT = 2 | 3 tiles).  Was: stemblock_kernel on the float32 pipe, 0.52 ms per 128 images at 416 (B0).
struct StemxrArgs {
    const float* img; void* out; const float* ws; const float* wd; float* part;
    int Hi, Wi, Ho, Wo, C1, ld_out, ld_part, tiles_x, tiles_y, nwaves;
};

template <class T, int NT, int ACT>
__global__ __launch_bounds__(256, 2) void stemxr_kernel(StemxrArgs a) {
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    int gw = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (gw >= a.nwaves) return;
    const int tx = gw % a.tiles_x; gw /= a.tiles_x;
    const int ty = gw % a.tiles_y;
    const int b = gw / a.tiles_y;
    const int yo0 = 14 * ty, yo1 = min(yo0 + 14, a.Ho);
    const int xs = 14 * tx - 1 + px;                       // this lane's stem-output column (= depthwise input column)
    const int xsc = min(max(xs, 0), a.Wo - 1);
    constexpr float HI = ACT == 0 ? 6.f : 1.f;
    const float hi = (xs >= 0 && xs < a.Wo) ? HI : 0.f;
    const int xo = 14 * tx + px - 1;
    const bool out_lane = px >= 1 && px <= 14 && xo < a.Wo;
    const float omask = out_lane ? 1.f : 0.f;
    typedef T t4 __attribute__((ext_vector_type(4)));
    typedef T t2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));

    // ---- stationary: stem A fragments (gathered from the pair-packed float32 rows, rounded), shifts, depthwise taps
    xr_u4 aw[NT];
    xr_f4 sh[NT], dh[NT], tp[NT][9], ssum[NT];
    unsigned ooff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        {   // A: lane (m = px, k group mg): W[16 j + m][k], k = 9 mg + i (mg < 3, i < 8) | taps 8, 17, 26 (mg == 3)
            const int ch = 16 * j + px;
            const float* wr = a.ws + (size_t)(ch >> 1) * 58 + (ch & 1);
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = mg < 3 ? 9 * mg + i : (i < 3 ? 9 * i + 8 : 0);
                v[i] = (ch < a.C1 && (mg < 3 || i < 3)) ? wr[2 * k] : 0.f;
            }
            unsigned u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){v[2 * i], v[2 * i + 1]}, t2));
            aw[j] = (xr_u4){u[0], u[1], u[2], u[3]};
        }
        const int ch = 16 * j + 4 * mg;                    // this lane's 4 channels of tile j in the MFMA result
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = ch + i;
            const bool live = c < a.C1;
            const float* wr = a.ws + (size_t)(c >> 1) * 58 + (c & 1);
            const float* dr = a.wd + (size_t)(c >> 1) * 22 + (c & 1);
            sh[j][i] = live ? wr[56] : 0.f;
            dh[j][i] = live ? dr[20] : 0.f;
#pragma unroll
            for (int q = 0; q < 9; ++q) tp[j][q][i] = live ? dr[2 * q] : 0.f;
        }
        ssum[j] = (xr_f4){0.f, 0.f, 0.f, 0.f};
        ooff[j] = (out_lane && ch < a.C1) ? (unsigned)ch * 2u : XR_DEAD;
    }
    const xr_rsrc isrc = xr_make_rsrc(a.img + (size_t)b * a.Hi * a.Wi * 3, (unsigned)(a.Hi * a.Wi * 3) * 4u);
    const xr_rsrc osrc = xr_make_rsrc(reinterpret_cast<T*>(a.out) + (size_t)b * a.Ho * a.Wo * a.ld_out, (unsigned)(a.Ho * a.Wo * a.ld_out) * 2u);
    const unsigned irow = (unsigned)a.Wi * 12u;            // bytes per image row
    const unsigned icol = (unsigned)(2 * xsc) * 12u;       // the window's first byte within a row (even sizes: no left padding)
    const bool last_col = 2 * xsc + 2 >= a.Wi;             // kx = 2 lies beyond the row: values 6, 7 (k groups 0..2) and group 3 are padding
    // the B operand of stem row ys for this lane: 8 image values of its k group, rounded
    auto load_b = [&](int ys) -> xr_u4 {
        const int ysc = min(max(ys, 0), a.Ho - 1);
        float v[8];
        if (mg < 3) {
            const int iy = 2 * ysc + mg;
            const unsigned off = iy < a.Hi ? (unsigned)iy * irow + icol : XR_DEAD;
            const xr_f4 lo = __builtin_bit_cast(xr_f4, __builtin_amdgcn_raw_buffer_load_b128(isrc, off, 0, 0));
            const xr_f4 hi4 = __builtin_bit_cast(xr_f4, __builtin_amdgcn_raw_buffer_load_b128(isrc, off == XR_DEAD ? XR_DEAD : off + 16u, 0, 0));
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi4[0]; v[5] = hi4[1];
            v[6] = last_col ? 0.f : hi4[2]; v[7] = last_col ? 0.f : hi4[3];
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int iy = 2 * ysc + i;
                const unsigned off = (iy < a.Hi && !last_col) ? (unsigned)iy * irow + icol + 32u : XR_DEAD;
                v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(isrc, off, 0, 0));
            }
            v[3] = v[4] = v[5] = v[6] = v[7] = 0.f;
        }
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){v[2 * i], v[2 * i + 1]}, t2));
        return (xr_u4){u[0], u[1], u[2], u[3]};
    };
    const int rbeg = yo0 - 1, nout = yo1 - yo0;
    xr_f4 ring[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j) { ring[j][0] = (xr_f4){0.f, 0.f, 0.f, 0.f}; ring[j][1] = ring[j][0]; }
    xr_u4 bcur = load_b(rbeg);
    for (int k = 0; k < nout + 2; ++k) {
        const int ys = rbeg + k;
        const xr_u4 bnext = load_b(ys + 1);
        const float hr = (ys >= 0 && ys < a.Ho) ? hi : 0.f;
        xr_f4 ec[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const xr_f4 d = xr_mfma<T>(aw[j], bcur, sh[j]);       // (the BN shift is the accumulator's initial value)
            ec[j] = xr_act4<ACT>(d, hr);
        }
        if (k >= 2) {
            const int yo = yo0 + k - 2;
            const unsigned opix = ((unsigned)yo * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_out * 2u;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                xr_f4 d = dh[j];
                xr_row3(d, ring[j][0], tp[j][0], tp[j][1], tp[j][2]);
                xr_row3(d, ring[j][1], tp[j][3], tp[j][4], tp[j][5]);
                xr_row3(d, ec[j], tp[j][6], tp[j][7], tp[j][8]);
                const xr_f4 v = xr_act4<ACT>(d, HI);
                const t4 o = __builtin_convertvector(v, t4);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xr_u2, o), osrc, ooff[j] == XR_DEAD ? XR_DEAD : opix + ooff[j], 0, 0);
                const xr_f4 stored = __builtin_convertvector(o, xr_f4);
#pragma unroll
                for (int i = 0; i < 4; ++i) ssum[j][i] = __builtin_fmaf(stored[i], omask, ssum[j][i]);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) { ring[j][0] = ring[j][1]; ring[j][1] = ec[j]; }
        bcur = bnext;
    }
    if (a.part != nullptr) {
        const int prow = ty * a.tiles_x + tx;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            xr_f4 v = ssum[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] += __shfl_xor(v[i], 8, 16);
                v[i] += __shfl_xor(v[i], 4, 16);
                v[i] += __shfl_xor(v[i], 2, 16);
                v[i] += __shfl_xor(v[i], 1, 16);
            }
            const int ch = 16 * j + 4 * mg;
            if (px == 0 && ch < a.C1) {
                float* p = a.part + ((size_t)b * a.tiles_x * a.tiles_y + prow) * a.ld_part + ch;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (ch + i < a.ld_part) p[i] = v[i];
            }
        }
    }
}

// whether the matrix-pipe register-chained entry is built for this STEMBLOCK op (no projection, 16-bit output, float32 image,
// even sizes, C1 <= 48, relu6 / swish)
bool yr_stemxr_takes(const yr_op& op) {
    return op.kind == YR_OP_STEMBLOCK && op.b1 == nullptr && (op.dtype == YR_BF16 || op.dtype == YR_F16) && op.out_dtype == op.dtype &&
           op.nsrc == 1 && op.src[0].dtype == YR_F32 && op.src[0].c == 3 && op.src[0].ld == 3 && op.src[0].h % 2 == 0 && op.src[0].w % 2 == 0 &&
           op.se_reduced <= 48 && op.se_reduced % 4 == 0 && op.cout == op.se_reduced && op.out_ld % 4 == 0 &&
           (op.act == YR_ACT_RELU6 || op.act == YR_ACT_SWISH);
}

template <class T>
static int launch_stemxr_t(const yr_op& op, int batch, hipStream_t s) {
    const yr_src& in = op.src[0];
    StemxrArgs a;
    a.img = (const float*)in.ptr; a.out = op.out; a.ws = op.wgt; a.wd = op.wgt2;
    a.part = const_cast<float*>(op.gate); a.ld_part = op.gate ? op.gate_ld : 0;
    a.Hi = in.h; a.Wi = in.w; a.Ho = in.h / 2; a.Wo = in.w / 2; a.C1 = op.se_reduced; a.ld_out = op.out_ld;
    a.tiles_x = (a.Wo + 13) / 14; a.tiles_y = (a.Ho + 13) / 14;
    a.nwaves = batch * a.tiles_x * a.tiles_y;
    const int nt = (a.C1 + 15) / 16, act = op.act == YR_ACT_RELU6 ? 0 : 1;
    static char nm[48];
    snprintf(nm, sizeof(nm), "stemxr_kernel<%s,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), nt, act);
    yr_note_kernel(nm);
    const dim3 grid((unsigned)((a.nwaves + 3) / 4));
#define SX_GO(NTV)                                                                                   \
    do {                                                                                            \
        if (act == 0) hipLaunchKernelGGL((stemxr_kernel<T, NTV, 0>), grid, dim3(256), 0, s, a);    \
        else hipLaunchKernelGGL((stemxr_kernel<T, NTV, 1>), grid, dim3(256), 0, s, a);             \
    } while (0)
    if (nt == 2) SX_GO(2); else if (nt == 3) SX_GO(3); else SX_GO(1);
#undef SX_GO
    YR_LAUNCH_CHECK();
    return YR_OK;
}

int yr_launch_stemxr(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(yr_stemxr_takes(op), "stemxr: the matrix-pipe entry is not built for this op");
    YR_REQUIRE(op.src[0].ptr && op.out && op.wgt && op.wgt2 && op.h == op.src[0].h / 2 && op.w == op.src[0].w / 2 && op.out_ld >= op.cout, "stemxr: bad arguments");
    YR_REQUIRE(op.gate == nullptr || (op.gate_ld >= op.cout && ((uintptr_t)op.gate % 4) == 0), "stemxr: bad squeeze-sum buffer");
    return op.dtype == YR_BF16 ? launch_stemxr_t<yr_bf16>(op, batch, s) : launch_stemxr_t<yr_f16>(op, batch, s);
}

// ------------------------------------------------------------------------------------------------------------------------
// The network entry of the 16-bit plans WITH the projection (stem 3x3 s2 + BN + act -> depthwise 3x3 + BN + act -> project 1x1 +
// BN; MobileNetV2 Conv1 + expanded_conv [3P], EfficientNet-lite stem + stage 1) in the register-chained form: stemxr_kernel's walk,
// and the depthwise results of the wave's NT tiles, rounded, are the B operands of v_mfma_f32_16x16x16 projection steps whose
// accumulators never leave the wave (one wave owns all channels of its 14 x 14 tile: no LDS, no barrier).  Parameters in the
// layout of stemblock_h.hip (the compiler's matrix-pipe layout, see launch_stemblock_h_t); float32 image.
struct StemxpArgs {
    const float* img; void* out; const void* ws; const float* ssc; const float* ssh; const float* wd; const void* wp; const float* bp;
    int Hi, Wi, Ho, Wo, C1, C1P, Cout, COP, ld_out, tiles_x, tiles_y, nwaves;
};

template <class T, int NT, int TO, int ACT>
__global__ __launch_bounds__(256, 2) void stemxp_kernel(StemxpArgs a) {
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    int gw = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (gw >= a.nwaves) return;
    const int tx = gw % a.tiles_x; gw /= a.tiles_x;
    const int ty = gw % a.tiles_y;
    const int b = gw / a.tiles_y;
    const int yo0 = 14 * ty, yo1 = min(yo0 + 14, a.Ho);
    const int xs = 14 * tx - 1 + px;                       // this lane's stem-output column (= depthwise input column)
    const int xsc = min(max(xs, 0), a.Wo - 1);
    constexpr float HI = ACT == 0 ? 6.f : 1.f;
    const float hi = (xs >= 0 && xs < a.Wo) ? HI : 0.f;
    const int xo = 14 * tx + px - 1;
    const bool out_lane = px >= 1 && px <= 14 && xo < a.Wo;
    typedef T t4 __attribute__((ext_vector_type(4)));
    typedef T t2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));

    // ---- stationary: stem A fragments, stem BN rows, depthwise taps (times the BN scale) and shift, projection A fragments
    xr_u4 aw[NT];
    xr_u2 wpf[NT][TO];
    xr_f4 es[NT], eh[NT], dh[NT], tp[NT][9];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        aw[j] = *reinterpret_cast<const xr_u4*>(reinterpret_cast<const char*>(a.ws) + ((size_t)(16 * j + px) * 32 + 8 * mg) * 2);
        const int ch = 16 * j + 4 * mg;                    // this lane's 4 channels of tile j in the MFMA result (< C1P: zero padded)
        es[j] = *reinterpret_cast<const xr_f4*>(a.ssc + ch);
        eh[j] = *reinterpret_cast<const xr_f4*>(a.ssh + ch);
#pragma unroll
        for (int q = 0; q < 9; ++q) tp[j][q] = *reinterpret_cast<const xr_f4*>(a.wd + (size_t)q * a.C1P + ch);
        dh[j] = *reinterpret_cast<const xr_f4*>(a.wd + (size_t)9 * a.C1P + ch);
#pragma unroll
        for (int t = 0; t < TO; ++t)
            wpf[j][t] = *reinterpret_cast<const xr_u2*>(reinterpret_cast<const char*>(a.wp) + ((size_t)(16 * t + px) * a.C1P + ch) * 2);
    }
    xr_f4 psc[TO], psh[TO];
    unsigned ooff[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        const int co = 16 * t + 4 * mg;
        psc[t] = *reinterpret_cast<const xr_f4*>(a.bp + co);
        psh[t] = *reinterpret_cast<const xr_f4*>(a.bp + a.COP + co);
        ooff[t] = (out_lane && co < a.Cout) ? (unsigned)co * 2u : XR_DEAD;
    }
    const xr_rsrc isrc = xr_make_rsrc(a.img + (size_t)b * a.Hi * a.Wi * 3, (unsigned)(a.Hi * a.Wi * 3) * 4u);
    const xr_rsrc osrc = xr_make_rsrc(reinterpret_cast<T*>(a.out) + (size_t)b * a.Ho * a.Wo * a.ld_out, (unsigned)(a.Ho * a.Wo * a.ld_out) * 2u);
    const unsigned irow = (unsigned)a.Wi * 12u;            // bytes per image row
    const unsigned icol = (unsigned)(2 * xsc) * 12u;       // the window's first byte within a row (even sizes: no left padding)
    const bool last_col = 2 * xsc + 2 >= a.Wi;             // kx = 2 lies beyond the row: values 6, 7 (k groups 0..2) and group 3 are padding
    auto load_b = [&](int ys) -> xr_u4 {                   // (as stemxr_kernel: 8 image values of the lane's k group, rounded)
        const int ysc = min(max(ys, 0), a.Ho - 1);
        float v[8];
        if (mg < 3) {
            const int iy = 2 * ysc + mg;
            const unsigned off = iy < a.Hi ? (unsigned)iy * irow + icol : XR_DEAD;
            const xr_f4 lo = __builtin_bit_cast(xr_f4, __builtin_amdgcn_raw_buffer_load_b128(isrc, off, 0, 0));
            const xr_f4 hi4 = __builtin_bit_cast(xr_f4, __builtin_amdgcn_raw_buffer_load_b128(isrc, off == XR_DEAD ? XR_DEAD : off + 16u, 0, 0));
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi4[0]; v[5] = hi4[1];
            v[6] = last_col ? 0.f : hi4[2]; v[7] = last_col ? 0.f : hi4[3];
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int iy = 2 * ysc + i;
                const unsigned off = (iy < a.Hi && !last_col) ? (unsigned)iy * irow + icol + 32u : XR_DEAD;
                v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(isrc, off, 0, 0));
            }
            v[3] = v[4] = v[5] = v[6] = v[7] = 0.f;
        }
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){v[2 * i], v[2 * i + 1]}, t2));
        return (xr_u4){u[0], u[1], u[2], u[3]};
    };
    const int rbeg = yo0 - 1, nout = yo1 - yo0;
    xr_f4 ring[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j) { ring[j][0] = (xr_f4){0.f, 0.f, 0.f, 0.f}; ring[j][1] = ring[j][0]; }
    xr_u4 bcur = load_b(rbeg);
    for (int k = 0; k < nout + 2; ++k) {
        const int ys = rbeg + k;
        const xr_u4 bnext = load_b(ys + 1);
        const float hr = (ys >= 0 && ys < a.Ho) ? hi : 0.f;
        xr_f4 ec[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) ec[j] = xr_bn_act4<ACT>(xr_mfma<T>(aw[j], bcur, (xr_f4){0.f, 0.f, 0.f, 0.f}), es[j], eh[j], hr);
        if (k >= 2) {
            const int yo = yo0 + k - 2;
            xr_f4 P[TO];
#pragma unroll
            for (int t = 0; t < TO; ++t) P[t] = (xr_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                xr_f4 d = dh[j];
                xr_row3(d, ring[j][0], tp[j][0], tp[j][1], tp[j][2]);
                xr_row3(d, ring[j][1], tp[j][3], tp[j][4], tp[j][5]);
                xr_row3(d, ec[j], tp[j][6], tp[j][7], tp[j][8]);
                const xr_u2 bop = __builtin_bit_cast(xr_u2, __builtin_convertvector(xr_act4<ACT>(d, HI), t4));   // rounded: the projection's operand type
#pragma unroll
                for (int t = 0; t < TO; ++t) P[t] = xr_mfma16<T>(wpf[j][t], bop, P[t]);
            }
            const unsigned opix = ((unsigned)yo * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_out * 2u;
#pragma unroll
            for (int t = 0; t < TO; ++t) {
                xr_f4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(P[t][i], psc[t][i], psh[t][i]);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xr_u2, __builtin_convertvector(v, t4)), osrc, ooff[t] == XR_DEAD ? XR_DEAD : opix + ooff[t], 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) { ring[j][0] = ring[j][1]; ring[j][1] = ec[j]; }
        bcur = bnext;
    }
}

// whether this matrix-pipe-layout STEMBLOCK op (stem + depthwise + projection) runs on stemxp_kernel: float32 image of even sizes,
// at most 48 stem channels and 32 outputs.  By SHAPE (stems of more than 32 channels; YOLORET_STEMXP=1: every stem it is built
// for, =0: none) - never by the tuner.
bool yr_stemxp_takes(const yr_op& op) {
    static const int mode = getenv("YOLORET_STEMXP") ? atoi(getenv("YOLORET_STEMXP")) : -1;
    if (mode == 0) return false;
    return op.kind == YR_OP_STEMBLOCK && op.scale != nullptr && op.b1 != nullptr && (op.dtype == YR_BF16 || op.dtype == YR_F16) && op.out_dtype == op.dtype &&
           op.nsrc == 1 && op.src[0].dtype == YR_F32 && op.src[0].c == 3 && op.src[0].ld == 3 && op.src[0].h % 2 == 0 && op.src[0].w % 2 == 0 &&
           op.se_reduced <= 48 && (mode == 1 || op.se_reduced > 32) && op.cout <= 32 && op.out_ld % 4 == 0 && (op.act == YR_ACT_RELU6 || op.act == YR_ACT_SWISH);
}

template <class T>
static int launch_stemxp_t(const yr_op& op, int batch, hipStream_t s) {
    const yr_src& in = op.src[0];
    StemxpArgs a;
    a.img = (const float*)in.ptr; a.out = op.out; a.ws = op.wgt; a.ssc = op.scale; a.ssh = op.shift; a.wd = op.wgt2; a.wp = op.b1; a.bp = op.b2;
    a.Hi = in.h; a.Wi = in.w; a.Ho = in.h / 2; a.Wo = in.w / 2; a.C1 = op.se_reduced; a.C1P = yr_round_up(op.se_reduced, 32);
    a.Cout = op.cout; a.COP = yr_round_up(op.cout, 16); a.ld_out = op.out_ld;
    a.tiles_x = (a.Wo + 13) / 14; a.tiles_y = (a.Ho + 13) / 14;
    a.nwaves = batch * a.tiles_x * a.tiles_y;
    const int nt = (a.C1 + 15) / 16, to = a.COP / 16, act = op.act == YR_ACT_RELU6 ? 0 : 1;
    static char nm[48];
    snprintf(nm, sizeof(nm), "stemxp_kernel<%s,%d,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), nt, to, act);
    yr_note_kernel(nm);
    const dim3 grid((unsigned)((a.nwaves + 3) / 4));
#define SP_GO(NTV, TOV)                                                                                   \
    do {                                                                                                 \
        if (act == 0) hipLaunchKernelGGL((stemxp_kernel<T, NTV, TOV, 0>), grid, dim3(256), 0, s, a);    \
        else hipLaunchKernelGGL((stemxp_kernel<T, NTV, TOV, 1>), grid, dim3(256), 0, s, a);             \
    } while (0)
    if (to == 1) { if (nt == 3) SP_GO(3, 1); else if (nt == 2) SP_GO(2, 1); else SP_GO(1, 1); }
    else { if (nt == 3) SP_GO(3, 2); else if (nt == 2) SP_GO(2, 2); else SP_GO(1, 2); }
#undef SP_GO
    YR_LAUNCH_CHECK();
    return YR_OK;
}

int yr_launch_stemxp(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(yr_stemxp_takes(op), "stemxp: the register-chained entry with projection is not built for this op");
    YR_REQUIRE(op.src[0].ptr && op.out && op.wgt && op.wgt2 && op.shift && op.b2 && op.h == op.src[0].h / 2 && op.w == op.src[0].w / 2 && op.out_ld >= op.cout && op.k == 3 && op.stride == 2,
               "stemxp: bad arguments");
    return op.dtype == YR_BF16 ? launch_stemxp_t<yr_bf16>(op, batch, s) : launch_stemxp_t<yr_f16>(op, batch, s);
}

// ------------------------------------------------------------------------------------------------------------------------
// YR_OP_MBH in the same form: the WHOLE block - expand -> depthwise 3x3 -> project 1x1 + BN (+ residual) - for blocks of at
// most 16 expanded tiles (the network fronts: MobileNetV2 block_1..6, EfficientNet-lite stage 2, lite0 stage 4 entry).  A
// workgroup's NW waves share one strip segment; wave w owns the expanded tile PAIR (2w, 2w + 1) = one 32-deep k step of the
// projection: its depthwise results, rounded to the 16-bit type, ARE the projection MFMA's B operand (the lane's 4 channels
// of tile 2w and of tile 2w + 1 = the 8 k values of its k group; the weight fragment is gathered in the same order), the
// partial projections of an output row meet in LDS (one barrier per output row, two buffers).  Same parameters as mbh.hip.
struct MbhrArgs {
    const void* x; void* out; const void* we; const float* prm; const void* wp; const float* sp; const float* hp;
    int H, W, Ho, Wo, Cin, CexpP, Cout, ld_in, ld_out, KP, pad_t, pad_l, strips, segs, seg_rows, T, has_res;
};

template <class T, int S, int ACT, int NC, int TO, int NW, int MW>
__global__ __launch_bounds__(64 * NW, MW) void mbhr_kernel(MbhrArgs a) {
    constexpr int K = 3, KK = 9, NOUT = (16 - K) / S + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    xr_f4* red = reinterpret_cast<xr_f4*>(lds);            // [2][NW][TO][64]
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int seg = bid % a.segs; bid /= a.segs;
    const int strip = bid % a.strips;
    const int b = bid / a.strips;
    const int yo0 = seg * a.seg_rows, yo1 = min(yo0 + a.seg_rows, a.Ho);
    const int xin = S * NOUT * strip - a.pad_l + px;
    const int xc = min(max(xin, 0), a.W - 1);
    constexpr float HI = ACT == 0 ? 6.f : 1.f;
    const float hi = (xin >= 0 && xin < a.W) ? HI : 0.f;
    const int jo = (px - 1) / S, xo = NOUT * strip + jo;
    const bool out_lane = px >= 1 && (px - 1) % S == 0 && jo < NOUT && xo < a.Wo;

    // ---- stationary: the wave's two expanded tiles (a tile beyond T: all-zero parameters -> its depthwise result is act(0) = 0)
    xr_u4 aw[2][NC], wpf[TO];
    xr_f4 es[2], eh[2], dh[2], tp[2][KK];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = 2 * w + j;
        const bool live = t < a.T;
        const int tc = live ? t : 0, ch = 16 * tc + 4 * mg;
        const char* wrow = reinterpret_cast<const char*>(a.we) + ((size_t)(16 * tc + px) * a.KP + 8 * mg) * 2;
        const xr_f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) aw[j][c] = live ? *reinterpret_cast<const xr_u4*>(wrow + 64 * c) : (xr_u4){0u, 0u, 0u, 0u};
        const xr_f4 dsc = *reinterpret_cast<const xr_f4*>(a.prm + (size_t)KK * a.CexpP + ch);
        dh[j] = live ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(KK + 1) * a.CexpP + ch) : z;
        es[j] = live ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(KK + 2) * a.CexpP + ch) : z;
        eh[j] = live ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(KK + 3) * a.CexpP + ch) : z;
#pragma unroll
        for (int q = 0; q < KK; ++q) tp[j][q] = live ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)q * a.CexpP + ch) * dsc : z;
    }
#pragma unroll
    for (int t = 0; t < TO; ++t) {   // project A fragment of cout tile t for this wave's k step: W[16 t + m][16 (2w) + 4 g ..] ++ W[..][16 (2w + 1) + 4 g ..]
        const int co = 16 * t + px;
        const char* prow = reinterpret_cast<const char*>(a.wp) + ((size_t)(co < a.Cout ? co : 0) * a.CexpP + 32 * w + 4 * mg) * 2;
        xr_u2 lo = *reinterpret_cast<const xr_u2*>(prow), hi2 = *reinterpret_cast<const xr_u2*>(prow + 32);
        if (co >= a.Cout) { lo = (xr_u2){0u, 0u}; hi2 = lo; }
        wpf[t] = (xr_u4){lo[0], lo[1], hi2[0], hi2[1]};
    }
    // the cout tile this wave finishes (t = w, if w < TO): project BN rows
    const int fco = 16 * w + 4 * mg;
    const bool flive = w < TO && out_lane && fco < a.Cout;
    const bool rlive = flive && a.has_res;
    xr_f4 fsc = {0.f, 0.f, 0.f, 0.f}, fsh = fsc;
    if (w < TO && fco < a.Cout) { fsc = *reinterpret_cast<const xr_f4*>(a.sp + fco); fsh = *reinterpret_cast<const xr_f4*>(a.hp + fco); }

    const xr_rsrc xsrc = xr_make_rsrc(reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.ld_in, (unsigned)(a.H * a.W * a.ld_in) * 2u);
    const xr_rsrc osrc = xr_make_rsrc(reinterpret_cast<T*>(a.out) + (size_t)b * a.Ho * a.Wo * a.ld_out, (unsigned)(a.Ho * a.Wo * a.ld_out) * 2u);
    const int rbeg = S * yo0 - a.pad_t, nout = yo1 - yo0;
    unsigned xoff[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) xoff[c] = (32 * c + 8 * mg < a.Cin) ? ((unsigned)xc * (unsigned)a.ld_in + 32u * c + 8u * mg) * 2u : XR_DEAD;
    const unsigned xrow = (unsigned)(a.W * a.ld_in) * 2u;
    struct XRow { xr_u4 m[NC]; };
    XRow xa, xb;
    auto load_row = [&](XRow& x, int r) {
        const unsigned so = (unsigned)min(max(r, 0), a.H - 1) * xrow;
#pragma unroll
        for (int c = 0; c < NC; ++c) x.m[c] = __builtin_bit_cast(xr_u4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[c], so, 0));
    };
    load_row(xa, rbeg);
    xr_f4 ring[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { ring[j][0] = (xr_f4){0.f, 0.f, 0.f, 0.f}; ring[j][1] = ring[j][0]; }
    int buf = 0;
    typedef T t4 __attribute__((ext_vector_type(4)));

    auto row = [&](auto emit_c, const int k, const int yo, const XRow& xc_, XRow& xn_) {
        constexpr bool EMIT = decltype(emit_c)::value;
        const int r = rbeg + k;
        load_row(xn_, r + 1);
        xr_u2 resv = {0u, 0u};
        if constexpr (EMIT)   // UNCONDITIONAL (a dead offset reads zeros without a residual): a load under a branch makes every later wait vmcnt(0)
            resv = __builtin_bit_cast(xr_u2, __builtin_amdgcn_raw_buffer_load_b64(xsrc, rlive ? (((unsigned)yo * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_in + (unsigned)fco) * 2u : XR_DEAD, 0, 0));
        __builtin_amdgcn_sched_barrier(0);   // (both loads are issued HERE: left alone the scheduler sinks the residual load to its use behind the barrier)
        const float hr = (r >= 0 && r < a.H) ? hi : 0.f;
        xr_f4 ec[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            xr_f4 d = (xr_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) d = xr_mfma<T>(aw[j][c], xc_.m[c], d);
            ec[j] = xr_bn_act4<ACT>(d, es[j], eh[j], hr);
        }
        if constexpr (EMIT) {
            t4 dq[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                xr_f4 d = dh[j];
                xr_row3(d, ring[j][0], tp[j][0], tp[j][1], tp[j][2]);
                xr_row3(d, ring[j][1], tp[j][3], tp[j][4], tp[j][5]);
                xr_row3(d, ec[j], tp[j][6], tp[j][7], tp[j][8]);
                const xr_f4 v = xr_act4<ACT>(d, HI);
                dq[j] = __builtin_convertvector(v, t4);      // rounded: the projection's operand type
            }
            const xr_u2 b0 = __builtin_bit_cast(xr_u2, dq[0]), b1 = __builtin_bit_cast(xr_u2, dq[1]);
            const xr_u4 bop = {b0[0], b0[1], b1[0], b1[1]};
            xr_f4* rb = red + buf * (NW * TO * 64);
#pragma unroll
            for (int t = 0; t < TO; ++t) rb[(w * TO + t) * 64 + lane] = xr_mfma<T>(wpf[t], bop, (xr_f4){0.f, 0.f, 0.f, 0.f});
            __syncthreads();
            xr_f4 acc = {0.f, 0.f, 0.f, 0.f};
            if (w < TO) {
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) acc += rb[(ww * TO + w) * 64 + lane];
            }
            xr_f4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(acc[i], fsc[i], fsh[i]);
            v += __builtin_convertvector(__builtin_bit_cast(t4, resv), xr_f4);   // (zeros without a residual)
            const t4 o = __builtin_convertvector(v, t4);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xr_u2, o), osrc,
                                                  flive ? (((unsigned)yo * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_out + (unsigned)fco) * 2u : XR_DEAD, 0, 0);
            buf ^= 1;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { ring[j][0] = ring[j][1]; ring[j][1] = ec[j]; }
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

template <class T, int S, int ACT, int NC, int TO, int NW>
static int launch_mbhr(const MbhrArgs& a0, int batch, int want_segs, hipStream_t s) {
    MbhrArgs a = a0;
    constexpr int NOUT = (16 - 3) / S + 1;
    constexpr int MW = (NW <= 6 && S == 2) ? 3 : 2;   // (measured allocations: 161-180 registers for the one-chunk blocks - three waves per SIMD take 168)
    static_assert(NW <= 4 * MW, "the workgroup's waves must fit one CU");
    a.strips = (a.Wo + NOUT - 1) / NOUT;
    const int walks = batch * a.strips;
    int segs = (2 * 1024 + walks * NW - 1) / (walks * NW);
    const int max_segs = (a.Ho + 5) / 6;
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    if (want_segs > 0) segs = want_segs < a.Ho ? want_segs : a.Ho;
    a.seg_rows = (a.Ho + segs - 1) / segs;
    a.segs = (a.Ho + a.seg_rows - 1) / a.seg_rows;
    const size_t lds = (size_t)2 * NW * TO * 64 * 16;
    static char nm[64];
    static const int nm_len = snprintf(nm, sizeof(nm), "mbhr_kernel<%s,%d,%d,%d,%d,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), S, ACT, NC, TO, NW, MW);
    (void)nm_len;
    yr_note_kernel(nm);
    auto kern = mbhr_kernel<T, S, ACT, NC, TO, NW, MW>;
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
// The whole block once more, for the shapes the tile-PAIR form above does not fit: 5x5 depthwise kernels, and expanded widths
// whose tile pairs would make a five-wave workgroup (144 channels = 9 tiles).  Wave w owns NT expanded tiles (any number) and the
// projection runs per TILE on v_mfma_f32_16x16x16_{bf16,f16}: the lane's 4 depthwise results of one tile, rounded to the
// 16-bit type, are exactly that instruction's B operand (k = 4 g .. 4 g + 3 of the tile's 16 channels).  Twice the MFMA
// issue slots per multiply-add of the 32-deep form - on a matrix pipe that idles under the depthwise VALU work.  5x5: the
// 25 taps x 4 channels per tile do not fit the register file next to a four-row ring: two registers per channel hold them
// lane-wise and DPP row broadcasts deliver them (xr_bc5_row below).

template <class T, int K, int S, int ACT, int NC, int TO, int NT, int NW, int MW>
__global__ __launch_bounds__(64 * NW, MW) void mbhq_kernel(MbhrArgs a) {
    constexpr int KK = K * K, PAD = K / 2, NOUT = (16 - K) / S + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    xr_f4* red = reinterpret_cast<xr_f4*>(lds);            // [2][NW][TO][64]
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int seg = bid % a.segs; bid /= a.segs;
    const int strip = bid % a.strips;
    const int b = bid / a.strips;
    const int yo0 = seg * a.seg_rows, yo1 = min(yo0 + a.seg_rows, a.Ho);
    constexpr bool PAIR = S == 2;   // stride 2: pairs of output rows, even input columns in lanes 0..7, odd ones in 8..15 (xr_row_pair)
    const int podd = PAIR ? px >> 3 : 0;
    const int xin = S * NOUT * strip - a.pad_l + (PAIR ? 2 * (px & 7) + podd : px);
    const int xc = min(max(xin, 0), a.W - 1);
    constexpr float HI = ACT == 0 ? 6.f : 1.f;
    const float hi = (xin >= 0 && xin < a.W) ? HI : 0.f;
    const int jo = PAIR ? (px & 7) : (px - PAD) / S, xo = NOUT * strip + jo;
    const bool out_lane = (PAIR ? true : (px >= PAD && (px - PAD) % S == 0)) && jo < NOUT && xo < a.Wo;

    // ---- stationary: the wave's NT expanded tiles (a tile beyond T: all-zero parameters -> its depthwise result is act(0) = 0)
    xr_u4 aw[NT][NC];
    xr_u2 wpf[NT][TO];
    constexpr bool BC5 = K == 5;   // 5x5: lane p of a row holds tap p / tap 16 + p (xr_bc5_row, xr_bc5_pair_row); 3x3: a register per tap
    xr_f4 es[NT], eh[NT], dh[NT], tp[NT][BC5 ? 2 : KK];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = NT * w + j;
        const bool live = t < a.T;
        const int tc = live ? t : 0, ch = 16 * tc + 4 * mg;
        const char* wrow = reinterpret_cast<const char*>(a.we) + ((size_t)(16 * tc + px) * a.KP + 8 * mg) * 2;
        const xr_f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) aw[j][c] = live ? *reinterpret_cast<const xr_u4*>(wrow + 64 * c) : (xr_u4){0u, 0u, 0u, 0u};
        dh[j] = live ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(KK + 1) * a.CexpP + ch) : z;
        es[j] = live ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(KK + 2) * a.CexpP + ch) : z;
        eh[j] = live ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(KK + 3) * a.CexpP + ch) : z;
        const xr_f4 dsc = *reinterpret_cast<const xr_f4*>(a.prm + (size_t)KK * a.CexpP + ch);
        if constexpr (!BC5) {
#pragma unroll
            for (int q = 0; q < KK; ++q) tp[j][q] = live ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)q * a.CexpP + ch) * dsc : z;
        } else {
            tp[j][0] = live ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)px * a.CexpP + ch) * dsc : z;
            tp[j][1] = (live && 16 + px < KK) ? *reinterpret_cast<const xr_f4*>(a.prm + (size_t)(16 + px) * a.CexpP + ch) * dsc : z;
        }
#pragma unroll
        for (int t2 = 0; t2 < TO; ++t2) {   // project A fragment of cout tile t2 for this tile's 16-deep k step: W[16 t2 + m][16 t + 4 g ..]
            const int co = 16 * t2 + px;
            const char* prow = reinterpret_cast<const char*>(a.wp) + ((size_t)(co < a.Cout ? co : 0) * a.CexpP + 16 * tc + 4 * mg) * 2;
            xr_u2 v = *reinterpret_cast<const xr_u2*>(prow);
            if (co >= a.Cout || !live) v = (xr_u2){0u, 0u};
            wpf[j][t2] = v;
        }
    }
    // the cout tile this wave finishes (t = w, if w < TO; with fewer waves than cout tiles, wave w also finishes w + NW, ...)
    constexpr int NF = (TO + NW - 1) / NW;
    bool flive[NF];
    int fco[NF];
    xr_f4 fsc[NF], fsh[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int t = w + f * NW;
        fco[f] = 16 * t + 4 * mg;
        const bool in = t < TO && fco[f] < a.Cout;
        flive[f] = in && out_lane;
        fsc[f] = in ? *reinterpret_cast<const xr_f4*>(a.sp + fco[f]) : (xr_f4){0.f, 0.f, 0.f, 0.f};
        fsh[f] = in ? *reinterpret_cast<const xr_f4*>(a.hp + fco[f]) : (xr_f4){0.f, 0.f, 0.f, 0.f};
    }

    const xr_rsrc xsrc = xr_make_rsrc(reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.ld_in, (unsigned)(a.H * a.W * a.ld_in) * 2u);
    const xr_rsrc osrc = xr_make_rsrc(reinterpret_cast<T*>(a.out) + (size_t)b * a.Ho * a.Wo * a.ld_out, (unsigned)(a.Ho * a.Wo * a.ld_out) * 2u);
    const int rbeg = S * yo0 - a.pad_t, nout = yo1 - yo0;
    unsigned xoff[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) xoff[c] = (32 * c + 8 * mg < a.Cin) ? ((unsigned)xc * (unsigned)a.ld_in + 32u * c + 8u * mg) * 2u : XR_DEAD;
    const unsigned xrow = (unsigned)(a.W * a.ld_in) * 2u;
    struct XRow { xr_u4 m[NC]; };
    XRow xa, xb;
    auto load_row = [&](XRow& x, int r) {
        const unsigned so = (unsigned)min(max(r, 0), a.H - 1) * xrow;
#pragma unroll
        for (int c = 0; c < NC; ++c) x.m[c] = __builtin_bit_cast(xr_u4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[c], so, 0));
    };
    load_row(xa, rbeg);
    xr_f4 ring[NT][K - 1];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < K - 1; ++q) ring[j][q] = (xr_f4){0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    typedef T t4 __attribute__((ext_vector_type(4)));

    auto row = [&](auto emit_c, const int k, const int yo, const XRow& xc_, XRow& xn_) {
        constexpr bool EMIT = decltype(emit_c)::value;
        const int r = rbeg + k;
        load_row(xn_, r + 1);
        xr_u2 resv[NF];
        if constexpr (EMIT) {   // UNCONDITIONAL (a dead offset reads zeros without a residual): a load under a branch makes every later wait vmcnt(0)
#pragma unroll
            for (int f = 0; f < NF; ++f)
                resv[f] = __builtin_bit_cast(xr_u2, __builtin_amdgcn_raw_buffer_load_b64(xsrc, (flive[f] && a.has_res) ? (((unsigned)yo * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_in + (unsigned)fco[f]) * 2u : XR_DEAD, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);   // (the loads are issued HERE: left alone the scheduler sinks the residual load to its use behind the barrier)
        const float hr = (r >= 0 && r < a.H) ? hi : 0.f;
        xr_f4 ec[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            xr_f4 d = (xr_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) d = xr_mfma<T>(aw[j][c], xc_.m[c], d);
            ec[j] = xr_bn_act4<ACT>(d, es[j], eh[j], hr);
        }
        if constexpr (EMIT) {
            xr_f4 P[TO];
#pragma unroll
            for (int t = 0; t < TO; ++t) P[t] = (xr_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                xr_f4 d = dh[j];
                if constexpr (K == 3) {
                    xr_row3(d, ring[j][0], tp[j][0], tp[j][1], tp[j][2]);
                    xr_row3(d, ring[j][1], tp[j][3], tp[j][4], tp[j][5]);
                    xr_row3(d, ec[j], tp[j][6], tp[j][7], tp[j][8]);
                } else {
                    float cs[5][4];   // the five column sums (see xr_bc5_row)
                    xr_bc5_row<0>(cs, ring[j][0], tp[j][0], tp[j][1]);
                    xr_bc5_row<1>(cs, ring[j][1], tp[j][0], tp[j][1]);
                    xr_bc5_row<2>(cs, ring[j][2], tp[j][0], tp[j][1]);
                    xr_bc5_row<3>(cs, ring[j][3], tp[j][0], tp[j][1]);
                    xr_bc5_row<4>(cs, ec[j], tp[j][0], tp[j][1]);
                    d = xr_bc5_finish(cs, d);
                }
                const xr_f4 v = xr_act4<ACT>(d, HI);
                const xr_u2 bop = __builtin_bit_cast(xr_u2, __builtin_convertvector(v, t4));   // rounded: the projection's operand type
#pragma unroll
                for (int t = 0; t < TO; ++t) P[t] = xr_mfma16<T>(wpf[j][t], bop, P[t]);
            }
            xr_f4* rb = red + buf * (NW * TO * 64);
#pragma unroll
            for (int t = 0; t < TO; ++t) rb[(w * TO + t) * 64 + lane] = P[t];
            __syncthreads();
            const unsigned opix = ((unsigned)yo * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_out;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int t = w + f * NW;
                xr_f4 acc = {0.f, 0.f, 0.f, 0.f};
                if (t < TO) {
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) acc += rb[(ww * TO + t) * 64 + lane];
                }
                xr_f4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(acc[i], fsc[f][i], fsh[f][i]);
                v += __builtin_convertvector(__builtin_bit_cast(t4, resv[f]), xr_f4);   // (zeros without a residual)
                const t4 o = __builtin_convertvector(v, t4);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xr_u2, o), osrc, flive[f] ? (opix + (unsigned)fco[f]) * 2u : XR_DEAD, 0, 0);
            }
            buf ^= 1;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q + 1 < K - 1; ++q) ring[j][q] = ring[j][q + 1];
            ring[j][K - 2] = ec[j];
        }
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    if constexpr (PAIR) {
        // (see mbxr_kernel: a pair of output rows reads input rows rho = 0 .. K + 1; the first K - 2 are carried over, four are new)
        xr_f4 d2[NT], cr[NT][K - 2];
        float csa[NT][3][4], csb[NT][3][4];   // (5x5: the column sums of the pair's even / odd row)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < K - 2; ++q) cr[j][q] = (xr_f4){0.f, 0.f, 0.f, 0.f};
        auto prow = [&](auto ph_c, const int k, const int yo, const XRow& xc_, XRow& xn_) {
            constexpr int PH = decltype(ph_c)::value;
            const int r = rbeg + k;
            load_row(xn_, r + 1);
            const float hr = (r >= 0 && r < a.H) ? hi : 0.f;
            xr_f4 ec[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                xr_f4 d = (xr_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NC; ++c) d = xr_mfma<T>(aw[j][c], xc_.m[c], d);
                ec[j] = xr_bn_act4<ACT>(d, es[j], eh[j], hr);
            }
            if constexpr (PH >= 1) {
                constexpr int RHO = K - 3 + PH;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (K == 3) {
                        if constexpr (PH == 1) {
                            d2[j] = dh[j];
                            xr_row_pair<K, false>(d2[j], cr[j][0], &tp[j][0]);
                        }
                        if constexpr (RHO < K) xr_row_pair<K, false>(d2[j], ec[j], &tp[j][K * (RHO < K ? RHO : 0)]);
                        if constexpr (RHO >= 2) xr_row_pair<K, true>(d2[j], ec[j], &tp[j][K * (RHO >= 2 ? RHO - 2 : 0)]);
                    } else {
                        if constexpr (PH == 1) {
#pragma unroll
                            for (int q = 0; q < 3; ++q)
#pragma unroll
                                for (int i = 0; i < 4; ++i) { csa[j][q][i] = 0.f; csb[j][q][i] = 0.f; }
                            xr_bc5_half_row<0>(csa[j], cr[j][0], tp[j][0], tp[j][1]);
                            xr_bc5_half_row<1>(csa[j], cr[j][1], tp[j][0], tp[j][1]);
                            xr_bc5_half_row<2>(csa[j], cr[j][2], tp[j][0], tp[j][1]);
                            xr_bc5_half_row<0>(csb[j], cr[j][2], tp[j][0], tp[j][1]);
                        }
                        if constexpr (RHO < K) xr_bc5_half_row<(RHO < K ? RHO : 0)>(csa[j], ec[j], tp[j][0], tp[j][1]);
                        xr_bc5_half_row<RHO - 2>(csb[j], ec[j], tp[j][0], tp[j][1]);
                        if constexpr (PH == 4) d2[j] = xr_bc5_pair_finish(csa[j], csb[j], dh[j]);
                    }
                }
            }
            if constexpr (PH == 4) {
                xr_f4 P[TO];
#pragma unroll
                for (int t = 0; t < TO; ++t) P[t] = (xr_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const xr_f4 v = xr_act4<ACT>(d2[j], HI);
                    const xr_u2 bop = __builtin_bit_cast(xr_u2, __builtin_convertvector(v, t4));
#pragma unroll
                    for (int t = 0; t < TO; ++t) P[t] = xr_mfma16<T>(wpf[j][t], bop, P[t]);
                }
                xr_f4* rb = red + buf * (NW * TO * 64);
#pragma unroll
                for (int t = 0; t < TO; ++t) rb[(w * TO + t) * 64 + lane] = P[t];
                __syncthreads();
                const int yl = yo + podd;   // (lanes 8..15 hold the row below)
                const bool rlive = yl < yo1;
                const unsigned opix = ((unsigned)yl * (unsigned)a.Wo + (unsigned)xo) * (unsigned)a.ld_out;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int t = w + f * NW;
                    xr_f4 acc = {0.f, 0.f, 0.f, 0.f};
                    if (t < TO) {
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) acc += rb[(ww * TO + t) * 64 + lane];
                    }
                    xr_f4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(acc[i], fsc[f][i], fsh[f][i]);
                    const t4 o = __builtin_convertvector(v, t4);   // (a stride-2 block has no residual)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xr_u2, o), osrc, (flive[f] && rlive) ? (opix + (unsigned)fco[f]) * 2u : XR_DEAD, 0, 0);
                }
                buf ^= 1;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int q = 0; q + 1 < K - 2; ++q) cr[j][q] = cr[j][q + 1];
                cr[j][K - 3] = ec[j];
            }
        };
        constexpr std::integral_constant<int, 0> P0{};
        constexpr std::integral_constant<int, 1> P1{};
        constexpr std::integral_constant<int, 2> P2{};
        constexpr std::integral_constant<int, 3> P3{};
        constexpr std::integral_constant<int, 4> P4{};
        int k = 0;
#pragma unroll
        for (int q = 0; q < (K - 2) / 2; ++q) { prow(P0, k, 0, xa, xb); prow(P0, k + 1, 0, xb, xa); k += 2; }
        prow(P0, k, 0, xa, xb); k += 1;
        for (int i = 0; i < nout; i += 2) {
            prow(P1, k, 0, xb, xa);
            prow(P2, k + 1, 0, xa, xb);
            prow(P3, k + 2, 0, xb, xa);
            prow(P4, k + 3, yo0 + i, xa, xb);
            k += 4;
        }
        return;
    }
    // rows 0 .. K - S - 1 warm the ring up; then every output row takes S input rows, the last of which emits
    int k = 0;
#pragma unroll
    for (int q = 0; q < (K - S) / 2; ++q) { row(N, k, 0, xa, xb); row(N, k + 1, 0, xb, xa); k += 2; }
    if constexpr ((K - S) % 2 != 0) { row(N, k, 0, xa, xb); k += 1; }
    if constexpr (S == 2) {   // (an odd warm-up: the current row's operands are in xb)
        for (int i = 0; i < nout; ++i) {
            row(N, k, 0, xb, xa);
            row(Y, k + 1, yo0 + i, xa, xb);
            k += 2;
        }
    } else {
        int i = 0;
        for (; i + 1 < nout; i += 2) {
            row(Y, k, yo0 + i, xa, xb);
            row(Y, k + 1, yo0 + i + 1, xb, xa);
            k += 2;
        }
        if (i < nout) row(Y, k, yo0 + i, xa, xb);
    }
}

template <class T, int K, int S, int ACT, int NC, int TO, int NT, int NW>
static int launch_mbhq(const MbhrArgs& a0, int batch, int want_segs, hipStream_t s) {
    MbhrArgs a = a0;
    constexpr int NOUT = (16 - K) / S + 1;
    constexpr int MW = 2;
    static_assert(NW <= 4 * MW, "the workgroup's waves must fit one CU");
    a.strips = (a.Wo + NOUT - 1) / NOUT;
    const int walks = batch * a.strips;
    int segs = (2 * 1024 + walks * NW - 1) / (walks * NW);
    const int max_segs = (a.Ho + 3 * K - 1) / (3 * K);   // (a segment recomputes K - S halo rows)
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    if (want_segs > 0) segs = want_segs < a.Ho ? want_segs : a.Ho;
    a.seg_rows = (a.Ho + segs - 1) / segs;
    if (S == 2) a.seg_rows += a.seg_rows & 1;   // (output rows are processed in pairs)
    a.segs = (a.Ho + a.seg_rows - 1) / a.seg_rows;
    const size_t lds = (size_t)2 * NW * TO * 64 * 16;
    static char nm[64];
    static const int nm_len = snprintf(nm, sizeof(nm), "mbhq_kernel<%s,%d,%d,%d,%d,%d,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, S, ACT, NC, TO, NT, NW);
    (void)nm_len;
    yr_note_kernel(nm);
    auto kern = mbhq_kernel<T, K, S, ACT, NC, TO, NT, NW, MW>;
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

// shapes of the tile-wise form: (kernel, input chunks, cout tiles, expanded tiles) -> (tiles per wave, waves)
struct MbhqShape { int k, s, nc, to, t, nt, nw; };   // (s: the stride the shape is built for, 0 = both)
static const MbhqShape MBHQ_SHAPES[] = {
    {5, 0, 1, 3, 9, 3, 3},     // 24 -> 144 -> 40 (EfficientNet-lite0 stage 3 entry, stride 2)
    {5, 0, 1, 3, 12, 3, 4},    // 32 -> 192 -> 48 (lite3 stage 3 entry, stride 2)
    {5, 0, 2, 3, 18, 3, 6},    // 48 -> 288 -> 48 (lite3 stage 3)
    {5, 0, 2, 3, 15, 2, 8},    // 40 -> 240 -> 40 (lite0 stage 3)
    {3, 0, 1, 2, 9, 3, 3},     // 24 -> 144 -> 24 / 32 (lite0 stage 2, lite3 stage 2 entry, MobileNetV2 x0.75 block_2, 3)
    // stride-2 blocks the tile-pair form (mbhr_kernel) is also built for: here they walk pairs of output rows
    {3, 2, 1, 2, 6, 2, 3},     // 16 -> 96 -> 24 (MobileNetV2 block_1, lite0 stage 2 entry)
    {3, 2, 2, 5, 15, 2, 8},    // 40 -> 240 -> 80 (lite0 stage 4 entry)
};
static const MbhqShape* mbhq_shape(int k, int stride, int cin, int cexp, int cout) {
    static const bool s2 = !(getenv("YOLORET_MBHQ_S2") && atoi(getenv("YOLORET_MBHQ_S2")) == 0);
    const int nc = yr_round_up(cin, 32) / 32, to = (cout + 15) / 16, t = cexp / 16;
    for (const MbhqShape& sh : MBHQ_SHAPES)
        if (sh.k == k && (sh.s == 0 || (sh.s == stride && s2)) && sh.nc == nc && sh.to == to && sh.t == t) return &sh;
    return nullptr;
}

template <class T, int S>
static int launch_mbhq_shape(const MbhrArgs& a, int k, int batch, int segs, hipStream_t s) {
    const MbhqShape* sh = mbhq_shape(k, S, a.Cin, a.T * 16, a.Cout);
    YR_REQUIRE(sh != nullptr, "mbhq: block %d -> %d -> %d (%d x %d) is not built", a.Cin, a.T * 16, a.Cout, k, k);
#define HQ_CASE(KV, NCV, TOV, TV, NTV, NWV) if (sh->k == KV && sh->nc == NCV && sh->to == TOV && sh->t == TV) return launch_mbhq<T, KV, S, 0, NCV, TOV, NTV, NWV>(a, batch, segs, s);
    HQ_CASE(5, 1, 3, 9, 3, 3)
    HQ_CASE(5, 1, 3, 12, 3, 4)
    HQ_CASE(5, 2, 3, 18, 3, 6)
    HQ_CASE(5, 2, 3, 15, 2, 8)
    HQ_CASE(3, 1, 2, 9, 3, 3)
    if constexpr (S == 2) {
        HQ_CASE(3, 1, 2, 6, 2, 3)
        HQ_CASE(3, 2, 5, 15, 2, 8)
    }
#undef HQ_CASE
    return YR_ERR_ARG;
}

// the whole-block register-chained form is built for: 3x3, at most 16 expanded tiles, cin <= 64, cout <= 80
bool yr_mbhr_takes(const yr_op& op) {
    return op.kind == YR_OP_MBH && (op.dtype == YR_BF16 || op.dtype == YR_F16) && (op.k & 0xff) == 3 && (op.stride == 1 || op.stride == 2) &&
           (op.act == YR_ACT_RELU6 || op.act == YR_ACT_SWISH) && op.cin % 8 == 0 && op.cin <= 64 && op.se_reduced % 16 == 0 && op.se_reduced <= 256 &&
           op.cout % 4 == 0 && op.cout <= 80 && op.nsrc == 1 && op.out_ld % 4 == 0;
}

template <class T, int S, int ACT>
static int launch_mbhr_shape(const MbhrArgs& a, int batch, int segs, hipStream_t s) {
    const int nc = a.KP / 32, to = (a.Cout + 15) / 16, nw = (a.T + 1) / 2;
#define HR_CASE(NCV, TOV, NWV) if (nc == NCV && to == TOV && nw == NWV) return launch_mbhr<T, S, ACT, NCV, TOV, NWV>(a, batch, segs, s);
    // (measured, tools/mbhr_probe.py: the five-wave workgroups of the 144-channel blocks - 24 -> 144 -> 24 / 32 / 48 - lose to the
    // LDS-tiled kernels, 0.29 vs 0.25 ms and 0.44 vs 0.33 ms: one such workgroup per CU at two waves per SIMD; not built)
    HR_CASE(1, 2, 3)    // 16 -> 96 -> 24 (MobileNetV2 block_1, lite0 stage 2 entry): 0.36 -> 0.24 ms at batch 128
    HR_CASE(1, 2, 6)    // 32 -> 192 -> 32 (lite3 stage 2): 0.23 -> 0.18 ms at batch 32
    HR_CASE(1, 3, 6)    // 32 -> 192 -> 48
    HR_CASE(2, 5, 8)    // 40 -> 240 -> 80 (lite0 stage 4 entry): 0.16 -> 0.075 ms
    HR_CASE(2, 3, 8)    // 40 -> 240 -> 40
#undef HR_CASE
    yr_set_error("mbhr: block %d -> %d -> %d is not built", a.Cin, a.T * 16, a.Cout);
    return YR_ERR_ARG;
}

static bool mbhr_pair_built(const yr_op& op) {
    if (!yr_mbhr_takes(op)) return false;
    const int nc = yr_round_up(op.cin, 32) / 32, to = (op.cout + 15) / 16, nw = (op.se_reduced / 16 + 1) / 2;
    const int key = nc * 10000 + to * 100 + nw;
    return key == 10203 || key == 10206 || key == 10306 || key == 20508 || key == 20308;
}
// the tile-wise form (mbhq_kernel): 3x3 / 5x5, ReLU6, the shapes of MBHQ_SHAPES
static bool mbhq_built(const yr_op& op) {
    static const bool on = !(getenv("YOLORET_MBHQ") && atoi(getenv("YOLORET_MBHQ")) == 0);
    const int K = op.k & 0xff;
    return on && op.kind == YR_OP_MBH && (op.dtype == YR_BF16 || op.dtype == YR_F16) && (K == 3 || K == 5) && (op.stride == 1 || op.stride == 2) &&
           op.act == YR_ACT_RELU6 && op.cin % 8 == 0 && op.cin <= 64 && op.se_reduced % 16 == 0 && op.cout % 4 == 0 && op.nsrc == 1 &&
           op.out_ld % 4 == 0 && mbhq_shape(K, op.stride, op.cin, op.se_reduced, op.cout) != nullptr;
}
bool yr_mbhr_built(const yr_op& op) { return mbhr_pair_built(op) || mbhq_built(op); }

template <class T>
static int launch_mbhr_t(const yr_op& op, int batch, int segs, hipStream_t s) {
    const yr_src& in = op.src[0];
    MbhrArgs a;
    a.x = in.ptr; a.out = op.out; a.we = op.wgt; a.prm = op.wgt2; a.wp = op.b1; a.sp = op.b2; a.hp = op.b2 + yr_round_up(op.cout, 8);
    a.H = in.h; a.W = in.w; a.Ho = op.h; a.Wo = op.w; a.Cin = in.c; a.CexpP = yr_round_up(op.se_reduced, 32); a.KP = yr_round_up(in.c, 32);
    a.Cout = op.cout; a.ld_in = in.ld; a.ld_out = op.out_ld; a.T = op.se_reduced / 16; a.has_res = op.res != nullptr;
    const int K = op.k & 0xff;
    const int pth = (a.Ho - 1) * op.stride + K - in.h, ptw = (a.Wo - 1) * op.stride + K - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.strips = a.segs = a.seg_rows = 0;
    if (mbhq_built(op)) return op.stride == 1 ? launch_mbhq_shape<T, 1>(a, K, batch, segs, s) : launch_mbhq_shape<T, 2>(a, K, batch, segs, s);
    if (op.act == YR_ACT_RELU6) return op.stride == 1 ? launch_mbhr_shape<T, 1, 0>(a, batch, segs, s) : launch_mbhr_shape<T, 2, 0>(a, batch, segs, s);
    return op.stride == 1 ? launch_mbhr_shape<T, 1, 1>(a, batch, segs, s) : launch_mbhr_shape<T, 2, 1>(a, batch, segs, s);
}

// Whether the register-chained form is what a plain launch (no forced tile) of this MBH / MBX op runs.  The choice is by SHAPE,
// never by the tuner (which only picks the row segments): the two forms round differently (the LDS-tiled mbn_h.hip keeps the
// expanded tile in the 16-bit type, the accumulation orders differ), and a batch must equal its images run one by one.
bool yr_mbxr_takes(const yr_op& op);
bool yr_mbh_prefers_chained(const yr_op& op) {
    static const bool mbxr_on = !(getenv("YOLORET_MBXR") && atoi(getenv("YOLORET_MBXR")) == 0);
    static const bool mbhr_on = !(getenv("YOLORET_MBHR") && atoi(getenv("YOLORET_MBHR")) == 0);
    if (((op.k >> 8) & 0xff) != 0) return ((op.k >> 8) & 0xff) == 255;
    return op.kind == YR_OP_MBX ? (mbxr_on && yr_mbxr_takes(op)) : (mbhr_on && yr_mbhr_built(op));
}

int yr_launch_mbhr(const yr_op& op, int batch, int segs, hipStream_t s) {
    YR_REQUIRE(yr_mbhr_built(op), "mbhr: the register-chained whole-block form is not built for this op");
    return op.dtype == YR_BF16 ? launch_mbhr_t<yr_bf16>(op, batch, segs, s) : launch_mbhr_t<yr_f16>(op, batch, segs, s);
}

// whether the register-chained form is built for this YR_OP_MBX op (mbh.hip asks before it dispatches here)
bool yr_mbxr_takes(const yr_op& op) {
    const int K = op.k & 0xff;
    return op.kind == YR_OP_MBX && (op.dtype == YR_BF16 || op.dtype == YR_F16) && (K == 3 || K == 5) && (op.stride == 1 || op.stride == 2) &&
           (op.act == YR_ACT_RELU6 || op.act == YR_ACT_SWISH) && op.cin % 8 == 0 && op.cin <= 128 && op.cout % 16 == 0 && op.nsrc == 1 &&
           op.out_ld % 4 == 0 && (op.gate == nullptr || op.gate_ld % 4 == 0);
}

template <class T>
static int launch_mbxr_t(const yr_op& op, int batch, int segs, hipStream_t s) {
    const yr_src& in = op.src[0];
    const int K = op.k & 0xff;
    MbxrArgs a;
    a.x = in.ptr; a.out = op.out; a.we = op.wgt; a.prm = op.wgt2;
    a.part = const_cast<float*>(op.gate); a.ld_part = op.gate ? op.gate_ld : 0; a.rows_cap = op.gate ? op.se_reduced : 0;
    a.H = in.h; a.W = in.w; a.Ho = op.h; a.Wo = op.w; a.Cin = in.c; a.CexpP = yr_round_up(op.cout, 32); a.KP = yr_round_up(in.c, 32);
    a.ld_in = in.ld; a.ld_out = op.out_ld; a.T = op.cout / 16;
    const int pth = (a.Ho - 1) * op.stride + K - in.h, ptw = (a.Wo - 1) * op.stride + K - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.strips = a.segs = a.seg_rows = a.groups = a.nwaves = 0;
    const int act = op.act == YR_ACT_RELU6 ? 0 : 1;
#define XR_GO(KV, SV) (act == 0 ? launch_mbxr_nc<T, KV, SV, 0>(a, batch, segs, s) : launch_mbxr_nc<T, KV, SV, 1>(a, batch, segs, s))
    if (K == 3) return op.stride == 1 ? XR_GO(3, 1) : XR_GO(3, 2);
    return op.stride == 1 ? XR_GO(5, 1) : XR_GO(5, 2);
#undef XR_GO
}

// (the argument checks of mbh.hip's launcher have run: pointers, strides, dims)
int yr_launch_mbxr(const yr_op& op, int batch, int segs, hipStream_t s) {
    YR_REQUIRE(yr_mbxr_takes(op), "mbxr: the register-chained form is not built for this op");
    return op.dtype == YR_BF16 ? launch_mbxr_t<yr_bf16>(op, batch, segs, s) : launch_mbxr_t<yr_f16>(op, batch, segs, s);
}
