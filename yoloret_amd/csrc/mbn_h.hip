// The first inverted-residual block of the 16-bit plans - expand 1x1 from at most 32 channels + BN + act -> depthwise 3x3
// STRIDE 2 + BN + act -> project 1x1 + BN, no residual (EfficientNet-lite0 stage 2 block 0: 208 x 208 x 16 -> 96 -> 104 x 104 x
// 24, efficientnet.py:467-536; MobileNetV2 block_1, override.py:339) - on the matrix pipe.  Until round 3 this block ran on
// the float32 lane-per-pixel kernel also in the 16-bit plans (mblane_s2_kernel: 0.42 ms per 128 images, the largest kernel
// of BASELINE config 3, bound by the 2328 float32 MACs per input pixel of its two 1x1 convolutions), because the general
// fused block kernel (mbh.hip: 32-channel chunks, float32 expanded tile, two barriers per chunk) is slower still on it.
// Same scheme as stemblock_h.hip:
//   One workgroup (4 waves) = 7 x 8 output pixels = a 15 x 17 halo tile of 255 input pixels = 16 MFMA pixel tiles.
//   1. expand: B operand = the pixel's input channels straight from global memory (16 bytes per lane, through a buffer
//      descriptor: pixels outside the image read as zeros), A = the expand kernel; CexpP / 16 MFMAs per pixel tile (k = Cin
//      padded to 32); BatchNorm, activation, zero outside the map (TF pads the depthwise conv's input), rounded to the
//      plan's type into Es[256][CexpP] - ALL expanded channels at once, one barrier.
//   2. depthwise + projection: wave = one tile of 16 output pixels; lane = (pixel, 8 channels) - the projection's B
//      operand: nine 16-byte LDS reads, 36 packed FMAs, BatchNorm, activation, rounding, and the 8 values ARE the operand
//      of the projection MFMAs of this k step.  Project BN, 8-byte stores.
// Parameters in YR_OP_MBH's layout (mbh.hip), which dispatches here.
#include "yr_common.h"

typedef float mbn_f4 __attribute__((ext_vector_type(4)));
typedef float mbn_f2 __attribute__((ext_vector_type(2)));
typedef float mbn_f8 __attribute__((ext_vector_type(8)));
typedef unsigned mbn_u4 __attribute__((ext_vector_type(4)));
template <class T> using mbn_v8 = T __attribute__((ext_vector_type(8)));
template <class T> using mbn_v4 = T __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t mbn_rsrc;

#define MBN_TH 7
#define MBN_TW 8
#define MBN_IH 15
#define MBN_IW 17

struct MbnArgs {
    const void* x; void* out;            // T
    const void* we;                      // expand Wt[CexpP][32] (T)
    const float* prm;                    // [13][CexpP]: 9 depthwise taps | dw BN scale | shift | expand BN scale | shift
    const void* wp; const float* sp; const float* hp;   // project Wt[Cout][CexpP] (T), BN scale / shift
    int Hi, Wi, Ho, Wo, Cin, CexpP, Cout, ld_in, ld_out, pad_t, pad_l, tiles_x, tiles_y, act;
};

template <class T>
__device__ __forceinline__ mbn_f4 mbn_mfma(mbn_u4 a, mbn_u4 b, mbn_f4 c) {
    if constexpr (yr_elem<T>::dtype == YR_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mbn_v8<__bf16>, a), __builtin_bit_cast(mbn_v8<__bf16>, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mbn_v8<_Float16>, a), __builtin_bit_cast(mbn_v8<_Float16>, b), c, 0, 0, 0);
}

template <int ACT>
__device__ __forceinline__ float mbn_act(float v) {
    if constexpr (ACT == 0) return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f);
    else return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));   // swish, as in mbh.hip
}

// NE = CexpP / 16 (even, <= 12), NCO = round_up(Cout, 16) / 16 (1 | 2).  More than 96 expanded channels go in PASSES of 96
// (lite3's 24 -> 144 -> 32 entry block: 96 + 64): the pixel operands stay in registers, Es holds one pass, the projection
// accumulates over all of them.
template <class T, int NE, int NCO, int ACT>
__global__ __launch_bounds__(256) void mbn_kernel(MbnArgs a) {
    constexpr int CEP = 16 * NE, KS = CEP / 32;                  // all expanded channels: row pitch of the parameter arrays, k steps
    constexpr int NEP = NE < 6 ? NE : 6, NP = (NE + 5) / 6;      // 16-channel tiles per pass, passes
    constexpr int LDE = 16 * NEP + 8;                            // Es row pitch 80 / 144 / 208 bytes: conflict-free 16-byte rows
    extern __shared__ __attribute__((aligned(16))) char mbn_lds[];
    T* Es = reinterpret_cast<T*>(mbn_lds);   // [256 halo pixels][LDE]
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
    const int tpi = a.tiles_x * a.tiles_y;
    const int t = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = t / tpi, r = t - b * tpi;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int oy0 = ty * MBN_TH, ox0 = tx * MBN_TW;
    const int iy0 = oy0 * 2 - a.pad_t, ix0 = ox0 * 2 - a.pad_l;

    // the halo pixels of this wave's four MFMA tiles: the expand GEMM's B operands, straight from global memory
    mbn_u4 xf[4];
    bool inmap[4];
    {
        const mbn_rsrc src = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<const T*>(a.x) + (size_t)b * a.Hi * a.Wi * a.ld_in), 0, (unsigned)(a.Hi * a.Wi * a.ld_in) * 2u, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = (wave * 4 + q) * 16 + li;
            const int hy = p / MBN_IW, hx = p - hy * MBN_IW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            inmap[q] = p < MBN_IH * MBN_IW && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const unsigned off = (unsigned)((iy * a.Wi + ix) * a.ld_in + 8 * g) * 2u;
            xf[q] = __builtin_bit_cast(mbn_u4, __builtin_amdgcn_raw_buffer_load_b128(src, inmap[q] && 8 * g < a.Cin ? off : 0x80000000u, 0, 0));
        }
    }
    // depthwise-phase identity: output pixel o = 16 wave + li of the tile's 56
    const int o = wave * 16 + li;
    const int oc = o < MBN_TH * MBN_TW ? o : MBN_TH * MBN_TW - 1;
    const int oy = oc >> 3, ox = oc & 7;
    const T* ewin = Es + (size_t)((2 * oy) * MBN_IW + 2 * ox) * LDE + 8 * g;
    mbn_f4 pacc[NCO];
#pragma unroll
    for (int n = 0; n < NCO; ++n) pacc[n] = (mbn_f4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        // ---- 1. expand: channels 96 ps ...
        if (ps > 0) __syncthreads();   // the previous pass's windows have been read
#pragma unroll
        for (int jj = 0; jj < NEP; ++jj) {
            const int j = ps * 6 + jj;
            if (j < NE) {
                const mbn_u4 wef = *reinterpret_cast<const mbn_u4*>(reinterpret_cast<const T*>(a.we) + (size_t)(16 * j + li) * 32 + 8 * g);
                const mbn_f4 sc = *reinterpret_cast<const mbn_f4*>(a.prm + 11 * CEP + 16 * j + 4 * g), sh = *reinterpret_cast<const mbn_f4*>(a.prm + 12 * CEP + 16 * j + 4 * g);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p = (wave * 4 + q) * 16 + li;
                    const mbn_f4 acc = mbn_mfma<T>(wef, xf[q], (mbn_f4){0.f, 0.f, 0.f, 0.f});
                    mbn_f4 y = __builtin_elementwise_fma(acc, sc, sh);
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = inmap[q] ? mbn_act<ACT>(y[i]) : 0.f;
                    *reinterpret_cast<mbn_v4<T>*>(Es + (size_t)p * LDE + 16 * jj + 4 * g) = __builtin_convertvector(y, mbn_v4<T>);
                }
            }
        }
        __syncthreads();

        // ---- 2. depthwise (stride 2) + projection over this pass's k steps
#pragma unroll
        for (int kk = 0; kk < NEP / 2; ++kk) {
            const int ks = ps * 3 + kk;
            if (ks < KS) {
                const int c0 = 32 * ks + 8 * g;
                mbn_u4 wpf[NCO];
#pragma unroll
                for (int n = 0; n < NCO; ++n) {
                    const int row = 16 * n + li;
                    wpf[n] = *reinterpret_cast<const mbn_u4*>(reinterpret_cast<const T*>(a.wp) + (size_t)(row < a.Cout ? row : 0) * CEP + c0);
                }
                mbn_f2 acc[4];
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const mbn_f4 wlo = *reinterpret_cast<const mbn_f4*>(a.prm + tp * CEP + c0), whi = *reinterpret_cast<const mbn_f4*>(a.prm + tp * CEP + c0 + 4);
                    const mbn_u4 raw = *reinterpret_cast<const mbn_u4*>(ewin + (size_t)((tp / 3) * MBN_IW + tp % 3) * LDE + 32 * kk);
                    const mbn_f8 xv = __builtin_convertvector(__builtin_bit_cast(mbn_v8<T>, raw), mbn_f8);   // (whole-vector cast: see stemblock_h.hip)
                    const mbn_f2 w2[4] = {(mbn_f2){wlo[0], wlo[1]}, (mbn_f2){wlo[2], wlo[3]}, (mbn_f2){whi[0], whi[1]}, (mbn_f2){whi[2], whi[3]}};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[c] = __builtin_elementwise_fma((mbn_f2){xv[2 * c], xv[2 * c + 1]}, w2[c], tp == 0 ? (mbn_f2){0.f, 0.f} : acc[c]);
                }
                const mbn_f4 slo = *reinterpret_cast<const mbn_f4*>(a.prm + 9 * CEP + c0), shi = *reinterpret_cast<const mbn_f4*>(a.prm + 9 * CEP + c0 + 4);
                const mbn_f4 hlo = *reinterpret_cast<const mbn_f4*>(a.prm + 10 * CEP + c0), hhi = *reinterpret_cast<const mbn_f4*>(a.prm + 10 * CEP + c0 + 4);
                mbn_f8 d;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const mbn_f2 s2 = c < 2 ? (mbn_f2){slo[2 * c], slo[2 * c + 1]} : (mbn_f2){shi[2 * c - 4], shi[2 * c - 3]};
                    const mbn_f2 h2 = c < 2 ? (mbn_f2){hlo[2 * c], hlo[2 * c + 1]} : (mbn_f2){hhi[2 * c - 4], hhi[2 * c - 3]};
                    const mbn_f2 y = __builtin_elementwise_fma(acc[c], s2, h2);
                    d[2 * c] = mbn_act<ACT>(y.x);
                    d[2 * c + 1] = mbn_act<ACT>(y.y);
                }
                const mbn_u4 frag = __builtin_bit_cast(mbn_u4, __builtin_convertvector(d, mbn_v8<T>));
#pragma unroll
                for (int n = 0; n < NCO; ++n) pacc[n] = mbn_mfma<T>(wpf[n], frag, pacc[n]);
            }
        }
    }
    // ---- 3. project BN, stores: lane = pixel o, couts 16 n + 4 g + 0..3
    const int gy = oy0 + oy, gx = ox0 + ox;
    if (o < MBN_TH * MBN_TW && gy < a.Ho && gx < a.Wo) {
        T* op = reinterpret_cast<T*>(a.out) + ((size_t)b * a.Ho * a.Wo + (size_t)gy * a.Wo + gx) * a.ld_out;
#pragma unroll
        for (int n = 0; n < NCO; ++n) {
            const int co = 16 * n + 4 * g;
            if (co < a.ld_out) {
                mbn_f4 y;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cc = co + i < a.Cout ? co + i : a.Cout - 1;
                    y[i] = co + i < a.Cout ? __builtin_fmaf(pacc[n][i], a.sp[cc], a.hp[cc]) : 0.f;
                }
                *reinterpret_cast<mbn_v4<T>*>(op + co) = __builtin_convertvector(y, mbn_v4<T>);
            }
        }
    }
}

template <class T, int NE, int NCO>
static int launch_mbn(const MbnArgs& a, int batch, hipStream_t s) {
    constexpr size_t lds = (size_t)256 * (16 * (NE < 6 ? NE : 6) + 8) * 2;
    static char nm[48];
    static const int nm_len = snprintf(nm, sizeof(nm), "mbn_kernel<%s,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), NE, NCO);
    (void)nm_len;
    yr_note_kernel(nm);
    const dim3 grid((unsigned)(batch * a.tiles_x * a.tiles_y));
    if (a.act == YR_ACT_RELU6) hipLaunchKernelGGL((mbn_kernel<T, NE, NCO, 0>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((mbn_kernel<T, NE, NCO, 1>), grid, dim3(256), lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// Whether yr_launch_mbh hands the op over (its checks have passed): the whole block, 3x3 stride 2, at most 32 inputs in whole
// 16-byte vectors, at most 192 expanded channels (two passes of 96), at most 32 outputs, no residual, ReLU6 or swish.
bool yr_mbn_takes(const yr_op& op) {
    return op.kind == YR_OP_MBH && (op.k & 0xff) == 3 && (op.k >> 8) == 0 && op.stride == 2 && op.src[0].c <= 32 && op.src[0].c % 8 == 0 && op.se_reduced <= 192 &&
           op.cout <= 32 && op.res == nullptr && (op.act == YR_ACT_RELU6 || op.act == YR_ACT_SWISH);
}

template <class T>
static int launch_mbn_t(const yr_op& op, int batch, hipStream_t s) {
    const yr_src& in = op.src[0];
    MbnArgs a;
    a.x = in.ptr; a.out = op.out; a.we = op.wgt; a.prm = op.wgt2; a.wp = op.b1;
    a.sp = op.b2; a.hp = op.b2 + yr_round_up(op.cout, 8);
    a.Cin = in.c; a.Cout = op.cout; a.CexpP = yr_round_up(op.se_reduced, 32);
    a.Hi = in.h; a.Wi = in.w; a.Ho = (in.h + 1) / 2; a.Wo = (in.w + 1) / 2;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "mbn: output dims mismatch");
    YR_REQUIRE((long long)a.Hi * a.Wi * in.ld * 2 < (1ll << 31), "mbn: one image of the map must be below 2 GB");
    a.ld_in = in.ld; a.ld_out = op.out_ld;
    const int pth = (a.Ho - 1) * 2 + 3 - in.h, ptw = (a.Wo - 1) * 2 + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.tiles_x = (a.Wo + MBN_TW - 1) / MBN_TW; a.tiles_y = (a.Ho + MBN_TH - 1) / MBN_TH;
    a.act = op.act;
    YR_REQUIRE((long long)batch * a.tiles_x * a.tiles_y < (1ll << 31), "mbn: grid too large");
    const int ne = a.CexpP / 16, nco = yr_round_up(op.cout, 16) / 16;
    switch (ne * 10 + nco) {
        case 21: return launch_mbn<T, 2, 1>(a, batch, s);
        case 22: return launch_mbn<T, 2, 2>(a, batch, s);
        case 41: return launch_mbn<T, 4, 1>(a, batch, s);
        case 42: return launch_mbn<T, 4, 2>(a, batch, s);
        case 61: return launch_mbn<T, 6, 1>(a, batch, s);
        case 62: return launch_mbn<T, 6, 2>(a, batch, s);
        case 81: return launch_mbn<T, 8, 1>(a, batch, s);
        case 82: return launch_mbn<T, 8, 2>(a, batch, s);
        case 101: return launch_mbn<T, 10, 1>(a, batch, s);
        case 102: return launch_mbn<T, 10, 2>(a, batch, s);
        case 121: return launch_mbn<T, 12, 1>(a, batch, s);
        case 122: return launch_mbn<T, 12, 2>(a, batch, s);
    }
    yr_set_error("mbn: widths Cexp=%d Cout=%d unsupported", op.se_reduced, op.cout);
    return YR_ERR_ARG;
}

int yr_launch_mbn(const yr_op& op, int batch, hipStream_t s) {
    if (op.dtype == YR_BF16) return launch_mbn_t<yr_bf16>(op, batch, s);
    return launch_mbn_t<yr_f16>(op, batch, s);
}
