// Network entry of the 16-bit plans on the matrix pipe: stem Conv2D 3x3 s2 (Cin = 3) + BN + act -> depthwise 3x3 s1 + BN +
// act -> project 1x1 + BN, one kernel, for MobileNetV2's Conv1 + expanded_conv and the SE-free EfficientNet-lite entry
// (reference code/yolo3/override.py:339, code/yolo3/efficientnet.py:467-536 with expand ratio 1).
//
// stemblock.hip does this lane-per-pixel on the float32 pipe - right for the float32 plans, where fp32 MFMA has no higher
// peak than packed FMA - and was also what the 16-bit plans ran: 0.31 ms per 128 images at 416 (EfficientNet-lite0,
// 59 TFLOP/s of float32 FMAs, profiles/r03_perop_c3*), three times what its 443 MB of traffic cost.  Of its 1664 MACs per
// stem pixel, 864 are the stem and 512 the projection - GEMMs with K = 27 and K = C1 that a 16-bit plan may run as
// v_mfma_f32_16x16x32_bf16 / _f16 like every other 1x1 convolution of the plan (16-bit operands, float32 accumulation):
//
//   One workgroup (4 waves) = a 14 x 14 tile of block outputs = a 16 x 16 halo tile of stem outputs (halo row = one
//   16-pixel MFMA tile; a wave takes four rows).
//   1. stem: the B operand of pixel (lane & 15), k group (lane >> 4) is 8 of the pixel's 27 window values.  The k space is
//      ORDERED so that a group is contiguous in the image: group g < 3 = image row 2y + g, values 0..7 of the row's 9
//      (kx, c); group 3 = value 8 of the three rows and five zeros - two 16-byte loads (one 8-byte load of a uint8 image)
//      per lane and tile, through a buffer descriptor (rows above / below the image read as zeros; columns beyond a row's
//      end are masked).  uint8 pixels are exact in either 16-bit type (the /255 goes into the BatchNorm scale); float32
//      pixels are rounded to the plan's type like every other MFMA operand.  C1P / 16 MFMAs per tile, BatchNorm,
//      activation, zero outside the map (TF pads the depthwise conv's INPUT), rounded into Es[256][C1P] in LDS.
//   2. depthwise + projection: a wave owns 3-4 output rows; lane = (column, 8 channels) - the projection's B-operand
//      layout - walks down its rows with a ring of three accumulator rows: per input row three 16-byte LDS reads feed the
//      nine taps of up to three output rows.  A finished row is rounded and IS the MFMA operand: no second LDS pass.
//      Project BN, 8-byte stores (the four k-group lanes of a pixel cover 32 contiguous bytes).
// Arithmetic relative to the float32 oracle: operands rounded where the plan's other layers round theirs (weights, stem
// output, depthwise output); oracle/params.py QuantStore rounds the same tensors.
#include "yr_common.h"

typedef float sbh_f4 __attribute__((ext_vector_type(4)));
typedef float sbh_f2 __attribute__((ext_vector_type(2)));
typedef float sbh_f8 __attribute__((ext_vector_type(8)));
typedef unsigned sbh_u4 __attribute__((ext_vector_type(4)));
typedef unsigned sbh_u2 __attribute__((ext_vector_type(2)));
template <class T> using sbh_v8 = T __attribute__((ext_vector_type(8)));
template <class T> using sbh_v4 = T __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t sbh_rsrc;

struct SbhArgs {
    const void* in;      // [B][Hi][Wi][3] float32 in [0,1] or uint8
    float in_scale;      // 1 | 1/255 (uint8): multiplies the stem BN scale
    void* out;           // T [B][Ho][Wo][ld_out]
    const void* ws;      // T [C1P][32]: stem weights, k space in the order described above
    const float* ssc;    // [C1P] stem BN scale
    const float* ssh;    // [C1P] stem BN shift
    const float* wd;     // [10][C1P]: nine depthwise taps times the BN scale | BN shift
    const void* wp;      // T [COP][C1P]: projection weights
    const float* bp;     // [2][COP]: projection BN scale | shift
    int Hi, Wi, Ho, Wo, ld_out, pad_t, pad_l, act, tiles_x, tiles_y;
};

template <class T>
__device__ __forceinline__ sbh_f4 sbh_mfma(sbh_u4 a, sbh_u4 b, sbh_f4 c) {
    if constexpr (yr_elem<T>::dtype == YR_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sbh_v8<__bf16>, a), __builtin_bit_cast(sbh_v8<__bf16>, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sbh_v8<_Float16>, a), __builtin_bit_cast(sbh_v8<_Float16>, b), c, 0, 0, 0);
}

template <bool RELU6, class T>
__device__ __forceinline__ float sbh_act(float v, int act) {
    if constexpr (RELU6) return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f);
    else return yr_apply_act_t<T>(v, act);
}

// NC1 = C1P / 16 (2 | 4), NCO = COP / 16 (1 | 2)
template <class T, int NC1, int NCO, bool RELU6, bool IN8>
__global__ __launch_bounds__(256) void stemblock_h_kernel(SbhArgs a) {
    constexpr int C1P = 16 * NC1, KS = C1P / 32, LDE = C1P + 8;   // Es row pitch 80 | 144 bytes: conflict-free 16-byte rows
    extern __shared__ __attribute__((aligned(16))) char sbh_lds[];
    T* Es = reinterpret_cast<T*>(sbh_lds);                                   // [256 halo pixels][LDE]
    float* Wd = reinterpret_cast<float*>(sbh_lds + 256 * LDE * sizeof(T));   // [10][C1P]
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
    const int tpi = a.tiles_x * a.tiles_y;
    const int t = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = t / tpi, r = t - b * tpi;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int sy0 = ty * 14 - 1, sx0 = tx * 14 - 1;   // the halo tile's origin in the stem-output map

    for (int i = tid; i < 10 * C1P; i += 256) Wd[i] = a.wd[i];

    // ---- 1. stem
    {
        sbh_u4 wsf[NC1];
        sbh_f4 ssc[NC1], ssh[NC1];
#pragma unroll
        for (int j = 0; j < NC1; ++j) {
            wsf[j] = *reinterpret_cast<const sbh_u4*>(reinterpret_cast<const T*>(a.ws) + (size_t)(16 * j + li) * 32 + 8 * g);
            ssc[j] = *reinterpret_cast<const sbh_f4*>(a.ssc + 16 * j + 4 * g) * a.in_scale;
            ssh[j] = *reinterpret_cast<const sbh_f4*>(a.ssh + 16 * j + 4 * g);
        }
        constexpr unsigned ES = IN8 ? 1u : 4u;   // bytes per image element
        const sbh_rsrc img = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<const char*>(a.in) + (size_t)b * a.Hi * a.Wi * 3 * ES), 0, (unsigned)(a.Hi * a.Wi * 3) * ES, 0x00020000);
        const int sx = sx0 + li;
        const int ixb = 2 * sx - a.pad_l;
        // which of the window's three columns exist (the rows are the descriptor's business); as masks of the packed pairs
        // (0,1) (2,3) (4,5) (6,7) of the lane's 8 values: kx = 0 0 0 1 1 1 2 2 (groups 0..2) | 2 2 2 - - - - - (group 3)
        const bool v0 = (unsigned)ixb < (unsigned)a.Wi, v1 = (unsigned)(ixb + 1) < (unsigned)a.Wi, v2 = (unsigned)(ixb + 2) < (unsigned)a.Wi;
        unsigned m[4];
        if (g < 3) {
            m[0] = v0 ? 0xffffffffu : 0u;
            m[1] = (v0 ? 0x0000ffffu : 0u) | (v1 ? 0xffff0000u : 0u);
            m[2] = v1 ? 0xffffffffu : 0u;
            m[3] = v2 ? 0xffffffffu : 0u;
        } else {
            m[0] = m[1] = v2 ? 0xffffffffu : 0u;
            m[2] = m[3] = 0u;
        }
        const bool g3 = g == 3;
        // element offsets (relative to row 2 sy - pad_t of the image) of the lane's 8 contiguous values and of group 3's two others
        const int rowe = a.Wi * 3;
        const int e_main = ixb * 3 + (g3 ? 8 : g * rowe);
        sbh_u4 xa[4], xb[4];     // float32: 8 values; uint8: xa[.].x, .y = the 8 bytes
        unsigned x1[4], x2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int hy = wave * 4 + q;
            const int e0 = (2 * (sy0 + hy) - a.pad_t) * rowe + e_main;
            // An access that is only partly inside the descriptor's range is not split by the hardware: the one such case -
            // the image's last row, last window column: values 6, 7 lie behind the end of the image (even sizes: always a
            // whole pair) - gets its own loads.
            if constexpr (IN8) {
                const unsigned lo = __builtin_amdgcn_raw_buffer_load_b32(img, (unsigned)e0, 0, 0);
                const unsigned m1 = __builtin_amdgcn_raw_buffer_load_b16(img, (unsigned)e0 + 4u, 0, 0);
                const unsigned m2 = __builtin_amdgcn_raw_buffer_load_b16(img, (unsigned)e0 + 6u, 0, 0);
                xa[q] = (sbh_u4){lo, (m1 & 0xffffu) | (m2 << 16), 0u, 0u};
                x1[q] = __builtin_amdgcn_raw_buffer_load_b8(img, (unsigned)(e0 + rowe), 0, 0);
                x2[q] = __builtin_amdgcn_raw_buffer_load_b8(img, (unsigned)(e0 + 2 * rowe), 0, 0);
            } else {
                xa[q] = __builtin_bit_cast(sbh_u4, __builtin_amdgcn_raw_buffer_load_b128(img, (unsigned)e0 * 4u, 0, 0));
                const sbh_u2 h0 = __builtin_bit_cast(sbh_u2, __builtin_amdgcn_raw_buffer_load_b64(img, (unsigned)e0 * 4u + 16u, 0, 0));
                const sbh_u2 h1 = __builtin_bit_cast(sbh_u2, __builtin_amdgcn_raw_buffer_load_b64(img, (unsigned)e0 * 4u + 24u, 0, 0));
                xb[q] = (sbh_u4){h0.x, h0.y, h1.x, h1.y};
                x1[q] = __builtin_amdgcn_raw_buffer_load_b32(img, (unsigned)(e0 + rowe) * 4u, 0, 0);
                x2[q] = __builtin_amdgcn_raw_buffer_load_b32(img, (unsigned)(e0 + 2 * rowe) * 4u, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int hy = wave * 4 + q;
            sbh_f8 v;
            if constexpr (IN8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = (float)((xa[q].x >> (8 * i)) & 0xffu);
                    v[4 + i] = (float)((xa[q].y >> (8 * i)) & 0xffu);
                }
                if (g3) { v[1] = (float)(x1[q] & 0xffu); v[2] = (float)(x2[q] & 0xffu); }
            } else {
                const sbh_f4 lo = __builtin_bit_cast(sbh_f4, xa[q]), hi = __builtin_bit_cast(sbh_f4, xb[q]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = lo[i]; v[4 + i] = hi[i]; }
                if (g3) { v[1] = __builtin_bit_cast(float, x1[q]); v[2] = __builtin_bit_cast(float, x2[q]); }
            }
            sbh_u4 frag = __builtin_bit_cast(sbh_u4, __builtin_convertvector(v, sbh_v8<T>));
#pragma unroll
            for (int i = 0; i < 4; ++i) frag[i] &= m[i];
            const bool inmap = (unsigned)(sy0 + hy) < (unsigned)a.Ho && (unsigned)sx < (unsigned)a.Wo;
            T* erow = Es + (size_t)(hy * 16 + li) * LDE + 4 * g;
#pragma unroll
            for (int j = 0; j < NC1; ++j) {
                const sbh_f4 acc = sbh_mfma<T>(wsf[j], frag, (sbh_f4){0.f, 0.f, 0.f, 0.f});
                sbh_f4 y = __builtin_elementwise_fma(acc, ssc[j], ssh[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = inmap ? sbh_act<RELU6, T>(y[i], a.act) : 0.f;
                *reinterpret_cast<sbh_v4<T>*>(erow + 16 * j) = __builtin_convertvector(y, sbh_v4<T>);
            }
        }
    }
    __syncthreads();

    // ---- 2. depthwise + projection: rows r0 .. r0 + nrow - 1 of the 14 (4 4 3 3), columns li < 14, channels 32 ks + 8 g ..
    const int r0 = wave * 4 - (wave == 3 ? 1 : 0), nrow = wave < 2 ? 4 : 3;
    sbh_f4 pacc[NCO][4];
#pragma unroll
    for (int n = 0; n < NCO; ++n)
#pragma unroll
        for (int o = 0; o < 4; ++o) pacc[n][o] = (sbh_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c0 = 32 * ks + 8 * g;
        sbh_f2 tw[9][4], sd[4];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const sbh_f4 lo = *reinterpret_cast<const sbh_f4*>(Wd + tp * C1P + c0), hi = *reinterpret_cast<const sbh_f4*>(Wd + tp * C1P + c0 + 4);
            tw[tp][0] = (sbh_f2){lo[0], lo[1]}; tw[tp][1] = (sbh_f2){lo[2], lo[3]}; tw[tp][2] = (sbh_f2){hi[0], hi[1]}; tw[tp][3] = (sbh_f2){hi[2], hi[3]};
        }
        {
            const sbh_f4 lo = *reinterpret_cast<const sbh_f4*>(Wd + 9 * C1P + c0), hi = *reinterpret_cast<const sbh_f4*>(Wd + 9 * C1P + c0 + 4);
            sd[0] = (sbh_f2){lo[0], lo[1]}; sd[1] = (sbh_f2){lo[2], lo[3]}; sd[2] = (sbh_f2){hi[0], hi[1]}; sd[3] = (sbh_f2){hi[2], hi[3]};
        }
        sbh_u4 wpf[NCO];
#pragma unroll
        for (int n = 0; n < NCO; ++n) wpf[n] = *reinterpret_cast<const sbh_u4*>(reinterpret_cast<const T*>(a.wp) + (size_t)(16 * n + li) * C1P + c0);
        sbh_f2 acc[3][4];
        const T* ecol = Es + (size_t)(r0 * 16 + li) * LDE + c0;
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {   // input rows r0 + rr feed output rows rr - ky (the 6th only a 4th output row)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const sbh_u4 raw = *reinterpret_cast<const sbh_u4*>(ecol + (size_t)(rr * 16 + kx) * LDE);
                // (whole-vector casts: hipcc 7.2 folds a bit_cast of raw[c] inside an unrolled loop to element 0)
                const sbh_f8 xf = __builtin_convertvector(__builtin_bit_cast(sbh_v8<T>, raw), sbh_f8);
                sbh_f2 x[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) x[c] = (sbh_f2){xf[2 * c], xf[2 * c + 1]};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int o = rr - ky;
                    if (o >= 0 && o < 4) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[o % 3][c] = __builtin_elementwise_fma(x[c], tw[ky * 3 + kx][c], ky == 0 && kx == 0 ? (sbh_f2){0.f, 0.f} : acc[o % 3][c]);
                    }
                }
            }
            if (rr >= 2) {
                const int o = rr - 2;
                sbh_f8 d;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const sbh_f2 y = acc[o % 3][c] + sd[c];
                    d[2 * c] = sbh_act<RELU6, T>(y.x, a.act);
                    d[2 * c + 1] = sbh_act<RELU6, T>(y.y, a.act);
                }
                const sbh_u4 frag = __builtin_bit_cast(sbh_u4, __builtin_convertvector(d, sbh_v8<T>));
#pragma unroll
                for (int n = 0; n < NCO; ++n) pacc[n][o] = sbh_mfma<T>(wpf[n], frag, pacc[n][o]);
            }
        }
    }
    // ---- 3. project BN, stores: lane = pixel (row, li), couts 16 n + 4 g + 0..3
    T* outp = reinterpret_cast<T*>(a.out) + (size_t)b * a.Ho * a.Wo * a.ld_out;
    const int gx = tx * 14 + li;
#pragma unroll
    for (int n = 0; n < NCO; ++n) {
        const int co = 16 * n + 4 * g;
        const sbh_f4 psc = *reinterpret_cast<const sbh_f4*>(a.bp + co), psh = *reinterpret_cast<const sbh_f4*>(a.bp + 16 * NCO + co);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int gy = ty * 14 + r0 + o;
            if (o < nrow && li < 14 && gy < a.Ho && gx < a.Wo && co < a.ld_out) {
                const sbh_f4 y = __builtin_elementwise_fma(pacc[n][o], psc, psh);
                *reinterpret_cast<sbh_v4<T>*>(outp + ((size_t)gy * a.Wo + gx) * a.ld_out + co) = __builtin_convertvector(y, sbh_v4<T>);
            }
        }
    }
}

template <class T, int NC1, int NCO>
static int launch_sbh(const SbhArgs& a, bool in8, int batch, hipStream_t s) {
    constexpr size_t lds = (size_t)256 * (16 * NC1 + 8) * 2 + (size_t)10 * 16 * NC1 * 4;
    static char nm[2][2][56];
    static bool named = false;
    if (!named) {
        for (int r = 0; r < 2; ++r)
            for (int u = 0; u < 2; ++u)
                snprintf(nm[r][u], sizeof(nm[r][u]), "stemblock_h_kernel<%s,%d,%d,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), NC1, NCO, r, u);
        named = true;
    }
    const bool relu6 = a.act == YR_ACT_RELU6;
    yr_note_kernel(nm[relu6 ? 1 : 0][in8 ? 1 : 0]);
    const dim3 grid((unsigned)(batch * a.tiles_x * a.tiles_y));
    if (in8) {
        if (relu6) hipLaunchKernelGGL((stemblock_h_kernel<T, NC1, NCO, true, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((stemblock_h_kernel<T, NC1, NCO, false, true>), grid, dim3(256), lds, s, a);
    } else {
        if (relu6) hipLaunchKernelGGL((stemblock_h_kernel<T, NC1, NCO, true, false>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((stemblock_h_kernel<T, NC1, NCO, false, false>), grid, dim3(256), lds, s, a);
    }
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// op fields (16-bit plans, the compiler's matrix-pipe layout - op.scale is set, which the float32-pipe layout never does):
// src[0] = dense 3-channel image (float32 | uint8); se_reduced = C1; cout; k = 3; stride = 2; act; C1P = round_up(C1, 32),
// COP = round_up(cout, 16), all zero padded:
//   wgt = stem T [C1P][32] (k order: rows 0..2 x values 0..7 | value 8 of rows 0..2 | 0 x 5);  scale / shift = stem BN [C1P]
//   wgt2 = [10][C1P] depthwise taps times the BN scale | BN shift;  b1 = project T [COP][C1P];  b2 = project BN [2][COP]
template <class T>
static int launch_stemblock_h_t(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY && op.src[0].c == 3 && op.src[0].ld == 3 && (op.src[0].dtype == YR_F32 || op.src[0].dtype == YR_U8),
               "stemblock: needs one dense 3-channel float32 (or uint8) source");
    const bool in8 = op.src[0].dtype == YR_U8;
    YR_REQUIRE(op.k == 3 && op.stride == 2, "stemblock: the stem is 3x3 stride 2");
    const yr_src& in = op.src[0];
    YR_REQUIRE(in.h % 2 == 0 && in.w % 2 == 0, "stemblock (matrix pipe): even image sizes only (%d x %d)", in.h, in.w);
    YR_REQUIRE(op.out_dtype == op.dtype && op.out_ld % 8 == 0, "stemblock: the output has the op's dtype, out_ld %% 8 == 0");
    YR_REQUIRE(in.ptr && op.out && op.wgt && op.wgt2 && op.scale && op.shift && op.b1 && op.b2, "stemblock: null pointer");
    YR_REQUIRE(op.se_reduced >= 1 && op.se_reduced <= 64 && op.cout >= 1 && op.cout <= 32, "stemblock (matrix pipe): widths C1=%d Cout=%d unsupported", op.se_reduced, op.cout);
    YR_REQUIRE((long long)in.h * in.w * 3 * 4 < (1ll << 31), "stemblock: image too large");
    SbhArgs a;
    a.in = in.ptr; a.in_scale = in8 ? 1.0f / 255.0f : 1.0f; a.out = op.out;
    a.ws = op.wgt; a.ssc = op.scale; a.ssh = op.shift; a.wd = op.wgt2; a.wp = op.b1; a.bp = op.b2;
    a.Hi = in.h; a.Wi = in.w; a.Ho = (in.h + 1) / 2; a.Wo = (in.w + 1) / 2;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w && op.out_ld >= op.cout, "stemblock: output dims mismatch");
    a.ld_out = op.out_ld;
    const int pth = (a.Ho - 1) * 2 + 3 - in.h, ptw = (a.Wo - 1) * 2 + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.act = op.act;
    a.tiles_x = (a.Wo + 13) / 14; a.tiles_y = (a.Ho + 13) / 14;
    const int nc1 = yr_round_up(op.se_reduced, 32) / 16, nco = yr_round_up(op.cout, 16) / 16;
    if (nc1 == 2) return nco == 1 ? launch_sbh<T, 2, 1>(a, in8, batch, s) : launch_sbh<T, 2, 2>(a, in8, batch, s);
    return nco == 1 ? launch_sbh<T, 4, 1>(a, in8, batch, s) : launch_sbh<T, 4, 2>(a, in8, batch, s);
}

int yr_launch_stemblock_h(const yr_op& op, int batch, hipStream_t s) {
    if (yr_stemxp_takes(op)) return yr_launch_stemxp(op, batch, s);   // (stems of more than 32 channels: the register-chained form, mbxr_h.hip)
    if (op.dtype == YR_BF16) return launch_stemblock_h_t<yr_bf16>(op, batch, s);
    if (op.dtype == YR_F16) return launch_stemblock_h_t<yr_f16>(op, batch, s);
    yr_set_error("stemblock (matrix pipe): 16-bit plans only");
    return YR_ERR_ARG;
}
