// Fused inverted-residual block: expand 1x1 + BN + act -> depthwise 3x3 (stride 1|2, TF SAME) +
// BN + act -> project 1x1 + BN (+ residual), one kernel, the expanded tensor never leaves the CU.
// Replaces, for MobileNetV2's block_* layers [3P] (reference code/yolo3/override.py:339 ->
// tf.keras.applications.MobileNetV2) and SE-free MBConv blocks (code/yolo3/efficientnet.py:467-536),
// the TF kernel chain Conv2D, FusedBatchNormV3, Relu6, DepthwiseConv2dNative, FusedBatchNormV3, Relu6,
// Conv2D, FusedBatchNormV3, AddV2 - whose 6x-expanded intermediates are 81 % of the unfused path's
// HBM traffic.
//
// One workgroup (4 waves) = one TH x TW tile of output pixels of one image, all output channels.
//   LDS: Xs = the input halo tile ((TH-1)*S+3) x ((TW-1)*S+3) pixels x Cin   (read from HBM once;
//        also the residual source), Es = the current 48-channel chunk of the expanded halo tile.
//   Per 48-wide chunk of expanded channels:
//     expand : Es[halo px][48] = act(BN(Xs[halo px][Cin] * We))      fp32 MFMA 16x16x4, pixels x chunk
//     dw+proj: each lane computes the 3x3 depthwise output for (its pixel, 4 channels) straight into
//              the MFMA operand layout (lane = pixel l&15, k-group l>>4) and feeds
//              acc[out px][Cout] += D[out px][48] * Wp[48][Cout]  - accumulators stay in registers.
//   Halo pixels outside the image are ZERO in Es (TF pads the expanded tensor, not the input).
#include "yr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Expanded channels per chunk: 48 for stride 1 (every MobileNetV2 / EfficientNet width is a multiple), 32 for
// stride 2, whose 4.5x larger halo tile would otherwise cap residency at 3 workgroups per CU (measured:
// block_1 0.52 -> 0.44 ms with 32, block_2 0.35 -> 0.39 ms with 32).  Es row stride = EC + 4 floats
// (an odd number of 16-byte slots => conflict-free ds_read_b128 across consecutive pixels).
#define MB_CHUNK(S) ((S) == 2 ? 32 : 48)

struct MbArgs {
    const float* x; float* out;
    const float* wet; const float* se; const float* he;     // expand: Wt[Cexp][kpi], scale/shift [ldE]
    const float* wdw; const float* sd; const float* hd;     // depthwise: [9][ldE], scale/shift [ldE]
    const float* wpt; const float* sp; const float* hp;     // project: Wt[Cout][ldE], scale/shift [Cout..]
    int Hi, Wi, Ho, Wo, Cin, Cexp, Cout, ld_in, ld_out, ldE, kpi;
    int pad_t, pad_l, tiles_x;
    int has_expand, has_res, act;
};

template <int TH, int TW, int S, int CTO>
__global__ __launch_bounds__(256, (CTO <= 2 ? 4 : 1)) void mbconv_kernel(MbArgs a) {  // narrow: <= 128 VGPRs -> 4 WGs/CU
    constexpr int MB_EC = MB_CHUNK(S), MB_ECT = MB_EC / 16, MB_LDES = MB_EC + 4;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, PH = IH * IW;
    constexpr int OPX = TH * TW, NMT_O = OPX / 16, NMT_H = (PH + 15) / 16;
    constexpr int MTO = (NMT_O + 3) / 4;
    static_assert(OPX % 16 == 0, "tile must hold whole 16-pixel MFMA tiles");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ldx = a.kpi + 4;
    float* Xs = lds;                                  // [PH][ldx]
    float* Es = Xs + PH * ldx;                        // [PH][MB_LDES]          (unused when !has_expand)
    float* Ps = Es + (a.has_expand ? PH * MB_LDES : 0);  // chunk params: wd[9][48], sd, hd, se, he [48]
    // Weight fragments are read straight from global memory (tiny, L1/L2-resident for the layers this
    // kernel is used on); staging them per chunk in LDS measured slower (less occupancy, one more phase).

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int b = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;

    // ---- 1. input halo tile -> LDS (zero outside the image and beyond Cin)
    //      Loads are issued in batches of XB before the first LDS store of the batch: otherwise every
    //      iteration pays a full HBM round trip (store-after-load ordering), measured 1.9x on the stem.
    {
        const int kq = a.kpi >> 2;
        constexpr int XB = 6;
        for (int base = 0; base < PH * kq; base += 256 * XB) {
            float4 v[XB];
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int idx = base + u * 256 + tid;
                const int p = idx / kq, q = idx - p * kq;
                const int hy = p / IW, hx = p - hy * IW;
                const int iy = iy0 + hy, ix = ix0 + hx;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < PH * kq && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) {
                    v[u] = *reinterpret_cast<const float4*>(a.x + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * a.ld_in + q * 4);
                    const int rem = a.Cin - q * 4;
                    if (rem < 4) { v[u].w = 0.f; if (rem < 3) v[u].z = 0.f; if (rem < 2) v[u].y = 0.f; }
                }
            }
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int idx = base + u * 256 + tid;
                if (idx < PH * kq) {
                    const int p = idx / kq, q = idx - p * kq;
                    *reinterpret_cast<float4*>(Xs + p * ldx + q * 4) = v[u];
                }
            }
        }
    }

    f32x4 acc_o[CTO][MTO];
#pragma unroll
    for (int c = 0; c < CTO; ++c)
#pragma unroll
        for (int m = 0; m < MTO; ++m) acc_o[c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int cto = (a.Cout + 15) >> 4;

    for (int e0 = 0; e0 < a.Cexp; e0 += MB_EC) {
        // ---- 2. chunk parameters -> LDS (zeros beyond Cexp so padded channels contribute act(0)=0)
        {
            constexpr int NP = (13 * MB_EC + 255) / 256;
            float pv[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {  // all loads first, then the LDS stores (one round trip)
                const int i = tid + u * 256;
                const int r = i / MB_EC, ch = i - r * MB_EC;
                const int e = e0 + ch;
                pv[u] = 0.f;
                if (i < 13 * MB_EC && e < a.Cexp) {
                    const float* src = r < 9 ? a.wdw + (size_t)r * a.ldE : (r == 9 ? a.sd : (r == 10 ? a.hd : (r == 11 ? a.se : a.he)));
                    if (r < 11 || a.has_expand) pv[u] = src[e];
                }
            }
#pragma unroll
            for (int u = 0; u < NP; ++u)
                if (tid + u * 256 < 13 * MB_EC) Ps[tid + u * 256] = pv[u];
        }
        __syncthreads();  // Xs (first chunk) and Ps visible; previous chunk's readers of Es are done (loop-end barrier)

        // ---- 3. expand GEMM over this wave's halo pixel tiles -> Es
        if (a.has_expand) {
            // narrow inputs (Cin <= 32): the chunk's weight fragments (<= 2 k-steps x 3 tiles) are loaded once
            // per wave and reused by all of its pixel tiles - otherwise every pixel tile waits for an L2 round trip
            const bool hoist = a.kpi <= 32;
            f32x4 wfe0[MB_ECT], wfe1[MB_ECT];
#pragma unroll
            for (int c = 0; c < MB_ECT; ++c) {
                const int e = e0 + c * 16 + li;
                wfe0[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                wfe1[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (hoist && e < a.Cexp) {
                    if (g * 4 < a.kpi) wfe0[c] = *reinterpret_cast<const f32x4*>(a.wet + (size_t)e * a.kpi + g * 4);
                    if (16 + g * 4 < a.kpi) wfe1[c] = *reinterpret_cast<const f32x4*>(a.wet + (size_t)e * a.kpi + 16 + g * 4);
                }
            }
            for (int mt = wave; mt < NMT_H; mt += 4) {
                const int p = mt * 16 + li;
                const int pc = p < PH ? p : PH - 1;
                f32x4 acc_e[MB_ECT];
#pragma unroll
                for (int c = 0; c < MB_ECT; ++c) acc_e[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (hoist) {
                    f32x4 x0 = (f32x4){0.f, 0.f, 0.f, 0.f}, x1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (g * 4 < a.kpi) x0 = *reinterpret_cast<const f32x4*>(Xs + pc * ldx + g * 4);
                    if (16 + g * 4 < a.kpi) x1 = *reinterpret_cast<const f32x4*>(Xs + pc * ldx + 16 + g * 4);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < MB_ECT; ++c)
                            acc_e[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfe0[c][s], x0[s], acc_e[c], 0, 0, 0);
                    if (a.kpi > 16) {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int c = 0; c < MB_ECT; ++c)
                                acc_e[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfe1[c][s], x1[s], acc_e[c], 0, 0, 0);
                    }
                }
                for (int k0 = 0; k0 < (hoist ? 0 : a.kpi); k0 += 16) {
                    const int k = k0 + g * 4;
                    f32x4 xf = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (k < a.kpi) xf = *reinterpret_cast<const f32x4*>(Xs + pc * ldx + k);
                    f32x4 wf[MB_ECT];
#pragma unroll
                    for (int c = 0; c < MB_ECT; ++c) {
                        const int e = e0 + c * 16 + li;
                        wf[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (e < a.Cexp && k < a.kpi) wf[c] = *reinterpret_cast<const f32x4*>(a.wet + (size_t)e * a.kpi + k);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < MB_ECT; ++c)
                            acc_e[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][s], xf[s], acc_e[c], 0, 0, 0);
                }
                if (p < PH) {
                    const int hy = p / IW, hx = p - hy * IW;
                    const int iy = iy0 + hy, ix = ix0 + hx;
                    const bool inside = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
#pragma unroll
                    for (int c = 0; c < MB_ECT; ++c) {
                        const int ch = c * 16 + g * 4;
                        const float4 sc = *reinterpret_cast<const float4*>(Ps + 11 * MB_EC + ch);
                        const float4 sh = *reinterpret_cast<const float4*>(Ps + 12 * MB_EC + ch);
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (inside)
                            v = yr_apply_act4(make_float4(__builtin_fmaf(acc_e[c][0], sc.x, sh.x), __builtin_fmaf(acc_e[c][1], sc.y, sh.y),
                                                          __builtin_fmaf(acc_e[c][2], sc.z, sh.z), __builtin_fmaf(acc_e[c][3], sc.w, sh.w)), a.act);
                        *reinterpret_cast<float4*>(Es + p * MB_LDES + ch) = v;
                    }
                }
            }
            __syncthreads();
        }

        // ---- 4. depthwise 3x3 straight into the MFMA operand + project accumulation
        const float* Ds = a.has_expand ? Es : Xs;
        const int ldd = a.has_expand ? MB_LDES : ldx;
        const int doff = a.has_expand ? 0 : e0;
#pragma unroll
        for (int m = 0; m < MTO; ++m) {
            const int mt = wave + 4 * m;
            if (mt < NMT_O) {
                const int o = mt * 16 + li;
                const int oy = o / TW, ox = o - oy * TW;
                const float* base = Ds + ((oy * S) * IW + ox * S) * ldd + doff;
#pragma unroll
                for (int kc = 0; kc < MB_ECT; ++kc) {
                    const int ch = kc * 16 + g * 4;
                    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.has_expand || e0 + ch < a.kpi) {
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const float4 v = *reinterpret_cast<const float4*>(base + (ky * IW + kx) * ldd + ch);
                                const float4 w = *reinterpret_cast<const float4*>(Ps + (ky * 3 + kx) * MB_EC + ch);
                                d.x = __builtin_fmaf(v.x, w.x, d.x); d.y = __builtin_fmaf(v.y, w.y, d.y);
                                d.z = __builtin_fmaf(v.z, w.z, d.z); d.w = __builtin_fmaf(v.w, w.w, d.w);
                            }
                    }
                    const float4 sc = *reinterpret_cast<const float4*>(Ps + 9 * MB_EC + ch);
                    const float4 sh = *reinterpret_cast<const float4*>(Ps + 10 * MB_EC + ch);
                    d = yr_apply_act4(make_float4(__builtin_fmaf(d.x, sc.x, sh.x), __builtin_fmaf(d.y, sc.y, sh.y),
                                                  __builtin_fmaf(d.z, sc.z, sh.z), __builtin_fmaf(d.w, sc.w, sh.w)), a.act);
                    const float df[4] = {d.x, d.y, d.z, d.w};
                    const int e = e0 + ch;
                    f32x4 wf[CTO];
#pragma unroll
                    for (int c = 0; c < CTO; ++c) {
                        const int n = c * 16 + li;
                        wf[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (c < cto && n < a.Cout && e < a.ldE) wf[c] = *reinterpret_cast<const f32x4*>(a.wpt + (size_t)n * a.ldE + e);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int c = 0; c < CTO; ++c)
                            if (c < cto) acc_o[c][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][s], df[s], acc_o[c][m], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // Es / Ps are rewritten by the next chunk
    }

    // ---- 5. epilogue: project BN (+ residual from the LDS input tile) -> HBM
#pragma unroll
    for (int m = 0; m < MTO; ++m) {
        const int mt = wave + 4 * m;
        if (mt >= NMT_O) continue;
        const int o = mt * 16 + li;
        const int oy = o / TW, ox = o - oy * TW;
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (gy >= a.Ho || gx >= a.Wo) continue;
        float* op = a.out + ((size_t)(b * a.Ho + gy) * a.Wo + gx) * a.ld_out;
        const float* rp = Xs + ((oy * S + a.pad_t) * IW + ox * S + a.pad_l) * ldx;  // block input at the centre tap
#pragma unroll
        for (int c = 0; c < CTO; ++c) {
            const int n = c * 16 + g * 4;
            if (c >= cto || n >= a.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n + r < a.Cout ? n + r : a.Cout - 1;
                v[r] = __builtin_fmaf(acc_o[c][m][r], a.sp[nn], a.hp[nn]);
                if (a.has_res) v[r] += rp[nn];
            }
            if (n + 3 < a.Cout && (a.ld_out & 3) == 0) {
                *reinterpret_cast<float4*>(op + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < a.Cout) op[n + r] = v[r];
            }
        }
    }
}

template <int TH, int TW, int S, int CTO>
static int launch_mb(const MbArgs& a, int batch, hipStream_t s) {
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, PH = IH * IW;
    constexpr int MB_EC = MB_CHUNK(S), MB_LDES = MB_EC + 4;
    const size_t lds = ((size_t)PH * (a.kpi + 4) + (a.has_expand ? (size_t)PH * MB_LDES : 0) + 13 * MB_EC) * sizeof(float);
    YR_REQUIRE(lds <= 160 * 1024, "mbconv: LDS tile of %zu bytes does not fit", lds);
    static bool attr = false;
    if (!attr) {
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)mbconv_kernel<TH, TW, S, CTO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    MbArgs b = a;
    b.tiles_x = (a.Wo + TW - 1) / TW;
    const int tiles_y = (a.Ho + TH - 1) / TH;
    static char nm[48];
    static const int nm_len = snprintf(nm, sizeof(nm), "mbconv_kernel<%d,%d,%d,%d>", TH, TW, S, CTO);
    (void)nm_len;
    yr_note_kernel(nm);
    hipLaunchKernelGGL((mbconv_kernel<TH, TW, S, CTO>), dim3(b.tiles_x * tiles_y, batch), dim3(256), lds, s, b);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <int TH, int TW, int S>
static int launch_mb_cto(const MbArgs& a, int batch, hipStream_t s) {
    const int cto = (a.Cout + 15) / 16;
    if (cto <= 2) return launch_mb<TH, TW, S, 2>(a, batch, s);
    if (cto <= 4) return launch_mb<TH, TW, S, 4>(a, batch, s);
    if (cto <= 8) return launch_mb<TH, TW, S, 8>(a, batch, s);
    if (cto <= 14) return launch_mb<TH, TW, S, 14>(a, batch, s);
    yr_set_error("mbconv: %d output channels exceed the fused kernel's limit (224)", a.Cout);
    return YR_ERR_ARG;
}

// op fields: src[0] = block input; se_reduced = expanded width; k = 3; stride = 1|2; act = expand/DW activation;
// wgt/scale/shift = expand (null => no expand, e.g. MobileNetV2 block 0); wgt2 = [9][ldE] DW weights followed by
// DW scale [ldE] and shift [ldE]; b1 = project Wt[Cout][ldE]; b2 = project scale [ldo] followed by shift [ldo],
// ldo = round_up(Cout,4); res (optional) must be the block input itself.
int yr_launch_mbconv(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY, "mbconv: needs one identity source");
    const yr_src& in = op.src[0];
    YR_REQUIRE(op.k == 3 && (op.stride == 1 || op.stride == 2), "mbconv: only 3x3 stride 1|2 is fused");
    YR_REQUIRE(in.ptr && op.out && op.wgt2 && op.b1 && op.b2, "mbconv: null pointer");
    YR_REQUIRE(in.ld % 4 == 0 && in.c == op.cin && in.ld >= yr_round_up(in.c, 4), "mbconv: bad input stride");
    YR_REQUIRE(((uintptr_t)in.ptr | (uintptr_t)op.wgt | (uintptr_t)op.wgt2 | (uintptr_t)op.b1) % 16 == 0, "mbconv: pointers must be 16-byte aligned");
    MbArgs a;
    YR_REQUIRE(op.dtype == YR_F32 && op.out_dtype == YR_F32 && in.dtype == YR_F32, "mbconv: float32 only");
    a.x = (const float*)in.ptr; a.out = (float*)op.out;
    a.has_expand = op.wgt != nullptr;
    a.Cin = in.c; a.Cexp = a.has_expand ? op.se_reduced : in.c; a.Cout = op.cout;
    YR_REQUIRE(a.Cexp >= 1 && (a.has_expand || op.se_reduced == in.c), "mbconv: bad expanded width");
    a.ldE = yr_round_up(a.Cexp, 4); a.kpi = yr_round_up(in.c, 4);
    a.wet = op.wgt; a.se = op.scale; a.he = op.shift;
    YR_REQUIRE(!a.has_expand || (op.scale && op.shift), "mbconv: expand BN missing");
    a.wdw = op.wgt2; a.sd = op.wgt2 + 9 * a.ldE; a.hd = a.sd + a.ldE;
    a.wpt = op.b1; a.sp = op.b2; a.hp = op.b2 + yr_round_up(op.cout, 4);
    a.Hi = in.h; a.Wi = in.w; a.Ho = (in.h + op.stride - 1) / op.stride; a.Wo = (in.w + op.stride - 1) / op.stride;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "mbconv: output dims mismatch");
    a.ld_in = in.ld; a.ld_out = op.out_ld;
    YR_REQUIRE(op.out_ld >= op.cout, "mbconv: out_ld too small");
    const int pth = (a.Ho - 1) * op.stride + 3 - in.h, ptw = (a.Wo - 1) * op.stride + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.has_res = op.res != nullptr;
    if (a.has_res) YR_REQUIRE(op.res == in.ptr && op.stride == 1 && in.c == op.cout, "mbconv: the residual must be the block input (stride 1, Cin == Cout)");
    a.act = op.act;
    a.tiles_x = 0;
    if (op.stride == 1) {
        // no expand stage (MobileNetV2 block 0): the work per pixel is tiny, use big tiles
        if (!a.has_expand && a.kpi <= 32) return launch_mb_cto<16, 16, 1>(a, batch, s);
        return launch_mb_cto<8, 8, 1>(a, batch, s);
    }
    // stride 2: the halo is 4.5x the output tile; 4x8 output tiles keep it (and Es) small
    return launch_mb_cto<4, 8, 2>(a, batch, s);
}
