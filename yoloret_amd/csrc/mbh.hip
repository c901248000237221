// Fused inverted-residual block on 16-bit activations (bfloat16 / float16 storage): expand 1x1 + BN + act ->
// depthwise KxK (K = 3|5, stride 1|2, TF 'SAME') + BN + act -> project 1x1 + BN (+ residual), ONE kernel; the expanded
// tensors (6x the block's input, 80 % of the unfused chain's HBM traffic) never leave the CU.  Both 1x1 convolutions run
// on v_mfma_f32_16x16x32_bf16 / _f16 (float32 accumulate), the depthwise stage on packed float32 FMAs from an LDS tile.
// Replaces, for MobileNetV2's block_* [3P] (reference code/yolo3/override.py:339 -> tf.keras.applications.MobileNetV2)
// and the SE-free MBConv blocks (code/yolo3/efficientnet.py:467-536), the TF kernel chain Conv2D, FusedBatchNormV3,
// Relu6, DepthwiseConv2dNative, FusedBatchNormV3, Relu6, Conv2D, FusedBatchNormV3, AddV2 - in the reduced-precision
// plans of BASELINE.json configs 3 and 5, where the float32 lane-per-pixel kernels (mblane.hip) are bound by the
// float32 pipe they do their 1x1 convs on, and the unfused chain by three launches per block.
//
// One workgroup (4 waves) = one th x tw tile of output pixels of one image (tile size chosen per layer by the host),
// all output channels.  LDS:
//   Xs [PH][KP+8]   the input halo tile, 16-bit, PH = ((th-1)S+K) x ((tw-1)S+K) pixels, KP = round_up(Cin, 32)
//                   (read from HBM once; also the residual source);
//   Es [PH][32+4]   the current 32-channel chunk of the EXPANDED halo tile, float32 (zero outside the image: TF pads
//                   the expanded tensor, not the input);
//   Ps [2][..]      the chunk's depthwise weights and the expand / depthwise BN parameters, double buffered.
// Per chunk of 32 expanded channels (two barriers):
//   expand : Es[p][32] = act(BN(Xs[p][:] . We))         MFMA, 16-pixel tiles of the halo round-robin over the waves
//   dw+proj: every lane computes the KxK depthwise result of (its output pixel, 8 channels) in float32 - exactly the
//            MFMA B-operand fragment of the projection - rounds it to the 16-bit type and feeds
//            acc[cout][pixel] += Wp[cout][32] . D[32][pixel]; the projection accumulators stay in registers.
// Epilogue: project BN (+ the block input from Xs) -> one 16-byte store of 8 consecutive couts per lane (the cout-pair
// permutation of pointwise_h.hip).
#include "yr_common.h"

typedef float mbh_f4 __attribute__((ext_vector_type(4)));
typedef float mbh_f8 __attribute__((ext_vector_type(8)));
typedef unsigned mbh_u4 __attribute__((ext_vector_type(4)));
template <class T> using mbh_v8 = T __attribute__((ext_vector_type(8)));

#define MBH_EC 32                 // expanded channels per chunk = one k-step of the projection MFMA
#define MBH_LDE (MBH_EC + 4)      // Es row stride in floats: an odd number of 16-byte slots

struct MbhArgs {
    const void* x; void* out;                                // T
    const void* we;                                          // expand Wt[CexpP][KP] (T)
    const float* prm;                                        // [K*K + 4][CexpP]: depthwise taps | dw BN scale | dw BN shift |
                                                             //                   expand BN scale | expand BN shift
    const void* wp; const float* sp; const float* hp;        // project Wt[Cout][CexpP] (T), BN scale / shift [Cout]
    int Hi, Wi, Ho, Wo, Cin, CexpP, Cout, ld_in, ld_out, KP;
    int pad_t, pad_l, th, tw, tiles_x, tiles_y, ih, iw, PH, OPX;
    int has_res, act;
};

template <class T>
__device__ __forceinline__ mbh_f4 mbh_mfma(mbh_u4 w, mbh_u4 x, mbh_f4 acc) {
    if constexpr (yr_elem<T>::dtype == YR_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mbh_v8<__bf16>, w), __builtin_bit_cast(mbh_v8<__bf16>, x), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mbh_v8<_Float16>, w), __builtin_bit_cast(mbh_v8<_Float16>, x), acc, 0, 0, 0);
}

template <bool RELU6>
__device__ __forceinline__ float mbh_act(float v, int act) {
    if (RELU6) return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f);   // clamp in one instruction (== min(max(v,0),6) for finite v)
    return yr_apply_act(v, act);
}

// CP: cout tile pairs (Cout <= 32*CP); MTO: 16-pixel output tiles per wave (th*tw <= 64*MTO)
template <class T, int K, int S, int CP, int MTO, bool RELU6>
__global__ __launch_bounds__(256, 2) void mbh_kernel(MbhArgs a) {
    constexpr int CT = 2 * CP, KK = K * K;
    constexpr int PSZ = (KK + 4) * MBH_EC;              // floats per parameter buffer: dw taps | sd | hd | se | he
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int ldx = a.KP + 8;                           // Xs row stride in elements (16-byte aligned, conflict-free)
    T* Xs = reinterpret_cast<T*>(lds_raw);
    float* Es = reinterpret_cast<float*>(lds_raw + (((size_t)a.PH * ldx * sizeof(T) + 15) & ~(size_t)15));
    float* Ps = Es + (size_t)a.PH * MBH_LDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int tpi = a.tiles_x * a.tiles_y;
    const int t = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int b = t / tpi, r = t - b * tpi;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int oy0 = ty * a.th, ox0 = tx * a.tw;
    const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
    const T* xin = reinterpret_cast<const T*>(a.x) + (size_t)b * a.Hi * a.Wi * a.ld_in;
    const T* we = reinterpret_cast<const T*>(a.we);
    const T* wp = reinterpret_cast<const T*>(a.wp);

    // ---- 1. input halo tile -> LDS (zero outside the image and beyond Cin; pad channels of the source may hold anything).
    //      Loads are issued in batches before the first LDS store of the batch (one HBM round trip per batch).
    {
        const int nq = a.KP >> 3;                       // 16-byte vectors per pixel row
        const int cq = (a.Cin + 7) >> 3;                // ... of which hold real channels
        const int total = a.PH * nq;
        constexpr int XB = 4;
        for (int base = 0; base < total; base += 256 * XB) {
            mbh_u4 v[XB];
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int idx = base + u * 256 + tid;
                const int p = idx / nq, q = idx - p * nq;
                const int hy = p / a.iw, hx = p - hy * a.iw;
                const int iy = iy0 + hy, ix = ix0 + hx;
                v[u] = (mbh_u4){0u, 0u, 0u, 0u};
                if (idx < total && q < cq && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) {
                    v[u] = *reinterpret_cast<const mbh_u4*>(xin + ((size_t)iy * a.Wi + ix) * a.ld_in + q * 8);
                    const int cv = a.Cin - q * 8;       // real channels in this vector
#pragma unroll
                    for (int d = 0; d < 4; ++d) v[u][d] = cv >= 2 * d + 2 ? v[u][d] : (cv == 2 * d + 1 ? (v[u][d] & 0xffffu) : 0u);
                }
            }
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int idx = base + u * 256 + tid;
                if (idx < total) {
                    const int p = idx / nq, q = idx - p * nq;
                    *reinterpret_cast<mbh_u4*>(Xs + (size_t)p * ldx + q * 8) = v[u];
                }
            }
        }
    }
    // chunk parameters: [KK taps][32] | dw scale | dw shift | expand scale | expand shift (CexpP is a multiple of 32)
    constexpr int NPV = (PSZ + 255) / 256;
    auto load_params = [&](int e0, float (&pv)[NPV]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NPV; ++u) {
            const int i = tid + u * 256;
            const int rr = i / MBH_EC, ch = i - rr * MBH_EC;
            pv[u] = 0.f;
            if (i < PSZ) {
                pv[u] = a.prm[(size_t)rr * a.CexpP + e0 + ch];
            }
        }
    };
    auto store_params = [&](float* dst, const float (&pv)[NPV]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NPV; ++u)
            if (tid + u * 256 < PSZ) dst[tid + u * 256] = pv[u];
    };
    {
        float pv[NPV];
        load_params(0, pv);
        store_params(Ps, pv);
    }

    mbh_f4 acc_o[CT][MTO];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int m = 0; m < MTO; ++m) acc_o[c][m] = (mbh_f4){0.f, 0.f, 0.f, 0.f};

    // projection weight rows of this lane (cout-pair permutation: MFMA row i of tile c <-> cout (c>>1)*32 + 8*(i>>2) + 4*(c&1) + (i&3))
    const T* wprow[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = (c >> 1) * 32 + 8 * (li >> 2) + 4 * (c & 1) + (li & 3);
        wprow[c] = wp + (size_t)(n < a.Cout ? n : 0) * a.CexpP + 8 * g;
    }
    const int nmt_h = (a.PH + 15) >> 4, nmt_o = (a.OPX + 15) >> 4;
    const int nks = a.KP >> 5;                          // k-steps of the expand GEMM
    const bool hoist = a.KP <= 64;                      // the chunk's expand-weight fragments fit registers: load once per wave
    __syncthreads();                                    // Xs and Ps[0] visible

    const int nchunks = a.CexpP >> 5;
    for (int ci = 0; ci < nchunks; ++ci) {
        const int e0 = ci * MBH_EC;
        const float* Pc = Ps + (ci & 1) * PSZ;
        // prefetches: this chunk's projection fragments, the next chunk's parameters (stored behind the dw phase)
        mbh_u4 wpf[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) wpf[c] = *reinterpret_cast<const mbh_u4*>(wprow[c] + e0);
        float pnext[NPV];
        const bool more = ci + 1 < nchunks;
        if (more) load_params(e0 + MBH_EC, pnext);

        // ---- 2. expand GEMM over this wave's 16-pixel halo tiles -> Es
        {
            const T* wer0 = we + (size_t)(e0 + li) * a.KP + 8 * g;          // tile 0: expanded channel e0 + li
            const T* wer1 = wer0 + (size_t)16 * a.KP;                       // tile 1: e0 + 16 + li
            mbh_u4 wh0[2], wh1[2];
            if (hoist) {
                wh0[0] = *reinterpret_cast<const mbh_u4*>(wer0);
                wh1[0] = *reinterpret_cast<const mbh_u4*>(wer1);
                wh0[1] = wh0[0]; wh1[1] = wh1[0];
                if (nks > 1) {
                    wh0[1] = *reinterpret_cast<const mbh_u4*>(wer0 + 32);
                    wh1[1] = *reinterpret_cast<const mbh_u4*>(wer1 + 32);
                }
            }
            const mbh_f4 sc0 = *reinterpret_cast<const mbh_f4*>(Pc + (KK + 2) * MBH_EC + 4 * g);
            const mbh_f4 sc1 = *reinterpret_cast<const mbh_f4*>(Pc + (KK + 2) * MBH_EC + 16 + 4 * g);
            const mbh_f4 sh0 = *reinterpret_cast<const mbh_f4*>(Pc + (KK + 3) * MBH_EC + 4 * g);
            const mbh_f4 sh1 = *reinterpret_cast<const mbh_f4*>(Pc + (KK + 3) * MBH_EC + 16 + 4 * g);
            for (int mt = wave; mt < nmt_h; mt += 4) {
                const int p = mt * 16 + li;
                const int pc = p < a.PH ? p : a.PH - 1;
                const T* xr = Xs + (size_t)pc * ldx + 8 * g;
                mbh_f4 e0acc = (mbh_f4){0.f, 0.f, 0.f, 0.f}, e1acc = (mbh_f4){0.f, 0.f, 0.f, 0.f};
                if (hoist) {
                    const mbh_u4 x0 = *reinterpret_cast<const mbh_u4*>(xr);
                    e0acc = mbh_mfma<T>(wh0[0], x0, e0acc);
                    e1acc = mbh_mfma<T>(wh1[0], x0, e1acc);
                    if (nks > 1) {
                        const mbh_u4 x1 = *reinterpret_cast<const mbh_u4*>(xr + 32);
                        e0acc = mbh_mfma<T>(wh0[1], x1, e0acc);
                        e1acc = mbh_mfma<T>(wh1[1], x1, e1acc);
                    }
                } else {
                    for (int ks = 0; ks < nks; ++ks) {
                        const mbh_u4 xf = *reinterpret_cast<const mbh_u4*>(xr + ks * 32);
                        const mbh_u4 w0 = *reinterpret_cast<const mbh_u4*>(wer0 + ks * 32);
                        const mbh_u4 w1 = *reinterpret_cast<const mbh_u4*>(wer1 + ks * 32);
                        e0acc = mbh_mfma<T>(w0, xf, e0acc);
                        e1acc = mbh_mfma<T>(w1, xf, e1acc);
                    }
                }
                if (p < a.PH) {
                    const int hy = p / a.iw, hx = p - hy * a.iw;
                    const int iy = iy0 + hy, ix = ix0 + hx;
                    const bool inside = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
                    mbh_f4 v0, v1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v0[q] = inside ? mbh_act<RELU6>(__builtin_fmaf(e0acc[q], sc0[q], sh0[q]), a.act) : 0.f;
                        v1[q] = inside ? mbh_act<RELU6>(__builtin_fmaf(e1acc[q], sc1[q], sh1[q]), a.act) : 0.f;
                    }
                    *reinterpret_cast<mbh_f4*>(Es + (size_t)p * MBH_LDE + 4 * g) = v0;
                    *reinterpret_cast<mbh_f4*>(Es + (size_t)p * MBH_LDE + 16 + 4 * g) = v1;
                }
            }
        }
        __syncthreads();   // (A) the expanded chunk is complete

        // ---- 3. depthwise KxK straight into the projection's B-operand fragment (8 channels per lane) + MFMA
        {
            const float* pw = Pc + 8 * g;               // this lane's 8 channels of every parameter row
            const mbh_f4 sd0 = *reinterpret_cast<const mbh_f4*>(pw + KK * MBH_EC), sd1 = *reinterpret_cast<const mbh_f4*>(pw + KK * MBH_EC + 4);
            const mbh_f4 hd0 = *reinterpret_cast<const mbh_f4*>(pw + (KK + 1) * MBH_EC), hd1 = *reinterpret_cast<const mbh_f4*>(pw + (KK + 1) * MBH_EC + 4);
#pragma unroll
            for (int m = 0; m < MTO; ++m) {
                const int mt = wave + 4 * m;
                if (mt < nmt_o) {
                    const int o = mt * 16 + li;
                    const int oc = o < a.OPX ? o : a.OPX - 1;
                    const int oy = oc / a.tw, ox = oc - oy * a.tw;
                    const float* base = Es + (size_t)((oy * S) * a.iw + ox * S) * MBH_LDE + 8 * g;
                    mbh_f4 d0 = (mbh_f4){0.f, 0.f, 0.f, 0.f}, d1 = (mbh_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ky = 0; ky < K; ++ky)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) {
                            const float* ep = base + (size_t)(ky * a.iw + kx) * MBH_LDE;
                            const mbh_f4 v0 = *reinterpret_cast<const mbh_f4*>(ep), v1 = *reinterpret_cast<const mbh_f4*>(ep + 4);
                            const mbh_f4 w0 = *reinterpret_cast<const mbh_f4*>(pw + (ky * K + kx) * MBH_EC);
                            const mbh_f4 w1 = *reinterpret_cast<const mbh_f4*>(pw + (ky * K + kx) * MBH_EC + 4);
                            d0 = __builtin_elementwise_fma(v0, w0, d0);
                            d1 = __builtin_elementwise_fma(v1, w1, d1);
                        }
                    d0 = __builtin_elementwise_fma(d0, sd0, hd0);
                    d1 = __builtin_elementwise_fma(d1, sd1, hd1);
                    mbh_f8 dv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        dv[q] = mbh_act<RELU6>(d0[q], a.act);
                        dv[4 + q] = mbh_act<RELU6>(d1[q], a.act);
                    }
                    const mbh_u4 df = __builtin_bit_cast(mbh_u4, __builtin_convertvector(dv, mbh_v8<T>));
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc_o[c][m] = mbh_mfma<T>(wpf[c], df, acc_o[c][m]);
                }
            }
        }
        if (more) store_params(Ps + ((ci + 1) & 1) * PSZ, pnext);
        __syncthreads();   // (B) Es may be rewritten; the next chunk's parameters are visible
    }

    // ---- 4. epilogue: project BN (+ the block input at the centre tap, from Xs) -> 16-byte stores
    T* outp = reinterpret_cast<T*>(a.out) + (size_t)b * a.Ho * a.Wo * a.ld_out;
#pragma unroll
    for (int m = 0; m < MTO; ++m) {
        const int mt = wave + 4 * m;
        if (mt >= nmt_o) continue;
        const int o = mt * 16 + li;
        if (o >= a.OPX) continue;
        const int oy = o / a.tw, ox = o - oy * a.tw;
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (gy >= a.Ho || gx >= a.Wo) continue;
        T* op = outp + ((size_t)gy * a.Wo + gx) * a.ld_out;
        const T* rp = Xs + (size_t)((oy * S + a.pad_t) * a.iw + ox * S + a.pad_l) * ldx;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            const int n = c * 32 + 8 * g;
            if (n >= a.Cout) continue;
            mbh_f8 v;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int nn = n + q < a.Cout ? n + q : a.Cout - 1;
                const float acc = q < 4 ? acc_o[2 * c][m][q] : acc_o[2 * c + 1][m][q - 4];
                v[q] = __builtin_fmaf(acc, a.sp[nn], a.hp[nn]);
            }
            if (a.has_res) {
                const mbh_f8 rx = __builtin_convertvector(*reinterpret_cast<const mbh_v8<T>*>(rp + n), mbh_f8);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += rx[q];
            }
            *reinterpret_cast<mbh_v8<T>*>(op + n) = __builtin_convertvector(v, mbh_v8<T>);
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
static size_t mbh_lds_bytes(int ph, int kp, int k) {
    return (((size_t)ph * (kp + 8) * 2 + 15) & ~(size_t)15) + (size_t)ph * MBH_LDE * 4 + (size_t)2 * (k * k + 4) * MBH_EC * 4;
}

template <class T, int K, int S, int CP, int MTO>
static int launch_mbh(const MbhArgs& a, int batch, hipStream_t s) {
    const size_t lds = mbh_lds_bytes(a.PH, a.KP, K);
    YR_REQUIRE(lds <= 160 * 1024, "mbh: LDS tile of %zu bytes does not fit", lds);
    static bool attr_dev[64] = {false};
    int dev = 0;
    YR_CHECK_HIP(hipGetDevice(&dev));
    if (!attr_dev[dev & 63]) {
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)mbh_kernel<T, K, S, CP, MTO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)mbh_kernel<T, K, S, CP, MTO, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_dev[dev & 63] = true;
    }
    static char nm[56];
    static const int nm_len = snprintf(nm, sizeof(nm), "mbh_kernel<%s,%d,%d,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, S, CP, MTO);
    (void)nm_len;
    yr_note_kernel(nm);
    const dim3 grid((unsigned)(batch * a.tiles_x * a.tiles_y));
    if (a.act == YR_ACT_RELU6) hipLaunchKernelGGL((mbh_kernel<T, K, S, CP, MTO, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((mbh_kernel<T, K, S, CP, MTO, false>), grid, dim3(256), lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T, int K, int S>
static int launch_mbh_shape(const MbhArgs& a, int cp, int mto, int batch, hipStream_t s) {
    switch (cp * 10 + mto) {
        case 11: return launch_mbh<T, K, S, 1, 1>(a, batch, s);
        case 13: return launch_mbh<T, K, S, 1, 3>(a, batch, s);
        case 21: return launch_mbh<T, K, S, 2, 1>(a, batch, s);
        case 23: return launch_mbh<T, K, S, 2, 3>(a, batch, s);
        case 41: return launch_mbh<T, K, S, 4, 1>(a, batch, s);
        case 43: return launch_mbh<T, K, S, 4, 3>(a, batch, s);
        default: yr_set_error("mbh: no kernel for %d cout pairs x %d pixel tiles per wave", cp, mto); return YR_ERR_ARG;
    }
}

// Output tile th x tw for a map of ho x wo: few wasted pixels in ragged edge tiles, a small halo-to-output ratio (the
// expand GEMM and its epilogue run on halo pixels), an LDS footprint that leaves >= 2 workgroups per CU, and enough
// workgroups to fill 256 CUs.  op.k may force a choice: k = K | th << 8 | tw << 16.
static void mbh_pick_tile(int ho, int wo, int batch, int k, int s, int kp, int* th_out, int* tw_out) {
    double best = 1e30;
    *th_out = 8; *tw_out = 8;
    for (int th = 4; th <= 16; ++th)
        for (int tw = 4; tw <= 16; ++tw) {
            const int opx = th * tw;
            if (opx > 192 || (opx > 64 && opx < 100)) continue;     // MTO = 1 (<= 64 outputs) or 3 (<= 192)
            const int ih = (th - 1) * s + k, iw = (tw - 1) * s + k, ph = ih * iw;
            const size_t lds = mbh_lds_bytes(ph, kp, k);
            if (lds > 76 * 1024) continue;                            // two workgroups per CU
            const int ty = (ho + th - 1) / th, tx = (wo + tw - 1) / tw;
            const double blocks = (double)batch * ty * tx;
            const int nmt_h = (ph + 15) / 16, nmt_o = (opx + 15) / 16;
            // work per block in MFMA-tile units: expand on the halo (rounded to whole tiles over 4 waves) + dw/project
            const double per_block = ((nmt_h + 3) / 4) * 1.0 + ((nmt_o + 3) / 4) * 1.6;
            const int per_cu = lds <= 50 * 1024 ? 3 : 2;
            const double rounds = blocks / (256.0 * per_cu);
            const double cost = per_block * (rounds < 1.0 ? 1.0 : rounds) * (per_cu == 3 ? 1.0 : 1.15);
            if (cost < best) { best = cost; *th_out = th; *tw_out = tw; }
        }
}

// op fields (YR_OP_MBH): src[0] = block input (16-bit, ld % 8 == 0); cin; se_reduced = expanded width Cexp; cout <= 128;
// k = K (3 | 5), optionally | th << 8 | tw << 16 to force the output tile; stride 1 | 2; act = expand / depthwise
// activation; res (optional) = the block input itself (stride 1, cin == cout).  With CexpP = round_up(Cexp, 32),
// KP = round_up(cin, 32), everything zero padded:
//   wgt  = expand Wt[CexpP][KP] (16-bit, in the blob: CexpP*KP/2 floats);
//   wgt2 = [K*K + 4][CexpP] float32: depthwise taps | depthwise BN scale | shift | expand BN scale | shift;
//   b1   = project Wt[cout][CexpP] (16-bit); b2 = project BN scale [round_up(cout,8)] ++ shift [round_up(cout,8)].
template <class T>
static int launch_mbh_t(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY, "mbh: needs one identity source");
    const yr_src& in = op.src[0];
    const int K = op.k & 0xff, fth = (op.k >> 8) & 0xff, ftw = (op.k >> 16) & 0xff;
    YR_REQUIRE((K == 3 || K == 5) && (op.stride == 1 || op.stride == 2), "mbh: depthwise %dx%d stride %d is not fused", K, K, op.stride);
    YR_REQUIRE(in.ptr && op.out && op.wgt && op.wgt2 && op.b1 && op.b2, "mbh: null pointer");
    YR_REQUIRE(in.dtype == op.dtype && op.out_dtype == op.dtype, "mbh: input and output have the op's 16-bit dtype");
    YR_REQUIRE(in.ld % 8 == 0 && op.out_ld % 8 == 0 && in.c == op.cin && in.ld >= yr_round_up(in.c, 8) && op.out_ld >= yr_round_up(op.cout, 8),
               "mbh: channel strides must be multiples of 8 and cover round_up(c,8)");
    YR_REQUIRE(((uintptr_t)in.ptr | (uintptr_t)op.out | (uintptr_t)op.wgt | (uintptr_t)op.b1 | (uintptr_t)op.wgt2) % 16 == 0, "mbh: pointers must be 16-byte aligned");
    YR_REQUIRE(op.se_reduced >= 1 && op.cout >= 1 && op.cout <= 128 && op.cin >= 1 && op.cin <= 128, "mbh: widths out of range (cin, cout <= 128)");
    MbhArgs a;
    a.x = in.ptr; a.out = op.out;
    a.Cin = in.c; a.Cout = op.cout;
    a.CexpP = yr_round_up(op.se_reduced, 32); a.KP = yr_round_up(in.c, 32);
    a.we = op.wgt; a.prm = op.wgt2;
    a.wp = op.b1; a.sp = op.b2; a.hp = op.b2 + yr_round_up(op.cout, 8);
    a.Hi = in.h; a.Wi = in.w; a.Ho = (in.h + op.stride - 1) / op.stride; a.Wo = (in.w + op.stride - 1) / op.stride;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "mbh: output dims mismatch");
    a.ld_in = in.ld; a.ld_out = op.out_ld;
    const int pth = (a.Ho - 1) * op.stride + K - in.h, ptw = (a.Wo - 1) * op.stride + K - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.has_res = op.res != nullptr;
    if (a.has_res) YR_REQUIRE(op.res == in.ptr && op.stride == 1 && in.c == op.cout, "mbh: the residual must be the block input (stride 1, cin == cout)");
    a.act = op.act;
    if (fth && ftw) { a.th = fth; a.tw = ftw; }
    else mbh_pick_tile(a.Ho, a.Wo, batch, K, op.stride, a.KP, &a.th, &a.tw);
    a.OPX = a.th * a.tw;
    YR_REQUIRE(a.OPX >= 1 && a.OPX <= 192, "mbh: output tile %dx%d out of range (<= 192 pixels)", a.th, a.tw);
    a.ih = (a.th - 1) * op.stride + K; a.iw = (a.tw - 1) * op.stride + K; a.PH = a.ih * a.iw;
    a.tiles_x = (a.Wo + a.tw - 1) / a.tw; a.tiles_y = (a.Ho + a.th - 1) / a.th;
    YR_REQUIRE((long long)batch * a.tiles_x * a.tiles_y < (1ll << 31), "mbh: grid too large");
    const int cp = op.cout <= 32 ? 1 : (op.cout <= 64 ? 2 : 4);
    const int mto = a.OPX <= 64 ? 1 : 3;
    if (K == 3 && op.stride == 1) return launch_mbh_shape<T, 3, 1>(a, cp, mto, batch, s);
    if (K == 3 && op.stride == 2) return launch_mbh_shape<T, 3, 2>(a, cp, mto, batch, s);
    if (K == 5 && op.stride == 1) return launch_mbh_shape<T, 5, 1>(a, cp, mto, batch, s);
    return launch_mbh_shape<T, 5, 2>(a, cp, mto, batch, s);
}

int yr_launch_mbh(const yr_op& op, int batch, hipStream_t s) {
    if (op.dtype == YR_BF16) return launch_mbh_t<yr_bf16>(op, batch, s);
    if (op.dtype == YR_F16) return launch_mbh_t<yr_f16>(op, batch, s);
    yr_set_error("mbh: the fused MFMA block kernel works on 16-bit activations (dtype %d given)", op.dtype);
    return YR_ERR_ARG;
}
