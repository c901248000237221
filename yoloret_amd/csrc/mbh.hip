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
//   Ps [..]         the chunk's depthwise weights and the expand / depthwise BN parameters;
//   Ws [32][KP+8]   the chunk's expand weights, 16-bit.  Both are SINGLE buffers rewritten in the phase that does not read
//                   them (expand weights + expand BN behind barrier (A), depthwise rows behind barrier (B)): 12 KB less than
//                   double buffers for a 112-channel 5x5 block - the difference between one and two workgroups per CU.
//                   Both are fetched one chunk ahead into
//                   registers and stored behind the depthwise phase, so no chunk starts by waiting for L2 (the first
//                   version read the weight fragments from global memory where they were used: 4-5 us per chunk on
//                   the 13x13 blocks, whose four k-steps each waited for a round trip).
// Per chunk of 32 expanded channels (two barriers):
//   expand : Es[p][32] = act(BN(Xs[p][:] . We))         MFMA, 16-pixel tiles of the halo round-robin over the waves
//   dw+proj: every lane computes the KxK depthwise result of (its output pixel, 8 channels) in float32 - exactly the
//            MFMA B-operand fragment of the projection - rounds it to the 16-bit type and feeds
//            acc[cout][pixel] += Wp[cout][32] . D[32][pixel]; the projection accumulators stay in registers.
// Epilogue: project BN (+ the block input from Xs) -> one 16-byte store of 8 consecutive couts per lane (the cout-pair
// permutation of pointwise_h.hip).
#include "yr_common.h"

#include <cstdlib>

bool yr_mbn_takes(const yr_op& op);                               // mbn_h.hip
int yr_launch_mbn(const yr_op& op, int batch, hipStream_t s);
bool yr_mbxr_takes(const yr_op& op);                              // mbxr_h.hip: YR_OP_MBX row-walking, register-chained
int yr_launch_mbxr(const yr_op& op, int batch, int segs, hipStream_t s);
bool yr_mbhr_built(const yr_op& op);                              // mbxr_h.hip: YR_OP_MBH (the whole block) in the same form
int yr_launch_mbhr(const yr_op& op, int batch, int segs, hipStream_t s);
bool yr_mbh_prefers_chained(const yr_op& op);

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
    // MODE 1 (expand + depthwise only): per-tile channel sums of the stored outputs, float32 [B][rows_cap][ld_part]
    float* part; int ld_part, rows_cap;
};

template <class T>
__device__ __forceinline__ mbh_f4 mbh_mfma(mbh_u4 w, mbh_u4 x, mbh_f4 acc) {
    if constexpr (yr_elem<T>::dtype == YR_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mbh_v8<__bf16>, w), __builtin_bit_cast(mbh_v8<__bf16>, x), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mbh_v8<_Float16>, w), __builtin_bit_cast(mbh_v8<_Float16>, x), acc, 0, 0, 0);
}

// ACT 0: relu6; 1: swish; 2: whatever op.act says.  The results are rounded to a 16-bit type right away (8 or 11
// significant bits), so swish uses the hardware exp2 / reciprocal (about 1 ulp of float32 each) instead of the pinned
// float32 expf and the IEEE division of the float32 plans (28 instructions per element, and the EfficientNet blocks
// apply it to 6x the block's input twice).
template <int ACT>
__device__ __forceinline__ float mbh_act(float v, int act) {
    if constexpr (ACT == 0) return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f);   // clamp in one instruction (== min(max(v,0),6) for finite v)
    else if constexpr (ACT == 1) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
    else return yr_apply_act(v, act);
}

// CP: cout tile pairs (Cout <= 32*CP); NG: groups of 32 output pixels per wave (th*tw <= 128*NG)
//
// Output pixels are enumerated in RUNS of 4 along x (tw % 4 == 0); 8 runs = one group = 32 pixels = two 16-pixel MFMA
// tiles; groups go round-robin over the 4 waves.  In the depthwise phase a lane owns (run rl = lane>>3, channel quad
// cq = lane&7): 4 horizontally adjacent outputs x 4 channels, register blocked like depthwise.hip - an input row of
// 3S+K positions is read once for the 4 outputs, the K*K tap weights of the lane's channels stay in registers for the
// whole chunk.  (The first version gave every lane one pixel x 8 channels, the projection's MFMA operand layout, and
// re-read all K*K taps and their weights from LDS per output: PMC showed the LDS pipe 69 % busy, 30 % of it bank
// conflicts, and half of the reads were weights.)  The 16-bit results go to a wave-private LDS patch Ds[32][32] and
// come back as B-operand fragments - same wave, program order, no workgroup barrier.
//
// MODE 0: the whole block.  MODE 1 (YR_OP_MBX; MBConv blocks WITH squeeze-excite, efficientnet.py:406-536, whose
// projection needs the gate computed from the complete depthwise map): expand + depthwise only - the depthwise results
// are stored (16-bit) and every workgroup also writes the per-channel sums of what it stored to its row of `part`
// (the squeeze, tf.reduce_mean over H, W at efficientnet.py:417, finished by SE_FC); the expanded INPUT of the depthwise
// conv - half of the chain's traffic - still never reaches HBM.
template <class T, int K, int S, int CP, int NG, int ACT, int MODE>
__global__ __launch_bounds__(256, (MODE == 1 || CP * NG <= 1 ? 3 : 2)) void mbh_kernel(MbhArgs a) {
    constexpr int CT = 2 * CP, KK = K * K, MTO = 2 * NG;
    constexpr int PSZ = (KK + 4) * MBH_EC;              // floats per parameter buffer: dw taps | sd | hd | se | he
    constexpr int COLS = 3 * S + K;                     // input positions per row feeding a run of 4 outputs
    constexpr int LDD = MBH_EC + 8;                     // Ds row stride in elements (80 bytes: conflict-free b128 rows)
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int ldx = a.KP + 8;                           // Xs / Ws row stride in elements (16-byte aligned, conflict-free)
    T* Xs = reinterpret_cast<T*>(lds_raw);
    float* Es = reinterpret_cast<float*>(lds_raw + (((size_t)a.PH * ldx * sizeof(T) + 15) & ~(size_t)15));
    constexpr int PDZ = (KK + 2) * MBH_EC;              // ... of which the depthwise phase reads the first PDZ, the expand phase the rest
    float* Ps = Es + (size_t)a.PH * MBH_LDE;
    T* Ws = reinterpret_cast<T*>(Ps + PSZ);             // [32][ldx]: expand weights of the current chunk
    T* Ds = Ws + (size_t)32 * ldx;                      // [4 waves][32][LDD]: depthwise results on their way to the MFMA
    float* Sw = reinterpret_cast<float*>(Ds);           // MODE 1 instead: [4 waves][32] channel sums of the current chunk

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
    // block-uniform: the halo tile lies inside the image (no zero padding to apply anywhere)
    const bool all_inside = iy0 >= 0 && ix0 >= 0 && iy0 + a.ih <= a.Hi && ix0 + a.iw <= a.Wi;

    if constexpr (MODE == 1) {
        // rows of `part` no tile owns (the buffer is sized for the smallest tile the host may choose): zero, shared out
        // over the image's workgroups
        if (a.part != nullptr && a.rows_cap > tpi) {
            const int extra = a.rows_cap - tpi, per = (extra + tpi - 1) / tpi;
            const int r0 = tpi + r * per, r1 = r0 + per < a.rows_cap ? r0 + per : a.rows_cap;
            float* pz = a.part + (size_t)b * a.rows_cap * a.ld_part;
            for (int i = r0 * a.ld_part + tid; i < r1 * a.ld_part; i += 256) pz[i] = 0.f;
        }
    }
    // ---- 1. input halo tile -> LDS (zero outside the image and beyond Cin; pad channels of the source may hold anything).
    //      Loads are issued in batches before the first LDS store of the batch (one HBM round trip per batch).
    {
        const int nq = a.KP >> 3;                       // 16-byte vectors per pixel row
        const int cq = (a.Cin + 7) >> 3;                // ... of which hold real channels
        const int total = a.PH * nq;
        const bool ragged = (a.Cin & 7) != 0;
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
                if (idx < total && q < cq && (all_inside || (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi))) {
                    v[u] = *reinterpret_cast<const mbh_u4*>(xin + ((size_t)iy * a.Wi + ix) * a.ld_in + q * 8);
                    if (ragged) {
                        const int cv = a.Cin - q * 8;   // real channels in this vector
#pragma unroll
                        for (int d = 0; d < 4; ++d) v[u][d] = cv >= 2 * d + 2 ? v[u][d] : (cv == 2 * d + 1 ? (v[u][d] & 0xffffu) : 0u);
                    }
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
            if (i < PSZ) pv[u] = a.prm[(size_t)rr * a.CexpP + e0 + ch];
        }
    };
    // lo..hi: the depthwise rows [0, PDZ) or the expand rows [PDZ, PSZ) - they are rewritten at different points of a chunk
    auto store_params = [&](float* dst, const float (&pv)[NPV], int lo, int hi) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NPV; ++u) {
            const int i = tid + u * 256;
            if (i >= lo && i < hi) dst[i] = pv[u];
        }
    };
    // expand weights of one chunk: 32 rows x KP/8 16-byte vectors, at most 2 per thread (KP <= 128)
    const int wq = a.KP >> 3, wtotal = 32 * wq;
    auto load_w = [&](int e0, mbh_u4 (&wv)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256;
            const int rr = i / wq, q = i - rr * wq;
            wv[u] = (mbh_u4){0u, 0u, 0u, 0u};
            if (i < wtotal) wv[u] = *reinterpret_cast<const mbh_u4*>(we + (size_t)(e0 + rr) * a.KP + q * 8);
        }
    };
    auto store_w = [&](T* dst, const mbh_u4 (&wv)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256;
            const int rr = i / wq, q = i - rr * wq;
            if (i < wtotal) *reinterpret_cast<mbh_u4*>(dst + (size_t)rr * ldx + q * 8) = wv[u];
        }
    };
    float pcur[NPV];                                    // the current chunk's parameters (its depthwise rows are stored at the chunk's top)
    {
        mbh_u4 wv[2];
        load_params(0, pcur);
        load_w(0, wv);
        store_params(Ps, pcur, PDZ, PSZ);
        store_w(Ws, wv);
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
        wprow[c] = MODE == 0 ? wp + (size_t)(n < a.Cout ? n : 0) * a.CexpP + 8 * g : nullptr;
    }
    const int nmt_h = (a.PH + 15) >> 4;
    const int nks = a.KP >> 5;                          // k-steps of the expand GEMM
    const bool hoist = a.KP <= 64;                      // the chunk's expand-weight fragments fit registers: read once per wave
    // which of this lane's halo pixels (expand tile slot j: pixel (wave + 4j)*16 + li) lie inside the image
    unsigned inside_bits = 0;
    if (!all_inside) {
        for (int j = 0, mt = wave; mt < nmt_h; ++j, mt += 4) {
            const int p = mt * 16 + li;
            const int hy = p / a.iw, hx = p - hy * a.iw;
            const int iy = iy0 + hy, ix = ix0 + hx;
            inside_bits |= (p < a.PH && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi ? 1u : 0u) << j;
        }
    }
    // depthwise-phase geometry of this lane: run rl of each of the wave's groups, channel quad cq
    const int cq4 = (lane & 7) * 4, rl = lane >> 3;
    const int nrx = a.tw >> 2, nruns = a.th * nrx, ngroups = (nruns + 7) >> 3;
    int es_off[NG];                                     // float offset of the run's window origin in Es (+ channel quad)
    unsigned out_off[NG];                               // MODE 1: element offset of the run's first output pixel (+ channel quad) in the image
    unsigned out_ok[NG];                                // MODE 1: which of the run's 4 pixels exist
#pragma unroll
    for (int m = 0; m < NG; ++m) {
        const int run = (wave + 4 * m) * 8 + rl;
        const int rc = run < nruns ? run : nruns - 1;
        const int oy = rc / nrx, ox = (rc - oy * nrx) * 4;
        es_off[m] = ((oy * S) * a.iw + ox * S) * MBH_LDE + cq4;
        const int gy = oy0 + oy, gx = ox0 + ox;
        out_off[m] = (unsigned)((gy * a.Wo + gx) * a.ld_out + cq4);
        out_ok[m] = 0u;
        if (MODE == 1 && run < nruns && gy < a.Ho)
            for (int i = 0; i < 4; ++i) out_ok[m] |= (gx + i < a.Wo ? 1u : 0u) << i;
    }
    T* outp = reinterpret_cast<T*>(a.out) + (size_t)b * a.Ho * a.Wo * a.ld_out;
    T* Dw = Ds + (size_t)wave * 32 * LDD;
    __syncthreads();                                    // Xs, Ps[0], Ws[0] visible

    const int nchunks = a.CexpP >> 5;
    for (int ci = 0; ci < nchunks; ++ci) {
        const int e0 = ci * MBH_EC;
        const float* Pc = Ps;
        store_params(Ps, pcur, 0, PDZ);    // this chunk's depthwise rows: every wave is past barrier (B) of the previous chunk
        // prefetches: this chunk's projection fragments; the next chunk's parameters and expand weights (into registers; the
        // expand rows / weights are stored behind this chunk's depthwise phase, the depthwise rows at the next chunk's top)
        mbh_u4 wpf[CT];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c) wpf[c] = *reinterpret_cast<const mbh_u4*>(wprow[c] + e0);
        }
        float pnext[NPV];
        mbh_u4 wnext[2];
        const bool more = ci + 1 < nchunks;
        if (more) {
            load_params(e0 + MBH_EC, pnext);
            load_w(e0 + MBH_EC, wnext);
        }
        const T* Wc = Ws;

        // ---- 2. expand GEMM over this wave's 16-pixel halo tiles -> Es
        {
            const T* wer0 = Wc + (size_t)li * ldx + 8 * g;                  // tile 0: expanded channel e0 + li (LDS)
            const T* wer1 = wer0 + (size_t)16 * ldx;                        // tile 1: e0 + 16 + li
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
            unsigned bits = inside_bits;
            for (int mt = wave; mt < nmt_h; mt += 4, bits >>= 1) {
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
                mbh_f4 v0 = __builtin_elementwise_fma(e0acc, sc0, sh0), v1 = __builtin_elementwise_fma(e1acc, sc1, sh1);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v0[q] = mbh_act<ACT>(v0[q], a.act); v1[q] = mbh_act<ACT>(v1[q], a.act); }
                if (!all_inside && !(bits & 1u)) {      // TF pads the EXPANDED tensor with zeros
                    v0 = (mbh_f4){0.f, 0.f, 0.f, 0.f};
                    v1 = v0;
                }
                if (p < a.PH) {
                    *reinterpret_cast<mbh_f4*>(Es + (size_t)p * MBH_LDE + 4 * g) = v0;
                    *reinterpret_cast<mbh_f4*>(Es + (size_t)p * MBH_LDE + 16 + 4 * g) = v1;
                }
            }
        }
        __syncthreads();   // (A) the expanded chunk is complete

        // ---- 3. depthwise KxK, 4 outputs x 4 channels per lane -> Ds (16-bit) -> B-operand fragments -> projection MFMA
        {
            mbh_f4 wk[K == 3 ? KK : 1];
            if constexpr (K == 3) {
#pragma unroll
                for (int tp = 0; tp < KK; ++tp) wk[tp] = *reinterpret_cast<const mbh_f4*>(Pc + tp * MBH_EC + cq4);
            }
            const mbh_f4 sd = *reinterpret_cast<const mbh_f4*>(Pc + KK * MBH_EC + cq4);
            const mbh_f4 hd = *reinterpret_cast<const mbh_f4*>(Pc + (KK + 1) * MBH_EC + cq4);
            mbh_f4 psum = (mbh_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < NG; ++m) {
                if (wave + 4 * m < ngroups) {           // wave-uniform
                    const float* base = Es + es_off[m];
                    mbh_f4 acc[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = (mbh_f4){0.f, 0.f, 0.f, 0.f};
                    // (MODE 1, K = 5: a real loop over the rows.  Fully unrolled and without the projection's fences the
                    // scheduler finishes pixel 0's 25 taps first and keeps every operand and weight for pixels 1..3 - in scratch.)
#pragma unroll (MODE == 1 && K == 5 ? 1 : K)
                    for (int ky = 0; ky < K; ++ky) {
                        const float* rowp = base + (size_t)(ky * a.iw) * MBH_LDE;
                        mbh_f4 col[COLS];
#pragma unroll
                        for (int j = 0; j < COLS; ++j) col[j] = *reinterpret_cast<const mbh_f4*>(rowp + j * MBH_LDE);
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) {
                            mbh_f4 w;
                            if constexpr (K == 3) w = wk[ky * K + kx];
                            else w = *reinterpret_cast<const mbh_f4*>(Pc + (ky * K + kx) * MBH_EC + cq4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_fma(col[i * S + kx], w, acc[i]);
                        }
                    }
                    typedef T t4 __attribute__((ext_vector_type(4)));
                    if constexpr (MODE == 1) {
                        // the depthwise map itself is the output: 8-byte stores (8 lanes = 64 contiguous bytes of a pixel);
                        // the sums are those of the STORED (rounded) values, like depthwise.hip's SE form
                        const bool ch_ok = e0 + cq4 < a.ld_out;
                        // every value and the sums first, the conditional stores last: anything used only behind a branch is
                        // sunk there by the compiler - with the stores interleaved, the FMA chains of pixels 1..3 moved behind
                        // pixel 0's branch and all K * COLS operands and K * K weights stayed alive (in scratch) until then
                        t4 dr[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            mbh_f4 d = __builtin_elementwise_fma(acc[i], sd, hd);
#pragma unroll
                            for (int q = 0; q < 4; ++q) d[q] = mbh_act<ACT>(d[q], a.act);
                            dr[i] = __builtin_convertvector(d, t4);
                            const mbh_f4 back = __builtin_convertvector(dr[i], mbh_f4);
                            const float keep = ch_ok && ((out_ok[m] >> i) & 1u) ? 1.0f : 0.0f;
                            psum += back * keep;
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (ch_ok && ((out_ok[m] >> i) & 1u))
                                *reinterpret_cast<t4*>(outp + (out_off[m] + (unsigned)(i * a.ld_out + e0))) = dr[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            mbh_f4 d = __builtin_elementwise_fma(acc[i], sd, hd);
#pragma unroll
                            for (int q = 0; q < 4; ++q) d[q] = mbh_act<ACT>(d[q], a.act);
                            *reinterpret_cast<t4*>(Dw + (size_t)(rl * 4 + i) * LDD + cq4) = __builtin_convertvector(d, t4);
                        }
                        // same wave, program order: the patch is complete before its fragments are read (the fence keeps the
                        // compiler from moving the reads up; LDS executes a wave's accesses in order)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            const mbh_u4 df = *reinterpret_cast<const mbh_u4*>(Dw + (size_t)(tt * 16 + li) * LDD + 8 * g);
#pragma unroll
                            for (int c = 0; c < CT; ++c) acc_o[c][2 * m + tt] = mbh_mfma<T>(wpf[c], df, acc_o[c][2 * m + tt]);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();    // the fragments are read before the next group overwrites the patch
                    }
                }
            }
            if constexpr (MODE == 1) {
                if (a.part != nullptr) {
                    // lanes l, l+8, ... l+56 hold the same channel quad: add them in a fixed butterfly order, then the waves
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = psum[q];
                        v += __shfl_xor(v, 8);
                        v += __shfl_xor(v, 16);
                        v += __shfl_xor(v, 32);
                        psum[q] = v;
                    }
                    if (lane < 8) *reinterpret_cast<mbh_f4*>(Sw + wave * MBH_EC + cq4) = psum;
                }
            }
        }
        if (more) {
            store_params(Ps, pnext, PDZ, PSZ);   // the next chunk's expand BN rows and weights: every wave is past barrier (A)
            store_w(Ws, wnext);
#pragma unroll
            for (int u = 0; u < NPV; ++u) pcur[u] = pnext[u];
        }
        __syncthreads();   // (B) Es may be rewritten; the next chunk's parameters and expand weights are visible
        if constexpr (MODE == 1) {
            // this chunk's 32 channel sums of the tile: waves added in index order (Sw is next written behind barrier (A))
            if (a.part != nullptr && tid < MBH_EC && e0 + tid < a.ld_part)
                a.part[((size_t)b * a.rows_cap + r) * a.ld_part + e0 + tid] =
                    ((Sw[tid] + Sw[MBH_EC + tid]) + Sw[2 * MBH_EC + tid]) + Sw[3 * MBH_EC + tid];
        }
    }
    if constexpr (MODE == 1) return;

    // ---- 4. epilogue: project BN (+ the block input at the centre tap, from Xs) -> 16-byte stores.
    //      Tile 2m+tt, lane li: run (wave+4m)*8 + 4tt + (li>>2), pixel li&3 of the run.
#pragma unroll
    for (int mm = 0; mm < MTO; ++mm) {
        const int run = (wave + 4 * (mm >> 1)) * 8 + 4 * (mm & 1) + (li >> 2);
        if (run >= nruns) continue;
        const int oy = run / nrx, ox = (run - oy * nrx) * 4 + (li & 3);
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
                const float acc = q < 4 ? acc_o[2 * c][mm][q] : acc_o[2 * c + 1][mm][q - 4];
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
static size_t mbh_lds_bytes(int ph, int kp, int k, int mode) {
    return (((size_t)ph * (kp + 8) * 2 + 15) & ~(size_t)15) + (size_t)ph * MBH_LDE * 4 + (size_t)(k * k + 4) * MBH_EC * 4 +
           (size_t)32 * (kp + 8) * 2 + (mode == 0 ? (size_t)4 * 32 * (MBH_EC + 8) * 2 : (size_t)4 * MBH_EC * 4);
}

template <class T, int K, int S, int CP, int NG, int MODE>
static int launch_mbh(const MbhArgs& a, int batch, hipStream_t s) {
    const size_t lds = mbh_lds_bytes(a.PH, a.KP, K, MODE);
    YR_REQUIRE(lds <= 160 * 1024, "mbh: LDS tile of %zu bytes does not fit", lds);
    static bool attr_dev[64] = {false};
    int dev = 0;
    YR_CHECK_HIP(hipGetDevice(&dev));
    if (!attr_dev[dev & 63]) {
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)mbh_kernel<T, K, S, CP, NG, 0, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)mbh_kernel<T, K, S, CP, NG, 1, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_dev[dev & 63] = true;
    }
    static char nm[56];
    static const int nm_len = MODE == 0 ? snprintf(nm, sizeof(nm), "mbh_kernel<%s,%d,%d,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, S, CP, NG)
                                        : snprintf(nm, sizeof(nm), "mbh_kernel<%s,%d,%d,x,%d>", yr_dtype_name(yr_elem<T>::dtype), K, S, NG);
    (void)nm_len;
    yr_note_kernel(nm);
    const dim3 grid((unsigned)(batch * a.tiles_x * a.tiles_y));
    if (a.act == YR_ACT_RELU6) hipLaunchKernelGGL((mbh_kernel<T, K, S, CP, NG, 0, MODE>), grid, dim3(256), lds, s, a);
    else if (a.act == YR_ACT_SWISH) hipLaunchKernelGGL((mbh_kernel<T, K, S, CP, NG, 1, MODE>), grid, dim3(256), lds, s, a);
    else {   // (the graph compiler fuses ReLU6 and swish blocks only - compiler.MBH_ACTS; the run-time-switch form spilled up to 924 bytes per lane)
        yr_set_error("mbh: activation %d is not fused (ReLU6 and swish are)", a.act);
        return YR_ERR_ARG;
    }
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T, int K, int S>
static int launch_mbh_shape(const MbhArgs& a, int cp, int ng, int batch, hipStream_t s) {
    switch (cp * 10 + ng) {
        case 1: return launch_mbh<T, K, S, 1, 1, 1>(a, batch, s);      // cp == 0: expand + depthwise only (MODE 1)
        case 2: return launch_mbh<T, K, S, 1, 2, 1>(a, batch, s);
        case 11: return launch_mbh<T, K, S, 1, 1, 0>(a, batch, s);
        case 12: return launch_mbh<T, K, S, 1, 2, 0>(a, batch, s);
        case 21: return launch_mbh<T, K, S, 2, 1, 0>(a, batch, s);
        case 22: return launch_mbh<T, K, S, 2, 2, 0>(a, batch, s);
        case 41: return launch_mbh<T, K, S, 4, 1, 0>(a, batch, s);
        default: yr_set_error("mbh: no kernel for %d cout pairs x %d pixel groups per wave", cp, ng); return YR_ERR_ARG;
    }
}

// Output tile th x tw for a map of ho x wo (tw a multiple of 4: outputs are processed in runs of 4 along x; at most
// 128 outputs per group slot, i.e. 128 for NG = 1, 256 for NG = 2): few wasted pixels in ragged edge tiles, a small
// halo-to-output ratio (the expand GEMM and its epilogue run on halo pixels), whole groups of 8 runs for the four waves,
// an LDS footprint that leaves several workgroups per CU, and enough workgroups to fill 256 CUs.  The constants were
// fitted to tools/mbh_probe.py on the MobileNetV2 block shapes.  op.k may force a choice: k = K | th << 8 | tw << 16.
// cp == 0: the expand + depthwise form; max_tiles > 0 bounds the tiles per image (rows of its partial-sum buffer).
static void mbh_pick_tile(int ho, int wo, int batch, int k, int s, int kp, int cp, int max_tiles, int* th_out, int* tw_out) {
    double best = 1e30;
    *th_out = 8; *tw_out = 8;
    const int max_groups = cp >= 4 ? 4 : 8;                       // NG <= 1 for 4 cout pairs (accumulator registers)
    // at least two workgroups per CU if any tile allows it (wide 5x5 stride-2 blocks do not: one workgroup per CU then)
    for (size_t lds_cap = 78 * 1024; best == 1e30 && lds_cap <= 160 * 1024; lds_cap += 82 * 1024)
    for (int th = 2; th <= 16; ++th)
        for (int tw = 4; tw <= 32; tw += 4) {
            const int runs = th * (tw / 4), groups = (runs + 7) / 8;
            if (groups > max_groups) continue;
            const int ih = (th - 1) * s + k, iw = (tw - 1) * s + k, ph = ih * iw;
            const size_t lds = mbh_lds_bytes(ph, kp, k, cp == 0);
            if (lds > lds_cap) continue;
            const int ty = (ho + th - 1) / th, tx = (wo + tw - 1) / tw;
            if (max_tiles > 0 && ty * tx > max_tiles) continue;
            const double blocks = (double)batch * ty * tx;
            const int nmt_h = (ph + 15) / 16;
            // work per block: expand on the halo (whole MFMA tiles over 4 waves) + depthwise/project (whole groups over 4 waves)
            const double per_block = ((nmt_h + 3) / 4) * 1.0 + ((groups + 3) / 4) * (cp == 0 ? 2.0 : 3.0) * (k == 5 ? 2.0 : 1.0) + 1.5;
            const int per_cu = lds <= 31 * 1024 ? 5 : lds <= 39 * 1024 ? 4 : lds <= 52 * 1024 ? 3 : 2;
            const double rounds = blocks / (256.0 * per_cu);
            const double occ = per_cu >= 4 ? 1.0 : per_cu == 3 ? 1.08 : 1.25;
            const double cost = per_block * (rounds < 1.0 ? 1.0 : rounds) * occ;
            if (cost < best) { best = cost; *th_out = th; *tw_out = tw; }
        }
}

// op fields (YR_OP_MBH): src[0] = block input (16-bit, ld % 8 == 0); cin; se_reduced = expanded width Cexp; cout <= 128;
// k = K (3 | 5), optionally | th << 8 | tw << 16 to force the output tile (tw % 4 == 0); stride 1 | 2; act = expand / depthwise
// activation; res (optional) = the block input itself (stride 1, cin == cout).  With CexpP = round_up(Cexp, 32),
// KP = round_up(cin, 32), everything zero padded:
//   wgt  = expand Wt[CexpP][KP] (16-bit, in the blob: CexpP*KP/2 floats);
//   wgt2 = [K*K + 4][CexpP] float32: depthwise taps | depthwise BN scale | shift | expand BN scale | shift;
//   b1   = project Wt[cout][CexpP] (16-bit); b2 = project BN scale [round_up(cout,8)] ++ shift [round_up(cout,8)].
// YR_OP_MBX (expand + depthwise, the first two thirds of the block): cout = Cexp = the width of the stored depthwise map;
// wgt, wgt2 as above; no b1 / b2 / res; gate (optional) = OUTPUT, float32 [B][se_reduced][gate_ld] per-tile channel sums
// (se_reduced = rows of the buffer >= tiles per image; the rows no tile owns are zeroed).
template <class T>
static int launch_mbh_t(const yr_op& op_in, int batch, hipStream_t s) {
    // forced tile th = 254: this file's (and mbn_h.hip's) own choice, never the register-chained forms - the tuner's way to
    // time both, and the A/B switch of the tests
    const bool legacy = ((op_in.k >> 8) & 0xff) == 254;
    yr_op op = op_in;
    if (legacy) op.k &= 0xff;
    const bool full = op.kind == YR_OP_MBH;
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY, "mbh: needs one identity source");
    const yr_src& in = op.src[0];
    const int K = op.k & 0xff, fth = (op.k >> 8) & 0xff, ftw = (op.k >> 16) & 0xff;
    YR_REQUIRE((K == 3 || K == 5) && (op.stride == 1 || op.stride == 2), "mbh: depthwise %dx%d stride %d is not fused", K, K, op.stride);
    YR_REQUIRE(in.ptr && op.out && op.wgt && op.wgt2 && (!full || (op.b1 && op.b2)), "mbh: null pointer");
    YR_REQUIRE(in.dtype == op.dtype && op.out_dtype == op.dtype, "mbh: input and output have the op's 16-bit dtype");
    YR_REQUIRE(in.ld % 8 == 0 && op.out_ld % 8 == 0 && in.c == op.cin && in.ld >= yr_round_up(in.c, 8) && op.out_ld >= yr_round_up(op.cout, 8),
               "mbh: channel strides must be multiples of 8 and cover round_up(c,8)");
    YR_REQUIRE(((uintptr_t)in.ptr | (uintptr_t)op.out | (uintptr_t)op.wgt | (uintptr_t)op.b1 | (uintptr_t)op.wgt2) % 16 == 0, "mbh: pointers must be 16-byte aligned");
    YR_REQUIRE(op.cout >= 1 && op.cin >= 1 && op.cin <= 128, "mbh: widths out of range (cin <= 128)");
    if (full) {
        // the whole block in the register-chained form (mbxr_h.hip: mbhr_kernel): forced tile th = 255 (tw = row segments);
        // without a forced tile it is the default where it is built (YOLORET_MBHR=0: the kernels below, for A/B runs)
        if (!legacy && yr_mbh_prefers_chained(op) && yr_mbhr_built(op)) {
            YR_REQUIRE(op.res == nullptr || (op.res == in.ptr && op.stride == 1 && in.c == op.cout), "mbh: the residual must be the block input (stride 1, cin == cout)");
            YR_REQUIRE((in.h + op.stride - 1) / op.stride == op.h && (in.w + op.stride - 1) / op.stride == op.w, "mbh: output dims mismatch");
            return yr_launch_mbhr(op, batch, fth == 255 ? ftw : 0, s);
        }
        YR_REQUIRE(fth != 255, "mbh: the register-chained whole-block form (tile 255) is not built for this op");
    }
    {   // the narrow stride-2 3x3 block at the network's front: its own kernel (mbn_h.hip; YOLORET_MBN=0: this one, for A/B runs)
        static const bool mbn_on = !(getenv("YOLORET_MBN") && atoi(getenv("YOLORET_MBN")) == 0);
        if (mbn_on && yr_mbn_takes(op)) return yr_launch_mbn(op, batch, s);
    }
    MbhArgs a;
    a.x = in.ptr; a.out = op.out;
    a.Cin = in.c; a.Cout = op.cout;
    a.KP = yr_round_up(in.c, 32);
    a.we = op.wgt; a.prm = op.wgt2;
    a.part = nullptr; a.ld_part = 0; a.rows_cap = 0;
    if (full) {
        YR_REQUIRE(op.se_reduced >= 1 && op.cout <= 128, "mbh: widths out of range (cout <= 128)");
        a.CexpP = yr_round_up(op.se_reduced, 32);
        a.wp = op.b1; a.sp = op.b2; a.hp = op.b2 + yr_round_up(op.cout, 8);
    } else {
        a.CexpP = yr_round_up(op.cout, 32);
        a.wp = nullptr; a.sp = nullptr; a.hp = nullptr;
        YR_REQUIRE(op.res == nullptr, "mbx: no residual in the expand + depthwise form");
        if (op.gate) {
            YR_REQUIRE(op.gate_ld % 4 == 0 && op.gate_ld >= op.cout && op.gate_ld <= a.CexpP && op.se_reduced >= 1 && ((uintptr_t)op.gate % 16) == 0,
                       "mbx: bad partial-sum buffer (ld %d for %d channels, %d rows)", op.gate_ld, op.cout, op.se_reduced);
            a.part = const_cast<float*>(op.gate); a.ld_part = op.gate_ld; a.rows_cap = op.se_reduced;
        }
    }
    a.Hi = in.h; a.Wi = in.w; a.Ho = (in.h + op.stride - 1) / op.stride; a.Wo = (in.w + op.stride - 1) / op.stride;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "mbh: output dims mismatch");
    a.ld_in = in.ld; a.ld_out = op.out_ld;
    const int pth = (a.Ho - 1) * op.stride + K - in.h, ptw = (a.Wo - 1) * op.stride + K - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.has_res = op.res != nullptr;
    if (a.has_res) YR_REQUIRE(op.res == in.ptr && op.stride == 1 && in.c == op.cout, "mbh: the residual must be the block input (stride 1, cin == cout)");
    a.act = op.act;
    if (!full) {
        // the expand + depthwise form has a second kernel (mbxr_h.hip): forced tile th = 255 selects it (tw = row segments),
        // no forced tile: it is the default where it is built (YOLORET_MBXR=0: the LDS-tiled form below, for A/B runs)
        if (!legacy && yr_mbh_prefers_chained(op) && yr_mbxr_takes(op)) return yr_launch_mbxr(op, batch, fth == 255 ? ftw : 0, s);
        YR_REQUIRE(fth != 255, "mbx: the register-chained form (tile 255) is not built for this op");
    }
    const int cp = !full ? 0 : op.cout <= 32 ? 1 : (op.cout <= 64 ? 2 : 4);
    if (fth && ftw) { a.th = fth; a.tw = ftw; }
    else mbh_pick_tile(a.Ho, a.Wo, batch, K, op.stride, a.KP, cp, a.rows_cap, &a.th, &a.tw);
    a.OPX = a.th * a.tw;
    const int groups = (a.th * (a.tw / 4) + 7) / 8;
    YR_REQUIRE(a.tw % 4 == 0 && a.th >= 1 && groups >= 1 && groups <= (cp >= 4 ? 4 : 8),
               "mbh: output tile %dx%d unsupported (tw %% 4 == 0, at most %d pixels)", a.th, a.tw, cp >= 4 ? 128 : 256);
    a.ih = (a.th - 1) * op.stride + K; a.iw = (a.tw - 1) * op.stride + K; a.PH = a.ih * a.iw;
    a.tiles_x = (a.Wo + a.tw - 1) / a.tw; a.tiles_y = (a.Ho + a.th - 1) / a.th;
    YR_REQUIRE((long long)batch * a.tiles_x * a.tiles_y < (1ll << 31) && (long long)a.Ho * a.Wo * a.ld_out < (1ll << 31), "mbh: grid or image too large");
    YR_REQUIRE(a.part == nullptr || a.tiles_x * a.tiles_y <= a.rows_cap, "mbx: %d tiles per image exceed the %d rows of the partial-sum buffer",
               a.tiles_x * a.tiles_y, a.rows_cap);
    const int ng = groups <= 4 ? 1 : 2;
    if (K == 3 && op.stride == 1) return launch_mbh_shape<T, 3, 1>(a, cp, ng, batch, s);
    if (K == 3 && op.stride == 2) return launch_mbh_shape<T, 3, 2>(a, cp, ng, batch, s);
    if (K == 5 && op.stride == 1) return launch_mbh_shape<T, 5, 1>(a, cp, ng, batch, s);
    return launch_mbh_shape<T, 5, 2>(a, cp, ng, batch, s);
}

int yr_launch_mbh(const yr_op& op, int batch, hipStream_t s) {
    if (op.dtype == YR_BF16) return launch_mbh_t<yr_bf16>(op, batch, s);
    if (op.dtype == YR_F16) return launch_mbh_t<yr_f16>(op, batch, s);
    yr_set_error("mbh: the fused MFMA block kernel works on 16-bit activations (dtype %d given)", op.dtype);
    return YR_ERR_ARG;
}
