// Pointwise (1x1) convolution on 16-bit activations and weights (bfloat16 / float16 storage, float32 accumulate):
// the reduced-precision form of pointwise.hip for BASELINE.json configs 3 and 5 ("bf16 pointwise on MFMA",
// "fp16 + fused RFCR upsample-concat-conv").  Same op, same fusions - UpSampling2D / MaxPooling2D / Concatenate /
// SE Multiply folded into the loads, BatchNorm scale/shift + activation + residual (+ 2x2 max) in the epilogue;
// reference code/yolo3/model.py:25-30,98-114,152-155,243-251,298-318, code/yolo3/efficientnet.py:485-496,517-533.
//
// GEMM view: D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k] on v_mfma_f32_16x16x32_bf16 / _f16:
//   A operand = weights  (lane l: cout row i = l&15, the 8 consecutive k of group g = l>>4: one 16-byte load)
//   B operand = pixels   (lane l: pixel j = l&15, the same 8 k)
//   D: lane holds pixel j = l&15, MFMA rows 4g..4g+3 (float32).
// The k space is the concatenation of the sources, each padded to a multiple of 8 channels (16 bytes), walked in
// chunks of 32.  Cout tiles come in PAIRS and the weight rows are assigned to MFMA rows so that row 4g+r of the pair's
// first tile is cout 8g+r and of its second tile cout 8g+4+r: lane group g then owns the 8 CONSECUTIVE couts
// 8g..8g+7 of a pixel - one 16-byte store of bf16/f16 (or two float4 stores for the float32 logit outputs).
// No LDS, no barriers (the direct form of pointwise.hip): every wave loads both operands straight into operand
// layout and keeps D chunks of loads in flight; the steady-state loop is branch-free so the waits are counted.
#include <stdlib.h>

#include "pwh_common.h"


// PT: 16-pixel tiles per wave, CP: 32-cout tile PAIRS per wave; 4 waves along the pixels: BM = 64*PT, BN = 32*CP.
template <class T, int PT, int CP, int D, int MODE>
__global__ __launch_bounds__(256) void pwh_kernel(PwArgs a) {
    constexpr int BM = 64 * PT, BN = 32 * CP, CT = 2 * CP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const unsigned ntn = (a.N + BN - 1) / BN;
    const unsigned L = yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (int)(L / ntn) * BM + wave * 16 * PT;
    const int n0 = (int)(L % ntn) * BN;
    const int kp = a.S.kp;
    const int kl = g * 8;  // this lane's k offset inside a 32-deep chunk

    PwhRow<MODE, T> row[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) row[p].init(a, m0 + p * 16 + li);
    const T* brow[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        // MFMA row li of tile t <-> cout: see the header (lane group g ends up owning couts 8g..8g+7 of the pair)
        const int n = n0 + (t >> 1) * 32 + 8 * (li >> 2) + 4 * (t & 1) + (li & 3);
        brow[t] = reinterpret_cast<const T*>(a.wt) + (size_t)(n < a.N ? n : 0) * kp;  // rows beyond N feed couts that are never stored
    }

    struct Frag {
        pwh_u4 x[PT], w[CT];
        float4 g0[MODE == 2 ? PT : 1], g1[MODE == 2 ? PT : 1];
        int cv[PT];
    };
    auto load = [&](auto pools_tag, int chunk, Frag& F) __attribute__((always_inline)) {
        constexpr bool POOLS = decltype(pools_tag)::value;
        const int kraw = chunk * 32 + kl;
#pragma unroll
        for (int p = 0; p < PT; ++p)
            row[p].template issue<POOLS>(a, kraw, kp, F.x[p], F.g0[MODE == 2 ? p : 0], F.g1[MODE == 2 ? p : 0], F.cv[p]);
        const int k = kraw < kp ? kraw : kp - 8;  // the k tail of the weights meets zeroed activations
#pragma unroll
        for (int t = 0; t < CT; ++t) F.w[t] = *reinterpret_cast<const pwh_u4*>(brow[t] + k);
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[t][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // uniform: most layers have whole octets and whole 32-deep chunks only, and then nothing needs masking
    const bool need_mask = MODE != 1 || (a.S.s[0].c & 7) != 0 || (kp & 31) != 0;

    auto use = [&](const Frag& F) __attribute__((always_inline)) {
        pwh_u4 xf[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p)
            xf[p] = need_mask ? pwh_finish<MODE, T>(F.x[p], F.g0[MODE == 2 ? p : 0], F.g1[MODE == 2 ? p : 0], F.cv[p]) : F.x[p];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int p = 0; p < PT; ++p) acc[t][p] = pwh_mfma<T>(F.w[t], xf[p], acc[t][p]);
    };

    auto run = [&](auto pools_tag) __attribute__((always_inline)) {
        const int nch = (kp + 31) >> 5;  // chunk j lives in F[j % D]
        Frag F[D];
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < nch) load(pools_tag, d, F[d]);
        int j0 = 0;
        for (; j0 + 2 * D - 1 < nch; j0 += D) {  // steady state, branch-free: the whole next group exists
#pragma unroll
            for (int d = 0; d < D; ++d) {
                use(F[d]);
                load(pools_tag, j0 + D + d, F[d]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2 * D - 1; ++t) {    // drain
            const int j = j0 + t;
            if (j < nch) {
                use(F[t % D]);
                if (j + D < nch) load(pools_tag, j + D, F[t % D]);
            }
        }
    };
    bool pooled = false;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < YR_MAX_SRC; ++i)
            pooled |= a.S.s[i].xform == YR_X_MAXPOOL2 || a.S.s[i].xform == YR_X_MAXPOOL4;
    }
    if (pooled) run(std::true_type{});
    else run(std::false_type{});

    // ---- epilogue: lane group g owns couts 8g..8g+7 of every pair (tiles 2c and 2c+1) for its PT pixels
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const int n = n0 + c * 32 + g * 8;
        float sc[8], sh[8];
        pwh_load_bn(a, n, sc, sh);
#pragma unroll
        for (int p = 0; p < PT; ++p) pwh_finish_oct<T>(a, acc[2 * c][p], acc[2 * c + 1][p], sc, sh, m0 + p * 16 + li, n, li);
    }
}

// Small-K form (k space of at most 32*NCH channels, one identity source, optional SE gate): the projections of the
// high-resolution blocks - [millions of pixels] x [32..128 channels] -> [16..48 couts].  In pwh_kernel a wave fetches its
// PT pixel tiles, multiplies and leaves: with a single 32-deep chunk there is nothing to pipeline INSIDE a tile set, so
// every workgroup's life is one exposed HBM round trip (EfficientNet-B0 stage 1 projection, 128 x 208 x 208 pixels:
// 0.29 ms at 1.85 TB/s).  Here a workgroup keeps the weight fragments of its cout tile in registers and WALKS a contiguous
// range of pixel tiles; the next tile's activations (and gate rows) are in flight while the current tile is multiplied
// and stored.  Same MFMA sequence per output as pwh_kernel: bit-identical results.
template <class T, int PT, int CP, int NCH, int MODE>
__global__ __launch_bounds__(256) void pwhp_kernel(PwArgs a, int tiles_per_wg) {
    static_assert(MODE == 1 || MODE == 2, "one identity source");
    constexpr int BM = 64 * PT, BN = 32 * CP, CT = 2 * CP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int n0 = (int)blockIdx.y * BN;
    const int kp = a.S.kp;
    const int kl = g * 8;
    const int ntm = (a.M + BM - 1) / BM;
    const int t_begin = (int)blockIdx.x * tiles_per_wg;
    const int t_end = t_begin + tiles_per_wg < ntm ? t_begin + tiles_per_wg : ntm;
    if (t_begin >= t_end) return;

    pwh_u4 wf[NCH][CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int n = n0 + (t >> 1) * 32 + 8 * (li >> 2) + 4 * (t & 1) + (li & 3);
        const T* brow = reinterpret_cast<const T*>(a.wt) + (size_t)(n < a.N ? n : 0) * kp;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int kraw = ch * 32 + kl;
            wf[ch][t] = *reinterpret_cast<const pwh_u4*>(brow + (kraw < kp ? kraw : kp - 8));
        }
    }
    float sc[CP][8], sh[CP][8];
#pragma unroll
    for (int c = 0; c < CP; ++c) pwh_load_bn(a, n0 + c * 32 + g * 8, sc[c], sh[c]);

    struct Tile {
        pwh_u4 x[NCH][PT];
        float4 g0[MODE == 2 ? NCH : 1][MODE == 2 ? PT : 1], g1[MODE == 2 ? NCH : 1][MODE == 2 ? PT : 1];
        int cv[NCH][PT];
    };
    auto fetch = [&](int mt, Tile& X) __attribute__((always_inline)) {
        const int m0 = mt * BM + wave * 16 * PT;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            PwhRow<MODE, T> row;
            row.init(a, m0 + p * 16 + li);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                row.template issue<false>(a, ch * 32 + kl, kp, X.x[ch][p], X.g0[MODE == 2 ? ch : 0][MODE == 2 ? p : 0],
                                          X.g1[MODE == 2 ? ch : 0][MODE == 2 ? p : 0], X.cv[ch][p]);
        }
    };
    const bool need_mask = MODE != 1 || (a.S.s[0].c & 7) != 0 || (kp & 31) != 0 || kp < 32 * NCH;
    auto compute = [&](int mt, const Tile& X) __attribute__((always_inline)) {
        f32x4 acc[CT][PT];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int p = 0; p < PT; ++p) acc[t][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            pwh_u4 xf[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p)
                xf[p] = need_mask ? pwh_finish<MODE, T>(X.x[ch][p], X.g0[MODE == 2 ? ch : 0][MODE == 2 ? p : 0],
                                                        X.g1[MODE == 2 ? ch : 0][MODE == 2 ? p : 0], X.cv[ch][p]) : X.x[ch][p];
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[t][p] = pwh_mfma<T>(wf[ch][t], xf[p], acc[t][p]);
        }
        const int m0 = mt * BM + wave * 16 * PT;
#pragma unroll
        for (int c = 0; c < CP; ++c)
#pragma unroll
            for (int p = 0; p < PT; ++p)
                pwh_finish_oct<T>(a, acc[2 * c][p], acc[2 * c + 1][p], sc[c], sh[c], m0 + p * 16 + li, n0 + c * 32 + g * 8, li);
    };
    Tile cur, nxt;
    fetch(t_begin, cur);
    for (int mt = t_begin; mt < t_end; ++mt) {
        if (mt + 1 < t_end) fetch(mt + 1, nxt);   // (uniform) the next tile's loads are in flight during this tile's MFMAs and stores
        compute(mt, cur);
        cur = nxt;                                // register moves: cheaper than a second copy of the multiply + epilogue code
    }
}

// LDS-tiled form: a workgroup computes up to 128 pixels x 128
// couts.  The direct kernel above re-fetches every operand fragment per wave - with PT = 4, CP = 1 six 1 KB loads per
// eight MFMAs - and the mid-size GEMMs of the unfused EfficientNet stages ([13k..51k pixels] x [136..1392] x [136..1392])
// ran at 1.0-1.7 TB/s whatever the tile shape (tools/pwh_probe.py): bound by the L1/L2 -> register path, not by HBM or
// the matrix pipe.  Here both operands of a 32-deep chunk are fetched once per workgroup (four 16-byte loads per thread),
// parked in LDS in FRAGMENT order - tile of 16 rows = 64 lanes x 16 bytes, lane (k group g, row i) at slot 16g + i, so
// the store of a wave's loads and every ds_read_b128 are linear, conflict-free - and each wave multiplies a 64 x 64
// sub-tile (4 + 4 fragment reads per 16 MFMAs).  Two LDS buffers, one barrier per chunk, the next chunk's global loads in
// flight during the MFMAs.  Same MFMA sequence and operand mapping per output as pwh_kernel: bit-identical results.
// PT x CT: 16 x 16 MFMA tiles per wave (pixels x couts); 2 x 2 waves: BM = 32 PT, BN = 32 CT.  MODE as in PwhRow: the
// pixel operand goes through the same row object as in pwh_kernel (gathers, pooled sources, SE gate and pad masking
// included) - only WHO fetches an octet differs: loader thread (row tid >> 2, k group tid & 3) instead of MFMA lane.
template <class T, int PT, int CT, int MODE, bool POOLS>
__global__ __launch_bounds__(256, 3) void pwhl_kernel(PwArgs a) {
    constexpr int BM = 32 * PT, BN = 32 * CT;
    constexpr int NWH = BN / 64, NXH = BM / 64;     // rows per loader thread
    __shared__ pwh_u4 frag[2][2 * CT + 2 * PT][64];   // [buffer][16-row tile: weights first, then pixels][fragment lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int wm = wave & 1, wn = wave >> 1;
    const unsigned ntn = (a.N + BN - 1) / BN;
    const unsigned L = yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (int)(L / ntn) * BM, n0 = (int)(L % ntn) * BN;
    const int kp = a.S.kp, nch = (kp + 31) >> 5;

    // loader: thread (row r = tid >> 2, k group lg = tid & 3) fetches rows r, r + 64 .. of both operands
    const int lr = tid >> 2, lg = tid & 3;
    const T* wrow[NWH];
    PwhRow<MODE, T> xrow[NXH];
#pragma unroll
    for (int h = 0; h < NWH; ++h) {
        const int rho = lr + 64 * h, tile = rho >> 4, i = rho & 15;
        const int n = n0 + (tile >> 1) * 32 + 8 * (i >> 2) + 4 * (tile & 1) + (i & 3);   // MFMA row i of tile <-> cout: as in pwh_kernel
        wrow[h] = reinterpret_cast<const T*>(a.wt) + (size_t)(n < a.N ? n : 0) * kp;
    }
#pragma unroll
    for (int h = 0; h < NXH; ++h) xrow[h].init(a, m0 + lr + 64 * h);
    const int slot = (lr >> 4) * 64 + lg * 16 + (lr & 15);   // + 256 per further 64 rows
    // uniform: most layers have whole octets and whole 32-deep chunks only, and then nothing needs masking
    const bool need_mask = MODE != 1 || (a.S.s[0].c & 7) != 0 || (kp & 31) != 0;
    struct Stage {
        pwh_u4 w[NWH], x[NXH];
        float4 g0[MODE == 2 ? NXH : 1], g1[MODE == 2 ? NXH : 1];
        int cv[NXH];
    };
    auto fetch = [&](int chunk, Stage& R) __attribute__((always_inline)) {
        const int kraw = chunk * 32 + lg * 8;
        const int k = kraw < kp ? kraw : kp - 8;   // the k tail of the weights meets zeroed activations
#pragma unroll
        for (int h = 0; h < NWH; ++h) R.w[h] = *reinterpret_cast<const pwh_u4*>(wrow[h] + k);
#pragma unroll
        for (int h = 0; h < NXH; ++h)
            xrow[h].template issue<POOLS>(a, kraw, kp, R.x[h], R.g0[MODE == 2 ? h : 0], R.g1[MODE == 2 ? h : 0], R.cv[h]);
    };
    auto park = [&](int buf, const Stage& R) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < NWH; ++h) (&frag[buf][0][0])[slot + 256 * h] = R.w[h];
#pragma unroll
        for (int h = 0; h < NXH; ++h)
            (&frag[buf][2 * CT][0])[slot + 256 * h] =
                need_mask ? pwh_finish<MODE, T>(R.x[h], R.g0[MODE == 2 ? h : 0], R.g1[MODE == 2 ? h : 0], R.cv[h]) : R.x[h];
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[t][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto multiply = [&](int buf) __attribute__((always_inline)) {
        pwh_u4 w[CT], x[PT];
#pragma unroll
        for (int t = 0; t < CT; ++t) w[t] = frag[buf][wn * CT + t][lane];
#pragma unroll
        for (int p = 0; p < PT; ++p) x[p] = frag[buf][2 * CT + wm * PT + p][lane];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int p = 0; p < PT; ++p) acc[t][p] = pwh_mfma<T>(w[t], x[p], acc[t][p]);
    };

    // chunk c is multiplied out of LDS buffer c & 1 while chunk c + 1 is in flight into registers.  One barrier per chunk:
    // the buffer parked into during chunk c is the one chunk c - 1 was read from, and everyone has passed the barrier
    // since.  (A second register stage - chunk c + 2 in flight as well - costs 10 VGPRs = one wave per SIMD on the 128 x 64
    // tile and measured 10-20 % SLOWER: waves in flight hide more latency than loads in flight per wave; three or four
    // stages did not win on the long-k projections of the 20 x 20 maps either - tools/pwh_probe.py.)
    Stage R;
    fetch(0, R);
    park(0, R);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) fetch(c + 1, R);   // (uniform)
        multiply(c & 1);
        if (c + 1 < nch) park((c & 1) ^ 1, R);
        __syncthreads();
    }

#pragma unroll
    for (int c = 0; c < CT / 2; ++c) {
        const int n = n0 + wn * 16 * CT + c * 32 + g * 8;
        float sc[8], sh[8];
        pwh_load_bn(a, n, sc, sh);
#pragma unroll
        for (int p = 0; p < PT; ++p) pwh_finish_oct<T>(a, acc[2 * c][p], acc[2 * c + 1][p], sc, sh, m0 + wm * 16 * PT + p * 16 + li, n, li);
    }
}

template <class T, int PT, int CP>
static int launch_h(const PwArgs& a, hipStream_t s);

template <class T, int PT, int CT>
static int launch_lds(const PwArgs& a, hipStream_t s) {
    if (a.dw_w) { yr_set_error("pointwise: the depthwise-folded source is a float32 feature"); return YR_ERR_ARG; }
    constexpr int BM = 32 * PT, BN = 32 * CT;
    const int mode = (a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY) ? (a.gate ? 2 : 1) : 0;
    if (mode == 0 && a.gate) { yr_set_error("pointwise: an SE gate needs one identity source"); return YR_ERR_ARG; }
    bool pooled = false;
    for (int i = 0; i < YR_MAX_SRC; ++i) pooled |= a.S.s[i].xform == YR_X_MAXPOOL2 || a.S.s[i].xform == YR_X_MAXPOOL4;
    dim3 grid((unsigned)((a.M + BM - 1) / BM) * (unsigned)((a.N + BN - 1) / BN));
    static char nm[4][48];   // spelled like the symbols: element type, PT, CT, MODE, POOLS
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pwhl_kernel<%s,%d,%d,0,0>", yr_dtype_name(yr_elem<T>::dtype), PT, CT) +
                              snprintf(nm[1], sizeof(nm[1]), "pwhl_kernel<%s,%d,%d,1,0>", yr_dtype_name(yr_elem<T>::dtype), PT, CT) +
                              snprintf(nm[2], sizeof(nm[2]), "pwhl_kernel<%s,%d,%d,2,0>", yr_dtype_name(yr_elem<T>::dtype), PT, CT) +
                              snprintf(nm[3], sizeof(nm[3]), "pwhl_kernel<%s,%d,%d,0,1>", yr_dtype_name(yr_elem<T>::dtype), PT, CT);
    (void)nm_len;
    yr_note_kernel(nm[mode == 0 && pooled ? 3 : mode]);
    if (mode == 1) hipLaunchKernelGGL((pwhl_kernel<T, PT, CT, 1, false>), grid, dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL((pwhl_kernel<T, PT, CT, 2, false>), grid, dim3(256), 0, s, a);
    else if (pooled) hipLaunchKernelGGL((pwhl_kernel<T, PT, CT, 0, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pwhl_kernel<T, PT, CT, 0, false>), grid, dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T, int PT, int CP, int NCH>
static int launch_hp(const PwArgs& a, int mode, hipStream_t s) {
    constexpr int BM = 64 * PT, BN = 32 * CP;
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    // about 8 workgroups per CU in flight and at least 4 tiles per workgroup (the first tile's fetch is exposed)
    int per = (ntm + 2047) / 2048;
    if (per < 4) per = 4;
    const int wgs = (ntm + per - 1) / per;
    static char nm[2][48];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pwhp_kernel<%s,%d,%d,%d,1>", yr_dtype_name(yr_elem<T>::dtype), PT, CP, NCH) +
                              snprintf(nm[1], sizeof(nm[1]), "pwhp_kernel<%s,%d,%d,%d,2>", yr_dtype_name(yr_elem<T>::dtype), PT, CP, NCH);
    (void)nm_len;
    yr_note_kernel(nm[mode - 1]);
    const dim3 grid((unsigned)wgs, (unsigned)ntn);
    if (mode == 1) hipLaunchKernelGGL((pwhp_kernel<T, PT, CP, NCH, 1>), grid, dim3(256), 0, s, a, per);
    else hipLaunchKernelGGL((pwhp_kernel<T, PT, CP, NCH, 2>), grid, dim3(256), 0, s, a, per);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T, int PT, int CP>
static int launch_h(const PwArgs& a, hipStream_t s) {
    constexpr int BM = 64 * PT, BN = 32 * CP;
    // chunks of loads in flight; four pixel tiles per wave keep two (three spilled 464-544 bytes per lane in the f16 gather form)
    constexpr int D = PT >= 4 ? 2 : (PT + 2 * CP <= 4 ? 4 : (PT + 2 * CP <= 8 ? 3 : 2));
    dim3 grid((unsigned)((a.M + BM - 1) / BM) * (unsigned)((a.N + BN - 1) / BN));
    const int mode = (a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY) ? (a.gate ? 2 : 1) : 0;
    if (mode == 0 && a.gate) { yr_set_error("pointwise: an SE gate needs one identity source"); return YR_ERR_ARG; }
    static char nm[3][48];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pwh_kernel<%s,%d,%d,%d,0>", yr_dtype_name(yr_elem<T>::dtype), PT, CP, D) +
                              snprintf(nm[1], sizeof(nm[1]), "pwh_kernel<%s,%d,%d,%d,1>", yr_dtype_name(yr_elem<T>::dtype), PT, CP, D) +
                              snprintf(nm[2], sizeof(nm[2]), "pwh_kernel<%s,%d,%d,%d,2>", yr_dtype_name(yr_elem<T>::dtype), PT, CP, D);
    (void)nm_len;
    yr_note_kernel(nm[mode]);
    if (mode == 1) hipLaunchKernelGGL((pwh_kernel<T, PT, CP, D, 1>), grid, dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL((pwh_kernel<T, PT, CP, D, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pwh_kernel<T, PT, CP, D, 0>), grid, dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

struct PwhCfg { int bm, bn; };
static const PwhCfg pwh_cfgs[] = {{64, 32}, {64, 64}, {64, 96}, {64, 128},
                                  {128, 32}, {128, 64}, {128, 96}, {128, 128}, {256, 32}, {256, 64}};
constexpr int PWH_NCFG = sizeof(pwh_cfgs) / sizeof(pwh_cfgs[0]);
// + the small-K walking form (pwhp_kernel) in four shapes; where it does not apply (more than 128 k, gathered sources)
// these indices run the plain kernel of the same tile shape, so every index is valid for every op
constexpr int PWH_NWALK = 8;   // ... and the LDS-tiled form (pwhl_kernel) in four shapes
// ... and the activation-stationary form (pointwise_hs.hip, k <= 256) and the all-couts k-streaming form (pointwise_hq.hip)
// in four variants each as the last indices; ops they do not take run the 128 x 64 LDS-tiled shape
constexpr int PWH_NSQ = 8;
int yr_pwh_num_cfgs() { return PWH_NCFG + PWH_NWALK + PWH_NSQ; }

template <class T, int PT, int CP>
static int launch_h(const PwArgs& a, hipStream_t s);

template <class T, int PT, int CP>
static int launch_walk(const PwArgs& a, hipStream_t s) {
    const int mode = (a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY) ? (a.gate ? 2 : 1) : 0;
    const int nch = (a.S.kp + 31) >> 5;
    if (mode == 0 || nch > 4 || a.dw_w != nullptr) return launch_h<T, PT, CP>(a, s);
    if (nch <= 1) return launch_hp<T, PT, CP, 1>(a, mode, s);
    if (nch <= 2) return launch_hp<T, PT, CP, 2>(a, mode, s);
    return launch_hp<T, PT, CP, 4>(a, mode, s);
}

template <class T>
static int launch_h_cfg(int cfg, const PwArgs& a, hipStream_t s) {
    switch (cfg) {
        case 0: return launch_h<T, 1, 1>(a, s);
        case 1: return launch_h<T, 1, 2>(a, s);
        case 2: return launch_h<T, 1, 3>(a, s);
        case 3: return launch_h<T, 1, 4>(a, s);
        case 4: return launch_h<T, 2, 1>(a, s);
        case 5: return launch_h<T, 2, 2>(a, s);
        case 6: return launch_h<T, 2, 3>(a, s);
        case 7: return launch_lds<T, 2, 4>(a, s);   // (the direct kernel of this 128 x 128 shape spilled 272 bytes per lane: its LDS-tiled twin)
        case 8: return launch_h<T, 4, 1>(a, s);
        case 9: return launch_h<T, 4, 2>(a, s);
        case 10: return launch_walk<T, 1, 1>(a, s);
        case 11: return launch_walk<T, 2, 1>(a, s);
        case 12: return launch_walk<T, 1, 2>(a, s);
        case 13: return launch_walk<T, 2, 2>(a, s);
        case 14: return launch_lds<T, 4, 4>(a, s);
        case 15: return launch_lds<T, 2, 4>(a, s);
        case 16: return launch_lds<T, 4, 2>(a, s);
        case 17: return launch_lds<T, 2, 2>(a, s);
        case 18: case 19: case 20: case 21: {
            const int rc = yr_pwhs_launch(yr_elem<T>::dtype, cfg - 18, a, s);
            return rc == YR_NOT_TAKEN ? launch_lds<T, 4, 2>(a, s) : rc;
        }
        case 22: case 23: case 24: case 25: {
            const int rc = yr_pwhq_launch(yr_elem<T>::dtype, cfg - 22, a, s);
            return rc == YR_NOT_TAKEN ? launch_lds<T, 4, 2>(a, s) : rc;
        }
        default: yr_set_error("pointwise: 16-bit tile shape %d out of range", cfg); return YR_ERR_ARG;
    }
}

// The K-SPLIT form for the passes of one or two images (se_reduced bit 17 of a POINTWISE op, set by the compiler's 'latency' variant of
// a 16-bit plan for its small maps; the float32 twin is pointwise_split.hip's pwk_kernel).  pwh_kernel gives a wave 16 pixels x 32
// couts and the whole k range: at 169 .. 1024 pixels a gated projection of 1152 channels is 36 chunks behind each other (9.8 us at
// one image).  Here a workgroup is ONE such tile and its four waves take the chunks w, w + 4, ... (PWKH_G of them in flight per wave);
// the four accumulator pairs meet in LDS in wave order and wave 0 runs pwh_kernel's epilogue.
#define PWKH_G 3
template <class T, int MODE>
__global__ __launch_bounds__(256) void pwkh_kernel(PwArgs a) {
    __shared__ f32x4 red[3][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const unsigned ntn = (a.N + 31) / 32;
    const unsigned L = yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (int)(L / ntn) * 16, n0 = (int)(L % ntn) * 32;
    const int kp = a.S.kp, nch = (kp + 31) >> 5;
    PwhRow<MODE, T> row;
    row.init(a, m0 + li);
    const T* brow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + 8 * (li >> 2) + 4 * t + (li & 3);   // (pwh_kernel's row assignment: lane group g ends up owning couts 8 g .. 8 g + 7)
        brow[t] = reinterpret_cast<const T*>(a.wt) + (size_t)(n < a.N ? n : 0) * kp;
    }
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    auto run = [&](auto pools_tag) __attribute__((always_inline)) {
        constexpr bool POOLS = decltype(pools_tag)::value;
        for (int c0 = wave; c0 < nch; c0 += 4 * PWKH_G) {
            pwh_u4 x[PWKH_G], w[PWKH_G][2];
            float4 g0[PWKH_G], g1[PWKH_G];
            int cv[PWKH_G];
            pw_unroll<PWKH_G>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const int kraw = (c0 + 4 * j) * 32 + g * 8;       // (beyond kp for a dead chunk: clamped addresses, cv = 0)
                row.template issue<POOLS>(a, kraw, kp, x[j], g0[j], g1[j], cv[j]);
                const int k = kraw < kp ? kraw : kp - 8;
                w[j][0] = *reinterpret_cast<const pwh_u4*>(brow[0] + k);
                w[j][1] = *reinterpret_cast<const pwh_u4*>(brow[1] + k);
            });
            pw_unroll<PWKH_G>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                if (c0 + 4 * j < nch) {      // (wave-uniform)
                    const pwh_u4 xf = pwh_finish<MODE, T>(x[j], g0[j], g1[j], cv[j]);
                    acc[0] = pwh_mfma<T>(w[j][0], xf, acc[0]);
                    acc[1] = pwh_mfma<T>(w[j][1], xf, acc[1]);
                }
            });
        }
    };
    bool pooled = false;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < YR_MAX_SRC; ++i)
            pooled |= a.S.s[i].xform == YR_X_MAXPOOL2 || a.S.s[i].xform == YR_X_MAXPOOL4;
    }
    if (pooled) run(std::true_type{});
    else run(std::false_type{});
    if (wave > 0) { red[wave - 1][0][lane] = acc[0]; red[wave - 1][1][lane] = acc[1]; }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) { acc[0] += red[w][0][lane]; acc[1] += red[w][1][lane]; }   // (in wave order)
    const int n = n0 + g * 8;
    float sc[8], sh[8];
    pwh_load_bn(a, n, sc, sh);
    pwh_finish_oct<T>(a, acc[0], acc[1], sc, sh, m0 + li, n, li);
}

template <class T>
static int launch_ksplit_t(const PwArgs& a, hipStream_t s) {
    const int mode = (a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY) ? (a.gate ? 2 : 1) : 0;
    if (mode == 0 && a.gate) { yr_set_error("pointwise: an SE gate needs one identity source"); return YR_ERR_ARG; }
    dim3 grid((unsigned)((a.M + 15) / 16) * (unsigned)((a.N + 31) / 32));
    static char nm[3][40];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pwkh_kernel<%s,0>", yr_dtype_name(yr_elem<T>::dtype)) +
                              snprintf(nm[1], sizeof(nm[1]), "pwkh_kernel<%s,1>", yr_dtype_name(yr_elem<T>::dtype)) +
                              snprintf(nm[2], sizeof(nm[2]), "pwkh_kernel<%s,2>", yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    yr_note_kernel(nm[mode]);
    if (mode == 1) hipLaunchKernelGGL((pwkh_kernel<T, 1>), grid, dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL((pwkh_kernel<T, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pwkh_kernel<T, 0>), grid, dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}
int yr_pwh_launch_ksplit(int dtype, const PwArgs& a, hipStream_t s) {
    if (dtype == YR_BF16) return launch_ksplit_t<yr_bf16>(a, s);
    if (dtype == YR_F16) return launch_ksplit_t<yr_f16>(a, s);
    yr_set_error("pointwise: dtype %d is not a 16-bit type", dtype);
    return YR_ERR_ARG;
}

// 16-bit ops: a filled PwArgs (yr_launch_pointwise did the argument checks that do not depend on the element type);
// cfg = op.k - 1 (autotuned tile shape) or -1: heuristic.
int yr_pw_launch_h(int dtype, int cfg, const PwArgs& a, hipStream_t s) {
    if (cfg < 0 || cfg >= PWH_NCFG + PWH_NWALK + PWH_NSQ) {
        // heuristic: the widest cout tile that still yields ~2 workgroups per CU, 128-row tiles when pixels abound
        const double Md = (double)a.M;
        double best = 1e30;
        for (int i = 0; i < PWH_NCFG; ++i) {
            const PwhCfg& c = pwh_cfgs[i];
            const double ntn = (double)((a.N + c.bn - 1) / c.bn), nblk = (double)((a.M + c.bm - 1) / c.bm) * ntn;
            const double t_mem = Md * a.S.kp * 2.0 * (1.0 + 0.25 * (ntn - 1.0)) + Md * a.N * 2.0 + nblk * c.bn * a.S.kp * 2.0 * 0.25;
            const double fill = nblk < 512.0 ? 512.0 / nblk : 1.0;
            const double regs = (c.bm / 64) * (c.bn / 16) > 16 ? 1.3 : 1.0;
            const double cost = t_mem * fill * regs;
            if (cost < best) { best = cost; cfg = i; }
        }
    }
    if (dtype == YR_BF16) return launch_h_cfg<yr_bf16>(cfg, a, s);
    if (dtype == YR_F16) return launch_h_cfg<yr_f16>(cfg, a, s);
    yr_set_error("pointwise: dtype %d is not a 16-bit type", dtype);
    return YR_ERR_ARG;
}
