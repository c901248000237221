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

#include "pw_common.h"

template <class T> using pwh_v8 = T __attribute__((ext_vector_type(8)));
typedef float pwh_f8 __attribute__((ext_vector_type(8)));
typedef unsigned pwh_u4 __attribute__((ext_vector_type(4)));
typedef pwh_u4 __attribute__((address_space(1))) pwh_gu4;   // 16 bytes in GLOBAL memory (global_load, not flat_load)

template <class T>
__device__ __forceinline__ f32x4 pwh_mfma(pwh_u4 w, pwh_u4 x, f32x4 acc) {
    if constexpr (yr_elem<T>::dtype == YR_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pwh_v8<__bf16>, w), __builtin_bit_cast(pwh_v8<__bf16>, x), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pwh_v8<_Float16>, w), __builtin_bit_cast(pwh_v8<_Float16>, x), acc, 0, 0, 0);
}

template <class T> __device__ __forceinline__ pwh_f8 pwh_widen(pwh_u4 v) { return __builtin_convertvector(__builtin_bit_cast(pwh_v8<T>, v), pwh_f8); }
template <class T> __device__ __forceinline__ pwh_u4 pwh_narrow(pwh_f8 v) { return __builtin_bit_cast(pwh_u4, __builtin_convertvector(v, pwh_v8<T>)); }

// elementwise maximum of two 16-bit octets (MaxPooling2D folded into the loads): widening is exact, so
// narrow(max(widen)) selects one of the inputs bit for bit
template <class T>
__device__ __forceinline__ pwh_u4 pwh_max(pwh_u4 a, pwh_u4 b) {
    const pwh_f8 x = pwh_widen<T>(a), y = pwh_widen<T>(b);
    pwh_f8 m;
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = fmaxf(x[i], y[i]);
    return pwh_narrow<T>(m);
}

// zero the elements >= cv of an octet (pad channels of a source may hold anything; k beyond kp: cv <= 0) - by select
__device__ __forceinline__ pwh_u4 pwh_mask(pwh_u4 v, int cv) {
#pragma unroll
    for (int d = 0; d < 4; ++d) v[d] = cv >= 2 * d + 2 ? v[d] : (cv == 2 * d + 1 ? (v[d] & 0xffffu) : 0u);
    return v;
}

// One GEMM row (pixel) of the activation operand; the 16-bit twin of PwRow (pw_common.h) with 8-channel vectors.
// MODE 0: generic gather (upsample / maxpool / concat sources), 1: one identity source, 2: identity + SE gate.
template <int MODE, class T>
struct PwhRow {
    const T* arow;                // MODE != 0: the pixel's contiguous row
    const float* grow;            // MODE == 2: SE gate row of the pixel's image (float32)
    const T *s0, *s1, *s2, *s3;   // MODE == 0: per-source row pointer (xform folded in), pre-offset by -kbase
    bool valid;

    static __device__ __forceinline__ const T* source_row(const DSrc& d, int b, int y, int x) {
        int sy = y, sx = x;
        if (d.xform == YR_X_UP2) { sy = y >> 1; sx = x >> 1; }
        else if (d.xform == YR_X_MAXPOOL2) { sy = y * 2; sx = x * 2; }
        else if (d.xform == YR_X_MAXPOOL4) { sy = y * 4; sx = x * 4; }
        return reinterpret_cast<const T*>(d.ptr) + ((size_t)(b * d.h + sy) * d.w + sx) * d.ld;
    }

    __device__ __forceinline__ void init(const PwArgs& a, int m) {
        valid = m < a.M;
        const int mm = pw_pixel_of_row(a, valid ? m : 0);
        const int hw = a.H * a.W;
        const int b = mm / hw;
        grow = MODE == 2 ? a.gate + (size_t)b * a.gate_ld : nullptr;
        arow = s0 = s1 = s2 = s3 = nullptr;
        if (MODE != 0) {
            arow = reinterpret_cast<const T*>(a.S.s[0].ptr) + (size_t)mm * a.S.s[0].ld;
        } else {
            const int rem = mm - b * hw;
            const int y = rem / a.W, x = rem - y * a.W;
            s0 = source_row(a.S.s[0], b, y, x) - a.S.s[0].kbase;
            s1 = source_row(a.S.s[1], b, y, x) - a.S.s[1].kbase;
            s2 = source_row(a.S.s[2], b, y, x) - a.S.s[2].kbase;
            s3 = source_row(a.S.s[3], b, y, x) - a.S.s[3].kbase;
        }
    }

    // Issue the loads of the octet at k (raw k may lie beyond kp: clamped).  v: raw channels, g0/g1: the gate's two
    // quads (MODE 2), cv: how many of the octet's channels are real (<= 0: none).  As in PwRow::issue nothing here
    // reads a loaded register and the main load is unconditional (see pw_common.h for why).
    template <bool POOLS = true>
    __device__ __forceinline__ void issue(const PwArgs& a, int kraw, int kp, pwh_u4& v, float4& g0, float4& g1, int& cv) const {
        const int k = kraw < kp ? kraw : kp - 8;
        int cvalid;
        if (MODE != 0) {
            v = *reinterpret_cast<const pwh_u4*>(arow + k);
            cvalid = a.S.s[0].c - k;
            if (MODE == 2) {
                g0 = *reinterpret_cast<const float4*>(grow + k);
                g1 = *reinterpret_cast<const float4*>(grow + k + 4);
            }
        } else {
            const bool q1 = k >= a.S.s[1].kbase, q2 = k >= a.S.s[2].kbase, q3 = k >= a.S.s[3].kbase;
            const T *p0 = s0, *p1 = s1, *p2 = s2, *p3 = s3;
            asm("" : "+v"(p0));
            asm("" : "+v"(p1));
            asm("" : "+v"(p2));
            asm("" : "+v"(p3));
            const T* rp = q3 ? p3 : q2 ? p2 : q1 ? p1 : p0;
#define PWH_PICK(name, e0, e1, e2, e3)                                                                  \
    const int name##0 = __builtin_amdgcn_readfirstlane(e0), name##1 = __builtin_amdgcn_readfirstlane(e1), \
              name##2 = __builtin_amdgcn_readfirstlane(e2), name##3 = __builtin_amdgcn_readfirstlane(e3); \
    const int name = q3 ? name##3 : q2 ? name##2 : q1 ? name##1 : name##0;
            PWH_PICK(kend, a.S.s[0].kbase + a.S.s[0].c, a.S.s[1].kbase + a.S.s[1].c, a.S.s[2].kbase + a.S.s[2].c,
                     a.S.s[3].kbase + a.S.s[3].c)
            cvalid = kend - k;
            const pwh_gu4* q = (const pwh_gu4*)(rp + k);
            v = q[0];
            if (POOLS) {
                PWH_PICK(xf, a.S.s[0].xform, a.S.s[1].xform, a.S.s[2].xform, a.S.s[3].xform)
                PWH_PICK(sw, a.S.s[0].w, a.S.s[1].w, a.S.s[2].w, a.S.s[3].w)
                PWH_PICK(sld, a.S.s[0].ld, a.S.s[1].ld, a.S.s[2].ld, a.S.s[3].ld)
                if (xf == YR_X_MAXPOOL2) {  // the three other taps are issued together (sld % 8 == 0: whole 16-byte steps)
                    const pwh_u4 v1 = q[sld >> 3], v2 = q[((size_t)sw * sld) >> 3], v3 = q[(((size_t)sw + 1) * sld) >> 3];
                    v = pwh_max<T>(pwh_max<T>(v, v1), pwh_max<T>(v2, v3));
                } else if (xf == YR_X_MAXPOOL4) {
                    for (int dy = 0; dy < 4; ++dy)
                        for (int dx = 0; dx < 4; ++dx) v = pwh_max<T>(v, q[(((size_t)dy * sw + dx) * sld) >> 3]);
                }
            }
#undef PWH_PICK
        }
        cv = (valid && kraw < kp) ? cvalid : 0;
    }
};

// the fetched octet with pad lanes zeroed and (MODE 2) multiplied by the SE gate: widened, one float32 product per
// channel (efficientnet.py:435 `se_tensor * input_tensor`), rounded back to the operand type
template <int MODE, class T>
__device__ __forceinline__ pwh_u4 pwh_finish(pwh_u4 v, const float4& g0, const float4& g1, int cv) {
    if (MODE == 2) {
        pwh_f8 x = pwh_widen<T>(v);
        x[0] *= g0.x; x[1] *= g0.y; x[2] *= g0.z; x[3] *= g0.w;
        x[4] *= g1.x; x[5] *= g1.y; x[6] *= g1.z; x[7] *= g1.w;
        v = pwh_narrow<T>(x);
    }
    return pwh_mask(v, cv);
}

// eight float32 values n..n+7 of row `row` of a [.][ld] float32 array with N real columns; columns >= N re-read a
// valid one (their results are never used).  Unconditional loads (see pw_load_quad).
__device__ __forceinline__ void pwh_load8_f32(const float* base, size_t row, int ld, int n, int N, float (&q)[8]) {
    const float* rp = base + row * ld;
    if ((ld & 3) == 0 && n + 8 <= ld) {
        const float4 t0 = *reinterpret_cast<const float4*>(rp + n), t1 = *reinterpret_cast<const float4*>(rp + n + 4);
        q[0] = t0.x; q[1] = t0.y; q[2] = t0.z; q[3] = t0.w; q[4] = t1.x; q[5] = t1.y; q[6] = t1.z; q[7] = t1.w;
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) q[r] = rp[n + r < N ? n + r : N - 1];
    }
}

// Epilogue of one accumulator OCTET: GEMM row m (may lie beyond M), couts n..n+7 (n may lie beyond N).  (Pre-BN
// addend,) BN scale/shift, activation, (residual,) (2x2 max,) store - all in float32, one rounding at the store.
template <class T>
__device__ __forceinline__ void pwh_finish_oct(const PwArgs& a, const f32x4& lo, const f32x4& hi, const float (&sc)[8],
                                               const float (&sh)[8], int m, int n, int li) {
    const int cnt = a.N - n;           // real couts in this octet (<= 0: none)
    const int nld = cnt > 0 ? n : 0;   // the column dead octets load from
    const int ml = m < a.M ? m : a.M - 1;
    float v[8], q[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = lo[r]; v[4 + r] = hi[r]; }
    if (a.pre) {  // uniform: the low-resolution share of a hoisted concat conv (float32) joins the accumulator before BN
        pwh_load8_f32(a.pre, pw_pre_row(a, ml), a.pre_ld, nld, a.N, q);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += q[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float t = __builtin_fmaf(v[r], sc[r], sh[r]);
        v[r] = a.out_f32 ? yr_apply_act(t, a.act) : yr_apply_act_t<T>(t, a.act);   // (uniform; float32 outputs keep the pinned path)
    }
    if (a.res) {  // uniform; the residual has the op's 16-bit type and a pitch that is a multiple of 8
        const pwh_f8 t = pwh_widen<T>(*reinterpret_cast<const pwh_u4*>(reinterpret_cast<const T*>(a.res) + (size_t)ml * a.res_ld + nld));
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += t[r];
    }
    int orow = m;
    bool keep = m < a.M && cnt > 0;
    if (a.pool) {  // uniform: MaxPooling2D(2) across the 4 adjacent lanes of a window
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r] = fmaxf(v[r], __shfl_xor(v[r], 1));
            v[r] = fmaxf(v[r], __shfl_xor(v[r], 2));
        }
        keep = keep && (li & 3) == 0;
        orow = m >> 2;
    }
    if (!keep) return;
    if (!a.out_f32) {  // 16-bit output: its pitch covers round_up(N, 8), pad channels may hold anything
        pwh_f8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = v[r];
        *reinterpret_cast<pwh_u4*>(reinterpret_cast<T*>(a.out) + (size_t)orow * a.out_ld + n) = pwh_narrow<T>(o);
        return;
    }
    // float32 output (the logit tensors: dense rows, only dword aligned; the hoisted partial sums)
    float* op = a.out + (size_t)orow * a.out_ld + n;
    const bool vec = (a.out_ld & 3) == 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c4 = cnt - 4 * h;
        if (c4 >= 4) {
            if (vec) *reinterpret_cast<float4*>(op + 4 * h) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
            else *reinterpret_cast<f32x4u*>(op + 4 * h) = (f32x4u){v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (r < c4) op[4 * h + r] = v[4 * h + r];
        }
    }
}

// BatchNorm scale / shift of couts n..n+7 (n a multiple of 8).  Whole octets inside N of 16-byte aligned arrays: two
// 16-byte loads each - as per-element loads they were 16 of a wave's 62 load instructions on a 5-chunk GEMM; couts
// beyond N re-read the last one (their results are never stored).
__device__ __forceinline__ void pwh_load_bn(const PwArgs& a, int n, float (&sc)[8], float (&sh)[8]) {
    if (n + 8 <= a.N && a.scale && a.shift && (((uintptr_t)a.scale | (uintptr_t)a.shift) & 15) == 0) {
        const float4 s0 = *reinterpret_cast<const float4*>(a.scale + n), s1 = *reinterpret_cast<const float4*>(a.scale + n + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(a.shift + n), h1 = *reinterpret_cast<const float4*>(a.shift + n + 4);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
        return;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int nc = n + r < a.N ? n + r : a.N - 1;
        sc[r] = a.scale ? a.scale[nc] : 1.f;
        sh[r] = a.shift ? a.shift[nc] : 0.f;
    }
}

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
    if (a.dw_w) return launch_h<T, PT / 2, CT>(a, s);   // (float32 plans only) the direct kernel of the same tile shape
    constexpr int BM = 32 * PT, BN = 32 * CT;
    const int mode = (a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY) ? (a.gate ? 2 : 1) : 0;
    if (mode == 0 && a.gate) { yr_set_error("pointwise: an SE gate needs one identity source"); return YR_ERR_ARG; }
    bool pooled = false;
    for (int i = 0; i < YR_MAX_SRC; ++i) pooled |= a.S.s[i].xform == YR_X_MAXPOOL2 || a.S.s[i].xform == YR_X_MAXPOOL4;
    dim3 grid((unsigned)((a.M + BM - 1) / BM) * (unsigned)((a.N + BN - 1) / BN));
    static char nm[3][40];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pwhl_kernel<%s,%d,%d,0>", yr_dtype_name(yr_elem<T>::dtype), PT, CT) +
                              snprintf(nm[1], sizeof(nm[1]), "pwhl_kernel<%s,%d,%d,1>", yr_dtype_name(yr_elem<T>::dtype), PT, CT) +
                              snprintf(nm[2], sizeof(nm[2]), "pwhl_kernel<%s,%d,%d,2>", yr_dtype_name(yr_elem<T>::dtype), PT, CT);
    (void)nm_len;
    yr_note_kernel(nm[mode]);
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
    constexpr int D = PT + 2 * CP <= 4 ? 4 : (PT + 2 * CP <= 8 ? 3 : 2);
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
constexpr int PWH_NWALK = 8;   // ... and the LDS-tiled form (pwhl_kernel) in four shapes as the last indices
int yr_pwh_num_cfgs() { return PWH_NCFG + PWH_NWALK; }

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
        case 7: return launch_h<T, 2, 4>(a, s);
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
        default: yr_set_error("pointwise: 16-bit tile shape %d out of range", cfg); return YR_ERR_ARG;
    }
}

// 16-bit ops: a filled PwArgs (yr_launch_pointwise did the argument checks that do not depend on the element type);
// cfg = op.k - 1 (autotuned tile shape) or -1: heuristic.
int yr_pw_launch_h(int dtype, int cfg, const PwArgs& a, hipStream_t s) {
    if (cfg < 0 || cfg >= PWH_NCFG + PWH_NWALK) {
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
