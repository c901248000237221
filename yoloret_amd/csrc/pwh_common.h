// 16-bit pointwise GEMM: what pointwise_h.hip (direct / walking / LDS-tiled forms) and pointwise_hs.hip / pointwise_hq.hip
// (activation-stationary and all-couts k-streaming forms) share - operand types, the MFMA wrapper, the activation row
// object with its gathers, the accumulator-octet epilogue.
#pragma once
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

// ---- branch-free 16-bit store (pointwise_hs.hip / pointwise_hq.hip) -------------------------------------------------
// A store under a per-lane branch cannot be COUNTED by the compiler's s_waitcnt insertion: any later wait for an older
// load (the weight block prefetched before this epilogue) then becomes vmcnt(0) and stalls until the stores are
// acknowledged too.  Through a buffer descriptor the store is unconditional: dead lanes (rows beyond M, couts beyond N,
// the three non-owner lanes of a pooling window) pass an offset beyond num_records and the hardware drops them.
typedef __amdgpu_buffer_rsrc_t pwh_rsrc;
__device__ __forceinline__ pwh_rsrc pwh_make_rsrc(const void* base, unsigned bytes) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ pwh_rsrc pwh_out_rsrc(const PwArgs& a, unsigned bytes) { return pwh_make_rsrc(a.out, bytes); }

// ReLU6 of eight ROUNDED 16-bit values, two per dword, as packed 16-bit INTEGER max / min: bf16 and f16 are sign-magnitude,
// so as signed integers every negative value (and -0) is below +0 = 0x0000 and the non-negative ones order like the numbers;
// rounding is monotonic and 0 and 6 are exact in both types, so clamp(round(x)) == round(clamp(x)) bit for bit.  8 packed
// instructions per octet instead of 16 float32 ones - in kernels bound by instruction issue.
template <class T>
__device__ __forceinline__ pwh_u4 pwh_relu6_packed(pwh_u4 v) {
    // (whole-vector form: element-wise bit casts of v[d] in an unrolled loop were folded to four copies of element 0 by hipcc 7.2)
    typedef short s8 __attribute__((ext_vector_type(8)));
    constexpr short six = yr_elem<T>::dtype == YR_BF16 ? (short)0x40c0 : (short)0x4600;
    const s8 zero = {0, 0, 0, 0, 0, 0, 0, 0}, hi = {six, six, six, six, six, six, six, six};
    s8 t = __builtin_bit_cast(s8, v);
    t = __builtin_elementwise_max(t, zero);
    t = __builtin_elementwise_min(t, hi);
    return __builtin_bit_cast(pwh_u4, t);
}

// pwh_finish_oct with the same arithmetic (bit for bit) for 16-bit outputs.  float32 outputs (logit tensors, hoisted
// partial sums: partial octets, dword-aligned rows) keep the branching stores of pwh_finish_oct, and a run-time switch
// between the two would make every wait after the join conservative again: such ops do not take these kernels.
template <class T>
__device__ __forceinline__ void pwh_finish_oct_b(const PwArgs& a, pwh_rsrc out, const f32x4& lo, const f32x4& hi, const float (&sc)[8],
                                                 const float (&sh)[8], int m, int n, int li) {
    const int cnt = a.N - n;
    const int nld = cnt > 0 ? n : 0;
    const int ml = m < a.M ? m : a.M - 1;
    float v[8], q[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = lo[r]; v[4 + r] = hi[r]; }
    if (a.pre) {  // uniform
        pwh_load8_f32(a.pre, pw_pre_row(a, ml), a.pre_ld, nld, a.N, q);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += q[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = __builtin_fmaf(v[r], sc[r], sh[r]);
    if (a.act == YR_ACT_NONE || a.act == YR_ACT_RELU6) {  // uniform: ONE branch per octet; the clamp form of both (same values bit for bit)
        const float lo = a.act == YR_ACT_RELU6 ? 0.0f : -__builtin_inff(), hi = a.act == YR_ACT_RELU6 ? 6.0f : __builtin_inff();
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fminf(fmaxf(v[r], lo), hi);
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = yr_apply_act_t<T>(v[r], a.act);
    }
    if (a.res) {  // uniform
        const pwh_f8 t = pwh_widen<T>(*reinterpret_cast<const pwh_u4*>(reinterpret_cast<const T*>(a.res) + (size_t)ml * a.res_ld + nld));
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += t[r];
    }
    int orow = m;
    bool keep = m < a.M && cnt > 0;
    if (a.pool) {  // uniform
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r] = fmaxf(v[r], __shfl_xor(v[r], 1));
            v[r] = fmaxf(v[r], __shfl_xor(v[r], 2));
        }
        keep = keep && (li & 3) == 0;
        orow = m >> 2;
    }
    pwh_f8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = v[r];
    const unsigned off = keep ? ((unsigned)orow * (unsigned)a.out_ld + (unsigned)n) * 2u : 0x7f000000u;
    __builtin_amdgcn_raw_buffer_store_b128(pwh_narrow<T>(o), out, off, 0, 0);
}

// rows of the output that a launch may address through a 32-bit byte offset
__host__ inline bool pwh_out_fits_rsrc(const PwArgs& a) {
    const unsigned long long rows = a.pool ? (unsigned long long)(a.M >> 2) : (unsigned long long)a.M;
    return !a.out_f32 && rows * (unsigned long long)a.out_ld * 2ull < 0x7f000000ull;   // (dead lanes sit at 0x7f000000 and beyond)
}
__host__ inline unsigned pwh_out_bytes(const PwArgs& a) {
    const unsigned long long rows = a.pool ? (unsigned long long)(a.M >> 2) : (unsigned long long)a.M;
    return (unsigned)(rows * (unsigned long long)a.out_ld * 2ull);
}
