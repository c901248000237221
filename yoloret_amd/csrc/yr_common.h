// Shared host/device helpers for libyoloret_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/yoloret_hip.h"

void yr_set_error(const char* fmt, ...);
// records the symbol the last launcher dispatched to (read back by yr_forward_profile)
void yr_note_kernel(const char* name);

#define YR_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            yr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return YR_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define YR_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            yr_set_error(__VA_ARGS__);        \
            return YR_ERR_ARG;                \
        }                                     \
    } while (0)

#define YR_LAUNCH_CHECK() YR_CHECK_HIP(hipGetLastError())

// per-kind launchers (defined in the .hip files); `op` holds device pointers.
int yr_launch_stem(const yr_op& op, int batch, hipStream_t s);
int yr_launch_pointwise(const yr_op& op, int batch, hipStream_t s);
int yr_launch_depthwise(const yr_op& op, int batch, hipStream_t s);
int yr_launch_depthwise_lds(int dtype, int k, const void* in, const float* w, const float* scale, const float* shift, void* out, int B, int H, int W,
                             int C8, int ld_in, int ld_w, int ld_out, int pad_t, int pad_l, int act, float* part, int ld_part, int part_rows,
                             hipStream_t s);
int yr_launch_depthwise_walk(int dtype, int k, const void* in, const float* w, const float* scale, const float* shift, void* out, int B, int H, int W,
                              int C8, int ld_in, int ld_w, int ld_out, int pad_t, int pad_l, int act, float* part, int ld_part, int part_rows,
                              hipStream_t s);
int yr_launch_se_mean(const yr_op& op, int batch, hipStream_t s);
int yr_launch_se_fc(const yr_op& op, int batch, hipStream_t s);
int yr_launch_wsum(const yr_op& op, int batch, hipStream_t s);
int yr_launch_gather(const yr_op& op, int batch, hipStream_t s);
int yr_launch_stemblock(const yr_op& op, int batch, hipStream_t s);
int yr_launch_mblane(const yr_op& op, int batch, hipStream_t s);
int yr_launch_mbh(const yr_op& op, int batch, hipStream_t s);
bool yr_mbh_prefers_chained(const yr_op& op);
bool yr_stemxp_takes(const yr_op& op);                                  // mbxr_h.hip: the entry with projection, register-chained
int yr_launch_stemxp(const yr_op& op, int batch, hipStream_t s);   // mbxr_h.hip: a plain launch of this MBH / MBX op runs the register-chained form
int yr_launch_mbr(const yr_op& op, int batch, hipStream_t s);
int yr_launch_mbe(const yr_op& op, int batch, hipStream_t s);
int yr_launch_mbk(const yr_op& op, int batch, hipStream_t s);   // YR_OP_MBR with k bit 6: the weight-streaming form (mbk.hip)
int yr_launch_head(const yr_op& op, int batch, hipStream_t s);      // headblock.hip
int yr_launch_absmax(const float* p, long long rows, int c, int ld, unsigned* out, hipStream_t s);   // elementwise.hip
int yr_launch_head_walk(const yr_op& op, int batch, hipStream_t s); // headwalk.hip (YR_OP_HEAD with k bit 6)
int yr_launch_head_walk_h(const yr_op& op, int batch, hipStream_t s); // headwalk_h.hip (... of a 16-bit plan)
int yr_launch_head_stream(const yr_op& op, int batch, hipStream_t s); // headstream.hip (YR_OP_HEAD with k bits 5 and 6: the weight-streaming form)
int yr_pointwise_num_cfgs(int dtype);
extern "C" int yr_pwt_chunks(int kp);   // chunks of 32 channels the pixel-stationary pointwise form (pointwise_stream.hip) runs a k space with; 0: not taken

static inline int yr_round_up(int v, int m) { return (v + m - 1) / m * m; }
// channels per 16 bytes of a tensor of this yr_dtype: the granule of `ld` and of the pointwise k-space
static inline int yr_vec_of(int dtype) { return dtype == YR_F32 ? 4 : 8; }
static inline const char* yr_dtype_name(int dtype) { return dtype == YR_BF16 ? "bf16" : dtype == YR_F16 ? "f16" : "f32"; }
static inline bool yr_dtype_ok(int dtype) { return dtype == YR_F32 || dtype == YR_BF16 || dtype == YR_F16; }

#ifdef __HIPCC__
// ---------------------------------------------------------------- device side
// The library is built with -ffp-contract=off: every fused multiply-add below is an
// explicit fmaf, so results do not depend on the optimiser's contraction choices.

// Pinned float32 exp: the same steps as oracle/csrc/yr_oracle.c:yro_expf (Cody-Waite
// reduction + degree-5 polynomial; each step one IEEE op) => bit-identical to the C oracle.
__device__ __forceinline__ float yr_expf(float x) {
    x = fminf(x, 88.0f);
    x = fmaxf(x, -87.0f);
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, r * r, r) + 1.0f;
    int e = ((int)n + 127) << 23;
    return y * __int_as_float(e);
}
__device__ __forceinline__ float yr_sigmoid(float x) { return 1.0f / (1.0f + yr_expf(-x)); }

__device__ __forceinline__ float yr_apply_act(float v, int act) {
    switch (act) {
        case YR_ACT_RELU6: return fminf(fmaxf(v, 0.0f), 6.0f);
        case YR_ACT_SWISH: return v * yr_sigmoid(v);
        case YR_ACT_SIGMOID: return yr_sigmoid(v);
        case YR_ACT_LEAKY: return v >= 0.0f ? v : v * 0.1f;
        default: return v;
    }
}
__device__ __forceinline__ float4 yr_apply_act4(float4 v, int act) {
    return make_float4(yr_apply_act(v.x, act), yr_apply_act(v.y, act), yr_apply_act(v.z, act), yr_apply_act(v.w, act));
}
// Activation of a value about to be STORED as T.  16-bit T: the store keeps 8 or 11 significant bits, so swish takes the
// hardware exp2 and reciprocal (about 1 ulp of float32 each, 5 instructions) instead of the pinned expf and the IEEE
// division above (28 instructions - as much as the whole 3x3x3 stem convolution per output element).  float32 T: the
// pinned path, bit-identical to the C oracle.
template <class T>
__device__ __forceinline__ float yr_apply_act_t(float v, int act) {
    if constexpr (sizeof(T) == 2) {
        if (act == YR_ACT_SWISH) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
    }
    return yr_apply_act(v, act);
}
template <class T>
__device__ __forceinline__ float4 yr_apply_act4_t(float4 v, int act) {
    return make_float4(yr_apply_act_t<T>(v.x, act), yr_apply_act_t<T>(v.y, act), yr_apply_act_t<T>(v.z, act), yr_apply_act_t<T>(v.w, act));
}
__device__ __forceinline__ float4 yr_max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// ---- element types.  float32, or the 16-bit STORAGE types of a reduced-precision plan: kernels widen to float32 on
// load (exact) and round to nearest-even on store (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32); arithmetic stays float32
// everywhere except the pointwise GEMM's MFMA operands.
typedef __bf16 yr_bf16;
typedef _Float16 yr_f16;
template <class T> struct yr_elem;
template <> struct yr_elem<float> { static constexpr int dtype = YR_F32, vec = 4; };
template <> struct yr_elem<yr_bf16> { static constexpr int dtype = YR_BF16, vec = 8; };
template <> struct yr_elem<yr_f16> { static constexpr int dtype = YR_F16, vec = 8; };

// four consecutive channels at p (float32: 16 bytes, 16-byte aligned; 16-bit: 8 bytes, 8-byte aligned) <-> float4
template <class T>
__device__ __forceinline__ float4 yr_ld4(const T* p) {
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const float4*>(p);
    } else {
        typedef T t4 __attribute__((ext_vector_type(4)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = __builtin_convertvector(*reinterpret_cast<const t4*>(p), f4);
        return make_float4(v[0], v[1], v[2], v[3]);
    }
}
template <class T>
__device__ __forceinline__ void yr_st4(T* p, float4 v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = v;
    } else {
        typedef T t4 __attribute__((ext_vector_type(4)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<t4*>(p) = __builtin_convertvector((f4){v.x, v.y, v.z, v.w}, t4);
    }
}
// two consecutive channels (float32: 8 bytes; 16-bit: 4 bytes)
template <class T>
__device__ __forceinline__ void yr_st2(T* p, float x, float y) {
    typedef T t2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<t2*>(p) = __builtin_convertvector((f2){x, y}, t2);
}
template <class T> __device__ __forceinline__ float yr_ld1(const T* p) { return (float)*p; }
template <class T> __device__ __forceinline__ void yr_st1(T* p, float v) { *p = (T)v; }

// Dispatch a launcher template over the op's element type: YR_BY_DTYPE(op.dtype, launch_x, args...) calls
// launch_x<float|yr_bf16|yr_f16>(args...).
#define YR_BY_DTYPE(dt, fn, ...)                                             \
    ((dt) == YR_F32 ? fn<float>(__VA_ARGS__)                                 \
     : (dt) == YR_BF16 ? fn<yr_bf16>(__VA_ARGS__)                            \
     : (dt) == YR_F16 ? fn<yr_f16>(__VA_ARGS__)                              \
                      : (yr_set_error("unknown dtype %d", (int)(dt)), (int)YR_ERR_ARG))

// The float16-plane cut of the SPLIT forms (mbr_common.h: x = h + 2^-11 m, h = f16(x), m = f16((x - h) 2^11)) for a pair of values,
// 4 VALU operations: v_cvt_pk_f16_f32 (h), v_pk_mul_f32 (2^11 x), and v_fma_mixlo_f16 / v_fma_mixhi_f16, which read a float16 half
// of h as a multiplicand, form h * -2^11 + 2^11 x in float32 (exact: x - h has at most 13 significant bits) and round it to the
// low / high half of m - the conversions back to float32, the subtraction and the second packed conversion of the plain code
// (6 operations; hipcc 7.2 does not select the mix instructions by itself) folded into two.  Bit-identical to the plain code
// (tools/cut_probe.hip: 4 M pairs incl. zeros, denormals, 65504, values below 2^-24).
__device__ __forceinline__ void yr_cut2(float x0, float x1, unsigned& h, unsigned& m) {
    typedef float cut_f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 cut_h2 __attribute__((ext_vector_type(2)));
    const cut_f2 x = (cut_f2){x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, cut_h2));
    const cut_f2 xs = x * 2048.0f;
    const float c = -2048.0f;
    // (two statements: m may then take the register of xs[0], which the first instruction reads before it writes.  s_nop 1: an MFMA
    //  that reads a register as its A / B operand must be two wait states behind the VALU instruction that wrote it; hipcc's hazard
    //  recogniser places those waits itself but does not look into asm - without them the MFMAs behind mbs_split8 read a stale m
    //  plane (logit errors of 5e-4: the h plane alone).  A VALU or LDS consumer needs no wait: tools/cut_probe.hip.)
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(m) : "v"(h), "s"(c), "v"(xs[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(m) : "v"(h), "s"(c), "v"(xs[1]));
}

// XCD-aware block order (cdna_hip_programming.md T1): hardware block b runs on XCD b % 8, each XCD has
// its own L2.  Map hardware ids so that every XCD walks a CONTIGUOUS range of logical blocks; logical
// neighbours (adjacent rows of a depthwise map, the cout tiles of one pixel tile) then share an L2.
// Bijective for any grid size; only speed depends on the placement assumption.
__device__ __forceinline__ unsigned yr_xcd_swizzle(unsigned bid, unsigned nb) {
    const unsigned q = nb >> 3, r = nb & 7u;
    const unsigned x = bid & 7u, i = bid >> 3;
    return x * q + (x < r ? x : r) + i;
}

// Device view of one concatenated source segment (see yr_src).  `ptr` is type-erased: the kernel template knows the
// element type (all k-space sources of an op share it).
struct DSrc {
    const float* ptr;
    int h, w;      // source spatial dims
    int c;         // channels taken
    int ld;        // channel stride
    int xform;     // yr_xform
    int kbase;     // start of this segment in the consumer's padded k-space (multiple of 4)
};

struct DSrcSet {
    DSrc s[YR_MAX_SRC];
    int n;
    int kp;        // total padded k (sum of round_up(c,4))
};

// Loads channels [kk, kk+4) of segment `s` for consumer pixel (b,y,x); lanes beyond
// the segment's channel count are returned as 0 (never multiplied through).
template <class T = float>
__device__ __forceinline__ float4 yr_load_src_quad(const DSrc& s, int b, int y, int x, int kk) {
    float4 v;
    const T* sp = reinterpret_cast<const T*>(s.ptr);
    if (s.xform == YR_X_IDENTITY) {
        v = yr_ld4<T>(sp + ((size_t)(b * s.h + y) * s.w + x) * s.ld + kk);
    } else if (s.xform == YR_X_UP2) {
        v = yr_ld4<T>(sp + ((size_t)(b * s.h + (y >> 1)) * s.w + (x >> 1)) * s.ld + kk);
    } else {
        const int p = (s.xform == YR_X_MAXPOOL2) ? 2 : 4;
        const T* base = sp + ((size_t)(b * s.h + y * p) * s.w + x * p) * s.ld + kk;
        v = yr_ld4<T>(base);
        for (int dy = 0; dy < p; ++dy)
            for (int dx = 0; dx < p; ++dx)
                v = yr_max4(v, yr_ld4<T>(base + ((size_t)dy * s.w + dx) * s.ld));
    }
    const int rem = s.c - kk;
    if (rem < 4) {
        if (rem < 4) v.w = 0.0f;
        if (rem < 3) v.z = 0.0f;
        if (rem < 2) v.y = 0.0f;
    }
    return v;
}

// k in the padded k-space -> (segment, offset); returns zero quad when k >= kp.
template <class T = float>
__device__ __forceinline__ float4 yr_load_cat_quad(const DSrcSet& S, int b, int y, int x, int k) {
    if (k >= S.kp) return make_float4(0.f, 0.f, 0.f, 0.f);
    int si = 0;
#pragma unroll
    for (int i = 1; i < YR_MAX_SRC; ++i)
        if (i < S.n && k >= S.s[i].kbase) si = i;
    // select fields without dynamic indexing of the struct array (keeps it in SGPRs)
    DSrc s = S.s[0];
    if (si == 1) s = S.s[1];
    if (si == 2) s = S.s[2];
    if (si == 3) s = S.s[3];
    return yr_load_src_quad<T>(s, b, y, x, k - s.kbase);
}
#endif

// host: build a DSrcSet from a yr_op (validates alignment); out_h/out_w are the consumer dims.
#ifdef __HIPCC__
static inline int yr_make_srcset(const yr_op& op, DSrcSet* S) {
    if (op.nsrc < 1 || op.nsrc > YR_MAX_SRC) { yr_set_error("nsrc=%d out of range", op.nsrc); return YR_ERR_ARG; }
    if (!yr_dtype_ok(op.dtype)) { yr_set_error("unknown dtype %d", op.dtype); return YR_ERR_ARG; }
    const int V = yr_vec_of(op.dtype);   // channels per 16 bytes: the granule of ld and of the k-space segments
    int k = 0;
    S->n = op.nsrc;
    for (int i = 0; i < YR_MAX_SRC; ++i) {
        DSrc& d = S->s[i];
        if (i >= op.nsrc) { d = S->s[0]; d.kbase = 1 << 30; continue; }
        const yr_src& s = op.src[i];
        int eh = s.h, ew = s.w;
        if (s.xform == YR_X_UP2) { eh *= 2; ew *= 2; }
        else if (s.xform == YR_X_MAXPOOL2) { eh /= 2; ew /= 2; }
        else if (s.xform == YR_X_MAXPOOL4) { eh /= 4; ew /= 4; }
        else if (s.xform == YR_X_DW3) {  // POINTWISE only (checked there): read through a 3x3 depthwise conv, TF 'SAME'
            const int st = op.se_reduced & 0xff;
            if (op.nsrc != 1 || (st != 1 && st != 2)) { yr_set_error("dw3 source: must be the only source, stride 1 or 2"); return YR_ERR_ARG; }
            eh = (eh + st - 1) / st; ew = (ew + st - 1) / st;
        }
        else if (s.xform != YR_X_IDENTITY) { yr_set_error("bad xform %d", s.xform); return YR_ERR_ARG; }
        if (eh != op.h || ew != op.w) { yr_set_error("src %d dims %dx%d (xform %d) do not give %dx%d", i, s.h, s.w, s.xform, op.h, op.w); return YR_ERR_ARG; }
        if (s.dtype != op.dtype) { yr_set_error("src %d has dtype %d, the op works in dtype %d", i, s.dtype, op.dtype); return YR_ERR_ARG; }
        if (s.ld % V != 0 || s.ld < yr_round_up(s.c, V)) { yr_set_error("src %d: ld=%d must be a multiple of %d and >= round_up(c=%d,%d)", i, s.ld, V, s.c, V); return YR_ERR_ARG; }
        if (((uintptr_t)s.ptr) % 16 != 0 || s.ptr == nullptr) { yr_set_error("src %d pointer null or not 16-byte aligned", i); return YR_ERR_ARG; }
        d.ptr = (const float*)s.ptr; d.h = s.h; d.w = s.w; d.c = s.c; d.ld = s.ld; d.xform = s.xform; d.kbase = k;
        k += yr_round_up(s.c, V);
    }
    S->kp = k;
    return YR_OK;
}
#endif
