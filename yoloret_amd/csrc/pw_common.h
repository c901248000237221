// Shared by pointwise.hip (direct kernel + dispatcher) and pointwise_lds.hip (LDS-staged kernel).
#pragma once
#include <type_traits>
#include <utility>

#include "yr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at a dword-aligned address
typedef f32x4 __attribute__((address_space(1))) pw_gf4;                // 4 floats in GLOBAL memory (global_load, not flat_load)
__device__ __forceinline__ float4 pw_ldg(const pw_gf4* q, size_t i = 0) {
    const f32x4 v = q[i];  // a native vector: HIP's float4 class would bind a generic reference and load flat again
    return make_float4(v[0], v[1], v[2], v[3]);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): loops over register arrays whose indices are
// constants for the FRONT END.  A `#pragma unroll` loop is unrolled late; until then the arrays are stack objects
// with a dynamic index, and what the optimiser does to them in between (address selects, conditional accesses)
// can leave them in scratch memory for good.
template <class F, int... I>
__device__ __forceinline__ void pw_unroll_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void pw_unroll(F f) {
    pw_unroll_impl(f, std::make_integer_sequence<int, N>{});
}

struct PwArgs {
    DSrcSet S;
    const float* wt;      // [N][kp]
    const float* scale;   // [N] or null
    const float* shift;   // [N] or null
    const float* res;     // residual [M][res_ld] or null
    const float* gate;    // SE gate [B][gate_ld] or null
    const float* pre;     // YR_X_UP2_ADD source [B][H/2][W/2][pre_ld] added to the accumulator before BN, or null
    float* out;           // [M][out_ld]
    int M, H, W, N;
    int out_ld, res_ld, gate_ld, pre_ld;
    int act;
    int pool;             // 1: the output is MaxPooling2D(2)(conv output); H, W, M stay the CONV's (pre-pool) dims
    // YR_X_DW3 source (null dw_w: none): S.s[0] is the depthwise INPUT [B][Hi][Wi][ld]; H, W are its OUTPUT dims
    const float* dw_w;      // [9][S.kp] depthwise weights, tap-major (ky, kx)
    const float* dw_scale;  // [S.kp] folded BN of the depthwise stage
    const float* dw_shift;  // [S.kp]
    int dw_stride, dw_act, dw_pad_t, dw_pad_l;
    // 16-bit ops (pointwise_h.hip; the float32 kernels ignore it): sources, weights and residual are bf16 / f16 behind the
    // type-erased pointers above; out_f32 = 1: `out` is float32 all the same (logit outputs, hoisted partial sums)
    int out_f32;
    // the pixel-stationary form's SECOND output (se_reduced bit 19 of a POINTWISE op; null: none): another 1x1 conv of the same (gated) source -
    // its couts follow the first one's in the weight planes and in scale / shift (each padded to a multiple of 16)
    float* out2;
    int out2_ld, N2, act2, pool2;
};

// GEMM row -> conv pixel.  Plain: identity.  Pooled output: rows are walked in 2x2-quad-major order, so the four
// pixels of a pooling window sit in four ADJACENT lanes of one MFMA tile and the window maximum is two DPP steps.
__device__ __forceinline__ int pw_pixel_of_row(const PwArgs& a, int m) {
    if (!a.pool) return m;
    const int q = m >> 2, sub = m & 3;
    const int wq = a.W >> 1, hwq = (a.H >> 1) * wq;
    const int b = q / hwq, r = q - b * hwq;
    const int yq = r / wq, xq = r - yq * wq;
    return (b * a.H + 2 * yq + (sub >> 1)) * a.W + 2 * xq + (sub & 1);
}

// MaxPooling2D(2) of the finished (BN + activation) values across the four lanes of a pooling window; lane
// (row & 3) == 0 then owns the result, which goes to pooled pixel row >> 2.
__device__ __forceinline__ void pw_pool4(float (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = fmaxf(v[r], __shfl_xor(v[r], 1));
        v[r] = fmaxf(v[r], __shfl_xor(v[r], 2));
    }
}

// Epilogue loads are UNCONDITIONAL (clamped addresses, results of dead lanes ignored): a load under a per-lane branch
// is followed by its own s_waitcnt, and a tile's epilogue then pays one L2 round trip per element group instead of
// one per tile.  Quad n..n+3 of row `row` ([.][ld] floats, N real channels, n a multiple of 4 and < N): one 16-byte
// load when the pitch allows it (`vec`: ld % 4 == 0, so the quad lies inside the row), else four clamped dwords.
__device__ __forceinline__ void pw_load_quad(const float* base, size_t row, int ld, int n, int N, bool vec, float (&q)[4]) {
    const float* rp = base + row * ld;
    if (vec) {
        const float4 t = *reinterpret_cast<const float4*>(rp + n);
        q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = rp[n + r < N ? n + r : N - 1];
    }
}

// Row of the YR_X_UP2_ADD source that output pixel m (a conv pixel, m < M) takes its pre-BatchNorm addend from.
__device__ __forceinline__ size_t pw_pre_row(const PwArgs& a, int m) {
    const int hw = a.H * a.W;
    const int b = m / hw, rem = m - b * hw;
    const int y = rem / a.W, x = rem - y * a.W;
    return (size_t)(b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1);
}

// The epilogue of one accumulator quad: output row m (a GEMM row; it may lie beyond M), couts n..n+3 (n may lie
// beyond N).  (Pre-BN addend,) BN scale/shift, activation, (residual,) (2x2 max,) store.  Branches are uniform or
// guard stores only.  li: the lane's row within its 16-row MFMA tile (the pooling window = lanes li&~3 .. +3).
__device__ __forceinline__ void pw_finish_quad(const PwArgs& a, const f32x4& acc, const f32x4& sc, const f32x4& sh, int m,
                                               int n, int li, bool vec_out, bool vec_res, bool vec_pre) {
    const int cnt = a.N - n;           // real couts in this quad (<= 0: none)
    const int nld = cnt > 0 ? n : 0;   // the column dead quads load from
    const int ml = m < a.M ? m : a.M - 1;
    float v[4], q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r];
    if (a.pre) {  // uniform: the low-resolution share of a hoisted concat conv joins the accumulator before BN
        pw_load_quad(a.pre, pw_pre_row(a, ml), a.pre_ld, nld, a.N, vec_pre, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += q[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = yr_apply_act(__builtin_fmaf(v[r], sc[r], sh[r]), a.act);
    if (a.res) {  // uniform
        pw_load_quad(a.res, (size_t)ml, a.res_ld, nld, a.N, vec_res, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += q[r];
    }
    int orow = m;
    bool keep = m < a.M && cnt > 0;
    if (a.pool) {  // uniform: MaxPooling2D(2) across the 4 adjacent lanes of a window
        pw_pool4(v);
        keep = keep && (li & 3) == 0;
        orow = m >> 2;
    }
    if (!keep) return;
    float* op = a.out + (size_t)orow * a.out_ld + n;
    if (cnt >= 4) {
        // dense rows (the 75-wide logit outputs): still ONE 16-byte store per lane, only 4-byte aligned
        // (global_store_dwordx4 takes dword-aligned addresses); four dword stores cost the y convs 40 % of their time
        if (vec_out) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        else *reinterpret_cast<f32x4u*>(op) = (f32x4u){v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (r < cnt) op[r] = v[r];
    }
}

// One output row (pixel) of the activation operand: where its channels come from.
// MODE 0: generic gather (upsample / maxpool / concat sources), 1: one identity source, 2: identity + SE gate
// (MODE 3, the depthwise-folded source, is PwDwRow below).
// The per-source row pointers are four named members, not an array: after unrolling, LLVM folds a select between
// loads of two array slots into ONE load with a selected address, which pins the array in scratch memory and puts
// a scratch round trip in front of every activation load.
static_assert(YR_MAX_SRC == 4, "PwRow spells out four sources");
template <int MODE>
struct PwRow {
    const float* arow;               // MODE != 0: the pixel's contiguous row
    const float* grow;               // MODE == 2: SE gate row of the pixel's image
    const float *s0, *s1, *s2, *s3;  // MODE == 0: per-source row pointer (xform folded in)
    bool valid;

    static __device__ __forceinline__ const float* source_row(const DSrc& d, int b, int y, int x) {
        int sy = y, sx = x;
        if (d.xform == YR_X_UP2) { sy = y >> 1; sx = x >> 1; }
        else if (d.xform == YR_X_MAXPOOL2) { sy = y * 2; sx = x * 2; }
        else if (d.xform == YR_X_MAXPOOL4) { sy = y * 4; sx = x * 4; }
        return d.ptr + ((size_t)(b * d.h + sy) * d.w + sx) * d.ld;
    }

    __device__ __forceinline__ void init(const PwArgs& a, int m) {
        valid = m < a.M;
        const int mm = pw_pixel_of_row(a, valid ? m : 0);
        const int hw = a.H * a.W;
        const int b = mm / hw;
        grow = MODE == 2 ? a.gate + (size_t)b * a.gate_ld : nullptr;
        arow = s0 = s1 = s2 = s3 = nullptr;
        if (MODE != 0) {
            arow = a.S.s[0].ptr + (size_t)mm * a.S.s[0].ld;
        } else {
            const int rem = mm - b * hw;
            const int y = rem / a.W, x = rem - y * a.W;
            s0 = source_row(a.S.s[0], b, y, x) - a.S.s[0].kbase;  // pre-offset: channel k of the conv is s_i[k]
            s1 = source_row(a.S.s[1], b, y, x) - a.S.s[1].kbase;
            s2 = source_row(a.S.s[2], b, y, x) - a.S.s[2].kbase;
            s3 = source_row(a.S.s[3], b, y, x) - a.S.s[3].kbase;
        }
    }

    // the same for conv pixel (b, y, x) given directly (headblock.hip: the rows of a workgroup are a region of one image)
    __device__ __forceinline__ void init_at(const PwArgs& a, int b, int y, int x, bool valid_) {
        valid = valid_;
        grow = MODE == 2 ? a.gate + (size_t)b * a.gate_ld : nullptr;
        arow = s0 = s1 = s2 = s3 = nullptr;
        if (MODE != 0) {
            arow = a.S.s[0].ptr + ((size_t)(b * a.H + y) * a.W + x) * a.S.s[0].ld;
        } else {
            s0 = source_row(a.S.s[0], b, y, x) - a.S.s[0].kbase;
            s1 = source_row(a.S.s[1], b, y, x) - a.S.s[1].kbase;
            s2 = source_row(a.S.s[2], b, y, x) - a.S.s[2].kbase;
            s3 = source_row(a.S.s[3], b, y, x) - a.S.s[3].kbase;
        }
    }

    // Issue the loads of the quad at k (raw k may lie beyond kp: clamped).  v: raw channels, gt: gate quad
    // (MODE 2), cv: how many of the quad's channels are real (<= 0: none).  Nothing here reads a loaded
    // register and the main load is unconditional: rows beyond M read row 0 (their outputs are never stored), the
    // k tail re-reads the last quad (zeroed through cv).  When EVERY load sits under a branch the compiler's
    // s_waitcnt insertion must assume the path that issued none and emits vmcnt(0), which serialises every
    // prefetch behind the newest load; only the extra taps of pooled sources stay conditional here.
    // POOLS = false promises that no source is pooled (the caller checked): the code is then straight-line.
    template <bool POOLS = true>
    __device__ __forceinline__ void issue(const PwArgs& a, int kraw, int kp, float4& v, float4& gt, int& cv) const {
        const int k = kraw < kp ? kraw : kp - 4;
        int cvalid;
        if (MODE != 0) {
            v = *reinterpret_cast<const float4*>(arow + k);
            cvalid = a.S.s[0].c - k;
            if (MODE == 2) gt = *reinterpret_cast<const float4*>(grow + k);
        } else {
            // segment of this quad (kbase ascends; unused segments have a huge kbase).  Everything that is picked
            // per lane is first made an opaque VALUE (empty asm / readfirstlane): a select between two loads (of
            // stack slots or of kernel-argument fields) is otherwise folded into ONE load from a selected address -
            // scratch traffic for the pointers, a per-lane global load of the argument block for the fields.
            const bool g1 = k >= a.S.s[1].kbase, g2 = k >= a.S.s[2].kbase, g3 = k >= a.S.s[3].kbase;
            const float *p0 = s0, *p1 = s1, *p2 = s2, *p3 = s3;  // pre-offset by -kbase (init)
            asm("" : "+v"(p0));
            asm("" : "+v"(p1));
            asm("" : "+v"(p2));
            asm("" : "+v"(p3));
            const float* rp = g3 ? p3 : g2 ? p2 : g1 ? p1 : p0;
#define PW_PICK(name, e0, e1, e2, e3)                                                                   \
    const int name##0 = __builtin_amdgcn_readfirstlane(e0), name##1 = __builtin_amdgcn_readfirstlane(e1), \
              name##2 = __builtin_amdgcn_readfirstlane(e2), name##3 = __builtin_amdgcn_readfirstlane(e3); \
    const int name = g3 ? name##3 : g2 ? name##2 : g1 ? name##1 : name##0;
            PW_PICK(kend, a.S.s[0].kbase + a.S.s[0].c, a.S.s[1].kbase + a.S.s[1].c, a.S.s[2].kbase + a.S.s[2].c,
                    a.S.s[3].kbase + a.S.s[3].c)
            cvalid = kend - k;
            // the selected pointer is spelled as a GLOBAL one: a generic (flat) load would also count on lgkmcnt, so
            // every wait for an LDS read would wait for the prefetch as well
            const pw_gf4* q = (const pw_gf4*)(rp + k);
            v = pw_ldg(q);
            int xf = 0, sw = 0, sld = 0;
            if (POOLS) {
                PW_PICK(xfp, a.S.s[0].xform, a.S.s[1].xform, a.S.s[2].xform, a.S.s[3].xform)
                PW_PICK(swp, a.S.s[0].w, a.S.s[1].w, a.S.s[2].w, a.S.s[3].w)
                PW_PICK(sldp, a.S.s[0].ld, a.S.s[1].ld, a.S.s[2].ld, a.S.s[3].ld)
                xf = xfp; sw = swp; sld = sldp;
            }
#undef PW_PICK
            if (!POOLS) {
            } else if (xf == YR_X_MAXPOOL2) {  // pooled sources are reduced here (the only path that waits at issue):
                // the three other taps are issued together - ONE round trip, not one per tap (sld % 4 == 0)
                const float4 v1 = pw_ldg(q, sld >> 2);
                const float4 v2 = pw_ldg(q, ((size_t)sw * sld) >> 2);
                const float4 v3 = pw_ldg(q, (((size_t)sw + 1) * sld) >> 2);
                v = yr_max4(yr_max4(v, v1), yr_max4(v2, v3));
            } else if (xf == YR_X_MAXPOOL4) {
                for (int dy = 0; dy < 4; ++dy)
                    for (int dx = 0; dx < 4; ++dx)
                        v = yr_max4(v, pw_ldg(q, (((size_t)dy * sw + dx) * sld) >> 2));
            }
        }
        cv = (valid && kraw < kp) ? cvalid : 0;
    }
};

// MODE 3: the row's channels are the output of a 3x3 depthwise conv (TF 'SAME', stride 1/2) + BN + activation over
// S.s[0] (YR_X_DW3).  issue() fetches the nine taps of a channel quad (clamped addresses, unconditional); finish()
// reproduces dw_kernel's arithmetic exactly - taps in (ky, kx) order as one fma chain from 0, padding taps contribute
// fma(0, w, acc) = acc, then fma(acc, scale, shift) and the activation - so the operand is bit-identical to what a
// DEPTHWISE op would have written.  Weights, scale and shift are read from LDS (`dwl`: [9][kp] | [kp] | [kp]).
struct PwDwRow {
    const float* img;   // the image's depthwise input
    int off[9];         // element offset of each (clamped) tap pixel
    unsigned mask;      // bit t: tap t lies inside the input
    bool valid;

    __device__ __forceinline__ void init(const PwArgs& a, int m) {
        valid = m < a.M;
        const int mm = valid ? m : 0;
        const DSrc& d = a.S.s[0];
        const int hw = a.H * a.W;
        const int b = mm / hw, rem = mm - b * hw;
        const int y = rem / a.W, x = rem - y * a.W;
        img = d.ptr + (size_t)b * d.h * d.w * d.ld;
        const int iy0 = y * a.dw_stride - a.dw_pad_t, ix0 = x * a.dw_stride - a.dw_pad_l;
        mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = iy0 + t / 3, ix = ix0 + t % 3;
            const bool in = iy >= 0 && iy < d.h && ix >= 0 && ix < d.w;
            mask |= (in ? 1u : 0u) << t;
            const int cy = iy < 0 ? 0 : (iy >= d.h ? d.h - 1 : iy), cx = ix < 0 ? 0 : (ix >= d.w ? d.w - 1 : ix);
            off[t] = (cy * d.w + cx) * d.ld;
        }
    }

    __device__ __forceinline__ void issue(const PwArgs& a, int kraw, int kp, float4 (&v)[9], int& cv) const {
        const int k = kraw < kp ? kraw : kp - 4;
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = *reinterpret_cast<const float4*>(img + off[t] + k);
        cv = (valid && kraw < kp) ? a.S.s[0].c - k : 0;
    }

    // kq4: the quad's k offset inside the LDS weight block (= the clamped k of issue())
    __device__ __forceinline__ float4 finish(const PwArgs& a, const float4 (&v)[9], int cv, const float* dwl, int kp, int k) const {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 w = *reinterpret_cast<const float4*>(dwl + t * kp + k);
            const bool in = (mask >> t) & 1u;
            const float4 e = make_float4(in ? v[t].x : 0.f, in ? v[t].y : 0.f, in ? v[t].z : 0.f, in ? v[t].w : 0.f);
            acc = make_float4(__builtin_fmaf(e.x, w.x, acc.x), __builtin_fmaf(e.y, w.y, acc.y),
                              __builtin_fmaf(e.z, w.z, acc.z), __builtin_fmaf(e.w, w.w, acc.w));
        }
        const float4 sc = *reinterpret_cast<const float4*>(dwl + 9 * kp + k);
        const float4 sh = *reinterpret_cast<const float4*>(dwl + 10 * kp + k);
        float4 r = yr_apply_act4(make_float4(__builtin_fmaf(acc.x, sc.x, sh.x), __builtin_fmaf(acc.y, sc.y, sh.y),
                                             __builtin_fmaf(acc.z, sc.z, sh.z), __builtin_fmaf(acc.w, sc.w, sh.w)), a.dw_act);
        r.x = cv > 0 ? r.x : 0.f;
        r.y = cv > 1 ? r.y : 0.f;
        r.z = cv > 2 ? r.z : 0.f;
        r.w = cv > 3 ? r.w : 0.f;
        return r;
    }
};

// the fetched quad with pad lanes zeroed (the source's pad lanes and the gate's may hold anything) and gated
template <int MODE>
__device__ __forceinline__ float4 pw_finish(float4 v, const float4& gt, int cvalid) {
    v.x = cvalid > 0 ? (MODE == 2 ? v.x * gt.x : v.x) : 0.f;
    v.y = cvalid > 1 ? (MODE == 2 ? v.y * gt.y : v.y) : 0.f;
    v.z = cvalid > 2 ? (MODE == 2 ? v.z * gt.z : v.z) : 0.f;
    v.w = cvalid > 3 ? (MODE == 2 ? v.w * gt.w : v.w) : 0.f;
    return v;
}

// LDS-staged kernel, tile shape index 0..13: (BM x BN) = 256x16, 128x32, 128x48, 128x64, 128x80, 128x96, 128x128,
// 64x16, 64x32, 64x48, 64x64, 64x80, 64x96, 64x128
int yr_pw_launch_lds(int shape, const PwArgs& a, hipStream_t s);
// the same tile shapes on the 16-bit matrix pipe with float32-grade operands (pointwise_split.hip: two float16 planes per operand)
int yr_pw_launch_split(int shape, const PwArgs& a, hipStream_t s);
// its k-split form for the passes of a few images (a workgroup = one 16 x 16 tile, the four waves split the k range; se_reduced bit 17)
int yr_pw_launch_ksplit(const PwArgs& a, hipStream_t s);
// its pixel-stationary form (pointwise_stream.hip; se_reduced bit 18: a.wt holds the weights' float16 planes, compiler.head_pack over
// yr_pwt_chunks(kp) chunks of 32 channels); yr_pwt_chunks: 0 = the form does not take a k space this deep
int yr_pw_launch_stream(const PwArgs& a, hipStream_t s);
extern "C" int yr_pwt_chunks(int kp);
// 16-bit kernel (pointwise_h.hip): cfg = tile shape index 0..yr_pwh_num_cfgs()-1, or -1 for its heuristic
int yr_pw_launch_h(int dtype, int cfg, const PwArgs& a, hipStream_t s);
int yr_pwh_num_cfgs();
// its k-split form for the passes of one or two images (se_reduced bit 17; a workgroup = one 16 x 32 tile, four waves split the k range)
int yr_pwh_launch_ksplit(int dtype, const PwArgs& a, hipStream_t s);
// its activation-stationary (pointwise_hs.hip) and all-couts k-streaming (pointwise_hq.hip) forms, variant 0..3 each;
// -1: the form does not take this op (nothing launched, no error set)
// what a form-specific launcher returns when the form does not take the op (the caller falls back to another form):
// positive, so it cannot be mistaken for a yr_status error (YR_ERR_ARG is -1), which is propagated as it is
#define YR_NOT_TAKEN 1
int yr_pwhs_launch(int dtype, int variant, const PwArgs& a, hipStream_t s);
int yr_pwhq_launch(int dtype, int variant, const PwArgs& a, hipStream_t s);
