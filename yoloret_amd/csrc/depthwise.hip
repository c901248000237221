// Depthwise k x k convolution (k in {3,5}, stride in {1,2}, TF 'SAME' padding) with fused
// BatchNorm scale/shift + ReLU6/Swish, NHWC, 16 bytes of channels per lane (4 float32 / 8 bf16 or f16).  Replaces TF's
// DepthwiseConv2dNative + FusedBatchNormV3 + Relu6/Swish used by MobileNetV2's *_depthwise
// layers [3P], reference code/yolo3/model.py:20-24 (RFCR 5x5) and
// code/yolo3/efficientnet.py:501-510 (MBConv).
//
// Each lane produces a YT x XT patch of output pixels for one channel quad: an input row is
// loaded once per XT outputs and reused by up to K/S vertically adjacent outputs from registers.
// Consecutive lanes take consecutive channel quads (16 B apart) => fully coalesced loads and
// stores.  Workgroups are walked in XCD-contiguous order so that vertically adjacent output rows
// (which share K-S input rows) are processed on the same XCD and hit its L2 (the round-robin
// default was measured to fetch each input row ~3x from HBM - profiles/r01_pmc_fetch.txt).
#include "yr_common.h"
#include "se_tail.h"
#include <cstdlib>

// T: element type of the input and output maps (float32, or bf16 / f16 storage: widened on load, rounded to nearest
// on store; the accumulation, BatchNorm and activation are float32 either way).  Weights, scale, shift: float32.
template <class T>
struct DwArgs {
    const T* in;         // [B][Hi][Wi][ld_in]
    const float* w;      // [K*K][ld_w] (channel-fastest)
    const float* scale;  // [C]
    const float* shift;  // [C]
    T* out;              // [B][Ho][Wo][ld_out]
    int B, Hi, Wi, Ho, Wo, C4;  // C4 = ceil(C / channels per lane)
    int ld_in, ld_w, ld_out;
    int pad_t, pad_l;
    int act;
    int xstrips, ystrips;  // ceil(Wo / XT), ceil(Ho / YT)
    long long total;       // B*ystrips*xstrips*C4  (SE form: per image)
    unsigned nblocks;
    // SE form (squeeze-excite squeeze fused in, efficientnet.py:417): grid = (workgroups per image, B); every workgroup
    // also writes the sums of ITS outputs per channel to part[b][blockIdx.x][..] (float32, fixed order: deterministic);
    // the SE_FC op adds the rows up instead of re-reading the whole map.  A workgroup's 256 lanes are `cw` channel vectors
    // x 256/cw pixel strips (dw_se_geometry): cw = C4 (all channels of a strip in one workgroup) or 32 - `ncb` workgroups then
    // share a group of strips and each writes its own channels of that group's row.
    float* part;
    int ld_part, cw, ncb;
    SeTail se;           // ABI 7: the workgroup that completes an image's rows also runs the SE block's FC pair (se.sums == nullptr: no)
};

__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y),
                       __builtin_fmaf(a.z, b.z, c.z), __builtin_fmaf(a.w, b.w, c.w));
}

// One 16-byte load = Q quads of channels (float32: 1, 16-bit: 2 - the same bytes in flight per lane, half the lanes:
// with 8-byte loads the 16-bit form took exactly as long as float32, the kernel is bound by loads issued, not bytes).
template <class T, int Q>
__device__ __forceinline__ void dw_load(const T* p, float4 (&v)[Q]) {
    if constexpr (Q == 1) {
        v[0] = *reinterpret_cast<const float4*>(p);
    } else {
        typedef T t8 __attribute__((ext_vector_type(8)));
        typedef float f8 __attribute__((ext_vector_type(8)));
        const f8 x = __builtin_convertvector(*reinterpret_cast<const t8*>(p), f8);
        v[0] = make_float4(x[0], x[1], x[2], x[3]);
        v[1] = make_float4(x[4], x[5], x[6], x[7]);
    }
}
template <class T, int Q>
__device__ __forceinline__ void dw_store(T* p, const float4 (&v)[Q]) {
    if constexpr (Q == 1) {
        *reinterpret_cast<float4*>(p) = v[0];
    } else {
        typedef T t8 __attribute__((ext_vector_type(8)));
        typedef float f8 __attribute__((ext_vector_type(8)));
        *reinterpret_cast<t8*>(p) = __builtin_convertvector((f8){v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w}, t8);
    }
}

template <int K, int S, int XT, int YT, class T, bool SE = false>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs<T> a) {
    constexpr int Q = yr_elem<T>::vec / 4;   // channel quads per lane
    __shared__ float4 red[SE ? YR_SE_TAIL_LDS / 4 : 1];   // the partial sums' meeting place (256 * Q quads), then the SE tail's scratch
    __shared__ unsigned se_flag;
    // SE: grid (workgroups per image, B), walked in XCD-contiguous order like the plain form (adjacent strips share input rows)
    const unsigned lin = SE ? yr_xcd_swizzle(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y) : 0u;
    const unsigned se_b = SE ? lin / gridDim.x : 0u, se_blk = SE ? lin - se_b * gridDim.x : 0u;
    long long gid = SE ? 0ll : (long long)yr_xcd_swizzle(blockIdx.x, a.nblocks) * 256 + threadIdx.x;
    bool live = gid < a.total;
    unsigned se_row = 0u;                    // SE: the row of `part` this workgroup's strips belong to
    if constexpr (SE) {
        const int cvl = (int)threadIdx.x % a.cw, sl = (int)threadIdx.x / a.cw;
        const int spb = 256 / a.cw;              // whole strips per workgroup (cw need not divide 256: the last lanes idle)
        const unsigned cb = se_blk % (unsigned)a.ncb;
        se_row = se_blk / (unsigned)a.ncb;
        const long long strips = (long long)a.ystrips * a.xstrips;
        long long strip = (long long)se_row * spb + (sl < spb ? sl : spb - 1);
        int cv = (int)cb * a.cw + cvl;
        live = sl < spb && strip < strips && cv < a.C4;
        if (strip >= strips) strip = strips - 1;   // idle lanes compute a valid location (no store, no sum) and take part
        if (cv >= a.C4) cv = a.C4 - 1;             // in the reduction
        gid = strip * a.C4 + cv;
    }
    if (!SE && !live) return;
    const int cq = (int)(gid % a.C4);
    long long t = gid / a.C4;
    const int xs = (int)(t % a.xstrips);
    t /= a.xstrips;
    const int ys = (int)(t % a.ystrips);
    const int b = SE ? (int)se_b : (int)(t / a.ystrips);
    const int c = cq * 4 * Q;
    const int x0 = xs * XT, y0 = ys * YT;
    constexpr int COLS = (XT - 1) * S + K;
    constexpr int ROWS = (YT - 1) * S + K;

    float4 acc[YT][XT][Q];
#pragma unroll
    for (int j = 0; j < YT; ++j)
#pragma unroll
        for (int i = 0; i < XT; ++i)
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[j][i][q] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int iy0 = y0 * S - a.pad_t;
    const int ix0 = x0 * S - a.pad_l;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int iy = iy0 + r;
        if (iy < 0 || iy >= a.Hi) continue;  // zero padding row
        const T* rowp = a.in + ((size_t)(b * a.Hi + iy) * a.Wi) * a.ld_in + c;
        float4 col[COLS][Q];
#pragma unroll
        for (int j = 0; j < COLS; ++j) {
            const int ix = ix0 + j;
            if (ix >= 0 && ix < a.Wi) {
                dw_load<T, Q>(rowp + (size_t)ix * a.ld_in, col[j]);
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q) col[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // input row r feeds output row j through kernel row ky = r - j*S
#pragma unroll
        for (int j = 0; j < YT; ++j) {
            const int ky = r - j * S;
            if (ky < 0 || ky >= K) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float4 wv = *reinterpret_cast<const float4*>(a.w + (size_t)(ky * K + kx) * a.ld_w + c + 4 * q);
#pragma unroll
                    for (int i = 0; i < XT; ++i) acc[j][i][q] = fma4(col[i * S + kx][q], wv, acc[j][i][q]);
                }
            }
        }
    }
    float4 sc[Q], sh[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        sc[q] = *reinterpret_cast<const float4*>(a.scale + c + 4 * q);
        sh[q] = *reinterpret_cast<const float4*>(a.shift + c + 4 * q);
    }
    float4 psum[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) psum[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < YT; ++j) {
        if (y0 + j >= a.Ho) continue;
        T* op = a.out + ((size_t)(b * a.Ho + y0 + j) * a.Wo + x0) * a.ld_out + c;
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            if (x0 + i < a.Wo) {
                float4 v[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) v[q] = yr_apply_act4_t<T>(fma4(acc[j][i][q], sc[q], sh[q]), a.act);
                if (!SE || live) dw_store<T, Q>(op + (size_t)i * a.ld_out, v);
                if constexpr (SE) {
                    if constexpr (Q == 2) {   // the mean is that of the STORED (rounded) values
                        typedef T t8 __attribute__((ext_vector_type(8)));
                        typedef float f8 __attribute__((ext_vector_type(8)));
                        const f8 rv = __builtin_convertvector(__builtin_convertvector((f8){v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w}, t8), f8);
                        v[0] = make_float4(rv[0], rv[1], rv[2], rv[3]);
                        v[1] = make_float4(rv[4], rv[5], rv[6], rv[7]);
                    }
                    if (live) {
#pragma unroll
                        for (int q = 0; q < Q; ++q) { psum[q].x += v[q].x; psum[q].y += v[q].y; psum[q].z += v[q].z; psum[q].w += v[q].w; }
                    }
                }
            }
        }
    }
    if constexpr (SE) {
        // lanes l, l + cw, l + 2*cw ... of the workgroup hold the same channel vector: add them in index order
#pragma unroll
        for (int q = 0; q < Q; ++q) red[threadIdx.x * Q + q] = psum[q];
        __syncthreads();
        const int cv = (int)(se_blk % (unsigned)a.ncb) * a.cw + (int)threadIdx.x;
        if ((int)threadIdx.x < a.cw && cv < a.C4) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                float4 s = red[threadIdx.x * Q + q];
                for (int l = threadIdx.x + a.cw; l < 256; l += a.cw) {
                    const float4 v = red[l * Q + q];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                yr_st_sums4(a.part + ((size_t)b * (gridDim.x / a.ncb) + se_row) * a.ld_part + (cv * Q + q) * 4, a.se.sums != nullptr, s.x, s.y, s.z, s.w);
            }
        }
        yr_se_tail_arrive<256>(a.se, b, 1u, &se_flag, reinterpret_cast<float*>(red));   // (red: YR_SE_TAIL_LDS floats)
    }
}

// SE form geometry (the compiler sizes the partial-sum buffer with the same formula, compiler.dw_se_geometry): `strips`
// pixel strips per image, C4 channel vectors -> cw lanes of a workgroup span channels, ncb workgroups cover C4, rows =
// groups of 256/cw strips = rows of the partial-sum buffer.
static inline void dw_se_geometry(long long strips, int c4, int* cw, int* ncb, int* rows) {
    // (a) all channel vectors of a strip in one workgroup (cw = C4, 256/C4 whole strips, the remaining lanes idle) or
    // (b) 32-vector blocks, several workgroups per strip group: whichever keeps more of the 256 lanes busy
    int w = 32;
    if (c4 <= 256) {
        const int busy_a = (256 / c4) * c4 * ((c4 + 31) / 32) * 32;   // compared as busy_a / 256 vs c4 / (ncb_b * 32)
        if (busy_a >= c4 * 256) w = c4;
    }
    *cw = w;
    *ncb = (c4 + w - 1) / w;
    *rows = (int)((strips + 256 / w - 1) / (256 / w));
}

template <int K, int S, int XT, class T>
static int launch_dw_se(DwArgs<T> a, int expect_rows, hipStream_t s) {
    a.xstrips = (a.Wo + XT - 1) / XT;
    a.ystrips = a.Ho;
    a.total = (long long)a.ystrips * a.xstrips * a.C4;
    int rows = 0;
    dw_se_geometry((long long)a.ystrips * a.xstrips, a.C4, &a.cw, &a.ncb, &rows);
    YR_REQUIRE(rows == expect_rows, "depthwise: the SE partial-sum buffer must hold %d rows per image (has %d)", rows, expect_rows);
    const long long blocks = (long long)rows * a.ncb;
    YR_REQUIRE(blocks * a.B < (1ll << 31), "depthwise: grid too large");
    a.se.arrivals = (unsigned)blocks;   // every workgroup of an image arrives once
    a.nblocks = (unsigned)blocks;
    static char nm[48];
    static const int nm_len = snprintf(nm, sizeof(nm), "dw_kernel<%d,%d,%d,1,%s,1>", K, S, XT, yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    yr_note_kernel(nm);
    hipLaunchKernelGGL((dw_kernel<K, S, XT, 1, T, true>), dim3((unsigned)blocks, a.B), dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <int K, int S, int XT, int YT, class T>
static int launch_dw(DwArgs<T> a, hipStream_t s) {
    a.xstrips = (a.Wo + XT - 1) / XT;
    a.ystrips = (a.Ho + YT - 1) / YT;
    a.total = (long long)a.B * a.ystrips * a.xstrips * a.C4;
    const long long blocks = (a.total + 255) / 256;
    YR_REQUIRE(blocks < (1ll << 31), "depthwise: grid too large");
    a.nblocks = (unsigned)blocks;
    static char nm[40];
    static const int nm_len = snprintf(nm, sizeof(nm), "dw_kernel<%d,%d,%d,%d,%s,0>", K, S, XT, YT, yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    yr_note_kernel(nm);
    hipLaunchKernelGGL((dw_kernel<K, S, XT, YT, T>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T>
static int launch_depthwise_t(const yr_op& op, int batch, hipStream_t s) {
    constexpr int V = yr_elem<T>::vec;
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY, "depthwise: needs one identity source");
    YR_REQUIRE(op.src[0].dtype == op.dtype && op.out_dtype == op.dtype, "depthwise: source and output must have the op's dtype");
    const yr_src& in = op.src[0];
    YR_REQUIRE(op.k == 3 || op.k == 5, "depthwise: kernel size %d unsupported", op.k);
    YR_REQUIRE(op.stride == 1 || op.stride == 2, "depthwise: stride %d unsupported", op.stride);
    YR_REQUIRE(in.c == op.cout && op.cin == op.cout, "depthwise: channel mismatch");
    YR_REQUIRE(in.ld % V == 0 && op.out_ld % V == 0 && in.ld >= yr_round_up(in.c, V) && op.out_ld >= yr_round_up(in.c, V),
               "depthwise: ld must be a multiple of %d and cover round_up(c,%d)", V, V);
    YR_REQUIRE(in.ptr && op.out && op.wgt && op.scale && op.shift, "depthwise: null pointer");
    YR_REQUIRE(((uintptr_t)in.ptr | (uintptr_t)op.out | (uintptr_t)op.wgt | (uintptr_t)op.scale | (uintptr_t)op.shift) % 16 == 0,
               "depthwise: pointers must be 16-byte aligned");
    DwArgs<T> a;
    a.in = (const T*)in.ptr; a.w = op.wgt; a.scale = op.scale; a.shift = op.shift; a.out = (T*)op.out;
    a.B = batch; a.Hi = in.h; a.Wi = in.w;
    a.Ho = (in.h + op.stride - 1) / op.stride;
    a.Wo = (in.w + op.stride - 1) / op.stride;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "depthwise: output dims %dx%d != SAME(%dx%d / %d)", op.h, op.w, in.h, in.w, op.stride);
    a.C4 = (in.c + V - 1) / V;                    // lanes per pixel: V channels (16 bytes) each
    a.ld_in = in.ld; a.ld_w = yr_round_up(in.c, V); a.ld_out = op.out_ld;   // weights / scale / shift: [..][round_up(c, V)]
    // TF 'SAME': pad_total = max((out-1)*s + k - in, 0); before = total/2 (extra goes bottom/right)
    const int pth = (a.Ho - 1) * op.stride + op.k - in.h, ptw = (a.Wo - 1) * op.stride + op.k - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2;
    a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.act = op.act;
    a.part = nullptr; a.ld_part = 0; a.cw = 1; a.ncb = 1;
    {
        const int rc = yr_make_se_tail(op, op.se_reduced, &a.se);
        if (rc) return rc;
    }
    if (op.gate) YR_REQUIRE(op.gate_ld % 4 == 0 && op.gate_ld >= yr_round_up(in.c, V) && ((uintptr_t)op.gate % 16) == 0, "depthwise: bad SE partial-sum buffer");
    // 16-bit 5x5 / 3x3 stride 1 on maps with at least one 64-channel chunk: the LDS-tiled form (depthwise_lds.hip, bit-identical
    // maps; its SE rows are its tiles - compiler.se_partials_from_depthwise sizes the buffer for whichever form
    // YOLORET_DW_LDS selects, 0 keeps this kernel for A/B runs)
    if constexpr (sizeof(T) == 2) {
#ifdef YR_DW_EXPERIMENT
        const bool lds_form = !(getenv("YOLORET_DW_LDS") && atoi(getenv("YOLORET_DW_LDS")) == 0);   // re-read per launch
#else
        static const bool lds_form = !(getenv("YOLORET_DW_LDS") && atoi(getenv("YOLORET_DW_LDS")) == 0);
#endif
        // (round 4: the walking form - whole column blocks, rows walked once - unless YOLORET_DW_WALK=0 keeps the tile walk for A/B runs)
        static const int walk_form = getenv("YOLORET_DW_WALK") ? atoi(getenv("YOLORET_DW_WALK")) : 1;
        // for the 5 x 5 maps (tools/dwq_probe.sh: 26 x 26 x 672 @128 83 -> 66 us, SE form 106 -> 72; the others -1 .. -10 %); the 3 x 3
        // maps are close to the HBM rate in either form and keep the tile walk (52 x 52 x 128 @128: 32 us against 35)
        if (lds_form && walk_form && op.k == 5 && op.stride == 1 && in.c >= 64)
            return yr_launch_depthwise_walk(op.dtype, op.k, in.ptr, op.wgt, op.scale, op.shift, op.out, batch, in.h, in.w, a.C4, a.ld_in, a.ld_w, a.ld_out,
                                            a.pad_t, a.pad_l, a.act, const_cast<float*>(op.gate), op.gate_ld, op.se_reduced, s);
        if (lds_form && (op.k == 5 || op.k == 3) && op.stride == 1 && in.c >= 64)
            return yr_launch_depthwise_lds(op.dtype, op.k, in.ptr, op.wgt, op.scale, op.shift, op.out, batch, in.h, in.w, a.C4, a.ld_in, a.ld_w, a.ld_out,
                                            a.pad_t, a.pad_l, a.act, const_cast<float*>(op.gate), op.gate_ld, op.se_reduced, s);
    }
    if (op.gate) {   // SE form: `gate` is an OUTPUT here - float32 [B][workgroups per image][gate_ld] channel sums
        a.part = const_cast<float*>(op.gate); a.ld_part = op.gate_ld;
        const int rows = op.se_reduced;     // rows per image the buffer was sized for
        if (op.k == 3 && op.stride == 1) return launch_dw_se<3, 1, 4, T>(a, rows, s);
        if (op.k == 3 && op.stride == 2) return launch_dw_se<3, 2, 2, T>(a, rows, s);
        if (op.k == 5 && op.stride == 1) return launch_dw_se<5, 1, 4, T>(a, rows, s);
        return launch_dw_se<5, 2, 2, T>(a, rows, s);
    }
#ifdef YR_DW_EXPERIMENT   // tools/dw5_probe.py: patch shapes of the 5x5 stride-1 form (XT*10 + YT in YR_DW_FORCE)
    if (op.k == 5 && op.stride == 1) {
        const char* e = getenv("YR_DW_FORCE");
        const int f = e ? atoi(e) : 0;
        if (f == 42) return launch_dw<5, 1, 4, 2, T>(a, s);
        if (f == 22) return launch_dw<5, 1, 2, 2, T>(a, s);
        if (f == 21) return launch_dw<5, 1, 2, 1, T>(a, s);
        if (f == 81) return launch_dw<5, 1, 8, 1, T>(a, s);
        if (f == 44) return launch_dw<5, 1, 4, 4, T>(a, s);
        if (f == 24) return launch_dw<5, 1, 2, 4, T>(a, s);
    }
#endif
    // small maps (13x13, 26x26) keep 1-row patches so the grid still fills the chip
    const bool big = (long long)batch * a.Ho * a.Wo * a.C4 >= (1ll << 21);
    // stride 1: 4x1 patches beat 4x2, 2x1, 2x2, 8x1 and 13x1 on every 13/26/52 map of the flagship (tools/dw_probe.py):
    // the neighbouring rows' re-reads hit L2, and the shorter patch keeps more loads in flight per CU
    if (op.k == 3 && op.stride == 1) return launch_dw<3, 1, 4, 1, T>(a, s);
    if (op.k == 3 && op.stride == 2) return big ? launch_dw<3, 2, 2, 2, T>(a, s) : launch_dw<3, 2, 2, 1, T>(a, s);
    if (op.k == 5 && op.stride == 1) return launch_dw<5, 1, 4, 1, T>(a, s);
    return launch_dw<5, 2, 2, 1, T>(a, s);
}

int yr_launch_depthwise(const yr_op& op, int batch, hipStream_t s) { return YR_BY_DTYPE(op.dtype, launch_depthwise_t, op, batch, s); }
