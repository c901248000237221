// Small memory-bound ops around the convolutions:
//   SE_MEAN  - tf.reduce_mean over H,W           (reference code/yolo3/efficientnet.py:391-403,417)
//   SE_FC    - 1x1+bias -> Swish -> 1x1+bias -> sigmoid on the pooled vector (efficientnet.py:419-434);
//              the Multiply (:435) is folded into the consumer pointwise kernel's loads.
//   WSUM     - WeightedSum of 4 gathered tensors  (reference code/yolo3/model.py:117-137,157)
//   GATHER   - materialised UpSampling2D / MaxPooling2D / Concatenate (model.py:139-144,164-166);
//              the fast path folds these into consumer loads, this kernel exists for
//              unfused use and for testing the gather machinery in isolation.
#include "yr_common.h"
#include "se_tail.h"

// ------------------------------------------------------------------ SE mean
template <class T>
struct MeanArgs {
    const T* in;      // [B][HW][ld]   (T: float32 or 16-bit storage; the sum and the mean are float32)
    float* out;       // [B][ld_out]
    int HW, C4, ld, ld_out, C;
    float inv;        // unused (division keeps reduce_mean's sum/count form)
};

// grid (ceil(C4/16), B); block 1024 = 64 pixel lanes x 16 channel quads.  Fixed summation
// order => run-to-run deterministic.
#define SE_MEAN_PL 64
template <class T>
__global__ __launch_bounds__(1024) void se_mean_kernel(MeanArgs<T> a) {
    __shared__ float4 part[SE_MEAN_PL][16];
    const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int cq = blockIdx.x * 16 + q;
    const int b = blockIdx.y;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cq < a.C4) {
        const T* p = a.in + (size_t)b * a.HW * a.ld + cq * 4;
#pragma unroll 4
        for (int i = pl; i < a.HW; i += SE_MEAN_PL) {
            const float4 v = yr_ld4<T>(p + (size_t)i * a.ld);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[pl][q] = s;
    __syncthreads();
    // two-level fixed-order combine: 64 -> 8 -> 1
    if (pl < 8 && cq < a.C4) {
        float4 t = part[pl * 8][q];
        for (int i = 1; i < 8; ++i) {
            const float4 v = part[pl * 8 + i][q];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        part[pl * 8][q] = t;
    }
    __syncthreads();
    if (pl == 0 && cq < a.C4) {
        float4 t = part[0][q];
        for (int i = 1; i < 8; ++i) {
            const float4 v = part[i * 8][q];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        const float n = (float)a.HW;
        float4 m = make_float4(t.x / n, t.y / n, t.z / n, t.w / n);
        const int c = cq * 4;
        if (c + 1 >= a.C) m.y = 0.f;
        if (c + 2 >= a.C) m.z = 0.f;
        if (c + 3 >= a.C) m.w = 0.f;
        *reinterpret_cast<float4*>(a.out + (size_t)b * a.ld_out + c) = m;
    }
}

template <class T>
static int launch_se_mean_t(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY, "se_mean: needs one identity source");
    const yr_src& in = op.src[0];
    YR_REQUIRE(in.dtype == op.dtype && op.out_dtype == YR_F32, "se_mean: the source has the op's dtype, the pooled vector is float32");
    YR_REQUIRE(in.ptr && op.out && in.ld % yr_elem<T>::vec == 0 && op.out_ld % 4 == 0 && op.out_ld >= yr_round_up(in.c, 4),
               "se_mean: bad pointers / strides");
    MeanArgs<T> a;
    a.in = (const T*)in.ptr; a.out = (float*)op.out; a.HW = in.h * in.w; a.C = in.c; a.C4 = (in.c + 3) / 4;
    a.ld = in.ld; a.ld_out = op.out_ld; a.inv = 0.f;
    static char nm[32];
    static const int nm_len = snprintf(nm, sizeof(nm), "se_mean_kernel<%s>", yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    yr_note_kernel(nm);
    hipLaunchKernelGGL(se_mean_kernel<T>, dim3((a.C4 + 15) / 16, batch), dim3(1024), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}
int yr_launch_se_mean(const yr_op& op, int batch, hipStream_t s) { return YR_BY_DTYPE(op.dtype, launch_se_mean_t, op, batch, s); }

// ------------------------------------------------------------------ SE FCs
template <class T>
struct FcArgs {
    const T* map;       // HW > 1: the full map [B][HW][ld_map] (element type T) whose spatial mean is the FC input (SE_MEAN merged in)
    int HW, ld_map;
    int pool;           // 0: `mean` is the pooled vector already; 1: `map` is a map to pool; 2: `mean` holds float32 rows of partial
                        // channel sums [B][HW][ld_mean] written by the SE form of a depthwise / fused-block kernel
    float count;        // what the pooled sums are divided by: HW, or the true pixel count when the rows are partial sums
    const float* mean;  // [B][ld_mean]
    SeFc fc;            // W1 [ldc][R4] | b1 [R4] | W2 [R][ldc] | b2 [ldc]
    float* gate;        // [B][ld_gate]
    int C, R, ldc, ld_mean, ld_gate;
};

// One workgroup per image; dynamic LDS = yr_se_fc_floats(C, R, 1024) floats (+ 1024 float4 when a map is pooled here).  The FC pair
// itself is se_tail.h's yr_se_fc_pair (round 5: a thread per quad of outputs and segment of inputs, 16-byte loads - the launch is
// three dependent round trips on 64 workgroups, so what counts is how few batches of loads each stage needs).
#define SE_FC_THREADS 1024
template <class T>
__global__ __launch_bounds__(SE_FC_THREADS) void se_fc_kernel(FcArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* mean = sm;                                       // [ldc]
    float* scratch = sm + a.ldc;                            // [R4 + 4 * SE_FC_THREADS]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.pool == 2) {
        yr_se_mean_rows<SE_FC_THREADS, false>(a.mean + (size_t)b * a.HW * a.ld_mean, a.HW, a.ld_mean, a.C, a.ldc, a.count, mean, scratch, tid);
    } else if (a.pool == 1) {
        // tf.reduce_mean over H,W first (the SE_MEAN op merged into this launch).  All channel quads at once: C4P = next power of two
        // >= C4 quads x (1024 / C4P) pixel lanes, then one fixed-order combine over the pixel lanes (deterministic).
        float4* red = reinterpret_cast<float4*>(sm + ((yr_se_fc_floats(a.C, a.R, SE_FC_THREADS) + 3) & ~(size_t)3));  // [PL][C4P], 16-byte aligned
        const int C4 = (a.C + 3) >> 2;
        int c4p = 1;
        while (c4p < C4 && c4p < SE_FC_THREADS) c4p <<= 1;
        const int PL = SE_FC_THREADS / c4p;
        for (int q0 = 0; q0 < C4; q0 += c4p) {   // one pass unless C > 4096
            const int cq = q0 + (tid & (c4p - 1)), pl = tid / c4p;
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cq < C4) {
                const T* p = a.map + (size_t)b * a.HW * a.ld_map + cq * 4;
#pragma unroll 8
                for (int i = pl; i < a.HW; i += PL) {
                    const float4 v = yr_ld4<T>(p + (size_t)i * a.ld_map);
                    s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
                }
            }
            red[tid] = s4;
            __syncthreads();
            if (pl == 0 && cq < C4) {
                float4 t = red[tid];
                for (int i = 1; i < PL; ++i) {
                    const float4 v = red[i * c4p + tid];
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
                const float n = a.count;
                const int c = cq * 4;
                mean[c] = t.x / n;
                if (c + 1 < a.ldc) mean[c + 1] = c + 1 < a.C ? t.y / n : 0.f;
                if (c + 2 < a.ldc) mean[c + 2] = c + 2 < a.C ? t.z / n : 0.f;
                if (c + 3 < a.ldc) mean[c + 3] = c + 3 < a.C ? t.w / n : 0.f;
            }
            __syncthreads();
        }
    } else {
        for (int c = tid; c < a.ldc; c += SE_FC_THREADS) mean[c] = (c < a.C) ? a.mean[(size_t)b * a.ld_mean + c] : 0.f;
        __syncthreads();
    }
    yr_se_fc_pair<SE_FC_THREADS>(a.fc, mean, scratch, a.gate + (size_t)b * a.ld_gate, tid);
    for (int c = a.ldc + tid; c < a.ld_gate; c += SE_FC_THREADS) a.gate[(size_t)b * a.ld_gate + c] = 0.f;   // pad columns of a wider gate row (16-bit plans: ld = round_up(C, 8))
}

template <class T>
static int launch_se_fc_t(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 1, "se_fc: needs one source (the pooled vector)");
    const yr_src& in = op.src[0];
    YR_REQUIRE(in.ptr && op.out && op.wgt && op.wgt2 && op.b1 && op.b2, "se_fc: null pointer");
    YR_REQUIRE(op.se_reduced >= 1 && in.c == op.cout, "se_fc: bad widths");
    YR_REQUIRE(op.out_dtype == YR_F32, "se_fc: the gate is float32");
    FcArgs<T> a;
    a.map = (const T*)in.ptr; a.HW = in.h * in.w; a.ld_map = in.ld;   // h*w > 1: the pooled vector is computed here (SE_MEAN merged)
    a.count = op.k > 0 ? (float)op.k : (float)a.HW;                   // k: pixel count when the rows are partial sums, not pixels
    a.pool = op.k > 0 ? 2 : a.HW > 1 ? 1 : 0;
    YR_REQUIRE((a.HW == 1 || op.k > 0) ? in.dtype == YR_F32 : in.dtype == op.dtype, "se_fc: a pooled vector / partial sums are float32, a map to pool has the op's dtype");
    YR_REQUIRE((a.HW == 1 && op.k <= 0) || (in.ld % yr_elem<T>::vec == 0 && ((uintptr_t)in.ptr % 16) == 0), "se_fc: the map to pool must be 16-byte addressable per pixel");
    a.mean = (const float*)in.ptr; a.gate = (float*)op.out;
    a.C = in.c; a.R = op.se_reduced; a.ldc = yr_round_up(in.c, 4); a.ld_mean = in.ld; a.ld_gate = op.out_ld;
    a.fc.w1 = op.wgt; a.fc.b1 = op.b1; a.fc.w2 = op.wgt2; a.fc.b2 = op.b2; a.fc.C = a.C; a.fc.R = a.R; a.fc.ldc = a.ldc;
    YR_REQUIRE(((uintptr_t)op.wgt | (uintptr_t)op.wgt2 | (uintptr_t)op.b1 | (uintptr_t)op.b2 | (uintptr_t)op.out) % 16 == 0, "se_fc: parameters and gate must be 16-byte aligned");
    YR_REQUIRE(op.out_ld >= a.ldc && op.out_ld % 4 == 0, "se_fc: gate ld too small");
    const size_t lds = ((yr_se_fc_floats(a.C, a.R, SE_FC_THREADS) + 3) & ~(size_t)3) * sizeof(float) + (a.pool == 1 ? SE_FC_THREADS * sizeof(float4) : 0);
    YR_REQUIRE(lds <= 64 * 1024, "se_fc: widths too large for LDS");
    static char nm[32];
    static const int nm_len = snprintf(nm, sizeof(nm), "se_fc_kernel<%s>", yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    yr_note_kernel(nm);
    hipLaunchKernelGGL(se_fc_kernel<T>, dim3(batch), dim3(SE_FC_THREADS), lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}
int yr_launch_se_fc(const yr_op& op, int batch, hipStream_t s) {
    // a pooled vector (h*w == 1) and per-workgroup partial sums (k > 0) are float32 whatever the plan's dtype
    if (op.nsrc == 1 && (op.src[0].h * op.src[0].w == 1 || op.k > 0)) return launch_se_fc_t<float>(op, batch, s);
    return YR_BY_DTYPE(op.dtype, launch_se_fc_t, op, batch, s);
}

// ------------------------------------------------------------------ WeightedSum
template <class T>
struct WsumArgs {
    DSrc s[4];
    const float* alpha;  // [4]
    T* out;
    int H, W, C4, ld_out;
    long long total;
};

template <class T>
__global__ __launch_bounds__(256) void wsum_kernel(WsumArgs<T> a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.total) return;
    const int cq = (int)(gid % a.C4);
    long long t = gid / a.C4;
    const int x = (int)(t % a.W);
    t /= a.W;
    const int y = (int)(t % a.H);
    const int b = (int)(t / a.H);
    const float a0 = a.alpha[0], a1 = a.alpha[1], a2 = a.alpha[2], a3 = a.alpha[3];
    const float4 v0 = yr_load_src_quad<T>(a.s[0], b, y, x, cq * 4);
    const float4 v1 = yr_load_src_quad<T>(a.s[1], b, y, x, cq * 4);
    const float4 v2 = yr_load_src_quad<T>(a.s[2], b, y, x, cq * 4);
    const float4 v3 = yr_load_src_quad<T>(a.s[3], b, y, x, cq * 4);
    // reference order (model.py:134): a0*m0 + a1*m1 + a2*m2 + a3*m3, left to right, no contraction
    float4 r;
    r.x = a0 * v0.x + a1 * v1.x + a2 * v2.x + a3 * v3.x;
    r.y = a0 * v0.y + a1 * v1.y + a2 * v2.y + a3 * v3.y;
    r.z = a0 * v0.z + a1 * v1.z + a2 * v2.z + a3 * v3.z;
    r.w = a0 * v0.w + a1 * v1.w + a2 * v2.w + a3 * v3.w;
    yr_st4<T>(a.out + ((size_t)(b * a.H + y) * a.W + x) * a.ld_out + cq * 4, r);
}

template <class T>
static int launch_wsum_t(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 4, "wsum: needs exactly 4 sources");
    YR_REQUIRE(op.out_dtype == op.dtype, "wsum: the output has the op's dtype");
    DSrcSet S;
    for (int i = 0; i < op.nsrc && i < YR_MAX_SRC; ++i) YR_REQUIRE(op.src[i].xform != YR_X_DW3, "dw3 sources are a POINTWISE feature");
    int rc = yr_make_srcset(op, &S);
    if (rc) return rc;
    for (int i = 0; i < 4; ++i) YR_REQUIRE(op.src[i].c == op.cout, "wsum: source %d has %d channels, expected %d", i, op.src[i].c, op.cout);
    YR_REQUIRE(op.wgt && op.out && op.out_ld % yr_elem<T>::vec == 0 && op.out_ld >= yr_round_up(op.cout, 4), "wsum: bad out/alpha");
    WsumArgs<T> a;
    for (int i = 0; i < 4; ++i) a.s[i] = S.s[i];
    a.alpha = op.wgt; a.out = (T*)op.out; a.H = op.h; a.W = op.w; a.C4 = (op.cout + 3) / 4; a.ld_out = op.out_ld;
    a.total = (long long)batch * op.h * op.w * a.C4;
    static char nm[32];
    static const int nm_len = snprintf(nm, sizeof(nm), "wsum_kernel<%s>", yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    yr_note_kernel(nm);
    hipLaunchKernelGGL(wsum_kernel<T>, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}
int yr_launch_wsum(const yr_op& op, int batch, hipStream_t s) { return YR_BY_DTYPE(op.dtype, launch_wsum_t, op, batch, s); }

// ------------------------------------------------------------------ gather (materialise)
template <class T>
struct GatherArgs {
    DSrcSet S;
    int dense_base[YR_MAX_SRC];  // start of each segment in the dense concat output
    T* out;
    int H, W, KQ, ld_out;
    long long total;
};

template <class T>
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs<T> a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.total) return;
    const int kq = (int)(gid % a.KQ);
    long long t = gid / a.KQ;
    const int x = (int)(t % a.W);
    t /= a.W;
    const int y = (int)(t % a.H);
    const int b = (int)(t / a.H);
    const int k = kq * 4;
    int si = 0;
    for (int i = 1; i < a.S.n; ++i)
        if (k >= a.S.s[i].kbase) si = i;
    DSrc s = a.S.s[0];
    int db = a.dense_base[0];
    if (si == 1) { s = a.S.s[1]; db = a.dense_base[1]; }
    if (si == 2) { s = a.S.s[2]; db = a.dense_base[2]; }
    if (si == 3) { s = a.S.s[3]; db = a.dense_base[3]; }
    const int kk = k - s.kbase;
    const float4 v = yr_load_src_quad<T>(s, b, y, x, kk);
    T* op = a.out + ((size_t)(b * a.H + y) * a.W + x) * a.ld_out + db + kk;
    const float vv[4] = {v.x, v.y, v.z, v.w};
    for (int j = 0; j < 4; ++j)
        if (kk + j < s.c) yr_st1<T>(op + j, vv[j]);
}

template <class T>
static int launch_gather_t(const yr_op& op, int batch, hipStream_t s) {
    GatherArgs<T> a;
    YR_REQUIRE(op.out_dtype == op.dtype, "gather: the output has the op's dtype");
    for (int i = 0; i < op.nsrc && i < YR_MAX_SRC; ++i) YR_REQUIRE(op.src[i].xform != YR_X_DW3, "dw3 sources are a POINTWISE feature");
    int rc = yr_make_srcset(op, &a.S);
    if (rc) return rc;
    int dense = 0;
    for (int i = 0; i < YR_MAX_SRC; ++i) {
        a.dense_base[i] = dense;
        if (i < op.nsrc) dense += op.src[i].c;
    }
    YR_REQUIRE(dense == op.cout && op.out && op.out_ld >= dense, "gather: cout %d != sum of sources %d (or bad out)", op.cout, dense);
    a.out = (T*)op.out; a.H = op.h; a.W = op.w; a.KQ = a.S.kp / 4; a.ld_out = op.out_ld;
    a.total = (long long)batch * op.h * op.w * a.KQ;
    static char nm[32];
    static const int nm_len = snprintf(nm, sizeof(nm), "gather_kernel<%s>", yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    yr_note_kernel(nm);
    hipLaunchKernelGGL(gather_kernel<T>, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}
int yr_launch_gather(const yr_op& op, int batch, hipStream_t s) { return YR_BY_DTYPE(op.dtype, launch_gather_t, op, batch, s); }

// ------------------------------------------------------------------ range check (yr_forward_ranges)
// max |x| over the `c` channels of every pixel row of a float32 tensor [rows][ld], NaN counted as +inf; the result is OR-ed into
// *out as the bits of a non-negative float (which order like unsigned integers).
__global__ __launch_bounds__(256) void absmax_kernel(const float* p, long long rows, int c, int ld, unsigned* out) {
    const int c4 = (c + 3) >> 2;
    const long long total = rows * c4;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / c4;
        const int q = (int)(i - r * c4) * 4;
        const float* e = p + r * ld + q;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q + j < c) {
                const float v = fabsf(e[j]);
                m = (v != v) ? __int_as_float(0x7f800000) : fmaxf(m, v);
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

int yr_launch_absmax(const float* p, long long rows, int c, int ld, unsigned* out, hipStream_t s) {
    if (rows <= 0 || c <= 0) return YR_OK;
    long long blocks = (rows * ((c + 3) / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, rows, c, ld, out);
    YR_LAUNCH_CHECK();
    return YR_OK;
}
