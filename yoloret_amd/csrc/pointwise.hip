// Pointwise (1x1) convolution as an fp32 MFMA GEMM with fused gather / BN / activation /
// residual.  Replaces the TF kernels Conv2D(1x1) + FusedBatchNormV3 + Relu6/Swish + AddV2
// (and the UpSampling2D / MaxPooling2D / Concatenate / Multiply feeding it) used by
// reference code/yolo3/model.py:25-30,98-114,152-155,243-251,298-318 and
// code/yolo3/efficientnet.py:485-496,517-533.
//
// GEMM view: D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k], k running over the
// consumer's padded concat space.  v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain):
//   A operand = weights  (lane l: cout i=l&15, k-group g=l>>4)
//   B operand = pixels   (lane l: pixel j=l&15, k-group g=l>>4)
//   D: lane holds pixel j=l&15, couts (l>>4)*4+r  -> one float4 store of 4 consecutive couts.
// A 16-wide k chunk is staged in LDS; lane group g reads k = k0+4g..4g+3 as one
// ds_read_b128 and uses component s in MFMA step s (the same k permutation on both operands).
#include "yr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PW_BK 16
#define PW_LDS_LD 20  // padded row stride (floats) of the staged tiles

struct PwArgs {
    DSrcSet S;
    const float* wt;      // [N][kp]
    const float* scale;   // [N] or null
    const float* shift;   // [N] or null
    const float* res;     // residual [M][res_ld] or null
    const float* gate;    // SE gate [B][gate_ld] or null
    float* out;           // [M][out_ld]
    int M, H, W, N;
    int out_ld, res_ld, gate_ld;
    int act;
};

// PT/CT: 16-wide pixel / cout MFMA tiles per wave; WM x WN waves (WM*WN == 4).
template <int PT, int CT, int WM, int WN>
__global__ __launch_bounds__(256) void pw_kernel(PwArgs a) {
    constexpr int BM = 16 * PT * WM;
    constexpr int BN = 16 * CT * WN;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * PW_LDS_LD];
    float* As = lds;                   // [BM][PW_LDS_LD] activations
    float* Bs = lds + BM * PW_LDS_LD;  // [BN][PW_LDS_LD] weights

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // loader mapping: quad kq of row r (+64 per pass)
    const int lr = tid >> 2, kq = tid & 3;
    constexpr int A_PASSES = BM / 64;
    constexpr int B_PASSES = (BN + 63) / 64;
    int pb[A_PASSES], py[A_PASSES], px[A_PASSES];
    bool pv[A_PASSES];
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
        const int m = m0 + lr + p * 64;
        pv[p] = m < a.M;
        const int mm = pv[p] ? m : 0;
        const int hw = a.H * a.W;
        pb[p] = mm / hw;
        const int rem = mm - pb[p] * hw;
        py[p] = rem / a.W;
        px[p] = rem - py[p] * a.W;
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15;
    const int kp = a.S.kp;
    for (int k0 = 0; k0 < kp; k0 += PW_BK) {
        const int k = k0 + kq * 4;
        // ---- stage activations
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pv[p]) {
                v = yr_load_cat_quad(a.S, pb[p], py[p], px[p], k);
                if (a.gate != nullptr && k < kp) {
                    const float4 gt = *reinterpret_cast<const float4*>(a.gate + (size_t)pb[p] * a.gate_ld + k);
                    const int rem = a.S.s[0].c - k;  // lanes beyond the channel count stay exactly 0
                    v.x *= gt.x;
                    v.y = rem > 1 ? v.y * gt.y : 0.f;
                    v.z = rem > 2 ? v.z * gt.z : 0.f;
                    v.w = rem > 3 ? v.w * gt.w : 0.f;
                }
            }
            *reinterpret_cast<float4*>(As + (lr + p * 64) * PW_LDS_LD + kq * 4) = v;
        }
        // ---- stage weights
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int r = lr + p * 64;
            if (r < BN) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int n = n0 + r;
                if (n < a.N && k < kp) v = *reinterpret_cast<const float4*>(a.wt + (size_t)n * kp + k);
                *reinterpret_cast<float4*>(Bs + r * PW_LDS_LD + kq * 4) = v;
            }
        }
        __syncthreads();
        // ---- fragments + MFMA
        f32x4 wf[CT], xf[PT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
            wf[c] = *reinterpret_cast<const f32x4*>(Bs + ((wn * CT + c) * 16 + li) * PW_LDS_LD + g * 4);
#pragma unroll
        for (int p = 0; p < PT; ++p)
            xf[p] = *reinterpret_cast<const f32x4*>(As + ((wm * PT + p) * 16 + li) * PW_LDS_LD + g * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][s], xf[p][s], acc[c][p], 0, 0, 0);
        __syncthreads();
    }

    // ---- epilogue: BN scale/shift, activation, residual, store (4 consecutive couts per lane)
    const bool vec_out = (a.out_ld & 3) == 0;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = n0 + (wn * CT + c) * 16 + g * 4;
        if (n >= a.N) continue;
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < a.N) {
                if (a.scale) sc[r] = a.scale[n + r];
                if (a.shift) sh[r] = a.shift[n + r];
            }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = m0 + (wm * PT + p) * 16 + li;
            if (m >= a.M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = yr_apply_act(__builtin_fmaf(acc[c][p][r], sc[r], sh[r]), a.act);
            if (a.res) {
                const float* rp = a.res + (size_t)m * a.res_ld + n;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < a.N) v[r] += rp[r];
            }
            float* op = a.out + (size_t)m * a.out_ld + n;
            if (vec_out && n + 3 < a.N) {
                *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < a.N) op[r] = v[r];
            }
        }
    }
}

template <int PT, int CT, int WM, int WN>
static int launch_cfg(const PwArgs& a, hipStream_t s) {
    constexpr int BM = 16 * PT * WM, BN = 16 * CT * WN;
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN);
    static char nm[40];
    static const int nm_len = snprintf(nm, sizeof(nm), "pw_kernel<%d,%d,%d,%d>", PT, CT, WM, WN);
    (void)nm_len;
    yr_note_kernel(nm);
    hipLaunchKernelGGL((pw_kernel<PT, CT, WM, WN>), grid, dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

int yr_launch_pointwise(const yr_op& op, int batch, hipStream_t s) {
    PwArgs a;
    int rc = yr_make_srcset(op, &a.S);
    if (rc) return rc;
    int cin = 0;
    for (int i = 0; i < op.nsrc; ++i) cin += op.src[i].c;
    YR_REQUIRE(cin == op.cin, "pointwise: sum of source channels %d != cin %d", cin, op.cin);
    YR_REQUIRE(op.cout >= 1 && op.out != nullptr && op.wgt != nullptr, "pointwise: missing out/weights");
    YR_REQUIRE(op.out_ld >= op.cout, "pointwise: out_ld %d < cout %d", op.out_ld, op.cout);
    YR_REQUIRE(((uintptr_t)op.wgt % 16) == 0, "pointwise: weights must be 16-byte aligned");
    if ((op.out_ld & 3) == 0) YR_REQUIRE(((uintptr_t)op.out % 16) == 0, "pointwise: out must be 16-byte aligned");
    if (op.gate) {
        YR_REQUIRE(op.nsrc == 1 && op.gate_ld % 4 == 0 && op.gate_ld >= a.S.kp, "pointwise: SE gate needs a single source and gate_ld >= kp");
        YR_REQUIRE(((uintptr_t)op.gate % 16) == 0, "pointwise: gate must be 16-byte aligned");
    }
    a.wt = op.wgt; a.scale = op.scale; a.shift = op.shift; a.res = op.res; a.gate = op.gate; a.out = op.out;
    a.H = op.h; a.W = op.w; a.N = op.cout;
    const long long M = (long long)batch * op.h * op.w;
    YR_REQUIRE(M > 0 && M < (1ll << 31), "pointwise: pixel count %lld out of range", M);
    a.M = (int)M;
    a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.gate_ld = op.gate_ld; a.act = op.act;
    // ---- tile selection.  Candidates (BM x BN); big-M layers take 128-row tiles, layers with few
    // pixels (13x13 / 26x26 maps) take 64-row tiles and, if still short of ~2 workgroups per CU,
    // narrower cout tiles - these layers are latency/occupancy-bound, not bandwidth-bound.
    struct Cfg { int bm, bn; int (*fn)(const PwArgs&, hipStream_t); };
    static const Cfg big[] = {{256, 16, launch_cfg<4, 1, 4, 1>}, {128, 32, launch_cfg<2, 2, 4, 1>},
                              {128, 48, launch_cfg<2, 3, 4, 1>}, {128, 64, launch_cfg<4, 2, 2, 2>},
                              {128, 80, launch_cfg<2, 5, 4, 1>}, {128, 96, launch_cfg<4, 3, 2, 2>},
                              {128, 128, launch_cfg<4, 4, 2, 2>}};
    static const Cfg small[] = {{64, 16, launch_cfg<1, 1, 4, 1>}, {64, 32, launch_cfg<1, 2, 4, 1>},
                                {64, 48, launch_cfg<1, 3, 4, 1>}, {64, 64, launch_cfg<1, 4, 4, 1>},
                                {64, 80, launch_cfg<1, 5, 4, 1>}, {64, 96, launch_cfg<1, 6, 4, 1>},
                                {64, 128, launch_cfg<1, 8, 4, 1>}};
    const int N = op.cout;
    // cost model (seconds, rough): activations re-read once per cout tile (mostly from L2/MALL),
    // MFMA work on the padded cout width, and a penalty when the grid cannot fill the chip.
    const double Md = (double)a.M, Kd = (double)a.S.kp;
    auto cost = [&](const Cfg& c) {
        const double ntn = (double)((N + c.bn - 1) / c.bn), npad = ntn * c.bn;
        const double nblk = (double)((a.M + c.bm - 1) / c.bm) * ntn;
        const double t_mem = (Md * Kd * 4.0 * (1.0 + 0.3 * (ntn - 1.0)) + Md * N * 4.0) / 4.0e12;
        const double t_cmp = 2.0 * Md * npad * Kd / (c.bm >= 128 ? 60.0e12 : 45.0e12);
        const double fill = nblk < 512.0 ? 512.0 / nblk : 1.0;
        return (t_mem + t_cmp) * fill;
    };
    const Cfg* best = &big[0];
    double bc = cost(big[0]);
    for (int i = 0; i < 7; ++i) {
        if (cost(big[i]) < bc) { bc = cost(big[i]); best = &big[i]; }
        if (cost(small[i]) < bc) { bc = cost(small[i]); best = &small[i]; }
    }
    return best->fn(a, s);
}
