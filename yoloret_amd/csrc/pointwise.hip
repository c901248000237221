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
// Lane group g uses k = k0+4g..4g+3 of a 16-deep chunk, component s in MFMA step s (the same k
// permutation on both operands).  Two kernels run that same MFMA sequence per output (bit-identical):
//   pw_kernel  (pointwise_lds.hip) - the chunk is staged through LDS by coalesced loads;
//   pwd_kernel (this file)         - no LDS, no barriers: every wave loads both operands straight into
//                                    operand layout and keeps D chunks of loads in flight.
// This file also holds the dispatcher (tile-shape table, heuristic, autotune hook).
#include <stdlib.h>

#include "pw_common.h"

// ------------------------------------------------------------------------------------------ direct kernel
// Every wave owns 16*PT pixels x 16*CT couts and loads BOTH MFMA operands straight from global memory in the
// operand layout (lane l: row l&15, k quad l>>4 - one 64-byte segment per row and instruction; measured on
// MI355X with tools/ldpat.hip: this map streams at the same 6.4 TB/s from HBM as the 4-adjacent-lanes map, and
// > 5 TB/s from L2).  D register sets keep D k chunks of loads in flight per wave (the steady-state loop is
// branch-free, so the compiler counts outstanding loads exactly: s_waitcnt vmcnt(11/10/9) for D = 4); waves
// never wait for each other.  Weight fragments are re-read per wave from L1/L2 (a layer's weights <= 350 KB).
template <int PT, int CT, int D, int MODE>
__global__ __launch_bounds__(256) void pwd_kernel(PwArgs a) {
    constexpr int BM = 64 * PT, BN = 16 * CT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const unsigned ntn = (a.N + BN - 1) / BN;
    const unsigned L = yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (int)(L / ntn) * BM + wave * 16 * PT;
    const int n0 = (int)(L % ntn) * BN;
    const int kp = a.S.kp;
    const int kl = g * 4;  // this lane's k offset inside a 16-deep chunk

    PwRow<MODE> row[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) row[p].init(a, m0 + p * 16 + li);
    const float* brow[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = n0 + c * 16 + li;
        brow[c] = a.wt + (size_t)(n < a.N ? n : 0) * kp;  // rows beyond N feed couts that are never stored
    }

    struct Frag {
        float4 x[PT], w[CT];
        float4 gt[MODE == 2 ? PT : 1];
        int cv[PT];
    };
    auto load = [&](int chunk, Frag& F) {
        const int kraw = chunk * 16 + kl;
#pragma unroll
        for (int p = 0; p < PT; ++p) row[p].issue(a, kraw, kp, F.x[p], F.gt[MODE == 2 ? p : 0], F.cv[p]);
        const int k = kraw < kp ? kraw : kp - 4;  // the k tail of the weights meets zeroed activations
#pragma unroll
        for (int c = 0; c < CT; ++c) F.w[c] = *reinterpret_cast<const float4*>(brow[c] + k);
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // uniform: most layers have whole quads and whole 16-deep chunks only, and then nothing needs masking
    // (rows beyond M only feed outputs that are never stored)
    const bool need_mask = MODE != 1 || (a.S.s[0].c & 3) != 0 || (kp & 15) != 0;

    auto use = [&](const Frag& F) {
        float xf[PT][4];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            float4 v = F.x[p];
            if (need_mask) v = pw_finish<MODE>(v, F.gt[MODE == 2 ? p : 0], F.cv[p]);
            xf[p][0] = v.x; xf[p][1] = v.y; xf[p][2] = v.z; xf[p][3] = v.w;
        }
        const float* wq = reinterpret_cast<const float*>(&F.w[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[c * 4 + s], xf[p][s], acc[c][p], 0, 0, 0);
    };

    const int nch = (kp + 15) >> 4;  // chunk j lives in F[j % D]
    Frag F[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nch) load(d, F[d]);
    int j0 = 0;
    for (; j0 + 2 * D - 1 < nch; j0 += D) {  // steady state, branch-free: the whole next group exists
#pragma unroll
        for (int d = 0; d < D; ++d) {
            use(F[d]);
            load(j0 + D + d, F[d]);
        }
    }
#pragma unroll
    for (int t = 0; t < 2 * D - 1; ++t) {    // drain: at most 2D-1 chunks left, the later ones still to be loaded
        const int j = j0 + t;
        if (j < nch) {
            use(F[t % D]);
            if (j + D < nch) load(j + D, F[t % D]);
        }
    }

    // ---- epilogue (pw_finish_quad): 4 consecutive couts per lane.  The BN scale / shift of ALL the wave's couts are
    // fetched first, unconditionally (clamped): loads under per-lane branches each wait for their own round trip.
    const bool vec_out = (a.out_ld & 3) == 0;
    const bool vec_res = (a.res_ld & 3) == 0, vec_pre = (a.pre_ld & 3) == 0;
    f32x4 sc[CT], sh[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + c * 16 + g * 4 + r;
            const int nc = n < a.N ? n : a.N - 1;
            sc[c][r] = a.scale ? a.scale[nc] : 1.f;
            sh[c][r] = a.shift ? a.shift[nc] : 0.f;
        }
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p)
            pw_finish_quad(a, acc[c][p], sc[c], sh[c], m0 + p * 16 + li, n0 + c * 16 + g * 4, li, vec_out, vec_res, vec_pre);
}

template <int PT, int CT>
static int launch_direct(const PwArgs& a, hipStream_t s) {
    constexpr int BM = 64 * PT, BN = 16 * CT;
    constexpr int D = PT * CT <= 2 ? 4 : (PT * CT <= 6 ? 3 : 2);
    dim3 grid((unsigned)((a.M + BM - 1) / BM) * (unsigned)((a.N + BN - 1) / BN));
    const int mode = (a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY) ? (a.gate ? 2 : 1) : 0;
    if (mode == 0 && a.gate) { yr_set_error("pointwise: an SE gate needs one identity source"); return YR_ERR_ARG; }
    static char nm[3][48];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pwd_kernel<%d,%d,%d,0>", PT, CT, D) +
                              snprintf(nm[1], sizeof(nm[1]), "pwd_kernel<%d,%d,%d,1>", PT, CT, D) +
                              snprintf(nm[2], sizeof(nm[2]), "pwd_kernel<%d,%d,%d,2>", PT, CT, D);
    (void)nm_len;
    yr_note_kernel(nm[mode]);
    if (mode == 1) hipLaunchKernelGGL((pwd_kernel<PT, CT, D, 1>), grid, dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL((pwd_kernel<PT, CT, D, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pwd_kernel<PT, CT, D, 0>), grid, dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <int SHAPE>
static int launch_lds(const PwArgs& a, hipStream_t s) { return yr_pw_launch_lds(SHAPE, a, s); }

int yr_launch_pointwise(const yr_op& op_in, int batch, hipStream_t s) {
    PwArgs a;
    yr_op op = op_in;
    a.pre = nullptr; a.pre_ld = 0;
    a.out2 = nullptr; a.out2_ld = a.N2 = a.act2 = a.pool2 = 0;
    YR_REQUIRE(yr_dtype_ok(op.dtype) && (op.out_dtype == op.dtype || op.out_dtype == YR_F32),
               "pointwise: dtype %d / out_dtype %d unsupported (the output has the op's dtype or is float32)", op.dtype, op.out_dtype);
    const bool narrow = op.dtype != YR_F32;
    const int V = yr_vec_of(op.dtype);
    if (op.nsrc >= 2 && op.src[op.nsrc - 1].xform == YR_X_UP2_ADD) {  // not a k-space source: see yr_xform
        const yr_src& p = op.src[op.nsrc - 1];
        YR_REQUIRE(p.ptr && p.c == op.cout && p.ld >= op.cout && p.h * 2 == op.h && p.w * 2 == op.w && p.dtype == YR_F32,
                   "pointwise: the up2_add source must be float32 [B,%d,%d,cout=%d]", op.h / 2, op.w / 2, op.cout);
        a.pre = (const float*)p.ptr; a.pre_ld = p.ld;
        op.nsrc -= 1;
    }
    for (int i = 0; i < op.nsrc; ++i)
        YR_REQUIRE(op.src[i].xform != YR_X_UP2_ADD, "pointwise: up2_add is only valid as the last of >= 2 sources");
    // stride 2 on a POINTWISE op: the output is MaxPooling2D(2) of the conv (+BN+act) result (model.py:139-144 after
    // the bottom-up convs); op.h/op.w are the pooled OUTPUT dims, the sources sit at twice that
    a.pool = 0;
    if (op.stride == 2) {
        YR_REQUIRE(a.pre == nullptr && op.res == nullptr, "pointwise: a pooled output takes no residual / up2_add");
        a.pool = 1;
        op.h *= 2; op.w *= 2;
    } else {
        YR_REQUIRE(op.stride == 0 || op.stride == 1, "pointwise: stride %d unsupported", op.stride);
    }
    int rc = yr_make_srcset(op, &a.S);
    if (rc) return rc;
    // the depthwise stage of an inverted-residual block folded into this conv's loads (yr_xform: YR_X_DW3)
    a.dw_w = a.dw_scale = a.dw_shift = nullptr;
    a.dw_stride = a.dw_act = a.dw_pad_t = a.dw_pad_l = 0;
    const bool dw = op.src[0].xform == YR_X_DW3;
    YR_REQUIRE(!(dw && narrow), "pointwise: a dw3 source is a float32 feature");
    if (dw) {
        const yr_src& e = op.src[0];
        YR_REQUIRE(a.pre == nullptr && !a.pool && op.gate == nullptr, "pointwise: a dw3 source takes no up2_add / pooled output / SE gate");
        YR_REQUIRE(op.wgt2 && op.b1 && op.b2 && (((uintptr_t)op.wgt2 | (uintptr_t)op.b1 | (uintptr_t)op.b2) % 16) == 0,
                   "pointwise: dw3 source needs 16-byte aligned wgt2 / b1 / b2");
        YR_REQUIRE(11ll * a.S.kp * (long long)sizeof(float) <= 48 * 1024, "pointwise: dw3 source with %d channels does not fit LDS", e.c);
        a.dw_w = op.wgt2; a.dw_scale = op.b1; a.dw_shift = op.b2;
        a.dw_stride = op.se_reduced & 0xff;
        a.dw_act = (op.se_reduced >> 8) & 0xff;
        // TF 'SAME': pad_total = max((out-1)*s + k - in, 0); before = total/2 (extra goes bottom/right)
        const int pth = (op.h - 1) * a.dw_stride + 3 - e.h, ptw = (op.w - 1) * a.dw_stride + 3 - e.w;
        a.dw_pad_t = (pth > 0 ? pth : 0) / 2;
        a.dw_pad_l = (ptw > 0 ? ptw : 0) / 2;
    }
    int cin = 0;
    for (int i = 0; i < op.nsrc; ++i) cin += op.src[i].c;
    YR_REQUIRE(cin == op.cin, "pointwise: sum of source channels %d != cin %d", cin, op.cin);
    YR_REQUIRE(op.cout >= 1 && op.out != nullptr && op.wgt != nullptr, "pointwise: missing out/weights");
    YR_REQUIRE(op.out_ld >= op.cout, "pointwise: out_ld %d < cout %d", op.out_ld, op.cout);
    YR_REQUIRE(((uintptr_t)op.wgt % 16) == 0, "pointwise: weights must be 16-byte aligned");
    if ((op.out_ld & 3) == 0) YR_REQUIRE(((uintptr_t)op.out % 16) == 0, "pointwise: out must be 16-byte aligned");
    if (op.out_dtype != YR_F32)
        YR_REQUIRE(op.out_ld % 8 == 0 && op.out_ld >= yr_round_up(op.cout, 8), "pointwise: a 16-bit output needs out_ld %% 8 == 0 and >= round_up(cout,8)");
    if (narrow && op.res)
        YR_REQUIRE(op.res_ld % 8 == 0 && op.res_ld >= yr_round_up(op.cout, 8) && ((uintptr_t)op.res % 16) == 0,
                   "pointwise: a 16-bit residual needs res_ld %% 8 == 0, >= round_up(cout,8) and a 16-byte aligned pointer");
    if (op.gate) {
        YR_REQUIRE(op.nsrc == 1 && op.gate_ld % 4 == 0 && op.gate_ld >= a.S.kp, "pointwise: SE gate needs a single source and gate_ld >= kp");
        YR_REQUIRE(((uintptr_t)op.gate % 16) == 0, "pointwise: gate must be 16-byte aligned");
    }
    YR_REQUIRE(a.S.kp >= V, "pointwise: no input channels");
    a.wt = op.wgt; a.scale = op.scale; a.shift = op.shift; a.res = (const float*)op.res; a.gate = op.gate; a.out = (float*)op.out;
    a.out_f32 = op.out_dtype == YR_F32;
    a.H = op.h; a.W = op.w; a.N = op.cout;
    const long long M = (long long)batch * op.h * op.w;
    YR_REQUIRE(M > 0 && M < (1ll << 31), "pointwise: pixel count %lld out of range", M);
    a.M = (int)M;
    a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.gate_ld = op.gate_ld; a.act = op.act;
    // ---- tile selection.  Candidates (BM x BN); big-M layers take 128-row tiles, layers with few
    // pixels (13x13 / 26x26 maps) take 64-row tiles and, if still short of ~2 workgroups per CU,
    // narrower cout tiles - these layers are latency/occupancy-bound, not bandwidth-bound.
    struct Cfg { int bm, bn; int (*fn)(const PwArgs&, hipStream_t); };
    static const Cfg cfgs[] = {{256, 16, launch_lds<0>}, {128, 32, launch_lds<1>}, {128, 48, launch_lds<2>},
                               {128, 64, launch_lds<3>}, {128, 80, launch_lds<4>}, {128, 96, launch_lds<5>},
                               {128, 128, launch_lds<6>},
                               {64, 16, launch_lds<7>}, {64, 32, launch_lds<8>}, {64, 48, launch_lds<9>},
                               {64, 64, launch_lds<10>}, {64, 80, launch_lds<11>}, {64, 96, launch_lds<12>},
                               {64, 128, launch_lds<13>},
                               // direct (LDS-free) variants, BM = 64*PT, BN = 16*CT
                               {64, 16, launch_direct<1, 1>}, {64, 32, launch_direct<1, 2>}, {64, 48, launch_direct<1, 3>},
                               {64, 64, launch_direct<1, 4>}, {64, 80, launch_direct<1, 5>}, {64, 96, launch_direct<1, 6>},
                               {64, 128, launch_direct<1, 8>},
                               {128, 32, launch_direct<2, 2>}, {128, 48, launch_direct<2, 3>}, {128, 64, launch_direct<2, 4>},
                               {128, 80, launch_direct<2, 5>}, {128, 96, launch_direct<2, 6>},
                               {256, 32, launch_direct<4, 2>}, {256, 48, launch_direct<4, 3>}, {256, 64, launch_direct<4, 4>}};
    if (narrow && (op.se_reduced & 0x20000) && a.S.kp >= 2 * 32 && a.dw_w == nullptr) return yr_pwh_launch_ksplit(op.dtype, a, s);   // (the plan asks for the k-split form)
    if (narrow) return yr_pw_launch_h(op.dtype, op.k - 1, a, s);   // bf16 / f16: pointwise_h.hip (its own tile table)
    constexpr int NLDS = 14;  // the first NLDS entries are the LDS-staged kernel (the heuristic below only ranks those)
    constexpr int NCFG = sizeof(cfgs) / sizeof(cfgs[0]);
    const int N = op.cout;
    // autotuned choice (yr_autotune stores the fastest shape per op and batch): op.k = 1 + index
    if (dw) {  // LDS-staged kernel only; shapes it is not built for fall back inside yr_pw_launch_lds
        const int shape = (op.k >= 1 && op.k <= NLDS) ? op.k - 1 : -1;
        return yr_pw_launch_lds(shape, a, s);
    }
    // The SPLIT form (pointwise_split.hip): every float32 conv at least 16 channels deep without a depthwise-folded source - by
    // the op's SHAPE, never by the tuner or the batch (the two forms round differently: a batch must equal its images run one
    // by one).  The tuner's index picks the tile shape (the direct kernels' indices map onto the LDS shapes).  YOLORET_PW_SPLIT=0:
    // the float32-MFMA kernels.
    static const bool split_on = !(getenv("YOLORET_PW_SPLIT") && atoi(getenv("YOLORET_PW_SPLIT")) == 0);
    const bool split = split_on && a.S.kp >= 16 && !(op.se_reduced & 0x10000);   // (bit 16 of se_reduced: the plan asks for the float32 MFMA - its few-image form)   // (a 16- or 24-deep conv pads its one step with zeros: the MFMAs are not what it waits for)
    // bit 17 of se_reduced: the plan asks for the k-split form (the 'nohead' variant's small maps; pointwise_split.hip) - a property of the
    // PLAN like the split form itself, so the tuner's index is not looked at
    if (op.se_reduced & 0x40000) {     // bit 18: the pixel-stationary form - op.wgt holds float16 planes, no other kernel can read them
        YR_REQUIRE(split, "pointwise: the plan stores this op's weights as float16 planes (se_reduced bit 18) but the split form is off");
        if (op.se_reduced & 0x80000) {     // bit 19: a second conv of the same source in the same launch (its output: gate_out, its width: se_hidden)
            YR_REQUIRE(op.gate_out != nullptr && op.se_hidden >= 1 && op.gate_out_ld >= op.se_hidden, "pointwise: the second output of a two-output op is missing or too narrow");
            a.out2 = op.gate_out; a.out2_ld = op.gate_out_ld; a.N2 = op.se_hidden;
            a.act2 = op.reserved0 & 0xff; a.pool2 = (op.reserved0 >> 8) & 1;
        }
        return yr_pw_launch_stream(a, s);
    }
    if (split && (op.se_reduced & 0x20000) && a.S.kp >= 2 * 32) return yr_pw_launch_ksplit(a, s);
    if (split && op.k >= 1 && op.k <= NCFG) return yr_pw_launch_split((op.k - 1) % NLDS, a, s);
    if (op.k >= 1 && op.k <= NCFG) return cfgs[op.k - 1].fn(a, s);
    // tuning override: YR_PW_CFG="BMxBN" forces one tile shape for every layer (experiments only)
    static const char* force = getenv("YR_PW_CFG");
    if (force) {
        int bm = 0, bn = 0;
        if (sscanf(force, "%dx%d", &bm, &bn) == 2)
            for (int i = 0; i < NCFG; ++i)
                if (cfgs[i].bm == bm && cfgs[i].bn == bn) return cfgs[i].fn(a, s);
    }
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
    const Cfg* best = &cfgs[0];
    double bc = cost(cfgs[0]);
    for (int i = 1; i < NLDS; ++i)
        if (cost(cfgs[i]) < bc) { bc = cost(cfgs[i]); best = &cfgs[i]; }
    if (split) return yr_pw_launch_split((int)(best - cfgs), a, s);
    return best->fn(a, s);
}

// number of tile shapes yr_launch_pointwise can be forced to through op.k (1-based) for ops of this dtype
int yr_pointwise_num_cfgs(int dtype) { return dtype == YR_F32 ? 29 : yr_pwh_num_cfgs(); }
