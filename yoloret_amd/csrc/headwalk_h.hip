// YR_OP_HEAD, WALKING form, 16-BIT plans (k bit 6, dtype bf16 / f16; round 5): the detection-head block's 1x1 conv + BN + ReLU6 ->
// depthwise 3x3 + BN + Swish -> squeeze-excite sums (reference code/yolo3/model.py:91-115, efficientnet.py:406-438,467-536) in one
// launch, for the EfficientNet configurations (BASELINE c3 / c5 and the SE EfficientNets), the 16-bit twin of headwalk.hip:
//   * a WAVE owns a strip of 16 input columns x a run of rows x NT = 2 output-channel tiles of 16 and walks down the rows; the conv
//     of one strip row is ONE v_mfma_f32_16x16x32_{bf16,f16} per (tile, 32-channel chunk) - the sources are 16-bit already, so the
//     pixel operand is the 16 bytes a lane loads (8 channels of its pixel, a row ahead), no plane cutting, and the weight fragments
//     of the wave's tiles (4 registers per tile and chunk: half of the float32 form's) stay in registers: up to 8 chunks;
//   * float32 from the accumulator on: BN scale / shift (+ the up-sampled pre-BN addend, float32), ReLU6, the depthwise conv by DPP
//     row shifts on the MFMA result registers (mbr_dw_row) with the last three rows of the walk in registers, BN shift as first
//     addend, Swish; ONE rounding to the 16-bit type at the 8-byte store.  The F-wide conv output (written once and read once by
//     the unfused pair: at 52 x 52 x 128 x 128 images 88 MB each way) never exists - and is never rounded to 16 bits either;
//   * squeeze-excite sums of the STORED (rounded) values, as dwp_kernel forms them, per (strip, segment) row: rows of a shape, not
//     of the batch.
// The k space is a concatenation of up to three identity sources in chunks of 32 channels per source; a gated single source
// (bu3 reads td3's map through its SE gate) has the gate folded into the stationary weights once per wave (w * g in float32, one
// rounding to the operand type - the pointwise kernels round x * g instead: the same size of error, on the other operand).
#include "mbr_common.h"
#include "pwh_common.h"

#define HWH_MAXK 8
struct HwhArgs {
    const void* src[3]; int ld[3]; int cs[3];    // k-space sources (identity, 16-bit): pointer, channel stride, channels
    int nsrc;
    int csrc[HWH_MAXK], ckl[HWH_MAXK];           // chunk -> source, first channel within the source
    const void* wa;      // A fragments [T][NKE][64 lanes][8 elements] of the 16-bit type (compiler.head_pack16; no BN scale inside)
    const float* wt;     // [T][11][16]: depthwise taps x BN scale | depthwise BN shift | conv BN shift
    const float* scale;  // conv BN scale [F]
    const float* pre; int pre_ld;                // float32 [B][H/2][W/2][pre_ld] or null
    const float* gate; int gate_ld;              // SE gate of the single source [B][gate_ld] or null
    void* out; int ld_out;
    int H, W, T, F, strips, segs, seg_rows, groups, act, dw_act;
    float* sums; int ld_sums;                    // [B][strips * segs][ld_sums] or null
};

template <class T16, int NKE, int NT, bool PRE, bool GATED>
__global__ __launch_bounds__(256, 2) void hwalkh_kernel(HwhArgs a) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef T16 t4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float tab[];
    for (int i = threadIdx.x; i < a.T * MBR_TAB; i += 256) tab[i] = a.wt[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a workgroup = four consecutive tile groups of one (image, strip, segment) - groups % 4 == 0 (launcher)
    int gw = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x) * 4 + wave;
    const int g = gw % a.groups; gw /= a.groups;
    const int seg = gw % a.segs; gw /= a.segs;
    const int strip = gw % a.strips;
    const int b = gw / a.strips;
    const int t0 = g * NT;
    const int yo0 = seg * a.seg_rows, yo1 = min(yo0 + a.seg_rows, a.H);
    const int xin = 14 * strip - 1 + px;
    const int xc = min(max(xin, 0), a.W - 1);
    const float hi = (xin >= 0 && xin < a.W) ? 1.f : 0.f;       // 0 outside the map: TF's zero padding of the depthwise input
    const int xo = 14 * strip + px - 1;
    const bool out_lane = px >= 1 && px <= 14 && xo < a.W;

    // ---- stationary: the weight fragments of this wave's tiles, conv BN scale / shift
    pwh_u4 wf[NT][NKE];
    v4f se[NT], psc[NT];
    unsigned ooff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = min(t0 + j, a.T - 1);
        const pwh_u4* pe = reinterpret_cast<const pwh_u4*>(a.wa) + ((size_t)t * NKE) * 64 + lane;
#pragma unroll
        for (int c = 0; c < NKE; ++c) wf[j][c] = pe[c * 64];
        se[j] = *reinterpret_cast<const v4f*>(a.wt + (size_t)t * MBR_TAB + 160 + 4 * mg);
        psc[j] = *reinterpret_cast<const v4f*>(a.scale + 16 * t + 4 * mg);
        ooff[j] = (t0 + j < a.T && out_lane) ? (16u * (t0 + j) + 4u * mg) * 2u : MBR_DEAD;
    }
    if constexpr (GATED) {
        // W (g . x) = (W diag g) x and the wave works on ONE image: its fragments take the gate once (float32 product, one rounding)
#pragma unroll
        for (int c = 0; c < NKE; ++c) {
            const float* gp = a.gate + (size_t)b * a.gate_ld + a.ckl[c] + 8 * mg;
            const bool ok = a.ckl[c] + 8 * mg < a.cs[0], ok2 = ok && a.ckl[c] + 8 * mg + 4 < ((a.cs[0] + 3) & ~3);
            const v4f g0 = ok ? *reinterpret_cast<const v4f*>(gp) : (v4f){0.f, 0.f, 0.f, 0.f};
            const v4f g1 = ok2 ? *reinterpret_cast<const v4f*>(gp + 4) : (v4f){0.f, 0.f, 0.f, 0.f};
            const pwh_f8 g8 = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j][c] = pwh_narrow<T16>(pwh_widen<T16>(wf[j][c]) * g8);   // (whole-vector casts: headwalk.hip's note)
        }
    }
    // ---- descriptors: one per source (whole batch: 32-bit offsets, the launcher checks the sizes), the addend, the output image
    const mbr_rsrc rs0 = mbr_make_rsrc(a.src[0], 0x7effffffu);
    const mbr_rsrc rs1 = mbr_make_rsrc(a.nsrc > 1 ? a.src[1] : a.src[0], 0x7effffffu);
    const mbr_rsrc rs2 = mbr_make_rsrc(a.nsrc > 2 ? a.src[2] : a.src[0], 0x7effffffu);
    const mbr_rsrc prs = mbr_make_rsrc(PRE ? (const void*)(a.pre + (size_t)b * (a.H >> 1) * (a.W >> 1) * a.pre_ld) : a.src[0], PRE ? (unsigned)((a.H >> 1) * (a.W >> 1) * a.pre_ld) * 4u : 0u);
    const mbr_rsrc osrc = mbr_make_rsrc(reinterpret_cast<char*>(a.out) + (size_t)b * a.H * a.W * a.ld_out * 2, (unsigned)(a.H * a.W * a.ld_out) * 2u);
    // per chunk: byte offset of this lane's 8 channels in its pixel (row offset added per row), or dead; the row pitch of its source
    unsigned xsoff[NKE], xpitch[NKE];
    int vcc[NKE];
#pragma unroll
    for (int c = 0; c < NKE; ++c) {
        const int s = a.csrc[c], kl = a.ckl[c] + 8 * mg;
        const int cs = s == 0 ? a.cs[0] : s == 1 ? a.cs[1] : a.cs[2], ld = s == 0 ? a.ld[0] : s == 1 ? a.ld[1] : a.ld[2];
        xsoff[c] = kl < ((cs + 7) & ~7) ? (unsigned)((b * a.H * a.W + xc) * ld + kl) * 2u : MBR_DEAD;
        xpitch[c] = (unsigned)(a.W * ld) * 2u;
        vcc[c] = cs - a.ckl[c];                 // valid channels of the chunk (>= 32: all)
    }
    const unsigned pcol = PRE ? (unsigned)((xc >> 1) * a.pre_ld + 4 * mg) * 4u : 0u;

    struct XRow { pwh_u4 m[NKE]; v4f p[PRE ? NT : 1]; };
    XRow xa, xb;
    auto load_row = [&](XRow& x, int r) __attribute__((always_inline)) {
        const int rc = min(max(r, 0), a.H - 1);
#pragma unroll
        for (int c = 0; c < NKE; ++c) {
            const unsigned so = (unsigned)rc * xpitch[c];
            const int s = c == 0 ? 0 : a.csrc[c];     // (chunk 0 is the first source's: headwalk.hip's note on descriptors in scratch)
            if (s == 0) x.m[c] = __builtin_amdgcn_raw_buffer_load_b128(rs0, xsoff[c], so, 0);
            else if (s == 1) x.m[c] = __builtin_amdgcn_raw_buffer_load_b128(rs1, xsoff[c], so, 0);
            else x.m[c] = __builtin_amdgcn_raw_buffer_load_b128(rs2, xsoff[c], so, 0);
        }
        if constexpr (PRE) {
            const unsigned so = (unsigned)((rc >> 1) * (a.W >> 1) * a.pre_ld) * 4u;
#pragma unroll
            for (int j = 0; j < NT; ++j)
                x.p[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(prs, t0 + j < a.T ? pcol + 64u * (t0 + j) : MBR_DEAD, so, 0));
        }
    };
    const int rbeg = yo0 - 1, nout = yo1 - yo0;
    load_row(xa, rbeg);
    v4f ea[NT], eb[NT], psum[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { ea[j] = (v4f){0.f, 0.f, 0.f, 0.f}; eb[j] = ea[j]; psum[j] = ea[j]; }
    const float actmax = a.act == YR_ACT_RELU6 ? 6.f : 3.0e38f;     // conv activation: ReLU6, or none
    const float actmin = a.act == YR_ACT_RELU6 ? 0.f : -3.0e38f;

    auto row = [&](auto emit_c, const int k, const int yo, const XRow& xc_, XRow& xn_) __attribute__((always_inline)) {
        constexpr bool EMIT = decltype(emit_c)::value;
        const int r = rbeg + k;
        load_row(xn_, r + 1);
        const float live = (r >= 0 && r < a.H) ? hi : 0.f;
        v4f ec[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) ec[j] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NKE; ++c) {
            pwh_u4 xv = xc_.m[c];
            if (vcc[c] < 32) xv = pwh_mask(xv, vcc[c] - 8 * mg);   // uniform: a source's last chunk - pad channels may hold anything
#pragma unroll
            for (int j = 0; j < NT; ++j) ec[j] = pwh_mfma<T16>(wf[j][c], xv, ec[j]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            v4f base = se[j];
            if constexpr (PRE) base = psc[j] * xc_.p[j] + base;       // (acc + pre) * scale + shift
            ec[j] = ec[j] * psc[j] + base;
#pragma unroll
            for (int i = 0; i < 4; ++i) ec[j][i] = __builtin_amdgcn_fmed3f(ec[j][i], actmin, actmax) * live;
        }
        if constexpr (EMIT) {
            const unsigned opix = ((unsigned)yo * (unsigned)a.W + (unsigned)xo) * (unsigned)a.ld_out * 2u;
            // (7 - 8 chunks: left to itself hipcc hoists the 20 tap reads out of the walk and then spills 30 - 130 bytes per lane;
            //  an offset it cannot see through keeps the reads - 20 ds_read_b128 per row - where they are)
            unsigned toff = 0;
            if constexpr (NKE >= 7) asm volatile("" : "+v"(toff));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const v4f* tb = reinterpret_cast<const v4f*>(tab + toff + min(t0 + j, a.T - 1) * MBR_TAB) + mg;
                v4f d = tb[36];
                mbr_dw_row(d, ea[j], tb[0], tb[4], tb[8]);
                mbr_dw_row(d, eb[j], tb[12], tb[16], tb[20]);
                mbr_dw_row(d, ec[j], tb[24], tb[28], tb[32]);
                if (a.dw_act == YR_ACT_SWISH) {   // uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = d[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(d[i] * -1.44269504088896341f));
                } else if (a.dw_act == YR_ACT_RELU6) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_fmed3f(d[i], 0.f, 6.f);
                }
                const t4 q = __builtin_convertvector(d, t4);         // the one rounding to the 16-bit type
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, q), osrc, ooff[j] == MBR_DEAD ? MBR_DEAD : opix + ooff[j], 0, 0);
                if (ooff[j] != MBR_DEAD) psum[j] += __builtin_convertvector(q, v4f);   // sums of what was stored
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) { ea[j] = eb[j]; eb[j] = ec[j]; }
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    row(N, 0, 0, xa, xb);
    row(N, 1, 0, xb, xa);
    int i = 0;
    for (; i + 1 < nout; i += 2) {
        row(Y, i + 2, yo0 + i, xa, xb);
        row(Y, i + 3, yo0 + i + 1, xb, xa);
    }
    if (i < nout) row(Y, i + 2, yo0 + i, xa, xb);
    if (a.sums == nullptr) return;   // uniform

    // ---- squeeze-excite sums: the 14 output columns of the strip meet by a fixed butterfly over the 16 lanes of a DPP row; lane
    // px == 0 of every row group writes its 4 channels of each tile into the (strip, segment) row
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        v4f s = psum[j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += __shfl_xor(s[q], o);
        if (px == 0 && t0 + j < a.T)
            *reinterpret_cast<v4f*>(a.sums + ((size_t)b * (a.strips * a.segs) + (size_t)(strip * a.segs + seg)) * a.ld_sums + 16 * (t0 + j) + 4 * mg) = s;
    }
}

template <class T16, int NKE, int NT>
static int launch_hwalkh(HwhArgs& a, int batch, hipStream_t s) {
    a.groups = (a.T + NT - 1) / NT;
    YR_REQUIRE(a.groups % 4 == 0, "head (walking form, 16-bit): %d tile groups are no multiple of the 4 waves of a workgroup", a.groups);
    const size_t lds = (size_t)a.T * MBR_TAB * 4;
    YR_REQUIRE(lds <= 64 * 1024, "head (walking form, 16-bit): %d channels exceed the LDS budget", a.T * 16);
    const bool pre = a.pre != nullptr, gated = a.gate != nullptr;
    YR_REQUIRE(!(pre && gated), "head (walking form, 16-bit): a gated source with a pre-BN addend is not built");
    static char nm[3][56];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "hwalkh_kernel<%s,%d,%d,0,0>", yr_dtype_name(yr_elem<T16>::dtype), NKE, NT) +
                              snprintf(nm[1], sizeof(nm[1]), "hwalkh_kernel<%s,%d,%d,1,0>", yr_dtype_name(yr_elem<T16>::dtype), NKE, NT) +
                              snprintf(nm[2], sizeof(nm[2]), "hwalkh_kernel<%s,%d,%d,0,1>", yr_dtype_name(yr_elem<T16>::dtype), NKE, NT);
    (void)nm_len;
    yr_note_kernel(nm[pre ? 1 : gated ? 2 : 0]);
    const dim3 grid((unsigned)(batch * a.strips * a.segs * (a.groups / 4)));
#define HWH_GO(P, G)                                                                                                       \
    {                                                                                                                      \
        auto kern = hwalkh_kernel<T16, NKE, NT, P, G>;                                                                     \
        static bool attr_set[64] = {};                                                                                     \
        int dev = 0;                                                                                                       \
        (void)hipGetDevice(&dev);                                                                                          \
        if (!attr_set[dev & 63] && lds > 48 * 1024) {                                                                      \
            YR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));   \
            attr_set[dev & 63] = true;                                                                                     \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);                                                              \
    }
    if (pre) HWH_GO(true, false) else if (gated) HWH_GO(false, true) else HWH_GO(false, false)
#undef HWH_GO
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T16>
static int launch_head_walk_h(HwhArgs& a, int nk, int batch, hipStream_t s) {
#define HWH_CASE(K) if (nk == K) return launch_hwalkh<T16, K, 2>(a, batch, s);
    HWH_CASE(1) HWH_CASE(2) HWH_CASE(3) HWH_CASE(4) HWH_CASE(5) HWH_CASE(6) HWH_CASE(7) HWH_CASE(8)
#undef HWH_CASE
    yr_set_error("head (walking form, 16-bit): %d chunks of 32 channels are not built", nk);
    return YR_ERR_ARG;
}

// op fields as YR_OP_HEAD (include/yoloret_hip.h) with k bit 6 and dtype = out_dtype = bf16 | f16: wgt = the 16-bit weight fragments
// [F / 16][NK][64 lanes][8] (compiler.head_pack16: the conv's weights as they are, no BN scale), scale = conv BN scale [F] float32,
// wgt2 = [F / 16][11][16] float32: depthwise taps x BN scale | depthwise BN shift | conv BN shift; se_reduced = yr_head_walk_rows(h, w).
int yr_launch_head_walk_h(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE((op.dtype == YR_BF16 || op.dtype == YR_F16) && op.out_dtype == op.dtype && op.out && op.wgt && op.wgt2 && op.scale, "head (walking form, 16-bit): bf16 / f16, non-null parameters");
    YR_REQUIRE((op.k & 0x3f) == 3 && op.stride == 1 && op.cout % 16 == 0 && op.out_ld % 4 == 0 && op.out_ld >= op.cout && ((uintptr_t)op.out % 8) == 0, "head (walking form, 16-bit): 3x3 stride 1, F a multiple of 16");
    YR_REQUIRE(op.gate_out == nullptr, "head (walking form, 16-bit): the squeeze-excite tail is not built (an SE_FC op finishes the sums)");
    const int act = (op.k >> 8) & 0xff;
    YR_REQUIRE(act == YR_ACT_RELU6 || act == YR_ACT_NONE, "head (walking form, 16-bit): conv activation ReLU6 or none");
    HwhArgs a;
    int nsrc = op.nsrc;
    a.pre = nullptr; a.pre_ld = 0;
    if (nsrc >= 2 && op.src[nsrc - 1].xform == YR_X_UP2_ADD) {
        const yr_src& ps = op.src[nsrc - 1];
        YR_REQUIRE(ps.dtype == YR_F32 && ps.ptr && ps.c == op.cout && ps.ld >= ps.c && ps.ld % 4 == 0 && ps.h * 2 == op.h && ps.w * 2 == op.w && ((uintptr_t)ps.ptr % 16) == 0,
                   "head (walking form, 16-bit): bad up2_add source");
        a.pre = (const float*)ps.ptr; a.pre_ld = ps.ld;
        --nsrc;
    }
    YR_REQUIRE(nsrc >= 1 && nsrc <= 3, "head (walking form, 16-bit): one to three k-space sources");
    int nk = 0;
    for (int i = 0; i < 3; ++i) { a.src[i] = nullptr; a.ld[i] = a.cs[i] = 0; }
    for (int i = 0; i < nsrc; ++i) {
        const yr_src& sr = op.src[i];
        YR_REQUIRE(sr.xform == YR_X_IDENTITY && sr.dtype == op.dtype && sr.ptr && sr.h == op.h && sr.w == op.w && sr.ld % 8 == 0 && sr.ld >= sr.c && ((uintptr_t)sr.ptr % 16) == 0,
                   "head (walking form, 16-bit): source %d must be an identity source of the op's type and the map's size (ld a multiple of 8)", i);
        YR_REQUIRE((uint64_t)batch * sr.h * sr.w * sr.ld * 2ull < 0x7e000000ull, "head (walking form, 16-bit): a source of %d images exceeds the 32-bit offsets of its loads", batch);
        a.src[i] = sr.ptr; a.ld[i] = sr.ld; a.cs[i] = sr.c;
        for (int j = 0; j < (sr.c + 31) / 32; ++j) {
            YR_REQUIRE(nk < HWH_MAXK, "head (walking form, 16-bit): more than %d chunks", HWH_MAXK);
            a.csrc[nk] = i; a.ckl[nk] = 32 * j; ++nk;
        }
    }
    for (int c = nk; c < HWH_MAXK; ++c) { a.csrc[c] = 0; a.ckl[c] = 0; }
    a.nsrc = nsrc;
    a.gate = nullptr; a.gate_ld = 0;
    if (op.res) {
        YR_REQUIRE(nsrc == 1 && op.res_ld >= ((op.src[0].c + 3) & ~3) && ((uintptr_t)op.res % 16) == 0 && op.res_ld % 4 == 0, "head (walking form, 16-bit): a gated source must be the single source");
        a.gate = (const float*)op.res; a.gate_ld = op.res_ld;
    }
    a.wa = op.wgt; a.wt = op.wgt2; a.scale = op.scale;
    a.out = op.out; a.ld_out = op.out_ld;
    a.H = op.h; a.W = op.w; a.T = op.cout / 16; a.F = op.cout; a.act = act; a.dw_act = op.act;
    a.strips = (op.w + 13) / 14;
    a.seg_rows = hw_seg_rows(op.h);
    a.segs = (op.h + a.seg_rows - 1) / a.seg_rows;
    if (a.pre) YR_REQUIRE(op.h % 2 == 0 && op.w % 2 == 0, "head (walking form, 16-bit): an up-sampled addend needs even dims");
    const int rows = a.strips * a.segs;
    if (op.gate) YR_REQUIRE(op.se_reduced == rows && op.gate_ld % 4 == 0 && op.gate_ld >= op.cout && ((uintptr_t)op.gate % 16) == 0,
                            "head (walking form, 16-bit): the squeeze-excite sums buffer must hold %d rows per image (se_reduced = %d)", rows, op.se_reduced);
    YR_REQUIRE((uint64_t)batch * op.h * op.w * op.out_ld * 2ull < 0x7e000000ull, "head (walking form, 16-bit): the output of %d images exceeds 32-bit offsets", batch);
    a.sums = const_cast<float*>(op.gate); a.ld_sums = op.gate_ld;
    return op.dtype == YR_BF16 ? launch_head_walk_h<yr_bf16>(a, nk, batch, s) : launch_head_walk_h<yr_f16>(a, nk, batch, s);
}
