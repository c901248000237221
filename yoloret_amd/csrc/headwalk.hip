// YR_OP_HEAD, WALKING form (k bit 6; round 5): the detection-head block's 1x1 conv + BN + ReLU6 -> depthwise 3x3 + BN + Swish ->
// squeeze-excite sums (reference code/yolo3/model.py:91-115, efficientnet.py:406-438,467-536) as mbe_kernel (mbr.hip) does the
// first two thirds of a MobileNetV2 block: a WAVE owns a strip of 16 input columns x a run of rows x NT output-channel tiles of 16
// and walks down the rows -
//   * the conv of one strip row on v_mfma_f32_16x16x32_f16 in the SPLIT form (float32 operands as two float16 planes, three MFMAs
//     per product), the weight fragments of its tiles STATIONARY in registers (cut by the plan: compiler.head_pack, BN scale folded
//     in), the pixel operand straight from global memory in operand layout, one row ahead;
//   * the MFMA result registers ARE the depthwise conv's input: horizontal taps by DPP row shifts, vertical taps = the last three
//     rows of the walk kept in registers; BN shift as first addend, Swish (hardware exp2 / rcp), one 16-byte store per lane and tile;
//   * no LDS but the tap table, no barrier inside the walk: the F-wide conv output never exists outside the register file.
// Round 5 measured why this form and not a GEMM tile: the LDS-tiled head kernels (headblock.hip) spend their time in per-workgroup
// latency chains (prologue, one DMA round trip per 32-channel chunk, LDS hand-over to the depthwise phase) at two workgroups per
// CU - 82-124 us per 26 x 26 / 52 x 52 head where the unfused chain took 90-112; a walking wave has ONE prologue per ~13 rows and
// its loads a row ahead.  What the head needs beyond mbe_kernel: the k space is a CONCATENATION of up to three identity sources
// (chunks of 32 channels per source, the plan's chunk table), an up-sampled pre-BN addend (a hoisted concat source: acc starts at
// shift + scale * pre), the SE gate of the source (bu3 reads td3's map through its gate), Swish, and the squeeze-excite sums: every
// wave adds up what it stores, per channel; a workgroup's four waves (same image, strip, segment - four tile groups) write their
// slices of the (strip, segment) row and arrive together; the workgroup that completes an image runs the FC pair (se_tail.h).
// Built for NKE <= 4 chunks of 32 channels (the weights must stay in registers, two cout tiles per wave): the 52 x 52 heads; the
// 26 x 26 ones (6-7 chunks: one tile per wave, 250 registers, the pixel operand cut 16 times - measured slower), 13 x 13 (K = 216 /
// 331, F = 512) and pooled sources stay on headblock.hip.
#include "mbr_common.h"
#include "se_tail.h"

#define HW_MAXK 7
#ifndef HW_NOHOIST_MIN_NKE
#define HW_NOHOIST_MIN_NKE 5   // from this many chunks on the tap table is read inside the walk (left alone hipcc hoists the reads and spills)
#endif
struct HwArgs {
    const float* src[3]; int ld[3]; int cs[3];   // k-space sources (identity): pointer, channel stride, channels
    int nsrc;
    int csrc[HW_MAXK], ckl[HW_MAXK];             // chunk -> source, first channel within the source
    const float* wa;     // expand A fragments [T][NKE][2 planes][64 lanes][8 halves] as float32 words (BN scale folded in)
    const float* wt;     // [T][11][16]: depthwise taps x BN scale | depthwise BN shift | conv BN shift
    const float* scale;  // conv BN scale [F] (the pre-BN addend is multiplied by it)
    const float* pre; int pre_ld;                // [B][H/2][W/2][pre_ld] or null
    const float* gate; int gate_ld;              // SE gate of the single source [B][gate_ld] or null
    float* out; int ld_out;
    int H, W, T, F, strips, segs, seg_rows, groups, nwaves, act, dw_act;
    float* sums; int ld_sums;                    // [B][strips * segs][ld_sums] or null
    SeTail se;
};

__device__ __forceinline__ float hw_swish(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f)); }

template <int NKE, int NT, bool PRE, bool GATED>
#ifndef HW_OCC
#define HW_OCC 2
#endif
__global__ __launch_bounds__(256, HW_OCC) void hwalk_kernel(HwArgs a) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float tab[];
    __shared__ unsigned se_flag;
    for (int i = threadIdx.x; i < a.T * MBR_TAB; i += 256) tab[i] = a.wt[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a workgroup = four consecutive tile groups of one (image, strip, segment) - groups % 4 == 0 (launcher)
    int gw = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x) * 4 + wave;
    const int g = gw % a.groups; gw /= a.groups;      // (the tile groups of one strip segment are neighbours: same pixels, L1 / L2)
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

    // ---- stationary: the weight planes of this wave's tiles, conv BN shift / scale, (the source's gate)
    mbs_u4 weh[NT][NKE], wem[NT][NKE];
    v4f se[NT], psc[PRE ? NT : 1];
    unsigned ooff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = min(t0 + j, a.T - 1);
        const mbs_u4* pe = reinterpret_cast<const mbs_u4*>(a.wa) + ((size_t)t * NKE) * 2 * 64 + lane;
#pragma unroll
        for (int c = 0; c < NKE; ++c) { weh[j][c] = pe[(2 * c) * 64]; wem[j][c] = pe[(2 * c + 1) * 64]; }
        se[j] = *reinterpret_cast<const v4f*>(a.wt + (size_t)t * MBR_TAB + 160 + 4 * mg);
        if constexpr (PRE) psc[j] = *reinterpret_cast<const v4f*>(a.scale + 16 * t + 4 * mg);
        ooff[j] = (t0 + j < a.T && out_lane) ? (16u * (t0 + j) + 4u * mg) * 4u : MBR_DEAD;
    }
    if constexpr (GATED) {
        // The SE gate of the source scales channel k of every pixel of image b - i.e. column k of the weights: W (g . x) = (W diag g) x.
        // The wave works on ONE image, so it scales its stationary fragments once (w = h + 2^-11 m back to float32, times the gate,
        // cut again: float32 rounding of the product, then the same 22-bit planes) instead of 8 multiplies per chunk and row plus
        // 8 registers per chunk for the gate.
#pragma unroll
        for (int c = 0; c < NKE; ++c) {
            const float* gp = a.gate + (size_t)b * a.gate_ld + a.ckl[c] + 8 * mg;
            const bool ok = a.ckl[c] + 8 * mg < a.cs[0], ok2 = ok && a.ckl[c] + 8 * mg + 4 < ((a.cs[0] + 3) & ~3);
            const v4f g0 = ok ? *reinterpret_cast<const v4f*>(gp) : (v4f){0.f, 0.f, 0.f, 0.f};
            const v4f g1 = ok2 ? *reinterpret_cast<const v4f*>(gp + 4) : (v4f){0.f, 0.f, 0.f, 0.f};
            const float gg[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                // (whole-vector casts: hipcc 7.2 folds __builtin_bit_cast of a vector ELEMENT inside an unrolled loop to element 0)
                typedef float hw_f8 __attribute__((ext_vector_type(8)));
                const hw_f8 hh = __builtin_convertvector(__builtin_bit_cast(mbs_h8, weh[j][c]), hw_f8), mm = __builtin_convertvector(__builtin_bit_cast(mbs_h8, wem[j][c]), hw_f8);
                const hw_f8 w8 = mm * 0.00048828125f + hh;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = w8[i] * gg[i];
                mbs_split8(v, weh[j][c], wem[j][c]);
            }
        }
    }
    // ---- descriptors: one per source (whole batch: offsets are 32-bit, the launcher checks the sizes), the addend, the output image
    const mbr_rsrc rs0 = mbr_make_rsrc(a.src[0], 0x7effffffu);
    const mbr_rsrc rs1 = mbr_make_rsrc(a.nsrc > 1 ? a.src[1] : a.src[0], 0x7effffffu);
    const mbr_rsrc rs2 = mbr_make_rsrc(a.nsrc > 2 ? a.src[2] : a.src[0], 0x7effffffu);
    const mbr_rsrc prs = mbr_make_rsrc(PRE ? a.pre + (size_t)b * (a.H >> 1) * (a.W >> 1) * a.pre_ld : a.src[0], PRE ? (unsigned)((a.H >> 1) * (a.W >> 1) * a.pre_ld) * 4u : 0u);
    const mbr_rsrc osrc = mbr_make_rsrc(a.out + (size_t)b * a.H * a.W * a.ld_out, (unsigned)(a.H * a.W * a.ld_out) * 4u);
    // per chunk: byte offset of this lane's 8 channels in its pixel (row offset added per row), or dead; the row pitch of its source
    unsigned xsoff[NKE], xsoff2[NKE], xpitch[NKE];
    int vcc[NKE];
#pragma unroll
    for (int c = 0; c < NKE; ++c) {
        const int s = a.csrc[c], kl = a.ckl[c] + 8 * mg;
        const int cs = s == 0 ? a.cs[0] : s == 1 ? a.cs[1] : a.cs[2], ld = s == 0 ? a.ld[0] : s == 1 ? a.ld[1] : a.ld[2];
        const int cq = (cs + 3) & ~3;
        const unsigned base = (unsigned)((b * a.H * a.W + xc) * ld + kl) * 4u;
        xsoff[c] = kl < cq ? base : MBR_DEAD;
        xsoff2[c] = kl + 4 < cq ? base + 16u : MBR_DEAD;
        xpitch[c] = (unsigned)(a.W * ld) * 4u;
        vcc[c] = cs - a.ckl[c];                 // valid channels of the chunk (>= 32: all)
    }
    const unsigned pcol = PRE ? (unsigned)((xc >> 1) * a.pre_ld + 4 * mg) * 4u : 0u;

    struct XRow { v4f m[2 * NKE]; v4f p[PRE ? NT : 1]; };
    XRow xa, xb;
    auto load_row = [&](XRow& x, int r) __attribute__((always_inline)) {
        const int rc = min(max(r, 0), a.H - 1);
#pragma unroll
        for (int c = 0; c < NKE; ++c) {
            const unsigned so = (unsigned)rc * xpitch[c];
            const int s = c == 0 ? 0 : a.csrc[c];     // (chunk 0 is the first source's: a compile-time fact the one-chunk kernels need -
            if (s == 0) {                              //  left with a run-time choice among descriptors they kept them in scratch memory)
                x.m[2 * c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs0, xsoff[c], so, 0));
                x.m[2 * c + 1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs0, xsoff2[c], so, 0));
            } else if (s == 1) {
                x.m[2 * c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs1, xsoff[c], so, 0));
                x.m[2 * c + 1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs1, xsoff2[c], so, 0));
            } else {
                x.m[2 * c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs2, xsoff[c], so, 0));
                x.m[2 * c + 1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs2, xsoff2[c], so, 0));
            }
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
    const float actmax = a.act == YR_ACT_RELU6 ? 6.f : 3.0e38f;     // conv activation: ReLU6, or none (lower bound below)
    const float actmin = a.act == YR_ACT_RELU6 ? 0.f : -3.0e38f;

    auto row = [&](auto emit_c, const int k, const int yo, const XRow& xc_, XRow& xn_) __attribute__((always_inline)) {
        constexpr bool EMIT = decltype(emit_c)::value;
        const int r = rbeg + k;
        load_row(xn_, r + 1);
        const float live = (r >= 0 && r < a.H) ? hi : 0.f;
        v4f ec[NT], e1[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            ec[j] = se[j];
            if constexpr (PRE) ec[j] = psc[j] * xc_.p[j] + ec[j];     // (acc + pre) * scale + shift, the scale folded into the weights
            e1[j] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < NKE; ++c) {
            float v[8] = {xc_.m[2 * c][0], xc_.m[2 * c][1], xc_.m[2 * c][2], xc_.m[2 * c][3], xc_.m[2 * c + 1][0], xc_.m[2 * c + 1][1], xc_.m[2 * c + 1][2], xc_.m[2 * c + 1][3]};
            if (vcc[c] < 32) {   // uniform: a source's last chunk - the lanes of a partial quad may hold anything (pad channels)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = 8 * mg + i < vcc[c] ? v[i] : 0.f;
            }
            mbs_u4 xh, xm;
            mbs_split8(v, xh, xm);
#pragma unroll
            for (int j = 0; j < NT; ++j) ec[j] = mbs_mfma(weh[j][c], xh, ec[j]);
#pragma unroll
            for (int j = 0; j < NT; ++j) e1[j] = mbs_mfma(weh[j][c], xm, e1[j]);
#pragma unroll
            for (int j = 0; j < NT; ++j) e1[j] = mbs_mfma(wem[j][c], xh, e1[j]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            ec[j] = __builtin_elementwise_fma(e1[j], (v4f){0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f}, ec[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) ec[j][i] = __builtin_amdgcn_fmed3f(ec[j][i], actmin, actmax) * live;
        }
        if constexpr (EMIT) {
            const unsigned opix = ((unsigned)yo * (unsigned)a.W + (unsigned)xo) * (unsigned)a.ld_out * 4u;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                unsigned toff = 0;
                if constexpr (NKE >= HW_NOHOIST_MIN_NKE) asm volatile("" : "+v"(toff));   // (keeps the tap reads inside the walk: headwalk_h.hip's note)
                const v4f* tb = reinterpret_cast<const v4f*>(tab + toff + min(t0 + j, a.T - 1) * MBR_TAB) + mg;
                v4f d = tb[36];
                mbr_dw_row(d, ea[j], tb[0], tb[4], tb[8]);
                mbr_dw_row(d, eb[j], tb[12], tb[16], tb[20]);
                mbr_dw_row(d, ec[j], tb[24], tb[28], tb[32]);
                if (a.dw_act == YR_ACT_SWISH) {   // uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = hw_swish(d[i]);
                } else if (a.dw_act == YR_ACT_RELU6) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_fmed3f(d[i], 0.f, 6.f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, d), osrc, ooff[j] == MBR_DEAD ? MBR_DEAD : opix + ooff[j], 0, 0);
                if (ooff[j] != MBR_DEAD) psum[j] += d;
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
    // px == 0 of every row group writes its 4 channels of each tile into the (strip, segment) row (write-through: se_tail.h)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        v4f s = psum[j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += __shfl_xor(s[q], o);
        if (px == 0 && t0 + j < a.T)
            yr_st_sums4(a.sums + ((size_t)b * (a.strips * a.segs) + (size_t)(strip * a.segs + seg)) * a.ld_sums + 16 * (t0 + j) + 4 * mg, a.se.sums != nullptr, s[0], s[1], s[2], s[3]);
    }
    yr_se_tail_arrive<256>(a.se, b, 1u, &se_flag, tab);   // (tab: the launcher sizes the dynamic LDS for the tail's scratch as well)
}

extern "C" int yr_head_walk_rows(int h, int w, int32_t* rows) {
    YR_REQUIRE(h > 0 && w > 0 && rows, "yr_head_walk_rows: bad arguments");
    const int sr = hw_seg_rows(h);
    *rows = ((w + 13) / 14) * ((h + sr - 1) / sr);
    return YR_OK;
}

template <int NKE, int NT>
static int launch_hwalk(HwArgs& a, int batch, hipStream_t s) {
    a.groups = (a.T + NT - 1) / NT;
    YR_REQUIRE(a.groups % 4 == 0, "head (walking form): %d tile groups are no multiple of the 4 waves of a workgroup", a.groups);
    a.nwaves = batch * a.strips * a.segs * a.groups;
    a.se.arrivals = (unsigned)(a.strips * a.segs * (a.groups / 4));
    size_t lds = (size_t)a.T * MBR_TAB * 4;
    if (a.se.sums) lds = lds > yr_se_tail_floats(a.se.C, a.se.R, 256) * 4 ? lds : yr_se_tail_floats(a.se.C, a.se.R, 256) * 4;
    YR_REQUIRE(lds <= 64 * 1024, "head (walking form): %d channels exceed the LDS budget", a.T * 16);
    const bool pre = a.pre != nullptr, gated = a.gate != nullptr;
    static char nm[4][48];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "hwalk_kernel<%d,%d,0,0>", NKE, NT) + snprintf(nm[1], sizeof(nm[1]), "hwalk_kernel<%d,%d,1,0>", NKE, NT) +
                              snprintf(nm[2], sizeof(nm[2]), "hwalk_kernel<%d,%d,0,1>", NKE, NT);
    (void)nm_len;
    yr_note_kernel(nm[pre ? 1 : gated ? 2 : 0]);
    YR_REQUIRE(!(pre && gated), "head (walking form): a gated source with a pre-BN addend is not built");
    const dim3 grid((unsigned)(a.nwaves / 4));
#define HW_GO(P, G)                                                                                                        \
    {                                                                                                                      \
        auto kern = hwalk_kernel<NKE, NT, P, G>;                                                                           \
        static bool attr_set[16] = {};                                                                                     \
        int dev = 0;                                                                                                       \
        (void)hipGetDevice(&dev);                                                                                          \
        if (dev >= 0 && dev < 16 && !attr_set[dev]) {                                                                      \
            YR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));   \
            attr_set[dev] = true;                                                                                          \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);                                                              \
    }
    if (pre) HW_GO(true, false) else if (gated) HW_GO(false, true) else HW_GO(false, false)
#undef HW_GO
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// op fields as YR_OP_HEAD (headblock.hip) with k bit 6: wgt = the float16 planes of compiler.head_pack with the conv's BN scale
// folded in; scale = the conv's BN scale [F] (for the pre-BN addend); wgt2 = [T = F / 16][11][16]: depthwise taps x BN scale |
// depthwise BN shift | conv BN shift (YR_OP_MBR's table); se_reduced = yr_head_walk_rows(h, w).
int yr_launch_head_walk(const yr_op& op, int batch, hipStream_t s) {
    if (op.dtype != YR_F32) return yr_launch_head_walk_h(op, batch, s);   // the 16-bit plans' twin (headwalk_h.hip)
    YR_REQUIRE(op.dtype == YR_F32 && op.out_dtype == YR_F32 && op.out && op.wgt && op.wgt2 && op.scale, "head (walking form): float32, non-null parameters");
    YR_REQUIRE((op.k & 0x3f) == 3 && op.stride == 1 && op.cout % 16 == 0 && op.out_ld % 4 == 0 && op.out_ld >= op.cout, "head (walking form): 3x3 stride 1, F a multiple of 16");
    const int act = (op.k >> 8) & 0xff;
    YR_REQUIRE(act == YR_ACT_RELU6 || act == YR_ACT_NONE, "head (walking form): conv activation ReLU6 or none");
    HwArgs a;
    int nsrc = op.nsrc;
    a.pre = nullptr; a.pre_ld = 0;
    if (nsrc >= 2 && op.src[nsrc - 1].xform == YR_X_UP2_ADD) {
        const yr_src& ps = op.src[nsrc - 1];
        YR_REQUIRE(ps.dtype == YR_F32 && ps.ptr && ps.c == op.cout && ps.ld >= ps.c && ps.ld % 4 == 0 && ps.h * 2 == op.h && ps.w * 2 == op.w && ((uintptr_t)ps.ptr % 16) == 0,
                   "head (walking form): bad up2_add source");
        a.pre = (const float*)ps.ptr; a.pre_ld = ps.ld;
        --nsrc;
    }
    YR_REQUIRE(nsrc >= 1 && nsrc <= 3, "head (walking form): one to three k-space sources");
    int nk = 0;
    for (int i = 0; i < 3; ++i) { a.src[i] = nullptr; a.ld[i] = a.cs[i] = 0; }
    for (int i = 0; i < nsrc; ++i) {
        const yr_src& sr = op.src[i];
        YR_REQUIRE(sr.xform == YR_X_IDENTITY && sr.dtype == YR_F32 && sr.ptr && sr.h == op.h && sr.w == op.w && sr.ld % 4 == 0 && sr.ld >= sr.c && ((uintptr_t)sr.ptr % 16) == 0,
                   "head (walking form): source %d must be a float32 identity source of the map's size", i);
        YR_REQUIRE((uint64_t)batch * sr.h * sr.w * sr.ld * 4ull < 0x7e000000ull, "head (walking form): a source of %d images exceeds the 32-bit offsets of its loads", batch);
        a.src[i] = (const float*)sr.ptr; a.ld[i] = sr.ld; a.cs[i] = sr.c;
        for (int j = 0; j < (sr.c + 31) / 32; ++j) {
            YR_REQUIRE(nk < HW_MAXK, "head (walking form): more than %d chunks", HW_MAXK);
            a.csrc[nk] = i; a.ckl[nk] = 32 * j; ++nk;
        }
    }
    for (int c = nk; c < HW_MAXK; ++c) { a.csrc[c] = 0; a.ckl[c] = 0; }
    a.nsrc = nsrc;
    a.gate = nullptr; a.gate_ld = 0;
    if (op.res) {
        YR_REQUIRE(nsrc == 1 && op.res_ld >= ((op.src[0].c + 3) & ~3), "head (walking form): a gated source must be the single source");
        a.gate = (const float*)op.res; a.gate_ld = op.res_ld;
    }
    a.wa = op.wgt; a.wt = op.wgt2; a.scale = op.scale;
    a.out = (float*)op.out; a.ld_out = op.out_ld;
    a.H = op.h; a.W = op.w; a.T = op.cout / 16; a.F = op.cout; a.act = act; a.dw_act = op.act;
    a.strips = (op.w + 13) / 14;
    a.seg_rows = hw_seg_rows(op.h);
    a.segs = (op.h + a.seg_rows - 1) / a.seg_rows;
    if (a.pre) YR_REQUIRE(op.h % 2 == 0 && op.w % 2 == 0, "head (walking form): an up-sampled addend needs even dims");
    const int rows = a.strips * a.segs;
    if (op.gate) YR_REQUIRE(op.se_reduced == rows && op.gate_ld % 4 == 0 && op.gate_ld >= op.cout && ((uintptr_t)op.gate % 16) == 0,
                            "head (walking form): the squeeze-excite sums buffer must hold %d rows per image (se_reduced = %d)", rows, op.se_reduced);
    const int rc = yr_make_se_tail(op, rows, &a.se);
    if (rc) return rc;
    a.sums = const_cast<float*>(op.gate); a.ld_sums = op.gate_ld;
#define HW_CASE(K, T2) if (nk == K) return launch_hwalk<K, T2>(a, batch, s);
    HW_CASE(1, 2) HW_CASE(2, 2) HW_CASE(3, 2) HW_CASE(4, 2) HW_CASE(5, 1) HW_CASE(6, 1) HW_CASE(7, 1)   // (5 .. 7 chunks: one tile per wave; the compiler takes them when HEAD_WALK_MAX_NK allows)
#undef HW_CASE
    yr_set_error("head (walking form): %d chunks of 32 channels are not built", nk);
    return YR_ERR_ARG;
}
