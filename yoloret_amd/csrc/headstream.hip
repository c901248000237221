// YR_OP_HEAD, WEIGHT-STREAMING form (k bits 5 and 6; round 6): the detection-head block's Concatenate + 1x1 conv + BN + ReLU6 ->
// depthwise 3x3 + BN + Swish -> squeeze-excite sums (reference code/yolo3/model.py:91-115, efficientnet.py:406-438,467-536) in mbk.hip's
// formulation: the PIXELS are stationary, the weights stream.
//   * a workgroup owns a strip of 16 columns x NW * ROWS rows of the map; a wave owns ROWS (1 | 2) of those rows for the whole kernel: the
//     float16 planes of its pixels over the WHOLE k space (up to 11 chunks of 32 channels: the concatenation of up to three sources,
//     gathered ONCE - identity or 2 x 2 max-pooled, times the source's SE gate) stay in registers;
//   * the conv's output channels stream past them in PAIRS OF TILES (32 channels): per pair, the weight planes of all k chunks arrive
//     through LDS-direct buffer loads (two chunk buffers: pair q + 2 is in flight while pair q + 1 is multiplied); the depthwise
//     tables of all tiles and the conv's BN scale sit in LDS from the prologue;
//   * conv (three float16-plane MFMAs per float32 product, BN scale folded into the planes, the up-sampled pre-BN addend of a hoisted
//     low-resolution conv joins with the shift) -> ReLU6 on the result registers -> the neighbour waves' rows through LDS (one barrier
//     per pair) -> depthwise taps by DPP row shifts -> Swish -> one 16-byte store per lane, tile and row; the per-channel sums of what
//     a wave stored leave as its row of the squeeze-excite sums buffer;
//   * the loop body is ONE hand-interleaved stream (mbk.hip): the conv MFMAs of pair q + 1 with the taps, the Swish and the stores of
//     pair q in their shadow.
// Why: the LDS-tiled head kernels (headblock.hip) spend their time in per-workgroup latency chains - 69 / 73 us for the 26 x 26 heads
// of MobileNetV2 x0.75 @416 at 64 images, whose arithmetic is worth 5 us -, and the walking form (headwalk.hip) needs the weights of its
// tiles in registers (at most 4 chunks).  Here a pair costs its MFMAs.  The sums are grouped by (strip, segment, wave): by the SHAPE.
#include "mbr_common.h"

#define HS_MAXK 11
struct HsArgs {
    const float* src[3]; int ld[3]; int cs[3]; int pool[3];   // k-space sources: pointer, channel stride, channels, 1 = 2 x 2 max of a map twice the size
    int nsrc;
    int csrc[HS_MAXK], ckl[HS_MAXK];             // chunk -> source, first channel within the source
    const float* wa;     // [T][NK][2 planes][64 lanes][8 halves] as float32 words (BN scale folded in): compiler.head_pack
    const float* wt;     // [T][11][16]: depthwise taps x BN scale | depthwise BN shift | conv BN shift
    const float* scale;  // conv BN scale [F] (the pre-BN addend is multiplied by it)
    const float* pre; int pre_ld;                // [B][H/2][W/2][pre_ld] or null
    const float* gate; int gate_ld;              // SE gate of the single source [B][gate_ld] or null
    float* out; int ld_out;
    float* sums; int ld_sums;                    // [B][strips * segs * NW][ld_sums] or null
    int H, W, T, F, strips, segs, act, dw_act;
    int csplit;          // workgroups that share one (image, strip, segment): each takes a contiguous run of the tile pairs (a launch-geometry choice: no result depends on it)
    unsigned wa_bytes, wt_bytes;
};

typedef __attribute__((address_space(3))) void* hs_lds_ptr;
#ifdef HS_TIMING      // (python tools/relink.py headstream.hip -DHS_TIMING; tools/hs_timing.py: per-phase shader-clock totals of every wave, written behind the output's last image)
#define HS_T(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define HS_T(i)
#endif

__device__ __forceinline__ float hs_swish(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f)); }

template <int NK, int ROWS, int NW, bool PRE>
__global__ __launch_bounds__(64 * NW) void hstream_kernel(HsArgs a) {
    constexpr int CHB = 2 * NK * 2048;           // bytes of one pair's weight planes
    constexpr int NPIECE = CHB / 1024, NR = NW * ROWS;
    constexpr int NPR = ROWS > 1 ? 2 : 1;        // rows a wave parks per tile: its first and (if it is another one) its last
    constexpr int XCB = (NW + 2) * NPR * 2 * 1024;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];     // [2][CHB] weight planes | [2][NW + 2][NPR][2][64] v4f parked rows | [T][11][16] tables | [F] conv BN scale
    char* const xch = lds_raw + 2 * CHB;
    float* const tabs = reinterpret_cast<float*>(xch + 2 * XCB);
    float* const lsc = tabs + ((a.T * MBR_TAB * 4 + 1023) & ~1023) / 4;      // (both land as whole 1 KB pieces)
    const int lane = threadIdx.x & 63, px = lane & 15, mg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int part = bid % a.csplit; bid /= a.csplit;
    const int seg = bid % a.segs; bid /= a.segs;
    const int strip = bid % a.strips;
    const int b = bid / a.strips;
    const int Q0 = part * (a.T / 2) / a.csplit, NQ = (part + 1) * (a.T / 2) / a.csplit;     // this workgroup's tile pairs [Q0, NQ)
#ifdef HS_TIMING
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif

    // ---- rows of this wave (mbk.hip's rule), columns of this lane
    const int ri0 = seg * (NR - 2), out0 = seg == 0 ? 0 : ri0 + 1;
    int rin[ROWS];
    bool emit[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int r = ri0 + ROWS * w + i;
        rin[i] = r;
        emit[i] = r >= out0 && r < a.H && (r + 1 < ri0 + NR || r + 1 >= a.H);
    }
    const int xin = 14 * strip - 1 + px;
    const int xc = min(max(xin, 0), a.W - 1);
    const bool col_in = xin >= 0 && xin < a.W;
    const int xo = 14 * strip + px - 1;
    const bool out_lane = px >= 1 && px <= 14 && xo < a.W;

    const mbr_rsrc wsrc = mbr_make_rsrc(a.wa, a.wa_bytes);
    auto issue_chunk = [&](const int q) {
        char* dst = lds_raw + (q & 1) * CHB;
        const int qs = min(q, NQ - 1);
#pragma unroll
        for (int u = 0; u < (NPIECE + NW - 1) / NW; ++u) {
            const int p = u * NW + w;
            if (NPIECE % NW == 0 || p < NPIECE)      // (wave-uniform)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (hs_lds_ptr)(dst + p * 1024), 16, (unsigned)(qs * CHB + p * 1024 + lane * 16), 0, 0, 0);
        }
    };
    issue_chunk(Q0);
    issue_chunk(Q0 + 1);
    // the tables of all tiles and the conv's BN scale: LDS-direct too (plain loads were 6 - 11 dependent round trips per thread: a quarter of
    // the kernel's cycles, tools/hs_timing.py); the zero rows above the first / below the last wave
    {
        const mbr_rsrc tsrc = mbr_make_rsrc(a.wt, a.wt_bytes), csrc_ = mbr_make_rsrc(a.scale, (unsigned)a.F * 4u);
        const int ntp = (a.T * MBR_TAB * 4 + 1023) / 1024, ncp = (a.F * 4 + 1023) / 1024;      // (reads beyond a descriptor return zeros)
        for (int p = w; p < ntp; p += NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(tsrc, (hs_lds_ptr)(reinterpret_cast<char*>(tabs) + p * 1024), 16, (unsigned)(p * 1024 + lane * 16), 0, 0, 0);
        for (int p = w; p < ncp; p += NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(csrc_, (hs_lds_ptr)(reinterpret_cast<char*>(lsc) + p * 1024), 16, (unsigned)(p * 1024 + lane * 16), 0, 0, 0);
    }
    HS_T(8)
    if (w < 2) {
#pragma unroll
        for (int u = 0; u < 2 * NPR * 2; ++u)
            reinterpret_cast<v4f*>(xch + (u / (NPR * 2)) * XCB + (w * (NW + 1) * NPR * 2 + u % (NPR * 2)) * 1024)[lane] = (v4f){0.f, 0.f, 0.f, 0.f};
    }

    // ---- the wave's pixels over the whole k space: gathered once, gated, cut into float16 planes.  ALL loads of a round go out
    // before anything is used (one round trip for the identity sources; three more rounds for the other pixels of pooled sources)
    const mbr_rsrc rs0 = mbr_make_rsrc(a.src[0], 0x7effffffu);
    const mbr_rsrc rs1 = mbr_make_rsrc(a.nsrc > 1 ? a.src[1] : a.src[0], 0x7effffffu);
    const mbr_rsrc rs2 = mbr_make_rsrc(a.nsrc > 2 ? a.src[2] : a.src[0], 0x7effffffu);
    mbs_u4 xh[ROWS][NK], xm[ROWS][NK];
    float live[ROWS];
    {
        // groups of at most 7 chunks of one row at a time: their raw pixels next to the planes already cut must fit the registers
        constexpr int GC = NK <= 7 ? NK : (NK + 1) / 2, NG = (NK + GC - 1) / GC;
        const bool any_pool = (a.pool[0] | a.pool[1] | a.pool[2]) != 0;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const bool row_in = rin[i] >= 0 && rin[i] < a.H;
            live[i] = row_in && col_in ? 1.f : 0.f;
            const int rc = min(max(rin[i], 0), a.H - 1);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                v4f lo[GC], hi[GC];
                auto fetch = [&](const int t, const bool first) {      // pixel t of a pooled source's 2 x 2 window (t = 0: every source)
#pragma unroll
                    for (int cc = 0; cc < GC; ++cc) {
                        const int c = g * GC + cc;
                        if (c >= NK) continue;
                        const int s = c == 0 ? 0 : a.csrc[c], kl = a.ckl[c] + 8 * mg;
                        const int cs = s == 0 ? a.cs[0] : s == 1 ? a.cs[1] : a.cs[2], ld = s == 0 ? a.ld[0] : s == 1 ? a.ld[1] : a.ld[2];
                        const int pool = s == 0 ? a.pool[0] : s == 1 ? a.pool[1] : a.pool[2];
                        const int cq = (cs + 3) & ~3;
                        const int sw = pool ? 2 * a.W : a.W, sh = pool ? 2 * a.H : a.H;
                        const int yy = pool ? 2 * rc + (t >> 1) : rc, xx = pool ? 2 * xc + (t & 1) : xc;
                        const unsigned base = (unsigned)(((b * sh + yy) * sw + xx) * ld + kl) * 4u;
                        const bool want = first || pool;
                        const unsigned o0 = want && kl < cq ? base : MBR_DEAD, o1 = want && kl + 4 < cq ? base + 16u : MBR_DEAD;
                        v4f l2, h2;
                        if (s == 0) { l2 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs0, o0, 0, 0)); h2 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs0, o1, 0, 0)); }
                        else if (s == 1) { l2 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs1, o0, 0, 0)); h2 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs1, o1, 0, 0)); }
                        else { l2 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs2, o0, 0, 0)); h2 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs2, o1, 0, 0)); }
                        if (first) { lo[cc] = l2; hi[cc] = h2; }
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { lo[cc][e] = pool ? fmaxf(lo[cc][e], l2[e]) : lo[cc][e]; hi[cc][e] = pool ? fmaxf(hi[cc][e], h2[e]) : hi[cc][e]; }
                        }
                    }
                };
                fetch(0, true);
                if (any_pool) {      // (kernel-uniform)
                    fetch(1, false); fetch(2, false); fetch(3, false);
                }
#pragma unroll
                for (int cc = 0; cc < GC; ++cc) {
                    const int c = g * GC + cc;
                    if (c >= NK) continue;
                    const int s = c == 0 ? 0 : a.csrc[c], kl = a.ckl[c] + 8 * mg;
                    const int cs = s == 0 ? a.cs[0] : s == 1 ? a.cs[1] : a.cs[2];
                    float v[8] = {lo[cc][0], lo[cc][1], lo[cc][2], lo[cc][3], hi[cc][0], hi[cc][1], hi[cc][2], hi[cc][3]};
                    const int vc = cs - a.ckl[c];          // valid channels of the chunk (>= 32: all); the lanes of a partial quad may hold anything
                    if (a.gate != nullptr) {               // (uniform) the SE gate of the single source
                        const float* gp = a.gate + (size_t)b * a.gate_ld + kl;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = 8 * mg + e < vc ? v[e] * gp[e] : 0.f;
                    } else if (vc < 32) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = 8 * mg + e < vc ? v[e] : 0.f;
                    }
                    mbs_split8(v, xh[i][c], xm[i][c]);
                }
            }
        }
    }
    HS_T(9)
    const mbr_rsrc prs = mbr_make_rsrc(PRE ? a.pre + (size_t)b * (a.H >> 1) * (a.W >> 1) * a.pre_ld : a.src[0], PRE ? (unsigned)((a.H >> 1) * (a.W >> 1) * a.pre_ld) * 4u : 0u);
    const mbr_rsrc osrc = mbr_make_rsrc(a.out + (size_t)b * a.H * a.W * a.ld_out, (unsigned)(a.H * a.W * a.ld_out) * 4u);
    const mbr_rsrc ssrc = mbr_make_rsrc(a.sums != nullptr ? a.sums + (size_t)b * (a.strips * a.segs * NW) * a.ld_sums : a.src[0],
                                        a.sums != nullptr ? (unsigned)(a.strips * a.segs * NW * a.ld_sums) * 4u : 0u);
    unsigned ppix[ROWS], opix[ROWS];       // byte offsets of the lane's pixel in the addend map / the output map (dead: not stored)
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int rc = min(max(rin[i], 0), a.H - 1);
        ppix[i] = PRE ? (unsigned)(((rc >> 1) * (a.W >> 1) + (xc >> 1)) * a.pre_ld + 4 * mg) * 4u : 0u;
        opix[i] = emit[i] && out_lane ? (unsigned)((rc * a.W + xo) * a.ld_out + 4 * mg) * 4u : MBR_DEAD;
    }
    const unsigned srow = (unsigned)(((strip * a.segs + seg) * NW + w) * a.ld_sums + 4 * mg) * 4u;
    const float actmax = a.act == YR_ACT_RELU6 ? 6.f : 3.0e38f;     // conv activation: ReLU6, or none
    const float actmin = a.act == YR_ACT_RELU6 ? 0.f : -3.0e38f;
    const v4f k11 = (v4f){0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f};

    // ---- conv of tile pair q out of chunk buffer cb: BN shift (+ scale x the up-sampled addend) as the accumulators' first value
    // the up-sampled addend of the pair being multiplied: loaded at the top of the turn, added behind its MFMAs (the widest two-row
    // form has no registers for that and adds at once)
    constexpr bool LATE = PRE && !(NK >= 7 && ROWS == 2);
    v4f pv[LATE ? ROWS : 1][2];
    int pvq = 0;
    auto conv_init = [&](const int q, v4f (&en)[ROWS][2]) {
        pvq = q;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = 2 * q + j;
            const v4f se = reinterpret_cast<const v4f*>(tabs + t * MBR_TAB)[40 + mg];
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                en[i][j] = se;
                if constexpr (LATE) pv[i][j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(prs, ppix[i] + 64u * t, 0, 0));
                else if constexpr (PRE)
                    en[i][j] = reinterpret_cast<const v4f*>(lsc + 16 * t)[mg] * __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(prs, ppix[i] + 64u * t, 0, 0)) + se;
            }
        }
    };
    auto conv_finish = [&](v4f (&en)[ROWS][2]) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (LATE) en[i][j] = reinterpret_cast<const v4f*>(lsc + 16 * (2 * pvq + j))[mg] * pv[i][j] + en[i][j];     // (acc + pre) * scale + shift, the scale folded into the planes
#pragma unroll
                for (int s = 0; s < 4; ++s) en[i][j][s] = __builtin_amdgcn_fmed3f(en[i][j][s], actmin, actmax) * live[i];
            }
    };
    auto park_rows = [&](const int q, const v4f (&ec)[ROWS][2]) {
        v4f* park = reinterpret_cast<v4f*>(xch + (q & 1) * XCB) + lane;        // [NW + 2][NPR: first | last row][2 tiles][64]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            park[(((w + 1) * NPR + 0) * 2 + j) * 64] = ec[0][j];
            if (NPR > 1) park[(((w + 1) * NPR + 1) * 2 + j) * 64] = ec[ROWS - 1][j];
        }
    };
    // ---- the loop body: the depthwise stage, Swish, stores and sums of pair q with the conv MFMAs of pair q + 1 (mbk.hip's slices)
    auto step = [&](auto conv_c, const int q, const char* cbn, const v4f (&ec)[ROWS][2], v4f (&en)[ROWS][2]) {
        constexpr bool CONV = decltype(conv_c)::value;      // false: the last pair - nothing left to multiply
        constexpr int MPG = 3 * ROWS, NM = 2 * NK * MPG;                      // conv: MFMAs per (tile, k chunk) group, MFMAs
        constexpr int PPG = 3 * ROWS, NDG = 6, NV = NDG * PPG;                // depthwise: parts per (tile, tap row) group, groups, parts
        constexpr int NX = 2 * ROWS * 2 + 2;                                  // then: Swish + store per (row, tile, half), the sums per tile
        constexpr int NS = (CONV && NM > NV + NX) ? NM : NV + NX;
        const u4* fe = reinterpret_cast<const u4*>(cbn) + lane;               // [2 tiles][NK][2 planes][64]
        const v4f* tbq = reinterpret_cast<const v4f*>(tabs + 2 * q * MBR_TAB) + mg;       // tile j: tbq + 44 j
        const v4f* park = reinterpret_cast<const v4f*>(xch + (q & 1) * XCB) + lane;
        v4f above[2], below[2], d[ROWS][2], e1[ROWS];
        u4 fr[2][2];
        v4f tp[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            above[j] = park[((w * NPR + NPR - 1) * 2 + j) * 64];               // slot w = wave w - 1
            below[j] = park[(((w + 2) * NPR + 0) * 2 + j) * 64];
            const v4f sh = tbq[44 * j + 36];
#pragma unroll
            for (int i = 0; i < ROWS; ++i) d[i][j] = sh;
        }
        if constexpr (CONV) { fr[0][0] = fe[0]; fr[0][1] = fe[64]; }
        tp[0][0] = tbq[0]; tp[0][1] = tbq[4]; tp[0][2] = tbq[8];
        __builtin_amdgcn_sched_barrier(0);
        HS_T(3)
        mbk_for<NS>([&](auto SS) {
            constexpr int sl = decltype(SS)::value;
            if constexpr (CONV && sl < NM) {
                constexpr int g = sl / MPG, r = sl % MPG, kind = r / ROWS, i = r % ROWS, j = g / NK, c = g % NK;
                if constexpr (r == 0 && g + 1 < 2 * NK) {
                    fr[(g + 1) & 1][0] = fe[((g + 1) * 2 + 0) * 64];
                    fr[(g + 1) & 1][1] = fe[((g + 1) * 2 + 1) * 64];
                }
                if constexpr (c == 0 && kind == 1) e1[i] = (v4f){0.f, 0.f, 0.f, 0.f};
                if constexpr (kind == 0) en[i][j] = mbs_mfma(fr[g & 1][0], xh[i][c], en[i][j]);
                else if constexpr (kind == 1) e1[i] = mbs_mfma(fr[g & 1][0], xm[i][c], e1[i]);
                else e1[i] = mbs_mfma(fr[g & 1][1], xh[i][c], e1[i]);
                if constexpr (c == NK - 1 && kind == 2) en[i][j] = __builtin_elementwise_fma(e1[i], k11, en[i][j]);
            }
            if constexpr (sl < NV) {
                constexpr int dg = sl / PPG, pr = sl % PPG, i = pr / 3, part = pr % 3, j = dg / 3, ky = dg % 3;
                if constexpr (pr == 0 && dg + 1 < NDG) {
                    constexpr int j1 = (dg + 1) / 3, ky1 = (dg + 1) % 3;
                    tp[(dg + 1) & 1][0] = tbq[44 * j1 + 12 * ky1];
                    tp[(dg + 1) & 1][1] = tbq[44 * j1 + 12 * ky1 + 4];
                    tp[(dg + 1) & 1][2] = tbq[44 * j1 + 12 * ky1 + 8];
                }
                constexpr int rr = i + ky - 1;      // the wave's row the tap row reads (-1: above, ROWS: below)
                mbk_dw_part(part, d[i][j], rr < 0 ? above[j] : rr >= ROWS ? below[j] : ec[rr < 0 ? 0 : rr >= ROWS ? 0 : rr][j], tp[dg & 1][part == 0 ? 1 : part == 1 ? 0 : 2]);
            } else if constexpr (sl < NV + 2 * ROWS * 2) {
                // Swish of half a tile row (two values), the store behind the second half
                constexpr int x = sl - NV, i = x / 4, j = (x / 2) % 2, hf = x % 2;
                if (a.dw_act == YR_ACT_SWISH) {   // uniform
                    d[i][j][2 * hf] = hs_swish(d[i][j][2 * hf]);
                    d[i][j][2 * hf + 1] = hs_swish(d[i][j][2 * hf + 1]);
                } else if (a.dw_act == YR_ACT_RELU6) {
                    d[i][j][2 * hf] = __builtin_amdgcn_fmed3f(d[i][j][2 * hf], 0.f, 6.f);
                    d[i][j][2 * hf + 1] = __builtin_amdgcn_fmed3f(d[i][j][2 * hf + 1], 0.f, 6.f);
                }
                if constexpr (hf == 1)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, d[i][j]), osrc, opix[i] == MBR_DEAD ? MBR_DEAD : opix[i] + 64u * (2 * q + j), 0, 0);
            } else if constexpr (sl < NV + NX) {
                // the squeeze-excite sums of tile j: what this wave stored, over its rows and the strip's 14 columns (fixed order)
                constexpr int j = sl - NV - 2 * ROWS * 2;
                v4f sacc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < ROWS; ++i)
                    if (opix[i] != MBR_DEAD) sacc += d[i][j];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float t = sacc[s];
                    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                        "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                        "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(t));
                    sacc[s] = t;
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, sacc), ssrc, (a.sums != nullptr && px == 0) ? srow + 64u * (2 * q + j) : MBR_DEAD, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        HS_T(4)
    };

    // ONE barrier per tile pair: behind barrier q every wave has parked its rows of pair q and the planes of pair q + 1 have landed.
    v4f ec[ROWS][2], en[ROWS][2];
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    HS_T(10)
    {   // the first pair's conv: plain order (once)
        conv_init(Q0, ec);
        const u4* fe = reinterpret_cast<const u4*>(lds_raw + (Q0 & 1) * CHB) + lane;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            v4f e1[ROWS];
#pragma unroll
            for (int i = 0; i < ROWS; ++i) e1[i] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NK; ++c) {
                const u4 wh = fe[((j * NK + c) * 2 + 0) * 64], wm = fe[((j * NK + c) * 2 + 1) * 64];
#pragma unroll
                for (int i = 0; i < ROWS; ++i) ec[i][j] = mbs_mfma(wh, xh[i][c], ec[i][j]);
#pragma unroll
                for (int i = 0; i < ROWS; ++i) e1[i] = mbs_mfma(wh, xm[i][c], e1[i]);
#pragma unroll
                for (int i = 0; i < ROWS; ++i) e1[i] = mbs_mfma(wm, xh[i][c], e1[i]);
            }
#pragma unroll
            for (int i = 0; i < ROWS; ++i) ec[i][j] = __builtin_elementwise_fma(e1[i], k11, ec[i][j]);
        }
        conv_finish(ec);
        park_rows(Q0, ec);
    }
    HS_T(0)
    // (the wait in front of the barrier is COUNTED: behind the planes of pair q + 1 - the oldest operations in flight - only this turn's
    //  stores were issued, 2 ROWS of the map + 2 of the sums, and they may stay in flight; lgkmcnt(0): the parked rows are written)
    constexpr int NST = 2 * ROWS + 2;
    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    for (int q = Q0; q + 1 < NQ; ++q) {
        if (q == Q0) __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(NST | 0x0070);                // vmcnt(NST) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        HS_T(1)
        conv_init(q + 1, en);                    // (its addend loads go out BEFORE the next planes: they are waited for first)
        issue_chunk(q + 2);
        HS_T(2)                      // into the buffer pair q's planes were read from (beyond the last pair: the last one again)
        step(Y, q, lds_raw + ((q + 1) & 1) * CHB, ec, en);
        conv_finish(en);
        park_rows(q + 1, en);
#pragma unroll
        for (int i = 0; i < ROWS; ++i) { ec[i][0] = en[i][0]; ec[i][1] = en[i][1]; }
        HS_T(5)
    }
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    HS_T(6)
    step(N, NQ - 1, lds_raw, ec, en);
#ifdef HS_TIMING
    HS_T(7)
    __builtin_amdgcn_s_waitcnt(0x0f70);
    {
        unsigned tv = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) tv = lane == i ? (unsigned)tacc[i] : tv;
        if (lane < 12) reinterpret_cast<unsigned*>(a.out)[(size_t)(gridDim.x / (a.strips * a.segs * a.csplit)) * a.H * a.W * a.ld_out + ((size_t)blockIdx.x * NW + w) * 16 + lane] = tv;
    }
#endif
}

// rows of the squeeze-excite sums the weight-streaming form writes per image: one per (strip, segment, wave) - a function of the shape
static void hs_geometry(int h, int w, int* rows_per_wave, int* nw, int* strips, int* segs) {
    *rows_per_wave = h >= 20 ? 2 : 1;
    *nw = 8;
    const int nr = *nw * *rows_per_wave;
    *strips = (w + 13) / 14;
    *segs = h <= nr ? 1 : (h - nr + nr - 3) / (nr - 2) + 1;
}

extern "C" int yr_head_stream_rows(int h, int w, int32_t* rows) {
    YR_REQUIRE(h > 0 && w > 0 && rows, "yr_head_stream_rows: bad arguments");
    int rpw, nw, strips, segs;
    hs_geometry(h, w, &rpw, &nw, &strips, &segs);
    *rows = strips * segs * nw;
    return YR_OK;
}

template <int NK, int ROWS, int NW>
static int launch_hstream(HsArgs& a, int batch, hipStream_t s) {
    constexpr int CHB = 2 * NK * 2048, NPR = ROWS > 1 ? 2 : 1;
    const size_t lds = (size_t)2 * CHB + (size_t)2 * (NW + 2) * NPR * 2 * 1024 + (size_t)((a.T * MBR_TAB * 4 + 1023) & ~1023) + (size_t)((a.F * 4 + 1023) & ~1023);
    YR_REQUIRE(lds <= 160 * 1024, "head (weight-streaming form): %zu bytes of LDS", lds);
    const bool pre = a.pre != nullptr;
    static char nm[2][48];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "hstream_kernel<%d,%d,%d,0>", NK, ROWS, NW) + snprintf(nm[1], sizeof(nm[1]), "hstream_kernel<%d,%d,%d,1>", NK, ROWS, NW);
    (void)nm_len;
    yr_note_kernel(nm[pre ? 1 : 0]);
    // fewer workgroups than CUs (the 13 x 13 heads at 64 images: 128): up to four workgroups share an (image, strip, segment), each
    // streaming its own run of tile pairs past the same pixels (every output channel is still computed by exactly one wave)
    const int wgs = batch * a.strips * a.segs;
    a.csplit = wgs >= 200 ? 1 : (256 + wgs - 1) / wgs;
    if (a.csplit > 4) a.csplit = 4;
    while (a.csplit > 1 && (a.T / 2) / a.csplit < 2) --a.csplit;
    const dim3 grid((unsigned)(wgs * a.csplit));
#define HS_GO(P)                                                                                                           \
    {                                                                                                                      \
        auto kern = hstream_kernel<NK, ROWS, NW, P>;                                                                       \
        static bool attr_set[16] = {};                                                                                     \
        int dev = 0;                                                                                                       \
        (void)hipGetDevice(&dev);                                                                                          \
        if (dev >= 0 && dev < 16 && !attr_set[dev]) {                                                                      \
            YR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
            attr_set[dev] = true;                                                                                          \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, s, a);                                                          \
    }
    if (pre) HS_GO(true) else HS_GO(false)
#undef HS_GO
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// op fields as the walking form of YR_OP_HEAD (headwalk.hip) with k bit 5 as well: wgt = the float16 planes of compiler.head_pack with the
// conv's BN scale folded in ([T][NK][2 planes][64][8]); scale = the conv's BN scale [F]; wgt2 = [T = F / 16][11][16] (YR_OP_MBR's table);
// se_reduced = yr_head_stream_rows(h, w); sources: one to three float32 sources, identity or maxpool2, an up2_add addend last; res = the SE
// gate of the single source.  F a multiple of 32.
int yr_launch_head_stream(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.dtype == YR_F32 && op.out_dtype == YR_F32 && op.out && op.wgt && op.wgt2 && op.scale, "head (weight-streaming form): float32, non-null parameters");
    YR_REQUIRE((op.k & 0x1f) == 3 && op.stride == 1 && op.cout % 32 == 0 && op.out_ld % 4 == 0 && op.out_ld >= op.cout, "head (weight-streaming form): 3x3 stride 1, F a multiple of 32");
    YR_REQUIRE(op.gate_out == nullptr, "head (weight-streaming form): the squeeze-excite tail is not built for this form");
    const int act = (op.k >> 8) & 0xff;
    YR_REQUIRE(act == YR_ACT_RELU6 || act == YR_ACT_NONE, "head (weight-streaming form): conv activation ReLU6 or none");
    HsArgs a;
    int nsrc = op.nsrc;
    a.pre = nullptr; a.pre_ld = 0;
    if (nsrc >= 2 && op.src[nsrc - 1].xform == YR_X_UP2_ADD) {
        const yr_src& ps = op.src[nsrc - 1];
        YR_REQUIRE(ps.dtype == YR_F32 && ps.ptr && ps.c == op.cout && ps.ld >= ps.c && ps.ld % 4 == 0 && ps.h * 2 == op.h && ps.w * 2 == op.w && ((uintptr_t)ps.ptr % 16) == 0,
                   "head (weight-streaming form): bad up2_add source");
        a.pre = (const float*)ps.ptr; a.pre_ld = ps.ld;
        --nsrc;
    }
    YR_REQUIRE(nsrc >= 1 && nsrc <= 3, "head (weight-streaming form): one to three k-space sources");
    int nk = 0;
    for (int i = 0; i < 3; ++i) { a.src[i] = nullptr; a.ld[i] = a.cs[i] = a.pool[i] = 0; }
    for (int i = 0; i < nsrc; ++i) {
        const yr_src& sr = op.src[i];
        const bool pool = sr.xform == YR_X_MAXPOOL2;
        YR_REQUIRE((sr.xform == YR_X_IDENTITY || pool) && sr.dtype == YR_F32 && sr.ptr && sr.h == (pool ? 2 : 1) * op.h && sr.w == (pool ? 2 : 1) * op.w && sr.ld % 4 == 0 &&
                   sr.ld >= sr.c && ((uintptr_t)sr.ptr % 16) == 0, "head (weight-streaming form): source %d must be a float32 identity or 2 x 2 max-pooled source of the map's size", i);
        YR_REQUIRE((uint64_t)batch * sr.h * sr.w * sr.ld * 4ull < 0x7e000000ull, "head (weight-streaming form): a source of %d images exceeds the 32-bit offsets of its loads", batch);
        a.src[i] = (const float*)sr.ptr; a.ld[i] = sr.ld; a.cs[i] = sr.c; a.pool[i] = pool ? 1 : 0;
        for (int j = 0; j < (sr.c + 31) / 32; ++j) {
            YR_REQUIRE(nk < HS_MAXK, "head (weight-streaming form): more than %d chunks", HS_MAXK);
            a.csrc[nk] = i; a.ckl[nk] = 32 * j; ++nk;
        }
    }
    for (int c = nk; c < HS_MAXK; ++c) { a.csrc[c] = 0; a.ckl[c] = 0; }
    a.nsrc = nsrc;
    a.gate = nullptr; a.gate_ld = 0;
    if (op.res) {
        YR_REQUIRE(nsrc == 1 && op.res_ld >= ((op.src[0].c + 3) & ~3), "head (weight-streaming form): a gated source must be the single source");
        a.gate = (const float*)op.res; a.gate_ld = op.res_ld;
    }
    a.wa = op.wgt; a.wt = op.wgt2; a.scale = op.scale;
    a.out = (float*)op.out; a.ld_out = op.out_ld;
    a.H = op.h; a.W = op.w; a.T = op.cout / 16; a.F = op.cout; a.act = act; a.dw_act = op.act;
    a.wa_bytes = (unsigned)a.T * (unsigned)nk * 2048u; a.wt_bytes = (unsigned)a.T * MBR_TAB * 4u;
    int rpw, nw;
    hs_geometry(op.h, op.w, &rpw, &nw, &a.strips, &a.segs);
    if (a.pre) YR_REQUIRE(op.h % 2 == 0 && op.w % 2 == 0, "head (weight-streaming form): an up-sampled addend needs even dims");
    const int rows = a.strips * a.segs * nw;
    if (op.gate) YR_REQUIRE(op.se_reduced == rows && op.gate_ld % 4 == 0 && op.gate_ld >= op.cout && ((uintptr_t)op.gate % 16) == 0,
                            "head (weight-streaming form): the squeeze-excite sums buffer must hold %d rows per image (se_reduced = %d)", rows, op.se_reduced);
    a.sums = const_cast<float*>(op.gate); a.ld_sums = op.gate_ld;
#define HS_CASE(K) if (nk == K) return rpw == 2 ? launch_hstream<K, 2, 8>(a, batch, s) : launch_hstream<K, 1, 8>(a, batch, s);
    HS_CASE(1) HS_CASE(2) HS_CASE(3) HS_CASE(4) HS_CASE(5) HS_CASE(6) HS_CASE(7)
#undef HS_CASE
#define HS_CASE1(K) if (nk == K && rpw == 1) return launch_hstream<K, 1, 8>(a, batch, s);      // (two rows of 8+ chunks do not fit the register file)
    HS_CASE1(8) HS_CASE1(9) HS_CASE1(10) HS_CASE1(11)
#undef HS_CASE1
    yr_set_error("head (weight-streaming form): %d chunks of 32 channels at %d rows per wave are not built", nk, rpw);
    return YR_ERR_ARG;
}
