// YR_OP_HEAD (ABI 7): the first two thirds of a detection-head block in ONE launch, float32 plans -
//   Conv2D 1x1 + BN + ReLU6  ->  MBConvBlock's DepthwiseConv2D 3x3 + BN + Swish  [-> squeeze-excite sums + gate]
// (reference code/yolo3/model.py:91-115 make_last_layers_efficientnet_lite, called six times per model at :238,259,279,296,310,323;
// the MBConv block code/yolo3/efficientnet.py:467-536 with expand ratio 1, its SE block :406-438).
//
// Through round 4 this was three launches - the conv as a split-form pointwise GEMM writing the F-wide map (F = 128 @52x52,
// 256 @26x26, 512 @13x13), dw_kernel reading it back (every byte of that launch a round trip of a map the launch before had just
// written), se_fc - 14 % of the MobileNetV2 x0.75 step.  Here a workgroup owns a REGION of one image (rows x columns, chosen by
// yr_head_regions from the map's SHAPE only) and a slice of BN output channels:
//   1. the conv over the region PLUS its one-pixel halo (the GEMM's rows are the region's pixels instead of a run of the
//      flattened batch), float32 operands as two float16 planes on v_mfma_f32_16x16x32_f16 (mbr.hip "SPLIT form");
//   2. the finished conv values (pre-BN addend of a hoisted up-sampled source, BN, ReLU6) go to LDS as E[pixel][BN] - they never
//      reach HBM;
//   3. the depthwise 3x3 reads E: a lane owns (output column, channel quad) and walks down the rows of its row group with three
//      accumulators (input row r feeds output rows r - 1, r, r + 1: three 16-byte LDS reads per 36 multiply-adds), BN shift as the
//      first addend (the scale is folded into the taps), Swish, one 16-byte store;
//   4. its outputs' per-channel sums meet in LDS in a fixed order and leave as one row slice of the squeeze-excite sums; the
//      workgroup that completes an image runs the SE block's FC pair (se_tail.h).
// The halo is recomputed by the neighbouring region: the GEMM is cheap on the 16-bit matrix pipe, what the kernel must not do is
// move the F-wide map twice.
//
// Two GEMM front ends feed the same finish (head_finish):
//   * head2_kernel (k bit 7: the plan packed the weights as float16 planes in fragment order, compiler.head_pack): the
//     activations of a 32-channel chunk travel global memory -> LDS by LDS-DIRECT loads (no staging registers, no address
//     arithmetic per element, no ds_write pass: round 5's profile of the first version showed ~50 VALU instructions per 16 bytes
//     fetched and the k loop VALU-bound at 2 workgroups per CU), XOR-swizzled on the source side so the fragment reads are
//     conflict-free; every wave reads ITS pixel tiles' fragments as float32 and cuts the planes in registers (each element is cut
//     once per workgroup); the weight planes of the chunk arrive the same way, already cut; two stages, one barrier per chunk, the
//     next chunk in flight under the MFMAs.  Sources: identity / up-sampled (per-lane addresses), an SE-gated single source (gate
//     vector in LDS, applied to the fragment), the up-sampled pre-BN addend.
//   * head_kernel (pws_common.h's loop: register-staged gathers): kept for convs with a POOLED source (a maximum over 4 / 16
//     loads cannot be an LDS-direct load) - td1 (rfcr's map through a 2x2 max-pool).
#include "pws_common.h"
#include "se_tail.h"
#include <cstdlib>

struct HeadArgs {
    PwArgs p;           // the 1x1 convolution: S, wt, scale, shift, gate / gate_ld (SE gate on the single source), pre / pre_ld, act, H, W, N = F;
                        // out / out_ld = the DEPTHWISE output
    const float* dw;    // [10][ldf]: nine taps (ky, kx) x depthwise BN scale | depthwise BN shift
    int ldf, dw_act;
    int nsy, nsx;       // regions per image
    float* sums;        // squeeze-excite sums [B][nsy * nsx][ld_sums] (nullptr: none)
    int ld_sums;
    SeTail se;          // (se.sums == nullptr: an SE_FC op finishes the sums)
    int nk;             // head2: 32-channel chunks of the k space (sum over sources of ceil(c / 32))
    int exp;            // tools/head_probe.py (YR_HEAD_EXP; 0 in every plan): 1 skip the k loop, 2 skip the depthwise phase, 4 skip the pre-BN addend, 8 skip the sums
};

// Regions of a head-block launch (shape only - the squeeze-excite sums are grouped by region, so the choice must not depend on
// the batch or a tuner): the map is cut into nsy x nsx regions of balanced size; a region with its one-pixel halo (clipped to the
// map) must fit the BM = 192 GEMM rows of a workgroup.  Fewest regions wins.
#define HEAD_BM 192
static void head_geometry(int H, int W, int* nsy, int* nsx) {
    long best = -1;
    for (int sx = 1; sx <= 16 && sx <= W; ++sx) {
        const int cw = (W + sx - 1) / sx, rw = cw + (sx > 2 ? 2 : sx > 1 ? 1 : 0);
        for (int sy = 1; sy <= H; ++sy) {
            const int ch = (H + sy - 1) / sy, rh = ch + (sy > 2 ? 2 : sy > 1 ? 1 : 0);
            if ((rh < H ? rh : H) * (rw < W ? rw : W) > HEAD_BM) continue;
            const long cost = (long)sy * sx;
            if (best < 0 || cost < best) { best = cost; *nsy = sy; *nsx = sx; }
            break;   // (more row segments only cost more)
        }
    }
    if (best < 0) *nsy = *nsx = 0;
}

extern "C" int yr_head_regions(int h, int w, int32_t* nsy, int32_t* nsx) {
    YR_REQUIRE(h > 0 && w > 0 && nsy && nsx, "yr_head_regions: bad arguments");
    int sy = 0, sx = 0;
    head_geometry(h, w, &sy, &sx);
    YR_REQUIRE(sy > 0, "yr_head_regions: a %d x %d map has no region split that fits a workgroup", h, w);
    *nsy = sy; *nsx = sx;
    return YR_OK;
}

__device__ __forceinline__ float head_swish(float v) {   // hardware exp2 / rcp: about 1 ulp each (the op's bar is 5e-5)
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}

struct HeadRegion { int b, n0, iy, ix, y0, y1, x0, x1, ylo, xlo, RW, RP; };

__device__ __forceinline__ HeadRegion head_region(const HeadArgs& h, const int BN) {
    const PwArgs& a = h.p;
    HeadRegion r;
    // grid: cout slice fastest (the slices of a region run back to back on one XCD: its pixels stay in that L2), then region, then image
    const unsigned ntn = (a.N + BN - 1) / BN;
    unsigned L = yr_xcd_swizzle(blockIdx.x, gridDim.x);
    r.n0 = (int)(L % ntn) * BN; L /= ntn;
    r.ix = (int)(L % (unsigned)h.nsx); L /= (unsigned)h.nsx;
    r.iy = (int)(L % (unsigned)h.nsy);
    r.b = (int)(L / (unsigned)h.nsy);
    r.y0 = r.iy * a.H / h.nsy; r.y1 = (r.iy + 1) * a.H / h.nsy; r.x0 = r.ix * a.W / h.nsx; r.x1 = (r.ix + 1) * a.W / h.nsx;
    r.ylo = r.y0 > 0 ? r.y0 - 1 : 0; r.xlo = r.x0 > 0 ? r.x0 - 1 : 0;
    const int yhi = r.y1 < a.H ? r.y1 + 1 : a.H, xhi = r.x1 < a.W ? r.x1 + 1 : a.W;
    r.RW = xhi - r.xlo; r.RP = (yhi - r.ylo) * r.RW;   // the region with its halo: RP <= BM pixels (head_geometry)
    return r;
}

// Everything behind the GEMM: conv epilogue -> E, depthwise from E, stores, squeeze-excite sums + tail.  acc / ac1: the wave's
// PT x CT accumulator tiles (pixel tiles wave * PT .. + PT - 1); lds_raw: at least BM * (BN + 4) * 4 bytes, free (the caller's
// last LDS reads are behind a barrier); ss: conv BN scale | shift of the slice.
template <int NTH, int PT, int CT>
__device__ __forceinline__ void head_finish(const HeadArgs& h, const HeadRegion& R, f32x4 (&acc)[CT][PT], f32x4 (&ac1)[CT][PT], char* lds_raw, const float* ss, unsigned* flag) {
    constexpr int BN = 16 * CT, QN = BN / 4, ELD = BN + 4;   // floats per pixel row of E (16 bytes of padding)
    const PwArgs& a = h.p;
    float* E = reinterpret_cast<float*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = R.b, n0 = R.n0;

    // ---- conv epilogue: (pre-BN addend,) BN, activation -> E[pixel][BN].  A lane holds 4 consecutive couts of pixel li of each of
    // its tiles.  The addend's loads are all issued before the first is used (one exposed round trip, not PT * CT).
    {
        const int g = lane >> 4, li = lane & 15;
        const bool use_pre = a.pre != nullptr && !(h.exp & 4), conv_relu6 = a.act == YR_ACT_RELU6;
        f32x4 pq[PT][CT];
        if (use_pre) {   // uniform
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int m = (wave * PT + p) * 16 + li;
                const int mm = m < R.RP ? m : 0;
                const int ry = mm / R.RW, rx = mm - ry * R.RW;
                const float* pr = a.pre + (((size_t)(b * (a.H >> 1) + ((R.ylo + ry) >> 1))) * (a.W >> 1) + ((R.xlo + rx) >> 1)) * a.pre_ld;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const int n = n0 + c * 16 + g * 4;
                    pq[p][c] = *reinterpret_cast<const f32x4*>(pr + (n < a.N ? n : 0));   // (pre_ld % 4 == 0, N % 4 == 0: checked by the launcher)
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = (wave * PT + p) * 16 + li;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int nl = c * 16 + g * 4;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + nl);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + BN + nl);
                f32x4 v = __builtin_elementwise_fma(ac1[c][p], (f32x4){0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f}, acc[c][p]);
                if (use_pre) v += pq[p][c];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], sc[r], sh[r]);
                if (conv_relu6) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], 0.f, 6.f);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = yr_apply_act(v[r], a.act);
                }
                *reinterpret_cast<f32x4*>(E + m * ELD + nl) = v;
            }
        }
    }
    // ---- the depthwise parameters of this thread's channel quad (NTH % QN == 0: every item of a thread has the same quad)
    const int NX = R.x1 - R.x0, NY = R.y1 - R.y0;
    const int q = tid % QN;
    const int nq = n0 + 4 * q;                                    // first channel of the quad
    const bool chan_ok = nq < a.N;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 tap[9][2], dsh[2];
    {
        const float* dp = h.dw + (chan_ok ? nq : 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(dp + (size_t)t * h.ldf);
            tap[t][0] = (f32x2){w4[0], w4[1]}; tap[t][1] = (f32x2){w4[2], w4[3]};
        }
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(dp + (size_t)9 * h.ldf);
        dsh[0] = (f32x2){s4[0], s4[1]}; dsh[1] = (f32x2){s4[2], s4[3]};
    }
    __syncthreads();

    // ---- depthwise 3x3 from E.  An item = (row group, output column, channel quad); row group grp owns output rows [ya, yb);
    // input row r feeds rows r - 1 (ky = 2), r (ky = 1), r + 1 (ky = 0).
    int ng = (NTH + NX * QN - 1) / (NX * QN);                     // row groups: enough items for every lane
    if (ng > NY) ng = NY;
    const int nitems = NX * QN * ng;
    f32x4 psum = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool dw_swish = h.dw_act == YR_ACT_SWISH, dw_relu6 = h.dw_act == YR_ACT_RELU6;
    if (!(h.exp & 2)) {
        for (int item = tid; item < nitems; item += NTH) {
            const int xi = (item / QN) % NX, grp = item / (QN * NX);
            const int ya = R.y0 + grp * NY / ng, yb = R.y0 + (grp + 1) * NY / ng;
            const int x = R.x0 + xi;
            const bool lok = x - 1 >= 0, rok = x + 1 < a.W;
            const float* ecol = E + (x - R.xlo) * ELD + 4 * q;
            const f32x2 z2 = (f32x2){0.f, 0.f};
            f32x2 a0[2] = {z2, z2}, a1[2] = {z2, z2}, a2[2];
            float* outp = a.out + ((size_t)b * a.H * a.W + x) * a.out_ld + nq;
            for (int r = ya - 1; r <= yb; ++r) {
                a2[0] = dsh[0]; a2[1] = dsh[1];
                if (r >= 0 && r < a.H) {   // (rows outside the map are TF's zero padding)
                    const float* er = ecol + (r - R.ylo) * R.RW * ELD;
                    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const f32x4 el = lok ? *reinterpret_cast<const f32x4*>(er - ELD) : zero;
                    const f32x4 em = *reinterpret_cast<const f32x4*>(er);
                    const f32x4 eg = rok ? *reinterpret_cast<const f32x4*>(er + ELD) : zero;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {   // packed multiply-adds (v_pk_fma_f32): two channels per instruction
                        const f32x2 l2 = (f32x2){el[2 * hh], el[2 * hh + 1]}, m2 = (f32x2){em[2 * hh], em[2 * hh + 1]}, g2 = (f32x2){eg[2 * hh], eg[2 * hh + 1]};
                        a0[hh] = __builtin_elementwise_fma(l2, tap[6][hh], a0[hh]);
                        a1[hh] = __builtin_elementwise_fma(l2, tap[3][hh], a1[hh]);
                        a2[hh] = __builtin_elementwise_fma(l2, tap[0][hh], a2[hh]);
                        a0[hh] = __builtin_elementwise_fma(m2, tap[7][hh], a0[hh]);
                        a1[hh] = __builtin_elementwise_fma(m2, tap[4][hh], a1[hh]);
                        a2[hh] = __builtin_elementwise_fma(m2, tap[1][hh], a2[hh]);
                        a0[hh] = __builtin_elementwise_fma(g2, tap[8][hh], a0[hh]);
                        a1[hh] = __builtin_elementwise_fma(g2, tap[5][hh], a1[hh]);
                        a2[hh] = __builtin_elementwise_fma(g2, tap[2][hh], a2[hh]);
                    }
                }
                if (r - 1 >= ya) {   // output row r - 1 is complete (r - 1 < yb holds inside the loop)
                    f32x4 v = (f32x4){a0[0][0], a0[0][1], a0[1][0], a0[1][1]};
                    if (dw_swish) {         // (uniform branches: a per-element switch made the compiler evaluate the pinned-expf paths as well)
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = head_swish(v[i]);
                    } else if (dw_relu6) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], 0.f, 6.f);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = yr_apply_act(v[i], h.dw_act);
                    }
                    if (chan_ok) {
                        *reinterpret_cast<f32x4*>(outp + (size_t)(r - 1) * a.W * a.out_ld) = v;
                        psum += v;
                    }
                }
                a0[0] = a1[0]; a0[1] = a1[1]; a1[0] = a2[0]; a1[1] = a2[1];
            }
        }
    }
    if (h.sums == nullptr || (h.exp & 8)) return;   // uniform

    // ---- squeeze-excite: the workgroup's per-channel sums in a fixed order -> its slice of the region's row; then the tail
    __syncthreads();   // E is no longer read
    f32x4* red = reinterpret_cast<f32x4*>(lds_raw);
    red[tid] = psum;   // (threads without an item: zeros)
    __syncthreads();
    if (tid < QN && n0 + 4 * tid < a.N) {
        f32x4 s = red[tid];
        for (int j = 1; j < NTH / QN; ++j) s += red[tid + j * QN];
        yr_st_sums4(h.sums + ((size_t)b * (h.nsy * h.nsx) + (R.iy * h.nsx + R.ix)) * h.ld_sums + n0 + 4 * tid, h.se.sums != nullptr, s[0], s[1], s[2], s[3]);
    }
    if (!(h.exp & 16)) yr_se_tail_arrive<NTH>(h.se, b, 1u, flag, reinterpret_cast<float*>(lds_raw));   // (16: probing - sums without the arrival)
}

// ------------------------------------------------------------------------------------------------------------------------
// Front end 1: pws_common.h's register-staged loop (any source transform).
template <int CT, bool SIMPLE>
__global__ __launch_bounds__(256, 2) void head_kernel(HeadArgs h) {
    constexpr int WM = 4, NTH = 64 * WM, PT = 3, BM = 16 * PT * WM, BN = 16 * CT;
    static_assert(BM == HEAD_BM, "region size");
    constexpr int RPP = NTH / PWS_KQ, A_PASSES = BM / RPP, B_PASSES = (BN + RPP - 1) / RPP;
    constexpr int STAGE = 2 * (BM + BN) * PWS_LD * 2, EBYTES = BM * (BN + 4) * 4;
    constexpr int LDSB = STAGE > EBYTES ? STAGE : EBYTES;
    __shared__ __attribute__((aligned(16))) char lds_raw[LDSB];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];     // the slice's conv BN scale | shift
    __shared__ unsigned flag;
    _Float16* lds = reinterpret_cast<_Float16*>(lds_raw);
    const PwArgs& a = h.p;
    const int tid = threadIdx.x;
    const HeadRegion R = head_region(h, BN);
    const int kp = a.S.kp;
    if (tid < BN) {
        const int n = R.n0 + tid < a.N ? R.n0 + tid : a.N - 1;
        ss[tid] = a.scale ? a.scale[n] : 1.f;
        ss[BN + tid] = a.shift ? a.shift[n] : 0.f;
    }
    const int lr = tid / PWS_KQ;
    constexpr int MODE = SIMPLE ? 2 : 0;
    const bool gated = SIMPLE && a.gate != nullptr;
    PwRow<MODE> row[A_PASSES];
    pw_unroll<A_PASSES>([&](auto P) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        const int m = lr + p * RPP;
        const bool valid = m < R.RP;
        const int mm = valid ? m : 0;
        const int ry = mm / R.RW, rx = mm - ry * R.RW;
        row[p].init_at(a, R.b, R.ylo + ry, R.xlo + rx, valid);
        if constexpr (SIMPLE)
            if (!gated) row[p].grow = a.wt;  // ungated: the gate load becomes a (cached, ignored) weight quad
    });
    const float* brow[B_PASSES];
    pw_unroll<B_PASSES>([&](auto P) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        const int n = R.n0 + lr + p * RPP;
        brow[p] = a.wt + (size_t)(n < a.N ? n : 0) * kp;
    });
    f32x4 acc[CT][PT], ac1[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) { acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f}; ac1[c][p] = acc[c][p]; }
    if (!(h.exp & 1)) pws_k_loop<NTH, PT, CT, WM, 1, MODE, A_PASSES, B_PASSES>(a, row, brow, gated, lds, acc, ac1);
    head_finish<NTH, PT, CT>(h, R, acc, ac1, lds_raw, ss, &flag);
}

// ------------------------------------------------------------------------------------------------------------------------
// Front end 2: LDS-direct activations, pre-cut weight planes.
//   k space = the sources' channels, each source in chunks of 32 (the last one zero-filled): chunk (s, j) = channels 32 j .. of source s.
//   p.wt = the float32 words that hold [NT = ceil(F / 16)][NK chunks][2 planes][64 lanes][8 halves] (compiler.head_pack): lane
//   (m = l % 16, g = l / 16) of cout tile t, chunk (s, j): W[16 t + m][k of channel 32 j + 8 g + i of source s], zero beyond the
//   source's channels and beyond F; h plane, then m plane = f16((w - h) 2^11).
typedef __amdgpu_buffer_rsrc_t head_rsrc;
__device__ __forceinline__ head_rsrc head_make_rsrc(const void* base, unsigned bytes) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#define HEAD_DEAD 0x7f000000u   // a byte offset beyond every descriptor's num_records: the load returns zeros
#define HEAD_ASTAGE (HEAD_BM * 128)                 // bytes of one activation stage: [192 rows][32 floats], 16-byte slots XOR-swizzled
typedef __attribute__((address_space(3))) void* head_lds_ptr;

// byte offset of source pixel (b, y, x) - the consumer's pixel through the source's transform - or a dead offset
__device__ __forceinline__ unsigned head_src_off(bool valid, int b, int y, int x, int xform, int sh, int sw, int ld) {
    const int sy = xform == YR_X_UP2 ? y >> 1 : y, sx = xform == YR_X_UP2 ? x >> 1 : x;
    return valid ? (unsigned)(((b * sh + sy) * sw + sx) * ld) * 4u : HEAD_DEAD;
}

template <int CT>
__global__ __launch_bounds__(256, 2) void head2_kernel(HeadArgs h) {
    constexpr int NTH = 256, PT = 3, BM = HEAD_BM, BN = 16 * CT;
    constexpr int BSTAGE = CT * 2 * 1024;                                       // weight planes of a chunk: [CT][2][64 lanes x 16 bytes]
    constexpr int STAGE = HEAD_ASTAGE + BSTAGE, EBYTES = BM * (BN + 4) * 4;
    constexpr int LDSB = 2 * STAGE > EBYTES ? 2 * STAGE : EBYTES;
    __shared__ __attribute__((aligned(1024))) char lds_raw[LDSB];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];
    __shared__ __attribute__((aligned(16))) float gl[512];                      // the SE gate of a gated single source (kp <= 512)
    __shared__ unsigned flag;
    const PwArgs& a = h.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const HeadRegion R = head_region(h, BN);
    if (tid < BN) {
        const int n = R.n0 + tid < a.N ? R.n0 + tid : a.N - 1;
        ss[tid] = a.scale ? a.scale[n] : 1.f;
        ss[BN + tid] = a.shift ? a.shift[n] : 0.f;
    }
    const bool gated = a.gate != nullptr;
    if (gated)
        for (int k = tid; k < 512; k += NTH) gl[k] = k < a.S.s[0].c ? a.gate[(size_t)R.b * a.gate_ld + k] : 0.f;

    // ---- what this thread fetches: slot (row, position) of each of its six LDS-direct loads per chunk, per source the byte
    // offset of that row's pixel.  Loads write LDS linearly (wave-uniform base + lane * 16), so the XOR swizzle that makes the
    // fragment reads conflict-free is applied to the SOURCE chunk: position c of row r holds the row's chunk c ^ ((r >> 1) & 7).
    const int cpos = lane & 7;
    unsigned off0[6], off1[6], off2[6];
    int cg[6];
    // (straight-line code with literal indices - loops and lambdas over these arrays left them in scratch memory)
#define HEAD_SLOT(i)                                                                                                              \
    {                                                                                                                             \
        const int m = ((i) * 4 + wave) * 8 + (lane >> 3);                                                                         \
        cg[i] = cpos ^ ((m >> 1) & 7);                                                                                            \
        const bool valid = m < R.RP;                                                                                              \
        const int mm = valid ? m : 0;                                                                                             \
        const int ry = mm / R.RW, rx = mm - ry * R.RW;                                                                            \
        const int y = R.ylo + ry, x = R.xlo + rx;                                                                                 \
        off0[i] = head_src_off(valid, R.b, y, x, a.S.s[0].xform, a.S.s[0].h, a.S.s[0].w, a.S.s[0].ld);                            \
        off1[i] = a.S.n > 1 ? head_src_off(valid, R.b, y, x, a.S.s[1].xform, a.S.s[1].h, a.S.s[1].w, a.S.s[1].ld) : HEAD_DEAD;    \
        off2[i] = a.S.n > 2 ? head_src_off(valid, R.b, y, x, a.S.s[2].xform, a.S.s[2].h, a.S.s[2].w, a.S.s[2].ld) : HEAD_DEAD;    \
    }
    HEAD_SLOT(0) HEAD_SLOT(1) HEAD_SLOT(2) HEAD_SLOT(3) HEAD_SLOT(4) HEAD_SLOT(5)
#undef HEAD_SLOT
    // (whole-batch descriptors: offsets are 32-bit - the launcher checks the sources are below 2 GB)
    const head_rsrc rs0 = head_make_rsrc(a.S.s[0].ptr, 0x7effffffu);
    // (no dynamic index into the kernel arguments: it would make the compiler copy the whole argument block to scratch)
    const head_rsrc rs1 = head_make_rsrc(a.S.n > 1 ? a.S.s[1].ptr : a.S.s[0].ptr, 0x7effffffu);
    const head_rsrc rs2 = head_make_rsrc(a.S.n > 2 ? a.S.s[2].ptr : a.S.s[0].ptr, 0x7effffffu);
    const unsigned ntiles = (unsigned)((a.N + 15) / 16);
    const int nk = h.nk;
    const unsigned tile0 = (unsigned)(R.n0 / 16);
    const head_rsrc rsw = head_make_rsrc(a.wt, ntiles * (unsigned)nk * 2048u);
    const int c0 = a.S.s[0].c, c1 = a.S.n > 1 ? a.S.s[1].c : 0, c2 = a.S.n > 2 ? a.S.s[2].c : 0;
    const int n0c = (c0 + 31) >> 5, n1c = (c1 + 31) >> 5;     // chunks of sources 0, 1

    f32x4 acc[CT][PT], ac1[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p) { acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f}; ac1[c][p] = acc[c][p]; }

    const int g = lane >> 4, li = lane & 15;
    if (!(h.exp & 1)) {
        // one loop body issues chunk ck + 1 and multiplies chunk ck (ck = -1: the prologue's issue of chunk 0) - written out in the
        // loop, not as a lambda called from two places: the closure kept the offset arrays in scratch memory
        for (int ck = -1; ck < nk; ++ck) {
            if (ck >= 0) __builtin_amdgcn_s_waitcnt(0x0f70);    // vmcnt(0): this wave's share of chunk ck has landed
            __syncthreads();                                    // ... everybody's; everybody is done with the other stage
            if (ck + 1 < nk) {
                const int ci = ck + 1, stage = ci & 1;
                const int s = ci < n0c ? 0 : ci < n0c + n1c ? 1 : 2;
                const int kl = (ci - (s == 0 ? 0 : s == 1 ? n0c : n0c + n1c)) * 32;
                const int cs = s == 0 ? c0 : s == 1 ? c1 : c2;
                const int cq = (cs + 3) & ~3;                          // whole quads of the source (a partial last quad is masked at the fragment)
                char* sb = lds_raw + stage * STAGE;
#define HEAD_ISSUE(i)                                                                                                  \
    {                                                                                                                  \
        const int k = kl + 4 * cg[i];                                                                                  \
        const unsigned ro = s == 0 ? off0[i] : s == 1 ? off1[i] : off2[i];                                             \
        const unsigned vo = k < cq ? ro + (unsigned)k * 4u : HEAD_DEAD;                                                \
        const head_lds_ptr dst = (head_lds_ptr)(sb + ((i) * 4 + wave) * 1024);                                         \
        if (s == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, vo, 0, 0, 0);                               \
        else if (s == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, vo, 0, 0, 0);                          \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, dst, 16, vo, 0, 0, 0);                                      \
    }
                HEAD_ISSUE(0) HEAD_ISSUE(1) HEAD_ISSUE(2) HEAD_ISSUE(3) HEAD_ISSUE(4) HEAD_ISSUE(5)
#undef HEAD_ISSUE
                // the chunk's weight planes: CT * 2 wave-loads of 1 KB shared out over the four waves
#pragma unroll
                for (int u = 0; u < (CT * 2 + 3) / 4; ++u) {
                    const int w = u * 4 + wave;                         // (tile t = w / 2, plane w % 2)
                    if (CT * 2 % 4 == 0 || w < CT * 2) {                // (wave-uniform)
                        unsigned tile = tile0 + (unsigned)(w / 2);
                        if (tile >= ntiles) tile = ntiles - 1;          // (a slice beyond F recomputes the last tile; its outputs are never stored)
                        const unsigned vo = ((tile * (unsigned)nk + (unsigned)ci) * 2u + (unsigned)(w & 1)) * 1024u + (unsigned)lane * 16u;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (head_lds_ptr)(sb + HEAD_ASTAGE + w * 1024), 16, vo, 0, 0, 0);
                    }
                }
            }
            if (ck < 0) continue;
            const char* sb = lds_raw + (ck & 1) * STAGE;
            const int s = ck < n0c ? 0 : ck < n0c + n1c ? 1 : 2;
            const int kl = (ck - (s == 0 ? 0 : s == 1 ? n0c : n0c + n1c)) * 32;
            const int vc = (s == 0 ? c0 : s == 1 ? c1 : c2) - kl;     // valid channels of this chunk (>= 32: all)
            pws_u4 xh[PT], xm[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int r = (wave * PT + p) * 16 + li;
                const int sw = (r >> 1) & 7;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(sb + r * 128 + (((2 * g) ^ sw) << 4));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(sb + r * 128 + (((2 * g + 1) ^ sw) << 4));
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (vc < 32) {   // uniform: the source's last chunk - lanes of a partial quad may hold anything (pad channels)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = 8 * g + i < vc ? v[i] : 0.f;
                }
                if (gated) {     // uniform: the SE gate of the (single) source
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gl + kl + 8 * g), g1 = *reinterpret_cast<const f32x4*>(gl + kl + 8 * g + 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v[i] *= g0[i]; v[4 + i] *= g1[i]; }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned hh, mm;
                    yr_cut2(v[2 * i], v[2 * i + 1], hh, mm);
                    xh[p][i] = hh;
                    xm[p][i] = mm;
                }
            }
            pws_u4 wh[CT], wm[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                wh[c] = *reinterpret_cast<const pws_u4*>(sb + HEAD_ASTAGE + (2 * c) * 1024 + lane * 16);
                wm[c] = *reinterpret_cast<const pws_u4*>(sb + HEAD_ASTAGE + (2 * c + 1) * 1024 + lane * 16);
            }
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[c][p] = pws_mfma(wh[c], xh[p], acc[c][p]);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) ac1[c][p] = pws_mfma(wh[c], xm[p], ac1[c][p]);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) ac1[c][p] = pws_mfma(wm[c], xh[p], ac1[c][p]);
        }
    }
    __syncthreads();   // the last chunk's fragments have been read: the stages become E
    head_finish<NTH, PT, CT>(h, R, acc, ac1, lds_raw, ss, &flag);
}

template <int CT, bool V2>
static int launch_head(const HeadArgs& h, int batch, hipStream_t s) {
    constexpr int BN = 16 * CT, NTH = 256;
    const unsigned ntn = (unsigned)((h.p.N + BN - 1) / BN);
    HeadArgs a = h;
    a.se.arrivals = (unsigned)(h.nsy * h.nsx) * ntn;
    const bool simple = a.p.S.n == 1 && a.p.S.s[0].xform == YR_X_IDENTITY;
    YR_REQUIRE(a.se.sums == nullptr || yr_se_tail_floats(a.se.C, a.se.R, NTH) * 4 <= (size_t)(HEAD_BM * (BN + 4) * 4), "head: SE widths too large for the tail's LDS");
    static char nm[3][40];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "head_kernel<%d,0>", CT) + snprintf(nm[1], sizeof(nm[1]), "head_kernel<%d,1>", CT) +
                              snprintf(nm[2], sizeof(nm[2]), "head2_kernel<%d>", CT);
    (void)nm_len;
    yr_note_kernel(nm[V2 ? 2 : simple ? 1 : 0]);
    dim3 grid((unsigned)batch * (unsigned)(a.nsy * a.nsx) * ntn);
    if constexpr (V2) {
        hipLaunchKernelGGL((head2_kernel<CT>), grid, dim3(NTH), 0, s, a);
    } else {
        if (simple) hipLaunchKernelGGL((head_kernel<CT, true>), grid, dim3(NTH), 0, s, a);
        else hipLaunchKernelGGL((head_kernel<CT, false>), grid, dim3(NTH), 0, s, a);
    }
    YR_LAUNCH_CHECK();
    return YR_OK;
}

int yr_launch_head(const yr_op& op, int batch, hipStream_t s) {
    if ((op.k & 0x60) == 0x60 && op.dtype == YR_F32) return yr_launch_head_stream(op, batch, s);   // the weight-streaming form (headstream.hip)
    if (op.k & 0x40) return yr_launch_head_walk(op, batch, s);   // the walking form (headwalk.hip; headwalk_h.hip for the 16-bit plans)
    YR_REQUIRE(op.dtype == YR_F32 && op.out_dtype == YR_F32, "head: the LDS-tiled forms are float32 only (16-bit plans: the walking form, k bit 6)");
    YR_REQUIRE((op.k & 0x7f) == 3 && op.stride == 1, "head: depthwise 3x3, stride 1");
    YR_REQUIRE(op.out && op.wgt && op.wgt2, "head: null pointer");
    const bool v2 = (op.k & 0x80) != 0;     // the weights are float16 planes in fragment order (compiler.head_pack)
    HeadArgs h;
    PwArgs& a = h.p;
    // ---- the convolution's sources, as yr_launch_pointwise reads them
    yr_op tmp = op;
    a.pre = nullptr; a.pre_ld = 0;
    if (tmp.nsrc >= 2 && tmp.src[tmp.nsrc - 1].xform == YR_X_UP2_ADD) {
        const yr_src& ps = tmp.src[tmp.nsrc - 1];
        YR_REQUIRE(ps.dtype == YR_F32 && ps.ptr && ps.c == op.cout && ps.ld >= ps.c && ps.ld % 4 == 0 && ps.h * 2 == op.h && ps.w * 2 == op.w && ((uintptr_t)ps.ptr % 16) == 0,
                   "head: bad up2_add source");
        a.pre = (const float*)ps.ptr; a.pre_ld = ps.ld;
        tmp.nsrc -= 1;
    }
    int rc = yr_make_srcset(tmp, &a.S);
    if (rc) return rc;
    h.nk = 0;
    for (int i = 0; i < a.S.n; ++i) {
        YR_REQUIRE(a.S.s[i].xform != YR_X_DW3 && a.S.s[i].xform != YR_X_UP2_ADD, "head: source transform %d is not supported", a.S.s[i].xform);
        if (v2) {
            YR_REQUIRE(a.S.s[i].xform == YR_X_IDENTITY || a.S.s[i].xform == YR_X_UP2, "head (packed weights): pooled sources need the float32 weight layout");
            YR_REQUIRE((uint64_t)batch * a.S.s[i].h * a.S.s[i].w * a.S.s[i].ld * 4ull < 0x7e000000ull, "head: a source of %d images exceeds the 32-bit offsets of its loads", batch);
        }
        h.nk += (a.S.s[i].c + 31) / 32;
    }
    if (v2) YR_REQUIRE(a.S.n <= 3, "head (packed weights): at most three k-space sources");
    YR_REQUIRE(a.S.kp >= 16, "head: the convolution must be at least 16 channels deep");
    a.wt = op.wgt; a.scale = op.scale; a.shift = op.shift; a.res = nullptr; a.res_ld = 0;
    a.gate = nullptr; a.gate_ld = 0;
    // `gate` is the squeeze-excite SUMS buffer this op writes; the SE gate of the single source (bu3_conv) arrives as res / res_ld
    if (op.res) {
        YR_REQUIRE(a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY && op.res_ld >= a.S.kp && a.S.kp <= 512, "head: a gated source must be the single identity source (at most 512 channels)");
        a.gate = (const float*)op.res; a.gate_ld = op.res_ld;
    }
    a.out = (float*)op.out; a.out_ld = op.out_ld;
    a.H = op.h; a.W = op.w; a.N = op.cout; a.M = batch * op.h * op.w;
    a.act = (op.k >> 8) & 0xff; a.pool = 0;
    a.dw_w = nullptr; a.dw_scale = a.dw_shift = nullptr; a.dw_stride = a.dw_act = a.dw_pad_t = a.dw_pad_l = 0; a.out_f32 = 1;
    YR_REQUIRE(op.out_ld % 4 == 0 && op.out_ld >= op.cout && op.cout % 4 == 0 && ((uintptr_t)op.out % 16) == 0, "head: output stride / width");
    if (a.pre) YR_REQUIRE(op.h % 2 == 0 && op.w % 2 == 0, "head: an up-sampled addend needs even dims");
    h.dw = op.wgt2; h.ldf = yr_round_up(op.cout, 4); h.dw_act = op.act;
    head_geometry(op.h, op.w, &h.nsy, &h.nsx);
    YR_REQUIRE(h.nsy > 0, "head: a %d x %d map has no region split that fits a workgroup", op.h, op.w);
    static const int exp = getenv("YR_HEAD_EXP") ? atoi(getenv("YR_HEAD_EXP")) : 0;
    h.exp = exp;
    if (op.gate) YR_REQUIRE(op.se_reduced == h.nsy * h.nsx && op.gate_ld % 4 == 0 && op.gate_ld >= op.cout && ((uintptr_t)op.gate % 16) == 0,
                            "head: the squeeze-excite sums buffer must hold %d rows per image (se_reduced = %d)", h.nsy * h.nsx, op.se_reduced);
    rc = yr_make_se_tail(op, h.nsy * h.nsx, &h.se);
    if (rc) return rc;
    h.sums = const_cast<float*>(op.gate); h.ld_sums = op.gate_ld;
    if (exp & 32) h.se.w = nullptr;   // (probing: arrival without the FC pair)
    const int cfg = (op.k >> 16) & 0xff;   // 0: the library's choice (a function of the shape); else the cout tiles of 16 per workgroup
    const int ct = cfg ? cfg : op.cout >= 64 ? 4 : op.cout >= 32 ? 2 : 1;
    if (v2) {
        if (ct >= 4) return launch_head<4, true>(h, batch, s);
        if (ct >= 2) return launch_head<2, true>(h, batch, s);
        return launch_head<1, true>(h, batch, s);
    }
    if (ct >= 4) return launch_head<4, false>(h, batch, s);
    if (ct >= 2) return launch_head<2, false>(h, batch, s);
    return launch_head<1, false>(h, batch, s);
}
