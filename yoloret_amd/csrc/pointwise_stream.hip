// Pointwise (1x1) convolution of float32 plans, PIXEL-STATIONARY form of the split arithmetic (round 6; se_reduced bit 18 of a
// POINTWISE op - a property of the plan: the op's weights are stored as float16 planes in fragment order, compiler.head_pack).
// The 1x1 convs of the neck and the heads (reference code/yolo3/model.py:91-157: Concatenate + Conv2D 1x1 + BN (+ ReLU6), the SE
// block's projection, the y convs) are HBM-bound - 14 .. 140 MB per launch at 64 images, worth 3 .. 28 us of memory time - and
// pws_kernel (pointwise_split.hip) spends 15 .. 57 us on them: a workgroup tile walks its k chunks one barrier pair at a time, ~165
// instructions of staging per chunk and wave for a handful of MFMAs.  Here (headstream.hip's conv stage without the depthwise one):
//   * one workgroup per CU, PERSISTENT: its run of cout tiles is RESIDENT in LDS - the host-cut float16 planes arrive through
//     LDS-direct buffer loads (one wait, one barrier; no barrier afterwards).  Weights beyond 150 KB are shared by several
//     workgroups (csplit) that walk the same pixels (read again, from the L2) with a run of the tiles each;
//   * a wave walks pixel tiles of ROWS x 16 pixels (GEMM rows): a tile's channels over the WHOLE k space are fetched at once,
//     straight into the MFMA operand layout (lane (li, g): pixel li, channels 32 c + 8 g .. + 7 of chunk c), times the SE gate,
//     cut into float16 planes in registers; the waves of a CU drift apart, so one's round trip hides behind the others' MFMAs;
//   * per cout tile: 3 ROWS NK MFMAs (the split form's three products per chunk, in pws_kernel's order: the results are
//     bit-identical to pws_kernel's), then BN, ReLU6 | none, (2 x 2 max,) one 16-byte store per lane.
// (A two-row form - 32 pixels per wave, half the LDS traffic of the weight fragments - is built and measured behind: YR_PWT_ROWS=2.)
// No result depends on the launch geometry: a cout of a pixel is one accumulator chain over the k chunks in ascending order.
#include <algorithm>
#include <cstdlib>

#include "pws_common.h"
#include "mbr_common.h"

typedef __attribute__((address_space(3))) void* pwt_lds_ptr;

// n / d for n < 2^31 (Granlund - Montgomery): d = 1: s < 0
struct PwtDiv { unsigned m; int s; };
static inline PwtDiv pwt_div_make(unsigned d) {
    if (d <= 1) return {0u, -1};
    int l = 0;
    while ((1u << l) < d) ++l;
    const unsigned long long m = ((1ull << (31 + l)) + d - 1) / d;
    return {(unsigned)m, l - 1};
}
__device__ __forceinline__ int pwt_div(int n, PwtDiv d) { return d.s < 0 ? n : (int)(__umulhi((unsigned)n, d.m) >> d.s); }

struct PwtArgs {
    const float* planes;     // [T][NK][2 planes][64 lanes][8 halves] as float32 words
    unsigned plane_bytes;
    int T;                   // cout tiles (ceil(N / 16))
    int csplit;              // workgroups that share the pixel tiles, each with its own run of the cout tiles
    int nwg;                 // workgroups per part: pixel tile i belongs to workgroup i % nwg, wave (i / nwg) % NW
    int ntiles;              // pixel tiles (ceil(M / (16 ROWS)))
    PwtDiv d_hw, d_wq, d_hwq;   // H W | W / 2 | (H / 2) (W / 2)
    unsigned in_bytes, out_bytes, gate_bytes;
    // two outputs (PwArgs::out2): cout tiles [0, TA) belong to the first, [TA, T) to the second; quad: GEMM rows are walked in 2 x 2-quad-major
    // order (one of the outputs is pooled)
    int TA, quad;
    unsigned out2_bytes;
};

// waves per workgroup, from the measured register needs (16 / 12 / 8 waves: 128 / 168 / 256 registers)
#ifndef PWT_MAX_WAVES
#define PWT_MAX_WAVES 16      // (experiments: -DPWT_MAX_WAVES=8 | 12 leaves room on the CU for the kernels of other steps in flight)
#endif
constexpr int pwt_waves_fit(int nk, int rows, int mode);
constexpr int pwt_waves(int nk, int rows, int mode) { return pwt_waves_fit(nk, rows, mode) < PWT_MAX_WAVES ? pwt_waves_fit(nk, rows, mode) : PWT_MAX_WAVES; }
constexpr int pwt_waves_fit(int nk, int rows, int mode) {
    const int nkr = nk * rows;
    if (mode == 0) return nkr <= 6 ? 16 : nkr <= 10 ? 12 : 8;
    if (mode == 1) return nkr <= 9 ? 16 : nkr <= 12 ? 12 : 8;
    if (rows == 1) return nk <= 8 ? 16 : nk <= 10 ? 12 : 8;       // gated: two sets of gate quads on top
    return nk <= 2 ? 16 : nk <= 4 ? 12 : 8;
}

#ifdef PWT_TIMING     // (python tools/relink.py pointwise_stream.hip -DPWT_TIMING; tools/pwt_timing.py: per-phase shader-clock totals of every wave)
__device__ unsigned pwt_dbg[256 * 16 * 8];
extern "C" int yr_pwt_dbg_read(unsigned* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pwt_dbg), (size_t)n * 4); }
#define PWT_T(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define PWT_T(i)
#endif

// GEMM row -> conv pixel (pw_pixel_of_row with the divisions by multiplication)
__device__ __forceinline__ int pwt_pixel_of_row(const PwArgs& a, const PwtArgs& x, int m) {
    if (!x.quad) return m;
    const int q = m >> 2, sub = m & 3;
    const int wq = a.W >> 1, hwq = (a.H >> 1) * wq;
    const int b = pwt_div(q, x.d_hwq), r = q - b * hwq;
    const int yq = pwt_div(r, x.d_wq), xq = r - yq * wq;
    return (b * a.H + 2 * yq + (sub >> 1)) * a.W + 2 * xq + (sub & 1);
}

template <int NK, int ROWS, int MODE>
__global__ __launch_bounds__(64 * pwt_waves(NK, ROWS, MODE)) void pwt_kernel(PwArgs a, PwtArgs x) {
    constexpr int NW = pwt_waves(NK, ROWS, MODE), TB = NK * 2048;       // bytes of one tile's planes
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];   // [run][NK][2][64] x 16 bytes | scale [16 run] | shift [16 run]
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int part = blockIdx.x % x.csplit, wg = blockIdx.x / x.csplit;
    const int t0 = part * x.T / x.csplit, t1 = (part + 1) * x.T / x.csplit, run = t1 - t0;
    float* const ss = reinterpret_cast<float*>(lds_raw + (size_t)run * TB);
#ifdef PWT_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif

    // ---- the run's planes: LDS-direct, 1 KB pieces dealt to the waves (beyond the array: zeros); BN scale and shift of the run's
    // couts by plain loads (issued now, written to LDS in front of the barrier)
    {
        const mbr_rsrc wsrc = mbr_make_rsrc(x.planes, x.plane_bytes);
        const int npiece = run * NK * 2;
        for (int p = w; p < npiece; p += NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (pwt_lds_ptr)(lds_raw + p * 1024), 16, (unsigned)(t0 * TB + p * 1024 + lane * 16), 0, 0, 0);
    }
    // (unconditional loads, selected afterwards: a load under a branch is waited for at the join - with the planes in flight in front of it)
    const int bn_i = (int)threadIdx.x;      // (run <= 32 tiles: 512 couts)
    const int bn_n = bn_i < 16 * run && (a.out2 != nullptr || 16 * t0 + bn_i < a.N) ? 16 * t0 + bn_i : 0;      // (two outputs: the vectors are padded to the tiles)
    const float bn_sl = (a.scale ? a.scale : x.planes)[bn_n], bn_hl = (a.shift ? a.shift : x.planes)[bn_n];

    // ---- a wave walks its pixel tiles; the loads of a tile go out ALL AT ONCE (one round trip), and those of the first tile before the
    // wait for the planes.  The SE gate's quads (a few KB per image: cache hits) are read behind them, chunk by chunk, so that they do not
    // double the registers in flight.
    const int kp = a.S.kp;
    const int stride = x.nwg * NW;
    int tile = wg + x.nwg * w;
    const mbr_rsrc xsrc = mbr_make_rsrc(a.S.s[0].ptr, x.in_bytes), osrc = mbr_make_rsrc(a.out, x.out_bytes);
    const mbr_rsrc osrc2 = mbr_make_rsrc(a.out2 ? a.out2 : a.out, a.out2 ? x.out2_bytes : 16u);
    const mbr_rsrc gsrc = mbr_make_rsrc(MODE == 2 ? a.gate : a.S.s[0].ptr, MODE == 2 ? x.gate_bytes : 16u);
    PwRow<0> row[MODE == 0 ? ROWS : 1];
    float4 xa[ROWS][NK][2];
    int cv[MODE == 0 ? ROWS : 1][MODE == 0 ? NK : 1][2];
    unsigned goff[ROWS];      // MODE 2: byte offset of the pixel's gate row
    int pix[ROWS];            // conv pixel of the lane's GEMM row (= the row, unless the rows are walked in quad-major order)
    // the gate's quads, two chunks at a time and one pair ahead of the cut that uses them (read one by one at their cut, each is an L2 round
    // trip in front of it: 16 chunks, 18 k cycles)
    constexpr int GQ = 2, NGQ = (NK + GQ - 1) / GQ;
    float4 gq[MODE == 2 ? 2 : 1][MODE == 2 ? ROWS : 1][GQ][2];
    auto gate_issue = [&](auto J) __attribute__((always_inline)) {      // the loads of chunk pair J (into register set J % 2)
        constexpr int j = decltype(J)::value;
        if constexpr (MODE == 2 && j < NGQ) {
            pw_unroll<ROWS * GQ * 2>([&](auto U) __attribute__((always_inline)) {
                constexpr int u = decltype(U)::value, i = u / (2 * GQ), cc = (u / 2) % GQ, q = u % 2, c = j * GQ + cc;
                const int k = c * 32 + g * 8 + 4 * q;
                if constexpr (c < NK) gq[j & 1][i][cc][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gsrc, k < kp ? goff[i] + 4u * k : MBR_DEAD, 0, 0));
            });
        }
    };
    const int csrc = a.S.s[0].c, ldsrc = a.S.s[0].ld;
    auto fetch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int m = (tile * ROWS + i) * 16 + li;
            if constexpr (MODE == 0) {
                row[i].init(a, m);
                pix[i] = pw_pixel_of_row(a, m < a.M ? m : 0);
                pw_unroll<NK * 2>([&](auto U) __attribute__((always_inline)) {
                    constexpr int u = decltype(U)::value, c = u / 2, q = u % 2;
                    float4 unused;
                    row[i].template issue<false>(a, c * 32 + g * 8 + 4 * q, kp, xa[i][c][q], unused, cv[i][c][q]);
                });
            } else {
                // one identity source: 32-bit offsets into a buffer descriptor; a quad beyond the k space (or of a row beyond M) reads zeros
                const int mm = pwt_pixel_of_row(a, x, m < a.M ? m : 0);
                pix[i] = mm;
                const unsigned poff = (unsigned)mm * (unsigned)ldsrc * 4u;
                if constexpr (MODE == 2) goff[i] = (unsigned)pwt_div(mm, x.d_hw) * (unsigned)a.gate_ld * 4u;
                pw_unroll<NK * 2>([&](auto U) __attribute__((always_inline)) {
                    constexpr int u = decltype(U)::value, c = u / 2, q = u % 2;
                    const int k = c * 32 + g * 8 + 4 * q;
                    xa[i][c][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, (m < a.M && k < kp) ? poff + 4u * k : MBR_DEAD, 0, 0));
                });
            }
        }
        gate_issue(std::integral_constant<int, 0>{});
    };
    // (the other side of every "fetch if there is another tile" overwrites the registers too: else the old pixels stay live - next to the
    //  planes cut from them - until the branch, twice the registers)
    auto no_fetch = [&]() __attribute__((always_inline)) {
        pw_unroll<ROWS * NK * 2>([&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            xa[u / (2 * NK)][(u / 2) % NK][u % 2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (MODE == 0) cv[u / (2 * NK)][(u / 2) % NK][u % 2] = 0;
        });
    };
    if (tile < x.ntiles) fetch();       // (wave-uniform)
    else no_fetch();
    PWT_T(0)
    if (bn_i < 16 * run) { ss[bn_i] = a.scale ? bn_sl : 1.f; ss[16 * run + bn_i] = a.shift ? bn_hl : 0.f; }
    __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0): this wave's pieces of the planes have landed
    __syncthreads();
    PWT_T(1)

    const f32x4 k11 = {0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f};
    const bool ragged = (csrc & 3) != 0, clamp = a.act != YR_ACT_NONE;
    while (tile < x.ntiles) {
        // ---- gate, zero the pad lanes, cut
        pws_u4 xh[ROWS][NK], xm[ROWS][NK];
        pw_unroll<ROWS * NK>([&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, c = u / ROWS, i = u % ROWS;      // (chunk-major: the gate's quads arrive pair by pair)
            float4 v0 = xa[i][c][0], v1 = xa[i][c][1];
            if constexpr (MODE == 0) {
                v0 = pw_finish<1>(v0, v0, cv[i][c][0]);
                v1 = pw_finish<1>(v1, v1, cv[i][c][1]);
            } else {
                const int k0 = c * 32 + g * 8;
                if constexpr (MODE == 2) {
                    if constexpr (i == 0 && c % GQ == 0) gate_issue(std::integral_constant<int, c / GQ + 1>{});
                    const float4 g0 = gq[(c / GQ) & 1][i][c % GQ][0], g1 = gq[(c / GQ) & 1][i][c % GQ][1];
                    v0 = make_float4(v0.x * g0.x, v0.y * g0.y, v0.z * g0.z, v0.w * g0.w);
                    v1 = make_float4(v1.x * g1.x, v1.y * g1.y, v1.z * g1.z, v1.w * g1.w);
                }
                if (ragged) {      // (uniform) the source's pad lanes (and the gate's) may hold anything
                    v0 = pw_finish<1>(v0, v0, csrc - k0);
                    v1 = pw_finish<1>(v1, v1, csrc - k0 - 4);
                }
            }
            unsigned h[4], mm[4];
            yr_cut2(v0.x, v0.y, h[0], mm[0]);
            yr_cut2(v0.z, v0.w, h[1], mm[1]);
            yr_cut2(v1.x, v1.y, h[2], mm[2]);
            yr_cut2(v1.z, v1.w, h[3], mm[3]);
            xh[i][c] = (pws_u4){h[0], h[1], h[2], h[3]};
            xm[i][c] = (pws_u4){mm[0], mm[1], mm[2], mm[3]};
        });
        PWT_T(2)
        // ---- where the lane's quads go: output row (pooled: the window's row, kept by the lane of its first pixel)
        unsigned obase[ROWS], obase2[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int m = (tile * ROWS + i) * 16 + li;
            const bool keep = m < a.M && (!a.pool || (li & 3) == 0), keep2 = m < a.M && (!a.pool2 || (li & 3) == 0);
            obase[i] = keep ? (unsigned)(a.pool ? m >> 2 : pix[i]) * (unsigned)a.out_ld * 4u : MBR_DEAD;
            obase2[i] = keep2 && a.out2 ? (unsigned)(a.pool2 ? m >> 2 : pix[i]) * (unsigned)a.out2_ld * 4u : MBR_DEAD;
        }
        for (int t = 0; t < run; ++t) {
            const pws_u4* fe = reinterpret_cast<const pws_u4*>(lds_raw + (size_t)t * TB) + lane;     // [NK][2 planes][64]
            f32x4 acc[ROWS], ac1[ROWS];
#pragma unroll
            for (int i = 0; i < ROWS; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; ac1[i] = acc[i]; }
#pragma unroll
            for (int c = 0; c < NK; ++c) {
                const pws_u4 wh = fe[(2 * c) * 64], wm = fe[(2 * c + 1) * 64];
#pragma unroll
                for (int i = 0; i < ROWS; ++i) acc[i] = pws_mfma(wh, xh[i][c], acc[i]);
#pragma unroll
                for (int i = 0; i < ROWS; ++i) ac1[i] = pws_mfma(wh, xm[i][c], ac1[i]);
#pragma unroll
                for (int i = 0; i < ROWS; ++i) ac1[i] = pws_mfma(wm, xh[i][c], ac1[i]);
            }
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + 16 * t + 4 * g);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + 16 * run + 16 * t + 4 * g);
            const bool second = t0 + t >= x.TA;      // (uniform) a tile of the second output
            const int n = 16 * (t0 + t - (second ? x.TA : 0)) + 4 * g, cnt = (second ? a.N2 : a.N) - n;       // real couts of the lane's quad (<= 0: none)
            const bool clamp_t = second ? a.act2 != YR_ACT_NONE : clamp, pool_t = second ? a.pool2 != 0 : a.pool != 0;
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                f32x4 v = __builtin_elementwise_fma(ac1[i], k11, acc[i]);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], sc[r], sh[r]);
                if (clamp_t) {      // (uniform) ReLU6
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], 0.f, 6.f);
                }
                if (pool_t) {     // (uniform) MaxPooling2D(2) across the 4 adjacent lanes of a window
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = fmaxf(v[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[r]), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
                        v[r] = fmaxf(v[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[r]), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
                    }
                }
                const unsigned ob = second ? obase2[i] : obase[i];
                const unsigned off = ob == MBR_DEAD || cnt <= 0 ? MBR_DEAD : ob + 4u * (unsigned)n;
                if (second) {
                    if (cnt >= 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), osrc2, off, 0, 0);
                    else {
#pragma unroll
                        for (int r = 0; r < 3; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), osrc2, r < cnt ? off + 4u * r : MBR_DEAD, 0, 0);
                    }
                } else if (cnt >= 4) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), osrc, off, 0, 0);
                } else {          // (the last quad of a dense 75-wide row)
#pragma unroll
                    for (int r = 0; r < 3; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), osrc, r < cnt ? off + 4u * r : MBR_DEAD, 0, 0);
                }
            }
        }
        PWT_T(3)
        tile += stride;
        if (tile < x.ntiles) fetch();
        else no_fetch();
        PWT_T(4)
#ifdef PWT_TIMING
        tacc[5] += 1;
#endif
    }
#ifdef PWT_TIMING
    __builtin_amdgcn_s_waitcnt(0x0070);
    PWT_T(6)
    if (lane < 8 && blockIdx.x < 256) {
        unsigned tv = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) tv = lane == i ? (unsigned)tacc[i] : tv;
        pwt_dbg[(blockIdx.x * 16 + w) * 8 + lane] = tv;
    }
#endif
}

template <int NK, int ROWS>
static int launch_pwt(const PwArgs& a, PwtArgs& x, hipStream_t s) {
    const bool simple = a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY;
    const int mode = !simple ? 0 : a.gate != nullptr ? 2 : 1;
    const int NW = pwt_waves(NK, ROWS, mode);
    x.ntiles = (a.M + 16 * ROWS - 1) / (16 * ROWS);
    // parts: the run of cout tiles of a workgroup must fit LDS (every part fetches the pixels again - from the L2)
    int cs = 1;
    while ((x.T + cs - 1) / cs * (size_t)(NK * 2048 + 128) > 150 * 1024 && cs < x.T) ++cs;
    x.csplit = cs;
    // workgroups per part: one per CU at most, and no more than leave every workgroup ceil(tiles / CUs) busy waves
    const int cus = 256 / cs;
    int aw = (x.ntiles + cus - 1) / cus;
    if (aw > NW) aw = NW;
    x.nwg = (x.ntiles + aw - 1) / aw;
    if (x.nwg > cus) x.nwg = cus;
    const size_t lds = (size_t)((x.T + cs - 1) / cs) * (NK * 2048 + 128);
    YR_REQUIRE(lds <= 160 * 1024 && (x.T + cs - 1) / cs <= 32, "pointwise (pixel-stationary form): %d tiles of %d chunks do not fit LDS", x.T, NK);
    static char nm[3][40];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pwt_kernel<%d,%d,0>", NK, ROWS) + snprintf(nm[1], sizeof(nm[1]), "pwt_kernel<%d,%d,1>", NK, ROWS) +
                              snprintf(nm[2], sizeof(nm[2]), "pwt_kernel<%d,%d,2>", NK, ROWS);
    (void)nm_len;
    yr_note_kernel(nm[mode]);
    const dim3 grid((unsigned)(x.nwg * cs));
#define PWT_GO(MD)                                                                                                         \
    {                                                                                                                      \
        auto kern = pwt_kernel<NK, ROWS, MD>;                                                                              \
        static bool attr_set[64] = {};                                                                                     \
        int dev = 0;                                                                                                       \
        (void)hipGetDevice(&dev);                                                                                          \
        if (!attr_set[dev & 63]) {                                                                                         \
            YR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
            attr_set[dev & 63] = true;                                                                                     \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(64 * pwt_waves(NK, ROWS, MD)), lds, s, a, x);                                 \
    }
    if (mode == 0) PWT_GO(0)
    else if (mode == 1) PWT_GO(1)
    else PWT_GO(2)
#undef PWT_GO
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// chunk counts the form is built for (a k space in between runs the next one: its planes are padded with zero chunks by the compiler)
extern "C" int yr_pwt_chunks(int kp) {
    static const int sizes[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16};
    const int nk = (kp + 31) / 32;
    for (int v : sizes)
        if (v >= nk) return v;
    return 0;
}

static int pwt_launch_images(const PwArgs& a, hipStream_t s) {
    const int nk = yr_pwt_chunks(a.S.kp);
    YR_REQUIRE(nk > 0, "pointwise (pixel-stationary form): a k space of %d channels is beyond 16 chunks", a.S.kp);
    YR_REQUIRE(a.dw_w == nullptr && a.pre == nullptr && a.res == nullptr, "pointwise (pixel-stationary form): no depthwise-folded source, addend or residual");
    YR_REQUIRE(a.act == YR_ACT_NONE || a.act == YR_ACT_RELU6, "pointwise (pixel-stationary form): activation %d (none and ReLU6 only)", a.act);
    for (int i = 0; i < a.S.n; ++i)
        YR_REQUIRE(a.S.s[i].xform != YR_X_MAXPOOL2 && a.S.s[i].xform != YR_X_MAXPOOL4, "pointwise (pixel-stationary form): no pooled source");
    PwtArgs x;
    x.TA = (a.N + 15) / 16;
    x.planes = a.wt; x.T = x.TA + (a.out2 ? (a.N2 + 15) / 16 : 0);
    x.quad = a.pool || (a.out2 && a.pool2);
    if (a.out2) YR_REQUIRE(a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY, "pointwise (pixel-stationary form): two outputs need a single identity source");
    if (a.out2) YR_REQUIRE(a.act2 == YR_ACT_NONE || a.act2 == YR_ACT_RELU6, "pointwise (pixel-stationary form): activation %d of the second output", a.act2);
    if (x.quad) YR_REQUIRE(a.H % 2 == 0 && a.W % 2 == 0, "pointwise (pixel-stationary form): a pooled output needs an even map");
    x.plane_bytes = (unsigned)x.T * (unsigned)nk * 2048u;
    x.d_hw = pwt_div_make((unsigned)(a.H * a.W));
    x.d_wq = pwt_div_make((unsigned)(a.W >> 1 > 0 ? a.W >> 1 : 1));
    x.d_hwq = pwt_div_make((unsigned)((a.H >> 1) * (a.W >> 1) > 0 ? (a.H >> 1) * (a.W >> 1) : 1));
    const long long inb = (long long)a.M * a.S.s[0].ld * 4, outb = (long long)(a.pool ? a.M / 4 : a.M) * a.out_ld * 4;
    YR_REQUIRE(inb < 0x7f000000ll && outb < 0x7f000000ll, "pointwise (pixel-stationary form): a map of %lld bytes is beyond the 32-bit offsets", inb > outb ? inb : outb);
    x.in_bytes = (unsigned)inb; x.out_bytes = (unsigned)outb;
    const long long outb2 = a.out2 ? (long long)(a.pool2 ? a.M / 4 : a.M) * a.out2_ld * 4 : 0;
    YR_REQUIRE(outb2 < 0x7f000000ll, "pointwise (pixel-stationary form): a second output of %lld bytes is beyond the 32-bit offsets", outb2);
    x.out2_bytes = (unsigned)outb2;
    x.gate_bytes = a.gate ? (unsigned)((long long)(a.M / (a.H * a.W)) * a.gate_ld * 4) : 0u;
    // 32 pixels per wave (half the fragment reads from LDS) measured behind 16 once the kernel was persistent and lean: bu3's two-output
    // launch 60 against 50 us, the 26 x 26 convs 14-28 against 14-23 (YR_PWT_ROWS=2 forces it: experiments)
    const char* rows_env = getenv("YR_PWT_ROWS");
    const bool two = rows_env && atoi(rows_env) == 2 && nk <= 6;
#define PWT_CASE(K) if (nk == K) return two ? launch_pwt<K, 2>(a, x, s) : launch_pwt<K, 1>(a, x, s);
#define PWT_CASE1(K) if (nk == K) return launch_pwt<K, 1>(a, x, s);
#ifdef PWT_ONLY
    PWT_CASE(PWT_ONLY)
#else
    PWT_CASE(1) PWT_CASE(2) PWT_CASE(3) PWT_CASE(4) PWT_CASE(5) PWT_CASE(6) PWT_CASE1(7) PWT_CASE1(8) PWT_CASE1(9) PWT_CASE1(10) PWT_CASE1(12) PWT_CASE1(16)
#endif
#undef PWT_CASE
#undef PWT_CASE1
    return YR_ERR_ARG;
}

// The kernel addresses its maps through 32-bit offsets (buffer descriptors): a batch whose maps pass 2 GB runs as several launches over
// runs of whole images (no result depends on the cut).
int yr_pw_launch_stream(const PwArgs& a0, hipStream_t s) {
    const long long hw = (long long)a0.H * a0.W;
    const int B = (int)(a0.M / hw);
    long long per_image = (a0.pool ? hw / 4 : hw) * a0.out_ld * 4;
    if (a0.out2) per_image = std::max(per_image, (a0.pool2 ? hw / 4 : hw) * a0.out2_ld * 4);
    for (int i = 0; i < a0.S.n; ++i) per_image = std::max(per_image, (long long)a0.S.s[i].h * a0.S.s[i].w * a0.S.s[i].ld * 4);
    const char* lim = getenv("YR_PWT_MAX_BYTES");      // (tests: a small limit exercises the cut)
    const long long fit = (lim && atoll(lim) > 0 ? atoll(lim) : 0x7e000000ll) / (per_image > 0 ? per_image : 1);
    if (fit >= B || (long long)B * hw != a0.M) return pwt_launch_images(a0, s);
    YR_REQUIRE(fit >= 1, "pointwise (pixel-stationary form): one image's map of %lld bytes is beyond the 32-bit offsets", per_image);
    for (int b0 = 0; b0 < B; b0 += (int)fit) {
        PwArgs a = a0;
        const int n = B - b0 < fit ? B - b0 : (int)fit;
        a.M = (int)(n * hw);
        for (int i = 0; i < a.S.n; ++i) a.S.s[i].ptr += (size_t)b0 * a.S.s[i].h * a.S.s[i].w * a.S.s[i].ld;
        if (a.gate) a.gate += (size_t)b0 * a.gate_ld;
        a.out += (size_t)b0 * (a.pool ? hw / 4 : hw) * a.out_ld;
        if (a.out2) a.out2 += (size_t)b0 * (a.pool2 ? hw / 4 : hw) * a.out2_ld;
        const int rc = pwt_launch_images(a, s);
        if (rc) return rc;
    }
    return YR_OK;
}
