// Depthwise K x K (K = 5 | 3) stride-1 convolution on 16-bit maps, LDS-tiled: the form for the wide EfficientNet stages whose
// blocks the fused kernels do not take (more than 128 block inputs: efficientnet.py:501-510 with kernel_size 5) and - K = 3,
// round 3 - for the detection heads' MBConv depthwise stages (code/yolo3/model.py:98-114: 52 x 52 x 128 ... 13 x 13 x 512,
// squeeze-excite form) and every other 16-bit 3 x 3 map with 64 channels or more.  Same arithmetic as dw_kernel<K,1,..> -
// float32 accumulation in (ky, kx) order, BatchNorm, activation, one rounding on store - so the two forms are
// bit-identical; only the data movement differs (described for K = 5; K = 3: 9 taps, a ring of 3 rows, 6 columns per row):
//
//   dw_kernel: lane = 4 outputs x 8 channels straight from global memory.  Every input element is fetched and widened by
//   ten lanes, the 25 x 8 float32 tap weights are re-fetched per lane (more load instructions than the data itself), and
//   at 134-154 VGPRs three waves per SIMD cannot hide five dependent round trips: 1.1-1.4 TB/s (profiles/r02_perop_c5*).
//
//   here: one workgroup = 64 channels x a TW x TH output tile.  The halo tile is fetched ONCE with independent 16-byte
//   loads (all in flight together) and parked in LDS transposed to [row][channel pair][x]; a lane owns ONE channel pair
//   (2 x 25 taps = 50 VGPRs, fetched once) and a strip of 4 output columns and slides down its band of rows with a ring
//   of 5 x 4 float2 accumulators: per input row 4 ds_read_b64, 16 conversions and 100 packed FMAs on 20 independent
//   chains (the packed-FMA issue limit of tools/peak.hip does not bite).  Row pitch TWp = TW + 6 = 2 (mod 4) words makes
//   both the transposing writes (8 channel groups x 8 pixels of a wave -> 64 distinct banks) and the b64 reads (32 pairs x
//   pitch: 32 distinct even banks) conflict-free.
#include "yr_common.h"
#include <cstdlib>

struct DwlArgs {
    const void* in;      // [B][H][W][ld_in] 16-bit
    const float* w;      // [25][ld_w]
    const float* scale;
    const float* shift;
    void* out;           // [B][H][W][ld_out]
    int B, H, W, C8;     // C8 = ceil(C / 8): 16-byte channel vectors per pixel
    int ld_in, ld_w, ld_out;
    int pad_t, pad_l, act;
    int tw, twp, th;     // output tile, LDS row pitch in words
    int nstrip, nband, band_rows;
    int ntx, nty, ncc;   // tiles along x / y, 64-channel chunks
    int step_r, step_j;  // 32 = step_r * (tw + 4) + step_j
    int stage_u;         // 16-byte loads per lane and staging round (the halo tile's loads spread evenly over the rounds)
    unsigned nblocks;
    float* part;         // squeeze-excite form: [B][ntx * nty][ld_part] float32 channel sums of what each tile stored, or null
    int ld_part;
    int dbg;             // YR_DW_EXPERIMENT builds: 1 = no compute phase, 2 = no global loads while staging
};

typedef float dwl_f2 __attribute__((ext_vector_type(2)));

template <class T>
__device__ __forceinline__ dwl_f2 dwl_widen(unsigned v) {
    typedef T t2 __attribute__((ext_vector_type(2)));
    return __builtin_convertvector(__builtin_bit_cast(t2, v), dwl_f2);
}

// ACT: 0 ReLU6, 1 swish (the fast form of 16-bit stores, yr_apply_act_t), 2 whatever a.act says (a switch per value)
template <int ACT, class T>
__device__ __forceinline__ float dwl_act(float v, int act) {
    if constexpr (ACT == 0) return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f);
    else if constexpr (ACT == 1) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
    else return yr_apply_act_t<T>(v, act);
}

constexpr int DWL_STAGE_U = 10;   // 16-byte loads in flight per lane while staging: the usual tile (up to 320 halo pixels) in ONE round trip

// Both the halo tile's loads and the stores go through buffer descriptors (one image of the map each, < 2 GB): a lane whose
// pixel is padding, whose channels are beyond C or whose column is beyond W passes an offset beyond num_records - the load
// returns zeros, the store is dropped - so neither needs clamped coordinates, selects or a per-lane branch (round 3: the
// kernel is VALU-bound and more than half of its instructions were NOT the multiply-adds, profiles/r03_pmc_sq_c3*).
typedef __amdgpu_buffer_rsrc_t dwl_rsrc;
typedef unsigned dwl_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dwl_rsrc dwl_make_rsrc(const void* base, unsigned bytes) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
constexpr unsigned DWL_DEAD = 0x80000000u;

// UNI: the two (strip, band) slots of every wave share their band (an even number of strips): the row conditions of the
// sliding window are then wave-uniform and compile to scalar branches instead of exec masks.
template <class T, int K, int ACT, bool SE, bool UNI>
__global__ __launch_bounds__(256) void dwl_kernel(DwlArgs a) {
    constexpr int HALO = K - 1, KK = K * K, NQ = (4 + HALO) / 2;   // extra rows / columns of the window; taps; 8-byte reads per input row
    extern __shared__ unsigned dwl_tile[];   // [th + HALO][32][twp], then 4 * twp words nobody reads (where staging slots beyond the tile write)
    const unsigned lin = yr_xcd_swizzle(blockIdx.x, a.nblocks);
    // spatially adjacent tiles of one channel chunk are consecutive: their halos meet in one XCD's L2
    const int tx = (int)(lin % (unsigned)a.ntx);
    unsigned t = lin / (unsigned)a.ntx;
    const int ty = (int)(t % (unsigned)a.nty);
    t /= (unsigned)a.nty;
    const int cc = (int)(t % (unsigned)a.ncc);
    const int b = (int)(t / (unsigned)a.ncc);
    const int x0 = tx * a.tw, y0 = ty * a.th;
    const int rows_here = min(a.th, a.H - y0);
    const int tid = (int)threadIdx.x;

    // ---- compute-phase identity (needed first: the tap weights are fetched before the tile so that they are in flight too)
    const int cp = tid & 31;
    const int sb = tid >> 5;
    const int strip = sb % a.nstrip, band = sb / a.nstrip;
    const int cfirst = cc * 64 + cp * 2;
    const bool chan_ok = cfirst < a.C8 * 8;
    const int cl = chan_ok ? cfirst : 0;
    dwl_f2 w[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) w[k] = *reinterpret_cast<const dwl_f2*>(a.w + (size_t)k * a.ld_w + cl);
    const dwl_f2 sc = *reinterpret_cast<const dwl_f2*>(a.scale + cl);
    const dwl_f2 sh = *reinterpret_cast<const dwl_f2*>(a.shift + cl);

    // ---- stage the halo tile: thread = (channel vector cv, pixel slot); pixels walk the tile row-major in steps of 32
    {
        const int cv = tid & 7, slot = tid >> 3;
        const int cols = a.tw + HALO;
        const int npix = (rows_here + HALO) * cols;
        const bool cv_ok = cc * 8 + cv < a.C8;
        const dwl_rsrc src = dwl_make_rsrc(reinterpret_cast<const T*>(a.in) + (size_t)b * a.H * a.W * a.ld_in, (unsigned)(a.H * a.W * a.ld_in) * 2u);
        const unsigned cvb = (unsigned)(cc * 8 + cv) * 16u;
        const int sink = (a.th + HALO) * 32 * a.twp;
        int r = slot / cols, j = slot - r * cols;
        for (int p0 = 0; p0 < npix; p0 += 32 * a.stage_u) {
            dwl_u4 v[DWL_STAGE_U];
            int woff[DWL_STAGE_U];
#pragma unroll
            for (int u = 0; u < DWL_STAGE_U; ++u) {
                if (u < a.stage_u) {   // uniform
                    const int iy = y0 - a.pad_t + r, ix = x0 - a.pad_l + j;
                    const bool in_tile = p0 + slot + 32 * u < npix;
                    const bool ok = cv_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && in_tile;
                    unsigned voff = (unsigned)((iy * a.W + ix) * a.ld_in) * 2u + cvb;
#ifdef YR_DW_EXPERIMENT
                    if (a.dbg == 2) voff = DWL_DEAD;
#endif
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(src, ok ? voff : DWL_DEAD, 0, 0);
                    woff[u] = in_tile ? (r * 32 + cv * 4) * a.twp + j : sink;
                    j += a.step_j; r += a.step_r;          // 32 pixels on, row-major
                    if (j >= cols) { j -= cols; ++r; }
                }
            }
#pragma unroll
            for (int u = 0; u < DWL_STAGE_U; ++u) {
                if (u < a.stage_u) {
                    unsigned* d = dwl_tile + woff[u];
                    d[0] = v[u].x; d[a.twp] = v[u].y; d[2 * a.twp] = v[u].z; d[3 * a.twp] = v[u].w;
                }
            }
        }
    }
    __syncthreads();

    // ---- slide down the band: input row rr of the band feeds output rows rr - ky
    const int yb0 = band * a.band_rows;
    int nrows = min(a.band_rows, rows_here - yb0);
    if (band >= a.nband) nrows = 0;
#ifdef YR_DW_EXPERIMENT
    if (a.dbg == 1) nrows = 0;
#endif
    if constexpr (UNI) nrows = __builtin_amdgcn_readfirstlane(nrows);
    const int nin = nrows > 0 ? nrows + HALO : 0;
    const int xo = x0 + strip * 4;
    const dwl_rsrc dst = dwl_make_rsrc(reinterpret_cast<T*>(a.out) + (size_t)b * a.H * a.W * a.ld_out, (unsigned)(a.H * a.W * a.ld_out) * 2u);
    const unsigned opitch = (unsigned)(a.W * a.ld_out) * 2u;
    unsigned ooff[4];   // the lane's four output pixels within a row (bytes), dead beyond W / C
#pragma unroll
    for (int i = 0; i < 4; ++i) ooff[i] = chan_ok && xo + i < a.W ? (unsigned)((xo + i) * a.ld_out + cl) * 2u : DWL_DEAD;
    unsigned orow = (unsigned)(y0 + yb0) * opitch;
    const unsigned* trow = dwl_tile + (yb0 * 32 + cp) * a.twp + strip * 4;
    const int tpitch = 32 * a.twp;
    dwl_f2 psum = (dwl_f2){0.f, 0.f};   // SE: what this lane stored, per channel (rounded values, fixed order)
    dwl_f2 acc[K][4];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[s][i] = (dwl_f2){0.f, 0.f};
    for (int rr0 = 0; rr0 < nin; rr0 += K) {
#pragma unroll
        for (int ph = 0; ph < K; ++ph) {
            const int rr = rr0 + ph;
            if (rr < nin) {
                const uint2* p = reinterpret_cast<const uint2*>(trow + rr * tpitch);
                dwl_f2 col[2 * NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const uint2 v = p[q];
                    col[2 * q] = dwl_widen<T>(v.x);
                    col[2 * q + 1] = dwl_widen<T>(v.y);
                }
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const int s = (ph - ky + K) % K;
                    if (rr - ky >= 0 && rr - ky < nrows) {   // an output row of this band (the bands of a wave agree except at the tile's last rows)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx)
#pragma unroll
                            for (int i = 0; i < 4; ++i)   // the row's first tap starts the sum from a literal zero: no pass that clears the slot
                                acc[s][i] = __builtin_elementwise_fma(col[i + kx], w[ky * K + kx], ky == 0 && kx == 0 ? (dwl_f2){0.f, 0.f} : acc[s][i]);
                    }
                }
                const int sd = (ph + 1) % K;               // the slot of output row rr - HALO
                if (rr >= HALO) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        typedef T t2 __attribute__((ext_vector_type(2)));
                        const dwl_f2 y = __builtin_elementwise_fma(acc[sd][i], sc, sh);
                        const t2 r = __builtin_convertvector((dwl_f2){dwl_act<ACT, T>(y.x, a.act), dwl_act<ACT, T>(y.y, a.act)}, t2);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), dst, orow + ooff[i], 0, 0);
                        if constexpr (SE) psum += ooff[i] != DWL_DEAD ? __builtin_convertvector(r, dwl_f2) : (dwl_f2){0.f, 0.f};
                    }
                    orow += opitch;
                }
            }
        }
    }
    if constexpr (SE) {
        // the 8 (strip, band) slots of a channel pair are added in slot order by the first wave: one row of `part` per tile
        dwl_f2* red = reinterpret_cast<dwl_f2*>(dwl_tile);
        __syncthreads();   // every wave is done with the tile
        red[tid] = psum;
        __syncthreads();
        if (tid < 32 && chan_ok) {
            dwl_f2 sum = red[tid];
#pragma unroll
            for (int k = 1; k < 8; ++k) sum += red[k * 32 + tid];
            *reinterpret_cast<dwl_f2*>(a.part + ((size_t)b * (a.ntx * a.nty) + ty * a.ntx + tx) * a.ld_part + cfirst) = sum;
        }
    }
}

// Tile geometry.  A workgroup's 256 lanes are 32 channel pairs x 8 (strip, band) slots: tw / 4 column strips x as many row
// bands as still fit.  Among the tile widths, take the one with the least lane-time: workgroups x (rows a lane walks + the
// 4 extra input rows of its band + its share of staging the halo tile), the tile height being what a third of a CU's LDS
// holds.  40x40 -> 8 x 20 tiles in 4 bands of 5 rows, 20x20 -> 8 x 20, 13x13 -> 16 x 13 in 2 bands.
static void dwl_geometry(int H, int W, int K, DwlArgs* a) {
    const int halo = K - 1;
    // cost of one output row of a lane, of one warm-up row of its band (fewer taps land), of staging one halo pixel - fitted
    // to tools/dw5_probe.py for K = 5, scaled by the taps for K = 3
    const long long row_cost = K == 5 ? 8000 : 3600, warm_cost = K == 5 ? 1280 : 576;
    long long best = 0;
    for (int tw = 4; tw <= 32; tw += 4) {
        const int nstrip = tw / 4, twp = tw + 6;
        const int ntx = (W + tw - 1) / tw;
        const int row_bytes = 32 * twp * 4;
        int th = 48 * 1024 / row_bytes - halo;
        if (th > H) th = H;
        if (th < 4) th = 4;
        const int nty = (H + th - 1) / th;
        th = (H + nty - 1) / nty;
        int nband = 8 / nstrip;
        if (nband > th) nband = th;
        const int band_rows = (th + nband - 1) / nband;
        nband = (th + band_rows - 1) / band_rows;
        // integers: compiler.dwl_geometry must pick the same tile
        const long long cost = (long long)(ntx * nty) * (row_cost * band_rows + warm_cost * halo + 30 * (tw + halo) * (th + halo));
        if (best == 0 || cost < best) {
            best = cost;
            a->ntx = ntx; a->tw = tw; a->twp = twp; a->nstrip = nstrip;
            a->nty = nty; a->th = th; a->nband = nband; a->band_rows = band_rows;
        }
    }
}

template <class T, int K, bool SE, bool UNI>
static int launch_dwl_u(const DwlArgs& a, size_t lds, hipStream_t s) {
    const int actv = a.act == YR_ACT_RELU6 ? 0 : (a.act == YR_ACT_SWISH ? 1 : 2);
    static char nm[3][48];   // spelled like the symbol (element type, K, activation variant, SE, UNI): profiles are joined on it
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "dwl_kernel<%s,%d,0,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, (int)SE, (int)UNI) +
                              snprintf(nm[1], sizeof(nm[1]), "dwl_kernel<%s,%d,1,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, (int)SE, (int)UNI) +
                              snprintf(nm[2], sizeof(nm[2]), "dwl_kernel<%s,%d,2,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, (int)SE, (int)UNI);
    (void)nm_len;
    yr_note_kernel(nm[actv]);
    if (a.act == YR_ACT_RELU6) hipLaunchKernelGGL((dwl_kernel<T, K, 0, SE, UNI>), dim3(a.nblocks), dim3(256), lds, s, a);
    else if (a.act == YR_ACT_SWISH) hipLaunchKernelGGL((dwl_kernel<T, K, 1, SE, UNI>), dim3(a.nblocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((dwl_kernel<T, K, 2, SE, UNI>), dim3(a.nblocks), dim3(256), lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T, int K, bool SE>
static int launch_dwl_t(DwlArgs a, int expect_rows, hipStream_t s) {
    constexpr int HALO = K - 1;
    dwl_geometry(a.H, a.W, K, &a);
    if (SE) YR_REQUIRE(a.ntx * a.nty == expect_rows, "depthwise (LDS form): the SE partial-sum buffer must hold %d rows per image (has %d)", a.ntx * a.nty, expect_rows);
    YR_REQUIRE((long long)a.H * a.W * (a.ld_in > a.ld_out ? a.ld_in : a.ld_out) * 2 < (1ll << 31), "depthwise (LDS form): one image of the map must be below 2 GB");
    a.ncc = (a.C8 + 7) / 8;
    a.step_r = 32 / (a.tw + HALO); a.step_j = 32 % (a.tw + HALO);
    {
        const int items = ((a.th + HALO) * (a.tw + HALO) + 31) / 32, rounds = (items + DWL_STAGE_U - 1) / DWL_STAGE_U;
        a.stage_u = (items + rounds - 1) / rounds;
    }
    const long long blocks = (long long)a.B * a.ncc * a.nty * a.ntx;
    YR_REQUIRE(blocks < (1ll << 31), "depthwise: grid too large");
    a.nblocks = (unsigned)blocks;
    a.dbg = getenv("YR_DWL_DBG") ? atoi(getenv("YR_DWL_DBG")) : 0;
    const size_t lds = ((size_t)(a.th + HALO) * 32 + 4) * a.twp * 4;
    return a.nstrip % 2 == 0 ? launch_dwl_u<T, K, SE, true>(a, lds, s) : launch_dwl_u<T, K, SE, false>(a, lds, s);
}

template <class T>
static int launch_dwl_k(const DwlArgs& a, int k, int part_rows, hipStream_t s) {
    if (k == 5) return a.part ? launch_dwl_t<T, 5, true>(a, part_rows, s) : launch_dwl_t<T, 5, false>(a, 0, s);
    return a.part ? launch_dwl_t<T, 3, true>(a, part_rows, s) : launch_dwl_t<T, 3, false>(a, 0, s);
}

int yr_launch_depthwise_lds(int dtype, int k, const void* in, const float* w, const float* scale, const float* shift, void* out, int B, int H, int W,
                            int C8, int ld_in, int ld_w, int ld_out, int pad_t, int pad_l, int act, float* part, int ld_part, int part_rows,
                            hipStream_t s) {
    if (k != 3 && k != 5) { yr_set_error("depthwise (LDS form): 3 x 3 and 5 x 5 only"); return YR_ERR_ARG; }
    DwlArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.out = out;
    a.B = B; a.H = H; a.W = W; a.C8 = C8;
    a.ld_in = ld_in; a.ld_w = ld_w; a.ld_out = ld_out;
    a.pad_t = pad_t; a.pad_l = pad_l; a.act = act;
    a.part = part; a.ld_part = ld_part;
    if (dtype == YR_BF16) return launch_dwl_k<yr_bf16>(a, k, part_rows, s);
    if (dtype == YR_F16) return launch_dwl_k<yr_f16>(a, k, part_rows, s);
    yr_set_error("depthwise (LDS form): 16-bit maps only");
    return YR_ERR_ARG;
}
