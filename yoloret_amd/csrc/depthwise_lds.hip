// Depthwise K x K (K = 5 | 3) stride-1 convolution on 16-bit maps, LDS-tiled: the form for the wide EfficientNet stages whose
// blocks the fused kernels do not take (more than 128 block inputs: efficientnet.py:501-510 with kernel_size 5) and - K = 3 -
// for the detection heads' MBConv depthwise stages (code/yolo3/model.py:98-114: 52 x 52 x 128 ... 13 x 13 x 512,
// squeeze-excite form) and every other 16-bit 3 x 3 map with 64 channels or more.  Same arithmetic as dw_kernel<K,1,..> -
// float32 accumulation in (ky, kx) order, BatchNorm, activation, one rounding on store - so the two forms are
// bit-identical; only the data movement differs:
//
//   dw_kernel: lane = 4 outputs x 8 channels straight from global memory.  Every input element is fetched and widened by
//   ten lanes, the 25 x 8 float32 tap weights are re-fetched per lane (more load instructions than the data itself), and
//   at 134-154 VGPRs three waves per SIMD cannot hide five dependent round trips: 1.1-1.4 TB/s (profiles/r02_perop_c5*).
//
//   here (round 3; round 2's form - one workgroup per tile, the tile staged through registers and transposed into LDS -
//   ran fetch and compute back to back: on B0's 26 x 26 x 672 map, tools/dw5_probe.py with YR_DW_EXPERIMENT, fetching
//   alone 52 us, computing alone 74 us, together 93-101 us): a workgroup keeps its 64 channels (a lane owns ONE channel
//   pair: 2 x 25 taps = 50 VGPRs and the BatchNorm pair, loaded ONCE) and WALKS a contiguous run of (image, tile)
//   positions.  While it computes tile q out of one LDS buffer, tile q + 1 arrives in the other through LDS-direct buffer
//   loads (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write pass, zeros for padding from the descriptor's
//   range check; fetching alone: 31 us).  The DMA deposits 64 lanes x 16 bytes contiguously, so a tile buffer is
//   [halo pixel][32 channel pairs] - the map's own order - and a lane reads its channel pair of 4 + HALO neighbouring
//   pixels with ds_read2_b32 whose 32 lanes of a half-wave hit 32 consecutive banks (SQ_LDS_BANK_CONFLICT = 0).
//
//   The 256 lanes are 32 channel pairs x 8 (strip of 4 output columns, band of rows) slots.  The band walk is STRAIGHT-LINE
//   code: the rows of a tile are shared out evenly (bands of R or R - 1 rows, R a template parameter), so which taps an
//   input row feeds is known at compile time - no row conditions, no branches - and the taps of one input row are issued
//   kx-major across all the output rows they feed: 4 (hi - lo + 1) independent accumulator chains between two dependent
//   packed FMAs instead of 4 (a dependent v_pk_fma_f32 issues only every ~13th slot, tools/peak.hip).  The order of the
//   adds into any one output is unchanged - (ky, kx) ascending.  Stores leave through a buffer descriptor; rows or columns
//   beyond the map and channels beyond C pass an offset beyond num_records and are dropped.
#include "yr_common.h"
#include <cstdlib>

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

// Loads and stores go through buffer descriptors over ONE image of the map (< 1 GB): a lane whose pixel is padding, whose
// channels are beyond C or whose column is beyond W passes an offset beyond num_records - the load returns zeros, the store
// is dropped - so neither needs clamped coordinates, selects or a per-lane branch.
typedef __amdgpu_buffer_rsrc_t dwl_rsrc;
__device__ __forceinline__ dwl_rsrc dwl_make_rsrc(const void* base, unsigned bytes) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
constexpr unsigned DWL_DEAD = 0x80000000u;   // load offsets

struct DwpArgs {
    const void* in;      // [B][H][W][ld_in] 16-bit
    const float* w;      // [K*K][ld_w]
    const float* scale;
    const float* shift;
    void* out;           // [B][H][W][ld_out]
    int B, H, W, C8;     // C8 = ceil(C / 8): 16-byte channel vectors per pixel
    int ld_in, ld_w, ld_out;
    int pad_t, pad_l, act;
    int tw, th;          // output tile
    int cols, npix;      // halo tile: cols = tw + HALO columns, npix = (th + HALO) * cols pixels
    int rounds;          // DMA rounds (32 pixels each: 4 waves x 8 pixels x 8 channel vectors)
    int buf_words;       // one tile buffer in 32-bit words (pixels rounded up to the last wave that loads)
    int nstrip, nband;   // tw / 4 column strips x row bands
    int nbig;            // bands 0 .. nbig - 1 have R rows, the others R - 1
    int ntx, nty, ncc;   // tiles along x / y, 64-channel chunks
    int step_r, step_j;  // 32 = step_r * cols + step_j
    int G;               // walkers per 64-channel chunk
    unsigned nq;         // tile positions per chunk: B * nty * ntx
    unsigned nblocks;
    float* part;         // squeeze-excite form: [B][ntx * nty][ld_part] float32 channel sums of what each tile stored, or null
    int ld_part;
    int dbg;             // YR_DW_EXPERIMENT builds: 1 = no compute phase, 2 = no global loads
};

template <int V> struct dwp_int { static constexpr int value = V; };
template <int N, class F>
__device__ __forceinline__ void dwp_static_for(F&& f) {
    if constexpr (N > 0) {
        dwp_static_for<N - 1>(f);
        f(dwp_int<N - 1>{});
    }
}

// One band of R output rows: R + HALO input rows, fully unrolled and branch-free.  rows_ok (per lane): the rows of the band
// that exist - a band of R - 1 rows, or one cut by the map's last row, computes the others on whatever the tile buffer
// holds there (zeros beyond the map; LDS reads beyond the allocation return zeros) and stores them nowhere.
constexpr unsigned DWP_DEAD = 0x40000000u;   // store offsets: row part + pixel part, either may be dead, the sum must not wrap
constexpr int DWP_ROUNDS = 7;               // DMA rounds of 32 pixels per tile buffer (208 halo pixels at most)
template <class T, int K, int ACT, bool SE, int R>
__device__ __forceinline__ void dwp_band(const unsigned* trow, int tpitch, const dwl_f2 (&w)[K * K], dwl_f2 sc, dwl_f2 sh, int act, dwl_rsrc dst,
                                         unsigned orow, unsigned opitch, const unsigned (&ooff)[4], int rows_ok, dwl_f2& psum) {
    constexpr int HALO = K - 1, NC = 4 + HALO;
    dwl_f2 acc[K][4];
    unsigned raw[NC];   // the NEXT input row: its reads are issued a step ahead, under the taps of this one
#pragma unroll
    for (int c = 0; c < NC; ++c) raw[c] = trow[c * 32];
    dwp_static_for<R + HALO>([&](auto RR) {
        constexpr int rr = decltype(RR)::value;
        constexpr int lo = rr - R + 1 > 0 ? rr - R + 1 : 0, hi = rr < HALO ? rr : HALO;   // input row rr feeds output rows rr - lo ... rr - hi
        dwl_f2 col[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) col[c] = dwl_widen<T>(raw[c]);
        if constexpr (rr + 1 < R + HALO) {
            const unsigned* p = trow + (rr + 1) * tpitch;
#pragma unroll
            for (int c = 0; c < NC; ++c) raw[c] = p[c * 32];
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int ky = lo; ky <= hi; ++ky)
#pragma unroll
                for (int i = 0; i < 4; ++i)   // an output row's first tap starts its sum from a literal zero
                    acc[(rr - ky) % K][i] = __builtin_elementwise_fma(col[i + kx], w[ky * K + kx], ky == 0 && kx == 0 ? (dwl_f2){0.f, 0.f} : acc[(rr - ky) % K][i]);
        if constexpr (rr >= HALO) {
            constexpr int o = rr - HALO, sd = o % K;
            const unsigned ro = o < rows_ok ? orow + (unsigned)o * opitch : DWP_DEAD;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                typedef T t2 __attribute__((ext_vector_type(2)));
                const dwl_f2 y = __builtin_elementwise_fma(acc[sd][i], sc, sh);
                const t2 r = __builtin_convertvector((dwl_f2){dwl_act<ACT, T>(y.x, act), dwl_act<ACT, T>(y.y, act)}, t2);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), dst, ro + ooff[i], 0, 0);
                if constexpr (SE) psum += ro + ooff[i] < DWP_DEAD ? __builtin_convertvector(r, dwl_f2) : (dwl_f2){0.f, 0.f};
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler hoists the reads and conversions of all rows to the top)
    });
}

template <class T, int K, int ACT, bool SE, int R>
__global__ __launch_bounds__(256) void dwp_kernel(DwpArgs a) {
    constexpr int KK = K * K;
    extern __shared__ unsigned dwp_lds[];   // two tile buffers; SE: then 2 x 256 float2 of partial sums
    const unsigned lin = yr_xcd_swizzle(blockIdx.x, a.nblocks);
    const int cc = (int)(lin / (unsigned)a.G), g = (int)(lin % (unsigned)a.G);
    unsigned q = (unsigned)((unsigned long long)g * a.nq / (unsigned)a.G);
    const unsigned q1 = (unsigned)((unsigned long long)(g + 1) * a.nq / (unsigned)a.G);
    if (q >= q1) return;
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    // ---- compute-phase identity: 32 channel pairs x 8 (strip, band) slots, slot = 2 wave + half-wave.  With 2 or 4 strips a
    // wave is ONE strip (two bands): in the tiles at the map's right edge whole waves have nothing to do and skip the walk
    const int cp = tid & 31;
    const int sb = tid >> 5;
    int strip, band;
    if (a.nstrip <= 4) { strip = (sb >> 1) % a.nstrip; band = (sb >> 1) / a.nstrip * 2 + (sb & 1); }
    else { strip = sb; band = 0; }
    const bool slot_ok = strip < a.nstrip && band < a.nband;
    const int cfirst = cc * 64 + cp * 2;
    const bool chan_ok = cfirst < a.C8 * 8;
    const int cl = chan_ok ? cfirst : 0;
    dwl_f2 w[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) w[k] = *reinterpret_cast<const dwl_f2*>(a.w + (size_t)k * a.ld_w + cl);
    const dwl_f2 sc = *reinterpret_cast<const dwl_f2*>(a.scale + cl);
    const dwl_f2 sh = *reinterpret_cast<const dwl_f2*>(a.shift + cl);

    // ---- DMA identity: lane = (pixel of the wave's group of 8, channel vector)
    const int cv = lane & 7;
    const bool cv_ok = cc * 8 + cv < a.C8;
    const unsigned cvb = (unsigned)(cc * 8 + cv) * 16u;
    const int pl0 = wave * 8 + (lane >> 3);
    // per DMA round (at most DWP_ROUNDS: 208 halo pixels / 32): this lane's halo pixel (row, column) and its byte offset from the
    // tile's origin - the same for every tile of the walk, so a fetch costs two adds, two compares and a select per round.
    // A slot beyond the tile (or channels beyond C) gets a column no map has.
    int rr_[DWP_ROUNDS], jj_[DWP_ROUNDS];
    unsigned rel_[DWP_ROUNDS];
    {
        int r = pl0 / a.cols, j = pl0 - r * a.cols;
#pragma unroll
        for (int u = 0; u < DWP_ROUNDS; ++u) {
            const bool slot = cv_ok && u * 32 + pl0 < a.npix;
            rr_[u] = r; jj_[u] = slot ? j : 0x7fff0000;
            rel_[u] = (unsigned)((r * a.W + j) * a.ld_in) * 2u + cvb;
            j += a.step_j; r += a.step_r;
            if (j >= a.cols) { j -= a.cols; ++r; }
        }
    }
    auto fetch = [&](int tx, int ty, int b, int buf) {
        const dwl_rsrc src = dwl_make_rsrc(reinterpret_cast<const T*>(a.in) + (size_t)b * a.H * a.W * a.ld_in, (unsigned)(a.H * a.W * a.ld_in) * 2u);
        const int iy0 = ty * a.th - a.pad_t, ix0 = tx * a.tw - a.pad_l;
        const unsigned tbase = (unsigned)((iy0 * a.W + ix0) * a.ld_in) * 2u;   // (may wrap below zero: the sums of the pixels inside the map do not)
#pragma unroll
        for (int u = 0; u < DWP_ROUNDS; ++u) {
            const int pw = u * 32 + wave * 8;   // the wave's first pixel of this round
            if (pw < a.npix) {                  // (wave-uniform: a wave of the last round wholly beyond the tile loads nothing)
                const bool ok = (unsigned)(iy0 + rr_[u]) < (unsigned)a.H && (unsigned)(ix0 + jj_[u]) < (unsigned)a.W;
                unsigned voff = tbase + rel_[u];
#ifdef YR_DW_EXPERIMENT
                if (a.dbg == 2) voff = DWL_DEAD;
#endif
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (__attribute__((address_space(3))) void*)(dwp_lds + buf * a.buf_words + pw * 32), 16,
                                                         ok ? voff : DWL_DEAD, 0, 0, 0);
            }
        }
    };

    // ---- the walk
    int tx = (int)(q % (unsigned)a.ntx);
    unsigned tq = q / (unsigned)a.ntx;
    int ty = (int)(tq % (unsigned)a.nty);
    int b = (int)(tq / (unsigned)a.nty);
    const int xs = strip * 4;
    const int yb0 = band * (R - 1) + min(band, a.nbig);
    const int band_rows = slot_ok ? (band < a.nbig ? R : R - 1) : 0;
    const unsigned* tslot = dwp_lds + (yb0 * a.cols + xs) * 32 + cp;
    const int tpitch = a.cols * 32;
    const unsigned opitch = (unsigned)(a.W * a.ld_out) * 2u;
    int buf = 0;
    int ptile = -1, pb = 0;   // SE: the tile whose partial sums wait in `red` for the next barrier
    bool stored = false;      // wave-uniform: the walk of the last tile issued its 4 R stores behind the fetch of this one
    fetch(tx, ty, b, 0);
    for (; q < q1; ++q) {
        // this wave's share of tile q has landed: memory operations complete in order, so all but the 4 R stores issued after
        // its loads (waiting for vmcnt(0) would add the round trip of stores nobody waits for to every tile)
        if (stored) __builtin_amdgcn_s_waitcnt(((4 * R) & 15) | (((4 * R) >> 4) << 14) | 0x0f70);
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();                      // everybody's share has; everybody is done with the other buffer
        if constexpr (SE) {
            if (ptile >= 0 && tid < 32 && chan_ok) {
                const dwl_f2* red = reinterpret_cast<const dwl_f2*>(dwp_lds + 2 * a.buf_words) + (buf ^ 1) * 256;
                dwl_f2 sum = red[tid];
#pragma unroll
                for (int k = 1; k < 8; ++k) sum += red[k * 32 + tid];
                *reinterpret_cast<dwl_f2*>(a.part + ((size_t)pb * (a.ntx * a.nty) + ptile) * a.ld_part + cfirst) = sum;
            }
        }
        int ntx_ = tx + 1, nty_ = ty, nb_ = b;
        if (ntx_ == a.ntx) { ntx_ = 0; ++nty_; if (nty_ == a.nty) { nty_ = 0; ++nb_; } }
        if (q + 1 < q1) fetch(ntx_, nty_, nb_, buf ^ 1);
        dwl_f2 psum = (dwl_f2){0.f, 0.f};
        const int xo = tx * a.tw + xs, yo = ty * a.th + yb0;
        const int rows_ok = xo < a.W ? min(band_rows, a.H - yo) : 0;
        bool live = rows_ok > 0;
#ifdef YR_DW_EXPERIMENT
        if (a.dbg == 1) live = false;
#endif
        stored = __builtin_amdgcn_ballot_w64(live) != 0;   // wave-uniform
        if (stored) {
            const dwl_rsrc dst = dwl_make_rsrc(reinterpret_cast<T*>(a.out) + (size_t)b * a.H * a.W * a.ld_out, (unsigned)(a.H * a.W * a.ld_out) * 2u);
            unsigned ooff[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ooff[i] = chan_ok && xo + i < a.W ? (unsigned)((xo + i) * a.ld_out + cl) * 2u : DWP_DEAD;
            dwp_band<T, K, ACT, SE, R>(tslot + buf * a.buf_words, tpitch, w, sc, sh, a.act, dst, (unsigned)yo * opitch, opitch, ooff, rows_ok, psum);
        }
        if constexpr (SE) {
            reinterpret_cast<dwl_f2*>(dwp_lds + 2 * a.buf_words)[buf * 256 + tid] = psum;   // added up after the next barrier
            ptile = ty * a.ntx + tx; pb = b;
        }
        tx = ntx_; ty = nty_; b = nb_;
        buf ^= 1;
    }
    if constexpr (SE) {
        __syncthreads();
        if (tid < 32 && chan_ok) {
            const dwl_f2* red = reinterpret_cast<const dwl_f2*>(dwp_lds + 2 * a.buf_words) + (buf ^ 1) * 256;
            dwl_f2 sum = red[tid];
#pragma unroll
            for (int k = 1; k < 8; ++k) sum += red[k * 32 + tid];
            *reinterpret_cast<dwl_f2*>(a.part + ((size_t)pb * (a.ntx * a.nty) + ptile) * a.ld_part + cfirst) = sum;
        }
    }
}

// Tile geometry of the walking form.  Three workgroups per CU: two tile buffers of at most 208 halo pixels x 128 bytes each
// (192 with the squeeze-excite partial sums behind them).  Tile widths with an even number of 4-column strips (a wave =
// one band); the rows of a tile go to the bands evenly, R = the larger share.  Among the widths take the one with the least
// lane-time per map: tiles x (per output row: the taps + BatchNorm, activation, store; per input row of a band:
// conversions and LDS reads; per DMA round; a fixed cost per tile) - instruction counts of the kernel, not fitted.
constexpr int DWP_MAX_R = 6;
static bool dwp_geometry(int H, int W, int K, bool se, DwpArgs* a, int* R) {
    const int halo = K - 1;
    const int cap = se ? 192 : 208;
    const int row_cost = K * K * 4 + 21, in_cost = 3 * (4 + halo), round_cost = 12, tile_cost = 60;
    const char* force = getenv("YR_DWL_TW");
    long long best = 0;
    for (int tw = 8; tw <= 32; tw += 8) {
        if (force && atoi(force) != tw) continue;
        const int cols = tw + halo, nstrip = tw / 4;
        int th = cap / cols - halo;
        if (th < 1) continue;
        if (th > H) th = H;
        const int nty = (H + th - 1) / th;
        th = (H + nty - 1) / nty;
        const int ntx = (W + tw - 1) / tw;
        int nband = 8 / nstrip;
        if (nband > th) nband = th;
        const int r = (th + nband - 1) / nband;
        if (r > DWP_MAX_R) continue;
        const int npix = (th + halo) * cols, rounds = (npix + 31) / 32;
        const long long cost = (long long)(ntx * nty) * (row_cost * r + in_cost * (r + halo) + round_cost * rounds + tile_cost);
        if (best == 0 || cost < best) {
            best = cost;
            a->tw = tw; a->th = th; a->cols = cols; a->npix = npix; a->rounds = rounds;
            a->nstrip = nstrip; a->nband = nband; a->nbig = th - nband * (r - 1); a->ntx = ntx; a->nty = nty;
            *R = r;
        }
    }
    return best != 0;
}

template <class T, int K, bool SE, int R>
static int launch_dwp_r(const DwpArgs& a, size_t lds, hipStream_t s) {
    const int actv = a.act == YR_ACT_RELU6 ? 0 : (a.act == YR_ACT_SWISH ? 1 : 2);
    static char nm[3][48];   // spelled like the symbol (element type, K, activation variant, SE, R): profiles are joined on it
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "dwp_kernel<%s,%d,0,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, (int)SE, R) +
                              snprintf(nm[1], sizeof(nm[1]), "dwp_kernel<%s,%d,1,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, (int)SE, R) +
                              snprintf(nm[2], sizeof(nm[2]), "dwp_kernel<%s,%d,2,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), K, (int)SE, R);
    (void)nm_len;
    yr_note_kernel(nm[actv]);
    if (a.act == YR_ACT_RELU6) hipLaunchKernelGGL((dwp_kernel<T, K, 0, SE, R>), dim3(a.nblocks), dim3(256), lds, s, a);
    else if (a.act == YR_ACT_SWISH) hipLaunchKernelGGL((dwp_kernel<T, K, 1, SE, R>), dim3(a.nblocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((dwp_kernel<T, K, 2, SE, R>), dim3(a.nblocks), dim3(256), lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T, int K, bool SE>
static int launch_dwp_t(DwpArgs a, int expect_rows, hipStream_t s) {
    int R = 0;
    YR_REQUIRE(dwp_geometry(a.H, a.W, K, SE, &a, &R), "depthwise (LDS form): no tile fits");
    if (SE) YR_REQUIRE(a.ntx * a.nty == expect_rows, "depthwise (LDS form): the SE partial-sum buffer must hold %d rows per image (has %d)", a.ntx * a.nty, expect_rows);
    YR_REQUIRE((long long)a.H * a.W * (a.ld_in > a.ld_out ? a.ld_in : a.ld_out) * 2 < (1ll << 30), "depthwise (LDS form): one image of the map must be below 1 GB");
    a.ncc = (a.C8 + 7) / 8;
    a.step_r = 32 / a.cols; a.step_j = 32 % a.cols;
    a.buf_words = (a.npix + 7) / 8 * 8 * 32;
    const long long nq = (long long)a.B * a.nty * a.ntx;
    YR_REQUIRE(nq < (1ll << 31), "depthwise: grid too large");
    a.nq = (unsigned)nq;
    // one generation of workgroups: 3 per CU, shared out among the channel chunks; each walks its run of tile positions
    static const int slots = getenv("YR_DWP_SLOTS") ? atoi(getenv("YR_DWP_SLOTS")) : 3 * 256;
    long long G = slots / a.ncc;
    if (G < 1) G = 1;
    if (G > nq) G = nq;
    a.G = (int)G;
    a.nblocks = (unsigned)(G * a.ncc);
    a.dbg = getenv("YR_DWL_DBG") ? atoi(getenv("YR_DWL_DBG")) : 0;
    const size_t lds = (size_t)a.buf_words * 8 + (SE ? 4096 : 0);
    switch (R) {
        case 1: return launch_dwp_r<T, K, SE, 1>(a, lds, s);
        case 2: return launch_dwp_r<T, K, SE, 2>(a, lds, s);
        case 3: return launch_dwp_r<T, K, SE, 3>(a, lds, s);
        case 4: return launch_dwp_r<T, K, SE, 4>(a, lds, s);
        case 5: if constexpr (K == 3) return launch_dwp_r<T, K, SE, 5>(a, lds, s);
        case 6: if constexpr (K == 3) return launch_dwp_r<T, K, SE, 6>(a, lds, s);
    }
    yr_set_error("depthwise (LDS form): band of %d rows", R);
    return YR_ERR_ARG;
}

template <class T>
static int launch_dwp_k(const DwpArgs& a, int k, int part_rows, hipStream_t s) {
    if (k == 5) return a.part ? launch_dwp_t<T, 5, true>(a, part_rows, s) : launch_dwp_t<T, 5, false>(a, 0, s);
    return a.part ? launch_dwp_t<T, 3, true>(a, part_rows, s) : launch_dwp_t<T, 3, false>(a, 0, s);
}

int yr_launch_depthwise_lds(int dtype, int k, const void* in, const float* w, const float* scale, const float* shift, void* out, int B, int H, int W,
                            int C8, int ld_in, int ld_w, int ld_out, int pad_t, int pad_l, int act, float* part, int ld_part, int part_rows,
                            hipStream_t s) {
    if (k != 3 && k != 5) { yr_set_error("depthwise (LDS form): 3 x 3 and 5 x 5 only"); return YR_ERR_ARG; }
    DwpArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.out = out;
    a.B = B; a.H = H; a.W = W; a.C8 = C8;
    a.ld_in = ld_in; a.ld_w = ld_w; a.ld_out = ld_out;
    a.pad_t = pad_t; a.pad_l = pad_l; a.act = act;
    a.part = part; a.ld_part = ld_part;
    if (dtype == YR_BF16) return launch_dwp_k<yr_bf16>(a, k, part_rows, s);
    if (dtype == YR_F16) return launch_dwp_k<yr_f16>(a, k, part_rows, s);
    yr_set_error("depthwise (LDS form): 16-bit maps only");
    return YR_ERR_ARG;
}
