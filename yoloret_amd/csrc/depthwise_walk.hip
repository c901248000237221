// Depthwise K x K (K = 5) stride-1 convolution on 16-bit maps, WALKING form (round 4): the successor of depthwise_lds.hip's
// tile walk for the wide EfficientNet stages (efficientnet.py:501-510, kernel_size 5, more than 128 block inputs).  The plans take
// it for the 5 x 5 maps; the 3 x 3 maps of the detection heads (code/yolo3/model.py:98-114) stay on the tile walk, which is at the
// memory system's rate there (depthwise.hip: launch_depthwise_t; the K = 3 instances of this kernel were probed - bit-identical, 35 us against 32 on 52 x 52 x 128 -
// and are not built).  Same arithmetic as dw_kernel<K,1,..> and dwp_kernel -
// float32 accumulation in (ky, kx) order, BatchNorm, activation, one rounding on store: the three forms are bit-identical.
//
// What the tile walk paid for (profiles/r03_traffic_*): a 13 x 8 output tile needs a 17 x 12 halo tile -
// 1.96 x the pixels, of which the memory side saw 1.46-1.6 x - and its lanes were bands of R or R - 1 rows of 4-column strips:
// 13 rows in bands of 4 + 3 + 3 + 3 run 16 rows of instructions, 26 columns in 8-wide tiles run 32.  Here a workgroup owns 64
// channels of a COLUMN BLOCK (at most 8 four-column strips: the whole width of a 13-, 20- or 26-wide map, half of a 52-wide
// one; two images where both fit the 8 half-wave slots) and walks DOWN a segment of its rows: a half-wave = one strip x 32
// channel pairs, the K x 4 partial sums of the K output rows an input row feeds live in registers (5 x 4 packed pairs), every
// input row is read from LDS once and every output row is finished exactly once - no band remainders, no rows recomputed inside a segment, no vertical halo
// inside a segment.  Rows arrive in groups of K through LDS-direct buffer loads (`buffer_load_dwordx4 ... lds`, zeros for
// padding from the descriptor's range check), two group buffers, one barrier per K rows; the rows of a group are straight-line
// code (the accumulator an input row's tap feeds is a compile-time index), the first and the last groups of a segment run a
// guarded copy (taps whose output row lies outside the segment are skipped by wave-uniform branches).
//
// Row segments: the rows of an image are cut into `nq` quanta fixed by the map's SHAPE; a launch cuts every image into segments
// of whole quanta so that the workgroups fill the chip (a segment re-reads and re-widens K - 1 halo rows but recomputes no tap).
// The squeeze-excite form writes one row of channel sums per (column block, quantum) - independent of the segmentation, so the
// sums (and everything behind them) do not depend on the batch size.
#include "yr_common.h"
#include <cstdlib>

typedef float dwq_f2 __attribute__((ext_vector_type(2)));

template <class T>
__device__ __forceinline__ dwq_f2 dwq_widen(unsigned v) {
    typedef T t2 __attribute__((ext_vector_type(2)));
    return __builtin_convertvector(__builtin_bit_cast(t2, v), dwq_f2);
}

// ACT: 0 ReLU6, 1 swish (the fast form of 16-bit stores, yr_apply_act_t), 2 whatever a.act says (a switch per value)
template <int ACT, class T>
__device__ __forceinline__ float dwq_act(float v, int act) {
    if constexpr (ACT == 0) return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f);
    else if constexpr (ACT == 1) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
    else return yr_apply_act_t<T>(v, act);
}

typedef __amdgpu_buffer_rsrc_t dwq_rsrc;
__device__ __forceinline__ dwq_rsrc dwq_make_rsrc(const void* base, unsigned bytes) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
constexpr unsigned DWQ_LDEAD = 0x80000000u;   // load offsets beyond any descriptor: the DMA writes zeros
constexpr unsigned DWQ_SDEAD = 0x40000000u;   // store offsets: row part (< 2^30) + pixel part (dead or < 2^30) never wraps

struct DwqArgs {
    const void* in;      // [B][H][W][ld_in] 16-bit
    const float* w;      // [K*K][ld_w]
    const float* scale;
    const float* shift;
    void* out;           // [B][H][W][ld_out]
    int B, H, W, C8;     // C8 = ceil(C / 8): 16-byte channel vectors per pixel
    int ld_in, ld_w, ld_out;
    int pad_t, pad_l, act;
    int ni, lni;         // images per workgroup (1 | 2) and its log2
    int spb, nblk;       // 4-column strips per column block, column blocks per image
    int colsp, U;        // pixels of one LDS row (spb * 4 + HALO rounded up to 8), colsp / 8
    int Q, nq;           // rows per quantum, quanta per image (fixed by H)
    int qps, nseg;       // quanta per segment, segments per image
    int ncc, nig;        // 64-channel chunks, image groups
    int nw;              // waves per workgroup
    int step_a, step_b;  // a wave's DMA slots are nw apart: nw / U rows and nw % U pixel groups
    int gwords;          // one group buffer in 32-bit words: K * ni * colsp * 32
    unsigned nblocks;
    float* part;         // squeeze-excite form: [B][nblk * nq][ld_part] float32 channel sums of what each (block, quantum) stored, or null
    int ld_part;
};

template <int V> struct dwq_int { static constexpr int value = V; };
template <int N, class F>
__device__ __forceinline__ void dwq_static_for(F&& f) {
    if constexpr (N > 0) {
        dwq_static_for<N - 1>(f);
        f(dwq_int<N - 1>{});
    }
}

// One group of K input rows (t0 = the first one's index within the segment, a multiple of K).  Input row t feeds the output
// rows t - ky (ky = 0 .. HALO) of the segment, whose partial sums sit in acc[(t - ky) % K]; after its taps output row
// t - HALO is complete.  GUARD: rows of the first group (t < HALO: the output rows above the segment do not exist), of the
// last ones (t >= n: those below) and rows beyond the segment's last input row take only the taps that exist.
template <class T, int K, int ACT, bool SE, bool GUARD>
__device__ __forceinline__ void dwq_group(const unsigned* lrow, int rpitch, dwq_f2 (&acc)[K][4], const dwq_f2 (&w)[K * K], dwq_f2 sc, dwq_f2 sh, int act,
                                          dwq_rsrc dst, unsigned& orow, unsigned opitch, const unsigned (&ooff)[4], int t0, int n, dwq_f2& psum, dwq_f2& psum2) {
    constexpr int HALO = K - 1, NC = 4 + HALO;
    unsigned raw[NC];   // the NEXT input row: its reads are issued a step ahead, under the taps of this one
#pragma unroll
    for (int c = 0; c < NC; ++c) raw[c] = lrow[c * 32];
    dwq_static_for<K>([&](auto UU) {
        constexpr int u = decltype(UU)::value;
        const int t = t0 + u;
        dwq_f2 col[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) col[c] = dwq_widen<T>(raw[c]);
        if constexpr (u + 1 < K) {
            const unsigned* p = lrow + (u + 1) * rpitch;
#pragma unroll
            for (int c = 0; c < NC; ++c) raw[c] = p[c * 32];
        }
        if constexpr (!GUARD) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int ky = 0; ky <= HALO; ++ky)
#pragma unroll
                    for (int i = 0; i < 4; ++i)   // an output row's first tap starts its sum from a literal zero
                        acc[(u - ky + K) % K][i] = __builtin_elementwise_fma(col[i + kx], w[ky * K + kx], ky == 0 && kx == 0 ? (dwq_f2){0.f, 0.f} : acc[(u - ky + K) % K][i]);
        } else {
            const int lo = t - n + 1, hi = t < HALO ? t : HALO;   // (wave-uniform)
#pragma unroll
            for (int ky = 0; ky <= HALO; ++ky)
                if (ky >= lo && ky <= hi) {
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[(u - ky + K) % K][i] = __builtin_elementwise_fma(col[i + kx], w[ky * K + kx], ky == 0 && kx == 0 ? (dwq_f2){0.f, 0.f} : acc[(u - ky + K) % K][i]);
                }
        }
        const bool done = GUARD ? (t >= HALO && t < n + HALO) : true;   // output row t - HALO of the segment is complete
        if (done) {
            constexpr int sd = (u + 1) % K;   // (u - HALO) mod K
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                typedef T t2 __attribute__((ext_vector_type(2)));
                const dwq_f2 y = __builtin_elementwise_fma(acc[sd][i], sc, sh);
                const t2 r = __builtin_convertvector((dwq_f2){dwq_act<ACT, T>(y.x, act), dwq_act<ACT, T>(y.y, act)}, t2);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), dst, ooff[i], orow, 0);   // (the row: a scalar offset)
                if constexpr (SE) {   // (the squeeze-excite sums: what was stored; a quantum of rows ends behind row K - 2 of a group)
                    const dwq_f2 v = ooff[i] < DWQ_SDEAD ? __builtin_convertvector(r, dwq_f2) : (dwq_f2){0.f, 0.f};
                    if constexpr (u == K - 1) psum2 += v; else psum += v;
                }
            }
            orow += opitch;
        }
        __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler hoists the reads and conversions of all rows to the top)
    });
}

template <class T, int K, int ACT, bool SE>
// (launch bounds: the generic activation's switch needs more registers than 3 waves per SIMD leave)
__global__ __launch_bounds__(256, K == 5 ? (ACT == 2 ? 2 : 3) : 4) void dwq_kernel(DwqArgs a) {
    constexpr int KK = K * K, HALO = K - 1;
    extern __shared__ unsigned dwq_lds[];   // two group buffers; SE: then 2 x 256 float2 of slot sums
    unsigned lin = yr_xcd_swizzle(blockIdx.x, a.nblocks);
    const int seg = (int)(lin % (unsigned)a.nseg); lin /= (unsigned)a.nseg;
    const int blk = (int)(lin % (unsigned)a.nblk); lin /= (unsigned)a.nblk;
    const int ig = (int)(lin % (unsigned)a.nig);
    const int cc = (int)(lin / (unsigned)a.nig);
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    // ---- the segment: output rows o0 .. o0 + n - 1 of every image of the group
    const int q0 = seg * a.qps;
    const int o0 = q0 * a.Q;
    const int o1 = min(a.H, (q0 + a.qps) * a.Q);
    const int n = o1 - o0;
    const int img0 = ig * a.ni;
    const unsigned imgbytes_in = (unsigned)(a.H * a.W * a.ld_in) * 2u, imgbytes_out = (unsigned)(a.H * a.W * a.ld_out) * 2u;
    const int nimg = min(a.ni, a.B - img0);

    // ---- compute identity: half-wave = (image of the group, strip), lane = channel pair
    const int cp = tid & 31, sl = tid >> 5;
    int isl = sl / a.spb;
    const int st = sl - isl * a.spb;
    const bool slot_ok = isl < a.ni;
    if (!slot_ok) isl = 0;
    const int cfirst = cc * 64 + cp * 2;
    const bool chan_ok = cfirst < a.C8 * 8;
    const int cl = chan_ok ? cfirst : 0;
    dwq_f2 w[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) w[k] = *reinterpret_cast<const dwq_f2*>(a.w + (size_t)k * a.ld_w + cl);
    const dwq_f2 sc = *reinterpret_cast<const dwq_f2*>(a.scale + cl);
    const dwq_f2 sh = *reinterpret_cast<const dwq_f2*>(a.shift + cl);

    // ---- DMA identity: lane = (pixel of the slot's group of 8, channel vector); a slot = (row of the group x image, 8 pixels)
    const int cv = lane & 7, px = lane >> 3;
    const bool cv_ok = cc * 8 + cv < a.C8;
    const int xb0 = blk * a.spb * 4;                                   // the block's first output column
    const int xl = xb0 - a.pad_l + px;                                  // this lane's map column in pixel group 0
    const unsigned lane_part = (unsigned)(px * a.ld_in) * 2u + (unsigned)(cc * 8 + cv) * 16u;
    const dwq_rsrc src = dwq_make_rsrc(reinterpret_cast<const T*>(a.in) + (size_t)img0 * a.H * a.W * a.ld_in, (unsigned)nimg * imgbytes_in);
    const int rowwords = a.colsp * 32;
    // A wave's DMA slots of a group - (row of the group x image, 8 pixels), every nw-th of the K * ni * U - are the same in every
    // group: their LDS word, row, byte offset from the group's first row and this lane's column check are worked out once
    // (scalars of the wave; the column checks one bit per slot), so a slot costs a handful of scalar operations per group.
    constexpr int MAXS = 10;   // K * ni * U / nw at most (dwq_geometry)
    int s_lds[MAXS], s_r[MAXS];
    unsigned s_off[MAXS], colmask = 0;
    {
        const int q = wave / a.U;
        int ri = q, u = wave - q * a.U;
        const int nri = K * a.ni;
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            const int r = ri >> a.lni, ii = ri & (a.ni - 1);
            s_r[j] = ri < nri ? r : 0x10000;   // (a slot beyond the group: a row no segment has)
            s_lds[j] = ri * rowwords + u * 256;
            s_off[j] = (unsigned)((r * a.W + xb0 - a.pad_l + u * 8) * a.ld_in) * 2u + (unsigned)ii * imgbytes_in;   // (may wrap below zero: the sums of pixels inside the map do not)
            colmask |= (cv_ok && (unsigned)(xl + u * 8) < (unsigned)a.W ? 1u : 0u) << j;
            u += a.step_b; ri += a.step_a;
            if (u >= a.U) { u -= a.U; ++ri; }
        }
    }
    const unsigned rowbytes_in = (unsigned)(a.W * a.ld_in) * 2u;
    auto issue = [&](int g, int buf) {
        const int t0 = g * K, iy0 = o0 - a.pad_t + t0;
        const unsigned gbase = (unsigned)iy0 * rowbytes_in;
        const unsigned* lbuf = dwq_lds + buf * a.gwords;
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            if (t0 + s_r[j] < n + HALO) {   // (wave-uniform; rows beyond the segment's last input row are never used)
                const bool ok = (unsigned)(iy0 + s_r[j]) < (unsigned)a.H && ((colmask >> j) & 1u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (__attribute__((address_space(3))) void*)(lbuf + s_lds[j]), 16,
                                                         ok ? gbase + s_off[j] + lane_part : DWQ_LDEAD, 0, 0, 0);
            }
        }
    };

    // ---- the walk
    const dwq_rsrc dst = dwq_make_rsrc(reinterpret_cast<T*>(a.out) + (size_t)img0 * a.H * a.W * a.ld_out, (unsigned)nimg * imgbytes_out);
    const unsigned opitch = (unsigned)(a.W * a.ld_out) * 2u;
    unsigned ooff[4];
    {
        const int xo = xb0 + st * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ooff[i] = slot_ok && chan_ok && xo + i < a.W ? (unsigned)isl * imgbytes_out + (unsigned)((xo + i) * a.ld_out + cl) * 2u : DWQ_SDEAD;
    }
    unsigned orow = (unsigned)o0 * opitch;
    const unsigned* lslot = dwq_lds + isl * rowwords + st * 4 * 32 + cp;
    const int rpitch = a.ni * rowwords;
    unsigned* red = dwq_lds + 2 * a.gwords;
    const int part_row0 = blk * a.nq + q0;
    dwq_f2 acc[K][4];
    dwq_f2 psum = (dwq_f2){0.f, 0.f}, psum2 = (dwq_f2){0.f, 0.f};
    auto flush = [&](int qi, int par) {   // one row of `part`: the slots' sums of quantum qi, added up in a fixed order
        dwq_f2* rd = reinterpret_cast<dwq_f2*>(red) + par * 256;
        rd[tid] = psum;
        __syncthreads();
        if (tid < a.ni * 32 && chan_ok && img0 + (tid >> 5) < a.B) {
            const dwq_f2* q = rd + (tid >> 5) * a.spb * 32 + (tid & 31);
            dwq_f2 sum = q[0];
            for (int k = 1; k < a.spb; ++k) sum += q[k * 32];
            *reinterpret_cast<dwq_f2*>(a.part + ((size_t)(img0 + (tid >> 5)) * (a.nblk * a.nq) + part_row0 + qi) * a.ld_part + cfirst) = sum;
        }
    };
    const int ng = (n + HALO + K - 1) / K;
    int buf = 0, qnext = a.Q, qi = 0;   // SE: the group whose row K - 2 finishes output row qnext - 1, the last of quantum qi (Q is a multiple of K)
    issue(0, 0);
    // one group: this wave's share of it has landed when all but the stores issued after its loads have completed (memory
    // operations complete in order; group 0 stores one row, every later group K rows)
    auto step = [&](int g, auto GG) {
        constexpr bool guard = decltype(GG)::value != 0;
        if (g == 0) __builtin_amdgcn_s_waitcnt(0x0f70);
        else if (g == 1) __builtin_amdgcn_s_waitcnt(4 | 0x0f70);
        else __builtin_amdgcn_s_waitcnt(((4 * K) & 15) | (((4 * K) >> 4) << 14) | 0x0f70);
        __syncthreads();                      // everybody's share has; everybody is done with the other buffer
        if (g + 1 < ng) issue(g + 1, buf ^ 1);
        const int t0 = g * K;
        dwq_group<T, K, ACT, SE, guard>(lslot + buf * a.gwords, rpitch, acc, w, sc, sh, a.act, dst, orow, opitch, ooff, t0, n, psum, psum2);
        if constexpr (SE) {
            if (t0 == qnext) {   // (t0 <= n here: a quantum that ends with the segment ends in the group of the segment's last row)
                flush(qi, qi & 1);
                ++qi; qnext += a.Q;
                psum = psum2;
            } else {
                psum += psum2;
            }
            psum2 = (dwq_f2){0.f, 0.f};
        }
        buf ^= 1;
    };
    // the first group and those that reach the segment's last K - 1 output rows run the guarded copy
    int g = 0;
    step(g++, dwq_int<1>{});
    for (; g * K + K <= n; ++g) step(g, dwq_int<0>{});
    for (; g < ng; ++g) step(g, dwq_int<1>{});
    if constexpr (SE) {
        if (qi * a.Q < n) flush(qi, qi & 1);   // the image's last quantum when it is a short one
    }
}

// Geometry.  Column blocks of at most 8 strips (4 waves), two images per workgroup where that keeps more waves resident; row quanta by the
// map's height alone (compiler.dwl_geometry mirrors nblk x nq: the rows of the squeeze-excite sums); the segmentation of a launch
// by instruction counts: generations of workgroups x (rows of a segment x the cost of a row + what a segment costs).
static void dwq_quanta(int H, int K, int* Q, int* nq) {   // quanta of a multiple of K rows: they end behind row K - 2 of a group
    int q = H / 10;
    if (q < 1) q = 1;
    if (q > 8) q = 8;
    *Q = yr_round_up((H + q - 1) / q, K);
    *nq = (H + *Q - 1) / *Q;
}

static bool dwq_geometry(int K, bool se, DwqArgs* a, size_t* lds) {
    const int halo = K - 1;
    const int strips = (a->W + 3) / 4;
    const int wave_cap = K == 5 ? 12 : 16;   // resident waves per CU at the kernel's register count (3 | 4 per SIMD)
    a->nblk = (strips + 7) / 8;
    a->spb = (strips + a->nblk - 1) / a->nblk;
    a->colsp = yr_round_up(a->spb * 4 + halo, 8);
    a->U = a->colsp / 8;
    int per_cu = 0;
    for (int ni = 1; ni <= 2; ++ni) {   // two images per workgroup when their strips fit the 8 slots and more waves stay resident that way
        if (ni * a->spb > 8 || ni > a->B) break;
        const int nw = (ni * a->spb + 1) / 2;
        if ((K * ni * a->U + nw - 1) / nw > 10) continue;   // DMA slots per wave and group (MAXS in the kernel)
        const size_t l = (size_t)K * ni * a->colsp * 32 * 8 + (se ? 2 * 256 * 8 : 0);
        int pc = (int)(160 * 1024 / l);
        if (pc > wave_cap / nw) pc = wave_cap / nw;
        if (ni == 1 || pc * nw >= per_cu * a->nw) { a->ni = ni; a->nw = nw; *lds = l; per_cu = pc; }
    }
    a->lni = a->ni == 2 ? 1 : 0;
    a->step_a = a->nw / a->U; a->step_b = a->nw % a->U;
    a->gwords = K * a->ni * a->colsp * 32;
    if (per_cu < 1) return false;
    dwq_quanta(a->H, K, &a->Q, &a->nq);
    a->ncc = (a->C8 + 7) / 8;
    a->nig = (a->B + a->ni - 1) / a->ni;
    static const int force = getenv("YR_DWQ_QPS") ? atoi(getenv("YR_DWQ_QPS")) : 0;
    const long long slots = 256ll * per_cu, units = (long long)a->ncc * a->nig * a->nblk;
    // time of a launch ~ (what a segment costs) x max(1, segments per resident slot): rows x the instructions of a row + the halo
    // rows' reads and conversions + the fixed part (taps, descriptors, the first group's round trip: about 3.5 rows' worth).
    // Fitted on tools/dwq_probe.py with YR_DWQ_QPS (26x26x480 @128: one segment 51 us, two 57; 40x40x816 @32: one 63, two 58).
    const int row_cost = K * K * 5 + 45, in_cost = 30, seg_cost = 600;
    double best = 0;
    for (int qps = 1; qps <= a->nq; ++qps) {
        if (force && qps != (force < a->nq ? force : a->nq)) continue;
        const int nseg = (a->nq + qps - 1) / qps;
        if ((nseg - 1) * qps >= a->nq) continue;
        const int n = qps * a->Q < a->H ? qps * a->Q : a->H;
        double gens = (double)(units * nseg) / (double)slots;
        if (gens < 1) gens = 1;
        const double cost = gens * ((double)n * row_cost + halo * in_cost + seg_cost);
        if (best == 0 || cost <= best) {   // (ties: the longer segments)
            best = cost;
            a->qps = qps; a->nseg = nseg;
        }
    }
    return best != 0;
}

template <class T, int K, bool SE>
static int launch_dwq_t(DwqArgs a, int expect_rows, hipStream_t s) {
    size_t lds = 0;
    YR_REQUIRE(dwq_geometry(K, SE, &a, &lds), "depthwise (walking form): the map's rows do not fit LDS");
    if (SE) YR_REQUIRE(a.nblk * a.nq == expect_rows, "depthwise (walking form): the SE partial-sum buffer must hold %d rows per image (has %d)", a.nblk * a.nq, expect_rows);
    YR_REQUIRE((long long)a.ni * a.H * a.W * (a.ld_in > a.ld_out ? a.ld_in : a.ld_out) * 2 < (1ll << 30), "depthwise (walking form): the images of a workgroup must be below 1 GB");
    const long long nb = (long long)a.ncc * a.nig * a.nblk * a.nseg;
    YR_REQUIRE(nb < (1ll << 31), "depthwise: grid too large");
    a.nblocks = (unsigned)nb;
    const int actv = a.act == YR_ACT_RELU6 ? 0 : (a.act == YR_ACT_SWISH ? 1 : 2);
    static char nm[3][48];   // spelled like the symbol (element type, K, activation variant, SE): profiles are joined on it
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "dwq_kernel<%s,%d,0,%d>", yr_dtype_name(yr_elem<T>::dtype), K, (int)SE) +
                              snprintf(nm[1], sizeof(nm[1]), "dwq_kernel<%s,%d,1,%d>", yr_dtype_name(yr_elem<T>::dtype), K, (int)SE) +
                              snprintf(nm[2], sizeof(nm[2]), "dwq_kernel<%s,%d,2,%d>", yr_dtype_name(yr_elem<T>::dtype), K, (int)SE);
    (void)nm_len;
    yr_note_kernel(nm[actv]);
    const dim3 block(a.nw * 64);   // (at most 256)
    if (a.act == YR_ACT_RELU6) hipLaunchKernelGGL((dwq_kernel<T, K, 0, SE>), dim3(a.nblocks), block, lds, s, a);
    else if (a.act == YR_ACT_SWISH) hipLaunchKernelGGL((dwq_kernel<T, K, 1, SE>), dim3(a.nblocks), block, lds, s, a);
    else hipLaunchKernelGGL((dwq_kernel<T, K, 2, SE>), dim3(a.nblocks), block, lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T>
static int launch_dwq_k(const DwqArgs& a, int k, int part_rows, hipStream_t s) {
    (void)k;   // (5: the 3 x 3 instances were measured against the tile walk - 52 x 52 x 128 @128: 35 us against 32, both at the HBM rate - and are not built)
    return a.part ? launch_dwq_t<T, 5, true>(a, part_rows, s) : launch_dwq_t<T, 5, false>(a, 0, s);
}

int yr_launch_depthwise_walk(int dtype, int k, const void* in, const float* w, const float* scale, const float* shift, void* out, int B, int H, int W,
                             int C8, int ld_in, int ld_w, int ld_out, int pad_t, int pad_l, int act, float* part, int ld_part, int part_rows,
                             hipStream_t s) {
    if (k != 5) { yr_set_error("depthwise (walking form): 5 x 5 only"); return YR_ERR_ARG; }
    DwqArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.out = out;
    a.B = B; a.H = H; a.W = W; a.C8 = C8;
    a.ld_in = ld_in; a.ld_w = ld_w; a.ld_out = ld_out;
    a.pad_t = pad_t; a.pad_l = pad_l; a.act = act;
    a.part = part; a.ld_part = ld_part;
    if (dtype == YR_BF16) return launch_dwq_k<yr_bf16>(a, k, part_rows, s);
    if (dtype == YR_F16) return launch_dwq_k<yr_f16>(a, k, part_rows, s);
    yr_set_error("depthwise (walking form): 16-bit maps only");
    return YR_ERR_ARG;
}
