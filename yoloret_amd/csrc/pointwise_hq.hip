// All-couts k-streaming form of the 16-bit pointwise convolution (pointwise_h.hip's op, same fusions, same MFMA sequence per
// output: bit-identical results) for LONG k spaces and at most 256 couts per workgroup - the PROJECTIONS of the unfused
// EfficientNet stages ([13k..87k pixels] x [480..1392] -> [112..232]; reference code/yolo3/efficientnet.py:517-533) and the
// wide convs of the detection heads (code/yolo3/model.py:98-114,243-251,298-318).
//
// Why: in the LDS-tiled kernel (pwhl_kernel) BOTH operands of a 32-deep chunk pass through LDS behind one barrier per chunk
// with one chunk in flight: a projection's workgroup lived for 21-44 barrier-synchronous round trips of ~100 cycles of MFMA
// work each (2.0-2.7 TB/s), and its activation tile was fetched once per 64-cout tile.
// Here the two operands take different roads.  The ACTIVATIONS - the HBM stream, each element needed by exactly one wave -
// never touch LDS: a wave loads the operand fragments of its 16 PT pixels straight from global memory into a ring of D
// chunks in registers (D chunks = the wave's bytes in flight, independent of any barrier), once, because the workgroup
// accumulates ALL couts of its range (2 CP tiles of 16: up to 256).  The WEIGHTS - small, L2-resident, shared by the four
// waves - are fetched cooperatively in full 128-byte lines, 64 k per step, parked in LDS in fragment order (two buffers,
// one barrier per 64 k, the next step's block in flight during this step's MFMAs).  Every load of the loop is unconditional
// (clamped addresses, masks applied at use), so all waits are counted.
#include <stdlib.h>

#include "pwh_common.h"

// PT: 16-pixel tiles per wave (4 waves along the pixels: BM = 64 PT); CP: cout tile pairs per workgroup (BN = 32 CP).
template <class T, int PT, int CP, int MODE>
__global__ __launch_bounds__(256, 2) void pwhq_kernel(PwArgs a, int nsplit, unsigned out_bytes) {
    constexpr int BM = 64 * PT, CT = 2 * CP;
    constexpr int D = MODE == 2 ? 2 : (PT == 1 ? 6 : 4);   // ring depth in 32-deep chunks (even: a 64-deep step = two slots)
    constexpr int SB = D / 2;                               // steps per unrolled body
    __shared__ pwh_u4 wf[2][2][CT][64];                     // [buffer][chunk of the step][cout tile][fragment lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const unsigned L = yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (int)(L / (unsigned)nsplit) * BM + wave * 16 * PT;
    const int n0 = (int)(L % (unsigned)nsplit) * (32 * CP);
    const int kp = a.S.kp;
    const int nst = (kp + 63) >> 6;                         // 64-deep steps
    const int kl = g * 8;
    const pwh_rsrc orsrc = pwh_out_rsrc(a, out_bytes);

    PwhRow<MODE, T> row[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) row[p].init(a, m0 + p * 16 + li);

    // weight loader (as in pointwise_hs.hip): a wave instruction covers 8 rows x 128 bytes; wave w takes tile w & 1 and
    // MFMA rows 8 (w >> 1) .. + 7 of every pair; conflict-free ds_write_b128 groups, linear fragment reads.
    const int lkk = lane >> 3, lrs = lane & 7;
    const int lt = wave & 1, lrow = (wave >> 1) * 8 + lrs;
    const int lcout = 8 * (lrow >> 2) + 4 * lt + (lrow & 3);
    const int lslot = (lkk >> 2) * (CT * 64) + lt * 64 + (lkk & 3) * 16 + lrow;   // + 128 per pair
    const int lk = lkk * 8;
    const T* wrow[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const int n = n0 + c * 32 + lcout;
        wrow[c] = reinterpret_cast<const T*>(a.wt) + (size_t)(n < a.N ? n : a.N - 1) * kp;   // rows beyond N feed couts that are never stored
    }
    auto wfetch = [&](int step, pwh_u4 (&R)[CP]) __attribute__((always_inline)) {
        const int kraw = step * 64 + lk;
        const int k = kraw < kp ? kraw : kp - 8;   // the k tail of the weights meets zeroed activations
#pragma unroll
        for (int c = 0; c < CP; ++c) R[c] = *reinterpret_cast<const pwh_u4*>(wrow[c] + k);
    };
    auto wpark = [&](int buf, const pwh_u4 (&R)[CP]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CP; ++c) (&wf[buf][0][0][0])[lslot + 128 * c] = R[c];
    };

    struct Slot {
        pwh_u4 x[PT];
        float4 g0[MODE == 2 ? PT : 1], g1[MODE == 2 ? PT : 1];
        int cv[PT];
    };
    auto xfetch = [&](int chunk, Slot& S) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PT; ++p)
            row[p].template issue<false>(a, chunk * 32 + kl, kp, S.x[p], S.g0[MODE == 2 ? p : 0], S.g1[MODE == 2 ? p : 0], S.cv[p]);
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[t][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (uniform) whole octets and whole 64-deep steps of one identity source need no masking - unless the body runs past nst
    const int nbody = (nst + SB - 1) / SB;
    const bool need_mask = MODE != 1 || (a.S.s[0].c & 7) != 0 || (kp & 63) != 0 || nbody * SB != nst;

    Slot S[D];
    pwh_u4 R[CP];
    wfetch(0, R);
#pragma unroll
    for (int d = 0; d < D; ++d) xfetch(d, S[d]);
    wpark(0, R);
    __syncthreads();
    // Steps beyond nst (the body is SB steps long) run on clamped addresses with every activation masked to zero: they add
    // +0 to the accumulators and keep the loop free of branches around memory instructions.
    for (int b = 0; b < nbody; ++b) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int step = b * SB + u;
            const int buf = step & 1;
            wfetch(step + 1, R);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Slot& Sx = S[2 * u + h];
                pwh_u4 xf[PT];
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    xf[p] = need_mask ? pwh_finish<MODE, T>(Sx.x[p], Sx.g0[MODE == 2 ? p : 0], Sx.g1[MODE == 2 ? p : 0], Sx.cv[p]) : Sx.x[p];
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    const pwh_u4 w = wf[buf][h][t][lane];
#pragma unroll
                    for (int p = 0; p < PT; ++p) acc[t][p] = pwh_mfma<T>(w, xf[p], acc[t][p]);
                }
                xfetch(2 * step + h + D, Sx);
            }
            wpark(buf ^ 1, R);
            __syncthreads();
        }
    }

    // ---- epilogue: lane group g owns couts 8g..8g+7 of every pair (tiles 2c and 2c+1) for its PT pixels
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const int n = n0 + c * 32 + g * 8;
        float sc[8], sh[8];
        pwh_load_bn(a, n, sc, sh);
#pragma unroll
        for (int p = 0; p < PT; ++p) pwh_finish_oct_b<T>(a, orsrc, acc[2 * c][p], acc[2 * c + 1][p], sc, sh, m0 + p * 16 + li, n, li);
    }
}

template <class T, int PT, int CP>
static int launch_q(const PwArgs& a, int mode, hipStream_t s) {
    constexpr int BM = 64 * PT;
    const int ntm = (a.M + BM - 1) / BM, nsplit = (a.N + 32 * CP - 1) / (32 * CP);
    static char nm[3][48];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "pwhq_kernel<%s,%d,%d,0>", yr_dtype_name(yr_elem<T>::dtype), PT, CP) +
                              snprintf(nm[1], sizeof(nm[1]), "pwhq_kernel<%s,%d,%d,1>", yr_dtype_name(yr_elem<T>::dtype), PT, CP) +
                              snprintf(nm[2], sizeof(nm[2]), "pwhq_kernel<%s,%d,%d,2>", yr_dtype_name(yr_elem<T>::dtype), PT, CP);
    (void)nm_len;
    yr_note_kernel(nm[mode]);
    const dim3 grid((unsigned)ntm * (unsigned)nsplit);
    const unsigned ob = pwh_out_bytes(a);
    if (mode == 1) hipLaunchKernelGGL((pwhq_kernel<T, PT, CP, 1>), grid, dim3(256), 0, s, a, nsplit, ob);
    else if (mode == 2) hipLaunchKernelGGL((pwhq_kernel<T, PT, CP, 2>), grid, dim3(256), 0, s, a, nsplit, ob);
    else hipLaunchKernelGGL((pwhq_kernel<T, PT, CP, 0>), grid, dim3(256), 0, s, a, nsplit, ob);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// pairs per workgroup: all of N when that is at most `cap` pairs, else equal parts of at most `cap`
template <class T, int PT>
static int launch_q_pt(const PwArgs& a, int mode, int cap, hipStream_t s) {
    const int npairs = (a.N + 31) / 32;
    const int parts = (npairs + cap - 1) / cap;
    const int per = (npairs + parts - 1) / parts;
    if (per <= 2) return launch_q<T, PT, 2>(a, mode, s);
    if (per <= 3) return launch_q<T, PT, 3>(a, mode, s);
    if (per <= 4) return launch_q<T, PT, 4>(a, mode, s);
    if (per <= 5) return launch_q<T, PT, 5>(a, mode, s);
    if constexpr (PT == 1) {   // (two pixel tiles x 12 or 16 cout tiles of accumulators do not fit the register file: cap = 5 there)
        if (per <= 6) return launch_q<T, PT, 6>(a, mode, s);
        return launch_q<T, PT, 8>(a, mode, s);
    }
    return YR_NOT_TAKEN;
}

// variant: 0 = one pixel tile per wave, up to 256 couts per workgroup; 1 = two pixel tiles, up to 160 couts; 2 / 3 = the same
// with up to 128 / 96 couts (more workgroups on the small maps, the activations read once per cout range - from L2).
// Returns YR_NOT_TAKEN (no launch, no error text) when the form does not take the op: the caller falls back; real errors propagate.
int yr_pwhq_launch(int dtype, int variant, const PwArgs& a, hipStream_t s) {
    const int mode = (a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY) ? (a.gate ? 2 : 1) : 0;
    if (a.dw_w != nullptr || (mode == 0 && a.gate) || !pwh_out_fits_rsrc(a)) return YR_NOT_TAKEN;
    for (int i = 0; i < YR_MAX_SRC; ++i)
        if (a.S.s[i].xform == YR_X_MAXPOOL2 || a.S.s[i].xform == YR_X_MAXPOOL4) return YR_NOT_TAKEN;
    const int cap = (variant & 1) ? (variant >= 2 ? 3 : 5) : (variant >= 2 ? 4 : 8);
    if (dtype == YR_BF16) return (variant & 1) ? launch_q_pt<yr_bf16, 2>(a, mode, cap, s) : launch_q_pt<yr_bf16, 1>(a, mode, cap, s);
    if (dtype == YR_F16) return (variant & 1) ? launch_q_pt<yr_f16, 2>(a, mode, cap, s) : launch_q_pt<yr_f16, 1>(a, mode, cap, s);
    return YR_NOT_TAKEN;
}
