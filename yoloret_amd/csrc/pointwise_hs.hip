// Activation-stationary form of the 16-bit pointwise convolution (pointwise_h.hip's op, same fusions, same MFMA sequence
// per output: bit-identical results) for k spaces of at most 256 channels - the EXPANSIONS of the unfused EfficientNet
// stages ([13k..87k pixels] x [80..232] -> [480..1392]; reference code/yolo3/efficientnet.py:485-496) and most convs of
// the detection heads (code/yolo3/model.py:98-114,243-251,298-318).
//
// Why: in the LDS-tiled kernel (pwhl_kernel) a workgroup lives for one 128 x 64 output tile.  An expansion has 10-22
// cout tiles per pixel tile, so the activation tile was fetched 13x (PMC: 408 MB requested from L2 for 97 MB of operands
// + output), every workgroup paid ceil(K / 32) barrier-synchronous round trips for ~100 cycles of MFMA work each, and the
// stores - 85 % of the bytes - started only after the last of them: 1.0-2.3 TB/s.
// Here a wave fetches the operand fragments of ITS 16 PT pixels over the whole k extent ONCE (one burst, NCH PT 16-byte
// loads per lane, kept in registers) and the workgroup walks the cout tile PAIRS of its range: the 32 x kp weight block of
// a pair is fetched cooperatively in full 128-byte lines (8 rows x 128 B per wave instruction), parked in LDS in
// fragment order (two buffers, one barrier per PAIR, the next pair's block in flight during this pair's MFMAs and
// stores), multiplied against the resident activations, finished (BN, activation, ...) and stored.  Every memory
// instruction of the loop is unconditional (clamped loads; stores through a buffer descriptor, dead lanes out of range),
// so the wait in front of the LDS park is COUNTED and does not wait for the stores just issued.
#include <stdlib.h>

#include "pwh_common.h"

#ifndef PWHS_ABL
#define PWHS_ABL 0   // tools/pwx_bench.hip builds ablations: 1 no stores, 2 no weight refetch / park / barrier, 4 no MFMA
#endif
constexpr int PWHS_MAX_PAIRS = 24;   // cout tile pairs one workgroup may walk (its BN rows live in LDS)

typedef float pwh_f2 __attribute__((ext_vector_type(2)));
template <class T> using pwh_t2 = T __attribute__((ext_vector_type(2)));

// PT: 16-pixel tiles per wave (4 waves along the pixels: BM = 64 PT); NCH: 32-deep chunks held (kp <= 32 NCH).
// PLAIN: no pre-BN addend, no residual, no pooled output, activation none or ReLU6 (every expansion): the epilogue is 4 packed
// FMAs, 16 min/max, 4 conversions and a store per accumulator octet - the generic one (pwh_finish_oct_b) spends ~4x that
// on paths the op does not take, and these loops are bound by instruction issue, not by a pipe.
// CLAMP (PLAIN only): the activation is ReLU6 (else none).
template <class T, int PT, int NCH, int MODE, bool PLAIN, bool CLAMP>
__global__ __launch_bounds__(256, PT == 1 ? 3 : 2) void pwhs_kernel(PwArgs a, int pairs_per_wg, int nsplit, unsigned out_bytes) {
    constexpr int BM = 64 * PT;
    constexpr int NST = (NCH + 1) / 2;             // 64-deep load steps of a pair's weight block
    constexpr int NBN = PWHS_MAX_PAIRS * 32 / 256;   // BN elements per thread at most
    __shared__ pwh_u4 wf[2][2 * NST][2][64];       // [buffer][chunk][tile of the pair][fragment lane]
    __shared__ __attribute__((aligned(16))) float bnl[2][PWHS_MAX_PAIRS * 32];   // BN scale | shift of the workgroup's cout range
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const unsigned L = yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (int)(L / (unsigned)nsplit) * BM + wave * 16 * PT;
    const int npairs = (a.N + 31) >> 5;
    const int pb = (int)(L % (unsigned)nsplit) * pairs_per_wg;
    const int pe = pb + pairs_per_wg < npairs ? pb + pairs_per_wg : npairs;
    const int kp = a.S.kp;
    const pwh_rsrc orsrc = pwh_out_rsrc(a, out_bytes);

    // ---- weight loader: a wave instruction covers 8 rows x 128 bytes (two chunks); wave w takes tile w & 1, MFMA rows
    // 8 (w >> 1) .. + 7 of every pair.  LDS slot = fragment lane 16 g + i of [chunk][tile]: the 8 lanes of a ds_write_b128
    // group write 8 consecutive slots - conflict-free - and every fragment read is linear.
    const int lkk = lane >> 3, lrs = lane & 7;
    const int lt = wave & 1, lrow = (wave >> 1) * 8 + lrs;                   // tile of the pair, MFMA row
    const int lcout = 8 * (lrow >> 2) + 4 * lt + (lrow & 3);               // cout within the pair (see pointwise_h.hip)
    const int lslot = (lkk >> 2) * 128 + lt * 64 + (lkk & 3) * 16 + lrow;  // + 256 per 64-deep step
    // through a buffer descriptor over the [N][kp] weights: one add per load and turn (these loops are bound by instruction
    // issue; the scalar-offset operand is not range-checked, so it cannot carry the pair) and rows beyond N read as zeros
    // (their couts are never stored)
    const pwh_rsrc wrsrc = pwh_make_rsrc(a.wt, (unsigned)a.N * (unsigned)kp * 2u);
    unsigned wvoff[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int k = st * 64 + lkk * 8;
        wvoff[st] = ((unsigned)lcout * (unsigned)kp + (unsigned)(k < kp ? k : kp - 8)) * 2u;   // the k tail meets zeroed activations
    }
    auto fetch = [&](int pair, pwh_u4 (&R)[NST]) __attribute__((always_inline)) {
        const unsigned soff = (unsigned)pair * 32u * (unsigned)kp * 2u;
#pragma unroll
        for (int st = 0; st < NST; ++st) R[st] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff[st] + soff, 0, 0);
    };
    auto park = [&](int buf, const pwh_u4 (&R)[NST]) __attribute__((always_inline)) {
#pragma unroll
        for (int st = 0; st < NST; ++st) (&wf[buf][0][0][0])[lslot + 256 * st] = R[st];
    };

    // ---- prologue: ONE round trip - the activation operand of this wave (all of k), the BatchNorm rows of the cout range
    // and the first weight block are all issued before anything waits
    pwh_u4 x[NCH][PT];
    const int kl = g * 8;
    const bool need_mask = MODE != 1 || (a.S.s[0].c & 7) != 0 || (kp & 31) != 0 || kp < 32 * NCH;
    float4 g0[MODE == 2 ? NCH : 1][MODE == 2 ? PT : 1], g1[MODE == 2 ? NCH : 1][MODE == 2 ? PT : 1];
    int cv[NCH][PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        PwhRow<MODE, T> row;
        row.init(a, m0 + p * 16 + li);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
            row.template issue<false>(a, ch * 32 + kl, kp, x[ch][p], g0[MODE == 2 ? ch : 0][MODE == 2 ? p : 0],
                                      g1[MODE == 2 ? ch : 0][MODE == 2 ? p : 0], cv[ch][p]);
    }
    float bsc[NBN], bsh[NBN];
#pragma unroll
    for (int e = 0; e < NBN; ++e) {   // unconditional loads of clamped addresses (a dependent load -> LDS store loop is one round trip per turn)
        const int n = pb * 32 + e * 256 + tid, nc = n < a.N ? n : a.N - 1;
        bsc[e] = 1.f;
        bsh[e] = 0.f;
        if (e * 256 < (pe - pb) * 32) {   // uniform
            if (a.scale) bsc[e] = a.scale[nc];
            if (a.shift) bsh[e] = a.shift[nc];
        }
    }
    pwh_u4 R[NST];
    fetch(pb, R);
#pragma unroll
    for (int e = 0; e < NBN; ++e) {
        bnl[0][e * 256 + tid] = bsc[e];
        bnl[1][e * 256 + tid] = bsh[e];
    }
    park(0, R);
    if (need_mask) {
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                x[ch][p] = pwh_finish<MODE, T>(x[ch][p], g0[MODE == 2 ? ch : 0][MODE == 2 ? p : 0],
                                               g1[MODE == 2 ? ch : 0][MODE == 2 ? p : 0], cv[ch][p]);
    }
    // PLAIN: byte offset of each pixel row of this lane + this lane group's 16 bytes of a pair's 64; a turn adds the pair's
    // 64 j.  Dead lanes (rows beyond M; in the LAST pair of N, octets beyond N) sit at 0x7f000000, beyond the descriptor's
    // range (the launcher checks), and a dead turn adds 2 GB: out of range for every lane, no 32-bit wrap either way.
    unsigned rowoff[PT], rowoff_last[PT];
    const bool tail_dead = (npairs - 1) * 32 + g * 8 >= a.N;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = m0 + p * 16 + li;
        rowoff[p] = m < a.M ? (unsigned)m * (unsigned)a.out_ld * 2u + (unsigned)g * 16u : 0x7f000000u;
        rowoff_last[p] = tail_dead ? 0x7f000000u : rowoff[p];
    }
    // The epilogue of pair j - 1 runs BESIDE the MFMAs of pair j (independent registers: the matrix pipe works while the
    // wave issues the BN / clamp / convert / store instructions); left to follow its own MFMAs it waited for their results
    // and the per-wave chain ds_read -> MFMA -> VALU -> store -> park -> barrier had nothing to overlap with at 2-3 waves
    // per SIMD.  Turn pb's "previous pair" is dead (offset beyond the range), one more epilogue follows the loop.
    auto finish = [&](const f32x4 (&lo)[PT], const f32x4 (&hi)[PT], int j, bool live) __attribute__((always_inline)) {
        const int bi = (j - pb) * 32 + g * 8;
        const float4 s0 = *reinterpret_cast<const float4*>(&bnl[0][bi]), s1 = *reinterpret_cast<const float4*>(&bnl[0][bi + 4]);
        const float4 h0 = *reinterpret_cast<const float4*>(&bnl[1][bi]), h1 = *reinterpret_cast<const float4*>(&bnl[1][bi + 4]);
        const int n = j * 32 + g * 8;
        if constexpr (PLAIN) {
            const pwh_f2 sc2[4] = {{s0.x, s0.y}, {s0.z, s0.w}, {s1.x, s1.y}, {s1.z, s1.w}};
            const pwh_f2 sh2[4] = {{h0.x, h0.y}, {h0.z, h0.w}, {h1.x, h1.y}, {h1.z, h1.w}};
            const bool last = j == npairs - 1;                       // (uniform)
            const unsigned joff = (live && !(PWHS_ABL & 1)) ? (unsigned)j * 64u : 0x80000000u;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                pwh_f2 r[4] = {{lo[p][0], lo[p][1]}, {lo[p][2], lo[p][3]}, {hi[p][0], hi[p][1]}, {hi[p][2], hi[p][3]}};
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = __builtin_elementwise_fma(r[q], sc2[q], sh2[q]);
                pwh_u4 v;   // pair by pair: one v_cvt_pk per dword (an 8-wide convert feeding the packed clamp was split into 8 + 4 perms)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r[q], pwh_t2<T>));
                if constexpr (CLAMP) v = pwh_relu6_packed<T>(v);
                __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, (last ? rowoff_last[p] : rowoff[p]) + joff, 0, 0);
            }
        } else {
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int p = 0; p < PT; ++p)
                pwh_finish_oct_b<T>(a, orsrc, lo[p], hi[p], sc, sh, (!live || (PWHS_ABL & 1)) ? 0x7fffffff : m0 + p * 16 + li, n, li);
        }
    };
    // one turn: the MFMAs of pair j into (clo_, chi_) beside the epilogue of pair j - 1 out of (plo_, phi_); turns come in
    // twos with the two accumulator sets swapping roles (no register copies that would wait for the MFMAs); a turn beyond
    // pe (odd counts) multiplies a stale buffer into a dead set and stores nothing
    auto turn = [&](int j, f32x4 (&clo_)[PT], f32x4 (&chi_)[PT], const f32x4 (&plo_)[PT], const f32x4 (&phi_)[PT]) __attribute__((always_inline)) {
        const int buf = (j - pb) & 1;
        if (!(PWHS_ABL & 2)) fetch(j + 1 < pe ? j + 1 : pe - 1, R);   // (clamped, not skipped: no branch around a memory instruction)
#pragma unroll
        for (int p = 0; p < PT; ++p) clo_[p] = chi_[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const pwh_u4 w0 = wf[buf][ch][0][lane], w1 = wf[buf][ch][1][lane];
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                if (PWHS_ABL & 4) { clo_[p][0] += __builtin_bit_cast(float, w0[0] ^ x[ch][p][0]); chi_[p][0] += __builtin_bit_cast(float, w1[1] ^ x[ch][p][1]); continue; }
                clo_[p] = pwh_mfma<T>(w0, x[ch][p], clo_[p]);
                chi_[p] = pwh_mfma<T>(w1, x[ch][p], chi_[p]);
            }
        }
        const bool live = j > pb && j <= pe;
        finish(plo_, phi_, live ? j - 1 : pb, live);
        if (!(PWHS_ABL & 2)) {
            park(buf ^ 1, R);
            __syncthreads();
        }
    };
    f32x4 alo[PT], ahi[PT], blo[PT], bhi[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) blo[p] = bhi[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int j = pb; j < pe; j += 2) {
        turn(j, alo, ahi, blo, bhi);
        turn(j + 1, blo, bhi, alo, ahi);
    }
    finish(blo, bhi, pe - 1, ((pe - pb) & 1) == 0);   // (odd counts: the dead turn finished the last pair)
}

template <class T, int PT, int NCH>
static int launch_s(const PwArgs& a, int mode, int target_wgs, hipStream_t s) {
    constexpr int BM = 64 * PT;
    const int ntm = (a.M + BM - 1) / BM, npairs = (a.N + 31) / 32;
    // cout ranges: as few as keep `target_wgs` workgroups in flight (each range re-reads the pixel tile - from L2 - and
    // pays its own prologue), at least two pairs per workgroup
    int nsplit = (target_wgs + ntm - 1) / ntm;
    if (nsplit > (npairs + 1) / 2) nsplit = (npairs + 1) / 2;
    if (nsplit < 1) nsplit = 1;
    if ((npairs + nsplit - 1) / nsplit > PWHS_MAX_PAIRS) nsplit = (npairs + PWHS_MAX_PAIRS - 1) / PWHS_MAX_PAIRS;
    const int per = (npairs + nsplit - 1) / nsplit;
    nsplit = (npairs + per - 1) / per;
    const bool plain = !a.pre && !a.res && !a.pool && (a.act == YR_ACT_NONE || a.act == YR_ACT_RELU6);
    const int form = !plain ? 0 : (a.act == YR_ACT_RELU6 ? 2 : 1);   // generic | plain | plain + ReLU6: the last two template arguments
    static char nm[3][3][56];
    static bool named = false;
    if (!named) {
        for (int m = 0; m < 3; ++m)
            for (int f = 0; f < 3; ++f)
                snprintf(nm[m][f], sizeof(nm[m][f]), "pwhs_kernel<%s,%d,%d,%d,%d,%d>", yr_dtype_name(yr_elem<T>::dtype), PT, NCH, m, f > 0, f > 1);
        named = true;
    }
    yr_note_kernel(nm[mode][form]);
    const dim3 grid((unsigned)ntm * (unsigned)nsplit);
    const unsigned ob = pwh_out_bytes(a);
#define PWHS_GO(MODE, PLAIN, CLAMP) hipLaunchKernelGGL((pwhs_kernel<T, PT, NCH, MODE, PLAIN, CLAMP>), grid, dim3(256), 0, s, a, per, nsplit, ob)
    if (plain && a.act == YR_ACT_RELU6) {
        if (mode == 1) PWHS_GO(1, true, true); else if (mode == 2) PWHS_GO(2, true, true); else PWHS_GO(0, true, true);
    } else if (plain) {
        if (mode == 1) PWHS_GO(1, true, false); else if (mode == 2) PWHS_GO(2, true, false); else PWHS_GO(0, true, false);
    } else {
        if (mode == 1) PWHS_GO(1, false, false); else if (mode == 2) PWHS_GO(2, false, false); else PWHS_GO(0, false, false);
    }
#undef PWHS_GO
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T, int PT>
static int launch_s_pt(const PwArgs& a, int mode, int target_wgs, hipStream_t s) {
    const int nch = (a.S.kp + 31) >> 5;
    if (nch <= 2) return launch_s<T, PT, 2>(a, mode, target_wgs, s);
    if (nch <= 3) return launch_s<T, PT, 3>(a, mode, target_wgs, s);
    if (nch <= 4) return launch_s<T, PT, 4>(a, mode, target_wgs, s);
    if (nch <= 5) return launch_s<T, PT, 5>(a, mode, target_wgs, s);
    if (nch <= 6) return launch_s<T, PT, 6>(a, mode, target_wgs, s);
    return launch_s<T, PT, 8>(a, mode, target_wgs, s);
}

// variant: 0 / 1 = one / two pixel tiles per wave with ~512 workgroups wanted, 2 / 3 = the same with ~1280.
// Returns YR_NOT_TAKEN (pw_common.h: positive, distinct from every yr_status error; no launch, no error text) when the form
// does not take the op: the caller falls back.  Real argument errors keep their negative codes and are propagated.
int yr_pwhs_launch(int dtype, int variant, const PwArgs& a, hipStream_t s) {
    const int mode = (a.S.n == 1 && a.S.s[0].xform == YR_X_IDENTITY) ? (a.gate ? 2 : 1) : 0;
    if (a.S.kp > 256 || a.dw_w != nullptr || (mode == 0 && a.gate) || !pwh_out_fits_rsrc(a)) return YR_NOT_TAKEN;
    for (int i = 0; i < YR_MAX_SRC; ++i)
        if (a.S.s[i].xform == YR_X_MAXPOOL2 || a.S.s[i].xform == YR_X_MAXPOOL4) return YR_NOT_TAKEN;
    const int target = variant >= 2 ? 1280 : 512;
    if (dtype == YR_BF16) return (variant & 1) ? launch_s_pt<yr_bf16, 2>(a, mode, target, s) : launch_s_pt<yr_bf16, 1>(a, mode, target, s);
    if (dtype == YR_F16) return (variant & 1) ? launch_s_pt<yr_f16, 2>(a, mode, target, s) : launch_s_pt<yr_f16, 1>(a, mode, target, s);
    return YR_NOT_TAKEN;
}
