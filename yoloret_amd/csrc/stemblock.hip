// Fused network entry: stem Conv2D 3x3 s2 (Cin=3) + BN + act  ->  depthwise 3x3 s1 + BN + act  ->
// project 1x1 + BN, one kernel.  For MobileNetV2 this is Conv1 + the "expanded_conv" block (block 0,
// which has no expand stage) [3P; reference code/yolo3/override.py:339]: the 208x208x24 stem output
// and depthwise output (2 x 4.15 MB per image at 416, written and re-read by the unfused chain) never
// leave the CU - HBM sees the 2.08 MB image in and the 2.77 MB block output out.
//
// One workgroup (4 waves) = a 14 x 14 tile of block-output pixels, whose 16 x 16 stem-output halo tile is
// exactly one halo pixel per lane.  The kernel is VALU-bound (1248 MACs per output pixel against 19 bytes
// of HBM traffic), so it is laid out for the FMA pipe, not for bandwidth:
//   * lane = pixel, every lane computes ALL channels of its pixel.  Weights are therefore wave-uniform
//     and are read through the constant address space (s_load -> SGPR operands of the FMAs): no LDS or
//     VGPR traffic for weights at all;
//   * channels go in pairs held as float2 so every FMA issues as v_pk_fma_f32 (two MACs per lane per issue);
//   * the loops over channel pairs are rolled: one pair's weights are one contiguous s_load burst (host
//     layout [CP][...], <= 70 SGPRs live, no SGPR spills); everything inside is straight-line;
//   * each lane reads its own 3 x 3 x 3 input window straight from global memory (three runs of 9 contiguous
//     floats as dword-aligned 16-byte loads; neighbouring windows overlap, the re-reads hit L1).  Staging a
//     33 x 33 x 3 input tile in LDS instead cost 13 loads + 13 LDS stores + 27 LDS reads per lane and two more
//     barriers: 0.198 ms vs 0.166 ms;
//   * LDS holds only the stem-output halo tile Es[CP][256] (float2 per lane: conflict-free writes, and the
//     depthwise taps read lane-consecutive float2s).  The depthwise result goes straight from registers into
//     the projection.
#include "yr_common.h"

#define SB_T 14             // output tile edge
#define SB_E (SB_T + 2)     // stem-output halo tile edge (== 16: one halo pixel per lane of 256)
#define SB_WS 58            // stem floats per channel pair:      27 taps x 2 (times the BN scale) | 1 1 | shift 2
#define SB_WD 22            // depthwise floats per channel pair:  9 taps x 2 (times the BN scale) | 1 1 | shift 2

typedef const float __attribute__((address_space(4))) * kptr;  // uniform reads of this become s_load
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load at a dword-aligned address

// T: element type of the OUTPUT map (image, weights and arithmetic: float32; one rounding at the store).
template <class T>
struct SbArgs {
    const void* in;    // [B][Hi][Wi][3] float32 in [0,1] - or uint8 (IN8): the decoded image bytes as they are; the /255 of
                       // code/yolo.py:106 (tf.io.decode_image(dtype=float32)) is then applied to the 27-tap sums (in_scale)
    float in_scale;    // 1 (float32 images) | 1/255 (uint8 images)
    T* out;            // [B][Ho][Wo][ld_out]   (Ho = ceil(Hi/2))
    const float* ws;   // stem       [CP][SB_WS]
    const float* wd;   // depthwise  [CP][SB_WD]
    const float* wp;   // project    [2*CP][COP]
    const float* bp;   // project    scale [COP] ++ shift [COP]
    int Hi, Wi, Ho, Wo, Cout, ld_out, pad_t, pad_l, act, tiles_x, tiles_y;
    // COP == 0 (stem + depthwise only, the entry of the squeeze-excite EfficientNets): per-tile channel sums of the stored
    // depthwise map, float32 [B][tiles_y * tiles_x][ld_part] (the squeeze; SE_FC adds the rows up), or null
    float* part; int ld_part;
};

__device__ __forceinline__ v2f sb_fma(v2f x, v2f y, v2f z) { return __builtin_elementwise_fma(x, y, z); }

template <bool RELU6, class T = float>
__device__ __forceinline__ v2f sb_act(v2f v, int act) {
    if (RELU6) return (v2f){fminf(fmaxf(v.x, 0.f), 6.f), fminf(fmaxf(v.y, 0.f), 6.f)};
    return (v2f){yr_apply_act_t<T>(v.x, act), yr_apply_act_t<T>(v.y, act)};   // (16-bit maps: hardware exp2 / rcp swish)
}

template <int CP, int COP, bool RELU6, class T, bool IN8>
__global__ __launch_bounds__(256, 4) void stemblock_kernel(SbArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    v2f* Es = reinterpret_cast<v2f*>(lds);                       // [CP][256]
    const int tid = threadIdx.x;
    const int t = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int tpi = a.tiles_x * a.tiles_y;
    const int b = t / tpi, r = t - b * tpi;
    const int ty = r / a.tiles_x;
    const int oy0 = ty * SB_T, ox0 = (r - ty * a.tiles_x) * SB_T;
    const int iy0 = 2 * (oy0 - 1) - a.pad_t, ix0 = 2 * (ox0 - 1) - a.pad_l;  // input coords of halo pixel (0,0), tap (0,0)

    // ---- phase 1: lane = halo pixel; its 3 x 3 x 3 input window (27 floats: three runs of 9 contiguous floats)
    // comes straight from global memory - neighbouring lanes' windows overlap, so the re-reads hit L1 - as two
    // dword-aligned 16-byte loads + one dword per row; then the stem conv, all channels, one pair per (rolled)
    // iteration -> Es.  (An LDS-staged input tile cost 13 loads + 13 LDS stores + 27 LDS reads per lane and two
    // barriers: 0.198 -> see DESIGN.md.)
    {
        const int ey = tid >> 4, ex = tid & 15;
        const int sy = oy0 - 1 + ey, sx = ox0 - 1 + ex;
        const bool valid = sy >= 0 && sy < a.Ho && sx >= 0 && sx < a.Wo;  // outside: zero (the depthwise's SAME padding)
        float in[27];
        const int iy = iy0 + 2 * ey, ic = (ix0 + 2 * ex) * 3, rowlen = a.Wi * 3;
        const bool interior = iy >= 0 && iy + 2 < a.Hi && ic >= 0 && ic + 8 < rowlen;
        if constexpr (IN8) {
            // uint8 image: a window row is 9 BYTES at a byte-aligned address - one global_load_dwordx3 (unaligned dword access
            // is enabled for global memory), nine conversions; the float32 batch (4x the bytes) is never written or read
            typedef unsigned u3 __attribute__((ext_vector_type(3), aligned(1)));
            const unsigned char* img = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * a.Hi * a.Wi * 3;
            if (interior && ic + 11 < rowlen) {      // (the 12-byte load stays inside the row)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const u3 v = *reinterpret_cast<const u3*>(img + (size_t)(iy + ky) * rowlen + ic);
#pragma unroll
                    for (int j = 0; j < 9; ++j) in[ky * 9 + j] = (float)((v[j >> 2] >> (8 * (j & 3))) & 0xffu);
                }
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int y = iy + ky, c = ic + j;
                        in[ky * 9 + j] = (valid && y >= 0 && y < a.Hi && c >= 0 && c < rowlen) ? (float)img[(size_t)y * rowlen + c] : 0.f;
                    }
            }
        } else {
        const float* img = reinterpret_cast<const float*>(a.in) + (size_t)b * a.Hi * a.Wi * 3;
        if (interior) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* rp = img + (size_t)(iy + ky) * rowlen + ic;
                const f4u v0 = *reinterpret_cast<const f4u*>(rp), v1 = *reinterpret_cast<const f4u*>(rp + 4);
                in[ky * 9 + 0] = v0.x; in[ky * 9 + 1] = v0.y; in[ky * 9 + 2] = v0.z; in[ky * 9 + 3] = v0.w;
                in[ky * 9 + 4] = v1.x; in[ky * 9 + 5] = v1.y; in[ky * 9 + 6] = v1.z; in[ky * 9 + 7] = v1.w;
                in[ky * 9 + 8] = rp[8];
            }
        } else {  // image border (the stem's SAME zero padding) and halo pixels outside the image
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const int y = iy + ky, c = ic + j;
                    in[ky * 9 + j] = (valid && y >= 0 && y < a.Hi && c >= 0 && c < rowlen) ? img[(size_t)y * rowlen + c] : 0.f;
                }
        }
        }
        const kptr ws = (kptr)a.ws;
        const float hi = valid ? 6.f : 0.f;
#pragma unroll 1
        for (int p = 0; p < CP; ++p) {
            const kptr w = ws + p * SB_WS;
            // four chains (taps k = c mod 4): a dependent v_pk_fma_f32 issues only every ~13th slot - two chains per wave cap
            // at 89 TF of the pipe's 128 (tools/peak.hip)
            v2f acc = {0.f, 0.f}, acc2 = {0.f, 0.f}, acc3 = {0.f, 0.f}, acc4 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 24; k += 4) {
                acc = sb_fma((v2f){in[k], in[k]}, (v2f){w[2 * k], w[2 * k + 1]}, acc);
                acc2 = sb_fma((v2f){in[k + 1], in[k + 1]}, (v2f){w[2 * k + 2], w[2 * k + 3]}, acc2);
                acc3 = sb_fma((v2f){in[k + 2], in[k + 2]}, (v2f){w[2 * k + 4], w[2 * k + 5]}, acc3);
                acc4 = sb_fma((v2f){in[k + 3], in[k + 3]}, (v2f){w[2 * k + 6], w[2 * k + 7]}, acc4);
            }
            acc = sb_fma((v2f){in[24], in[24]}, (v2f){w[48], w[49]}, acc);
            acc2 = sb_fma((v2f){in[25], in[25]}, (v2f){w[50], w[51]}, acc2);
            acc3 = sb_fma((v2f){in[26], in[26]}, (v2f){w[52], w[53]}, acc3);
            acc = (acc + acc2) + (acc3 + acc4);
            // BN shift; the BN scale is folded into the taps by the host (w[54..55] hold 1).  uint8 images: the taps were
            // multiplied with 0..255, the sum takes the 1/255 here (float32 images: an exact fma with 1)
            if constexpr (IN8) acc = sb_fma(acc, (v2f){a.in_scale, a.in_scale}, (v2f){w[56], w[57]});
            else acc += (v2f){w[56], w[57]};
            if (RELU6) {                         // upper clamp 0 outside the map = the depthwise's zero padding, for free
                acc = (v2f){__builtin_amdgcn_fmed3f(acc.x, 0.f, hi), __builtin_amdgcn_fmed3f(acc.y, 0.f, hi)};
            } else {
                acc = sb_act<false, T>(acc, a.act);
                if (!valid) acc = (v2f){0.f, 0.f};
            }
            Es[p * 256 + tid] = acc;
        }
    }
    __syncthreads();

    if constexpr (COP == 0) {
        // ---- phase 2, no projection: the depthwise map itself is the output (2*CP channels per lane, contiguous), and the
        // tile's per-channel sums of the STORED values go to its row of `part` (wave butterfly, then the four waves in order)
        const int py = tid / SB_T, px = tid - py * SB_T;
        const int gy = oy0 + py, gx = ox0 + px;
        const bool live = tid < SB_T * SB_T && gy < a.Ho && gx < a.Wo;
        const kptr wd = (kptr)a.wd;
        const v2f* e0 = Es + (tid < SB_T * SB_T ? py * SB_E + px : 0);
        v2f dres[CP];
#pragma unroll
        for (int p = 0; p < CP; ++p) {
            const kptr w = wd + p * SB_WD;
            const v2f* e = e0 + p * 256;
            v2f dr[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int k = ky * 3 + kx;
                    dr[ky] = sb_fma(e[ky * SB_E + kx], (v2f){w[2 * k], w[2 * k + 1]}, dr[ky]);
                }
            v2f d = (dr[0] + dr[1]) + dr[2];
            d = sb_act<RELU6, T>(d + (v2f){w[20], w[21]}, a.act);
            // what is stored (one rounding), widened again: the squeeze is the mean of the stored map
            dres[p] = (v2f){(float)(T)d.x, (float)(T)d.y};
        }
        if (live) {
            T* op = a.out + (((size_t)b * a.Ho + gy) * a.Wo + gx) * a.ld_out;
#pragma unroll
            for (int p = 0; p < CP; p += 2) {   // 4 channels per store (CP is even)
                if (2 * p + 3 < a.Cout) {
                    yr_st4<T>(op + 2 * p, make_float4(dres[p].x, dres[p].y, dres[p + 1].x, dres[p + 1].y));
                } else {
                    if (2 * p < a.Cout) yr_st1<T>(op + 2 * p, dres[p].x);
                    if (2 * p + 1 < a.Cout) yr_st1<T>(op + 2 * p + 1, dres[p].y);
                    if (2 * p + 2 < a.Cout) yr_st1<T>(op + 2 * p + 2, dres[p + 1].x);
                }
            }
        }
        if (a.part != nullptr) {
            __syncthreads();                  // every reader of Es is done: its first floats become the wave sums
            float* ws4 = lds;                 // [4 waves][2 * CP]
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int p = 0; p < CP; ++p) {
                v2f v = live ? dres[p] : (v2f){0.f, 0.f};
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); }
                if (lane == 0) { ws4[wave * 2 * CP + 2 * p] = v.x; ws4[wave * 2 * CP + 2 * p + 1] = v.y; }
            }
            __syncthreads();
            if (tid < 2 * CP && tid < a.ld_part)
                a.part[((size_t)b * tpi + r) * a.ld_part + tid] = ((ws4[tid] + ws4[2 * CP + tid]) + ws4[4 * CP + tid]) + ws4[6 * CP + tid];
        }
        return;
    }
    // ---- phase 2: lane = output pixel; depthwise 3x3 + BN + act in registers, projected immediately
    if (tid < SB_T * SB_T) {
        const int py = tid / SB_T, px = tid - py * SB_T;
        const int gy = oy0 + py, gx = ox0 + px;
        v2f o[COP > 0 ? COP / 2 : 1];
#pragma unroll
        for (int n = 0; n < COP / 2; ++n) o[n] = (v2f){0.f, 0.f};
        const kptr wd = (kptr)a.wd;
        const kptr wp = (kptr)a.wp;
        const v2f* e0 = Es + py * SB_E + px;
#pragma unroll 1
        for (int p = 0; p < CP; ++p) {
            const kptr w = wd + p * SB_WD;
            const kptr pw = wp + p * 2 * COP;
            const v2f* e = e0 + p * 256;
            v2f dr[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};    // one chain per tap row
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int k = ky * 3 + kx;
                    dr[ky] = sb_fma(e[ky * SB_E + kx], (v2f){w[2 * k], w[2 * k + 1]}, dr[ky]);
                }
            v2f d = (dr[0] + dr[1]) + dr[2];
            d = sb_act<RELU6>(d + (v2f){w[20], w[21]}, a.act);   // (scale folded into the taps; w[18..19] hold 1)
#pragma unroll
            for (int n = 0; n < COP / 2; ++n) o[n] = sb_fma((v2f){d.x, d.x}, (v2f){pw[2 * n], pw[2 * n + 1]}, o[n]);
#pragma unroll
            for (int n = 0; n < COP / 2; ++n) o[n] = sb_fma((v2f){d.y, d.y}, (v2f){pw[COP + 2 * n], pw[COP + 2 * n + 1]}, o[n]);
        }
        if (gy < a.Ho && gx < a.Wo) {
            const kptr bp = (kptr)a.bp;
            T* op = a.out + (((size_t)b * a.Ho + gy) * a.Wo + gx) * a.ld_out;
#pragma unroll
            for (int n = 0; n < COP / 2; ++n) o[n] = sb_fma(o[n], (v2f){bp[2 * n], bp[2 * n + 1]}, (v2f){bp[COP + 2 * n], bp[COP + 2 * n + 1]});
#pragma unroll
            for (int n = 0; n < COP; n += 4) {
                if (n + 3 < a.Cout && (a.ld_out & 3) == 0) {
                    yr_st4<T>(op + n, make_float4(o[n / 2].x, o[n / 2].y, o[n / 2 + 1].x, o[n / 2 + 1].y));
                } else {
                    if (n < a.Cout) yr_st1<T>(op + n, o[n / 2].x);
                    if (n + 1 < a.Cout) yr_st1<T>(op + n + 1, o[n / 2].y);
                    if (n + 2 < a.Cout) yr_st1<T>(op + n + 2, o[n / 2 + 1].x);
                    if (n + 3 < a.Cout) yr_st1<T>(op + n + 3, o[n / 2 + 1].y);
                }
            }
        }
    }
}

template <int CP, int COP, class T>
static int launch_sb(const SbArgs<T>& a, bool in8, int batch, hipStream_t s) {
    constexpr size_t lds = (size_t)CP * 256 * 2 * sizeof(float);   // (COP == 0: its head doubles as the [4][2*CP] wave sums)
    static_assert(lds <= 64 * 1024, "stemblock LDS tile too large");
    static char nm[2][2][56];
    static bool named = false;
    if (!named) {
        for (int r = 0; r < 2; ++r)
            for (int u = 0; u < 2; ++u)
                snprintf(nm[r][u], sizeof(nm[r][u]), u ? "stemblock_kernel<%d,%d,%d,%s,u8>" : "stemblock_kernel<%d,%d,%d,%s>", CP, COP, r, yr_dtype_name(yr_elem<T>::dtype));
        named = true;
    }
    const bool relu6 = a.act == YR_ACT_RELU6;
    yr_note_kernel(nm[relu6 ? 1 : 0][in8 ? 1 : 0]);
    const dim3 grid((unsigned)(batch * a.tiles_x * a.tiles_y));
    if (in8) {
        if (relu6) hipLaunchKernelGGL((stemblock_kernel<CP, COP, true, T, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((stemblock_kernel<CP, COP, false, T, true>), grid, dim3(256), lds, s, a);
    } else {
        if (relu6) hipLaunchKernelGGL((stemblock_kernel<CP, COP, true, T, false>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((stemblock_kernel<CP, COP, false, T, false>), grid, dim3(256), lds, s, a);
    }
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// op fields: src[0] = dense 3-channel image; cin = 3; se_reduced = C1 (stem width); cout; k = 3; stride = 2;
// act = stem/DW activation.  With CP = round_up(C1,4)/2 channel pairs, COP = round_up(cout,8), all zero padded:
//   wgt  = stem, per channel pair:      [CP][27 taps (ky,kx,ci) x 2, times the BN scale | 1 1 | BN shift 2]   (58 floats per pair)
//   wgt2 = depthwise, per channel pair: [CP][ 9 taps (ky,kx)    x 2, times the BN scale | 1 1 | BN shift 2]   (22 floats per pair)
//   b1   = project W[2*CP][COP] (input-channel major);  b2 = project BN scale [COP] ++ shift [COP].
template <class T>
static int launch_stemblock_t(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY && op.src[0].c == 3 && op.src[0].ld == 3 && (op.src[0].dtype == YR_F32 || op.src[0].dtype == YR_U8),
               "stemblock: needs one dense 3-channel float32 (or uint8) source");
    const bool in8 = op.src[0].dtype == YR_U8;
    YR_REQUIRE(op.k == 3 && op.stride == 2, "stemblock: the stem is 3x3 stride 2");
    const yr_src& in = op.src[0];
    SbArgs<T> a;
    YR_REQUIRE(op.out_dtype == op.dtype && op.out_ld % (yr_elem<T>::vec == 8 ? 8 : 1) == 0, "stemblock: the output has the op's dtype (16-bit: out_ld %% 8 == 0)");
    a.in = in.ptr; a.in_scale = in8 ? 1.0f / 255.0f : 1.0f; a.out = (T*)op.out;
    YR_REQUIRE(op.se_reduced >= 1 && op.cout >= 1, "stemblock: bad widths (C1=%d, Cout=%d)", op.se_reduced, op.cout);
    const int c1p = yr_round_up(op.se_reduced, 4), cop = yr_round_up(op.cout, 8);
    const bool noproj = op.b1 == nullptr;   // stem + depthwise only (cout == C1): the depthwise map and its squeeze sums leave
    YR_REQUIRE(in.ptr && op.out && op.wgt && op.wgt2 && (noproj || op.b2), "stemblock: null pointer");
    a.ws = op.wgt; a.wd = op.wgt2; a.wp = op.b1; a.bp = op.b2;
    a.part = nullptr; a.ld_part = 0;
    a.Cout = op.cout;
    a.Hi = in.h; a.Wi = in.w; a.Ho = (in.h + 1) / 2; a.Wo = (in.w + 1) / 2;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w && op.out_ld >= op.cout, "stemblock: output dims mismatch");
    a.ld_out = op.out_ld;
    const int pth = (a.Ho - 1) * 2 + 3 - in.h, ptw = (a.Wo - 1) * 2 + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.act = op.act;
    a.tiles_x = (a.Wo + SB_T - 1) / SB_T; a.tiles_y = (a.Ho + SB_T - 1) / SB_T;
    if (noproj) {
        YR_REQUIRE(op.cout == op.se_reduced && op.out_ld % 4 == 0, "stemblock: without a projection the output is the depthwise map (cout == C1, out_ld %% 4 == 0)");
        if (op.gate) {   // OUTPUT: per-tile channel sums
            YR_REQUIRE(op.gate_ld >= op.cout && ((uintptr_t)op.gate % 4) == 0, "stemblock: bad squeeze-sum buffer");
            a.part = const_cast<float*>(op.gate); a.ld_part = op.gate_ld;
        }
        switch (c1p / 2) {
            case 16: return launch_sb<16, 0, T>(a, in8, batch, s);    // EfficientNet-B0 / B1 (32)
            case 20: return launch_sb<20, 0, T>(a, in8, batch, s);    // B3 (40)
            case 24: return launch_sb<24, 0, T>(a, in8, batch, s);    // B4 / B5 (48)
            default: yr_set_error("stemblock: stem width C1=%d unsupported without a projection", op.se_reduced); return YR_ERR_ARG;
        }
    }
    switch (c1p / 2 * 100 + cop) {
        case 1216: return launch_sb<12, 16, T>(a, in8, batch, s);
        case 1616: return launch_sb<16, 16, T>(a, in8, batch, s);
        case 1624: return launch_sb<16, 24, T>(a, in8, batch, s);
        case 2024: return launch_sb<20, 24, T>(a, in8, batch, s);
        case 2416: return launch_sb<24, 16, T>(a, in8, batch, s);
        case 2424: return launch_sb<24, 24, T>(a, in8, batch, s);
        default: yr_set_error("stemblock: widths C1=%d Cout=%d unsupported", op.se_reduced, op.cout); return YR_ERR_ARG;
    }
}

int yr_launch_stemblock_h(const yr_op& op, int batch, hipStream_t s);   // stemblock_h.hip: the matrix-pipe form of the 16-bit plans

bool yr_stemxr_takes(const yr_op& op);                                   // mbxr_h.hip: stem + first depthwise of the SE networks, matrix pipe
int yr_launch_stemxr(const yr_op& op, int batch, hipStream_t s);

int yr_launch_stemblock(const yr_op& op_in, int batch, hipStream_t s) {
    yr_op op = op_in;
    // k = 3 | 1 << 8: the plan asks for the matrix-pipe form of the stem + depthwise entry (image and stem kernel rounded to the
    // 16-bit type like every MFMA operand: a property of the PLAN, compiler.FUSE_STEMDW_MFMA, so that every batch size runs the same)
    const bool mfma_entry = ((op.k >> 8) & 0xff) == 1;
    op.k &= 0xff;
    if (mfma_entry) {
        YR_REQUIRE(yr_stemxr_takes(op), "stemblock: the matrix-pipe stem + depthwise form (k = 3 | 1 << 8) is not built for this op");
        return yr_launch_stemxr(op, batch, s);
    }
    if (op.dtype != YR_F32 && op.scale != nullptr) return yr_launch_stemblock_h(op, batch, s);   // (the compiler's matrix-pipe parameter layout)
    return YR_BY_DTYPE(op.dtype, launch_stemblock_t, op, batch, s);
}
