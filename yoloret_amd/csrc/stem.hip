// Stem: Conv2D 3x3 stride 2, Cin=3, TF 'SAME' padding, + BatchNorm + ReLU6/Swish.
// Replaces Conv2D + FusedBatchNormV3 + Relu6 of MobileNetV2's Conv1 [3P] and the EfficientNet
// stem (reference code/yolo3/efficientnet.py:636-645).
//
// One workgroup = an 8 x 32 tile of output pixels, one lane per pixel, all output channels in
// registers.  The 17 x 65 x 3 input halo tile is staged in LDS with coalesced row loads (every input
// element is read from HBM once); the 27 x Cout weights are wave-uniform and come through the scalar
// cache (s_load), so the inner loop is 27 LDS reads + 27*Cout FMAs per pixel.
#include "yr_common.h"

#define ST_TH 8
#define ST_TW 32
#define ST_IH (2 * ST_TH + 1)
#define ST_IW (2 * ST_TW + 1)

// T: element type of the OUTPUT map (the image, the weights and all arithmetic are float32).
template <class T>
struct StemArgs {
    const float* in;     // [B][Hi][Wi][3] dense float32 in [0,1] - or, in_u8 != 0, uint8 image bytes (x / 255 applied while staging)
    int in_u8;
    const float* w;      // [27][ldw]  (tap-major: (ky*3+kx)*3+ci), zero padded to ldw
    const float* scale;  // [ldw]
    const float* shift;  // [ldw]
    const float* wpair;  // optional [ldw/2][58]: per channel PAIR 27 taps x 2 (times the BN scale) | 1 1 | BN shift 2 (stemblock's layout)
    T* out;              // [B][Ho][Wo][ld_out]
    int B, Hi, Wi, Ho, Wo, ldw, ld_out, pad_t, pad_l, act, tiles_x, tiles_y;
};

template <int CQ, class T>  // cout quads held per lane
__global__ __launch_bounds__(256) void stem_kernel(StemArgs<T> a) {
    __shared__ float tile[ST_IH * ST_IW * 3];
    __shared__ __attribute__((aligned(16))) float wl[27 * CQ * 4];
    __shared__ __attribute__((aligned(16))) float otile[256 * CQ * 4];
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * CQ * 4; i += 256) wl[i] = a.w[i];  // ldw == CQ*4 (checked by the launcher)
    const int t = blockIdx.x;
    const int b = t / (a.tiles_x * a.tiles_y);
    const int r = t - b * a.tiles_x * a.tiles_y;
    const int ty0 = (r / a.tiles_x) * ST_TH, tx0 = (r % a.tiles_x) * ST_TW;
    const int iy0 = ty0 * 2 - a.pad_t, ix0 = tx0 * 2 - a.pad_l;
    // input tile -> LDS: ST_IH rows of ST_IW*3 contiguous floats (zero outside the image)
    // (all loads are issued before the first LDS store: one HBM round trip per workgroup, not 13)
    constexpr int NLD = (ST_IH * ST_IW * 3 + 255) / 256;
    float stage[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int i = tid + u * 256;
        const int ry = i / (ST_IW * 3), rc = i - ry * (ST_IW * 3);
        const int iy = iy0 + ry, ixc = ix0 * 3 + rc;  // ixc = ix*3 + ci
        stage[u] = 0.f;
        if (i < ST_IH * ST_IW * 3 && iy >= 0 && iy < a.Hi && ixc >= 0 && ixc < a.Wi * 3)
            stage[u] = a.in_u8 ? (float)reinterpret_cast<const unsigned char*>(a.in)[((size_t)b * a.Hi + iy) * a.Wi * 3 + ixc] * (1.0f / 255.0f)
                               : a.in[((size_t)b * a.Hi + iy) * a.Wi * 3 + ixc];
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u)
        if (tid + u * 256 < ST_IH * ST_IW * 3) tile[tid + u * 256] = stage[u];
    __syncthreads();
    const int py = tid / ST_TW, px = tid - py * ST_TW;
    float4 acc[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the tap loops stay rolled: fully unrolled, the compiler hoists all 27*CQ weight reads and spills
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
        const float* rowp = tile + (py * 2 + ky) * (ST_IW * 3) + px * 6;
#pragma unroll 3
        for (int j = 0; j < 9; ++j) {  // j = kx*3 + ci: 9 contiguous floats of the row
            const float v = rowp[j];
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                const float4 wv = *reinterpret_cast<const float4*>(wl + ((ky * 9 + j) * CQ + q) * 4);  // LDS broadcast
                acc[q].x = __builtin_fmaf(v, wv.x, acc[q].x);
                acc[q].y = __builtin_fmaf(v, wv.y, acc[q].y);
                acc[q].z = __builtin_fmaf(v, wv.z, acc[q].z);
                acc[q].w = __builtin_fmaf(v, wv.w, acc[q].w);
            }
        }
    }
    // BN + activation, then through LDS so that the tile is written as whole contiguous rows
    // (a lane owns 4*CQ consecutive floats of ONE pixel; written directly, each store instruction would
    // touch 64 different cache lines)
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        const float4 sc = *reinterpret_cast<const float4*>(a.scale + q * 4);
        const float4 sh = *reinterpret_cast<const float4*>(a.shift + q * 4);
        float4 v = make_float4(__builtin_fmaf(acc[q].x, sc.x, sh.x), __builtin_fmaf(acc[q].y, sc.y, sh.y),
                               __builtin_fmaf(acc[q].z, sc.z, sh.z), __builtin_fmaf(acc[q].w, sc.w, sh.w));
        *reinterpret_cast<float4*>(otile + (tid * CQ + q) * 4) = yr_apply_act4_t<T>(v, a.act);
    }
    __syncthreads();
    for (int i = tid; i < 256 * CQ; i += 256) {
        const int p = i / CQ, q = i - p * CQ;
        const int oy = ty0 + p / ST_TW, ox = tx0 + (p % ST_TW);
        if (oy < a.Ho && ox < a.Wo)
            yr_st4<T>(a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.ld_out + q * 4, *reinterpret_cast<const float4*>(otile + i * 4));
    }
}

// The same tile with the weights as SCALAR operands (the network-entry kernel's scheme, stemblock.hip): one channel pair per
// rolled iteration, its 58 packed floats through s_load, 27 v_pk_fma_f32 in four accumulator chains (a dependent packed FMA
// issues only every ~13th slot: tools/peak.hip), BN shift added, activation, 8-byte store into the LDS output tile.  The
// LDS-weight form above costs 27 x Cout/4 broadcast ds_read_b128 and 27 x Cout scalar FMAs per pixel and is bound by
// the LDS pipe (EfficientNet-B0 @416, 128 images: 0.32 ms at 24 TFLOP/s); this one issues half the VALU instructions and
// no weight reads from LDS.  Used when the op carries the pair-packed weights (wgt2).
typedef const float __attribute__((address_space(4))) * st_kptr;
typedef float st_v2f __attribute__((ext_vector_type(2)));
template <int CQ, class T>
__global__ __launch_bounds__(256) void stem_pair_kernel(StemArgs<T> a) {
    // output tile in LDS in the OUTPUT type (a 16-bit plan: half the bytes, five workgroups per CU instead of three);
    // pixel stride: an odd number of pair slots, so the pair stores of a half wave spread over all banks
    constexpr int OS = CQ * 4 + 2;
    __shared__ float tile[ST_IH * ST_IW * 3];
    __shared__ __attribute__((aligned(8))) T otile[256 * OS];
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    const int b = t / (a.tiles_x * a.tiles_y);
    const int r = t - b * a.tiles_x * a.tiles_y;
    const int ty0 = (r / a.tiles_x) * ST_TH, tx0 = (r % a.tiles_x) * ST_TW;
    const int iy0 = ty0 * 2 - a.pad_t, ix0 = tx0 * 2 - a.pad_l;
    constexpr int NLD = (ST_IH * ST_IW * 3 + 255) / 256;
    float stage[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int i = tid + u * 256;
        const int ry = i / (ST_IW * 3), rc = i - ry * (ST_IW * 3);
        const int iy = iy0 + ry, ixc = ix0 * 3 + rc;
        stage[u] = 0.f;
        if (i < ST_IH * ST_IW * 3 && iy >= 0 && iy < a.Hi && ixc >= 0 && ixc < a.Wi * 3)
            stage[u] = a.in_u8 ? (float)reinterpret_cast<const unsigned char*>(a.in)[((size_t)b * a.Hi + iy) * a.Wi * 3 + ixc] * (1.0f / 255.0f)
                               : a.in[((size_t)b * a.Hi + iy) * a.Wi * 3 + ixc];
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u)
        if (tid + u * 256 < ST_IH * ST_IW * 3) tile[tid + u * 256] = stage[u];
    __syncthreads();
    const int py = tid / ST_TW, px = tid - py * ST_TW;
    float in[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int j = 0; j < 9; ++j) in[ky * 9 + j] = tile[(py * 2 + ky) * (ST_IW * 3) + px * 6 + j];
    const st_kptr ws = (st_kptr)a.wpair;
#pragma unroll 1
    for (int p = 0; p < 2 * CQ; ++p) {
        const st_kptr w = ws + p * 58;
        st_v2f c0 = {0.f, 0.f}, c1 = {0.f, 0.f}, c2 = {0.f, 0.f}, c3 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 24; k += 4) {
            c0 = __builtin_elementwise_fma((st_v2f){in[k], in[k]}, (st_v2f){w[2 * k], w[2 * k + 1]}, c0);
            c1 = __builtin_elementwise_fma((st_v2f){in[k + 1], in[k + 1]}, (st_v2f){w[2 * k + 2], w[2 * k + 3]}, c1);
            c2 = __builtin_elementwise_fma((st_v2f){in[k + 2], in[k + 2]}, (st_v2f){w[2 * k + 4], w[2 * k + 5]}, c2);
            c3 = __builtin_elementwise_fma((st_v2f){in[k + 3], in[k + 3]}, (st_v2f){w[2 * k + 6], w[2 * k + 7]}, c3);
        }
        c0 = __builtin_elementwise_fma((st_v2f){in[24], in[24]}, (st_v2f){w[48], w[49]}, c0);
        c1 = __builtin_elementwise_fma((st_v2f){in[25], in[25]}, (st_v2f){w[50], w[51]}, c1);
        c2 = __builtin_elementwise_fma((st_v2f){in[26], in[26]}, (st_v2f){w[52], w[53]}, c2);
        c0 = (c0 + c1) + (c2 + c3);
        c0 += (st_v2f){w[56], w[57]};
        c0 = (st_v2f){yr_apply_act_t<T>(c0.x, a.act), yr_apply_act_t<T>(c0.y, a.act)};
        yr_st2<T>(otile + tid * OS + 2 * p, c0.x, c0.y);
    }
    __syncthreads();
    for (int i = tid; i < 256 * CQ; i += 256) {
        const int p = i / CQ, q = i - p * CQ;
        const int oy = ty0 + p / ST_TW, ox = tx0 + (p % ST_TW);
        if (oy < a.Ho && ox < a.Wo) {   // (already rounded: a copy of 4 elements)
            const T* o = otile + p * OS + q * 4;
            T* g = a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.ld_out + q * 4;
            typedef T t2 __attribute__((ext_vector_type(2)));
            typedef T t4 __attribute__((ext_vector_type(4)));
            const t2 lo = *reinterpret_cast<const t2*>(o), hi2 = *reinterpret_cast<const t2*>(o + 2);   // (LDS rows are pair aligned)
            *reinterpret_cast<t4*>(g) = (t4){lo[0], lo[1], hi2[0], hi2[1]};
        }
    }
}

template <int CQ, class T>
static int launch_stem(const StemArgs<T>& a, hipStream_t s) {
    static char nm[32];
    static const int nm_len = snprintf(nm, sizeof(nm), "stem_kernel<%d,%s>", CQ, yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    static char nmp[40];
    static const int nmp_len = snprintf(nmp, sizeof(nmp), "stem_pair_kernel<%d,%s>", CQ, yr_dtype_name(yr_elem<T>::dtype));
    (void)nmp_len;
    if (a.wpair) {
        yr_note_kernel(nmp);
        hipLaunchKernelGGL((stem_pair_kernel<CQ, T>), dim3((unsigned)(a.B * a.tiles_x * a.tiles_y)), dim3(256), 0, s, a);
        YR_LAUNCH_CHECK();
        return YR_OK;
    }
    yr_note_kernel(nm);
    hipLaunchKernelGGL((stem_kernel<CQ, T>), dim3((unsigned)(a.B * a.tiles_x * a.tiles_y)), dim3(256), 0, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <class T>
static int launch_stem_t(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY && op.src[0].c == 3 && op.src[0].ld == 3 && (op.src[0].dtype == YR_F32 || op.src[0].dtype == YR_U8),
               "stem: needs one dense 3-channel float32 (or uint8) source");
    YR_REQUIRE(op.out_dtype == op.dtype, "stem: the output has the op's dtype");
    YR_REQUIRE(op.k == 3 && op.stride == 2, "stem: only 3x3 stride 2 is supported");
    const yr_src& in = op.src[0];
    StemArgs<T> a;
    a.in = (const float*)in.ptr; a.in_u8 = in.dtype == YR_U8; a.w = op.wgt; a.scale = op.scale; a.shift = op.shift; a.wpair = op.wgt2; a.out = (T*)op.out;
    YR_REQUIRE(a.in && a.w && a.scale && a.shift && a.out, "stem: null pointer");
    a.B = batch; a.Hi = in.h; a.Wi = in.w; a.Ho = (in.h + 1) / 2; a.Wo = (in.w + 1) / 2;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "stem: output dims mismatch");
    a.ldw = yr_round_up(op.cout, 4); a.ld_out = op.out_ld;
    YR_REQUIRE(op.out_ld % yr_elem<T>::vec == 0 && op.out_ld >= a.ldw, "stem: out_ld must be a multiple of %d and >= round_up(cout,4)", yr_elem<T>::vec);
    const int pth = (a.Ho - 1) * 2 + 3 - in.h, ptw = (a.Wo - 1) * 2 + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.act = op.act;
    a.tiles_x = (a.Wo + ST_TW - 1) / ST_TW; a.tiles_y = (a.Ho + ST_TH - 1) / ST_TH;
    YR_REQUIRE((long long)batch * a.tiles_x * a.tiles_y < (1ll << 31), "stem: grid too large");
    const int cq = a.ldw / 4;
    // the kernel holds exactly ldw/4 cout quads per lane (no guards in the unrolled FMA block)
    switch (cq) {
        case 2: return launch_stem<2, T>(a, s);
        case 4: return launch_stem<4, T>(a, s);
        case 6: return launch_stem<6, T>(a, s);      // MobileNetV2 x0.75 (24)
        case 8: return launch_stem<8, T>(a, s);      // EfficientNet-B0 (32)
        case 10: return launch_stem<10, T>(a, s);    // EfficientNet-B3 (40)
        case 12: return launch_stem<12, T>(a, s);    // MobileNetV2 x1.4 (48), EfficientNet-B4
        case 14: return launch_stem<14, T>(a, s);
        case 16: return launch_stem<16, T>(a, s);
        default: yr_set_error("stem: %d output channels: only multiples of 8 up to 64 are supported", op.cout); return YR_ERR_ARG;
    }
}

int yr_launch_stem(const yr_op& op, int batch, hipStream_t s) { return YR_BY_DTYPE(op.dtype, launch_stem_t, op, batch, s); }
