// Fused inverted-residual block, lane-per-pixel formulation: expand 1x1 + BN + act -> depthwise 3x3
// (stride 1|2, TF SAME) + BN + act -> project 1x1 + BN (+ residual), for NARROW block inputs (Cin <= 32:
// MobileNetV2 block_1..5 [3P], reference code/yolo3/override.py:339; SE-free MBConv, efficientnet.py:467-536).
//
// Why not the MFMA formulation (mbconv.hip) here: with K = Cin <= 32 the fp32 MFMA path spends its time
// moving fragments through LDS and waiting at phase barriers (measured on MI355X: matrix pipe 25-31 % busy,
// LDS 35 %), and fp32 MFMA has no higher peak than packed fp32 FMA (both 256 flop/clk/CU).  So, as in
// stemblock.hip, every lane owns one pixel:
//   * the pixel's Cin input channels live in VGPRs for the whole kernel (also the residual source);
//   * all weights are wave-uniform -> constant-address-space reads (s_load, SGPR operands), no LDS/VGPR traffic;
//   * channel pairs are float2 so all FMAs issue as v_pk_fma_f32;
//   * the expanded tensor goes through LDS only as Es[pair][256 halo pixels] (float2 per lane, conflict-free),
//     in double-buffered chunks of ML_CH pairs: one barrier per chunk;
//   * the depthwise result feeds the projection from registers; project accumulators stay in VGPRs.
// Stride 1: workgroup = 16 x 16 halo pixels = 14 x 14 outputs; a lane's output pixel is its own halo pixel
// (border lanes idle in the depthwise/project phase).  Stride 2: workgroup = 15 x 17 halo pixels = 7 x 8
// outputs; the depthwise/project phase gives each WAVE a quarter of the chunk's channel pairs for all 56
// outputs (weights stay wave-uniform), and the four partial projections are summed through LDS at the end.
#include "yr_common.h"

#define ML_CH 8   // channel pairs per Es chunk (2 KB each per buffer)

typedef const float __attribute__((address_space(4))) * kptr;
typedef float v2f __attribute__((ext_vector_type(2)));

// T: element type of the block input and output (float32, or bf16 / f16 storage: widened on load - the pixel's
// channels then live in float32 registers as before - and rounded once at the store).  Weights: float32.
template <class T>
struct MlArgs {
    const T* x; T* out;
    const float* we;   // expand     [P][CINP x 2 | scale 2 | shift 2]
    const float* wd;   // depthwise  [P][9 x 2    | scale 2 | shift 2]
    const float* wp;   // project    [2P][COP]
    const float* bp;   // project BN scale [COP] ++ shift [COP]
    int Hi, Wi, Ho, Wo, Cin, Cout, ld_in, ld_out, npairs, pad_t, pad_l, act, tiles_x, tiles_y, has_res;
};

__device__ __forceinline__ v2f ml_fma(v2f x, v2f y, v2f z) { return __builtin_elementwise_fma(x, y, z); }

template <bool RELU6>
__device__ __forceinline__ v2f ml_act(v2f v, int act) {
    if (RELU6) return (v2f){fminf(fmaxf(v.x, 0.f), 6.f), fminf(fmaxf(v.y, 0.f), 6.f)};
    return (v2f){yr_apply_act(v.x, act), yr_apply_act(v.y, act)};
}

// expand one channel pair for this lane's pixel: act(sum_k x[k] * W'[k][pair] + shift), W' = W * BN scale (folded by the
// host: the packed rows keep their scale slot, holding 1).  `hi` is the lane's upper clamp: 6 inside the image, 0 outside
// - TF pads the EXPANDED tensor with zeros for the depthwise, and min(max(v, 0), 0) is that zero for free.  (With the BN
// multiply-add, two moves of the shift into VGPRs and two selects this loop was 25 VALU instructions for 16 useful ones
// on block_1, and it runs on every lane of the 15 x 17 halo: 72 % of that kernel's instructions.)
template <int CQ, bool RELU6>
__device__ __forceinline__ v2f ml_expand(const float4 (&x)[CQ], kptr w, int act, float hi) {
    // FOUR independent accumulator chains: a dependent v_pk_fma_f32 issues only every ~13th slot (tools/peak.hip: two
    // chains per wave cap at 89 TF however many waves are resident, sixteen reach 128 TF), and this loop is most of the kernel
    v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        a0 = ml_fma((v2f){x[q].x, x[q].x}, (v2f){w[8 * q + 0], w[8 * q + 1]}, a0);
        a1 = ml_fma((v2f){x[q].y, x[q].y}, (v2f){w[8 * q + 2], w[8 * q + 3]}, a1);
        a2 = ml_fma((v2f){x[q].z, x[q].z}, (v2f){w[8 * q + 4], w[8 * q + 5]}, a2);
        a3 = ml_fma((v2f){x[q].w, x[q].w}, (v2f){w[8 * q + 6], w[8 * q + 7]}, a3);
    }
    a0 = (a0 + a1) + (a2 + a3);
    a0 += (v2f){w[8 * CQ + 2], w[8 * CQ + 3]};
    if (RELU6) return (v2f){__builtin_amdgcn_fmed3f(a0.x, 0.f, hi), __builtin_amdgcn_fmed3f(a0.y, 0.f, hi)};
    const v2f v = {yr_apply_act(a0.x, act), yr_apply_act(a0.y, act)};
    return hi > 0.f ? v : (v2f){0.f, 0.f};
}

template <int CQ, class T>
__device__ __forceinline__ void ml_load_x(float4 (&x)[CQ], const MlArgs<T>& a, int b, int hy, int hx, bool inside) {
#pragma unroll
    for (int q = 0; q < CQ; ++q) x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside) {
        const T* xp = a.x + ((size_t)(b * a.Hi + hy) * a.Wi + hx) * a.ld_in;
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            if (q * 4 < a.Cin) {
                float4 v = yr_ld4<T>(xp + q * 4);
                const int rem = a.Cin - q * 4;  // pad lanes of the source may hold anything
                if (rem < 4) { v.w = 0.f; if (rem < 3) v.z = 0.f; if (rem < 2) v.y = 0.f; }
                x[q] = v;
            }
        }
    }
}

// One expanded-channel pair of the second phase: depthwise over the 3x3 taps e[ky*ROW + kx*STEP'] (float2
// per tap), BN, act, then the pair's two rows of the projection into o[].  All nine LDS reads and all
// scalar weight loads are issued first and pinned by one empty asm, so the wave pays ONE latency per pair
// (left to itself the compiler interleaves each read with its use: nine serialised round trips).
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v16f __attribute__((address_space(4), aligned(4))) * k16ptr;
typedef const v8f __attribute__((address_space(4), aligned(4))) * k8ptr;
typedef const v4f __attribute__((address_space(4), aligned(4))) * k4ptr;
typedef const v2f __attribute__((address_space(4), aligned(4))) * k2ptr;

template <int COP, int ROW, bool RELU6>
__device__ __forceinline__ void ml_dw_project(const v2f* e, kptr w, kptr pw, v2f (&o)[COP / 2], int act) {
    static_assert(COP % 8 == 0 && COP <= 48, "project width");
    v2f ev[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) ev[ky * 3 + kx] = e[ky * ROW + kx];
    v16f wa = *(k16ptr)w;            // taps 0..7
    v4f wb = *(k4ptr)(w + 16);       // tap 8, BN scale
    v2f wc = *(k2ptr)(w + 20);       // BN shift
#define ML_PIN_EV "+v"(ev[0]), "+v"(ev[1]), "+v"(ev[2]), "+v"(ev[3]), "+v"(ev[4]), "+v"(ev[5]), "+v"(ev[6]), "+v"(ev[7]), "+v"(ev[8])
    constexpr int NR = COP / 8;      // one projection row = COP floats = NR groups of 8
    constexpr bool BOTH = COP <= 24; // both rows fit the SGPR budget next to the depthwise weights (22 + 2*COP <= 70)
    v8f r0[NR], r1[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r0[i] = *(k8ptr)(pw + 8 * i);
    if constexpr (BOTH) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r1[i] = *(k8ptr)(pw + COP + 8 * i);
    }
    if constexpr (NR == 2) asm volatile("" : ML_PIN_EV, "+s"(wa), "+s"(wb), "+s"(wc), "+s"(r0[0]), "+s"(r0[1]), "+s"(r1[0]), "+s"(r1[1]));
    else if constexpr (NR == 3) asm volatile("" : ML_PIN_EV, "+s"(wa), "+s"(wb), "+s"(wc), "+s"(r0[0]), "+s"(r0[1]), "+s"(r0[2]), "+s"(r1[0]), "+s"(r1[1]), "+s"(r1[2]));
    else if constexpr (NR == 4) asm volatile("" : ML_PIN_EV, "+s"(wa), "+s"(wb), "+s"(wc), "+s"(r0[0]), "+s"(r0[1]), "+s"(r0[2]), "+s"(r0[3]));
    else if constexpr (NR == 5) asm volatile("" : ML_PIN_EV, "+s"(wa), "+s"(wb), "+s"(wc), "+s"(r0[0]), "+s"(r0[1]), "+s"(r0[2]), "+s"(r0[3]), "+s"(r0[4]));
    else asm volatile("" : ML_PIN_EV, "+s"(wa), "+s"(wb), "+s"(wc), "+s"(r0[0]), "+s"(r0[1]), "+s"(r0[2]), "+s"(r0[3]), "+s"(r0[4]), "+s"(r0[5]));
#undef ML_PIN_EV
    v2f d = {0.f, 0.f}, d1 = {0.f, 0.f}, d2 = {0.f, 0.f};   // one chain per tap row (see ml_expand)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d = ml_fma(ev[k], (v2f){wa[2 * k], wa[2 * k + 1]}, d);
        d1 = ml_fma(ev[3 + k], (v2f){wa[6 + 2 * k], wa[7 + 2 * k]}, d1);
    }
    d2 = ml_fma(ev[6], (v2f){wa[12], wa[13]}, d2);
    d2 = ml_fma(ev[7], (v2f){wa[14], wa[15]}, d2);
    d2 = ml_fma(ev[8], (v2f){wb[0], wb[1]}, d2);
    d = (d + d1) + d2;
    d = ml_act<RELU6>(d + wc, act);          // (BN scale folded into the taps by the host; wb[2..3] hold 1)
#pragma unroll
    for (int n = 0; n < COP / 2; ++n) o[n] = ml_fma((v2f){d.x, d.x}, (v2f){r0[n / 4][(2 * n) % 8], r0[n / 4][(2 * n) % 8 + 1]}, o[n]);
    if constexpr (!BOTH) {  // wide projections: the second row is loaded (one burst, one wait) once the first is consumed
#pragma unroll
        for (int i = 0; i < NR; ++i) r1[i] = *(k8ptr)(pw + COP + 8 * i);
        if constexpr (NR == 4) asm volatile("" : "+s"(r1[0]), "+s"(r1[1]), "+s"(r1[2]), "+s"(r1[3]));
        else if constexpr (NR == 5) asm volatile("" : "+s"(r1[0]), "+s"(r1[1]), "+s"(r1[2]), "+s"(r1[3]), "+s"(r1[4]));
        else asm volatile("" : "+s"(r1[0]), "+s"(r1[1]), "+s"(r1[2]), "+s"(r1[3]), "+s"(r1[4]), "+s"(r1[5]));
    }
#pragma unroll
    for (int n = 0; n < COP / 2; ++n) o[n] = ml_fma((v2f){d.y, d.y}, (v2f){r1[n / 4][(2 * n) % 8], r1[n / 4][(2 * n) % 8 + 1]}, o[n]);
}

// ------------------------------------------------------------------------------------------ stride 1
// IDENT: a block WITHOUT an expand conv (expand ratio 1: depthwise + project on the block input itself, efficientnet.py:
// 467-484 skipped when expand_ratio == 1 - every EfficientNet stage 1): the "expanded" tensor is the input, copied to LDS
// pair by pair (zero outside the image = TF's padding of the depthwise input); a.we is unused.
template <int CQ, int COP, bool RELU6, class T, bool IDENT = false>
__global__ __launch_bounds__(256, (4 * CQ + COP > 72 ? 2 : 4)) void mblane_s1_kernel(MlArgs<T> a) {
    constexpr int TL = 14, WE = 8 * CQ + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    v2f* Es = reinterpret_cast<v2f*>(lds);  // [2][ML_CH][256]
    const int tid = threadIdx.x;
    const int t = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int tpi = a.tiles_x * a.tiles_y;
    const int b = t / tpi, r = t - b * tpi;
    const int ty = r / a.tiles_x;
    const int oy0 = ty * TL, ox0 = (r - ty * a.tiles_x) * TL;
    const int ey = tid >> 4, ex = tid & 15;
    const int hy = oy0 - 1 + ey, hx = ox0 - 1 + ex;  // stride 1, 3x3 SAME: pad 1
    const bool inside = hy >= 0 && hy < a.Hi && hx >= 0 && hx < a.Wi;
    const bool is_out = inside && ey >= 1 && ey <= TL && ex >= 1 && ex <= TL;
    float4 x[CQ];
    ml_load_x<CQ, T>(x, a, b, hy, hx, inside);
    const float hi = inside ? 6.f : 0.f;     // (relu6's upper clamp; 0 = the expanded tensor's zero padding)
    v2f o[COP / 2];
#pragma unroll
    for (int n = 0; n < COP / 2; ++n) o[n] = (v2f){0.f, 0.f};
    const kptr we = (kptr)a.we, wd = (kptr)a.wd, wp = (kptr)a.wp;

    if constexpr (IDENT) {
        constexpr int NP = (2 * CQ + ML_CH - 1) / ML_CH * ML_CH;   // == a.npairs (the launcher checks)
#pragma unroll
        for (int p0 = 0; p0 < NP; p0 += ML_CH) {                   // unrolled: the pair picks its registers statically
            v2f* buf = Es + ((p0 / ML_CH) & 1) * (ML_CH * 256);
#pragma unroll
            for (int j = 0; j < ML_CH; ++j) {
                const int pr = p0 + j;
                v2f v = {0.f, 0.f};
                if (pr < 2 * CQ) v = (pr & 1) ? (v2f){x[pr / 2 < CQ ? pr / 2 : 0].z, x[pr / 2 < CQ ? pr / 2 : 0].w}
                                              : (v2f){x[pr / 2 < CQ ? pr / 2 : 0].x, x[pr / 2 < CQ ? pr / 2 : 0].y};
                buf[j * 256 + tid] = v;
            }
            __syncthreads();
            const int jn = 2 * CQ - p0 < ML_CH ? 2 * CQ - p0 : ML_CH;   // the zero pairs that pad the parameter rows are skipped
            if (is_out) {
#pragma unroll 1
                for (int j = 0; j < jn; ++j) {
                    const kptr w = wd + (p0 + j) * 22;
                    const kptr pw = wp + (p0 + j) * 2 * COP;
                    ml_dw_project<COP, 16, RELU6>(buf + j * 256 + tid - 17, w, pw, o, a.act);
                }
            }
        }
    } else {
    for (int p0 = 0; p0 < a.npairs; p0 += ML_CH) {
        v2f* buf = Es + ((p0 / ML_CH) & 1) * (ML_CH * 256);
#pragma unroll 1
        for (int j = 0; j < ML_CH; ++j) {
            buf[j * 256 + tid] = ml_expand<CQ, RELU6>(x, we + (p0 + j) * WE, a.act, hi);
        }
        __syncthreads();  // also orders this chunk's writes after the readers of the same buffer two chunks back
        if (is_out) {
#pragma unroll 1
            for (int j = 0; j < ML_CH; ++j) {
                const kptr w = wd + (p0 + j) * 22;
                const kptr pw = wp + (p0 + j) * 2 * COP;
                ml_dw_project<COP, 16, RELU6>(buf + j * 256 + tid - 17, w, pw, o, a.act);
            }
        }
    }
    }
    (void)hi; (void)we;
    if (is_out && hy < a.Ho && hx < a.Wo) {
        const kptr bp = (kptr)a.bp;
        T* op = a.out + ((size_t)(b * a.Ho + hy) * a.Wo + hx) * a.ld_out;
#pragma unroll
        for (int n = 0; n < COP / 2; ++n) o[n] = ml_fma(o[n], (v2f){bp[2 * n], bp[2 * n + 1]}, (v2f){bp[COP + 2 * n], bp[COP + 2 * n + 1]});
#pragma unroll
        for (int n = 0; n < COP; n += 4) {
            float4 v = make_float4(o[n / 2].x, o[n / 2].y, o[n / 2 + 1].x, o[n / 2 + 1].y);
            if (a.has_res && n / 4 < CQ) { const float4 rx = x[n / 4 < CQ ? n / 4 : 0]; v.x += rx.x; v.y += rx.y; v.z += rx.z; v.w += rx.w; }
            if (n + 3 < a.Cout && (a.ld_out & 3) == 0) {
                yr_st4<T>(op + n, v);
            } else {
                if (n < a.Cout) yr_st1<T>(op + n, v.x);
                if (n + 1 < a.Cout) yr_st1<T>(op + n + 1, v.y);
                if (n + 2 < a.Cout) yr_st1<T>(op + n + 2, v.z);
                if (n + 3 < a.Cout) yr_st1<T>(op + n + 3, v.w);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ stride 2
template <int CQ, int COP, bool RELU6, class T>
__global__ __launch_bounds__(256, (4 * CQ + COP > 72 ? 2 : 4)) void mblane_s2_kernel(MlArgs<T> a) {
    constexpr int TH = 7, TW = 8, IH = 2 * TH + 1, IW = 2 * TW + 1, WE = 8 * CQ + 4;
    static_assert(IH * IW <= 256 && COP % 8 == 0, "tile / width assumptions");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    v2f* Es = reinterpret_cast<v2f*>(lds);  // [2][ML_CH][256]; reused as the reduction buffer [4][COP/2][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform: keeps the weight reads scalar
    const int t = (int)yr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int tpi = a.tiles_x * a.tiles_y;
    const int b = t / tpi, r = t - b * tpi;
    const int ty = r / a.tiles_x;
    const int oy0 = ty * TH, ox0 = (r - ty * a.tiles_x) * TW;
    const int ey = tid / IW, ex = tid - ey * IW;
    const int hy = 2 * oy0 - a.pad_t + ey, hx = 2 * ox0 - a.pad_l + ex;
    const bool inside = tid < IH * IW && hy >= 0 && hy < a.Hi && hx >= 0 && hx < a.Wi;
    float4 x[CQ];
    ml_load_x<CQ, T>(x, a, b, hy, hx, inside);
    const float hi = inside ? 6.f : 0.f;
    // depthwise/project phase: lane = output pixel, wave = which quarter of each chunk's pairs
    const int py = lane >> 3, px = lane & 7;
    const int gy = oy0 + py, gx = ox0 + px;
    const bool is_out = lane < TH * TW && gy < a.Ho && gx < a.Wo;
    v2f o[COP / 2];
#pragma unroll
    for (int n = 0; n < COP / 2; ++n) o[n] = (v2f){0.f, 0.f};
    const kptr we = (kptr)a.we, wd = (kptr)a.wd, wp = (kptr)a.wp;

    for (int p0 = 0; p0 < a.npairs; p0 += ML_CH) {
        v2f* buf = Es + ((p0 / ML_CH) & 1) * (ML_CH * 256);
#pragma unroll 1
        for (int j = 0; j < ML_CH; ++j) {
            buf[j * 256 + tid] = ml_expand<CQ, RELU6>(x, we + (p0 + j) * WE, a.act, hi);
        }
        __syncthreads();
        if (is_out) {
#pragma unroll 1
            for (int j = wave; j < ML_CH; j += 4) {
                const kptr w = wd + (p0 + j) * 22;
                const kptr pw = wp + (p0 + j) * 2 * COP;
                ml_dw_project<COP, IW, RELU6>(buf + j * 256 + (2 * py) * IW + 2 * px, w, pw, o, a.act);
            }
        }
    }
    // ---- sum the four waves' partial projections: red[wave][pair n][lane]
    __syncthreads();  // every reader of Es is done
    v2f* red = Es;
#pragma unroll
    for (int n = 0; n < COP / 2; ++n) red[(wave * (COP / 2) + n) * 64 + lane] = o[n];
    __syncthreads();
    if (is_out) {
        constexpr int NQ = COP / 8;  // pairs per thread: this thread finishes channels [wave*COP/4, (wave+1)*COP/4)
        const kptr bp = (kptr)a.bp;
        T* op = a.out + ((size_t)(b * a.Ho + gy) * a.Wo + gx) * a.ld_out;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int n2 = wave * NQ + i;
            v2f s = red[n2 * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) s += red[(w * (COP / 2) + n2) * 64 + lane];
            // wave is uniform, so these are still scalar loads
            s = ml_fma(s, (v2f){bp[2 * n2], bp[2 * n2 + 1]}, (v2f){bp[COP + 2 * n2], bp[COP + 2 * n2 + 1]});
            if (2 * n2 + 1 < a.Cout) yr_st2<T>(op + 2 * n2, s.x, s.y);
            else if (2 * n2 < a.Cout) yr_st1<T>(op + 2 * n2, s.x);
        }
    }
}

template <int S, int CQ, int COP, class T>
static int launch_ml(const MlArgs<T>& a, int batch, hipStream_t s) {
    constexpr size_t es = (size_t)2 * ML_CH * 256 * sizeof(v2f), red = (size_t)4 * (COP / 2) * 64 * sizeof(v2f);
    constexpr size_t lds = (S == 2 && red > es) ? red : es;  // stride 2 reuses Es as the cross-wave reduction buffer
    // (named as rocprof prints the symbol - template arguments in order - so that profiles can be joined by name)
    static char nm[2][56];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), S == 1 ? "mblane_s%d_kernel<%d,%d,0,%s,0>" : "mblane_s%d_kernel<%d,%d,0,%s>", S, CQ, COP, yr_dtype_name(yr_elem<T>::dtype)) +
                              snprintf(nm[1], sizeof(nm[1]), S == 1 ? "mblane_s%d_kernel<%d,%d,1,%s,0>" : "mblane_s%d_kernel<%d,%d,1,%s>", S, CQ, COP, yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    yr_note_kernel(nm[a.act == YR_ACT_RELU6 ? 1 : 0]);
    const dim3 grid((unsigned)(batch * a.tiles_x * a.tiles_y));
    if (S == 1) {
        if (a.act == YR_ACT_RELU6) hipLaunchKernelGGL((mblane_s1_kernel<CQ, COP, true, T>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((mblane_s1_kernel<CQ, COP, false, T>), grid, dim3(256), lds, s, a);
    } else {
        if (a.act == YR_ACT_RELU6) hipLaunchKernelGGL((mblane_s2_kernel<CQ, COP, true, T>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((mblane_s2_kernel<CQ, COP, false, T>), grid, dim3(256), lds, s, a);
    }
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// blocks without an expand conv (stride 1): the widths built
template <int CQ, int COP, class T>
static int launch_ml_ident(const MlArgs<T>& a, int batch, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * ML_CH * 256 * sizeof(v2f);
    static char nm[2][56];
    static const int nm_len = snprintf(nm[0], sizeof(nm[0]), "mblane_s1_kernel<%d,%d,0,%s,1>", CQ, COP, yr_dtype_name(yr_elem<T>::dtype)) +
                              snprintf(nm[1], sizeof(nm[1]), "mblane_s1_kernel<%d,%d,1,%s,1>", CQ, COP, yr_dtype_name(yr_elem<T>::dtype));
    (void)nm_len;
    yr_note_kernel(nm[a.act == YR_ACT_RELU6 ? 1 : 0]);
    YR_REQUIRE(a.npairs == (2 * CQ + ML_CH - 1) / ML_CH * ML_CH, "mblane: a block without expand conv has Cexp == Cin");
    const dim3 grid((unsigned)(batch * a.tiles_x * a.tiles_y));
    if (a.act == YR_ACT_RELU6) hipLaunchKernelGGL((mblane_s1_kernel<CQ, COP, true, T, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((mblane_s1_kernel<CQ, COP, false, T, true>), grid, dim3(256), lds, s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

template <int S, class T>
static int launch_ml_widths(const MlArgs<T>& a, int cq, int cop, int batch, hipStream_t s) {
    switch (cq * 100 + cop) {
        case 416: return launch_ml<S, 4, 16, T>(a, batch, s);
        case 424: return launch_ml<S, 4, 24, T>(a, batch, s);
        case 624: return launch_ml<S, 6, 24, T>(a, batch, s);
        case 632: return launch_ml<S, 6, 32, T>(a, batch, s);
        case 640: return launch_ml<S, 6, 40, T>(a, batch, s);
        case 832: return launch_ml<S, 8, 32, T>(a, batch, s);
        case 840: return launch_ml<S, 8, 40, T>(a, batch, s);
        case 648: return launch_ml<S, 6, 48, T>(a, batch, s);
        case 848: return launch_ml<S, 8, 48, T>(a, batch, s);
        default: yr_set_error("mblane: widths Cin=%d Cout=%d unsupported", a.Cin, a.Cout); return YR_ERR_ARG;
    }
}

// op fields: src[0] = block input (ld % 4 == 0); se_reduced = expanded width Cexp; k = 3; stride = 1|2;
// act = expand/DW activation; res (optional) must be the block input itself.  Parameters are packed per
// expanded-channel PAIR, P = round_up(ceil(Cexp/2), 8) pairs, CINP = round_up(Cin,4), COP = round_up(Cout,8),
// zero padded (a zero pair contributes act(0) * 0 = 0):
//   wgt  = expand     [P][CINP x 2 (input channel major), times the BN scale | 1 1 | BN shift 2]
//   wgt2 = depthwise  [P][9 taps x 2, times the BN scale | 1 1 | BN shift 2]
//   b1   = project    W[2P][COP] (expanded-channel major);   b2 = project BN scale [COP] ++ shift [COP].
template <class T>
static int launch_mblane_t(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY, "mblane: needs one identity source");
    const yr_src& in = op.src[0];
    YR_REQUIRE(op.k == 3 && (op.stride == 1 || op.stride == 2), "mblane: only 3x3 stride 1|2 is fused");
    YR_REQUIRE(in.ptr && op.out && op.wgt2 && op.b1 && op.b2, "mblane: null pointer");
    const bool ident = op.wgt == nullptr;   // no expand conv: depthwise + project on the block input
    if (ident) YR_REQUIRE(op.stride == 1 && op.se_reduced == in.c, "mblane: a block without expand weights needs stride 1 and Cexp == Cin");
    YR_REQUIRE(in.ld % yr_elem<T>::vec == 0 && op.out_ld % yr_elem<T>::vec == 0 && in.c == op.cin && in.ld >= yr_round_up(in.c, 4),
               "mblane: channel strides must be multiples of %d", yr_elem<T>::vec);
    YR_REQUIRE(((uintptr_t)in.ptr) % 16 == 0 && ((uintptr_t)op.out) % 8 == 0, "mblane: pointers must be 16-byte aligned");
    YR_REQUIRE(op.se_reduced >= 1 && op.cout >= 1 && op.out_ld >= op.cout, "mblane: bad widths");
    MlArgs<T> a;
    YR_REQUIRE(op.out_dtype == op.dtype && in.dtype == op.dtype, "mblane: input and output have the op's dtype");
    a.x = (const T*)in.ptr; a.out = (T*)op.out;
    a.we = op.wgt; a.wd = op.wgt2; a.wp = op.b1; a.bp = op.b2;
    a.Cin = in.c; a.Cout = op.cout;
    a.npairs = yr_round_up((op.se_reduced + 1) / 2, ML_CH);
    a.Hi = in.h; a.Wi = in.w; a.Ho = (in.h + op.stride - 1) / op.stride; a.Wo = (in.w + op.stride - 1) / op.stride;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "mblane: output dims mismatch");
    a.ld_in = in.ld; a.ld_out = op.out_ld;
    const int pth = (a.Ho - 1) * op.stride + 3 - in.h, ptw = (a.Wo - 1) * op.stride + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.has_res = op.res != nullptr;
    if (a.has_res) YR_REQUIRE(op.res == in.ptr && op.stride == 1 && in.c == op.cout, "mblane: the residual must be the block input (stride 1, Cin == Cout)");
    a.act = op.act;
    const int cq = yr_round_up(in.c, 4) / 4, cop = yr_round_up(op.cout, 8);
    if (op.stride == 1) {
        a.tiles_x = (a.Wo + 13) / 14; a.tiles_y = (a.Ho + 13) / 14;
        if (ident) {
            if (cq == 6 && cop == 24) return launch_ml_ident<6, 24, T>(a, batch, s);
            if (cq == 4 && cop == 16) return launch_ml_ident<4, 16, T>(a, batch, s);
            if (cq == 8 && cop == 32) return launch_ml_ident<8, 32, T>(a, batch, s);
            yr_set_error("mblane: widths Cin=%d Cout=%d unsupported without expand conv", a.Cin, a.Cout);
            return YR_ERR_ARG;
        }
        return launch_ml_widths<1, T>(a, cq, cop, batch, s);
    }
    YR_REQUIRE(a.ld_out % 2 == 0, "mblane: stride-2 stores need an even output channel stride");
    a.tiles_x = (a.Wo + 7) / 8; a.tiles_y = (a.Ho + 6) / 7;
    return launch_ml_widths<2, T>(a, cq, cop, batch, s);
}

int yr_launch_mblane(const yr_op& op, int batch, hipStream_t s) { return YR_BY_DTYPE(op.dtype, launch_mblane_t, op, batch, s); }
