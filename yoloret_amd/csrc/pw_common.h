// Shared by pointwise.hip (direct kernel + dispatcher) and pointwise_lds.hip (LDS-staged kernel).
#pragma once
#include "yr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at a dword-aligned address

struct PwArgs {
    DSrcSet S;
    const float* wt;      // [N][kp]
    const float* scale;   // [N] or null
    const float* shift;   // [N] or null
    const float* res;     // residual [M][res_ld] or null
    const float* gate;    // SE gate [B][gate_ld] or null
    const float* pre;     // YR_X_UP2_ADD source [B][H/2][W/2][pre_ld] added to the accumulator before BN, or null
    float* out;           // [M][out_ld]
    int M, H, W, N;
    int out_ld, res_ld, gate_ld, pre_ld;
    int act;
    int pool;             // 1: the output is MaxPooling2D(2)(conv output); H, W, M stay the CONV's (pre-pool) dims
};

// GEMM row -> conv pixel.  Plain: identity.  Pooled output: rows are walked in 2x2-quad-major order, so the four
// pixels of a pooling window sit in four ADJACENT lanes of one MFMA tile and the window maximum is two DPP steps.
__device__ __forceinline__ int pw_pixel_of_row(const PwArgs& a, int m) {
    if (!a.pool) return m;
    const int q = m >> 2, sub = m & 3;
    const int wq = a.W >> 1, hwq = (a.H >> 1) * wq;
    const int b = q / hwq, r = q - b * hwq;
    const int yq = r / wq, xq = r - yq * wq;
    return (b * a.H + 2 * yq + (sub >> 1)) * a.W + 2 * xq + (sub & 1);
}

// MaxPooling2D(2) of the finished (BN + activation) values across the four lanes of a pooling window; lane
// (row & 3) == 0 then owns the result, which goes to pooled pixel row >> 2.
__device__ __forceinline__ void pw_pool4(float (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = fmaxf(v[r], __shfl_xor(v[r], 1));
        v[r] = fmaxf(v[r], __shfl_xor(v[r], 2));
    }
}

// The pre-BatchNorm addend of output pixel m, couts n..n+3 (zeros without one): nearest 2x upsampling of `pre`.
__device__ __forceinline__ void pw_pre_addend(const PwArgs& a, int m, int n, float (&p)[4]) {
    p[0] = p[1] = p[2] = p[3] = 0.f;
    if (a.pre == nullptr) return;
    const int hw = a.H * a.W;
    const int b = m / hw, rem = m - b * hw;
    const int y = rem / a.W, x = rem - y * a.W;
    const float* src = a.pre + ((size_t)(b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1)) * a.pre_ld + n;
    if (n + 3 < a.N && (a.pre_ld & 3) == 0) {  // n is a multiple of 4: one 16-byte load
        const float4 v = *reinterpret_cast<const float4*>(src);
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (n + r < a.N) p[r] = src[r];
}

// LDS-staged kernel, tile shape index 0..13: (BM x BN) = 256x16, 128x32, 128x48, 128x64, 128x80, 128x96, 128x128,
// 64x16, 64x32, 64x48, 64x64, 64x80, 64x96, 64x128
int yr_pw_launch_lds(int shape, const PwArgs& a, hipStream_t s);
