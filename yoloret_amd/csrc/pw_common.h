// Shared by pointwise.hip (direct kernel + dispatcher) and pointwise_lds.hip (LDS-staged kernel).
#pragma once
#include "yr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PwArgs {
    DSrcSet S;
    const float* wt;      // [N][kp]
    const float* scale;   // [N] or null
    const float* shift;   // [N] or null
    const float* res;     // residual [M][res_ld] or null
    const float* gate;    // SE gate [B][gate_ld] or null
    float* out;           // [M][out_ld]
    int M, H, W, N;
    int out_ld, res_ld, gate_ld;
    int act;
};

// LDS-staged kernel, tile shape index 0..13: (BM x BN) = 256x16, 128x32, 128x48, 128x64, 128x80, 128x96, 128x128,
// 64x16, 64x32, 64x48, 64x64, 64x80, 64x96, 64x128
int yr_pw_launch_lds(int shape, const PwArgs& a, hipStream_t s);
