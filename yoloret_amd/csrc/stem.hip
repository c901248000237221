// Stem: Conv2D 3x3 stride 2, Cin=3, TF 'SAME' padding, + BatchNorm + ReLU6/Swish.
// Replaces Conv2D + FusedBatchNormV3 + Relu6 of MobileNetV2's Conv1 [3P] and the EfficientNet
// stem (reference code/yolo3/efficientnet.py:636-645).
//
// One lane = one output pixel x one cout quad, cout-quad fastest: stores are perfectly
// coalesced float4s; the 27 input taps are shared by the Cout/4 neighbouring lanes through L1.
#include "yr_common.h"

struct StemArgs {
    const float* in;     // [B][Hi][Wi][3] dense
    const float* w;      // [27][ldw]  (tap-major: (ky*3+kx)*3+ci), zero padded to ldw
    const float* scale;  // [ldw]
    const float* shift;  // [ldw]
    float* out;          // [B][Ho][Wo][ld_out]
    int B, Hi, Wi, Ho, Wo, C4, ldw, ld_out, pad_t, pad_l, act;
    long long total;
};

__global__ __launch_bounds__(256) void stem_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [27][ldw]
    for (int i = threadIdx.x; i < 27 * a.ldw; i += 256) wl[i] = a.w[i];
    __syncthreads();
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.total) return;
    const int cq = (int)(gid % a.C4);
    long long t = gid / a.C4;
    const int x = (int)(t % a.Wo);
    t /= a.Wo;
    const int y = (int)(t % a.Ho);
    const int b = (int)(t / a.Ho);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = y * 2 - a.pad_t + ky;
        if (iy < 0 || iy >= a.Hi) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = x * 2 - a.pad_l + kx;
            if (ix < 0 || ix >= a.Wi) continue;
            const float* p = a.in + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * 3;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = p[ci];
                const float4 wv = *reinterpret_cast<const float4*>(wl + ((ky * 3 + kx) * 3 + ci) * a.ldw + cq * 4);
                acc.x = __builtin_fmaf(v, wv.x, acc.x);
                acc.y = __builtin_fmaf(v, wv.y, acc.y);
                acc.z = __builtin_fmaf(v, wv.z, acc.z);
                acc.w = __builtin_fmaf(v, wv.w, acc.w);
            }
        }
    }
    const float4 sc = *reinterpret_cast<const float4*>(a.scale + cq * 4);
    const float4 sh = *reinterpret_cast<const float4*>(a.shift + cq * 4);
    float4 v = make_float4(__builtin_fmaf(acc.x, sc.x, sh.x), __builtin_fmaf(acc.y, sc.y, sh.y),
                           __builtin_fmaf(acc.z, sc.z, sh.z), __builtin_fmaf(acc.w, sc.w, sh.w));
    v = yr_apply_act4(v, a.act);
    *reinterpret_cast<float4*>(a.out + ((size_t)(b * a.Ho + y) * a.Wo + x) * a.ld_out + cq * 4) = v;
}

int yr_launch_stem(const yr_op& op, int batch, hipStream_t s) {
    YR_REQUIRE(op.nsrc == 1 && op.src[0].xform == YR_X_IDENTITY && op.src[0].c == 3 && op.src[0].ld == 3,
               "stem: needs one dense 3-channel source");
    YR_REQUIRE(op.k == 3 && op.stride == 2, "stem: only 3x3 stride 2 is supported");
    const yr_src& in = op.src[0];
    StemArgs a;
    a.in = in.ptr; a.w = op.wgt; a.scale = op.scale; a.shift = op.shift; a.out = op.out;
    YR_REQUIRE(a.in && a.w && a.scale && a.shift && a.out, "stem: null pointer");
    a.B = batch; a.Hi = in.h; a.Wi = in.w; a.Ho = (in.h + 1) / 2; a.Wo = (in.w + 1) / 2;
    YR_REQUIRE(a.Ho == op.h && a.Wo == op.w, "stem: output dims mismatch");
    a.C4 = (op.cout + 3) / 4; a.ldw = a.C4 * 4; a.ld_out = op.out_ld;
    YR_REQUIRE(op.out_ld % 4 == 0 && op.out_ld >= a.ldw, "stem: out_ld must be a multiple of 4 and >= round_up(cout,4)");
    const int pth = (a.Ho - 1) * 2 + 3 - in.h, ptw = (a.Wo - 1) * 2 + 3 - in.w;
    a.pad_t = (pth > 0 ? pth : 0) / 2; a.pad_l = (ptw > 0 ? ptw : 0) / 2;
    a.act = op.act;
    a.total = (long long)batch * a.Ho * a.Wo * a.C4;
    const long long blocks = (a.total + 255) / 256;
    YR_REQUIRE(blocks < (1ll << 31), "stem: grid too large");
    yr_note_kernel("stem_kernel");
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)blocks), dim3(256), 27 * a.ldw * sizeof(float), s, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}
