// Micro-benchmark behind DESIGN.md section 6 ("the float32 MFMA and the float32 VALU are the same arithmetic units; the 16-bit matrix pipe
// is not"): the same wave issues NM matrix instructions and NV packed float32 FMAs per loop turn, from registers only.
//   mode 0: v_mfma_f32_16x16x4_f32 alone          mode 1: v_pk_fma_f32 alone          mode 2: both interleaved
//   mode 3: v_mfma_f32_16x16x32_f16 alone         mode 4: the f16 MFMAs + the same v_pk_fma_f32s interleaved
// If two kinds of work share a pipe, time(both) ~ time(a) + time(b); if they do not, time(both) ~ max.
//   hipcc --offload-arch=gfx950 -O3 tools/pipe_overlap.hip -o /tmp/pipe_overlap && /tmp/pipe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int NM = 8, NV = 32;   // matrix instructions / packed FMAs per loop turn (independent chains)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 acc[NM];
    v2f va[NV];
    for (int i = 0; i < NM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < NV; ++i) va[i] = (v2f){0.f, (float)i};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b - i); }
    const v2f x = {a, 0.5f}, y = {1.0f, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (MODE == 0 || MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (MODE == 3 || MODE == 4) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
            if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) va[i * (NV / NM) + j] = __builtin_elementwise_fma(va[i * (NV / NM) + j], x, y);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NV; ++i) s += va[i].x + va[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static float run(float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters / 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {   // waves per SIMD: 256 CUs x wps workgroups of 4 waves
        const int blocks = 256 * wps;
        hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
        const float t0 = run<0>(out, blocks, iters), t1 = run<1>(out, blocks, iters), t2 = run<2>(out, blocks, iters), t3 = run<3>(out, blocks, iters),
                    t4 = run<4>(out, blocks, iters);
        const double waves = (double)blocks * 4;
        const double f_m32 = waves * iters * NM * 2.0 * 16 * 16 * 4, f_pk = waves * iters * NV * 2.0 * 128, f_m16 = waves * iters * NM * 2.0 * 16 * 16 * 32;
        printf("%d wave(s) per SIMD, per loop turn %d matrix instructions + %d v_pk_fma_f32:\n", wps, NM, NV);
        printf("  fp32 MFMA alone          %7.3f ms  %7.1f TFLOP/s\n", t0, f_m32 / (t0 * 1e-3) * 1e-12);
        printf("  v_pk_fma_f32 alone       %7.3f ms  %7.1f TFLOP/s\n", t1, f_pk / (t1 * 1e-3) * 1e-12);
        printf("  fp32 MFMA + pk_fma       %7.3f ms  (sum %7.3f, max %7.3f)\n", t2, t0 + t1, t0 > t1 ? t0 : t1);
        printf("  f16 MFMA 16x16x32 alone  %7.3f ms  %7.1f TFLOP/s\n", t3, f_m16 / (t3 * 1e-3) * 1e-12);
        printf("  f16 MFMA + pk_fma        %7.3f ms  (sum %7.3f, max %7.3f)\n", t4, t3 + t1, t3 > t1 ? t3 : t1);
        hipFree(out);
    }
    return 0;
}
