// Micro-benchmark: sustained fp32 rates of v_mfma_f32_16x16x4_f32, v_mfma_f32_32x32x2_f32 and v_pk_fma_f32
// from registers only (no memory), to know the real ceilings the conv kernels are measured against.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_pkfma(float* out, int iters) {
    v2f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v2f){0.f, (float)i};
    v2f a = {threadIdx.x * 1e-3f, 0.5f}, b = {1.0f, 1.0f + threadIdx.x * 1e-4f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_elementwise_fma(a, b, acc[i]);
        asm volatile("" : "+v"(a));
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the lane kernels' form: one operand a wave-uniform SGPR pair (weights), the other a VGPR broadcast to both halves
template <int NACC>
__global__ __launch_bounds__(256) void k_pkfma_sgpr(float* out, int iters, float w0, float w1) {
    v2f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v2f){0.f, (float)i};
    float x = threadIdx.x * 1e-3f;
    v2f b = {w0, w1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_elementwise_fma((v2f){x, x}, b, acc[i]);
        asm volatile("" : "+v"(x));
        asm volatile("" : "+s"(b));
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipEventRecord(a); f(); f(); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 3;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8192 * 4);
    const int iters = 4000;
    for (int blocks : {256, 512, 1024, 2048}) {
        float ms = timeit([&] { k_mfma16<8><<<blocks, 256>>>(out, iters); });
        printf("mfma16x16x4 f32  blocks %4d (%d waves/SIMD): %.1f TF\n", blocks, blocks / 256, 2.0 * 1024 * 8 * iters * blocks * 4 / ms / 1e9);
        ms = timeit([&] { k_mfma32<4><<<blocks, 256>>>(out, iters); });
        printf("mfma32x32x2 f32  blocks %4d: %.1f TF\n", blocks, 2.0 * 2048 * 4 * iters * blocks * 4 / ms / 1e9);
        ms = timeit([&] { k_pkfma<16><<<blocks, 256>>>(out, iters); });
        printf("v_pk_fma_f32     blocks %4d: %.1f TF\n", blocks, 2.0 * 128 * 16 * iters * blocks * 4 / ms / 1e9);
        ms = timeit([&] { k_pkfma_sgpr<16><<<blocks, 256>>>(out, iters, 1.0f, 1.0001f); });
        printf("v_pk_fma_f32 (SGPR pair x broadcast VGPR) blocks %4d: %.1f TF\n", blocks, 2.0 * 128 * 16 * iters * blocks * 4 / ms / 1e9);
        ms = timeit([&] { k_pkfma_sgpr<2><<<blocks, 256>>>(out, iters * 8, 1.0f, 1.0001f); });
        printf("v_pk_fma_f32 (SGPR form, 2 chains as in mblane) blocks %4d: %.1f TF\n", blocks, 2.0 * 128 * 2 * iters * 8 * blocks * 4 / ms / 1e9);
    }
    return 0;
}
