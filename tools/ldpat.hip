// Micro-benchmark: global_load_dwordx4 throughput for two lane->address maps over a row-major [M][K] fp32 matrix
//   0: coalesced       lane l -> row (l>>2) of a 16-row group, quad (l&3)      (4 adjacent lanes = 64 contiguous bytes)
//   1: MFMA operand    lane l -> row (l&15),                   quad (l>>4)      (what v_mfma_f32_16x16x4 wants in registers)
// Each wave reads 16 rows x K floats, 16 floats (one 64-byte segment per row) per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE, int K>
__global__ __launch_bounds__(256) void rd(const float* __restrict__ x, float* out, int M) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int r = MODE == 0 ? (lane >> 2) : (lane & 15), q = MODE == 0 ? (lane & 3) : (lane >> 4);
    const long row = (long)wave * 16 + r;
    if (row >= M) return;
    const float4* p = reinterpret_cast<const float4*>(x + row * K) + q;
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K / 16; ++k) { float4 v = p[k * 4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}
template <int MODE, int K>
static void run(const float* x, float* out, int M) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = (M / 16 * 64 + 255) / 256;
    for (int i = 0; i < 3; ++i) rd<MODE, K><<<blocks, 256>>>(x, out, M);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) rd<MODE, K><<<blocks, 256>>>(x, out, M);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("mode %d K %3d M %8d: %.4f ms  %.1f GB/s\n", MODE, K, M, ms, (double)M * K * 4 / ms / 1e6);
}
int main() {
    float *x, *out;
    const size_t bytes = (size_t)1 << 30;
    hipMalloc(&x, bytes); hipMalloc(&out, 64); hipMemset(x, 0, bytes);
    // HBM-resident (1 GiB) and MALL/L2-resident (32 MiB) working sets
    run<0, 64>(x, out, 1 << 22); run<1, 64>(x, out, 1 << 22);
    run<0, 288>(x, out, 900000); run<1, 288>(x, out, 900000);
    run<0, 64>(x, out, 1 << 17); run<1, 64>(x, out, 1 << 17);
    run<0, 288>(x, out, 43264); run<1, 288>(x, out, 43264);
    run<0, 48>(x, out, 43264); run<1, 48>(x, out, 43264);
    return 0;
}
