// Issue cost of VALU instruction kinds on gfx950, in cycles per wave64 instruction per SIMD (one..four waves per SIMD):
// v_fma_f32, v_exp_f32, v_rcp_f32, v_pk_fma_f32, v_fmac_f32_dpp, v_cvt_pk_bf16_f32, v_pk_mul_f32 - what a swish / a depthwise
// tap / a rounding really costs in the register-chained block kernels (mbxr_h.hip, mbr.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float w = 0.999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f2 pw = {w, w};
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if constexpr (KIND == 1) asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if constexpr (KIND == 2) asm volatile(REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if constexpr (KIND == 3) asm volatile(REP8("v_fmac_f32_dpp %0, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %2, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %4, %5, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %6, %7, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %4, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %6, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if constexpr (KIND == 4) asm volatile(REP8("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if constexpr (KIND == 5) asm volatile(REP8("v_med3_f32 %0, %0, %8, %1\n v_med3_f32 %1, %1, %8, %2\n v_med3_f32 %2, %2, %8, %3\n v_med3_f32 %3, %3, %8, %4\n v_med3_f32 %4, %4, %8, %5\n v_med3_f32 %5, %5, %8, %6\n v_med3_f32 %6, %6, %8, %7\n v_med3_f32 %7, %7, %8, %0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if constexpr (KIND == 6) asm volatile(REP8("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if constexpr (KIND == 7) asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(pw));
        if constexpr (KIND == 8) asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(pw));
        // 4 transcendentals + 4 plain, interleaved: the sum of their costs, or do they overlap?
        if constexpr (KIND == 9) asm volatile(REP8("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %8\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %8, %8\n v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %8, %8\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %8, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        // 2 transcendentals + 6 plain
        if constexpr (KIND == 10) asm volatile(REP8("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %7, %7, %8, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
    }
    out[blockIdx.x * 256 + threadIdx.x] = p0[0] + p1[1] + p2[0] + p3[1];
    out[blockIdx.x * 256 + threadIdx.x] += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND>
static void run(const char* name, float* out) {
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;   // 4 waves per block = 1 per SIMD per CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // instructions per SIMD = wps waves x iters x 64; clock 2.4 GHz
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * 64);
        printf("%-22s %d wave(s)/SIMD: %.2f cycles per wave instruction (at 2.4 GHz)\n", name, wps, cyc);
    }
}
int main() {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    run<0>("v_fma_f32", out); run<1>("v_exp_f32", out); run<2>("v_rcp_f32", out); run<3>("v_fmac_f32_dpp", out);
    run<4>("v_cvt_pk_bf16_f32", out); run<5>("v_med3_f32", out); run<6>("v_exp_f16", out);
    run<7>("v_pk_mul_f32", out); run<8>("v_pk_fma_f32", out); run<9>("4 exp + 4 fma mixed", out); run<10>("2 exp + 6 fma mixed", out);
    return 0;
}
