// Does v_fmac_f32_dpp ... row_newbcast:n broadcast lane n of every 16-lane row on gfx950?  (prints "ok" / the first mismatch)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
    const int l = threadIdx.x;
    float a = 0.f, e = 1.0f + l, t = 100.f * (l >> 4) + (l & 15);
    asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" : "+v"(a) : "v"(t), "v"(e));
    o[l] = a;
}
int main() {
    float* d; hipMalloc(&d, 256); float h[64];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        const float want = (100.f * (l >> 4) + 5) * (1.0f + l);
        if (h[l] != want) { printf("lane %d: got %g want %g\n", l, h[l], want); return 1; }
    }
    printf("row_newbcast ok\n");
    return 0;
}
