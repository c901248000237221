#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cut_old(v2f x, unsigned& h, unsigned& m) {
    const h2 hh = __builtin_convertvector(x, h2);
    const v2f r = x - __builtin_convertvector(hh, v2f);
    const h2 mm = __builtin_convertvector(r * 2048.0f, h2);
    h = __builtin_bit_cast(unsigned, hh); m = __builtin_bit_cast(unsigned, mm);
}
__device__ __forceinline__ void cut_new(v2f x, unsigned& h, unsigned& m) {
    const h2 hh = __builtin_convertvector(x, h2);
    const v2f xs = x * 2048.0f;
    h = __builtin_bit_cast(unsigned, hh);
    const float c = -2048.0f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(m) : "v"(h), "s"(c), "v"(xs[0]), "v"(xs[1]));
}
// the same with a VALU consumer right behind the partial writes: WITHOUT a wait state the consumer reads a stale register (gfx950's
// dst-sel forwarding hazard; the compiler's hazard recogniser does not look into asm) - nop = 0 shows it, nop = 1 is yr_cut2's form
template <int NOP>
__device__ __forceinline__ void cut_consumed(v2f x, unsigned& h, unsigned& mcopy) {
    const h2 hh = __builtin_convertvector(x, h2);
    const v2f xs = x * 2048.0f;
    h = __builtin_bit_cast(unsigned, hh);
    const float c = -2048.0f;
    unsigned m;
    if (NOP)
        asm("v_fma_mixlo_f16 %0, %2, %3, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %2, %3, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0\n\tv_mov_b32 %1, %0"
            : "=&v"(m), "=&v"(mcopy) : "v"(h), "s"(c), "v"(xs[0]), "v"(xs[1]));
    else
        asm("v_fma_mixlo_f16 %0, %2, %3, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %2, %3, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_mov_b32 %1, %0"
            : "=&v"(m), "=&v"(mcopy) : "v"(h), "s"(c), "v"(xs[0]), "v"(xs[1]));
}
__global__ void k(const v2f* x, unsigned* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h0, m0, h1, m1, h2_, m2, h3, m3;
    cut_old(x[i], h0, m0); cut_new(x[i], h1, m1); cut_consumed<0>(x[i], h2_, m2); cut_consumed<1>(x[i], h3, m3);
    o[6 * i] = h0; o[6 * i + 1] = m0; o[6 * i + 2] = h1; o[6 * i + 3] = m1; o[6 * i + 4] = m2; o[6 * i + 5] = m3;
}
int main() {
    const int n = 1 << 22;
    float* hx = (float*)malloc(n * 8);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) {
        int kind = rand() % 8;
        float u = (float)rand() / RAND_MAX * 2.f - 1.f;
        float v = kind == 0 ? u * 6.f : kind == 1 ? u * 60000.f : kind == 2 ? u * 1e-4f : kind == 3 ? u * 1e-7f : kind == 4 ? ldexpf(u, rand() % 40 - 30) : kind == 5 ? (float)(rand() % 13 - 6) : kind == 6 ? u * 1e-10f : u * 100.f;
        if (i < 8) { const float sp[8] = {0.f, -0.f, 65504.f, -65504.f, 65519.f, 6.1e-5f, 5.96e-8f, 1.f}; v = sp[i]; }
        hx[i] = v;
    }
    v2f* dx; unsigned* dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 24);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    unsigned* ho = (unsigned*)malloc(n * 24);
    hipMemcpy(ho, dout, n * 24, hipMemcpyDeviceToHost);
    long bad = 0, bad_nonop = 0, bad_nop = 0;
    for (int i = 0; i < n; ++i) {
        if (ho[6 * i] != ho[6 * i + 2] || ho[6 * i + 1] != ho[6 * i + 3]) { if (bad < 10) printf("diff at %d: x = %g %g old %08x %08x new %08x %08x\n", i, hx[2 * i], hx[2 * i + 1], ho[6 * i], ho[6 * i + 1], ho[6 * i + 2], ho[6 * i + 3]); ++bad; }
        bad_nonop += ho[6 * i + 4] != ho[6 * i + 1];
        bad_nop += ho[6 * i + 5] != ho[6 * i + 1];
    }
    printf("pairs %d, differing %ld; with a VALU consumer right behind the partial writes: %ld differ without a wait state, %ld with s_nop 0\n", n, bad, bad_nonop, bad_nop);
    return bad != 0 || bad_nop != 0;
}
