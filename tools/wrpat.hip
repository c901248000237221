// Micro-benchmark: how fast can a [M][N] 16-bit map be WRITTEN, by store pattern (no loads, no arithmetic)?
//   0: linear            every wave instruction writes 1 KB contiguous (the memset ceiling)
//   1: 16 rows x 64 B    a wave owns 16 rows and walks the columns 32 at a time: the accumulator-octet layout of the 16-bit
//                        pointwise kernels (lane group g = 16 bytes of a row's 64-byte segment)
//   2: 8 rows x 128 B    the same wave, two column groups per step exchanged between lanes li and li ^ 8: whole 128-byte lines
//   3: pattern 1 through a buffer descriptor (what pointwise_hs.hip issues)
// build: hipcc --offload-arch=gfx950 -O3 tools/wrpat.hip -o tools/_bin/wrpat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void wr(unsigned short* out, int M, int N, int nsplit, unsigned bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    const int mt = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
    const int npairs = N / 32, per = (npairs + nsplit - 1) / nsplit;
    const int pb = sp * per, pe = pb + per < npairs ? pb + per : npairs;
    const u4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    if (MODE == 0) {
        // the workgroup's share of the map as contiguous 4 KB pieces
        const size_t total = (size_t)M * N * 2, share = (total / gridDim.x) & ~(size_t)4095;
        char* p = (char*)out + (size_t)blockIdx.x * share + threadIdx.x * 16;
        for (size_t o = 0; o + 4096 <= share; o += 4096) *(u4*)(p + o) = v;
        return;
    }
    const int m = mt * 64 + wave * 16;
    if (MODE == 1) {
        unsigned short* rp = out + (size_t)(m + li) * N + g * 8;
        for (int j = pb; j < pe; ++j)
            if (m + li < M) *(u4*)(rp + j * 32) = v;
    } else if (MODE == 2) {
        // lanes li < 8: rows m + li, first half of the 128-byte line; li >= 8: row m + li - 8, second half; then rows + 8
        const int r0 = m + (li & 7);
        unsigned short* rp = out + (size_t)r0 * N + (li >> 3) * 32 + g * 8;
        for (int j = pb; j + 1 < pe; j += 2) {
            if (r0 < M) *(u4*)(rp + j * 32) = v;
            if (r0 + 8 < M) *(u4*)(rp + (size_t)8 * N + j * 32) = v;
        }
    } else {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)out), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)out >> 32));
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
        const unsigned base = m + li < M ? ((unsigned)(m + li) * N + g * 8) * 2u : 0xffffffffu;
        for (int j = pb; j < pe; ++j) __builtin_amdgcn_raw_buffer_store_b128(v, r, m + li < M ? base + j * 64 : base, 0, 0);
    }
}
template <int MODE>
static void run(unsigned short* out, int M, int N, int nsplit) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = (M + 63) / 64 * nsplit;
    for (int i = 0; i < 3; ++i) wr<MODE><<<blocks, 256>>>(out, M, N, nsplit, (unsigned)((size_t)M * N * 2));
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) wr<MODE><<<blocks, 256>>>(out, M, N, nsplit, (unsigned)((size_t)M * N * 2));
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("pattern %d  M %7d N %4d nsplit %d (%5d workgroups): %.4f ms  %.0f GB/s\n", MODE, M, N, nsplit, blocks, ms, (double)M * N * 2 / ms / 1e6);
}
int main() {
    unsigned short* out;
    hipMalloc(&out, (size_t)2 << 30);
    const int shapes[][3] = {{86528, 672, 1}, {86528, 672, 2}, {86528, 672, 7}, {51200, 816, 2}, {12800, 1392, 4}, {21632, 1152, 4}, {692224, 672, 1}};
    for (auto& s : shapes) {
        run<0>(out, s[0], s[1], s[2]); run<1>(out, s[0], s[1], s[2]); run<2>(out, s[0], s[1], s[2]); run<3>(out, s[0], s[1], s[2]);
    }
    return 0;
}
