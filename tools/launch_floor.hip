// What a chain of dependent launches costs at batch 1 (DESIGN.md section 6, round 5 "batch 1"): N kernels back to back in one stream, then a
// synchronize - the shape of one latency pass (55 launches).  The kernel is a chain of `depth` dependent global-load round trips (a k
// loop with one chunk in flight) on `wgs` workgroups: depth 0 = the launch floor itself.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void chain(const int* __restrict__ next, float* out, int depth) {
    int i = (blockIdx.x * 256 + threadIdx.x) & 0xffff;
    float s = 0.f;
    for (int d = 0; d < depth; ++d) { i = next[i]; s += (float)i; __syncthreads(); }
    if (s == -1.f) out[0] = s;      // (never: keeps the loads)
    if (threadIdx.x == 0 && depth == 0) out[blockIdx.x] = 1.f;
}

static double wall_us(hipStream_t st, const int* nx, float* out, int n, int wgs, int depth, hipGraphExec_t ge) {
    std::vector<double> ts;
    for (int r = 0; r < 60; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        if (ge) hipGraphLaunch(ge, st);
        else for (int i = 0; i < n; ++i) hipLaunchKernelGGL(chain, dim3(wgs), dim3(256), 0, st, nx, out, depth);
        hipStreamSynchronize(st);
        ts.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const int N = 55;
    int* nx; float* out;
    hipMalloc(&nx, 65536 * 4); hipMalloc(&out, 1 << 20);
    std::vector<int> h(65536);
    for (int i = 0; i < 65536; ++i) h[i] = (i * 40503 + 12345) & 0xffff;
    hipMemcpy(nx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    printf("%d dependent launches + synchronize, p50 wall (us): per launch = wall / %d\n", N, N);
    for (int wgs : {1, 24, 256, 2048})
        for (int depth : {0, 4, 8, 16, 24}) {
            const double w = wall_us(st, nx, out, N, wgs, depth, nullptr);
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(chain, dim3(wgs), dim3(256), 0, st, nx, out, depth);
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            const double wg = wall_us(st, nx, out, N, wgs, depth, ge);
            printf("wgs %5d depth %2d: stream %7.1f us (%5.2f / launch)   graph %7.1f us (%5.2f / launch)\n", wgs, depth, w, w / N, wg, wg / N);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    return 0;
}
