// Stand-alone timing harness for the 16-bit pointwise forms of pointwise_hs.hip / pointwise_hq.hip (no Python, no library):
// includes the kernel sources, so ablation builds are one -D away:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include [-DPWHS_ABL=1] tools/pwx_bench.hip -o tools/_bin/pwx
//   tools/_bin/pwx [s|q] [variant...]
#include "../yoloret_amd/csrc/pointwise_hs.hip"
#include "../yoloret_amd/csrc/pointwise_hq.hip"
#include <vector>
#include <string>
void yr_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
static const char* last_kernel = "";
void yr_note_kernel(const char* name) { last_kernel = name; }

struct Shape { const char* name; int b, h, k, n; };
int main(int argc, char** argv) {
    const bool q = argc > 1 && argv[1][0] == 'q';
    std::vector<int> variants;
    for (int i = 2; i < argc; ++i) variants.push_back(atoi(argv[i]));
    if (variants.empty()) variants = {0, 1, 2, 3};
    const Shape ex[] = {{"lite0 s5 expand", 128, 26, 112, 672}, {"lite3 s5 expand", 32, 40, 136, 816}, {"lite3 s6 expand", 32, 20, 232, 1392},
                        {"lite0 s6 expand", 128, 13, 192, 1152}, {"bu3_conv", 128, 52, 75, 128}};
    const Shape pr[] = {{"lite0 s5 project", 128, 26, 672, 112}, {"lite3 s5 project", 32, 40, 816, 136}, {"lite3 s6 project", 32, 20, 1392, 232},
                        {"lite0 s6 project", 128, 13, 1152, 192}, {"td1_conv", 128, 13, 288, 512}};
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (const Shape& sh : (q ? pr : ex)) {
        const int M = sh.b * sh.h * sh.h, kp = (sh.k + 7) / 8 * 8, ldo = (sh.n + 7) / 8 * 8;
        std::vector<unsigned short> hx((size_t)M * kp), hw((size_t)sh.n * kp);
        unsigned r = 12345;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return (unsigned short)(0x3c00 + ((r >> 16) & 0x3ff) - ((r >> 27) & 1) * 0x8000 * 0); };   // bf16/f16 bit patterns near 1
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd();
        void *x, *w, *out; float *sc, *shf;
        (void)hipMalloc(&x, hx.size() * 2); (void)hipMalloc(&w, hw.size() * 2); (void)hipMalloc(&out, (size_t)M * ldo * 2);
        (void)hipMalloc(&sc, sh.n * 4 + 64); (void)hipMalloc(&shf, sh.n * 4 + 64);
        (void)hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        std::vector<float> ones(sh.n + 16, 1e-3f), zeros(sh.n + 16, 0.f);
        (void)hipMemcpy(sc, ones.data(), sh.n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(shf, zeros.data(), sh.n * 4, hipMemcpyHostToDevice);
        PwArgs a = {};
        for (int i = 0; i < YR_MAX_SRC; ++i) { a.S.s[i].kbase = 1 << 30; a.S.s[i].ptr = (const float*)x; a.S.s[i].h = sh.h; a.S.s[i].w = sh.h; a.S.s[i].ld = kp; }
        a.S.s[0].c = sh.k; a.S.s[0].kbase = 0; a.S.s[0].xform = YR_X_IDENTITY; a.S.n = 1; a.S.kp = kp;
        a.wt = (const float*)w; a.scale = sc; a.shift = shf; a.out = (float*)out;
        a.M = M; a.H = sh.h; a.W = sh.h; a.N = sh.n; a.out_ld = ldo; a.act = YR_ACT_RELU6;
        const double mb = (double)M * (kp + ldo) * 2 / 1e6;
        printf("%-17s %4.0f MB ", sh.name, mb);
        for (int v : variants) {
            int rc = 0;
            for (int i = 0; i < 3; ++i) rc = q ? yr_pwhq_launch(YR_BF16, v, a, 0) : yr_pwhs_launch(YR_BF16, v, a, 0);
            if (rc != 0) { printf(" v%d n/a", v); continue; }
            (void)hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) (void)(q ? yr_pwhq_launch(YR_BF16, v, a, 0) : yr_pwhs_launch(YR_BF16, v, a, 0));
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  v%d %5.1f us (%.2f TB/s)", v, ms / 20 * 1e3, mb / (ms / 20) / 1e3);
        }
        printf("   [%s]\n", last_kernel);
        (void)hipFree(x); (void)hipFree(w); (void)hipFree(out); (void)hipFree(sc); (void)hipFree(shf);
    }
    return 0;
}
