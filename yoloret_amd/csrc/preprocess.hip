// Letterbox preprocessing on the GPU: decoded uint8 HWC image -> float32 [H,W,3] network input.
// Replaces, after the host-side JPEG/PNG decode, the TF ops of reference code/yolo.py:105-112
// (tf.io.decode_image(dtype=float32) == uint8 * (1/255)) and code/yolo3/utils.py:67-83 (letterbox_image:
// tf.image.resize bilinear, half-pixel centres, no antialias [3P], then pad_to_bounding_box with zeros).
// float32 arithmetic in TF's operation order, no FMA contraction => bit-identical to oracle/preprocess.py.
#include "yr_common.h"

struct LbArgs {
    const unsigned char* src;  // [ih][iw][3]
    float* dst;                // [H][W][3]
    int ih, iw, H, W, nh, nw, dy, dx;
    float sy, sx;              // ih/nh, iw/nw  (CalculateResizeScale)
};

// grid.y = image index of a batch of equally sized sources (yr_letterbox_batch); 0 for a single image
__global__ __launch_bounds__(256) void letterbox_kernel(LbArgs a) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.H * a.W) return;
    a.src += (size_t)blockIdx.y * a.ih * a.iw * 3;
    a.dst += (size_t)blockIdx.y * a.H * a.W * 3;
    const int y = gid / a.W, x = gid - y * a.W;
    float* o = a.dst + (size_t)gid * 3;
    const int ry = y - a.dy, rx = x - a.dx;
    if (ry < 0 || ry >= a.nh || rx < 0 || rx >= a.nw) { o[0] = o[1] = o[2] = 0.0f; return; }
    const float inv255 = 1.0f / 255.0f;
    const float fy = ((float)ry + 0.5f) * a.sy - 0.5f, fx = ((float)rx + 0.5f) * a.sx - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), y1 = min((int)ceilf(fy), a.ih - 1);
    const int x0 = max((int)flx, 0), x1 = min((int)ceilf(fx), a.iw - 1);
    const float ly = fy - fly, lx = fx - flx;
    const unsigned char* p00 = a.src + ((size_t)y0 * a.iw + x0) * 3;
    const unsigned char* p01 = a.src + ((size_t)y0 * a.iw + x1) * 3;
    const unsigned char* p10 = a.src + ((size_t)y1 * a.iw + x0) * 3;
    const unsigned char* p11 = a.src + ((size_t)y1 * a.iw + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float tl = (float)p00[c] * inv255, tr = (float)p01[c] * inv255;
        const float bl = (float)p10[c] * inv255, br = (float)p11[c] * inv255;
        const float top = tl + (tr - tl) * lx;
        const float bot = bl + (br - bl) * lx;
        o[c] = top + (bot - top) * ly;
    }
}

extern "C" int yr_letterbox_batch(const unsigned char* src_u8, int batch, int ih, int iw, float* dst, int H, int W, void* stream) {
    YR_REQUIRE(src_u8 && dst && batch > 0 && batch < 65536 && ih > 0 && iw > 0 && H > 0 && W > 0, "letterbox: bad arguments");
    LbArgs a;
    a.src = src_u8; a.dst = dst; a.ih = ih; a.iw = iw; a.H = H; a.W = W;
    // utils.py:76-79: nh/nw in float64 then truncated; offsets by floor division
    const double r = ((double)W / iw < (double)H / ih) ? (double)W / iw : (double)H / ih;
    a.nh = (int)((double)ih * r); a.nw = (int)((double)iw * r);
    YR_REQUIRE(a.nh > 0 && a.nw > 0, "letterbox: image collapses to zero size");
    a.dy = (H - a.nh) / 2; a.dx = (W - a.nw) / 2;
    a.sy = (float)ih / (float)a.nh; a.sx = (float)iw / (float)a.nw;
    hipLaunchKernelGGL(letterbox_kernel, dim3((H * W + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

extern "C" int yr_letterbox(const unsigned char* src_u8, int ih, int iw, float* dst, int H, int W, void* stream) {
    return yr_letterbox_batch(src_u8, 1, ih, iw, dst, H, W, stream);
}
