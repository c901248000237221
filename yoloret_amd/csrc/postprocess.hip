// Decode + per-class NMS + detection packing.  Replaces the TF kernels behind reference
// code/yolo3/model.py: yolo_head :344-371, yolo_correct_boxes :374-399,
// yolo_boxes_and_scores :402-428, and the per-class tf.image.non_max_suppression /
// tf.gather / tf.cast loop of yolo_eval :474-490.
//
// Arithmetic is float32 in the reference's operation order with no FMA contraction
// (library built with -ffp-contract=off) and exp() pinned to yr_expf, so that the results
// are bit-identical to the C oracle (oracle/csrc/yr_oracle.c) on identical logits.
#include "yr_common.h"
#include <cstdlib>

// ------------------------------------------------------------------ letterbox inverse terms
struct Letterbox {
    float input_h, input_w, image_h, image_w, off_h, off_w, scale_h, scale_w, mul_h, mul_w;
};

// model.py:379-387 (all float32, same operation order)
__device__ __forceinline__ Letterbox yr_letterbox(int in_h, int in_w, int img_h, int img_w) {
    Letterbox L;
    L.input_h = (float)in_h; L.input_w = (float)in_w;
    L.image_h = (float)img_h; L.image_w = (float)img_w;
    const float max_shape = fmaxf(L.image_h, L.image_w);
    const float ratio_h = L.image_h / max_shape, ratio_w = L.image_w / max_shape;
    const float boxed_h = L.input_h * ratio_h, boxed_w = L.input_w * ratio_w;
    L.off_h = (L.input_h - boxed_h) / 2.0f; L.off_w = (L.input_w - boxed_w) / 2.0f;
    L.scale_h = L.image_h / boxed_h; L.scale_w = L.image_w / boxed_w;
    L.mul_h = L.input_h * L.scale_h; L.mul_w = L.input_w * L.scale_w;
    return L;
}

__device__ __forceinline__ float yr_clip(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// model.py:386-398
__device__ __forceinline__ float4 yr_correct_box(const Letterbox& L, float bx, float by, float bw, float bh) {
    const float cy = (by * L.input_h - L.off_h) * L.scale_h;
    const float cx = (bx * L.input_w - L.off_w) * L.scale_w;
    const float hh = bh * L.mul_h, ww = bw * L.mul_w;
    return make_float4(yr_clip(cy - hh / 2.0f, 0.0f, L.image_h), yr_clip(cx - ww / 2.0f, 0.0f, L.image_w),
                       yr_clip(cy + hh / 2.0f, 0.0f, L.image_h), yr_clip(cx + ww / 2.0f, 0.0f, L.image_w));
}

// ------------------------------------------------------------------ fused decode (3 scales)
struct DecodeArgs {
    const float* y[3];
    const float* z[3];      // zoom-in TTA pass (model.py:408-417) or null
    int passes;             // 1, or 2 with the zoom pass: a cell then holds 2A boxes (concat on the anchor axis)
    float zoom_mul, zoom_add;
    int gh[3], gw[3];
    int nstart[3];      // first box index of each scale
    int tile_start[4];  // first block of each scale along grid.x
    float anchors[3][8][2];
    int num_scales, A, C, N, in_h, in_w;
    const int32_t* image_hw;
    float* boxes;   // [B][N][4]
    float* scores;  // [B][C][N]
};

#define DEC_TILE 256
// grid (tiles, B); block 256: one lane per box; the tile's logits are contiguous in HBM and
// staged through LDS with coalesced loads (row stride C+5 floats; odd strides are conflict-free).
__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a) {
    extern __shared__ float sm[];
    const int b = blockIdx.y;
    int s = 0;
    if (a.num_scales > 1 && (int)blockIdx.x >= a.tile_start[1]) s = 1;
    if (a.num_scales > 2 && (int)blockIdx.x >= a.tile_start[2]) s = 2;
    const int gh = a.gh[s], gw = a.gw[s];
    const int ns = gh * gw * a.A;
    const int t0 = ((int)blockIdx.x - a.tile_start[s]) * DEC_TILE;
    const int cnt = min(DEC_TILE, ns - t0);
    const int row = a.C + 5;
    const int pass = (int)blockIdx.z;  // 0: the plain logits, 1: the zoom pass
    const float* src = (pass ? a.z[s] : a.y[s]) + ((size_t)b * ns + t0) * row;
    // batches of DEC_LB loads before their LDS stores (a store after each load would serialise the HBM round trips; 32 = the whole
    // tile of a 20-class model in ONE round trip - round 5: with batches of 8 a workgroup went to HBM four times in a row), every
    // load unconditional at a clamped index (a predicated load gets its own branch and wait)
    constexpr int DEC_LB = 32;
    const int last = cnt * row - 1;
    for (int i0 = 0; i0 <= last; i0 += 256 * DEC_LB) {
        float v[DEC_LB];
#pragma unroll
        for (int u = 0; u < DEC_LB; ++u) v[u] = src[min(i0 + u * 256 + (int)threadIdx.x, last)];
#pragma unroll
        for (int u = 0; u < DEC_LB; ++u) {
            const int i = i0 + u * 256 + threadIdx.x;
            if (i <= last) sm[i] = v[u];
        }
    }
    __syncthreads();
    const int j = threadIdx.x;
    if (j >= cnt) return;
    const float* t = sm + j * row;
    const int n = t0 + j;
    const int an = n % a.A;
    const int cell = n / a.A;
    const int w = cell % gw, h = cell / gw;
    const Letterbox L = yr_letterbox(a.in_h, a.in_w, a.image_hw[b * 2], a.image_hw[b * 2 + 1]);
    // model.py:363-367
    float bx = (yr_sigmoid(t[0]) + (float)w) / (float)gw;
    float by = (yr_sigmoid(t[1]) + (float)h) / (float)gh;
    float bw = yr_expf(t[2]) * a.anchors[s][an][0] / L.input_w;
    float bh = yr_expf(t[3]) * a.anchors[s][an][1] / L.input_h;
    if (pass) {  // model.py:411-412 (multiply, then add: two roundings, no contraction)
        bx = bx * a.zoom_mul + a.zoom_add; by = by * a.zoom_mul + a.zoom_add;
        bw = bw * a.zoom_mul; bh = bh * a.zoom_mul;
    }
    const float conf = yr_sigmoid(t[4]);
    const int gn = a.passes == 1 ? a.nstart[s] + n : (a.nstart[s] + cell * a.A) * 2 + pass * a.A + an;
    *reinterpret_cast<float4*>(a.boxes + ((size_t)b * a.N + gn) * 4) = yr_correct_box(L, bx, by, bw, bh);
    float* sp = a.scores + (size_t)b * a.C * a.N + gn;
    for (int c = 0; c < a.C; ++c) sp[(size_t)c * a.N] = conf * yr_sigmoid(t[5 + c]);  // model.py:426
}

static int decode_launch(const float* y1, const float* y2, const float* y3, const float* const* zoom, float zoom_mul,
                         float zoom_add, int batch, int in_h, int in_w, int num_anchors, int num_classes, int num_scales,
                         const float* anchors_host, const int32_t* image_hw, float* boxes, float* scores, void* stream) {
    YR_REQUIRE(num_scales >= 1 && num_scales <= 3, "decode: num_scales must be 1..3");
    YR_REQUIRE(num_anchors >= 1 && num_anchors <= 8, "decode: num_anchors must be 1..8");
    YR_REQUIRE(in_h % 32 == 0 && in_w % 32 == 0 && in_h > 0 && in_w > 0, "decode: input size must be a multiple of 32");
    YR_REQUIRE(y1 && anchors_host && image_hw && boxes && scores && batch > 0, "decode: null pointer / empty batch");
    DecodeArgs a;
    const float* ys[3] = {y1, y2, y3};
    const int total_anchors = 3 * num_anchors;
    int n = 0, tiles = 0;
    for (int s = 0; s < 3; ++s) {
        a.y[s] = ys[s < num_scales ? s : 0];
        a.z[s] = zoom ? zoom[s < num_scales ? s : 0] : nullptr;
        if (zoom && s < num_scales) YR_REQUIRE(zoom[s] != nullptr, "decode: zoom logits %d are null", s + 1);
        const int stride = 32 >> s;
        a.gh[s] = in_h / stride; a.gw[s] = in_w / stride;
        a.nstart[s] = n; a.tile_start[s] = tiles;
        if (s < num_scales) {
            YR_REQUIRE(ys[s] != nullptr, "decode: y%d is null", s + 1);
            // anchor_mask = [[6,7,8],[3,4,5],[0,1,2]][-num_scales:] (model.py:444-445)
            const int mask_row = 3 - num_scales + s;           // row of the full mask table
            const int first = total_anchors - (mask_row + 1) * num_anchors;
            for (int k = 0; k < num_anchors; ++k) {
                a.anchors[s][k][0] = anchors_host[(first + k) * 2];
                a.anchors[s][k][1] = anchors_host[(first + k) * 2 + 1];
            }
            const int ns = a.gh[s] * a.gw[s] * num_anchors;
            n += ns; tiles += (ns + DEC_TILE - 1) / DEC_TILE;
        }
    }
    a.tile_start[3] = tiles;
    a.passes = zoom ? 2 : 1; a.zoom_mul = zoom_mul; a.zoom_add = zoom_add;
    a.num_scales = num_scales; a.A = num_anchors; a.C = num_classes; a.N = n * a.passes; a.in_h = in_h; a.in_w = in_w;
    a.image_hw = image_hw; a.boxes = boxes; a.scores = scores;
    const size_t lds = (size_t)DEC_TILE * (num_classes + 5) * sizeof(float);
    YR_REQUIRE(lds <= 160 * 1024, "decode: too many classes for the LDS tile");
    hipLaunchKernelGGL(decode_kernel, dim3(tiles, batch, a.passes), dim3(256), lds, (hipStream_t)stream, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

extern "C" int yr_decode(const float* y1, const float* y2, const float* y3, int batch, int in_h, int in_w,
                         int num_anchors, int num_classes, int num_scales, const float* anchors_host,
                         const int32_t* image_hw, float* boxes, float* scores, void* stream) {
    return decode_launch(y1, y2, y3, nullptr, 1.f, 0.f, batch, in_h, in_w, num_anchors, num_classes, num_scales,
                         anchors_host, image_hw, boxes, scores, stream);
}

extern "C" int yr_decode_zoom(const float* y1, const float* y2, const float* y3, const float* z1, const float* z2,
                              const float* z3, float zoom_mul, float zoom_add, int batch, int in_h, int in_w,
                              int num_anchors, int num_classes, int num_scales, const float* anchors_host,
                              const int32_t* image_hw, float* boxes, float* scores, void* stream) {
    const float* zoom[3] = {z1, z2, z3};
    YR_REQUIRE(z1 != nullptr, "decode: zoom logits are null");
    return decode_launch(y1, y2, y3, zoom, zoom_mul, zoom_add, batch, in_h, in_w, num_anchors, num_classes, num_scales,
                         anchors_host, image_hw, boxes, scores, stream);
}

// ------------------------------------------------------------------ yolo_head / yolo_correct_boxes (reference layouts)
struct HeadArgs {
    const float* feats;
    float anchors[8][2];
    int gh, gw, A, C, in_h, in_w;
    long long total;  // B*gh*gw*A
    float *xy, *wh, *conf, *probs, *scores;
};

__global__ __launch_bounds__(256) void yolo_head_kernel(HeadArgs a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.total) return;
    const int an = (int)(gid % a.A);
    const long long cell = gid / a.A;
    const int w = (int)(cell % a.gw);
    const int h = (int)((cell / a.gw) % a.gh);
    const float* t = a.feats + (size_t)gid * (a.C + 5);
    a.xy[gid * 2] = (yr_sigmoid(t[0]) + (float)w) / (float)a.gw;
    a.xy[gid * 2 + 1] = (yr_sigmoid(t[1]) + (float)h) / (float)a.gh;
    a.wh[gid * 2] = yr_expf(t[2]) * a.anchors[an][0] / (float)a.in_w;
    a.wh[gid * 2 + 1] = yr_expf(t[3]) * a.anchors[an][1] / (float)a.in_h;
    const float conf = yr_sigmoid(t[4]);
    a.conf[gid] = conf;
    for (int c = 0; c < a.C; ++c) {
        const float p = yr_sigmoid(t[5 + c]);
        a.probs[(size_t)gid * a.C + c] = p;
        if (a.scores) a.scores[(size_t)gid * a.C + c] = conf * p;  // model.py:426
    }
}

extern "C" int yr_yolo_head(const float* feats, int batch, int gh, int gw, int num_anchors, int num_classes,
                            const float* anchors_host, int in_h, int in_w, float* box_xy, float* box_wh,
                            float* conf, float* probs, float* scores, void* stream) {
    YR_REQUIRE(feats && anchors_host && box_xy && box_wh && conf && probs, "yolo_head: null pointer");
    YR_REQUIRE(num_anchors >= 1 && num_anchors <= 8 && batch > 0 && gh > 0 && gw > 0, "yolo_head: bad sizes");
    HeadArgs a;
    a.feats = feats; a.gh = gh; a.gw = gw; a.A = num_anchors; a.C = num_classes; a.in_h = in_h; a.in_w = in_w;
    for (int k = 0; k < num_anchors; ++k) { a.anchors[k][0] = anchors_host[k * 2]; a.anchors[k][1] = anchors_host[k * 2 + 1]; }
    a.total = (long long)batch * gh * gw * num_anchors;
    a.xy = box_xy; a.wh = box_wh; a.conf = conf; a.probs = probs; a.scores = scores;
    hipLaunchKernelGGL(yolo_head_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

__global__ __launch_bounds__(256) void correct_boxes_kernel(const float* xy, const float* wh, long long n_per_image,
                                                            long long total, int in_h, int in_w,
                                                            const int32_t* image_hw, float* boxes) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int b = (int)(gid / n_per_image);
    const Letterbox L = yr_letterbox(in_h, in_w, image_hw[b * 2], image_hw[b * 2 + 1]);
    *reinterpret_cast<float4*>(boxes + gid * 4) = yr_correct_box(L, xy[gid * 2], xy[gid * 2 + 1], wh[gid * 2], wh[gid * 2 + 1]);
}

extern "C" int yr_correct_boxes(const float* box_xy, const float* box_wh, int batch, int64_t n_per_image, int in_h,
                                int in_w, const int32_t* image_hw, float* boxes, void* stream) {
    YR_REQUIRE(box_xy && box_wh && image_hw && boxes && batch > 0 && n_per_image > 0, "correct_boxes: bad arguments");
    const long long total = (long long)batch * n_per_image;
    hipLaunchKernelGGL(correct_boxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       box_xy, box_wh, (long long)n_per_image, total, in_h, in_w, image_hw, boxes);
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// ------------------------------------------------------------------ NMS
// NonMaxSuppression's IOU() [3P], float32, same operation order as the oracle.
__device__ __forceinline__ float yr_iou(float4 bi, float4 bj) {
    const float ymin_i = fminf(bi.x, bi.z), xmin_i = fminf(bi.y, bi.w);
    const float ymax_i = fmaxf(bi.x, bi.z), xmax_i = fmaxf(bi.y, bi.w);
    const float ymin_j = fminf(bj.x, bj.z), xmin_j = fminf(bj.y, bj.w);
    const float ymax_j = fmaxf(bj.x, bj.z), xmax_j = fmaxf(bj.y, bj.w);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    const float iy = fmaxf(fminf(ymax_i, ymax_j) - fmaxf(ymin_i, ymin_j), 0.0f);
    const float ix = fmaxf(fminf(xmax_i, xmax_j) - fmaxf(xmin_i, xmin_j), 0.0f);
    const float inter = iy * ix;
    return inter / (area_i + area_j - inter);
}

struct NmsArgs {
    const float* boxes;   // [B][N][4]
    const float* scores;  // [B][C][N]
    int N, C, max_boxes;
    float score_thr, iou_thr;
    int32_t* out_idx;     // [B][C][max_boxes]
    int32_t* out_count;   // [B][C]
};

#define NMS_DEAD (-__builtin_inff())
// grid (C, B); one workgroup per (image, class).  key[i] = score if still a candidate else -inf,
// resident in LDS.  Each round: arg-max by (score desc, index asc) with wave shuffles, select,
// then every lane suppresses its candidates whose IoU with the pick exceeds the threshold -
// the same set and order as TF's pop-and-test loop.
__global__ __launch_bounds__(256) void nms_kernel(NmsArgs a) {
    extern __shared__ float key[];
    __shared__ float ws[4];
    __shared__ int wi[4];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sc = a.scores + ((size_t)b * a.C + c) * a.N;
    const float4* bx = reinterpret_cast<const float4*>(a.boxes) + (size_t)b * a.N;
    for (int i = tid; i < a.N; i += 256) {
        const float s = sc[i];
        key[i] = (s > a.score_thr) ? s : NMS_DEAD;
    }
    __syncthreads();
    int32_t* oi = a.out_idx + ((size_t)b * a.C + c) * a.max_boxes;
    int picked = 0;
    for (; picked < a.max_boxes; ++picked) {
        float bs = NMS_DEAD;
        int bi = 0x7fffffff;
        for (int i = tid; i < a.N; i += 256) {
            const float s = key[i];
            if (s > bs) { bs = s; bi = i; }  // ascending scan: first (smallest-index) max is kept
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float s2 = __shfl_xor(bs, o);
            const int i2 = __shfl_xor(bi, o);
            if (s2 > bs || (s2 == bs && i2 < bi)) { bs = s2; bi = i2; }
        }
        if (lane == 0) { ws[wave] = bs; wi[wave] = bi; }
        __syncthreads();
        bs = ws[0]; bi = wi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (ws[w] > bs || (ws[w] == bs && wi[w] < bi)) { bs = ws[w]; bi = wi[w]; }
        if (!(bs > NMS_DEAD)) break;  // uniform: no candidate left
        if (tid == 0) { oi[picked] = bi; key[bi] = NMS_DEAD; }
        const float4 pb = bx[bi];
        __syncthreads();
        for (int i = tid; i < a.N; i += 256) {
            if (key[i] > NMS_DEAD) {
                if (yr_iou(pb, bx[i]) > a.iou_thr) key[i] = NMS_DEAD;
            }
        }
        __syncthreads();
    }
    if (tid == 0) a.out_count[(size_t)b * a.C + c] = picked;
    for (int i = picked + tid; i < a.max_boxes; i += 256) oi[i] = -1;
}

// Lazy greedy NMS - the order TF's kernel itself works in (N < 65536).  One workgroup of T lanes per
// (image, class):
//   1. candidates (score > thr) are compacted with wave ballots into an LDS list of `cap` entries
//      (score + uint16 index); lane t owns entries t, t+T, ... and insertion-sorts them once by
//      (score desc, index asc), so popping its best is O(1) (advance a head pointer, prefetch that box);
//   2. one step = pop the block-wide best - its box travels through the shuffle reduction, so there is no
//      dependent global load - and test it ONLY against the boxes selected so far (every wave redundantly,
//      lanes = selected boxes, __any): select or drop.  On typical data that is ~max_boxes steps of O(1)
//      work instead of max_boxes sweeps over thousands of candidates.
// The rounds are latency-bound, so throughput comes from the number of problems resident per CU: the first
// launch uses T=256 and cap=NMS_CAP1 (5200 entries = 31.2 KB LDS, 5 problems/CU).  A problem with MORE candidates
// keeps only its highest-scoring ones: a 2048-bin histogram of the scores (linear in [thr, 1]) gives the lowest bin B*
// whose suffix holds <= cap candidates, and the list is refilled with the candidates of bins >= B*.  That is exact:
// greedy NMS pops candidates in descending score order, and the list holds EVERY candidate scoring at least as high as
// any candidate in it, so the pops - and with them the picks - are those of the full list for as long as the list
// lasts.  Only if it runs dry before max_boxes picks (or one bin alone overflows the cap) is the problem flagged
// (count = -1) and redone by a second launch with T=1024 and cap=N (only flagged problems do any work).  With random
// weights an EfficientNet head puts > 5200 of the 10647 boxes of every class above the 0.2 threshold: the second
// launch used to do all the work there (2.4 ms per 128-image batch; now 0.3 ms).
#ifndef NMS_CAP1
#define NMS_CAP1 5200  // first-pass list capacity: the largest that still fits 5 problems per CU (measured on the bench
                       // data: 6144 -> 4/CU 0.157 ms, 5200 -> 5/CU 0.114 ms, 4096 -> 0.234 ms because overflows take the second pass)
#endif
struct NmsLazyArgs {
    NmsArgs a;
    int cap;            // list capacity (entries)
    int only_overflow;  // second pass: handle only problems flagged -1
};

template <int T>
__global__ __launch_bounds__(T) void nms_lazy_kernel(NmsLazyArgs L) {
    const NmsArgs& a = L.a;
    extern __shared__ float4 nms_lds[];
    float4* selb = nms_lds;                                                   // [max_boxes] selected boxes
    float* ks = reinterpret_cast<float*>(selb + a.max_boxes);                 // [cap] scores
    unsigned short* ki = reinterpret_cast<unsigned short*>(ks + L.cap);      // [cap] box indices
    __shared__ float ws[T / 64];
    __shared__ int wi[T / 64];
    __shared__ float4 wb[T / 64];
    __shared__ int cnt;
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (L.only_overflow && a.out_count[(size_t)b * a.C + c] != -1) return;  // solved by the first pass
    const float* sc = a.scores + ((size_t)b * a.C + c) * a.N;
    const float4* bx = reinterpret_cast<const float4*>(a.boxes) + (size_t)b * a.N;
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int i0 = 0; i0 < a.N; i0 += T) {
        const int i = i0 + tid;
        const float s = i < a.N ? sc[i] : 0.f;
        const bool keep = i < a.N && s > a.score_thr;
        const unsigned long long mask = __ballot(keep);
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(&cnt, __popcll(mask));
        base = __shfl(base, 0);
        if (keep) {
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < L.cap) { ks[pos] = s; ki[pos] = (unsigned short)i; }
        }
    }
    __syncthreads();
    int total = cnt;
    int32_t* oi = a.out_idx + ((size_t)b * a.C + c) * a.max_boxes;
    bool truncated = false;
    if (total > L.cap) {  // uniform: keep the highest-scoring <= cap candidates (see above)
        constexpr int NB = 2048;
        // the list area is free again: histogram, per-thread partial sums and the result live there (no static LDS -
        // one more KB per workgroup would cost the fifth resident problem per CU)
        unsigned* hist = reinterpret_cast<unsigned*>(ks);
        unsigned* part = hist + NB;
        int* bstar_p = reinterpret_cast<int*>(part + T);
        const float scale = (float)NB / (1.0f - a.score_thr);
        auto bin_of = [&](float v) { const int q = (int)((v - a.score_thr) * scale); return q < 0 ? 0 : (q > NB - 1 ? NB - 1 : q); };
        if ((size_t)L.cap * 6 < (size_t)(NB + T + 1) * 4 || !(a.score_thr < 1.0f)) {
            if (tid == 0) a.out_count[(size_t)b * a.C + c] = -1;
            return;
        }
        for (int i = tid; i < NB; i += T) hist[i] = 0u;
        __syncthreads();
        for (int i = tid; i < a.N; i += T) {
            const float v = sc[i];
            if (v > a.score_thr) atomicAdd(&hist[bin_of(v)], 1u);
        }
        __syncthreads();
        constexpr int BPT = NB / T > 0 ? NB / T : 1;          // bins per thread (descending: thread 0 owns the top bins)
        unsigned mine = 0;
        if (tid * BPT < NB)
            for (int j = 0; j < BPT; ++j) mine += hist[NB - 1 - (tid * BPT + j)];
        part[tid] = mine;
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0;
            int bs_ = NB;                                     // bins >= bs_ are kept
            for (int t2 = 0; t2 < T && t2 * BPT < NB; ++t2) {
                if (cum + part[t2] <= (unsigned)L.cap) { cum += part[t2]; bs_ = NB - (t2 + 1) * BPT; continue; }
                for (int j = 0; j < BPT; ++j) {               // the thread block that crosses the cap: bin by bin
                    const unsigned hcount = hist[NB - 1 - (t2 * BPT + j)];
                    if (cum + hcount > (unsigned)L.cap) break;
                    cum += hcount;
                    bs_ = NB - 1 - (t2 * BPT + j);
                }
                break;
            }
            *bstar_p = bs_;
            cnt = 0;
        }
        __syncthreads();
        const int bst = *bstar_p;
        if (bst >= NB) {  // the top bin alone overflows the list: the second launch takes the whole problem
            if (tid == 0) a.out_count[(size_t)b * a.C + c] = -1;
            return;
        }
        __syncthreads();  // every reader of hist is done before the list is rebuilt over it
        for (int i0 = 0; i0 < a.N; i0 += T) {
            const int i = i0 + tid;
            const float v = i < a.N ? sc[i] : 0.f;
            const bool keep = i < a.N && v > a.score_thr && bin_of(v) >= bst;
            const unsigned long long mask = __ballot(keep);
            int base = 0;
            if (lane == 0 && mask) base = atomicAdd(&cnt, __popcll(mask));
            base = __shfl(base, 0);
            if (keep) {
                const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                ks[pos] = v; ki[pos] = (unsigned short)i;
            }
        }
        __syncthreads();
        total = cnt;
        truncated = true;
    }
    const int n = total > tid ? (total - tid + T - 1) / T : 0;
    for (int j = 1; j < n; ++j) {
        const float s = ks[tid + T * j];
        const unsigned short i = ki[tid + T * j];
        int k = j - 1;
        while (k >= 0) {
            const float sk = ks[tid + T * k];
            const unsigned short ik = ki[tid + T * k];
            if (sk > s || (sk == s && ik < i)) break;
            ks[tid + T * (k + 1)] = sk;
            ki[tid + T * (k + 1)] = ik;
            --k;
        }
        ks[tid + T * (k + 1)] = s;
        ki[tid + T * (k + 1)] = i;
    }
    int head = 0;
    float ls; int li; float4 lb;
    auto load_head = [&]() {
        if (head < n) { ls = ks[tid + T * head]; li = ki[tid + T * head]; lb = bx[li]; }
        else { ls = NMS_DEAD; li = 0x7fffffff; lb = make_float4(0.f, 0.f, 0.f, 0.f); }
    };
    load_head();
    int nsel = 0;
    while (nsel < a.max_boxes) {
        float bs = ls;
        int bi = li;
        float4 pb = lb;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float s2 = __shfl_xor(bs, o);
            const int i2 = __shfl_xor(bi, o);
            const float q0 = __shfl_xor(pb.x, o), q1 = __shfl_xor(pb.y, o), q2 = __shfl_xor(pb.z, o), q3 = __shfl_xor(pb.w, o);
            if (s2 > bs || (s2 == bs && i2 < bi)) { bs = s2; bi = i2; pb = make_float4(q0, q1, q2, q3); }
        }
        if (lane == 0) { ws[wave] = bs; wi[wave] = bi; wb[wave] = pb; }
        __syncthreads();
        int bw = 0;
        bs = ws[0]; bi = wi[0];
#pragma unroll
        for (int w = 1; w < T / 64; ++w)
            if (ws[w] > bs || (ws[w] == bs && wi[w] < bi)) { bs = ws[w]; bi = wi[w]; bw = w; }
        pb = wb[bw];
        if (!(bs > NMS_DEAD)) {        // uniform: nothing left
            if (truncated) {           // ... of a truncated list: lower-scoring candidates may still be picked - redo in full
                if (tid == 0) a.out_count[(size_t)b * a.C + c] = -1;
                return;
            }
            break;
        }
        // the owner advances to its next best entry (that box's load overlaps the test below)
        if (li == bi) {
            ++head;
            load_head();
        }
        // suppressed by an already selected box?  (lanes stride the selected list; same answer in every wave)
        bool sup = false;
        for (int j = lane; j < nsel; j += 64) sup = sup || (yr_iou(pb, selb[j]) > a.iou_thr);
        const bool suppressed = __any(sup);
        __syncthreads();  // everyone is done with ws/wi/wb and selb before they change
        if (!suppressed) {
            if (tid == 0) { oi[nsel] = bi; selb[nsel] = pb; }
            ++nsel;
            __syncthreads();  // selb[nsel-1] visible
        }
    }
    if (tid == 0) a.out_count[(size_t)b * a.C + c] = nsel;
    for (int i = nsel + tid; i < a.max_boxes; i += T) oi[i] = -1;
}

// Band-wise lazy greedy NMS: the first launch of yr_nms.  Greedy NMS pops candidates in descending (score, then
// ascending index) order and tests each only against the boxes selected so far, so only the top of the order ever
// matters - typically a few more entries than max_boxes.  One workgroup of T lanes per (image, class):
//   1. every lane reads its <= SR scores ONCE into registers; candidates (score > thr) go into a 2048-bin histogram
//      of the scores (linear in [thr, 1], monotone in the score) in LDS;
//   2. a BAND is the widest run of bins below the previous band that holds <= 64 (then 128, ... T) candidates (found
//      from a block-wide scan of the histogram).  Its candidates are compacted from the registers (one per lane), every lane fetches
//      its box and computes the rank of its entry among the band's by (score desc, index asc) - T broadcast reads,
//      no sort, no barrier - and scatters (box, index) to that rank;
//   3. wave 0 walks the ranked band: entry r is selected unless its IoU with an already selected box exceeds the
//      threshold (lanes = selected boxes, __any) - a single wave, no workgroup barrier per step;
//   4. bands repeat, downwards, until max_boxes are selected or the candidates run out.
// Exact: a higher bin means a strictly higher score, so band after band in rank order IS the global descending order,
// ties included (equal scores share a bin).  The kernel this replaces as the first pass (nms_lazy_kernel<256>, still
// the second pass) built and insertion-sorted a list of up to 5200 candidates and found every pick by a block-wide
// arg-max with three barriers: ~100 us per problem, latency-bound; this one is ~10 us.  A problem whose next bin alone
// holds more than a band's capacity (a mass of equal scores) is flagged (count = -1) for the second pass.
template <int T, int SR>
#ifndef NMS_BAND_WPE
#define NMS_BAND_WPE 5
#endif
__global__ __launch_bounds__(T, (T <= 256 ? NMS_BAND_WPE : 4)) void nms_band_kernel(NmsArgs a) {   // (second argument: waves per SIMD)
    constexpr int NB = 2048, BPT = NB / T;   // bins per lane; lane t owns the descending bins NB-1-(t*BPT+j)
    static_assert(NB % T == 0, "bins divide over the lanes");
    extern __shared__ float4 nms_lds[];
    float4* selb = nms_lds;                  // [max_boxes] selected boxes
    __shared__ unsigned hist[NB];
    __shared__ int cum[T];                   // candidates in the bins of lanes 0..t (inclusive)
    __shared__ __attribute__((aligned(16))) float ks[T];   // the band: scores, box indices (compacted, unordered)
    __shared__ __attribute__((aligned(16))) int ki[T];
    __shared__ float4 sbox[T];               // ... ranked: boxes, box indices
    __shared__ int sidx[T];
    __shared__ int salive[T];                // ... not suppressed by a box selected in an earlier band
    __shared__ int wtot[T / 64];
    __shared__ int cnt, lo_s, nsel_s;
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sc = a.scores + ((size_t)b * a.C + c) * a.N;
    const float4* bx = reinterpret_cast<const float4*>(a.boxes) + (size_t)b * a.N;
    int32_t* oi = a.out_idx + ((size_t)b * a.C + c) * a.max_boxes;
    const float thr = a.score_thr;
    const float scale = (float)NB / (1.0f - thr);
    auto bin_of = [&](float v) { const int q = (int)((v - thr) * scale); return q < 0 ? 0 : (q > NB - 1 ? NB - 1 : q); };

    float sreg[SR];
#pragma unroll
    for (int u = 0; u < SR; ++u) {
        const int i = u * T + tid;
        const float v = sc[i < a.N ? i : a.N - 1];   // (clamped, unconditional: all SR loads in flight at once - written
        sreg[u] = i < a.N ? v : thr;                 //  as a branch per load they ran one HBM round trip after the other)
                                                     // thr itself is not a candidate
    }
    for (int i = tid; i < NB; i += T) hist[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SR; ++u) {
        if (sreg[u] > thr) atomicAdd(&hist[bin_of(sreg[u])], 1u);
        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (keeps the unrolled bodies from piling up their temporaries)
    }
    __syncthreads();
    // inclusive scan over the lanes of the per-lane bin sums (descending bins)
    int mine = 0;
#pragma unroll
    for (int j = 0; j < BPT; ++j) mine += (int)hist[NB - 1 - (tid * BPT + j)];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wtot[w];
    incl += before;
    cum[tid] = incl;
    __syncthreads();
    const int total = cum[T - 1];

    int hi = NB, base = 0, nsel = 0;         // bins >= hi are done; they held `base` candidates
    // band capacity: 64 entries first (max_boxes picks rarely need more than a few dozen pops, and ranking costs
    // capacity^2 LDS reads), doubling per band up to one entry per lane
    for (int bcap = 64 < T ? 64 : T; base < total; bcap = 2 * bcap < T ? 2 * bcap : T) {   // (workgroup-uniform)
        if (tid == 0) { lo_s = NB; cnt = 0; }
        __syncthreads();
        {   // the lowest bin lo < hi with (candidates in bins [lo, hi)) <= bcap
            int s = tid ? cum[tid - 1] : 0, mylo = NB;
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                const int bin = NB - 1 - (tid * BPT + j);
                s += (int)hist[bin];
                if (bin < hi && s - base <= bcap) mylo = bin;
            }
            if (mylo < NB) atomicMin(&lo_s, mylo);
        }
        __syncthreads();
        const int lo = lo_s;
        if (lo >= NB) {                      // the next bin alone overflows THIS band capacity
            if (bcap < T) continue;          // (workgroup-uniform) a wider band may still hold it: 65..T equal or near-equal scores
            if (tid == 0) a.out_count[(size_t)b * a.C + c] = -1;   // ... more than a band can ever hold: the second pass redoes the problem
            return;
        }
        {   // compact the band from the registers: per-lane count, wave scan, ONE LDS atomic per wave, then the writes
            static_assert(SR <= 64, "one mask bit per score register");
            unsigned long long km = 0ull;
#pragma unroll
            for (int u = 0; u < SR; ++u) {
                const float v = sreg[u];
                bool keep = v > thr;
                if (keep) { const int q = bin_of(v); keep = q >= lo && q < hi; }
                km |= (unsigned long long)(keep ? 1 : 0) << u;
                if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            const int mycnt = __popcll(km);
            int incl2 = mycnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl2, o);
                if (lane >= o) incl2 += v;
            }
            int wbase = 0;
            if (lane == 63 && incl2 > 0) wbase = atomicAdd(&cnt, incl2);
            wbase = __shfl(wbase, 63);
            int pos = wbase + incl2 - mycnt;
#pragma unroll
            for (int u = 0; u < SR; ++u)
                if ((km >> u) & 1ull) { ks[pos] = sreg[u]; ki[pos] = u * T + tid; ++pos; }
        }
        __syncthreads();
        const int n = cnt;                   // 1 <= n <= T
        if (tid < n) {
            const float s = ks[tid];
            const int i = ki[tid];
            const float4 box = bx[i];        // (in flight while the rank is counted)
            int rank = 0;
            for (int j = 0; j < n; j += 4) { // LDS broadcast reads, four entries at a time (entries >= n are stale: masked)
                const float4 s4 = *reinterpret_cast<const float4*>(ks + j);
                const int4 i4 = *reinterpret_cast<const int4*>(ki + j);
                rank += (s4.x > s || (s4.x == s && i4.x < i)) ? 1 : 0;
                rank += (j + 1 < n && (s4.y > s || (s4.y == s && i4.y < i))) ? 1 : 0;
                rank += (j + 2 < n && (s4.z > s || (s4.z == s && i4.z < i))) ? 1 : 0;
                rank += (j + 3 < n && (s4.w > s || (s4.w == s && i4.w < i))) ? 1 : 0;
            }
            sbox[rank] = box; sidx[rank] = i;
            // every entry against the boxes selected in EARLIER bands, all lanes at once (the entry's box is in registers; selb[]
            // by broadcast reads): what is left for the walk below is the band's own order
            bool dead = false;
            for (int j = 0; j < nsel; ++j) dead = dead || (yr_iou(box, selb[j]) > a.iou_thr);
            salive[rank] = dead ? 0 : 1;
        }
        __syncthreads();
        if (wave == 0) {
            // The walk of the ranked band (round 5).  Greedy NMS selects entry r iff no box selected before it suppresses it.  The
            // kernel used to test one entry per step against the selected boxes (lanes = selected boxes): n steps per band, and on
            // data where most candidates are suppressed (random weights: thousands of overlapping boxes above the threshold) band
            // after band of 64 .. T steps for a handful of picks - 167 us per 2560 problems on the SE EfficientNet-B0 head.  Now a
            // lane OWNS the entries lane, lane + 64, ..: a step = the lowest-ranked entry still alive (ballots, a few scalar
            // operations) is selected, and every lane drops those of its later entries the pick suppresses - steps = picks, <= max_boxes
            // over all bands.  Same picks in the same order: an entry dies by exactly the boxes selected before it.
            constexpr int NS = T / 64;
            bool al[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) al[k] = lane + 64 * k < n && salive[lane + 64 * k] != 0;
            while (nsel < a.max_boxes) {
                int r = -1;
#pragma unroll
                for (int k = NS - 1; k >= 0; --k) {
                    const unsigned long long m = __ballot(al[k]);
                    if (m) r = 64 * k + __builtin_ctzll(m);     // (wave-uniform; the lowest slot with a live entry wins)
                }
                if (r < 0) break;
                const float4 pb = sbox[r];
                if (lane == 0) { oi[nsel] = sidx[r]; selb[nsel] = pb; }
                ++nsel;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int e = lane + 64 * k;
                    if (al[k]) al[k] = e > r && !(yr_iou(sbox[e], pb) > a.iou_thr);   // (argument order as in the test against selb[] above: candidate, selected)
                }
            }
            if (lane == 0) nsel_s = nsel;
        }
        __syncthreads();
        nsel = nsel_s;
        if (nsel >= a.max_boxes) break;
        base += n;
        hi = lo;
    }
    if (tid == 0) a.out_count[(size_t)b * a.C + c] = nsel;
    for (int i = nsel + tid; i < a.max_boxes; i += T) oi[i] = -1;
}

extern "C" int yr_nms(const float* boxes, const float* scores, int batch, int n, int num_classes, int max_boxes,
                      float score_thr, float iou_thr, int32_t* out_idx, int32_t* out_count, void* stream) {
    YR_REQUIRE(boxes && scores && out_idx && out_count, "nms: null pointer");
    YR_REQUIRE(batch > 0 && n > 0 && num_classes > 0 && max_boxes > 0, "nms: bad sizes");
    YR_REQUIRE(((uintptr_t)boxes % 16) == 0, "nms: boxes must be 16-byte aligned");
    const size_t lds_limit = 150 * 1024;
    // the > 64 KB dynamic-LDS opt-in is a per-device function attribute: remember it per device
    static bool attr_set_dev[64] = {false};
    int cur_dev = 0;
    YR_CHECK_HIP(hipGetDevice(&cur_dev));
    bool& attr_set = attr_set_dev[cur_dev & 63];
    if (!attr_set) {
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)nms_lazy_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        YR_CHECK_HIP(hipFuncSetAttribute((const void*)nms_lazy_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
        attr_set = true;
    }
    NmsLazyArgs L;
    NmsArgs& a = L.a;
    a.boxes = boxes; a.scores = scores; a.N = n; a.C = num_classes; a.max_boxes = max_boxes;
    a.score_thr = score_thr; a.iou_thr = iou_thr; a.out_idx = out_idx; a.out_count = out_count;
    const size_t sel_bytes = (size_t)max_boxes * sizeof(float4);
    const size_t lds_full = sel_bytes + (size_t)n * 6 + 16;
    hipStream_t st = (hipStream_t)stream;
    if (n < 65536 && lds_full <= lds_limit) {
        // pass 1: band-wise lists for every problem (scores in registers: the lane count follows n); pass 2: full-capacity
        // sorted lists, only where pass 1 met a bin with more candidates than a band holds
        static const bool band_on = [] { const char* e = getenv("YOLORET_NMS_BAND"); return !(e && e[0] == '0'); }();   // (A/B switch)
        const bool band_ok = band_on && score_thr < 1.0f && score_thr > -1e30f;   // a finite linear histogram range [thr, 1]
        if (band_ok && n <= 256 * 44) {
            hipLaunchKernelGGL((nms_band_kernel<256, 44>), dim3(num_classes, batch), dim3(256), sel_bytes, st, a);
        } else if (band_ok && n <= 512 * 50) {
            hipLaunchKernelGGL((nms_band_kernel<512, 50>), dim3(num_classes, batch), dim3(512), sel_bytes, st, a);
        } else if (band_ok) {
            hipLaunchKernelGGL((nms_band_kernel<1024, 38>), dim3(num_classes, batch), dim3(1024), sel_bytes, st, a);   // n <= 38400
        } else {
            L.cap = n < NMS_CAP1 ? n : NMS_CAP1;
            L.only_overflow = 0;
            hipLaunchKernelGGL(nms_lazy_kernel<256>, dim3(num_classes, batch), dim3(256), sel_bytes + (size_t)L.cap * 6 + 16, st, L);
        }
        if (band_ok || L.cap < n) {
            L.cap = n;
            L.only_overflow = 1;
            hipLaunchKernelGGL(nms_lazy_kernel<1024>, dim3(num_classes, batch), dim3(1024), lds_full, st, L);
        }
    } else {
        const size_t lds = (size_t)n * sizeof(float);
        YR_REQUIRE(lds <= lds_limit, "nms: %d boxes per image exceed the LDS-resident limit (38400)", n);
        hipLaunchKernelGGL(nms_kernel, dim3(num_classes, batch), dim3(256), lds, st, a);
    }
    YR_LAUNCH_CHECK();
    return YR_OK;
}

// ------------------------------------------------------------------ pack detections
struct PackArgs {
    const float* boxes; const float* scores; const int32_t* idx; const int32_t* cnt;
    int N, C, max_boxes;
    int32_t* det; int32_t* det_count;
};

// one block per image: exclusive prefix over the class counts, then gather + truncate-cast
// (model.py:481-490: class-ascending, pick order inside a class).
__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
    extern __shared__ int pre[];  // [C+1]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int s = 0;
        for (int c = 0; c < a.C; ++c) { pre[c] = s; s += a.cnt[(size_t)b * a.C + c]; }
        pre[a.C] = s;
        a.det_count[b] = s;
    }
    __syncthreads();
    const int K = pre[a.C];
    const int slots = a.C * a.max_boxes;
    int32_t* d = a.det + (size_t)b * slots * 6;
    for (int t = tid; t < slots; t += 256) {
        const int c = t / a.max_boxes, j = t - c * a.max_boxes;
        if (j < a.cnt[(size_t)b * a.C + c]) {
            const int i = a.idx[((size_t)b * a.C + c) * a.max_boxes + j];
            const float4 bx = reinterpret_cast<const float4*>(a.boxes)[(size_t)b * a.N + i];
            int32_t* r = d + (size_t)(pre[c] + j) * 6;
            r[0] = (int32_t)bx.x; r[1] = (int32_t)bx.y; r[2] = (int32_t)bx.z; r[3] = (int32_t)bx.w;  // tf.cast truncates
            r[4] = __float_as_int(a.scores[((size_t)b * a.C + c) * a.N + i]);
            r[5] = c;
        }
    }
    for (int t = K + tid; t < slots; t += 256) {
        int32_t* r = d + (size_t)t * 6;
        r[0] = r[1] = r[2] = r[3] = 0; r[4] = 0; r[5] = -1;
    }
}

extern "C" int yr_pack_detections(const float* boxes, const float* scores, const int32_t* nms_idx,
                                  const int32_t* nms_count, int batch, int n, int num_classes, int max_boxes,
                                  int32_t* det, int32_t* det_count, void* stream) {
    YR_REQUIRE(boxes && scores && nms_idx && nms_count && det && det_count, "pack: null pointer");
    YR_REQUIRE(batch > 0 && n > 0 && num_classes > 0 && max_boxes > 0, "pack: bad sizes");
    PackArgs a;
    a.boxes = boxes; a.scores = scores; a.idx = nms_idx; a.cnt = nms_count; a.N = n; a.C = num_classes;
    a.max_boxes = max_boxes; a.det = det; a.det_count = det_count;
    hipLaunchKernelGGL(pack_kernel, dim3(batch), dim3(256), (num_classes + 1) * sizeof(int), (hipStream_t)stream, a);
    YR_LAUNCH_CHECK();
    return YR_OK;
}
