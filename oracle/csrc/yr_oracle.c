/* C restatement of the reference's decode + per-class NMS.  TEST INFRASTRUCTURE
 * ONLY (see oracle/__init__.py) - the product never links or loads this.
 *
 * Follows /root/reference/code/yolo3/model.py: yolo_head :344-371,
 * yolo_correct_boxes :374-399, yolo_boxes_and_scores :402-428, yolo_eval
 * :431-491, and TensorFlow's NonMaxSuppressionV3 (third-party).
 *
 * exp() is pinned to an explicit float32 algorithm (yro_expf: Cody-Waite
 * reduction + degree-5 polynomial, every step an IEEE mul/add/fma/rint) so
 * that this file and the HIP decode kernel, which implements the same steps,
 * agree BIT FOR BIT; against libm/NumPy exp it is within 2 ulp.  TF's own
 * exp/sigmoid are Eigen polynomial approximations of the same class (C.3).
 * Build with -ffp-contract=off (oracle/Makefile): no a*b+c contraction other
 * than the explicit fmaf calls.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

float yro_expf(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    int32_t e = ((int32_t)n + 127) << 23; /* 2^n, n in [-126,127] */
    float s;
    memcpy(&s, &e, 4);
    return y * s;
}

float yro_sigmoid(float x) { return 1.0f / (1.0f + yro_expf(-x)); }

static inline float clipf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

/* One scale of one image.  feats [G_h][G_w][A][C+5]; anchors [A][2] (w,h).
 * Writes boxes[(n0+n)*4 .. ] = (ymin,xmin,ymax,xmax) and class-major scores
 * scores[c*N + n0 + n], n = (h*G_w+w)*A+a. */
void yro_decode_scale(const float* feats, int gh, int gw, int A, int C, const float* anchors,
                      int in_h, int in_w, int img_h, int img_w, int n0, int N,
                      float* boxes, float* scores) {
    const float input_h = (float)in_h, input_w = (float)in_w;
    const float image_h = (float)img_h, image_w = (float)img_w;
    const float max_shape = fmaxf(image_h, image_w);
    const float ratio_h = image_h / max_shape, ratio_w = image_w / max_shape;
    const float boxed_h = input_h * ratio_h, boxed_w = input_w * ratio_w;
    const float off_h = (input_h - boxed_h) / 2.0f, off_w = (input_w - boxed_w) / 2.0f;
    const float scale_h = image_h / boxed_h, scale_w = image_w / boxed_w;
    const float hw_mul_h = input_h * scale_h, hw_mul_w = input_w * scale_w;
    for (int h = 0; h < gh; ++h)
        for (int w = 0; w < gw; ++w)
            for (int a = 0; a < A; ++a) {
                const int n = (h * gw + w) * A + a;
                const float* t = feats + (size_t)n * (C + 5);
                float bx = (yro_sigmoid(t[0]) + (float)w) / (float)gw;
                float by = (yro_sigmoid(t[1]) + (float)h) / (float)gh;
                float bw = yro_expf(t[2]) * anchors[a * 2 + 0] / input_w;
                float bh = yro_expf(t[3]) * anchors[a * 2 + 1] / input_h;
                float conf = yro_sigmoid(t[4]);
                float cy = (by * input_h - off_h) * scale_h;
                float cx = (bx * input_w - off_w) * scale_w;
                float hh = bh * hw_mul_h, ww = bw * hw_mul_w;
                float* o = boxes + (size_t)(n0 + n) * 4;
                o[0] = clipf(cy - hh / 2.0f, 0.0f, image_h);
                o[1] = clipf(cx - ww / 2.0f, 0.0f, image_w);
                o[2] = clipf(cy + hh / 2.0f, 0.0f, image_h);
                o[3] = clipf(cx + ww / 2.0f, 0.0f, image_w);
                for (int c = 0; c < C; ++c)
                    scores[(size_t)c * N + n0 + n] = conf * yro_sigmoid(t[5 + c]);
            }
}

/* The zoom-in test-time-augmentation branch of yolo_boxes_and_scores (model.py:408-417): a second set of
 * logits (the network run on a centre crop) is decoded by yolo_head with the same anchors, its boxes are mapped
 * back by box_xy_z*zoom_mul + zoom_add and box_wh_z*zoom_mul (model.py:411-412; the reference hard-codes
 * zoom_mul = 224/416, zoom_add = (416-224)/(2*416)), and xy / wh / confidence / class probabilities are
 * concatenated on the ANCHOR axis (tf.concat(..., -2), :414-417) before yolo_correct_boxes.  So a cell holds
 * 2A boxes: n = (h*G_w+w)*2A + a for the plain pass and + A for the zoom pass; N counts both. */
void yro_decode_scale_zoom(const float* feats, const float* zoom_feats, int gh, int gw, int A, int C,
                           const float* anchors, int in_h, int in_w, int img_h, int img_w,
                           float zoom_mul, float zoom_add, int n0, int N, float* boxes, float* scores) {
    const float input_h = (float)in_h, input_w = (float)in_w;
    const float image_h = (float)img_h, image_w = (float)img_w;
    const float max_shape = fmaxf(image_h, image_w);
    const float ratio_h = image_h / max_shape, ratio_w = image_w / max_shape;
    const float boxed_h = input_h * ratio_h, boxed_w = input_w * ratio_w;
    const float off_h = (input_h - boxed_h) / 2.0f, off_w = (input_w - boxed_w) / 2.0f;
    const float scale_h = image_h / boxed_h, scale_w = image_w / boxed_w;
    const float hw_mul_h = input_h * scale_h, hw_mul_w = input_w * scale_w;
    for (int h = 0; h < gh; ++h)
        for (int w = 0; w < gw; ++w)
            for (int pass = 0; pass < 2; ++pass)
                for (int a = 0; a < A; ++a) {
                    const int cell = h * gw + w;
                    const float* t = (pass ? zoom_feats : feats) + (size_t)(cell * A + a) * (C + 5);
                    float bx = (yro_sigmoid(t[0]) + (float)w) / (float)gw;
                    float by = (yro_sigmoid(t[1]) + (float)h) / (float)gh;
                    float bw = yro_expf(t[2]) * anchors[a * 2 + 0] / input_w;
                    float bh = yro_expf(t[3]) * anchors[a * 2 + 1] / input_h;
                    if (pass) {
                        bx = bx * zoom_mul + zoom_add; by = by * zoom_mul + zoom_add;
                        bw = bw * zoom_mul; bh = bh * zoom_mul;
                    }
                    float conf = yro_sigmoid(t[4]);
                    float cy = (by * input_h - off_h) * scale_h;
                    float cx = (bx * input_w - off_w) * scale_w;
                    float hh = bh * hw_mul_h, ww = bw * hw_mul_w;
                    const int n = cell * 2 * A + pass * A + a;
                    float* o = boxes + (size_t)(n0 + n) * 4;
                    o[0] = clipf(cy - hh / 2.0f, 0.0f, image_h);
                    o[1] = clipf(cx - ww / 2.0f, 0.0f, image_w);
                    o[2] = clipf(cy + hh / 2.0f, 0.0f, image_h);
                    o[3] = clipf(cx + ww / 2.0f, 0.0f, image_w);
                    for (int c = 0; c < C; ++c)
                        scores[(size_t)c * N + n0 + n] = conf * yro_sigmoid(t[5 + c]);
                }
}

/* NonMaxSuppression's IOU() [3P]. */
float yro_iou(const float* bi, const float* bj) {
    const float ymin_i = fminf(bi[0], bi[2]), xmin_i = fminf(bi[1], bi[3]);
    const float ymax_i = fmaxf(bi[0], bi[2]), xmax_i = fmaxf(bi[1], bi[3]);
    const float ymin_j = fminf(bj[0], bj[2]), xmin_j = fminf(bj[1], bj[3]);
    const float ymax_j = fmaxf(bj[0], bj[2]), xmax_j = fmaxf(bj[1], bj[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0 || area_j <= 0) return 0.0f;
    const float iy = fmaxf(fminf(ymax_i, ymax_j) - fmaxf(ymin_i, ymin_j), 0.0f);
    const float ix = fmaxf(fminf(xmax_i, xmax_j) - fmaxf(xmin_i, xmin_j), 0.0f);
    const float inter = iy * ix;
    return inter / (area_i + area_j - inter);
}

/* Hard NMS, TF semantics: candidates score>thr (strict); best = (score desc,
 * index asc); a candidate is dropped iff IoU>iou_thr (strict) with a selected
 * box.  `alive` is caller scratch of N bytes.  Returns the number selected. */
int yro_nms(const float* boxes, const float* scores, int N, int max_out, float iou_thr,
            float score_thr, int32_t* out_idx, unsigned char* alive) {
    for (int i = 0; i < N; ++i) alive[i] = scores[i] > score_thr;
    int k = 0;
    while (k < max_out) {
        int best = -1;
        float bs = 0.0f;
        for (int i = 0; i < N; ++i)
            if (alive[i] && (best < 0 || scores[i] > bs)) { best = i; bs = scores[i]; }
        if (best < 0) break;
        out_idx[k++] = best;
        alive[best] = 0;
        for (int i = 0; i < N; ++i)
            if (alive[i] && yro_iou(boxes + (size_t)best * 4, boxes + (size_t)i * 4) > iou_thr) alive[i] = 0;
    }
    return k;
}

/* yolo_eval for one image given decoded boxes/scores: out records are
 * class-ascending then pick order (model.py:474-490).  Returns K. */
int yro_eval_image(const float* boxes, const float* scores, int N, int C, int max_boxes,
                   float score_thr, float iou_thr, int32_t* out_boxes /*[C*max,4]*/,
                   float* out_scores, int32_t* out_classes, int32_t* out_index,
                   unsigned char* alive, int32_t* idx_tmp) {
    int K = 0;
    for (int c = 0; c < C; ++c) {
        int k = yro_nms(boxes, scores + (size_t)c * N, N, max_boxes, iou_thr, score_thr, idx_tmp, alive);
        for (int j = 0; j < k; ++j, ++K) {
            const float* b = boxes + (size_t)idx_tmp[j] * 4;
            for (int q = 0; q < 4; ++q) out_boxes[K * 4 + q] = (int32_t)b[q]; /* tf.cast truncates */
            out_scores[K] = scores[(size_t)c * N + idx_tmp[j]];
            out_classes[K] = c;
            out_index[K] = idx_tmp[j];
        }
    }
    return K;
}
