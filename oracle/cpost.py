"""ctypes binding of oracle/csrc/yr_oracle.c (test infrastructure only).

The C restatement pins exp() to an explicit float32 algorithm so that the HIP
decode kernel can be compared bit-for-bit; ``oracle.postprocess`` (NumPy, libm
exp) is the readable form and agrees with it to ~2 ulp.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ANCHOR_MASK = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])
    return os.path.join(_HERE, '_build', 'libyr_oracle.so')


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, '_build', 'libyr_oracle.so')
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.yro_expf.restype = ctypes.c_float
        L.yro_expf.argtypes = [ctypes.c_float]
        L.yro_sigmoid.restype = ctypes.c_float
        L.yro_sigmoid.argtypes = [ctypes.c_float]
        L.yro_iou.restype = ctypes.c_float
        L.yro_nms.restype = ctypes.c_int
        L.yro_eval_image.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def expf(x):
    L = lib()
    x = np.asarray(x, np.float32)
    return np.array([L.yro_expf(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


# model.py:411-412: the zoom pass is mapped back with these hard-coded constants (python floats -> fp32 tensors)
ZOOM_MUL = np.float32(224 / 416)
ZOOM_ADD = np.float32((416 - 224) / (2 * 416))


def decode_image(yolo_outputs, anchors, num_classes, image_shape, num_scales=3, zoom_outputs=None):
    """One image: list of [G,G,A,C+5] -> boxes [N,4] f32, scores class-major [C,N] f32.
    With zoom_outputs (the zoom-in TTA pass, model.py:408-417) every cell holds 2A boxes and N doubles."""
    L = lib()
    anchors = np.ascontiguousarray(anchors, np.float32)
    mask = ANCHOR_MASK[-num_scales:]
    A = yolo_outputs[0].shape[2]
    N = sum(y.shape[0] * y.shape[1] * A for y in yolo_outputs[:num_scales])
    if zoom_outputs is not None:
        N *= 2
        in_h, in_w = yolo_outputs[0].shape[0] * 32, yolo_outputs[0].shape[1] * 32
        boxes = np.empty((N, 4), np.float32)
        scores = np.empty((num_classes, N), np.float32)
        n0 = 0
        for l in range(num_scales):
            y = np.ascontiguousarray(yolo_outputs[l], np.float32)
            z = np.ascontiguousarray(zoom_outputs[l], np.float32)
            assert z.shape == y.shape
            a = np.ascontiguousarray(anchors[mask[l]])
            L.yro_decode_scale_zoom(_p(y), _p(z), y.shape[0], y.shape[1], A, num_classes, _p(a), in_h, in_w,
                                    int(image_shape[0]), int(image_shape[1]), ctypes.c_float(ZOOM_MUL),
                                    ctypes.c_float(ZOOM_ADD), n0, N, _p(boxes), _p(scores))
            n0 += y.shape[0] * y.shape[1] * A * 2
        return boxes, scores
    in_h, in_w = yolo_outputs[0].shape[0] * 32, yolo_outputs[0].shape[1] * 32
    boxes = np.empty((N, 4), np.float32)
    scores = np.empty((num_classes, N), np.float32)
    n0 = 0
    for l in range(num_scales):
        y = np.ascontiguousarray(yolo_outputs[l], np.float32)
        a = np.ascontiguousarray(anchors[mask[l]])
        L.yro_decode_scale(_p(y), y.shape[0], y.shape[1], A, num_classes, _p(a), in_h, in_w,
                           int(image_shape[0]), int(image_shape[1]), n0, N, _p(boxes), _p(scores))
        n0 += y.shape[0] * y.shape[1] * A
    return boxes, scores


def nms(boxes, scores, max_out, iou_thr, score_thr):
    L = lib()
    boxes = np.ascontiguousarray(boxes, np.float32)
    scores = np.ascontiguousarray(scores, np.float32)
    n = scores.shape[0]
    out = np.empty(max_out, np.int32)
    alive = np.empty(n, np.uint8)
    k = L.yro_nms(_p(boxes), _p(scores), n, int(max_out), ctypes.c_float(iou_thr),
                  ctypes.c_float(score_thr), _p(out), _p(alive))
    return out[:k].copy()


def eval_image(boxes, scores_cm, max_boxes=20, score_threshold=.6, iou_threshold=.5):
    """boxes [N,4], scores class-major [C,N] -> (boxes i32 [K,4], scores [K], classes [K], index [K])."""
    L = lib()
    boxes = np.ascontiguousarray(boxes, np.float32)
    scores_cm = np.ascontiguousarray(scores_cm, np.float32)
    C, N = scores_cm.shape
    ob = np.empty((C * max_boxes, 4), np.int32)
    os_ = np.empty(C * max_boxes, np.float32)
    oc = np.empty(C * max_boxes, np.int32)
    oi = np.empty(C * max_boxes, np.int32)
    alive = np.empty(N, np.uint8)
    tmp = np.empty(max_boxes, np.int32)
    K = L.yro_eval_image(_p(boxes), _p(scores_cm), N, C, int(max_boxes), ctypes.c_float(score_threshold),
                         ctypes.c_float(iou_threshold), _p(ob), _p(os_), _p(oc), _p(oi), _p(alive), _p(tmp))
    return ob[:K].copy(), os_[:K].copy(), oc[:K].copy(), oi[:K].copy()


def yolo_eval(yolo_outputs, anchors, num_scales, num_classes, image_shape, max_boxes=20,
              score_threshold=.6, iou_threshold=.5, zoom_outputs=None):
    b, s = decode_image(yolo_outputs, anchors, num_classes, image_shape, num_scales, zoom_outputs)
    return eval_image(b, s, max_boxes, score_threshold, iou_threshold)
