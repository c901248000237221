"""ctypes binding of libyoloret_hip.so (include/yoloret_hip.h).

PyTorch-ROCm tensors are only the containers (device memory + streams); every
arithmetic step happens in the HIP kernels behind the C-ABI.  There is no CPU
fallback: if the library is missing, loading fails loudly.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('YOLORET_LIB') or os.path.join(_HERE, 'libyoloret_hip.so')     # (YOLORET_LIB: another build of the same ABI, e.g. for same-box A/B runs)

YR_MAX_SRC = 4
ACT = {'none': 0, None: 0, 'relu6': 1, 'swish': 2, 'sigmoid': 3, 'leaky': 4}
XFORM = {'identity': 0, 'up2': 1, 'maxpool2': 2, 'maxpool4': 3, 'up2_add': 4, 'dw3': 5}
OP_STEM, OP_POINTWISE, OP_DEPTHWISE, OP_SE_MEAN, OP_SE_FC, OP_WSUM, OP_GATHER, OP_MBCONV = 1, 2, 3, 4, 5, 6, 7, 8
OP_STEMBLOCK, OP_MBLANE, OP_MBH, OP_MBX, OP_MBR, OP_MBE, OP_HEAD = 9, 10, 11, 12, 13, 14, 15
# yr_dtype: element type of activation tensors / pointwise weights (include/yoloret_hip.h)
DTYPE = {'f32': 0, 'float32': 0, None: 0, 'bf16': 1, 'bfloat16': 1, 'f16': 2, 'float16': 2, 'u8': 3, 'uint8': 3}   # (u8: images only)
DTYPE_NAME = {0: 'f32', 1: 'bf16', 2: 'f16', 3: 'u8'}
ESIZE = {0: 4, 1: 2, 2: 2, 3: 1}
VEC = {0: 4, 1: 8, 2: 8}          # channels per 16 bytes: granule of `ld` and of the pointwise k-space
TORCH_DTYPE = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16, 3: torch.uint8}


def dtype_id(d):
    """'f32' | 'bf16' | 'f16' | 'float32' | 'bfloat16' | 'float16' | torch dtype | yr_dtype int -> yr_dtype int."""
    if isinstance(d, int):
        if d in DTYPE_NAME:
            return d
    elif isinstance(d, torch.dtype):
        for k, v in TORCH_DTYPE.items():
            if v == d:
                return k
    elif d in DTYPE:
        return DTYPE[d]
    raise ValueError('unsupported dtype %r (float32, bfloat16, float16)' % (d,))


def to_bits16(a, dtype):
    """float32 array -> uint16 bit patterns of the 16-bit type, round to nearest even (what the kernels' stores do)."""
    a = np.ascontiguousarray(a, np.float32)
    if dtype_id(dtype) == 2:
        return a.astype(np.float16).view(np.uint16)
    u = a.view(np.uint32).astype(np.uint64)
    return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) & 0xffff).astype(np.uint16)


def from_bits16(b, dtype):
    b = np.ascontiguousarray(b, np.uint16)
    if dtype_id(dtype) == 2:
        return b.view(np.float16).astype(np.float32)
    return (b.astype(np.uint32) << 16).view(np.float32)
OP_NAMES = {1: 'stem', 2: 'pointwise', 3: 'depthwise', 4: 'se_mean', 5: 'se_fc', 6: 'wsum', 7: 'gather', 8: 'mbconv',
            9: 'stemblock', 10: 'mblane', 11: 'mbh', 12: 'mbx', 13: 'mbr', 14: 'mbe', 15: 'head'}


class YrSrc(ctypes.Structure):
    _fields_ = [('ptr', ctypes.c_void_p), ('buf', ctypes.c_int32), ('h', ctypes.c_int32), ('w', ctypes.c_int32),
                ('c', ctypes.c_int32), ('ld', ctypes.c_int32), ('xform', ctypes.c_int32), ('dtype', ctypes.c_int32)]


class YrOp(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('act', ctypes.c_int32), ('h', ctypes.c_int32), ('w', ctypes.c_int32),
                ('cin', ctypes.c_int32), ('cout', ctypes.c_int32), ('k', ctypes.c_int32), ('stride', ctypes.c_int32),
                ('nsrc', ctypes.c_int32), ('se_reduced', ctypes.c_int32),
                ('dtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32),
                ('src', YrSrc * YR_MAX_SRC),
                ('out', ctypes.c_void_p), ('out_buf', ctypes.c_int32), ('out_ld', ctypes.c_int32),
                ('res', ctypes.c_void_p), ('res_buf', ctypes.c_int32), ('res_ld', ctypes.c_int32),
                ('gate', ctypes.c_void_p), ('gate_buf', ctypes.c_int32), ('gate_ld', ctypes.c_int32),
                ('wgt', ctypes.c_void_p), ('wgt_off', ctypes.c_int64),
                ('scale', ctypes.c_void_p), ('scale_off', ctypes.c_int64),
                ('shift', ctypes.c_void_p), ('shift_off', ctypes.c_int64),
                ('wgt2', ctypes.c_void_p), ('wgt2_off', ctypes.c_int64),
                ('b1', ctypes.c_void_p), ('b1_off', ctypes.c_int64),
                ('b2', ctypes.c_void_p), ('b2_off', ctypes.c_int64),
                # ABI 7: the SE tail (squeeze-excite finished by the op that produces the map)
                ('gate_out', ctypes.c_void_p), ('gate_out_buf', ctypes.c_int32), ('gate_out_ld', ctypes.c_int32),
                ('se_hidden', ctypes.c_int32), ('reserved0', ctypes.c_int32),
                ('se_w', ctypes.c_void_p), ('se_w_off', ctypes.c_int64),
                ('sync', ctypes.c_void_p)]


class YrBuf(ctypes.Structure):
    _fields_ = [('bytes_per_image', ctypes.c_int64), ('arena_off_per_image', ctypes.c_int64),
                ('external_slot', ctypes.c_int32), ('dtype', ctypes.c_int32)]


ABI_VERSION = 9   # == YR_ABI_VERSION of include/yoloret_hip.h
EXPORTS = ['yr_last_error', 'yr_abi_version', 'yr_abi_sizeof', 'yr_create', 'yr_create_from_blob', 'yr_plan_io_dims', 'yr_destroy', 'yr_load_weights', 'yr_workspace_bytes',
           'yr_forward', 'yr_forward_profile', 'yr_forward_ranges', 'yr_autotune', 'yr_get_tuning', 'yr_set_tuning', 'yr_plan_num_launches', 'yr_op_run', 'yr_head_regions', 'yr_head_walk_rows', 'yr_head_stream_rows', 'yr_pwt_chunks', 'yr_decode', 'yr_decode_zoom', 'yr_yolo_head', 'yr_correct_boxes',
           'yr_nms', 'yr_pack_detections', 'yr_letterbox', 'yr_letterbox_batch']

_lib = None


class YoloretHipError(RuntimeError):
    pass


def lib():
    """Loads libyoloret_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise YoloretHipError(
                'libyoloret_hip.so is missing (%s). Build it with `python -m yoloret_amd.build` '
                '(hipcc --offload-arch=gfx950); there is no CPU fallback.' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.yr_last_error.restype = ctypes.c_char_p
        L.yr_abi_version.restype = ctypes.c_int
        L.yr_workspace_bytes.restype = ctypes.c_size_t
        L.yr_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.yr_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.yr_create_from_blob.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.yr_plan_io_dims.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.yr_destroy.argtypes = [ctypes.c_void_p]
        L.yr_destroy.restype = None
        L.yr_load_weights.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.yr_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.yr_plan_num_launches.argtypes = [ctypes.c_void_p]
        L.yr_forward_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.yr_forward_ranges.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        L.yr_autotune.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        L.yr_get_tuning.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.yr_set_tuning.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.yr_op_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.yr_head_regions.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.yr_head_walk_rows.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.yr_head_stream_rows.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.yr_decode.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 5
        L.yr_decode_zoom.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_float] * 2 + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 5
        L.yr_yolo_head.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + \
            [ctypes.c_void_p] * 6
        L.yr_correct_boxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.yr_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_float] + \
            [ctypes.c_void_p] * 3
        L.yr_pack_detections.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
        L.yr_letterbox.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_void_p]
        L.yr_letterbox_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        if L.yr_abi_version() != ABI_VERSION:
            raise YoloretHipError('libyoloret_hip.so ABI version mismatch')
        L.yr_abi_sizeof.argtypes = [ctypes.c_int]
        for which, st in enumerate((YrSrc, YrOp, YrBuf)):
            if L.yr_abi_sizeof(which) != ctypes.sizeof(st):
                raise YoloretHipError('struct %s: %d bytes here, %d in libyoloret_hip.so'
                                      % (st.__name__, ctypes.sizeof(st), L.yr_abi_sizeof(which)))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise YoloretHipError('libyoloret_hip: %s (status %d)' % (lib().yr_last_error().decode(), rc))


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _require_cuda_f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError('%s must be a contiguous float32 CUDA tensor' % name)


def make_src(t, c=None, xform='identity', ld=None):
    """yr_src for a [B,H,W,ld] tensor (float32 / bfloat16 / float16) of which `c` channels are taken."""
    s = YrSrc()
    s.dtype = dtype_id(t.dtype)
    s.ptr = t.data_ptr()
    s.buf = -1
    s.h, s.w = t.shape[1], t.shape[2]
    s.ld = t.shape[3] if ld is None else ld
    s.c = s.ld if c is None else c
    s.xform = XFORM[xform] if isinstance(xform, str) else int(xform)
    return s


def run_op(op, batch, device=None):
    """One fused op on `device` (default: the current device) - the tensors behind its pointers must live there."""
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(lib().yr_op_run(ctypes.byref(op), int(batch), stream_ptr(device)))


def new_op(kind, act='none'):
    op = YrOp()
    op.kind = kind
    op.act = ACT[act] if not isinstance(act, int) else act
    op.out_buf = op.res_buf = op.gate_buf = op.gate_out_buf = -1
    for f in ('wgt_off', 'scale_off', 'shift_off', 'wgt2_off', 'b1_off', 'b2_off', 'se_w_off'):
        setattr(op, f, -1)
    return op


# ----------------------------------------------------------------------------- post-processing wrappers
def num_boxes(in_h, in_w, num_anchors=3, num_scales=3):
    return sum((in_h // (32 >> s)) * (in_w // (32 >> s)) * num_anchors for s in range(num_scales))


# model.py:411-412: the zoom-in TTA pass is mapped back with these hard-coded constants
ZOOM_MUL = float(np.float32(224 / 416))
ZOOM_ADD = float(np.float32((416 - 224) / (2 * 416)))
ZOOM_RATIO = (224 * 224) / (416 * 416)     # utils.py:7: the central_crop fraction of the zoom pass (yolo.py:108-109)


def decode(ys, anchors, num_classes, image_hw, input_hw, num_scales=3, zoom_ys=None):
    """ys: list of [B,G,G,A*(C+5)] (or [B,G,G,A,C+5]) logits -> boxes [B,N,4], scores [B,C,N].
    zoom_ys: the logits of the zoom-in TTA pass (model.py:408-417) -> boxes [B,2N,4], scores [B,C,2N]."""
    if len(ys) < num_scales:
        raise ValueError('decode: %d logit tensors for %d scales' % (len(ys), num_scales))
    anchors = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(-1, 2))
    if anchors.shape[0] % 3 or anchors.shape[0] == 0:
        raise ValueError('decode: the anchor list must hold 3 scales x A anchors')
    a_ = anchors.shape[0] // 3
    in_h_, in_w_ = int(input_hw[0]), int(input_hw[1])
    for i, y in enumerate(ys[:num_scales]):
        _require_cuda_f32(y, 'y%d' % (i + 1))
        # the kernel derives the grids from input_hw and the strides, not from the tensors: the shapes must agree,
        # otherwise it would read past the end of y
        gh, gw = in_h_ // (32 >> i), in_w_ // (32 >> i)
        if y.dim() < 3 or tuple(y.shape[1:3]) != (gh, gw) or y[0].numel() != gh * gw * a_ * (num_classes + 5) \
                or y.shape[0] != ys[0].shape[0] or y.device != ys[0].device:
            raise ValueError('y%d has shape %s, expected [B=%d,%d,%d,%d*(%d+5)] for input %dx%d'
                             % (i + 1, tuple(y.shape), ys[0].shape[0], gh, gw, a_, num_classes, in_h_, in_w_))
    if not (isinstance(image_hw, torch.Tensor) and image_hw.dtype == torch.int32 and image_hw.is_contiguous()
            and image_hw.device == ys[0].device and tuple(image_hw.shape) == (ys[0].shape[0], 2)):
        raise ValueError('image_hw must be a contiguous int32 [B,2] tensor on the logits\' device (image_hw_tensor)')
    with torch.cuda.device(ys[0].device):
        return _decode(ys, anchors, num_classes, image_hw, input_hw, num_scales, zoom_ys)


def _decode(ys, anchors, num_classes, image_hw, input_hw, num_scales, zoom_ys):
    if zoom_ys is not None:
        for i, (y, z) in enumerate(zip(ys[:num_scales], zoom_ys[:num_scales])):
            _require_cuda_f32(z, 'zoom y%d' % (i + 1))
            if z.shape != y.shape:
                raise ValueError('zoom y%d has shape %s, expected %s' % (i + 1, tuple(z.shape), tuple(y.shape)))
        b = ys[0].shape[0]
        anchors = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(-1, 2))
        a = anchors.shape[0] // 3
        in_h, in_w = int(input_hw[0]), int(input_hw[1])
        n = 2 * num_boxes(in_h, in_w, a, num_scales)
        dev = ys[0].device
        boxes = torch.empty((b, n, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((b, num_classes, n), dtype=torch.float32, device=dev)
        yp = [_ptr(ys[i]) if i < num_scales else None for i in range(3)]
        zp = [_ptr(zoom_ys[i]) if i < num_scales else None for i in range(3)]
        check(lib().yr_decode_zoom(yp[0], yp[1], yp[2], zp[0], zp[1], zp[2], ZOOM_MUL, ZOOM_ADD, b, in_h, in_w, a,
                                   num_classes, num_scales, anchors.ctypes.data_as(ctypes.c_void_p), _ptr(image_hw),
                                   _ptr(boxes), _ptr(scores), stream_ptr(dev)))
        return boxes, scores
    b = ys[0].shape[0]
    anchors = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(-1, 2))
    a = anchors.shape[0] // 3
    in_h, in_w = int(input_hw[0]), int(input_hw[1])
    n = num_boxes(in_h, in_w, a, num_scales)
    dev = ys[0].device
    boxes = torch.empty((b, n, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((b, num_classes, n), dtype=torch.float32, device=dev)
    yp = [_ptr(ys[i]) if i < num_scales else None for i in range(3)]
    check(lib().yr_decode(yp[0], yp[1], yp[2], b, in_h, in_w, a, num_classes, num_scales,
                          anchors.ctypes.data_as(ctypes.c_void_p), _ptr(image_hw), _ptr(boxes), _ptr(scores),
                          stream_ptr(dev)))
    return boxes, scores


def nms(boxes, scores, max_boxes=20, score_threshold=.6, iou_threshold=.5):
    """boxes [B,N,4], scores [B,C,N] -> idx [B,C,max] int32 (-1 padded), count [B,C] int32."""
    _require_cuda_f32(boxes, 'boxes')
    _require_cuda_f32(scores, 'scores')
    if scores.dim() != 3 or tuple(boxes.shape) != (scores.shape[0], scores.shape[2], 4) or boxes.device != scores.device:
        raise ValueError('nms: boxes %s / scores %s, expected [B,N,4] / [B,C,N] on one device' % (tuple(boxes.shape), tuple(scores.shape)))
    b, c, n = scores.shape
    idx = torch.empty((b, c, max_boxes), dtype=torch.int32, device=boxes.device)
    cnt = torch.empty((b, c), dtype=torch.int32, device=boxes.device)
    with torch.cuda.device(boxes.device):
        check(lib().yr_nms(_ptr(boxes), _ptr(scores), b, n, c, int(max_boxes), float(score_threshold),
                           float(iou_threshold), _ptr(idx), _ptr(cnt), stream_ptr(boxes.device)))
    return idx, cnt


def pack_detections(boxes, scores, idx, cnt):
    """-> det [B, C*max, 6] int32 words, det_count [B] int32 (see yoloret_hip.h)."""
    b, c, n = scores.shape
    max_boxes = idx.shape[2]
    det = torch.empty((b, c * max_boxes, 6), dtype=torch.int32, device=boxes.device)
    det_count = torch.empty((b,), dtype=torch.int32, device=boxes.device)
    with torch.cuda.device(boxes.device):
        check(lib().yr_pack_detections(_ptr(boxes), _ptr(scores), _ptr(idx), _ptr(cnt), b, n, c, max_boxes,
                                       _ptr(det), _ptr(det_count), stream_ptr(boxes.device)))
    return det, det_count


def yolo_head(feats, anchors, input_hw, with_scores=False):
    """feats [B,G,G,A,C+5] -> box_xy, box_wh [B,G,G,A,2], conf [B,G,G,A,1], probs [B,G,G,A,C]
    (+ scores = conf*probs [B,G,G,A,C] when with_scores)."""
    _require_cuda_f32(feats, 'feats')
    b, gh, gw, a, ch = feats.shape
    c = ch - 5
    anchors = np.ascontiguousarray(np.asarray(anchors, np.float32).reshape(-1, 2))
    if anchors.shape[0] != a:
        raise ValueError('yolo_head: %d anchors for %d anchor slots' % (anchors.shape[0], a))
    dev = feats.device
    xy = torch.empty((b, gh, gw, a, 2), dtype=torch.float32, device=dev)
    wh = torch.empty((b, gh, gw, a, 2), dtype=torch.float32, device=dev)
    conf = torch.empty((b, gh, gw, a, 1), dtype=torch.float32, device=dev)
    probs = torch.empty((b, gh, gw, a, c), dtype=torch.float32, device=dev)
    scores = torch.empty_like(probs) if with_scores else None
    with torch.cuda.device(dev):
        check(lib().yr_yolo_head(_ptr(feats), b, gh, gw, a, c, anchors.ctypes.data_as(ctypes.c_void_p),
                                 int(input_hw[0]), int(input_hw[1]), _ptr(xy), _ptr(wh), _ptr(conf), _ptr(probs),
                                 _ptr(scores), stream_ptr(dev)))
    return (xy, wh, conf, probs, scores) if with_scores else (xy, wh, conf, probs)


def correct_boxes(box_xy, box_wh, input_hw, image_hw):
    _require_cuda_f32(box_xy, 'box_xy')
    _require_cuda_f32(box_wh, 'box_wh')
    b = box_xy.shape[0]
    n = box_xy[0].numel() // 2
    boxes = torch.empty(tuple(box_xy.shape[:-1]) + (4,), dtype=torch.float32, device=box_xy.device)
    if box_wh.shape != box_xy.shape or box_wh.device != box_xy.device:
        raise ValueError('correct_boxes: box_xy %s and box_wh %s differ' % (tuple(box_xy.shape), tuple(box_wh.shape)))
    with torch.cuda.device(box_xy.device):
        check(lib().yr_correct_boxes(_ptr(box_xy), _ptr(box_wh), b, n, int(input_hw[0]), int(input_hw[1]),
                                     _ptr(image_hw), _ptr(boxes), stream_ptr(box_xy.device)))
    return boxes


def letterbox(image_u8, input_hw, out=None):
    """image_u8: uint8 CUDA tensor [ih,iw,3] -> float32 [H,W,3] letterboxed network input; a batch of equally sized
    images [B,ih,iw,3] -> [B,H,W,3] in one launch."""
    if not (isinstance(image_u8, torch.Tensor) and image_u8.is_cuda and image_u8.dtype == torch.uint8
            and image_u8.dim() in (3, 4) and image_u8.shape[-1] == 3 and image_u8.is_contiguous()):
        raise ValueError('image must be a contiguous uint8 CUDA tensor [h,w,3] or [B,h,w,3]')
    h, w = int(input_hw[0]), int(input_hw[1])
    batched = image_u8.dim() == 4
    shape = ((image_u8.shape[0],) if batched else ()) + (h, w, 3)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=image_u8.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != image_u8.device:
        raise ValueError('out must be a contiguous float32 tensor %s on the image\'s device' % (shape,))
    with torch.cuda.device(image_u8.device):
        check(lib().yr_letterbox_batch(_ptr(image_u8), image_u8.shape[0] if batched else 1, image_u8.shape[-3],
                                       image_u8.shape[-2], _ptr(out), h, w, stream_ptr(image_u8.device)))
    return out


def image_hw_tensor(image_shape, batch, device):
    """image_shape: (h,w) or [B,2] -> int32 [B,2] device tensor."""
    if isinstance(image_shape, torch.Tensor):
        t = image_shape.to(device=device, dtype=torch.int32).reshape(-1, 2)
    else:
        t = torch.as_tensor(np.asarray(image_shape).reshape(-1, 2).astype(np.int32), device=device)
    if t.shape[0] == 1 and batch > 1:
        t = t.expand(batch, 2)
    if t.shape[0] != batch:
        raise ValueError('image_shape must be (h,w) or one (h,w) per image')
    return t.contiguous()


# ----------------------------------------------------------------------------- serialised plans
PLAN_MAGIC = b'YRPLAN\0\0'


def pack_plan(ops, bufs, weights, in_hw, out_hwc, tuning=None):
    """ctypes YrOp / YrBuf arrays + float32 parameter blob (+ {batch: [cfg per op]}) -> the byte blob
    yr_create_from_blob reads (layout: include/yoloret_hip.h)."""
    import struct
    tuning = tuning or {}
    weights = np.ascontiguousarray(weights, np.float32)
    head = PLAN_MAGIC + struct.pack('<6IQ2i9i3i', ABI_VERSION, len(ops), len(bufs), ctypes.sizeof(YrOp), ctypes.sizeof(YrBuf),
                                    len(tuning), weights.size, int(in_hw[0]), int(in_hw[1]),
                                    *[int(v) for hwc in out_hwc for v in hwc], 0, 0, 0)
    assert len(head) == 96
    parts = [head, bytes(ops), bytes(bufs), weights.tobytes()]
    for batch in sorted(tuning):
        tab = np.asarray([batch] + list(tuning[batch]), np.int32)
        assert tab.size == 1 + len(ops)
        parts.append(tab.tobytes())
    return b''.join(parts)


class PlanHandle:
    """A model instantiated from a serialised plan through yr_create_from_blob - nothing of the graph compiler is
    involved (what a C / C++ / cgo / JNI host does; this class is the ctypes rendition of INTEGRATION.md's recipe).
    __call__(images [B,H,W,3] float32 CUDA) -> [y1, y2, y3] raw logits [B,G,G,A*(C+5)]."""

    def __init__(self, blob, device=None):
        blob = bytes(blob) if not isinstance(blob, (bytes, bytearray)) else blob
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self._h = ctypes.c_void_p()
        buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
        with torch.cuda.device(self.device):
            check(lib().yr_create_from_blob(buf, len(blob), ctypes.byref(self._h)))
        in_hw = (ctypes.c_int32 * 2)()
        out = (ctypes.c_int32 * 9)()
        check(lib().yr_plan_io_dims(self._h, in_hw, out))
        self.input_hw = (in_hw[0], in_hw[1])
        self.output_hwc = [tuple(out[3 * i:3 * i + 3]) for i in range(3)]
        self._ws = None

    def __call__(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and tuple(x.shape[1:]) == self.input_hw + (3,)):
            raise ValueError('input must be a float32 CUDA tensor [B,%d,%d,3]' % self.input_hw)
        x = x.contiguous()
        b = x.shape[0]
        L = lib()
        with torch.cuda.device(x.device):
            need = L.yr_workspace_bytes(self._h, b)
            if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
                self._ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
            ys = [torch.empty((b,) + hwc, dtype=torch.float32, device=x.device) for hwc in self.output_hwc]
            check(L.yr_forward(self._h, _ptr(x), b, _ptr(ys[0]), _ptr(ys[1]), _ptr(ys[2]), _ptr(self._ws),
                               self._ws.numel(), stream_ptr(x.device)))
        return ys

    def __del__(self):
        try:
            if self._h:
                lib().yr_destroy(self._h)
        except Exception:
            pass
