"""YOLO-ReT model definition and post-processing - host-side mirror of reference
code/yolo3/model.py for the detection forward path, on MI355X.

Same names, argument order, defaults and return structure as the reference;
``torch`` CUDA tensors replace ``tf.Tensor`` as the container and all arithmetic
runs in the HIP kernels of libyoloret_hip.so (no CPU fallback).

Graph builders (record a ``yoloret_amd.layers`` graph, lowered by ``yoloret_amd.compiler``):
  MobilenetSeparableConv2D :14-30, _make_divisible :32-39,
  make_last_layers_efficientnet_lite :91-115, WeightedSum :117-137, downsample_layer :139-144,
  rfcr_module :146-168, yolov3_body :170-342.
Post-processing (launch HIP kernels): yolo_head :344-371, yolo_correct_boxes :374-399,
  yolo_boxes_and_scores :402-428, yolo_eval :431-491, YoloEval :494-526.

Batch semantics: the reference folds the batch axis into the box list before NMS
(:425-427) and pins inference to batch 1 (yolo.py:84).  Here a batch means "the
reference applied to each image independently" (SURVEY.md D3): ``yolo_eval`` on a batch
of B>1 returns per-image lists; on B==1 it returns exactly the reference's 3 tensors.
"""
import numpy as np
import torch

from .. import layers as L
from .. import runtime as rt
from ..engine import Model
from ..layers import WeightedSum  # model.py:117-137
from . import efficientnet as _effnet
from .efficientnet import BlockArgs, EfficientNetB0, EfficientNetB3, MBConvBlock, get_model_params
from .override import mobilenet_v2
from .utils import compose

AdvLossModel = Model  # yolo3/train.py:10 - on this path only a Model container (model.py:342)


def MobilenetSeparableConv2D(filters, kernel_size, strides=(1, 1), padding='valid', use_bias=True, name='sepconv'):
    """DW kxk + BN + ReLU6 -> 1x1 + BN + ReLU6 (model.py:14-30)."""
    return compose(
        L.DepthwiseConv2D(kernel_size, padding=padding, use_bias=use_bias, strides=strides, name=name + '_dw'),
        L.BatchNormalization(name=name + '_dw_BN'), L.ReLU(6., name=name + '_dw_relu'),
        L.Conv2D(filters, 1, padding='same', use_bias=use_bias, strides=1, name=name + '_pw'),
        L.BatchNormalization(name=name + '_pw_BN'), L.ReLU(6., name=name + '_pw_relu'))


def _make_divisible(v, divisor, min_value=None):
    """model.py:32-39."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def make_last_layers_efficientnet_lite(x, block_args, global_params, quantize=False, name='head'):
    """Head block (model.py:91-115): 1x1->F + BN + ReLU6 -> MBConv(k3,e1,se.25,F->O); y = 1x1 O->O."""
    if global_params.data_format not in (None, 'channels_last'):
        raise ValueError('only channels_last is supported')
    num_filters = block_args.input_filters * block_args.expand_ratio
    x = compose(
        L.Conv2D(num_filters, kernel_size=1, padding='same', use_bias=False, name=name + '_conv'),
        L.BatchNormalization(epsilon=global_params.batch_norm_epsilon, momentum=global_params.batch_norm_momentum,
                             name=name + '_conv_BN'),
        L.ReLU(6., name=name + '_conv_relu'),
        MBConvBlock(block_args, global_params, drop_connect_rate=global_params.drop_connect_rate,
                    quantize=quantize, name=name + '_mb'))(x)
    y = L.Conv2D(block_args.output_filters, kernel_size=1, padding='same', use_bias=False, name=name + '_y')(x)
    return x, y


def downsample_layer(x, stride=2):
    """MaxPooling2D((stride, stride)) (model.py:139-144)."""
    return L.MaxPooling2D((stride, stride))(x)


def rfcr_module(inp_arr):
    """RFCR multi-scale fusion (model.py:146-168)."""
    b1c = L.Conv2D(48, kernel_size=1, padding='same', use_bias=False, name='rfcr_b1c')(inp_arr[0])
    b2c = L.Conv2D(48, kernel_size=1, padding='same', use_bias=False, name='rfcr_b2c')(inp_arr[1])
    b3c = L.Conv2D(48, kernel_size=1, padding='same', use_bias=False, name='rfcr_b3c')(inp_arr[2])
    b4c = L.Conv2D(48, kernel_size=1, padding='same', use_bias=False, name='rfcr_b4c')(inp_arr[3])
    bc = WeightedSum(name='rfcr_wsum')([L.UpSampling2D()(b1c), b2c, downsample_layer(b3c), b4c])
    bc = MobilenetSeparableConv2D(96, kernel_size=(5, 5), use_bias=False, padding='same', name='rfcr_sep')(bc)
    b1 = L.Concatenate()([inp_arr[0], downsample_layer(bc)])
    b2 = L.Concatenate()([inp_arr[1], bc])
    b3 = L.Concatenate()([inp_arr[2], L.UpSampling2D()(bc)])
    return b1, b2, b3


def _conv_bn_relu6(filters, name, bn_name):
    return compose(L.Conv2D(filters, kernel_size=1, padding='same', use_bias=False, name=name),
                   L.BatchNormalization(momentum=0.9, name=bn_name), L.ReLU(6., name=name + '_relu6'))


# (the reference wires B3 only, model.py:205-217; its efficientnet.py:231-244 scales B0 .. B7 and the taps are stage ends, so every width builds)
_EFFNET_BUILDERS = {'efficientnetb%d' % i: getattr(_effnet, 'EfficientNetB%d' % i) for i in range(8)}


def yolov3_body(inputs, model_name, num_anchors, **kwargs):
    """Builds backbone + RFCR + FPN/PANet heads (model.py:170-342) and returns a ``Model``
    mapping images [B,H,W,3] to [y1,y2,y3] = raw logits [B,G,G,num_anchors,num_classes+5]
    for strides 32/16/8.  ``inputs``: ``yoloret_amd.layers.Input(shape=[H,W,3])``.
    ``model_name`` in {'mobilenetv2x75','mobilenetv2x14','efficientnetb3'} as in the reference,
    plus the build-defined 'efficientnetb0' .. 'efficientnetb7' (the other widths of efficientnet.py:231-244) and '<efficientnet>-lite'
    variants.
    kwargs: num_classes, drop_rate, data_format, batch_norm_momentum, batch_norm_epsilon,
    drop_connect_rate (efficientnet.py:259-265); unknown keys raise ValueError."""
    L.reset_names()
    _, global_params, _ = get_model_params('efficientnet-b0', kwargs)
    num_classes = global_params.num_classes
    if global_params.data_format not in (None, 'channels_last'):
        raise ValueError('only data_format="channels_last" is supported')
    if model_name == 'mobilenetv2x75' or model_name == 'mobilenetv2x14':
        alpha = 0.75 if model_name == 'mobilenetv2x75' else 1.4
        backbone = mobilenet_v2(default_batchnorm_momentum=0.9, alpha=alpha, input_tensor=inputs,
                                include_top=False, weights=None)
        b1 = backbone.get_layer('block_15_add').output
        b2 = backbone.get_layer('block_12_add').output
        b3 = backbone.get_layer('block_5_add').output
        b4 = backbone.get_layer('block_2_add').output
    else:
        base, lite = (model_name[:-5], True) if model_name.endswith('-lite') else (model_name, False)
        if base not in _EFFNET_BUILDERS:
            raise ValueError('unknown model_name %r' % (model_name,))
        backbone = _EFFNET_BUILDERS[base](include_top=False, weights=None, input_tensor=inputs, lite=lite)
        # add_17 / add_12 / add_4 / add_2 of B3 (model.py:213-216) == ends of stages 6/5/3/2
        b1 = backbone.get_layer('stage6').output
        b2 = backbone.get_layer('stage5').output
        b3 = backbone.get_layer('stage3').output
        b4 = backbone.get_layer('stage2').output
    b4 = downsample_layer(b4, stride=4)

    b1, b2, b3 = rfcr_module([b1, b2, b3, b4])

    out_filters = num_anchors * (num_classes + 5)
    block_args = BlockArgs(kernel_size=3, num_repeat=1, input_filters=512, output_filters=out_filters,
                           expand_ratio=1, id_skip=True, se_ratio=0.25, strides=[1, 1])
    # top-down pass (fpn=True); with panet=True its y convs are discarded (model.py:240-241,260-261,280-281)
    x, _ = make_last_layers_efficientnet_lite(b1, block_args, global_params, name='td1')
    c1 = x
    x = _conv_bn_relu6(256, 'block_20_conv', 'block_20_BN')(x)
    x = L.Concatenate()([L.UpSampling2D()(x), b2])
    block_args = block_args._replace(input_filters=256)
    x, _ = make_last_layers_efficientnet_lite(x, block_args, global_params, name='td2')
    c2 = x
    x = _conv_bn_relu6(128, 'block_24_conv', 'block_24_BN')(x)
    x = L.Concatenate()([L.UpSampling2D()(x), b3])
    block_args = block_args._replace(input_filters=128)
    x, _ = make_last_layers_efficientnet_lite(x, block_args, global_params, name='td3')
    c3 = x
    # bottom-up pass (panet=True, model.py:283-323)
    x, y3 = make_last_layers_efficientnet_lite(c3, block_args, global_params, name='bu3')
    x = _conv_bn_relu6(128, 'bu3_down_conv', 'bu3_down_BN')(x)
    x = L.Concatenate()([downsample_layer(x), c2])
    block_args = block_args._replace(input_filters=256)
    x, y2 = make_last_layers_efficientnet_lite(x, block_args, global_params, name='bu2')
    x = _conv_bn_relu6(256, 'bu2_down_conv', 'bu2_down_BN')(x)
    x = L.Concatenate()([downsample_layer(x), c1])
    block_args = block_args._replace(input_filters=512)
    x, y1 = make_last_layers_efficientnet_lite(x, block_args, global_params, name='bu1')

    y1 = L.Reshape5(num_anchors, name='y1')(y1)
    y2 = L.Reshape5(num_anchors, name='y2')(y2)
    y3 = L.Reshape5(num_anchors, name='y3')(y3)
    return AdvLossModel(backbone.inputs, [y1, y2, y3])


# ----------------------------------------------------------------------------- post-processing
def _hw(shape):
    if isinstance(shape, torch.Tensor):
        shape = shape.tolist()
    return int(shape[0]), int(shape[1])


def yolo_head(feats, anchors, input_shape, calc_loss=False):
    """Convert final layer features to bounding box parameters (model.py:344-371).
    feats [B,G,G,A,C+5] CUDA f32; returns (box_xy, box_wh, box_confidence, box_class_probs)."""
    if calc_loss:
        raise NotImplementedError('calc_loss=True is the training path (out of scope)')
    return rt.yolo_head(feats.contiguous(), anchors, _hw(input_shape))


def yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape):
    """Letterbox inverse + clip (model.py:374-399); image_shape (h,w) or one per image."""
    hw = rt.image_hw_tensor(image_shape, box_xy.shape[0], box_xy.device)
    return rt.correct_boxes(box_xy.contiguous(), box_wh.contiguous(), _hw(input_shape), hw)


def yolo_boxes_and_scores(feats, anchors, num_classes, input_shape, image_shape, zoom_feats=None):
    """model.py:402-428.  Returns boxes [B*G*G*A,4] and scores [B*G*G*A,C], batch folded in
    exactly like the reference's reshapes (:425,427)."""
    box_xy, box_wh, _, _, box_scores = rt.yolo_head(feats.contiguous(), anchors, _hw(input_shape),
                                                     with_scores=True)
    if zoom_feats is not None:
        # zoom-in TTA (model.py:408-417): same head on the second set of logits, mapped back with the reference's
        # hard-coded constants, concatenated on the anchor axis.  This per-scale, reference-layout helper glues the
        # two head outputs with elementwise tensor ops (multiply, then add: the same two fp32 roundings); the fused
        # path (yolo_eval / yr_decode_zoom) does it inside the decode kernel.
        xy_z, wh_z, _, _, scores_z = rt.yolo_head(zoom_feats.contiguous(), anchors, _hw(input_shape), with_scores=True)
        xy_z = xy_z * rt.ZOOM_MUL + rt.ZOOM_ADD
        wh_z = wh_z * rt.ZOOM_MUL
        box_xy = torch.cat([box_xy, xy_z], -2)
        box_wh = torch.cat([box_wh, wh_z], -2)
        box_scores = torch.cat([box_scores, scores_z], -2)
    boxes = yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape).reshape(-1, 4)
    return boxes, box_scores.reshape(-1, num_classes)


def yolo_eval(yolo_outputs, anchors, num_scales, num_classes, image_shape, max_boxes=20, score_threshold=.6,
              iou_threshold=.5, zoom_outputs=None):
    """Evaluate YOLO model on given input and return filtered boxes (model.py:431-491).

    yolo_outputs: [y1,y2,y3] CUDA tensors [B,G,G,A,C+5]; image_shape: (h,w) or [B,2].
    B==1: returns (boxes int32 [K,4] (ymin,xmin,ymax,xmax), scores f32 [K], classes int32 [K]),
    class-ascending then NMS pick order.  B>1: returns a list of such triples, one per image."""
    det, cnt = yolo_eval_packed(yolo_outputs, anchors, num_scales, num_classes, image_shape, max_boxes,
                                score_threshold, iou_threshold, zoom_outputs=zoom_outputs)
    res = unpack_detections(det, cnt)
    return res[0] if len(res) == 1 else res


def yolo_eval_packed(yolo_outputs, anchors, num_scales, num_classes, image_shape, max_boxes=20,
                     score_threshold=.6, iou_threshold=.5, zoom_outputs=None):
    """Device-resident form of ``yolo_eval``: decode -> per-(image,class) NMS -> packed records.
    Returns det int32 [B, C*max_boxes, 6] and det_count int32 [B] (layout: include/yoloret_hip.h).
    zoom_outputs: the logits of the zoom-in TTA pass (model.py:454-459); the NMS then runs over 2N boxes."""
    ys = [y.contiguous() for y in yolo_outputs[:num_scales]]
    zs = None if zoom_outputs is None else [z.contiguous() for z in zoom_outputs[:num_scales]]
    b = ys[0].shape[0]
    anchors = np.asarray(anchors, np.float32).reshape(-1, 2)
    input_hw = (ys[0].shape[1] * 32, ys[0].shape[2] * 32)  # model.py:449
    hw = rt.image_hw_tensor(image_shape, b, ys[0].device)
    boxes, scores = rt.decode(ys, anchors, num_classes, hw, input_hw, num_scales, zoom_ys=zs)
    idx, count = rt.nms(boxes, scores, max_boxes, score_threshold, iou_threshold)
    return rt.pack_detections(boxes, scores, idx, count)


def unpack_detections(det, det_count):
    """Packed records -> list of (boxes int32 [K,4], scores f32 [K], classes int32 [K]) on the host side
    of the boundary (one synchronising copy of the counts)."""
    counts = det_count.tolist()
    out = []
    for i, k in enumerate(counts):
        rows = det[i, :k]
        out.append((rows[:, 0:4].contiguous(), rows[:, 4].contiguous().view(torch.float32), rows[:, 5].contiguous()))
    return out


class YoloEval:
    """Keras-layer wrapper around yolo_eval (model.py:494-526)."""

    def __init__(self, anchors, num_scales, num_classes, max_boxes=20, score_threshold=.6, iou_threshold=.5,
                 name=None, **kwargs):
        self.anchors = anchors
        self.num_scales = num_scales
        self.num_classes = num_classes
        self.max_boxes = max_boxes
        self.score_threshold = score_threshold
        self.iou_threshold = iou_threshold
        self.name = name

    def call(self, yolo_outputs, image_shape, zoom_outputs=None):
        return yolo_eval(yolo_outputs, self.anchors, self.num_scales, self.num_classes, image_shape,
                         self.max_boxes, self.score_threshold, self.iou_threshold, zoom_outputs=zoom_outputs)

    __call__ = call

    def get_config(self):
        return {'name': self.name, 'anchors': self.anchors, 'num_scales': self.num_scales,
                'num_classes': self.num_classes, 'max_boxes': self.max_boxes,
                'score_threshold': self.score_threshold, 'iou_threshold': self.iou_threshold}
