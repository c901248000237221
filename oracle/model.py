"""NumPy restatement of the reference's graph builder, executed eagerly.

Test infrastructure only (see ``oracle/__init__.py``).  Follows, function by
function, /root/reference/code/yolo3/model.py, .../efficientnet.py and the
third-party ``tf.keras.applications.MobileNetV2`` graph.
All tensors NHWC; ``P`` is a parameter provider (``oracle.params.ParamStore``).
``P.store(x)`` marks the output of every convolution (+ BN + activation), gate and merge of the GRAPH - the places
where the unfused layer-by-layer execution holds a tensor in memory: the identity for the float32 oracle, a rounding
to bfloat16 / float16 for ``oracle.params.QuantStore`` (the emulation the 16-bit configs are measured against; the
reference itself has no reduced-precision mode).  The placement follows the reference's layers, not the product's
fusion boundaries; ``P.entry`` (image + stem kernel) is the one point that depends on the plan under test, and the
test passes that in (``QuantStore(round_entry=...)``).

Layer names: MobileNetV2 layers use the Keras names (``Conv1``, ``bn_Conv1``,
``expanded_conv_*``, ``block_{b}_{expand,depthwise,project}[_BN]``); layers the
reference leaves to Keras auto-naming get the explicit names used below (the
same names the product's graph builder uses - that is the weight contract).
"""
import collections
import math

import numpy as np

from . import nn

BN_EPS = 1e-3  # Keras default == efficientnet.py:218 == MobileNetV2's own value


# --------------------------------------------------------------------------- helpers
def _make_divisible(v, divisor, min_value=None):
    """reference code/yolo3/model.py:32-39 (== override.py:56-63)."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def _bn(P, name, x):
    g, b, m, v = P.bn(name, x.shape[-1])
    return nn.batchnorm(x, g, b, m, v, BN_EPS)


def _conv1x1(P, name, x, cout, bias=False):
    y = nn.pointwise(x, P.conv(name, 1, x.shape[-1], cout)[0, 0])
    if bias:
        y = y + P.bias(name, cout).astype(x.dtype)
    return y


# --------------------------------------------------------------------------- MobileNetV2 [3P]
MBV2_BLOCKS = [  # (filters, stride, expansion) for block ids 0..16
    (16, 1, 1), (24, 2, 6), (24, 1, 6), (32, 2, 6), (32, 1, 6), (32, 1, 6),
    (64, 2, 6), (64, 1, 6), (64, 1, 6), (64, 1, 6), (96, 1, 6), (96, 1, 6), (96, 1, 6),
    (160, 2, 6), (160, 1, 6), (160, 1, 6), (320, 1, 6)]


def mobilenet_v2(P, x, alpha, last_block=15):
    """tf.keras.applications.MobileNetV2(alpha, include_top=False) up to
    ``block_{last_block}``; returns dict of named activations.  Called by the
    reference at code/yolo3/model.py:180,193 via override.py:290-341 (the
    override only swaps BatchNormalization momentum, override.py:207-227)."""
    acts = {}
    first = _make_divisible(32 * alpha, 8)
    x, w = P.entry(x, P.conv('Conv1', 3, 3, first))
    x = nn.conv2d(x, w, stride=2, padding='same')
    x = P.store(nn.relu6(_bn(P, 'bn_Conv1', x)))
    acts['Conv1_relu'] = x
    for b, (f, s, t) in enumerate(MBV2_BLOCKS[:last_block + 1]):
        prefix = 'expanded_conv_' if b == 0 else 'block_%d_' % b
        cin = x.shape[-1]
        cout = _make_divisible(int(f * alpha), 8)
        inp = x
        if b > 0:
            x = _conv1x1(P, prefix + 'expand', x, t * cin)
            x = P.store(nn.relu6(_bn(P, prefix + 'expand_BN', x)))
        x = nn.depthwise(x, P.dw(prefix + 'depthwise', 3, x.shape[-1]), stride=s, padding='same')
        x = P.store(nn.relu6(_bn(P, prefix + 'depthwise_BN', x)))
        x = _conv1x1(P, prefix + 'project', x, cout)
        x = _bn(P, prefix + 'project_BN', x)
        if cin == cout and s == 1:
            x = P.store(inp + x)
            acts['block_%d_add' % b] = x
        else:
            x = P.store(x)
        acts['block_%d_out' % b] = x
    return acts


# --------------------------------------------------------------------------- EfficientNet (in-repo)
BlockArgs = collections.namedtuple('BlockArgs', [
    'kernel_size', 'num_repeat', 'input_filters', 'output_filters',
    'expand_ratio', 'id_skip', 'strides', 'se_ratio'])

# efficientnet.py:208-216 stage strings, decoded (r,k,s,e,i,o,se)
EFFNET_STAGES = [(1, 3, 1, 1, 32, 16, 0.25), (2, 3, 2, 6, 16, 24, 0.25), (2, 5, 2, 6, 24, 40, 0.25),
                 (3, 3, 2, 6, 40, 80, 0.25), (3, 5, 1, 6, 80, 112, 0.25), (4, 5, 2, 6, 112, 192, 0.25),
                 (1, 3, 1, 6, 192, 320, 0.25)]
EFFNET_COEFFS = {  # efficientnet.py:231-244 (width, depth)
    'efficientnet-b0': (1.0, 1.0), 'efficientnet-b1': (1.0, 1.1), 'efficientnet-b2': (1.1, 1.2),
    'efficientnet-b3': (1.2, 1.4), 'efficientnet-b4': (1.4, 1.8), 'efficientnet-b5': (1.6, 2.2),
    'efficientnet-b6': (1.8, 2.6), 'efficientnet-b7': (2.0, 3.1)}


def round_filters(filters, width, divisor=8, min_depth=None):
    """efficientnet.py:364-380."""
    if not width:
        return filters
    filters *= width
    min_depth = min_depth or divisor
    new_filters = max(min_depth, int(filters + divisor / 2) // divisor * divisor)
    if new_filters < 0.9 * filters:
        new_filters += divisor
    return int(new_filters)


def round_repeats(repeats, depth):
    """efficientnet.py:383-388."""
    if not depth:
        return repeats
    return int(math.ceil(depth * repeats))


def se_block(P, name, x, input_filters, se_ratio):
    """efficientnet.py:406-438: mean_hw -> 1x1(+bias) -> Swish -> 1x1(+bias) -> sigmoid -> multiply."""
    reduced = max(1, int(input_filters * se_ratio))
    s = nn.mean_hw(x)
    s = nn.swish(_conv1x1(P, name + '_se_reduce', s, reduced, bias=True))
    s = nn.sigmoid(_conv1x1(P, name + '_se_expand', s, x.shape[-1], bias=True))
    return P.store(s * x)


def mbconv_block(P, name, x, a, lite=False):
    """efficientnet.py:467-536 at inference (DropConnect is identity).  ``lite``
    is build-defined (no reference counterpart): no SE, ReLU6 instead of Swish."""
    act = nn.relu6 if lite else nn.swish
    inp = x
    filters = a.input_filters * a.expand_ratio
    if a.expand_ratio != 1:
        x = P.store(act(_bn(P, name + '_expand_BN', _conv1x1(P, name + '_expand', x, filters))))
    x = nn.depthwise(x, P.dw(name + '_dw', a.kernel_size, x.shape[-1]), stride=a.strides[0], padding='same')
    x = P.store(act(_bn(P, name + '_dw_BN', x)))
    has_se = (not lite) and a.se_ratio is not None and 0 < a.se_ratio <= 1
    if has_se:
        x = se_block(P, name, x, a.input_filters, a.se_ratio)
    x = _bn(P, name + '_project_BN', _conv1x1(P, name + '_project', x, a.output_filters))
    if a.id_skip and all(s == 1 for s in a.strides) and a.input_filters == a.output_filters:
        x = x + inp
    return P.store(x)


def efficientnet(P, x, width, depth, lite=False, last_stage=6):
    """efficientnet.py:611-710 (include_top=False), stages 1..last_stage;
    returns {'stage{n}': activation at end of stage n}."""
    act = nn.relu6 if lite else nn.swish
    stem = round_filters(32, width)
    w = P.conv('stem_conv', 3, 3, stem)
    x, w = P.entry(x, w)
    x = nn.conv2d(x, w, stride=2, padding='same')
    x = P.store(act(_bn(P, 'stem_BN', x)))
    acts = {}
    for si, (r, k, s, e, i, o, se) in enumerate(EFFNET_STAGES[:last_stage], start=1):
        a = BlockArgs(kernel_size=k, num_repeat=round_repeats(r, depth),
                      input_filters=round_filters(i, width), output_filters=round_filters(o, width),
                      expand_ratio=e, id_skip=True, strides=[s, s], se_ratio=se)
        x = mbconv_block(P, 'stage%d_block0' % si, x, a, lite)
        if a.num_repeat > 1:
            a = a._replace(input_filters=a.output_filters, strides=[1, 1])
        for rep in range(1, a.num_repeat):
            x = mbconv_block(P, 'stage%d_block%d' % (si, rep), x, a, lite)
        acts['stage%d' % si] = x
    return acts


# --------------------------------------------------------------------------- RFCR + head
def mobilenet_separable_conv2d(P, name, x, filters, kernel_size):
    """model.py:14-30 (live use: RFCR, k=5, bias-free, SAME)."""
    x = nn.depthwise(x, P.dw(name + '_dw', kernel_size, x.shape[-1]), 1, 'same')
    x = P.store(nn.relu6(_bn(P, name + '_dw_BN', x)))
    x = _conv1x1(P, name + '_pw', x, filters)
    return P.store(nn.relu6(_bn(P, name + '_pw_BN', x)))


def rfcr_module(P, inp_arr):
    """model.py:146-168."""
    b1c = P.store(_conv1x1(P, 'rfcr_b1c', inp_arr[0], 48))
    b2c = P.store(_conv1x1(P, 'rfcr_b2c', inp_arr[1], 48))
    b3c = P.store(_conv1x1(P, 'rfcr_b3c', inp_arr[2], 48))
    b4c = P.store(_conv1x1(P, 'rfcr_b4c', inp_arr[3], 48))
    a = P.alpha('rfcr_wsum').astype(b1c.dtype)
    # model.py:134 - left-to-right sum
    bc = P.store(a[0] * nn.upsample2(b1c) + a[1] * b2c + a[2] * nn.maxpool(b3c, 2) + a[3] * b4c)
    bc = mobilenet_separable_conv2d(P, 'rfcr_sep', bc, 96, 5)
    b1 = nn.concat([inp_arr[0], nn.maxpool(bc, 2)])
    b2 = nn.concat([inp_arr[1], bc])
    b3 = nn.concat([inp_arr[2], nn.upsample2(bc)])
    return b1, b2, b3


def make_last_layers_efficientnet_lite(P, name, x, input_filters, out_filters):
    """model.py:91-115: 1x1->F +BN+ReLU6 -> MBConv(k3,s1,e1,se.25,F->O) ; y = 1x1 O->O."""
    x = P.store(nn.relu6(_bn(P, name + '_conv_BN', _conv1x1(P, name + '_conv', x, input_filters))))
    a = BlockArgs(kernel_size=3, num_repeat=1, input_filters=input_filters, output_filters=out_filters,
                  expand_ratio=1, id_skip=True, strides=[1, 1], se_ratio=0.25)
    x = mbconv_block(P, name + '_mb', x, a)
    y = _conv1x1(P, name + '_y', x, out_filters)
    return x, y


def _cbr(P, name, bn_name, x, filters):
    return P.store(nn.relu6(_bn(P, bn_name, _conv1x1(P, name, x, filters))))


def backbone_taps(P, x, model_name):
    """Taps b1..b4 as picked at model.py:186-190 / :199-203 / :213-217."""
    if model_name in ('mobilenetv2x75', 'mobilenetv2x14'):
        alpha = 0.75 if model_name == 'mobilenetv2x75' else 1.4
        acts = mobilenet_v2(P, x, alpha)
        return (acts['block_15_add'], acts['block_12_add'], acts['block_5_add'],
                nn.maxpool(acts['block_2_add'], 4))
    name, lite = model_name, False
    if name.endswith('-lite'):
        name, lite = name[:-5], True
    key = {'efficientnetb%d' % i: 'efficientnet-b%d' % i for i in range(8)}[name]
    w, d = EFFNET_COEFFS[key]
    acts = efficientnet(P, x, w, d, lite=lite)
    # add_17/add_12/add_4/add_2 of B3 == ends of stages 6/5/3/2
    return acts['stage6'], acts['stage5'], acts['stage3'], nn.maxpool(acts['stage2'], 4)


def yolov3_body(P, x, model_name, num_anchors=3, num_classes=20):
    """model.py:170-342 at inference; returns [y1,y2,y3] raw logits
    [B,G,G,num_anchors,num_classes+5] for strides 32,16,8."""
    out_f = num_anchors * (num_classes + 5)
    b1, b2, b3, b4 = backbone_taps(P, x, model_name)
    b1, b2, b3 = rfcr_module(P, [b1, b2, b3, b4])
    # top-down (fpn=True); with panet=True the first-pass y convs are discarded (model.py:240-241)
    x, _ = make_last_layers_efficientnet_lite(P, 'td1', b1, 512, out_f)
    c1 = x
    x = _cbr(P, 'block_20_conv', 'block_20_BN', x, 256)
    x = nn.concat([nn.upsample2(x), b2])
    x, _ = make_last_layers_efficientnet_lite(P, 'td2', x, 256, out_f)
    c2 = x
    x = _cbr(P, 'block_24_conv', 'block_24_BN', x, 128)
    x = nn.concat([nn.upsample2(x), b3])
    x, _ = make_last_layers_efficientnet_lite(P, 'td3', x, 128, out_f)
    c3 = x
    # bottom-up (panet=True), model.py:283-323
    x, y3 = make_last_layers_efficientnet_lite(P, 'bu3', c3, 128, out_f)
    x = _cbr(P, 'bu3_down_conv', 'bu3_down_BN', x, 128)
    x = nn.concat([nn.maxpool(x, 2), c2])
    x, y2 = make_last_layers_efficientnet_lite(P, 'bu2', x, 256, out_f)
    x = _cbr(P, 'bu2_down_conv', 'bu2_down_BN', x, 256)
    x = nn.concat([nn.maxpool(x, 2), c1])
    x, y1 = make_last_layers_efficientnet_lite(P, 'bu1', x, 512, out_f)
    rs = lambda y: y.reshape(y.shape[0], y.shape[1], y.shape[2], num_anchors, num_classes + 5)
    return [rs(y1), rs(y2), rs(y3)]
