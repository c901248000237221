"""MobileNetV2 backbone graph - host-side mirror of reference code/yolo3/override.py
(``mobilenet_v2`` :290-341), which forwards to ``tf.keras.applications.MobileNetV2``
with only BatchNormalization momentum overridden (:207-227).  The architecture
itself is third-party and is restated here with the Keras layer
names (``Conv1``, ``bn_Conv1``, ``expanded_conv_*``, ``block_{b}_*``) that the
reference uses to pick its taps (model.py:186-190).
"""
from .. import layers as L
from .efficientnet import BackboneModel


def _make_divisible(v, divisor, min_value=None):
    """override.py:56-63 (== model.py:32-39)."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


# (filters, stride, expansion) of inverted-residual blocks 0..16 [3P]
_BLOCKS = [(16, 1, 1), (24, 2, 6), (24, 1, 6), (32, 2, 6), (32, 1, 6), (32, 1, 6), (64, 2, 6), (64, 1, 6),
           (64, 1, 6), (64, 1, 6), (96, 1, 6), (96, 1, 6), (96, 1, 6), (160, 2, 6), (160, 1, 6), (160, 1, 6),
           (320, 1, 6)]


def _inverted_res_block(x, expansion, stride, alpha, filters, block_id, momentum):
    in_channels = x.shape[2]
    pointwise_filters = _make_divisible(int(filters * alpha), 8)
    prefix = 'block_{}_'.format(block_id) if block_id else 'expanded_conv_'
    inputs = x
    if block_id:
        x = L.Conv2D(expansion * in_channels, kernel_size=1, padding='same', use_bias=False, name=prefix + 'expand')(x)
        x = L.BatchNormalization(epsilon=1e-3, momentum=momentum, name=prefix + 'expand_BN')(x)
        x = L.ReLU(6., name=prefix + 'expand_relu')(x)
    # stride 2: Keras zero-pads ((0,1),(0,1)) then convolves VALID == TF 'SAME' for even sizes
    x = L.DepthwiseConv2D(kernel_size=3, strides=stride, use_bias=False, padding='same', name=prefix + 'depthwise')(x)
    x = L.BatchNormalization(epsilon=1e-3, momentum=momentum, name=prefix + 'depthwise_BN')(x)
    x = L.ReLU(6., name=prefix + 'depthwise_relu')(x)
    x = L.Conv2D(pointwise_filters, kernel_size=1, padding='same', use_bias=False, name=prefix + 'project')(x)
    x = L.BatchNormalization(epsilon=1e-3, momentum=momentum, name=prefix + 'project_BN')(x)
    if in_channels == pointwise_filters and stride == 1:
        return L.Add(name=prefix + 'add')([inputs, x]), True
    return x, False


def mobilenet_v2(default_batchnorm_momentum=0.9, alpha=1.0, input_tensor=None, include_top=False,
                 weights=None, last_block=15, **kwargs):
    """override.py:290-341.  ``include_top`` must be False and ``weights`` None (the ImageNet
    download of model.py:181 needs a network; load a full detector weight file instead).
    Blocks after ``last_block`` (15 = the deepest tap, ``block_15_add``) are dead for
    detection and are not built."""
    if include_top:
        raise ValueError('the detection path builds MobileNetV2 with include_top=False')
    if weights is not None:
        raise ValueError('pretrained backbone weights are not available offline; pass weights=None')
    if input_tensor is None:
        raise ValueError('input_tensor is required')
    if input_tensor.shape[0] % 32 or input_tensor.shape[1] % 32:
        raise ValueError('input size must be a multiple of 32')
    m = default_batchnorm_momentum
    named = {}
    first = _make_divisible(32 * alpha, 8)
    x = L.Conv2D(first, kernel_size=3, strides=2, padding='same', use_bias=False, name='Conv1')(input_tensor)
    x = L.BatchNormalization(epsilon=1e-3, momentum=m, name='bn_Conv1')(x)
    x = L.ReLU(6., name='Conv1_relu')(x)
    for b, (f, s, t) in enumerate(_BLOCKS[:last_block + 1]):
        x, added = _inverted_res_block(x, t, s, alpha, f, b, m)
        prefix = 'block_%d_' % b if b else 'expanded_conv_'
        named[prefix + ('add' if added else 'project_BN')] = x
    return BackboneModel(input_tensor, x, named)
