"""EfficientNet / MBConv / SE graph builders - host-side mirror of reference
code/yolo3/efficientnet.py for the detection forward path (inference only).

Same names, argument meaning and error behaviour as the reference
(``BlockArgs``, ``GlobalParams``, ``BlockDecoder``, ``efficientnet``,
``efficientnet_params``, ``get_model_params``, ``round_filters``,
``round_repeats``, ``Swish``, ``Mean``, ``SEBlock``, ``MBConvBlock``,
``EfficientNet``, ``EfficientNetB0..B7``), but layers come from
``yoloret_amd.layers`` and record a graph that is lowered to HIP kernels.
Layers that Keras would auto-name take explicit names via ``name=`` so that a
weight dict is reproducible: ``stage{n}_block{r}_{expand,dw,se_reduce,se_expand,project}[_BN]``.
"""
import collections
import math
import re

from .. import layers as L
from ..layers import Mean, Swish  # re-exported: efficientnet.py:327-331, :391-403

# The reference's two configuration records (efficientnet.py:110-130): field names are API (callers pass them as
# keyword arguments and use ._replace), every field defaults to None.
_GLOBAL_FIELDS = ('batch_norm_momentum batch_norm_epsilon dropout_rate data_format num_classes width_coefficient '
                  'depth_coefficient depth_divisor min_depth drop_connect_rate').split()
_BLOCK_FIELDS = 'kernel_size num_repeat input_filters output_filters expand_ratio id_skip strides se_ratio'.split()
GlobalParams = collections.namedtuple('GlobalParams', _GLOBAL_FIELDS, defaults=(None,) * len(_GLOBAL_FIELDS))
BlockArgs = collections.namedtuple('BlockArgs', _BLOCK_FIELDS, defaults=(None,) * len(_BLOCK_FIELDS))

# Block-string notation (efficientnet.py:133-200), e.g. 'r2_k5_s22_e6_i24_o40_se0.25[_noskip]':
# token prefix -> (BlockArgs field, parser, formatter); 's' holds the two stride digits.
_TOKENS = collections.OrderedDict([
    ('r', ('num_repeat', int, '%d')), ('k', ('kernel_size', int, '%d')),
    ('s', ('strides', lambda v: [int(ch) for ch in v], None)), ('e', ('expand_ratio', int, '%s')),
    ('i', ('input_filters', int, '%d')), ('o', ('output_filters', int, '%d')), ('se', ('se_ratio', float, '%s'))])
_TOKEN_RE = re.compile(r'^(se|[a-z])(\d.*)$')

# Stage table of the B0 baseline network (efficientnet.py:208-216) as rows of
# (repeats, kernel, stride, expand ratio, input filters, output filters); every stage has se_ratio 0.25.
_B0_STAGES = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
              (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))

# Compound-scaling coefficients (efficientnet.py:231-244): name -> (width, depth, resolution, dropout)
_SCALING = {0: (1.0, 1.0, 224, 0.2), 1: (1.0, 1.1, 240, 0.2), 2: (1.1, 1.2, 260, 0.3), 3: (1.2, 1.4, 300, 0.3),
            4: (1.4, 1.8, 380, 0.4), 5: (1.6, 2.2, 456, 0.4), 6: (1.8, 2.6, 528, 0.5), 7: (2.0, 3.1, 600, 0.5)}


class BlockDecoder(object):
    """String <-> BlockArgs (efficientnet.py:133-200), driven by the _TOKENS table."""

    def _decode_block_string(self, block_string):
        if not isinstance(block_string, str):
            raise AssertionError('block string expected')
        fields = {}
        for tok in block_string.split('_'):
            m = _TOKEN_RE.match(tok)
            if m and m.group(1) in _TOKENS:
                name, parse, _ = _TOKENS[m.group(1)]
                fields[name] = (m.group(2), parse)
        if 'strides' not in fields or len(fields['strides'][0]) != 2:
            raise ValueError('Strides options should be a pair of integers.')
        values = {name: parse(raw) for name, (raw, parse) in fields.items()}
        for prefix, (name, _, _) in _TOKENS.items():      # the reference indexes options['r'], ['k'], ... : KeyError
            if prefix != 'se' and name not in values:
                raise KeyError(prefix)
        values.setdefault('se_ratio', None)
        return BlockArgs(id_skip='noskip' not in block_string, **values)

    def _encode_block_string(self, block):
        parts = []
        for prefix, (name, _, fmt) in _TOKENS.items():
            value = getattr(block, name)
            if prefix == 's':
                parts.append('s%d%d' % tuple(value[:2]))
            elif prefix == 'se':
                if 0 < value <= 1:     # (se_ratio None: TypeError, as in the reference's comparison)
                    parts.append('se%s' % value)
            else:
                parts.append(prefix + fmt % value)
        if block.id_skip is False:
            parts.append('noskip')
        return '_'.join(parts)

    def decode(self, string_list):
        if not isinstance(string_list, list):
            raise AssertionError('a list of block strings expected')
        return [self._decode_block_string(s) for s in string_list]

    def encode(self, blocks_args):
        return [self._encode_block_string(b) for b in blocks_args]


def efficientnet(width_coefficient=None, depth_coefficient=None, dropout_rate=0.2, drop_connect_rate=0.2):
    """(list of BlockArgs, GlobalParams) of the baseline network scaled by the two coefficients
    (efficientnet.py:203-228)."""
    stages = [BlockArgs(kernel_size=k, num_repeat=r, input_filters=i, output_filters=o, expand_ratio=e, id_skip=True,
                        strides=[s, s], se_ratio=0.25) for r, k, s, e, i, o in _B0_STAGES]
    global_params = GlobalParams(batch_norm_momentum=0.99, batch_norm_epsilon=1e-3, dropout_rate=dropout_rate,
                                 drop_connect_rate=drop_connect_rate, data_format='channels_last', num_classes=1000,
                                 width_coefficient=width_coefficient, depth_coefficient=depth_coefficient,
                                 depth_divisor=8, min_depth=None)
    return stages, global_params


def efficientnet_params(model_name):
    """'efficientnet-b<n>' -> (width, depth, resolution, dropout) (efficientnet.py:231-244); KeyError otherwise."""
    m = re.match(r'^efficientnet-b([0-7])$', model_name)
    if not m:
        raise KeyError(model_name)
    return _SCALING[int(m.group(1))]


def get_model_params(model_name, override_params=None):
    """efficientnet.py:247-267.  Unlike the reference this does not mutate (or print) the caller's dict; 'drop_rate'
    is still accepted and ignored, and unknown keys still raise ValueError (from namedtuple._replace)."""
    if not model_name.startswith('efficientnet'):
        raise NotImplementedError('model name is not pre-defined: %s' % model_name)
    width, depth, resolution, dropout = efficientnet_params(model_name)
    blocks_args, global_params = efficientnet(width, depth, dropout)
    overrides = {k: v for k, v in (override_params or {}).items() if k != 'drop_rate'}
    if overrides:
        global_params = global_params._replace(**overrides)
    return blocks_args, global_params, resolution


def round_filters(filters, global_params):
    """Channel count scaled by the width coefficient, to a multiple of depth_divisor, never more than 10 % below the
    scaled value (efficientnet.py:364-380)."""
    width = global_params.width_coefficient
    if not width:
        return filters
    divisor = global_params.depth_divisor
    scaled = filters * width
    floor = global_params.min_depth or divisor
    rounded = max(floor, int(scaled + divisor / 2) // divisor * divisor)
    return int(rounded + divisor if rounded < 0.9 * scaled else rounded)


def round_repeats(repeats, global_params):
    """Block repeats scaled by the depth coefficient, rounded up (efficientnet.py:383-388)."""
    depth = global_params.depth_coefficient
    return int(math.ceil(depth * repeats)) if depth else repeats


def _require_channels_last(global_params):
    if global_params.data_format not in (None, 'channels_last'):
        raise ValueError('only data_format="channels_last" (NHWC) is supported on MI355X')


def SEBlock(block_args, global_params, quantize=False, name='se'):
    """Squeeze-excite (efficientnet.py:406-438)."""
    num_reduced_filters = max(1, int(block_args.input_filters * block_args.se_ratio))
    filters = block_args.input_filters * block_args.expand_ratio
    _require_channels_last(global_params)

    def block(inputs):
        x = Mean([1, 2], name=name + '_se_mean')(inputs)
        x = L.Conv2D(num_reduced_filters, kernel_size=[1, 1], strides=[1, 1], padding='same', use_bias=True,
                     name=name + '_se_reduce')(x)
        x = Swish(name=name + '_se_swish')(x)
        x = L.Conv2D(filters, kernel_size=[1, 1], strides=[1, 1], padding='same', use_bias=True,
                     name=name + '_se_expand')(x)
        x = L.Activation('sigmoid', name=name + '_se_sigmoid')(x)
        return L.Multiply(name=name + '_se_mul')([x, inputs])

    return block


def MBConvBlock(block_args, global_params, drop_connect_rate=None, quantize=False, name='mbconv', lite=False):
    """MBConv at inference (efficientnet.py:467-536); DropConnect (:334-361) is the identity.
    ``lite`` is build-defined (no reference counterpart): no SE, ReLU6 instead of Swish."""
    eps = global_params.batch_norm_epsilon
    mom = global_params.batch_norm_momentum
    _require_channels_last(global_params)
    has_se = (block_args.se_ratio is not None) and (block_args.se_ratio > 0) and (block_args.se_ratio <= 1) and not lite
    filters = block_args.input_filters * block_args.expand_ratio
    kernel_size = block_args.kernel_size
    act = (lambda n: L.ReLU(6., name=n)) if lite else (lambda n: Swish(name=n))

    def block(inputs):
        if block_args.expand_ratio != 1:
            x = L.Conv2D(filters, kernel_size=[1, 1], strides=[1, 1], padding='same', use_bias=False,
                         name=name + '_expand')(inputs)
            x = L.BatchNormalization(momentum=mom, epsilon=eps, name=name + '_expand_BN')(x)
            x = act(name + '_expand_act')(x)
        else:
            x = inputs
        x = L.DepthwiseConv2D([kernel_size, kernel_size], strides=block_args.strides, padding='same',
                              use_bias=False, name=name + '_dw')(x)
        x = L.BatchNormalization(momentum=mom, epsilon=eps, name=name + '_dw_BN')(x)
        x = act(name + '_dw_act')(x)
        if has_se:
            x = SEBlock(block_args, global_params, quantize=quantize, name=name)(x)
        x = L.Conv2D(block_args.output_filters, kernel_size=[1, 1], strides=[1, 1], padding='same',
                     use_bias=False, name=name + '_project')(x)
        x = L.BatchNormalization(momentum=mom, epsilon=eps, name=name + '_project_BN')(x)
        if block_args.id_skip:
            if all(s == 1 for s in block_args.strides) and block_args.input_filters == block_args.output_filters:
                x = L.Add(name=name + '_add')([x, inputs])
        return x

    return block


class BackboneModel:
    """What the reference gets back from tf.keras.Model(inputs, outputs) for a backbone:
    ``inputs``, ``output`` and ``get_layer(name).output`` (model.py:186-190,213-217)."""

    def __init__(self, inputs, output, named):
        self.inputs = [inputs]
        self.output = output
        self._named = named

    def get_layer(self, name):
        class _L:
            pass
        if name not in self._named:
            raise ValueError('No such layer: %s' % name)
        layer = _L()
        layer.output = self._named[name]
        layer.name = name
        return layer


def EfficientNet(input_shape, block_args_list, global_params, include_top=True, pooling=None,
                 input_tensor=None, quantize=False, lite=False, last_stage=None):
    """efficientnet.py:611-710 with include_top=False semantics for detection.  Stage ends are
    exposed as layers ``stage{n}`` (the reference taps them by Keras auto-names ``add_17`` ...,
    model.py:213-217, valid only for B3; positional names work for every width/depth)."""
    if include_top:
        raise ValueError('the detection path builds EfficientNet with include_top=False')
    eps = global_params.batch_norm_epsilon
    mom = global_params.batch_norm_momentum
    _require_channels_last(global_params)
    inputs = input_tensor if input_tensor is not None else L.Input(shape=input_shape)
    act = (lambda n: L.ReLU(6., name=n)) if lite else (lambda n: Swish(name=n))
    x = L.Conv2D(filters=round_filters(32, global_params), kernel_size=[3, 3], strides=[2, 2], padding='same',
                 use_bias=False, name='stem_conv')(inputs)
    x = L.BatchNormalization(momentum=mom, epsilon=eps, name='stem_BN')(x)
    x = act('stem_act')(x)
    named = {}
    for si, block_args in enumerate(block_args_list, start=1):
        if last_stage is not None and si > last_stage:
            break
        assert block_args.num_repeat > 0
        block_args = block_args._replace(
            input_filters=round_filters(block_args.input_filters, global_params),
            output_filters=round_filters(block_args.output_filters, global_params),
            num_repeat=round_repeats(block_args.num_repeat, global_params))
        x = MBConvBlock(block_args, global_params, name='stage%d_block0' % si, lite=lite)(x)
        if block_args.num_repeat > 1:
            block_args = block_args._replace(input_filters=block_args.output_filters, strides=[1, 1])
        for rep in range(1, block_args.num_repeat):
            x = MBConvBlock(block_args, global_params, name='stage%d_block%d' % (si, rep), lite=lite)(x)
        named['stage%d' % si] = x
    return BackboneModel(inputs, x, named)


def _get_model_by_name(model_name, input_shape=None, include_top=True, weights=None, classes=1000,
                       pooling=None, input_tensor=None, lite=False):
    """efficientnet.py:713-791 minus the ImageNet download (no network; weights are loaded
    into the whole detector afterwards)."""
    if weights not in {None}:
        raise ValueError('pretrained backbone weights are not available offline; pass weights=None and '
                         'load a full detector weight file instead')
    block_args_list, global_params, default_input_shape = get_model_params(model_name, override_params={'num_classes': classes})
    if input_shape is None:
        input_shape = (default_input_shape, default_input_shape, 3)
    # only stages 1..6 feed the detector
    return EfficientNet(input_shape, block_args_list, global_params, include_top=include_top, pooling=pooling,
                        input_tensor=input_tensor, lite=lite, last_stage=6)


def _make(name):
    def f(include_top=True, input_shape=None, weights=None, classes=1000, pooling=None, input_tensor=None, lite=False):
        return _get_model_by_name(name, include_top=include_top, input_shape=input_shape, weights=weights,
                                  classes=classes, pooling=pooling, input_tensor=input_tensor, lite=lite)
    f.__name__ = 'EfficientNetB' + name[-1]
    f.__doc__ = 'efficientnet.py:793-910 (%s), truncated after stage 6 for detection.' % name
    return f


EfficientNetB0 = _make('efficientnet-b0')
EfficientNetB1 = _make('efficientnet-b1')
EfficientNetB2 = _make('efficientnet-b2')
EfficientNetB3 = _make('efficientnet-b3')
EfficientNetB4 = _make('efficientnet-b4')
EfficientNetB5 = _make('efficientnet-b5')
EfficientNetB6 = _make('efficientnet-b6')
EfficientNetB7 = _make('efficientnet-b7')
