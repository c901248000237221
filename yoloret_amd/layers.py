"""Symbolic NHWC graph with a Keras-like layer vocabulary.

The reference builds its network with ``tf.keras.layers`` calls
(code/yolo3/model.py, code/yolo3/efficientnet.py).  The host-side mirror keeps
that style - ``Conv2D(...)(x)``, ``BatchNormalization()(x)``, ``Concatenate()([a, b])`` -
but records a small graph instead of creating TF ops; ``yoloret_amd.compiler``
then lowers the graph to the fused HIP ops of libyoloret_hip.so.

Every layer owns named parameters (``<name>/kernel`` ...) with Keras layouts
(SURVEY.md A.5), so a weight dict keyed by those names fully specifies a model.
"""
import itertools
import os

_uid = itertools.count()

# ---- compute policy: the counterpart of tf.keras.mixed_precision.set_global_policy for models built afterwards.
# 'float32' (default; also YOLORET_DTYPE), 'mixed_bfloat16', 'mixed_float16': activations between the fused ops and
# the 1x1-conv weights are stored in the 16-bit type, variables / BatchNorm / accumulation / logits stay float32.
_POLICIES = {'float32': 'float32', 'mixed_bfloat16': 'bfloat16', 'mixed_float16': 'float16',
             'bfloat16': 'bfloat16', 'float16': 'float16', 'bf16': 'bfloat16', 'f16': 'float16', 'f32': 'float32'}
_policy = [None]


def set_global_policy(name):
    if name not in _POLICIES:
        raise ValueError('unknown policy %r (float32, mixed_bfloat16, mixed_float16)' % (name,))
    _policy[0] = _POLICIES[name]


def global_policy_dtype():
    if _policy[0] is not None:
        return _policy[0]
    env = os.environ.get('YOLORET_DTYPE', 'float32')
    if env not in _POLICIES:
        raise ValueError('YOLORET_DTYPE=%r (float32, bfloat16, float16)' % (env,))
    return _POLICIES[env]


class Tensor:
    """Symbolic activation; ``shape`` = (H, W, C), the batch axis is implicit."""

    def __init__(self, shape, node=None, name=None):
        self.shape = tuple(int(s) for s in shape)
        self.node = node
        self.name = name or ('t%d' % next(_uid))

    def __repr__(self):
        return 'Tensor(%s, %s)' % (self.name, 'x'.join(map(str, self.shape)))


class Node:
    def __init__(self, op, inputs, output_shape, attrs=None, params=None, name=None):
        self.op = op
        self.inputs = list(inputs)
        self.attrs = dict(attrs or {})
        self.params = dict(params or {})   # param name -> shape
        self.seq = next(_uid)              # creation order (what Keras' automatic layer names are numbered by)
        self.name = name or ('%s_%d' % (op, self.seq))
        self.output = Tensor(output_shape, self, self.name + ':0')


def Input(shape, batch_size=None, dtype='float32', name='input'):
    """tf.keras.layers.Input(shape=[H,W,3], batch_size=...) (reference code/yolo.py:82-84).  dtype='uint8': the model takes
    the decoded image BYTES [B,H,W,3] (already of the network's size) and the network-entry kernel applies the x / 255 of
    tf.io.decode_image(dtype=float32) (code/yolo.py:106) itself - the float32 batch is never materialised."""
    if dtype not in ('float32', 'uint8'):
        raise ValueError('Input dtype must be float32 or uint8, not %r' % (dtype,))
    t = Tensor(shape, None, name)
    t.batch_size = batch_size
    t.dtype = dtype
    return t


def _same_out(n, s):
    return -(-n // s)


class Layer:
    _counters = {}

    def __init__(self, name=None):
        if name is None:
            base = type(self).__name__.lower()
            i = Layer._counters.get(base, 0)
            Layer._counters[base] = i + 1
            name = base if i == 0 else '%s_%d' % (base, i)
        self.name = name


class Conv2D(Layer):
    def __init__(self, filters, kernel_size=1, strides=1, padding='same', use_bias=True, name=None, **_):
        super().__init__(name)
        self.filters = int(filters)
        self.k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else int(kernel_size)
        self.s = strides[0] if isinstance(strides, (tuple, list)) else int(strides)
        self.padding, self.use_bias = padding, use_bias

    def __call__(self, x):
        h, w, c = x.shape
        if self.padding != 'same':
            raise ValueError('only padding="same" convolutions occur on the detection path')
        params = {self.name + '/kernel': (self.k, self.k, c, self.filters)}
        if self.use_bias:
            params[self.name + '/bias'] = (self.filters,)
        return Node('conv2d', [x], (_same_out(h, self.s), _same_out(w, self.s), self.filters),
                    dict(k=self.k, stride=self.s, use_bias=self.use_bias), params, self.name).output


class DepthwiseConv2D(Layer):
    def __init__(self, kernel_size, strides=1, padding='same', use_bias=True, name=None, **_):
        super().__init__(name)
        self.k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else int(kernel_size)
        self.s = strides[0] if isinstance(strides, (tuple, list)) else int(strides)
        if padding != 'same' or use_bias:
            raise ValueError('depthwise layers on the detection path are SAME-padded and bias-free')

    def __call__(self, x):
        h, w, c = x.shape
        return Node('depthwise', [x], (_same_out(h, self.s), _same_out(w, self.s), c),
                    dict(k=self.k, stride=self.s), {self.name + '/depthwise_kernel': (self.k, self.k, c)},
                    self.name).output


class BatchNormalization(Layer):
    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, name=None, **_):
        super().__init__(name)
        self.epsilon = float(epsilon)

    def __call__(self, x):
        c = x.shape[2]
        params = {self.name + '/' + p: (c,) for p in ('gamma', 'beta', 'moving_mean', 'moving_variance')}
        return Node('batchnorm', [x], x.shape, dict(epsilon=self.epsilon), params, self.name).output


class _Act(Layer):
    kind = None

    def __call__(self, x):
        return Node('act', [x], x.shape, dict(kind=self.kind), None, self.name).output


class ReLU(_Act):
    def __init__(self, max_value=None, name=None):
        super().__init__(name)
        if max_value != 6.:
            raise ValueError('only ReLU(6.) occurs on the detection path')
        self.kind = 'relu6'


class Swish(_Act):
    """reference code/yolo3/efficientnet.py:327-331."""
    kind = 'swish'


class Activation(_Act):
    def __init__(self, activation, name=None):
        super().__init__(name)
        if activation != 'sigmoid':
            raise ValueError('only Activation("sigmoid") occurs on the detection path')
        self.kind = 'sigmoid'


class LeakyReLU(_Act):
    """Darknet's LeakyReLU(0.1) (reference code/yolo3/darknet.py:17-23); no live caller."""
    kind = 'leaky'


class Add(Layer):
    def __call__(self, xs):
        a, b = xs
        assert a.shape == b.shape, (a, b)
        return Node('add', [a, b], a.shape, None, None, self.name).output


class Multiply(Layer):
    def __call__(self, xs):
        a, b = xs
        big = a if a.shape[0] * a.shape[1] >= b.shape[0] * b.shape[1] else b
        return Node('multiply', [a, b], big.shape, None, None, self.name).output


class Concatenate(Layer):
    def __call__(self, xs):
        h, w = xs[0].shape[:2]
        assert all(x.shape[:2] == (h, w) for x in xs), xs
        return Node('concat', list(xs), (h, w, sum(x.shape[2] for x in xs)), None, None, self.name).output


class UpSampling2D(Layer):
    def __call__(self, x):
        h, w, c = x.shape
        return Node('upsample2', [x], (2 * h, 2 * w, c), None, None, self.name).output


class MaxPooling2D(Layer):
    def __init__(self, pool_size=(2, 2), name=None):
        super().__init__(name)
        self.s = pool_size[0] if isinstance(pool_size, (tuple, list)) else int(pool_size)

    def __call__(self, x):
        h, w, c = x.shape
        return Node('maxpool', [x], (h // self.s, w // self.s, c), dict(size=self.s), None, self.name).output


class Mean(Layer):
    """reference code/yolo3/efficientnet.py:391-403 (spatial mean, keepdims)."""

    def __init__(self, spatial_dims=(1, 2), name=None):
        super().__init__(name)

    def __call__(self, x):
        return Node('mean', [x], (1, 1, x.shape[2]), None, None, self.name).output


class WeightedSum(Layer):
    """reference code/yolo3/model.py:117-137."""

    def __call__(self, xs):
        assert len(xs) == 4 and all(x.shape == xs[0].shape for x in xs), xs
        return Node('wsum', list(xs), xs[0].shape, None, {self.name + '/alpha': (4,)}, self.name).output


class Reshape5(Layer):
    """The output Lambda of model.py:325-340: [B,G,G,A*(C+5)] -> [B,G,G,A,C+5] (a view)."""

    def __init__(self, num_anchors, name=None):
        super().__init__(name)
        self.a = num_anchors

    def __call__(self, x):
        n = Node('reshape5', [x], x.shape, dict(num_anchors=self.a), None, self.name)
        return n.output


def reset_names():
    Layer._counters.clear()
