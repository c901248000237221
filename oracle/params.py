"""Seeded synthetic parameters, keyed by layer name (test infrastructure).

There are no checkpoints (reference .MISSING_LARGE_BLOBS), so both the oracle
and the product draw weights from the same recipe.  Values depend only on
(seed, name, shape, kind) - never on creation order - so the oracle's graph
walk and the product's graph builder can ask in any order and agree.

Recipe (with non-trivial SE biases / WeightedSum alpha so those
paths are exercised): conv & depthwise kernels N(0, 2/fan_in); BN gamma~U(.5,1.5),
beta~N(0,.1^2), mean~N(0,.1^2), var~U(.5,1.5); bias~N(0,.1^2); alpha~U(.5,1.5).
Layouts are Keras' (SURVEY.md A.5): Conv2D [kh,kw,Cin,Cout], DepthwiseConv2D
[kh,kw,C,1] (stored squeezed [kh,kw,C]), BN 4x[C].
"""
import zlib

import numpy as np


class ParamStore:
    def __init__(self, seed=1234):
        self.seed = int(seed)
        self.values = {}

    def _rng(self, name):
        return np.random.default_rng([self.seed, zlib.crc32(name.encode())])

    def _get(self, name, shape, fn):
        v = self.values.get(name)
        if v is None:
            v = fn(self._rng(name)).astype(np.float32)
            self.values[name] = v
        assert tuple(v.shape) == tuple(shape), (name, v.shape, shape)
        return v

    def conv(self, name, k, cin, cout):
        std = np.sqrt(2.0 / (k * k * cin))
        return self._get(name + '/kernel', (k, k, cin, cout),
                         lambda r: r.normal(0, std, (k, k, cin, cout)))

    def dw(self, name, k, c):
        std = np.sqrt(2.0 / (k * k))
        return self._get(name + '/depthwise_kernel', (k, k, c),
                         lambda r: r.normal(0, std, (k, k, c)))

    def bias(self, name, c):
        return self._get(name + '/bias', (c,), lambda r: r.normal(0, 0.1, (c,)))

    def bn(self, name, c):
        g = self._get(name + '/gamma', (c,), lambda r: r.uniform(0.5, 1.5, (c,)))
        b = self._get(name + '/beta', (c,), lambda r: r.normal(0, 0.1, (c,)))
        m = self._get(name + '/moving_mean', (c,), lambda r: r.normal(0, 0.1, (c,)))
        v = self._get(name + '/moving_variance', (c,), lambda r: r.uniform(0.5, 1.5, (c,)))
        return g, b, m, v

    def alpha(self, name, n=4):
        return self._get(name + '/alpha', (n,), lambda r: r.uniform(0.5, 1.5, (n,)))


def synthetic_images(batch, h, w, seed=20240416):
    """Uniform [0,1) NHWC float32, the range tf.io.decode_image(dtype=float32)
    produces (reference code/yolo.py:106)."""
    return np.random.default_rng(seed).random((batch, h, w, 3), dtype=np.float32)
