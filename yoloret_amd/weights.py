"""Synthetic (random-init) parameters for a compiled ``Model``.

The reference's checkpoints are not available (reference .MISSING_LARGE_BLOBS) and there is no
network, so benches and demos run on seeded random weights of the right architecture.  Values
depend only on (seed, recipe, parameter name, shape) - the recipe is documented in DESIGN.md
(and restated independently by the test oracle, which must produce identical values):
  'survey'      kernels N(0, 2/fan_in); BN gamma, var ~U(.5,1.5), beta, mean ~N(0,.1^2)
  'conditioned' like 'survey' but N(0, 1/fan_in) for activation-free 1x1 convs and BN gamma,
                var ~U(.8,1.2) (keeps fp32 rounding noise from being amplified ~1e3x)
  biases ~N(0,.1^2); WeightedSum alpha ~U(.5,1.5).
"""
import zlib

import numpy as np


def _linear_conv(layer):
    return layer.endswith(('project', '_y', 'se_expand')) or layer.startswith('rfcr_b')


def synthetic_weights(model, seed=1234, recipe='survey'):
    """-> {parameter name: float32 array} for every parameter of ``model`` (see Model.param_shapes)."""
    if recipe not in ('survey', 'conditioned'):
        raise ValueError('unknown recipe %r' % (recipe,))
    lo, hi = (0.5, 1.5) if recipe == 'survey' else (0.8, 1.2)
    out = {}
    for name, shape in model.param_shapes.items():
        rng = np.random.default_rng([int(seed), zlib.crc32(name.encode())])
        layer, kind = name.rsplit('/', 1)
        if kind == 'kernel':
            k, _, cin, _ = shape
            gain = 1.0 if (recipe == 'conditioned' and _linear_conv(layer)) else 2.0
            v = rng.normal(0, np.sqrt(gain / (k * k * cin)), shape)
        elif kind == 'depthwise_kernel':
            v = rng.normal(0, np.sqrt(2.0 / (shape[0] * shape[1])), shape)
        elif kind in ('gamma', 'moving_variance'):
            v = rng.uniform(lo, hi, shape)
        elif kind in ('beta', 'moving_mean', 'bias'):
            v = rng.normal(0, 0.1, shape)
        elif kind == 'alpha':
            v = rng.uniform(0.5, 1.5, shape)
        else:
            raise ValueError('unknown parameter kind %r' % (name,))
        out[name] = v.astype(np.float32)
    return out


def synthetic_images(batch, h, w, seed=20240416):
    """Uniform [0,1) NHWC float32 - the range of decoded images (reference code/yolo.py:106)."""
    return np.random.default_rng(seed).random((batch, h, w, 3), dtype=np.float32)
