"""Seeded synthetic parameters, keyed by layer name (test infrastructure).

There are no checkpoints (reference .MISSING_LARGE_BLOBS), so both the oracle
and the product draw weights from the same recipe.  Values depend only on
(seed, recipe, name, shape, kind) - never on creation order - so the oracle's graph
walk and the product's graph builder can ask in any order and agree.

Two recipes:
  'survey'      SURVEY.md 8(d): conv & depthwise kernels N(0, 2/fan_in); BN gamma~U(.5,1.5),
                beta~N(0,.1^2), mean~N(0,.1^2), var~U(.5,1.5).  Every linear (activation-free)
                1x1 conv doubles the variance, so the random network amplifies perturbations
                ~1e3x from input to logits: two correct fp32 implementations (NumPy fp32 vs
                NumPy fp64!) already differ by ~1e-3.  Used for the bench workload and for the
                "no worse than CPU fp32, measured against fp64" parity test.
  'conditioned' variance preserving: N(0, 1/fan_in) for the linear 1x1 convs (project, y, RFCR
                taps, SE expand), N(0, 2/fan_in) otherwise; BN gamma, var ~U(.8,1.2).  fp32 noise
                stays ~1e-5, so the strict 1e-4 logits bar is meaningful.  Default for parity tests.
SE biases ~N(0,.1^2) and WeightedSum alpha~U(.5,1.5) (non-trivial so those paths are exercised).
Layouts are Keras' (SURVEY.md A.5): Conv2D [kh,kw,Cin,Cout], DepthwiseConv2D
[kh,kw,C,1] (stored squeezed [kh,kw,C]), BN 4x[C].
"""
import zlib

import numpy as np

RECIPES = ('survey', 'conditioned')


def is_linear_conv(name):
    """1x1 convs not followed by an activation on the detection path."""
    return name.endswith(('project', '_y', 'se_expand')) or name.startswith('rfcr_b')


class ParamStore:
    def __init__(self, seed=1234, recipe='conditioned'):
        assert recipe in RECIPES
        self.seed = int(seed)
        self.recipe = recipe
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
        gain = 1.0 if (self.recipe == 'conditioned' and is_linear_conv(name)) else 2.0
        std = np.sqrt(gain / (k * k * cin))
        return self._get(name + '/kernel', (k, k, cin, cout),
                         lambda r: r.normal(0, std, (k, k, cin, cout)))

    def dw(self, name, k, c):
        std = np.sqrt(2.0 / (k * k))
        return self._get(name + '/depthwise_kernel', (k, k, c),
                         lambda r: r.normal(0, std, (k, k, c)))

    def bias(self, name, c):
        return self._get(name + '/bias', (c,), lambda r: r.normal(0, 0.1, (c,)))

    def bn(self, name, c):
        lo, hi = (0.5, 1.5) if self.recipe == 'survey' else (0.8, 1.2)
        g = self._get(name + '/gamma', (c,), lambda r: r.uniform(lo, hi, (c,)))
        b = self._get(name + '/beta', (c,), lambda r: r.normal(0, 0.1, (c,)))
        m = self._get(name + '/moving_mean', (c,), lambda r: r.normal(0, 0.1, (c,)))
        v = self._get(name + '/moving_variance', (c,), lambda r: r.uniform(lo, hi, (c,)))
        return g, b, m, v

    def alpha(self, name, n=4):
        return self._get(name + '/alpha', (n,), lambda r: r.uniform(0.5, 1.5, (n,)))

    @staticmethod
    def store(x):
        """Where a fused op writes its result to memory (oracle/model.py): float32 keeps the value as it is."""
        return x

    @staticmethod
    def entry(x, w):
        """The image and the stem kernel as the network's first convolution takes them: float32 as they are."""
        return x, w


def round16(a, dtype):
    """float32 -> nearest-even bfloat16 / float16 -> float32 (NumPy restatement of the storage rounding)."""
    a = np.ascontiguousarray(a, np.float32)
    if dtype in ('f16', 'float16'):
        return a.astype(np.float16).astype(np.float32)
    assert dtype in ('bf16', 'bfloat16'), dtype
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xffffffff).astype(np.uint32).view(np.float32).reshape(a.shape)


class QuantStore(ParamStore):
    """Emulation of a 16-bit storage plan (BASELINE.json configs 3 / 5) on the float32 NumPy oracle: the SAME
    parameters as ParamStore(seed, recipe) (``values`` stays float32 - it is what the product is given), but
    activations are rounded to `dtype` wherever a fused op stores them and the 1x1-convolution kernels (the MFMA
    operands; not the SE FCs, which stay float32) are rounded once.  Arithmetic in between is float32."""

    def __init__(self, seed=1234, recipe='conditioned', dtype='bf16', round_entry=False):
        """round_entry: whether the image and the stem kernel are rounded too - a property of the plan UNDER TEST (its network
        entry runs on the 16-bit matrix pipe or on the float32 pipe), so the test passes it in (tests/util.py:
        entry_on_matrix_pipe(model)); the oracle's graph code knows nothing of the product's kernels.  Everything else is the
        UNFUSED placement: one rounding at every convolution / gate / merge output of the graph, whatever the product fuses
        (a fused kernel keeps intermediates in float32: it can only be more accurate than this, never less)."""
        super().__init__(seed, recipe)
        self.dtype = dtype
        self.round_entry = bool(round_entry)

    def store(self, x):
        return round16(x, self.dtype) if x.dtype == np.float32 else x

    def conv(self, name, k, cin, cout):
        w = super().conv(name, k, cin, cout)
        if k == 1 and not name.endswith(('_se_reduce', '_se_expand')):
            return round16(w, self.dtype)
        return w

    def entry(self, x, w):
        """Image and stem kernel as the network entry takes them: rounded like every 1x1 convolution's operands when the plan
        under test runs its stem on the matrix pipe (round_entry; decoded uint8 pixels are exact in either type), else as is."""
        return (round16(x, self.dtype), round16(w, self.dtype)) if self.round_entry else (x, w)


def synthetic_images(batch, h, w, seed=20240416):
    """Uniform [0,1) NHWC float32, the range tf.io.decode_image(dtype=float32)
    produces (reference code/yolo.py:106)."""
    return np.random.default_rng(seed).random((batch, h, w, 3), dtype=np.float32)
