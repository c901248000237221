"""NumPy NHWC restatement of the TF/Keras kernels the reference's detection
path executes.  Test infrastructure only (see
``oracle/__init__.py``).  Every function works in the dtype of its input
(float32 for parity, float64 for the "distance to exact" report).

Third-party semantics restated here (TensorFlow is not in /root/reference):
  * padding='same'  (C.1): out=ceil(in/s), extra pixel goes bottom/right.
  * BatchNormalization inference (C.2): gamma*(x-mean)/sqrt(var+eps)+beta.
  * ReLU(6.), Swish=x*sigmoid(x) (reference code/yolo3/efficientnet.py:327-331).
  * UpSampling2D() nearest 2x, MaxPooling2D((s,s)) VALID (C.4).
"""
import math

import numpy as np


def same_pad(in_size, k, s):
    """TF 'SAME' padding amounts (before, after) and output size."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return before, total - before, out


def _pad_hw(x, k, stride, padding):
    if padding == 'same':
        pt, pb, ho = same_pad(x.shape[1], k, stride)
        pl, pr, wo = same_pad(x.shape[2], k, stride)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    elif padding == 'valid':
        ho = (x.shape[1] - k) // stride + 1
        wo = (x.shape[2] - k) // stride + 1
    else:
        raise ValueError(padding)
    return x, ho, wo


def conv2d(x, w, stride=1, padding='same'):
    """Conv2D, x [B,H,W,Cin], w HWIO [k,k,Cin,Cout] (Keras kernel layout)."""
    k = w.shape[0]
    assert w.shape[1] == k
    if k == 1 and stride == 1:
        return pointwise(x, w[0, 0])
    xp, ho, wo = _pad_hw(x, k, stride, padding)
    b = x.shape[0]
    out = np.zeros((b, ho, wo, w.shape[3]), dtype=x.dtype)
    for dy in range(k):
        for dx in range(k):
            patch = xp[:, dy:dy + (ho - 1) * stride + 1:stride,
                       dx:dx + (wo - 1) * stride + 1:stride, :]
            out += patch.reshape(-1, w.shape[2]).dot(
                w[dy, dx].astype(x.dtype)).reshape(b, ho, wo, -1)
    return out


def pointwise(x, w):
    """1x1 Conv2D, w [Cin,Cout]."""
    sh = x.shape
    return x.reshape(-1, sh[-1]).dot(w.astype(x.dtype)).reshape(*sh[:-1], w.shape[1])


def depthwise(x, w, stride=1, padding='same'):
    """DepthwiseConv2D (multiplier 1), w [k,k,C] (Keras [k,k,C,1] squeezed)."""
    k = w.shape[0]
    xp, ho, wo = _pad_hw(x, k, stride, padding)
    out = np.zeros((x.shape[0], ho, wo, x.shape[3]), dtype=x.dtype)
    for dy in range(k):
        for dx in range(k):
            out += xp[:, dy:dy + (ho - 1) * stride + 1:stride,
                      dx:dx + (wo - 1) * stride + 1:stride, :] * w[dy, dx].astype(x.dtype)
    return out


def batchnorm(x, gamma, beta, mean, var, eps=1e-3):
    t = x.dtype
    inv = gamma.astype(t) / np.sqrt(var.astype(t) + t.type(eps))
    return (x - mean.astype(t)) * inv + beta.astype(t)


def relu6(x):
    return np.minimum(np.maximum(x, 0), 6).astype(x.dtype)


def sigmoid(x):
    one = x.dtype.type(1)
    return one / (one + np.exp(-x))


def swish(x):
    return x * sigmoid(x)


def leaky_relu(x, alpha=0.1):
    return np.where(x >= 0, x, x * x.dtype.type(alpha))


def upsample2(x):
    return np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)


def maxpool(x, s):
    b, h, w, c = x.shape
    ho, wo = h // s, w // s
    return x[:, :ho * s, :wo * s, :].reshape(b, ho, s, wo, s, c).max(axis=(2, 4))


def concat(xs):
    return np.concatenate(xs, axis=-1)


def mean_hw(x):
    return x.mean(axis=(1, 2), keepdims=True, dtype=x.dtype)


def macs_conv(ho, wo, k, cin, cout):
    return ho * wo * k * k * cin * cout
