"""Shared helpers for the parity tests (host<->device staging, tolerances)."""
import numpy as np
import torch

ANCHORS = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326],
                   np.float32).reshape(-1, 2)


def round_up(v, m):
    return (v + m - 1) // m * m


def to_dev(a, dev, ld=None, fill=np.nan):
    """[B,H,W,C] numpy -> device tensor [B,H,W,ld]; pad channels are filled with NaN by default so
    that any kernel that lets padding leak into results fails loudly."""
    a = np.asarray(a, np.float32)
    c = a.shape[-1]
    ld = round_up(c, 4) if ld is None else ld
    if ld != c:
        p = np.full(a.shape[:-1] + (ld,), fill, np.float32)
        p[..., :c] = a
        a = p
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def from_dev(t, c=None):
    a = t.detach().cpu().numpy()
    return a if c is None else a[..., :c]


def assert_close(got, ref, tol=1e-4, what=''):
    """|got-ref| <= tol*max(1,|ref|) elementwise (SURVEY.md H3 form of the 1e-4 logits bar)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert np.isfinite(got).all(), '%s: non-finite values in result' % what
    m = float(err.max()) if err.size else 0.0
    assert m <= tol, '%s: max scaled error %.3e > %.1e at %s' % (what, m, tol, np.unravel_index(err.argmax(), err.shape))
    return m
