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


# ---- 16-bit storage (bfloat16 / float16) helpers shared by the reduced-precision tests
REL = {'bf16': 2.0 ** -8, 'f16': 2.0 ** -11}     # half an ulp, relative (normal range)
TINY = {'bf16': 1e-30, 'f16': 2.0 ** -25}        # half an ulp in the subnormal range of float16


def _rt():
    from yoloret_amd import runtime as rt
    return rt


def q16(a, dt):
    """float32 array rounded to the 16-bit type and widened back."""
    rt = _rt()
    a = np.ascontiguousarray(a, np.float32)
    return rt.from_bits16(rt.to_bits16(a, dt), dt).reshape(a.shape)


def to_dev16(a, dev, dt, ld=None, poison=True):
    """[..., C] float32 (values representable in dt) -> device tensor [..., ld] of the 16-bit type; pad channels NaN."""
    rt = _rt()
    a = np.asarray(a, np.float32)
    c = a.shape[-1]
    ld = round_up(c, 8) if ld is None else ld
    if ld != c:
        p = np.full(a.shape[:-1] + (ld,), np.nan if poison else 0.0, np.float32)
        p[..., :c] = a
        a = p
    bits = rt.to_bits16(a, dt).reshape(a.shape)
    t = torch.from_numpy(bits.view(np.int16)).to(dev)
    return t.view(rt.TORCH_DTYPE[rt.dtype_id(dt)])


def from_dev16(t, dt, c=None):
    rt = _rt()
    a = rt.from_bits16(t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16), dt).reshape(tuple(t.shape))
    return a if c is None else a[..., :c]


def assert_rounded_once(got, ref64, dt, what, slack=2e-5):
    """|got - ref| <= half ulp_dt(ref) (x1.02) + slack*max(1,|ref|): one rounding of a float32-accurate value."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref64, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), '%s: non-finite values' % what
    tol = 1.02 * REL[dt] * np.abs(ref) + TINY[dt] + slack * np.maximum(1.0, np.abs(ref))
    bad = np.abs(got - ref) > tol
    assert not bad.any(), '%s: %d of %d beyond half an ulp; worst |d|/tol = %.2f' % (
        what, bad.sum(), bad.size, float((np.abs(got - ref) / tol).max()))


def entry_on_matrix_pipe(model):
    """Whether the plan's network entry takes image and stem kernel as 16-bit MFMA operands (stemblock_h.hip: a STEMBLOCK op in
    the matrix-pipe layout, which carries BN `scale` rows, or the stem + depthwise entry asked for in its matrix-pipe form, k = 3 | 1 << 8) - what QuantStore(round_entry=...) must emulate for this plan."""
    rt = _rt()
    op = model.plan.ops[0]
    return model.plan.dtype != 0 and op.kind == rt.OP_STEMBLOCK and ('scale' in op.params or ((op.k >> 8) & 0xff) == 1)
