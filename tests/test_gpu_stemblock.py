"""Fused network-entry kernel (stem 3x3 s2 + BN + act -> DW 3x3 + BN + act -> project 1x1 + BN) against the
three oracle ops composed, through yr_op_run."""
import zlib

import numpy as np
import pytest
import torch

from oracle import nn
from tests.util import assert_close, assert_rounded_once, from_dev, from_dev16, round_up

pytestmark = pytest.mark.gpu


def _vec(a, dev, n=None):
    a = np.asarray(a, np.float32).ravel()
    if n is not None and n != a.size:
        p = np.zeros(n, np.float32)
        p[:a.size] = a
        a = p
    return torch.from_numpy(a).to(dev)


CASES = [((64, 64), 24, 16, 'relu6'), ((416, 416), 24, 16, 'relu6'), ((32, 96), 48, 24, 'relu6'),
         ((30, 22), 32, 16, 'swish'), ((50, 34), 40, 24, 'relu6')]


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('hw,c1,cout,act', CASES)
def test_stemblock(dev, hw, c1, cout, act, dt):
    """dt: element type of the OUTPUT map (the image is float32 in every plan)."""
    from yoloret_amd import runtime as rt
    rng = np.random.default_rng(zlib.crc32(str((hw, c1, cout)).encode()))
    b = 2
    x = rng.random((b, hw[0], hw[1], 3), dtype=np.float32)
    actf = {'relu6': nn.relu6, 'swish': nn.swish}[act]
    ws = (rng.standard_normal((3, 3, 3, c1)) * np.sqrt(2.0 / 27)).astype(np.float32)
    ss, hs = rng.uniform(0.5, 1.5, c1).astype(np.float32), rng.normal(0, 0.3, c1).astype(np.float32)
    wd = (rng.standard_normal((3, 3, c1)) * np.sqrt(2.0 / 9)).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, c1).astype(np.float32), rng.normal(0, 0.3, c1).astype(np.float32)
    wp = (rng.standard_normal((c1, cout)) * np.sqrt(1.0 / c1)).astype(np.float32)
    sp, hp = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.3, cout).astype(np.float32)
    t = actf((nn.conv2d(x, ws, 2, 'same') * ss + hs).astype(np.float32))
    t = actf((nn.depthwise(t, wd, 1, 'same') * sd + hd).astype(np.float32))
    ref = (nn.pointwise(t, wp) * sp + hp).astype(np.float32)
    c1p, ldo, cop = round_up(c1, 4), round_up(cout, 4), round_up(cout, 8)

    def per_pair(w, scale, shift):    # [taps][c1] + BN -> [c1p/2][taps x 2, times the scale | 1 1 | shift 2]  (include/yoloret_hip.h)
        rows = np.zeros((w.shape[0] + 2, c1p), np.float32)
        rows[:-2, :c1], rows[-2, :c1], rows[-1, :c1] = (w * scale[None]).astype(np.float32), 1.0, shift   # BN scale folded into the taps
        return np.ascontiguousarray(rows.reshape(-1, c1p // 2, 2).transpose(1, 0, 2))
    wpp = np.zeros((c1p, cop), np.float32)
    wpp[:c1, :cout] = wp
    pb = np.zeros((2, cop), np.float32)
    pb[0, :cout], pb[1, :cout] = sp, hp
    keep = [_vec(per_pair(ws.reshape(27, c1), ss, hs), dev), _vec(per_pair(wd.reshape(9, c1), sd, hd), dev),
            _vec(wpp, dev), _vec(pb, dev)]
    xd = torch.from_numpy(x).to(dev)
    if dt != 'f32':
        ldo = round_up(cout, 8)
    out = torch.full((b, ref.shape[1], ref.shape[2], ldo), float('nan'), dtype=rt.TORCH_DTYPE[rt.dtype_id(dt)], device=dev)
    op = rt.new_op(rt.OP_STEMBLOCK, act)
    op.dtype = op.out_dtype = rt.dtype_id(dt)
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc, op.se_reduced = ref.shape[1], ref.shape[2], 3, cout, 3, 2, 1, c1
    op.src[0] = rt.make_src(xd, c=3, ld=3)
    op.wgt, op.wgt2, op.b1, op.b2 = [k.data_ptr() for k in keep]
    op.out, op.out_ld = out.data_ptr(), ldo
    rt.run_op(op, b)
    torch.cuda.synchronize()
    if dt == 'f32':
        assert_close(from_dev(out, cout), ref, 3e-5, 'stemblock')
    else:
        assert_rounded_once(from_dev16(out, dt, cout), ref, dt, 'stemblock %s' % dt, slack=3e-5)


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f16', 'bf16-mfma', 'f16-mfma'])
@pytest.mark.parametrize('with_sums', [True, False])
@pytest.mark.parametrize('hw,c1,act', [((64, 64), 32, 'swish'), ((62, 90), 40, 'swish'), ((30, 22), 48, 'relu6'), ((416, 416), 32, 'swish')])
def test_stem_plus_depthwise(dev, hw, c1, act, with_sums, dt):
    """YR_OP_STEMBLOCK without a projection (the entry of the squeeze-excite EfficientNets, efficientnet.py:636-645 + the
    first block's depthwise conv): the depthwise map, one rounding at the store, plus the per-tile channel sums of the
    STORED values (the squeeze), every row of the sum buffer written.  '-mfma': the matrix-pipe form (k = 3 | 1 << 8,
    mbxr_h.hip: stemxr_kernel): image and (BN-scaled) stem kernel are 16-bit MFMA operands - the reference rounds them."""
    from yoloret_amd import runtime as rt
    mfma = dt.endswith('-mfma')
    dt = dt.split('-')[0]
    rng = np.random.default_rng(zlib.crc32(str((hw, c1)).encode()))
    b = 2
    x = rng.random((b, hw[0], hw[1], 3), dtype=np.float32)
    actf = {'relu6': nn.relu6, 'swish': nn.swish}[act]
    ws = (rng.standard_normal((3, 3, 3, c1)) * np.sqrt(2.0 / 27)).astype(np.float32)
    ss, hs = rng.uniform(0.5, 1.5, c1).astype(np.float32), rng.normal(0, 0.3, c1).astype(np.float32)
    wd = (rng.standard_normal((3, 3, c1)) * np.sqrt(2.0 / 9)).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, c1).astype(np.float32), rng.normal(0, 0.3, c1).astype(np.float32)
    if mfma:
        from tests.util import q16
        t = actf(nn.conv2d(q16(x, dt).astype(np.float64), q16((ws * ss).astype(np.float32), dt).astype(np.float64), 2, 'same') + hs)
    else:
        t = actf(nn.conv2d(x.astype(np.float64), ws.astype(np.float64), 2, 'same') * ss + hs)
    ref = actf(nn.depthwise(t, wd.astype(np.float64), 1, 'same') * sd + hd)
    c1p = round_up(c1, 4)

    def per_pair(w, scale, shift):
        rows = np.zeros((w.shape[0] + 2, c1p), np.float32)
        rows[:-2, :c1], rows[-2, :c1], rows[-1, :c1] = (w * scale[None]).astype(np.float32), 1.0, shift
        return np.ascontiguousarray(rows.reshape(-1, c1p // 2, 2).transpose(1, 0, 2))
    keep = [_vec(per_pair(ws.reshape(27, c1), ss, hs), dev), _vec(per_pair(wd.reshape(9, c1), sd, hd), dev)]
    xd = torch.from_numpy(x).to(dev)
    did = rt.dtype_id(dt)
    ho, wo = ref.shape[1], ref.shape[2]
    ldo = round_up(c1, 8)
    out = torch.full((b, ho, wo, ldo), float('nan'), dtype=rt.TORCH_DTYPE[did], device=dev)
    rows = ((ho + 13) // 14) * ((wo + 13) // 14)
    part = torch.full((b, rows, ldo), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_STEMBLOCK, act)
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc, op.se_reduced = ho, wo, 3, c1, 3 | (1 << 8 if mfma else 0), 2, 1, c1
    op.src[0] = rt.make_src(xd, c=3, ld=3)
    op.wgt, op.wgt2 = keep[0].data_ptr(), keep[1].data_ptr()
    op.out, op.out_ld = out.data_ptr(), ldo
    if with_sums:
        op.gate, op.gate_ld = part.data_ptr(), ldo
    rt.run_op(op, b)
    torch.cuda.synchronize()
    got = from_dev(out, c1) if dt == 'f32' else from_dev16(out, dt, c1)
    if dt == 'f32':
        assert_close(got, ref.astype(np.float32), 3e-5, 'stem+dw')
    else:
        assert_rounded_once(got, ref, dt, 'stem+dw %s' % dt, slack=5e-5)
    if with_sums:
        p = part.cpu().numpy()[..., :c1]
        assert np.isfinite(p).all(), 'every row of the sum buffer is written'
        want = np.asarray(got, np.float64).sum(axis=(1, 2))
        np.testing.assert_allclose(p.astype(np.float64).sum(axis=1), want, rtol=2e-5, atol=2e-4 * np.sqrt(ho * wo))
    else:
        assert torch.isnan(part).all()


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('u8', [False, True])
@pytest.mark.parametrize('hw,c1,cout,act', [((64, 64), 24, 16, 'relu6'), ((416, 416), 32, 16, 'relu6'), ((32, 96), 48, 24, 'relu6'),
                                            ((30, 22), 32, 16, 'swish'), ((50, 34), 40, 24, 'relu6')])
def test_stemblock_matrix_pipe(dev, hw, c1, cout, act, dt, u8):
    """The 16-bit plans' form of the same op (stemblock_h.hip: stem and projection on the MFMA pipe; taken when op.scale is
    set): image, stem kernel and projection kernel are 16-bit operands, the stem and depthwise outputs are rounded to the
    plan's type on their way (LDS / MFMA operand).  Against the NumPy chain with the same roundings; what is left is the
    accumulation order and the occasional intermediate that rounds the other way (a 16-bit ulp of one operand of a 1 x 1)."""
    from oracle.params import round16
    from yoloret_amd import runtime as rt
    rng = np.random.default_rng(zlib.crc32(str((hw, c1, cout, 'h')).encode()))
    b = 2
    if u8:
        xi = rng.integers(0, 256, (b, hw[0], hw[1], 3), dtype=np.uint8)
        x = xi.astype(np.float32)            # exact in either type; the / 255 multiplies the BN scale
        in_scale = np.float32(1.0 / 255.0)
    else:
        x = round16(rng.random((b, hw[0], hw[1], 3), dtype=np.float32), dt)
        xi = None
        in_scale = np.float32(1.0)
    actf = {'relu6': nn.relu6, 'swish': nn.swish}[act]
    ws = round16((rng.standard_normal((3, 3, 3, c1)) * np.sqrt(2.0 / 27)).astype(np.float32), dt)
    ss, hs = rng.uniform(0.5, 1.5, c1).astype(np.float32), rng.normal(0, 0.3, c1).astype(np.float32)
    wd = (rng.standard_normal((3, 3, c1)) * np.sqrt(2.0 / 9)).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, c1).astype(np.float32), rng.normal(0, 0.3, c1).astype(np.float32)
    wp = round16((rng.standard_normal((c1, cout)) * np.sqrt(1.0 / c1)).astype(np.float32), dt)
    sp, hp = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.3, cout).astype(np.float32)
    t = round16(actf((nn.conv2d(x, ws, 2, 'same') * (ss * in_scale) + hs).astype(np.float32)), dt)
    t = round16(actf((nn.depthwise(t, (wd * sd[None, None]).astype(np.float32), 1, 'same') + hd).astype(np.float32)), dt)
    ref = (nn.pointwise(t, wp) * sp + hp).astype(np.float32)
    c1m, com, ldo = round_up(c1, 32), round_up(cout, 16), round_up(cout, 8)
    korder = [g * 9 + i for g in range(3) for i in range(8)] + [i * 9 + 8 for i in range(3)]
    wsm = np.zeros((c1m, 32), np.float32)
    wsm[:c1, :27] = ws.reshape(27, c1)[korder].T
    dwr = np.zeros((10, c1m), np.float32)
    dwr[:9, :c1], dwr[9, :c1] = (wd.reshape(9, c1) * sd[None]).astype(np.float32), hd
    wpm = np.zeros((com, c1m), np.float32)
    wpm[:cout, :c1] = wp.T
    pb = np.zeros((2, com), np.float32)
    pb[0, :cout], pb[1, :cout] = sp, hp
    did = rt.dtype_id(dt)

    def dev16(a):
        bits = rt.to_bits16(np.ascontiguousarray(a, np.float32).ravel(), did)
        return torch.from_numpy(bits.view(np.int16).copy()).to(dev)
    keep = [dev16(wsm), _vec(ss, dev, c1m), _vec(hs, dev, c1m), _vec(dwr, dev), dev16(wpm), _vec(pb, dev)]
    xd = torch.from_numpy(xi if u8 else x).to(dev)
    out = torch.full((b, ref.shape[1], ref.shape[2], ldo), float('nan'), dtype=rt.TORCH_DTYPE[did], device=dev)
    op = rt.new_op(rt.OP_STEMBLOCK, act)
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc, op.se_reduced = ref.shape[1], ref.shape[2], 3, cout, 3, 2, 1, c1
    op.src[0] = rt.make_src(xd, c=3, ld=3)
    op.wgt, op.scale, op.shift, op.wgt2, op.b1, op.b2 = [k.data_ptr() for k in keep]
    op.out, op.out_ld = out.data_ptr(), ldo
    rt.run_op(op, b)
    torch.cuda.synchronize()
    got = from_dev16(out, dt, cout)
    err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    mx, mean = (3e-2, 4e-3) if dt == 'bf16' else (4e-3, 5e-4)
    assert np.isfinite(got).all() and err.max() <= mx and err.mean() <= mean, (err.max(), err.mean(), np.unravel_index(err.argmax(), err.shape))
