"""Lane-per-pixel fused inverted-residual block (YR_OP_MBLANE: expand 1x1 + BN + act -> DW 3x3 s1|s2 + BN + act ->
project 1x1 + BN (+ residual)) against the three oracle ops composed, through yr_op_run."""
import zlib

import numpy as np
import pytest
import torch

from oracle import nn
from tests.util import assert_close, assert_rounded_once, from_dev, from_dev16, q16, round_up, to_dev, to_dev16

pytestmark = pytest.mark.gpu

CASES = [
    # (h, w, cin, cexp, cout, stride, residual, act)
    (16, 16, 16, 96, 24, 2, False, 'relu6'),     # MobileNetV2 block_1 shape
    (104, 104, 16, 96, 24, 2, False, 'relu6'),   # ... at a size with many tiles
    (13, 13, 24, 144, 24, 1, True, 'relu6'),     # block_2 (+add)
    (52, 52, 24, 144, 32, 2, False, 'relu6'),    # block_3
    (30, 44, 32, 192, 32, 1, True, 'relu6'),     # block_4/5, ragged tiles
    (9, 7, 24, 144, 32, 2, False, 'relu6'),      # odd size, stride 2 (pad 1/1)
    (15, 17, 32, 192, 40, 1, False, 'swish'),    # SE-free MBConv flavour
    (8, 8, 16, 100, 20, 1, False, 'relu6'),      # expanded width not a multiple of 16, cout not a multiple of 4
    (21, 9, 14, 50, 14, 1, True, 'relu6'),       # cin / cout with pad lanes (NaN-filled by to_dev)
    (52, 52, 24, 144, 48, 2, False, 'relu6'),    # block_6: wide projection (second row loaded after the first)
    (20, 20, 32, 192, 48, 1, False, 'relu6'),    # widest built shape
]


def _act(t, act):
    return {'relu6': nn.relu6, 'swish': nn.swish}[act](t)


def _pairs(rows, scale, shift, e2):
    cexp = rows.shape[1]
    full = np.zeros((rows.shape[0] + 2, e2), np.float32)
    full[:-2, :cexp], full[-2, :cexp], full[-1, :cexp] = (rows * scale[None]).astype(np.float32), 1.0, shift   # BN scale folded (yoloret_hip.h)
    return np.ascontiguousarray(full.reshape(-1, e2 // 2, 2).transpose(1, 0, 2))


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('case', CASES, ids=[str(i) for i in range(len(CASES))])
def test_mblane(dev, case, dt):
    """dt: element type of the block's input and output (16-bit: the block computes in float32 from registers and
    rounds once at the store; the residual is the 16-bit input widened)."""
    from yoloret_amd import runtime as rt
    h, w, cin, cexp, cout, s, residual, act = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    b = 2
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    if dt != 'f32':
        x = q16(x, dt)
    we = (rng.standard_normal((cin, cexp)) * np.sqrt(2.0 / cin)).astype(np.float32)
    se, he = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    t = _act((nn.pointwise(x, we) * se + he).astype(np.float32), act)
    wd = (rng.standard_normal((3, 3, cexp)) * np.sqrt(2.0 / 9)).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    t = _act((nn.depthwise(t, wd, s, 'same') * sd + hd).astype(np.float32), act)
    wp = (rng.standard_normal((cexp, cout)) * np.sqrt(1.0 / cexp)).astype(np.float32)
    sp, hp = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.3, cout).astype(np.float32)
    ref = (nn.pointwise(t, wp) * sp + hp).astype(np.float32)
    if residual:
        ref = ref + x
    cinp, cop, ldo = round_up(cin, 4), round_up(cout, 8), round_up(cout, 4)
    e2 = 2 * round_up((cexp + 1) // 2, 8)
    wep = np.zeros((cinp, cexp), np.float32)
    wep[:cin] = we
    wpp = np.zeros((e2, cop), np.float32)
    wpp[:cexp, :cout] = wp
    pb = np.zeros((2, cop), np.float32)
    pb[0, :cout], pb[1, :cout] = sp, hp
    keep = [torch.from_numpy(np.ascontiguousarray(a).ravel()).to(dev)
            for a in (_pairs(wep, se, he, e2), _pairs(wd.reshape(9, cexp), sd, hd, e2), wpp, pb)]
    xd = to_dev(x, dev) if dt == 'f32' else to_dev16(x, dev, dt)
    op = rt.new_op(rt.OP_MBLANE, act)
    op.dtype = op.out_dtype = rt.dtype_id(dt)
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc, op.se_reduced = ref.shape[1], ref.shape[2], cin, cout, 3, s, 1, cexp
    op.src[0] = rt.make_src(xd, c=cin)
    op.wgt, op.wgt2, op.b1, op.b2 = [k.data_ptr() for k in keep]
    if residual:
        op.res, op.res_ld = xd.data_ptr(), xd.shape[3]
    if dt != 'f32':
        ldo = round_up(cout, 8)
    out = torch.full((b, ref.shape[1], ref.shape[2], ldo), float('nan'), dtype=rt.TORCH_DTYPE[rt.dtype_id(dt)], device=dev)
    op.out, op.out_ld = out.data_ptr(), ldo
    rt.run_op(op, b)
    torch.cuda.synchronize()
    if dt == 'f32':
        assert_close(from_dev(out, cout), ref, 5e-5, 'mblane %s' % (case,))
    else:
        assert_rounded_once(from_dev16(out, dt, cout), ref, dt, 'mblane %s %s' % (dt, case), slack=5e-5)


IDENT_CASES = [
    # (h, w, c, cout == c, residual, act): blocks WITHOUT expand conv (expand ratio 1; EfficientNet stage 1, efficientnet.py:467)
    (30, 44, 24, True, 'relu6'), (13, 13, 24, False, 'relu6'), (17, 15, 16, True, 'swish'), (9, 21, 32, True, 'relu6'),
    (160, 160, 24, True, 'relu6'), (8, 8, 22, False, 'relu6'),
]


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('case', IDENT_CASES, ids=[str(i) for i in range(len(IDENT_CASES))])
def test_mblane_without_expand(dev, case, dt):
    """wgt = NULL: depthwise 3x3 stride 1 + BN + act -> project 1x1 + BN (+ residual) on the block input itself."""
    from yoloret_amd import runtime as rt
    h, w, c, residual, act = case
    cout = c
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    b = 2
    x = rng.standard_normal((b, h, w, c)).astype(np.float32)
    if dt != 'f32':
        x = q16(x, dt)
    wd = (rng.standard_normal((3, 3, c)) * np.sqrt(2.0 / 9)).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(0, 0.3, c).astype(np.float32)
    t = _act((nn.depthwise(x, wd, 1, 'same') * sd + hd).astype(np.float32), act)
    wp = (rng.standard_normal((c, cout)) * np.sqrt(1.0 / c)).astype(np.float32)
    sp, hp = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.3, cout).astype(np.float32)
    ref = (nn.pointwise(t, wp) * sp + hp).astype(np.float32)
    if residual:
        ref = ref + x
    cop = round_up(cout, 8)
    e2 = 2 * round_up((c + 1) // 2, 8)
    wpp = np.zeros((e2, cop), np.float32)
    wpp[:c, :cout] = wp
    pb = np.zeros((2, cop), np.float32)
    pb[0, :cout], pb[1, :cout] = sp, hp
    keep = [torch.from_numpy(np.ascontiguousarray(a).ravel()).to(dev) for a in (_pairs(wd.reshape(9, c), sd, hd, e2), wpp, pb)]
    xd = to_dev(x, dev) if dt == 'f32' else to_dev16(x, dev, dt)
    op = rt.new_op(rt.OP_MBLANE, act)
    op.dtype = op.out_dtype = rt.dtype_id(dt)
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc, op.se_reduced = h, w, c, cout, 3, 1, 1, c
    op.src[0] = rt.make_src(xd, c=c)
    op.wgt2, op.b1, op.b2 = [k.data_ptr() for k in keep]      # no `wgt`: no expand conv
    if residual:
        op.res, op.res_ld = xd.data_ptr(), xd.shape[3]
    ldo = round_up(cout, 4) if dt == 'f32' else round_up(cout, 8)
    out = torch.full((b, h, w, ldo), float('nan'), dtype=rt.TORCH_DTYPE[rt.dtype_id(dt)], device=dev)
    op.out, op.out_ld = out.data_ptr(), ldo
    rt.run_op(op, b)
    torch.cuda.synchronize()
    if dt == 'f32':
        assert_close(from_dev(out, cout), ref, 5e-5, 'mblane without expand %s' % (case,))
    else:
        assert_rounded_once(from_dev16(out, dt, cout), ref, dt, 'mblane without expand %s %s' % (dt, case), slack=5e-5)


def test_mblane_rejects_unsupported_widths(dev):
    from yoloret_amd import runtime as rt
    x = torch.zeros((1, 8, 8, 64), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_MBLANE, 'relu6')
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc, op.se_reduced = 8, 8, 64, 64, 3, 1, 1, 384
    op.src[0] = rt.make_src(x, c=64)
    op.wgt = op.wgt2 = op.b1 = op.b2 = x.data_ptr()
    out = torch.zeros((1, 8, 8, 64), dtype=torch.float32, device=dev)
    op.out, op.out_ld = out.data_ptr(), 64
    with pytest.raises(rt.YoloretHipError, match='unsupported'):
        rt.run_op(op, 1)
