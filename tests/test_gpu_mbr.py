"""Row-walking register-chained fused inverted-residual block (YR_OP_MBR, csrc/mbr.hip: expand 1x1 + BN + ReLU6 ->
DW 3x3 s1|s2 + BN + ReLU6 -> project 1x1 + BN (+ residual), float32 on the fp32 matrix pipe) against the three oracle ops
composed, through yr_op_run.  Reference: MobileNetV2 blocks [3P] via code/yolo3/override.py:290-341."""
import zlib

import numpy as np
import pytest
import torch

from oracle import nn
from tests.util import assert_close, from_dev, to_dev

pytestmark = pytest.mark.gpu

CASES = [
    # (h, w, cin, cexp, cout, stride, residual, nw, segs)
    (16, 16, 16, 96, 24, 2, False, 0, 0),      # MobileNetV2 x0.75 block_1 shape
    (104, 104, 16, 96, 24, 2, False, 3, 0),    # ... many strips / segments, three waves per workgroup
    (31, 45, 16, 96, 24, 2, False, 2, 3),      # odd sizes (pad 1/1), ragged strips and segments
    (13, 13, 24, 144, 24, 1, True, 0, 0),      # block_2 (+add): cin = 16 + 8 (the two-step tail chunk)
    (52, 52, 24, 144, 32, 2, False, 0, 0),     # MobileNetV2 x1.4 block_1 shape
    (52, 52, 24, 144, 24, 2, False, 0, 0),     # x0.75 block_3
    (27, 27, 24, 144, 48, 2, False, 0, 2),     # x0.75 block_6, odd size
    (30, 44, 32, 192, 32, 1, True, 0, 0),      # block_4/5, ragged strips
    (30, 44, 32, 192, 32, 1, True, 0, 5),      # ... forced segments
    (52, 52, 32, 192, 48, 2, False, 0, 0),     # block_6
    (26, 26, 48, 288, 48, 1, True, 6, 0),      # block_7..9
    (26, 26, 48, 288, 48, 1, True, 8, 2),      # ... eight waves, uneven tile shares (3,3,2,2,2,2,2,2)
    (9, 7, 48, 288, 48, 1, True, 8, 1),        # tiny map
    (26, 26, 48, 288, 72, 1, False, 6, 0),     # block_10
    (26, 26, 48, 288, 72, 1, False, 8, 0),
]


def _reference(x, we, se, he, wd, sd, hd, wp, sp, hp, s, residual):
    t = nn.relu6((nn.pointwise(x, we) * se + he).astype(np.float32))
    t = nn.relu6((nn.depthwise(t, wd, s, 'same') * sd + hd).astype(np.float32))
    ref = (nn.pointwise(t, wp) * sp + hp).astype(np.float32)
    return ref + x if residual else ref


def make_block(case, dev, b=2, seed=None, split=False):
    from yoloret_amd import runtime as rt
    from yoloret_amd.compiler import mbr_pack, mbs_pack
    h, w, cin, cexp, cout, s, residual, nw, segs = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()) if seed is None else seed)
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    we = (rng.standard_normal((cin, cexp)) * np.sqrt(2.0 / cin)).astype(np.float32)
    se, he = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    wd = (rng.standard_normal((3, 3, cexp)) * np.sqrt(2.0 / 9)).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    wp = (rng.standard_normal((cexp, cout)) * np.sqrt(1.0 / cexp)).astype(np.float32)
    sp, hp = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.3, cout).astype(np.float32)
    packed = (mbs_pack(we.T, se, he, wd.reshape(9, cexp), sd, hd, wp.T, sp, hp, nw) if split else
              mbr_pack(we.T, se, he, wd.reshape(9, cexp), sd, hd, wp.T, sp, hp))
    keep = [torch.from_numpy(np.ascontiguousarray(a).ravel()).to(dev) for a in packed]
    xd = to_dev(x, dev)
    ho, wo = (h + s - 1) // s, (w + s - 1) // s
    op = rt.new_op(rt.OP_MBR, 'relu6')
    op.dtype = op.out_dtype = rt.dtype_id('f32')
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc, op.se_reduced = ho, wo, cin, cout, 3 | int(split) << 7 | nw << 8 | segs << 16, s, 1, cexp
    op.src[0] = rt.make_src(xd, c=cin)
    op.wgt, op.wgt2, op.b2 = [k.data_ptr() for k in keep]
    if residual:
        op.res, op.res_ld = xd.data_ptr(), xd.shape[3]
    out = torch.full((b, ho, wo, cout), float('nan'), dtype=torch.float32, device=dev)
    op.out, op.out_ld = out.data_ptr(), cout
    return op, out, (x, we, se, he, wd, sd, hd, wp, sp, hp, s, residual), keep + [xd]


@pytest.mark.parametrize('case', CASES, ids=[str(i) for i in range(len(CASES))])
def test_mbr(dev, case):
    from yoloret_amd import runtime as rt
    op, out, params, keep = make_block(case, dev)
    ref = _reference(*params)
    rt.run_op(op, 2)
    torch.cuda.synchronize()
    assert_close(from_dev(out), ref, 5e-5, 'mbr %s' % (case,))


MBS_CASES = [
    # the SPLIT form (k bit 7): both 1x1 convolutions on the 16-bit matrix pipe, float32 operands as two float16 planes each
    (16, 16, 16, 96, 24, 2, False, 3, 0),
    (104, 104, 16, 96, 24, 2, False, 2, 0),    # block_1 as shipped: two waves (three tiles each: one full pair + an odd tile)
    (31, 45, 16, 96, 24, 2, False, 3, 3),      # odd sizes, ragged strips and segments
    (13, 13, 24, 144, 24, 1, True, 3, 0),      # block_2 (+add)
    (52, 40, 24, 144, 24, 1, True, 3, 2),
    (52, 52, 24, 144, 24, 2, False, 3, 0),     # block_3
    (27, 27, 24, 144, 48, 2, False, 3, 2),     # block_6, odd size
    (26, 26, 48, 288, 48, 1, True, 6, 0),      # block_7..9: two K = 32 steps
    (9, 7, 48, 288, 48, 1, True, 6, 1),        # tiny map
    (26, 26, 48, 288, 72, 1, False, 6, 0),     # block_10
]


@pytest.mark.parametrize('case', MBS_CASES, ids=[str(i) for i in range(len(MBS_CASES))])
def test_mbr_split_form(dev, case):
    """Same bar as the float32-MFMA form (5e-5 of the float64-free oracle composition): two float16 planes per operand and three
    products keep 22 bits of every factor.  Inputs with a wide dynamic range (x 100, x 0.01 per image) included."""
    from yoloret_amd import runtime as rt
    op, out, params, keep = make_block(case, dev, split=True)
    ref = _reference(*params)
    rt.run_op(op, 2)
    torch.cuda.synchronize()
    assert_close(from_dev(out), ref, 5e-5, 'mbr split %s' % (case,))


MBE_CASES = [
    # (h, w, cin, cexp, stride, segs): YR_OP_MBE = expand 1x1 + BN + ReLU6 -> depthwise 3x3 + BN + ReLU6, the map stored
    (26, 26, 72, 432, 1, 0),      # MobileNetV2 x0.75 block_11, 12
    (26, 26, 72, 432, 2, 0),      # block_13
    (13, 13, 120, 720, 1, 0),     # block_14, 15 (cin = 112 + 8: the two-step tail chunk)
    (13, 13, 120, 720, 1, 2),
    (9, 21, 72, 112, 1, 1),       # ragged: 7 tiles = 2 full groups of 3 + a short one, two strips
    (15, 11, 48, 288, 2, 3),      # odd sizes (pad 1 / 1), segments
    (32, 32, 88, 528, 1, 0),      # MobileNetV2 x1.4 block_7..9
    (16, 16, 224, 1344, 1, 0),    # x1.4 block_14, 15: 84 tiles, one per wave
    (32, 32, 136, 816, 2, 0),     # x1.4 block_13
]


@pytest.mark.parametrize('split', [False, True], ids=['f32', 'split'])
@pytest.mark.parametrize('case', MBE_CASES, ids=[str(i) for i in range(len(MBE_CASES))])
def test_mbe(dev, case, split):
    from yoloret_amd import runtime as rt
    from yoloret_amd.compiler import mbr_pack, mbs_pack
    h, w, cin, cexp, s, segs = case
    if split and cin == 224:
        pytest.skip('the split form is not built for 224 block inputs (compiler.MBS_MBE_CINS)')
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    b = 2
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    we = (rng.standard_normal((cin, cexp)) * np.sqrt(2.0 / cin)).astype(np.float32)
    se, he = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    wd = (rng.standard_normal((3, 3, cexp)) * np.sqrt(2.0 / 9)).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    t = nn.relu6((nn.pointwise(x, we) * se + he).astype(np.float32))
    ref = nn.relu6((nn.depthwise(t, wd, s, 'same') * sd + hd).astype(np.float32))
    wa, tab, _ = (mbs_pack(we.T, se, he, wd.reshape(9, cexp), sd, hd, None, None, None, 0) if split else
                  mbr_pack(we.T, se, he, wd.reshape(9, cexp), sd, hd, None, None, None))
    keep = [torch.from_numpy(np.ascontiguousarray(a).ravel()).to(dev) for a in (wa, tab)]
    xd = to_dev(x, dev)
    ho, wo = ref.shape[1], ref.shape[2]
    op = rt.new_op(rt.OP_MBE, 'relu6')
    op.dtype = op.out_dtype = rt.dtype_id('f32')
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = ho, wo, cin, cexp, 3 | int(split) << 7 | segs << 16, s, 1
    op.src[0] = rt.make_src(xd, c=cin)
    op.wgt, op.wgt2 = [k.data_ptr() for k in keep]
    out = torch.full((b, ho, wo, cexp), float('nan'), dtype=torch.float32, device=dev)
    op.out, op.out_ld = out.data_ptr(), cexp
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_close(from_dev(out), ref, 5e-5, 'mbe %s' % (case,))


MBK_CASES = [
    # the WEIGHT-STREAMING form (k bits 6, 7; csrc/mbk.hip): (h, w, cin, cexp, cout, stride, residual, rows per wave, waves per workgroup)
    (26, 26, 48, 288, 48, 1, True, 2, 8),      # MobileNetV2 x0.75 block_7..9: two K = 32 steps of the expand conv, 9 pairs
    (19, 33, 48, 288, 72, 1, False, 2, 8),     # block_10 (no residual), ragged
    (26, 26, 72, 432, 72, 1, True, 2, 8),      # block_11, 12: two strips, two row segments, 27 tiles (an odd last pair)
    (13, 13, 72, 432, 72, 1, True, 2, 8),      # one segment (13 <= 16 rows), one strip
    (37, 30, 72, 432, 72, 1, True, 2, 8),      # three segments, three strips (ragged)
    (26, 26, 72, 432, 120, 2, False, 2, 8),    # block_13: even size (pad 0 / 1), two segments
    (27, 31, 72, 432, 120, 2, False, 2, 8),    # odd size (pad 1 / 1): the first wave's first row lies above the image
    (9, 7, 72, 432, 120, 2, False, 2, 8),      # tiny map, one segment
    (13, 13, 120, 720, 120, 1, True, 1, 8),    # block_14, 15: one row per wave, two segments
    (20, 17, 120, 720, 120, 1, True, 1, 8),    # four segments, two strips
    (5, 5, 120, 720, 120, 1, True, 1, 8),      # fewer rows than waves
]


def make_block_k(case, dev, b=2, seed=None):
    from yoloret_amd import runtime as rt
    from yoloret_amd.compiler import mbk_pack
    h, w, cin, cexp, cout, s, residual, rows, nw = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()) if seed is None else seed)
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    we = (rng.standard_normal((cin, cexp)) * np.sqrt(2.0 / cin)).astype(np.float32)
    se, he = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    wd = (rng.standard_normal((3, 3, cexp)) * np.sqrt(2.0 / 9)).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    wp = (rng.standard_normal((cexp, cout)) * np.sqrt(1.0 / cexp)).astype(np.float32)
    sp, hp = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.3, cout).astype(np.float32)
    packed = mbk_pack(we.T, se, he, wd.reshape(9, cexp), sd, hd, wp.T, sp, hp)
    keep = [torch.from_numpy(np.ascontiguousarray(a).ravel()).to(dev) for a in packed]
    xd = to_dev(x, dev)
    ho, wo = (h + s - 1) // s, (w + s - 1) // s
    op = rt.new_op(rt.OP_MBR, 'relu6')
    op.dtype = op.out_dtype = rt.dtype_id('f32')
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc, op.se_reduced = ho, wo, cin, cout, 3 | 0xc0 | nw << 8 | rows << 16, s, 1, cexp
    op.src[0] = rt.make_src(xd, c=cin)
    op.wgt, op.b2 = [k.data_ptr() for k in keep]
    if residual:
        op.res, op.res_ld = xd.data_ptr(), xd.shape[3]
    out = torch.full((b, ho, wo, cout), float('nan'), dtype=torch.float32, device=dev)
    op.out, op.out_ld = out.data_ptr(), cout
    return op, out, (x, we, se, he, wd, sd, hd, wp, sp, hp, s, residual), keep + [xd]


@pytest.mark.parametrize('case', MBK_CASES, ids=[str(i) for i in range(len(MBK_CASES))])
def test_mbr_streaming_form(dev, case):
    """The whole block in one launch where the fragments do not fit a CU's register file: same bar as the other forms (5e-5 of the
    oracle composition); every output element written exactly once (the buffer starts as NaN)."""
    from yoloret_amd import runtime as rt
    op, out, params, keep = make_block_k(case, dev)
    ref = _reference(*params)
    rt.run_op(op, 2)
    torch.cuda.synchronize()
    assert_close(from_dev(out), ref, 5e-5, 'mbr streaming %s' % (case,))


def test_mbr_streaming_form_is_batch_independent(dev):
    """Image i of a batch of 5 equals image i run alone, bit for bit (the sums are grouped by the map's shape only)."""
    from yoloret_amd import runtime as rt
    case = MBK_CASES[2]
    op, out, params, keep = make_block_k(case, dev, b=5, seed=3)
    rt.run_op(op, 5)
    torch.cuda.synchronize()
    full = from_dev(out).copy()
    x = params[0]
    for i in (0, 4):
        op1, out1, _, keep1 = make_block_k(case, dev, b=5, seed=3)
        xd1 = to_dev(x[i:i + 1], dev)
        op1.src[0] = rt.make_src(xd1, c=case[2])
        op1.res = xd1.data_ptr()
        rt.run_op(op1, 1)
        torch.cuda.synchronize()
        assert np.array_equal(from_dev(out1)[0], full[i])
