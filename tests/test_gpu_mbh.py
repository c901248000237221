"""Fused inverted-residual block on 16-bit MFMA (YR_OP_MBH: expand 1x1 + BN + act -> depthwise KxK s1|s2 + BN + act ->
project 1x1 + BN (+ residual), 16-bit in / out) against the three oracle ops composed in float64, through yr_op_run.

Rounding model of the kernel (what the reference below restates): input, expand and project weights are 16-bit values;
the expanded tensor stays float32 on chip; the depthwise result is rounded ONCE to the 16-bit type (it is the MFMA
operand of the projection); the block output is rounded once at the store.  A depthwise value that sits on a rounding
boundary may round the other way in float32 than in the float64 reference - one operand ulp in one of Cexp products -
hence the extra slack (one ulp of a value up to 6 times a projection weight) on top of the output's own half ulp."""
import zlib

import numpy as np
import pytest
import torch

from oracle import nn
from tests.util import assert_rounded_once, from_dev16, q16, round_up, to_dev16

pytestmark = pytest.mark.gpu

CASES = [
    # (h, w, cin, cexp, cout, k, stride, residual, act, forced tile (th, tw) or None)
    (16, 16, 16, 96, 24, 3, 2, False, 'relu6', None),      # MobileNetV2 block_1 shape
    (104, 104, 16, 96, 24, 3, 2, False, 'relu6', None),    # ... at a size with many tiles
    (26, 26, 24, 144, 24, 3, 1, True, 'relu6', None),      # block_2 (+add)
    (52, 52, 24, 144, 32, 3, 2, False, 'relu6', None),     # block_3
    (30, 44, 32, 192, 32, 3, 1, True, 'relu6', None),      # block_4/5, ragged tiles
    (26, 26, 48, 288, 48, 3, 1, True, 'relu6', None),      # block_7..9: two k-steps of the expand GEMM
    (26, 26, 72, 432, 72, 3, 1, True, 'relu6', None),      # block_11/12: three k-steps (weights not hoisted)
    (26, 26, 72, 432, 120, 3, 2, False, 'relu6', None),    # block_13
    (13, 13, 120, 720, 120, 3, 1, True, 'relu6', None),    # block_14/15: four k-steps, four cout pairs
    (9, 7, 24, 144, 32, 3, 2, False, 'relu6', None),       # odd size, stride 2
    (15, 17, 40, 240, 40, 5, 1, True, 'relu6', None),      # EfficientNet-lite stage 3 (k5)
    (16, 16, 24, 144, 40, 5, 2, False, 'relu6', None),     # ... its stride-2 entry block
    (13, 13, 80, 480, 112, 5, 1, False, 'relu6', None),    # stage 5 entry (k5, cin != cout)
    (8, 8, 16, 100, 20, 3, 1, False, 'relu6', None),       # expanded width not a multiple of 32, cout not of 8
    (21, 9, 14, 50, 14, 3, 1, True, 'relu6', None),        # cin / cout with pad lanes (NaN-filled)
    (26, 26, 24, 144, 24, 3, 1, True, 'relu6', (13, 12)),  # forced large tile (two pixel groups per wave)
    (26, 26, 48, 288, 48, 3, 1, True, 'relu6', (7, 8)),    # forced odd tile (14 runs: ragged groups)
    (13, 13, 120, 720, 120, 3, 1, True, 'relu6', (3, 4)),  # tiny tile: a single partial group
    (20, 20, 32, 192, 48, 3, 1, False, 'swish', None),     # generic activation path
    (22, 18, 24, 144, 32, 3, 2, False, 'swish', None),     # the narrow stride-2 front block (mbn_h.hip) with swish, two passes
    (12, 30, 32, 192, 16, 3, 2, False, 'relu6', None),     # ... 32 inputs, 192 expanded channels (two full passes), one cout tile
    (20, 24, 32, 192, 48, 5, 2, False, 'relu6', None),     # EfficientNet-lite3 stage 3 entry (k5 s2; tile-wise chained form: 4 waves x 3 tiles)
    (23, 17, 48, 288, 48, 5, 1, True, 'relu6', None),      # ... stage 3 (k5, two input chunks, 6 waves x 3 tiles), odd sizes
    (27, 31, 24, 144, 40, 5, 2, False, 'relu6', None),     # lite0 stage 3 entry at an odd size (pad 2/2)
]


def _act(t, act):
    return {'relu6': nn.relu6, 'swish': nn.swish}[act](t)


def _chained_built(cin, cexp, cout, k, act='relu6'):
    """mbxr_h.hip: yr_mbhr_built - the shapes the register-chained whole-block kernels are instantiated for: the tile-pair form
    (mbhr_kernel, 3x3) and the tile-wise form (mbhq_kernel: 3x3 / 5x5, ReLU6, MBHQ_SHAPES)."""
    if not (cin % 8 == 0 and cin <= 64 and cexp % 16 == 0 and cout % 4 == 0):
        return False
    nc, to = round_up(cin, 32) // 32, (cout + 15) // 16
    if k == 3 and cexp <= 256 and cout <= 80 and (nc, to, (cexp // 16 + 1) // 2) in {(1, 2, 3), (1, 2, 6), (1, 3, 6), (2, 5, 8), (2, 3, 8)}:
        return True
    return act == 'relu6' and (k, nc, to, cexp // 16) in {(5, 1, 3, 9), (5, 1, 3, 12), (5, 2, 3, 18), (5, 2, 3, 15), (3, 1, 2, 9)}


@pytest.mark.parametrize('form', ['lds', 'chained'])
@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('case', CASES, ids=[str(i) for i in range(len(CASES))])
def test_mbh(dev, case, dt, form):
    """form 'lds': the LDS-tiled kernels (mbh.hip, mbn_h.hip; forced tile 254 = their own tile choice); 'chained': the
    row-walking register-chained kernel (mbxr_h.hip: mbhr_kernel; forced tile 255, one or three row segments) where built."""
    from yoloret_amd import runtime as rt
    h, w, cin, cexp, cout, k, s, residual, act, tile = case
    if form == 'chained':
        if tile is not None or not _chained_built(cin, cexp, cout, k, act):
            pytest.skip('the register-chained form is not built for this shape')
        tile = (255, 3 if h > 20 else 0)
    elif tile is None:
        tile = (254, 0)
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    b = 2
    x = q16(rng.standard_normal((b, h, w, cin)), dt)
    we = q16(rng.standard_normal((cin, cexp)) * np.sqrt(2.0 / cin), dt)
    se, he = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    t = _act(nn.pointwise(x.astype(np.float64), we.astype(np.float64)) * se + he, act)
    mbn = k == 3 and s == 2 and cin <= 32 and cin % 8 == 0 and cexp <= 192 and cout <= 32 and not residual and tile == (254, 0)
    if mbn:
        # the narrow stride-2 block at the network's front runs on mbn_h.hip, which keeps the WHOLE expanded halo tile on chip
        # in the 16-bit type (the unfused chain's rounding point, oracle/model.py P.store); an expanded value on a rounding
        # boundary may go the other way than in float64: twice the slack
        t = q16(t.astype(np.float32), dt).astype(np.float64)
    wd = (rng.standard_normal((k, k, cexp)) * np.sqrt(2.0 / (k * k))).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    t = _act(nn.depthwise(t, wd.astype(np.float64), s, 'same') * sd + hd, act)
    t = q16(t.astype(np.float32), dt).astype(np.float64)            # the projection's MFMA operand
    wp = q16(rng.standard_normal((cexp, cout)) * np.sqrt(1.0 / cexp), dt)
    sp, hp = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.3, cout).astype(np.float32)
    ref = nn.pointwise(t, wp.astype(np.float64)) * sp + hp
    if residual:
        ref = ref + x
    cexp_p, kp, ldo = round_up(cexp, 32), round_up(cin, 32), round_up(cout, 8)
    wet = np.zeros((cexp_p, kp), np.float32); wet[:cexp, :cin] = we.T
    dwp = np.zeros((k * k + 4, cexp_p), np.float32)      # depthwise taps | dw BN scale | shift | expand BN scale | shift
    dwp[:k * k, :cexp], dwp[k * k, :cexp], dwp[k * k + 1, :cexp] = wd.reshape(k * k, cexp), sd, hd
    dwp[k * k + 2, :cexp], dwp[k * k + 3, :cexp] = se, he
    wpt = np.zeros((cout, cexp_p), np.float32); wpt[:, :cexp] = wp.T
    pb = np.zeros((2, ldo), np.float32); pb[0, :cout], pb[1, :cout] = sp, hp

    def dev16(a):
        return torch.from_numpy(rt.to_bits16(a, dt).view(np.int16).reshape(a.shape)).to(dev)

    def dev32(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    keep = [dev16(wet), dev32(dwp), dev16(wpt), dev32(pb)]
    xd = to_dev16(x, dev, dt)
    did = rt.dtype_id(dt)
    op = rt.new_op(rt.OP_MBH, act)
    op.dtype = op.out_dtype = did
    ho, wo = ref.shape[1], ref.shape[2]
    op.h, op.w, op.cin, op.cout, op.stride, op.nsrc, op.se_reduced = ho, wo, cin, cout, s, 1, cexp
    op.k = k | ((tile[0] << 8) | (tile[1] << 16) if tile else 0)
    op.src[0] = rt.make_src(xd, c=cin)
    op.wgt, op.wgt2, op.b1, op.b2 = [t_.data_ptr() for t_ in keep]
    if residual:
        op.res, op.res_ld = xd.data_ptr(), xd.shape[3]
    out = torch.full((b, ho, wo, ldo), float('nan'), dtype=rt.TORCH_DTYPE[did], device=dev)
    op.out, op.out_ld = out.data_ptr(), ldo
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_rounded_once(from_dev16(out, dt, cout), ref, dt, 'mbh %s %s' % (dt, case), slack={'bf16': 4e-3, 'f16': 5e-4}[dt] * (2 if mbn else 1))


def test_mbh_rejects_what_it_is_not_built_for(dev):
    from yoloret_amd import runtime as rt
    x = torch.zeros((1, 8, 8, 160), dtype=torch.bfloat16, device=dev)
    out = torch.zeros((1, 8, 8, 160), dtype=torch.bfloat16, device=dev)
    op = rt.new_op(rt.OP_MBH, 'relu6')
    op.dtype = op.out_dtype = rt.DTYPE['bf16']
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc, op.se_reduced = 8, 8, 160, 160, 3, 1, 1, 960
    op.src[0] = rt.make_src(x, c=160)
    op.wgt = op.wgt2 = op.b1 = op.b2 = x.data_ptr()
    op.out, op.out_ld = out.data_ptr(), 160
    with pytest.raises(rt.YoloretHipError, match='widths out of range'):
        rt.run_op(op, 1)
    op.dtype = op.out_dtype = 0
    with pytest.raises(rt.YoloretHipError, match='16-bit'):
        rt.run_op(op, 1)


MBX_CASES = [
    # (h, w, cin, cexp, k, stride, act, forced tile or None)   - EfficientNet MBConv blocks with squeeze-excite
    (52, 52, 16, 96, 3, 2, 'swish', None),       # B0 stage 2 entry (expanded width: 3 chunks)
    (26, 26, 24, 144, 3, 1, 'swish', None),      # stage 2 (144 = 4.5 chunks: masked tail)
    (30, 22, 24, 144, 5, 2, 'swish', None),      # stage 3 entry (k5 s2), ragged tiles
    (26, 26, 40, 240, 5, 1, 'swish', None),      # stage 3
    (13, 13, 80, 480, 3, 1, 'swish', None),      # stage 4
    (13, 13, 112, 672, 5, 1, 'swish', None),     # stage 5: four k-steps of the expand GEMM
    (14, 14, 112, 672, 5, 2, 'swish', None),     # stage 6 entry
    (27, 31, 16, 96, 3, 2, 'swish', None),       # stride 2 at odd sizes (pad 1/1): the paired-row walk, an odd last pair
    (23, 29, 24, 144, 5, 2, 'swish', None),      # ... 5x5 (pad 2/2)
    (37, 18, 32, 192, 3, 2, 'relu6', None),      # ... several segments
    (26, 26, 32, 192, 3, 1, 'relu6', (13, 12)),  # two pixel groups per wave
    (9, 11, 14, 52, 3, 1, 'swish', (4, 8)),      # pad lanes in the input (NaN-filled), width not a multiple of 8
    (20, 20, 48, 288, 5, 1, 'swish', (8, 16)),
]


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('with_sums', [True, False])
@pytest.mark.parametrize('case', MBX_CASES, ids=[str(i) for i in range(len(MBX_CASES))])
def test_mbx(dev, case, with_sums, dt):
    """YR_OP_MBX: expand 1x1 + BN + act -> depthwise + BN + act, the depthwise map stored once (16-bit) and its per-tile
    channel sums (the squeeze of squeeze-excite) written beside it."""
    from yoloret_amd import runtime as rt
    h, w, cin, cexp, k, s, act, tile = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    b = 2
    x = q16(rng.standard_normal((b, h, w, cin)), dt)
    we = q16(rng.standard_normal((cin, cexp)) * np.sqrt(2.0 / cin), dt)
    se, he = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    t = _act(nn.pointwise(x.astype(np.float64), we.astype(np.float64)) * se + he, act)
    wd = (rng.standard_normal((k, k, cexp)) * np.sqrt(2.0 / (k * k))).astype(np.float32)
    sd, hd = rng.uniform(0.5, 1.5, cexp).astype(np.float32), rng.normal(0, 0.3, cexp).astype(np.float32)
    ref = _act(nn.depthwise(t, wd.astype(np.float64), s, 'same') * sd + hd, act)
    cexp_p, kp, ldo = round_up(cexp, 32), round_up(cin, 32), round_up(cexp, 8)
    wet = np.zeros((cexp_p, kp), np.float32); wet[:cexp, :cin] = we.T
    dwp = np.zeros((k * k + 4, cexp_p), np.float32)
    dwp[:k * k, :cexp], dwp[k * k, :cexp], dwp[k * k + 1, :cexp] = wd.reshape(k * k, cexp), sd, hd
    dwp[k * k + 2, :cexp], dwp[k * k + 3, :cexp] = se, he
    keep = [torch.from_numpy(rt.to_bits16(wet, dt).view(np.int16).reshape(wet.shape)).to(dev), torch.from_numpy(dwp).to(dev)]
    xd = to_dev16(x, dev, dt)
    did = rt.dtype_id(dt)
    ho, wo = ref.shape[1], ref.shape[2]
    rows = ((ho + 3) // 4) * ((wo + 7) // 8)
    op = rt.new_op(rt.OP_MBX, act)
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.stride, op.nsrc = ho, wo, cin, cexp, s, 1
    op.k = k | ((tile[0] << 8) | (tile[1] << 16) if tile else 0)
    op.src[0] = rt.make_src(xd, c=cin)
    op.wgt, op.wgt2 = keep[0].data_ptr(), keep[1].data_ptr()
    out = torch.full((b, ho, wo, ldo), float('nan'), dtype=rt.TORCH_DTYPE[did], device=dev)
    op.out, op.out_ld = out.data_ptr(), ldo
    part = torch.full((b, rows, ldo), float('nan'), dtype=torch.float32, device=dev)
    if with_sums:
        op.gate, op.gate_ld, op.se_reduced = part.data_ptr(), ldo, rows
    rt.run_op(op, b)
    torch.cuda.synchronize()
    got = from_dev16(out, dt, cexp)
    # hardware exp2 / reciprocal in the swish (about 1e-6 relative) on top of the half ulp of the one rounding
    assert_rounded_once(got, ref, dt, 'mbx %s %s' % (dt, case), slack=5e-5)
    if ldo > cexp:
        assert np.all(from_dev16(out, dt, ldo)[..., cexp:] == 0), 'pad channels of the stored map are zero'
    if with_sums:
        p = part.cpu().numpy()
        assert np.all(np.isfinite(p)), 'every row of the partial-sum buffer is written (unused rows zeroed)'
        sums = p.astype(np.float64).sum(axis=1)[:, :cexp]
        want = got.astype(np.float64).sum(axis=(1, 2))               # sums of the STORED values
        np.testing.assert_allclose(sums, want, rtol=2e-5, atol=2e-4 * np.sqrt(ho * wo))
    else:
        assert torch.isnan(part).all()


def test_mbx_rejects_a_short_sum_buffer(dev):
    from yoloret_amd import runtime as rt
    x = torch.zeros((1, 26, 26, 24), dtype=torch.bfloat16, device=dev)
    out = torch.zeros((1, 26, 26, 144), dtype=torch.bfloat16, device=dev)
    part = torch.zeros((1, 4, 144), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_MBX, 'swish')
    op.dtype = op.out_dtype = rt.DTYPE['bf16']
    op.h, op.w, op.cin, op.cout, op.stride, op.nsrc = 26, 26, 24, 144, 1, 1
    op.k = 3 | (4 << 8) | (8 << 16)
    op.src[0] = rt.make_src(x, c=24)
    op.wgt = op.wgt2 = x.data_ptr()
    op.out, op.out_ld = out.data_ptr(), 144
    op.gate, op.gate_ld, op.se_reduced = part.data_ptr(), 144, 4
    with pytest.raises(rt.YoloretHipError, match='exceed the 4 rows'):
        rt.run_op(op, 1)
