"""The 16-bit (bfloat16 / float16 storage) forms of the fused ops and of the whole graph - BASELINE.json configs 3
("bf16 pointwise on MFMA, fp32 decode/NMS") and 5 ("fp16 + fused RFCR upsample-concat-conv").

The reference has no reduced-precision mode (build-defined), so the bars are:
  * per op: inputs and weights exactly representable in the 16-bit type; the result must equal the float64 value of
    the same expression rounded ONCE to the 16-bit type, up to half an ulp of that type plus float32 accumulation
    noise (i.e. the kernels compute in float32 and round only at the store);
  * whole graph: measured against the float32 oracle, the HIP path must be no less accurate than a NumPy emulation of
    the same storage format (oracle.params.QuantStore: float32 arithmetic, activations rounded to the 16-bit type at
    the fused-op boundaries, 1x1-conv weights rounded once) - x1.5 slack - and the detections are compared with the
    float32 oracle's (agreement reported and bounded).  SURVEY.md H3: "bf16/fp16 configs cannot meet 1e-4: report
    error vs fp32 and detection agreement instead".
"""
import zlib

import numpy as np
import pytest
import torch

from oracle import nn
from tests.util import assert_rounded_once, from_dev16, q16, round_up, to_dev16

pytestmark = pytest.mark.gpu

DTYPES = ['bf16', 'f16']


def _rt():
    from yoloret_amd import runtime as rt
    return rt


def _dev_vec(a, dev, n=None):
    a = np.asarray(a, np.float32).ravel().copy()
    if n is not None and n != a.size:
        a = np.concatenate([a, np.zeros(n - a.size, np.float32)])
    return torch.from_numpy(a).to(dev)


def _act(x, act):
    return {'none': lambda v: v, 'relu6': nn.relu6, 'swish': nn.swish, 'sigmoid': nn.sigmoid, 'leaky': nn.leaky_relu}[act](x)


def _xform(x, xf):
    return {'identity': lambda v: v, 'up2': nn.upsample2, 'maxpool2': lambda v: nn.maxpool(v, 2),
            'maxpool4': lambda v: nn.maxpool(v, 4)}[xf](x)


def _src_dims(h, w, xf):
    return {'identity': (h, w), 'up2': (h // 2, w // 2), 'maxpool2': (h * 2, w * 2), 'maxpool4': (h * 4, w * 4)}[xf]


def run_pointwise16(dev, dt, rng, b, h, w, segs, cout, act='none', bn=True, residual=False, gate=False, out_f32=False,
                    dense_out=False, pool=False, pre=False, cfg=0, ksplit=False):
    """One 16-bit POINTWISE op through yr_op_run against float64 NumPy on the same (exactly representable) operands.
    h, w: the conv's resolution (pool=True: the output is its 2x2 max)."""
    rt = _rt()
    did = rt.dtype_id(dt)
    srcs_np, srcs_dev = [], []
    for c, xf in segs:
        sh, sw = _src_dims(h, w, xf)
        a = q16(rng.standard_normal((b, sh, sw, c)), dt)
        srcs_np.append(a)
        srcs_dev.append(to_dev16(a, dev, dt))
    cin = sum(c for c, _ in segs)
    wk = q16(rng.standard_normal((cin, cout)) * np.sqrt(2.0 / cin), dt)
    kp = sum(round_up(c, 8) for c, _ in segs)
    wt = np.zeros((cout, kp), np.float32)
    d = kb = 0
    for c, _ in segs:
        wt[:, kb:kb + c] = wk[d:d + c].T
        d += c
        kb += round_up(c, 8)
    x = nn.concat([_xform(a, xf) for a, (_, xf) in zip(srcs_np, segs)]).astype(np.float64)
    keep = []
    op = rt.new_op(rt.OP_POINTWISE, act)
    op.dtype, op.out_dtype = did, (0 if out_f32 else did)
    if gate:
        gate_np = rng.uniform(0.1, 1.0, (b, 1, 1, cin)).astype(np.float32)
        # the kernel forms x*gate in float32 and rounds the product to the operand type (efficientnet.py:435 is a tensor)
        x = q16((x.astype(np.float32) * gate_np), dt).astype(np.float64)
        g = np.full((b, round_up(cin, 8)), np.nan, np.float32)
        g[:, :cin] = gate_np.reshape(b, cin)
        gd = torch.from_numpy(g).to(dev)
        keep.append(gd)
        op.gate, op.gate_ld = gd.data_ptr(), gd.shape[1]
    ref = x.reshape(-1, cin).dot(wk.astype(np.float64)).reshape(b, h, w, cout)
    nsrc = len(segs)
    for i, (t, (c, xf)) in enumerate(zip(srcs_dev, segs)):
        op.src[i] = rt.make_src(t, c=c, xform=xf)
    if pre:   # the hoisted share of a concat conv: float32 [B, h/2, w/2, cout], added before BatchNorm
        p_np = rng.standard_normal((b, h // 2, w // 2, cout)).astype(np.float32)
        ref = ref + nn.upsample2(p_np).astype(np.float64)
        ldp = round_up(cout, 4)
        pp = np.full((b, h // 2, w // 2, ldp), np.nan, np.float32)
        pp[..., :cout] = p_np
        pd = torch.from_numpy(pp).to(dev)
        keep.append(pd)
        op.src[nsrc] = rt.make_src(pd, c=cout, xform='up2_add')
        nsrc += 1
    if bn:
        scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        shift = rng.normal(0, 0.3, cout).astype(np.float32)
        ref = ref * scale + shift
        keep += [_dev_vec(scale, dev), _dev_vec(shift, dev)]
        op.scale, op.shift = keep[-2].data_ptr(), keep[-1].data_ptr()
    ref = _act(ref, act)
    if residual:
        res_np = q16(rng.standard_normal((b, h, w, cout)), dt)
        ref = ref + res_np
        r = to_dev16(res_np, dev, dt)
        keep.append(r)
        op.res, op.res_ld = r.data_ptr(), r.shape[3]
    oh, ow = h, w
    if pool:
        ref = nn.maxpool(ref, 2)
        oh, ow = h // 2, w // 2
        op.stride = 2
    wd = torch.from_numpy(rt.to_bits16(wt, dt).view(np.int16).reshape(wt.shape)).to(dev)
    keep.append(wd)
    op.wgt = wd.data_ptr()
    op.h, op.w, op.cin, op.cout, op.nsrc = oh, ow, cin, cout, nsrc
    if out_f32:
        out_ld = cout if dense_out else round_up(cout, 4)
        out = torch.full((b, oh, ow, out_ld), float('nan'), dtype=torch.float32, device=dev)
    else:
        out_ld = round_up(cout, 8)
        out = torch.full((b, oh, ow, out_ld), float('nan'), dtype=rt.TORCH_DTYPE[did], device=dev)
    op.out, op.out_ld = out.data_ptr(), out_ld
    op.k = cfg
    if ksplit:
        op.se_reduced |= 0x20000      # the k-split form of the plans for one or two images (pointwise_h.hip: pwkh_kernel)
    rt.run_op(op, b)
    torch.cuda.synchronize()
    if out_f32:
        got = out.cpu().numpy()[..., :cout]
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        assert np.isfinite(got).all() and err.max() <= 2e-5, 'pointwise16 f32 out %s: %.3e' % (segs, err.max())
    else:
        got = from_dev16(out, dt, cout)
        assert_rounded_once(got, ref, dt, 'pointwise16 %s %s cfg %d' % (dt, segs, cfg))
    return got


PW16 = [
    # (h, w, segs, cout, act, bn, residual, gate, out_f32, dense_out, pool, pre)
    (13, 13, [(16, 'identity')], 96, 'relu6', True, False, False, False, False, False, False),
    (13, 11, [(24, 'identity')], 16, 'none', True, False, False, False, False, False, False),
    (7, 9, [(144, 'identity')], 24, 'none', True, True, False, False, False, False, False),
    (13, 13, [(720, 'identity')], 120, 'none', True, True, False, False, False, False, False),
    (26, 26, [(72, 'identity')], 432, 'swish', True, False, False, False, False, False, False),
    (13, 13, [(120, 'identity'), (96, 'maxpool2')], 512, 'relu6', True, False, False, False, False, False, False),
    (26, 26, [(256, 'up2'), (72, 'identity'), (96, 'identity')], 256, 'relu6', True, False, False, False, False, False, False),
    (12, 12, [(128, 'up2'), (24, 'identity'), (96, 'up2')], 128, 'relu6', True, False, False, False, False, False, False),
    (13, 13, [(128, 'maxpool2'), (75, 'identity')], 256, 'relu6', True, False, False, False, False, False, False),
    (13, 13, [(512, 'identity')], 75, 'none', True, False, True, False, False, False, False),        # SE-gated project
    (13, 13, [(75, 'identity')], 75, 'none', False, False, False, True, True, False, False),          # y conv: dense fp32 logits
    (26, 26, [(24, 'maxpool4')], 48, 'none', False, False, False, False, False, False, False),         # rfcr_b4c
    (5, 5, [(75, 'identity')], 255, 'swish', True, False, False, True, True, False, False),
    (8, 8, [(37, 'identity'), (22, 'up2')], 50, 'leaky', True, True, False, False, False, False, False),
    (1, 1, [(128, 'identity')], 32, 'sigmoid', True, False, False, False, False, False, False),
    (12, 12, [(128, 'identity')], 128, 'relu6', True, False, False, False, False, True, False),        # pooled output (bu*_down_conv)
    (14, 10, [(48, 'identity'), (96, 'identity')], 203, 'relu6', True, False, False, False, False, True, False),
    (12, 12, [(72, 'identity'), (96, 'identity')], 256, 'relu6', True, False, False, False, False, False, True),   # hoisted: up2_add
    (6, 6, [(256, 'identity')], 75, 'none', False, False, False, True, False, False, False),         # the _lowres half: fp32 out, padded
    (13, 13, [(128, 'identity')], 75, 'none', True, False, True, False, False, False, False),        # SE-gated project, k <= 256 (stationary form)
    (10, 10, [(320, 'identity')], 600, 'relu6', True, False, False, False, False, False, False),      # three cout ranges of the k-streaming form
    (9, 9, [(200, 'identity')], 1400, 'relu6', True, True, False, False, False, False, False),        # 44 cout pairs walked by the stationary form
]


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('case', PW16, ids=[str(i) for i in range(len(PW16))])
def test_pointwise16(dev, dt, case):
    h, w, segs, cout, act, bn, residual, gate, out_f32, dense, pool, pre = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    run_pointwise16(dev, dt, rng, 3, h, w, segs, cout, act, bn, residual, gate, out_f32, dense, pool, pre)


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('case', PW16, ids=[str(i) for i in range(len(PW16))])
def test_pointwise16_ksplit_form(dev, dt, case):
    """The k-split form (se_reduced bit 17: a workgroup = one 16 x 32 tile, its four waves split the k range and meet in LDS in wave
    order; what the 16-bit plan for one or two images asks of its small maps) at the per-op bar of the other forms, every source mode."""
    h, w, segs, cout, act, bn, residual, gate, out_f32, dense, pool, pre = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    run_pointwise16(dev, dt, rng, 3, h, w, segs, cout, act, bn, residual, gate, out_f32, dense, pool, pre, ksplit=True)


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('ci', list(range(22)))
def test_pointwise16_tile_shapes_are_bit_identical(dev, dt, ci):
    """Every tile shape runs the same MFMA sequence per output: the autotuner may swap them freely.  Shapes 1-10 the direct
    kernel, 11-14 the walking small-K form, 15-18 the LDS-tiled form, 19-22 the activation-stationary form, 23-26 the all-couts
    k-streaming form (every source mode: gathers with upsampled / max-pooled / concatenated sources, the SE gate, pooled
    outputs, float32 outputs, hoisted partial sums; ops a form does not take fall back inside the library)."""
    h, w, segs, cout, act, bn, residual, gate, out_f32, dense, pool, pre = PW16[ci]
    outs = []
    n = 26
    for cfg in range(0, n + 1):
        rng = np.random.default_rng(zlib.crc32(str(PW16[ci]).encode()))
        outs.append(run_pointwise16(dev, dt, rng, 2, h, w, segs, cout, act, bn, residual, gate, out_f32, dense, pool, pre, cfg=cfg))
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('k,s,h,w,c,act', [(3, 1, 13, 13, 96, 'relu6'), (3, 2, 15, 17, 144, 'relu6'), (5, 1, 9, 12, 40, 'swish'),
                                           (5, 2, 14, 14, 75, 'swish'), (3, 1, 5, 7, 10, 'none'), (3, 2, 32, 32, 32, 'relu6'),
                                           # 5x5 stride 1 with 64 channels or more: the LDS-tiled form (depthwise_lds.hip) - one tile with
                                           # two bands, tiles of 8 columns in four bands, ragged tiles, a channel tail, a map smaller than the
                                           # kernel, wider than one 32-column tile
                                           (5, 1, 13, 13, 200, 'relu6'), (5, 1, 20, 20, 72, 'swish'), (5, 1, 40, 40, 136, 'relu6'),
                                           (5, 1, 7, 45, 64, 'none'), (5, 1, 37, 9, 100, 'swish'), (5, 1, 3, 3, 64, 'relu6'),
                                           (5, 1, 26, 26, 96, 'leaky'),
                                           # 3x3 stride 1 with 64 channels or more: the same LDS-tiled form (heads: 52 x 52 x 128 ... 13 x 13 x 512)
                                           (3, 1, 52, 52, 128, 'swish'), (3, 1, 26, 26, 256, 'relu6'), (3, 1, 13, 13, 512, 'swish'), (3, 1, 7, 45, 64, 'none'),
                                           (3, 1, 37, 9, 100, 'leaky'), (3, 1, 2, 2, 72, 'relu6')])
def test_depthwise16(dev, dt, k, s, h, w, c, act):
    rt = _rt()
    did = rt.dtype_id(dt)
    rng = np.random.default_rng(k * 1000 + s * 100 + c)
    b = 2
    x = q16(rng.standard_normal((b, h, w, c)), dt)
    wk = (rng.standard_normal((k, k, c)) * np.sqrt(2.0 / (k * k))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.normal(0, 0.3, c).astype(np.float32)
    ref = _act(nn.depthwise(x.astype(np.float64), wk.astype(np.float64), s, 'same') * scale + shift, act)
    ldc = round_up(c, 8)
    wp = np.zeros((k * k, ldc), np.float32)
    wp[:, :c] = wk.reshape(k * k, c)
    pad = lambda v: np.concatenate([v, np.zeros(ldc - c, np.float32)])
    xd = to_dev16(x, dev, dt)
    ho, wo = -(-h // s), -(-w // s)
    out = torch.full((b, ho, wo, round_up(c, 8)), float('nan'), dtype=rt.TORCH_DTYPE[did], device=dev)
    keep = [_dev_vec(wp, dev), _dev_vec(pad(scale), dev), _dev_vec(pad(shift), dev)]
    op = rt.new_op(rt.OP_DEPTHWISE, act)
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = ho, wo, c, c, k, s, 1
    op.src[0] = rt.make_src(xd, c=c)
    op.wgt, op.scale, op.shift = [t.data_ptr() for t in keep]
    op.out, op.out_ld = out.data_ptr(), out.shape[3]
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_rounded_once(from_dev16(out, dt, c), ref, dt, 'depthwise16 k%d s%d' % (k, s))


@pytest.mark.parametrize('pairs', [False, True])
@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('cout,act', [(24, 'relu6'), (32, 'swish'), (40, 'swish')])
def test_stem16(dev, dt, cout, act, pairs):
    """float32 image in, 16-bit map out."""
    rt = _rt()
    did = rt.dtype_id(dt)
    rng = np.random.default_rng(cout)
    b, h, w = 2, 37, 50
    x = rng.random((b, h, w, 3), dtype=np.float32)
    wk = (rng.standard_normal((3, 3, 3, cout)) * np.sqrt(2.0 / 27)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    ref = _act(nn.conv2d(x.astype(np.float64), wk.astype(np.float64), 2, 'same') * scale + shift, act)
    ldw = round_up(cout, 4)
    wp = np.zeros((27, ldw), np.float32)
    wp[:, :cout] = wk.reshape(27, cout)
    xd = torch.from_numpy(x).to(dev)
    ho, wo = (h + 1) // 2, (w + 1) // 2
    out = torch.full((b, ho, wo, round_up(cout, 8)), float('nan'), dtype=rt.TORCH_DTYPE[did], device=dev)
    keep = [_dev_vec(wp, dev), _dev_vec(scale, dev), _dev_vec(shift, dev)]
    op = rt.new_op(rt.OP_STEM, act)
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = ho, wo, 3, cout, 3, 2, 1
    op.src[0] = rt.make_src(xd, c=3)
    op.wgt, op.scale, op.shift = [t.data_ptr() for t in keep]
    if pairs:
        from tests.test_gpu_ops import _stem_pairs
        keep.append(_dev_vec(_stem_pairs(wp, scale, shift, ldw), dev))
        op.wgt2 = keep[-1].data_ptr()
    op.out, op.out_ld = out.data_ptr(), out.shape[3]
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_rounded_once(from_dev16(out, dt, cout), ref, dt, 'stem16 %d' % cout)


@pytest.mark.parametrize('dt', DTYPES)
def test_weighted_sum_and_gather16(dev, dt):
    """WSUM over four gathered 16-bit sources (reference add order, one rounding) and the materialising GATHER
    (exact: a copy / maximum of 16-bit values)."""
    rt = _rt()
    did = rt.dtype_id(dt)
    rng = np.random.default_rng(5)
    b, h, w, c = 2, 12, 12, 48
    srcs = [('up2', q16(rng.standard_normal((b, h // 2, w // 2, c)), dt)), ('identity', q16(rng.standard_normal((b, h, w, c)), dt)),
            ('maxpool2', q16(rng.standard_normal((b, 2 * h, 2 * w, c)), dt)), ('maxpool4', q16(rng.standard_normal((b, 4 * h, 4 * w, c)), dt))]
    alpha = rng.uniform(0.5, 1.5, 4).astype(np.float32)
    xs = [_xform(a, xf) for xf, a in srcs]
    ref = sum(np.float64(alpha[i]) * xs[i].astype(np.float64) for i in range(4))
    devs = [to_dev16(a, dev, dt) for _, a in srcs]
    out = torch.full((b, h, w, round_up(c, 8)), float('nan'), dtype=rt.TORCH_DTYPE[did], device=dev)
    ad = _dev_vec(alpha, dev)
    op = rt.new_op(rt.OP_WSUM)
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.nsrc = h, w, c, c, 4
    for i, ((xf, _), t) in enumerate(zip(srcs, devs)):
        op.src[i] = rt.make_src(t, c=c, xform=xf)
    op.wgt = ad.data_ptr()
    op.out, op.out_ld = out.data_ptr(), out.shape[3]
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_rounded_once(from_dev16(out, dt, c), ref, dt, 'wsum16')
    # gather: concat of three sources with ragged widths
    gs = [('up2', q16(rng.standard_normal((b, h // 2, w // 2, 37)), dt)), ('identity', q16(rng.standard_normal((b, h, w, 22)), dt)),
          ('maxpool2', q16(rng.standard_normal((b, 2 * h, 2 * w, 75)), dt))]
    refg = nn.concat([_xform(a, xf) for xf, a in gs])
    gd = [to_dev16(a, dev, dt) for _, a in gs]
    ctot = refg.shape[-1]
    outg = torch.zeros((b, h, w, round_up(ctot, 8)), dtype=rt.TORCH_DTYPE[did], device=dev)
    op = rt.new_op(rt.OP_GATHER)
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.nsrc = h, w, ctot, ctot, 3
    for i, ((xf, a), t) in enumerate(zip(gs, gd)):
        op.src[i] = rt.make_src(t, c=a.shape[-1], xform=xf)
    op.out, op.out_ld = outg.data_ptr(), outg.shape[3]
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert np.array_equal(from_dev16(outg, dt, ctot), refg)


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('h,w,c,r,merged', [(13, 13, 75, 18, True), (7, 5, 480, 20, True), (26, 26, 144, 6, False)])
def test_squeeze_excite16(dev, dt, h, w, c, r, merged):
    """SE on a 16-bit map: float32 mean of the widened values, float32 FCs, float32 gate (efficientnet.py:406-434)."""
    rt = _rt()
    did = rt.dtype_id(dt)
    rng = np.random.default_rng(h * 100 + c)
    b = 3
    x = q16(rng.standard_normal((b, h, w, c)), dt)
    w1 = (rng.standard_normal((c, r)) * np.sqrt(1.0 / c)).astype(np.float32)
    b1 = rng.normal(0, 0.1, r).astype(np.float32)
    w2 = (rng.standard_normal((r, c)) * np.sqrt(1.0 / r)).astype(np.float32)
    b2 = rng.normal(0, 0.1, c).astype(np.float32)
    mean = x.astype(np.float64).mean(axis=(1, 2))
    hid = nn.swish(mean.dot(w1.astype(np.float64)) + b1)
    ref = nn.sigmoid(hid.dot(w2.astype(np.float64)) + b2)
    ldc = round_up(c, 4)
    w1t = np.zeros((ldc, round_up(r, 4)), np.float32); w1t[:c, :r] = w1        # (ABI 7: W1 [ldc][R4], b1 [R4])
    w2p = np.zeros((r, ldc), np.float32); w2p[:, :c] = w2
    b2p = np.zeros(ldc, np.float32); b2p[:c] = b2
    keep = [_dev_vec(w1t, dev), _dev_vec(b1, dev, round_up(r, 4)), _dev_vec(w2p, dev), _dev_vec(b2p, dev)]
    xd = to_dev16(x, dev, dt)
    ldg = round_up(c, 8)
    gate = torch.full((b, ldg), float('nan'), dtype=torch.float32, device=dev)
    src = rt.make_src(xd, c=c)
    if not merged:
        mean_d = torch.full((b, 1, 1, ldc), float('nan'), dtype=torch.float32, device=dev)
        op = rt.new_op(rt.OP_SE_MEAN)
        op.dtype, op.out_dtype = did, 0
        op.h = op.w = 1
        op.cin = op.cout = c
        op.nsrc = 1
        op.src[0] = src
        op.out, op.out_ld = mean_d.data_ptr(), ldc
        rt.run_op(op, b)
        got_mean = mean_d.cpu().numpy().reshape(b, ldc)[:, :c]
        assert np.abs(got_mean - mean).max() <= 2e-6 * max(1.0, np.abs(mean).max())
        src = rt.make_src(mean_d, c=c)
    op = rt.new_op(rt.OP_SE_FC)
    op.dtype, op.out_dtype = did, 0
    op.h = op.w = 1
    op.cin = op.cout = c
    op.se_reduced = r
    op.nsrc = 1
    op.src[0] = src
    op.wgt, op.b1, op.wgt2, op.b2 = [t.data_ptr() for t in keep]
    op.out, op.out_ld = gate.data_ptr(), ldg
    rt.run_op(op, b)
    torch.cuda.synchronize()
    g = gate.cpu().numpy()
    assert np.abs(g[:, :c] - ref).max() <= 2e-5
    assert (g[:, c:] == 0).all()


def test_dtype_mismatches_are_refused(dev):
    rt = _rt()
    x = torch.zeros((1, 4, 4, 16), dtype=torch.bfloat16, device=dev)
    out = torch.zeros((1, 4, 4, 16), dtype=torch.float16, device=dev)
    wgt = torch.zeros((16, 16), dtype=torch.bfloat16, device=dev)
    op = rt.new_op(rt.OP_POINTWISE)
    op.dtype, op.out_dtype = rt.DTYPE['bf16'], rt.DTYPE['f16']
    op.h = op.w = 4
    op.cin = op.cout = 16
    op.nsrc = 1
    op.src[0] = rt.make_src(x)
    op.wgt = wgt.data_ptr()
    op.out, op.out_ld = out.data_ptr(), 16
    with pytest.raises(rt.YoloretHipError, match='out_dtype'):
        rt.run_op(op, 1)
    op.out_dtype = rt.DTYPE['bf16']
    op.src[0].dtype = rt.DTYPE['f32']
    with pytest.raises(rt.YoloretHipError, match='dtype'):
        rt.run_op(op, 1)


# ------------------------------------------------------------------------------------------------ whole graph
def _graph16(dev, model_name, hw, b, dt, recipe='conditioned', seed=1234):
    """-> (model, x, fp32 oracle logits, emulated-16-bit oracle logits, HIP logits)."""
    from oracle import model as om, params
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    L.set_global_policy({'bf16': 'mixed_bfloat16', 'f16': 'mixed_float16'}[dt])
    try:
        m = yolov3_body(L.Input(shape=[hw[0], hw[1], 3]), model_name, 3, num_classes=20)
    finally:
        L.set_global_policy('float32')
    assert m.plan.dtype == _rt().dtype_id(dt)
    P = params.ParamStore(seed, recipe)
    x = params.synthetic_images(b, hw[0], hw[1])
    ref = om.yolov3_body(P, x, model_name, 3, 20)
    from tests.util import entry_on_matrix_pipe
    emu = om.yolov3_body(params.QuantStore(seed, recipe, dt, round_entry=entry_on_matrix_pipe(m)), x, model_name, 3, 20)
    m.set_weights(P.values)
    ys = m(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    assert all(y.dtype == torch.float32 for y in ys)          # logits leave the graph as float32
    return m, x, ref, emu, [y.cpu().numpy() for y in ys]


def _errs(a, ref):
    e = np.abs(a.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    return float(e.max()), float(e.mean())


def _check_vs_emulation(ref, emu, got, what):
    worst = 0.0
    for i, (r, e, g) in enumerate(zip(ref, emu, got)):
        assert g.shape == r.shape and np.isfinite(g).all()
        gm, ga = _errs(g, r)
        em, ea = _errs(e, r)
        print('%s y%d  scaled error vs the fp32 oracle (max / mean):  HIP %.2e / %.2e   NumPy emulation %.2e / %.2e'
              % (what, i + 1, gm, ga, em, ea))
        assert ga <= 1.5 * ea + 1e-6, '%s y%d: mean error %.3e vs emulation %.3e' % (what, i + 1, ga, ea)
        assert gm <= 2.0 * em + 1e-5, '%s y%d: max error %.3e vs emulation %.3e' % (what, i + 1, gm, em)
        worst = max(worst, gm)
    return worst


@pytest.mark.parametrize('dt', DTYPES)
@pytest.mark.parametrize('model_name,hw', [('mobilenetv2x75', (64, 64)), ('mobilenetv2x14', (64, 96)), ('efficientnetb0', (96, 64)),
                                           ('efficientnetb3', (64, 64)), ('efficientnetb0-lite', (64, 64)), ('mobilenetv2x75', (352, 352))])
def test_logits16_small(dev, dt, model_name, hw):
    _, _, ref, emu, got = _graph16(dev, model_name, hw, 2, dt)
    _check_vs_emulation(ref, emu, got, '%s@%dx%d %s' % (model_name, hw[0], hw[1], dt))


@pytest.mark.parametrize('dt,model_name,hw', [('bf16', 'efficientnetb0', (416, 416)), ('f16', 'efficientnetb3-lite', (320, 320)), ('bf16', 'efficientnetb0-lite', (224, 160))])
def test_logits16_with_every_walkable_head_block_fused(dev, monkeypatch, dt, model_name, hw):
    """The measured default fuses only td3 in a 16-bit plan (compiler.HEAD_WALK16_MAX_NK = 2); with the limit at 8 the plan
    runs td2, td3, bu3 and bu2 on headwalk_h.hip (7 - 8 chunks, a gated source, up-sampled addends) - held to the same bar."""
    from yoloret_amd import compiler
    monkeypatch.setattr(compiler, 'HEAD_WALK16_MAX_NK', 8)
    m, _, ref, emu, got = _graph16(dev, model_name, hw, 2, dt)
    heads = [o.name for o in m.plan.ops if o.kind == _rt().OP_HEAD]
    assert sorted(heads) == ['bu2_head', 'bu3_head', 'td2_head', 'td3_head'], heads
    _check_vs_emulation(ref, emu, got, '%s@%dx%d %s, four head blocks fused' % (model_name, hw[0], hw[1], dt))


# Absolute bars of the whole-graph 16-bit tests: (scaled max, scaled mean) logit error against the float32 oracle, about
# twice what round 3 measured (profiles/r03_fullres_tests.txt).  The error is a property of model x format on RANDOM
# weights far more than of the kernels: the squeeze-excite / swish EfficientNets sit at 5e-3 (bf16) and 1e-3 (f16), the
# ReLU6 networks without gates (MobileNetV2, the -lite forms) amplify storage rounding 20-80x - and the NumPy emulation
# of the format shows the same numbers.
CEIL16 = {('efficientnetb0', 'bf16'): (1e-2, 1.5e-3), ('efficientnetb3', 'f16'): (2.5e-3, 3e-4),
          ('efficientnetb0-lite', 'bf16'): (2.5e-1, 2.5e-2), ('efficientnetb3-lite', 'f16'): (2e-1, 1.7e-2),
          ('mobilenetv2x75', 'bf16'): (3e-1, 4e-2), ('mobilenetv2x75', 'f16'): (5e-2, 5.5e-3)}
MIN_AGREE = {'bf16': 0.8, 'f16': 0.9}      # share of the float32 oracle's (class, box) picks a 16-bit plan must reproduce (measured 87-90 % / 96-99.8 %)


def _detection_agreement(ys_a, ys_b, hw, thr=0.2):
    """(class, box index) picks of the oracle's post-processing on two sets of logits: |A & B|, |A|, |B|."""
    from oracle import cpost
    from tests.util import ANCHORS
    inter = na = nb = 0
    for i in range(ys_a[0].shape[0]):
        _, _, ca, ia = cpost.yolo_eval([y[i] for y in ys_a], ANCHORS, 3, 20, hw, 20, thr, 0.5)
        _, _, cb, ib = cpost.yolo_eval([y[i] for y in ys_b], ANCHORS, 3, 20, hw, 20, thr, 0.5)
        sa, sb = set(zip(ca.tolist(), ia.tolist())), set(zip(cb.tolist(), ib.tolist()))
        inter += len(sa & sb); na += len(sa); nb += len(sb)
    return inter, na, nb


@pytest.mark.parametrize('model_name,size,dt', [('efficientnetb0', 416, 'bf16'),         # config 3 (reference-faithful SE + Swish)
                                                ('efficientnetb0-lite', 416, 'bf16'),    # config 3, build-defined lite form
                                                ('efficientnetb3', 640, 'f16'),          # config 5
                                                ('efficientnetb3-lite', 640, 'f16'),
                                                ('mobilenetv2x75', 416, 'bf16'), ('mobilenetv2x75', 416, 'f16')])
def test_baseline_configs_16bit(dev, model_name, size, dt):
    """BASELINE.json configs 3 and 5 at their resolution: logits against the float32 oracle and against the NumPy
    emulation of the storage format, detections (GPU post-processing == the oracle's on the same logits, bit for bit;
    agreement of the picks with the float32 oracle's reported)."""
    from oracle import cpost
    from tests.util import ANCHORS
    from yoloret_amd.yolo3.model import yolo_eval
    b, hw = 1, (size, size)
    m, x, ref, emu, got = _graph16(dev, model_name, hw, b, dt)
    worst = _check_vs_emulation(ref, emu, got, '%s@%d %s' % (model_name, size, dt))
    # the fusion-independent bar: the emulation rounds at EVERY convolution output of the graph (the unfused placement, whatever
    # the plan fuses; only its entry follows the plan) - over the three outputs the HIP path's mean error must not exceed it
    # (x 1.05: the two differ by which of two equally good roundings a value takes)
    hip_mean = float(np.mean([_errs(g, r)[1] for g, r in zip(got, ref)]))
    emu_mean = float(np.mean([_errs(e, r)[1] for e, r in zip(emu, ref)]))
    assert hip_mean <= 1.05 * emu_mean + 1e-7, 'mean scaled logit error %.3e above the unfused-placement emulation %.3e' % (hip_mean, emu_mean)
    ys = [torch.from_numpy(g).to(dev) for g in got]
    res = yolo_eval(ys, ANCHORS, 3, 20, hw, max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
    gb, gs, gc = [t.cpu().numpy() for t in res]
    ob, os_, oc, _ = cpost.yolo_eval([g[0] for g in got], ANCHORS, 3, 20, hw, 20, 0.2, 0.5)
    assert np.array_equal(gb, ob) and np.array_equal(gs, os_) and np.array_equal(gc, oc)   # decode / NMS are float32: bit-exact
    inter, na, nb = _detection_agreement(got, ref, hw)
    inter_e, ne, _ = _detection_agreement(emu, ref, hw)
    print('%s@%d %s: max scaled logit error %.2e; detections HIP %d, fp32 oracle %d, common %d (%.1f %%); NumPy emulation: common %d of %d'
          % (model_name, size, dt, worst, na, nb, inter, 100.0 * inter / max(nb, 1), inter_e, ne))
    # the 16-bit path may flip decisions that sit within its logit noise of a threshold - no more of them than the
    # emulation of the same format does (+ slack for the small counts)
    assert nb > 0 and inter >= 0.8 * min(inter_e, nb) - 2
    # ... and ABSOLUTE bars, so that kernel and emulation cannot degrade together unnoticed: scaled max and mean logit error
    # against the float32 oracle and the share of the float32 oracle's picks the 16-bit path reproduces
    cmax, cmean = CEIL16[(model_name, dt)]
    mean = max(_errs(g, r)[1] for g, r in zip(got, ref))
    assert worst <= cmax and mean <= cmean, 'scaled logit error max %.3e / mean %.3e above the ceilings %.1e / %.1e' % (worst, mean, cmax, cmean)
    assert inter >= MIN_AGREE[dt] * nb, 'only %d of the float32 oracle\'s %d picks reproduced' % (inter, nb)


def test_policy_and_dtype_plumbing(dev):
    """set_global_policy / Model(dtype=...) / YOLORET_DTYPE select the plan's element type; float32 stays the default;
    the 16-bit plan halves the arena and the pointwise weights."""
    from yoloret_amd import layers as L
    from yoloret_amd.engine import Model
    from yoloret_amd.yolo3.model import yolov3_body
    rt = _rt()
    m32 = yolov3_body(L.Input(shape=[96, 96, 3]), 'efficientnetb0', 3, num_classes=20)
    assert m32.plan.dtype == 0 and all(o.dtype == 0 for o in m32.plan.ops)
    L.set_global_policy('mixed_float16')
    try:
        m16 = yolov3_body(L.Input(shape=[96, 96, 3]), 'efficientnetb0', 3, num_classes=20)
    finally:
        L.set_global_policy('float32')
    assert m16.plan.dtype == rt.DTYPE['f16']
    assert m16.plan.arena_bytes_per_image < 0.6 * m32.plan.arena_bytes_per_image
    assert m16.plan.blob_floats < 0.75 * m32.plan.blob_floats
    assert all(b.dtype == 0 for b in m16.plan.bufs if b.external_slot >= 0)
    with pytest.raises(ValueError):
        L.set_global_policy('int8')
    with pytest.raises(ValueError):
        Model(m32.inputs, m32.outputs, dtype='float64')


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('k,s,h,w,c,r', [(3, 1, 13, 13, 512, 128), (3, 1, 26, 26, 256, 64), (3, 1, 52, 52, 128, 32), (5, 1, 13, 9, 64, 4),
                                         (3, 2, 27, 26, 128, 8), (5, 2, 14, 14, 256, 16),
                                         # channel-vector counts that do not divide 256 (several workgroups share a row):
                                         (3, 1, 40, 36, 40, 10), (5, 1, 20, 20, 816, 34), (5, 2, 21, 20, 1392, 58), (3, 1, 9, 7, 100, 6),
                                         (3, 1, 13, 13, 1152, 48), (5, 1, 40, 40, 136, 8), (5, 1, 26, 26, 100, 8), (5, 1, 70, 37, 64, 4)])
def test_depthwise_se_form(dev, dt, k, s, h, w, c, r):
    """The squeeze of squeeze-excite as an epilogue of the depthwise conv (efficientnet.py:417 after :501-510): the SE
    form writes the same map as the plain op, bit for bit, plus per-workgroup channel sums; SE_FC (k = pixel count) on
    those rows gives the oracle's gate.  All three element types (float32: the head blocks of the headline config)."""
    rt = _rt()
    did = rt.dtype_id(dt)
    V = rt.VEC[did]
    rng = np.random.default_rng(k * 100 + c + r)
    b = 3
    x = rng.standard_normal((b, h, w, c)).astype(np.float32)
    if dt != 'f32':
        x = q16(x, dt)
    wk = (rng.standard_normal((k, k, c)) * np.sqrt(2.0 / (k * k))).astype(np.float32)
    scale, shift = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(0, 0.3, c).astype(np.float32)
    ldc = round_up(c, V)
    wp = np.zeros((k * k, ldc), np.float32); wp[:, :c] = wk.reshape(k * k, c)
    pad = lambda v: np.concatenate([v, np.zeros(ldc - c, np.float32)])
    tdt = rt.TORCH_DTYPE[did]
    xd = torch.from_numpy(x).to(dev) if dt == 'f32' else to_dev16(x, dev, dt)
    ho, wo = -(-h // s), -(-w // s)
    keep = [_dev_vec(wp, dev), _dev_vec(pad(scale), dev), _dev_vec(pad(shift), dev)]

    def dw(gate=None, rows=0):
        out = torch.full((b, ho, wo, ldc), float('nan'), dtype=tdt, device=dev)
        op = rt.new_op(rt.OP_DEPTHWISE, 'swish')
        op.dtype = op.out_dtype = did
        op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = ho, wo, c, c, k, s, 1
        op.src[0] = rt.make_src(xd, c=c)
        op.wgt, op.scale, op.shift = [t.data_ptr() for t in keep]
        op.out, op.out_ld = out.data_ptr(), ldc
        if gate is not None:
            op.gate, op.gate_ld, op.se_reduced = gate.data_ptr(), gate.shape[2], rows
        rt.run_op(op, b)
        torch.cuda.synchronize()
        return out
    plain = dw()
    from yoloret_amd.compiler import dw_se_geometry, dwl_geometry, DW_LDS
    c4 = (c + V - 1) // V
    xt = 4 if s == 1 else 2
    rows = dw_se_geometry(ho * ((wo + xt - 1) // xt), c4)[2]
    if c4 <= 256 and 256 % c4 == 0:
        assert rows == (ho * ((wo + xt - 1) // xt) * c4 + 255) // 256
    if DW_LDS and dt != 'f32' and k in (3, 5) and s == 1 and c >= 64:     # the LDS-tiled form: one row per tile
        ntx, nty = dwl_geometry(ho, wo, k)
        rows = ntx * nty
    part = torch.full((b, rows, ldc), float('nan'), dtype=torch.float32, device=dev)
    fused = dw(part, rows)
    assert torch.equal(plain.view(torch.int16 if dt != 'f32' else torch.int32), fused.view(torch.int16 if dt != 'f32' else torch.int32))
    stored = plain.float().cpu().numpy()[..., :c].astype(np.float64)       # the mean is that of the stored values
    sums = part.cpu().numpy()[..., :c].astype(np.float64).sum(axis=1)
    assert np.abs(sums - stored.sum(axis=(1, 2))).max() <= 2e-5 * max(1.0, np.abs(stored).sum(axis=(1, 2)).max())
    # SE_FC on the rows
    w1 = (rng.standard_normal((c, r)) * np.sqrt(1.0 / c)).astype(np.float32)
    b1 = rng.normal(0, 0.1, r).astype(np.float32)
    w2 = (rng.standard_normal((r, c)) * np.sqrt(1.0 / r)).astype(np.float32)
    b2 = rng.normal(0, 0.1, c).astype(np.float32)
    mean = stored.mean(axis=(1, 2))
    ref = nn.sigmoid(nn.swish(mean.dot(w1.astype(np.float64)) + b1).dot(w2.astype(np.float64)) + b2)
    l4 = round_up(c, 4)
    w1t = np.zeros((l4, round_up(r, 4)), np.float32); w1t[:c, :r] = w1
    w2p = np.zeros((r, l4), np.float32); w2p[:, :c] = w2
    b2p = np.zeros(l4, np.float32); b2p[:c] = b2
    k2 = [_dev_vec(w1t, dev), _dev_vec(b1, dev, round_up(r, 4)), _dev_vec(w2p, dev), _dev_vec(b2p, dev)]
    gate = torch.full((b, ldc), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_SE_FC)
    op.dtype, op.out_dtype = did, 0
    op.h = op.w = 1
    op.cin = op.cout = c
    op.se_reduced, op.nsrc, op.k = r, 1, ho * wo
    op.src[0] = rt.make_src(part.view(b, rows, 1, ldc), c=c)
    op.wgt, op.b1, op.wgt2, op.b2 = [t.data_ptr() for t in k2]
    op.out, op.out_ld = gate.data_ptr(), ldc
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert np.abs(gate.cpu().numpy()[:, :c] - ref).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('dt', DTYPES)
def test_depthwise_walk_is_batch_independent(dev, dt):
    """The walking 5x5 form (depthwise_walk.hip) cuts the rows of an image into segments by how many workgroups a LAUNCH has
    (a batch of 2: one segment per row quantum; a batch of 400 on this map: whole images) - the stored map and the
    squeeze-excite sums (one row per quantum whatever the segmentation) of an image must not depend on it.  Also an odd batch
    (the second image slot of the last workgroup is empty) and the float64 reference on the large batch."""
    rt = _rt()
    from yoloret_amd.compiler import dwl_geometry, DW_LDS, DW_WALK
    if not (DW_LDS and DW_WALK):
        pytest.skip('the walking form is switched off')
    did = rt.dtype_id(dt)
    k, c = 5, 64
    rng = np.random.default_rng(77)
    wk = (rng.standard_normal((k, k, c)) * 0.2).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.normal(0, 0.3, c).astype(np.float32)
    keep = [_dev_vec(wk.reshape(k * k, c), dev), _dev_vec(scale, dev), _dev_vec(shift, dev)]
    for h, w, big in [(26, 26, 400), (13, 13, 7), (33, 20, 301)]:
        ntx, nty = dwl_geometry(h, w, k)
        x = q16(rng.standard_normal((big, h, w, c)), dt)
        xd = to_dev16(x, dev, dt)

        def run(b):
            out = torch.full((b, h, w, c), float('nan'), dtype=rt.TORCH_DTYPE[did], device=dev)
            part = torch.full((b, ntx * nty, c), float('nan'), dtype=torch.float32, device=dev)
            op = rt.new_op(rt.OP_DEPTHWISE, 'swish')
            op.dtype = op.out_dtype = did
            op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = h, w, c, c, k, 1, 1
            op.src[0] = rt.make_src(xd[:b], c=c)
            op.wgt, op.scale, op.shift = [t.data_ptr() for t in keep]
            op.out, op.out_ld = out.data_ptr(), c
            op.gate, op.gate_ld, op.se_reduced = part.data_ptr(), c, ntx * nty
            rt.run_op(op, b)
            torch.cuda.synchronize()
            return out, part
        out_big, part_big = run(big)
        ref = _act(nn.depthwise(x[:16].astype(np.float64), wk.astype(np.float64), 1, 'same') * scale + shift, 'swish')
        assert_rounded_once(from_dev16(out_big[:16], dt, c), ref, dt, 'walking depthwise %dx%d' % (h, w))
        assert not torch.isnan(part_big).any()
        for b in (1, 2, 3):
            out_b, part_b = run(b)
            assert torch.equal(out_b.view(torch.int16), out_big[:b].view(torch.int16)), (h, w, b)
            assert torch.equal(part_b, part_big[:b]), (h, w, b)
        sums = part_big.sum(dim=1).double().cpu().numpy()
        stored = out_big.double().sum(dim=(1, 2)).cpu().numpy()
        assert np.abs(sums - stored).max() <= 2e-5 * max(1.0, np.abs(stored).max())


@pytest.mark.gpu
def test_depthwise_lds_form_geometry_mirror(dev):
    """compiler.dwl_geometry must pick the tile the launcher picks (depthwise_lds.hip) for every map size: the SE partial-sum
    buffer is sized from it and the launcher refuses a buffer of another height.  64 channels, one image, a grid of map
    sizes (the stage maps of 320 ... 1280-pixel inputs and odd ones); the sums must equal those of the stored map."""
    rt = _rt()
    from yoloret_amd.compiler import dwl_geometry, DW_LDS
    if not DW_LDS:
        pytest.skip('YOLORET_DW_LDS=0')
    did = rt.dtype_id('f16')
    c, k = 64, 5
    rng = np.random.default_rng(5)
    wp = (rng.standard_normal((k * k, c)) * 0.2).astype(np.float32)
    keep = [_dev_vec(wp, dev), _dev_vec(np.ones(c, np.float32), dev), _dev_vec(np.zeros(c, np.float32), dev)]
    sizes = [3, 5, 10, 13, 16, 19, 20, 26, 32, 33, 38, 40, 52, 64, 80]
    for h in sizes:
        for w in sizes:
            x = torch.randn((1, h, w, c), device=dev).to(torch.float16)
            ntx, nty = dwl_geometry(h, w)
            part = torch.full((1, ntx * nty, c), float('nan'), dtype=torch.float32, device=dev)
            out = torch.empty_like(x)
            op = rt.new_op(rt.OP_DEPTHWISE, 'relu6')
            op.dtype = op.out_dtype = did
            op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = h, w, c, c, k, 1, 1
            op.src[0] = rt.make_src(x, c=c)
            op.wgt, op.scale, op.shift = [t.data_ptr() for t in keep]
            op.out, op.out_ld = out.data_ptr(), c
            op.gate, op.gate_ld, op.se_reduced = part.data_ptr(), c, ntx * nty
            rt.run_op(op, 1)     # raises if the launcher's tile differs from the mirror's
            torch.cuda.synchronize()
            sums = part.sum(dim=1).double().cpu().numpy()
            ref = out.double().sum(dim=(1, 2)).cpu().numpy()
            assert np.abs(sums - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max()), (h, w)


@pytest.mark.parametrize('dt,model_name,hw', [('bf16', 'efficientnetb0', (224, 224)), ('f16', 'efficientnetb3', (192, 160))])
def test_se_model_results_do_not_depend_on_the_tuning_table_or_the_batch(dev, monkeypatch, dt, model_name, hw):
    """SURVEY.md 4.1 (last row): N ranks with B / N images each must equal one rank with B.  Every rank autotunes (or installs rank 0's
    table) per batch size, so the RESULT may not depend on a table entry or on the batch: the squeeze sums of the register-chained
    expand + depthwise ops (YR_OP_MBX, mbxr_h.hip) leave per (strip, quantum of output rows) - quanta fixed by the map's shape, a
    tuned row segment is a whole number of them.  Tables with 1 / 2 / 3 / 6 row segments for every chained block op, the library's own
    choice, and image 0 of a batch of 3 run alone: bit-equal logits."""
    import ctypes
    monkeypatch.setenv('YOLORET_AUTOTUNE', '0')
    rt = _rt()
    m, x, ref, emu, base = _graph16(dev, model_name, hw, 3, dt)
    xd = torch.from_numpy(x).to(dev)
    chained = [i for i, o in enumerate(m.plan.ops) if o.kind in (rt.OP_MBX, rt.OP_MBH)]
    assert sum(1 for i in chained if m.plan.ops[i].kind == rt.OP_MBX) >= 4
    idx, hd = m._handle(xd.device, 3)
    n = len(m.plan.ops)
    for segs in (1, 2, 3, 6):
        cfg = [0] * n
        for i in chained:
            cfg[i] = 255 << 8 | segs << 16        # the register-chained form with this many row segments per strip
        arr = (ctypes.c_int32 * n)(*cfg)
        if rt.lib().yr_set_tuning(hd, 3, arr, n) != 0:      # (an op the chained form is not built for refuses tile 255: leave those alone)
            for i in chained:
                if m.plan.ops[i].kind == rt.OP_MBH:
                    cfg[i] = 0
            arr = (ctypes.c_int32 * n)(*cfg)
            rt.check(rt.lib().yr_set_tuning(hd, 3, arr, n))
        got = [y.cpu().numpy() for y in m(xd)]
        for g, b0 in zip(got, base):
            assert np.array_equal(g, b0), '%d row segments change the logits' % segs
    one = [y.cpu().numpy() for y in m(xd[:1])]
    for g, b0 in zip(one, base):
        assert np.array_equal(g[0], b0[0]), 'image 0 alone differs from image 0 of the batch'
