"""Per-kernel parity: every fused HIP op (through the C-ABI, yr_op_run) against the NumPy
oracle on the same seeded inputs.  fp32 tolerance 1e-5*max(1,|ref|) per op."""
import ctypes
import zlib

import numpy as np
import pytest
import torch

from oracle import nn
from tests.util import assert_close, from_dev, round_up, to_dev

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _rt():
    from yoloret_amd import runtime as rt
    return rt


def _dev_vec(a, dev, n=None):
    a = np.asarray(a, np.float32).ravel()
    if n is not None and n != a.size:
        p = np.zeros(n, np.float32)
        p[:a.size] = a
        a = p
    return torch.from_numpy(a).to(dev)


def _act_np(x, act):
    return {'none': lambda v: v, 'relu6': nn.relu6, 'swish': nn.swish, 'sigmoid': nn.sigmoid,
            'leaky': nn.leaky_relu}[act](x)


def _xform_np(x, xf):
    return {'identity': lambda v: v, 'up2': nn.upsample2, 'maxpool2': lambda v: nn.maxpool(v, 2),
            'maxpool4': lambda v: nn.maxpool(v, 4)}[xf](x)


def _src_dims(h, w, xf):
    return {'identity': (h, w), 'up2': (h // 2, w // 2), 'maxpool2': (h * 2, w * 2), 'maxpool4': (h * 4, w * 4)}[xf]


def run_pointwise(dev, rng, b, h, w, segs, cout, act='none', bn=True, residual=False, gate=False, out_ld=None, cfg=0, ksplit=False, stream=False):
    rt = _rt()
    srcs_np, srcs_dev = [], []
    for c, xf in segs:
        sh, sw = _src_dims(h, w, xf)
        a = rng.standard_normal((b, sh, sw, c)).astype(np.float32)
        srcs_np.append(a)
        srcs_dev.append(to_dev(a, dev))
    cin = sum(c for c, _ in segs)
    wk = (rng.standard_normal((cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32)
    kp = sum(round_up(c, 4) for c, _ in segs)
    wt = np.zeros((cout, kp), np.float32)
    d = kb = 0
    for c, _ in segs:
        wt[:, kb:kb + c] = wk[d:d + c].T
        d += c
        kb += round_up(c, 4)
    x = nn.concat([_xform_np(a, xf) for a, (_, xf) in zip(srcs_np, segs)])
    gate_np = None
    if gate:
        gate_np = rng.uniform(0.1, 1.0, (b, 1, 1, cin)).astype(np.float32)
        x = gate_np * x
    ref = nn.pointwise(x, wk)
    scale = shift = None
    if bn:
        scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        shift = rng.normal(0, 0.3, cout).astype(np.float32)
        ref = ref * scale + shift
    ref = _act_np(ref.astype(np.float32), act)
    res_np = None
    if residual:
        res_np = rng.standard_normal((b, h, w, cout)).astype(np.float32)
        ref = ref + res_np
    out_ld = round_up(cout, 4) if out_ld is None else out_ld
    out = torch.full((b, h, w, out_ld), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_POINTWISE, act)
    op.h, op.w, op.cin, op.cout, op.nsrc = h, w, cin, cout, len(segs)
    for i, (t, (c, xf)) in enumerate(zip(srcs_dev, segs)):
        op.src[i] = rt.make_src(t, c=c, xform=xf)
    keep = [_dev_vec(wt, dev)]
    if stream:            # the pixel-stationary form (pointwise_stream.hip): se_reduced bit 18, the weights as float16 planes
        from yoloret_amd import compiler
        keep = [_dev_vec(compiler.head_pack(wt, [kp], nk=compiler.pwt_chunks(kp)), dev)]
        op.se_reduced |= 0x40000
    op.wgt = keep[0].data_ptr()
    if bn:
        keep += [_dev_vec(scale, dev), _dev_vec(shift, dev)]
        op.scale, op.shift = keep[1].data_ptr(), keep[2].data_ptr()
    if residual:
        r = to_dev(res_np, dev)
        keep.append(r)
        op.res, op.res_ld = r.data_ptr(), r.shape[3]
    if gate:
        g = to_dev(gate_np.reshape(b, 1, 1, cin), dev)
        keep.append(g)
        op.gate, op.gate_ld = g.data_ptr(), g.shape[3]
    op.out, op.out_ld = out.data_ptr(), out_ld
    if ksplit:
        op.se_reduced |= 0x20000      # the k-split form of the few-image plans (pointwise_split.hip: pwk_kernel)
    op.k = cfg            # 0: heuristic tile shape; 1..yr_pointwise_num_cfgs(): forced (15..: the LDS-free direct kernel)
    rt.run_op(op, b)
    torch.cuda.synchronize()
    got = from_dev(out, cout)
    assert_close(got, ref, TOL, 'pointwise %s cfg %d' % (segs, cfg))
    return got


PW_CASES = [
    # (h, w, segs, cout, act, bn, residual, gate, dense_out)
    (13, 13, [(16, 'identity')], 96, 'relu6', True, False, False, False),
    (13, 11, [(24, 'identity')], 16, 'none', True, False, False, False),
    (7, 9, [(144, 'identity')], 24, 'none', True, True, False, False),
    (13, 13, [(720, 'identity')], 120, 'none', True, True, False, False),
    (26, 26, [(72, 'identity')], 432, 'relu6', True, False, False, False),
    (13, 13, [(120, 'identity'), (96, 'maxpool2')], 512, 'relu6', True, False, False, False),
    (26, 26, [(256, 'up2'), (72, 'identity'), (96, 'identity')], 256, 'relu6', True, False, False, False),
    (12, 12, [(128, 'up2'), (24, 'identity'), (96, 'up2')], 128, 'relu6', True, False, False, False),
    (13, 13, [(128, 'maxpool2'), (75, 'identity')], 256, 'relu6', True, False, False, False),
    (13, 13, [(512, 'identity')], 75, 'none', True, False, True, False),
    (13, 13, [(75, 'identity')], 75, 'none', False, False, False, True),   # y conv: dense 75-wide logits
    (26, 26, [(24, 'maxpool4')], 48, 'none', False, False, False, False),  # rfcr_b4c
    (5, 5, [(75, 'identity')], 255, 'swish', True, False, False, True),
    (8, 8, [(37, 'identity'), (22, 'up2')], 50, 'leaky', True, True, False, False),
    (1, 1, [(128, 'identity')], 32, 'sigmoid', True, False, False, False),
]


@pytest.mark.parametrize('case', PW_CASES, ids=[str(i) for i in range(len(PW_CASES))])
def test_pointwise(dev, case):
    h, w, segs, cout, act, bn, residual, gate, dense = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    run_pointwise(dev, rng, 3, h, w, segs, cout, act, bn, residual, gate, out_ld=cout if dense else None)


@pytest.mark.parametrize('case', PW_CASES, ids=[str(i) for i in range(len(PW_CASES))])
def test_pointwise_direct_kernel(dev, case):
    """The LDS-free kernel (forced tile shapes 15..29) == oracle, and bit-identical to the LDS-staged kernel:
    both run the same MFMA sequence per output, which is what lets yr_autotune swap shapes freely."""
    h, w, segs, cout, act, bn, residual, gate, dense = case
    outs = []
    for cfg in (0, 15, 16, 19, 22, 24, 27, 29):
        rng = np.random.default_rng(zlib.crc32(str(case).encode()))
        outs.append(run_pointwise(dev, rng, 3, h, w, segs, cout, act, bn, residual, gate, out_ld=cout if dense else None, cfg=cfg))
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize('case', PW_CASES, ids=[str(i) for i in range(len(PW_CASES))])
def test_pointwise_ksplit_form(dev, case):
    """The k-split form (se_reduced bit 17: what the float32 plan for a few images asks of its small maps - a workgroup is one 16 x 16
    tile, its four waves split the k range and meet in LDS) == oracle at the per-op bar, with every source transform, gate, residual,
    dense rows (that the plan for a few images really runs it: tests/test_gpu_graph.py)."""
    h, w, segs, cout, act, bn, residual, gate, dense = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    run_pointwise(dev, rng, 3, h, w, segs, cout, act, bn, residual, gate, out_ld=cout if dense else None, ksplit=True)


@pytest.mark.parametrize('b', [3, 40])
@pytest.mark.parametrize('case', PW_CASES, ids=[str(i) for i in range(len(PW_CASES))])
def test_pointwise_stream_form(dev, case, b):
    """The pixel-stationary form (se_reduced bit 18: what the throughput plan asks of its 1x1 convs - a wave keeps the float16 planes of
    its 16 or 32 pixels over the whole k space, the cout tiles' host-cut planes sit in LDS) == oracle at the per-op bar with every
    source transform, gate, residual and dense rows, and BIT-IDENTICAL to the tiled split kernel: the same three MFMAs per chunk in
    the same order (so a plan may mix the two forms across its variants)."""
    from yoloret_amd import compiler
    h, w, segs, cout, act, bn, residual, gate, dense = case
    kp = sum(round_up(c, 4) for c, _ in segs)
    if (kp < 16 or not compiler.pwt_chunks(kp) or h * w == 1 or any(xf.startswith('maxpool') for _, xf in segs) or residual
            or act not in ('none', 'relu6')):
        pytest.skip('not a shape of the form')
    outs = []
    for st in (False, True):
        rng = np.random.default_rng(zlib.crc32(str(case).encode()))
        outs.append(run_pointwise(dev, rng, b, h, w, segs, cout, act, bn, residual, gate, out_ld=cout if dense else None, stream=st))
    assert np.array_equal(outs[0], outs[1])


def test_pointwise_stream_form_many_tiles_per_wave(dev):
    """... at a pixel count where a wave walks several pixel tiles (bu3_y's shape at 50 images), gated."""
    outs = []
    for st in (False, True):
        rng = np.random.default_rng(11)
        outs.append(run_pointwise(dev, rng, 50, 52, 52, [(128, 'identity')], 75, 'none', True, False, True, out_ld=75, stream=st))
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize('gated', [False, True])
@pytest.mark.parametrize('shape', [(52, 52, 128, 75, 128, 40), (26, 26, 75, 75, 256, 3), (8, 12, 40, 20, 48, 5)])
def test_pointwise_stream_two_outputs(dev, shape, gated):
    _two_outputs(dev, shape, gated)


def _two_outputs(dev, shape, gated):
    """The two-output form (se_reduced bits 18 + 19: a head's y conv and the bottom-up path's down conv read the same gated map in ONE
    launch - first output dense and unpooled, second ReLU6 + MaxPooling2D(2)) == each conv run alone on the tiled split kernel, bit
    for bit (bu3_y + bu3_down_conv and bu2_y + bu2_down_conv of MobileNetV2 x0.75 @416, and a ragged small case)."""
    from yoloret_amd import compiler
    rt = _rt()
    h, w, cin, n1, n2, b = shape
    rng = np.random.default_rng(zlib.crc32(str(shape).encode()))
    ld = round_up(cin, 4)
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    xd = to_dev(x, dev)
    w1 = (rng.standard_normal((n1, ld)) * np.sqrt(2.0 / cin)).astype(np.float32)
    w2 = (rng.standard_normal((n2, ld)) * np.sqrt(2.0 / cin)).astype(np.float32)
    w1[:, cin:] = 0
    w2[:, cin:] = 0
    sc1, sh1 = rng.uniform(0.5, 1.5, n1).astype(np.float32), rng.normal(0, 0.3, n1).astype(np.float32)
    sc2, sh2 = rng.uniform(0.5, 1.5, n2).astype(np.float32), rng.normal(0, 0.3, n2).astype(np.float32)
    g = to_dev(rng.uniform(0.1, 1.0, (b, 1, 1, ld)).astype(np.float32), dev) if gated else None

    def single(wt, sc, sh, n, act, pooled, out_ld):
        oh, ow = (h // 2, w // 2) if pooled else (h, w)
        out = torch.full((b, oh, ow, out_ld), float('nan'), dtype=torch.float32, device=dev)
        op = rt.new_op(rt.OP_POINTWISE, act)
        op.h, op.w, op.cin, op.cout, op.nsrc, op.stride = oh, ow, cin, n, 1, 2 if pooled else 0
        op.src[0] = rt.make_src(xd, c=cin)
        keep = [_dev_vec(wt, dev), _dev_vec(sc, dev), _dev_vec(sh, dev)]
        op.wgt, op.scale, op.shift = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
        if gated:
            op.gate, op.gate_ld = g.data_ptr(), ld
        op.out, op.out_ld = out.data_ptr(), out_ld
        rt.run_op(op, b)
        torch.cuda.synchronize()
        return from_dev(out, n)
    ref1 = single(w1, sc1, sh1, n1, 'none', False, n1)
    ref2 = single(w2, sc2, sh2, n2, 'relu6', True, round_up(n2, 4))
    nk = compiler.pwt_chunks(ld)
    ta, tb = (n1 + 15) // 16, (n2 + 15) // 16
    planes = np.concatenate([compiler.head_pack(w1, [ld], nk=nk), compiler.head_pack(w2, [ld], nk=nk)])
    sc, sh = np.ones(16 * (ta + tb), np.float32), np.zeros(16 * (ta + tb), np.float32)
    sc[:n1], sc[16 * ta:16 * ta + n2], sh[:n1], sh[16 * ta:16 * ta + n2] = sc1, sc2, sh1, sh2
    out1 = torch.full((b, h, w, n1), float('nan'), dtype=torch.float32, device=dev)
    out2 = torch.full((b, h // 2, w // 2, round_up(n2, 4)), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_POINTWISE, 'none')
    op.h, op.w, op.cin, op.cout, op.nsrc = h, w, cin, n1, 1
    op.src[0] = rt.make_src(xd, c=cin)
    keep = [_dev_vec(planes, dev), _dev_vec(sc, dev), _dev_vec(sh, dev)]
    op.wgt, op.scale, op.shift = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
    if gated:
        op.gate, op.gate_ld = g.data_ptr(), ld
    op.out, op.out_ld = out1.data_ptr(), n1
    op.gate_out, op.gate_out_ld, op.se_hidden, op.reserved0 = out2.data_ptr(), round_up(n2, 4), n2, rt.ACT['relu6'] | 1 << 8
    op.se_reduced |= 0xc0000
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert np.array_equal(from_dev(out1, n1), ref1)
    assert np.array_equal(from_dev(out2, n2), ref2)
    return from_dev(out1, n1), from_dev(out2, n2)


def test_pointwise_stream_form_two_rows_per_wave(dev, monkeypatch):
    """The form with 32 pixels per wave (YR_PWT_ROWS=2: built, measured behind 16 pixels per wave, kept for experiments) computes the
    same bits: single and two-output launches, gated."""
    monkeypatch.setenv('YR_PWT_ROWS', '2')
    _two_outputs(dev, (52, 52, 128, 75, 128, 9), True)
    _two_outputs(dev, (8, 12, 40, 20, 48, 5), False)
    outs = []
    for st in (False, True):
        rng = np.random.default_rng(12)
        outs.append(run_pointwise(dev, rng, 7, 26, 26, [(75, 'identity')], 128, 'relu6', True, False, False, stream=st))
    assert np.array_equal(outs[0], outs[1])


def test_pointwise_stream_form_cuts_a_batch_beyond_the_32_bit_offsets(dev, monkeypatch):
    """The kernel addresses its maps through buffer descriptors (32-bit offsets): a batch whose maps pass 2 GB runs as several launches
    over whole images.  With the limit lowered to three images' worth the result is the uncut one, bit for bit (gated, pooled second
    output: every pointer that moves with the cut)."""
    shape = (26, 26, 75, 75, 256, 7)
    monkeypatch.delenv('YR_PWT_MAX_BYTES', raising=False)
    import tests.test_gpu_ops as me
    outs = []
    for lim in (None, str(3 * 26 * 26 * 76 * 4 + 100)):
        if lim:
            monkeypatch.setenv('YR_PWT_MAX_BYTES', lim)
        outs.append(me._two_outputs(dev, shape, True))
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))


def test_pointwise_ksplit_pooled_output(dev):
    """... and with the MaxPooling2D(2) output of the bottom-up convs (rows walked in 2 x 2-quad-major order inside the 16-row tile)."""
    rt = _rt()
    rng = np.random.default_rng(5)
    b, h, w, cin, cout = 2, 13, 13, 128, 96        # pooled OUTPUT dims; the conv runs at 26 x 26
    x = rng.standard_normal((b, 2 * h, 2 * w, cin)).astype(np.float32)
    wk = (rng.standard_normal((cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    y = np.minimum(np.maximum(nn.pointwise(x, wk) * scale + shift, 0), 6).astype(np.float32)
    ref = y.reshape(b, h, 2, w, 2, cout).max(axis=(2, 4))
    outs = []
    for ks in (False, True, 'stream'):
        xd = to_dev(x, dev)
        out = torch.full((b, h, w, cout), float('nan'), dtype=torch.float32, device=dev)
        op = rt.new_op(rt.OP_POINTWISE, 'relu6')
        op.h, op.w, op.cin, op.cout, op.nsrc, op.stride = h, w, cin, cout, 1, 2
        op.src[0] = rt.make_src(xd, c=cin, xform='identity')
        keep = [_dev_vec(np.ascontiguousarray(wk.T), dev), _dev_vec(scale, dev), _dev_vec(shift, dev)]
        op.wgt, op.scale, op.shift = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
        op.out, op.out_ld = out.data_ptr(), cout
        if ks == 'stream':
            from yoloret_amd import compiler
            keep[0] = _dev_vec(compiler.head_pack(np.ascontiguousarray(wk.T), [cin], nk=compiler.pwt_chunks(cin)), dev)
            op.wgt = keep[0].data_ptr()
            op.se_reduced |= 0x40000
        elif ks:
            op.se_reduced |= 0x20000
        rt.run_op(op, b)
        torch.cuda.synchronize()
        outs.append(from_dev(out, cout))
        assert_close(outs[-1], ref, TOL, 'pooled output, ksplit %s' % ks)


def test_pointwise_large_m(dev):
    """block_1_expand's shape at batch 2: 208x208x16 -> 96 (M = 86528, ragged last tile)."""
    rng = np.random.default_rng(7)
    run_pointwise(dev, rng, 2, 208, 208, [(16, 'identity')], 96, 'relu6')


DW_CASES = [(3, 1, 13, 13, 24), (3, 2, 26, 26, 96), (3, 2, 14, 10, 144), (5, 1, 26, 26, 48), (5, 2, 16, 16, 40),
            (3, 1, 7, 5, 75), (3, 1, 52, 52, 128), (5, 2, 9, 9, 20), (3, 2, 9, 7, 8)]


@pytest.mark.parametrize('k,s,h,w,c', DW_CASES)
@pytest.mark.parametrize('act', ['relu6', 'swish'])
def test_depthwise(dev, k, s, h, w, c, act):
    rt = _rt()
    rng = np.random.default_rng(k * 1000 + s * 100 + h + c)
    b = 2
    x = rng.standard_normal((b, h, w, c)).astype(np.float32)
    wk = (rng.standard_normal((k, k, c)) * np.sqrt(2.0 / (k * k))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
    shift = rng.normal(0, 0.3, c).astype(np.float32)
    ref = _act_np((nn.depthwise(x, wk, s, 'same') * scale + shift).astype(np.float32), act)
    ldc = round_up(c, 4)
    xd = to_dev(x, dev, fill=0.0)  # pad channels meet zero weights; keep them finite
    wd = np.zeros((k * k, ldc), np.float32)
    wd[:, :c] = wk.reshape(k * k, c)
    keep = [_dev_vec(wd, dev), _dev_vec(scale, dev, ldc), _dev_vec(shift, dev, ldc)]
    ho, wo = ref.shape[1:3]
    out = torch.full((b, ho, wo, ldc), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_DEPTHWISE, act)
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = ho, wo, c, c, k, s, 1
    op.src[0] = rt.make_src(xd, c=c)
    op.wgt, op.scale, op.shift = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
    op.out, op.out_ld = out.data_ptr(), ldc
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_close(from_dev(out, c), ref, TOL, 'depthwise')


def _stem_pairs(wd, scale, shift, ldw):
    """[27][ldw] + BN -> [ldw/2][27 taps x 2, times the scale | 1 1 | shift 2]: the pair-packed form (include/yoloret_hip.h)"""
    sc, sh = np.zeros(ldw, np.float32), np.zeros(ldw, np.float32)
    sc[:scale.size], sh[:shift.size] = scale, shift
    rows = np.concatenate([(wd * sc[None]).astype(np.float32), np.ones((1, ldw), np.float32), sh[None]])
    return np.ascontiguousarray(rows.reshape(29, ldw // 2, 2).transpose(1, 0, 2)).reshape(ldw // 2, 58)


@pytest.mark.parametrize('pairs', [False, True])
@pytest.mark.parametrize('hw,cout,act', [((64, 64), 24, 'relu6'), ((32, 96), 40, 'swish'), ((30, 22), 48, 'relu6'), ((41, 67), 32, 'swish')])
def test_stem(dev, hw, cout, act, pairs):
    rt = _rt()
    rng = np.random.default_rng(cout)
    b = 2
    x = rng.random((b, hw[0], hw[1], 3), dtype=np.float32)
    wk = (rng.standard_normal((3, 3, 3, cout)) * np.sqrt(2.0 / 27)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    ref = _act_np((nn.conv2d(x, wk, 2, 'same') * scale + shift).astype(np.float32), act)
    ldw = round_up(cout, 4)
    wd = np.zeros((27, ldw), np.float32)
    wd[:, :cout] = wk.reshape(27, cout)
    xd = torch.from_numpy(x).to(dev)
    keep = [_dev_vec(wd, dev), _dev_vec(scale, dev, ldw), _dev_vec(shift, dev, ldw)]
    ho, wo = ref.shape[1:3]
    out = torch.full((b, ho, wo, ldw), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_STEM, act)
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = ho, wo, 3, cout, 3, 2, 1
    op.src[0] = rt.make_src(xd, c=3, ld=3)
    op.wgt, op.scale, op.shift = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
    if pairs:   # the scalar-operand kernel
        keep.append(_dev_vec(_stem_pairs(wd, scale, shift, ldw), dev))
        op.wgt2 = keep[-1].data_ptr()
    op.out, op.out_ld = out.data_ptr(), ldw
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_close(from_dev(out, cout), ref, TOL, 'stem')


@pytest.mark.parametrize('h,w,c,r', [(13, 13, 512, 128), (26, 26, 256, 64), (52, 52, 128, 32), (7, 5, 75, 6), (9, 9, 20, 1)])
def test_squeeze_excite(dev, h, w, c, r):
    rt = _rt()
    rng = np.random.default_rng(c + r)
    b = 3
    x = rng.standard_normal((b, h, w, c)).astype(np.float32)
    w1 = (rng.standard_normal((c, r)) * np.sqrt(2.0 / c)).astype(np.float32)
    b1 = rng.normal(0, 0.1, r).astype(np.float32)
    w2 = (rng.standard_normal((r, c)) * np.sqrt(2.0 / r)).astype(np.float32)
    b2 = rng.normal(0, 0.1, c).astype(np.float32)
    mean = nn.mean_hw(x)
    gate = nn.sigmoid(nn.pointwise(nn.swish(nn.pointwise(mean, w1) + b1), w2) + b2)
    ldc = round_up(c, 4)
    xd = to_dev(x, dev)
    md = torch.full((b, 1, 1, ldc), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_SE_MEAN)
    op.h, op.w, op.cin, op.cout, op.nsrc = 1, 1, c, c, 1
    op.src[0] = rt.make_src(xd, c=c)
    op.out, op.out_ld = md.data_ptr(), ldc
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_close(from_dev(md, c), mean, TOL, 'se_mean')
    w1t = np.zeros((ldc, round_up(r, 4)), np.float32)      # (ABI 7: W1 [ldc][R4] - the Keras kernel as it is, b1 [R4])
    w1t[:c, :r] = w1
    w2p = np.zeros((r, ldc), np.float32)
    w2p[:, :c] = w2
    keep = [_dev_vec(w1t, dev), _dev_vec(b1, dev, round_up(r, 4)), _dev_vec(w2p, dev), _dev_vec(b2, dev, ldc)]
    gd = torch.full((b, 1, 1, ldc), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_SE_FC)
    op.h, op.w, op.cin, op.cout, op.nsrc, op.se_reduced = 1, 1, c, c, 1, r
    op.src[0] = rt.make_src(md, c=c)
    op.wgt, op.b1, op.wgt2, op.b2 = [k.data_ptr() for k in keep]
    op.out, op.out_ld = gd.data_ptr(), ldc
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_close(from_dev(gd, c), gate, TOL, 'se_fc')
    # the merged form (SE_FC pools the full map itself; its own fixed summation order)
    g2 = torch.full((b, 1, 1, ldc), float('nan'), dtype=torch.float32, device=dev)
    op.src[0] = rt.make_src(xd, c=c)
    op.out = g2.data_ptr()
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert_close(from_dev(g2, c), gate, TOL, 'se_fc with the mean merged in')


def test_weighted_sum_bit_exact(dev):
    """model.py:134's left-to-right a0*x0+a1*x1+a2*x2+a3*x3 is reproduced exactly (no contraction)."""
    rt = _rt()
    rng = np.random.default_rng(3)
    b, h, w, c = 2, 26, 26, 48
    x0 = rng.standard_normal((b, h // 2, w // 2, c)).astype(np.float32)
    x1 = rng.standard_normal((b, h, w, c)).astype(np.float32)
    x2 = rng.standard_normal((b, h * 2, w * 2, c)).astype(np.float32)
    x3 = rng.standard_normal((b, h, w, c)).astype(np.float32)
    a = rng.uniform(0.5, 1.5, 4).astype(np.float32)
    ref = a[0] * nn.upsample2(x0) + a[1] * x1 + a[2] * nn.maxpool(x2, 2) + a[3] * x3
    ts = [to_dev(t, dev) for t in (x0, x1, x2, x3)]
    ad = _dev_vec(a, dev)
    out = torch.full((b, h, w, c), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_WSUM)
    op.h, op.w, op.cin, op.cout, op.nsrc = h, w, c, c, 4
    for i, (t, xf) in enumerate(zip(ts, ['up2', 'identity', 'maxpool2', 'identity'])):
        op.src[i] = rt.make_src(t, c=c, xform=xf)
    op.wgt = ad.data_ptr()
    op.out, op.out_ld = out.data_ptr(), c
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert np.array_equal(from_dev(out), ref)


def test_gather_concat_bit_exact(dev):
    rt = _rt()
    rng = np.random.default_rng(4)
    b, h, w = 2, 12, 8
    segs = [(37, 'identity'), (22, 'up2'), (8, 'maxpool2'), (5, 'maxpool4')]
    arrs, ts = [], []
    for c, xf in segs:
        sh, sw = _src_dims(h, w, xf)
        a = rng.standard_normal((b, sh, sw, c)).astype(np.float32)
        arrs.append(a)
        ts.append(to_dev(a, dev))
    ref = nn.concat([_xform_np(a, xf) for a, (_, xf) in zip(arrs, segs)])
    ctot = ref.shape[-1]
    out = torch.full((b, h, w, ctot), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_GATHER)
    op.h, op.w, op.cin, op.cout, op.nsrc = h, w, ctot, ctot, 4
    for i, (t, (c, xf)) in enumerate(zip(ts, segs)):
        op.src[i] = rt.make_src(t, c=c, xform=xf)
    op.out, op.out_ld = out.data_ptr(), ctot
    rt.run_op(op, b)
    torch.cuda.synchronize()
    assert np.array_equal(from_dev(out), ref)


def test_bad_arguments_report_errors(dev):
    """C-ABI error behaviour: negative status + message, no exception across the boundary."""
    rt = _rt()
    op = rt.new_op(rt.OP_POINTWISE)
    op.nsrc = 0
    rc = rt.lib().yr_op_run(ctypes.byref(op), 1, None)
    assert rc == -1 and b'nsrc' in rt.lib().yr_last_error()
    with pytest.raises(rt.YoloretHipError):
        rt.run_op(op, 1)
    x = torch.zeros((1, 4, 4, 6), device=dev)  # ld not a multiple of 4
    op = rt.new_op(rt.OP_POINTWISE)
    op.h, op.w, op.cin, op.cout, op.nsrc = 4, 4, 6, 8, 1
    op.src[0] = rt.make_src(x, c=6)
    with pytest.raises(rt.YoloretHipError, match='multiple of 4'):
        rt.run_op(op, 1)


def test_removed_op_kind_is_refused(dev):
    """YR_OP_MBCONV (8) was removed in ABI 5 (mbconv.hip: no shipped plan selected it; YR_OP_MBR / MBE / MBLANE take its blocks):
    the number stays reserved and the C-ABI says so instead of crashing."""
    rt = _rt()
    op = rt.new_op(rt.OP_MBCONV)
    op.nsrc = 1
    rc = rt.lib().yr_op_run(ctypes.byref(op), 1, None)
    assert rc == -1 and b'removed in ABI 5' in rt.lib().yr_last_error()
