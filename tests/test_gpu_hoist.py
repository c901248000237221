"""The upsample-hoisting rewrite: W.[up2(a); b] = up2(Wa.a) + Wb.b.  (i) the YR_X_UP2_ADD source of the pointwise
op through yr_op_run against the oracle's plain conv over the concatenation, for every tile shape; (ii) the compiler
applies it to the FPN top-down convs and the whole graph still matches the oracle."""
import zlib

import numpy as np
import pytest
import torch

from oracle import nn
from tests.util import assert_close, from_dev, round_up, to_dev

pytestmark = pytest.mark.gpu


def _pw(rt, dev, srcs, wt, cout, h, w, b, act='none', scale=None, shift=None, cfg=0):
    keep = [torch.from_numpy(np.ascontiguousarray(wt)).to(dev)]
    op = rt.new_op(rt.OP_POINTWISE, act)
    op.h, op.w, op.cout, op.nsrc = h, w, cout, len(srcs)
    op.cin = sum(c for _, c, xf in srcs if xf != 'up2_add')
    for i, (t, c, xf) in enumerate(srcs):
        op.src[i] = rt.make_src(t, c=c, xform=xf)
    op.wgt = keep[0].data_ptr()
    if scale is not None:
        keep += [torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev)]
        op.scale, op.shift = keep[1].data_ptr(), keep[2].data_ptr()
    out = torch.full((b, h, w, round_up(cout, 4)), float('nan'), dtype=torch.float32, device=dev)
    op.out, op.out_ld, op.k = out.data_ptr(), out.shape[3], cfg
    rt.run_op(op, b)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize('case', [(26, 26, [128, 96], [24], 128, 'relu6'), (12, 20, [256], [72, 96], 75, 'relu6'),
                                  (8, 8, [30], [13], 18, 'swish')], ids=['td3', 'td2', 'ragged'])
def test_up2_add_source(dev, case):
    from yoloret_amd import runtime as rt
    h, w, lo_c, hi_c, cout, act = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    b = 2
    lo = [rng.standard_normal((b, h // 2, w // 2, c)).astype(np.float32) for c in lo_c]
    hi = [rng.standard_normal((b, h, w, c)).astype(np.float32) for c in hi_c]
    cin = sum(lo_c) + sum(hi_c)
    wk = (rng.standard_normal((cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    x = nn.concat([nn.upsample2(a) for a in lo] + hi)
    act_np = {'relu6': nn.relu6, 'swish': nn.swish}[act]
    ref = act_np((nn.pointwise(x, wk) * scale + shift).astype(np.float32))

    def packed(cs, rows):
        wt = np.zeros((cout, sum(round_up(c, 4) for c in cs)), np.float32)
        kb = d = 0
        for c in cs:
            wt[:, kb:kb + c] = rows[d:d + c].T
            d += c
            kb += round_up(c, 4)
        return wt
    nlo = sum(lo_c)
    lo_d = [to_dev(a, dev) for a in lo]
    hi_d = [to_dev(a, dev) for a in hi]
    for cfg in (0, 7, 15, 22):
        p = _pw(rt, dev, [(t, c, 'identity') for t, c in zip(lo_d, lo_c)], packed(lo_c, wk[:nlo]), cout, h // 2, w // 2, b, cfg=cfg)
        out = _pw(rt, dev, [(t, c, 'identity') for t, c in zip(hi_d, hi_c)] + [(p, cout, 'up2_add')], packed(hi_c, wk[nlo:]),
                  cout, h, w, b, act, scale, shift, cfg=cfg)
        assert_close(from_dev(out, cout), ref, 3e-5, 'hoisted conv %s cfg %d' % (case, cfg))
    with pytest.raises(rt.YoloretHipError, match='up2_add'):   # wrong position / shape
        _pw(rt, dev, [(p, cout, 'up2_add'), (hi_d[0], hi_c[0], 'identity')], packed(hi_c[:1], wk[nlo:nlo + hi_c[0]]), cout, h, w, b)


def test_compiler_hoists_the_top_down_convs(dev):
    from oracle import model as om, params
    from yoloret_amd import compiler, layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[128, 128, 3]), 'mobilenetv2x75', 3, num_classes=20)
    names = [o.name for o in m.plan.ops]
    assert 'td2_conv_lowres' in names and 'td3_conv_lowres' in names and 'td1_conv_lowres' not in names
    td3 = next(o for o in m.plan.ops if o.name in ('td3_conv', 'td3_head'))     # (round 5: conv + depthwise of a head block are one YR_OP_HEAD op)
    assert td3.cin == 24 and [s.xform for s in td3.srcs] == ['identity', 'up2_add']
    saved = compiler.HOIST_UPSAMPLE
    try:
        compiler.HOIST_UPSAMPLE = False
        plain = yolov3_body(L.Input(shape=[128, 128, 3]), 'mobilenetv2x75', 3, num_classes=20)
    finally:
        compiler.HOIST_UPSAMPLE = saved
    assert plain.plan.total_macs() == m.plan.total_macs()
    assert abs(plain.plan.algorithmic_bytes_per_image() - m.plan.algorithmic_bytes_per_image()) < 1
    P = params.ParamStore(11, 'conditioned')
    x = params.synthetic_images(2, 128, 128)
    ref = om.yolov3_body(P, x, 'mobilenetv2x75', 3, 20)
    for model in (m, plain):
        model.set_weights(P.values)
        for y, r in zip(model(torch.from_numpy(x).to(dev)), ref):
            assert_close(y.cpu().numpy().reshape(r.shape), r, 1e-4, 'hoisted graph')


@pytest.mark.parametrize('case', [(52, 52, 75, 128, 'relu6'), (12, 20, 24, 48, 'none'), (6, 10, 37, 50, 'swish')],
                         ids=['bu3_down', 'rfcr_b3c', 'ragged'])
def test_pointwise_with_pooled_output(dev, case):
    """stride = 2 on a pointwise op: out = MaxPooling2D(2)(act(BN(conv(x)))) written by the conv itself, for the
    LDS-staged and the direct kernel (forced tile shapes); identical to pooling the unfused result."""
    from yoloret_amd import runtime as rt
    h, w, cin, cout, act = case
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    b = 3
    x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
    wk = (rng.standard_normal((cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    act_np = {'relu6': nn.relu6, 'swish': nn.swish, 'none': lambda v: v}[act]
    full = act_np((nn.pointwise(x, wk) * scale + shift).astype(np.float32))
    ref = nn.maxpool(full, 2)
    wt = np.zeros((cout, round_up(cin, 4)), np.float32)
    wt[:, :cin] = wk.T
    xd = to_dev(x, dev)
    keep = [torch.from_numpy(a).to(dev) for a in (wt, scale, shift)]
    outs = []
    for cfg in (0, 2, 9, 15, 17, 23):
        op = rt.new_op(rt.OP_POINTWISE, act)
        op.h, op.w, op.cin, op.cout, op.nsrc, op.stride, op.k = h // 2, w // 2, cin, cout, 1, 2, cfg
        op.src[0] = rt.make_src(xd, c=cin)
        op.wgt, op.scale, op.shift = [t.data_ptr() for t in keep]
        out = torch.full((b, h // 2, w // 2, round_up(cout, 4)), float('nan'), dtype=torch.float32, device=dev)
        op.out, op.out_ld = out.data_ptr(), out.shape[3]
        rt.run_op(op, b)
        torch.cuda.synchronize()
        outs.append(from_dev(out, cout))
        assert_close(outs[-1], ref, 3e-5, 'pooled conv %s cfg %d' % (case, cfg))
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    full_d = _pw(rt, dev, [(xd, cin, 'identity')], wt, cout, h, w, b, act, scale, shift)
    assert np.array_equal(nn.maxpool(from_dev(full_d, cout), 2), outs[0])   # exactly the pooled unfused result


def test_compiler_pools_in_the_producer(dev):
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[128, 128, 3]), 'mobilenetv2x75', 3, num_classes=20)
    pooled = {o.name: (o.h, o.w) for o in m.plan.ops if getattr(o, 'stride', 0) == 2 and o.kind == 2}
    # (the throughput plan runs a down conv as the SECOND output of the launch that also computes the head's y conv: compiler.fuse_stream_pairs)
    pooled.update({o.second_name: (o.gate_out.h, o.gate_out.w) for o in m.plan.ops if o.kind == 2 and (getattr(o, 'reserved0', 0) >> 8) & 1})
    assert set(pooled) == {'bu3_down_conv', 'bu2_down_conv', 'rfcr_b3c'}
    assert pooled['bu3_down_conv'] == (8, 8)
    assert not any(s.xform == 'maxpool2' and s.buf.name.endswith('_pooled') for o in m.plan.ops for s in o.srcs)
    bu2 = next(o for o in m.plan.ops if o.name in ('bu2_conv', 'bu2_head'))
    assert [s.xform for s in bu2.srcs] == ['identity', 'identity']


@pytest.mark.parametrize('name,size', [('mobilenetv2x75', 224), ('mobilenetv2x14', 160)])
def test_depthwise_folded_into_project_is_bit_identical(dev, name, size):
    """compiler.fold_depthwise_into_project (xform 'dw3', opt-in): the projection computes its block's depthwise stage
    in its own loader with dw_kernel's arithmetic, so the logits equal the unfolded plan's bit for bit (stride 1 and 2,
    borders, odd maps: 224/32 = 7, 160/32 = 5)."""
    from yoloret_amd import compiler, layers as L, runtime as rt
    from yoloret_amd.weights import synthetic_images, synthetic_weights
    from yoloret_amd.yolo3.model import yolov3_body
    outs = {}
    for fold in (False, True):
        saved = compiler.FOLD_DW, compiler.FUSE_MBR, compiler.FUSE_MBE
        compiler.FOLD_DW, compiler.FUSE_MBR, compiler.FUSE_MBE = fold, False, False   # (the register-chained block kernels take the same blocks)
        try:
            m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
        finally:
            compiler.FOLD_DW, compiler.FUSE_MBR, compiler.FUSE_MBE = saved
        m.small_batch = 0
        folded = [o for o in m.plan.ops if o.kind == rt.OP_POINTWISE and o.srcs[0].xform == 'dw3']
        assert (len(folded) >= 5) == fold
        if fold:
            assert {o.se_reduced & 0xff for o in folded} == {1, 2}
        m.set_weights(synthetic_weights(m, 7, 'conditioned'))
        x = torch.from_numpy(synthetic_images(3, size, size)).to(dev)
        outs[fold] = [y.cpu().numpy() for y in m(x)]
    import os
    split = os.environ.get('YOLORET_PW_SPLIT', '1') != '0'   # (round 4: the unfolded projections run in the split form - float16 planes on
    for a, b in zip(outs[False], outs[True]):                #  the 16-bit matrix pipe -, the folded ones keep the float32 MFMA: same values,
        if split:                                            #  other roundings; YOLORET_PW_SPLIT=0 restores the bit-for-bit comparison)
            assert_close(a, b, 2e-5, 'dw3-folded plan vs unfolded plan')
        else:
            assert np.array_equal(a, b)


def test_compiler_folds_head_projections_into_their_1x1_consumers(dev):
    """compiler.fold_projection_into_consumers: W_c (s_p W_p d + h_p) = (W_c diag(s_p) W_p) d + W_c h_p.  The 52 x 52 head
    projections (td3 -> bu3_conv; bu3 -> {y, down conv + maxpool}) and bu1 -> y disappear, their consumers read the gated
    depthwise map; accounting unchanged; logits equal the unfolded plan's to rounding and the oracle's within 1e-4
    (reference: code/yolo3/model.py:98-114,296-308; efficientnet.py:517-533)."""
    from oracle import model as om, params
    from yoloret_amd import compiler, layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[128, 128, 3]), 'mobilenetv2x75', 3, num_classes=20)
    names = [o.name for o in m.plan.ops]
    assert not any(n in names for n in ('td3_mb_project', 'bu3_mb_project', 'bu1_mb_project'))
    assert all(n in names for n in ('td1_mb_project', 'td2_mb_project', 'bu2_mb_project'))   # composed MACs would be 1.14x .. 2.3x
    folded = {o.name: o for o in m.plan.ops if getattr(o, 'folded_projection', None)}
    pair = folded['bu3_y']       # bu3's y conv and down conv: one launch with two outputs (compiler.fuse_stream_pairs), both composed
    assert sorted(folded) == ['bu1_y', 'bu3_head', 'bu3_y'] and pair.second_name == 'bu3_down_conv' and all(f.folded_projection for f in pair.fused)      # (bu3_head: bu3_conv + its depthwise, YR_OP_HEAD)
    assert all((o.res if o.kind == 15 else o.gate) is not None and o.cin == o.srcs[0].c and o.srcs[0].buf.name.endswith('_mb_dw') for o in folded.values())
    assert (pair.reserved0 >> 8) & 1    # the pooled store still rides on the (composed) conv
    saved = compiler.FOLD_PROJ
    try:
        compiler.FOLD_PROJ = False
        plain = yolov3_body(L.Input(shape=[128, 128, 3]), 'mobilenetv2x75', 3, num_classes=20)
    finally:
        compiler.FOLD_PROJ = saved
    assert len(plain.plan.ops) == len(m.plan.ops) + 3
    assert plain.plan.total_macs() == m.plan.total_macs()
    assert abs(plain.plan.algorithmic_bytes_per_image() - m.plan.algorithmic_bytes_per_image()) < 1
    P = params.ParamStore(12, 'conditioned')
    x = params.synthetic_images(2, 128, 128)
    ref = om.yolov3_body(P, x, 'mobilenetv2x75', 3, 20)
    got = []
    for model in (m, plain):
        model.set_weights(P.values)
        ys = [y.cpu().numpy() for y in model(torch.from_numpy(x).to(dev))]
        got.append(ys)
        for y, r in zip(ys, ref):
            assert_close(y.reshape(r.shape), r, 1e-4, 'graph with folded projections')
    for a, b in zip(*got):
        assert_close(a, b, 3e-5, 'folded vs unfolded plan')


@pytest.mark.parametrize('model_name,dtype', [('mobilenetv2x75', 'f32'), ('efficientnetb0', 'f32')])
def test_compiler_folds_the_rfcr_weighted_sum(dev, model_name, dtype):
    """compiler.fold_weighted_sum: a0 up2(W1 x1) + a1 W2 x2 + a2 maxpool2(W3 x3) + a3 W4 maxpool4(x4) as ONE pointwise conv over
    [up2(x1) | x2 | maxpool2(W3 x3) | maxpool4(x4)] with the weights [a0 W1 | a1 W2 | a2 I | a3 W4] (reference:
    code/yolo3/model.py:117-137 WeightedSum, :146-168 rfcr_module).  The alphas are unconstrained in the reference: the test
    makes the one behind the max-pool NEGATIVE (it must not be pulled through the maximum)."""
    from oracle import model as om, params
    from yoloret_amd import compiler, layers as L, runtime as rt
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[128, 128, 3]), model_name, 3, num_classes=20)
    names = [o.name for o in m.plan.ops]
    assert 'rfcr_b3c' in names and not any(n in names for n in ('rfcr_b1c', 'rfcr_b2c', 'rfcr_b4c'))
    ws = next(o for o in m.plan.ops if o.name == 'rfcr_wsum')
    assert ws.kind == rt.OP_POINTWISE and getattr(ws, 'folded_wsum', False)
    assert [s.xform for s in ws.srcs] == ['up2', 'identity', 'identity', 'maxpool4'] and ws.srcs[2].buf.name == 'rfcr_b3c_pooled'
    saved = compiler.FOLD_WSUM
    try:
        compiler.FOLD_WSUM = False
        plain = yolov3_body(L.Input(shape=[128, 128, 3]), model_name, 3, num_classes=20)
    finally:
        compiler.FOLD_WSUM = saved
    assert len(plain.plan.ops) == len(m.plan.ops) + 3 and any(o.kind == rt.OP_WSUM for o in plain.plan.ops)
    assert plain.plan.total_macs() == m.plan.total_macs()
    assert abs(plain.plan.algorithmic_bytes_per_image() - m.plan.algorithmic_bytes_per_image()) < 1
    P = params.ParamStore(5, 'conditioned')
    P.values['rfcr_wsum/alpha'] = np.array([0.8, 1.1, -0.9, 0.6], np.float32)
    x = params.synthetic_images(2, 128, 128)
    ref = om.yolov3_body(P, x, model_name, 3, 20)
    got = []
    for model in (m, plain):
        model.set_weights(P.values)
        ys = [y.cpu().numpy() for y in model(torch.from_numpy(x).to(dev))]
        got.append(ys)
        for y, r in zip(ys, ref):
            assert_close(y.reshape(r.shape), r, 1e-4, 'graph with the folded weighted sum')
    for a, b in zip(*got):
        assert_close(a, b, 3e-5, 'folded vs literal weighted sum')
