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
    td3 = next(o for o in m.plan.ops if o.name == 'td3_conv')
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
