"""YR_OP_HEAD (headblock.hip) and the SE tail (se_tail.h) through the C-ABI against the NumPy oracle.

HEAD = the first two thirds of a detection-head block in one launch (reference code/yolo3/model.py:91-115: Conv2D 1x1 + BN + ReLU6
-> MBConvBlock's depthwise 3x3 + BN + Swish -> the SE block's squeeze / FC pair, efficientnet.py:406-438,467-536); the conv runs in
the split form (two float16 planes per operand), hence the 5e-5 bar of the split-form block kernels (tests/test_gpu_mbr.py)."""
import ctypes
import zlib

import numpy as np
import pytest
import torch

from oracle import nn
from tests.util import assert_close, from_dev, round_up, to_dev
from tests.test_gpu_ops import _act_np, _dev_vec, _src_dims, _xform_np

pytestmark = pytest.mark.gpu
TOL = 5e-5


def _rt():
    from yoloret_amd import runtime as rt
    return rt


def se_pack(w1, b1, w2, b2):
    """The `se_w` layout of include/yoloret_hip.h: W1 [ldc][R4] | W2 [R][ldc] | b1 [R4] | b2 [ldc]."""
    c, r = w1.shape
    ldc, r4 = round_up(c, 4), round_up(r, 4)
    w1p = np.zeros((ldc, r4), np.float32)
    w1p[:c, :r] = w1
    w2p = np.zeros((r, ldc), np.float32)
    w2p[:, :c] = w2
    b1p = np.zeros(r4, np.float32)
    b1p[:r] = b1
    b2p = np.zeros(ldc, np.float32)
    b2p[:c] = b2
    return np.concatenate([w1p.ravel(), w2p.ravel(), b1p, b2p])


def se_params(rng, c, r):
    w1 = (rng.standard_normal((c, r)) * np.sqrt(2.0 / c)).astype(np.float32)
    b1 = rng.normal(0, 0.1, r).astype(np.float32)
    w2 = (rng.standard_normal((r, c)) * np.sqrt(2.0 / r)).astype(np.float32)
    b2 = rng.normal(0, 0.1, c).astype(np.float32)
    return w1, b1, w2, b2


def walk_ok(segs, f, pre, gated, conv_act):
    """compiler.fuse_head_blocks' rule for the walking form"""
    nk = sum((c + 31) // 32 for c, _ in segs)
    nt = 2 if nk <= 4 else 1
    return (all(xf == 'identity' for _, xf in segs) and len(segs) <= 3 and nk <= 7 and f % 16 == 0 and (f // 16) % nt == 0 and (f // 16 // nt) % 4 == 0
            and conv_act in ('relu6', 'none') and not (gated and pre))


def stream_ok(h, segs, f, pre, gated, conv_act):
    """compiler.fuse_head_blocks' rule for the weight-streaming form (headstream.hip)"""
    nk = sum((c + 31) // 32 for c, _ in segs)
    return (all(xf in ('identity', 'maxpool2') for _, xf in segs) and len(segs) <= 3 and f % 32 == 0 and nk <= (7 if h >= 20 else 11)
            and conv_act in ('relu6', 'none') and not (gated and (len(segs) != 1 or segs[0][1] != 'identity')))


def se_gate_ref(y, w1, b1, w2, b2):
    mean = nn.mean_hw(y.astype(np.float64)).astype(np.float64)
    hid = mean @ w1.astype(np.float64) + b1
    hid = hid / (1.0 + np.exp(-hid))
    return 1.0 / (1.0 + np.exp(-(hid @ w2.astype(np.float64) + b2)))


def run_head(dev, rng, b, h, w, segs, f, pre=False, gated=False, se=None, conv_act='relu6', dw_act='swish', tail=True, scale_x=1.0, cfg=0, tile=False, form=None, xgen=None):
    """segs: [(channels, xform)] of the conv's concatenated sources; pre: an up-sampled pre-BN addend (YR_X_UP2_ADD);
    gated: SE gate on the (single identity) source; se: hidden width R (None: no squeeze-excite sums at all); tail: the op also
    runs the FC pair (False: sums only); tile: every image of the batch is the same image (drawn once); form: 'walk' (headwalk.hip),
    'dma' (headblock.hip, LDS-direct), 'pws' (headblock.hip, register-staged gathers) or None = what the compiler would pick.
    -> (map, gate | None)"""
    rt = _rt()
    nb = b
    if tile:
        b = 1
    srcs_np, srcs_dev = [], []
    for c, xf in segs:
        sh, sw = _src_dims(h, w, xf)
        a = np.repeat((rng.standard_normal((b, sh, sw, c)) * scale_x).astype(np.float32) if xgen is None else xgen(rng, (b, sh, sw, c)), nb // b, axis=0)
        srcs_np.append(a)
        srcs_dev.append(to_dev(a, dev))
    cin = sum(c for c, _ in segs)
    wk = (rng.standard_normal((cin, f)) * np.sqrt(2.0 / cin)).astype(np.float32)
    kp = sum(round_up(c, 4) for c, _ in segs)
    wt = np.zeros((f, kp), np.float32)
    d = kb = 0
    for c, _ in segs:
        wt[:, kb:kb + c] = wk[d:d + c].T
        d += c
        kb += round_up(c, 4)
    x = nn.concat([_xform_np(a, xf) for a, (_, xf) in zip(srcs_np, segs)]).astype(np.float64)
    gate_np = None
    if gated:
        gate_np = np.repeat(rng.uniform(0.1, 1.0, (b, 1, 1, cin)).astype(np.float32), nb // b, axis=0)
        x = (gate_np * x.astype(np.float32)).astype(np.float64)     # (the kernel forms the product in float32)
    e = x @ wk.astype(np.float64)
    pre_np = None
    if pre:
        pre_np = np.repeat((rng.standard_normal((b, h // 2, w // 2, f)) * scale_x).astype(np.float32), nb // b, axis=0)
        e = e + nn.upsample2(pre_np).astype(np.float64)
    b = nb
    cs = rng.uniform(0.5, 1.5, f).astype(np.float32)
    ch = rng.normal(0, 0.3, f).astype(np.float32)
    e = _act_np(e * cs + ch, conv_act)
    dk = (rng.standard_normal((3, 3, f)) * np.sqrt(2.0 / 9)).astype(np.float32)
    ds = rng.uniform(0.5, 1.5, f).astype(np.float32)
    dh = rng.normal(0, 0.3, f).astype(np.float32)
    y = _act_np(nn.depthwise(e, dk.astype(np.float64), 1, 'same') * ds + dh, dw_act)
    ldf = round_up(f, 4)
    dwp = np.zeros((10, ldf), np.float32)
    dwp[:9, :f] = (dk.reshape(9, f) * ds[None]).astype(np.float32)
    dwp[9, :f] = dh

    out = torch.full((b, h, w, ldf), float('nan'), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_HEAD, dw_act)
    op.h, op.w, op.cin, op.cout, op.stride = h, w, cin, f, 1
    if form is None:
        form = 'walk' if walk_ok(segs, f, pre, gated, conv_act) else 'dma' if all(xf in ('identity', 'up2') for _, xf in segs) and len(segs) <= 3 else 'pws'
    op.k = 3 | rt.ACT[conv_act] << 8 | cfg << 16 | {'walk': 0x40, 'stream': 0x60, 'dma': 0x80, 'pws': 0}[form]
    from yoloret_amd.compiler import head_pack
    if form == 'dma':
        wt = head_pack(wt, [c for c, _ in segs])
    elif form in ('walk', 'stream'):      # planes with the conv's BN scale folded in; YR_OP_MBR's tap table [T][11][16]
        wt = head_pack((wt * cs[:, None]).astype(np.float32), [c for c, _ in segs])
        t16 = f // 16
        tab = np.zeros((t16, 11, 16), np.float32)
        tab[:, :9] = (dk.reshape(9, f) * ds[None]).astype(np.float32).reshape(9, t16, 16).transpose(1, 0, 2)
        tab[:, 9], tab[:, 10] = dh.reshape(t16, 16), ch.reshape(t16, 16)
        dwp = tab
    n = 0
    for t, (c, xf) in zip(srcs_dev, segs):
        op.src[n] = rt.make_src(t, c=c, xform=xf)
        n += 1
    keep = [_dev_vec(wt, dev), _dev_vec(cs, dev), _dev_vec(ch, dev), _dev_vec(dwp, dev)]
    if pre:
        pd = to_dev(pre_np, dev, fill=0.0)
        keep.append(pd)
        op.src[n] = rt.make_src(pd, c=f, xform='up2_add')
        n += 1
    op.nsrc = n
    op.wgt, op.scale, op.shift, op.wgt2 = [k.data_ptr() for k in keep[:4]]
    if gated:
        g = to_dev(gate_np.reshape(b, 1, 1, cin), dev)
        keep.append(g)
        op.res, op.res_ld = g.data_ptr(), g.shape[3]
    op.out, op.out_ld = out.data_ptr(), ldf
    sums = gate_out = None
    if se is not None:
        nsy, nsx = ctypes.c_int32(), ctypes.c_int32(1)
        if form == 'stream':
            rt.check(rt.lib().yr_head_stream_rows(h, w, ctypes.byref(nsy)))
            from yoloret_amd.compiler import head_stream_rows
            assert nsy.value == head_stream_rows(h, w)
        elif form == 'walk':
            rt.check(rt.lib().yr_head_walk_rows(h, w, ctypes.byref(nsy)))
        else:
            rt.check(rt.lib().yr_head_regions(h, w, ctypes.byref(nsy), ctypes.byref(nsx)))
        rows = nsy.value * nsx.value
        sums = torch.full((b, rows, ldf), float('nan'), dtype=torch.float32, device=dev)
        op.gate, op.gate_ld, op.se_reduced = sums.data_ptr(), ldf, rows
        sp = se_params(rng, f, se)
        if tail:
            gate_out = torch.full((b, ldf), float('nan'), dtype=torch.float32, device=dev)
            keep += [_dev_vec(se_pack(*sp), dev), torch.zeros(b, dtype=torch.int32, device=dev)]
            op.gate_out, op.gate_out_ld, op.se_hidden = gate_out.data_ptr(), ldf, se
            op.se_w, op.sync = keep[-2].data_ptr(), keep[-1].data_ptr()
    for rep in range(2):      # twice: the arrival counters must be back at zero after a launch
        rt.run_op(op, b)
    torch.cuda.synchronize()
    got = from_dev(out, f)
    assert_close(got, y, TOL, 'head %s -> %d' % (segs, f))
    if se is not None:
        s = from_dev(sums, f).astype(np.float64).sum(axis=1)
        assert_close(s / (h * w), y.mean(axis=(1, 2)), TOL, 'head: squeeze-excite sums')
        if tail:
            assert int(keep[-1].abs().sum().item()) == 0, 'arrival counters were not reset'
            assert_close(from_dev(gate_out, f), se_gate_ref(y, *sp).reshape(b, f), TOL, 'head: squeeze-excite gate (SE tail)')
    run_head.last = dict(x=x, wk=wk, cs=cs, ch=ch, dk=dk, ds=ds, dh=dh, y=y, srcs=srcs_np, segs=segs, conv_act=conv_act, dw_act=dw_act)   # (tests/test_gpu_split_range.py)
    return got, (from_dev(gate_out, f) if gate_out is not None else None)


HEAD_CASES = [
    # (h, w, segs, F, pre, gated, R)                                           the six head blocks of MobileNetV2 x0.75 @416
    (13, 13, [(120, 'identity'), (96, 'maxpool2')], 512, False, False, 128),   # td1
    (26, 26, [(72, 'identity'), (96, 'identity')], 256, True, False, 64),      # td2 (256 up-sampled channels hoisted: up2_add)
    (52, 52, [(24, 'identity')], 128, True, False, 32),                        # td3
    (52, 52, [(128, 'identity')], 128, False, True, 32),                       # bu3 (reads td3's map through its SE gate)
    (26, 26, [(128, 'identity'), (75, 'identity')], 256, False, False, 64),    # bu2
    (13, 13, [(256, 'identity'), (75, 'identity')], 512, False, False, 128),   # bu1
    # other maps: 640 / 512 inputs, ragged widths, channel counts that are no multiple of 16, one-region and many-region splits
    (20, 20, [(232, 'identity'), (96, 'maxpool2')], 512, False, False, 128),
    (40, 40, [(136, 'identity'), (96, 'identity')], 256, True, False, 64),
    (80, 80, [(48, 'identity')], 128, True, False, 32),
    (16, 16, [(160, 'identity'), (37, 'up2')], 84, False, False, 6),
    (7, 5, [(40, 'identity')], 20, False, False, 1),
    (104, 104, [(24, 'identity')], 144, False, False, 6),                      # an SE-EfficientNet stage-2 block in float32
    (9, 31, [(16, 'maxpool4'), (20, 'identity')], 36, False, False, 9),
    # more shapes: three sources, a partial last quad (75 of 76), a map of 15 columns (two strips, one column), four chunks from two sources
    (15, 15, [(40, 'identity'), (75, 'identity'), (64, 'identity')], 64, False, False, 16),
    (15, 15, [(24, 'identity'), (75, 'identity')], 128, False, False, 16),
    (28, 30, [(160, 'identity')], 128, True, False, 32),
    (13, 13, [(96, 'identity'), (128, 'identity')], 64, False, False, 8),
]


@pytest.mark.parametrize('form', ['stream', 'walk', 'dma', 'pws'])
@pytest.mark.parametrize('case', HEAD_CASES, ids=[str(i) for i in range(len(HEAD_CASES))])
def test_head_block(dev, case, form):
    """every case in every form that takes it (the compiler picks stream > walk > dma > pws)"""
    h, w, segs, f, pre, gated, r = case
    if form == 'stream' and not stream_ok(h, segs, f, pre, gated, 'relu6'):
        pytest.skip('shape not built in the weight-streaming form')
    if form == 'walk' and not walk_ok(segs, f, pre, gated, 'relu6'):
        pytest.skip('shape not built in the walking form')
    if form == 'dma' and not (all(xf in ('identity', 'up2') for _, xf in segs) and len(segs) <= 3):
        pytest.skip('pooled sources stay on the register-staged form')
    rng = np.random.default_rng(zlib.crc32(str(case).encode()))
    run_head(dev, rng, 3, h, w, segs, f, pre=pre, gated=gated, se=r, form=form, tail=form != 'stream')


def test_head_stream_form_variants_and_batch_independence(dev):
    """The weight-streaming form: no squeeze-excite sums at all, ReLU6 / no activation, a conv without activation, a pooled source
    beside two identity ones, 11 chunks at one row per wave; an image of a batch equals the image run alone (map and sums)."""
    rng = np.random.default_rng(7)
    run_head(dev, rng, 2, 13, 13, [(120, 'identity')], 96, se=None, dw_act='relu6', form='stream')
    run_head(dev, rng, 2, 26, 26, [(72, 'identity'), (96, 'identity')], 256, pre=True, se=64, tail=False, conv_act='none', dw_act='none', form='stream')
    run_head(dev, rng, 2, 19, 33, [(40, 'identity'), (75, 'identity'), (64, 'maxpool2')], 64, se=16, tail=False, form='stream')
    run_head(dev, rng, 2, 13, 13, [(256, 'identity'), (96, 'identity')], 512, se=128, tail=False, form='stream')     # 11 chunks
    run_head(dev, rng, 3, 45, 61, [(128, 'identity')], 128, gated=True, se=32, tail=False, form='stream')             # odd size, many strips / segments
    # tiny maps (a 64 x 64 input): fewer rows than waves, fewer columns than a strip
    run_head(dev, rng, 3, 8, 8, [(24, 'identity')], 128, pre=True, se=32, tail=False, form='stream')
    run_head(dev, rng, 3, 4, 4, [(72, 'identity'), (96, 'identity')], 256, pre=True, se=64, tail=False, form='stream')
    run_head(dev, rng, 3, 2, 2, [(120, 'identity'), (96, 'maxpool2')], 512, se=128, tail=False, form='stream')
    run_head(dev, rng, 3, 2, 2, [(256, 'identity'), (75, 'identity')], 512, se=128, tail=False, form='stream')
    for case in [(52, 52, [(24, 'identity')], 128, True, 32), (13, 13, [(120, 'identity'), (96, 'maxpool2')], 512, False, 128)]:
        h, w, segs, f, pre, r = case
        one, _ = run_head(dev, np.random.default_rng(12), 1, h, w, segs, f, pre=pre, se=r, tail=False, form='stream')
        many, _ = run_head(dev, np.random.default_rng(12), 5, h, w, segs, f, pre=pre, se=r, tail=False, tile=True, form='stream')
        for i in range(5):
            assert np.array_equal(many[i], one[0]), 'image %d of the batch differs from the image run alone' % i
    # how many workgroups share an (image, strip, segment) follows the launch's size (1 image: four, 110 images of 13 x 13: one) - and changes nothing
    h, w, segs, f, pre, r = (13, 13, [(120, 'identity'), (96, 'maxpool2')], 512, False, 128)
    one, _ = run_head(dev, np.random.default_rng(13), 1, h, w, segs, f, pre=pre, se=r, tail=False, form='stream')
    many, _ = run_head(dev, np.random.default_rng(13), 110, h, w, segs, f, pre=pre, se=r, tail=False, tile=True, form='stream')
    assert np.array_equal(many[0], one[0]) and np.array_equal(many[109], one[0])


def test_head_block_variants(dev):
    """sums without the tail (an SE_FC op finishes them), no squeeze-excite at all, other activations, forced cout tiles."""
    rng = np.random.default_rng(5)
    run_head(dev, rng, 2, 26, 26, [(72, 'identity'), (96, 'identity')], 256, se=64, tail=False)
    run_head(dev, rng, 2, 13, 13, [(120, 'identity')], 96, se=None, dw_act='relu6')
    run_head(dev, rng, 2, 26, 26, [(80, 'identity')], 480, se=20, conv_act='swish')
    a = run_head(dev, np.random.default_rng(6), 2, 13, 13, [(120, 'identity')], 128, se=32, cfg=1)
    b = run_head(dev, np.random.default_rng(6), 2, 13, 13, [(120, 'identity')], 128, se=32, cfg=4)
    # the map does not depend on the cout tiles per workgroup; the sums' grouping (row groups per workgroup) does - which is why a
    # plan never varies it (k bits 16-23 stay 0: the library's choice is a function of the shape)
    assert np.array_equal(a[0], b[0]), 'the cout tiles per workgroup must not change the map'
    assert_close(a[1], b[1], 1e-6, 'gate under another cout tiling')


def test_head_block_batch_independent(dev):
    """An image's map and gate do not depend on the batch it runs in, nor on which workgroup completes it: the regions are a
    function of the shape, the sums are added in index order."""
    for case in [(52, 52, [(24, 'identity')], 128, True, 32), (13, 13, [(120, 'identity'), (96, 'maxpool2')], 512, False, 128)]:
        h, w, segs, f, pre, r = case
        one, g1 = run_head(dev, np.random.default_rng(12), 1, h, w, segs, f, pre=pre, se=r)
        many, gm = run_head(dev, np.random.default_rng(12), 7, h, w, segs, f, pre=pre, se=r, tile=True)
        for i in range(7):
            assert np.array_equal(many[i], one[0]) and np.array_equal(gm[i], g1[0]), 'image %d of the batch differs from the image run alone' % i


def test_se_tail_of_depthwise(dev):
    """The SE form of dw_kernel with the tail: map, sums and gate in one launch (what SE_FC with k = h*w would have written)."""
    rt = _rt()
    from yoloret_amd.compiler import dw_se_geometry
    for (k, s, h, w, c, r) in [(3, 1, 13, 13, 512, 128), (3, 1, 26, 26, 256, 64), (5, 2, 16, 16, 40, 10), (3, 2, 27, 27, 96, 4), (5, 1, 9, 7, 75, 6)]:
        rng = np.random.default_rng(k * 100 + c)
        b = 3
        x = rng.standard_normal((b, h, w, c)).astype(np.float32)
        wk = (rng.standard_normal((k, k, c)) * np.sqrt(2.0 / (k * k))).astype(np.float32)
        scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
        shift = rng.normal(0, 0.3, c).astype(np.float32)
        ref = _act_np((nn.depthwise(x, wk, s, 'same') * scale + shift).astype(np.float32), 'swish')
        ldc = round_up(c, 4)
        ho, wo = ref.shape[1:3]
        xt = 4 if s == 1 else 2
        rows = dw_se_geometry(ho * ((wo + xt - 1) // xt), ldc // 4)[2]
        xd = to_dev(x, dev, fill=0.0)
        wd = np.zeros((k * k, ldc), np.float32)
        wd[:, :c] = wk.reshape(k * k, c)
        sp = se_params(rng, c, r)
        keep = [_dev_vec(wd, dev), _dev_vec(scale, dev, ldc), _dev_vec(shift, dev, ldc), _dev_vec(se_pack(*sp), dev),
                torch.zeros(b, dtype=torch.int32, device=dev)]
        out = torch.full((b, ho, wo, ldc), float('nan'), dtype=torch.float32, device=dev)
        sums = torch.full((b, rows, ldc), float('nan'), dtype=torch.float32, device=dev)
        gate = torch.full((b, ldc), float('nan'), dtype=torch.float32, device=dev)
        op = rt.new_op(rt.OP_DEPTHWISE, 'swish')
        op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = ho, wo, c, c, k, s, 1
        op.src[0] = rt.make_src(xd, c=c)
        op.wgt, op.scale, op.shift = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
        op.out, op.out_ld = out.data_ptr(), ldc
        op.gate, op.gate_ld, op.se_reduced = sums.data_ptr(), ldc, rows
        op.gate_out, op.gate_out_ld, op.se_hidden = gate.data_ptr(), ldc, r
        op.se_w, op.sync = keep[3].data_ptr(), keep[4].data_ptr()
        for rep in range(3):
            rt.run_op(op, b)
        torch.cuda.synchronize()
        assert_close(from_dev(out, c), ref, 2e-5, 'depthwise (SE tail)')
        assert int(keep[4].abs().sum().item()) == 0
        assert_close(from_dev(gate, c), se_gate_ref(ref, *sp).reshape(b, c), 2e-5, 'SE tail gate')


# ---------------------------------------------------------------------------------------------------------------------
# YR_OP_HEAD of the 16-bit plans (headwalk_h.hip): sources and conv weights exactly representable in the 16-bit type, float32 from
# the accumulator on, ONE rounding at the store - the bar of tests/test_gpu_narrow.py (the float64 value of the same expression
# rounded once: half an ulp of the type + float32 noise).  The F-wide conv output is NOT rounded in between (the unfused pair of
# launches rounds it: the fused op is the more accurate one).

def run_head16(dev, dt, rng, b, h, w, segs, f, pre=False, gated=False, se=True, conv_act='relu6', dw_act='swish', tile=False):
    from tests.util import assert_rounded_once, from_dev16, q16, to_dev16
    from yoloret_amd.compiler import head_pack16
    rt = _rt()
    did = rt.dtype_id(dt)
    nb = b
    if tile:
        b = 1
    srcs_np, srcs_dev = [], []
    for c in segs:
        a = np.repeat(q16(rng.standard_normal((b, h, w, c)), dt), nb // b, axis=0)
        srcs_np.append(a)
        srcs_dev.append(to_dev16(a, dev, dt))
    cin = sum(segs)
    wk = q16(rng.standard_normal((cin, f)) * np.sqrt(2.0 / cin), dt)
    kp = sum(round_up(c, 8) for c in segs)
    wt = np.zeros((f, kp), np.float32)
    d = kb = 0
    for c in segs:
        wt[:, kb:kb + c] = wk[d:d + c].T
        d += c
        kb += round_up(c, 8)
    x = nn.concat(srcs_np).astype(np.float64)
    wk64 = wk.astype(np.float64)
    gate_np = None
    if gated:
        gate_np = np.repeat(rng.uniform(0.1, 1.0, (b, 1, 1, cin)).astype(np.float32), nb // b, axis=0)
        # the kernel folds the gate into its stationary weights: w * g in float32, rounded once to the operand type
        wg = q16(wk[None] * gate_np.reshape(nb, cin, 1), dt).astype(np.float64)     # [B][cin][f]
        e = np.einsum('bhwc,bcf->bhwf', x, wg)
    else:
        e = x @ wk64
    pre_np = None
    if pre:
        pre_np = np.repeat(rng.standard_normal((b, h // 2, w // 2, f)).astype(np.float32), nb // b, axis=0)
        e = e + nn.upsample2(pre_np).astype(np.float64)
    b = nb
    cs = rng.uniform(0.5, 1.5, f).astype(np.float32)
    ch = rng.normal(0, 0.3, f).astype(np.float32)
    e = _act_np(e * cs + ch, conv_act)
    dk = (rng.standard_normal((3, 3, f)) * np.sqrt(2.0 / 9)).astype(np.float32)
    ds = rng.uniform(0.5, 1.5, f).astype(np.float32)
    dh = rng.normal(0, 0.3, f).astype(np.float32)
    y = _act_np(nn.depthwise(e, dk.astype(np.float64), 1, 'same') * ds + dh, dw_act)
    t16 = f // 16
    tab = np.zeros((t16, 11, 16), np.float32)
    tab[:, :9] = (dk.reshape(9, f) * ds[None]).astype(np.float32).reshape(9, t16, 16).transpose(1, 0, 2)
    tab[:, 9], tab[:, 10] = dh.reshape(t16, 16), ch.reshape(t16, 16)
    frag = head_pack16(wt, segs)
    wd = torch.from_numpy(rt.to_bits16(frag, dt).view(np.int16)).to(dev)
    ldf = round_up(f, 8)
    out = to_dev16(np.full((b, h, w, ldf), np.nan, np.float32), dev, dt)
    op = rt.new_op(rt.OP_HEAD, dw_act)
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.stride = h, w, cin, f, 1
    op.k = 3 | rt.ACT[conv_act] << 8 | 0x40
    n = 0
    for t, c in zip(srcs_dev, segs):
        op.src[n] = rt.make_src(t, c=c, xform='identity')
        n += 1
    keep = [wd, _dev_vec(cs, dev), _dev_vec(tab, dev)]
    if pre:
        pd = to_dev(pre_np, dev, fill=0.0)
        keep.append(pd)
        op.src[n] = rt.make_src(pd, c=f, xform='up2_add')
        n += 1
    op.nsrc = n
    op.wgt, op.scale, op.wgt2 = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
    if gated:
        g = to_dev(gate_np.reshape(b, 1, 1, cin), dev)
        keep.append(g)
        op.res, op.res_ld = g.data_ptr(), g.shape[3]
    op.out, op.out_ld = out.data_ptr(), ldf
    sums = None
    if se:
        rows = ctypes.c_int32()
        rt.check(rt.lib().yr_head_walk_rows(h, w, ctypes.byref(rows)))
        sums = torch.full((b, rows.value, ldf), float('nan'), dtype=torch.float32, device=dev)
        op.gate, op.gate_ld, op.se_reduced = sums.data_ptr(), ldf, rows.value
    rt.run_op(op, b)
    torch.cuda.synchronize()
    got = from_dev16(out, dt, f)
    # float32 noise of two chained stages (conv accumulation over up to 256 channels, nine taps): 1e-4 of slack beside the half ulp
    assert_rounded_once(got, y, dt, 'head16 %s %s -> %d' % (dt, segs, f), slack=1e-4)
    s = None
    if se:
        s = from_dev(sums, f)
        assert_close(s.astype(np.float64).sum(axis=1) / (h * w), got.astype(np.float64).mean(axis=(1, 2)), 2e-5, 'head16: squeeze-excite sums of the stored values')
    return got, s


HEAD16_CASES = [
    # (h, w, segs, F, pre, gated)                                       the walkable head blocks of the EfficientNet configurations
    (52, 52, [40], 128, True, False),            # td3 (B0 @416)
    (52, 52, [128], 128, False, True),           # bu3
    (26, 26, [112, 96], 256, True, False),       # td2: 4 + 3 chunks
    (26, 26, [128, 75], 256, False, False),      # bu2: a partial last octet (75 of 80)
    (40, 40, [136, 96], 256, True, False),       # td2 (B3 @640): 5 + 3 = 8 chunks
    (80, 80, [48], 128, True, False),            # td3 (B3)
    # other shapes: three sources, ragged widths (two strips with one column in the second), one chunk, F = 512, no ReLU6
    (15, 15, [40, 75, 64], 128, False, False),
    (13, 13, [96, 128], 512, False, False),
    (28, 30, [160], 128, True, False),
    (7, 5, [24], 128, False, False),
    (20, 20, [200], 256, False, True),
]


@pytest.mark.parametrize('dt', ['bf16', 'f16'])
@pytest.mark.parametrize('case', HEAD16_CASES, ids=[str(i) for i in range(len(HEAD16_CASES))])
def test_head_block_16bit(dev, dt, case):
    h, w, segs, f, pre, gated = case
    rng = np.random.default_rng(zlib.crc32((str(case) + dt).encode()))
    run_head16(dev, dt, rng, 3, h, w, segs, f, pre=pre, gated=gated)


def test_head_block_16bit_variants(dev):
    rng = np.random.default_rng(77)
    run_head16(dev, 'bf16', rng, 2, 26, 26, [72], 128, se=False, dw_act='relu6')
    run_head16(dev, 'f16', rng, 2, 13, 13, [120, 40], 256, conv_act='none')


def test_head_block_16bit_batch_independent(dev):
    """an image's stored map and its squeeze-excite sums do not depend on the batch it runs in (rows of the sums = strips x row
    segments of the SHAPE)."""
    for dt in ('bf16', 'f16'):
        one, s1 = run_head16(dev, dt, np.random.default_rng(12), 1, 52, 52, [40], 128, pre=True)
        many, sm = run_head16(dev, dt, np.random.default_rng(12), 7, 52, 52, [40], 128, pre=True, tile=True)
        for i in range(7):
            assert np.array_equal(many[i], one[0]) and np.array_equal(sm[i], s1[0]), 'image %d of the batch differs from the image run alone' % i
