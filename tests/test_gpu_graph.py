"""Whole-graph parity through the drop-in surface (yolov3_body -> Model -> yolo_eval) against
the NumPy oracle on the same seeded weights and images.

Bar (BASELINE.json north_star): fp32 logits within 1e-4 - applied as |a-b| <= 1e-4*max(1,|b|)
 - and a bit-exact NMS index set.  The NMS stage is bit-exact on identical inputs
(test_gpu_postprocess.py); end to end the decisions also depend on ~1e-6 logit differences, so
here the oracle's post-processing is re-run on the GPU's own logits and must match exactly, and
the agreement with the oracle's end-to-end detections is measured and bounded."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import cpost
from oracle import model as om
from oracle import params
from tests.util import ANCHORS, assert_close

pytestmark = pytest.mark.gpu


def _build(model_name, hw, num_classes, seed=1234, recipe='conditioned'):
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[hw[0], hw[1], 3]), model_name, 3, num_classes=num_classes)
    P = params.ParamStore(seed, recipe)
    return m, P


def _run_both(dev, model_name, hw, b, num_classes=20, recipe='conditioned'):
    m, P = _build(model_name, hw, num_classes, recipe=recipe)
    x = params.synthetic_images(b, hw[0], hw[1])
    ref = om.yolov3_body(P, x, model_name, 3, num_classes)
    m.set_weights(P.values)  # the oracle's walk created every parameter the product needs
    ys = m(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    return m, x, ref, ys


@pytest.mark.parametrize('model_name,hw', [('mobilenetv2x75', (64, 64)), ('mobilenetv2x14', (64, 96)),
                                           ('efficientnetb0', (64, 64)), ('efficientnetb3', (96, 64)),
                                           ('efficientnetb0-lite', (64, 64))])
def test_logits_small(dev, model_name, hw):
    _, _, ref, ys = _run_both(dev, model_name, hw, 3)
    for i, (y, r) in enumerate(zip(ys, ref)):
        assert tuple(y.shape) == r.shape
        assert_close(y.cpu().numpy(), r, 1e-4, '%s y%d' % (model_name, i + 1))


def test_logits_416_and_detections(dev):
    """BASELINE config 2's model at full resolution, small batch."""
    from yoloret_amd.yolo3.model import yolo_eval
    b = 2
    m, x, ref, ys = _run_both(dev, 'mobilenetv2x75', (416, 416), b)
    worst = 0.0
    for i, (y, r) in enumerate(zip(ys, ref)):
        assert tuple(y.shape) == (b, 416 // (32 >> i), 416 // (32 >> i), 3, 25)
        worst = max(worst, assert_close(y.cpu().numpy(), r, 1e-4, 'y%d' % (i + 1)))
    print('max scaled logit error vs oracle: %.2e' % worst)
    res = yolo_eval(ys, ANCHORS, 3, 20, (416, 416), max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
    assert len(res) == b
    _check_detections_with_margins(ys, ref, res, (416, 416))


# A decision (score > thr, IoU > thr, which candidate is popped next) whose margin is below this cannot be required
# to come out the same in two float32 implementations whose logits are only required to agree to 1e-4: the logit bar
# bounds score differences by ~5e-5 and IoU differences by a few 1e-4.  Measured margins of the flips are printed.
DECISION_NOISE = 1e-4


def _check_detections_with_margins(ys, ref, res, hw, thr=0.2, iou=0.5, num_classes=20):
    """(1) the oracle's post-processing of the GPU's own logits equals the GPU's detections bit for bit;
    (2) end to end against the oracle's logits every (image, class) NMS problem must return the SAME picks unless one
    of the oracle's decisions on that problem sat within DECISION_NOISE of a threshold - a flat
    percentage would hide a real bug, a margin cannot."""
    from oracle import postprocess as pp
    agree = total = 0
    flips, min_margin_all = [], np.inf
    for i in range(len(res)):
        gb, gs, gc = [t.cpu().numpy() for t in res[i]]
        ob, os_, oc, gi = cpost.yolo_eval([y[i].cpu().numpy() for y in ys], ANCHORS, 3, num_classes, hw, 20, thr, iou)
        assert np.array_equal(gb, ob) and np.array_equal(gs, os_) and np.array_equal(gc, oc)
        boxes_r, scores_r = pp.decode_image([r[i] for r in ref], ANCHORS, num_classes, hw)
        got = {}
        for c_, k_ in zip(gc.tolist(), gi.tolist()):
            got.setdefault(c_, []).append(k_)
        for c in range(num_classes):
            picks, margin = pp.nms_decision_margin(boxes_r, scores_r[:, c], 20, iou, thr)
            min_margin_all = min(min_margin_all, margin)
            total += len(picks)
            mine = got.get(c, [])
            agree += len(set(picks.tolist()) & set(mine))
            if picks.tolist() != mine:
                flips.append((i, c, margin))
    print('end-to-end detections: %d/%d picks in common; %d of %d (image, class) problems differ; smallest decision '
          'margin overall %.2e, on the differing problems %s'
          % (agree, total, len(flips), num_classes * len(res), min_margin_all, ', '.join('%.1e' % m for _, _, m in flips) or '-'))
    assert total > 0
    for i, c, margin in flips:
        assert margin <= DECISION_NOISE, ('image %d class %d: picks differ although every decision of the oracle had a '
                                          'margin >= %.2e' % (i, c, margin))


def test_logits_survey_recipe_vs_fp64(dev):
    """SURVEY.md 8(d)'s weight recipe amplifies rounding noise ~1e3x (oracle/params.py): NumPy fp32
    and NumPy fp64 already differ by ~1e-3 on it, so a 1e-4 bar against ANY fp32 implementation is
    not meaningful there.  The bar instead: measured against the fp64 oracle, the HIP path must be
    no less accurate than the NumPy fp32 oracle (x1.5 slack)."""
    b, hw = 1, (416, 416)
    m, P = _build('mobilenetv2x75', hw, 20, recipe='survey')
    x = params.synthetic_images(b, *hw)
    ref32 = om.yolov3_body(P, x, 'mobilenetv2x75', 3, 20)
    ref64 = om.yolov3_body(P, x.astype(np.float64), 'mobilenetv2x75', 3, 20)
    m.set_weights(P.values)
    ys = m(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    for i, (y, r32, r64) in enumerate(zip(ys, ref32, ref64)):
        den = np.maximum(1.0, np.abs(r64))
        e_gpu = np.abs(y.cpu().numpy().astype(np.float64) - r64) / den
        e_np = np.abs(r32.astype(np.float64) - r64) / den
        print('y%d  max/mean scaled error vs fp64:  HIP %.2e / %.2e   NumPy-fp32 %.2e / %.2e'
              % (i + 1, e_gpu.max(), e_gpu.mean(), e_np.max(), e_np.mean()))
        assert e_gpu.mean() <= 1.5 * e_np.mean() + 1e-6
        assert e_gpu.max() <= 1.5 * e_np.max() + 1e-5


def test_batch_equals_per_image(dev):
    """Batched execution == the reference applied to each image independently."""
    m, P = _build('mobilenetv2x75', (96, 96), 20)
    x = params.synthetic_images(4, 96, 96)
    om.yolov3_body(P, x[:1], 'mobilenetv2x75', 3, 20)
    m.set_weights(P.values)
    xd = torch.from_numpy(x).to(dev)
    full = [y.clone() for y in m(xd)]
    for i in range(4):
        one = m(xd[i:i + 1].contiguous())
        for a, bfull in zip(one, full):
            assert torch.equal(a[0], bfull[i])


def test_model_errors(dev):
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[64, 64, 3]), 'mobilenetv2x75', 3, num_classes=20)
    with pytest.raises(RuntimeError, match='weights'):
        m(torch.zeros((1, 64, 64, 3), device=dev))
    with pytest.raises(ValueError, match='missing parameters'):
        m.set_weights({})
    with pytest.raises(ValueError):
        m(torch.zeros((1, 32, 64, 3), device=dev))
    with pytest.raises(ValueError):
        m(torch.zeros((1, 64, 64, 3)))  # CPU tensor: no fallback


def test_unfused_plan_matches_fused(dev):
    """The fused inverted-residual kernels (lane-per-pixel front, register-chained matrix-pipe blocks, expand + depthwise) and
    the unfused op chain agree (and both match the oracle)."""
    from yoloret_amd import layers as L
    from yoloret_amd import runtime as rt
    from yoloret_amd.yolo3.model import yolov3_body
    import os
    hw = (96, 96)
    P = params.ParamStore(1234)
    x = params.synthetic_images(2, *hw)
    ref = om.yolov3_body(P, x, 'mobilenetv2x75', 3, 20)
    outs = {}
    from yoloret_amd import compiler
    for fuse in ('1', '1e', '0'):     # ('1e': the deep blocks as expand + depthwise | projection - what they ran before the weight-streaming form)
        os.environ['YOLORET_FUSE'] = fuse[0]
        saved = compiler.FUSE_MAX_CIN, compiler.FUSE_MIN_PIXELS, compiler.FUSE_MBK
        compiler.FUSE_MAX_CIN, compiler.FUSE_MIN_PIXELS = 1 << 20, 0  # fuse every eligible block, also the deep ones
        compiler.FUSE_MBK = fuse == '1'
        try:
            m = yolov3_body(L.Input(shape=[hw[0], hw[1], 3]), 'mobilenetv2x75', 3, num_classes=20)
        finally:
            os.environ.pop('YOLORET_FUSE', None)
            compiler.FUSE_MAX_CIN, compiler.FUSE_MIN_PIXELS, compiler.FUSE_MBK = saved
        kinds = set(o.kind for o in m.plan.ops)
        fused_kinds = {rt.OP_STEMBLOCK, rt.OP_MBLANE, rt.OP_MBR, rt.OP_MBE}
        # (at 96 x 96 the lane-per-pixel kernel's minimum map size keeps block_1..3 unfused: the matrix-pipe forms carry the test)
        if fuse == '1':     # block_7..15 in the weight-streaming form (k bit 6): no expand + depthwise op is left
            assert {rt.OP_STEMBLOCK, rt.OP_MBR} <= kinds and rt.OP_MBE not in kinds and sum(1 for o in m.plan.ops if o.kind == rt.OP_MBR and o.k & 0x40) == 9
        elif fuse == '1e':
            assert {rt.OP_STEMBLOCK, rt.OP_MBR, rt.OP_MBE} <= kinds and not any(o.kind == rt.OP_MBR and o.k & 0x40 for o in m.plan.ops)
        else:
            assert not (fused_kinds & kinds)
        m.set_weights(P.values)
        outs[fuse] = [y.cpu().numpy() for y in m(torch.from_numpy(x).to(dev))]
        for y, r in zip(outs[fuse], ref):
            assert_close(y, r, 1e-4, 'fuse=%s' % fuse)


def test_tuning_table_round_trip(dev, tmp_path, monkeypatch):
    """yr_autotune's table can be saved (yr_get_tuning) and installed in a fresh handle (yr_set_tuning, via
    YOLORET_TUNE_CACHE): same logits bit for bit, no trial launches the second time."""
    import json
    from yoloret_amd import layers as L
    from yoloret_amd import runtime as rt
    from yoloret_amd.yolo3.model import yolov3_body
    cache = tmp_path / 'tuned.json'
    monkeypatch.setenv('YOLORET_TUNE_CACHE', str(cache))
    monkeypatch.setenv('YOLORET_AUTOTUNE', '1')
    P = params.ParamStore(7, 'conditioned')
    x = torch.from_numpy(params.synthetic_images(2, 96, 96)).to(dev)
    outs = []
    for _ in range(2):
        m = yolov3_body(L.Input(shape=[96, 96, 3]), 'mobilenetv2x75', 3, num_classes=20)
        om.yolov3_body(P, params.synthetic_images(1, 96, 96), 'mobilenetv2x75', 3, 20)
        m.set_weights(P.values)
        outs.append([y.cpu().numpy() for y in m(x)])
    table = json.load(open(cache))
    (key, cfg), = table.items()
    assert key.endswith(':2') and len(cfg) == len(m.plan.ops)
    # tuned entries: a pointwise tile shape, or the walk geometry of a register-chained block (waves << 8 | row segments << 16)
    assert all((c == 0) or (o.kind in (rt.OP_POINTWISE, rt.OP_MBR, rt.OP_MBE)) for c, o in zip(cfg, m.plan.ops)) and any(cfg)
    assert all((c & 0xff) == 0 for c, o in zip(cfg, m.plan.ops) if o.kind in (rt.OP_MBR, rt.OP_MBE))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    bad = (ctypes.c_int32 * 3)(1, 2, 3)
    idx, hd = m._handle(x.device)
    with pytest.raises(rt.YoloretHipError, match='yr_set_tuning'):
        rt.check(rt.lib().yr_set_tuning(hd, 2, bad, 3))


@pytest.mark.parametrize('name,size', [('mobilenetv2x75', 352), ('mobilenetv2x14', 224)])
def test_odd_grids_with_all_graph_rewrites(dev, name, size):
    """Input sizes whose grids are odd (352 -> 11/22/44, 224 -> 7/14/28): the hoisted (up2_add) and pooled-output
    convs, the lane-per-pixel blocks and the ragged tiles of every kernel against the oracle's torch-CPU graph."""
    from oracle import torch_ref
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
    assert sum(o.name.endswith('_lowres') for o in m.plan.ops) == 2
    assert sum((getattr(o, 'stride', 0) == 2) + ((getattr(o, 'reserved0', 0) >> 8) & 1) for o in m.plan.ops if o.kind == 2) == 3      # (... as a two-output conv's second output)
    P = params.ParamStore(77, 'conditioned')
    x = params.synthetic_images(2, size, size)
    ref = torch_ref.TorchReference(P, name, 3, 20)(x)
    m.set_weights(P.values)
    for y, r in zip(m(torch.from_numpy(x).to(dev)), ref):
        assert_close(y.cpu().numpy().reshape(r.shape), r, 1e-4, '%s@%d' % (name, size))


def test_small_batch_plan_of_float32_models(dev):
    """Round 5: a float32 model runs batches of up to 4 images on the 'nohead' variant (the throughput plan with the head blocks'
    conv and depthwise as two launches - YR_OP_HEAD is a long chain per workgroup at a few images); same 1e-4 bar, same object."""
    from yoloret_amd import runtime as rt
    hw = (128, 128)
    m, P = _build('mobilenetv2x75', hw, 20)
    x = params.synthetic_images(6, hw[0], hw[1])
    ref = om.yolov3_body(P, x, 'mobilenetv2x75', 3, 20)
    m.set_weights(P.values)
    m.small_batch = 4
    assert m.small_variant == 'nohead' and m.variant(4) == 'nohead' and m.variant(2) == 'nohead_k' and m.variant(5) == 'throughput'
    xt = torch.from_numpy(x).to(dev)
    small = [y.cpu().numpy() for y in m(xt[:3])]
    big = [y.cpu().numpy() for y in m(xt)]
    k_small, k_big = [o.kind for o in m.plan_for(3).ops], [o.kind for o in m.plan_for(6).ops]
    assert rt.OP_HEAD not in k_small and k_big.count(rt.OP_HEAD) == 6 and k_small.count(rt.OP_MBR) + k_small.count(rt.OP_MBE) == k_big.count(rt.OP_MBR) == 15
    for i, r in enumerate(ref):
        assert_close(small[i], r[:3], 1e-4, 'nohead plan, output %d' % i)
        assert_close(big[i], r, 1e-4, 'throughput plan, output %d' % i)
    # its small maps run the k-split form of the split pointwise kernel (se_reduced bit 17; a property of the plan, so a batch of the
    # variant still equals its images run one by one, bit for bit)
    # batches of one or two images ('nohead_k'): its small maps run the k-split form of the split pointwise kernel (se_reduced bit 17; a
    # property of the plan, so a batch of the variant still equals its images run one by one, bit for bit)
    # batches between the few-image plans and Model.mbk_batch ('mid'): the throughput plan without the weight-streaming block form
    m.mbk_batch = 6
    assert m.variant(5) == 'mid' and m.variant(6) == 'throughput'
    mid = [y.cpu().numpy() for y in m(xt[:5])]
    k6 = lambda b_: sum(1 for o in m.plan_for(b_).ops if o.kind == rt.OP_MBR and o.k & 0x40)
    assert k6(5) == 0 and k6(6) == 9 and [o.kind for o in m.plan_for(5).ops].count(rt.OP_HEAD) == 6 and all(k6(b_) == 0 for b_ in (1, 3))
    for i, r in enumerate(ref):
        assert_close(mid[i], r[:5], 1e-4, 'mid plan, output %d' % i)
    flagged = [o.name for o in m.plan_for(2).ops if o.kind == rt.OP_POINTWISE and o.se_reduced & 0x20000]
    assert flagged and not any(o.se_reduced & 0x20000 for b_ in (3, 6) for o in m.plan_for(b_).ops if o.kind == rt.OP_POINTWISE)
    ran = dict((r['name'], r['kernel']) for r in m.profile(xt[:2], iters=1))
    assert any(ran[n].startswith('pwk_kernel') for n in flagged), ran
    two = [y.cpu().numpy() for y in m(xt[:2])]
    one = [y.cpu().numpy() for y in m(xt[1:2])]
    for i, r in enumerate(ref):
        assert_close(two[i], r[:2], 1e-4, 'nohead_k plan, output %d' % i)
        assert np.array_equal(one[i][0], two[i][1])


@pytest.mark.parametrize('model_name', ['mobilenetv2x75', 'efficientnetb0'])
def test_small_batch_plan(dev, model_name):
    """Batches up to Model.small_batch run the plan without block fusion (its own handle, blob and tile table); it
    meets the same 1e-4 bar, and one model object serves both plans side by side."""
    from yoloret_amd import runtime as rt
    hw = (128, 128)
    m, P = _build(model_name, hw, 20)
    x = params.synthetic_images(6, hw[0], hw[1])
    ref = om.yolov3_body(P, x, model_name, 3, 20)
    m.set_weights(P.values)
    m.small_batch, m.small_variant = 4, 'latency'
    assert m.variant(2) == 'latency' and m.variant(6) == 'throughput'
    xt = torch.from_numpy(x).to(dev)
    small = [y.cpu().numpy() for y in m(xt[:2])]      # latency plan
    big = [y.cpu().numpy() for y in m(xt)]            # throughput plan, same object
    again = [y.cpu().numpy() for y in m(xt[2:5])]     # latency plan again (3 images)
    lat_kinds = set(o.kind for o in m.plan_for(2).ops)
    assert rt.OP_MBLANE not in lat_kinds and rt.OP_MBCONV not in lat_kinds
    for i, r in enumerate(ref):
        assert_close(small[i], r[:2], 1e-4, 'latency plan, output %d' % i)
        assert_close(big[i], r, 1e-4, 'throughput plan, output %d' % i)
        assert_close(again[i], r[2:5], 1e-4, 'latency plan (3 images), output %d' % i)
    assert len(m._handles) == 2
    rows = m.profile(xt[:1], iters=2)
    assert len(rows) == len(m.plan_for(1).ops) and all(r['ms'] > 0 for r in rows)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json's other configurations at the resolutions they are quoted on (fp32 here; the bf16 / fp16 forms of
# c3 and c5 are in tests/test_gpu_narrow.py).  Reference for the expected logits: the oracle's torch-CPU graph
# (oracle/torch_ref.py, itself checked against the NumPy restatement in tests/test_golden.py).
@pytest.mark.parametrize('name,size,b', [('mobilenetv2x14', 512, 2),        # c4's model (model.py:192-203)
                                         ('efficientnetb0', 416, 1),        # SE EfficientNet-B0 (efficientnet.py:611-710)
                                         ('efficientnetb0-lite', 416, 1),   # c3's model
                                         ('efficientnetb3', 640, 1),        # the reference's EfficientNet-B3 (model.py:205-217)
                                         ('efficientnetb3-lite', 640, 1)])  # c5's model
def test_full_resolution_configs(dev, name, size, b):
    from oracle import torch_ref
    m, P = _build(name, (size, size), 20)
    x = params.synthetic_images(b, size, size)
    ref = torch_ref.TorchReference(P, name, 3, 20)(x)
    m.set_weights(P.values)
    ys = m(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    worst = 0.0
    for i, (y, r) in enumerate(zip(ys, ref)):
        worst = max(worst, assert_close(y.cpu().numpy().reshape(r.shape), r, 1e-4, '%s@%d y%d' % (name, size, i + 1)))
    print('%s@%d: max scaled logit error vs the torch-CPU oracle %.2e' % (name, size, worst))


# ---------------------------------------------------------------------------------------------------------------
# The operating points the reference itself ships (code/README.md:80-93, code/main.py:79-81,171-176): MobileNetV2 x0.75 @320 on VOC
# (C = 20) and EfficientNet-B3 @416 / @224 on COCO (C = 80: head width 255, 80 NMS problems per image, 1600-row records), evaluated in
# MAP mode - score_threshold 0.0 (every box with a positive score is an NMS candidate) - and at the demo's 0.2.
@pytest.mark.parametrize('name,size,classes', [('mobilenetv2x75', 320, 20), ('efficientnetb3', 416, 80), ('efficientnetb3', 224, 80)])
def test_reference_operating_points(dev, name, size, classes):
    from oracle import torch_ref
    from yoloret_amd.yolo3.model import yolo_eval
    b, hw = 2, (size, size)
    m, P = _build(name, hw, classes)
    x = params.synthetic_images(b, size, size)
    ref = [np.asarray(r) for r in torch_ref.TorchReference(P, name, 3, classes)(x)]
    m.set_weights(P.values)
    ys = m(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    worst = 0.0
    for i, (y, r) in enumerate(zip(ys, ref)):
        g = size // (32 >> i)
        assert tuple(y.shape) == (b, g, g, 3, classes + 5)
        worst = max(worst, assert_close(y.cpu().numpy().reshape(r.shape), r, 1e-4, '%s@%d C=%d y%d' % (name, size, classes, i + 1)))
    print('%s@%d C=%d: max scaled logit error vs the torch-CPU oracle %.2e' % (name, size, classes, worst))
    ref5 = [r.reshape(b, r.shape[1], r.shape[2], 3, classes + 5) for r in ref]
    for thr in (0.0, 0.2):       # MAP mode (main.py:171-176) | the demo's threshold
        res = yolo_eval(ys, ANCHORS, 3, classes, hw, max_boxes=20, score_threshold=thr, iou_threshold=0.5)
        assert len(res) == b
        if thr == 0.0:           # every class has more than 20 boxes with a positive score: all classes x max_boxes rows
            assert all(len(r[1]) == classes * 20 for r in res)
        _check_detections_with_margins(ys, ref5, res, hw, thr=thr, num_classes=classes)


@pytest.mark.parametrize('name', ['efficientnetb1', 'efficientnetb2', 'efficientnetb4', 'efficientnetb5'])
def test_other_efficientnet_widths(dev, name):
    """The compound-scaling table's other widths (code/yolo3/efficientnet.py:231-244; the reference wires B3 only, model.py:205-217):
    whole-graph float32 logits within 1e-4 of the torch-CPU oracle at 128 x 128 - shapes no kernel's shape list was written for."""
    from oracle import torch_ref
    b, size, classes = 2, 128, 20
    m, P = _build(name, (size, size), classes)
    x = params.synthetic_images(b, size, size)
    ref = [np.asarray(r) for r in torch_ref.TorchReference(P, name, 3, classes)(x)]
    m.set_weights(P.values)
    ys = m(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    for i, (y, r) in enumerate(zip(ys, ref)):
        assert_close(y.cpu().numpy().reshape(r.shape), r, 1e-4, '%s@%d y%d' % (name, size, i + 1))


def test_coco_width_records_packed_and_gathered(dev):
    """C = 80 at a batch of 8: 1600 rows per image through yr_pack_detections (yolo_eval_packed) and the multi-GPU record path
    (DetectionGatherer, one rank): the unpacked records equal the per-image lists of yolo_eval and the C oracle's, in MAP mode."""
    from yoloret_amd.parallel import DetectionGatherer
    from yoloret_amd.yolo3.model import yolo_eval, yolo_eval_packed, unpack_detections
    b, hw, classes = 8, (160, 160), 80
    m, P = _build('mobilenetv2x75', hw, classes)
    x = params.synthetic_images(b, *hw)
    om.yolov3_body(P, x[:1], 'mobilenetv2x75', 3, classes)
    m.set_weights(P.values)
    ys = m(torch.from_numpy(x).to(dev))
    for thr in (0.0, 0.2):
        det, cnt = yolo_eval_packed(ys, ANCHORS, 3, classes, hw, 20, thr, 0.5)
        assert tuple(det.shape) == (b, classes * 20, 6) and tuple(cnt.shape) == (b,)
        all_det, all_cnt = DetectionGatherer()(det, cnt)
        res = unpack_detections(all_det, all_cnt)
        lists = yolo_eval(ys, ANCHORS, 3, classes, hw, max_boxes=20, score_threshold=thr, iou_threshold=0.5)
        assert len(res) == len(lists) == b
        for i in range(b):
            gb, gs, gc = [t.cpu().numpy() for t in res[i]]
            lb, ls, lc = [t.cpu().numpy() for t in lists[i]]
            ob, os_, oc, _ = cpost.yolo_eval([y[i].cpu().numpy() for y in ys], ANCHORS, 3, classes, hw, 20, thr, 0.5)
            assert np.array_equal(gb, ob) and np.array_equal(gs, os_) and np.array_equal(gc, oc)
            assert np.array_equal(gb, lb) and np.array_equal(gs, ls) and np.array_equal(gc, lc)
            if thr == 0.0:
                assert int(all_cnt[i]) == classes * 20


def test_c2_batch64_properties(dev):
    """BASELINE config 2 at its full batch (64 images, the plan and tile table the bench runs).  The oracle cannot run
    64 images at 416 in seconds, so: (i) sampled images must equal their own batch-1 run through the SAME plan bit for
    bit (size-independent property: batching == the reference applied per image), (ii) two sampled images
    are checked against the NumPy oracle at the 1e-4 bar, (iii) the oracle's post-processing of the GPU's own logits
    must equal the GPU's detections exactly for the sampled images."""
    from yoloret_amd.yolo3.model import yolo_eval_packed, unpack_detections
    b, hw = 64, (416, 416)
    m, P = _build('mobilenetv2x75', hw, 20)
    m.small_batch = 0
    x = params.synthetic_images(b, *hw)
    sample = [0, 37, 63]
    ref = om.yolov3_body(P, x[sample[:2]], 'mobilenetv2x75', 3, 20)
    m.set_weights(P.values)
    xd = torch.from_numpy(x).to(dev)
    ys = [y.clone() for y in m(xd)]
    for j, i in enumerate(sample):
        one = m(xd[i:i + 1].contiguous())
        for a, full in zip(one, ys):
            assert torch.equal(a[0], full[i]), 'image %d differs between the batch-64 and the batch-1 run' % i
        if j < 2:
            for y, r in zip(ys, ref):
                assert_close(y[i].cpu().numpy(), r[j], 1e-4, 'B=64 image %d' % i)
    det, cnt = yolo_eval_packed(ys, ANCHORS, 3, 20, hw, 20, 0.2, 0.5)
    res = unpack_detections(det, cnt)
    assert len(res) == b
    for i in sample:
        gb, gs, gc = [t.cpu().numpy() for t in res[i]]
        ob, os_, oc, _ = cpost.yolo_eval([y[i].cpu().numpy() for y in ys], ANCHORS, 3, 20, hw, 20, 0.2, 0.5)
        assert np.array_equal(gb, ob) and np.array_equal(gs, os_) and np.array_equal(gc, oc)


def test_plan_blob_round_trip(dev, tmp_path):
    """Model.save_plan() -> yr_create_from_blob in a fresh process that never imports the graph compiler: same
    logits bit for bit, tile tables included (the recipe a non-Python host follows)."""
    import subprocess
    import sys
    from yoloret_amd import runtime as rt
    m, P = _build('mobilenetv2x75', (96, 96), 20)
    x = params.synthetic_images(3, 96, 96)
    om.yolov3_body(P, x[:1], 'mobilenetv2x75', 3, 20)
    m.set_weights(P.values)
    want = [y.cpu().numpy() for y in m(torch.from_numpy(x).to(dev))]      # autotunes batch 3
    blob_path, x_path, out_path = tmp_path / 'model.yrplan', tmp_path / 'x.npy', tmp_path / 'out.npz'
    data = m.save_plan(str(blob_path))
    np.save(x_path, x)
    import struct
    n_tables = struct.unpack_from('<I', data, 8 + 5 * 4)[0]
    assert data[:8] == rt.PLAN_MAGIC and n_tables == 1
    code = ('import sys, numpy as np, torch\n'
            'from yoloret_amd import runtime as rt\n'
            'h = rt.PlanHandle(open(sys.argv[1], "rb").read())\n'
            'assert h.input_hw == (96, 96) and [c for _, _, c in h.output_hwc] == [75, 75, 75]\n'
            'ys = h(torch.from_numpy(np.load(sys.argv[2])).cuda())\n'
            'torch.cuda.synchronize()\n'
            'assert "yoloret_amd.compiler" not in sys.modules and "yoloret_amd.engine" not in sys.modules\n'
            'np.savez(sys.argv[3], *[y.cpu().numpy() for y in ys])\n')
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, '-c', code, str(blob_path), str(x_path), str(out_path)], cwd=root)
    z = np.load(out_path)
    for i, w in enumerate(want):     # the C-ABI's logits are [B,G,G,A*(C+5)]; Model.__call__ views them as [B,G,G,A,C+5]
        assert np.array_equal(z['arr_%d' % i].reshape(w.shape), w)
    # a 16-bit plan travels the same way
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    from yoloret_amd.weights import synthetic_weights
    xd = torch.from_numpy(x).to(dev)
    # (-lite: the throughput plan's matrix-pipe network entry and first stride-2 block - parameter layouts the loader's
    # extent checks know since round 3)
    for name16 in ('efficientnetb0', 'efficientnetb0-lite'):
        L.set_global_policy('mixed_bfloat16')
        try:
            m16 = yolov3_body(L.Input(shape=[96, 96, 3]), name16, 3, num_classes=20)
        finally:
            L.set_global_policy('float32')
        m16.small_batch = 0
        m16.set_weights(synthetic_weights(m16, 3, 'conditioned'))
        want16 = [y.cpu().numpy() for y in m16(xd)]
        if name16.endswith('-lite'):
            assert m16.plan.ops[0].kind == rt.OP_STEMBLOCK and 'scale' in m16.plan.ops[0].params and m16.plan.ops[1].kind == rt.OP_MBH
        h = rt.PlanHandle(m16.save_plan())
        for a, w in zip(h(xd), want16):
            assert np.array_equal(a.cpu().numpy().reshape(w.shape), w)


@pytest.mark.parametrize('name,policy', [('mobilenetv2x75', 'float32'), ('efficientnetb0-lite', 'mixed_bfloat16'),
                                         ('efficientnetb0', 'float32'), ('efficientnetb3', 'mixed_float16')])
def test_uint8_network_entry(dev, name, policy):
    """Input(dtype='uint8'): the network-entry kernel (the fused stem block of the MobileNetV2 / -lite models, the stem and
    stem + depthwise kernels of the squeeze-excite EfficientNets) reads the decoded image BYTES and applies the x / 255 of
    tf.io.decode_image(dtype=float32) (reference code/yolo.py:106) itself.  Against the float32-input model fed u8 / 255:
    the same logits up to the rounding of (sum w u) / 255 versus sum w (u / 255) (float32 plans: 2e-5 scaled; 16-bit plans
    round activations, so a last-bit difference at the entry may flip a later rounding: within the plan's own noise);
    float32 plans also against the oracle at the 1e-4 bar.  Odd sizes exercise the border paths."""
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    hw = (96, 160)
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (3, hw[0], hw[1], 3), dtype=np.uint8)
    xf = (u8.astype(np.float32) / np.float32(255.0))
    L.set_global_policy(policy)
    try:
        mf = yolov3_body(L.Input(shape=[hw[0], hw[1], 3]), name, 3, num_classes=20)
        m8 = yolov3_body(L.Input(shape=[hw[0], hw[1], 3], dtype='uint8'), name, 3, num_classes=20)
    finally:
        L.set_global_policy('float32')
    P = params.ParamStore(1234, 'conditioned')
    ref = om.yolov3_body(P, xf, name, 3, 20)
    for m in (mf, m8):
        m.set_weights(P.values)
        m.small_batch = 0
    yf = [y.cpu().numpy() for y in mf(torch.from_numpy(xf).to(dev))]
    y8 = [y.cpu().numpy() for y in m8(torch.from_numpy(u8).to(dev))]
    assert m8.plan.input_buf.dtype == 3 and m8.plan.ops[0].srcs[0].buf.dtype == 3
    with pytest.raises(ValueError, match='uint8'):
        m8(torch.from_numpy(xf).to(dev))
    for i, (a, b, r) in enumerate(zip(y8, yf, ref)):
        if policy == 'float32':
            assert_close(a, b, 2e-5, '%s y%d uint8 entry vs float32 entry' % (name, i + 1))   # (measured 0.4e-5 .. 1.0e-5)
            assert_close(a, r, 1e-4, '%s y%d uint8 entry vs oracle' % (name, i + 1))
        else:
            e8 = np.abs(a - r) / np.maximum(1.0, np.abs(r))
            ef = np.abs(b - r) / np.maximum(1.0, np.abs(r))
            assert e8.mean() <= 1.25 * ef.mean() + 1e-6 and e8.max() <= 2.0 * ef.max() + 1e-5, (name, i, e8.max(), ef.max())
    # the latency plan (no block fusion: plain stem kernel) takes the bytes too
    m8.small_batch = 4
    y8s = [y.cpu().numpy() for y in m8(torch.from_numpy(u8[:2]).to(dev))]
    for a, b in zip(y8s, yf):
        if policy == 'float32':
            assert_close(a, b[:2], 2e-5, '%s latency plan, uint8 entry' % name)
        else:
            assert np.isfinite(a).all()


@pytest.mark.parametrize('name,size', [('mobilenetv2x75', 416), ('efficientnetb3', 640)])
def test_latency_plan_at_full_resolution_batch_1(dev, name, size):
    """Batch 1 at the BASELINE resolution is the reference's ONLY operating point (code/yolo.py:83-84: Input(batch_size=1))
    and what `p50_ms_b1` times: the latency plan (Model.small_batch = 4: no block fusion, its own blob and tile table) must
    meet the 1e-4 logit bar there, and the oracle's post-processing of the GPU's logits must equal the GPU's detections
    bit for bit.  (The other tests pin YOLORET_SMALL_BATCH=0 so that their small batches exercise the fused kernels.)"""
    from oracle import torch_ref
    from yoloret_amd import runtime as rt
    from yoloret_amd.yolo3.model import yolo_eval
    m, P = _build(name, (size, size), 20)
    x = params.synthetic_images(1, size, size)
    ref = torch_ref.TorchReference(P, name, 3, 20)(x)   # (the torch-CPU graph: checked against the NumPy restatement in tests/test_golden.py)
    m.set_weights(P.values)
    m.small_batch, m.small_variant = 4, 'latency'
    assert m.variant(1) == 'latency'
    kinds = set(o.kind for o in m.plan_for(1).ops)
    assert not kinds & {rt.OP_MBLANE, rt.OP_MBCONV, rt.OP_MBR, rt.OP_MBH}
    ys = m(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    for i, (y, r) in enumerate(zip(ys, ref)):
        assert_close(y.cpu().numpy().reshape(r.shape), r, 1e-4, '%s@%d latency plan y%d' % (name, size, i + 1))
    res = yolo_eval(ys, ANCHORS, 3, 20, (size, size), max_boxes=20, score_threshold=0.2, iou_threshold=0.5)   # (one image: one triple)
    _check_detections_with_margins(ys, [r.reshape(tuple(y.shape)) for y, r in zip(ys, ref)], [res], (size, size))
