"""Decode / NMS / pack parity through the C-ABI.

Bars: decode is BIT-EXACT against the C oracle (both pin exp to the same float32 algorithm
and run without FMA contraction) and within 2e-6 relative of the NumPy/libm oracle; the NMS
index set and order are BIT-EXACT against both oracles on identical boxes/scores; packed
detections equal the oracle's yolo_eval output exactly."""
import numpy as np
import pytest
import torch

from oracle import cpost
from oracle import postprocess as pp
from tests.util import ANCHORS

pytestmark = pytest.mark.gpu


def _rt():
    from yoloret_amd import runtime as rt
    return rt


def _logits(rng, b, hw, c, scale=3.0):
    return [(rng.standard_normal((b, hw[0] // s, hw[1] // s, 3, c + 5)) * scale).astype(np.float32)
            for s in (32, 16, 8)]


@pytest.mark.parametrize('hw,c,image_shapes', [
    ((416, 416), 20, [(416, 416), (375, 500), (500, 375)]),
    ((320, 320), 20, [(240, 320)]),
    ((64, 96), 80, [(100, 333), (64, 96)]),
])
def test_decode_bit_exact_vs_c_oracle(dev, hw, c, image_shapes):
    rt = _rt()
    rng = np.random.default_rng(11)
    b = len(image_shapes)
    ys = _logits(rng, b, hw, c)
    ys[0][0, 0, 0, 0, :] = [30.0, -30.0, 9.0, -9.0, 100.0] + [0.0] * c  # saturation / clip paths
    yd = [torch.from_numpy(y).to(dev) for y in ys]
    ihw = rt.image_hw_tensor(np.array(image_shapes), b, dev)
    boxes, scores = rt.decode(yd, ANCHORS, c, ihw, hw)
    torch.cuda.synchronize()
    boxes, scores = boxes.cpu().numpy(), scores.cpu().numpy()
    for i in range(b):
        rb, rs = cpost.decode_image([y[i] for y in ys], ANCHORS, c, image_shapes[i])
        assert np.array_equal(boxes[i], rb), 'boxes differ from the C oracle (image %d)' % i
        assert np.array_equal(scores[i], rs), 'scores differ from the C oracle (image %d)' % i
        nb, ns = pp.decode_image([y[i] for y in ys], ANCHORS, c, image_shapes[i])
        assert np.allclose(boxes[i], nb, rtol=2e-6, atol=2e-4)  # boxes are in pixels (<= 500)
        assert np.allclose(scores[i].T, ns, rtol=2e-6, atol=1e-7)


def test_decode_zero_logits_kat(dev):
    """all-zero logits, input 416, image (375,500)."""
    rt = _rt()
    ys = [torch.zeros((1, g, g, 3, 25), device=dev) for g in (13, 26, 52)]
    ihw = rt.image_hw_tensor((375, 500), 1, dev)
    boxes, scores = rt.decode(ys, ANCHORS, 20, ihw, (416, 416))
    b = boxes[0].cpu().numpy()
    assert (scores.cpu().numpy() == 0.25).all()
    assert b[(6 * 13 + 6) * 3].astype(np.int32).tolist() == [133, 180, 241, 319]
    assert b[2].astype(np.int32).tolist() == [0, 0, 152, 243]
    assert b[(12 * 13 + 12) * 3 + 1].astype(np.int32).tolist() == [299, 387, 375, 500]
    idx, cnt = rt.nms(boxes, scores, 20, 0.2, 0.5)
    want = [0, 1, 2, 6, 10, 12, 16, 18, 20, 22, 24, 28, 30, 34, 36, 38, 43, 45, 51, 52]  # Appendix D.4
    assert (cnt.cpu().numpy() == 20).all()
    assert (idx.cpu().numpy()[0] == np.array(want)).all()


def test_yolo_head_and_correct_boxes_layouts(dev):
    rt = _rt()
    rng = np.random.default_rng(5)
    feats = (rng.standard_normal((2, 13, 13, 3, 25)) * 2).astype(np.float32)
    fd = torch.from_numpy(feats).to(dev)
    anchors = ANCHORS[[6, 7, 8]]
    xy, wh, conf, probs, sc = rt.yolo_head(fd, anchors, (416, 416), with_scores=True)
    ihw = rt.image_hw_tensor((375, 500), 2, dev)
    boxes = rt.correct_boxes(xy, wh, (416, 416), ihw)
    torch.cuda.synchronize()
    for i in range(2):
        rxy, rwh, rconf, rprobs = pp.yolo_head(feats[i], anchors, (416, 416))
        assert np.allclose(xy[i].cpu().numpy(), rxy, rtol=2e-6, atol=1e-7)
        assert np.allclose(wh[i].cpu().numpy(), rwh, rtol=2e-6, atol=1e-7)
        assert np.allclose(conf[i].cpu().numpy(), rconf, rtol=2e-6, atol=1e-7)
        assert np.allclose(probs[i].cpu().numpy(), rprobs, rtol=2e-6, atol=1e-7)
        assert np.allclose(sc[i].cpu().numpy(), rconf * rprobs, rtol=3e-6, atol=1e-7)
        rb = pp.yolo_correct_boxes(rxy, rwh, (416, 416), (375, 500))
        assert np.allclose(boxes[i].cpu().numpy(), rb, rtol=3e-6, atol=3e-4)


def _random_boxes(rng, n, size=416.0, degenerate=True):
    cy, cx = rng.uniform(0, size, n), rng.uniform(0, size, n)
    h, w = rng.uniform(2, size / 2, n), rng.uniform(2, size / 2, n)
    b = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], 1)
    b = np.clip(b, 0, size).astype(np.float32)
    if degenerate:
        b[::17, 2] = b[::17, 0]          # zero-height boxes (area 0 -> IoU 0)
        b[5::23] = b[5::23][:, [2, 3, 0, 1]]  # flipped corners (canonicalised by IOU())
        b[7::29] = b[6::29][:len(b[7::29])]   # exact duplicates
    return b


@pytest.mark.parametrize('n,c,score_thr,ties', [(10647, 20, 0.2, False), (10647, 4, 0.0, False),
                                                (3000, 7, 0.3, True), (100, 3, 0.99, False), (257, 2, 0.1, True),
                                                (30000, 2, 0.3, False), (25200, 3, 0.2, True)])
def test_nms_bit_exact(dev, n, c, score_thr, ties):
    rt = _rt()
    rng = np.random.default_rng(n + c)
    b = 2
    boxes = np.stack([_random_boxes(rng, n) for _ in range(b)])
    scores = rng.random((b, c, n), dtype=np.float32)
    if ties:
        scores = np.round(scores * 8) / 8  # many exactly equal scores -> index tie-break matters
    scores = scores.astype(np.float32)
    idx, cnt = rt.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 20, score_thr, 0.5)
    torch.cuda.synchronize()
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    for i in range(b):
        for k in range(c):
            ref = pp.non_max_suppression(boxes[i], scores[i, k], 20, 0.5, score_thr)
            refc = cpost.nms(boxes[i], scores[i, k], 20, 0.5, score_thr)
            assert np.array_equal(ref, refc)
            assert cnt[i, k] == len(ref)
            assert np.array_equal(idx[i, k, :len(ref)], ref), (i, k)
            assert (idx[i, k, len(ref):] == -1).all()


def test_nms_empty_and_single(dev):
    rt = _rt()
    boxes = torch.tensor([[[0., 0., 10., 10.], [1., 1., 9., 9.], [20., 20., 30., 30.]]], device=dev)
    scores = torch.tensor([[[0.1, 0.1, 0.1], [0.9, 0.8, 0.7]]], device=dev)
    idx, cnt = rt.nms(boxes, scores, 5, 0.5, 0.5)
    assert cnt.cpu().tolist() == [[0, 2]]
    assert idx.cpu().tolist() == [[[-1] * 5, [0, 2, -1, -1, -1]]]


def test_pack_matches_oracle_eval(dev):
    rt = _rt()
    rng = np.random.default_rng(21)
    b, c, hw = 3, 20, (416, 416)
    ys = _logits(rng, b, hw, c, scale=2.5)
    shapes = [(416, 416), (375, 500), (300, 300)]
    yd = [torch.from_numpy(y).to(dev) for y in ys]
    ihw = rt.image_hw_tensor(np.array(shapes), b, dev)
    boxes, scores = rt.decode(yd, ANCHORS, c, ihw, hw)
    idx, cnt = rt.nms(boxes, scores, 20, 0.2, 0.5)
    det, dcnt = rt.pack_detections(boxes, scores, idx, cnt)
    torch.cuda.synchronize()
    det, dcnt = det.cpu().numpy(), dcnt.cpu().numpy()
    for i in range(b):
        rb, rs, rc, _ = cpost.yolo_eval([y[i] for y in ys], ANCHORS, 3, c, shapes[i], 20, 0.2, 0.5)
        k = dcnt[i]
        assert k == len(rs)
        assert np.array_equal(det[i, :k, 0:4], rb)
        assert np.array_equal(det[i, :k, 4].view(np.float32), rs)
        assert np.array_equal(det[i, :k, 5], rc)
        assert (det[i, k:, 5] == -1).all() and (det[i, k:, :5] == 0).all()


@pytest.mark.parametrize('hw,c,image_shapes', [((416, 416), 20, [(375, 500), (416, 416)]), ((64, 96), 7, [(100, 333)])])
def test_decode_zoom_tta_bit_exact(dev, hw, c, image_shapes):
    """yr_decode_zoom (the zoom-in TTA branch, model.py:408-417): 2A boxes per cell, bit-exact vs the C oracle,
    and the full yolo_eval over the doubled box set equals the oracle's."""
    rt = _rt()
    from yoloret_amd.yolo3.model import yolo_eval, yolo_boxes_and_scores
    rng = np.random.default_rng(23)
    b = len(image_shapes)
    ys, zs = _logits(rng, b, hw, c), _logits(rng, b, hw, c)
    yd = [torch.from_numpy(y).to(dev) for y in ys]
    zd = [torch.from_numpy(z).to(dev) for z in zs]
    ihw = rt.image_hw_tensor(np.array(image_shapes), b, dev)
    boxes, scores = rt.decode(yd, ANCHORS, c, ihw, hw, zoom_ys=zd)
    torch.cuda.synchronize()
    boxes, scores = boxes.cpu().numpy(), scores.cpu().numpy()
    n = rt.num_boxes(hw[0], hw[1])
    assert boxes.shape == (b, 2 * n, 4) and scores.shape == (b, c, 2 * n)
    for i in range(b):
        rb, rs = cpost.decode_image([y[i] for y in ys], ANCHORS, c, image_shapes[i], zoom_outputs=[z[i] for z in zs])
        assert np.array_equal(boxes[i], rb) and np.array_equal(scores[i], rs)
    res = yolo_eval(yd, ANCHORS, 3, c, np.array(image_shapes), score_threshold=.3, zoom_outputs=zd)
    res = [res] if b == 1 else res
    for i in range(b):
        ob, os_, oc, _ = cpost.yolo_eval([y[i] for y in ys], ANCHORS, 3, c, image_shapes[i], 20, .3, .5,
                                         zoom_outputs=[z[i] for z in zs])
        assert np.array_equal(res[i][0].cpu().numpy(), ob) and np.array_equal(res[i][1].cpu().numpy(), os_)
        assert np.array_equal(res[i][2].cpu().numpy(), oc)
    # the per-scale reference-layout helper agrees with the fused kernel (scale 0 of image 0)
    bx, sc = yolo_boxes_and_scores(yd[0][:1], ANCHORS[[6, 7, 8]], c, hw, image_shapes[0], zoom_feats=zd[0][:1])
    n0 = (hw[0] // 32) * (hw[1] // 32) * 6
    assert np.array_equal(bx.cpu().numpy(), boxes[0, :n0]) and np.array_equal(sc.cpu().numpy(), scores[0, :, :n0].T)
    with pytest.raises(ValueError):
        rt.decode(yd, ANCHORS, c, ihw, hw, zoom_ys=[zd[0], zd[0], zd[2]])


@pytest.mark.parametrize('kind', ['uniform', 'dense_top', 'duplicates', 'ties', 'one_bin'])
def test_nms_with_more_candidates_than_the_first_pass_list(dev, kind):
    """More candidates above the score threshold than the first launch's list holds (5200): it keeps the highest-scoring
    ones (histogram cut) and must still return exactly the oracle's picks - falling back to the full-capacity launch
    when the kept list runs dry before max_boxes picks (heavy suppression) or when scores pile up in one histogram bin."""
    rt = _rt()
    rng = np.random.default_rng({'uniform': 1, 'dense_top': 2, 'duplicates': 3, 'ties': 4, 'one_bin': 5}[kind])
    n, c, b = 10647, 3, 2
    boxes = np.stack([_random_boxes(rng, n) for _ in range(b)])
    scores = (0.2 + 0.8 * rng.random((b, c, n))).astype(np.float32)          # every box is a candidate
    if kind == 'dense_top':
        scores = (1.0 - 1e-3 * rng.random((b, c, n))).astype(np.float32)     # all within the top two bins
    elif kind == 'duplicates':                                                # a few distinct boxes: everything suppressed
        proto = _random_boxes(rng, 7, degenerate=False)
        boxes = np.stack([proto[rng.integers(0, 7, n)] for _ in range(b)])
    elif kind == 'ties':
        scores = (np.round(scores * 64) / 64).astype(np.float32)
    elif kind == 'one_bin':
        scores = np.full((b, c, n), 0.75, np.float32)                        # one value: index order decides
    idx, cnt = rt.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 20, 0.2, 0.5)
    torch.cuda.synchronize()
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    for i in range(b):
        for k in range(c):
            ref = cpost.nms(boxes[i], scores[i, k], 20, 0.5, 0.2)
            assert cnt[i, k] == len(ref), (kind, i, k, cnt[i, k], len(ref))
            assert np.array_equal(idx[i, k, :len(ref)], ref), (kind, i, k)
            assert (idx[i, k, len(ref):] == -1).all()
