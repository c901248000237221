"""Pins the oracle (CPU, no GPU): hand-derived known-answer tests of SURVEY.md Appendix D,
agreement of the three restatements of decode/NMS (NumPy, C, line-by-line brute force), and the
independent torch-CPU implementation of the conv stack.  The reference itself ships no tests,
fixtures or checkpoints, so these - not reference outputs - are what pins the oracle
("parity unpinned", see oracle/__init__.py)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import cpost
from oracle import model as om
from oracle import nn, params
from oracle import postprocess as pp
from tests.util import ANCHORS


# ---------------------------------------------------------------- Appendix D.1 / D.2 / D.6
def test_channel_rounding_tables():
    fs = (16, 24, 32, 64, 96, 160, 320)
    assert [om._make_divisible(int(f * 0.75), 8) for f in fs] == [16, 24, 24, 48, 72, 120, 240]
    assert [om._make_divisible(int(f * 1.4), 8) for f in fs] == [24, 32, 48, 88, 136, 224, 448]
    assert om._make_divisible(32 * 0.75, 8) == 24 and om._make_divisible(32 * 1.4, 8) == 48
    assert [om.round_filters(f, 1.2) for f in (32, 16, 24, 40, 80, 112, 192, 320, 1280)] == \
        [40, 24, 32, 48, 96, 136, 232, 384, 1536]
    assert [om.round_repeats(r, 1.4) for r in (1, 2, 2, 3, 3, 4, 1)] == [2, 3, 3, 5, 5, 6, 2]


@pytest.mark.parametrize('size,n', [(416, 10647), (512, 16128), (640, 25200), (320, 6300)])
def test_box_counts(size, n):
    assert sum(3 * (size // s) ** 2 for s in (32, 16, 8)) == n


def test_structure_mobilenetv2x75():
    P = params.ParamStore(1)
    x = params.synthetic_images(1, 64, 64)
    acts = om.mobilenet_v2(P, x, 0.75)
    assert acts['block_15_add'].shape == (1, 2, 2, 120) and acts['block_12_add'].shape == (1, 4, 4, 72)
    assert acts['block_5_add'].shape == (1, 8, 8, 24) and acts['block_2_add'].shape == (1, 16, 16, 24)
    ys = om.yolov3_body(P, x, 'mobilenetv2x75')
    assert [y.shape for y in ys] == [(1, 2, 2, 3, 25), (1, 4, 4, 3, 25), (1, 8, 8, 3, 25)]


# ---------------------------------------------------------------- Appendix D.3 - D.5 (decode / NMS)
def test_decode_zero_logits_kat():
    ys = [np.zeros((g, g, 3, 25), np.float32) for g in (13, 26, 52)]
    for dec in (lambda: pp.decode_image(ys, ANCHORS, 20, (375, 500)),
                lambda: tuple(a if i == 0 else a.T for i, a in enumerate(cpost.decode_image(ys, ANCHORS, 20, (375, 500))))):
        b, s = dec()
        assert (s == 0.25).all()
        assert np.allclose(b[(6 * 13 + 6) * 3], [133.41348, 180.28848, 241.58655, 319.71155], atol=2e-4)
        assert b[(6 * 13 + 6) * 3].astype(np.int32).tolist() == [133, 180, 241, 319]
        assert b[2].astype(np.int32).tolist() == [0, 0, 152, 243]
        assert b[(12 * 13 + 12) * 3 + 1].astype(np.int32).tolist() == [299, 387, 375, 500]


def test_nms_tie_break_kat():
    ys = [np.zeros((g, g, 3, 25), np.float32) for g in (13, 26, 52)]
    b, s = pp.decode_image(ys, ANCHORS, 20, (375, 500))
    want = [0, 1, 2, 6, 10, 12, 16, 18, 20, 22, 24, 28, 30, 34, 36, 38, 43, 45, 51, 52]
    assert pp.non_max_suppression(b, s[:, 3], 20, 0.5, 0.2).tolist() == want
    assert cpost.nms(b, s[:, 3], 20, 0.5, 0.2).tolist() == want
    bb, ss, cc = pp.yolo_eval(ys, ANCHORS, 3, 20, (375, 500), score_threshold=0.2)
    assert len(ss) == 400 and cc.tolist() == sorted(cc.tolist()) and bb.dtype == np.int32


def test_letterbox_terms():
    # model.py:381-385 for input 416, image (375,500): ratio (.75,1), boxed (312,416), offset (52,0), scale 1.2019231
    xy = np.array([[[[0.5, 0.5]]]], np.float32)
    wh = np.array([[[[0.25, 0.25]]]], np.float32)
    b = pp.yolo_correct_boxes(xy, wh, (416, 416), (375, 500))[0, 0, 0]
    cy, cx = (0.5 * 416 - 52) * np.float32(375 / 312), 0.5 * 416 * np.float32(500 / 416)
    hh, ww = 0.25 * 416 * np.float32(375 / 312), 0.25 * 416 * np.float32(500 / 416)
    assert np.allclose(b, [cy - hh / 2, cx - ww / 2, cy + hh / 2, cx + ww / 2], rtol=1e-6)


def test_pinned_expf_accuracy():
    x = np.linspace(-30, 30, 20001).astype(np.float32)
    e = cpost.expf(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 2.5e-7  # ~2 ulp of float32


def test_c_and_numpy_decode_agree():
    rng = np.random.default_rng(0)
    ys = [(rng.standard_normal((g, g, 3, 25)) * 3).astype(np.float32) for g in (13, 26, 52)]
    b, s = pp.decode_image(ys, ANCHORS, 20, (375, 500))
    bc, sc = cpost.decode_image(ys, ANCHORS, 20, (375, 500))
    assert np.allclose(b, bc, rtol=2e-6, atol=2e-4) and np.allclose(s, sc.T, rtol=2e-6, atol=1e-7)


boxes_st = st.lists(st.tuples(st.integers(0, 20), st.integers(0, 20), st.integers(0, 20), st.integers(0, 20),
                              st.integers(0, 8)), min_size=1, max_size=40)


@settings(max_examples=200, deadline=None)
@given(boxes_st, st.sampled_from([0.0, 0.3, 0.5]), st.sampled_from([0.0, 0.25, 0.5]))
def test_nms_reformulation_equals_tf_loop(raw, iou_thr, score_thr):
    """arg-max/suppress rounds == TF's pop-and-test loop, incl. ties, degenerate and flipped boxes."""
    boxes = np.array([[r[0], r[1], r[2], r[3]] for r in raw], np.float32)
    scores = np.array([r[4] / 8.0 for r in raw], np.float32)
    want = pp.nms_bruteforce(boxes, scores, 5, iou_thr, score_thr)
    assert pp.non_max_suppression(boxes, scores, 5, iou_thr, score_thr).tolist() == want.tolist()
    assert cpost.nms(boxes, scores, 5, iou_thr, score_thr).tolist() == want.tolist()


# ---------------------------------------------------------------- conv stack: NumPy vs torch-CPU
def test_numpy_ops_match_torch_cpu():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(2)
    for (h, w, c, k, s) in [(13, 13, 24, 3, 1), (26, 26, 16, 3, 2), (9, 7, 8, 5, 2), (14, 14, 12, 5, 1), (8, 8, 3, 3, 2)]:
        x = rng.standard_normal((2, h, w, c)).astype(np.float32)
        wk = rng.standard_normal((k, k, c)).astype(np.float32)
        got = nn.depthwise(x, wk, s, 'same')
        pt, pb, _ = nn.same_pad(h, k, s)
        pl, pr, _ = nn.same_pad(w, k, s)
        xt = F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (pl, pr, pt, pb))
        ref = F.conv2d(xt, torch.from_numpy(wk).permute(2, 0, 1).unsqueeze(1), stride=s, groups=c).permute(0, 2, 3, 1).numpy()
        assert np.allclose(got, ref, atol=1e-4)
        wf = rng.standard_normal((k, k, c, 5)).astype(np.float32)
        got = nn.conv2d(x, wf, s, 'same')
        ref = F.conv2d(xt, torch.from_numpy(wf).permute(3, 2, 0, 1), stride=s).permute(0, 2, 3, 1).numpy()
        assert np.allclose(got, ref, atol=1e-4)
    x = rng.standard_normal((2, 8, 12, 5)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    assert np.array_equal(nn.maxpool(x, 2), F.max_pool2d(xt, 2).permute(0, 2, 3, 1).numpy())
    assert np.array_equal(nn.maxpool(x, 4), F.max_pool2d(xt, 4).permute(0, 2, 3, 1).numpy())
    assert np.array_equal(nn.upsample2(x), F.interpolate(xt, scale_factor=2, mode='nearest').permute(0, 2, 3, 1).numpy())


@pytest.mark.parametrize('name,hw', [('mobilenetv2x75', (96, 96)), ('mobilenetv2x14', (64, 64)),
                                     ('efficientnetb0', (64, 64)), ('efficientnetb3', (64, 96))])
def test_numpy_graph_matches_torch_graph(name, hw):
    from oracle import torch_ref
    P = params.ParamStore(1234)
    x = params.synthetic_images(2, *hw)
    a = om.yolov3_body(P, x, name)
    b = torch_ref.TorchReference(P, name)(x)
    for u, v in zip(a, b):
        assert u.shape == v.shape and np.abs(u - v).max() < 5e-5


def test_zoom_tta_decode_known_answers():
    """The zoom-in TTA branch (model.py:408-417): zero logits put every plain box centre at its cell centre and every
    zoom box centre at centre*224/416 + 96/416, with 224/416 of the size; C and NumPy oracles agree."""
    from oracle import cpost, postprocess as pp
    from tests.util import ANCHORS
    ys = [np.zeros((g, g, 3, 7), np.float32) for g in (13, 26, 52)]
    b, s = cpost.decode_image(ys, ANCHORS, 2, (416, 416), zoom_outputs=ys)
    nb, ns = pp.decode_image(ys, ANCHORS, 2, (416, 416), zoom_outputs=ys)
    assert b.shape == (2 * 10647, 4) and np.allclose(b, nb, atol=1e-3) and np.allclose(s.T, ns, atol=1e-7)
    assert (s == 0.25).all()
    # scale 0 (stride 32, 13x13), cell (h=3, w=6), anchor (116, 90): plain box, then its zoom twin A=3 entries later
    n = (3 * 13 + 6) * 6
    cy = lambda box: (box[0] + box[2]) / 2
    cx = lambda box: (box[1] + box[3]) / 2
    assert cy(b[n]) == pytest.approx(3.5 / 13 * 416, rel=1e-6) and cx(b[n]) == pytest.approx(6.5 / 13 * 416, rel=1e-6)
    assert b[n][2] - b[n][0] == pytest.approx(90.0, rel=1e-6) and b[n][3] - b[n][1] == pytest.approx(116.0, rel=1e-6)
    assert cy(b[n + 3]) == pytest.approx(3.5 / 13 * 224 + 96, rel=1e-6)    # (y*224/416 + 96/416) * 416
    assert cx(b[n + 3]) == pytest.approx(6.5 / 13 * 224 + 96, rel=1e-6)    # = 208: the centre maps to the centre
    assert b[n + 3][2] - b[n + 3][0] == pytest.approx(90.0 * 224 / 416, rel=1e-5)
    # the plain pass is embedded unchanged
    b0, s0 = cpost.decode_image(ys, ANCHORS, 2, (416, 416))
    assert np.array_equal(b[:169 * 6].reshape(169, 2, 3, 4)[:, 0].reshape(-1, 4), b0[:169 * 3])
