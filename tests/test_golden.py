"""Committed golden vectors (tests/golden/, made by tests/golden/make_golden.py).

CPU: the oracle reproduces them (incl. the MobileNetV2 taps computed by the independent
`transformers` port - the only executable pin for the third-party part of the graph).
GPU: the HIP path reproduces the detector fixture through the drop-in surface."""
import os

import numpy as np
import pytest

from oracle import cpost
from oracle import model as om
from oracle import params
from oracle import postprocess as pp
from tests.util import ANCHORS, assert_close

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_oracle_backbone_matches_independent_port():
    z = np.load(os.path.join(G, 'backbone_mbv2_hf.npz'))
    for alpha, tag in ((0.75, 'x75'), (1.4, 'x14')):
        P = params.ParamStore(1234)
        x = params.synthetic_images(2, 64, 96, seed=7)
        acts = om.mobilenet_v2(P, x, alpha)
        for b in (2, 5, 12, 15):
            ref = z['%s_block_%d_add' % (tag, b)]
            assert acts['block_%d_add' % b].shape == ref.shape
            assert np.abs(acts['block_%d_add' % b] - ref).max() < 1e-4


def test_oracle_reproduces_detector_fixture():
    z = np.load(os.path.join(G, 'detector_tiny.npz'))
    P = params.ParamStore(1234)
    x = params.synthetic_images(2, 64, 64, seed=11)
    ys = om.yolov3_body(P, x, 'mobilenetv2x75', 3, 20)
    for y, k in zip(ys, ('y1', 'y2', 'y3')):
        assert_close(y, z[k], 1e-5, k)
    for i, shape in enumerate([(64, 64), (48, 100)]):
        b, s, c = pp.yolo_eval([z[k][i] for k in ('y1', 'y2', 'y3')], ANCHORS, 3, 20, shape, 20, 0.2, 0.5)
        assert np.array_equal(b, z['boxes%d' % i]) and np.array_equal(c, z['classes%d' % i])
        assert np.allclose(s, z['scores%d' % i], rtol=1e-6)
        cb, cs, cc, _ = cpost.yolo_eval([z[k][i] for k in ('y1', 'y2', 'y3')], ANCHORS, 3, 20, shape, 20, 0.2, 0.5)
        assert np.array_equal(cc, c) and np.abs(cb - b).max() <= 1  # C oracle pins exp(): boxes may differ in the last integer


def test_nms_fixture():
    z = np.load(os.path.join(G, 'nms_cases.npz'))
    assert pp.non_max_suppression(z['boxes'], z['scores'], 20, 0.5, 0.2).tolist() == z['picks'].tolist()
    assert cpost.nms(z['boxes'], z['scores'], 20, 0.5, 0.2).tolist() == z['picks'].tolist()


@pytest.mark.gpu
def test_hip_path_reproduces_detector_fixture(dev):
    import torch
    from yoloret_amd import layers as L
    from yoloret_amd import runtime as rt
    from yoloret_amd.weights import synthetic_weights
    from yoloret_amd.yolo3.model import yolov3_body
    z = np.load(os.path.join(G, 'detector_tiny.npz'))
    m = yolov3_body(L.Input(shape=[64, 64, 3]), 'mobilenetv2x75', 3, num_classes=20)
    m.set_weights(synthetic_weights(m, 1234, 'conditioned'))  # the product's own generator == the oracle's recipe
    x = params.synthetic_images(2, 64, 64, seed=11)
    ys = m(torch.from_numpy(x).to(dev))
    for y, k in zip(ys, ('y1', 'y2', 'y3')):
        assert_close(y.cpu().numpy(), z[k], 1e-4, k)
    zb = torch.from_numpy(z['nms_boxes'] if 'nms_boxes' in z else np.load(os.path.join(G, 'nms_cases.npz'))['boxes'])[None].to(dev)
    zs = torch.from_numpy(np.load(os.path.join(G, 'nms_cases.npz'))['scores'])[None, None].to(dev)
    idx, cnt = rt.nms(zb.contiguous(), zs.contiguous(), 20, 0.2, 0.5)
    picks = np.load(os.path.join(G, 'nms_cases.npz'))['picks']
    assert idx[0, 0, :int(cnt[0, 0])].cpu().tolist() == picks.tolist()


def test_oracle_efficientnet_matches_independent_port():
    """EfficientNet-B0 / -B3 stage ends (the detection taps) of the oracle against the activations an unrelated
    implementation of the published architecture (`transformers` EfficientNetModel) produced from the same weights
    (tests/golden/make_golden.py: hf_efficientnet_taps)."""
    z = np.load(os.path.join(G, 'backbone_effnet_hf.npz'))
    for key, tag in (('efficientnet-b0', 'b0'), ('efficientnet-b3', 'b3')):
        width, depth = om.EFFNET_COEFFS[key]
        P = params.ParamStore(1234)
        x = params.synthetic_images(2, 64, 96, seed=7)
        acts = om.efficientnet(P, x, width, depth)
        for si in (2, 3, 5, 6):
            ref = z['%s_stage%d' % (tag, si)]
            assert acts['stage%d' % si].shape == ref.shape
            assert np.abs(acts['stage%d' % si] - ref).max() < 1e-5
