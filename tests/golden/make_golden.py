#!/usr/bin/env python
"""Generates tests/golden/*.npz.  Run in the BUILD container only:

    python tests/golden/make_golden.py

What the fixtures are (data only - inputs and expected outputs):
  backbone_mbv2_hf.npz   MobileNetV2 x0.75 / x1.4 tap activations computed by an INDEPENDENT PyTorch port
                         of the architecture (the `transformers` wheel's MobileNetV2Model with
                         tf_padding=True), loaded with the oracle's seeded weights.  This is the only
                         executable cross-check available for the third-party part of the graph
                         (tf.keras.applications.MobileNetV2 is not in /root/reference; TensorFlow is absent).
  backbone_effnet_hf.npz The same for the EfficientNet-B0 / -B3 backbone (SE + Swish MBConv, reference
                         code/yolo3/efficientnet.py:406-536,611-710): ends of stages 2/3/5/6 from `transformers`'
                         EfficientNetModel with the oracle's weights.  Pins the oracle's reading of efficientnet.py
                         (block order, SE width, padding, BN epsilon, repeats/width rounding) against a second,
                         unrelated implementation of the published architecture.
  detector_tiny.npz      End-to-end logits + detections of the NumPy oracle for a 64x64 MobileNetV2x0.75
                         detector (labelled: oracle = this repo's CPU restatement, NOT TensorFlow).
  nms_cases.npz          Boxes/scores with ties / degenerate boxes and the line-by-line TF-loop picks.
The reference ships no golden vectors, tests or checkpoints, so nothing here comes
from /root/reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import model as om  # noqa: E402
from oracle import params, postprocess as pp  # noqa: E402

ANCHORS = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326], np.float32).reshape(-1, 2)


def hf_mobilenetv2_taps(P, x, alpha):
    """Tap activations from the transformers port, fed with the oracle's parameters."""
    import torch
    from transformers import MobileNetV2Config, MobileNetV2Model
    cfg = MobileNetV2Config(depth_multiplier=alpha, tf_padding=True, finegrained_output=True, layer_norm_eps=1e-3)
    m = MobileNetV2Model(cfg, add_pooling_layer=False).eval()
    sd = m.state_dict()

    def conv(dst, name, dw=False):
        if dw:
            k = P.values[name + '/depthwise_kernel']            # [3,3,C]
            sd[dst + '.convolution.weight'] = torch.from_numpy(k).permute(2, 0, 1).unsqueeze(1).contiguous()
        else:
            k = P.values[name + '/kernel']                      # HWIO
            sd[dst + '.convolution.weight'] = torch.from_numpy(k).permute(3, 2, 0, 1).contiguous()

    def bn(dst, name):
        sd[dst + '.normalization.weight'] = torch.from_numpy(P.values[name + '/gamma'])
        sd[dst + '.normalization.bias'] = torch.from_numpy(P.values[name + '/beta'])
        sd[dst + '.normalization.running_mean'] = torch.from_numpy(P.values[name + '/moving_mean'])
        sd[dst + '.normalization.running_var'] = torch.from_numpy(P.values[name + '/moving_variance'])

    conv('conv_stem.first_conv', 'Conv1'); bn('conv_stem.first_conv', 'bn_Conv1')
    conv('conv_stem.conv_3x3', 'expanded_conv_depthwise', True); bn('conv_stem.conv_3x3', 'expanded_conv_depthwise_BN')
    conv('conv_stem.reduce_1x1', 'expanded_conv_project'); bn('conv_stem.reduce_1x1', 'expanded_conv_project_BN')
    for b in range(1, 16):
        pre, dst = 'block_%d_' % b, 'layer.%d' % (b - 1)
        conv(dst + '.expand_1x1', pre + 'expand'); bn(dst + '.expand_1x1', pre + 'expand_BN')
        conv(dst + '.conv_3x3', pre + 'depthwise', True); bn(dst + '.conv_3x3', pre + 'depthwise_BN')
        conv(dst + '.reduce_1x1', pre + 'project'); bn(dst + '.reduce_1x1', pre + 'project_BN')
    m.load_state_dict(sd)
    with torch.no_grad():
        hs = m(torch.from_numpy(x).permute(0, 3, 1, 2), output_hidden_states=True).hidden_states
    # hidden_states[i] = output of layer i (i = block i+1)
    return {b: hs[b - 1].permute(0, 2, 3, 1).numpy() for b in (2, 5, 12, 15)}


def hf_efficientnet_taps(P, x, width, depth):
    """Ends of stages 2/3/5/6 (the taps of reference code/yolo3/model.py:213-216) from the `transformers` EfficientNet
    port - an implementation of the published architecture that shares nothing with oracle/model.py or with the
    reference's efficientnet.py - fed with the oracle's parameters.  Input sizes must keep every stride-2 layer's
    input even: the port pads stride-2 convs Keras-applications style, which equals TF 'SAME' only then."""
    import math
    import torch
    from transformers import EfficientNetConfig, EfficientNetModel
    cfg = EfficientNetConfig(width_coefficient=width, depth_coefficient=depth, hidden_dim=om.round_filters(1280, width),
                             batch_norm_eps=1e-3, hidden_act='swish')
    m = EfficientNetModel(cfg).eval()
    sd = m.state_dict()

    def conv(dst, name, dw=False):
        if dw:
            k = P.values[name + '/depthwise_kernel']            # [k,k,C]
            sd[dst + '.weight'] = torch.from_numpy(k).permute(2, 0, 1).unsqueeze(1).contiguous()
        else:
            k = P.values[name + '/kernel']                      # HWIO
            sd[dst + '.weight'] = torch.from_numpy(k).permute(3, 2, 0, 1).contiguous()
        if name + '/bias' in P.values and not dw:
            sd[dst + '.bias'] = torch.from_numpy(P.values[name + '/bias'])

    def bn(dst, name):
        sd[dst + '.weight'] = torch.from_numpy(P.values[name + '/gamma'])
        sd[dst + '.bias'] = torch.from_numpy(P.values[name + '/beta'])
        sd[dst + '.running_mean'] = torch.from_numpy(P.values[name + '/moving_mean'])
        sd[dst + '.running_var'] = torch.from_numpy(P.values[name + '/moving_variance'])

    conv('embeddings.convolution', 'stem_conv'); bn('embeddings.batchnorm', 'stem_BN')
    blk, ends = 0, {}
    for si, (r, k, s_, e, i, o, se) in enumerate(om.EFFNET_STAGES[:6], start=1):
        for rep in range(int(math.ceil(depth * r))):
            name, dst = 'stage%d_block%d' % (si, rep), 'encoder.blocks.%d' % blk
            if e != 1:
                conv(dst + '.expansion.expand_conv', name + '_expand'); bn(dst + '.expansion.expand_bn', name + '_expand_BN')
            conv(dst + '.depthwise_conv.depthwise_conv', name + '_dw', True); bn(dst + '.depthwise_conv.depthwise_norm', name + '_dw_BN')
            conv(dst + '.squeeze_excite.reduce', name + '_se_reduce'); conv(dst + '.squeeze_excite.expand', name + '_se_expand')
            conv(dst + '.projection.project_conv', name + '_project'); bn(dst + '.projection.project_bn', name + '_project_BN')
            blk += 1
        ends[si] = blk     # hidden_states[blk] = output of the stage's last block
    missing = [k for k in m.state_dict() if k not in sd]
    assert not missing
    m.load_state_dict(sd)
    with torch.no_grad():
        hs = m(torch.from_numpy(x).permute(0, 3, 1, 2), output_hidden_states=True).hidden_states
    return {si: hs[ends[si]].permute(0, 2, 3, 1).numpy() for si in (2, 3, 5, 6)}


def main():
    out = {}
    for key, tag in (('efficientnet-b0', 'b0'), ('efficientnet-b3', 'b3')):
        width, depth = om.EFFNET_COEFFS[key]
        P = params.ParamStore(1234)
        x = params.synthetic_images(2, 64, 96, seed=7)
        acts = om.efficientnet(P, x, width, depth)      # creates the seeded parameters
        hf = hf_efficientnet_taps(P, x, width, depth)
        for si in (2, 3, 5, 6):
            d = float(np.abs(hf[si] - acts['stage%d' % si]).max())
            print('EfficientNet-%s stage%d: HF port vs oracle max |diff| = %.2e  shape %s' % (tag.upper(), si, d, hf[si].shape))
            assert d < 1e-4
            out['%s_stage%d' % (tag, si)] = hf[si].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'backbone_effnet_hf.npz'), **out)

    out = {}
    for alpha, tag in ((0.75, 'x75'), (1.4, 'x14')):
        P = params.ParamStore(1234)
        x = params.synthetic_images(2, 64, 96, seed=7)
        acts = om.mobilenet_v2(P, x, alpha)           # creates the seeded parameters
        hf = hf_mobilenetv2_taps(P, x, alpha)
        for b in (2, 5, 12, 15):
            d = float(np.abs(hf[b] - acts['block_%d_add' % b]).max())
            print('MobileNetV2 %s block_%d_add: HF port vs oracle max |diff| = %.2e  shape %s' % (tag, b, d, hf[b].shape))
            assert d < 1e-4
            out['%s_block_%d_add' % (tag, b)] = hf[b].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'backbone_mbv2_hf.npz'), **out)

    P = params.ParamStore(1234)
    x = params.synthetic_images(2, 64, 64, seed=11)
    ys = om.yolov3_body(P, x, 'mobilenetv2x75', 3, 20)
    det = pp.yolo_eval_batch(ys, ANCHORS, 3, 20, [(64, 64), (48, 100)], max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
    d = {'y1': ys[0], 'y2': ys[1], 'y3': ys[2]}
    for i, (b, s, c) in enumerate(det):
        d['boxes%d' % i], d['scores%d' % i], d['classes%d' % i] = b, s, c
    np.savez_compressed(os.path.join(HERE, 'detector_tiny.npz'), **d)

    rng = np.random.default_rng(5)
    n = 300
    cy, cx = rng.uniform(0, 100, n), rng.uniform(0, 100, n)
    h, w = rng.uniform(1, 50, n), rng.uniform(1, 50, n)
    boxes = np.clip(np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], 1), 0, 100).astype(np.float32)
    boxes[::13, 2] = boxes[::13, 0]
    boxes[5::17] = boxes[5::17][:, [2, 3, 0, 1]]
    boxes[9::19] = boxes[8::19][:len(boxes[9::19])]
    scores = (np.round(rng.random(n) * 16) / 16).astype(np.float32)
    picks = pp.nms_bruteforce(boxes, scores, 20, 0.5, 0.2)
    np.savez_compressed(os.path.join(HERE, 'nms_cases.npz'), boxes=boxes, scores=scores, picks=picks)
    print('wrote fixtures to', HERE)


if __name__ == '__main__':
    main()
