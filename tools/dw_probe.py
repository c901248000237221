#!/usr/bin/env python
"""Per-op depthwise timing (GPU) under the current YR_DW_FORCE setting: prints name, map, ms for each depthwise launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yoloret_amd import layers as L
from yoloret_amd.weights import synthetic_images, synthetic_weights
from yoloret_amd.yolo3.model import yolov3_body

m = yolov3_body(L.Input(shape=[416, 416, 3]), 'mobilenetv2x75', 3, num_classes=20)
m.set_weights(synthetic_weights(m, 1, 'conditioned'))
x = torch.from_numpy(synthetic_images(64, 416, 416)).cuda()
rows = [r for r in m.profile(x, iters=20) if r['kind'] == 'depthwise']
tot = 0.0
for r in rows:
    tot += r['ms']
    print('%-22s %8.4f ms  %6.1f MB' % (r['name'], r['ms'], r['hbm_bytes'] / 1e6))
print('total %.4f' % tot)
