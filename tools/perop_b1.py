#!/usr/bin/env python
"""Per-op times of the flagship forward at batch 1 (where the launches go in the p50 latency)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yoloret_amd import layers as L, weights as W
from yoloret_amd.yolo3.model import yolov3_body
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = yolov3_body(L.Input(shape=[416, 416, 3]), 'mobilenetv2x75', 3, num_classes=20)
m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
x = torch.from_numpy(W.synthetic_images(b, 416, 416)).cuda()
rows = m.profile(x, iters=20)
tot = sum(r['ms'] for r in rows)
for r in sorted(rows, key=lambda r: -r['ms'])[:40]:
    print('%-22s %-26s %.4f ms' % (r['name'], r['kernel'], r['ms']))
print('total %.4f ms over %d ops' % (tot, len(rows)))
