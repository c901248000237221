#!/usr/bin/env python
"""Batch-1 latency of the full step (forward + decode + NMS + pack) under the current fusion knobs (env):
    YOLORET_FUSE_MAX_CIN=1000 YOLORET_FUSE_MIN_PIXELS=1 python tools/lat_probe.py [model] [size] [f32|bf16|f16]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yoloret_amd import layers as L, runtime as rt, weights as W
from yoloret_amd.pipeline import DetectionPipeline
from yoloret_amd.yolo3.model import yolov3_body
from yoloret_amd.yolo3.utils import get_anchors

dev = torch.device('cuda', 0)
name = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2x75'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
dt = sys.argv[3] if len(sys.argv) > 3 else 'f32'
L.set_global_policy({'f32': 'float32', 'bf16': 'mixed_bfloat16', 'f16': 'mixed_float16'}[dt])
m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
L.set_global_policy('float32')
m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
import collections
print(dict(collections.Counter(rt.OP_NAMES[o.kind] for o in m.plan.ops)), len(m.plan.ops), 'ops')
pipe = DetectionPipeline(m, get_anchors('model_data/yolo_anchors.txt'), 20, score_threshold=0.2, iou_threshold=0.5, max_boxes=20)
x = torch.from_numpy(W.synthetic_images(1, size, size)).to(dev)
shape = torch.tensor([[size, size]], dtype=torch.int32, device=dev)
for _ in range(20):
    pipe(x, shape)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    pipe(x, shape)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print('p50 %.4f ms  min %.4f ms' % (np.median(ts) * 1e3, min(ts) * 1e3))
