#!/usr/bin/env python
"""p50 of one full step (forward + decode + NMS + pack) at batch 1, 2, 4, 8 for the plan the environment selects
(YOLORET_SMALL_BATCH=0: the throughput plan at every batch; default 4: the few-image plan up to batch 4).
    python tools/lat_sweep.py [model] [size] [dtype]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yoloret_amd import layers as L, weights as W
from yoloret_amd.pipeline import DetectionPipeline
from yoloret_amd.yolo3.model import yolov3_body
from yoloret_amd.yolo3.utils import get_anchors

name = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2x75'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
dt = sys.argv[3] if len(sys.argv) > 3 else 'f32'
dev = torch.device('cuda', 0)
L.set_global_policy({'f32': 'float32', 'bf16': 'mixed_bfloat16', 'f16': 'mixed_float16'}[dt])
m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
L.set_global_policy('float32')
m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
pipe = DetectionPipeline(m, get_anchors('model_data/yolo_anchors.txt'), 20, score_threshold=0.2, iou_threshold=0.5, max_boxes=20)
out = []
for b in [int(v) for v in os.environ.get("LAT_BATCHES", "1,2,4,8").split(",")]:
    x = torch.from_numpy(W.synthetic_images(b, size, size)).to(dev)
    shape = torch.tensor([[size, size]] * b, dtype=torch.int32, device=dev)
    for _ in range(30):
        pipe(x, shape)
    torch.cuda.synchronize()
    ts = []
    for _ in range(150):
        t0 = time.perf_counter()
        pipe(x, shape)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out.append('b%d %.3f ms' % (b, np.median(ts) * 1e3))
print('%s @%d %s small_batch=%s: %s' % (name, size, dt, os.environ.get('YOLORET_SMALL_BATCH', '4'), '  '.join(out)), flush=True)
