#!/usr/bin/env python
"""Experiment: one batch-64 step vs the same step as K independent sub-batches replayed on K streams
(images are independent units, so the sub-batches share nothing but the weights)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yoloret_amd import layers as L, weights as W
from yoloret_amd.pipeline import DetectionPipeline
from yoloret_amd.yolo3.model import yolov3_body
from yoloret_amd.yolo3.utils import get_anchors

dev = torch.device('cuda', 0)
anchors = get_anchors('model_data/yolo_anchors.txt')
B = 64
x = torch.from_numpy(W.synthetic_images(B, 416, 416, seed=20240416)).to(dev)
hw = torch.tensor([[416, 416]] * B, dtype=torch.int32, device=dev)


def make():
    m = yolov3_body(L.Input(shape=[416, 416, 3]), 'mobilenetv2x75', 3, num_classes=20)
    m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
    return m


def timeit(step, n=30):
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


base = DetectionPipeline(make(), anchors, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
print('baseline       %.4f ms' % timeit(lambda: base(x, hw)))
ref = [t.clone() for t in base(x, hw)]
for K in (2, 4):
    models = [make() for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    pipe = DetectionPipeline(models[0], anchors, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
    ys = pipe._buffers(B, dev)['ys']
    n = B // K

    def step():
        cur = torch.cuda.current_stream()
        for k in range(K):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                models[k](x[k * n:(k + 1) * n], out=[y[k * n:(k + 1) * n] for y in ys])
        for k in range(K):
            cur.wait_stream(streams[k])
        return pipe.postprocess(ys, hw)
    ms = timeit(step)
    det, cnt = step()
    torch.cuda.synchronize()
    same = bool((cnt == ref[1]).all()) and bool((det == ref[0]).all())
    print('split into %d   %.4f ms   identical=%s' % (K, ms, same))
