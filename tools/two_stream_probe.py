#!/usr/bin/env python
"""Does running TWO half-batches concurrently on two HIP streams beat one full batch?  Every kernel of this path is bound by
latency / occupancy, and many 26x26 / 13x13 launches are one wave of workgroups with a long tail: a second independent
stream gives the dispatcher something to fill those holes with.
    python tools/two_stream_probe.py [model] [size] [batch] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloret_amd import layers as L, weights as W
from yoloret_amd.pipeline import DetectionPipeline
from yoloret_amd.yolo3.model import yolov3_body
from yoloret_amd.yolo3.utils import get_anchors

name = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2x75'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dt = sys.argv[4] if len(sys.argv) > 4 else 'f32'
dev = torch.device('cuda:0')
anchors = get_anchors('model_data/yolo_anchors.txt')


def make(b):
    L.set_global_policy({'f32': 'float32', 'bf16': 'mixed_bfloat16', 'f16': 'mixed_float16'}[dt])
    m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
    L.set_global_policy('float32')
    m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
    p = DetectionPipeline(m, anchors, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
    x = torch.from_numpy(W.synthetic_images(b, size, size)).to(dev)
    hw = torch.tensor([[size, size]] * b, dtype=torch.int32, device=dev)
    return p, x, hw


def run(parts, steps=30):
    streams = [torch.cuda.Stream(dev) for _ in parts]
    def step():
        for (p, x, hw), s in zip(parts, streams):
            with torch.cuda.stream(s):
                p(x, hw)
    for _ in range(60):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


one = make(B)
t1 = run([one])
print('%s@%d %s: one stream, batch %d: %.4f ms/step  %.0f img/s' % (name, size, dt, B, t1 * 1e3, B / t1))
for nparts in (2, 3, 4):
    if B % nparts:
        continue
    parts = [make(B // nparts) for _ in range(nparts)]
    t = run(parts)
    print('   %d streams x batch %d: %.4f ms/step  %.0f img/s  (%+.1f %%)' % (nparts, B // nparts, t * 1e3, B / t, 100 * (t1 / t - 1)))
for n in (2, 3, 4):
    parts = [make(B) for _ in range(n)]
    t = run(parts)
    print('   %d steps in flight x batch %d: %.4f ms per %d steps  %.0f img/s  (%+.1f %%)' % (n, B, t * 1e3, n, n * B / t, 100 * (n * t1 / t - 1)))
