#!/usr/bin/env python
"""Does a tile table tuned UNDER LOAD (two other steps running concurrently, as in DetectionPipeline(depth=3)) beat the one
tuned on an idle GPU?   python tools/tune_under_load_probe.py [model] [size] [batch] [dtype]"""
import os, sys, time, threading, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloret_amd import layers as L, weights as W, runtime as rt
from yoloret_amd.pipeline import DetectionPipeline
from yoloret_amd.yolo3.model import yolov3_body
from yoloret_amd.yolo3.utils import get_anchors

name = sys.argv[1] if len(sys.argv) > 1 else 'efficientnetb0-lite'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dt = sys.argv[4] if len(sys.argv) > 4 else 'bf16'
dev = torch.device('cuda:0')
anchors = get_anchors('model_data/yolo_anchors.txt')


def make():
    L.set_global_policy({'f32': 'float32', 'bf16': 'mixed_bfloat16', 'f16': 'mixed_float16'}[dt])
    m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
    L.set_global_policy('float32')
    m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
    return m


x = torch.from_numpy(W.synthetic_images(B, size, size)).to(dev)
hw = torch.tensor([[size, size]] * B, dtype=torch.int32, device=dev)


def throughput(m, depth=3, steps=30):
    p = DetectionPipeline(m, anchors, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=depth)
    for _ in range(60):
        p(x, hw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        p(x, hw)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


m = make()
m(x)                                   # isolated autotune
torch.cuda.synchronize()
print('%s@%d %s B=%d  tuned idle:        depth 3 %.0f img/s   depth 1 %.0f img/s' % (name, size, dt, B, throughput(m), throughput(m, 1)))
# background load: two more model instances stepping on their own streams from a thread
bg = [make(), make()]
for g in bg:
    g(x)
torch.cuda.synchronize()
stop = False


def loop():
    ss = [torch.cuda.Stream(dev) for _ in bg]
    while not stop:
        for g, s in zip(bg, ss):
            with torch.cuda.stream(s):
                g(x)
        for s in ss:
            s.synchronize()


th = threading.Thread(target=loop)
th.start()
time.sleep(0.2)
idx, hd = m._handle(dev, B)
ws = m._workspace[idx]
ys = [torch.empty((B, ob.h, ob.w, ob.c), dtype=torch.float32, device=dev) for ob in m.plan.output_bufs]
rt.check(rt.lib().yr_autotune(hd, rt._ptr(x), B, rt._ptr(ys[0]), rt._ptr(ys[1]), rt._ptr(ys[2]), rt._ptr(ws), ws.numel(), rt.stream_ptr(dev), 3))
stop = True
th.join()
torch.cuda.synchronize()
print('%s@%d %s B=%d  tuned under load:  depth 3 %.0f img/s   depth 1 %.0f img/s' % (name, size, dt, B, throughput(m), throughput(m, 1)))
