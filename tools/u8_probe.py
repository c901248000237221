"""Where the host-fed uint8 path loses time: resident float32 / resident uint8 / fed uint8 with 4, 6, 8 feeder slots, and the H2D
copy alone (config 2, depth 3).  GPU: python tools/u8_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoloret_amd import layers as L
from yoloret_amd.pipeline import DetectionPipeline, HostFeeder
from yoloret_amd.weights import synthetic_weights
from yoloret_amd.yolo3.model import yolov3_body
from yoloret_amd.yolo3.utils import get_anchors
dev = torch.device('cuda:0')
anchors = get_anchors('model_data/yolo_anchors.txt')
b, size, depth = 64, 416, 3
mf = yolov3_body(L.Input(shape=[size, size, 3]), 'mobilenetv2x75', 3, num_classes=20)
m8 = yolov3_body(L.Input(shape=[size, size, 3], dtype='uint8'), 'mobilenetv2x75', 3, num_classes=20)
w = synthetic_weights(mf, 1, 'survey'); mf.set_weights(w); m8.set_weights(w)
hw = torch.tensor([[size, size]] * b, dtype=torch.int32, device=dev)
u8h = torch.randint(0, 256, (b, size, size, 3), dtype=torch.uint8).pin_memory()
u8d = u8h.to(dev); xf = (u8d.float() / 255).contiguous()
def run(pipe, x, n=40):
    for _ in range(8): pipe(x, hw)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): pipe(x, hw)
    torch.cuda.synchronize(); return b * n / (time.perf_counter() - t)
pf = DetectionPipeline(mf, anchors, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=depth)
p8 = DetectionPipeline(m8, anchors, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=depth)
for rep in range(2):
    print('resident f32 %.0f   resident u8 %.0f' % (run(pf, xf), run(p8, u8d)))
feeder = HostFeeder(tuple(u8h.shape), (size, size), dev, slots=depth + 1)
def step():
    feeder.submit(u8h); xb, slot = feeder.take_raw(); p8(xb, hw); feeder.mark_released(slot, p8.done)
feeder.submit(u8h)
for _ in range(10): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(40): step()
torch.cuda.synchronize(); print('fed u8 %.0f' % (b * 40 / (time.perf_counter() - t)))
# H2D alone
cs = torch.cuda.Stream(dev)
dst = [torch.empty_like(u8d) for _ in range(4)]
torch.cuda.synchronize(); t = time.perf_counter()
with torch.cuda.stream(cs):
    for i in range(40): dst[i % 4].copy_(u8h, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 40
print('H2D alone: %.3f ms per batch = %.1f GB/s' % (dt * 1e3, u8h.numel() / dt / 1e9))
# fed, depth sweep of slots
for slots in (4, 6, 8):
    fd = HostFeeder(tuple(u8h.shape), (size, size), dev, slots=slots)
    def step2():
        fd.submit(u8h); xb, slot = fd.take_raw(); p8(xb, hw); fd.mark_released(slot, p8.done)
    fd.submit(u8h)
    for _ in range(10): step2()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(40): step2()
    torch.cuda.synchronize(); print('fed u8 slots %d: %.0f' % (slots, b * 40 / (time.perf_counter() - t)))
