import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from yoloret_amd import runtime as rt, layers as L, weights as W
from yoloret_amd.pipeline import DetectionPipeline
from yoloret_amd.yolo3.model import yolov3_body
from yoloret_amd.yolo3.utils import get_anchors
dev = torch.device('cuda:0')
m = yolov3_body(L.Input(shape=[416, 416, 3]), 'mobilenetv2x75', 3, num_classes=20)
m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
pipe = DetectionPipeline(m, get_anchors('model_data/yolo_anchors.txt'), 20)
x = torch.from_numpy(W.synthetic_images(64, 416, 416)).to(dev)
hw = torch.tensor([[416, 416]] * 64, dtype=torch.int32, device=dev)
pipe(x, hw)
v = pipe._buffers(64, dev)
boxes, scores = v['boxes'], v['scores']
print('cand frac >0.2: %.3f  >0.5: %.3f' % ((scores > 0.2).float().mean().item(), (scores > 0.5).float().mean().item()))
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for thr, mb in [(0.2, 20), (0.2, 5), (0.2, 1), (0.5, 20), (0.9, 20), (0.999, 20)]:
    ms = t(lambda: rt.nms(boxes, scores, mb, thr, 0.5))
    idx, cnt = rt.nms(boxes, scores, mb, thr, 0.5)
    print('thr %.3f max_boxes %2d: %.3f ms  (avg picks %.1f)' % (thr, mb, ms, cnt.float().mean().item()))
