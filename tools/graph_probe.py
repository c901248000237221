#!/usr/bin/env python
"""Is a whole step replayed as ONE HIP graph faster than its ~75 eager launches at the BENCH batch sizes (it is at batch 1)?
    python tools/graph_probe.py [model] [size] [batch] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloret_amd import layers as L, weights as W
from yoloret_amd.pipeline import DetectionPipeline
from yoloret_amd.yolo3.model import yolov3_body
from yoloret_amd.yolo3.utils import get_anchors
name = sys.argv[1] if len(sys.argv) > 1 else 'efficientnetb0-lite'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dt = sys.argv[4] if len(sys.argv) > 4 else 'bf16'
dev = torch.device('cuda:0')
anchors = get_anchors('model_data/yolo_anchors.txt')
L.set_global_policy({'f32': 'float32', 'bf16': 'mixed_bfloat16', 'f16': 'mixed_float16'}[dt])
m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
L.set_global_policy('float32')
m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
x = torch.from_numpy(W.synthetic_images(B, size, size)).to(dev)
hw = torch.tensor([[size, size]] * B, dtype=torch.int32, device=dev)
p = DetectionPipeline(m, anchors, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
def run(steps=40):
    for _ in range(40):
        p(x, hw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        p(x, hw)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)
e = run()
p.enable_graph(True)
g = run()
print('%s@%d %s B=%d: eager %.0f img/s, one graph per step %.0f img/s (%+.1f %%)' % (name, size, dt, B, e, g, 100 * (g / e - 1)))
