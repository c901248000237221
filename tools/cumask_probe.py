#!/usr/bin/env python
"""Steps in flight on DISJOINT CU partitions (hipExtStreamCreateWithCUMask): every context's stream owns 256 / n CUs, so a
kernel has n times the workgroups per CU it would have on the whole chip (smaller tails) and the contexts never compete for a
CU's LDS / registers.   python tools/cumask_probe.py [model] [size] [batch] [dtype]"""
import ctypes, os, sys, time
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
torch.cuda.set_device(dev)
anchors = get_anchors('model_data/yolo_anchors.txt')
hip = ctypes.CDLL('libamdhip64.so')
NCU = torch.cuda.get_device_properties(dev).multi_processor_count


def masked_stream(cus):
    words = (NCU + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in cus:
        mask[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


L.set_global_policy({'f32': 'float32', 'bf16': 'mixed_bfloat16', 'f16': 'mixed_float16'}[dt])
m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
L.set_global_policy('float32')
m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
x = torch.from_numpy(W.synthetic_images(B, size, size)).to(dev)
hw = torch.tensor([[size, size]] * B, dtype=torch.int32, device=dev)


def run(depth, streams=None, steps=40):
    p = DetectionPipeline(m, anchors, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=depth)
    if streams:
        for c, s in zip(p._ctx, streams):
            c['stream'] = s
    for _ in range(60):
        p(x, hw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        p(x, hw)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


print('%s@%d %s B=%d, %d CUs' % (name, size, dt, B, NCU))
print('  depth 3, ordinary streams: %.0f img/s' % run(3))
for n in (2, 3, 4, 8):
    per = NCU // n
    for how in ('contiguous', 'interleaved'):
        if how == 'contiguous':
            parts = [range(i * per, (i + 1) * per) for i in range(n)]
        else:
            parts = [range(i, NCU, n) for i in range(n)]
        try:
            ss = [masked_stream(list(pt)) for pt in parts]
            print('  %d partitions of %d CUs (%s): %.0f img/s' % (n, per, how, run(n, ss)))
        except Exception as e:
            print('  %d partitions (%s): failed: %s' % (n, how, e))
