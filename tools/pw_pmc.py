#!/usr/bin/env python
"""Runs the flagship forward with ONE pointwise op on tile shape `cfg` and every other pointwise op on `other`, so a
rocprofv3 --pmc pass over this script isolates that op's counters under its kernel name:
    rocprofv3 --kernel-trace --pmc ... -- python tools/pw_pmc.py block_11_project 12 [other=7]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['YOLORET_AUTOTUNE'] = '0'
from yoloret_amd import layers as L, runtime as rt, weights as W
from yoloret_amd.yolo3.model import yolov3_body

name, cfg = sys.argv[1], int(sys.argv[2])
other = int(sys.argv[3]) if len(sys.argv) > 3 else 7
B = int(os.environ.get('PW_PMC_BATCH', '64'))
dev = torch.device('cuda', 0)
m = yolov3_body(L.Input(shape=[416, 416, 3]), 'mobilenetv2x75', 3, num_classes=20)
m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
x = torch.from_numpy(W.synthetic_images(B, 416, 416)).to(dev)
m(x)
idx, hd = m._handle(dev)
n = len(m.plan.ops)
tab = (ctypes.c_int32 * n)()
for k, o in enumerate(m.plan.ops):
    if o.kind == rt.OP_POINTWISE:
        tab[k] = cfg if o.name == name else other
rt.check(rt.lib().yr_set_tuning(hd, B, tab, n))
for _ in range(5):
    m(x)
torch.cuda.synchronize()
row = [r for r in m.profile(x, iters=5) if r['name'] == name][0]
print(name, row['kernel'], 'B=%d %.4f ms = %.4f ms per 64 images' % (B, row['ms'], row['ms'] * 64 / B))
