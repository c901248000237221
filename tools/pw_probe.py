#!/usr/bin/env python
"""Per-layer timing of every pointwise tile shape (LDS-staged 1..14, direct 15..29) inside the real forward pass:
    [B=64] python tools/pw_probe.py name [name ...]      (default: the gated project convs of the heads)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yoloret_amd import layers as L, runtime as rt, weights as W
from yoloret_amd.yolo3.model import yolov3_body

names = sys.argv[1:] or ['bu3_mb_project', 'bu2_mb_project', 'bu1_mb_project', 'td3_mb_project']
dev = torch.device('cuda', 0)
os.environ['YOLORET_AUTOTUNE'] = '0'
m = yolov3_body(L.Input(shape=[416, 416, 3]), 'mobilenetv2x75', 3, num_classes=20)
m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
B = int(os.environ.get('B', '64'))
x = torch.from_numpy(W.synthetic_images(B, 416, 416)).to(dev)
m(x)
idx, hd = m._handle(dev, B)
plan = m.plan_for(B)
n = len(plan.ops)
shapes = ['%dx%d' % s for s in [(256, 16), (128, 32), (128, 48), (128, 64), (128, 80), (128, 96), (128, 128), (64, 16), (64, 32),
                                 (64, 48), (64, 64), (64, 80), (64, 96), (64, 128)]] + \
         ['d%dx%d' % s for s in [(64, 16), (64, 32), (64, 48), (64, 64), (64, 80), (64, 96), (64, 128), (128, 32), (128, 48),
                                  (128, 64), (128, 80), (128, 96), (256, 32), (256, 48), (256, 64)]]
for name in names:
    i = next(k for k, o in enumerate(plan.ops) if o.name == name)
    o = plan.ops[i]
    row = []
    for cfg in range(1, 30):
        tab = (ctypes.c_int32 * n)()
        tab[i] = cfg
        rt.check(rt.lib().yr_set_tuning(hd, B, tab, n))
        ms = m.profile(x, iters=20)[i]['ms']
        row.append((ms, shapes[cfg - 1]))
    best = sorted(row)[:6]
    print('%-18s K=%d N=%d %dx%d  ' % (name, o.cin, o.cout, o.h, o.w) + '  '.join('%s %.4f' % (s, t) for t, s in best)
          + '   | best direct ' + '%s %.4f' % min((t, s) for t, s in row if s.startswith('d'))[::-1])
