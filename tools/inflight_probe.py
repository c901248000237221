"""Debug aid: does a forward pass depend on what the workspace held before it (a read of bytes nothing wrote)?
Runs config 2 with the workspace pre-filled with zeros, 0xFF (NaN) and 0x7F bytes and compares the logits bit for bit; then per op
(yr_forward_ranges) names the ops that read a NaN.   gpurun -- python tools/inflight_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoloret_amd import layers as L
from yoloret_amd.weights import synthetic_weights, synthetic_images
from yoloret_amd.yolo3.model import yolov3_body

dev = torch.device('cuda:0')
size, b = 416, int(os.environ.get('B', 64))
m = yolov3_body(L.Input(shape=[size, size, 3]), os.environ.get('BACKBONE', 'mobilenetv2x75'), 3, num_classes=20)
m.set_weights(synthetic_weights(m, 1234, 'survey'))
x = torch.from_numpy(synthetic_images(b, size, size, seed=21)).to(dev)
ys = [y.cpu().numpy().copy() for y in m(x)]
torch.cuda.synchronize()
idx = 0
for fill in (0, 0xFF, 0x7F, 0x3F):
    for k in list(m._workspace):
        m._workspace[k].fill_(fill)
    got = [y.cpu().numpy().copy() for y in m(x)]
    torch.cuda.synchronize()
    print('fill %02x:' % fill, [('same' if np.array_equal(a, c) else 'DIFF %.3e nan=%d' % (np.nanmax(np.abs(a - c)), np.isnan(c).sum())) for a, c in zip(ys, got)])
for k in list(m._workspace):
    m._workspace[k].fill_(0xFF)
r = m.check_ranges(x, on_exceed='report')
bad = [n for n, v in r.items() if not np.isfinite(v)]
print('ops reading a NaN with the workspace poisoned:', bad)

# ---- three contexts at once on three streams against the serial pass, raw logits
if os.environ.get('INFLIGHT', '1') != '0':
    xs = [torch.from_numpy(synthetic_images(b, size, size, seed=s)).to(dev) for s in (21, 22, 23)]
    want = []
    for xx in xs:
        want.append([y.cpu().numpy().copy() for y in m(xx)])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    nbad = 0
    for rnd in range(6):
        outs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs.append(m(xs[i], ctx=i + 1))
        torch.cuda.synchronize()
        for i, ys_ in enumerate(outs):
            for j, y in enumerate(ys_):
                a = y.cpu().numpy()
                if not np.array_equal(a, want[i][j]):
                    nbad += 1
                    d = np.abs(a.astype(np.float64) - want[i][j])
                    print('round %d ctx %d out %d: %d values differ, max %.3e' % (rnd, i + 1, j, int((d > 0).sum()), d.max()))
    print('in flight: %d mismatching outputs' % nbad)
