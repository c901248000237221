"""Debug aid: SE_FC alone on one stream while the full model runs on two others."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoloret_amd import runtime as rt, layers as L
from yoloret_amd.weights import synthetic_weights, synthetic_images
from yoloret_amd.yolo3.model import yolov3_body
dev = torch.device('cuda:0')
b, c, r, rows = 64, 512, 128, 1
size = 416
m = yolov3_body(L.Input(shape=[size, size, 3]), os.environ.get('BACKBONE', 'mobilenetv2x75'), 3, num_classes=20)
m.set_weights(synthetic_weights(m, 1234, 'survey'))
x = torch.from_numpy(synthetic_images(b, size, size, seed=21)).to(dev)
m(x, ctx=1); m(x, ctx=2)
if os.environ.get('LIST'):
    for i, o in enumerate(m.plan.ops):
        print(i, o.name, o.kind)
rng = np.random.default_rng(1)
r4 = (r + 3) & ~3
w1 = torch.from_numpy((rng.standard_normal((c, r4)) * np.sqrt(2.0 / c)).astype(np.float32)).to(dev)
b1 = torch.from_numpy(rng.normal(0, 0.1, r4).astype(np.float32)).to(dev)
w2 = torch.from_numpy((rng.standard_normal((r, c)) * np.sqrt(2.0 / r)).astype(np.float32)).to(dev)
b2 = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32)).to(dev)
sums = torch.from_numpy(rng.standard_normal((b, rows, 1, c)).astype(np.float32) * 50).to(dev)
NG = 64
gates = [torch.zeros((b, 1, 1, c), dtype=torch.float32, device=dev) for _ in range(NG)]
ops = []
for g in gates:
    op = rt.new_op(rt.OP_SE_FC)
    op.h, op.w, op.cin, op.cout, op.nsrc, op.se_reduced, op.k = 1, 1, c, c, 1, r, 169
    op.src[0] = rt.make_src(sums, c=c)
    op.wgt, op.b1, op.wgt2, op.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    op.out, op.out_ld = g.data_ptr(), c
    ops.append(op)
rt.run_op(ops[0], b)
torch.cuda.synchronize()
want = gates[0].cpu().numpy().copy()
streams = [torch.cuda.Stream(dev) for _ in range(3)]
bad = 0
for it in range(int(os.environ.get('ITERS', 10))):
    if os.environ.get('NOISE', '1') != '0':
        for i in (1, 2):
            with torch.cuda.stream(streams[i]):
                for _ in range(int(os.environ.get('REP', 1))):
                    m(x, ctx=i)
    with torch.cuda.stream(streams[0]):
        for op in ops:
            rt.run_op(op, b)
    torch.cuda.synchronize()
    for g in gates:
        if not np.array_equal(g.cpu().numpy(), want):
            bad += 1
print('%d of %d isolated SE_FC launches differ beside the model' % (bad, NG * int(os.environ.get('ITERS', 10))))
