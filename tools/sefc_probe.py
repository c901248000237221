"""Debug aid: SE_FC alone (partial-sum rows -> gate) on three streams at once beside unrelated work, against its serial result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoloret_amd import runtime as rt
dev = torch.device('cuda:0')
b, c, r, rows = 64, int(os.environ.get('C', 512)), int(os.environ.get('R', 128)), int(os.environ.get('ROWS', 1))
rng = np.random.default_rng(1)
r4 = (r + 3) & ~3
w1 = torch.from_numpy((rng.standard_normal((c, r4)) * np.sqrt(2.0 / c)).astype(np.float32)).to(dev)
b1 = torch.from_numpy(rng.normal(0, 0.1, r4).astype(np.float32)).to(dev)
w2 = torch.from_numpy((rng.standard_normal((r, c)) * np.sqrt(2.0 / r)).astype(np.float32)).to(dev)
b2 = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32)).to(dev)
ctxs = []
for i in range(3):
    sums = torch.from_numpy(rng.standard_normal((b, rows, 1, c)).astype(np.float32) * 50).to(dev)
    gate = torch.zeros((b, 1, 1, c), dtype=torch.float32, device=dev)
    op = rt.new_op(rt.OP_SE_FC)
    op.h, op.w, op.cin, op.cout, op.nsrc, op.se_reduced, op.k = 1, 1, c, c, 1, r, 169
    op.src[0] = rt.make_src(sums, c=c)
    op.wgt, op.b1, op.wgt2, op.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    op.out, op.out_ld = gate.data_ptr(), c
    rt.run_op(op, b)
    torch.cuda.synchronize()
    ctxs.append((op, sums, gate, gate.cpu().numpy().copy()))
streams = [torch.cuda.Stream(dev) for _ in range(3)]
noise = [torch.randn(2048, 2048, device=dev) for _ in range(3)]
bad = 0
for it in range(int(os.environ.get('ITERS', 200))):
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            ctxs[i][2].zero_()
            y = noise[i] @ noise[i]
            rt.run_op(ctxs[i][0], b)
            y = noise[i] @ noise[i]
    torch.cuda.synchronize()
    for i in range(3):
        g = ctxs[i][2].cpu().numpy()
        if not np.array_equal(g, ctxs[i][3]):
            bad += 1
            if bad < 4:
                d = [j for j in range(b) if not np.array_equal(g[j], ctxs[i][3][j])]
                print('iter %d ctx %d: images %s' % (it, i, d))
print('C %d R %d rows %d: %d mismatches' % (c, r, rows, bad))
