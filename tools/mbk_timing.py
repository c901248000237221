"""Per-phase shader-clock totals of the weight-streaming block kernel (a -DMBK_TIMING build: python tools/relink.py mbk.hip -DMBK_TIMING).
    python tools/mbk_timing.py [block] [batch]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from tests.test_gpu_mbr import make_block_k
from yoloret_amd import runtime as rt
from tools.mbk_probe import BLOCKS
name = sys.argv[1] if len(sys.argv) > 1 else 'block_11'
b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
shape = BLOCKS[name][0]
dev = torch.device('cuda:0')
op, out, params, keep = make_block_k(shape, dev, b=b, seed=1)
h, w, cin, cexp, cout, s, res, rows, nw = shape
big = torch.zeros(out.numel() + 4096 * 8 * 8, dtype=torch.float32, device=dev)
op.out = big.data_ptr()
for _ in range(3):
    rt.run_op(op, b)
torch.cuda.synchronize()
strips = -(-((w + s - 1) // s) // (14 // s))
nr = nw * rows
segs = (1 if h <= nr else (h - nr + nr - 3) // (nr - 2) + 1) if s == 1 else 2
nwg = b * strips * segs
t = big.view(torch.int32)[out.numel():out.numel() + nwg * nw * 8].cpu().numpy().astype(np.int64).reshape(nwg, nw, 8) & 0xffffffff
names = ['prologue + pair 0 expand', 'wait + barrier', 'chunk issue + first reads', 'slices (expand | taps)', 'phase 2 (clamp, cut)', 'phase 3 (project | clamps)', 'park + copy', 'last pair + epilogue']
tot = t.sum(axis=2)
print('%s batch %d: %d workgroups x %d waves; cycles per wave: mean %.0f, max %.0f' % (name, b, nwg, nw, tot.mean(), tot.max()))
for i, n in enumerate(names):
    print('  %-28s mean %8.0f  (%.1f %%)   slowest wave of a workgroup, mean %8.0f' % (n, t[:, :, i].mean(), 100.0 * t[:, :, i].mean() / tot.mean(), t[:, :, i].max(axis=1).mean()))
