"""Per-phase shader-clock totals of the weight-streaming head kernel (a -DHS_TIMING build: python tools/relink.py headstream.hip -DHS_TIMING).
    python tools/hs_timing.py [td1|td2|bu2|bu1|td3|bu3] [batch]"""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from yoloret_amd import runtime as rt
from yoloret_amd.compiler import head_pack, head_stream_geometry
from tools.head_probe import HEADS
name = sys.argv[1] if len(sys.argv) > 1 else 'td2'
b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
h, w, segs, f, pre, gated, r, _ = HEADS[name]
dev = torch.device('cuda:0')
rng = np.random.default_rng(1)
ru = lambda v, m: (v + m - 1) // m * m
keep = []
op = rt.new_op(rt.OP_HEAD, 'swish')
n = 0
for c, xf in segs:
    sh, sw = (h, w) if xf == 'identity' else (2 * h, 2 * w)
    t = torch.from_numpy(rng.standard_normal((b, sh, sw, ru(c, 4))).astype(np.float32)).to(dev)
    keep.append(t)
    op.src[n] = rt.make_src(t, c=c, xform=xf)
    n += 1
if pre:
    t = torch.from_numpy(rng.standard_normal((b, h // 2, w // 2, f)).astype(np.float32)).to(dev)
    keep.append(t)
    op.src[n] = rt.make_src(t, c=f, xform='up2_add')
    n += 1
op.nsrc = n
cin, kp = sum(c for c, _ in segs), sum(ru(c, 4) for c, _ in segs)
wt = (rng.standard_normal((f, kp)) * np.sqrt(2.0 / cin)).astype(np.float32)
par = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev) for a in
       (head_pack(wt, [c for c, _ in segs]), rng.uniform(0.5, 1.5, f), rng.standard_normal((f // 16, 11, 16)) * 0.3)]
keep += par
op.wgt, op.scale, op.wgt2 = [p.data_ptr() for p in par]
if gated:
    g = torch.from_numpy(rng.uniform(0.1, 1.0, (b, ru(cin, 4))).astype(np.float32)).to(dev)
    keep.append(g)
    op.res, op.res_ld = g.data_ptr(), g.shape[1]
op.h, op.w, op.cin, op.cout, op.stride = h, w, cin, f, 1
op.k = 3 | rt.ACT['relu6'] << 8 | 0x60
rpw, nw, strips, nsegs = head_stream_geometry(h, w)
rows = strips * nsegs * nw
sums = torch.zeros((b, rows, f), dtype=torch.float32, device=dev)
op.gate, op.gate_ld, op.se_reduced = sums.data_ptr(), f, rows
out_elems = b * h * w * f
big = torch.zeros(out_elems + 16384 * 8 * 16, dtype=torch.float32, device=dev)
op.out, op.out_ld = big.data_ptr(), f
for _ in range(3):
    rt.run_op(op, b)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    rt.run_op(op, b)
ev[1].record()
torch.cuda.synchronize()
wgs = b * strips * nsegs
cs = 1 if wgs >= 200 else min(4, (256 + wgs - 1) // wgs)
while cs > 1 and (f // 32) // cs < 2:
    cs -= 1
nwg = wgs * cs
t = big.view(torch.int32)[out_elems:out_elems + nwg * nw * 16].cpu().numpy().astype(np.int64).reshape(nwg, nw, 16)[:, :, :12] & 0xffffffff
names = ['first pair (plain order), park', 'wait + barrier', 'addend loads + chunk issue', 'first reads of a turn', 'slices (conv | taps, Swish, stores)', 'clamp + park', 'last barrier', '-', 'start: descriptors, DMA issue', 'gather + cut', 'wait for the first planes + barrier', '-']
tot = t.sum(axis=2)
print('%s batch %d: %.1f us; %d workgroups (%d per strip segment) x %d waves; cycles per wave: mean %.0f, max %.0f' % (name, b, ev[0].elapsed_time(ev[1]) * 50, nwg, cs, nw, tot.mean(), tot.max()))
for i, nm in enumerate(names):
    print('  %-36s mean %8.0f  (%.1f %%)   slowest wave of a workgroup, mean %8.0f' % (nm, t[:, :, i].mean(), 100.0 * t[:, :, i].mean() / tot.mean(), t[:, :, i].max(axis=1).mean()))
