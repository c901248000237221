#!/usr/bin/env python
"""Latency of the squeeze-excite FC op (YR_OP_SE_FC) on the head shapes of the bench workload, fed by per-workgroup
partial sums (rows > 1, k = pixel count) or by an already pooled vector; next to it an empty-ish launch (WSUM of one tiny
map) for the launch floor.  GPU: python tools/sefc_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloret_amd import runtime as rt

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


COLD = os.environ.get('SEFC_COLD', '0') != '0'   # 1: a 1.5 GB fill between launches (caches cold, as inside a real step)
junk = torch.empty(3 * 2 ** 27, dtype=torch.float32, device=dev) if COLD else None


def timed(op, n=50):
    rt.run_op(op, B)
    torch.cuda.synchronize()
    if not COLD:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            rt.run_op(op, B)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tot = 0.0
    for _ in range(10):
        junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rt.run_op(op, B)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / 10 * 1e3


for name, rows, c, r, px in [('td1/bu1 13x13', 11, 384, 16, 169), ('td2/bu2 26x26', 43, 192, 8, 676), ('td3/bu3 52x52', 169, 96, 4, 2704),
                             ('B0 s2b0 mbx', 338, 96, 4, 10816), ('B0 s5b1 mbx', 28, 672, 28, 676), ('B0 s6b1 merged 13x13', 0, 1152, 48, 169)]:
    ldc = (c + 3) // 4 * 4
    w1 = torch.randn((r, ldc), device=dev) * 0.1
    w2 = torch.randn((r, ldc), device=dev) * 0.1
    b1, b2 = torch.zeros(r, device=dev), torch.zeros(ldc, device=dev)
    gate = torch.empty((B, ldc), device=dev)
    res = []
    for mode in ('partials', 'pooled'):
        op = rt.new_op(rt.OP_SE_FC, 'none')
        op.cin = op.cout = c
        op.se_reduced, op.nsrc, op.h, op.w = r, 1, 1, 1
        if mode == 'partials':
            if rows == 0:
                src = torch.randn((B, px, 1, ldc), device=dev)      # merged mean over the full map (float32 here)
                op.k = 0
            else:
                src = torch.randn((B, rows, 1, ldc), device=dev)
                op.k = px
        else:
            src = torch.randn((B, 1, 1, ldc), device=dev)
            op.k = 0
        op.src[0] = rt.make_src(src, c=c)
        op.wgt, op.b1, op.wgt2, op.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
        op.out, op.out_ld = gate.data_ptr(), ldc
        res.append(timed(op))
    print('%-22s C %4d R %2d rows %3d: partial sums %.1f us   pooled vector %.1f us' % (name, c, r, rows, res[0], res[1]))
