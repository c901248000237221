#!/usr/bin/env python
"""16-bit pointwise GEMM shapes of the unfused EfficientNet stages (expand: small K, wide N - write-bound; project: the
reverse) under every tile shape of pointwise_h.hip, stand-alone.  GPU: python tools/pwh_probe.py [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloret_amd import runtime as rt

dev = torch.device('cuda:0')
dt = sys.argv[1] if len(sys.argv) > 1 else 'f16'
did = rt.dtype_id(dt)
tdt = rt.TORCH_DTYPE[did]
ncfg = 26
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1      # shape index
only_cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # tile shape (1-based), 0 = all
shape_i = -1
for name, b, h, k, n in [('lite0 s5 expand', 128, 26, 112, 672), ('lite0 s5 project', 128, 26, 672, 112), ('lite3 s5 expand', 32, 40, 136, 816), ('lite3 s5 project', 32, 40, 816, 136), ('lite3 s6 expand', 32, 20, 232, 1392),
                         ('lite3 s6 project', 32, 20, 1392, 232), ('lite0 s6 expand', 128, 13, 192, 1152), ('lite0 s6 project', 128, 13, 1152, 192)]:
    shape_i += 1
    if only >= 0 and shape_i != only:
        continue
    kp, ldo = (k + 7) // 8 * 8, (n + 7) // 8 * 8
    x = torch.randn((b, h, h, kp), device=dev).to(tdt)
    w = (torch.randn((n, kp), device=dev) * 0.05).to(tdt)
    out = torch.empty((b, h, h, ldo), device=dev, dtype=tdt)
    sc, sh = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    op = rt.new_op(rt.OP_POINTWISE, 'relu6')
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.nsrc = h, h, k, n, 1
    op.src[0] = rt.make_src(x, c=k)
    op.wgt, op.scale, op.shift = w.data_ptr(), sc.data_ptr(), sh.data_ptr()
    op.out, op.out_ld = out.data_ptr(), ldo
    res = []
    for cfg in ([only_cfg] if only_cfg else range(1, ncfg + 1)):
        op.k = cfg
        for _ in range(3):
            rt.run_op(op, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rt.run_op(op, b)
        e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / 20 * 1e3, cfg))
    mb = b * h * h * (kp + ldo) * 2 / 1e6
    print('%-18s %.0f MB  ' % (name, mb) + '  '.join('c%d %.1f' % (c, t) for t, c in sorted(res)[:5]) + '  |  ' + '  '.join('c%d %.1f' % (c, t) for t, c in res[18:]) + '   best %.2f TB/s' % (mb / min(res)[0]))
