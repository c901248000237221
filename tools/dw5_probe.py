#!/usr/bin/env python
"""5x5 stride-1 depthwise (the EfficientNet stages MBX does not take: more than 128 block inputs) under different
output patches per lane.  Needs the library built with YOLORET_HIPCC_FLAGS=-DYR_DW_EXPERIMENT; YR_DW_FORCE = XT*10+YT
(unset: the product's 4x1).  GPU: python tools/dw5_probe.py [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloret_amd import runtime as rt

dev = torch.device('cuda:0')
dt = sys.argv[1] if len(sys.argv) > 1 else 'f16'
did = rt.dtype_id(dt)
V = rt.VEC[did]
for name, b, h, c in [('B3 stage5 40x40', 32, 40, 816), ('B3 stage6 20x20', 32, 20, 1392), ('B0 stage5 26x26', 128, 26, 672),
                      ('B0 stage6 13x13', 128, 13, 1152), ('B0 stage3 52x52', 128, 52, 240)]:
    ldc = (c + V - 1) // V * V
    x = torch.randn((b, h, h, ldc), device=dev).to(rt.TORCH_DTYPE[did])
    out = torch.empty_like(x)
    w = torch.randn((25, ldc), device=dev)
    sc, sh = torch.ones(ldc, device=dev), torch.zeros(ldc, device=dev)
    op = rt.new_op(rt.OP_DEPTHWISE, sys.argv[2] if len(sys.argv) > 2 else 'swish')
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = h, h, c, c, 5, 1, 1
    op.src[0] = rt.make_src(x, c=c)
    op.wgt, op.scale, op.shift = w.data_ptr(), sc.data_ptr(), sh.data_ptr()
    op.out, op.out_ld = out.data_ptr(), ldc
    res = []
    ref = None
    for f in ('lds', '', 'lds1', 'lds2'):
        os.environ['YOLORET_DW_LDS'] = '1' if f.startswith('lds') else '0'
        os.environ['YR_DWL_DBG'] = f[3:] if f.startswith('lds') and len(f) > 3 else '0'
        os.environ['YR_DW_FORCE'] = '' if f.startswith('lds') else f
        out.zero_()
        rt.run_op(op, b)
        torch.cuda.synchronize()
        if f == 'lds':
            lds_out = out.clone()
        elif f == '':
            same = torch.equal(lds_out.view(torch.int16), out.view(torch.int16))
            res.append('bit-identical' if same else 'DIFFERENT (max %.3g)' % (lds_out.float() - out.float()).abs().max().item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rt.run_op(op, b)
        e1.record()
        torch.cuda.synchronize()
        res.append('%s %.1f' % (f or '41', e0.elapsed_time(e1) / 20 * 1e3))
    gb = 2 * b * h * h * ldc * rt.ESIZE[did] / 1e9
    print('%-18s %s us   (in+out %.0f MB)' % (name, '  '.join(res), gb * 1e3))
