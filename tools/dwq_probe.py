#!/usr/bin/env python
"""16-bit k x k stride-1 depthwise on the maps of the EfficientNets: the form the environment selects (YOLORET_DW_WALK=0: the
tile walk of depthwise_lds.hip, YOLORET_DW_LDS=0: dw_kernel straight from global memory; default: depthwise_walk.hip), its
time per launch and a digest of the output - tools/dwq_probe.sh runs the three forms and compares the digests (the forms
are bit-identical).  GPU: python tools/dwq_probe.py [se]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloret_amd import runtime as rt
from yoloret_amd.compiler import dwl_geometry, dw_se_geometry, DW_LDS

dev = torch.device('cuda:0')
se = len(sys.argv) > 1 and sys.argv[1] == 'se'
SHAPES = [('bf16', 'B0 s3 52x52x240 k5', 128, 52, 240, 5, 'swish'), ('bf16', 'B0 s5 26x26x480 k5', 128, 26, 480, 5, 'relu6'),
          ('bf16', 'B0 s5 26x26x672 k5', 128, 26, 672, 5, 'relu6'), ('bf16', 'B0 s6 13x13x1152 k5', 128, 13, 1152, 5, 'relu6'),
          ('bf16', 'head 52x52x128 k3', 128, 52, 128, 3, 'relu6'), ('bf16', 'head 26x26x256 k3', 128, 26, 256, 3, 'relu6'),
          ('bf16', 'head 13x13x512 k3', 128, 13, 512, 3, 'relu6'),
          ('f16', 'B3 s3 80x80x288 k5', 32, 80, 288, 5, 'swish'), ('f16', 'B3 s5 40x40x576 k5', 32, 40, 576, 5, 'relu6'),
          ('f16', 'B3 s5 40x40x816 k5', 32, 40, 816, 5, 'relu6'), ('f16', 'B3 s6 20x20x1392 k5', 32, 20, 1392, 5, 'relu6'),
          ('f16', 'head 80x80x128 k3', 32, 80, 128, 3, 'relu6'), ('f16', 'odd 33x19x72 k5 b3', 3, 33, 72, 5, 'swish')]
ONLY = os.environ.get('YR_PROBE_ONLY', '')
for dt, name, b, h, c, k, act in SHAPES:
    if ONLY and not any(o in name for o in ONLY.split(',')):
        continue
    w_ = 19 if name.startswith('odd') else h
    did = rt.dtype_id(dt)
    V = rt.VEC[did]
    ldc = (c + V - 1) // V * V
    g = torch.Generator(device='cpu').manual_seed(h * 1000 + c)
    x = torch.randn((b, h, w_, ldc), generator=g).to(dev).to(rt.TORCH_DTYPE[did])
    out = torch.empty_like(x)
    w = (torch.randn((k * k, ldc), generator=g) * 0.3).to(dev)
    sc, sh = (1 + 0.1 * torch.randn(ldc, generator=g)).to(dev), (0.1 * torch.randn(ldc, generator=g)).to(dev)
    op = rt.new_op(rt.OP_DEPTHWISE, act)
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = h, w_, c, c, k, 1, 1
    op.src[0] = rt.make_src(x, c=c)
    op.wgt, op.scale, op.shift = w.data_ptr(), sc.data_ptr(), sh.data_ptr()
    op.out, op.out_ld = out.data_ptr(), ldc
    part = None
    if se:
        rows = dw_se_geometry(h * ((w_ + 3) // 4), (c + V - 1) // V)[2]
        if DW_LDS and c >= 64:
            ntx, nty = dwl_geometry(h, w_, k)
            rows = ntx * nty
        part = torch.zeros((b, rows, ldc), dtype=torch.float32, device=dev)
        op.gate, op.gate_ld, op.se_reduced = part.data_ptr(), ldc, rows
    out.zero_()
    rt.run_op(op, b)
    torch.cuda.synchronize()
    dig = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10]
    extra = ''
    if se:
        s = part.sum(dim=1).double()[:, :c]
        ref = out.double().sum(dim=(1, 2))[:, :c]
        extra = '  se_err %.2e' % ((s - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        rt.run_op(op, b)
    e0.record()
    for _ in range(20):
        rt.run_op(op, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = 2 * b * h * w_ * ldc * 2 / 1e6
    print('%-22s %s %7.1f us  %6.0f GB/s  %s%s' % (name, dt, us, mb / us, dig, extra), flush=True)
