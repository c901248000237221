"""Times YR_OP_HEAD (csrc/headblock.hip) on the six head blocks of MobileNetV2 x0.75 @416 at batch 64, stand-alone.
    [YR_HEAD_EXP=bits] [YR_HEAD_BM=192|384] python tools/head_probe.py [cfg ...]
YR_HEAD_EXP: 1 skip the k loop, 2 skip the depthwise phase, 4 skip the pre-BN addend, 8 skip the sums (and the tail)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from yoloret_amd import runtime as rt        # noqa: E402
from yoloret_amd.compiler import head_pack    # noqa: E402

# name: (h, w, [(c, xform)], F, pre, gated, R, unfused conv + depthwise + se_fc in us (round 4))
HEADS = {'td1': (13, 13, [(120, 'identity'), (96, 'maxpool2')], 512, False, False, 128, 34.7 + 20.9 + 21.6),
         'td2': (26, 26, [(72, 'identity'), (96, 'identity')], 256, True, False, 64, 49.7 + 28.7 + 13.9),
         'td3': (52, 52, [(24, 'identity')], 128, True, False, 32, 42.7 + 47.4 + 11.8),
         'bu3': (52, 52, [(128, 'identity')], 128, False, True, 32, 56.2 + 44.9 + 11.3),
         'bu2': (26, 26, [(128, 'identity'), (75, 'identity')], 256, False, False, 64, 47.5 + 28.4 + 13.8),
         'bu1': (13, 13, [(256, 'identity'), (75, 'identity')], 512, False, False, 128, 33.8 + 21.0 + 21.2)}


def ru(v, m):
    return (v + m - 1) // m * m


def main():
    dev = torch.device('cuda:0')
    b = 64
    cfgs = [int(a) for a in sys.argv[1:]] or [0]
    rng = np.random.default_rng(1)
    for name, (h, w, segs, f, pre, gated, r, old) in HEADS.items():
        keep = []
        op = rt.new_op(rt.OP_HEAD, 'swish')
        dims = {'identity': (h, w), 'maxpool2': (2 * h, 2 * w)}
        n = 0
        for c, xf in segs:
            sh, sw = dims[xf]
            t = torch.from_numpy(rng.standard_normal((b, sh, sw, ru(c, 4))).astype(np.float32)).to(dev)
            keep.append(t)
            op.src[n] = rt.make_src(t, c=c, xform=xf)
            n += 1
        if pre:
            t = torch.from_numpy(rng.standard_normal((b, h // 2, w // 2, f)).astype(np.float32)).to(dev)
            keep.append(t)
            op.src[n] = rt.make_src(t, c=f, xform='up2_add')
            n += 1
        cin, kp = sum(c for c, _ in segs), sum(ru(c, 4) for c, _ in segs)
        par = [torch.from_numpy(a.astype(np.float32)).to(dev) for a in
               (rng.standard_normal((f, kp)) * np.sqrt(2.0 / cin), rng.uniform(0.5, 1.5, f), rng.normal(0, 0.3, f), rng.standard_normal((10, f)) * 0.3,
                rng.standard_normal(f * ru(r, 4) + r * f + ru(r, 4) + f) * 0.05)]
        op.nsrc, op.h, op.w, op.cin, op.cout, op.stride = n, h, w, cin, f, 1
        import os
        nk = sum((c + 31) // 32 for c, _ in segs)
        walk = os.environ.get('YR_HEAD_FORM', 'walk') == 'walk' and all(xf == 'identity' for _, xf in segs) and nk <= 4 and not (gated and pre)
        packed = not walk and all(xf in ('identity', 'up2') for _, xf in segs)
        if packed:
            par[0] = torch.from_numpy(head_pack(par[0].cpu().numpy(), [c for c, _ in segs])).to(dev)
        if walk:
            par[0] = torch.from_numpy(head_pack(par[0].cpu().numpy() * par[1].cpu().numpy()[:, None], [c for c, _ in segs])).to(dev)
            par[3] = torch.from_numpy((rng.standard_normal((f // 16, 11, 16)) * 0.3).astype(np.float32)).to(dev)
        op.wgt, op.scale, op.shift, op.wgt2 = [p.data_ptr() for p in par[:4]]
        if gated:
            g = torch.rand((b, kp), device=dev)
            keep.append(g)
            op.res, op.res_ld = g.data_ptr(), kp
        out = torch.empty((b, h, w, f), dtype=torch.float32, device=dev)
        nsy, nsx = ctypes.c_int32(), ctypes.c_int32()
        nsx = ctypes.c_int32(1)
        if walk:
            rt.check(rt.lib().yr_head_walk_rows(h, w, ctypes.byref(nsy)))
        else:
            rt.check(rt.lib().yr_head_regions(h, w, ctypes.byref(nsy), ctypes.byref(nsx)))
        rows = nsy.value * nsx.value
        sums = torch.empty((b, rows, f), dtype=torch.float32, device=dev)
        gate = torch.empty((b, f), dtype=torch.float32, device=dev)
        sync = torch.zeros(b, dtype=torch.int32, device=dev)
        op.out, op.out_ld = out.data_ptr(), f
        op.gate, op.gate_ld, op.se_reduced = sums.data_ptr(), f, rows
        op.gate_out, op.gate_out_ld, op.se_hidden, op.se_w, op.sync = gate.data_ptr(), f, r, par[4].data_ptr(), sync.data_ptr()
        moved = sum(t.numel() * 4 for t in keep) + out.numel() * 4
        for cfg in cfgs:
            op.k = 3 | 1 << 8 | cfg << 16 | (0x80 if packed else 0x40 if walk else 0)
            for _ in range(3):
                rt.run_op(op, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                rt.run_op(op, b)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            print('%s %dx%d K %3d F %3d rows %dx%d cfg %d  %6.1f us  %5.2f TB/s moved  (unfused chain: %.1f us)' % (name, h, w, kp, f, nsy.value, nsx.value, cfg, us, moved / us * 1e-6, old), flush=True)


if __name__ == '__main__':
    main()
