"""Times YR_OP_MBE (csrc/mbr.hip: mbe_kernel) on the deep MobileNetV2 blocks at batch 64 over row segments per strip.
    python tools/mbe_probe.py"""
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, '.')
from yoloret_amd import runtime as rt        # noqa: E402
from yoloret_amd.compiler import mbr_pack    # noqa: E402

BLOCKS = {'block_11': (26, 26, 72, 432, 1, 0.085), 'block_13': (26, 26, 72, 432, 2, 0.067), 'block_14': (13, 13, 120, 720, 1, 0.0525),
          'x14_block_7': (32, 32, 88, 528, 1, 0.0), 'x14_block_11': (32, 32, 136, 816, 1, 0.0), 'x14_block_14': (16, 16, 224, 1344, 1, 0.0)}


def main():
    dev = torch.device('cuda:0')
    b = 64
    for name in (sys.argv[1:] or list(BLOCKS)):
        h, w, cin, cexp, s, old = BLOCKS[name]
        rng = np.random.default_rng(1)
        x = torch.from_numpy(rng.standard_normal((b, h, w, cin)).astype(np.float32)).to(dev)
        we = (rng.standard_normal((cin, cexp)) * np.sqrt(2.0 / cin)).astype(np.float32)
        wd = (rng.standard_normal((3, 3, cexp)) * np.sqrt(2.0 / 9)).astype(np.float32)
        one, zero = np.ones(cexp, np.float32), np.zeros(cexp, np.float32)
        wa, tab, _ = mbr_pack(we.T, one, zero, wd.reshape(9, cexp), one, zero, None, None, None)
        keep = [torch.from_numpy(np.ascontiguousarray(a).ravel()).to(dev) for a in (wa, tab)]
        ho, wo = (h + s - 1) // s, (w + s - 1) // s
        out = torch.empty((b, ho, wo, cexp), dtype=torch.float32, device=dev)
        macs = b * (h * w * cin * cexp + ho * wo * 9 * cexp)
        for segs in (0, 1, 2, 3, 4, 6):
            if segs > ho:
                continue
            op = rt.new_op(rt.OP_MBE, 'relu6')
            op.dtype = op.out_dtype = 0
            op.h, op.w, op.cin, op.cout, op.k, op.stride, op.nsrc = ho, wo, cin, cexp, 3 | segs << 16, s, 1
            op.src[0] = rt.make_src(x, c=cin)
            op.wgt, op.wgt2 = [k.data_ptr() for k in keep]
            op.out, op.out_ld = out.data_ptr(), cexp
            for _ in range(3):
                rt.run_op(op, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                rt.run_op(op, b)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            print('%-13s segs %d  %.4f ms  %5.1f TF  (unfused expand + depthwise: %.4f ms)' % (name, segs, ms, 2 * macs / ms * 1e-9, old), flush=True)


if __name__ == '__main__':
    main()
