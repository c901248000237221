"""YR_OP_MBR in its float32-MFMA form against its SPLIT form (two float16 planes per operand on the 16-bit matrix pipe, mbr.hip SP) on
the MobileNetV2 x0.75 @416 block shapes at batch 64: time per launch, the largest difference between the two, and on a small map
the error of both against a float64 composition of the three layers.
    python tools/mbs_probe.py [block ...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from tests.test_gpu_mbr import make_block   # noqa: E402
from tools.mbr_probe import BLOCKS, timed    # noqa: E402
from yoloret_amd import runtime as rt        # noqa: E402
from oracle import nn                        # noqa: E402

NW = {'block_1': 3, 'block_2': 3, 'block_3': 3, 'block_4': 3, 'block_6': 3, 'block_7': 6, 'block_10': 6}


def ref64(x, we, se, he, wd, sd, hd, wp, sp, hp, s, residual):
    f = np.float64
    t = np.clip(np.einsum('bhwc,cd->bhwd', x.astype(f), we.astype(f)) * se + he, 0, 6)
    t = np.clip(nn.depthwise(t, wd.astype(f), s, 'same') * sd + hd, 0, 6)
    r = np.einsum('bhwc,cd->bhwd', t, wp.astype(f)) * sp + hp
    return r + x if residual else r


def main():
    dev = torch.device('cuda:0')
    for name in sys.argv[1:] or list(BLOCKS):
        shape, _ = BLOCKS[name]
        h, w, cin, cexp, cout, s, res = shape
        macs = 64 * ((h * w * cin * cexp) + ((h + s - 1) // s) * ((w + s - 1) // s) * (9 * cexp + cexp * cout))
        row = []
        outs = []
        for split in (False, True):
            op, out, params, keep = make_block(shape + (NW[name], 0), dev, b=64, seed=1, split=split)
            ms = timed(op, 64)
            outs.append(out.clone())
            row.append('%s %.4f ms %6.1f TF' % ('split' if split else 'fp32 ', ms, 2 * macs / ms * 1e-9))
            del op, out, keep
        diff = (outs[0] - outs[1]).abs().max().item()
        small = (24, 28, cin, cexp, cout, s, res, NW[name], 0)
        errs = []
        for split in (False, True):
            op, out, params, keep = make_block(small, dev, b=2, seed=2, split=split)
            rt.run_op(op, 2)
            torch.cuda.synchronize()
            r = ref64(*params)
            errs.append(np.abs(out.cpu().numpy().astype(np.float64) - r).max() / max(1.0, np.abs(r).max()))
        print('%-9s %s | %s | max |fp32 - split| %.2e | error vs float64: fp32 %.2e split %.2e' % (name, row[0], row[1], diff, errs[0], errs[1]), flush=True)


if __name__ == '__main__':
    main()
