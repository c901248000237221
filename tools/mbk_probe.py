"""Times YR_OP_MBR's weight-streaming form (csrc/mbk.hip) on the deep MobileNetV2 x0.75 @416 blocks at batch 64 against what the
round-5 plan ran for the same block (YR_OP_MBE + the projection as a pointwise op: profiles/r05_perop.txt).
    python tools/mbk_probe.py [block ...]"""
import os
import sys

import torch

sys.path.insert(0, '.')
from tests.test_gpu_mbr import make_block_k   # noqa: E402
from yoloret_amd import runtime as rt          # noqa: E402

BLOCKS = {   # name: (h, w, cin, cexp, cout, stride, residual, rows, nw), r05 time of the two launches in ms
    'block_11': ((26, 26, 72, 432, 72, 1, True, 2, 8), 0.0419 + 0.0341),
    'block_13': ((26, 26, 72, 432, 120, 2, False, 2, 8), 0.0366 + 0.0197),
    'block_7': ((26, 26, 48, 288, 48, 1, True, 2, 8), 0.0354),
    'block_10': ((26, 26, 48, 288, 72, 1, False, 2, 8), 0.0388),
    'block_14': ((13, 13, 120, 720, 120, 1, True, 1, 8), 0.0349 + 0.0279),
}


def timed(op, b, n=50):
    for _ in range(5):
        rt.run_op(op, b)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        rt.run_op(op, b)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


def main():
    dev = torch.device('cuda:0')
    for name in sys.argv[1:] or list(BLOCKS):
        shape, old = BLOCKS[name]
        h, w, cin, cexp, cout, s, res = shape[:7]
        for rows, nw in [BLOCKS[name][0][7:9]]:
          if os.environ.get('MBK_PROBE_CFG') and os.environ['MBK_PROBE_CFG'] != '%d,%d' % (rows, nw):
              continue
          shape = shape[:7] + (rows, nw)
          for b in [int(v) for v in os.environ.get('MBK_PROBE_BATCH', '64,1').split(',')]:
            macs = b * ((h * w * cin * cexp) + ((h + s - 1) // s) * ((w + s - 1) // s) * (9 * cexp + cexp * cout))
            op, out, params, keep = make_block_k(shape, dev, b=b, seed=1)
            ms = timed(op, b)
            print('%-9s rows %d nw %d batch %2d  %.4f ms  %6.1f TF  (r05, batch 64: %.4f ms in two launches)' % (name, rows, nw, b, ms, 2 * macs / ms * 1e-9, old), flush=True)
            del op, out, params, keep


if __name__ == '__main__':
    main()
