"""Times YR_OP_MBR (csrc/mbr.hip) on the MobileNetV2 x0.75 @416 block shapes at batch 64 against the op(s) the shipped
plan runs for the same block (per-op table of bench: profiles/r03_perop.txt), sweeping waves per workgroup and segments.
    python tools/mbr_probe.py [block ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from tests.test_gpu_mbr import make_block   # noqa: E402
from yoloret_amd import runtime as rt        # noqa: E402

BLOCKS = {   # name: (h, w, cin, cexp, cout, stride, residual), r03 time of the shipped kernels in ms
    'block_1': ((208, 208, 16, 96, 24, 2, False), 0.2173),
    'block_2': ((104, 104, 24, 144, 24, 1, True), 0.2042),
    'block_3': ((104, 104, 24, 144, 24, 2, False), 0.1275),
    'block_4': ((52, 52, 24, 144, 24, 1, True), 0.0665),
    'block_6': ((52, 52, 24, 144, 48, 2, False), 0.0845),
    'block_7': ((26, 26, 48, 288, 48, 1, True), 0.0835),
    'block_10': ((26, 26, 48, 288, 72, 1, False), 0.0850),
}
BATCH = {}
SPLIT = os.environ.get('MBR_PROBE_SPLIT', '0') != '0'    # the split form (float16 planes on the 16-bit matrix pipe): nw 3 / 6 only
NWS = {'block_1': [2, 3], 'block_2': [3], 'block_3': [3], 'block_4': [3], 'block_6': [3], 'block_7': [6, 8], 'block_10': [6, 8]}


def timed(op, b, n=30):
    for _ in range(3):
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
    names = sys.argv[1:] or list(BLOCKS)
    for name in names:
        b = BATCH.get(name, 64)
        shape, old = BLOCKS[name]
        h, w, cin, cexp, cout, s, res = shape
        macs = b * ((h * w * cin * cexp) + ((h + s - 1) // s) * ((w + s - 1) // s) * (9 * cexp + cexp * cout))
        for nw in (([int(v) for v in os.environ['MBR_PROBE_NW'].split(',')] if os.environ.get('MBR_PROBE_NW') else [3 if shape[3] <= 144 else 6]) if SPLIT else NWS[name]):
            for segs in ([0] if os.environ.get('MBR_PROBE_ONE') else [0, 1, 2, 4, 8, 13, 26]):
                if segs > (h + s - 1) // s:
                    continue
                op, out, params, keep = make_block(shape + (nw, segs), dev, b=b, seed=1, split=SPLIT)
                ms = timed(op, b)
                print('%-9s nw %d segs %2d  %.4f ms  %6.1f TF  (shipped %.4f ms)' % (name, nw, segs, ms, 2 * macs / ms * 1e-9, old), flush=True)
                del op, out, params, keep


if __name__ == '__main__':
    main()
