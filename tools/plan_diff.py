"""Which op of a plan first differs between two builds of the same model?  Runs the model twice - once as it is, once with the
environment switches given on the command line applied to the compiler - with arena reuse off (YOLORET_NO_ARENA_REUSE=1: every
intermediate keeps its memory), and compares every buffer both plans have, in the order the first plan writes them.
    python tools/plan_diff.py mobilenetv2x75 64 2 HEAD_WALK=False"""
import os
import sys

os.environ['YOLORET_NO_ARENA_REUSE'] = '1'
os.environ['YOLORET_AUTOTUNE'] = '0'
sys.path.insert(0, '.')
import numpy as np   # noqa: E402
import torch         # noqa: E402
from yoloret_amd import compiler, layers as L, runtime as rt, weights as W   # noqa: E402
from yoloret_amd.yolo3.model import yolov3_body   # noqa: E402


def run(name, size, b, sets):
    saved = {k: getattr(compiler, k) for k in sets}
    for k, v in sets.items():
        setattr(compiler, k, v)
    try:
        m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
    finally:
        for k, v in saved.items():
            setattr(compiler, k, v)
    m.set_weights(W.synthetic_weights(m, 1234, 'conditioned'))
    x = torch.from_numpy(W.synthetic_images(b, size, size)).cuda()
    ys = m(x)
    torch.cuda.synchronize()
    ws = m._workspace[0]
    bufs = {}
    for op in m.plan.ops:
        for bf in (op.out, op.gate_out):
            if bf is None or bf.external_slot >= 0 or bf.dtype != 0:
                continue
            raw = ws[bf.offset * b: bf.offset * b + bf.bytes * b].view(torch.float32).cpu().numpy().reshape(b, bf.h, bf.w, bf.ld)[..., :bf.c]
            bufs[bf.name] = (op.name, raw.copy())
    return bufs, [y.cpu().numpy() for y in ys]


def main():
    name, size, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    sets = {}
    for a in sys.argv[4:]:
        k, v = a.split('=')
        sets[k] = eval(v)
    A, ya = run(name, size, b, {})
    B, yb = run(name, size, b, sets)
    for k, (opn, va) in A.items():
        if k in B and B[k][1].shape == va.shape:
            d = np.abs(va - B[k][1]) / np.maximum(1.0, np.abs(va))
            print('%-28s %-22s max scaled diff %.3e%s' % (k, opn, float(np.nanmax(d)) if d.size else 0.0, '   <-- non-finite' if not np.isfinite(va).all() else ''))
    for i, (p, q) in enumerate(zip(ya, yb)):
        print('y%d max scaled diff %.3e' % (i + 1, float((np.abs(p - q) / np.maximum(1.0, np.abs(p))).max())))


if __name__ == '__main__':
    main()
