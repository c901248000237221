"""Which op first differs when three steps are in flight?  Arena reuse off (every intermediate keeps its memory); context 1 runs
alone, its workspace is kept, then contexts 1-3 run at once on three streams and context 1's buffers are compared in plan order.
    gpurun -- python tools/inflight_diff.py"""
import os, sys
os.environ['YOLORET_NO_ARENA_REUSE'] = os.environ.get('YOLORET_NO_ARENA_REUSE', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoloret_amd import layers as L
from yoloret_amd.weights import synthetic_weights, synthetic_images
from yoloret_amd.yolo3.model import yolov3_body

dev = torch.device('cuda:0')
size, b = 416, int(os.environ.get('B', 64))
m = yolov3_body(L.Input(shape=[size, size, 3]), os.environ.get('BACKBONE', 'mobilenetv2x75'), 3, num_classes=20)
m.set_weights(synthetic_weights(m, 1234, 'survey'))
xs = [torch.from_numpy(synthetic_images(b, size, size, seed=s)).to(dev) for s in (21, 22, 23)]
if os.environ.get('SAME_CTX'):      # every context the same batch: a leak between contexts would not show
    xs = [xs[0], xs[0].clone(), xs[0].clone()]
if os.environ.get('SAME_IMG'):      # every image of a batch the same: a leak between images would not show
    xs = [x[:1].repeat(b, 1, 1, 1).contiguous() for x in xs]
for i in range(3):
    m(xs[i], ctx=i + 1)
torch.cuda.synchronize()


def snap():
    ws = m._workspace[(0, 1)]
    out = {}
    for op in m.plan.ops:
        for bf in (op.out, op.gate_out, op.gate if op.kind in (3, 15) else None):   # (3, 15: DEPTHWISE, HEAD - gate = the sums they write)
            if bf is None or bf.external_slot >= 0 or bf.name in out:
                continue
            out[bf.name] = (op.name, ws[bf.offset * b: bf.offset * b + bf.bytes * b].cpu().numpy().copy())
    return out


m(xs[0], ctx=1)
torch.cuda.synchronize()
ref = snap()
streams = [torch.cuda.Stream(dev) for _ in range(3)]
for rnd in range(int(os.environ.get('ROUNDS', 8))):
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            m(xs[i], ctx=i + 1)
    torch.cuda.synchronize()
    got = snap()
    bad = [(k, ref[k][0], int((got[k][1] != ref[k][1]).sum()), ref[k][1].size) for k in ref if not np.array_equal(got[k][1], ref[k][1])]
    print('round %d: %d buffers differ; first: %s' % (rnd, len(bad), bad[:4]))
    if bad and os.environ.get('DETAIL', '1') != '0':
        k = bad[0][0]
        a = ref[k][1].view(np.float32).reshape(b, -1)
        g = got[k][1].view(np.float32).reshape(b, -1)
        imgs = [i for i in range(b) if not np.array_equal(a[i], g[i])]
        print('   %s: images %s' % (k, imgs))
        i = imgs[0]
        idx = np.nonzero(a[i] != g[i])[0]
        print('   image %d: %d of %d floats differ, first at %s; ref %s got %s' % (i, idx.size, a.shape[1], idx[:8], a[i][idx[:6]], g[i][idx[:6]]))
        # is the wrong gate some OTHER image's gate?
        for j in range(b):
            if j != i and np.array_equal(g[i], a[j]):
                print('   == the serial gate of image %d' % j)
