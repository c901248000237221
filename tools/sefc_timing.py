"""Per-phase shader-clock totals of se_fc_kernel (elementwise.hip built with -DSEFC_TIMING: python tools/relink.py elementwise.hip
-DSEFC_TIMING): the squeeze-excite FC pair of a head block alone, rows of partial sums in, 64 images.
    python tools/sefc_timing.py C R ROWS"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoloret_amd import runtime as rt
dev = torch.device('cuda:0')
c, r, rows = (int(v) for v in sys.argv[1:4])
b = 64
rng = np.random.default_rng(1)
r4 = (r + 3) & ~3
w1 = torch.from_numpy((rng.standard_normal((c, r4)) * np.sqrt(2.0 / c)).astype(np.float32)).to(dev)
b1 = torch.from_numpy(rng.normal(0, 0.1, r4).astype(np.float32)).to(dev)
w2 = torch.from_numpy((rng.standard_normal((r, c)) * np.sqrt(2.0 / r)).astype(np.float32)).to(dev)
b2 = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32)).to(dev)
sums = torch.from_numpy(rng.standard_normal((b, rows, 1, c)).astype(np.float32) * 50).to(dev)
gate = torch.zeros((b, 1, 1, c), dtype=torch.float32, device=dev)
op = rt.new_op(rt.OP_SE_FC)
op.h, op.w, op.cin, op.cout, op.nsrc, op.se_reduced, op.k = 1, 1, c, c, 1, r, 169
op.src[0] = rt.make_src(sums, c=c)
op.wgt, op.b1, op.wgt2, op.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
op.out, op.out_ld = gate.data_ptr(), c
for _ in range(3):
    rt.run_op(op, b)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    rt.run_op(op, b)
ev[1].record()
torch.cuda.synchronize()
print('C %d R %d rows %d: %.1f us per launch (back to back)' % (c, r, rows, ev[0].elapsed_time(ev[1]) * 50))
lib = rt.lib()
if hasattr(lib, 'yr_sefc_dbg_read'):
    buf = (ctypes.c_uint * (64 * 16 * 8))()
    lib.yr_sefc_dbg_read(buf, 64 * 16 * 8)
    t = np.frombuffer(buf, dtype=np.uint32).reshape(64, 16, 8).astype(np.int64)
    for i, nm in enumerate(['channel means of the rows', 'FC pair', 'drain']):
        print('  %-28s mean %8.0f  slowest wave %8.0f' % (nm, t[:, :, i].mean(), t[:, :, i].max()))
