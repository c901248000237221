"""Per-phase shader-clock totals of the pixel-stationary pointwise kernel (pointwise_stream.hip built with -DPWT_TIMING:
python tools/relink.py pointwise_stream.hip -DPWT_TIMING): a POINTWISE op of the given shape alone, 64 images.
    python tools/pwt_timing.py H W K N [gated] [batch]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoloret_amd import runtime as rt, compiler
h, w, k, n = (int(v) for v in sys.argv[1:5])
gated = len(sys.argv) > 5 and sys.argv[5] == '1'
b = int(sys.argv[6]) if len(sys.argv) > 6 else 64
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
ld = (k + 3) & ~3
x = torch.randn((b, h, w, ld), device=dev)
wt = (rng.standard_normal((n, ld)) * np.sqrt(2.0 / k)).astype(np.float32)
wt[:, k:] = 0
nk = compiler.pwt_chunks(ld)
planes = torch.from_numpy(compiler.head_pack(wt, [ld], nk=nk)).to(dev)
sc = torch.ones(n, device=dev); sh = torch.zeros(n, device=dev)
out = torch.zeros((b, h, w, n), device=dev)
op = rt.new_op(rt.OP_POINTWISE, 'none')
op.h, op.w, op.cin, op.cout, op.nsrc = h, w, k, n, 1
op.src[0] = rt.make_src(x, c=k)
op.wgt, op.scale, op.shift = planes.data_ptr(), sc.data_ptr(), sh.data_ptr()
if gated:
    g = torch.rand((b, 1, 1, ld), device=dev)
    op.gate, op.gate_ld = g.data_ptr(), ld
op.out, op.out_ld = out.data_ptr(), n
op.se_reduced |= 0x40000
for _ in range(3):
    rt.run_op(op, b)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    rt.run_op(op, b)
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 50
mb = (b * h * w * (ld + n) * 4) / 1e6
print('%dx%d K %d -> N %d%s, %d images: %.1f us, %.1f MB -> %.2f TB/s' % (h, w, k, n, ' gated' if gated else '', b, us, mb, mb / us / 1e6 * 1e6 / 1e6))
lib = rt.lib()
if hasattr(lib, 'yr_pwt_dbg_read'):
    buf = (ctypes.c_uint * (256 * 16 * 8))()
    lib.yr_pwt_dbg_read(buf, 256 * 16 * 8)
    t = np.frombuffer(buf, dtype=np.uint32).reshape(256, 16, 8).astype(np.int64)
    act = t[:, :, 5] > 0
    names = ['start: DMA issue, first fetch issue', 'wait for the planes + barrier', 'wait for the pixels + cut', 'multiply + epilogue + stores', 'next fetch issue', 'tiles', 'drain']
    tot = t[:, :, [0, 1, 2, 3, 4, 6]].sum(axis=2)
    print('  waves with work %d of %d; cycles per wave: mean %.0f max %.0f' % (act.sum(), act.size, tot[act].mean(), tot[act].max()))
    for i, nm in enumerate(names):
        print('  %-38s mean %9.0f' % (nm, t[:, :, i][act].mean()))
