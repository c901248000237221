#!/usr/bin/env python
"""Times the fused 16-bit block kernels (YR_OP_MBH; YR_OP_MBX = expand + depthwise with squeeze sums, names x_*) on the
MobileNetV2 / EfficientNet block shapes of the bench workloads for a range of forced output tiles
(GPU; python tools/mbh_probe.py [batch] [filter])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoloret_amd import runtime as rt

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
FILT = sys.argv[2] if len(sys.argv) > 2 else ''
BLOCKS = [  # name, h(in), cin, cexp, cout, k, s, res
    ('block_1', 208, 16, 96, 24, 3, 2, 0), ('block_2', 104, 24, 144, 24, 3, 1, 1), ('block_3', 104, 24, 144, 32, 3, 2, 0),
    ('block_4', 52, 32, 192, 32, 3, 1, 1), ('block_6', 52, 32, 192, 48, 3, 2, 0), ('block_7', 26, 48, 288, 48, 3, 1, 1),
    ('block_10', 26, 48, 288, 72, 3, 1, 0), ('block_11', 26, 72, 432, 72, 3, 1, 1), ('block_13', 26, 72, 432, 120, 3, 2, 0),
    ('block_14', 13, 120, 720, 120, 3, 1, 1),
    # EfficientNet-lite0 @416 blocks (5x5 depthwise)
    ('l_s3b0', 104, 24, 144, 40, 5, 2, 0), ('l_s3b1', 52, 40, 240, 40, 5, 1, 1), ('l_s5b0', 26, 80, 480, 112, 5, 1, 0),
    ('l_s5b1', 26, 112, 672, 112, 5, 1, 1), ('l3_s2b0', 320, 24, 144, 32, 3, 2, 0), ('l3_s3b0', 160, 32, 192, 48, 5, 2, 0),
    # EfficientNet-B0 @416 squeeze-excite blocks: expand + depthwise (cout 0 = MBX, swish)
    ('x_s2b0', 208, 16, 96, 0, 3, 2, 0), ('x_s2b1', 104, 24, 144, 0, 3, 1, 0), ('x_s3b0', 104, 24, 144, 0, 5, 2, 0),
    ('x_s3b1', 52, 40, 240, 0, 5, 1, 0), ('x_s4b0', 52, 40, 240, 0, 3, 2, 0), ('x_s4b1', 26, 80, 480, 0, 3, 1, 0),
    ('x_s5b0', 26, 80, 480, 0, 5, 1, 0), ('x_s5b1', 26, 112, 672, 0, 5, 1, 0), ('x_s6b0', 26, 112, 672, 0, 5, 2, 0)]
TILES = [(4, 8), (8, 4), (8, 8), (7, 8), (4, 4), (6, 8), (4, 16), (8, 16), (16, 8), (13, 12), (13, 8), (7, 12), (13, 16), (16, 12), (13, 4), (4, 12), (2, 16), (7, 16), (5, 8), (16, 16), (8, 12), (7, 4)]
if os.environ.get('MBH_TILE'):
    TILES = [tuple(int(v) for v in os.environ['MBH_TILE'].split('x'))]
rng = np.random.default_rng(0)
dt = 'bf16'
did = rt.dtype_id(dt)
for name, h, cin, cexp, cout, k, s, res in BLOCKS:
    if FILT and FILT not in name:
        continue
    mbx = cout == 0
    if mbx:
        cout = cexp
    cexp_p, kp, ldo = (cexp + 31) // 32 * 32, (cin + 31) // 32 * 32, (cout + 7) // 8 * 8
    ho = -(-h // s)
    x = torch.randn((B, h, h, (cin + 7) // 8 * 8), device=dev).to(torch.bfloat16)
    wet = (torch.randn((cexp_p, kp), device=dev) * 0.1).to(torch.bfloat16)
    prm = torch.rand((k * k + 4, cexp_p), device=dev)
    wpt = (torch.randn((cout, cexp_p), device=dev) * 0.05).to(torch.bfloat16)
    pb = torch.rand((2, ldo), device=dev)
    out = torch.empty((B, ho, ho, ldo), dtype=torch.bfloat16, device=dev)
    op = rt.new_op(rt.OP_MBX if mbx else rt.OP_MBH, 'swish' if mbx else 'relu6')
    op.dtype = op.out_dtype = did
    op.h, op.w, op.cin, op.cout, op.stride, op.nsrc, op.se_reduced = ho, ho, cin, cout, s, 1, cexp
    if mbx:
        rows_cap = -(-ho // 4) * (-(-ho // 8) if ho * ho > 1000 else -(-ho // 4))
        part = torch.empty((B, rows_cap, ldo), dtype=torch.float32, device=dev)
        op.gate, op.gate_ld, op.se_reduced = part.data_ptr(), ldo, rows_cap
    op.src[0] = rt.make_src(x, c=cin)
    op.wgt, op.wgt2, op.b1, op.b2 = wet.data_ptr(), prm.data_ptr(), wpt.data_ptr(), pb.data_ptr()
    if res:
        op.res, op.res_ld = x.data_ptr(), x.shape[3]
    op.out, op.out_ld = out.data_ptr(), ldo
    rows = []
    for tile in ([None] if not os.environ.get('MBH_TILE') else []) + TILES:
        op.k = k | (((tile[0] << 8) | (tile[1] << 16)) if tile else 0)
        try:
            rt.run_op(op, B)
        except rt.YoloretHipError as e:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rt.run_op(op, B)
        e1.record()
        torch.cuda.synchronize()
        rows.append((e0.elapsed_time(e1) / 20, tile))
    auto = rows[0][0]
    if os.environ.get('MBH_TILE'):
        print(name, rows)
        continue
    rows.sort(key=lambda r: r[0])
    flops = 2.0 * B * (h * h * cin * cexp + ho * ho * cexp * (k * k + (0 if mbx else cout)))
    print('%-9s auto %.4f ms | best: %s   (%.1f TF at best)' % (name, auto, '  '.join('%sx%s %.4f' % (t[0], t[1], ms) if t else 'auto %.4f' % ms for ms, t in rows[:6]),
                                                             flops / rows[0][0] / 1e9))
