"""Times YR_OP_MBH in its forms - the LDS-tiled kernels' own choice (tile 254) against the register-chained whole-block kernel
(mbxr_h.hip: mbhr_kernel, tile 255, 0 .. 8 row segments) - on the front blocks of the 16-bit configurations.
    python tools/mbhr_probe.py [bf16|f16]"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from yoloret_amd import runtime as rt        # noqa: E402

SHAPES = {  # name: (batch, h, w, cin, cexp, cout, stride, residual)
    'lite0 stage2_block0': (128, 208, 208, 16, 96, 24, 2, False),
    'lite0 stage2_block1': (128, 104, 104, 24, 144, 24, 1, True),
    'lite0 stage4_block0': (128, 52, 52, 40, 240, 80, 2, False),
    'lite3 stage2_block0': (32, 320, 320, 24, 144, 32, 2, False),
    'lite3 stage2_block1': (32, 160, 160, 32, 192, 32, 1, True),
}


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    dev = torch.device('cuda:0')
    did = rt.dtype_id(dt)
    tdt = rt.TORCH_DTYPE[did]
    rng = np.random.default_rng(0)
    for name, (b, h, w, cin, cexp, cout, s, res) in SHAPES.items():
        cexp_p, kp, ldo = (cexp + 31) // 32 * 32, (cin + 31) // 32 * 32, (cout + 7) // 8 * 8
        x = torch.randn((b, h, w, cin), device=dev).to(tdt)
        wet = (torch.randn((cexp_p, kp), device=dev) * (2.0 / cin) ** 0.5).to(tdt)
        prm = torch.from_numpy(np.concatenate([rng.standard_normal((9, cexp_p)) * 0.47, np.ones((1, cexp_p)), np.zeros((1, cexp_p)),
                                               np.ones((1, cexp_p)), np.zeros((1, cexp_p))]).astype(np.float32)).to(dev)
        wpt = (torch.randn((cout, cexp_p), device=dev) * (1.0 / cexp) ** 0.5).to(tdt)
        pb = torch.cat([torch.ones(ldo, device=dev), torch.zeros(ldo, device=dev)])
        ho, wo = (h + s - 1) // s, (w + s - 1) // s
        out = torch.empty((b, ho, wo, ldo), dtype=tdt, device=dev)
        for tile in [(254, 0), (255, 0), (255, 1), (255, 2), (255, 4), (255, 8)]:
            op = rt.new_op(rt.OP_MBH, 'relu6')
            op.dtype = op.out_dtype = did
            op.h, op.w, op.cin, op.cout, op.stride, op.nsrc, op.se_reduced = ho, wo, cin, cout, s, 1, cexp
            op.k = 3 | tile[0] << 8 | tile[1] << 16
            op.src[0] = rt.make_src(x, c=cin)
            op.wgt, op.wgt2, op.b1, op.b2 = wet.data_ptr(), prm.data_ptr(), wpt.data_ptr(), pb.data_ptr()
            if res:
                op.res, op.res_ld = x.data_ptr(), cin
            op.out, op.out_ld = out.data_ptr(), ldo
            try:
                for _ in range(3):
                    rt.run_op(op, b)
            except rt.YoloretHipError as e:
                print('%-22s tile %s: %s' % (name, tile, str(e)[:60]))
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                rt.run_op(op, b)
            e1.record()
            torch.cuda.synchronize()
            print('%-22s %s tile %-9s %.4f ms' % (name, dt, tile, e0.elapsed_time(e1) / 20), flush=True)


if __name__ == '__main__':
    main()
