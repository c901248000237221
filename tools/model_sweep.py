#!/usr/bin/env python
"""Full-resolution parity + speed sweep over the BASELINE.json model configs (GPU): logits vs the oracle's
torch-CPU implementation, which fused kernels the plan uses, and batch throughput of the forward pass."""
import collections, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import params, torch_ref
from yoloret_amd import layers as L, runtime as rt
from yoloret_amd.yolo3.model import yolov3_body

dev = torch.device('cuda', 0)
for name, size, bt in (('mobilenetv2x75', 416, 64), ('mobilenetv2x14', 512, 32), ('efficientnetb0', 416, 32),
                       ('efficientnetb0-lite', 416, 32), ('efficientnetb3', 640, 8)):
    m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
    P = params.ParamStore(1234, 'conditioned')
    ref_model = torch_ref.TorchReference(P, name, 3, 20)
    x = params.synthetic_images(2, size, size)
    ref = ref_model(x)
    m.set_weights(P.values)
    got = [y.cpu().numpy().reshape(r.shape) for y, r in zip(m(torch.from_numpy(x).to(dev)), ref)]
    err = max(float(np.max(np.abs(g - r)) / max(1.0, float(np.max(np.abs(r))))) for g, r in zip(got, ref))
    kinds = collections.Counter(rt.OP_NAMES[o.kind] for o in m.plan.ops)
    xb = torch.from_numpy(params.synthetic_images(bt, size, size)).to(dev)
    for _ in range(3):
        m(xb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m(xb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print('%-20s @%d  max scaled logit err %.2e  forward B=%d: %.2f ms = %.0f img/s  ops %s'
          % (name, size, err, bt, dt * 1e3, bt / dt, dict(kinds)))
