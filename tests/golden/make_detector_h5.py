#!/usr/bin/env python
"""Writes tests/golden/detector_mbv2x75_q.h5: a FULL detector (MobileNetV2 x0.75 + RFCR + the three heads, 20 classes) in the Keras
weights-only HDF5 layout the reference restores from (code/yolo.py:87, train.py:182-186), small enough to commit - so that
tests/test_gpu_yolo.py::test_yolo_facade_loads_keras_h5 runs on a box without h5py (the product reads HDF5 itself; only WRITING the
fixture needs h5py, which the build image's conda interpreter has).

    python tests/golden/make_detector_h5.py

The parameters are `quantized_weights(model, 7)`: the 'synthetic:7' recipe rounded to a grid of one of each tensor's standard
deviation - 1.9 M float32 values of ~2 bits of entropy each, which HDF5's shuffle + gzip filters store in ~1.5 MB instead of
7.5 MB.  The test recomputes them with the same function and compares the detections of the two detectors."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CONDA = '/opt/conda/bin/python3.9'
OUT = os.path.join(ROOT, 'tests', 'golden', 'detector_mbv2x75_q.h5')
GAP = 7      # auto-numbered layers skipped (make_keras_h5.py --gap-every): the reference's discarded layers shift the numbering


def quantized_weights(model, seed):
    """{name: float32 array}: synthetic_weights(model, seed, 'survey') on a coarse per-tensor grid (moving variances stay positive)."""
    from yoloret_amd.weights import synthetic_weights
    out = {}
    for k, v in synthetic_weights(model, seed, 'survey').items():
        v = np.asarray(v, np.float32)
        step = np.float32(2.0 ** np.round(np.log2(max(float(v.std()), 1e-6) / 1.0)))
        q = (np.round(v / step) * step).astype(np.float32)
        if k.endswith('moving_variance'):
            q = np.maximum(q, step)
        out[k] = q
    return out


def main():
    from tests.test_h5 import MBV2_NAMED, _layers_json
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    L.reset_names()
    m = yolov3_body(L.Input(shape=[96, 96, 3]), 'mobilenetv2x75', 3, num_classes=20)
    w = quantized_weights(m, 7)
    with tempfile.TemporaryDirectory() as t:
        np.savez(os.path.join(t, 'w.npz'), **w)
        _layers_json(m, os.path.join(t, 'layers.json'), MBV2_NAMED)
        subprocess.check_call([CONDA, os.path.join(ROOT, 'tools', 'make_keras_h5.py'), os.path.join(t, 'w.npz'), os.path.join(t, 'layers.json'),
                               OUT, '--gap-every', str(GAP), '--compress'])
    print(OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
