"""Helpers adjacent to the detection path (reference code/yolo3/utils.py)."""
import os
from functools import reduce

import numpy as np

_MODEL_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'model_data')


def compose(*funcs):
    """Compose arbitrarily many functions, evaluated left to right (utils.py:56-64)."""
    if funcs:
        return reduce(lambda f, g: lambda *a, **kw: g(f(*a, **kw)), funcs)
    raise ValueError('Composition of empty sequence not supported.')


def _resolve(path):
    """Accepts the reference's relative ``model_data/...`` paths as well as real paths."""
    if os.path.exists(path):
        return path
    alt = os.path.join(_MODEL_DATA, os.path.basename(path))
    if os.path.exists(alt):
        return alt
    if os.path.exists(alt + '.txt'):  # yolo.py:174's default lacks the extension
        return alt + '.txt'
    raise FileNotFoundError(path)


def get_anchors(anchors_path):
    """float32 [9,2] (w,h) anchors (utils.py:100-104)."""
    with open(_resolve(anchors_path)) as f:
        anchors = f.readline()
    anchors = [float(x) for x in anchors.split(',')]
    return np.array(anchors, np.float32).reshape(-1, 2)


def get_classes(classes_path):
    """class names, one per line (utils.py:115-120)."""
    with open(_resolve(classes_path)) as f:
        class_names = f.readlines()
    return [c.strip() for c in class_names]
