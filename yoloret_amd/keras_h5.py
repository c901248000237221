"""Keras weights-only HDF5 checkpoints -> the parameter dict of a ``yoloret_amd.engine.Model``.

The reference restores its detector with ``self.model.load_weights(self.model_path)`` (code/yolo.py:87) from files
written by ``ModelCheckpoint(save_weights_only=True)`` / ``model.save_weights`` (code/train.py:74-91,182-186).  Such a
file lists the model's layers by their KERAS names.  Most layers of the reference's graph are created without a name
(code/yolo3/model.py:20-30,98-114,152-155,243-251; code/yolo3/efficientnet.py:419-434,485-527,636-645), so Keras
names them ``conv2d``, ``conv2d_1``, ``batch_normalization_7`` ... - numbered per class in CREATION order, with gaps
where a created layer is not part of the saved model (the discarded top-down ``y`` convs, EfficientNet's unused top
conv).  The build's graph builder gives every layer a descriptive name instead (``rfcr_b1c``, ``td2_mb_project`` ...)
but creates them in the same order, call for call.  The mapping therefore is:

  1. a file layer whose name equals one of the model's layer names is taken as is (tf.keras.applications.MobileNetV2
     names all its layers: ``Conv1``, ``bn_Conv1``, ``block_3_expand`` ...);
  2. the remaining file layers are grouped by class (name without the ``_<n>`` suffix), sorted by ``<n>`` and matched
     one to one, in order, with the model's remaining layers of that class in creation order; every shape is checked.

Layouts are Keras' (SURVEY.md A.5): Conv2D kernel [kh,kw,Cin,Cout] (+ bias [Cout]); DepthwiseConv2D
depthwise_kernel [kh,kw,C,1]; BatchNormalization gamma, beta, moving_mean, moving_variance [C]; WeightedSum alpha [4].
"""
import re

import numpy as np

from .h5lite import read_keras_weights

_CLASS_OF = (('depthwise_kernel', 'depthwise_conv2d'), ('kernel', 'conv2d'), ('gamma', 'batch_normalization'),
             ('alpha', 'weighted_sum'))


def _class_of(param_names):
    for key, cls in _CLASS_OF:
        if key in param_names:
            return cls
    raise ValueError('layer with parameters %s has no Keras class on the detection path' % sorted(param_names))


def model_layers(model):
    """[(layer name, Keras class, {param: shape})] of `model` in creation order."""
    layers = {}
    for full, shape in model.param_shapes.items():
        lname, pname = full.rsplit('/', 1)
        layers.setdefault(lname, {})[pname] = tuple(shape)
    seq = model.plan.layer_seq
    return [(n, _class_of(p), p) for n, p in sorted(layers.items(), key=lambda kv: seq[kv[0]])]


def map_keras_layers(model, file_layers):
    """file_layers: {keras layer name: {param: array}} -> {model parameter name: float32 array} (every parameter of
    the model, nothing else); raises ValueError describing the first mismatch."""
    mine = model_layers(model)
    by_name = {n: (cls, p) for n, cls, p in mine}
    file_layers = {k: v for k, v in file_layers.items() if v}        # layers without weights (ReLU, Add ...) carry none
    pairs = []
    rest_file, used = {}, set()
    for kname, wd in file_layers.items():
        if kname in by_name:
            pairs.append((kname, kname, wd))
            used.add(kname)
            continue
        m = re.match(r'^(.*?)(?:_(\d+))?$', kname)
        rest_file.setdefault(m.group(1), []).append((int(m.group(2) or 0), kname, wd))
    rest_mine = {}
    for n, cls, p in mine:
        if n not in used:
            rest_mine.setdefault(cls, []).append(n)
    for cls in sorted(set(rest_file) | set(rest_mine)):
        f_list = sorted(rest_file.get(cls, []))
        m_list = rest_mine.get(cls, [])
        if len(f_list) != len(m_list):
            raise ValueError('the file holds %d automatically named %s layers, the model has %d (a checkpoint of another '
                             'architecture / input configuration?)' % (len(f_list), cls, len(m_list)))
        pairs += [(mn, kname, wd) for mn, (_, kname, wd) in zip(m_list, f_list)]
    out = {}
    for mn, kname, wd in pairs:
        want = by_name[mn][1]
        if set(wd) != set(want):
            raise ValueError('layer %s (file: %s): parameters %s, expected %s' % (mn, kname, sorted(wd), sorted(want)))
        for pname, shape in want.items():
            a = np.asarray(wd[pname], np.float32)
            if pname == 'depthwise_kernel' and a.ndim == 4 and a.shape[-1] == 1:
                a = a[..., 0]
            if tuple(a.shape) != shape:
                raise ValueError('layer %s (file: %s): %s has shape %s, expected %s' % (mn, kname, pname, a.shape, shape))
            out['%s/%s' % (mn, pname)] = a
    missing = [k for k in model.param_shapes if k not in out]
    if missing:
        raise ValueError('the file has no weights for %s%s' % (', '.join(missing[:6]), ' ...' if len(missing) > 6 else ''))
    return out


def load_keras_h5(model, path):
    return map_keras_layers(model, read_keras_weights(path))
