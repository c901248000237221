#!/usr/bin/env python
"""Writes a Keras-layout weights-only HDF5 file (what ``tf.keras.Model.save_weights('x.h5')`` produces; reference
code/train.py:74-91,182-186) from a parameter dict - the fixture generator for the HDF5 import tests.

Needs h5py, which only the build image's conda interpreter has (the product reads the format itself):

    /opt/conda/bin/python3.9 tools/make_keras_h5.py weights.npz layers.json out.h5 [--gap-every N] [--compress]

weights.npz : '<layer>/<param>' -> array (yoloret_amd.engine.Model.save_weights)
layers.json : [[layer name, keras class, explicitly_named], ...] in CREATION order.  Layers that are not explicitly
              named in the reference get Keras' automatic names - '<class>', '<class>_1', ... numbered per class in
              creation order; --gap-every N skips one number after every N-th such layer (the reference creates layers
              that never reach the saved model: discarded `y` convs, EfficientNet's top conv).  Layer groups are written
              in SORTED name order, not creation order, so a reader cannot lean on the file order.
              A row whose layer name is null is a GHOST: a layer of that class the reference creates at that point but
              which never reaches the saved model - it takes its number and nothing is written (the second backbone
              `backbone_transfer` built only to copy ImageNet weights from, code/yolo3/model.py:180-181,193-194,206-207,
              shifts every head layer's index by the backbone's layer count).
"""
import json
import sys

import h5py
import numpy as np


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    npz, layers_json, out = args[:3]
    gap = 0
    if '--gap-every' in sys.argv:
        gap = int(sys.argv[sys.argv.index('--gap-every') + 1])
        args = [a for a in args if a != str(gap)] if False else args
    compress = '--compress' in sys.argv
    z = np.load(npz)
    layers = json.load(open(layers_json))
    counters, names = {}, {}
    for lname, cls, explicit in layers:
        if explicit and lname is not None:
            names[lname] = lname
            continue
        i = counters.get(cls, 0)
        if lname is not None:
            names[lname] = cls if i == 0 else '%s_%d' % (cls, i)
        i += 1
        if gap and i % gap == 0:
            i += 1
        counters[cls] = i
    order = {'kernel': 0, 'bias': 1, 'depthwise_kernel': 0, 'gamma': 0, 'beta': 1, 'moving_mean': 2, 'moving_variance': 3, 'alpha': 0}
    with h5py.File(out, 'w') as f:
        f.attrs['backend'] = b'tensorflow'
        f.attrs['keras_version'] = b'2.4.0'
        f.attrs['layer_names'] = [n.encode('utf8') for n in sorted(names.values())]
        for lname, kname in sorted(names.items(), key=lambda kv: kv[1]):
            g = f.create_group(kname)
            params = sorted([k.split('/', 1)[1] for k in z.files if k.rsplit('/', 1)[0] == lname], key=lambda p: order[p])
            g.attrs['weight_names'] = [('%s/%s:0' % (kname, p)).encode('utf8') for p in params]
            for p in params:
                a = z['%s/%s' % (lname, p)]
                if p == 'depthwise_kernel':
                    a = a.reshape(a.shape + (1,))       # Keras: [kh, kw, C, 1]
                kw = dict(compression='gzip', compression_opts=9, shuffle=True, chunks=True) if compress and a.size > 64 else {}
                g.create_dataset('%s/%s:0' % (kname, p), data=a.astype(np.float32), **kw)
    print('wrote %s: %d layers (%d automatically named)' % (out, len(names), sum(1 for l in layers if not l[2])))


if __name__ == '__main__':
    main()
