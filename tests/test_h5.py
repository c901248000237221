"""Keras weights-only HDF5 import (reference code/yolo.py:87 ``load_weights``; writer side code/train.py:74-91):
the product's own HDF5 reader (yoloret_amd/h5lite.py) and the Keras-name mapping (yoloret_amd/keras_h5.py).

Fixtures: tests/golden/keras_toy.h5 was written by h5py (tools/make_keras_h5.py under the build image's conda
interpreter) from tests/golden/keras_toy.npz; when that interpreter is present the full-size detector checkpoints are
generated on the fly as well."""
import json
import os
import subprocess

import numpy as np
import pytest

from yoloret_amd import h5lite, keras_h5

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA = '/opt/conda/bin/python3.9'
MBV2_NAMED = ('Conv1', 'bn_Conv1', 'expanded_conv_', 'block_')   # tf.keras.applications.MobileNetV2 + model.py:243-270


def _has_h5py():
    try:
        return os.path.exists(CONDA) and subprocess.call([CONDA, '-c', 'import h5py'], stderr=subprocess.DEVNULL) == 0
    except OSError:
        return False


def _toy_model():
    """A few layers of every kind on the detection path, small enough to commit as a fixture."""
    from yoloret_amd import layers as L
    from yoloret_amd.engine import Model
    L.reset_names()
    x = L.Input(shape=[32, 32, 3])
    t = L.Conv2D(8, 3, strides=2, padding='same', use_bias=False, name='Conv1')(x)
    t = L.ReLU(6., name='r0')(L.BatchNormalization(name='bn_Conv1')(t))
    t = L.ReLU(6., name='r1')(L.BatchNormalization(name='a_bn')(L.DepthwiseConv2D(3, padding='same', use_bias=False, name='a_dw')(t)))
    a = L.Conv2D(12, 1, padding='same', use_bias=False, name='a_pw')(t)
    b = L.Conv2D(12, 1, padding='same', use_bias=True, name='b_pw')(t)
    c = L.Conv2D(12, 1, padding='same', use_bias=False, name='c_pw')(t)
    d = L.Conv2D(12, 1, padding='same', use_bias=False, name='d_pw')(t)
    s = L.WeightedSum(name='ws')([a, b, c, d])
    s = L.ReLU(6., name='r2')(L.BatchNormalization(name='b_bn')(L.DepthwiseConv2D(5, padding='same', use_bias=False, name='b_dw')(s)))
    ys = [L.Conv2D(6, 1, padding='same', use_bias=False, name='y%d' % i)(s) for i in range(3)]
    return Model(x, ys)


def _layers_json(model, path, named_prefixes=()):
    rows = [[n, cls, n.startswith(tuple(named_prefixes)) if named_prefixes else False] for n, cls, _ in keras_h5.model_layers(model)]
    json.dump(rows, open(path, 'w'))
    return rows


def test_reader_on_committed_fixture():
    """HDF5 structures h5py's defaults produce - symbol-table groups, contiguous and chunked+shuffle+gzip datasets,
    fixed-length string attributes - byte for byte against the arrays the file was written from."""
    z = np.load(os.path.join(G, 'keras_toy.npz'))
    m = _toy_model()
    got = keras_h5.load_keras_h5(m, os.path.join(G, 'keras_toy.h5'))
    assert set(got) == set(z.files) == set(m.param_shapes)
    for k in z.files:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], z[k]), k
    f = h5lite.File(os.path.join(G, 'keras_toy.h5'))
    assert f.attrs['backend'] == b'tensorflow'
    names = [n.decode() for n in f.attrs['layer_names']]
    assert 'Conv1' in names and 'conv2d' in names and 'conv2d_1' in names and 'weighted_sum' in names
    assert 'conv2d_3' not in names          # the generator left a gap in the numbering (--gap-every 3)
    raw = h5lite.read_keras_weights(os.path.join(G, 'keras_toy.h5'))
    assert raw['depthwise_conv2d']['depthwise_kernel'].shape == (3, 3, 8, 1)        # Keras layout on disk


def test_mapping_errors():
    m = _toy_model()
    raw = h5lite.read_keras_weights(os.path.join(G, 'keras_toy.h5'))
    broken = dict(raw)
    del broken['conv2d_1']
    with pytest.raises(ValueError, match='automatically named conv2d'):
        keras_h5.map_keras_layers(m, broken)
    broken = {k: dict(v) for k, v in raw.items()}
    broken['conv2d']['kernel'] = broken['conv2d']['kernel'][..., :5]
    with pytest.raises(ValueError, match='shape'):
        keras_h5.map_keras_layers(m, broken)
    with pytest.raises(h5lite.H5Error, match='not an HDF5 file'):
        h5lite.File(b'definitely not hdf5' * 10)
    data = open(os.path.join(G, 'keras_toy.h5'), 'rb').read()
    with pytest.raises(h5lite.H5Error):
        h5lite.read_keras_weights(data[:len(data) // 2])


@pytest.mark.skipif(not _has_h5py(), reason='needs the build image\'s conda interpreter with h5py to WRITE the checkpoints')
@pytest.mark.parametrize('name,size', [('mobilenetv2x75', 416), ('efficientnetb3', 320)])
def test_full_detector_checkpoint_round_trip(tmp_path, name, size):
    """A Keras-layout checkpoint of the whole detector (automatic layer names with gaps, groups in sorted order,
    gzip-compressed datasets) -> Model.load_weights('x.h5') == the parameters it was written from, bit for bit; and
    the packed device blob equals the one built from the .npz path."""
    from yoloret_amd import layers as L, weights as W
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=20)
    wd = W.synthetic_weights(m, 11, 'survey')
    np.savez(tmp_path / 'w.npz', **wd)
    _layers_json(m, tmp_path / 'layers.json', MBV2_NAMED if name.startswith('mobilenet') else ('block_2',))
    subprocess.check_call([CONDA, os.path.join(ROOT, 'tools', 'make_keras_h5.py'), str(tmp_path / 'w.npz'),
                           str(tmp_path / 'layers.json'), str(tmp_path / 'ckpt.h5'), '--gap-every', '9', '--compress'])
    m.load_weights(str(tmp_path / 'ckpt.h5'))
    got = m.get_weights()
    assert set(got) == set(wd)
    for k in wd:
        assert np.array_equal(got[k], wd[k]), k
    blob_h5 = m.plan.build_blob(got)
    m.load_weights(str(tmp_path / 'w.npz'))
    assert np.array_equal(blob_h5, m.plan.build_blob(m.get_weights()))


@pytest.mark.skipif(not _has_h5py(), reason='needs the build image\'s conda interpreter with h5py to WRITE the checkpoint')
def test_checkpoint_with_the_reference_construction_order(tmp_path):
    """The reference builds the backbone TWICE - `backbone` and `backbone_transfer`, the second only to copy ImageNet
    weights from (code/yolo3/model.py:180-181,193-194,206-207) - before RFCR and the heads: in a real checkpoint the
    backbone's automatically named layers are conv2d .. conv2d_<N-1>, the second copy takes N .. 2N-1 and is never saved,
    and the first RFCR conv is conv2d_<2N>.  This fixture numbers the layers exactly so (every parameter layer the graph
    builder creates before `rfcr_b1c` is repeated as a ghost behind the backbone); the mapping only relies on the ORDER of
    the indices per class, so it must restore every parameter bit for bit."""
    from yoloret_amd import layers as L, weights as W
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[128, 128, 3]), 'efficientnetb3', 3, num_classes=20)
    wd = W.synthetic_weights(m, 23, 'survey')
    np.savez(tmp_path / 'w.npz', **wd)
    rows = [[n, cls, False] for n, cls, _ in keras_h5.model_layers(m)]
    first_head = [i for i, r in enumerate(rows) if r[0].startswith('rfcr_')][0]
    ghosts = [[None, cls, False] for _, cls, _ in rows[:first_head]]
    rows = rows[:first_head] + ghosts + rows[first_head:]
    json.dump(rows, open(tmp_path / 'layers.json', 'w'))
    subprocess.check_call([CONDA, os.path.join(ROOT, 'tools', 'make_keras_h5.py'), str(tmp_path / 'w.npz'),
                           str(tmp_path / 'layers.json'), str(tmp_path / 'ckpt.h5')])
    raw = h5lite.read_keras_weights(str(tmp_path / 'ckpt.h5'))
    n_backbone_convs = sum(1 for r in rows[:first_head] if r[1] == 'conv2d')
    assert 'conv2d_%d' % (n_backbone_convs - 1) in raw and 'conv2d_%d' % n_backbone_convs not in raw      # the transfer copy's numbers are absent
    assert 'conv2d_%d' % (2 * n_backbone_convs) in raw                                                     # the first RFCR conv
    m.load_weights(str(tmp_path / 'ckpt.h5'))
    got = m.get_weights()
    assert set(got) == set(wd)
    for k in wd:
        assert np.array_equal(got[k], wd[k]), k


def test_reader_refuses_cyclic_headers_and_reads_chunked_name_lists():
    """(i) an object-header continuation message that points back at its own block must end in H5Error, not in an endless
    loop; (ii) Keras splits name lists beyond HDF5's 64 KB attribute limit into layer_names0, layer_names1, ...
    (save_attributes_to_hdf5_group): the reader concatenates them."""
    data = bytearray(open(os.path.join(G, 'keras_toy.h5'), 'rb').read())
    f = h5lite.File(bytes(data))
    # (ii) through the mapping helper: a root whose attributes carry the chunked form
    names = [n for n in f.attrs['layer_names']]
    half = len(names) // 2
    f.attrs.pop('layer_names')
    f.attrs['layer_names0'], f.attrs['layer_names1'] = names[:half], names[half:]
    orig = h5lite.File
    try:
        h5lite.File = lambda *_a, **_k: f
        raw = h5lite.read_keras_weights(b'ignored')
    finally:
        h5lite.File = orig
    assert len(raw) == len(names)
    # (i) version-1 object header of the root group: turn its first message into a continuation pointing at itself
    root_hdr = f._superblock()
    assert f._u(root_hdr, 1) == 1                                       # h5py's default: version-1 object headers
    base = f._base + root_hdr
    blk = root_hdr + 16
    data[base + 16:base + 18] = (0x10).to_bytes(2, 'little')            # message type: continuation
    data[base + 18:base + 20] = (16).to_bytes(2, 'little')              # 16-byte body: address, length
    data[base + 24:base + 32] = blk.to_bytes(8, 'little')
    data[base + 32:base + 40] = (64).to_bytes(8, 'little')
    with pytest.raises(h5lite.H5Error, match='cycle'):
        h5lite.read_keras_weights(bytes(data))


def test_committed_full_detector_checkpoint_reads_back():
    """tests/golden/detector_mbv2x75_q.h5 (tests/golden/make_detector_h5.py: the whole MobileNetV2 x0.75 detector, 164 layer groups,
    chunked + shuffled + gzip'd datasets) read by the product's own HDF5 reader == the parameters it was written from."""
    from tests.golden.make_detector_h5 import OUT, quantized_weights
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    L.reset_names()
    m = yolov3_body(L.Input(shape=[96, 96, 3]), 'mobilenetv2x75', 3, num_classes=20)
    got = keras_h5.load_keras_h5(m, OUT)
    want = quantized_weights(m, 7)
    assert set(got) == set(want) and all(np.array_equal(got[k], want[k]) for k in want)
