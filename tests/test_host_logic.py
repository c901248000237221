"""Host-side logic that needs no GPU: the graph builder / compiler against SURVEY.md's tables,
the drop-in surface's argument handling, weight recipes, and that the C-ABI library loads and
exports every symbol include/yoloret_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import model as om
from oracle import params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(name='mobilenetv2x75', size=416, classes=20, **kw):
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3.model import yolov3_body
    return yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=classes, **kw)


@pytest.mark.parametrize('name,size,macs_m', [('mobilenetv2x75', 416, 1069.3), ('mobilenetv2x14', 512, 3325.7),
                                              ('efficientnetb0', 416, 1626.1), ('efficientnetb3', 640, 7679.7)])
def test_macs_match_survey(name, size, macs_m):
    assert abs(_model(name, size).plan.total_macs() / 1e6 - macs_m) < 0.06


def test_plan_structure_and_accounting():
    from yoloret_amd import compiler, runtime as rt
    knobs = ('FUSE_MAX_CIN', 'FUSE_STEM', 'HOIST_UPSAMPLE', 'POOL_IN_PRODUCER', 'MERGE_SE_MEAN', 'FOLD_DW', 'FUSE_MBR', 'FOLD_PROJ', 'FUSE_MBE', 'FOLD_WSUM', 'PW_STREAM_PAIRS')
    saved = [getattr(compiler, k) for k in knobs]
    try:
        for k, v in zip(knobs, (0, False, False, False, False, False, False, False, False, False, False)):   # every rewrite off: the plan = SURVEY.md Appendix B rows
            setattr(compiler, k, v)
        p = _model().plan
    finally:
        for k, v in zip(knobs, saved):
            setattr(compiler, k, v)
    kinds = [o.kind for o in p.ops]
    assert kinds.count(rt.OP_POINTWISE) == 55 and kinds.count(rt.OP_DEPTHWISE) == 23 and kinds.count(rt.OP_STEM) == 1
    assert kinds.count(rt.OP_SE_MEAN) == 6 and kinds.count(rt.OP_SE_FC) == 6 and kinds.count(rt.OP_WSUM) == 1
    assert kinds.count(rt.OP_GATHER) == 0   # upsample / maxpool / concat never materialised
    assert abs(p.algorithmic_bytes_per_image() / 1e6 - 198.9) < 0.2
    widths = {o.name: o.cin for o in p.ops}
    assert widths['td1_conv'] == 216 and widths['td2_conv'] == 424 and widths['td3_conv'] == 248
    assert widths['bu2_conv'] == 203 and widths['bu1_conv'] == 331
    assert [(b.h, b.w, b.c) for b in p.output_bufs] == [(13, 13, 75), (26, 26, 75), (52, 52, 75)]
    fm = _model()
    fused = fm.plan
    from yoloret_amd.weights import synthetic_weights
    assert np.isfinite(fused.build_blob(synthetic_weights(fm, 1, 'survey'))).all()   # every fused packing runs
    nlane, nmbr = (sum(o.kind == k for o in fused.ops) for k in (rt.OP_MBLANE, rt.OP_MBR))
    assert fused.ops[0].kind == rt.OP_STEMBLOCK and nlane + nmbr >= 6 and nmbr >= 4   # block_1..6 fused as before; the 26 x 26 blocks on mbr.hip
    assert abs(fused.algorithmic_bytes_per_image() - p.algorithmic_bytes_per_image()) < 1  # accounting is fusion-invariant
    assert fused.total_macs() == p.total_macs()
    assert fused.arena_bytes_per_image < p.arena_bytes_per_image


def test_fold_depthwise_plan_keeps_the_accounting():
    from yoloret_amd import compiler, runtime as rt
    from yoloret_amd.weights import synthetic_weights
    saved = compiler.FOLD_DW, compiler.FUSE_MBR
    compiler.FOLD_DW, compiler.FUSE_MBR = True, False   # (an alternative to fusing the same blocks into one kernel)
    saved_mbe, compiler.FUSE_MBE = compiler.FUSE_MBE, False
    try:
        compiler.FOLD_DW = saved[0]
        base = _model().plan
        compiler.FOLD_DW = True
        fm = _model()
    finally:
        compiler.FOLD_DW, compiler.FUSE_MBR = saved
        compiler.FUSE_MBE = saved_mbe
    p = fm.plan
    folded = [o for o in p.ops if o.kind == rt.OP_POINTWISE and o.srcs[0].xform == 'dw3']
    assert [o.name for o in folded] == ['block_%d_project' % i for i in range(7, 16)]   # block_16: cout 240 > one tile
    assert sum(o.kind == rt.OP_DEPTHWISE for o in p.ops) == sum(o.kind == rt.OP_DEPTHWISE for o in base.ops) - 9
    assert p.total_macs() == base.total_macs()
    assert abs(p.algorithmic_bytes_per_image() - base.algorithmic_bytes_per_image()) < 1
    assert p.arena_bytes_per_image <= base.arena_bytes_per_image
    assert np.isfinite(p.build_blob(synthetic_weights(fm, 1, 'survey'))).all()
    b13 = next(o for o in folded if o.name == 'block_13_project')
    assert b13.se_reduced == (2 | (rt.ACT['relu6'] << 8)) and (b13.srcs[0].buf.h, b13.h) == (26, 13)


def test_plan_variants(monkeypatch):
    """Batches up to small_batch run the plan without block fusion; both plans carry the same parameters, MACs and
    conv-granular bytes, and pack the same weights."""
    from yoloret_amd import runtime as rt
    from yoloret_amd.weights import synthetic_weights
    monkeypatch.setenv('YOLORET_SMALL_BATCH', '4')
    assert _model().small_variant == 'nohead'      # float32 default (round 5): the throughput plan without YR_OP_HEAD
    nh = _model().plan_for(2)
    assert rt.OP_HEAD not in [o.kind for o in nh.ops] and rt.OP_MBR in [o.kind for o in nh.ops]
    # ... and one or two images its 'nohead_k' twin, whose pointwise convs on maps of at most 32 x 32 conv pixels ask for the k-split form (se_reduced bit 17; pointwise_split.hip:
    # pwk_kernel) - a flag of the few-image PLAN: the throughput plan has none, and neither has a plan that keeps the float32 MFMA
    from yoloret_amd import compiler
    pw = [o for o in nh.ops if o.kind == rt.OP_POINTWISE]
    px = lambda o: o.h * o.w * (4 if getattr(o, 'stride', 0) == 2 else 1)
    assert _model().variant(2) == 'nohead_k' and _model().variant(3) == 'nohead'
    assert pw and all(bool(o.se_reduced & 0x20000) == (px(o) <= 32 * 32) for o in pw if not (o.se_reduced & 0x10000))
    assert any(o.se_reduced & 0x20000 for o in pw) and any(not (o.se_reduced & 0x20000) for o in pw)
    assert not any(o.se_reduced & 0x20000 for o in _model().plan.ops + _model().plan_for(3).ops if o.kind == rt.OP_POINTWISE)
    monkeypatch.setattr(compiler, 'KSPLIT_MAX_PIXELS', 0)
    assert not any(o.se_reduced & 0x20000 for o in _model().plan_for(2).ops if o.kind == rt.OP_POINTWISE)
    monkeypatch.undo()
    monkeypatch.setenv('YOLORET_SMALL_BATCH', '4')
    monkeypatch.setenv('YOLORET_SMALL_VARIANT', 'latency')
    m = _model()
    assert m.small_batch == 4 and [m.variant(b) for b in (1, 4, 5, 64)] == ['latency', 'latency', 'throughput', 'throughput']
    lat, thr = m.plan_for(1), m.plan_for(64)
    assert thr is m.plan and lat is not thr and m.plan_for(3) is lat
    kinds = [o.kind for o in lat.ops]
    assert rt.OP_MBLANE not in kinds and rt.OP_MBCONV not in kinds and kinds[0] == rt.OP_STEMBLOCK
    # the squeeze of every SE block rides on its depthwise kernel as partial sums (no SE_MEAN launches, no merged pooling)
    assert kinds.count(rt.OP_SE_MEAN) == 0 and kinds.count(rt.OP_DEPTHWISE) == 22 and kinds.count(rt.OP_SE_FC) == 6
    assert sum(1 for o in lat.ops if o.kind == rt.OP_DEPTHWISE and o.gate is not None) == 6
    # the throughput plan (ABI 7): the six head blocks' 1x1 conv -> depthwise -> squeeze-excite sums are one launch each; the FC pair
    # of the SE block stays a launch of its own (the SE tail is opt-in: se_tail.h)
    assert sum(1 for o in thr.ops if o.kind == rt.OP_HEAD and o.gate is not None) == 6 and [o.kind for o in thr.ops].count(rt.OP_SE_FC) == 6
    from yoloret_amd import compiler
    monkeypatch.setattr(compiler, 'SE_TAIL', True)
    tail = _model().plan
    assert sum(1 for o in tail.ops if o.kind == rt.OP_HEAD and o.gate_out is not None and 'se_w' in o.params) == 6 and rt.OP_SE_FC not in [o.kind for o in tail.ops]
    assert tail.total_macs() == thr.total_macs() and abs(tail.algorithmic_bytes_per_image() - thr.algorithmic_bytes_per_image()) < 1
    assert lat.param_shapes == thr.param_shapes and lat.total_macs() == thr.total_macs()
    assert abs(lat.algorithmic_bytes_per_image() - thr.algorithmic_bytes_per_image()) < 1
    assert [(b.h, b.w, b.c) for b in lat.output_bufs] == [(b.h, b.w, b.c) for b in thr.output_bufs]
    m.set_weights(synthetic_weights(m, 1, 'survey'))
    assert np.isfinite(m._blob_of('latency')).all() and not np.array_equal(m._blob_of('latency'), m._blob_of('throughput'))
    monkeypatch.setenv('YOLORET_SMALL_BATCH', '0')
    assert _model().variant(1) == 'throughput'


@pytest.mark.parametrize('policy', ['float32', 'mixed_bfloat16'])
def test_arena_has_no_overlapping_live_buffers(policy):
    from yoloret_amd import layers as L
    L.set_global_policy(policy)
    try:
        p = _model('efficientnetb0', 64).plan
    finally:
        L.set_global_policy('float32')
    arena = [b for b in p.bufs if b.external_slot < 0]
    for i, a in enumerate(arena):
        for b in arena[i + 1:]:
            overlap_t = not (a.last_use < b.first_def or b.last_use < a.first_def)
            overlap_m = not (a.offset + a.bytes <= b.offset or b.offset + b.bytes <= a.offset)   # offsets are bytes per image
            assert not (overlap_t and overlap_m), (a.name, b.name)


def test_squeeze_excite_blocks_of_a_16_bit_plan_fuse_expand_and_depthwise():
    """16-bit EfficientNet plan: every MBConv block with squeeze-excite and an expand conv of at most 128 inputs becomes
    MBX (expand + depthwise, per-tile channel sums out) -> SE_FC reading those sums -> gated projection; parameters,
    MACs and the conv-granular byte accounting are those of the unfused float32 plan."""
    from yoloret_amd import layers as L, runtime as rt, compiler as C
    f32 = _model('efficientnetb0', 416, 80).plan
    L.set_global_policy('mixed_bfloat16')
    keep = C.MBX_K5_MAX_CEXP
    try:
        default = _model('efficientnetb0', 416, 80).plan
        C.MBX_K5_MAX_CEXP = 10 ** 6       # every eligible block (the default leaves the 5x5 stride-1 blocks unfused: measured)
        p = _model('efficientnetb0', 416, 80).plan
    finally:
        C.MBX_K5_MAX_CEXP = keep
        L.set_global_policy('float32')
    ops = p.ops
    mbx = [i for i, o in enumerate(ops) if o.kind == rt.OP_MBX]
    assert len(mbx) == 11 and [ops[i].name for i in mbx][:2] == ['stage2_block0_mbx', 'stage2_block1_mbx']
    dmbx = [o for o in default.ops if o.kind == rt.OP_MBX]
    assert len(dmbx) == 7 and not any(o.k == 5 and o.stride == 1 for o in dmbx)
    assert default.param_shapes == f32.param_shapes and default.total_macs() == f32.total_macs()
    for i in mbx:
        m, fc, proj = ops[i], ops[i + 1], ops[i + 2]
        assert fc.kind == rt.OP_SE_FC and proj.kind == rt.OP_POINTWISE
        assert m.gate is not None and m.gate.dtype == 0 and fc.srcs[0].buf is m.gate and fc.k == m.h * m.w
        rows = ((m.h + 3) // 4) * ((m.w + 7) // 8 if m.h * m.w > 1000 else (m.w + 3) // 4)
        assert m.se_reduced == rows == m.gate.h and m.gate.ld >= m.cout and m.out.dtype == rt.DTYPE['bf16']
        assert proj.srcs[0].buf is m.out and proj.gate is fc.out and m.srcs[0].c <= 128
        assert m.params['wgt'][0] == ((m.cout + 31) // 32 * 32, (m.cin + 31) // 32 * 32) and m.params['wgt'][2] == rt.DTYPE['bf16']
        assert m.params['wgt2'][0] == (m.k * m.k + 4, (m.cout + 31) // 32 * 32)
    # the 13x13 blocks with 192 inputs stay unfused
    assert sum(1 for o in ops if o.kind == rt.OP_DEPTHWISE and o.name.startswith('stage6_block')) == 3
    assert p.param_shapes == f32.param_shapes and p.total_macs() == f32.total_macs()
    assert abs(2 * p.algorithmic_bytes_per_image() - f32.algorithmic_bytes_per_image()) < 1


@pytest.mark.parametrize('name', ['mobilenetv2x75', 'mobilenetv2x14', 'efficientnetb0', 'efficientnetb3', 'efficientnetb0-lite'])
def test_parameter_inventory_matches_oracle(name):
    m = _model(name, 64)
    P = params.ParamStore(3)
    om.yolov3_body(P, params.synthetic_images(1, 64, 64), name, 3, 20)
    want = {k: v.shape for k, v in P.values.items() if not re.match(r'td\d_y/', k)}  # top-down y convs are dead (panet)
    assert {k: tuple(v) for k, v in m.param_shapes.items()} == {k: tuple(v) for k, v in want.items()}


@pytest.mark.parametrize('recipe', ['survey', 'conditioned'])
def test_product_weight_recipe_equals_oracle_recipe(recipe):
    from yoloret_amd.weights import synthetic_images, synthetic_weights
    m = _model('mobilenetv2x75', 64)
    P = params.ParamStore(1234, recipe)
    om.yolov3_body(P, params.synthetic_images(1, 64, 64), 'mobilenetv2x75', 3, 20)
    w = synthetic_weights(m, 1234, recipe)
    for k, v in w.items():
        assert np.array_equal(v, P.values[k]), k
    assert np.array_equal(synthetic_images(2, 32, 32), params.synthetic_images(2, 32, 32))


def test_blob_folds_batchnorm():
    m = _model('mobilenetv2x75', 64)
    from yoloret_amd.weights import synthetic_weights
    w = synthetic_weights(m, 1, 'survey')
    blob = m.plan.build_blob(w)
    op = next(o for o in m.plan.ops if o.name == 'block_20_conv')
    off = op.offsets['scale']
    g, v = w['block_20_BN/gamma'], w['block_20_BN/moving_variance']
    assert np.allclose(blob[off:off + 256], g / np.sqrt(v + 1e-3), rtol=1e-6)


def test_surface_errors_and_defaults():
    from yoloret_amd import layers as L
    from yoloret_amd.yolo3 import efficientnet as E
    from yoloret_amd.yolo3.model import YoloEval, _make_divisible, yolov3_body
    with pytest.raises(ValueError):
        yolov3_body(L.Input(shape=[64, 64, 3]), 'resnet50', 3, num_classes=20)
    with pytest.raises(ValueError):      # unknown GlobalParams field, as namedtuple._replace in the reference
        yolov3_body(L.Input(shape=[64, 64, 3]), 'mobilenetv2x75', 3, num_classes=20, bogus=1)
    with pytest.raises(ValueError):
        yolov3_body(L.Input(shape=[60, 64, 3]), 'mobilenetv2x75', 3, num_classes=20)
    yolov3_body(L.Input(shape=[64, 64, 3]), 'mobilenetv2x75', 3, num_classes=20, drop_rate=0.2, data_format='channels_last')
    with pytest.raises(NotImplementedError):
        E.get_model_params('resnet', {})
    with pytest.raises(ValueError):
        E.BlockDecoder()._decode_block_string('r1_k3_s1_e1_i32_o16')
    a = E.BlockDecoder().decode(['r2_k5_s22_e6_i24_o40_se0.25'])[0]
    assert (a.kernel_size, a.num_repeat, a.strides, a.se_ratio) == (5, 2, [2, 2], 0.25)
    assert E.BlockDecoder().encode([a]) == ['r2_k5_s22_e6_i24_o40_se0.25']
    assert _make_divisible(24 * 0.75, 8) == 24 and _make_divisible(7, 8) == 8
    ev = YoloEval(np.zeros((9, 2)), 3, 20)
    assert ev.get_config()['max_boxes'] == 20 and ev.score_threshold == .6 and ev.iou_threshold == .5


def test_utils():
    from yoloret_amd.yolo3.utils import compose, get_anchors, get_classes
    assert compose(lambda x: x + 1, lambda x: x * 2)(3) == 8
    with pytest.raises(ValueError):
        compose()
    a = get_anchors('model_data/yolo_anchors.txt')
    assert a.shape == (9, 2) and a.dtype == np.float32 and a[0].tolist() == [10, 13] and a[-1].tolist() == [373, 326]
    assert len(get_classes('model_data/voc_classes.txt')) == 20 and len(get_classes('model_data/coco_classes.txt')) == 80


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from yoloret_amd import build, runtime as rt
    lib = build.build()
    L = ctypes.CDLL(lib)
    header = open(os.path.join(ROOT, 'include', 'yoloret_hip.h')).read()
    declared = set(re.findall(r'\b(yr_[a-z_0-9]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    for sym in declared:
        assert hasattr(L, sym), 'libyoloret_hip.so does not export %s' % sym
    assert declared == set(rt.EXPORTS)
    L.yr_abi_version.restype = ctypes.c_int
    assert L.yr_abi_version() == 9 == rt.ABI_VERSION
    # struct layouts agree with the header's (the library reports its own sizeof)
    for which, st in enumerate((rt.YrSrc, rt.YrOp, rt.YrBuf)):
        assert L.yr_abi_sizeof(which) == ctypes.sizeof(st), st.__name__
    assert ctypes.sizeof(rt.YrSrc) == 40 and ctypes.sizeof(rt.YrBuf) == 24
    rt.lib()   # the binding's own load-time checks


def test_product_does_not_import_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'yoloret_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M):
                    bad.append(f)
    assert not bad, 'product modules import the test oracle: %s' % bad


# kernels that may keep values in scratch memory, with the bytes per lane they are known to use (prefix of the mangled
# name -> cap).  Everything else must not spill: a new spill is a register-allocation accident (an `#pragma unroll` that
# was not obeyed, a select between array slots, a register cap set too low) and is caught at build time, without a GPU.
SCRATCH_ALLOWED = {
    '_Z10mbh_kernel': 200,         # the expand + depthwise (squeeze-excite) forms and the 4-cout-pair forms park prefetched parameters
    '_Z16mblane_s1_kernel': 104,   # 16-bit instantiations: output staging
    '_Z16mblane_s2_kernel': 104,
    '_Z15nms_band_kernel': 200,    # the per-lane score list of the band-wise NMS
    '_Z10pwh_kernelIDF16_Li4ELi1ELi2ELi2E': 68,    # f16 direct form, four pixel tiles, gated source
    '_Z10dwq_kernelIDF16_Li5ELi1ELb1E': 16,        # f16 5x5 swish squeeze-excite walk: three values parked outside the row loop at 168 registers
    '_Z10pws_kernelILi2ELi2ELi4ELi1ELb0E': 12,     # the 128 x 32 tile with gathered sources at 168 registers (since the k loop moved to pws_common.h; no plan of the BASELINE models picks it)
    '_Z10mbr_kernel': 12,          # split form (..ELb1EEv): the three-wave stride-1 block at 168 registers and 48 -> 288 -> 72 at 256 park two values
    '_Z12hwalk_kernelILi4ELi2ELb1ELb0E': 8,        # walking head, four chunks + an up-sampled addend at 256 registers (no BASELINE plan picks it)
}


def test_no_unexpected_scratch_spills():
    """Reads the per-kernel resource report the build keeps (yoloret_amd/build.py: -Rpass-analysis=kernel-resource-usage)."""
    from yoloret_amd import build as b
    b.build()
    bad, n = [], 0
    for src, rows in b.kernel_resources().items():
        for name, vgprs, scratch, occ, lds in rows:
            n += 1
            if scratch > 0:
                cap = max([c for pre, c in SCRATCH_ALLOWED.items() if name.startswith(pre)] or [0])
                if scratch > cap:
                    bad.append('%s: %s spills %d bytes per lane (%d VGPRs, %d waves/SIMD)' % (src, name, scratch, vgprs, occ))
    assert n > 500, 'resource report incomplete (%d kernels)' % n
    assert not bad, 'kernels with scratch memory:\n' + '\n'.join(bad)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from yoloret_amd import runtime as rt
    monkeypatch.setattr(rt, '_lib', None)
    monkeypatch.setattr(rt, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(rt.YoloretHipError, match='no CPU fallback'):
        rt.lib()


def test_central_crop_matches_tf_rule():
    """tf.image.central_crop: start = int((n - n*f)/2), size = n - 2*start (the zoom-in pass, yolo.py:108-109)."""
    from yoloret_amd import runtime as rt
    from yoloret_amd.yolo import central_crop
    img = np.arange(10 * 7 * 3, dtype=np.uint8).reshape(10, 7, 3)
    out = central_crop(img, 0.5)
    assert out.shape == (6, 5, 3) and np.array_equal(out, img[2:8, 1:6])
    assert central_crop(img, 1.0).shape == (10, 7, 3)
    z = central_crop(np.zeros((375, 500, 3), np.uint8), rt.ZOOM_RATIO)
    assert z.shape == (375 - 2 * int((375 - 375 * rt.ZOOM_RATIO) / 2), 500 - 2 * int((500 - 500 * rt.ZOOM_RATIO) / 2), 3)
    assert abs(rt.ZOOM_RATIO - 0.2899) < 1e-4 and abs(rt.ZOOM_MUL - 224 / 416) < 1e-7
    with pytest.raises(ValueError):
        central_crop(img, 0.0)


def test_serialised_plan_is_validated():
    """yr_create_from_blob refuses anything that is not a plan of this ABI (no GPU needed: the checks come first)."""
    from yoloret_amd import layers as L, runtime as rt, weights as W
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[64, 64, 3]), 'mobilenetv2x75', 3, num_classes=20)
    m.set_weights(W.synthetic_weights(m, 1, 'conditioned'))
    data = m.save_plan()
    assert data[:8] == rt.PLAN_MAGIC and len(data) > 96
    lib = rt.lib()
    h = ctypes.c_void_p()

    def load(b):
        buf = (ctypes.c_char * len(b)).from_buffer_copy(b)
        return lib.yr_create_from_blob(buf, len(b), ctypes.byref(h))
    assert load(b'NOTAPLAN' + data[8:]) == -1 and b'magic' in lib.yr_last_error()
    assert load(data[:-4]) == -1 and b'bytes' in lib.yr_last_error()
    bad_abi = bytearray(data)
    bad_abi[8] = 1
    assert load(bytes(bad_abi)) == -1 and b'ABI' in lib.yr_last_error()
    assert load(data[:50]) == -1


def _plan16(name, size, policy='mixed_bfloat16'):
    from yoloret_amd import layers as L
    L.set_global_policy(policy)
    try:
        return _model(name, size).plan
    finally:
        L.set_global_policy('float32')


def test_round3_plan_choices_of_the_16bit_configurations():
    """What the graph compiler decides for BASELINE configs 3 and 5 since round 3 (measured choices):
    the fused network entry carries the matrix-pipe parameter layout (stem BN as its own `scale` / `shift` rows, the stem kernel
    as a [C1P][32] 16-bit matrix in the kernel's k order) for stems of at most 32 channels and even image sizes only; the first
    stride-2 block is a YR_OP_MBH op (which the library hands to mbn_h.hip) instead of the float32 lane kernel; lite0's
    52 x 52 x 240 5x5 block runs unfused (LDS-walk depthwise), lite3's 80 x 80 x 288 blocks stay fused."""
    from yoloret_amd import runtime as rt
    p3 = _plan16('efficientnetb0-lite', 416)
    e = p3.ops[0]
    assert e.kind == rt.OP_STEMBLOCK and 'scale' in e.params and e.params['wgt'][0] == (32, 32) and e.params['wgt'][2] == p3.dtype
    assert e.params['wgt2'][0] == (10, 32) and e.params['b1'][0] == (16, 32)
    blk = p3.ops[1]
    assert blk.kind == rt.OP_MBH and blk.stride == 2 and blk.cin == 16 and blk.se_reduced == 96 and blk.cout == 24
    assert rt.OP_MBLANE not in [o.kind for o in p3.ops]
    names = [o.name for o in p3.ops]
    assert any(n.startswith('stage3_block1') and n.endswith('_dw') for n in names), 'lite0 stage 3 block 1 runs unfused'
    assert any(n.startswith('stage3_block0') and n.endswith('_mbh') for n in names), 'its stride-2 entry block stays fused'
    # the permuted stem kernel holds exactly the 27 taps of every output channel
    import numpy as np
    from yoloret_amd.weights import synthetic_weights
    m = _model('efficientnetb0-lite', 64)
    wd = synthetic_weights(m, 3, 'survey')
    p = _plan16('efficientnetb0-lite', 64)
    ws = p.ops[0].params['wgt'][1](wd)
    full = np.asarray(wd['stem_conv/kernel']).reshape(27, -1)
    assert ws.shape == (32, 32) and np.array_equal(np.sort(ws[:, :27], axis=1), np.sort(full.T, axis=1)) and not ws[:, 27:].any()
    # an odd image size keeps the float32-pipe entry kernel (pair-packed layout: no separate BN rows)
    assert 'scale' not in _plan16('efficientnetb0-lite', 63).ops[0].params
    p5 = _plan16('efficientnetb3-lite', 640, 'mixed_float16')
    e5 = p5.ops[0]      # 40 stem channels: the matrix-pipe layout too (stemxp_kernel, the register-chained form with projection)
    assert e5.kind == rt.OP_STEMBLOCK and e5.params['wgt'][0] == (64, 32) and e5.params['wgt2'][0] == (10, 64) and e5.params['b1'][0] == (32, 64)
    k5 = [o for o in p5.ops if o.kind == rt.OP_MBH and o.k == 5 and o.stride == 1]
    assert len(k5) == 2 and all(o.h == 80 and o.se_reduced == 288 for o in k5)
    first_s2 = next(o for o in p5.ops if o.kind == rt.OP_MBH and o.stride == 2)
    assert first_s2.cin == 24 and first_s2.se_reduced == 144 and first_s2.cout == 32            # mbn_h.hip's two-pass shape


def test_split_form_refuses_weights_beyond_the_float16_range():
    """The split forms cut every float32 operand into two float16 planes: folded weights must stay below 65504 (compiler.mbs_pack
    refuses them at plan-build time - activations cannot be checked there: INTEGRATION.md numerics note (v))."""
    import pytest
    from yoloret_amd.compiler import mbs_pack, mbs_wave_pairs
    rng = np.random.default_rng(0)
    cin, cexp, cout = 24, 144, 24
    we = rng.standard_normal((cexp, cin)).astype(np.float32)
    dw = rng.standard_normal((9, cexp)).astype(np.float32)
    wp = rng.standard_normal((cout, cexp)).astype(np.float32)
    one = lambda n: np.ones(n, np.float32)
    wa, tab, b2 = mbs_pack(we, one(cexp), one(cexp), dw, one(cexp), one(cexp), wp, one(cout), one(cout), 3)
    assert wa.dtype == np.float32 and wa.size == (9 * 1 + len(mbs_wave_pairs(9, 3)) * 2) * 512 and tab.shape == (9, 11, 16) and b2.shape == (32,)
    planes = wa.view(np.float16).reshape(-1, 2, 64, 8).astype(np.float64)
    # expand tile 0, lane 0 (m = 0, g = 0): h + 2^-11 m reproduces the weight to 22 bits
    got = planes[0, 0, 0, :] + planes[0, 1, 0, :] / 2048.0
    assert np.abs(got - we[0, :8]).max() <= 2.0 ** -21 * np.abs(we[0, :8]).max()
    big = one(cexp).copy()
    big[3] = 7.0e4
    with pytest.raises(AssertionError):
        mbs_pack(we, big, one(cexp), dw, one(cexp), one(cexp), wp, one(cout), one(cout), 3)


def test_fold_projection_handles_chains_of_linear_convs():
    """three consecutive linear 1x1 convs (P1 -> C1 -> C2) - after P1 is folded into C1 the ORIGINAL C1 must not be
    folded into C2 as a projection of its own (its replacement was never emitted: the plan read a buffer nobody wrote).  Whatever
    the pass folds, every buffer an op reads must be written by an earlier op (or be the image), and the composed plan must compute
    the same function: checked against a float64 composition of the layers."""
    from yoloret_amd import layers as L, runtime as rt
    from yoloret_amd.engine import Model
    L.reset_names()
    x = L.Input(shape=[16, 16, 3])
    t = L.ReLU(6.)(L.BatchNormalization(name='bn0')(L.Conv2D(8, 3, strides=2, use_bias=False, name='stem')(x)))
    for i, (f, act) in enumerate([(12, False), (10, False), (16, False), (8, True)]):
        t = L.BatchNormalization(name='bn%d' % (i + 1))(L.Conv2D(f, 1, use_bias=False, name='c%d' % (i + 1))(t))
        if act:
            t = L.ReLU(6.)(t)
    y = L.Conv2D(5, 1, use_bias=True, name='head')(t)
    m = Model(x, [y])
    ops = m.plan.ops
    written = {id(m.plan.input_buf)}
    for op in ops:
        for s in op.srcs:
            assert id(s.buf) in written, '%s reads %s, which no earlier op writes' % (op.name, s.buf.name)
        written.add(id(op.out))
    # the algebra: fold the chain by hand in float64 and compare the plan's composed pointwise parameters
    rng = np.random.default_rng(3)
    wd = {k: rng.standard_normal(s).astype(np.float32) * 0.3 for k, s in m.plan.param_shapes.items()}
    for k in wd:
        if k.endswith('moving_variance'):
            wd[k] = np.abs(wd[k]) + 0.5
    v = rng.standard_normal((7, 8))           # seven "pixels" behind the stem

    def bn(t_, name):
        g, b_, mu, var = (wd['%s/%s' % (name, p)].astype(np.float64) for p in ('gamma', 'beta', 'moving_mean', 'moving_variance'))
        return (t_ - mu) / np.sqrt(var + 1e-3) * g + b_
    ref = v
    for i in range(1, 5):
        ref = bn(ref @ wd['c%d/kernel' % i].reshape(-1, wd['c%d/kernel' % i].shape[-1]).astype(np.float64), 'bn%d' % i)
    ref = np.clip(ref, 0, 6) @ wd['head/kernel'].reshape(8, 5).astype(np.float64) + wd['head/bias']
    got = v
    for op in ops[1:]:
        assert op.kind == rt.OP_POINTWISE and len(op.srcs) == 1
        w_ = np.asarray(op.params['wgt'][1](wd), np.float64)[:, :got.shape[1]]
        got = got @ w_.T
        if 'scale' in op.params:
            got = got * np.asarray(op.params['scale'][1](wd), np.float64)[:op.cout] + np.asarray(op.params['shift'][1](wd), np.float64)[:op.cout]
        if op.act == 'relu6':
            got = np.clip(got, 0, 6)
    assert len(ops) < 6, 'nothing was folded'
    assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_head_blocks_of_the_16bit_plans_and_their_fragment_packing(monkeypatch):
    """Round 5: the head blocks of a 16-bit plan whose sources are identity sources (td2, td3, bu3, bu2 in every EfficientNet
    configuration) CAN run as YR_OP_HEAD ops in the walking form (headwalk_h.hip; compiler.HEAD_WALK16_MAX_NK = 8); the measured
    default takes the blocks of at most two chunks (td3).  td1 (a pooled source) and bu1 (11 chunks of 32 channels) stay conv +
    depthwise.  compiler.head_pack16 is the MFMA A-fragment order of a [F][kp] matrix whose k space is cut into chunks of 32
    channels per source; a serialised plan carrying the ops passes yr_create_from_blob's extent checks."""
    import numpy as np
    from yoloret_amd import compiler, layers as L, runtime as rt, weights as W
    from yoloret_amd.yolo3.model import yolov3_body
    assert [o.name for o in _plan16('efficientnetb0', 416).ops if o.kind == rt.OP_HEAD] == ['td3_head']      # the default
    monkeypatch.setattr(compiler, 'HEAD_WALK16_MAX_NK', 8)
    for name, size, policy in [('efficientnetb0', 416, 'mixed_bfloat16'), ('efficientnetb3-lite', 640, 'mixed_float16')]:
        p = _plan16(name, size, policy)
        heads = {o.name: o for o in p.ops if o.kind == rt.OP_HEAD}
        assert sorted(heads) == ['bu2_head', 'bu3_head', 'td2_head', 'td3_head'], (name, sorted(heads))
        for o in heads.values():
            assert o.k & 0x40 and o.dtype == p.dtype and o.out.dtype == p.dtype and o.gate is not None and o.gate.dtype == 0
            assert o.se_reduced == compiler.head_walk_rows(o.h, o.w) == o.gate.h
            nk = sum((s.c + 31) // 32 for s in o.srcs if s.xform != 'up2_add')
            assert nk <= 8 and o.params['wgt'][0] == ((o.cout // 16) * nk * 512,) and o.params['wgt'][2] == p.dtype
        assert heads['bu3_head'].res is not None and len(heads['bu3_head'].srcs) == 1          # reads td3's map through its SE gate
        names = [o.name for o in p.ops]
        assert 'td1_conv' in names and 'bu1_conv' in names and 'td2_conv' not in names
    # the fragment order: lane (m, g) of tile t, chunk j of source s holds W[16 t + m][32 j + 8 g + i of s]
    rng = np.random.default_rng(0)
    segs = [40, 75]
    kp = sum((c + 7) // 8 * 8 for c in segs)
    wt = rng.standard_normal((32, kp)).astype(np.float32)
    fr = compiler.head_pack16(wt, segs).reshape(2, 5, 4, 16, 8)           # [t][chunk][g][m][i]; chunks: 2 of source 0, 3 of source 1
    assert np.array_equal(fr[1, 0, 2, 3], wt[16 + 3, 16:24])
    assert np.array_equal(fr[0, 1, 0, 5], wt[5, 32:40]) and not fr[:, 1, 1:].any()            # source 0 ends at channel 40
    assert np.array_equal(fr[1, 4, 1, 0, :3], wt[16, 40 + 64 + 8:40 + 64 + 11]) and not fr[:, 4, 1, :, 3:].any()   # 75 = 64 + 11
    # a serialised 16-bit plan with head ops is accepted as far as a box without a GPU can tell (extent / size checks come first)
    L.set_global_policy('mixed_bfloat16')
    try:
        m = yolov3_body(L.Input(shape=[64, 64, 3]), 'efficientnetb0', 3, num_classes=20)
    finally:
        L.set_global_policy('float32')
    m.set_weights(W.synthetic_weights(m, 1, 'conditioned'))
    data = m.save_plan()
    h = ctypes.c_void_p()
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
    rc = rt.lib().yr_create_from_blob(buf, len(data), ctypes.byref(h))
    err = rt.lib().yr_last_error()
    assert rc == 0 or (b'does not fit' not in err and b'bytes' not in err and b'parameter' not in err), err
    if rc == 0:
        rt.lib().yr_destroy(h)


def test_stream_form_chunk_counts_agree_between_compiler_and_library():
    """compiler.pwt_chunks (what the plan pads a pixel-stationary conv's weight planes to) == yr_pwt_chunks (what the kernel is built
    for); the throughput plan of the headline model carries the form and its two-output pairs, the other variants do not."""
    import ctypes
    from yoloret_amd import compiler, runtime as rt
    L = rt.lib()
    L.yr_pwt_chunks.argtypes, L.yr_pwt_chunks.restype = [ctypes.c_int], ctypes.c_int
    assert all(L.yr_pwt_chunks(kp) == compiler.pwt_chunks(kp) for kp in range(4, 700, 4))
    m = _model()
    tp = [o for o in m.plan_for(64).ops if o.kind == rt.OP_POINTWISE]
    assert sum(bool(o.se_reduced & 0x40000) for o in tp) >= 10 and sorted(o.name for o in tp if o.se_reduced & 0x80000) == ['bu2_y', 'bu3_y']
    assert all(o.gate_out is not None and o.se_hidden > 0 for o in tp if o.se_reduced & 0x80000)
    m.small_batch, m.mbk_batch = 4, 24
    for b in (1, 3, 10):        # 'nohead_k', 'nohead', 'mid'
        assert m.variant(b) != 'throughput' and not any(o.se_reduced & 0xc0000 for o in m.plan_for(b).ops if o.kind == rt.OP_POINTWISE)


def test_nosplit_names_hold_in_every_plan_variant():
    """Model.check_ranges reports op names of the variant the first batch ran ('td1_head' in the throughput plan,
    'td1_conv' where YR_OP_HEAD is not fused; 'block_11_mbr' | 'block_11_mbe' | 'block_11_expand').  Compiled with the SAME nosplit set,
    no variant may keep the equivalent convolution in a split form - in either direction."""
    from yoloret_amd import compiler as C
    from yoloret_amd import layers as L
    from yoloret_amd import runtime as rt
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[416, 416, 3]), 'mobilenetv2x75', 3, num_classes=20)
    assert C.nosplit_aliases(['td1_head']) == frozenset(['td1_head', 'td1_conv'])
    assert C.nosplit_aliases(['block_11_mbe']) >= frozenset(['block_11_mbr', 'block_11_mbe', 'block_11_expand'])
    heads = [o.name for o in m.plan.ops if o.kind == rt.OP_HEAD]
    assert len(heads) == 6
    for reported in (heads, [n.replace('_head', '_conv') for n in heads]):          # as the throughput plan names them | as 'nohead' does
        for variant in (True, 'nohead', 'nohead_k'):
            plan = C.compile_graph(m.inputs[0], m.outputs, variant, 0, frozenset(reported))
            split = [plan.ops[i].name for i in C.split_form_ops(plan)]
            for n in heads:
                base = n[:-len('_head')]
                assert base + '_head' not in split and base + '_conv' not in split, (variant, reported[0], split)
            conv = {o.name: o for o in plan.ops}
            assert all(conv[n.replace('_head', '_conv')].se_reduced & 0x10000 for n in heads)       # the float32 MFMA, in every variant
    # an inverted-residual block: one-launch form (weight-streaming / register-chained) <-> expand + depthwise | projection
    for reported in (['block_11_mbr'], ['block_11_mbe'], ['block_3_mbr']):
        for variant in (True, 'nohead'):
            plan = C.compile_graph(m.inputs[0], m.outputs, variant, 0, frozenset(reported))
            base = reported[0].rsplit('_', 1)[0]
            ops = {o.name: o for o in plan.ops}
            blk = [o for n, o in ops.items() if n in (base + '_mbr', base + '_mbe')]
            assert blk and all(not (o.k & 0x80) for o in blk), (reported, variant, [(o.name, hex(o.k)) for o in blk])
    saved = C.FUSE_MBK
    C.FUSE_MBK = False
    try:       # ... and with the weight-streaming form off, the name it would have had still binds the expand + depthwise op
        plan = C.compile_graph(m.inputs[0], m.outputs, True, 0, frozenset(['block_12_mbr']))
        assert not ({o.name: o for o in plan.ops}['block_12_mbe'].k & 0x80)
    finally:
        C.FUSE_MBK = saved
