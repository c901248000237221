"""``Model``: what ``tf.keras.Model(inputs, outputs)`` is to the reference's detection
path (reference code/yolo3/model.py:342 returns ``AdvLossModel(backbone.inputs, [y1,y2,y3])``,
which on this path is only a callable container; code/yolo.py:87,152 call
``load_weights`` and ``__call__`` on it).

Holds the compiled plan, the parameter dict and the native handle; ``__call__`` enqueues
the whole forward on the current HIP stream through one C-ABI call (yr_forward).
"""
import ctypes
import os

import numpy as np
import torch

from . import runtime as rt
from .compiler import compile_graph


class Model:
    def __init__(self, inputs, outputs, name=None, fuse=None, dtype=None):
        """dtype: element type of the activations BETWEEN the fused ops and of the 1x1-conv weights - 'float32'
        (default), 'bfloat16' or 'float16' (BASELINE.json configs 3 / 5).  None takes the global policy
        (yoloret_amd.layers.set_global_policy, the counterpart of tf.keras.mixed_precision.set_global_policy).
        Images in, logits out, BatchNorm, depthwise weights, SE vectors and all accumulation stay float32."""
        if fuse is None:
            fuse = os.environ.get('YOLORET_FUSE', '1') != '0'
        if dtype is None:
            from .layers import global_policy_dtype
            dtype = global_policy_dtype()
        self.dtype = rt.dtype_id(dtype)
        self.inputs = inputs if isinstance(inputs, (list, tuple)) else [inputs]
        self.outputs = list(outputs)
        self.name = name
        self._fuse = fuse
        self._nosplit = frozenset()   # plan ops moved off the split (float16-plane) forms by check_ranges
        self._ranges_checked = False
        self.range_check = os.environ.get('YOLORET_RANGE_CHECK', '1') != '0'
        # The automatic guard measures the FIRST batch per set of weights.  range_check_every = N re-measures every N-th call (an
        # instrumented pass costs about two forwards) for deployments whose inputs may drift out of the range the first batch had;
        # 0 (default): first batch only - a later batch that pushes an activation past 65504 is NOT detected.
        self.range_check_every = int(os.environ.get('YOLORET_RANGE_CHECK_EVERY', '0'))
        self._calls = 0
        self.plan = compile_graph(self.inputs[0], self.outputs, fuse, self.dtype)
        # batches of up to SMALL_BATCH images run a second plan without block fusion (compiler.py: 'latency');
        # compiled on first use, same parameters, its own weight blob / handle / tile table.  Default (tools/lat_sweep.py, round 4):
        # 16-bit plans up to 2 images (lite0 @416 batch 1: 0.66 ms against 0.68, batch 4: 0.74 against 0.72); float32 plans never -
        # since their blocks run in the split form the fused plan is the faster one at every batch (MobileNetV2 x0.75 @416 batch 1:
        # 0.60 ms against 0.65, batch 4: 0.66 against 0.79)
        # (round 5: float32 plans run batches of up to 4 images on the 'nohead' variant - block fusion and split forms kept, the head
        # blocks' conv and depthwise as two launches: compiler.py)
        self.small_batch = int(os.environ.get('YOLORET_SMALL_BATCH', '4' if self.dtype == 0 else '2')) if fuse is True else 0
        self.small_variant = os.environ.get('YOLORET_SMALL_VARIANT', 'nohead' if self.dtype == 0 else 'latency')   # what those batches run
        # (... and the smallest of them - up to ksplit_batch images - on 'nohead_k': the same plan with the k-split form of the pointwise convs
        # on its small maps (compiler.py: KSPLIT_MAX_PIXELS).  tools/lat_sweep.py, p50 of a full step in ms with / without the form:
        # MobileNetV2 x0.75 @416 batch 1 / 2 / 4: 0.478 / 0.515 / 0.612 against 0.553 / 0.572 / 0.624; x1.4 @512: 0.615 / 0.750 / 1.053
        # against 0.714 / 0.776 / 0.946 - at four images the wide model's 16 x 16 tiles re-fetch more than the shorter chains save)
        self.ksplit_batch = int(os.environ.get('YOLORET_KSPLIT_BATCH', '2'))
        # (round 6: float32 plans run batches below mbk_batch on 'mid' - the throughput plan without the weight-streaming block form
        # (compiler.py: a one-workgroup-per-CU chain that is as long at 8 images as at 64); tools/lat_sweep.py: p50 at 8 / 16 / 32 images
        # 0.745 / 0.896 / 1.338 ms without the form, 0.826 / 0.925 / 1.293 with it)
        self.mbk_batch = int(os.environ.get('YOLORET_MBK_BATCH', '24')) if (fuse is True and self.dtype == 0) else 0
        self._plans = {'throughput': self.plan}
        self._weights = None
        self._blobs = {}
        self.autotune = os.environ.get('YOLORET_AUTOTUNE', '1') != '0'
        self._tuned = set()    # (device index, batch) pairs already autotuned
        self._handles = {}     # device index -> yr_handle*
        self._workspace = {}   # device index -> torch.uint8 tensor
        self._out_anchors = [o.node.attrs['num_anchors'] if o.node is not None and o.node.op == 'reshape5' else None
                             for o in self.outputs]

    # ------------------------------------------------------------------ parameters
    @property
    def param_shapes(self):
        """{keras-style parameter name: shape} for every live layer."""
        return dict(self.plan.param_shapes)

    def count_params(self):
        return int(sum(int(np.prod(s)) for s in self.plan.param_shapes.values()))

    def set_weights(self, weights):
        """weights: mapping name -> array covering every entry of ``param_shapes``."""
        wd = {}
        missing = [k for k in self.plan.param_shapes if k not in weights]
        if missing:
            raise ValueError('missing parameters: %s%s' % (', '.join(missing[:8]), ' ...' if len(missing) > 8 else ''))
        for k, shape in self.plan.param_shapes.items():
            a = np.asarray(weights[k], np.float32)
            if a.size != int(np.prod(shape)):
                raise ValueError('parameter %s has %d elements, expected shape %s' % (k, a.size, (shape,)))
            wd[k] = a.reshape(shape)
        self._weights = wd
        self._blobs = {}
        self._ranges_checked = False      # new weights, new activation ranges
        for (dev, variant), h in self._handles.items():
            blob = self._blob_of(variant)
            with torch.cuda.device(dev):
                rt.check(rt.lib().yr_load_weights(h, blob.ctypes.data_as(ctypes.c_void_p), blob.size))

    def get_weights(self):
        return dict(self._weights) if self._weights is not None else None

    def load_weights(self, path):
        """reference code/yolo.py:87 ``self.model.load_weights(self.model_path)``: a Keras weights-only HDF5 checkpoint
        (``*.h5`` / ``*.hdf5`` / ``*.keras.h5``; read by yoloret_amd.h5lite, layers matched as yoloret_amd.keras_h5
        describes), or an ``.npz`` written by ``save_weights`` (parameter name -> array)."""
        with open(path, 'rb') as f:
            magic = f.read(8)
        if magic == b'\x89HDF\r\n\x1a\n':
            from .keras_h5 import load_keras_h5
            self.set_weights(load_keras_h5(self, path))
            return
        with np.load(path) as z:
            self.set_weights({k: z[k] for k in z.files})

    def save_plan(self, path=None, batch=None):
        """The compiled model as ONE self-contained byte blob for yr_create_from_blob (include/yoloret_hip.h): fused
        op list, buffer table, parameter blob and every tile table autotuned so far.  A host without this Python
        package instantiates the model from the file.  batch: which plan variant (None: the
        throughput plan).  Returns the bytes; also written to `path` when given."""
        variant = 'throughput' if batch is None else self.variant(batch)
        plan = self._plans.get(variant) or self.plan_for(batch)
        blob = self._blob_of(variant)
        ops, bufs = plan.c_arrays()
        tuning = {}
        for (idx, v), h in self._handles.items():
            if v != variant:
                continue
            for (i2, b) in self._tuned:
                if i2 == idx and self.variant(b) == variant:
                    arr = (ctypes.c_int32 * len(plan.ops))()
                    if rt.lib().yr_get_tuning(h, b, arr, len(plan.ops)) == 0:
                        tuning[b] = list(arr)
        data = rt.pack_plan(ops, bufs, blob, plan.input_shape[:2], [(ob.h, ob.w, ob.c) for ob in plan.output_bufs], tuning)
        if path is not None:
            with open(path, 'wb') as f:
                f.write(data)
        return data

    def save_weights(self, path):
        if self._weights is None:
            raise RuntimeError('no weights set')
        np.savez(path, **self._weights)

    # ------------------------------------------------------------------ execution
    def variant(self, batch):
        """Which plan a batch of this size runs: from mbk_batch images on 'throughput', below it 'mid' (float32 plans: the same without the
        weight-streaming block form); up to small_batch images `small_variant` - 'latency' (no block fusion: 16-bit plans) or
        'nohead' (the throughput plan without YR_OP_HEAD: float32 plans; 'nohead_k' up to ksplit_batch images: its small maps' pointwise
        convs in the k-split form)."""
        if not 0 < batch <= self.small_batch:
            return 'mid' if 0 < batch < self.mbk_batch else 'throughput'
        if self.small_variant == 'nohead' and batch <= self.ksplit_batch:
            return 'nohead_k'
        return self.small_variant

    def plan_for(self, batch):
        v = self.variant(batch)
        if v not in self._plans:
            self._plans[v] = compile_graph(self.inputs[0], self.outputs, v, self.dtype, self._nosplit)
        return self._plans[v]

    def _blob_of(self, variant):
        if self._weights is None:
            raise RuntimeError('weights have not been set (set_weights / load_weights)')
        from .compiler import WeightRangeError
        for _attempt in range(64):
            if variant in self._blobs:
                return self._blobs[variant]
            if variant not in self._plans:       # (the plans were dropped by a fallback below)
                self._plans[variant] = compile_graph(self.inputs[0], self.outputs, self._fuse if variant == 'throughput' else variant, self.dtype, self._nosplit)
                if variant == 'throughput':
                    self.plan = self._plans[variant]
            try:
                self._blobs[variant] = self._plans[variant].build_blob(self._weights)
            except WeightRangeError as e:
                # a WEIGHT of a split-form op beyond the float16 range: that op runs the float32 MFMA in every plan variant (the same
                # fallback check_ranges takes for activations) - never a failed assertion, never a clamped weight
                if e.op_name in self._nosplit or self.dtype != 0:
                    raise
                self._drop_split_forms([e.op_name])
        raise RuntimeError('no plan without out-of-range weights found')

    def _drop_split_forms(self, names):
        """The named plan ops (and what the same convolutions are called in the other plan variants: compiler.nosplit_aliases) leave
        the split forms.  Everything compiled / uploaded so far belongs to the old plans: destroyed - after the device has finished
        whatever other contexts still have in flight on them."""
        from .compiler import nosplit_aliases
        self._nosplit = nosplit_aliases(self._nosplit | frozenset(names))
        if self._handles:
            torch.cuda.synchronize()
        for h in self._handles.values():
            rt.lib().yr_destroy(h)
        self._handles, self._blobs, self._tuned = {}, {}, set()
        self.plan = compile_graph(self.inputs[0], self.outputs, self._fuse, self.dtype, self._nosplit)
        self._plans = {'throughput': self.plan}
        self.range_fallbacks = sorted(self._nosplit)

    def _handle(self, device, batch=0):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        variant = self.variant(batch)
        h = self._handles.get((idx, variant))
        if h is None:
            self.plan_for(batch)
            blob = self._blob_of(variant)          # (may rebuild the plans: a weight beyond the float16 range leaves the split forms)
            plan = self.plan_for(batch)
            ops, bufs = plan.c_arrays()
            hp = ctypes.c_void_p()
            with torch.cuda.device(idx):
                rt.check(rt.lib().yr_create(ops, len(ops), bufs, len(bufs), ctypes.byref(hp)))
                rt.check(rt.lib().yr_load_weights(hp, blob.ctypes.data_as(ctypes.c_void_p), blob.size))
            self._handles[(idx, variant)] = h = hp
        return idx, h

    def workspace_bytes(self, batch):
        """== yr_workspace_bytes: the arena of intermediate tensors, then (ABI 7) the arrival counters of the ops that finish a
        squeeze-excite block themselves (one word per such op and image, 16-byte aligned behind the arena)."""
        plan = self.plan_for(batch)
        n_sync = sum(1 for op in plan.ops if op.gate_out is not None)
        return (plan.arena_bytes_per_image * batch + 15) // 16 * 16 + 4 * n_sync * batch

    def __call__(self, x, out=None, ctx=0):
        """ctx: execution context.  The plan handle is read-only during a forward; what a step owns is its WORKSPACE (the
        arena of intermediate tensors).  Calls with different `ctx` use different workspaces and may therefore be in
        flight at the same time on different HIP streams (pipeline.DetectionPipeline(depth=2)); calls with the same
        `ctx` must be stream-ordered.  ctx 0 is the workspace of plain `model(x)` calls on the caller's stream;
        DetectionPipeline(depth > 1) runs its contexts as ctx 1 .. depth, so a serial step on the same Model never shares an
        arena with a step in flight on a context stream."""
        h, w, c = self.plan.input_shape
        want = rt.TORCH_DTYPE[self.plan.input_buf.dtype]       # float32, or uint8 for a model built on Input(dtype='uint8')
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == want):
            raise ValueError('input must be a %s CUDA tensor [B,%d,%d,%d] (NHWC)' % (str(want).replace('torch.', ''), h, w, c))
        if x.dim() != 4 or tuple(x.shape[1:]) != (h, w, c):
            raise ValueError('input shape %s does not match the model input [B,%d,%d,%d]' % (tuple(x.shape), h, w, c))
        x = x.contiguous()
        b = x.shape[0]
        self._calls += 1
        if self.range_check and self.dtype == 0 and (not self._ranges_checked or (self.range_check_every > 0 and self._calls % self.range_check_every == 0)):
            # once per set of weights, on the first batch seen (and every range_check_every-th call): the split-form ops' operands
            # must stay inside the float16 range
            self._ranges_checked = True
            self.check_ranges(x, on_exceed='fallback')
        idx, hd = self._handle(x.device, b)
        need = self.workspace_bytes(b)
        wkey = idx if ctx == 0 else (idx, ctx)
        ws = self._workspace.get(wkey)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
            self._workspace[wkey] = ws
        ys = out
        if ys is None:
            ys = [torch.empty((b, ob.h, ob.w, ob.c), dtype=torch.float32, device=x.device)
                  for ob in self.plan.output_bufs]
        yp = [rt._ptr(y) for y in ys] + [None] * (3 - len(ys))
        with torch.cuda.device(idx):
            if self.autotune and (idx, b) not in self._tuned:
                # first call with this batch size: time every pointwise tile shape / block segmentation per layer once (~0.1 s).
                # The choice changes speed only, never results: every shape of a conv runs the same sum per output, the block kernels'
                # maps do not depend on the row segments, and since round 6 neither do the squeeze sums of the register-chained
                # expand + depthwise ops (grouped by row quanta fixed by the map's shape: tests/test_gpu_narrow.py::
                # test_se_model_results_do_not_depend_on_the_tuning_table_or_the_batch).  The one exception is a forced LDS-tiled
                # YR_OP_MBX (YOLORET_MBXR=0, an A/B switch): its sums are grouped by the tuned tile.
                self._tuned.add((idx, b))
                if not self._load_tuning(hd, b):
                    rt.check(rt.lib().yr_autotune(hd, rt._ptr(x), b, yp[0], yp[1], yp[2], rt._ptr(ws), ws.numel(),
                                                  rt.stream_ptr(x.device), 3))
                    self._save_tuning(hd, b)
            rt.check(rt.lib().yr_forward(hd, rt._ptr(x), b, yp[0], yp[1], yp[2], rt._ptr(ws), ws.numel(),
                                         rt.stream_ptr(x.device)))
        res = []
        for y, a in zip(ys, self._out_anchors):
            res.append(y.view(b, y.shape[1], y.shape[2], a, y.shape[3] // a) if a else y)
        return res

    predict = __call__

    # ---- the tuning table as a value: what rank 0 broadcasts to the other ranks (parallel.share_tuning)
    def get_tuning(self, batch, device=None):
        """The tile / segment table yr_autotune chose for `batch` images on `device` (one int per plan op: yr_get_tuning), or None if
        that batch size has not been tuned there."""
        device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        idx, hd = self._handle(device, batch)
        n = len(self.plan_for(batch).ops)
        arr = (ctypes.c_int32 * n)()
        with torch.cuda.device(idx):
            if rt.lib().yr_get_tuning(hd, batch, arr, n) != 0:
                return None
        return [int(v) for v in arr]

    def set_tuning(self, batch, table, device=None):
        """Installs a table (get_tuning's, from this or another process with the same plan) for `batch` images: the first call with
        that batch size then runs no trial launches.  Raises on a table of the wrong length or with an entry the op cannot take."""
        device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        idx, hd = self._handle(device, batch)
        n = len(self.plan_for(batch).ops)
        if len(table) != n:
            raise ValueError('tuning table of %d entries for a plan of %d ops' % (len(table), n))
        arr = (ctypes.c_int32 * n)(*[int(v) for v in table])
        with torch.cuda.device(idx):
            rt.check(rt.lib().yr_set_tuning(hd, batch, arr, n))
        self._tuned.add((idx, batch))

    # ---- tuning cache: YOLORET_TUNE_CACHE=<file.json> keeps the autotuned tile table across processes
    # (tune once, deploy many; also keeps profiler runs free of the tuner's trial launches)
    def _tune_key(self, b):
        import zlib
        sig = zlib.crc32(' '.join('%s:%d:%d:%d:%d:%d' % (o.name, o.kind, o.cin, o.cout, o.dtype, o.k & 0xffff) for o in self.plan_for(b).ops).encode())   # (k: kernel size, split bit, waves per workgroup the fragments are packed for)
        return '%08x:%d' % (sig, b)

    def _load_tuning(self, hd, b):
        path = os.environ.get('YOLORET_TUNE_CACHE')
        if not path or not os.path.exists(path):
            return False
        import json
        try:
            table = json.load(open(path)).get(self._tune_key(b))
        except (OSError, ValueError):
            return False
        n = len(self.plan_for(b).ops)
        if not isinstance(table, list) or len(table) != n:
            return False
        arr = (ctypes.c_int32 * n)(*[int(v) for v in table])
        rt.check(rt.lib().yr_set_tuning(hd, b, arr, n))
        return True

    def _save_tuning(self, hd, b):
        path = os.environ.get('YOLORET_TUNE_CACHE')
        if not path:
            return
        import json
        n = len(self.plan_for(b).ops)
        arr = (ctypes.c_int32 * n)()
        rt.check(rt.lib().yr_get_tuning(hd, b, arr, n))
        try:
            data = json.load(open(path)) if os.path.exists(path) else {}
        except (OSError, ValueError):
            data = {}
        data[self._tune_key(b)] = list(arr)
        with open(path, 'w') as f:
            json.dump(data, f)

    def check_ranges(self, x, limit=None, on_exceed='fallback'):
        """The activation-range guard of the split forms.  A float32 plan runs its 1x1 convolutions on the 16-bit matrix pipe with
        every float32 operand as two float16 planes (22 significant bits, float32-grade results) - which needs |x| < 65504, a bound
        the reference's float32 convolutions do not have (code/yolo3/model.py:20-30; MobileNetV2's linear bottlenecks and residual
        sums are unclamped, override.py:290-341).  This runs one instrumented pass on `x` (yr_forward_ranges: the largest |value|
        every op reads) and, for split-form ops that see more than `limit` (default compiler.SPLIT_LIMIT = 60000) or a NaN:
          on_exceed='fallback': rebuilds the plan with those ops on the float32-MFMA forms (full float32 range; never a silent
                                 clamped value) - what Model.__call__ does once per set of weights on the first batch it sees;
          on_exceed='raise':    raises ValueError naming them;   on_exceed='report': changes nothing.
        Returns {op name: max |input|} for every op of the plan that ran."""
        from . import compiler as C
        if limit is None:
            limit = C.SPLIT_LIMIT
        if self.dtype != 0:
            return {}
        x = x.contiguous()
        b = x.shape[0]
        result = {}
        for _attempt in range(2):       # (second pass: the rebuilt plan, which must come out clean)
            idx, hd = self._handle(x.device, b)
            plan = self.plan_for(b)
            need = self.workspace_bytes(b)
            ws = self._workspace.get((idx, 'ranges'))      # its own arena: a step in flight on another context keeps its workspace
            if ws is None or ws.numel() < need:
                ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
                self._workspace[(idx, 'ranges')] = ws
            ys = [torch.empty((b, ob.h, ob.w, ob.c), dtype=torch.float32, device=x.device) for ob in plan.output_bufs]
            mx = (ctypes.c_float * len(plan.ops))()
            with torch.cuda.device(idx):
                rt.check(rt.lib().yr_forward_ranges(hd, rt._ptr(x), b, rt._ptr(ys[0]), rt._ptr(ys[1]), rt._ptr(ys[2]), rt._ptr(ws), ws.numel(),
                                                    rt.stream_ptr(x.device), mx))
            result = {op.name: float(mx[i]) for i, op in enumerate(plan.ops)}
            bad = [plan.ops[i].name for i in C.split_form_ops(plan) if not (mx[i] <= limit)]
            bad += [plan.ops[i].second_name for i in C.split_form_ops(plan) if not (mx[i] <= limit) and getattr(plan.ops[i], 'second_name', None)]   # (a two-output conv: both convs read that map)
            if not bad:
                self._ranges_checked = True       # (a 'report' / 'raise' that found something leaves the automatic guard armed)
                break
            if on_exceed == 'report':
                break
            if on_exceed == 'raise':
                raise ValueError('activations beyond the float16 range (%.0f) enter split-form ops: %s'
                                 % (limit, ', '.join('%s (%.3g)' % (n, result[n]) for n in bad)))
            # fallback: those ops leave the split forms (in every plan variant)
            if _attempt == 1:
                raise RuntimeError('activations beyond the float16 range still enter split-form ops after the fallback: %s'
                                   % ', '.join('%s (%.3g)' % (n, result[n]) for n in bad))
            self._drop_split_forms(bad)
        return result

    def profile(self, x, iters=5):
        """Per-op timing (hipEvent pair around every launch, averaged over `iters` replays).
        Returns a list of dicts: name, kind, kernel (symbol), ms, macs, bytes (algorithmic in+out+residual)."""
        x = x.contiguous()
        b = x.shape[0]
        plan = self.plan_for(b)
        idx, hd = self._handle(x.device, b)
        self(x)  # allocates the workspace and validates the input
        ws = self._workspace[idx]
        ys = [torch.empty((b, ob.h, ob.w, ob.c), dtype=torch.float32, device=x.device)
              for ob in plan.output_bufs]
        n = len(plan.ops)
        ms = (ctypes.c_float * n)()
        names = (ctypes.c_char_p * n)()
        with torch.cuda.device(idx):
            rt.check(rt.lib().yr_forward_profile(hd, rt._ptr(x), b, rt._ptr(ys[0]), rt._ptr(ys[1]), rt._ptr(ys[2]),
                                                 rt._ptr(ws), ws.numel(), rt.stream_ptr(x.device), int(iters),
                                                 ms, names))
        out = []
        per_op_bytes = plan.algorithmic_bytes_per_op()
        per_op_hbm = plan.hbm_bytes_per_op()   # fused ops: only what still has to cross HBM (block in + out)
        for i, op in enumerate(plan.ops):
            # which pipe the multiply-adds run on: the 1x1 convolutions of a 16-bit plan on bf16 / f16 MFMA (also inside the
            # fused block kernels mbh / mbx, whose depthwise stage stays on packed float32 FMAs); everything else float32
            m16, m32 = 0, None      # (m32: multiply-adds on the float32 pipe where that is not macs - m16)
            if op.dtype != 0:
                if op.kind == rt.OP_POINTWISE:
                    m16 = op.macs
                elif op.kind in (rt.OP_MBH, rt.OP_MBX):
                    kk = (op.k & 0xff) ** 2
                    cexp = op.se_reduced if op.kind == rt.OP_MBH else op.cout
                    m16 = max(op.macs - op.h * op.w * kk * cexp, 0)
            elif op.kind in (rt.OP_MBE, rt.OP_MBR) and op.k & 0x80:
                # the split form: the 1x1 convolutions run on the 16-bit pipe as THREE float16 products per multiply-add (what that
                # pipe executes), the depthwise stage on the float32 pipe
                m32 = op.h * op.w * 9 * (op.cout if op.kind == rt.OP_MBE else op.se_reduced)
                m16 = 3 * max(op.macs - m32, 0)
            elif op.kind == rt.OP_HEAD:     # the 1x1 convolution in the split form, the depthwise stage on the float32 pipe
                m32 = op.h * op.w * 9 * op.cout
                m16 = 3 * max(op.macs - m32, 0)
            out.append(dict(name=op.name, kind=rt.OP_NAMES[op.kind], kernel=(names[i] or b'').decode(),
                            ms=float(ms[i]), macs=op.macs * b, macs_mfma16=m16 * b, macs_fp32=(op.macs - m16 if m32 is None else m32) * b, bytes=per_op_bytes[i] * b, hbm_bytes=per_op_hbm[i] * b))
        return out

    def __del__(self):
        try:
            for h in self._handles.values():
                rt.lib().yr_destroy(h)
        except Exception:
            pass
