"""Lowers a ``yoloret_amd.layers`` graph to the fused op list libyoloret_hip.so executes.

Fusions:
  * Conv2D 1x1 [+bias] [+BatchNorm] [+ReLU6/Swish/sigmoid] [+Add]  -> one POINTWISE op
    (BN folded to a per-channel scale/shift applied to the fp32 accumulator);
  * UpSampling2D / MaxPooling2D / Concatenate feeding a 1x1 conv      -> folded into its loads
    (never materialised); Multiply(SE gate, x) feeding a 1x1 conv     -> gate applied on load;
  * DepthwiseConv2D + BatchNorm + activation                          -> one DEPTHWISE op;
  * Conv2D 3x3 s2 on the 3-channel image + BatchNorm + activation     -> STEM op;
  * Mean -> Conv+bias -> Swish -> Conv+bias -> sigmoid                -> SE_MEAN + SE_FC ops.
Buffers are placed in one arena with liveness-based reuse; offsets are per image so the
plan is batch-independent.
"""
import ctypes
import os

import numpy as np

from . import runtime as rt


def round_up(v, m):
    return (v + m - 1) // m * m


NO_ARENA_REUSE = os.environ.get('YOLORET_NO_ARENA_REUSE', '0') != '0'   # debugging: every intermediate keeps its own memory (tools/plan_diff.py)


class Buf:
    def __init__(self, bid, h, w, c, ld, external_slot=-1, name='', dtype=0):
        self.id, self.h, self.w, self.c, self.ld = bid, h, w, c, ld
        self.external_slot = external_slot
        self.name = name
        self.dtype = dtype            # yr_dtype of the elements (rt.DTYPE)
        self.elems = h * w * ld
        self.bytes = self.elems * rt.ESIZE[dtype]
        self.first_def = None
        self.last_use = -1
        self.offset = -1


class Seg:
    def __init__(self, buf, c, xform='identity'):
        self.buf, self.c, self.xform = buf, c, xform


class Value:
    """How a symbolic tensor is read: concatenated segments, optionally SE-gated."""

    def __init__(self, segs, gate=None):
        self.segs, self.gate = segs, gate

    @property
    def plain(self):
        return len(self.segs) == 1 and self.segs[0].xform == 'identity' and self.gate is None


class OpRec:
    """Python-side record of one fused op (turned into a ctypes YrOp at pack time)."""

    def __init__(self, kind, name, **kw):
        self.kind, self.name = kind, name
        self.act = 'none'
        self.h = self.w = self.cin = self.cout = 0
        self.k = self.stride = 0
        self.se_reduced = 0
        self.srcs = []
        self.out = None
        self.res = None
        self.gate = None
        self.gate_out = None  # ABI 7, the SE tail: the gate vector this op writes itself (se_hidden = the FC pair's hidden width, params['se_w'])
        self.se_hidden = 0
        self.reserved0 = 0    # (a two-output POINTWISE op: the second output's activation | pooled << 8)
        self.params = {}      # role -> (shape, numpy builder fn(weights) -> float32 array[, yr_dtype it is stored as])
        self.offsets = {}     # role -> float offset in blob
        self.macs = 0
        self.dtype = 0        # yr_dtype the op works in (sources, residual, pointwise weights)
        self.__dict__.update(kw)


class WeightRangeError(ValueError):
    """A weight of a split-form op (float16 planes) is beyond the float16 range.  `op_name`: the plan op."""

    def __init__(self, op_name, msg):
        ValueError.__init__(self, msg)
        self.op_name = op_name


class Plan:
    def __init__(self, ops, bufs, inputs, outputs, param_shapes, input_shape, dtype=0):
        self.ops, self.bufs = ops, bufs
        self.dtype = dtype            # yr_dtype of the activations between ops (images, logits and SE vectors: float32)
        self.input_buf, self.output_bufs = inputs, outputs
        self.param_shapes = param_shapes
        self.input_shape = input_shape
        self._liveness()
        self._assign_arena()
        self._assign_blob_offsets()

    def _liveness(self):
        """(Re)derive first definition / last use from the final op list, drop buffers that no op
        touches any more (intermediates removed by fusion) and renumber the table."""
        for b in self.bufs:
            b.first_def, b.last_use = None, -1
        self.input_buf.first_def = -1
        for i, op in enumerate(self.ops):
            if op.out.first_def is None:
                op.out.first_def = i
            for s in op.srcs:
                s.buf.last_use = max(s.buf.last_use, i)
            for b in (op.res, op.gate, op.gate_out):
                if b is not None:
                    b.last_use = max(b.last_use, i)
                    if (b is op.gate and op.kind in (rt.OP_DEPTHWISE, rt.OP_MBX, rt.OP_STEMBLOCK, rt.OP_HEAD) or b is op.gate_out) and b.first_def is None:
                        b.first_def = i          # SE form: the depthwise op WRITES its per-workgroup channel sums there (and, with the SE tail, the gate)
        self.bufs = [b for b in self.bufs if b.first_def is not None]
        for i, b in enumerate(self.bufs):
            b.id = i

    # -- arena: first-fit with liveness reuse (offsets in BYTES per image, multiples of 16)
    def _assign_arena(self):
        arena = [b for b in self.bufs if b.external_slot < 0]
        for b in arena:
            if b.last_use < b.first_def:
                b.last_use = b.first_def
        live = []  # (offset, size, last_use)
        total = 0
        for b in sorted(arena, key=lambda x: x.first_def):
            if not NO_ARENA_REUSE:
                live = [l for l in live if l[2] >= b.first_def]  # a buffer read by op i may not be overwritten by op i
            size = round_up(b.bytes, 16)
            off = 0
            for lo, ls, _ in sorted(live):
                if off + size <= lo:
                    break
                off = max(off, lo + ls)
            b.offset = off
            live.append((off, size, b.last_use))
            total = max(total, off + size)
        self.arena_bytes_per_image = total

    def _assign_blob_offsets(self):
        off = 0
        self.blob_layout = []
        for op in self.ops:
            for role, prm in op.params.items():
                shape, dt = prm[0], (prm[2] if len(prm) > 2 else 0)
                n = int(np.prod(shape))
                op.offsets[role] = off
                self.blob_layout.append((op, role, off, shape, dt))
                off += round_up((n * rt.ESIZE[dt] + 3) // 4, 4)   # the blob is addressed in floats; a 16-bit matrix takes n/2
        self.blob_floats = max(off, 4)

    def build_blob(self, weights):
        """(raises WeightRangeError naming the op when a split-form op's weights do not fit float16 planes: engine.Model then rebuilds
        the plan with that op on the float32 MFMA)"""
        try:
            return self._build_blob(weights)
        except WeightRangeError:
            raise

    def _build_blob(self, weights):
        """The flat parameter blob in op order.  float32 everywhere except the pointwise weight matrices of a 16-bit
        plan, which are rounded (to nearest even) to that type here, once, and stored packed two per float slot."""
        blob = np.zeros(self.blob_floats, np.float32)
        for op, role, off, shape, dt in self.blob_layout:
            try:
                arr = np.asarray(op.params[role][1](weights), np.float32)
            except AssertionError as e:      # (head_pack / mbs_pack: the planes of a weight beyond the float16 range)
                if 'beyond the float16 range' in str(e):
                    raise WeightRangeError(op.name, str(e))
                raise
            assert tuple(arr.shape) == tuple(shape), (op.name, role, arr.shape, shape)
            if dt == 0:
                if (op.kind == rt.OP_POINTWISE and role == 'wgt' and op.dtype == 0 and PW_SPLIT and not (op.se_reduced & 0x50000)     # (bit 18: planes - head_pack has checked)
                        and arr.size and float(np.abs(arr).max()) >= 60000.0):
                    raise WeightRangeError(op.name, '%s: a weight of %.3g is beyond the float16 range the split pointwise form needs (YOLORET_PW_SPLIT=0 '
                                           'runs the float32-MFMA kernels)' % (op.name, float(np.abs(arr).max())))
                blob[off:off + arr.size] = arr.ravel()
            else:
                bits = rt.to_bits16(arr.ravel(), dt)
                if bits.size % 2:
                    bits = np.concatenate([bits, np.zeros(1, np.uint16)])
                blob[off:off + bits.size // 2] = bits.view(np.float32)
        return blob

    def c_arrays(self):
        ops = (rt.YrOp * len(self.ops))()
        for i, r in enumerate(self.ops):
            o = ops[i]
            o.kind, o.act = r.kind, rt.ACT[r.act]
            o.h, o.w, o.cin, o.cout, o.k, o.stride = r.h, r.w, r.cin, r.cout, r.k, r.stride
            o.nsrc, o.se_reduced = len(r.srcs), r.se_reduced
            o.dtype, o.out_dtype = r.dtype, r.out.dtype
            for j, s in enumerate(r.srcs):
                o.src[j].ptr = None
                o.src[j].buf = s.buf.id
                o.src[j].h, o.src[j].w, o.src[j].c, o.src[j].ld = s.buf.h, s.buf.w, s.c, s.buf.ld
                o.src[j].xform = rt.XFORM[s.xform]
                o.src[j].dtype = s.buf.dtype
            o.out_buf, o.out_ld = r.out.id, r.out.ld
            o.res_buf, o.res_ld = (r.res.id, r.res.ld) if r.res is not None else (-1, 0)
            o.gate_buf, o.gate_ld = (r.gate.id, r.gate.ld) if r.gate is not None else (-1, 0)
            o.gate_out_buf, o.gate_out_ld = (r.gate_out.id, r.gate_out.ld) if r.gate_out is not None else (-1, 0)
            o.se_hidden = r.se_hidden
            o.reserved0 = getattr(r, 'reserved0', 0)
            roles = {'wgt': 'wgt_off', 'scale': 'scale_off', 'shift': 'shift_off', 'wgt2': 'wgt2_off',
                     'b1': 'b1_off', 'b2': 'b2_off', 'se_w': 'se_w_off'}
            for role, field in roles.items():
                setattr(o, field, r.offsets.get(role, -1))
        bufs = (rt.YrBuf * len(self.bufs))()
        for i, b in enumerate(self.bufs):
            assert b.id == i
            bufs[i].bytes_per_image = b.bytes
            bufs[i].arena_off_per_image = b.offset if b.external_slot < 0 else -1
            bufs[i].external_slot = b.external_slot
            bufs[i].dtype = b.dtype
        return ops, bufs

    # -- reporting
    def fallback_ops(self):
        """Names of the ops that run a GENERIC form where a fused one exists for their kind of block: a DEPTHWISE 3x3 / 5x5 op fed by a
        1x1 convolution's private output (an inverted-residual or head block no fused kernel's shape list took), and HEAD blocks on
        the register-staged front end (pooled sources).  bench.py prints the list: a shape that is on no whitelist must not go unnoticed."""
        producer = {id(op.out): op for op in self.ops}
        nreaders = {}
        for op in self.ops:
            for s_ in op.srcs:
                nreaders[id(s_.buf)] = nreaders.get(id(s_.buf), 0) + 1
        names = []
        for op in self.ops:
            if op.kind == rt.OP_DEPTHWISE and len(op.srcs) == 1:
                p_ = producer.get(id(op.srcs[0].buf))
                if p_ is not None and p_.kind == rt.OP_POINTWISE and p_.act in ('relu6', 'swish') and nreaders.get(id(p_.out), 0) == 1:
                    names.append(op.name)
            elif op.kind == rt.OP_HEAD and not (op.k & 0xc0):
                names.append(op.name)
        return names

    def total_macs(self):
        return sum(op.macs for op in self.ops)

    def algorithmic_bytes_per_op(self):
        """Per-op share of SURVEY.md 8(d)'s bytes_alg, per image, fp32: conv-granular activation
        reads + writes (+ residual reads); upsample / maxpool / concat / SE-scale free; the weighted
        sum is charged its source reads and its consumer DW no input read (they are one fused op
        in the accounting); SE mean costs its C outputs."""
        wsum_outs = set(id(op.out) for op in self.ops if op.kind == rt.OP_WSUM or getattr(op, 'folded_wsum', False))

        def src_elems(op, s):
            # SURVEY.md Appendix B accounting (the agreed figure: 198.9 MB/img for MBV2x0.75@416):
            # the 4x4-pooled tap (rfcr_b4c) is counted at its pre-pool size, every other source
            # at the consumer's size (td2 reads 26x26x424, bu2 reads 26x26x203).
            if s.xform == 'maxpool4':
                return s.buf.h * s.buf.w * s.c
            return op.h * op.w * s.c

        def elems_of(op):
            elems = sum(elems_of(f) for f in getattr(op, 'absorbed', ()))   # (an SE_FC op whose work the op's SE tail does)
            if getattr(op, 'fused', None):
                # the accounting stays conv-granular (the figure everyone computes from): a fused
                # op (block kernels; a projection with its depthwise stage folded into the loads) is
                # charged what its convolutions would move unfused
                return elems + sum(elems_of(f) for f in op.fused)
            if getattr(op, 'accounted_in', None):
                return 0   # the low-resolution half of a hoisted conv: charged to the conv it was split from
            if op.kind in (rt.OP_STEM, rt.OP_POINTWISE, rt.OP_DEPTHWISE):
                if hasattr(op, 'accounting_hw'):   # pooled-output conv: charged at the conv's own resolution
                    ah, aw = op.accounting_hw
                    return (ah * aw * op.cout + sum(ah * aw * s.c for s in op.srcs))
                if not (op.kind == rt.OP_POINTWISE and op.h == 1 and op.w == 1):
                    elems += op.h * op.w * op.cout
                    if op.kind == rt.OP_DEPTHWISE or op.kind == rt.OP_STEM:
                        s = op.srcs[0]
                        if id(s.buf) not in wsum_outs:
                            elems += s.buf.h * s.buf.w * s.c
                    else:
                        elems += sum(src_elems(op, s) for s in getattr(op, 'accounting_srcs', op.srcs))
                    if op.res is not None:
                        elems += op.h * op.w * op.cout
            elif op.kind == rt.OP_WSUM:   # (a source pooled by its producer is still charged at its pre-pool size)
                elems += sum(int(np.prod(getattr(s.buf, 'accounting_hw', (s.buf.h, s.buf.w)))) * s.c for s in op.srcs)
            elif op.kind == rt.OP_SE_MEAN:
                elems += op.cout
            elif op.kind == rt.OP_SE_FC:
                elems += getattr(op, 'merged_mean', 0)
            return elems

        return [elems_of(op) * rt.ESIZE[self.dtype] for op in self.ops]

    def hbm_bytes_per_op(self):
        """What each op actually has to move through HBM per image when intermediates of fused ops
        stay on chip (reads of every source + writes of the output, fp32)."""
        out = []
        for op in self.ops:
            rd = sum(s.buf.h * s.buf.w * s.c * rt.ESIZE[s.buf.dtype] for s in op.srcs)
            if op.kind == rt.OP_POINTWISE and op.res is not None:
                rd += op.h * op.w * op.cout * rt.ESIZE[op.res.dtype]
            out.append(rd + op.out.h * op.out.w * op.out.c * rt.ESIZE[op.out.dtype])
        return out

    def algorithmic_bytes_per_image(self):
        return sum(self.algorithmic_bytes_per_op())

    def weight_bytes(self):
        return int(sum(int(np.prod(s)) for s in self.param_shapes.values())) * 4


# Measured on MI355X (profiles/, round 1): the fused kernel beats the three-kernel chain on the
# high-resolution, narrow blocks (MobileNetV2 block_1..5: Cin <= 32), where the expanded tensor is
# 81 % of the traffic; on the deep 26x26 / 13x13 blocks its 8x8 tiles under-fill the chip and the
# unfused GEMMs win, so those stay unfused until the kernel grows a large-M variant.
FUSE_MAX_CIN = int(os.environ.get('YOLORET_FUSE_MAX_CIN', '32'))
FUSE_MIN_PIXELS = int(os.environ.get('YOLORET_FUSE_MIN_PIXELS', '1600'))  # output H*W of the block
FUSE_NO_EXPAND = os.environ.get('YOLORET_FUSE_NO_EXPAND', '0') != '0'    # also fuse DW+project blocks without expand
FUSE_LANE_NO_EXPAND = os.environ.get('YOLORET_FUSE_LANE_NO_EXPAND', '1') != '0'   # ... in the lane-per-pixel kernel (any dtype)
MBLANE_IDENT_WIDTHS = {(4, 16), (6, 24), (8, 32)}     # (Cin quads, padded Cout) of launch_ml_ident
FUSE_STEM = os.environ.get('YOLORET_FUSE_STEM', '1') != '0'              # stem + first (expand-free) block in one kernel
FUSE_LANE = os.environ.get('YOLORET_FUSE_LANE', '1') != '0'              # narrow fused blocks use mblane.hip instead of mbconv.hip
FUSE_LANE_MIN_PIXELS = int(os.environ.get('YOLORET_FUSE_LANE_MIN_PIXELS', '600'))  # mblane still wins on 26x26 outputs (block_6)
MBLANE_WIDTHS = {(4, 16), (4, 24), (6, 24), (6, 32), (6, 40), (6, 48), (8, 32), (8, 40), (8, 48)}  # (CINP/4, round_up(cout,8)) built in mblane.hip
STEMBLOCK_WIDTHS ={(12, 16), (16, 16), (16, 24), (20, 24), (24, 16), (24, 24)}  # (C1p/2, round_up(cout,8)) built in stemblock.hip
STEM_MFMA = os.environ.get('YOLORET_STEM_MFMA', '1') != '0'   # 16-bit plans: the network entry on the matrix pipe (stemblock_h.hip)
# ... for stems of at most 32 channels (MobileNetV2 x0.75 / x1.0, EfficientNet-lite0..2: 0.31 -> 0.22 ms per 128 images at 416).  The
# depthwise stage works on 32-channel k steps: lite3's 40 channels pay for 64 and lose to the float32-pipe kernel (0.27 vs 0.24 ms)
# (round 4: stems of 33..48 channels run on the register-chained form with projection, stemxp_kernel in mbxr_h.hip, same parameter layout)
STEM_MFMA_MAX_C1 = int(os.environ.get('YOLORET_STEM_MFMA_MAX_C1', '48'))


# float32 plans: inverted-residual blocks on the row-walking register-chained matrix-pipe kernel (mbr.hip).  (cin, cexp, cout,
# stride, residual) -> (waves per workgroup, row segments; 0 = the library's choice[, largest output map in pixels]): the shapes built there and measured
# ahead of what the plan would run otherwise (tools/mbr_probe.py, batch 64).  YOLORET_FUSE_MBR=0 switches it off,
# YOLORET_MBR_BLOCKS="block_7,block_8" restricts it to the named blocks.
FUSE_MBR = os.environ.get('YOLORET_FUSE_MBR', '1') != '0'
MBR_BLOCKS = [b for b in os.environ.get('YOLORET_MBR_BLOCKS', '').split(',') if b]
MBR_SHAPES = {
    # stride 2 (paired output rows: even columns in lanes 0..7, odd ones in 8..15, two output rows share one projection)
    (16, 96, 24, 2, False): (2, 4),      # MobileNetV2 x0.75 block_1 (208 x 208 -> 104 x 104): lane kernel 232 us -> 201
    (24, 144, 24, 2, False): (3, 0),     # block_3 (104 x 104 -> 52 x 52): lane kernel 132 us -> 111
    (24, 144, 48, 2, False): (3, 0),     # block_6 (52 x 52 -> 26 x 26): 84 -> 40 us
    (24, 144, 32, 2, False): (3, 0),     # MobileNetV2 x1.4 block_1
    (32, 192, 48, 2, False): (4, 0),     # x1.4 block_3
    (24, 144, 24, 1, True): (3, 0),      # block_4, 5 at 52 x 52 (66 -> 60 us); block_2 at 104 x 104: lane kernel 209 us -> 186 (behind block_1 on mbr; 240 behind the lane kernel)
    (32, 192, 32, 1, True): (4, 0),      # MobileNetV2 x1.4 block_2 (c4 +2.4 % in flight behind block_1 on mbr)
    (48, 288, 48, 1, True): (8, 0),      # block_7..9
    (48, 288, 72, 1, False): (8, 0),     # block_10
}
# The SPLIT form of YR_OP_MBR (mbr.hip SP, round 4): both 1x1 convolutions on the 16-bit matrix pipe, every float32 operand as two
# float16 planes (22 bits; measured error against float64 = the float32-MFMA form's, tools/mbs_probe.py) - the float32 MFMA
# shares the FMA lanes with the depthwise stage, the 16-bit pipe does not.  (cin, cexp, cout, stride, residual) -> waves per
# workgroup the fragments are packed for.  YOLORET_MBR_SPLIT=0 keeps the float32-MFMA form.
MBR_SPLIT = os.environ.get('YOLORET_MBR_SPLIT', '1') != '0'
PW_SPLIT = os.environ.get('YOLORET_PW_SPLIT', '1') != '0'     # (read by the library; here only for the weight-range check of build_blob)
MBS_SHAPES = {
    (16, 96, 24, 2, False): 2, (24, 144, 24, 1, True): 3, (24, 144, 24, 2, False): 3, (24, 144, 48, 2, False): 3,
    (48, 288, 48, 1, True): 6, (48, 288, 72, 1, False): 6,
    (24, 144, 32, 2, False): 3, (32, 192, 32, 1, True): 4, (32, 192, 48, 2, False): 4,     # MobileNetV2 x1.4
}
MBS_MBE_CINS = (48, 72, 88, 120, 136)      # YR_OP_MBE's split form is built for these block input widths
# The WEIGHT-STREAMING form of YR_OP_MBR (mbk.hip, round 6; k bits 6 and 7): the blocks whose fragments do not fit one CU's register
# file as ONE launch - the pixels stay in registers (a wave owns one or two input rows of a 16-column strip and the projection
# accumulators of its output rows), the weights stream through LDS one pair of expanded tiles at a time.  (cin, cexp, cout, stride,
# residual) -> (input rows per wave, waves per workgroup).  Before: YR_OP_MBE + a separate projection launch (the 6x-wide depthwise map
# written and read back: block_11 87 + 100 MB).  YOLORET_FUSE_MBK=0 switches it off.
FUSE_MBK = os.environ.get('YOLORET_FUSE_MBK', '1') != '0'
# the pixel-stationary form of the float32 plans' 1x1 convs (pointwise_stream.hip; se_reduced bit 18, the weights stored as float16 planes)
PW_STREAM = os.environ.get('YOLORET_PW_STREAM', '1') != '0'
PW_STREAM_MID = os.environ.get('YOLORET_PW_STREAM_MID', '0') != '0'     # ... in the 'mid' plan variant (batches below Model.mbk_batch) too
PWT_CHUNKS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16)      # (yr_pwt_chunks)


def pwt_chunks(kp):
    """Chunks of 32 channels the pixel-stationary pointwise form runs a k space of kp channels with (0: it does not take it)."""
    nk = (kp + 31) // 32
    return next((v for v in PWT_CHUNKS if v >= nk), 0)


PW_STREAM_PAIRS = os.environ.get('YOLORET_PW_STREAM_PAIRS', '1') != '0'


def fuse_stream_pairs(ops, output_buf_ids=()):
    """Two pixel-stationary POINTWISE ops over the SAME (gated) map - a head's y conv and the down conv of the bottom-up path (reference
    code/yolo3/model.py:139-151) - become one launch with two outputs (se_reduced bit 19): the map is read once.  The first output is
    the one a plan output aliases (if any); the second travels as gate_out / se_hidden (couts) / reserved0 (activation | pooled << 8),
    its tiles behind the first's in the weight planes, scale and shift padded to the tiles."""
    if not PW_STREAM_PAIRS:
        return ops
    ops = list(ops)

    def conv_dims(o):
        f = 2 if o.stride == 2 else 1
        return o.h * f, o.w * f

    def ok(o):
        return (o.kind == rt.OP_POINTWISE and (o.se_reduced & 0xc0000) == 0x40000 and len(o.srcs) == 1 and o.srcs[0].xform == 'identity'
                and o.gate_out is None)
    i = 0
    while i < len(ops):
        a = ops[i]
        if ok(a):
            for j in range(i + 1, len(ops)):
                b = ops[j]
                if (ok(b) and b.srcs[0].buf is a.srcs[0].buf and b.srcs[0].c == a.srcs[0].c and b.gate is a.gate and conv_dims(a) == conv_dims(b)
                        and b.out is not a.out and b.out.dtype == 0 and a.out.dtype == 0):
                    first, second = (b, a) if (b.out.external_slot >= 0 or b.out.id in output_buf_ids) and not (a.out.external_slot >= 0 or a.out.id in output_buf_ids) else (a, b)
                    if second.out.external_slot >= 0 or second.out.id in output_buf_ids:
                        continue        # (two plan outputs: the second output is an arena buffer)
                    m = OpRec(rt.OP_POINTWISE, first.name, act=first.act, h=first.h, w=first.w, cin=first.cin, cout=first.cout, k=first.k, stride=first.stride,
                              se_reduced=first.se_reduced | 0x80000, srcs=list(first.srcs), out=first.out, gate=first.gate, macs=first.macs + second.macs, dtype=0)
                    m.gate_out, m.se_hidden, m.reserved0 = second.out, second.cout, rt.ACT[second.act] | ((1 if second.stride == 2 else 0) << 8)
                    m.fused = [first, second]
                    m.second_name = second.name
                    if getattr(first, 'folded_projection', None):
                        m.folded_projection = first.folded_projection
                    ta, tb = (first.cout + 15) // 16, (second.cout + 15) // 16
                    fp, sp = first.params, second.params

                    def planes(wd, fp=fp, sp=sp):
                        return np.concatenate([np.asarray(fp['wgt'][1](wd), np.float32).ravel(), np.asarray(sp['wgt'][1](wd), np.float32).ravel()])

                    def vec(role, fp=fp, sp=sp, ta=ta, tb=tb, na=first.cout, nb=second.cout):
                        def f(wd):
                            o = np.full(16 * (ta + tb), 1.0 if role == 'scale' else 0.0, np.float32)      # (a conv without BN scale / bias: 1 / 0)
                            if role in fp:
                                o[:na] = np.asarray(fp[role][1](wd), np.float32).ravel()[:na]
                            if role in sp:
                                o[16 * ta:16 * ta + nb] = np.asarray(sp[role][1](wd), np.float32).ravel()[:nb]
                            return o
                        return f
                    m.params = {'wgt': ((fp['wgt'][0][0] + sp['wgt'][0][0],), planes, 0), 'scale': ((16 * (ta + tb),), vec('scale')), 'shift': ((16 * (ta + tb),), vec('shift'))}
                    ops[i] = m
                    del ops[j]
                    break
        i += 1
    return ops


def pw_stream_form(o):
    """Turn the float32 POINTWISE op o into its pixel-stationary form if it can take it: se_reduced bit 18, 'wgt' as head_pack planes."""
    if (o.kind != rt.OP_POINTWISE or o.dtype != 0 or (o.se_reduced & 0x70000) or any(s_.xform in ('dw3', 'maxpool2', 'maxpool4', 'up2_add') for s_ in o.srcs)
            or o.res is not None or o.act not in ('none', 'relu6')
            or 'wgt' not in o.params or (len(o.params['wgt']) > 2 and o.params['wgt'][2] != 0)):
        return False
    (cout, kp), fn = o.params['wgt'][0], o.params['wgt'][1]
    nk = pwt_chunks(kp)
    if nk > 12:       # (512 channels at 13 x 13: 20 us against the tiled kernel's 18 - one wave's chain over 16 chunks)
        return False
    if kp < 16 or not nk or o.out.dtype != 0 or (o.h == 1 and o.w == 1):
        return False
    nt = (cout + 15) // 16

    def planes(wd, fn=fn, cout=cout, kp=kp, nk=nk):
        return head_pack(np.asarray(fn(wd), np.float32).reshape(cout, kp), [kp], nk=nk)
    o.params = dict(o.params)
    o.params['wgt'] = ((nt * nk * 512,), planes, 0)
    o.se_reduced |= 0x40000
    return True
MBK_SHAPES = {
    (48, 288, 48, 1, True): (2, 8),       # MobileNetV2 x0.75 block_7..9 (26 x 26): 35 us on the weight-stationary form (mbr.hip) -> 24
    (48, 288, 72, 1, False): (2, 8),      # block_10: 39 -> 27 us
    (72, 432, 72, 1, True): (2, 8),       # block_11, 12 (26 x 26): 42 + 34 us (expand + depthwise | projection) -> 46
    (72, 432, 120, 2, False): (2, 8),     # block_13 (26 x 26 -> 13 x 13)
    (120, 720, 120, 1, True): (1, 8),     # block_14, 15 (13 x 13)
}
# float32 plans, blocks too wide for mbr.hip's one-workgroup form: expand + depthwise in one register-chained kernel (YR_OP_MBE),
# the projection stays a pointwise op.  Block input widths built in mbr.hip (MBE_CASE).
FUSE_MBE = os.environ.get('YOLORET_FUSE_MBE', '1') != '0'
# (measured, MobileNetV2 x0.75 @416 batch 64: block_11 / 12 85 -> 60 us, block_13 67 -> 56 us; the 13 x 13 blocks (120 inputs) tie
# at 53 us untuned, 43 us with the tuned row segments; MobileNetV2 x1.4 @512: 88- and 136-wide blocks c4 +4 %, the 224-wide ones -4 %)
MBE_CINS = set(int(v) for v in os.environ.get('YOLORET_MBE_CINS', '48,72,88,120,136').split(',') if v)   # (48: MobileNetV2 x1.4 block_6, 48 -> 288 -> 88 stride 2: c4 +1.7 %)
FUSE_MBH = os.environ.get('YOLORET_FUSE_MBH', '1') != '0'   # 16-bit plans: inverted-residual blocks on the MFMA block kernel (mbh.hip)
MBH_LANE_MIN_PIXELS = int(os.environ.get('YOLORET_MBH_LANE_MIN_PIXELS', '10000'))
MBH_LANE_MAX_CIN = int(os.environ.get('YOLORET_MBH_LANE_MAX_CIN', '16'))
MBN = os.environ.get('YOLORET_MBN', '1') != '0'   # the narrow stride-2 3x3 front block of the 16-bit plans on its own matrix-pipe kernel (mbn_h.hip)
# (kernel size, stride) pairs the fused 16-bit block kernels do NOT take, e.g. '51,52' = 5x5 stride 1 and 2 (A/B runs)
MBH_SKIP = set(os.environ.get('YOLORET_MBH_SKIP', '').replace(' ', '').split(',')) - {''}
MBX_SKIP = set(os.environ.get('YOLORET_MBX_SKIP', '').replace(' ', '').split(',')) - {''}
# 5x5 stride-1 blocks: since the LDS-tiled depthwise (depthwise_lds.hip) and the LDS-tiled pointwise form exist, the unfused
# chain beats the fused block kernel where the block is deep (many 32-channel chunks, each a barrier-synchronous round of a
# small tile): EfficientNet-lite0 stage 5 (26 x 26, 480 / 672 expanded channels) 0.31 -> 0.16 ms per block, the model
# 28.05k -> 29.5k img/s; lite3's fused 5x5 blocks (80 x 80, 288 expanded channels) are 1.3 % better fused.  The squeeze-excite
# form only fuses expand + depthwise, and its 5x5 stride-1 blocks lose to the unfused pair everywhere (B0 19.8k -> 20.7k, B3
# 4.87k -> 4.94k).  Stride-2 5x5 blocks stay fused (lite0 -3.7 % unfused), 3x3 blocks too (lite3 -11 %).
MBH_K5_MAX_CEXP = int(os.environ.get('YOLORET_MBH_K5_MAX_CEXP', '320'))
# Round 3, with the walking depthwise form (depthwise_lds.hip: 52 x 52 x 240 at batch 128 in 0.12 ms, was 0.17): on maps of at
# most MBH_K5_SMALL_MAP pixels the unfused chain also wins from MBH_K5_SMALL_CEXP expanded channels on - lite0 stage 3
# block 1 (52 x 52, 240 expanded): 36.0k -> 37.4k img/s with three steps in flight (serial 32.7k -> 32.9k: the three shorter
# kernels overlap with the other steps' better than one long block kernel); lite3's 80 x 80 x 288 blocks stay fused
# (9.8k -> 9.4k img/s unfused).
MBH_K5_SMALL_MAP = int(os.environ.get('YOLORET_MBH_K5_SMALL_MAP', '4096'))
MBH_K5_SMALL_CEXP = int(os.environ.get('YOLORET_MBH_K5_SMALL_CEXP', '200'))
MBX_K5_MAX_CEXP = int(os.environ.get('YOLORET_MBX_K5_MAX_CEXP', '0'))
MBH_K3_MAX_CEXP = int(os.environ.get('YOLORET_MBH_K3_MAX_CEXP', '1000000'))   # the same switch for 3x3 stride-1 blocks
FUSE_STEMDW = os.environ.get('YOLORET_FUSE_STEMDW', '1') != '0'   # stem + first depthwise of the SE EfficientNets in one kernel
FUSE_STEMDW_MFMA = os.environ.get('YOLORET_FUSE_STEMDW_MFMA', '1') != '0'   # ... on the 16-bit matrix pipe, register-chained (16-bit plans)
FUSE_MBX = os.environ.get('YOLORET_FUSE_MBX', '1') != '0'   # 16-bit plans: expand + depthwise of squeeze-excite MBConv blocks in one kernel
MBH_ACTS = ('relu6', 'swish')   # (swish in the fused 16-bit kernels: hardware exp2 / rcp, no register spills)


HOIST_UPSAMPLE = os.environ.get('YOLORET_HOIST', '1') != '0'


def hoist_upsampled_sources(ops, bufs):
    """A 1x1 convolution commutes with nearest-neighbour upsampling: W.[up2(a); b] = up2(Wa.a) + Wb.b.  The FPN
    top-down convs (td2_conv, td3_conv: model.py:253-255,273-275 feed `Concatenate([UpSampling2D(x), b])` into the
    first 1x1 conv of make_last_layers) take 60-90 % of their input channels from 2x-upsampled maps, so that share
    of the GEMM is computed at the SOURCE resolution (a quarter of the pixels) into a small buffer P, and the
    full-resolution conv keeps only the remaining sources and adds up2(P) to its accumulator before BatchNorm
    (source xform 'up2_add').  Same mathematics, a different fp32 summation grouping (inside the 1e-4 logit bar);
    td3_conv drops from 248 to 24 input channels at 52x52, td2_conv from 424 to 168 at 26x26."""
    out = []
    for op in ops:
        lo = [s for s in op.srcs if s.xform == 'up2']
        hi = [s for s in op.srcs if s.xform != 'up2']
        ok = (op.kind == rt.OP_POINTWISE and op.gate is None and lo and hi and len(hi) + 1 <= rt.YR_MAX_SRC
              and len(set((s.buf.h, s.buf.w) for s in lo)) == 1 and not (op.h == 1 and op.w == 1)
              and sum(s.c for s in lo) >= sum(s.c for s in hi))
        if not ok:
            out.append(op)
            continue
        V = rt.VEC[op.dtype]
        pads = [round_up(s.c, V) for s in op.srcs]
        base = [sum(pads[:i]) for i in range(len(pads))]
        lo_cols = [(base[i], pads[i]) for i, s in enumerate(op.srcs) if s.xform == 'up2']
        hi_cols = [(base[i], pads[i]) for i, s in enumerate(op.srcs) if s.xform != 'up2']
        wshape, wfn = op.params['wgt'][:2]
        h_lo, w_lo = lo[0].buf.h, lo[0].buf.w
        # (float32 in every plan: a pre-BatchNorm partial sum, rounding it to 16 bits would cost the conv its accuracy)
        p = Buf(len(bufs), h_lo, w_lo, op.cout, round_up(op.cout, 4), name=op.name + '_lowres', dtype=0)
        bufs.append(p)

        def cols(sel, wfn=wfn):
            return lambda wd: np.ascontiguousarray(np.concatenate([wfn(wd)[:, b:b + n] for b, n in sel], axis=1))
        low = OpRec(rt.OP_POINTWISE, op.name + '_lowres', act='none', h=h_lo, w=w_lo, cin=sum(s.c for s in lo),
                    cout=op.cout, srcs=[Seg(s.buf, s.c, 'identity') for s in lo], out=p, macs=0, dtype=op.dtype)
        low.params = {'wgt': ((op.cout, sum(n for _, n in lo_cols)), cols(lo_cols), op.dtype)}
        low.accounted_in = op.name     # its traffic and MACs are part of `op` in the conv-granular accounting
        top = OpRec(rt.OP_POINTWISE, op.name, act=op.act, h=op.h, w=op.w, cin=sum(s.c for s in hi), cout=op.cout,
                    srcs=hi + [Seg(p, op.cout, 'up2_add')], out=op.out, res=op.res, macs=op.macs, dtype=op.dtype)
        top.params = dict(op.params)
        top.params['wgt'] = ((op.cout, sum(n for _, n in hi_cols)), cols(hi_cols), op.dtype)
        top.accounting_srcs = list(op.srcs)   # SURVEY 8(d) charges the conv its original (concatenated) input
        out += [low, top]
    return out


# Two 1x1 convolutions with nothing non-linear between them are one: the projection of the detection heads' MBConv block has
# BatchNorm but no activation (efficientnet.py:517-533), and what reads its 75-channel output are 1x1 convolutions again
# (the `y` conv of make_last_layers, model.py:110-114; the bottom-up 75 -> 128 conv + maxpool, :298-308; the next block's
# first conv).  W_c (s_p * W_p d + h_p) = (W_c diag(s_p) W_p) d + W_c h_p: each consumer reads the gated depthwise map itself
# with the composed weights (products formed in float64, rounded once), the projection launch and its 75-channel tensor
# disappear.  Done where it does not cost arithmetic: composed MACs <= FOLD_PROJ_MAX_RATIO x the MACs of the convs replaced
# (the 52 x 52 heads, F = 128: td3 -> bu3_conv 0.85, bu3 -> {y, down} 1.05; bu1 -> y 0.87; not the F = 256 / 512 ones).
FOLD_PROJ = os.environ.get('YOLORET_FOLD_PROJ', '1') != '0'
FOLD_PROJ_MAX_RATIO = float(os.environ.get('YOLORET_FOLD_PROJ_MAX_RATIO', '1.1'))


def fold_projection_into_consumers(ops, output_buf_ids):
    readers = {}
    for op in ops:
        for s in op.srcs:
            readers.setdefault(id(s.buf), []).append((op, s))
        for b in (op.res, op.gate):
            if b is not None:
                readers.setdefault(id(b), []).append((op, None))
    drop, repl = set(), {}
    for P in ops:
        rd = readers.get(id(P.out), [])
        # (a chain of linear 1x1 convs P1 -> C1 -> C2: once P1 is folded into C1, the ORIGINAL C1 must not be folded into C2 as a
        # projection of its own - its replacement reads P1's source, the original read P1's dropped output)
        if id(P) in repl or id(P) in drop:
            continue
        if (P.kind != rt.OP_POINTWISE or P.act != 'none' or 'scale' not in P.params or P.res is not None or len(P.srcs) != 1
                or P.srcs[0].xform != 'identity' or P.out.external_slot >= 0 or P.out.id in output_buf_ids or not rd
                or (P.h == 1 and P.w == 1) or getattr(P, 'accounted_in', None)):
            continue
        if any(s is None or C.kind != rt.OP_POINTWISE or len(C.srcs) != 1 or s.xform != 'identity' or C.gate is not None
               or C.res is not None or s.c != P.cout or C.dtype != P.dtype or id(C) in repl or id(C) in drop for C, s in rd):
            continue
        cin, cp = P.cin, P.cout
        if sum(cin * C.cout for C, _ in rd) > FOLD_PROJ_MAX_RATIO * (cin * cp + sum(cp * C.cout for C, _ in rd)):
            continue
        pw, psc, psh = P.params['wgt'][1], P.params['scale'][1], P.params['shift'][1]
        for n, (C, s) in enumerate(rd):
            m = OpRec(rt.OP_POINTWISE, C.name, act=C.act, h=C.h, w=C.w, cin=cin, cout=C.cout, srcs=[Seg(P.srcs[0].buf, P.srcs[0].c, 'identity')],
                      out=C.out, gate=P.gate, macs=C.macs + (P.macs if n == 0 else 0), dtype=P.dtype)
            cw = C.params['wgt'][1]
            csc = C.params['scale'][1] if 'scale' in C.params else None
            csh = C.params['shift'][1] if 'shift' in C.params else None

            def wgt(wd, pw=pw, psc=psc, cw=cw, cp=cp):
                return (cw(wd)[:, :cp].astype(np.float64) @ (psc(wd)[:cp, None].astype(np.float64) * pw(wd)[:cp].astype(np.float64))).astype(np.float32)

            def shift(wd, psh=psh, cw=cw, csc=csc, csh=csh, cp=cp):
                b = cw(wd)[:, :cp].astype(np.float64) @ psh(wd)[:cp].astype(np.float64)
                if csc is not None:
                    b = csc(wd).astype(np.float64) * b + csh(wd).astype(np.float64)
                return b.astype(np.float32)
            m.params = {'wgt': (P.params['wgt'][0][:0] + (C.cout, P.params['wgt'][0][1]), wgt, P.dtype),   # (16-bit plans: ONE rounding of the composed matrix)
                        'scale': ((C.cout,), csc if csc is not None else (lambda wd, n_=C.cout: np.ones(n_, np.float32))),
                        'shift': ((C.cout,), shift)}
            # conv-granular accounting stays that of the convolutions replaced: the first consumer carries the
            # projection's share, every consumer is charged the 75-channel input it used to read
            m.fused = [P, C] if n == 0 else [C]
            m.folded_projection = P.name
            repl[id(C)] = m
        drop.add(id(P))
    return [repl.get(id(op), op) for op in ops if id(op) not in drop]


# RFCR's weighted sum (model.py:117-137, 146-168) adds four 48-channel maps, three of which are LINEAR 1x1 convolutions (no
# BatchNorm, no activation) of backbone features: a0 * up2(W1 x1) + a1 * W2 x2 + a2 * maxpool2(W3 x3) + a3 * W4 maxpool4(x4).
# A nearest-neighbour upsample commutes with a 1x1 convolution and so does the scalar, so the three linear terms are ONE
# pointwise convolution over the concatenation [up2(x1) | x2 | maxpool4(x4)] with the weight matrix [a0 W1 | a1 W2 | a3 W4]; the
# term behind the max-pool (not linear: the alphas are unconstrained, a2 may be negative) rides along as a fourth source with the
# weight block a2 * I.  Four launches (three convs + the sum; 62 us of the MobileNetV2 x0.75 step at batch 64) become one
# conv at 26 x 26, and three 48-channel maps are never written.  Rounding differs from the reference's order of operations
# like any fp32 re-association (the 1e-4 logit bar holds; YOLORET_FOLD_WSUM=0 keeps the literal chain, which the op-level
# bit-exactness test of the WeightedSum kernel uses).
FOLD_WSUM = os.environ.get('YOLORET_FOLD_WSUM', '1') != '0'


def fold_weighted_sum(ops, output_buf_ids, V=4):
    producer = {id(op.out): op for op in ops}
    nreaders = {}
    for op in ops:
        for sg in op.srcs:
            nreaders[id(sg.buf)] = nreaders.get(id(sg.buf), 0) + 1
        for b in (op.res, op.gate):
            if b is not None:
                nreaders[id(b)] = nreaders.get(id(b), 0) + 1
    drop, repl = set(), {}
    for W in ops:
        if W.kind != rt.OP_WSUM or len(W.srcs) != 4 or W.out.id in output_buf_ids:
            continue
        segs, parts, folded = [], [], []   # parts: (alpha index, weight function or None = identity block, columns)
        for i, sg in enumerate(W.srcs):
            P = producer.get(id(sg.buf))
            ok = (P is not None and P.kind == rt.OP_POINTWISE and P.act == 'none' and 'scale' not in P.params and P.res is None
                  and P.gate is None and getattr(P, 'stride', 0) != 2 and nreaders.get(id(P.out), 0) == 1 and P.out.external_slot < 0
                  and P.out.id not in output_buf_ids and P.dtype == W.dtype and sg.c == P.cout and not getattr(P, 'accounted_in', None)
                  and (sg.xform == 'identity' or (sg.xform == 'up2' and all(q.xform == 'identity' for q in P.srcs))))
            if ok:
                segs += [Seg(q.buf, q.c, 'up2' if sg.xform == 'up2' else q.xform) for q in P.srcs]
                parts.append((i, P.params['wgt'][1], sum(round_up(q.c, V) for q in P.srcs)))
                folded.append(P)
            else:
                segs.append(Seg(sg.buf, sg.c, sg.xform))
                parts.append((i, None, round_up(sg.c, V)))
        if len(folded) < 2 or len(segs) > rt.YR_MAX_SRC or any(sg.xform not in ('identity', 'up2', 'maxpool2', 'maxpool4') for sg in segs):
            continue
        cout, kp = W.cout, sum(p[2] for p in parts)
        m = OpRec(rt.OP_POINTWISE, W.name, act='none', h=W.h, w=W.w, cin=sum(sg.c for sg in segs), cout=cout, srcs=segs, out=W.out,
                  macs=sum(P.macs for P in folded), dtype=W.dtype)
        alpha = W.params['wgt'][1]

        def wgt(wd, parts=parts, alpha=alpha, cout=cout, kp=kp):
            a = np.asarray(alpha(wd), np.float64)
            o, kb = np.zeros((cout, kp), np.float64), 0
            for i, wf, cols in parts:
                if wf is None:
                    o[:, kb:kb + cout] = a[i] * np.eye(cout)
                else:
                    o[:, kb:kb + cols] = a[i] * np.asarray(wf(wd), np.float64)[:cout]
                kb += cols
            return o.astype(np.float32)
        m.params = {'wgt': ((cout, kp), wgt, W.dtype)}
        m.fused = folded + [W]        # conv-granular accounting: the convolutions and the sum it replaces
        m.folded_wsum = True
        repl[id(W)] = m
        drop.update(id(P) for P in folded)
    return [repl.get(id(op), op) for op in ops if id(op) not in drop]


MERGE_SE_MEAN = os.environ.get('YOLORET_MERGE_SE_MEAN', '1') != '0'


def merge_se_mean(ops, only_after_depthwise=False):
    """SE's reduce_mean and its FC pair are two latency-bound launches of one workgroup per image; when the mean
    feeds nothing but the FCs the SE_FC op takes the full map as its source and pools it itself (a fixed summation
    order of its own): six launches fewer per MBV2 step.  only_after_depthwise (the small-batch plan): merge only where
    se_partials_from_depthwise will then turn the pooling into partial sums of the depthwise kernel - a merged launch
    that pools a whole map in one workgroup is what that plan avoids."""
    producer = {id(op.out): op for op in ops}
    readers = {}
    for op in ops:
        for s in op.srcs:
            readers.setdefault(id(s.buf), []).append(op)
        for b in (op.res, op.gate):
            if b is not None:
                readers.setdefault(id(b), []).append(op)
    drop = set()
    for op in ops:
        if op.kind != rt.OP_SE_MEAN or op.out.external_slot >= 0:
            continue
        rd = readers.get(id(op.out), [])
        if only_after_depthwise:
            d = producer.get(id(op.srcs[0].buf))
            if d is None or d.kind != rt.OP_DEPTHWISE or d.gate is not None or d.k not in (3, 5) or d.stride not in (1, 2):
                continue
        if len(rd) == 1 and rd[0].kind == rt.OP_SE_FC and len(rd[0].srcs) == 1 and rd[0].srcs[0].buf is op.out:
            fc = rd[0]
            fc.srcs = [Seg(op.srcs[0].buf, op.srcs[0].c, 'identity')]
            fc.merged_mean = op.cout     # the accounting still charges the mean its C outputs
            drop.add(id(op))
    return [op for op in ops if id(op) not in drop]


SE_PARTIALS = os.environ.get('YOLORET_SE_PARTIALS', '1') != '0'


def dw_se_geometry(strips, c4):
    """== dw_se_geometry() in depthwise.hip: (channel vectors per workgroup, workgroups per strip group, rows of the
    partial-sum buffer) of the squeeze-excite form of the depthwise kernel."""
    cw = 32
    if c4 <= 256 and (256 // c4) * c4 * ((c4 + 31) // 32) * 32 >= c4 * 256:
        cw = c4        # all channel vectors of a strip in one workgroup keeps at least as many lanes busy as 32-vector blocks
    return cw, (c4 + cw - 1) // cw, (strips + 256 // cw - 1) // (256 // cw)


DW_LDS = os.environ.get('YOLORET_DW_LDS', '1') != '0'


DW_WALK = int(os.environ.get('YOLORET_DW_WALK', '1'))       # 0: the tile walk everywhere


def dwl_geometry(h, w, k=5, se=True):
    """(rows along x, rows along y) of the squeeze-excite sums the 16-bit k x k stride-1 depthwise form writes per image.
    The walking form (depthwise_walk.hip, == dwq_geometry() / dwq_quanta() there): column blocks of at most 8 four-column
    strips x row quanta fixed by the map's height - independent of the batch and of how a launch cuts the rows into segments."""
    if DW_WALK and k == 5:     # launch_depthwise_t's choice (depthwise.hip)
        strips = (w + 3) // 4
        q = min(8, max(1, h // 10))
        rows = round_up((h + q - 1) // q, k)      # quanta of a multiple of k rows
        return (strips + 7) // 8, (h + rows - 1) // rows
    return dwp_geometry(h, w, k, se)


def dwp_geometry(h, w, k=5, se=True):
    """== dwp_geometry() in depthwise_lds.hip (integer arithmetic, the same choice): (tiles along x, tiles along y) of the
    LDS-tiled k x k stride-1 depthwise form; its squeeze-excite variant (se: two tile buffers of at most 192 halo pixels,
    208 without the partial sums behind them) writes one row of channel sums per tile."""
    halo = k - 1
    cap = 192 if se else 208
    row_cost, in_cost, round_cost, tile_cost = k * k * 4 + 21, 3 * (4 + halo), 12, 60
    best = None
    for tw in range(8, 33, 8):
        cols, nstrip = tw + halo, tw // 4
        th = cap // cols - halo
        if th < 1:
            continue
        th = min(th, h)
        nty = (h + th - 1) // th
        th = (h + nty - 1) // nty
        ntx = (w + tw - 1) // tw
        nband = min(8 // nstrip, th)
        band_rows = (th + nband - 1) // nband
        if band_rows > 6:
            continue
        rounds = ((th + halo) * cols + 31) // 32
        cost = ntx * nty * (row_cost * band_rows + in_cost * (band_rows + halo) + round_cost * rounds + tile_cost)
        if best is None or cost < best[0]:
            best = (cost, ntx, nty)
    return best[1], best[2]


def dw_uses_lds_form(d):
    """launch_depthwise_t's choice (depthwise.hip): 16-bit 5x5 or 3x3, stride 1, with at least one 64-channel chunk."""
    return DW_LDS and d.dtype != 0 and d.k in (3, 5) and d.stride == 1 and d.cout >= 64


def se_partials_from_depthwise(ops, bufs):
    """the squeeze of squeeze-excite (tf.reduce_mean over H, W; efficientnet.py:417) as an epilogue of
    the depthwise conv that produces the map.  The SE_FC op with the merged mean re-reads the whole map for it (one
    workgroup per image: 54 MB per launch on the 52x52 head block); here every workgroup of the depthwise kernel adds up
    the outputs it has just computed, per channel, in a fixed order, and writes one float32 row; SE_FC adds the rows up
    and divides by the pixel count."""
    producer = {}
    for op in ops:
        producer[id(op.out)] = op
    for fc in ops:
        if fc.kind != rt.OP_SE_FC or not getattr(fc, 'merged_mean', 0) or len(fc.srcs) != 1:
            continue
        d = producer.get(id(fc.srcs[0].buf))
        if d is None or d.kind != rt.OP_DEPTHWISE or d.gate is not None or d.k not in (3, 5) or d.stride not in (1, 2):
            continue
        v = rt.VEC[d.dtype]
        c4 = (d.cout + v - 1) // v
        xt = 4 if d.stride == 1 else 2
        rows = dw_se_geometry(d.h * ((d.w + xt - 1) // xt), c4)[2]
        if dw_uses_lds_form(d):
            ntx, nty = dwl_geometry(d.h, d.w, d.k)
            rows = ntx * nty
        part = Buf(len(bufs), rows, 1, d.cout, round_up(d.cout, v), name=d.name + ':se_sums', dtype=0)
        bufs.append(part)
        d.gate, d.se_reduced = part, rows
        fc.srcs = [Seg(part, d.cout, 'identity')]
        fc.k = d.h * d.w                                                 # what the summed rows are divided by
    return ops


POOL_IN_PRODUCER = os.environ.get('YOLORET_POOL_FUSE', '1') != '0'


def pool_into_producers(ops, bufs, output_buf_ids):
    """`downsample_layer` = MaxPooling2D(2) (model.py:139-144) sits right after the bottom-up 1x1 convs and RFCR's
    b3 conv, and its input has no other reader.  Instead of writing the full-resolution map and letting the consumer
    take the maximum of four loads per element, the producing pointwise op stores the pooled map itself (op.stride =
    2: GEMM rows in 2x2-quad-major order, window maximum across four lanes): a quarter of the bytes written, a
    quarter read.  Arithmetic is unchanged (max of the same finished values)."""
    readers = {}
    for op in ops:
        for s in op.srcs:
            readers.setdefault(id(s.buf), []).append((op, s))
        for b in (op.res, op.gate):
            if b is not None:
                readers.setdefault(id(b), []).append((op, None))
    for op in ops:
        rd = readers.get(id(op.out), [])
        if (op.kind != rt.OP_POINTWISE or op.res is not None or op.out.external_slot >= 0 or op.out.id in output_buf_ids
                or not rd or any(s is None or s.xform != 'maxpool2' for _, s in rd) or op.h % 2 or op.w % 2
                or any(s.xform == 'up2_add' for s in op.srcs) or getattr(op, 'stride', 0) == 2):
            continue
        q = Buf(len(bufs), op.h // 2, op.w // 2, op.out.c, op.out.ld, name=op.out.name + '_pooled', dtype=op.out.dtype)
        bufs.append(q)
        op.accounting_hw = q.accounting_hw = (op.h, op.w)   # SURVEY 8(d) charges the conv its full-resolution output
        op.out, op.h, op.w, op.stride = q, op.h // 2, op.w // 2, 2
        for _, s in rd:
            s.buf, s.xform = q, 'identity'
    return ops


# Off by default (measured, MBV2x0.75@416 batch 64): the folded projection equals depthwise + projection on the 26x26
# blocks (0.046 vs 0.048 ms, 0.0735 vs 0.075 ms) and loses on the 13x13 ones (0.081 vs 0.056 ms): a 64-row tile per
# workgroup leaves the depthwise arithmetic to 2-3 workgroups per CU where dw_kernel spreads it over 8 waves per SIMD.
# 21.4k vs 21.9k img/s end to end.  On maps of 32 x 32 and more it wins (MobileNetV2 x1.4 @512, batch 64: block_4/5 at
# 64 x 64: 0.233 vs 0.193 + 0.118 ms; block_7..9 at 32 x 32: 0.14 vs 0.16 ms; 8.70k -> 8.98k img/s): folded by default
# where the projection's map has at least FOLD_DW_MIN_PIXELS pixels.  YOLORET_FOLD_DW = 0: never, 1: every eligible block.
FOLD_DW = os.environ.get('YOLORET_FOLD_DW', 'auto')
FOLD_DW_MIN_PIXELS = int(os.environ.get('YOLORET_FOLD_DW_MIN_PIXELS', '1024'))
FOLD_DW_MAX_COUT = int(os.environ.get('YOLORET_FOLD_DW_MAX_COUT', '128'))   # one cout tile: the depthwise work is done once
FOLD_DW_MAX_C = 1088                                                       # 11 * round_up(C,4) floats of LDS <= 48 KB


def fold_depthwise_into_project(ops, output_buf_ids, min_pixels=0):
    """DEPTHWISE 3x3 (+BN+act) whose only reader is a plain POINTWISE projection: the projection reads the
    depthwise INPUT through xform 'dw3' and computes the depthwise stage in its loader (pointwise_lds.hip, PwDwRow),
    bit-identically; the depthwise output - as wide as the expand output - never reaches HBM.  Only where the
    projection has a single cout tile (cout <= FOLD_DW_MAX_COUT), so the depthwise work is not repeated."""
    readers = {}
    for op in ops:
        for s in op.srcs:
            readers[s.buf.id] = readers.get(s.buf.id, 0) + 1
        for b in (op.res, op.gate):
            if b is not None:
                readers[b.id] = readers.get(b.id, 0) + 1
    out, i = [], 0
    while i < len(ops):
        d = ops[i]
        p = ops[i + 1] if i + 1 < len(ops) else None
        if (p is not None and d.kind == rt.OP_DEPTHWISE and d.dtype == 0 and d.k == 3 and d.stride in (1, 2)
                and len(d.srcs) == 1 and d.srcs[0].xform == 'identity' and d.srcs[0].c == d.srcs[0].buf.c
                and p.kind == rt.OP_POINTWISE and len(p.srcs) == 1 and p.srcs[0].buf is d.out
                and p.srcs[0].xform == 'identity' and p.gate is None and not getattr(p, 'stride', 0)
                and readers.get(d.out.id, 0) == 1 and d.out.id not in output_buf_ids
                and p.cout <= FOLD_DW_MAX_COUT and d.cout <= FOLD_DW_MAX_C and 'scale' in p.params
                and p.h * p.w >= min_pixels):
            f = OpRec(rt.OP_POINTWISE, p.name, act=p.act, h=p.h, w=p.w, cin=p.cin, cout=p.cout,
                      srcs=[Seg(d.srcs[0].buf, d.srcs[0].c, 'dw3')], out=p.out, res=p.res,
                      se_reduced=d.stride | (rt.ACT[d.act] << 8), macs=p.macs + d.macs)
            f.params = dict(p.params)
            f.params['wgt2'], f.params['b1'], f.params['b2'] = d.params['wgt'], d.params['scale'], d.params['shift']
            f.fused = [d, p]
            out.append(f)
            i += 2
            continue
        out.append(d)
        i += 1
    return out


# ---- ABI 7: detection-head blocks as ONE launch, squeeze-excite finished by its producer ------------------------------------
# make_last_layers_efficientnet_lite (model.py:91-115) = Conv2D 1x1 + BN + ReLU6 -> MBConvBlock (expand ratio 1: depthwise 3x3 + BN +
# Swish -> SE -> project).  Through round 4 the first two thirds were three launches (split pointwise GEMM, dw_kernel, se_fc) and the
# F-wide conv output (F = 128 @52x52, 256 @26x26, 512 @13x13) went to HBM and back.  fuse_head_blocks turns conv + depthwise into a
# YR_OP_HEAD op (headblock.hip: the conv over a region with halo, its output kept in LDS, depthwise from there, per-region channel
# sums); se_tail_into_producers then hands the SE block's FC pair to whichever op writes the sums (HEAD, or a DEPTHWISE op in the SE
# form of dw_kernel): the workgroup that completes an image runs it (se_tail.h) and the SE_FC launch disappears.
FUSE_HEAD = os.environ.get('YOLORET_FUSE_HEAD', '1') != '0'
FUSE_HEAD_ALL = os.environ.get('YOLORET_FUSE_HEAD', '1') == '2'     # also conv -> depthwise pairs without squeeze-excite sums
KSPLIT_MAX_PIXELS = int(os.environ.get('YOLORET_KSPLIT_MAX_PIXELS', str(32 * 32)))   # maps (conv pixels per image) up to which the 'nohead_k' plan's pointwise convs run the k-split form (0: off; 52 x 52 maps lose)
SE_TAIL = os.environ.get('YOLORET_SE_TAIL', '0') != '0'   # OPT-IN: correct in one stream, not with steps in flight on several (se_tail.h: STATUS)
SE_TAIL_LDS = 4608 - 1024 - 4      # == YR_SE_TAIL_LDS - 4 * 256 threads (se_tail.h): channels + hidden units the tail's LDS scratch holds
HEAD_WALK_MAX_NK = min(7, int(os.environ.get('YOLORET_HEAD_WALK_MAX_NK', '4')))   # (measured, MobileNetV2 x0.75 @416 batch 64: 1 chunk 52 us against the LDS-direct kernel's 95, 4 chunks 96 | 124, 6 chunks 85 | 82, 7 chunks 96 | 84: one tile per wave and 250 registers from 5 chunks on)
HEAD_WALK = os.environ.get('YOLORET_HEAD_WALK', '1') != '0'   # head blocks of at most 7 chunks of 32 identity-source channels on the walking kernel (headwalk.hip)
HEAD_DMA = os.environ.get('YOLORET_HEAD_DMA', '1') != '0'   # head blocks without a pooled source on the LDS-direct kernel
# Round 6: the WEIGHT-STREAMING form of YR_OP_HEAD (headstream.hip, k bits 5 and 6; mbk.hip's formulation - the pixels of a wave's one or
# two rows stationary over the whole k space, the conv's output channels streaming past them in pairs of tiles): identity and 2 x 2
# max-pooled sources (gathered once), up to 11 chunks of 32 channels at one row per wave (maps below 20 rows), 7 at two.
# YOLORET_HEAD_STREAM=0 switches it off, a comma list of block names ('td2,bu2') restricts it.
_hs = os.environ.get('YOLORET_HEAD_STREAM', '1')
HEAD_STREAM = _hs != '0'
HEAD_STREAM_ONLY = [b for b in _hs.split(',') if b and b not in ('0', '1')]


def head_stream_geometry(h, w):
    """== hs_geometry (headstream.hip): (rows per wave, waves per workgroup, strips, row segments) of the weight-streaming head form."""
    rpw = 2 if h >= 20 else 1
    nw = 8
    nr = nw * rpw
    return rpw, nw, (w + 13) // 14, 1 if h <= nr else (h - nr + nr - 3) // (nr - 2) + 1


def head_stream_rows(h, w):
    """== yr_head_stream_rows: rows of squeeze-excite sums that form writes per image - one per (strip, segment, wave)."""
    rpw, nw, strips, segs = head_stream_geometry(h, w)
    return strips * segs * nw


def head_regions(h, w):
    """== head_geometry() in headblock.hip (yr_head_regions): (nsy, nsx) regions a YR_OP_HEAD launch cuts an h x w map into - a
    region with its one-pixel halo (clipped to the map) fits the 192 GEMM rows of a workgroup; fewest regions wins."""
    best = None
    for sx in range(1, min(16, w) + 1):
        cw = (w + sx - 1) // sx
        rw = cw + (2 if sx > 2 else 1 if sx > 1 else 0)
        for sy in range(1, h + 1):
            ch = (h + sy - 1) // sy
            rh = ch + (2 if sy > 2 else 1 if sy > 1 else 0)
            if min(rh, h) * min(rw, w) > 192:
                continue
            if best is None or sy * sx < best[0]:
                best = (sy * sx, sy, sx)
            break
    if best is None:
        raise ValueError('head_regions: a %d x %d map has no region split' % (h, w))
    return best[1], best[2]


def head_walk_rows(h, w):
    """== yr_head_walk_rows (headwalk.hip): rows of the squeeze-excite sums the walking form of YR_OP_HEAD writes per image -
    strips of 14 columns x row segments of about 13 rows (a function of the shape)."""
    if h <= 16:
        sr = h
    else:
        ns = (h + 12) // 13
        sr = (h + ns - 1) // ns
    return ((w + 13) // 14) * ((h + sr - 1) // sr)


def head_pack(wt, seg_c, V=4, nk=None):
    """(nk: pad the chunk list with zero chunks to this length - the pixel-stationary pointwise form is built for some chunk counts only.)
    The 1x1 convolution's weights Wt [F][kp] (k space = the sources' channels, each padded to V) as the float16 planes the
    LDS-direct head kernel reads (headblock.hip, YR_OP_HEAD with k bit 7): the k space cut into chunks of 32 channels PER SOURCE,
    [ceil(F / 16)][NK][2 planes][64 lanes][8 halves] - lane (m = l % 16, g = l / 16) of cout tile t, chunk j of source s:
    W[16 t + m][channel 32 j + 8 g + i of s], zero beyond the source / beyond F; h plane, then m = f16((w - h) 2^11).
    -> the float32 words that hold them."""
    wt = np.asarray(wt, np.float32)
    F = wt.shape[0]
    NT = (F + 15) // 16
    chunks, kb = [], 0
    for c in seg_c:
        chunks += [(kb + 32 * j, min(32, c - 32 * j)) for j in range((c + 31) // 32)]
        kb += round_up(c, V)
    assert float(np.abs(wt).max()) < 60000.0, 'head_pack: a weight beyond the float16 range'
    nch = len(chunks) if nk is None else nk
    assert nch >= len(chunks)
    W = np.zeros((NT * 16, nch, 32), np.float32)
    for ci, (k0, vc) in enumerate(chunks):
        W[:F, ci, :vc] = wt[:, k0:k0 + vc]
    W = W.reshape(NT, 16, nch, 4, 8).transpose(0, 2, 3, 1, 4)          # [t][chunk][g][m][i]  (lane = 16 g + m)
    h = W.astype(np.float16)
    m = ((W - h.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    out = np.stack([h, m], axis=2)                                             # [t][chunk][plane][g][m][i]
    return np.ascontiguousarray(out).reshape(-1).view(np.float32)


def head_pack16(wt, seg_c, V=8):
    """The 1x1 convolution's weights Wt [F][kp] (k space = the sources' channels, each padded to V = 8) in the fragment order of the
    16-bit walking head kernel (headwalk_h.hip): [F / 16][NK][64 lanes][8] with the k space cut into chunks of 32 channels PER SOURCE -
    lane (m = l % 16, g = l / 16) of cout tile t, chunk j of source s: W[16 t + m][channel 32 j + 8 g + i of s], zero beyond the
    source.  float32 values; Plan.build_blob rounds them to the plan's 16-bit type (they are the conv's weights as they are: the BN
    scale stays a float32 vector)."""
    wt = np.asarray(wt, np.float32)
    F = wt.shape[0]
    assert F % 16 == 0
    chunks, kb = [], 0
    for c in seg_c:
        chunks += [(kb + 32 * j, min(32, c - 32 * j)) for j in range((c + 31) // 32)]
        kb += round_up(c, V)
    W = np.zeros((F, len(chunks), 32), np.float32)
    for ci, (k0, vc) in enumerate(chunks):
        W[:, ci, :vc] = wt[:, k0:k0 + vc]
    W = W.reshape(F // 16, 16, len(chunks), 4, 8).transpose(0, 2, 3, 1, 4)          # [t][chunk][g][m][i]  (lane = 16 g + m)
    return np.ascontiguousarray(W).reshape(-1)


def _head_block16(c, d, readers, output_buf_ids):
    """A 16-bit plan's head block (conv 1x1 -> depthwise 3x3 with squeeze-excite sums) as YR_OP_HEAD in the walking form
    (headwalk_h.hip), or None: identity sources of the plan's type (+ a float32 up2_add addend), at most HEAD_WALK16_MAX_NK chunks of
    32 channels, F a multiple of 128 (four waves x two cout tiles per workgroup)."""
    if d is None or not FUSE_HEAD or not HEAD_WALK:
        return None
    kseg = [s_.c for s_ in c.srcs if s_.xform != 'up2_add']
    nk = sum((c_ + 31) // 32 for c_ in kseg)
    F = c.cout
    ok = (c.act in ('relu6', 'none') and 'scale' in c.params and c.res is None and not getattr(c, 'stride', 0) and not getattr(c, 'accounted_in', None)
          and all(s_.xform in ('identity', 'up2_add') for s_ in c.srcs) and all(s_.buf.dtype == c.dtype and s_.buf.ld % 8 == 0 for s_ in c.srcs if s_.xform != 'up2_add')
          and all(s_.buf.dtype == 0 for s_ in c.srcs if s_.xform == 'up2_add') and sum(s_.xform == 'up2_add' for s_ in c.srcs) <= 1
          and (not c.srcs or c.srcs[-1].xform == 'up2_add' or all(s_.xform == 'identity' for s_ in c.srcs))
          and 1 <= len(kseg) <= 3 and nk <= HEAD_WALK16_MAX_NK and F % 128 == 0 and F * 11 * 4 <= 64 * 1024
          and (c.gate is None or (len(c.srcs) == 1 and c.srcs[0].xform == 'identity'))
          and c.out.external_slot < 0 and c.out.id not in output_buf_ids and readers.get(id(c.out), 0) == 1 and c.out.dtype == c.dtype
          and d.kind == rt.OP_DEPTHWISE and d.dtype == c.dtype and d.k == 3 and d.stride == 1 and len(d.srcs) == 1 and d.srcs[0].buf is c.out
          and d.srcs[0].xform == 'identity' and d.srcs[0].c == F and d.act in ('swish', 'relu6', 'none') and d.out.ld % 4 == 0 and d.out.dtype == c.dtype
          and d.res is None and d.gate_out is None and not (c.h == 1 and c.w == 1) and (d.gate is not None or FUSE_HEAD_ALL))
    if not ok:
        return None
    m = OpRec(rt.OP_HEAD, c.name.rsplit('_', 1)[0] + '_head', act=d.act, h=d.h, w=d.w, cin=c.cin, cout=F, k=3 | rt.ACT[c.act] << 8 | 0x40, stride=1,
              srcs=list(c.srcs), out=d.out, res=c.gate, macs=c.macs + d.macs, dtype=c.dtype)
    m.fused = [c, d]
    if getattr(c, 'folded_projection', None):
        m.folded_projection = c.folded_projection
    cp, dp = c.params, d.params

    def frags(wd, cp=cp, kseg=kseg, F=F):
        return head_pack16(np.asarray(cp['wgt'][1](wd), np.float32)[:F], kseg)

    def tab(wd, cp=cp, dp=dp, F=F):
        T = F // 16
        o = np.zeros((T, 11, 16), np.float32)
        kk = np.asarray(dp['wgt'][1](wd), np.float32).reshape(9, -1)[:, :F]
        o[:, :9] = (kk * np.asarray(dp['scale'][1](wd), np.float32)[None, :F]).astype(np.float32).reshape(9, T, 16).transpose(1, 0, 2)
        o[:, 9], o[:, 10] = np.asarray(dp['shift'][1](wd), np.float32)[:F].reshape(T, 16), np.asarray(cp['shift'][1](wd), np.float32)[:F].reshape(T, 16)
        return o
    m.params = {'wgt': (((F // 16) * nk * 512,), frags, c.dtype), 'scale': cp['scale'], 'wgt2': ((F // 16, 11, 16), tab)}
    if d.gate is not None:     # the squeeze-excite sums: one row per (strip, row segment)
        part = d.gate
        part.h, part.w = head_walk_rows(d.h, d.w), 1
        part.elems = part.h * part.w * part.ld
        part.bytes = part.elems * rt.ESIZE[part.dtype]
        m.gate, m.se_reduced = part, part.h
    if hasattr(c, 'accounting_srcs'):
        m.fused[0].accounting_srcs = c.accounting_srcs
    return m


# (measured, round 5, SE EfficientNet-B0 bf16 @416 batch 128 | B3 f16 @640 batch 32, us, fused | conv + depthwise: td3 (2 chunks) 69 | 90, 52 | 62;
#  td2 (7 / 8 chunks) 64 | 73, 53 | 54; bu3 (4 chunks, gated) 86 | 87, 60 | 58; bu2 (7) 62 | 57, 49 | 46 - the 16-bit pair of launches runs at
#  3.6 - 4 TB/s on half the float32 bytes, the fused launch is bound by the depthwise stage's VALU work (DPP multiply-adds on 14 of 16
#  lanes, 15 rows walked per 13 stored, Swish at the transcendental rate); with steps in flight all four fused: B0 +0.4 %, B3 -3.8 %.
#  Default: the blocks of at most 2 chunks - td3.)
HEAD_WALK16_MAX_NK = min(8, int(os.environ.get('YOLORET_HEAD_WALK16_MAX_NK', '2')))


def fuse_head_blocks(ops, bufs, output_buf_ids, nosplit=frozenset(), stream_ok=True):
    readers = {}
    for op in ops:
        for s_ in op.srcs:
            readers[id(s_.buf)] = readers.get(id(s_.buf), 0) + 1
        for b in (op.res, op.gate):
            if b is not None:
                readers[id(b)] = readers.get(id(b), 0) + 1
    out, i = [], 0
    while i < len(ops):
        c = ops[i]
        d = ops[i + 1] if i + 1 < len(ops) else None
        kp = sum(round_up(s_.c, 4) for s_ in c.srcs if s_.xform != 'up2_add')
        if c.kind == rt.OP_POINTWISE and c.dtype != 0:
            m16 = _head_block16(c, d, readers, output_buf_ids)
            if m16 is not None:
                out.append(m16)
                i += 2
            else:
                out.append(c)
                i += 1
            continue
        ok = (d is not None and c.kind == rt.OP_POINTWISE and c.dtype == 0 and c.act in ('relu6', 'none', 'swish', 'leaky') and 'scale' in c.params
              and c.res is None and not getattr(c, 'stride', 0) and not (c.se_reduced & 0x10000) and not getattr(c, 'accounted_in', None)
              and all(s_.xform in ('identity', 'up2', 'maxpool2', 'maxpool4', 'up2_add') for s_ in c.srcs)
              and (c.gate is None or (len(c.srcs) == 1 and c.srcs[0].xform == 'identity'))
              and kp >= 16 and c.cout % 4 == 0 and c.out.external_slot < 0 and c.out.id not in output_buf_ids and readers.get(id(c.out), 0) == 1
              and d.kind == rt.OP_DEPTHWISE and d.dtype == 0 and d.k == 3 and d.stride == 1 and len(d.srcs) == 1 and d.srcs[0].buf is c.out
              and d.srcs[0].xform == 'identity' and d.srcs[0].c == c.cout and d.act in ('swish', 'relu6', 'none') and d.out.ld % 4 == 0
              and d.out.dtype == 0 and not (c.h == 1 and c.w == 1) and (d.gate is not None or FUSE_HEAD_ALL)
              and c.name.rsplit('_', 1)[0] + '_head' not in nosplit)       # (the head kernels exist in the split form only)
        if not ok:
            if c.kind == rt.OP_POINTWISE and c.name.rsplit('_', 1)[0] + '_head' in nosplit:
                c.se_reduced |= 0x10000      # the unfused conv of a head block that left the split form stays off it as well
            out.append(c)
            i += 1
            continue
        F, ldf = c.cout, round_up(c.cout, 4)
        m = OpRec(rt.OP_HEAD, c.name.rsplit('_', 1)[0] + '_head', act=d.act, h=d.h, w=d.w, cin=c.cin, cout=F, k=3 | rt.ACT[c.act] << 8, stride=1,
                  srcs=list(c.srcs), out=d.out, res=c.gate, macs=c.macs + d.macs, dtype=0)
        m.fused = [c, d]
        if getattr(c, 'folded_projection', None):
            m.folded_projection = c.folded_projection
        m.params = {'wgt': c.params['wgt'], 'scale': c.params['scale'], 'shift': c.params['shift']}
        dp = d.params
        kseg = [s_.c for s_ in c.srcs if s_.xform != 'up2_add']
        nk = sum((c_ + 31) // 32 for c_ in kseg)
        nt = 2 if nk <= 4 else 1
        bname_h = c.name.rsplit('_', 1)[0]
        ksrc = [s_ for s_ in c.srcs if s_.xform != 'up2_add']
        # ... where it measured ahead (MobileNetV2 x0.75 @416, 64 images, us, streaming | before): the 26 x 26 heads - td2 48 | 69, bu2 46 | 73 -; not the
        # 13 x 13 ones (two workgroups per image on half the chip: td1 50 | 51, bu1 56 | 37) nor the 52 x 52 ones (four generations of
        # workgroups, each with its prologue: td3 64 | 42, bu3 103 | 81).  YOLORET_HEAD_STREAM=<names> forces it for the named blocks.
        # (... and, with up to four workgroups sharing an (image, strip, segment) by runs of tile pairs, td1 - 7 chunks, a pooled source: 52 -> 37 us; bu1 -
        #  11 chunks - stays: 42 | 40)
        hs_shape = HEAD_STREAM_ONLY or (20 <= d.h <= 30 and d.w <= 28 and nk >= 5) or (d.h < 20 and d.w <= 14 and 5 <= nk <= 8)
        stream = (HEAD_STREAM and stream_ok and hs_shape and not SE_TAIL and (not HEAD_STREAM_ONLY or bname_h in HEAD_STREAM_ONLY)      # (the opt-in SE tail lives in the older forms) and all(s_.xform in ('identity', 'maxpool2', 'up2_add') for s_ in c.srcs)
                  and all(s_.xform != 'up2_add' for s_ in c.srcs[:-1]) and 1 <= len(ksrc) <= 3 and F % 32 == 0 and c.act in ('relu6', 'none')
                  and nk <= (7 if head_stream_geometry(d.h, d.w)[0] == 2 else 11) and not (c.gate is not None and (len(ksrc) != 1 or ksrc[0].xform != 'identity'))
                  and all(s_.buf.ld % 4 == 0 for s_ in ksrc)
                  and 4 * nk * 2048 + 2 * 10 * (2 if d.h >= 20 else 1) * 2048 + F * 11 * 4 + F * 4 <= 160 * 1024)
        walk = (not stream and HEAD_WALK and all(s_.xform in ('identity', 'up2_add') for s_ in c.srcs) and len(kseg) <= 3 and nk <= HEAD_WALK_MAX_NK and F % 16 == 0 and (F // 16 // nt) % 4 == 0
                and F // 16 % nt == 0 and c.act in ('relu6', 'none') and not (c.gate is not None and len(kseg) != len(c.srcs)) and F * 11 * 4 <= 64 * 1024)
        if walk or stream:
            # the walking form (headwalk.hip, k bit 6) / the weight-streaming form (headstream.hip, k bits 5 and 6): weights as float16
            # planes with the conv's BN scale folded in, YR_OP_MBR's tap table
            m.k |= 0x60 if stream else 0x40
            cp = c.params

            def planes(wd, cp=cp, kseg=kseg, F=F):
                return head_pack((cp['wgt'][1](wd)[:F] * cp['scale'][1](wd)[:F, None]).astype(np.float32), kseg)

            def tab(wd, cp=cp, dp=dp, F=F):
                T = F // 16
                o = np.zeros((T, 11, 16), np.float32)
                o[:, :9] = (dp['wgt'][1](wd).reshape(9, -1)[:, :F] * dp['scale'][1](wd)[None, :F]).astype(np.float32).reshape(9, T, 16).transpose(1, 0, 2)
                o[:, 9], o[:, 10] = dp['shift'][1](wd)[:F].reshape(T, 16), cp['shift'][1](wd)[:F].reshape(T, 16)
                return o
            m.params = {'wgt': (((F // 16) * nk * 512,), planes), 'scale': c.params['scale'], 'wgt2': ((F // 16, 11, 16), tab)}
        elif all(s_.xform in ('identity', 'up2', 'up2_add') for s_ in c.srcs) and len(kseg) <= 3 and HEAD_DMA:
            # no pooled source: the LDS-direct kernel, weights as float16 planes in fragment order (k bit 7)
            m.k |= 0x80
            m.params['wgt'] = ((((F + 15) // 16) * nk * 512,), lambda wd, wf=c.params['wgt'][1], kseg=kseg: head_pack(wf(wd), kseg))

        def dw_rows(wd, dp=dp, F=F, ldf=ldf):
            o = np.zeros((10, ldf), np.float32)
            o[:9, :F] = (dp['wgt'][1](wd).reshape(9, -1)[:, :F] * dp['scale'][1](wd)[None, :F]).astype(np.float32)
            o[9, :F] = dp['shift'][1](wd)[:F]
            return o
        if not (walk or stream):
            m.params['wgt2'] = ((10, ldf), dw_rows)
        if d.gate is not None:     # the squeeze-excite sums: one row per region (per strip and row segment in the walking form)
            nsy, nsx = (head_stream_rows(d.h, d.w), 1) if stream else (head_walk_rows(d.h, d.w), 1) if walk else head_regions(d.h, d.w)
            part = d.gate
            part.h, part.w = nsy * nsx, 1
            part.elems = part.h * part.w * part.ld
            part.bytes = part.elems * rt.ESIZE[part.dtype]
            m.gate, m.se_reduced = part, nsy * nsx
        if hasattr(c, 'accounting_srcs'):
            m.fused[0].accounting_srcs = c.accounting_srcs
        out.append(m)
        i += 2
    return out


def se_tail_into_producers(ops):
    producer_of_sums = {id(op.gate): op for op in ops
                        if op.gate is not None and (op.kind == rt.OP_HEAD or (op.kind == rt.OP_DEPTHWISE and not dw_uses_lds_form(op)))}
    drop = set()
    for fc in ops:
        if fc.kind != rt.OP_SE_FC or len(fc.srcs) != 1 or not fc.k > 0:
            continue
        P = producer_of_sums.get(id(fc.srcs[0].buf))
        if P is None or P.gate_out is not None or fc.out.external_slot >= 0:
            continue
        if P.kind == rt.OP_HEAD and (P.k & 0x60) == 0x60:      # (the weight-streaming head form has no tail: its waves never meet)
            continue
        C, R = fc.cin, fc.se_reduced
        ldc = round_up(C, 4)
        if C != P.cout or fc.k != P.h * P.w or ldc + R > SE_TAIL_LDS:
            continue
        fp = fc.params

        def se_w(wd, fp=fp, R=R, ldc=ldc):
            """W1 [ldc][R4] | W2 [R][ldc] | b1 [R4] | b2 [ldc] (include/yoloret_hip.h: se_w)"""
            return np.concatenate([np.asarray(fp['wgt'][1](wd), np.float32).ravel(), np.asarray(fp['wgt2'][1](wd), np.float32).ravel(),
                                   np.asarray(fp['b1'][1](wd), np.float32).ravel(), np.asarray(fp['b2'][1](wd), np.float32).ravel()])
        P.params['se_w'] = ((ldc * round_up(R, 4) + R * ldc + round_up(R, 4) + ldc,), se_w)
        P.gate_out, P.se_hidden = fc.out, R
        P.macs += fc.macs
        P.absorbed = list(getattr(P, 'absorbed', ())) + [fc]
        drop.add(id(fc))
    return [op for op in ops if id(op) not in drop]



def mbr_pack(we_t, e_scale, e_shift, dw, d_scale, d_shift, wp_t, p_scale, p_shift):
    """Parameters of a YR_OP_MBR block in the layout of include/yoloret_hip.h (mbr.hip): the MFMA A fragments of both 1x1
    convolutions in REGISTER order (one coalesced dword load per register), BN scales folded in.
    we_t [cexp][>=cin] expand (pointwise layout), dw [9][>=cexp] depthwise taps, wp_t [cout][>=cexp] project; BN vectors per layer.
    -> (wgt [T][KE + 4 TO][64], wgt2 [T][11][16], b2 [16 TO]) float32."""
    cexp = dw.shape[1] // 16 * 16
    if wp_t is None:     # YR_OP_MBE: expand + depthwise only
        wp_t, p_scale, p_shift = np.zeros((0, cexp), np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32)
    cout = wp_t.shape[0]
    cin = we_t.shape[1] // 8 * 8
    assert dw.shape[1] >= cexp and cexp % 16 == 0 and cin % 16 in (0, 8), (dw.shape, we_t.shape)
    T, TO, KE, nmain = cexp // 16, (cout + 15) // 16, cin // 4, cin // 16
    wef = (we_t[:cexp, :cin] * e_scale[:cexp, None]).astype(np.float32)
    wpf = np.zeros((16 * TO, cexp), np.float32)
    wpf[:cout] = (wp_t[:, :cexp] * p_scale[:cout, None]).astype(np.float32)
    lane = np.arange(64)
    m, g = lane % 16, lane // 16
    wa = np.zeros((T, KE + 4 * TO, 64), np.float32)
    for j in range(T):
        for q in range(KE):
            c, s = divmod(q, 4)
            kidx = 16 * c + 4 * g + s if c < nmain else 16 * nmain + 2 * g + s   # (a trailing 8-channel chunk: two steps)
            wa[j, q] = wef[16 * j + m, kidx]
        for t in range(TO):
            for s in range(4):
                wa[j, KE + 4 * t + s] = wpf[16 * t + m, 16 * j + 4 * g + s]
    tab = np.zeros((T, 11, 16), np.float32)
    tab[:, :9] = (dw[:, :cexp] * d_scale[None, :cexp]).astype(np.float32).reshape(9, T, 16).transpose(1, 0, 2)
    tab[:, 9], tab[:, 10] = d_shift[:cexp].reshape(T, 16), e_shift[:cexp].reshape(T, 16)
    b2 = np.zeros(16 * TO, np.float32)
    b2[:cout] = p_shift[:cout]
    return wa, tab, b2


def mbs_wave_pairs(T, nw):
    """The expanded-tile pairs of the SPLIT form of YR_OP_MBR in the order of the packed projection fragments: mbr_kernel gives
    the first T % nw waves one tile more than the others; a wave pairs ITS tiles (t0, t0 + 1), (t0 + 2, t0 + 3) ... and an odd
    last one with nothing."""
    ntl, r = divmod(T, nw)
    pairs, t0 = [], 0
    for w in range(nw):
        nt = ntl + (1 if w < r else 0)
        for q in range(0, nt, 2):
            pairs.append((t0 + q, t0 + q + 1 if q + 1 < nt else None))
        t0 += nt
    return pairs


def mbs_pack(we_t, e_scale, e_shift, dw, d_scale, d_shift, wp_t, p_scale, p_shift, nw):
    """Parameters of a YR_OP_MBR block in its SPLIT form (k bit 7; mbr.hip, SP): both 1x1 convolutions on the 16-bit matrix pipe
    with every float32 operand cut into two float16 planes, w = h + 2^-11 m.  wgt = [T][NKE][2 planes][64 lanes][8 halves] for the
    expand conv (lane (m, g) of tile j, step c: We[16 j + m][32 c + 8 g + i] * BN scale, zero beyond cin), then per tile pair of
    mbs_wave_pairs(T, nw) [TO][2 planes][64][8]: Wp[16 t + m][16 tA + 4 g + i] (i < 4) | [16 tB + 4 g + i - 4] * BN scale - as the
    float32 words that hold them; wgt2 and b2 as mbr_pack."""
    _, tab, b2 = mbr_pack(we_t, e_scale, e_shift, dw, d_scale, d_shift, wp_t, p_scale, p_shift)
    cexp = dw.shape[1] // 16 * 16
    if wp_t is None:     # YR_OP_MBE: the expand part only
        wp_t, p_scale, nw = np.zeros((0, cexp), np.float32), np.zeros(0, np.float32), 0
    cout, cin = wp_t.shape[0], we_t.shape[1] // 8 * 8
    T, TO, NKE = cexp // 16, (cout + 15) // 16, (cin + 31) // 32
    wef = np.zeros((cexp, 32 * NKE), np.float32)
    wef[:, :cin] = (we_t[:cexp, :cin] * e_scale[:cexp, None]).astype(np.float32)
    wpf = np.zeros((16 * TO, cexp), np.float32)
    wpf[:cout] = (wp_t[:, :cexp] * p_scale[:cout, None]).astype(np.float32)
    assert max(np.abs(wef).max(), np.abs(wpf).max() if wpf.size else 0.0) < 60000.0, 'mbs_pack: a weight beyond the float16 range'

    def planes(x):
        h = x.astype(np.float16)
        m = ((x - h.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        return h, m
    lane = np.arange(64)
    m_, g = lane % 16, lane // 16
    i8 = np.arange(8)
    ex = np.zeros((T, NKE, 2, 64, 8), np.float16)
    for j in range(T):
        for c in range(NKE):
            v = wef[(16 * j + m_)[:, None], 32 * c + 8 * g[:, None] + i8[None, :]]
            ex[j, c, 0], ex[j, c, 1] = planes(v)
    pairs = mbs_wave_pairs(T, nw) if nw else []
    pr = np.zeros((len(pairs), TO, 2, 64, 8), np.float16)
    for q, (ta, tb) in enumerate(pairs):
        for t in range(TO):
            v = np.zeros((64, 8), np.float32)
            v[:, :4] = wpf[(16 * t + m_)[:, None], 16 * ta + 4 * g[:, None] + i8[None, :4]]
            if tb is not None:
                v[:, 4:] = wpf[(16 * t + m_)[:, None], 16 * tb + 4 * g[:, None] + i8[None, :4]]
            pr[q, t, 0], pr[q, t, 1] = planes(v)
    wa = np.concatenate([ex.ravel(), pr.ravel()]).view(np.float32)
    return wa, tab, b2


def mbk_chunk_words(cin, cout):
    """float32 words of one chunk of the weight-streaming form (one pair of expanded tiles): expand fragments of both tiles, project
    fragments of the pair, the two tiles' [11][16] tables padded to 2 KB."""
    return (4 * ((cin + 31) // 32) + 2 * ((cout + 15) // 16)) * 256 + 512


def mbk_pack(we_t, e_scale, e_shift, dw, d_scale, d_shift, wp_t, p_scale, p_shift):
    """Parameters of a YR_OP_MBR block in its WEIGHT-STREAMING form (k bits 6, 7; mbk.hip): ceil(T / 2) chunks, chunk q = the pair of
    expanded tiles (2 q, 2 q + 1) = [2 tiles][NKE][2 planes][64 lanes][8 halves] expand fragments | [TO][2 planes][64][8] project
    fragments | [2 tiles][11][16] float32 (taps x BN scale | depthwise BN shift | expand BN shift) | zeros up to 2 KB - the fragments
    are mbs_pack's (one wave: consecutive tiles pair up), re-ordered so that a chunk is ONE contiguous piece of the blob.
    -> (wgt [NQ * chunk words], b2 [16 TO])."""
    wa, tab, b2 = mbs_pack(we_t, e_scale, e_shift, dw, d_scale, d_shift, wp_t, p_scale, p_shift, 1)
    cexp = dw.shape[1] // 16 * 16
    cout, cin = wp_t.shape[0], we_t.shape[1] // 8 * 8
    T, TO, NKE = cexp // 16, (cout + 15) // 16, (cin + 31) // 32
    NQ = (T + 1) // 2
    ex = wa[:T * NKE * 512].reshape(T, NKE * 512)
    pr = wa[T * NKE * 512:].reshape(NQ, TO * 512)
    cw = mbk_chunk_words(cin, cout)
    out = np.zeros((NQ, cw), np.float32)
    for q in range(NQ):
        o = 0
        for t in (2 * q, 2 * q + 1):
            if t < T:
                out[q, o:o + NKE * 512] = ex[t]
            o += NKE * 512
        out[q, o:o + TO * 512] = pr[q]
        o += TO * 512
        for t in (2 * q, 2 * q + 1):
            if t < T:
                out[q, o:o + 176] = tab[t].ravel()
            o += 176
    return out.ravel(), b2


def mbk_segs(stride, h, pad_t, nw, rows):
    """Row segments the weight-streaming form cuts a map of h input rows into (mbk.hip: mbk_segs)."""
    ho = (h + stride - 1) // stride
    if stride == 1:
        nr = nw * rows
        return 1 if h <= nr else (h - nr + nr - 3) // (nr - 2) + 1
    s = 0
    while True:
        ylast = s * (nw - 1) + nw - 1
        end = ylast if 2 * ylast - pad_t + 2 >= h else ylast - 1
        if end >= ho - 1:
            return s + 1
        s += 1


def fuse_inverted_residuals(ops, output_buf_ids, blocks=True, dtype=0, bufs=None, nosplit=frozenset(), mbk=True):
    """Peephole over the lowered op list: [POINTWISE expand+act ->] DEPTHWISE 3x3+act -> POINTWISE
    project (+residual == block input) becomes one MBCONV op whose expanded tensors never reach HBM.
    blocks=False (the small-batch plan) keeps only the network-entry fusion (stem + first block).
    16-bit plans (dtype != 0): the lane-per-pixel kernels (stemblock, mblane) take 16-bit inputs / outputs - they
    compute in float32 from registers, so only their loads and stores change; the MFMA block kernel (mbconv.hip) is
    float32 only and is not selected.  Blocks with squeeze-excite (the projection needs the gate of the complete depthwise
    map) get their first two thirds fused instead: POINTWISE expand -> DEPTHWISE becomes one MBX op that stores the
    depthwise map and the per-tile channel sums the SE_FC op finishes the squeeze from."""
    max_cin = FUSE_MAX_CIN if blocks else 0
    readers = {}
    for op in ops:
        for s in op.srcs:
            readers[s.buf.id] = readers.get(s.buf.id, 0) + 1
        for b in (op.res, op.gate):
            if b is not None:
                readers[b.id] = readers.get(b.id, 0) + 1

    def private(buf):  # read by exactly one op and not a model output
        return readers.get(buf.id, 0) == 1 and buf.id not in output_buf_ids and buf.external_slot < 0

    def plain1(op):
        return len(op.srcs) == 1 and op.srcs[0].xform == 'identity' and op.gate is None

    def lane_ok(exp, block_in, proj, dw=None):  # the lane-per-pixel kernel (mblane.hip) is built for this block shape
        widths = (round_up(block_in.c, 4) // 4, round_up(proj.cout, 8))
        if exp is None:   # a block without expand conv (expand ratio 1): stride 1, a few widths (launch_ml_ident)
            return (FUSE_LANE and FUSE_LANE_NO_EXPAND and dw is not None and dw.stride == 1 and proj.out.ld % 2 == 0
                    and widths in MBLANE_IDENT_WIDTHS)
        return FUSE_LANE and proj.out.ld % 2 == 0 and widths in MBLANE_WIDTHS

    def pad_to(fn, n, ld):
        def f(wd):
            o = np.zeros(ld, np.float32)
            o[:n] = fn(wd)[:n]
            return o
        return f

    out, i = [], 0
    while i < len(ops):
        e = ops[i]
        # ---- network entry: STEM -> DEPTHWISE 3x3 s1 -> POINTWISE project (MobileNetV2 Conv1 + block 0)
        if FUSE_STEM and e.kind == rt.OP_STEM and private(e.out) and i + 2 < len(ops) and e.act in ('relu6', 'swish'):
            d, p = ops[i + 1], ops[i + 2]
            if (d.kind == rt.OP_DEPTHWISE and d.k == 3 and d.stride == 1 and plain1(d) and d.srcs[0].buf is e.out
                    and d.act == e.act and private(d.out) and p.kind == rt.OP_POINTWISE and plain1(p)
                    and p.srcs[0].buf is d.out and p.act == 'none' and p.res is None and 'scale' in p.params
                    and (round_up(e.cout, 4) // 2, round_up(p.cout, 8)) in STEMBLOCK_WIDTHS):
                c1, cout = e.cout, p.cout
                c1p, cop = round_up(c1, 4), round_up(cout, 8)
                m = OpRec(rt.OP_STEMBLOCK, e.name + '_block0', act=e.act, h=p.h, w=p.w, cin=3, cout=cout, k=3, stride=2,
                          se_reduced=c1, srcs=[e.srcs[0]], out=p.out, macs=e.macs + d.macs + p.macs, dtype=dtype)
                m.fused = [e, d, p]

                def per_pair(prm, taps, c1p=c1p):
                    """[taps][c1p] weights + scale/shift [c1p]  ->  [c1p/2][taps x 2, times the scale | 1 1 | shift 2]
                    (the kernel adds the shift to the sum of products: one float32 rounding of w * scale per weight)"""
                    def f(wd):
                        sc = pad_to(prm['scale'][1], c1p, c1p)(wd)
                        w = (prm['wgt'][1](wd).reshape(taps, -1)[:, :c1p] * sc[None]).astype(np.float32)
                        rows = np.concatenate([w, np.ones((1, c1p), np.float32),
                                               pad_to(prm['shift'][1], c1p, c1p)(wd)[None]])     # [taps+2][c1p]
                        return np.ascontiguousarray(rows.reshape(taps + 2, c1p // 2, 2).transpose(1, 0, 2)).reshape(c1p // 2, -1)
                    return ((c1p // 2, (taps + 2) * 2), f)

                def proj_w(wd, pw=p.params['wgt'][1], c1p=c1p, cop=cop, cout=cout):
                    o = np.zeros((c1p, cop), np.float32)
                    o[:, :cout] = pw(wd)[:, :c1p].T       # pointwise layout is Wt[cout][kp]
                    return o
                pp2 = p.params
                img = e.srcs[0].buf
                if STEM_MFMA and dtype != 0 and img.h % 2 == 0 and img.w % 2 == 0 and c1 <= STEM_MFMA_MAX_C1 and cout <= 32:
                    # 16-bit plans: the matrix-pipe form (stemblock_h.hip).  Stem weights as the plan's type, [C1P][32] with the k
                    # space in the kernel's order: image rows 0..2 x values 0..7 of the row's 9 (kx, c) | value 8 of rows 0..2 | 0 x 5
                    c1m, com = round_up(c1, 32), round_up(cout, 16)
                    korder = [g * 9 + i_ for g in range(3) for i_ in range(8)] + [i_ * 9 + 8 for i_ in range(3)]

                    def stem_w(wd, ew=e.params['wgt'][1], c1=c1, c1m=c1m, korder=korder):
                        o = np.zeros((c1m, 32), np.float32)
                        o[:c1, :27] = ew(wd)[korder, :c1].T
                        return o

                    def dw_rows(wd, dp=d.params, c1=c1, c1m=c1m):
                        o = np.zeros((10, c1m), np.float32)
                        sc = dp['scale'][1](wd)[:c1]
                        o[:9, :c1] = (dp['wgt'][1](wd).reshape(9, -1)[:, :c1] * sc[None]).astype(np.float32)
                        o[9, :c1] = dp['shift'][1](wd)[:c1]
                        return o

                    def proj_m(wd, pw=p.params['wgt'][1], c1=c1, c1m=c1m, com=com, cout=cout):
                        o = np.zeros((com, c1m), np.float32)
                        o[:cout, :c1] = pw(wd)[:cout, :c1]       # pointwise layout is Wt[cout][kp]
                        return o
                    m.params['wgt'] = ((c1m, 32), stem_w, dtype)
                    m.params['scale'] = ((c1m,), pad_to(e.params['scale'][1], c1, c1m))
                    m.params['shift'] = ((c1m,), pad_to(e.params['shift'][1], c1, c1m))
                    m.params['wgt2'] = ((10, c1m), dw_rows)
                    m.params['b1'] = ((com, c1m), proj_m, dtype)
                    m.params['b2'] = ((2 * com,), lambda wd, pp2=pp2, cout=cout, com=com: np.concatenate(
                        [pad_to(pp2['scale'][1], cout, com)(wd), pad_to(pp2['shift'][1], cout, com)(wd)]))
                    out.append(m)
                    i += 3
                    continue
                m.params['wgt'] = per_pair(e.params, 27)
                m.params['wgt2'] = per_pair(d.params, 9)
                m.params['b1'] = ((c1p, cop), proj_w)
                m.params['b2'] = ((2 * cop,), lambda wd, pp2=pp2, cout=cout, cop=cop: np.concatenate(
                    [pad_to(pp2['scale'][1], cout, cop)(wd), pad_to(pp2['shift'][1], cout, cop)(wd)]))
                out.append(m)
                i += 3
                continue
        # ---- network entry of the squeeze-excite EfficientNets: STEM -> DEPTHWISE 3x3 s1 whose map feeds an SE block (the
        # projection waits for the gate): stem + depthwise as one kernel, the depthwise map and its per-tile channel sums out.
        # 16-bit plans only: measured (B0 @416, 128 images) 0.53 vs 0.23 + 0.34 ms with bf16 maps - the kernel is bound by its
        # two swish passes, the 708 MB it no longer moves buy 7 % - and 0.77 vs 0.36 + 0.39 ms in float32 (pinned expf)
        if (FUSE_STEM and FUSE_STEMDW and dtype != 0 and blocks and bufs is not None and e.kind == rt.OP_STEM and private(e.out) and i + 1 < len(ops)
                and e.act in ('relu6', 'swish') and round_up(e.cout, 4) // 2 in (16, 20, 24)):
            d = ops[i + 1]
            fc = None
            if (d.kind == rt.OP_DEPTHWISE and d.k == 3 and d.stride == 1 and len(d.srcs) == 1 and d.srcs[0].xform == 'identity'
                    and d.srcs[0].buf is e.out and d.act == e.act and d.out.external_slot < 0 and d.out.id not in output_buf_ids
                    and d.out.ld % 4 == 0):
                fc = next((o for o in ops if o.kind == rt.OP_SE_FC and getattr(o, 'merged_mean', 0) and len(o.srcs) == 1
                           and (o.srcs[0].buf is d.out or (d.gate is not None and o.srcs[0].buf is d.gate))), None)
            if fc is not None:
                c1 = e.cout
                c1p = round_up(c1, 4)
                rows = ((d.h + 13) // 14) * ((d.w + 13) // 14)           # one row per 14 x 14 output tile (stemblock.hip)
                part = d.gate
                if part is None:
                    part = Buf(len(bufs), rows, 1, c1, round_up(c1, 4), name=d.name + ':se_sums', dtype=0)
                    bufs.append(part)
                    fc.srcs = [Seg(part, c1, 'identity')]
                    fc.k = d.h * d.w
                else:
                    part.h, part.elems = rows, rows * part.w * part.ld
                    part.bytes = part.elems * rt.ESIZE[part.dtype]
                # the matrix-pipe form of this entry (mbxr_h.hip: stemxr_kernel; float32 image of even size, at most 48 stem channels):
                # asked for by the PLAN (k = 3 | 1 << 8), so that every batch size rounds the same way
                src0 = e.srcs[0].buf
                mfma_entry = (FUSE_STEMDW_MFMA and src0.dtype == 0 and src0.h % 2 == 0 and src0.w % 2 == 0 and c1 <= 48 and c1 % 4 == 0
                              and e.act in ('relu6', 'swish') and d.out.ld % 4 == 0)
                m = OpRec(rt.OP_STEMBLOCK, e.name + '_dw', act=e.act, h=d.h, w=d.w, cin=3, cout=c1, k=3 | (1 << 8 if mfma_entry else 0), stride=2,
                          se_reduced=c1, srcs=[e.srcs[0]], out=d.out, gate=part, macs=e.macs + d.macs, dtype=dtype)
                m.fused = [e, d]

                def per_pair2(prm, taps, c1p=c1p):
                    def f(wd):
                        sc = pad_to(prm['scale'][1], c1p, c1p)(wd)
                        w = (prm['wgt'][1](wd).reshape(taps, -1)[:, :c1p] * sc[None]).astype(np.float32)
                        rows_ = np.concatenate([w, np.ones((1, c1p), np.float32), pad_to(prm['shift'][1], c1p, c1p)(wd)[None]])
                        return np.ascontiguousarray(rows_.reshape(taps + 2, c1p // 2, 2).transpose(1, 0, 2)).reshape(c1p // 2, -1)
                    return ((c1p // 2, (taps + 2) * 2), f)
                m.params['wgt'] = per_pair2(e.params, 27)
                m.params['wgt2'] = per_pair2(d.params, 9)
                out.append(m)
                i += 2
                continue
        exp = dw = proj = None
        j = i
        if (e.kind == rt.OP_POINTWISE and plain1(e) and e.res is None and e.act in ('relu6', 'swish')
                and private(e.out) and 'scale' in e.params and not (e.h == 1 and e.w == 1) and j + 1 < len(ops)):
            exp, j = e, j + 1
        d = ops[j] if j < len(ops) else None
        # 16-bit plans: the MFMA block kernel (mbh.hip) takes every expand -> depthwise 3x3 | 5x5 -> project block with up
        # to 128 input / output channels, whatever the map size (measured: it beats the unfused chain on every
        # MobileNetV2 block at batch 64 and the float32 lane kernels where both apply)
        if (FUSE_MBR and dtype == 0 and blocks and exp is not None and d is not None and d.kind == rt.OP_DEPTHWISE and d.k == 3
                and d.stride in (1, 2) and plain1(d) and private(d.out) and d.srcs[0].buf is exp.out and d.act == exp.act == 'relu6'
                and j + 1 < len(ops)):
            p, bi = ops[j + 1], exp.srcs[0]
            key = (bi.c, d.cin, p.cout if p.kind == rt.OP_POINTWISE else 0, d.stride, p.res is not None)
            bname = exp.name.rsplit('_', 1)[0]
            if (FUSE_MBK and mbk and MBR_SPLIT and key in MBK_SHAPES and bname + '_mbr' not in nosplit and p.kind == rt.OP_POINTWISE and plain1(p)
                    and p.srcs[0].buf is d.out and p.act == 'none' and 'scale' in p.params and bi.xform == 'identity'
                    and bi.buf.ld % 4 == 0 and p.out.ld % 4 == 0 and bi.buf.dtype == 0 and p.out.dtype == 0
                    and (p.res is None or (p.res is bi.buf and d.stride == 1 and p.cout == bi.c))):
                # the weight-streaming form: one launch, the expanded tensor and the depthwise map stay on the CU (mbk.hip)
                cin, cexp, cout = bi.c, d.cin, p.cout
                rows, nw = MBK_SHAPES[key]
                T, TO = cexp // 16, (cout + 15) // 16
                m = OpRec(rt.OP_MBR, bname + '_mbr', act='relu6', h=p.h, w=p.w, cin=cin, cout=cout, k=3 | 0xc0 | nw << 8 | rows << 16,
                          stride=d.stride, se_reduced=cexp, srcs=[bi], out=p.out, res=p.res, macs=exp.macs + d.macs + p.macs, dtype=0)
                m.fused = [exp, d, p]
                ep, dp, pp = exp.params, d.params, p.params

                def packed_k(which, ep=ep, dp=dp, pp=pp):
                    def f(wd):
                        return mbk_pack(ep['wgt'][1](wd), ep['scale'][1](wd), ep['shift'][1](wd), dp['wgt'][1](wd).reshape(9, -1),
                                        dp['scale'][1](wd), dp['shift'][1](wd), pp['wgt'][1](wd), pp['scale'][1](wd), pp['shift'][1](wd))[which]
                    return f
                m.params = {'wgt': (((T + 1) // 2 * mbk_chunk_words(cin, cout),), packed_k(0)), 'b2': ((16 * TO,), packed_k(1))}
                out.append(m)
                i = j + 2
                continue
            if (key in MBR_SHAPES and (not MBR_BLOCKS or bname in MBR_BLOCKS) and p.kind == rt.OP_POINTWISE and plain1(p)
                    and (len(MBR_SHAPES[key]) < 3 or p.h * p.w <= MBR_SHAPES[key][2])
                    and p.srcs[0].buf is d.out and p.act == 'none' and 'scale' in p.params and bi.xform == 'identity'
                    and bi.buf.ld % 4 == 0 and p.out.ld % 4 == 0 and bi.buf.dtype == 0 and p.out.dtype == 0
                    and (p.res is None or (p.res is bi.buf and d.stride == 1 and p.cout == bi.c))):
                cin, cexp, cout = bi.c, d.cin, p.cout
                nw, segs = MBR_SHAPES[key][:2]
                split = MBR_SPLIT and key in MBS_SHAPES and bname + '_mbr' not in nosplit     # (nosplit: Model.check_ranges found operands beyond the float16 range)
                if split:
                    nw = MBS_SHAPES[key]
                T, TO, KE = cexp // 16, (cout + 15) // 16, cin // 4
                m = OpRec(rt.OP_MBR, bname + '_mbr', act='relu6', h=p.h, w=p.w, cin=cin, cout=cout, k=3 | (0x80 if split else 0) | nw << 8 | segs << 16,
                          stride=d.stride, se_reduced=cexp, srcs=[bi], out=p.out, res=p.res, macs=exp.macs + d.macs + p.macs, dtype=0)
                m.fused = [exp, d, p]
                ep, dp, pp = exp.params, d.params, p.params
                if split:
                    def packed_s(which, ep=ep, dp=dp, pp=pp, nw=nw):
                        def f(wd):
                            return mbs_pack(ep['wgt'][1](wd), ep['scale'][1](wd), ep['shift'][1](wd), dp['wgt'][1](wd).reshape(9, -1),
                                            dp['scale'][1](wd), dp['shift'][1](wd), pp['wgt'][1](wd), pp['scale'][1](wd), pp['shift'][1](wd), nw)[which]
                        return f
                    nwords = (T * ((cin + 31) // 32) + len(mbs_wave_pairs(T, nw)) * TO) * 512
                    m.params = {'wgt': ((nwords,), packed_s(0)), 'wgt2': ((T, 11, 16), packed_s(1)), 'b2': ((16 * TO,), packed_s(2))}
                    out.append(m)
                    i = j + 2
                    continue

                def packed(which, ep=ep, dp=dp, pp=pp, cexp=cexp):
                    def f(wd):
                        return mbr_pack(ep['wgt'][1](wd), ep['scale'][1](wd), ep['shift'][1](wd), dp['wgt'][1](wd).reshape(9, -1),
                                        dp['scale'][1](wd), dp['shift'][1](wd), pp['wgt'][1](wd), pp['scale'][1](wd), pp['shift'][1](wd))[which]
                    return f
                m.params = {'wgt': ((T, KE + 4 * TO, 64), packed(0)), 'wgt2': ((T, 11, 16), packed(1)), 'b2': ((16 * TO,), packed(2))}
                out.append(m)
                i = j + 2
                continue
        if (FUSE_MBE and dtype == 0 and blocks and exp is not None and d is not None and d.kind == rt.OP_DEPTHWISE and d.k == 3
                and d.stride in (1, 2) and plain1(d) and d.gate is None and d.srcs[0].buf is exp.out and d.act == exp.act == 'relu6'
                and exp.srcs[0].c in MBE_CINS and d.cin % 16 == 0 and d.cin * 11 * 4 <= 64 * 1024 and exp.srcs[0].xform == 'identity'
                and exp.srcs[0].buf.ld % 4 == 0 and d.out.ld % 4 == 0 and exp.srcs[0].buf.dtype == 0 and d.out.dtype == 0
                and d.out.ld == round_up(d.cin, 4)):
            bi, cexp = exp.srcs[0], d.cin
            T, KE = cexp // 16, bi.c // 4
            mbe_split = (MBR_SPLIT and bi.c in MBS_MBE_CINS and exp.name.rsplit('_', 1)[0] + '_mbe' not in nosplit
                         and exp.name.rsplit('_', 1)[0] + '_mbr' not in nosplit)     # (... or the block's one-launch form was found out of range)
            m = OpRec(rt.OP_MBE, exp.name.rsplit('_', 1)[0] + '_mbe', act='relu6', h=d.h, w=d.w, cin=bi.c, cout=cexp, k=3 | (0x80 if mbe_split else 0), stride=d.stride,
                      srcs=[bi], out=d.out, macs=exp.macs + d.macs, dtype=0)
            m.fused = [exp, d]
            ep, dp = exp.params, d.params
            if mbe_split:
                def packed_es(which, ep=ep, dp=dp):
                    def f(wd):
                        wa, tab, _ = mbs_pack(ep['wgt'][1](wd), ep['scale'][1](wd), ep['shift'][1](wd), dp['wgt'][1](wd).reshape(9, -1),
                                              dp['scale'][1](wd), dp['shift'][1](wd), None, None, None, 0)
                        return wa if which == 0 else tab
                    return f
                m.params = {'wgt': ((T * ((bi.c + 31) // 32) * 512,), packed_es(0)), 'wgt2': ((T, 11, 16), packed_es(1))}
                out.append(m)
                i = j + 1
                continue

            def packed_e(which, ep=ep, dp=dp):
                def f(wd):
                    wa, tab, _ = mbr_pack(ep['wgt'][1](wd), ep['scale'][1](wd), ep['shift'][1](wd), dp['wgt'][1](wd).reshape(9, -1),
                                          dp['scale'][1](wd), dp['shift'][1](wd), None, None, None)
                    return wa if which == 0 else tab
                return f
            m.params = {'wgt': ((T, KE, 64), packed_e(0)), 'wgt2': ((T, 11, 16), packed_e(1))}
            out.append(m)
            i = j + 1
            continue
        mbh = None
        if (FUSE_MBH and dtype != 0 and blocks and exp is not None and d is not None and d.kind == rt.OP_DEPTHWISE and d.k in (3, 5)
                and d.stride in (1, 2) and '%d%d' % (d.k, d.stride) not in MBH_SKIP and not (d.k == 5 and d.stride == 1 and (d.cin > MBH_K5_MAX_CEXP or (d.cin > MBH_K5_SMALL_CEXP and d.h * d.w <= MBH_K5_SMALL_MAP))) and not (d.k == 3 and d.stride == 1 and d.cin > MBH_K3_MAX_CEXP) and plain1(d) and private(d.out) and d.srcs[0].buf is exp.out and d.act == exp.act
                and d.act in MBH_ACTS and j + 1 < len(ops)):
            p = ops[j + 1]
            bi = exp.srcs[0]
            if (p.kind == rt.OP_POINTWISE and plain1(p) and p.srcs[0].buf is d.out and p.act == 'none' and 'scale' in p.params
                    and p.out.dtype == dtype and bi.c <= 128 and p.cout <= 128 and bi.xform == 'identity' and bi.buf.dtype == dtype
                    and (p.res is None or (p.res is bi.buf and d.stride == 1 and p.cout == bi.c))):
                mbh = (d, p)
                # measured (tools/mbh_probe.py, batch 64): only on the first, stride-2 block (16 -> 96 -> 24 channels,
                # 208 x 208 -> 104 x 104) the float32 lane-per-pixel kernel is still ahead (0.223 vs 0.253 ms: its 17 x 17
                # halo tile leaves two workgroups per CU); from block_2 on (0.205 vs 0.138 ms) everything goes to mbh.  With 24
                # block inputs the lane kernel loses there too (EfficientNet-lite3 stage 2 entry, 320 x 320 -> 160 x 160, 32 images:
                # 0.48 vs 0.41 ms): only blocks of at most 16 inputs stay on it
                # round 3: that block has its own matrix-pipe kernel (mbn_h.hip, dispatched by yr_launch_mbh: 3x3 stride 2, at most 32
                # inputs in whole 16-byte vectors, at most 192 expanded channels, at most 32 outputs)
                mbn = MBN and d.k == 3 and d.stride == 2 and bi.c <= 32 and bi.c % 8 == 0 and d.cin <= 192 and p.cout <= 32 and p.res is None
                if not mbn and lane_ok(exp, bi, p) and d.k == 3 and d.stride == 2 and p.h * p.w >= MBH_LANE_MIN_PIXELS and bi.c <= MBH_LANE_MAX_CIN:
                    mbh = None
        mbx = None
        if (FUSE_MBX and mbh is None and bufs is not None and dtype != 0 and blocks and exp is not None and d is not None
                and d.kind == rt.OP_DEPTHWISE and d.k in (3, 5) and d.stride in (1, 2) and '%d%d' % (d.k, d.stride) not in MBX_SKIP and not (d.k == 5 and d.stride == 1 and d.cin > MBX_K5_MAX_CEXP) and len(d.srcs) == 1
                and d.srcs[0].xform == 'identity' and d.srcs[0].buf is exp.out and d.act == exp.act and d.out.dtype == dtype
                and exp.srcs[0].buf.dtype == dtype and exp.srcs[0].c <= 128 and d.out.external_slot < 0 and d.out.id not in output_buf_ids):
            # the squeeze of this depthwise map: an SE_FC op reading it (merged mean) or the partial sums it already writes
            mbx = next((o for o in ops if o.kind == rt.OP_SE_FC and getattr(o, 'merged_mean', 0) and len(o.srcs) == 1
                        and (o.srcs[0].buf is d.out or (d.gate is not None and o.srcs[0].buf is d.gate))), None)
        if mbx is not None:
            fc, bi = mbx, exp.srcs[0]
            cin, cexp, kk = bi.c, d.cin, d.k * d.k
            cexp_p, kp = round_up(cexp, 32), round_up(cin, 32)
            # partial-sum rows: enough for a 4 x 8 output tile (4 x 4 on small maps, where wide blocks need small tiles)
            rows = ((d.h + 3) // 4) * ((d.w + 7) // 8 if d.h * d.w > 1000 else (d.w + 3) // 4)
            part = d.gate
            if part is None:
                part = Buf(len(bufs), rows, 1, cexp, round_up(cexp, rt.VEC[dtype]), name=d.name + ':se_sums', dtype=0)
                bufs.append(part)
                fc.srcs = [Seg(part, cexp, 'identity')]
                fc.k = d.h * d.w
            else:
                part.h, part.elems = rows, rows * part.w * part.ld
                part.bytes = part.elems * rt.ESIZE[part.dtype]
            m = OpRec(rt.OP_MBX, exp.name.rsplit('_', 1)[0] + '_mbx', act=d.act, h=d.h, w=d.w, cin=cin, cout=cexp, k=d.k,
                      stride=d.stride, se_reduced=rows, srcs=[bi], out=d.out, gate=part, macs=exp.macs + d.macs, dtype=dtype)
            m.fused = [exp, d]
            ep, dp = exp.params, d.params

            def expand_wt(wd, ep=ep, cexp=cexp, cin=cin, cexp_p=cexp_p, kp=kp):
                o = np.zeros((cexp_p, kp), np.float32)
                o[:cexp, :cin] = ep['wgt'][1](wd)[:, :cin]            # pointwise layout Wt[cexp][k-space], one source
                return o

            def chunk_params(wd, ep=ep, dp=dp, cexp=cexp, cexp_p=cexp_p, kk=kk):
                o = np.zeros((kk + 4, cexp_p), np.float32)            # taps | dw scale | dw shift | expand scale | expand shift
                o[:kk, :cexp] = dp['wgt'][1](wd)[:, :cexp]
                o[kk, :cexp], o[kk + 1, :cexp] = dp['scale'][1](wd)[:cexp], dp['shift'][1](wd)[:cexp]
                o[kk + 2, :cexp], o[kk + 3, :cexp] = ep['scale'][1](wd)[:cexp], ep['shift'][1](wd)[:cexp]
                return o
            m.params = {'wgt': ((cexp_p, kp), expand_wt, dtype), 'wgt2': ((kk + 4, cexp_p), chunk_params)}
            out.append(m)
            i = j + 1
            continue
        if mbh is not None:
            d, p = mbh
            bi = exp.srcs[0]
            cin, cexp, cout, kk = bi.c, d.cin, p.cout, d.k * d.k
            cexp_p, kp, ldo = round_up(cexp, 32), round_up(cin, 32), round_up(cout, 8)
            m = OpRec(rt.OP_MBH, exp.name.rsplit('_', 1)[0] + '_mbh', act=d.act, h=p.h, w=p.w, cin=cin, cout=cout, k=d.k,
                      stride=d.stride, se_reduced=cexp, srcs=[bi], out=p.out, res=p.res, macs=exp.macs + d.macs + p.macs, dtype=dtype)
            m.fused = [exp, d, p]
            ep, dp, pp = exp.params, d.params, p.params

            def expand_wt(wd, ep=ep, cexp=cexp, cin=cin, cexp_p=cexp_p, kp=kp):
                o = np.zeros((cexp_p, kp), np.float32)
                o[:cexp, :cin] = ep['wgt'][1](wd)[:, :cin]            # pointwise layout Wt[cexp][k-space], one source
                return o

            def chunk_params(wd, ep=ep, dp=dp, cexp=cexp, cexp_p=cexp_p, kk=kk):
                o = np.zeros((kk + 4, cexp_p), np.float32)            # taps | dw scale | dw shift | expand scale | expand shift
                o[:kk, :cexp] = dp['wgt'][1](wd)[:, :cexp]
                o[kk, :cexp], o[kk + 1, :cexp] = dp['scale'][1](wd)[:cexp], dp['shift'][1](wd)[:cexp]
                o[kk + 2, :cexp], o[kk + 3, :cexp] = ep['scale'][1](wd)[:cexp], ep['shift'][1](wd)[:cexp]
                return o

            def project_wt(wd, pp=pp, cexp=cexp, cexp_p=cexp_p, cout=cout):
                o = np.zeros((cout, cexp_p), np.float32)
                o[:, :cexp] = pp['wgt'][1](wd)[:, :cexp]
                return o

            def project_bn(wd, pp=pp, cout=cout, ldo=ldo):
                o = np.zeros((2, ldo), np.float32)
                o[0, :cout], o[1, :cout] = pp['scale'][1](wd)[:cout], pp['shift'][1](wd)[:cout]
                return o.ravel()
            m.params = {'wgt': ((cexp_p, kp), expand_wt, dtype), 'wgt2': ((kk + 4, cexp_p), chunk_params),
                        'b1': ((cout, cexp_p), project_wt, dtype), 'b2': ((2 * ldo,), project_bn)}
            out.append(m)
            i = j + 2
            continue
        if (d is not None and d.kind == rt.OP_DEPTHWISE and d.k == 3 and plain1(d) and private(d.out)
                and (exp is None or (d.srcs[0].buf is exp.out and d.act == exp.act)) and d.act in ('relu6', 'swish')
                and d.srcs[0].buf.ld == round_up(d.cin, rt.VEC[dtype]) and j + 1 < len(ops)):
            p = ops[j + 1]
            block_in = exp.srcs[0] if exp is not None else d.srcs[0]
            lane = lane_ok(exp, block_in, p, d)
            if (lane and p.kind == rt.OP_POINTWISE and plain1(p) and p.srcs[0].buf is d.out and p.act == 'none'
                    and 'scale' in p.params and p.cout <= 224 and block_in.buf.ld % 4 == 0
                    and (exp is not None or FUSE_NO_EXPAND or lane) and block_in.c <= max_cin
                    and p.h * p.w >= (FUSE_LANE_MIN_PIXELS if lane else FUSE_MIN_PIXELS)
                    and (p.res is None or (p.res is block_in.buf and d.stride == 1 and p.cout == block_in.c))):
                dw, proj = d, p
        if proj is None:
            out.append(e)
            i += 1
            continue
        block_in = exp.srcs[0] if exp is not None else dw.srcs[0]
        cexp, cout = dw.cin, proj.cout
        lde, ldo = round_up(cexp, 4), round_up(cout, 4)
        m = OpRec(rt.OP_MBLANE, (exp or dw).name.rsplit('_', 1)[0] + '_mblane', act=dw.act, h=proj.h, w=proj.w,
                  cin=block_in.c, cout=cout, k=3, stride=dw.stride, se_reduced=cexp, srcs=[block_in], out=proj.out,
                  res=proj.res, macs=(exp.macs if exp else 0) + dw.macs + proj.macs, dtype=dtype)
        m.fused = [o for o in (exp, dw, proj) if o is not None]

        def padded(fn, n, ld):
            def f(wd):
                o = np.zeros(ld, np.float32)
                o[:n] = fn(wd)[:n]
                return o
            return f
        cinp, cop = round_up(block_in.c, 4), round_up(cout, 8)
        if lane_ok(exp, block_in, proj, dw):
            # lane-per-pixel formulation (mblane.hip): everything packed per expanded-channel pair
            npair = round_up((cexp + 1) // 2, 8)
            e2 = 2 * npair

            def pairs(rows, scale, shift, e2=e2, cexp=cexp):
                """rows [K][>=cexp] + BN [>=cexp]  ->  [P][K x 2, times the BN scale | 1 1 | shift 2]  (the kernels add
                the shift to the sum of products; the scale slot stays in the layout)"""
                full = np.zeros((rows.shape[0] + 2, e2), np.float32)
                full[:-2, :cexp] = (rows[:, :cexp] * scale[None, :cexp]).astype(np.float32)
                full[-2, :cexp], full[-1, :cexp] = 1.0, shift[:cexp]
                return np.ascontiguousarray(full.reshape(-1, e2 // 2, 2).transpose(1, 0, 2)).reshape(e2 // 2, -1)
            ep, dwp, pp_ = (exp.params if exp is not None else None), dw.params, proj.params

            def expand_w(wd, ep=ep, cinp=cinp, pairs=pairs):
                wt = ep['wgt'][1](wd)                      # pointwise layout Wt[cexp][kp]
                rows = np.zeros((cinp, wt.shape[0]), np.float32)
                rows[:wt.shape[1]] = wt.T[:cinp]
                return pairs(rows, ep['scale'][1](wd), ep['shift'][1](wd))

            def proj_w(wd, pp_=pp_, e2=e2, cop=cop, cout=cout, cexp=cexp):
                o = np.zeros((e2, cop), np.float32)
                o[:cexp, :cout] = pp_['wgt'][1](wd)[:, :cexp].T
                return o
            if exp is not None:          # (without expand conv the op has no `wgt`: the kernel copies the input pairs)
                m.params['wgt'] = ((npair, cinp * 2 + 4), expand_w)
            m.params['wgt2'] = ((npair, 22), lambda wd, dwp=dwp, pairs=pairs: pairs(dwp['wgt'][1](wd), dwp['scale'][1](wd), dwp['shift'][1](wd)))
            m.params['b1'] = ((e2, cop), proj_w)
            m.params['b2'] = ((2 * cop,), lambda wd, pp_=pp_, cout=cout, cop=cop: np.concatenate(
                [padded(pp_['scale'][1], cout, cop)(wd), padded(pp_['shift'][1], cout, cop)(wd)]))
            out.append(m)
            i = j + 2
            continue
        raise AssertionError('a fused float32 block that the lane-per-pixel kernel does not take (mbconv.hip was removed)')
    return out


class Compiler:
    def __init__(self, inputs, outputs, fuse=True, dtype=0, nosplit=frozenset()):
        self.fuse = fuse
        self.nosplit = nosplit_aliases(nosplit)     # names of plan ops that must not run a split (float16-plane) form: Model.check_ranges
        self.dtype = rt.dtype_id(dtype)   # element type of the activations between ops
        self.V = rt.VEC[self.dtype]
        self.inputs = inputs
        self.outputs = list(outputs)
        self.bufs = []
        self.ops = []
        self.values = {}       # Tensor -> Value
        self.param_shapes = {}
        self.layer_seq = {}    # layer name -> creation sequence number (layers with parameters, reachable from the outputs)
        self.consumers = {}

    # ---------------------------------------------------------------- graph utilities
    def _topo(self):
        order, seen = [], set()

        def visit(t):
            n = t.node
            if n is None or id(n) in seen:
                return
            seen.add(id(n))
            for i in n.inputs:
                visit(i)
            order.append(n)

        import sys
        sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
        for o in self.outputs:
            visit(o)
        for n in order:
            for i in n.inputs:
                self.consumers.setdefault(id(i), []).append(n)
        return order

    def _sole_consumer(self, t, op=None, **attrs):
        cs = self.consumers.get(id(t), [])
        if len(cs) != 1 or any(t is o for o in self.outputs):
            return None
        n = cs[0]
        if op is not None and n.op != op:
            return None
        for k, v in attrs.items():
            if n.attrs.get(k) != v:
                return None
        return n

    def _new_buf(self, h, w, c, ld=None, external_slot=-1, name='', dtype=None):
        """Arena buffers hold the plan's activation type unless told otherwise; external ones (images, logits) and
        the SE vectors are float32 in every plan."""
        if dtype is None:
            dtype = self.dtype if external_slot < 0 else 0
        b = Buf(len(self.bufs), h, w, c, round_up(c, rt.VEC[dtype]) if ld is None else ld, external_slot, name, dtype)
        self.bufs.append(b)
        return b

    def _emit(self, op):
        op.dtype = self.dtype
        idx = len(self.ops)
        self.ops.append(op)
        op.out.first_def = idx if op.out.first_def is None else op.out.first_def
        for s in op.srcs:
            s.buf.last_use = max(s.buf.last_use, idx)
        for b in (op.res, op.gate):
            if b is not None:
                b.last_use = max(b.last_use, idx)
        return op

    def _out_buf_for(self, t, name):
        """Model outputs are written straight into the caller's dense y buffers."""
        for slot, o in enumerate(self.outputs, start=1):
            if t is o or (o.node is not None and o.node.op == 'reshape5' and o.node.inputs[0] is t):
                h, w, c = t.shape
                return self._new_buf(h, w, c, ld=c, external_slot=slot, name=name)
        h, w, c = t.shape
        return self._new_buf(h, w, c, name=name)

    def _plain(self, t):
        """Value of t as a single identity segment, materialising through GATHER if needed."""
        v = self.values[id(t)]
        if v.plain:
            return v.segs[0]
        if v.gate is not None:
            raise NotImplementedError('SE-gated tensor consumed by something other than a 1x1 conv')
        h, w, c = t.shape
        out = self._new_buf(h, w, c, name=t.name + ':gather')
        op = OpRec(rt.OP_GATHER, t.name + ':gather', h=h, w=w, cin=c, cout=c, srcs=list(v.segs), out=out)
        self._emit(op)
        seg = Seg(out, c)
        self.values[id(t)] = Value([seg])
        return seg

    # ---------------------------------------------------------------- parameter helpers
    @staticmethod
    def _bn_fold(bn_node, bias_name, cout, pad_to=None):
        """-> (scale_fn, shift_fn): y = acc*scale + shift  ==  BN(acc + bias)."""
        n = cout if pad_to is None else pad_to

        def parts(wd):
            if bn_node is not None:
                p = bn_node.name + '/'
                g, b = wd[p + 'gamma'].astype(np.float64), wd[p + 'beta'].astype(np.float64)
                m, v = wd[p + 'moving_mean'].astype(np.float64), wd[p + 'moving_variance'].astype(np.float64)
                scale = g / np.sqrt(v + bn_node.attrs['epsilon'])
                shift = b - m * scale
            else:
                scale, shift = np.ones(cout), np.zeros(cout)
            if bias_name is not None:
                shift = shift + wd[bias_name].astype(np.float64) * scale
            return scale, shift

        def pad(a):
            out = np.zeros(n, np.float32)
            out[:cout] = a
            return out

        return (lambda wd: pad(parts(wd)[0])), (lambda wd: pad(parts(wd)[1]))

    def _absorb_bn_act(self, t):
        """Follow t -> [BatchNorm] -> [act]; returns (bn_node|None, act, last_tensor)."""
        bn = self._sole_consumer(t, 'batchnorm')
        if bn is not None:
            t = bn.output
        a = self._sole_consumer(t, 'act')
        act = 'none'
        if a is not None:
            act, t = a.attrs['kind'], a.output
        return bn, act, t

    # ---------------------------------------------------------------- lowering
    def compile(self):
        order = self._topo()
        for n in order:
            self.param_shapes.update(n.params)
            if n.params:
                self.layer_seq[n.name] = n.seq
        x = self.inputs
        h, w, c = x.shape
        in_buf = self._new_buf(h, w, c, ld=c, external_slot=0, name='images', dtype=rt.dtype_id(getattr(x, 'dtype', 'float32')))   # float32, or uint8 image bytes
        in_buf.first_def = -1
        self.values[id(x)] = Value([Seg(in_buf, c)])
        done = set()
        for n in order:
            if id(n) in done:
                continue
            getattr(self, '_lower_' + n.op)(n, done)
        outs = []
        for o in self.outputs:
            v = self.values[id(o)]
            if not v.plain or v.segs[0].buf.external_slot < 1:
                raise NotImplementedError('model outputs must be produced by a 1x1 convolution')
            outs.append(v.segs[0].buf)
        ops = self.ops
        if self.fuse:
            if FOLD_PROJ:
                ops = fold_projection_into_consumers(ops, set(b.id for b in outs))
            if HOIST_UPSAMPLE:
                ops = hoist_upsampled_sources(ops, self.bufs)
            if POOL_IN_PRODUCER:
                ops = pool_into_producers(ops, self.bufs, set(b.id for b in outs))
            if FOLD_WSUM:
                ops = fold_weighted_sum(ops, set(b.id for b in outs), self.V)
            # fuse == 'latency': the plan for batches of a few images.  A fused inverted-residual block is ONE long
            # workgroup chain (block_4 at batch 1: 14 workgroups x 60 us) where its three unfused launches take 10 us
            # each, and a merged SE launch pools 2704 pixels in one workgroup: at batch <= 4 the unfused plan is
            # 20 % faster end to end (0.92 -> 0.73 ms at batch 1), from batch 8 on the fused one wins.  The squeeze as
            # partial sums of the depthwise kernel has no such workgroup: it is kept (one launch fewer per SE block).
            latency = self.fuse == 'latency'
            if MERGE_SE_MEAN and (not latency or SE_PARTIALS):
                ops = merge_se_mean(ops, only_after_depthwise=latency)
                if SE_PARTIALS:
                    ops = se_partials_from_depthwise(ops, self.bufs)
            # fuse == 'mid': the throughput plan WITHOUT the weight-streaming block form, for the batches between the few-image plans and
            # Model.mbk_batch.  That form is one workgroup per CU whose lifetime is a chain of tile pairs - as long at 8 images as at 64 -;
            # the forms it replaced cut a small batch into more, shorter workgroups (tools/lat_sweep.py, p50 of a step in ms with / without
            # it: 8 images 0.826 / 0.745, 16: 0.925 / 0.896, 32: 1.293 / 1.338).  The few-image plans ('nohead', 'nohead_k') do not take it either.
            ops = fuse_inverted_residuals(ops, set(b.id for b in outs), blocks=not latency, dtype=self.dtype, bufs=self.bufs, nosplit=self.nosplit,
                                          mbk=self.fuse is True)
            # fuse == 'nohead': the float32 plan for a few images - the throughput plan without YR_OP_HEAD (a head block's conv + depthwise
            # in one launch is one long chain per workgroup: at batch 1 td1 takes 43 us against 18 + 11 for its two launches;
            # tools/lat_variants.sh, round 5: p50 @416 batch 1 / 2 / 4 / 8 = 0.606 / 0.622 / 0.666 / 0.767 ms against 0.648 / 0.656 / 0.687 / 0.766)
            if FUSE_HEAD and not latency and self.fuse not in ('nohead', 'nohead_k'):
                ops = fuse_head_blocks(ops, self.bufs, set(b.id for b in outs), nosplit=self.nosplit, stream_ok=self.fuse is True)     # ('mid': the forms of round 5 - one chain per workgroup is as long at 8 images as at 64)
            if SE_TAIL:
                ops = se_tail_into_producers(ops)
            fold = FOLD_DW if isinstance(FOLD_DW, str) else ('1' if FOLD_DW else '0')   # (tests assign booleans)
            if fold != '0' and not latency and self.dtype == 0:
                ops = fold_depthwise_into_project(ops, set(b.id for b in outs), 0 if fold == '1' else FOLD_DW_MIN_PIXELS)
            if latency and self.dtype == 0:
                # the plan for a few images keeps the float32-MFMA pointwise kernels: at 169 ... 2704 pixels per map a conv waits for
                # its round trips, not for the matrix pipe, and the split form's plane-cutting stage is pure overhead there
                # (batch 1 @416: p50 0.64 ms against 0.68).  se_reduced bit 16 of a POINTWISE op = "not the split form".
                for o in ops:
                    if o.kind == rt.OP_POINTWISE and not any(s_.xform == 'dw3' for s_ in o.srcs):
                        o.se_reduced |= 0x10000
            if latency and self.dtype != 0 and KSPLIT_MAX_PIXELS > 0:
                # the 16-bit plan for one or two images: its small maps' pointwise convs in the k-split form too (pointwise_h.hip: pwkh_kernel -
                # a gated projection of 1152 channels is 36 chunks behind each other in pwh_kernel: 9.8 us at one image)
                for o in ops:
                    if o.kind == rt.OP_POINTWISE and not any(s_.xform == 'dw3' for s_ in o.srcs):
                        if o.h * o.w * (4 if getattr(o, 'stride', 0) == 2 else 1) <= KSPLIT_MAX_PIXELS:
                            o.se_reduced |= 0x20000
            if self.fuse == 'nohead_k' and self.dtype == 0 and KSPLIT_MAX_PIXELS > 0:
                # fuse == 'nohead_k' (one or two images: Model.ksplit_batch): the 'nohead' plan whose small maps take the K-SPLIT form of the
                # split pointwise kernel (se_reduced bit 17; pointwise_split.hip:
                # pwk_kernel): at 169 .. 2704 pixels a conv is a few workgroups, each one latency chain of k chunks - there a workgroup
                # is one 16 x 16 tile and its four waves split the k range.  A property of the plan (the sums are grouped by wave).
                for o in ops:
                    if o.kind == rt.OP_POINTWISE and not any(s_.xform == 'dw3' for s_ in o.srcs) and not (o.se_reduced & 0x10000):
                        if o.h * o.w * (4 if getattr(o, 'stride', 0) == 2 else 1) <= KSPLIT_MAX_PIXELS:
                            o.se_reduced |= 0x20000
        for o in ops:     # (also without fusion: a float32 POINTWISE op named by Model.check_ranges keeps the float32 MFMA)
            if o.kind == rt.OP_POINTWISE and o.name in self.nosplit:
                o.se_reduced |= 0x10000
        if PW_STREAM and PW_SPLIT and (self.fuse is True or (self.fuse == 'mid' and PW_STREAM_MID)) and self.dtype == 0:
            # the throughput plan's 1x1 convs in the pixel-stationary form (pointwise_stream.hip): HBM-bound launches that the tiled kernel
            # ran at a third of the memory rate
            for o in ops:
                pw_stream_form(o)
            if self.fuse is True:
                ops = fuse_stream_pairs(ops, set(b.id for b in outs))
        plan = Plan(ops, self.bufs, in_buf, outs, self.param_shapes, self.inputs.shape, self.dtype)
        plan.layer_seq = dict(self.layer_seq)
        return plan

    def _lower_conv2d(self, n, done):
        x = n.inputs[0]
        k, s = n.attrs['k'], n.attrs['stride']
        cin, cout = x.shape[2], n.output.shape[2]
        kname = n.name + '/kernel'
        bias = n.name + '/bias' if n.attrs['use_bias'] else None
        if k == 3 and s == 2 and cin == 3:
            bn, act, last = self._absorb_bn_act(n.output)
            src = self._plain(x)
            if src.buf.ld != 3:
                raise NotImplementedError('stem expects the dense 3-channel image')
            out = self._out_buf_for(last, n.name)
            ldw = round_up(cout, 4)
            op = OpRec(rt.OP_STEM, n.name, act=act, h=last.shape[0], w=last.shape[1], cin=3, cout=cout, k=3,
                       stride=2, srcs=[src], out=out, macs=last.shape[0] * last.shape[1] * 27 * cout)

            def wfn(wd, kname=kname, ldw=ldw, cout=cout):
                o = np.zeros((27, ldw), np.float32)
                o[:, :cout] = wd[kname].reshape(27, cout)  # (ky,kx,ci) major == HWIO flattening
                return o
            sc, sh = self._bn_fold(bn, bias, cout, ldw)

            def wpair(wd, wfn=wfn, sc=sc, sh=sh, ldw=ldw):
                """[27][ldw] + BN -> [ldw/2][27 taps x 2, times the BN scale | 1 1 | shift 2] (stemblock's stem layout)"""
                rows = np.concatenate([(wfn(wd) * sc(wd)[None]).astype(np.float32), np.ones((1, ldw), np.float32), sh(wd)[None]])
                return np.ascontiguousarray(rows.reshape(29, ldw // 2, 2).transpose(1, 0, 2)).reshape(ldw // 2, 58)
            op.params = {'wgt': ((27, ldw), wfn), 'scale': ((ldw,), sc), 'shift': ((ldw,), sh), 'wgt2': ((ldw // 2, 58), wpair)}
            self._emit(op)
            self._finish(n, bn, last, out, done)
            return
        if k != 1 or s != 1:
            raise NotImplementedError('Conv2D %dx%d stride %d with %d input channels is not on the detection path'
                                      % (k, k, s, cin))
        # squeeze-excite FC pair?
        if x.shape[0] == 1 and x.shape[1] == 1 and self._try_se_fc(n, done):
            return
        bn, act, last = self._absorb_bn_act(n.output)
        res = None
        add = self._sole_consumer(last, 'add')
        add_node = None
        if add is not None:
            other = add.inputs[0] if add.inputs[1] is last else add.inputs[1]
            if id(other) in self.values:
                res = self._plain(other).buf
                add_node, last = add, add.output
        v = self.values[id(x)]
        segs = list(v.segs)
        if len(segs) > rt.YR_MAX_SRC:
            raise NotImplementedError('more than %d concatenated sources' % rt.YR_MAX_SRC)
        out = self._out_buf_for(last, n.name)
        hh, ww = last.shape[0], last.shape[1]
        op = OpRec(rt.OP_POINTWISE, n.name, act=act, h=hh, w=ww, cin=cin, cout=cout, srcs=segs, out=out,
                   res=res, gate=v.gate, macs=hh * ww * cin * cout)
        V = self.V
        kp = sum(round_up(sg.c, V) for sg in segs)

        def wfn(wd, kname=kname, segs=[sg.c for sg in segs], kp=kp, cout=cout, V=V):
            wk = wd[kname].reshape(-1, cout)  # [cin, cout]
            o = np.zeros((cout, kp), np.float32)
            d = kb = 0
            for c_ in segs:
                o[:, kb:kb + c_] = wk[d:d + c_].T
                d += c_
                kb += round_up(c_, V)
            return o
        op.params = {'wgt': ((cout, kp), wfn, self.dtype)}
        if bn is not None or bias is not None:
            sc, sh = self._bn_fold(bn, bias, cout)
            op.params['scale'] = ((cout,), sc)
            op.params['shift'] = ((cout,), sh)
        self._emit(op)
        self._finish(n, bn, last, out, done, add_node)

    def _finish(self, n, bn, last, out, done, add_node=None):
        """Mark absorbed nodes done and publish the op's output value."""
        t = n.output
        done.add(id(n))
        while t is not last:
            c = self.consumers[id(t)][0]
            done.add(id(c))
            t = c.output
        self.values[id(last)] = Value([Seg(out, last.shape[2])])

    def _try_se_fc(self, n, done):
        """mean -> conv(+bias) -> Swish -> conv(+bias) -> sigmoid (efficientnet.py:417-434)."""
        x = n.inputs[0]
        if x.node is None or x.node.op != 'mean' or not n.attrs['use_bias']:
            return False
        a1 = self._sole_consumer(n.output, 'act', kind='swish')
        if a1 is None:
            return False
        c2 = self._sole_consumer(a1.output, 'conv2d', k=1)
        if c2 is None or not c2.attrs['use_bias']:
            return False
        a2 = self._sole_consumer(c2.output, 'act', kind='sigmoid')
        if a2 is None:
            return False
        c, r = x.shape[2], n.output.shape[2]
        if c2.output.shape[2] != c:
            return False
        src = self._plain(x)
        ldc = round_up(c, 4)
        out = self._new_buf(1, 1, c, ld=round_up(c, self.V), name=n.name + ':gate', dtype=0)   # float32; ld covers the consumer's k-space
        op = OpRec(rt.OP_SE_FC, n.name, h=1, w=1, cin=c, cout=c, se_reduced=r, srcs=[src], out=out,
                   macs=2 * c * r)
        k1, b1, k2, b2 = n.name + '/kernel', n.name + '/bias', c2.name + '/kernel', c2.name + '/bias'

        r4 = round_up(r, 4)

        def w1(wd):       # [ldc][R4]: the Keras kernel [1, 1, C, R] as it is, rows padded to quads (se_tail.h: yr_se_fc_pair)
            o = np.zeros((ldc, r4), np.float32)
            o[:c, :r] = wd[k1].reshape(c, r)
            return o

        def bb1(wd):
            o = np.zeros(r4, np.float32)
            o[:r] = wd[b1]
            return o

        def w2(wd):
            o = np.zeros((r, ldc), np.float32)
            o[:, :c] = wd[k2].reshape(r, c)
            return o

        def bb2(wd):
            o = np.zeros(ldc, np.float32)
            o[:c] = wd[b2]
            return o
        op.params = {'wgt': ((ldc, r4), w1), 'b1': ((r4,), bb1), 'wgt2': ((r, ldc), w2),
                     'b2': ((ldc,), bb2)}
        self._emit(op)
        for m in (n, a1, c2, a2):
            done.add(id(m))
        self.values[id(a2.output)] = Value([Seg(out, c)])
        return True

    def _lower_depthwise(self, n, done):
        x = n.inputs[0]
        k, s = n.attrs['k'], n.attrs['stride']
        c = x.shape[2]
        bn, act, last = self._absorb_bn_act(n.output)
        src = self._plain(x)
        out = self._out_buf_for(last, n.name)
        ldc = round_up(c, self.V)
        op = OpRec(rt.OP_DEPTHWISE, n.name, act=act, h=last.shape[0], w=last.shape[1], cin=c, cout=c, k=k,
                   stride=s, srcs=[src], out=out, macs=last.shape[0] * last.shape[1] * k * k * c)
        kname = n.name + '/depthwise_kernel'

        def wfn(wd):
            o = np.zeros((k * k, ldc), np.float32)
            o[:, :c] = wd[kname].reshape(k * k, c)
            return o
        sc, sh = self._bn_fold(bn, None, c, ldc)
        op.params = {'wgt': ((k * k, ldc), wfn), 'scale': ((ldc,), sc), 'shift': ((ldc,), sh)}
        self._emit(op)
        self._finish(n, bn, last, out, done)

    def _lower_mean(self, n, done):
        x = n.inputs[0]
        src = self._plain(x)
        c = x.shape[2]
        out = self._new_buf(1, 1, c, name=n.name, dtype=0)
        self._emit(OpRec(rt.OP_SE_MEAN, n.name, h=1, w=1, cin=c, cout=c, srcs=[src], out=out))
        done.add(id(n))
        self.values[id(n.output)] = Value([Seg(out, c)])

    def _lower_multiply(self, n, done):
        a, b = n.inputs
        gate, x = (a, b) if a.shape[0] * a.shape[1] == 1 else (b, a)
        if gate.shape[:2] != (1, 1) or gate.shape[2] != x.shape[2]:
            raise NotImplementedError('Multiply is only supported as the squeeze-excite gate')
        g = self._plain(gate)
        xs = self._plain(x)
        done.add(id(n))
        self.values[id(n.output)] = Value([xs], gate=g.buf)

    def _lower_concat(self, n, done):
        segs = []
        for t in n.inputs:
            v = self.values[id(t)]
            if v.gate is not None:
                raise NotImplementedError('concat of an SE-gated tensor')
            segs.extend(v.segs)
        done.add(id(n))
        self.values[id(n.output)] = Value(segs)
        if len(segs) > rt.YR_MAX_SRC:
            self._plain(n.output)

    def _xform(self, n, done, xf):
        v = self.values[id(n.inputs[0])]
        if v.gate is not None or any(s.xform != 'identity' for s in v.segs):
            seg = self._plain(n.inputs[0])
            v = Value([seg])
        done.add(id(n))
        self.values[id(n.output)] = Value([Seg(s.buf, s.c, xf) for s in v.segs])

    def _lower_upsample2(self, n, done):
        self._xform(n, done, 'up2')

    def _lower_maxpool(self, n, done):
        size = n.attrs['size']
        if size not in (2, 4):
            raise NotImplementedError('MaxPooling2D size %d' % size)
        h, w = n.inputs[0].shape[:2]
        if h % size or w % size:
            raise NotImplementedError('MaxPooling2D on a size not divisible by the pool')
        self._xform(n, done, 'maxpool%d' % size)

    def _lower_wsum(self, n, done):
        segs = []
        for t in n.inputs:
            v = self.values[id(t)]
            if len(v.segs) != 1 or v.gate is not None:
                segs.append(self._plain(t))
            else:
                segs.append(v.segs[0])
        h, w, c = n.output.shape
        out = self._out_buf_for(n.output, n.name)
        aname = n.name + '/alpha'
        op = OpRec(rt.OP_WSUM, n.name, h=h, w=w, cin=c, cout=c, srcs=segs, out=out)
        op.params = {'wgt': ((4,), lambda wd: wd[aname])}
        self._emit(op)
        done.add(id(n))
        self.values[id(n.output)] = Value([Seg(out, c)])

    def _lower_reshape5(self, n, done):
        done.add(id(n))
        self.values[id(n.output)] = self.values[id(n.inputs[0])]

    def _lower_batchnorm(self, n, done):
        raise NotImplementedError('BatchNormalization %s does not follow a convolution' % n.name)

    def _lower_act(self, n, done):
        raise NotImplementedError('activation %s does not follow a fused convolution' % n.name)

    def _lower_add(self, n, done):
        raise NotImplementedError('Add %s does not follow a fused 1x1 convolution' % n.name)


def compile_graph(inputs, outputs, fuse=True, dtype=0, nosplit=frozenset()):
    return Compiler(inputs, outputs, fuse, dtype, nosplit).compile()


SPLIT_LIMIT = 60000.0     # |operand| a split-form op accepts (float16's largest finite value is 65504)

# The same convolution has a different op name in different plan variants: a detection-head block is '<x>_head' where YR_OP_HEAD
# fuses it and '<x>_conv' (+ '<x>_mb_depthwise') where it does not ('nohead', 'nohead_k', 'latency'); an inverted-residual block is
# '<b>_mbr' (one launch), '<b>_mbe' (+ '<b>_project') or '<b>_expand' (...) depending on which fused form takes it.  A name
# Model.check_ranges reports from ONE variant must move the op off the split forms in ALL of them (a model guarded at
# batch 64 and then called at batch 1 ran the unfused conv in the split form).
_NOSPLIT_GROUPS = (('_head', '_conv'), ('_mbr', '_mbe', '_expand'))


def nosplit_aliases(names):
    """names -> the same set plus every name the same convolution carries in another plan variant."""
    out = set(names)
    for n in names:
        for grp in _NOSPLIT_GROUPS:
            for suf in grp:
                if n.endswith(suf):
                    out.update(n[:-len(suf)] + other for other in grp)
    return frozenset(out)


def split_form_ops(plan):
    """Indices of the ops of a float32 plan whose GEMM operands travel as float16 planes (|x| < 65504 required): POINTWISE ops at
    least 16 channels deep without the keep-float32 flag (and no depthwise-folded source), MBR / MBE with k bit 7, HEAD."""
    idx = []
    for i, o in enumerate(plan.ops):
        if o.dtype != 0:
            continue
        if o.kind == rt.OP_POINTWISE:
            kp = sum(round_up(s_.c, 4) for s_ in o.srcs if s_.xform != 'up2_add')
            if PW_SPLIT and kp >= 16 and not (o.se_reduced & 0x10000) and not any(s_.xform == 'dw3' for s_ in o.srcs):
                idx.append(i)
        elif o.kind in (rt.OP_MBR, rt.OP_MBE) and o.k & 0x80:
            idx.append(i)
        elif o.kind == rt.OP_HEAD:
            idx.append(i)
    return idx
