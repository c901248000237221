"""Image-sharded data parallelism for the detection path: one process per GPU, weights
replicated, each rank runs forward+decode+NMS+pack on its contiguous slice of the global
batch, then ONE all-gather of the fixed-size detection records (RCCL over xGMI; backend
'nccl' is RCCL on ROCm, 'gloo' on CPU for tests).  SURVEY.md 8(e): the payload is ~9.6 KB per
image (C=20), i.e. latency-bound - a single flat collective per batch, never per image.

The reference has no inference-time collective (its only multi-GPU path is training under
tf.distribute.MirroredStrategy, reference code/train.py:55); this is the build's counterpart
for the north-star's "all-gather of detections".
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world_size):
    """Contiguous split: rank r owns images [r*B/W, (r+1)*B/W); B must divide evenly."""
    if global_batch % world_size:
        raise ValueError('global batch %d is not divisible by world size %d' % (global_batch, world_size))
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def share_tuning(model, batch, group=None, src=0, device=None):
    """Every rank runs the SAME tuning table: rank `src` has tuned `batch` images (its first call did: yr_autotune), the others install
    its table (Model.get_tuning / set_tuning) instead of timing their own.  Results do not depend on the table (engine.Model.__call__),
    speeds do - and a rank that measured beside a noisy neighbour would otherwise run a different step than its peers and the
    max-over-ranks step time is the slowest rank's.  The counterpart of "same variables on every replica" (reference code/train.py:55,
    tf.distribute.MirroredStrategy).  One small broadcast (an int per plan op); a no-op for a single process.  Returns the table."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return model.get_tuning(batch, device)
    rank = dist.get_rank(group)
    table = model.get_tuning(batch, device) if rank == src else None
    n = len(model.plan_for(batch).ops)
    on_gpu = dist.get_backend(group) == 'nccl'
    dev = (device if device is not None else torch.device('cuda', torch.cuda.current_device())) if on_gpu else torch.device('cpu')
    # entry 0: 1 if the source has a table (an untuned source leaves everybody untuned), then the n entries
    msg = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    if rank == src and table is not None:
        msg[0] = 1
        msg[1:] = torch.tensor(table, dtype=torch.int32)
    dist.broadcast(msg, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    got = [int(v) for v in msg.cpu().tolist()]
    if not got[0]:
        return None
    if rank != src:
        model.set_tuning(batch, got[1:], device)
    return got[1:]


class GatherHandle:
    """One all-gather in flight (DetectionGatherer.start).  wait() makes the CURRENT stream wait for it and returns
    (all_det [W*b,S,6], all_cnt [W*b]) - views of one of the gatherer's two result buffers, valid until the second
    start() after this one."""

    def __init__(self, work, event, out, world, b, s, six, ready=None, after=None):
        self._work, self._event, self._out = work, event, out
        self._after = after              # single rank, records produced on another stream: what wait() must wait for
        self.released = None             # overlapped form: fires when the collective no longer reads the send buffer
        self._dims = (world, b, s, six)
        self._ready = ready              # single rank without a collective: the local (det, det_count) as they are

    def elapsed_ms(self):
        """Milliseconds the collective took on its stream (DetectionGatherer(timing=True), overlapped form; after the events
        have fired - e.g. after a device synchronize); None otherwise."""
        t0 = getattr(self, '_t0', None)
        if t0 is None or self.released is None:
            return None
        return t0.elapsed_time(self.released)

    def wait(self):
        if self._ready is not None:
            if self._after is not None:
                torch.cuda.current_stream(self._ready[0].device).wait_event(self._after)
                self._after = None
            return self._ready
        if self._work is not None:
            self._work.wait()            # the current stream waits for the collective (no host block on GPU backends)
            self._work = None
        if self._event is not None:
            torch.cuda.current_stream(self._out.device).wait_event(self._event)
            self._event = None
        world, b, s, six = self._dims
        words = b * s * six + b
        out = self._out.view(world, words)
        return out[:, :b * s * six].reshape(world * b, s, six), out[:, b * s * six:].reshape(world * b)


class DetectionGatherer:
    """Preallocated all-gather of (det, det_count); result rows are in global image order.

    __call__ is the plain form (the result is ready for whatever is enqueued next on the current stream).
    start() / GatherHandle.wait() is the overlapped form of SURVEY.md 8(e): the collective is issued on a SECOND stream
    behind an event recorded after the step's pack kernel, so the next batch's forward runs while the records travel
    over xGMI; the caller waits for step i's handle after having enqueued step i+1.  Both the send side (the
    pipeline's record buffer, DetectionPipeline(record_slots=2)) and the result side are double buffered, so a step
    never overwrites bytes a collective in flight still reads or a consumer still holds."""

    def __init__(self, group=None, always=False, timing=False):
        """always=True issues the collective even for a 1-rank group (exercises RCCL on a single GPU).  timing=True: the
        overlapped form brackets every collective with a timing event pair on ITS stream (GatherHandle.elapsed_ms())."""
        self.timing = bool(timing)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.always = always and dist.is_initialized()
        self._out = [None, None]
        self._slot = 0
        self._send = None
        self._stream = None

    def _message(self, det, det_count, record):
        b, s, six = det.shape
        words = b * s * six + b
        if record is None or record.numel() != words:
            if self._send is None or self._send.numel() != words or self._send.device != det.device:
                self._send = torch.empty(words, dtype=torch.int32, device=det.device)
            self._send[:b * s * six].copy_(det.reshape(-1))
            self._send[b * s * six:].copy_(det_count)
            record = self._send
        return record, words

    def start(self, det, det_count, record=None, overlap=True, after=None, pipeline=None):
        """det int32 [b, S, 6], det_count int32 [b] (local shard); `record`: the flat buffer both are views of
        (DetectionPipeline.record) - sent as is; without it the two tensors are first staged into one message.
        after: the event that marks the records complete when they were produced on ANOTHER stream than the current one
        (DetectionPipeline(depth > 1).done); default: everything enqueued on the current stream so far.  The handle's
        `released` event (overlapped form) fires when the collective has read the records.  pipeline: the
        DetectionPipeline(depth > 1) that produced them - its context is released with that event here (what
        `pipeline.release(handle.released)` does by hand)."""
        ctx = pipeline._last if pipeline is not None else None   # (the context of the step just issued, before anything else runs)
        b, s, six = det.shape
        if self.world == 1 and not self.always:
            return GatherHandle(None, None, det, 1, b, s, six, ready=(det, det_count), after=after if det.is_cuda else None)
        if after is not None and det.is_cuda and (record is None or record.numel() != b * s * six + b or not overlap):
            # the staging copy of _message() and the plain collective run on the CURRENT stream: it must see complete records
            torch.cuda.current_stream(det.device).wait_event(after)
        record, words = self._message(det, det_count, record)
        slot = self._slot
        self._slot ^= 1
        out = self._out[slot]
        if out is None or out.numel() != self.world * words or out.device != det.device:
            out = self._out[slot] = torch.empty(self.world * words, dtype=torch.int32, device=det.device)
        if not (overlap and det.is_cuda):
            dist.all_gather_into_tensor(out, record, group=self.group)
            return GatherHandle(None, None, out, self.world, b, s, six)
        if self._stream is None or self._stream.device != det.device:
            from .pipeline import _shared_stream
            self._stream = _shared_stream(det.device, 'gather')
        ready = after
        if ready is None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(det.device))      # after the step's pack kernel
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(ready)
            record.record_stream(self._stream)
            out.record_stream(self._stream)
            t0 = None
            if self.timing:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(self._stream)
            work = dist.all_gather_into_tensor(out, record, group=self.group, async_op=True)
            work.wait()                                              # orders the side stream behind the collective
            done = torch.cuda.Event(enable_timing=self.timing)
            done.record(self._stream)
        h = GatherHandle(None, done, out, self.world, b, s, six)
        h.released = done
        h._t0 = t0
        if pipeline is not None:
            pipeline.release(done, ctx)
        return h

    def __call__(self, det, det_count, record=None):
        """det int32 [b, S, 6], det_count int32 [b] (local shard) -> ([W*b, S, 6], [W*b])."""
        if self.world == 1 and not self.always:
            return det, det_count
        return self.start(det, det_count, record, overlap=False).wait()
