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


class DetectionGatherer:
    """Preallocated all-gather of (det, det_count); result rows are in global image order."""

    def __init__(self, group=None, always=False):
        """always=True issues the collective even for a 1-rank group (exercises RCCL on a single GPU)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.always = always and dist.is_initialized()
        self._out = None
        self._send = None

    def __call__(self, det, det_count, record=None):
        """det int32 [b, S, 6], det_count int32 [b] (local shard) -> ([W*b, S, 6], [W*b]).
        `record`: the flat buffer both are views of (DetectionPipeline.record) - sent as is;
        without it the two tensors are first staged into one message."""
        if self.world == 1 and not self.always:
            return det, det_count
        b, s, six = det.shape
        words = b * s * six + b
        if record is None or record.numel() != words:
            if self._send is None or self._send.numel() != words or self._send.device != det.device:
                self._send = torch.empty(words, dtype=torch.int32, device=det.device)
            self._send[:b * s * six].copy_(det.reshape(-1))
            self._send[b * s * six:].copy_(det_count)
            record = self._send
        if self._out is None or self._out.numel() != self.world * words or self._out.device != det.device:
            self._out = torch.empty(self.world * words, dtype=torch.int32, device=det.device)
        dist.all_gather_into_tensor(self._out, record, group=self.group)
        out = self._out.view(self.world, words)
        all_det = out[:, :b * s * six].reshape(self.world * b, s, six)
        all_cnt = out[:, b * s * six:].reshape(self.world * b)
        return all_det, all_cnt
