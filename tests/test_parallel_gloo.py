"""N>1 path on CPU: world_size-2 gloo processes run the shard bookkeeping + the single all-gather of
packed detection records (yoloret_amd.parallel) and must reproduce the single-rank result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_records(global_batch, slots, seed=0):
    rng = np.random.default_rng(seed)
    det = rng.integers(0, 1000, (global_batch, slots, 6), dtype=np.int32)
    cnt = rng.integers(0, slots + 1, (global_batch,), dtype=np.int32)
    for i in range(global_batch):
        det[i, cnt[i]:] = 0
    return det, cnt


def _worker(rank, world, port, q, use_record):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from yoloret_amd.parallel import DetectionGatherer, shard_range
    det, cnt = _fake_records(8, 40)
    lo, hi = shard_range(8, rank, world)
    d, c = torch.from_numpy(det[lo:hi].copy()), torch.from_numpy(cnt[lo:hi].copy())
    record = None
    if use_record:
        record = torch.cat([d.reshape(-1), c])
        d = record[:d.numel()].view(d.shape)
        c = record[d.numel():]
    g = DetectionGatherer()
    for _ in range(2):  # second call reuses the preallocated buffers
        all_det, all_cnt = g(d, c, record)
    # the overlapped form: two gathers in flight, results live in alternating buffers until the second start() after them
    h1 = g.start(d, c, record)
    h2 = g.start(d, c, record)
    (d1, c1), (d2, c2) = h1.wait(), h2.wait()
    assert d1.data_ptr() != d2.data_ptr() and torch.equal(d1, all_det) and torch.equal(d2, all_det) and torch.equal(c1, c2)
    q.put((rank, all_det.numpy().copy(), all_cnt.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('use_record', [False, True])
def test_all_gather_of_detections_world2(use_record):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_record)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    det, cnt = _fake_records(8, 40)
    for rank, d, c in got:
        assert np.array_equal(d, det) and np.array_equal(c, cnt), 'rank %d gathered wrong records' % rank


def test_shard_range():
    from yoloret_amd.parallel import shard_range
    assert [shard_range(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_world1_is_identity():
    from yoloret_amd.parallel import DetectionGatherer
    d, c = torch.zeros((2, 4, 6), dtype=torch.int32), torch.zeros(2, dtype=torch.int32)
    a, b = DetectionGatherer()(d, c)
    assert a is d and b is c


# ---------------------------------------------------------------------------------------------------------------------
# The same path on REAL records: every rank turns its shard of one seeded batch of head logits into the pipeline's packed
# records (decode -> per-class NMS -> pack: here by the CPU oracle, the checker of the GPU kernels, standing in for them),
# the records go through shard_range + DetectionGatherer, and the product's unpack_detections() of the gathered bytes must give
# what the unsharded batch gives - boxes, score BITS and classes, image by image in global order.
REAL_B, REAL_HW, REAL_C, REAL_MAX = 8, (64, 96), 20, 20


def _real_records(lo, hi):
    sys.path.insert(0, ROOT)
    from oracle import cpost
    from yoloret_amd.yolo3.utils import get_anchors
    anchors = get_anchors(os.path.join(ROOT, 'yoloret_amd', 'model_data', 'yolo_anchors.txt'))
    rng = np.random.default_rng(2024)
    h, w = REAL_HW
    ys = [(rng.standard_normal((REAL_B, h // s, w // s, 3, REAL_C + 5)) * 2.0).astype(np.float32) for s in (32, 16, 8)]
    for y in ys:
        y[..., 4] -= 5.0          # few cells carry an object: a ragged batch (some classes empty, no class full)
    shapes = rng.integers(40, 200, (REAL_B, 2))
    slots = REAL_C * REAL_MAX
    det = np.zeros((hi - lo, slots, 6), np.int32)
    cnt = np.zeros(hi - lo, np.int32)
    for i in range(lo, hi):
        b, s, c, _ = cpost.yolo_eval([y[i] for y in ys], anchors, 3, REAL_C, shapes[i], REAL_MAX, 0.3, 0.5)
        k = len(s)
        det[i - lo, :k, 0:4] = b
        det[i - lo, :k, 4] = s.view(np.int32)          # the score's float32 bits ride in the int32 record
        det[i - lo, :k, 5] = c
        cnt[i - lo] = k
    return det, cnt


def _real_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from yoloret_amd.parallel import DetectionGatherer, shard_range
    from yoloret_amd.yolo3.model import unpack_detections
    lo, hi = shard_range(REAL_B, rank, world)
    det, cnt = _real_records(lo, hi)
    record = torch.cat([torch.from_numpy(det).reshape(-1), torch.from_numpy(cnt)])     # DetectionPipeline.record's layout
    d = record[:det.size].view(det.shape)
    c = record[det.size:]
    all_det, all_cnt = DetectionGatherer()(d, c, record)
    res = [(b.numpy().copy(), s.numpy().copy(), k.numpy().copy()) for b, s, k in unpack_detections(all_det, all_cnt)]
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_real_records_sharded_gather_unpack_equals_unsharded():
    from yoloret_amd.yolo3.model import unpack_detections
    det, cnt = _real_records(0, REAL_B)
    assert cnt.sum() > REAL_B and cnt.min() < cnt.max()      # a ragged batch of real detections
    want = unpack_detections(torch.from_numpy(det), torch.from_numpy(cnt))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in got:
        assert len(res) == REAL_B
        for i, ((b, s, k), (wb, ws, wk)) in enumerate(zip(res, want)):
            assert np.array_equal(b, wb.numpy()) and np.array_equal(s.view(np.int32), ws.numpy().view(np.int32)) and np.array_equal(k, wk.numpy()), \
                'rank %d, image %d' % (rank, i)


class _FakeTunedModel:
    """Stands in for engine.Model in share_tuning (which needs a GPU): a plan of `n` ops and a tuning table per batch size."""

    def __init__(self, n, tables):
        self._n, self.tables, self.installed = n, dict(tables), []

    def plan_for(self, batch):
        class P:
            pass
        p = P()
        p.ops = [None] * self._n
        return p

    def get_tuning(self, batch, device=None):
        return self.tables.get(batch)

    def set_tuning(self, batch, table, device=None):
        if len(table) != self._n:
            raise ValueError('table length')
        self.tables[batch] = list(table)
        self.installed.append(batch)


def _tuning_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from yoloret_amd.parallel import share_tuning
    n = 52
    own = [(7 * i + 100 * rank) % 29 << (8 if i % 3 else 0) for i in range(n)]       # every rank "tuned" something else
    m = _FakeTunedModel(n, {32: own})
    got = share_tuning(m, 32)
    untuned = share_tuning(_FakeTunedModel(n, {}), 16)     # the source has no table for this batch: nobody installs anything
    q.put((rank, got, m.tables[32], m.installed, untuned))
    dist.barrier()
    dist.destroy_process_group()


def test_share_tuning_installs_rank0_table_everywhere():
    """bench.py / a multi-GPU host: rank 0 tunes, every other rank installs ITS table (one broadcast of an int per plan op) instead of
    timing its own - the ranks then run identical steps.  (That results do not depend on the table at all is the GPU test
    tests/test_gpu_narrow.py::test_se_model_results_do_not_depend_on_the_tuning_table_or_the_batch.)"""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_tuning_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    t0 = [(7 * i) % 29 << (8 if i % 3 else 0) for i in range(52)]
    for rank, got, table, installed, untuned in res:
        assert got == t0 and table == t0 and untuned is None
        assert installed == ([] if rank == 0 else [32])
