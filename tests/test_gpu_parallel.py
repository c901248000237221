"""The N>1 exchange on real hardware with ONE rank: an RCCL ('nccl' backend) process group of world size 1, the
all-gather of the detection records issued for real (DetectionGatherer(always=True)) - plain and overlapped on the
second stream with double-buffered records.  The driver's GPU test run thereby loads librccl and checks
the gathered bytes; the world-size-2 logic is covered on CPU by tests/test_parallel_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import params
from tests.util import ANCHORS

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_single_rank_rccl_all_gather_of_detections(dev):
    import torch.distributed as dist
    from yoloret_amd import layers as L
    from yoloret_amd.parallel import DetectionGatherer
    from yoloret_amd.pipeline import DetectionPipeline
    from yoloret_amd.weights import synthetic_weights
    from yoloret_amd.yolo3.model import yolov3_body
    assert not dist.is_initialized()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1,
                            device_id=dev)
    try:
        m = yolov3_body(L.Input(shape=[96, 96, 3]), 'mobilenetv2x75', 3, num_classes=20)
        m.set_weights(synthetic_weights(m, 5, 'survey'))
        pipe = DetectionPipeline(m, ANCHORS, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, record_slots=2)
        g = DetectionGatherer(always=True)
        assert g.world == 1 and g.always
        b = 4
        xs = [torch.from_numpy(params.synthetic_images(b, 96, 96, seed=s)).to(dev) for s in (1, 2, 3)]
        hw = torch.tensor([[96, 96]] * b, dtype=torch.int32, device=dev)
        # reference: every batch on its own, no collective
        want = []
        for x in xs:
            det, cnt = pipe(x, hw)
            torch.cuda.synchronize()
            want.append((det.cpu().numpy().copy(), cnt.cpu().numpy().copy()))
        assert sum(int(c.sum()) for _, c in want) > 0 and not np.array_equal(want[0][0], want[1][0])
        # plain collective
        det, cnt = pipe(xs[0], hw)
        ad, ac = g(det, cnt, pipe.record)
        torch.cuda.synchronize()
        assert np.array_equal(ad.cpu().numpy(), want[0][0]) and np.array_equal(ac.cpu().numpy(), want[0][1])
        # overlapped: step i's records travel while step i+1 runs; each handle is waited for one step late
        handles, got = [], []
        for x in xs:
            det, cnt = pipe(x, hw)
            handles.append(g.start(det, cnt, pipe.record))
            if len(handles) >= 2:
                d, c = handles[-2].wait()
                got.append((d.clone(), c.clone()))
        d, c = handles[-1].wait()
        got.append((d.clone(), c.clone()))
        torch.cuda.synchronize()
        for (d, c), (wd, wc) in zip(got, want):
            assert np.array_equal(d.cpu().numpy(), wd) and np.array_equal(c.cpu().numpy(), wc)
        # two steps in flight (depth 2: each step on its own stream / workspace / buffers) with the overlapped collective
        # reading the records behind the step's own completion event; a context runs again only after the collective that
        # read its records is done (release).  Five steps over three different batches: every context is reused.
        pipe2 = DetectionPipeline(m, ANCHORS, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=2)
        handles, got = [], []
        order = [0, 1, 2, 0, 2]
        for i in order:
            det, cnt = pipe2(xs[i], hw)
            h = g.start(det, cnt, pipe2.record, after=pipe2.done)
            pipe2.release(h.released)
            handles.append(h)
            if len(handles) >= 2:
                d, c = handles[-2].wait()
                got.append((d.clone(), c.clone()))
        d, c = handles[-1].wait()
        got.append((d.clone(), c.clone()))
        torch.cuda.synchronize()
        for (d, c), i in zip(got, order):
            assert np.array_equal(d.cpu().numpy(), want[i][0]) and np.array_equal(c.cpu().numpy(), want[i][1])
        # BASELINE config 2 as bench.py --force-dist runs it: 64 images @416 per step, THREE steps in flight, the collective
        # behind every step's own completion event, the context released by the gatherer itself (start(pipeline=...))
        size, b = 416, 64
        m2 = yolov3_body(L.Input(shape=[size, size, 3]), 'mobilenetv2x75', 3, num_classes=20)
        m2.set_weights(synthetic_weights(m2, 1234, 'survey'))
        xs = [torch.from_numpy(params.synthetic_images(b, size, size, seed=s)).to(dev) for s in (31, 32, 33)]
        hw = torch.tensor([[size, size]] * b, dtype=torch.int32, device=dev)
        ser = DetectionPipeline(m2, ANCHORS, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
        want = []
        for x in xs:
            det, cnt = ser(x, hw)
            torch.cuda.synchronize()
            want.append((det.cpu().numpy().copy(), cnt.cpu().numpy().copy()))
        pipe3 = DetectionPipeline(m2, ANCHORS, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=3)
        handles, got = [], []
        order = [0, 1, 2, 2, 0, 1, 1]
        for i in order:
            det, cnt = pipe3(xs[i], hw)
            handles.append(g.start(det, cnt, pipe3.record, after=pipe3.done, pipeline=pipe3))
            if len(handles) >= 2:                   # (a handle's result views live until the second start() after it)
                d, c = handles[-2].wait()
                got.append((d.clone(), c.clone()))
        d, c = handles[-1].wait()
        got.append((d.clone(), c.clone()))
        torch.cuda.synchronize()
        assert len(got) == len(order)
        for (d, c), i in zip(got, order):
            assert np.array_equal(d.cpu().numpy(), want[i][0]) and np.array_equal(c.cpu().numpy(), want[i][1]), i
    finally:
        dist.destroy_process_group()


def test_steps_in_flight_give_the_same_detections(dev):
    """DetectionPipeline(depth=3): consecutive calls on three execution contexts (stream, model workspace, buffers each),
    up to three steps in flight - the detections of every step equal those of the strictly serial pipeline, also when a
    context is reused while its neighbours are still running, and a 16-bit plan behaves the same."""
    from yoloret_amd import layers as L
    from yoloret_amd.pipeline import DetectionPipeline
    from yoloret_amd.weights import synthetic_weights
    from yoloret_amd.yolo3.model import yolov3_body
    for policy in ('float32', 'mixed_bfloat16'):
        L.set_global_policy(policy)
        try:
            m = yolov3_body(L.Input(shape=[128, 128, 3]), 'efficientnetb0-lite', 3, num_classes=20)
        finally:
            L.set_global_policy('float32')
        m.set_weights(synthetic_weights(m, 7, 'survey'))
        b = 6
        xs = [torch.from_numpy(params.synthetic_images(b, 128, 128, seed=s)).to(dev) for s in (11, 12, 13, 14)]
        hw = torch.tensor([[128, 128]] * b, dtype=torch.int32, device=dev)
        serial = DetectionPipeline(m, ANCHORS, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
        want = []
        for x in xs:
            det, cnt = serial(x, hw)
            torch.cuda.synchronize()
            want.append((det.cpu().numpy().copy(), cnt.cpu().numpy().copy()))
        assert sum(int(c.sum()) for _, c in want) > 0
        deep = DetectionPipeline(m, ANCHORS, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=3)
        order = [0, 1, 2, 3, 1, 0, 3, 2, 2]
        outs = []
        for i in order:
            det, cnt = deep(xs[i], hw)
            outs.append((det, cnt, deep.done, i))
            if len(outs) >= 3:                      # consume the step issued two calls ago, before its context is reused
                d, c, ev, j = outs[-3]
                ev.synchronize()
                assert np.array_equal(d.cpu().numpy(), want[j][0]) and np.array_equal(c.cpu().numpy(), want[j][1]), (policy, j)
        torch.cuda.synchronize()
        for d, c, ev, j in outs[-2:]:
            assert np.array_equal(d.cpu().numpy(), want[j][0]) and np.array_equal(c.cpu().numpy(), want[j][1]), (policy, j)


def _two_rank_worker(rank, world, port, q):
    """One of two processes sharing the GPU: its contiguous shard of the global batch through a depth-2 pipeline (two steps
    in flight on two streams), the records of both steps gathered over a 2-rank gloo group (host tensors)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from yoloret_amd import layers as L
    from yoloret_amd.parallel import DetectionGatherer, shard_range
    from yoloret_amd.pipeline import DetectionPipeline
    from yoloret_amd.weights import synthetic_weights
    from yoloret_amd.yolo3.model import yolov3_body
    dev = torch.device('cuda:0')
    m = yolov3_body(L.Input(shape=[96, 96, 3]), 'mobilenetv2x75', 3, num_classes=20)
    m.set_weights(synthetic_weights(m, 5, 'survey'))
    gb = 8
    lo, hi = shard_range(gb, rank, world)
    # every rank runs rank 0's tuning table (parallel.share_tuning): rank 0 tunes on its first call, rank 1 installs the table
    from yoloret_amd.parallel import share_tuning
    if rank == 0:
        m(torch.from_numpy(params.synthetic_images(gb, 96, 96, seed=1)[lo:hi]).to(dev))
        torch.cuda.synchronize()
    table = share_tuning(m, hi - lo, device=dev)
    assert table is not None and m.get_tuning(hi - lo, dev) == table and (0, hi - lo) in m._tuned
    pipe = DetectionPipeline(m, ANCHORS, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=2)
    g = DetectionGatherer()
    hw = torch.tensor([[96, 96]] * (hi - lo), dtype=torch.int32, device=dev)
    outs = [table]
    # consume as you go: step i is gathered before step i + 2 (which rewrites its context's buffers) is issued
    pend = []
    for seed in (1, 2, 3):
        x = torch.from_numpy(params.synthetic_images(gb, 96, 96, seed=seed)[lo:hi]).to(dev)
        if len(pend) == 2:
            det, cnt, done = pend.pop(0)
            done.synchronize()
            outs.append(tuple(t.clone() for t in g(det.cpu(), cnt.cpu())))    # (the gatherer's result buffers alternate)
        det, cnt = pipe(x, hw)
        pend.append((det, cnt, pipe.done))
    for det, cnt, done in pend:
        done.synchronize()
        outs.append(tuple(t.clone() for t in g(det.cpu(), cnt.cpu())))
    q.put((rank, [outs[0]] + [(d.numpy().copy(), c.numpy().copy()) for d, c in outs[1:]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_the_single_rank_records(dev):
    """Two processes on the one GPU (one shard each, depth-2 pipelines = four streams in all) + a 2-rank gloo all-gather of
    the packed records == the records of the whole batch run by a single serial pipeline, for three consecutive batches:
    sharding, steps in flight and the collective's row order together (the RCCL form of the collective is
    test_single_rank_rccl_all_gather_of_detections)."""
    import torch.multiprocessing as mp
    from yoloret_amd import layers as L
    from yoloret_amd.pipeline import DetectionPipeline
    from yoloret_amd.weights import synthetic_weights
    from yoloret_amd.yolo3.model import yolov3_body
    m = yolov3_body(L.Input(shape=[96, 96, 3]), 'mobilenetv2x75', 3, num_classes=20)
    m.set_weights(synthetic_weights(m, 5, 'survey'))
    serial = DetectionPipeline(m, ANCHORS, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
    hw = torch.tensor([[96, 96]] * 8, dtype=torch.int32, device=dev)
    want = []
    for seed in (1, 2, 3):
        det, cnt = serial(torch.from_numpy(params.synthetic_images(8, 96, 96, seed=seed)).to(dev), hw)
        torch.cuda.synchronize()
        want.append((det.cpu().numpy().copy(), cnt.cpu().numpy().copy()))
    assert sum(int(c.sum()) for _, c in want) > 0
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] and any(res[0][0]), 'both ranks must run rank 0\'s (non-trivial) tuning table'
    for rank in (0, 1):
        assert len(res[rank]) == 4
        for (d, c), (wd, wc) in zip(res[rank][1:], want):
            assert np.array_equal(d, wd) and np.array_equal(c, wc), 'rank %d' % rank
