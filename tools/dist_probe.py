"""Where the overlapped all-gather costs time with steps in flight (one RCCL rank on one GPU, config 2): the step loop of
bench.py with its parts switched on one by one.  GPU: python tools/dist_probe.py [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from yoloret_amd import layers as L
from yoloret_amd.parallel import DetectionGatherer
from yoloret_amd.pipeline import DetectionPipeline
from yoloret_amd.weights import synthetic_weights
from yoloret_amd.yolo3.model import yolov3_body
from yoloret_amd.yolo3.utils import get_anchors
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29581', rank=0, world_size=1, device_id=dev)
anchors = get_anchors('model_data/yolo_anchors.txt')
b, size = 64, 416
m = yolov3_body(L.Input(shape=[size, size, 3]), 'mobilenetv2x75', 3, num_classes=20)
m.set_weights(synthetic_weights(m, 1, 'survey'))
x = torch.rand((b, size, size, 3), device=dev)
hw = torch.tensor([[size, size]] * b, dtype=torch.int32, device=dev)
pipe = DetectionPipeline(m, anchors, 20, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=depth)
g = DetectionGatherer(always=True)
consumer = torch.cuda.Stream(dev)
def run(mode, n=60):
    pending = [None]
    def step():
        det, cnt = pipe(x, hw)
        if mode == 'none':
            return
        h = g.start(det, cnt, pipe.record, after=pipe.done)
        if mode in ('release', 'all') and h.released is not None:
            pipe.release(h.released)
        prev, pending[0] = pending[0], h
        if mode in ('wait', 'all') and prev is not None:
            with torch.cuda.stream(consumer):
                prev.wait()
    for _ in range(20): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return b * n / (time.perf_counter() - t)
for mode in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('none', 'start', 'release', 'wait', 'all', 'none')):
    print('depth %d  %-8s %.0f img/s' % (depth, mode, run(mode)))
dist.barrier()
torch.cuda.synchronize()
print('after a dist.barrier():')
for mode in ('all', 'none', 'all'):
    print('depth %d  %-8s %.0f img/s' % (depth, mode, run(mode)))
dist.destroy_process_group()
