#!/usr/bin/env python
"""Headline bench: images/sec of the detection forward path (backbone -> RFCR -> heads ->
decode -> per-class NMS -> packed detections [-> all-gather when N>1]) on synthetic batches that
are already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]:
MobileNetV2-0.75x @416, batch 64, fp32, random weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job img/s,
`roofline` for the dominant kernel symbol (measured live with hipEvents), `roofline_step` for the
whole step against the conv-granular algorithmic bytes of SURVEY.md 8(d), and `cpu_baseline` =
the oracle's torch-CPU port timed on this box's host cores (N=1, rank 0 only).
"""
import argparse
import json
import os
import re
import sys
import time

# Steps in flight run on several HIP streams (three execution contexts, the all-gather's side stream, RCCL's own, the consumer's):
# with the runtime's default of 4 hardware queues they share queues and a step waits behind another stream's collective
# (one RCCL rank, depth 3: 22-25k instead of 27.8k img/s, tools/dist_probe.py).  Read by the HIP runtime when it initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy peak 6290
FP32_PEAK_TFLOPS = 157.3  # fp32 vector == fp32 MFMA peak
MFMA16_PEAK_TFLOPS = 2500.0  # dense bf16 / f16 MFMA peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU per step (weak scaling)')
    ap.add_argument('--model', default='mobilenetv2x75')
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--classes', type=int, default=20)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'f16'],
                    help='element type of the activations between fused ops and of the 1x1-conv weights (BASELINE configs 3 / 5 '
                         'run bf16 / f16; the headline config 2 is f32)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the CPU baseline sample')
    ap.add_argument('--profile-iters', type=int, default=3)
    ap.add_argument('--per-op', action='store_true', help='also print the per-op table to stderr')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise RCCL and issue the detection all-gather even with one rank (single-GPU check of the N>1 path)')
    ap.add_argument('--depth', type=int, default=0,
                    help='steps in flight (DetectionPipeline(depth=..)): consecutive steps run on consecutive execution contexts / HIP '
                         'streams and fill each other\'s idle CUs; 1 = strictly one step after the other; 0 (default) = 3, with or without the '
                         'all-gather (one RCCL rank: depth 3 28.2k img/s against 28.3k without the collective, depth 2 26.9k)')
    ap.add_argument('--no-latency', action='store_true',
                    help='skip the batch-1 p50 loop (use under rocprofv3 so that every launch is a batch-%d launch)' % 64)
    ap.add_argument('--no-fp32-forms', action='store_true', help='skip the sub-run with every float32 GEMM on the float32 MFMA')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the short runs of BASELINE.json configs 3, 4, 5 (and the SE EfficientNets) that the default 1-GPU run attaches as `other_configs`')
    return ap.parse_args()


# BASELINE.json's other single-GPU workloads (the share of one GPU) and the reference's own
# squeeze-excite EfficientNets beside the build-defined `-lite` forms.  `alg_mb`: SURVEY 8(d)'s conv-granular bytes per image.
OTHER_CONFIGS = [
    ('c3', 'efficientnetb0-lite', 416, 128, 'bf16'), ('c4', 'mobilenetv2x14', 512, 64, 'f32'), ('c5', 'efficientnetb3-lite', 640, 32, 'f16'),
    ('c3_se', 'efficientnetb0', 416, 128, 'bf16'), ('c5_se', 'efficientnetb3', 640, 32, 'f16'),
]


def other_configs(dev, anchors, classes, steps, depth, budget_s=75.0):
    """{name: {img_s (steps in flight), serial_img_s, roofline_step_frac, dtype, workload, ...}}: each configuration set up
    exactly like the headline (same weight recipe, same pipeline, autotuned tile table), `steps` timed steps with `depth`
    steps in flight and `steps` strictly serial ones.  Never `value`."""
    from yoloret_amd import layers as L
    from yoloret_amd import weights as W
    from yoloret_amd.pipeline import DetectionPipeline
    from yoloret_amd.yolo3.model import yolov3_body
    res = {}
    t_all = time.perf_counter()
    for tag, name, size, b, dt in OTHER_CONFIGS:
        if time.perf_counter() - t_all > budget_s:
            res[tag] = {'skipped': 'time budget of %.0f s for the other configurations used up' % budget_s}
            continue
        t_cfg = time.perf_counter()
        L.set_global_policy({'f32': 'float32', 'bf16': 'mixed_bfloat16', 'f16': 'mixed_float16'}[dt])
        try:
            m = yolov3_body(L.Input(shape=[size, size, 3]), name, 3, num_classes=classes)
        finally:
            L.set_global_policy('float32')
        m.set_weights(W.synthetic_weights(m, 1234, 'survey'))
        x = torch.from_numpy(W.synthetic_images(b, size, size, seed=20240416)).to(dev)
        hw = torch.tensor([[size, size]] * b, dtype=torch.int32, device=dev)
        out = {}
        for d in (depth, 1):
            pipe = DetectionPipeline(m, anchors, classes, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=d)
            for _ in range(12):     # set-up (allocation, tile autotuning on the first call) + clocks
                pipe(x, hw)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                pipe(x, hw)
            torch.cuda.synchronize(dev)
            out[d] = b * steps / (time.perf_counter() - t0)
            del pipe
        plan = m.plan
        n_boxes = sum(3 * (size // s) ** 2 for s in (32, 16, 8))
        alg_img = plan.algorithmic_bytes_per_image() + n_boxes * (classes + 5) * 4 + n_boxes * (4 + classes) * 4 + plan.weight_bytes() / b
        res[tag] = {'img_s': round(out[depth], 1), 'serial_img_s': round(out[1], 1), 'steps': steps, 'steps_in_flight': depth,
                    'roofline_step_frac': round(out[depth] * alg_img / (HBM_PEAK_GBS * 1e9), 4),
                    'serial_roofline_step_frac': round(out[1] * alg_img / (HBM_PEAK_GBS * 1e9), 4),
                    'hbm_roofline_img_s': round(HBM_PEAK_GBS * 1e9 / alg_img, 0), 'alg_bytes_per_image': int(alg_img),
                    'dtype': dt, 'launches_per_step': len(plan.ops) + 4,
                    'workload': '%s @%d, batch %d (one GPU\'s share), %s, C=%d, same recipe and pipeline as the headline' % (name, size, b, dt, classes),
                    'setup_and_run_s': round(time.perf_counter() - t_cfg, 1)}
        if dt == 'f32':   # the float32-compute axis (the contract figure of a float32 plan since round 5: see roofline_step)
            fl = 2.0 * plan.total_macs()
            res[tag]['fp32_frac'] = round(out[depth] * fl / 1e12 / FP32_PEAK_TFLOPS, 4)
            res[tag]['serial_fp32_frac'] = round(out[1] * fl / 1e12 / FP32_PEAK_TFLOPS, 4)
            res[tag]['fp32_roofline_img_s'] = round(FP32_PEAK_TFLOPS * 1e12 / fl, 0)
        res[tag]['fallback_ops'] = plan.fallback_ops()
        # what the step really moves and executes: the plan's minimum traffic (every launched op's sources + output),
        # the measured FETCH / WRITE traffic where a committed PMC pass of this configuration exists, and the multiply-adds by pipe
        try:
            rows = m.profile(x, iters=2)
            moved_img = sum(r.get('hbm_bytes', r['bytes']) for r in rows) / b + n_boxes * (classes + 5) * 4 + n_boxes * (4 + classes) * 4
            m16_img = sum(r.get('macs_mfma16', 0) for r in rows) / b
            m32_img = sum(r.get('macs_fp32', r['macs'] - r.get('macs_mfma16', 0)) for r in rows) / b
            pmc_img = pmc_bytes_per_image(rows, '_%s_%d_b%d_%s' % (name.replace('-', ''), size, b, dt), b)
            for key, rate in (('', out[depth]), ('serial_', out[1])):
                fr = {'hbm_moved_plan': rate * moved_img / 1e9 / HBM_PEAK_GBS, 'mfma16': rate * 2.0 * m16_img / 1e12 / MFMA16_PEAK_TFLOPS,
                      'fp32_executed': rate * 2.0 * m32_img / 1e12 / FP32_PEAK_TFLOPS}
                if pmc_img:
                    fr['hbm_moved_pmc'] = rate * pmc_img / 1e9 / HBM_PEAK_GBS
                res[tag][key + 'frac_hbm_moved_plan'] = round(fr['hbm_moved_plan'], 4)
                res[tag][key + 'frac_hbm_moved_pmc'] = round(fr['hbm_moved_pmc'], 4) if pmc_img else None
                res[tag][key + 'frac_mfma16_peak'] = round(fr['mfma16'], 4)
                res[tag][key + 'frac_fp32_peak_executed'] = round(fr['fp32_executed'], 4)
                res[tag][key + 'frac_executed'] = round(max(fr.values()), 4)       # the busiest resource by what is EXECUTED, not credited
            res[tag]['moved_bytes_per_image_plan'] = int(moved_img)
            res[tag]['moved_bytes_per_image_pmc'] = int(pmc_img) if pmc_img else None
        except Exception as e:      # (the extra fields are diagnostics: never fail the bench line for them)
            res[tag]['executed_fractions_error'] = str(e)[:160]
        if dt != 'f32':   # (tests/test_gpu_narrow.py, tests/test_gpu_fullbatch.py: measured against the float32 oracle on the conditioned recipe)
            res[tag]['accuracy_note'] = ('16-bit storage plan: logits are NOT within 1e-4 of the float32 reference - scaled max / mean logit error vs '
                                         'the float32 oracle and the share of its detections reproduced are in README.md (16-bit plans) and profiles/')
        del m, x, hw
        torch.cuda.empty_cache()
    return res


def pmc_bytes_per_image(rows, tag, b):
    """HBM bytes per image of one step from the newest committed PMC passes of a configuration (profiles/rNN_traffic<tag>.json: per kernel
    symbol (2 x FETCH_SIZE + WRITE_SIZE) per launch, tools/rocpd_summary.py) x the launches per step of every symbol in `rows`
    (Model.profile) - None unless every kernel of the step is covered (decode / NMS / pack are not in `rows`: the graph only)."""
    import glob
    try:
        tfile = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_traffic%s.json' % tag)))[-1]
        table = {canon_symbol(k): v for k, v in json.load(open(tfile)).items()}
    except (IndexError, OSError, ValueError):
        return None
    tot, missing = 0.0, 0
    for r in rows:
        want = canon_symbol(r['kernel'])
        t = table.get(want)
        if t is None and want.endswith('>'):
            hits = [v for k, v in table.items() if k.startswith(want[:-1] + ',')]
            t = hits[0] if len(hits) == 1 else None
        if t is None:
            # a tile shape the tuner picked in THIS run but not in the profiled one: the op's plan bytes stand in (at most a few ops)
            missing += 1
            if missing > max(3, len(rows) // 10):
                return None
            tot += r.get('hbm_bytes', r['bytes'])
            continue
        tot += t['traffic_bytes']
    return tot / b


def canon_symbol(name):
    """Kernel symbol in one spelling: no spaces, bools as 1/0, element types as f32/bf16/f16."""
    name = name.replace(' ', '').replace('true', '1').replace('false', '0')
    return name.replace('__bf16', 'bf16').replace('_Float16', 'f16').replace('float', 'f32')


def parked_share(symbol):
    """{'wait_any_share', 'source'} of `symbol` from the newest committed SQ pass (profiles/rNN_pmc_sq*.txt: rows of kernel, counter,
    calls, avg_per_launch, sum - tools/rocpd_summary.py), or None."""
    import glob
    want = canon_symbol(symbol)
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_sq*.txt')) + glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_*block*.txt')), reverse=True):
        vals = {}
        try:
            for line in open(f):
                m = re.match(r'^(\S.*?)\s+(SQ_WAVE_CYCLES|SQ_WAIT_ANY)\s+\d+\s+([0-9.]+)', line)
                if m and (canon_symbol(m.group(1)) == want or canon_symbol(m.group(1)).startswith(want[:-1] + ',')):
                    vals[m.group(2)] = float(m.group(3))
        except OSError:
            continue
        if 'SQ_WAVE_CYCLES' in vals and 'SQ_WAIT_ANY' in vals and vals['SQ_WAVE_CYCLES'] > 0:
            return {'wait_any_share': round(vals['SQ_WAIT_ANY'] / vals['SQ_WAVE_CYCLES'], 4), 'source': os.path.relpath(f, ROOT)}
    return None


def fp32_mfma_forms(a):
    """The same headline run with every float32 GEMM on the float32 MFMA (YOLORET_MBR_SPLIT=0 YOLORET_PW_SPLIT=0; the head-block
    kernels exist in the split form only: YOLORET_FUSE_HEAD=0) - for a strict reader of `dtype: f32`.  A sub-process: the library
    reads the switches once."""
    import subprocess
    env = dict(os.environ, YOLORET_MBR_SPLIT='0', YOLORET_PW_SPLIT='0', YOLORET_FUSE_HEAD='0', YOLORET_TUNE_CACHE='')
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', '20', '--warmup', str(a.warmup), '--batch', str(a.batch), '--model', a.model, '--size', str(a.size),
           '--classes', str(a.classes), '--no-cpu-baseline', '--no-latency', '--no-other-configs', '--no-fp32-forms', '--depth', str(a.depth)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {'img_s': d['value'], 'serial_img_s': (d.get('serial_steps') or {}).get('img_s'), 'steps': 20,
                'switches': 'YOLORET_MBR_SPLIT=0 YOLORET_PW_SPLIT=0 YOLORET_FUSE_HEAD=0 (v_mfma_f32_16x16x4_f32 for every 1x1 convolution)'}
    except Exception as e:
        return {'error': str(e)[:200]}


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU
    box reports 256 logical CPUs but runs the job under a 16-CPU quota; oversubscribing it 16x made the
    oneDNN baseline ~100x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(model_name, size, classes, anchors, seconds, gpu_logits=None):
    """The oracle's torch-CPU port (oneDNN) + C decode/NMS on all host cores; bounded sample.  BASELINE.md section 3: the
    throughput sample in batches of 8 (`value`, plus the median of the per-batch times, at least 5 of them) and the
    batch-1 latency the reference itself runs at (code/yolo.py:83-84): p50 over 200 runs after 20 warm-ups (fewer when
    200 would not fit `seconds`; the count is stated)."""
    from oracle import cpost, params, torch_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    P = params.ParamStore(1234, 'survey')
    ref = torch_ref.TorchReference(P, model_name, 3, classes)
    b = 8
    x = params.synthetic_images(b, size, size)

    def one(xb):
        ys = ref(xb)
        for i in range(xb.shape[0]):
            cpost.yolo_eval([y[i] for y in ys], anchors, 3, classes, (size, size), 20, 0.2, 0.5)
    one(x)  # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    n = 0
    per_batch = []
    while True:
        t1 = time.perf_counter()
        one(x)
        per_batch.append(time.perf_counter() - t1)
        n += b
        dt = time.perf_counter() - t0
        if (dt >= seconds and len(per_batch) >= 5) or n >= 4096:
            break
    x1 = x[:1]
    for _ in range(20):
        one(x1)
    lat = []
    t0 = time.perf_counter()
    while len(lat) < 200 and (len(lat) < 20 or time.perf_counter() - t0 < seconds):
        t1 = time.perf_counter()
        one(x1)
        lat.append(time.perf_counter() - t1)
    parity = None
    if gpu_logits is not None:
        # the checker's other job in this leg: the BENCH's own workload (SURVEY 8(d) weight recipe, which amplifies rounding noise
        # ~1e3 x: DESIGN.md 5) is held against the oracle in float64 - the HIP path's error next to what a float32 CPU implementation
        # (this oracle in float32) shows against the same float64 values.  The 1e-4 bar itself is enforced on the variance-preserving
        # recipe (tests/test_gpu_graph.py); on this recipe no float32 implementation meets it against another.
        try:
            xs = x[:2]
            r64 = torch_ref.TorchReference(P, model_name, 3, classes, dtype=torch.float64)(xs.astype(np.float64))
            r32 = ref(xs)
            g = gpu_logits(xs)
            eg, e32 = [], []
            for a64, a32, ag in zip(r64, r32, g):
                den = np.maximum(1.0, np.abs(a64))
                eg.append(np.abs(np.asarray(ag, np.float64).reshape(a64.shape) - a64) / den)
                e32.append(np.abs(np.asarray(a32, np.float64) - a64) / den)
            parity = {'recipe': 'survey (the bench workload)', 'images': 2,
                      'hip_logit_err_vs_fp64': {'max': float('%.3g' % max(e.max() for e in eg)), 'mean': float('%.3g' % np.mean([e.mean() for e in eg]))},
                      'cpu_fp32_logit_err_vs_fp64': {'max': float('%.3g' % max(e.max() for e in e32)), 'mean': float('%.3g' % np.mean([e.mean() for e in e32]))},
                      'note': 'scaled |y - y64| / max(1, |y64|); the 1e-4 bar of north_star is enforced on the conditioned recipe (tests), where both figures are ~1e-5'}
        except Exception as e:
            parity = {'error': str(e)[:160]}
    return {'value': round(n / dt, 2), 'unit': 'img/s', 'cores': cores, 'cpu': cpu_model(), 'kind': 'port', 'workload_parity': parity,
            'median_ms_b8': round(float(np.median(per_batch)) * 1e3, 2), 'runs_b8': len(per_batch),
            'p50_ms_b1': round(float(np.median(lat)) * 1e3, 2), 'runs_b1': len(lat),
            'sample': '%d images (batches of %d) of the same %s@%d workload through oracle/torch_ref.py '
                      '(torch-CPU/oneDNN fp32) + oracle C decode/NMS, %.1f s; then %d batch-1 runs after 20 warm-ups'
                      % (n, b, model_name, size, dt, len(lat))}


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d' % (a.gpus, a.gpus))
    import torch.distributed as dist
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or a.force_dist
    if a.depth <= 0:
        a.depth = 3   # (also with the all-gather in every step, since the streams have their own hardware queues: GPU_MAX_HW_QUEUES above)
    # RCCL prints a banner (hostname, library path, ...) on STDOUT when the first communicator is created; stdout
    # must carry exactly one JSON line, so fd 1 points at stderr until the set-up step (which runs the first
    # collective) is over
    saved_stdout = None
    if use_dist:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from yoloret_amd import layers as L
    from yoloret_amd import weights as W
    from yoloret_amd.parallel import DetectionGatherer
    from yoloret_amd.pipeline import DetectionPipeline
    from yoloret_amd.yolo3.model import yolov3_body
    from yoloret_amd.yolo3.utils import get_anchors

    anchors = get_anchors('model_data/yolo_anchors.txt')
    L.set_global_policy({'f32': 'float32', 'bf16': 'mixed_bfloat16', 'f16': 'mixed_float16'}[a.dtype])
    model = yolov3_body(L.Input(shape=[a.size, a.size, 3]), a.model, 3, num_classes=a.classes)
    model.set_weights(W.synthetic_weights(model, 1234, 'survey'))
    pipe = DetectionPipeline(model, anchors, a.classes, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5,
                             record_slots=2 if (use_dist and a.depth <= 1) else 1, depth=a.depth)
    # latency, ingestion and per-kernel measurements run strictly one step after the other
    pipe1 = pipe if a.depth <= 1 else DetectionPipeline(model, anchors, a.classes, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5)
    gather = DetectionGatherer(always=a.force_dist, timing=use_dist)
    timed_handles = []
    b = a.batch
    x = torch.from_numpy(W.synthetic_images(b, a.size, a.size, seed=20240416 + rank)).to(dev)
    image_hw = torch.tensor([[a.size, a.size]] * b, dtype=torch.int32, device=dev)
    # one resident batch PER STEP IN FLIGHT (different images each): consecutive steps do not re-read one input that sits in the
    # 256 MiB Infinity Cache (the 133 MB batch of a single buffer did)
    xs_res = [x] + [torch.from_numpy(W.synthetic_images(b, a.size, a.size, seed=20240416 + rank + 1000 * (i + 1))).to(dev) for i in range(max(a.depth, 1) - 1)]
    turn_res = [0]

    pending = [None]
    consumer = torch.cuda.Stream(dev) if a.depth > 1 else torch.cuda.current_stream(dev)

    def step():
        # N > 1: the all-gather of step i's records runs on a second stream while step i+1's forward is enqueued; its
        # handle is waited for one step later (the final sync() covers the last one).  Every step still contains
        # exactly one collective.
        xi = xs_res[turn_res[0] % len(xs_res)]
        turn_res[0] += 1
        det, cnt = pipe(xi, image_hw)
        # (pipeline=pipe: the context that produced these records runs again only after the collective has read them)
        h = gather.start(det, cnt, pipe.record, after=pipe.done, pipeline=pipe)
        if timed_handles is not None and use_dist:
            timed_handles.append(h)
        prev, pending[0] = pending[0], h
        if prev is None:
            return None
        # the consumer of the gathered detections waits on ITS stream: on the launch stream the wait would sit in front of
        # the next step's `ready` event and tie every step to the collective two steps back (measured: 24.7k vs 28.4k img/s)
        with torch.cuda.stream(consumer):
            return prev.wait()

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if use_dist:
        # every rank runs rank 0's tuning table (parallel.share_tuning): results never depend on the table, speeds do, and the job's
        # step time is its slowest rank's - a rank that tuned beside a noisy neighbour must not run a different step than its peers
        from yoloret_amd.parallel import share_tuning
        if rank == 0:
            model(x)
            torch.cuda.synchronize(dev)
        share_tuning(model, b, device=dev)
    step()  # one-time setup outside every timed/warm-up count: buffer allocation + per-layer tile autotuning
    sync()
    # The set-up also brings the device to its sustained clocks: the tuner's ~0.5 s of launches do that as a side
    # effect, a cached tile table (YOLORET_TUNE_CACHE) would skip them and measure 3-4 % low (measured: 22.0k vs
    # 21.2k img/s).  A fixed 80 untimed steps (about a quarter second; a COUNT, not a duration: with N > 1 every
    # step is a collective, so all ranks must run the same number) make both paths start from the same state.
    for _ in range(80):
        step()
    sync()
    if saved_stdout is not None:
        sys.stdout.flush()
        try:   # RCCL's banner sits in the C library's stdout buffer (fully buffered on a pipe): it must leave through the
            import ctypes   # redirected descriptor NOW, or it would follow the JSON line at exit
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    for _ in range(a.warmup):
        step()
    sync()
    del timed_handles[:]      # (only the timed steps' collectives are kept)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, t)           # every rank's own clock around the same timed steps ...
        per_rank = [float(v) for v in allt.cpu()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)       # ... the job's time is the slowest rank's
        elapsed = float(t.item())
    ms_per_step = elapsed / a.steps * 1e3
    value = world * b * a.steps / elapsed

    out = None
    if rank == 0:
        plan = model.plan
        n_boxes = pipe1.n
        alg_fwd = plan.algorithmic_bytes_per_image()
        alg_dec = n_boxes * (a.classes + 5) * 4 + n_boxes * (4 + a.classes) * 4
        alg_img = alg_fwd + alg_dec + plan.weight_bytes() / b
        flops_img = 2.0 * plan.total_macs()
        # ---- live per-kernel measurement (hipEvent pair around every launch, same stream)
        prof = model.profile(x, iters=max(1, a.profile_iters))   # (--profile-iters 0: one pass, the line needs its per-kernel table)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ys = pipe1.forward(x)
        post_reps = []
        reps = 7          # (the median of seven: one pass disturbed by the tail of the steps in flight made NMS the "dominant kernel" once)
        torch.cuda.synchronize(dev)
        for _ in range(reps):
            v = pipe1._buffers(b, dev)
            ev[0].record()
            from yoloret_amd import runtime as rt
            rt.decode([y for y in v['ys']], anchors, a.classes, image_hw, (a.size, a.size))
            ev[1].record()
            rt.nms(v['boxes'], v['scores'], 20, 0.2, 0.5)
            ev[2].record()
            rt.pack_detections(v['boxes'], v['scores'], v['idx'], v['cnt'])
            ev[3].record()
            torch.cuda.synchronize(dev)
            post_reps.append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
        post_ms = np.median(np.asarray(post_reps), axis=0)
        rows = prof + [
            dict(name='decode', kind='decode', kernel='decode_kernel', ms=post_ms[0], macs=0, bytes=alg_dec * b),
            dict(name='nms', kind='nms', kernel=('nms_band_kernel<%s>(+nms_lazy_kernel<1024> overflow pass)' % ('256,44' if n_boxes <= 11264 else '512,50' if n_boxes <= 25600 else '1024,38')) if n_boxes * 6 + 16 + 320 <= 150 * 1024 else 'nms_kernel',
                 ms=post_ms[1], macs=0,
                 bytes=(a.classes * n_boxes * 4 + n_boxes * 16) * b),
            dict(name='pack', kind='pack', kernel='pack_kernel', ms=post_ms[2], macs=0, bytes=0)]
        by = {}
        for r in rows:
            k = by.setdefault(r['kernel'], dict(ms=0.0, bytes=0, hbm=0, macs=0, macs16=0, macs32=0, launches=0))
            k['ms'] += r['ms']; k['bytes'] += r['bytes']; k['macs'] += r['macs']; k['launches'] += 1
            k['macs16'] += r.get('macs_mfma16', 0)
            k['macs32'] += r.get('macs_fp32', r['macs'] - r.get('macs_mfma16', 0))   # (split-form blocks: 3 float16 products per 1x1 multiply-add on the 16-bit pipe, the depthwise stage here)
            k['hbm'] += r.get('hbm_bytes', r['bytes'])
        dom = max(by, key=lambda k: by[k]['ms'])
        d = by[dom]
        avg_ms = d['ms'] / d['launches']

        def pipes(v):
            """One symbol (or family) against the three resources: HBM bytes it must move, its multiply-adds on the 16-bit
            matrix pipe (2.5 PFLOP/s dense) and those on the float32 pipe (fp32 MFMA / packed FMA: 157.3 TFLOP/s) - a fused
            16-bit block kernel has both kinds, and summing them against one peak means nothing."""
            sec = v['ms'] * 1e-3
            gbs = v['hbm'] / sec / 1e9 if sec else 0.0
            tf16 = 2.0 * v['macs16'] / sec / 1e12 if sec else 0.0
            tf32 = 2.0 * v['macs32'] / sec / 1e12 if sec else 0.0
            fr = {'hbm': gbs / HBM_PEAK_GBS, 'mfma16': tf16 / MFMA16_PEAK_TFLOPS, 'fp32': tf32 / FP32_PEAK_TFLOPS}
            if a.dtype == 'f32' and v['macs16']:
                # a float32 plan's SPLIT-form kernel (float16 planes on the 16-bit pipe): the contract's figure is its ALGORITHMIC
                # float32 multiply-adds against the dense float32 MFMA peak - the rate the float32 pipe would need for the same
                # arithmetic; what each pipe really executes stays in by_pipe (fp32_executed_*, mfma16_*: three products per multiply-add)
                fr['fp32_executed'] = fr['fp32']
                tf32 = 2.0 * v['macs'] / sec / 1e12 if sec else 0.0
                fr['fp32'] = tf32 / FP32_PEAK_TFLOPS
            return gbs, tf16, tf32, fr
        gbs, tf16, tf32, fr = pipes(d)
        pipe_name = max((k for k in fr if k != 'fp32_executed'), key=fr.get)
        # The dominant kernel against the roofline that bounds it (the resource with the largest fraction).  Bytes = what
        # the op must move through HBM (its sources + its output; a fused block kernel: the block's input + output only).
        if pipe_name == 'hbm':
            roofline = {'bound': 'hbm', 'kernel': dom, 'launches_per_step': d['launches'], 'avg_launch_ms': round(avg_ms, 4),
                        'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(fr['hbm'], 4), 'traffic': None}
        else:
            tfl, peak_tf = (tf16, MFMA16_PEAK_TFLOPS) if pipe_name == 'mfma16' else (tf32, FP32_PEAK_TFLOPS)
            roofline = {'bound': 'mfma', 'pipe': '16-bit MFMA' if pipe_name == 'mfma16' else 'float32 (fp32 MFMA / packed FMA)',
                        'kernel': dom, 'launches_per_step': d['launches'], 'avg_launch_ms': round(avg_ms, 4),
                        'achieved': round(tfl, 2), 'peak': peak_tf, 'unit': 'TFLOP/s', 'frac': round(tfl / peak_tf, 4), 'traffic': None,
                        'flops_per_launch': int(2.0 * (d['macs16'] if pipe_name == 'mfma16' else d['macs'] if 'fp32_executed' in fr else d['macs32']) / d['launches'])}
            if pipe_name == 'fp32' and 'fp32_executed' in fr:
                roofline['pipe'] = 'float32 arithmetic (algorithmic multiply-adds against the dense float32 MFMA peak; executed as three float16-plane products per multiply-add on the 16-bit matrix pipe + the depthwise stage on the float32 pipe: by_pipe)'
        roofline['by_pipe'] = {'hbm_gbs': round(gbs, 1), 'frac_hbm': round(fr['hbm'], 4), 'mfma16_tflops': round(tf16, 2),
                               'frac_mfma16': round(fr['mfma16'], 4), 'fp32_tflops': round(tf32, 2), 'frac_fp32': round(fr['fp32'], 4)}
        if 'fp32_executed' in fr:
            roofline['by_pipe']['frac_fp32_executed'] = round(fr['fp32_executed'], 4)
        # the same per kernel FAMILY: the lane-per-pixel front, the fused MFMA blocks ... are several symbols each
        fams = [('lane_per_pixel_front', ('mblane', 'stemblock', 'stem_')),
                ('head_blocks', ('head_kernel', 'head2_kernel', 'hwalk_kernel', 'hwalkh_kernel', 'hstream_kernel')),
                ('fused_blocks', ('mbh_kernel', 'mbn_kernel', 'mbr_kernel', 'mbk_kernel', 'mbe_kernel', 'mbx_kernel', 'mbxr_kernel', 'mbhr_kernel', 'mbhq_kernel', 'stemxr_kernel', 'stemxp_kernel')),
                ('pointwise', ('pw_kernel', 'pwd_kernel', 'pws_kernel', 'pwt_kernel', 'pwk_kernel', 'pwh', 'pwl')), ('depthwise', ('dw_kernel', 'dwp_kernel', 'dwq_kernel', 'dwl')),
                ('elementwise', ('wsum', 'gather', 'letterbox')),
                ('squeeze_excite', ('se_',)), ('postprocess', ('decode', 'nms', 'pack'))]
        known = tuple(p_ for _, ps in fams for p_ in ps)
        fams.append(('other', tuple(sym for sym in by if not sym.startswith(known)) or ('\0',)))   # (nothing may fall through: the shares add up to 1)
        total_ms = sum(v['ms'] for v in by.values())
        roofline_family = {}
        for fname, prefixes in fams:
            agg = dict(ms=0.0, hbm=0, macs=0, macs16=0, macs32=0, launches=0)
            for sym, v in by.items():
                if sym.startswith(prefixes):
                    for kk in agg:
                        agg[kk] += v[kk]
            if agg['launches']:
                g_, t16_, t32_, fr_ = pipes(agg)
                roofline_family[fname] = {'ms': round(agg['ms'], 4), 'share_of_step': round(agg['ms'] / total_ms, 4), 'launches': agg['launches'],
                                          'moved_gbs': round(g_, 1), 'frac_hbm': round(fr_['hbm'], 4), 'fp32_tflops': round(t32_, 2),
                                          'frac_fp32': round(fr_['fp32'], 4), 'mfma16_tflops': round(t16_, 2), 'frac_mfma16': round(fr_['mfma16'], 4)}
        roofline['bytes_per_launch'] = int(d['hbm'] / d['launches'])   # the `achieved` GB/s = this / avg_launch_ms
        # what the kernel really executes on its busiest resource (a split-form kernel is judged on ALGORITHMIC float32 FLOPs above)
        roofline['frac_executed'] = round(max(fr['hbm'], fr['mfma16'], fr.get('fp32_executed', fr['fp32'])), 4)
        roofline['parked'] = parked_share(dom)     # share of wave cycles waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) from the committed PMC pass, or None
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (profiles/rNN_traffic.json: (2*FETCH_SIZE + WRITE_SIZE) KiB, see tools/rocpd_summary.py); None if absent
        pmc_step_bytes = None
        try:
            import glob
            tag = '' if (a.model, a.size, a.batch, a.dtype) == ('mobilenetv2x75', 416, 64, 'f32') else \
                '_%s_%d_b%d_%s' % (a.model.replace('-', ''), a.size, a.batch, a.dtype)
            tfile = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_traffic%s.json' % tag)))[-1]
            # rocprof prints the full template argument list (bools as true/false, types by their C++ names); the
            # launchers name a kernel by its leading arguments: compare canonical forms, the launcher's as a prefix
            table = {canon_symbol(k): v for k, v in json.load(open(tfile)).items()}

            def lookup(sym):
                want = canon_symbol(sym)
                tr = table.get(want)
                if tr is None and want.endswith('>'):
                    hits = [v for k, v in table.items() if k.startswith(want[:-1] + ',')]
                    tr = hits[0] if len(hits) == 1 else None
                return tr
            tr = lookup(dom)
            if tr:
                roofline['traffic'] = tr['traffic_bytes']
                roofline['traffic_source'] = os.path.relpath(tfile, ROOT)
                roofline['alg_bytes_per_launch'] = int(d['bytes'] / d['launches'])   # conv-granular
            # the whole step's measured traffic: sum over symbols of (PMC bytes per launch x launches per step)
            tot, miss = 0.0, []
            for sym, v in by.items():
                names = [sym.split('(')[0], 'nms_lazy_kernel<1024>'] if sym.startswith('nms_band') else [sym]
                for nme in names:
                    t = lookup(nme)
                    if t:
                        tot += t['traffic_bytes'] * v['launches']
                    elif nme != 'nms_lazy_kernel<1024>':
                        miss.append(nme)
            if not miss:
                pmc_step_bytes = tot
        except (IndexError, OSError, ValueError):
            pass
        per_gpu = value / world
        step_s = b / per_gpu
        alg_gbs = per_gpu * alg_img / 1e9
        moved_min = sum(r.get('hbm_bytes', r['bytes']) for r in rows)     # per step: every op's sources + output
        macs16_img = sum(r.get('macs_mfma16', 0) for r in rows) / b       # multiply-adds per image on the 16-bit matrix pipe
        macs32_img = sum(r.get('macs_fp32', r['macs'] - r.get('macs_mfma16', 0)) for r in rows) / b   # ... on the float32 pipe (fp32 MFMA / packed FMA)
        # `achieved` / `frac` follow SURVEY.md 8(d)'s agreed accounting: conv-granular ALGORITHMIC bytes (a fused kernel is
        # credited the bytes of the convolutions it replaces) - a measure of work done per second, NOT of bandwidth used.
        # What actually crosses HBM is `moved_*`: the plan's minimum (each launched op's inputs + output) and, when the
        # committed PMC passes cover this configuration, the measured FETCH/WRITE traffic.
        # Round 5: for the float32 plans that accounting is SATURATED (0.98 in round 4 - one more speed-up and
        # the credited GB/s exceed the 8 TB/s peak), so the contract figure of a float32 plan is now SURVEY 8(d)'s other roofline - the
        # ALGORITHMIC float32 FLOPs of the step against the dense float32 peak (2.139 GFLOP per image -> 73.5 k img/s for config 2) - with
        # the bytes really moved beside it; the credited figure stays as `credited_hbm_frac` (it may exceed 1: it is not a bandwidth).
        # The 16-bit plans keep the conv-granular HBM figure (0.5-0.65: nowhere near saturated).
        alg_tf = per_gpu * flops_img / 1e12
        if a.dtype == 'f32':
            head = {'bound': 'fp32', 'achieved': round(alg_tf, 2), 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(alg_tf / FP32_PEAK_TFLOPS, 4),
                    'accounting': 'algorithmic float32 FLOPs per image (SURVEY 8d: %.3f GFLOP) x img/s against the dense float32 MFMA peak; the 1x1 '
                                  'convolutions execute as three float16-plane products on the 16-bit matrix pipe (tflops_mfma16)' % (flops_img / 1e9),
                    'fp32_roofline_img_s': round(FP32_PEAK_TFLOPS * 1e12 / flops_img, 0),
                    'credited_hbm_frac': round(alg_gbs / HBM_PEAK_GBS, 4),
                    'credited_hbm_note': 'conv-granular algorithmic bytes (SURVEY 8d) / 8 TB/s: credited work, not bandwidth - a fused kernel is credited bytes it never moves, so this may exceed 1'}
        else:
            head = {'bound': 'hbm', 'achieved': round(alg_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(alg_gbs / HBM_PEAK_GBS, 4),
                    'accounting': 'conv-granular algorithmic bytes (SURVEY 8d): credited work, not moved bytes'}
        roofline_step = dict(head)
        roofline_step.update({
                         'alg_bytes_per_image': int(alg_img), 'alg_gbs': round(alg_gbs, 1),
                         'moved_bytes_per_image_plan': int(moved_min / b),
                         'moved_gbs_plan': round(moved_min / step_s / 1e9, 1),
                         'frac_hbm_moved_plan': round(moved_min / step_s / 1e9 / HBM_PEAK_GBS, 4),
                         'moved_bytes_per_image_pmc': int(pmc_step_bytes / b) if pmc_step_bytes else None,
                         'moved_gbs_pmc': round(pmc_step_bytes / step_s / 1e9, 1) if pmc_step_bytes else None,
                         'frac_hbm_moved_pmc': round(pmc_step_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4) if pmc_step_bytes else None,
                         'tflops': round(per_gpu * flops_img / 1e12, 2),
                         # by pipe: the multiply-adds of a 16-bit plan's 1x1 convolutions run on the 16-bit matrix pipe
                         'tflops_mfma16': round(per_gpu * 2.0 * macs16_img / 1e12, 2),
                         'frac_mfma16_peak': round(per_gpu * 2.0 * macs16_img / 1e12 / MFMA16_PEAK_TFLOPS, 4),
                         'tflops_fp32_executed': round(per_gpu * 2.0 * macs32_img / 1e12, 2),
                         'frac_fp32_peak_executed': round(per_gpu * 2.0 * macs32_img / 1e12 / FP32_PEAK_TFLOPS, 4),
                         'hbm_roofline_img_s': round(HBM_PEAK_GBS * 1e9 / alg_img, 0),
                         'sum_kernel_ms': round(sum(r['ms'] for r in rows), 3),
                         'launches_per_step': len(rows),
                         # ops that did NOT get the fused form a whitelist exists for (a shape that is on no list silently runs a slower form)
                         'fallback_ops': plan.fallback_ops()})
        if a.per_op:
            # `moved`: the op's own sources + output (what must cross HBM); `credited`: the conv-granular accounting
            # of SURVEY 8(d), where a fused / hoisted op carries the bytes and MACs of the convolutions it stands for
            # (a `_lowres` half is listed with the conv it was split from: its credited columns are 0 by design)
            sys.stderr.write('%-26s %-30s %9s %12s %10s %12s %10s %8s\n' % ('op', 'kernel', 'ms', 'moved MB', 'moved GB/s',
                                                                              'credited MB', 'cred GB/s', 'TF'))
            for r in sorted(rows, key=lambda r: -r['ms']):
                mv = r.get('hbm_bytes', r['bytes'])
                g1 = mv / (r['ms'] * 1e-3) / 1e9 if r['ms'] > 0 else 0
                g2 = r['bytes'] / (r['ms'] * 1e-3) / 1e9 if r['ms'] > 0 else 0
                tf = 2.0 * r['macs'] / (r['ms'] * 1e-3) / 1e12 if r['ms'] > 0 else 0
                sys.stderr.write('%-26s %-30s %9.4f %12.2f %10.1f %12.2f %10.1f %8.2f\n'
                                 % (r['name'], r['kernel'], r['ms'], mv / 1e6, g1, r['bytes'] / 1e6, g2, tf))
            for k, v in sorted(by.items(), key=lambda kv: -kv[1]['ms']):
                sys.stderr.write('SYMBOL %-30s n=%3d total %8.4f ms  moved %8.1f GB/s  credited %8.1f GB/s  %7.2f TF\n'
                                 % (k, v['launches'], v['ms'], v['hbm'] / (v['ms'] * 1e-3) / 1e9 if v['ms'] else 0,
                                    v['bytes'] / (v['ms'] * 1e-3) / 1e9 if v['ms'] else 0,
                                    2.0 * v['macs'] / (v['ms'] * 1e-3) / 1e12 if v['ms'] else 0))
        # ---- the same K steps strictly one after the other (no steps in flight): what one step costs on an otherwise idle GPU
        serial_steps = None
        if world == 1 and a.depth > 1:
            for _ in range(a.warmup + 3):
                pipe1(x, image_hw)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(a.steps):
                pipe1(x, image_hw)
            torch.cuda.synchronize(dev)
            dts = (time.perf_counter() - t1) / a.steps
            serial_steps = {'img_s': round(b / dts, 1), 'ms_per_step': round(dts * 1e3, 4)}
        # ---- p50 per-image latency at B=1 (the second half of BASELINE.json's metric)
        p50 = None
        if world == 1 and not a.no_latency:
            x1 = x[:1].contiguous()
            hw1 = image_hw[:1].contiguous()
            for _ in range(20):
                pipe1(x1, hw1)
            torch.cuda.synchronize(dev)
            ts = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(200):
                e0.record()
                pipe1(x1, hw1)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            p50_eager = round(float(np.median(ts)), 4)
            p50 = p50_eager
            # the same step replayed as ONE HIP graph launch (batch 1 is launch-bound: ~80 launches of a few us)
            try:
                pipe1.enable_graph(True)
                for _ in range(20):
                    pipe1(x1, hw1)
                torch.cuda.synchronize(dev)
                ts = []
                for _ in range(200):
                    e0.record()
                    pipe1(x1, hw1)
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1))
                p50_graph = round(float(np.median(ts)), 4)
                p50 = min(p50_eager, p50_graph)
            except Exception as e:  # graph capture unavailable: the eager figure stands
                p50_graph = None
                sys.stderr.write('hip graph replay failed: %s\n' % (e,))
            finally:
                pipe1.enable_graph(False)
        # ---- SURVEY.md 8(d) "incl. H2D": the same step fed from host memory the sane way - pinned uint8 batch over
        # PCIe (a quarter of the float32 bytes), /255 + letterbox on the GPU (yr_letterbox_batch), then the step.
        # Reported next to `value`, never as `value` (the boundary of the headline is a resident batch).
        incl_h2d = None
        if world == 1 and not a.no_latency:
            from yoloret_amd import runtime as rt
            u8 = (x * 255.0).round().clamp(0, 255).to(torch.uint8).cpu().pin_memory()
            dbuf = torch.empty(u8.shape, dtype=torch.uint8, device=dev)

            def step_serial():   # one step after the other: x is rewritten for every batch here
                det, cnt = pipe1(x, image_hw)
                return gather.start(det, cnt, pipe1.record).wait()

            def step_h2d():
                dbuf.copy_(u8, non_blocking=True)
                rt.letterbox(dbuf, (a.size, a.size), out=x)
                return step_serial()
            for _ in range(3):
                step_h2d()
            sync()
            t1 = time.perf_counter()
            nh = 20
            for _ in range(nh):
                step_h2d()
            sync()
            dth = (time.perf_counter() - t1) / nh
            incl_h2d = {'img_s': round(b / dth, 1), 'ms_per_step': round(dth * 1e3, 4),
                        'host_bytes_per_image': int(u8[0].numel()),
                        'path': 'pinned uint8 [B,H,W,3] -> hipMemcpyAsync H2D -> yr_letterbox_batch (u8/255, letterbox) -> step; '
                                'copy, conversion and step serialised on one stream'}
            # the same with the copy of batch i+1 on its own stream behind step i (yoloret_amd.pipeline.HostFeeder)
            from yoloret_amd.pipeline import HostFeeder
            feeder = HostFeeder(tuple(u8.shape), (a.size, a.size), dev, slots=2)

            def step_fed():
                feeder.submit(u8)
                feeder.take(out=x)
                return step_serial()
            feeder.submit(u8)
            for _ in range(3):
                step_fed()
            sync()
            t1 = time.perf_counter()
            for _ in range(nh):
                step_fed()
            sync()
            dtf = (time.perf_counter() - t1) / nh
            feeder.take(out=x)
            sync()
            incl_h2d['overlapped'] = {'img_s': round(b / dtf, 1), 'ms_per_step': round(dtf * 1e3, 4),
                                      'path': 'HostFeeder: the H2D copy of batch i+1 on a second stream while batch i computes (two device buffers)'}
            if a.depth > 1:
                # ... and with the shipped number of steps in flight: one input buffer per context, rewritten (copy stream ->
                # conversion on the launch stream) only after the step that last read it has finished
                # (2 * depth input buffers: with one per context the conversion of batch i waits for step i - depth, which is still
                # in flight, and copy + conversion sit on the steps' critical path)
                nbuf = 2 * a.depth
                xs = [x] + [torch.empty_like(x) for _ in range(nbuf - 1)]
                done_ev = [None] * nbuf
                turn = [0]

                def step_fed_deep():
                    k = turn[0] % nbuf
                    turn[0] += 1
                    feeder.submit(u8)
                    if done_ev[k] is not None:
                        torch.cuda.current_stream(dev).wait_event(done_ev[k])
                    feeder.take(out=xs[k])
                    pipe(xs[k], image_hw)
                    done_ev[k] = pipe.done
                feeder.submit(u8)              # primed: every take() finds the batch submitted one call earlier
                for _ in range(2 * a.depth):
                    step_fed_deep()
                sync()
                t1 = time.perf_counter()
                for _ in range(nh):
                    step_fed_deep()
                sync()
                dtd = (time.perf_counter() - t1) / nh
                feeder.take(out=x)
                sync()
                incl_h2d['in_flight'] = {'img_s': round(b / dtd, 1), 'ms_per_step': round(dtd * 1e3, 4), 'steps_in_flight': a.depth,
                                         'path': 'HostFeeder + DetectionPipeline(depth): copy of batch i+1 on the copy stream, conversion on the launch stream, '
                                                 'steps on their contexts\' streams; 2 x depth float32 input buffers'}
        # ---- the same with a uint8 network entry: the model is built on Input(dtype='uint8'), its first kernel reads the image
        # bytes (x / 255 inside): no conversion launch, no float32 batch (133 MB written + read per 64 images)
        if incl_h2d is not None:
            try:
                model8 = yolov3_body(L.Input(shape=[a.size, a.size, 3], dtype='uint8'), a.model, 3, num_classes=a.classes)
                model8.set_weights(model.get_weights())
                pipe8 = DetectionPipeline(model8, anchors, a.classes, 3, max_boxes=20, score_threshold=0.2, iou_threshold=0.5, depth=a.depth)
                # 2 * depth + 2 device batches: the copy into a slot waits for the step that last read it - with depth + 1 slots that
                # step is still in flight and every copy (0.6 ms per 64 images) sits on the steps' critical path (24.1k img/s), with
                # 8 slots the copies run ahead (27.9k = 0.99 of the resident figure; tools/u8_probe.py)
                feeder8 = HostFeeder(tuple(u8.shape), (a.size, a.size), dev, slots=2 * a.depth + 2)

                def step_u8():
                    feeder8.submit(u8)
                    xb, slot = feeder8.take_raw()
                    pipe8(xb, image_hw)
                    feeder8.mark_released(slot, pipe8.done if a.depth > 1 else None)
                feeder8.submit(u8)              # primed: every take finds the batch submitted one call earlier
                for _ in range(2 * a.depth + 2):
                    step_u8()
                sync()
                t1 = time.perf_counter()
                for _ in range(nh):
                    step_u8()
                sync()
                dt8 = (time.perf_counter() - t1) / nh
                incl_h2d['uint8_entry'] = {'img_s': round(b / dt8, 1), 'ms_per_step': round(dt8 * 1e3, 4), 'steps_in_flight': a.depth,
                                           'frac_of_resident': round(b / dt8 / per_gpu, 4),
                                           'path': 'HostFeeder.take_raw + a model built on Input(dtype=uint8): the copy of batch i+1 on the copy stream, '
                                                   'the network-entry kernel reads the bytes (x/255 inside), no conversion launch'}
            except Exception as e:   # a model whose entry op has no uint8 form
                incl_h2d['uint8_entry'] = {'error': str(e)[:200]}
        out = {'metric': 'images/sec (+ p50 per-image ms) MobileNetV2-0.75x @416, 1/2/4/8 MI355X',
               'value': round(value, 1), 'unit': 'img/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
               'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
               'config': {'workload': '%s @%d, batch %d per GPU, %s, random weights (SURVEY 8(d) recipe; parity tests run the '
                                      'variance-preserving recipe, see DESIGN.md 5), C=%d: forward + decode + per-class NMS + pack%s'
                                      % (a.model, a.size, b,
                                         {'f32': 'fp32', 'bf16': 'bf16 activations + 1x1 weights on bf16 MFMA (fp32 accumulate, logits, decode, NMS)',
                                          'f16': 'fp16 activations + 1x1 weights on f16 MFMA (fp32 accumulate, logits, decode, NMS)'}[a.dtype],
                                         a.classes, ' + all-gather of detections' if world > 1 else '')
                                      + ('; %d steps in flight (each step = one whole batch on its own HIP stream and workspace)' % a.depth if a.depth > 1 else '')
                                      + ('; the -lite form = no squeeze-excite, ReLU6, and it KEEPS the width-scaled stem' if a.model.endswith('-lite') else ''),
                          'global_batch': b * world, 'parallelism': 'dp%d (image-sharded)' % world},
               'steps_in_flight': a.depth, 'serial_steps': serial_steps, 'p50_ms_b1': p50, 'roofline': roofline, 'roofline_family': roofline_family, 'roofline_step': roofline_step, 'incl_h2d': incl_h2d}
        if p50 is not None:
            out['p50_ms_b1_detail'] = {'eager_launches': p50_eager, 'hip_graph_replay': p50_graph}
        if use_dist:
            out['rccl_ranks'] = world
            out['per_rank_img_s'] = [round(b * a.steps / v, 1) for v in per_rank]
            out['collective'] = 'one all_gather_into_tensor of the packed records per step, on a second stream (overlapped with the next forward)'
            cms = [h.elapsed_ms() for h in timed_handles[:a.steps]]
            cms = [v for v in cms if v is not None]
            if cms:   # event pair around the collective on its own stream: what the exchange costs, not what the step waits for it
                out['collective_ms_per_step'] = {'mean': round(float(np.mean(cms)), 4), 'p50': round(float(np.median(cms)), 4), 'max': round(float(np.max(cms)), 4),
                                                 'bytes_per_rank': int(pipe.record.numel() * 4)}
        from yoloret_amd.pipeline import hw_queue_check
        out['hw_queues'] = hw_queue_check(dev)
        if (world == 1 and not a.no_other_configs and not a.force_dist
                and (a.model, a.size, a.batch, a.dtype) == ('mobilenetv2x75', 416, 64, 'f32')):
            out['other_configs'] = other_configs(dev, anchors, a.classes, max(a.steps, 20), a.depth)
        if world == 1 and a.dtype == 'f32' and not a.no_fp32_forms and not a.force_dist:
            out['fp32_mfma_forms'] = fp32_mfma_forms(a)
            # for a strict reader of `dtype: f32`: the headline's 1x1 convolutions run as three float16-plane products (22 bits per factor,
            # float32-grade: DESIGN.md 4-5); with every GEMM on v_mfma_f32_16x16x4_f32 instead the same step gives
            out['value_strict_fp32'] = out['fp32_mfma_forms'].get('img_s')
            out['serial_strict_fp32'] = out['fp32_mfma_forms'].get('serial_img_s')
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(a.model, a.size, a.classes, anchors, a.cpu_seconds,
                                               gpu_logits=(lambda xs: [y.cpu().numpy() for y in model(torch.from_numpy(np.ascontiguousarray(xs)).to(dev))]) if a.dtype == 'f32' else None)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
