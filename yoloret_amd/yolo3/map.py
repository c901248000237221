"""PASCAL-VOC mean-average-precision harness around the detection path.

Mirrors the framework-free part of the reference's `MAPCallback` (code/yolo3/map.py:10-32 `_voc_ap`, :55-74
`parse_text`, :76-221 `calculate_aps`, :223-253 ctor / `on_train_end`): same label-file format, same matching
rule (detections of a class sorted by score; a detection is a true positive if its best-overlapping ground-truth
box of that class in the same image has IoU > threshold - computed with the VOC "+1 pixel" convention - and has
not been claimed yet), same AP definition (area under the monotone precision envelope), classes without any
detection score AP 0, mAP = mean over classes.  It is host-side NumPy: the device work is the model call.
One deliberate difference: detections with EQUAL scores are ranked in input order (a stable sort) - the reference's
`np.argsort(-confidence)` leaves their order to NumPy's unstable default, so its AP can vary between runs on ties.

The TFRecord branch of the reference (map.py:33-53) needs TensorFlow's proto parser and is not provided.
"""
import glob
import os
from timeit import default_timer as timer

import numpy as np


def voc_ap(rec, prec):
    """Area under the precision/recall curve after making precision monotonically non-increasing
    (reference map.py:16-32)."""
    mrec = np.concatenate(([0.0], np.asarray(rec, np.float64), [1.0]))
    mpre = np.concatenate(([0.0], np.asarray(prec, np.float64), [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]          # precision envelope, right to left
    step = np.nonzero(mrec[1:] != mrec[:-1])[0]             # where recall changes
    return float(np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1]))


def parse_text(line):
    """One label line: `<image path> xmin ymin xmax ymax label [xmin ymin xmax ymax label ...]`
    -> (path, float32 [n,5] rows (xmin, ymin, xmax, ymax, label))   (reference map.py:55-74)."""
    tok = line.split()
    if not tok:
        raise ValueError('empty label line')
    if (len(tok) - 1) % 5 != 0:
        raise ValueError('label line for %s has %d values, not a multiple of 5' % (tok[0], len(tok) - 1))
    return tok[0], np.asarray(tok[1:], np.float32).reshape(-1, 5)


def _overlaps(bbgt, bb):
    """IoU of one box against rows of ground truth, VOC convention (inclusive pixel coordinates: +1)."""
    iw = np.maximum(np.minimum(bbgt[:, 2], bb[2]) - np.maximum(bbgt[:, 0], bb[0]) + 1.0, 0.0)
    ih = np.maximum(np.minimum(bbgt[:, 3], bb[3]) - np.maximum(bbgt[:, 1], bb[1]) + 1.0, 0.0)
    inter = iw * ih
    union = ((bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0)
             + (bbgt[:, 2] - bbgt[:, 0] + 1.0) * (bbgt[:, 3] - bbgt[:, 1] + 1.0) - inter)
    return inter / union


def evaluate_detections(pred_res, true_res, num_classes, iou=0.5):
    """Per-class AP.
    pred_res: rows [image index, class, score, left, top, right, bottom]
    true_res: {image index: array [n,5] of (xmin, ymin, xmax, ymax, label)}
    Returns {class: AP}; a class without detections gets 0 (reference map.py:157-162)."""
    pred = np.asarray(pred_res, np.float64).reshape(-1, 7)
    aps = {}
    for cls in range(num_classes):
        p = pred[pred[:, 1] == cls]
        if p.shape[0] == 0:
            aps[cls] = 0
            continue
        gt, claimed, npos = {}, {}, 0
        for index, boxes in true_res.items():
            boxes = np.asarray(boxes, np.float64).reshape(-1, 5)
            sel = boxes[boxes[:, 4] == cls, :4]
            gt[index] = sel
            claimed[index] = np.zeros(sel.shape[0], bool)
            npos += sel.shape[0]
        order = np.argsort(-p[:, 2], kind='stable')
        tp = np.zeros(order.size)
        fp = np.zeros(order.size)
        for j, r in enumerate(order):
            index = int(p[r, 0])
            bbgt = gt[index]
            best, jmax = -np.inf, -1
            if bbgt.size:
                ov = _overlaps(bbgt, p[r, 3:7])
                jmax = int(np.argmax(ov))
                best = ov[jmax]
            if best > iou and not claimed[index][jmax]:
                tp[j] = 1.0
                claimed[index][jmax] = True
            else:
                fp[j] = 1.0   # no overlap above the threshold, or that ground-truth box is already taken
        fp, tp = np.cumsum(fp), np.cumsum(tp)
        rec = tp / np.maximum(float(npos), np.finfo(np.float64).eps)
        prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
        aps[cls] = voc_ap(rec, prec)
    return aps


class MAPCallback:
    """Same construction and use as the reference callback (map.py:223-253): `MAPCallback(glob_path, input_shape,
    class_names, iou=.5, batch_size=1)`, `set_model(m)` with `m([encoded image bytes]) -> (boxes (top, left,
    bottom, right), scores, classes)` (yoloret_amd.yolo.YoloModel), then `calculate_aps()` or `on_train_end(logs)`.
    `glob_path` matches text label files (see parse_text); image paths are taken as written, or relative to
    `root` if given."""

    def __init__(self, glob_path, input_shape, class_names, iou=.5, batch_size=1, root=None):
        self.input_shape = input_shape
        self.class_names = class_names
        self.num_classes = len(class_names)
        self.glob_path = glob_path
        self.iou = iou
        self.batch_size = batch_size
        self.root = root
        self.model = None
        self.seconds_per_image = None

    def set_model(self, model):
        self.model = model

    def _records(self):
        files = sorted(glob.glob(self.glob_path))
        if not files:
            raise FileNotFoundError('no label file matches %r' % (self.glob_path,))
        for f in files:
            if f.endswith(('.tfrecord', '.tfrecords')):
                raise NotImplementedError('TFRecord label files need TensorFlow; use the text format')
            with open(f) as fh:
                for line in fh:
                    if line.strip():
                        yield parse_text(line)

    @staticmethod
    def _host(a):
        return a.detach().cpu().numpy() if hasattr(a, 'detach') else np.asarray(a)

    def calculate_aps(self):
        if self.model is None:
            raise RuntimeError('MAPCallback: set_model() first')
        true_res, pred_res = {}, []
        start = timer()
        idx = 0
        for path, bbox in self._records():
            full = path if self.root is None else os.path.join(self.root, path)
            with open(full, 'rb') as fh:
                image = fh.read()
            boxes, scores, classes = (self._host(t) for t in self.model([image]))
            for (top, left, bottom, right), score, cls in zip(boxes, scores, classes):
                pred_res.append([idx, cls, score, left, top, right, bottom])
            true_res[idx] = bbox
            idx += 1
        self.seconds_per_image = (timer() - start) / max(idx, 1)
        return evaluate_detections(pred_res, true_res, self.num_classes, self.iou)

    def on_train_end(self, logs=None):
        logs = {} if logs is None else logs
        aps = self.calculate_aps()
        for cls in range(self.num_classes):
            if cls in aps:
                print(self.class_names[cls] + ' ap: ', aps[cls])
        m = float(np.mean([aps[c] for c in aps]))
        print('mAP: ', m)
        logs['mAP'] = m
        return logs
