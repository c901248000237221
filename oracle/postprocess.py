"""NumPy restatement of the reference's decode + per-class NMS (test infrastructure).

Follows /root/reference/code/yolo3/model.py:
  yolo_head :344-371, yolo_correct_boxes :374-399, yolo_boxes_and_scores :402-428,
  yolo_eval :431-491, and tf.image.non_max_suppression == NonMaxSuppressionV3
  (third-party, restated from its published behaviour - SURVEY.md C.6).
All arithmetic float32, in the reference's operation order.  The reference folds
the batch axis into the box list (model.py:425-427) and is only correct for
B=1 (yolo.py:84); here every function takes ONE image's maps ([G,G,A,C+5]) and
the batched drivers apply them per image.
"""
import numpy as np

ANCHOR_MASK = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]  # model.py:444
F = np.float32


def _sigmoid(x):
    return F(1) / (F(1) + np.exp(-x, dtype=np.float32))


def yolo_head(feats, anchors, input_shape):
    """model.py:344-371 for one image. feats [Gh,Gw,A,C+5] f32; anchors [A,2] (w,h);
    input_shape (H,W).  Returns box_xy, box_wh [Gh,Gw,A,2], conf [..,1], probs [..,C]."""
    feats = np.asarray(feats, np.float32)
    gh, gw = feats.shape[0:2]
    grid_y = np.tile(np.arange(gh).reshape(-1, 1, 1, 1), [1, gw, 1, 1])
    grid_x = np.tile(np.arange(gw).reshape(1, -1, 1, 1), [gh, 1, 1, 1])
    grid = np.concatenate([grid_x, grid_y], -1).astype(np.float32)
    anchors_t = np.asarray(anchors, np.float32).reshape(1, 1, -1, 2)
    box_xy = (_sigmoid(feats[..., :2]) + grid) / np.array([gw, gh], np.float32)
    box_wh = np.exp(feats[..., 2:4], dtype=np.float32) * anchors_t / \
        np.array([input_shape[1], input_shape[0]], np.float32)
    conf = _sigmoid(feats[..., 4:5])
    probs = _sigmoid(feats[..., 5:])
    return box_xy, box_wh, conf, probs


def yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape):
    """model.py:374-399; returns [...,4] = (y_min,x_min,y_max,x_max) in image pixels."""
    box_yx = box_xy[..., ::-1]
    box_hw = box_wh[..., ::-1]
    input_shape = np.asarray(input_shape, np.float32)
    image_shape = np.asarray(image_shape, np.float32)
    max_shape = np.maximum(image_shape[0], image_shape[1])
    ratio = image_shape / max_shape
    boxed_shape = input_shape * ratio
    offset = (input_shape - boxed_shape) / F(2.)
    scale = image_shape / boxed_shape
    box_yx = (box_yx * input_shape - offset) * scale
    box_hw = box_hw * (input_shape * scale)
    box_mins = box_yx - (box_hw / F(2.))
    box_maxes = box_yx + (box_hw / F(2.))
    return np.concatenate([
        np.clip(box_mins[..., 0:1], F(0), image_shape[0]),
        np.clip(box_mins[..., 1:2], F(0), image_shape[1]),
        np.clip(box_maxes[..., 0:1], F(0), image_shape[0]),
        np.clip(box_maxes[..., 1:2], F(0), image_shape[1])], -1).astype(np.float32)


def yolo_boxes_and_scores(feats, anchors, num_classes, input_shape, image_shape, zoom_feats=None):
    """model.py:402-428.  zoom_feats (:408-417, the zoom-in TTA pass; no caller enables it): decoded with the
    same head, mapped back with the hard-coded 224/416 constants and concatenated on the anchor axis."""
    box_xy, box_wh, conf, probs = yolo_head(feats, anchors, input_shape)
    if zoom_feats is not None:
        xy_z, wh_z, conf_z, probs_z = yolo_head(zoom_feats, anchors, input_shape)
        xy_z = xy_z * F(224 / 416) + F((416 - 224) / (2 * 416))
        wh_z = wh_z * F(224 / 416)
        box_xy = np.concatenate([box_xy, xy_z], -2)
        box_wh = np.concatenate([box_wh, wh_z], -2)
        conf = np.concatenate([conf, conf_z], -2)
        probs = np.concatenate([probs, probs_z], -2)
    boxes = yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape).reshape(-1, 4)
    scores = (conf * probs).reshape(-1, num_classes)
    return boxes, scores


def decode_image(yolo_outputs, anchors, num_classes, image_shape, num_scales=3, zoom_outputs=None):
    """The concat of model.py:453-469 for one image: boxes [N,4], scores [N,C];
    scale order 32,16,8; flat index ((h*G+w)*A+a) inside a scale (2A per cell with zoom_outputs)."""
    anchors = np.asarray(anchors, np.float32)
    mask = ANCHOR_MASK[-num_scales:]
    input_shape = (yolo_outputs[0].shape[0] * 32, yolo_outputs[0].shape[1] * 32)  # model.py:449
    bs, ss = [], []
    for l in range(num_scales):
        b, s = yolo_boxes_and_scores(yolo_outputs[l], anchors[mask[l]], num_classes,
                                     input_shape, image_shape,
                                     None if zoom_outputs is None else zoom_outputs[l])
        bs.append(b)
        ss.append(s)
    return np.concatenate(bs, 0), np.concatenate(ss, 0)


def iou_one_to_many(box, boxes):
    """NonMaxSuppression's IOU() [3P], float32, in its operation order."""
    def canon(b):
        return (np.minimum(b[..., 0], b[..., 2]), np.minimum(b[..., 1], b[..., 3]),
                np.maximum(b[..., 0], b[..., 2]), np.maximum(b[..., 1], b[..., 3]))
    ymin_i, xmin_i, ymax_i, xmax_i = canon(np.asarray(box, np.float32))
    ymin_j, xmin_j, ymax_j, xmax_j = canon(np.asarray(boxes, np.float32))
    area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i)
    area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j)
    iy = np.maximum(np.minimum(ymax_i, ymax_j) - np.maximum(ymin_i, ymin_j), F(0))
    ix = np.maximum(np.minimum(xmax_i, xmax_j) - np.maximum(xmin_i, xmin_j), F(0))
    inter = iy * ix
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = inter / (area_i + area_j - inter)
    return np.where((area_i <= 0) | (area_j <= 0), F(0), iou).astype(np.float32)


def non_max_suppression(boxes, scores, max_output_size, iou_threshold, score_threshold):
    """tf.image.non_max_suppression [3P] (hard NMS): candidates score>thr (strict),
    order (score desc, index asc), suppress iff IoU>iou_threshold (strict).
    Returns selected indices (int32) in pick order."""
    boxes = np.asarray(boxes, np.float32)
    key = np.asarray(scores, np.float32).copy()
    alive = key > F(score_threshold)
    picked = []
    while len(picked) < max_output_size and alive.any():
        masked = np.where(alive, key, -np.inf)
        i = int(np.argmax(masked))  # first max -> smallest index among ties
        picked.append(i)
        alive[i] = False
        idx = np.nonzero(alive)[0]
        if idx.size:
            iou = iou_one_to_many(boxes[i], boxes[idx])
            alive[idx[iou > F(iou_threshold)]] = False
    return np.asarray(picked, np.int32)


def nms_decision_margin(boxes, scores, max_output_size, iou_threshold, score_threshold):
    """``non_max_suppression`` plus the smallest MARGIN of any decision that shaped its result: the
    score gap between a popped candidate and the runner-up (ordering), the distance of an evaluated IoU from the IoU
    threshold, the distance of a score from the score threshold.  Only candidates that were reachable count: when the
    loop ends with max_output_size picks, boxes scoring below the last pick can neither be picked nor change a pick,
    whatever side of a threshold they are on.  Two runs whose scores / IoUs differ by less than the margin take the
    same decisions and return the same picks; a disagreement on a problem whose margin exceeds the numerical noise of
    its inputs is a real error.  -> (picks int32, margin float)."""
    boxes = np.asarray(boxes, np.float32)
    key = np.asarray(scores, np.float32).copy()
    thr = float(F(score_threshold))
    alive = key > F(score_threshold)
    picked, order_margin, iou_steps = [], np.inf, []
    while len(picked) < max_output_size and alive.any():
        masked = np.where(alive, key, -np.inf)
        i = int(np.argmax(masked))
        masked[i] = -np.inf
        runner = float(masked.max())
        if np.isfinite(runner):
            order_margin = min(order_margin, float(key[i]) - runner)   # 0 for exact ties (broken by index in every run)
        picked.append(i)
        alive[i] = False
        idx = np.nonzero(alive)[0]
        if idx.size:
            iou = iou_one_to_many(boxes[i], boxes[idx])
            iou_steps.append((key[idx].copy(), np.abs(iou.astype(np.float64) - float(F(iou_threshold)))))
            alive[idx[iou > F(iou_threshold)]] = False
    # candidates that could have been reached: everything when the list ran dry, else those scoring >= the last pick
    floor = float(key[picked[-1]]) if len(picked) >= max_output_size else -np.inf
    margin = order_margin
    reach = key >= floor
    if reach.any():
        margin = min(margin, float(np.min(np.abs(key[reach].astype(np.float64) - thr))))
    for sc, d in iou_steps:
        m = sc >= floor
        if m.any():
            margin = min(margin, float(d[m].min()))
    return np.asarray(picked, np.int32), margin


def nms_bruteforce(boxes, scores, max_output_size, iou_threshold, score_threshold):
    """Line-by-line form of the TF kernel's loop (pop best; test against the
    already-selected, most recent first).  O(N*K) Python; small cases only -
    used to validate ``non_max_suppression``'s reformulation."""
    boxes = np.asarray(boxes, np.float32)
    scores = np.asarray(scores, np.float32)
    cand = [i for i in range(len(scores)) if scores[i] > F(score_threshold)]
    cand.sort(key=lambda i: (-float(scores[i]), i))
    selected = []
    for i in cand:
        if len(selected) >= max_output_size:
            break
        ok = True
        for j in reversed(selected):
            if iou_one_to_many(boxes[i], boxes[j][None])[0] > F(iou_threshold):
                ok = False
                break
        if ok:
            selected.append(i)
    return np.asarray(selected, np.int32)


def yolo_eval(yolo_outputs, anchors, num_scales, num_classes, image_shape, max_boxes=20,
              score_threshold=.6, iou_threshold=.5, return_indices=False, zoom_outputs=None):
    """model.py:431-491 for ONE image (yolo_outputs: list of [G,G,A,C+5])."""
    boxes, box_scores = decode_image(yolo_outputs, anchors, num_classes, image_shape, num_scales, zoom_outputs)
    boxes_, scores_, classes_, idxs = [], [], [], []
    for c in range(num_classes):
        nms_index = non_max_suppression(boxes, box_scores[:, c], max_boxes,
                                        iou_threshold, score_threshold)
        boxes_.append(boxes[nms_index])
        scores_.append(box_scores[nms_index, c])
        classes_.append(np.full(len(nms_index), c, np.int32))
        idxs.append(nms_index)
    out = (np.concatenate(boxes_, 0).astype(np.int32),  # tf.cast truncates toward zero
           np.concatenate(scores_, 0).astype(np.float32),
           np.concatenate(classes_, 0))
    return out + (idxs,) if return_indices else out


def yolo_eval_batch(ys, anchors, num_scales, num_classes, image_shapes, **kw):
    """Per-image application to a batch: ys = [y1,y2,y3] with leading batch axis."""
    b = ys[0].shape[0]
    image_shapes = np.broadcast_to(np.asarray(image_shapes), (b, 2))
    return [yolo_eval([y[i] for y in ys], anchors, num_scales, num_classes,
                      image_shapes[i], **kw) for i in range(b)]
