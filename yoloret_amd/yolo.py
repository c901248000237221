"""Detection wrappers - host-side mirror of reference code/yolo.py for the inference path:
``YoloModel`` (:51-165) and ``YOLO`` (:168-315), same constructor arguments, attributes and return
structure, torch/NumPy instead of TensorFlow, all device arithmetic in libyoloret_hip.so.

Differences that follow from the platform, not from taste:
  * weights: a Keras weights-only ``.h5`` checkpoint of the reference is read directly (``Model.load_weights`` -> the package's
    own HDF5 subset reader ``h5lite.py`` + the Keras layer-name mapping ``keras_h5.py``; reference code/yolo.py:87), an ``.npz``
    (``Model.save_weights``) or a dict work too; the reference's own checkpoints are not shipped (.MISSING_LARGE_BLOBS), so
    ``model_path='synthetic[:seed]'`` draws the seeded random weights used by the bench;
  * image bytes are decoded on the host with PIL (JPEG/PNG decode is not on the GPU path);
    /255, bilinear letterbox resize and zero padding run in the HIP letterbox kernel;
  * ``YoloModel`` also accepts a batch: a list of encoded images -> a list of per-image results
    (the reference is batch-1 only, yolo.py:84); a single image returns exactly the reference's triple;
  * ``detect_image(..., draw=True)`` (the reference's default, yolo.py:276-313) returns the annotated PIL image; the
    font file the reference loads is not shipped with it, so PIL's built-in font stands in when it is absent.
    The export_* / video paths are not mirrored.
"""
import colorsys
import io
import os
from functools import partial
from timeit import default_timer as timer

import numpy as np
import torch

from . import layers as L
from . import runtime as rt
from .pipeline import DetectionPipeline
from .weights import synthetic_weights
from .yolo3.enums import BACKBONE
from .yolo3.model import YoloEval, unpack_detections, yolo_eval_packed, yolov3_body
from .yolo3.utils import get_anchors, get_classes


def _decode_to_u8(image_bytes):
    """tf.io.decode_image(channels=3) on the host: encoded bytes -> uint8 [h,w,3]."""
    from PIL import Image
    img = Image.open(io.BytesIO(image_bytes)).convert('RGB')
    return np.array(img, dtype=np.uint8)  # a writable copy


class YoloModel:
    def __init__(self, model_body, num_anchors, num_scales, classes, model_path, anchors, input_shape, score=0.2,
                 nms=0.5, with_classes=False, name=None, device=None, **kwargs):
        self.model_body = model_body
        self.num_anchors = num_anchors
        self.num_scales = num_scales
        self.classes = classes
        self.with_classes = with_classes
        self.num_classes = len(classes)
        self.model_path = model_path
        self.anchors = anchors
        self.score = score
        self.nms = nms
        self.input_shapes = tuple(input_shape)
        self.name = name
        self.device = torch.device(device if device is not None else 'cuda:0')
        self.model = self.model_body(L.Input(shape=[*input_shape, 3], batch_size=1, dtype='float32'),
                                     num_anchors=self.num_anchors // self.num_scales, num_classes=self.num_classes)
        self._load_weights(model_path)
        self.yolo_eval = YoloEval(self.anchors, self.num_scales, self.num_classes, score_threshold=self.score,
                                  iou_threshold=self.nms, name='yolo')
        self._pipe = DetectionPipeline(self.model, self.anchors, self.num_classes, self.num_scales, max_boxes=20,
                                       score_threshold=self.score, iou_threshold=self.nms)

    def _load_weights(self, model_path):
        if isinstance(model_path, dict):
            self.model.set_weights(model_path)
        elif isinstance(model_path, str) and model_path.startswith('synthetic'):
            seed = int(model_path.split(':')[1]) if ':' in model_path else 1234
            self.model.set_weights(synthetic_weights(self.model, seed, 'survey'))
        else:
            self.model.load_weights(model_path)

    def parse_image(self, image, zoom_in=False):
        """yolo.py:105-112: returns (decoded uint8 image [h,w,3] on the host, letterboxed float32 [H,W,3] on the GPU)."""
        decoded = _decode_to_u8(image) if isinstance(image, (bytes, bytearray)) else np.ascontiguousarray(image, np.uint8)
        if zoom_in:
            decoded = central_crop(decoded, rt.ZOOM_RATIO)   # yolo.py:108-109
        letterboxed = rt.letterbox(torch.from_numpy(decoded).to(self.device), self.input_shapes)
        return decoded, letterboxed

    def call(self, input, zoom_in=False, layer_num=0):
        """yolo.py:117-165.  ``input``: [image_bytes] (or a list of several) -> (boxes int32 [K,4] as
        (ymin,xmin,ymax,xmax) in original-image pixels, scores float32 [K], classes int32 [K]) on the GPU;
        a list of such triples when more than one image is given."""
        if isinstance(input, (bytes, bytearray)):
            input = [input]
        b = len(input)
        x = torch.empty((b, *self.input_shapes, 3), dtype=torch.float32, device=self.device)
        shapes = []
        for i, img in enumerate(input):
            decoded = _decode_to_u8(img) if isinstance(img, (bytes, bytearray)) else np.ascontiguousarray(img, np.uint8)
            rt.letterbox(torch.from_numpy(decoded).to(self.device), self.input_shapes, out=x[i])
            shapes.append(decoded.shape[:2])
        image_hw = rt.image_hw_tensor(np.asarray(shapes, np.int32), b, self.device)
        if zoom_in:
            # yolo.py:154-159: a second pass over the central crop of every image, merged inside the decode
            xz = torch.empty_like(x)
            for i, img in enumerate(input):
                _, lb = self.parse_image(img, zoom_in=True)
                xz[i].copy_(lb)
            ys = [y.clone() for y in self.model(x)]
            det, cnt = yolo_eval_packed(ys, self.anchors, self.num_scales, self.num_classes, image_hw, 20, self.score,
                                        self.nms, zoom_outputs=self.model(xz))
        else:
            det, cnt = self._pipe(x, image_hw)
        res = unpack_detections(det, cnt)
        if self.with_classes:
            res = [(bx, sc, [self.classes[int(c)] for c in cl.tolist()]) for bx, sc, cl in res]
        return res[0] if b == 1 else res

    __call__ = call


def central_crop(image, central_fraction):
    """tf.image.central_crop [3P] on a host [h,w,c] array: per axis, start = int((n - n*fraction) / 2) and
    size = n - 2*start (yolo.py:108-109 passes the AREA ratio 224^2/416^2 as the per-axis fraction)."""
    if not 0.0 < central_fraction <= 1.0:
        raise ValueError('central_fraction must be within (0, 1]')
    h, w = image.shape[:2]
    y0 = int((h - h * central_fraction) / 2)
    x0 = int((w - w * central_fraction) / 2)
    return np.ascontiguousarray(image[y0:h - y0, x0:w - x0])


class YOLO(object):
    def __init__(self, FLAGS):
        """yolo.py:169-181: dict-configured facade (keys: backbone, classes_path, anchors_path, input_size,
        score, nms, with_classes, num_scales, model)."""
        self.backbone = FLAGS.get('backbone', BACKBONE.MOBILENETV2x75)
        self.class_names = get_classes(FLAGS.get('classes_path', 'model_data/voc_classes.txt'))
        self.anchors = get_anchors(FLAGS.get('anchors_path', 'model_data/yolo_anchors'))
        self.input_shape = FLAGS.get('input_size', (416, 416))
        self.score = FLAGS.get('score', 0.2)
        self.nms = FLAGS.get('nms', 0.5)
        self.with_classes = FLAGS.get('with_classes', False)
        self.num_scales = FLAGS.get('num_scales', 3)
        self.generate(FLAGS)

    def generate(self, FLAGS):
        """yolo.py:183-233 (the weights-only branch :195-213; there is no SavedModel to load here)."""
        model_path = FLAGS['model']
        if isinstance(model_path, str) and not model_path.startswith('synthetic'):
            model_path = os.path.expanduser(model_path)
        num_anchors = len(self.anchors)
        num_classes = len(self.class_names)
        names = {BACKBONE.MOBILENETV2x75: 'mobilenetv2x75', BACKBONE.MOBILENETV2x14: 'mobilenetv2x14',
                 BACKBONE.EFFICIENTNETB3: 'efficientnetb3'}
        if self.backbone not in names:
            raise ValueError('unknown backbone %r' % (self.backbone,))
        model_body = partial(yolov3_body, model_name=names[self.backbone], num_anchors=num_anchors // 3,
                             num_classes=num_classes, drop_rate=0.2, data_format='channels_last')
        self.yolo_model = YoloModel(model_body, num_anchors, self.num_scales, self.class_names, model_path, self.anchors,
                                    self.input_shape, self.score, self.nms, self.with_classes,
                                    device=FLAGS.get('device'))
        hsv_tuples = [(x / len(self.class_names), 1., 1.) for x in range(len(self.class_names))]
        colors = [colorsys.hsv_to_rgb(*x) for x in hsv_tuples]
        colors = [(int(c[0] * 255), int(c[1] * 255), int(c[2] * 255)) for c in colors]
        rs = np.random.RandomState(10101)  # fixed seed for consistent colours across runs (yolo.py:230)
        rs.shuffle(colors)
        self.colors = colors

    def detect_image(self, image, draw=True):
        """yolo.py:235-315.  ``image``: bytes or a file-like object; returns numpy (boxes, scores, classes)."""
        image_data = image if isinstance(image, (bytes, bytearray)) else image.read()
        start = timer()
        out_boxes, out_scores, out_classes = self.yolo_model([image_data])
        out_boxes, out_scores = out_boxes.cpu().numpy(), out_scores.cpu().numpy()
        if not self.with_classes:
            out_classes = out_classes.cpu().numpy()
        self.last_seconds = timer() - start
        if draw:
            if isinstance(image, (bytes, bytearray)):
                import io
                image = io.BytesIO(image_data)
            elif hasattr(image, 'seek'):
                image.seek(0)
            return self._draw(image, out_boxes, out_scores, out_classes)
        return out_boxes, out_scores, out_classes

    def _draw(self, image, boxes, scores, classes):
        """The annotated PIL image of yolo.py:276-313: one frame per detection in the class colour, 'name score' label
        above the box (inside it when there is no room).  The reference loads font/FiraMono-Medium.otf, which it does
        not ship; that file is used when present, PIL's built-in font otherwise."""
        from PIL import Image, ImageDraw, ImageFont
        img = Image.open(image).convert('RGB')
        size = int(np.floor(3e-2 * img.size[1] + 0.5))
        try:
            font = ImageFont.truetype('font/FiraMono-Medium.otf', size=size)
        except OSError:
            font = ImageFont.load_default()
        pen = ImageDraw.Draw(img)
        width = max(1, (img.size[0] + img.size[1]) // 300)
        for i in reversed(range(len(classes))):
            c = classes[i]
            if self.with_classes:
                c = self.class_names.index(c.decode('utf-8') if isinstance(c, bytes) else str(c))
            c = int(c)
            text = '%s %.2f' % (self.class_names[c], float(scores[i]))
            top, left, bottom, right = [int(v) for v in boxes[i]]
            x0, y0, x1, y1 = pen.textbbox((0, 0), text, font=font)
            tw, th = x1 - x0, y1 - y0
            ty = top - th if top - th >= 0 else top + 1
            for t in range(width):
                if right - t >= left + t and bottom - t >= top + t:
                    pen.rectangle([left + t, top + t, right - t, bottom - t], outline=self.colors[c])
            pen.rectangle([left, ty, left + tw, ty + th], fill=self.colors[c])
            pen.text((left, ty), text, fill=(0, 0, 0), font=font)
        return img
