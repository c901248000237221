"""NumPy restatement of the reference's image parsing (test infrastructure only).

reference code/yolo.py:105-112: tf.io.decode_image(channels=3, dtype=float32) (uint8 * 1/255) then
code/yolo3/utils.py:67-83 letterbox_image: nh/nw = trunc(float64 product), tf.image.resize (bilinear,
half-pixel centres, no antialias [3P: ResizeBilinear kernel]) and pad_to_bounding_box (zeros).
float32 in the kernel's operation order.
"""
import numpy as np


def letterbox_image(img_u8, size):
    """img_u8 [ih,iw,3] uint8 -> float32 [h,w,3]; also returns (nh, nw, dy, dx)."""
    ih, iw = img_u8.shape[:2]
    h, w = int(size[0]), int(size[1])
    r = min(w / iw, h / ih)
    nh, nw = int(float(ih) * r), int(float(iw) * r)
    dy, dx = (h - nh) // 2, (w - nw) // 2
    f = img_u8.astype(np.float32) * np.float32(1.0 / 255.0)
    sy, sx = np.float32(ih) / np.float32(nh), np.float32(iw) / np.float32(nw)
    fy = (np.arange(nh, dtype=np.float32) + np.float32(0.5)) * sy - np.float32(0.5)
    fx = (np.arange(nw, dtype=np.float32) + np.float32(0.5)) * sx - np.float32(0.5)
    y0 = np.maximum(np.floor(fy).astype(np.int64), 0)
    y1 = np.minimum(np.ceil(fy).astype(np.int64), ih - 1)
    x0 = np.maximum(np.floor(fx).astype(np.int64), 0)
    x1 = np.minimum(np.ceil(fx).astype(np.int64), iw - 1)
    ly = (fy - np.floor(fy)).astype(np.float32)[:, None, None]
    lx = (fx - np.floor(fx)).astype(np.float32)[None, :, None]
    tl, tr = f[y0][:, x0], f[y0][:, x1]
    bl, br = f[y1][:, x0], f[y1][:, x1]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    res = (top + (bot - top) * ly).astype(np.float32)
    out = np.zeros((h, w, 3), np.float32)
    out[dy:dy + nh, dx:dx + nw] = res
    return out, (nh, nw, dy, dx)
