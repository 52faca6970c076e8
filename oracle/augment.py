"""numpy restatement of the reference's per-image augmentation arithmetic - TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference calls OpenCV (cv2.warpAffine / cv2.resize, utils.py:16-36) from data/batch_provider.py:196-266.  OpenCV is not
in this image (and is an un-vendored dependency of the reference), so its published resampling rules are restated here and
PARITY WITH cv2 IS UNPINNED (OpenCV's fixed-point coordinate tables are not modelled):
  warpAffine(src, getRotationMatrix2D((W/2, H/2), angle, 1), INTER_LINEAR, BORDER_CONSTANT 0): dst(x, y) = bilinear src at M^-1 (x, y)
  resize(src, (W, H), INTER_LINEAR): source coordinate (o + 0.5) * scale - 0.5, taps clamped to the image
  *_as_onehot: one-hot channels through the same resampling, then argmax (first maximum)."""
import numpy as np


def _bilinear_zero(img, sx, sy):
    H, W = img.shape
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = (sx - x0).astype(np.float32), (sy - y0).astype(np.float32)

    def at(x, y):
        ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
        return np.where(ok, img[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)], 0).astype(np.float32)
    return ((1 - fy) * ((1 - fx) * at(x0, y0) + fx * at(x0 + 1, y0)) + fy * ((1 - fx) * at(x0, y0 + 1) + fx * at(x0 + 1, y0 + 1))).astype(np.float32)


def rotate_image(img, c, s):
    H, W = img.shape
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    cx, cy = np.float32(W * 0.5), np.float32(H * 0.5)
    dx, dy = xx - cx, yy - cy
    return _bilinear_zero(img.astype(np.float32), np.float32(c) * dx - np.float32(s) * dy + cx, np.float32(s) * dx + np.float32(c) * dy + cy)


def resize_image(img, size):
    h, w = img.shape
    H, W = size
    fy = ((np.arange(H, dtype=np.float32) + np.float32(0.5)) * np.float32(h / H) - np.float32(0.5))
    fx = ((np.arange(W, dtype=np.float32) + np.float32(0.5)) * np.float32(w / W) - np.float32(0.5))
    y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    wy, wx = (fy - y0).astype(np.float32), (fx - x0).astype(np.float32)
    wy[y0 < 0], wx[x0 < 0] = 0, 0
    y0, x0 = np.clip(y0, 0, h - 1), np.clip(x0, 0, w - 1)
    y1, x1 = np.clip(y0 + 1, 0, h - 1), np.clip(x0 + 1, 0, w - 1)
    a = img.astype(np.float32)
    top = (1 - wx)[None, :] * a[y0][:, x0] + wx[None, :] * a[y0][:, x1]
    bot = (1 - wx)[None, :] * a[y1][:, x0] + wx[None, :] * a[y1][:, x1]
    return ((1 - wy)[:, None] * top + wy[:, None] * bot).astype(np.float32)


def _onehot(lbl, nlabels):
    return [(lbl == k).astype(np.float32) for k in range(nlabels)]


def augment(img, lbl, prm, nlabels):
    """One image / label pair through batch_provider.py:186-266 with the parameter row `prm` of draw_augmentation."""
    do_rot, c, s, do_scale, p_x, p_y, r, flips = [float(v) for v in prm]
    img, lbl = img.astype(np.float32), lbl.astype(np.int64)
    if do_rot:
        img = rotate_image(img, c, s)
        lbl = np.argmax(np.stack([rotate_image(ch, c, s) for ch in _onehot(lbl, nlabels)], -1), axis=-1)
    if do_scale:
        p_x, p_y, r = int(p_x), int(p_y), int(r)
        n_x, n_y = img.shape
        img = resize_image(img[p_y:p_y + r, p_x:p_x + r], (n_x, n_y))
        lbl = np.argmax(np.stack([resize_image(ch[p_y:p_y + r, p_x:p_x + r], (n_x, n_y)) for ch in _onehot(lbl, nlabels)], -1), axis=-1)
    if int(flips) & 1:
        img, lbl = np.fliplr(img), np.fliplr(lbl)
    if int(flips) & 2:
        img, lbl = np.flipud(img), np.flipud(lbl)
    return img, lbl
