"""Image pre-processing of `lib/utils/image.py:16-130`: read (BGR), optional mirror, resize so that the short side is
SCALES[i][0] with the long side capped at SCALES[i][1], zero-pad to IMAGE_STRIDE, subtract PIXEL_MEANS and emit RGB-ordered
[1, 3, H, W]; boxes are scaled, rounded and clipped.  cv2 is not installed here: decoding uses Pillow, `.npy` arrays are read
as is, and the resize is a numpy restatement of `cv2.resize(im, None, None, fx=s, fy=s, interpolation=cv2.INTER_LINEAR)`:
plain 2-tap bilinear for up- AND down-scaling (no antialiasing support widening), source coordinate (d + 0.5) / s - 0.5
clamped to the image, and uint8 images come back ROUNDED to uint8 like cv2's output (the mean is subtracted from the rounded
values).  cv2 evaluates the taps in 11-bit fixed point: results can differ by one grey level on a fraction of the pixels --
unpinned (no cv2 here to compare with)."""
import os
import random

import numpy as np


def imread_bgr(path):
    if path.endswith('.npy'):
        im = np.load(path)
        assert im.ndim == 3 and im.shape[2] == 3
        return im
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return rgb[:, :, ::-1]


def _bilinear_axis(n_src, n_dst, scale):
    """cv2 INTER_LINEAR taps of one axis: (left index, right index, weight of the right tap) per destination position."""
    f = (np.arange(n_dst, dtype=np.float64) + 0.5) / scale - 0.5
    i0 = np.floor(f).astype(np.int64)
    w1 = f - i0
    lo = i0 < 0
    i0[lo] = 0; w1[lo] = 0.0
    hi = i0 >= n_src - 1
    i0[hi] = n_src - 1; w1[hi] = 0.0
    return i0, np.minimum(i0 + 1, n_src - 1), w1


def resize(im, target_size, max_size, stride=0):
    """image.py:78-108.  Returns (resized [+ padded] float image, im_scale)."""
    h, w = im.shape[:2]
    size_min, size_max = min(h, w), max(h, w)
    im_scale = float(target_size) / float(size_min)
    if np.round(im_scale * size_max) > max_size:
        im_scale = float(max_size) / float(size_max)
    nw, nh = int(round(w * im_scale)), int(round(h * im_scale))
    x0, x1, wx = _bilinear_axis(w, nw, im_scale)
    y0, y1, wy = _bilinear_axis(h, nh, im_scale)
    src = im.astype(np.float64)
    top = src[y0][:, x0] * (1.0 - wx)[None, :, None] + src[y0][:, x1] * wx[None, :, None]
    bot = src[y1][:, x0] * (1.0 - wx)[None, :, None] + src[y1][:, x1] * wx[None, :, None]
    out = top * (1.0 - wy)[:, None, None] + bot * wy[:, None, None]
    if im.dtype == np.uint8:                  # cv2 returns the image in the input's type: rounded, saturated
        out = np.clip(np.rint(out), 0, 255)
    out = out.astype(np.float32)
    if stride == 0:
        return out, im_scale
    ph = int(np.ceil(out.shape[0] / float(stride)) * stride)
    pw = int(np.ceil(out.shape[1] / float(stride)) * stride)
    padded = np.zeros((ph, pw, out.shape[2]), dtype=out.dtype)
    padded[:out.shape[0], :out.shape[1], :] = out
    return padded, im_scale


def transform(im, pixel_means):
    """image.py:110-122: [H, W, 3] BGR -> [1, 3, H, W] in R, G, B order with the means subtracted."""
    t = np.zeros((1, 3, im.shape[0], im.shape[1]), dtype=np.float32)
    for i in range(3):
        t[0, i] = im[:, :, 2 - i] - pixel_means[2 - i]
    return t


def clip_boxes(boxes, im_shape):
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def get_image(roidb, config, rng=random):
    """image.py:16-45: -> (list of [1,3,H,W] float32 arrays, roidb copies with scaled `boxes` and `im_info`)."""
    ims, out = [], []
    for rec in roidb:
        assert os.path.exists(rec['image']), '%s does not exist' % rec['image']
        im = imread_bgr(rec['image'])
        if rec['flipped']:
            im = im[:, ::-1, :]
        new = dict(rec)
        target_size, max_size = config.SCALES[rng.randrange(len(config.SCALES))]
        im, im_scale = resize(im, target_size, max_size, stride=config.network.IMAGE_STRIDE)
        t = transform(im, config.network.PIXEL_MEANS)
        ims.append(t)
        im_info = [t.shape[2], t.shape[3], im_scale]
        new['boxes'] = clip_boxes(np.round(rec['boxes'].astype(np.float64).copy() * im_scale), im_info[:2])
        new['im_info'] = im_info
        out.append(new)
    return ims, out


def tensor_vstack(tensor_list, pad=0):
    """image.py:tensor_vstack: stack along axis 0, zero-padding every other axis to the largest extent."""
    ndim = len(tensor_list[0].shape)
    dims = [max(t.shape[d] for t in tensor_list) for d in range(1, ndim)]
    n = sum(t.shape[0] for t in tensor_list)
    out = np.full([n] + dims, pad, dtype=tensor_list[0].dtype)
    k = 0
    for t in tensor_list:
        sl = (slice(k, k + t.shape[0]),) + tuple(slice(0, s) for s in t.shape[1:])
        out[sl] = t
        k += t.shape[0]
    return out
