"""Oracle: ROIAlign forward (numpy).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against the reference: msracver/Relation-Networks-for-Object-Detection has no ROIAlign (its graphs pool with
mx.symbol.ROIPooling, symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py:252-253; MXNet v1.1.0, the pinned
dependency, predates mx.contrib.sym.ROIAlign).  BASELINE.json's north_star names a ROIAlign op, so the PUBLISHED algorithm is restated here:
He, Gkioxari, Dollar, Girshick, "Mask R-CNN" (2017), section 3, in the form every public implementation shares (Detectron RoIAlign,
MXNet >= 1.3 contrib.ROIAlign, torchvision.ops.roi_align):

  roi_start = x1 * scale - off, roi_end = x2 * scale - off                       off = 0.5 if aligned else 0
  roi_w = roi_end_w - roi_start_w   (not aligned: max(., 1));  bin_w = roi_w / PW;  grid_w = sampling_ratio or ceil(roi_w / PW)
  sample (iy, ix) of bin (ph, pw):  y = roi_start_h + ph bin_h + (iy + 0.5) bin_h / grid_h
  bilinear(y, x) = 0 outside (-1, H) x (-1, W); y, x clamped at 0; the last row / column has no upper neighbour
  out = sum of samples / max(grid_h grid_w, 1)

What pins it instead (tests/test_oracle_golden.py, tests/test_gpu_roi_align.py): known answers that follow from the definition alone --
a constant map pools to the constant; an AFFINE map f(y, x) = a y + b x + c pools to f at the bin centre exactly (bilinear interpolation
reproduces affine functions and the samples are symmetric about the centre) for boxes inside the map; a one-hot map returns the bilinear
weights; the float32 evaluation follows the same operation order as csrc/roi_align.hip (bit-identical there) and a float64 evaluation of
the same formulas bounds its rounding.
"""
import numpy as np

F32 = np.float32


def _corners(y, x, H, W, dt):
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return None
    y = dt(max(y, dt(0))); x = dt(max(x, dt(0)))
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1; y = dt(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1; x = dt(xl)
    else:
        xh = xl + 1
    ly, lx = dt(y - dt(yl)), dt(x - dt(xl))
    hy, hx = dt(dt(1) - ly), dt(dt(1) - lx)
    return yl, xl, yh, xh, dt(hy * hx), dt(hy * lx), dt(ly * hx), dt(ly * lx)


def roi_align(data, rois, pooled=(7, 7), spatial_scale=0.0625, sampling_ratio=2, aligned=False, dtype=F32, return_samples=False):
    """data [B,C,H,W], rois [R,5] (batch index, x1, y1, x2, y2) -> [R,C,PH,PW] in `dtype` arithmetic (float32: the kernel's operation
    order; float64: the definition)."""
    dt = dtype
    data = np.asarray(data, dtype=dt)
    rois = np.asarray(rois, dtype=F32)
    B, C, H, W = data.shape
    PH, PW = pooled
    out = np.zeros((rois.shape[0], C, PH, PW), dtype=dt)
    samples = []
    for r, roi in enumerate(rois):
        b = int(roi[0])
        off = dt(0.5) if aligned else dt(0)
        sc = dt(F32(spatial_scale))
        sw, sh = dt(dt(roi[1]) * sc - off), dt(dt(roi[2]) * sc - off)
        ew, eh = dt(dt(roi[3]) * sc - off), dt(dt(roi[4]) * sc - off)
        rw, rh = dt(ew - sw), dt(eh - sh)
        if not aligned:
            rw, rh = max(rw, dt(1)), max(rh, dt(1))
        bh, bw = dt(rh / dt(PH)), dt(rw / dt(PW))
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(dt(rh / dt(PH))))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(dt(rw / dt(PW))))
        count = dt(max(gh * gw, 1))
        for ph in range(PH):
            for pw in range(PW):
                acc = np.zeros(C, dtype=dt)
                for iy in range(gh):
                    y = dt(dt(sh + dt(dt(ph) * bh)) + dt(dt(dt(dt(iy) + dt(0.5)) * bh) / dt(gh)))
                    for ix in range(gw):
                        x = dt(dt(sw + dt(dt(pw) * bw)) + dt(dt(dt(dt(ix) + dt(0.5)) * bw) / dt(gw)))
                        k = _corners(y, x, H, W, dt)
                        if return_samples:
                            samples.append((r, ph, pw, float(y), float(x)))
                        if k is None:
                            continue
                        yl, xl, yh, xh, w1, w2, w3, w4 = k
                        v = (w1 * data[b, :, yl, xl] + w2 * data[b, :, yl, xh]).astype(dt)
                        v = (v + w3 * data[b, :, yh, xl]).astype(dt)
                        v = (v + w4 * data[b, :, yh, xh]).astype(dt)
                        acc = (acc + v).astype(dt)
                out[r, :, ph, pw] = acc / count
    return (out, samples) if return_samples else out
