"""Oracle: max ROIPooling (numpy).  TEST INFRASTRUCTURE ONLY.

The reference calls MXNet's built-in `mx.symbol.ROIPooling(pooled_size=(7,7),
spatial_scale=0.0625)` (symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_
multi_head_16.py:252-253).  MXNet v1.1.0 is an un-vendored dependency, so this is a
restatement of its published kernel (src/operator/roi_pooling.cu, the Caffe
Fast-RCNN ROIPool): PARITY UNPINNED.  All bin arithmetic is float32 as in the
Dtype=float instantiation.
"""
import numpy as np

F32 = np.float32


def _round_half_away(x):
    """C `round()` on float32."""
    x = F32(x)
    return int(np.floor(x + F32(0.5))) if x >= 0 else int(np.ceil(x - F32(0.5)))


def roi_pooling(data, rois, pooled_size=(7, 7), spatial_scale=0.0625, return_argmax=False):
    """data [B, C, H, W] fp32, rois [R, 5] (batch_idx, x1, y1, x2, y2) fp32
    -> out [R, C, PH, PW] fp32 (argmax = flat h*W+w index or -1)."""
    data = np.asarray(data, dtype=F32)
    rois = np.asarray(rois, dtype=F32)
    _, C, H, W = data.shape
    PH, PW = pooled_size
    R = rois.shape[0]
    out = np.zeros((R, C, PH, PW), dtype=F32)
    arg = np.full((R, C, PH, PW), -1, dtype=np.int32)
    s = F32(spatial_scale)
    for r in range(R):
        b = int(rois[r, 0])
        rs_w = _round_half_away(rois[r, 1] * s)
        rs_h = _round_half_away(rois[r, 2] * s)
        re_w = _round_half_away(rois[r, 3] * s)
        re_h = _round_half_away(rois[r, 4] * s)
        rw = max(re_w - rs_w + 1, 1)
        rh = max(re_h - rs_h + 1, 1)
        bin_h = F32(rh) / F32(PH)
        bin_w = F32(rw) / F32(PW)
        for ph in range(PH):
            hs = int(np.floor(F32(ph) * bin_h))
            he = int(np.ceil(F32(ph + 1) * bin_h))
            hs = min(max(hs + rs_h, 0), H)
            he = min(max(he + rs_h, 0), H)
            for pw in range(PW):
                ws = int(np.floor(F32(pw) * bin_w))
                we = int(np.ceil(F32(pw + 1) * bin_w))
                ws = min(max(ws + rs_w, 0), W)
                we = min(max(we + rs_w, 0), W)
                if he <= hs or we <= ws:
                    continue                      # empty bin -> 0, argmax -1
                win = data[b, :, hs:he, ws:we].reshape(C, -1)
                k = np.argmax(win, axis=1)        # first max in row-major scan (strict >)
                out[r, :, ph, pw] = win[np.arange(C), k]
                ww = we - ws
                arg[r, :, ph, pw] = (hs + k // ww) * W + (ws + k % ww)
    if return_argmax:
        return out, arg
    return out
