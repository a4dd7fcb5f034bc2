"""`lib/nms/nms.py:21-141`, `lib/nms/gpu_nms.pyx:18-33`, `lib/nms/cpu_nms.pyx:17-68` on the HIP library.

| reference                                   | here                                                                    |
|---------------------------------------------|-------------------------------------------------------------------------|
| `gpu_nms(dets f32[n,5], thresh, device_id)` | sort on the host exactly like the .pyx, then the C symbol `_nms` (`lib/nms/gpu_nms.hpp:1-2`, same prototype, host pointers) |
| `cpu_nms(dets, thresh)`                     | same kernels; float32 IoU, suppress when `ovr >= thresh` (cpu_nms.pyx:65) = `ovr > nextafter(thresh, -inf)` |
| `nms(dets, thresh)` (numpy, dets' dtype)    | `relnet_class_nms_ex` greedy branch, float64, keeps `ovr <= thresh` (nms.py:79) |
| `soft_nms(dets, thresh, max_dets)`          | `relnet_class_nms_ex` Gaussian branch, float64 (nms.py:85-141)          |
| `*_wrapper(thresh[, ...])`                  | the same closures (nms.py:21-42)                                        |

Return conventions are the reference's: `gpu_nms` / `cpu_nms` / `nms` return the kept indices into the UNSORTED input
(in descending score order), `soft_nms` returns the re-scored `dets[keep]` rows and, like the reference (nms.py:114),
writes the decayed scores back into the caller's array.  Ties: the reference's `argsort()[::-1]` is an unstable
sort; here equal scores are ordered by descending index.  There is no CPU fallback: without a GPU these raise.
"""
import ctypes

import numpy as np
import torch

from .. import lib as _lib
from .. import ops


def _sorted_f32(dets):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.ndim != 2 or dets.shape[1] < 5:
        raise ValueError("dets must be [n, >=5] rows of x1, y1, x2, y2, score; got %s" % (dets.shape,))
    order = dets[:, 4].argsort(kind='stable')[::-1].astype(np.int32)         # gpu_nms.pyx:27-28
    return np.ascontiguousarray(dets[order, :]), order


def _nms_c(sorted_dets, thresh, device_id):
    n, dim = sorted_dets.shape
    keep = np.zeros(n, dtype=np.int32)
    num_out = ctypes.c_int(0)
    if not torch.cuda.is_available():
        raise _lib.RelnetError("gpu_nms needs a GPU (HIP kernels only; no CPU fallback)")
    _lib.load()._nms(keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(num_out),
                     sorted_dets.ctypes.data_as(ctypes.c_void_p), n, dim, float(thresh), int(device_id))
    return keep[:num_out.value]


def gpu_nms(dets, thresh, device_id=0):
    """gpu_nms.pyx:18-33: indices (into `dets`) of the boxes kept by greedy NMS, IoU > thresh suppressed."""
    if dets.shape[0] == 0:
        return []
    sorted_dets, order = _sorted_f32(dets)
    return list(order[_nms_c(sorted_dets, thresh, device_id)])


def cpu_nms(dets, thresh):
    """cpu_nms.pyx:17-68 semantics (float32, `ovr >= thresh` suppresses) on the same device kernels."""
    if dets.shape[0] == 0:
        return []
    sorted_dets, order = _sorted_f32(dets)
    t = np.nextafter(np.float32(thresh), np.float32(-np.inf))                # ovr >= t  <=>  ovr > pred(t) in float32
    return [int(i) for i in order[_nms_c(sorted_dets, float(t), 0)]]


def _class_nms_single(dets, param, soft, max_picks):
    d = np.ascontiguousarray(dets, dtype=np.float64)
    n = d.shape[0]
    if n > 1024:
        raise ValueError("at most 1024 candidates per call (TOP_ROIS of the largest configuration is 1000), got %d" % n)
    boxes = torch.as_tensor(d[None, :, :4].copy()).cuda()
    scores = torch.as_tensor(d[None, :, 4].copy()).cuda()
    out, counts, index = ops.class_nms(None, boxes, -np.inf, param, soft, max_picks=max_picks, scores64=scores,
                                       want_index=True)
    k = int(counts[0, 0])
    return out[0, 0, :k].cpu().numpy(), index[0, 0, :k].cpu().numpy().astype(np.intp)


def nms(dets, thresh):
    """nms.py:45-82: keep list (indices into dets, picked in descending score order); overlap <= thresh survives."""
    if dets.shape[0] == 0:
        return []
    _, keep = _class_nms_single(dets, thresh, False, 0)
    return list(keep)


def soft_nms(dets, thresh, max_dets):
    """nms.py:96-141: Gaussian soft-NMS, sigma = thresh; returns dets[keep] with the decayed scores."""
    if dets.shape[0] == 0:
        return np.zeros((0, 5))
    rows, keep = _class_nms_single(dets, thresh, True, 0 if max_dets == -1 else max_dets)
    dets[keep, 4] = rows[:, 4].astype(dets.dtype, copy=False)                # nms.py:114 writes into the caller's array
    return dets[keep, :]


def py_nms_wrapper(thresh):
    def _nms(dets):
        return nms(dets, thresh)
    return _nms


def py_softnms_wrapper(thresh, max_dets=-1):
    def _nms(dets):
        return soft_nms(dets, thresh, max_dets)
    return _nms


def cpu_nms_wrapper(thresh):
    def _nms(dets):
        return cpu_nms(dets, thresh)
    return _nms


def gpu_nms_wrapper(thresh, device_id):
    def _nms(dets):
        return gpu_nms(dets, thresh, device_id)
    return _nms
