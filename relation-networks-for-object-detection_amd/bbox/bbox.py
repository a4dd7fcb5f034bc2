"""`bbox_overlaps_cython(boxes, query_boxes)` of lib/bbox/bbox.pyx:15-55 on `relnet_bbox_overlaps`.

Callers in the reference: core/rcnn.py:303 (sample_rois_v2), operator_py/nms_multi_target.py:51,
lib/rpn/rpn.py:163 (assign_anchor) -- all pass float64 numpy arrays and get a float64 [N, K] matrix back.
numpy in -> numpy out (one H2D / D2H pair, as the compiled extension's callers expect); CUDA tensors in ->
CUDA tensor out, nothing leaves the device.  No CPU fallback."""
import numpy as np
import torch

from .. import lib as _lib
from .. import ops


def bbox_overlaps_cython(boxes, query_boxes):
    if isinstance(boxes, torch.Tensor):
        return ops.bbox_overlaps(boxes.double(), query_boxes.double())
    b = np.ascontiguousarray(boxes, dtype=np.float64)
    q = np.ascontiguousarray(query_boxes, dtype=np.float64)
    if b.ndim != 2 or q.ndim != 2 or b.shape[1] < 4 or q.shape[1] < 4:
        raise ValueError("boxes [N, 4] and query_boxes [K, 4] expected, got %s and %s" % (b.shape, q.shape))
    if b.shape[0] == 0 or q.shape[0] == 0:
        return np.zeros((b.shape[0], q.shape[0]), dtype=np.float64)
    if not torch.cuda.is_available():
        raise _lib.RelnetError("bbox_overlaps_cython needs a GPU (HIP kernels only; no CPU fallback)")
    out = ops.bbox_overlaps(torch.as_tensor(b[:, :4].copy()).cuda(), torch.as_tensor(q[:, :4].copy()).cuda())
    return out.cpu().numpy()


def bbox_overlaps(boxes, query_boxes):
    """bbox_transform.py:18-19."""
    return bbox_overlaps_cython(boxes, query_boxes)
