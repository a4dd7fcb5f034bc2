"""ctypes front end of oracle/_ref/libref_cuda*.so (TEST INFRASTRUCTURE ONLY): the reference's own CUDA kernels
(deformable_im2col.cuh, deformable_psroi_pooling.cu, nms_kernel.cu), compiled unedited for gfx950 by oracle/build_ref.py.
They run on the GPU, so this module is usable on the GPU box only; numpy in, numpy out (float32, NCHW)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def path(fma=False):
    return os.path.join(HERE, '_ref', 'libref_cuda_fma.so' if fma else 'libref_cuda.so')


def available(fma=False):
    return os.path.exists(path(fma))


def load(fma=False):
    if fma not in _libs:
        if not available(fma):
            raise RuntimeError('%s is missing: run `python oracle/build_ref.py` where /root/reference exists' % path(fma))
        _libs[fma] = C.CDLL(path(fma))
    return _libs[fma]


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _out_dim(n, k, pad, stride, dil):
    return (n + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def _geom(kernel, pad, stride, dilate):
    return [C.c_int(int(v)) for v in (kernel[0], kernel[1], pad[0], pad[1], stride[0], stride[1], dilate[0], dilate[1])]


def deformable_im2col(data, offset, kernel, pad, stride, dilate, dg, fma=False):
    """data [C,H,W], offset [dg*2*kh*kw, Ho, Wo] -> col [C*kh*kw, Ho, Wo] (deformable_im2col_gpu_kernel)."""
    data, offset = _f(data), _f(offset)
    Cc, H, W = data.shape
    Ho, Wo = _out_dim(H, kernel[0], pad[0], stride[0], dilate[0]), _out_dim(W, kernel[1], pad[1], stride[1], dilate[1])
    col = np.empty((Cc * kernel[0] * kernel[1], Ho, Wo), np.float32)
    rc = load(fma).ref_deformable_im2col(_p(data), _p(offset), Cc, H, W, *_geom(kernel, pad, stride, dilate), int(dg), _p(col))
    assert rc == 0, rc
    return col


def deformable_col2im(col, offset, im_shape, kernel, pad, stride, dilate, dg, fma=False):
    """col [C*kh*kw, Ho, Wo], offset -> grad_im [C,H,W] (deformable_col2im_gpu_kernel, atomicAdd onto zeros)."""
    col, offset = _f(col), _f(offset)
    Cc, H, W = im_shape
    g = np.empty((Cc, H, W), np.float32)
    rc = load(fma).ref_deformable_col2im(_p(col), _p(offset), Cc, H, W, *_geom(kernel, pad, stride, dilate), int(dg), _p(g))
    assert rc == 0, rc
    return g


def deformable_col2im_coord(col, data, offset, kernel, pad, stride, dilate, dg, fma=False):
    """-> grad_offset, same shape as offset (deformable_col2im_coord_gpu_kernel)."""
    col, data, offset = _f(col), _f(data), _f(offset)
    Cc, H, W = data.shape
    g = np.empty_like(offset)
    rc = load(fma).ref_deformable_col2im_coord(_p(col), _p(data), _p(offset), Cc, H, W, *_geom(kernel, pad, stride, dilate), int(dg), _p(g))
    assert rc == 0, rc
    return g


def psroi_forward(data, rois, trans, spatial_scale, output_dim, group_size, pooled, part, sample_per_part, trans_std, fma=False):
    """data [N,C,H,W], rois [R,5], trans [R,2*ncls,part,part] or None -> (top_data, top_count) [R,output_dim,P,P]."""
    data, rois = _f(data), _f(rois)
    trans = None if trans is None else _f(trans)
    N, Cc, H, W = data.shape
    R = rois.shape[0]
    ncls = 1 if trans is None else trans.shape[1] // 2
    top, cnt = np.empty((R, output_dim, pooled, pooled), np.float32), np.empty((R, output_dim, pooled, pooled), np.float32)
    rc = load(fma).ref_psroi_forward(_p(data), _p(rois), _p(trans), N, Cc, H, W, R, int(trans is None), C.c_float(spatial_scale), int(output_dim),
                                     int(group_size), int(pooled), int(part), int(sample_per_part), C.c_float(trans_std), ncls, _p(top), _p(cnt))
    assert rc == 0, rc
    return top, cnt


def psroi_backward(top_diff, top_count, data, rois, trans, spatial_scale, output_dim, group_size, pooled, part, sample_per_part,
                   trans_std, fma=False):
    """-> (in_grad [N,C,H,W], trans_grad like trans | None)."""
    top_diff, top_count, data, rois = _f(top_diff), _f(top_count), _f(data), _f(rois)
    trans = None if trans is None else _f(trans)
    N, Cc, H, W = data.shape
    R = rois.shape[0]
    ncls = 1 if trans is None else trans.shape[1] // 2
    gi = np.empty_like(data)
    gt = None if trans is None else np.empty_like(trans)
    rc = load(fma).ref_psroi_backward(_p(top_diff), _p(top_count), _p(data), _p(rois), _p(trans), N, Cc, H, W, R, int(trans is None),
                                      C.c_float(spatial_scale), int(output_dim), int(group_size), int(pooled), int(part), int(sample_per_part),
                                      C.c_float(trans_std), ncls, _p(gi), _p(gt))
    assert rc == 0, rc
    return gi, gt


def nms(dets_sorted, thresh):
    """The reference's `_nms` (lib/nms/nms_kernel.cu:80-144) on boxes [n,5] already sorted by descending score -> kept row indices."""
    d = _f(dets_sorted)
    n = d.shape[0]
    keep = np.empty(max(n, 1), np.int32)
    num = C.c_int(0)
    rc = load().ref_nms(_p(keep), C.byref(num), _p(d), n, C.c_float(thresh))
    assert rc == 0, rc
    return keep[:num.value].copy()
