#!/usr/bin/env python
"""Golden vectors from the reference's own CUDA kernels, compiled unedited for gfx950 (oracle/build_ref.py ->
oracle/_ref/libref_cuda.so) and RUN ON THE GPU BOX:

    python tests/golden/gen_golden_gpu.py [out.npz]        (default tests/golden/ref_cuda.npz)

Inputs are seeded (tests/golden/cases.py-style generators below); the outputs of deformable_im2col_gpu_kernel,
deformable_col2im(_coord)_gpu_kernel, DeformablePSROIPoolForward / BackwardAccKernel and `_nms` are stored next to them, once
with -ffp-contract=off and (forward kernels) once with the compiler's default contraction.  tests/test_oracle_refcuda.py holds
oracle/deform.py and oracle/nms.py to these vectors on any machine; tests/test_gpu_refcuda.py compares the product's HIP kernels
with the reference kernels directly.  Atomic accumulations (col2im, psroi backward) are stored too, but their summation order is
not deterministic: they are compared with a tolerance."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refcuda as RC  # noqa: E402


def dcn_case(seed, C, H, W, k, pad, stride, dil, dg, off_std):
    rng = np.random.default_rng(seed)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    data = rng.standard_normal((C, H, W)).astype(np.float32)
    offset = (rng.standard_normal((dg * 2 * k * k, Ho, Wo)) * off_std).astype(np.float32)
    # a few offsets that leave the map, land exactly on integers and on the last row / column
    flat = offset.reshape(-1)
    idx = rng.integers(0, flat.size, 24)
    flat[idx[:8]] = np.float32(50.0); flat[idx[8:16]] = np.float32(-50.0); flat[idx[16:]] = np.round(flat[idx[16:]])
    return data, offset, (k, k), (pad, pad), (stride, stride), (dil, dil), dg


def psroi_case(seed, N, out_dim, group, H, W, R, ncls, part, no_trans):
    rng = np.random.default_rng(seed)
    C = out_dim * group * group
    data = rng.standard_normal((N, C, H, W)).astype(np.float32)
    x1 = rng.uniform(-20, W * 16 - 40, R); y1 = rng.uniform(-20, H * 16 - 40, R)
    bw = rng.uniform(1, W * 12, R); bh = rng.uniform(1, H * 12, R)
    rois = np.stack([rng.integers(0, N, R).astype(np.float64), x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
    rois[0, 1:] = [10.5, 20.5, 10.5, 20.5]                 # degenerate roi, .5 coordinates (CUDA round: half away from zero)
    rois[1, 1:] = [-100, -100, -90, -95]                   # entirely outside
    trans = None if no_trans else (rng.standard_normal((R, 2 * ncls, part, part)) * 1.5).astype(np.float32)
    return data, rois, trans


def nms_case(seed, n, ties):
    rng = np.random.default_rng(seed)
    x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n)
    w = rng.uniform(8, 300, n); h = rng.uniform(8, 300, n)
    s = np.sort(rng.uniform(0, 1, n))[::-1]
    d = np.stack([x1, y1, x1 + w, y1 + h, s], 1).astype(np.float32)
    if ties:                                               # duplicated boxes (IoU exactly 1) and IoU == threshold-ish clusters
        d[5:n:7, :4] = d[4:n - 1:7, :4][:len(d[5:n:7])]
    return d


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tests', 'golden', 'ref_cuda.npz')
    z = {}
    dcn = {'a': (11, 8, 9, 11, 3, 1, 1, 1, 2, 1.0), 'b': (12, 8, 12, 10, 3, 2, 1, 2, 4, 2.5), 'c': (13, 4, 11, 13, 3, 1, 2, 1, 1, 0.7)}
    for name, spec in dcn.items():
        data, offset, kernel, pad, stride, dil, dg = dcn_case(*spec)
        z['dcn_%s_spec' % name] = np.asarray(spec, np.float64)
        z['dcn_%s_data' % name], z['dcn_%s_offset' % name] = data, offset
        col = RC.deformable_im2col(data, offset, kernel, pad, stride, dil, dg)
        z['dcn_%s_col' % name] = col
        z['dcn_%s_col_fma' % name] = RC.deformable_im2col(data, offset, kernel, pad, stride, dil, dg, fma=True)
        rng = np.random.default_rng(spec[0] + 100)
        gcol = rng.standard_normal(col.shape).astype(np.float32)
        z['dcn_%s_gcol' % name] = gcol
        z['dcn_%s_grad_im' % name] = RC.deformable_col2im(gcol, offset, data.shape, kernel, pad, stride, dil, dg)
        z['dcn_%s_grad_offset' % name] = RC.deformable_col2im_coord(gcol, data, offset, kernel, pad, stride, dil, dg)
    ps = {'trans': (21, 2, 8, 1, 12, 15, 24, 1, 7, False), 'notrans': (22, 2, 6, 1, 12, 15, 16, 1, 7, True),
          'group': (23, 1, 4, 3, 10, 12, 12, 2, 3, False)}
    for name, spec in ps.items():
        data, rois, trans = psroi_case(*spec)
        out_dim, group, part = spec[2], spec[3], spec[8]
        pooled = 7 if group == 1 else 3
        args = (0.0625, out_dim, group, pooled, part, 4, 0.1)
        z['psroi_%s_spec' % name] = np.asarray(spec[:9] + (int(spec[9]),) + (pooled,), np.float64)
        z['psroi_%s_data' % name], z['psroi_%s_rois' % name] = data, rois
        if trans is not None:
            z['psroi_%s_trans' % name] = trans
        top, cnt = RC.psroi_forward(data, rois, trans, *args)
        z['psroi_%s_top' % name], z['psroi_%s_count' % name] = top, cnt
        z['psroi_%s_top_fma' % name] = RC.psroi_forward(data, rois, trans, *args, fma=True)[0]
        gtop = np.random.default_rng(spec[0] + 100).standard_normal(top.shape).astype(np.float32)
        gi, gt = RC.psroi_backward(gtop, cnt, data, rois, trans, *args)
        z['psroi_%s_gtop' % name], z['psroi_%s_in_grad' % name] = gtop, gi
        if gt is not None:
            z['psroi_%s_trans_grad' % name] = gt
    for name, (seed, n, ties, thr) in {'plain': (31, 700, False, 0.7), 'ties': (32, 450, True, 0.7), 'low': (33, 300, True, 0.3),
                                       'one': (34, 1, False, 0.5), 'block': (35, 64, False, 0.5), 'block1': (36, 65, True, 0.5)}.items():
        d = nms_case(seed, n, ties)
        z['nms_%s_dets' % name], z['nms_%s_thresh' % name] = d, np.float32(thr)
        z['nms_%s_keep' % name] = RC.nms(d, thr)
    np.savez_compressed(out, **z)
    print('wrote %s: %d arrays, %.1f KB' % (out, len(z), os.path.getsize(out) / 1e3))


if __name__ == '__main__':
    main()
