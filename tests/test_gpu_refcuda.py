"""The product's HIP kernels against the reference's own CUDA kernels, both running on the MI355X.

oracle/_ref/libref_cuda.so (built by oracle/build_ref.py where /root/reference exists, shipped with the snapshot) holds
deformable_im2col.cuh, deformable_psroi_pooling.cu and nms_kernel.cu compiled unedited for gfx950.  Skipped when that library
is absent (a checkout without /root/reference); the committed vectors of tests/golden/ref_cuda.npz then carry the pin through
tests/test_oracle_refcuda.py.  Sizes here are the benchmark's (res5 map 38 x 63, 300 rois, 6000 sorted boxes)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refcuda as RC  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not RC.available(), reason='oracle/_ref/libref_cuda.so not built (no /root/reference)')]
F = np.float32


def _ops():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    return ops


def test_deformable_im2col_equals_reference_kernel_res5_geometry():
    ops = _ops()
    rng = np.random.default_rng(5)
    C, H, W, k, pad, dil, dg = 64, 38, 63, 3, 2, 2, 4
    data = rng.normal(0, 1, (1, C, H, W)).astype(F)
    off = rng.normal(0, 2.0, (1, 2 * k * k * dg, H, W)).astype(F)
    col, (ho, wo) = ops.deformable_im2col(torch.as_tensor(data).cuda(), torch.as_tensor(off).cuda(), k, 1, dil, pad, dg)
    got = col.cpu().numpy().reshape(1, ho, wo, k * k, C).transpose(0, 4, 3, 1, 2).reshape(C * k * k, ho, wo)
    want = RC.deformable_im2col(data[0], off[0], (k, k), (pad, pad), (1, 1), (dil, dil), dg)
    assert np.array_equal(got, want)


@pytest.mark.parametrize('no_trans', [False, True])
def test_deformable_psroi_pooling_equals_reference_kernel_full_size(no_trans):
    ops = _ops()
    rng = np.random.default_rng(6)
    B, C, H, W, P, R = 2, 256, 38, 63, 7, 600
    data = rng.normal(0, 1, (B, C, H, W)).astype(F)
    x1 = rng.uniform(-40, W * 16 - 40, R); y1 = rng.uniform(-40, H * 16 - 40, R)
    bw = rng.uniform(1, 500, R); bh = rng.uniform(1, 400, R)
    rois = np.stack([np.repeat(np.arange(B), R // B).astype(F), x1, y1, x1 + bw, y1 + bh], 1).astype(F)
    trans = None if no_trans else rng.normal(0, 1.0, (R, 2, P, P)).astype(F)
    out, cnt = ops.deformable_psroi_pool(torch.as_tensor(data).cuda(), torch.as_tensor(rois).cuda(),
                                         None if no_trans else torch.as_tensor(trans).cuda(), 0.0625, C, 1, P, P, 4, 0.1, no_trans,
                                         want_top_count=True)
    want, wcnt = RC.psroi_forward(data, rois, trans, 0.0625, C, 1, P, P, 4, 0.1)
    assert np.array_equal(cnt.cpu().numpy(), wcnt)
    assert np.array_equal(out.cpu().numpy(), want)


def test_psroi_backward_close_to_reference_kernel():
    ops = _ops()
    rng = np.random.default_rng(8)
    B, C, H, W, P, R = 2, 32, 20, 31, 7, 40
    data = rng.normal(0, 1, (B, C, H, W)).astype(F)
    x1 = rng.uniform(-40, W * 16 - 40, R); y1 = rng.uniform(-40, H * 16 - 40, R)
    rois = np.stack([rng.integers(0, B, R).astype(F), x1, y1, x1 + rng.uniform(1, 400, R), y1 + rng.uniform(1, 300, R)], 1).astype(F)
    trans = rng.normal(0, 1.0, (R, 2, P, P)).astype(F)
    gout = rng.normal(0, 1, (R, C, P, P)).astype(F)
    _, cnt = RC.psroi_forward(data, rois, trans, 0.0625, C, 1, P, P, 4, 0.1)
    wi, wt = RC.psroi_backward(gout, cnt, data, rois, trans, 0.0625, C, 1, P, P, 4, 0.1)
    gd, gt = ops.deformable_psroi_pool_bwd(torch.as_tensor(gout).cuda(), torch.as_tensor(data).cuda(), torch.as_tensor(rois).cuda(),
                                           torch.as_tensor(trans).cuda(), 0.0625, C, 1, P, P, 4, 0.1, False)
    assert np.abs(gd.cpu().numpy() - wi).max() <= 2e-5 * np.abs(wi).max()           # (both sides sum with float atomics)
    assert np.abs(gt.cpu().numpy() - wt).max() <= 2e-4 * np.abs(wt).max()


@pytest.mark.parametrize('n,thresh,dup', [(6000, 0.7, False), (6000, 0.7, True), (3000, 0.3, True), (65, 0.5, False)])
def test_nms_equals_reference_nms(n, thresh, dup):
    """The product's `_nms` (same C prototype as lib/nms/gpu_nms.hpp) against the reference's `_nms` on sorted proposals-like boxes."""
    from relnet_amd import nms
    rng = np.random.default_rng(n + int(10 * thresh))
    x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n)
    d = np.stack([x1, y1, x1 + rng.uniform(8, 300, n), y1 + rng.uniform(8, 300, n), np.sort(rng.uniform(0, 1, n))[::-1]], 1).astype(F)
    if dup:
        d[5:n:7, :4] = d[4:n - 1:7, :4][:len(d[5:n:7])]
    want = RC.nms(d, thresh)
    got = nms.gpu_nms(d, thresh, 0)
    assert [int(i) for i in got] == [int(i) for i in want]
