"""oracle/deform.py and oracle/nms.py held to the reference's OWN CUDA kernels.

tests/golden/ref_cuda.npz was produced on an MI355X by tests/golden/gen_golden_gpu.py from oracle/_ref/libref_cuda.so =
relation_rcnn/operator_cxx/nn/deformable_im2col.cuh, relation_rcnn/operator_cxx/deformable_psroi_pooling.cu and
lib/nms/nms_kernel.cu compiled UNEDITED for gfx950 (oracle/build_ref.py).  This closes the "restated, unpinned" gap of the
deformable operators and of the CUDA NMS order:

  * forward sampling (column matrix, pooled bins, top_count): the numpy restatement equals the reference kernels BIT FOR BIT when
    those are compiled with -ffp-contract=off; the fused-multiply-add build (nvcc's default --fmad=true; WHICH products it
    fuses is not knowable without nvcc) gives the identical column matrix and pooled bins within 1.1e-6 of the output scale
    (asserted as a bound, not a pin);
  * `_nms`: identical keep lists, including duplicated boxes (IoU exactly 1), 64 / 65-box block boundaries and one box;
  * backward kernels (atomicAdd accumulation, order not deterministic): oracle/deform_torch.py's float64 autograd within 1e-5.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import deform, nms as ONMS  # noqa: E402

Z = np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_cuda.npz'))


def _dcn(name):
    seed, C, H, W, k, pad, stride, dil, dg, _ = Z['dcn_%s_spec' % name]
    g = lambda v: (int(v), int(v))
    return Z['dcn_%s_data' % name], Z['dcn_%s_offset' % name], g(k), g(pad), g(stride), g(dil), int(dg)


@pytest.mark.parametrize('name', ['a', 'b', 'c'])
def test_im2col_restatement_equals_reference_kernel(name):
    data, offset, kernel, pad, stride, dil, dg = _dcn(name)
    got = deform.deformable_im2col(data, offset, kernel, pad, stride, dil, dg)
    want = Z['dcn_%s_col' % name]
    assert got.shape == want.shape and np.array_equal(got, want)
    assert (want == 0).any() and (want != 0).any()            # taps that leave the map and taps that do not
    fma = Z['dcn_%s_col_fma' % name]                          # contraction on: a handful of last-bit differences at most
    assert np.abs(fma - want).max() <= 4 * np.spacing(np.abs(want).max())


@pytest.mark.parametrize('name', ['a', 'b', 'c'])
def test_col2im_kernels_equal_float64_autograd(name):
    """deformable_col2im_gpu_kernel / deformable_col2im_coord_gpu_kernel against autograd of the restated forward."""
    from oracle import deform_torch as DT
    data, offset, kernel, pad, stride, dil, dg = _dcn(name)
    td = torch.tensor(data, dtype=torch.float64, requires_grad=True)
    to = torch.tensor(offset, dtype=torch.float64, requires_grad=True)
    col = DT.deformable_im2col(td, to, kernel, pad, stride, dil, dg)
    gcol = torch.as_tensor(Z['dcn_%s_gcol' % name]).double()
    (col.reshape(gcol.shape) * gcol).sum().backward()
    gi, go = Z['dcn_%s_grad_im' % name], Z['dcn_%s_grad_offset' % name]
    assert np.abs(gi - td.grad.numpy()).max() <= 1e-5 * np.abs(gi).max()
    # the coordinate gradient is not differentiable where a sample sits exactly on a cell border (the generator rounds a few
    # offsets to integers on purpose): compare where both agree it is smooth
    ref = to.grad.numpy()
    close = np.abs(go - ref) <= 1e-4 * np.abs(ref).max()
    assert close.mean() >= 0.97, close.mean()


def _psroi(name):
    spec = Z['psroi_%s_spec' % name]
    out_dim, group, part, no_trans, pooled = int(spec[2]), int(spec[3]), int(spec[8]), bool(spec[9]), int(spec[10])
    trans = None if no_trans else Z['psroi_%s_trans' % name]
    return Z['psroi_%s_data' % name], Z['psroi_%s_rois' % name], trans, out_dim, group, pooled, part, no_trans


@pytest.mark.parametrize('name', ['trans', 'notrans', 'group'])
def test_psroi_restatement_equals_reference_kernel(name):
    data, rois, trans, out_dim, group, pooled, part, no_trans = _psroi(name)
    got, cnt = deform.deformable_psroi_pooling(data, rois, trans, 0.0625, out_dim, group, pooled, part, 4, 0.1, no_trans)
    assert np.array_equal(cnt, Z['psroi_%s_count' % name])
    want = Z['psroi_%s_top' % name]
    assert np.array_equal(got, want)
    assert (cnt == 0).any() and (cnt == 16).any() and ((cnt > 0) & (cnt < 16)).any()      # outside / inside / clipped bins
    # contraction on (nvcc's default): 13-15 % of the bins move, by at most 1.1e-6 of the output scale (measured: gen_golden_gpu.py)
    assert np.abs(Z['psroi_%s_top_fma' % name] - want).max() <= 4e-6 * np.abs(want).max()


@pytest.mark.parametrize('name', ['trans', 'notrans', 'group'])
def test_psroi_backward_kernel_equals_float64_autograd(name):
    from oracle import deform_torch as DT
    data, rois, trans, out_dim, group, pooled, part, no_trans = _psroi(name)
    td = torch.tensor(data, dtype=torch.float64, requires_grad=True)
    tt = None if no_trans else torch.tensor(trans, dtype=torch.float64, requires_grad=True)
    y = DT.deformable_psroi_pooling(td, rois, tt, 0.0625, out_dim, group, pooled, part, 4, 0.1, no_trans)
    (y * torch.as_tensor(Z['psroi_%s_gtop' % name]).double()).sum().backward()
    gi = Z['psroi_%s_in_grad' % name]
    assert np.abs(gi - td.grad.numpy()).max() <= 2e-5 * np.abs(gi).max()
    if not no_trans:
        gt = Z['psroi_%s_trans_grad' % name]
        assert np.abs(gt - tt.grad.numpy()).max() <= 2e-4 * np.abs(gt).max()


@pytest.mark.parametrize('name', ['plain', 'ties', 'low', 'one', 'block', 'block1'])
def test_nms_restatement_equals_reference_nms(name):
    d, thr = Z['nms_%s_dets' % name], float(Z['nms_%s_thresh' % name])
    want = Z['nms_%s_keep' % name]
    got = ONMS.nms_sorted_f32(d[:, :4], thr)
    assert [int(i) for i in got] == [int(i) for i in want]
    assert len(want) >= 1 and (name in ('one',) or len(want) < len(d))
