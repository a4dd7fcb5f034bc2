"""GPU parity of the training-loss kernels against oracle/losses.py, whose gradients
come from torch-CPU autograd.  fp32 elementwise math: 2e-6 relative (expf / logf vs libm)."""
import os
import sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import losses as OL  # noqa: E402

pytestmark = pytest.mark.gpu
F = np.float32


def _close(a, b, rtol=2e-6, atol=1e-9):
    a = a.cpu().numpy() if torch.is_tensor(a) else a
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol + rtol * np.abs(b).max())


def test_softmax_output_rcnn_head_with_ohem_ignore():
    import relnet_amd  # noqa: F401
    from relnet_amd import losses
    rng = np.random.default_rng(1)
    n, C = 308, 81
    data = rng.normal(0, 2, (n, C)).astype(F)
    label = rng.integers(0, C, n).astype(F)
    label[rng.random(n) < 0.58] = -1                      # OHEM keeps 128 of ~300 rois, the rest is ignored
    prob, grad = losses.softmax_output(torch.as_tensor(data).cuda(), torch.as_tensor(label).cuda(), use_ignore=True)
    wp, wg = OL.softmax_output(data, label, use_ignore=True)
    _close(prob, wp); _close(grad, wg)
    assert np.all(grad.cpu().numpy()[label == -1] == 0)
    # without use_ignore every row counts ('valid' then divides by the number of rows)
    label2 = np.abs(label)
    prob, grad = losses.softmax_output(torch.as_tensor(data).cuda(), torch.as_tensor(label2).cuda())
    _close(grad, OL.softmax_output(data, label2)[1])
    # inference use: probabilities only
    prob, none = losses.softmax_output(torch.as_tensor(data).cuda())
    _close(prob, wp); assert none is None


def test_softmax_output_rpn_multi_output_full_size():
    """rpn_cls_prob: Reshape(0,2,-1,0) logits [1, 2, 12*38, 63], labels {-1, 0, 1} with 256 valid anchors."""
    import relnet_amd  # noqa: F401
    from relnet_amd import losses
    rng = np.random.default_rng(2)
    data = rng.normal(0, 1.5, (2, 2, 12 * 38, 63)).astype(F)
    label = np.full((2, 12 * 38 * 63), -1, F)
    for b in range(2):
        sel = rng.choice(label.shape[1], 256, replace=False)
        label[b, sel] = (rng.random(256) < 0.5).astype(F)
    prob, grad = losses.softmax_output(torch.as_tensor(data).cuda(), torch.as_tensor(label).cuda(), multi_output=True,
                                       use_ignore=True)
    wp, wg = OL.softmax_output(data, label, multi_output=True, use_ignore=True)
    _close(prob, wp); _close(grad, wg)
    g = grad.cpu().numpy()
    assert np.count_nonzero(np.abs(g).sum(1).reshape(2, -1)) == 512
    # all anchors ignored: gradient is exactly zero, no division by zero
    _, g0 = losses.softmax_output(torch.as_tensor(data).cuda(), torch.full((2, 12 * 38 * 63), -1.0).cuda(), multi_output=True,
                                  use_ignore=True)
    assert float(g0.abs().max()) == 0.0


@pytest.mark.parametrize('sigma,scale', [(3.0, 1.0 / 256), (1.0, 1.0 / 128)])
def test_smooth_l1_loss(sigma, scale):
    import relnet_amd  # noqa: F401
    from relnet_amd import losses
    rng = np.random.default_rng(3)
    pred = rng.normal(0, 0.6, (308, 8)).astype(F)
    target = rng.normal(0, 0.6, (308, 8)).astype(F)
    pred[0, :4] = target[0, :4] + np.array([1 / sigma ** 2, -1 / sigma ** 2, 0, 1e-8], F)   # on the branch boundary
    weight = (rng.random((308, 8)) < 0.3).astype(F)
    loss, grad = losses.smooth_l1_loss(torch.as_tensor(pred).cuda(), torch.as_tensor(target).cuda(),
                                       torch.as_tensor(weight).cuda(), sigma, scale)
    wl, wg = OL.smooth_l1_loss(pred, target, weight, sigma, scale)
    _close(loss, wl); _close(grad, wg)


def test_nms_loss():
    import relnet_amd  # noqa: F401
    from relnet_amd import losses
    rng = np.random.default_rng(4)
    score = (rng.random((100, 80, 5)) * rng.random((100, 80, 5))).astype(F)
    score[0, 0] = [0.0, 1e-9, 0.5, 1.0 - 1e-7, 1.0]
    target = (rng.random((100, 80, 5)) < 0.02).astype(F)
    target[0, 0] = [1, 1, 0, 0, 1]
    pos, neg, grad = losses.nms_loss(torch.as_tensor(score).cuda(), torch.as_tensor(target).cuda(), 100, 5)
    wp, wn, wg = OL.nms_loss(score, target, 100, 5)
    fin = np.isfinite(wp) & np.isfinite(wn) & np.isfinite(wg)
    assert fin.mean() > 0.999
    np.testing.assert_allclose(pos.cpu().numpy()[fin], wp[fin], rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(neg.cpu().numpy()[fin], wn[fin], rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(grad.cpu().numpy()[fin], wg[fin], rtol=3e-6, atol=1e-9)
    assert np.array_equal(np.isfinite(grad.cpu().numpy()), np.isfinite(wg))
