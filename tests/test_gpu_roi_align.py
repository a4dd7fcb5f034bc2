"""ROIAlign (csrc/roi_align.hip) against oracle/roi_align.py and against known answers that follow from the published definition alone
(the reference has no ROIAlign to run: oracle header)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases  # noqa: E402
from oracle import roi_align as ORA  # noqa: E402

pytestmark = pytest.mark.gpu


def _ops():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    return ops


def _rois(n, seed, B=2):
    b = cases.random_boxes(n, seed, min_size=8, max_size=420)
    rng = np.random.default_rng(seed)
    r = np.hstack([rng.integers(0, B, (n, 1)).astype(np.float32), b]).astype(np.float32)
    r[0] = [0, -20.0, -30.0, 50.0, 40.0]              # sticks out of the map at the top left
    r[1] = [1, 900.0, 500.0, 1100.0, 700.0]           # ... and at the bottom right (samples beyond H / W contribute zero)
    r[2] = [0, 100.0, 100.0, 100.5, 100.5]            # smaller than one feature pixel (not aligned: extent clamped to 1)
    return r


@pytest.mark.parametrize('sampling_ratio,aligned', [(2, False), (0, False), (2, True), (3, True)])
def test_roi_align_float32_is_the_oracle_bit_for_bit(sampling_ratio, aligned):
    ops = _ops()
    rng = np.random.default_rng(5)
    data = rng.normal(0, 1, (2, 24, 38, 63)).astype(np.float32)
    rois = _rois(40, 6)
    want = ORA.roi_align(data, rois, (7, 7), 1 / 16.0, sampling_ratio, aligned)
    ref64 = ORA.roi_align(data, rois, (7, 7), 1 / 16.0, sampling_ratio, aligned, dtype=np.float64)
    d, r = torch.as_tensor(data).cuda(), torch.as_tensor(rois).cuda()
    got = ops.roi_align(d, r, (7, 7), 1 / 16.0, sampling_ratio, aligned).cpu().numpy()
    assert np.array_equal(got, want)
    # the float32 kernel against the float64 definition: coordinates near a pixel boundary may fall on the other side in float32
    # (a different pair of corners with nearly the same weights: bilinear interpolation is continuous), so the bound is on values
    assert np.abs(got - ref64).max() <= 1e-4
    # same values from a channels-last map and into a channels-last output
    got_cl = ops.roi_align(d.contiguous(memory_format=torch.channels_last), r, (7, 7), 1 / 16.0, sampling_ratio, aligned, channels_last_out=True)
    assert got_cl.permute(0, 2, 3, 1).is_contiguous() and np.array_equal(got_cl.cpu().numpy(), want)


def test_roi_align_known_answers_from_the_definition():
    ops = _ops()
    H, W, C = 38, 63, 16
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    coef = np.random.default_rng(2).normal(0, 1, (C, 3))
    data = (coef[:, 0, None, None] * yy + coef[:, 1, None, None] * xx + coef[:, 2, None, None])[None].astype(np.float32)      # affine per channel
    # boxes well inside the map: every sample has all four neighbours
    rois = np.array([[0, 64, 48, 400, 300], [0, 160, 80, 167.3, 91.9], [0, 33.3, 20.1, 700.7, 500.2]], np.float32)
    d, r = torch.as_tensor(data).cuda(), torch.as_tensor(rois).cuda()
    for aligned in (False, True):
        got = ops.roi_align(d, r, (7, 7), 1 / 16.0, 2, aligned).cpu().numpy().astype(np.float64)
        off = 0.5 if aligned else 0.0
        for i, roi in enumerate(rois.astype(np.float64)):
            sw, sh = roi[1] / 16 - off, roi[2] / 16 - off
            rw, rh = roi[3] / 16 - off - sw, roi[4] / 16 - off - sh
            if not aligned:
                rw, rh = max(rw, 1.0), max(rh, 1.0)
            cy = sh + (np.arange(7) + 0.5) * rh / 7
            cx = sw + (np.arange(7) + 0.5) * rw / 7
            want = coef[:, 0, None, None] * cy[None, :, None] + coef[:, 1, None, None] * cx[None, None, :] + coef[:, 2, None, None]
            assert np.abs(got[i] - want).max() <= 2e-4 * np.abs(want).max(), (aligned, i)
    # a constant map pools to the constant wherever the bin's samples are inside the map
    const = torch.full((1, 8, H, W), 3.25).cuda()
    got = ops.roi_align(const, r, (7, 7), 1 / 16.0, 2).cpu().numpy()
    assert np.array_equal(got, np.full_like(got, 3.25))
    # a one-hot map returns the bilinear weight of that pixel: one sample per bin at the bin centre
    hot = torch.zeros((1, 1, H, W)); hot[0, 0, 10, 20] = 1.0
    roi = torch.tensor([[0, 16 * 19.25, 16 * 9.5, 16 * 20.25, 16 * 10.5]])          # 1 x 1 feature pixel box, centre (y 10.0, x 19.75)
    got = ops.roi_align(hot.cuda(), roi.cuda(), (1, 1), 1 / 16.0, 1).cpu().numpy()
    assert abs(float(got[0, 0, 0, 0]) - 0.75) < 1e-6


@pytest.mark.parametrize('C', [256, 64, 24])
def test_roi_align_bf16_channels_last_fast_path(C):
    """The throughput form (thread = bin x 8 channels, 16-byte corner loads) on the detector's layout: NHWC bf16 in, (R, 7, 7, C) out."""
    ops = _ops()
    rng = np.random.default_rng(8)
    data = torch.as_tensor(rng.normal(0, 1, (2, 38, 63, C)).astype(np.float32)).cuda().to(torch.bfloat16)
    rois = _rois(300, 9)
    d_nchw = data.permute(0, 3, 1, 2)                                   # logical NCHW, channels-last memory
    got = ops.roi_align(d_nchw, torch.as_tensor(rois).cuda(), (7, 7), 1 / 16.0, 2, channels_last_out=True)
    want = ORA.roi_align(data.float().cpu().numpy().transpose(0, 3, 1, 2), rois, (7, 7), 1 / 16.0, 2)
    g = got.float().cpu().numpy()
    assert np.abs(g - want).max() <= 2 ** -8 * max(1.0, np.abs(want).max())       # one bf16 rounding of the float32 result
    # the generic kernel (NCHW-contiguous bf16 copy) rounds the same float32 values: identical bits
    got2 = ops.roi_align(d_nchw.contiguous(), torch.as_tensor(rois).cuda(), (7, 7), 1 / 16.0, 2)
    assert torch.equal(got2, got.contiguous())


@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
def test_roi_align_backward_is_the_adjoint(layout):
    """<roi_align(x), g> == <x, roi_align_bwd(g)> for random x, g (the operator is linear in x), and against float64 autograd of the
    definition written with torch ops."""
    ops = _ops()
    rng = np.random.default_rng(11)
    B, C, H, W = 2, 16, 20, 30
    x = rng.normal(0, 1, (B, C, H, W)).astype(np.float32)
    rois = _rois(24, 12)
    rois[:, 1:] *= 0.45                                                  # fit the 20 x 30 map at scale 1/16
    g = rng.normal(0, 1, (24, C, 7, 7)).astype(np.float32)
    xd, rd, gd = torch.as_tensor(x).cuda(), torch.as_tensor(rois).cuda(), torch.as_tensor(g).cuda()
    if layout == 'nhwc':
        xd = xd.contiguous(memory_format=torch.channels_last)
    y = ops.roi_align(xd, rd, (7, 7), 1 / 16.0, 2)
    gx = ops.roi_align_bwd(gd, rd, (B, C, H, W), 1 / 16.0, 2, channels_last=(layout == 'nhwc'))
    lhs = float((y.double() * gd.double()).sum())
    rhs = float((xd.double() * gx.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
    # float64 autograd: gather form of the same samples
    _, samples = ORA.roi_align(x[:, :1], rois, (7, 7), 1 / 16.0, 2, dtype=np.float64, return_samples=True)
    xt = torch.as_tensor(x).double().requires_grad_(True)
    out = torch.zeros((24, C, 7, 7), dtype=torch.float64)
    for (r, ph, pw, yy, xx) in samples:
        k = ORA._corners(np.float64(yy), np.float64(xx), H, W, np.float64)
        if k is None:
            continue
        yl, xl, yh, xh, w1, w2, w3, w4 = k
        b = int(rois[r, 0])
        out[r, :, ph, pw] = out[r, :, ph, pw] + (w1 * xt[b, :, yl, xl] + w2 * xt[b, :, yl, xh] + w3 * xt[b, :, yh, xl] + w4 * xt[b, :, yh, xh]) / 4.0
    (out * torch.as_tensor(g).double()).sum().backward()
    assert float((gx.cpu().double() - xt.grad).abs().max()) <= 2e-5 * float(xt.grad.abs().max())


def test_roi_align_through_the_mx_facade():
    """mx.contrib.sym.ROIAlign(data, rois, pooled_size, spatial_scale, sample_ratio) on the graph facade == ops.roi_align."""
    import relnet_amd  # noqa: F401
    from relnet_amd import mx as MX, ops
    mx = MX.install()
    data, rois = mx.sym.Variable('data'), mx.sym.Variable('rois')
    s = mx.contrib.sym.ROIAlign(data=data, rois=rois, pooled_size=(7, 7), spatial_scale=0.0625, sample_ratio=2, name='roi_align')
    rng = np.random.default_rng(3)
    x = torch.as_tensor(rng.normal(0, 1, (1, 32, 38, 63)).astype(np.float32)).cuda()
    r_ = _rois(20, 4, B=1)
    r_[:, 0] = 0                                                     # one image
    r = torch.as_tensor(r_).cuda()
    assert s.infer_shape(data=tuple(x.shape), rois=tuple(r.shape))[1] == [(20, 32, 7, 7)]
    ex = s.bind(mx.gpu(0), args={}, dtype=torch.float32)
    out = ex.forward(is_train=False, data=x, rois=r)[0]
    assert torch.equal(out.data.float(), ops.roi_align(x, r, (7, 7), 0.0625, 2))


def test_detector_with_roi_align_feeds_the_head_with_the_oracle_pooled_features():
    """Detector(cfg.roi_align = True): the pooled features that reach fc_new_1 are ROIAlign of THIS run's conv_new_1_relu map on THIS run's
    proposals (teacher forced), in the (ph, pw, c) order the permuted fc_new_1 weight expects."""
    import relnet_amd  # noqa: F401
    from relnet_amd import backbone, detector
    p = backbone.init_params(seed=7)
    cfg = detector.Config(); cfg.roi_align = True; cfg.rpn_post_nms_top_n = 60
    H, W = 256, 320
    det = detector.Detector(p, dtype=torch.bfloat16, cfg=cfg, im_hw=(H, W))
    data = torch.randn(2, 3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    out = det.forward(data, torch.tensor([[H, W, 1.0]] * 2).cuda(), keep_features=True)
    feat = out['features']['conv_new_1_relu'].float().cpu().numpy()
    rois = out['rois'].view(-1, 5).cpu().numpy()
    want = ORA.roi_align(feat, rois, (7, 7), 1 / 16.0, 2).transpose(0, 2, 3, 1).reshape(2, 60, -1)
    got = out['pooled'].float().cpu().numpy()
    assert np.abs(got - want).max() <= 2 ** -8 * max(1.0, np.abs(want).max())
    assert torch.isfinite(out['cls_score']).all() and int(out['num_detections'].sum()) > 0
