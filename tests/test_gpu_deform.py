"""GPU parity of the DCN operators against oracle/deform.py.

Tolerances: sampling (column matrix, pooled bins, top_count) is bit-exact -- the device code and the
oracle execute the same separately rounded fp32 operations; convolution outputs (a GEMM over K = kh*kw*C)
are compared at 1e-5 * sqrt(K) relative to the output scale (summation order differs)."""
import os
import sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import deform  # noqa: E402

pytestmark = pytest.mark.gpu
F = np.float32


def _mods():
    import relnet_amd
    from relnet_amd import ops, operator_cxx
    return ops, operator_cxx


def _bf16_round(x):
    return torch.as_tensor(x).to(torch.bfloat16).float().numpy()


def _col_to_oracle_layout(col, B, C, Ho, Wo, kk):
    """[B*Ho*Wo, kk*C] (tap, c) -> [B, C*kk, Ho, Wo] with row c*kk + tap."""
    return col.reshape(B, Ho, Wo, kk, C).transpose(0, 4, 3, 1, 2).reshape(B, C * kk, Ho, Wo)


CASES = [
    # B, C, H, W, k, pad, stride, dil, dg, offset sigma
    (2, 16, 13, 17, 3, 2, 1, 2, 4, 2.0),
    (1, 8, 12, 9, 3, 1, 2, 1, 1, 4.0),
    (1, 32, 38, 63, 3, 2, 1, 2, 4, 1.0),        # res5 geometry, fewer channels
]


@pytest.mark.parametrize("case", CASES)
def test_im2col_fp32_nchw_bit_exact(case):
    ops, _ = _mods()
    B, C, H, W, k, pad, st, dil, dg, sig = case
    rng = np.random.default_rng(7)
    data = rng.normal(0, 1, (B, C, H, W)).astype(F)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // st + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // st + 1
    off = rng.normal(0, sig, (B, 2 * k * k * dg, Ho, Wo)).astype(F)
    col, (ho, wo) = ops.deformable_im2col(torch.as_tensor(data).cuda(), torch.as_tensor(off).cuda(), k, st, dil, pad, dg)
    assert (ho, wo) == (Ho, Wo)
    got = _col_to_oracle_layout(col.cpu().numpy(), B, C, Ho, Wo, k * k)
    want = np.stack([deform.deformable_im2col(data[b], off[b], (k, k), (pad, pad), (st, st), (dil, dil), dg) for b in range(B)])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", CASES)
def test_im2col_bf16_channels_last_bit_exact(case):
    """Throughput layout: channels-last bf16 activations, vector kernel; fp32 blend rounded once to bf16."""
    ops, _ = _mods()
    B, C, H, W, k, pad, st, dil, dg, sig = case
    rng = np.random.default_rng(8)
    data = _bf16_round(rng.normal(0, 1, (B, C, H, W)).astype(F))
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // st + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // st + 1
    off = rng.normal(0, sig, (B, 2 * k * k * dg, Ho, Wo)).astype(F)
    x = torch.as_tensor(data).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    o = torch.as_tensor(off).cuda().contiguous(memory_format=torch.channels_last)      # offsets channels-last too
    col, _ = ops.deformable_im2col(x, o, k, st, dil, pad, dg)
    assert col.dtype == torch.bfloat16
    got = _col_to_oracle_layout(col.float().cpu().numpy(), B, C, Ho, Wo, k * k)
    want = np.stack([deform.deformable_im2col(data[b], off[b], (k, k), (pad, pad), (st, st), (dil, dil), dg) for b in range(B)])
    assert np.array_equal(got, _bf16_round(want))
    # the generic-stride kernel (NCHW bf16) gives the same bits
    col2, _ = ops.deformable_im2col(torch.as_tensor(data).cuda().to(torch.bfloat16), torch.as_tensor(off).cuda(), k, st, dil, pad, dg)
    assert torch.equal(col, col2)


def test_deformable_convolution_operator_fp32():
    """`mx.contrib.sym.DeformableConvolution` call surface, fp32 NCHW, against the oracle."""
    _, cxx = _mods()
    rng = np.random.default_rng(9)
    B, C, H, W, Co, k, pad, dil, dg = 2, 64, 19, 23, 48, 3, 2, 2, 4
    data = rng.normal(0, 1, (B, C, H, W)).astype(F)
    off = rng.normal(0, 1.5, (B, 2 * k * k * dg, H, W)).astype(F)
    wgt = rng.normal(0, 0.05, (Co, C, k, k)).astype(F)
    bias = rng.normal(0, 0.1, Co).astype(F)
    t = lambda a: torch.as_tensor(a).cuda()
    out = cxx.contrib.DeformableConvolution(data=t(data), offset=t(off), weight=t(wgt), bias=t(bias), num_filter=Co,
                                            pad=(pad, pad), kernel=(k, k), num_deformable_group=dg, stride=(1, 1),
                                            dilate=(dil, dil))
    want = deform.deformable_convolution(data, off, wgt, bias, (k, k), (1, 1), (dil, dil), (pad, pad), dg)
    assert out.shape == want.shape
    tol = 1e-5 * np.sqrt(k * k * C) * np.abs(want).max()
    assert np.abs(out.cpu().numpy() - want).max() <= tol
    out2 = cxx.contrib.DeformableConvolution(data=t(data), offset=t(off), weight=t(wgt), num_filter=Co, pad=(pad, pad),
                                             kernel=(k, k), num_deformable_group=dg, dilate=(dil, dil), no_bias=True)
    assert np.abs(out2.cpu().numpy() + bias[None, :, None, None] - want).max() <= 2 * tol


def test_deformable_convolution_errors():
    _, cxx = _mods()
    z = lambda *s: torch.zeros(s, device='cuda')
    with pytest.raises(ValueError):          # offset map of the wrong size
        cxx.contrib.DeformableConvolution(data=z(1, 8, 10, 10), offset=z(1, 18, 9, 10), weight=z(4, 8, 3, 3), num_filter=4,
                                          pad=(1, 1), kernel=(3, 3), no_bias=True)
    with pytest.raises(ValueError):          # offset channels do not match num_deformable_group
        cxx.contrib.DeformableConvolution(data=z(1, 8, 10, 10), offset=z(1, 18, 10, 10), weight=z(4, 8, 3, 3), num_filter=4,
                                          pad=(1, 1), kernel=(3, 3), num_deformable_group=2, no_bias=True)
    with pytest.raises(ValueError):          # bias missing
        cxx.contrib.DeformableConvolution(data=z(1, 8, 10, 10), offset=z(1, 18, 10, 10), weight=z(4, 8, 3, 3), num_filter=4,
                                          pad=(1, 1), kernel=(3, 3))


def test_zero_offsets_equal_own_convolution_full_size():
    """Full res5 size (512 channels, 38x63, dilation 2), bf16: with zero offsets the deformable path must
    reproduce the implicit-GEMM convolution kernel bit for bit up to GEMM tiling (same products, fp32 sums)."""
    ops, _ = _mods()
    torch.manual_seed(3)
    B, C, H, W, Co = 2, 512, 38, 63, 512
    x = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    w = (torch.randn(Co, C, 3, 3, device='cuda') * 0.02)
    wp = ops.pack_conv_weight(w)
    bias = torch.randn(Co, device='cuda') * 0.1
    ref = ops.conv2d_nhwc(x, wp, bias, ksize=3, stride=1, pad=2, dil=2, relu=True)
    off = torch.zeros(B, H, W, 72, device='cuda').permute(0, 3, 1, 2)
    out = ops.deformable_conv(x.permute(0, 3, 1, 2), off, wp, bias, 3, 1, 2, 2, 4, relu=True)
    assert out.shape == (B, Co, H, W)
    d = (out.permute(0, 2, 3, 1).float() - ref.float()).abs().max().item()
    assert d <= 2e-2 * ref.float().abs().max().item()


@pytest.mark.parametrize("B,C,H,W,Co,dg,stride,sigma", [(2, 512, 38, 63, 512, 4, 1, 2.0), (1, 64, 19, 23, 36, 1, 1, 6.0),
                                                       (3, 128, 21, 30, 256, 4, 2, 3.0)])
def test_deformable_conv_bf16_channels_last_offsets_layouts(B, C, H, W, Co, dg, stride, sigma):
    """The throughput form (bf16 channels-last sampling kernel + NT GEMM, the res5 shape on the hand-scheduled ring kernel) against
    float32 torch on the column matrix pinned bit-exact above; offsets given in NCHW and in channels-last memory (the layout the
    offset convolution's GEMM produces) must give the identical tensor."""
    ops, _ = _mods()
    torch.manual_seed(11)
    x = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16).permute(0, 3, 1, 2)
    w = torch.randn(Co, C, 3, 3, device='cuda') * 0.02
    wp = ops.pack_conv_weight(w)
    bias = torch.randn(Co, device='cuda') * 0.1
    Ho = (H + 4 - 5) // stride + 1; Wo = (W + 4 - 5) // stride + 1
    off = torch.randn(B, 18 * dg, Ho, Wo, device='cuda') * sigma
    col, _ = ops.deformable_im2col(x, off, 3, stride, 2, 2, dg, col_dtype=torch.bfloat16)
    for relu in (True, False):
        got = ops.deformable_conv(x, off, wp, bias, 3, stride, 2, 2, dg, relu=relu)
        want = col.float() @ wp.float().t() + bias
        want = (want.relu() if relu else want).view(B, Ho, Wo, Co).permute(0, 3, 1, 2)
        assert got.shape == (B, Co, Ho, Wo)
        assert (got.float() - want).abs().max().item() <= 2 ** -7 * want.abs().max().item()
    off_cl = off.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert torch.equal(ops.deformable_conv(x, off_cl, wp, bias, 3, stride, 2, 2, dg), ops.deformable_conv(x, off, wp, bias, 3, stride, 2, 2, dg))


PSROI_CASES = [
    # group, no_trans, num_classes, output_dim, spp, trans_std
    (1, True, 1, 16, 4, 0.0),
    (1, False, 1, 16, 4, 0.1),
    (3, False, 2, 8, 2, 0.1),
    (1, False, 2, 128, 4, 0.1),                    # 64 channels per class: the wave-reduced offset-gradient path of the backward
]


def _rois(rng, R, B, W, H):
    x1 = rng.uniform(-40, W * 16 - 40, R); y1 = rng.uniform(-40, H * 16 - 40, R)
    w = rng.uniform(1, 500, R); h = rng.uniform(1, 400, R)
    r = np.stack([rng.integers(0, B, R).astype(F), x1, y1, x1 + w, y1 + h], 1).astype(F)
    r[0, 1:] = [50.5, 60.5, 50.5, 60.5]            # degenerate roi; .5 exercises round-half-away
    return r


@pytest.mark.parametrize("case", PSROI_CASES)
def test_psroi_fp32_bit_exact(case):
    ops, _ = _mods()
    group, no_trans, ncls, od, spp, tstd = case
    rng = np.random.default_rng(11)
    B, H, W, P, R = 2, 20, 31, 7, 40
    data = rng.normal(0, 1, (B, od * group * group, H, W)).astype(F)
    rois = _rois(rng, R, B, W, H)
    trans = None if no_trans else rng.normal(0, 1.0, (R, 2 * ncls, P, P)).astype(F)
    out, cnt = ops.deformable_psroi_pool(torch.as_tensor(data).cuda(), torch.as_tensor(rois).cuda(),
                                         None if no_trans else torch.as_tensor(trans).cuda(), 0.0625, od, group, P, P, spp,
                                         tstd, no_trans, want_top_count=True)
    want, wcnt = deform.deformable_psroi_pooling(data, rois, trans, 0.0625, od, group, P, P, spp, tstd, no_trans)
    assert np.array_equal(cnt.cpu().numpy(), wcnt)
    assert np.array_equal(out.cpu().numpy(), want)
    assert (wcnt < spp * spp).any()                # border clipping is exercised


def test_psroi_bf16_channels_last_bit_exact_full_size():
    """Pipeline layout at full size: conv_new_1 map [B,38,63,256] bf16 channels-last, 300 rois per image,
    learned offsets; output [R,7,7,256] feeding fc_new_1 (weight columns permuted)."""
    ops, _ = _mods()
    rng = np.random.default_rng(12)
    B, C, H, W, P, R = 2, 256, 38, 63, 7, 600
    data = _bf16_round(np.maximum(rng.normal(0, 1, (B, C, H, W)), 0).astype(F))
    rois = _rois(rng, R, B, W, H)
    trans = rng.normal(0, 1.0, (R, 2, P, P)).astype(F)
    x = torch.as_tensor(data).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out = ops.deformable_psroi_pool(x, torch.as_tensor(rois).cuda(), torch.as_tensor(trans).cuda(), 0.0625, C, 1, P, P, 4,
                                    0.1, False, channels_last_out=True)
    assert out.permute(0, 2, 3, 1).is_contiguous()
    want, _ = deform.deformable_psroi_pooling(data, rois, trans, 0.0625, C, 1, P, P, 4, 0.1, False)
    assert np.array_equal(out.float().cpu().numpy(), _bf16_round(want))
    # offset branch of the graph: no_trans pooling (SYM_DCN_RELNMS:1073-1074)
    out0 = ops.deformable_psroi_pool(x, torch.as_tensor(rois).cuda(), None, 0.0625, C, 1, P, P, 4, 0.0, True, channels_last_out=True)
    want0, _ = deform.deformable_psroi_pooling(data, rois, None, 0.0625, C, 1, P, P, 4, 0.0, True)
    assert np.array_equal(out0.float().cpu().numpy(), _bf16_round(want0))


def test_psroi_operator_surface():
    _, cxx = _mods()
    rng = np.random.default_rng(13)
    data = rng.normal(0, 1, (1, 8, 12, 15)).astype(F)
    rois = _rois(rng, 9, 1, 15, 12)
    trans = rng.normal(0, 1, (9, 2, 7, 7)).astype(F)
    t = lambda a: torch.as_tensor(a).cuda()
    out = cxx.contrib.DeformablePSROIPooling(data=t(data), rois=t(rois), trans=t(trans), group_size=1, pooled_size=7,
                                             sample_per_part=4, no_trans=False, part_size=7, output_dim=8, spatial_scale=0.0625,
                                             trans_std=0.1)
    want, _ = deform.deformable_psroi_pooling(data, rois, trans, 0.0625, 8, 1, 7, 7, 4, 0.1, False)
    assert np.array_equal(out.cpu().numpy(), want)
    prop = cxx.DeformablePSROIPoolingProp(spatial_scale=0.0625, output_dim=8, group_size=1, pooled_size=7, no_trans='True')
    assert prop.ListArguments() == ['data', 'rois'] and prop.ListOutputs() == ['output', 'top_count']
    with pytest.raises(ValueError):
        prop.InferShape([(1, 8, 12, 15), (9, 4)])


def test_dcn_detector_stagewise_fp32_and_bf16():
    """Config-4 graph (deformable res5 + deformable PSROI pooling), teacher forced: fp32 backbone vs the
    torch-CPU/numpy restatement, pooling stage on the GPU's own feature map and rois, then the bf16 MFMA path
    against the fp32 path."""
    import relnet_amd  # noqa: F401
    from relnet_amd import backbone, detector, ops
    from oracle import network as ON
    H, W = 160, 224
    p = backbone.init_params(seed=11, dcn_offset_std=0.02)
    g = torch.Generator().manual_seed(12)
    p['conv_new_1_bias'] = torch.rand(256, generator=g) * 0.1 + 0.05
    p['offset_weight'] = torch.randn(98, 256 * 49, generator=g) * 0.05      # |trans| up to a few units
    data = torch.randn(1, 3, H, W, generator=g)
    im_info = torch.tensor([[H, W, 1.0]])
    cfg = detector.Config(); cfg.rpn_post_nms_top_n = 64; cfg.dcn = True
    det = detector.Detector(p, dtype=torch.float32, im_hw=(H, W), cfg=cfg)
    f = det.backbone.forward(data.cuda())
    with torch.no_grad():
        c4, c5 = ON.backbone(data, p, dcn=True)
        c5_plain = ON.res5(c4, p, dcn=False)
    assert (c5 - c5_plain).abs().max() > 1e-2 * c5.abs().max()          # the offsets really deform the sampling
    for got, want in ((f['conv4'], c4), (f['conv5'], c5)):
        err = (got.float().cpu() - want).abs().max().item() / want.abs().max().item()
        assert err < 3e-4, err
    out = det.forward(data.cuda(), im_info.cuda())
    rois = out['rois'][0].cpu().numpy()
    feat = f['conv_new_1_relu'].float().cpu().numpy()
    pn = {k: v.numpy() for k, v in p.items()}
    pooled_o, trans_o = ON.dcn_pool(feat, rois, pn)
    # device: same two poolings + FC on the same feature map
    r5 = out['rois'].view(-1, 5)
    t0 = ops.deformable_psroi_pool(f['conv_new_1_relu'], r5, None, 0.0625, 256, 1, 7, 7, 4, 0.0, True, channels_last_out=True)
    trans = ops.gemm_nt(t0.permute(0, 2, 3, 1).reshape(r5.shape[0], -1), det.w_offset, det.b_offset).view(-1, 2, 7, 7)
    assert np.abs(trans.cpu().numpy() - trans_o).max() <= 1e-4 * max(np.abs(trans_o).max(), 1.0)
    assert np.abs(trans_o).max() > 0.5
    pooled = ops.deformable_psroi_pool(f['conv_new_1_relu'], r5, torch.as_tensor(trans_o).cuda(), 0.0625, 256, 1, 7, 7, 4, 0.1, False)
    assert np.array_equal(pooled.cpu().numpy(), pooled_o)                # teacher-forced offsets -> bit exact
    assert torch.isfinite(out['cls_score']).all() and out['num_detections'].shape == (1,)
    # bf16 throughput path, batch of 2
    det16 = detector.Detector(p, dtype=torch.bfloat16, im_hw=(H, W), cfg=cfg)
    data2 = torch.cat([data, torch.randn(1, 3, H, W, generator=g)]).cuda()
    f16 = det16.backbone.forward(data2)
    err = (f16['conv5'][0].float() - f['conv5'][0]).abs().max().item() / f['conv5'].abs().max().item()
    assert err < 6e-2, err
    o16 = det16.forward(data2, torch.tensor([[H, W, 1.0]] * 2).cuda())
    assert torch.isfinite(o16['cls_score']).all() and o16['rois'].shape == (2, 64, 5)


# ------------------------------------------------------------------------------------------------------------------
# backward (DeformableConvolutionOp::Backward, DeformablePSROIPoolBackwardAcc) vs float64 autograd of the restated forward
# ------------------------------------------------------------------------------------------------------------------
def _relerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize('dtype,C,dg', [('f32', 64, 4), ('bf16', 64, 4), ('f32', 256, 2), ('bf16', 256, 2), ('bf16', 256, 4)])
def test_deformable_convolution_backward(dtype, C, dg):
    """C = 256, dg = 2: 128 channels per deformable group -> the in-wave reduction path of the offset gradient (res5 shape).
    bf16 with >= 64 channels per group runs the round-6 pair of kernels (data gradient GATHERED per feature cell from the pairs within one
    cell of their undeformed tap position, atomics for the others): offsets of sigma 1.5 put about half of the pairs on each path."""
    ops, _ = _mods()
    from oracle import deform_torch as DT
    rng = np.random.default_rng(21)
    B, H, W, Co, k, pad, dil = 2, 11, 13, 64, 3, 2, 2
    if dg == 4 and C == 256:
        H, W = 19, 37
    rnd = _bf16_round if dtype == 'bf16' else (lambda a: a)
    data = rnd(rng.normal(0, 1, (B, C, H, W)).astype(F))
    off = rng.normal(0, 1.5, (B, 2 * k * k * dg, H, W)).astype(F)
    wgt = rnd(rng.normal(0, 0.05, (Co, C, k, k)).astype(F))
    dy = rnd(rng.normal(0, 1, (B, Co, H, W)).astype(F))
    td = torch.tensor(data, dtype=torch.float64, requires_grad=True)
    to = torch.tensor(off, dtype=torch.float64, requires_grad=True)
    tw = torch.tensor(wgt, dtype=torch.float64, requires_grad=True)
    y = DT.deformable_convolution(td, to, tw, (k, k), (1, 1), (dil, dil), (pad, pad), dg)
    (y * torch.as_tensor(dy).double()).sum().backward()
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    x = torch.as_tensor(data).cuda().to(tdt).contiguous(memory_format=torch.channels_last)
    o = torch.as_tensor(off).cuda()
    g = torch.as_tensor(dy).cuda().to(tdt).contiguous(memory_format=torch.channels_last)
    wp = ops.pack_conv_weight(torch.as_tensor(wgt), dtype=tdt)
    gd, go, gw = ops.deformable_conv_bwd(x, o, wp, g, 3, 1, dil, pad, dg)
    tol = 2e-4 if dtype == 'f32' else 2e-2
    assert _relerr(gd.permute(0, 3, 1, 2).cpu().numpy(), td.grad.numpy()) <= tol
    assert _relerr(go.permute(0, 3, 1, 2).cpu().numpy(), to.grad.numpy()) <= tol
    want_w = tw.grad.permute(0, 2, 3, 1).reshape(Co, -1).numpy()
    assert _relerr(gw.cpu().numpy(), want_w) <= (2e-4 if dtype == 'f32' else 5e-3)


@pytest.mark.parametrize('sigma', [0.3, 1.5, 40.0])
def test_col2im_gather_form_equals_the_atomic_scatter(sigma):
    """relnet_deformable_col2im on a res5-sized layer (8 x 38 x 63 x 512, 4 deformable groups, dilation 2): the gather + offset kernels
    (mode 0: window radius 3; modes 11 / 12: radius 1 / 2) against the one-kernel atomic scatter (mode 1) and against every pair treated as
    FAR (mode 2).  sigma 0.3: every pair is NEAR (no data-gradient atomics at all); 1.5: a mix; 40: nearly all FAR or outside the map.  Same products, different fp32
    summation order."""
    ops, _ = _mods()
    from relnet_amd import lib
    L = lib.load()
    g_ = torch.Generator().manual_seed(int(sigma * 10))
    B, C, H, W, Co, dg = 8, 512, 38, 63, 512, 4
    x = torch.randn(B, C, H, W, generator=g_).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    o = (torch.randn(B, 18 * dg, H, W, generator=g_) * sigma).cuda()
    dy = torch.randn(B, Co, H, W, generator=g_).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wp = (torch.randn(Co, 9 * C, generator=g_) * 0.02).cuda().to(torch.bfloat16)
    res = {}
    try:
        for mode in (1, 0, 2, 11, 12):
            L.relnet_deformable_col2im_debug(mode)
            gd, go, _ = ops.deformable_conv_bwd(x, o, wp, dy, 3, 1, 2, 2, dg)
            torch.cuda.synchronize()
            res[mode] = (gd.clone(), go.clone())
    finally:
        L.relnet_deformable_col2im_debug(0)
    for mode in (0, 2, 11, 12):
        for what, a, b_ in (('grad_data', res[mode][0], res[1][0]), ('grad_offset', res[mode][1], res[1][1])):
            scale = b_.abs().max().item()
            assert scale > 0
            err = (a - b_).abs().max().item() / scale
            assert err < 2e-5, (sigma, mode, what, err)


@pytest.mark.parametrize('B,with_trans,chlast', [(8, True, True), (8, False, True), (2, True, False)])
def test_psroi_backward_four_channels_per_thread_equals_one(B, with_trans, chlast):
    """relnet_deformable_psroi_pool_bwd on the training step's shapes (features 38 x 63 x 256, 308 rois per image, 7 x 7 bins, 4 x 4 samples,
    class-agnostic offsets): the four-channels-per-thread kernel (round 6: roi geometry / offsets / per-axis cell sums built once per four
    channels) against the one-channel kernel (relnet_deformable_psroi_pool_bwd_debug(1)), channels-last bf16 and NCHW fp32 operands."""
    ops, _ = _mods()
    from relnet_amd import lib
    L = lib.load()
    g_ = torch.Generator().manual_seed(7 * B + int(with_trans))
    C, H, W, Rpi = 256, 38, 63, 308
    dt = torch.bfloat16 if chlast else torch.float32
    feat = torch.randn(B, H, W, C, generator=g_).to(dt).cuda().permute(0, 3, 1, 2)
    if not chlast:
        feat = feat.contiguous()
    rois = []
    for b in range(B):
        x1 = torch.rand(Rpi, generator=g_) * 700; y1 = torch.rand(Rpi, generator=g_) * 400
        w_ = 30 + torch.rand(Rpi, generator=g_) * 500; h_ = 30 + torch.rand(Rpi, generator=g_) * 300
        rois.append(torch.stack([torch.full((Rpi,), float(b)), x1, y1, (x1 + w_).clamp(max=999), (y1 + h_).clamp(max=599)], 1))
    rois = torch.cat(rois, 0).contiguous().cuda()
    R = rois.shape[0]
    trans = (torch.randn(R, 2, 7, 7, generator=g_) * 2.0).cuda() if with_trans else None
    gout = torch.randn(R, 7, 7, C, generator=g_).to(dt).cuda().permute(0, 3, 1, 2)
    if not chlast:
        gout = gout.contiguous()
    res = {}
    try:
        for mode in (1, 0):
            L.relnet_deformable_psroi_pool_bwd_debug(mode)
            gd, gt = ops.deformable_psroi_pool_bwd(gout, feat, rois, trans, 0.0625, C, 1, 7, 7, 4, 0.1 if with_trans else 0.0, not with_trans)
            torch.cuda.synchronize()
            res[mode] = (gd.clone(), None if gt is None else gt.clone())
    finally:
        L.relnet_deformable_psroi_pool_bwd_debug(0)
    scale = res[1][0].abs().max().item()
    assert scale > 0
    err = (res[0][0] - res[1][0]).abs().max().item() / scale
    assert err < 2e-5, ('grad_data', err)
    if with_trans:
        ts = res[1][1].abs().max().item()
        assert ts > 0 and (res[0][1] - res[1][1]).abs().max().item() / ts < 2e-5


@pytest.mark.parametrize("case", PSROI_CASES)
def test_psroi_backward(case):
    ops, _ = _mods()
    from oracle import deform_torch as DT
    group, no_trans, ncls, od, spp, tstd = case
    rng = np.random.default_rng(22)
    B, H, W, P, R = 2, 14, 17, 7, 12
    data = rng.normal(0, 1, (B, od * group * group, H, W)).astype(F)
    rois = _rois(rng, R, B, W, H)
    trans = None if no_trans else rng.normal(0, 1.0, (R, 2 * ncls, P, P)).astype(F)
    gout = rng.normal(0, 1, (R, od, P, P)).astype(F)
    td = torch.tensor(data, dtype=torch.float64, requires_grad=True)
    tt = None if no_trans else torch.tensor(trans, dtype=torch.float64, requires_grad=True)
    y = DT.deformable_psroi_pooling(td, rois, tt, 0.0625, od, group, P, P, spp, tstd, no_trans)
    (y * torch.as_tensor(gout).double()).sum().backward()
    gd, gt = ops.deformable_psroi_pool_bwd(torch.as_tensor(gout).cuda(), torch.as_tensor(data).cuda(), torch.as_tensor(rois).cuda(),
                                           None if no_trans else torch.as_tensor(trans).cuda(), 0.0625, od, group, P, P, spp, tstd, no_trans)
    assert _relerr(gd.cpu().numpy(), td.grad.numpy()) <= 2e-5
    if not no_trans:
        assert _relerr(gt.cpu().numpy(), tt.grad.numpy()) <= 2e-4
    else:
        assert gt is None


def test_operator_backward_protocol():
    """`DeformableConvolutionOp::Backward` / `DeformablePSROIPoolingOp::Backward` through the mirrored OperatorProperty /
    Operator classes (deformable_convolution-inl.h:145-237, deformable_psroi_pooling-inl.h:97-140): NCHW fp32 blobs,
    per-input OpReqType (write / add / null), bias gradient, vs float64 autograd of the restated forward."""
    _, cxx = _mods()
    from oracle import deform_torch as DT
    rng = np.random.default_rng(31)
    B, C, H, W, Co, k, pad, dil, dg = 2, 64, 9, 11, 64, 3, 2, 2, 4
    data = rng.normal(0, 1, (B, C, H, W)).astype(F)
    off = rng.normal(0, 1.0, (B, 2 * k * k * dg, H, W)).astype(F)
    wgt = rng.normal(0, 0.05, (Co, C, k, k)).astype(F)
    bias = rng.normal(0, 1, (Co,)).astype(F)
    dy = rng.normal(0, 1, (B, Co, H, W)).astype(F)
    td, to, tw = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (data, off, wgt))
    y = DT.deformable_convolution(td, to, tw, (k, k), (1, 1), (dil, dil), (pad, pad), dg)
    (y * torch.as_tensor(dy).double()).sum().backward()
    prop = cxx.DeformableConvolutionProp(kernel=(k, k), num_filter=Co, pad=(pad, pad), dilate=(dil, dil), num_deformable_group=dg)
    assert prop.TypeString() == '_contrib_DeformableConvolution' and prop.ListArguments() == ['data', 'offset', 'weight', 'bias']
    assert prop.DeclareBackwardDependency(['og'], ['d', 'o', 'w', 'b'], ['out']) == ['og', 'd', 'o', 'w']
    op = prop.CreateOperatorEx()
    d = lambda a: torch.as_tensor(a).cuda()
    in_data = [d(data), d(off), d(wgt), d(bias)]
    prior = torch.ones(Co, C, k, k, device='cuda')
    in_grad = [torch.full((B, C, H, W), 7.0, device='cuda'), torch.empty(B, 2 * k * k * dg, H, W, device='cuda'), prior.clone(),
               torch.full((Co,), 5.0, device='cuda')]
    op.Backward(None, [d(dy)], in_data, [None], ['write', 'write', 'add', 'null'], in_grad)
    assert _relerr(in_grad[0].cpu().numpy(), td.grad.numpy()) <= 3e-4
    assert _relerr(in_grad[1].cpu().numpy(), to.grad.numpy()) <= 3e-4
    assert _relerr((in_grad[2] - prior).cpu().numpy(), tw.grad.numpy()) <= 3e-4          # kAddTo accumulated onto the ones
    assert bool((in_grad[3] == 5.0).all())                                               # kNullOp left alone
    op.Backward(None, [d(dy)], in_data, [None], ['null', 'null', 'null', 'write'], in_grad)
    assert _relerr(in_grad[3].cpu().numpy(), dy.sum((0, 2, 3))) <= 1e-5
    # pooling operator
    P, R, od = 7, 10, 16
    feat = rng.normal(0, 1, (B, od, 14, 17)).astype(F)
    rois = _rois(rng, R, B, 17, 14)
    trans = rng.normal(0, 1.0, (R, 2, P, P)).astype(F)
    gout = rng.normal(0, 1, (R, od, P, P)).astype(F)
    tf = torch.tensor(feat, dtype=torch.float64, requires_grad=True)
    tt = torch.tensor(trans, dtype=torch.float64, requires_grad=True)
    yp = DT.deformable_psroi_pooling(tf, rois, tt, 0.0625, od, 1, P, P, 4, 0.1, False)
    (yp * torch.as_tensor(gout).double()).sum().backward()
    pprop = cxx.DeformablePSROIPoolingProp(spatial_scale=0.0625, output_dim=od, group_size=1, pooled_size=P, part_size=P,
                                           sample_per_part=4, trans_std=0.1, no_trans=False)
    assert pprop.DeclareBackwardDependency(['og'], ['d', 'r', 't'], ['out', 'cnt']) == ['og', 'd', 'r', 't', 'cnt']
    pop = pprop.CreateOperatorEx()
    out = [torch.empty(R, od, P, P, device='cuda'), torch.empty(R, od, P, P, device='cuda')]
    pop.Forward(None, [d(feat), d(rois), d(trans)], ['write', 'write'], out)
    gin = [torch.empty(B, od, 14, 17, device='cuda'), torch.zeros(R, 5, device='cuda'), torch.empty(R, 2, P, P, device='cuda')]
    pop.Backward(None, [d(gout)], [d(feat), d(rois), d(trans)], out, ['write', 'null', 'write'], gin)
    assert _relerr(gin[0].cpu().numpy(), tf.grad.numpy()) <= 2e-5 and _relerr(gin[2].cpu().numpy(), tt.grad.numpy()) <= 2e-4
    assert bool((gin[1] == 0).all())
    with pytest.raises(ValueError):
        pop.Backward(None, [d(gout)], [d(feat), d(rois), d(trans)], out, ['inplace', 'null', 'write'], gin)
