"""Pixel-wise residual-boundary kernel (csrc/bottleneck.hip): relu(W3 mid2 + b3 + x) and the next unit's reduce in one
launch, against the two convolution launches it replaces and a float64 evaluation of the same bf16 operands."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mid', [64, 128, 256])
@pytest.mark.parametrize('shape', [(1, 5, 7), (2, 38, 63), (3, 150, 250)])
def test_bottleneck_chain_vs_two_convolutions(shape, mid):
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    B, H, W = shape
    if mid == 128 and H == 150:
        H, W = 75, 125                     # the res3 map of a 600 x 1000 image
    if mid == 256 and H == 150:
        B, H, W = 20, 38, 63               # res4 maps, 375 sets of four tiles: one or two per workgroup (256 CUs)
    cout = 4 * mid
    g = torch.Generator().manual_seed(B * 1000 + H + mid)
    bf = torch.bfloat16
    m2 = torch.relu(torch.randn(B, H, W, mid, generator=g)).to(bf).cuda()
    x = torch.relu(torch.randn(B, H, W, cout, generator=g)).to(bf).cuda()
    w3 = (torch.randn(cout, mid, generator=g) * 0.1).to(bf).cuda()
    w1 = (torch.randn(mid, cout, generator=g) * 0.05).to(bf).cuda()
    b3 = (torch.randn(cout, generator=g) * 0.1).cuda()
    b1 = (torch.randn(mid, generator=g) * 0.1).cuda()
    xn, m1 = ops.bottleneck_chain(m2, x, ops.pack_w_frag(w3), ops.pack_chain_w1(w1), b3, b1)
    xe, none = ops.bottleneck_chain(m2, x, ops.pack_w_frag(w3), None, b3, None)        # expand-only form: same x_next bits
    assert none is None and torch.equal(xe, xn)
    xi = x.clone()                                                                      # in place over the shortcut operand: same bits again
    xn_i, m1_i = ops.bottleneck_chain(m2, xi, ops.pack_w_frag(w3), ops.pack_chain_w1(w1), b3, b1, inplace=True)
    assert xn_i.data_ptr() == xi.data_ptr() and torch.equal(xn_i, xn) and torch.equal(m1_i, m1)
    # (a) the two launches of the unfused path
    xn_ref = ops.conv2d_nhwc(m2, w3, b3, relu=True, resid=x)
    m1_ref = ops.conv2d_nhwc(xn_ref, w1, b1, relu=True)
    # (b) float64 on the same bf16 operands, with the bf16 rounding of x_next that both paths apply
    xn64 = torch.relu(m2.double() @ w3.double().t() + b3.double() + x.double())
    m164 = torch.relu(xn.double() @ w1.double().t() + b1.double())          # from THIS kernel's x_next: isolates the 2nd product
    ulp = lambda t: torch.maximum(t.abs(), torch.tensor(2.0 ** -126, dtype=torch.float64, device=t.device)) * 2.0 ** -7   # >= the bf16 spacing at |t|
    # correctly rounded up to the fp32 accumulation noise (~1e-6 of the summed magnitudes) that can move a near-tie or a
    # value next to the ReLU threshold
    ex = (xn.double() - xn64).abs() - 0.5 * ulp(xn64)
    em = (m1.double() - m164).abs() - 0.5 * ulp(m164)
    print('shape %s: worst excess over half a bf16 step: x_next %.2e, mid1 %.2e' % (shape, ex.max().item(), em.max().item()))
    assert ex.max().item() <= 2e-5 and em.max().item() <= 2e-5
    # the two-launch path differs by at most one bf16 ulp, and only rarely (different fp32 summation order)
    dx = (xn.float() - xn_ref.float()).abs().double()
    dm = (m1.float() - m1_ref.float()).abs().double()
    fx, fm = (dx > 0).double().mean().item(), (dm > 0).double().mean().item()
    print('  vs two launches: x_next differs in %.3f%% of elements (max %.1f steps), mid1 in %.3f%% (max %.1f steps)'
          % (100 * fx, (dx / ulp(xn64)).max().item(), 100 * fm, (dm / (ulp(m164) + 1e-6)).max().item()))
    assert (dx <= ulp(xn64) * 1.01 + 1e-6).all() and fx < 0.02     # (+ fp32 summation-order noise next to the ReLU threshold)
    assert (dm <= 4 * ulp(m164) + 1e-2).all() and fm < 0.05        # (a one-step difference of x_next times |W1|)

@pytest.mark.parametrize('shape', [(1, 5, 7), (2, 38, 63), (3, 150, 250)])
@pytest.mark.parametrize('reduce', [True, False])
def test_bottleneck_chain_with_projection_shortcut(shape, reduce):
    """relnet_bottleneck_chain_proj (res2a: the 1x1 projection shortcut as four more k-steps of the expand product) against a
    float64 evaluation of the same bf16 operands: x_next = relu(W3 mid2 + Wp x_in + b3 + bp) is correctly rounded (no bf16
    rounding of the shortcut on the way), mid1' as in the plain chain kernel; and against the three convolution launches it
    replaces (which DO round the projection to bf16: within one bf16 step)."""
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    B, H, W = shape
    mid, cout = 64, 256
    g = torch.Generator().manual_seed(B * 1000 + H + 7)
    bf = torch.bfloat16
    m2 = torch.relu(torch.randn(B, H, W, mid, generator=g)).to(bf).cuda()
    xin = torch.relu(torch.randn(B, H, W, mid, generator=g)).to(bf).cuda()
    w3 = (torch.randn(cout, mid, generator=g) * 0.1).to(bf).cuda()
    wp = (torch.randn(cout, mid, generator=g) * 0.1).to(bf).cuda()
    w1 = (torch.randn(mid, cout, generator=g) * 0.05).to(bf).cuda()
    b3, bp = (torch.randn(cout, generator=g) * 0.1).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()
    b1 = (torch.randn(mid, generator=g) * 0.1).cuda()
    xn, m1 = ops.bottleneck_chain_proj(m2, xin, ops.pack_w_frag(w3), ops.pack_w_frag(wp), ops.pack_chain_w1(w1) if reduce else None,
                                       (b3 + bp).contiguous(), b1 if reduce else None)
    xn64 = torch.relu(m2.double() @ w3.double().t() + xin.double() @ wp.double().t() + (b3 + bp).double())
    ulp = lambda t: torch.maximum(t.abs(), torch.tensor(2.0 ** -126, dtype=torch.float64, device=t.device)) * 2.0 ** -7
    ex = (xn.double() - xn64).abs() - 0.5 * ulp(xn64)
    assert ex.max().item() <= 4e-5, ex.max().item()
    if reduce:
        m164 = torch.relu(xn.double() @ w1.double().t() + b1.double())
        em = (m1.double() - m164).abs() - 0.5 * ulp(m164)
        assert em.max().item() <= 2e-5, em.max().item()
    else:
        assert m1 is None
    sc = ops.conv2d_nhwc(xin, wp, bp)                                   # the launches of the unfused path
    ref = ops.conv2d_nhwc(m2, w3, b3, relu=True, resid=sc)
    assert ((xn.float() - ref.float()).abs().double() <= 1.01 * ulp(xn64) + 0.5 * ulp(sc.double()) + 1e-6).all()


@pytest.mark.parametrize('mid,shape', [(256, (2, 38, 63)), (512, (2, 38, 63)), (256, (1, 5, 7)), (512, (3, 19, 32))])
def test_expand_only_chain_wide(mid, shape):
    """res4 / res5 expand + shortcut + ReLU on the chain kernel (weights through the LDS ring, k split in two for mid = 512):
    float64 on the same bf16 operands, and the implicit-GEMM launch it replaces."""
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(mid + H)
    bf = torch.bfloat16
    m2 = torch.relu(torch.randn(B, H, W, mid, generator=g)).to(bf).cuda()
    x = torch.relu(torch.randn(B, H, W, 4 * mid, generator=g)).to(bf).cuda()
    w3 = (torch.randn(4 * mid, mid, generator=g) * 0.05).to(bf).cuda()
    b3 = (torch.randn(4 * mid, generator=g) * 0.1).cuda()
    xn, none = ops.bottleneck_chain(m2, x, ops.pack_w_frag(w3), None, b3, None)
    assert none is None
    ref64 = torch.relu(m2.double() @ w3.double().t() + b3.double() + x.double())
    step = torch.maximum(ref64.abs(), torch.tensor(2.0 ** -126, dtype=torch.float64, device='cuda')) * 2.0 ** -7
    ex = ((xn.double() - ref64).abs() - 0.5 * step).max().item()
    y2 = ops.conv2d_nhwc(m2, w3, b3, relu=True, resid=x)
    d = (xn.float() - y2.float()).abs().double()
    print('mid %d %s: excess over half a bf16 step %.2e; differs from the implicit-GEMM launch in %.4f%% of elements' % (mid, shape, ex, 100 * (d > 0).double().mean().item()))
    assert ex <= 5e-5 and (d <= step * 1.01 + 1e-5).all()


def test_backbone_with_and_without_chain_kernel():
    """The detector trunk with the res2 boundaries fused equals the unfused trunk up to bf16 rounding of a few elements."""
    import relnet_amd  # noqa: F401
    from relnet_amd import backbone
    p = backbone.init_params(seed=5)
    data = torch.randn(2, 3, 608, 1008, generator=torch.Generator().manual_seed(1)).cuda() * 50     # res2 maps above ops.chain_worthwhile's threshold
    a = backbone.Backbone(p, dtype=torch.bfloat16, chain=True)
    b = backbone.Backbone(p, dtype=torch.bfloat16, chain=False)
    assert [k for k in sorted(a.chain) if k[0] in '23'] == ['2a', '2b', '2c', '3a', '3b1', '3b2', '3b3'] and a.chain['2c'][1] is None
    assert sum(k[0] == '4' for k in a.chain) == 23 and all(a.chain[k][1] is None for k in a.chain if k[0] == '5' or k == '4b22') and not b.chain
    assert all(a.chain[k][1] is not None for k in a.chain if k[0] == '4' and k != '4b22')        # res4: expand + the next unit's reduce (r04)
    assert sum(k[0] == '5' for k in a.chain) == 3
    assert sorted(a.chain_proj) == ['2a'] and not b.chain_proj
    ref = backbone.Backbone(p, dtype=torch.float32).forward(data)            # float32 trunk: the yardstick for both
    fa, fb = a.forward(data), b.forward(data)
    assert a.last_chain_units[:3] == ['2a', '2b', '2c']
    for k in ('conv4', 'conv5', 'rpn_cls_score', 'rpn_bbox_pred'):
        r = ref[k].float()
        ea, eb = ((fa[k].float() - r).norm() / r.norm()).item(), ((fb[k].float() - r).norm() / r.norm()).item()
        # measured (r04): 0.0086 - 0.0104 for both trunks; the fused one is never the worse of the two by more than rounding noise
        assert ea <= 1.5e-2 and eb <= 1.5e-2 and ea <= eb * 1.05, (k, ea, eb)
        d = (fa[k].float() - fb[k].float()).abs().max().item()
        s = fb[k].float().abs().max().item()
        assert d <= 3e-2 * s, (k, d, s)                # (the res2a projection is not rounded to bf16 in the fused trunk: 0.021 measured)


@pytest.mark.parametrize('C', [64])
@pytest.mark.parametrize('shape', [(1, 5, 7), (2, 38, 63), (2, 150, 250), (1, 8, 32), (3, 17, 65)])
def test_conv3x3_halo_kernels(shape, C):
    """Halo-resident 3x3 kernel against float64 on the same bf16 operands (correctly rounded up to fp32 accumulation noise)
    and against the implicit-GEMM launch it replaces; ragged tiles in both directions, image borders = zero padding."""
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    B, H, W = shape
    if C == 256 and H == 150:
        H, W = 75, 125
    g = torch.Generator().manual_seed(7 * B + H + C)
    bf = torch.bfloat16
    x = torch.relu(torch.randn(B, H, W, C, generator=g)).to(bf).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) * (0.05 if C == 64 else 0.02)).to(bf)
    b = (torch.randn(C, generator=g) * 0.1).cuda()
    wp = ops.pack_conv_weight(w, bf, 'cuda')
    for relu in (True, False):
        y = ops.conv3x3_halo(x, ops.pack_w_frag(wp, panel_only=False), b, relu=relu)
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().cuda(), b.double(), padding=1).permute(0, 2, 3, 1)
        ref = torch.relu(ref) if relu else ref
        step = torch.maximum(ref.abs(), torch.tensor(2.0 ** -126, dtype=torch.float64, device='cuda')) * 2.0 ** -7
        ex = ((y.double() - ref).abs() - 0.5 * step).max().item()
        y2 = ops.conv2d_nhwc(x, wp, b, ksize=3, pad=1, relu=relu)
        d = (y.float() - y2.float()).abs().double()
        print('shape %s relu=%d: excess over half a bf16 step %.2e; differs from the implicit-GEMM launch in %.4f%% of elements'
              % (shape, relu, ex, 100 * (d > 0).double().mean().item()))
        assert ex <= 5e-5
        assert (d <= step * 1.01 + 1e-5).all()          # (C = 256 sums the channel halves in a different order than the implicit GEMM)
