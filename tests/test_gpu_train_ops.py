"""GPU checks of the training building blocks (train_ops.py) against torch-CPU float32 autograd on the SAME bf16-rounded
operands.  bf16 MFMA products with fp32 accumulation: gradients within 1.5e-2 of their max-abs (outputs that are
themselves rounded to bf16), weight gradients (fp32 outputs) within 3e-3."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def _mods():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops, train_ops
    return ops, train_ops


def _bf(t):
    return t.to(torch.bfloat16).float()


def _rel(a, b):
    return (a.float().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize('stride', [1, 2])
def test_conv1x1_backward(stride):
    ops, T = _mods()
    g = torch.Generator().manual_seed(1)
    B, H, W, Cin, Cout = 2, 38, 63, 256, 128
    x = _bf(torch.randn(B, Cin, H, W, generator=g)).requires_grad_(True)
    w = _bf(torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride)
    dy = _bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda().to(torch.bfloat16)
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda().to(torch.bfloat16)
    wp = ops.pack_conv_weight(w.detach())
    dx, dw = T.conv1x1_bwd(xd, wp, dyd, stride=stride)
    assert _rel(dx.permute(0, 3, 1, 2), x.grad) <= 1.5e-2
    assert _rel(dw.reshape(Cout, Cin, 1, 1), w.grad) <= 3e-3
    if stride == 1:       # shortcut gradient accumulated in the GEMM epilogue
        add = _bf(torch.randn(B, H, W, Cin, generator=g))
        dx2, _ = T.conv1x1_bwd(xd, wp, dyd, dx_add=add.cuda().to(torch.bfloat16))
        assert _rel(dx2.permute(0, 3, 1, 2), x.grad + add.permute(0, 3, 1, 2)) <= 1.5e-2


@pytest.mark.parametrize('dil', [1, 2])
def test_conv3x3_backward(dil):
    ops, T = _mods()
    g = torch.Generator().manual_seed(2)
    B, H, W, Cin, Cout = 2, 19, 31, 128, 64
    x = _bf(torch.randn(B, Cin, H, W, generator=g)).requires_grad_(True)
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03).requires_grad_(True)
    y = F.conv2d(x, w, padding=dil, dilation=dil)
    dy = _bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda().to(torch.bfloat16)
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda().to(torch.bfloat16)
    dx, dw = T.conv3x3_bwd(xd, T.pack_conv_dgrad_weight(w.detach()), dyd, dil=dil)
    assert _rel(dx.permute(0, 3, 1, 2), x.grad) <= 1.5e-2
    want_dw = w.grad.permute(0, 2, 3, 1).reshape(Cout, -1)              # pack_conv_weight order
    assert _rel(dw, want_dw) <= 3e-3


def test_linear_backward_and_wgrad_split_k():
    ops, T = _mods()
    g = torch.Generator().manual_seed(3)
    P, K, N = 5000, 1024, 96                                            # long contraction -> several K splits
    x = _bf(torch.randn(P, K, generator=g)).requires_grad_(True)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).requires_grad_(True)
    b = torch.zeros(N, requires_grad=True)
    y = x @ w.t() + b
    dy = _bf(torch.randn(P, N, generator=g))
    y.backward(dy)
    dx, dw, db = T.linear_bwd(x.detach().cuda().to(torch.bfloat16), w.detach().cuda().to(torch.bfloat16),
                              dy.cuda().to(torch.bfloat16))
    assert _rel(dx, x.grad) <= 1.5e-2 and _rel(dw, w.grad) <= 3e-3 and _rel(db, b.grad) <= 1e-5
    assert T._splits_for(N, K, P) > 1


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_relu_bwd_and_sgd(dtype):
    ops, T = _mods()
    g = torch.Generator().manual_seed(4)
    n = 3 * 1000 + 5                                                    # exercises the scalar tail
    y = torch.relu(torch.randn(n, generator=g)).to(dtype).cuda()
    dy = torch.randn(n, generator=g).to(dtype).cuda()
    add = torch.randn(n, generator=g).to(dtype).cuda()
    want = (dy.float() * (y.float() > 0)).to(dtype)
    assert torch.equal(T.relu_bwd(dy, y), want)
    assert torch.equal(T.relu_bwd(dy, y, add), (dy.float() * (y.float() > 0) + add.float()).to(dtype))
    if dtype == torch.float32:
        w = torch.randn(n, generator=g).cuda(); m = torch.randn(n, generator=g).cuda() * 0.1
        gr = torch.randn(n, generator=g).cuda()
        w0, m0 = w.clone(), m.clone()
        wb = torch.empty(n, device='cuda', dtype=torch.bfloat16)
        T.sgd_update(w, m, gr, lr=0.0005, momentum=0.9, wd=0.0005, w_bf16=wb)
        m1 = 0.9 * m0 - 0.0005 * (gr + 0.0005 * w0)
        assert torch.allclose(m, m1, rtol=1e-5, atol=1e-7) and torch.allclose(w, w0 + m1, rtol=1e-5, atol=1e-6)
        assert torch.equal(wb, w.to(torch.bfloat16))
