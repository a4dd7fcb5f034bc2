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


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('hw', [(38, 63), (75, 125), (8, 8)])
def test_strided_scatter_is_the_adjoint_of_the_input_sampling(hw, dt):
    """relnet_strided_scatter: zeros + out[:, ::2, ::2] = low (+ the ReLU mask of the full-resolution map), bit for bit; and the two-branch
    stride-2 backward summed at the sampled resolution equals the sum of the two scattered gradients."""
    ops, T = _mods()
    g = torch.Generator().manual_seed(5)
    B, (H, W), C = 2, hw, 64
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    low = torch.randn(B, Ho, Wo, C, generator=g).cuda().to(dt)
    y = torch.relu(torch.randn(B, H, W, C, generator=g)).cuda().to(dt)
    ref = torch.zeros(B, H, W, C, device='cuda', dtype=dt)
    ref[:, ::2, ::2] = low
    out = T.strided_scatter(low, (B, H, W, C), 2)
    assert torch.equal(out, ref)
    outm = T.strided_scatter(low, (B, H, W, C), 2, mask=y)
    assert torch.equal(outm, ref * (y > 0))
    with pytest.raises(RuntimeError):
        T.strided_scatter(low[:, :-1].contiguous(), (B, H, W, C), 2)
    if dt == torch.bfloat16:
        Cout = 128
        x = torch.randn(B, H, W, C, generator=g).cuda().to(dt)
        w1 = (torch.randn(Cout, C, generator=g) * 0.05).cuda().to(dt)
        w2 = (torch.randn(Cout, C, generator=g) * 0.05).cuda().to(dt)
        dy1 = torch.randn(B, Ho, Wo, Cout, generator=g).cuda().to(dt)
        dy2 = torch.randn(B, Ho, Wo, Cout, generator=g).cuda().to(dt)
        a, _ = T.conv1x1_bwd(x, w1, dy1, stride=2)
        b, _ = T.conv1x1_bwd(x, w2, dy2, stride=2)
        la, _ = T.conv1x1_bwd(x, w1, dy1, stride=2, low_res=True)
        ls, _ = T.conv1x1_bwd(x, w2, dy2, stride=2, low_res=True, dx_add=la)
        s = T.strided_scatter(ls, (B, H, W, C), 2)
        assert _rel(s, (a.float() + b.float()).cpu()) <= 1e-2
        assert float(s[:, 1::2].abs().max()) == 0.0 and float(s[:, :, 1::2].abs().max()) == 0.0


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


def test_tr_probe_prints_the_transposed_read_map():
    """What ds_read_b64_tr_b16 returns for lane-linear addresses over value == index (csrc/wgrad.hip builds its MFMA fragments
    on it): lane l of a 16-lane group must receive column (l & 15) of the group's row-major [4][16] block."""
    import relnet_amd  # noqa: F401
    from relnet_amd import lib as L, ops
    out = torch.zeros(256, dtype=torch.int16, device='cuda')
    L.call('relnet_debug_tr_probe', out.data_ptr(), ops._stream())
    got = out.cpu().numpy().astype(np.int64).reshape(64, 4)
    lane = np.arange(64)[:, None]
    want = (lane >> 4) * 64 + (lane & 15) + 16 * np.arange(4)[None, :]
    print('tr probe lanes 0..3, 16, 17, 63:', got[[0, 1, 2, 3, 16, 17, 63]].tolist())
    assert np.array_equal(got, want), got.tolist()


@pytest.mark.parametrize('case', ['fc', 'fc_ragged', 'conv1x1_s2', 'conv3x3', 'conv3x3_dil2', 'rpn_out', 'wide'])
def test_wgrad_tn_matches_float64(case):
    """relnet_wgrad (dY^T X from the row-major operands: transposed LDS reads, implicit im2col, atomics over the pixel splits)
    against float64 on the bf16 inputs, incl. ragged pixel counts / channel counts, strided and dilated gathers, the folded
    BatchNorm row factor, accumulation into a non-zero buffer and a strided view of a wider buffer; plain-LDS-read mode too."""
    import torch.nn.functional as F
    import relnet_amd  # noqa: F401
    from relnet_amd import lib as L, ops
    g = torch.Generator().manual_seed(hash(case) % 1000)
    bf = torch.bfloat16
    rnd = lambda *s: torch.randn(*s, generator=g)
    scale = None
    if case in ('fc', 'fc_ragged', 'wide'):
        P, Cout, K = {'fc': (2464, 256, 384), 'fc_ragged': (1237, 89, 136), 'wide': (700, 1024, 1152)}[case]
        cpad = (Cout + 63) // 64 * 64
        dy = torch.zeros(P, cpad); dy[:, :Cout] = rnd(P, Cout)
        x = rnd(P, K)
        dyb, xb = dy.to(bf).cuda(), x.to(bf).cuda()
        want = dyb[:, :Cout].double().t().cpu() @ xb.double().cpu()
        conv = None
        xk = xb
    else:
        ks, stride, dil, Cin, Cout, B, H, W = {'conv1x1_s2': (1, 2, 1, 256, 128, 2, 37, 51), 'conv3x3': (3, 1, 1, 128, 128, 2, 19, 23),
                                               'conv3x3_dil2': (3, 1, 2, 64, 256, 3, 14, 17), 'rpn_out': (1, 1, 1, 512, 72, 2, 38, 63)}[case]
        pad = dil if ks == 3 else 0
        xb = rnd(B, H, W, Cin).to(bf).cuda()
        Ho, Wo = (H + 2 * pad - dil * (ks - 1) - 1) // stride + 1, (W + 2 * pad - dil * (ks - 1) - 1) // stride + 1
        dyb = rnd(B * Ho * Wo, Cout).to(bf).cuda()
        # float64 reference through autograd of conv2d
        w = torch.zeros(Cout, Cin, ks, ks, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(xb.double().cpu().permute(0, 3, 1, 2), w, None, stride=stride, padding=pad, dilation=dil)
        (y * dyb.double().cpu().view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)).sum().backward()
        want = w.grad.permute(0, 2, 3, 1).reshape(Cout, -1)              # pack_conv_weight order
        conv = None if (ks == 1 and stride == 1) else (ks, stride, dil, pad)
        xk = xb.view(-1, Cin) if conv is None else xb
        scale = (torch.rand(Cout, generator=g) + 0.5).cuda()
        want = want * (scale.double().cpu() ** 2).view(-1, 1)
    for plain in (0, 1):
        L.load().relnet_wgrad_debug_plain(plain)
        try:
            base = torch.randn(want.shape[0], want.shape[1] + 24, generator=g).cuda()          # accumulate into a view of a wider buffer
            out = base[:, 8:8 + want.shape[1]]
            before = out.clone()
            ops.wgrad_tn(dyb, xk, out=out, row_scale=scale, cout=want.shape[0], conv=conv)
            got = (out - before).double().cpu()
        finally:
            L.load().relnet_wgrad_debug_plain(0)
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err <= 2e-5, (case, plain, err)              # fp32 accumulation of exact bf16 products
        assert torch.equal(base[:, :8], base[:, :8]) and torch.isfinite(base).all()
    fresh = ops.wgrad_tn(dyb, xk, row_scale=scale, cout=want.shape[0], conv=conv)
    assert (fresh.double().cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    # the same layer inside a grouped launch with two other layers (stream-K shares cut across layer boundaries), and with
    # more / fewer persistent workgroups than work units
    other = [(torch.randn(900, 64, generator=g).to(bf).cuda(), torch.randn(900, 96, generator=g).to(bf).cuda()),
             (torch.randn(130, 320, generator=g).to(bf).cuda(), torch.randn(130, 264, generator=g).to(bf).cuda())]
    # share rule (round 6): 1 = stream-K shares + float atomics (the default), 2 = whole output tiles per workgroup + plain read-modify-write
    # (opt-in: measured slower, but one writer per element = bit-identical reruns)
    for tiles_mode, blocks in ((1, 0), (1, 7), (1, 1000), (2, 0), (2, 7), (2, 1000)):
        L.load().relnet_wgrad_tune(blocks, 0, 0)
        L.load().relnet_wgrad_debug_tiles(tiles_mode)
        runs = []
        try:
            for _ in range(2):
                q = ops.WgradQueue()
                o0 = torch.zeros(64, 96, device='cuda'); o1 = torch.zeros(320, 264, device='cuda')
                om = torch.zeros(want.shape, device='cuda')
                q.add(other[0][0], other[0][1], o0)
                q.add(dyb, xk, om, scale, want.shape[0], conv)
                q.add(other[1][0], other[1][1], o1)
                assert len(q) == 3
                q.flush()
                runs.append(om.clone())
        finally:
            L.load().relnet_wgrad_tune(0, 0, 0)
            L.load().relnet_wgrad_debug_tiles(0)
        if tiles_mode == 2:
            assert torch.equal(runs[0], runs[1]), (case, blocks)
        assert (om.double().cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item(), (case, tiles_mode, blocks)
        for o, (dy_, x_) in ((o0, other[0]), (o1, other[1])):
            w_ = dy_.double().t().cpu() @ x_.double().cpu()
            assert (o.double().cpu() - w_).abs().max().item() <= 2e-5 * w_.abs().max().item(), (case, tiles_mode, blocks)
    # two layers of one group accumulating into the SAME memory: the entry point must see the overlap and keep the atomics
    L.load().relnet_wgrad_debug_tiles(2)
    try:
        q = ops.WgradQueue()
        om = torch.zeros(want.shape, device='cuda')
        q.add(dyb, xk, om, scale, want.shape[0], conv)
        q.add(dyb, xk, om, scale, want.shape[0], conv)
        q.flush()
    finally:
        L.load().relnet_wgrad_debug_tiles(0)
    assert (om.double().cpu() - 2 * want).abs().max().item() <= 4e-5 * want.abs().max().item(), case


def test_weight_relayout_one_launch_for_all_layers():
    """relnet_weight_relayout (the data-gradient copies of every weight of a step in one grouped launch) against the torch
    definition: taps = 1 -> W^T zero padded to 64 output channels ([Cin, pad(Cout)], what linear_bwd / conv1x1_bwd take as w_t),
    taps = 9 -> the tap-flipped, (Cout, Cin)-transposed 3x3 filter [Cin, 9 * Cout] (what conv3x3_bwd convolves dy with).
    Ragged shapes: 72 / 89 output channels (rpn_out, cls_score | bbox_pred), Cin not a multiple of 64, many problems so that
    the tile -> problem binary search crosses every boundary; and a refresh after the source changed (views, not copies)."""
    ops, _ = _mods()
    torch.manual_seed(5)
    shapes = [(256, 1024, 1), (72, 512, 1), (89, 1024, 1), (256, 256, 9), (512, 512, 9), (1024, 256, 1), (64, 96, 1),
              (128, 128, 9), (2048, 512, 1), (24, 40, 9)]
    flat = torch.randn(sum(co * ci * t for co, ci, t in shapes), device='cuda').to(torch.bfloat16)
    rl = ops.WeightRelayout('cuda')
    views, off = [], 0
    for i, (co, ci, t) in enumerate(shapes):
        v = flat[off:off + co * ci * t].view(co, t * ci); off += co * ci * t
        views.append(v)
        rl.add('w%d' % i, v, taps=t, pad_co=1 if t == 9 else 64)
    rl.build()
    for rnd in range(2):
        if rnd == 1:
            flat.copy_(torch.randn_like(flat.float()).to(torch.bfloat16))          # "SGD update": same storage, new values
        rl.run()
        for i, (co, ci, t) in enumerate(shapes):
            got = rl.get('w%d' % i)
            w = views[i]
            if t == 1:
                want = torch.zeros(ci, (co + 63) // 64 * 64, device='cuda', dtype=torch.bfloat16)
                want[:, :co] = w.t()
            else:
                want = w.view(co, 3, 3, ci).flip(1, 2).permute(3, 1, 2, 0).reshape(ci, 9 * co)
            assert got.shape == want.shape, (i, got.shape, want.shape)
            assert torch.equal(got, want), (i, shapes[i], rnd)


@pytest.mark.parametrize('shape,dt', [((8, 38, 63, 256), torch.bfloat16), ((2464, 89), torch.float32), ((3, 5, 7, 72), torch.bfloat16),
                                      ((70000, 64), torch.bfloat16), ((1, 1024), torch.float32)])
def test_colsum_add_accumulates_bias_gradients(shape, dt):
    """relnet_colsum_add: out[c] += sum over the leading dims of x[..., c] (bias gradient into the flat buffer) against a float64 sum
    of the same (bf16-rounded) values; accumulates on top of what is there; a row-strided (sliced) 2-D operand."""
    ops, T = _mods()
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(dt).cuda()
    C = shape[-1]
    out = torch.full((C,), 0.5, device='cuda')
    T.colsum_add(x, out)
    want = x.double().reshape(-1, C).sum(0) + 0.5
    scale = x.double().abs().reshape(-1, C).sum(0).max().item()
    assert (out.double() - want).abs().max().item() <= 2e-6 * scale + 1e-5
    if len(shape) == 2 and C > 8:
        xs = x[:, :C - 8]                                # row stride > columns
        out2 = torch.zeros(C - 8, device='cuda')
        T.colsum_add(xs, out2)
        assert (out2.double() - xs.double().sum(0)).abs().max().item() <= 2e-6 * scale + 1e-5


def test_colsum_queue_one_grouped_launch_for_a_bucket():
    """train_ops.ColsumQueue / relnet_colsum_add_grouped (round 6): the 12 - 13 bias gradients of a training step in one launch per <= 16
    problems.  Mixed shapes (19 152 pixels x 128 / 256 channels, 2 464 rois x 1 024 / 2 048, a row-strided slice, a 100-row matrix), 20 problems
    (two launches), accumulation on top of existing values; an fp32 and a ragged-width operand take the single-launch kernel at once."""
    ops, T = _mods()
    g = torch.Generator().manual_seed(5)
    bf = torch.bfloat16
    shapes = [(19152, 128), (19152, 256), (2464, 1024), (2464, 2048), (100, 64), (64000, 128)] * 3 + [(700, 512), (3, 8)]
    xs = [torch.randn(r, c, generator=g).to(bf).cuda() for r, c in shapes]
    xs[2] = torch.randn(2464, 1536, generator=g).to(bf).cuda()[:, 512:]           # row pitch 1536, 1024 columns
    odd = [torch.randn(2464, 89, generator=g).cuda(), torch.randn(50, 12, generator=g).to(bf).cuda()]
    outs = [torch.full((x.shape[1],), 0.25, device='cuda') for x in xs + odd]
    q = T.ColsumQueue()
    T.COLSUM_QUEUE = q
    try:
        for x, o in zip(xs + odd, outs):
            T.colsum_add(x, o)
        assert len(q) == len(xs)                    # the two odd operands were summed immediately
        for x, o in zip(odd, outs[len(xs):]):
            assert (o.double() - (x.double().sum(0) + 0.25)).abs().max().item() <= 1e-4 * max(x.double().abs().sum(0).max().item(), 1)
        assert torch.equal(outs[0], torch.full_like(outs[0], 0.25))              # (queued: nothing launched yet)
        q.flush()
        assert len(q) == 0
    finally:
        T.COLSUM_QUEUE = None
    for x, o in zip(xs, outs):
        want = x.double().sum(0) + 0.25
        scale = x.double().abs().sum(0).max().item()
        assert (o.double() - want).abs().max().item() <= 2e-6 * scale + 1e-5, tuple(x.shape)


@pytest.mark.parametrize('tile', [0, 1, 3, 4, 5, 22])
def test_gemm_nt_mask_is_gemm_plus_shortcut_times_relu_mask(tile):
    """relnet_gemm_nt_mask: (A W^T + resid) where mask > 0, else 0, on every LDS-tiled configuration it may run on (tile 0 = the one
    pick_tile chooses) -- against float64 on the same bf16 operands; ragged M (not a multiple of any tile)."""
    ops, T = _mods()
    from relnet_amd import lib
    g = torch.Generator().manual_seed(11)
    M, N, K = 2394 + 37, 1024, 256
    a = _bf(torch.randn(M, K, generator=g)); w = _bf(torch.randn(N, K, generator=g) * 0.06)
    r = _bf(torch.randn(M, N, generator=g)); mk = _bf(torch.randn(M, N, generator=g).clamp(min=0))      # ~half the mask is exactly 0 (a ReLU output)
    want = (a.double() @ w.double().t() + r.double()) * (mk > 0).double()
    c = lambda t: t.cuda().to(torch.bfloat16)
    L = lib.load()
    try:
        L.relnet_gemm_force_tile(tile)
        got = ops.gemm_nt_mask(c(a), c(w), c(mk), resid=c(r))
        got0 = ops.gemm_nt_mask(c(a), c(w), c(mk))                           # without a shortcut operand
    finally:
        L.relnet_gemm_force_tile(0)
    assert _rel(got, want.float()) <= 1e-2
    assert bool((got.float().cpu()[mk <= 0] == 0).all())                     # masked entries are exact zeros
    assert _rel(got0, ((a.double() @ w.double().t()) * (mk > 0).double()).float()) <= 1e-2
    # ... and through the training building block: the identity-shortcut unit's data gradient, already masked for the previous unit
    B, H, W_ = 1, 33, 73
    P = B * H * W_
    x = c(mk[:P]).view(B, H, W_, N)                                          # the unit's input = previous unit's ReLU output
    dy = c(a[:P]).view(B, H, W_, K)
    wp = c(w.t().contiguous())                                               # forward weight [Cout = K, Cin = N]
    dx, _ = T.conv1x1_bwd(x, wp, dy, dx_add=c(r[:P]).view(B, H, W_, N), out_mask=x)
    assert _rel(dx.reshape(P, N), want[:P].float()) <= 1e-2


def test_weight_fragpack_equals_the_load_time_packers():
    """relnet_weight_fragpack (one grouped launch per training step) == relnet_pack_w_frag / ops.pack_chain_w1 (inference, load time),
    bit for bit, for every bottleneck width the chain kernels are built for."""
    ops, _ = _mods()
    g = torch.Generator().manual_seed(5)
    fp = ops.FragRepack('cuda')
    ws = {}
    for mid in (64, 128, 256, 512):
        w3 = (torch.randn(4 * mid, mid, generator=g) * 0.1).cuda().to(torch.bfloat16)
        w1 = (torch.randn(mid, 4 * mid, generator=g) * 0.1).cuda().to(torch.bfloat16)
        ws[mid] = (w3, w1)
        fp.add(('w3', mid), w3, 0); fp.add(('w1', mid), w1, 1)
    fp.build(); fp.run()
    torch.cuda.synchronize()
    for mid, (w3, w1) in ws.items():
        assert torch.equal(fp.get(('w3', mid)).view(-1), ops.pack_w_frag(w3, panel_only=False).view(-1)), mid
        assert torch.equal(fp.get(('w1', mid)).view(-1), ops.pack_chain_w1(w1).view(-1)), mid
    # the copies follow the weights: a second run after an in-place update
    ws[128][0].mul_(2)
    fp.run()
    assert torch.equal(fp.get(('w3', 128)).view(-1), ops.pack_w_frag(ws[128][0], panel_only=False).view(-1))


def test_relation_bwd_pack_and_lnms_scatter():
    ops, _ = _mods()
    g = torch.Generator().manual_seed(9)
    B, N, M, d = 3, 44, 40, 128
    dq, dk, dvw = torch.randn(B, N, d, generator=g), torch.randn(B, M, d, generator=g), torch.randn(B, M, d, generator=g)
    a3 = ops.relation_bwd_pack(dq.cuda(), dk.cuda(), dvw.cuda()).cpu()
    want = torch.zeros(B, N, 3 * d)
    want[:, :, :d] = dq; want[:, :M, d:2 * d] = dk; want[:, :M, 2 * d:] = dvw
    assert torch.equal(a3, want.to(torch.bfloat16))
    # adjoint of the per-class sort + take of the learn-NMS head
    Bn, Nn, C, F = 2, 50, 7, 12
    rank = torch.stack([torch.stack([torch.randperm(Nn, generator=g)[:F] for _ in range(C)]) for _ in range(Bn)]).to(torch.int32)     # [B,C,F]
    rank[1, 3, -2:] = -1                                                      # padding entries are skipped
    ds = torch.randn(Bn, F, C, generator=g)
    got = ops.lnms_scatter_bwd(ds.cuda(), rank.cuda(), Nn).cpu()
    ref = torch.zeros(Bn, Nn, C)
    for b in range(Bn):
        for c in range(C):
            for f in range(F):
                if rank[b, c, f] >= 0:
                    ref[b, rank[b, c, f], c] += ds[b, f, c]
    assert torch.allclose(got, ref, atol=1e-6)


@pytest.mark.parametrize('B,dt', [(8, torch.bfloat16), (1, torch.bfloat16), (9, torch.float32)])
def test_roi_pool_backward_owner_form_equals_scatter_and_definition(B, dt):
    """relnet_roi_pool_bwd_cl (round 6): one workgroup per (image, 8 channels) accumulates its H x W slab in LDS and flushes it once (steps of >= 8 images x 256
    channels; the one-image case runs the scatter kernel on both sides), against
    the atomic scatter kernel (relnet_roi_pool_bwd_debug(1)) and against the definition grad_in[b, c, argmax[r, c, ph, pw]] += grad_out[r, c, ph, pw]
    in float64.  Heavily overlapping rois (the contended case), rois of the images interleaved (the kernel selects by batch index, not by position),
    accumulation on top of a non-zero buffer is the caller's zeros + add."""
    ops, T = _mods()
    from relnet_amd import lib as L
    g = torch.Generator().manual_seed(40 + B)
    H, W, C, Rpi = 38, 63, 256, 77
    feat = torch.randn(B, H, W, C, generator=g).to(dt).cuda().permute(0, 3, 1, 2)
    rois = []
    for b in range(B):
        x1 = torch.rand(Rpi, generator=g) * 500; y1 = torch.rand(Rpi, generator=g) * 300
        w_ = 60 + torch.rand(Rpi, generator=g) * 400; h_ = 60 + torch.rand(Rpi, generator=g) * 250
        rois.append(torch.stack([torch.full((Rpi,), float(b)), x1, y1, (x1 + w_).clamp(max=999), (y1 + h_).clamp(max=599)], 1))
    rois = torch.cat(rois, 0)
    rois = rois[torch.randperm(rois.shape[0], generator=g)].contiguous().cuda()
    pooled, argmax = ops.roi_pool(feat, rois, (7, 7), 1.0 / 16, channels_last_out=True, want_argmax=True)
    gout = torch.randn(pooled.shape, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    assert gout.stride() == argmax.stride()
    res = {}
    for mode in (1, 0):
        L.load().relnet_roi_pool_bwd_debug(mode)
        try:
            res[mode] = ops.roi_pool_bwd(gout, argmax, rois, (B, C, H, W), channels_last=True).clone()
        finally:
            L.load().relnet_roi_pool_bwd_debug(0)
    torch.cuda.synchronize()
    want = np.zeros((B, C, H * W))
    am = argmax.cpu().numpy(); go = gout.double().cpu().numpy(); rb = rois[:, 0].long().cpu().numpy()
    for r in range(rois.shape[0]):
        a = am[r].reshape(C, -1); v = go[r].reshape(C, -1)
        for k in range(49):
            ok = a[:, k] >= 0
            np.add.at(want[rb[r]], (np.nonzero(ok)[0], a[ok, k]), v[ok, k])
    want = want.reshape(B, C, H, W)
    scale = np.abs(want).max()
    for mode in (0, 1):
        got = res[mode].double().cpu().numpy()
        assert np.abs(got - want).max() <= 2e-5 * scale, (mode, np.abs(got - want).max() / scale)
