"""GPU parity of the relation-module backward (csrc/relation_bwd.hip + GEMM composition) against torch-CPU float64
autograd of the restated module (oracle/relation_torch.py).  fp32 path: every gradient within 2e-4 of its own
max-abs (fp32 accumulation over up to 300 x 1024 terms); bf16 path: 4e-2."""
import os
import sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import cases  # noqa: E402
from oracle import relation_torch as ORT  # noqa: E402

pytestmark = pytest.mark.gpu


def _autograd(feat, boxes, p, m, d_out):
    ft = torch.tensor(feat.astype(np.float64), requires_grad=True)
    pt = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in p.items()}
    y = ORT.relation_module(ft, boxes, pt, 1, m)
    (y * torch.as_tensor(d_out.astype(np.float64))).sum().backward()
    g = {k: v.grad.numpy() for k, v in pt.items()}
    g['d_roi_feat'] = ft.grad.numpy()
    return g


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_transpose_2d(dtype):
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    x = torch.randn(3, 300, 1024, device='cuda').to(dt)
    y = ops.transpose_2d(x[:, :, 64:640], pad_cols_to=32)              # strided rows, padded output
    assert y.shape == (3, 576, 320)
    assert torch.equal(y[:, :, :300], x[:, :, 64:640].transpose(1, 2)) and float(y[:, :, 300:].abs().max()) == 0
    z = ops.transpose_2d(x[0, :77, :130])
    assert torch.equal(z, x[0, :77, :130].t())


@pytest.mark.parametrize('n,m,seed,std', [(48, 48, 11, 0.03), (40, 32, 12, 0.05), (300, 300, 41, 0.02), (333, 300, 43, 0.02)])
def test_relation_module_backward_fp32(n, m, seed, std):
    import relnet_amd  # noqa: F401
    from relnet_amd import relation
    boxes, feat, p = cases.relation_case(n, m, seed, std)
    rng = np.random.default_rng(seed + 500)
    d_out = rng.normal(0, 1, (n, 1024)).astype(np.float32)
    want = _autograd(feat, boxes, p, m, d_out)
    pt = {k: torch.as_tensor(v) for k, v in p.items()}
    got = relation.attention_module_backward(torch.as_tensor(feat).cuda(), torch.as_tensor(boxes).cuda(), pt,
                                             torch.as_tensor(d_out).cuda(), nongt_dim=m, dtype=torch.float32)
    assert set(got) == set(want)
    for k in sorted(want):
        g = got[k].cpu().numpy().reshape(want[k].shape)
        if k == 'key_1_bias':      # exactly zero in exact arithmetic (softmax is invariant to a shift of all keys)
            assert np.abs(want[k]).max() < 1e-12 and np.abs(g).max() <= 1e-5 * np.abs(want['query_1_bias']).max()
            continue
        # pair_pos_fc1 gradients are DISCONTINUOUS at the 1e-6 floor of log(max(G, 1e-6)): a pair whose G rounds to
        # the other side of the floor in fp32 moves a term of size dL / 1e-6, so they get a wider bar
        tol = 2e-3 if k.startswith('pair_pos') else 2e-4
        assert _rel(g, want[k]) <= tol, (k, _rel(g, want[k]))
    # the clamp / ReLU branches are exercised: some geometry weights sit on the 1e-6 floor
    assert np.abs(want['pair_pos_fc1_1_weight']).max() > 0


def test_relation_module_backward_batched_bf16():
    """bf16 operands (MFMA 32x32x16), batch of 2 images with shared weights: parameter gradients sum over images."""
    import relnet_amd  # noqa: F401
    from relnet_amd import relation
    n, m = 300, 300
    boxes0, feat0, p = cases.relation_case(n, m, 41, 0.02)
    boxes1 = cases.random_boxes(n, 777)
    rng = np.random.default_rng(9)
    feat1 = rng.normal(0, 1, feat0.shape).astype(np.float32)
    d_out = rng.normal(0, 1, (2, n, 1024)).astype(np.float32)
    r = lambda a: torch.as_tensor(a).to(torch.bfloat16).float().numpy()          # operands as the kernel sees them
    w0 = _autograd(r(feat0), boxes0, {k: (r(v) if 'pair_pos' not in k and 'bias' not in k else v) for k, v in p.items()}, m, r(d_out[0]))
    w1 = _autograd(r(feat1), boxes1, {k: (r(v) if 'pair_pos' not in k and 'bias' not in k else v) for k, v in p.items()}, m, r(d_out[1]))
    pt = {k: torch.as_tensor(v) for k, v in p.items()}
    got = relation.attention_module_backward(torch.as_tensor(np.stack([feat0, feat1])).cuda(),
                                             torch.as_tensor(np.stack([boxes0, boxes1])).cuda(), pt,
                                             torch.as_tensor(d_out).cuda(), nongt_dim=m, dtype=torch.bfloat16)
    for k in sorted(w0):
        want = np.stack([w0[k], w1[k]]) if k == 'd_roi_feat' else w0[k] + w1[k]
        g = got[k].cpu().numpy().reshape(want.shape)
        if k == 'key_1_bias':
            assert np.abs(g).max() <= 1e-2 * np.abs(w0['query_1_bias'] + w1['query_1_bias']).max()
            continue
        assert _rel(g, want) <= 4e-2, (k, _rel(g, want))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_key_count_masks_padding_rows(dtype):
    """Fixed-size roi buffers with a different number of REAL rows per image (FPN dummy rois, short proposal lists): with
    key_count[b] the first key_count[b] rows are the keys of image b -- forward and backward must equal the same module run
    on that image alone with nongt_dim = key_count[b] (float32: to rounding of the shared GEMMs; bf16: to its tolerance),
    whatever the padding rows contain."""
    import relnet_amd  # noqa: F401
    from relnet_amd import relation
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    tol = 2e-5 if dtype == 'f32' else 3e-2
    n = 70
    counts = [70, 41, 33, 64]
    B = len(counts)
    boxes, feat, p = cases.relation_case(n, n, 77, 0.04)
    rng = np.random.default_rng(5)
    feats = np.stack([feat + 0.3 * rng.normal(0, 1, feat.shape).astype(np.float32) for _ in range(B)])
    bxs = np.stack([boxes] * B)
    for b, c in enumerate(counts):        # poison the padding rows: they must not influence the real ones
        feats[b, c:] = 50.0 * rng.normal(0, 1, (n - c, 1024))
        bxs[b, c:] = 0.0
    d_out = rng.normal(0, 1, (B, n, 1024)).astype(np.float32)
    for b, c in enumerate(counts):
        d_out[b, c:] = 0.0                # padded rows carry no loss (label -1, zero weights)
    pt = {k: torch.as_tensor(v) for k, v in p.items()}
    kc = torch.tensor(counts, dtype=torch.int32).cuda()
    F_, X_, D_ = torch.as_tensor(feats).cuda(), torch.as_tensor(bxs).cuda(), torch.as_tensor(d_out).cuda()
    y = relation.attention_module_multi_head(F_, X_, pt, nongt_dim=n, dtype=dt, key_count=kc)
    g = relation.attention_module_backward(F_, X_, pt, D_, nongt_dim=n, dtype=dt, key_count=kc)
    acc = {}
    for b, c in enumerate(counts):
        yb = relation.attention_module_multi_head(F_[b, :c], X_[b, :c], pt, nongt_dim=c, dtype=dt)
        err = (y[b, :c].float() - yb.float()).abs().max().item() / yb.float().abs().max().item()
        assert err <= tol, (b, c, err)
        gb = relation.attention_module_backward(F_[b, :c], X_[b, :c], pt, D_[b, :c], nongt_dim=c, dtype=dt)
        e = _rel(g['d_roi_feat'][b, :c].float().cpu().numpy(), gb['d_roi_feat'].float().cpu().numpy())
        assert e <= 10 * tol, ('d_roi_feat', b, e)
        for k, v in gb.items():
            if k != 'd_roi_feat':
                acc[k] = acc.get(k, 0) + v.double().cpu().numpy()
    assert torch.isfinite(g['d_roi_feat']).all()
    for k, want in acc.items():           # parameter gradients: the sum over the images' own runs
        if k == 'key_1_bias':
            continue
        e = _rel(g[k].double().cpu().numpy().reshape(want.shape), want)
        assert e <= 10 * tol, (k, e)


@pytest.mark.parametrize('n,m,kc,B', [(100, 100, False, 5), (128, 96, False, 5), (37, 37, False, 5), (70, 70, True, 5), (100, 100, True, 21)])
def test_small_n_fused_backward_equals_the_two_kernel_form(n, m, kc, B, monkeypatch):
    """N, Mpad <= 128 (the learn-NMS head's relation module: 100 ranked rois per (image, class)): relation_attention_bwd_small_kernel -- q part and kv
    part of one (image, head) in one workgroup, K / VW / K^T staged in LDS, S and dL handed over through LDS as the bf16 MFMA operands they become
    anyway -- against the two-kernel form (fp32 S / dL maps through HBM) on the same operands: same rounding points, so dQ, dK, dVW and dL agree bit for bit."""
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    g = torch.Generator().manual_seed(n * 7 + m)
    H, d = 16, 1024                 # B = 21: 336 (image, head) pairs on 256 persistent workgroups -- the loop with the next pair's operands prefetched
    bt = torch.bfloat16
    mpad, npad = ops.pad32(m), ops.pad32(n)
    qk = (torch.randn(B, n, 2 * d, generator=g) * 0.3).cuda().to(bt)
    q, k = qk[:, :, :d], qk[:, :m, d:]
    vw = (torch.randn(B, m, d, generator=g) * 0.3).cuda().to(bt)
    dy = torch.randn(B, n, d, generator=g).cuda().to(bt)
    y = torch.randn(B, n, d, generator=g).cuda().to(bt)
    bout = torch.randn(d, generator=g).cuda()
    bias = (torch.randn(B, H, n, mpad, generator=g) - 2.0).cuda()
    kt = torch.zeros(B, d, mpad, device='cuda', dtype=bt); ops.transpose_2d(k, out=kt)
    qt = ops.transpose_2d(q, pad_cols_to=32); dyt = ops.transpose_2d(dy, pad_cols_to=32)
    key_count = torch.tensor(([m, max(1, m // 2), m - 3, 1, m] * ((B + 4) // 5))[:B], dtype=torch.int32).cuda() if kc else None
    res = {}
    for small in ('1', '0'):
        monkeypatch.setenv('RELNET_REL_BWD_SMALL', small)
        res[small] = ops.relation_attention_bwd(q, k, kt, vw, bias, dy, y, bout, qt, dyt, m, key_count=key_count)
    assert res['1'][3] is None and res['0'][3] is not None              # the fused form never writes the S map
    for i, name in ((0, 'dq'), (1, 'dk'), (2, 'dvw')):
        assert torch.equal(res['1'][i], res['0'][i]), name
    assert torch.equal(res['1'][4][..., :m], res['0'][4][..., :m])       # dL (columns >= M are padding in both)
    assert torch.isfinite(res['1'][0]).all() and float(res['1'][1].abs().max()) > 0
    # packed output: the kernel writes bf16 (dQ | dK | dVW) straight into the projection backward's operand = relnet_relation_bwd_pack of the above
    monkeypatch.setenv('RELNET_REL_BWD_SMALL', '1')
    a3 = torch.zeros(B, n, 3 * d, device='cuda', dtype=bt)
    out = ops.relation_attention_bwd(q, k, kt, vw, bias, dy, y, bout, qt, dyt, m, key_count=key_count, packed_out=a3)
    assert out[0] is a3 and out[1] is None and out[2] is None
    assert torch.equal(a3, ops.relation_bwd_pack(res['1'][0], res['1'][1], res['1'][2]))
    assert torch.equal(out[4][..., :m], res['1'][4][..., :m])
