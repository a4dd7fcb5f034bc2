"""Every instantiation of the bf16 MFMA GEMM / implicit-GEMM convolution kernel (csrc/gemm.hip), forced one by one:
tile configurations x {plain GEMM, 1x1 conv (stride 1 / 2), 3x3 conv (dilation 1 / 2), 7x7 stem} x XCD swizzle on / off
x row-panel walk (n_loop 1 / 2 / 4) x {bf16, f32} outputs, on shapes with >= 16 tiles, ragged M / N edges and the
row counts of the benchmark (M = 4 x 2394 and 54 x 2394 = 129 276).

Oracle: a convolution is unambiguous (reference graph: symbols/resnet_v1_101_rcnn_base.py:29-693, MXNet Convolution =
cross-correlation, FullyConnected = x W^T + b).  Checked two ways:
  * ALL outputs against torch float32 on the same device (library code used as the checker only);
  * a sample of output rows (first / last rows, tile boundaries, random) against float64 numpy on the host, computed
    from the definition -- independent of any library and transpose-detecting (non-symmetric random operands)."""
import itertools
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rn():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops, lib
    L = lib.load()
    yield ops, L
    L.relnet_gemm_force_tile(0); L.relnet_gemm_force_nloop(0); L.relnet_gemm_set_swizzle(1)


def _variants(L, quick=False):
    nt = L.relnet_gemm_tile_count()
    tiles = range(1, nt + 1)
    if os.environ.get('RELNET_TEST_TILES'):            # development aid: only these tile configurations
        tiles = [int(t) for t in os.environ['RELNET_TEST_TILES'].split(',')]
    if quick:
        return [(t, s, 1) for t in tiles for s in (0, 1)]
    return [(t, s, nl) for t in tiles for s in (0, 1) for nl in (1, 2, 4)]


def _set(L, t, s, nl):
    L.relnet_gemm_force_tile(t); L.relnet_gemm_set_swizzle(s); L.relnet_gemm_force_nloop(nl)


def _sample_rows(M, rng, n=48):
    edge = [0, 1, 31, 32, 63, 64, 127, 128, 255, 256, 257, 511, 512, M - 257, M - 256, M - 129, M - 65, M - 33, M - 2, M - 1]
    rows = [r for r in edge if 0 <= r < M] + list(rng.integers(0, M, n))
    return np.unique(np.asarray(rows, dtype=np.int64))


def _check(got, want_dev, rows, want_rows64, tol, what):
    g = got.reshape(want_dev.shape).float()
    scale = want_dev.abs().max().item()
    err = (g - want_dev).abs().max().item() / scale
    assert err < tol, (what, 'all outputs vs torch fp32', err)
    sub = g.reshape(-1, g.shape[-1])[torch.as_tensor(rows, device=g.device)].double().cpu().numpy()
    err64 = np.abs(sub - want_rows64).max() / np.abs(want_rows64).max()
    assert err64 < tol, (what, 'sampled rows vs float64', err64)


@pytest.mark.parametrize('M,N,K,extras', [
    (9576, 256, 256, 'bias+resid+relu'), (9576, 1024, 1024, 'bias'), (9576, 72, 512, 'bias'),
    (3000, 1024, 12544, 'bias+relu'), (129276, 256, 1024, 'bias+relu'), (4100, 2048, 1024, 'rowbias'),
    (16200, 89, 1024, 'bias')])
def test_gemm_nt_every_tile(rn, M, N, K, extras):
    ops, L = rn
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().to(torch.bfloat16)
    bias_mode = 2 if 'rowbias' in extras else 1
    b = torch.randn(M if bias_mode == 2 else N, generator=g).cuda()
    relu = 'relu' in extras
    rng = np.random.default_rng(M)
    rows = _sample_rows(M, rng)
    a64, w64 = a[torch.as_tensor(rows).cuda()].double().cpu().numpy(), w.double().cpu().numpy()
    for odt, tol in ((torch.bfloat16, 1e-2), (torch.float32, 3e-5 * K ** 0.5)):
        res = torch.randn(M, N, generator=g).cuda().to(odt) if 'resid' in extras else None
        want = a.float() @ w.float().t() + (b[:, None] if bias_mode == 2 else b[None, :])
        w64r = a64 @ w64.T + (b.double().cpu().numpy()[rows][:, None] if bias_mode == 2 else b.double().cpu().numpy()[None, :])
        if res is not None:
            want = want + res.float(); w64r = w64r + res[torch.as_tensor(rows).cuda()].double().cpu().numpy()
        if relu:
            want = want.relu(); w64r = np.maximum(w64r, 0)
        for (t, s, nl) in _variants(L, quick=(M > 20000 or K > 4096)):
            _set(L, t, s, nl)
            got = ops.gemm_nt(a, w, b, bias_mode=bias_mode, resid=res, relu=relu, out_dtype=odt)
            _check(got, want, rows, w64r, tol, ('gemm', M, N, K, 'tile', t, 'swz', s, 'nloop', nl, str(odt)))
    _set(L, 0, 1, 0)


def _conv_rows64(x, w_oihw, bias, rows, Hout, Wout, stride, pad, dil):
    """float64 definition of the convolution for the sampled output pixels (NHWC input, OIHW weights)."""
    xn = x.double().cpu().numpy()
    wn = w_oihw.double().cpu().numpy()
    B, H, W, Cin = xn.shape
    Cout, _, R, S = wn.shape
    out = np.zeros((len(rows), Cout))
    for i, m in enumerate(rows):
        b, rem = divmod(int(m), Hout * Wout)
        oy, ox = divmod(rem, Wout)
        acc = bias.double().cpu().numpy().copy()
        for r in range(R):
            iy = oy * stride - pad + r * dil
            if iy < 0 or iy >= H:
                continue
            for s in range(S):
                ix = ox * stride - pad + s * dil
                if 0 <= ix < W:
                    acc += wn[:, :, r, s] @ xn[b, iy, ix]
        out[i] = acc
    return out


@pytest.mark.parametrize('B,hw,cin,cout,k,stride,dil,extras', [
    (4, (38, 63), 256, 256, 3, 1, 1, 'relu'),            # res4 3x3, M = 9576, K = 2304
    (4, (38, 63), 512, 512, 3, 1, 2, 'relu'),            # res5 dilated 3x3, K = 4608
    (4, (38, 63), 256, 1024, 1, 1, 1, 'resid+relu'),     # res4 expand + shortcut
    (4, (75, 125), 512, 256, 1, 2, 1, 'relu'),           # strided 1x1 (first conv of a stage), M = 9576
    (2, (150, 250), 64, 64, 3, 1, 1, 'relu'),            # res2 3x3: N = 64, K = 576
    (3, (38, 63), 1024, 72, 1, 1, 1, ''),                # rpn_out: ragged N
    (54, (38, 63), 256, 256, 3, 1, 1, 'relu'),           # the benchmark's res4 3x3: M = 129 276
    (54, (38, 63), 256, 1024, 1, 1, 1, 'resid+relu'),    # the benchmark's res4 expand
    (54, (38, 63), 1024, 256, 1, 1, 1, 'relu'),          # the benchmark's res4 reduce
    (2, (150, 250), 64, 256, 1, 1, 1, 'resid+relu'),     # res2 expand: K = 64   (row-panel kernel, tile 13)
    (3, (75, 125), 128, 512, 1, 1, 1, 'resid+relu'),     # res3 expand: K = 128
    (5, (38, 63), 512, 2048, 1, 1, 1, 'resid+relu'),     # res5 expand: K = 512, ragged last row panel
    (3, (38, 63), 256, 512, 1, 1, 1, 'relu'),            # K = 256 without a shortcut
])
def test_conv2d_every_tile(rn, B, hw, cin, cout, k, stride, dil, extras):
    ops, L = rn
    H, W = hw
    g = torch.Generator().manual_seed(cin + cout + k + B)
    x = torch.randn(B, H, W, cin, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).cuda().to(torch.bfloat16)
    b = torch.randn(cout, generator=g).cuda()
    pad = dil * (k // 2)
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), w.float(), b, stride=stride, padding=pad, dilation=dil).permute(0, 2, 3, 1)
    Hout, Wout = ref.shape[1], ref.shape[2]
    M = B * Hout * Wout
    rows = _sample_rows(M, np.random.default_rng(M + k), n=24)
    r64 = _conv_rows64(x, w, b, rows, Hout, Wout, stride, pad, dil)
    wp = ops.pack_conv_weight(w)
    wf = ops.pack_w_frag(wp) if (k == 1 and stride == 1) else None          # fragment-order copy for the row-panel kernel (tile 14)
    relu = 'relu' in extras
    big = M > 20000
    for odt, tol in ((torch.bfloat16, 1e-2), (torch.float32, 3e-5 * (cin * k * k) ** 0.5)):
        if big and odt == torch.float32 and cout > 256:
            continue
        res = torch.randn(ref.shape, generator=g).cuda().to(odt) if 'resid' in extras else None
        want, w64 = ref, r64
        if res is not None:
            want = want + res.float(); w64 = w64 + res.reshape(M, cout)[torch.as_tensor(rows).cuda()].double().cpu().numpy()
        if relu:
            want = want.relu(); w64 = np.maximum(w64, 0)
        for (t, s, nl) in _variants(L, quick=big):
            _set(L, t, s, nl)
            got = ops.conv2d_nhwc(x, wp, b, ksize=k, stride=stride, pad=pad, dil=dil, relu=relu, resid=res, out_dtype=odt, w_frag=wf)
            _check(got, want, rows, w64, tol, ('conv', B, hw, cin, cout, k, stride, dil, 'tile', t, 'swz', s, 'nloop', nl, str(odt)))
    _set(L, 0, 1, 0)


def test_stem_every_tile(rn):
    ops, L = rn
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 300, 500, generator=g).cuda()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).cuda()
    b = torch.randn(64, generator=g).cuda()
    want = torch.relu(F.conv2d(x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), b, stride=2, padding=3)).permute(0, 2, 3, 1)
    wp = ops.pack_stem_weight(w)
    for (t, s, nl) in _variants(L, quick=True):
        _set(L, t, s, nl)
        got = ops.stem_conv7(x, wp, b, relu=True)
        err = (got.float() - want).abs().max().item() / want.abs().max().item()
        assert got.shape == want.shape and err < 1e-2, (t, s, nl, err)
    _set(L, 0, 1, 0)


def test_auto_selection_covers_the_benchmark_tiles(rn):
    """The shapes above must make the AUTOMATIC choice land on every configuration the benchmark uses."""
    ops, L = rn
    L.relnet_gemm_force_tile(0)
    picks = set()
    for (M, N, K, batch, odt) in ((129276, 1024, 256, 1, 1), (129276, 256, 2304, 1, 1), (9576, 256, 2304, 1, 1),
                                  (2025000, 64, 576, 1, 1), (506250, 128, 1152, 1, 1), (16200, 1024, 12544, 1, 1),
                                  (1024, 300, 1024, 54, 1), (129276, 72, 512, 1, 0), (2304, 256, 129276 // 64 * 64, 1, 0)):
        picks.add(L.relnet_gemm_pick_tile(M, N, K, batch, odt))
    assert picks <= set(range(1, L.relnet_gemm_tile_count() + 1)) and len(picks) >= 3, picks


@pytest.mark.parametrize('tile', [2, 3])
def test_gemm_nt_f16_matches_float64(tile):
    """relnet_gemm_nt_f16 (IEEE-half operands, v_mfma_f32_32x32x16_f16, fp32 out): the measurement twin of the bf16 GEMM -- against
    float64 on the same fp16-rounded operands, ragged M / N."""
    import relnet_amd  # noqa: F401
    from relnet_amd import lib
    g = torch.Generator().manual_seed(tile)
    M, N, K = 777, 320, 448
    a = torch.randn(M, K, generator=g).to(torch.float16)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.float16)
    want = a.double() @ w.double().t()
    ad, wd = a.cuda(), w.cuda()
    out = torch.zeros(M, N, device='cuda', dtype=torch.float32)
    lib.call('relnet_gemm_nt_f16', ad.data_ptr(), ad.stride(0), wd.data_ptr(), wd.stride(0), out.data_ptr(), out.stride(0), M, N, K, tile,
             torch.cuda.current_stream().cuda_stream)
    err = (out.cpu().double() - want).abs().max().item() / want.abs().max().item()
    assert err <= 2e-5, err               # fp16 products are exact in fp32; only the fp32 accumulation order differs


def test_split_k_tile_on_the_one_image_shapes(rn):
    """Tile configuration 23 (round 6): the k-loop of a launch of <= 1 workgroup per CU split over 2..8 workgroups per tile, fp32 partials
    summed in split order by the last arriver.  On the one-image shapes pick_tile's rule chooses it by itself (res4 3x3: 152 tiles x 36
    k-slabs, fc_new_1: 80 tiles x 196 k-slabs, res5 3x3 dilated, rpn_out with a ragged N): against float64 from the definition; the result
    must not depend on the arrival order (bit-identical over repeated launches, which also shows the tile counters return to zero), and two
    streams launching at the same time use different slots of the work area."""
    ops, L = rn
    _set(L, 0, 1, 0)
    g = torch.Generator().manual_seed(23)
    rng = np.random.default_rng(23)
    assert ops.gemm_workspace() is not None
    # fc_new_1 at one image (300 rois) and its ways: auto (6), forced 2 .. 8, off
    M, N, K = 300, 1024, 12544
    a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().to(torch.bfloat16)
    b = torch.randn(N, generator=g).cuda()
    want64 = np.maximum(a.double().cpu().numpy() @ w.double().cpu().numpy().T + b.double().cpu().numpy(), 0)
    outs = {}
    for ways in (0, 2, 3, 5, 8, 1):
        L.relnet_gemm_debug_splitk(ways)
        L.relnet_gemm_force_tile(23 if ways >= 2 else 0)
        for odt, tol in ((torch.bfloat16, 1e-2), (torch.float32, 2e-5 * K ** 0.5)):
            got = [ops.gemm_nt(a, w, b, relu=True, out_dtype=odt) for _ in range(3)]
            assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2]), ('not deterministic', ways, odt)
            err = np.abs(got[0].double().cpu().numpy() - want64).max() / np.abs(want64).max()
            assert err < tol, (ways, odt, err)
            outs[(ways, odt)] = got[0]
    L.relnet_gemm_debug_splitk(0); L.relnet_gemm_force_tile(0)
    # (the split changes the fp32 summation order only: bf16 outputs of the split and unsplit launches agree to an ulp of bf16)
    d = (outs[(0, torch.bfloat16)].float() - outs[(1, torch.bfloat16)].float()).abs().max().item()
    assert d <= 2 ** -7 * outs[(1, torch.bfloat16)].float().abs().max().item()
    # convolutions of the one-image step, with shortcut / ReLU epilogues run by the last arriver
    for (hw, cin, cout, k, dil, extras) in (((38, 63), 256, 256, 3, 1, 'relu'), ((38, 63), 512, 512, 3, 2, 'relu'),
                                           ((38, 63), 1024, 256, 1, 1, 'resid+relu'), ((38, 63), 512, 72, 1, 1, ''),
                                           ((38, 63), 1024, 512, 3, 1, 'relu')):          # (the last: rpn_conv_3x3, split by the -2 rule itself)
        x = torch.randn(1, hw[0], hw[1], cin, generator=g).cuda().to(torch.bfloat16)
        wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(torch.bfloat16)
        wp = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().cuda()
        bias = torch.randn(cout, generator=g).cuda()
        Mo = hw[0] * hw[1]
        rows = _sample_rows(Mo, rng, 32)
        w64 = _conv_rows64(x, wt.cuda(), bias, rows, hw[0], hw[1], 1, dil * (k // 2), dil)
        for odt, tol in ((torch.bfloat16, 1e-2), (torch.float32, 3e-5 * (cin * k * k) ** 0.5)):
            res = torch.randn(1, hw[0], hw[1], cout, generator=g).cuda().to(odt) if 'resid' in extras else None
            want = w64 + (res.view(-1, cout)[torch.as_tensor(rows).cuda()].double().cpu().numpy() if res is not None else 0)
            if 'relu' in extras:
                want = np.maximum(want, 0)
            runs = []
            for ways in (3, 1):               # three ways on tile 23 (forced) against unsplit
                L.relnet_gemm_debug_splitk(ways); L.relnet_gemm_force_tile(23 if ways > 1 else 0)
                runs.append(ops.conv2d_nhwc(x, wp, bias, ksize=k, pad=dil * (k // 2), dil=dil, relu='relu' in extras, resid=res, out_dtype=odt))
            L.relnet_gemm_debug_splitk(-2 if cin * k * k >= 128 * 64 else 3); L.relnet_gemm_force_tile(0 if cin * k * k >= 128 * 64 else 23)
            again = ops.conv2d_nhwc(x, wp, bias, ksize=k, pad=dil * (k // 2), dil=dil, relu='relu' in extras, resid=res, out_dtype=odt)
            L.relnet_gemm_debug_splitk(0); L.relnet_gemm_force_tile(0)
            assert torch.equal(runs[0], again)            # (rpn_conv_3x3: the automatic -2 rule picks the same three ways)
            for r_ in runs:
                sub = r_.view(-1, cout)[torch.as_tensor(rows).cuda()].double().cpu().numpy()
                assert np.abs(sub - want).max() / np.abs(want).max() < tol, (cin, cout, k, odt)
            full = (runs[0].float() - runs[1].float()).abs().max().item() / runs[1].float().abs().max().item()
            assert full < (1e-2 if odt == torch.bfloat16 else 1e-5), full
    # two streams at once: each owns a slot, results equal the single-stream ones
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(4):
        with torch.cuda.stream(s1):
            o1 = ops.gemm_nt(a, w, b, relu=True, out_dtype=torch.bfloat16)
        with torch.cuda.stream(s2):
            o2 = ops.gemm_nt(a, w, b, relu=True, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(o1, outs[(0, torch.bfloat16)]) and torch.equal(o2, outs[(0, torch.bfloat16)])
