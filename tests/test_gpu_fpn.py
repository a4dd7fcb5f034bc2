"""GPU parity of the FPN configuration's pieces against oracle/fpn.py.

Integer work (level assignment, regrouping, max pooling) is bit-exact; convolution stacks are compared at the
tolerances written next to each assertion."""
import os
import sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import cases  # noqa: E402
from oracle import fpn as OF  # noqa: E402
from oracle import network as ON  # noqa: E402
from oracle import relation as OR  # noqa: E402

pytestmark = pytest.mark.gpu
F = np.float32


def _mods():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops, backbone, detector
    return ops, backbone, detector


def _proposals(n, seed, im_h=800, im_w=1024):
    """n boxes spanning all four pyramid levels, with some exactly ON the level boundaries
    (sqrt(w*h) = 112, 224, 448: log2 = -1, 0, 1 exactly)."""
    rng = np.random.default_rng(seed)
    side = np.exp(rng.uniform(np.log(12), np.log(700), n))
    ar = np.exp(rng.uniform(-0.7, 0.7, n))
    w = np.minimum(side * ar, im_w - 2); h = np.minimum(side / ar, im_h - 2)
    x1 = rng.uniform(0, im_w - 1 - w); y1 = rng.uniform(0, im_h - 1 - h)
    b = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(F)
    for i, s in enumerate((112, 224, 448)):
        b[i] = [10, 20, 10 + s - 1, 20 + s - 1]                 # w = h = s exactly
        b[3 + i] = [5, 5, 5 + 2 * s - 1, 5 + s / 2 - 1]         # w*h = s^2 with w != h
    return b


def test_roi_dispatch_bit_exact():
    ops, _, _ = _mods()
    B, N = 3, 1000
    boxes = np.stack([_proposals(N, 100 + b) for b in range(B)])
    boxes[2, :, :] = _proposals(N, 7)                            # third image: forced empty level 0
    small = OF.roi_levels(boxes[2]) == 0
    boxes[2, small] = [100, 100, 400, 400]
    rois, level, perm, counts = ops.fpn_roi_dispatch(torch.as_tensor(boxes).cuda())
    for b in range(B):
        wr, wl, wp, wc = OF.roi_dispatch(boxes[b], dummy_for_empty=False)
        wr[:, 0] = b
        assert np.array_equal(counts[b].cpu().numpy(), wc)
        assert np.array_equal(level[b].cpu().numpy(), wl)
        assert np.array_equal(perm[b].cpu().numpy(), wp)
        assert np.array_equal(rois[b].cpu().numpy(), wr)
    assert int(counts[2, 0]) == 0 and (counts[:2] > 0).all()
    # boundary boxes land in the upper level (floor of an exact integer)
    lv = OF.roi_levels(boxes[0][:6])
    assert lv.tolist() == [1, 2, 3, 1, 2, 3]
    # 5-column input (batch index + box) gives the same grouping
    b5 = torch.cat([torch.zeros(B, N, 1), torch.as_tensor(boxes)], 2).cuda()
    r2 = ops.fpn_roi_dispatch(b5)
    assert torch.equal(r2[0], rois) and torch.equal(r2[2], perm)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_upsample2x_add(dtype):
    ops, _, _ = _mods()
    torch.manual_seed(1)
    top = torch.randn(2, 25, 32, 256, device='cuda').to(dtype)
    lat = torch.randn(2, 50, 64, 256, device='cuda').to(dtype)
    want = (lat.float() + top.float().repeat_interleave(2, 1).repeat_interleave(2, 2)).to(dtype)
    got = ops.upsample2x_add_(lat.clone(), top)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.upsample2x_add_(torch.zeros(1, 51, 64, 8, device='cuda').to(dtype), torch.zeros(1, 25, 32, 8, device='cuda').to(dtype))


def test_roi_pool_fpn_bit_exact():
    ops, _, _ = _mods()
    rng = np.random.default_rng(5)
    B, C, N = 2, 16, 300
    H, W = 256, 320
    feats = [rng.normal(0, 1, (B, C, H // s, W // s)).astype(F) for s in (4, 8, 16, 32)]
    boxes = np.stack([_proposals(N, 30 + b, H, W) for b in range(B)])
    rois, level, perm, counts = ops.fpn_roi_dispatch(torch.as_tensor(boxes).cuda())
    out = ops.roi_pool_fpn([torch.as_tensor(f).cuda() for f in feats], (1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0),
                           rois.view(-1, 5), level.view(-1))
    r = rois.cpu().numpy(); lv = level.cpu().numpy()
    for b in range(B):
        want = OF.pool_levels(feats, r[b], lv[b])
        assert np.array_equal(out[b * N:(b + 1) * N].cpu().numpy(), want)
    # bf16 channels-last maps, channels-last output (the pipeline layout)
    f16 = [torch.as_tensor(f).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for f in feats]
    o16 = ops.roi_pool_fpn(f16, (1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0), rois.view(-1, 5), level.view(-1), channels_last_out=True)
    fr = [t.float().cpu().numpy() for t in f16]
    want = np.concatenate([OF.pool_levels(fr, r[b], lv[b]) for b in range(B)])
    assert np.array_equal(o16.float().cpu().numpy(), want)


def test_fpn_detector_stagewise():
    ops, backbone, detector = _mods()
    H, W, N = 160, 224, 200
    p = backbone.init_params(seed=21, fpn=True)
    g = torch.Generator().manual_seed(22)
    for k in ('cls_score_weight', 'bbox_pred_weight'):
        p[k] = torch.randn(p[k].shape, generator=g) * 0.05
    for lvl in (4, 8, 16, 32):                         # O(1) pyramid features instead of the N(0, 0.01) init's ~1e-3
        p['fpn_ft%d_1x1_weight' % lvl] = p['fpn_ft%d_1x1_weight' % lvl] * 5
        p['fpn_ft%d_3x3_weight' % lvl] = p['fpn_ft%d_3x3_weight' % lvl] * 5
        p['fpn_ft%d_3x3_bias' % lvl] = torch.rand(256, generator=g) * 0.1
    data = torch.randn(1, 3, H, W, generator=g)
    boxes = _proposals(N, 23, H, W)[None]
    im_info = torch.tensor([[H, W, 1.0]])
    det = detector.FPNDetector(p, dtype=torch.float32)
    f = det.backbone.forward(data.cuda())
    with torch.no_grad():
        c2, c3, c4, c5 = ON.backbone(data, p, fpn=True)
        want = OF.fpn_neck(c2, c3, c4, c5, p)
    assert c5.shape[2:] == (H // 32, W // 32)
    for name, w in zip(('fpn_ft4', 'fpn_ft8', 'fpn_ft16', 'fpn_ft32'), want):
        got = f[name].float().cpu()
        assert got.shape == w.shape
        err = (got - w).abs().max().item() / w.abs().max().item()
        assert err < 3e-4, (name, err)               # fp32 conv stacks, different summation orders
    out = det.forward(data.cuda(), torch.as_tensor(boxes).cuda(), im_info.cuda())
    assert int(out['num_rows'][0]) == N and out['rois'].shape[1] == N + 4      # every level populated: N real rows, 4 padding rows
    for k in ('rois', 'roi_level', 'cls_score', 'bbox_pred'):
        out[k] = out[k][:, :N]
    rois = out['rois'][0].cpu().numpy(); lv = out['roi_level'][0].cpu().numpy()
    feats = [f[n].float().cpu().numpy() for n in ('fpn_ft4', 'fpn_ft8', 'fpn_ft16', 'fpn_ft32')]
    pooled_o = OF.pool_levels(feats, rois, lv)
    pn = {k: v.numpy() for k, v in p.items()}
    pn['fc_new_1_weight'], pn['fc_new_1_bias'] = pn['roi_pool_fc1_weight'], pn['roi_pool_fc1_bias']
    pn['fc_new_2_weight'], pn['fc_new_2_bias'] = pn['roi_pool_fc2_weight'], pn['roi_pool_fc2_bias']
    r = OR.relation_head(pooled_o, rois, pn, return_intermediates=True)
    cs, bp = r['cls_score'], r['bbox_pred']
    assert np.abs(out['cls_score'][0].cpu().numpy() - cs).max() <= 2e-4 * np.abs(cs).max()
    assert np.abs(out['bbox_pred'][0].cpu().numpy() - bp).max() <= 2e-4 * max(np.abs(bp).max(), 1e-3)
    assert int(out['num_detections'][0]) > 0
    # bf16 MFMA path, batch 2, same graph
    det16 = detector.FPNDetector(p, dtype=torch.bfloat16)
    data2 = torch.cat([data, torch.randn(1, 3, H, W, generator=g)]).cuda()
    boxes2 = torch.as_tensor(np.stack([boxes[0], _proposals(N, 24, H, W)])).cuda()
    f16 = det16.backbone.forward(data2)
    for name in ('fpn_ft4', 'fpn_ft32'):
        err = (f16[name][0].float() - f[name][0]).abs().max().item() / f[name].abs().max().item()
        assert err < 6e-2, (name, err)
    o16 = det16.forward(data2, boxes2, torch.tensor([[H, W, 1.0]] * 2).cuda())
    assert torch.isfinite(o16['cls_score']).all() and o16['cls_score'].shape == (2, N + 4, 81)
    assert torch.equal(o16['rois'][0, :N], out['rois'][0])      # dispatch does not depend on the dtype


def test_fpn_empty_level_dummy_roi_matches_reference_loader(golden):
    """A pyramid level that receives no roi gets the reference loader's all-zero dummy roi (core/rcnn.py:61-71): the device
    dispatch must produce EXACTLY the rows of `get_rcnn_testbatch` (tests/golden/fpn.npz, written by running the reference's
    own loader), and the dummy row must behave like a real roi of the graph: a key of both relation modules and a scored
    row -- checked against the oracle head evaluated on the reference's row list."""
    ops, backbone, detector = _mods()
    g = golden['fpn']
    for name in ('all_levels', 'empty_level0'):
        boxes = g[name + '/boxes']
        want = np.vstack([g['%s/rois_%d' % (name, l)] for l in range(4)]).astype(np.float32)
        rois, level, perm, counts, n_rows = ops.fpn_roi_dispatch(torch.as_tensor(boxes[None]).cuda(), pad_empty=True)
        n = int(n_rows[0])
        assert n == len(want) and rois.shape[1] == len(boxes) + 4
        assert np.array_equal(rois[0, :n].cpu().numpy(), want), name
        lv_want = np.concatenate([np.full(len(g['%s/rois_%d' % (name, l)]), l) for l in range(4)])
        assert np.array_equal(level[0, :n].cpu().numpy(), lv_want)
        pm = perm[0].cpu().numpy()
        assert (pm[n:] == -1).all() and (rois[0, n:, 1:] == 0).all()
        real = pm[:n][pm[:n] >= 0]
        assert sorted(real.tolist()) == list(range(len(boxes)))                         # a permutation of the input rows
        if name == 'empty_level0':
            assert pm[0] == -1 and (want[0] == 0).all() and int(counts[0, 0]) == 0      # the dummy roi leads the list
    # n_valid: rows past it leave the level lists and land behind the real rows
    boxes = g['all_levels/boxes'][None]
    nv = torch.tensor([250], dtype=torch.int32).cuda()
    rois, level, perm, counts, n_rows = ops.fpn_roi_dispatch(torch.as_tensor(boxes).cuda(), n_valid=nv, pad_empty=True)
    ref = ops.fpn_roi_dispatch(torch.as_tensor(boxes[:, :250].copy()).cuda(), pad_empty=True)
    n = int(n_rows[0])
    assert n == int(ref[4][0]) and torch.equal(rois[0, :n], ref[0][0, :n]) and torch.equal(perm[0, :n], ref[2][0, :n])
    assert sorted(perm[0, n:n + 150].cpu().tolist()) == list(range(250, 400)) and (rois[0, n:, 1:] == 0).all()


def test_fpn_detector_with_empty_level_equals_oracle_on_reference_rows():
    """FPNDetector on proposals that leave pyramid level 0 empty (two images, the second with all levels populated): per-roi
    outputs of the real rows == the oracle head on the reference's row list INCLUDING the dummy roi (float32), the padding
    rows are no keys (their content cannot change any real row) and never become detections."""
    ops, backbone, detector = _mods()
    H, W, N = 256, 320, 40
    p = backbone.init_params(seed=3, fpn=True)
    g = torch.Generator().manual_seed(4)
    for k in ('cls_score_weight', 'bbox_pred_weight'):
        p[k] = torch.randn(p[k].shape, generator=g) * 0.05
    for lvl in (4, 8, 16, 32):
        p['fpn_ft%d_1x1_weight' % lvl] = p['fpn_ft%d_1x1_weight' % lvl] * 5
        p['fpn_ft%d_3x3_weight' % lvl] = p['fpn_ft%d_3x3_weight' % lvl] * 5
        p['fpn_ft%d_3x3_bias' % lvl] = torch.rand(256, generator=g) * 0.1
    xy = torch.rand(N, 2, generator=g) * 20
    big = torch.cat([xy, xy + 150.0 + 100 * torch.rand(N, 2, generator=g)], 1).clamp(max=W - 1)    # levels 1-3 only
    boxes = torch.stack([big, torch.as_tensor(_proposals(N, 24, H, W))]).float().contiguous()
    data = torch.randn(2, 3, H, W, generator=g).cuda()
    im_info = torch.tensor([[H, W, 1.0]] * 2).cuda()
    det = detector.FPNDetector(p, dtype=torch.float32)
    out = det.forward(data, boxes.cuda(), im_info)
    counts = out['level_counts'].cpu().numpy()
    n_rows = out['num_rows'].cpu().numpy()
    assert counts[0, 0] == 0 and (counts[1] > 0).all() and n_rows.tolist() == [N + int((counts[0] == 0).sum()), N]
    f = det.backbone.forward(data)
    pn = {k: v.numpy() for k, v in p.items()}
    pn['fc_new_1_weight'], pn['fc_new_1_bias'] = pn['roi_pool_fc1_weight'], pn['roi_pool_fc1_bias']
    pn['fc_new_2_weight'], pn['fc_new_2_bias'] = pn['roi_pool_fc2_weight'], pn['roi_pool_fc2_bias']
    for b in range(2):
        rois_o, lv_o, perm_o, _ = OF.roi_dispatch(boxes[b].numpy(), dummy_for_empty=True)       # pinned to the reference loader
        n = len(rois_o)
        assert n == n_rows[b]
        got_rois = out['rois'][b, :n].cpu().numpy()
        assert np.array_equal(got_rois[:, 1:], rois_o[:, 1:]) and np.array_equal(out['roi_level'][b, :n].cpu().numpy(), lv_o)
        feats = [f[k][b:b + 1].float().cpu().numpy() for k in ('fpn_ft4', 'fpn_ft8', 'fpn_ft16', 'fpn_ft32')]
        r = OR.relation_head(OF.pool_levels(feats, rois_o, lv_o), rois_o, pn, return_intermediates=True)
        for key in ('cls_score', 'bbox_pred'):
            want = r[key]
            err = np.abs(out[key][b, :n].cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-3)
            assert err <= 2e-4, (b, key, err)
        # padding rows: zero probability, hence never detections
        assert (out['cls_prob'][b, n:] == 0).all()
        nd = int(out['num_detections'][b])
        assert nd > 0
    # the same graph WITHOUT the dummy roi differs on the real rows (it is one more relation key): the row matters
    det.pad_empty_levels = False
    o2 = det.forward(data, boxes.cuda(), im_info)
    d = (o2['cls_score'][0, :N - 0] - out['cls_score'][0, 1:N + 1]).abs().max().item()
    assert d > 1e-6
    assert torch.equal(o2['cls_score'][1], out['cls_score'][1, :N]) or \
        (o2['cls_score'][1] - out['cls_score'][1, :N]).abs().max().item() <= 1e-4 * out['cls_score'][1].abs().max().item()
