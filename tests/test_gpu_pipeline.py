"""GPU parity of the remaining hot-path rows, stage by stage ("teacher forced": each HIP stage
is checked against the oracle applied to the SAME inputs, so a 1e-6 difference in one stage
cannot flip a discrete decision (sort order, NMS) in the next and hide a real error)."""
import numpy as np
import pytest
import torch

import cases
from oracle import network as ON
from oracle import proposal as OP
from oracle import roi_pooling as ORP
from oracle import relation as OR
from oracle import postprocess as OPP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rn():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops, lib, backbone, detector
    lib.load()
    return ops, backbone, detector


def _np(t):
    return t.detach().float().cpu().numpy()


@pytest.mark.parametrize('N', [120, 700])      # 700: the 4-wavefront kernel of the FPN graphs (N up to 1024)
def test_postprocess_softnms_and_nms(rn, N):
    ops, _, _ = rn
    rng = np.random.default_rng(77)
    B, C = 2, 9
    rois = np.stack([np.hstack((np.full((N, 1), b, np.float32), cases.random_boxes(N, 80 + b))) for b in range(B)])
    cls_score = rng.normal(0, 2.5, (B, N, C)).astype(np.float32)
    bbox = rng.normal(0, 0.2, (B, N, 8)).astype(np.float32)
    im_info = np.array([[600, 1000, 1.5], [600, 1000, 1.0]], np.float32)
    d = lambda x: torch.as_tensor(x).cuda()
    prob, boxes = ops.detect_head(d(cls_score.reshape(B * N, C)), d(bbox.reshape(B * N, 8)), d(rois.reshape(B * N, 5)), d(im_info), N)
    prob, boxes = prob.view(B, N, C), boxes.view(B, N, 4)
    for soft, param in ((True, 0.6), (False, 0.5)):
        dets, counts = ops.class_nms(prob, boxes, 1e-3, param, soft)
        out, out_count, thresh, total = ops.image_topk(dets, counts, 100)
        for b in range(B):
            want_prob = OPP.softmax_rows(cls_score[b])
            np.testing.assert_allclose(_np(prob[b]), want_prob, rtol=2e-6, atol=1e-9)
            _, wboxes = OPP.im_detect(rois[b], want_prob, bbox[b], im_info[b])
            np.testing.assert_allclose(boxes[b].cpu().numpy(), wboxes[:, 4:8], rtol=1e-12, atol=1e-9)
            # NMS stages on the GPU's own probabilities / boxes
            full = np.zeros((N, 8)); full[:, 4:8] = boxes[b].cpu().numpy()
            want = OPP.detections(_np(prob[b]), full, C, 1e-3, param, soft, 100)
            raw = OPP.detections(_np(prob[b]), full, C, 1e-3, param, soft, -1)
            for c in range(C - 1):
                k = int(counts[b, c])
                assert k == len(raw[c])
                np.testing.assert_allclose(dets[b, c, :k].cpu().numpy(), raw[c], rtol=1e-10, atol=1e-12)
            got = out[b, :int(out_count[b])].cpu().numpy()
            flat = np.concatenate([np.hstack((np.full((len(w), 1), c + 1.0), w[:, 4:5], w[:, :4])) for c, w in enumerate(want)])
            assert got.shape == flat.shape
            np.testing.assert_allclose(got, flat.astype(np.float32), rtol=1e-6)


@pytest.mark.parametrize('relation', [True, False])
def test_detector_fp32_stagewise(rn, relation):
    ops, backbone, detector = rn
    H, W = 192, 256
    torch.manual_seed(0)
    p = backbone.init_params(seed=3)
    # spread the class scores / box deltas more than the N(0,0.01) init does
    g = torch.Generator().manual_seed(5)
    for k in ('cls_score_weight', 'bbox_pred_weight'):
        p[k] = torch.randn(p[k].shape, generator=g) * 0.05
    # exact score ties (identical all-zero pooled features) are order-unspecified in the reference
    # (unstable argsort, nms.py:108,133): keep the feature map strictly positive so none occur
    p['conv_new_1_bias'] = torch.rand(256, generator=g) * 0.1 + 0.05
    data = torch.randn(1, 3, H, W, generator=g)      # unit-variance pixels: O(1) RPN deltas
    im_info = torch.tensor([[H, W, 1.0]])
    cfg = detector.Config()
    cfg.rpn_post_nms_top_n = 100
    det = detector.Detector(p, dtype=torch.float32, relation=relation, im_hw=(H, W), cfg=cfg)
    f = det.backbone.forward(data.cuda())
    out = det.forward(data.cuda(), im_info.cuda())
    # A1/A2: backbone + RPN head vs torch-CPU fp32 restatement
    with torch.no_grad():
        c4, c5 = ON.backbone(data, p)
        cls, box, feat = ON.rpn_and_feat(c4, c5, p)
    for got, want in ((f['conv4'], c4), (f['conv5'], c5), (f['conv_new_1_relu'], feat),
                      (f['rpn_cls_score'], cls), (f['rpn_bbox_pred'], box)):
        err = (got.float().cpu() - want).abs().max().item() / want.abs().max().item()
        assert err < 2e-4, err
    # A3: proposal on the GPU's RPN maps -> bit exact rois
    prob = ON.rpn_softmax(_np(f['rpn_cls_score']))
    rois_o, _, dbg = OP.proposal(prob, _np(f['rpn_bbox_pred']), im_info.numpy(), 16, cfg.anchor_scales,
                                 cfg.anchor_ratios, 6000, 100, 0.7, 0, return_debug=True)
    assert dbg['n_kept'] >= 100
    rois = _np(out['rois'][0])
    # fp32 softmax of two logits: the GPU expf may differ by an ulp from numpy -> compare decisions
    assert np.array_equal(rois, rois_o) or np.abs(rois - rois_o).max() < 1e-3
    # A4: ROIPooling of the GPU feature map -> bit exact
    pooled_o = ORP.roi_pooling(_np(f['conv_new_1_relu']), rois)
    pooled = ops.roi_pool(f['conv_new_1_relu'], out['rois'].view(-1, 5), channels_last_out=True)
    assert np.array_equal(_np(pooled), pooled_o)
    # A6/A7: head on those pooled features
    pn = {k: v.numpy() for k, v in p.items()}
    if relation:
        r = OR.relation_head(pooled_o, rois, pn, return_intermediates=True)
        cs, bp = r['cls_score'], r['bbox_pred']
    else:
        cs, bp, _ = ON.plain_head(pooled_o, pn)
    assert np.abs(_np(out['cls_score'][0]) - cs).max() <= 2e-4 * np.abs(cs).max()
    assert np.abs(_np(out['bbox_pred'][0]) - bp).max() <= 2e-4 * max(np.abs(bp).max(), 1e-3)
    # A9: post-processing of the GPU's probabilities / boxes
    full = np.zeros((rois.shape[0], 8)); full[:, 4:8] = out['pred_boxes'][0].cpu().numpy()
    prob0 = _np(out['cls_prob'][0])
    want = OPP.detections(prob0, full, 81, 1e-3, 0.6, True, 100)
    n = int(out['num_detections'][0])
    flat = np.concatenate([np.hstack((np.full((len(w), 1), c + 1.0), w[:, 4:5], w[:, :4])) for c, w in enumerate(want)])
    got = _np(out['detections'][0, :n])
    assert n == len(flat)
    # (class, score) sequence always matches; boxes too unless two rois tie EXACTLY (rois that
    # quantise to the same pooling bins give bit-identical features in the plain 2FC head; the
    # reference's unstable argsort leaves the order of such ties unspecified)
    np.testing.assert_allclose(got[:, :2], flat[:, :2].astype(np.float32), rtol=1e-5)
    uniq = np.array([np.sum(np.isclose(flat[:, 1], sc, rtol=0, atol=0) & (flat[:, 0] == c)) == 1 for c, sc in flat[:, :2]])
    has_tie = any(len(np.unique(prob0[:, j])) < len(prob0) for j in range(1, 81))
    if not has_tie:
        np.testing.assert_allclose(got, flat.astype(np.float32), rtol=1e-5)
    else:
        # co-quantised rois = two IDENTICAL probability rows (all 81 classes); a single equal fp32 value in one class among
        # 100 rois x 80 classes is a coincidence (~2 M representable values in the spread of one class, ~4e5 pairs: it happens on
        # some boxes / convolution algorithms) and only makes the order of those two entries unspecified
        row_tie = len(np.unique(prob0, axis=0)) < len(prob0)
        assert not (relation and row_tie), "the relation head should separate co-quantised rois"
        np.testing.assert_allclose(got[uniq][:, 1], flat[uniq][:, 1].astype(np.float32), rtol=1e-5)


def test_detector_bf16_batch_runs_and_is_close(rn):
    """bf16 throughput path at a reduced image size: same graph, finite outputs, rois valid."""
    ops, backbone, detector = rn
    H, W = 192, 256
    p = backbone.init_params(seed=4)
    g = torch.Generator().manual_seed(6)
    data = torch.randn(2, 3, H, W, generator=g)
    im_info = torch.tensor([[H, W, 1.0], [H, W, 1.0]])
    cfg = detector.Config(); cfg.rpn_post_nms_top_n = 64
    out = detector.Detector(p, dtype=torch.bfloat16, im_hw=(H, W), cfg=cfg).forward(data.cuda(), im_info.cuda())
    assert torch.isfinite(out['cls_prob']).all() and torch.isfinite(out['pred_boxes']).all()
    assert out['rois'].shape == (2, 64, 5) and (out['rois'][1, :, 0] == 1).all()
    assert (out['num_detections'] > 0).all()


@pytest.mark.parametrize('n,in_line', [(2, False), (3, True)])
def test_batches_in_flight_give_each_batch_its_own_forward_results(rn, n, in_line):
    """detector.InFlight: n captured steps (n detector instances, n resident batches; in_line: RPN branch on the step's own stream) replayed round-robin on n streams.  Every
    replay must give exactly what an eager forward of that instance gives on the slot's CURRENT input -- also when the other slot's
    replay is in flight beside it, and after the slot's input tensor has been overwritten in place."""
    ops, backbone, detector = rn
    H, W = 192, 256
    p = backbone.init_params(seed=4)
    g = torch.Generator().manual_seed(16)
    im_info = torch.tensor([[H, W, 1.0], [H, W, 1.0]]).cuda()
    cfg = detector.Config(); cfg.rpn_post_nms_top_n = 64
    dets = [detector.Detector(p, dtype=torch.bfloat16, im_hw=(H, W), cfg=cfg) for _ in range(n)]
    for d in dets:
        d.overlap_rpn = not in_line
    datas = [torch.randn(2, 3, H, W, generator=g).cuda() for _ in range(n)]
    keys = ('rois', 'cls_prob', 'pred_boxes', 'num_detections', 'det_boxes', 'det_scores', 'det_classes')
    with torch.no_grad():
        fl = detector.InFlight([lambda i=i: dets[i].forward(datas[i], im_info) for i in range(n)])
        assert len(fl) == n and fl.streams[0] != fl.streams[1]
        for rnd in range(3):
            if rnd == 2:                       # new batches into the resident input tensors
                for d in datas:
                    d.copy_(torch.randn(d.shape, generator=g).cuda())
                torch.cuda.synchronize()
            ref = [{k: v.clone() for k, v in dets[i].forward(datas[i], im_info).items() if k in keys} for i in range(n)]
            torch.cuda.synchronize()
            slots = [fl.submit() for _ in range(2 * n)]        # every slot twice, nothing waited for in between
            assert slots == list(range(n)) * 2
            for i in range(n):
                out = fl.result(i)
                for k in ref[i]:
                    assert torch.equal(out[k], ref[i][k]), (rnd, i, k)
        assert not torch.equal(fl.result(0)['rois'], fl.result(1)['rois'])          # (two different batches)


@pytest.mark.parametrize('cin,cout,k,stride,dil,hw', [(64, 64, 1, 1, 1, (37, 50)), (256, 128, 1, 2, 1, (38, 51)),
                                                      (128, 128, 3, 1, 1, (19, 33)), (512, 512, 3, 1, 2, (13, 21)),
                                                      (1024, 72, 1, 1, 1, (12, 16)), (64, 256, 1, 1, 1, (75, 64))])
def test_conv2d_nhwc_implicit_gemm(rn, cin, cout, k, stride, dil, hw):
    """Implicit-GEMM NHWC convolution (+bias, +residual, ReLU) vs torch conv2d in float64."""
    ops, _, _ = rn
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(cin + cout + k)
    B, (H, W) = 3, hw
    x = torch.randn(B, H, W, cin, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).cuda().to(torch.bfloat16)
    b = torch.randn(cout, generator=g).cuda()
    pad = dil * (k // 2)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=stride, padding=pad, dilation=dil)
    res = torch.randn(ref.shape, generator=g).cuda().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
    want = torch.relu(ref.permute(0, 2, 3, 1) + res.double())
    got = ops.conv2d_nhwc(x, ops.pack_conv_weight(w), b, ksize=k, stride=stride, pad=pad, dil=dil, relu=True, resid=res)
    assert got.shape == want.shape
    err = (got.double() - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-2, err
    got32 = ops.conv2d_nhwc(x, ops.pack_conv_weight(w), b, ksize=k, stride=stride, pad=pad, dil=dil, out_dtype=torch.float32)
    err32 = (got32.double() - ref.permute(0, 2, 3, 1)).abs().max().item() / ref.abs().max().item()
    assert err32 < 2e-5 * (cin * k * k) ** 0.5, err32


def test_backbone_hip_matches_library_path(rn):
    """bf16 backbone on the implicit-GEMM kernels vs the same folded weights through MIOpen."""
    ops, backbone, _ = rn
    p = backbone.init_params(seed=8)
    g = torch.Generator().manual_seed(9)
    data = torch.randn(2, 3, 160, 224, generator=g).cuda()
    a = backbone.Backbone(p, torch.bfloat16, impl='hip').forward(data)
    b = backbone.Backbone(p, torch.float32, impl='miopen').forward(data)
    for k in ('conv4', 'conv5', 'conv_new_1_relu', 'rpn_cls_score', 'rpn_bbox_pred'):
        assert a[k].shape == b[k].shape, k
        err = (a[k].float() - b[k]).abs().max().item() / b[k].abs().max().item()
        assert err < 6e-2, (k, err)          # ~100 bf16 layers deep


def test_detector_learn_nms_runs(rn):
    """config-3 inference graph: relation head + learn-NMS head, bf16, batch of 2."""
    ops, backbone, detector = rn
    H, W = 192, 256
    p = backbone.init_params(seed=4)
    g = torch.Generator().manual_seed(6)
    data = torch.randn(2, 3, H, W, generator=g)
    im_info = torch.tensor([[H, W, 1.0], [H, W, 1.0]])
    cfg = detector.Config(); cfg.rpn_post_nms_top_n = 128; cfg.learn_nms = True; cfg.first_n = 50
    out = detector.Detector(p, dtype=torch.bfloat16, im_hw=(H, W), cfg=cfg).forward(data.cuda(), im_info.cuda())
    assert out['nms_final_score'].shape == (2, 50, 80) and out['sorted_bbox'].shape == (2, 50, 80, 4)
    assert torch.isfinite(out['nms_multi_score']).all() and (out['nms_multi_score'] >= 0).all()
    fin = out['nms_final_score']
    want = (fin > 1e-3).flatten(1).sum(1).clamp(max=100)
    assert (out['num_detections'] >= want).all()          # >= : ties at the 100th score are all kept
    s = out['sorted_score']
    assert (s[:, :-1] >= s[:, 1:]).all()                  # ranks are in descending score order


def test_stem_conv7_matches_torch(rn):
    """conv1 7x7/2 pad 3 (Cin = 3) on the MFMA kernel via the padded NHWC4 repack vs torch conv2d."""
    ops, _, _ = rn
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(21)
    for (H, W) in ((37, 52), (120, 200)):
        x = torch.randn(2, 3, H, W, generator=g).cuda()
        w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).cuda()
        b = torch.randn(64, generator=g).cuda()
        xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
        want = torch.relu(F.conv2d(xb.double(), wb.double(), b.double(), stride=2, padding=3)).permute(0, 2, 3, 1)
        got = ops.stem_conv7(x, ops.pack_stem_weight(w), b, relu=True)
        assert got.shape == want.shape
        assert (got.double() - want).abs().max().item() <= 1e-2 * want.abs().max().item()
        pooled = ops.stem_bias_relu_pool(got, torch.zeros(64, device='cuda'))
        ref = F.max_pool2d(got.permute(0, 3, 1, 2).float(), 3, 2, 0, ceil_mode=True).permute(0, 2, 3, 1)
        assert torch.equal(pooled.float(), ref)


def test_detector_bf16_fullsize_stagewise(rn):
    """The BENCHMARKED configuration (BASELINE configs[1]): 600x1000 images, 300 rois, bf16, batch > 1, relation head
    + soft-NMS -- every stage against the oracle on the same inputs (teacher forced), at the sizes at which the
    256x256 conv tiles, the XCD swizzle and the LDS attention kernel are the ones that run."""
    ops, backbone, detector = rn
    H, W, B = 600, 1000, 2
    p = backbone.init_params(seed=1)
    g = torch.Generator().manual_seed(11)
    for k in ('cls_score_weight', 'bbox_pred_weight'):
        p[k] = torch.randn(p[k].shape, generator=g) * 0.05
    p['conv_new_1_bias'] = torch.rand(256, generator=g) * 0.1 + 0.05
    data = torch.randn(B, 3, H, W, generator=g)
    im_info = torch.tensor([[H, W, 1.0]] * B)
    det = detector.Detector(p, dtype=torch.bfloat16, im_hw=(H, W))
    f = det.backbone.forward(data.cuda())
    out = det.forward(data.cuda(), im_info.cuda())
    assert out['rois'].shape == (B, 300, 5)
    # A1/A2: bf16 HIP backbone (~100 layers) vs the float32 torch-CPU restatement
    with torch.no_grad():
        c4, c5 = ON.backbone(data, p)
        cls, box, feat = ON.rpn_and_feat(c4, c5, p)
    for name, got, want in (('conv4', f['conv4'], c4), ('conv5', f['conv5'], c5), ('conv_new_1', f['conv_new_1_relu'], feat),
                            ('rpn_cls_score', f['rpn_cls_score'], cls), ('rpn_bbox_pred', f['rpn_bbox_pred'], box)):
        d_ = got.float().cpu() - want
        rel_l2 = (d_.norm() / want.norm()).item()
        rel_max = d_.abs().max().item() / want.abs().max().item()
        assert rel_l2 < 2.5e-2 and rel_max < 8e-2, (name, rel_l2, rel_max)
    pn = {k: v.numpy() for k, v in p.items()}
    for b in range(B):
        # A3: proposal on the GPU's own RPN maps -> identical roi rows (discrete decisions on identical inputs)
        prob = ON.rpn_softmax(_np(f['rpn_cls_score'][b:b + 1]))
        rois_o, _, dbg = OP.proposal(prob, _np(f['rpn_bbox_pred'][b:b + 1]), im_info[b:b + 1].numpy(), 16, det.cfg.anchor_scales,
                                     det.cfg.anchor_ratios, 6000, 300, 0.7, 0, return_debug=True)
        rois = _np(out['rois'][b])
        assert (rois[:, 0] == b).all()
        n_same = int((np.abs(rois[:, 1:] - rois_o[:, 1:]).max(axis=1) == 0).sum())
        assert n_same == 300 or np.abs(rois[:, 1:] - rois_o[:, 1:]).max() < 1e-3, (b, n_same)
        # A4: ROIPooling of the GPU feature map -> bit exact
        r0 = rois.copy(); r0[:, 0] = 0
        pooled_o = ORP.roi_pooling(_np(f['conv_new_1_relu'][b:b + 1]), r0)
        pooled = ops.roi_pool(f['conv_new_1_relu'], out['rois'][b].contiguous(), channels_last_out=True)
        assert np.array_equal(_np(pooled), pooled_o)
        # A6/A7: bf16 2FC + relation head vs the float32 oracle on the same pooled features
        r = OR.relation_head(pooled_o, r0, pn, return_intermediates=True)
        for key in ('cls_score', 'bbox_pred'):
            want = r[key]
            err = np.abs(_np(out[key][b]) - want).max() / max(np.abs(want).max(), 1e-3)
            assert err <= 3e-2, (key, b, err)
        # A9: post-processing of the GPU's probabilities / boxes (float64, exact up to exp ulps)
        full = np.zeros((300, 8)); full[:, 4:8] = out['pred_boxes'][b].cpu().numpy()
        prob0 = _np(out['cls_prob'][b])
        want = OPP.detections(prob0, full, 81, 1e-3, 0.6, True, 100)
        n = int(out['num_detections'][b])
        flat = np.concatenate([np.hstack((np.full((len(w_), 1), c + 1.0), w_[:, 4:5], w_[:, :4])) for c, w_ in enumerate(want)])
        assert n == len(flat), (n, len(flat))
        np.testing.assert_allclose(_np(out['detections'][b, :n]), flat.astype(np.float32), rtol=1e-5)
    # the same comparison as bench.py's `parity` block reports it
    from oracle import parity as OPAR
    rep = OPAR.stagewise(det, data.cuda(), im_info.cuda(), p, images=[1])
    w = rep['worst']
    assert w['proposal_rows_identical'] == 300 and w['roi_pool_mismatches'] == 0 and w['detections_all_matched'], rep
    assert w['cls_prob_max_abs_err'] < 2e-2 and w['bbox_pred_max_rel_err'] < 3e-2, rep


def test_benched_trunk_configuration_in_network(rn):
    """The configuration that produces the headline number, inside the network: with the chain kernels forced on for every
    stage (ops.CHAIN_MIN_PIXELS lowered: at 54 images of 600x1000 they run by themselves) and B >= 4 (RPN head + proposal on
    the side stream, fork / join inside Backbone.forward), ONE Detector.forward call is compared with the float32 oracle:
    backbone maps directly, every later stage teacher forced (oracle/parity.py:stagewise, what bench.py prints)."""
    ops, backbone, detector = rn
    from oracle import parity as OPAR
    H, W, B = 320, 480, 4
    p = backbone.init_params(seed=1)
    g = torch.Generator().manual_seed(13)
    for k in ('cls_score_weight', 'bbox_pred_weight'):
        p[k] = torch.randn(p[k].shape, generator=g) * 0.05
    p['conv_new_1_bias'] = torch.rand(256, generator=g) * 0.1 + 0.05
    data = torch.randn(B, 3, H, W, generator=g).cuda()
    im_info = torch.tensor([[H, W, 1.0]] * B).cuda()
    det = detector.Detector(p, dtype=torch.bfloat16, im_hw=(H, W))
    saved = dict(ops.CHAIN_MIN_PIXELS)
    try:
        ops.CHAIN_MIN_PIXELS.update({k: 0 for k in ops.CHAIN_MIN_PIXELS})
        rep = OPAR.stagewise(det, data, im_info, p, images=[0, 3])
        out = det.forward(data, im_info, keep_features=True)
    finally:
        ops.CHAIN_MIN_PIXELS.update(saved)
    units = rep['chain_kernel_units']
    assert len(units) == 33 and {'2a', '3b3', '4b10', '4b22', '5a', '5c'} <= set(units), units       # every residual unit of the trunk
    assert rep['rpn_side_stream'] and rep['same_forward_call']
    w = rep['worst']
    for k in ('conv4', 'conv5', 'conv_new_1_relu', 'rpn_cls_score', 'rpn_bbox_pred'):
        assert w['backbone_%s_rel_l2' % k] < 2.5e-2 and w['backbone_%s_rel_max' % k] < 8e-2, (k, rep['backbone'][k])
    assert w['proposal_rows_identical'] == 300 and w['roi_pool_mismatches'] == 0 and w['detections_all_matched'], rep
    assert w['cls_score_max_rel_err'] < 3e-2 and w['bbox_pred_max_rel_err'] < 3e-2, rep
    assert w['attention_1_max_rel_err'] < 3e-2 and w['attention_2_max_rel_err'] < 3e-2, rep
    # the chain kernels repeat the products of the tiled convolution kernels they replace in the same order; res2a's fused
    # projection shortcut keeps the shortcut in fp32 (one bf16 rounding fewer), so the two trunks differ by bf16 rounding noise
    ref = det.forward(data, im_info, keep_features=True)
    assert ref['features'] is not out['features'] and len(det.backbone.last_chain_units) < 33
    for k in ('conv4', 'conv5'):
        a, b = out['features'][k].float(), ref['features'][k].float()
        d = (a - b).abs().max().item()
        assert ((a - b).norm() / b.norm()).item() < 1e-2 and d <= 5e-2 * b.abs().max().item(), (k, d)


@pytest.mark.parametrize('shape', [(2, 600, 1000), (1, 37, 52), (3, 121, 200), (1, 800, 1024)])
def test_fused_stem_equals_the_three_launch_stem(rn, shape):
    """conv1 7x7/2 + bias + ReLU + pool1 in one kernel (no conv map in HBM) == stem_conv7 + stem_bias_relu_pool bit for bit
    (same k order of the products), incl. image sizes whose last pooling window is clipped; fp32 and bf16 input."""
    ops, _, _ = rn
    B, H, W = shape
    g = torch.Generator().manual_seed(31 + H)
    x = torch.randn(B, 3, H, W, generator=g).cuda()
    w = ops.pack_stem_weight(torch.randn(64, 3, 7, 7, generator=g) * 0.1)
    b = torch.randn(64, generator=g).cuda()
    want = ops.stem_bias_relu_pool(ops.stem_conv7(x, w, b, relu=True), torch.zeros(64, device='cuda'))
    got = ops.stem_fused(x, w, b)
    assert got.shape == want.shape and torch.equal(got, want)
    got16 = ops.stem_fused(x.to(torch.bfloat16), w, b)
    assert torch.equal(got16, want)                     # the fp32 image is rounded to bf16 on the way in either way
