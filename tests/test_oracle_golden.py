"""The CPU oracle against golden vectors produced by the reference's own python
(tests/golden/gen_golden.py) plus hand-checkable known answers."""
import numpy as np
import pytest

import cases
from oracle import boxes as OB
from oracle import nms as ON
from oracle import relation as OR
from oracle import learn_nms as OL


def test_anchor_known_answers():
    # classic py-faster-rcnn table (ratios .5,1,2 x scales 8,16,32)
    want = np.array([[-84, -40, 99, 55], [-176, -88, 191, 103], [-360, -184, 375, 199],
                     [-56, -56, 71, 71], [-120, -120, 135, 135], [-248, -248, 263, 263],
                     [-36, -80, 51, 95], [-80, -168, 95, 183], [-168, -344, 183, 359]], dtype=np.float64)
    assert np.array_equal(OB.generate_anchors(), want)
    cfg = OB.generate_anchors(16, (0.5, 1, 2), (4, 8, 16, 32))
    assert cfg.shape == (12, 4)
    assert np.array_equal(cfg[[0, 4, 8]], [[-38, -16, 53, 31], [-24, -24, 39, 39], [-14, -36, 29, 51]])


def test_anchors_match_reference(golden):
    g = golden['boxes']
    assert np.array_equal(OB.generate_anchors(), g['anchors_default'])
    assert np.array_equal(OB.generate_anchors(16, (0.5, 1, 2), (4, 8, 16, 32)), g['anchors_cfg'])


def test_box_decode_clip_transform_match_reference(golden):
    g = golden['boxes']
    # float64 deltas: every op is IEEE float64 in both -> bit exact
    p64 = OB.bbox_pred(g['pred_boxes_in'], g['pred_deltas_in'].astype(np.float64))
    assert p64.dtype == np.float64 and np.array_equal(p64, g['pred_out_f64deltas'])
    # float32 deltas: the reference calls numpy's float32 exp (a SIMD approximation that
    # differs between numpy builds); the oracle pins the correctly rounded fp32 exp, so
    # agreement is to 1 ulp of that exp (relative 2^-23 on the box extent).
    pred = OB.bbox_pred(g['pred_boxes_in'], g['pred_deltas_in'])
    ext = np.abs(g['pred_out'][:, 2::4] - g['pred_out'][:, 0::4]).max()
    assert np.abs(pred - g['pred_out']).max() <= 2.0 ** -23 * ext
    pred = g['pred_out']
    assert np.array_equal(OB.clip_boxes(pred, (cases.IM_H, cases.IM_W)), g['clip_out'])
    t = OB.bbox_transform(g['pred_boxes_in'], g['transform_gt_in'])
    assert np.array_equal(t, g['transform_out'])
    ov = OB.bbox_overlaps(g['pred_boxes_in'][:20], g['transform_gt_in'][:15])
    assert np.array_equal(ov, g['overlaps_out'])


def test_iou_micro_cases():
    a = np.array([0, 0, 9, 9], np.float32)
    b = np.array([[0, 0, 9, 9], [5, 0, 14, 9], [10, 10, 19, 19], [9, 9, 18, 18]], np.float32)
    iou = ON.iou_f32(a, b)
    # +1 pixel extents: areas 100; overlaps 100, 50, 0, 1
    np.testing.assert_allclose(iou, [1.0, 50 / 150., 0.0, 1 / 199.], rtol=1e-6)


def test_nms_and_softnms_match_reference(golden):
    g = golden['nms']
    for name, (n, seed) in {'a': (300, 31), 'b': (1000, 32)}.items():
        dets = cases.dets_case(n, seed)
        for t in (0.3, 0.5, 0.7):
            want = g['nms_%s_%d' % (name, int(t * 10))]
            assert np.array_equal(np.asarray(ON.py_nms(dets, t)), want)
            # the CUDA-order variant (fp32 IoU, strict >) selects the same boxes here
            assert np.array_equal(np.asarray(ON.gpu_nms(dets, t)), want)
        assert np.array_equal(ON.soft_nms(dets, 0.6, -1), g['softnms_%s' % name])
        assert np.array_equal(ON.soft_nms(dets, 0.6, 100), g['softnms_%s_max100' % name])


def test_nms_three_box_order():
    dets = np.array([[0, 0, 10, 10, 0.5], [1, 1, 11, 11, 0.9], [50, 50, 60, 60, 0.1]], np.float32)
    assert ON.gpu_nms(dets, 0.5) == [1, 2]
    assert ON.py_nms(dets, 0.5) == [1, 2]


def test_relation_module_matches_reference_graph(golden):
    g = golden['relation']
    for name, (n, m, seed, std) in cases.RELATION_CASES.items():
        boxes, feat, p = cases.relation_case(n, m, seed, std)
        pm = OR.position_matrix(boxes, m)
        pe = OR.position_embedding(pm)
        assert np.array_equal(pm, g[name + '/position_matrix'])
        assert np.array_equal(pe, g[name + '/position_embedding'])
        r = OR.relation_module(feat, pe, p, index=1, nongt_dim=m, return_intermediates=True)
        np.testing.assert_allclose(r['logits'], g[name + '/logits'], rtol=0, atol=2e-6)
        np.testing.assert_allclose(r['softmax'], g[name + '/softmax'], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(r['output'], g[name + '/output'], rtol=1e-5, atol=1e-6)


def test_logit_conditioning_is_what_design_md_says():
    """DESIGN.md "tolerances": the fp32 graph multiplies the log-geometry by 100 before
    sin/cos and takes log(relu(.)) afterwards, so a 1-ulp change of the position matrix
    moves some logits by far more than 1e-4 although the module output barely moves.
    Hence the HIP path reproduces the position matrix bit-exactly (same IEEE ops,
    correctly rounded log) and the logit bound is conditioning-aware."""
    name, (n, m, seed, std) = next(iter(cases.RELATION_CASES.items()))
    boxes, feat, p = cases.relation_case(n, m, seed, std)
    pm = OR.position_matrix(boxes, m)
    r = OR.relation_module(feat, OR.position_embedding(pm), p, 1, m, return_intermediates=True)
    pm_ulp = np.nextafter(pm, np.float32(100))
    r2 = OR.relation_module(feat, OR.position_embedding(pm_ulp), p, 1, m, return_intermediates=True)
    dl = np.abs(r['logits'] - r2['logits'])
    assert dl.max() > 1e-3                       # ill-conditioned entries exist
    assert np.abs(r['output'] - r2['output']).max() < 1e-4
    # a 1-ulp change of the sin/cos values, by contrast, is harmless where G >= 1e-3
    r3 = OR.relation_module(feat, np.nextafter(OR.position_embedding(pm), np.float32(2)), p, 1, m,
                            return_intermediates=True)
    well = r['aff_weight'] >= 1e-3
    assert np.abs(r['logits'] - r3['logits'])[well].max() < 1e-4


def test_learn_nms_matches_reference_operator(golden):
    g = golden['learn_nms']
    for name, (n, c, first_n, seed) in cases.LEARN_NMS_CASES.items():
        cls_score, bbox_pred, rois, im_info, feat, p = cases.learn_nms_case(n, c, seed)
        multi, sbox, sscore = OL.learn_nms(cls_score, bbox_pred, rois, im_info, feat, p,
                                           num_fg_classes=c, first_n=first_n, nongt_dim=n)
        np.testing.assert_allclose(sscore, g[name + '/sorted_score'], rtol=1e-6, atol=0)
        np.testing.assert_allclose(sbox, g[name + '/sorted_bbox'], rtol=0, atol=1e-4)
        np.testing.assert_allclose(multi, g[name + '/nms_multi_score'], rtol=2e-5, atol=1e-7)


def test_training_targets_match_reference(golden):
    """proposal_target (sample_rois_v2), BoxAnnotatorOHEM, nms_multi_target vs the reference's own code."""
    from oracle import targets as OT
    g = golden['targets']
    rois, gt_boxes, cls_score, bbox_pred = cases.targets_case(90, 7, 61)
    r, lab, bt, bw = OT.proposal_target(rois, gt_boxes)
    assert np.array_equal(r, g['pt/rois']) and np.array_equal(lab, g['pt/label'])
    assert (lab > 0).sum() >= 10 and (lab == 0).sum() >= 10
    assert np.array_equal(bw, g['pt/bbox_weight'])
    # numpy's float32 log in the reference run vs the correctly rounded log of the oracle: 1 ulp
    np.testing.assert_allclose(bt, g['pt/bbox_target'], rtol=3e-7, atol=1e-6)
    lo, wo, _ = OT.box_annotator_ohem(cls_score, bbox_pred, g['pt/label'], g['pt/bbox_target'], g['pt/bbox_weight'], 32)
    assert np.array_equal(lo, g['ohem/labels']) and np.array_equal(wo, g['ohem/bbox_weights'])
    assert (lo >= 0).sum() == 32
    bbox, gt_box, score = cases.nms_target_case(40, 6, 9, 62)
    t = OT.nms_multi_target(bbox, gt_box, score)
    assert np.array_equal(t, g['nmt/target']) and t.sum() > 0


def test_torch_relation_matches_numpy_oracle():
    """oracle/relation_torch.py (the autograd checker of the backward kernels) computes the same forward as the
    numpy oracle that is pinned to the reference's Python above."""
    import torch
    from oracle import relation as OR, relation_torch as ORT
    for n, m, seed, std in ((40, 32, 12, 0.05), (48, 48, 11, 0.01)):
        boxes, feat, p = cases.relation_case(n, m, seed, std)
        pe = OR.position_embedding(OR.position_matrix(boxes, m))
        want = OR.relation_module(feat, pe, p, 1, m)
        pt = {k: torch.as_tensor(v.astype(np.float64)) for k, v in p.items()}
        got = ORT.relation_module(torch.as_tensor(feat.astype(np.float64)), boxes, pt, 1, m).numpy()
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


def test_fpn_roi_dispatch_matches_reference_loader(golden):
    """oracle/fpn.py:roi_dispatch vs the reference's own loader code (core/rcnn.py:get_rcnn_testbatch with
    ROIDispatch), including boxes exactly on the level boundaries and the all-zero dummy roi of an empty level."""
    from oracle import fpn as OF
    g = golden['fpn']
    for name in ('all_levels', 'empty_level0'):
        boxes = g[name + '/boxes']
        assert np.array_equal(boxes, cases.fpn_proposals(400, 51) if name == 'all_levels' else boxes)
        rois, level, perm, counts = OF.roi_dispatch(boxes, dummy_for_empty=True)
        off = 0
        for l in range(4):
            want = g['%s/rois_%d' % (name, l)]
            got = rois[off:off + len(want)]
            assert np.array_equal(got.astype(np.float64), want), (name, l)
            assert (level[off:off + len(want)] == l).all()
            off += len(want)
        assert off == len(rois)
    assert OF.roi_dispatch(g['empty_level0/boxes'])[3][0] == 0           # level 0 really is empty there


def test_proposal_operator_matches_reference_operator(golden):
    """oracle/proposal.py vs the reference's own ProposalOperator.forward (operator_py/proposal.py:51-168) run under the numpy
    MXNet stand-in: the same 300 / 50 proposals in the same order (scores bit-identical); boxes within 2 ulps of the box size -- numpy's
    float32 exp in the reference's bbox_pred is not correctly rounded, the oracle's is (oracle/boxes.py)."""
    from oracle import proposal as OP
    g = golden['proposal']
    for name, (seed, pre, post) in {'full': (61, 6000, 300), 'small': (62, 600, 50)}.items():
        cls_prob, deltas, im_info = cases.rpn_case(seed)
        rois, scores = OP.proposal(cls_prob, deltas, im_info, 16, (4, 8, 16, 32), (0.5, 1, 2), pre, post, 0.7, 0)
        want = g[name + '/rois']
        assert np.array_equal(scores.reshape(-1), g[name + '/score'].reshape(-1))
        # a 1-ulp exp error in the predicted width / height (up to ~1000 px) moves a corner by up to 2 ulps OF THAT SIZE
        assert rois.shape == want.shape and np.abs(rois - want).max() <= 2 * np.spacing(np.float32(1000.0))
        assert np.array_equal(rois[:, 0], want[:, 0])


@pytest.mark.parametrize('name', ['rel_n300_m300_std01', 'rel_n333_m300_std05'])
def test_relation_module_full_size_matches_reference_run(name):
    """The oracle pinned at the BENCHMARK size (N = M = 300, and the training shape N = 333 / M = 300) by
    tests/golden/relation_large.npz = the reference's own graph code run on the numpy MXNet stand-in
    (gen_golden.py --only-large).  (The N = 1000 case is exercised by the GPU suite; its oracle needs ~2 GB.)"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'relation_large.npz'))
    n, m, seed, std, k = cases.RELATION_LARGE_CASES[name]
    boxes, feat, p = cases.relation_case(n, m, seed, std)
    rows = g[name + '/rows']
    assert np.array_equal(rows, cases.kept_rows(n, k, seed))
    pm = OR.position_matrix(boxes, m)
    assert np.array_equal(pm[rows], g[name + '/position_matrix'])
    r = OR.relation_module(feat, OR.position_embedding(pm), p, index=1, nongt_dim=m, return_intermediates=True)
    np.testing.assert_allclose(r['logits'][rows], g[name + '/logits'], rtol=0, atol=4e-6)
    np.testing.assert_allclose(r['output'][rows], g[name + '/output'], rtol=2e-5, atol=2e-6)


def test_learn_nms_benchmark_shape_matches_reference_operator():
    """300 rois x 80 classes x first_n 100 through the reference's LearnNmsOperator.forward (relation_large.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'relation_large.npz'))
    for name, (n, c, first_n, seed) in cases.LEARN_NMS_LARGE_CASES.items():
        cls_score, bbox_pred, rois, im_info, feat, p = cases.learn_nms_case(n, c, seed)
        multi, sbox, sscore = OL.learn_nms(cls_score, bbox_pred, rois, im_info, feat, p, num_fg_classes=c, first_n=first_n, nongt_dim=n)
        np.testing.assert_allclose(sscore, g[name + '/sorted_score'], rtol=1e-6, atol=0)
        np.testing.assert_allclose(sbox, g[name + '/sorted_bbox'], rtol=0, atol=1e-4)
        np.testing.assert_allclose(multi, g[name + '/nms_multi_score'], rtol=5e-5, atol=2e-7)


def test_learn_nms_fpn_yaml_values_match_reference_operator():
    """The learn-NMS head as the FPN experiment words it (..._rcnn_fpn_relation_learn_nms_8epoch.yaml:141,166-167: 1000 rois,
    FIRST_N 150, LEARN_NMS_CLASS_SCORE_TH 0.05) through the reference's LearnNmsOperator.forward (learn_nms_fpn.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'learn_nms_fpn.npz'))
    for name, (n, c, first_n, seed, th) in cases.LEARN_NMS_FPN_CASES.items():
        cls_score, bbox_pred, rois, im_info, feat, p = cases.learn_nms_fpn_case(n, c, seed)
        multi, sbox, sscore, dbg = OL.learn_nms(cls_score, bbox_pred, rois, im_info, feat, p, num_fg_classes=c, first_n=first_n,
                                                class_thresh=th, nongt_dim=None, return_intermediates=True)
        want = g[name + '/nms_multi_score']
        assert want.shape == (first_n, c, 5)
        # the 0.05 rule drops 18 classes here, the default 0.01 would drop 9: the threshold is exercised
        assert len(dbg['valid']) == 62 and np.array_equal(np.where(want.max(axis=(0, 2)) > 0)[0], dbg['valid'])
        assert (sscore.max(axis=0) < 0.01).sum() == 9
        np.testing.assert_allclose(sscore, g[name + '/sorted_score'], rtol=1e-6, atol=0)
        np.testing.assert_allclose(sbox, g[name + '/sorted_bbox'], rtol=0, atol=1e-4)
        np.testing.assert_allclose(multi, want, rtol=5e-5, atol=2e-7)


def test_assign_anchor_oracle_matches_reference_loader():
    """oracle/anchors.py (sampler='numpy') vs tests/golden/rpn_targets.npz = the output of the reference's own
    lib/rpn/rpn.py:assign_anchor with the same numpy seed: identical labels incl. the random fg / bg sub-sampling, identical
    weights, targets to 2e-6; the hash sampler (the device kernel's definition of the random subset) changes nothing else."""
    from oracle import anchors as OA
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rpn_targets.npz'))
    for name, seed in (('six_gt', 3), ('crowded', 4)):
        L, T, W, Lall = OA.assign_anchor((38, 63), g[name + '/gt'], (600, 1000), seed=seed, return_all=True)
        assert np.array_equal(L, g[name + '/label'][0]) and np.array_equal(W, g[name + '/bbox_weight'][0])
        assert np.abs(T - g[name + '/bbox_target'][0]).max() <= 2e-6
        Lh, Th, Wh, Lall_h = OA.assign_anchor((38, 63), g[name + '/gt'], (600, 1000), sampler='hash', seed=1234, image_index=2, return_all=True)
        assert np.array_equal(Lall_h, Lall) and np.array_equal(Th, T)
        assert (Lh == 1).sum() == (L == 1).sum() and (Lh == 0).sum() == (L == 0).sum() == 256 - (L == 1).sum()
        assert (Lall[Lh == 1] == 1).all() and (Lall[Lh == 0] == 0).all()              # a subset of the candidates
        assert np.array_equal(Wh.reshape(12, 4, 38, 63).sum(1) == 4, Lh.reshape(12, 38, 63) == 1)
    # a different seed draws a different subset of the same size
    L2 = OA.assign_anchor((38, 63), g['crowded/gt'], (600, 1000), sampler='hash', seed=99)[0]
    L3 = OA.assign_anchor((38, 63), g['crowded/gt'], (600, 1000), sampler='hash', seed=100)[0]
    assert (L2 == 1).sum() == (L3 == 1).sum() == 128 and not np.array_equal(L2, L3)


def test_roi_align_oracle_known_answers():
    """oracle/roi_align.py has no reference code to be pinned by (the reference has no ROIAlign): what holds it are the answers that
    follow from the published definition -- affine maps pool to their value at the bin centre, constants to the constant, a one-hot map to
    the bilinear weight, float32 within rounding of float64."""
    from oracle import roi_align as ORA
    H, W, C = 20, 30, 5
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    coef = np.random.default_rng(2).normal(0, 1, (C, 3))
    data = (coef[:, 0, None, None] * yy + coef[:, 1, None, None] * xx + coef[:, 2, None, None])[None]
    rois = np.array([[0, 40, 30, 200, 150], [0, 100.5, 60.25, 130.75, 99.5]], np.float32)
    for aligned in (False, True):
        for sr in (1, 2, 3, 0):
            got = ORA.roi_align(data, rois, (7, 7), 1 / 16.0, sr, aligned, dtype=np.float64)
            off = 0.5 if aligned else 0.0
            for i, roi in enumerate(rois.astype(np.float64)):
                sw, sh = roi[1] / 16 - off, roi[2] / 16 - off
                rw, rh = roi[3] / 16 - off - sw, roi[4] / 16 - off - sh
                if not aligned:
                    rw, rh = max(rw, 1.0), max(rh, 1.0)
                cy = sh + (np.arange(7) + 0.5) * rh / 7
                cx = sw + (np.arange(7) + 0.5) * rw / 7
                want = coef[:, 0, None, None] * cy[None, :, None] + coef[:, 1, None, None] * cx[None, None, :] + coef[:, 2, None, None]
                assert np.abs(got[i] - want).max() <= 1e-10 * np.abs(want).max()
            g32 = ORA.roi_align(data, rois, (7, 7), 1 / 16.0, sr, aligned)
            assert g32.dtype == np.float32 and np.abs(g32 - got).max() <= 2e-5 * np.abs(got).max()
    const = np.full((1, 2, H, W), 3.25)
    assert np.array_equal(ORA.roi_align(const, rois, (3, 3), 1 / 16.0, 2), np.full((2, 2, 3, 3), 3.25, np.float32))
    hot = np.zeros((1, 1, H, W)); hot[0, 0, 10, 20] = 1.0
    roi = np.array([[0, 16 * 19.25, 16 * 9.5, 16 * 20.25, 16 * 10.5]], np.float32)
    assert abs(float(ORA.roi_align(hot, roi, (1, 1), 1 / 16.0, 1)[0, 0, 0, 0]) - 0.75) < 1e-6
    # a box hanging over the border: samples beyond (-1, H) x (-1, W) contribute zero, the divisor stays sr^2
    out = ORA.roi_align(const, np.array([[0, -64, -64, 31, 31]], np.float32), (2, 2), 1 / 16.0, 2, dtype=np.float64)
    assert out[0, 0, 0, 0] == 0.0 and 0 < out[0, 0, 1, 1] <= 3.25
