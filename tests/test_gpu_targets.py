"""GPU parity of the training-target operators (A10) against golden vectors produced by the reference's
own sample_rois_v2 / BoxAnnotatorOHEMOperator / NmsMultiTargetOp, and against the oracle at full size."""
import numpy as np
import pytest
import torch

import cases
from oracle import targets as OT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rn():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops, operator_py, lib
    lib.load()
    return ops, operator_py


def _d(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def test_targets_match_reference_goldens(rn, golden):
    ops, operator_py = rn
    g = golden['targets']
    rois, gt_boxes, cls_score, bbox_pred = cases.targets_case(90, 7, 61)
    # mirrored CustomOp protocol, the reference's call sites (symbols/..._learn_nms.py:293-307, 367-371, 536-538)
    r, lab, bt, bw = operator_py.Custom(rois=_d(rois), gt_boxes=_d(gt_boxes), op_type='proposal_target',
                                        num_classes=2, batch_images=1, batch_rois=-1, fg_fraction=0.25)
    assert np.array_equal(r.cpu().numpy(), g['pt/rois']) and np.array_equal(lab.cpu().numpy(), g['pt/label'])
    assert np.array_equal(bw.cpu().numpy(), g['pt/bbox_weight'])
    np.testing.assert_allclose(bt.cpu().numpy(), g['pt/bbox_target'], rtol=3e-7, atol=1e-6)   # numpy f32 log vs exact
    want_bt = OT.proposal_target(rois, gt_boxes)[2]
    assert np.array_equal(bt.cpu().numpy(), want_bt)                                            # oracle: bit exact
    lo, wo = operator_py.Custom(cls_score=_d(cls_score), bbox_pred=_d(bbox_pred), labels=_d(g['pt/label']),
                                bbox_targets=_d(g['pt/bbox_target']), bbox_weights=_d(g['pt/bbox_weight']),
                                op_type='BoxAnnotatorOHEM', num_classes=81, num_reg_classes=2, roi_per_img=32)
    assert np.array_equal(lo.cpu().numpy(), g['ohem/labels']) and np.array_equal(wo.cpu().numpy(), g['ohem/bbox_weights'])
    bbox, gt_box, score = cases.nms_target_case(40, 6, 9, 62)
    t = operator_py.Custom(bbox=_d(bbox), gt_bbox=_d(gt_box), score=_d(score), op_type='nms_multi_target',
                           target_thresh=np.array([0.5, 0.6, 0.7, 0.8, 0.9]))
    assert np.array_equal(t.cpu().numpy(), g['nmt/target'])


def test_targets_full_size_batched_vs_oracle(rn):
    """N = 300 proposals + 8..20 gt boxes, 81 classes, first_n = 100, batch of 3 images with ragged gt counts."""
    ops, _ = rn
    B, N, Gmax = 3, 300, 20
    ng = [20, 8, 13]
    cs = [cases.targets_case(N, ng[b], 70 + b) for b in range(B)]
    gt = np.zeros((B, Gmax, 5), np.float32)
    for b in range(B):
        gt[b, :ng[b]] = cs[b][1]
    rois = np.stack([c[0] for c in cs])
    num_gt = torch.tensor(ng, dtype=torch.int32).cuda()
    r, lab, bt, bw = ops.proposal_target(_d(rois), _d(gt), num_gt)
    rng = np.random.default_rng(5)
    cls_score = rng.normal(0, 2, (B, N + Gmax, 81)).astype(np.float32)
    bbox_pred = rng.normal(0, 0.5, (B, N + Gmax, 8)).astype(np.float32)
    lo, wo, loss = ops.box_annotator_ohem(_d(cls_score), _d(bbox_pred), lab, bt, bw, 128, want_loss=True)
    for b in range(B):
        wr, wl, wt, ww = OT.proposal_target(cs[b][0], cs[b][1])
        k = N + ng[b]
        wr = wr.copy(); wr[N:, 0] = b        # appended gt rows carry their image's index (the reference runs one image per
        #                                      device and writes 0 there, proposal_target.py:79; ROI pooling of a batch needs b)
        assert np.array_equal(r[b, :k].cpu().numpy(), wr) and np.array_equal(lab[b, :k].cpu().numpy(), wl)
        assert np.array_equal(bt[b, :k].cpu().numpy(), wt) and np.array_equal(bw[b, :k].cpu().numpy(), ww)
        assert (lab[b, k:] == -1).all() and (bw[b, k:] == 0).all()
        olo, owo, oloss = OT.box_annotator_ohem(cls_score[b, :k], bbox_pred[b, :k], wl, wt, ww, 128)
        np.testing.assert_allclose(loss[b, :k].cpu().numpy(), oloss, rtol=2e-6)
        assert np.array_equal(lo[b, :k].cpu().numpy(), olo) and np.array_equal(wo[b, :k].cpu().numpy(), owo)
        assert (lo[b, k:] == -1).all()
    cs2 = [cases.nms_target_case(100, 80, ng[b], 80 + b) for b in range(B)]
    gt2 = np.zeros((B, Gmax, 5), np.float32)
    for b in range(B):
        gt2[b, :ng[b]] = cs2[b][1][0]
    t = ops.nms_multi_target(_d(np.stack([c[0] for c in cs2])), _d(gt2), _d(np.stack([c[2] for c in cs2])), num_gt)
    for b in range(B):
        want = OT.nms_multi_target(cs2[b][0], cs2[b][1], cs2[b][2])
        assert np.array_equal(t[b].cpu().numpy(), want) and want.sum() > 0


def test_assign_anchor_on_device_matches_oracle_and_reference_golden(rn):
    """relnet_assign_anchor (lib/rpn/rpn.py:80-244 on the device, B images per launch) against oracle/anchors.py -- itself
    held bit-for-bit to the reference's own function (tests/golden/rpn_targets.npz) -- with the kernel's hash sampler:
    identical labels (before AND after the random fg / bg sub-sampling), identical weights, targets to 1e-6; images with
    6, 40 and 0 gt boxes in one launch, one of them smaller than the padded batch frame."""
    import os
    ops, _ = rn
    from oracle import anchors as OA
    from relnet_amd.operator_py.proposal import generate_anchors
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rpn_targets.npz'))
    gts = [g['six_gt/gt'], g['crowded/gt'], np.zeros((0, 5), np.float32), g['six_gt/gt'][:3] * 0.5]
    hw = [(600, 1000), (600, 1000), (600, 1000), (300, 520)]               # the last image covers part of the frame only
    B, G = len(gts), max(len(x) for x in gts)
    pad = np.zeros((B, G, 5), np.float32)
    for b, x in enumerate(gts):
        pad[b, :len(x)] = x
    num_gt = torch.tensor([len(x) for x in gts], dtype=torch.int32).cuda()
    im_info = torch.tensor([[h, w, 1.0] for h, w in hw]).cuda()
    base = generate_anchors(16, (0.5, 1, 2), (4, 8, 16, 32))
    seed = 20260924
    L, T, W, Lall = ops.assign_anchor(_d(pad), num_gt, im_info, base, (38, 63), seed=seed, want_all=True)
    for b in range(B):
        wl, wt, ww, wall = OA.assign_anchor((38, 63), gts[b], hw[b], sampler='hash', seed=seed, image_index=b, return_all=True)
        assert np.array_equal(Lall[b].cpu().numpy(), wall), b
        assert np.array_equal(L[b].cpu().numpy(), wl), (b, int((L[b].cpu().numpy() != wl).sum()))
        assert np.array_equal(W[b].cpu().numpy(), ww), b
        assert np.abs(T[b].cpu().numpy() - wt).max() <= 1e-6, b
        nfg, nbg = int((wl == 1).sum()), int((wl == 0).sum())
        assert nfg <= 128 and nfg + nbg == 256
    # the deterministic part equals the reference run itself (golden labels are a random subset of these candidates)
    for b, name in ((0, 'six_gt'), (1, 'crowded')):
        gl = g[name + '/label'][0]
        la = Lall[b].cpu().numpy()
        assert (la[gl == 1] == 1).all() and (la[gl == 0] == 0).all()
        assert np.abs(T[b].cpu().numpy() - g[name + '/bbox_target'][0]).max() <= 2e-6
    # seed_dev: a device step counter shifts the seed (what a captured training graph advances between replays)
    ctr = torch.tensor([5], dtype=torch.int64).cuda()
    L5 = ops.assign_anchor(_d(pad), num_gt, im_info, base, (38, 63), seed=seed - 5, seed_dev=ctr)[0]
    assert torch.equal(L5, L)
    L6 = ops.assign_anchor(_d(pad), num_gt, im_info, base, (38, 63), seed=seed + 1)[0]
    assert not torch.equal(L6[1], L[1]) and int((L6[1] == 1).sum()) == 128
