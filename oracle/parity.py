"""Oracle: stage-by-stage ("teacher forced") comparison of the GPU detector with the CPU restatement, on the GPU's own
intermediate tensors.  TEST INFRASTRUCTURE ONLY (tests/ and the `parity` block of bench.py's JSON line).

Every stage of the reference test graph (SYM_REL:176-322 + core/tester.py:148-156,244-277) is evaluated by the oracle
on the SAME inputs the GPU stage consumed, so that a rounding difference in one stage cannot flip a discrete decision
(sort order, NMS) of the next and either hide or fake an error:
    proposal   oracle/proposal.py on the GPU's RPN maps      -> number of identical roi rows (bit-exact bar)
    roi_pool   oracle/roi_pooling.py on the GPU's feature map -> mismatching elements (bit-exact bar)
    head       oracle/relation.py (float32) on the GPU's pooled features -> max |cls_prob| error, bbox_pred error
    post       oracle/postprocess.py on the GPU's probabilities / boxes  -> detection-set agreement
"""
import numpy as np

from . import network as ON
from . import proposal as OP
from . import roi_pooling as ORP
from . import relation as OR
from . import postprocess as OPP


def _np(t):
    return t.detach().float().cpu().numpy()


def stagewise(det, data, im_info, params, images=None, relation=True):
    """det: relnet_amd.detector.Detector; data [B,3,H,W] / im_info [B,3] device tensors; params: the name -> tensor dict
    the detector was built from.  Returns a dict of plain numbers (per-image lists + worst cases)."""
    import torch
    c = det.cfg
    with torch.no_grad():
        f = det.backbone.forward(data)
        out = det.forward(data, im_info)
    B = data.shape[0]
    images = list(range(B)) if images is None else list(images)
    pn = {k: (v.numpy() if hasattr(v, 'numpy') else np.asarray(v)) for k, v in params.items()}
    N = out['rois'].shape[1]
    res = dict(images=len(images), rois_per_image=N, proposal_rows_identical=[], roi_pool_mismatches=[],
               cls_prob_max_abs_err=[], bbox_pred_max_rel_err=[], detections_gpu=[], detections_oracle=[],
               detections_matched=[])
    info = _np(im_info)
    for b in images:
        prob = ON.rpn_softmax(_np(f['rpn_cls_score'][b:b + 1]))
        rois_o, _ = OP.proposal(prob, _np(f['rpn_bbox_pred'][b:b + 1]), info[b:b + 1], c.feat_stride, c.anchor_scales,
                                c.anchor_ratios, c.rpn_pre_nms_top_n, c.rpn_post_nms_top_n, c.rpn_nms_thresh, c.rpn_min_size)
        rois = _np(out['rois'][b])
        res['proposal_rows_identical'].append(int((np.abs(rois[:, 1:] - rois_o[:, 1:]).max(axis=1) == 0).sum()))
        r0 = rois.copy(); r0[:, 0] = 0
        feat = _np(f['conv_new_1_relu'][b:b + 1])
        pooled_o = ORP.roi_pooling(feat, r0)
        from relnet_amd import ops
        pooled = _np(ops.roi_pool(f['conv_new_1_relu'], out['rois'][b].contiguous(), channels_last_out=True))
        res['roi_pool_mismatches'].append(int((pooled != pooled_o).sum()))
        if relation:
            r = OR.relation_head(pooled_o, r0, pn, return_intermediates=True)
            cs, bp = r['cls_score'], r['bbox_pred']
        else:
            cs, bp, _ = ON.plain_head(pooled_o, pn)
        res['cls_prob_max_abs_err'].append(float(np.abs(_np(out['cls_prob'][b]) - OPP.softmax_rows(cs)).max()))
        res['bbox_pred_max_rel_err'].append(float(np.abs(_np(out['bbox_pred'][b]) - bp).max() / max(np.abs(bp).max(), 1e-6)))
        full = np.zeros((N, 8)); full[:, 4:8] = out['pred_boxes'][b].cpu().numpy()
        want = OPP.detections(_np(out['cls_prob'][b]), full, c.num_classes, c.score_thresh, c.nms, c.softnms, c.max_per_image)
        flat = np.concatenate([np.hstack((np.full((len(w_), 1), k + 1.0), w_[:, 4:5], w_[:, :4])) for k, w_ in enumerate(want)])
        n = int(out['num_detections'][b])
        got = _np(out['detections'][b, :n])
        m = min(n, len(flat))
        ok = np.isclose(got[:m], flat[:m].astype(np.float32), rtol=1e-5, atol=1e-6).all(axis=1) if m else np.zeros(0, bool)
        res['detections_gpu'].append(n); res['detections_oracle'].append(int(len(flat))); res['detections_matched'].append(int(ok.sum()))
    res['worst'] = dict(proposal_rows_identical=min(res['proposal_rows_identical']), roi_pool_mismatches=max(res['roi_pool_mismatches']),
                        cls_prob_max_abs_err=max(res['cls_prob_max_abs_err']), bbox_pred_max_rel_err=max(res['bbox_pred_max_rel_err']),
                        detections_all_matched=bool(res['detections_matched'] == res['detections_oracle'] == res['detections_gpu']))
    return res


def logit_report(logits, want, gw, wp):
    """How far the attention logits `weighted_aff` (SYM_REL:139) are from the reference values, stated exactly.

    logits / want [N, H, M]; gw = geometry weight relu(E.w_h + b_h) [N, H, M]; wp = pair_pos_fc1 weight [H, 64].
    north_star's bar is |dL| <= 1e-4 in float32.  L contains log(max(G, 1e-6)) of a value G that the graph itself
    computes from sin / cos of arguments scaled by 100 (SYM_REL:36): where G is tiny its float32 rounding noise
    (~1e-7 ||w_h||_1) is amplified by 1 / G, for ANY implementation including MXNet's own kernels.  Reported:
      max_abs_err            over ALL logits
      frac_within_1e-4       share of all logits under the strict bar
      frac_well_conditioned  share with G >= 2e-3 ||w_h||_1; max_abs_err_well_conditioned over those (must be <= 1e-4)
      max_softmax_weighted   max over all logits of softmax(want) * |dL| -- the error as the module output sees it
      max_bound_ratio        max of |dL| / (1e-4 + 8e-7 ||w_h||_1 / max(G, 1e-6)): <= 1 means inside the conditioned bound
    """
    logits, want, gw = np.asarray(logits, np.float64), np.asarray(want, np.float64), np.asarray(gw, np.float64)
    s = np.abs(np.asarray(wp, np.float64)).sum(axis=1)[None, :, None]
    dl = np.abs(logits - want)
    well = gw >= 2e-3 * s
    e = np.exp(want - want.max(axis=2, keepdims=True))
    sm = e / e.sum(axis=2, keepdims=True)
    return dict(max_abs_err=float(dl.max()), frac_within_1e_4=float((dl <= 1e-4).mean()),
                frac_well_conditioned=float(well.mean()),
                max_abs_err_well_conditioned=float(dl[well].max()) if well.any() else 0.0,
                max_softmax_weighted=float((sm * dl).max()),
                max_bound_ratio=float((dl / (1e-4 + 8e-7 * s / np.maximum(gw, 1e-6))).max()))
