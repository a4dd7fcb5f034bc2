"""Oracle: stage-by-stage ("teacher forced") comparison of the GPU detector with the CPU restatement, on the GPU's own
intermediate tensors.  TEST INFRASTRUCTURE ONLY (tests/ and the `parity` block of bench.py's JSON line).

Every stage of the reference test graph (SYM_REL:176-322 + core/tester.py:148-156,244-277) is evaluated by the oracle
on the SAME inputs the GPU stage consumed, so that a rounding difference in one stage cannot flip a discrete decision
(sort order, NMS) of the next and either hide or fake an error:
    backbone   oracle/network.py (float32 torch-CPU) on the raw images -> relative error of conv4 / conv5 / conv_new_1 / RPN maps
               (not teacher forced: ~100 bf16 layers against float32, bar = the bf16 tolerance of tests/test_gpu_pipeline.py)
    proposal   oracle/proposal.py on the GPU's RPN maps      -> number of identical roi rows (bit-exact bar)
    roi_pool   oracle/roi_pooling.py on the GPU's feature map -> mismatching elements (bit-exact bar)
    head       oracle/relation.py (float32) on the GPU's pooled features -> cls_score / bbox_pred / attention_1 / attention_2 errors
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


def _rel(got, want):
    """(relative L2 error, max |error| / max |want|) of a GPU tensor against the oracle's."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    d = got - want
    return (float(np.sqrt((d * d).sum()) / max(np.sqrt((want * want).sum()), 1e-30)),
            float(np.abs(d).max() / max(np.abs(want).max(), 1e-30)))


def stagewise(det, data, im_info, params, images=None, relation=True, backbone=True):
    """det: relnet_amd.detector.Detector; data [B,3,H,W] / im_info [B,3] device tensors; params: the name -> tensor dict
    the detector was built from.  Returns a dict of plain numbers (per-image lists + worst cases).

    EVERY compared tensor comes from ONE `det.forward(data, im_info, keep_features=True)` call over the WHOLE batch -- the
    configuration that is timed (at 54 images: the res3-res5 chain kernels, the RPN head on its side stream, the batched
    proposal / pooling / attention launches) -- and the oracle is evaluated on the checked images only:
      backbone  (A1/A2) oracle/network.py float32 torch-CPU ResNet-101 + RPN head + conv_new_1 on the raw image
                -> relative L2 / max error of conv4, conv5, conv_new_1_relu, rpn_cls_score, rpn_bbox_pred
      proposal / roi_pool / head / post: teacher forced on the GPU's own tensors, as described in the module docstring;
                the head is compared on the raw `cls_score` / `bbox_pred` logits and on the two relation-module outputs
                (`attention_1`, `attention_2`), relative to their own scale -- a softmax over near-uniform scores would hide a
                broken relation module, the module outputs themselves cannot."""
    import torch
    c = det.cfg
    with torch.no_grad():
        out = det.forward(data, im_info, keep_features=True)
    f, hd = out['features'], out['head']
    B = data.shape[0]
    images = list(range(B)) if images is None else list(images)
    pn = {k: (v.numpy() if hasattr(v, 'numpy') else np.asarray(v)) for k, v in params.items()}
    N = out['rois'].shape[1]
    res = dict(images=len(images), batch=B, rois_per_image=N, proposal_rows_identical=[], proposals_kept_before_padding=[],
               roi_pool_mismatches=[], cls_score_max_rel_err=[], cls_prob_max_abs_err=[], bbox_pred_max_rel_err=[],
               detections_gpu=[], detections_oracle=[], detections_matched=[],
               same_forward_call=True, chain_kernel_units=getattr(det.backbone, 'last_chain_units', None),
               stage_sub_batches=getattr(det.backbone, 'last_stage_split', None),
               rpn_side_stream=bool(getattr(det.backbone, 'last_side_stream', False)))
    if relation:
        res.update(attention_1_max_rel_err=[], attention_2_max_rel_err=[])
    if backbone:
        res['backbone'] = {k: dict(rel_l2=[], rel_max=[]) for k in ('conv4', 'conv5', 'conv_new_1_relu', 'rpn_cls_score', 'rpn_bbox_pred')}
    info = _np(im_info)
    num_kept = out['num_kept'].cpu().numpy()
    for b in images:
        if backbone:
            with torch.no_grad():
                img = data[b:b + 1].detach().float().cpu()
                c4, c5 = ON.backbone(img, params)
                cls_o, box_o, feat_o = ON.rpn_and_feat(c4, c5, params)
            for key, want in (('conv4', c4), ('conv5', c5), ('conv_new_1_relu', feat_o), ('rpn_cls_score', cls_o), ('rpn_bbox_pred', box_o)):
                l2, mx = _rel(_np(f[key][b:b + 1]), want.numpy())
                res['backbone'][key]['rel_l2'].append(l2); res['backbone'][key]['rel_max'].append(mx)
        prob = ON.rpn_softmax(_np(f['rpn_cls_score'][b:b + 1]))
        rois_o, _ = OP.proposal(prob, _np(f['rpn_bbox_pred'][b:b + 1]), info[b:b + 1], c.feat_stride, c.anchor_scales,
                                c.anchor_ratios, c.rpn_pre_nms_top_n, c.rpn_post_nms_top_n, c.rpn_nms_thresh, c.rpn_min_size)
        rois = _np(out['rois'][b])
        res['proposal_rows_identical'].append(int((np.abs(rois[:, 1:] - rois_o[:, 1:]).max(axis=1) == 0).sum()))
        res['proposals_kept_before_padding'].append(int(num_kept[b]))
        r0 = rois.copy(); r0[:, 0] = 0
        feat = _np(f['conv_new_1_relu'][b:b + 1])
        pooled_o = ORP.roi_pooling(feat, r0)
        # the pooled features the head of THIS call consumed: memory order (ph, pw, c) -> the oracle's (c, ph, pw)
        pooled = _np(out['pooled'][b]).reshape(N, pooled_o.shape[2], pooled_o.shape[3], pooled_o.shape[1]).transpose(0, 3, 1, 2)
        res['roi_pool_mismatches'].append(int((pooled != pooled_o).sum()))
        if relation:
            r = OR.relation_head(pooled_o, r0, pn, return_intermediates=True)
            cs, bp = r['cls_score'], r['bbox_pred']
            for i in (1, 2):
                res['attention_%d_max_rel_err' % i].append(_rel(_np(hd['attention_%d' % i][b]), r['attention_%d' % i])[1])
        else:
            cs, bp, _ = ON.plain_head(pooled_o, pn)
        res['cls_score_max_rel_err'].append(_rel(_np(out['cls_score'][b]), cs)[1])
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
    short = [k for k in res['proposals_kept_before_padding'] if k < N]
    res['proposal_padding'] = ('none: every checked image kept >= %d boxes after NMS' % N) if not short else \
        ('%d checked image(s) kept fewer than %d boxes (min %d): rows past the kept count are PADDING -- keep[i mod n_kept] here, '
         'npr.choice (random) in the reference (proposal.py:154-156); only the first n_kept rows are comparable to MXNet'
         % (len(short), N, min(short)))
    res['worst'] = dict(proposal_rows_identical=min(res['proposal_rows_identical']), roi_pool_mismatches=max(res['roi_pool_mismatches']),
                        cls_score_max_rel_err=max(res['cls_score_max_rel_err']),
                        cls_prob_max_abs_err=max(res['cls_prob_max_abs_err']), bbox_pred_max_rel_err=max(res['bbox_pred_max_rel_err']),
                        detections_all_matched=bool(res['detections_matched'] == res['detections_oracle'] == res['detections_gpu']))
    if relation:
        res['worst'].update(attention_1_max_rel_err=max(res['attention_1_max_rel_err']), attention_2_max_rel_err=max(res['attention_2_max_rel_err']))
    if backbone:
        res['worst'].update({'backbone_%s_rel_l2' % k: max(v['rel_l2']) for k, v in res['backbone'].items()})
        res['worst'].update({'backbone_%s_rel_max' % k: max(v['rel_max']) for k, v in res['backbone'].items()})
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
