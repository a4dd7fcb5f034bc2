#!/usr/bin/env python
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN PYTHON.

Run in the build container only (needs /root/reference; the GPU box has neither the
reference nor a need for this script -- tests read the committed .npz files):

    python tests/golden/gen_golden.py [--ref /root/reference]

What runs from the reference, unchanged, imported from where it lies:
  lib/rpn/generate_anchor.py            generate_anchors
  lib/bbox/bbox_transform.py            bbox_pred (nonlinear_pred), clip_boxes,
                                        nonlinear_transform, bbox_overlaps_py
  lib/nms/nms.py                        nms, soft_nms
  relation_rcnn/symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py
                                        extract_position_matrix, extract_position_embedding,
                                        attention_module_multi_head
  relation_rcnn/operator_py/learn_nms.py  LearnNmsOperator.forward (+ its nd helpers)
  relation_rcnn/core/rcnn.py              get_rcnn_testbatch (ROIDispatch: FPN level assignment + regrouping)
  lib/rpn/rpn.py                          assign_anchor (print statements rewritten in memory by lib2to3's fix_print)
  relation_rcnn/operator_py/proposal.py   ProposalOperator.forward (same print rewrite; gpu_nms -> the reference's py_nms)
  lib/dataset/pycocotools/cocoeval.py     COCOeval.evaluate / evaluateImg / accumulate / summarize (print + tuple-parameter
                                          rewrite; the C `mask.iou` -> a numpy transcription of maskApi.c:bbIou)

Shims needed because the reference is Python-2 / numpy-1 / MXNet-1.1.0 code (none of
them edits a reference file): `xrange`, `np.float`/`np.int` aliases, `cPickle`, stub
modules for the compiled Cython/CUDA extensions (`bbox`, `cpu_nms`, `gpu_nms`), an in-memory
lib2to3 `fix_print` pass for the files whose only Python-3 problem is the print statement
(`_load_py2`), and the numpy MXNet stand-in in tests/golden/refshim (its docstring says what
that does and does not pin).  No oracle/ code is used to produce any golden value.
"""
import argparse
import builtins
import importlib.util
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # repo root (oracle package)
import cases  # noqa: E402

F32 = np.float32


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_py2(name, path):
    """Import a Python-2 reference file whose only Python-3 problem is the `print` statement: the source is read from
    where it lies and passed through lib2to3's fix_print IN MEMORY (a mechanical statement -> function rewrite)."""
    from lib2to3.refactor import RefactoringTool
    src = open(path).read()
    src3 = str(RefactoringTool(['lib2to3.fixes.fix_print']).refactor_string(src if src.endswith('\n') else src + '\n', path))
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile(src3, path, 'exec'), mod.__dict__)
    return mod


def setup_reference(ref):
    builtins.xrange = range
    if not hasattr(np, 'float'):
        np.float = float
    if not hasattr(np, 'int'):
        np.int = int
    sys.modules['cPickle'] = pickle
    sys.path.insert(0, os.path.join(HERE, 'refshim'))
    import mxnet  # noqa: F401  (the numpy stand-in)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def _no_ext(*a, **k):
        raise RuntimeError("compiled reference extension is not available")

    stub('bbox', bbox_overlaps_cython=_no_ext)
    stub('cpu_nms', cpu_nms=_no_ext)
    stub('gpu_nms', gpu_nms=_no_ext)
    op = stub('operator_py')
    op.__path__ = []
    for sub in ('proposal', 'proposal_target', 'box_annotator_ohem'):
        stub('operator_py.' + sub)
    sys.path.insert(0, os.path.join(ref, 'lib'))            # utils.symbol
    ga = _load('ref_generate_anchor', os.path.join(ref, 'lib/rpn/generate_anchor.py'))
    bt = _load('ref_bbox_transform', os.path.join(ref, 'lib/bbox/bbox_transform.py'))
    nm = _load('ref_nms', os.path.join(ref, 'lib/nms/nms.py'))
    sym_dir = os.path.join(ref, 'relation_rcnn/symbols')
    _load('resnet_v1_101_rcnn_base', os.path.join(sym_dir, 'resnet_v1_101_rcnn_base.py'))
    rel = _load('ref_sym_rel', os.path.join(
        sym_dir, 'resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py'))
    lnms = _load('ref_learn_nms', os.path.join(ref, 'relation_rcnn/operator_py/learn_nms.py'))
    # training-target operators: the compiled `bbox_overlaps_cython` is replaced by the reference's
    # own pure-python twin bbox_overlaps_py (bbox_transform.py:22-42)
    bt.bbox_overlaps = bt.bbox_overlaps_py
    pk = stub('bbox')
    pk.__path__ = []
    pk.bbox_overlaps_cython = _no_ext
    sys.modules['bbox.bbox_transform'] = bt
    # lib/bbox/bbox_regression.py has Python-2 print statements: loaded through the in-memory fix_print rewrite
    sys.modules['bbox_transform'] = bt
    sys.modules['bbox.bbox_regression'] = _load_py2('ref_bbox_regression', os.path.join(ref, 'lib/bbox/bbox_regression.py'))
    stub('utils.image', get_image=_no_ext, tensor_vstack=_no_ext)
    ohem = _load('ref_ohem', os.path.join(ref, 'relation_rcnn/operator_py/box_annotator_ohem.py'))
    nmt = _load('ref_nms_multi_target', os.path.join(ref, 'relation_rcnn/operator_py/nms_multi_target.py'))
    rcnn = _load('ref_rcnn', os.path.join(ref, 'relation_rcnn/core/rcnn.py'))
    setup_reference.extra = (ohem, nmt, rcnn)
    return ga, bt, nm, rel, lnms


def gen_boxes(ga, bt, out):
    d = {}
    d['anchors_default'] = ga.generate_anchors()
    d['anchors_cfg'] = ga.generate_anchors(base_size=16, ratios=np.array([0.5, 1, 2]),
                                           scales=np.array([4, 8, 16, 32]))
    rng = np.random.default_rng(5)
    boxes = cases.random_boxes(200, 6)
    deltas = rng.normal(0, 0.5, (200, 8)).astype(F32)
    pred = bt.nonlinear_pred(boxes, deltas)
    d['pred_boxes_in'] = boxes
    d['pred_deltas_in'] = deltas
    d['pred_out'] = pred
    d['pred_out_f64deltas'] = bt.nonlinear_pred(boxes, deltas.astype(np.float64))
    d['clip_out'] = bt.clip_boxes(pred.copy(), (cases.IM_H, cases.IM_W))
    gt = cases.random_boxes(200, 7)
    d['transform_gt_in'] = gt
    d['transform_out'] = bt.nonlinear_transform(boxes.astype(np.float64), gt.astype(np.float64))
    d['overlaps_out'] = bt.bbox_overlaps_py(boxes[:20].astype(np.float64), gt[:15].astype(np.float64))
    np.savez_compressed(os.path.join(out, 'boxes.npz'), **d)


def gen_nms(nm, out):
    d = {}
    for name, (n, seed) in {'a': (300, 31), 'b': (1000, 32)}.items():
        dets = cases.dets_case(n, seed)
        for t in (0.3, 0.5, 0.7):
            d['nms_%s_%d' % (name, int(t * 10))] = np.asarray(nm.nms(dets.copy(), t), dtype=np.int64)
        d['softnms_%s' % name] = nm.soft_nms(dets.copy(), 0.6, -1)
        d['softnms_%s_max100' % name] = nm.soft_nms(dets.copy(), 0.6, 100)
    np.savez_compressed(os.path.join(out, 'nms.npz'), **d)


def gen_relation(rel, out):
    import mxnet as mx
    cls = rel.resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16
    net = cls()
    d = {}
    for name, (n, m, seed, std) in cases.RELATION_CASES.items():
        boxes, feat, p = cases.relation_case(n, m, seed, std)
        mx.PARAMS.clear(); mx.TRACE.clear()
        mx.PARAMS.update({k: mx.NDArray(v) for k, v in p.items()})
        pm = net.extract_position_matrix(mx.NDArray(boxes), nongt_dim=m)
        pe = net.extract_position_embedding(pm, feat_dim=64)
        y = net.attention_module_multi_head(mx.NDArray(feat), pe, nongt_dim=m, fc_dim=16,
                                            feat_dim=1024, index=1, group=16,
                                            dim=(1024, 1024, 1024))
        logits, soft = mx.TRACE['softmax_1']
        d[name + '/position_matrix'] = pm.asnumpy()
        d[name + '/position_embedding'] = pe.asnumpy()
        d[name + '/logits'] = logits                      # weighted_aff [N, 16, M]
        d[name + '/softmax'] = soft
        d[name + '/output'] = y.asnumpy()
    np.savez_compressed(os.path.join(out, 'relation.npz'), **d)


def gen_relation_large(rel, lnms, out):
    """Full-size relation modules (N = M = 300, N = 333 / M = 300, N = M = 1000) and the 300-roi x 80-class learn-NMS
    operator through the reference's own Python; only a subset of query rows is stored (cases.kept_rows)."""
    import mxnet as mx
    cls = rel.resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16
    net = cls()
    d = {}
    for name, (n, m, seed, std, k) in cases.RELATION_LARGE_CASES.items():
        boxes, feat, p = cases.relation_case(n, m, seed, std)
        mx.PARAMS.clear(); mx.TRACE.clear()
        mx.PARAMS.update({kk: mx.NDArray(v) for kk, v in p.items()})
        pm = net.extract_position_matrix(mx.NDArray(boxes), nongt_dim=m)
        pe = net.extract_position_embedding(pm, feat_dim=64)
        y = net.attention_module_multi_head(mx.NDArray(feat), pe, nongt_dim=m, fc_dim=16, feat_dim=1024, index=1, group=16,
                                            dim=(1024, 1024, 1024))
        logits, soft = mx.TRACE['softmax_1']
        rows = cases.kept_rows(n, k, seed)
        d[name + '/rows'] = rows
        d[name + '/logits'] = logits[rows]                # weighted_aff [rows, 16, M]
        d[name + '/output'] = y.asnumpy()[rows]
        d[name + '/position_matrix'] = pm.asnumpy()[rows]
        print('   ', name, 'done', flush=True)
        del pm, pe, y, logits, soft
    for name, (n, c, first_n, seed) in cases.LEARN_NMS_LARGE_CASES.items():
        cls_score, bbox_pred, rois, im_info, feat, p = cases.learn_nms_case(n, c, seed)
        op = lnms.LearnNmsOperator(num_fg_classes=c, bbox_means=None, bbox_stds=None, first_n=first_n, class_agnostic=True,
                                   num_thresh=5, class_thresh=0.01, nongt_dim=n, has_non_gt_index=False)
        in_data = [mx.NDArray(x) for x in (cls_score, bbox_pred, rois, im_info, feat)]
        in_data += [mx.NDArray(p[kk]) for kk in cases.LEARN_NMS_ARG_ORDER]
        outs = [mx.nd.zeros((first_n, c, 5)), mx.nd.zeros((first_n, c, 4)), mx.nd.zeros((first_n, c))]
        mx.TRACE.clear()
        op.forward(False, ['write'] * 3, in_data, outs, [])
        d[name + '/nms_multi_score'] = outs[0].asnumpy()
        d[name + '/sorted_bbox'] = outs[1].asnumpy()
        d[name + '/sorted_score'] = outs[2].asnumpy()
        print('   ', name, 'done', flush=True)
    np.savez_compressed(os.path.join(out, 'relation_large.npz'), **d)


def gen_learn_nms_fpn(lnms, out):
    """LearnNmsOperator.forward at the FPN experiment's values (1000 rois, 80 classes, first_n 150, class_thresh 0.05;
    symbols/..._fpn_..._learn_nms.py:1328-1362 passes nongt_dim=None and, at test time, no non_gt_index)."""
    import mxnet as mx
    d = {}
    for name, (n, c, first_n, seed, th) in cases.LEARN_NMS_FPN_CASES.items():
        cls_score, bbox_pred, rois, im_info, feat, p = cases.learn_nms_fpn_case(n, c, seed)
        op = lnms.LearnNmsOperator(num_fg_classes=c, bbox_means=None, bbox_stds=None, first_n=first_n, class_agnostic=True,
                                   num_thresh=5, class_thresh=th, nongt_dim=None, has_non_gt_index=False)
        in_data = [mx.NDArray(x) for x in (cls_score, bbox_pred, rois, im_info, feat)]
        in_data += [mx.NDArray(p[kk]) for kk in cases.LEARN_NMS_ARG_ORDER]
        outs = [mx.nd.zeros((first_n, c, 5)), mx.nd.zeros((first_n, c, 4)), mx.nd.zeros((first_n, c))]
        mx.TRACE.clear()
        op.forward(False, ['write'] * 3, in_data, outs, [])
        d[name + '/nms_multi_score'] = outs[0].asnumpy()
        d[name + '/sorted_bbox'] = outs[1].asnumpy()
        d[name + '/sorted_score'] = outs[2].asnumpy()
        print('   ', name, 'done: classes with a non-zero score', int((outs[0].asnumpy().max(axis=(0, 2)) > 0).sum()), flush=True)
    np.savez_compressed(os.path.join(out, 'learn_nms_fpn.npz'), **d)


def gen_learn_nms(lnms, out):
    import mxnet as mx
    d = {}
    for name, (n, c, first_n, seed) in cases.LEARN_NMS_CASES.items():
        cls_score, bbox_pred, rois, im_info, feat, p = cases.learn_nms_case(n, c, seed)
        op = lnms.LearnNmsOperator(num_fg_classes=c, bbox_means=None, bbox_stds=None,
                                   first_n=first_n, class_agnostic=True, num_thresh=5,
                                   class_thresh=0.01, nongt_dim=n, has_non_gt_index=False)
        in_data = [mx.NDArray(x) for x in (cls_score, bbox_pred, rois, im_info, feat)]
        in_data += [mx.NDArray(p[k]) for k in cases.LEARN_NMS_ARG_ORDER]
        outs = [mx.nd.zeros((first_n, c, 5)), mx.nd.zeros((first_n, c, 4)), mx.nd.zeros((first_n, c))]
        mx.TRACE.clear()
        op.forward(False, ['write'] * 3, in_data, outs, [])
        d[name + '/nms_multi_score'] = outs[0].asnumpy()
        d[name + '/sorted_bbox'] = outs[1].asnumpy()
        d[name + '/sorted_score'] = outs[2].asnumpy()
        d[name + '/logits'] = mx.TRACE['nms_softmax_1'][0]
    np.savez_compressed(os.path.join(out, 'learn_nms.npz'), **d)


def gen_targets(out):
    import mxnet as mx
    ohem, nmt, rcnn = setup_reference.extra

    class _NS(object):
        pass
    cfg = _NS(); cfg.TRAIN = _NS()
    cfg.CLASS_AGNOSTIC = True
    cfg.TRAIN.BG_THRESH_HI = 0.5
    cfg.TRAIN.BBOX_NORMALIZATION_PRECOMPUTED = True
    cfg.TRAIN.BBOX_MEANS = (0.0, 0.0, 0.0, 0.0)
    cfg.TRAIN.BBOX_STDS = (0.1, 0.1, 0.2, 0.2)
    cfg.TRAIN.BBOX_WEIGHTS = np.array([1.0, 1.0, 1.0, 1.0])
    d = {}
    rois, gt_boxes, cls_score, bbox_pred = cases.targets_case(90, 7, 61)
    # proposal_target.py:64-67 (restated glue) + the reference's sample_rois_v2
    all_rois = np.vstack((rois, np.hstack((np.zeros((gt_boxes.shape[0], 1), F32), gt_boxes[:, :-1]))))
    r, lab, bt_, bw = rcnn.sample_rois_v2(all_rois, 81, cfg, gt_boxes=gt_boxes)
    d['pt/rois'] = r; d['pt/label'] = lab; d['pt/bbox_target'] = bt_; d['pt/bbox_weight'] = bw
    op = ohem.BoxAnnotatorOHEMOperator(81, 2, 32)
    ins = [mx.NDArray(x) for x in (cls_score, bbox_pred, lab, bt_, bw)]
    outs = [mx.nd.zeros(lab.shape), mx.nd.zeros(bw.shape)]
    op.forward(True, ['write'] * 2, ins, outs, [])
    d['ohem/labels'] = outs[0].asnumpy(); d['ohem/bbox_weights'] = outs[1].asnumpy()
    bbox, gt_box, score = cases.nms_target_case(40, 6, 9, 62)
    op = nmt.NmsMultiTargetOp(np.array([0.5, 0.6, 0.7, 0.8, 0.9]))
    outs = [mx.nd.zeros((40, 6, 5))]
    op.forward(True, ['write'], [mx.NDArray(bbox), mx.NDArray(gt_box), mx.NDArray(score)], outs, [])
    d['nmt/target'] = outs[0].asnumpy()
    np.savez_compressed(os.path.join(out, 'targets.npz'), **d)


def gen_rpn_targets(ref, out):
    """RPN anchor labels / regression targets by the reference's own loader code (lib/rpn/rpn.py:assign_anchor),
    with numpy's global generator seeded (the function subsamples fg / bg with numpy.random.choice)."""
    sys.modules['generate_anchor'] = sys.modules['ref_generate_anchor']
    rpn = _load_py2('ref_rpn', os.path.join(ref, 'lib/rpn/rpn.py'))

    class NS(object):
        pass
    cfg = NS(); cfg.TRAIN = NS()
    cfg.TRAIN.RPN_CLOBBER_POSITIVES = False
    cfg.TRAIN.RPN_NEGATIVE_OVERLAP, cfg.TRAIN.RPN_POSITIVE_OVERLAP = 0.3, 0.7
    cfg.TRAIN.RPN_FG_FRACTION, cfg.TRAIN.RPN_BATCH_SIZE = 0.5, 256
    cfg.TRAIN.RPN_BBOX_WEIGHTS = (1.0, 1.0, 1.0, 1.0)
    d = {}
    for name, (seed, G) in {'six_gt': (3, 6), 'crowded': (4, 40)}.items():
        gt = cases.rpn_gt_boxes(G, seed)
        np.random.seed(seed)
        lab = rpn.assign_anchor((1, 48, 38, 63), gt, np.array([[600, 1000, 1.0]], F32), cfg, feat_stride=16,
                                scales=(4, 8, 16, 32), ratios=(0.5, 1, 2), allowed_border=0)
        d[name + '/gt'] = gt
        d[name + '/label'] = lab['label']
        d[name + '/bbox_target'] = lab['bbox_target']
        d[name + '/bbox_weight'] = lab['bbox_weight']
    np.savez_compressed(os.path.join(out, 'rpn_targets.npz'), **d)


def gen_proposal(ref, out):
    """The whole proposal operator by the reference's own operator_py/proposal.py:ProposalOperator.forward under the numpy
    MXNet stand-in.  Its hard-coded gpu_nms_wrapper (a compiled CUDA extension) is replaced by the reference's own numpy
    NMS, lib/nms/nms.py:py_nms_wrapper -- identical to nms_kernel.cu except at IoU == thresh exactly (`<=` vs `>`)."""
    import mxnet as mx
    ga, nm = sys.modules['ref_generate_anchor'], sys.modules['ref_nms']
    pk = types.ModuleType('rpn'); pk.__path__ = []; sys.modules['rpn'] = pk
    sys.modules['rpn.generate_anchor'] = ga
    pk2 = types.ModuleType('nms'); pk2.__path__ = []; sys.modules['nms'] = pk2
    sys.modules['nms.nms'] = nm
    prop = _load_py2('ref_proposal', os.path.join(ref, 'relation_rcnn/operator_py/proposal.py'))
    prop.gpu_nms_wrapper = lambda thresh, device_id: nm.py_nms_wrapper(thresh)
    d = {}
    for name, (seed, pre, post) in {'full': (61, 6000, 300), 'small': (62, 600, 50)}.items():
        cls_prob, deltas, im_info = cases.rpn_case(seed)
        op = prop.ProposalOperator(16, '(4, 8, 16, 32)', '(0.5, 1, 2)', True, pre, post, 0.7, 0)
        outs = [mx.nd.zeros((post, 5)), mx.nd.zeros((post, 1))]
        np.random.seed(seed)
        op.forward(False, ['write', 'write'], [mx.NDArray(cls_prob), mx.NDArray(deltas), mx.NDArray(im_info)], outs, [])
        d[name + '/rois'] = outs[0].asnumpy()
        d[name + '/score'] = outs[1].asnumpy()
    np.savez_compressed(os.path.join(out, 'proposal.npz'), **d)


def gen_fpn(out):
    """ROI -> pyramid-level dispatch of the FPN graphs, by running the reference's own loader code
    (relation_rcnn/core/rcnn.py:get_rcnn_testbatch, cfg.network.ROIDispatch) on float32 proposals."""
    rcnn = setup_reference.extra[2]
    rcnn.get_image = lambda roidb, cfg: ([np.zeros((1, 3, 8, 8), F32) for _ in roidb], roidb)

    class NS(object):
        pass
    cfg = NS(); cfg.network = NS(); cfg.TEST = NS()
    cfg.network.ROIDispatch = True
    cfg.TEST.LEARN_NMS = False
    d = {}
    cases_ = {'all_levels': cases.fpn_proposals(400, 51), 'empty_level0': None}
    b = cases.fpn_proposals(300, 52)
    small = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1) < 112.0 ** 2
    b[small] = [100, 100, 400, 400]
    cases_['empty_level0'] = b
    for name, boxes in cases_.items():
        data, _, _ = rcnn.get_rcnn_testbatch([{'boxes': boxes, 'im_info': [800, 1024, 1.0]}], cfg)
        for l in range(4):
            d['%s/rois_%d' % (name, l)] = np.asarray(data[0]['rois_%d' % l])
        d['%s/boxes' % name] = boxes
    np.savez_compressed(os.path.join(out, 'fpn.npz'), **d)


def gen_cocoeval(ref, out):
    """The 12 COCO bbox statistics + the precision / recall arrays by the reference's own evaluation code
    (lib/dataset/pycocotools/cocoeval.py: evaluate, evaluateImg, accumulate, summarize), read from where it lies and made
    importable in memory by lib2to3's mechanical Python-2 fixers (print, tuple_params: `lambda (ind, g): ...`, filter / map / zip /
    dict / xrange: list-returning builtins).  Its two dependencies are replaced:
    `mask.iou` (the C extension, maskApi.c:98-109 bbIou) by a numpy transcription of those twelve lines, and the COCO index
    object by a 30-line stand-in serving getImgIds / getCatIds / getAnnIds / loadAnns from the same annotation lists."""
    from lib2to3.refactor import RefactoringTool
    if not hasattr(np, 'float'):
        np.float = float                              # numpy-1 alias the file uses (as in setup_reference)
    path = os.path.join(ref, 'lib/dataset/pycocotools/cocoeval.py')
    src = open(path).read()
    src3 = str(RefactoringTool(['lib2to3.fixes.fix_' + f for f in ('print', 'tuple_params', 'filter', 'map', 'zip', 'dict', 'xrange')]).refactor_string(src + '\n', path))

    def bb_iou(dt, gt, iscrowd):                      # maskApi.c:98-109 behind _mask.pyx:iou (:205-233): [len(dt), len(gt)],
        if len(dt) == 0 or len(gt) == 0:              # and an empty LIST when either side is empty (:213-214)
            return []
        dt, gt = np.asarray(dt, np.float64).reshape(-1, 4), np.asarray(gt, np.float64).reshape(-1, 4)
        o = np.zeros((len(dt), len(gt)))
        for g in range(len(gt)):
            ga = gt[g, 2] * gt[g, 3]
            for d in range(len(dt)):
                da = dt[d, 2] * dt[d, 3]
                w = min(dt[d, 2] + dt[d, 0], gt[g, 2] + gt[g, 0]) - max(dt[d, 0], gt[g, 0])
                if w <= 0:
                    continue
                h = min(dt[d, 3] + dt[d, 1], gt[g, 3] + gt[g, 1]) - max(dt[d, 1], gt[g, 1])
                if h <= 0:
                    continue
                i = w * h
                o[d, g] = i / (da if iscrowd[g] else da + ga - i)
        return o

    mask_stub = types.ModuleType('mask'); mask_stub.iou = bb_iou
    sys.modules['mask'] = mask_stub
    mod = types.ModuleType('ref_cocoeval'); mod.__file__ = path
    exec(compile(src3, path, 'exec'), mod.__dict__)

    class Index(object):                              # the slice of pycocotools.coco.COCO that cocoeval.py calls
        def __init__(self, anns):
            self.anns = {a['id']: a for a in anns}
        def getImgIds(self): return sorted({a['image_id'] for a in self.anns.values()})
        def getCatIds(self): return sorted({a['category_id'] for a in self.anns.values()})
        def getAnnIds(self, imgIds=(), catIds=()):
            imgIds, catIds = set(imgIds), set(catIds)
            return [i for i, a in self.anns.items() if (not imgIds or a['image_id'] in imgIds) and (not catIds or a['category_id'] in catIds)]
        def loadAnns(self, ids): return [self.anns[i] for i in ids]

    import copy
    gts, dts = cases.cocoeval_case()
    gt_index = Index(copy.deepcopy(gts))
    lin = np.linspace                                 # numpy-1 era call: np.linspace(.5, .95, np.round(...) + 1) with a FLOAT count
    np.linspace = lambda a, b, num=50, **kw: lin(a, b, int(num), **kw)
    try:
        ev = mod.COCOeval(gt_index, Index(copy.deepcopy(dts)))
    finally:
        np.linspace = lin
    ev.params.useSegm = 0                             # bbox evaluation (lib/dataset/coco.py:_do_python_eval)
    ev.params.imgIds = sorted({a['image_id'] for a in gts + dts})
    ev.params.catIds = sorted({a['category_id'] for a in gts + dts})
    ev.evaluate(); ev.accumulate(); ev.summarize()
    np.savez_compressed(os.path.join(out, 'cocoeval.npz'), stats=np.asarray(ev.stats, np.float64),
                        precision=ev.eval['precision'], recall=ev.eval['recall'])
    print('cocoeval stats', np.round(ev.stats, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=HERE)
    ap.add_argument('--only-large', action='store_true', help='regenerate relation_large.npz only (minutes of numpy at N = 1000)')
    ap.add_argument('--skip-large', action='store_true')
    ap.add_argument('--only-lnms-fpn', action='store_true', help='regenerate learn_nms_fpn.npz only')
    a = ap.parse_args()
    ga, bt, nm, rel, lnms = setup_reference(a.ref)
    if a.only_lnms_fpn:
        return gen_learn_nms_fpn(lnms, a.out)
    if not a.skip_large:
        gen_relation_large(rel, lnms, a.out)
    if a.only_large:
        return
    gen_boxes(ga, bt, a.out)
    gen_nms(nm, a.out)
    gen_relation(rel, a.out)
    gen_learn_nms(lnms, a.out)
    gen_learn_nms_fpn(lnms, a.out)
    gen_targets(a.out)
    gen_fpn(a.out)
    gen_rpn_targets(a.ref, a.out)
    gen_proposal(a.ref, a.out)
    gen_cocoeval(a.ref, a.out)
    for f in sorted(os.listdir(a.out)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(a.out, f)), 'bytes')


if __name__ == '__main__':
    main()
