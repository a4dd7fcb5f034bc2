"""The reference's OWN graphs (tests/golden/symbols/*.json = relation_rcnn/symbols/*.py run unchanged on the `mx` facade,
tests/golden/gen_symbol_json.py) executed on the GPU by mx/executor.py, against the hand-wired Detector and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import network as ON
from oracle import proposal as OP
from oracle import roi_pooling as ORP
from oracle import relation as OR

pytestmark = pytest.mark.gpu
SYM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'symbols')


@pytest.fixture(scope='module')
def rn():
    import relnet_amd  # noqa: F401
    from relnet_amd import mx, backbone, detector, lib
    lib.load()
    return mx, backbone, detector


def _params(backbone, seed, spread=True):
    p = backbone.init_params(seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    if spread:
        for k in ('cls_score_weight', 'bbox_pred_weight'):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.05
        p['conv_new_1_bias'] = torch.rand(256, generator=g) * 0.1 + 0.05
    return p, g


def _bind(mx, name, p, dtype):
    sym = mx.sym.load(os.path.join(SYM_DIR, name + '.json'))
    args = {k: p[k] for k in sym.list_arguments() if k in p}
    aux = {k: p[k] for k in sym.list_auxiliary_states()}
    return sym, sym.bind(mx.gpu(0), args=args, aux_states=aux, dtype=dtype)


def _np(t):
    return t.detach().float().cpu().numpy()


def test_relation_graph_bf16_matches_detector(rn):
    """Reference relation test graph (SYM_REL:176-322) at the benchmark's size, bf16, vs Detector on the same weights."""
    mx, backbone, detector = rn
    H, W = 600, 1000
    p, g = _params(backbone, 1)
    data = torch.randn(1, 3, H, W, generator=g)
    im_info = torch.tensor([[H, W, 1.0]])
    sym, exe = _bind(mx, 'rcnn_end2end_relation_8epoch_test', p, torch.bfloat16)
    rep = exe.fused_report
    assert rep['attention_modules'] == 2 and rep['conv_chains'] >= 104, rep
    # res2a|b, res2b|c, res3a|b1, res3b1|b2, res3b2|b3 + the last units res2c, res3b3, and (r04) res4's 22 boundaries + its last unit
    assert rep.get('block_boundaries') == 30, rep
    assert all(err < 3e-2 for _, err in rep['probe']), rep['probe']
    outs = exe.forward(is_train=False, data=data, im_info=im_info)
    names = sym.list_outputs()
    assert names[:3] == ['rois_output', 'cls_prob_reshape_output', 'bbox_pred_reshape_output']
    rois, cls_prob, bbox_pred, att1, att2 = [o.asnumpy() for o in outs]
    assert rois.shape == (300, 5) and cls_prob.shape == (1, 300, 81) and bbox_pred.shape == (1, 300, 8) and att1.shape == (300, 1024)
    det = detector.Detector(p, dtype=torch.bfloat16, im_hw=(H, W))
    # the graph executor runs res2a's projection shortcut as its own convolution (bf16 map); the Detector's fused form keeps it
    # in fp32 inside the expand kernel -- one rounding fewer, so a row-exact comparison needs the same form on both sides
    det.backbone.chain_proj = {}
    ref = det.forward(data.cuda(), im_info.cuda())
    r_ref = _np(ref['rois'][0])
    same = (np.abs(rois - r_ref).max(axis=1) == 0)
    assert same.sum() >= 285, same.sum()                     # RPN softmax: torch expression vs in-kernel -> ulp-level score ties
    cp_ref = _np(ref['cls_prob'][0])
    if same.all():
        assert np.abs(cls_prob[0] - cp_ref).max() < 2e-2
        assert np.abs(bbox_pred[0] - _np(ref['bbox_pred'][0])).max() < 2e-2 * max(1.0, np.abs(bbox_pred).max())
    np.testing.assert_allclose(cls_prob[0].sum(1), 1.0, atol=1e-3)


def test_relation_graph_fp32_matches_oracle(rn):
    """float32 executor on a small image: each stage of the reference graph vs the oracle on the same inputs."""
    mx, backbone, detector = rn
    H, W = 192, 256
    p, g = _params(backbone, 3)
    data = torch.randn(1, 3, H, W, generator=g)
    im_info = torch.tensor([[H, W, 1.0]])
    sym = mx.sym.load(os.path.join(SYM_DIR, 'rcnn_end2end_relation_8epoch_test.json'))
    internals = sym.get_internals()
    want = ['rois_output', 'conv_new_1_relu_output', 'roi_pool_output', 'fc_new_1_output', 'cls_score_output', 'bbox_pred_output',
            'rpn_cls_prob_reshape_output', 'rpn_bbox_pred_output']
    group = mx.sym.Group([internals[n] for n in want] + [sym[3], sym[4]])
    exe = group.bind(mx.gpu(0), args={k: p[k] for k in group.list_arguments() if k in p},
                     aux_states={k: p[k] for k in group.list_auxiliary_states()}, dtype=torch.float32)
    assert exe.fused_report['attention_modules'] == 2 and all(e < 2e-4 for _, e in exe.fused_report['probe']), exe.fused_report
    outs = [o.asnumpy() for o in exe.forward(is_train=False, data=data, im_info=im_info)]
    rois, feat, pooled, fc1, cls_score, bbox_pred, rpn_prob, rpn_box, att1, att2 = outs
    # the graph's nongt_dim is the cfg's 300: with fewer surviving boxes the reference pads by re-sampling kept ones
    with torch.no_grad():
        c4, c5 = ON.backbone(data, p)
        cls, box, feat_o = ON.rpn_and_feat(c4, c5, p)
    assert np.abs(feat - feat_o.numpy()).max() < 2e-4 * feat_o.abs().max().item()
    assert np.abs(rpn_prob - ON.rpn_softmax(cls.numpy())).max() < 1e-5
    rois_o, _ = OP.proposal(rpn_prob, rpn_box, im_info.numpy(), 16, (4, 8, 16, 32), (0.5, 1, 2), 6000, 300, 0.7, 0)
    pad_free = min(300, int(len(np.unique(rois_o, axis=0))))
    assert np.array_equal(rois[:pad_free], rois_o[:pad_free]) or np.abs(rois[:pad_free] - rois_o[:pad_free]).max() < 1e-3
    assert np.array_equal(pooled, ORP.roi_pooling(feat, rois))
    pn = {k: v.numpy() for k, v in p.items()}
    r = OR.relation_head(pooled, rois, pn, return_intermediates=True)
    for got, key in ((fc1, 'fc_new_1'), (att1, 'attention_1'), (att2, 'attention_2'), (cls_score, 'cls_score'), (bbox_pred, 'bbox_pred')):
        w = r[key]
        assert np.abs(got - w).max() <= 2e-4 * max(np.abs(w).max(), 1e-3), key


def test_plain_and_learn_nms_graphs_run(rn):
    mx, backbone, detector = rn
    H, W = 192, 256
    p, g = _params(backbone, 5)
    data = torch.randn(1, 3, H, W, generator=g)
    im_info = torch.tensor([[H, W, 1.0]])
    sym, exe = _bind(mx, 'rcnn_end2end_8epoch_test', p, torch.bfloat16)
    assert exe.fused_report['attention_modules'] == 0
    rois, cls_prob, bbox_pred = [o.asnumpy() for o in exe.forward(is_train=False, data=data, im_info=im_info)]
    det = detector.Detector(p, dtype=torch.bfloat16, relation=False, im_hw=(H, W))
    ref = det.forward(data.cuda(), im_info.cuda())
    same = (np.abs(rois - _np(ref['rois'][0])).max(axis=1) == 0)
    assert same.sum() >= 0.9 * len(same)
    if same.all():
        assert np.abs(cls_prob[0] - _np(ref['cls_prob'][0])).max() < 2e-2
    # relation + learn-NMS test graph (SYM_RELNMS:240-569, `learn_nms` CustomOp)
    sym, exe = _bind(mx, 'rcnn_end2end_relation_learn_nms_8epoch_test', p, torch.bfloat16)
    outs = exe.forward(is_train=False, data=data, im_info=im_info)
    od = dict(zip(sym.list_outputs(), outs))
    assert len(outs) == 6 and outs[0].shape == (300, 5)
    shapes = sorted(o.shape for o in outs[3:])
    assert shapes == sorted([(100, 80, 4), (100, 80), (100, 80)]), {k: v.shape for k, v in od.items()}
    assert all(np.isfinite(o.asnumpy()).all() for o in outs)
