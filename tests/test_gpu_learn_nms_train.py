"""The learn-NMS head's TRAIN branch (symbols/..._learn_nms.py:424-551; relnet_amd.train.Trainer._lnms_forward_backward) at the values
the FPN experiment words -- FIRST_N 150 (..._rcnn_fpn_relation_learn_nms_8epoch.yaml:141), 80 classes -- in isolation against float64
autograd of oracle/train_graph.py:learn_nms_loss.  At first_n = 150 the class-batched relation module has Mpad = 160: past the
N, Mpad <= 128 gate of relation_attention_bwd_small_kernel, i.e. on the two-kernel backward (first_n = 100 runs the small kernel)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases  # noqa: E402
from oracle import train_graph as OT  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('first_n', [150, 100])
def test_learn_nms_head_gradients_at_the_fpn_yaml_first_n(first_n):
    import relnet_amd  # noqa: F401
    from relnet_amd import backbone, train, ops
    B, N, C, G = 2, 300, 80, 6
    p = backbone.init_params(seed=3)
    g_ = torch.Generator().manual_seed(78)
    p['nms_logit_bias'] = torch.zeros(5)       # un-saturate the duplicate classifier (init bias -3) so that its gradients are not tiny
    for k in ('nms_logit_weight', 'nms_rank_weight', 'roi_feat_embedding_weight', 'nms_query_1_weight', 'nms_key_1_weight',
              'nms_linear_out_1_weight', 'nms_pair_pos_fc1_1_weight'):
        p[k] = torch.randn(p[k].shape, generator=g_) * 0.05
    cfg = train.TrainConfig()
    cfg.learn_nms, cfg.first_n = True, first_n
    tr = train.Trainer(p, cfg, im_hw=(600, 1000))
    assert ops.relation_bwd_small_ok(torch.bfloat16, first_n, ops.pad32(first_n)) == (first_n <= 128)
    ins = [cases.learn_nms_case(N, C, 90 + b) for b in range(B)]
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    cls_score = d(np.stack([i[0] for i in ins]))
    bbox_pred = d(np.stack([i[1] for i in ins]))
    rois = d(np.stack([i[2] for i in ins]))
    im_info = d(np.concatenate([i[3] for i in ins]))
    feat = d(np.stack([i[4] for i in ins])).to(torch.bfloat16)
    # gt boxes ON the top-scoring proposals of a few classes so that positive NMS targets exist
    gt = np.zeros((B, G, 5), np.float32)
    for b in range(B):
        cs = ins[b][0]
        for j in range(G):
            c = 1 + 7 * j
            gt[b, j, :4] = ins[b][2][np.argmax(cs[:, c]), 1:]
            gt[b, j, 4] = c
    num_gt = torch.full((B,), G, dtype=torch.int32, device='cuda')
    tr.W.grad.zero_(); tr.Bv.grad.zero_()
    tr._relayout.run()
    d_cls, d_feat, lo = tr._lnms_forward_backward(cls_score, bbox_pred, rois, im_info, feat, d(gt), num_gt)
    tr._flush_wgrads()
    torch.cuda.synchronize()
    assert lo['nms_multi_target'].sum() > 0
    names = ['nms_rank', 'roi_feat_embedding', 'nms_pair_pos_fc1_1', 'nms_logit', 'nms_query_1', 'nms_key_1', 'nms_linear_out_1']
    want_w = {n: 0 for n in names}
    want_b = {n: 0 for n in names}
    for b in range(B):
        pd = {k + s: p[k + s].double().clone().requires_grad_(True) for k in names for s in ('_weight', '_bias')}
        cs64 = torch.as_tensor(ins[b][0]).double().requires_grad_(True)
        ft64 = feat[b].cpu().double().requires_grad_(True)
        loss, multi = OT.learn_nms_loss(cs64, ft64, pd, lo['nms_rank_idx'][b].cpu().numpy(), lo['nms_class_boxes'][b].cpu().numpy(),
                                        lo['nms_multi_target'][b].cpu().numpy(), first_n)
        loss.backward()
        got_m = lo['nms_multi_score'][b].cpu().double()
        assert (got_m - multi.detach()).abs().max() <= 0.03 * multi.abs().max()
        for what, got, want in (('d_cls_score', d_cls[b], cs64.grad), ('d_fc_all_2_relu', d_feat[b], ft64.grad)):
            got = got.cpu().double()
            cos = float((got * want).sum() / (got.norm() * want.norm()))
            assert cos >= 0.995 and abs(float(got.norm() / want.norm()) - 1) <= 0.03, (what, b, cos, float(got.norm() / want.norm()))
        for n in names:
            want_w[n] = want_w[n] + pd[n + '_weight'].grad
            want_b[n] = want_b[n] + pd[n + '_bias'].grad
    gw = lambda n: tr.W.view(tr.W.grad, n).cpu().double()
    gb = lambda n: tr.Bv.view(tr.Bv.grad, n).cpu().double()
    got_w = {n: gw(n) for n in ('nms_rank', 'roi_feat_embedding', 'nms_pair_pos_fc1_1', 'nms_logit')}
    got_b = {n: gb(n) for n in ('nms_rank', 'roi_feat_embedding', 'nms_pair_pos_fc1_1', 'nms_logit', 'nms_linear_out_1')}
    got_w['nms_query_1'], got_w['nms_key_1'] = gw('nms_qk_1')[:1024], gw('nms_qk_1')[1024:]
    got_b['nms_query_1'], got_b['nms_key_1'] = gb('nms_qk_1')[:1024], gb('nms_qk_1')[1024:]
    got_w['nms_linear_out_1'] = gw('nms_linear_out_1')
    bad, report = [], []
    for kind, got, want in (('weight', got_w, want_w), ('bias', got_b, want_b)):
        for n in names:
            w_, g = want[n].reshape(-1), got[n].reshape(-1)
            cos = float((w_ * g).sum() / max(float(w_.norm() * g.norm()), 1e-300))
            report.append('%-22s %-6s |want| %.3e |got| %.3e cos %.5f' % (n, kind, float(w_.norm()), float(g.norm()), cos))
            # (the key bias shifts every logit of a query row equally: its true gradient is ~0 up to rounding -- norm only)
            if n == 'nms_key_1' and kind == 'bias':
                continue
            lim = 0.98 if 'pair_pos' in n else 0.995
            if float(w_.norm()) > 1e-9 and (cos < lim or abs(float(g.norm() / w_.norm()) - 1) > 0.04):
                bad.append(report[-1])
    assert not bad, '\n'.join(bad) + '\n--- all ---\n' + '\n'.join(report)


def test_fused_glue_kernels_equal_the_tensor_operator_chains():
    """csrc/lnms_train.hip (round 6: pad_params, residual_relu, cond_multi, cond_bwd, take_bwd, softmax_bwd -- one kernel per
    element-wise chain of the branch) against the torch-operator form it replaces (cfg.lnms_fused_glue = False), same inputs, same
    trainer: losses, multi scores, both returned gradients (the class-score gradient ACCUMULATED into a non-zero [B, R, 81] buffer
    with R > N) and every parameter gradient of the head."""
    import relnet_amd  # noqa: F401
    from relnet_amd import backbone, train
    B, N, C, G, R = 2, 300, 80, 6, 308
    p = backbone.init_params(seed=3)
    g_ = torch.Generator().manual_seed(79)
    p['nms_logit_bias'] = torch.zeros(5)
    for k in ('nms_logit_weight', 'nms_rank_weight', 'roi_feat_embedding_weight', 'nms_query_1_weight', 'nms_key_1_weight',
              'nms_linear_out_1_weight', 'nms_pair_pos_fc1_1_weight'):
        p[k] = torch.randn(p[k].shape, generator=g_) * 0.05
    cfg = train.TrainConfig()
    cfg.learn_nms, cfg.first_n = True, 100
    tr = train.Trainer(p, cfg, im_hw=(600, 1000))
    ins = [cases.learn_nms_case(N, C, 190 + b) for b in range(B)]
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    cls_score, bbox_pred, rois = (d(np.stack([i[k] for i in ins])) for k in (0, 1, 2))
    im_info = d(np.concatenate([i[3] for i in ins]))
    feat = d(np.stack([i[4] for i in ins])).to(torch.bfloat16)
    gt = np.zeros((B, G, 5), np.float32)
    for b in range(B):
        for j in range(G):
            c = 1 + 7 * j
            gt[b, j, :4] = ins[b][2][np.argmax(ins[b][0][:, c]), 1:]
            gt[b, j, 4] = c
    num_gt = torch.full((B,), G, dtype=torch.int32, device='cuda')
    base = torch.randn(B, R, C + 1, generator=g_).cuda() * 1e-3
    res = {}
    for fused in (False, True):
        cfg.lnms_fused_glue = fused
        tr.W.grad.zero_(); tr.Bv.grad.zero_()
        tr._relayout.run()
        acc = base.clone()
        d_cls, d_feat, lo = tr._lnms_forward_backward(cls_score, bbox_pred, rois, im_info, feat, d(gt), num_gt, d_cls_out=acc)
        tr._flush_wgrads()
        if d_cls is not None:
            assert not fused
            acc[:, :N] += d_cls
        else:
            assert fused
        torch.cuda.synchronize()
        res[fused] = dict(d_cls=acc.clone(), d_feat=d_feat.clone(), multi=lo['nms_multi_score'].clone(), pos=lo['nms_pos_loss'].clone(),
                          neg=lo['nms_neg_loss'].clone(), wg=tr.W.grad.clone(), bg=tr.Bv.grad.clone())
    # the per-image geometry table gathered by the class ranks (relnet_lnms_gather_bias) IS the direct per-class evaluation, bit for bit
    from relnet_amd import ops, lib as _lib
    from relnet_amd.relation import pack_pair_pos
    F_ = cfg.first_n
    class M_(object):
        pass
    m_ = M_(); m_.wp = tr.W.view(tr.W.master, 'nms_pair_pos_fc1_1'); m_.bp = tr.b('nms_pair_pos_fc1_1')
    wp_t, bp = pack_pair_pos([m_], 'cuda')
    direct = ops.geometry_bias(lo['nms_class_boxes'].view(B * C, F_, 4), wp_t, bp, F_, fast32=True)[0]
    boxes = torch.empty((B, N, 4), device='cuda')
    for b in range(B):          # class_boxes[b, c, f] = boxes[b, rank_idx[b, c, f]]: recover the image's box list from any class that ranks the roi
        ri = lo['nms_rank_idx'][b].long()
        boxes[b].index_copy_(0, ri.reshape(-1), lo['nms_class_boxes'][b].reshape(-1, 4))
    img = ops.geometry_bias(boxes, wp_t, bp, N, fast32=True)[0]
    gathered = torch.empty_like(direct)
    _lib.call('relnet_lnms_gather_bias', img.data_ptr(), lo['nms_rank_idx'].data_ptr(), gathered.data_ptr(), B, C, N, img.shape[-1], F_, gathered.shape[-1],
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(gathered[..., :F_], direct[..., :F_])
    a, b_ = res[False], res[True]
    assert torch.equal(a['d_cls'][:, N:], base[:, N:]) and torch.equal(b_['d_cls'][:, N:], base[:, N:])       # rows past N untouched
    for k in ('multi', 'pos', 'neg', 'd_cls', 'd_feat', 'wg', 'bg'):
        x, y = a[k].double().reshape(-1), b_[k].double().reshape(-1)
        cos = float((x * y).sum() / (x.norm() * y.norm()))
        assert cos > 0.99999 and abs(float(y.norm() / x.norm()) - 1) < 1e-3, (k, cos, float(y.norm() / x.norm()))
    for n in ('nms_rank', 'roi_feat_embedding', 'nms_pair_pos_fc1_1', 'nms_logit', 'nms_qk_1', 'nms_linear_out_1'):
        x, y = tr.W.view(a['wg'], n).double().reshape(-1), tr.W.view(b_['wg'], n).double().reshape(-1)
        cos = float((x * y).sum() / (x.norm() * y.norm()))
        assert cos > 0.9999 and abs(float(y.norm() / x.norm()) - 1) < 2e-3, (n, cos)
