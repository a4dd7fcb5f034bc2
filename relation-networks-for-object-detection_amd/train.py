"""End-to-end training step of the relation network (SURVEY.md section 8, rows A10 + A13), one process per GPU.

Graph: the TRAIN branch of relation_rcnn/symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py
:176-322 (reference config experiments/relation_rcnn/cfgs/resnet_v1_101_coco_trainvalminus_rcnn_end2end_relation_
8epoch.yaml): backbone -> RPN losses + proposal -> proposal_target (300 proposals + gt rows, BATCH_ROIS -1) ->
ROIPooling -> fc_new_1 -> relation_1 -> fc_new_2 -> relation_2 (keys = the first 300 rows) -> cls_score / bbox_pred
-> BoxAnnotatorOHEM (128) -> SoftmaxOutput / smooth_l1 losses; then the adjoint of all of it, ONE summed all-reduce
of the trainable gradients (core/module.py + kvstore 'device' in the reference, rescale_grad = 1.0) and
mx.optimizer.SGD (momentum 0.9, wd 5e-4; train_end2end.py:163-168).  Frozen, as cfgs/*.yaml:23-29: conv1, res2 and
every BatchNorm gamma / beta.  The learn-NMS head's training graph is not built yet.

MI355X-first choices:
  * master weights live in ONE flat fp32 buffer already in the kernels' layouts (convs [Cout][R][S][Cin] with the
    frozen BatchNorm scale folded in, FCs [out][in]); momentum, gradients and the bf16 working copy are flat buffers of
    the same shape, so the optimizer is one launch and the all-reduce one collective over 288 GB-class HBM;
  * folding the frozen BN scale s into w' = w s is exact: dL/dw = s dL/dw', so SGD on w' uses the gradient s^2 dL/dw'
    and the same weight decay; `export_params` divides the scale out again;
  * activations are kept in HBM between forward and backward (~0.45 GB / image, bf16); the relation modules are
    recomputed from their inputs instead of storing [16, N, M] maps.
"""
import math

import numpy as np
import torch

from . import ops, losses, train_ops as T
from . import dist as D
from .backbone import unit_names, conv_bn_names, fold_bn, EPS
from .relation import attention_module_backward, _module_forward, pack_pair_pos
from .detector import Config, fc1_channels_last_perm
from .operator_py.proposal import generate_anchors, propose_batch


class TrainConfig(Config):
    rpn_batch_size = 256          # TRAIN.RPN_BATCH_SIZE
    batch_rois_ohem = 128         # TRAIN.BATCH_ROIS_OHEM
    lr = 0.0005
    momentum = 0.9
    wd = 0.0005


class _Flat(object):
    """Named fp32 tensors carved out of one flat buffer (64-element aligned slices)."""

    def __init__(self, named, device):
        self.slices, off = {}, 0
        for n, t in named:
            self.slices[n] = (off, tuple(t.shape))
            off += (t.numel() + 63) // 64 * 64
        self.size = off
        self.master = torch.zeros(off, device=device, dtype=torch.float32)
        for n, t in named:
            self.view(self.master, n).copy_(t)
        self.mom = torch.zeros_like(self.master)
        self.grad = torch.zeros_like(self.master)
        self.work = self.master.to(torch.bfloat16)

    def view(self, buf, name):
        off, shape = self.slices[name]
        n = 1
        for d in shape:
            n *= d
        return buf[off:off + n].view(shape)


class Trainer(object):
    def __init__(self, params, cfg=None, device='cuda', im_hw=(600, 1000)):
        self.cfg = cfg or TrainConfig()
        self.device, self.im_hw = device, im_hw
        c = self.cfg
        dev = device
        f32 = lambda t: torch.as_tensor(t).to(dev, torch.float32).contiguous()
        self.units = unit_names()
        # ---- frozen part (conv1, res2): the inference kernels with folded BN
        self.frozen = {}
        w1, b1 = fold_bn(params['conv1_weight'], params['bn_conv1_gamma'], params['bn_conv1_beta'],
                         params['bn_conv1_moving_mean'], params['bn_conv1_moving_var'])
        self.w_stem, self.b_stem = ops.pack_stem_weight(w1, torch.bfloat16, dev), f32(b1)
        self.zero_bias64 = torch.zeros(64, device=dev, dtype=torch.float32)
        weights, biases = [], []
        self.bn_scale, self.conv_bias, self.ksize = {}, {}, {'rpn_conv_3x3': 3, 'rpn_out': 1, 'conv_new_1': 1}
        for conv, bn, oc, ic, k in conv_bn_names():
            if conv == 'conv1':
                continue
            w, b = fold_bn(params[conv + '_weight'], params[bn + '_gamma'], params[bn + '_beta'],
                           params[bn + '_moving_mean'], params[bn + '_moving_var'])
            self.ksize[conv] = k
            self.conv_bias[conv] = f32(b)                                   # beta - mean * s: frozen
            if conv.startswith('res2'):
                self.frozen[conv] = ops.pack_conv_weight(w, torch.bfloat16, dev)
            else:
                s = (params[bn + '_gamma'].double() / torch.sqrt(params[bn + '_moving_var'].double() + EPS)).float()
                self.bn_scale[conv] = f32(s)
                weights.append((conv, w.permute(0, 2, 3, 1).reshape(oc, -1)))
        # ---- RPN head, conv_new_1, 2FC head, relation modules
        def conv_w(name):
            w = params[name + '_weight']
            return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
        weights.append(('rpn_conv_3x3', conv_w('rpn_conv_3x3'))); biases.append(('rpn_conv_3x3', params['rpn_conv_3x3_bias']))
        self.na2 = params['rpn_cls_score_weight'].shape[0]
        weights.append(('rpn_out', torch.cat([conv_w('rpn_cls_score'), conv_w('rpn_bbox_pred')], 0)))
        biases.append(('rpn_out', torch.cat([params['rpn_cls_score_bias'], params['rpn_bbox_pred_bias']], 0)))
        weights.append(('conv_new_1', conv_w('conv_new_1'))); biases.append(('conv_new_1', params['conv_new_1_bias']))
        self.fc1_perm = fc1_channels_last_perm()
        weights.append(('fc_new_1', params['fc_new_1_weight'][:, self.fc1_perm])); biases.append(('fc_new_1', params['fc_new_1_bias']))
        weights.append(('fc_new_2', params['fc_new_2_weight'])); biases.append(('fc_new_2', params['fc_new_2_bias']))
        self.num_classes = params['cls_score_weight'].shape[0]
        weights.append(('cls_bbox', torch.cat([params['cls_score_weight'], params['bbox_pred_weight']], 0)))
        biases.append(('cls_bbox', torch.cat([params['cls_score_bias'], params['bbox_pred_bias']], 0)))
        for i in (1, 2):
            weights.append(('qk_%d' % i, torch.cat([params['query_%d_weight' % i], params['key_%d_weight' % i]], 0)))
            biases.append(('qk_%d' % i, torch.cat([params['query_%d_bias' % i], params['key_%d_bias' % i]], 0)))
            wo = params['linear_out_%d_weight' % i]
            weights.append(('linear_out_%d' % i, wo.reshape(wo.shape[0], wo.shape[1])))
            biases.append(('linear_out_%d' % i, params['linear_out_%d_bias' % i]))
            weights.append(('pair_pos_fc1_%d' % i, params['pair_pos_fc1_%d_weight' % i]))
            biases.append(('pair_pos_fc1_%d' % i, params['pair_pos_fc1_%d_bias' % i]))
        self.W = _Flat(weights, dev)          # weight decay applies
        self.Bv = _Flat(biases, dev)          # biases: wd_mult 0 (MXNet's rule for names not ending in _weight / _gamma)
        self.anchors = torch.as_tensor(generate_anchors(c.feat_stride, c.anchor_ratios, c.anchor_scales), dtype=torch.float64, device=dev)
        self.step_count = 0

    # ---- accessors ----------------------------------------------------------------------------------------
    def w(self, name):          # bf16 working copy
        return self.W.view(self.W.work, name)

    def b(self, name):          # fp32 bias
        return self.Bv.view(self.Bv.master, name)

    def num_trainable(self):
        return sum(int(np.prod(s)) for _, s in self.W.slices.values()) + sum(int(np.prod(s)) for _, s in self.Bv.slices.values())

    def _add_wgrad(self, name, dw, scale_rows=None):
        g = self.W.view(self.W.grad, name)
        dw = dw.reshape(g.shape)
        if scale_rows is not None:
            dw = dw * (scale_rows * scale_rows).view(-1, 1)
        g.add_(dw)

    def _add_bgrad(self, name, db):
        self.Bv.view(self.Bv.grad, name).add_(db.reshape(-1))

    # ---- forward -------------------------------------------------------------------------------------------
    def _conv(self, x, name, stride=1, pad=0, dil=1, relu=False, resid=None, bias=None, out_dtype=None, w=None):
        w = self.w(name) if w is None else w
        bias = self.conv_bias[name] if bias is None else bias
        return ops.conv2d_nhwc(x, w, bias, ksize=self.ksize.get(name, 1), stride=stride, pad=pad, dil=dil, relu=relu,
                               resid=resid, out_dtype=out_dtype)

    def forward_backward(self, data, im_info, gt_boxes, rpn_label, rpn_bbox_target, rpn_bbox_weight, num_gt=None):
        """data [B,3,H,W] fp32; gt_boxes [B,G,5]; rpn_label [B, A*h*w] ((a,y,x) order), rpn_bbox_target / weight
        [B, 4A, h, w] (lib/rpn/rpn.py:assign_anchor layouts).  Accumulates gradients into the flat buffers and
        returns the loss values (reference metric names)."""
        c = self.cfg
        B = data.shape[0]
        self.W.grad.zero_(); self.Bv.grad.zero_()
        # -- frozen stem + res2
        x = ops.stem_conv7(data, self.w_stem, self.b_stem, relu=True)
        x = ops.stem_bias_relu_pool(x, self.zero_bias64)
        saved = []
        conv4 = None
        for stage, nm, ic, mc, oc, stride, dil, proj in self.units:
            n1, na, nb, nc = 'res%s_branch1' % nm, 'res%s_branch2a' % nm, 'res%s_branch2b' % nm, 'res%s_branch2c' % nm
            if stage == 5 and conv4 is None:
                conv4 = x
            if stage == 2:
                fw = lambda n: self.frozen[n]
                sc = self._conv(x, n1, stride=stride, w=fw(n1)) if proj else x
                y = self._conv(x, na, stride=stride, relu=True, w=fw(na))
                y = self._conv(y, nb, pad=dil, dil=dil, relu=True, w=fw(nb))
                x = self._conv(y, nc, relu=True, resid=sc, w=fw(nc))
                continue
            sc = self._conv(x, n1, stride=stride) if proj else x
            y1 = self._conv(x, na, stride=stride, relu=True)
            y2 = self._conv(y1, nb, pad=dil, dil=dil, relu=True)
            out = self._conv(y2, nc, relu=True, resid=sc)
            saved.append((stage, nm, stride, dil, proj, x, y1, y2, out))
            x = out
        conv5 = x
        feat = self._conv(conv5, 'conv_new_1', relu=True, bias=self.b('conv_new_1'))
        r = self._conv(conv4, 'rpn_conv_3x3', pad=1, relu=True, bias=self.b('rpn_conv_3x3'))
        rpn = self._conv(r, 'rpn_out', bias=self.b('rpn_out'), out_dtype=torch.float32)             # [B,h,w,72]
        h, wd_ = rpn.shape[1], rpn.shape[2]
        na2 = self.na2
        # -- RPN losses (per image, like one image per device in the reference)
        score_nchw = rpn[..., :na2].permute(0, 3, 1, 2).contiguous()                                 # [B,2A,h,w]
        d_score = torch.empty_like(score_nchw)
        out = {}
        for b in range(B):
            _, g = losses.softmax_output(score_nchw[b:b + 1].view(1, 2, -1), rpn_label[b:b + 1], multi_output=True,
                                         use_ignore=True, ignore_label=-1.0)
            d_score[b:b + 1] = g.view(1, na2, h, wd_)
        delta = rpn[..., na2:].contiguous()                                                          # NHWC [B,h,w,4A]
        tgt = rpn_bbox_target.permute(0, 2, 3, 1).contiguous()
        wgt = rpn_bbox_weight.permute(0, 2, 3, 1).contiguous()
        rpn_l1, d_delta = losses.smooth_l1_loss(delta, tgt, wgt, 3.0, 1.0 / c.rpn_batch_size)
        out['rpn_bbox_loss'] = rpn_l1.sum() / B
        d_rpn = torch.cat([d_score.permute(0, 2, 3, 1), d_delta], 3).to(torch.bfloat16).contiguous()
        # -- proposals and their targets (no gradient: proposal.py:170-173, proposal_target.py:95-97)
        nchw = lambda t: t.permute(0, 3, 1, 2)
        rois, _ = propose_batch(nchw(rpn[..., :na2]), nchw(rpn[..., na2:]), im_info, self.anchors, c.feat_stride,
                                c.rpn_pre_nms_top_n, c.rpn_post_nms_top_n, c.rpn_nms_thresh, c.rpn_min_size,
                                im_hw=self.im_hw, softmax_pairs=True)
        N = rois.shape[1]
        rois_t, label, bbox_target, bbox_weight = ops.proposal_target(rois, gt_boxes, num_gt)
        R = rois_t.shape[1]
        pooled, argmax = ops.roi_pool(nchw(feat), rois_t.view(B * R, 5), (7, 7), 1.0 / c.feat_stride,
                                      channels_last_out=True, want_argmax=True)
        pooled2 = pooled.permute(0, 2, 3, 1).reshape(B * R, -1)
        # -- 2FC head + relation modules (keys = the first N rows of each image)
        bt = torch.bfloat16
        mods = [self._rel_params(i) for i in (1, 2)]
        wp_t, bp = pack_pair_pos(mods, self.device)
        bias = ops.geometry_bias(rois_t, wp_t, bp, N, half=True)
        f1 = ops.gemm_nt(pooled2, self.w('fc_new_1'), self.b('fc_new_1')).reshape(B, R, -1)
        _, x1, _ = _module_forward(f1, mods[0], bias[0], N, False, True, False)
        f2 = ops.gemm_nt(x1.reshape(B * R, -1), self.w('fc_new_2'), self.b('fc_new_2')).reshape(B, R, -1)
        _, x2, _ = _module_forward(f2, mods[1], bias[1], N, False, True, False)
        cb = ops.gemm_nt(x2.reshape(B * R, -1), self.w('cls_bbox'), self.b('cls_bbox'), out_dtype=torch.float32).reshape(B, R, -1)
        nc = self.num_classes
        cls_score, bbox_pred = cb[:, :, :nc].contiguous(), cb[:, :, nc:].contiguous()
        labels_ohem, weights_ohem = ops.box_annotator_ohem(cls_score, bbox_pred, label, bbox_target, bbox_weight, c.batch_rois_ohem)
        d_cls = torch.empty_like(cls_score)
        for b in range(B):
            _, g = losses.softmax_output(cls_score[b], labels_ohem[b], use_ignore=True, ignore_label=-1.0)
            d_cls[b] = g
        l1, d_bbox = losses.smooth_l1_loss(bbox_pred, bbox_target, weights_ohem, 1.0, 1.0 / c.batch_rois_ohem)
        out['bbox_loss'] = l1.sum() / B
        out['num_ohem'] = (labels_ohem >= 0).sum()
        # ================= backward =================
        d_cb = torch.cat([d_cls, d_bbox], 2).reshape(B * R, -1).to(bt)
        d_x2, dw, db = T.linear_bwd(x2.reshape(B * R, -1), self.w('cls_bbox'), d_cb)
        self._add_wgrad('cls_bbox', dw); self._add_bgrad('cls_bbox', db)
        d_f2 = self._relation_bwd(2, mods[1], f2, x2, rois_t, d_x2.reshape(B, R, -1), N)
        d_x1, dw, db = T.linear_bwd(x1.reshape(B * R, -1), self.w('fc_new_2'), d_f2.reshape(B * R, -1))
        self._add_wgrad('fc_new_2', dw); self._add_bgrad('fc_new_2', db)
        d_f1 = self._relation_bwd(1, mods[0], f1, x1, rois_t, d_x1.reshape(B, R, -1), N)
        d_pool, dw, db = T.linear_bwd(pooled2, self.w('fc_new_1'), d_f1.reshape(B * R, -1))
        self._add_wgrad('fc_new_1', dw); self._add_bgrad('fc_new_1', db)
        # ROIPooling backward -> gradient of conv_new_1_relu
        d_feat = ops.roi_pool_bwd(d_pool.view(B * R, 7, 7, -1).permute(0, 3, 1, 2), argmax, rois_t.view(B * R, 5),
                                  (B, feat.shape[3], feat.shape[1], feat.shape[2]))
        d_feat = d_feat.permute(0, 2, 3, 1).to(bt).contiguous()
        g = T.relu_bwd(d_feat, feat)
        d_x, dw = T.conv1x1_bwd(conv5, self.w('conv_new_1'), g)
        self._add_wgrad('conv_new_1', dw); self._add_bgrad('conv_new_1', g.float().sum((0, 1, 2)))
        # RPN head backward (joins the trunk at conv4)
        d_r, dw = T.conv1x1_bwd(r, self.w('rpn_out'), d_rpn)
        self._add_wgrad('rpn_out', dw); self._add_bgrad('rpn_out', d_rpn.float().sum((0, 1, 2)))
        g_r = T.relu_bwd(d_r, r)
        d_conv4_rpn, dw = T.conv3x3_bwd(conv4, self._dgrad_w('rpn_conv_3x3', 512), g_r, dil=1)
        self._add_wgrad('rpn_conv_3x3', dw); self._add_bgrad('rpn_conv_3x3', g_r.float().sum((0, 1, 2)))
        # trunk: res5 -> res3
        for stage, nm, stride, dil, proj, x_in, y1, y2, o in reversed(saved):
            n1, na, nb, nc_ = 'res%s_branch1' % nm, 'res%s_branch2a' % nm, 'res%s_branch2b' % nm, 'res%s_branch2c' % nm
            if stage == 4 and d_conv4_rpn is not None and nm == '4b22':
                d_x = d_x + d_conv4_rpn          # conv4 = output of res4b22 feeds both res5 and the RPN head
                d_conv4_rpn = None
            g_out = T.relu_bwd(d_x, o)
            d_y2, dw = T.conv1x1_bwd(y2, self.w(nc_), g_out)
            self._add_wgrad(nc_, dw, self.bn_scale[nc_])
            g_y2 = T.relu_bwd(d_y2, y2)
            d_y1, dw = T.conv3x3_bwd(y1, self._dgrad_w(nb, y2.shape[3]), g_y2, dil=dil)
            self._add_wgrad(nb, dw, self.bn_scale[nb])
            g_y1 = T.relu_bwd(d_y1, y1)
            first = (stage == 3 and proj)         # res3a: its input comes from the frozen res2 -> no data gradient
            if proj:
                d_a, dw = T.conv1x1_bwd(x_in, self.w(na), g_y1, stride=stride, need_dx=not first)
                self._add_wgrad(na, dw, self.bn_scale[na])
                d_s, dw = T.conv1x1_bwd(x_in, self.w(n1), g_out, stride=stride, need_dx=not first,
                                        dx_add=d_a if (stride == 1 and not first) else None)
                self._add_wgrad(n1, dw, self.bn_scale[n1])
                d_x = None if first else (d_s if stride == 1 else d_s + d_a)
            else:
                d_x, dw = T.conv1x1_bwd(x_in, self.w(na), g_y1, dx_add=g_out)       # identity shortcut
                self._add_wgrad(na, dw, self.bn_scale[na])
        out['rois'] = rois_t
        out['label'] = labels_ohem
        out['bbox_target'], out['bbox_weight'] = bbox_target, weights_ohem
        out['bbox_pred'] = bbox_pred
        out['cls_score'] = cls_score
        return out

    def _dgrad_w(self, name, cout):
        """[Cout, 9*Cin] packed forward weights -> [Cin, 9*Cout] tap-flipped data-gradient weights."""
        w = self.w(name)
        cin = w.shape[1] // 9
        return w.view(cout, 3, 3, cin).flip(1, 2).permute(3, 1, 2, 0).reshape(cin, 9 * cout).contiguous()

    def _rel_params(self, i):
        class P(object):
            pass
        m = P()
        m.wqk, m.bqk = self.w('qk_%d' % i), self.b('qk_%d' % i)
        m.wout, m.bout = self.w('linear_out_%d' % i), self.b('linear_out_%d' % i)
        m.wp = self.W.view(self.W.master, 'pair_pos_fc1_%d' % i)
        m.bp = self.b('pair_pos_fc1_%d' % i)
        return m

    def _relation_bwd(self, i, mod, f, x_act, rois, d_x, N):
        """x_act = relu(f + relation_i(f)); returns d f and accumulates the module's parameter gradients."""
        g = T.relu_bwd(d_x.contiguous(), x_act)
        r = attention_module_backward(f, rois, None, g, nongt_dim=N, index=i, dtype=torch.bfloat16, packed=mod)
        d = mod.wqk.shape[0] // 2
        self._add_wgrad('qk_%d' % i, torch.cat([r['query_%d_weight' % i], r['key_%d_weight' % i]], 0))
        self._add_bgrad('qk_%d' % i, torch.cat([r['query_%d_bias' % i], r['key_%d_bias' % i]], 0))
        self._add_wgrad('linear_out_%d' % i, r['linear_out_%d_weight' % i].reshape(d, -1))
        self._add_bgrad('linear_out_%d' % i, r['linear_out_%d_bias' % i])
        self._add_wgrad('pair_pos_fc1_%d' % i, r['pair_pos_fc1_%d_weight' % i])
        self._add_bgrad('pair_pos_fc1_%d' % i, r['pair_pos_fc1_%d_bias' % i])
        return (r['d_roi_feat'] + g.float()).to(torch.bfloat16)          # residual path + module path

    # ---- optimizer ----------------------------------------------------------------------------------------
    def all_reduce(self):
        """ONE summed all-reduce per flat buffer over RCCL (MXNet kvstore 'device' + rescale_grad 1.0 semantics)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.W.grad, op=dist.ReduceOp.SUM)
            dist.all_reduce(self.Bv.grad, op=dist.ReduceOp.SUM)

    def update(self, lr=None):
        c = self.cfg
        lr = c.lr if lr is None else lr
        T.sgd_update(self.W.master, self.W.mom, self.W.grad, lr, c.momentum, c.wd, 1.0, w_bf16=self.W.work)
        T.sgd_update(self.Bv.master, self.Bv.mom, self.Bv.grad, lr, c.momentum, 0.0, 1.0)
        self.step_count += 1

    def step(self, *batch, **kw):
        out = self.forward_backward(*batch, **kw)
        self.all_reduce()
        self.update()
        return out


def assign_anchor(feat_hw, gt_boxes, im_hw, cfg, seed=0, allowed_border=0):
    """Host-side RPN label preparation (the reference does this in its data loader: lib/rpn/rpn.py:80-244).
    gt_boxes [G,5] numpy; returns label [A*h*w] ((a,y,x) order), bbox_target [4A,h,w], bbox_weight [4A,h,w]."""
    rng = np.random.RandomState(seed)
    base = generate_anchors(cfg.feat_stride, cfg.anchor_ratios, cfg.anchor_scales).astype(np.float64)
    A = base.shape[0]
    fh, fw = feat_hw
    sx, sy = np.meshgrid(np.arange(fw) * cfg.feat_stride, np.arange(fh) * cfg.feat_stride)
    shifts = np.stack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()], 1)
    anchors = (base[None] + shifts[:, None]).reshape(-1, 4)                  # (y, x, a) major -> K*A rows
    total = anchors.shape[0]
    inside = np.where((anchors[:, 0] >= -allowed_border) & (anchors[:, 1] >= -allowed_border) &
                      (anchors[:, 2] < im_hw[1] + allowed_border) & (anchors[:, 3] < im_hw[0] + allowed_border))[0]
    a = anchors[inside]
    labels = np.full(len(inside), -1, np.float32)
    targets = np.zeros((len(inside), 4), np.float32)
    if gt_boxes.size > 0:
        g = gt_boxes[:, :4].astype(np.float64)
        iw = np.minimum(a[:, None, 2], g[None, :, 2]) - np.maximum(a[:, None, 0], g[None, :, 0]) + 1
        ih = np.minimum(a[:, None, 3], g[None, :, 3]) - np.maximum(a[:, None, 1], g[None, :, 1]) + 1
        inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
        aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
        ga = (g[:, 2] - g[:, 0] + 1) * (g[:, 3] - g[:, 1] + 1)
        ov = inter / (aa[:, None] + ga[None] - inter)
        amax = ov.argmax(1)
        mx = ov[np.arange(len(a)), amax]
        gt_best = ov.max(0)
        labels[mx < 0.3] = 0                                                  # RPN_NEGATIVE_OVERLAP (no clobber)
        labels[np.where(ov == gt_best[None])[0]] = 1
        labels[mx >= 0.7] = 1                                                 # RPN_POSITIVE_OVERLAP
        # bbox_transform(anchors, matched gt), lib/bbox/bbox_transform.py:74-100
        gm = g[amax]
        ew, eh = a[:, 2] - a[:, 0] + 1, a[:, 3] - a[:, 1] + 1
        ecx, ecy = a[:, 0] + 0.5 * (ew - 1), a[:, 1] + 0.5 * (eh - 1)
        gw_, gh_ = gm[:, 2] - gm[:, 0] + 1, gm[:, 3] - gm[:, 1] + 1
        gcx, gcy = gm[:, 0] + 0.5 * (gw_ - 1), gm[:, 1] + 0.5 * (gh_ - 1)
        targets[:] = np.stack([(gcx - ecx) / (ew + 1e-14), (gcy - ecy) / (eh + 1e-14), np.log(gw_ / ew), np.log(gh_ / eh)], 1)
    else:
        labels[:] = 0
    num_fg = int(0.5 * cfg.rpn_batch_size)
    fg = np.where(labels == 1)[0]
    if len(fg) > num_fg:
        labels[rng.choice(fg, len(fg) - num_fg, replace=False)] = -1
    bg = np.where(labels == 0)[0]
    num_bg = cfg.rpn_batch_size - int((labels == 1).sum())
    if len(bg) > num_bg:
        labels[rng.choice(bg, len(bg) - num_bg, replace=False)] = -1
    weights = np.zeros((len(inside), 4), np.float32)
    weights[labels == 1] = 1.0
    L = np.full(total, -1, np.float32); L[inside] = labels
    Tg = np.zeros((total, 4), np.float32); Tg[inside] = targets
    Wg = np.zeros((total, 4), np.float32); Wg[inside] = weights
    L = L.reshape(fh, fw, A).transpose(2, 0, 1).reshape(-1)
    Tg = Tg.reshape(fh, fw, A * 4).transpose(2, 0, 1)
    Wg = Wg.reshape(fh, fw, A * 4).transpose(2, 0, 1)
    return L, Tg, Wg
