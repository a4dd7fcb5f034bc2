"""End-to-end training step of the relation network (SURVEY.md section 8, rows A10 + A13), one process per GPU.

Graph: the TRAIN branch of relation_rcnn/symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py
:176-322 (reference config experiments/relation_rcnn/cfgs/resnet_v1_101_coco_trainvalminus_rcnn_end2end_relation_
8epoch.yaml): backbone -> RPN losses + proposal -> proposal_target (300 proposals + gt rows, BATCH_ROIS -1) ->
ROIPooling -> fc_new_1 -> relation_1 -> fc_new_2 -> relation_2 (keys = the first 300 rows) -> cls_score / bbox_pred
-> BoxAnnotatorOHEM (128) -> SoftmaxOutput / smooth_l1 losses; then the adjoint of all of it, ONE summed all-reduce
of the trainable gradients (core/module.py + kvstore 'device' in the reference, rescale_grad = 1.0) and
mx.optimizer.SGD (momentum 0.9, wd 5e-4; train_end2end.py:163-168).  Frozen, as cfgs/*.yaml:23-29: conv1, res2 and
every BatchNorm gamma / beta.  With cfg.learn_nms the learn-NMS head's train branch (symbols/..._learn_nms.py:424-551:
per-class sort, rank + appearance embedding, class-batched relation module, sigmoid x score, pos / neg log losses
against nms_multi_target) is trained jointly = BASELINE configs[2].

MI355X-first choices:
  * master weights live in ONE flat fp32 buffer already in the kernels' layouts (convs [Cout][R][S][Cin] with the
    frozen BatchNorm scale folded in, FCs [out][in]); momentum, gradients and the bf16 working copy are flat buffers of
    the same shape, so the optimizer is one launch and the all-reduce one collective over 288 GB-class HBM;
  * folding the frozen BN scale s into w' = w s is exact: dL/dw = s dL/dw', so SGD on w' uses the gradient s^2 dL/dw'
    and the same weight decay; `export_params` divides the scale out again;
  * activations are kept in HBM between forward and backward (~0.45 GB / image, bf16); the relation modules are
    recomputed from their inputs instead of storing [16, N, M] maps.
"""

import os

import numpy as np
import torch

from . import ops, losses, train_ops as T
from . import dist as D
from .backbone import unit_names, conv_bn_names, fold_bn, EPS
from .relation import attention_module_backward, _module_forward, pack_pair_pos, GradSink
from .detector import Config, fc1_channels_last_perm
from .operator_py.proposal import generate_anchors, propose_batch


class TrainConfig(Config):
    rpn_batch_size = 256          # TRAIN.RPN_BATCH_SIZE
    batch_rois_ohem = 128         # TRAIN.BATCH_ROIS_OHEM
    lr = 0.0005
    momentum = 0.9
    wd = 0.0005
    nms_loss_scale = 1.0          # TRAIN.nms_loss_scale
    nms_pos_scale = 4.0           # TRAIN.nms_pos_scale
    nms_eps = 1e-8
    bbox_means = (0.0, 0.0, 0.0, 0.0)
    bbox_stds = (0.1, 0.1, 0.2, 0.2)
    relation = True               # two relation modules in the 2FC head (False: the plain head of resnet_v1_101_rcnn_learn_nms_1024_...)
    enable_ohem = True            # TRAIN.ENABLE_OHEM (False: SoftmaxOutput over all rois, bbox loss scaled by 1 / 300)
    fixed_params = None           # network.FIXED_PARAMS of the experiment file (None: dist.FIXED_PARAMS, the end2end yamls' list)


def all_reduce_sum(*buffers):
    """SUM each flat gradient buffer over the ranks (no-op in a single process): the whole data-parallel exchange of a
    training step is these two collectives (weights 271 MB, biases 0.1 MB)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for b in buffers:
            dist.all_reduce(b, op=dist.ReduceOp.SUM)
        if buffers and buffers[0].is_cuda and dist.get_backend() != 'nccl':
            torch.cuda.synchronize()            # gloo on device tensors (one-GPU test path), see dist.BucketedAllReduce.finish


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
        self.fpn = bool(getattr(c, 'fpn', False))
        self.relation = bool(getattr(c, 'relation', True))
        # cfg.trunk_fp32 (parity tests only): conv1 .. res5 forward AND backward in float32 on the exact-fp32 MFMA kernels
        # (relnet_conv2d_nhwc_f32 / relnet_gemm_nt f32, master weights read directly, no chain kernels, no fused ReLU-mask epilogue) with
        # the SAME wiring code (_trunk_forward / _trunk_backward / train_ops) -- removes the bf16 noise of ~100 layers from the
        # end-to-end gradient comparison, so that the trunk's wiring is pinned to float64 autograd at 1e-4 instead of 2e-2
        self.trunk_fp32 = bool(getattr(c, 'trunk_fp32', False))
        assert not (self.trunk_fp32 and getattr(c, 'dcn', False)), "trunk_fp32 covers the plain and FPN trunks"
        if torch.device(dev).type == 'cuda':
            ops.asm_selfcheck()
        self.units = unit_names(self.fpn)
        # the tensors this step never changes (frozen by name, cfgs/*.yaml:23-29, and the BatchNorm running statistics):
        # kept on the host under the reference's names so that a checkpoint holds the complete arg / aux dictionaries
        self._fixed_src = {k: torch.as_tensor(v).detach().to('cpu', torch.float32).clone() for k, v in params.items()
                           if ('moving_' in k) or not D.is_trainable(k)}
        # ---- frozen part (conv1, res2): the inference kernels with folded BN
        self.frozen = {}
        w1, b1 = fold_bn(params['conv1_weight'], params['bn_conv1_gamma'], params['bn_conv1_beta'],
                         params['bn_conv1_moving_mean'], params['bn_conv1_moving_var'])
        self.w_stem, self.b_stem = ops.pack_stem_weight(w1, torch.bfloat16, dev), f32(b1)
        # conv1 + res2 never change: they run on the inference kernels (fused stem, halo 3x3, chain kernels) of a Backbone built from the
        # same parameters (cfg.frozen_on_inference_kernels = False keeps the per-layer convolution launches)
        self._frozen_backbone = None
        if getattr(c, 'frozen_on_inference_kernels', True) and torch.device(dev).type == 'cuda':
            from .backbone import Backbone
            self._frozen_backbone = Backbone(params, dtype=torch.float32 if self.trunk_fp32 else torch.bfloat16, device=dev, fpn=self.fpn, frozen_only=True)
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
        if self.fpn:        # FPN neck (symbols/..._fpn_...:804-840); no RPN in these graphs (HAS_RPN: false)
            for lvl in (32, 16, 8, 4):
                for k_, ks in (('1x1', 1), ('3x3', 3)):
                    n = 'fpn_ft%d_%s' % (lvl, k_)
                    weights.append((n, conv_w(n))); biases.append((n, params[n + '_bias'])); self.ksize[n] = ks
            fc1n, fc2n = 'roi_pool_fc1', 'roi_pool_fc2'
        else:
            weights.append(('rpn_conv_3x3', conv_w('rpn_conv_3x3'))); biases.append(('rpn_conv_3x3', params['rpn_conv_3x3_bias']))
            self.na2 = params['rpn_cls_score_weight'].shape[0]
            weights.append(('rpn_out', torch.cat([conv_w('rpn_cls_score'), conv_w('rpn_bbox_pred')], 0)))
            biases.append(('rpn_out', torch.cat([params['rpn_cls_score_bias'], params['rpn_bbox_pred_bias']], 0)))
            weights.append(('conv_new_1', conv_w('conv_new_1'))); biases.append(('conv_new_1', params['conv_new_1_bias']))
            fc1n, fc2n = 'fc_new_1', 'fc_new_2'
        self.fc_names = (fc1n, fc2n)
        self.fc1_perm = fc1_channels_last_perm()
        weights.append(('fc_new_1', params[fc1n + '_weight'][:, self.fc1_perm])); biases.append(('fc_new_1', params[fc1n + '_bias']))
        weights.append(('fc_new_2', params[fc2n + '_weight'])); biases.append(('fc_new_2', params[fc2n + '_bias']))
        self.num_classes = params['cls_score_weight'].shape[0]
        weights.append(('cls_bbox', torch.cat([params['cls_score_weight'], params['bbox_pred_weight']], 0)))
        biases.append(('cls_bbox', torch.cat([params['cls_score_bias'], params['bbox_pred_bias']], 0)))
        for i in ((1, 2) if self.relation else ()):
            weights.append(('qk_%d' % i, torch.cat([params['query_%d_weight' % i], params['key_%d_weight' % i]], 0)))
            biases.append(('qk_%d' % i, torch.cat([params['query_%d_bias' % i], params['key_%d_bias' % i]], 0)))
            wo = params['linear_out_%d_weight' % i]
            weights.append(('linear_out_%d' % i, wo.reshape(wo.shape[0], wo.shape[1])))
            biases.append(('linear_out_%d' % i, params['linear_out_%d_bias' % i]))
            weights.append(('pair_pos_fc1_%d' % i, params['pair_pos_fc1_%d_weight' % i]))
            biases.append(('pair_pos_fc1_%d' % i, params['pair_pos_fc1_%d_bias' % i]))
        if c.dcn:           # deformable res5 (offset convs) + deformable PSROI pooling's offset FC (SYM_DCN_RELNMS:700-746,1073-1080)
            for u in 'abc':
                n = 'res5%s_branch2b_offset' % u
                weights.append((n, conv_w(n))); biases.append((n, params[n + '_bias']))
                self.ksize[n] = 3
        if c.learn_nms:     # learn-NMS head (symbols/..._learn_nms.py:424-551), trained end to end with the detector
            for n in ('nms_rank', 'roi_feat_embedding', 'nms_pair_pos_fc1_1', 'nms_logit'):
                weights.append((n, params[n + '_weight'])); biases.append((n, params[n + '_bias']))
            weights.append(('nms_qk_1', torch.cat([params['nms_query_1_weight'], params['nms_key_1_weight']], 0)))
            biases.append(('nms_qk_1', torch.cat([params['nms_query_1_bias'], params['nms_key_1_bias']], 0)))
            weights.append(('nms_linear_out_1', params['nms_linear_out_1_weight'].reshape(128, 128)))
            biases.append(('nms_linear_out_1', params['nms_linear_out_1_bias']))
            from .learn_nms import rank_embedding
            self.rank_emb = rank_embedding(c.first_n, 1024).to(dev, torch.bfloat16).contiguous()
        self.lr_mult_tail = None
        if c.dcn:           # `offset` FC has lr_mult = 0.01 (SYM_DCN_RELNMS:1075): keep it LAST so it is one tail range
            weights.append(('offset', params['offset_weight'][:, self.fc1_perm])); biases.append(('offset', params['offset_bias']))
        self.W = _Flat(weights, dev)          # weight decay applies
        self.Bv = _Flat(biases, dev)          # biases: wd_mult 0 (MXNet's rule for names not ending in _weight / _gamma)
        if c.dcn:
            self.lr_mult_tail = (self.W.slices['offset'][0], self.Bv.slices['offset'][0], 0.01)
        # network.FIXED_PARAMS (core/module.py:753-764: a parameter is fixed when ANY pattern is a substring of its name -- 'cls_score' also fixes
        # rpn_cls_score).  The end2end experiments fix conv1 / res2 / the BatchNorm affine terms, which this class never registers as trainable; the
        # learn-NMS-only experiment (..._rcnn_end2end_learn_nms_3epoch.yaml:23-40, JOINT_TRAINING false) fixes the whole detector: what is left are
        # the learn-NMS head's parameters.  A fused slice (rpn_out = rpn_cls_score | rpn_bbox_pred, ...) is frozen when all its members are.
        fixed = tuple(getattr(c, 'fixed_params', None) or D.FIXED_PARAMS)
        members = {'rpn_out': ('rpn_cls_score', 'rpn_bbox_pred'), 'cls_bbox': ('cls_score', 'bbox_pred'), 'fc_new_1': (fc1n,), 'fc_new_2': (fc2n,),
                   'qk_1': ('query_1', 'key_1'), 'qk_2': ('query_2', 'key_2'), 'nms_qk_1': ('nms_query_1', 'nms_key_1')}
        def _frozen(flat_name):
            fr = [not D.is_trainable(m + '_weight', fixed) for m in members.get(flat_name, (flat_name,))]
            assert all(fr) or not any(fr), "FIXED_PARAMS splits the fused slice %s" % flat_name
            return all(fr)
        self.frozen_names = {n for n in self.W.slices if _frozen(n)}
        lnms_names = {'nms_rank', 'roi_feat_embedding', 'nms_pair_pos_fc1_1', 'nms_logit', 'nms_qk_1', 'nms_linear_out_1'}
        # the pruned step: every parameter outside the learn-NMS head is fixed -> no gradient leaves that head, nothing behind it is differentiated
        self.lnms_only = bool(c.learn_nms) and bool(self.frozen_names) and all((n in self.frozen_names) != (n in lnms_names) for n in self.W.slices)
        assert not self.frozen_names or self.lnms_only, "fixed parameter sets other than the end2end yamls' / the learn-NMS-only yaml's are not built: %s" % sorted(self.frozen_names)
        self.anchors_host = generate_anchors(c.feat_stride, c.anchor_ratios, c.anchor_scales)      # float64 [A,4], host (kernel argument)
        self.anchors = torch.as_tensor(self.anchors_host, dtype=torch.float64, device=dev)
        self.step_count = 0
        # state that used to be created on first use -- inside a hipGraph capture that turned into a captured fill / mid-capture
        # stream and attribute calls: the device step counter of the anchor sampler, the RPN side stream, the queue of
        # weight-gradient products
        self._anchor_step = torch.zeros(1, device=dev, dtype=torch.int64)
        self._side = torch.cuda.Stream(device=dev) if torch.cuda.is_available() and torch.device(dev).type == 'cuda' else None
        self._wq = ops.WgradQueue()
        # bias-gradient column sums of a gradient bucket in one grouped launch (train_ops.ColsumQueue; cfg.colsum_grouped / RELNET_COLSUM_GROUP=0: one launch each)
        self._cq = T.ColsumQueue()
        # parameter-only gradient work (the relation modules' geometry backward) on a side stream beside the data-gradient chain, joined before the
        # bucket is announced: OPT-IN (cfg.aux_stream / RELNET_AUX_STREAM=1).  Measured, same box, r06: 6.95 -> 7.16-7.67 ms at one image, 17.93 -> 18.02-18.12 ms
        # at 8 -- like the weight-gradient side stream, every fork / join inside the captured step costs more than the overlap returns
        self._aux = torch.cuda.Stream(device=dev) if (self._side is not None and getattr(c, 'aux_stream', os.environ.get('RELNET_AUX_STREAM', '0') != '0')) else None
        self._aux_keep, self._aux_pending = [], False
        self._colsum_grouped = bool(getattr(c, 'colsum_grouped', os.environ.get('RELNET_COLSUM_GROUP', '1') != '0'))
        # every data-parallel rank samples its own fg / bg anchor subsets (cfg.rank_in_anchor_seed = False: the same subsets on every
        # rank, what the two-rank gradient-sum check needs)
        self._rank, self._world = (0, 1) if not getattr(c, 'rank_in_anchor_seed', True) else ((torch.distributed.get_rank(), torch.distributed.get_world_size())
                                   if torch.distributed.is_available() and torch.distributed.is_initialized() else (0, 1))
        # data-gradient copies of the weights (W^T, tap-flipped 3x3 filters): ONE grouped launch per step instead of a
        # transpose / flip / copy per layer inside the backward pass
        self._relayout = ops.WeightRelayout(dev)
        for name in self.W.slices:
            taps = 9 if self.ksize.get(name, 1) == 3 else 1
            group = None
            if name == 'nms_qk_1':                           # learn-NMS module: [Wq; Wk]^T | padded Wout^T (8 real of 64 columns per head)
                self._relayout.add(name, self.w(name), taps=1, pad_co=64, group=('rel_cat_nms', 0, 3 * 1024))
                wo = self.w('nms_linear_out_1')
                for h in range(16):
                    self._relayout.add('nms_lo_h%d' % h, wo[8 * h:8 * h + 8], taps=1, pad_co=8, group=('rel_cat_nms', 2048 + 64 * h, 3 * 1024))
                continue
            if name == 'nms_logit':                          # [T,128] -> [128, 64] (T real columns): data-gradient operand of the learn-NMS logit layer
                self._relayout.add(name, self.w(name), taps=1, pad_co=64)
                continue
            if name.startswith(('pair_pos_fc1', 'nms_')) or name.endswith('_offset') or self._trunk32(name):
                continue                                    # consumed in other layouts (relation_bwd kernels, padded DCN offset convs; float32 trunk)
            if name.startswith(('qk_', 'linear_out_')):     # [Wq; Wk]^T | Wout^T side by side: the [1024, 3072] operand of the ONE
                i = name.rsplit('_', 1)[1]                  # projection-backward GEMM of relation module i (relation.GradSink)
                group = ('rel_cat_' + i, 0 if name.startswith('qk_') else self.W.slices['qk_' + i][1][0])
            self._relayout.add(name, self.w(name), taps=taps, pad_co=1 if taps == 9 else 64, group=group)
        self._relayout.build()
        # ---- res3 .. res5 block boundaries on the inference chain kernels (csrc/bottleneck.hip): expand + shortcut + ReLU of unit u
        # and reduce + ReLU of unit u + 1 in one launch.  They read the weights in MFMA-fragment order; the trained weights change every
        # step, so ONE grouped launch per step (ops.FragRepack) rewrites those copies next to the data-gradient layouts above.
        # cfg.train_chain = False keeps the per-layer convolution launches (the round-4 form).
        self._fragpack = ops.FragRepack(dev)
        self.chain_units = {}                               # unit -> (has_next_reduce, next unit)
        if getattr(c, 'train_chain', os.environ.get('RELNET_TRAIN_CHAIN', '1') != '0') and torch.device(dev).type == 'cuda' and not self.trunk_fp32:
            trainable = [u for u in self.units if u[0] >= 3]
            for (st, nm, ic, mc, oc, stride, dil, proj), nxt in zip(trainable, trainable[1:] + [None]):
                if c.dcn and st == 5:
                    continue
                with_reduce = nxt is not None and nxt[0] == st and not nxt[7] and mc in ops.CHAIN_MIDS
                if not with_reduce and mc not in ops.CHAIN_EXPAND_MIDS:
                    continue
                self._fragpack.add('w3:' + nm, self.w('res%s_branch2c' % nm), 0)
                if with_reduce:
                    self._fragpack.add('w1:' + nxt[1], self.w('res%s_branch2a' % nxt[1]), 1)
                self.chain_units[nm] = (with_reduce, nxt[1] if with_reduce else None)
            self._fragpack.build()
        # data gradient through `relu(expand + shortcut)` with the ReLU mask in the GEMM epilogue (relnet_gemm_nt_mask) instead of a
        # separate relnet_relu_bwd pass over three [pixels, 4 mid] maps per unit
        self.mask_epilogue = getattr(c, 'mask_epilogue', os.environ.get('RELNET_TRAIN_MASK_EPI', '1') != '0') and not self.trunk_fp32
        # weight-gradient products of the trunk launched on a side stream every `wgrad_overlap` units (0, the default: one grouped
        # launch per gradient bucket on the main stream): built to let the persistent stream-K kernel fill the CUs that the
        # data-gradient GEMMs of a 19 152-pixel map leave idle.  Measured (r05, same box, 8 images, ms per step): off 20.27, every
        # 8 units 20.37, 4: 20.52, 2: 20.95, 1: 21.71 -- the co-resident workgroups slow the GEMMs by more than the overlap returns
        # and smaller groups lose stream-K efficiency; kept as a knob (cfg.wgrad_overlap / RELNET_WGRAD_OVERLAP)
        self.wgrad_overlap = int(getattr(c, 'wgrad_overlap', os.environ.get('RELNET_WGRAD_OVERLAP', '0')))
        self._wgrad_side = torch.cuda.Stream(device=dev) if (self.wgrad_overlap > 0 and self._side is not None) else None
        self._wgrad_keep, self._wgrad_pending = [], False
        self._scratch_bufs = {}

    # ---- accessors ----------------------------------------------------------------------------------------
    def _trunk32(self, name):
        return self.trunk_fp32 and name.startswith('res') and not name.endswith('_offset')

    def w(self, name):          # bf16 working copy (the float32 master weights for the trunk layers of a cfg.trunk_fp32 trainer)
        return self.W.view(self.W.master if self._trunk32(name) else self.W.work, name)

    def b(self, name):          # fp32 bias
        return self.Bv.view(self.Bv.master, name)

    def wt(self, name):         # data-gradient layout of w(name), refreshed by _relayout.run() at the start of every step
        if self._trunk32(name):
            return None         # (float32 trunk: train_ops transposes / flips the master weights on the fly)
        return self._relayout.get(name)

    def num_trainable(self):
        return sum(int(np.prod(s)) for _, s in self.W.slices.values()) + sum(int(np.prod(s)) for _, s in self.Bv.slices.values())

    def _add_wgrad(self, name, dw, scale_rows=None):
        """dw: the gradient [rows, cols], or the split-K partial sums [splits, rows, cols] of train_ops.wgrad(keep_splits=True)
        (summed, scaled by the folded-BN factor and accumulated by ONE kernel)."""
        if dw is None:              # already accumulated by relnet_wgrad (wgrad_to)
            return
        g = self.W.view(self.W.grad, name)
        if dw.dim() == 3 and dw.is_contiguous() and dw.shape[1] * dw.shape[2] == g.numel() and g.shape[-1] % 4 == 0:
            T.wgrad_accumulate(dw, g, scale_rows)
            return
        if dw.dim() == 3:
            dw = dw.sum(0)
        dw = dw.reshape(g.shape)
        if scale_rows is not None:
            dw = dw * (scale_rows * scale_rows).view(-1, 1)
        g.add_(dw)

    def _wg(self, name, scale_rows=None):
        """(2-D fp32 view of `name`'s slice of the flat gradient buffer, folded-BatchNorm row factor | None): the target
        relnet_wgrad accumulates into (train_ops `wgrad_to` protocol)."""
        g = self.W.view(self.W.grad, name)
        return g.view(g.shape[0], -1), scale_rows, self._wq       # the products wait there for ONE grouped stream-K launch per gradient bucket

    def _flush_wgrads(self, final=True):
        """Launch the queued weight-gradient products (csrc/wgrad.hip, one grouped launch).  With a weight-gradient side stream the
        launch goes there (fork: side waits for everything queued on the current stream); final=True also joins -- the current stream
        waits for every product launched on the side stream since the last join (a gradient bucket is complete only then).  The
        operand tensors stay referenced until the join: their blocks must not be handed out again on the main stream while the
        side stream still reads them."""
        self._cq.flush()             # the queued bias-gradient column sums: one grouped launch
        if self._aux_pending:        # deferred parameter-gradient work: the bucket is complete only once it has run
            torch.cuda.current_stream().wait_stream(self._aux)
            self._aux_pending, self._aux_keep = False, []
        side = self._wgrad_side
        if side is None:
            self._wq.flush()
            return
        main = torch.cuda.current_stream()
        if len(self._wq):
            side.wait_stream(main)
            self._wgrad_keep.append(list(self._wq.keep))
            with torch.cuda.stream(side):
                self._wgrad_keep.append(self._wq.flush(workgroups=int(os.environ.get('RELNET_WGRAD_SIDE_WGS', '0'))))
            self._wgrad_pending = True
        if final and self._wgrad_pending:
            main.wait_stream(side)
            self._wgrad_pending = False
            self._wgrad_keep = []

    def _defer(self, fn, keep):
        """Run fn() on the auxiliary stream after everything queued so far (GradSink.defer); joined in _flush_wgrads.  `keep`: the tensors fn
        reads -- referenced until the join, and marked as used on that stream for the caching allocator (outside a capture)."""
        if self._aux is None:
            fn()
            return
        main = torch.cuda.current_stream()
        self._aux.wait_stream(main)
        with torch.cuda.stream(self._aux):
            fn()
        if not torch.cuda.is_current_stream_capturing():
            for t in keep:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self._aux)
        self._aux_keep.append(keep)
        self._aux_pending = True

    def _scratch(self, name, shape, dtype):
        """Persistent zero-initialised work buffer (created on first use, i.e. in the eager warm-up step, never inside a capture)."""
        key = (name, tuple(shape), dtype)
        if key not in self._scratch_bufs:
            self._scratch_bufs[key] = torch.zeros(shape, device=self.device, dtype=dtype)
        return self._scratch_bufs[key]

    def _add_bgrad(self, name, db):
        if db is not None:          # (None: already accumulated by T.colsum_add through _bg(name))
            self.Bv.view(self.Bv.grad, name).add_(db.reshape(-1))

    def _bg(self, name):
        """fp32 view of `name`'s slice of the flat bias-gradient buffer: the target T.colsum_add / linear_bwd(bgrad_to=) accumulate into."""
        return self.Bv.view(self.Bv.grad, name).view(-1)

    # ---- forward -------------------------------------------------------------------------------------------
    def _conv(self, x, name, stride=1, pad=0, dil=1, relu=False, resid=None, bias=None, out_dtype=None, w=None):
        w = self.w(name) if w is None else w
        bias = self.conv_bias[name] if bias is None else bias
        return ops.conv2d_nhwc(x, w, bias, ksize=self.ksize.get(name, 1), stride=stride, pad=pad, dil=dil, relu=relu,
                               resid=resid, out_dtype=out_dtype)

    def _trunk_forward(self, data, at_conv4=None):
        """Frozen stem + res2, then res3..res5 keeping every ReLU output.  Returns (conv5, conv4, saved units, stage ends).
        at_conv4(conv4): called once conv4 exists, before res5 is queued (the RPN branch forks there)."""
        c = self.cfg
        fb = self._frozen_backbone
        if self.trunk_fp32:
            assert fb is not None
            x = fb.forward_res2(data)                 # float32 NHWC (Backbone impl 'hip32')
        else:
            x = fb.forward_res2(data) if fb is not None else ops.stem_fused(data, self.w_stem, self.b_stem)
        saved = []
        conv4 = None
        ends = {}
        y_pre = None               # reduce output of the coming unit when the previous unit's chain kernel already produced it
        for stage, nm, ic, mc, oc, stride, dil, proj in self.units:
            n1, na, nb, nc = 'res%s_branch1' % nm, 'res%s_branch2a' % nm, 'res%s_branch2b' % nm, 'res%s_branch2c' % nm
            if stage == 5 and conv4 is None:
                conv4 = x
                if at_conv4 is not None:
                    at_conv4(conv4)
            if proj and stage > 2:
                ends[stage - 1] = x
            if stage == 2 and fb is not None:
                continue                   # (ran inside forward_res2)
            if stage == 2:
                fw = lambda n: self.frozen[n]
                sc = self._conv(x, n1, stride=stride, w=fw(n1)) if proj else x
                y = self._conv(x, na, stride=stride, relu=True, w=fw(na))
                y = self._conv(y, nb, pad=dil, dil=dil, relu=True, w=fw(nb))
                x = self._conv(y, nc, relu=True, resid=sc, w=fw(nc))
                continue
            sc = self._conv(x, n1, stride=stride) if proj else x
            y1 = y_pre if y_pre is not None else self._conv(x, na, stride=stride, relu=True)
            y_pre = None
            off = None
            if c.dcn and stage == 5:       # 72-channel offset conv -> DeformableConvolution(num_deformable_group=4) + BN + ReLU
                off = self._conv(y1, nb + '_offset', pad=2, dil=2, bias=self.b(nb + '_offset'), out_dtype=torch.float32)
                y2, col = ops.deformable_conv(y1.permute(0, 3, 1, 2), off.permute(0, 3, 1, 2), self.w(nb), self.conv_bias[nb],
                                              3, 1, 2, 2, 4, relu=True, want_col=True)
                y2 = y2.permute(0, 2, 3, 1)
                off = (off, col)           # (the sampled column matrix rides along to the backward: the weight gradient's X operand)
            else:
                y2 = self._conv(y1, nb, pad=dil, dil=dil, relu=True)
            ch = self.chain_units.get(nm)
            if ch is not None and ops.chain_worthwhile(y2.numel() // y2.shape[-1], mc):
                # expand + shortcut + ReLU and (inside a stage) the NEXT unit's reduce + ReLU in one pixel-wise kernel; every activation
                # the backward needs (x_next, mid1_next) is still written, nothing is overwritten in place
                w1f = self._fragpack.get('w1:' + ch[1]) if ch[0] else None
                b1 = self.conv_bias['res%s_branch2a' % ch[1]] if ch[0] else None
                out, y_pre = ops.bottleneck_chain(y2, sc.contiguous(), self._fragpack.get('w3:' + nm), w1f, self.conv_bias[nc], b1)
            else:
                out = self._conv(y2, nc, relu=True, resid=sc)
            saved.append((stage, nm, stride, dil, proj, x, y1, y2, out, off))
            x = out
        ends[5] = x
        return x, conv4, saved, ends

    def rpn_targets(self, gt_boxes, num_gt, im_info, feat_hw):
        """lib/rpn/rpn.py:80-244 (`assign_anchor`, host numpy inside the reference's AnchorLoader) on the device for the whole
        batch: one C-ABI call, no host round trip.  The random fg / bg subset is keyed by cfg.seed + a device step counter that
        is advanced here, i.e. also by every replay of a captured step."""
        c = self.cfg
        out = ops.assign_anchor(gt_boxes, num_gt, im_info, self.anchors_host, feat_hw, c.feat_stride, c.rpn_batch_size,
                                getattr(c, 'rpn_fg_fraction', 0.5), getattr(c, 'rpn_negative_overlap', 0.3),
                                getattr(c, 'rpn_positive_overlap', 0.7), seed=getattr(c, 'seed', 0) * self._world + self._rank, seed_dev=self._anchor_step)
        self._anchor_step += 1
        return out

    def forward_backward(self, *args, **kwargs):
        """One forward + backward pass (arguments: see `_forward_backward_impl` of the trainer class).  While it runs, bias-gradient column sums
        are queued (train_ops.COLSUM_QUEUE) and launched grouped when a gradient bucket completes (`_flush_wgrads`)."""
        T.COLSUM_QUEUE = self._cq if self._colsum_grouped else None
        try:
            return self._forward_backward_impl(*args, **kwargs)
        finally:
            self._cq.flush()            # (nothing is left unless a caller skipped the bucket announcements)
            T.COLSUM_QUEUE = None

    def _forward_backward_impl(self, data, im_info, gt_boxes, rpn_label=None, rpn_bbox_target=None, rpn_bbox_weight=None, num_gt=None):
        """data [B,3,H,W] fp32; gt_boxes [B,G,5]; rpn_label [B, A*h*w] ((a,y,x) order), rpn_bbox_target / weight
        [B, 4A, h, w] (lib/rpn/rpn.py:assign_anchor layouts) -- or None: computed on the device from gt_boxes
        (`rpn_targets`).  Accumulates gradients into the flat buffers and returns the loss values (reference metric names)."""
        c = self.cfg
        B = data.shape[0]
        self._grad_buckets().reset()
        self.W.grad.zero_(); self.Bv.grad.zero_()
        # W^T / tap-flipped copies of the current weights for every data-gradient product: only the BACKWARD reads them.  With the RPN branch on
        # its side stream they are made after conv_new_1, where the main stream otherwise waits for the proposals (one image: the branch's
        # single-workgroup sort / NMS kernels are the longer path and leave the chip idle); otherwise here
        late_relayout = (self._side is not None and getattr(self, 'overlap_rpn', os.environ.get('RELNET_TRAIN_OVERLAP', '1') != '0')
                         and os.environ.get('RELNET_LATE_RELAYOUT', '1') != '0')
        if not late_relayout:
            self._relayout.run()
        self._fragpack.run()            # the fragment-order copies the chain kernels of the forward read
        # The RPN branch (head convolutions, anchor targets, losses, proposals, proposal targets: everything that hangs off conv4)
        # runs on a side stream BESIDE res5 / conv_new_1 -- its top-k / sort / NMS / target kernels are one workgroup per image and
        # leave the GPU idle on their own; fork when conv4 exists, join before ROI pooling (graph edges under capture).
        out = {}
        br = {}
        main = torch.cuda.current_stream()
        nchw = lambda t: t.permute(0, 3, 1, 2)

        def _conv4_hw(n):                  # spatial size of conv4 (stride 16): conv1 7x7 / 2 pad 3, pool1 3x3 / 2 (ceil), res3a and res4a stride 2
            n = (n + 2 * 3 - 7) // 2 + 1
            n = -(-(n - 3) // 2) + 1
            n = (n - 1) // 2 + 1
            return (n - 1) // 2 + 1
        early = {}
        if rpn_label is None and getattr(self, 'overlap_rpn', os.environ.get('RELNET_TRAIN_OVERLAP', '1') != '0') and self._side is not None:
            # the anchor targets depend on the ground truth only (lib/rpn/rpn.py:assign_anchor runs in the LOADER in the reference): on the side stream
            # from the very start of the step, beside the stem -- their 0.23 ms (8 images) used to sit between the RPN head and the proposals, on
            # the longer of the two branches that meet at ROI pooling
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                early['t'] = self.rpn_targets(gt_boxes, num_gt, im_info, (_conv4_hw(data.shape[2]), _conv4_hw(data.shape[3])))

        def rpn_branch(conv4):
            r = self._conv(conv4, 'rpn_conv_3x3', pad=1, relu=True, bias=self.b('rpn_conv_3x3'))
            rpn = self._conv(r, 'rpn_out', bias=self.b('rpn_out'), out_dtype=torch.float32)             # [B,h,w,72]
            h, wd_ = rpn.shape[1], rpn.shape[2]
            na2 = self.na2
            lbl, tgt_in, wgt_in = rpn_label, rpn_bbox_target, rpn_bbox_weight
            if lbl is None and early.get('t') is not None and tuple(early['t'][1].shape[2:]) == (h, wd_):
                lbl, tgt_in, wgt_in = early['t']           # (computed on this stream at the start of the step)
            elif lbl is None:
                lbl, tgt_in, wgt_in = self.rpn_targets(gt_boxes, num_gt, im_info, (h, wd_))
            # -- RPN losses (per image, like one image per device in the reference)
            score_nchw = rpn[..., :na2].permute(0, 3, 1, 2).contiguous()                                 # [B,2A,h,w]
            # per-image 'valid' normalisation (one image per executor in the reference) in ONE launch: group = one image's anchors
            _, d_score = losses.softmax_output(score_nchw.view(B, 2, -1), lbl, multi_output=True, use_ignore=True,
                                               ignore_label=-1.0, group=(na2 // 2) * h * wd_)
            d_score = d_score.view(B, na2, h, wd_)
            delta = rpn[..., na2:].contiguous()                                                          # NHWC [B,h,w,4A]
            tgt = tgt_in.permute(0, 2, 3, 1).contiguous()
            wgt = wgt_in.permute(0, 2, 3, 1).contiguous()
            rpn_l1, d_delta = losses.smooth_l1_loss(delta, tgt, wgt, 3.0, 1.0 / c.rpn_batch_size)
            out['rpn_bbox_loss'] = T.scalar_sum(rpn_l1, 1.0 / B)
            d_rpn = torch.cat([d_score.permute(0, 2, 3, 1), d_delta], 3).to(torch.bfloat16).contiguous()
            # -- proposals and their targets (no gradient: proposal.py:170-173, proposal_target.py:95-97)
            rois, _ = propose_batch(nchw(rpn[..., :na2]), nchw(rpn[..., na2:]), im_info, self.anchors, c.feat_stride,
                                    c.rpn_pre_nms_top_n, c.rpn_post_nms_top_n, c.rpn_nms_thresh, c.rpn_min_size,
                                    im_hw=self.im_hw, softmax_pairs=True)
            N = rois.shape[1]
            rois_t, label, bbox_target, bbox_weight = ops.proposal_target(rois, gt_boxes, num_gt)
            R = rois_t.shape[1]
            br.update(r=r, d_rpn=d_rpn, rois_t=rois_t, label=label, bbox_target=bbox_target, bbox_weight=bbox_weight, N=N, R=R)
            if rpn_bwd_side:
                br['ev'] = torch.cuda.Event()
                br['ev'].record()               # main resumes here (rois and their targets exist); the RPN head's backward follows on this stream
                if late_relayout:
                    br['bwd'] = (conv4, r, d_rpn)     # ... once the main stream has made the weight copies it reads (enqueued below)
                else:
                    rpn_backward(conv4, r, d_rpn)

        def rpn_backward(conv4, r, d_rpn):
            # RPN head backward (joins the trunk at conv4).  It depends on the RPN losses only: with the branch on its side stream it runs there,
            # beside ROI pooling / the 2FC head / the learn-NMS branch (launches of <= 150 workgroups), and with it the two RPN weight gradients as
            # their own grouped launch on 64 workgroups (a quarter of the chip: the head's kernels beside it keep theirs);
            # RELNET_RPN_WGRAD_SIDE=0 leaves those in the 'heads' bucket's launch on the main stream
            nwg = int(os.environ.get('RELNET_RPN_WGRAD_SIDE', '64')) if rpn_bwd_side else 0
            rq = ops.WgradQueue() if nwg else None
            wg = (lambda n: self._wg(n)[:2] + (rq,)) if nwg else self._wg
            g_r, dw = T.conv1x1_bwd(r, self.w('rpn_out'), d_rpn, w_t=self.wt('rpn_out'), keep_splits=True, wgrad_to=wg('rpn_out'), relu_mask=r)
            T.colsum_add(d_rpn, self._bg('rpn_out'))
            d_conv4_rpn, dw = T.conv3x3_bwd(conv4, self._dgrad_w('rpn_conv_3x3', 512), g_r, dil=1, keep_splits=True, wgrad_to=wg('rpn_conv_3x3'))
            T.colsum_add(g_r, self._bg('rpn_conv_3x3'))
            br.update(g_r=g_r, d_conv4_rpn=d_conv4_rpn)
            if nwg:
                br['rq_keep'] = rq.flush(workgroups=nwg)

        rpn_bwd_side = False
        side = None
        if getattr(self, 'overlap_rpn', os.environ.get('RELNET_TRAIN_OVERLAP', '1') != '0'):
            side = self._side
            rpn_bwd_side = os.environ.get('RELNET_RPN_BWD_SIDE', '1') != '0' and not self.lnms_only

            def fork(conv4):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    rpn_branch(conv4.to(torch.bfloat16))
            conv5, conv4, saved, _ = self._trunk_forward(data, at_conv4=fork)
        else:
            conv5, conv4, saved, _ = self._trunk_forward(data)
            rpn_branch(conv4.to(torch.bfloat16))
        # (a float32 trunk -- cfg.trunk_fp32, parity tests -- hands bf16 copies to the bf16 heads and takes their gradients back as float32)
        conv5, conv4 = conv5.to(torch.bfloat16), conv4.to(torch.bfloat16)
        feat = self._conv(conv5, 'conv_new_1', relu=True, bias=self.b('conv_new_1'))
        if late_relayout:
            self._relayout.run()
            if br.get('bwd') is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    rpn_backward(*br.pop('bwd'))
        if side is not None:
            if rpn_bwd_side:
                main.wait_event(br['ev'])
            else:
                main.wait_stream(side)
            # the branch's tensors were allocated on the side stream and are consumed on the main one through the backward pass:
            # tell the caching allocator, so that their blocks are not handed out again on the side stream while main still reads them
            if not torch.cuda.is_current_stream_capturing():      # (a capture's private pool keeps its blocks for the graph's lifetime)
                for v in br.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(main)
        r, d_rpn, rois_t, label, bbox_target, bbox_weight, N, R = (br[k] for k in ('r', 'd_rpn', 'rois_t', 'label', 'bbox_target',
                                                                                      'bbox_weight', 'N', 'R'))
        r5 = rois_t.view(B * R, 5)
        if c.dcn:
            sc_ = 1.0 / c.feat_stride
            t0 = ops.deformable_psroi_pool(nchw(feat), r5, None, sc_, feat.shape[3], 1, 7, 7, c.dcn_sample_per_part, 0.0, True,
                                           channels_last_out=True)
            t0f = t0.permute(0, 2, 3, 1).reshape(B * R, -1)
            trans = ops.gemm_nt(t0f, self.w('offset'), self.b('offset'), out_dtype=torch.float32).view(B * R, 2, 7, 7)
            pooled = ops.deformable_psroi_pool(nchw(feat), r5, trans, sc_, feat.shape[3], 1, 7, 7, c.dcn_sample_per_part,
                                               c.dcn_trans_std, False, channels_last_out=True)
            argmax = None
        else:
            pooled, argmax = ops.roi_pool(nchw(feat), r5, (7, 7), 1.0 / c.feat_stride, channels_last_out=True, want_argmax=True)
        pooled2 = pooled.permute(0, 2, 3, 1).reshape(B * R, -1)
        d_pool, hs = self._head_forward_backward(pooled2, rois_t, N, label, bbox_target, bbox_weight, im_info, gt_boxes, num_gt, out)
        bt = torch.bfloat16
        x2, f1, cls_score, bbox_pred, labels_ohem, weights_ohem = hs
        if self.lnms_only:          # JOINT_TRAINING false: the detector is fixed, the step ends with the learn-NMS head's parameter gradients
            self._bucket_ready('heads')
            out['rois'], out['label'] = rois_t, labels_ohem
            out['bbox_target'], out['bbox_weight'] = bbox_target, weights_ohem
            out['bbox_pred'], out['cls_score'], out['fc_all_2_relu'] = bbox_pred, cls_score, x2
            return out
        # ROIPooling backward -> gradient of conv_new_1_relu
        if c.dcn:
            gp = d_pool.view(B * R, 7, 7, -1).permute(0, 3, 1, 2)
            gd1, gtrans = ops.deformable_psroi_pool_bwd(gp, nchw(feat), r5, trans, sc_, feat.shape[3], 1, 7, 7,
                                                        c.dcn_sample_per_part, c.dcn_trans_std, False)
            d_t0, dw, db = T.linear_bwd(t0f, self.w('offset'), gtrans.view(B * R, -1).to(bt), w_t=self.wt('offset'), keep_splits=True, wgrad_to=self._wg('offset'))
            self._add_bgrad('offset', db)
            gd2, _ = ops.deformable_psroi_pool_bwd(d_t0.view(B * R, 7, 7, -1).permute(0, 3, 1, 2), nchw(feat), r5, None, sc_,
                                                   feat.shape[3], 1, 7, 7, c.dcn_sample_per_part, 0.0, True)
            d_feat = (gd1 + gd2).permute(0, 2, 3, 1).to(bt).contiguous()       # logical NCHW stored NHWC -> NHWC bf16
        else:
            d_feat = ops.roi_pool_bwd(d_pool.view(B * R, 7, 7, -1).permute(0, 3, 1, 2), argmax, r5,
                                      (B, feat.shape[3], feat.shape[1], feat.shape[2]), channels_last=True)
            d_feat = d_feat.permute(0, 2, 3, 1).to(bt)          # NHWC memory already: one conversion pass
        g = T.relu_bwd(d_feat, feat)
        d_x, dw = T.conv1x1_bwd(conv5, self.w('conv_new_1'), g, w_t=self.wt('conv_new_1'), keep_splits=True, wgrad_to=self._wg('conv_new_1'))
        T.colsum_add(g, self._bg('conv_new_1'))
        # RPN head backward (joins the trunk at conv4)
        if rpn_bwd_side:
            main.wait_stream(side)
        else:
            rpn_backward(conv4, r, d_rpn)
        d_conv4_rpn = br['d_conv4_rpn']
        if self.trunk_fp32:
            d_x, d_conv4_rpn = d_x.float(), d_conv4_rpn.float()
        self._trunk_backward(saved, d_x, {'4b22': d_conv4_rpn})
        out['rois'] = rois_t
        out['label'] = labels_ohem
        out['bbox_target'], out['bbox_weight'] = bbox_target, weights_ohem
        out['bbox_pred'] = bbox_pred
        out['cls_score'] = cls_score
        return out

    def _trunk_backward(self, saved, d_x, inject):
        """res5 -> res3 backward.  d_x = gradient of the last unit's output; inject[unit] = extra gradient of that unit's output."""
        B = saved[0][5].shape[0]
        bt = torch.bfloat16
        # trunk: res5 -> res3.  Gradient buckets are announced as they complete (heads first; with DCN the res5 offset
        # convolutions live in the heads bucket, which is then complete only after res5): dist.BucketedAllReduce overlaps
        # each bucket's all-reduce with the rest of the backward pass
        dcn = self.cfg.dcn
        self._grad_buckets()
        if not dcn:
            self._bucket_ready('heads')
        prev_bucket = None
        masked = False            # d_x already carries the ReLU mask of the unit's output (relnet_gemm_nt_mask of the unit after it)
        since_flush = 0
        order = list(reversed(saved))
        for pos, (stage, nm, stride, dil, proj, x_in, y1, y2, o, off) in enumerate(order):
            bucket = self._unit_bucket[nm]
            if prev_bucket is not None and bucket != prev_bucket:
                if dcn and prev_bucket == 'res5':
                    self._bucket_ready('res5', 'heads')        # both complete at this point: one cut, two collectives
                else:
                    self._bucket_ready(prev_bucket)
                since_flush = 0
            prev_bucket = bucket
            n1, na, nb, nc_ = 'res%s_branch1' % nm, 'res%s_branch2a' % nm, 'res%s_branch2b' % nm, 'res%s_branch2c' % nm
            if inject.get(nm) is not None:       # a second consumer of this unit's output (RPN head at conv4, FPN laterals)
                assert not masked
                d_x = inject[nm] if d_x is None else d_x + inject[nm]
            g_out = d_x if masked else T.relu_bwd(d_x, o)
            masked = False
            # (ReLU masks of the two inner activations ride in the data-gradient kernels' epilogues)
            g_y2, dw = T.conv1x1_bwd(y2, self.w(nc_), g_out, w_t=self.wt(nc_), keep_splits=True, wgrad_to=self._wg(nc_, self.bn_scale[nc_]), relu_mask=y2)
            if off is not None:            # deformable branch2b: data + offset gradients, then the offset conv's own backward
                off, col = off
                gd, goff, dw = ops.deformable_conv_bwd(y1.permute(0, 3, 1, 2), off.permute(0, 3, 1, 2), self.w(nb),
                                                       g_y2.permute(0, 3, 1, 2), 3, 1, 2, 2, 4, col=col)
                self._add_wgrad(nb, dw, self.bn_scale[nb])
                no = nb + '_offset'
                goff_p = torch.zeros((B, off.shape[1], off.shape[2], 128), device=off.device, dtype=bt)   # 72 -> 128 channels
                goff_p[..., :72] = goff
                wd_ = self.w(no).view(72, 3, 3, -1).flip(1, 2).permute(3, 1, 2, 0)                     # [Cin,3,3,72]
                wdp = torch.zeros((wd_.shape[0], 3, 3, 128), device=off.device, dtype=bt); wdp[..., :72] = wd_
                d_off_in, dwo = T.conv3x3_bwd(y1, wdp.reshape(wd_.shape[0], -1).contiguous(), goff_p, dil=2, keep_splits=True)
                self._add_wgrad(no, dwo.sum(0)[:72]); self._add_bgrad(no, goff.sum((0, 1, 2)))
                d_y1 = (gd + d_off_in.float()).to(bt).contiguous()
            else:
                g_y1, dw = T.conv3x3_bwd(y1, self._dgrad_w(nb, y2.shape[3]), g_y2, dil=dil, keep_splits=True, wgrad_to=self._wg(nb, self.bn_scale[nb]),
                                         relu_mask=y1)
            if off is not None:
                g_y1 = T.relu_bwd(d_y1, y1)
            first = (stage == 3 and proj)         # res3a: its input comes from the frozen res2 -> no data gradient
            if proj:
                # both branches read the same (sampled) input: their data gradients are summed in the second GEMM's epilogue -- at the SAMPLED
                # resolution for a stride-2 unit, then scattered to the input's resolution once, with the previous unit's ReLU mask in the same
                # pass when that output has no other consumer (was: 2 zero fills + 2 strided copies + a full-resolution add + relu_bwd)
                prev_nm = order[pos + 1][1] if pos + 1 < len(order) else None
                fold_on = os.environ.get('RELNET_TRAIN_FOLD', '1') != '0'          # (A/B switch: 0 = the separate passes)
                lr = stride != 1 and fold_on
                # a second consumer's gradient of this unit's INPUT (the RPN head at conv4, in front of the stride-1 res5a) rides in the first
                # GEMM's epilogue; the input's ReLU mask can then ride in the second one's
                inj = inject.get(prev_nm) if (fold_on and stride == 1 and not first and prev_nm is not None) else None
                fold = inj is not None and inj.dtype == g_y1.dtype and tuple(inj.shape) == tuple(x_in.shape) and inj.is_contiguous()
                d_a, dw = T.conv1x1_bwd(x_in, self.w(na), g_y1, stride=stride, need_dx=not first, w_t=self.wt(na), keep_splits=True, wgrad_to=self._wg(na, self.bn_scale[na]),
                                        dx_add=inj if fold else None, low_res=lr)
                if fold:
                    inject = {k: v for k, v in inject.items() if k != prev_nm}
                can_mask = self.mask_epilogue and not first and prev_nm is not None and inject.get(prev_nm) is None
                d_s, dw = T.conv1x1_bwd(x_in, self.w(n1), g_out, stride=stride, need_dx=not first, w_t=self.wt(n1),
                                        dx_add=d_a if (not first and (lr or stride == 1)) else None, keep_splits=True, wgrad_to=self._wg(n1, self.bn_scale[n1]), low_res=lr,
                                        out_mask=x_in if (can_mask and fold_on and stride == 1) else None)
                d_x = None if first else (d_s if (lr or stride == 1) else d_s + d_a)
                masked = can_mask and fold_on and stride == 1
                if lr and not first:
                    d_x = T.strided_scatter(d_s, tuple(x_in.shape), stride, mask=x_in if can_mask else None)
                    masked = can_mask
            else:
                # identity shortcut: d x_in = g_y1 W1 + g_out; x_in is the previous unit's ReLU output, so its mask can ride in this GEMM's
                # epilogue -- unless that output has a second consumer whose gradient must be added before the mask (inject)
                prev_nm = order[pos + 1][1] if pos + 1 < len(order) else None
                can_mask = self.mask_epilogue and prev_nm is not None and inject.get(prev_nm) is None
                d_x, dw = T.conv1x1_bwd(x_in, self.w(na), g_y1, dx_add=g_out, w_t=self.wt(na), keep_splits=True, wgrad_to=self._wg(na, self.bn_scale[na]),
                                        out_mask=x_in if can_mask else None)
                masked = can_mask
            since_flush += 1
            if self._wgrad_side is not None and since_flush >= self.wgrad_overlap:
                self._flush_wgrads(final=False)       # this group of weight gradients starts now, beside the next units' data gradients
                since_flush = 0
        self._bucket_ready(prev_bucket)

    def _head_forward_backward(self, pooled2, rois_t, N, label, bbox_target, bbox_weight, im_info, gt_boxes, num_gt, out,
                               key_count=None):
        """2FC head + relation modules + OHEM + losses (+ learn-NMS branch) and their adjoint down to the pooled features.
        pooled2 [B*R, 12544] bf16 ((ph, pw, c) column order), rois_t [B,R,5] with the N non-gt rows first.
        key_count [B] int32 (optional): how many of the first N rows of each image are real proposals (the rest is padding
        of a truncated / short proposal list: no relation keys, label -1, never ranked by the learn-NMS branch).
        Returns (d_pool [B*R,12544] bf16, (x2, f1, cls_score, bbox_pred, labels_ohem, weights_ohem))."""
        c = self.cfg
        B, R = rois_t.shape[0], rois_t.shape[1]
        # keys of both relation modules = the first N rows of each image
        bt = torch.bfloat16
        if self.relation:
            mods = [self._rel_params(i) for i in (1, 2)]
            wp_t, bp = pack_pair_pos(mods, self.device)
            # float32 ln G for the training forward (not the fp16 matrix-core bias of inference): the backward needs exactly this G
            # and the outputs computed from it (relation.attention_module_backward), so they are produced once, here
            bias = ops.geometry_bias(rois_t, wp_t, bp, N, fast32=True)
            f1 = ops.gemm_nt(pooled2, self.w('fc_new_1'), self.b('fc_new_1')).reshape(B, R, -1)
            caches = [{}, {}]           # Q|K and VW^T projections of the forward, reused by the backward
            # (VW^T buffers: persistent -- only columns [:N] are written, the pad columns stay zero from step to step: no per-step fill)
            vw_ = lambda i: self._scratch('vwt_%d_%d' % (i, N), (B, mods[0].wout.shape[0], bias.shape[-1]), bt)     # keyed on N: a smaller N in the same padded width would inherit stale columns
            _, x1, _ = _module_forward(f1, mods[0], bias[0], N, True, True, False, vwt_buf=vw_(1), key_count=key_count, cache=caches[0])
            f2 = ops.gemm_nt(x1.reshape(B * R, -1), self.w('fc_new_2'), self.b('fc_new_2')).reshape(B, R, -1)
            _, x2, _ = _module_forward(f2, mods[1], bias[1], N, True, True, False, vwt_buf=vw_(2), key_count=key_count, cache=caches[1])
        else:       # plain 2FC head (resnet_v1_101_rcnn.py:96-174 / ..._learn_nms_1024_...:176-216): fc_new_i + ReLU in the GEMM epilogue
            x1 = ops.gemm_nt(pooled2, self.w('fc_new_1'), self.b('fc_new_1'), relu=True).reshape(B, R, -1)
            x2 = ops.gemm_nt(x1.reshape(B * R, -1), self.w('fc_new_2'), self.b('fc_new_2'), relu=True).reshape(B, R, -1)
            f1 = x1
        cb = ops.gemm_nt(x2.reshape(B * R, -1), self.w('cls_bbox'), self.b('cls_bbox'), out_dtype=torch.float32).reshape(B, R, -1)
        nc = self.num_classes
        cls_score, bbox_pred = cb[:, :, :nc].contiguous(), cb[:, :, nc:].contiguous()
        if getattr(c, 'enable_ohem', True):
            labels_ohem, weights_ohem = ops.box_annotator_ohem(cls_score, bbox_pred, label, bbox_target, bbox_weight, c.batch_rois_ohem)
            box_norm = c.batch_rois_ohem
        else:       # ENABLE_OHEM false (symbols/..._learn_nms_1024_...:232-241): every roi counts (padding rows of a short proposal list keep label -1
            labels_ohem, weights_ohem = label, bbox_weight          # and are ignored), the box loss is scaled by 1 / 300 (BATCH_ROIS < 0)
            box_norm = 300
        _, d_cls = losses.softmax_output(cls_score.view(B * R, -1), labels_ohem.reshape(-1), use_ignore=True, ignore_label=-1.0, group=R)
        d_cls = d_cls.view(B, R, -1)
        l1, d_bbox = losses.smooth_l1_loss(bbox_pred, bbox_target, weights_ohem, 1.0, 1.0 / box_norm)
        out['bbox_loss'] = T.scalar_sum(l1, 1.0 / B)
        out['num_ohem'] = T.scalar_sum(labels_ohem, count_nonneg=True)
        d_x2_lnms = None
        if c.learn_nms:     # the learn-NMS head sees the first N (non-gt) rows; its gradient joins cls_score and fc_all_2_relu
            d_cls_l, d_x2_lnms, lo = self._lnms_forward_backward(cls_score[:, :N], bbox_pred[:, :N], rois_t[:, :N].contiguous(),
                                                                 im_info, x2[:, :N], gt_boxes, num_gt, n_valid=key_count, d_cls_out=d_cls)
            if d_cls_l is not None:         # (None: already accumulated into d_cls[:, :N] by relnet_lnms_softmax_bwd)
                d_cls[:, :N] += d_cls_l
            out.update(lo)
        if self.lnms_only:          # every parameter in front of the learn-NMS head is fixed: nothing is differentiated past this point
            return None, (x2, f1, cls_score, bbox_pred, labels_ohem, weights_ohem)
        # ================= backward =================
        d_cb = torch.cat([d_cls, d_bbox], 2).reshape(B * R, -1).to(bt)
        d_x2, dw, db = T.linear_bwd(x2.reshape(B * R, -1), self.w('cls_bbox'), d_cb, w_t=self.wt('cls_bbox'), keep_splits=True, wgrad_to=self._wg('cls_bbox'), bgrad_to=self._bg('cls_bbox'))
        self._add_bgrad('cls_bbox', db)
        if d_x2_lnms is not None:
            d_x2 = d_x2.reshape(B, R, -1)
            d_x2[:, :N] += d_x2_lnms.to(d_x2.dtype)
        if self.relation:
            d_f2 = self._relation_bwd(2, mods[1], f2, x2, rois_t, d_x2.reshape(B, R, -1), N, key_count, caches[1])
        else:
            d_f2 = T.relu_bwd(d_x2.reshape(B, R, -1).contiguous(), x2)
        d_x1, dw, db = T.linear_bwd(x1.reshape(B * R, -1), self.w('fc_new_2'), d_f2.reshape(B * R, -1), w_t=self.wt('fc_new_2'), keep_splits=True, wgrad_to=self._wg('fc_new_2'), bgrad_to=self._bg('fc_new_2'))
        self._add_bgrad('fc_new_2', db)
        if self.relation:
            d_f1 = self._relation_bwd(1, mods[0], f1, x1, rois_t, d_x1.reshape(B, R, -1), N, key_count, caches[0])
        else:
            d_f1 = T.relu_bwd(d_x1.reshape(B, R, -1).contiguous(), x1)
        d_pool, dw, db = T.linear_bwd(pooled2, self.w('fc_new_1'), d_f1.reshape(B * R, -1), w_t=self.wt('fc_new_1'), keep_splits=True, wgrad_to=self._wg('fc_new_1'), bgrad_to=self._bg('fc_new_1'))
        self._add_bgrad('fc_new_1', db)
        return d_pool, (x2, f1, cls_score, bbox_pred, labels_ohem, weights_ohem)

    def _lnms_forward_backward(self, cls_score, bbox_pred, rois, im_info, feat, gt_boxes, num_gt, n_valid=None, d_cls_out=None):
        """Train branch of the learn-NMS head (symbols/..._learn_nms.py:424-551) and its adjoint.
        cls_score [B,N,81] fp32, bbox_pred [B,N,8] (BlockGrad), rois [B,N,5], feat = fc_all_2_relu[:, :N] bf16.
        Returns (d cls_score [B,N,81] fp32, d feat [B,N,1024] (bf16, as the projection's backward GEMM leaves it), losses); with d_cls_out [B,R,81] fp32 (contiguous, R >= N) the
        class-score gradient is accumulated into d_cls_out[:, :N] instead and None is returned in its place.
        cfg.lnms_fused_glue (default on; RELNET_LNMS_GLUE=0 = the tensor-operator chains of rounds 3 - 5, kept for the equality test):
        the element-wise chains of the branch run as single kernels (csrc/lnms_train.hip)."""
        import ctypes
        from . import lib as _lib
        c = self.cfg
        fused = bool(getattr(c, 'lnms_fused_glue', os.environ.get('RELNET_LNMS_GLUE', '1') != '0'))
        B, N, C1 = cls_score.shape
        C, F, Tn = C1 - 1, c.first_n, len(c.nms_target_thresh)
        dev, bt, s_ = cls_score.device, torch.bfloat16, ops._stream()
        cs, bp_ = cls_score.contiguous().view(B * N, C1), bbox_pred.contiguous().view(B * N, -1)
        prob = torch.empty((B, N, C), device=dev, dtype=torch.float32)
        boxes = torch.empty((B, N, 4), device=dev, dtype=torch.float32)
        means, stds = (ctypes.c_float * 4)(*c.bbox_means), (ctypes.c_float * 4)(*c.bbox_stds)
        _lib.call('relnet_lnms_prepare_ex', cs.data_ptr(), cs.stride(0), bp_.data_ptr(), bp_.stride(0), rois.data_ptr(),
                  im_info.data_ptr(), prob.data_ptr(), boxes.data_ptr(), B, N, C1, 4, means, stds, ops._ptr(n_valid), s_)
        rank_idx = torch.empty((B, C, F), device=dev, dtype=torch.int32)
        sorted_score = torch.empty((B, F, C), device=dev, dtype=torch.float32)
        sorted_bbox = torch.empty((B, F, C, 4), device=dev, dtype=torch.float32)
        class_boxes = torch.empty((B, C, F, 4), device=dev, dtype=torch.float32)
        class_max = torch.empty((B, C), device=dev, dtype=torch.float32)
        _lib.call('relnet_lnms_sort', prob.data_ptr(), boxes.data_ptr(), rank_idx.data_ptr(), sorted_score.data_ptr(),
                  sorted_bbox.data_ptr(), class_boxes.data_ptr(), class_max.data_ptr(), B, N, C, F, s_)
        rank_feat = ops.gemm_nt(self.rank_emb, self.w('nms_rank'), self.b('nms_rank'), out_dtype=torch.float32)      # [F,128]
        feat2 = feat.contiguous().view(B * N, -1)
        roi_emb = ops.gemm_nt(feat2, self.w('roi_feat_embedding'), self.b('roi_feat_embedding'))
        x = torch.empty((B, C, F, 128), device=dev, dtype=bt)
        _lib.call('relnet_lnms_embed', roi_emb.data_ptr(), rank_feat.data_ptr(), rank_idx.data_ptr(), x.data_ptr(),
                  B, N, C, F, 128, ops._dt(x), s_)
        BC = B * C
        xr = x.view(BC, F, 128)
        # relation module over (image, class): 16 heads x 64 for Q/K, each head's 8 output channels padded to a 64-wide tile
        class M_(object):
            pass
        mod = M_()
        mod.wqk, mod.bqk = self.w('nms_qk_1'), self.b('nms_qk_1')
        wo, bo = self.w('nms_linear_out_1'), self.b('nms_linear_out_1')
        mod.wout = self._scratch('nms_wout_pad', (1024, 128), bt)           # persistent: the 56 pad rows of every head stay zero
        mod.bout = self._scratch('nms_bout_pad', (1024,), torch.float32)
        if fused:       # the four padded operands of the step (Wout / bout per head, the 64-row logit matrix) refreshed by one launch
            w_logit, b_logit = self._scratch('nms_wlogit_pad', (64, 128), bt), self._scratch('nms_blogit_pad', (64,), torch.float32)
            _lib.call('relnet_lnms_pad_params', wo.data_ptr(), bo.data_ptr(), self.w('nms_logit').data_ptr(), self.b('nms_logit').data_ptr(),
                      mod.wout.data_ptr(), mod.bout.data_ptr(), w_logit.data_ptr(), b_logit.data_ptr(), Tn, s_)
        else:
            mod.wout.view(16, 64, 128)[:, :8] = wo.view(16, 8, 128)
            mod.bout.view(16, 64)[:, :8] = bo.view(16, 8)
        mod.wp = self.W.view(self.W.master, 'nms_pair_pos_fc1_1')
        mod.bp = self.b('nms_pair_pos_fc1_1')
        wp_t, bp = pack_pair_pos([mod], dev)
        cb = class_boxes.view(BC, F, 4)
        if fused and n_valid is None and getattr(c, 'lnms_geometry_table', os.environ.get('RELNET_LNMS_GEOM_TABLE', '1') != '0'):
            # ln G once per IMAGE (B N^2 pairs instead of B C F^2: 9 x fewer at 300 rois / 80 classes / first_n 100), then gathered per class by its ranks
            # (relnet_lnms_gather_bias): the class's boxes are the image's boxes re-ordered, same arithmetic on the same pairs -> bit-identical
            img = ops.geometry_bias(boxes, wp_t, bp, N, fast32=True)[0]                    # [B,16,N,Npad]
            bias = torch.empty((BC, 16, F, ops.pad32(F)), device=dev, dtype=torch.float32)
            _lib.call('relnet_lnms_gather_bias', img.data_ptr(), rank_idx.data_ptr(), bias.data_ptr(), B, C, N, img.shape[-1], F, bias.shape[-1], s_)
        else:
            bias = ops.geometry_bias(cb, wp_t, bp, F, fast32=True)[0]
        lcache = {}
        att, _, _ = _module_forward(xr, mod, bias, F, True, False, False, vwt_buf=self._scratch('vwt_nms_%d' % F, (BC, 1024, bias.shape[-1]), bt),
                                    cache=lcache)                                               # [BC,F,1024]
        if fused:
            assert att.is_contiguous() and att.shape == (BC, F, 1024) and att.dtype == bt and xr.is_contiguous()
            allf = torch.empty((BC, F, 128), device=dev, dtype=bt)
            _lib.call('relnet_lnms_residual_relu', att.data_ptr(), xr.data_ptr(), allf.data_ptr(), BC * F, s_)
            logit64 = ops.gemm_nt(allf.view(BC * F, 128), w_logit, b_logit, out_dtype=torch.float32)      # [BC*F, 64], T real columns
            cond = torch.empty((B, F, C, Tn), device=dev, dtype=torch.float32)
            multi = torch.empty_like(cond)
            _lib.call('relnet_lnms_cond_multi', logit64.data_ptr(), logit64.stride(0), sorted_score.data_ptr(), cond.data_ptr(), multi.data_ptr(),
                      B, C, F, Tn, s_)
        else:
            att128 = att.view(BC, F, 16, 64)[..., :8].reshape(BC, F, 128)
            allf = torch.relu(xr + att128).contiguous()
            w_logit = torch.zeros((64, 128), device=dev, dtype=bt); w_logit[:Tn] = self.w('nms_logit')
            b_logit = torch.zeros(64, device=dev, dtype=torch.float32); b_logit[:Tn] = self.b('nms_logit')
            logit = ops.gemm_nt(allf.view(BC * F, 128), w_logit, b_logit, out_dtype=torch.float32)[:, :Tn]
            cond = torch.sigmoid(logit).view(B, C, F, Tn).permute(0, 2, 1, 3).contiguous()          # [B,F,C,T]
            multi = sorted_score.unsqueeze(3) * cond
        target = ops.nms_multi_target(sorted_bbox, gt_boxes, sorted_score, num_gt, c.nms_target_thresh)
        pos, neg, d_multi = losses.nms_loss(multi, target, F, Tn, c.nms_loss_scale, c.nms_pos_scale, c.nms_eps)
        lo = dict(nms_pos_loss=T.scalar_sum(pos, 1.0 / B), nms_neg_loss=T.scalar_sum(neg, 1.0 / B), nms_multi_score=multi, nms_multi_target=target,
                  sorted_score=sorted_score, nms_rank_idx=rank_idx, nms_class_boxes=class_boxes)
        # ---------------- adjoint ----------------
        if fused:
            d_sorted = torch.empty((B, F, C), device=dev, dtype=torch.float32)
            d_logit_p = torch.empty((BC * F, 64), device=dev, dtype=bt)
            _lib.call('relnet_lnms_cond_bwd', d_multi.data_ptr(), cond.data_ptr(), sorted_score.data_ptr(), d_sorted.data_ptr(), d_logit_p.data_ptr(),
                      B, C, F, Tn, s_)
        else:
            d_sorted = (d_multi * cond).sum(3)                                                      # [B,F,C]
            d_logit = (d_multi * sorted_score.unsqueeze(3) * cond * (1.0 - cond)).permute(0, 2, 1, 3).reshape(BC * F, Tn)
            d_logit_p = torch.zeros((BC * F, 64), device=dev, dtype=bt); d_logit_p[:, :Tn] = d_logit
        wl_t = self.wt('nms_logit')
        if fused and wl_t is not None:
            # logit layer backward on the step's own machinery: W^T from the relayout table, the weight gradient in the bucket's grouped launch
            # (Tn real of 64 padded output columns), the bias gradient in its grouped column sum (was 8 library launches: transpose, fills, casts, sums)
            d_allf = ops.gemm_nt(d_logit_p, wl_t)
            T._wg_call(self._wg('nms_logit'), d_logit_p, allf.view(BC * F, 128), cout=Tn)
            T.colsum_add(d_logit_p[:, :Tn], self._bg('nms_logit'))
        else:
            d_allf, dw, db = T.linear_bwd(allf.view(BC * F, 128), w_logit, d_logit_p, w_t=None, keep_splits=True)
            self._add_wgrad('nms_logit', dw.sum(0)[:Tn]); self._add_bgrad('nms_logit', db[:Tn])
        g = T.relu_bwd(d_allf, allf.view(BC * F, 128))                                          # [BC*F,128] bf16
        dY = self._scratch('nms_dy_pad', (BC, F, 1024), bt)                   # persistent: columns 8 .. 63 of every head stay zero
        dY.view(BC, F, 16, 64)[..., :8] = g.view(BC, F, 16, 8)
        wcat_t = self._relayout.get('rel_cat_nms')
        if wcat_t is not None and getattr(c, 'relation_sink', os.environ.get('RELNET_REL_SINK', '1') != '0'):
            # gradients through relation.GradSink: [dQ | dK | dVW] packed once, ONE projection-backward GEMM (K = 3072) with the residual
            # gradient g in its epilogue; the [Wq; Wk] product goes straight into the flat buffer, the padded Wout product through a
            # [1024, 128] scratch whose 8 real rows per head are then added to the compact [128, 128] gradient
            gq = self.W.view(self.W.grad, 'nms_qk_1')
            glo = self._scratch('nms_dwout_pad', (1024, 128), torch.float32)
            glo.zero_()

            def wg(dy2d, x2d):
                ops.wgrad_tn(dy2d[:, :2048], x2d, out=gq.view(2048, 128))
                ops.wgrad_tn(dy2d[:, 2048:], x2d, out=glo)
            sink = GradSink(wcat_t, g.view(BC, F, 128), wg, self._bg('nms_qk_1'), None,
                            self.W.view(self.W.grad, 'nms_pair_pos_fc1_1'), self._bg('nms_pair_pos_fc1_1'), self._scratch, defer=self._defer)
            r = attention_module_backward(xr, cb, None, dY, nongt_dim=F, index=1, dtype=bt, packed=mod, cache=lcache, sink=sink)
            self.W.view(self.W.grad, 'nms_linear_out_1').view(16, 8, 128).add_(glo.view(16, 64, 128)[:, :8])
            T.colsum_add(g.view(BC * F, 128), self._bg('nms_linear_out_1'))    # (dY's real columns are g's columns)
            d_x = r['d_roi_feat'].view(B, C, F, 128)                            # bf16: residual + module
        else:
            r = attention_module_backward(xr, cb, None, dY, nongt_dim=F, index=1, dtype=bt, packed=mod, cache=lcache)
            self._add_wgrad('nms_qk_1', torch.cat([r['query_1_weight'], r['key_1_weight']], 0))
            self._add_bgrad('nms_qk_1', torch.cat([r['query_1_bias'], r['key_1_bias']], 0))
            self._add_wgrad('nms_linear_out_1', r['linear_out_1_weight'].reshape(16, 64, 128)[:, :8].reshape(128, 128))
            self._add_bgrad('nms_linear_out_1', r['linear_out_1_bias'].view(16, 64)[:, :8].reshape(128))
            self._add_wgrad('nms_pair_pos_fc1_1', r['pair_pos_fc1_1_weight']); self._add_bgrad('nms_pair_pos_fc1_1', r['pair_pos_fc1_1_bias'])
            d_x = (r['d_roi_feat'] + g.view(BC, F, 128).float()).view(B, C, F, 128)             # residual + module
        if fused and d_x.is_contiguous():
            d_rank = torch.zeros((F, 128), device=dev, dtype=torch.float32)                     # sum over (image, class): one column-sum launch (fp32 accumulation)
            _lib.call('relnet_colsum_add', d_x.data_ptr(), F * 128, BC, F * 128, ops._dt(d_x), d_rank.data_ptr(), s_)
            T._wg_call(self._wg('nms_rank'), d_rank.to(bt), self.rank_emb)
            T.colsum_add(d_rank, self._bg('nms_rank'))
        else:
            d_rank = d_x.sum((0, 1), dtype=torch.float32)                                       # [F,128] (fp32 accumulation whatever d_x's dtype)
            self._add_wgrad('nms_rank', T.wgrad(d_rank.to(bt), self.rank_emb)); self._add_bgrad('nms_rank', d_rank.sum(0))
        if fused and d_x.dtype == bt and d_x.is_contiguous() and C <= 128:
            d_emb = torch.empty((B * N, 128), device=dev, dtype=bt)          # every row written: gathered per roi over the classes that rank it (fp32 sums)
            _lib.call('relnet_lnms_take_bwd', d_x.data_ptr(), rank_idx.data_ptr(), d_emb.data_ptr(), B, N, C, F, s_)
        else:
            d_emb = torch.zeros((B * N, 128), device=dev, dtype=torch.float32)
            flat = (rank_idx.long() + (torch.arange(B, device=dev) * N).view(B, 1, 1)).view(-1)
            d_emb.index_add_(0, flat, d_x.reshape(-1, 128).float())                             # take() backward (a roi is ranked in up to 80 classes: fp32 sums)
            d_emb = d_emb.to(bt)
        d_feat, dw, db = T.linear_bwd(feat2, self.w('roi_feat_embedding'), d_emb, w_t=self.wt('roi_feat_embedding'), keep_splits=True, wgrad_to=self._wg('roi_feat_embedding'), bgrad_to=self._bg('roi_feat_embedding'))
        self._add_bgrad('roi_feat_embedding', db)
        # sort / slice backward -> cls_prob -> softmax backward (background column has no direct gradient)
        d_prob = ops.lnms_scatter_bwd(d_sorted.contiguous(), rank_idx, N)      # d_prob[b, rank_idx[b,c,f], c] += d_sorted[b,f,c]
        if fused and d_cls_out is not None and d_cls_out.is_contiguous() and d_cls_out.dtype == torch.float32 and d_cls_out.shape[2] == C1:
            _lib.call('relnet_lnms_softmax_bwd', prob.data_ptr(), d_prob.data_ptr(), d_cls_out.data_ptr(), d_cls_out.stride(1), d_cls_out.stride(0),
                      B, N, C, s_)
            return None, d_feat.view(B, N, -1), lo
        p_bg = 1.0 - prob.sum(2, keepdim=True)
        inner = (prob * d_prob).sum(2, keepdim=True)
        d_cls = torch.cat([-p_bg * inner, prob * (d_prob - inner)], 2)
        return d_cls, d_feat.view(B, N, -1), lo

    def _dgrad_w(self, name, cout):
        """[Cout, 9*Cin] packed forward weights -> [Cin, 9*Cout] tap-flipped data-gradient weights."""
        wt = self.wt(name)
        if wt is not None:
            return wt
        w = self.w(name)
        cin = w.shape[1] // 9
        return w.view(cout, 3, 3, cin).flip(1, 2).permute(3, 1, 2, 0).reshape(cin, 9 * cout).contiguous()

    def _rel_params(self, i):
        class P(object):
            pass
        m = P()
        m.wqk, m.bqk = self.w('qk_%d' % i), self.b('qk_%d' % i)
        m.wout, m.bout = self.w('linear_out_%d' % i), self.b('linear_out_%d' % i)
        m.wqk_t, m.wout_t = self.wt('qk_%d' % i), self.wt('linear_out_%d' % i)
        m.wp = self.W.view(self.W.master, 'pair_pos_fc1_%d' % i)
        m.bp = self.b('pair_pos_fc1_%d' % i)
        return m

    def _relation_bwd(self, i, mod, f, x_act, rois, d_x, N, key_count=None, cache=None):
        """x_act = relu(f + relation_i(f)); returns d f (bf16, residual path + module path) and accumulates the module's parameter
        gradients straight into the flat buffers (relation.GradSink: one pack kernel, one projection-backward GEMM with the residual
        gradient in its epilogue, one queued weight-gradient product for [Wq; Wk; Wout], column-sum kernels for the biases)."""
        g = T.relu_bwd(d_x.contiguous(), x_act)
        d = mod.wqk.shape[0] // 2
        oq, sq = self.W.slices['qk_%d' % i]
        oo, so = self.W.slices['linear_out_%d' % i]
        wcat_t = self._relayout.get('rel_cat_%d' % i)
        if wcat_t is not None and oo == oq + sq[0] * sq[1] and getattr(self.cfg, 'relation_sink', os.environ.get('RELNET_REL_SINK', '1') != '0'):
            g3 = self.W.grad[oq:oq + (sq[0] + so[0]) * sq[1]].view(sq[0] + so[0], sq[1])      # d[Wq; Wk; Wout], adjacent slices of the flat buffer
            sink = GradSink(wcat_t, g, lambda dy2d, x2d: T._wg_call((g3, None, self._wq), dy2d, x2d),
                            self._bg('qk_%d' % i), self._bg('linear_out_%d' % i),
                            self.W.view(self.W.grad, 'pair_pos_fc1_%d' % i), self._bg('pair_pos_fc1_%d' % i), self._scratch, defer=self._defer)
            r = attention_module_backward(f, rois, None, g, nongt_dim=N, index=i, dtype=torch.bfloat16, packed=mod, key_count=key_count,
                                          cache=cache, sink=sink)
            return r['d_roi_feat']
        r = attention_module_backward(f, rois, None, g, nongt_dim=N, index=i, dtype=torch.bfloat16, packed=mod, key_count=key_count,
                                      cache=cache)
        self._add_wgrad('qk_%d' % i, torch.cat([r['query_%d_weight' % i], r['key_%d_weight' % i]], 0))
        self._add_bgrad('qk_%d' % i, torch.cat([r['query_%d_bias' % i], r['key_%d_bias' % i]], 0))
        self._add_wgrad('linear_out_%d' % i, r['linear_out_%d_weight' % i].reshape(d, -1))
        self._add_bgrad('linear_out_%d' % i, r['linear_out_%d_bias' % i])
        self._add_wgrad('pair_pos_fc1_%d' % i, r['pair_pos_fc1_%d_weight' % i])
        self._add_bgrad('pair_pos_fc1_%d' % i, r['pair_pos_fc1_%d_bias' % i])
        return (r['d_roi_feat'] + g.float()).to(torch.bfloat16)          # residual path + module path

    # ---- checkpoint view -----------------------------------------------------------------------------------
    def export_params(self):
        """The trainable parameters back under the reference's names and layouts (fp32, on the host): convolution
        weights [O,I,kh,kw] with the folded BatchNorm scale divided out again, fc_new_1 / roi_pool_fc1 columns back in
        (c, ph, pw) order, the fused RPN / query|key / cls|bbox matrices split.  Frozen tensors are not returned."""
        out = {}
        wv = lambda n: self.W.view(self.W.master, n).detach().cpu()
        bv = lambda n: self.Bv.view(self.Bv.master, n).detach().cpu()

        def unpack(w2d, k):
            o = w2d.shape[0]
            return w2d.view(o, k, k, -1).permute(0, 3, 1, 2).contiguous()

        for name in self.W.slices:
            if name in self.bn_scale:                                   # res3..res5 convolutions
                out[name + '_weight'] = unpack(wv(name) / self.bn_scale[name].cpu().view(-1, 1), self.ksize[name])
            elif name.startswith('fpn_') or name in ('rpn_conv_3x3', 'conv_new_1') or name.endswith('_offset'):
                out[name + '_weight'] = unpack(wv(name), self.ksize.get(name, 1)); out[name + '_bias'] = bv(name)
        if not self.fpn:
            w, b = wv('rpn_out'), bv('rpn_out')
            out['rpn_cls_score_weight'], out['rpn_bbox_pred_weight'] = unpack(w[:self.na2], 1), unpack(w[self.na2:], 1)
            out['rpn_cls_score_bias'], out['rpn_bbox_pred_bias'] = b[:self.na2].clone(), b[self.na2:].clone()
        inv = torch.empty_like(self.fc1_perm); inv[self.fc1_perm] = torch.arange(len(self.fc1_perm))
        n1, n2 = self.fc_names
        out[n1 + '_weight'], out[n1 + '_bias'] = wv('fc_new_1')[:, inv].contiguous(), bv('fc_new_1')
        out[n2 + '_weight'], out[n2 + '_bias'] = wv('fc_new_2').clone(), bv('fc_new_2')
        nc = self.num_classes
        out['cls_score_weight'], out['bbox_pred_weight'] = wv('cls_bbox')[:nc].clone(), wv('cls_bbox')[nc:].clone()
        out['cls_score_bias'], out['bbox_pred_bias'] = bv('cls_bbox')[:nc].clone(), bv('cls_bbox')[nc:].clone()
        mods = ([('', i) for i in (1, 2)] if self.relation else []) + ([('nms_', 1)] if self.cfg.learn_nms else [])
        for pre, i in mods:
            qk, bq = wv('%sqk_%d' % (pre, i)), bv('%sqk_%d' % (pre, i))
            h = qk.shape[0] // 2
            out['%squery_%d_weight' % (pre, i)], out['%skey_%d_weight' % (pre, i)] = qk[:h].clone(), qk[h:].clone()
            out['%squery_%d_bias' % (pre, i)], out['%skey_%d_bias' % (pre, i)] = bq[:h].clone(), bq[h:].clone()
            wo = wv('%slinear_out_%d' % (pre, i))
            out['%slinear_out_%d_weight' % (pre, i)] = wo.reshape(wo.shape[0], wo.shape[1], 1, 1).clone()
            out['%slinear_out_%d_bias' % (pre, i)] = bv('%slinear_out_%d' % (pre, i))
            out['%spair_pos_fc1_%d_weight' % (pre, i)] = wv('%spair_pos_fc1_%d' % (pre, i)).clone()
            out['%spair_pos_fc1_%d_bias' % (pre, i)] = bv('%spair_pos_fc1_%d' % (pre, i))
        if self.cfg.learn_nms:
            for n in ('nms_rank', 'roi_feat_embedding', 'nms_logit'):
                out[n + '_weight'], out[n + '_bias'] = wv(n).clone(), bv(n)
        if self.cfg.dcn:
            out['offset_weight'], out['offset_bias'] = wv('offset')[:, inv].contiguous(), bv('offset')
        return out

    def checkpoint_params(self):
        """(arg_params, aux_params) as MXNet's Module hands them to the epoch-end callback (core/module.py fit ->
        callback.do_checkpoint): every trainable tensor (export_params) plus the frozen ones under `arg`, the BatchNorm
        running statistics under `aux`.  `checkpoint.do_checkpoint(prefix, means, stds)` turns them into the reference's
        `.params` file incl. the de-normalised `bbox_pred_*_test` pair (core/callback.py:54-61)."""
        arg = self.export_params()
        aux = {}
        for k, v in self._fixed_src.items():
            (aux if 'moving_' in k else arg).setdefault(k, v.clone())
        return arg, aux

    def save_checkpoint(self, prefix, epoch):
        """`<prefix>-<epoch+1:04d>.params` exactly as train_end2end.py:149-152 + core/callback.py:54-61 write it
        (means / stds tiled over the regression classes: 2 when class agnostic)."""
        from . import checkpoint as ck
        c = self.cfg
        reps = 2 if getattr(c, 'class_agnostic', True) else c.num_classes
        means, stds = list(c.bbox_means) * reps, list(c.bbox_stds) * reps
        arg, aux = self.checkpoint_params()
        return ck.do_checkpoint(prefix, means, stds)(epoch, None, arg, aux)

    # ---- optimizer ----------------------------------------------------------------------------------------
    def _grad_buckets(self):
        """Weight-gradient buckets in buffer (= forward) order: res3 | res4 (first half) | res4 (second half) | res5 | everything after
        the trunk (RPN, conv_new_1 / FPN neck, 2FC + relation + learn-NMS heads).  The backward pass completes them last to first.
        res4's 104 MB are two buckets (cut at unit b11): its second half travels while the first half is still being differentiated,
        and what is left exposed after the last backward kernel is res3 (4.9 MB) plus the tail of a 47 MB message instead of 104 MB."""
        if getattr(self, '_buckets', None) is None and self.lnms_only:
            # (learn-NMS-only step: one bucket -- only the head's tail of the buffer ever holds a non-zero gradient)
            self._bucket_cuts, self._bucket_names, self._unit_bucket = [0, self.W.size], ('heads',), {}
            self._buckets = D.BucketedAllReduce(self.W.grad, self._bucket_cuts)
        if getattr(self, '_buckets', None) is None:
            trunk = [n for n in self.W.slices if n in self.bn_scale]            # BN-folded res3..res5 convolutions
            first = lambda pre: min(self.W.slices[n][0] for n in trunk if n.startswith(pre))
            trunk_end = max(self.W.slices[n][0] + (int(np.prod(self.W.slices[n][1])) + 63) // 64 * 64 for n in trunk)
            res4 = [u[1] for u in self.units if u[0] == 4]
            half = res4[len(res4) // 2]                                        # '4b11' of 4a, 4b1 .. 4b22
            cuts = [0, first('res4'), first('res' + half + '_'), first('res5'), trunk_end, self.W.size]
            self._bucket_names = ('res3', 'res4_lo', 'res4_hi', 'res5', 'heads')
            if getattr(self.cfg, 'split_res4_bucket', True) is False:
                cuts = [0, first('res4'), first('res5'), trunk_end, self.W.size]
                self._bucket_names = ('res3', 'res4_lo', 'res5', 'heads')
                half = None
            assert cuts == sorted(set(cuts)), cuts
            self._bucket_cuts = cuts
            self._unit_bucket = {}
            hi = False
            for u in self.units:
                if u[0] < 3:
                    continue
                if u[0] == 4:
                    hi = hi or (u[1] == half)
                    self._unit_bucket[u[1]] = 'res4_hi' if hi else 'res4_lo'
                else:
                    self._unit_bucket[u[1]] = 'res%d' % u[0]
            self._buckets = D.BucketedAllReduce(self.W.grad, cuts)
        return self._buckets

    def _bucket_ready(self, *names):
        """Called by the backward pass when the last gradient of a bucket (or of several at once) has been queued (no-op on one rank)."""
        self._flush_wgrads()         # the bucket is complete only once its queued weight gradients have been launched (and joined)
        bk = self._grad_buckets()
        idxs = [self._bucket_names.index(n) for n in names]
        cut = getattr(self, '_capture_cut', None)
        if cut is not None:          # CapturedStep: close the current hipGraph here; the collective(s) go between two graphs
            cut(idxs[0] if len(idxs) == 1 else tuple(idxs))
            return
        for idx in idxs:
            bk.ready(idx)

    def all_reduce(self, wait=True):
        """Summed all-reduce of the gradients over RCCL (MXNet kvstore 'device' + rescale_grad 1.0 semantics): the weight
        buckets -- those the backward pass has not already launched -- and the 0.1 MB bias buffer.  wait=False only LAUNCHES what is
        missing: `update()` then waits bucket by bucket, in the order they were launched, and runs SGD on a bucket as soon as ITS
        sum has landed -- the optimizer of the heads / res5 buckets overlaps the all-reduce of res4 / res3 still in flight."""
        bk = self._grad_buckets()
        if not wait and bk.active():
            order = bk.launch_rest()
            self._bias_work = D.all_reduce_async(self.Bv.grad)
            return order
        order = bk.finish()
        all_reduce_sum(self.Bv.grad)
        self._bias_work = None
        return order

    def _trainable_ranges(self, buf):
        """Contiguous element ranges of a flat buffer that hold trainable slices (everything, unless network.FIXED_PARAMS fixes some)."""
        if not self.frozen_names:
            return [(0, buf.size)]
        key = id(buf)
        cache = self.__dict__.setdefault('_range_cache', {})
        if key not in cache:
            rs = []
            for n, (off, shape) in sorted(buf.slices.items(), key=lambda kv: kv[1][0]):
                if n in self.frozen_names:
                    continue
                end = off + (int(np.prod(shape)) + 63) // 64 * 64
                if rs and rs[-1][1] == off:
                    rs[-1] = (rs[-1][0], end)
                else:
                    rs.append((off, end))
            cache[key] = rs
        return cache[key]

    def _sgd(self, buf, lo, hi, lr, wd, bf16=True):
        for a, b_ in self._trainable_ranges(buf):          # (fixed parameters: neither gradient step nor weight decay nor momentum)
            l, h = max(lo, a), min(hi, b_)
            if h > l:
                T.sgd_update(buf.master[l:h], buf.mom[l:h], buf.grad[l:h], lr, self.cfg.momentum, wd, 1.0,
                             w_bf16=buf.work[l:h] if bf16 else None)

    def update(self, lr=None):
        """mx.optimizer.SGD over the flat buffers.  One rank (or everything already waited for): one launch for the weights, one for
        the biases.  With bucket all-reduces in flight (all_reduce(wait=False)): per bucket, in launch order, wait + SGD on its slice."""
        c = self.cfg
        lr = c.lr if lr is None else lr
        W, Bv = self.W, self.Bv
        wo, bo, mult = self.lr_mult_tail if self.lr_mult_tail is not None else (W.size, Bv.size, 1.0)
        bk = self._grad_buckets()
        pending = []
        if bk.active():
            # every bucket must have been exchanged before its slice is updated: a caller that skipped all_reduce() (or whose backward
            # pass announced only some buckets) gets the missing collectives launched here instead of a silently local update
            if not bk.exchanged:                    # (exchanged: all_reduce(wait=True) already summed this pass's gradients)
                if len(bk.done) < bk.n:
                    bk.launch_rest()
                if getattr(self, '_bias_work', None) is None:
                    self._bias_work = D.all_reduce_async(self.Bv.grad)
                assert len(bk.done) == bk.n
            pending = bk.pending_order()
        self.update_order = []                      # (bucket index, its collective had completed when SGD was queued) -- test hook
        ranges = [(self._bucket_cuts[i], self._bucket_cuts[i + 1], i) for i in pending] if pending else [(0, W.size, None)]
        timing = getattr(self, 'comm_timing', None)   # list (bench.py --gpus N): per bucket, (name, event before the wait, event after it)
        for lo, hi, i in ranges:
            if i is not None:
                self.update_order.append((i, bk.is_completed(i)))
                if timing is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                bk.wait(i)
                if timing is not None:
                    e1.record()
                    timing.append((self._bucket_names[i], e0, e1))
            # parameters with lr_mult != 1 (DCN `offset` FC) sit at the tail of both flat buffers
            self._sgd(W, lo, min(hi, wo), lr, c.wd)
            self._sgd(W, max(lo, wo), hi, lr * mult, c.wd)
        if pending:
            bk.clear()
        if getattr(self, '_bias_work', None) is not None:
            if timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            D.wait_work(self._bias_work, self.Bv.grad)
            if timing is not None:
                e1.record()
                timing.append(('biases', e0, e1))
            self._bias_work = None
        self._sgd(Bv, 0, bo, lr, 0.0, bf16=False)
        self._sgd(Bv, bo, Bv.size, lr * mult, 0.0, bf16=False)
        self.step_count += 1

    def comm_report(self, steps):
        """Exposed communication of the steps since `comm_timing = []` was set: per gradient bucket, the time the COMPUTE stream sat
        in that bucket's wait (event before / after the wait, both on the compute stream: zero when the collective had finished under
        the backward pass), averaged over `steps`; plus their sum per step.  Call after a device synchronisation."""
        per = {}
        for name, e0, e1 in (getattr(self, 'comm_timing', None) or []):
            per[name] = per.get(name, 0.0) + e0.elapsed_time(e1)
        per = {k: v / max(steps, 1) for k, v in per.items()}
        mb = {n: 4e-6 * (self._bucket_cuts[i + 1] - self._bucket_cuts[i]) for i, n in enumerate(self._bucket_names)}
        mb['biases'] = 4e-6 * self.Bv.size
        return {'exposed_wait_ms_per_step': {k: round(v, 4) for k, v in per.items()},
                'exposed_comm_ms_per_step': round(sum(per.values()), 4),
                'bucket_megabytes': {k: round(v, 1) for k, v in mb.items()},
                'how': 'HIP events on the compute stream around each bucket\'s wait in Trainer.update(): the time the optimizer was held up by a '
                       'collective still in flight (0 = fully hidden under the backward pass)'}

    def step(self, *batch, **kw):
        out = self.forward_backward(*batch, **kw)
        self.all_reduce(wait=False)
        self.update()
        return out


class CapturedStep(object):
    """forward + backward of a trainer captured as a CHAIN of hipGraphs, cut where a gradient bucket completes
    (heads | res5 | res4 | res3, the order the backward pass finishes them): on replay every bucket's summed all-reduce is
    issued from the host BETWEEN two graph launches, on the communication stream, and runs under the next segment's kernels --
    the overlap of dist.BucketedAllReduce without the ~10 ms of eager Python launches per step, and without capturing the
    collective itself (a captured RCCL call would pin the communicator's buffers into the graph).
    The reference overlaps nothing: kvstore push / pull runs after the whole backward pass (core/module.py:569-591).

        step = CapturedStep(trainer, batch)          # one eager warm-up must have run before (kernel attributes, caches)
        out = step.replay(); trainer.all_reduce(); trainer.update()
    """

    def __init__(self, trainer, batch, kwargs=None, segments=None):
        """segments: True = cut at every gradient bucket; None (default) = cut only when there is somebody to exchange the buckets with
        (a process group of more than one rank) -- on one rank the whole pass is ONE hipGraph (round 6: five graph launches and their seams
        less per step)."""
        self.tr = trainer
        do_cut = trainer._grad_buckets().active() if segments is None else bool(segments)
        if trainer.step_count == 0:              # kernel attributes (hipFuncSetAttribute), allocator pools and caches must exist before a capture
            with torch.no_grad():
                trainer.forward_backward(*batch, **(kwargs or {}))
                trainer.all_reduce()             # completes the bucket exchanges that pass announced (every rank does the same)
            torch.cuda.synchronize()
        self.segments = []                       # [(hipGraph, bucket index or None)]
        pool = torch.cuda.graph_pool_handle()    # one private pool: tensors made in one segment stay valid in the next ones
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        cur = [None]

        def begin():
            cur[0] = torch.cuda.CUDAGraph()
            # thread_local: with a process group up, RCCL's watchdog thread polls events while this thread captures; in the default 'global'
            # capture mode a CUDA call from ANOTHER thread invalidates the capture (never an issue on one rank / over gloo)
            cur[0].capture_begin(pool=pool, capture_error_mode='thread_local')

        def cut(idx):
            if not do_cut:               # one rank: nothing is exchanged between the segments, keep capturing into the same graph
                return
            cur[0].capture_end()
            self.segments.append((cur[0], idx))
            if idx == 0 or (isinstance(idx, tuple) and 0 in idx):   # bucket 0 (res3) is announced by the last kernel of the backward pass: nothing follows, so no
                cur[0] = None            # further capture is opened (an empty hipGraph used to be captured here and warned about on stderr)
            else:
                begin()

        torch.cuda.synchronize()
        with torch.cuda.stream(side), torch.no_grad():
            begin()
            trainer._capture_cut = cut
            import warnings
            try:
                self.out = trainer.forward_backward(*batch, **(kwargs or {}))
            except BaseException:
                trainer._capture_cut = None
                try:                              # leave capture mode, but let the ORIGINAL error propagate
                    if cur[0] is not None:
                        cur[0].capture_end()
                except Exception:
                    pass
                raise
            trainer._capture_cut = None
            if cur[0] is not None:       # (a trainer whose backward did not end with bucket 0: keep its tail)
                with warnings.catch_warnings(record=True) as caught:
                    warnings.simplefilter('always')
                    cur[0].capture_end()
                if not any('Graph is empty' in str(w.message) for w in caught):
                    self.segments.append((cur[0], None))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def replay(self):
        """Launch the segments in order; after each one, announce its bucket (asynchronous all-reduce on >1 rank)."""
        bk = self.tr._grad_buckets()
        bk.reset()
        for graph, idx in self.segments:
            graph.replay()
            for i in (() if idx is None else idx if isinstance(idx, tuple) else (idx,)):
                bk.ready(i)
        return self.out


class FPNTrainer(Trainer):
    """Training step of the FPN relation graphs (BASELINE configs[4]; symbols/resnet_v1_101_rcnn_fpn_attention_1024_
    pairwise_position_multi_head_16_learn_nms.py, get_symbol_rcnn train branch :1040-1200): proposals are an input
    (HAS_RPN: false, TOP_ROIS 1000), their labels / targets come from proposal_target-style sampling with the gt rows
    appended, rois are dispatched to the pyramid levels (core/rcnn.py:53-74) and pooled from fpn_ft4..ft32.

    Row order: the reference keeps a level-major order over ALL rows and selects the relation keys with `nongt_index`
    (:881-885, 931); here the non-gt rows are dispatched first and the gt rows after them, so the keys are the first N
    rows -- the same set of keys, and every per-roi output is mapped back through `perm`."""

    scales = (1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0)

    def __init__(self, params, cfg=None, device='cuda'):
        cfg = cfg or TrainConfig()
        cfg.fpn = True
        Trainer.__init__(self, params, cfg, device, im_hw=None)

    def _forward_backward_impl(self, data, im_info, gt_boxes, proposals, num_gt=None, num_proposals=None):
        """data [B,3,H,W] (H, W multiples of 32), proposals [B,N,4] fp32, gt_boxes [B,G,5]; num_proposals [B] int32
        (optional): real rows of `proposals` per image -- the reference hands every image its own roi count
        (core/rcnn.py:128-146, TOP_ROIS truncation only); a batched step pads to a common N and the padded rows are
        taken out on the device: zero box, label -1, zero weights (relnet_proposal_target_ex), moved behind the real rows by
        the level dispatch, no relation keys (key_count), never ranked by the learn-NMS branch."""
        c = self.cfg
        B, N = proposals.shape[:2]
        bt = torch.bfloat16
        if data.shape[2] % 32 or data.shape[3] % 32:
            raise ValueError("FPN images must be padded to IMAGE_STRIDE 32, got %s" % (tuple(data.shape),))
        self._grad_buckets().reset()
        self.W.grad.zero_(); self.Bv.grad.zero_()
        self._relayout.run()            # W^T / tap-flipped copies of the current weights for every data-gradient product
        self._fragpack.run()            # ... and the fragment-order copies the chain kernels of the forward read
        out = {}
        conv5, conv4, saved, ends = self._trunk_forward(data)
        # ---- neck: 1x1 laterals (+bias), nearest 2x upsampling + sum, 3x3 output convs
        src = {32: ends[5], 16: ends[4], 8: ends[3], 4: ends[2]}
        if self.trunk_fp32:             # float32 trunk (parity tests): bf16 copies for the bf16 neck / heads
            src = {k: v.to(bt) for k, v in src.items()}
        tops = {32: self._conv(src[32], 'fpn_ft32_1x1', bias=self.b('fpn_ft32_1x1'))}
        for lvl in (16, 8, 4):
            tops[lvl] = ops.upsample2x_add_(self._conv(src[lvl], 'fpn_ft%d_1x1' % lvl, bias=self.b('fpn_ft%d_1x1' % lvl)), tops[lvl * 2])
        feats = {lvl: self._conv(tops[lvl], 'fpn_ft%d_3x3' % lvl, pad=1, bias=self.b('fpn_ft%d_3x3' % lvl)) for lvl in (4, 8, 16, 32)}
        # ---- rois: labels / targets, then level dispatch (non-gt rows first, gt rows after)
        rois5 = torch.cat([torch.zeros((B, N, 1), device=proposals.device), proposals], 2)
        rois5[:, :, 0] = torch.arange(B, device=proposals.device, dtype=torch.float32).view(B, 1)
        rois_t, label, bbox_target, bbox_weight = ops.proposal_target(rois5.contiguous(), gt_boxes, num_gt, num_rois=num_proposals)
        R = rois_t.shape[1]
        key_count = None
        if num_proposals is not None:
            ra, la, pa, _, key_count = ops.fpn_roi_dispatch(rois_t[:, :N].contiguous(), n_valid=num_proposals)
        else:
            ra, la, pa, _ = ops.fpn_roi_dispatch(rois_t[:, :N].contiguous())
        rb, lb, pb, _ = ops.fpn_roi_dispatch(rois_t[:, N:].contiguous())
        rois_s = torch.cat([ra, rb], 1).contiguous()
        level = torch.cat([la, lb], 1).contiguous()
        perm = torch.cat([pa, pb + N], 1).long()
        gat = lambda t: torch.gather(t, 1, perm.view(B, R, *([1] * (t.dim() - 2))).expand(B, R, *t.shape[2:]))
        label, bbox_target, bbox_weight = gat(label), gat(bbox_target).contiguous(), gat(bbox_weight).contiguous()
        nchw = lambda t: t.permute(0, 3, 1, 2)
        lv = [nchw(feats[4]), nchw(feats[8]), nchw(feats[16]), nchw(feats[32])]
        pooled, argmax = ops.roi_pool_fpn(lv, self.scales, rois_s.view(B * R, 5), level.view(-1), (7, 7), channels_last_out=True,
                                          want_argmax=True)
        pooled2 = pooled.permute(0, 2, 3, 1).reshape(B * R, -1)
        d_pool, hs = self._head_forward_backward(pooled2, rois_s, N, label.contiguous(), bbox_target, bbox_weight, im_info,
                                                 gt_boxes, num_gt, out, key_count=key_count)
        x2, f1, cls_score, bbox_pred, labels_ohem, weights_ohem = hs
        # ---- pooling backward into the four pyramid maps
        g_lv = ops.roi_pool_fpn_bwd(d_pool.view(B * R, 7, 7, -1).permute(0, 3, 1, 2), argmax, rois_s.view(B * R, 5), level.view(-1),
                                    [tuple(t.shape) for t in lv], channels_last=True)
        d_feats = {lvl: g.permute(0, 2, 3, 1).to(bt) for lvl, g in zip((4, 8, 16, 32), g_lv)}
        # ---- neck backward (top-down pathway reversed: finest level first, gradients flow up to the coarser tops)
        d_tops, inject, d_c5 = {}, {}, None
        for lvl in (4, 8, 16, 32):
            n3, n1 = 'fpn_ft%d_3x3' % lvl, 'fpn_ft%d_1x1' % lvl
            d_top, dw = T.conv3x3_bwd(tops[lvl], self._dgrad_w(n3, 256), d_feats[lvl], dil=1, keep_splits=True, wgrad_to=self._wg(n3))
            T.colsum_add(d_feats[lvl], self._bg(n3))             # (bias gradients: the bucket's grouped column-sum launch)
            if lvl in d_tops:                                    # + the gradient that came down from the finer level (fp32 block sums)
                d_top = torch.add(d_top, d_tops[lvl]).to(bt)
            if lvl < 32:       # tops[lvl] = lateral + up2x(tops[2 lvl]): adjoint of nearest upsampling = 2x2 block sums (fp32 accumulation, read as bf16)
                Bh, Hh, Wh, Ch = d_top.shape
                d_tops[lvl * 2] = d_top.view(Bh, Hh // 2, 2, Wh // 2, 2, Ch).sum((2, 4), dtype=torch.float32)
            d_top = d_top.contiguous()
            d_src, dw = T.conv1x1_bwd(src[lvl], self.w(n1), d_top, need_dx=(lvl != 4), w_t=self.wt(n1), keep_splits=True, wgrad_to=self._wg(n1))   # res2c is frozen
            T.colsum_add(d_top, self._bg(n1))
            if lvl == 32:
                d_c5 = d_src
            elif lvl == 16:
                inject['4b22'] = d_src
            elif lvl == 8:
                inject['3b3'] = d_src
        if self.trunk_fp32:
            d_c5, inject = d_c5.float(), {k: v.float() for k, v in inject.items()}
        self._trunk_backward(saved, d_c5, inject)
        out.update(rois=rois_s, perm=perm, roi_level=level, label=labels_ohem, bbox_target=bbox_target, bbox_weight=weights_ohem,
                   bbox_pred=bbox_pred, cls_score=cls_score)
        out['intermediates'] = dict(feats=feats, pooled=pooled2)        # forward values for the stage-wise parity test
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
