"""End-to-end test-time detector of the relation-network hot path (one process per GPU,
B images per launch): backbone -> RPN -> proposal -> ROIPooling -> 2FC (+2 relation modules)
-> cls/bbox -> decode -> per-class soft-NMS / NMS -> max_per_image.

Graph: relation_rcnn/symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py
:176-322 (test branch; `relation=False` gives resnet_v1_101_rcnn.py:96-174, the plain 2FC
head of BASELINE config 1) + core/tester.py:148-156,244-277.  Hyper-parameters default to
experiments/relation_rcnn/cfgs/resnet_v1_101_coco_trainvalminus_rcnn_end2end_relation_8epoch.yaml.
"""
import os

import torch

from . import ops
from .backbone import Backbone
from .relation import RelationHead
from .learn_nms import LearnNMS
from .operator_py.proposal import generate_anchors, propose_batch


class Config(object):
    feat_stride = 16
    anchor_scales = (4, 8, 16, 32)
    anchor_ratios = (0.5, 1, 2)
    rpn_pre_nms_top_n = 6000
    rpn_post_nms_top_n = 300
    rpn_nms_thresh = 0.7
    rpn_min_size = 0
    num_classes = 81
    nms = 0.6                 # TEST.NMS (sigma for soft-NMS, IoU threshold for NMS)
    softnms = True            # TEST.SOFTNMS
    score_thresh = 1e-3       # tester.py:175
    max_per_image = 100
    learn_nms = False         # TEST.LEARN_NMS: learned duplicate removal instead of (soft-)NMS
    first_n = 100             # TEST.FIRST_N
    learn_nms_class_thresh = 0.01   # TEST.LEARN_NMS_CLASS_SCORE_TH
    nms_target_thresh = (0.5, 0.6, 0.7, 0.8, 0.9)   # network.NMS_TARGET_THRESH
    merge_method = -1         # TEST.MERGE_METHOD (mean over thresholds)
    dcn = False               # deformable res5 + DeformablePSROIPooling (symbols/..._dcn_...py)
    dcn_sample_per_part = 4
    dcn_trans_std = 0.1
    roi_align = False         # True: ROIAlign (ops.roi_align, sampling_ratio 2) feeds fc_new_1 instead of the graphs' ROIPooling -- the operator
    roi_align_sampling = 2    # north_star names; no reference graph uses it (SYM_REL:252-253 is ROIPooling), so it is off by default

    @classmethod
    def from_experiment(cls, name, train=False):
        """The hyper-parameters of one shipped experiment file (config.EXPERIMENTS key = the YAML stem without
        `resnet_v1_101_coco_trainvalminus_`), e.g. 'rcnn_fpn_relation_learn_nms_8epoch' -> first_n 150, learn_nms_class_thresh 0.05,
        and for a TrainConfig batch_rois_ohem 512, lr 0.00125 (..._rcnn_fpn_relation_learn_nms_8epoch.yaml:62,92,141,166-167).
        train=True reads TRAIN.FIRST_N / TRAIN.LEARN_NMS where the test graph reads TEST.*."""
        from .config import experiment
        e = experiment(name)
        c = cls()
        n, t, te = e.network, e.TRAIN, e.TEST
        c.experiment = name
        c.symbol = e.symbol
        c.feat_stride = n.RPN_FEAT_STRIDE
        c.anchor_scales, c.anchor_ratios = tuple(n.ANCHOR_SCALES), tuple(n.ANCHOR_RATIOS)
        c.num_classes = e.dataset.NUM_CLASSES
        c.nms_target_thresh = tuple(float(v) for v in str(n.NMS_TARGET_THRESH).split(','))
        src = t if train else te
        c.rpn_pre_nms_top_n, c.rpn_post_nms_top_n = src.RPN_PRE_NMS_TOP_N, src.RPN_POST_NMS_TOP_N
        c.rpn_nms_thresh, c.rpn_min_size = src.RPN_NMS_THRESH, src.RPN_MIN_SIZE
        c.nms, c.softnms, c.max_per_image = te.NMS, te.SOFTNMS, te.max_per_image
        c.learn_nms = bool(t.LEARN_NMS if train else te.LEARN_NMS)
        c.first_n = (t.FIRST_N if train else te.FIRST_N) or cls.first_n
        c.learn_nms_class_thresh = te.LEARN_NMS_CLASS_SCORE_TH
        c.merge_method = te.MERGE_METHOD
        c.dcn = '_dcn' in e.symbol
        # relation modules in the 2FC head: every `..._attention_...` symbol except the learn-NMS-only graph, whose name carries the attention
        # suffix for its learn-NMS head's module (resnet_v1_101_rcnn_learn_nms_1024_attention_...: plain fc_new_1 / fc_new_2, symbols/...:176-216)
        c.relation = ('_attention_' in e.symbol) and not e.symbol.startswith('resnet_v1_101_rcnn_learn_nms')
        c.fpn = '_fpn' in e.symbol
        c.top_rois = t.TOP_ROIS if train else te.TOP_ROIS          # proposals per image of the HAS_RPN: false (FPN) graphs
        c.scales = tuple(e.SCALES[0])
        if hasattr(cls, 'batch_rois_ohem'):                          # a TrainConfig
            c.batch_rois_ohem, c.enable_ohem = t.BATCH_ROIS_OHEM, bool(t.ENABLE_OHEM)
            c.joint_training = bool(t.JOINT_TRAINING) or not t.LEARN_NMS
            c.lr, c.momentum, c.wd = t.lr, t.momentum, t.wd
            c.rpn_batch_size = t.RPN_BATCH_SIZE
            c.nms_loss_scale, c.nms_pos_scale = t.nms_loss_scale, t.nms_pos_scale
            c.bbox_means, c.bbox_stds = tuple(t.BBOX_MEANS), tuple(t.BBOX_STDS)
            c.fixed_params = list(n.FIXED_PARAMS)
        return c


def fc1_channels_last_perm(c=256, ph=7, pw=7):
    """Column permutation of fc_new_1_weight for pooled features stored (ph, pw, c):
    new column (s*C + ch) <- reference column (ch*49 + s)."""
    s = torch.arange(ph * pw).view(-1, 1)
    ch = torch.arange(c).view(1, -1)
    return (ch * (ph * pw) + s).reshape(-1)


class Detector(object):
    def __init__(self, params, dtype=torch.bfloat16, device='cuda', cfg=None, relation=True,
                 im_hw=(600, 1000), stem='hip'):
        self.cfg = cfg or Config()
        self.dtype, self.device, self.relation, self.im_hw = dtype, device, relation, im_hw
        self.overlap_rpn = True
        # RPN head + proposal beside res5 on a side stream: under hipGraph capture the fork / join are graph edges and it pays from one
        # image (3.0 -> 2.75-2.80 ms, r03); launched eagerly at 1-2 images the extra events cost more than the overlap gives
        # (3.3 -> 4.5 ms, r02), so eager calls fork from 3 images.  None = that rule; RELNET_OVERLAP_MIN_IMAGES overrides it.
        env = os.environ.get('RELNET_OVERLAP_MIN_IMAGES')
        self.overlap_min_images = int(env) if env else None
        self.backbone = Backbone(params, dtype, device, stem=stem, dcn=self.cfg.dcn)
        if self.cfg.dcn:          # FC 12544 -> 2*7*7 offsets (SYM_DCN_RELNMS:1075), columns in (ph, pw, c) order
            self.w_offset = params['offset_weight'][:, fc1_channels_last_perm()].to(device, dtype).contiguous()
            self.b_offset = params['offset_bias'].to(device, torch.float32).contiguous()
        self.head = RelationHead(params, dtype, device, fc1_perm=fc1_channels_last_perm(),
                                 use_relation=relation)
        self.lnms = None
        if self.cfg.learn_nms:
            self.lnms = LearnNMS(params, self.cfg.num_classes - 1, self.cfg.first_n, len(self.cfg.nms_target_thresh),
                                 self.cfg.learn_nms_class_thresh, None, None, self.cfg.merge_method,
                                 self.cfg.score_thresh, self.cfg.max_per_image, dtype=dtype, device=device)
        self.anchors = torch.as_tensor(generate_anchors(self.cfg.feat_stride, self.cfg.anchor_ratios,
                                                        self.cfg.anchor_scales), dtype=torch.float64, device=device)

    @classmethod
    def from_checkpoint(cls, prefix, epoch, **kw):
        """Test-time construction from an MXNet `.params` file, as function/test_rcnn.py:57 does:
        `load_param(prefix, epoch, process=True)` -- the `*_test` tensors (bbox_pred de-normalised by
        BBOX_STDS / MEANS, core/callback.py:54-61) replace their training-time names."""
        from . import checkpoint as ck
        arg, aux = ck.load_param(prefix, epoch, process=True)
        return cls(ck.merge_params(arg, aux), **kw)

    def forward(self, data, im_info, post=True, keep_features=False):
        """data [B,3,H,W], im_info [B,3] fp32 (device).  No host synchronisation inside.
        keep_features: also return the backbone maps of THIS call under 'features' (conv4, conv5, conv_new_1_relu, RPN maps)
        and the head's intermediates (attention_1/2, fc_all_1/2_relu) under 'head' -- what oracle/parity.py checks."""
        c = self.cfg
        B = data.shape[0]
        propose = lambda cls, box: propose_batch(cls.float(), box.float(), im_info, self.anchors, c.feat_stride,
                                                 c.rpn_pre_nms_top_n, c.rpn_post_nms_top_n, c.rpn_nms_thresh, c.rpn_min_size,
                                                 im_hw=self.im_hw, softmax_pairs=True, want_num=True)
        min_images = self.overlap_min_images
        if min_images is None:
            min_images = 1 if torch.cuda.is_current_stream_capturing() else 3
        if self.overlap_rpn and self.backbone.impl == 'hip' and B >= min_images:   # RPN head + proposal on a side stream, beside res5
            # (under hipGraph replay the fork / join are graph edges: at one image per step 3.0 -> 2.75-2.80 ms, at two 3.63 -> 3.30 ms,
            #  r03 A/B with RELNET_OVERLAP_MIN_IMAGES; round 2's eager-mode measurement had said the opposite)
            f = self.backbone.forward(data, rpn_hook=propose)
            rois, roi_scores, num_kept = f['rpn_hook']
        else:
            f = self.backbone.forward(data)
            rois, roi_scores, num_kept = propose(f['rpn_cls_score'], f['rpn_bbox_pred'])
        N = rois.shape[1]
        if c.dcn:                 # SYM_DCN_RELNMS:1073-1080
            feat, r5, sc = f['conv_new_1_relu'], rois.view(B * N, 5), 1.0 / c.feat_stride
            t0 = ops.deformable_psroi_pool(feat, r5, None, sc, feat.shape[1], 1, 7, 7, c.dcn_sample_per_part, 0.0, True,
                                           channels_last_out=True)
            trans = ops.gemm_nt(t0.permute(0, 2, 3, 1).reshape(B * N, -1), self.w_offset, self.b_offset,
                                out_dtype=torch.float32).view(B * N, 2, 7, 7)
            pooled = ops.deformable_psroi_pool(feat, r5, trans, sc, feat.shape[1], 1, 7, 7, c.dcn_sample_per_part,
                                               c.dcn_trans_std, False, channels_last_out=True)
        elif c.roi_align:
            pooled = ops.roi_align(f['conv_new_1_relu'], rois.view(B * N, 5), (7, 7), 1.0 / c.feat_stride, c.roi_align_sampling,
                                   channels_last_out=True)
        else:
            pooled = ops.roi_pool(f['conv_new_1_relu'], rois.view(B * N, 5), (7, 7), 1.0 / c.feat_stride,
                                  channels_last_out=True)
        pooled = pooled.permute(0, 2, 3, 1).reshape(B, N, -1)              # (ph, pw, c) order, no copy
        if keep_features and self.relation:
            hd = self.head.forward(pooled, rois, return_intermediates=True)
            cls_score, bbox_pred, feat = hd['cls_score'], hd['bbox_pred'], hd['fc_all_2_relu']
        else:
            hd = None
            cls_score, bbox_pred, feat = self.head.forward(pooled, rois)
        out = dict(rois=rois, roi_scores=roi_scores, num_kept=num_kept, cls_score=cls_score, bbox_pred=bbox_pred, fc_all_2_relu=feat)
        if keep_features:
            out['features'], out['head'], out['pooled'] = f, hd, pooled
        if self.lnms is not None and post:                 # symbols/..._learn_nms.py:518-565 + tester.py:231-242
            out.update(self.lnms.forward(cls_score.contiguous(), bbox_pred.contiguous(), rois, im_info, feat))
            return out
        prob, boxes = ops.detect_head(cls_score.reshape(B * N, -1), bbox_pred.reshape(B * N, -1),
                                      rois.view(B * N, 5), im_info, N)
        out['cls_prob'], out['pred_boxes'] = prob.view(B, N, -1), boxes.view(B, N, 4)
        if post:
            # (top_k: a class list stops once its next pick cannot reach the image's max_per_image best scores -- the lists are
            #  prefixes of the full per-class lists that contain everything image_topk keeps: relnet_class_nms_topk)
            dets, counts = ops.class_nms(out['cls_prob'], out['pred_boxes'], c.score_thresh, c.nms, c.softnms,
                                         max_picks=c.max_per_image, top_k=c.max_per_image)
            det, det_count, thresh, total = ops.image_topk(dets, counts, c.max_per_image)
            out.update(class_dets=dets, class_counts=counts, detections=det, num_detections=det_count,
                       image_thresh=thresh)
        return out



class InFlight(object):
    """Throughput mode of the captured step: n steps -- each a forward of its OWN detector instance on its own resident inputs (own
    activations, scratch and split-K work area) -- are captured as n hipGraphs on n streams and replayed round-robin, so that
    consecutive batches overlap: the tail of batch k (ROI pooling, the 2FC head, per-class NMS: partial waves of workgroups) and the
    single-workgroup proposal kernels run beside the trunk of batch k + 1.  Measured on one MI355X (tools/pipeline_probe.py), images/s one at a
    time -> in flight:
      three steps with the RPN branch in line (`det.overlap_rpn = False`: one hardware queue per step):  1 image per step 440-466 -> 826-860,
          2 images 637 -> 1228, 8 images 1494-1559 -> 2238-2269, 54 images 2723-2756 -> 2875-2878, 108 images 2813 -> 2901;
      two steps that keep their RPN side stream (two queues each):  1 image 479 -> 735, 8 images 1569 -> 2145, 108 images 2810 -> 2866.
    The runtime has four hardware queues: four single-queue steps (1 image: 675) or three two-queue steps (608) are SLOWER than these.
    Every graph must be captured on ITS stream: two graphs captured on one stream share the runtime's queues and do not overlap (+0 - 1 %).
    The reference runs one batch at a time per device (core/tester.py:pred_eval); a replay gives exactly what one forward of that instance gives.

        dets = [Detector(params, ...) for _ in range(3)]
        for d in dets: d.overlap_rpn = False
        fl = InFlight([lambda i=i: dets[i].forward(data[i], im_info) for i in range(3)])
        i = fl.submit()                 # replays slot k % n on its stream (refresh that slot's input tensors before, on any stream ordered before it)
        out = fl.result(i)              # waits for THAT replay only; the tensors are overwritten by the slot's next replay
    """

    def __init__(self, steps, warmup=2, capture_error_mode='thread_local'):
        self.graphs, self.streams, self.outs, self.done = [], [], [], []
        with torch.no_grad():
            for step in steps:
                for _ in range(warmup):          # eager, on the caller's stream: kernel attributes, allocator pools, self-checks
                    step()
                torch.cuda.synchronize()
                s = torch.cuda.Stream()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s, capture_error_mode=capture_error_mode):
                    out = step()
                self.graphs.append(g); self.streams.append(s); self.outs.append(out); self.done.append(torch.cuda.Event())
        torch.cuda.synchronize()
        self.k = 0

    def __len__(self):
        return len(self.graphs)

    def submit(self):
        i = self.k % len(self.graphs)
        self.k += 1
        with torch.cuda.stream(self.streams[i]):
            self.graphs[i].replay()
            self.done[i].record()
        return i

    def result(self, i):
        self.done[i].synchronize()
        return self.outs[i]


class FPNDetector(object):
    """Test graph of the FPN relation configuration (symbols/resnet_v1_101_rcnn_fpn_attention_1024_pairwise_position_
    multi_head_16.py / ..._learn_nms.py, get_symbol_rcnn test branch :1085-1200): proposals are an INPUT
    (HAS_RPN: false, TOP_ROIS 1000), dispatched to pyramid levels (core/rcnn.py:53-74), pooled from
    fpn_ft4..fpn_ft32, then roi_pool_fc1/fc2 with two relation modules over all N proposals.

    Rows of every per-roi output are in the reference's level-major order; `perm` maps them back to the input
    order.  Images must be padded to a multiple of 32 (IMAGE_STRIDE) like the reference's loader does.

    Row count: the reference's loader appends one all-zero roi for every pyramid level that received none
    (core/rcnn.py:61-71) -- a real row of its graph (pooled, scored, and a key of both relation modules).  With
    `pad_empty_levels` (default) the device dispatch does the same into a fixed buffer of N + 4 rows per image;
    `num_rows` [B] says how many of them are real (N + the image's empty levels), the rest is padding that the relation
    kernels skip as keys and the post-processing never reports.  `num_proposals` [B] int32 handles images that bring fewer
    than N proposals (rows past it are padding too)."""

    scales = (1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0)

    def __init__(self, params, dtype=torch.bfloat16, device='cuda', cfg=None, relation=True, stem='hip'):
        self.cfg = cfg or Config()
        self.dtype, self.device = dtype, device
        self.backbone = Backbone(params, dtype, device, stem=stem, fpn=True)
        self.head = RelationHead(params, dtype, device, fc1_perm=fc1_channels_last_perm(), use_relation=relation,
                                 fc_names=('roi_pool_fc1', 'roi_pool_fc2'))
        self.lnms = None
        if self.cfg.learn_nms:
            self.lnms = LearnNMS(params, self.cfg.num_classes - 1, self.cfg.first_n, len(self.cfg.nms_target_thresh),
                                 self.cfg.learn_nms_class_thresh, None, None, self.cfg.merge_method,
                                 self.cfg.score_thresh, self.cfg.max_per_image, dtype=dtype, device=device)

    pad_empty_levels = True

    def forward(self, data, proposals, im_info, post=True, num_proposals=None):
        """data [B,3,H,W] (H, W multiples of 32); proposals [B,N,4] fp32 xyxy; im_info [B,3]; num_proposals [B] int32
        (optional, device): valid rows of `proposals` per image.  No host synchronisation inside."""
        c = self.cfg
        B, N = proposals.shape[:2]
        if data.shape[2] % 32 or data.shape[3] % 32:
            raise ValueError("FPN images must be padded to IMAGE_STRIDE 32, got %s" % (tuple(data.shape),))
        f = self.backbone.forward(data)
        n_rows = None
        if self.pad_empty_levels or num_proposals is not None:
            rois, level, perm, counts, n_rows = ops.fpn_roi_dispatch(proposals.contiguous(), n_valid=num_proposals,
                                                                     pad_empty=self.pad_empty_levels)
        else:
            rois, level, perm, counts = ops.fpn_roi_dispatch(proposals.contiguous())
        N = rois.shape[1]
        pooled = ops.roi_pool_fpn([f['fpn_ft4'], f['fpn_ft8'], f['fpn_ft16'], f['fpn_ft32']], self.scales,
                                  rois.view(B * N, 5), level.view(-1), (7, 7), channels_last_out=True)
        pooled = pooled.permute(0, 2, 3, 1).reshape(B, N, -1)
        cls_score, bbox_pred, feat = self.head.forward(pooled, rois, key_count=n_rows)
        out = dict(rois=rois, roi_level=level, perm=perm, level_counts=counts, num_rows=n_rows, cls_score=cls_score,
                   bbox_pred=bbox_pred, fc_all_2_relu=feat)
        if self.lnms is not None and post:
            out.update(self.lnms.forward(cls_score.contiguous(), bbox_pred.contiguous(), rois, im_info, feat, n_valid=n_rows))
            return out
        prob, boxes = ops.detect_head(cls_score.reshape(B * N, -1), bbox_pred.reshape(B * N, -1),
                                      rois.view(B * N, 5), im_info, N, n_valid=n_rows)
        out['cls_prob'], out['pred_boxes'] = prob.view(B, N, -1), boxes.view(B, N, 4)
        if post:
            dets, cnts = ops.class_nms(out['cls_prob'], out['pred_boxes'], c.score_thresh, c.nms, c.softnms,
                                       max_picks=c.max_per_image, top_k=c.max_per_image)
            det, det_count, thresh, total = ops.image_topk(dets, cnts, c.max_per_image)
            out.update(class_dets=dets, class_counts=cnts, detections=det, num_detections=det_count, image_thresh=thresh)
        return out
