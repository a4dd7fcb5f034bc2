"""End-to-end test-time detector of the relation-network hot path (one process per GPU,
B images per launch): backbone -> RPN -> proposal -> ROIPooling -> 2FC (+2 relation modules)
-> cls/bbox -> decode -> per-class soft-NMS / NMS -> max_per_image.

Graph: relation_rcnn/symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py
:176-322 (test branch; `relation=False` gives resnet_v1_101_rcnn.py:96-174, the plain 2FC
head of BASELINE config 1) + core/tester.py:148-156,244-277.  Hyper-parameters default to
experiments/relation_rcnn/cfgs/resnet_v1_101_coco_trainvalminus_rcnn_end2end_relation_8epoch.yaml.
"""
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
        self.backbone = Backbone(params, dtype, device, stem=stem)
        self.head = RelationHead(params, dtype, device, fc1_perm=fc1_channels_last_perm(),
                                 use_relation=relation)
        self.lnms = None
        if self.cfg.learn_nms:
            self.lnms = LearnNMS(params, self.cfg.num_classes - 1, self.cfg.first_n, len(self.cfg.nms_target_thresh),
                                 self.cfg.learn_nms_class_thresh, None, None, self.cfg.merge_method,
                                 self.cfg.score_thresh, self.cfg.max_per_image, dtype=dtype, device=device)
        self.anchors = torch.as_tensor(generate_anchors(self.cfg.feat_stride, self.cfg.anchor_ratios,
                                                        self.cfg.anchor_scales), dtype=torch.float64, device=device)

    def forward(self, data, im_info, post=True):
        """data [B,3,H,W], im_info [B,3] fp32 (device).  No host synchronisation inside."""
        c = self.cfg
        B = data.shape[0]
        f = self.backbone.forward(data)
        rois, roi_scores = propose_batch(f['rpn_cls_score'].float(), f['rpn_bbox_pred'].float(), im_info,
                                         self.anchors, c.feat_stride, c.rpn_pre_nms_top_n, c.rpn_post_nms_top_n,
                                         c.rpn_nms_thresh, c.rpn_min_size, im_hw=self.im_hw, softmax_pairs=True)
        N = rois.shape[1]
        pooled = ops.roi_pool(f['conv_new_1_relu'], rois.view(B * N, 5), (7, 7), 1.0 / c.feat_stride,
                              channels_last_out=True)
        pooled = pooled.permute(0, 2, 3, 1).reshape(B, N, -1)              # (ph, pw, c) order, no copy
        cls_score, bbox_pred, feat = self.head.forward(pooled, rois)
        out = dict(rois=rois, cls_score=cls_score, bbox_pred=bbox_pred, fc_all_2_relu=feat)
        if self.lnms is not None and post:                 # symbols/..._learn_nms.py:518-565 + tester.py:231-242
            out.update(self.lnms.forward(cls_score.contiguous(), bbox_pred.contiguous(), rois, im_info, feat))
            return out
        prob, boxes = ops.detect_head(cls_score.reshape(B * N, -1), bbox_pred.reshape(B * N, -1),
                                      rois.view(B * N, 5), im_info, N)
        out['cls_prob'], out['pred_boxes'] = prob.view(B, N, -1), boxes.view(B, N, 4)
        if post:
            dets, counts = ops.class_nms(out['cls_prob'], out['pred_boxes'], c.score_thresh, c.nms, c.softnms,
                                         max_picks=c.max_per_image)
            det, det_count, thresh, total = ops.image_topk(dets, counts, c.max_per_image)
            out.update(class_dets=dets, class_counts=counts, detections=det, num_detections=det_count,
                       image_thresh=thresh)
        return out
