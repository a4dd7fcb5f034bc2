"""Hyper-parameter tree of the reference (`relation_rcnn/config/config.py:18-198`): the same nested keys and defaults
and the same `update_config(yaml_file)` overlay (unknown top-level keys raise, `:177-198`), without easydict.
The graph files read it as `cfg.dataset.NUM_CLASSES`, `cfg.TRAIN.RPN_POST_NMS_TOP_N`, ... and it is pickled into
the `proposal_target` operator (`SYM_REL:220`), so attribute AND item access both work and it pickles.

`experiment(name)` returns a fresh tree with the values of the shipped COCO experiment files
(`experiments/relation_rcnn/cfgs/resnet_v1_101_coco_trainvalminus_rcnn_*.yaml`) that differ from the defaults --
those files are inputs of the path (they fix every static shape); their values are restated here so that the
GPU box, which has no reference checkout, can build the same graphs.
"""
import copy

import numpy as np


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def __reduce__(self):
        return (AttrDict, (dict(self),))


def _defaults():
    c = AttrDict()
    c.MXNET_VERSION = ''; c.output_path = ''; c.symbol = ''; c.gpus = ''
    c.CLASS_AGNOSTIC = True
    c.SCALES = [(600, 1000)]
    c.default = AttrDict(frequent=20, kvstore='device')
    n = c.network = AttrDict()
    n.pretrained = ''; n.pretrained_epoch = 0; n.PIXEL_MEANS = np.array([0, 0, 0]); n.IMAGE_STRIDE = 0
    n.RPN_FEAT_STRIDE = 16; n.RCNN_FEAT_STRIDE = 16
    n.FIXED_PARAMS = ['gamma', 'beta']; n.FIXED_PARAMS_SHARED = ['gamma', 'beta']
    n.ANCHOR_SCALES = (8, 16, 32); n.ANCHOR_RATIOS = (0.5, 1, 2); n.NUM_ANCHORS = 9
    n.ROIDispatch = False; n.USE_NONGT_INDEX = False; n.NMS_TARGET_THRESH = '0.5'
    c.dataset = AttrDict(dataset='PascalVOC', image_set='2007_trainval', test_image_set='2007_test', root_path='./data',
                         dataset_path='./data/VOCdevkit', NUM_CLASSES=21)
    t = c.TRAIN = AttrDict()
    t.lr = 0; t.lr_step = ''; t.lr_factor = 0.1; t.warmup = False; t.warmup_lr = 0; t.warmup_step = 0
    t.momentum = 0.9; t.wd = 0.0005; t.begin_epoch = 0; t.end_epoch = 0; t.model_prefix = ''
    t.rpn_loss_scale = 3.0; t.nms_loss_scale = 1.0; t.nms_pos_scale = 4.0
    t.ALTERNATE = AttrDict(RPN_BATCH_IMAGES=0)
    t.FC_DROPOUT_RATIO = 0; t.ATTENTION_DROPOUT_RATIO = 0; t.ATTENTION_SCALE_METHOD = 0
    t.RESUME = False; t.FLIP = True; t.SHUFFLE = True; t.ENABLE_OHEM = False; t.BATCH_IMAGES = 2; t.END2END = False
    t.ASPECT_GROUPING = True; t.TOP_ROIS = -1; t.BATCH_ROIS = 128; t.BATCH_ROIS_OHEM = 128
    t.FG_FRACTION = 0.25; t.FG_THRESH = 0.5; t.BG_THRESH_HI = 0.5; t.BG_THRESH_LO = 0.0
    t.BBOX_REGRESSION_THRESH = 0.5; t.BBOX_WEIGHTS = np.array([1.0, 1.0, 1.0, 1.0])
    t.RPN_BATCH_SIZE = 256; t.RPN_FG_FRACTION = 0.5; t.RPN_POSITIVE_OVERLAP = 0.7; t.RPN_NEGATIVE_OVERLAP = 0.3
    t.RPN_CLOBBER_POSITIVES = False; t.RPN_BBOX_WEIGHTS = (1.0, 1.0, 1.0, 1.0); t.RPN_POSITIVE_WEIGHT = -1.0
    t.CXX_PROPOSAL = True; t.RPN_NMS_THRESH = 0.7; t.RPN_PRE_NMS_TOP_N = 12000; t.RPN_POST_NMS_TOP_N = 2000
    t.RPN_MIN_SIZE = 16; t.BBOX_NORMALIZATION_PRECOMPUTED = False
    t.BBOX_MEANS = (0.0, 0.0, 0.0, 0.0); t.BBOX_STDS = (0.1, 0.1, 0.2, 0.2)
    t.LEARN_NMS = False; t.JOINT_TRAINING = False; t.FIRST_N = 100
    e = c.TEST = AttrDict()
    e.HAS_RPN = False; e.BATCH_IMAGES = 1; e.TOP_ROIS = 2000; e.CXX_PROPOSAL = True
    e.RPN_NMS_THRESH = 0.7; e.RPN_PRE_NMS_TOP_N = 6000; e.RPN_POST_NMS_TOP_N = 300; e.RPN_MIN_SIZE = 16
    e.PROPOSAL_NMS_THRESH = 0.7; e.PROPOSAL_PRE_NMS_TOP_N = 20000; e.PROPOSAL_POST_NMS_TOP_N = 2000; e.PROPOSAL_MIN_SIZE = 16
    e.SOFTNMS = False; e.LEARN_NMS = False; e.FIRST_N = 0; e.MERGE_METHOD = -1; e.NMS = 0.3; e.max_per_image = 300
    e.test_epoch = 0; e.LEARN_NMS_CLASS_SCORE_TH = 0.01
    return c


config = _defaults()


def _overlay(cfg, exp):
    for k, v in exp.items():
        if k not in cfg:
            raise ValueError("key must exist in config.py")                 # config.py:197-198
        if isinstance(v, dict):
            if k == 'TRAIN' and 'BBOX_WEIGHTS' in v:
                v = dict(v, BBOX_WEIGHTS=np.array(v['BBOX_WEIGHTS']))
            if k == 'network' and 'PIXEL_MEANS' in v:
                v = dict(v, PIXEL_MEANS=np.array(v['PIXEL_MEANS']))
            for vk, vv in v.items():
                cfg[k][vk] = vv
        elif k == 'SCALES':
            cfg[k][0] = tuple(v)
        else:
            cfg[k] = v
    return cfg


def update_config(config_file, cfg=None):
    """Overlay an experiment YAML on the tree (config.py:177-198); `cfg` defaults to the module-level `config`."""
    import yaml
    with open(config_file) as f:
        exp = yaml.safe_load(f)
    return _overlay(config if cfg is None else cfg, exp)


_COCO_E2E = dict(
    MXNET_VERSION='mxnet_v1.1.0', CLASS_AGNOSTIC=True, SCALES=(600, 1000), default=dict(frequent=100, kvstore='device'),
    network=dict(PIXEL_MEANS=[103.06, 115.90, 123.15], IMAGE_STRIDE=0, RCNN_FEAT_STRIDE=16, RPN_FEAT_STRIDE=16,
                 FIXED_PARAMS=['conv1', 'bn_conv1', 'res2', 'bn2', 'gamma', 'beta'],
                 FIXED_PARAMS_SHARED=['conv1', 'bn_conv1', 'res2', 'bn2', 'res3', 'bn3', 'res4', 'bn4', 'gamma', 'beta'],
                 ANCHOR_RATIOS=[0.5, 1, 2], ANCHOR_SCALES=[4, 8, 16, 32], NUM_ANCHORS=12),
    dataset=dict(NUM_CLASSES=81, dataset='coco', image_set='train2014+valminusminival2014', test_image_set='minival2014'),
    TRAIN=dict(lr=0.0005, lr_step='5.33', end_epoch=8, ENABLE_OHEM=True, BATCH_IMAGES=1, END2END=True, BATCH_ROIS=-1,
               BATCH_ROIS_OHEM=128, BG_THRESH_LO=0, CXX_PROPOSAL=False, RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300,
               RPN_MIN_SIZE=0, BBOX_NORMALIZATION_PRECOMPUTED=True),
    TEST=dict(HAS_RPN=True, BATCH_IMAGES=1, CXX_PROPOSAL=False, RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300, RPN_MIN_SIZE=0,
              PROPOSAL_MIN_SIZE=0, NMS=0.6, SOFTNMS=True, test_epoch=8, max_per_image=100))

_S = 'resnet_v1_101_rcnn'
_A = '_attention_1024_pairwise_position_multi_head_16'
_LN = dict(network=dict(NMS_TARGET_THRESH='0.5, 0.6, 0.7, 0.8, 0.9'), TRAIN=dict(LEARN_NMS=True, FIRST_N=100, JOINT_TRAINING=True),
           TEST=dict(NMS=10.0, LEARN_NMS=True, LEARN_NMS_CLASS_SCORE_TH=0.01, FIRST_N=100))
_FPN = dict(SCALES=(800, 1000), network=dict(IMAGE_STRIDE=32, ROIDispatch=True, USE_NONGT_INDEX=False),
            TRAIN=dict(lr=0.00125, END2END=False, TOP_ROIS=1000, BATCH_ROIS_OHEM=512), TEST=dict(HAS_RPN=False, TOP_ROIS=1000))
_FPN_REL = dict(network=dict(USE_NONGT_INDEX=True))
_NMS03 = dict(TEST=dict(SOFTNMS=False, NMS=0.3))
_FPN_LN = dict(TRAIN=dict(FIRST_N=150), TEST=dict(FIRST_N=150, LEARN_NMS_CLASS_SCORE_TH=0.05))
_LN_ONLY = dict(network=dict(FIXED_PARAMS=['conv1', 'bn_conv1', 'res2', 'bn2', 'res3', 'bn3', 'res4', 'bn4', 'gamma', 'beta',
                                           'rpn_conv_3x3', 'res5', 'bn5', 'fc_new', 'conv_new_1', 'cls_score', 'bbox_pred']),
                TRAIN=dict(end_epoch=3, lr_step='2.0', ENABLE_OHEM=False, JOINT_TRAINING=False), TEST=dict(test_epoch=3))

#: experiment name -> (symbol class name, overlays on _COCO_E2E); names are the YAML stems minus the common prefix
EXPERIMENTS = {
    'rcnn_end2end_8epoch': (_S, []),
    'rcnn_end2end_relation_8epoch': (_S + _A, [dict(TEST=dict(test_epoch=7))]),
    'rcnn_end2end_relation_learn_nms_8epoch': (_S + _A + '_learn_nms', [_LN]),
    'rcnn_end2end_learn_nms_3epoch': (_S + '_learn_nms_1024' + _A, [_LN, _LN_ONLY]),
    'rcnn_dcn_end2end_8epoch': (_S + '_dcn', []),
    'rcnn_dcn_end2end_relation_8epoch': (_S + '_dcn' + _A, [_NMS03]),
    'rcnn_dcn_end2end_relation_learn_nms_8epoch': (_S + '_dcn' + _A + '_learn_nms', [_LN]),
    'rcnn_fpn_8epoch': (_S + '_fpn', [_FPN]),
    'rcnn_fpn_relation_8epoch': (_S + '_fpn' + _A, [_FPN, _FPN_REL, _NMS03]),
    'rcnn_fpn_relation_learn_nms_8epoch': (_S + '_fpn' + _A + '_learn_nms', [_FPN, _FPN_REL, _LN, _FPN_LN]),
}


def experiment(name):
    """Fresh config tree of one shipped experiment (key = YAML file stem without `resnet_v1_101_coco_trainvalminus_`)."""
    symbol, overlays = EXPERIMENTS[name]
    cfg = _overlay(_defaults(), copy.deepcopy(_COCO_E2E))
    for o in overlays:
        _overlay(cfg, copy.deepcopy(o))
    cfg.symbol = symbol
    return cfg
