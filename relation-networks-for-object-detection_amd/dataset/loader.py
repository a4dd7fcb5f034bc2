"""Data iterators of `relation_rcnn/core/loader.py` for the PyTorch-hosted path.

  TestLoader    :25-168   one image per step (`get_rpn_testbatch` / `get_rcnn_testbatch`): data, im_info (+ the image's
                          precomputed proposals when `has_rpn` is False)
  AnchorLoader  :402-607  end2end training: image, im_info, gt_boxes and the RPN anchor targets (`assign_anchor`,
                          lib/rpn/rpn.py:80-244 -- host numpy as in the reference: train.assign_anchor -- or, with
                          device_targets=True, left to the trainer's device kernel relnet_assign_anchor)
  ROIIter       :170-400  training on precomputed proposals (FPN / alternate training): image, im_info, gt_boxes,
                          proposals.  The reference's `get_rcnn_batch` also computes labels / regression targets / the
                          pyramid dispatch on the host; here those are device kernels inside the trainer
                          (relnet_proposal_target, relnet_fpn_roi_dispatch), so the iterator delivers their inputs.
Shuffling with aspect-ratio grouping is the reference's (:496-513): horizontal and vertical images are permuted
separately, batches are formed inside a group, then the batches are permuted.
Batches are dicts of torch tensors (host); `.cuda()` them or pass `device=`.
"""
import numpy as np
import torch

from .image import get_image, tensor_vstack


def _gt_boxes(rec):
    """get_rpn_batch (lib/rpn/rpn.py:62-69): rows (x1, y1, x2, y2, cls) of the entries with a class."""
    if rec['gt_classes'].size > 0:
        idx = np.where(rec['gt_classes'] != 0)[0]
        gt = np.empty((len(idx), 5), dtype=np.float32)
        gt[:, 0:4] = rec['boxes'][idx, :]
        gt[:, 4] = rec['gt_classes'][idx]
        return gt
    return np.empty((0, 5), dtype=np.float32)


def _pad_rows(arrs, width, fill=0.0):
    n = max([a.shape[0] for a in arrs] + [1])
    out = np.full((len(arrs), n, width), fill, dtype=np.float32)
    for i, a in enumerate(arrs):
        out[i, :a.shape[0]] = a
    return out, np.array([a.shape[0] for a in arrs], dtype=np.int32)


class _Iter(object):
    drop_last = True

    def __init__(self, roidb, config, batch_size, shuffle, aspect_grouping, seed, device):
        self.roidb, self.cfg, self.batch_size = roidb, config, batch_size
        self.shuffle, self.aspect_grouping, self.device = shuffle, aspect_grouping, device
        self.size = len(roidb)
        self.index = np.arange(self.size)
        self.rng = np.random.RandomState(seed)
        self.cur = 0
        self.reset()

    def reset(self):
        self.cur = 0
        if not self.shuffle:
            return
        if self.aspect_grouping:
            widths = np.array([r['width'] for r in self.roidb])
            heights = np.array([r['height'] for r in self.roidb])
            horz = widths >= heights
            inds = np.hstack((self.rng.permutation(np.where(horz)[0]), self.rng.permutation(np.where(~horz)[0])))
            extra = inds.shape[0] % self.batch_size
            n_full = inds.shape[0] - extra
            if n_full:
                blocks = np.reshape(inds[:n_full], (-1, self.batch_size))
                inds[:n_full] = np.reshape(blocks[self.rng.permutation(blocks.shape[0]), :], (-1,))
            self.index = inds
        else:
            self.rng.shuffle(self.index)

    def __iter__(self):
        return self

    def __len__(self):
        return self.size // self.batch_size if self.drop_last else -(-self.size // self.batch_size)

    def _tensors(self, d):
        return {k: (torch.as_tensor(v).to(self.device) if isinstance(v, np.ndarray) else v) for k, v in d.items()}


class TestLoader(_Iter):
    drop_last = False
    __test__ = False          # not a pytest class

    def __init__(self, roidb, config, batch_size=1, shuffle=False, has_rpn=False, device='cpu'):
        self.has_rpn = has_rpn
        super(TestLoader, self).__init__(roidb, config, batch_size, shuffle, False, 0, device)

    def __next__(self):
        if self.cur >= self.size:
            raise StopIteration
        lo, hi = self.cur, min(self.cur + self.batch_size, self.size)
        recs = [self.roidb[self.index[i]] for i in range(lo, hi)]
        self.cur += self.batch_size
        ims, recs = get_image(recs, self.cfg)
        out = dict(data=tensor_vstack(ims), im_info=np.array([r['im_info'] for r in recs], dtype=np.float32),
                   index=[int(self.index[i]) for i in range(lo, hi)])
        if not self.has_rpn:          # get_rcnn_testbatch: the image's own proposals (already scaled by get_image)
            out['proposals'], out['num_proposals'] = _pad_rows([r['boxes'].astype(np.float32) for r in recs], 4)
        return self._tensors(out)


def _conv4_size(n):
    """conv1 7x7/2 pad 3 -> pool1 3x3/2 'full' -> res3 /2 -> res4 /2 (SYM_BASE:30-36,99,179)."""
    n = (n + 2 * 3 - 7) // 2 + 1
    n = -(-(n - 3) // 2) + 1
    n = (n - 1) // 2 + 1
    return (n - 1) // 2 + 1


class AnchorLoader(_Iter):
    def __init__(self, roidb, config, batch_size=1, shuffle=False, aspect_grouping=False, seed=0, device='cpu',
                 feat_shape_fn=None, device_targets=False):
        """device_targets: leave `assign_anchor` to the trainer (relnet_assign_anchor on the GPU, Trainer.rpn_targets): the
        batch then carries only data / im_info / gt_boxes / num_gt and the loader does no per-anchor work on the host."""
        self.device_targets = device_targets
        self.feat_shape_fn = feat_shape_fn or (lambda h, w: (_conv4_size(h), _conv4_size(w)))
        super(AnchorLoader, self).__init__(roidb, config, batch_size, shuffle, aspect_grouping, seed, device)

    def __next__(self):
        from .. import train
        if self.cur + self.batch_size > self.size:
            raise StopIteration
        recs = [self.roidb[self.index[i]] for i in range(self.cur, self.cur + self.batch_size)]
        self.cur += self.batch_size
        ims, recs = get_image(recs, self.cfg)
        data = tensor_vstack(ims)
        H, W = data.shape[2], data.shape[3]
        tc = train.TrainConfig()
        tc.anchor_scales, tc.anchor_ratios = tuple(self.cfg.network.ANCHOR_SCALES), tuple(self.cfg.network.ANCHOR_RATIOS)
        tc.rpn_batch_size, tc.rpn_fg_fraction = self.cfg.TRAIN.RPN_BATCH_SIZE, self.cfg.TRAIN.RPN_FG_FRACTION
        tc.rpn_positive_overlap, tc.rpn_negative_overlap = self.cfg.TRAIN.RPN_POSITIVE_OVERLAP, self.cfg.TRAIN.RPN_NEGATIVE_OVERLAP
        gts = [_gt_boxes(r) for r in recs]
        if self.device_targets:
            gt_pad, num_gt = _pad_rows(gts, 5)
            return self._tensors(dict(data=data, im_info=np.array([r['im_info'] for r in recs], dtype=np.float32),
                                      gt_boxes=gt_pad, num_gt=num_gt))
        fh, fw = self.feat_shape_fn(H, W)
        labs, tgts, wgts = [], [], []
        for r, gt in zip(recs, gts):
            L, T, Wt = train.assign_anchor((fh, fw), gt, (r['im_info'][0], r['im_info'][1]), tc, seed=int(self.rng.randint(1 << 30)))
            labs.append(L); tgts.append(T); wgts.append(Wt)
        gt_pad, num_gt = _pad_rows(gts, 5)
        return self._tensors(dict(data=data, im_info=np.array([r['im_info'] for r in recs], dtype=np.float32), gt_boxes=gt_pad,
                                  num_gt=num_gt, label=np.stack(labs), bbox_target=np.stack(tgts), bbox_weight=np.stack(wgts)))


class ROIIter(_Iter):
    def __init__(self, roidb, config, batch_size=1, shuffle=False, aspect_grouping=False, seed=0, device='cpu'):
        super(ROIIter, self).__init__(roidb, config, batch_size, shuffle, aspect_grouping, seed, device)

    def __next__(self):
        if self.cur + self.batch_size > self.size:
            raise StopIteration
        recs = [self.roidb[self.index[i]] for i in range(self.cur, self.cur + self.batch_size)]
        self.cur += self.batch_size
        ims, recs = get_image(recs, self.cfg)
        props, gts = [], []
        for r in recs:
            is_gt = np.asarray(r.get('is_gt', np.zeros(len(r['boxes'])))) > 0
            props.append(r['boxes'][~is_gt].astype(np.float32))                 # proposals (gt rows are appended on device)
            gt = np.hstack((r['boxes'][is_gt].astype(np.float32), np.asarray(r['max_classes'])[is_gt, None].astype(np.float32))) \
                if is_gt.any() else np.empty((0, 5), np.float32)
            gts.append(gt)
        top = self.cfg.TRAIN.TOP_ROIS
        if top > 0:                  # load_rpn_roidb's top_roi: TRUNCATE only (the reference keeps a variable roi count)
            props = [p[:top] for p in props]
        # zero rows up to a common length (TOP_ROIS when set: static shapes for a captured graph); `num_proposals` carries the
        # TRUE count of every image and the trainer masks the padded rows on the device (FPNTrainer.forward_backward)
        p_pad, num_p = _pad_rows(props + ([np.zeros((top, 4), np.float32)] if top > 0 else []), 4)
        if top > 0:
            p_pad, num_p = p_pad[:-1], num_p[:-1]
        gt_pad, num_gt = _pad_rows(gts, 5)
        return self._tensors(dict(data=tensor_vstack(ims), im_info=np.array([r['im_info'] for r in recs], dtype=np.float32),
                                  proposals=p_pad, num_proposals=num_p, gt_boxes=gt_pad, num_gt=num_gt))
