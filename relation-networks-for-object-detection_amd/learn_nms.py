"""Host side of the learned duplicate-removal head (test time).

Mirrors relation_rcnn/operator_py/learn_nms.py (LearnNmsOperator.forward :238-401) and the merge of
symbols/..._learn_nms.py:553-560, batched over B images, with no host synchronisation: the
reference's numpy class filtering (:296-303) becomes an on-device valid-class predicate.
"""
import ctypes

import torch

from . import ops
from . import lib as _lib
from .relation import pack_pair_pos


def rank_embedding(rank_dim, feat_dim=1024, wave_length=1000.0):
    """learn_nms.py:129-140 in float32 (MXNet arange / broadcast_power / broadcast_div), sin/cos
    correctly rounded.  A constant of the network -> evaluated once on the host."""
    r = torch.arange(0, rank_dim, dtype=torch.float32)[:, None]
    k = torch.arange(0, feat_dim // 2, dtype=torch.float32)
    dim = torch.pow(torch.tensor(float(wave_length), dtype=torch.float32), torch.tensor(2.0 / feat_dim, dtype=torch.float32) * k)[None, :]
    div = (r / dim).double()
    return torch.cat([torch.sin(div), torch.cos(div)], 1).float()


class LearnNMS(object):
    def __init__(self, params, num_fg_classes=80, first_n=100, num_thresh=5, class_thresh=0.01,
                 bbox_means=None, bbox_stds=None, merge_method=-1, score_thresh=1e-3, max_per_image=100,
                 dtype=torch.bfloat16, device='cuda'):
        t = lambda x, dt: torch.as_tensor(x).to(device=device, dtype=dt).contiguous()
        p = params
        self.dtype, self.device = dtype, device
        self.C, self.F, self.T = num_fg_classes, first_n, num_thresh
        self.class_thresh, self.merge, self.score_thresh, self.max_per_image = class_thresh, merge_method, score_thresh, max_per_image
        self.means = None if bbox_means is None else (ctypes.c_float * 4)(*[float(v) for v in bbox_means])
        self.stds = None if bbox_stds is None else (ctypes.c_float * 4)(*[float(v) for v in bbox_stds])
        self.w_emb, self.b_emb = t(p['roi_feat_embedding_weight'], dtype), t(p['roi_feat_embedding_bias'], torch.float32)
        # rank_feat = FC(rank embedding): constant of the weights (learn_nms.py:327-330)
        re = rank_embedding(first_n, 1024).double()
        cpu = lambda x: torch.as_tensor(x).detach().cpu()
        self.rank_feat = t(re @ cpu(p['nms_rank_weight']).double().t() + cpu(p['nms_rank_bias']).double(), torch.float32)
        self.wqk = t(torch.cat([cpu(p['nms_query_1_weight']), cpu(p['nms_key_1_weight'])], 0), dtype)
        self.bqk = t(torch.cat([cpu(p['nms_query_1_bias']), cpu(p['nms_key_1_bias'])], 0), torch.float32)
        # grouped linear_out 16 x (128 -> 8): pad every head's 8 value rows to the 64-wide attention tile
        wo = cpu(p['nms_linear_out_1_weight']).reshape(128, 128).float()
        bo = cpu(p['nms_linear_out_1_bias']).float()
        wpad, bpad = torch.zeros(1024, 128), torch.zeros(1024)
        for h in range(16):
            wpad[h * 64:h * 64 + 8] = wo[h * 8:(h + 1) * 8]
            bpad[h * 64:h * 64 + 8] = bo[h * 8:(h + 1) * 8]
        self.wout_pad, self.bout_pad = t(wpad, dtype), t(bpad, torch.float32)

        class _M(object):
            pass
        m = _M()
        m.wp = cpu(p['nms_pair_pos_fc1_1_weight']).float()
        m.bp = cpu(p['nms_pair_pos_fc1_1_bias']).float()
        self.wp_t, self.bp = pack_pair_pos([m], device)
        self.w_logit, self.b_logit = t(p['nms_logit_weight'], torch.float32), t(p['nms_logit_bias'], torch.float32)
        self._vwt = {}

    def forward(self, cls_score, bbox_pred, rois, im_info, feat, want_detections=True, n_valid=None):
        """cls_score [B,N,C+1] fp32, bbox_pred [B,N,4*num_reg] fp32, rois [B,N,5], im_info [B,3],
        feat = fc_all_2_relu [B,N,1024] -> dict(nms_multi_score [B,F,C,T], sorted_bbox [B,F,C,4],
        sorted_score [B,F,C], nms_final_score [B,F,C], detections ...)."""
        B, N, C1 = cls_score.shape
        C, F, T = self.C, self.F, self.T
        dev = cls_score.device
        s = ops._stream()
        cs = cls_score.reshape(B * N, C1)
        bp = bbox_pred.reshape(B * N, -1)
        assert cs.stride(1) == 1 and bp.stride(1) == 1 and cs.dtype == torch.float32 and bp.dtype == torch.float32
        prob = torch.empty((B, N, C), device=dev, dtype=torch.float32)
        boxes = torch.empty((B, N, 4), device=dev, dtype=torch.float32)
        _lib.call('relnet_lnms_prepare_ex', cs.data_ptr(), cs.stride(0), bp.data_ptr(), bp.stride(0), rois.data_ptr(),
                  im_info.data_ptr(), prob.data_ptr(), boxes.data_ptr(), B, N, C1, 4, self.means, self.stds,
                  ops._ptr(n_valid), s)       # n_valid [B] int32: rows past it are padding (probability 0: they sort last)
        rank_idx = torch.empty((B, C, F), device=dev, dtype=torch.int32)
        sorted_score = torch.empty((B, F, C), device=dev, dtype=torch.float32)
        sorted_bbox = torch.empty((B, F, C, 4), device=dev, dtype=torch.float32)
        class_boxes = torch.empty((B, C, F, 4), device=dev, dtype=torch.float32)
        class_max = torch.empty((B, C), device=dev, dtype=torch.float32)
        _lib.call('relnet_lnms_sort', prob.data_ptr(), boxes.data_ptr(), rank_idx.data_ptr(), sorted_score.data_ptr(),
                  sorted_bbox.data_ptr(), class_boxes.data_ptr(), class_max.data_ptr(), B, N, C, F, s)
        # (feat may carry rows past N -- the gt rows of a training graph: learn_nms.py:335-339 embeds fc_all_2_relu unsliced and
        #  takes rows by rank, so the ranks only ever address its first N rows)
        Nf = feat.shape[1]
        assert Nf >= N
        roi_emb = ops.gemm_nt(feat.reshape(B * Nf, -1).to(self.dtype), self.w_emb, self.b_emb)         # [B*Nf,128]
        x = torch.empty((B, C, F, 128), device=dev, dtype=self.dtype)
        _lib.call('relnet_lnms_embed', roi_emb.data_ptr(), self.rank_feat.data_ptr(), rank_idx.data_ptr(), x.data_ptr(),
                  B, Nf, C, F, 128, ops._dt(x), s)
        BC = B * C
        xr = x.view(BC, F, 128)
        qk = ops.gemm_nt(xr.reshape(BC * F, 128), self.wqk, self.bqk).view(BC, F, 2048)
        Mpad = ops.pad32(F)
        key = (BC, Mpad, F)            # keyed on F too: pad columns [F, Mpad) must stay zero
        if key not in self._vwt:
            self._vwt[key] = torch.zeros((BC, 1024, Mpad), device=dev, dtype=self.dtype)
        vwt = self._vwt[key]
        ops.gemm_nt(self.wout_pad, xr, out=vwt, n_cols=F)
        bias = ops.geometry_bias(class_boxes.view(BC, F, 4), self.wp_t, self.bp, F, half=self.dtype == torch.bfloat16)[0]
        att, _, _ = ops.relation_attention(qk[:, :, :1024], qk[:, :, 1024:], vwt, bias, bout=self.bout_pad, M=F)
        multi = torch.empty((B, F, C, T), device=dev, dtype=torch.float32)
        final = torch.empty((B, F, C), device=dev, dtype=torch.float32)
        dets = torch.zeros((B, C, F, 5), device=dev, dtype=torch.float64) if want_detections else None
        counts = torch.empty((B, C), device=dev, dtype=torch.int32) if want_detections else None
        _lib.call('relnet_lnms_score', x.data_ptr(), att.data_ptr(), self.w_logit.data_ptr(), self.b_logit.data_ptr(),
                  sorted_score.data_ptr(), sorted_bbox.data_ptr(), class_max.data_ptr(), im_info.data_ptr(),
                  multi.data_ptr(), final.data_ptr(), ops._ptr(dets), ops._ptr(counts), B, C, F, 128, T, 16, 8, 64,
                  self.merge, float(self.class_thresh), float(self.score_thresh), ops._dt(x), s)
        out = dict(nms_multi_score=multi, sorted_bbox=sorted_bbox, sorted_score=sorted_score, nms_final_score=final)
        if want_detections:
            det, det_count, thresh, total = ops.image_topk(dets, counts, self.max_per_image)
            out.update(detections=det, num_detections=det_count, class_dets=dets, class_counts=counts)
        return out
