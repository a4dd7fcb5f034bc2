"""Oracle: box-geometry features and the object-relation module (numpy).
TEST INFRASTRUCTURE ONLY.

Follows relation_rcnn/symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_
multi_head_16.py (SYM_REL): :46-83 extract_position_matrix, :29-44
extract_position_embedding, :85-151 attention_module_multi_head, :244-276 the 2FC
head with two relation modules.  The wiring is pinned by tests/golden (the
reference's own methods executed on the numpy MXNet stand-in); the semantics of
the individual MXNet operators are restated.

`dtype=np.float32` reproduces the fp32 op sequence of the graph (element-wise ops
rounded per op, contractions accumulated in float64 then rounded once);
`dtype=np.float64` is the exact shadow used to bound conditioning.
"""
import math

import numpy as np


def _mm(a, b, dtype):
    return (a.astype(np.float64) @ b.astype(np.float64)).astype(dtype)


def cr(f, x):
    """Correctly rounded transcendental in the dtype of x: evaluate in float64, round
    once.  (numpy's float32 sin/cos/log/exp are SIMD approximations that are off by
    1 ulp for 8-40 % of arguments and vary between numpy builds; the geometry chain
    log -> x100 -> sin amplifies a 1-ulp log difference to ~5e-5, so the oracle pins
    the portable definition.  The HIP path evaluates log the same way.)"""
    x = np.asarray(x)
    return f(x.astype(np.float64)).astype(x.dtype)


def position_matrix(boxes, nongt_dim, dtype=np.float32):
    """boxes [N,4] (x1,y1,x2,y2) -> [N, nongt_dim, 4].  SYM_REL:46-83:
    w = x2-x1+1 (:59), cx = 0.5(x1+x2) (:61); dx = log(max(|cx_i-cx_j| / w_i, 1e-3))
    (:64-67), same for y (:68-71), log(w_i/w_j), log(h_i/h_j) (:72-77); keys are the
    first nongt_dim boxes (:79-81)."""
    b = np.asarray(boxes, dtype=dtype)
    x1, y1, x2, y2 = (b[:, i:i + 1] for i in range(4))
    one, half = dtype(1.0), dtype(0.5)
    w = x2 - x1 + one
    h = y2 - y1 + one
    cx = half * (x1 + x2)
    cy = half * (y1 + y2)
    dx = cr(np.log, np.maximum(np.abs((cx - cx.T) / w), dtype(1e-3)))
    dy = cr(np.log, np.maximum(np.abs((cy - cy.T) / h), dtype(1e-3)))
    dw = cr(np.log, w / w.T)
    dh = cr(np.log, h / h.T)
    return np.stack([m[:, :nongt_dim] for m in (dx, dy, dw, dh)], axis=2).astype(dtype)


def embedding_divisors(feat_dim=64, wave_length=1000, dtype=np.float32):
    """dim_mat of SYM_REL:32-35: wave_length ** ((8/feat_dim) * k), k < feat_dim/8,
    evaluated in `dtype` (MXNet arange/broadcast_power are fp32)."""
    k = np.arange(0, feat_dim // 8).astype(dtype)
    return np.power(dtype(wave_length), dtype(8.0 / feat_dim) * k).astype(dtype)


def position_embedding(pos_mat, feat_dim=64, wave_length=1000, dtype=np.float32):
    """[N,M,4] -> [N,M,feat_dim].  SYM_REL:29-44: arg = (100*p) / dim_mat (:36-37),
    per component [sin x8, cos x8] (:38-43)."""
    p = np.asarray(pos_mat, dtype=dtype)
    div = (dtype(100.0) * p)[..., None] / embedding_divisors(feat_dim, wave_length, dtype)
    emb = np.concatenate((cr(np.sin, div), cr(np.cos, div)), axis=3)
    return emb.reshape(p.shape[0], p.shape[1], feat_dim).astype(dtype)


def relation_module(roi_feat, pos_emb, params, index=1, nongt_dim=None, fc_dim=16,
                    feat_dim=1024, dim=(1024, 1024, 1024), group=16, dtype=np.float32,
                    return_intermediates=False, prefix=''):
    """attention_module_multi_head, SYM_REL:85-151.

    roi_feat [N, feat_dim]; pos_emb [N, M, emb]; params holds
    pair_pos_fc1_{i}_{weight[fc_dim,emb],bias}, query_{i}_*, key_{i}_* [dim, feat_dim],
    linear_out_{i}_weight [dim2, feat_dim, 1, 1] (num_group=fc_dim), _bias [dim2].
      G = relu(E Wp^T + bp)                                   :109-116
      Q = F Wq^T + bq ; K = F[:M] Wk^T + bk  (heads of dim/group) :120-129
      A = (1/sqrt(d_head)) Q_h K_h^T                          :132-135
      L = log(max(G, 1e-6)) + A        <- "attention logits"  :139
      S = softmax over keys                                   :140
      O[n,h,:] = sum_m S[n,h,m] F[m,:]   (V = raw features)   :130,144
      Y = grouped 1x1 conv over [h*feat_dim + c]              :146-150
    """
    f = np.asarray(roi_feat, dtype=dtype)
    e = np.asarray(pos_emb, dtype=dtype)
    n = f.shape[0]
    m = e.shape[1] if nongt_dim is None else nongt_dim
    def g(k):
        stem, kind = k.rsplit('_', 1)
        return np.asarray(params['%s%s_%d_%s' % (prefix, stem, index, kind)], dtype=dtype)
    wp, bp = g('pair_pos_fc1_weight'), g('pair_pos_fc1_bias')
    wq, bq = g('query_weight'), g('query_bias')
    wk, bk = g('key_weight'), g('key_bias')
    wo, bo = g('linear_out_weight'), g('linear_out_bias')
    fk = f[:m]
    d_head = dim[1] // group
    # geometry weight
    aff_w = np.maximum(_mm(e.reshape(n * m, -1), wp.T, np.float64) + bp.astype(np.float64), 0.0)
    aff_w = aff_w.astype(dtype).reshape(n, m, fc_dim).transpose(0, 2, 1)       # [N, h, M]
    q = (_mm(f, wq.T, np.float64) + bq.astype(np.float64)).astype(dtype)
    k = (_mm(fk, wk.T, np.float64) + bk.astype(np.float64)).astype(dtype)
    qh = q.reshape(n, group, d_head).transpose(1, 0, 2)
    kh = k.reshape(m, group, d_head).transpose(1, 0, 2)
    aff = np.matmul(qh.astype(np.float64), kh.astype(np.float64).transpose(0, 2, 1)).astype(dtype)
    aff_scale = (dtype(1.0 / math.sqrt(float(d_head))) * aff).transpose(1, 0, 2)  # [N, h, M]
    logits = (cr(np.log, np.maximum(aff_w, dtype(1e-6))) + aff_scale).astype(dtype)
    l64 = logits.astype(np.float64)
    ex = np.exp(l64 - l64.max(axis=2, keepdims=True))
    soft = (ex / ex.sum(axis=2, keepdims=True)).astype(dtype)
    out_t = _mm(soft.reshape(n * fc_dim, m), fk, dtype).reshape(n, fc_dim, feat_dim)
    wo_g = wo.reshape(fc_dim, dim[2] // fc_dim, feat_dim)
    y = np.einsum('nhc,hoc->nho', out_t.astype(np.float64), wo_g.astype(np.float64))
    y = (y.reshape(n, dim[2]) + bo.astype(np.float64)).astype(dtype)
    if return_intermediates:
        return dict(aff_weight=aff_w, q=q, k=k, logits=logits, softmax=soft, output_t=out_t,
                    output=y)
    return y


def fc(x, w, b, dtype=np.float32):
    """mx.symbol.FullyConnected: y = x W^T + b with W [out, in]."""
    x = np.asarray(x, dtype=dtype)
    return (_mm(x.reshape(x.shape[0], -1), np.asarray(w, dtype=dtype).T, np.float64)
            + np.asarray(b, dtype=np.float64)).astype(dtype)


def relation_head(roi_pool, rois, params, nongt_dim=None, num_classes=81, dtype=np.float32,
                  return_intermediates=False):
    """2FC head with two relation modules, SYM_REL:254-280.

    roi_pool [N, 256, 7, 7]; rois [N, 5]; the relation input is the PRE-ReLU fc
    output and the residual add precedes the ReLU (:263-276)."""
    n = roi_pool.shape[0]
    m = n if nongt_dim is None else nongt_dim
    boxes = np.asarray(rois)[:, 1:5]
    pm = position_matrix(boxes, m, dtype)
    pe = position_embedding(pm, 64, 1000, dtype)
    p = params
    fc1 = fc(roi_pool, p['fc_new_1_weight'], p['fc_new_1_bias'], dtype)
    att1 = relation_module(fc1, pe, p, 1, m, dtype=dtype)
    x1 = np.maximum(fc1 + att1, dtype(0))
    fc2 = fc(x1, p['fc_new_2_weight'], p['fc_new_2_bias'], dtype)
    att2 = relation_module(fc2, pe, p, 2, m, dtype=dtype)
    x2 = np.maximum(fc2 + att2, dtype(0))
    cls_score = fc(x2, p['cls_score_weight'], p['cls_score_bias'], dtype)
    bbox = fc(x2, p['bbox_pred_weight'], p['bbox_pred_bias'], dtype)
    z = cls_score.astype(np.float64)
    ez = np.exp(z - z.max(axis=1, keepdims=True))
    cls_prob = (ez / ez.sum(axis=1, keepdims=True)).astype(dtype)
    if return_intermediates:
        return dict(fc_new_1=fc1, attention_1=att1, fc_all_1_relu=x1, fc_new_2=fc2,
                    attention_2=att2, fc_all_2_relu=x2, cls_score=cls_score,
                    cls_prob=cls_prob, bbox_pred=bbox, position_embedding=pe)
    return cls_prob, bbox


def init_relation_params(rng, index, feat_dim=1024, dim=(1024, 1024, 1024), fc_dim=16,
                         emb_dim=64, std=0.01, prefix=''):
    """N(0, std) weights / zero biases as init_weight_attention_multi_head, SYM_REL:327-344."""
    f32 = np.float32
    p = {}
    def nrm(*shape):
        return rng.normal(0.0, std, size=shape).astype(f32)
    p['%spair_pos_fc1_%d_weight' % (prefix, index)] = nrm(fc_dim, emb_dim)
    p['%spair_pos_fc1_%d_bias' % (prefix, index)] = np.zeros(fc_dim, f32)
    p['%squery_%d_weight' % (prefix, index)] = nrm(dim[0], feat_dim)
    p['%squery_%d_bias' % (prefix, index)] = np.zeros(dim[0], f32)
    p['%skey_%d_weight' % (prefix, index)] = nrm(dim[1], feat_dim)
    p['%skey_%d_bias' % (prefix, index)] = np.zeros(dim[1], f32)
    p['%slinear_out_%d_weight' % (prefix, index)] = nrm(dim[2], feat_dim, 1, 1)
    p['%slinear_out_%d_bias' % (prefix, index)] = np.zeros(dim[2], f32)
    return p
