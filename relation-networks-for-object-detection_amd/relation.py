"""Host side of the object-relation module and the 2FC relation head.

Mirrors the reference's symbol-level interface for this path
(relation_rcnn/symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py):
`attention_module_multi_head(roi_feat, position_embedding, nongt_dim, fc_dim, feat_dim, dim,
group, index)` (:85-151) and the fc_new_1 -> relation -> fc_new_2 -> relation -> cls/bbox head
(:254-280), with parameters under the reference's names (`query_1_weight`, ...).  The
[N, M, 64] position embedding is never materialised: the geometry kernel consumes the ROI
boxes directly, so `position_embedding` is replaced by the rois themselves.
"""

import torch

from . import ops


class RelationParams(object):
    """Relation-module weights packed once for the kernels (all device tensors).

    wqk  [2048, F]  = [query_i_weight; key_i_weight]     bqk [2048]
    wout [dim2, F]  = linear_out_i_weight[:, :, 0, 0]     bout [dim2]
    wp_t/bp are shared across the modules that see the same boxes (see RelationHead)."""

    def __init__(self, params, index, dtype, device, prefix=''):
        g = lambda n: params['%s%s_%d_%s' % (prefix, n.rsplit('_', 1)[0], index, n.rsplit('_', 1)[1])]
        t = lambda x, dt: torch.as_tensor(x).to(device=device, dtype=dt).contiguous()
        self.wqk = t(torch.cat([torch.as_tensor(g('query_weight')), torch.as_tensor(g('key_weight'))], 0), dtype)
        self.bqk = t(torch.cat([torch.as_tensor(g('query_bias')), torch.as_tensor(g('key_bias'))], 0), torch.float32)
        wo = torch.as_tensor(g('linear_out_weight'))
        self.wout = t(wo.reshape(wo.shape[0], wo.shape[1]), dtype)
        self.bout = t(g('linear_out_bias'), torch.float32)
        self.wp = torch.as_tensor(g('pair_pos_fc1_weight')).to(torch.float32)      # [16, 64]
        self.bp = torch.as_tensor(g('pair_pos_fc1_bias')).to(torch.float32)
        self.wp_dev, self.bp_dev = t(self.wp, torch.float32), t(self.bp, torch.float32)   # operands of the fused kernel


def pack_pair_pos(mods, device):
    """[16,64] pair_pos_fc1 weights of the modules -> wp_t [64, nmod*16], bp [nmod*16]."""
    # (one launch per output: the transposed views are concatenated straight into the contiguous result; a single module's bias is used as it is --
    #  a training step packs these from the master weights every step)
    if len(mods) == 1:
        return mods[0].wp.t().contiguous().to(device), mods[0].bp.contiguous().to(device)
    wp_t = torch.cat([m.wp.t() for m in mods], 1).contiguous().to(device)
    bp = torch.cat([m.bp for m in mods], 0).to(device)
    return wp_t, bp


def attention_module_multi_head(roi_feat, rois, params, nongt_dim=None, fc_dim=16, feat_dim=1024,
                                dim=(1024, 1024, 1024), group=16, index=1, dtype=None,
                                return_logits=False, packed=None, bias=None, fused=False, key_count=None):
    """Drop-in for SYM_REL.attention_module_multi_head (:85-151) on device tensors.

    roi_feat [N, feat_dim] or [B, N, feat_dim]; `rois` [.., N, 4|5] takes the place of the
    reference's `position_embedding` argument (the embedding is computed inside the kernel);
    returns the module output [.., N, dim[2]] (and the logits `weighted_aff` [.., N, 16, M])."""
    assert fc_dim == 16 and group == 16 and dim == (1024, 1024, 1024) and feat_dim == 1024, \
        "the HIP kernels are specialised to the reference configuration (16 heads x 64)"
    squeeze = roi_feat.dim() == 2
    f = roi_feat[None] if squeeze else roi_feat
    bx = rois[None] if rois.dim() == 2 else rois
    dtype = dtype or f.dtype
    f = f.to(dtype).contiguous()
    B, N, _ = f.shape
    M = N if nongt_dim is None else nongt_dim
    mod = packed or RelationParams(params, index, dtype, f.device)
    bx = bx.to(torch.float32).contiguous()
    if bias is None and not (fused and not return_logits and key_count is None and fused_ok(dtype, M)):
        wp_t, bp = pack_pair_pos([mod], f.device)
        bias = ops.geometry_bias(bx, wp_t, bp, M, half=(dtype == torch.bfloat16 and not return_logits))[0]
    y, _, logits = _module_forward(f, mod, bias, M, want_out=True, want_act=False,
                                   want_logits=return_logits, rois=bx, key_count=key_count)
    if squeeze:
        y = y[0]
        logits = logits[0] if logits is not None else None
    return (y, logits) if return_logits else y


def fused_ok(dtype, M, heads=16):
    """The one-kernel geometry + attention path (ops.relation_attention_fused) covers the bf16 throughput configuration."""
    return dtype == torch.bfloat16 and heads == 16 and M <= ops.FUSED_MAX_KEYS


def _module_forward(f, mod, bias, M, want_out, want_act, want_logits, vwt_buf=None, rois=None, key_count=None, cache=None):
    """`bias` None + `rois` given: fused geometry + attention kernel (no bias tensor, no logits output).
    key_count [B] int32: per-image number of real keys among the first M rows (ops.relation_attention).
    cache (dict, training): receives the projections qk [B,N,2d] and vwt [B,d,Mpad], which attention_module_backward reuses
    (they do not depend on the geometry weight; the module output is recomputed there, see its comment)."""
    B, N, F = f.shape
    qk = ops.gemm_nt(f.reshape(B * N, F), mod.wqk, mod.bqk).reshape(B, N, -1)
    Mpad = bias.shape[-1] if bias is not None else ops.pad32(M)
    if vwt_buf is None:
        vwt_buf = torch.zeros((B, mod.wout.shape[0], Mpad), device=f.device, dtype=f.dtype)
    # VW^T[b] = Wout F_b[:M]^T  (linear_out re-associated in front of the softmax sum)
    ops.gemm_nt(mod.wout, f[:, :M, :], out=vwt_buf, n_cols=M)
    d = mod.wqk.shape[0] // 2
    if bias is None:
        assert rois is not None and not want_logits and key_count is None
        y, act = ops.relation_attention_fused(qk[:, :, :d], qk[:, :M, d:], vwt_buf, rois, mod.wp_dev, mod.bp_dev,
                                              bout=mod.bout, resid=f if want_act else None, M=M, want_out=want_out,
                                              want_act=want_act)
        return y, act, None
    r = ops.relation_attention(qk[:, :, :d], qk[:, :M, d:], vwt_buf, bias, bout=mod.bout,
                               resid=f if want_act else None, M=M, want_out=want_out,
                               want_act=want_act, want_logits=want_logits, key_count=key_count)
    if cache is not None:
        cache.update(qk=qk, vwt=vwt_buf)
        if bias.dtype == torch.float32 and want_out and not want_logits:
            # training forward on the float32 geometry: the backward reuses ln G and the module output instead of recomputing them
            cache.update(bias=bias, y=r[0])
    return r


class RelationHead(object):
    """fc_new_1 -> relation_1 -> ReLU -> fc_new_2 -> relation_2 -> ReLU -> cls_score / bbox_pred
    (SYM_REL:254-280).  `dtype` bf16 is the throughput path, float32 the parity path."""

    def __init__(self, params, dtype=torch.bfloat16, device='cuda', fc1_perm=None, use_relation=True,
                 fc_names=('fc_new_1', 'fc_new_2'), fused=False):
        t = lambda x, dt: torch.as_tensor(x).to(device=device, dtype=dt).contiguous()
        self.dtype, self.device = dtype, device
        n1, n2 = fc_names          # ('roi_pool_fc1', 'roi_pool_fc2') in the FPN graphs
        w1 = torch.as_tensor(params[n1 + '_weight'])
        if fc1_perm is not None:            # (C,7,7) -> (7,7,C) input order for channels-last pooling
            w1 = w1[:, fc1_perm]
        self.w1, self.b1 = t(w1, dtype), t(params[n1 + '_bias'], torch.float32)
        self.w2, self.b2 = t(params[n2 + '_weight'], dtype), t(params[n2 + '_bias'], torch.float32)
        wcb = torch.cat([torch.as_tensor(params['cls_score_weight']), torch.as_tensor(params['bbox_pred_weight'])], 0)
        bcb = torch.cat([torch.as_tensor(params['cls_score_bias']), torch.as_tensor(params['bbox_pred_bias'])], 0)
        self.num_classes = int(params['cls_score_weight'].shape[0])
        self.wcb, self.bcb = t(wcb, dtype), t(bcb, torch.float32)
        self.use_relation = use_relation
        # fused: one geometry + attention kernel per module, no bias tensor in HBM -- built and parity-tested, but at B = 54
        # it is slower (272 us / module) than the matrix-core geometry kernel + LDS attention kernel
        self.fused = fused
        if use_relation:
            self.mods = [RelationParams(params, i, dtype, device) for i in (1, 2)]
            self.wp_t, self.bp = pack_pair_pos(self.mods, device)
        self._vwt = {}

    def _vwt_buf(self, B, Mpad, M):
        key = (B, Mpad, M)            # keyed on M too: a smaller M in the same padded width would inherit stale columns [M, Mpad)
        if key not in self._vwt:      # pad columns stay zero: the GEMM only writes [:, :, :M]
            self._vwt[key] = [torch.zeros((B, 1024, Mpad), device=self.device, dtype=self.dtype) for _ in range(2)]
        return self._vwt[key]

    def forward(self, pooled, rois, nongt_dim=None, return_intermediates=False, key_count=None):
        """pooled [B, N, 12544] (dtype), rois [B, N, 5] fp32 -> cls_score [B,N,C], bbox_pred [B,N,8]
        (fp32 logits; softmax / decoding live in postprocess).  key_count [B] int32: real rows per image when the roi
        buffer is padded to a fixed size (FPN dummy rois, truncated proposal lists); padded rows are no relation keys."""
        B, N, K = pooled.shape
        if not self.use_relation:        # plain 2FC head, resnet_v1_101_rcnn.py:125-134
            x1 = ops.gemm_nt(pooled.reshape(B * N, K), self.w1, self.b1, relu=True)
            x2 = ops.gemm_nt(x1, self.w2, self.b2, relu=True).reshape(B, N, -1)
            cb = ops.gemm_nt(x2.reshape(B * N, -1), self.wcb, self.bcb, out_dtype=torch.float32).reshape(B, N, -1)
            return cb[:, :, :self.num_classes], cb[:, :, self.num_classes:], x2
        M = N if nongt_dim is None else nongt_dim
        if self.fused and fused_ok(self.dtype, M) and key_count is None:
            bias = (None, None)                 # geometry evaluated inside the attention kernel of each module
        else:
            bias = ops.geometry_bias(rois, self.wp_t, self.bp, M, half=self.dtype == torch.bfloat16)   # [2,B,16,N,Mpad]
        vw = self._vwt_buf(B, ops.pad32(M), M)
        f1 = ops.gemm_nt(pooled.reshape(B * N, K), self.w1, self.b1).reshape(B, N, -1)
        y1, x1, _ = _module_forward(f1, self.mods[0], bias[0], M, return_intermediates, True, False, vw[0], rois, key_count)
        f2 = ops.gemm_nt(x1.reshape(B * N, -1), self.w2, self.b2).reshape(B, N, -1)
        y2, x2, _ = _module_forward(f2, self.mods[1], bias[1], M, return_intermediates, True, False, vw[1], rois, key_count)
        cb = ops.gemm_nt(x2.reshape(B * N, -1), self.wcb, self.bcb, out_dtype=torch.float32).reshape(B, N, -1)
        cls_score, bbox_pred = cb[:, :, :self.num_classes], cb[:, :, self.num_classes:]
        if return_intermediates:
            return dict(fc_new_1=f1, attention_1=y1, fc_all_1_relu=x1, fc_new_2=f2, attention_2=y2,
                        fc_all_2_relu=x2, cls_score=cls_score, bbox_pred=bbox_pred)
        return cls_score, bbox_pred, x2


class GradSink(object):
    """Where a trainer wants the gradients of one relation module (attention_module_backward(sink=...)): instead of returning fp32
    tensors that the caller concatenates and adds into its buffers (~20 elementwise launches per module), the backward writes

      wcat_t   [Fd, 3 d] bf16  ([Wq; Wk]^T | Wout^T side by side: ops.WeightRelayout group) -- operand of the ONE projection-backward GEMM
      resid    [B, N, Fd] bf16  gradient that bypasses the module (the residual path), added in that GEMM's epilogue
      wgrad    callable(dy2d [P, 3 d] bf16, x2d [P, Fd] bf16): accumulate d[Wq; Wk; Wout] = dy2d^T x2d (the trainer queues it for its
               grouped stream-K launch, straight into the flat gradient buffer)
      b_qk     fp32 view [2 d], b_out fp32 view [d] (or None: the caller sums it): bias gradients, accumulated by relnet_colsum_add
      dwp, dbp fp32 views [16, 64] / [16] of pair_pos_fc1's gradient: the geometry backward accumulates into them atomically
      defer    optional callable(fn, keep): see __init__
      scratch  callable(name, shape, dtype) -> persistent ZERO-initialised buffer (pad columns of the transposed operands stay zero
               from step to step: no per-step fill)"""

    def __init__(self, wcat_t, resid, wgrad, b_qk, b_out, dwp, dbp, scratch, defer=None):
        self.wcat_t, self.resid, self.wgrad, self.b_qk, self.b_out, self.dwp, self.dbp, self.scratch = \
            wcat_t, resid, wgrad, b_qk, b_out, dwp, dbp, scratch
        # defer: None, or callable(fn, keep) -- work that only produces PARAMETER gradients (the geometry backward) may be run by the trainer
        # beside the data-gradient chain (a side stream joined before the gradient bucket is announced); `keep` = the tensors fn reads
        self.defer = defer


def attention_module_backward(roi_feat, rois, params, d_out, nongt_dim=None, index=1, dtype=None, packed=None, key_count=None,
                              cache=None, sink=None):
    """Gradient of `attention_module_multi_head` (the adjoint MXNet's autograd derives from SYM_REL:85-151).

    roi_feat [B,N,1024] (or [N,1024]), rois [..,N,4|5], d_out = d loss / d module output, same shape.
    Returns dict: d_roi_feat [B,N,1024] and the parameter gradients under the reference's names
    (`query_i_weight`, `key_i_bias`, `linear_out_i_weight` [1024,1024,1,1], `pair_pos_fc1_i_weight`, ...), all fp32.

    Every contraction runs on the GEMM / attention kernels: weight gradients are A^T B products, formed as
    gemm_nt over transposed copies (relnet_transpose_2d); the forward is recomputed from roi_feat (nothing but
    the inputs has to be kept alive between forward and backward)."""
    squeeze = roi_feat.dim() == 2
    f = roi_feat[None] if squeeze else roi_feat
    bx = (rois[None] if rois.dim() == 2 else rois).to(torch.float32).contiguous()
    dY = d_out[None] if squeeze else d_out
    dtype = dtype or f.dtype
    f = f.to(dtype).contiguous()
    dY = dY.to(dtype).contiguous()
    B, N, Fd = f.shape
    M = N if nongt_dim is None else nongt_dim
    mod = packed or RelationParams(params, index, dtype, f.device)
    # fp32 ln G [B,16,N,Mpad] in float32 libm arithmetic and the module output y computed FROM IT (the softmax backward needs
    # D = dY.(y - bout) consistent with the softmax weights it re-derives from this G): taken from the training forward when it ran
    # on that geometry (`cache` holds bias / y / qk / vwt), recomputed otherwise.
    # Measured and rejected (r03, tools/dbg_geom.py + tests/test_gpu_train_step.py::test_fpn_training_step...): taking G from the
    # matrix-core kernel (fp16 products, hardware log / sin: |dG| <= 3e-4, 2e-5 on average) or from the forward's own fp16 bias
    # moves the pair_pos_fc1 gradient by 50 % in norm on rows whose keys are all (nearly) clamped -- there Z = sum_j G_j e^a_j is
    # ~1e-5 and d pre_j = e^a_j (ds_j - D) / Z turns an absolute G error of 1e-4 into an O(1) change; the reference's
    # log(max(G, 1e-6)) is that ill-conditioned.  The projections Q|K and VW^T do not depend on G and are
    # taken from the forward (`cache` of _module_forward).
    have = cache is not None and cache.get('bias') is not None and cache.get('y') is not None and cache.get('qk') is not None
    if have:
        bias = cache['bias']
    else:
        wp_t, bp = pack_pair_pos([mod], f.device)
        bias = ops.geometry_bias(bx, wp_t, bp, M, fast32=(dtype == torch.bfloat16))[0]
    Mpad = bias.shape[-1]
    d = mod.wqk.shape[0] // 2
    kpad = 64 if dtype == torch.bfloat16 else 16                        # GEMM K granularity
    if cache is not None and cache.get('qk') is not None and cache['vwt'].shape[-1] == Mpad:
        qk, vwt = cache['qk'], cache['vwt']
    else:
        qk = ops.gemm_nt(f.reshape(B * N, Fd), mod.wqk, mod.bqk).reshape(B, N, 2 * d)
        vwt = torch.zeros((B, d, Mpad), device=f.device, dtype=dtype)
        ops.gemm_nt(mod.wout, f[:, :M, :], out=vwt, n_cols=M)
    q, k = qk[:, :, :d], qk[:, :M, d:]
    y = cache['y'] if have else ops.relation_attention(q, k, vwt, bias, bout=mod.bout, M=M, want_out=True, key_count=key_count)[0]
    # ---- operand layouts of the backward kernels
    vw = ops.gemm_nt(f[:, :M, :].reshape(B * M, Fd) if M == N else f[:, :M, :].contiguous().reshape(B * M, Fd),
                     mod.wout).reshape(B, M, d)                          # F_K Wout^T, not transposed
    if ops.relation_bwd_small_ok(dtype, N, Mpad):
        kt = qt = dyt = None       # the small-N backward kernel transposes K / Q / dY on the fly from LDS: no transposed copies
    elif sink is not None:         # persistent zero-padded buffers: the transposes below write [:, :, :rows], the pad columns stay zero
        Npad = ops.pad_to(N, 32)
        # (keyed on the row counts as well as the shapes: a later call with the same padded shape but fewer rows must not see the
        #  previous call's rows in [rows_new, rows_old) -- ADVICE r05)
        kt = sink.scratch('kt_%d' % M, (B, d, Mpad), dtype)
        qt = sink.scratch('qt_%d' % N, (B, d, Npad), dtype)
        dyt = sink.scratch('dyt_%d' % N, (B, d, Npad), dtype)
        ops.transpose_2d(k, out=kt); ops.transpose_2d(q, out=qt); ops.transpose_2d(dY, out=dyt)
    else:
        kt = torch.zeros((B, d, Mpad), device=f.device, dtype=dtype)
        ops.transpose_2d(k, out=kt)
        qt = ops.transpose_2d(q, pad_cols_to=32)
        dyt = ops.transpose_2d(dY, pad_cols_to=32)
    packed = None
    if sink is not None and dtype == torch.bfloat16 and ops.relation_bwd_small_ok(dtype, N, Mpad):
        # small-N form (the learn-NMS head's module): the backward kernel writes (dQ | dK | dVW) as bf16 straight into the projection
        # backward's operand; its key blocks past row M are never written and stay zero in the persistent buffer
        packed = sink.scratch('a3_%d_%d' % (index, M), (B, N, 3 * d), dtype)       # per module and key count (rows >= M of the key blocks must be zero): the trainer's QUEUED weight-gradient product reads it after this call returns
    dq, dk, dvw, prob, dlog = ops.relation_attention_bwd(q, k, kt, vw, bias, dY, y, mod.bout, qt, dyt, M, key_count=key_count, packed_out=packed)
    if sink is not None and dtype == torch.bfloat16:
        # ---- gradients straight into the trainer's buffers (GradSink): one pack kernel, ONE projection-backward GEMM with the residual
        # gradient in its epilogue, ONE queued weight-gradient product for [Wq; Wk; Wout], two column-sum kernels for the biases
        if sink.defer is not None:
            sink.defer(lambda: ops.geometry_bias_bwd(bx, bias, dlog, M, fast=True, out=(sink.dwp, sink.dbp)), (bx, bias, dlog))
        else:
            ops.geometry_bias_bwd(bx, bias, dlog, M, fast=True, out=(sink.dwp, sink.dbp))
        a3 = packed if packed is not None else ops.relation_bwd_pack(dq, dk, dvw)   # [B, N, 3 d] bf16 = (dQ | dK | dVW), key blocks zero past M
        a3_2d = a3.view(B * N, 3 * d)
        d_f = ops.gemm_nt(a3_2d, sink.wcat_t, resid=None if sink.resid is None else sink.resid.reshape(B * N, Fd)).reshape(B, N, Fd)
        sink.wgrad(a3_2d, f.reshape(B * N, Fd))
        from . import train_ops as _T
        _T.colsum_add(a3_2d[:, :2 * d], sink.b_qk)
        if sink.b_out is not None:
            _T.colsum_add(dY.reshape(B * N, d), sink.b_out)
        return {'d_roi_feat': d_f[0] if squeeze else d_f}
    dwp, dbp = ops.geometry_bias_bwd(bx, bias, dlog, M, fast=(dtype == torch.bfloat16))
    # ---- projections: Q|K = F [Wq;Wk]^T + b,  VW = F_K Wout^T
    dqk = torch.zeros((B, N, 2 * d), device=f.device, dtype=dtype)
    dqk[:, :, :d] = dq
    dqk[:, :M, d:] = dk
    dvw_t = dvw.to(dtype)
    wqk_t = getattr(mod, 'wqk_t', None)                                  # [Fd, 2d]: the trainer's per-step copies when present
    wout_t = getattr(mod, 'wout_t', None)                                # [Fd, d]
    wqk_t = ops.transpose_2d(mod.wqk) if wqk_t is None else wqk_t
    wout_t = ops.transpose_2d(mod.wout) if wout_t is None else wout_t
    d_f = ops.gemm_nt(dqk.reshape(B * N, 2 * d), wqk_t, out_dtype=torch.float32).reshape(B, N, Fd)
    d_fk = ops.gemm_nt(dvw_t.reshape(B * M, d), wout_t, out_dtype=torch.float32).reshape(B, M, Fd)
    d_f[:, :M] += d_fk
    # weight gradients: dW[out, in] = sum_rows dOut[row, out] X[row, in]
    fk = f[:, :M, :].contiguous().reshape(B * M, Fd)
    if dtype == torch.bfloat16:          # straight from the row-major operands (csrc/wgrad.hip), no transposed copies
        d_wqk = ops.wgrad_tn(dqk.reshape(B * N, 2 * d), f.reshape(B * N, Fd))         # [2d, Fd]
        d_wout = ops.wgrad_tn(dvw_t.reshape(B * M, d), fk)                            # [d, Fd]
    else:
        f_t = ops.transpose_2d(f.reshape(B * N, Fd), pad_cols_to=kpad)       # [Fd, pad(B N)]
        dqk_t = ops.transpose_2d(dqk.reshape(B * N, 2 * d), pad_cols_to=kpad)
        d_wqk = ops.gemm_nt(dqk_t, f_t, out_dtype=torch.float32)             # [2d, Fd]
        fk_t = ops.transpose_2d(fk, pad_cols_to=kpad)
        dvw_tt = ops.transpose_2d(dvw_t.reshape(B * M, d), pad_cols_to=kpad)
        d_wout = ops.gemm_nt(dvw_tt, fk_t, out_dtype=torch.float32)          # [d, Fd]
    i = index
    grads = {
        'd_roi_feat': d_f[0] if squeeze else d_f,
        'query_%d_weight' % i: d_wqk[:d], 'key_%d_weight' % i: d_wqk[d:],
        'query_%d_bias' % i: dq.sum((0, 1)), 'key_%d_bias' % i: dk.sum((0, 1)),
        'linear_out_%d_weight' % i: d_wout.reshape(d, Fd, 1, 1),
        'linear_out_%d_bias' % i: dY.float().sum((0, 1)),
        'pair_pos_fc1_%d_weight' % i: dwp, 'pair_pos_fc1_%d_bias' % i: dbp,
    }
    return grads
