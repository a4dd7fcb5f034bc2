"""Torch-tensor front ends of the C-ABI kernels (device memory + streams are torch's;
all arithmetic happens in librelnet_hip.so).  Every function requires CUDA(HIP) tensors
and raises otherwise -- there is no CPU or eager fallback.
"""
import math
import os

import torch

from . import lib as _lib

F32, BF16 = _lib.F32, _lib.BF16


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("unsupported dtype %s (float32 or bfloat16)" % t.dtype)


def _chk(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise _lib.RelnetError("relnet ops need GPU tensors (HIP kernels only; no CPU fallback)")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def pad32(m):
    return (m + 31) // 32 * 32


_GEMM_WS = {}
_GEMM_WS_BYTES = 16384 + (16 << 20)        # counters + partial tiles: the default rule splits <= 128 tiles 4 ways = 8.4 MB at most; a launch that does not fit runs unsplit


def gemm_workspace():
    """Names the split-K work area of the CURRENT stream to the library (relnet_gemm_set_workspace, tile configuration 23) before a
    GEMM / convolution call: one 16 MB area per launching stream, made on the stream's first GEMM and kept for the life of the
    process -- launches on one stream are ordered, which is what the area needs.  A stream that is being captured into a hipGraph
    gets an area per CAPTURE (relnet_stream_capture_id): it comes from that capture's memory pool, the zeroing of its 16 KB of
    counters is a node of the graph, and the reference held here keeps the block from being handed out again.  The library only
    splits launches of at most one 64 x 64 workgroup per CU with k-loops of >= 128 slabs (the one-image step)."""
    st = torch.cuda.current_stream().cuda_stream
    cap = _lib.load().relnet_stream_capture_id(st) if torch.cuda.is_current_stream_capturing() else 0
    key = (torch.cuda.current_device(), st, cap)
    ws = _GEMM_WS.get(key)
    if ws is None:
        ws = torch.empty(_GEMM_WS_BYTES, device='cuda:%d' % key[0], dtype=torch.uint8)
        ws[:16384].zero_()
        _GEMM_WS[key] = ws
    _lib.call('relnet_gemm_set_workspace', ws.data_ptr(), ws.numel())
    return ws


def gemm_nt(a, w, bias=None, bias_mode=1, resid=None, relu=False, out=None, out_dtype=None,
            n_cols=None):
    """out = a @ w^T (+bias) (+resid) (relu).  a [M,K] or [batch,M,K] (a 2-D `a` with a
    3-D `w` is broadcast over the batch and vice versa); w [N,K]; rows K-contiguous.
    `out` may be a wider pre-allocated buffer ([.., M, ldc]); n_cols limits N."""
    _chk(a, w, bias, resid, out)
    batch = 1
    if a.dim() == 3:
        batch = a.shape[0]
    if w.dim() == 3:
        batch = max(batch, w.shape[0])
    M, K = a.shape[-2], a.shape[-1]
    N = w.shape[-2] if n_cols is None else n_cols
    assert w.shape[-1] == K, (a.shape, w.shape)
    assert a.stride(-1) == 1 and w.stride(-1) == 1
    sa = a.stride(0) if a.dim() == 3 else 0
    sw = w.stride(0) if w.dim() == 3 else 0
    odt = out_dtype or (out.dtype if out is not None else a.dtype)
    if out is None:
        shape = (batch, M, N) if batch > 1 or a.dim() == 3 or w.dim() == 3 else (M, N)
        out = torch.empty(shape, device=a.device, dtype=odt)
    assert out.stride(-1) == 1
    sc = out.stride(0) if out.dim() == 3 else 0
    if resid is not None:
        assert resid.dtype == out.dtype and resid.stride() == out.stride()
    gemm_workspace()
    _lib.call('relnet_gemm_nt', a.data_ptr(), a.stride(-2), sa, w.data_ptr(), w.stride(-2), sw,
              out.data_ptr(), out.stride(-2), sc, _ptr(bias), bias_mode if bias is not None else 0,
              _ptr(resid), int(relu), M, N, K, batch, _dt(a), _dt(out), _stream(),
              tag='M%d_N%d_K%d_b%d' % (M, N, K, batch))
    return out


_ASM_SELFCHECK = None


def asm_selfcheck(force=False):
    """Runtime twin of build.py's ISA guard for the hand-scheduled k-loop (gemm.hip tiles 18 / 19: accumulators in literal AGPRs
    across asm statements -- correct only as long as the compiler leaves those registers alone): once per process, on the first
    Backbone / Trainer built on a GPU, one 768 x 512 x 1024 product on tile 19 is compared with the compiler-scheduled ring tile 8.
    On a mismatch the asm tiles are switched off for the process (relnet_gemm_debug_asm(0): pick_tile then never chooses them)
    and the fact is announced on stderr.  -> True (match / not checked yet on this device) or False."""
    global _ASM_SELFCHECK
    if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
        return True
    dev = torch.cuda.current_device()
    if _ASM_SELFCHECK is None:
        _ASM_SELFCHECK = {}
    if dev in _ASM_SELFCHECK and not force:           # (cached per DEVICE: a process may drive several GPUs)
        return _ASM_SELFCHECK[dev]
    if os.environ.get('RELNET_DEBUG_KNOBS') == '1' and (os.environ.get('RELNET_GEMM_FORCE_TILE') or os.environ.get('RELNET_GEMM_ASM')):
        return True                         # an A/B run pinned the tile choice by hand
    lib = _lib.load()
    g = torch.Generator().manual_seed(19)
    a = torch.randn(768, 1024, generator=g).cuda().to(torch.bfloat16)
    w = torch.randn(512, 1024, generator=g).cuda().to(torch.bfloat16)
    prev = lib.relnet_gemm_get_forced_tile()       # a tile forced by the caller before the first model is built survives the check
    try:
        lib.relnet_gemm_force_tile(19); y19 = gemm_nt(a, w).float()
        lib.relnet_gemm_force_tile(8); y8 = gemm_nt(a, w).float()
    finally:
        lib.relnet_gemm_force_tile(prev)
    err, scale = float((y19 - y8).abs().max()), float(y8.abs().max())
    ok = _ASM_SELFCHECK[dev] = bool(err <= 1e-2 * scale) and bool(torch.isfinite(y19).all())
    if not ok:
        import sys
        lib.relnet_gemm_debug_asm(0)
        sys.stderr.write('relnet: hand-scheduled GEMM tile 19 disagrees with tile 8 (max |diff| %.3g of %.3g): asm tiles DISABLED for this '
                         'process (compiler-scheduled ring tile instead)\n' % (err, scale))
    return ok


def gemm_nt_mask(a, w, mask, resid=None, out=None):
    """out = (a @ w^T + resid) where mask > 0, else 0 (bf16; a [M,K], w [N,K], resid / mask / out [M,N] with one row stride):
    the data gradient through `relu(conv1x1(.) + shortcut)` in one launch (relnet_gemm_nt_mask)."""
    _chk(a, w, mask, resid, out)
    M, K = a.shape
    N = w.shape[0]
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and mask.dtype == torch.bfloat16
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1 and tuple(mask.shape) == (M, N) and mask.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    assert out.stride(1) == 1 and mask.stride(0) == out.stride(0)
    if resid is not None:
        assert resid.dtype == torch.bfloat16 and tuple(resid.shape) == (M, N) and resid.stride() == out.stride()
    _lib.call('relnet_gemm_nt_mask', a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0),
              _ptr(resid), mask.data_ptr(), M, N, K, _stream(), tag='M%d_N%d_K%d' % (M, N, K))
    return out


def embedding_divisors(feat_dim=64, wave_length=1000.0):
    """fp32 dim_mat of the reference graph (SYM_REL:32-35): wave_length ** ((8/feat_dim) k),
    evaluated in float32 like MXNet's arange / broadcast_power."""
    k = torch.arange(0, feat_dim // 8, dtype=torch.float32)
    return torch.pow(torch.tensor(float(wave_length), dtype=torch.float32),
                     torch.tensor(8.0 / feat_dim, dtype=torch.float32) * k)


def geometry_bias(boxes, wp_t, bp, M=None, divisors=None, debug=False, half=False, fast32=False, mfma32=False):
    """boxes [B,N,4|5] fp32 (xyxy, or batch_idx + xyxy); wp_t [64, nmod*16]; bp [nmod*16]
    -> bias [nmod, B, 16, N, Mpad] fp32 = log(max(relu(E Wp^T + bp), 1e-6)) (the oracle's arithmetic: correctly rounded
    sin / cos / log, float64 accumulation; fast32: float32 libm arithmetic; mfma32: float32 ln(.) from the matrix-core kernel,
    the values the bf16 forward saw -- the training backward's recompute), or with half=True fp16 log2(.) from the
    matrix-core kernel (bf16 throughput path).
    debug=True also returns (position_matrix [B,N,M,4], position_embedding [B,N,M,64])."""
    _chk(boxes, wp_t, bp)
    assert boxes.dtype == torch.float32 and boxes.is_contiguous()
    B, N, bs = boxes.shape
    off = 1 if bs == 5 else 0
    M = N if M is None else M
    Mpad = pad32(M)
    nmod = wp_t.shape[1] // 16
    assert wp_t.shape == (64, nmod * 16) and wp_t.is_contiguous() and wp_t.dtype == torch.float32
    div = (divisors if divisors is not None else embedding_divisors()).to('cpu', torch.float32).contiguous()
    bias = torch.empty((nmod, B, 16, N, Mpad), device=boxes.device, dtype=torch.float16 if half else torch.float32)
    pm = pe = None
    if debug:
        pm = torch.empty((B, N, M, 4), device=boxes.device, dtype=torch.float32)
        pe = torch.empty((B, N, M, 64), device=boxes.device, dtype=torch.float32)
    _lib.call('relnet_geometry_bias', boxes.data_ptr(), bs, off, wp_t.data_ptr(), bp.data_ptr(),
              div.data_ptr(), bias.data_ptr(), 1 if half else (2 if mfma32 else (-1 if fast32 else 0)), _ptr(pm), _ptr(pe), B, N, M, Mpad, 16, nmod, _stream())
    if debug:
        return bias, pm, pe
    return bias


def relation_attention(q, k, vwt, bias, bout=None, resid=None, M=None, want_out=True,
                       want_act=False, want_logits=False, heads=16, key_count=None):
    """q [B,N,>=H*64] (row stride free), k [B,>=M,..], vwt [B,H*64,Mpad] (zero padded),
    bias [B,H,N,Mpad] fp32 -> (out [B,N,H*64] | None, relu(resid+out) | None, logits | None).
    key_count [B] int32 (optional): image b has only key_count[b] <= M keys, the rest of its key rows are padding."""
    _chk(q, k, vwt, bias, bout, resid, key_count)
    assert key_count is None or (key_count.dtype == torch.int32 and key_count.is_contiguous() and key_count.numel() == q.shape[0])
    B, N = q.shape[0], q.shape[1]
    H = heads
    Mpad = vwt.shape[-1]
    M = k.shape[1] if M is None else M
    assert bias.shape == (B, H, N, Mpad) and bias.is_contiguous() and bias.dtype in (torch.float32, torch.float16)
    assert vwt.shape[1] == H * 64 and vwt.stride(-1) == 1 and q.stride(-1) == 1 and k.stride(-1) == 1
    dt = q.dtype
    out = torch.empty((B, N, H * 64), device=q.device, dtype=dt) if want_out else None
    act = torch.empty((B, N, H * 64), device=q.device, dtype=dt) if want_act else None
    logits = torch.empty((B, N, H, M), device=q.device, dtype=torch.float32) if want_logits else None
    rs = (resid.stride(1), resid.stride(0)) if resid is not None else (0, 0)
    _lib.call('relnet_relation_attention_kc',
              q.data_ptr(), q.stride(1), q.stride(0), k.data_ptr(), k.stride(1), k.stride(0),
              vwt.data_ptr(), vwt.stride(1), vwt.stride(0), bias.data_ptr(), int(bias.dtype == torch.float16), bias.stride(0),
              _ptr(bout), _ptr(resid), rs[0], rs[1],
              _ptr(out), H * 64, N * H * 64, _ptr(act), H * 64, N * H * 64, _ptr(logits),
              B, H, N, M, Mpad, 1.0 / math.sqrt(64.0), _dt(q), _dt(q), _ptr(key_count), _stream())
    return out, act, logits


FUSED_MAX_KEYS = 640          # key table + two bias tiles must fit the 160 KiB LDS (csrc/relation.hip)


def relation_attention_fused(q, k, vwt, boxes, wp, bp, bout=None, resid=None, M=None, want_out=True,
                             want_act=False, heads=16, divisors=None):
    """Geometry bias + attention of one relation module in one kernel (bf16): operands as `relation_attention`, with the
    bias tensor replaced by its inputs: boxes [B,N,4|5] fp32, wp [16,64] / bp [16] fp32 (pair_pos_fc1 of the module).
    -> (out | None, relu(resid + out) | None)."""
    _chk(q, k, vwt, boxes, wp, bp, bout, resid)
    B, N = q.shape[0], q.shape[1]
    H = heads
    Mpad = vwt.shape[-1]
    M = k.shape[1] if M is None else M
    assert q.dtype == torch.bfloat16 and H == 16 and M <= FUSED_MAX_KEYS
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.shape[:2] == (B, N)
    assert wp.shape == (16, 64) and wp.dtype == torch.float32 and wp.is_contiguous() and bp.dtype == torch.float32
    assert vwt.shape[1] == H * 64 and vwt.stride(-1) == 1 and q.stride(-1) == 1 and k.stride(-1) == 1
    bs = boxes.shape[2]
    div = (divisors if divisors is not None else embedding_divisors()).to('cpu', torch.float32).contiguous()
    out = torch.empty((B, N, H * 64), device=q.device, dtype=q.dtype) if want_out else None
    act = torch.empty((B, N, H * 64), device=q.device, dtype=q.dtype) if want_act else None
    rs = (resid.stride(1), resid.stride(0)) if resid is not None else (0, 0)
    _lib.call('relnet_relation_attention_fused',
              q.data_ptr(), q.stride(1), q.stride(0), k.data_ptr(), k.stride(1), k.stride(0),
              vwt.data_ptr(), vwt.stride(1), vwt.stride(0), boxes.data_ptr(), bs, 1 if bs == 5 else 0,
              wp.data_ptr(), bp.data_ptr(), div.data_ptr(), _ptr(bout), _ptr(resid), rs[0], rs[1],
              _ptr(out), H * 64, N * H * 64, _ptr(act), H * 64, N * H * 64,
              B, H, N, M, Mpad, 1.0 / math.sqrt(64.0), _stream())
    return out, act


# ---------------------------------------------------------------------------------------
# RPN proposal path
# ---------------------------------------------------------------------------------------
def _strides4(t):
    import ctypes
    return (ctypes.c_long * 4)(*[int(s) for s in t.stride()])


def proposal_decode(cls_prob, bbox_deltas, im_info, base_anchors, feat_stride=16, min_size=0,
                    im_hw=None, softmax_pairs=False):
    """cls_prob [B,2A,H,W], bbox_deltas [B,4A,H,W] fp32 (any strides), im_info [B,3],
    base_anchors float64 [A,4] -> boxes [B,n,4] fp32, scores [B,n] fp32 in (y,x,a) order over
    the cropped grid int(im_h/stride) x int(im_w/stride) (proposal.py:85)."""
    _chk(cls_prob, bbox_deltas, im_info, base_anchors)
    assert cls_prob.dtype == torch.float32 and bbox_deltas.dtype == torch.float32
    assert base_anchors.dtype == torch.float64 and im_info.dtype == torch.float32
    B = cls_prob.shape[0]
    A = base_anchors.shape[0]
    if im_hw is None:                      # reads im_info on the host (a device sync)
        info = im_info.detach().cpu()
        assert bool((info[:, :2] == info[0, :2]).all()), "one (h, w) per batch"
        im_hw = (info[0, 0].item(), info[0, 1].item())
    h, w = int(im_hw[0] / feat_stride), int(im_hw[1] / feat_stride)
    h, w = min(h, cls_prob.shape[2]), min(w, cls_prob.shape[3])
    n = h * w * A
    boxes = torch.empty((B, n, 4), device=cls_prob.device, dtype=torch.float32)
    scores = torch.empty((B, n), device=cls_prob.device, dtype=torch.float32)
    _lib.call('relnet_proposal_decode', cls_prob.data_ptr(), _strides4(cls_prob), bbox_deltas.data_ptr(),
              _strides4(bbox_deltas), im_info.data_ptr(), base_anchors.data_ptr(), boxes.data_ptr(),
              scores.data_ptr(), B, A, h, w, feat_stride, int(min_size), int(softmax_pairs), _stream())
    return boxes, scores


def topk_sort(scores, boxes, K):
    """Descending top-K: -> det [B,K,5] (x1,y1,x2,y2,score), index [B,K] int32, count [B]."""
    _chk(scores, boxes)
    B, n = scores.shape
    K = min(K, n)
    det = torch.empty((B, K, 5), device=scores.device, dtype=torch.float32)
    index = torch.empty((B, K), device=scores.device, dtype=torch.int32)
    count = torch.empty((B,), device=scores.device, dtype=torch.int32)
    _lib.call('relnet_topk_sort', scores.data_ptr(), boxes.data_ptr(), det.data_ptr(), index.data_ptr(),
              count.data_ptr(), B, n, K, _stream())
    return det, index, count


def nms_sorted(det, thresh, post=0, counts=None, max_keep=None, want_keep=False, batch_index_base=0):
    """det [B,n,5] sorted by score.  Returns dict(rois [B,post,5], scores [B,post],
    keep [B,max_keep] int32, num_keep [B])."""
    _chk(det, counts)
    B, n, _ = det.shape
    cb = (n + 63) // 64
    mask = torch.empty((B, n, cb), device=det.device, dtype=torch.int64)
    _lib.call('relnet_nms_mask', det.data_ptr(), _ptr(counts), mask.data_ptr(), B, n, n, float(thresh), _stream())
    max_keep = max_keep or (post if post > 0 else n)
    rois = torch.zeros((B, post, 5), device=det.device, dtype=torch.float32) if post > 0 else None
    rscores = torch.zeros((B, post), device=det.device, dtype=torch.float32) if post > 0 else None
    keep = torch.full((B, max_keep), -1, device=det.device, dtype=torch.int32) if want_keep else None
    num = torch.empty((B,), device=det.device, dtype=torch.int32)
    _lib.call('relnet_nms_scan', mask.data_ptr(), det.data_ptr(), _ptr(counts), _ptr(rois), _ptr(rscores),
              _ptr(keep), num.data_ptr(), B, n, n, post, max_keep, batch_index_base, _stream())
    return dict(rois=rois, scores=rscores, keep=keep, num_keep=num)


def nms_greedy(det, thresh, post, counts=None, want_keep=False, batch_index_base=0):
    """Fused greedy NMS (no bitmask) keeping the first `post` boxes: same outputs as
    nms_sorted(det, thresh, post=post)."""
    _chk(det, counts)
    B, n, _ = det.shape
    rois = torch.zeros((B, post, 5), device=det.device, dtype=torch.float32)
    rscores = torch.zeros((B, post), device=det.device, dtype=torch.float32)
    keep = torch.full((B, post), -1, device=det.device, dtype=torch.int32) if want_keep else None
    num = torch.empty((B,), device=det.device, dtype=torch.int32)
    _lib.call('relnet_nms_greedy', det.data_ptr(), _ptr(counts), rois.data_ptr(), rscores.data_ptr(), _ptr(keep),
              num.data_ptr(), B, n, n, post, float(thresh), batch_index_base, _stream())
    return dict(rois=rois, scores=rscores, keep=keep, num_keep=num)


def roi_pool(data, rois, pooled=(7, 7), spatial_scale=0.0625, channels_last_out=False,
             want_argmax=False, batch_index_base=0):
    """data: logical [B,C,H,W] tensor of any strides (NCHW or channels_last memory format);
    rois [R,5].  Output logical [R,C,PH,PW]; memory order (R,PH,PW,C) when
    channels_last_out (then `.permute(0,2,3,1)` is contiguous)."""
    _chk(data, rois)
    assert rois.dtype == torch.float32 and rois.is_contiguous()
    B, Cc, H, W = data.shape
    R = rois.shape[0]
    PH, PW = pooled
    if channels_last_out:
        out = torch.empty((R, PH, PW, Cc), device=data.device, dtype=data.dtype).permute(0, 3, 1, 2)
    else:
        out = torch.empty((R, Cc, PH, PW), device=data.device, dtype=data.dtype)
    arg = torch.empty_strided(out.shape, out.stride(), device=data.device, dtype=torch.int32) if want_argmax else None
    _lib.call('relnet_roi_pool_fwd', data.data_ptr(), _strides4(data), rois.data_ptr(), out.data_ptr(),
              _strides4(out), _ptr(arg), R, Cc, H, W, PH, PW, float(spatial_scale), batch_index_base,
              _dt(data), _stream())
    return (out, arg) if want_argmax else out


def roi_pool_bwd(grad_out, argmax, rois, in_shape, batch_index_base=0, channels_last=False):
    """Adjoint of roi_pool: grad_out / argmax logical [R,C,PH,PW] with IDENTICAL strides (as returned by
    roi_pool(want_argmax=True)); in_shape (B,C,H,W) -> fp32 gradient of the feature map, logical [B,C,H,W]; memory NCHW, or
    NHWC with channels_last (coalesced atomics; `.permute(0, 2, 3, 1)` of the result is then contiguous)."""
    _chk(grad_out, argmax, rois)
    assert argmax.dtype == torch.int32 and tuple(grad_out.stride()) == tuple(argmax.stride()) and grad_out.shape == argmax.shape
    B, Cc, H, W = in_shape
    R, _, PH, PW = grad_out.shape
    if channels_last:
        gin = torch.zeros((B, H, W, Cc), device=grad_out.device, dtype=torch.float32).permute(0, 3, 1, 2)
        # (relnet_roi_pool_bwd_cl: from 8 images x 256 channels up, one workgroup per (image, 8 channels) accumulates its slab in LDS -- no global atomics)
        _lib.call('relnet_roi_pool_bwd_cl', grad_out.data_ptr(), argmax.data_ptr(), _strides4(grad_out), rois.data_ptr(), gin.data_ptr(),
                  B, H, W, R, Cc, PH, PW, batch_index_base, _dt(grad_out), _stream())
        return gin
    else:
        gin = torch.zeros((B, Cc, H, W), device=grad_out.device, dtype=torch.float32)
    _lib.call('relnet_roi_pool_bwd_ex', grad_out.data_ptr(), argmax.data_ptr(), _strides4(grad_out), rois.data_ptr(),
              gin.data_ptr(), gin.stride(0), gin.stride(1), gin.stride(3), R, Cc, W, PH, PW, batch_index_base, _dt(grad_out), _stream())
    return gin


def roi_align(data, rois, pooled=(7, 7), spatial_scale=0.0625, sampling_ratio=2, aligned=False, channels_last_out=False,
              batch_index_base=0):
    """ROIAlign (relnet_roi_align_fwd; Mask R-CNN section 3 / mx.contrib.sym.ROIAlign(data, rois, pooled_size, spatial_scale, sample_ratio)).
    data: logical [B,C,H,W] of any strides, fp32 / bf16; rois [R,5] fp32.  Output logical [R,C,PH,PW] (memory (R,PH,PW,C) when
    channels_last_out).  The reference graphs use ROIPooling (`roi_pool`); this is the operator north_star names next to it."""
    _chk(data, rois)
    assert rois.dtype == torch.float32 and rois.is_contiguous()
    B, Cc, H, W = data.shape
    R = rois.shape[0]
    PH, PW = pooled
    if channels_last_out:
        out = torch.empty((R, PH, PW, Cc), device=data.device, dtype=data.dtype).permute(0, 3, 1, 2)
    else:
        out = torch.empty((R, Cc, PH, PW), device=data.device, dtype=data.dtype)
    _lib.call('relnet_roi_align_fwd', data.data_ptr(), _strides4(data), rois.data_ptr(), out.data_ptr(), _strides4(out), R, Cc, H, W,
              PH, PW, float(spatial_scale), int(sampling_ratio), int(bool(aligned)), batch_index_base, _dt(data), _stream())
    return out


def roi_align_bwd(grad_out, rois, in_shape, spatial_scale=0.0625, sampling_ratio=2, aligned=False, batch_index_base=0,
                  channels_last=False):
    """Adjoint of roi_align: grad_out logical [R,C,PH,PW] (any strides) -> fp32 gradient of the feature map, logical [B,C,H,W]
    (memory NHWC with channels_last: coalesced atomics)."""
    _chk(grad_out, rois)
    B, Cc, H, W = in_shape
    R, _, PH, PW = grad_out.shape
    if channels_last:
        gin = torch.zeros((B, H, W, Cc), device=grad_out.device, dtype=torch.float32).permute(0, 3, 1, 2)
    else:
        gin = torch.zeros((B, Cc, H, W), device=grad_out.device, dtype=torch.float32)
    _lib.call('relnet_roi_align_bwd', grad_out.data_ptr(), _strides4(grad_out), rois.data_ptr(), gin.data_ptr(), _strides4(gin), R, Cc,
              H, W, PH, PW, float(spatial_scale), int(sampling_ratio), int(bool(aligned)), batch_index_base, _dt(grad_out), _stream())
    return gin


# ---------------------------------------------------------------------------------------
# detection post-processing
# ---------------------------------------------------------------------------------------
def detect_head(cls_score, bbox_pred, rois, im_info, rois_per_image, delta_off=4, n_valid=None):
    """cls_score [R,C] fp32 logits, bbox_pred [R,4*num_reg] fp32, rois [R,5], im_info [B,3]
    -> cls_prob [R,C] fp32, boxes [R,4] float64 (decoded class-agnostic fg box / im scale).
    n_valid [B] int32 (optional): rows past n_valid[b] of image b are padding -> all-zero probabilities and boxes."""
    _chk(cls_score, bbox_pred, rois, im_info, n_valid)
    R, Cn = cls_score.shape
    assert cls_score.dtype == torch.float32 and bbox_pred.dtype == torch.float32
    assert cls_score.stride(1) == 1 and bbox_pred.stride(1) == 1 and rois.is_contiguous()
    prob = torch.empty((R, Cn), device=cls_score.device, dtype=torch.float32)
    boxes = torch.empty((R, 4), device=cls_score.device, dtype=torch.float64)
    _lib.call('relnet_detect_head_ex', cls_score.data_ptr(), cls_score.stride(0), bbox_pred.data_ptr(),
              bbox_pred.stride(0), rois.data_ptr(), im_info.data_ptr(), prob.data_ptr(), boxes.data_ptr(),
              R, Cn, rois_per_image, delta_off, _ptr(n_valid), _stream())
    return prob, boxes


def class_nms(cls_prob, boxes, score_thresh=1e-3, nms_param=0.6, soft=True, max_picks=0, scores64=None,
              want_index=False, top_k=0):
    """cls_prob [B,N,C] fp32, boxes [B,N,4] float64 -> dets [B,C-1,N,5] float64 (pick order),
    counts [B,C-1] int32.  max_picks > 0 truncates every class list after that many picks
    (exactly the rows that can survive the image-level max_per_image cut).
    scores64 [B,N] float64 (one class; cls_prob None) is the `dets[:, 4]` form of lib/nms/nms.py;
    want_index also returns pick_index [B,C-1,N] int32 (roi index of every pick).
    top_k > 0 (the detector passes max_per_image): a class list additionally stops as soon as its next pick cannot be among the
    top_k scores of its image (relnet_class_nms_topk) -- a prefix of the full lists that contains every pick image_topk keeps."""
    _chk(cls_prob, boxes, scores64)
    if scores64 is not None:
        assert cls_prob is None and scores64.dtype == torch.float64 and scores64.is_contiguous()
        B, N = scores64.shape
        Cn = 2
    else:
        B, N, Cn = cls_prob.shape
        assert cls_prob.is_contiguous() and cls_prob.dtype == torch.float32
    assert boxes.is_contiguous() and boxes.dtype == torch.float64 and boxes.shape == (B, N, 4)
    dets = torch.zeros((B, Cn - 1, N, 5), device=boxes.device, dtype=torch.float64)
    counts = torch.empty((B, Cn - 1), device=boxes.device, dtype=torch.int32)
    index = torch.full((B, Cn - 1, N), -1, device=boxes.device, dtype=torch.int32) if want_index else None
    if top_k > 0 and scores64 is None and not want_index and N <= 1024:
        hist = torch.zeros((B, _lib.load().relnet_class_nms_hist_bins()), device=boxes.device, dtype=torch.int32)
        _lib.call('relnet_class_nms_topk', cls_prob.data_ptr(), boxes.data_ptr(), dets.data_ptr(), counts.data_ptr(), hist.data_ptr(),
                  B, N, Cn, float(score_thresh), float(nms_param), int(soft), int(max_picks), int(top_k), _stream())
        return dets, counts
    _lib.call('relnet_class_nms_ex', _ptr(cls_prob), _ptr(scores64), boxes.data_ptr(), dets.data_ptr(), counts.data_ptr(),
              _ptr(index), B, N, Cn, float(score_thresh), float(nms_param), int(soft), int(max_picks), _stream())
    return (dets, counts, index) if want_index else (dets, counts)


def bbox_overlaps(boxes, query_boxes):
    """float64 IoU matrix [N,K] of device tensors boxes [N,4], query_boxes [K,4] (lib/bbox/bbox.pyx:15-55)."""
    _chk(boxes, query_boxes)
    assert boxes.dtype == torch.float64 and query_boxes.dtype == torch.float64
    boxes, query_boxes = boxes.contiguous(), query_boxes.contiguous()
    N, K = boxes.shape[0], query_boxes.shape[0]
    assert boxes.shape == (N, 4) and query_boxes.shape == (K, 4)
    out = torch.zeros((N, K), device=boxes.device, dtype=torch.float64)
    _lib.call('relnet_bbox_overlaps', boxes.data_ptr(), query_boxes.data_ptr(), out.data_ptr(), N, K, _stream())
    return out


def image_topk(dets, counts, max_per_image=100, max_out=None):
    """-> out [B,max_out,6] (class, score, x1,y1,x2,y2), out_count [B], thresh [B] float64, total [B]."""
    _chk(dets, counts)
    B, NC, N, _ = dets.shape
    max_out = max_out or (max_per_image + 28)
    thresh = torch.empty((B,), device=dets.device, dtype=torch.float64)
    total = torch.empty((B,), device=dets.device, dtype=torch.int32)
    out = torch.zeros((B, max_out, 6), device=dets.device, dtype=torch.float32)
    out_count = torch.empty((B,), device=dets.device, dtype=torch.int32)
    _lib.call('relnet_image_topk', dets.data_ptr(), counts.data_ptr(), thresh.data_ptr(), total.data_ptr(),
              out.data_ptr(), out_count.data_ptr(), B, NC, N, max_per_image, max_out, _stream())
    return out, out_count, thresh, total


# ---------------------------------------------------------------------------------------
# NHWC convolution (implicit GEMM on the bf16 MFMA kernel)
# ---------------------------------------------------------------------------------------
def pack_conv_weight(w_oihw, dtype=torch.bfloat16, device='cuda'):
    """[Cout, Cin, R, S] -> [Cout, R*S*Cin] (k = (r*S+s)*Cin + ic)."""
    return w_oihw.permute(0, 2, 3, 1).reshape(w_oihw.shape[0], -1).to(device=device, dtype=dtype).contiguous()


def pack_w_frag(w_packed, panel_only=True):
    """[Cout, K] bf16 -> the same weights in MFMA fragment order: block (n / 32, k / 16) = 64 lanes x 16 bytes, lane l <-
    W[32 nb + (l & 31)][16 kb + 8 (l >> 5) ..].  panel_only: None unless the shape qualifies for the row-panel 1x1 kernel
    (K in {64,128,256,512}, Cout % 256 == 0); otherwise any Cout % 32 == 0, K % 16 == 0."""
    _chk(w_packed)
    N, K = w_packed.shape
    if panel_only and (w_packed.dtype != torch.bfloat16 or K not in (64, 128, 256, 512) or N % 256):
        return None
    assert w_packed.dtype == torch.bfloat16 and N % 32 == 0 and K % 16 == 0 and w_packed.stride(1) == 1
    out = torch.empty_like(w_packed)
    _lib.call('relnet_pack_w_frag', w_packed.data_ptr(), w_packed.stride(0), out.data_ptr(), N, K, _stream())
    return out


HALO3_CHANNELS = (64,)            # widths with a halo-resident 3x3 kernel (res2)


def conv3x3_halo(x, w_frag, bias, relu=True):
    """3x3 / stride 1 / pad 1 convolution + bias (+ ReLU) of a dense NHWC bf16 tensor with C in HALO3_CHANNELS channels in and out, input
    tile + halo resident in LDS (csrc/bottleneck.hip).  w_frag = pack_w_frag(packed weight [C, 9 C], panel_only=False)."""
    _chk(x, w_frag, bias)
    B, H, W, C = x.shape
    assert C in HALO3_CHANNELS and x.is_contiguous() and x.dtype == torch.bfloat16 and bias.dtype == torch.float32
    assert w_frag.numel() == 9 * C * C
    out = torch.empty_like(x)
    _lib.call('relnet_conv3x3_c%d' % C, x.data_ptr(), w_frag.data_ptr(), bias.data_ptr(), int(relu), out.data_ptr(), B, H, W, _stream())
    return out


def conv3x3_c64(x, w_frag, bias, relu=True):
    return conv3x3_halo(x, w_frag, bias, relu)


CHAIN_MIDS = (64, 128, 256)      # bottleneck widths relnet_bottleneck_chain is built for (res2: weights LDS-resident; res3 / res4: streamed per pass)
CHAIN_EXPAND_MIDS = (64, 128, 256, 512)      # ... and for its expand + shortcut + ReLU form without the second product (res4, res5 too)


CHAIN_MIN_PIXELS = {64: 16384, 128: 65536, 256: 8192, 512: 49152, 'streamed': 98304}     # tests lower these to run the chain kernels on small maps


def chain_worthwhile(pixels, mid):
    """The chain kernels are persistent with one workgroup per CU; the streamed forms (mid >= 128) walk lock-step sets of
    8 x 32 pixels, so they need >= ~1.5 sets per CU (256 CUs) to beat the tiled convolution kernels (measured: B = 1 / B = 8
    steps are slower with them on the small late-stage maps).  mid = 256 (res4: expand + next reduce in one role-specialised launch,
    sets of four tiles) replaces two launches and wins from 4 images of 600 x 1000 up (r04, same box, ms per step with / without:
    2 images 3.17 / 2.87, 4 images 4.17 / 4.52, 8: 5.13 / 5.44, 27: 10.72 / 11.53, 40: 15.92 / 17.49).  res3 (mid 128) from ~7 images
    (8 images: +0.7 %), res5's expand (mid 512) from ~21 (27 images: 11.00 -> 10.62 ms; at 8 images it loses 1.4 %)."""
    # upper bound: the chain kernels address x / x_next with 32-bit byte offsets (4 mid channels x 2 bytes per pixel must stay below
    # 4 GiB: ~223 images of 600 x 1000 in res4); beyond it the trunk falls back to the tiled convolution kernels instead of raising
    return pixels >= CHAIN_MIN_PIXELS[mid if mid in CHAIN_MIN_PIXELS else 'streamed'] and pixels * 8 * mid < (1 << 32)


def pack_chain_w1(w_packed):
    """Reduce weights [mid, 4 mid] bf16 of the NEXT block -> fragment order of bottleneck_chain's second product: block
    (row tile rt, k-step ks) = 64 lanes x 8 values, lane (l31, half) slot t <- W[32 rt + l31][16 ks + 8 (t >> 2) + 4 half + (t & 3)]
    (the contraction index in the order the accumulator registers of the first product hold it).  Pure indexing (a weight
    re-ordering done once at load time): runs on whatever device holds `w_packed`."""
    N, K = w_packed.shape
    assert w_packed.dtype == torch.bfloat16 and N % 32 == 0 and K % 16 == 0
    dev = w_packed.device
    rt = torch.arange(N // 32, device=dev).view(-1, 1, 1, 1)
    ks = torch.arange(K // 16, device=dev).view(1, -1, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 1, -1, 1)
    t = torch.arange(8, device=dev).view(1, 1, 1, -1)
    row = rt * 32 + (lane & 31)
    col = ks * 16 + 8 * (t >> 2) + 4 * (lane >> 5) + (t & 3)
    return w_packed[row, col].contiguous()


def bottleneck_chain(mid2, x, w3_frag, w1_frag, b3, b1, inplace=False):
    """x_next = relu(conv1x1(mid2; W3, b3) + x); mid1_next = relu(conv1x1(x_next; W1n, b1n)) in one kernel (NHWC bf16).
    mid2 [.., mid], x [.., 4 mid] dense; w3_frag = pack_w_frag(W3), w1_frag = pack_chain_w1(W1n).  w1_frag = b1 = None (last unit
    of a stage): only x_next is produced.  inplace: x_next is written over x (every wavefront reads a 32-pixel x 64-channel
    slice of x before it stores the same slice of x_next, and no other wavefront touches those pixels).
    -> (x_next, mid1_next | None)"""
    _chk(mid2, x, w3_frag, w1_frag, b3, b1)
    mid = mid2.shape[-1]
    assert mid2.is_contiguous() and x.is_contiguous() and x.shape[-1] == 4 * mid and x.shape[:-1] == mid2.shape[:-1]
    assert mid2.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and b3.dtype == torch.float32
    assert (w1_frag is None) == (b1 is None) and (b1 is None or b1.dtype == torch.float32)
    assert mid in (CHAIN_MIDS if w1_frag is not None else CHAIN_EXPAND_MIDS)
    xn = x if inplace else torch.empty_like(x)
    m1 = torch.empty_like(mid2) if w1_frag is not None else None
    _lib.call('relnet_bottleneck_chain', mid2.data_ptr(), x.data_ptr(), w3_frag.data_ptr(), _ptr(w1_frag), b3.data_ptr(),
              _ptr(b1), xn.data_ptr(), _ptr(m1), mid2.numel() // mid, mid, _stream())
    return xn, m1


def bottleneck_chain_proj(mid2, x_in, w3_frag, wp_frag, w1_frag, b3p, b1):
    """First unit of a stage whose shortcut is a stride-1 1x1 projection (res2a): x_next = relu(conv1x1(mid2; W3) + conv1x1(x_in; Wp)
    + b3p), b3p = b3 + bp, and (w1_frag given) mid1_next = relu(conv1x1(x_next; W1n, b1n)) in one kernel -- the projection's
    4 mid-channel output is never written or read back.  mid2, x_in [.., 64] dense bf16.  -> (x_next [.., 256], mid1_next | None)"""
    _chk(mid2, x_in, w3_frag, wp_frag, w1_frag, b3p, b1)
    mid = mid2.shape[-1]
    assert mid == 64 and x_in.shape == mid2.shape and mid2.is_contiguous() and x_in.is_contiguous()
    assert mid2.dtype == torch.bfloat16 and x_in.dtype == torch.bfloat16 and b3p.dtype == torch.float32
    assert (w1_frag is None) == (b1 is None)
    xn = torch.empty(mid2.shape[:-1] + (4 * mid,), device=mid2.device, dtype=torch.bfloat16)
    m1 = torch.empty_like(mid2) if w1_frag is not None else None
    _lib.call('relnet_bottleneck_chain_proj', mid2.data_ptr(), x_in.data_ptr(), w3_frag.data_ptr(), wp_frag.data_ptr(), _ptr(w1_frag),
              b3p.data_ptr(), _ptr(b1), xn.data_ptr(), _ptr(m1), mid2.numel() // mid, mid, _stream())
    return xn, m1


def conv2d_nhwc(x, w_packed, bias, ksize=1, stride=1, pad=0, dil=1, relu=False, resid=None, out=None,
                out_dtype=None, w_frag=None):
    """x [B,H,W,Cin] bf16 (last dim contiguous; pixel/image strides free), w_packed
    [Cout, k*k*Cin], bias fp32 [Cout] -> [B,Hout,Wout,Cout]; optional fused residual + ReLU."""
    _chk(x, w_packed, bias, resid, out)
    if x.dtype == torch.float32:        # float32 parity path: the exact-fp32 MFMA kernel in convolution mode
        return conv2d_nhwc_f32(x, w_packed, bias, ksize, stride, pad, dil, relu, resid, out)
    assert x.dtype == torch.bfloat16 and w_packed.dtype == torch.bfloat16 and x.stride(3) == 1
    B, H, W, Cin = x.shape
    assert x.stride(1) == W * x.stride(2), "rows of an image must be dense in W"
    Cout = w_packed.shape[0]
    assert w_packed.shape[1] == ksize * ksize * Cin and w_packed.is_contiguous()
    Hout = (H + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    Wout = (W + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    odt = out_dtype or (out.dtype if out is not None else x.dtype)
    if out is None:
        out = torch.empty((B, Hout, Wout, Cout), device=x.device, dtype=odt)
    assert out.shape == (B, Hout, Wout, Cout) and out.stride(3) == 1
    assert out.stride(1) == Wout * out.stride(2) and out.stride(0) == Hout * out.stride(1)
    if resid is not None:
        assert resid.dtype == out.dtype and resid.stride() == out.stride()
    gemm_workspace()
    _lib.call('relnet_conv2d_nhwc_wf', x.data_ptr(), x.stride(2), x.stride(0), w_packed.data_ptr(), _ptr(w_frag), _ptr(bias),
              _ptr(resid), int(relu), out.data_ptr(), out.stride(2), B, H, W, Cin, Cout, ksize, ksize,
              stride, dil, pad, _dt(out), _stream(),
              tag='M%d_N%d_K%d_k%d' % (B * Hout * Wout, Cout, ksize * ksize * Cin, ksize))
    return out


def conv2d_nhwc_f32(x, w_packed, bias, ksize=1, stride=1, pad=0, dil=1, relu=False, resid=None, out=None):
    """float32 twin of conv2d_nhwc on the exact-fp32 MFMA kernel (relnet_conv2d_nhwc_f32): x [B,H,W,Cin] fp32 with Cin % 16 == 0,
    w_packed [Cout, k*k*Cin] fp32, bias fp32 | None -> [B,Hout,Wout,Cout] fp32.  relu: False / True / 2 (resid is a ReLU mask)."""
    _chk(x, w_packed, bias, resid, out)
    assert x.dtype == torch.float32 and w_packed.dtype == torch.float32 and x.stride(3) == 1
    B, H, W, Cin = x.shape
    assert x.stride(1) == W * x.stride(2), "rows of an image must be dense in W"
    Cout = w_packed.shape[0]
    assert w_packed.shape[1] == ksize * ksize * Cin and w_packed.is_contiguous()
    Hout = (H + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    Wout = (W + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty((B, Hout, Wout, Cout), device=x.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.shape == (B, Hout, Wout, Cout) and out.stride(3) == 1
    assert out.stride(1) == Wout * out.stride(2) and out.stride(0) == Hout * out.stride(1)
    if resid is not None:
        assert resid.dtype == torch.float32 and resid.stride() == out.stride()
    _lib.call('relnet_conv2d_nhwc_f32', x.data_ptr(), x.stride(2), x.stride(0), w_packed.data_ptr(), _ptr(bias), _ptr(resid), int(relu),
              out.data_ptr(), out.stride(2), B, H, W, Cin, Cout, ksize, ksize, stride, dil, pad, _stream(),
              tag='M%d_N%d_K%d_k%d_f32' % (B * Hout * Wout, Cout, ksize * ksize * Cin, ksize))
    return out


def maxpool_nhwc_f32(x, ksize=3, stride=2):
    """pool1 of the float32 path: ceil-mode max pooling without padding (pooling_convention='full'), x [B,H,W,C] fp32 dense."""
    _chk(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, H, W, Cc = x.shape
    Ho = -(-(H - ksize) // stride) + 1
    Wo = -(-(W - ksize) // stride) + 1
    if (Ho - 1) * stride >= H:
        Ho -= 1
    if (Wo - 1) * stride >= W:
        Wo -= 1
    out = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    _lib.call('relnet_maxpool_nhwc_f32', x.data_ptr(), out.data_ptr(), B, H, W, Cc, ksize, stride, _stream())
    return out


def stem_bias_relu_pool(x_nhwc, bias, ksize=3, stride=2):
    """x [B,H,W,C] bf16 contiguous (conv output without bias) -> relu(maxpool_ceil(x) + bias) [B,Ho,Wo,C]."""
    _chk(x_nhwc, bias)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.is_contiguous() and bias.dtype == torch.float32
    B, H, W, Cc = x_nhwc.shape
    Ho = -(-(H - ksize) // stride) + 1
    Wo = -(-(W - ksize) // stride) + 1
    if (Ho - 1) * stride >= H:
        Ho -= 1
    if (Wo - 1) * stride >= W:
        Wo -= 1
    out = torch.empty((B, Ho, Wo, Cc), device=x_nhwc.device, dtype=torch.bfloat16)
    _lib.call('relnet_stem_bias_relu_pool', x_nhwc.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, Cc,
              ksize, stride, _stream())
    return out


def stem_fused(data, w256, bias):
    """data [B,3,H,W] fp32/bf16 NCHW -> relu(pool1(relu(conv1 7x7/2 + bias))) as NHWC bf16 [B,Hp,Wp,64] in ONE kernel
    (csrc/stem.hip:stem_fused_kernel): bit-identical to stem_conv7(relu=True) + stem_bias_relu_pool(zero bias)."""
    _chk(data, w256, bias)
    data = data.contiguous()
    B, Cin, H, W = data.shape
    assert Cin == 3 and w256.shape == (64, 256) and w256.dtype == torch.bfloat16 and bias.dtype == torch.float32
    Hc, Wc = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = -(-(Hc - 3) // 2) + 1, -(-(Wc - 3) // 2) + 1
    if (Hp - 1) * 2 >= Hc:
        Hp -= 1
    if (Wp - 1) * 2 >= Wc:
        Wp -= 1
    out = torch.empty((B, Hp, Wp, 64), device=data.device, dtype=torch.bfloat16)
    _lib.call('relnet_stem_fused', data.data_ptr(), _dt(data), w256.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, _stream())
    return out


def pack_stem_weight(w_oihw, dtype=torch.bfloat16, device='cuda'):
    """[Cout, 3, 7, 7] -> [Cout, 256] with k = ty*32 + tx*4 + c (zeros for ty = 7, tx = 7, c = 3)."""
    co = w_oihw.shape[0]
    w = torch.zeros(co, 8, 8, 4, dtype=torch.float32)
    w[:, :7, :7, :3] = w_oihw.detach().cpu().float().permute(0, 2, 3, 1)
    return w.reshape(co, 256).to(device=device, dtype=dtype).contiguous()


def stem_conv7(data, w256, bias, relu=True):
    """data [B,3,H,W] fp32/bf16 NCHW contiguous -> relu(conv 7x7 / 2, pad 3) as NHWC bf16 [B,Ho,Wo,Cout]."""
    _chk(data, w256, bias)
    data = data.contiguous()
    B, Cin, H, W = data.shape
    assert Cin == 3
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = 2 * (Ho - 1) + 8, 2 * (Wo - 1) + 8
    Hp, Wp = max(Hp, H + 6), max(Wp + (Wp & 1), W + 6 + ((W + 6) & 1))
    packed = torch.empty((B, Hp, Wp, 4), device=data.device, dtype=torch.bfloat16)
    _lib.call('relnet_stem_pack_input', data.data_ptr(), packed.data_ptr(), B, H, W, Hp, Wp, 3, _dt(data), _stream())
    Cout = w256.shape[0]
    out = torch.empty((B, Ho, Wo, Cout), device=data.device, dtype=torch.bfloat16)
    _lib.call('relnet_stem_conv7', packed.data_ptr(), w256.data_ptr(), _ptr(bias), int(relu), out.data_ptr(), Cout,
              B, Hp, Wp, Ho, Wo, Cout, _dt(out), _stream())
    return out


# ---------------------------------------------------------------------------------------
# training-target operators
# ---------------------------------------------------------------------------------------
def _d4(v):
    import ctypes
    return (ctypes.c_double * 4)(*[float(x) for x in v])


def proposal_target(rois, gt_boxes, num_gt=None, num_reg=2, class_agnostic=True, bg_thresh_hi=0.5,
                    means=(0.0, 0.0, 0.0, 0.0), stds=(0.1, 0.1, 0.2, 0.2), weights=(1.0, 1.0, 1.0, 1.0), num_rois=None):
    """rois [B,N,5], gt_boxes [B,Gmax,5] (x1,y1,x2,y2,cls), num_gt [B] int32 (default: all Gmax valid)
    -> rois_out [B,N+Gmax,5], label [B,N+Gmax], bbox_target / bbox_weight [B,N+Gmax,4*num_reg]
    (BATCH_ROIS = -1 semantics; rows past N + num_gt[b] carry label -1 and zero weights).
    num_rois [B] int32 (optional): only the first num_rois[b] input rows are proposals; the padded rows come out with a
    zero box, label -1 and zero weights."""
    _chk(rois, gt_boxes, num_gt, num_rois)
    B, N, _ = rois.shape
    G = gt_boxes.shape[1]
    dev = rois.device
    if num_gt is None:
        num_gt = torch.full((B,), G, device=dev, dtype=torch.int32)
    R = N + G
    ro = torch.empty((B, R, 5), device=dev, dtype=torch.float32)
    lab = torch.empty((B, R), device=dev, dtype=torch.float32)
    bt = torch.empty((B, R, 4 * num_reg), device=dev, dtype=torch.float32)
    bw = torch.empty((B, R, 4 * num_reg), device=dev, dtype=torch.float32)
    _lib.call('relnet_proposal_target_ex', rois.contiguous().data_ptr(), gt_boxes.contiguous().data_ptr(), num_gt.data_ptr(),
              ro.data_ptr(), lab.data_ptr(), bt.data_ptr(), bw.data_ptr(), B, N, G, num_reg, int(class_agnostic),
              float(bg_thresh_hi), _d4(means), _d4(stds), _d4(weights), _ptr(num_rois), _stream())
    return ro, lab, bt, bw


def assign_anchor(gt_boxes, num_gt, im_info, base_anchors, feat_hw, feat_stride=16, rpn_batch_size=256, fg_fraction=0.5,
                  negative_overlap=0.3, positive_overlap=0.7, clobber_positives=False, allowed_border=0, seed=0,
                  want_all=False, seed_dev=None):
    """lib/rpn/rpn.py:80-244 for B images on the device.  gt_boxes [B,Gmax,5] fp32, num_gt [B] int32 (None: all valid),
    im_info [B,3], base_anchors [A,4] float64 (host or device) -> label [B, A*h*w] ((a,y,x) order), bbox_target /
    bbox_weight [B,4A,h,w] (+ the pre-sub-sampling labels with want_all).  `seed` (+ the device int64 word `seed_dev`, e.g. a
    step counter advanced inside a captured graph) selects the random fg / bg subset."""
    import ctypes
    _chk(gt_boxes, num_gt, im_info)
    B, G, _ = gt_boxes.shape
    dev = gt_boxes.device
    assert gt_boxes.dtype == torch.float32 and im_info.dtype == torch.float32
    if num_gt is None:
        num_gt = torch.full((B,), G, device=dev, dtype=torch.int32)
    if G == 0:                       # images without any box: one (ignored) gt slot keeps the buffers non-empty
        gt_boxes, G = torch.zeros((B, 1, 5), device=dev, dtype=torch.float32), 1
    base = torch.as_tensor(base_anchors).detach()
    if base.is_cuda:
        raise _lib.RelnetError('assign_anchor: base_anchors is a HOST table (a device tensor would need a synchronising copy)')
    base = base.to(torch.float64).contiguous()
    A = base.shape[0]
    fh, fw = feat_hw
    label = torch.empty((B, A * fh * fw), device=dev, dtype=torch.float32)
    bt = torch.empty((B, 4 * A, fh, fw), device=dev, dtype=torch.float32)
    bw = torch.empty((B, 4 * A, fh, fw), device=dev, dtype=torch.float32)
    lall = torch.empty_like(label) if want_all else None
    ws = torch.empty((B, G), device=dev, dtype=torch.int64)
    _lib.call('relnet_assign_anchor', gt_boxes.contiguous().data_ptr(), num_gt.data_ptr(), im_info.contiguous().data_ptr(),
              base.data_ptr(), label.data_ptr(), bt.data_ptr(), bw.data_ptr(), _ptr(lall), ws.data_ptr(), B, A, fh, fw, G,
              int(feat_stride), int(rpn_batch_size), int(fg_fraction * rpn_batch_size), float(negative_overlap),
              float(positive_overlap), int(clobber_positives), int(allowed_border), ctypes.c_ulonglong(int(seed) & (2 ** 64 - 1)),
              _ptr(seed_dev), _stream())
    return (label, bt, bw, lall) if want_all else (label, bt, bw)


def box_annotator_ohem(cls_score, bbox_pred, labels, bbox_targets, bbox_weights, roi_per_img=128, want_loss=False):
    """[B,R,C], [B,R,D], [B,R], [B,R,D], [B,R,D] -> labels_ohem [B,R], bbox_weights_ohem [B,R,D]."""
    _chk(cls_score, bbox_pred, labels, bbox_targets, bbox_weights)
    B, R, Cn = cls_score.shape
    D = bbox_pred.shape[2]
    lo = torch.empty_like(labels)
    wo = torch.empty_like(bbox_weights)
    loss = torch.empty((B, R), device=labels.device, dtype=torch.float32) if want_loss else None
    _lib.call('relnet_box_annotator_ohem', cls_score.contiguous().data_ptr(), bbox_pred.contiguous().data_ptr(),
              labels.contiguous().data_ptr(), bbox_targets.contiguous().data_ptr(), bbox_weights.contiguous().data_ptr(),
              lo.data_ptr(), wo.data_ptr(), _ptr(loss), B, R, Cn, D, int(roi_per_img), _stream())
    return (lo, wo, loss) if want_loss else (lo, wo)


def nms_multi_target(bbox, gt_boxes, score, num_gt=None, target_thresh=(0.5, 0.6, 0.7, 0.8, 0.9)):
    """bbox [B,F,C,4], gt_boxes [B,Gmax,5], score [B,F,C] -> nms_multi_target [B,F,C,T]."""
    _chk(bbox, gt_boxes, score, num_gt)
    import ctypes
    B, F, Cn, _ = bbox.shape
    G = gt_boxes.shape[1]
    if num_gt is None:
        num_gt = torch.full((B,), G, device=bbox.device, dtype=torch.int32)
    T = len(target_thresh)
    out = torch.empty((B, F, Cn, T), device=bbox.device, dtype=torch.float32)
    th = (ctypes.c_double * T)(*[float(t) for t in target_thresh])
    _lib.call('relnet_nms_multi_target', bbox.contiguous().data_ptr(), gt_boxes.contiguous().data_ptr(), num_gt.data_ptr(),
              score.contiguous().data_ptr(), out.data_ptr(), B, F, Cn, G, th, T, _stream())
    return out


# ---------------------------------------------------------------------------------------
# DCN configuration
# ---------------------------------------------------------------------------------------
def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def deformable_im2col(data, offset, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(0, 0),
                      num_deformable_group=1, col_dtype=None, col=None):
    """data: logical [B,C,H,W] of any strides (NCHW fp32 or channels_last bf16); offset: fp32 logical
    [B, 2*kh*kw*dg, Ho, Wo] of any strides -> col [B*Ho*Wo, kh*kw*C] (column (i*kw+j)*C + c);
    nn/deformable_im2col.cuh:215-262."""
    _chk(data, offset, col)
    assert offset.dtype == torch.float32
    kh, kw = _pair(kernel); sh, sw = _pair(stride); dh, dw = _pair(dilate); ph, pw = _pair(pad)
    B, Cc, H, W = data.shape
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    if tuple(offset.shape) != (B, 2 * kh * kw * num_deformable_group, Ho, Wo):
        raise ValueError("offset shape %s, expected %s (output size and 2*kh*kw*num_deformable_group channels, "
                         "deformable_convolution-inl.h:389-396)" % (tuple(offset.shape), (B, 2 * kh * kw * num_deformable_group, Ho, Wo)))
    cdt = col_dtype or (col.dtype if col is not None else data.dtype)
    if col is None:
        col = torch.empty((B * Ho * Wo, kh * kw * Cc), device=data.device, dtype=cdt)
    assert col.stride(1) == 1 and col.shape[0] == B * Ho * Wo
    _lib.call('relnet_deformable_im2col', data.data_ptr(), _strides4(data), offset.data_ptr(), _strides4(offset),
              col.data_ptr(), col.stride(0), B, Cc, H, W, kh, kw, ph, pw, sh, sw, dh, dw, num_deformable_group,
              _dt(data), _dt(col), _stream(), tag='B%d_C%d_%dx%d' % (B, Cc, Ho, Wo))
    return col, (Ho, Wo)


def deformable_conv(data, offset, w_packed, bias=None, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(0, 0),
                    num_deformable_group=1, relu=False, out_dtype=None, want_col=False):
    """DeformableConvolutionOp::Forward (deformable_convolution-inl.h:91-143) with the bias (or folded BatchNorm) and ReLU
    fused into the GEMM epilogue: sampling kernel (relnet_deformable_im2col, column matrix in (tap, channel) order) + NT GEMM
    (the res5 shape, K = 4608, runs on the hand-scheduled ring kernel).
    w_packed [Cout, kh*kw*Cin] (pack_conv_weight order).  Returns logical [B, Cout, Ho, Wo] in channels-last memory."""
    col, (Ho, Wo) = deformable_im2col(data, offset, kernel, stride, dilate, pad, num_deformable_group,
                                      col_dtype=w_packed.dtype)
    y = gemm_nt(col, w_packed, bias, relu=relu, out_dtype=out_dtype)
    y = y.view(data.shape[0], Ho, Wo, w_packed.shape[0]).permute(0, 3, 1, 2)
    # want_col (training): the sampled column matrix [B Ho Wo, kh kw C] is also the X operand of the weight gradient -- kept, it saves the backward a
    # second sampling pass (deformable_conv_bwd(col=...))
    return (y, col) if want_col else y


def deformable_psroi_pool(data, rois, trans=None, spatial_scale=0.0625, output_dim=256, group_size=1,
                          pooled_size=7, part_size=0, sample_per_part=1, trans_std=0.0, no_trans=False,
                          channels_last_out=False, want_top_count=False, batch_index_base=0):
    """DeformablePSROIPooling forward (deformable_psroi_pooling.cu:51-138).  data logical
    [B, output_dim*group_size^2, H, W] (any strides); rois [R,5]; trans [R, 2*num_classes, part, part]
    fp32 unless no_trans.  Output logical [R, output_dim, P, P] (memory (R,P,P,C) when channels_last_out)."""
    _chk(data, rois, trans)
    assert rois.dtype == torch.float32 and rois.is_contiguous()
    B, Cc, H, W = data.shape
    R, P = rois.shape[0], int(pooled_size)
    num_classes = 0
    if not no_trans:
        if trans is None:
            raise ValueError("DeformablePSROIPooling: trans is required unless no_trans (deformable_psroi_pooling-inl.h:70-72)")
        assert trans.dtype == torch.float32 and trans.is_contiguous()
        part = part_size or P
        num_classes = trans.shape[1] // 2
        assert tuple(trans.shape) == (R, 2 * num_classes, part, part), trans.shape
    if channels_last_out:
        out = torch.empty((R, P, P, output_dim), device=data.device, dtype=data.dtype).permute(0, 3, 1, 2)
    else:
        out = torch.empty((R, output_dim, P, P), device=data.device, dtype=data.dtype)
    cnt = torch.empty_strided(out.shape, out.stride(), device=data.device, dtype=torch.float32) if want_top_count else None
    _lib.call('relnet_deformable_psroi_pool_fwd', data.data_ptr(), _strides4(data), rois.data_ptr(),
              0 if no_trans else trans.data_ptr(), out.data_ptr(), _strides4(out), _ptr(cnt), R, Cc, H, W,
              int(output_dim), int(group_size), P, int(part_size), int(sample_per_part), float(spatial_scale),
              float(trans_std), num_classes, batch_index_base, _dt(data), _stream())
    return (out, cnt) if want_top_count else out


# ---------------------------------------------------------------------------------------
# FPN configuration
# ---------------------------------------------------------------------------------------
def fpn_roi_dispatch(rois, batch_index_base=0, n_valid=None, pad_empty=False):
    """rois [B,N,4] (xyxy) or [B,N,5] (idx + xyxy) fp32 -> (rois_sorted [B,N',5], level [B,N'] int32,
    perm [B,N'] int32, counts [B,4] int32); core/rcnn.py:53-74 on the device.  N' = N, or N + 4 with pad_empty: a pyramid
    level without a roi then gets the reference's all-zero dummy roi (rcnn.py:61-71; perm -1) and a fifth result,
    n_rows [B] int32 = the image's real rows (rois + dummies), is returned; rows past it are padding.
    n_valid [B] int32 (optional): only the first n_valid[b] input rows are rois (the others are moved behind the real rows)."""
    _chk(rois, n_valid)
    assert rois.dtype == torch.float32 and rois.is_contiguous() and rois.dim() == 3
    B, N, bs = rois.shape
    n_out = N + 4 if pad_empty else N
    want_rows = pad_empty or n_valid is not None
    out = torch.empty((B, n_out, 5), device=rois.device, dtype=torch.float32)
    level = torch.empty((B, n_out), device=rois.device, dtype=torch.int32)
    perm = torch.empty((B, n_out), device=rois.device, dtype=torch.int32)
    counts = torch.empty((B, 4), device=rois.device, dtype=torch.int32)
    n_rows = torch.empty((B,), device=rois.device, dtype=torch.int32) if want_rows else None
    _lib.call('relnet_fpn_roi_dispatch_ex', rois.data_ptr(), bs, bs - 4, out.data_ptr(), level.data_ptr(),
              perm.data_ptr(), counts.data_ptr(), B, N, batch_index_base, _ptr(n_valid), int(pad_empty), n_out,
              _ptr(n_rows), _stream())
    return (out, level, perm, counts, n_rows) if want_rows else (out, level, perm, counts)


def roi_pool_fpn(levels, scales, rois, roi_level, pooled=(7, 7), channels_last_out=False, batch_index_base=0,
                 want_argmax=False):
    """levels: list of logical [B,C,H_l,W_l] maps (any strides, same C / dtype); rois [R,5]; roi_level [R] int32
    -> [R,C,PH,PW] (memory (R,PH,PW,C) when channels_last_out): 4 x ROIPooling + Concat in one launch."""
    import ctypes
    _chk(rois, roi_level, *levels)
    assert rois.dtype == torch.float32 and rois.is_contiguous() and roi_level.dtype == torch.int32
    nl = len(levels)
    Cc = levels[0].shape[1]
    R = rois.shape[0]
    PH, PW = pooled
    ptrs = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in levels])
    strides = (ctypes.c_long * (4 * nl))(*[int(s) for t in levels for s in t.stride()])
    hs = (ctypes.c_int * nl)(*[int(t.shape[2]) for t in levels])
    ws = (ctypes.c_int * nl)(*[int(t.shape[3]) for t in levels])
    sc = (ctypes.c_float * nl)(*[float(x) for x in scales])
    dt = levels[0].dtype
    assert all(t.dtype == dt and t.shape[1] == Cc for t in levels)
    if channels_last_out:
        out = torch.empty((R, PH, PW, Cc), device=rois.device, dtype=dt).permute(0, 3, 1, 2)
    else:
        out = torch.empty((R, Cc, PH, PW), device=rois.device, dtype=dt)
    arg = torch.empty_strided(out.shape, out.stride(), device=rois.device, dtype=torch.int32) if want_argmax else None
    _lib.call('relnet_roi_pool_fpn_fwd', ctypes.addressof(ptrs), ctypes.addressof(strides), ctypes.addressof(hs),
              ctypes.addressof(ws), ctypes.addressof(sc), nl, rois.data_ptr(), roi_level.data_ptr(), out.data_ptr(),
              _strides4(out), _ptr(arg), R, Cc, PH, PW, batch_index_base, _dt(levels[0]), _stream())
    return (out, arg) if want_argmax else out


def roi_pool_fpn_bwd(grad_out, argmax, rois, roi_level, level_shapes, batch_index_base=0, channels_last=False):
    """Adjoint of roi_pool_fpn: level_shapes = [(B,C,H_l,W_l)] -> list of fp32 gradients, logical [B,C,H_l,W_l] (memory NCHW,
    or NHWC with channels_last)."""
    import ctypes
    _chk(grad_out, argmax, rois, roi_level)
    assert argmax.dtype == torch.int32 and tuple(grad_out.stride()) == tuple(argmax.stride())
    R, Cc, PH, PW = grad_out.shape
    if channels_last:
        gins = [torch.zeros((sh[0], sh[2], sh[3], sh[1]), device=grad_out.device, dtype=torch.float32).permute(0, 3, 1, 2) for sh in level_shapes]
    else:
        gins = [torch.zeros(tuple(sh), device=grad_out.device, dtype=torch.float32) for sh in level_shapes]
    nl = len(gins)
    ptrs = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in gins])
    gb = (ctypes.c_long * nl)(*[int(t.stride(0)) for t in gins])
    gc = (ctypes.c_long * nl)(*[int(t.stride(1)) for t in gins])
    gp = (ctypes.c_long * nl)(*[int(t.stride(3)) for t in gins])
    _lib.call('relnet_roi_pool_fpn_bwd_ex', grad_out.data_ptr(), argmax.data_ptr(), _strides4(grad_out), rois.data_ptr(),
              roi_level.data_ptr(), ctypes.addressof(ptrs), ctypes.addressof(gb), ctypes.addressof(gc), ctypes.addressof(gp),
              nl, R, Cc, PH, PW, batch_index_base, _dt(grad_out), _stream())
    return gins


def upsample2x_add_(lateral, top):
    """lateral [B,H,W,C] += nearest-upsampled top [B,H/2,W/2,C] (both NHWC contiguous), in place."""
    _chk(lateral, top)
    assert lateral.is_contiguous() and top.is_contiguous() and lateral.dtype == top.dtype
    B, H, W, Cc = lateral.shape
    if tuple(top.shape) != (B, H // 2, W // 2, Cc) or H % 2 or W % 2:
        raise ValueError("upsample2x_add: lateral %s is not exactly twice top %s (pad images to IMAGE_STRIDE 32)"
                         % (tuple(lateral.shape), tuple(top.shape)))
    _lib.call('relnet_upsample2x_add', top.data_ptr(), lateral.data_ptr(), B, H, W, Cc, _dt(lateral), _stream())
    return lateral


# ---------------------------------------------------------------------------------------
# relation module backward (training)
# ---------------------------------------------------------------------------------------
def pad_to(n, m):
    return (n + m - 1) // m * m


def transpose_2d(x, out=None, pad_cols_to=1):
    """x [rows, cols] or [batch, rows, cols] (last dim contiguous) -> [.., cols, pad(rows)] (zero padded)."""
    _chk(x, out)
    assert x.stride(-1) == 1
    batch = x.shape[0] if x.dim() == 3 else 1
    rows, cols = x.shape[-2], x.shape[-1]
    rp = pad_to(rows, pad_cols_to)
    if out is None:
        shape = (batch, cols, rp) if x.dim() == 3 else (cols, rp)
        out = torch.empty(shape, device=x.device, dtype=x.dtype)
        if rp != rows:
            out[..., rows:].zero_()                       # only the padding columns need clearing
    assert out.stride(-1) == 1 and out.shape[-2] == cols and out.shape[-1] >= rows
    _lib.call('relnet_transpose_2d', x.data_ptr(), x.stride(-2), x.stride(0) if x.dim() == 3 else 0, out.data_ptr(),
              out.stride(-2), out.stride(0) if out.dim() == 3 else 0, rows, cols, batch, _dt(x), _stream())
    return out


class WeightRelayout(object):
    """Data-gradient copies of a set of bf16 weights, refreshed by ONE launch (relnet_weight_relayout).
    add(name, w [Cout, taps*Cin], taps, pad_co) registers a weight VIEW (its storage must stay where it is: the trainer's flat
    bf16 working copy); build() allocates the copies [Cin, taps * pad(Cout)] (zero: the pad columns stay zero) and uploads the
    descriptor table; run() launches; get(name) returns the copy: W^T for taps = 1, the tap-flipped transposed 3x3 filter
    ([Cin, 9 * Cout], what the implicit-GEMM convolution of dy needs) for taps = 9."""

    def __init__(self, device):
        self.device = device
        self.items = []
        self.views = {}
        self.table = None

    def add(self, name, w, taps=1, pad_co=64, group=None):
        """group = (group name, column offset): the copy is written into columns [offset, offset + pad(Cout)) of a buffer shared by
        the group's members (taps = 1, equal Cin) -- e.g. [Wq; Wk]^T | Wout^T side by side = the [Fd, 3 d] operand of the one-GEMM
        projection backward of a relation module; get(group name) returns the whole buffer, get(name) the member's columns."""
        assert w.dtype == torch.bfloat16 and w.is_contiguous() and w.shape[1] % taps == 0
        assert group is None or taps == 1
        self.items.append((name, w, taps, pad_co, group))

    def build(self):
        if not self.items:
            return
        sizes = []
        groups = {}                                               # group name -> [cin, total columns]
        for name, w, taps, pad_co, group in self.items:
            cout, cin = w.shape[0], w.shape[1] // taps
            dst_co = (cout + pad_co - 1) // pad_co * pad_co
            sizes.append((cout, cin, dst_co))
            if group is not None:                                 # (group name, column offset[, total columns of the group buffer])
                gi = groups.setdefault(group[0], [cin, 0])
                assert gi[0] == cin and group[1] % 8 == 0, (name, group)
                gi[1] = max(gi[1], group[1] + dst_co, group[2] if len(group) > 2 else 0)
        total = sum(cin * taps * dst_co for (_, _, taps, _, group), (cout, cin, dst_co) in zip(self.items, sizes) if group is None)
        total += sum(cin * cols + 8 for cin, cols in groups.values())
        self.flat = torch.zeros(total + 8 * len(self.items) + 64, device=self.device, dtype=torch.bfloat16)
        arr = (_lib.RelayoutDesc * len(self.items))()
        off = 0
        tile = 0
        for gname, (cin, cols) in groups.items():
            off = (off + 7) // 8 * 8
            self.views[gname] = self.flat[off:off + cin * cols].view(cin, cols)
            off += cin * cols
        for i, ((name, w, taps, pad_co, group), (cout, cin, dst_co)) in enumerate(zip(self.items, sizes)):
            if group is not None:
                gbuf = self.views[group[0]]
                v = gbuf[:, group[1]:group[1] + dst_co]           # strided member view: rows of the group buffer
                dst_ld = gbuf.shape[1]
            else:
                off = (off + 7) // 8 * 8                          # 16-byte aligned copies
                n = cin * taps * dst_co
                v = self.flat[off:off + n].view(cin, taps * dst_co)
                dst_ld = taps * dst_co
                off += n
            self.views[name] = v
            d = arr[i]
            d.src, d.dst = w.data_ptr(), v.data_ptr()
            d.cout, d.cin, d.taps, d.dst_ld, d.dst_co = cout, cin, taps, dst_ld, dst_co
            d.tiles_co, d.tiles_ci, d.tile_start = (cout + 63) // 64, (cin + 63) // 64, tile
            tile += taps * d.tiles_co * d.tiles_ci
        self.total_tiles = tile
        raw = bytes(memoryview(arr))
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self.n = len(self.items)

    def run(self):
        if self.table is not None:
            _lib.call('relnet_weight_relayout', self.table.data_ptr(), self.n, self.total_tiles, _stream(), tag='n%d' % self.n)

    def get(self, name):
        return self.views.get(name)


class FragRepack(object):
    """MFMA-fragment-order copies of a set of bf16 weights, refreshed by ONE launch per step (relnet_weight_fragpack): what
    relnet_bottleneck_chain reads (pack_w_frag order for a unit's expand weights W3, pack_chain_w1 order for the next unit's reduce
    weights W1').  add(key, w [N, K] view of the trainer's flat bf16 working copy, mode) -> build() -> run() -> get(key)."""

    def __init__(self, device):
        self.device = device
        self.items, self.views, self.table = [], {}, None

    def add(self, key, w, mode):
        assert w.dtype == torch.bfloat16 and w.is_contiguous() and w.dim() == 2 and w.shape[0] % 32 == 0 and w.shape[1] % 16 == 0
        assert mode in (0, 1)
        self.items.append((key, w, mode))

    def build(self):
        if not self.items:
            return
        total = sum(w.numel() for _, w, _ in self.items)
        self.flat = torch.zeros(total + 8, device=self.device, dtype=torch.bfloat16)
        arr = (_lib.FragPackDesc * len(self.items))()
        off = blk = 0
        for i, (key, w, mode) in enumerate(self.items):
            N, K = w.shape
            v = self.flat[off:off + N * K]
            self.views[key] = v
            d = arr[i]
            d.src, d.dst, d.ldw, d.N, d.K, d.mode, d.block_start = w.data_ptr(), v.data_ptr(), w.stride(0), N, K, mode, blk
            blk += ((N // 32) * (K // 16) * 64 + 255) // 256
            off += N * K
        self.total_blocks, self.n = blk, len(self.items)
        self.table = torch.frombuffer(bytearray(bytes(memoryview(arr))), dtype=torch.uint8).to(self.device)

    def run(self):
        if self.table is not None:
            _lib.call('relnet_weight_fragpack', self.table.data_ptr(), self.n, self.total_blocks, _stream(), tag='n%d' % self.n)

    def get(self, key):
        return self.views.get(key)


def relation_bwd_pack(dq, dk, dvw, out=None):
    """dq [B,N,d], dk / dvw [B,M,d] fp32 -> out [B,N,3d] bf16 = (dQ | dK | dVW), rows >= M of the key blocks zero."""
    _chk(dq, dk, dvw, out)
    B, N, d = dq.shape
    M = dk.shape[1]
    assert dq.dtype == torch.float32 and dk.shape == (B, M, d) and dvw.shape == (B, M, d) and dq.is_contiguous() and dk.is_contiguous() and dvw.is_contiguous()
    if out is None:
        out = torch.empty((B, N, 3 * d), device=dq.device, dtype=torch.bfloat16)
    assert out.is_contiguous() and out.shape == (B, N, 3 * d) and out.dtype == torch.bfloat16
    _lib.call('relnet_relation_bwd_pack', dq.data_ptr(), dk.data_ptr(), dvw.data_ptr(), out.data_ptr(), B, N, M, d, _stream())
    return out


def lnms_scatter_bwd(d_sorted, rank_idx, N):
    """d_sorted [B,F,C] fp32, rank_idx [B,C,F] int32 -> d_prob [B,N,C] fp32 with d_prob[b, rank_idx[b,c,f], c] += d_sorted[b,f,c]."""
    _chk(d_sorted, rank_idx)
    B, F, Cn = d_sorted.shape
    assert rank_idx.shape == (B, Cn, F) and rank_idx.dtype == torch.int32 and rank_idx.is_contiguous()
    assert d_sorted.dtype == torch.float32 and d_sorted.is_contiguous()
    d_prob = torch.zeros((B, N, Cn), device=d_sorted.device, dtype=torch.float32)
    _lib.call('relnet_lnms_scatter_bwd', d_sorted.data_ptr(), rank_idx.data_ptr(), d_prob.data_ptr(), B, N, Cn, F, _stream())
    return d_prob


class WgradQueue(object):
    """Weight-gradient products collected for ONE grouped launch (csrc/wgrad.hip: stream-K over the (layer, tile, slab) units
    of all queued layers).  `add` has the signature of `wgrad_tn`; `flush` launches what is queued.  The queue keeps the
    operand tensors alive until the launch has been issued (stream order does the rest)."""

    def __init__(self):
        self.items, self.keep = [], []

    def add(self, dy2d, x, out, row_scale=None, cout=None, conv=None):
        _chk(dy2d, x, out, row_scale)
        assert dy2d.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy2d.dim() == 2 and dy2d.stride(1) == 1
        P = dy2d.shape[0]
        cout = dy2d.shape[1] if cout is None else cout
        if conv is None:
            assert x.dim() == 2 and x.shape[0] == P and x.stride(1) == 1
            cin, ks, stride, dil, pad = x.shape[1], 1, 1, 1, 0
            B = Ho = Wo = Hin = Win = 1
            x_pix = x.stride(0)
        else:
            ks, stride, dil, pad = conv
            B, Hin, Win, cin = x.shape
            assert x.stride(3) == 1 and x.stride(1) == Win * x.stride(2) and x.stride(0) == Hin * x.stride(1)
            Ho = (Hin + 2 * pad - dil * (ks - 1) - 1) // stride + 1
            Wo = (Win + 2 * pad - dil * (ks - 1) - 1) // stride + 1
            assert B * Ho * Wo == P, (x.shape, conv, P)
            x_pix = x.stride(2)
        K = ks * ks * cin
        assert out.dtype == torch.float32 and tuple(out.shape) == (cout, K) and out.stride(1) == 1
        if row_scale is not None:
            assert row_scale.dtype == torch.float32 and row_scale.numel() == cout and row_scale.is_contiguous()
        d = _lib.WgradDesc(dy2d.data_ptr(), dy2d.stride(0), dy2d.shape[1], x.data_ptr(), x_pix, out.data_ptr(), out.stride(0),
                           _ptr(row_scale) or None, P, cout, cin, ks, stride, dil, pad, B, Ho, Wo, Hin, Win)
        self.items.append(d)
        self.keep.append((dy2d, x, out, row_scale))

    def __len__(self):
        return len(self.items)

    def flush(self, workgroups=0):
        """One grouped launch of everything queued.  workgroups: size of the persistent grid (0 = one per CU) -- a launch that runs
        BESIDE other work on another stream takes a part of the chip only."""
        n = len(self.items)
        if n == 0:
            return
        arr = (_lib.WgradDesc * n)(*self.items)
        lib = _lib.load()
        ws = torch.empty(int(lib.relnet_wgrad_workspace_bytes(n)), device=self.keep[0][0].device, dtype=torch.uint8)
        import ctypes
        if workgroups:
            lib.relnet_wgrad_tune(int(workgroups), 0, 0)
        try:
            _lib.call('relnet_wgrad_grouped', ctypes.addressof(arr), n, ws.data_ptr(), _stream(), tag='n%d' % n)
        finally:
            if workgroups:
                lib.relnet_wgrad_tune(0, 0, 0)
        self.items, self.keep = [], []
        return ws          # the caller may hold on to it; stream order already protects it from reuse on this stream


def wgrad_tn(dy2d, x, out=None, row_scale=None, cout=None, conv=None):
    """Weight gradient  out [Cout, K] fp32 += row_scale^2 * dy2d^T X  straight from the pixel-major operands (csrc/wgrad.hip:
    LDS-transposed MFMA fragments, implicit im2col, stream-K over (tile, 64-pixel slab) units; no transposed copies).
    dy2d [P, >= Cout] bf16 (row stride a multiple of 8; columns >= cout must be zero padding).
    conv None: x [P, K] bf16 rows;  conv = (ksize, stride, dil, pad): x [B, Hin, Win, Cin] NHWC (any pixel stride) and
    dy2d the [B * Hout * Wout, Cout] gradient of that convolution's output, K = ksize^2 Cin in pack_conv_weight order.
    out None: a fresh zero tensor is returned (otherwise accumulated into `out`, e.g. a view of the flat gradient buffer).
    One layer per launch; `WgradQueue` groups many layers into one."""
    if out is None:
        cin = x.shape[-1]
        k = 1 if conv is None else conv[0]
        out = torch.zeros((dy2d.shape[1] if cout is None else cout, k * k * cin), device=dy2d.device, dtype=torch.float32)
    q = WgradQueue()
    q.add(dy2d, x, out, row_scale, cout, conv)
    q.flush()
    return out


def relation_bwd_small_ok(dtype, N, Mpad):
    """The one-workgroup-per-(image, head) backward (relation_attention_bwd_small_kernel) covers bf16 operands with N, Mpad <= 128."""
    return dtype == torch.bfloat16 and N <= 128 and Mpad <= 128 and os.environ.get('RELNET_REL_BWD_SMALL', '1') != '0'


def relation_attention_bwd(q, k, kt, vw, bias, dy, y, bout, qt, dyt, M, heads=16, key_count=None, packed_out=None):
    """Adjoint of relation_attention: -> (dq [B,N,H*64], dk [B,M,H*64], dvw [B,M,H*64], prob | None, dlog [B,H,N,Mpad]), fp32
    (prob is None on the small-N path: S never leaves LDS there)."""
    _chk(q, k, kt, vw, bias, dy, y, bout, qt, dyt, key_count)
    B, N = q.shape[0], q.shape[1]
    H = heads
    Mpad = bias.shape[-1]
    assert bias.dtype == torch.float32 and bias.shape == (B, H, N, Mpad) and bias.is_contiguous()
    dev = q.device
    # N, Mpad <= 128 (the learn-NMS head's module): one workgroup per (image, head) does the q and the kv part with S / dL in LDS --
    # no `prob` map at all (prob = NULL selects that kernel); RELNET_REL_BWD_SMALL=0 keeps the two-kernel form for the A/B
    small = relation_bwd_small_ok(q.dtype, N, Mpad)
    if small:          # that kernel transposes K / Q / dY on the fly from their row-major images in LDS: kt / qt / dyt are not read (may be None)
        kt = qt = dyt = None
        Npad = pad32(N)
    else:
        Npad = qt.shape[-1]
        assert kt.shape[1] == H * 64 and kt.shape[2] >= Mpad and dyt.shape == qt.shape and qt.shape[1] == H * 64
    prob = None if small else torch.empty((B, H, N, Mpad), device=dev, dtype=torch.float32)
    dlog = torch.empty((B, H, N, Mpad), device=dev, dtype=torch.float32)
    if packed_out is not None:
        # packed_out [B, N, 3 H 64] bf16 with zero key blocks past row M (a persistent buffer): the small-N kernel writes (dQ | dK | dVW) there
        # as bf16 -- the operand of the projection backward -- and no fp32 dq / dk / dvw exist; -> (packed_out, None, None, None, dlog)
        assert small and packed_out.dtype == torch.bfloat16 and packed_out.is_contiguous() and tuple(packed_out.shape) == (B, N, 3 * H * 64)
        dq, dk, dvw = packed_out, None, None
    else:
        dq = torch.empty((B, N, H * 64), device=dev, dtype=torch.float32)
        dk = torch.empty((B, M, H * 64), device=dev, dtype=torch.float32)
        dvw = torch.empty((B, M, H * 64), device=dev, dtype=torch.float32)
    _lib.call('relnet_relation_attention_bwd_kc',
              q.data_ptr(), q.stride(1), q.stride(0), k.data_ptr(), k.stride(1), k.stride(0),
              _ptr(kt), kt.stride(1) if kt is not None else 0, kt.stride(0) if kt is not None else 0, vw.data_ptr(), vw.stride(1), vw.stride(0),
              bias.data_ptr(), bias.stride(0), dy.data_ptr(), dy.stride(1), dy.stride(0),
              y.data_ptr(), y.stride(1), y.stride(0), _ptr(bout), _ptr(qt), qt.stride(1) if qt is not None else 0, qt.stride(0) if qt is not None else 0,
              _ptr(dyt), dyt.stride(1) if dyt is not None else 0, dyt.stride(0) if dyt is not None else 0, _ptr(prob), dlog.data_ptr(), dq.data_ptr(),
              _ptr(dk), _ptr(dvw), B, H, N, M, Mpad, Npad, 1.0 / math.sqrt(64.0), _dt(q), _ptr(key_count), _stream())
    return dq, dk, dvw, prob, dlog


def geometry_bias_bwd(boxes, bias, dlog, M, divisors=None, fast=False, out=None):
    """boxes [B,N,4|5]; bias / dlog [B,16,N,Mpad] fp32 -> (d pair_pos_fc1 weight [16,64], d bias [16]) fp32.
    out = (dwp, dbp): contiguous fp32 tensors the kernel ACCUMULATES into (atomic adds; e.g. views of a flat gradient buffer)."""
    _chk(boxes, bias, dlog)
    assert boxes.dtype == torch.float32 and boxes.is_contiguous()
    B, N, bs = boxes.shape
    assert bias.shape == dlog.shape and bias.shape[:3] == (B, 16, N) and bias.is_contiguous() and dlog.is_contiguous()
    div = (embedding_divisors() if divisors is None else divisors).to(torch.float32).cpu().contiguous()
    if out is not None:
        dwp, dbp = out
        assert dwp.dtype == torch.float32 and dbp.dtype == torch.float32 and dwp.is_contiguous() and dbp.is_contiguous()
        assert dwp.numel() == 16 * 64 and dbp.numel() == 16
    else:
        dwp = torch.zeros((16, 64), device=boxes.device, dtype=torch.float32)
        dbp = torch.zeros((16,), device=boxes.device, dtype=torch.float32)
    _lib.call('relnet_geometry_bias_bwd', boxes.data_ptr(), bs, 1 if bs == 5 else 0, bias.data_ptr(), dlog.data_ptr(),
              div.data_ptr(), dwp.data_ptr(), dbp.data_ptr(), B, N, M, bias.shape[-1], int(fast), _stream())
    return dwp, dbp


def deformable_conv_bwd(data, offset, w_packed, dy, kernel=(3, 3), stride=(1, 1), dilate=(1, 1), pad=(0, 0),
                        num_deformable_group=1, col=None):
    """Adjoint of deformable_conv.  data logical [B,C,H,W], offset fp32 logical [B,2*kh*kw*dg,Ho,Wo], dy logical
    [B,Cout,Ho,Wo] with channels-last memory ([B,Ho,Wo,Cout] contiguous).
    -> (grad_data fp32 [B,H,W,C] (NHWC), grad_offset fp32 [B,Ho,Wo,2*kh*kw*dg] (NHWC), grad_weight fp32 [Cout, kh*kw*C])."""
    from . import train_ops as T
    _chk(data, offset, w_packed, dy)
    kh, kw = _pair(kernel); sh, sw = _pair(stride); dh, dw = _pair(dilate); ph, pw = _pair(pad)
    B, Cc, H, W = data.shape
    Cout = w_packed.shape[0]
    dy2 = dy.permute(0, 2, 3, 1).reshape(-1, Cout)
    assert dy2.stride(1) == 1
    # column gradient: dcol [P, kh*kw*C] = dY [P, Cout] . W [Cout, kh*kw*C]
    gran = 64 if dy2.dtype == torch.bfloat16 else 16
    w_t = transpose_2d(w_packed, pad_cols_to=gran)
    dyp = dy2
    if Cout % gran:
        dyp = torch.zeros((dy2.shape[0], w_t.shape[1]), device=dy2.device, dtype=dy2.dtype)
        dyp[:, :Cout] = dy2
    # (bf16 layers: the column gradient is rounded to bf16 like every other data gradient of the bf16 trunk -- the gather form of col2im reads
    #  each of its rows ~4 times)
    dcol = gemm_nt(dyp, w_t, out_dtype=torch.float32 if data.dtype == torch.float32 else torch.bfloat16)
    gdata = torch.zeros((B, H, W, Cc), device=data.device, dtype=torch.float32).permute(0, 3, 1, 2)
    goff = torch.zeros((B, offset.shape[2], offset.shape[3], offset.shape[1]), device=data.device, dtype=torch.float32).permute(0, 3, 1, 2)
    _lib.call('relnet_deformable_col2im', dcol.data_ptr(), dcol.stride(0), _dt(dcol), data.data_ptr(), _strides4(data), _dt(data),
              offset.data_ptr(), _strides4(offset), gdata.data_ptr(), _strides4(gdata), goff.data_ptr(), _strides4(goff),
              B, Cc, H, W, kh, kw, ph, pw, sh, sw, dh, dw, num_deformable_group, _stream())
    if col is None or col.dtype != dy2.dtype:       # (col: the forward's column matrix, deformable_conv(want_col=True))
        col, _ = deformable_im2col(data, offset, kernel, stride, dilate, pad, num_deformable_group, col_dtype=dy2.dtype)
    gw = T.wgrad(dy2, col)
    return gdata.permute(0, 2, 3, 1), goff.permute(0, 2, 3, 1), gw


def deformable_psroi_pool_bwd(grad_out, data, rois, trans=None, spatial_scale=0.0625, output_dim=256, group_size=1,
                              pooled_size=7, part_size=0, sample_per_part=1, trans_std=0.0, no_trans=False,
                              batch_index_base=0):
    """Adjoint of deformable_psroi_pool: -> (grad_data fp32 logical [B,C,H,W] stored NHWC, grad_trans fp32 | None)."""
    _chk(grad_out, data, rois, trans)
    B, Cc, H, W = data.shape
    R = rois.shape[0]
    assert grad_out.dtype == data.dtype and rois.dtype == torch.float32 and rois.is_contiguous()
    num_classes = 0
    gtrans = None
    if not no_trans:
        assert trans is not None and trans.dtype == torch.float32 and trans.is_contiguous()
        num_classes = trans.shape[1] // 2
        gtrans = torch.zeros_like(trans)
    gdata = torch.zeros((B, H, W, Cc), device=data.device, dtype=torch.float32).permute(0, 3, 1, 2)
    _lib.call('relnet_deformable_psroi_pool_bwd', grad_out.data_ptr(), _strides4(grad_out), data.data_ptr(), _strides4(data),
              rois.data_ptr(), 0 if no_trans else trans.data_ptr(), gdata.data_ptr(), _strides4(gdata), _ptr(gtrans),
              R, Cc, H, W, int(output_dim), int(group_size), int(pooled_size), int(part_size), int(sample_per_part),
              float(spatial_scale), float(trans_std), num_classes, batch_index_base, _dt(data), _stream())
    return gdata, gtrans
