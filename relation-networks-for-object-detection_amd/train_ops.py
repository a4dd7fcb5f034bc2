"""Backward building blocks of the training step, composed from the HIP kernels:

  linear_bwd / conv1x1_bwd / conv3x3_bwd   data gradient  = GEMM / implicit-GEMM conv with transposed (and, for
                                           3x3, tap-flipped) weights;
                                           weight gradient = dY^T X as a split-K gemm_nt over transposed copies
                                           (relnet_transpose_2d), K = pixels, partial sums reduced in fp32
  relu_bwd                                 gradient through the fused (+residual)+ReLU epilogues
  sgd_update                               mx.optimizer.SGD (train_end2end.py:163-168)

MXNet derives these by autograd from the symbols; there is no backward source in the reference to follow.  Layout
is the forward's: NHWC bf16 activations, weights packed [Cout][R][S][Cin].
"""
import torch

from . import lib as _lib
from . import ops
from .ops import _chk, _dt, _ptr, _stream, pad_to


def relu_bwd(dy, y, add=None, out=None):
    """dx = dy * (y > 0) (+ add); y is the saved forward output of the ReLU."""
    _chk(dy, y, add, out)
    assert dy.is_contiguous() and y.is_contiguous() and dy.shape == y.shape and dy.dtype == y.dtype
    if add is not None:
        assert add.is_contiguous() and add.shape == dy.shape and add.dtype == dy.dtype
    out = torch.empty_like(dy) if out is None else out
    _lib.call('relnet_relu_bwd', dy.data_ptr(), y.data_ptr(), _ptr(add), out.data_ptr(), dy.numel(), _dt(dy), _stream())
    return out


def strided_scatter(low, shape, stride, mask=None):
    """Adjoint of x[:, ::stride, ::stride, :] (NHWC): low [B,Ho,Wo,C] -> [B,H,W,C] with zeros between the samples, times (mask > 0) when the
    full-resolution map is a ReLU output with no other consumer (one pass: relnet_strided_scatter)."""
    _chk(low, mask)
    B, H, W, C = shape
    assert low.is_contiguous() and low.shape[0] == B and low.shape[3] == C
    if mask is not None:
        assert mask.is_contiguous() and tuple(mask.shape) == tuple(shape) and mask.dtype == low.dtype
    out = torch.empty(shape, device=low.device, dtype=low.dtype)
    _lib.call('relnet_strided_scatter', low.data_ptr(), _ptr(mask), out.data_ptr(), B, H, W, C, low.shape[1], low.shape[2], int(stride), _dt(low), _stream())
    return out


class ColsumQueue(object):
    """Bias-gradient column sums collected for ONE grouped launch per gradient bucket (relnet_colsum_add_grouped, <= 16 problems per
    launch).  `add` keeps the operand alive until `flush`; operands the grouped kernel does not take (fp32, ragged widths) are summed at once."""

    def __init__(self):
        self.items = []

    def __len__(self):
        return len(self.items)

    def add(self, x2, out):
        ok = (x2.dtype == torch.bfloat16 and x2.shape[1] % 8 == 0 and x2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0)
        if not ok:
            _lib.call('relnet_colsum_add', x2.data_ptr(), x2.stride(0), x2.shape[0], x2.shape[1], _dt(x2), out.data_ptr(), _stream())
            return
        self.items.append((x2, out))

    def flush(self):
        import ctypes as C
        items, self.items = self.items, []
        for i in range(0, len(items), 16):
            grp = items[i:i + 16]
            n = len(grp)
            xs = (C.c_void_p * n)(*[x.data_ptr() for x, _ in grp])
            lds = (C.c_long * n)(*[x.stride(0) for x, _ in grp])
            rows = (C.c_long * n)(*[x.shape[0] for x, _ in grp])
            cols = (C.c_int * n)(*[x.shape[1] for x, _ in grp])
            outs = (C.c_void_p * n)(*[o.data_ptr() for _, o in grp])
            _lib.call('relnet_colsum_add_grouped', xs, lds, rows, cols, outs, n, _stream())
        return items            # (still referenced by the caller until the launch has been issued)


#: the queue bias-gradient sums go to while a training step is being recorded (train.Trainer sets it around forward_backward and flushes it
#: when a gradient bucket completes); None = every colsum_add launches at once
COLSUM_QUEUE = None


def colsum_add(x, out):
    """out[c] += sum over all leading dims of x[..., c]  (bias gradient accumulated in place; x bf16 / fp32 with a dense last dim,
    out fp32 [C] -- a view of the flat gradient buffer).  With an active COLSUM_QUEUE the sum is deferred to the queue's grouped launch:
    x must not be modified before that flush."""
    C = x.shape[-1]
    assert out.dtype == torch.float32 and out.numel() == C and out.is_contiguous() and x.stride(-1) == 1
    x2 = x.reshape(-1, C)
    _chk(x2, out)
    if COLSUM_QUEUE is not None:
        COLSUM_QUEUE.add(x2, out)
        return
    _lib.call('relnet_colsum_add', x2.data_ptr(), x2.stride(0), x2.shape[0], C, _dt(x2), out.data_ptr(), _stream())


def sgd_update(w, mom, grad, lr, momentum=0.9, wd=0.0005, rescale_grad=1.0, w_bf16=None):
    """In place on fp32 `w` / `mom`; optional bf16 copy refreshed in the same pass."""
    _chk(w, mom, grad, w_bf16)
    assert w.dtype == torch.float32 and mom.dtype == torch.float32 and grad.dtype == torch.float32
    assert w.is_contiguous() and mom.is_contiguous() and grad.is_contiguous() and grad.numel() == w.numel()
    if w_bf16 is not None:
        assert w_bf16.dtype == torch.bfloat16 and w_bf16.is_contiguous() and w_bf16.numel() == w.numel()
    _lib.call('relnet_sgd_update', w.data_ptr(), mom.data_ptr(), grad.data_ptr(), _ptr(w_bf16), w.numel(), float(lr),
              float(momentum), float(wd), float(rescale_grad), _stream())


def _splits_for(m, n, k):
    """split-K factor of a weight-gradient GEMM [m, n] with a long contraction k: enough tiles to fill 256 CUs."""
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    s = max(1, min(512 // max(tiles, 1), k // 512))
    return s


def wgrad(dy2d, x2d, keep_splits=False):
    """dW [Cout, K] = dY^T X for dY [P, Cout], X [P, K] (bf16 or fp32, rows = pixels / rois); fp32 result.
    keep_splits: return the [splits, Cout, K] partial sums (for wgrad_accumulate) instead of their sum."""
    P, Cout = dy2d.shape
    K = x2d.shape[1]
    gran = 64 if dy2d.dtype == torch.bfloat16 else 16
    s = _splits_for(Cout, K, P)
    Pp = pad_to(P, gran * s)
    dyt = ops.transpose_2d(dy2d, pad_cols_to=gran * s)                 # [Cout, Pp], zero padded
    xt = ops.transpose_2d(x2d, pad_cols_to=gran * s)                   # [K, Pp]
    ks = Pp // s
    a3 = dyt.as_strided((s, Cout, ks), (ks, dyt.stride(0), 1))
    w3 = xt.as_strided((s, K, ks), (ks, xt.stride(0), 1))
    part = ops.gemm_nt(a3, w3, out_dtype=torch.float32)                 # [s, Cout, K] partial sums
    if keep_splits:
        return part
    return part.sum(0) if s > 1 else part[0]


def scalar_sum(x, scale=1.0, count_nonneg=False):
    """0-dim fp32 tensor = scale * x.sum() (or the number of entries >= 0): relnet_reduce_scalar, one launch (the loss values /
    OHEM count a step reports; torch needs sum + div / ge + sum)."""
    _chk(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        return (x >= 0).sum().float() if count_nonneg else x.sum() * scale
    out = torch.empty((), device=x.device, dtype=torch.float32)
    _lib.call('relnet_reduce_scalar', x.data_ptr(), x.numel(), float(scale), int(bool(count_nonneg)), out.data_ptr(), _stream())
    return out


def wgrad_accumulate(parts, grad, row_scale=None):
    """grad [rows, cols] (fp32 view of the flat gradient buffer) += row_scale^2 * parts.sum(0), one kernel."""
    _chk(parts, grad, row_scale)
    if parts.dim() == 2:
        parts = parts[None]
    S, rows, cols = parts.shape
    assert parts.is_contiguous() and grad.is_contiguous() and grad.numel() == rows * cols and parts.dtype == torch.float32
    _lib.call('relnet_wgrad_accumulate', parts.data_ptr(), S, rows, cols, _ptr(row_scale), grad.data_ptr(), _stream())


def _tn_ok(dy2d, x):
    """relnet_wgrad takes bf16 operands whose rows are 16-byte aligned."""
    return (dy2d.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy2d.stride(-1) == 1 and x.stride(-1) == 1 and
            dy2d.stride(0) % 8 == 0 and dy2d.shape[1] % 8 == 0 and x.shape[-1] % 8 == 0 and dy2d.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def linear_bwd(x2d, w, dy2d, need_dx=True, w_t=None, keep_splits=False, wgrad_to=None, relu_mask=None, bgrad_to=None):
    """y = x W^T + b  ->  (dx [P,K] in x's dtype | None, dW [N,K] fp32, db [N] fp32).
    wgrad_to = (grad [N,K] fp32 view, row_scale | None): the weight gradient is ACCUMULATED there by relnet_wgrad (one
    kernel straight from the row-major operands) and None is returned in its place.
    relu_mask [P,K] (x's dtype): dx is zeroed where relu_mask <= 0 inside the GEMM epilogue (x = relu(.) fused in the forward)."""
    dx = None
    gran = 64 if dy2d.dtype == torch.bfloat16 else 16
    N = dy2d.shape[1]
    dyp = dy2d
    if N % gran and (need_dx or wgrad_to is not None):        # e.g. the 81 + 8 outputs of cls_score | bbox_pred
        dyp = torch.zeros((dy2d.shape[0], pad_to(N, gran)), device=dy2d.device, dtype=dy2d.dtype)
        dyp[:, :N] = dy2d
    if need_dx:
        w_t = ops.transpose_2d(w, pad_cols_to=gran) if w_t is None else w_t      # [K, pad(N)], zero padded
        dx = ops.gemm_nt(dyp, w_t) if relu_mask is None else ops.gemm_nt(dyp, w_t, resid=relu_mask, relu=2)
    db = None
    if bgrad_to is not None:            # bias gradient accumulated straight into its slice of the flat buffer (one kernel)
        colsum_add(dy2d, bgrad_to)
    else:
        db = dy2d.float().sum(0)
    if wgrad_to is not None and _tn_ok(dyp, x2d):
        _wg_call(wgrad_to, dyp, x2d, cout=N)
        return dx, None, db
    dw = wgrad(dy2d, x2d, keep_splits)
    if wgrad_to is not None:
        _accumulate(wgrad_to, dw)
        dw = None
    return dx, dw, db


def _wg_call(wgrad_to, dy2d, x, cout=None, conv=None):
    """wgrad_to = (grad view, row_scale | None[, ops.WgradQueue]): queue the product for the stage's grouped launch, or run it now."""
    q = wgrad_to[2] if len(wgrad_to) > 2 else None
    if q is not None:
        q.add(dy2d, x, wgrad_to[0], wgrad_to[1], cout, conv)
    else:
        ops.wgrad_tn(dy2d, x, out=wgrad_to[0], row_scale=wgrad_to[1], cout=cout, conv=conv)


def _accumulate(wgrad_to, dw):
    """Fallback of the `wgrad_to` protocol for operands relnet_wgrad does not take: split-K partial sums -> += into the view."""
    g, scale = wgrad_to[0], wgrad_to[1]
    if dw.dim() == 3 and dw.is_contiguous() and dw.shape[1] * dw.shape[2] == g.numel() and g.shape[-1] % 4 == 0:
        wgrad_accumulate(dw, g, scale)
        return
    if dw.dim() == 3:
        dw = dw.sum(0)
    dw = dw.reshape(g.shape)
    if scale is not None:
        dw = dw * (scale * scale).view(-1, 1)
    g.add_(dw)


def pack_conv_dgrad_weight(w_oihw, dtype=torch.bfloat16, device='cuda'):
    """[Cout, Cin, R, S] -> [Cin, R*S*Cout] with the taps flipped: the data gradient of a stride-1 'same'
    convolution is the convolution of dY with this kernel (same padding / dilation)."""
    w = torch.flip(w_oihw, dims=(2, 3)).permute(1, 2, 3, 0)          # [Cin, R, S, Cout]
    return w.reshape(w.shape[0], -1).to(device=device, dtype=dtype).contiguous()


def conv1x1_bwd(x, w_packed, dy, stride=1, need_dx=True, w_t=None, dx_add=None, keep_splits=False, wgrad_to=None, relu_mask=None,
                out_mask=None, low_res=False):
    """x [B,H,W,Cin], dy [B,Ho,Wo,Cout] NHWC; w_packed [Cout,Cin].  -> (dx [B,H,W,Cin] | None, dW fp32).
    dx_add (stride 1 only): a second gradient of x's shape added in the GEMM epilogue (the shortcut branch).
    wgrad_to: see linear_bwd (the strided input rows are gathered inside relnet_wgrad: no sub-sampled copy of x).
    relu_mask (stride 1, instead of dx_add): x itself when x = relu(.) -- dx comes out already multiplied by (x > 0).
    out_mask (stride 1, WITH dx_add): dx = (dy W + dx_add) * (out_mask > 0) in one launch (relnet_gemm_nt_mask): the unit's input is the
    previous unit's ReLU output, so the result is that unit's masked output gradient (no separate relu_bwd pass).
    low_res (stride > 1): dx is returned at dy's resolution [B,Ho,Wo,Cin] (+ dx_add of that shape), not scattered to x's."""
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    P = dy.shape[0] * dy.shape[1] * dy.shape[2]
    tn = wgrad_to is not None and _tn_ok(dy.reshape(P, Cout), x) and x.is_contiguous()
    xs = x if (stride == 1 or tn) else x[:, ::stride, ::stride, :].contiguous()
    dy2 = dy.reshape(P, Cout)
    dx = None
    if need_dx:
        gran = 64 if dy.dtype == torch.bfloat16 else 16
        w_t = ops.transpose_2d(w_packed, pad_cols_to=gran) if w_t is None else w_t       # [Cin, pad(Cout)]
        dyp = dy2
        if Cout % gran:                           # e.g. the RPN's 24 + 48 output channels
            dyp = torch.zeros((P, w_t.shape[1]), device=dy.device, dtype=dy.dtype)
            dyp[:, :Cout] = dy2
        if stride == 1 and out_mask is not None:
            assert relu_mask is None and out_mask.shape == x.shape and out_mask.is_contiguous()
            dx = ops.gemm_nt_mask(dyp, w_t, out_mask.reshape(P, Cin), resid=None if dx_add is None else dx_add.reshape(P, Cin)).reshape(B, H, W, Cin)
        elif stride == 1 and relu_mask is not None:
            assert dx_add is None
            dx = ops.gemm_nt(dyp, w_t, resid=relu_mask.reshape(P, Cin), relu=2).reshape(B, H, W, Cin)
        elif stride == 1:
            dx = ops.gemm_nt(dyp, w_t, resid=None if dx_add is None else dx_add.reshape(P, Cin)).reshape(B, H, W, Cin)
        elif low_res:             # the caller sums both branches at the sampled resolution and scatters ONCE (strided_scatter)
            dx = ops.gemm_nt(dyp, w_t, resid=None if dx_add is None else dx_add.reshape(P, Cin)).reshape(dy.shape[0], dy.shape[1], dy.shape[2], Cin)
        else:
            assert dx_add is None
            dx = strided_scatter(ops.gemm_nt(dyp, w_t).reshape(dy.shape[0], dy.shape[1], dy.shape[2], Cin), (B, H, W, Cin), stride)
    if tn:
        if stride == 1:
            _wg_call(wgrad_to, dy2, x.reshape(P, Cin))
        else:
            _wg_call(wgrad_to, dy2, x, conv=(1, stride, 1, 0))
        return dx, None
    dw = wgrad(dy2, xs.reshape(P, Cin), keep_splits)
    if wgrad_to is not None:
        _accumulate(wgrad_to, dw)
        dw = None
    return dx, dw


_ZERO_OFF = {}


def conv3x3_bwd(x, w_dgrad_packed, dy, dil=1, need_dx=True, keep_splits=False, wgrad_to=None, cout=None, relu_mask=None):
    """3x3, stride 1, pad = dil.  x [B,H,W,Cin], dy [B,H,W,Cout]; w_dgrad_packed from pack_conv_dgrad_weight.
    -> (dx | None, dW [Cout, 9*Cin] fp32 in pack_conv_weight order).  wgrad_to: see linear_bwd (implicit im2col inside
    relnet_wgrad: the [pixels][9 Cin] patch matrix is never written); cout: real output channels when dy is zero padded."""
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    dx = None
    if need_dx:       # relu_mask: dx * (x > 0) in the convolution's epilogue (x = relu(.) of the forward)
        dx = ops.conv2d_nhwc(dy, w_dgrad_packed, None, ksize=3, stride=1, pad=dil, dil=dil,
                             relu=2 if relu_mask is not None else False, resid=relu_mask)
    dy2 = dy.reshape(B * H * W, Cout)
    if wgrad_to is not None and _tn_ok(dy2, x) and x.is_contiguous():
        _wg_call(wgrad_to, dy2, x, cout=cout, conv=(3, 1, dil, dil))
        return dx, None
    key = (B, H, W, x.device)
    if key not in _ZERO_OFF:
        _ZERO_OFF[key] = torch.zeros((B, H, W, 18), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)
    col, _ = ops.deformable_im2col(x.permute(0, 3, 1, 2), _ZERO_OFF[key], 3, 1, dil, dil, 1)    # [P, 9*Cin] patches
    dw = wgrad(dy2, col, keep_splits)
    if wgrad_to is not None:
        if cout is not None:
            dw = (dw.sum(0) if dw.dim() == 3 else dw)[:cout]
        _accumulate(wgrad_to, dw)
        dw = None
    return dx, dw
